"""Scenario-generation helpers on the hot path (reference src/utils/scenario_gen.py:19-28)."""
import torch


def detach_embed_info(embed_info_attached):
    out = {}
    for k, v in embed_info_attached.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach()
        elif isinstance(v, tuple):
            out[k] = (v[0].detach(), v[1].detach())
    return out


def determine_feasibility_nusc(samples, normalizer, feasibility_thresh, feasibility_time=0, feasibility_vel=0.0,
                               feasibility_infront_min=None, check_non_drivable_separation=True, map_env=None, map_idx=None):
    """Is a scene a plausible seed for scenario generation?  (reference src/utils/scenario_gen.py:30-107)
    ``samples`` (NA,NS,FT,4) NORMALISED futures of ONE scene, agent 0 = ego.  An agent is feasible if, in some sample
    and step >= ``feasibility_time``, it comes within ``feasibility_thresh`` metres of the ego (optionally only counting
    steps where it is in front of the ego), the closest approach is not across non-drivable area, and it moves faster
    than ``feasibility_vel`` per step somewhere.  Returns (feasible, closest step, closest distance), each (NA-1,);
    three Nones for an ego-only scene."""
    import torch
    from ..datasets import nuscenes_utils as nutils
    if samples.size(0) == 1:
        return None, None, None
    if feasibility_infront_min is not None and not (-1 <= feasibility_infront_min <= 1):
        raise AssertionError('feasibility_infront_min must lie in [-1, 1]')
    s = normalizer.unnormalize(samples)
    ego, oth = s[0:1], s[1:]
    NA, NS, FT, _ = oth.size()
    t0 = feasibility_time
    dist = torch.norm(ego[..., :2] - oth[..., :2], dim=-1)[:, :, t0:]                 # (NA, NS, FT')
    if feasibility_infront_min is not None:
        to_agent = oth[:, :, t0:, :2] - ego[:, :, t0:, :2]
        to_agent = to_agent / torch.norm(to_agent, dim=-1, keepdim=True)
        infront = torch.sum(to_agent * ego[:, :, t0:, 2:4], dim=-1) >= feasibility_infront_min
        dist = torch.where(infront, dist, torch.full_like(dist, float('inf')))
    best_over_samples, best_sample = torch.min(dist, dim=1)                           # (NA, FT')
    feasible_dist, step = torch.min(best_over_samples, dim=1)
    step = step + t0
    feasible = (dist < feasibility_thresh).sum(dim=[1, 2]) > 0
    if check_non_drivable_separation:
        ar = torch.arange(NA, device=samples.device)
        samp = best_sample[ar, step - t0]
        agent_xy = oth[ar, samp, step][:, :2]
        ego_xy = ego.expand(NA, NS, FT, 4)[ar, samp, step][:, :2]
        blocked = nutils.check_line_layer(map_env.nusc_raster[:, 0], map_env.nusc_dx, agent_xy, ego_xy, map_idx.expand(NA))
        feasible = torch.logical_and(feasible, ~blocked)
    vel = torch.norm(oth[:, :, 1:, :2] - oth[:, :, :-1, :2], dim=-1)
    feasible = torch.logical_and(feasible, vel.amax(dim=(1, 2)) > feasibility_vel)
    return feasible, step, feasible_dist


def prepare_output_dict(scene_graph, map_idx, map_env, dt, model, init_fut_traj, adv_fut_traj, sol_fut_traj=None,
                        attack_agt=None, attack_t=None, adv_z=None, sol_z=None, prior_distrib=None,
                        attack_bike_params=None, internal_ego_traj=None):
    """The scenario wire format consumed by the eval / clustering / viz tools (reference src/utils/scenario_gen.py:189-254):
    a JSON-ready dict with keys ``N, dt, map, lw, sem, past, fut_init, fut_adv`` and, when given, ``fut_internal_ego,
    fut_sol, attack_agt, attack_t, z_adv, z_sol, z_prior{mean,var}, attack_bike_prof``.  Trajectories and ``lw`` are
    stored UNNORMALISED as nested lists."""
    nrm = model.get_normalizer()

    def world(t):
        return nrm.unnormalize(t).cpu().numpy().tolist()

    def plain(t):
        return t.detach().cpu().numpy().tolist()
    out = {'N': int(init_fut_traj.size(0)), 'dt': dt, 'map': map_env.map_list[map_idx]}
    out['lw'] = plain(model.get_att_normalizer().unnormalize(scene_graph.lw))
    out['sem'] = plain(scene_graph.sem)
    out['past'] = world(scene_graph.past_gt)
    out['fut_init'] = world(init_fut_traj)
    out['fut_adv'] = world(adv_fut_traj)
    if internal_ego_traj is not None:
        out['fut_internal_ego'] = world(internal_ego_traj)
    if sol_fut_traj is not None:
        out['fut_sol'] = world(sol_fut_traj)
    if attack_agt is not None:
        out['attack_agt'] = int(attack_agt)
    if attack_t is not None:
        out['attack_t'] = int(attack_t)
    if adv_z is not None:
        out['z_adv'] = plain(adv_z)
    if sol_z is not None:
        out['z_sol'] = plain(sol_z)
    if prior_distrib is not None:
        out['z_prior'] = {'mean': plain(prior_distrib[0]), 'var': plain(prior_distrib[1])}
    if attack_bike_params is not None:
        out['attack_bike_prof'] = plain(attack_bike_params)
    return out
