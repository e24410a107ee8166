"""Solution optimisation loop (reference src/utils/sol_optim.py:19-123)."""
import torch
import torch.optim as optim

from .adv_gen_optim import collate_tgt_other_z, _collate_index, two_rollouts
from .graphed import GraphedIteration, adam_kwargs, graph_mode


def run_find_solution_optim(cur_z, final_result_traj, future_len, lr, loss_weights, model, scene_graph, map_env,
                            map_idx, num_iters, embed_info, tgt_prior_distrib, other_prior_distrib, log=None):
    from ..losses.adv_gen_nusc import AvoidCollLoss, TgtMatchingLoss
    dev = cur_z.device
    B = map_idx.size(0)
    NA = final_result_traj.size(0)
    unn = model.get_normalizer().unnormalize
    tgt_mask = torch.zeros((NA,), dtype=torch.bool, device=dev)
    tgt_mask[scene_graph.ptr[:-1].to(dev)] = True
    other_match = unn(final_result_traj[:, 0][~tgt_mask])
    other_match = other_match.view(other_match.size(0), 1, other_match.size(1), 4)
    tgt_z = tgt_prior_distrib[0].view(B, 1, -1).clone().detach()
    tgt_z.requires_grad = True
    other_z_all = cur_z[~tgt_mask].view(NA - B, 1, -1).clone().detach()
    other_z_all.requires_grad = True
    graphed = graph_mode(NA, dev, log, rollouts=2)
    sol_optim = optim.Adam([tgt_z, other_z_all], lr=lr, **adam_kwargs(graphed))
    _, _, other_idx = _collate_index(scene_graph, dev)
    w = {k[4:]: v for k, v in loss_weights.items() if k[:4] == 'sol_'}
    avoid_loss = AvoidCollLoss(w, model.get_att_normalizer().unnormalize(scene_graph.lw), map_idx[scene_graph.batch],
                               map_env, tgt_z.clone().detach(), veh_coll_buffer=0.5, single_veh_idx=0, ptr=scene_graph.ptr)
    match_loss = TgtMatchingLoss(w)
    def iteration():
        sol_optim.zero_grad()
        z_a = collate_tgt_other_z(scene_graph, tgt_z, other_z_all.detach())
        z_b = collate_tgt_other_z(scene_graph, tgt_z.detach(), other_z_all)
        out_a, out_b = two_rollouts(model, embed_info, scene_graph, map_idx, map_env, z_a, dict(nfuture=future_len), z_b, dict(),
                                    overlap=not graphed, same_values=True)      # (a replayed graph keeps one stream, see AdvClosure)
        tgt_pred = unn(out_a['future_pred']).transpose(0, 1).reshape(NA, future_len, 4)
        lt = avoid_loss(tgt_pred, tgt_z, tgt_prior_distrib)
        lo = match_loss(unn(out_b['future_pred']).index_select(0, other_idx), other_match, other_z_all, other_prior_distrib)
        loss = lt['loss'] + lo['loss']
        loss.backward()
        if log is not None:
            loss_dict = {'tgt_' + k: v for k, v in lt.items()}
            loss_dict.update({'other_' + k: v for k, v in lo.items()})
            log(loss_dict, tgt_z, other_z_all)
        sol_optim.step()
        return loss
    it = GraphedIteration(iteration, graphed)
    for _ in range(num_iters):
        it()
    cur_z = collate_tgt_other_z(scene_graph, tgt_z, other_z_all)
    with torch.no_grad():
        sol_decoder_out = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env)
    sol_result_traj = sol_decoder_out['future_pred'].clone().detach()
    sol_result_traj[~tgt_mask] = model.get_normalizer().normalize(other_match)
    return cur_z, sol_result_traj, sol_decoder_out


def compute_sol_success(final_result_traj, model, scene_graph, map_env, map_idx, use_map_coll=True):
    """Did the solution optimisation find a collision-free ego future?  All inputs NORMALISED; agent 0 of
    ``final_result_traj`` (NA,1,FT,4) is the solution (reference src/utils/sol_optim.py:126-165)."""
    import numpy as np
    from ..losses.adv_gen_nusc import check_single_veh_coll
    from ..losses.traffic_model import compute_coll_rate_env
    nrm, att = model.get_normalizer(), model.get_att_normalizer()
    sol_fut = nrm.unnormalize(final_result_traj[0, 0])
    other_fut = nrm.unnormalize(final_result_traj[1:, 0])
    coll_all, _ = check_single_veh_coll(sol_fut, att.unnormalize(scene_graph.lw[0]), other_fut, att.unnormalize(scene_graph.lw[1:]))
    impossible = bool(np.sum(coll_all) > 0)
    if use_map_coll:
        env = compute_coll_rate_env(scene_graph, map_idx, final_result_traj.contiguous(), map_env, nrm, att, ego_only=True)
        impossible = impossible or bool(env['did_collide'].cpu().numpy()[0, 0])
    return not impossible
