"""Adversarial optimisation loop (reference src/utils/adv_gen_optim.py:19-211), open-loop 'ego' planner
mode.  The rule-based closed-loop planner (planner_name == 'hardcode') is a CPU numpy component outside
this round's scope and raises NotImplementedError."""
import torch
import torch.optim as optim


def collate_tgt_other_z(scene_graph, tgt_z, other_z):
    """Interleave ego latents (B,[NS,]D) and the others' (NA-B,[NS,]D) into graph order (reference :19-36),
    as one index_select instead of a per-scene concatenation loop."""
    ptr = scene_graph.ptr
    B = tgt_z.size(0)
    NA = int(ptr[-1])
    dev = other_z.device
    ego = ptr[:-1].to(dev)
    is_ego = torch.zeros((NA,), dtype=torch.bool, device=dev)
    is_ego[ego] = True
    src = torch.empty((NA,), dtype=torch.long, device=dev)
    src[is_ego] = torch.arange(B, device=dev)
    src[~is_ego] = torch.arange(NA - B, device=dev) + B
    return torch.cat([tgt_z, other_z], dim=0).index_select(0, src)


def run_adv_gen_optim(cur_z, lr, loss_weights, model, scene_graph, map_env, map_idx, num_iters, embed_info,
                      planner_name, tgt_prior_distrib, other_prior_distrib, feasibility_time, feasibility_infront_min,
                      planner=None, planner_viz_out=None, attack_agt_idx=None, future_len=None, veh_coll_buffer=0.1,
                      log=None):
    from ..losses.adv_gen_nusc import TgtMatchingLoss, AdvGenLoss
    if planner_name != 'ego':
        raise NotImplementedError("only planner_name='ego' (open loop) is implemented; the rule-based planner is "
                                  "a CPU component outside the HIP hot path")
    dev = cur_z.device
    NA = cur_z.size(0)
    ego_inds = scene_graph.ptr[:-1].to(dev)
    ego_mask = torch.zeros((NA,), dtype=torch.bool, device=dev)
    ego_mask[ego_inds] = True
    if attack_agt_idx is not None:
        attack_agt_idx = torch.as_tensor(attack_agt_idx).to(ego_inds) + ego_inds
    if future_len is None:
        future_len = model.FT
    tgt_z = cur_z[ego_mask].clone().detach()
    tgt_z.requires_grad = True
    other_z_all = cur_z[~ego_mask].clone().detach()
    other_z_all.requires_grad = True
    cur_z = collate_tgt_other_z(scene_graph, tgt_z, other_z_all)
    adv_optim = optim.Adam([tgt_z, other_z_all], lr=lr)
    unn = model.get_normalizer().unnormalize
    tgt_loss = TgtMatchingLoss(loss_weights)
    adv_loss = AdvGenLoss(loss_weights, model.get_att_normalizer().unnormalize(scene_graph.lw),
                          map_idx[scene_graph.batch], map_env, cur_z[~ego_mask].clone().detach(), scene_graph.ptr,
                          veh_coll_buffer=veh_coll_buffer, crash_loss_min_time=feasibility_time,
                          crash_loss_min_infront=feasibility_infront_min)
    planner_fut = scene_graph.future_gt[ego_mask][:, :, :4]
    assert planner_fut.size(1) == future_len
    for _ in range(num_iters):
        adv_optim.zero_grad()
        z_a = collate_tgt_other_z(scene_graph, tgt_z, other_z_all.clone().detach())
        z_b = collate_tgt_other_z(scene_graph, tgt_z.clone().detach(), other_z_all)
        out_a = model.decode_embedding(z_a, embed_info, scene_graph, map_idx, map_env, ext_future=planner_fut,
                                       nfuture=future_len)
        out_b = model.decode_embedding(z_b, embed_info, scene_graph, map_idx, map_env, ext_future=planner_fut,
                                       nfuture=future_len)
        lt = tgt_loss(unn(out_a['future_pred'][ego_mask]), unn(planner_fut), tgt_z, tgt_prior_distrib)
        la = adv_loss(unn(out_b['future_pred']), unn(planner_fut), other_z_all, other_prior_distrib,
                      attack_agt_idx=attack_agt_idx)
        loss_dict = {'tgt_match_' + k: v for k, v in lt.items()}
        loss_dict.update({'adv_' + k: v for k, v in la.items()})
        loss = loss_dict['tgt_match_loss'] + loss_dict['adv_loss']
        loss.backward()
        if log is not None:
            log(loss_dict)
        adv_optim.step()

    cur_z = collate_tgt_other_z(scene_graph, tgt_z, other_z_all)
    with torch.no_grad():
        final_decoder_out = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env, nfuture=future_len)
    final_result_traj = final_decoder_out['future_pred'].unsqueeze(1).clone().detach()
    final_result_traj[ego_inds, torch.zeros_like(ego_inds)] = scene_graph.future_gt[ego_mask][:, :, :4]
    tgt_traj = final_result_traj[ego_inds, torch.zeros_like(ego_inds)]
    with torch.no_grad():
        fin = adv_loss(unn(final_decoder_out['future_pred']), unn(tgt_traj), cur_z[~ego_mask].clone().detach(),
                       other_prior_distrib, return_mins=True)
    cur_min_agt = cur_min_t = None
    if 'min_agt' in fin:
        cur_min_agt = fin['min_agt'] + scene_graph.ptr[:-1].cpu().numpy()
    if 'min_t' in fin:
        cur_min_t = fin['min_t']
    return cur_z, final_result_traj, final_decoder_out, cur_min_agt, cur_min_t


def compute_adv_gen_success(final_result_traj, model, scene_graph, attack_agt):
    """Did the scenario make the attacker collide with the planner?  All inputs NORMALISED; ``final_result_traj``
    (NA,1,FT,4) with agent 0 = the planner's true reaction (reference src/utils/adv_gen_optim.py:214-235)."""
    from ..losses.adv_gen_nusc import check_single_veh_coll
    nrm, att = model.get_normalizer(), model.get_att_normalizer()
    planner_fut = nrm.unnormalize(final_result_traj[0, 0])
    other_fut = nrm.unnormalize(final_result_traj[1:, 0])
    planner_lw = att.unnormalize(scene_graph.lw[0])
    other_lw = att.unnormalize(scene_graph.lw[1:])
    coll_all, _ = check_single_veh_coll(planner_fut, planner_lw, other_fut, other_lw)
    return bool(coll_all[attack_agt - 1])
