"""Solution optimisation loop (reference src/utils/sol_optim.py:19-123)."""
import torch
import torch.optim as optim

from .adv_gen_optim import collate_tgt_other_z


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
    sol_optim = optim.Adam([tgt_z, other_z_all], lr=lr)
    w = {k[4:]: v for k, v in loss_weights.items() if k[:4] == 'sol_'}
    avoid_loss = AvoidCollLoss(w, model.get_att_normalizer().unnormalize(scene_graph.lw), map_idx[scene_graph.batch],
                               map_env, tgt_z.clone().detach(), veh_coll_buffer=0.5, single_veh_idx=0, ptr=scene_graph.ptr)
    match_loss = TgtMatchingLoss(w)
    for _ in range(num_iters):
        sol_optim.zero_grad()
        z_a = collate_tgt_other_z(scene_graph, tgt_z, other_z_all.detach())
        out_a = model.decode_embedding(z_a, embed_info, scene_graph, map_idx, map_env, nfuture=future_len)
        z_b = collate_tgt_other_z(scene_graph, tgt_z.detach(), other_z_all)
        out_b = model.decode_embedding(z_b, embed_info, scene_graph, map_idx, map_env)
        tgt_pred = unn(out_a['future_pred']).transpose(0, 1).reshape(NA, future_len, 4)
        lt = avoid_loss(tgt_pred, tgt_z, tgt_prior_distrib)
        loss_dict = {'tgt_' + k: v for k, v in lt.items()}
        lo = match_loss(unn(out_b['future_pred'])[~tgt_mask], other_match, other_z_all, other_prior_distrib)
        loss_dict.update({'other_' + k: v for k, v in lo.items()})
        loss = loss_dict['tgt_loss'] + loss_dict['other_loss']
        loss.backward()
        if log is not None:
            log(loss_dict)
        sol_optim.step()
    cur_z = collate_tgt_other_z(scene_graph, tgt_z, other_z_all)
    with torch.no_grad():
        sol_decoder_out = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env)
    sol_result_traj = sol_decoder_out['future_pred'].clone().detach()
    sol_result_traj[~tgt_mask] = model.get_normalizer().normalize(other_match)
    return cur_z, sol_result_traj, sol_decoder_out
