"""Initialisation optimisation: fit latents to observed futures (reference src/utils/init_optim.py:11-68).
Same signature and return values; one closure (decode -> TgtMatchingLoss -> backward) + one Adam step per
iteration.  Progress bars / per-term ``.item()`` prints are dropped (they are host syncs)."""
import torch
import torch.optim as optim

from .graphed import GraphedIteration, adam_kwargs, graph_mode


def run_init_optim(cur_z, init_traj, traj_vis, lr, loss_weights, model, scene_graph, map_env, map_idx, num_iters,
                   embed_info, prior_distrib, log=None):
    from ..losses.adv_gen_nusc import TgtMatchingLoss
    vis_idx = torch.nonzero((traj_vis == 1.0).reshape(-1)).flatten()        # once: x[mask] would synchronise every iteration
    init_traj = model.get_normalizer().unnormalize(init_traj).reshape(-1, init_traj.size(-1)).index_select(0, vis_idx)
    cur_z = cur_z.clone().detach()
    cur_z.requires_grad = True
    graphed = graph_mode(cur_z.shape[0], cur_z.device, log)
    init_optim = optim.Adam([cur_z], lr=lr, **adam_kwargs(graphed))
    match_loss = TgtMatchingLoss({k[5:]: v for k, v in loss_weights.items() if k[:5] == 'init_'})
    def iteration():
        init_optim.zero_grad()
        pred = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env)['future_pred']
        pred = model.get_normalizer().unnormalize(pred).reshape(-1, pred.size(-1)).index_select(0, vis_idx)
        loss_dict = match_loss(pred, init_traj, cur_z, prior_distrib)
        loss_dict['loss'].backward()
        if log is not None:
            log(loss_dict, cur_z)
        init_optim.step()
        return loss_dict['loss']
    it = GraphedIteration(iteration, graphed)
    for _ in range(num_iters):
        it()
    with torch.no_grad():
        init_decoder_out = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env)
    return cur_z, init_decoder_out['future_pred'].clone().detach(), init_decoder_out
