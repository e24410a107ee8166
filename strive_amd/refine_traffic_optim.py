"""Collision-refinement optimisation (reference src/refine_traffic_optim.py:146-226, Adam and LBFGS branches).
Same signature; ``z_init`` may be injected (the reference samples it unseeded from the prior)."""
import torch
import torch.optim as optim

from .losses.adv_gen_nusc import AvoidCollLoss
from .utils.graphed import GraphedIteration, adam_kwargs, graph_mode
from .utils.scenario_gen import detach_embed_info


def refine_traffic_optim(scene_graph, map_idx, map_env, model, loss_weights, num_iters, samp_future_len,
                         save_future_len, optim_use_adam, lr, z_init=None, log=None):
    with torch.no_grad():
        if z_init is None:
            sample_pred = model.sample_batched(scene_graph, map_idx, map_env, 1, include_mean=False)
            init_future_pred = sample_pred['future_pred'][:, 0]
            z_init = sample_pred['z_samp'][:, 0].clone().detach()
        else:
            init_future_pred = None
        embed_info = detach_embed_info(model.embed(scene_graph, map_idx, map_env))
        if init_future_pred is None:
            init_future_pred = model.decode_embedding(z_init, embed_info, scene_graph, map_idx, map_env)['future_pred']
    cur_z = z_init.clone().detach()
    cur_z.requires_grad = True
    graphed = optim_use_adam and graph_mode(cur_z.shape[0], cur_z.device, log)
    if optim_use_adam:
        scene_optim = optim.Adam([cur_z], lr=lr, **adam_kwargs(graphed))
    else:       # --optim_use_lbfgs (reference :53-55, 170-173): 20 inner iterations with a strong-Wolfe line search per step
        scene_optim = optim.LBFGS([cur_z], max_iter=20, lr=lr, line_search_fn='strong_wolfe')
    avoid_loss = AvoidCollLoss(loss_weights, model.get_att_normalizer().unnormalize(scene_graph.lw),
                               map_idx[scene_graph.batch], map_env, cur_z.clone().detach(), veh_coll_buffer=0.2)
    def closure():
        scene_optim.zero_grad()
        pred = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env, nfuture=samp_future_len)
        loss_dict = avoid_loss(model.get_normalizer().unnormalize(pred['future_pred']), cur_z, embed_info['prior_out'])
        loss_dict['loss'].backward()
        if log is not None:
            log(loss_dict, cur_z)
        return loss_dict['loss']
    def iteration():
        loss = closure()
        scene_optim.step()
        return loss
    # (one scene per batch is the shipped operating point: the iteration is replayed as a HIP graph there, utils/graphed.py)
    it = GraphedIteration(iteration, graphed)
    for _ in range(num_iters):
        if optim_use_adam:
            it()
        else:
            scene_optim.step(closure)       # (torch's LBFGS reads the loss on the host for its line search, like the reference's run)
    with torch.no_grad():
        final = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env, nfuture=save_future_len)
    return init_future_pred, cur_z, final['future_pred'].unsqueeze(1).clone().detach(), embed_info
