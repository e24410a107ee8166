"""ORACLE (test infrastructure only) -- the four latent-optimisation loops.

CPU restatement of the Adam-over-latents drivers that call the hot path hundreds of times per
scene batch.  Each iteration is exactly one closure (zero_grad -> rollout(s) -> losses ->
backward) followed by one ``Adam.step()`` with torch defaults, like the reference.  Progress
bars, prints and ``.item()`` logging are dropped; a per-iteration trace of latents and loss
entries can be recorded for the G6 golden fixture.

Two test hooks, not in the reference, let a test evaluate ONE closure of a loop at latents another implementation visited
(``num_iters = 1``): ``init_z`` -- the latents the loop STARTED from, which the init-z loss terms pull towards (a loop normally
takes them from its own starting point) -- and ``crop_poses`` -- the rollout poses at which the raster is cropped
(OracleTrafficModel.decode); ``trace[-1]['crop_flips']`` then says where the forced poses changed a crop.
"""
import torch

from .losses import AvoidColl, AdvGen, tgt_matching_loss


def collate_tgt_other_z(ptr, tgt_z, other_z):
    """Interleave per-scene ego latents and the other agents' latents back into graph order
    (ego first in every scene).  (reference src/utils/adv_gen_optim.py:19-36)"""
    parts = []
    prev = 0
    for b in range(tgt_z.shape[0]):
        n = int(ptr[b + 1] - ptr[b]) - 1
        parts.append(tgt_z[b:b + 1])
        parts.append(other_z[prev:prev + n])
        prev += n
    return torch.cat(parts, dim=0)


def _trace_entry(z_list, loss_dict):
    ent = {'z': [z.detach().clone() for z in z_list],
           'grad': [z.grad.detach().clone() if z.grad is not None else torch.zeros_like(z) for z in z_list]}
    for k, v in loss_dict.items():
        if torch.is_tensor(v):
            ent[k] = v.detach().clone()
    return ent


def refine_loop(model, g, map_idx, map_env, embed_info, z_init, weights, num_iters, lr, nfuture,
                veh_coll_buffer=0.2, trace=None, init_z=None, crop_poses=None):
    """Collision-refinement optimisation.  (reference src/refine_traffic_optim.py:146-226, Adam branch)"""
    z = z_init.clone().detach()
    z.requires_grad = True
    opt = torch.optim.Adam([z], lr=lr)
    loss_fn = AvoidColl(weights, model.get_att_normalizer().unnormalize(g.lw), map_idx[g.batch], map_env,
                        (z if init_z is None else init_z).clone().detach(), veh_coll_buffer=veh_coll_buffer)
    for _ in range(num_iters):
        opt.zero_grad()
        pred = model.decode_embedding(z, embed_info, g, map_idx, map_env, nfuture=nfuture, crop_poses=crop_poses)['future_pred']
        ld = loss_fn(model.get_normalizer().unnormalize(pred), z, embed_info['prior_out'])
        ld['loss'].backward()
        if trace is not None:
            ent = _trace_entry([z], ld)
            ent['grad'] = z.grad.detach().clone()
            ent['crop_flips'] = model.last_crop_flips
            trace.append(ent)
        opt.step()
    return z.detach()


def refine_fn(model, g, map_idx, map_env, weights, num_iters, samp_future_len, save_future_len, use_adam, lr, eps):
    """The whole refine_traffic_optim() of the reference (src/refine_traffic_optim.py:146-226): one prior sample (``eps (1,NA,D)``
    injected: the reference draws it unseeded), embed, then Adam -- one closure + one step per iteration -- or LBFGS(max_iter 20,
    strong-Wolfe line search) -- ``step(closure)`` per iteration -- on the AvoidColl objective, and the final rollout.
    Returns (init_future_pred, z, result_traj (NA,1,save_future_len,4))."""
    with torch.no_grad():
        samp = model.sample_batched(g, map_idx, map_env, eps, include_mean=False)
        embed_info = model.embed(g, map_idx, map_env)
    init_future_pred = samp['future_pred'][:, 0]
    z = samp['z_samp'][:, 0].clone().detach()
    z.requires_grad = True
    if use_adam:
        opt = torch.optim.Adam([z], lr=lr)
    else:
        opt = torch.optim.LBFGS([z], max_iter=20, lr=lr, line_search_fn='strong_wolfe')
    loss_fn = AvoidColl(weights, model.get_att_normalizer().unnormalize(g.lw), map_idx[g.batch], map_env, z.clone().detach(),
                        veh_coll_buffer=0.2)

    def closure():
        opt.zero_grad()
        pred = model.decode_embedding(z, embed_info, g, map_idx, map_env, nfuture=samp_future_len)['future_pred']
        ld = loss_fn(model.get_normalizer().unnormalize(pred), z, embed_info['prior_out'])
        ld['loss'].backward()
        return ld['loss']
    for _ in range(num_iters):
        if use_adam:
            closure()
            opt.step()
        else:
            opt.step(closure)
    with torch.no_grad():
        out = model.decode_embedding(z, embed_info, g, map_idx, map_env, nfuture=save_future_len)['future_pred']
    return init_future_pred, z.detach(), out.unsqueeze(1).clone().detach()


def init_loop(model, g, map_idx, map_env, embed_info, z_init, init_traj, traj_vis, weights, num_iters, lr,
              prior_out, trace=None, crop_poses=None):
    """Fit latents to observed futures.  (reference src/utils/init_optim.py:11-68)"""
    tgt = model.get_normalizer().unnormalize(init_traj)[traj_vis == 1.0]
    z = z_init.clone().detach()
    z.requires_grad = True
    opt = torch.optim.Adam([z], lr=lr)
    w = {k[5:]: v for k, v in weights.items() if k[:5] == 'init_'}
    for _ in range(num_iters):
        opt.zero_grad()
        pred = model.decode_embedding(z, embed_info, g, map_idx, map_env, crop_poses=crop_poses)['future_pred']
        pred = model.get_normalizer().unnormalize(pred)[traj_vis == 1.0]
        ld = tgt_matching_loss(w, pred, tgt, z, prior_out)
        ld['loss'].backward()
        if trace is not None:
            trace.append(_trace_entry([z], ld))
            trace[-1]['crop_flips'] = model.last_crop_flips
        opt.step()
    return z.detach()


def adv_loop(model, g, map_idx, map_env, embed_info, z_init, weights, num_iters, lr, tgt_prior, other_prior,
             feasibility_time=0, feasibility_infront_min=None, attack_agt_idx=None, future_len=None,
             veh_coll_buffer=0.1, trace=None, planner=None, init_z=None, crop_poses=None):
    """Adversarial optimisation in open-loop ('ego' planner) mode: the planner trajectory is the
    ego's ground-truth future, injected into both rollouts as ``ext_future``; two rollouts with
    complementary detach so each latent group only sees its own loss.
    (reference src/utils/adv_gen_optim.py:39-211, planner_name == 'ego')"""
    ptr = g.ptr
    NA = z_init.shape[0]
    ego_mask = torch.zeros((NA,), dtype=torch.bool)
    ego_mask[ptr[:-1]] = True
    if attack_agt_idx is not None:
        attack_agt_idx = torch.as_tensor(attack_agt_idx).to(ptr) + ptr[:-1]
    FT = model.FT if future_len is None else future_len
    tgt_z = z_init[ego_mask].clone().detach()
    tgt_z.requires_grad = True
    other_z = z_init[~ego_mask].clone().detach()
    other_z.requires_grad = True
    opt = torch.optim.Adam([tgt_z, other_z], lr=lr)
    cur = collate_tgt_other_z(ptr, tgt_z, other_z)
    adv = AdvGen(weights, model.get_att_normalizer().unnormalize(g.lw), map_idx[g.batch], map_env,
                 (cur[~ego_mask] if init_z is None else init_z).clone().detach(), ptr, veh_coll_buffer=veh_coll_buffer,
                 crash_loss_min_time=feasibility_time, crash_loss_min_infront=feasibility_infront_min)
    planner_fut = g.future_gt[ego_mask][:, :, :4]
    unn = model.get_normalizer().unnormalize
    if planner is not None:
        # closed loop (reference :90-103, 133-139, planner_name == 'hardcode'): `planner` is any object with the
        # reference's reset / rollout protocol; it reacts to the current rollout of the other agents in every
        # iteration, nothing is injected into the decoder and the adversarial loss sees the model's own ego prediction
        import numpy as np
        B = ptr.shape[0] - 1
        planner.reset(unn(g.past_gt[:, -1, :]), model.get_att_normalizer().unnormalize(g.lw), g.batch, B, map_idx)
        agt_ptr = (ptr - torch.arange(B + 1)).numpy()
        plan_t = np.linspace(model.dt, model.dt * FT, FT)
    for _ in range(num_iters):
        opt.zero_grad()
        z_a = collate_tgt_other_z(ptr, tgt_z, other_z.clone().detach())
        z_b = collate_tgt_other_z(ptr, tgt_z.clone().detach(), other_z)
        ext = planner_fut if planner is None else None
        pa = model.decode_embedding(z_a, embed_info, g, map_idx, map_env, ext_future=ext, nfuture=FT, crop_poses=crop_poses)
        flips = model.last_crop_flips
        pb = model.decode_embedding(z_b, embed_info, g, map_idx, map_env, ext_future=ext, nfuture=FT, crop_poses=crop_poses)
        if planner is not None:
            agt = unn(pa['future_pred'][~ego_mask]).detach().cpu().numpy()
            planner_fut = model.get_normalizer().normalize(planner.rollout(agt, plan_t, agt_ptr, plan_t, control_all=False).to(g.future_gt))
            adv_tgt = pb['future_pred'][ego_mask]
        else:
            adv_tgt = planner_fut
        lt = tgt_matching_loss(weights, unn(pa['future_pred'][ego_mask]), unn(planner_fut), tgt_z, tgt_prior)
        la = adv(unn(pb['future_pred']), unn(adv_tgt), other_z, other_prior, attack_agt_idx=attack_agt_idx)
        ld = {'tgt_match_' + k: v for k, v in lt.items()}
        ld.update({'adv_' + k: v for k, v in la.items()})
        loss = ld['tgt_match_loss'] + ld['adv_loss']
        loss.backward()
        if trace is not None:
            trace.append(_trace_entry([tgt_z, other_z], ld))
            trace[-1]['crop_flips'] = flips
        opt.step()
    return collate_tgt_other_z(ptr, tgt_z, other_z).detach()


def sol_loop(model, g, map_idx, map_env, embed_info, cur_z, final_result_traj, future_len, weights, num_iters,
             lr, tgt_prior, other_prior, trace=None, init_z=None, start_z=None, crop_poses=None):
    """Solution optimisation: ego avoids collisions (rollout of ``future_len`` steps through the
    multi-sample code path with NS=1) while the others keep matching the adversarial scenario.
    (reference src/utils/sol_optim.py:19-123)"""
    ptr = g.ptr
    B = map_idx.shape[0]
    NA = final_result_traj.shape[0]
    tgt_mask = torch.zeros((NA,), dtype=torch.bool)
    tgt_mask[ptr[:-1]] = True
    unn = model.get_normalizer().unnormalize
    other_match = unn(final_result_traj[:, 0][~tgt_mask])
    other_match = other_match.view(other_match.shape[0], 1, other_match.shape[1], 4)
    tgt_z = (tgt_prior[0] if start_z is None else start_z[0]).view(B, 1, -1).clone().detach()      # (start_z: test hook, (ego, others))
    tgt_z.requires_grad = True
    other_z = (cur_z[~tgt_mask] if start_z is None else start_z[1]).view(NA - B, 1, -1).clone().detach()
    other_z.requires_grad = True
    opt = torch.optim.Adam([tgt_z, other_z], lr=lr)
    w = {k[4:]: v for k, v in weights.items() if k[:4] == 'sol_'}
    avoid = AvoidColl(w, model.get_att_normalizer().unnormalize(g.lw), map_idx[g.batch], map_env,
                      (tgt_z if init_z is None else init_z.view(B, 1, -1)).clone().detach(), veh_coll_buffer=0.5, single_veh_idx=0, ptr=ptr)
    for _ in range(num_iters):
        opt.zero_grad()
        z_a = collate_tgt_other_z(ptr, tgt_z, other_z.detach())
        pa = model.decode_embedding(z_a, embed_info, g, map_idx, map_env, nfuture=future_len, crop_poses=crop_poses)
        flips = model.last_crop_flips
        z_b = collate_tgt_other_z(ptr, tgt_z.detach(), other_z)
        pb = model.decode_embedding(z_b, embed_info, g, map_idx, map_env, crop_poses=crop_poses)
        tp = unn(pa['future_pred']).transpose(0, 1).reshape(NA, future_len, 4)
        lt = avoid(tp, tgt_z, tgt_prior)
        ld = {'tgt_' + k: v for k, v in lt.items()}
        lo = tgt_matching_loss(w, unn(pb['future_pred'])[~tgt_mask], other_match, other_z, other_prior)
        ld.update({'other_' + k: v for k, v in lo.items()})
        loss = ld['tgt_loss'] + ld['other_loss']
        loss.backward()
        if trace is not None:
            trace.append(_trace_entry([tgt_z, other_z], ld))
            trace[-1]['crop_flips'] = flips
        opt.step()
    return collate_tgt_other_z(ptr, tgt_z, other_z).detach()
