"""Shared by the loop-parity tests: drive the init / adversarial / solution loops of the oracle (CPU) or of the product
(HIP) from the states stored in tests/golden/g6_loops.npz -- which the reference's OWN loop functions produced
(make_golden.py::g6_loops) -- and compare the per-iteration traces.

Why two rasters: over the uniform raster ('u') the crop, hence the map feature, does not depend on the pose, so the whole
chain rollout -> losses -> backward -> Adam is smooth and an independent fp32 implementation must track the reference
tightly for many iterations.  Over the textured raster ('t') every rollout step re-samples the raster at poses that depend
on the previous step: a 1e-7 pose difference can flip crop pixels and move a map feature by ~4e-3, and the hard collision
thresholds amplify further, so only the first closure is comparable entry-wise and later iterations by their loss terms."""
import numpy as np
import torch

import make_golden as mg
from strive_amd import synth


def loop_inputs(kind):
    batch, map_idx, _, _ = mg.g5_inputs(None, None)
    raster, dx = mg.loop_rasters(kind)
    return batch, map_idx, raster, dx


def loop_start(g, kind, name, emb_prior, ego):
    """Starting latents of loop `name` = what the reference's previous loop returned (init starts from the injected z0)."""
    pm, pv = emb_prior
    if name == 'init':
        return synth.make_latents(pm, pv, key='g6l/z')
    if name == 'adv':
        return torch.from_numpy(g[kind + '/init/z_out'])
    return torch.from_numpy(g[kind + '/adv/z_out'])


def trace_logger(trace):
    def log(ld, *zs):
        ent = {'z': [z.detach().cpu().clone() for z in zs], 'grad': [z.grad.detach().cpu().clone() for z in zs]}
        for k, v in ld.items():
            if torch.is_tensor(v):
                ent[k] = v.detach().cpu()
        trace.append(ent)
    return log


def frac_within(a, b, atol, rtol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.mean(np.abs(a - b) <= atol + rtol * np.abs(b)))


def compare_trace(trace, g, tag, loss_rtol, loss_atol, grad_rtol, z_atol, z_frac=0.99, n_iters=None, report=None, grad_skip=(), grad_row_frac=None, kink_after=None):
    """`trace[it]` = {'z': [..], 'grad': [..], loss entries} of iteration `it` BEFORE its Adam step, like the fixture.

    * every loss-dict entry (mean) within loss_rtol / loss_atol at every iteration;
    * gradients: relative L2 error of each leaf's gradient <= grad_rtol;
    * latents: at least `z_frac` of the entries within z_atol.  (Adam's first steps are lr*g/(|g|+eps): an entry whose
      gradient is at the noise level can take a +-lr step in either direction, so a small fraction of entries may differ by
      up to 2*lr while everything else agrees; those entries are near-irrelevant to the loss by construction.)
    Returns the worst observed figures (for the calibration log)."""
    keys = [str(k) for k in g[tag + '/loss_keys']]
    want_losses = g[tag + '/losses']
    n = len(trace) if n_iters is None else n_iters
    assert len(trace) >= n and want_losses.shape[0] >= n
    worst = {'loss_rel': 0.0, 'grad_rel': 0.0, 'z_frac': 1.0, 'z_max': 0.0}
    nz = len(trace[0]['z'])
    for it in range(n):
        assert set(keys) <= set(trace[it].keys()), 'iteration %d: missing loss entries %s' % (it, sorted(set(keys) - set(trace[it])))
        got = np.array([float(torch.mean(trace[it][k])) for k in keys])
        want = want_losses[it]
        rel = np.abs(got - want) / (loss_atol / max(loss_rtol, 1e-30) + np.abs(want))
        worst['loss_rel'] = max(worst['loss_rel'], float(rel.max()))
        relax = 10.0 if ('first_kink' in worst and it > worst['first_kink']) else 1.0     # after a kink event: see below
        bad = np.abs(got - want) > relax * (loss_atol + loss_rtol * np.abs(want))
        assert not bad.any(), '%s iteration %d: loss entries off: %s' % (
            tag, it, ', '.join('%s got %.6g want %.6g' % (keys[i], got[i], want[i]) for i in np.nonzero(bad)[0]))
        for i in range(nz):
            gg = trace[it]['grad'][i].double().numpy().reshape(-1)
            gw = g['%s/grad%d' % (tag, i)][it].astype(np.float64).reshape(-1)
            gr = float(np.linalg.norm(gg - gw) / max(np.linalg.norm(gw), 1e-30))
            if grad_row_frac is not None:
                # kinks (arg-min over circle pairs, arg-max messages, clamps) make the gradient discontinuous: a 1e-7
                # difference upstream can change ONE agent's gradient row by O(1) while all others agree.  Row-wise check:
                # at least `grad_row_frac` of the agents' rows within grad_rtol, and the whole gradient's direction.
                G = trace[it]['grad'][i].double().numpy()
                W = g['%s/grad%d' % (tag, i)][it].astype(np.float64)
                G, W = G.reshape(G.shape[0], -1), W.reshape(W.shape[0], -1)
                rows = np.linalg.norm(G - W, axis=1) / np.maximum(np.linalg.norm(W, axis=1), 1e-30)
                cos = float((gg * gw).sum() / max(np.linalg.norm(gg) * np.linalg.norm(gw), 1e-30))
                assert cos >= 0.99, '%s iteration %d: gradient direction cos = %.4f' % (tag, it, cos)
                ok = float(np.mean(rows <= grad_rtol))
                if it in grad_skip:
                    ok = 1.0          # (a different NUMBER of active pairs rescales every row: direction only, see the caller)
                if ok < 1.0 and kink_after is not None and it >= kink_after:
                    worst['first_kink'] = min(worst.get('first_kink', it), it)
                if 'first_kink' not in worst:
                    # before the first kink event (never before iteration `kink_after`): the rows agree tightly.  AT the event the
                    # free-running comparison keeps the direction (cos, above), the losses and the latents only: with ONE scene in
                    # the batch a circle pair changing sides of its hinge moves every agent's gradient through the interaction net,
                    # and whether the two runs are on the same side is decided by the last bit of either run -- the product's or the
                    # CPU oracle's, whose convolutions depend on the host (round 4: the verdict of the row check at such an iteration
                    # flipped between runs of identical builds on different boxes / test selections).  The gradient AT the oracle's latents of EVERY iteration is pinned separately, without
                    # accumulation: test_refine_closure_at_the_oracle_latents_every_iteration.
                    assert ok >= grad_row_frac, '%s iteration %d: only %.3f of the gradient rows within %.1e (rows: %s; cos %.5f)' % (
                        tag, it, ok, grad_rtol, ' '.join('%.3g' % r for r in rows), cos)
                gr = float(np.median(rows)) if 'first_kink' not in worst else 0.0
            if it in grad_skip:
                gr = 0.0
            worst['grad_rel'] = max(worst['grad_rel'], gr)
            assert gr <= grad_rtol, '%s iteration %d: gradient of leaf %d off by %.3g (relative L2)' % (tag, it, i, gr)
            zg = trace[it]['z'][i].double().numpy().reshape(-1)
            zw = g['%s/z%d' % (tag, i)][it].astype(np.float64).reshape(-1)
            post = 'first_kink' in worst and it > worst['first_kink']
            fr = frac_within(zg, zw, z_atol * (5.0 if post else 1.0))
            worst['z_frac'] = min(worst['z_frac'], fr)
            worst['z_max'] = max(worst['z_max'], float(np.abs(zg - zw).max()))
            assert fr >= (min(z_frac, 0.9) if post else z_frac), '%s iteration %d: only %.4f of the latent entries of leaf %d within %.1e' % (tag, it, fr, i, z_atol)
    if report is not None:
        report.append((tag, n, worst))
    return worst


def run_oracle_loop(name, kind, g, orc, n):
    from oracle import loops
    batch, map_idx, raster, dx = loop_inputs(kind)
    env = synth.SyntheticMapEnv(raster, dx)
    with torch.no_grad():
        emb = orc.embed(batch, map_idx, env)
    NA = batch.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    pm, pv = emb['prior_out']
    tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
    z = loop_start(g, kind, name, emb['prior_out'], ego)
    trace = []
    if name == 'init':
        loops.init_loop(orc, batch, map_idx, env, emb, z, batch.future_gt[:, :, :4], batch.future_vis, mg.LOOP_WEIGHTS, n, mg.LOOP_LR,
                        emb['prior_out'], trace=trace)
    elif name == 'adv':
        loops.adv_loop(orc, batch, map_idx, env, emb, z, mg.LOOP_WEIGHTS, n, mg.LOOP_LR, tp, op, feasibility_time=2,
                       feasibility_infront_min=0.0, trace=trace)
    else:
        fin = torch.from_numpy(g[kind + '/adv/final_result_traj'])
        loops.sol_loop(orc, batch, map_idx, env, emb, z, fin, 16, mg.LOOP_WEIGHTS, n, mg.LOOP_LR, tp, op, trace=trace)
    return trace


def run_product_loop(name, kind, g, m, n, device, embed_from=None):
    """The product's loop functions (strive_amd.utils.*_optim) on `device`; the embed outputs come from the product's own
    embed() unless `embed_from` (an oracle) is given."""
    from strive_amd.utils.init_optim import run_init_optim
    from strive_amd.utils.adv_gen_optim import run_adv_gen_optim
    from strive_amd.utils.sol_optim import run_find_solution_optim
    from strive_amd.utils.scenario_gen import detach_embed_info
    batch, map_idx, raster, dx = loop_inputs(kind)
    env = synth.SyntheticMapEnv(raster.clone(), dx.clone()).to(device)
    bg = batch.clone().to(device)
    mi = map_idx.to(device)
    with torch.no_grad():
        emb = detach_embed_info(m.embed(bg, mi, env))
    NA = bg.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool, device=device)
    ego[bg.ptr[:-1].to(device)] = True
    pm, pv = emb['prior_out']
    tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
    z = loop_start(g, kind, name, (pm.cpu(), pv.cpu()), ego.cpu()).to(device)
    trace = []
    log = trace_logger(trace)
    if name == 'init':
        res = run_init_optim(z, bg.future_gt[:, :, :4], bg.future_vis, mg.LOOP_LR, mg.LOOP_WEIGHTS, m, bg, env, mi, n, emb,
                             emb['prior_out'], log=log)
    elif name == 'adv':
        res = run_adv_gen_optim(z, mg.LOOP_LR, mg.LOOP_WEIGHTS, m, bg, env, mi, n, emb, 'ego', tp, op, 2, 0.0, log=log)
    else:
        fin = torch.from_numpy(g[kind + '/adv/final_result_traj']).to(device)
        res = run_find_solution_optim(z, fin, 16, mg.LOOP_LR, mg.LOOP_WEIGHTS, m, bg, env, mi, n, emb, tp, op, log=log)
    return trace, res


def hardcode_inputs():
    lg, batch, map_idx, raster, dx = mg.g6h_inputs()
    return lg, batch, map_idx, raster, dx


def run_oracle_hardcode_loop(g, orc, n):
    """oracle rollouts / losses + the oracle's restatement of the rule-based planner in closed loop"""
    from oracle import loops
    from oracle.planner import HardcodeNuscPlanner, PlannerConfig, CONFIG_DICT
    lg, batch, map_idx, raster, dx = hardcode_inputs()
    env = synth.SyntheticMapEnv(raster, dx, lane_graph=lg)
    with torch.no_grad():
        emb = orc.embed(batch, map_idx, env)
    NA = batch.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    pm, pv = emb['prior_out']
    z0 = synth.make_latents(pm, pv, key='g6h/z')
    planner = HardcodeNuscPlanner(env, PlannerConfig(**CONFIG_DICT['default']))
    trace = []
    orc.dt = 0.5
    loops.adv_loop(orc, batch, map_idx, env, emb, z0, mg.LOOP_WEIGHTS, n, mg.LOOP_LR, (pm[ego], pv[ego]), (pm[~ego], pv[~ego]),
                   feasibility_time=2, feasibility_infront_min=0.0, trace=trace, planner=planner)
    return trace


def run_product_hardcode_loop(g, m, n, device):
    from strive_amd.utils.adv_gen_optim import run_adv_gen_optim
    from strive_amd.utils.scenario_gen import detach_embed_info
    from strive_amd.planners.planner import PlannerConfig
    from strive_amd.planners.hardcode_goalcond_nusc import HardcodeNuscPlanner, CONFIG_DICT
    lg, batch, map_idx, raster, dx = hardcode_inputs()
    env = synth.SyntheticMapEnv(raster.clone(), dx.clone(), lane_graph=lg).to(device)
    bg = batch.clone().to(device)
    mi = map_idx.to(device)
    with torch.no_grad():
        emb = detach_embed_info(m.embed(bg, mi, env))
    NA = bg.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool, device=device)
    ego[bg.ptr[:-1].to(device)] = True
    pm, pv = emb['prior_out']
    z0 = synth.make_latents(pm.cpu(), pv.cpu(), key='g6h/z').to(device)
    planner = HardcodeNuscPlanner(env, PlannerConfig(**CONFIG_DICT['default']))
    trace = []
    res = run_adv_gen_optim(z0, mg.LOOP_LR, mg.LOOP_WEIGHTS, m, bg, env, mi, n, emb, 'hardcode', (pm[ego], pv[ego]), (pm[~ego], pv[~ego]),
                            2, 0.0, planner=planner, log=trace_logger(trace))
    return trace, res


def check_trace_at_product_latents(name, kind, g, m, orc, trace, device, loss_rtol=2e-3, loss_atol=1e-4, grad_rel=2e-2, report=None):
    """Iteration by iteration, WITHOUT accumulation: the oracle's closure of loop ``name`` evaluated at the latents the PRODUCT's
    loop visited (``trace`` of run_product_loop), cropping the raster at the product's rollout poses (oracle/loops.py test hooks:
    num_iters = 1, init_z, crop_poses) -- i.e. the same smooth function on both sides, also over the textured raster, also after
    the two free-running runs have passed a kink or a crop flip at slightly different latents.  Every loss entry (mean) within
    loss_rtol / loss_atol and every leaf's gradient within ``grad_rel`` (relative L2) at EVERY iteration.  Returns the worst figures."""
    from oracle import loops
    from strive_amd.utils.scenario_gen import detach_embed_info
    batch, map_idx, raster, dx = loop_inputs(kind)
    env_c = synth.SyntheticMapEnv(raster, dx)
    env_g = synth.SyntheticMapEnv(raster.clone(), dx.clone()).to(device)
    bg, mi = batch.clone().to(device), map_idx.to(device)
    with torch.no_grad():
        emb_g = detach_embed_info(m.embed(bg, mi, env_g))
    emb_c = {k: (tuple(t.cpu() for t in v) if isinstance(v, tuple) else v.cpu()) for k, v in emb_g.items()}      # the product's embed on both sides
    NA = batch.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    pm, pv = emb_c['prior_out']
    tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
    worst = {'loss_rel': 0.0, 'grad_rel': 0.0, 'crop_flips': 0}
    z_first = [z.clone() for z in trace[0]['z']]
    fin = torch.from_numpy(g[kind + '/adv/final_result_traj']) if name == 'sol' else None
    ext = batch.future_gt[ego][:, :, :4].contiguous() if name == 'adv' else None
    for it, ent in enumerate(trace):
        zs = [z.clone() for z in ent['z']]
        if name == 'init':
            zc = zs[0]
        else:
            zc = loops.collate_tgt_other_z(batch.ptr, zs[0].reshape(zs[0].shape[0], -1), zs[1].reshape(zs[1].shape[0], -1))
        with torch.no_grad():       # the product's rollout at these latents (bit-reproducible): the poses its crops were taken at
            kw = {'nfuture': 16} if name == 'sol' else ({'ext_future': ext.to(device)} if name == 'adv' else {})
            poses = m.decode_embedding(zc.to(device), emb_g, bg, mi, env_g, **kw)['future_pred'].cpu()
        t = []
        if name == 'init':
            loops.init_loop(orc, batch, map_idx, env_c, emb_c, zc, batch.future_gt[:, :, :4], batch.future_vis, mg.LOOP_WEIGHTS, 1,
                            mg.LOOP_LR, emb_c['prior_out'], trace=t, crop_poses=poses)
        elif name == 'adv':
            z0c = loops.collate_tgt_other_z(batch.ptr, z_first[0], z_first[1])
            loops.adv_loop(orc, batch, map_idx, env_c, emb_c, zc, mg.LOOP_WEIGHTS, 1, mg.LOOP_LR, tp, op, feasibility_time=2,
                           feasibility_infront_min=0.0, trace=t, init_z=z0c[~ego], crop_poses=poses)
        else:
            loops.sol_loop(orc, batch, map_idx, env_c, emb_c, zc, fin, 16, mg.LOOP_WEIGHTS, 1, mg.LOOP_LR, tp, op, trace=t,
                           init_z=z_first[0].reshape(z_first[0].shape[0], -1),
                           start_z=(zs[0].reshape(zs[0].shape[0], -1), zs[1].reshape(zs[1].shape[0], -1)), crop_poses=poses)
        want = t[0]
        worst['crop_flips'] += int(want['crop_flips'].sum())
        for k, v in want.items():
            if k in ('z', 'grad', 'crop_flips') or k not in ent or not torch.is_tensor(v):
                continue
            a, b = float(torch.mean(ent[k].float())), float(torch.mean(v.float()))
            worst['loss_rel'] = max(worst['loss_rel'], abs(a - b) / (loss_atol / loss_rtol + abs(b)))
            assert abs(a - b) <= loss_atol + loss_rtol * abs(b), '%s/%s iteration %d at the product latents: %s %.6g vs %.6g' % (kind, name, it, k, a, b)
        for i, (gg, gw) in enumerate(zip(ent['grad'], want['grad'])):
            gg, gw = gg.double().reshape(-1), gw.double().reshape(-1)
            rel = float((gg - gw).norm() / max(float(gw.norm()), 1e-30))
            worst['grad_rel'] = max(worst['grad_rel'], rel)
            assert rel <= grad_rel, '%s/%s iteration %d at the product latents: gradient of leaf %d off by %.3g (relative L2)' % (kind, name, it, i, rel)
    if report is not None:
        report.append(('%s/%s@product' % (kind, name), len(trace), worst))
    return worst
