"""The BASELINE.json configurations beyond the headline, exercised on the MI355X (-m gpu):
  * configs[1] with its BACKWARD at 32 x 16 agents (per-scene separability of the closure's gradient),
  * configs[2]: the adversarial closure (two rollouts, complementary detach) on ~512 agents in scenes of 2..30,
  * configs[4]'s arithmetic: NC = 5 semantic classes (reduce_cats) against a fixture generated from the reference,
  * sample_batched (NS prior samples) + the feasibility gate on its output.
"""
import os
import sys

import numpy as np
import pytest
import torch

import make_golden as mg
from util import golden, oracle_model, product_model, assert_close, assert_close_frac
from strive_amd import synth
from strive_amd.graph import Batch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RT, AT = 1e-4, 2e-5


@pytest.fixture(scope='module')
def model():
    assert torch.cuda.is_available(), 'gpu tests need the MI355X'
    return product_model(device=DEV)


def dev_env(raster, dx):
    return synth.SyntheticMapEnv(raster.clone(), dx.clone()).to(DEV)


def uniform(H=2048):
    raster = torch.zeros((1, 4, H, H), dtype=torch.uint8)
    raster[:, 0] = 1
    return raster, torch.tensor([[0.25, 0.25]], dtype=torch.float64)


# ------------------------------------------------------------------------------------------------
# NC = 5
# ------------------------------------------------------------------------------------------------

def test_nc5_embed_and_rollout_golden():
    g = golden('g4b_nc5.npz')
    m5, sd5 = product_model(NC=5, device=DEV, key='weights5')
    batch, map_idx, raster, dx = mg.build_inputs(mg.G4B_SIZES, 'g4b', NC=5)
    assert batch.sem.shape[1] == 5
    for pre, (ra, dxx), tight in (('u_', mg.loop_rasters('u'), True), ('', (raster, dx), False)):
        env = dev_env(ra, dxx)
        bg = batch.clone().to(DEV)
        with torch.no_grad():
            emb = m5.embed(bg, map_idx.to(DEV), env)
        assert_close(emb['map_feat'], g[pre + 'map_feat'], RT, AT, pre + 'map_feat')
        assert_close(emb['past_feat'], g[pre + 'past_feat'], RT, AT, pre + 'past_feat')
        if not pre:
            assert_close(emb['prior_out'][0], g['prior_mu'], RT, AT, 'prior mu')
            assert_close(emb['prior_out'][1], g['prior_var'], RT, 1e-4, 'prior var')
            assert_close(emb['posterior_out'][0], g['post_mu'], RT, AT, 'post mu')
            assert_close(emb['posterior_out'][1], g['post_var'], RT, 1e-4, 'post var')
        z = synth.make_latents(torch.from_numpy(g['prior_mu']), torch.from_numpy(g['prior_var']), key='g4b/z')
        zg = z.to(DEV).requires_grad_(True)
        femb = {'map_feat': torch.from_numpy(g[pre + 'map_feat']).to(DEV), 'past_feat': torch.from_numpy(g[pre + 'past_feat']).to(DEV)}
        pred = m5.decode_embedding(zg, femb, bg, map_idx.to(DEV), env)['future_pred']
        rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'g4b/r', -1.0, 1.0)).to(DEV)
        (pred * rw).sum().backward()
        gz = g[pre + 'gz']
        if tight:
            assert_close(pred, g[pre + 'pred'], RT, AT, 'NC=5 future_pred (uniform raster)')
            assert_close(zg.grad, gz, 2e-3, 1e-6 + 2e-4 * float(np.abs(gz).max()), 'NC=5 dL/dz (uniform raster)')
        else:
            assert_close(pred, g['pred'], 0, 1e-2, 'NC=5 future_pred (textured)')
            assert_close(zg.grad, gz, 0, 5e-2 * float(np.abs(gz).max()), 'NC=5 dL/dz (textured)')


# ------------------------------------------------------------------------------------------------
# configs[1] with backward: the gradient of a per-scene separable objective at 32 x 16
# ------------------------------------------------------------------------------------------------

def _separable_objective(m, pred, veh):
    """sum(pred * R) + sum of vehicle-collision penalties: no batch-wide normalisation, so scenes do not interact."""
    from strive_amd.losses.adv_gen_nusc import interp_traj
    rw = synth.f32(synth.counter_uniform((16, pred.shape[1], 4), 'gc/rw', -1.0, 1.0)).to(pred.device)
    rw = rw.repeat(pred.shape[0] // 16, 1, 1)
    fine = interp_traj(m.get_normalizer().unnormalize(pred), 3)
    pen, mask = veh.block_penalties(fine)
    return (pred * rw).sum() + (pen * mask).sum()


def test_headline_closure_backward_is_per_scene(model):
    from strive_amd.losses.adv_gen_nusc import VehCollLoss
    m, sd = model
    raster, dx = synth.make_raster(2048, 2048)
    env = dev_env(raster, dx)
    batch, map_idx = synth.make_batch([16] * 32, key='gc/full', map_extent=(512.0, 512.0))
    bg = batch.clone().to(DEV)
    mi = map_idx.to(DEV)
    with torch.no_grad():
        emb = m.embed(bg, mi, env)
    z = synth.make_latents(emb['prior_out'][0].cpu(), emb['prior_out'][1].cpu(), key='gc/full/z').to(DEV)
    zf = z.clone().requires_grad_(True)
    pred = m.decode_embedding(zf, emb, bg, mi, env, nfuture=16)['future_pred']
    veh = VehCollLoss(m.get_att_normalizer().unnormalize(bg.lw), buffer_dist=0.2, ptr=bg.ptr)
    _separable_objective(m, pred, veh).backward()
    assert pred.shape == (512, 16, 4) and torch.isfinite(zf.grad).all() and float(zf.grad.abs().max()) > 0
    scenes = batch.to_data_list()
    for b in (5, 17):
        sb = Batch.from_data_list([scenes[b]]).to(DEV)
        lo = 16 * b
        e1 = {'map_feat': emb['map_feat'][lo:lo + 16].contiguous(), 'past_feat': emb['past_feat'][lo:lo + 16].contiguous()}
        z1 = z[lo:lo + 16].clone().requires_grad_(True)
        p1 = m.decode_embedding(z1, e1, sb, mi[b:b + 1], env, nfuture=16)['future_pred']
        v1 = VehCollLoss(m.get_att_normalizer().unnormalize(sb.lw), buffer_dist=0.2, ptr=sb.ptr)
        _separable_objective(m, p1, v1).backward()
        assert_close(p1, pred[lo:lo + 16], 0, 1e-6, 'scene %d alone: future_pred' % b)
        assert_close(z1.grad, zf.grad[lo:lo + 16], 1e-5, 1e-6 * float(zf.grad.abs().max()), 'scene %d alone: dL/dz' % b)


# ------------------------------------------------------------------------------------------------
# configs[2]: the adversarial closure at ~512 agents in scenes of 2..30
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('NC', [2, 5])
def test_adv_closure_at_size(model, NC):
    import bench
    from strive_amd.utils.adv_gen_optim import AdvClosure
    from strive_amd.utils.scenario_gen import detach_embed_info
    if NC == 2:
        m, sd = model
    else:
        m, sd = product_model(NC=5, device=DEV, key='weights5')
    sizes = bench.variable_scene_sizes(512, 'gc/adv')
    assert sum(sizes) == 512 and min(sizes) >= 2 and max(sizes) <= 30
    raster, dx = uniform()
    env = dev_env(raster, dx)
    batch, map_idx = synth.make_batch(sizes, key='gc/adv%d' % NC, NC=NC, map_extent=(512.0, 512.0))
    bg = batch.clone().to(DEV)
    mi = map_idx.to(DEV)
    with torch.no_grad():
        emb = detach_embed_info(m.embed(bg, mi, env))
    NA = 512
    ego = torch.zeros((NA,), dtype=torch.bool, device=DEV)
    ego[bg.ptr[:-1].to(DEV)] = True
    pm, pv = emb['prior_out']
    c = AdvClosure(emb['posterior_out'][0].clone(), 0.05, bench.ADV_WEIGHTS, m, bg, env, mi, emb, (pm[ego], pv[ego]),
                   (pm[~ego], pv[~ego]), 2, 0.0, veh_coll_buffer=0.1)
    seen = {}

    def log(ld, tz, oz):
        seen.update({k: v.detach().clone() for k, v in ld.items() if torch.is_tensor(v)})
        seen['g_tgt'], seen['g_other'] = tz.grad.clone(), oz.grad.clone()
    z_t0, z_o0 = c.tgt_z.detach().clone(), c.other_z.detach().clone()
    loss = c.step(log=log)
    assert torch.isfinite(loss) and all(torch.isfinite(v).all() for v in seen.values())
    for k in ('tgt_match_match_ext_loss', 'tgt_match_loss', 'adv_adv_crash_loss', 'adv_motion_prior_loss', 'adv_init_loss',
              'adv_coll_veh_loss', 'adv_coll_veh_plan_loss', 'adv_coll_env_loss', 'adv_loss'):
        assert k in seen, k
    assert seen['adv_adv_crash_loss'].shape == (len(sizes),) and seen['adv_motion_prior_loss'].shape == (NA - len(sizes),)
    assert (c.tgt_z.detach() - z_t0).abs().max() > 0 and (c.other_z.detach() - z_o0).abs().max() > 0
    # complementary detach: the ego latents' gradient is that of the matching loss alone, the others' that of the
    # adversarial loss alone (reference src/utils/adv_gen_optim.py:120-121)
    tz = z_t0.clone().requires_grad_(True)
    oz = z_o0.clone().requires_grad_(True)
    from strive_amd.utils.adv_gen_optim import collate_tgt_other_z
    unn = m.get_normalizer().unnormalize
    pa = m.decode_embedding(collate_tgt_other_z(bg, tz, oz), emb, bg, mi, env, ext_future=c.planner_fut, nfuture=12)['future_pred']
    lt = c.tgt_loss(unn(pa[ego]), unn(c.planner_fut), tz, c.tgt_prior)
    g_t, g_o_from_match = torch.autograd.grad(lt['loss'], [tz, oz])
    assert_close(seen['g_tgt'], g_t, 1e-4, 1e-7, 'ego latents: gradient of the matching loss only')
    assert float(g_o_from_match.abs().max()) > 0, 'the matching loss does depend on the others (so the detach matters)'
    pb = m.decode_embedding(collate_tgt_other_z(bg, tz, oz), emb, bg, mi, env, ext_future=c.planner_fut, nfuture=12)['future_pred']
    la = c.adv_loss(unn(pb), unn(c.planner_fut), oz, c.other_prior)
    g_t_from_adv, g_o = torch.autograd.grad(la['loss'], [tz, oz])
    assert_close(seen['g_other'], g_o, 1e-4, 1e-7 * max(1.0, float(g_o.abs().max())), 'other latents: gradient of the adversarial loss only')
    # a sample of scenes (largest, smallest, one in between) against the oracle, rollout forward + d/dz
    orc = oracle_model(sd, NC=NC)
    env_c = synth.SyntheticMapEnv(raster, dx)
    scenes = batch.to_data_list()
    order = sorted(range(len(sizes)), key=lambda i: sizes[i])
    for b in (order[-1], order[0], order[len(order) // 2]):
        lo, hi = int(batch.ptr[b]), int(batch.ptr[b + 1])
        sb = Batch.from_data_list([scenes[b]])
        e1 = {'map_feat': emb['map_feat'][lo:hi].cpu(), 'past_feat': emb['past_feat'][lo:hi].cpu()}
        zz = collate_tgt_other_z(bg, z_t0, z_o0)[lo:hi].cpu().clone().requires_grad_(True)
        ext = c.planner_fut[b:b + 1].cpu()
        pc = orc.decode_embedding(zz, e1, sb, map_idx[b:b + 1], env_c, ext_future=ext, nfuture=12)['future_pred']
        rw = synth.f32(synth.counter_uniform(tuple(pc.shape), 'gc/adv/rw%d' % b, -1.0, 1.0))
        gc_, = torch.autograd.grad((pc * rw).sum(), [zz])
        zf = collate_tgt_other_z(bg, z_t0, z_o0).clone().requires_grad_(True)
        pf = m.decode_embedding(zf, emb, bg, mi, env, ext_future=c.planner_fut, nfuture=12)['future_pred']
        (pf[lo:hi] * rw.to(DEV)).sum().backward()
        assert_close(pf[lo:hi], pc.detach(), RT, AT, 'scene %d (n=%d) future_pred vs oracle' % (b, hi - lo))
        gmax = float(gc_.abs().max())
        # 12 steps x up to 29 neighbours of max-aggregation: a few entries may sit on an arg-max near-tie; and these scenes
        # lie up to 512 m from the origin (normalised x ~ 34: fp32 position differences carry 4e-6 instead of 1e-7)
        assert_close_frac(zf.grad[lo:hi], gc_, 2e-3, 1e-6 + 2e-4 * gmax, 0.9, 1e-2 * gmax, 'scene %d d/dz vs oracle' % b)
        rel = float((zf.grad[lo:hi].cpu() - gc_).norm() / gc_.norm())
        assert rel < 1e-2, 'scene %d d/dz vs oracle: relative L2 error %.3g' % (b, rel)
        others = torch.ones((NA,), dtype=torch.bool)
        others[lo:hi] = False
        assert float(zf.grad[others.to(DEV)].abs().max()) == 0.0, 'no gradient may leak into other scenes'


# ------------------------------------------------------------------------------------------------
# sample_batched + feasibility
# ------------------------------------------------------------------------------------------------

def test_sample_batched_golden_and_feasibility(model):
    from strive_amd.utils.scenario_gen import determine_feasibility_nusc
    from oracle import losses as ol
    m, sd = model
    g = golden('g7_sample.npz')
    batch, map_idx, raster, dx = mg.build_inputs([4, 2], 'g7')
    NA = batch.past.shape[0]
    eps = synth.f32(synth.counter_normal((3, NA, 32), 'g7/eps'))
    orc = oracle_model(sd)
    saved = m.rsample
    try:
        m.rsample = lambda mean, var: mean + eps.to(mean.device) * torch.sqrt(var)      # the reference draws eps unseeded
        with torch.no_grad():
            so = m.sample_batched(batch.clone().to(DEV), map_idx.to(DEV), dev_env(raster, dx), 3, include_mean=True, nfuture=8)
            uraster, udx = mg.loop_rasters('u')
            env_u = dev_env(uraster, udx)
            su = m.sample_batched(batch.clone().to(DEV), map_idx.to(DEV), env_u, 3, include_mean=True, nfuture=8)
    finally:
        m.rsample = saved
    assert so['future_pred'].shape == (NA, 3, 8, 4) and so['z_samp'].shape == (NA, 3, 32)
    # textured raster vs the reference (7 re-sampled steps: loose on the trajectories, tight on what precedes them)
    assert_close(so['z_samp'], g['z_samp'], RT, AT, 'z_samp')
    assert_close(so['z_logprob'], g['z_logprob'], 1e-4, 1e-3, 'z_logprob')
    assert_close(so['z_mdist'], g['z_mdist'], 1e-4, 1e-4, 'z_mdist')
    assert_close(so['future_pred'], g['future_pred'], 0, 1e-2, 'sample future_pred (textured)')
    assert_close(so['z_samp'][:, -1], so['prior_out'][0], 0, 0, 'include_mean puts the prior mean last')
    # uniform raster vs the oracle: tight
    with torch.no_grad():
        wo = orc.sample_batched(batch, map_idx, synth.SyntheticMapEnv(uraster, udx), eps, include_mean=True, nfuture=8)
    assert_close(su['future_pred'], wo['future_pred'], RT, AT, 'sample future_pred (uniform)')
    assert_close(su['z_logprob'], wo['z_logprob'], 1e-4, 1e-3, 'z_logprob (uniform)')
    # the feasibility gate of adv_scenario_gen.py:160-174 on the first scene's samples (device tensors in, same verdicts as
    # the oracle's restatement on the oracle's samples)
    nrm = m.get_normalizer()
    sc0 = slice(0, 4)
    for th, t0, vel, front in ((15.0, 0, 0.0, None), (40.0, 1, 0.0, 0.0), (5.0, 0, 1.0, -0.5)):
        f, st, ds = determine_feasibility_nusc(su['future_pred'][sc0].clone(), nrm, th, feasibility_time=t0, feasibility_vel=vel,
                                               feasibility_infront_min=front, check_non_drivable_separation=True, map_env=env_u,
                                               map_idx=map_idx[0:1].to(DEV))
        fo, sto, dso = ol.determine_feasibility(wo['future_pred'][sc0].clone(), orc.get_normalizer(), th, time=t0, vel=vel,
                                                infront_min=front, check_sep=True, raster=uraster, dx=udx, map_idx=map_idx[0:1])
        assert np.array_equal(f.cpu().numpy(), fo.numpy()) and np.array_equal(st.cpu().numpy(), sto.numpy())
        np.testing.assert_allclose(ds.cpu().numpy(), dso.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.gpu
def test_sample_batched_at_the_reference_operating_point(model):
    """adv_scenario_gen.py:160-174 calls sample_batched(NS = 20, include_mean = True) on every scene before it is accepted.  At
    32 scenes x 16 agents x 20 samples (10,240 rollout rows, uniform raster so that comparisons are tight) the size-independent
    properties of the joint rollout: sample s of the batch equals a plain 2-D rollout of z[:, s] (the reference's NS code path
    against its 2-D path), the last sample is the prior mean's rollout, every scene equals itself rolled out alone, z statistics
    equal their closed forms, and a sample of rows equals the oracle."""
    m, sd = model
    sizes = [16] * 32
    batch, map_idx = synth.make_batch(sizes, key='ns20', FT=12)
    uraster, udx = mg.loop_rasters('u')
    env = dev_env(uraster, udx)
    bg, mi = batch.clone().to(DEV), map_idx.to(DEV)
    NA, NS, FT = batch.past.shape[0], 20, 12
    eps = synth.f32(synth.counter_normal((NS, NA, 32), 'ns20/eps')).to(DEV)
    saved = m.rsample
    try:
        m.rsample = lambda mean, var: mean + eps[:, :mean.shape[1]] * torch.sqrt(var)
        with torch.no_grad():
            so = m.sample_batched(bg, mi, env, NS, include_mean=True, nfuture=FT)
    finally:
        m.rsample = saved
    fp, z = so['future_pred'], so['z_samp']
    assert fp.shape == (NA, NS, FT, 4) and z.shape == (NA, NS, 32) and bool(torch.isfinite(fp).all())
    mu, var = so['prior_out']
    assert torch.equal(z[:, -1], mu)
    with torch.no_grad():
        emb = m.embed(bg, mi, env)
        for s in (0, 7, NS - 1):
            one = m.decode_embedding(z[:, s].contiguous(), emb, bg, mi, env, nfuture=FT)['future_pred']
            assert_close(fp[:, s], one, 1e-4, 2e-5, 'sample %d of the joint rollout vs its own 2-D rollout' % s)
    # closed forms of the latent statistics
    lp = (-0.5 * ((z - mu[:, None]) ** 2 / var[:, None]) - 0.5 * torch.log(2 * np.pi * var[:, None])).sum(-1)
    assert_close(so['z_logprob'], lp, 1e-4, 1e-3, 'z_logprob')
    assert_close(so['z_mdist'], torch.norm((z - mu[:, None]) / torch.sqrt(var[:, None]), dim=-1), 1e-4, 1e-4, 'z_mdist')
    # scenes 3 and 17 alone
    from strive_amd.graph import Batch
    for b in (3, 17):
        sub = Batch.from_data_list([batch.to_data_list()[b]])
        smi = map_idx[b:b + 1]
        rows = slice(16 * b, 16 * b + 16)
        try:
            m.rsample = lambda mean, var, rows=rows: mean + eps[:, rows] * torch.sqrt(var)
            with torch.no_grad():
                alone = m.sample_batched(sub.clone().to(DEV), smi.to(DEV), env, NS, include_mean=True, nfuture=FT)
        finally:
            m.rsample = saved
        assert_close(alone['future_pred'], fp[rows], 1e-4, 2e-5, 'scene %d alone vs inside the batch' % b)
    # the oracle on scene 3, samples 0..2 (CPU: 3 x 16 rollout rows)
    orc = oracle_model(sd)
    sub = Batch.from_data_list([batch.to_data_list()[3]])
    with torch.no_grad():
        wo = orc.sample_batched(sub, map_idx[3:4], synth.SyntheticMapEnv(uraster, udx), eps[:3, 48:64].cpu(), include_mean=False,
                                nfuture=FT)
    assert_close(fp[48:64, :3], wo['future_pred'], RT, AT, 'joint rollout rows vs the oracle')
