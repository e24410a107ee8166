"""The training path (BASELINE.json configs[3]): TrafficModel.forward as an autograd graph of HIP
Functions with PARAMETER gradients, TrafficModelLoss on it, and the data-parallel gradient exchange.

  * CPU (-m "not gpu"): the product's forward -> loss -> backward on a tiny scene through the host-emulation build of the
    same .hip sources, all 174 parameter gradients against torch autograd of the oracle; the gloo world-size-2
    data-parallel step against the single-process step.
  * GPU (-m gpu): the G5 scene (reference fixture: loss terms + 8 weight-gradient tensors produced by the reference's own
    TrafficModel / TrafficModelLoss / backward) and all 174 gradients against the oracle.
"""
import os
import sys

import numpy as np
import pytest
import torch

import make_golden as mg
from util import golden, oracle_model, product_model, assert_close
from strive_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))

TW = {'recon': 1.0, 'kl': 0.004, 'coll_veh_prior': 0.05, 'coll_env_prior': 0.1}     # train_traffic.cfg:17-21


def _oracle_step(sd, batch, map_idx, env, eps_post, eps_prior, FT=12, crop_poses=None):
    from oracle import losses as ol
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    orc = oracle_model(sdg, FT=FT)
    out = orc.forward(batch, map_idx, env, eps_post=eps_post, eps_prior=eps_prior, crop_poses=crop_poses)
    ld = ol.traffic_model_loss(TW, batch, out, orc.get_normalizer(), orc.get_att_normalizer(), map_idx, env)
    ld['loss'].sum().backward()
    return out, ld, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sdg.items()}


def _product_step(m, batch, map_idx, env, eps_post, eps_prior):
    from strive_amd.losses.traffic_model import TrafficModelLoss
    m.train()
    for p in m.parameters():
        p.grad = None
    seq = [eps_post, eps_prior]
    saved = m.rsample
    m.rsample = lambda mean, var: mean + seq.pop(0).to(mean.device) * torch.sqrt(var)      # the reference draws eps unseeded
    try:
        out = m(batch, map_idx, env, future_sample=True)
    finally:
        m.rsample = saved
    lf = TrafficModelLoss(TW, m.get_normalizer(), m.get_att_normalizer())
    ld = lf(batch, out, map_idx=map_idx, map_env=env)
    ld['loss'][0].backward()
    err = lf.compute_err(batch, out, m.get_normalizer())
    return out, ld, {n: p.grad for n, p in m.named_parameters()}, err


def _compare_grads(got, want, rtol, frac_atol, what, rel_l2=None):
    assert set(got) == set(want) and len(got) == 174
    worst = ('', 0.0)
    for n in want:
        assert got[n] is not None, 'no gradient for %s' % n
        w = want[n]
        scale = float(w.abs().max())
        g = got[n].detach().cpu()
        rel = float((g - w).norm() / max(float(w.norm()), 1e-30))
        if rel > worst[1]:
            worst = (n, rel)
        assert_close(g, w, rtol, 1e-7 + frac_atol * scale, '%s: grad %s' % (what, n))
        if rel_l2 is not None and float(w.norm()) > 1e-12:
            assert rel <= rel_l2, '%s: grad %s: relative L2 error %.3g > %.1e' % (what, n, rel, rel_l2)
    return worst


@pytest.fixture(scope='module')
def emu_ops():
    import build as emu_build
    from strive_amd import _lib as L, ops
    emu = L.StriveLib(emu_build.build(), require_all=True)
    orig = (ops._lib_for, L.get_lib)
    ops._lib_for = lambda *tensors: emu           # CPU tensors + the emulated library: test infrastructure only
    L.get_lib = lambda: emu
    from util import poison_new_workspaces
    orig_ws = poison_new_workspaces(ops)          # new scratch buffers start as NaN bytes, not as whatever torch.empty holds
    yield emu
    ops._workspace = orig_ws
    ops._lib_for, L.get_lib = orig


def test_training_step_all_gradients_emulated(emu_ops):
    """forward(future_sample=True) -> TrafficModelLoss -> backward on 3 agents with FT = 2, every one of the 174 parameter
    gradients against autograd of the oracle."""
    m, sd = product_model(FT=2)
    batch, map_idx = synth.make_batch([2, 1], key='train/emu', FT=2)
    raster, dx = synth.make_raster(1024, 1024)
    env = synth.SyntheticMapEnv(raster, dx)
    NA = batch.past.shape[0]
    eps_post = synth.f32(synth.counter_normal((NA, 32), 'train/eps_post'))
    eps_prior = synth.f32(synth.counter_normal((NA, 32), 'train/eps_prior'))
    out_o, ld_o, g_o = _oracle_step(sd, batch, map_idx, env, eps_post, eps_prior, FT=2)
    from util import poisoned_empty
    with poisoned_empty():          # every buffer the product allocates uninitialised (outputs, tapes, scratch) starts as NaN
        out, ld, g, err = _product_step(m, batch.clone(), map_idx, env, eps_post, eps_prior)
    assert_close(out['future_pred'], out_o['future_pred'], 1e-4, 2e-5, 'future_pred')
    assert_close(out['future_samp'], out_o['future_samp'], 1e-4, 2e-5, 'future_samp')
    for k in ('loss', 'recon_loss', 'kl_loss', 'coll_veh_prior', 'coll_env_prior'):
        assert_close(ld[k], ld_o[k], 2e-3, 2e-3 if 'env' in k else 1e-5, k)
    worst = _compare_grads(g, g_o, 5e-3, 2e-3, 'emulated training step')
    print('worst relative L2 gradient error: %s %.3g' % worst)
    assert set(err) == {'pos_err', 'ang_err', 'z_logprob', 'z_mdist'} and err['pos_err'].shape == (NA * 2,)
    # DataParallelTrainer hands the HIP backward calls its flat gradient bucket (ops.GradSink: no scratch, no per-parameter adds):
    # the bucket after its forward -> loss -> backward equals the gradients autograd collected above, and the direct path was
    # actually taken for every block of HIP-side parameters (SGD with lr 0: the step itself changes nothing)
    from strive_amd import ops
    from strive_amd.distributed import DataParallelTrainer
    from strive_amd.losses.traffic_model import TrafficModelLoss
    want = torch.cat([g[n].reshape(-1) for n, _ in m.named_parameters()]).clone()
    hits, seg0 = [], ops.GradSink.segment

    def counting(self, ps):
        r = seg0(self, ps)
        hits.append((len(ps), r is not None))
        return r
    seq = [eps_post, eps_prior]
    saved = m.rsample
    m.rsample = lambda mean, var: mean + seq.pop(0).to(mean.device) * torch.sqrt(var)
    ops.GradSink.segment = counting
    # ... and this second step decodes the two samples one after the other (the round-4 form) where the first one ran them as one
    # rollout over the batch stacked twice (ops.decoder_rollout_stacked, the default): all 174 gradients of the two forms agree
    assert m.stack_rollouts
    m.stack_rollouts = False
    from strive_amd import params
    epoch = params.param_epoch()
    try:
        tr = DataParallelTrainer(m, TrafficModelLoss(TW, m.get_normalizer(), m.get_att_normalizer()), torch.optim.SGD(m.parameters(), lr=0.0))
        res = tr.step(batch.clone(), map_idx, env)
    finally:
        ops.GradSink.segment = seg0
        m.rsample = saved
        m.stack_rollouts = True
    assert res is not None, tr.last_error
    assert params.param_epoch() == epoch + 1        # the step announced its parameter update (see test_packs_follow_a_fused_optimiser)
    rel = float((tr.bucket[:-1] - want).norm() / want.norm())
    assert rel < 1e-5, rel
    assert len(hits) >= 6 and all(h[1] for h in hits), hits
    # the optimisation entry points still give d/dz only and never touch parameter gradients
    for p in m.parameters():
        p.grad = None
    with torch.no_grad():
        emb = m.embed(batch.clone(), map_idx, env)
    z = emb['prior_out'][0].clone().requires_grad_(True)
    m.decode_embedding(z, emb, batch.clone(), map_idx, env)['future_pred'].sum().backward()
    assert z.grad is not None and all(p.grad is None for p in m.parameters())
    m.eval()
    # the same forward without an autograd graph (validation: reference src/train_traffic.py:186-199 under no_grad) takes the stacked
    # rollout too, through the optimisation path's Function: the same trajectories, bit for bit
    seq = [eps_post, eps_prior]
    m.rsample = lambda mean, var: mean + seq.pop(0).to(mean.device) * torch.sqrt(var)
    try:
        with torch.no_grad():
            out_v = m(batch.clone(), map_idx, env, future_sample=True)
    finally:
        m.rsample = saved
    assert torch.equal(out_v['future_pred'], out['future_pred']) and torch.equal(out_v['future_samp'], out['future_samp'])


def test_packs_follow_a_fused_optimiser():
    """torch's fused optimisers write the parameters without bumping their version counters, which the weight-pack and max |w|
    caches key on: params.parameters_changed() (called by DataParallelTrainer.step after every optimiser step) starts a new epoch
    and the next use rebuilds."""
    from strive_amd import ops, params
    lin = torch.nn.Linear(8, 4)
    built = []

    def build():
        built.append(lin.weight.detach().clone())
        return len(built)
    assert ops._cached_pack(lin, 'k', lin, build) == 1 and ops._cached_pack(lin, 'k', lin, build) == 1
    a0 = params.absmax(lin.weight)
    for p in lin.parameters():
        p.grad = torch.ones_like(p)
    try:
        opt = torch.optim.Adam(lin.parameters(), lr=0.5, fused=True)
    except (RuntimeError, TypeError, ValueError):
        pytest.skip('no fused Adam for CPU tensors in this torch build')
    opt.step()
    assert not torch.equal(built[0], lin.weight.detach())
    params.parameters_changed()
    assert ops._cached_pack(lin, 'k', lin, build) == 2 and torch.equal(built[1], lin.weight.detach())
    assert ops._cached_pack_hit(lin, 'k', lin)
    assert params.absmax(lin.weight) == float(lin.weight.detach().abs().max()) != a0


@pytest.mark.gpu
def test_training_step_golden_and_all_gradients():
    """The G5 scene on the MI355X: loss terms and 8 weight-gradient tensors of the REFERENCE's own training step (fixture),
    and all 174 gradients against the oracle."""
    DEV = 'cuda:0'
    g = golden('g5_losses.npz')
    m, sd = product_model(device=DEV)
    batch, map_idx, raster, dx = mg.g5_inputs(None, None)
    NA = batch.past.shape[0]
    eps_post = synth.f32(synth.counter_normal((NA, 32), 'g5/eps_post'))
    eps_prior = synth.f32(synth.counter_normal((NA, 32), 'g5/eps_prior'))
    env_g = synth.SyntheticMapEnv(raster.clone(), dx.clone()).to(DEV)
    out, ld, grads, err = _product_step(m, batch.clone().to(DEV), map_idx.to(DEV), env_g, eps_post, eps_prior)
    # textured raster, 11 re-sampled steps, against the REFERENCE's own step: the crops of both runs are compared exactly
    # (tests/util.py); trajectories tight up to a scene's first observed crop difference, the loose bound only downstream of one;
    # the loss terms at the tight tolerance if no crop differed anywhere, else at the loose one
    from util import crop_flips, clean_mask, assert_close_flip_gated
    from strive_amd.constants import state_norm_tensors
    mean, std = state_norm_tensors()
    env_tc = synth.SyntheticMapEnv(raster, dx)
    nflip = 0
    for key in ('future_pred', 'future_samp'):
        want = torch.from_numpy(g['train_' + key])
        flips = crop_flips(env_g, env_tc, out[key].detach().cpu(), want, map_idx[batch.batch], mean[:4], std[:4])
        nflip += int(flips.sum())
        n_clean, n_all = assert_close_flip_gated(out[key], want, clean_mask(flips, batch.batch), 1e-4, 2e-5, 1e-2, 'train ' + key, min_clean=NA)
        print('training step, textured, %s vs the reference: %d crop differences, %d of %d cells tight' % (key, int(flips.sum()), n_clean, n_all))
    for k in ('loss', 'recon_loss', 'kl_loss'):
        assert_close(ld[k], g['train_' + k], 2e-2 if nflip else 2e-3, 1e-3 if nflip else 1e-5, 'train ' + k)
    assert int(g['train_ngrads']) == 174 and all(v is not None for v in grads.values())
    for n in ('decoder_net.mlp_out.net.6.weight', 'decoder_memory.weight_hh_l0', 'map_conv.0.weight', 'map_feature.weight',
              'prior_net.msg.0.edge_mlp.net.0.weight', 'past_encoder.net.0.weight', 'posterior_net.mlp_in.net.0.weight',
              'future_encoder.net.9.bias'):
        w = g['train_grad/' + n]
        got = grads[n].detach().cpu().reshape(-1)[:w.size].numpy().reshape(w.shape) if w.size < grads[n].numel() else \
            grads[n].detach().cpu().numpy()
        rel = float(np.linalg.norm(got - w) / max(np.linalg.norm(w), 1e-30))
        assert rel < (0.1 if nflip else 1e-3), 'reference gradient %s: relative L2 error %.3g (%d crop differences observed)' % (n, rel, nflip)
    # tight over the TEXTURED raster: the oracle's step cropping at the product's rollout poses (the same smooth function on both
    # sides: the reference crops at pos.detach(), traffic_model.py:694-695) -- trajectories, loss terms and all 174 gradients at the
    # uniform-raster tolerances, the map CNN's weight gradients over re-sampled crops included
    out_t, ld_t, g_t = _oracle_step(sd, batch, map_idx, env_tc, eps_post, eps_prior,
                                    crop_poses=(out['future_pred'].detach().cpu(), out['future_samp'].detach().cpu()))
    assert_close(out['future_pred'], out_t['future_pred'], 1e-4, 2e-5, 'future_pred (textured, same crops)')
    assert_close(out['future_samp'], out_t['future_samp'], 1e-4, 2e-5, 'future_samp (textured, same crops)')
    for k in ('loss', 'recon_loss', 'kl_loss', 'coll_veh_prior', 'coll_env_prior'):
        assert_close(ld[k], ld_t[k], 2e-3, 2e-3 if 'env' in k else 1e-5, k + ' (textured, same crops)')
    # (measured on the MI355X: worst tensor 9.6e-5 -- map_conv.0.bias, whose gradient sums conv1's output adjoint over 176 re-sampled
    # crops -- against 1.6e-5 over the uniform raster: bound 2e-4)
    worst_t = _compare_grads(grads, g_t, 2e-3, 5e-4, 'training step (textured raster, same crops)', rel_l2=2e-4)
    print('textured raster, same crops: worst relative L2 gradient error: %s %.3g (the oracle\'s own poses would have changed %d crops)' % (
        worst_t + (int(out_t['crop_flips_pred'].sum()) + int(out_t['crop_flips_samp'].sum()),)))
    # tight: uniform raster (smooth chain) against the oracle, all 174 gradients
    ur, udx = mg.loop_rasters('u')
    env_c = synth.SyntheticMapEnv(ur, udx)
    out_o, ld_o, g_o = _oracle_step(sd, batch, map_idx, env_c, eps_post, eps_prior)
    env_u = synth.SyntheticMapEnv(ur.clone(), udx.clone()).to(DEV)
    out, ld, grads, err = _product_step(m, batch.clone().to(DEV), map_idx.to(DEV), env_u, eps_post, eps_prior)
    assert_close(out['future_pred'], out_o['future_pred'], 1e-4, 2e-5, 'future_pred (uniform)')
    for k in ('loss', 'recon_loss', 'kl_loss', 'coll_veh_prior', 'coll_env_prior'):
        assert_close(ld[k], ld_o[k], 2e-3, 2e-3 if 'env' in k else 1e-5, k + ' (uniform)')
    # measured on the MI355X: worst tensor 1.4e-5 relative L2 (map_conv.1.weight; the CNN backward multiplies two-piece bf16
    # operands, 2^-16 per product) -- the bound is 7 x that, so a precision regression of the backward fails here
    worst = _compare_grads(grads, g_o, 2e-3, 5e-4, 'training step (uniform raster)', rel_l2=1e-4)
    print('worst relative L2 gradient error: %s %.3g' % worst)
    m.eval()


# ------------------------------------------------------------------------------------------------
# data parallel: two gloo ranks, scenes sharded, against the single-process step on the whole batch
# ------------------------------------------------------------------------------------------------

def _dp_worker(rank, world, port, sizes, out_path):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))
    import build as emu_build
    from strive_amd import _lib as L, ops
    from strive_amd.distributed import DataParallelTrainer, shard_scenes
    from strive_amd.losses.traffic_model import TrafficModelLoss
    from strive_amd.graph import Batch
    torch.set_num_threads(2)
    emu = L.StriveLib(emu_build.build(), require_all=True)
    ops._lib_for = lambda *tensors: emu
    L.get_lib = lambda: emu
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    m, sd = product_model(FT=2)
    m.train()
    scenes = [synth.make_scene(n, 'train/dp/%d' % b, FT=2) for b, n in enumerate(sizes)]
    # one scene loses a future frame so that the visible-frame counts differ between ranks
    scenes[0].future_vis[0, 1] = 0.0
    mine = shard_scenes(sizes, world)[rank]
    batch = Batch.from_data_list([scenes[i] for i in mine])
    map_idx = torch.zeros((len(mine),), dtype=torch.long)
    raster, dx = synth.make_raster(1024, 1024)
    env = synth.SyntheticMapEnv(raster, dx)
    NA_all = sum(sizes)
    eps = synth.f32(synth.counter_normal((2, NA_all, 32), 'train/dp/eps'))
    offs = np.cumsum([0] + list(sizes))
    rows = torch.cat([torch.arange(offs[i], offs[i + 1]) for i in mine])
    seq = []
    m.rsample = lambda mean, var: mean + seq.pop(0) * torch.sqrt(var)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    tr = DataParallelTrainer(m, TrafficModelLoss(TW, m.get_normalizer(), m.get_att_normalizer()), opt)
    seq[:] = [eps[0][rows], eps[1][rows]]
    out = tr.step(batch, map_idx, env)
    res = {'loss': float(out['global_loss']), 'params': {n: p.detach().clone() for n, p in m.named_parameters()}}
    # second step: rank 1 raises inside its forward -> nobody may step (skip vote)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    if world > 1:
        seq[:] = [eps[0][rows], eps[1][rows]]
        if rank == 1:
            def boom(mean, var):
                raise RuntimeError('injected failure')
            m.rsample = boom
        # the vote travels host to host (a gloo collective of a CPU tensor), never through a read of a device tensor: from here on
        # any bool() / .item() of the bucket would be the per-step synchronisation rounds 3-4 had at N > 1
        assert tr.vote_transport.startswith('host collective'), tr.vote_transport
        real_bool = torch.Tensor.__bool__

        def guarded(t):
            assert t.data_ptr() != tr.bucket[-1:].data_ptr(), 'the skip vote was read back from the gradient bucket'
            return real_bool(t)
        torch.Tensor.__bool__ = guarded
        try:
            out2 = tr.step(batch, map_idx, env)
        finally:
            torch.Tensor.__bool__ = real_bool
        res['skipped'] = out2 is None
        res['unchanged'] = all(torch.equal(before[n], p.detach()) for n, p in m.named_parameters())
    torch.save(res, out_path % rank)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_data_parallel_step_equals_single_process(tmp_path):
    """Scenes [2, 1] on two gloo ranks (one scene each) vs both scenes in one process: same global loss, same parameters
    after the step (gradients of terms with different denominators are combined exactly), and the skip vote."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    sizes = [2, 1]
    single = str(tmp_path / 'single_%d.pt')
    _dp_worker(0, 1, port, sizes, single)
    from strive_amd import _lib as L, ops          # the in-process run patched these: restore for the other tests
    import importlib
    importlib.reload(L)
    importlib.reload(ops)
    multi = str(tmp_path / 'dp_%d.pt')
    mp.spawn(_dp_worker, args=(2, port, sizes, multi), nprocs=2, join=True)
    ref = torch.load(single % 0)
    r0, r1 = torch.load(multi % 0), torch.load(multi % 1)
    assert abs(r0['loss'] - ref['loss']) < 1e-5 * max(1.0, abs(ref['loss'])), (r0['loss'], ref['loss'])
    for n, p in ref['params'].items():
        assert torch.equal(r0['params'][n], r1['params'][n]), 'ranks diverged on ' + n
        assert_close(r0['params'][n], p, 1e-4, 1e-6, 'parameter after the step: ' + n)
    assert r0['skipped'] and r1['skipped'] and r0['unchanged'] and r1['unchanged']


@pytest.mark.gpu
def test_training_backward_forms_agree():
    """The training step's gradients do not depend on how its backward is scheduled: the map CNN's backward handed to the library's
    side stream one rollout step at a time while the reverse sweep continues (STRIVE_TRAIN_OVERLAP_ROWS=1: eleven fork / join groups),
    in the default groups, or as one call after the sweep (STRIVE_TRAIN_OVERLAP=0); and the separate-rollout form
    (STRIVE_STACK_ROLLOUTS off).  All 174 tensors, at the noise level of the atomic additions."""
    DEV = 'cuda:0'
    m, sd = product_model(device=DEV)
    batch, map_idx, raster, dx = mg.g5_inputs(None, None)
    NA = batch.past.shape[0]
    eps_post = synth.f32(synth.counter_normal((NA, 32), 'g5/eps_post'))
    eps_prior = synth.f32(synth.counter_normal((NA, 32), 'g5/eps_prior'))
    env_g = synth.SyntheticMapEnv(raster.clone(), dx.clone()).to(DEV)

    def step(env_vars, stack=True):
        from strive_amd import _lib as L
        saved = {k: os.environ.get(k) for k in env_vars}
        os.environ.update(env_vars)
        L.sync_all_options_from_env()           # (the library's switches are options since ABI 17: the host maps STRIVE_<NAME> onto them)
        for k, v in env_vars.items():
            if k[len('STRIVE_'):].lower() in L.get_lib().option_names():
                assert L.get_lib().get_option(k[len('STRIVE_'):].lower()) == int(v), k
        m.stack_rollouts = stack
        try:
            out, ld, grads, _ = _product_step(m, batch.clone().to(DEV), map_idx.to(DEV), env_g, eps_post, eps_prior)
            torch.cuda.synchronize()
        finally:
            m.stack_rollouts = True
            for k, v in saved.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
            L.sync_all_options_from_env()
        return out, ld, {k: v.detach().clone() for k, v in grads.items()}
    ref_out, ref_ld, ref = step({'STRIVE_TRAIN_OVERLAP': '0'})
    # ('kept activations refused': ops._alloc_kept declines the buffer -- STRIVE_KEEP_MAX_BYTES, or 75 % of the free memory in
    #  production -- and the step takes the recomputing backward instead of failing: ADVICE r05)
    for what, env_vars, stack in (('one group per step', {'STRIVE_TRAIN_OVERLAP_ROWS': '1'}, True), ('default groups', {}, True),
                                  ('separate rollouts', {}, False), ('kept activations refused', {'STRIVE_KEEP_MAX_BYTES': '1'}, True)):
        out, ld, got = step(env_vars, stack)
        assert torch.equal(out['future_pred'], ref_out['future_pred']) and torch.equal(out['future_samp'], ref_out['future_samp']), what
        worst = ('', 0.0)
        for k in ref:
            den = float(ref[k].norm())
            if den < 1e-12:
                continue
            rel = float((got[k] - ref[k]).norm()) / den
            worst = (k, rel) if rel > worst[1] else worst
            assert rel < 2e-5, '%s: gradient %s differs from the one-call form by %.3g' % (what, k, rel)
        print('%s: worst relative difference to the one-call form %.3g (%s)' % (what, worst[1], worst[0]))


@pytest.mark.gpu
def test_map_cnn_backward_chunks_add_up():
    """strive_map_cnn_bwd over 300 crops (two internal chunks, 256 + 44: the rollout's batched call spans several) equals the
    sum of the gradients of the two parts computed by separate single-chunk calls (additivity of the weight gradient)."""
    from strive_amd import ops
    DEV = 'cuda:0'
    m, sd = product_model(device=DEV)
    raster, dx = synth.make_raster(1024, 1024, M=2)
    env = synth.SyntheticMapEnv(raster, dx).to(DEV)
    n = 300
    fr = np.zeros((n, 4))
    fr[:, 0] = synth.counter_uniform((n,), 'chunks/x', 20.0, 236.0)
    fr[:, 1] = synth.counter_uniform((n,), 'chunks/y', 20.0, 236.0)
    ang = synth.counter_uniform((n,), 'chunks/h', -np.pi, np.pi)
    fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
    pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(DEV).contiguous()
    mi = torch.tensor([i % 2 for i in range(n)]).to(DEV)
    d_feat = synth.f32(synth.counter_uniform((n, 64), 'chunks/df', -1.0, 1.0)).to(DEV)
    names = [k for k, _ in m.named_parameters() if k.startswith('map_')]

    def grads(lo, hi):
        for p in m.parameters():
            p.requires_grad_(True)
            p.grad = None
        with torch.enable_grad(), ops.weight_grad_mode(True):
            feat = ops.encode_map(m, pos[lo:hi].contiguous(), torch.arange(hi - lo).to(DEV), mi[lo:hi].contiguous(), env)
            feat.backward(d_feat[lo:hi].contiguous())
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if k in names}
    assert ops.keep_cnn_activations()           # (the default: conv1 .. conv4's outputs kept by the forward, not recomputed)
    whole, a, b = grads(0, n), grads(0, 256), grads(256, n)
    worst = 0.0
    for k in names:
        want = a[k] + b[k]
        rel = float((whole[k] - want).norm() / max(float(want.norm()), 1e-30))
        worst = max(worst, rel)
        assert rel < 2e-5, 'map CNN gradient %s: chunked call differs from the sum of its parts by %.3g' % (k, rel)
    print('worst relative difference: %.3g' % worst)
    # the same call with the forward recomputed inside the backward (STRIVE_KEEP_CNN_ACTIVATIONS=0, the round-4 form): the kept
    # activations are the very numbers the recompute produces, only the order of the atomic additions differs
    os.environ['STRIVE_KEEP_CNN_ACTIVATIONS'] = '0'
    try:
        assert not ops.keep_cnn_activations()
        rec = grads(0, n)
    finally:
        del os.environ['STRIVE_KEEP_CNN_ACTIVATIONS']
    worst = 0.0
    for k in names:
        rel = float((whole[k] - rec[k]).norm() / max(float(rec[k].norm()), 1e-30))
        worst = max(worst, rel)
        # (conv5 / conv6's kept outputs come from the fused tail, whose conv6 adds its three product terms in another order than the
        #  recompute's conv_bf6s_kernel: ulps in the activations, ~1e-5 in a gradient -- the noise level of the oracle comparisons)
        assert rel < 5e-5, 'map CNN gradient %s: kept activations vs recomputed forward differ by %.3g' % (k, rel)
    print('kept vs recomputed: worst relative difference %.3g' % worst)



@pytest.mark.gpu
def test_cnn_backward_measurement_hook():
    """strive_map_cnn_bwd_bench_dgrad (bench.py's training-line roofline): launches on the workspace of a finished backward call,
    refuses layers without a data gradient and more than one chunk, and leaves the next real backward unchanged."""
    from strive_amd import ops, _lib as L
    DEV = 'cuda:0'
    m, sd = product_model(device=DEV)
    raster, dx = synth.make_raster(1024, 1024, M=2)
    env = synth.SyntheticMapEnv(raster, dx).to(DEV)
    n = 40
    fr = np.zeros((n, 4))
    fr[:, 0] = synth.counter_uniform((n,), 'hook/x', 20.0, 236.0)
    fr[:, 1] = synth.counter_uniform((n,), 'hook/y', 20.0, 236.0)
    fr[:, 2] = 1.0
    pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(DEV).contiguous()
    mi = torch.zeros(n, dtype=torch.long, device=DEV)
    d_feat = synth.f32(synth.counter_uniform((n, 64), 'hook/df', -1.0, 1.0)).to(DEV)

    def grads():
        for p in m.parameters():
            p.requires_grad_(True)
            p.grad = None
        with torch.enable_grad(), ops.weight_grad_mode(True):
            ops.encode_map(m, pos, torch.arange(n).to(DEV), mi, env).backward(d_feat)
        return torch.cat([p.grad.reshape(-1) for k, p in m.named_parameters() if k.startswith('map_')]).clone()
    first = grads()
    lib = L.get_lib()
    wsb = lib.query('strive_map_cnn_bwd_workspace_bytes', n)
    ws = ops._workspace(torch.device(DEV), wsb, 'cnn_bwd')
    st = L.stream_ptr(pos)
    for layer in (5, 4, 3, 2, 1):
        lib.call('strive_map_cnn_bwd_bench_dgrad', layer, n, L.ptr(ws), ws.numel(), st)
    torch.cuda.synchronize()
    for layer, nn in ((0, n), (6, n), (1, 257)):
        with pytest.raises(L.StriveHipError):
            lib.call('strive_map_cnn_bwd_bench_dgrad', layer, nn, L.ptr(ws), ws.numel(), st)
    again = grads()
    assert float((again - first).norm() / first.norm()) < 1e-6


@pytest.mark.gpu
def test_lagged_absmax_serves_the_right_tensor_a_bounded_lag_behind():
    """params.lagged_absmax (the training step's operand scales without a host wait): an entry belongs to ONE tensor -- a tensor
    freed and another one allocated (possibly at the same address, same shape) gets its own exact first-sight value, never the
    dead tensor's bound -- and the value served is the read-back of exactly two calls earlier plus the slack, so with a slack that
    bounds two steps' growth it never undercuts the true max |w| and does not depend on when the copies happen to arrive."""
    from strive_amd import params
    dev = 'cuda:0'
    params._ABSMAX.clear(); params._LAG.clear(); params._LAG_BATCH.clear()
    with params.lagged_absmax(slack=0.25):
        a = torch.full((4096,), 3.0, device=dev)
        assert params.absmax(a) == 3.0                         # first sight: exact
        addr = a.data_ptr()
        del a
        torch.cuda.synchronize()
        b = torch.full((4096,), 7.0, device=dev)               # the caching allocator hands the block out again
        same_address = b.data_ptr() == addr
        assert params.absmax(b) == 7.0, 'a new tensor must not inherit a dead tensor\'s bound (same address: %s)' % same_address
        # the entry kept `a`'s storage alive, so `b` cannot alias it while the entry exists
        assert not same_address
        served = []
        for step in range(8):                                  # an "optimiser" that grows max |w| by 0.1 per step
            b.add_(0.1)
            params.prefetch_absmax([b])
            v = params.absmax(b)
            true = float(b.abs().max())
            assert v >= true - 1e-6, 'step %d: served bound %.4f below the true max %.4f' % (step, v, true)
            served.append(round(v - true, 4))
        # fixed two-call lag: from the third call on the bound is (value two calls ago) + slack = true - 0.2 + 0.25
        assert all(abs(s - 0.05) < 1e-3 for s in served[2:]), served
    params._ABSMAX.clear(); params._LAG.clear(); params._LAG_BATCH.clear()
