"""Kernel LOGIC on the CPU: the unmodified strive_amd/csrc/*.hip sources compiled as host C++ against
tests/hipemu (fibers for threads, rendezvous for barriers / shuffles / MFMA) and driven through the same C ABI,
checked against the oracle and the golden vectors.  This is test infrastructure for a GPU-less build container:
it validates indexing, LDS staging, MFMA fragment layouts, the reverse-time backward and the ABI plumbing -- it
says nothing about performance and is never loaded by the product."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import make_golden as mg
from util import golden, oracle_model, product_model, assert_close, poisoned_workspace, nan_empty
from strive_amd import _lib as L, params, synth
from strive_amd.constants import NUSC_BIKE_PARAMS
from oracle import mapenv, losses as olosses
from oracle import model as om
from oracle.geometry import Normalizer

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))


@pytest.fixture(scope='module')
def emu():
    import build as emu_build
    return L.StriveLib(emu_build.build(), require_all=True)


@pytest.fixture(scope='module')
def sd():
    return product_model()[1]


def test_crop_and_coll_point(emu):
    g = golden('g2_crop.npz')
    raster, dx, frame, mapixes, lw = mg.g2_inputs()
    env = synth.SyntheticMapEnv(raster, dx)
    mp = params.pack_map(env, 'cpu')
    N = frame.shape[0]
    out = torch.zeros((N, 4, 256, 256), dtype=torch.uint8)
    mi = mapixes.int()
    emu.call('strive_map_crop_u8', mp.ref(), L.ptr(frame), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi), N, L.ptr(out), None)
    assert torch.equal(out, mapenv.map_crop(raster, dx, frame, mapixes, env.bounds))
    for i in (0, 1, 7):
        assert np.array_equal(np.packbits(out[i].numpy()), g['crop_full_%d' % i])
    ok = ~torch.isnan(frame[:, 0])
    cars, lws, mis = frame[ok].contiguous(), lw[ok].contiguous(), mi[ok].contiguous()
    gl, gw = mapenv.coll_grid_size(dx, lws)
    pt = torch.zeros((cars.shape[0], 2))
    cnt = torch.zeros((cars.shape[0],), dtype=torch.int32)
    emu.call('strive_coll_point', mp.ref(), L.ptr(cars), L.ptr(lws), L.ptr(mis), cars.shape[0], gl, gw,
             L.ptr(torch.linspace(-1, 1, gl)), L.ptr(torch.linspace(-1, 1, gw)), L.ptr(pt), L.ptr(cnt), None)
    np.testing.assert_allclose(pt.numpy(), g['coll_pt'], rtol=0, atol=1e-3, equal_nan=True)


def test_mlp_and_gnn(emu, sd):
    g1, g3 = golden('g1_ops.npz'), golden('g3_gnn.npz')
    x = synth.f32(synth.counter_uniform((6, 38), 'g1/mlp_in', -1.0, 1.0))
    mp = params.pack_mlp(sd, 'past_encoder')
    y = torch.zeros((6, 64))
    emu.call('strive_mlp_fwd', mp.ref(), L.ptr(x), 6, L.ptr(y), None)
    assert_close(y, g1['mlp_past_encoder'], 1e-4, 1e-5, 'mlp')
    batch, map_idx, raster, dx = mg.build_inputs(mg.G3_SIZES, 'g3')
    NA = batch.past.shape[0]
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    for name, prefix, fin in (('decoder', 'decoder_net', 164), ('prior', 'prior_net', 130), ('posterior', 'posterior_net', 194)):
        x = synth.f32(synth.counter_uniform((NA, fin), 'g3/x/' + name, -1.0, 1.0))
        pos = batch.past[:, -1, :4].clone().contiguous()
        if name == 'prior':
            pos[1, 0] = float('nan')
        gp = params.pack_gnn(sd, prefix, 2)
        wsb = emu.query('strive_gnn_workspace_bytes', gp.ref(), sc.ref())
        ws = torch.zeros(wsb, dtype=torch.uint8)
        out = torch.zeros((NA, g3[name + '_out'].shape[1]))
        emu.call('strive_gnn_fwd', gp.ref(), sc.ref(), L.ptr(x), L.ptr(pos), L.ptr(batch.sem.contiguous()), L.ptr(out),
                 L.ptr(ws), wsb, None)
        assert_close(out, g3[name + '_out'], 1e-4, 1e-5, name)


def test_map_cnn_one_agent(emu, sd, monkeypatch):
    """Both forward chains: batches of up to 32 samples give conv1 one workgroup per tile and conv3 / conv4 one 32-channel block per
    workgroup (STRIVE_CNN_SMALL_BATCH=0 selects the throughput chain); the GroupNorm moments of both are fp32 sums over the same 16-value units
    added in float64, so the two give the same features."""
    raster, dx, frame, mapixes, lw = mg.g2_inputs()
    env = synth.SyntheticMapEnv(raster, dx)
    fr = frame[7:8].contiguous()
    mi = mapixes[7:8].int().contiguous()
    crop = mapenv.map_crop(raster, dx, fr, mi.long(), env.bounds)
    want = om.map_cnn(sd, crop.float())
    cnn, mp = params.pack_cnn(sd), params.pack_map(env, 'cpu')
    wsb = emu.query('strive_map_cnn_workspace_bytes', 1)
    ws = torch.zeros(wsb, dtype=torch.uint8)
    feat = torch.zeros((1, 64))
    emu.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(fr), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi), 1, L.ptr(feat),
             L.ptr(ws), wsb, None)
    assert_close(feat, want, 1e-4, 1e-5, 'cnn')
    # the workspace arrives uninitialised: with every byte 0xFF (NaN) the same bits
    ws_nan = torch.full((wsb,), 0xFF, dtype=torch.uint8)
    again = torch.zeros((1, 64))
    emu.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(fr), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi), 1, L.ptr(again),
             L.ptr(ws_nan), wsb, None)
    assert torch.equal(again, feat), 'the CNN read workspace it had not written'
    monkeypatch.setenv('STRIVE_CNN_SMALL_BATCH', '0')
    big = torch.zeros((1, 64))
    emu.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(fr), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi), 1, L.ptr(big),
             L.ptr(ws_nan.fill_(0xFF)), wsb, None)
    assert_close(big, want, 1e-4, 1e-5, 'cnn, throughput chain')
    assert torch.equal(big, feat), 'the two chains: %.3g apart' % float((big - feat).abs().max())


def _rollout(emu, sd, sizes, FT, NS=1, ext=False, NC=2):
    batch, map_idx, raster, dx = mg.build_inputs(sizes, 'emu', NC=NC)
    env = synth.SyntheticMapEnv(raster, dx)
    orc = oracle_model(sd, NC=NC)
    with torch.no_grad():
        emb = orc.embed(batch, map_idx, env) if FT > 1 else None
    NA = batch.past.shape[0]
    if emb is None:   # FT == 1 needs no map feature from the CNN: use counter-generated features
        emb = {'map_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/mf', -1, 1)),
               'past_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/pf', -1, 1)),
               'prior_out': (torch.zeros((NA, 32)), torch.ones((NA, 32)))}
    z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='emu/z')
    if NS > 1:
        z = torch.stack([z, synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='emu/z2')], dim=1)
    z = z.contiguous().requires_grad_(True)
    extf = batch.future_gt[batch.ptr[:-1]][:, :FT, :4].contiguous() if ext else None
    pred = orc.decode(batch, emb['map_feat'], emb['past_feat'], z, map_idx, env, ext_future=extf, nfuture=FT)
    rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'emu/rw', -1.0, 1.0))
    gz, = torch.autograd.grad((pred * rw).sum(), [z])
    sn, an = orc.get_normalizer(), orc.get_att_normalizer()
    dec = params.pack_decoder(sd, NC, env, 'cpu', sn, an, NUSC_BIKE_PARAMS)
    sc = params.pack_scenes(batch.ptr, NS, 'cpu')
    R = NA * NS
    tb = emu.query('strive_rollout_tape_bytes', dec.ref(), sc.ref(), FT)
    wb = emu.query('strive_rollout_workspace_bytes', dec.ref(), sc.ref(), FT)
    tape, ws = torch.zeros(tb, dtype=torch.uint8), torch.zeros(wb, dtype=torch.uint8)
    traj = torch.zeros((R, FT, 4))
    zz = z.detach().reshape(R, 32).contiguous()
    mi = map_idx[batch.batch].int().contiguous()
    lw, sem = batch.lw.contiguous(), batch.sem.contiguous()
    emu.call('strive_rollout_fwd', dec.ref(), sc.ref(), L.ptr(batch.past[:, -1, :].contiguous()), L.ptr(lw), L.ptr(sem),
             L.ptr(emb['past_feat'].contiguous()), L.ptr(emb['map_feat'].contiguous()), L.ptr(zz), L.ptr(mi), L.ptr(extf), FT,
             L.ptr(traj), L.ptr(tape), tb, L.ptr(ws), wb, None)
    dz = torch.zeros((R, 32))
    emu.call('strive_rollout_bwd', dec.ref(), sc.ref(), L.ptr(lw), L.ptr(sem), L.ptr(zz), L.ptr(extf), FT,
             L.ptr(rw.reshape(R, FT, 4).contiguous()), L.ptr(dz), L.ptr(tape), tb, L.ptr(ws), wb, None)
    assert_close(traj, pred.detach().reshape(R, FT, 4), 1e-4, 1e-5, 'rollout fwd')
    assert_close(dz, gz.reshape(R, 32), 2e-3, 1e-6 + 1e-4 * float(gz.abs().max()), 'rollout bwd')


@pytest.mark.parametrize('sizes,NS,ext', [([3, 1, 5], 1, False), ([4, 2], 1, True), ([2, 3], 2, False), ([19], 1, False)])
def test_rollout_single_step(emu, sd, sizes, NS, ext):
    """FT = 1: GNN + bicycle + local transform and their adjoints (no CNN / GRU in a single step)."""
    _rollout(emu, sd, sizes, 1, NS=NS, ext=ext)


def test_rollout_single_step_five_classes(emu):
    """NC = 5 (reduce_cats): node input 167, edge input 142 -- every sem-bearing offset of the kernels moves."""
    sd5 = product_model(NC=5, key='weights5')[1]
    _rollout(emu, sd5, [3, 1, 4], 1, NC=5)
    _rollout(emu, sd5, [2, 3], 1, NS=2, NC=5)


def test_rollout_two_steps(emu, sd):
    """FT = 2 adds the GRU memory step and one fused crop+CNN evaluation per agent (and their place in the
    reverse sweep)."""
    _rollout(emu, sd, [2, 1], 2)


@pytest.mark.slow
def test_rollout_three_steps_ext(emu, sd):
    _rollout(emu, sd, [3, 1], 3, ext=True)


def test_veh_coll_fwd_bwd(emu, sd):
    g = golden('g5_losses.npz')
    batch, map_idx, raster, dx = mg.g5_inputs(None, None)
    orc = oracle_model(sd)
    veh_att = orc.get_att_normalizer().unnormalize(batch.lw)
    traj = olosses.interp_traj(orc.get_normalizer().unnormalize(torch.from_numpy(g['adv_pred'])), 3).contiguous()
    NA, T, _ = traj.shape
    from strive_amd.ops import SceneInfo
    from strive_amd.losses.adv_gen_nusc import _linspace5
    info = SceneInfo(batch.ptr, 'cpu')
    sc = info.pack(1)
    rad = (veh_att[:, 1] / 2.).contiguous()
    cent = _linspace5(-(veh_att[:, 0] / 2.) + rad, (veh_att[:, 0] / 2.) - rad).contiguous()
    pen = torch.zeros((T, info.P))
    hit = torch.zeros((T, info.P), dtype=torch.uint8)
    amin = torch.zeros((T, info.P), dtype=torch.uint8)
    emu.call('strive_veh_coll_fwd', sc.ref(), L.ptr(info.pair_off), info.P, L.ptr(traj), T, L.ptr(cent), L.ptr(rad), 0.1,
             L.ptr(pen), L.ptr(hit), L.ptr(amin), None)
    tc = traj.clone().requires_grad_(True)
    vc = olosses.VehColl(veh_att, ptr=batch.ptr, buffer_dist=0.1)
    dense, cmask = vc(tc, return_raw=True)
    # slot (t, i, j) -> dense[t, i, j]
    sizes = info.sizes.tolist()
    si, sj = [], []
    for b, n in enumerate(sizes):
        lo = int(batch.ptr[b])
        for i in range(n):
            for j in range(n):
                si.append(lo + i)
                sj.append(lo + j)
    si, sj = torch.tensor(si), torch.tensor(sj)
    assert_close(pen, dense.detach()[:, si, sj], 1e-4, 1e-5, 'pen')
    assert torch.equal(hit.bool() & (si != sj).view(1, -1), cmask[:, si, sj])
    wgt = synth.f32(synth.counter_uniform((T, info.P), 'emu/vw', 0.0, 1.0)) * (hit.bool() & (si != sj).view(1, -1))
    (dense[:, si, sj] * wgt).sum().backward()
    d_traj = torch.zeros_like(traj)
    emu.call('strive_veh_coll_bwd', sc.ref(), L.ptr(info.pair_off), info.P, L.ptr(traj), T, L.ptr(cent), L.ptr(rad), 0.1,
             L.ptr(wgt.contiguous()), L.ptr(amin), L.ptr(d_traj), None)
    assert_close(d_traj, tc.grad, 1e-3, 1e-5, 'veh bwd')


def test_interp_traj_fwd_bwd(emu):
    """strive_interp_traj_fwd/bwd vs the oracle's F.interpolate + renormalisation (reference
    src/losses/adv_gen_nusc.py:13-26) and vs the golden vector generated from the reference itself."""
    from strive_amd.ops import _interp_taps
    g = golden('g1_ops.npz')
    N, T, S = 7, 12, 3
    x = synth.f32(synth.counter_normal((N, T, 4), 'interp/x'))
    x[:, :, 2:4] = x[:, :, 2:4] / x[:, :, 2:4].norm(dim=-1, keepdim=True)
    i0, i1, w0, w1 = _interp_taps(T, S, 'cpu')
    out = torch.zeros((N, T * S, 4))
    emu.call('strive_interp_traj_fwd', L.ptr(x), N, T, T * S, L.ptr(i0), L.ptr(i1), L.ptr(w0), L.ptr(w1), L.ptr(out), None)
    xr = x.clone().requires_grad_(True)
    want = olosses.interp_traj(xr, S)
    np.testing.assert_allclose(out.numpy(), want.detach().numpy(), rtol=1e-5, atol=2e-6)
    d_out = synth.f32(synth.counter_normal((N, T * S, 4), 'interp/g'))
    want.backward(d_out)
    d_in = torch.zeros_like(x)
    emu.call('strive_interp_traj_bwd', L.ptr(x), L.ptr(d_out), N, T, T * S, S, L.ptr(i0), L.ptr(i1), L.ptr(w0), L.ptr(w1),
             L.ptr(d_in), None)
    np.testing.assert_allclose(d_in.numpy(), xr.grad.numpy(), rtol=2e-5, atol=5e-6)
    # golden: the reference's interp_traj on the g1 trajectory
    if True:
        xi = synth.f32(synth.counter_uniform((5, 12, 4), 'g1/traj', -2.0, 2.0)).contiguous()   # make_golden.py's input
        Ng, Tg, _ = xi.shape
        a0, a1, b0, b1 = _interp_taps(Tg, 3, 'cpu')
        og = torch.zeros((Ng, Tg * 3, 4))
        emu.call('strive_interp_traj_fwd', L.ptr(xi), Ng, Tg, Tg * 3, L.ptr(a0), L.ptr(a1), L.ptr(b0), L.ptr(b1), L.ptr(og), None)
        np.testing.assert_allclose(og.numpy(), g['interp'], rtol=1e-5, atol=3e-6)


def test_rect_iou_kernel(emu):
    """strive_rect_iou vs the float64 oracle on random vehicle boxes (overlapping, disjoint, identical, NaN)."""
    from oracle.geometry import rect_iou
    P = 96
    u = synth.counter_uniform((P, 8), 'iouk/u', -1.0, 1.0)
    a = np.stack([4 * u[:, 0], 4 * u[:, 1], np.cos(3.1 * u[:, 2]), np.sin(3.1 * u[:, 2])], axis=1)
    b = np.stack([4 * u[:, 3], 4 * u[:, 4], np.cos(3.1 * u[:, 5]), np.sin(3.1 * u[:, 5])], axis=1)
    la = np.stack([4.5 + u[:, 6], 1.9 + 0.3 * u[:, 7]], axis=1)
    lb = np.stack([4.0 - 0.5 * u[:, 7], 1.8 + 0.2 * u[:, 6]], axis=1)
    b[0], lb[0] = a[0], la[0]                       # identical boxes
    b[1, :2] = a[1, :2] + 40.0                      # far apart
    b[2, 0] = np.nan                                # NaN pose
    a[3, 2:] *= 7.5                                 # un-normalised heading vector (atan2 only sees the direction)
    ta, tb, tla, tlb = (synth.f32(x).contiguous() for x in (a, b, la, lb))
    out = torch.zeros((P,), dtype=torch.float64)
    emu.call('strive_rect_iou', L.ptr(ta), L.ptr(tla), L.ptr(tb), L.ptr(tlb), P, L.ptr(out), None)
    want = np.array([rect_iou(ta[i].numpy(), tla[i].numpy(), tb[i].numpy(), tlb[i].numpy()) for i in range(P)])
    assert np.isnan(out[2].item()) and np.isnan(want[2])
    ok = ~np.isnan(want)
    np.testing.assert_allclose(out.numpy()[ok], want[ok], rtol=0, atol=1e-12)
    assert abs(out[0].item() - 1.0) < 1e-12 and out[1].item() == 0.0
    assert (want[ok] > 0.02).sum() > 10 and (want[ok] == 0).sum() > 10      # both outcomes are exercised


def test_map_cnn_eight_agents(emu, sd, monkeypatch):
    """Eight agents: conv5's workgroups take 7 samples each (one full, one with a single sample), conv6 / fc take 8 / 4."""
    raster, dx, frame, mapixes, lw = mg.g2_inputs()
    env = synth.SyntheticMapEnv(raster, dx)
    n = 8
    fr = np.zeros((n, 4))
    fr[:, 0] = synth.counter_uniform((n,), 'e8/x', 40.0, 200.0)
    fr[:, 1] = synth.counter_uniform((n,), 'e8/y', 40.0, 200.0)
    ang = synth.counter_uniform((n,), 'e8/h', -np.pi, np.pi)
    fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
    fr = synth.f32(fr).contiguous()
    mi = torch.tensor([i % 2 for i in range(n)], dtype=torch.int32)
    crop = mapenv.map_crop(raster, dx, fr, mi.long(), env.bounds)
    want = om.map_cnn(sd, crop.float())
    cnn, mp = params.pack_cnn(sd), params.pack_map(env, 'cpu')
    wsb = emu.query('strive_map_cnn_workspace_bytes', n)
    ws = torch.zeros(wsb, dtype=torch.uint8)
    feat = torch.zeros((n, 64))
    emu.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(fr), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi), n, L.ptr(feat),
             L.ptr(ws), wsb, None)
    assert_close(feat, want, 1e-4, 1e-5, 'cnn x8')
    # samples per workgroup of the fused tail in the small-batch chain (1 by default): per sample the same sums in the same order
    for tail_s, m_ in (('2', 8), ('4', 8), ('2', 5)):
        monkeypatch.setenv('STRIVE_CNN_TAIL_S', tail_s)
        other = torch.zeros((m_, 64))
        emu.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(fr[:m_].contiguous()), L.f4([0] * 4), L.f4([1] * 4),
                 L.ptr(mi[:m_].contiguous()), m_, L.ptr(other), L.ptr(ws), wsb, None)
        assert torch.equal(other, feat[:m_]), 'tail with %s samples per workgroup' % tail_s
    monkeypatch.delenv('STRIVE_CNN_TAIL_S')
    # the fused tail (conv5 + conv6 + Linear in one kernel, what strive_map_cnn_fwd runs) against the separate kernels of the
    # training recompute on the SAME conv4 output left in the workspace; 6 of the 8 poses = one full and one half workgroup.  The
    # separate kernels read the throughput chain's statistics slots, so this half runs that chain (the call above ran the small one).
    monkeypatch.setenv('STRIVE_CNN_SMALL_BATCH', '0')
    for m_ in (8, 6):
        f_fused, f_sep = torch.zeros((m_, 64)), torch.zeros((m_, 64))
        args = (mp.ref(), cnn.ref())
        tail = (L.ptr(fr[:m_].contiguous()), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi[:m_].contiguous()), m_)
        wsb2 = emu.query('strive_map_cnn_workspace_bytes', m_)
        ws2 = torch.zeros(wsb2, dtype=torch.uint8)
        emu.call('strive_map_cnn_fwd', *args, *tail, L.ptr(f_fused), L.ptr(ws2), wsb2, None)
        for layer in (4, 5, 6):
            emu.call('strive_map_cnn_bench_layer', *args, layer, *tail, L.ptr(f_sep), L.ptr(ws2), wsb2, None)
        assert_close(f_fused, f_sep, 2e-6, 2e-6, 'fused tail vs separate kernels (%d poses)' % m_)
        assert_close(f_fused, want[:m_], 1e-4, 1e-5, 'cnn x%d' % m_)


def test_map_cnn_backward_over_kept_ranges(emu, sd):
    """The forward that keeps its activations (strive_map_cnn_fwd_keep, written in two calls at row offsets like the rollout does
    step by step) gives the features of strive_map_cnn_fwd; the backward over the kept rows as one call (strive_map_cnn_bwd_kept)
    equals the recomputing backward (strive_map_cnn_bwd) and the sum of the per-range calls (strive_map_cnn_bwd_kept_range: what
    the training rollout hands its side stream group by group)."""
    raster, dx, frame, mapixes, lw = mg.g2_inputs()
    env = synth.SyntheticMapEnv(raster, dx)
    n = 3
    fr = np.zeros((n, 4))
    fr[:, 0] = synth.counter_uniform((n,), 'kr/x', 40.0, 200.0)
    fr[:, 1] = synth.counter_uniform((n,), 'kr/y', 40.0, 200.0)
    ang = synth.counter_uniform((n,), 'kr/h', -np.pi, np.pi)
    fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
    fr = synth.f32(fr).contiguous()
    mi = torch.tensor([i % 2 for i in range(n)], dtype=torch.int32)
    d_feat = synth.f32(synth.counter_uniform((n, 64), 'kr/df', -1.0, 1.0)).contiguous()
    cnn, mp = params.pack_cnn(sd), params.pack_map(env, 'cpu')
    z4, o4 = L.f4([0] * 4), L.f4([1] * 4)
    wsb = emu.query('strive_map_cnn_workspace_bytes', n)
    ws = torch.zeros(wsb, dtype=torch.uint8)
    feat, feat_k = torch.zeros((n, 64)), torch.zeros((n, 64))
    emu.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(fr), z4, o4, L.ptr(mi), n, L.ptr(feat), L.ptr(ws), wsb, None)
    kb = emu.query('strive_map_cnn_keep_bytes', n)
    kept = torch.full((kb,), 0xFF, dtype=torch.uint8)          # (NaN bytes: every kept row must be written before it is read)
    for lo, hi in ((0, 2), (2, 3)):
        emu.call('strive_map_cnn_fwd_keep', mp.ref(), cnn.ref(), L.ptr(fr[lo:hi].contiguous()), z4, o4, L.ptr(mi[lo:hi].contiguous()), hi - lo,
                 L.ptr(feat_k[lo:hi]), L.ptr(ws), wsb, L.ptr(kept), kb, n, lo, None)
    assert_close(feat_k, feat, 2e-6, 2e-6, 'features of the keeping forward')       # (the standard chain for conv3 / conv4; same sums)
    npar = emu.query('strive_map_cnn_param_count')
    bwb = emu.query('strive_map_cnn_bwd_workspace_bytes', n)
    bws = torch.zeros(bwb, dtype=torch.uint8)
    g_rec, g_kept, g_rng = torch.zeros(npar), torch.zeros(npar), torch.zeros(npar)
    emu.call('strive_map_cnn_bwd', mp.ref(), cnn.ref(), L.ptr(fr), z4, o4, L.ptr(mi), n, L.ptr(d_feat), L.ptr(g_rec), L.ptr(bws), bwb, None)
    emu.call('strive_map_cnn_bwd_kept', mp.ref(), cnn.ref(), L.ptr(fr), z4, o4, L.ptr(mi), n, L.ptr(d_feat), L.ptr(g_kept), L.ptr(kept), kb,
             L.ptr(bws), bwb, None)
    for lo, hi in ((2, 3), (0, 2)):         # (the rollout hands its last steps over first)
        emu.call('strive_map_cnn_bwd_kept_range', mp.ref(), cnn.ref(), L.ptr(fr[lo:hi].contiguous()), z4, o4, L.ptr(mi[lo:hi].contiguous()),
                 hi - lo, L.ptr(d_feat[lo:hi].contiguous()), L.ptr(g_rng), L.ptr(kept), kb, n, lo, L.ptr(bws), bwb, None)
    assert bool(torch.isfinite(g_kept).all()) and float(g_rec.norm()) > 0.0
    for what, g in (('kept vs recomputed', g_kept), ('ranges vs one call', g_rng)):
        ref = g_rec if what.startswith('kept') else g_kept
        rel = float((g - ref).norm() / ref.norm())
        assert rel < 2e-5, '%s: flat map-CNN gradient differs by %.3g' % (what, rel)
    # rows outside the kept arrays are refused
    with pytest.raises(L.StriveHipError):
        emu.call('strive_map_cnn_bwd_kept_range', mp.ref(), cnn.ref(), L.ptr(fr), z4, o4, L.ptr(mi), n, L.ptr(d_feat), L.ptr(g_rng), L.ptr(kept),
                 kb, n, 1, L.ptr(bws), bwb, None)


# ------------------------------------------------------------------------------------------------
# training backward: weight gradients (flat buffers in named_parameters() order) vs torch autograd of the oracle
# ------------------------------------------------------------------------------------------------

def _grad_sd(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def _flat_grads(sdg, prefix):
    return torch.cat([(v.grad if v.grad is not None else torch.zeros_like(v)).reshape(-1) for k, v in sdg.items()
                      if k.startswith(prefix + '.')])


def _check_flat(got, want, sdg, prefix, rtol=2e-3, what=''):
    """per-parameter comparison so a failure names the tensor"""
    off = 0
    for k, v in sdg.items():
        if not k.startswith(prefix + '.'):
            continue
        n = v.numel()
        w = want[off:off + n]
        g = got[off:off + n]
        scale = float(w.abs().max())
        assert_close(g, w, rtol, 1e-6 + 2e-4 * scale, '%s grad %s' % (what, k))
        off += n
    assert off == got.numel() == want.numel()


def test_mlp_bwd_weight_and_input_gradients(emu, sd):
    sdg = _grad_sd(sd)
    x = synth.f32(synth.counter_uniform((7, 38), 'emu/mlpb/x', -1.0, 1.0)).requires_grad_(True)
    y = om.mlp(sdg, 'past_encoder', x)
    rw = synth.f32(synth.counter_uniform(tuple(y.shape), 'emu/mlpb/r', -1.0, 1.0))
    (y * rw).sum().backward()
    mp = params.pack_mlp(sd, 'past_encoder')
    n = emu.query('strive_mlp_param_count', mp.ref())
    want = _flat_grads(sdg, 'past_encoder')
    assert n == want.numel()
    dp = torch.zeros(n)
    dx = torch.zeros((7, 38))
    emu.call('strive_mlp_bwd', mp.ref(), L.ptr(x.detach().contiguous()), L.ptr(rw.contiguous()), 7, L.ptr(dx), L.ptr(dp), None)
    assert_close(dx, x.grad, 2e-3, 1e-6, 'mlp dx')
    _check_flat(dp, want, sdg, 'past_encoder', what='mlp')
    # accumulation + no input gradient requested
    emu.call('strive_mlp_bwd', mp.ref(), L.ptr(x.detach().contiguous()), L.ptr(rw.contiguous()), 7, None, L.ptr(dp), None)
    _check_flat(dp, 2 * want, sdg, 'past_encoder', what='mlp (accumulated)')


@pytest.mark.parametrize('name,prefix,fin', [('decoder', 'decoder_net', 164), ('posterior', 'posterior_net', 194)])
def test_gnn_bwd_weight_and_input_gradients(emu, sd, name, prefix, fin):
    sdg = _grad_sd(sd)
    batch, map_idx, raster, dx_ = mg.build_inputs([3, 5, 1, 2], 'emu/gnnb')
    NA = batch.past.shape[0]
    x = synth.f32(synth.counter_uniform((NA, fin), 'emu/gnnb/x/' + name, -1.0, 1.0)).requires_grad_(True)
    pos = batch.past[:, -1, :4].clone().contiguous()
    y = om.interaction_net(sdg, prefix, x, pos, batch.sem, batch.edge_index)
    rw = synth.f32(synth.counter_uniform(tuple(y.shape), 'emu/gnnb/r/' + name, -1.0, 1.0))
    (y * rw).sum().backward()
    gp = params.pack_gnn(sd, prefix, 2)
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    n = emu.query('strive_gnn_param_count', gp.ref())
    want = _flat_grads(sdg, prefix)
    assert n == want.numel()
    wsb = emu.query('strive_gnn_bwd_workspace_bytes', gp.ref(), sc.ref())
    ws = torch.zeros(wsb, dtype=torch.uint8)
    dp = torch.zeros(n)
    dxo = torch.zeros((NA, fin))
    emu.call('strive_gnn_bwd', gp.ref(), sc.ref(), L.ptr(x.detach().contiguous()), L.ptr(pos), L.ptr(batch.sem.contiguous()),
             L.ptr(rw.contiguous()), L.ptr(dxo), L.ptr(dp), L.ptr(ws), wsb, None)
    assert_close(dxo, x.grad, 2e-3, 1e-6 + 2e-4 * float(x.grad.abs().max()), name + ' dx')
    _check_flat(dp, want, sdg, prefix, what=name)


def _rollout_train(emu, sd, sizes, FT):
    sdg = _grad_sd(sd)
    batch, map_idx, raster, dx_ = mg.build_inputs(sizes, 'emu/rt')
    env = synth.SyntheticMapEnv(raster, dx_)
    NA = batch.past.shape[0]
    orc = oracle_model(sdg)
    mf = synth.f32(synth.counter_uniform((NA, 64), 'emu/rt/mf', -1, 1)).requires_grad_(True)
    pf = synth.f32(synth.counter_uniform((NA, 64), 'emu/rt/pf', -1, 1)).requires_grad_(True)
    z = synth.f32(synth.counter_normal((NA, 32), 'emu/rt/z')).requires_grad_(True)
    pred = orc.decode(batch, mf, pf, z, map_idx, env, nfuture=FT)
    rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'emu/rt/rw', -1.0, 1.0))
    (pred * rw).sum().backward()
    sn, an = orc.get_normalizer(), orc.get_att_normalizer()
    dec = params.pack_decoder(sd, 2, env, 'cpu', sn, an, NUSC_BIKE_PARAMS)
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    tb = emu.query('strive_rollout_tape_bytes', dec.ref(), sc.ref(), FT)
    wb = emu.query('strive_rollout_train_workspace_bytes', dec.ref(), sc.ref(), FT)
    tape, ws = torch.zeros(tb, dtype=torch.uint8), torch.zeros(wb, dtype=torch.uint8)
    traj = torch.zeros((NA, FT, 4))
    zz = z.detach().contiguous()
    mi = map_idx[batch.batch].int().contiguous()
    lw, sem = batch.lw.contiguous(), batch.sem.contiguous()
    emu.call('strive_rollout_fwd', dec.ref(), sc.ref(), L.ptr(batch.past[:, -1, :].contiguous()), L.ptr(lw), L.ptr(sem),
             L.ptr(pf.detach().contiguous()), L.ptr(mf.detach().contiguous()), L.ptr(zz), L.ptr(mi), None, FT,
             L.ptr(traj), L.ptr(tape), tb, L.ptr(ws), wb, None)
    assert_close(traj, pred.detach(), 1e-4, 1e-5, 'rollout fwd')
    ng, nr, nc = emu.query('strive_gnn_param_count', dec.struct.gnn), emu.query('strive_gru_param_count'), \
        emu.query('strive_map_cnn_param_count')
    dz, dpf, dmf = torch.zeros((NA, 32)), torch.zeros((NA, 64)), torch.zeros((NA, 64))
    dg, dr, dc = torch.zeros(ng), torch.zeros(nr), torch.zeros(nc)
    emu.call('strive_rollout_bwd_train', dec.ref(), sc.ref(), L.ptr(lw), L.ptr(sem), L.ptr(zz), None, L.ptr(mi), FT,
             L.ptr(rw.contiguous()), L.ptr(dz), L.ptr(dpf), L.ptr(dmf), L.ptr(dg), L.ptr(dr), L.ptr(dc), L.ptr(tape), tb,
             L.ptr(ws), wb, None)
    at = 1e-4 if FT == 1 else 3e-3      # FT > 1: the step-1 map feature comes from two different fp32 CNN evaluations
    assert_close(dz, z.grad, 2e-3, 1e-6 + at * float(z.grad.abs().max()), 'dz')
    assert_close(dpf, pf.grad, 2e-3, 1e-6 + at * float(pf.grad.abs().max()), 'd past_feat')
    assert_close(dmf, mf.grad, 2e-3, 1e-6 + at * float(mf.grad.abs().max()), 'd map_feat')
    _check_flat(dg, _flat_grads(sdg, 'decoder_net'), sdg, 'decoder_net', what='rollout')
    if FT > 1:
        _check_flat(dr, _flat_grads(sdg, 'decoder_memory'), sdg, 'decoder_memory', what='rollout')
        want_c = torch.cat([_flat_grads(sdg, 'map_conv'), _flat_grads(sdg, 'map_feature')])
        off = 0
        for k, v in sdg.items():
            if k.startswith('map_conv.') or k.startswith('map_feature.'):
                n = v.numel()
                w = want_c[off:off + n]
                assert_close(dc[off:off + n], w, 5e-3, 1e-6 + 5e-4 * float(w.abs().max()), 'rollout grad ' + k)
                off += n
        assert off == nc
    else:
        assert float(dr.abs().max()) == 0.0 and float(dc.abs().max()) == 0.0


def test_rollout_train_backward_single_step(emu, sd):
    """FT = 1: decoder_net weight gradients + the adjoints of past_feat / map_feat / z."""
    _rollout_train(emu, sd, [3, 1, 4], 1)


def test_rollout_train_backward_two_steps(emu, sd):
    """FT = 2: adds the GRU memory's weight gradients and one map-CNN backward (crop at the detached step-0 pose)."""
    _rollout_train(emu, sd, [2], 2)


@pytest.mark.parametrize('single', [False, True])
def test_avoid_coll_loss_fused(emu, sd, single, monkeypatch):
    """strive_avoid_coll_fwd/bwd (AvoidCollLoss as one call per direction) against the oracle's term-by-term loss: objective,
    d/d trajectory and d/d latent; all agents with a (NA,D) latent, and the solution loop's form (environment term and
    latents for the first agent of each scene only, latent (B,1,D): the init term divides by B*D there)."""
    from strive_amd import ops
    from strive_amd.losses.adv_gen_nusc import AvoidCollLoss
    monkeypatch.setattr(ops, '_lib_for', lambda *tensors: emu)
    monkeypatch.setattr(ops, '_ws_cache', {})                           # fresh scratch buffers for this test ...
    monkeypatch.setattr(ops, '_workspace', poisoned_workspace(ops))    # ... that start as NaN bytes (tests/util.py)
    monkeypatch.setattr(torch, 'empty', nan_empty())                   # and so does everything else allocated uninitialised
    g = golden('g5_losses.npz')
    batch, map_idx, raster, dx = mg.g5_inputs(None, None)
    orc = oracle_model(sd)
    env_p = synth.SyntheticMapEnv(raster, dx)
    env_o = env_p            # the oracle reads nusc_raster / nusc_dx only
    veh_att = orc.get_att_normalizer().unnormalize(batch.lw)
    traj = orc.get_normalizer().unnormalize(torch.from_numpy(g['adv_pred'])).contiguous()
    NA, B = traj.shape[0], batch.ptr.shape[0] - 1
    D = 32
    NZ = B if single else NA
    shape = (NZ, 1, D) if single else (NZ, D)
    z0 = synth.f32(synth.counter_uniform((NZ, D), 'emu/avz', -1.0, 1.0))
    mu = synth.f32(synth.counter_uniform((NZ, D), 'emu/avm', -0.5, 0.5))
    var = synth.f32(synth.counter_uniform((NZ, D), 'emu/avv', 0.3, 2.0))
    init = synth.f32(synth.counter_uniform((NZ, D), 'emu/avi', -1.0, 1.0)).view(shape)
    w = {'coll_veh': 1.5, 'coll_env': 0.7, 'motion_prior': 0.02, 'init_z': 0.3}
    kw = dict(veh_coll_buffer=0.3)
    if single:
        kw.update(single_veh_idx=0, ptr=batch.ptr)
    else:
        kw.update(ptr=batch.ptr)
    mapixes = map_idx[batch.batch]

    def run(loss_fn):
        tr = traj.clone().requires_grad_(True)
        z = z0.clone().view(shape).requires_grad_(True)
        out = loss_fn(tr, z, (mu, var))
        out['loss'].backward()
        return out, tr.grad, z.grad

    oo, otr, oz = run(olosses.AvoidColl(w, veh_att, mapixes, env_o, init, **kw))
    fused = AvoidCollLoss(w, veh_att, mapixes, env_p, init, **kw)
    po, ptr_, pz = run(fused)
    assert fused._fused is not None, 'the fused call was not taken'
    assert len(oo['coll_veh_loss']) > 3 and len(oo['coll_env_loss']) > 3, 'the case must have collisions of both kinds'
    assert_close(po['loss'].detach(), oo['loss'].detach(), 2e-5, 1e-6, 'loss')
    # (the collision point is a float64 mean here and an fp32 torch.sum in the oracle: 1e-4 m apart at map coordinates of
    # 1e3 m, i.e. up to a few 1e-3 of a 0.1 m distance and of its gradient)
    assert_close(ptr_, otr, 5e-3, 1e-6, 'd traj')
    assert_close(pz, oz, 1e-4, 1e-7, 'd z')
    for k in ('coll_veh_loss', 'coll_env_loss', 'motion_prior_loss', 'init_loss'):
        assert_close(po[k].detach(), oo[k].detach(), 1e-4, 1e-5, k)
    # one weight at a time (skipped terms must not contribute), and an empty selection
    for k in w:
        w1 = {q: (w[q] if q == k else 0.0) for q in w}
        o1 = run(olosses.AvoidColl(w1, veh_att, mapixes, env_o, init, **kw))
        p1 = run(AvoidCollLoss(w1, veh_att, mapixes, env_p, init, **kw))
        assert_close(p1[0]['loss'].detach(), o1[0]['loss'].detach(), 2e-5, 1e-6, 'loss ' + k)
        if o1[1] is not None:
            assert_close(p1[1], o1[1], 5e-3, 1e-6, 'd traj ' + k)
        else:
            assert float(p1[1].abs().max()) == 0.0
        if o1[2] is not None:
            assert_close(p1[2], o1[2], 1e-4, 1e-7, 'd z ' + k)
        else:
            assert float(p1[2].abs().max()) == 0.0


@pytest.mark.parametrize('mt,infront,atk,far', [(2, 0.0, None, False), (0, None, 'idx', False), (2, 0.0, None, True)])
def test_adv_gen_loss_fused(emu, sd, mt, infront, atk, far, monkeypatch):
    """strive_adv_gen_fwd/bwd (AdvGenLoss as one call per direction) against the oracle's term-by-term loss: objective, the
    gradients w.r.t. the trajectories, the planner trajectory and the latents, the soft-min argmax; with the behind mask, with
    an attacker selection, and with every attacker behind its target (the batch-wide escape that drops the mask)."""
    from strive_amd import ops
    from strive_amd.losses.adv_gen_nusc import AdvGenLoss
    monkeypatch.setattr(ops, '_lib_for', lambda *tensors: emu)
    monkeypatch.setattr(ops, '_ws_cache', {})                           # fresh scratch buffers for this test ...
    monkeypatch.setattr(ops, '_workspace', poisoned_workspace(ops))    # ... that start as NaN bytes (tests/util.py)
    monkeypatch.setattr(torch, 'empty', nan_empty())                   # and so does everything else allocated uninitialised
    g = golden('g5_losses.npz')
    batch, map_idx, raster, dx = mg.g5_inputs(None, None)
    orc = oracle_model(sd)
    env = synth.SyntheticMapEnv(raster, dx)
    unn = orc.get_normalizer().unnormalize
    veh_att = orc.get_att_normalizer().unnormalize(batch.lw)
    mapixes = map_idx[batch.batch]
    NA, B = batch.past.shape[0], batch.ptr.shape[0] - 1
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    traj = unn(torch.from_numpy(g['adv_pred'])).contiguous()
    tgt = unn(batch.future_gt[ego][:, :, :4]).contiguous()
    if far:
        tgt = tgt.clone()
        tgt[:, :, 0] += 500.0
        tgt[:, :, 2] = 1.0
        tgt[:, :, 3] = 0.0
    D = 32
    z0 = synth.f32(synth.counter_uniform((NA - B, D), 'emu/agz', -1.0, 1.0))
    mu = synth.f32(synth.counter_uniform((NA - B, D), 'emu/agm', -0.5, 0.5))
    var = synth.f32(synth.counter_uniform((NA - B, D), 'emu/agv', 0.3, 2.0))
    init = synth.f32(synth.counter_uniform((NA - B, D), 'emu/agi', -1.0, 1.0))
    aidx = None if atk is None else torch.tensor([1, 2, 1]) + batch.ptr[:-1]
    kw = dict(veh_coll_buffer=0.1, crash_loss_min_time=mt, crash_loss_min_infront=infront)

    def run(loss_fn):
        tr = traj.clone().requires_grad_(True)
        tg = tgt.clone().requires_grad_(True)
        z = z0.clone().requires_grad_(True)
        out = loss_fn(tr, tg, z, (mu, var), return_mins=True, attack_agt_idx=aidx)
        out['loss'].backward()
        return out, tr.grad, tg.grad, z.grad

    oo, otr, otg, oz = run(olosses.AdvGen(mg.ADV_WEIGHTS, veh_att, mapixes, env, init, batch.ptr, **kw))
    fused = AdvGenLoss(mg.ADV_WEIGHTS, veh_att, mapixes, env, init, batch.ptr, **kw)
    po, ptr_, ptg, pz = run(fused)
    assert fused._fused is not None, 'the fused call was not taken'
    assert_close(po['loss'].detach(), oo['loss'].detach(), 2e-5, 1e-5, 'loss')
    assert np.array_equal(np.asarray(po['min_agt']), np.asarray(oo['min_agt'])) and np.array_equal(np.asarray(po['min_t']), np.asarray(oo['min_t']))
    assert_close(ptr_, otr, 5e-3, 1e-5, 'd traj')
    assert_close(ptg, otg if otg is not None else torch.zeros_like(ptg), 5e-3, 1e-5, 'd tgt')
    assert_close(pz, oz, 1e-4, 1e-7, 'd z')
    for k in oo:
        if k in ('min_agt', 'min_t', 'loss'):
            continue
        assert_close(po[k].detach(), oo[k].detach(), 2e-3, 2e-3 if 'env' in k else 2e-4, k)


def _loss_case(sizes, key, window=8.0, M=2):
    """ragged scenes packed tightly (collisions of both kinds), two maps, trajectories = the recorded futures"""
    batch, map_idx, raster, dx = mg.build_inputs(sizes, key, window=window, M=M)
    env = synth.SyntheticMapEnv(raster, dx)
    state_n = Normalizer(*[t.double() for t in __import__('strive_amd.constants', fromlist=['x']).state_norm_tensors()])
    att_n = Normalizer(*[t.double() for t in __import__('strive_amd.constants', fromlist=['x']).att_norm_tensors()])
    traj = state_n.unnormalize(batch.future_gt.double())[:, :, :4].float().contiguous()
    veh_att = att_n.unnormalize(batch.lw.double()).float()
    return batch, map_idx, env, traj, veh_att


def test_fused_losses_ragged_scenes(emu, monkeypatch):
    """The fused loss calls on ragged batches: a one-agent scene (no pairs, no attacker rows), two maps, an attacker selection
    that leaves one scene without candidates (all soft-min weights 0, :134-135) -- against the oracle."""
    from strive_amd import ops
    from strive_amd.losses.adv_gen_nusc import AvoidCollLoss, AdvGenLoss
    monkeypatch.setattr(ops, '_lib_for', lambda *tensors: emu)
    monkeypatch.setattr(ops, '_ws_cache', {})                           # fresh scratch buffers for this test ...
    monkeypatch.setattr(ops, '_workspace', poisoned_workspace(ops))    # ... that start as NaN bytes (tests/util.py)
    monkeypatch.setattr(torch, 'empty', nan_empty())                   # and so does everything else allocated uninitialised
    D = 32
    # AvoidCollLoss, sizes with a singleton scene
    batch, map_idx, env, traj, veh_att = _loss_case([1, 4, 2], 'emu/ragged_a')
    NA = traj.shape[0]
    z0 = synth.f32(synth.counter_uniform((NA, D), 'emu/rz', -1.0, 1.0))
    mu = synth.f32(synth.counter_uniform((NA, D), 'emu/rm', -0.5, 0.5))
    var = synth.f32(synth.counter_uniform((NA, D), 'emu/rv', 0.3, 2.0))
    w = {'coll_veh': 1.5, 'coll_env': 0.7, 'motion_prior': 0.02, 'init_z': 0.3}
    res = []
    for cls, e in ((olosses.AvoidColl, env), (AvoidCollLoss, env)):
        tr = traj.clone().requires_grad_(True)
        z = z0.clone().requires_grad_(True)
        fn = cls(w, veh_att, map_idx[batch.batch], e, z0 * 0.5, veh_coll_buffer=0.3, ptr=batch.ptr)
        out = fn(tr, z, (mu, var))
        out['loss'].backward()
        res.append((out['loss'].detach(), tr.grad, z.grad))
    assert_close(res[1][0], res[0][0], 2e-5, 1e-6, 'avoid loss')
    assert_close(res[1][1], res[0][1], 5e-3, 1e-6, 'avoid d traj')
    assert_close(res[1][2], res[0][2], 1e-4, 1e-7, 'avoid d z')
    # AdvGenLoss: every scene has an attacker row, but the selection masks all of scene 1's candidates
    batch, map_idx, env, traj, veh_att = _loss_case([2, 5, 3], 'emu/ragged_b')
    NA, B = traj.shape[0], 3
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    tgt = traj[ego].clone()
    tgt[:, :, :2] += 1.5
    zne = synth.f32(synth.counter_uniform((NA - B, D), 'emu/rz2', -1.0, 1.0))
    mu = synth.f32(synth.counter_uniform((NA - B, D), 'emu/rm2', -0.5, 0.5))
    var = synth.f32(synth.counter_uniform((NA - B, D), 'emu/rv2', 0.3, 2.0))
    aidx = torch.tensor([int(batch.ptr[0]) + 1, int(batch.ptr[2]) + 2])          # scene 0 and scene 2 only
    res = []
    for cls in (olosses.AdvGen, AdvGenLoss):
        tr = traj.clone().requires_grad_(True)
        tg = tgt.clone().requires_grad_(True)
        z = zne.clone().requires_grad_(True)
        fn = cls(mg.ADV_WEIGHTS, veh_att, map_idx[batch.batch], env, zne * 0.5, batch.ptr, veh_coll_buffer=0.1,
                 crash_loss_min_time=1, crash_loss_min_infront=-0.5)
        out = fn(tr, tg, z, (mu, var), return_mins=True, attack_agt_idx=aidx)
        out['loss'].backward()
        res.append((out['loss'].detach(), tr.grad, tg.grad, z.grad, out['adv_crash_loss'].detach()))
    assert float(res[0][4][1]) == 0.0, 'scene 1 must have no candidate attacker in this case'
    assert_close(res[1][0], res[0][0], 2e-5, 1e-5, 'adv loss')
    assert_close(res[1][4], res[0][4], 1e-4, 1e-5, 'crash per scene')
    assert_close(res[1][1], res[0][1], 5e-3, 1e-5, 'adv d traj')
    assert_close(res[1][2], res[0][2], 5e-3, 1e-5, 'adv d tgt')
    assert_close(res[1][3], res[0][3], 1e-4, 1e-7, 'adv d z')


def _adv_quarantine_case(dead, key='emu/quarantine', sizes=(3, 5, 2, 4), D=32):
    """a ragged batch, the same batch without the scenes in ``dead``, and the row maps between them"""
    batch, map_idx, env, traj, veh_att = _loss_case(list(sizes), key)
    NA, B = traj.shape[0], len(sizes)
    ptr = batch.ptr
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[ptr[:-1]] = True
    scene = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    keep_scene = torch.tensor([b not in dead for b in range(B)])
    keep = keep_scene[scene]
    tgt = traj[ego].clone()
    tgt[:, :, :2] += 1.5
    zne = synth.f32(synth.counter_uniform((NA - B, D), key + '/z', -1.0, 1.0))
    mu = synth.f32(synth.counter_uniform((NA - B, D), key + '/m', -0.5, 0.5))
    var = synth.f32(synth.counter_uniform((NA - B, D), key + '/v', 0.3, 2.0))
    keep_ne = keep[~ego]
    sub_sizes = [n for b, n in enumerate(sizes) if b not in dead]
    sub_ptr = torch.tensor([0] + list(np.cumsum(sub_sizes)), dtype=ptr.dtype)
    full = dict(traj=traj, tgt=tgt, z=zne, mu=mu, var=var, veh_att=veh_att, mapixes=map_idx[batch.batch], ptr=ptr, init=zne * 0.5)
    sub = dict(traj=traj[keep], tgt=tgt[keep_scene], z=zne[keep_ne], mu=mu[keep_ne], var=var[keep_ne], veh_att=veh_att[keep],
               mapixes=map_idx[batch.batch][keep], ptr=sub_ptr, init=(zne * 0.5)[keep_ne])
    return env, full, sub, keep, keep_scene, keep_ne


@pytest.mark.parametrize('dead,infront', [([1], -0.5), ([0, 3], None), ([2], 0.0)])
def test_adv_gen_loss_quarantined_scenes_equal_the_batch_without_them(emu, monkeypatch, dead, infront):
    """StriveAdvGen.scene_alive (the closed loop's answer to a planner rollout that fails in ONE scene; the reference's
    batch_size-1 run loses that scene, adv_scenario_gen.py:540-543, and rebuilds batches without the scenes it gives up,
    :323-356): the fused AdvGenLoss with scenes masked out -- their planner trajectory NaN -- gives the other scenes the SAME
    BITS as the loss evaluated on the batch rebuilt without them (every gradient; the objective up to the grouping of its float64
    partial sums), zero gradients to the masked scenes, and agrees with the masked term-by-term path."""
    from strive_amd import ops
    from strive_amd.losses.adv_gen_nusc import AdvGenLoss, TgtMatchingLoss
    monkeypatch.setattr(ops, '_lib_for', lambda *tensors: emu)
    monkeypatch.setattr(ops, '_ws_cache', {})
    monkeypatch.setattr(ops, '_workspace', poisoned_workspace(ops))
    monkeypatch.setattr(torch, 'empty', nan_empty())
    env, full, sub, keep, keep_scene, keep_ne = _adv_quarantine_case(dead)
    kw = dict(veh_coll_buffer=0.1, crash_loss_min_time=1, crash_loss_min_infront=infront)
    ego_full = torch.zeros((full['traj'].shape[0],), dtype=torch.bool)
    ego_full[full['ptr'][:-1]] = True

    grid = {}

    def run(d, alive=None, nan_tgt=False, terms=False):
        fn = AdvGenLoss(mg.ADV_WEIGHTS, d['veh_att'], d['mapixes'], env, d['init'], d['ptr'], **kw)
        # the reference's one batch-wide constant: get_coll_point sizes its sampling grid from the batch MEAN of the vehicle sizes
        # (nuscenes_utils.py:351-354).  A quarantined batch keeps the grid it was built with, so the rebuilt batch is given the same
        if d is full:
            grid.update({3 * d['traj'].shape[1]: fn.env_coll_loss._grid_size(int((~ego_full).sum()), 3 * d['traj'].shape[1])})
        fn.env_coll_loss._grid = dict(grid)
        tr = d['traj'].clone().requires_grad_(True)
        tgv = d['tgt'].clone()
        if nan_tgt:
            tgv[~keep_scene] = float('nan')
        tg = tgv.requires_grad_(True)
        z = d['z'].clone().requires_grad_(True)
        call = fn.forward_terms if terms else fn
        out = call(tr, tg, z, (d['mu'], d['var']), return_mins=True, **({} if alive is None else {'scene_alive': alive}))
        if not terms:
            assert fn._fused is not None, 'the fused call was not taken'
        out['loss'].backward()
        return out, tr.grad, tg.grad, z.grad
    alive = keep_scene.to(torch.uint8)
    of, trf, tgf, zf = run(full, alive, nan_tgt=True)
    os_, trs, tgs, zs = run(sub)
    assert torch.isfinite(of['loss']) and torch.isfinite(trf).all() and torch.isfinite(tgf).all() and torch.isfinite(zf).all()
    assert_close(of['loss'].detach(), os_['loss'].detach(), 1e-6, 1e-7, 'loss')
    assert torch.equal(trf[keep], trs) and torch.equal(tgf[keep_scene], tgs) and torch.equal(zf[keep_ne], zs), \
        'the alive scenes must receive the gradients of the batch without the masked ones, bit for bit'
    assert float(trf[~keep].abs().max()) == 0.0 and float(tgf[~keep_scene].abs().max()) == 0.0 and float(zf[~keep_ne].abs().max()) == 0.0
    alive_b = [b for b in range(len(keep_scene)) if bool(keep_scene[b])]
    assert np.array_equal(np.asarray(of['min_agt'])[alive_b], np.asarray(os_['min_agt']))
    assert np.array_equal(np.asarray(of['min_t'])[alive_b], np.asarray(os_['min_t']))
    # the masked term-by-term path (what the logging entries are computed by)
    ot, trt, tgt_, zt = run(full, alive, nan_tgt=True, terms=True)
    assert_close(ot['loss'].detach(), of['loss'].detach(), 2e-5, 1e-5, 'terms loss')
    assert_close(trt, trf, 5e-3, 1e-5, 'terms d traj')
    assert_close(zt, zf, 1e-4, 1e-7, 'terms d z')
    # everybody alive = no mask
    o1, tr1, tg1, z1 = run(full, torch.ones_like(alive))
    o0, tr0, tg0, z0 = run(full)
    assert torch.equal(o1['loss'], o0['loss']) and torch.equal(tr1, tr0) and torch.equal(tg1, tg0) and torch.equal(z1, z0)
    # the matching loss: same contract
    w = {'match_ext': 10.0, 'motion_prior_ext': 0.001}
    B, T = full['tgt'].shape[0], full['tgt'].shape[1]
    pred = (full['tgt'] + 0.3).clone()
    res = []
    for p_, t_, a_ in ((pred, full['tgt'].clone(), alive), (pred[keep_scene], full['tgt'][keep_scene].clone(), None)):
        if a_ is not None:
            t_[~keep_scene] = float('nan')
        p_ = p_.clone().requires_grad_(True)
        out = TgtMatchingLoss(w)(p_, t_, None, None, **({} if a_ is None else {'scene_alive': a_}))
        out['loss'].backward()
        res.append((out['loss'].detach(), p_.grad))
    assert_close(res[0][0], res[1][0], 1e-6, 1e-7, 'matching loss')
    assert torch.equal(res[0][1][keep_scene], res[1][1]) and float(res[0][1][~keep_scene].abs().max()) == 0.0


def _pack_dense_case(lib, dev, M, K, key):
    """strive_pack_dense against the torch layout code it replaces (params.dense_fragments): transpose and both fragment
    tables, byte for byte."""
    from strive_amd import params, _lib as L
    w = synth.f32(synth.counter_uniform((M, K), key, -0.7, 0.7)).to(dev).contiguous()
    sc = params._pow2_scale(float(w.abs().max()))
    wt = torch.empty((K, M), dtype=torch.float32, device=dev)
    wf = torch.empty((((M + 15) // 16) * ((K + 31) // 32) * 512,), dtype=torch.int32, device=dev)
    wbf = torch.empty((((K + 15) // 16) * ((M + 31) // 32) * 512,), dtype=torch.int32, device=dev)
    lib.call('strive_pack_dense', L.ptr(w), M, K, sc, L.ptr(wt), L.ptr(wf), L.ptr(wbf), L.stream_ptr(w))
    if dev != 'cpu':
        torch.cuda.synchronize()
    assert torch.equal(wt, w.t().contiguous())
    assert torch.equal(wf, params.dense_fragments(w, sc).reshape(-1))
    assert torch.equal(wbf, params.dense_fragments(w.t().contiguous(), sc).reshape(-1))


@pytest.mark.parametrize('M,K', [(128, 128), (64, 128), (128, 164), (192, 64), (32, 40), (2, 128), (128, 4)])
def test_pack_dense_equals_torch_layout(emu, M, K):
    _pack_dense_case(emu, 'cpu', M, K, 'pack/%d/%d' % (M, K))


def _pack_conv_case(lib, dev):
    """the convolution fragment tables as one launch each (strive_pack_split_gather through params._split_gather) against the
    torch layout code (two-piece split + cat + index_select), byte for byte: conv1's layout, a 5x5 and a 3x3 layer."""
    from strive_amd import params, ops
    saved = ops._lib_for

    def packed(fn, w, sc, use_lib):
        def lib_for(*tensors):
            if use_lib:
                return lib
            raise L.StriveHipError('no library: the torch layout code')
        ops._lib_for = lib_for
        try:
            return fn(w, sc)
        finally:
            ops._lib_for = saved
    for fn, shape, key in ((params._conv1_fragments, (16, 4, 7, 7), 'c1'), (params._conv_bf6_fragments, (32, 16, 5, 5), 'c2'),
                           (params._conv_bf6_fragments, (64, 64, 3, 3), 'c4')):
        w = synth.f32(synth.counter_uniform(shape, 'packconv/' + key, -0.3, 0.3)).to(dev).contiguous()
        w.view(-1)[3] = 0.0
        sc = params._pow2_scale(float(w.abs().max()))
        a, b = packed(fn, w, sc, True), packed(fn, w, sc, False)
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), key


def test_pack_conv_fragments_equal_torch_layout(emu):
    _pack_conv_case(emu, 'cpu')


# ---- scene-resident rollout kernels (csrc/scene_rollout.h) against the launch-per-phase kernels on the same inputs ----
def _rollout_both_paths(emu, sd, sizes, FT, ext=False, NC=2, monkeypatch=None, fill=0):
    batch, map_idx, raster, dx = mg.build_inputs(sizes, 'emu', NC=NC)
    env = synth.SyntheticMapEnv(raster, dx)
    orc = oracle_model(sd, NC=NC)
    NA = batch.past.shape[0]
    emb = {'map_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/mf', -1, 1)),
           'past_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/pf', -1, 1))}
    z = synth.f32(synth.counter_uniform((NA, 32), 'emu/zz', -1.5, 1.5)).contiguous()
    extf = batch.future_gt[batch.ptr[:-1]][:, :FT, :4].contiguous() if ext else None
    sn, an = orc.get_normalizer(), orc.get_att_normalizer()
    dec = params.pack_decoder(sd, NC, env, 'cpu', sn, an, NUSC_BIKE_PARAMS)
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    tb = emu.query('strive_rollout_tape_bytes', dec.ref(), sc.ref(), FT)
    wb = emu.query('strive_rollout_workspace_bytes', dec.ref(), sc.ref(), FT)
    rw = synth.f32(synth.counter_uniform((NA, FT, 4), 'emu/rw', -1.0, 1.0)).contiguous()
    mi = map_idx[batch.batch].int().contiguous()
    lw, sem = batch.lw.contiguous(), batch.sem.contiguous()
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('STRIVE_SCENE_KERNELS', mode)
        assert emu.query('strive_rollout_scene_resident', dec.ref(), sc.ref()) == int(mode)
        # (`fill`: what the caller-owned tape and workspace hold before the calls -- 0xFF = NaN in every float the kernels might read)
        tape, ws = torch.full((tb,), fill, dtype=torch.uint8), torch.full((wb,), fill, dtype=torch.uint8)
        traj = torch.zeros((NA, FT, 4))
        emu.call('strive_rollout_fwd', dec.ref(), sc.ref(), L.ptr(batch.past[:, -1, :].contiguous()), L.ptr(lw), L.ptr(sem),
                 L.ptr(emb['past_feat'].contiguous()), L.ptr(emb['map_feat'].contiguous()), L.ptr(z), L.ptr(mi), L.ptr(extf), FT,
                 L.ptr(traj), L.ptr(tape), tb, L.ptr(ws), wb, None)
        dz = torch.zeros((NA, 32))
        emu.call('strive_rollout_bwd', dec.ref(), sc.ref(), L.ptr(lw), L.ptr(sem), L.ptr(z), L.ptr(extf), FT,
                 L.ptr(rw), L.ptr(dz), L.ptr(tape), tb, L.ptr(ws), wb, None)
        out[mode] = (traj, dz, tape, ws)
    # cross pairing: the tape is the same layout, so the sweep of one path runs on the tape of the other
    monkeypatch.setenv('STRIVE_SCENE_KERNELS', '1')
    dz_x = torch.zeros((NA, 32))
    ws = torch.full((wb,), fill, dtype=torch.uint8)
    emu.call('strive_rollout_bwd', dec.ref(), sc.ref(), L.ptr(lw), L.ptr(sem), L.ptr(z), L.ptr(extf), FT,
             L.ptr(rw), L.ptr(dz_x), L.ptr(out['0'][2]), tb, L.ptr(ws), wb, None)
    return out, dz_x


def test_forward_node_phases_on_scene_tiles_equal_the_phase_kernels(emu, sd, monkeypatch):
    """Scenes of more than 16 agents (round 6): the forward step's node-level phases on scene_fwd_step_kernel in 16-row tiles
    (grid (B, 1, tiles), modes 1 and 4; the edge rows on gnn_edge_kernel in between) against the launch-per-phase kernels (option
    scene_tiles = 0): trajectories to fp32 rounding, the per-phase reverse sweep on either tape; a scene of 19 agents (two tiles, the
    second with 3 rows), one of 16 (one full tile) and one of 2, teacher-forced ego rows, two steps."""
    sizes, FT = [19, 16, 2], 2
    batch, map_idx, raster, dx = mg.build_inputs(sizes, 'emu', NC=2)
    env = synth.SyntheticMapEnv(raster, dx)
    orc = oracle_model(sd)
    NA = batch.past.shape[0]
    emb = {'map_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/mf', -1, 1)),
           'past_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/pf', -1, 1))}
    z = synth.f32(synth.counter_uniform((NA, 32), 'emu/zz', -1.5, 1.5)).contiguous()
    extf = batch.future_gt[batch.ptr[:-1]][:, :FT, :4].contiguous()
    dec = params.pack_decoder(sd, 2, env, 'cpu', orc.get_normalizer(), orc.get_att_normalizer(), NUSC_BIKE_PARAMS)
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    tb = emu.query('strive_rollout_tape_bytes', dec.ref(), sc.ref(), FT)
    wb = emu.query('strive_rollout_workspace_bytes', dec.ref(), sc.ref(), FT)
    rw = synth.f32(synth.counter_uniform((NA, FT, 4), 'emu/rw', -1.0, 1.0)).contiguous()
    mi = map_idx[batch.batch].int().contiguous()
    lw, sem = batch.lw.contiguous(), batch.sem.contiguous()
    out = {}
    for tiles in ('0', '1'):
        monkeypatch.setenv('STRIVE_SCENE_TILES', tiles)
        assert emu.query('strive_rollout_scene_resident', dec.ref(), sc.ref()) == (2 if tiles == '1' else 0)
        tape, ws = torch.full((tb,), 0xFF, dtype=torch.uint8), torch.full((wb,), 0xFF, dtype=torch.uint8)
        traj = torch.zeros((NA, FT, 4))
        emu.call('strive_rollout_fwd', dec.ref(), sc.ref(), L.ptr(batch.past[:, -1, :].contiguous()), L.ptr(lw), L.ptr(sem),
                 L.ptr(emb['past_feat'].contiguous()), L.ptr(emb['map_feat'].contiguous()), L.ptr(z), L.ptr(mi), L.ptr(extf), FT,
                 L.ptr(traj), L.ptr(tape), tb, L.ptr(ws), wb, None)
        dz = torch.full((NA, 32), float('nan'))
        w2 = torch.full((wb,), 0xFF, dtype=torch.uint8)
        emu.call('strive_rollout_bwd', dec.ref(), sc.ref(), L.ptr(lw), L.ptr(sem), L.ptr(z), L.ptr(extf), FT,
                 L.ptr(rw), L.ptr(dz), L.ptr(tape), tb, L.ptr(w2), wb, None)
        out[tiles] = (traj, dz)
    (t0, d0), (t1, d1) = out['0'], out['1']
    assert torch.isfinite(t1).all() and torch.isfinite(d1).all()
    assert_close(t1, t0, 2e-5, 2e-6, 'forward node phases on scene tiles')
    scale = float(d0.abs().max())
    assert_close(d1, d0, 1e-3, 2e-5 * scale, 'per-phase sweep on the tape the scene tiles wrote')


@pytest.mark.parametrize('sizes,FT,ext', [([3, 1, 5, 2], 1, False), ([16, 9], 1, True), ([4, 2], 3, True)])
def test_scene_resident_rollout_equals_phase_kernels(emu, sd, sizes, FT, ext, monkeypatch):
    """One workgroup per scene (a step = one launch, the reverse sweep = one launch) against the launch-per-phase kernels:
    same trajectories and latent gradients to fp32 rounding; scenes of 1, 2 and 16 agents (no edges / one edge row / four
    chunks of edge rows), teacher-forced ego rows, several steps (GRU + CNN between the steps)."""
    out, dz_x = _rollout_both_paths(emu, sd, sizes, FT, ext=ext, monkeypatch=monkeypatch)
    (t0, d0, _, _), (t1, d1, _, _) = out['0'], out['1']
    assert_close(t1, t0, 2e-5, 2e-6, 'scene-resident forward')
    scale = float(d0.abs().max())
    assert_close(d1, d0, 1e-3, 2e-5 * scale, 'scene-resident backward')
    assert_close(dz_x, d0, 1e-3, 2e-5 * scale, 'scene-resident sweep on the phase kernels\' tape')


@pytest.mark.parametrize('sizes,FT,ext', [([16, 9], 2, True), ([13, 1, 2], 2, False)])
def test_stepwise_sweep_equals_the_one_launch_sweep(emu, sd, sizes, FT, ext, monkeypatch):
    """The reverse sweep of the scene-resident path as ONE launch per step with K workgroups per scene sharing the scene's edge
    chunks (scene_bwd_sweep_kernel<.., true>; the default from 12 agents per scene on) against the one-launch sweep, on the same
    tape: the partial sums are added in the one-launch order, so the latent gradients are the same BITS whenever every workgroup
    walks at most one chunk (K >= the scene's chunks) and equal to fp32 rounding otherwise; the workspace arrives as NaN bytes
    (nothing may be read before a launch has written it; the emulator runs a launch's workgroups one after the other, which is
    how it found the partial sums of step t + 1 being overwritten by a faster workgroup's sums of step t: two buffers by parity)."""
    batch, map_idx, raster, dx = mg.build_inputs(sizes, 'emu', NC=2)
    env = synth.SyntheticMapEnv(raster, dx)
    orc = oracle_model(sd)
    NA = batch.past.shape[0]
    emb = {'map_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/mf', -1, 1)),
           'past_feat': synth.f32(synth.counter_uniform((NA, 64), 'emu/pf', -1, 1))}
    z = synth.f32(synth.counter_uniform((NA, 32), 'emu/zz', -1.5, 1.5)).contiguous()
    extf = batch.future_gt[batch.ptr[:-1]][:, :FT, :4].contiguous() if ext else None
    dec = params.pack_decoder(sd, 2, env, 'cpu', orc.get_normalizer(), orc.get_att_normalizer(), NUSC_BIKE_PARAMS)
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    tb = emu.query('strive_rollout_tape_bytes', dec.ref(), sc.ref(), FT)
    wb = emu.query('strive_rollout_workspace_bytes', dec.ref(), sc.ref(), FT)
    rw = synth.f32(synth.counter_uniform((NA, FT, 4), 'emu/rw', -1.0, 1.0)).contiguous()
    mi = map_idx[batch.batch].int().contiguous()
    lw, sem = batch.lw.contiguous(), batch.sem.contiguous()
    monkeypatch.setenv('STRIVE_SCENE_KERNELS', '1')
    tape, ws = torch.full((tb,), 0xFF, dtype=torch.uint8), torch.full((wb,), 0xFF, dtype=torch.uint8)
    traj = torch.zeros((NA, FT, 4))
    emu.call('strive_rollout_fwd', dec.ref(), sc.ref(), L.ptr(batch.past[:, -1, :].contiguous()), L.ptr(lw), L.ptr(sem),
             L.ptr(emb['past_feat'].contiguous()), L.ptr(emb['map_feat'].contiguous()), L.ptr(z), L.ptr(mi), L.ptr(extf), FT,
             L.ptr(traj), L.ptr(tape), tb, L.ptr(ws), wb, None)

    def sweep(K):
        if K is None:
            monkeypatch.delenv('STRIVE_SWEEP_STEP', raising=False)
        else:
            monkeypatch.setenv('STRIVE_SWEEP_STEP', str(K))
        dz = torch.full((NA, 32), float('nan'))
        w2 = torch.full((wb,), 0xFF, dtype=torch.uint8)
        emu.call('strive_rollout_bwd', dec.ref(), sc.ref(), L.ptr(lw), L.ptr(sem), L.ptr(z), L.ptr(extf), FT,
                 L.ptr(rw), L.ptr(dz), L.ptr(tape), tb, L.ptr(w2), wb, None)
        return dz
    d_ref = sweep(0)
    assert torch.isfinite(d_ref).all() and float(d_ref.abs().max()) > 0
    scale = float(d_ref.abs().max())
    chunks = (max(sizes) * (max(sizes) - 1) + 63) // 64
    for K in (4, 3, 2, 1):
        d = sweep(K)
        if K >= chunks or K == 1:
            assert torch.equal(d, d_ref), 'stepwise sweep, K = %d: %.3g apart' % (K, float((d - d_ref).abs().max()))
        else:
            assert_close(d, d_ref, 1e-4, 1e-6 * scale, 'stepwise sweep, K = %d' % K)
    assert chunks >= 3 and torch.equal(sweep(None), d_ref), 'the default (stepwise from 3 chunks per scene on) gives the same bits'


@pytest.mark.parametrize('sizes', [[16, 9], [15, 2, 1]])
def test_forward_edge_chunks_on_k_workgroups_equal_the_other_forms(emu, sd, sizes, monkeypatch):
    """Scenes of >= 15 agents, forward step: K workgroups of the scene kernel per scene share the edge chunks and leave partial
    running maxima that the second launch folds (round 5, STRIVE_SCENE_FWD_K) -- against the round-4 split (one workgroup per
    target in gnn_edge_kernel) and the one-launch scene step: trajectories, aggregated messages and arg-max on the tape
    bit-identical to the one-launch step for every K (the fold keeps the chunk loop's tie rule), and the reverse sweep on that tape
    gives the same bits."""
    res = {}
    for name, env in (('one launch', {'STRIVE_SCENE_SPLIT': '0'}), ('K=4', {'STRIVE_SCENE_FWD_K': '4'}), ('K=2', {'STRIVE_SCENE_FWD_K': '2'}),
                      ('K=1', {'STRIVE_SCENE_FWD_K': '1'}), ('default', {}), ('per-target edge kernel', {'STRIVE_SCENE_FWD_K': '0'})):
        for k in ('STRIVE_SCENE_SPLIT', 'STRIVE_SCENE_FWD_K'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out, _ = _rollout_both_paths(emu, sd, sizes, 1, ext=True, monkeypatch=monkeypatch, fill=0xFF)
        res[name] = out['1']
    t0, d0, tape0, _ = res['one launch']
    for name in ('K=4', 'K=2', 'K=1', 'default'):
        t, d, tape, _ = res[name]
        assert torch.equal(t, t0) and torch.equal(d, d0), '%s: forward / backward differ from the one-launch step' % name
        assert torch.equal(tape, tape0), '%s: the tape differs' % name
    t, d, _, _ = res['per-target edge kernel']
    assert_close(t, t0, 2e-5, 2e-6, 'round-4 split forward')


@pytest.mark.parametrize('sizes,FT,ext', [([3, 1, 5, 2], 1, False), ([16, 9], 1, True),
                                          ([4, 2], 2, False)])
def test_rollout_reads_nothing_it_did_not_write(emu, sd, sizes, FT, ext, monkeypatch):
    """The tape and the workspace are caller-owned and arrive uninitialised (torch.empty): with every byte 0xFF (NaN in any float
    a kernel might pick up) both kernel paths must give the bits they give on zeroed buffers -- no row, slot or padding of either
    buffer is read before the call that needs it has written it."""
    zero, dz_zero = _rollout_both_paths(emu, sd, sizes, FT, ext=ext, monkeypatch=monkeypatch, fill=0)
    nan, dz_nan = _rollout_both_paths(emu, sd, sizes, FT, ext=ext, monkeypatch=monkeypatch, fill=0xFF)
    for mode in ('0', '1'):
        assert torch.isfinite(nan[mode][0]).all() and torch.isfinite(nan[mode][1]).all()
        assert torch.equal(zero[mode][0], nan[mode][0]), 'trajectories depend on what the buffers held (STRIVE_SCENE_KERNELS=%s)' % mode
        assert torch.equal(zero[mode][1], nan[mode][1]), 'latent gradients depend on what the buffers held (STRIVE_SCENE_KERNELS=%s)' % mode
    assert torch.equal(dz_zero, dz_nan)


def _cnn_features(emu, args, fr, mi, n):
    wsb = emu.query('strive_map_cnn_workspace_bytes', n)
    ws = torch.zeros(wsb, dtype=torch.uint8)
    feat = torch.zeros((n, 64))
    emu.call('strive_map_cnn_fwd', *args, L.ptr(fr), L.f4([0] * 4), L.f4([1] * 4), L.ptr(mi), n, L.ptr(feat), L.ptr(ws), wsb, None)
    return feat
