"""Direct parity of two building blocks the rollout kernels evaluate in registers -- the kinematic bicycle step (with its
clamps ACTIVE: speed at 0 and at 50 m/s, yaw rate at +-2 pi) and the rigid frame change -- against the reference's own outputs
(fixture G1: TrafficModel.sim_traj and transform2frame, tests/golden/make_golden.py::g1_ops) and, for the adjoints, the
oracle's autograd.  The C-ABI entries strive_bicycle_step / strive_rel_pose call the very device functions the rollout uses
(bike_forward / bike_backward, rel_pose / rel_pose_bwd).  CPU: through the host emulation; -m gpu: on the MI355X.
"""
import os
import sys

import numpy as np
import pytest
import torch

from util import golden, assert_close
from oracle import geometry
from strive_amd import _lib as L, synth
from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))


def _emu():
    import build as emu_build
    return L.StriveLib(emu_build.build(), require_all=True)


def _dyn(identity):
    d = L.StriveDecoder()
    sm, ss = state_norm_tensors()
    am, as_ = att_norm_tensors()
    for i in range(6):
        d.state_mean[i] = 0.0 if identity else float(sm[i])
        d.state_std[i] = 1.0 if identity else float(ss[i])
    for i in range(2):
        d.att_mean[i] = 0.0 if identity else float(am[i])
        d.att_std[i] = 1.0 if identity else float(as_[i])
    d.a_mean, d.a_std = (0.0, 1.0) if identity else NUSC_BIKE_PARAMS['a_stats']
    d.ddh_mean, d.ddh_std = (0.0, 1.0) if identity else NUSC_BIKE_PARAMS['ddh_stats']
    d.dt, d.max_hdot, d.max_s = NUSC_BIKE_PARAMS['dt'], NUSC_BIKE_PARAMS['maxhdot'], NUSC_BIKE_PARAMS['maxs']
    return d


def g1_bicycle_inputs():
    st = synth.f32(synth.counter_uniform((9, 6), 'g1/state', -1.0, 1.0))
    st[:, 0:2] *= 100.0
    st[:, 4] = torch.tensor([0.0, 0.2, 3.0, 49.9, 12.0, 0.01, 7.0, 25.0, 1.0])
    st[:, 5] = torch.tensor([0.0, 6.2, -6.2, 0.1, -0.1, 0.3, 0.0, 1.0, -1.0])
    a = synth.f32(synth.counter_uniform((9,), 'g1/a', -4.0, 4.0))
    a[0], a[3] = -3.0, 4.0
    ddh = synth.f32(synth.counter_uniform((9,), 'g1/ddh', -0.5, 0.5))
    ddh[1], ddh[2] = 0.5, -0.5
    vlen = synth.f32(synth.counter_uniform((9,), 'g1/len', 3.5, 6.0))
    return st, a, ddh, vlen


def bicycle(lib, dev, d, st, dec, lw0, g_out=None):
    N = st.shape[0]
    st, dec, lw0 = st.contiguous().to(dev), dec.contiguous().to(dev), lw0.contiguous().to(dev)
    out = torch.zeros((N, 6), device=dev)
    gs, gd = torch.zeros((N, 6), device=dev), torch.zeros((N, 2), device=dev)
    go = None if g_out is None else g_out.contiguous().to(dev)
    lib.call('strive_bicycle_step', d, L.ptr(st), L.ptr(dec), L.ptr(lw0), L.ptr(go), L.ptr(out), L.ptr(gs), L.ptr(gd), N, None)
    return out.cpu(), gs.cpu(), gd.cpu()


def check_bicycle(lib, dev):
    g = golden('g1_ops.npz')
    st, a, ddh, vlen = g1_bicycle_inputs()
    dec = torch.stack([a, ddh], -1)
    # (i) the reference's own sim_traj on unnormalised states, identity normalisers
    go = synth.f32(synth.counter_uniform((9, 6), 'bb/go', -1.0, 1.0))
    out, gs, gd = bicycle(lib, dev, _dyn(True), st, dec, vlen, go)
    assert_close(out, g['bicycle'], 1e-5, 1e-5, 'bicycle step vs the reference (clamp-active rows included)')
    sv, dv = st.clone().requires_grad_(True), dec.clone().requires_grad_(True)
    ref = geometry.bicycle_step(sv, dv[:, 0], dv[:, 1], vlen, NUSC_BIKE_PARAMS['dt'], NUSC_BIKE_PARAMS['maxhdot'], NUSC_BIKE_PARAMS['maxs'])
    (ref * go).sum().backward()
    assert_close(gs, sv.grad, 1e-4, 1e-5, 'd/d state')
    assert_close(gd, dv.grad, 1e-4, 1e-5, 'd/d (a, ddh)')
    # the clamps are active where the fixture says: no gradient to the acceleration at s = 0 (row 0) and s = 50 (row 3), none to
    # the yaw acceleration at |hdot| = 2 pi (rows 1, 2); and they are inactive elsewhere
    assert float(out[0, 4]) == 0.0 and float(out[3, 4]) == 50.0
    assert abs(float(out[1, 5]) - 2 * np.pi) < 1e-6 and abs(float(out[2, 5]) + 2 * np.pi) < 1e-6
    assert gd[0, 0] == 0 and gd[3, 0] == 0 and gd[1, 1] == 0 and gd[2, 1] == 0
    assert torch.equal(gd == 0, dv.grad == 0) and int((gd != 0).sum()) >= 11        # the same rows saturate, the others do not
    # (ii) the rollout's normalised form (MeanStdNormalizer statistics, a / ddh statistics, vehicle length from lw)
    nrm, att = geometry.Normalizer(*state_norm_tensors()), geometry.Normalizer(*att_norm_tensors())
    st_n = nrm.normalize(st)
    lw_n = att.normalize(torch.stack([vlen, torch.full_like(vlen, 2.0)], -1))
    dec_n = synth.f32(synth.counter_uniform((9, 2), 'bb/dec', -3.0, 3.0))
    dec_n[0, 0], dec_n[3, 0] = -4.0, 4.0
    out_n, gs_n, gd_n = bicycle(lib, dev, _dyn(False), st_n, dec_n, lw_n[:, 0].contiguous(), go)
    sv, dv = st_n.clone().requires_grad_(True), dec_n.clone().requires_grad_(True)
    a_u = dv[:, 0] * NUSC_BIKE_PARAMS['a_stats'][1] + NUSC_BIKE_PARAMS['a_stats'][0]
    d_u = dv[:, 1] * NUSC_BIKE_PARAMS['ddh_stats'][1] + NUSC_BIKE_PARAMS['ddh_stats'][0]
    ref = nrm.normalize(geometry.bicycle_step(nrm.unnormalize(sv), a_u, d_u, att.unnormalize(lw_n)[:, 0], NUSC_BIKE_PARAMS['dt'],
                                              NUSC_BIKE_PARAMS['maxhdot'], NUSC_BIKE_PARAMS['maxs']))
    (ref * go).sum().backward()
    assert_close(out_n, ref.detach(), 1e-5, 1e-5, 'normalised bicycle step')
    assert_close(gs_n, sv.grad, 1e-4, 1e-5, 'normalised d/d state')
    assert_close(gd_n, dv.grad, 1e-4, 1e-6, 'normalised d/d decoder output')
    assert gd_n[0, 0] == 0 and gd_n[3, 0] == 0


def check_rel_pose(lib, dev):
    g = golden('g1_ops.npz')
    frame = synth.f32(synth.counter_uniform((7, 4), 'g1/frame', -2.0, 2.0))
    poses = synth.f32(synth.counter_uniform((7, 5, 4), 'g1/poses', -3.0, 3.0))
    go = synth.f32(synth.counter_uniform((7, 5, 4), 'bb/rgo', -1.0, 1.0))
    out = torch.zeros((7, 5, 4), device=dev)
    gf, gp = torch.zeros((7, 4), device=dev), torch.zeros((7, 5, 4), device=dev)
    lib.call('strive_rel_pose', L.ptr(frame.to(dev)), L.ptr(poses.to(dev)), L.ptr(go.to(dev)), L.ptr(out), L.ptr(gf), L.ptr(gp), 7, 5, None)
    assert_close(out.cpu(), g['t2f_fwd'], 1e-5, 1e-5, 'rel_pose vs the reference transform2frame')
    fv, pv = frame.clone().requires_grad_(True), poses.clone().requires_grad_(True)
    (geometry.transform2frame(fv, pv) * go).sum().backward()
    assert_close(gf.cpu(), fv.grad, 1e-4, 1e-5, 'd/d frame')
    assert_close(gp.cpu(), pv.grad, 1e-4, 1e-5, 'd/d poses')


def test_bicycle_step_emulated():
    check_bicycle(_emu(), 'cpu')


def test_rel_pose_emulated():
    check_rel_pose(_emu(), 'cpu')


@pytest.mark.gpu
def test_bicycle_step_and_rel_pose_gpu():
    lib = L.get_lib()
    check_bicycle(lib, 'cuda:0')
    check_rel_pose(lib, 'cuda:0')
