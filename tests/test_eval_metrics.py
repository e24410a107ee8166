"""Evaluation metrics of the reference's test loop (the part that needs neither the nuScenes devkit nor
the dataset): compute_disp_err and compute_coll_rate_veh (reference src/losses/traffic_model.py:297-364, 465-545).

Fixture g11 (tests/golden/make_golden.py::g11_eval) holds the reference's own outputs on three synthetic scenes; its shapely
call -- absent here -- was served by an exact rational-arithmetic polygon stand-in, everything around it is the reference's code.
CPU: oracle restatements and the product's host math against the fixture, the product's box test through the emulated kernel.
GPU (-m gpu): the product on the MI355X (ONE strive_rect_iou launch for all pairs x samples x steps).
"""
import os
import sys

import numpy as np
import pytest
import torch

import make_golden as mg
from util import golden, assert_close
from oracle import losses as olosses
from oracle.geometry import Normalizer
from strive_amd import _lib as L, ops
from strive_amd.constants import state_norm_tensors, att_norm_tensors
from strive_amd.datasets.utils import MeanStdNormalizer
from strive_amd.losses.traffic_model import compute_disp_err, compute_coll_rate_veh

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))


@pytest.fixture()
def emu_ops():
    import build as emu_build
    emu = L.StriveLib(emu_build.build(), require_all=True)
    orig = (ops._lib_for, L.get_lib)
    ops._lib_for = lambda *tensors: emu           # CPU tensors + the emulated library: test infrastructure only
    L.get_lib = lambda: emu
    from util import poison_new_workspaces, nan_empty
    orig_ws = poison_new_workspaces(ops)          # new scratch buffers start as NaN bytes, not as whatever torch.empty holds
    orig_empty, torch.empty = torch.empty, nan_empty()      # ... and so does every output allocated with torch.empty
    yield emu
    torch.empty = orig_empty
    ops._workspace = orig_ws
    ops._lib_for, L.get_lib = orig


def _check_disp(got, g, tag):
    for k in ('pos_minADE', 'pos_minFDE', 'ang_minADE', 'ang_minFDE', 'APD'):
        assert_close(got[k], g['%s/%s' % (tag, k)], 1e-5, 2e-5, '%s %s' % (tag, k))


def test_oracle_eval_metrics_match_reference():
    g = golden('g11_eval.npz')
    batch, pred = mg.g11_inputs()
    nrm, att = Normalizer(*state_norm_tensors()), Normalizer(*att_norm_tensors())
    clean = torch.nan_to_num(pred, nan=0.0)
    _check_disp(olosses.compute_disp_err(batch.future_gt, batch.ptr, clean, nrm), g, 'disp')
    _check_disp(olosses.compute_disp_err(batch.future_gt, batch.ptr, clean[:, :, :6], nrm), g, 'disp6')
    did, count = olosses.compute_coll_rate_veh(batch.edge_index, batch.lw, pred, nrm, att)
    assert np.array_equal(did, g['veh/did_collide']) and [float(count), float(did.size)] == g['veh/num'].tolist()
    assert did.sum() >= 2


def test_product_eval_metrics_emulated(emu_ops):
    g = golden('g11_eval.npz')
    batch, pred = mg.g11_inputs()
    nrm, att = MeanStdNormalizer(*state_norm_tensors()), MeanStdNormalizer(*att_norm_tensors())
    clean = torch.nan_to_num(pred, nan=0.0)
    _check_disp(compute_disp_err(batch, {'future_pred': clean}, nrm), g, 'disp')
    _check_disp(compute_disp_err(batch, {'future_pred': clean[:, :, :6]}, nrm), g, 'disp6')
    cv = compute_coll_rate_veh(batch, {'future_pred': pred.clone()}, nrm, att)
    assert np.array_equal(cv['did_collide'], g['veh/did_collide'])
    assert [cv['num_coll_veh'], cv['num_traj_veh']] == g['veh/num'].tolist()
    assert np.array_equal(compute_coll_rate_veh(batch, pred.clone(), nrm, att)['did_collide'], g['veh/did_collide'])   # tensor form


@pytest.mark.gpu
def test_product_eval_metrics_gpu():
    dev = 'cuda:0'
    g = golden('g11_eval.npz')
    batch, pred = mg.g11_inputs()
    bg = batch.clone().to(dev)
    nrm, att = MeanStdNormalizer(*state_norm_tensors()), MeanStdNormalizer(*att_norm_tensors())
    clean = torch.nan_to_num(pred, nan=0.0).to(dev)
    _check_disp(compute_disp_err(bg, {'future_pred': clean}, nrm), g, 'disp')
    _check_disp(compute_disp_err(bg, {'future_pred': clean[:, :, :6].contiguous()}, nrm), g, 'disp6')
    cv = compute_coll_rate_veh(bg, {'future_pred': pred.to(dev)}, nrm, att)
    assert np.array_equal(cv['did_collide'], g['veh/did_collide'])
    assert [cv['num_coll_veh'], cv['num_traj_veh']] == g['veh/num'].tolist()
    # a larger batch against the oracle: 48 agents x 6 samples x 12 steps
    big, _, _, _ = mg.build_inputs([16, 16, 16], 'g11/big')
    NA = big.past.shape[0]
    p = big.future_gt[:, :12, :4].unsqueeze(1).expand(NA, 6, 12, 4).clone()
    from strive_amd import synth
    p[..., :2] += synth.f32(synth.counter_uniform((NA, 6, 1, 2), 'g11/big/off', -0.25, 0.25))
    cpu_n, cpu_a = Normalizer(*state_norm_tensors()), Normalizer(*att_norm_tensors())
    want, count = olosses.compute_coll_rate_veh(big.edge_index, big.lw, p, cpu_n, cpu_a)
    got = compute_coll_rate_veh(big.clone().to(dev), p.to(dev), nrm, att)
    assert np.array_equal(got['did_collide'], want) and got['num_coll_veh'] == float(count) and want.sum() > 5
