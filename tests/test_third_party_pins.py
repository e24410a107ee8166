"""What can be pinned at the two third-party boundaries the reference leaves unpinned:

  * torch-scatter 2.0.7's ``scatter(reduce='max')`` on EXACT ties.  The kernels route the aggregate's gradient to the tied
    message with the LOWEST source index (first-index rule, what torch_scatter's arg-max does); the stand-in used to import
    the reference (torch's ``scatter_reduce('amax')``) splits it evenly among the tied messages.  Forward values are
    identical either way; gradients differ only on exact ties, and their SUM over the tied sources is the same.  This test
    builds such a tie (two identical source agents) and documents the rule.
  * shapely's polygon IoU: the float64 clipping restatement against a committed Monte-Carlo table (10^7 points per pair,
    tests/golden/make_iou_table.py) -- a numerical bound on the "parity unpinned" IoU value.
"""
import os
import sys

import numpy as np
import pytest
import torch

from util import golden, product_model, assert_close
from strive_amd import _lib as L, params, synth
from oracle import model as om
from oracle.geometry import rect_iou

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))


@pytest.fixture(scope='module')
def emu():
    import build as emu_build
    return L.StriveLib(emu_build.build(), require_all=True)


def test_max_aggregation_tie_rule_first_index(emu):
    sd = product_model()[1]
    batch, _ = synth.make_batch([4], key='tie')
    NA = 4
    x = synth.f32(synth.counter_uniform((NA, 164), 'tie/x', -1.0, 1.0))
    pos = batch.past[:, -1, :4].clone().contiguous()
    sem = batch.sem.clone()
    # agents 1 and 2 are indistinguishable as SOURCES: same features, same pose, same class -> identical messages to 0 and 3
    x[2] = x[1]
    pos[2] = pos[1]
    sem[2] = sem[1]
    xg = x.clone().requires_grad_(True)
    y = om.interaction_net(sd, 'decoder_net', xg, pos, sem, batch.edge_index)
    rw = synth.f32(synth.counter_uniform(tuple(y.shape), 'tie/r', -1.0, 1.0))
    rw[1:3] = 0.0                     # look at the targets 0 and 3 only (1 and 2 are targets of each other too)
    (y * rw).sum().backward()
    gp = params.pack_gnn(sd, 'decoder_net', 2)
    sc = params.pack_scenes(batch.ptr, 1, 'cpu')
    wsb = emu.query('strive_gnn_bwd_workspace_bytes', gp.ref(), sc.ref())
    ws = torch.zeros(wsb, dtype=torch.uint8)
    out = torch.zeros((NA, 2))
    wsf = torch.zeros(emu.query('strive_gnn_workspace_bytes', gp.ref(), sc.ref()), dtype=torch.uint8)
    emu.call('strive_gnn_fwd', gp.ref(), sc.ref(), L.ptr(x), L.ptr(pos), L.ptr(sem.contiguous()), L.ptr(out), L.ptr(wsf),
             wsf.numel(), None)
    assert_close(out, y.detach(), 1e-4, 1e-5, 'forward is independent of the tie rule')
    dx = torch.zeros((NA, 164))
    dp = torch.zeros(emu.query('strive_gnn_param_count', gp.ref()))
    emu.call('strive_gnn_bwd', gp.ref(), sc.ref(), L.ptr(x), L.ptr(pos), L.ptr(sem.contiguous()), L.ptr(rw.contiguous()),
             L.ptr(dx), L.ptr(dp), L.ptr(ws), wsb, None)
    want = xg.grad
    # rows 0 and 3 (not part of the tie as sources of each other... they are sources for 3 and 0): unaffected by the rule
    # only through their own message; compare the tied pair by its sum and the rest entry-wise
    assert_close(dx[1] + dx[2], want[1] + want[2], 2e-3, 1e-6 + 2e-4 * float(want.abs().max()), 'sum over the tied sources')
    tied_share_first = float(dx[1].abs().sum()), float(dx[2].abs().sum())
    even_first = float(want[1].abs().sum()), float(want[2].abs().sum())
    # torch's amax splits evenly: both tied rows get the same gradient; the kernels give the tied channels' whole
    # gradient to source 1 (lowest index) -- source 2 keeps only what does not come through tied channels (nothing here)
    assert abs(even_first[0] - even_first[1]) < 1e-6 * max(even_first[0], 1e-12)
    assert tied_share_first[0] > 0.0 and tied_share_first[1] == 0.0, tied_share_first


def test_rect_iou_against_monte_carlo_table():
    g = golden('iou_mc_table.npz')
    n = g['iou_clip'].shape[0]
    assert n == 50 and int(g['npts']) == 10_000_000
    worst = 0.0
    for i in range(n):
        ex = rect_iou(g['box_a'][i], g['lw_a'][i], g['box_b'][i], g['lw_b'][i])
        assert abs(ex - g['iou_clip'][i]) < 1e-12                    # the restatement reproduces its own committed values
        if g['iou_mc_se'][i] > 0:
            worst = max(worst, abs(ex - g['iou_mc'][i]) / g['iou_mc_se'][i])
            assert abs(ex - g['iou_mc'][i]) < 5 * g['iou_mc_se'][i] + 1e-6, (i, ex, g['iou_mc'][i], g['iou_mc_se'][i])
        else:
            assert ex == 0.0 == g['iou_mc'][i]
    # the table spans disjoint, touching and heavily overlapping pairs
    assert (g['iou_clip'] == 0).sum() >= 5 and (g['iou_clip'] > 0.3).sum() >= 5
    print('worst deviation from the Monte-Carlo estimate: %.2f sigma' % worst)


def test_rect_iou_kernel_against_table(emu):
    g = golden('iou_mc_table.npz')
    a, b = synth.f32(g['box_a']), synth.f32(g['box_b'])
    la, lb = synth.f32(g['lw_a']), synth.f32(g['lw_b'])
    out = torch.zeros((a.shape[0],), dtype=torch.float64)
    emu.call('strive_rect_iou', L.ptr(a), L.ptr(la), L.ptr(b), L.ptr(lb), a.shape[0], L.ptr(out), None)
    # inputs are fp32 here (the product's tensors), the table was made from float64 boxes: 1e-6 is the input rounding
    np.testing.assert_allclose(out.numpy(), g['iou_clip'], rtol=0, atol=2e-6)
