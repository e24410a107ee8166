"""What can be pinned at the two third-party boundaries the reference leaves unpinned:

  * torch-scatter 2.0.7's ``scatter(reduce='max')`` on EXACT ties.  The kernels route the aggregate's gradient to the tied
    message with the LOWEST source index (first-index rule, what torch_scatter's arg-max does); the stand-in used to import
    the reference (torch's ``scatter_reduce('amax')``) splits it evenly among the tied messages.  Forward values are
    identical either way; gradients differ only on exact ties, and their SUM over the tied sources is the same.  This test
    builds such a tie (two identical source agents) and documents the rule.
  * shapely's polygon IoU: the float64 clipping restatement (oracle) and the HIP kernel against (i) a committed Monte-Carlo
    table (10^7 points per pair, tests/golden/make_iou_table.py) and (ii) an EXACT table: 10,000 box pairs whose IoU was
    computed in rational arithmetic by a different algorithm from get_corners' own float64 corners
    (tests/golden/make_iou_exact.py: random, near-parallel, identical, nested, edge- / corner-touching, 2^-10 m overlaps,
    axis-aligned closed forms, far apart).  shapely / GEOS evaluate the same geometric quantity in float64; what remains
    unpinned is their rounding (~1e-15), not the geometry.

The tie rule and both tables also run on the MI355X (-m gpu).
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


# ------------------------------------------------------------------------------------------------
# exact table (rational arithmetic, independent algorithm)
# ------------------------------------------------------------------------------------------------

def _exact_table():
    g = golden('iou_exact.npz')
    names = [str(n) for n in g['names']]
    cat = lambda k: np.concatenate([g[n + '/' + k] for n in names])
    return names, g, cat('a'), cat('la'), cat('b'), cat('lb'), cat('iou')


def _exact_tol(names, g):
    """float64 clipping against exact arithmetic: 1e-12, except the near-parallel pairs placed up to 2 km from the origin,
    where the crossing of two almost parallel edges is conditioned like 1 / sin(angle) x the coordinate's ulp (measured
    3.6e-10; shapely's float64 clipping has the same conditioning) -- still seven orders below the 0.02 threshold's margin"""
    return np.concatenate([np.full(g[n + '/iou'].shape, 2e-9 if n == 'near_parallel' else 1e-12) for n in names])


def test_exact_table_covers_the_hard_cases():
    names, g, a, la, b, lb, iou = _exact_table()
    assert iou.shape[0] >= 10000
    assert np.all(g['identical/iou'] == 1.0) and np.all(g['far/iou'] == 0.0)
    assert (g['touching/iou'] == 0).sum() >= 300 and ((g['touching/iou'] > 0) & (g['touching/iou'] < 1e-3)).sum() >= 100
    thr = 0.02
    assert (np.abs(iou - thr) < 5e-3).sum() >= 50            # pairs close to the collision threshold exist


def test_oracle_rect_iou_against_exact_table():
    names, g, a, la, b, lb, iou = _exact_table()
    tol = _exact_tol(names, g)
    worst = 0.0
    for i in range(0, iou.shape[0], 3):                       # every third pair: keeps the CPU suite short; the GPU test runs all
        got = rect_iou(a[i], la[i], b[i], lb[i])
        worst = max(worst, abs(got - iou[i]))
        assert abs(got - iou[i]) < tol[i], (i, got, iou[i])
    print('oracle rect_iou vs exact rational IoU: worst |diff| %.2e' % worst)


def test_rect_iou_kernel_emulated_against_exact_table(emu):
    names, g, a, la, b, lb, iou = _exact_table()
    sel = np.arange(0, iou.shape[0], 2)
    ta, tla, tb, tlb = (torch.from_numpy(np.ascontiguousarray(v[sel])) for v in (a, la, b, lb))
    out = torch.zeros((sel.shape[0],), dtype=torch.float64)
    emu.call('strive_rect_iou', L.ptr(ta), L.ptr(tla), L.ptr(tb), L.ptr(tlb), sel.shape[0], L.ptr(out), None)
    d = np.abs(out.numpy() - iou[sel])
    print('emulated kernel vs exact rational IoU: worst |diff| %.2e' % d.max())
    assert np.all(d < _exact_tol(names, g)[sel])
    assert np.array_equal(out.numpy() > 0.02, iou[sel] > 0.02)


@pytest.mark.gpu
def test_rect_iou_kernel_gpu_against_exact_and_monte_carlo_tables():
    from strive_amd import ops
    dev = 'cuda:0'
    names, g, a, la, b, lb, iou = _exact_table()
    ta, tla, tb, tlb = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (a, la, b, lb))
    got = ops.rect_iou(ta, tla, tb, tlb).cpu().numpy()
    d = np.abs(got - iou)
    print('strive_rect_iou (MI355X) vs exact rational IoU on %d pairs: worst |diff| %.2e' % (iou.shape[0], d.max()))
    assert np.all(d < _exact_tol(names, g))
    assert np.array_equal(got > 0.02, iou > 0.02)                  # the reference's collision decision (VEH_COLL_THRESH)
    nan = ta.clone()
    nan[::7, 1] = float('nan')
    assert bool(torch.isnan(ops.rect_iou(nan, tla, tb, tlb)[::7]).all())
    m = golden('iou_mc_table.npz')
    out = ops.rect_iou(synth.f32(m['box_a']).to(dev), synth.f32(m['lw_a']).to(dev), synth.f32(m['box_b']).to(dev),
                       synth.f32(m['lw_b']).to(dev)).cpu().numpy()
    np.testing.assert_allclose(out, m['iou_clip'], rtol=0, atol=2e-6)
    ok = m['iou_mc_se'] > 0
    assert np.all(np.abs(out[ok] - m['iou_mc'][ok]) < 5 * m['iou_mc_se'][ok] + 2e-6)


@pytest.mark.gpu
def test_max_aggregation_tie_rule_first_index_gpu():
    """the same exact tie as above through the product's ops on the MI355X: forward unaffected, the tied channels' gradient
    goes to the source with the lowest index, the sum over the tied sources equals autograd's"""
    from strive_amd import ops
    from util import product_model
    dev = 'cuda:0'
    m, sd = product_model(device=dev)
    batch, _ = synth.make_batch([4], key='tie')
    NA = 4
    x = synth.f32(synth.counter_uniform((NA, 164), 'tie/x', -1.0, 1.0))
    pos = batch.past[:, -1, :4].clone().contiguous()
    sem = batch.sem.clone()
    x[2] = x[1]
    pos[2] = pos[1]
    sem[2] = sem[1]
    xg = x.clone().requires_grad_(True)
    y = om.interaction_net(sd, 'decoder_net', xg, pos, sem, batch.edge_index)
    rw = synth.f32(synth.counter_uniform(tuple(y.shape), 'tie/r', -1.0, 1.0))
    rw[1:3] = 0.0
    (y * rw).sum().backward()
    lib = L.get_lib()
    gp = params.pack_gnn({k: v.to(dev) for k, v in sd.items()}, 'decoder_net', 2)
    sc = params.pack_scenes(batch.ptr, 1, dev)
    wsb = lib.query('strive_gnn_bwd_workspace_bytes', gp.ref(), sc.ref())
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    wsf = torch.zeros(lib.query('strive_gnn_workspace_bytes', gp.ref(), sc.ref()), dtype=torch.uint8, device=dev)
    xd, pd, sd_, rd = x.to(dev), pos.to(dev), sem.contiguous().to(dev), rw.contiguous().to(dev)
    out = torch.zeros((NA, 2), device=dev)
    lib.call('strive_gnn_fwd', gp.ref(), sc.ref(), L.ptr(xd), L.ptr(pd), L.ptr(sd_), L.ptr(out), L.ptr(wsf), wsf.numel(), None)
    assert_close(out, y.detach(), 1e-4, 1e-5, 'forward is independent of the tie rule')
    dx = torch.zeros((NA, 164), device=dev)
    dp = torch.zeros(lib.query('strive_gnn_param_count', gp.ref()), device=dev)
    lib.call('strive_gnn_bwd', gp.ref(), sc.ref(), L.ptr(xd), L.ptr(pd), L.ptr(sd_), L.ptr(rd), L.ptr(dx), L.ptr(dp), L.ptr(ws), wsb,
             None)
    torch.cuda.synchronize()
    dx = dx.cpu()
    want = xg.grad
    assert_close(dx[1] + dx[2], want[1] + want[2], 2e-3, 1e-6 + 2e-4 * float(want.abs().max()), 'sum over the tied sources')
    assert float(dx[1].abs().sum()) > 0.0 and float(dx[2].abs().sum()) == 0.0
