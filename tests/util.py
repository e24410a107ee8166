"""Shared builders for the test-suite (inputs regenerated from strive_amd.synth's counter generator)."""
import os

import numpy as np
import torch

from strive_amd import synth
from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def product_model(NC=2, FT=12, device='cpu', key='weights'):
    from strive_amd.models.traffic_model import TrafficModel
    from strive_amd.datasets.utils import MeanStdNormalizer
    m = TrafficModel(4, FT, 256, NC)
    sd = synth.fill_state_dict(m.state_dict(), key=key)
    m.load_state_dict(sd)
    m.set_normalizer(MeanStdNormalizer(*state_norm_tensors()))
    m.set_att_normalizer(MeanStdNormalizer(*att_norm_tensors()))
    m.set_bicycle_params(NUSC_BIKE_PARAMS)
    m.eval()
    return m.to(device), sd


def oracle_model(sd, NC=2, FT=12):
    from oracle.model import OracleTrafficModel
    from oracle.geometry import Normalizer
    return OracleTrafficModel(sd, Normalizer(*state_norm_tensors()), Normalizer(*att_norm_tensors()), NUSC_BIKE_PARAMS,
                              FT=FT, NC=NC)


def assert_close(a, b, rtol, atol, what=''):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError('%s: %d/%d entries off; worst at %s: got %.8g want %.8g (|d|=%.3g)' %
                             (what, bad.sum(), bad.size, i, a[i], b[i], err[i]))


def assert_close_frac(a, b, rtol, atol, frac, hard_atol, what=''):
    """Gradients through max-aggregation / LayerNorm-ReLU kinks: a 1e-6 difference can move an arg-max between two
    near-tied messages and change a handful of gradient entries at O(1) relative level while everything else agrees.
    At least `frac` of the entries must be within (rtol, atol) and every entry within hard_atol."""
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    err = np.abs(a - b)
    ok = float(np.mean(err <= atol + rtol * np.abs(b)))
    assert ok >= frac, '%s: only %.4f of the entries within rtol %.1e / atol %.1e' % (what, ok, rtol, atol)
    assert float(err.max()) <= hard_atol, '%s: worst entry off by %.3g (> %.3g)' % (what, float(err.max()), hard_atol)


def poisoned_workspace(ops):
    """Test infrastructure for the emulated (CPU) library: a replacement for ops._workspace whose NEW scratch buffers start as 0xFF
    bytes (NaN in any float a kernel might pick up) instead of whatever torch.empty returns -- a kernel that reads scratch it has not
    written shows up as NaN / a mismatch against the oracle."""
    orig = ops._workspace

    def _ws(device, nbytes, tag='ws'):
        known = {id(v) for v in ops._ws_cache.values()}
        buf = orig(device, nbytes, tag)
        if id(buf) not in known and buf.device.type == 'cpu':
            buf.fill_(0xFF)
        return buf
    return _ws


def poison_new_workspaces(ops):
    """Install poisoned_workspace(ops); returns the function to restore."""
    orig = ops._workspace
    ops._workspace = poisoned_workspace(ops)
    return orig


def nan_empty():
    """torch.empty that returns NaN floats / 0xFF bytes / a recognisable negative integer instead of whatever the allocator holds
    (test infrastructure, CPU emulation runs): outputs, tapes and scratch the product allocates with torch.empty and a kernel then
    fails to write completely show up as NaN / a mismatch against the oracle."""
    import torch
    orig = torch.empty

    def empty(*a, **k):
        t = orig(*a, **k)
        if t.device.type == 'cpu' and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(float('nan'))
            elif t.dtype == torch.uint8:
                t.fill_(0xFF)
            elif t.dtype in (torch.int32, torch.int64):
                t.fill_(-0x0F0F0F0F)
        return t
    return empty


class poisoned_empty(object):
    """Context manager installing nan_empty() as torch.empty."""

    def __enter__(self):
        import torch
        self.torch, self.orig = torch, torch.empty
        torch.empty = nan_empty()
        return self

    def __exit__(self, *exc):
        self.torch.empty = self.orig
        return False
