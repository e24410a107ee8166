"""Deterministic synthetic scenes, rasters and weights for tests and bench.py.

Nothing here comes from nuScenes (dataset + devkit are unavailable).  All
values are produced by a counter-based generator (splitmix64 over a flat index) so that this
container, the CPU oracle and the MI355X box build bit-identical inputs without relying on any
RNG state.

Scene layout follows what the reference's dataset emits (reference
src/datasets/nuscenes_dataset.py:678-702): per scene a fully connected directed graph without
self loops, ego at node 0, tensors ``past (n,PT,6)``, ``future (n,FT,6)``, ``lw (n,2)``,
``sem (n,NC)``, visibility flags, everything NORMALISED with the nuScenes statistics
(reference src/datasets/utils.py:130-193).
"""
import math

import numpy as np
import torch

from .graph import Data, Batch, clique_edge_index
from .constants import NUSC_NORM_STATS_CAR_TRUCK, state_norm_tensors, att_norm_tensors

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def _key_hash(key):
    h = np.uint64(0xCBF29CE484222325)
    with np.errstate(over='ignore'):
        for ch in str(key).encode():
            h = ((h ^ np.uint64(ch)) * np.uint64(0x100000001B3)) & _MASK
    return h


def counter_uniform(shape, key, lo=0.0, hi=1.0):
    """float64 numpy array of U[lo,hi) values; element i depends only on (key, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx ^ _splitmix64(_key_hash(key)))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).reshape(shape)


def counter_normal(shape, key):
    """float64 standard normal via Box-Muller on two counter streams."""
    u1 = counter_uniform(shape, str(key) + '/bm1')
    u2 = counter_uniform(shape, str(key) + '/bm2')
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


# --------------------------------------------------------------------------------------------
# raster
# --------------------------------------------------------------------------------------------

def make_raster(H=1024, W=1024, C=4, M=1, dx=0.25, key='raster'):
    """Procedural uint8 raster ``(M,C,H,W)`` + float64 ``dx (M,2)``.

    Layer 0: drivable "road bands" (a grid of wide horizontal and vertical roads, roughly 70 % ones),
    layers 1..C-1: sparse divider lines.  Values are 0/1 like the reference's binarised masks
    (reference src/datasets/map_env.py:110-124).  ``dx`` gets slightly non-round float64 values for
    M>1 so the fp64 divide matters.
    """
    rast = np.zeros((M, C, H, W), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    for m in range(M):
        period = 160 + 24 * m
        road_w = 72 + 8 * m
        off = 37 * m
        road = (((yy + off) % period) < road_w) | (((xx + 2 * off) % period) < road_w)
        rast[m, 0] = road.astype(np.uint8)
        for c in range(1, C):
            p = period // 2 + 7 * c
            line = (((yy + 3 * c + off) % p) < 2) | (((xx + 5 * c) % (p + 11)) < 2)
            rast[m, c] = (line & road).astype(np.uint8)
    dxs = np.zeros((M, 2), dtype=np.float64)
    for m in range(M):
        dxs[m, 0] = dx * (1.0 + (0.00013 * m))
        dxs[m, 1] = dx * (1.0 - (0.00007 * m))
    return torch.from_numpy(rast), torch.from_numpy(dxs)


class SyntheticMapEnv(object):
    """Duck-typed stand-in for the reference's NuScenesMapEnv (src/datasets/map_env.py:22-61):
    carries ``nusc_raster``, ``nusc_dx``, ``bounds``, ``L``, ``W``, ``map_list``.  It deliberately has
    no ``get_map_crop``: the product's TrafficModel reads the raster tensors directly through the HIP
    crop kernel, and the oracle uses its own restatement."""

    def __init__(self, raster, dx, bounds=(-17.0, -38.5, 60.0, 38.5), L=256, W=256):
        self.nusc_raster = raster
        self.nusc_dx = dx
        self.bounds = list(bounds)
        self.L = L
        self.W = W
        self.map_list = ['synthetic-%d' % i for i in range(raster.shape[0])]
        self.num_layers = raster.shape[1]

    def to(self, device):
        self.nusc_raster = self.nusc_raster.to(device)
        self.nusc_dx = self.nusc_dx.to(device)
        return self


# --------------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------------

def make_scene(n, key, PT=4, FT=12, NC=2, map_extent=(256.0, 256.0), dt=0.5, window=120.0,
               with_future=True):
    """One scene of ``n`` agents as a :class:`Data` (NORMALISED), ego at node 0."""
    Hm, Wm = map_extent
    cx = counter_uniform((), key + '/cx', 0.35 * Wm, 0.65 * Wm)
    cy = counter_uniform((), key + '/cy', 0.35 * Hm, 0.65 * Hm)
    px = cx + counter_uniform((n,), key + '/px', -0.5 * window, 0.5 * window)
    py = cy + counter_uniform((n,), key + '/py', -0.5 * window, 0.5 * window)
    px = np.clip(px, 20.0, Wm - 20.0)
    py = np.clip(py, 20.0, Hm - 20.0)
    h = counter_uniform((n,), key + '/h', -math.pi, math.pi)
    s = counter_uniform((n,), key + '/s', 0.0, 5.0)
    hdot = counter_uniform((n,), key + '/hd', -0.05, 0.05)
    T = PT + FT
    ts = (np.arange(T) - (PT - 1)) * dt  # t=0 at last past step
    hh = h[:, None] + hdot[:, None] * ts[None, :]
    # integrate positions for constant speed / yaw-rate motion
    x = np.zeros((n, T))
    y = np.zeros((n, T))
    x[:, PT - 1] = px
    y[:, PT - 1] = py
    for t in range(PT, T):
        x[:, t] = x[:, t - 1] + s * np.cos(hh[:, t]) * dt
        y[:, t] = y[:, t - 1] + s * np.sin(hh[:, t]) * dt
    for t in range(PT - 2, -1, -1):
        x[:, t] = x[:, t + 1] - s * np.cos(hh[:, t + 1]) * dt
        y[:, t] = y[:, t + 1] - s * np.sin(hh[:, t + 1]) * dt
    state = np.stack([x, y, np.cos(hh), np.sin(hh), np.broadcast_to(s[:, None], (n, T)),
                      np.broadcast_to(hdot[:, None], (n, T))], axis=-1)
    lw = np.stack([4.8 + 0.3 * counter_normal((n,), key + '/l'),
                   2.0 + 0.1 * counter_normal((n,), key + '/w')], axis=-1)
    lw = np.clip(lw, [3.0, 1.5], [7.0, 2.6])
    cls = (counter_uniform((n,), key + '/cls') * NC).astype(np.int64) % NC
    sem = np.zeros((n, NC))
    sem[np.arange(n), cls] = 1.0

    smean, sstd = state_norm_tensors()
    amean, astd = att_norm_tensors()
    state_t = (f32(state) - smean) / sstd
    lw_t = (f32(lw) - amean) / astd
    d = Data(
        x=torch.empty((n,)), pos=torch.empty((n,)),
        edge_index=clique_edge_index(n),
        past=state_t[:, :PT].contiguous(), past_gt=state_t[:, :PT].clone(),
        sem=f32(sem), lw=lw_t,
        past_vis=torch.ones((n, PT)),
    )
    if with_future:
        d.future = state_t[:, PT:].contiguous()
        d.future_gt = state_t[:, PT:].clone()
        d.future_vis = torch.ones((n, FT))
    return d


def make_batch(sizes, key='scene', PT=4, FT=12, NC=2, map_extent=(256.0, 256.0), M=1, with_future=True):
    """Batch of scenes with the given agent counts -> (Batch, map_idx (B,) long)."""
    scenes = [make_scene(n, '%s/%d' % (key, b), PT=PT, FT=FT, NC=NC, map_extent=map_extent,
                         with_future=with_future) for b, n in enumerate(sizes)]
    batch = Batch.from_data_list(scenes)
    map_idx = torch.tensor([b % M for b in range(len(sizes))], dtype=torch.long)
    return batch, map_idx


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------

def fill_state_dict(sd, key='weights', scale=1.0):
    """Overwrite every tensor of a TrafficModel ``state_dict`` with counter-based values:
    Linear/Conv/GRU weights and biases U(+-scale/sqrt(fan_in)); LayerNorm/GroupNorm weight 1+small, bias small.
    Returns a new dict (fp32, CPU)."""
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        is_norm = False
        parts = name.split('.')
        # MLP: net.{1,4,7} are LayerNorm; map_conv.{1,4,...} GroupNorm (reference models/common.py:26-39)
        if parts[0] == 'map_conv' and int(parts[1]) % 3 == 1:
            is_norm = True
        if 'net' in parts:
            li = int(parts[parts.index('net') + 1])
            if li % 3 == 1:
                is_norm = True
        if is_norm:
            if parts[-1] == 'weight':
                v = 1.0 + 0.1 * counter_uniform(shape, key + '/' + name, -1.0, 1.0)
            else:
                v = 0.05 * counter_uniform(shape, key + '/' + name, -1.0, 1.0)
        else:
            if len(shape) >= 2:
                fan_in = int(np.prod(shape[1:]))
            else:
                fan_in = None
            if fan_in is None:
                # bias: need fan-in of the matching weight
                wname = name.replace('bias', 'weight')
                wshape = tuple(sd[wname].shape)
                fan_in = int(np.prod(wshape[1:]))
            bound = scale / math.sqrt(fan_in)
            v = counter_uniform(shape, key + '/' + name, -bound, bound)
        out[name] = f32(v).reshape(shape)
    return out


def make_latents(prior_mu, prior_var, key='z', scale=0.5):
    """z = mu + scale*sigma*eps with counter-based eps (the reference never seeds its RNG,
    src/models/traffic_model.py:706-712, so parity tests always inject z explicitly)."""
    eps = f32(counter_normal(tuple(prior_mu.shape), key)).to(prior_mu.device)
    return prior_mu + scale * torch.sqrt(prior_var) * eps
