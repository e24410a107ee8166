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

    def __init__(self, raster, dx, bounds=(-17.0, -38.5, 60.0, 38.5), L=256, W=256, lane_graph=None):
        self.lane_graphs = None if lane_graph is None else {'synthetic-%d' % i: lane_graph for i in range(raster.shape[0])}
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
               with_future=True, poses=None):
    """One scene of ``n`` agents as a :class:`Data` (NORMALISED), ego at node 0.  ``poses`` = (px, py, h, s) arrays
    override the drawn positions / headings / speeds (e.g. agents placed on a lane graph)."""
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
    if poses is not None:
        px, py, h, s = [np.asarray(a, dtype=np.float64) for a in poses]
        hdot = np.zeros((n,))
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

def lane_keeping_weights(sd, yaw=0.05, acc=0.3):
    """A copy of a synthetic state dict whose decoder steers and accelerates gently: the last layer of ``decoder_net.mlp_out``
    (2 outputs per agent and step: acceleration, yaw acceleration -- reference src/models/traffic_model.py:655-662) scaled by
    ``acc`` / ``yaw``.  With the plain random weights the predicted agents swing 40-60 degrees off their lanes within the 6 s
    horizon and end up 10 m from any lane: the rule-based planner (reference src/planners/hardcode_goalcond_nusc.py), whose routes
    follow lanes, then gives up on a third of the scenes ('arc length outside a route'), which says something about random
    weights, not about adv_gen_rule_based.cfg.  Same shapes, same kernels, same cost: only the closed-loop workloads use it."""
    out = {k: v.clone() for k, v in sd.items()}
    w, b = out['decoder_net.mlp_out.net.6.weight'], out['decoder_net.mlp_out.net.6.bias']
    w[0] *= acc
    b[0] *= acc
    w[1] *= yaw
    b[1] *= yaw
    return out


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


# --------------------------------------------------------------------------------------------
# lane graph (the rule-based planner's map input)
# --------------------------------------------------------------------------------------------

def assemble_lane_graph(lanes, outgoing):
    """Lane polylines + lane-level connectivity -> the lane-graph dict the reference's map environment holds per map
    (reference src/datasets/nuscenes_utils.py:50-123, built there from the nuScenes devkit): ``xy (n,2)`` node positions,
    ``in_edges`` / ``out_edges`` (lists of node lists), ``edges (m,5)`` = (x0, y0, unit direction, length) of every directed
    node pair, ``edgeixes (m,2)`` and ``ee2ix`` {(v0,v1): edge index}.  ``lanes``: list of (k_i, 2) arrays (k_i >= 2);
    ``outgoing[i]``: ids of the lanes that continue lane i (its last node connects to their first node)."""
    start, xys = [], []
    for ln in lanes:
        start.append(len(xys))
        xys.extend(np.asarray(ln, dtype=np.float64).tolist())
    n = len(xys)
    out_edges = [[] for _ in range(n)]
    in_edges = [[] for _ in range(n)]
    for i, ln in enumerate(lanes):
        for k in range(len(ln) - 1):
            out_edges[start[i] + k].append(start[i] + k + 1)
            in_edges[start[i] + k + 1].append(start[i] + k)
        for j in outgoing[i]:
            out_edges[start[i] + len(ln) - 1].append(start[j])
    for i, ln in enumerate(lanes):                 # incoming connections in lane order, like the devkit's connectivity
        for j in range(len(lanes)):
            if i in outgoing[j]:
                in_edges[start[i]].append(start[j] + len(lanes[j]) - 1)
    xy = np.asarray(xys, dtype=np.float64)
    edges, edgeixes, ee2ix = [], [], {}
    for v0 in range(n):
        for v1 in out_edges[v0]:
            d = xy[v1] - xy[v0]
            ln_ = float(np.linalg.norm(d))
            ee2ix[(v0, v1)] = len(edges)
            edges.append([xy[v0, 0], xy[v0, 1], d[0] / ln_, d[1] / ln_, ln_])
            edgeixes.append([v0, v1])
    return {'xy': xy, 'in_edges': in_edges, 'out_edges': out_edges, 'edges': np.asarray(edges),
            'edgeixes': np.asarray(edgeixes, dtype=np.int64), 'ee2ix': ee2ix}


def make_lane_graph(extent=256.0, period=40.0, centre=9.0, lane_off=3.0, step=2.0):
    """Lane graph of the synthetic raster (`make_raster`, map 0: roads of width 18 m every 40 m in both directions): one
    lane per direction on every road, cut into segments at the crossings; at a crossing every arriving segment continues
    straight (first connection) and turns right (second connection), so node out-degrees of 2 and in-degrees of 2 occur."""
    cs = [c for c in np.arange(centre, extent, period)]
    cuts = [0.0] + cs + [extent]            # segment borders along a road: the crossing road centres

    def seg(p0, p1):
        L = float(np.linalg.norm(np.asarray(p1) - np.asarray(p0)))
        k = max(2, int(round(L / step)) + 1)
        t = np.linspace(0.0, 1.0, k)[:, None]
        return (1 - t) * np.asarray(p0, dtype=np.float64)[None] + t * np.asarray(p1, dtype=np.float64)[None]

    lanes, meta = [], []                     # meta: (orientation 'h'/'v', road index, direction +-1, segment index)
    for ri, c in enumerate(cs):
        for d in (+1, -1):
            for si in range(len(cuts) - 1):
                a, b = (cuts[si], cuts[si + 1]) if d > 0 else (cuts[si + 1], cuts[si])
                # keep segment ends 0.5 m away from the crossing centre so that consecutive lanes do not share a point
                a2 = a + 0.5 * np.sign(b - a)
                b2 = b - 0.5 * np.sign(b - a)
                y = c - d * lane_off           # drive on the right: eastbound lane below the centre line
                lanes.append(seg((a2, y), (b2, y)))
                meta.append(('h', ri, d, si))
                x = c + d * lane_off           # northbound lane to the right of the centre line
                lanes.append(seg((x, a2), (x, b2)))
                meta.append(('v', ri, d, si))
    index = {m: i for i, m in enumerate(meta)}
    nseg = len(cuts) - 1
    outgoing = [[] for _ in lanes]
    for i, (o, ri, d, si) in enumerate(meta):
        nxt = si + 1 if d > 0 else si - 1
        if 0 <= nxt < nseg:
            outgoing[i].append(index[(o, ri, d, nxt)])          # straight on
            # right turn at the crossing this segment ends at: crossing road index = the cut between si and nxt
            ci = si if d > 0 else si - 1                         # index into cs of that crossing road
            if 0 <= ci < len(cs):
                if o == 'h':
                    # heading +x turns right to -y (southbound, d = -1); heading -x turns right to +y
                    td = -1 if d > 0 else +1
                    tseg = ri if td < 0 else ri + 1              # southbound: the segment below road ri; northbound: above
                    key = ('v', ci, td, tseg)
                else:
                    td = +1 if d > 0 else -1                     # heading +y turns right to +x; heading -y to -x
                    tseg = ri + 1 if td > 0 else ri
                    key = ('h', ci, td, tseg)
                if key in index:
                    outgoing[i].append(index[key])
    return assemble_lane_graph(lanes, outgoing)


def lane_scene_poses(lane_graph, n, key, radius=35.0, centre=(128.0, 128.0), min_gap=6.0, speed=(2.0, 8.0)):
    """(px, py, h, s) of ``n`` agents sitting on lane nodes within ``radius`` of ``centre`` (ego = the node closest to the
    centre), headed along their lane, speeds ``speed`` (default 2..8 m/s), no two closer than ``min_gap`` (default 6 m)."""
    xy = lane_graph['xy']
    d = np.linalg.norm(xy - np.asarray(centre)[None], axis=1)
    cand = [int(i) for i in np.argsort(d) if d[i] <= radius and len(lane_graph['out_edges'][int(i)]) > 0]
    pick = [cand[0]]
    order = np.argsort(counter_uniform((len(cand),), key + '/pick'))
    for k in order:
        c = cand[int(k)]
        if len(pick) >= n:
            break
        if all(np.linalg.norm(xy[c] - xy[q]) >= min_gap for q in pick):
            pick.append(c)
    assert len(pick) == n, 'not enough lane nodes for %d agents' % n
    px, py = xy[pick, 0].copy(), xy[pick, 1].copy()
    nxt = [lane_graph['out_edges'][c][0] for c in pick]
    hd = xy[nxt] - xy[pick]
    h = np.arctan2(hd[:, 1], hd[:, 0])
    s = counter_uniform((n,), key + '/ls', speed[0], speed[1])
    return px, py, h, s
