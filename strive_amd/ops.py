"""Operator layer: torch tensors in, libstrive_hip.so calls, torch tensors out.

Everything here requires ROCm device tensors and the hipcc-built library; a CPU tensor or a missing
library raises ``StriveHipError`` -- there is deliberately no CPU code path (the CPU restatement of
the algorithm lives in ``oracle/`` and is test infrastructure only).  PyTorch is used for device
memory, streams and as the autograd host: the decoder rollout and the vehicle-collision penalty are
``torch.autograd.Function``s whose forward/backward are single C-ABI calls.
"""
import ctypes as C
import threading

import os

import torch

from . import _lib as L
from . import params
from ._lib import StriveHipError

__all__ = ['mlp_forward', 'gnn_forward', 'encode_map', 'encode_traj', 'decoder_rollout', 'map_crop', 'coll_point',
           'veh_coll_penalties', 'scene_info', 'transform2frame']


# ------------------------------------------------------------------------------------------------
# plumbing
# ------------------------------------------------------------------------------------------------

def _lib_for(*tensors):
    """The product rule: HIP or nothing."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise StriveHipError('strive_amd operators run on the MI355X through libstrive_hip.so and need ROCm '
                                 'device tensors; got a %s tensor (no CPU fallback exists -- the CPU restatement '
                                 'in oracle/ is test infrastructure)' % t.device)
    return L.get_lib()


def _stream(t):
    return L.stream_ptr(t)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


_ws_cache = {}
_ws_lock = threading.Lock()
_ws_local = threading.local()


class graph_workspaces(object):
    """Context: scratch buffers requested inside belong to ``store`` -- a dict owned by the HIP graph being captured
    (utils/graphed.GraphedIteration) -- instead of the process-wide cache: a capture allocates from the graph's private pool, and
    such a buffer must neither outlive its graph in a global table nor be handed to another loop's capture.  Per thread."""

    def __init__(self, store):
        self.store = store

    def __enter__(self):
        self.prev = getattr(_ws_local, 'store', None)
        _ws_local.store = self.store
        return self

    def __exit__(self, *exc):
        _ws_local.store = self.prev
        return False


def _workspace(device, nbytes, tag='ws'):
    """Scratch buffer per (device, tag, current stream): launches on different streams never share one.  Inside a
    ``graph_workspaces`` context the buffers live in that context's store."""
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == 'cuda' else 0
    key = (str(device), tag, stream)
    store = getattr(_ws_local, 'store', None)
    cache = _ws_cache if store is None else store
    with _ws_lock:
        buf = cache.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
            cache[key] = buf
    return buf


def _param_signature(*modules):
    return (params.param_epoch(),) + tuple((p.data_ptr(), p._version) for m in modules for p in m.parameters())


def _cached_pack(owner, key, module_for_sig, builder, extra_sig=()):
    """Re-pack only when a parameter changed (in-place updates bump ``_version``; updates that do not -- fused optimisers,
    ``p.data`` -- are announced with params.parameters_changed()) or moved, or when ``extra_sig`` (plain values the pack also
    depends on) changed.  ``module_for_sig`` is a module or a tuple of modules."""
    cache = owner.__dict__.setdefault('_strive_packs', {})
    mods = module_for_sig if isinstance(module_for_sig, (tuple, list)) else (module_for_sig,)
    sig = (_param_signature(*mods), extra_sig)
    ent = cache.get(key)
    if ent is None or ent[0] != sig:
        ent = (sig, builder())
        cache[key] = ent
    return ent[1]


def _cached_pack_hit(owner, key, module_for_sig, extra_sig=()):
    """Would ``_cached_pack`` find an up-to-date entry (nothing is built)?"""
    cache = owner.__dict__.get('_strive_packs')
    if cache is None:
        return False
    mods = module_for_sig if isinstance(module_for_sig, (tuple, list)) else (module_for_sig,)
    ent = cache.get(key)
    return ent is not None and ent[0] == (_param_signature(*mods), extra_sig)


def _sd_of(module, prefix):
    return {prefix + '.' + k: v for k, v in module.state_dict().items()}


# ------------------------------------------------------------------------------------------------
# weight-gradient mode (the training path)
# ------------------------------------------------------------------------------------------------

class _WeightGradMode(object):
    """The latent-optimisation loops differentiate w.r.t. the latents only, and that is all the HIP path computes when its
    operators are called directly (embed / decode_embedding / sample_batched).  ``TrafficModel.forward`` -- the training
    entry point (reference src/train_traffic.py:103-118) -- switches this mode on, and the same operators then run as
    autograd Functions that also return the gradients of every parameter they read."""
    on = False

    def __init__(self, enable):
        self.enable = bool(enable)

    def __enter__(self):
        self.prev = _WeightGradMode.on
        _WeightGradMode.on = self.enable
        return self

    def __exit__(self, *exc):
        _WeightGradMode.on = self.prev


def weight_grad_mode(enable=True):
    return _WeightGradMode(enable)


def _wgrad():
    return _WeightGradMode.on and torch.is_grad_enabled()


class GradSink(object):
    """Where the training Functions put parameter gradients when the caller keeps ALL gradients in one flat buffer whose
    ``p.grad`` are views (strive_amd.distributed.DataParallelTrainer): every ``strive_*_bwd`` call ACCUMULATES into the flat
    layout of its module, so when a module's parameters are consecutive in the buffer the call can add straight into it -- no
    zero-filled scratch, no per-parameter ``add`` by autograd's AccumulateGrad (350 launches per training step).  The Functions
    then return ``None`` for those parameters (= no further contribution)."""
    current = None

    def __init__(self, params, bucket):
        self.bucket = bucket
        self.offset = {}
        off = 0
        for p in params:
            self.offset[id(p)] = (off, p.numel())
            off += p.numel()

    def segment(self, ps):
        """flat view of the bucket that IS the concatenated gradients of ``ps`` (consecutive in the bucket, and every p.grad still
        the bucket view), else None"""
        if not ps:
            return None
        ent = self.offset.get(id(ps[0]))
        if ent is None:
            return None
        off0, off = ent[0], ent[0]
        for p in ps:
            e = self.offset.get(id(p))
            if e is None or e[0] != off or p.grad is None or p.grad.data_ptr() != self.bucket.data_ptr() + 4 * off:
                return None
            off += e[1]
        return self.bucket[off0:off]

    def __enter__(self):
        self.prev = GradSink.current
        GradSink.current = self
        return self

    def __exit__(self, *exc):
        GradSink.current = self.prev


def _grad_target(ps, count, dev):
    """(flat fp32 tensor of ``count`` gradients to ACCUMULATE into, whether the caller must hand the pieces back to autograd)"""
    sink = GradSink.current
    if sink is not None and os.environ.get('STRIVE_NO_GRADSINK', '0') != '1':      # (A/B switch)
        seg = sink.segment(list(ps))
        if seg is not None and seg.numel() == count and seg.device == dev:
            return seg, False
    return torch.zeros((count,), dtype=torch.float32, device=dev), True


def _grad_return(flat, ps, give_back):
    return tuple(_split_like(flat, ps)) if give_back else (None,) * len(ps)


def _split_like(flat, params):
    out, off = [], 0
    for p in params:
        n = p.numel()
        out.append(flat[off:off + n].view(p.shape))
        off += n
    assert off == flat.numel(), 'flat gradient size %d does not match the parameters (%d)' % (flat.numel(), off)
    return out


# ------------------------------------------------------------------------------------------------
# scene structure
# ------------------------------------------------------------------------------------------------

class SceneInfo(object):
    def __init__(self, ptr, device):
        self.ptr = ptr.to(device)
        self.ptr_cpu = ptr.cpu().clone()
        self.ei_sig = None
        self.ptr_sig = None
        self.sizes = (ptr[1:] - ptr[:-1]).cpu()
        self.B = int(self.sizes.shape[0])
        self.NA = int(ptr[-1])
        self.device = device
        self._packs = {}
        sz = self.sizes.to(torch.long)
        self.pair_off = torch.zeros((self.NA,), dtype=torch.int32)
        if self.NA > 0:
            per_agent = torch.repeat_interleave(sz, sz)
            self.pair_off[1:] = torch.cumsum(per_agent, 0)[:-1].to(torch.int32)
        self.P = int((sz * sz).sum())
        self.pair_off = self.pair_off.to(device)

    def pack(self, NS):
        p = self._packs.get(NS)
        if p is None:
            p = params.pack_scenes(self.ptr.cpu(), NS, self.device)
            self._packs[NS] = p
        return p

    def stacked(self, copies):
        """the batch ``copies`` times over as ONE batch of copies * B scenes (rows of copy c at [c * NA, (c + 1) * NA)):
        what decoder_rollout_stacked hands the library"""
        st = self.__dict__.setdefault('_stacked', {})
        s = st.get(copies)
        if s is None:
            ptr = self.ptr_cpu
            s = SceneInfo(torch.cat([ptr[:1]] + [ptr[1:] + c * self.NA for c in range(copies)]), self.device)
            st[copies] = s
        return s


def _expected_clique_keys(ptr_cpu, NA):
    keys = []
    for b in range(ptr_cpu.shape[0] - 1):
        lo, hi = int(ptr_cpu[b]), int(ptr_cpu[b + 1])
        n = hi - lo
        if n <= 1:
            continue
        src = torch.arange(lo, hi).view(n, 1).expand(n, n)
        dst = torch.arange(lo, hi).view(1, n).expand(n, n)
        keep = src != dst
        keys.append(src[keep] * NA + dst[keep])
    return torch.cat(keys) if keys else torch.zeros((0,), dtype=torch.long)


def scene_info(scene_graph):
    """Scene offsets of a batched graph; verifies once that ``edge_index`` is the per-scene clique the
    kernels assume (reference src/datasets/nuscenes_dataset.py:678-687 always builds exactly that)."""
    ptr = scene_graph.ptr
    cached = scene_graph.__dict__.get('_strive_scene_info')
    ei = scene_graph.edge_index if 'edge_index' in scene_graph else None
    # identity + in-place version of the two structure tensors: a cheap check that never touches the device (comparing
    # ptr by value would be a device->host copy, i.e. a synchronisation, on every rollout).  Re-collated or edited graphs
    # carry new tensors / bumped versions and are re-validated.
    ei_sig = None if ei is None else (ei.data_ptr(), ei._version, tuple(ei.shape))
    ptr_sig = (ptr.data_ptr(), ptr._version, tuple(ptr.shape))
    if cached is not None and cached.device == scene_graph.past.device and cached.ei_sig == ei_sig and cached.ptr_sig == ptr_sig:
        return cached
    if cached is not None and cached.device == scene_graph.past.device and cached.ei_sig == ei_sig and \
            cached.ptr_cpu.shape == ptr.shape and torch.equal(cached.ptr_cpu, ptr.cpu()):
        cached.ptr_sig = ptr_sig          # same offsets in a new tensor
        return cached
    info = SceneInfo(ptr.cpu(), scene_graph.past.device)
    info.ei_sig = ei_sig
    info.ptr_sig = ptr_sig
    if 'edge_index' in scene_graph:
        ei = scene_graph.edge_index.cpu()
        exp = _expected_clique_keys(ptr.cpu(), info.NA)
        got = ei[0] * info.NA + ei[1]
        if got.shape[0] != exp.shape[0] or not torch.equal(torch.sort(got)[0], torch.sort(exp)[0]):
            raise NotImplementedError('strive_amd message passing requires per-scene fully connected graphs without '
                                      'self loops (what the reference dataset builds); got a different edge_index')
    scene_graph.__dict__['_strive_scene_info'] = info
    return info


# ------------------------------------------------------------------------------------------------
# small differentiable glue (plain torch; used outside the fused kernels)
# ------------------------------------------------------------------------------------------------

def transform2frame(frame, poses, inverse=False):
    """reference src/utils/transforms.py:78-139 for 4-d poses (x,y,hx,hy); plain torch glue."""
    c = frame[:, 2].unsqueeze(1)
    s = frame[:, 3].unsqueeze(1)
    fx = frame[:, 0].unsqueeze(1)
    fy = frame[:, 1].unsqueeze(1)
    px, py, pc, ps = poses[..., 0], poses[..., 1], poses[..., 2], poses[..., 3]
    if inverse:
        out = [(c * px - s * py) + fx, (s * px + c * py) + fy, pc * c - ps * s, ps * c + pc * s]
    else:
        dx, dy = px - fx, py - fy
        out = [c * dx + s * dy, -s * dx + c * dy, pc * c + ps * s, ps * c - pc * s]
    return torch.stack(out, dim=-1)


# ------------------------------------------------------------------------------------------------
# MLP / GNN forward (embed-time networks; no gradient)
# ------------------------------------------------------------------------------------------------

def _no_grad_inputs(what, *tensors):
    if torch.is_grad_enabled():
        for t in tensors:
            if t is not None and t.requires_grad:
                raise NotImplementedError('%s: the HIP path provides d/dz of the decoder rollout only; gradients '
                                          'w.r.t. this input are not implemented' % what)


class _MLPFn(torch.autograd.Function):
    """MLP with parameter gradients (training path): forward = strive_mlp_fwd, backward = strive_mlp_bwd."""

    @staticmethod
    def forward(ctx, x2, h, *ps):
        y = torch.empty((x2.shape[0], h.O), dtype=torch.float32, device=x2.device)
        h.lib.call('strive_mlp_fwd', h.pk.ref(), L.ptr(x2), x2.shape[0], L.ptr(y), _stream(x2))
        ctx.h, ctx.x2, ctx.ps = h, x2, ps
        return y

    @staticmethod
    def backward(ctx, dy):
        h, x2 = ctx.h, ctx.x2
        n = h.lib.query('strive_mlp_param_count', h.pk.ref())
        dp, give_back = _grad_target(ctx.ps, n, x2.device)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        h.lib.call('strive_mlp_bwd', h.pk.ref(), L.ptr(x2), L.ptr(_f32c(dy)), x2.shape[0], L.ptr(dx), L.ptr(dp), _stream(x2))
        return (dx, None) + _grad_return(dp, ctx.ps, give_back)


def mlp_forward(mlp_module, x):
    lib = _lib_for(x)
    pk = _cached_pack(mlp_module, 'mlp', mlp_module, lambda: params.pack_mlp(_sd_of(mlp_module, 'm'), 'm'))
    shp = x.shape
    if shp[-1] != pk.struct.dims[0]:
        raise ValueError('MLP expects %d input features, got %d' % (pk.struct.dims[0], shp[-1]))
    O = pk.struct.dims[pk.struct.nlayers]
    if _wgrad():
        h = _RolloutCtx()
        h.lib, h.pk, h.O = lib, pk, O
        x2 = x.to(torch.float32).contiguous().reshape(-1, shp[-1])
        return _MLPFn.apply(x2, h, *list(mlp_module.parameters())).reshape(tuple(shp[:-1]) + (O,))
    _no_grad_inputs('MLP.forward', x)
    x2 = _f32c(x).reshape(-1, shp[-1])
    y = torch.empty((x2.shape[0], O), dtype=torch.float32, device=x.device)
    lib.call('strive_mlp_fwd', pk.ref(), L.ptr(x2), x2.shape[0], L.ptr(y), _stream(x))
    return y.reshape(tuple(shp[:-1]) + (O,))


class _GNNFn(torch.autograd.Function):
    """SceneInteractionNet with parameter gradients (training path): strive_gnn_fwd / strive_gnn_bwd."""

    @staticmethod
    def forward(ctx, x2, h, *ps):
        out = torch.empty((h.R, h.O), dtype=torch.float32, device=x2.device)
        wsb = h.lib.query('strive_gnn_workspace_bytes', h.pk.ref(), h.sc.ref())
        ws = _workspace(x2.device, wsb)
        h.lib.call('strive_gnn_fwd', h.pk.ref(), h.sc.ref(), L.ptr(x2), L.ptr(h.pos), L.ptr(h.sem), L.ptr(out), L.ptr(ws),
                   ws.numel(), _stream(x2))
        ctx.h, ctx.x2, ctx.ps = h, x2, ps
        return out

    @staticmethod
    def backward(ctx, d_out):
        h, x2 = ctx.h, ctx.x2
        n = h.lib.query('strive_gnn_param_count', h.pk.ref())
        dp, give_back = _grad_target(ctx.ps, n, x2.device)
        dx = torch.empty_like(x2)
        wsb = h.lib.query('strive_gnn_bwd_workspace_bytes', h.pk.ref(), h.sc.ref())
        ws = _workspace(x2.device, wsb, 'gnn_bwd')
        h.lib.call('strive_gnn_bwd', h.pk.ref(), h.sc.ref(), L.ptr(x2), L.ptr(h.pos), L.ptr(h.sem), L.ptr(_f32c(d_out)), L.ptr(dx),
                   L.ptr(dp), L.ptr(ws), ws.numel(), _stream(x2))
        return (dx, None) + _grad_return(dp, ctx.ps, give_back)


def gnn_forward(net, scene_graph):
    x, pos, sem = scene_graph.x, scene_graph.pos, scene_graph.sem
    lib = _lib_for(x, pos, sem)
    train = _wgrad()
    if train:
        _no_grad_inputs('SceneInteractionNet.forward (poses)', pos)
    else:
        _no_grad_inputs('SceneInteractionNet.forward', x, pos)
    info = scene_info(scene_graph)
    multi = x.dim() == 3
    NS = x.shape[1] if multi else 1
    NC = sem.shape[1]
    pk = _cached_pack(net, 'gnn', net, lambda: params.pack_gnn(_sd_of(net, 'g'), 'g', NC))
    sc = info.pack(NS)
    R = info.NA * NS
    if x.shape[-1] != pk.struct.mlp_in.dims[0] or x.shape[0] != info.NA:
        raise ValueError('interaction net expects (%d, %d) node features, got %s' % (info.NA, pk.struct.mlp_in.dims[0], tuple(x.shape)))
    p2 = _f32c(pos).reshape(R, 4)
    O = pk.struct.mlp_out.dims[pk.struct.mlp_out.nlayers]
    if train:
        h = _RolloutCtx()
        h.lib, h.pk, h.sc, h.R, h.O, h.pos, h.sem = lib, pk, sc, R, O, p2, _f32c(sem)
        out = _GNNFn.apply(x.to(torch.float32).contiguous().reshape(R, -1), h, *list(net.parameters()))
        return out.reshape(info.NA, NS, O) if multi else out
    x2 = _f32c(x).reshape(R, -1)
    out = torch.empty((R, O), dtype=torch.float32, device=x.device)
    wsb = lib.query('strive_gnn_workspace_bytes', pk.ref(), sc.ref())
    ws = _workspace(x.device, wsb)
    lib.call('strive_gnn_fwd', pk.ref(), sc.ref(), L.ptr(x2), L.ptr(p2), L.ptr(_f32c(sem)), L.ptr(out), L.ptr(ws),
             ws.numel(), _stream(x))
    return out.reshape(info.NA, NS, O) if multi else out


# ------------------------------------------------------------------------------------------------
# map
# ------------------------------------------------------------------------------------------------

def _map_pack(map_env, device):
    cache = map_env.__dict__.setdefault('_strive_map_packs', {})
    key = (str(device), map_env.nusc_raster.data_ptr(), tuple(map_env.bounds), map_env.L, map_env.W)
    pk = cache.get(key)
    if pk is None:
        pk = params.pack_map(map_env, device)
        cache.clear()
        cache[key] = pk
    return pk


def map_crop(map_env, frames, mapixes, pos_mean=(0, 0, 0, 0), pos_std=(1, 1, 1, 1), bounds=None, L_=None, W_=None):
    """uint8 crop (N,C,L,W) around frames (N,4); get_map_obs semantics, bit-exact."""
    lib = _lib_for(frames)
    if bounds is not None or L_ is not None or W_ is not None:
        env = _OverrideEnv(map_env, bounds, L_, W_)
    else:
        env = map_env
    pk = _map_pack(env, frames.device)
    N = frames.shape[0]
    out = torch.empty((N, pk.struct.C, pk.struct.L, pk.struct.Wc), dtype=torch.uint8, device=frames.device)
    lib.call('strive_map_crop_u8', pk.ref(), L.ptr(_f32c(frames)), L.f4(pos_mean), L.f4(pos_std),
             L.ptr(mapixes.to(torch.int32).contiguous()), N, L.ptr(out), _stream(frames))
    return out


class _OverrideEnv(object):
    def __init__(self, env, bounds, L_, W_):
        self.nusc_raster, self.nusc_dx = env.nusc_raster, env.nusc_dx
        self.bounds = list(env.bounds if bounds is None else bounds)
        self.L = env.L if L_ is None else L_
        self.W = env.W if W_ is None else W_


def cnn_pack(model):
    """Packed map-CNN weights (two-piece fp16 fragment tables in MFMA operand order + their power-of-two scales + fc), re-packed
    when any map_conv / map_feature parameter changes."""
    return _cached_pack(model, 'cnn', (model.map_conv, model.map_feature), lambda: params.pack_cnn(model.state_dict()))


def keep_cnn_activations():
    """Training forward: keep the raw outputs of the map CNN's six convolutions for the backward instead of recomputing them there
    (strive_map_cnn_fwd_keep / strive_rollout_fwd_keep, 1.76 MB per crop; STRIVE_KEEP_CNN_ACTIVATIONS=0: recompute, the round-4 form)."""
    return os.environ.get('STRIVE_KEEP_CNN_ACTIVATIONS', '1') != '0'


def _alloc_kept(nbytes, device):
    """The kept-activation buffer of a training forward, or None -> the caller takes the recomputing path.  1.76 MB x crops grows
    to tens of GB for large batches (4 scenes of 60 agents x 11 steps x 2 decodes: 10 GB): it is not requested when it would not
    leave a quarter of the currently free device memory (or STRIVE_KEEP_MAX_BYTES) for the rest of the step, and an allocation
    that fails all the same falls back too -- instead of surfacing as an out-of-memory error inside the trainer's step (which
    counts it as a failed batch and skips it)."""
    cap = os.environ.get('STRIVE_KEEP_MAX_BYTES')
    try:
        free = torch.cuda.mem_get_info(device)[0] + (torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
    except Exception:
        free = None
    limit = int(cap) if cap else (None if free is None else int(0.75 * free))
    if limit is not None and nbytes > limit:
        return None
    try:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    except torch.cuda.OutOfMemoryError:
        return None


def _cnn_params(model):
    return list(model.map_conv.parameters()) + list(model.map_feature.parameters())


class _CNNFn(torch.autograd.Function):
    """encode_map with parameter gradients (training path): strive_map_cnn_fwd / strive_map_cnn_bwd (the crop is data)."""

    @staticmethod
    def forward(ctx, h, *ps):
        feat = torch.empty((h.N, 64), dtype=torch.float32, device=h.p2.device)
        wsb = h.lib.query('strive_map_cnn_workspace_bytes', h.N)
        ws = _workspace(h.p2.device, wsb, 'cnn')
        ctx.kept = None
        if keep_cnn_activations():
            # the convolutions' raw outputs stay for the backward (1.76 MB per crop) instead of being recomputed there
            ctx.kept = _alloc_kept(h.lib.query('strive_map_cnn_keep_bytes', h.N), h.p2.device)
        if ctx.kept is not None:
            h.lib.call('strive_map_cnn_fwd_keep', h.mp.ref(), h.cnn.ref(), L.ptr(h.p2), h.mean4, h.std4, L.ptr(h.mapix), h.N, L.ptr(feat),
                       L.ptr(ws), ws.numel(), L.ptr(ctx.kept), ctx.kept.numel(), h.N, 0, _stream(h.p2))
        else:
            h.lib.call('strive_map_cnn_fwd', h.mp.ref(), h.cnn.ref(), L.ptr(h.p2), h.mean4, h.std4, L.ptr(h.mapix), h.N, L.ptr(feat),
                       L.ptr(ws), ws.numel(), _stream(h.p2))
        ctx.h, ctx.ps = h, ps
        return feat

    @staticmethod
    def backward(ctx, d_feat):
        h = ctx.h
        dev = h.p2.device
        dp, give_back = _grad_target(ctx.ps, h.lib.query('strive_map_cnn_param_count'), dev)
        wsb = h.lib.query('strive_map_cnn_bwd_workspace_bytes', h.N)
        ws = _workspace(dev, wsb, 'cnn_bwd')
        if ctx.kept is not None:
            h.lib.call('strive_map_cnn_bwd_kept', h.mp.ref(), h.cnn.ref(), L.ptr(h.p2), h.mean4, h.std4, L.ptr(h.mapix), h.N,
                       L.ptr(_f32c(d_feat)), L.ptr(dp), L.ptr(ctx.kept), ctx.kept.numel(), L.ptr(ws), ws.numel(), _stream(h.p2))
            ctx.kept = None
        else:
            h.lib.call('strive_map_cnn_bwd', h.mp.ref(), h.cnn.ref(), L.ptr(h.p2), h.mean4, h.std4, L.ptr(h.mapix), h.N,
                       L.ptr(_f32c(d_feat)), L.ptr(dp), L.ptr(ws), ws.numel(), _stream(h.p2))
        return (None,) + _grad_return(dp, ctx.ps, give_back)


def encode_map(model, pos, batch_of_agent, map_idx, map_env):
    """Map feature at NORMALISED ``pos`` (NA,4)/(NA,NS,4) -> (NA,[NS,]64): fused crop + CNN.  No gradient w.r.t. the pose
    (the crop is a lookup); parameter gradients on the training path only."""
    lib = _lib_for(pos)
    multi = pos.dim() == 3
    NA = pos.shape[0]
    NS = pos.shape[1] if multi else 1
    dev = pos.device
    mp = _map_pack(map_env, dev)
    cnn = cnn_pack(model)
    mapix = map_idx.to(dev)[batch_of_agent.to(dev)].to(torch.int32)
    if multi:
        mapix = mapix.view(NA, 1).expand(NA, NS).reshape(-1)
    mapix = mapix.contiguous()
    p2 = _f32c(pos).reshape(NA * NS, 4)
    nm = model.normalizer
    if _wgrad():
        h = _RolloutCtx()
        h.lib, h.mp, h.cnn, h.p2, h.mapix, h.N = lib, mp, cnn, p2, mapix, NA * NS
        h.mean4, h.std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
        feat = _CNNFn.apply(h, *_cnn_params(model))
        return feat.reshape(NA, NS, 64) if multi else feat
    feat = torch.empty((NA * NS, 64), dtype=torch.float32, device=dev)
    wsb = lib.query('strive_map_cnn_workspace_bytes', NA * NS)
    ws = _workspace(dev, wsb, 'cnn')
    lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(p2), L.f4(nm.mean_vals[:4].tolist()),
             L.f4(nm.std_vals[:4].tolist()), L.ptr(mapix), NA * NS, L.ptr(feat), L.ptr(ws), ws.numel(), _stream(pos))
    return feat.reshape(NA, NS, 64) if multi else feat


def encode_map_crop(model, crop):
    """Map feature of PRE-CROPPED uint8 rasters ``(N,4,256,256)`` -> (N,64): the CNN alone, no gradient.
    (reference src/models/traffic_model.py:104-113 applied to ``map_env.get_map_crop`` output)"""
    lib = _lib_for(crop)
    if crop.dtype != torch.uint8 or crop.dim() != 4 or tuple(crop.shape[1:]) != (4, 256, 256):
        raise StriveHipError('encode_map_crop expects uint8 (N,4,256,256), got %s %s' % (crop.dtype, tuple(crop.shape)))
    crop = crop.contiguous()
    N = crop.shape[0]
    dev = crop.device
    cnn = cnn_pack(model)
    feat = torch.empty((N, 64), dtype=torch.float32, device=dev)
    wsb = lib.query('strive_map_cnn_workspace_bytes', N)
    ws = _workspace(dev, wsb, 'cnn')
    lib.call('strive_map_cnn_fwd_from_crop', cnn.ref(), L.ptr(crop), N, L.ptr(feat), L.ptr(ws), ws.numel(), _stream(crop))
    return feat


def _rel_pose_data(frame, poses):
    """transform2frame(frame (N,4), poses (N,M,4)) for DATA (no gradient: the encoders' inputs are detached, reference
    src/models/traffic_model.py:453-523) as one strive_rel_pose launch instead of ~16 elementwise ones; plain torch where
    a gradient is asked for."""
    if frame.requires_grad or poses.requires_grad:
        return transform2frame(frame, poses)
    lib = _lib_for(frame, poses)
    fr, po = _f32c(frame), _f32c(poses)
    out = torch.empty_like(po)
    lib.call('strive_rel_pose', L.ptr(fr), L.ptr(po), None, L.ptr(out), None, None, po.shape[0], po.shape[1], _stream(po))
    return out


def encode_traj(model, encoder, g, traj, vis):
    """Past / future trajectory encoder input assembly (torch glue) + HIP MLP.
    (reference src/models/traffic_model.py:453-523)"""
    NA, T, _ = traj.shape
    local = _rel_pose_data(g.past[:, -1, :4], traj[:, :, :4])
    local = torch.cat([local, traj[:, :, 4:]], dim=2)
    local = torch.where((vis == 0.0).unsqueeze(-1), torch.zeros_like(local), local)
    local = torch.cat([local, vis.unsqueeze(-1)], dim=-1)
    att = g.lw.unsqueeze(1).expand(NA, T, 2)
    enc_in = torch.cat([torch.cat([local, att], dim=-1).reshape(NA, -1), g.sem], dim=1)
    return encoder(enc_in.detach())


_lin_tables = {}


def _linspace_pair(gl, gw, dev):
    """the two fp32 linspace(-1, 1, .) tables of get_coll_point, uploaded once per (size, device): a host->device copy per
    call would make the host wait for the stream to drain"""
    key = (gl, gw, str(dev))
    t = _lin_tables.get(key)
    if t is None:
        t = (torch.linspace(-1.0, 1.0, gl).to(dev), torch.linspace(-1.0, 1.0, gw).to(dev))
        _lin_tables[key] = t
    return t


def coll_point(map_env, cars, lw, mapixes, gl, gw):
    """get_coll_point on layer 0: (N,2) collision points (NaN = none / fully off) and off-pixel counts."""
    lib = _lib_for(cars)
    dev = cars.device
    pk = _map_pack(map_env, dev)
    N = cars.shape[0]
    pt = torch.empty((N, 2), dtype=torch.float32, device=dev)
    cnt = torch.empty((N,), dtype=torch.int32, device=dev)
    lin_l, lin_w = _linspace_pair(int(gl), int(gw), dev)
    lib.call('strive_coll_point', pk.ref(), L.ptr(_f32c(cars)), L.ptr(_f32c(lw)), L.ptr(mapixes.to(torch.int32).contiguous()),
             N, int(gl), int(gw), L.ptr(lin_l), L.ptr(lin_w), L.ptr(pt), L.ptr(cnt), _stream(cars))
    return pt, cnt


# ------------------------------------------------------------------------------------------------
# decoder rollout
# ------------------------------------------------------------------------------------------------

class _RolloutCtx(object):
    pass


class _RolloutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, h):
        lib = h.lib
        dev = z.device
        zz = _f32c(z).reshape(h.R, 32)
        traj = torch.empty((h.R, h.FT, 4), dtype=torch.float32, device=dev)
        tape = torch.empty(h.tape_bytes, dtype=torch.uint8, device=dev)
        ws = _workspace(dev, h.ws_bytes, 'rollout')
        lib.call('strive_rollout_fwd', h.dec.ref(), h.sc.ref(), L.ptr(h.past_last), L.ptr(h.lw), L.ptr(h.sem),
                 L.ptr(h.past_feat), L.ptr(h.map_feat), L.ptr(zz), L.ptr(h.mapix), L.ptr(h.ext), h.FT, L.ptr(traj),
                 L.ptr(tape), tape.numel(), L.ptr(ws), ws.numel(), _stream(z))
        ctx.h = h
        ctx.tape = tape
        ctx.zz = zz
        ctx.zshape = z.shape
        return traj

    @staticmethod
    def backward(ctx, d_traj):
        h = ctx.h
        dev = d_traj.device
        dz = torch.empty((h.R, 32), dtype=torch.float32, device=dev)
        ws = _workspace(dev, h.ws_bytes, 'rollout')
        h.lib.call('strive_rollout_bwd', h.dec.ref(), h.sc.ref(), L.ptr(h.lw), L.ptr(h.sem), L.ptr(ctx.zz), L.ptr(h.ext),
                   h.FT, L.ptr(_f32c(d_traj)), L.ptr(dz), L.ptr(ctx.tape), ctx.tape.numel(), L.ptr(ws), ws.numel(),
                   _stream(d_traj))
        return dz.reshape(ctx.zshape), None


_pair_streams = {}


class _RolloutPairFn(torch.autograd.Function):
    """Two decodes that differ only in WHICH latents they are differentiated for (reference src/utils/adv_gen_optim.py:120-121,
    src/utils/sol_optim.py:73-76: ``z_a = collate(tgt_z, other_z.detach())``, ``z_b = collate(tgt_z.detach(), other_z)`` -- the same
    values, the same scene, the same injected future).  Their forward rollouts are the same computation, so it is done ONCE:
    one strive_rollout_fwd, one tape; the two results are handed back as two tensors and each gets its own reverse sweep over the
    shared tape (the sweep is linear in the trajectory adjoint and only reads the tape).  Port B may be shorter (``FT_b`` <= FT:
    the solution loop decodes 16 steps for the ego term and 12 for the others'; the rollout is causal, so B = the first FT_b
    steps).  Gradients: d/dz_a from port A's adjoint, d/dz_b from port B's; autograd then drops the detached halves."""

    @staticmethod
    def forward(ctx, z_a, z_b, h):
        lib = h.lib
        dev = z_a.device
        if os.environ.get('STRIVE_CHECK_PAIR') == '1' and not (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
            # debug aid (one host synchronisation per call; the GPU tests set it): only z_a is rolled out, so a caller whose two
            # latents differ in VALUE -- not just in which half is detached -- would silently get z_a's trajectory for both
            if not torch.equal(z_a.detach(), z_b.detach()):
                raise ValueError('decode_embedding_pair: the two latents must hold the same values (complementary detach of the '
                                 'same leaves); use two decode_embedding calls for different latents')
        zz = _f32c(z_a).reshape(h.R, 32)
        traj = torch.empty((h.R, h.FT, 4), dtype=torch.float32, device=dev)
        tape = torch.empty(h.tape_bytes, dtype=torch.uint8, device=dev)
        ws = _workspace(dev, h.ws_bytes, 'rollout')
        lib.call('strive_rollout_fwd', h.dec.ref(), h.sc.ref(), L.ptr(h.past_last), L.ptr(h.lw), L.ptr(h.sem),
                 L.ptr(h.past_feat), L.ptr(h.map_feat), L.ptr(zz), L.ptr(h.mapix), L.ptr(h.ext), h.FT, L.ptr(traj),
                 L.ptr(tape), tape.numel(), L.ptr(ws), ws.numel(), _stream(z_a))
        ctx.h, ctx.tape, ctx.zz = h, tape, zz
        ctx.shape_a, ctx.shape_b = z_a.shape, z_b.shape
        # a port without an incoming adjoint gets None, not zeros: the closed loop differentiates port B's loss while the planner
        # is still working on port A's input (two backward calls over the one tape), and a sweep of zeros is a whole sweep
        ctx.set_materialize_grads(False)
        return traj, traj[:, :h.FT_b].clone()

    @staticmethod
    def backward(ctx, d_a, d_b):
        h = ctx.h
        dev = (d_a if d_a is not None else d_b).device
        ws = _workspace(dev, h.ws_bytes, 'rollout')

        def sweep(d_traj):
            dz = torch.empty((h.R, 32), dtype=torch.float32, device=dev)
            h.lib.call('strive_rollout_bwd', h.dec.ref(), h.sc.ref(), L.ptr(h.lw), L.ptr(h.sem), L.ptr(ctx.zz), L.ptr(h.ext),
                       h.FT, L.ptr(d_traj), L.ptr(dz), L.ptr(ctx.tape), ctx.tape.numel(), L.ptr(ws), ws.numel(), _stream(d_traj))
            return dz
        ga = gb = None
        want_a = ctx.needs_input_grad[0] and d_a is not None
        want_b = ctx.needs_input_grad[1] and d_b is not None
        both = want_a and want_b and dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing()
        side = None
        if want_b:
            db = _f32c(d_b)
            if h.FT_b < h.FT:           # the steps port B does not have carry no adjoint
                db = torch.cat([db, torch.zeros((h.R, h.FT - h.FT_b, 4), dtype=torch.float32, device=dev)], dim=1)
            if both:
                # the two sweeps are independent latency chains (one workgroup per scene each): B's runs on a side stream under A's
                cur = torch.cuda.current_stream(dev)
                side = _pair_streams.get(str(dev))
                if side is None:
                    side = torch.cuda.Stream(dev)
                    _pair_streams[str(dev)] = side
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ws_b = _workspace(dev, h.ws_bytes, 'rollout')          # (scratch is per stream)
                    dzb = torch.empty((h.R, 32), dtype=torch.float32, device=dev)
                    h.lib.call('strive_rollout_bwd', h.dec.ref(), h.sc.ref(), L.ptr(h.lw), L.ptr(h.sem), L.ptr(ctx.zz), L.ptr(h.ext),
                               h.FT, L.ptr(db), L.ptr(dzb), L.ptr(ctx.tape), ctx.tape.numel(), L.ptr(ws_b), ws_b.numel(), _stream(db))
                db.record_stream(side)
                gb = dzb.reshape(ctx.shape_b)
            else:
                gb = sweep(db).reshape(ctx.shape_b)
        if want_a:
            ga = sweep(_f32c(d_a)).reshape(ctx.shape_a)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
            gb.record_stream(torch.cuda.current_stream(dev))
        return ga, gb, None


class _RolloutTrainFn(torch.autograd.Function):
    """autoregressive_decoder with parameter gradients (training path): strive_rollout_fwd / strive_rollout_bwd_train.
    Differentiable inputs: z, past_feat, map_feat; parameters: decoder_net, decoder_memory, map_conv, map_feature."""

    @staticmethod
    def forward(ctx, z, past_feat, map_feat, h, *ps):
        lib = h.lib
        dev = z.device
        zz = z.to(torch.float32).contiguous().reshape(h.R, 32)
        pf = past_feat.to(torch.float32).contiguous()
        mf = map_feat.to(torch.float32).contiguous()
        traj = torch.empty((h.R, h.FT, 4), dtype=torch.float32, device=dev)
        tape = torch.empty(h.tape_bytes, dtype=torch.uint8, device=dev)
        ws = _workspace(dev, h.ws_bytes, 'rollout')
        ctx.kept = None
        if keep_cnn_activations() and h.FT > 1:
            ctx.kept = _alloc_kept(lib.query('strive_rollout_keep_bytes', h.dec.ref(), h.sc.ref(), h.FT), dev)
        if ctx.kept is not None:
            lib.call('strive_rollout_fwd_keep', h.dec.ref(), h.sc.ref(), L.ptr(h.past_last), L.ptr(h.lw), L.ptr(h.sem), L.ptr(pf),
                     L.ptr(mf), L.ptr(zz), L.ptr(h.mapix), L.ptr(h.ext), h.FT, L.ptr(traj), L.ptr(tape), tape.numel(), L.ptr(ws),
                     ws.numel(), L.ptr(ctx.kept), ctx.kept.numel(), _stream(z))
        else:
            lib.call('strive_rollout_fwd', h.dec.ref(), h.sc.ref(), L.ptr(h.past_last), L.ptr(h.lw), L.ptr(h.sem), L.ptr(pf), L.ptr(mf),
                     L.ptr(zz), L.ptr(h.mapix), L.ptr(h.ext), h.FT, L.ptr(traj), L.ptr(tape), tape.numel(), L.ptr(ws), ws.numel(),
                     _stream(z))
        ctx.h, ctx.tape, ctx.zz, ctx.zshape, ctx.ps = h, tape, zz, z.shape, ps
        return traj

    @staticmethod
    def backward(ctx, d_traj):
        h = ctx.h
        lib = h.lib
        dev = d_traj.device
        dz = torch.empty((h.R, 32), dtype=torch.float32, device=dev)
        dpf = torch.empty((h.R, 64), dtype=torch.float32, device=dev)
        dmf = torch.empty((h.R, 64), dtype=torch.float32, device=dev)
        ng = lib.query('strive_gnn_param_count', C.byref(h.dec.struct.gnn))
        nr = lib.query('strive_gru_param_count')
        nc = lib.query('strive_map_cnn_param_count')
        # the three flat blocks of the call: decoder_net | decoder_memory | map_conv + map_feature (ctx.ps in that order)
        ps = list(ctx.ps)
        counts, blocks, k = (ng, nr, nc), [], 0
        for c in counts:
            sub, tot = [], 0
            while tot < c:
                sub.append(ps[k]); tot += ps[k].numel(); k += 1
            assert tot == c, 'parameter list does not match the flat gradient layout'
            blocks.append(sub)
        targets = [_grad_target(sub, c, dev) for sub, c in zip(blocks, counts)]
        wsb = lib.query('strive_rollout_train_workspace_bytes', h.dec.ref(), h.sc.ref(), h.FT)
        ws = _workspace(dev, wsb, 'rollout_train')
        if ctx.kept is not None:
            lib.call('strive_rollout_bwd_train_kept', h.dec.ref(), h.sc.ref(), L.ptr(h.lw), L.ptr(h.sem), L.ptr(ctx.zz), L.ptr(h.ext),
                     L.ptr(h.mapix), h.FT, L.ptr(_f32c(d_traj)), L.ptr(dz), L.ptr(dpf), L.ptr(dmf), L.ptr(targets[0][0]),
                     L.ptr(targets[1][0]), L.ptr(targets[2][0]), L.ptr(ctx.tape), ctx.tape.numel(), L.ptr(ctx.kept), ctx.kept.numel(),
                     L.ptr(ws), ws.numel(), _stream(d_traj))
            ctx.kept = None
        else:
            lib.call('strive_rollout_bwd_train', h.dec.ref(), h.sc.ref(), L.ptr(h.lw), L.ptr(h.sem), L.ptr(ctx.zz), L.ptr(h.ext),
                     L.ptr(h.mapix), h.FT, L.ptr(_f32c(d_traj)), L.ptr(dz), L.ptr(dpf), L.ptr(dmf), L.ptr(targets[0][0]),
                     L.ptr(targets[1][0]), L.ptr(targets[2][0]), L.ptr(ctx.tape), ctx.tape.numel(), L.ptr(ws), ws.numel(), _stream(d_traj))
        grads = ()
        for (flat, give_back), sub in zip(targets, blocks):
            grads += _grad_return(flat, sub, give_back)
        return (dz.reshape(ctx.zshape), dpf, dmf, None) + grads


class conv2_plain(object):
    """Context: decoder rollouts enqueued inside it hand the library a descriptor with StriveCNN.conv2_plain = 1 (conv2 on its
    non-persistent kernel: the caller runs small kernels on another stream meanwhile, utils.adv_gen_optim.two_rollouts)."""
    on = False

    def __init__(self, enable=True):
        self.enable = bool(enable)

    def __enter__(self):
        self.prev = conv2_plain.on
        conv2_plain.on = self.enable
        return self

    def __exit__(self, *exc):
        conv2_plain.on = self.prev


def _decoder_variant(pack):
    """the same decoder descriptor with conv2_plain set (a copy of the struct; the tensors stay owned by ``pack``)"""
    v = pack.__dict__.get('_conv2_plain_variant')
    if v is None:
        v = params.Packed(type(pack.struct)())
        C.memmove(C.byref(v.struct), C.byref(pack.struct), C.sizeof(pack.struct))
        v.struct.cnn.conv2_plain = 1
        v.keep.append(pack)
        pack.__dict__['_conv2_plain_variant'] = v
    return v


def _decoder_pack_key(model, map_env, dev):
    key = ('dec', str(dev), map_env.nusc_raster.data_ptr())
    # plain values the pack copies: normaliser statistics and the bicycle parameters (not object identities)
    nm, an, bp = model.normalizer, model.att_normalizer, model.bicycle_params
    extra = (tuple(nm.mean_vals.tolist()), tuple(nm.std_vals.tolist()), tuple(an.mean_vals.tolist()), tuple(an.std_vals.tolist()),
             tuple(sorted((k, tuple(v) if isinstance(v, (tuple, list)) else v) for k, v in bp.items())),
             tuple(map_env.bounds), map_env.L, map_env.W)
    return key, extra


def decoder_packs_ready(model, g, map_env, NS, dev):
    """True when a rollout of (model, g, map_env) with NS samples would build nothing: the decoder's weight pack, the scene
    descriptor and the map pack are cached.  Callers that fork side streams (utils.adv_gen_optim.two_rollouts) run serially
    on their own stream until this holds, so that packs are always built -- and their memory owned -- by the caller's stream
    and never written on one stream while another one reads them."""
    if model.normalizer is None or model.att_normalizer is None or model.bicycle_params is None:
        return False
    info = scene_info(g)                      # (host work + two small uploads on the caller's stream the first time)
    key, extra = _decoder_pack_key(model, map_env, dev)
    return _cached_pack_hit(model, key, model, extra) and NS in info._packs


def decoder_rollout_pair(model, g, map_feat, past_feat, z_a, z_b, map_idx, map_env, ext_future, FT_a, FT_b):
    """Two decodes of the SAME latent values (``z_a`` and ``z_b`` differ only in which leaves they reach, see _RolloutPairFn)
    with one forward rollout: -> (traj_a (NA,[1,]FT_a,4) differentiable w.r.t. z_a, traj_b (NA,[1,]FT_b,4) w.r.t. z_b),
    FT_b <= FT_a.  The caller guarantees the values are equal (the loops build both from the same two leaves)."""
    if _wgrad():
        raise NotImplementedError('decoder_rollout_pair serves the latent-optimisation loops, not the training forward')
    if z_a.shape != z_b.shape or int(FT_b) > int(FT_a):
        raise ValueError('decoder_rollout_pair: latents %s vs %s, FT %d vs %d' % (tuple(z_a.shape), tuple(z_b.shape), FT_a, FT_b))
    h, info, multi, NS = _rollout_context(model, g, map_feat, past_feat, z_a, map_idx, map_env, ext_future, FT_a, False)
    h.FT_b = int(FT_b)
    ta, tb = _RolloutPairFn.apply(z_a, z_b, h)
    if multi:
        return ta.reshape(info.NA, NS, h.FT, 4), tb.reshape(info.NA, NS, h.FT_b, 4)
    return ta.reshape(info.NA, h.FT, 4), tb.reshape(info.NA, h.FT_b, 4)


def decoder_rollout(model, g, map_feat, past_feat, z, map_idx, map_env, ext_future, FT):
    """autoregressive_decoder as one fused call; differentiable w.r.t. ``z`` only."""
    train = _wgrad()
    h, info, multi, NS = _rollout_context(model, g, map_feat, past_feat, z, map_idx, map_env, ext_future, FT, train)
    if train:
        ps = list(model.decoder_net.parameters()) + list(model.decoder_memory.parameters()) + _cnn_params(model)
        traj = _RolloutTrainFn.apply(z, past_feat, map_feat, h, *ps)
        return traj.reshape(info.NA, h.FT, 4)
    traj = _RolloutFn.apply(z, h)
    return traj.reshape(info.NA, NS, h.FT, 4) if multi else traj.reshape(info.NA, h.FT, 4)


def decoder_rollout_stacked(model, g, map_feat, past_feat, zs, map_idx, map_env, FT):
    """``[decoder_rollout(..., z) for z in zs]`` (2-D latents, no ext_future) as ONE rollout over the batch stacked len(zs) times:
    the scenes of copy c are scenes c * B .. (c + 1) * B - 1 of one batch of len(zs) * B scenes, so every kernel of a step runs
    once over all copies instead of once per copy.  Scenes do not interact, so each copy's trajectories are what its own
    rollout gives; used by the training forward (reference src/models/traffic_model.py:217-224 decodes the posterior sample and
    the prior sample one after the other).  Not part of the reference's API."""
    k = len(zs)
    if k == 1:
        return [decoder_rollout(model, g, map_feat, past_feat, zs[0], map_idx, map_env, None, FT)]
    if any(z.dim() != 2 or z.shape != zs[0].shape for z in zs):
        raise ValueError('decoder_rollout_stacked takes 2-D latents of one shape')
    train = _wgrad()
    h, info, _, _ = _rollout_context(model, g, map_feat, past_feat, zs[0], map_idx, map_env, None, FT, train, copies=k)
    z = torch.cat(list(zs), 0)
    if train:
        ps = list(model.decoder_net.parameters()) + list(model.decoder_memory.parameters()) + _cnn_params(model)
        traj = _RolloutTrainFn.apply(z, torch.cat([past_feat] * k, 0), torch.cat([map_feat] * k, 0), h, *ps)
    else:
        traj = _RolloutFn.apply(z, h)
    return list(traj.reshape(k, info.NA, h.FT, 4).unbind(0))


def _rollout_context(model, g, map_feat, past_feat, z, map_idx, map_env, ext_future, FT, train, copies=1):
    """descriptors, per-call tensors and sizes of one rollout (shared by decoder_rollout and decoder_rollout_pair);
    ``copies`` > 1: of the batch stacked that many times (decoder_rollout_stacked)"""
    lib = _lib_for(z, map_feat, past_feat, g.past)
    if model.normalizer is None or model.att_normalizer is None or model.bicycle_params is None:
        raise RuntimeError('set_normalizer / set_att_normalizer / set_bicycle_params must be called before decoding')
    if train:
        _no_grad_inputs('decoder (ext_future)', ext_future)
    else:
        _no_grad_inputs('decoder', map_feat, past_feat, ext_future)
    dev = z.device
    info = scene_info(g)
    multi = z.dim() == 3
    if train and multi:
        raise NotImplementedError('the training backward takes 2-D latents (reference TrafficModel.forward never passes samples)')
    NS = z.shape[1] if multi else 1
    if ext_future is not None and multi:
        raise NotImplementedError('ext_future together with multiple samples (the reference crashes there too)')
    NC = g.sem.shape[1]

    def build():
        return params.pack_decoder(model.state_dict(), NC, map_env, dev, model.normalizer, model.att_normalizer,
                                   model.bicycle_params, cnn=cnn_pack(model), map_pack=_map_pack(map_env, dev))
    key, extra = _decoder_pack_key(model, map_env, dev)
    h = _RolloutCtx()
    h.lib = lib
    h.dec = _cached_pack(model, key, model, build, extra_sig=extra)
    if conv2_plain.on:
        h.dec = _decoder_variant(h.dec)
    h.sc = info.pack(NS) if copies == 1 else info.stacked(copies).pack(NS)
    h.R = info.NA * NS * copies
    h.FT = int(FT)
    h.past_last = _f32c(g.past[:, -1, :])
    h.lw = _f32c(g.lw)
    h.sem = _f32c(g.sem)
    h.past_feat = _f32c(past_feat)
    h.map_feat = _f32c(map_feat)
    h.mapix = map_idx.to(dev)[g.batch.to(dev)].to(torch.int32).contiguous()
    if copies > 1:
        if multi or ext_future is not None:
            raise NotImplementedError('stacked rollouts take 2-D latents and no ext_future')
        rep = lambda t: torch.cat([t] * copies, 0)
        h.past_last, h.lw, h.sem, h.mapix = rep(h.past_last), rep(h.lw), rep(h.sem), rep(h.mapix)
        if not train:       # (the training Function takes the two features as differentiable arguments, stacked by the caller)
            h.past_feat, h.map_feat = rep(h.past_feat), rep(h.map_feat)
    h.ext = None if ext_future is None else _f32c(ext_future)
    if h.ext is not None and tuple(h.ext.shape) != (info.B, h.FT, 4):
        raise ValueError('ext_future must be (B, FT, 4), got %s' % (tuple(h.ext.shape),))
    h.tape_bytes = lib.query('strive_rollout_tape_bytes', h.dec.ref(), h.sc.ref(), h.FT)
    h.ws_bytes = lib.query('strive_rollout_workspace_bytes', h.dec.ref(), h.sc.ref(), h.FT)
    return h, info, multi, NS


# ------------------------------------------------------------------------------------------------
# vehicle collision penalties
# ------------------------------------------------------------------------------------------------

class _VehCollFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, traj, h):
        lib = h.lib
        dev = traj.device
        tr = _f32c(traj)
        NA, T, _ = tr.shape
        pen = torch.empty((T, h.P), dtype=torch.float32, device=dev)
        hit = torch.empty((T, h.P), dtype=torch.uint8, device=dev)
        amin = torch.empty((T, h.P), dtype=torch.uint8, device=dev)
        lib.call('strive_veh_coll_fwd', h.sc.ref(), L.ptr(h.pair_off), h.P, L.ptr(tr), T, L.ptr(h.cent_x), L.ptr(h.rad),
                 C.c_float(h.buffer), L.ptr(pen), L.ptr(hit), L.ptr(amin), _stream(traj))
        ctx.h = h
        ctx.save_for_backward(tr, amin)
        ctx.mark_non_differentiable(hit)
        return pen, hit

    @staticmethod
    def backward(ctx, d_pen, _d_hit):
        h = ctx.h
        tr, amin = ctx.saved_tensors
        NA, T, _ = tr.shape
        d_traj = torch.zeros_like(tr)
        h.lib.call('strive_veh_coll_bwd', h.sc.ref(), L.ptr(h.pair_off), h.P, L.ptr(tr), T, L.ptr(h.cent_x), L.ptr(h.rad),
                   C.c_float(h.buffer), L.ptr(_f32c(d_pen)), L.ptr(amin), L.ptr(d_traj), _stream(tr))
        return d_traj, None


class VehCollSetup(object):
    """Per-batch constants of the circle-approximation penalty."""

    def __init__(self, info, cent_x, rad, buffer):
        self.lib = _lib_for(cent_x)
        self.sc = info.pack(1)
        self.pair_off = info.pair_off
        self.P = info.P
        self.cent_x = _f32c(cent_x)
        self.rad = _f32c(rad)
        self.buffer = float(buffer)


def veh_coll_penalties(traj, setup):
    """(pen (T,P) differentiable w.r.t. traj, hit (T,P) uint8) over in-scene ordered pairs."""
    _lib_for(traj)
    return _VehCollFn.apply(traj[:, :, :4], setup)


# ------------------------------------------------------------------------------------------------
# trajectory up-sampling
# ------------------------------------------------------------------------------------------------

_interp_tables = {}


def _interp_taps(T, scale, device):
    """Taps of F.interpolate(mode='linear', align_corners=False, scale_factor=scale), evaluated in fp32 the way ATen's
    area_pixel_compute_source_index does: src = max(rscale*(j+0.5)-0.5, 0), i0 = floor(src), i1 = min(i0+1, T-1),
    w1 = src - i0, w0 = 1 - w1."""
    key = (T, scale, str(device))
    t = _interp_tables.get(key)
    if t is None:
        import numpy as np
        rs = np.float32(1.0 / float(scale))
        j = np.arange(T * scale, dtype=np.float32)
        src = rs * (j + np.float32(0.5)) - np.float32(0.5)
        src = np.maximum(src, np.float32(0.0)).astype(np.float32)
        i0 = np.floor(src).astype(np.int32)
        i1 = np.minimum(i0 + 1, T - 1).astype(np.int32)
        w1 = (src - i0.astype(np.float32)).astype(np.float32)
        w0 = (np.float32(1.0) - w1).astype(np.float32)
        t = tuple(torch.from_numpy(a).to(device) for a in (i0, i1, w0, w1))
        _interp_tables[key] = t
    return t


class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, traj, scale):
        lib = _lib_for(traj)
        x = _f32c(traj)
        N, T, _ = x.shape
        TO = T * scale
        i0, i1, w0, w1 = _interp_taps(T, scale, x.device)
        out = torch.empty((N, TO, 4), dtype=torch.float32, device=x.device)
        ctx.save_for_backward(x)
        ctx.scale = scale
        if N == 0:
            return out
        lib.call('strive_interp_traj_fwd', L.ptr(x), N, T, TO, L.ptr(i0), L.ptr(i1), L.ptr(w0), L.ptr(w1), L.ptr(out), _stream(x))
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, = ctx.saved_tensors
        N, T, _ = x.shape
        scale = ctx.scale
        i0, i1, w0, w1 = _interp_taps(T, scale, x.device)
        d_in = torch.empty_like(x)
        if N == 0:
            return d_in, None
        _lib_for(x).call('strive_interp_traj_bwd', L.ptr(x), L.ptr(_f32c(d_out)), N, T, T * scale, scale, L.ptr(i0), L.ptr(i1),
                         L.ptr(w0), L.ptr(w1), L.ptr(d_in), _stream(x))
        return d_in, None


# ------------------------------------------------------------------------------------------------
# AvoidCollLoss in one call per direction
# ------------------------------------------------------------------------------------------------

class AvoidCollSetup(object):
    """Constants of one batch for strive_avoid_coll_fwd/bwd (include/strive_hip.h StriveAvoidColl).  ``veh`` is the
    VehCollSetup of the batch, ``pair_valid`` (P,) bool, ``env_agent`` (NE,) long, ``env_lw`` (NE,2), ``env_mapix`` (NE,),
    ``env_pdist`` (NE,), ``grid`` = callable T_fine -> (gl, gw), ``init_z`` (NA,D) or None, ``weights`` the four loss weights
    in the order (coll_veh, coll_env, motion_prior, init_z)."""

    def __init__(self, info, veh, pair_valid, env_agent, env_lw, env_mapix, env_pdist, map_env, grid, init_z, weights, scale=3):
        dev = veh.cent_x.device
        self.lib = veh.lib
        self.info, self.veh, self.map_env, self.grid, self.scale = info, veh, map_env, grid, int(scale)
        self.sc = info.pack(1)
        self.pair_valid = pair_valid.to(torch.uint8).contiguous()
        self.env_agent = env_agent.to(device=dev, dtype=torch.int32).contiguous()
        inv = torch.full((info.NA,), -1, dtype=torch.int32, device=dev)
        inv[env_agent.to(dev).long()] = torch.arange(self.env_agent.numel(), dtype=torch.int32, device=dev)
        self.env_of_agent = inv
        self.env_lw = _f32c(env_lw)
        self.env_mapix = env_mapix.to(device=dev, dtype=torch.int32).contiguous()
        self.env_pdist = _f32c(env_pdist)
        self.init_z = None if init_z is None else _f32c(init_z.detach())
        self.weights = tuple(float(w) for w in weights)
        self._structs = {}

    def fill_base(self, h, T, NZ, D, singleton):
        """fill a StriveAvoidColl; returns the tensors its pointers refer to (keep them alive with the struct)"""
        dev = self.veh.cent_x.device
        TO = T * self.scale
        i0, i1, w0, w1 = _interp_taps(T, self.scale, dev)
        gl, gw = self.grid(TO) if self.weights[1] > 0.0 and self.env_agent.numel() > 0 else (1, 1)
        lin_l, lin_w = _linspace_pair(int(gl), int(gw), dev)
        h.pair_off, h.P = self.veh.pair_off.data_ptr(), self.veh.P
        h.cent_x, h.rad, h.buffer = self.veh.cent_x.data_ptr(), self.veh.rad.data_ptr(), self.veh.buffer
        h.pair_valid = self.pair_valid.data_ptr()
        h.i0, h.i1, h.w0, h.w1, h.scale = i0.data_ptr(), i1.data_ptr(), w0.data_ptr(), w1.data_ptr(), self.scale
        h.NE = self.env_agent.numel()
        h.env_agent, h.env_of_agent = self.env_agent.data_ptr(), self.env_of_agent.data_ptr()
        h.env_lw, h.env_mapix, h.env_pdist = self.env_lw.data_ptr(), self.env_mapix.data_ptr(), self.env_pdist.data_ptr()
        h.gl, h.gw, h.lin_l, h.lin_w = int(gl), int(gw), lin_l.data_ptr(), lin_w.data_ptr()
        h.init_z = None if self.init_z is None else self.init_z.data_ptr()
        h.NZ, h.D = NZ, D
        h.prior_den, h.init_den = float(NZ), float(NZ * D if singleton else NZ)
        h.w_veh, h.w_env, h.w_prior, h.w_init = self.weights
        return (i0, i1, w0, w1, lin_l, lin_w)

    def struct_for(self, T, NZ, D, singleton):
        key = (T, NZ, D, singleton)
        st = self._structs.get(key)
        if st is None:
            h = L.StriveAvoidColl()
            keep = self.fill_base(h, T, NZ, D, singleton)
            nbytes = self.lib.query('strive_avoid_coll_workspace_bytes', self.sc.ref(), C.byref(h), T)
            st = (h, keep, int(nbytes))
            self._structs[key] = st
        return st


class _AvoidCollFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, traj, z, mu, var, h):
        lib = h.lib
        singleton = z.dim() == 3
        tr, zc, muc, varc = _f32c(traj), _f32c(z).reshape(z.shape[0], -1), _f32c(mu), _f32c(var)
        NA, T, _ = tr.shape
        if zc.shape != muc.shape or zc.shape != varc.shape or (h.init_z is not None and h.init_z.numel() != zc.numel()):
            raise ValueError('avoid_coll_loss: latent %s, prior %s / %s, init_z %s do not match' % (
                tuple(z.shape), tuple(mu.shape), tuple(var.shape), None if h.init_z is None else tuple(h.init_z.shape)))
        ctx.key = (T, zc.shape[0], zc.shape[1], singleton)
        ctx.zshape = z.shape
        st, _keep, nbytes = h.struct_for(*ctx.key)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=tr.device)
        out = torch.empty((8,), dtype=torch.float32, device=tr.device)
        pk = _map_pack(h.map_env, tr.device)
        lib.call('strive_avoid_coll_fwd', h.sc.ref(), pk.ref(), C.byref(st), L.ptr(tr), T, L.ptr(zc), L.ptr(muc), L.ptr(varc),
                 L.ptr(out), L.ptr(ws), nbytes, _stream(tr))
        ctx.h = h
        ctx.save_for_backward(tr, zc, muc, varc, ws)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, d_loss, _d_out):
        h = ctx.h
        tr, zc, muc, varc, ws = ctx.saved_tensors
        NA, T, _ = tr.shape
        st, _keep, nbytes = h.struct_for(*ctx.key)
        d_traj = torch.empty_like(tr)
        d_z = torch.zeros_like(zc)
        h.lib.call('strive_avoid_coll_bwd', h.sc.ref(), C.byref(st), L.ptr(tr), T, L.ptr(zc), L.ptr(muc), L.ptr(varc),
                   L.ptr(_f32c(d_loss).reshape(1)), L.ptr(ws), nbytes, L.ptr(d_traj), L.ptr(d_z), _stream(tr))
        return d_traj, d_z.view(ctx.zshape), None, None, None


def avoid_coll_loss(traj, z, mu, var, setup):
    """-> (loss 0-dim, differentiable w.r.t. traj and z; out (8,) = loss, the four means, the two counts)."""
    if mu.requires_grad or var.requires_grad:
        raise NotImplementedError('the fused AvoidCollLoss treats the prior as a constant (the optimisation loops detach it)')
    return _AvoidCollFn.apply(traj[:, :, :4], z, mu, var, setup)


class AdvGenSetup(object):
    """Constants of one batch for strive_adv_gen_fwd/bwd (include/strive_hip.h StriveAdvGen).  ``base`` is an AvoidCollSetup
    whose environment agents are the non-ego agents and whose weights are (coll_veh, coll_env, motion_prior, init_z);
    ``ne_ptr`` (B+1,) offsets of the scenes in the non-ego order; ``slot_ne`` (P,) non-ego row of the non-ego member of every
    ego-involving pair slot (-1 elsewhere); ``extra`` = (adv_crash, coll_veh_plan, motion_prior_atk, init_z_atk)."""

    def __init__(self, base, ne_ptr, slot_ne, t0, infront, extra):
        dev = base.veh.cent_x.device
        self.base, self.lib, self.sc, self.map_env = base, base.lib, base.sc, base.map_env
        self.ne_ptr = ne_ptr.to(device=dev, dtype=torch.int32).contiguous()
        self.slot_ne = slot_ne.to(device=dev, dtype=torch.int32).contiguous()
        self.t0, self.infront = int(t0), infront
        self.extra = tuple(float(w) for w in extra)
        self.NE = base.env_agent.numel()
        self._structs = {}
        self._masks = {}

    def atk_mask(self, attack_agt_idx, NA):
        """(NE,) uint8 selection of the allowed attackers from GLOBAL agent indices (cached per index tensor)"""
        if attack_agt_idx is None:
            return None
        key = (attack_agt_idx.data_ptr(), attack_agt_idx._version, tuple(attack_agt_idx.shape))
        m = self._masks.get(key)
        if m is None:
            dev = self.base.veh.cent_x.device
            am = torch.zeros((NA,), dtype=torch.uint8, device=dev)
            am[attack_agt_idx.to(dev).long()] = 1
            m = am.index_select(0, self.base.env_agent.long()).contiguous()
            self._masks.clear()
            self._masks[key] = m
        return m

    def struct_for(self, T, D, mask, alive=None):
        key = (T, D, None if mask is None else mask.data_ptr(), None if alive is None else alive.data_ptr())
        st = self._structs.get(key)
        if st is None:
            h = L.StriveAdvGen()
            keep = self.base.fill_base(h.base, T, self.NE, D, False)
            h.ne_ptr, h.slot_ne = self.ne_ptr.data_ptr(), self.slot_ne.data_ptr()
            h.atk_mask = None if mask is None else mask.data_ptr()
            h.scene_alive = None if alive is None else alive.data_ptr()
            h.t0 = self.t0
            h.use_infront = 0 if self.infront is None else 1
            h.infront = 0.0 if self.infront is None else float(self.infront)
            h.w_crash, h.w_plan, h.w_prior_atk, h.w_init_atk = self.extra
            nbytes = self.lib.query('strive_adv_gen_workspace_bytes', self.sc.ref(), C.byref(h), T)
            if nbytes == 0:
                raise StriveHipError('strive_adv_gen_workspace_bytes rejected the configuration (T=%d, t0=%d)' % (T, self.t0))
            st = (h, (keep, mask, alive), int(nbytes))
            if len(self._structs) > 8:
                self._structs.clear()
            self._structs[key] = st
        return st


class _AdvGenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, traj, tgt, z, mu, var, h, mask, alive=None):
        lib = h.lib
        tr, tg, zc, muc, varc = _f32c(traj), _f32c(tgt), _f32c(z), _f32c(mu), _f32c(var)
        NA, T, _ = tr.shape
        if zc.shape != muc.shape or zc.shape != varc.shape or zc.shape[0] != h.NE or tg.shape != (h.sc.struct.B, T, 4):
            raise ValueError('adv_gen_loss: shapes traj %s tgt %s z %s prior %s do not match the batch' % (
                tuple(traj.shape), tuple(tgt.shape), tuple(z.shape), tuple(mu.shape)))
        if alive is not None and (alive.dtype != torch.uint8 or tuple(alive.shape) != (h.sc.struct.B,) or alive.device != tr.device or
                                  not alive.is_contiguous()):
            raise ValueError('adv_gen_loss: scene_alive must be a contiguous uint8 tensor (B,) on the device of the trajectories')
        st, _keep, nbytes = h.struct_for(T, zc.shape[1], mask, alive)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=tr.device)
        out = torch.empty((16,), dtype=torch.float32, device=tr.device)
        NT = T - h.t0
        soft = torch.empty((h.NE, NT), dtype=torch.float32, device=tr.device)
        rew = torch.empty((h.NE,), dtype=torch.float32, device=tr.device)
        pk = _map_pack(h.map_env, tr.device)
        lib.call('strive_adv_gen_fwd', h.sc.ref(), pk.ref(), C.byref(st), L.ptr(tr), L.ptr(tg), T, L.ptr(zc), L.ptr(muc), L.ptr(varc),
                 L.ptr(out), L.ptr(soft), L.ptr(rew), L.ptr(ws), nbytes, _stream(tr))
        ctx.h, ctx.mask, ctx.D, ctx.alive = h, mask, zc.shape[1], alive
        ctx.save_for_backward(tr, tg, zc, muc, varc, ws)
        ctx.mark_non_differentiable(out, soft, rew)
        return out[0].clone(), out, soft, rew

    @staticmethod
    def backward(ctx, d_loss, _d_out, _d_soft, _d_rew):
        h = ctx.h
        tr, tg, zc, muc, varc, ws = ctx.saved_tensors
        NA, T, _ = tr.shape
        st, _keep, nbytes = h.struct_for(T, ctx.D, ctx.mask, ctx.alive)
        d_traj, d_tgt, d_z = torch.empty_like(tr), torch.empty_like(tg), torch.zeros_like(zc)
        h.lib.call('strive_adv_gen_bwd', h.sc.ref(), C.byref(st), L.ptr(tr), L.ptr(tg), T, L.ptr(zc), L.ptr(muc), L.ptr(varc),
                   L.ptr(_f32c(d_loss).reshape(1)), L.ptr(ws), nbytes, L.ptr(d_traj), L.ptr(d_tgt), L.ptr(d_z), _stream(tr))
        return d_traj, d_tgt, d_z, None, None, None, None, None


def adv_gen_loss(traj, tgt, z, mu, var, setup, attack_agt_idx=None, scene_alive=None):
    """-> (loss 0-dim, differentiable w.r.t. traj, tgt and z; out (16,); soft (NE, T - t0); rew (NE,)).  ``scene_alive`` (B,) uint8
    on the device (or None): scenes with 0 have left the batch (StriveAdvGen.scene_alive) -- the values and gradients of the others
    are those of the batch rebuilt without them; the mask is read when the kernels run, so a tensor the planner call of the same
    iteration wrote is honoured without a host round trip."""
    if mu.requires_grad or var.requires_grad:
        raise NotImplementedError('the fused AdvGenLoss treats the prior as a constant (the optimisation loops detach it)')
    mask = setup.atk_mask(attack_agt_idx, traj.shape[0])
    return _AdvGenFn.apply(traj[:, :, :4], tgt[:, :, :4], z, mu, var, setup, mask, scene_alive)


def rect_iou(box_a, lw_a, box_b, lw_b):
    """IoU of P rotated vehicle boxes: poses (P,4) = (x, y, hx, hy), sizes (P,2) = (l, w) -> float64 (P,), NaN where a pose
    contains NaN (HIP kernel; the reference loops over shapely polygons, src/losses/adv_gen_nusc.py:517-623)."""
    lib = _lib_for(box_a, box_b)
    a, la, b, lb = _f32c(box_a), _f32c(lw_a), _f32c(box_b), _f32c(lw_b)
    P = a.shape[0]
    if tuple(a.shape) != (P, 4) or tuple(b.shape) != (P, 4) or tuple(la.shape) != (P, 2) or tuple(lb.shape) != (P, 2):
        raise ValueError('rect_iou expects (P,4) poses and (P,2) sizes')
    out = torch.empty((P,), dtype=torch.float64, device=a.device)
    if P > 0:
        lib.call('strive_rect_iou', L.ptr(a), L.ptr(la), L.ptr(b), L.ptr(lb), P, L.ptr(out), _stream(a))
    return out


def interp_traj(traj, scale):
    """(N,T,4) -> (N,T*scale,4): linear up-sampling + heading renormalisation, HIP forward and backward."""
    return _InterpFn.apply(traj[:, :, :4], int(scale))
