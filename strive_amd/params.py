"""Pack TrafficModel parameters into the descriptor structs of include/strive_hip.h.

Packing is pure layout work (transposes / permutes with torch on the parameters' own device); the
packed tensors are kept alive by the returned holder object.  Layouts are documented in
include/strive_hip.h next to each struct.
"""
import ctypes as C
import os

import torch

from . import _lib as L


def _c(t):
    return t.detach().to(torch.float32).contiguous()


class Packed(object):
    """A ctypes struct plus the tensors its pointers refer to."""

    def __init__(self, struct):
        self.struct = struct
        self.keep = []

    def hold(self, t):
        self.keep.append(t)
        return t.data_ptr()

    def ref(self):
        return C.byref(self.struct)


# ---- max |w| of parameter tensors: ONE device read-back for as many tensors as the caller can name up front ----
# (the power-of-two operand scales are host scalars -- they are kernel arguments -- so each needs max |w| on the host; fetching
#  them one by one was ~150 stream synchronisations per training step, where every pack is rebuilt after the optimiser step)
_ABSMAX = {}          # key -> (value, tensor): the tensor reference keeps its storage alive, so an address is never reused by
                      # another tensor while its entry exists (temporaries of non-fp32 parameters would otherwise alias)


# Parameter epoch.  The weight packs and max |w| bounds are cached per (address, in-place version) of a parameter -- but not every
# update bumps the version counter: torch's FUSED optimisers (Adam(fused=True): torch._fused_adam_) write the parameters without
# touching it, and so does anything that goes through ``p.data``.  ``parameters_changed()`` starts a new epoch: every cached pack
# and bound is rebuilt at its next use.  DataParallelTrainer.step calls it after every optimiser step; a training loop of your own
# must do the same if its optimiser is one of those.
_PARAM_EPOCH = [0]


def parameters_changed():
    _PARAM_EPOCH[0] += 1
    return _PARAM_EPOCH[0]


def param_epoch():
    return _PARAM_EPOCH[0]


def _absmax_key(t):
    return (t.data_ptr(), t._version, tuple(t.shape), str(t.device), _PARAM_EPOCH[0])


class lagged_absmax(object):
    """Context (the training step): prefetch_absmax() does not wait for the device.  It enqueues the read-back of max |t| into
    pinned memory and serves, for every tensor, the value read back exactly two calls earlier (the first call of a tensor
    waits and serves its own) plus ``slack`` -- the most the tensors can have grown since: ``slack`` = a bound on two optimiser
    steps' change of an entry (Adam: ~4 lr per step).  The power-of-two operand scales are then chosen with a factor 2 of extra head-room
    (_pow2_scale), so a stale, slightly-too-small bound cannot overflow fp16: |w| scale <= 16384 (1 + slack / max|w|) << 65504."""
    active = None

    def __init__(self, slack):
        self.slack = float(slack)

    def __enter__(self):
        self.prev = lagged_absmax.active
        lagged_absmax.active = self
        return self

    def __exit__(self, *exc):
        lagged_absmax.active = self.prev


# Lagged read-backs.  An entry is keyed by what identifies the parameter's storage slot (address, shape, device) AND keeps the
# tensor, so its address cannot be handed to another tensor while the entry exists (a second model in the process, a replaced
# p.data: each gets its own first-sight read-back instead of a stranger's stale bound).  The lag is FIXED: the value served at a
# call is the one read back _LAG_DEPTH calls of the same tensor group earlier (its event is waited for if it has not arrived --
# it practically always has), never older and never a matter of ev.query() timing, so the scales a run chooses are reproducible
# and the caller's slack (a bound on _LAG_DEPTH optimiser steps) always covers it.
_LAG_DEPTH = 2
_LAG = {}             # (data_ptr, shape, device) -> [value, tensor]
_LAG_BATCH = {}       # tensor group (tuple of keys) -> FIFO of read-backs under way: (pinned values, event)


def _lag_key(t):
    return (t.data_ptr(), tuple(t.shape), str(t.device))


def _lagged_prefetch(tensors, slack):
    todo = [t.detach() for t in tensors if t.numel() > 0 and _absmax_key(t.detach()) not in _ABSMAX]
    if not todo:
        return
    dev = todo[0].device
    if dev.type != 'cuda':
        vals = torch.stack(torch._foreach_norm([t.to(torch.float32) for t in todo], float('inf'))).tolist()
        for t, v in zip(todo, vals):
            _ABSMAX[_absmax_key(t)] = (float(v), t)
        return
    keys = [_lag_key(t) for t in todo]
    group = tuple(keys)
    fifo = _LAG_BATCH.setdefault(group, [])
    # enqueue this call's read-back
    norms = torch.stack(torch._foreach_norm([t.to(torch.float32) for t in todo], float('inf')))
    host = torch.empty((len(todo),), dtype=torch.float32).pin_memory()
    host.copy_(norms, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    fifo.append((host, ev))
    # first sight of a tensor (new address / shape, or an entry whose tensor is not this storage any more): this read-back is
    # waited for and served exactly
    first = any(k not in _LAG or _LAG[k][1].data_ptr() != k[0] for k in keys)
    if first:
        ev.synchronize()
        vals = host.tolist()
        del fifo[:]
        if len(_LAG) + len(todo) > 4096:
            _LAG.clear()
        for k, v, t in zip(keys, vals, todo):
            _LAG[k] = [float(v), t]
        slack = 0.0
    else:
        while len(fifo) > _LAG_DEPTH:                       # the read-back of _LAG_DEPTH calls ago becomes the value served
            h, e = fifo.pop(0)
            e.synchronize()
            for k, v in zip(keys, h.tolist()):
                _LAG[k][0] = float(v)
    if len(_ABSMAX) + len(todo) > 2048:
        _ABSMAX.clear()
    for t, k in zip(todo, keys):
        _ABSMAX[_absmax_key(t)] = (_LAG[k][0] + slack, t)


def prefetch_absmax(tensors):
    """Compute and cache max |t| for every tensor of ``tensors`` not yet cached at its current version; one synchronisation
    (none inside a ``lagged_absmax`` context, see there)."""
    if lagged_absmax.active is not None:
        return _lagged_prefetch(list(tensors), lagged_absmax.active.slack)
    todo, keys = [], []
    for t in tensors:
        t = t.detach()
        k = _absmax_key(t)
        if k not in _ABSMAX and t.numel() > 0 and k not in keys:
            todo.append(t)
            keys.append(k)
    if not todo:
        return
    if len(_ABSMAX) + len(todo) > 2048:
        _ABSMAX.clear()
    vals = torch.stack(torch._foreach_norm([t.to(torch.float32) for t in todo], float('inf'))).tolist()
    for k, v, t in zip(keys, vals, todo):
        _ABSMAX[k] = (float(v), t)


def absmax(t):
    t = t.detach()
    k = _absmax_key(t)
    if k not in _ABSMAX:
        prefetch_absmax([t])
    ent = _ABSMAX.get(k)
    return ent[0] if ent is not None else 0.0


# ---- operand layouts as ONE gather: the layout code below runs once per shape on an index tensor (slot -> element of the
# flattened two-piece split, or the appended zero), and every later pack is cat + index_select ----
_LAYOUT = {}


def _split_gather(w, scale, key, build_index, what):
    """_layout_gather(_f16_split2(w, scale), key, build_index) -- as one launch (strive_pack_split_gather) when the library serves
    ``w``'s device: the cached index table is the layout, the split happens in the kernel (same bytes)."""
    lib = _pack_lib(w)
    if lib is None or os.environ.get('STRIVE_CHECK_PACKS') == '1':
        return _layout_gather(_f16_split2(w, scale, what), key, build_index)
    w = w.to(torch.float32).contiguous()
    shape = (2,) + tuple(w.shape)
    ck = (key, shape, str(w.device), 'i32')
    idx = _LAYOUT.get(ck)
    if idx is None:
        n = 2 * w.numel()
        pidx = torch.arange(n, dtype=torch.int64).view(shape)
        idx = build_index(pidx, n).reshape(-1).to(torch.int32).to(w.device)
        _LAYOUT[ck] = idx
    out = torch.empty((idx.numel(),), dtype=torch.float16, device=w.device)
    lib.call('strive_pack_split_gather', L.ptr(w), w.numel(), L.ptr(idx), idx.numel(), float(scale), L.ptr(out), L.stream_ptr(w))
    return out


def _layout_gather(pieces, key, build_index):
    """pieces: (2, ...) fp16.  build_index(pidx, zero) lays out an int64 tensor ``pidx`` of the same shape (zero = index of
    the appended 0) exactly like the fragment order; cached per (key, device)."""
    ck = (key, tuple(pieces.shape), str(pieces.device))
    idx = _LAYOUT.get(ck)
    if idx is None:
        n = pieces.numel()
        pidx = torch.arange(n, dtype=torch.int64).view(pieces.shape)
        idx = build_index(pidx, n).reshape(-1).to(pieces.device)
        _LAYOUT[ck] = idx
    flat = torch.cat([pieces.reshape(-1), torch.zeros((1,), dtype=pieces.dtype, device=pieces.device)])
    return flat.index_select(0, idx)


def _pack_lib(t):
    """The library for the one-launch pack kernels: device tensors (or the test emulation); host tensors without it keep the
    torch layout code below (same bytes)."""
    from . import ops
    try:
        return ops._lib_for(t)
    except L.StriveHipError:
        return None


def dense_fragments(a, scale):
    """(M, K) fp32 matrix -> int32 tensor of its two-piece fp16 split (scaled by ``scale``) in the operand order of
    v_mfma_f32_16x16x32_f16 (include/strive_hip.h StriveMLP.wf): [row tile][k-step][piece][lane][8 x fp16]."""
    M, K = a.shape
    MT, KS = (M + 15) // 16, (K + 31) // 32
    pieces = _f16_split2(a, scale, 'dense layer')                             # (2, M, K)

    def index(pidx, zero):
        pad = torch.full((2, MT * 16, KS * 32), zero, dtype=torch.int64)
        pad[:, :M, :K] = pidx
        return pad.view(2, MT, 16, KS, 4, 8).permute(1, 3, 0, 4, 2, 5).contiguous()      # (tile, step, piece, g, row, j)
    fr = _layout_gather(pieces, 'dense', index)
    return fr.view(torch.int16).view(-1, 8).contiguous().view(torch.int32)


def _fill_mlp(s, holder, sd, prefix):
    """sd keys ``<prefix>.net.{0,3,6,9}`` Linear, ``.net.{1,4,7}`` LayerNorm
    (reference src/models/common.py:26-39)."""
    k = 0
    dims = []
    ws_ = []
    while (prefix + '.net.%d.weight' % (3 * len(ws_))) in sd:
        ws_.append(_c(sd[prefix + '.net.%d.weight' % (3 * len(ws_))]))
    prefetch_absmax(ws_)
    while (prefix + '.net.%d.weight' % (3 * k)) in sd:
        w = ws_[k]
        b = _c(sd[prefix + '.net.%d.bias' % (3 * k)])
        if k == 0:
            dims.append(w.shape[1])
        dims.append(w.shape[0])
        s.w[k] = holder.hold(w)
        s.b[k] = holder.hold(b)
        # matrix-core operands (csrc/mlp_dev.h dense_mfma); STRIVE_DENSE_VALU=1 withholds them (A/B measurements: the
        # layers then run on the vector ALUs in fp32 from w / wt)
        frag = w.shape[0] >= 32 and w.shape[1] >= 32 and os.environ.get('STRIVE_DENSE_VALU', '0') != '1'
        sc = _pow2_scale(_checked_absmax(w, 'dense layer')) if frag else 1.0
        lib = _pack_lib(w)
        if lib is not None:
            # transpose + both fragment tables in ONE launch (strive_pack_dense)
            M, K = w.shape
            wt = torch.empty((K, M), dtype=torch.float32, device=w.device)
            s.wt[k] = holder.hold(wt)
            wf = wbf = None
            if frag:
                wf = torch.empty((((M + 15) // 16) * ((K + 31) // 32) * 512,), dtype=torch.int32, device=w.device)
                wbf = torch.empty((((K + 15) // 16) * ((M + 31) // 32) * 512,), dtype=torch.int32, device=w.device)
                s.wsc[k] = sc
                s.wf[k] = holder.hold(wf)
                s.wbf[k] = holder.hold(wbf)
            lib.call('strive_pack_dense', L.ptr(w), M, K, sc, L.ptr(wt), L.ptr(wf) if frag else None, L.ptr(wbf) if frag else None,
                     L.stream_ptr(w))
        else:
            s.wt[k] = holder.hold(w.t().contiguous())
            if frag:
                s.wsc[k] = sc
                s.wf[k] = holder.hold(dense_fragments(w, sc))
                s.wbf[k] = holder.hold(dense_fragments(w.t().contiguous(), sc))
        gk = prefix + '.net.%d.weight' % (3 * k + 1)
        if gk in sd:
            s.ln_g[k] = holder.hold(_c(sd[gk]))
            s.ln_b[k] = holder.hold(_c(sd[prefix + '.net.%d.bias' % (3 * k + 1)]))
        k += 1
    if k < 2 or k > L.MAXL:
        raise ValueError('MLP %s has %d Linear layers; the HIP kernels support 2..%d' % (prefix, k, L.MAXL))
    for d in dims[1:-1]:
        if d != 128:
            raise ValueError('MLP %s: hidden width %d, HIP kernels are built for 128' % (prefix, d))
    s.nlayers = k
    for i, d in enumerate(dims):
        s.dims[i] = d
    return dims


def pack_mlp(sd, prefix):
    p = Packed(L.StriveMLP())
    p.dims = _fill_mlp(p.struct, p, sd, prefix)
    return p


def _fill_gnn(s, holder, sd, prefix, NC):
    _fill_mlp(s.mlp_in, holder, sd, prefix + '.mlp_in')
    _fill_mlp(s.edge, holder, sd, prefix + '.msg.0.edge_mlp')
    _fill_mlp(s.update, holder, sd, prefix + '.msg.0.update_mlp')
    _fill_mlp(s.mlp_out, holder, sd, prefix + '.mlp_out')
    D = s.mlp_in.dims[s.mlp_in.nlayers]
    s.D = D
    s.NC = NC
    if s.edge.dims[0] != 2 * (D + NC) + 4 or s.update.dims[0] != 2 * D + NC:
        raise ValueError('GNN %s: unexpected edge/update input sizes' % prefix)
    if D not in (64, 128):
        raise ValueError('GNN %s: node size %d unsupported (64 or 128)' % (prefix, D))


def pack_gnn(sd, prefix, NC):
    p = Packed(L.StriveGNN())
    _fill_gnn(p.struct, p, sd, prefix, NC)
    return p


def _fill_gru(s, holder, sd, prefix):
    prefetch_absmax([_c(sd['%s.weight_%s_l%d' % (prefix, k, l)]) for l in range(3) for k in ('ih', 'hh')])
    for l in range(3):
        wih = _c(sd['%s.weight_ih_l%d' % (prefix, l)])
        whh = _c(sd['%s.weight_hh_l%d' % (prefix, l)])
        s.wih[l] = holder.hold(wih)
        s.whh[l] = holder.hold(whh)
        s.wih_t[l] = holder.hold(wih.t().contiguous())
        s.whh_t[l] = holder.hold(whh.t().contiguous())
        s.bih[l] = holder.hold(_c(sd['%s.bias_ih_l%d' % (prefix, l)]))
        s.bhh[l] = holder.hold(_c(sd['%s.bias_hh_l%d' % (prefix, l)]))
        # matrix-core operands of the scene-resident rollout kernels (csrc/scene_rollout.h): both fragment tables of every
        # 192 x 64 matrix (layer 0's 192 x 4 input matrix stays on the vector ALUs)
        if os.environ.get('STRIVE_DENSE_VALU', '0') == '1':
            continue
        for name, w in (('hh', whh), ('ih', wih)):
            if w.shape[1] < 32:
                continue
            sc = _pow2_scale(_checked_absmax(w, 'GRU'))
            M, K = w.shape
            lib = _pack_lib(w)
            if lib is not None:
                wf = torch.empty((((M + 15) // 16) * ((K + 31) // 32) * 512,), dtype=torch.int32, device=w.device)
                wbf = torch.empty((((K + 15) // 16) * ((M + 31) // 32) * 512,), dtype=torch.int32, device=w.device)
                lib.call('strive_pack_dense', L.ptr(w), M, K, sc, None, L.ptr(wf), L.ptr(wbf), L.stream_ptr(w))
            else:
                wf, wbf = dense_fragments(w, sc), dense_fragments(w.t().contiguous(), sc)
            getattr(s, 'w%s_f' % name)[l] = holder.hold(wf)
            getattr(s, 'w%s_bf' % name)[l] = holder.hold(wbf)
            getattr(s, '%s_sc' % name)[l] = sc


def _pow2_scale(bound, target=None):
    """largest power of two S with bound * S <= target (fp16's largest finite value is 65504); inside a lagged_absmax context
    the bound is one step old and the target leaves another factor 2"""
    import math
    if target is None:
        target = 16384.0 if lagged_absmax.active is not None else 32768.0
    if not (bound > 0.0) or not math.isfinite(bound):
        return 1.0
    return float(2.0 ** math.floor(math.log2(target / bound)))


def _checked_absmax(w, what):
    v = absmax(w)
    import math
    if not math.isfinite(v):
        raise ValueError('%s weights are not finite' % what)
    return v


def _f16_split2(w, scale, what):
    """w * scale = w0 + w1 up to 2^-24 |w|: w0 = fp16 (round to nearest even), w1 = fp16 of the exact remainder.  ``scale``
    comes from _pow2_scale(_checked_absmax(w)): max |w| scale <= 32768, so both pieces are finite.
    CONTRACT: max |w| is cached per (address, in-place version) of the parameter (_ABSMAX).  Parameters must therefore only be
    modified through operations that bump the version counter (optimiser steps, ``p.copy_()``, ``p.mul_()`` ...); an edit through
    ``p.data``, an external alias or a FUSED optimiser (torch._fused_adam_ does not bump it) must be followed by
    ``parameters_changed()`` (DataParallelTrainer.step does that) -- otherwise it leaves a stale bound, and a weight that grew past 2 x the bound would overflow its fp16 pieces
    to inf without an error.  STRIVE_CHECK_PACKS=1 verifies every packed piece on the device (one synchronisation per pack)."""
    ws = w.to(torch.float32) * scale
    w0 = ws.to(torch.float16)
    w1 = (ws - w0.to(torch.float32)).to(torch.float16)
    if os.environ.get('STRIVE_CHECK_PACKS') == '1' and not bool(torch.isfinite(w0).all() and torch.isfinite(w1).all()):
        raise ValueError('%s: fp16 operand pieces are not finite (stale max |w|? see _f16_split2)' % what)
    return torch.stack([w0, w1], dim=0)


def _conv1_fragments(w, scale):
    """(16,4,7,7) fp32 -> int32 tensor holding [ky][piece][lane][8 x fp16]: the two-piece fp16 split of every (scaled)
    weight in the k order of conv1b_kernel (lane group g: window columns 2g, 2g+1; element = parity*4 + layer;
    column 7 is zero padding)."""
    def index(pidx, zero):                                           # pidx: (2, co, ci, ky, kx) slots of the two-piece split
        pad = torch.full((2, 16, 4, 7, 1), zero, dtype=torch.int64)
        pk = torch.cat([pidx, pad], dim=4).view(2, 16, 4, 7, 4, 2)   # kx -> (g, parity)
        return pk.permute(3, 0, 4, 1, 5, 2).contiguous()             # [ky][piece][g][co][parity][ci] = (7, 2, 4, 16, 2, 4)
    frag = _split_gather(w, scale, 'conv1', index, 'conv1')
    return frag.view(torch.int16).view(7, 2, 64, 8).contiguous().view(torch.int32)


BF6_PASS_CH = 8     # input channels staged per pass by conv_bf6_kernel (BfCfg::PASS_CH)


def conv_tap_order(k=5):
    """Window taps (ky, kx) held by lane halves h = 0, 1 of MFMA step s in conv_bf6_kernel (None = zero-weight slot).
    5x5: steps 0-9 = row s // 2, columns (s & 1) + 2h; steps 10-11 = column 4 of rows 2 (s - 10) + h; step 12 = (4, 4).
    3x3: steps 0-2 = row s, columns 0 and 2; step 3 = column 1 of rows 0, 1; step 4 = (2, 1)."""
    if k == 5:
        order = [[(s_ >> 1, (s_ & 1) + 2 * h) for h in range(2)] for s_ in range(10)]
        order += [[(0, 4), (1, 4)], [(2, 4), (3, 4)], [(4, 4), None]]
    elif k == 3:
        order = [[(s_, 0), (s_, 2)] for s_ in range(3)]
        order += [[(0, 1), (1, 1)], [(2, 1), None]]
    else:
        raise NotImplementedError('conv_bf6_kernel has tap orders for 5x5 and 3x3 windows')
    return order


def _conv_bf6_fragments(w, scale, pass_ch=BF6_PASS_CH):
    """(co, ci, k, k) fp32, k = 5 or 3 -> int32 tensor [pass = ci/8][step][co/32][piece 2][lane 64][8 x fp16]: the two-piece
    fp16 split of every (scaled) weight in the k order of conv_bf6_kernel (lane = 32 h + output channel % 32, element e =
    input channel 8 pass + e, window tap = conv_tap_order(k)[step][h])."""
    co, ci, k, _ = w.shape
    if pass_ch != 8:
        raise NotImplementedError('conv_bf6_kernel stages 8 channels at a time')
    npass, csplit = ci // pass_ch, co // 32
    order = conv_tap_order(k)

    def index(pidx, zero):
        pi = pidx.permute(0, 1, 3, 4, 2).contiguous()                                # (2, co, ky, kx, ci)
        out = torch.full((npass, len(order), csplit, 2, 2, 32, 8), zero, dtype=torch.int64)
        for p_ in range(npass):
            for s_, taps in enumerate(order):
                for h, tap in enumerate(taps):
                    if tap is None:
                        continue
                    ky, kx = tap
                    blk = pi[:, :, ky, kx, pass_ch * p_:pass_ch * p_ + 8]            # (2, co, 8)
                    out[p_, s_, :, :, h] = blk.reshape(2, csplit, 32, 8).permute(1, 0, 2, 3)
        return out
    out = _split_gather(w, scale, 'bf6', index, 'conv')                              # slots of the (2, co, ci, ky, kx) split
    return out.view(torch.int16).view(-1, 8).contiguous().view(torch.int32)


def _fill_cnn(s, holder, sd):
    expect = [(16, 4, 7), (32, 16, 5), (64, 32, 5), (64, 64, 3), (128, 64, 3), (128, 128, 3)]
    for l in range(6):
        w = _c(sd['map_conv.%d.weight' % (3 * l)])
        co, ci, k = expect[l]
        if tuple(w.shape) != (co, ci, k, k):
            raise NotImplementedError('the HIP map CNN implements the reference default architecture only '
                                      '(layer %d weight %s, expected %s)' % (l, tuple(w.shape), (co, ci, k, k)))
        s.w[l] = holder.hold(w)                                           # (not read by the kernels any more, strive_hip.h)
        s.w_torch[l] = holder.hold(w)                                     # (co, ci, ky, kx): the training backward's data gradient
        s.b[l] = holder.hold(_c(sd['map_conv.%d.bias' % (3 * l)]))
        s.gn_g[l] = holder.hold(_c(sd['map_conv.%d.weight' % (3 * l + 1)]))
        s.gn_b[l] = holder.hold(_c(sd['map_conv.%d.bias' % (3 * l + 1)]))
    # power-of-two scales: weights so that max |w| * wscale <= 32768; layer inputs (l >= 1) from the bound
    # relu(gamma * xhat + beta) <= max|gamma| * sqrt(C H W) + max|beta|  (|xhat| <= sqrt(#elements) for any sample)
    in_elems = [0, 16 * 125 * 125, 32 * 61 * 61, 64 * 29 * 29, 64 * 14 * 14, 128 * 6 * 6]
    prefetch_absmax([_c(sd['map_conv.%d.weight' % (3 * l)]) for l in range(6)] +
                    [_c(sd['map_conv.%d.%s' % (3 * l - 2, n)]) for l in range(1, 6) for n in ('weight', 'bias')])
    for l in range(6):
        w = _c(sd['map_conv.%d.weight' % (3 * l)])
        s.wscale[l] = _pow2_scale(_checked_absmax(w, 'conv'))
        if l == 0:
            s.xscale[l] = 1.0
        else:
            g = absmax(_c(sd['map_conv.%d.weight' % (3 * l - 2)]))
            b = absmax(_c(sd['map_conv.%d.bias' % (3 * l - 2)]))
            s.xscale[l] = min(_pow2_scale(g * in_elems[l] ** 0.5 + b, 30000.0 if lagged_absmax.active is not None else 60000.0), 1024.0)
    s.w1_frag = holder.hold(_conv1_fragments(_c(sd['map_conv.0.weight']), s.wscale[0]))
    s.w2_frag = holder.hold(_conv_bf6_fragments(_c(sd['map_conv.3.weight']), s.wscale[1]))
    s.w3_frag = holder.hold(_conv_bf6_fragments(_c(sd['map_conv.6.weight']), s.wscale[2]))
    s.w4_frag = holder.hold(_conv_bf6_fragments(_c(sd['map_conv.9.weight']), s.wscale[3]))
    s.w5_frag = holder.hold(_conv_bf6_fragments(_c(sd['map_conv.12.weight']), s.wscale[4]))
    s.w6_frag = holder.hold(_conv_bf6_fragments(_c(sd['map_conv.15.weight']), s.wscale[5]))
    fw = _c(sd['map_feature.weight'])
    if tuple(fw.shape) != (64, 512):
        raise NotImplementedError('map_feature must be Linear(512, 64)')
    s.fc_wt = holder.hold(fw.t().contiguous())
    s.fc_b = holder.hold(_c(sd['map_feature.bias']))


def pack_cnn(sd):
    p = Packed(L.StriveCNN())
    _fill_cnn(p.struct, p, sd)
    return p


def _fill_map(s, holder, map_env, device, with_px4=True):
    raster = map_env.nusc_raster
    dx = map_env.nusc_dx
    if raster.dtype != torch.uint8 or dx.dtype != torch.float64:
        raise TypeError('map raster must be uint8 and nusc_dx float64 (reference src/datasets/map_env.py:165-166)')
    raster = raster.to(device).contiguous()
    dx = dx.to(device).contiguous()
    M, Cc, H, W = raster.shape
    s.raster = holder.hold(raster)
    s.dx = holder.hold(dx)
    s.M, s.C, s.H, s.W = M, Cc, H, W
    b = map_env.bounds
    # fp32 linspace tables, generated exactly like the reference does (torch.linspace, fp32)
    s.lwise = holder.hold(torch.linspace(b[0], b[2], map_env.L).to(device))
    s.wwise = holder.hold(torch.linspace(b[1], b[3], map_env.W).to(device))
    s.L, s.Wc = map_env.L, map_env.W
    if Cc == 4 and with_px4:
        # pixel-interleaved copy of the raster (one 32-bit word per pixel) for the fused crop -> conv1 gather
        s.raster_px4 = holder.hold(raster.permute(0, 2, 3, 1).contiguous().view(torch.int32).view(M, H, W))


def pack_map(map_env, device):
    p = Packed(L.StriveMap())
    _fill_map(p.struct, p, map_env, device)
    return p


def pack_decoder(sd, NC, map_env, device, state_norm, att_norm, bike, cnn=None, map_pack=None):
    """state_norm / att_norm: objects with mean_vals/std_vals (MeanStdNormalizer API); bike: dict; ``cnn``: an up-to-date
    pack_cnn() of the same parameters to share (its tensors are kept alive by the returned pack) instead of packing the map
    CNN a second time; ``map_pack``: likewise a pack_map() of (map_env, device) -- the training step rebuilds the decoder pack
    after every optimiser step, and building the map pack again (interleaved raster copy, two host-to-device table uploads)
    was HALF of that step's host time (16 of 33 ms, profiles/r04_train_host_cprofile.txt)."""
    p = Packed(L.StriveDecoder())
    s = p.struct
    _fill_gnn(s.gnn, p, sd, 'decoder_net', NC)
    _fill_gru(s.gru, p, sd, 'decoder_memory')
    if cnn is not None:
        C.memmove(C.byref(s.cnn), C.byref(cnn.struct), C.sizeof(L.StriveCNN))
        p.keep.append(cnn)
    else:
        _fill_cnn(s.cnn, p, sd)
    if map_pack is not None:
        C.memmove(C.byref(s.map), C.byref(map_pack.struct), C.sizeof(L.StriveMap))
        p.keep.append(map_pack)
    else:
        _fill_map(s.map, p, map_env, device)
    for i in range(6):
        s.state_mean[i] = float(state_norm.mean_vals[i])
        s.state_std[i] = float(state_norm.std_vals[i])
    for i in range(2):
        s.att_mean[i] = float(att_norm.mean_vals[i])
        s.att_std[i] = float(att_norm.std_vals[i])
    s.a_mean, s.a_std = bike['a_stats']
    s.ddh_mean, s.ddh_std = bike['ddh_stats']
    s.dt = bike['dt']
    s.max_hdot = bike['maxhdot']
    s.max_s = bike['maxs']
    s.scene_par = p.hold(_scene_par(sd, NC).to(device))
    return p


def _scene_par(sd, NC):
    """The small decoder parameters in the order of csrc/scene_rollout.h Par (include/strive_hip.h StriveDecoder.scene_par):
    what the scene-resident rollout kernels keep in LDS."""
    def t(k):
        return _c(sd[k]).reshape(-1)

    def mlp(prefix, nlin):
        bs = [t('%s.net.%d.bias' % (prefix, 3 * k)) for k in range(nlin)]
        ln = [(t('%s.net.%d.weight' % (prefix, 3 * k + 1)), t('%s.net.%d.bias' % (prefix, 3 * k + 1))) for k in range(nlin - 1)]
        return bs, ln
    out = []
    b, ln = mlp('decoder_net.mlp_in', 3)
    out += [b[0], b[1], b[2], ln[0][0], ln[0][1], ln[1][0], ln[1][1]]
    b, ln = mlp('decoder_net.msg.0.edge_mlp', 3)
    w0 = _c(sd['decoder_net.msg.0.edge_mlp.net.0.weight'])                    # (128, 132 + 2 NC)
    wrel_t = w0[:, 128 + 2 * NC:128 + 2 * NC + 4].t().contiguous().reshape(-1)  # (4, 128)
    out += [b[0], b[1], b[2], ln[0][0], ln[0][1], ln[1][0], ln[1][1], wrel_t]
    b, ln = mlp('decoder_net.msg.0.update_mlp', 2)
    out += [b[0], b[1], ln[0][0], ln[0][1]]
    b, ln = mlp('decoder_net.mlp_out', 3)
    pad2 = torch.zeros((2,), dtype=torch.float32, device=b[2].device)
    out += [b[0], b[1], b[2], pad2, ln[0][0], ln[0][1], ln[1][0], ln[1][1], t('decoder_net.mlp_out.net.6.weight')]
    out += [t('decoder_memory.bias_ih_l%d' % l) for l in range(3)] + [t('decoder_memory.bias_hh_l%d' % l) for l in range(3)]
    wih0 = _c(sd['decoder_memory.weight_ih_l0'])                               # (192, 4)
    out += [wih0.t().contiguous().reshape(-1), wih0.reshape(-1)]
    blk = torch.cat(out)
    assert blk.numel() == 6340, 'scene_par layout: %d floats' % blk.numel()
    return blk.contiguous()


def pack_scenes(ptr, NS, device):
    """ptr: (B+1,) long tensor of scene offsets -> StriveScenes (int32 device tensors)."""
    p = Packed(L.StriveScenes())
    ptr32 = ptr.to(device=device, dtype=torch.int32).contiguous()
    sizes = (ptr[1:] - ptr[:-1]).to('cpu')
    B = sizes.shape[0]
    scene_of = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), sizes).to(device).contiguous()
    p.struct.NA = int(ptr[-1])
    p.struct.NS = int(NS)
    p.struct.B = int(B)
    p.struct.max_n = int(sizes.max()) if B > 0 else 0
    p.struct.n_edges = int((sizes.to(torch.int64) * (sizes.to(torch.int64) - 1)).sum()) * int(NS) if B > 0 else 0
    p.struct.ptr = p.hold(ptr32)
    p.struct.scene_of = p.hold(scene_of)
    p.sizes = sizes
    return p
