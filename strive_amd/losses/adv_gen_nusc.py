"""Optimisation-time losses with the reference's names, signatures and returned dict keys
(reference src/losses/adv_gen_nusc.py:14-512, 625-673).

The two expensive terms run in HIP kernels through strive_amd.ops:
  * ``VehCollLoss``  -- strive_veh_coll_fwd/bwd over in-scene ordered pairs only (the reference enumerates
    all NA^2 pairs of the batch and masks cross-scene ones afterwards);
  * ``EnvCollLoss``  -- strive_coll_point (raster gather) for the collision point.
Everything else (interpolation, prior NLL, soft-min bookkeeping, means) is elementwise torch glue on the
device.  Per-scene Python loops and ``print`` calls of the reference's AdvGenLoss are replaced by segment
operations; returned values are the same.
"""
import numpy as np
import torch
from torch import nn

from .common import log_normal
from .. import ops


class LossDict(dict):
    """The reference's loss dicts hold, besides the scalar ``'loss'``, the COMPACTED 1-D lists of the penalties currently
    in collision (``pen[mask]``): tensors whose length depends on the data, so producing them costs a device->host
    synchronisation.  The optimisation loops only back-propagate ``'loss'``; the lists are read for logging.  This dict has
    the same keys and yields the same tensors, but compacts a list only when it is actually read -- a closure that reads
    nothing but ``'loss'`` never stalls the GPU queue.  ``'loss'`` is the differentiable entry; the lazily produced entries are
    evaluated without autograd on a snapshot of the closure's inputs.  Bulk access (iteration, ``dict(ld)``, ``copy``, ``pop``,
    pickling) resolves every entry first."""

    def __init__(self):
        super(LossDict, self).__init__()
        self._thunks = {}

    def set_lazy(self, key, thunk):
        dict.__setitem__(self, key, None)
        self._thunks[key] = thunk

    def _resolve(self, key):
        th = self._thunks.pop(key, None)
        if th is not None:
            dict.__setitem__(self, key, th())

    def __getitem__(self, key):
        self._resolve(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key in self:
            return self[key]
        return default

    def items(self):
        for k in list(self._thunks):
            self._resolve(k)
        return dict.items(self)

    def values(self):
        for k in list(self._thunks):
            self._resolve(k)
        return dict.values(self)

    # bulk access through CPython's dict fast paths (dict(ld), {**ld}, ld.copy(), other.update(ld), pop, pickling) bypasses
    # __getitem__: resolve everything first so that no None placeholder can leak out
    def _resolve_all(self):
        for k in list(self._thunks):
            self._resolve(k)

    def __iter__(self):
        self._resolve_all()
        return dict.__iter__(self)

    def keys(self):
        self._resolve_all()
        return dict.keys(self)

    def copy(self):
        self._resolve_all()
        return dict(dict.items(self))

    def pop(self, key, *default):
        self._resolve(key)
        return dict.pop(self, key, *default)

    def __reduce__(self):
        self._resolve_all()
        return (dict, (dict(dict.items(self)),))


def _masked_mean(pen, mask):
    """mean of pen[mask], 0 when nothing is selected (= the reference's ``[0.]`` sentinel), without compaction."""
    cnt = mask.sum()
    return torch.where(mask, pen, torch.zeros_like(pen)).sum() / torch.clamp(cnt, min=1).to(pen.dtype)


def _setup_signature(weights, init_z):
    """what a fused-loss descriptor was built from, cheap to compare on every call (no device access)"""
    zsig = None if init_z is None else (init_z.data_ptr(), init_z._version, tuple(init_z.shape))
    return (tuple(sorted((k, float(v)) for k, v in weights.items())), zsig)


def _compact_or_zero(pen, mask):
    v = pen[mask]
    if v.numel() == 0:
        return torch.Tensor([0.0]).to(pen.device)
    return v


def interp_traj(future_pred, scale_factor=3):
    """Linear up-sampling in time + heading renormalisation (reference :625-644)."""
    multi = future_pred.dim() == 4
    if multi:
        NA, NS, T, _ = future_pred.size()
        future_pred = future_pred.reshape(NA * NS, T, 4)
    if float(scale_factor) != int(scale_factor) or int(scale_factor) < 1:
        raise NotImplementedError('interp_traj: integer scale factors only')
    up = ops.interp_traj(future_pred, int(scale_factor))
    if multi:
        up = up.reshape(NA, NS, up.size(1), 4)
    return up


def _expand_targets(tgt, sizes_minus_one):
    """Repeat each scene's target row for that scene's non-ego agents."""
    return torch.repeat_interleave(tgt, sizes_minus_one.to(tgt.device), dim=0)


def _behind(attacker_fut, tgt_expanded, crash_min_infront):
    d = attacker_fut[:, :, :2] - tgt_expanded[:, :, :2]
    d = d / torch.norm(d, dim=-1, keepdim=True)
    return torch.sum(d * tgt_expanded[:, :, 2:4], dim=-1) < crash_min_infront


def check_behind(attacker_fut, tgt_fut, ptr, crash_min_infront):
    """(NA-B,T) bool, True where the attacker is behind the target (reference :646-673)."""
    sizes = (ptr[1:] - ptr[:-1]) - 1
    return _behind(attacker_fut, _expand_targets(tgt_fut, sizes), crash_min_infront)


class MotionPriorLoss(nn.Module):
    """Negative log-likelihood of z under the prior (reference :343-364)."""

    def forward(self, z, prior_out):
        mu, var = prior_out[0], prior_out[1]
        if z.dim() == 3:
            mu, var = mu.unsqueeze(1), var.unsqueeze(1)
        return -log_normal(z, mu, var)


class TgtMatchingLoss(nn.Module):
    """(reference :14-51) -- keeps the reference's behaviour that the prior term adds
    ``w * tgt_loss.mean()`` to the objective while reporting the prior NLL (line 46)."""

    def __init__(self, loss_weights):
        super(TgtMatchingLoss, self).__init__()
        self.loss_weights = loss_weights
        self.motion_prior_loss = MotionPriorLoss()

    def forward(self, future_pred, tgt_traj, z, prior_out, scene_alive=None):
        """``scene_alive`` (B,) uint8 / bool on the device or None (not in the reference): scenes with 0 have left the batch --
        their target may be NaN (a failed planner rollout), they add nothing to the mean, their rows get zero gradients and the
        mean's denominator counts the alive scenes only, i.e. the other scenes see the batch rebuilt without them."""
        out = LossDict()
        loss = 0.0
        tgt_loss = None
        alive = None
        if scene_alive is not None:
            alive = scene_alive.to(torch.bool).view(-1, *([1] * (future_pred.dim() - 1)))
            tgt_traj = torch.where(alive, tgt_traj, future_pred.detach())          # (no NaN may reach the subtraction's backward)

        def mean(v):
            if alive is None:
                return v.mean()
            per_scene = v[0].numel()
            return v.sum() / torch.clamp(scene_alive.sum() * per_scene, min=1).to(v.dtype)
        if self.loss_weights['match_ext'] > 0.0:
            tgt_loss = torch.sum((future_pred - tgt_traj) ** 2, dim=-1)
            loss = loss + self.loss_weights['match_ext'] * mean(tgt_loss)
            out['match_ext_loss'] = tgt_loss
        if self.loss_weights['motion_prior_ext'] > 0.0:
            # reported but never part of the objective (reference :46 adds the matching term again): evaluated when read
            out.set_lazy('motion_prior_ext_loss', lambda: self.motion_prior_loss(z, prior_out))
            loss = loss + self.loss_weights['motion_prior_ext'] * mean(tgt_loss)
        out['loss'] = loss
        return out


def _linspace5(lo, hi):
    """torch.linspace(lo, hi, 5) for tensors of endpoints, with linspace's own evaluation order
    (first half from the start, second half from the end) so values match the reference's
    per-agent ``torch.linspace(cent_min.item(), cent_max.item(), 5)`` (reference :435)."""
    step = (hi - lo) / 4.0
    return torch.stack([lo, lo + step, hi - step * 2.0, hi - step, hi], dim=1)


class VehCollLoss(nn.Module):
    """Circle-approximation vehicle collision penalty (reference :405-512)."""

    def __init__(self, veh_att, num_circ=5, buffer_dist=0.0, single_veh_idx=None, ptr=None):
        super(VehCollLoss, self).__init__()
        if num_circ != 5:
            raise NotImplementedError('the HIP collision kernel is built for 5 circles per vehicle')
        self.veh_att = veh_att
        self.buffer_dist = buffer_dist
        self.single_veh_idx = single_veh_idx
        NA = veh_att.size(0)
        dev = veh_att.device
        if ptr is None:
            ptr = torch.tensor([0, NA], dtype=torch.long)
        self.ptr = ptr
        self.info = ops.SceneInfo(ptr.cpu(), dev)
        self.veh_rad = veh_att[:, 1] / 2.
        cent_min = -(veh_att[:, 0] / 2.) + self.veh_rad
        cent_max = (veh_att[:, 0] / 2.) - self.veh_rad
        self.cent_x = _linspace5(cent_min, cent_max)
        self.num_circ = num_circ
        self.setup = ops.VehCollSetup(self.info, self.cent_x, self.veh_rad, buffer_dist)
        # slot bookkeeping: slot = pair_off[i] + (j - ptr[scene(i)])
        sizes = self.info.sizes.to(torch.long)
        agent_scene = torch.repeat_interleave(torch.arange(self.info.B), sizes)
        lo = ptr.cpu()[:-1][agent_scene]
        n_of_agent = sizes[agent_scene]
        self.slot_i = torch.repeat_interleave(torch.arange(NA), n_of_agent).to(dev)               # (P,)
        within = torch.arange(self.info.P) - torch.repeat_interleave(self.info.pair_off.cpu().to(torch.long), n_of_agent)
        self.slot_j = (torch.repeat_interleave(lo, n_of_agent) + within).to(dev)                   # (P,)
        valid = self.slot_i != self.slot_j
        if single_veh_idx is not None:
            sel = torch.zeros((NA,), dtype=torch.bool, device=dev)
            sel[(ptr.cpu()[:-1] + single_veh_idx).to(dev)] = True
            valid = valid & (sel[self.slot_i] | sel[self.slot_j])
        self.valid = valid

    def forward(self, traj, att_inds=None, return_raw=False):
        """traj (NA,T,4) UNNORMALISED.  Returns the 1-D penalties of colliding valid pairs in the reference's
        (t, i, j) order, ``[0.]`` if none; with ``return_raw`` the dense (T,NA,NA) penalty matrix and mask
        (cross-scene entries: penalty 0, mask False)."""
        if att_inds is not None:
            raise NotImplementedError('att_inds is not supported by the HIP collision kernel')
        pen, hit = ops.veh_coll_penalties(traj, self.setup)
        mask = hit.bool() & self.valid.view(1, -1)
        if return_raw:
            T, NA = pen.size(0), traj.size(0)
            dense = torch.zeros((T, NA, NA), dtype=pen.dtype, device=pen.device)
            dmask = torch.zeros((T, NA, NA), dtype=torch.bool, device=pen.device)
            dense[:, self.slot_i, self.slot_j] = pen
            dmask[:, self.slot_i, self.slot_j] = mask
            return dense, dmask
        v = pen[mask]                       # (one host sync: the compaction itself)
        if v.numel() == 0:
            return torch.Tensor([0.0]).to(traj.device)
        return v

    def block_penalties(self, traj):
        """(pen (T,P), colliding&valid mask (T,P)) in slot layout -- what the fused loss modules use."""
        pen, hit = ops.veh_coll_penalties(traj, self.setup)
        return pen, hit.bool() & self.valid.view(1, -1)


class EnvCollLoss(nn.Module):
    """Off-road penalty from the estimated collision point (reference :366-403)."""

    def __init__(self, veh_att, mapixes, map_env):
        super(EnvCollLoss, self).__init__()
        self.map_env = map_env
        self.mapixes = mapixes
        self.penalty_dists = torch.sqrt((veh_att[:, 0] ** 2 / 4.0) + (veh_att[:, 1] ** 2 / 4.0))
        self.veh_att = veh_att
        self._grid = {}

    def _grid_size(self, NA, T):
        # batch-mean size over the (NA*T, 2) expanded attributes, like nuscenes_utils.py:351-354; constant per T
        g = self._grid.get(T)
        if g is None:
            att = self.veh_att.view(NA, 1, 2).expand(NA, T, 2).reshape(NA * T, 2)
            mdx = torch.mean(self.map_env.nusc_dx) * 0.5
            mlw = torch.mean(att, dim=0)
            g = (torch.round(mlw[0] / mdx).int().item(), torch.round(mlw[1] / mdx).int().item())
            self._grid[T] = g
        return g

    def valid_penalties(self, traj):
        """(pen (NA*T,), valid (NA*T,) bool): penalty 1 - |c - p|/r for rows with a collision point."""
        NA, T, _ = traj.size()
        flat = traj.reshape(NA * T, 4)
        att = self.veh_att.view(NA, 1, 2).expand(NA, T, 2).reshape(NA * T, 2)
        mix = self.mapixes.view(NA, 1).expand(NA, T).reshape(NA * T)
        gl, gw = self._grid_size(NA, T)
        pt, _ = ops.coll_point(self.map_env, flat.detach(), att, mix, gl, gw)
        valid = ~torch.isnan(torch.sum(pt, dim=1))
        safe_pt = torch.where(valid.unsqueeze(1), pt, flat[:, :2].detach() + 1.0)
        d = torch.norm(flat[:, :2] - safe_pt, dim=1)
        pdist = self.penalty_dists.view(NA, 1).expand(NA, T).reshape(NA * T)
        return 1.0 - (d / pdist), valid

    def forward(self, traj):
        pen, valid = self.valid_penalties(traj)
        v = pen[valid]
        if v.numel() == 0:
            return torch.Tensor([0.0]).to(traj.device)
        return v


class AvoidCollLoss(nn.Module):
    """(reference :264-341)"""

    def __init__(self, loss_weights, veh_att, mapixes, map_env, init_z, veh_coll_buffer=0.0, single_veh_idx=None, ptr=None):
        super(AvoidCollLoss, self).__init__()
        self.loss_weights = loss_weights
        self.init_z = init_z
        self.single_veh_idx = single_veh_idx
        self.ptr = ptr
        self.use_single_agt = single_veh_idx is not None
        self.motion_prior_loss = MotionPriorLoss()
        self.veh_coll_loss = VehCollLoss(veh_att, buffer_dist=veh_coll_buffer, single_veh_idx=single_veh_idx, ptr=ptr)
        if self.use_single_agt:
            assert ptr is not None
            self.single_mask = torch.zeros((veh_att.size(0),), dtype=torch.bool, device=veh_att.device)
            self.single_mask[(ptr[:-1] + single_veh_idx).to(veh_att.device)] = True
            self.single_idx = torch.nonzero(self.single_mask).flatten()
            veh_att = veh_att[self.single_mask]
            mapixes = mapixes[self.single_mask]
        self.env_coll_loss = EnvCollLoss(veh_att, mapixes, map_env)
        self._fused = None
        self._fused_sig = None

    def _setup(self):
        sig = _setup_signature(self.loss_weights, self.init_z)
        if self._fused is None or self._fused_sig != sig:          # (weights or init_z edited after construction: rebuild)
            self._fused_sig = sig
            w = self.loss_weights
            env = self.env_coll_loss
            NA = self.veh_coll_loss.veh_att.size(0)
            dev = self.veh_coll_loss.veh_att.device
            env_agent = self.single_idx if self.use_single_agt else torch.arange(NA, device=dev)
            NE = env_agent.numel()
            self._fused = ops.AvoidCollSetup(
                self.veh_coll_loss.info, self.veh_coll_loss.setup, self.veh_coll_loss.valid, env_agent, env.veh_att, env.mapixes,
                env.penalty_dists, env.map_env, lambda TO: env._grid_size(NE, TO), self.init_z,
                (w['coll_veh'], w['coll_env'], w['motion_prior'], w['init_z']))
        return self._fused

    def forward(self, future_pred, z, prior_out):
        """One HIP call forward, one backward (strive_avoid_coll_fwd/bwd) for the objective; the per-term entries of the
        reference's dict are evaluated by ``forward_terms`` only when somebody reads them (logging)."""
        if future_pred.dim() != 3 or not (z.dim() == 2 or (z.dim() == 3 and z.size(1) == 1)) or prior_out[0].dim() != 2 or \
                prior_out[0].requires_grad or prior_out[1].requires_grad:
            return self.forward_terms(future_pred, z, prior_out)
        out = LossDict()
        loss, _ = ops.avoid_coll_loss(future_pred, z, prior_out[0], prior_out[1], self._setup())
        out['loss'] = loss
        terms = {}

        # the logging entries are evaluated on a snapshot of the closure's inputs (the optimiser updates z in place afterwards) and
        # without building a second autograd graph: 'loss' is the differentiable entry
        fp_s, z_s = future_pred.detach(), z.detach().clone()

        def term(key):
            def thunk():
                if not terms:
                    with torch.no_grad():
                        ft = self.forward_terms(fp_s, z_s, prior_out)
                        for k in list(ft.keys()):
                            terms[k] = ft[k]          # (resolves the lazy lists of that dict)
                return terms[key]
            return thunk
        w = self.loss_weights
        for key, wk in (('coll_veh_loss', 'coll_veh'), ('coll_env_loss', 'coll_env'), ('motion_prior_loss', 'motion_prior'),
                        ('init_loss', 'init_z')):
            if w[wk] > 0.0:
                out.set_lazy(key, term(key))
        return out

    def forward_terms(self, future_pred, z, prior_out):
        """The same objective term by term with torch glue between the HIP kernels (reference :290-341)."""
        w = self.loss_weights
        loss = 0.0
        out = LossDict()
        fine = interp_traj(future_pred, scale_factor=3)
        if w['coll_veh'] > 0.0:
            pen, m = self.veh_coll_loss.block_penalties(fine)
            loss = loss + w['coll_veh'] * _masked_mean(pen, m)
            out.set_lazy('coll_veh_loss', lambda: _compact_or_zero(pen, m))
        if w['coll_env'] > 0.0:
            epen, ev = self.env_coll_loss.valid_penalties(fine if not self.use_single_agt else fine.index_select(0, self.single_idx))
            loss = loss + w['coll_env'] * _masked_mean(epen, ev)
            out.set_lazy('coll_env_loss', lambda: _compact_or_zero(epen, ev))
        if w['motion_prior'] > 0.0:
            p = self.motion_prior_loss(z, prior_out)
            loss = loss + w['motion_prior'] * p.mean()
            out['motion_prior_loss'] = p
        if w['init_z'] > 0.0:
            i = torch.sum((self.init_z - z) ** 2, dim=1)
            loss = loss + w['init_z'] * i.mean()
            out['init_loss'] = i
        out['loss'] = loss
        return out


class AdvGenLoss(nn.Module):
    """Adversarial objective (reference :53-262)."""

    def __init__(self, loss_weights, veh_att, mapixes, map_env, init_z, ptr, veh_coll_buffer=0.0,
                 crash_loss_min_time=0, crash_loss_min_infront=None):
        super(AdvGenLoss, self).__init__()
        dev = veh_att.device
        self.loss_weights = loss_weights
        self.init_z = init_z
        self.motion_prior_loss = MotionPriorLoss()
        self.ptr = ptr
        ptr_c = ptr.cpu()
        self.graph_sizes = ptr_c[1:] - ptr_c[:-1]
        NA = veh_att.size(0)
        self.B = self.graph_sizes.size(0)
        self.ego_mask = torch.zeros((NA,), dtype=torch.bool, device=dev)
        self.ego_mask[ptr_c[:-1].to(dev)] = True
        # index tensors instead of boolean-mask indexing in forward(): x[mask] has a data-dependent shape and costs a
        # device->host synchronisation on every call
        self.nonego_idx = torch.nonzero(~self.ego_mask).flatten()
        self.nonego_ptr = ptr_c - torch.arange(len(ptr_c))
        self.veh_coll_loss = VehCollLoss(veh_att, buffer_dist=veh_coll_buffer, ptr=ptr)
        self.env_coll_loss = EnvCollLoss(veh_att[~self.ego_mask], mapixes[~self.ego_mask], map_env)
        self.crash_min_t = crash_loss_min_time
        self.crash_min_infront = crash_loss_min_infront
        if crash_loss_min_infront is not None:
            assert -1 <= crash_loss_min_infront <= 1
        # scene id of every non-ego agent; ego involvement of every pair slot
        self.seg = torch.repeat_interleave(torch.arange(self.B), self.graph_sizes - 1).to(dev)      # (NA-B,)
        vl = self.veh_coll_loss
        self.slot_ego = self.ego_mask[vl.slot_i] | self.ego_mask[vl.slot_j]
        nonego_index = torch.cumsum((~self.ego_mask).to(torch.long), 0) - 1
        self.slot_i_ne, self.slot_j_ne = nonego_index[vl.slot_i], nonego_index[vl.slot_j]
        self.slot_i_ego, self.slot_j_ego = self.ego_mask[vl.slot_i], self.ego_mask[vl.slot_j]
        self.slot_scene = torch.repeat_interleave(torch.arange(self.B), self.graph_sizes).to(dev)[vl.slot_i]      # (P,) scene of a pair slot
        self._fused = None
        self._fused_sig = None

    def _segment_softmin(self, din):
        """softmin over all (agent, t) entries of each scene; all-inf scenes give zeros (reference :133-135)."""
        NT = din.size(1)
        seg = self.seg.view(-1, 1).expand_as(din)
        neg = -din
        mx = torch.full((self.B,), float('-inf'), device=din.device).scatter_reduce(0, seg.reshape(-1), neg.reshape(-1),
                                                                                    reduce='amax', include_self=True)
        e = torch.exp(neg - mx[self.seg].view(-1, 1))
        den = torch.zeros((self.B,), device=din.device).scatter_add(0, seg.reshape(-1), e.reshape(-1))
        soft = e / den[self.seg].view(-1, 1)
        return torch.where(torch.isnan(soft), torch.zeros_like(soft), soft)

    def _setup(self):
        sig = _setup_signature(self.loss_weights, self.init_z)
        if self._fused is None or self._fused_sig != sig:
            self._fused_sig = sig
            w = self.loss_weights
            env, vl = self.env_coll_loss, self.veh_coll_loss
            NE = self.nonego_idx.numel()
            base = ops.AvoidCollSetup(
                vl.info, vl.setup, vl.valid, self.nonego_idx, env.veh_att, env.mapixes, env.penalty_dists, env.map_env,
                lambda TO: env._grid_size(NE, TO), self.init_z,
                (w.get('coll_veh', 0.0), w.get('coll_env', 0.0), w.get('motion_prior', 0.0), w.get('init_z', 0.0)))
            slot_ne = torch.where(self.slot_ego, torch.where(self.slot_j_ego, self.slot_i_ne, self.slot_j_ne).clamp(min=0),
                                  torch.full_like(self.slot_i_ne, -1))
            self._fused = ops.AdvGenSetup(base, self.nonego_ptr, slot_ne, self.crash_min_t, self.crash_min_infront,
                                          (w.get('adv_crash', 0.0), w.get('coll_veh_plan', 0.0), w.get('motion_prior_atk', 0.0),
                                           w.get('init_z_atk', 0.0)))
        return self._fused

    def _fusable(self, future_pred, tgt_traj, z, prior_out):
        w = self.loss_weights
        nmax = int(self.graph_sizes.max()) - 1 if self.B > 0 else 0
        NT = future_pred.size(1) - self.crash_min_t if future_pred.dim() == 3 else 0
        return (w.get('adv_crash', 0.0) > 0.0 and future_pred.dim() == 3 and tgt_traj.dim() == 3 and z.dim() == 2 and NT > 0 and
                nmax <= 64 and nmax * NT <= 1024 and not prior_out[0].requires_grad and not prior_out[1].requires_grad)

    def forward(self, future_pred, tgt_traj, z, prior_out, return_mins=False, attack_agt_idx=None, scene_alive=None):
        """One HIP call forward, one backward (strive_adv_gen_fwd/bwd) for the objective; the per-term entries of the
        reference's dict are evaluated by ``forward_terms`` only when somebody reads them (logging).
        ``scene_alive`` (B,) uint8 on the device or None (not in the reference): scenes with 0 have left the batch (closed loop:
        their planner rollout failed) -- every sum, count and mean is the one of the batch rebuilt without them, their agents get
        zero gradients, ``tgt_traj`` may be NaN for them."""
        if scene_alive is not None:
            scene_alive = scene_alive.to(torch.uint8).contiguous()
            tgt_traj = torch.where(scene_alive.to(torch.bool).view(-1, 1, 1), tgt_traj, torch.zeros_like(tgt_traj))
        if not self._fusable(future_pred, tgt_traj, z, prior_out):
            return self.forward_terms(future_pred, tgt_traj, z, prior_out, return_mins=return_mins, attack_agt_idx=attack_agt_idx,
                                      scene_alive=scene_alive)
        w = self.loss_weights
        loss, _, soft, _ = ops.adv_gen_loss(future_pred, tgt_traj, z, prior_out[0], prior_out[1], self._setup(),
                                            attack_agt_idx=attack_agt_idx, scene_alive=scene_alive)
        out = LossDict()
        terms = {}

        fp_s, tg_s, z_s = future_pred.detach(), tgt_traj.detach(), z.detach().clone()      # snapshot: see AvoidCollLoss.forward
        alive_s = None if scene_alive is None else scene_alive.clone()

        def term(key):
            def thunk():
                if not terms:
                    with torch.no_grad():
                        ft = self.forward_terms(fp_s, tg_s, z_s, prior_out, attack_agt_idx=attack_agt_idx, scene_alive=alive_s)
                        for k in list(ft.keys()):
                            terms[k] = ft[k]
                return terms[key]
            return thunk
        for key, present in (('init_loss', w.get('init_z', 0.0) > 0.0), ('motion_prior_loss', w.get('motion_prior', 0.0) > 0.0),
                             ('coll_veh_loss', w.get('coll_veh', 0.0) > 0.0), ('coll_veh_plan_loss', w.get('coll_veh_plan', 0.0) > 0.0),
                             ('coll_env_loss', w.get('coll_env', 0.0) > 0.0), ('adv_crash_loss', True)):
            if present:
                out.set_lazy(key, term(key))
        out['loss'] = loss
        if return_mins:
            NT = future_pred.size(1) - self.crash_min_t
            flat = soft.detach().cpu()
            cur_min_agt, cur_min_t = [], []
            for b in range(self.B):
                a0, a1 = int(self.nonego_ptr[b]), int(self.nonego_ptr[b + 1])
                k = int(torch.max(flat[a0:a1].reshape(-1), dim=0)[1])
                cur_min_agt.append(k // NT + 1)
                cur_min_t.append(k % NT + self.crash_min_t)
            out['min_agt'] = np.array(cur_min_agt, dtype=int)
            out['min_t'] = np.array(cur_min_t, dtype=int)
        return out

    def forward_terms(self, future_pred, tgt_traj, z, prior_out, return_mins=False, attack_agt_idx=None, scene_alive=None):
        """The same objective term by term with torch glue between the HIP kernels (reference :105-262).  With ``scene_alive``
        (see ``forward``) every term is restricted to the agents / pairs / scenes that are still in the batch."""
        w = self.loss_weights
        NA, B = future_pred.size(0), tgt_traj.size(0)
        dev = future_pred.device
        crash = soft = None
        cur_min_agt = cur_min_t = None
        sa = ne_alive = slot_alive = None
        if scene_alive is not None:
            sa = scene_alive.to(device=dev, dtype=torch.bool)
            ne_alive = sa[self.seg]                                               # (NA-B,)
            slot_alive = sa[self.slot_scene]
            tgt_traj = torch.where(sa.view(-1, 1, 1), tgt_traj, torch.zeros_like(tgt_traj))
        if w.get('adv_crash', 0.0) > 0.0:
            atk = future_pred.index_select(0, self.nonego_idx)[:, self.crash_min_t:, :]
            tgt = tgt_traj[:, self.crash_min_t:, :4]
            tgt_e = tgt[self.seg]                         # every attacker's own target row (index built once: no host sync)
            dist = torch.norm(atk[:, :, :2] - tgt_e[:, :, :2], dim=-1)
            din = dist
            inf = torch.full_like(din, float('inf'))
            if self.crash_min_infront is not None:
                behind = _behind(atk.detach(), tgt_e.detach(), self.crash_min_infront)
                always = (torch.sum(behind, dim=1, keepdim=True) == behind.size(1)).expand_as(behind)
                all_behind = torch.sum(always) == always.numel()
                always = always & ~all_behind
                din = torch.where(always, inf, din)
            if attack_agt_idx is not None:
                am = torch.zeros((NA,), dtype=torch.bool, device=dev)
                am[attack_agt_idx.to(dev)] = True
                am = am.index_select(0, self.nonego_idx).unsqueeze(1).expand_as(din)
                din = torch.where(~am, inf, din)
            if ne_alive is not None:
                din = torch.where(ne_alive.view(-1, 1), din, inf)                 # all-inf scene: soft-min weights 0 (:134-135)
            NT = future_pred.size(1) - self.crash_min_t
            soft = self._segment_softmin(din)
            weighted = soft * dist ** 2
            crash = torch.zeros((B,), device=dev).scatter_add(0, self.seg, weighted.sum(dim=1))
            if return_mins:
                flat = soft.detach().cpu()
                cur_min_agt, cur_min_t = [], []
                for b in range(B):
                    a0, a1 = int(self.nonego_ptr[b]), int(self.nonego_ptr[b + 1])
                    k = int(torch.max(flat[a0:a1].reshape(-1), dim=0)[1])
                    cur_min_agt.append(k // NT + 1)
                    cur_min_t.append(k % NT + self.crash_min_t)
        rew = 1.0 - torch.sum(soft.detach(), dim=1)          # (NA-B,) high for non-attackers

        prior_l = None
        if w.get('motion_prior', 0.0) > 0.0:
            prior_l = self.motion_prior_loss(z, prior_out) * (rew * w['motion_prior'] + (1.0 - rew) * w['motion_prior_atk'])

        fine = interp_traj(future_pred, scale_factor=3)
        veh_l = plan_l = None
        if ('coll_veh' in w or 'coll_veh_plan' in w) and (w['coll_veh'] > 0.0 or w['coll_veh_plan'] > 0.0):
            pen, cmask = self.veh_coll_loss.block_penalties(fine)
            if slot_alive is not None:
                cmask = cmask & slot_alive.view(1, -1)
            if w['coll_veh'] > 0.0:
                m_veh = cmask & (~self.slot_ego).view(1, -1)
                veh_l = (pen, m_veh)
            if w['coll_veh_plan'] > 0.0:
                # weight of a planner-involving pair = prior_reweight of its non-ego member (reference :192-204)
                one = torch.ones((1,), device=dev)
                rew1 = torch.cat([rew, one])
                wi = torch.where(self.slot_j_ego & ~self.slot_i_ego, rew1[self.slot_i_ne.clamp(min=0)], one)
                wj = torch.where(self.slot_i_ego & ~self.slot_j_ego, rew1[self.slot_j_ne.clamp(min=0)], one)
                pw = torch.where(self.slot_j_ego, wi, wj)
                m_plan = cmask & self.slot_ego.view(1, -1)
                plan_l = (pen * pw.view(1, -1), m_plan)

        env_l = None
        if w.get('coll_env', 0.0) > 0.0:
            env_l = self.env_coll_loss.valid_penalties(fine.index_select(0, self.nonego_idx))
            if ne_alive is not None:
                TO = fine.size(1)
                env_l = (env_l[0], env_l[1] & ne_alive.view(-1, 1).expand(-1, TO).reshape(-1))
        init_l = None
        if w.get('init_z', 0.0) > 0.0:
            coeff = rew * w['init_z'] + (1.0 - rew) * w['init_z_atk']
            if ne_alive is not None:
                coeff = coeff * ne_alive.to(coeff.dtype)
            init_l = torch.sum(torch.sum((self.init_z - z) ** 2, dim=1) * coeff)

        loss = 0.0
        out = LossDict()
        if init_l is not None:
            loss = loss + init_l.mean()
            out['init_loss'] = init_l
        if prior_l is not None:
            if ne_alive is None:
                loss = loss + prior_l.mean()
            else:
                pz = prior_l * ne_alive.to(prior_l.dtype).view(-1, *([1] * (prior_l.dim() - 1)))
                loss = loss + pz.sum() / torch.clamp(ne_alive.sum() * (prior_l.numel() // max(1, prior_l.size(0))), min=1).to(prior_l.dtype)
            out['motion_prior_loss'] = prior_l
        # the three collision terms are means over the entries currently in collision ([0.] if none): formed as masked
        # sums / counts on the device; the compacted lists themselves are produced only if somebody reads them
        if veh_l is not None:
            loss = loss + w['coll_veh'] * _masked_mean(*veh_l)
            out.set_lazy('coll_veh_loss', lambda: _compact_or_zero(*veh_l))
        if plan_l is not None:
            loss = loss + w['coll_veh_plan'] * _masked_mean(*plan_l)
            out.set_lazy('coll_veh_plan_loss', lambda: _compact_or_zero(*plan_l))
        if env_l is not None:
            loss = loss + w['coll_env'] * _masked_mean(*env_l)
            out.set_lazy('coll_env_loss', lambda: _compact_or_zero(*env_l))
        if crash is not None:
            if sa is None:
                loss = loss + w['adv_crash'] * crash.mean()
            else:
                loss = loss + w['adv_crash'] * (crash * sa.to(crash.dtype)).sum() / torch.clamp(sa.sum(), min=1).to(crash.dtype)
            out['adv_crash_loss'] = crash
        out['loss'] = loss
        if return_mins and cur_min_agt is not None:
            out['min_agt'] = np.array(cur_min_agt, dtype=int)
            out['min_t'] = np.array(cur_min_t, dtype=int)
        return out


# ------------------------------------------------------------------------------------------------
# collision metrics used as success tests by the optimisation loops (reference :515-623)
# ------------------------------------------------------------------------------------------------
VEH_COLL_THRESH = 0.02   # IoU must be over this to count as a collision for the metric (not the loss)


def check_single_veh_coll(traj_tgt, lw_tgt, traj_others, lw_others):
    """Does the target trajectory (T,4) collide with each of the other trajectories (N,T,4)?  UNNORMALISED inputs;
    NaN frames of the others are skipped.  Returns numpy ``veh_coll (N,) bool`` and ``coll_time (N,) int`` (first
    colliding step, T if none) like the reference (:517-565), computed in one HIP launch over all (agent, step) pairs."""
    import numpy as np
    N, FT, _ = traj_others.size()
    if N == 0:
        return np.zeros((0,), dtype=bool), np.zeros((0,), dtype=int)
    a = traj_tgt[:, :4].unsqueeze(0).expand(N, FT, 4).reshape(N * FT, 4)
    la = lw_tgt.view(1, 2).expand(N * FT, 2)
    b = traj_others[:, :, :4].reshape(N * FT, 4)
    lb = lw_others.view(N, 1, 2).expand(N, FT, 2).reshape(N * FT, 2)
    iou = ops.rect_iou(a, la, b, lb).view(N, FT)
    hit = iou > VEH_COLL_THRESH                                  # NaN compares False: those frames are skipped
    any_hit = hit.any(dim=1)
    first = torch.where(any_hit, torch.argmax(hit.to(torch.int32), dim=1), torch.full((N,), FT, device=hit.device, dtype=torch.long))
    return any_hit.cpu().numpy().astype(bool), first.cpu().numpy().astype(int)


def check_pairwise_veh_coll(traj, lw):
    """Collision bookkeeping over all pairs of the given trajectories (N,T,4), UNNORMALISED (reference :567-623): agent i
    is marked iff it overlaps some agent j > i at some step."""
    import numpy as np
    N, FT, _ = traj.size()
    ii, jj = torch.triu_indices(N, N, offset=1, device=traj.device)
    coll = torch.zeros((N,), dtype=torch.bool, device=traj.device)
    if ii.numel() > 0:
        Pn = ii.numel()
        a = traj[ii][:, :, :4].reshape(Pn * FT, 4)
        b = traj[jj][:, :, :4].reshape(Pn * FT, 4)
        la = lw[ii].view(Pn, 1, 2).expand(Pn, FT, 2).reshape(Pn * FT, 2)
        lb = lw[jj].view(Pn, 1, 2).expand(Pn, FT, 2).reshape(Pn * FT, 2)
        pair_hit = (ops.rect_iou(a, la, b, lb).view(Pn, FT) > VEH_COLL_THRESH).any(dim=1)
        coll.index_put_((ii[pair_hit],), torch.ones((int(pair_hit.sum()),), dtype=torch.bool, device=traj.device))
    did = coll.cpu().numpy().astype(bool)
    return {'num_coll_veh': float(np.sum(did)), 'num_traj_veh': float(N), 'did_collide': did}
