"""ORACLE (test infrastructure only) -- optimisation-time and training-time losses.

CPU restatement in plain torch of the loss terms on STRIVE's latent-optimisation hot path.
It keeps the reference's batch-wide pair enumeration (``T*NA*NA`` pairs, cross-scene pairs masked
afterwards) because that is what the reference computes and what the CPU baseline must time; the
HIP product only evaluates the in-scene blocks.  Functions return the same tensors/dict keys
as the reference's modules.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .geometry import transform2frame
from .mapenv import coll_point


def log_normal(x, m, v):
    """Sum over the last dim of log N(x; m, v).  (reference src/losses/common.py:26-41)"""
    lp = -torch.log(torch.sqrt(v)) - math.log(math.sqrt(2 * math.pi)) - ((x - m) ** 2 / (2 * v))
    return torch.sum(lp, dim=-1)


def kl_normal(qm, qv, pm, pv):
    """KL(q||p) of diagonal normals, summed over the last dim.  (reference src/losses/common.py:8-24)"""
    e = 0.5 * (torch.log(pv) - torch.log(qv) + qv / pv + (qm - pm).pow(2) / pv - 1)
    return e.sum(-1)


def motion_prior_nll(z, prior_out):
    """(reference src/losses/adv_gen_nusc.py:343-364)"""
    mu, var = prior_out
    if z.dim() == 3:
        mu, var = mu.unsqueeze(1), var.unsqueeze(1)
    return -log_normal(z, mu, var)


def interp_traj(pred, scale_factor=3):
    """Linear x3 up-sampling in time of (x,y,hx,hy) with heading renormalised.
    (reference src/losses/adv_gen_nusc.py:625-644)"""
    multi = pred.dim() == 4
    if multi:
        NA, NS, T, _ = pred.shape
        pred = pred.reshape(NA * NS, T, 4)
    up = F.interpolate(pred.transpose(1, 2), scale_factor=scale_factor, mode='linear').transpose(1, 2)
    h = up[:, :, 2:4]
    up = torch.cat([up[:, :, :2], h / torch.norm(h, dim=-1, keepdim=True)], dim=-1)
    if multi:
        up = up.reshape(NA, NS, up.shape[1], 4)
    return up


def check_behind(attacker_fut, tgt_fut, ptr, thresh):
    """(NA-B,T) bool: attacker is behind the target (cosine of target heading vs target->attacker
    direction below ``thresh``).  (reference src/losses/adv_gen_nusc.py:646-673)"""
    sizes = (ptr[1:] - ptr[:-1]).tolist()
    tgt = torch.cat([tgt_fut[b:b + 1].expand(sizes[b] - 1, -1, -1) for b in range(len(sizes))], dim=0)
    d = attacker_fut[:, :, :2] - tgt[:, :, :2]
    d = d / torch.norm(d, dim=-1, keepdim=True)
    return torch.sum(d * tgt[:, :, 2:4], dim=-1) < thresh


class VehColl(object):
    """Circle-approximation vehicle-vehicle penalty.

    mode 'optim' restates VehCollLoss of reference src/losses/adv_gen_nusc.py:405-512 (compacted
    list of penalties of colliding valid pairs, ``[0.]`` sentinel, or raw ``(T,NA,NA)`` tensors);
    mode 'train' restates the variant in reference src/losses/traffic_model.py:166-238 (every valid
    pair, zeros included, plus the number of ordered in-scene pairs)."""

    def __init__(self, veh_att, ptr=None, buffer_dist=0.0, single_veh_idx=None, num_circ=5, mode='optim'):
        NA = veh_att.shape[0]
        dev = veh_att.device
        self.mode = mode
        self.num_circ = num_circ
        rad = veh_att[:, 1] / 2.0
        lo = -(veh_att[:, 0] / 2.0) + rad
        hi = (veh_att[:, 0] / 2.0) - rad
        cx = torch.stack([torch.linspace(lo[i].item(), hi[i].item(), num_circ) for i in range(NA)], dim=0).to(dev)
        self.centroids = torch.stack([cx, torch.zeros_like(cx), torch.ones_like(cx), torch.zeros_like(cx)], dim=2)
        self.penalty_dists = rad.view(NA, 1) + rad.view(1, NA) + buffer_dist
        if ptr is None:
            ptr = torch.tensor([0, NA], dtype=torch.long, device=dev)
        self.ptr = ptr
        mask = torch.zeros((NA, NA), dtype=torch.bool, device=dev)
        for b in range(1, len(ptr)):
            mask[ptr[b - 1]:ptr[b], ptr[b - 1]:ptr[b]] = True
        mask = mask & ~torch.eye(NA, dtype=torch.bool, device=dev)
        if single_veh_idx is not None:
            sel = torch.zeros((NA,), dtype=torch.bool, device=dev)
            sel[ptr[:-1] + single_veh_idx] = True
            mask = mask & (sel.view(1, NA) | sel.view(NA, 1))
        self.valid_mask = mask
        sizes = ptr[1:] - ptr[:-1]
        self.num_pairs = torch.sum(sizes * sizes - sizes)

    def min_dists(self, traj):
        NA, T, _ = traj.shape
        C = self.num_circ
        fr = traj[:, :, :4].reshape(NA * T, 4)
        cent = self.centroids.view(NA, 1, C, 4).expand(NA, T, C, 4).reshape(NA * T, C, 4)
        world = transform2frame(fr, cent, inverse=True).view(NA, T, C, 4)[:, :, :, :2].transpose(0, 1)
        c1 = world.view(T, NA, 1, C, 2).expand(T, NA, NA, C, 2).reshape(T * NA * NA, C, 2)
        c2 = world.view(T, 1, NA, C, 2).expand(T, NA, NA, C, 2).reshape(T * NA * NA, C, 2)
        d = torch.cdist(c1, c2).view(T * NA * NA, C * C)
        return torch.min(d, 1)[0].view(T, NA, NA)

    def __call__(self, traj, return_raw=False):
        NA, T, _ = traj.shape
        md = self.min_dists(traj)
        pd = self.penalty_dists.view(1, NA, NA)
        coll = (md <= pd) & self.valid_mask.view(1, NA, NA)
        if self.mode == 'train':
            pen = torch.where(coll, 1.0 - (md / pd), torch.zeros_like(pd))
            return pen[self.valid_mask.view(1, NA, NA).expand(T, NA, NA)], self.num_pairs
        if not return_raw and torch.sum(coll) == 0:
            return torch.Tensor([0.0]).to(traj.device)
        pen = 1.0 - (md / pd)
        if return_raw:
            return pen, coll
        return pen[coll]


class EnvColl(object):
    """Off-road penalty from the estimated collision point.
    mode 'optim': reference src/losses/adv_gen_nusc.py:366-403 (compacted valid penalties / [0.]);
    mode 'train': reference src/losses/traffic_model.py:240-295 (dense (NA,T), zeros elsewhere)."""

    def __init__(self, veh_att, mapixes, map_env, mode='optim'):
        self.map_env = map_env
        self.mapixes = mapixes
        self.veh_att = veh_att
        self.penalty_dists = torch.sqrt((veh_att[:, 0] ** 2 / 4.0) + (veh_att[:, 1] ** 2 / 4.0))
        self.mode = mode

    def __call__(self, traj):
        NA, T, _ = traj.shape
        flat = traj.reshape(NA * T, 4)
        att = self.veh_att.view(NA, 1, 2).expand(NA, T, 2).reshape(NA * T, 2)
        mix = self.mapixes.view(NA, 1).expand(NA, T).reshape(NA * T)
        pt = coll_point(self.map_env.nusc_raster[:, 0], self.map_env.nusc_dx, flat.detach(), att, mix)
        valid = ~torch.isnan(torch.sum(pt, dim=1))
        pdist = self.penalty_dists.view(NA, 1).expand(NA, T).reshape(NA * T)
        if self.mode == 'train':
            allp = torch.zeros((NA * T,), device=traj.device)
            if torch.sum(valid) == 0:
                return allp.view(NA, T)
            d = torch.norm(flat[:, :2][valid] - pt[valid], dim=1)
            allp[valid] = 1.0 - (d / pdist[valid])
            return allp.view(NA, T)
        if torch.sum(torch.isnan(pt)) == NA * T * 2:
            return torch.Tensor([0.0]).to(traj.device)
        d = torch.norm(flat[:, :2][valid] - pt[valid], dim=1)
        return 1.0 - (d / pdist[valid])


def tgt_matching_loss(weights, future_pred, tgt_traj, z, prior_out):
    """(reference src/losses/adv_gen_nusc.py:14-51) -- including its quirk that the prior term adds
    ``w_prior * tgt_loss.mean()`` to the objective and only *reports* the prior NLL (line 46)."""
    out = {}
    loss = 0.0
    tgt = None
    if weights['match_ext'] > 0.0:
        tgt = torch.sum((future_pred - tgt_traj) ** 2, dim=-1)
        loss = loss + weights['match_ext'] * tgt.mean()
        out['match_ext_loss'] = tgt
    if weights['motion_prior_ext'] > 0.0:
        out['motion_prior_ext_loss'] = motion_prior_nll(z, prior_out)
        loss = loss + weights['motion_prior_ext'] * tgt.mean()
    out['loss'] = loss
    return out


class AvoidColl(object):
    """(reference src/losses/adv_gen_nusc.py:264-341)"""

    def __init__(self, weights, veh_att, mapixes, map_env, init_z, veh_coll_buffer=0.0,
                 single_veh_idx=None, ptr=None):
        self.w = weights
        self.init_z = init_z
        self.single = single_veh_idx is not None
        self.veh = VehColl(veh_att, ptr=ptr, buffer_dist=veh_coll_buffer, single_veh_idx=single_veh_idx)
        if self.single:
            self.single_mask = torch.zeros((veh_att.shape[0],), dtype=torch.bool, device=veh_att.device)
            self.single_mask[ptr[:-1] + single_veh_idx] = True
            veh_att = veh_att[self.single_mask]
            mapixes = mapixes[self.single_mask]
        self.env = EnvColl(veh_att, mapixes, map_env)

    def __call__(self, future_pred, z, prior_out):
        out = {}
        loss = 0.0
        fine = interp_traj(future_pred, 3)
        if self.w['coll_veh'] > 0.0:
            v = self.veh(fine)
            loss = loss + self.w['coll_veh'] * v.mean()
            out['coll_veh_loss'] = v
        if self.w['coll_env'] > 0.0:
            e = self.env(fine if not self.single else fine[self.single_mask])
            loss = loss + self.w['coll_env'] * e.mean()
            out['coll_env_loss'] = e
        if self.w['motion_prior'] > 0.0:
            p = motion_prior_nll(z, prior_out)
            loss = loss + self.w['motion_prior'] * p.mean()
            out['motion_prior_loss'] = p
        if self.w['init_z'] > 0.0:
            i = torch.sum((self.init_z - z) ** 2, dim=1)
            loss = loss + self.w['init_z'] * i.mean()
            out['init_loss'] = i
        out['loss'] = loss
        return out


class AdvGen(object):
    """Adversarial objective: per-scene softmin crash term, attacker-aware prior / init weights,
    vehicle and environment collision regularisers.  (reference src/losses/adv_gen_nusc.py:53-262;
    the reference's progress prints at 145-148 are not reproduced)"""

    def __init__(self, weights, veh_att, mapixes, map_env, init_z, ptr, veh_coll_buffer=0.0,
                 crash_loss_min_time=0, crash_loss_min_infront=None):
        self.w = weights
        self.init_z = init_z
        self.ptr = ptr
        self.sizes = ptr[1:] - ptr[:-1]
        NA = veh_att.shape[0]
        self.ego_mask = torch.zeros((NA,), dtype=torch.bool)
        self.ego_mask[ptr[:-1]] = True
        self.nonego_ptr = ptr - torch.arange(len(ptr)).to(ptr)
        self.veh = VehColl(veh_att, ptr=ptr, buffer_dist=veh_coll_buffer)
        self.env = EnvColl(veh_att[~self.ego_mask], mapixes[~self.ego_mask], map_env)
        self.min_t = crash_loss_min_time
        self.min_infront = crash_loss_min_infront

    def __call__(self, future_pred, tgt_traj, z, prior_out, return_mins=False, attack_agt_idx=None):
        w = self.w
        NA = future_pred.shape[0]
        B = tgt_traj.shape[0]
        crash = soft = None
        min_agt = min_t = None
        if w.get('adv_crash', 0.0) > 0.0:
            atk = future_pred[~self.ego_mask][:, self.min_t:, :]
            tgt = tgt_traj[:, self.min_t:, :4]
            tgt_e = torch.cat([tgt[b:b + 1].expand(int(self.sizes[b]) - 1, -1, -1) for b in range(B)], dim=0)
            dist = torch.norm(atk[:, :, :2] - tgt_e[:, :, :2], dim=-1)
            din = dist
            if self.min_infront is not None:
                behind = check_behind(atk.detach(), tgt.detach(), self.ptr, self.min_infront)
                always = (torch.sum(behind, dim=1, keepdim=True) == behind.shape[1]).expand_as(behind)
                if torch.sum(always) == always.numel():
                    always = torch.zeros_like(always)
                din = torch.where(always, float('inf') * torch.ones_like(din), din)
            if attack_agt_idx is not None:
                am = torch.zeros(NA, dtype=torch.bool)
                am[attack_agt_idx] = True
                am = am[~self.ego_mask].unsqueeze(1).expand_as(din)
                din = torch.where(~am, float('inf') * torch.ones_like(din), din)
            NT = future_pred.shape[1] - self.min_t
            soft = []
            for b in range(B):
                sm = F.softmin(din[self.nonego_ptr[b]:self.nonego_ptr[b + 1]].reshape(-1), dim=0)
                if torch.isnan(sm[0]).item():
                    sm = torch.zeros_like(sm)
                soft.append(sm)
            min_agt = [(torch.max(s, dim=0)[1].item() // NT) + 1 for s in soft]
            min_t = [(torch.max(s, dim=0)[1].item() % NT) + self.min_t for s in soft]
            soft = torch.cat(soft, dim=0)
            weighted = soft * dist.reshape(-1) ** 2
            crash = torch.stack([torch.sum(weighted[(self.nonego_ptr[b] * NT):(self.nonego_ptr[b + 1] * NT)])
                                 for b in range(B)])
        rew = 1.0 - torch.sum(soft.detach().reshape(NA - B, -1), dim=1)

        prior_l = None
        if w.get('motion_prior', 0.0) > 0.0:
            coeff = rew * w['motion_prior'] + (1.0 - rew) * w['motion_prior_atk']
            prior_l = motion_prior_nll(z, prior_out) * coeff

        fine = interp_traj(future_pred, 3)
        veh_l = plan_l = None
        if ('coll_veh' in w or 'coll_veh_plan' in w) and (w['coll_veh'] > 0.0 or w['coll_veh_plan'] > 0.0):
            pens, cmask = self.veh(fine, return_raw=True)
            ego_rc = torch.zeros((NA, NA), dtype=torch.bool)
            ego_rc[self.ptr[:-1], :] = True
            ego_rc[:, self.ptr[:-1]] = True
            if w['coll_veh'] > 0.0:
                m = cmask & (~ego_rc).view(1, NA, NA)
                veh_l = torch.Tensor([0.0]) if torch.sum(m) == 0 else pens[m]
            if w['coll_veh_plan'] > 0.0:
                cw = torch.ones((NA,))
                cw[~self.ego_mask] = rew
                pm = torch.ones((NA, NA))
                for b in range(B):
                    pm[self.ptr[b], :] = cw
                    pm[:, self.ptr[b]] = cw
                m = cmask & ego_rc.view(1, NA, NA)
                plan_l = torch.Tensor([0.0]) if torch.sum(m) == 0 else (pens * pm.view(1, NA, NA))[m]

        env_l = None
        if w.get('coll_env', 0.0) > 0.0:
            env_l = self.env(fine[~self.ego_mask])
        init_l = None
        if w.get('init_z', 0.0) > 0.0:
            coeff = rew * w['init_z'] + (1.0 - rew) * w['init_z_atk']
            init_l = torch.sum(torch.sum((self.init_z - z) ** 2, dim=1) * coeff)

        out = {}
        loss = 0.0
        if init_l is not None:
            loss = loss + init_l.mean()
            out['init_loss'] = init_l
        if prior_l is not None:
            loss = loss + prior_l.mean()
            out['motion_prior_loss'] = prior_l
        if veh_l is not None:
            loss = loss + w['coll_veh'] * veh_l.mean()
            out['coll_veh_loss'] = veh_l
        if plan_l is not None:
            loss = loss + w['coll_veh_plan'] * plan_l.mean()
            out['coll_veh_plan_loss'] = plan_l
        if env_l is not None:
            loss = loss + w['coll_env'] * env_l.mean()
            out['coll_env_loss'] = env_l
        if crash is not None:
            loss = loss + w['adv_crash'] * crash.mean()
            out['adv_crash_loss'] = crash
        out['loss'] = loss
        if return_mins and min_agt is not None:
            out['min_agt'] = np.array(min_agt, dtype=int)
            out['min_t'] = np.array(min_t, dtype=int)
        return out


def traffic_model_loss(weights, g, pred, state_norm, att_norm, map_idx=None, map_env=None):
    """Training objective: reconstruction NLL on visible steps, KL(q||p), prior-sample vehicle and
    (ego-only) environment collision penalties.  (reference src/losses/traffic_model.py:34-118)"""
    vis = g.future_vis == 1.0
    gt = g.future_gt[vis]
    pf = pred['future_pred'][vis]
    recon = -log_normal(pf, gt[:, :4], torch.ones_like(pf))
    pm, pv = pred['prior_out']
    qm, qv = pred['posterior_out']
    kl = kl_normal(qm, qv, pm, pv)
    loss = weights['recon'] * recon.mean() + weights['kl'] * kl.mean()
    out = {'recon_loss': recon, 'kl_loss': kl}
    if weights['coll_veh_prior'] > 0.0 and 'future_samp' in pred:
        vl = VehColl(att_norm.unnormalize(g.lw), ptr=g.ptr, mode='train')
        pens, npairs = vl(state_norm.unnormalize(pred['future_samp']))
        cv = torch.sum(pens) / npairs
        loss = loss + weights['coll_veh_prior'] * cv
        out['coll_veh_prior'] = cv.view((1,))
    if weights['coll_env_prior'] > 0.0 and 'future_samp' in pred:
        ego = g.ptr[:-1]
        el = EnvColl(att_norm.unnormalize(g.lw[ego]), map_idx, map_env, mode='train')
        ce = el(state_norm.unnormalize(pred['future_samp'][ego]))
        loss = loss + weights['coll_env_prior'] * ce.mean()
        out['coll_env_prior'] = ce.view(-1)
    out['loss'] = loss.view((1,))
    return out


# ------------------------------------------------------------------------------------------------
# Success / feasibility tests around the optimisation loops
# ------------------------------------------------------------------------------------------------
VEH_COLL_THRESH = 0.02     # reference src/losses/adv_gen_nusc.py:515
ENV_COLL_THRESH = 0.05     # reference src/losses/traffic_model.py:17


def check_single_veh_coll(traj_tgt, lw_tgt, traj_others, lw_others):
    """(veh_coll (N,) bool, coll_time (N,) int): first step at which the target box overlaps each other agent's box with
    IoU > 0.02, FT if never; NaN frames of the others are skipped (reference src/losses/adv_gen_nusc.py:517-565)."""
    import numpy as np
    from .geometry import rect_iou
    tt, lt = traj_tgt.cpu().numpy(), lw_tgt.cpu().numpy()
    to, lo = traj_others.cpu().numpy(), lw_others.cpu().numpy()
    N, FT, _ = to.shape
    coll = np.zeros((N,), dtype=bool)
    when = np.ones((N,), dtype=int) * FT
    for j in range(N):
        for t in range(FT):
            if np.isnan(to[j, t]).any():
                continue
            if rect_iou(tt[t], lt, to[j, t], lo[j]) > VEH_COLL_THRESH:
                coll[j] = True
                when[j] = t
                break
    return coll, when


def check_pairwise_veh_coll(traj, lw):
    """Agent i counts as collided iff it overlaps (IoU > 0.02) some agent j > i at some step
    (reference src/losses/adv_gen_nusc.py:567-623)."""
    import numpy as np
    from .geometry import rect_iou
    tr, l = traj.cpu().numpy(), lw.cpu().numpy()
    N, FT, _ = tr.shape
    coll = np.zeros((N,), dtype=bool)
    for i in range(N):
        for j in range(i + 1, N):
            if coll[i]:
                break
            for t in range(FT):
                if rect_iou(tr[i, t], l[i], tr[j, t], l[j]) > VEH_COLL_THRESH:
                    coll[i] = True
                    break
    return {'num_coll_veh': float(coll.sum()), 'num_traj_veh': float(N), 'did_collide': coll}


def determine_feasibility(samples, normalizer, thresh, time=0, vel=0.0, infront_min=None, check_sep=True, raster=None,
                          dx=None, map_idx=None):
    """Restatement of determine_feasibility_nusc (reference src/utils/scenario_gen.py:30-107) for ONE scene:
    samples (NA,NS,FT,4) normalised, agent 0 = ego.  Returns (feasible (NA-1,), step (NA-1,), dist (NA-1,))."""
    from . import mapenv
    if samples.size(0) == 1:
        return None, None, None
    s = normalizer.unnormalize(samples)
    ego, oth = s[0:1], s[1:]
    NA, NS, FT, _ = oth.shape
    d = torch.norm(ego[..., :2] - oth[..., :2], dim=-1)[:, :, time:]
    if infront_min is not None:
        e2a = oth[:, :, time:, :2] - ego[:, :, time:, :2]
        e2a = e2a / torch.norm(e2a, dim=-1, keepdim=True)
        cs = torch.sum(e2a * ego[:, :, time:, 2:4], dim=-1)
        d = torch.where(cs >= infront_min, d, torch.full_like(d, float('inf')))
    per_t, which = torch.min(d, dim=1)                         # over samples -> (NA, FT')
    dist, step = torch.min(per_t, dim=1)
    step = step + time
    feasible = (d < thresh).sum(dim=[1, 2]) > 0
    if check_sep:
        ar = torch.arange(NA)
        samp = which[ar, step - time]
        a_xy = oth[ar, samp, step][:, :2]
        e_xy = ego.expand(NA, NS, FT, 4)[ar, samp, step][:, :2]
        cut = mapenv.line_hits_layer(raster[:, 0], dx, a_xy, e_xy, map_idx.expand(NA))
        feasible = feasible & ~cut
    v = torch.norm(oth[:, :, 1:, :2] - oth[:, :, :-1, :2], dim=-1)
    feasible = feasible & (v.amax(dim=(1, 2)) > vel)
    return feasible, step, dist


def compute_disp_err(future_gt, ptr, future_pred, normalizer):
    """Ego-only sample displacement errors (reference src/losses/traffic_model.py:297-364): future_gt (NA,FT,6) and
    future_pred (NA,NS,FT',4) normalised; per scene min-over-samples ADE / FDE of position (m) and heading (deg), and the
    average pairwise distance between samples."""
    FT = min(future_pred.size(2), future_gt.size(1))
    ego = ptr[:-1]
    gt = normalizer.unnormalize(future_gt[:, :FT])[ego]
    pr = normalizer.unnormalize(future_pred[:, :, :FT])[ego]
    B, NS = pr.shape[0], pr.shape[1]
    out = {k: torch.zeros((B,)) for k in ('pos_minADE', 'pos_minFDE', 'ang_minADE', 'ang_minFDE', 'APD')}
    for b in range(B):
        ade, fde, aade, afde = [], [], [], []
        for s in range(NS):
            d = torch.norm(gt[b, :, :2] - pr[b, s, :, :2], dim=-1)
            gh = gt[b, :, 2:4] / torch.norm(gt[b, :, 2:4], dim=-1, keepdim=True)
            ph = pr[b, s, :, 2:4] / torch.norm(pr[b, s, :, 2:4], dim=-1, keepdim=True)
            a = torch.rad2deg(torch.acos(torch.sum(gh * ph, dim=-1).clamp(-1, 1)))
            ade.append(d.mean()); fde.append(d[-1]); aade.append(a.mean()); afde.append(a[-1])
        out['pos_minADE'][b], out['pos_minFDE'][b] = torch.stack(ade).min(), torch.stack(fde).min()
        out['ang_minADE'][b], out['ang_minFDE'][b] = torch.stack(aade).min(), torch.stack(afde).min()
        tot = 0.0
        for s in range(NS):
            for q in range(NS):
                tot = tot + torch.norm(pr[b, s, :, :2] - pr[b, q, :, :2], dim=-1).sum()
        out['APD'][b] = tot / (NS * (NS - 1) * FT)
    return out


def compute_coll_rate_veh(edge_index, lw, future_pred, state_normalizer, att_normalizer):
    """(did_collide (NA,NS) bool, count): per sample, every connected pair (j, i) with j > i is walked in edge order; agent i
    is marked at the first step where the two boxes overlap with IoU > 0.02; pairs of an already marked agent and NaN frames
    are skipped (reference src/losses/traffic_model.py:465-545)."""
    import numpy as np
    from .geometry import rect_iou
    tr = state_normalizer.unnormalize(future_pred).cpu().numpy()
    att = att_normalizer.unnormalize(lw).cpu().numpy()
    NA, NS, FT, _ = tr.shape
    did = np.zeros((NA, NS), dtype=bool)
    count = 0
    for s in range(NS):
        for aj, ai in edge_index.cpu().numpy().T:
            if aj <= ai or did[ai, s]:
                continue
            for t in range(FT):
                if np.isnan(tr[ai, s, t]).any() or np.isnan(tr[aj, s, t]).any():
                    continue
                if rect_iou(tr[ai, s, t], att[ai], tr[aj, s, t], att[aj]) > VEH_COLL_THRESH:
                    did[ai, s] = True
                    count += 1
                    break
    return did, count
