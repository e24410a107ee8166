"""Adversarial optimisation loop (reference src/utils/adv_gen_optim.py:19-211): open loop against the recorded ego
future (planner_name == 'ego') and closed loop against the rule-based planner (planner_name == 'hardcode',
strive_amd.planners.hardcode_goalcond_nusc: one C-ABI call per iteration on the futures as they sit on the device, where the
reference copies them to the host and walks its scenes in numpy)."""
import os

import torch
import torch.optim as optim

from .graphed import GraphedIteration, adam_kwargs, graph_mode


def _collate_index(scene_graph, device):
    """Row of cat([tgt_z, other_z]) that belongs to each agent of the batched graph (ego first in every scene); built once
    per graph and device (boolean-mask assignment synchronises with the host, the loops call this twice per closure)."""
    ptr = scene_graph.ptr
    cache = scene_graph.__dict__.get('_strive_collate')
    sig = (str(device), ptr.data_ptr(), ptr._version, tuple(ptr.shape))     # identity check: no device->host copy per call
    if cache is not None and cache[0] == sig:
        return cache[2], cache[3], cache[4]
    B = ptr.shape[0] - 1
    NA = int(ptr[-1])
    ego = ptr[:-1].to(device)
    is_ego = torch.zeros((NA,), dtype=torch.bool, device=device)
    is_ego[ego] = True
    src = torch.empty((NA,), dtype=torch.long, device=device)
    src[is_ego] = torch.arange(B, device=device)
    src[~is_ego] = torch.arange(NA - B, device=device) + B
    ego_idx = torch.nonzero(is_ego).flatten()
    other_idx = torch.nonzero(~is_ego).flatten()
    scene_graph.__dict__['_strive_collate'] = (sig, None, src, ego_idx, other_idx)
    return src, ego_idx, other_idx


def collate_tgt_other_z(scene_graph, tgt_z, other_z):
    """Interleave ego latents (B,[NS,]D) and the others' (NA-B,[NS,]D) into graph order (reference :19-36),
    as one index_select instead of a per-scene concatenation loop."""
    src, _, _ = _collate_index(scene_graph, other_z.device)
    return torch.cat([tgt_z, other_z], dim=0).index_select(0, src)


_rollout_streams = {}


def _shared_forward_on():
    return os.environ.get('STRIVE_SHARED_ROLLOUT', '1') != '0'        # (A/B switch)


def shared_forward_applies(model, z_a, kw_a, kw_b):
    """True when two_rollouts(..., same_values=True) will run ONE forward rollout for both results (ops._RolloutPairFn)."""
    fa, fb = kw_a.get('nfuture'), kw_b.get('nfuture')
    fa = model.FT if fa is None else fa
    fb = model.FT if fb is None else fb
    return bool(z_a.is_cuda and _shared_forward_on() and hasattr(model, 'decode_embedding_pair') and
                kw_a.get('ext_future') is kw_b.get('ext_future') and fa >= fb)


_planner_streams = {}


def two_rollouts(model, embed_info, scene_graph, map_idx, map_env, z_a, kw_a, z_b, kw_b, overlap=True, after_a=None, same_values=False):
    """The two rollouts of an adversarial / solution closure are independent until the losses.  On the MI355X they run on two
    HIP streams: the map CNN of one (bandwidth-bound, fills the chip) overlaps the GNN / GRU kernels of the other (latency
    chains on ~128 workgroups), forward and -- because autograd replays every node on the stream its forward ran on -- backward
    (measured: adversarial closure 20.4 -> 17.8 ms).  Same kernels, same inputs, same results; scratch buffers are per stream
    (ops._workspace).  Splitting ONE rollout into two scene halves the same way gains nothing (refine closure 13.51 vs 13.48 ms:
    the half-size CNN launches lose in tail effects what the hidden GNN time wins).
    ``after_a(out_a)`` (optional) is enqueued right behind rollout A on ITS stream -- the closed loop's planner rollout only
    needs A's futures, so its small kernels run under rollout B's map CNN; its result is returned as a third value."""
    from .. import ops
    dev = z_a.device
    if same_values and z_a.is_cuda and _shared_forward_on() and hasattr(model, 'decode_embedding_pair') and \
            kw_a.get('ext_future') is kw_b.get('ext_future'):
        # ``same_values``: z_a and z_b hold the same numbers (complementary detach of the same two leaves), so the two forward
        # rollouts are ONE computation: it is done once and each result keeps its own reverse sweep (ops._RolloutPairFn).  The
        # map CNN -- three quarters of a rollout, forward only -- runs once per iteration instead of twice: adversarial closure on
        # 512 agents 17.3 -> ~11.5 ms (profiles/r04_bench_line_adv.json)
        fa, fb = kw_a.get('nfuture'), kw_b.get('nfuture')
        fa = model.FT if fa is None else fa
        fb = model.FT if fb is None else fb
        if fa >= fb:
            out_a, out_b = model.decode_embedding_pair(z_a, z_b, embed_info, scene_graph, map_idx, map_env, ext_future=kw_a.get('ext_future'),
                                                       nfuture_a=fa, nfuture_b=fb)
            return (out_a, out_b) if after_a is None else (out_a, out_b, after_a(out_a))
    ns = lambda z: z.shape[1] if z.dim() == 3 else 1
    # weight / scene / map packs are built lazily inside the first rollout: that one runs on the caller's stream (the packs
    # then belong to it and are complete before any side stream is forked), see ops.decoder_packs_ready
    if not (z_a.is_cuda and overlap and ops.decoder_packs_ready(model, scene_graph, map_env, ns(z_a), dev)
            and ops.decoder_packs_ready(model, scene_graph, map_env, ns(z_b), dev)):
        out_a = model.decode_embedding(z_a, embed_info, scene_graph, map_idx, map_env, **kw_a)
        out_b = model.decode_embedding(z_b, embed_info, scene_graph, map_idx, map_env, **kw_b)
        return (out_a, out_b) if after_a is None else (out_a, out_b, after_a(out_a))
    cur = torch.cuda.current_stream(dev)
    streams = _rollout_streams.get(str(dev))
    if streams is None:
        # stream A outranks stream B: in the closed loop the planner waits for rollout A only and then runs under rollout B
        streams = (torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev))
        _rollout_streams[str(dev)] = streams
    outs, extra = [], None
    # the planner's small kernels run under the other rollout's CNN: conv2 stays on its non-persistent kernel for the calls
    # enqueued meanwhile (measured: 21.9 against 22.8 ms per closed-loop closure) -- a field of the descriptor those calls are
    # given (StriveCNN.conv2_plain), not a switch inside the library
    with ops.conv2_plain(after_a is not None):
        for i, (st, z, kw) in enumerate(zip(streams, (z_a, z_b), (kw_a, kw_b))):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(model.decode_embedding(z, embed_info, scene_graph, map_idx, map_env, **kw))
                if i == 0 and after_a is not None:
                    extra = after_a(outs[0])
    for st, o in zip(streams, outs):
        cur.wait_stream(st)
        for v in o.values():
            if torch.is_tensor(v):
                v.record_stream(cur)          # produced on a side stream, consumed (and later freed) on the caller's
    if after_a is None:
        return outs[0], outs[1]
    if torch.is_tensor(extra):
        extra.record_stream(cur)
    return outs[0], outs[1], extra


class AdvClosure(object):
    """State + one iteration of the adversarial optimisation (reference src/utils/adv_gen_optim.py:39-171):
    the two leaf latent groups, Adam over both, the two loss modules and ``step()`` = one closure (zero_grad, two
    rollouts with complementary detach, TgtMatchingLoss + AdvGenLoss, backward) followed by one ``Adam.step()``.
    ``run_adv_gen_optim`` and bench.py's ``--workload adv`` both drive this object."""

    def __init__(self, cur_z, lr, loss_weights, model, scene_graph, map_env, map_idx, embed_info, tgt_prior_distrib,
                 other_prior_distrib, feasibility_time, feasibility_infront_min, planner_fut=None, attack_agt_idx=None,
                 future_len=None, veh_coll_buffer=0.1, planner_name='ego', planner=None, on_planner_error='raise'):
        """``on_planner_error`` (closed loop only; not in the reference): what a planner rollout that fails in some scene -- an
        object pushed off its lane, a capacity limit: the cases in which the reference's numpy planner raises and its batch_size-1
        run loses that scene (adv_scenario_gen.py:540-543) -- costs a BATCH here.  'raise' (default, the reference's behaviour): the
        whole batch (StriveHipError from the planner's deferred check), with the planner overlapped under the adversarial half.
        'drop' (opt-in; bench.py and the full pipeline pass it): that scene alone; from the
        iteration of the failure on it is masked out of both losses on the device (the planner's ``alive`` flags go straight to the
        loss kernels, no host round trip), the other scenes continue exactly as in a batch rebuilt without it (the reference's
        remedy for scenes it gives up after the init stage, :323-356), and the caller is told (``failed_scenes``, a RuntimeWarning
        from run_adv_gen_optim; a batch that lost EVERY scene raises).  The planner object itself is not reconfigured: every rollout /
        check of this closure passes its own ``on_error``."""
        from ..losses.adv_gen_nusc import TgtMatchingLoss, AdvGenLoss
        dev = cur_z.device
        NA = cur_z.size(0)
        self.model, self.scene_graph, self.map_env, self.map_idx, self.embed_info = model, scene_graph, map_env, map_idx, embed_info
        self.ego_inds = scene_graph.ptr[:-1].to(dev)
        self.ego_mask = torch.zeros((NA,), dtype=torch.bool, device=dev)
        self.ego_mask[self.ego_inds] = True
        _, self.ego_idx, self.other_idx = _collate_index(scene_graph, dev)
        if attack_agt_idx is not None:
            attack_agt_idx = torch.as_tensor(attack_agt_idx).to(self.ego_inds) + self.ego_inds
        self.attack_agt_idx = attack_agt_idx
        self.future_len = model.FT if future_len is None else future_len
        self.tgt_z = cur_z[self.ego_mask].clone().detach()
        self.tgt_z.requires_grad = True
        self.other_z = cur_z[~self.ego_mask].clone().detach()
        self.other_z.requires_grad = True
        # the iteration can be replayed as a HIP graph (utils/graphed.py).  Closed loop: the planner's status flags accumulate in
        # one device tensor over the replays (its per-iteration look at them is skipped while capturing) and the loop checks them
        # after its last iteration, like the eager loop does
        self.graphed = graph_mode(NA, dev, rollouts=2, closed_loop=planner_name == 'hardcode')
        self.optim = optim.Adam([self.tgt_z, self.other_z], lr=lr, **adam_kwargs(self.graphed))
        self._iter = None
        self.unn = model.get_normalizer().unnormalize
        self.tgt_prior, self.other_prior = tgt_prior_distrib, other_prior_distrib
        self.tgt_loss = TgtMatchingLoss(loss_weights)
        self.adv_loss = AdvGenLoss(loss_weights, model.get_att_normalizer().unnormalize(scene_graph.lw),
                                   map_idx[scene_graph.batch], map_env, self.collated()[~self.ego_mask].clone().detach(),
                                   scene_graph.ptr, veh_coll_buffer=veh_coll_buffer, crash_loss_min_time=feasibility_time,
                                   crash_loss_min_infront=feasibility_infront_min)
        self.planner_name, self.planner = planner_name, planner
        if on_planner_error not in ('drop', 'raise'):
            raise ValueError("on_planner_error must be 'drop' or 'raise'")
        # failures are reported instead of raised ...
        self.report_failures = planner_name == 'hardcode' and on_planner_error == 'drop' and hasattr(planner, 'failed_scenes')
        # ... and a failed scene is masked out of the losses while the others go on -- in a batch that HAS others: the shipped
        # configs' one-scene batch keeps the overlapped iteration (its only scene failing leaves nothing to protect; the scene is
        # reported all the same and its results are meaningless, like the reference's raise for it)
        self.quarantine = self.report_failures and int(scene_graph.ptr.shape[0]) - 1 > 1
        # two-stream rollouts (see two_rollouts); set False to serialise.  A replayed HIP graph keeps ONE stream: with the fork /
        # join captured, hipGraphLaunch of the 16-agent adversarial iteration took 5.4 ms on the host and the iteration 8.6 ms
        # against 4.2 ms eager (profiles/r04_graph_ab.txt)
        self.overlap = not self.graphed
        if planner_name == 'ego':
            # open loop: the planner's trajectory is the ego's recorded future, injected into both rollouts
            self.planner_fut = scene_graph.future_gt[self.ego_mask][:, :, :4] if planner_fut is None else planner_fut
            assert self.planner_fut.size(1) == self.future_len
        elif planner_name == 'hardcode':
            # closed loop (reference :90-103): the rule-based planner reacts to the current rollout in every iteration;
            # nothing is injected into the decoder and the adversarial loss sees the model's own ego prediction
            import numpy as np
            if planner is None:
                raise ValueError("planner_name='hardcode' needs a planner object (planners.hardcode_goalcond_nusc.HardcodeNuscPlanner)")
            B = scene_graph.ptr.shape[0] - 1
            planner.reset(self.unn(scene_graph.past_gt[:, -1, :]), model.get_att_normalizer().unnormalize(scene_graph.lw),
                          scene_graph.batch, B, map_idx)
            # (with 'drop' the planner's own opportunistic look at the flags -- rollout -> check(wait=False) -- must not raise: plan()
            # passes on_error='report' with every rollout; the caller's planner keeps its own setting)
            self.agt_ptr = (scene_graph.ptr.cpu() - torch.arange(B + 1)).numpy()
            self.plan_t = np.linspace(model.dt, model.dt * self.future_len, self.future_len)
            self.planner_fut = None
            if hasattr(planner, 'prepare'):
                planner.prepare(self.plan_t, self.agt_ptr, self.plan_t)       # device tables built here, on the caller's stream
        else:
            raise NotImplementedError("planner_name must be 'ego' or 'hardcode'")

    def collated(self, detach_tgt=False, detach_other=False):
        t = self.tgt_z.clone().detach() if detach_tgt else self.tgt_z
        o = self.other_z.clone().detach() if detach_other else self.other_z
        return collate_tgt_other_z(self.scene_graph, t, o)

    def plan(self, future_pred):
        """The rule-based planner's reaction (B, FT, 4), NORMALISED, to the non-ego agents of ``future_pred`` (reference
        :133-139): the futures stay on the device, the plan comes back as a device tensor -- no host round trip."""
        agt = self.unn(future_pred.index_select(0, self.other_idx)).detach()
        kw = {'on_error': 'report'} if self.report_failures else {}
        fut = self.planner.rollout(agt, self.plan_t, self.agt_ptr, self.plan_t, control_all=False, **kw).to(self.scene_graph.future_gt)
        return self.model.get_normalizer().normalize(fut)

    def _two_rollouts(self, z_a, z_b, after_a=None):
        kw = dict(ext_future=self.planner_fut, nfuture=self.future_len)
        return two_rollouts(self.model, self.embed_info, self.scene_graph, self.map_idx, self.map_env, z_a, kw, z_b, kw,
                            overlap=self.overlap, after_a=after_a, same_values=True)

    def _step_closed_loop_overlapped(self, z_a, z_b, log):
        """Closed loop with the shared forward rollout: the planner (65 dependent float64 launches, one workgroup per scene) only
        needs rollout A's futures and only the matching loss needs its plan, so it runs on a side stream while the adversarial loss
        and ITS reverse sweep run on the caller's; the matching loss and the second sweep follow the join.  The two losses reach
        disjoint leaves (complementary detach), so two backward calls leave exactly the gradients of one call on their sum."""
        dev = z_a.device
        out_a, out_b = self._two_rollouts(z_a, z_b)
        cur = torch.cuda.current_stream(dev)
        side = _planner_streams.get(str(dev))
        if side is None:
            side = torch.cuda.Stream(dev)
            _planner_streams[str(dev)] = side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            planner_fut = self.plan(out_a['future_pred'])
        out_a['future_pred'].record_stream(side)
        adv_tgt = out_b['future_pred'].index_select(0, self.ego_idx)
        la = self.adv_loss(self.unn(out_b['future_pred']), self.unn(adv_tgt), self.other_z, self.other_prior,
                           attack_agt_idx=self.attack_agt_idx)
        la['loss'].backward(retain_graph=True)
        cur.wait_stream(side)
        planner_fut.record_stream(cur)
        lt = self.tgt_loss(self.unn(out_a['future_pred'].index_select(0, self.ego_idx)), self.unn(planner_fut), self.tgt_z,
                           self.tgt_prior)
        lt['loss'].backward()
        loss = lt['loss'].detach() + la['loss'].detach()
        if log is not None:
            loss_dict = {'tgt_match_' + k: v for k, v in lt.items()}
            loss_dict.update({'adv_' + k: v for k, v in la.items()})
            log(loss_dict, self.tgt_z, self.other_z)
        self.optim.step()
        return loss

    def step(self, log=None):
        """(reference src/utils/adv_gen_optim.py:107-171)"""
        m, g = self.model, self.scene_graph
        self.optim.zero_grad()
        z_a = self.collated(detach_other=True)      # ego latents get the matching loss only
        z_b = self.collated(detach_tgt=True)        # the others get the adversarial loss only
        kw = dict(ext_future=self.planner_fut, nfuture=self.future_len)
        if self.planner_name == 'hardcode' and self.overlap and os.environ.get('STRIVE_PLANNER_OVERLAP', '1') != '0' and \
                shared_forward_applies(m, z_a, kw, kw) and not self.quarantine:
            # (with quarantine BOTH losses wait for this iteration's planner rollout: which scenes are in the batch is its result)
            return self._step_closed_loop_overlapped(z_a, z_b, log)
        alive = None
        if self.planner_name == 'hardcode':
            # the planner reacts to rollout A only: it is enqueued behind A on A's stream and runs under rollout B
            out_a, out_b, planner_fut = self._two_rollouts(z_a, z_b, after_a=lambda o: self.plan(o['future_pred']))
            adv_tgt = out_b['future_pred'].index_select(0, self.ego_idx)      # the differentiable stand-in for the planner
            if self.quarantine:
                alive = self.planner.alive           # (B,) uint8 on the device, written by the rollout just enqueued
        else:
            out_a, out_b = self._two_rollouts(z_a, z_b)
            planner_fut = adv_tgt = self.planner_fut
        mask_kw = {} if alive is None else {'scene_alive': alive}
        lt = self.tgt_loss(self.unn(out_a['future_pred'].index_select(0, self.ego_idx)), self.unn(planner_fut), self.tgt_z,
                           self.tgt_prior, **mask_kw)
        la = self.adv_loss(self.unn(out_b['future_pred']), self.unn(adv_tgt), self.other_z, self.other_prior,
                           attack_agt_idx=self.attack_agt_idx, **mask_kw)
        loss = lt['loss'] + la['loss']
        loss.backward()
        if log is not None:
            # (reading the dict entries compacts the collision lists = host synchronisations: only when somebody looks)
            loss_dict = {'tgt_match_' + k: v for k, v in lt.items()}
            loss_dict.update({'adv_' + k: v for k, v in la.items()})
            log(loss_dict, self.tgt_z, self.other_z)
        self.optim.step()
        return loss


def run_adv_gen_optim(cur_z, lr, loss_weights, model, scene_graph, map_env, map_idx, num_iters, embed_info,
                      planner_name, tgt_prior_distrib, other_prior_distrib, feasibility_time, feasibility_infront_min,
                      planner=None, planner_viz_out=None, attack_agt_idx=None, future_len=None, veh_coll_buffer=0.1,
                      log=None, on_planner_error='raise'):
    """Same arguments and return value as the reference (src/utils/adv_gen_optim.py:39-211).  ``on_planner_error`` (closed loop,
    see AdvClosure; default 'raise' = the reference, whose planner raises at :133-139): with 'drop' a scene whose planner rollout fails
    leaves the losses from that iteration on instead of ending the whole batch; a RuntimeWarning names the scenes lost this way,
    a batch that lost all of its scenes raises, and they are listed in ``final_decoder_out['scenes_dropped']`` (with the limits they hit in
    ``final_decoder_out['planner_failures']``) -- their rows of the returned tensors are meaningless (the planner's trajectory
    is NaN) and the caller drops them like the reference drops scenes after its init stage (adv_scenario_gen.py:323-356)."""
    if planner_viz_out is not None:
        # (the reference renders the final closed-loop rollout frame by frame, :186-190: matplotlib + ffmpeg, outside the path)
        raise NotImplementedError('planner_viz_out: planner visualisation is not part of this package; render the returned '
                                  'final_result_traj instead')
    c = AdvClosure(cur_z, lr, loss_weights, model, scene_graph, map_env, map_idx, embed_info, tgt_prior_distrib,
                   other_prior_distrib, feasibility_time, feasibility_infront_min, attack_agt_idx=attack_agt_idx,
                   future_len=future_len, veh_coll_buffer=veh_coll_buffer, planner_name=planner_name, planner=planner,
                   on_planner_error=on_planner_error)
    if log is None and c.graphed:
        it = GraphedIteration(c.step, True)
        for _ in range(num_iters):
            it()
    else:
        for _ in range(num_iters):
            c.step(log=log)
    ego_inds, ego_mask, unn, adv_loss, future_len = c.ego_inds, c.ego_mask, c.unn, c.adv_loss, c.future_len

    cur_z = c.collated()
    with torch.no_grad():
        final_decoder_out = model.decode_embedding(cur_z, embed_info, scene_graph, map_idx, map_env, nfuture=future_len)
    final_result_traj = final_decoder_out['future_pred'].unsqueeze(1).clone().detach()
    if planner_name == 'ego':
        final_result_traj[ego_inds, torch.zeros_like(ego_inds)] = scene_graph.future_gt[ego_mask][:, :, :4]
    else:       # the planner's actual reaction to the final scenario (reference :184-192)
        final_result_traj[ego_inds, torch.zeros_like(ego_inds)] = c.plan(final_decoder_out['future_pred'])
        if hasattr(planner, 'check'):
            # deferred capacity / range status of every planner rollout of the loop: raises, or names the quarantined scenes
            failures = planner.check(on_error='report') if c.report_failures else planner.check()
            final_decoder_out['planner_failures'] = dict(failures or {})
            final_decoder_out['scenes_dropped'] = sorted((failures or {}).keys())
            if failures:
                B = int(scene_graph.ptr.shape[0]) - 1
                if len(failures) >= B:
                    from .._lib import StriveHipError
                    raise StriveHipError('run_adv_gen_optim: the planner rollout failed in every scene of the batch (%s): nothing to '
                                         'return' % '; '.join('scene %d: %s' % (b, v[0]) for b, v in sorted(failures.items())))
                import warnings
                warnings.warn('run_adv_gen_optim(on_planner_error=\'drop\'): planner rollout failed in scene(s) %s -- their rows of the '
                              'returned tensors are meaningless (final_decoder_out[\'scenes_dropped\'])' % sorted(failures.keys()),
                              RuntimeWarning)
    tgt_traj = final_result_traj[ego_inds, torch.zeros_like(ego_inds)]
    fin_kw = {'scene_alive': planner.alive} if c.report_failures else {}
    with torch.no_grad():
        fin = adv_loss(unn(final_decoder_out['future_pred']), unn(tgt_traj), cur_z[~ego_mask].clone().detach(),
                       other_prior_distrib, return_mins=True, **fin_kw)
    cur_min_agt = cur_min_t = None
    if 'min_agt' in fin:
        cur_min_agt = fin['min_agt'] + scene_graph.ptr[:-1].cpu().numpy()
    if 'min_t' in fin:
        cur_min_t = fin['min_t']
    return cur_z, final_result_traj, final_decoder_out, cur_min_agt, cur_min_t


def compute_adv_gen_success(final_result_traj, model, scene_graph, attack_agt):
    """Did the scenario make the attacker collide with the planner?  All inputs NORMALISED; ``final_result_traj``
    (NA,1,FT,4) with agent 0 = the planner's true reaction (reference src/utils/adv_gen_optim.py:214-235)."""
    from ..losses.adv_gen_nusc import check_single_veh_coll
    nrm, att = model.get_normalizer(), model.get_att_normalizer()
    planner_fut = nrm.unnormalize(final_result_traj[0, 0])
    other_fut = nrm.unnormalize(final_result_traj[1:, 0])
    planner_lw = att.unnormalize(scene_graph.lw[0])
    other_lw = att.unnormalize(scene_graph.lw[1:])
    coll_all, _ = check_single_veh_coll(planner_fut, planner_lw, other_fut, other_lw)
    return bool(coll_all[attack_agt - 1])
