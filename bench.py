#!/usr/bin/env python3
"""bench.py -- adv-optim agent*timesteps/s of STRIVE's latent-optimisation closure on MI355X.

A "step" is one optimisation closure of one of the reference's loops over one scene batch, followed by its Adam
step.  Workloads (``--workload``):

  refine       (default; BASELINE.json configs[1], the headline) the refine loop's closure (reference
               src/refine_traffic_optim.py:186-218): zero_grad -> TrafficModel.decode_embedding(z, nfuture=FT) ->
               AvoidCollLoss -> backward -> Adam, on 32 synthetic scenes x 16 agents per GPU (NA = 512), fp32.  With N
               GPUs every rank owns its own 32 scenes (scenes are independent: weak scaling, no data-path collective).
  adv          (configs[2]) the adversarial loop's closure (reference src/utils/adv_gen_optim.py:107-171, planner 'ego'):
               two rollouts with complementary detach + TgtMatchingLoss + AdvGenLoss + backward + Adam on ~512 agents in
               scenes of 2..30 agents; advances 2*NA*FT agent*timesteps.  Weak scaling like `refine`.
               ``--planner hardcode``: the same closure in CLOSED LOOP against the rule-based planner (adv_gen_rule_based.cfg:
               planner 'hardcode', planner_cfg 'default'; reference :133-139): agents on a synthetic lane graph, one
               HardcodeNuscPlanner.rollout (31 planner steps per scene, on the device) per closure, nothing injected into
               the decoder.
  sharded4096  (configs[4]) ONE batch of ~4096 agents with NC = 5 classes (reduce_cats), split over the ranks by
               strive_amd.distributed.shard_scenes; every rank runs the adversarial closure on its scenes.  The job is
               fixed, so this is strong scaling.
  full         (configs[2] end to end) the whole per-batch sequence of adv_scenario_gen.py:262-462 with the iteration counts of
               adv_gen_rule_based.cfg on ~512 agents on a synthetic lane graph: embed -> init optimisation (75 iterations, lr 0.1)
               -> planner rollout + init optimisation towards it (100) -> scenes whose planner already collides are dropped ->
               closed-loop adversarial optimisation (200, planner 'hardcode') -> success tests (rotated-box IoU kernel) ->
               solution optimisation of the succeeded scenes (200) -> success tests.  One "step" = one batch through all of it;
               agent*timesteps = rollouts x agents x steps actually decoded.  --iters scales the four iteration counts.
  sample       the feasibility pre-pass of adv_scenario_gen.py:160-174: TrafficModel.sample_batched(NS = 20, include_mean = True)
               under no_grad on 32 scenes x 16 agents: embed (map crop + CNN, past encoder, prior network) + ONE joint rollout
               of NA x 20 sample rows (forward only); advances NA*20*FT agent*timesteps.  Weak scaling.
  train        (configs[3]) one training step of train_traffic.cfg (reference src/train_traffic.py:103-131): batch of 4
               scenes x 16 agents per GPU, TrafficModel.forward(future_sample=True) = posterior-sample + prior-sample rollout,
               TrafficModelLoss, backward to all 174 parameter tensors, flat-bucket gradient all-reduce (RCCL) with the skip
               vote, Adam(lr 1e-5); advances 2*NA*FT agent*timesteps.  Weak scaling (data parallel).

``python bench.py --gpus N`` launches the N ranks itself (re-executing under torch.distributed.run on 127.0.0.1) when it
was not started by a launcher; under a launcher (WORLD_SIZE set) it is one rank of the job.

Output: ONE JSON line on rank 0 with two extra objects:
  roofline     -- the dominant kernel (a map-CNN convolution on the fp16 matrix cores), timed live with events on the
                  launching stream, against BOTH roofs (matrix: algorithmic FLOPs x issued products / duration; HBM:
                  algorithmic bytes / duration) -- `bound` names the one with the larger fraction;
                  `bandwidth_kernels` = achieved GB/s of the byte-bound kernels (algorithmic bytes / live duration).
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm, oracle/) timed on this host: 1 warm-up + 3 timed
                  closures on a bounded sample of the same workload, autograd incl. weight gradients like the reference.
Any failure while collecting either makes the run exit non-zero (after printing the line with the error recorded).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 matrix peak
PEAK_HBM_GBS = 8000.0              # same guide: HBM3E ~8 TB/s
# algorithmic FLOPs per agent of each map-CNN kernel (2 * Cout * OH * OW * Cin * k * k), SURVEY.md §8(a) a8
# (the fifth kernel is the fused tail: conv5 + conv6 + Linear(512, 64) in one launch)
CONV_FLOPS = [2 * 16 * 125 * 125 * 4 * 49, 2 * 32 * 61 * 61 * 16 * 25, 2 * 64 * 29 * 29 * 32 * 25,
              2 * 64 * 14 * 14 * 64 * 9, 2 * 128 * 6 * 6 * 64 * 9 + 2 * 128 * 2 * 2 * 128 * 9 + 2 * 512 * 64]
# conv2 runs on specialised producer / consumer waves (conv_ws_kernel), conv3 / conv4 on specialised waves with streamed weights
# (conv_wsx_kernel) unless the options conv_ws / conv_wsx (STRIVE_CONV_WS=0, STRIVE_CONV_WSX=0) keep conv_bf6_kernel: time the ones the
# closure launches
CONV_WS = os.environ.get('STRIVE_CONV_WS', '1') != '0'
CONV_WSX = os.environ.get('STRIVE_CONV_WSX', '1') != '0'
CONV_NAMES = ['conv1b_kernel<true> (fused crop -> conv1)', 'conv_ws_kernel<conv2>' if CONV_WS else 'conv_bf6_kernel<conv2>',
              'conv_wsx_kernel<conv3>' if CONV_WSX else 'conv_bf6_kernel<conv3>',
              'conv_wsx_kernel<conv4>' if CONV_WSX else 'conv_bf6_kernel<conv4>', 'cnn_tail_kernel (conv5 + conv6 + Linear)']
# strive_map_cnn_bench_layer ids of the kernels strive_map_cnn_fwd launches
CONV_LAYER_IDS = [0, 51 if CONV_WS else 1, 52 if CONV_WSX else 2, 53 if CONV_WSX else 3, 7]
# algorithmic HBM bytes per agent of the CNN kernels: input read once + output written once (fp32 activations, uint8 raster)
CONV_BYTES = [4 * 256 * 256 + 16 * 125 * 125 * 4, (16 * 125 * 125 + 32 * 61 * 61) * 4, (32 * 61 * 61 + 64 * 29 * 29) * 4,
              (64 * 29 * 29 + 64 * 14 * 14) * 4, (64 * 14 * 14 + 64) * 4]
# matrix-core work actually issued per algorithmic FLOP and the dense peak it runs against
# (/opt/skills/guides/MI355X_MICROARCH.md: bf16 / fp16 dense 2516 TFLOP/s, f32 157.3): conv1 = 2 fp16 weight pieces x the
# uint8 crop, conv2-conv6 = 3 fp16 products per fp32 product (two-piece round-to-nearest split of both operands)
PEAK_F16_MFMA_TFLOPS = 2516.0
CONV_ISSUE = [(2, PEAK_F16_MFMA_TFLOPS, 'f16'), (3, PEAK_F16_MFMA_TFLOPS, 'f16'), (3, PEAK_F16_MFMA_TFLOPS, 'f16'),
              (3, PEAK_F16_MFMA_TFLOPS, 'f16'), (3, PEAK_F16_MFMA_TFLOPS, 'f16')]
NK = len(CONV_NAMES)

REFINE_WEIGHTS = {'coll_veh': 100.0, 'coll_env': 100.0, 'init_z': 0.01, 'motion_prior': 1.0}   # refine_traffic_optim.cfg:26-29
ADV_WEIGHTS = {'coll_veh': 20.0, 'coll_veh_plan': 20.0, 'coll_env': 20.0, 'init_z': 0.5, 'init_z_atk': 0.05,
               'motion_prior': 1.0, 'motion_prior_atk': 0.005, 'motion_prior_ext': 0.0001, 'match_ext': 10.0,
               'adv_crash': 2.0}                                                               # adv_gen_rule_based.cfg:33-42


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------

def variable_scene_sizes(total, key, lo=2, hi=30):
    """Scene sizes n_b in [lo, hi] summing to `total` (the last scene takes the remainder), deterministic."""
    from strive_amd import synth
    draws = synth.counter_uniform((4 * total // (lo + hi) + 64,), key + '/sizes', lo, hi + 1).astype(int).tolist()
    sizes, acc = [], 0
    for n in draws:
        n = min(max(int(n), lo), hi)
        if acc + n > total - lo:
            break
        sizes.append(n)
        acc += n
    rest = total - acc
    while rest > hi:
        sizes.append(hi)
        rest -= hi
    if rest >= 1:
        sizes.append(rest)
    assert sum(sizes) == total
    return sizes


def workload_scenes(args, rank, world):
    """-> (list of (scene size, scene key) owned by this rank, description, scaling)."""
    if args.workload == 'refine':
        own = [(args.agents, 'bench/r%d/%d' % (rank, b)) for b in range(args.scenes)]
        return own, '%d scenes x %d agents per GPU' % (args.scenes, args.agents), 'weak'
    if args.workload == 'adv':
        sizes = variable_scene_sizes(args.total_agents or 512, 'bench/adv/r%d' % rank)
        own = [(n, 'bench/adv/r%d/%d' % (rank, b)) for b, n in enumerate(sizes)]
        return own, '%d agents per GPU in %d scenes of 2..30' % (sum(sizes), len(sizes)), 'weak'
    if args.workload == 'full':
        sizes = variable_scene_sizes(args.total_agents or 512, 'bench/adv/r%d' % rank)
        own = [(n, 'bench/adv/r%d/%d' % (rank, b)) for b, n in enumerate(sizes)]
        return own, '%d agents per GPU in %d scenes of 2..30' % (sum(sizes), len(sizes)), 'weak'
    if args.workload == 'sample':
        own = [(args.agents, 'bench/sample/r%d/%d' % (rank, b)) for b in range(args.scenes)]
        return own, '%d scenes x %d agents per GPU x 20 samples' % (args.scenes, args.agents), 'weak'
    if args.workload == 'train':
        nsc = 4 if args.scenes == 32 else args.scenes          # train_traffic.cfg: batch_size 4 (scenes per GPU)
        own = [(args.agents, 'bench/train/r%d/%d' % (rank, b)) for b in range(nsc)]
        return own, '%d scenes x %d agents per GPU' % (nsc, args.agents), 'weak'
    from strive_amd.distributed import shard_scenes
    sizes = variable_scene_sizes(args.total_agents or 4096, 'bench/sharded')
    part = shard_scenes(sizes, world)
    own = [(sizes[i], 'bench/sharded/%d' % i) for i in part[rank]]
    return own, 'one batch of %d agents in %d scenes of 2..30, scene-sharded over %d ranks' % (sum(sizes), len(sizes), world), 'strong'


def build_model(device, NC, lane_keeping=False):
    """Random-init weights of the reference architecture (synth.fill_state_dict); the closed-loop workloads (rule-based planner) take the
    lane-keeping variant (synth.lane_keeping_weights: the decoder's output layer scaled down so that agents follow their lanes)."""
    from strive_amd import synth
    from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors
    from strive_amd.models.traffic_model import TrafficModel
    from strive_amd.datasets.utils import MeanStdNormalizer
    m = TrafficModel(4, 12, 256, NC)
    sd = synth.fill_state_dict(m.state_dict())
    m.load_state_dict(synth.lane_keeping_weights(sd) if lane_keeping else sd)
    m.set_normalizer(MeanStdNormalizer(*state_norm_tensors()))
    m.set_att_normalizer(MeanStdNormalizer(*att_norm_tensors()))
    m.set_bicycle_params(NUSC_BIKE_PARAMS)
    return m.eval().to(device)


def build_batch(own, NC, raster_px, FT_data=12, lane_graph=None):
    """Synthetic scenes; with a lane graph the agents sit on its nodes (the rule-based planner needs lanes to follow), the
    scenes spread over the crossings of the map."""
    import numpy as np
    from strive_amd import synth
    from strive_amd.graph import Batch
    extent = raster_px * 0.25
    if lane_graph is None:
        scenes = [synth.make_scene(n, key, FT=FT_data, NC=NC, map_extent=(extent, extent)) for n, key in own]
    else:
        scenes = []
        per_row = max(1, int((extent - 240.0) // 120.0))
        for b, (n, key) in enumerate(own):
            centre = (129.0 + 120.0 * (b % per_row), 129.0 + 120.0 * ((b // per_row) % per_row))
            # (10 m between agents and speeds of 4..6 m/s: with 6 m and 2..8 m/s a quarter of the scenes had the rule-based planner
            # run into a slower agent on its own lane before the adversary did anything -- scenes the reference drops after its
            # init stage, src/adv_scenario_gen.py:323-356)
            poses = synth.lane_scene_poses(lane_graph, n, key + '/lane', radius=34.0 + 2.2 * n, centre=centre, min_gap=10.0, speed=(4.0, 6.0))
            scenes.append(synth.make_scene(n, key, FT=FT_data, NC=NC, poses=poses))
    batch = Batch.from_data_list(scenes)
    return batch, torch.zeros((len(own),), dtype=torch.long)


def build_env(raster_px, device, lane_graph=None):
    from strive_amd import synth
    raster, dx = synth.make_raster(raster_px, raster_px)
    return synth.SyntheticMapEnv(raster, dx, lane_graph=lane_graph).to(device)


def refine_closure_factory(m, env, batch, map_idx, FT, device):
    from strive_amd import synth
    from strive_amd.losses.adv_gen_nusc import AvoidCollLoss
    from strive_amd.utils.scenario_gen import detach_embed_info
    g = batch.to(device)
    mi = map_idx.to(device)
    with torch.no_grad():
        emb = detach_embed_info(m.embed(g, mi, env))
    z0 = synth.make_latents(emb['prior_out'][0].cpu(), emb['prior_out'][1].cpu(), key='bench/z').to(device)
    z = z0.clone().requires_grad_(True)
    # like strive_amd.refine_traffic_optim: small batches replay the iteration as a HIP graph (strive_amd/utils/graphed.py)
    from strive_amd.utils.graphed import GraphedIteration, adam_kwargs, graph_mode
    graphed = graph_mode(z.shape[0], device)
    opt = torch.optim.Adam([z], lr=0.05, **adam_kwargs(graphed))
    loss_fn = AvoidCollLoss(REFINE_WEIGHTS, m.get_att_normalizer().unnormalize(g.lw), mi[g.batch], env, z0.clone(),
                            veh_coll_buffer=0.2)

    def iteration():
        opt.zero_grad()
        pred = m.decode_embedding(z, emb, g, mi, env, nfuture=FT)['future_pred']
        ld = loss_fn(m.get_normalizer().unnormalize(pred), z, emb['prior_out'])
        ld['loss'].backward()
        opt.step()
        return ld['loss']
    step = GraphedIteration(iteration, graphed)

    def graph_vs_eager():
        """After the timed region: from the SAME latents and Adam state, one iteration replayed from the graph and one run eagerly;
        the largest difference of the latents they leave (the driver-timed path against the one the eager tests cover)."""
        if step.graph is None:
            return None
        st = opt.state[z]
        saved = (z.detach().clone(), {k: v.clone() for k, v in st.items() if torch.is_tensor(v)})

        def restore():
            with torch.no_grad():
                z.copy_(saved[0])
                for k, v in saved[1].items():
                    st[k].copy_(v)
        step()
        zg = z.detach().clone()
        restore()
        iteration()
        ze = z.detach().clone()
        restore()
        torch.cuda.synchronize()
        return float((zg - ze).abs().max()), float((ze - saved[0]).abs().max())
    step.graph_vs_eager = graph_vs_eager
    return step, emb, g, mi, 1


def adv_closure_factory(m, env, batch, map_idx, FT, device, on_planner_error='drop'):
    """reference src/utils/adv_gen_optim.py:39-171 in 'ego' planner mode, latents initialised at the posterior mean like
    adv_scenario_gen.py:297-312 does after the init optimisation."""
    from strive_amd.utils.adv_gen_optim import AdvClosure
    from strive_amd.utils.scenario_gen import detach_embed_info
    g = batch.to(device)
    mi = map_idx.to(device)
    with torch.no_grad():
        emb = detach_embed_info(m.embed(g, mi, env))
    NA = g.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool, device=device)
    ego[g.ptr[:-1].to(device)] = True
    pm, pv = emb['prior_out']
    if getattr(env, 'lane_graphs', None) is not None:
        # closed loop (adv_gen_rule_based.cfg: planner 'hardcode', planner_cfg 'default'; adv_scenario_gen.py:277-281)
        from strive_amd.planners.planner import PlannerConfig
        from strive_amd.planners.hardcode_goalcond_nusc import HardcodeNuscPlanner, CONFIG_DICT
        planner = HardcodeNuscPlanner(env, PlannerConfig(**CONFIG_DICT['default']))
        c = AdvClosure(emb['posterior_out'][0].clone(), 0.05, ADV_WEIGHTS, m, g, env, mi, emb, (pm[ego], pv[ego]),
                       (pm[~ego], pv[~ego]), 2, 0.0, future_len=FT, veh_coll_buffer=0.1, planner_name='hardcode', planner=planner,
                       on_planner_error=on_planner_error)

        from strive_amd.utils.graphed import GraphedIteration
        step = GraphedIteration(c.step, c.graphed)
        step.planner = planner
        step.closure = c
        return step, emb, g, mi, 2
    c = AdvClosure(emb['posterior_out'][0].clone(), 0.05, ADV_WEIGHTS, m, g, env, mi, emb, (pm[ego], pv[ego]),
                   (pm[~ego], pv[~ego]), 2, 0.0, planner_fut=g.future_gt[ego][:, :FT, :4].contiguous(), future_len=FT,
                   veh_coll_buffer=0.1)
    from strive_amd.utils.graphed import GraphedIteration
    return GraphedIteration(c.step, c.graphed), emb, g, mi, 2


def full_pipeline_factory(m, env, batch, map_idx, FT, device, scale=1.0):
    """One scene batch through the reference's whole optimisation sequence (src/adv_scenario_gen.py:262-462, planner 'hardcode',
    configs/adv_gen_rule_based.cfg: num_iters 200, lr 0.05, sol_future_len 16, feasibility_time 2 (default), infront_min 0)."""
    import numpy as np
    from strive_amd.graph import Batch
    from strive_amd.utils.init_optim import run_init_optim
    from strive_amd.utils.adv_gen_optim import run_adv_gen_optim, compute_adv_gen_success
    from strive_amd.utils.sol_optim import run_find_solution_optim, compute_sol_success
    from strive_amd.utils.scenario_gen import detach_embed_info
    from strive_amd.losses.adv_gen_nusc import check_single_veh_coll
    from strive_amd.losses.traffic_model import compute_coll_rate_env
    from strive_amd.planners.planner import PlannerConfig
    from strive_amd.planners.hardcode_goalcond_nusc import HardcodeNuscPlanner, CONFIG_DICT
    n_init, n_fit, n_adv, n_sol = [max(1, int(round(v * scale))) for v in (75, 100, 200, 200)]
    weights = dict(ADV_WEIGHTS)
    weights.update({'init_motion_prior_ext': 0.01, 'init_match_ext': 10.0, 'sol_motion_prior': 0.005, 'sol_coll_veh': 10.0,
                    'sol_coll_env': 10.0, 'sol_motion_prior_ext': 0.001, 'sol_match_ext': 10.0, 'sol_init_z': 0.0})
    lr, sol_future_len = 0.05, 16
    stats = {}

    def step():
        g = batch.clone().to(device)
        mi = map_idx.to(device)
        nrm, att = m.get_normalizer(), m.get_att_normalizer()
        B, NA = int(mi.shape[0]), int(g.past.shape[0])
        ego = torch.zeros((NA,), dtype=torch.bool, device=device)
        ego[g.ptr[:-1].to(device)] = True
        units = 0
        with torch.no_grad():
            emb_att = m.embed(g, mi, env)
        emb = detach_embed_info(emb_att)
        z = emb_att['posterior_out'][0].detach()
        init_traj = g.future_gt[:, :, :4].clone().detach()
        z, fit, _ = run_init_optim(z, init_traj, g.future_vis, 0.1, weights, m, g, env, mi, n_init, emb, emb['prior_out'])
        units += (n_init + 1) * NA * m.FT
        planner = HardcodeNuscPlanner(env, PlannerConfig(**CONFIG_DICT['default']))
        planner.reset(nrm.unnormalize(g.past_gt[:, -1, :]), att.unnormalize(g.lw), g.batch, B, mi)
        plan_t = np.linspace(m.dt, m.dt * m.FT, m.FT)
        agt_ptr = (g.ptr.cpu() - torch.arange(B + 1)).numpy()
        plan0 = planner.rollout(nrm.unnormalize(fit[~ego]).contiguous(), plan_t, agt_ptr, plan_t).to(g.future_gt)
        # a scene whose planner rollout fails here (NaN plan; the reference's planner raises for it) is given up like the scenes
        # whose planner collides after the init stage (:323-356): until the batch is rebuilt it is fitted to its own prediction
        init_failed = planner.check(on_error='report')
        plan0_n = nrm.normalize(plan0)
        if init_failed:
            bad = torch.zeros((B,), dtype=torch.bool, device=device)
            bad[torch.tensor(sorted(init_failed), device=device)] = True
            plan0_n = torch.where(bad.view(-1, 1, 1), fit[ego][:, :, :4].detach(), plan0_n)
        init_traj[ego] = plan0_n
        z, fit, _ = run_init_optim(z, init_traj, g.future_vis, lr, weights, m, g, env, mi, n_fit, emb, emb['prior_out'])
        units += (n_fit + 1) * NA * m.FT
        ptr = g.ptr.cpu().tolist()
        bvalid = []
        for b in range(B):
            coll, _ = check_single_veh_coll(nrm.unnormalize(fit[ptr[b]]), att.unnormalize(g.lw[ptr[b]]),
                                            nrm.unnormalize(fit[ptr[b] + 1:ptr[b + 1]]), att.unnormalize(g.lw[ptr[b] + 1:ptr[b + 1]]))
            bvalid.append(int(np.sum(coll)) == 0 and b not in init_failed)
        stats['scenes'], stats['planner_collides_after_init'] = B, B - sum(bvalid) - len(init_failed)
        stats['scenes_dropped'] = {'planner_failed_at_init': len(init_failed), 'planner_failed_in_adv_loop': 0, 'limits': sorted(
            set(n for v in init_failed.values() for n in v))}
        if sum(bvalid) == 0:
            raise RuntimeError('no scene of the synthetic batch is left after the init stage (planner collides or fails in every one)')
        if sum(bvalid) < B:                     # rebuild the batch without those scenes (reference :330-360)
            keep = torch.tensor(bvalid)
            avalid = torch.zeros((NA,), dtype=torch.bool)
            for b in range(B):
                if bvalid[b]:
                    avalid[ptr[b]:ptr[b + 1]] = True
            dl = g.to_data_list()
            g = Batch.from_data_list([d for b, d in enumerate(dl) if bvalid[b]]).to(device)
            mi = mi[keep.to(device)]
            z, init_traj = z[avalid.to(device)], init_traj[avalid.to(device)]
            B, NA = int(mi.shape[0]), int(g.past.shape[0])
            ego = torch.zeros((NA,), dtype=torch.bool, device=device)
            ego[g.ptr[:-1].to(device)] = True
            with torch.no_grad():
                emb_att = m.embed(g, mi, env)
            emb = detach_embed_info(emb_att)
            ptr = g.ptr.cpu().tolist()
        with torch.no_grad():
            init_pred = m.decode_embedding(z, emb_att, g, mi, env)['future_pred'].detach()
            compute_coll_rate_env(g, mi, init_pred.unsqueeze(1).contiguous(), env, nrm, att, ego_only=False)
        pm, pv = emb['prior_out']
        tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
        planner = HardcodeNuscPlanner(env, PlannerConfig(**CONFIG_DICT['default']))
        cur_z, fin, dec_out, agt, tt = run_adv_gen_optim(z.clone().detach(), lr, weights, m, g, env, mi, n_adv, emb, 'hardcode', tp, op, 2,
                                                         0.0, planner=planner, on_planner_error='drop')
        units += (2 * n_adv + 1) * NA * m.FT
        # scenes the closed loop lost to a failing planner rollout (masked out of the losses on the device from that iteration on):
        # no success test, no solution stage, no scenario file -- what the reference's run of that scene alone would have produced
        lost = set(dec_out.get('scenes_dropped', []))
        stats['scenes_dropped']['planner_failed_in_adv_loop'] = len(lost)
        stats['scenes_dropped']['limits'] = sorted(set(stats['scenes_dropped']['limits']) | set(
            n for v in dec_out.get('planner_failures', {}).values() for n in v))
        stats['scenes_after_adv'] = B - len(lost)
        dl = g.to_data_list()
        ok = [b not in lost and compute_adv_gen_success(fin[ptr[b]:ptr[b + 1]], m, Batch.from_data_list([dl[b]]), int(agt[b]) - ptr[b])
              for b in range(B)]
        stats['adv_succeeded'] = int(sum(ok))
        stats['sol_forced'] = False
        if sum(ok) == 0:
            # random-initialised weights rarely yield a successful attack; so that the run still goes through the solution stage
            # (the reference only enters it for succeeded scenes) the first quarter of the surviving scenes is sent there, and reported
            alive_b = [b for b in range(B) if b not in lost]
            chosen = set(alive_b[:max(1, len(alive_b) // 4)])
            ok = [b in chosen for b in range(B)]
            stats['sol_forced'] = True
        stats['sol_scenes'] = int(sum(ok))
        sol_ok = 0
        if sum(ok) > 0:
            sg = Batch.from_data_list([dl[b] for b in range(B) if ok[b]])
            am = torch.zeros((NA,), dtype=torch.bool, device=device)
            bm = torch.tensor(ok, device=device)
            for b in range(B):
                if ok[b]:
                    am[ptr[b]:ptr[b + 1]] = True
            sNA = int(am.sum())
            sego = torch.zeros((sNA,), dtype=torch.bool, device=device)
            sego[sg.ptr[:-1].to(device)] = True
            semb = {k: (v[am] if torch.is_tensor(v) else (v[0][am], v[1][am])) for k, v in emb.items()}
            stp = (tp[0][bm], tp[1][bm])
            sop = (semb['prior_out'][0][~sego], semb['prior_out'][1][~sego])
            _, sol_traj, _ = run_find_solution_optim(cur_z.clone().detach()[am], fin[am], sol_future_len, lr, weights, m, sg, env, mi[bm],
                                                     n_sol, semb, stp, sop)
            units += n_sol * sNA * (sol_future_len + m.FT) + sNA * m.FT
            sdl = sg.to_data_list()
            sptr = sg.ptr.cpu().tolist()
            smi = mi[bm]
            for b in range(len(sdl)):
                sol_ok += bool(compute_sol_success(sol_traj[sptr[b]:sptr[b + 1]][:, 0:1], m, Batch.from_data_list([sdl[b]]),
                                                   env, smi[b:b + 1]))
        stats['sol_succeeded'] = int(sol_ok)
        stats['units'] = units
        # what the reference writes per scene at adv_scenario_gen.py:465-538 (kept on the device; step.write_scenarios dumps it)
        last.clear()
        last.update(dict(g=g, mi=mi, init=init_pred, fin=fin, z=cur_z, agt=agt, tt=tt, prior=(pm, pv), ptr=ptr, lost=lost))
        return torch.tensor(float(stats['adv_succeeded']))
    last = {}

    def write_scenarios(out_dir):
        import json
        import os
        from strive_amd.utils.scenario_gen import prepare_output_dict
        os.makedirs(out_dir, exist_ok=True)
        dl, ptr = last['g'].to_data_list(), last['ptr']
        written = 0
        for b in range(len(dl)):
            if b in last.get('lost', ()):
                continue
            written += 1
            lo, hi = ptr[b], ptr[b + 1]
            d = prepare_output_dict(dl[b], int(last['mi'][b]), env, m.dt, m, last['init'][lo:hi], last['fin'][lo:hi, 0],
                                    attack_agt=int(last['agt'][b]) - lo, attack_t=int(last['tt'][b]), adv_z=last['z'][lo:hi],
                                    prior_distrib=(last['prior'][0][lo:hi], last['prior'][1][lo:hi]))
            with open(os.path.join(out_dir, 'scene_%04d.json' % b), 'w') as f:
                json.dump(d, f)
        return written
    step.write_scenarios = write_scenarios
    step.stats = stats
    return step, None, batch.to(device), map_idx.to(device), 2


def sample_step_factory(m, env, batch, map_idx, FT, device, NS=20):
    """reference src/adv_scenario_gen.py:160-174 (sample_batched before the feasibility test)"""
    g = batch.to(device)
    mi = map_idx.to(device)

    def step():
        with torch.no_grad():
            out = m.sample_batched(g, mi, env, NS, include_mean=True, nfuture=FT)
        return out['future_pred'].abs().mean()
    return step, None, g, mi, NS


TRAIN_WEIGHTS = {'recon': 1.0, 'kl': 0.004, 'coll_veh_prior': 0.05, 'coll_env_prior': 0.1}      # train_traffic.cfg:17-21


def train_step_factory(m, env, batch, map_idx, FT, device):
    """reference src/train_traffic.py:103-131 through strive_amd.distributed.DataParallelTrainer (one rank = plain step)."""
    from strive_amd.distributed import DataParallelTrainer
    from strive_amd.losses.traffic_model import TrafficModelLoss
    g = batch.to(device)
    mi = map_idx.to(device)
    m.train()
    # (torch's default multi-tensor Adam like the reference's optim.Adam, train_traffic.py:92-95.  Adam(fused=True) was measured no
    #  faster here -- 21.3 vs 21.4 ms per step, profiles/r05_ab_train_small_kernels.json -- and does not bump the parameters' version
    #  counters; DataParallelTrainer.step announces every update to the pack caches itself: params.parameters_changed().
    #  STRIVE_BENCH_ADAM_FUSED=1 selects it.)
    fused = torch.device(device).type == 'cuda' and os.environ.get('STRIVE_BENCH_ADAM_FUSED', '0') == '1'
    opt = torch.optim.Adam(m.parameters(), lr=1e-5, **({'fused': True} if fused else {}))
    tr = DataParallelTrainer(m, TrafficModelLoss(TRAIN_WEIGHTS, m.get_normalizer(), m.get_att_normalizer()), opt)

    def step():
        out = tr.step(g, mi, env)
        if out is None:
            raise RuntimeError('training step was skipped: %r' % (tr.last_error,))
        return out['global_loss'][0]
    return step, None, g, mi, 2


# ------------------------------------------------------------------------------------------------
# side measurements
# ------------------------------------------------------------------------------------------------

TRAFFIC_FILES = ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json', 'r02_traffic.json', 'r01_traffic.json')
UTIL_FILES = ('r06_util.json', 'r05_util.json')
PEAK_LDS_BYTES_PER_CLK_CU = 256.0   # /opt/skills/guides/MI355X_MICROARCH.md, LDS section: ds_read_b64 / b128, conflict-free (157 TB/s at 2.4 GHz)


def _measured_traffic(kernel_name, with_source=False):
    """HBM bytes per launch of a kernel from the COMMITTED PMC passes (profiles/r0N_traffic.json, written by
    profiles/make_traffic.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command) -- not measured in this
    run: counters need their own rocprofv3 passes.  None if not collected for this kernel."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(REPO, 'profiles', name)) as f:
                ent = json.load(f).get(kernel_name)
            if ent is not None:
                return (ent.get('bytes_per_launch'), 'profiles/' + name) if with_source else ent.get('bytes_per_launch')
        except (OSError, ValueError):
            continue
    return (None, None) if with_source else None


def _measured_util(kernel_name, launch_s):
    """The two on-chip roofs of a kernel from the COMMITTED counter passes (profiles/r0N_util.json, written by profiles/make_util.py
    from rocprofv3 --pmc SQ_LDS_IDX_ACTIVE ... / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... runs of this same command):
    ``lds`` = {frac: share of the launch during which a CU's LDS array is busy (chip average), bank_conflict_share, GBps / bytes:
    that share expressed against the guide's 256 B/clk/CU read peak at the launch's own clock, peak}; ``mfma_pipe`` = share of the
    launch during which a SIMD's matrix pipe is busy.  None if not collected."""
    for name in UTIL_FILES:
        try:
            with open(os.path.join(REPO, 'profiles', name)) as f:
                ent = json.load(f).get(kernel_name)
        except (OSError, ValueError):
            continue
        if ent is None:
            continue
        clk = ent['cycles_per_launch'] / launch_s if launch_s else 0.0          # core clock of this launch (cycles of the counter run / live duration)
        peak = PEAK_LDS_BYTES_PER_CLK_CU * 256 * clk / 1e9
        return {'lds': {'frac': ent['lds_array_busy_frac'], 'bank_conflict_share': ent['lds_bank_conflict_share_of_array_cycles'],
                        'bytes': int(ent['lds_array_busy_frac'] * peak * 1e9 * launch_s), 'GBps': round(ent['lds_array_busy_frac'] * peak, 1),
                        'peak': round(peak, 1), 'unit': 'GB/s (LDS-array cycles x 256 B/clk/CU at the launch clock of %.2f GHz)' % (clk / 1e9)},
                'mfma_pipe': {'frac': ent['mfma_pipe_busy_frac'], 'unit': 'share of the launch a SIMD matrix pipe is busy (chip average)'},
                'source': 'profiles/' + name}
    return None


def _event_time(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def time_dominant_kernel(m, env, g, mi, emb, device, reps=20):
    """Time each CNN kernel in isolation (events on torch's current stream = the launching stream) and return
    the roofline record of the one that takes the most time per rollout step."""
    from strive_amd import ops, _lib as L
    lib = L.get_lib()
    N = min(512, g.past.shape[0])      # = the size of the launches inside the timed closure (one CNN chunk)
    pos = g.past[:N, -1, :4].contiguous()
    mapix = mi[g.batch][:N].to(torch.int32).contiguous()
    mp = ops._map_pack(env, device)
    cnn = ops.cnn_pack(m)
    wsb = lib.query('strive_map_cnn_workspace_bytes', N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    feat = torch.empty((N, 64), device=device)
    nm = m.normalizer
    mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
    st = L.stream_ptr(pos)
    lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
    torch.cuda.synchronize()
    # Each kernel alone, twenty launches back to back between two events.  (Tried in round 6: the five kernels in the closure's order
    # with an event between every two -- an event record drains and releases the caches at system scope, which adds 15-30 us to every
    # kernel (conv2 240 us against 202 in the rocprofv3 trace of the same run); the isolated loop is 4-7 % above the in-closure average
    # because a kernel that follows itself starts while its own 0.25-0.5 GB of output is still being written back.)
    times = []
    for layer in CONV_LAYER_IDS:
        times.append(_event_time(lambda: lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4,
                                                  std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st), reps))
    dom = max(range(NK), key=lambda l: times[l])
    mult, peak, mdt = CONV_ISSUE[dom]
    alg = CONV_FLOPS[dom] * N / times[dom] / 1e12          # algorithmic (fp32-equivalent) TFLOP/s
    ach = alg * mult                                       # matrix-core FLOP/s actually issued
    traffic, traffic_source = _measured_traffic(CONV_NAMES[dom], with_source=True)
    # Which roof binds?  Since the fp16 x 3 scheme halved the matrix work these kernels sit closer to the HBM roof than to the
    # matrix roof: report the one with the larger fraction as `bound` and keep the other beside it.  HBM bytes per launch =
    # the algorithmic minimum (input once + output once); `traffic` = the PMC-measured bytes.
    hbm_gbs = CONV_BYTES[dom] * N / times[dom] / 1e9
    f_mfma, f_hbm = ach / peak, hbm_gbs / PEAK_HBM_GBS
    if f_hbm >= f_mfma:
        rec = {'bound': 'hbm', 'kernel': CONV_NAMES[dom], 'achieved': round(hbm_gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
               'frac': round(f_hbm, 4), 'traffic': traffic, 'traffic_source': traffic_source,
               'achieved_measured_traffic_GBps': None if traffic is None else round(traffic / times[dom] / 1e9, 1),
               'frac_measured_traffic': None if traffic is None else round(traffic / times[dom] / 1e9 / PEAK_HBM_GBS, 4)}
    else:
        rec = {'bound': 'mfma', 'kernel': CONV_NAMES[dom], 'achieved': round(ach, 3), 'peak': peak, 'unit': 'TFLOP/s',
               'frac': round(f_mfma, 4), 'traffic': traffic, 'traffic_source': traffic_source}
    rec.update({'mfma': {'issued_tflops': round(ach, 3), 'peak': peak, 'frac': round(f_mfma, 4), 'dtype': mdt,
                         'products_per_fp32_product': mult},
                'hbm': {'algorithmic_GBps': round(hbm_gbs, 1), 'peak': PEAK_HBM_GBS, 'frac': round(f_hbm, 4)},
                'algorithmic_tflops': round(alg, 3),
                'launch_us': round(times[dom] * 1e6, 2), 'agents_per_launch': N,
                'all_layers_us': [round(t * 1e6, 2) for t in times],
                'all_layers_algorithmic_tflops': [round(CONV_FLOPS[l] * N / times[l] / 1e12, 2) for l in range(NK)]})
    # byte-bound view of the same launches: algorithmic bytes (input once + output once) / live duration
    bw = {}
    for l in range(NK):
        gbs = CONV_BYTES[l] * N / times[l] / 1e9
        bw[CONV_NAMES[l]] = {'algorithmic_bytes': CONV_BYTES[l] * N, 'us': round(times[l] * 1e6, 2), 'GBps': round(gbs, 1),
                             'frac_of_hbm_peak': round(gbs / PEAK_HBM_GBS, 4), 'traffic': _measured_traffic(CONV_NAMES[l])}
    rec['bandwidth_kernels'] = bw
    # the on-chip roofs (round 5): which resource is busiest inside each kernel -- HBM (algorithmic bytes above), the LDS arrays
    # or the matrix pipes (committed counter passes); `binding` names the largest of the three fractions
    util = {}
    for l in range(NK):
        u = _measured_util(CONV_NAMES[l], times[l])
        if u is None:
            continue
        fr = {'hbm': round(CONV_BYTES[l] * N / times[l] / 1e9 / PEAK_HBM_GBS, 4), 'lds': u['lds']['frac'], 'mfma_pipe': u['mfma_pipe']['frac']}
        u['fractions'] = fr
        u['binding'] = max(fr, key=fr.get)
        util[CONV_NAMES[l]] = u
    if CONV_NAMES[dom] in util:
        rec['lds'] = util[CONV_NAMES[dom]]['lds']
        rec['mfma_pipe'] = util[CONV_NAMES[dom]]['mfma_pipe']
        rec['binding_resource'] = util[CONV_NAMES[dom]]['binding']
    rec['on_chip_roofs'] = util
    return rec


# the data-gradient kernels of the CNN backward (csrc/map_cnn_bwd_mfma.h, dgrad_mfma_kernel<L>, L = 1..5 = d conv2 .. d conv6 input):
# (Cin, Cout, k, IH, OH) per layer; algorithmic FLOPs per sample 2 Cin Cout k^2 OH^2, bytes = dy read once + d input written once
DGRAD_SHAPES = {1: (16, 32, 5, 125, 61), 2: (32, 64, 5, 61, 29), 3: (64, 64, 3, 29, 14), 4: (64, 128, 3, 14, 6), 5: (128, 128, 3, 6, 2)}
PEAK_BF16_MFMA_TFLOPS = 2516.0     # same guide: bf16 dense = f16 dense


def time_train_dominant_kernel(m, env, g, mi, device, reps=20):
    """The training step's largest single launch (profiles/r0N_train_kernel_stats.txt): the data gradient of conv2,
    cnnbwd::dgrad_mfma_kernel<1>, on one backward chunk of 256 samples -- the size the rollout's CNN backward launches (64 agents x
    11 re-encoded steps in chunks of 256).  strive_map_cnn_bwd fills the workspace once, then strive_map_cnn_bwd_bench_dgrad times
    every data-gradient kernel alone (events on the launching stream).  Operands are two bf16 pieces, three products per fp32
    product; bytes = the output gradient read once + the input gradient written once."""
    from strive_amd import ops, _lib as L
    lib = L.get_lib()
    N = 256
    NA = g.past.shape[0]
    idx = torch.arange(N, device=device) % NA
    pos = g.past[idx, -1, :4].contiguous()
    mapix = mi[g.batch][idx].to(torch.int32).contiguous()
    mp = ops._map_pack(env, device)
    cnn = ops.cnn_pack(m)
    nm = m.normalizer
    mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
    st = L.stream_ptr(pos)
    gen = torch.Generator(device='cpu').manual_seed(7)
    d_feat = torch.randn((N, 64), generator=gen).to(device)
    dp = torch.zeros(lib.query('strive_map_cnn_param_count'), device=device)
    wsb = lib.query('strive_map_cnn_bwd_workspace_bytes', N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    lib.call('strive_map_cnn_bwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(d_feat), L.ptr(dp),
             L.ptr(ws), wsb, st)
    torch.cuda.synchronize()
    layers = sorted(DGRAD_SHAPES)
    times = {l: _event_time(lambda: lib.call('strive_map_cnn_bwd_bench_dgrad', l, N, L.ptr(ws), wsb, st), reps) for l in layers}
    dom = max(layers, key=lambda l: times[l])
    cin, cout, k, ih, oh = DGRAD_SHAPES[dom]
    flops = 2.0 * cin * cout * k * k * oh * oh * N
    by = (cout * oh * oh + cin * ih * ih) * 4 * N
    issued = 3.0 * flops / times[dom] / 1e12
    gbs = by / times[dom] / 1e9
    f_mfma, f_hbm = issued / PEAK_BF16_MFMA_TFLOPS, gbs / PEAK_HBM_GBS
    rec = {'kernel': 'cnnbwd::dgrad_mfma_kernel<%d>' % dom, 'launch_us': round(times[dom] * 1e6, 2), 'samples_per_launch': N,
           'traffic': None, 'algorithmic_tflops': round(flops / times[dom] / 1e12, 3),
           'mfma': {'issued_tflops': round(issued, 3), 'peak': PEAK_BF16_MFMA_TFLOPS, 'frac': round(f_mfma, 4), 'dtype': 'bf16',
                    'products_per_fp32_product': 3},
           'hbm': {'algorithmic_GBps': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'frac': round(f_hbm, 4), 'algorithmic_bytes': by},
           'all_dgrad_layers_us': {str(l): round(times[l] * 1e6, 2) for l in layers},
           'note': 'largest single launch of the training step; the step itself is a flat profile (no kernel above 7 % of the GPU '
                   'time, profiles/r0N_train_kernel_stats.txt)'}
    if f_hbm >= f_mfma:
        rec.update({'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(f_hbm, 4)})
    else:
        rec.update({'bound': 'mfma', 'achieved': round(issued, 3), 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(f_mfma, 4)})
    return rec


def time_bandwidth_kernels(m, env, g, mi, device, reps=20):
    """Achieved GB/s of the byte-bound kernels outside the CNN: the stand-alone raster crop, the
    vehicle-collision penalty and the off-road collision point, each on this batch's own sizes."""
    from strive_amd import ops
    from strive_amd.losses.adv_gen_nusc import VehCollLoss
    out = {}
    N = min(512, g.past.shape[0])
    unn = m.get_normalizer().unnormalize
    pos = unn(g.past[:N, -1, :])[:, :4].contiguous()
    mapix = mi[g.batch][:N]
    t = _event_time(lambda: ops.map_crop(env, pos, mapix), reps)
    by = N * 4 * 256 * 256 * 2          # gather 1 B per crop pixel, write 1 B
    out['map_crop_u8_kernel'] = {'algorithmic_bytes': by, 'us': round(t * 1e6, 2), 'GBps': round(by / t / 1e9, 1),
                                 'frac_of_hbm_peak': round(by / t / 1e9 / PEAK_HBM_GBS, 4)}
    NA = g.past.shape[0]
    T = 48
    traj = unn(g.future_gt[:, :12, :])[:, :, :4].repeat(1, 4, 1).contiguous()
    veh_att = m.get_att_normalizer().unnormalize(g.lw)
    vl = VehCollLoss(veh_att, buffer_dist=0.2, ptr=g.ptr)
    t = _event_time(lambda: ops.veh_coll_penalties(traj, vl.setup), reps)
    P = int(((g.ptr[1:] - g.ptr[:-1]) ** 2).sum())
    by = NA * T * 16 + T * P * 6        # trajectories read once; pen (4 B) + hit + amin (1 B each) written per pair slot
    out['veh_coll_fwd_kernel'] = {'algorithmic_bytes': by, 'us': round(t * 1e6, 2), 'GBps': round(by / t / 1e9, 1),
                                  'frac_of_hbm_peak': round(by / t / 1e9 / PEAK_HBM_GBS, 4), 'pairs': P, 'T': T}
    # one GNN message pass of the decoder net (node1 -> edge -> node2 kernels, what every rollout step runs) on this batch's
    # scenes: algorithmic bytes = node features in, embedding out, the edge layer-0 partials and node embeddings written and read
    # back once (P, Q, X, aggregated messages), the relative poses' inputs, and the weights once
    net = m.decoder_net
    F_in = net.mlp_in.net[0].in_features if hasattr(net, 'mlp_in') else None
    if F_in is not None:
        keep = (g.x if 'x' in g else None, g.pos if 'pos' in g else None)
        g.x = torch.randn((NA, F_in), device=device)
        g.pos = g.past[:, -1, :4].contiguous()
        t = _event_time(lambda: ops.gnn_forward(net, g), reps)
        D_out = int(ops.gnn_forward(net, g).shape[-1])
        Dn = net.mlp_in.net[-1].out_features
        nparam = sum(p.numel() for p in net.parameters())
        by = (NA * (F_in + D_out + 4) + 2 * NA * (2 * 128 + 2 * Dn)) * 4 + nparam * 4
        out['gnn_step_fwd (node1 + edge + node2)'] = {'algorithmic_bytes': by, 'us': round(t * 1e6, 2), 'GBps': round(by / t / 1e9, 1),
                                                      'frac_of_hbm_peak': round(by / t / 1e9 / PEAK_HBM_GBS, 4), 'agents': NA,
                                                      'edges': P - NA, 'note': 'three launches; latency-bound chains of small layers'}
        if keep[0] is not None:
            g.x = keep[0]
        if keep[1] is not None:
            g.pos = keep[1]
    cars = traj.reshape(NA * T, 4)
    lw = veh_att.unsqueeze(1).expand(NA, T, 2).reshape(NA * T, 2).contiguous()
    mp = mi[g.batch].unsqueeze(1).expand(NA, T).reshape(-1)
    gl, gw = 39, 16
    t = _event_time(lambda: ops.coll_point(env, cars, lw, mp, gl, gw), reps)
    by = NA * T * (gl * gw + 16 + 8 + 12)
    out['coll_point_kernel'] = {'algorithmic_bytes': by, 'us': round(t * 1e6, 2), 'GBps': round(by / t / 1e9, 1),
                                'frac_of_hbm_peak': round(by / t / 1e9 / PEAK_HBM_GBS, 4)}
    return out


def time_planner(step, device, reps=5):
    """The planner rollout of the closed loop on its own (events on the launching stream), on the futures of the last closure."""
    c = step.closure
    with torch.no_grad():
        out = c.model.decode_embedding(c.collated(), c.embed_info, c.scene_graph, c.map_idx, c.map_env, nfuture=c.future_len)
    t = _event_time(lambda: c.plan(out['future_pred']), reps, warm=1)
    c.planner.check(on_error='report')
    lg = next(iter(c.map_env.lane_graphs.values()))
    return {'ms_per_rollout': round(t * 1e3, 3), 'scenes': int(c.scene_graph.ptr.shape[0] - 1), 'planner_steps': 31,
            'lane_graph_nodes': int(lg['xy'].shape[0]), 'lane_graph_edges': int(lg['edges'].shape[0]), 'arithmetic': 'float64',
            'note': 'reference / oracle: host numpy, about 25 ms per scene and planner step'}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _one_socket():
    """(physical cores, hardware threads) of ONE socket -- what the reference's single-socket CPU run would use."""
    try:
        sockets, cpus, cores = set(), 0, 0
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    sockets.add(line.split(':', 1)[1].strip())
                elif line.startswith('processor'):
                    cpus += 1
                elif line.startswith('cpu cores') and not cores:
                    cores = int(line.split(':', 1)[1])
        if sockets and cpus:
            threads = max(1, cpus // len(sockets))
            return (cores or threads), threads
    except (OSError, ValueError):
        pass
    return torch.get_num_threads(), torch.get_num_threads()


def _one_socket_threads():
    return _one_socket()[1]


def cpu_baseline(FT, scenes=4, agents=16, raster_px=4096, timed=3, warm=1):
    """The oracle's refine closure (decode + AvoidCollLoss + backward) on `scenes` scenes of the same workload, in the
    reference's structure: materialised fp32/fp64/int64 coordinate tensors per crop, per-edge MLP, and -- because the
    reference's parameters require grad and encode_map runs with grad enabled -- autograd through the map CNN including its
    (unused) weight gradients.  1 warm-up closure + `timed` timed closures."""
    from strive_amd import synth
    from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors
    from strive_amd.models.traffic_model import TrafficModel
    from oracle.model import OracleTrafficModel
    from oracle.geometry import Normalizer
    from oracle.losses import AvoidColl
    from oracle import mapenv as _omap
    threads = _one_socket_threads()
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    old_struct = _omap.REFERENCE_CHANNEL_STRUCTURE
    # the crop's coordinate pipeline with the reference's tensor structure (one grid per raster channel, nuscenes_utils.py:217-230,
    # 253-262): 4 x the arithmetic of the oracle's channel-free form, same values -- the CPU number carries the reference's cost
    _omap.REFERENCE_CHANNEL_STRUCTURE = True
    try:
        m = TrafficModel(4, 12, 256, 2)
        sd = synth.fill_state_dict(m.state_dict())
        for v in sd.values():
            v.requires_grad_(True)          # reference: model parameters require grad during the optimisation loops
        orc = OracleTrafficModel(sd, Normalizer(*state_norm_tensors()), Normalizer(*att_norm_tensors()), NUSC_BIKE_PARAMS)
        raster, dx = synth.make_raster(raster_px, raster_px)
        extent = raster_px * 0.25
        env = synth.SyntheticMapEnv(raster, dx)
        batch, map_idx = build_batch([(agents, 'bench/r0/%d' % b) for b in range(scenes)], 2, raster_px)
        with torch.no_grad():
            emb = orc.embed(batch, map_idx, env)
        z0 = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='bench/z')
        z = z0.clone().requires_grad_(True)
        opt = torch.optim.Adam([z], lr=0.05)
        lf = AvoidColl(REFINE_WEIGHTS, orc.get_att_normalizer().unnormalize(batch.lw), map_idx[batch.batch], env, z0.clone(),
                       veh_coll_buffer=0.2)

        def closure():
            opt.zero_grad()
            for v in sd.values():
                v.grad = None
            pred = orc.decode_embedding(z, emb, batch, map_idx, env, nfuture=FT)['future_pred']
            ld = lf(orc.get_normalizer().unnormalize(pred), z, emb['prior_out'])
            ld['loss'].backward()
            opt.step()
        for _ in range(warm):
            closure()
        each = []
        for _ in range(timed):
            t0 = time.perf_counter()
            closure()
            each.append(time.perf_counter() - t0)
        dt = sum(each) / timed
    finally:
        torch.set_num_threads(old_threads)
        _omap.REFERENCE_CHANNEL_STRUCTURE = old_struct
    n = scenes * agents * FT
    return {'value': round(n / dt, 2), 'unit': 'agent*timesteps/s', 'cores': _one_socket()[0], 'threads': threads, 'kind': 'port',
            'cpu': _cpu_model(), 'closure_s': [round(v, 2) for v in each],
            'sample': '%d scenes x %d agents, FT=%d: %d warm-up + %d timed refine closures (decode + AvoidCollLoss + backward '
                      'incl. the CNN weight gradients the reference computes + Adam) of the CPU oracle with the reference\'s per-channel '
                      'crop coordinates (nuscenes_utils.py:217-230), %.1f s per closure; '
                      'torch threads = the hardware threads of one socket, `cores` = its physical cores' % (scenes, agents, FT, warm, timed, dt)}


def cpu_baseline_record(FT, full=False):
    """The bench line's `cpu_baseline`.  SURVEY 8(d) quotes the CPU at C2 (32 scenes x 16 agents), where one closure of the
    oracle takes minutes; the default run therefore times two bounded samples of the same workload -- 4 x 16 agents (1 warm-up + 1 timed
    closure) and 16 x 16 agents (2 timed closures) -- reports the LARGER sample as `value` and states both, so the trend
    towards C2 is visible (it differs by host: 77 -> 31 agent*timesteps/s from 4 to 32 scenes on the 8-vCPU survey container,
    79 -> 90 from 4 to 16 scenes on the 128-thread EPYC of the GPU boxes).  ``--cpu-baseline-full`` times C2 itself (1 warm-up +
    2 timed closures, several minutes); profiles/r06_cpu_baseline_c2.json holds that run (75.6 agent*timesteps/s)."""
    if full:
        rec = cpu_baseline(FT, scenes=32, agents=16, timed=2)
        rec['sample'] = 'C2 itself: ' + rec['sample']
        return rec
    # (with the reference's per-channel crop structure a 16 x 16 closure takes ~65 s on the GPU boxes' EPYC: the small sample is also
    # the process's warm-up -- thread pool, allocator, convolution primitives -- and the large one times its first two closures, both
    # listed in `closure_s`, so that the default run stays at about three minutes of CPU work)
    small = cpu_baseline(FT, scenes=4, agents=16, timed=1, warm=1)
    large = cpu_baseline(FT, scenes=16, agents=16, timed=2, warm=0)
    rec = dict(large)
    rec['sample'] = ('bounded samples of C2 (32 x 16 agents): %s || %s.  `value` is the 16 x 16 sample (4 x 16: %.1f, 16 x 16: %.1f '
                     'agent*timesteps/s); C2 itself, timed with --cpu-baseline-full, is in profiles/r06_cpu_baseline_c2.json' %
                     (small['sample'], large['sample'], small['value'], large['value']))
    rec['samples'] = [{'scenes': 4, 'agents': 16, 'value': small['value'], 'closure_s': small['closure_s']},
                      {'scenes': 16, 'agents': 16, 'value': large['value'], 'closure_s': large['closure_s']}]
    return rec


# ------------------------------------------------------------------------------------------------
# launch
# ------------------------------------------------------------------------------------------------

def pin_host_threads(local, nlocal):
    """N ranks of one node share its host cores: every rank enqueues ~200-700 launches per step from Python (1.6-5.6 ms of host
    work per closure at one rank), so they must not fight over cores or oversubscribe them with intra-op threads.  Each local
    rank gets an equal, disjoint slice of the cores the job may use and at most 8 torch threads."""
    info = {'affinity': None, 'torch_threads': torch.get_num_threads()}
    if nlocal <= 1:
        return info
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, len(cpus) // nlocal)
        mine = cpus[local * per:(local + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        info['affinity'] = '%d cores (%d-%d)' % (len(mine), mine[0], mine[-1])
        info['cores'] = list(mine)
        torch.set_num_threads(max(1, min(8, len(mine))))
    except (AttributeError, OSError):
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // nlocal)))
    info['torch_threads'] = torch.get_num_threads()
    return info


def gather_ranks(use_dist, world, device, dt_local, units_local, rank_record):
    """(max over ranks of the timed region, units of all ranks, every rank's record, what the collective backend reports)"""
    dt, units, per_rank = dt_local, units_local, [rank_record]
    backend = {'rccl_world_size': None, 'rccl_backend': None}
    if use_dist:
        import torch.distributed as dist
        tt = torch.tensor([dt_local], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        uu = torch.tensor([units_local], device=device, dtype=torch.float64)
        dist.all_reduce(uu, op=dist.ReduceOp.SUM)
        units = int(uu.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_record)
        backend = {'rccl_world_size': dist.get_world_size(), 'rccl_backend': dist.get_backend()}
    backend['distinct_devices'] = len(set(r[3] for r in per_rank))
    return dt, units, per_rank, backend


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=['refine', 'adv', 'sharded4096', 'train', 'sample', 'full'], default='refine')
    ap.add_argument('--scenes', type=int, default=32)
    ap.add_argument('--agents', type=int, default=16)
    ap.add_argument('--total-agents', type=int, default=0, help='adv / sharded4096: agents in the batch (default 512 / 4096)')
    ap.add_argument('--nc', type=int, default=0, help='semantic classes (default 2; 5 for sharded4096 = reduce_cats)')
    ap.add_argument('--ft', type=int, default=0, help='rollout steps per closure (default: 16 for refine = '
                                                      'refine_traffic_optim.cfg samp_future_len, 12 otherwise)')
    ap.add_argument('--planner', choices=['ego', 'hardcode'], default='ego',
                    help="adv: 'ego' = open loop against the recorded ego future, 'hardcode' = closed loop against the rule-based "
                         "planner (adv_gen_rule_based.cfg)")
    ap.add_argument('--on-planner-error', choices=['drop', 'raise'], default='drop',
                    help="closed loop: 'drop' = a scene whose planner rollout fails is masked out of the losses on the device and "
                         "reported (scenes_dropped); 'raise' = the reference's behaviour, the run ends (planner overlapped under the "
                         "adversarial half of the iteration)")
    ap.add_argument('--iters', type=float, default=1.0, help='full: scale of the iteration counts 75 / 100 / 200 / 200')
    ap.add_argument('--scenario-out', default='', help='full: directory for the scenario JSON of every scene (reference '
                    'src/adv_scenario_gen.py:465-538 -> prepare_output_dict), written from the device tensors after the timed region')
    ap.add_argument('--raster', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-full', action='store_true', help='time the CPU oracle on C2 itself (32 x 16 agents; ~10 minutes)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--dry-run', action='store_true', help='CPU/gloo: join the ranks, build the scene partition, no kernels')
    ap.add_argument('--fail-rank', type=int, default=-1, help='dry run only (test hook): this rank raises before the collectives')
    args = ap.parse_args(argv)
    if not args.nc:
        args.nc = 5 if args.workload == 'sharded4096' else 2
    if not args.ft:
        args.ft = 16 if args.workload == 'refine' else 12
    return args


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # not under a launcher: start the N ranks ourselves, one process per GPU, rendezvous on 127.0.0.1
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    use_dist = world > 1
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    if local >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    host_pin = pin_host_threads(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build(verbose=False)
    if use_dist:
        dist.barrier()

    own, desc, scaling = workload_scenes(args, rank, world)
    lane_graph = None
    if args.workload == 'full':
        args.planner = 'hardcode'
    m = build_model(device, args.nc, lane_keeping=args.planner == 'hardcode')
    if args.planner == 'hardcode':
        if args.workload not in ('adv', 'full'):
            raise SystemExit("bench.py: --planner hardcode belongs to --workload adv / full")
        from strive_amd import synth
        lane_graph = synth.make_lane_graph(extent=args.raster * 0.25)
    env = build_env(args.raster, device, lane_graph)
    batch, map_idx = build_batch(own, args.nc, args.raster, lane_graph=lane_graph)
    factory = {'refine': refine_closure_factory, 'train': train_step_factory, 'sample': sample_step_factory}.get(args.workload, adv_closure_factory)
    if args.workload == 'full':
        step, emb, g, mi, rollouts = full_pipeline_factory(m, env, batch, map_idx, args.ft, device, scale=args.iters)
    elif factory is adv_closure_factory:
        step, emb, g, mi, rollouts = factory(m, env, batch, map_idx, args.ft, device, on_planner_error=args.on_planner_error)
    else:
        step, emb, g, mi, rollouts = factory(m, env, batch, map_idx, args.ft, device)
    for _ in range(args.warmup):
        step()
    hip_graph = bool(getattr(step, 'enabled', False))
    if hip_graph:
        # the iteration is captured after GraphedIteration's eager calls: finish that before the timed region starts
        while step.graph is None and step.enabled:
            step()
        hip_graph = step.graph is not None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    for _ in range(args.steps):
        loss = step()
    dt_host = time.perf_counter() - t0          # all work enqueued (diagnostic: host-bound when ~ dt_local)
    dt_host_cpu = time.thread_time() - c0       # CPU time of the enqueueing thread: the wall time above also holds the waits for a
    barrier()                                   # free slot in the hardware queue once the GPU is the bottleneck
    dt_local = time.perf_counter() - t0
    gve = step.graph_vs_eager() if (hip_graph and getattr(step, 'graph_vs_eager', None) is not None) else None
    if args.workload == 'full' and args.scenario_out and rank == 0:
        step.stats['scenarios_written'] = step.write_scenarios(args.scenario_out)
    planner_ms = None
    scenes_dropped = None
    if getattr(step, 'planner', None) is not None and args.workload != 'full':
        # capacity / range status of every planner rollout of the run, per scene: a scene whose rollout failed (an object pushed off
        # its lane by these random-weight scenes: the case in which the reference's planner raises) was masked out of the losses on
        # the device from that iteration on (AdvClosure on_planner_error='drop'); the others ran on as in a batch without it
        failed_scenes = step.planner.check(on_error='report' if args.on_planner_error == 'drop' else 'raise')
        scenes_dropped = {'count': len(failed_scenes), 'of': int(g.ptr.shape[0] - 1), 'scenes': sorted(failed_scenes),
                          'agents': int(sum(int(g.ptr[b + 1] - g.ptr[b]) for b in failed_scenes)),
                          'limits': sorted(set(n for v in failed_scenes.values() for n in v)),
                          'note': 'units count every scene of the batch: a dropped scene is still rolled out and planned, only its '
                                  'loss terms and gradients are masked'}
        planner_ms = time_planner(step, device)
    dt = dt_local
    NA = int(g.past.shape[0])
    units_local = rollouts * NA * args.ft * args.steps
    if args.workload == 'full':
        units_local = step.stats['units'] * args.steps
    units = units_local
    props = torch.cuda.get_device_properties(device)
    dev_id = str(getattr(props, 'uuid', '')) or '%s/pci%s' % (props.name, getattr(props, 'pci_bus_id', local))
    dt, units, per_rank, backend = gather_ranks(use_dist, world, device, dt_local, units_local,
                                                [rank, NA, round(dt_local / args.steps * 1e3, 3), dev_id, local, host_pin.get('cores')])
    if args.planner == 'hardcode':
        closure_adv = ('closed-loop adversarial closure: 2 x decode_embedding(nfuture=%d) with complementary detach + '
                       'HardcodeNuscPlanner.rollout (31 planner steps per scene, device) + TgtMatchingLoss + AdvGenLoss + backward + Adam')
    else:
        closure_adv = ('adversarial closure: 2 x decode_embedding(nfuture=%d, ext_future) with complementary detach + '
                       'TgtMatchingLoss + AdvGenLoss + backward + Adam')
    closure = {'refine': 'refine closure: decode_embedding(nfuture=%d) + AvoidCollLoss + backward + Adam',
               'adv': closure_adv,
               'sharded4096': 'adversarial closure: 2 x decode_embedding(nfuture=%d, ext_future) + TgtMatchingLoss + '
                              'AdvGenLoss + backward + Adam',
               'full': 'adv_scenario_gen.py:262-462 for one batch: embed, init optimisation 75 + 100 iterations around a planner rollout, '
                       'closed-loop adversarial optimisation 200 iterations (planner hardcode, FT %d), IoU success tests, solution '
                       'optimisation 200 iterations of the succeeded scenes',
               'sample': 'TrafficModel.sample_batched(NS=20, include_mean=True, nfuture=%d) under no_grad: embed + one joint '
                         'rollout of NA x 20 rows, forward only',
               'train': 'training step: TrafficModel.forward(future_sample=True) (2 rollouts of %d steps) + TrafficModelLoss + '
                        'backward to 174 parameter tensors + gradient all-reduce + Adam'}[args.workload] % args.ft
    cfg_ref = {'refine': 'BASELINE.json configs[1]', 'adv': 'BASELINE.json configs[2]', 'sharded4096': 'BASELINE.json configs[4]',
               'train': 'BASELINE.json configs[3]', 'sample': 'SURVEY 8 a18: adv_scenario_gen.py:160-174',
               'full': 'BASELINE.json configs[2], end to end'}
    # which decoder kernels served the rollouts of this batch (include/strive_hip.h strive_rollout_scene_resident)
    rollout_kernels = None
    try:
        from strive_amd import ops as _ops, _lib as _L
        pk = [v[1] for k, v in m.__dict__.get('_strive_packs', {}).items() if isinstance(k, tuple) and k and k[0] == 'dec']
        sc = _ops.scene_info(g).pack(1)
        if pk:
            kind = _L.get_lib().query('strive_rollout_scene_resident', pk[0].ref(), sc.ref())
            rollout_kernels = {1: 'scene-resident', 2: 'per-phase; forward node phases on the scene kernel in 16-row tiles'}.get(kind, 'per-phase')
            if rollout_kernels == 'scene-resident':
                # the reverse sweep's form (csrc/rollout.hip strive_rollout_bwd: stepwise from 3 edge chunks per scene on, STRIVE_SWEEP_STEP)
                mx = int(sc.struct.max_n)
                chunks = (mx * (mx - 1) + 63) // 64
                k_env = os.environ.get('STRIVE_SWEEP_STEP')
                K = int(k_env) if k_env is not None else (chunks if chunks >= 3 else 0)
                rollout_kernels += ('; reverse sweep: one launch per step, %d workgroups per scene' % min(K, 4)) if K >= 1 else \
                    '; reverse sweep: one launch, one workgroup per scene'
    except Exception:
        rollout_kernels = None
    out = {
        'metric': 'adv-optim agent*timesteps/sec (decoder fwd+bwd)',
        'value': round(units / dt, 1), 'unit': 'agent*timesteps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s, %s (%s)' % (closure, desc, cfg_ref[args.workload]),
                   'agents_per_gpu': NA, 'FT': args.ft, 'NC': args.nc, 'rollouts_per_closure': rollouts,
                   'units_note': (None if rollouts != 2 or args.workload == 'full' else
                                  'reference-equivalent units: the reference decodes twice per iteration (complementary detach, '
                                  'adv_gen_optim.py:120-131) and both decodes are counted; executed here: 1 forward rollout + 2 '
                                  'reverse sweeps over its tape (ops._RolloutPairFn)'),
                   'raster': '%dx%d x4 uint8 @0.25 m' % (args.raster, args.raster),
                   'parallelism': 'scene-sharded replicas x%d' % world,
                   'arithmetic': ('fp32 everywhere; the map CNN on the fp16 matrix cores with two-piece round-to-nearest operand splits '
                                  '(3 products per fp32 product, dropped terms <= 2^-24), fp32 accumulate' if args.workload != 'train' else
                                  'fp32 everywhere; forward map CNN and dense layers on the fp16 matrix cores with two-piece round-to-nearest '
                                  'operand splits (3 products per fp32 product, dropped terms <= 2^-24); the CNN BACKWARD (data and weight '
                                  'gradients) with two-piece bf16 operand splits: 2^-16 per product (TF32-class), fp32 accumulate'),
                   'rollout_kernels': rollout_kernels, 'hip_graph': hip_graph,
                   'weights': ('random init (synth.fill_state_dict)' if args.planner != 'hardcode' else
                               'random init, decoder output layer scaled (acceleration x0.3, yaw acceleration x0.05: synth.lane_keeping_weights) '
                               'so that the predicted agents follow their lanes -- what the rule-based planner assumes')},
        # the timed path (graph replay) against one eager iteration from the same latents and Adam state, after the timed region:
        # max |z_replayed - z_eager| (and how far that iteration moved the latents); the run FAILS above 1e-5
        'graph_vs_eager_max_abs': None if gve is None else gve[0],
        'graph_vs_eager_step_size': None if gve is None else gve[1],
        'final_loss': float(loss.detach().cpu()),
        'host_enqueue_ms_per_step': round(dt_host / args.steps * 1e3, 3),     # diagnostic: the host has queued everything by then
        'host_cpu_ms_per_step': round(dt_host_cpu / args.steps * 1e3, 3),    # of which the enqueueing thread was on a core
        'planner': None if planner_ms is None else planner_ms,
        'scenes_dropped': scenes_dropped if args.workload != 'full' else step.stats.get('scenes_dropped'),
        'pipeline': dict(step.stats) if args.workload == 'full' else None,
        'per_rank': [{'rank': r[0], 'agents': r[1], 'ms_per_step': r[2], 'device': r[3], 'local_rank': r[4], 'cores': r[5] if len(r) > 5 else None}
                     for r in per_rank],
        # what the collective backend saw (N > 1: RCCL = torch.distributed 'nccl'): world size it reports and the distinct devices
        'rccl_world_size': backend['rccl_world_size'], 'rccl_backend': backend['rccl_backend'],
        'distinct_devices': backend['distinct_devices'],
        'host': host_pin,
    }
    failed = gve is not None and not (gve[0] <= 1e-5)
    if rank == 0:
        if not args.no_roofline and args.workload not in ('train', 'sample', 'full'):
            try:
                out['roofline'] = time_dominant_kernel(m, env, g, mi, emb, device)
                out['roofline']['bandwidth_kernels'].update(time_bandwidth_kernels(m, env, g, mi, device))
                # the closure as a whole = 305 MFLOP algorithmic per agent*timestep (map CNN
                # forward 300.4 + GNN/GRU/dynamics forward+backward), against the fp32 matrix peak
                # matrix work is issued as fp16 x 3 (conv1: x 2), so the roof it runs against is the f16 dense peak:
                # `useful` counts the algorithmic FLOPs, `issued` the products actually sent to the matrix cores.
                # traffic_over_survey_bytes = HBM bytes the five CNN kernels move per agent*step (committed PMC passes) over
                # SURVEY 8(d)'s algorithmic 268 KB: the layer-by-layer design writes and re-reads its fp32 intermediates.
                per_gpu = out['value'] / world
                alg_tf = per_gpu * 305.0e6 / 1e12
                issued = per_gpu * (sum(f * mlt for f, (mlt, _, _) in zip(CONV_FLOPS, CONV_ISSUE)) + 4.7e6) / 1e12
                tr = [_measured_traffic(nm) for nm in CONV_NAMES]
                napl = out['roofline']['agents_per_launch']
                out['roofline']['whole_path'] = {
                    'algorithmic_tflops': round(alg_tf, 2),
                    'frac_of_f16_matrix_peak_useful': round(alg_tf / PEAK_F16_MFMA_TFLOPS, 4),
                    'frac_of_f16_matrix_peak_issued': round(issued / PEAK_F16_MFMA_TFLOPS, 4),
                    'survey_bytes_per_agent_step': 268 * 1024,
                    'cnn_traffic_bytes_per_agent_step': None if any(t is None for t in tr) else int(sum(tr) / napl),
                    'traffic_over_survey_bytes': None if any(t is None for t in tr) else round(sum(tr) / napl / (268 * 1024), 2),
                    'traffic_source': _measured_traffic(CONV_NAMES[0], with_source=True)[1]}
            except Exception as e:      # keep the headline number, but the run fails
                out['roofline'] = {'error': repr(e)}
                failed = True
        if not args.no_roofline and args.workload == 'train':
            try:
                out['roofline'] = time_train_dominant_kernel(m, env, g, mi, device)
            except Exception as e:
                out['roofline'] = {'error': repr(e)}
                failed = True
        if world == 1 and not args.no_cpu_baseline and args.workload not in ('train', 'sample', 'full'):
            try:
                out['cpu_baseline'] = cpu_baseline_record(args.ft, full=args.cpu_baseline_full)
            except Exception as e:
                out['cpu_baseline'] = {'error': repr(e)}
                failed = True
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


def dry_run(args, rank, world):
    """No GPU: the ranks join a gloo group, build their part of the workload, go through the SAME timing / gathering code as
    the real run (barrier, max over ranks, summed units, per-rank records, what the collective backend reports) around a
    stand-in step, and rank 0 checks that the union of the per-rank scene lists is exactly the job (disjoint + complete for
    `sharded4096`).  `--fail-rank R`: rank R raises before the collectives -- the job must end non-zero, not hang."""
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', '0'))
    host_pin = pin_host_threads(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    if args.fail_rank == rank:
        raise RuntimeError('bench.py --fail-rank %d: this rank fails before the collectives' % rank)
    own, desc, scaling = workload_scenes(args, rank, world)
    mine = {'rank': rank, 'scenes': [k for _, k in own], 'agents': sum(n for n, _ in own)}
    everyone = [mine]
    if use_dist:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        dist.barrier()
    t0 = time.perf_counter()
    x = torch.zeros((256, 256))
    for _ in range(args.steps):
        x = x @ x + 1.0                        # the stand-in "step"
    if use_dist:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    rollouts = 2 if args.workload in ('adv', 'sharded4096', 'train') else 1
    units_local = rollouts * mine['agents'] * args.ft * args.steps
    dt, units, per_rank, backend = gather_ranks(use_dist, world, torch.device('cpu'), dt_local, units_local,
                                                [rank, mine['agents'], round(dt_local / args.steps * 1e3, 3), 'cpu:rank%d/pid%d' % (rank, os.getpid()), local,
                                                 host_pin.get('cores')])
    if rank == 0:
        keys = [k for e in everyone for k in e['scenes']]
        rec = {'dry_run': True, 'n_gpus': world, 'ranks_joined': sorted(e['rank'] for e in everyone), 'scaling': scaling,
               'workload': desc, 'agents_per_rank': [e['agents'] for e in everyone], 'total_agents': sum(e['agents'] for e in everyone),
               'scenes_per_rank': [len(e['scenes']) for e in everyone], 'disjoint': len(keys) == len(set(keys)),
               'units': units, 'ms_per_step': round(dt / args.steps * 1e3, 3),
               'per_rank': [{'rank': r[0], 'agents': r[1], 'ms_per_step': r[2], 'device': r[3], 'local_rank': r[4], 'cores': r[5]} for r in per_rank],
               'rccl_world_size': backend['rccl_world_size'], 'rccl_backend': backend['rccl_backend'],
               'distinct_devices': backend['distinct_devices'], 'host': host_pin}
        if args.workload == 'sharded4096':
            sizes = variable_scene_sizes(args.total_agents or 4096, 'bench/sharded')
            rec['complete'] = sorted(keys) == sorted('bench/sharded/%d' % i for i in range(len(sizes)))
        print(json.dumps(rec), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # a rank that fails must END (the launcher then stops the other ranks) instead of waiting in the process group's
        # destructor for peers that are blocked in a collective it will never join
        import traceback
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
