#!/usr/bin/env python3
"""bench.py -- adv-optim agent*timesteps/s of STRIVE's latent-optimisation closure on MI355X.

A "step" is one optimisation closure of the reference's refine loop (reference
src/refine_traffic_optim.py:186-218) over one scene batch:
    zero_grad -> TrafficModel.decode_embedding(z, nfuture=FT)        (FT autoregressive decoder steps)
              -> AvoidCollLoss (vehicle + environment collision, prior NLL, init-z)
              -> backward to z.grad -> Adam step on z
and advances NA*FT agent*timesteps.  Workload (BASELINE.json configs[1]): 32 synthetic scenes x 16 agents
(NA = 512), fp32, synthetic 4096^2 4-layer raster, counter-generated weights (no checkpoint / dataset is
available offline).  With N GPUs every rank owns its own 32 scenes (scenes are independent: weak scaling, no
data-path collective) and the reported value is the whole-job aggregate.

Output: ONE JSON line on rank 0 (see README of the task for the contract) with two extra objects:
  roofline     -- the dominant kernel (a map-CNN convolution, MFMA fp32 bound), timed live with events on the
                  launching stream; achieved = algorithmic FLOPs per launch / average launch duration.
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm, oracle/) timed on this host on a bounded
                  sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 matrix peak
# algorithmic FLOPs per agent of each map-CNN kernel (2 * Cout * OH * OW * Cin * k * k), SURVEY.md §8(a) a8
CONV_FLOPS = [2 * 16 * 125 * 125 * 4 * 49, 2 * 32 * 61 * 61 * 16 * 25, 2 * 64 * 29 * 29 * 32 * 25,
              2 * 64 * 14 * 14 * 64 * 9, 2 * 128 * 6 * 6 * 64 * 9, 2 * 128 * 2 * 2 * 128 * 9]
CONV_NAMES = ['conv1b_kernel<true> (fused crop -> conv1)', 'conv_bf6_kernel<conv2>', 'conv_bf6_kernel<conv3>',
              'conv_bf6_kernel<conv4>', 'conv_bf6s_kernel<conv5>', 'conv_bf6s_kernel<conv6>']
# matrix-core work actually issued per algorithmic FLOP and the dense peak it runs against
# (/opt/skills/guides/MI355X_MICROARCH.md: bf16 dense 2516 TFLOP/s, f32 157.3): conv1 = 3 exact bf16 weight pieces,
# conv2-conv6 = 6 bf16 products per fp32 product
PEAK_BF16_MFMA_TFLOPS = 2516.0
CONV_ISSUE = [(3, PEAK_BF16_MFMA_TFLOPS, 'bf16'), (6, PEAK_BF16_MFMA_TFLOPS, 'bf16'), (6, PEAK_BF16_MFMA_TFLOPS, 'bf16'),
              (6, PEAK_BF16_MFMA_TFLOPS, 'bf16'), (6, PEAK_BF16_MFMA_TFLOPS, 'bf16'), (6, PEAK_BF16_MFMA_TFLOPS, 'bf16')]

REFINE_WEIGHTS = {'coll_veh': 100.0, 'coll_env': 100.0, 'init_z': 0.01, 'motion_prior': 1.0}   # refine_traffic_optim.cfg:26-29


def build_workload(device, scenes, agents, FT, seed_key, raster_px):
    from strive_amd import synth
    from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors
    from strive_amd.models.traffic_model import TrafficModel
    from strive_amd.datasets.utils import MeanStdNormalizer
    m = TrafficModel(4, 12, 256, 2)
    m.load_state_dict(synth.fill_state_dict(m.state_dict()))
    m.set_normalizer(MeanStdNormalizer(*state_norm_tensors()))
    m.set_att_normalizer(MeanStdNormalizer(*att_norm_tensors()))
    m.set_bicycle_params(NUSC_BIKE_PARAMS)
    m = m.eval().to(device)
    raster, dx = synth.make_raster(raster_px, raster_px)
    extent = raster_px * 0.25
    env = synth.SyntheticMapEnv(raster, dx).to(device)
    batch, map_idx = synth.make_batch([agents] * scenes, key=seed_key, FT=12, map_extent=(extent, extent))
    return m, env, batch, map_idx


def gpu_closure_factory(m, env, batch, map_idx, FT, device):
    from strive_amd import synth
    from strive_amd.losses.adv_gen_nusc import AvoidCollLoss
    from strive_amd.utils.scenario_gen import detach_embed_info
    g = batch.to(device)
    mi = map_idx.to(device)
    with torch.no_grad():
        emb = detach_embed_info(m.embed(g, mi, env))
    z0 = synth.make_latents(emb['prior_out'][0].cpu(), emb['prior_out'][1].cpu(), key='bench/z').to(device)
    z = z0.clone().requires_grad_(True)
    opt = torch.optim.Adam([z], lr=0.05)
    loss_fn = AvoidCollLoss(REFINE_WEIGHTS, m.get_att_normalizer().unnormalize(g.lw), mi[g.batch], env, z0.clone(),
                            veh_coll_buffer=0.2)

    def step():
        opt.zero_grad()
        pred = m.decode_embedding(z, emb, g, mi, env, nfuture=FT)['future_pred']
        ld = loss_fn(m.get_normalizer().unnormalize(pred), z, emb['prior_out'])
        ld['loss'].backward()
        opt.step()
        return ld['loss']
    return step, z, emb, g, mi


def _measured_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic.json, written
    from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); None if not collected for this kernel."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_traffic.json')) as f:
            t = json.load(f)
        ent = t.get(kernel_name)
        return None if ent is None else ent.get('bytes_per_launch')
    except Exception:
        return None


def time_dominant_kernel(m, env, g, mi, emb, device, reps=20):
    """Time each CNN kernel in isolation (events on torch's current stream = the launching stream) and return
    the roofline record of the one that takes the most time per rollout step."""
    from strive_amd import ops, _lib as L
    lib = L.get_lib()
    N = min(512, g.past.shape[0])      # = the size of the launches inside the timed closure (one CNN chunk)
    pos = g.past[:N, -1, :4].contiguous()
    mapix = mi[g.batch][:N].to(torch.int32).contiguous()
    mp = ops._map_pack(env, device)
    cnn = ops._cached_pack(m, 'cnn', m.map_conv, lambda: None)
    wsb = lib.query('strive_map_cnn_workspace_bytes', N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    feat = torch.empty((N, 64), device=device)
    nm = m.normalizer
    mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
    st = L.stream_ptr(pos)
    lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
    torch.cuda.synchronize()
    times = []
    for layer in range(6):
        for _ in range(3):
            lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N,
                     L.ptr(feat), L.ptr(ws), wsb, st)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N,
                     L.ptr(feat), L.ptr(ws), wsb, st)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3 / reps)
    dom = max(range(6), key=lambda l: times[l])
    mult, peak, mdt = CONV_ISSUE[dom]
    alg = CONV_FLOPS[dom] * N / times[dom] / 1e12          # algorithmic (fp32-equivalent) TFLOP/s
    ach = alg * mult                                       # matrix-core FLOP/s actually issued
    rec = {'bound': 'mfma', 'kernel': CONV_NAMES[dom], 'achieved': round(ach, 3), 'peak': peak,
           'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': _measured_traffic(CONV_NAMES[dom]),
           'mfma_dtype': mdt, 'products_per_fp32_product': mult, 'algorithmic_tflops': round(alg, 3),
           'frac_of_f32_matrix_peak': round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
           'launch_us': round(times[dom] * 1e6, 2), 'agents_per_launch': N,
           'all_layers_us': [round(t * 1e6, 2) for t in times],
           'all_layers_algorithmic_tflops': [round(CONV_FLOPS[l] * N / times[l] / 1e12, 2) for l in range(6)]}
    return rec


def cpu_baseline(FT, scenes=4, agents=16, raster_px=4096):
    """The oracle's refine closure (decode + AvoidCollLoss + backward) on `scenes` scenes of the same workload."""
    from strive_amd import synth
    from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors
    from strive_amd.models.traffic_model import TrafficModel
    from oracle.model import OracleTrafficModel
    from oracle.geometry import Normalizer
    from oracle.losses import AvoidColl
    m = TrafficModel(4, 12, 256, 2)
    sd = synth.fill_state_dict(m.state_dict())
    orc = OracleTrafficModel(sd, Normalizer(*state_norm_tensors()), Normalizer(*att_norm_tensors()), NUSC_BIKE_PARAMS)
    raster, dx = synth.make_raster(raster_px, raster_px)
    extent = raster_px * 0.25
    env = synth.SyntheticMapEnv(raster, dx)
    batch, map_idx = synth.make_batch([agents] * scenes, key='bench/r0', FT=12, map_extent=(extent, extent))
    with torch.no_grad():
        emb = orc.embed(batch, map_idx, env)
    z0 = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='bench/z')
    z = z0.clone().requires_grad_(True)
    lf = AvoidColl(REFINE_WEIGHTS, orc.get_att_normalizer().unnormalize(batch.lw), map_idx[batch.batch], env, z0.clone(),
                   veh_coll_buffer=0.2)
    t0 = time.time()
    pred = orc.decode_embedding(z, emb, batch, map_idx, env, nfuture=FT)['future_pred']
    ld = lf(orc.get_normalizer().unnormalize(pred), z, emb['prior_out'])
    ld['loss'].backward()
    dt = time.time() - t0
    n = scenes * agents * FT
    return {'value': round(n / dt, 2), 'unit': 'agent*timesteps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d scenes x %d agents, FT=%d, 1 closure (decode + AvoidCollLoss + backward) of the CPU oracle, '
                      '%.1f s' % (scenes, agents, FT, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scenes', type=int, default=32)
    ap.add_argument('--agents', type=int, default=16)
    ap.add_argument('--ft', type=int, default=16, help='rollout steps per closure (refine_traffic_optim.cfg: samp_future_len 16)')
    ap.add_argument('--raster', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)

    import __graft_entry__ as ge
    ge.build(verbose=False)

    m, env, batch, map_idx = build_workload(device, args.scenes, args.agents, args.ft, 'bench/r%d' % rank, args.raster)
    step, z, emb, g, mi = gpu_closure_factory(m, env, batch, map_idx, args.ft, device)
    for _ in range(args.warmup):
        step()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    NA = args.scenes * args.agents
    units = NA * args.ft * args.steps * world
    out = {
        'metric': 'adv-optim agent*timesteps/sec (decoder fwd+bwd)',
        'value': round(units / dt, 1), 'unit': 'agent*timesteps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'refine closure: decode_embedding(nfuture=%d) + AvoidCollLoss + backward + Adam, '
                               '%d scenes x %d agents per GPU (BASELINE.json configs[1])' % (args.ft, args.scenes, args.agents),
                   'agents_per_gpu': NA, 'FT': args.ft, 'raster': '%dx%d x4 uint8 @0.25 m' % (args.raster, args.raster),
                   'parallelism': 'scene-sharded replicas x%d' % world,
                   'arithmetic': 'fp32 everywhere; the map CNN on bf16 matrix cores with exact 3-way operand splits, fp32 accumulate'},
        'final_loss': float(loss.detach().cpu()),
    }
    if rank == 0:
        if not args.no_roofline:
            try:
                out['roofline'] = time_dominant_kernel(m, env, g, mi, emb, device)
                # the closure as a whole = 305 MFLOP algorithmic per agent*timestep (map CNN
                # forward 300.4 + GNN/GRU/dynamics forward+backward), against the fp32 matrix peak
                per_gpu = out['value'] / world
                out['roofline']['whole_path'] = {
                    'algorithmic_tflops': round(per_gpu * 305.0e6 / 1e12, 2),
                    'frac_of_f32_matrix_peak': round(per_gpu * 305.0e6 / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            except Exception as e:      # keep the headline number even if the side measurement fails
                out['roofline'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(args.ft)
            except Exception as e:
                out['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
