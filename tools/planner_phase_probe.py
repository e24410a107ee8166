"""Phase profile of planner_ego_kernel (round 4): STRIVE_PLANNER_PROF makes strive_planner_rollout accumulate clock64 deltas of scene 0's
ego kernel in the spare tail of the planner workspace; this tool runs the closed-loop closure of bench.py on one scene and prints them.

usage (GPU box): python tools/planner_phase_probe.py [agents]"""
import os
import sys

os.environ['STRIVE_PLANNER_PROF'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    agents = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    args = bench.parse_args(['--workload', 'adv', '--planner', 'hardcode', '--total-agents', str(agents), '--scenes', '1'])
    dev = torch.device('cuda:0')
    from strive_amd import ops, synth
    own, desc, _ = bench.workload_scenes(args, 0, 1)
    lane = synth.make_lane_graph(extent=args.raster * 0.25)
    m = bench.build_model(dev, args.nc)
    env = bench.build_env(args.raster, dev, lane)
    batch, map_idx = bench.build_batch(own, args.nc, args.raster, lane_graph=lane)
    step = bench.adv_closure_factory(m, env, batch, map_idx, args.ft, dev)[0]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    bufs = [v for k, v in ops._ws_cache.items() if k[1] == 'planner']
    assert bufs, 'no planner workspace'
    # the profile slots sit behind the carved part: find them by zeroing every spare tail candidate -- simplest: zero the whole buffer tail
    for b in bufs:
        b.zero_()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    names = ['risk + choice + action', 'match + chains + route', 'speed profiles', 'circles', '  match_and_cluster', '  build_chains x2',
             '  assemble_route']
    for b in bufs:
        q = b.view(torch.int64)
        nz = torch.nonzero(q).flatten()
        if nz.numel() == 0:
            continue
        # the 8 slots are the only non-zero 64-bit words that grow monotonically with the rollouts; print the last 8 non-zero words
        vals = q[nz[-7:]].tolist() if nz.numel() >= 7 else q[nz].tolist()
        tot = float(sum(vals[:4]))
        print('clock64 ticks of scene 0 (ego kernel, 32 launches per rollout), %d rollouts; share of the kernel:' % n)
        for nm, v in zip(names, vals):
            print('  %-26s %12d ticks  %5.1f %%' % (nm, v, 100.0 * v / tot))


if __name__ == '__main__':
    main()
