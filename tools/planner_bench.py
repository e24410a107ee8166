#!/usr/bin/env python3
"""The device planner alone on bench.py's closed-loop batch (512 agents in 36 scenes on the 1 km^2 synthetic lane graph): N rollouts
on constant-velocity futures, wall time per rollout, and trajectory / route statistics.  Run under rocprofv3 --kernel-trace
--stats for the per-kernel split (profiles/r03_planner_*)."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, REPO)
import bench                                                   # noqa: E402
from strive_amd import synth                                   # noqa: E402
from strive_amd.planners.planner import PlannerConfig          # noqa: E402
from strive_amd.planners.hardcode_goalcond_nusc import HardcodeNuscPlanner, CONFIG_DICT     # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    lg = synth.make_lane_graph(extent=1024.0)
    sizes = bench.variable_scene_sizes(512, 'bench/adv/r0')
    own = [(k, 'bench/adv/r0/%d' % b) for b, k in enumerate(sizes)]
    batch, map_idx = bench.build_batch(own, 2, 4096, lane_graph=lg)
    m = bench.build_model(dev, 2, lane_keeping=True)
    unn = m.get_normalizer().unnormalize
    env = synth.SyntheticMapEnv(torch.zeros((1, 4, 8, 8), dtype=torch.uint8), torch.tensor([[0.25, 0.25]], dtype=torch.float64), lane_graph=lg)
    pl = HardcodeNuscPlanner(env, PlannerConfig(**CONFIG_DICT['default']))
    g = batch.to(dev)
    B = len(sizes)
    pl.reset(unn(g.past_gt[:, -1, :]), m.get_att_normalizer().unnormalize(g.lw), g.batch, B, map_idx)
    ego = torch.zeros((g.past.shape[0],), dtype=torch.bool, device=dev)
    ego[g.ptr[:-1].to(dev)] = True
    obs = unn(g.future_gt[~ego][:, :12, :4]).contiguous()
    t = np.linspace(0.5, 6.0, 12)
    ptr = (g.ptr.cpu() - torch.arange(B + 1)).numpy()
    plan = pl.rollout(obs, t, ptr, t)
    pl.check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan = pl.rollout(obs, t, ptr, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pl.check()
    print('planner alone: %d scenes / %d agents, 31 planner steps: %.3f ms per rollout; plan finite: %s' %
          (B, sum(sizes), 1e3 * dt, bool(torch.isfinite(plan).all())))
    if os.environ.get('STRIVE_PLANNER_PROF') == '1':
        # the option's counters (csrc/planner.hip): slots 0-3 scene 0's ego kernel, 4-6 its route phases, 8-13 the routes kernel, all waves
        q = pl._ws.view(torch.uint8)[pl._ws_prof_offset:pl._ws_prof_offset + 24 * 8].view(torch.int64)
        q.zero_()
        for _ in range(n):
            pl.rollout(obs, t, ptr, t)
        torch.cuda.synchronize()
        v = q.tolist()
        names = ['ego: risk + choice + action', 'ego: match + chains + route', 'ego: speed profiles', 'ego: circles', 'ego:   match_and_cluster',
                 'ego:   build_chains x 2', 'ego:   assemble_route', '-', 'routes: match_and_cluster', 'routes: build_chains x 2',
                 'routes: assemble + emit', 'routes:   emit alone', 'routes: poses (waves)', 'routes: routes',
                 'ego: .. route preload + counts', 'ego: .. minima + tanh', 'ego: .. products', 'ego: .. choice', '-', '-',
                 'routes: .. node positions', 'routes: .. closest point + arc lengths', 'routes: .. resampling + blend', 'routes: .. headings + knot arc lengths']
        for nm, x in zip(names, v):
            print('  %-30s %14d' % (nm, x))
        if v[12]:
            print('  routes kernel per pose: %.0f ticks in the three phases, %.2f routes; per route assemble %.0f, emit %.0f ticks' %
                  ((v[8] + v[9] + v[10]) / v[12], v[13] / v[12], (v[10] - v[11]) / max(v[13], 1), v[11] / max(v[13], 1)))


if __name__ == '__main__':
    main()
