"""cProfile of the host side of the training step (bench.py --workload train): where the Python time of a step goes."""
import cProfile
import io
import os
import pstats
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench

args = bench.parse_args(['--workload', 'train'])
dev = torch.device('cuda', 0)
own, desc, _ = bench.workload_scenes(args, 0, 1)
m = bench.build_model(dev, args.nc)
env = bench.build_env(1024, dev)
batch, map_idx = bench.build_batch(own, args.nc, 1024)
step, emb, g, mi, _ = bench.train_step_factory(m, env, batch, map_idx, args.ft, dev)
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
ps = pstats.Stats(pr, stream=buf).sort_stats('cumulative')
ps.print_stats(60)
txt = buf.getvalue().replace(R + '/', '')
print(txt[:9000])
