"""Which torch operators a refine closure launches between the HIP calls (torch profiler, one closure after warm-up)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from torch.profiler import profile, ProfilerActivity
scenes, agents = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 16
workload = sys.argv[3] if len(sys.argv) > 3 else 'refine'
dev = torch.device('cuda:0')
args = bench.parse_args(['--scenes', str(scenes), '--agents', str(agents), '--raster', '1024', '--workload', workload])
if not args.nc:
    args.nc = 2
own = bench.workload_scenes(args, 0, 1)[0]
m = bench.build_model(dev, args.nc)
env = bench.build_env(args.raster, dev)
batch, map_idx = bench.build_batch(own, args.nc, args.raster)
step = (bench.refine_closure_factory if workload == 'refine' else bench.adv_closure_factory)(m, env, batch, map_idx, args.ft, dev)[0]
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.cpu_parent is None]
print('top-level ops in order (name, #gpu kernels under it):')
def nk(e):
    n = len(e.kernels)
    for c in e.cpu_children:
        n += nk(c)
    return n
tot = 0
for e in ev:
    k = nk(e)
    tot += k
    print('  %-60s %d' % (e.name[:60], k))
print('total kernels', tot)
