"""Which torch calls inside one optimisation closure synchronise the host with the GPU?  (torch.cuda.set_sync_debug_mode)
Run on the GPU box: python tools/sync_audit.py [refine|adv|train]"""
import os
import sys
import warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'refine'
args = bench.parse_args(['--workload', wl])
dev = torch.device('cuda', 0)
own, desc, _ = bench.workload_scenes(args, 0, 1)
m = bench.build_model(dev, args.nc)
env = bench.build_env(1024, dev)
batch, map_idx = bench.build_batch(own, args.nc, 1024)
fac = {'refine': bench.refine_closure_factory, 'train': bench.train_step_factory}.get(wl, bench.adv_closure_factory)
step, emb, g, mi, _ = fac(m, env, batch, map_idx, args.ft, dev)
step()
step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step()
torch.cuda.set_sync_debug_mode('default')
print('%s closure: %d synchronising calls' % (wl, len(w)))
seen = {}
for x in w:
    key = '%s:%d %s' % (os.path.relpath(x.filename, R) if x.filename.startswith(R) else x.filename, x.lineno, str(x.message)[:100])
    seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print('  %3d x %s' % (v, k))
