"""Map-CNN training backward alone (strive_map_cnn_bwd through ops.encode_map with parameter gradients) on N crops:
run under `rocprofv3 --kernel-trace --stats` to see the per-layer kernels without the rest of a training step.
usage: python tools/cnn_bwd_probe.py [N] [iters]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from util import product_model
from strive_amd import synth, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 704
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda:0')
m, sd = product_model(device=dev)
for p in m.parameters():
    p.requires_grad_(True)
raster, dx = synth.make_raster(1024, 1024, M=2)
env = synth.SyntheticMapEnv(raster, dx).to(dev)
fr = np.zeros((n, 4))
fr[:, 0] = synth.counter_uniform((n,), 'st/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((n,), 'st/y', 20.0, 236.0)
ang = synth.counter_uniform((n,), 'st/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev).contiguous()
mi = torch.tensor([i % 2 for i in range(n)]).to(dev)
d_feat = torch.from_numpy(synth.counter_uniform((n, 64), 'st/df', -1.0, 1.0).astype(np.float32)).to(dev)
for it in range(iters + 1):
    with torch.enable_grad(), ops.weight_grad_mode(True):
        feat = ops.encode_map(m, pos, torch.arange(n).to(dev), mi, env)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        feat.backward(d_feat)
        t1.record()
        torch.cuda.synchronize()
    if it:
        print('N=%d backward %.3f ms' % (n, t0.elapsed_time(t1)))
