"""Timing probes for the fused crop+conv1 kernel (debug variants 11/12/13 of strive_map_cnn_bench_layer)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from util import product_model
from strive_amd import ops, _lib as L, synth
dev = torch.device('cuda', 0)
m, sd = product_model(device=dev)
raster, dx = synth.make_raster(1024, 1024, M=2)
env = synth.SyntheticMapEnv(raster, dx).to(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fr = np.zeros((N, 4))
fr[:, 0] = synth.counter_uniform((N,), 'st/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((N,), 'st/y', 20.0, 236.0)
ang = synth.counter_uniform((N,), 'st/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev).contiguous()
mi = torch.tensor([i % 2 for i in range(N)]).to(dev)
ops.encode_map(m, pos, torch.arange(N).to(dev), mi, env)
lib = L.get_lib()
mapix = mi.to(torch.int32).contiguous()
mp = ops._map_pack(env, dev); cnn = ops.cnn_pack(m)
wsb = lib.query('strive_map_cnn_workspace_bytes', N); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); feat = torch.empty((N, 64), device=dev)
nm = m.normalizer; mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist()); st = L.stream_ptr(pos)
def t(layer, reps=20):
    for _ in range(3): lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
print(json.dumps({'full': t(0), 'no_gather': t(11), 'no_fp64': t(12), 'third_mfma': t(13), 'no_store': t(14)}))
