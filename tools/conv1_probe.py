"""Timing probes for the fused crop+conv1 kernel (debug variants 11/12/13 of strive_map_cnn_bench_layer)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from strive_amd import ops, _lib as L
dev = torch.device('cuda', 0)
m, env, batch, map_idx = bench.build_workload(dev, 32, 16, 16, 'bench/r0', 4096)
g = batch.to(dev); mi = map_idx.to(dev)
with torch.no_grad():
    m.embed(g, mi, env)
lib = L.get_lib()
N = 256
pos = g.past[:N, -1, :4].contiguous()
mapix = mi[g.batch][:N].to(torch.int32).contiguous()
mp = ops._map_pack(env, dev); cnn = ops.cnn_pack(m)
wsb = lib.query('strive_map_cnn_workspace_bytes', N); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); feat = torch.empty((N, 64), device=dev)
nm = m.normalizer; mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist()); st = L.stream_ptr(pos)
def t(layer, reps=20):
    for _ in range(3): lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / reps
print(json.dumps({'full': t(0), 'no_gather': t(11), 'no_fp64': t(12), 'third_mfma': t(13), 'no_store': t(14)}))
