"""Where conv2 / conv3 spend their time: the kernel without its matrix steps, without its output stores, without its input
loads (strive_map_cnn_bench_layer codes 31-34 / 41-44; the outputs of those launches are invalid)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from util import product_model
from strive_amd import synth, ops, _lib as L
dev = torch.device('cuda:0')
m, sd = product_model(device=dev)
raster, dx = synth.make_raster(1024, 1024, M=2)
env = synth.SyntheticMapEnv(raster, dx).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fr = np.zeros((n, 4))
fr[:, 0] = synth.counter_uniform((n,), 'st/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((n,), 'st/y', 20.0, 236.0)
ang = synth.counter_uniform((n,), 'st/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev).contiguous()
mi = torch.tensor([i % 2 for i in range(n)]).to(dev)
ops.encode_map(m, pos, torch.arange(n).to(dev), mi, env)
lib = L.get_lib()
mp = ops._map_pack(env, dev); cnn = ops.cnn_pack(m)
mapix = mi.to(torch.int32).contiguous()
wsb = lib.query('strive_map_cnn_workspace_bytes', n)
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
feat = torch.zeros((n, 64), device=dev)
nm = m.normalizer
mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
st = L.stream_ptr(pos)


def run(layer, reps=20):
    for _ in range(3):
        lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
for name, base, dbg0 in (('conv2', 1, 30), ('conv3', 2, 40)):
    print('%s, %d samples: full %.1f us | no matrix steps %.1f | no output stores %.1f | no input loads %.1f | neither steps nor stores %.1f' %
          (name, n, run(base), run(dbg0 + 1), run(dbg0 + 2), run(dbg0 + 3), run(dbg0 + 4)))
    lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
