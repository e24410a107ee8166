"""Phase profile of conv_ws_kernel (bench-layer code 81): clock sums per workgroup of consumer wave 0 and the first producer wave.
(clock64 ticks are NOT core cycles under load -- tools/tick_probe.hip: 1.49 ticks/ns with two matrix-bound waves per SIMD at an
unchanged 2.0 GHz matrix issue rate -- read them as shares.)"""
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
lib = L.get_lib()
n = 512
fr = np.zeros((n, 4))
fr[:, 0] = synth.counter_uniform((n,), 'st/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((n,), 'st/y', 20.0, 236.0)
ang = synth.counter_uniform((n,), 'st/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev).contiguous()
mi = torch.tensor([i % 2 for i in range(n)]).to(dev)
ops.encode_map(m, pos, torch.arange(n).to(dev), mi, env)
mp = ops._map_pack(env, dev); cnn = ops.cnn_pack(m)
mapix = mi.to(torch.int32).contiguous()
wsb = lib.query('strive_map_cnn_workspace_bytes', n)
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
feat = torch.zeros((n, 64), device=dev)
nm = m.normalizer
mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
st = L.stream_ptr(pos)
lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
names = ['consumer: matrix steps', 'consumer: epilogue', 'consumer: barrier wait', 'producer: request', 'producer: stage', 'producer: barrier wait']
for layer in [int(a) for a in sys.argv[1:]] or [81]:
    for rep in range(2):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
        e1.record(); torch.cuda.synchronize()
    t = feat.view(-1)[:16].view(torch.int64).cpu().numpy()
    wgs = int(t[0]); d = t[1:7].astype(np.float64) / max(wgs, 1)
    print('code %d: %d workgroups, kernel %.1f us; mean clock ticks per workgroup:' % (layer, wgs, e0.elapsed_time(e1) * 1e3))
    for k in range(6):
        print('   %-28s %9.0f ticks' % (names[k], d[k]))
    print('   consumer total %.0f, producer total %.0f ticks' % (d[:3].sum(), d[3:].sum()))
