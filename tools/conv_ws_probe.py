"""conv2 (bench-layer codes 1 / 51) as conv_bf6_kernel against the specialised-wave form (conv_ws_kernel): time per launch and bit
identity of the outputs.  (conv3's specialised-wave forms, codes 52 / 53, were measured in round 4 -- profiles/r04_conv3_ws_probe.txt --
and removed from the library in round 5: git show 5eda569:strive_amd/csrc/map_cnn.hip.)  usage: python tools/conv_ws_probe.py [N ...]"""
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
for n in (int(a) for a in (sys.argv[1:] or ['512'])):
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

    def run(layer, reps=20):
        for _ in range(3):
            lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    def align(v):
        return (v + 255) // 256 * 256
    o1 = align(16 * 125 * 125 * 4 * n)          # act[0] | act[1] | act[2] ... (256-byte aligned blocks)
    nb1 = 32 * 61 * 61 * 4 * n
    o2 = o1 + align(nb1)
    nb2 = 64 * 29 * 29 * 4 * n
    for name, a_id, b_id, off, nb in (('conv2', 1, 51, o1, nb1),):
        ws[off:off + nb].zero_(); run(a_id, 1); torch.cuda.synchronize(); a = ws[off:off + nb].clone()
        ws[off:off + nb].zero_(); run(b_id, 1); torch.cuda.synchronize(); b = ws[off:off + nb].clone()
        print('N=%d %s: conv_bf6_kernel %.1f us, specialised waves %.1f us, outputs bit-identical: %s (non-zero: %s)' % (
            n, name, run(a_id), run(b_id), bool(torch.equal(a, b)), bool(a.any())), flush=True)
