"""A/B of single CNN layers through strive_map_cnn_bench_layer: interleaved rounds of launches of each code, time per launch
(HIP events on the launching stream) and whether the layer's output block (and the statistics block) is bit-identical to the first
code of its group.  STRIVE_CONV_WS_DBG=<mask> (option conv_ws_dbg) switches phases of the specialised-wave kernels off for timing.
usage: [PROBE_GROUPS="[('conv3', 2, [52, 2])]"] python tools/conv_variants_probe.py [N] [rounds]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
torch.cuda.synchronize()


def run(layer, reps=20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def align(v):
    return (v + 255) // 256 * 256


sizes = [16 * 125 * 125, 32 * 61 * 61, 64 * 29 * 29, 64 * 14 * 14, 128 * 36, 128 * 4]
offs, o = [], 0
for sz in sizes:
    offs.append(o); o += align(sz * 4 * n)
stats_off = o
# groups: (name, output block index, codes); the first code is the reference form
GROUPS = eval(os.environ.get('PROBE_GROUPS', 'None')) or [
    ('conv1', 0, [0]),
    ('conv2', 1, [51, 1]),          # specialised waves | conv_bf6_kernel
    ('conv3', 2, [52, 2]),          # specialised waves + streamed weights | conv_bf6_kernel
    ('conv4', 3, [53, 3]),
]
for name, blk, codes in GROUPS:
    off, nb = offs[blk], sizes[blk] * 4 * n
    ref = None
    same = {}
    for c in codes:
        ws[off:off + nb].zero_(); run(c, 1); torch.cuda.synchronize()
        out = ws[off:off + nb].clone()
        st_blk = ws[stats_off:].clone()
        if ref is None:
            ref, ref_st = out, st_blk
        same[c] = (bool(torch.equal(out, ref)), bool(out.any()),
                   float((st_blk.view(torch.float64) - ref_st.view(torch.float64)).abs().max()) if st_blk.numel() % 8 == 0 else -1.0)
    # restore the reference form's outputs for the next group
    run(codes[0], 1)
    t = {c: [] for c in codes}
    for c in codes: run(c, 3)
    for _ in range(rounds):
        for c in codes:
            t[c].append(run(c))
    for c in codes:
        print('%s N=%d code %2d: median %.1f us  min %.1f us  bit-identical to code %d: %s (non-zero %s, max |d stats| %.3g)' % (
            name, n, c, float(np.median(t[c])), min(t[c]), codes[0], same[c][0], same[c][1], same[c][2]), flush=True)
