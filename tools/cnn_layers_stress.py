"""Run-to-run reproducibility of the fused crop -> conv1 kernel: repeat the layer-1 launch (measurement hook, layer 0)
and compare its output bit for bit with the crop kernel followed by the from-crop conv1.  This is the script that
isolated the packed-add corruption to conv1 workgroups sharing a CU; on a correct build
every line reports 0 differing runs."""
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
for n in (40, 96, 300):
    fr = np.zeros((n, 4))
    fr[:, 0] = synth.counter_uniform((n,), 'st/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((n,), 'st/y', 20.0, 236.0)
    ang = synth.counter_uniform((n,), 'st/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
    pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev).contiguous()
    mi = torch.tensor([i % 2 for i in range(n)]).to(dev)
    ops.encode_map(m, pos, torch.arange(n).to(dev), mi, env)
    mp = ops._map_pack(env, dev)
    cnn = ops.cnn_pack(m)
    mapix = mi.to(torch.int32).contiguous()
    wsb = lib.query('strive_map_cnn_workspace_bytes', n)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    feat = torch.empty((n, 64), device=dev)
    nm = m.normalizer
    mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
    st = L.stream_ptr(pos)
    nb = n * 16 * 125 * 125 * 4

    def conv1_out():
        torch.cuda.synchronize()
        return ws[:nb].view(torch.float32).view(n, -1).clone()

    # the crop kernel gets the poses exactly as the fused kernel un-normalises them (fp32 pos * 15): one ulp in a pose
    # can move a crop pixel
    crop = ops.map_crop(env, (pos.cpu() * torch.tensor([15., 15., 1., 1.])).to(dev), mapix)
    lib.call('strive_map_cnn_fwd_from_crop', cnn.ref(), L.ptr(crop), n, L.ptr(feat), L.ptr(ws), wsb, st)
    base = conv1_out()
    bad, rows = 0, set()
    for it in range(20):
        lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), 0, L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat),
                 L.ptr(ws), wsb, st)
        ne = (conv1_out() != base).any(dim=1)
        if bool(ne.any()):
            bad += 1
            rows |= set(torch.nonzero(ne).flatten().tolist())
    print('n=%3d: %d/20 fused conv1 launches differ from crop -> conv1; agents %s' % (n, bad, sorted(rows)[:16]))
