import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
import bench
from strive_amd import ops, _lib as L
dev = torch.device('cuda:0')
args = bench.parse_args([])
own, _, _ = bench.workload_scenes(args, 0, 1)
m = bench.build_model(dev, 2); env = bench.build_env(4096, dev)
batch, map_idx = bench.build_batch(own, 2, 4096)
g = batch.to(dev); mi = map_idx.to(dev)
lib = L.get_lib(); N = 512
pos = g.past[:N, -1, :4].contiguous(); mapix = mi[g.batch][:N].to(torch.int32).contiguous()
mp = ops._map_pack(env, dev); cnn = ops.cnn_pack(m)
wsb = lib.query('strive_map_cnn_workspace_bytes', N); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); feat = torch.empty((N, 64), device=dev)
nm = m.normalizer; mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist()); st = L.stream_ptr(pos)
lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st)
for layer, name in ((0, 'conv1 full'), (14, 'conv1 no store'), (11, 'conv1 no gather'), (12, 'conv1 no fp64 pixel math'), (13, 'conv1 1 of 2 weight pieces')):
    t = bench._event_time(lambda: lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), N, L.ptr(feat), L.ptr(ws), wsb, st), 20)
    print('%-28s %.1f us' % (name, t * 1e6))
