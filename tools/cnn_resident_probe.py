#!/usr/bin/env python3
"""Would keeping the CNN's intermediate activations cache-resident pay?  (review item: a sample-pipelined conv3 -> conv4 ->
tail that reads conv2's output while it is still in the L2 / Infinity Cache.)

The measurement hook strive_map_cnn_bench_layer re-launches ONE layer on the activations a full forward left behind.  Repeating
a layer on N samples keeps its input resident whenever it fits a cache level: conv3's input (conv2's output, 476 KB per sample)
is 15 MB at N = 32 (inside the 8 x 4 MB L2s), 61 MB at N = 128 and 122 MB at N = 256 (inside the 256 MB Infinity Cache), 244 MB
at N = 512 (the size of the closure's launches: streamed from HBM).  If the per-sample time of conv3 / conv4 / the tail at the
cache-resident sizes is not better than at N = 512, their time is not HBM time and a pipelined back half has nothing to win.
Prints microseconds per launch and per sample for every kernel of the stack."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
from util import product_model                       # noqa: E402
from strive_amd import synth, ops, _lib as L         # noqa: E402

NAMES = {0: 'conv1 (fused crop)', 1: 'conv2', 2: 'conv3', 3: 'conv4', 7: 'tail (conv5+conv6+fc)'}


def main():
    dev = torch.device('cuda:0')
    m, sd = product_model(device=dev)
    raster, dx = synth.make_raster(4096, 4096)
    env = synth.SyntheticMapEnv(raster, dx).to(dev)
    lib = L.get_lib()
    mp = ops._map_pack(env, dev)
    cnn = ops.cnn_pack(m)
    nm = m.normalizer
    mean4, std4 = L.f4(nm.mean_vals[:4].tolist()), L.f4(nm.std_vals[:4].tolist())
    print('%-24s' % 'kernel' + ''.join('%22s' % ('N = %d' % n) for n in (32, 64, 128, 256, 512)))
    rows = {k: [] for k in NAMES}
    for n in (32, 64, 128, 256, 512):
        fr = np.zeros((n, 4))
        fr[:, 0] = synth.counter_uniform((n,), 'rp/x', 100.0, 900.0)
        fr[:, 1] = synth.counter_uniform((n,), 'rp/y', 100.0, 900.0)
        ang = synth.counter_uniform((n,), 'rp/h', -np.pi, np.pi)
        fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
        pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev).contiguous()
        mapix = torch.zeros((n,), dtype=torch.int32, device=dev)
        wsb = lib.query('strive_map_cnn_workspace_bytes', n)
        ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
        feat = torch.empty((n, 64), device=dev)
        st = L.stream_ptr(pos)
        lib.call('strive_map_cnn_fwd', mp.ref(), cnn.ref(), L.ptr(pos), mean4, std4, L.ptr(mapix), n, L.ptr(feat), L.ptr(ws), wsb, st)
        torch.cuda.synchronize()
        for layer in NAMES:
            def run():
                lib.call('strive_map_cnn_bench_layer', mp.ref(), cnn.ref(), layer, L.ptr(pos), mean4, std4, L.ptr(mapix), n,
                         L.ptr(feat), L.ptr(ws), wsb, st)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            rows[layer].append(e0.elapsed_time(e1) * 1e3 / 20)
    for layer, name in NAMES.items():
        print('%-24s' % name + ''.join('%12.1f us %6.3f' % (t, t / n) for t, n in zip(rows[layer], (32, 64, 128, 256, 512))))
    print('(second number: us per sample)')


if __name__ == '__main__':
    main()
