"""Repeat the fused crop+CNN on the same inputs and report any run that is not bitwise identical to the first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from util import product_model
from strive_amd import synth, ops
dev = 'cuda:0'
m, sd = product_model(device=dev)
raster, dx = synth.make_raster(1024, 1024, M=2)
env = synth.SyntheticMapEnv(raster, dx).to(dev)
for n in (40, 96, 300, 700):
    fr = np.zeros((n, 4))
    fr[:, 0] = synth.counter_uniform((n,), 'st/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((n,), 'st/y', 20.0, 236.0)
    ang = synth.counter_uniform((n,), 'st/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
    pos = (synth.f32(fr) / torch.tensor([15., 15., 1., 1.])).to(dev)
    mi = torch.tensor([i % 2 for i in range(n)]).to(dev); ba = torch.arange(n).to(dev)
    ref = ops.encode_map(m, pos, ba, mi, env).clone()
    bad = 0
    for it in range(40):
        # perturb allocator / timing a little
        junk = torch.randn((1 + it * 1000,), device=dev)
        got = ops.encode_map(m, pos, ba, mi, env)
        if not torch.equal(got, ref):
            bad += 1
            d = (got - ref).abs()
            rows = torch.nonzero(d.amax(dim=1) > 0).flatten().tolist()
            if bad <= 3: print('n=%d it=%d: %d rows differ, max %.3g, rows %s' % (n, it, len(rows), float(d.max()), rows[:12] + rows[-4:]))
    print('n=%d: %d/40 runs differ' % (n, bad))
