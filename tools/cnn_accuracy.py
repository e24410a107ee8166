"""Accuracy of the HIP map CNN against a float64 evaluation of the same network, next to torch CPU fp32's own error."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'tests', 'golden'))
import numpy as np, torch
from util import product_model
from strive_amd import synth, ops
from oracle import mapenv, model as om
import make_golden as mg
dev = 'cuda:0'
m, sd = product_model(device=dev)
raster, dx, frame, mapixes, lw = mg.g2_inputs()
env = synth.SyntheticMapEnv(raster.clone(), dx.clone()).to(dev)
n = 64
fr = np.zeros((n, 4))
fr[:, 0] = synth.counter_uniform((n,), 'acc/x', 20.0, 236.0); fr[:, 1] = synth.counter_uniform((n,), 'acc/y', 20.0, 236.0)
ang = synth.counter_uniform((n,), 'acc/h', -np.pi, np.pi); fr[:, 2], fr[:, 3] = np.cos(ang), np.sin(ang)
fr = synth.f32(fr)
mi = torch.tensor([i % 2 for i in range(n)])
pos_n = fr / torch.tensor([15., 15., 1., 1.])
crop = mapenv.map_crop(raster, dx, pos_n * torch.tensor([15., 15., 1., 1.]), mi, env.bounds)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
ref64 = om.map_cnn(sd64, crop.double())
cpu32 = om.map_cnn(sd, crop.float())
hip = ops.encode_map(m, pos_n.to(dev), torch.arange(n).to(dev), mi.to(dev), env).cpu()
scale = float(ref64.abs().max())
for name, x in (('torch cpu fp32', cpu32), ('hip', hip)):
    e = (x.double() - ref64).abs()
    print('%-15s max abs err %.3e  mean abs err %.3e  (feature scale %.3f)' % (name, float(e.max()), float(e.mean()), scale))
