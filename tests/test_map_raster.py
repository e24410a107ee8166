"""SURVEY.md section 8(f) #4, the buildable piece: NuScenesMapEnv.__init__'s rasterisation (reference
src/datasets/map_env.py:79-166) from polygon / line geometry on the device.

NO REFERENCE FIXTURE IS POSSIBLE: the reference takes geometry and rasteriser from the nuscenes devkit
(NuScenesMap.get_map_mask -> cv2.fillPoly / cv2.polylines), which is absent.  The kernel is pinned to its own stated rule
(include/strive_hip.h strive_map_rasterize) by an exact rational-arithmetic oracle (oracle/raster.py); what is restated from
the reference and checked here is the layout arithmetic around it: pixel counts, nusc_dx, channel order with the road layers
collapsed, the Singapore flip, zero padding."""
import os
import sys

import numpy as np
import pytest
import torch

from strive_amd import synth
from strive_amd.datasets.map_env import rasterize_maps, map_env_from_geometry, map_pixel_size
from oracle import raster as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))
LAYERS = ('drivable_area', 'lane', 'carpark_area', 'road_divider', 'lane_divider')


def lattice_world(key, size=(24.0, 32.0), npoly=7, nline=4):
    """polygons (some with a hole, some concave) and polylines on a 1/8 m lattice: every test of the rule is exact in float64"""
    u = synth.counter_uniform((npoly + nline, 16), key, 0.0, 1.0)
    H, W = size
    polys, lines = [], []
    for i in range(npoly):
        cx, cy = 2 + u[i, 0] * (W - 4), 2 + u[i, 1] * (H - 4)
        nv = 4 + int(u[i, 2] * 4)
        ang = np.sort(u[i, 3:3 + nv]) * 2 * np.pi
        rad = 1.0 + u[i, 8:8 + nv] * 5.0
        ext = np.round(np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1) * 8) / 8
        rings = [ext]
        if i % 3 == 0:
            rings.append(np.round(np.array([[cx - 0.5, cy - 0.5], [cx + 0.5, cy - 0.5], [cx + 0.5, cy + 0.5], [cx - 0.5, cy + 0.5]]) * 8) / 8)
        polys.append(rings)
    for i in range(nline):
        k = npoly + i
        pts = np.stack([u[k, 0:4] * W, u[k, 4:8] * H], axis=1)
        lines.append(np.round(pts * 8) / 8)
    # an axis-aligned box whose edges pass exactly through lookup points and a 32-pixel tile border (boundary pixels are set)
    polys.append([np.array([[4.0, 4.0], [8.0, 4.0], [8.0, 8.0], [4.0, 8.0]])])
    return polys, lines


def maps_for(key):
    p1, l1 = lattice_world(key + '/a')
    p2, l2 = lattice_world(key + '/b', size=(12.5, 9.75), npoly=3, nline=2)
    return {'singapore-synth': {'size': (24.0, 32.0), 'layers': {'drivable_area': p1[:4], 'lane': p1[4:6], 'carpark_area': p1[6:],
                                                                'road_divider': l1[:2], 'lane_divider': l1[2:]}},
            'boston-synth': {'size': (12.5, 9.75), 'layers': {'drivable_area': p2, 'road_divider': l2}}}


def check_against_oracle(raster, dx, names, maps, device_flip=True):
    assert raster.dtype == torch.uint8 and raster.shape[:2] == (2, 4) and names == list(maps.keys())
    sizes = [map_pixel_size(maps[n]['size'], 4) for n in names]
    assert tuple(raster.shape[2:]) == (max(s[0][0] for s in sizes), max(s[0][1] for s in sizes))
    r = raster.cpu().numpy()
    for mi, n in enumerate(names):
        (H, W), d = sizes[mi]
        np.testing.assert_array_equal(dx[mi].cpu().numpy(), np.array(maps[n]['size']) / np.array([H, W]))     # reference :88-92
        lay = maps[n]['layers']
        flip = n.startswith('singapore') and device_flip
        hw = 0.5 * (d[0] + d[1])
        want = [orc.rasterize_layer(H, W, d[1], d[0], polygons=lay.get('drivable_area', []) + lay.get('lane', []), flip_rows=flip),
                orc.rasterize_layer(H, W, d[1], d[0], polygons=lay.get('carpark_area', []), flip_rows=flip),
                orc.rasterize_layer(H, W, d[1], d[0], lines=lay.get('road_divider', []), half_width=hw, flip_rows=flip),
                orc.rasterize_layer(H, W, d[1], d[0], lines=lay.get('lane_divider', []), half_width=hw, flip_rows=flip)]
        for c in range(4):
            assert np.array_equal(r[mi, c, :H, :W], want[c]), 'map %s channel %d: %d pixels differ' % (n, c, int((r[mi, c, :H, :W] != want[c]).sum()))
            assert r[mi, c, H:, :].sum() == 0 and r[mi, c, :, W:].sum() == 0, 'padding must stay zero'
        assert want[0].sum() > 100 and want[2].sum() > 20


@pytest.fixture()
def emu_ops():
    import build as emu_build
    from strive_amd import _lib as L, ops
    emu = L.StriveLib(emu_build.build(), require_all=True)
    orig = (ops._lib_for, L.get_lib)
    ops._lib_for = lambda *tensors: emu
    L.get_lib = lambda: emu
    from util import poison_new_workspaces, nan_empty
    orig_ws = poison_new_workspaces(ops)          # new scratch buffers start as NaN bytes, not as whatever torch.empty holds
    orig_empty, torch.empty = torch.empty, nan_empty()      # ... and so does every output allocated with torch.empty
    yield emu
    torch.empty = orig_empty
    ops._workspace = orig_ws
    ops._lib_for, L.get_lib = orig


def test_rasteriser_emulated_equals_the_exact_rule(emu_ops):
    maps = maps_for('raster/cpu')
    raster, dx, names = rasterize_maps(maps, LAYERS, pix_per_m=4)
    check_against_oracle(raster, dx, names, maps)
    # without the flip the Singapore map is the mirror image about the x axis
    r2, _, _ = rasterize_maps(maps, LAYERS, pix_per_m=4, flip_singapore=False)
    (H, W), _ = map_pixel_size(maps['singapore-synth']['size'], 4)
    assert torch.equal(r2[0, :, :H, :W].flip(1), raster[0, :, :H, :W]) and torch.equal(r2[1], raster[1])


@pytest.mark.gpu
def test_rasteriser_gpu_equals_the_exact_rule_and_feeds_the_crop():
    dev = 'cuda:0'
    maps = maps_for('raster/gpu')
    raster, dx, names = rasterize_maps(maps, LAYERS, pix_per_m=4, device=dev)
    assert raster.is_cuda
    check_against_oracle(raster, dx, names, maps)
    # the environment built from it serves get_map_crop like one built from ready rasters (bit-exact gather, oracle/mapenv.py)
    from oracle import mapenv
    env = map_env_from_geometry(maps, LAYERS, pix_per_m=4, device=dev)
    assert env.map_list == names and env.num_layers == 4 and env.layer_map['lane'] == 0 and env.layer_map['road_divider'] == 2
    frames = torch.tensor([[12.0, 9.0, 0.6, 0.8], [20.0, 15.5, -1.0, 0.0], [5.0, 6.0, 0.0, 1.0]])
    mi = torch.tensor([0, 0, 1])
    crop = env.get_map_crop_pos(frames.to(dev), mi.to(dev)).cpu()
    assert torch.equal(crop, mapenv.map_crop(raster.cpu(), dx.cpu(), frames, mi, env.bounds))
    assert crop.sum() > 0
