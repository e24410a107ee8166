"""Map environment with the reference's attribute surface (reference src/datasets/map_env.py:22-203).

The reference rasterises nuScenes maps with the devkit in its constructor; neither the devkit nor the
dataset exists here, so this class is constructed from ready raster tensors (``nusc_raster`` uint8
(M,C,H,W), ``nusc_dx`` float64 (M,2)).  ``get_map_crop`` is the HIP gather."""
import torch

from .. import ops


class NuScenesMapEnv(object):
    def __init__(self, nusc_raster, nusc_dx, bounds=[-17.0, -38.5, 60.0, 38.5], L=256, W=256, device='cpu',
                 map_list=None, layers=('drivable_area', 'carpark_area', 'road_divider', 'lane_divider')):
        self.device = torch.device(device)
        self.nusc_raster = nusc_raster.to(self.device)
        self.nusc_dx = nusc_dx.to(self.device)
        self.bounds = list(bounds)
        self.L, self.W = L, W
        self.layer_names = list(layers)
        self.num_layers = self.nusc_raster.shape[1]
        self.map_list = map_list if map_list is not None else ['map-%d' % i for i in range(self.nusc_raster.shape[0])]

    def get_map_crop(self, scene_graph, map_idx, bounds=None, L=None, W=None):
        """Crop (N,C,L,W) uint8 around ``scene_graph.pos`` (UNNORMALISED), N = NA or NA*NS.  (reference :168-203)"""
        pos = scene_graph.pos
        NA = pos.size(0)
        mapixes = map_idx[scene_graph.batch]
        if pos.dim() == 3:
            NS = pos.size(1)
            pos = pos.reshape(NA * NS, -1)
            mapixes = mapixes.unsqueeze(1).expand(NA, NS).reshape(-1)
        return ops.map_crop(self, pos, mapixes, bounds=bounds, L_=L, W_=W)

    def get_map_crop_pos(self, pos, mapixes, bounds=None, L=None, W=None):
        """(reference :205-228)"""
        return ops.map_crop(self, pos, mapixes, bounds=bounds, L_=L, W_=W)


# ------------------------------------------------------------------------------------------------------------------------
# Rasterisation from geometry (reference src/datasets/map_env.py:79-166).  The reference gets polygons / lines AND their
# rasteriser from the nuscenes devkit (NuScenesMap.get_map_mask); neither the devkit nor the map files exist here, so this is
# the one part of that constructor that can be built: geometry in, nusc_raster / nusc_dx out, on the device
# (include/strive_hip.h strive_map_rasterize -- pinned to its own stated rule by oracle/raster.py, not to the devkit).
# ------------------------------------------------------------------------------------------------------------------------
ROAD_LAYERS = ('drivable_area', 'road_segment', 'lane')          # collapsed into channel 0 (reference :63, :108-113)
LINE_LAYERS = ('road_divider', 'lane_divider')                   # line geometry in the devkit


def map_pixel_size(size_m, pix_per_m):
    """(H, W) pixels and float64 metres per pixel of a map of ``size_m`` = (height, width) metres: round(size pix_per_m) and
    size / pixels, exactly as the reference forms them (:88-92)."""
    import numpy as np
    msize = np.array(size_m, dtype=np.float64)
    px = np.round(msize * pix_per_m).astype(np.int32)
    return (int(px[0]), int(px[1])), msize / px


def _bin_shapes(rings_of_shape, kinds, H, W, dx, half_width):
    """CSR tables of strive_map_rasterize: vertices, ring / shape offsets, per-tile shape lists (32 x 32 pixel tiles)."""
    import numpy as np
    verts, ring_ptr, shape_ptr = [], [0], [0]
    tiles_x, tiles_y = (W + 31) // 32, (H + 31) // 32
    per_tile = [[] for _ in range(tiles_x * tiles_y)]
    for s, rings in enumerate(rings_of_shape):
        pts = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 2) for r in rings], axis=0)
        for r in rings:
            r = np.asarray(r, dtype=np.float64).reshape(-1, 2)
            verts.append(r)
            ring_ptr.append(ring_ptr[-1] + r.shape[0])
        shape_ptr.append(len(ring_ptr) - 1)
        grow = half_width if kinds[s] else 0.0
        # pixel (r, c) stands for the point (c dx_x, r dx_y): the tiles whose points can lie inside the grown bounding box
        c0 = int(np.floor((pts[:, 0].min() - grow) / dx[0])) // 32
        c1 = int(np.ceil((pts[:, 0].max() + grow) / dx[0])) // 32
        r0 = int(np.floor((pts[:, 1].min() - grow) / dx[1])) // 32
        r1 = int(np.ceil((pts[:, 1].max() + grow) / dx[1])) // 32
        for ty in range(max(r0, 0), min(r1, tiles_y - 1) + 1):
            for tx in range(max(c0, 0), min(c1, tiles_x - 1) + 1):
                per_tile[ty * tiles_x + tx].append(s)
    tile_ptr = np.zeros((len(per_tile) + 1,), dtype=np.int32)
    tile_ptr[1:] = np.cumsum([len(t) for t in per_tile])
    tile_shapes = np.array([s for t in per_tile for s in t] or [0], dtype=np.int32)
    v = np.concatenate(verts, axis=0) if verts else np.zeros((1, 2))
    return (v, np.array(ring_ptr, dtype=np.int32), np.array(shape_ptr, dtype=np.int32), np.array(kinds or [0], dtype=np.uint8),
            tile_ptr, tile_shapes)


def rasterize_maps(maps, layers=('drivable_area', 'carpark_area', 'road_divider', 'lane_divider'), pix_per_m=4, flip_singapore=True,
                   device='cpu', line_half_width_px=1.0):
    """``maps``: {name: {'size': (height_m, width_m), 'layers': {layer name: [shape, ...]}}} where a polygon shape is a list of
    rings [exterior, hole, ...] (each (n, 2) x, y metres) and a shape of a LINE_LAYERS layer is one (n, 2) polyline.
    -> (nusc_raster uint8 (M, C, maxH, maxW) zero padded, nusc_dx float64 (M, 2), map_list): channel 0 = the road layers collapsed,
    then one channel per other layer in ``layers`` order, maps whose name starts with 'singapore' flipped about the x axis when
    ``flip_singapore`` -- the reference's layout (:79-166).  Lines are ``2 line_half_width_px`` pixels wide (the devkit draws its
    dividers 2 pixels wide)."""
    import ctypes as C
    import numpy as np
    from .. import _lib as L
    dev = torch.device(device)
    lib = ops._lib_for(torch.zeros((1,), device=dev))
    names = list(maps.keys())
    road = [l for l in layers if l in ROAD_LAYERS]
    other = [l for l in layers if l not in ROAD_LAYERS]
    channels = ([road] if road else []) + [[l] for l in other]
    sizes = [map_pixel_size(maps[n]['size'], pix_per_m) for n in names]
    maxH, maxW = max(s[0][0] for s in sizes), max(s[0][1] for s in sizes)
    raster = torch.zeros((len(names), len(channels), maxH, maxW), dtype=torch.uint8, device=dev)
    for mi, name in enumerate(names):
        (H, W), dx_hw = sizes[mi]
        dx = (float(dx_hw[1]), float(dx_hw[0]))          # x uses the width's metres per pixel, y the height's
        flip = bool(flip_singapore and name.split('-')[0] == 'singapore')
        for ci, group in enumerate(channels):
            for lname in group:
                shapes = maps[name]['layers'].get(lname, [])
                if not shapes:
                    continue
                is_line = lname in LINE_LAYERS
                rings = [[np.asarray(s, dtype=np.float64)] if is_line else [np.asarray(r, dtype=np.float64) for r in s] for s in shapes]
                hw = line_half_width_px * 0.5 * (dx[0] + dx[1])
                tabs = _bin_shapes(rings, [1 if is_line else 0] * len(rings), H, W, dx, hw)
                t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in tabs]
                job = L.StriveRasterJob()
                job.verts, job.ring_ptr, job.shape_ptr, job.shape_kind, job.tile_ptr, job.tile_shapes = [x.data_ptr() for x in t]
                job.H, job.W, job.out_pitch, job.flip_rows = H, W, maxW, int(flip)
                job.dx_x, job.dx_y, job.half_width = dx[0], dx[1], hw
                lib.call('strive_map_rasterize', C.byref(job), L.ptr(raster[mi, ci]), L.stream_ptr(raster))
    nusc_dx = torch.from_numpy(np.stack([s[1] for s in sizes], axis=0)).to(dev)
    return raster, nusc_dx, names


def map_env_from_geometry(maps, layers=('drivable_area', 'carpark_area', 'road_divider', 'lane_divider'), pix_per_m=4, flip_singapore=True,
                          device='cpu', bounds=[-17.0, -38.5, 60.0, 38.5], L=256, W=256):
    """NuScenesMapEnv with the reference constructor's products (nusc_raster, nusc_dx, map_list, layer order) from geometry."""
    raster, dx, names = rasterize_maps(maps, layers, pix_per_m, flip_singapore, device)
    road = [l for l in layers if l in ROAD_LAYERS]
    env = NuScenesMapEnv(raster, dx, bounds=bounds, L=L, W=W, device=device, map_list=names, layers=layers)
    env.layer_map = {l: 0 for l in road}
    env.layer_map.update({l: i + (1 if road else 0) for i, l in enumerate(l_ for l_ in layers if l_ not in ROAD_LAYERS)})
    return env
