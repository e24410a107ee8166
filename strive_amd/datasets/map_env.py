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
