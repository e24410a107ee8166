"""Raster lookups with the reference's names (reference src/datasets/nuscenes_utils.py:205-390).
``get_map_obs`` and ``get_coll_point`` run the HIP gather kernels; the two small checks used only
outside the optimisation closure are torch glue."""
import numpy as np
import torch

from .. import ops


class _RawEnv(object):
    def __init__(self, maps, dx, bounds, L, W):
        self.nusc_raster, self.nusc_dx, self.bounds, self.L, self.W = maps, dx, list(bounds), L, W


def gen_car_coords(xys, hs, C, L, W, bounds=None, ls=None, ws=None):
    """World coordinates (B,C,L,W,2) of a grid attached to each car (reference :205-232); torch glue."""
    B = hs.size(0)
    dev = hs.device
    if bounds is not None:
        lwise = torch.linspace(bounds[0], bounds[2], L, device=dev).view(1, 1, L, 1)
        wwise = torch.linspace(bounds[1], bounds[3], W, device=dev).view(1, 1, 1, W)
    elif ls is not None and ws is not None:
        lwise = torch.linspace(-1.0, 1.0, L, device=dev).view(1, 1, L, 1) * ls.view(B, 1, 1, 1) / 2
        wwise = torch.linspace(-1.0, 1.0, W, device=dev).view(1, 1, 1, W) * ws.view(B, 1, 1, 1) / 2
    else:
        raise ValueError('pass either bounds or ls and ws')
    hc, hsn = hs[:, 0].view(B, 1, 1, 1), hs[:, 1].view(B, 1, 1, 1)
    out = torch.stack((lwise * hc - wwise * hsn, lwise * hsn + wwise * hc), 4) + xys.view(B, 1, 1, 1, 2)
    return out.expand(B, C, L, W, 2)


def get_map_obs(maps, dx, frame, mapixes, bounds, L=256, W=256):
    """uint8 crop (B,C,L,W) in the frame of each car, bit-exact vs the reference (:234-264)."""
    return ops.map_crop(_RawEnv(maps, dx, bounds, L, W), frame, mapixes)


def _grid_size(dx, lw, half):
    mdx = torch.mean(dx) * (0.5 if half else 1.0)
    mlw = torch.mean(lw, dim=0)
    return torch.round(mlw[0] / mdx).int().item(), torch.round(mlw[1] / mdx).int().item()


def get_coll_point(drivables, dx, cars, lw, mapixes, return_iou=False):
    """Mean position of the non-drivable samples in each car box, NaN if none / all (reference :334-390).
    ``drivables`` is the (M,H,W) layer-0 view of the raster."""
    gl, gw = _grid_size(dx, lw, True)
    M, H, W = drivables.shape
    if drivables.is_contiguous():
        rast = drivables.view(M, 1, H, W)
    elif drivables.stride(2) == 1 and drivables.stride(1) == W and drivables.stride(0) % (H * W) == 0:
        # ``nusc_raster[:, 0]`` of a (M,C,H,W) raster: re-view the parent storage, no copy
        Cc = drivables.stride(0) // (H * W)
        rast = torch.as_strided(drivables, (M, Cc, H, W), (Cc * H * W, H * W, W, 1))
    else:
        rast = drivables.contiguous().view(M, 1, H, W)
    env = _RawEnv(rast, dx, [0, 0, 0, 0], 1, 1)
    pt, cnt = ops.coll_point(env, cars, lw, mapixes, gl, gw)
    if return_iou:
        frac = cnt.to(torch.float32) / float(gl * gw)
        frac = torch.where((cnt == 0) | (cnt == gl * gw), torch.full_like(frac, float('nan')), frac)
        return pt, frac
    return pt


def _pixels(xys, dx, mapixes, H, W):
    B = xys.shape[0]
    pix = torch.round(xys / dx[mapixes].view(B, *([1] * (xys.dim() - 2)), 2)).long()
    outside = (pix[..., 1] < 0) | (pix[..., 1] >= H) | (pix[..., 0] < 0) | (pix[..., 0] >= W)
    pix[outside] = 0
    return pix


def check_on_layer(drivables, dx, cars, lw, mapixes):
    """Fraction of the car box on pixels marked 1 (reference :266-298); torch glue."""
    L, W = _grid_size(dx, lw, False)
    B = cars.size(0)
    xys = gen_car_coords(cars[:, :2], cars[:, 2:], 1, L, W, ls=lw[:, 0], ws=lw[:, 1])[:, 0]
    pix = _pixels(xys, dx, mapixes, drivables.shape[1], drivables.shape[2])
    m = mapixes.view(B, 1, 1).expand(B, L, W)
    return torch.sum(drivables[m, pix[..., 1], pix[..., 0]].float(), dim=[1, 2]) / (L * W)


def check_line_layer(drivables, dx, start, end, mapixes):
    """True where the segment crosses a 0 pixel (reference :300-332); torch glue."""
    B = start.size(0)
    n = torch.max(torch.round(torch.norm(start - end, dim=-1) / torch.mean(dx)).int()).item()
    w = torch.linspace(0.0, 1.0, n).view(1, n, 1).to(start.device)
    pts = start.view(B, 1, 2) * (1.0 - w) + end.view(B, 1, 2) * w
    pix = torch.round(pts / dx[mapixes].view(B, 1, 2)).long()
    m = mapixes.view(B, 1).expand(B, n)
    return torch.sum(drivables[m, pix[..., 1], pix[..., 0]] == 0, dim=-1) > 0
