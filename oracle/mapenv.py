"""ORACLE (test infrastructure only) -- rasterised-map lookups.

CPU restatement of the raster gathers on the hot path: the per-agent map crop, the
off-road collision point, and the two small layer checks.  Coordinates are formed in fp32 in the
reference's operation order, divided by the float64 metres-per-pixel in float64, rounded half to
even, and out-of-bounds samples read pixel (0,0) -- all as the reference does.
"""
import numpy as np
import torch


def car_grid(xy, hvec, L, W, bounds=None, ls=None, ws=None):
    """World coordinates ``(B,L,W,2)`` of an ``L x W`` grid attached to each car.

    Restates gen_car_coords (reference src/datasets/nuscenes_utils.py:205-232) without the
    redundant channel dimension: with ``bounds=[lo_l, lo_w, hi_l, hi_w]`` the grid spans fixed metric
    bounds; with ``ls/ws`` it spans each car's own length/width.  dim 1 runs along the heading,
    dim 2 across it.
    """
    B = hvec.shape[0]
    dev = hvec.device
    if bounds is not None:
        lwise = torch.linspace(bounds[0], bounds[2], L, device=dev).view(1, L, 1)
        wwise = torch.linspace(bounds[1], bounds[3], W, device=dev).view(1, 1, W)
    else:
        lwise = torch.linspace(-1.0, 1.0, L, device=dev).view(1, L, 1) * ls.view(B, 1, 1) / 2
        wwise = torch.linspace(-1.0, 1.0, W, device=dev).view(1, 1, W) * ws.view(B, 1, 1) / 2
    hcos = hvec[:, 0].view(B, 1, 1)
    hsin = hvec[:, 1].view(B, 1, 1)
    gx = (lwise * hcos - wwise * hsin) + xy[:, 0].view(B, 1, 1)
    gy = (lwise * hsin + wwise * hcos) + xy[:, 1].view(B, 1, 1)
    return torch.stack([gx, gy], dim=3)


def _to_pixels(xys, dx, mapixes, H, W):
    """fp32 world coords -> int64 pixel indices, OOB -> (0,0).
    (reference src/datasets/nuscenes_utils.py:253-262, 286-294, 360-370)"""
    B = xys.shape[0]
    pix = xys / dx[mapixes].view(B, 1, 1, 2)          # promotes to float64
    pix = torch.round(pix).long()
    outside = (pix[..., 1] < 0) | (pix[..., 1] >= H) | (pix[..., 0] < 0) | (pix[..., 0] >= W)
    pix[outside] = 0
    return pix


# The reference builds the crop's sample coordinates once PER RASTER CHANNEL (gen_car_coords expands the local grid to
# B x C x L x W before rotating it, nuscenes_utils.py:217-230, and get_map_obs divides, rounds and bounds-checks that
# B x C x L x W x 2 tensor, :253-262): C = 4 times the arithmetic of the channel-free form above, same values.  The parity tests
# use the light form; bench.py's cpu_baseline sets this switch so that the CPU number it reports carries the reference's cost.
REFERENCE_CHANNEL_STRUCTURE = False


def _map_crop_per_channel(maps, dx, frame, mapixes, bounds, L, W):
    """get_map_obs with the reference's tensor structure (one coordinate grid per channel); bit-identical to map_crop's output"""
    B, C = frame.shape[0], maps.shape[1]
    dev = frame.device
    lwise = torch.linspace(bounds[0], bounds[2], L, device=dev).view(1, 1, L, 1).expand(B, C, L, W)
    wwise = torch.linspace(bounds[1], bounds[3], W, device=dev).view(1, 1, 1, W).expand(B, C, L, W)
    hcos = frame[:, 2].view(B, 1, 1, 1)
    hsin = frame[:, 3].view(B, 1, 1, 1)
    xys = torch.stack((lwise * hcos - wwise * hsin, lwise * hsin + wwise * hcos), 4) + frame[:, :2].view(B, 1, 1, 1, 2)
    xys[torch.isnan(xys)] = 0.0
    xys = xys / dx[mapixes].view(B, 1, 1, 1, 2)
    xys = torch.round(xys).long()
    m = mapixes.view(B, 1, 1, 1).expand(B, C, L, W)
    c = torch.arange(C, device=dev).view(1, C, 1, 1).expand(B, C, L, W)
    outside = (xys[..., 1] < 0) | (xys[..., 1] >= maps.shape[2]) | (xys[..., 0] < 0) | (xys[..., 0] >= maps.shape[3])
    xys[outside] = 0
    return maps[m, c, xys[..., 1], xys[..., 0]]


def map_crop(maps, dx, frame, mapixes, bounds, L=256, W=256):
    """uint8 crop ``(B,C,L,W)`` around ``frame (B,4)`` (UNNORMALISED x,y,hx,hy).
    Restates get_map_obs (reference src/datasets/nuscenes_utils.py:234-264); NaN frames sample
    world (0,0)."""
    if REFERENCE_CHANNEL_STRUCTURE:
        return _map_crop_per_channel(maps, dx, frame, mapixes, bounds, L, W)
    xys = car_grid(frame[:, :2], frame[:, 2:4], L, W, bounds=bounds)
    xys = torch.where(torch.isnan(xys), torch.zeros_like(xys), xys)
    pix = _to_pixels(xys, dx, mapixes, maps.shape[2], maps.shape[3])
    B, C = frame.shape[0], maps.shape[1]
    m = mapixes.view(B, 1, 1, 1).expand(B, C, L, W)
    c = torch.arange(C, device=frame.device).view(1, C, 1, 1).expand(B, C, L, W)
    py = pix[..., 1].unsqueeze(1).expand(B, C, L, W)
    px = pix[..., 0].unsqueeze(1).expand(B, C, L, W)
    return maps[m, c, py, px]


def coll_grid_size(dx, lw, half=True):
    """(L, W) sample counts from the batch-mean vehicle size and mean resolution
    (reference src/datasets/nuscenes_utils.py:351-354 with the 0.5 factor, :279-282 without)."""
    mdx = torch.mean(dx) * (0.5 if half else 1.0)
    mlw = torch.mean(lw, dim=0)
    L = torch.round(mlw[0] / mdx).int().item()
    W = torch.round(mlw[1] / mdx).int().item()
    return L, W


def coll_point(drivables, dx, cars, lw, mapixes, return_frac=False):
    """Mean world position of the non-drivable samples inside each car box; NaN where the car is
    fully on or fully off the drivable layer.  Restates get_coll_point
    (reference src/datasets/nuscenes_utils.py:334-390)."""
    L, W = coll_grid_size(dx, lw, half=True)
    B = cars.shape[0]
    world = car_grid(cars[:, :2], cars[:, 2:4], L, W, ls=lw[:, 0], ws=lw[:, 1])
    pix = _to_pixels(world, dx, mapixes, drivables.shape[1], drivables.shape[2])
    m = mapixes.view(B, 1, 1).expand(B, L, W)
    car_pix = drivables[m, pix[..., 1], pix[..., 0]].unsqueeze(-1)
    off = car_pix == 0
    n_off = torch.sum(off, dim=(1, 2))                       # (B,1)
    pt = (world * off).sum(dim=(1, 2)) / n_off              # nan when n_off == 0
    full = n_off[:, 0] == L * W
    pt[full] = np.nan
    if return_frac:
        frac = n_off[:, 0] / float(L * W)
        frac[frac == 0] = np.nan
        frac[full] = np.nan
        return pt, frac
    return pt


def on_layer_fraction(drivables, dx, cars, lw, mapixes):
    """Fraction of the car box on pixels marked 1.  Restates check_on_layer
    (reference src/datasets/nuscenes_utils.py:266-298)."""
    L, W = coll_grid_size(dx, lw, half=False)
    B = cars.shape[0]
    world = car_grid(cars[:, :2], cars[:, 2:4], L, W, ls=lw[:, 0], ws=lw[:, 1])
    pix = _to_pixels(world, dx, mapixes, drivables.shape[1], drivables.shape[2])
    m = mapixes.view(B, 1, 1).expand(B, L, W)
    car_pix = drivables[m, pix[..., 1], pix[..., 0]]
    return torch.sum(car_pix.float(), dim=[1, 2]) / (L * W)


def line_hits_layer(drivables, dx, start, end, mapixes):
    """True where the segment start->end crosses a 0 pixel.  Restates check_line_layer
    (reference src/datasets/nuscenes_utils.py:300-332; no bounds handling there either)."""
    B = start.shape[0]
    length = torch.norm(start - end, dim=-1)
    n = torch.max(torch.round(length / torch.mean(dx)).int()).item()
    w = torch.linspace(0.0, 1.0, n).view(1, n, 1).to(start.device)
    pts = start.view(B, 1, 2) * (1.0 - w) + end.view(B, 1, 2) * w
    pix = torch.round(pts / dx[mapixes].view(B, 1, 2)).long()
    m = mapixes.view(B, 1).expand(B, n)
    vals = drivables[m, pix[..., 1], pix[..., 0]]
    return torch.sum(vals == 0, dim=-1) > 0
