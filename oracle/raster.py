"""TEST INFRASTRUCTURE ONLY: exact restatement of the rasterisation rule of strive_map_rasterize (include/strive_hip.h) in rational
arithmetic -- the ground truth the device kernel is pinned to.

PARITY UNPINNED AGAINST THE REFERENCE: src/datasets/map_env.py:79-166 rasterises with the nuscenes devkit
(NuScenesMap.get_map_mask -> cv2.fillPoly / cv2.polylines), which is absent here; what IS restated from the reference is the
layout arithmetic around it (pixel counts round(size pix_per_m), nusc_dx = size / pixels, road layers collapsed into channel 0,
Singapore maps flipped about the x axis, zero padding to the largest map: :88-92, :108-113, :125-127, :158-162)."""
from fractions import Fraction

import numpy as np


def _f(v):
    return Fraction(float(v))                    # exact value of the float64


def point_in_polygon(px, py, rings):
    """inside or on the boundary, even-odd over the rings"""
    odd = False
    for ring in rings:
        n = len(ring)
        for i in range(n):
            x0, y0 = ring[i]
            x1, y1 = ring[(i + 1) % n]
            ex, ey = x1 - x0, y1 - y0
            cr = ex * (py - y0) - ey * (px - x0)
            if cr == 0 and min(x0, x1) <= px <= max(x0, x1) and min(y0, y1) <= py <= max(y0, y1):
                return True
            if (y0 <= py < y1 and cr > 0) or (y1 <= py < y0 and cr < 0):
                odd = not odd
    return odd


def point_near_polyline(px, py, line, hw):
    hw2 = hw * hw
    for i in range(len(line) - 1):
        x0, y0 = line[i]
        x1, y1 = line[i + 1]
        ex, ey = x1 - x0, y1 - y0
        qx, qy = px - x0, py - y0
        len2 = ex * ex + ey * ey
        t = (qx * ex + qy * ey) / len2 if len2 > 0 else Fraction(0)
        t = min(max(t, Fraction(0)), Fraction(1))
        dx, dy = qx - t * ex, qy - t * ey
        if dx * dx + dy * dy <= hw2:
            return True
    return False


def rasterize_layer(H, W, dx_x, dx_y, polygons=(), lines=(), half_width=0.0, flip_rows=False):
    """(H, W) uint8: polygons = [[ring, ...], ...] with rings as (n, 2) arrays, lines = [(n, 2) array, ...]"""
    polys = [[[(_f(x), _f(y)) for x, y in np.asarray(r).reshape(-1, 2)] for r in p] for p in polygons]
    lns = [[(_f(x), _f(y)) for x, y in np.asarray(l).reshape(-1, 2)] for l in lines]
    fx, fy, hw = _f(dx_x), _f(dx_y), _f(half_width)
    out = np.zeros((H, W), dtype=np.uint8)
    for r in range(H):
        py = r * fy
        for c in range(W):
            px = c * fx
            hit = any(point_in_polygon(px, py, p) for p in polys) or any(point_near_polyline(px, py, l, hw) for l in lns)
            if hit:
                out[(H - 1 - r) if flip_rows else r, c] = 1
    return out
