"""ORACLE (test infrastructure only) -- rigid transforms, bicycle dynamics, normalisers.

CPU restatement, in plain torch fp32, of the small geometric operators on STRIVE's
latent-optimisation hot path.  Each function cites the reference lines it restates.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (strive_amd/) never does.
"""
import torch


class Normalizer(object):
    """(x - mean) / std on the leading ``D`` components of the last dim.
    Restates MeanStdNormalizer (reference src/datasets/utils.py:44-113)."""

    def __init__(self, mean, std):
        self.mean_vals = mean.to(torch.float32)
        self.std_vals = std.to(torch.float32)

    def _ms(self, x):
        d = x.shape[-1]
        return self.mean_vals[:d].to(x.device), self.std_vals[:d].to(x.device)

    def normalize(self, x):
        m, s = self._ms(x)
        return (x - m) / s

    def unnormalize(self, x):
        m, s = self._ms(x)
        return x * s + m


def transform2frame(frame, poses, inverse=False):
    """Poses ``(B,N,4)`` of (x,y,hx,hy) into the frame ``(B,4)`` (or back out with ``inverse``).

    Restates reference src/utils/transforms.py:78-139 with the 2x2 products written out:
    forward   t' = R_f (t - t_f),  h' = first column of R_p R_f
    inverse   t' = R_f^T t + t_f,  h' = first column of R_p R_f^T
    with R_f = [[c, s], [-s, c]], R_p = [[pc, -ps], [ps, pc]].  Headings are not renormalised.
    """
    # The 2x2 products go through torch.matmul on (B,N,2,2) operands, as the reference does, so that the
    # CPU oracle reproduces the reference's rounding (batched-matmul inner products) bit for bit; the rollout
    # re-samples the map raster at every step, and a 1-ulp pose difference can flip a crop pixel.
    B, N, _ = poses.shape
    c, s = frame[:, 2], frame[:, 3]
    Rf = torch.stack([c, s, -s, c], dim=1).reshape(B, 1, 2, 2).expand(B, N, 2, 2)
    pc, ps = poses[:, :, 2], poses[:, :, 3]
    Rp = torch.stack([pc, -ps, ps, pc], dim=2).reshape(B, N, 2, 2)
    ft = frame[:, :2].reshape(B, 1, 2)
    if inverse:
        R = torch.matmul(Rp, Rf.transpose(2, 3))
        t = torch.matmul(Rf.transpose(2, 3), poses[:, :, :2].reshape(B, N, 2, 1))[:, :, :, 0] + ft
    else:
        R = torch.matmul(Rp, Rf)
        t = torch.matmul(Rf, (poses[:, :, :2] - ft).reshape(B, N, 2, 1))[:, :, :, 0]
    return torch.cat([t, torch.stack([R[:, :, 0, 0], R[:, :, 1, 0]], dim=2)], dim=-1)


def bicycle_step(state_u, a, ddh, veh_len, dt, max_hdot, max_s):
    """One kinematic-bicycle step on UNNORMALISED states ``(N,6)`` (x,y,hx,hy,s,hdot) with
    acceleration ``a (N,)``, yaw acceleration ``ddh (N,)`` and vehicle length ``veh_len (N,)``.

    Restates kinematics2angle -> car_dynamics -> kinematics2vec
    (reference src/utils/transforms.py:8-29, src/models/common.py:47-68,
    src/models/traffic_model.py:714-733) for a single step.
    """
    x, y, hx, hy, s, hdot = [state_u[:, i] for i in range(6)]
    h = torch.atan2(hy, hx)
    new_hdot = (hdot + ddh * dt).clamp(-max_hdot, max_hdot)
    new_h = h + dt * s.abs() / veh_len * new_hdot
    new_s = (s + a * dt).clamp(0.0, max_s)
    new_y = y + new_s * new_h.sin() * dt
    new_x = x + new_s * new_h.cos() * dt
    return torch.stack([new_x, new_y, new_h.cos(), new_h.sin(), new_s, new_hdot], dim=1)


# ------------------------------------------------------------------------------------------------
# Rotated-rectangle IoU (success / collision-metric tests).  The reference evaluates shapely polygons built from
# get_corners (reference src/datasets/nuscenes_utils.py:392-428) -- shapely 1.7.1 / GEOS are absent from this image and
# from /root/reference, so this restates the published geometry (convex polygon intersection area, shoelace formula)
# in float64 numpy; PARITY UNPINNED for the IoU value itself (no reference output can be generated here); the tests
# pin it against closed-form cases and a grid-sampling estimate instead.
# ------------------------------------------------------------------------------------------------
def rect_corners(box, lw):
    """(4,2) float64 corners, counter-clockwise, as get_corners builds them: R(atan2(hy, hx)) * (+-l/2, +-w/2) + xy."""
    import numpy as np
    l, w = float(lw[0]), float(lw[1])
    base = np.array([[-l / 2., -w / 2.], [l / 2., -w / 2.], [l / 2., w / 2.], [-l / 2., w / 2.]], dtype=np.float64)
    h = np.arctan2(float(box[3]), float(box[2]))
    rot = np.array([[np.cos(h), np.sin(h)], [-np.sin(h), np.cos(h)]])
    return base @ rot + np.asarray(box[:2], dtype=np.float64)


def _poly_area(p):
    import numpy as np
    x, y = p[:, 0], p[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y)))


def rect_iou(box_a, lw_a, box_b, lw_b):
    """intersection area / union area of two vehicle boxes, float64; NaN if a pose contains NaN."""
    import numpy as np
    if np.any(np.isnan(np.asarray(box_a, dtype=np.float64))) or np.any(np.isnan(np.asarray(box_b, dtype=np.float64))):
        return float('nan')
    pa, pb = rect_corners(box_a, lw_a), rect_corners(box_b, lw_b)
    poly = [tuple(v) for v in pa]
    for e in range(4):                                  # keep the part of `poly` left of every edge of b
        e0, e1 = pb[e], pb[(e + 1) % 4]
        d = e1 - e0
        side = [d[0] * (v[1] - e0[1]) - d[1] * (v[0] - e0[0]) for v in poly]
        nxt = []
        for i in range(len(poly)):
            j = (i + 1) % len(poly)
            if side[i] >= 0.0:
                nxt.append(poly[i])
            if (side[i] >= 0.0) != (side[j] >= 0.0):
                t = side[i] / (side[i] - side[j])
                nxt.append((poly[i][0] + t * (poly[j][0] - poly[i][0]), poly[i][1] + t * (poly[j][1] - poly[i][1])))
        poly = nxt
        if not poly:
            break
    inter = _poly_area(np.array(poly)) if len(poly) >= 3 else 0.0
    return inter / (_poly_area(pa) + _poly_area(pb) - inter)
