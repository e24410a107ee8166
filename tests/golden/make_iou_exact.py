"""Exact rotated-box IoU table: tests/golden/iou_exact.npz.

The reference's success / collision tests (src/losses/adv_gen_nusc.py:517-623, src/losses/traffic_model.py:465-545) call
shapely -- absent from this image and from /root/reference -- for ``intersection.area / union.area`` of two vehicle boxes whose
corners come from get_corners (src/datasets/nuscenes_utils.py:416-428).  This generator pins that value independently of the
repo's own clipping code (oracle/geometry.py::rect_iou and the HIP kernel strive_rect_iou both use Sutherland-Hodgman in
float64):

  * corners exactly as get_corners builds them (float64 numpy: arctan2, cos / sin, dot);
  * from there EXACT rational arithmetic (fractions.Fraction of the float64 corner coordinates) with a different algorithm:
    the intersection polygon is assembled from the vertices of either box that lie in the other one and the proper
    crossings of their edges, ordered around their centroid by exact cross products, area by the shoelace formula;
  * IoU = inter / (area_a + area_b - inter) as a Fraction, rounded once to float64.

10,500 box pairs (float32 inputs, what the product passes to the kernel): random overlaps, near-parallel high-IoU pairs,
identical boxes, nested boxes, axis-aligned pairs that touch along an edge / at a corner / overlap by 2^-10 m, axis-aligned
partial overlaps (checked here against their closed form) and far-apart pairs.  Run:  python tests/golden/make_iou_exact.py
"""
import os
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from strive_amd import synth          # noqa: E402  (counter-based generator only)


def corners(box, lw):
    """get_corners (reference src/datasets/nuscenes_utils.py:416-428) in float64"""
    l, w = float(lw[0]), float(lw[1])
    base = np.array([[-l / 2., -w / 2.], [l / 2., -w / 2.], [l / 2., w / 2.], [-l / 2., w / 2.]])
    h = np.arctan2(float(box[3]), float(box[2]))
    rot = np.array([[np.cos(h), np.sin(h)], [-np.sin(h), np.cos(h)]])
    out = np.dot(base, rot)
    out += np.asarray(box[:2], dtype=np.float64)
    return out


def _frac_poly(c):
    return [(Fraction(float(x)), Fraction(float(y))) for x, y in c]


def _cross(o, a, b):
    return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])


def _area2(p):
    s = Fraction(0)
    for i in range(len(p)):
        j = (i + 1) % len(p)
        s += p[i][0] * p[j][1] - p[j][0] * p[i][1]
    return s


def _ccw(p):
    return p if _area2(p) >= 0 else p[::-1]


def _inside(q, poly):
    return all(_cross(poly[i], poly[(i + 1) % len(poly)], q) >= 0 for i in range(len(poly)))


def _half(v):
    return 0 if (v[1] > 0 or (v[1] == 0 and v[0] > 0)) else 1


def exact_iou(box_a, lw_a, box_b, lw_b):
    A, B = _ccw(_frac_poly(corners(box_a, lw_a))), _ccw(_frac_poly(corners(box_b, lw_b)))
    pts = [v for v in A if _inside(v, B)] + [v for v in B if _inside(v, A)]
    for i in range(4):
        p, p2 = A[i], A[(i + 1) % 4]
        r = (p2[0] - p[0], p2[1] - p[1])
        for j in range(4):
            q, q2 = B[j], B[(j + 1) % 4]
            s = (q2[0] - q[0], q2[1] - q[1])
            den = r[0] * s[1] - r[1] * s[0]
            if den == 0:
                continue                                # parallel: overlaps are covered by the vertex tests
            qp = (q[0] - p[0], q[1] - p[1])
            t = (qp[0] * s[1] - qp[1] * s[0]) / den
            u = (qp[0] * r[1] - qp[1] * r[0]) / den
            if 0 <= t <= 1 and 0 <= u <= 1:
                pts.append((p[0] + t * r[0], p[1] + t * r[1]))
    pts = list(dict.fromkeys(pts))
    area_a, area_b = _area2(A) / 2, _area2(B) / 2
    inter = Fraction(0)
    if len(pts) >= 3:
        cx = sum(p[0] for p in pts) / len(pts)
        cy = sum(p[1] for p in pts) / len(pts)
        rel = [(p[0] - cx, p[1] - cy) for p in pts]
        import functools

        def cmp(a, b):
            ha, hb = _half(a), _half(b)
            if ha != hb:
                return -1 if ha < hb else 1
            c = a[0] * b[1] - a[1] * b[0]
            return -1 if c > 0 else (1 if c < 0 else 0)
        rel.sort(key=functools.cmp_to_key(cmp))
        inter = abs(_area2(rel)) / 2
    union = area_a + area_b - inter
    return float(inter / union), float(inter), float(area_a), float(area_b)


def _pose(x, y, ang):
    return np.stack([x, y, np.cos(ang), np.sin(ang)], -1)


def cases():
    u = lambda n, key, lo, hi: synth.counter_uniform((n,), 'iou_exact/' + key, lo, hi)
    out = []

    def add(name, a, la, b, lb):
        out.append((name, a.astype(np.float32), la.astype(np.float32), b.astype(np.float32), lb.astype(np.float32)))
    n = 6000
    la = np.stack([u(n, 'r/la', 3.5, 6.0), u(n, 'r/wa', 1.6, 2.4)], -1)
    lb = np.stack([u(n, 'r/lb', 3.5, 6.0), u(n, 'r/wb', 1.6, 2.4)], -1)
    ax, ay = u(n, 'r/ax', -100, 100), u(n, 'r/ay', -100, 100)
    add('random', _pose(ax, ay, u(n, 'r/ha', -np.pi, np.pi)), la, _pose(ax + u(n, 'r/dx', -6, 6), ay + u(n, 'r/dy', -6, 6),
                                                                       u(n, 'r/hb', -np.pi, np.pi)), lb)
    n = 1500
    la = np.stack([u(n, 'p/la', 3.5, 6.0), u(n, 'p/wa', 1.6, 2.4)], -1)
    ha = u(n, 'p/ha', -np.pi, np.pi)
    ax, ay = u(n, 'p/ax', 0, 2000), u(n, 'p/ay', 0, 2000)
    add('near_parallel', _pose(ax, ay, ha), la, _pose(ax + u(n, 'p/dx', -0.5, 0.5), ay + u(n, 'p/dy', -0.5, 0.5),
                                                      ha + u(n, 'p/dh', -0.05, 0.05)), la * u(n, 'p/sc', 0.9, 1.1)[:, None])
    n = 500
    la = np.stack([u(n, 'i/la', 3.5, 6.0), u(n, 'i/wa', 1.6, 2.4)], -1)
    pa = _pose(u(n, 'i/ax', -50, 50), u(n, 'i/ay', -50, 50), u(n, 'i/ha', -np.pi, np.pi))
    add('identical', pa, la, pa.copy(), la.copy())
    n = 500
    la = np.stack([u(n, 'n/la', 5.0, 8.0), u(n, 'n/wa', 2.5, 3.5)], -1)
    ha = u(n, 'n/ha', -np.pi, np.pi)
    ax, ay = u(n, 'n/ax', -50, 50), u(n, 'n/ay', -50, 50)
    add('nested', _pose(ax, ay, ha), la, _pose(ax + u(n, 'n/dx', -0.3, 0.3), ay + u(n, 'n/dy', -0.3, 0.3), ha + u(n, 'n/dh', -0.2, 0.2)),
        np.stack([u(n, 'n/lb', 0.5, 1.5), u(n, 'n/wb', 0.3, 0.8)], -1))
    # axis-aligned, sizes on a 1/4 m lattice so that every coordinate is exact in float32 and float64
    n = 500
    q = lambda key, lo, hi: np.round(u(n, key, lo, hi) * 4) / 4
    la = np.stack([q('t/la', 3.5, 6.0), q('t/wa', 1.5, 2.5)], -1)
    lb = np.stack([q('t/lb', 3.5, 6.0), q('t/wb', 1.5, 2.5)], -1)
    ax, ay = q('t/ax', -40, 40), q('t/ay', -40, 40)
    zero = np.zeros(n)
    touch = (la[:, 0] + lb[:, 0]) / 2
    kind = np.arange(n) % 4
    dx = np.where(kind == 0, touch, np.where(kind == 1, touch - 2.0 ** -10, np.where(kind == 2, touch, touch + 3.0)))
    dy = np.where(kind == 2, (la[:, 1] + lb[:, 1]) / 2, q('t/dy', -0.5, 0.5))            # kind 2: corner contact
    add('touching', _pose(ax, ay, zero), la, _pose(ax + dx, ay + dy, zero), lb)
    n = 500
    la = np.stack([q('a/la', 3.5, 6.0), q('a/wa', 1.5, 2.5)], -1)
    lb = np.stack([q('a/lb', 3.5, 6.0), q('a/wb', 1.5, 2.5)], -1)
    ax, ay = q('a/ax', -40, 40), q('a/ay', -40, 40)
    add('axis_aligned', _pose(ax, ay, np.zeros(n)), la, _pose(ax + q('a/dx', -5, 5), ay + q('a/dy', -2, 2), np.zeros(n)), lb)
    n = 500
    la = np.stack([u(n, 'f/la', 3.5, 6.0), u(n, 'f/wa', 1.6, 2.4)], -1)
    ax, ay = u(n, 'f/ax', -100, 100), u(n, 'f/ay', -100, 100)
    add('far', _pose(ax, ay, u(n, 'f/ha', -np.pi, np.pi)), la, _pose(ax + 20 + u(n, 'f/dx', 0, 50), ay + u(n, 'f/dy', -50, 50),
                                                                     u(n, 'f/hb', -np.pi, np.pi)), la)
    return out


def closed_form_axis_aligned(a, la, b, lb):
    ox = np.maximum(0.0, np.minimum(a[:, 0] + la[:, 0] / 2, b[:, 0] + lb[:, 0] / 2) - np.maximum(a[:, 0] - la[:, 0] / 2, b[:, 0] - lb[:, 0] / 2))
    oy = np.maximum(0.0, np.minimum(a[:, 1] + la[:, 1] / 2, b[:, 1] + lb[:, 1] / 2) - np.maximum(a[:, 1] - la[:, 1] / 2, b[:, 1] - lb[:, 1] / 2))
    inter = ox * oy
    return inter / (la[:, 0] * la[:, 1] + lb[:, 0] * lb[:, 1] - inter)


def main():
    out, names = {}, []
    for name, a, la, b, lb in cases():
        iou = np.zeros((a.shape[0],))
        for i in range(a.shape[0]):
            iou[i] = exact_iou(a[i], la[i], b[i], lb[i])[0]
        if name in ('touching', 'axis_aligned'):
            cf = closed_form_axis_aligned(a.astype(np.float64), la.astype(np.float64), b.astype(np.float64), lb.astype(np.float64))
            assert np.abs(cf - iou).max() < 1e-15, (name, np.abs(cf - iou).max())
        if name == 'identical':
            assert np.all(iou == 1.0)
        if name == 'far':
            assert np.all(iou == 0.0)
        print('%-14s %5d pairs  IoU in [%.4f, %.4f], %d zero, %d above the 0.02 collision threshold' %
              (name, len(iou), iou.min(), iou.max(), int((iou == 0).sum()), int((iou > 0.02).sum())))
        names.append(name)
        out[name + '/a'], out[name + '/la'], out[name + '/b'], out[name + '/lb'], out[name + '/iou'] = a, la, b, lb, iou
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'iou_exact.npz'), **out)
    print('wrote iou_exact.npz: %d pairs' % sum(out[n + '/iou'].shape[0] for n in names))


if __name__ == '__main__':
    main()
