#!/usr/bin/env python3
"""Bound the parity-unpinned rotated-rectangle IoU numerically (shapely / GEOS, which the reference calls at
src/losses/adv_gen_nusc.py:517-623, is not available here): 50 random box pairs, the oracle's float64 convex clipping
(oracle/geometry.py::rect_iou) next to a Monte-Carlo estimate from 10^7 uniform points per pair with its standard error.
Writes tests/golden/iou_mc_table.npz (data only).  Usage: python tests/golden/make_iou_table.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from strive_amd import synth                      # noqa: E402
from oracle.geometry import rect_iou               # noqa: E402

NPAIR, NPTS = 50, 10_000_000


def inside(pts, box, lw):
    h = np.arctan2(box[3], box[2])
    d = pts - box[:2]
    u = d[:, 0] * np.cos(h) + d[:, 1] * np.sin(h)
    v = -d[:, 0] * np.sin(h) + d[:, 1] * np.cos(h)
    return (np.abs(u) <= lw[0] / 2) & (np.abs(v) <= lw[1] / 2)


def main():
    r = synth.counter_uniform((NPAIR, 10), 'iou_mc/pairs')
    boxes_a, boxes_b, lws_a, lws_b, exact, mc, se = [], [], [], [], [], [], []
    rng = np.random.default_rng(20260927)
    for i in range(NPAIR):
        a_ang, b_ang = (r[i, 0] - 0.5) * 2 * np.pi, (r[i, 1] - 0.5) * 2 * np.pi
        a = np.array([0.0, 0.0, np.cos(a_ang), np.sin(a_ang)])
        off = 4.5 * r[i, 2] ** 0.5
        phi = 2 * np.pi * r[i, 3]
        b = np.array([off * np.cos(phi), off * np.sin(phi), np.cos(b_ang), np.sin(b_ang)])
        la = np.array([3.5 + 3.0 * r[i, 4], 1.6 + 0.9 * r[i, 5]])
        lb = np.array([3.5 + 3.0 * r[i, 6], 1.6 + 0.9 * r[i, 7]])
        ex = rect_iou(a, la, b, lb)
        # sample the bounding square of both boxes
        R = 0.5 * np.hypot(max(la[0], lb[0]), max(la[1], lb[1])) + off
        pts = rng.uniform(-R, R, size=(NPTS, 2))
        ia, ib = inside(pts, a, la), inside(pts, b, lb)
        n_i, n_u = int((ia & ib).sum()), int((ia | ib).sum())
        est = n_i / max(n_u, 1)
        # binomial error of the ratio estimator n_i / n_u (n_i ~ Binomial(n_u, iou))
        err = np.sqrt(max(est * (1 - est), 1e-12) / max(n_u, 1))
        boxes_a.append(a); boxes_b.append(b); lws_a.append(la); lws_b.append(lb)
        exact.append(ex); mc.append(est); se.append(err)
        print('%2d  clip %.6f  mc %.6f +- %.6f  (%+.1f sigma)' % (i, ex, est, err, (ex - est) / err))
    np.savez_compressed(os.path.join(HERE, 'iou_mc_table.npz'), box_a=np.array(boxes_a), box_b=np.array(boxes_b),
                        lw_a=np.array(lws_a), lw_b=np.array(lws_b), iou_clip=np.array(exact), iou_mc=np.array(mc),
                        iou_mc_se=np.array(se), npts=np.array(NPTS))


if __name__ == '__main__':
    main()
