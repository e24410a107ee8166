"""Gaussian log-likelihood and KL terms (reference src/losses/common.py:8-41); elementwise torch glue."""
import math

import torch


def kl_normal(qm, qv, pm, pv):
    """KL(q || p) between diagonal normals, summed over the last dim (reference :8-24)."""
    e = 0.5 * (torch.log(pv) - torch.log(qv) + qv / pv + (qm - pm).pow(2) / pv - 1)
    return e.sum(-1)


def log_normal(x, m, v):
    """Sum over the last dim of the log density of N(m, v) at x (reference :26-41)."""
    lp = -torch.log(torch.sqrt(v)) - math.log(math.sqrt(2 * math.pi)) - ((x - m) ** 2 / (2 * v))
    return torch.sum(lp, dim=-1)
