"""Scenario-generation helpers on the hot path (reference src/utils/scenario_gen.py:19-28)."""
import torch


def detach_embed_info(embed_info_attached):
    out = {}
    for k, v in embed_info_attached.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach()
        elif isinstance(v, tuple):
            out[k] = (v[0].detach(), v[1].detach())
    return out
