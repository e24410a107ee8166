"""Normalisers and nuScenes constants with the reference's names (reference src/datasets/utils.py:44-227)."""
import torch

from ..constants import NUSC_BIKE_PARAMS, NUSC_NORM_STATS, BIKE_MAXS, BIKE_MAXHDOT  # noqa: F401


class MeanStdNormalizer(object):
    """(data - mean) / std on the leading D components of the last dim (reference :44-113)."""

    def __init__(self, mean_vals, std_vals):
        self.mean_vals = mean_vals.to(torch.float)
        self.std_vals = std_vals.to(torch.float)
        self.D = self.mean_vals.size(0)
        self._dev = {}

    def _ms(self, x):
        d = x.size(-1)
        key = (str(x.device), d)
        ent = self._dev.get(key)
        if ent is None:
            ent = (self.mean_vals[:d].to(x.device), self.std_vals[:d].to(x.device))
            self._dev[key] = ent
        return ent

    def normalize(self, state_data):
        m, s = self._ms(state_data)
        return (state_data - m) / s

    def unnormalize(self, state_data):
        m, s = self._ms(state_data)
        return (state_data * s) + m

    def normalize_single(self, state_data, state_idx):
        return (state_data - self.mean_vals[state_idx].to(state_data.device)) / self.std_vals[state_idx].to(state_data.device)

    def unnormalize_single(self, state_data, state_idx):
        return (state_data * self.std_vals[state_idx].to(state_data.device)) + self.mean_vals[state_idx].to(state_data.device)


def normalize_scene_graph(scene_graph, state_normalizer, att_normalizer, unnorm=False):
    """In-place (un)normalisation of past/future/pos/lw (reference :207-227).  The HIP model never calls
    this (it does not mutate the graph); kept for drivers that do."""
    sf = state_normalizer.unnormalize if unnorm else state_normalizer.normalize
    af = att_normalizer.unnormalize if unnorm else att_normalizer.normalize
    for k in ('past', 'past_gt', 'future', 'future_gt', 'pos'):
        if k in scene_graph and len(scene_graph[k].size()) > 1:
            scene_graph[k] = sf(scene_graph[k])
    if 'lw' in scene_graph and len(scene_graph.lw.size()) > 1:
        scene_graph.lw = af(scene_graph.lw)
    return scene_graph


def get_ego_inds(scene_graph):
    """Boolean numpy mask of the first agent of every scene of a batched graph (reference :229-236)."""
    import numpy as np
    b = scene_graph.batch.cpu().numpy()
    return np.append([True], (b[1:] - b[:-1]) == 1)


def read_adv_scenes(scene_path):
    """Load every ``*.json`` scenario of a directory (sorted by name) written by ``prepare_output_dict`` into the dicts the
    evaluation tools use: ``name, map, dt, veh_att, scene_past, scene_fut`` (= ``fut_adv``) and, when present,
    ``attack_t, sem`` (reference src/datasets/utils.py:10-38)."""
    import glob
    import json
    import os
    scenes = []
    for path in sorted(glob.glob(os.path.join(scene_path, '*.json'))):
        with open(path, 'r') as f:
            jd = json.load(f)
        if jd is None:
            continue
        sc = {'name': os.path.basename(path)[:-5], 'map': jd['map'], 'dt': jd['dt'],
              'veh_att': torch.tensor(jd['lw']), 'scene_past': torch.tensor(jd['past']), 'scene_fut': torch.tensor(jd['fut_adv'])}
        if 'attack_t' in jd:
            sc['attack_t'] = jd['attack_t']
        if 'sem' in jd:
            sc['sem'] = torch.tensor(jd['sem'])
        scenes.append(sc)
    return scenes
