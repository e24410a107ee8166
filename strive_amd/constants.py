"""nuScenes normalisation statistics and bicycle-model parameters used on the hot path.

Values are the ones the reference hard-codes (reference src/datasets/utils.py:118-193); they are
data, not code, and are part of the drop-in contract.
"""
import math

import torch

BIKE_MAXS = 50.0
BIKE_MAXHDOT = 2.0 * math.pi

NUSC_BIKE_PARAMS = {
    'maxs': BIKE_MAXS,
    'maxhdot': BIKE_MAXHDOT,
    'dt': 0.5,
    'a_stats': (0.409074, 1.045530),
    'ddh_stats': (0.000046, 0.075032),
}

_COMMON = {
    'l': (4.844294, 1.084860),
    'w': (2.021752, 0.299647),
    's': (1.802009, 3.507907),
    'h': (0.0, 1.0),
    'hdot': (-0.000037, 0.055684),
    'lscale': (0.0, 15.0),
    'a': (0.409074, 1.045530),
    'ddh': (0.000046, 0.075032),
}

NUSC_NORM_STATS = {
    ('car', 'truck'): dict(_COMMON),
    ('bus', 'car', 'construction', 'emergency', 'truck'): dict(_COMMON),
    ('bus', 'car', 'construction', 'cyclist', 'emergency', 'motorcycle', 'pedestrian', 'truck'): dict(_COMMON),
    ('car', 'cyclist', 'motorcycle', 'pedestrian', 'truck'): dict(_COMMON),
    ('bus', 'car', 'motorcycle', 'trailer', 'truck'): {
        'l': (5.135896, 2.072248),
        'w': (2.042160, 0.409259),
        's': (1.789616, 3.480962),
        'h': (0.0, 1.0),
        'hdot': (-0.000115, 0.058249),
        'lscale': (0.0, 15.0),
    },
    ('bus', 'car', 'construction', 'cyclist', 'emergency', 'motorcycle', 'pedestrian', 'trailer', 'truck'): {
        k: (0.0, 1.0) for k in ('l', 'w', 's', 'h', 'hdot', 'lscale', 'a', 'ddh')
    },
}

NUSC_NORM_STATS_CAR_TRUCK = NUSC_NORM_STATS[('car', 'truck')]


def state_norm_tensors(ninfo=None):
    """(mean, std) fp32 tensors for the 6-d state (x,y,hx,hy,s,hdot), ordered as the reference's
    dataset builds them (reference src/datasets/nuscenes_dataset.py:212-217)."""
    ni = NUSC_NORM_STATS_CAR_TRUCK if ninfo is None else ninfo
    mean = [ni['lscale'][0], ni['lscale'][0], ni['h'][0], ni['h'][0], ni['s'][0], ni['hdot'][0]]
    std = [ni['lscale'][1], ni['lscale'][1], ni['h'][1], ni['h'][1], ni['s'][1], ni['hdot'][1]]
    return torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)


def att_norm_tensors(ninfo=None):
    """(mean, std) for (l, w) (reference src/datasets/nuscenes_dataset.py:218-222)."""
    ni = NUSC_NORM_STATS_CAR_TRUCK if ninfo is None else ninfo
    return (torch.tensor([ni['l'][0], ni['w'][0]], dtype=torch.float32),
            torch.tensor([ni['l'][1], ni['w'][1]], dtype=torch.float32))
