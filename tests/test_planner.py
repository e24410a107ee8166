"""The rule-based planner: strive_amd.planners.hardcode_goalcond_nusc against the reference's own
HardcodeNuscPlanner.rollout on a synthetic lane graph (fixture g10, tests/golden/make_golden.py::g10_planner)."""
import time

import numpy as np
import torch

import make_golden as mg
from util import golden
from strive_amd.planners.planner import PlannerConfig
from strive_amd.planners.hardcode_goalcond_nusc import HardcodeNuscPlanner, CONFIG_DICT, LinearPath, LaneGraph
from strive_amd import synth


def test_lane_graph_format():
    lg = synth.make_lane_graph()
    n, m = lg['xy'].shape[0], lg['edges'].shape[0]
    assert lg['edgeixes'].shape == (m, 2) and len(lg['in_edges']) == len(lg['out_edges']) == n and len(lg['ee2ix']) == m
    assert sum(len(o) for o in lg['out_edges']) == m == sum(len(i) for i in lg['in_edges'])
    for k, (v0, v1) in enumerate(lg['edgeixes'][:200]):
        assert lg['ee2ix'][(int(v0), int(v1))] == k and int(v0) in lg['in_edges'][int(v1)]
        d = lg['xy'][v1] - lg['xy'][v0]
        np.testing.assert_allclose(lg['edges'][k], [lg['xy'][v0, 0], lg['xy'][v0, 1], d[0] / np.linalg.norm(d), d[1] / np.linalg.norm(d),
                                                    np.linalg.norm(d)], atol=1e-12)
    assert max(len(o) for o in lg['out_edges']) == 2 and max(len(i) for i in lg['in_edges']) == 2     # branches and merges exist


def test_linear_path_is_interp1d():
    from scipy.interpolate import interp1d
    t = np.array([-3.0, -1.0, 0.0, 0.5, 4.0])
    y = synth.counter_uniform((5, 4), 'lp/y', -2.0, 2.0)
    q = np.array([-3.0, -2.2, -1.0, 0.0, 0.25, 3.999, 4.0])
    np.testing.assert_array_equal(LinearPath(t, y)(q), interp1d(t, y, axis=0, bounds_error=True, assume_sorted=True)(q))
    try:
        LinearPath(t, y)(np.array([4.1]))
        raise AssertionError('out-of-range query must raise')
    except ValueError:
        pass


def test_planner_rollout_matches_reference():
    g = golden('g10_planner.npz')
    lg, st, att, mask, obs, t, ptr = mg.g10_inputs()
    for name in ('default', 'final_tuned_val_1'):
        pl = HardcodeNuscPlanner(mg._LaneEnv(lg), PlannerConfig(**CONFIG_DICT[name]))
        pl.reset(st, att, mask, len(mg.G10_SIZES), torch.zeros((len(mg.G10_SIZES),), dtype=torch.long))
        t0 = time.time()
        plan = pl.rollout(obs.copy(), t, ptr, t, control_all=False)
        dt = time.time() - t0
        assert plan.dtype == torch.float64 and tuple(plan.shape) == (2, 12, 4)
        np.testing.assert_allclose(plan.numpy(), g['plan_' + name], rtol=0, atol=1e-9)
        again = pl.rollout(obs.copy(), t, ptr, t, control_all=False)
        assert torch.equal(plan, again)
        print('%s: 2 scenes x 31 planner steps in %.2f s' % (name, dt))
    # the plan moves along the ego's lane and never jumps
    step = np.linalg.norm(np.diff(g['plan_default'][:, :, :2], axis=1), axis=-1)
    assert step.max() < 0.5 * 20.0 + 1e-6


def test_clustered_matches_keep_one_per_connected_group():
    lg = synth.make_lane_graph()
    G = LaneGraph(lg)
    # a pose on a straight lane matches several consecutive edges; they are one cluster
    v = 40
    x, y = lg['xy'][v]
    nxt = lg['out_edges'][v][0]
    h = np.arctan2(*(lg['xy'][nxt] - lg['xy'][v])[::-1])
    e, p = G.match(x, y, h, 1.0 - np.cos(np.radians(20.0)), 2.0)
    assert len(e) >= 2
    ke, kp = G.cluster(x, y, e, p)
    assert len(ke) == 1 and np.linalg.norm(kp[0] - np.array([x, y])) < 1e-9
