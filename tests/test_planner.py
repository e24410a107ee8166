"""The rule-based planner.

  * oracle/planner.py (numpy restatement) against the reference's own HardcodeNuscPlanner.rollout on a synthetic lane graph
    (fixture g10, tests/golden/make_golden.py::g10_planner) -- CPU;
  * the product planner (strive_amd/planners/hardcode_goalcond_nusc.py -> strive_amd/csrc/planner.hip, ONE C-ABI call for all
    scenes) against the same fixture and against the oracle on larger random worlds (branching routes, objects off every
    lane, reversing and parked objects, observations that end early, two maps) -- through the host emulation of the kernels
    on CPU, and on the MI355X (-m gpu), there also at the 512-agent batch of adv_gen_rule_based.cfg.
"""
import os
import sys
import time

import numpy as np
import pytest
import torch

import make_golden as mg
from util import golden
from oracle import planner as oplan
from strive_amd import synth
from strive_amd import _lib as L
from strive_amd import ops
from strive_amd.planners.planner import PlannerConfig
from strive_amd.planners.hardcode_goalcond_nusc import HardcodeNuscPlanner, CONFIG_DICT, edge_grid

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))
os.environ['STRIVE_POISON_WS'] = '1'          # the planner's workspace starts as garbage in every test of this file


@pytest.fixture()
def emu_ops():
    import build as emu_build
    emu = L.StriveLib(emu_build.build(), require_all=True)
    orig = (ops._lib_for, L.get_lib)
    ops._lib_for = lambda *tensors: emu           # CPU tensors + the emulated library: test infrastructure only
    L.get_lib = lambda: emu
    from util import poison_new_workspaces, nan_empty
    orig_ws = poison_new_workspaces(ops)          # new scratch buffers start as NaN bytes, not as whatever torch.empty holds
    orig_empty, torch.empty = torch.empty, nan_empty()      # ... and so does every output allocated with torch.empty
    yield emu
    torch.empty = orig_empty
    ops._workspace = orig_ws
    ops._lib_for, L.get_lib = orig


# ------------------------------------------------------------------------------------------------
# inputs
# ------------------------------------------------------------------------------------------------

def random_world(sizes, key, nmaps=1, T=12, tail=True):
    """Scenes on the synthetic lane graph: agents on lane nodes, curved constant-speed futures (fp32 like the model's
    output), per scene one agent moved off every lane, one parked, one reversing, one whose observations end early."""
    lg = synth.make_lane_graph()
    t = np.linspace(0.5, 0.5 * T, T)
    states, atts, mask, obs = [], [], [], []
    for b, n in enumerate(sizes):
        cx = 128.0 + 40.0 * ((b % 3) - 1)
        cy = 128.0 + 40.0 * (((b // 3) % 3) - 1)
        px, py, h, s = synth.lane_scene_poses(lg, n, '%s/%d' % (key, b), radius=38.0, centre=(cx, cy))
        if n > 3:
            px[3] += 9.0; py[3] += 7.0; h[3] += 0.9
        if n > 2:
            s[2] = 0.0
        lw = np.stack([4.2 + 0.4 * synth.counter_uniform((n,), '%s/l%d' % (key, b)),
                       1.9 + 0.2 * synth.counter_uniform((n,), '%s/w%d' % (key, b))], -1)
        if n > 5:
            lw[5] = lw[5, ::-1] * np.array([0.5, 2.2])          # wider than long: boxes2circles swaps the axes
        states.append(np.stack([px, py, np.cos(h), np.sin(h), s, np.zeros(n)], -1))
        atts.append(lw)
        mask += [b] * n
        om = synth.counter_uniform((n - 1, 1), '%s/om%d' % (key, b), -0.03, 0.03)          # yaw rates
        hh = h[1:, None] + om * t[None]
        sp = s[1:, None] * np.ones((1, T))
        if n > 4:
            sp[3] = -2.5                                          # the object of index 4 backs up
        dx = np.cumsum(sp * np.cos(hh) * 0.5, axis=1)
        dy = np.cumsum(sp * np.sin(hh) * 0.5, axis=1)
        fut = np.stack([px[1:, None] + dx, py[1:, None] + dy, np.cos(hh), np.sin(hh)], -1)
        if tail and n > 2:
            fut[1, 7:] = np.nan
            if n > 6:
                fut[5, 0:] = np.nan                               # never observed: leaves after the first step
        obs.append(fut)
    ptr = np.concatenate([[0], np.cumsum([n - 1 for n in sizes])])
    map_idx = torch.tensor([b % nmaps for b in range(len(sizes))], dtype=torch.long)
    return (lg, synth.f32(np.concatenate(states)), synth.f32(np.concatenate(atts)), torch.tensor(mask),
            np.concatenate(obs).astype(np.float32), t, ptr, map_idx)


def both_planners(world, cfg_name, device='cpu'):
    lg, st, att, mask, obs, t, ptr, map_idx = world
    nmaps = int(map_idx.max()) + 1
    B = len(ptr) - 1
    cfg = CONFIG_DICT[cfg_name]
    orc = oplan.HardcodeNuscPlanner(mg._LaneEnv(lg, nmaps), oplan.PlannerConfig(**cfg))
    orc.reset(st, att, mask, B, map_idx)
    dev = HardcodeNuscPlanner(mg._LaneEnv(lg, nmaps), PlannerConfig(**cfg))
    dev.reset(st.to(device), att.to(device), mask.to(device), B, map_idx)
    return orc, dev


# ------------------------------------------------------------------------------------------------
# CPU: oracle against the reference, host logic
# ------------------------------------------------------------------------------------------------

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
    np.testing.assert_array_equal(oplan.LinearPath(t, y)(q), interp1d(t, y, axis=0, bounds_error=True, assume_sorted=True)(q))
    with pytest.raises(ValueError):
        oplan.LinearPath(t, y)(np.array([4.1]))


def test_oracle_planner_matches_reference():
    g = golden('g10_planner.npz')
    lg, st, att, mask, obs, t, ptr = mg.g10_inputs()
    for name in ('default', 'final_tuned_val_1'):
        pl = oplan.HardcodeNuscPlanner(mg._LaneEnv(lg), oplan.PlannerConfig(**oplan.CONFIG_DICT[name]))
        pl.reset(st, att, mask, len(mg.G10_SIZES), torch.zeros((len(mg.G10_SIZES),), dtype=torch.long))
        plan = pl.rollout(obs.copy(), t, ptr, t, control_all=False)
        assert plan.dtype == torch.float64 and tuple(plan.shape) == (2, 12, 4)
        np.testing.assert_allclose(plan.numpy(), g['plan_' + name], rtol=0, atol=1e-9)
        again = pl.rollout(obs.copy(), t, ptr, t, control_all=False)
        assert torch.equal(plan, again)
    # the plan moves along the ego's lane and never jumps
    step = np.linalg.norm(np.diff(g['plan_default'][:, :, :2], axis=1), axis=-1)
    assert step.max() < 0.5 * 20.0 + 1e-6


def test_clustered_matches_keep_one_per_connected_group():
    lg = synth.make_lane_graph()
    G = oplan.LaneGraph(lg)
    # a pose on a straight lane matches several consecutive edges; they are one cluster
    v = 40
    x, y = lg['xy'][v]
    nxt = lg['out_edges'][v][0]
    h = np.arctan2(*(lg['xy'][nxt] - lg['xy'][v])[::-1])
    e, p = G.match(x, y, h, 1.0 - np.cos(np.radians(20.0)), 2.0)
    assert len(e) >= 2
    ke, kp = G.cluster(x, y, e, p)
    assert len(ke) == 1 and np.linalg.norm(kp[0] - np.array([x, y])) < 1e-9


def test_edge_grid_finds_every_edge_a_full_scan_finds():
    """the device reads one grid cell instead of scanning all edges (reference :298-322): same matches, same order"""
    lg = synth.make_lane_graph()
    G = oplan.LaneGraph(lg)
    grid = edge_grid(np.asarray(lg['edges'], dtype=np.float64), 2.0, 4.0)
    ptr, ce = grid['cell_ptr'], grid['cell_edges']
    assert ptr[-1] == len(ce) and np.all(np.diff(ptr) >= 0)
    q = synth.counter_uniform((4000, 3), 'grid/q', 0.0, 1.0)
    ee = {tuple(int(v) for v in e): k for k, e in enumerate(lg['edgeixes'])}
    nonempty = 0
    for x, y, h in q * np.array([270.0, 270.0, 2 * np.pi]) - np.array([7.0, 7.0, np.pi]):
        e, _ = G.match(x, y, h, 2.0, 2.0)            # cdistmax 2: every heading passes, only the distance test remains
        want = [ee[tuple(int(v) for v in k)] for k in e]
        fx, fy = np.floor((x - grid['gx0']) / grid['gcell']), np.floor((y - grid['gy0']) / grid['gcell'])
        if not (0 <= fx < grid['gnx'] and 0 <= fy < grid['gny']):
            assert want == []
            continue
        c = int(fy) * grid['gnx'] + int(fx)
        cand = ce[ptr[c]:ptr[c + 1]]
        assert np.all(np.diff(cand) > 0)
        assert set(want) <= set(cand.tolist())
        nonempty += len(want) > 0
    assert nonempty > 500


# ------------------------------------------------------------------------------------------------
# the device planner through the host emulation of its kernels (CPU)
# ------------------------------------------------------------------------------------------------

def test_device_planner_emulated_matches_reference(emu_ops):
    g = golden('g10_planner.npz')
    lg, st, att, mask, obs, t, ptr = mg.g10_inputs()
    for name in ('default', 'final_tuned_val_1'):
        pl = HardcodeNuscPlanner(mg._LaneEnv(lg), PlannerConfig(**CONFIG_DICT[name]))
        pl.reset(st, att, mask, len(mg.G10_SIZES), torch.zeros((len(mg.G10_SIZES),), dtype=torch.long))
        plan = pl.rollout(obs.copy(), t, ptr, t, control_all=False)
        assert plan.dtype == torch.float64 and tuple(plan.shape) == (2, 12, 4)
        np.testing.assert_allclose(plan.numpy(), g['plan_' + name], rtol=0, atol=1e-9)
        again = pl.rollout(torch.from_numpy(obs.copy()), t, ptr, t, control_all=False)      # tensor input, reset state untouched
        assert torch.equal(plan, again)


@pytest.mark.parametrize('cfg_name', ['default', 'final_tuned_val_1'])
def test_device_planner_emulated_random_world(emu_ops, cfg_name):
    world = random_world([7, 4, 1, 9], 'pw/' + cfg_name, nmaps=2)
    orc, dev = both_planners(world, cfg_name)
    _, _, _, _, obs, t, ptr, _ = world
    want = orc.rollout(obs.copy(), t, ptr, t, control_all=False)
    got = dev.rollout(obs.copy(), t, ptr, t, control_all=False)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-9)
    # shorter horizon / different output times (refine-style 16 x 0.5 s is covered by the GPU test)
    t2 = np.array([0.4, 1.0, 2.0])
    want = orc.rollout(obs.copy(), t, ptr, t2, control_all=False)
    got = dev.rollout(obs.copy(), t, ptr, t2, control_all=False)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-9)


def test_device_planner_with_the_ego_in_the_middle_of_its_scene(emu_ops):
    """reset(..., ego_idx=2): the observed agents are rows 0, 1, 3, 4 of every scene (reference :109-127, create_other_agents)"""
    lg = synth.make_lane_graph()
    t = np.linspace(0.5, 6.0, 12)
    states, atts, mask, obs = [], [], [], []
    for b, n in enumerate((5, 4)):
        px, py, h, s = synth.lane_scene_poses(lg, n, 'pe/%d' % b, radius=30.0, centre=(128.0 + 40.0 * b, 128.0))
        lw = np.stack([4.2 + 0.4 * synth.counter_uniform((n,), 'pe/l%d' % b), 1.9 + 0.2 * synth.counter_uniform((n,), 'pe/w%d' % b)], -1)
        states.append(np.stack([px, py, np.cos(h), np.sin(h), s, np.zeros(n)], -1))
        atts.append(lw)
        mask += [b] * n
        fut = np.stack([px[:, None] + s[:, None] * np.cos(h[:, None]) * t[None], py[:, None] + s[:, None] * np.sin(h[:, None]) * t[None],
                        np.broadcast_to(np.cos(h[:, None]), (n, 12)), np.broadcast_to(np.sin(h[:, None]), (n, 12))], -1)
        obs.append(np.delete(fut, 2, axis=0))
    st, att = synth.f32(np.concatenate(states)), synth.f32(np.concatenate(atts))
    obs = np.concatenate(obs).astype(np.float32)
    ptr = np.array([0, 4, 7])
    mi = torch.zeros((2,), dtype=torch.long)
    cfg = CONFIG_DICT['default']
    orc = oplan.HardcodeNuscPlanner(mg._LaneEnv(lg), oplan.PlannerConfig(**cfg))
    orc.reset(st, att, torch.tensor(mask), 2, mi, ego_idx=2)
    dev = HardcodeNuscPlanner(mg._LaneEnv(lg), PlannerConfig(**cfg))
    dev.reset(st, att, torch.tensor(mask), 2, mi, ego_idx=2)
    want = orc.rollout(obs.copy(), t, ptr, t, control_all=False)
    got = dev.rollout(obs.copy(), t, ptr, t, control_all=False)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-9)
    # and it is the ego's lane the plan follows: the first pose lies within one step of agent 2's start
    assert np.linalg.norm(got[0, 0, :2].numpy() - states[0][2, :2]) < 0.5 * 10.0


def test_device_planner_routes_match_oracle(emu_ops):
    """every route of a pose (matches -> clusters -> chains -> blended arc-length path), knot by knot"""
    lg = synth.make_lane_graph()
    G = oplan.LaneGraph(lg)
    cfgd = CONFIG_DICT['final_tuned_val_1']
    cfg = PlannerConfig(**cfgd)
    world = random_world([3], 'rt')
    pl = HardcodeNuscPlanner(mg._LaneEnv(lg), cfg)
    pl.reset(world[1], world[2], world[3], 1, torch.zeros((1,), dtype=torch.long))
    row_obj, row_scene, NR = pl._row_maps(world[6])
    desc = pl._descriptor(row_obj, row_scene, NR)
    tmax = cfg.nsteps * cfg.preddt
    cdistmax = 1.0 - np.cos(np.radians(cfg.cdistang))
    poses = [(lg['xy'][40, 0] + 0.3, lg['xy'][40, 1] - 0.4, None, 6.0), (lg['xy'][200, 0], lg['xy'][200, 1] + 0.8, None, 0.0),
             (lg['xy'][333, 0] - 0.5, lg['xy'][333, 1], None, -3.0), (50.0, 50.0, 0.7, 4.0), (lg['xy'][75, 0], lg['xy'][75, 1], None, 27.0)]
    nbranch = 0
    for x, y, h, s in poses:
        if h is None:
            v = int(np.argmin(np.linalg.norm(lg['xy'] - np.array([[x, y]]), axis=1)))
            d = lg['xy'][lg['out_edges'][v][0]] - lg['xy'][v] if lg['out_edges'][v] else lg['xy'][v] - lg['xy'][lg['in_edges'][v][0]]
            h = float(np.arctan2(d[1], d[0])) + 0.05
        e, p = G.match(x, y, h, cdistmax, cfg.xydistmax)
        e, p = G.cluster(x, y, e, p)
        back = 1.0 if s > 0 else 1.0 + abs(s) * tmax
        fwd = 1.0 + cfg.smax * tmax if s < 0 else max(1.0 + cfg.smax * tmax, 1.0 + s * tmax)
        want = oplan.routes_through(G, e, p, back, fwd, cfg.xydistmax, np.array([x, y]), h)
        maxr, maxk = 64, 384
        nr = torch.zeros((1,), dtype=torch.int32)
        nk = torch.zeros((maxr,), dtype=torch.int32)
        kn = torch.zeros((maxr, maxk, 5), dtype=torch.float64)
        status = torch.zeros((8,), dtype=torch.int32)
        emu_ops.call('strive_planner_routes', desc, 0, L.ptr(torch.tensor([x, y, h, s], dtype=torch.float64)), maxr, maxk, L.ptr(nr),
                     L.ptr(nk), L.ptr(kn), L.ptr(status), None)
        assert int(status.abs().sum()) == 0, status
        assert int(nr) == len(want)
        nbranch += len(want) > 1
        for r, route in enumerate(want):
            n = len(route.t)
            assert int(nk[r]) == n
            np.testing.assert_allclose(kn[r, :n, 0].numpy(), route.t, rtol=0, atol=1e-10)
            np.testing.assert_allclose(kn[r, :n, 1:].numpy(), route.y, rtol=0, atol=1e-10)
    assert nbranch >= 2


def test_device_planner_reports_exceeded_limits(emu_ops):
    """a world the fixed-size tables cannot hold (an object at 60 m/s needs more route knots than the kernel keeps) is
    reported, not silently mis-planned"""
    world = list(random_world([3], 'lim', tail=False))
    st = world[1].clone()
    st[1, 4] = 60.0
    world[1] = st
    _, dev = both_planners(tuple(world), 'default')
    with pytest.raises(L.StriveHipError, match='route knots'):
        dev.rollout(world[4].copy(), world[5], world[6], world[5], control_all=False)


def test_device_planner_failure_is_per_scene(emu_ops):
    """One scene of a batch exceeds a limit (an object at 60 m/s needs more route knots than the kernel keeps; the reference's
    planner raises for that scene's rollout, which with its batch_size-1 runs costs that scene, adv_scenario_gen.py:540-543):
    the status row, the NaN plan and the ``alive`` flag of THAT scene say so; every other scene gets, bit for bit, the plan of
    the batch without the failing one; 'report' mode names the scene, 'raise' mode names it in the error."""
    sizes = [3, 4, 2]
    world = list(random_world(sizes, 'lim3', tail=False))
    lg, st, att, mask, obs, t, ptr, mi = world
    clean = both_planners(tuple(world), 'default')[1]
    clean.defer_check = True
    want = clean.rollout(torch.from_numpy(obs.copy()), t, ptr, t, control_all=False)
    assert clean.check() == {} and bool(clean.alive.all())
    bad_scene = 1
    st2 = st.clone()
    st2[sizes[0] + 1, 4] = 60.0                    # a non-ego object of scene 1
    world[1] = st2
    dev = both_planners(tuple(world), 'default')[1]
    dev.defer_check = True
    got = dev.rollout(torch.from_numpy(obs.copy()), t, ptr, t, control_all=False)
    assert dev.alive.tolist() == [1, 0, 1]
    assert bool(torch.isnan(got[bad_scene]).all()) and not bool(torch.isnan(got[[0, 2]]).any())
    assert torch.equal(got[[0, 2]], want[[0, 2]])
    failed = dev.check(on_error='report')
    assert list(failed) == [bad_scene] and any('route knots' in n for n in failed[bad_scene])
    assert dev.failed_scenes() == failed, "'report' leaves the flags set"
    # the flags are sticky per scene: a later clean rollout of the same world keeps scene 1 out, and only scene 1
    world[1] = st
    dev._world['init'] = clean._world['init']
    dev.on_error = 'report'                        # (what the quarantining loop sets: rollout's own look at the flags must not raise)
    again = dev.rollout(torch.from_numpy(obs.copy()), t, ptr, t, control_all=False)
    assert dev.alive.tolist() == [1, 0, 1] and torch.equal(again, want)
    with pytest.raises(L.StriveHipError, match=r'scene\(s\) 1: .*route knots'):
        dev.check(on_error='raise')
    assert dev.check() == {} and dev.alive.tolist() == [1, 1, 1]          # cleared by the raising check


def test_device_planner_status_of_earlier_rollouts_is_kept(emu_ops):
    """The optimisation loops look at the planner's status once, after their last iteration.  The flags live in one status
    tensor per planner that the kernels only set: a limit exceeded in an EARLIER rollout is still reported after later,
    clean rollouts (it used to be overwritten by the newest rollout's status), and a raising check() clears it."""
    world = list(random_world([3, 2], 'lim2', tail=False))
    _, dev = both_planners(tuple(world), 'default')
    dev.defer_check = True                      # what rollouts on device tensors do inside the optimisation loops
    obs = torch.from_numpy(world[4].copy())
    dev.rollout(obs, world[5], world[6], world[5], control_all=False)
    dev.check()                                 # clean
    dev.traj_cap = 1                            # one rollout with a trajectory list that cannot hold a scene's predictions
    plan_bad = dev.rollout(obs, world[5], world[6], world[5], control_all=False)
    dev.traj_cap = type(dev).traj_cap
    assert bool(torch.isnan(plan_bad).any()), 'the overflowing rollout must poison its plan'
    # it surfaces at a later rollout's non-blocking look at the flags (here, on the host device, the very next one; on the GPU
    # when the asynchronous snapshot has arrived) or at the loop's final check() -- never lost behind later, clean rollouts
    real_check = dev.check
    dev.check = lambda wait=True, on_error=None: real_check(wait, on_error=on_error) if wait else None   # loop without the opportunistic look
    for _ in range(2):
        plan = dev.rollout(obs, world[5], world[6], world[5], control_all=False)       # later rollouts are clean
        assert not bool(torch.isnan(plan).any())
    dev.check = real_check
    with pytest.raises(L.StriveHipError, match='traj'):
        dev.check()
    dev.check()                                 # cleared by the raising check


# ------------------------------------------------------------------------------------------------
# MI355X
# ------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_device_planner_gpu_matches_reference_and_oracle():
    dev = 'cuda:0'
    g = golden('g10_planner.npz')
    lg, st, att, mask, obs, t, ptr = mg.g10_inputs()
    for name in ('default', 'final_tuned_val_1'):
        pl = HardcodeNuscPlanner(mg._LaneEnv(lg), PlannerConfig(**CONFIG_DICT[name]))
        pl.reset(st.to(dev), att.to(dev), mask.to(dev), len(mg.G10_SIZES), torch.zeros((len(mg.G10_SIZES),), dtype=torch.long))
        plan = pl.rollout(torch.from_numpy(obs.copy()).to(dev), t, ptr, t, control_all=False)
        assert plan.is_cuda and plan.dtype == torch.float64
        pl.check()
        np.testing.assert_allclose(plan.cpu().numpy(), g['plan_' + name], rtol=0, atol=1e-9)
        host = pl.rollout(obs.copy(), t, ptr, t, control_all=False)                          # the reference's calling convention
        assert not host.is_cuda and torch.equal(host, plan.cpu())
    for cfg_name, sizes in (('default', [7, 4, 1, 9]), ('final_tuned_val_1', [12, 3, 16, 6, 2, 8])):
        world = random_world(sizes, 'pw/' + cfg_name, nmaps=2)
        orc, prod = both_planners(world, cfg_name, device=dev)
        _, _, _, _, obs, t, ptr, _ = world
        for tq in (t, np.linspace(0.5, 8.0, 16)[:12]):
            want = orc.rollout(obs.copy(), t, ptr, tq, control_all=False)
            got = prod.rollout(torch.from_numpy(obs.copy()).to(dev), t, ptr, tq, control_all=False)
            prod.check()
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=1e-9)


# scenes of the 512-agent world in which the reference's route construction raises (interp1d bounds error, :433-556: an
# object whose closest lane point lies more than the 2 m slack behind its matched edge) -- computed by the oracle
# (test_oracle_rejects_these_scenes_of_the_512_agent_world, slow); the device reports exactly these and plans the others
P512_REJECTED = [0, 2, 3, 15, 18, 26, 29, 30, 34]


def p512_world():
    import bench
    sizes = bench.variable_scene_sizes(512, 'bench/adv/r0')
    return sizes, random_world(sizes, 'p512')


def sub_world(world, sizes, pick):
    lg, st, att, mask, obs, t, ptr, _ = world
    sub_mask = torch.cat([torch.full((sizes[b],), i, dtype=torch.long) for i, b in enumerate(pick)])
    rows = torch.cat([mask.eq(b).nonzero().flatten() for b in pick])
    orow = np.concatenate([np.arange(ptr[b], ptr[b + 1]) for b in pick])
    sub_ptr = np.concatenate([[0], np.cumsum([sizes[b] - 1 for b in pick])])
    return (lg, st[rows], att[rows], sub_mask, obs[orow], t, sub_ptr, torch.zeros((len(pick),), dtype=torch.long))


@pytest.mark.slow
def test_oracle_rejects_these_scenes_of_the_512_agent_world():
    sizes, world = p512_world()
    bad = []
    for b in range(len(sizes)):
        sub = sub_world(world, sizes, [b])
        orc = oplan.HardcodeNuscPlanner(mg._LaneEnv(sub[0]), oplan.PlannerConfig(**oplan.CONFIG_DICT['default']))
        orc.reset(sub[1], sub[2], sub[3], 1, sub[7])
        try:
            orc.rollout(sub[4].copy(), sub[5], sub[6], sub[5], control_all=False)
        except ValueError:
            bad.append(b)
    assert bad == P512_REJECTED


@pytest.mark.gpu
def test_device_planner_gpu_512_agents():
    """the batch of adv_gen_rule_based.cfg: ~512 agents in scenes of 2..30.  The scenes the reference's route construction
    rejects are reported (NaN plan + error on check), every other scene is planned: reproducibly, independently of the batch it
    is planned in, and -- a sample -- equal to the oracle."""
    dev = 'cuda:0'
    sizes, world = p512_world()
    lg, st, att, mask, obs, t, ptr, map_idx = world
    _, prod = both_planners(world, 'default', device=dev)
    obs_d = torch.from_numpy(obs.copy()).to(dev)
    got = prod.rollout(obs_d, t, ptr, t, control_all=False)
    failed = prod.check(on_error='report')
    assert sorted(failed) == P512_REJECTED and all(any('outside a route' in n for n in v) for v in failed.values())
    assert np.nonzero(prod.alive.cpu().numpy() == 0)[0].tolist() == P512_REJECTED
    with pytest.raises(L.StriveHipError, match='outside a route'):
        prod.check()
    torch.cuda.synchronize()
    t0 = time.time()
    again = prod.rollout(obs_d, t, ptr, t, control_all=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    prod._pending = None
    nan_scene = torch.isnan(got).flatten(1).any(1).cpu().numpy()
    assert np.nonzero(nan_scene)[0].tolist() == P512_REJECTED
    ok = torch.from_numpy(~nan_scene).to(dev)
    assert torch.equal(got[ok], again[ok])
    print('device planner: %d scenes / %d agents, 31 planner steps in %.2f ms' % (len(sizes), sum(sizes), 1e3 * dt))
    pick = [1, 5, 11]
    sub = sub_world(world, sizes, pick)
    o3, p3 = both_planners(sub, 'default', device=dev)
    want = o3.rollout(sub[4].copy(), t, sub[6], t, control_all=False)
    np.testing.assert_allclose(got[pick].cpu().numpy(), want.numpy(), rtol=0, atol=1e-9)
    alone = p3.rollout(torch.from_numpy(sub[4].copy()).to(dev), t, sub[6], t, control_all=False)
    p3.check()
    assert torch.equal(alone, got[pick])
    # an oracle-rejected scene raises there too
    bad = sub_world(world, sizes, [15])
    o1, _ = both_planners(bad, 'default', device=dev)
    with pytest.raises(ValueError):
        o1.rollout(bad[4].copy(), t, bad[6], t, control_all=False)


def test_bench_world_is_one_the_oracle_planner_completes():
    """bench.py's closed-loop workloads (--planner hardcode, --workload full) are only a statement about adv_gen_rule_based.cfg if the
    rule-based planner can drive their synthetic world.  Every third scene of the 512-agent bench batch (lane graph, poses 10 m apart
    at 4..6 m/s, the lane-keeping synthetic weights): the ORACLE model's decode of the posterior mean (what iteration 0 of the
    adversarial loop hands the planner, reference src/utils/adv_gen_optim.py:133-139) goes through the ORACLE's restatement of
    HardcodeNuscPlanner.rollout scene by scene -- which raises, like the reference, when an object leaves its route.  At most one in ten
    scenes may be rejected (with the plain random weights the agents swing off their lanes and a third is)."""
    import bench
    from oracle.planner import HardcodeNuscPlanner as OPlanner, PlannerConfig as OConfig, CONFIG_DICT as OCFG
    from util import oracle_model, product_model
    from strive_amd.graph import Batch
    sizes = bench.variable_scene_sizes(512, 'bench/adv/r0')
    own = [(n, 'bench/adv/r0/%d' % b) for b, n in enumerate(sizes)]
    lg = synth.make_lane_graph(extent=1024.0)
    batch, _ = bench.build_batch(own, 2, 4096, lane_graph=lg)
    scenes = batch.to_data_list()[0::3]
    sub = Batch.from_data_list(scenes)
    B = len(scenes)
    map_idx = torch.zeros((B,), dtype=torch.long)
    raster, dx = synth.make_raster(4096, 4096)
    env = synth.SyntheticMapEnv(raster, dx, lane_graph=lg)
    _, sd = product_model()
    orc = oracle_model(synth.lane_keeping_weights(sd))
    with torch.no_grad():
        emb = orc.embed(sub, map_idx, env)
        pred = orc.decode_embedding(emb['posterior_out'][0], emb, sub, map_idx, env, nfuture=12)['future_pred']
    unn = orc.get_normalizer().unnormalize
    fut = unn(pred).numpy()                                            # (NA, 12, 4)
    st = unn(sub.past_gt[:, -1, :])
    att = orc.get_att_normalizer().unnormalize(sub.lw)
    ptr = sub.ptr.numpy()
    t = np.linspace(0.5, 6.0, 12)
    rejected = []
    for b in range(B):
        lo, hi = int(ptr[b]), int(ptr[b + 1])
        pl = OPlanner(mg._LaneEnv(lg), OConfig(**OCFG['default']))
        pl.reset(st[lo:hi], att[lo:hi], torch.zeros((hi - lo,), dtype=torch.long), 1, torch.zeros((1,), dtype=torch.long))
        try:
            plan = pl.rollout(fut[lo + 1:hi].astype(np.float32), t, np.array([0, hi - lo - 1]), t, control_all=False)
            assert np.isfinite(np.asarray(plan)).all()
        except (AssertionError, ValueError) as e:
            rejected.append((b, str(e)[:80]))
    print('bench world, %d of %d scenes: the oracle planner rejects %d %s' % (B, len(sizes), len(rejected), rejected))
    assert len(rejected) <= B // 10, rejected
