"""Rule-based lane-following planner (reference src/planners/hardcode_goalcond_nusc.py), the planner
``adv_gen_rule_based.cfg`` attacks in closed loop (src/utils/adv_gen_optim.py:133-139: one ``rollout`` per optimisation
iteration) -- same class name, constructor, ``reset`` / ``rollout`` signatures, configuration dictionaries and outputs.

The reference walks its scenes one after the other in numpy (31 planner steps per scene, ~25 ms each).  Here ``rollout`` is
ONE C-ABI call (``strive_planner_rollout``, strive_amd/csrc/planner.hip) for all scenes of the batch: the non-ego objects'
lane routes and predicted trajectories of every planner step are built in one launch (only the ego's pose chains the
steps), then two launches per planner step score the ego's speed profiles and move it.  float64, numpy's operation order.

Host work happens once, in ``reset`` (reference :109-127): the initial world of every scene, and per map the lane graph
packed for the device (node connection records, CSR lists, the edge table and a uniform grid over the edges that replaces
get_lane_matches' scan of all edges, :298-322).  ``rollout`` accepts the futures of the non-ego agents as a device tensor
(no device->host copy in the optimisation loop) or, like the reference, as a numpy array.
"""
import ctypes as C
import os

import numpy as np
import torch

from .planner import PlannerNusc, PlannerConfig
from .. import _lib as L
from .. import ops

DEF_CONFIG = {
    'dt': 0.2, 'preddt': 0.2, 'nsteps': 25, 'cdistang': 20.0, 'xydistmax': 2.0, 'smax': 15.0, 'accmax': 3.0,
    'predsfacs': [0.5, 1.0], 'predafacs': [0.5], 'interacdist': 70.0, 'planaccfacs': [1.0], 'plannspeeds': 5,
    'col_plim': 0.1, 'score_wmin': 0.7, 'score_wfac': 0.05,
}
# tuned on generated validation scenarios (reference :43-59)
TUNED_VAL_FINAL_1 = dict(DEF_CONFIG, smax=20.0, accmax=4.0, score_wmin=0.3, score_wfac=0.02)
CONFIG_DICT = {'default': DEF_CONFIG, 'final_tuned_val_1': TUNED_VAL_FINAL_1}

STATUS_NAMES = ('lane matches of one pose (> 96)', 'match clusters (> 16)', 'lane chains of one match (> 48 forward / 16 backward)',
                'nodes of one chain (> 316)', 'route knots (> 384: speed above ~28 m/s)', 'arc length outside a route / time outside the plan',
                'predicted trajectories per scene and step (raise traj_cap)', 'action speed check')

LANE_NODE = np.dtype([('n', '<i4'), ('node', '<i4', (4,)), ('pad', '<i4', (3,)), ('len', '<f8', (4,))])
assert LANE_NODE.itemsize == 64


def _connection_records(lists, lengths_of):
    n = len(lists)
    rec = np.zeros((n,), dtype=LANE_NODE)
    ptr = np.zeros((n + 1,), dtype=np.int32)
    idx, lens = [], []
    for v, conn in enumerate(lists):
        rec['n'][v] = len(conn)
        for j, c in enumerate(conn):
            ln = lengths_of(v, int(c))
            if j < 4:
                rec['node'][v, j] = c
                rec['len'][v, j] = ln
            idx.append(int(c))
            lens.append(ln)
        ptr[v + 1] = len(idx)
    return rec, ptr, np.asarray(idx, dtype=np.int32).reshape(-1), np.asarray(lens, dtype=np.float64).reshape(-1)


def edge_grid(edges, margin, cell):
    """Uniform grid over the directed edges (x0, y0, dx, dy, len): every edge is listed, ascending, in all cells its bounding
    box grown by ``margin`` touches.  A point closer than ``margin`` to an edge therefore finds it in its own cell."""
    p0 = edges[:, 0:2]
    p1 = p0 + edges[:, 4:5] * edges[:, 2:4]
    pad = margin * (1.0 + 1e-9) + 1e-6
    lo = np.minimum(p0, p1) - pad
    hi = np.maximum(p0, p1) + pad
    gx0, gy0 = float(np.floor(lo[:, 0].min())), float(np.floor(lo[:, 1].min()))
    ix0 = np.floor((lo[:, 0] - gx0) / cell).astype(np.int64)
    ix1 = np.floor((hi[:, 0] - gx0) / cell).astype(np.int64)
    iy0 = np.floor((lo[:, 1] - gy0) / cell).astype(np.int64)
    iy1 = np.floor((hi[:, 1] - gy0) / cell).astype(np.int64)
    gnx, gny = int(ix1.max()) + 1, int(iy1.max()) + 1
    cells, eids = [], []
    e = np.arange(edges.shape[0], dtype=np.int64)
    for oy in range(int((iy1 - iy0).max()) + 1):
        for ox in range(int((ix1 - ix0).max()) + 1):
            sel = (ix0 + ox <= ix1) & (iy0 + oy <= iy1)
            cells.append((iy0[sel] + oy) * gnx + ix0[sel] + ox)
            eids.append(e[sel])
    cells, eids = np.concatenate(cells), np.concatenate(eids)
    order = np.lexsort((eids, cells))
    cells, eids = cells[order], eids[order]
    ptr = np.zeros((gnx * gny + 1,), dtype=np.int32)
    np.add.at(ptr, cells + 1, 1)
    return dict(gx0=gx0, gy0=gy0, gcell=float(cell), gnx=gnx, gny=gny, cell_ptr=np.cumsum(ptr, dtype=np.int64).astype(np.int32),
                cell_edges=eids.astype(np.int32))


class PackedLaneGraph(object):
    """One map's lane-graph dict (reference src/datasets/nuscenes_utils.py:50-123) as device arrays + StrivePlannerMap."""

    def __init__(self, lg, xydistmax, device, cell=4.0):
        xy = np.ascontiguousarray(np.asarray(lg['xy'], dtype=np.float64))
        edges = np.ascontiguousarray(np.asarray(lg['edges'], dtype=np.float64))
        eix = np.asarray(lg['edgeixes'], dtype=np.int64)
        # |xy[b] - xy[a]| as the reference's expand_verts evaluates it (np.linalg.norm of the 1-D difference, :395-399)
        cache = {}

        def length(a, b):
            key = (a, b) if a < b else (b, a)
            v = cache.get(key)
            if v is None:
                v = float(np.linalg.norm(xy[b] - xy[a]))
                cache[key] = v
            return v
        succ, sp, si, sl = _connection_records(lg['out_edges'], length)
        pred, pp, pi, pl = _connection_records(lg['in_edges'], length)
        grid = edge_grid(edges, float(xydistmax), float(cell))
        host = dict(xy=xy, succ=succ, pred=pred, succ_ptr=sp, succ_idx=si, succ_len=sl, pred_ptr=pp, pred_idx=pi, pred_len=pl,
                    edges=edges, edge_ix=eix.astype(np.int32), cell_ptr=grid['cell_ptr'], cell_edges=grid['cell_edges'])
        self.t = {}
        for k, v in host.items():
            a = np.ascontiguousarray(v)
            if a.size == 0:
                a = np.zeros((1,), dtype=a.dtype)          # never hand a NULL pointer to the library
            self.t[k] = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)
        self.N, self.M = xy.shape[0], edges.shape[0]
        self.grid = grid
        self.xydistmax = float(xydistmax)

    def fill(self, m):
        for k, t in self.t.items():
            setattr(m, k, t.data_ptr())
        m.N, m.M, m.gnx, m.gny = self.N, self.M, self.grid['gnx'], self.grid['gny']
        m.gx0, m.gy0, m.gcell = self.grid['gx0'], self.grid['gy0'], self.grid['gcell']


def planner_cfg_struct(cfg):
    c = L.StrivePlannerCfg()
    for k in ('dt', 'preddt', 'xydistmax', 'smax', 'accmax', 'interacdist', 'col_plim', 'score_wmin', 'score_wfac'):
        setattr(c, k, float(getattr(cfg, k)))
    c.cdistmax = float(1.0 - np.cos(np.radians(cfg.cdistang)))
    c.tmax = float(cfg.nsteps * cfg.preddt)
    for name, cnt in (('predsfacs', 'npredsfacs'), ('predafacs', 'npredafacs'), ('planaccfacs', 'nplanaccfacs')):
        vals = [float(v) for v in getattr(cfg, name)]
        if not 1 <= len(vals) <= 4:
            raise NotImplementedError('planner config %s: 1..4 entries are supported, got %d' % (name, len(vals)))
        for i, v in enumerate(vals):
            getattr(c, name)[i] = v
        setattr(c, cnt, len(vals))
    c.nsteps, c.plannspeeds = int(cfg.nsteps), int(cfg.plannspeeds)
    return c


class HardcodeNuscPlanner(PlannerNusc):
    traj_cap = 512          # predicted trajectories kept per scene and planner step (objects x routes x speed profiles)

    def __init__(self, map_env, cfg):
        super(HardcodeNuscPlanner, self).__init__(map_env, cfg)
        assert isinstance(self.cfg, PlannerConfig)
        self.lane_graphs = self.map_env.lane_graphs
        self._graphs = {}
        self.B = self.batch_mask = self.batch_maps = None
        self.ego_idx = 0
        self._world = None
        self._pending = None      # (pinned snapshot of the status flags, event) of a read-back that is under way, or None
        self._status = None       # device flags (B, 8): the kernels only SET them, so one tensor accumulates every rollout since
                                  # reset(); row b belongs to scene b alone
        self.alive = None         # device (B,) uint8, written at the end of every rollout: 1 = no flag of the scene is set.  The
                                  # closed loop hands it to the loss kernels (no host round trip): a failed scene is quarantined
        self.on_error = 'raise'   # what check() does with failed scenes: 'raise' (the reference's numpy planner raises) or
                                  # 'report' (return them: the caller drops those scenes, utils/adv_gen_optim.py)
        self.defer_check = None   # None: rollouts on device tensors defer the status check to check() (no synchronisation inside
                                  # an optimisation closure), host calls check at once; True / False force either
        self._rows = {}
        self._tables = {}

    # ---- reset: initial world + lane graphs on the device (reference :80-127) -----------------------------------------
    def reset(self, init_state, vehicle_atts, batch_mask, batch_size, map_idx, ego_idx=0):
        """init_state (NA,6) UNNORMALISED (x, y, hx, hy, s, hdot), vehicle_atts (NA,2) (l, w), batch_mask (NA) scene of every
        agent, map_idx (B).  The device the planner runs on is init_state's."""
        dev = init_state.device
        ops._lib_for(init_state)
        self._status = torch.zeros((int(batch_size), len(STATUS_NAMES)), dtype=torch.int32, device=dev)
        self.alive = torch.ones((int(batch_size),), dtype=torch.uint8, device=dev)
        self._pending = None
        self.ego_idx, self.B, self.batch_mask = int(ego_idx), int(batch_size), batch_mask
        state = init_state.detach().cpu().numpy()
        att = vehicle_atts.detach().cpu().numpy()
        mask = batch_mask.detach().cpu().numpy()
        rows, counts = [], []
        for b in range(self.B):
            idx = np.nonzero(mask == b)[0]
            counts.append(len(idx))
            for i in idx:                                       # state_conv (:80-98): heading angle in the state's precision
                x, y, hc, hs, s, _ = state[i]
                rows.append([float(x), float(y), float(np.arctan2(hs, hc)), float(s), float(att[i, 0]), float(att[i, 1])])
        if min(counts) < 1 or not 0 <= self.ego_idx < min(counts):
            raise ValueError('every scene needs an ego at position %d' % self.ego_idx)
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.batch_maps = [self.map_env.map_list[int(map_idx[b])] for b in range(self.B)]
        names = sorted(set(self.batch_maps), key=self.batch_maps.index)
        if len(names) > 4:
            raise NotImplementedError('at most 4 maps per batch (nuScenes has 4)')
        graphs = []
        for name in names:
            key = (name, float(self.cfg.xydistmax), str(dev))
            g = self._graphs.get(key)
            if g is None:
                g = PackedLaneGraph(self.lane_graphs[name], self.cfg.xydistmax, dev)
                self._graphs[key] = g
            graphs.append(g)
        self._world = dict(
            dev=dev, counts=counts, ptr_np=ptr, graphs=graphs,
            ptr=torch.from_numpy(ptr).to(dev),
            scene_map=torch.tensor([names.index(n) for n in self.batch_maps], dtype=torch.int32, device=dev),
            init=torch.tensor(rows, dtype=torch.float64, device=dev).reshape(-1, 6).contiguous())
        self._rows = {}

    def _row_maps(self, agent_ptr):
        key = tuple(int(v) for v in agent_ptr)
        ent = self._rows.get(key)
        if ent is None:
            w = self._world
            obj, scene = [], []
            for b in range(self.B):
                n = w['counts'][b]
                if key[b + 1] - key[b] != n - 1 or key[b] != len(obj):
                    raise ValueError('agent_ptr does not describe the non-ego agents of the reset() batch')
                for i in range(n):
                    if i != self.ego_idx:
                        obj.append(int(w['ptr_np'][b]) + i)
                        scene.append(b)
            dev = w['dev']
            ent = (torch.tensor(obj + [0], dtype=torch.int32, device=dev), torch.tensor(scene + [0], dtype=torch.int32, device=dev), len(obj))
            self._rows = {key: ent}
        return ent

    def _table(self, arr):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float64))
        key = a.tobytes()
        t = self._tables.get(key)
        if t is None:
            if len(self._tables) > 16:
                self._tables.clear()
            t = torch.from_numpy(a.copy()).to(self._world['dev'])
            self._tables[key] = t
        return t

    def _descriptor(self, row_obj, row_scene, NR):
        w = self._world
        p = L.StrivePlanner()
        p.cfg = planner_cfg_struct(self.cfg)
        p.nmaps = len(w['graphs'])
        for i, g in enumerate(w['graphs']):
            g.fill(p.maps[i])
        p.B, p.NO, p.NR, p.ego_idx = self.B, int(w['ptr_np'][-1]), NR, self.ego_idx
        p.ptr, p.scene_map, p.init = w['ptr'].data_ptr(), w['scene_map'].data_ptr(), w['init'].data_ptr()
        p.row_obj, p.row_scene = row_obj.data_ptr(), row_scene.data_ptr()
        return p

    def prepare(self, agent_t, agent_ptr, planner_t):
        """Build (on the current stream) the small device tables a rollout with these times / offsets will use, so that
        rollouts issued later from side streams only read them."""
        planner_t = np.asarray(planner_t, dtype=np.float64)
        nstep = int(planner_t[-1] / self.cfg.dt)
        self._table(np.asarray(agent_t, dtype=np.float64))
        self._table(planner_t)
        self._table(np.linspace(self.cfg.dt, self.cfg.dt * nstep, nstep + 1))
        self._row_maps(np.asarray(agent_ptr).reshape(-1))

    # ---- deferred status check: no host synchronisation inside an optimisation closure ----------------------------------
    def _read_status(self, wait):
        """Host copy (B, 8) of the flags or None: ``wait=True`` reads them back (one synchronisation); ``wait=False`` never blocks --
        it returns a snapshot whose asynchronous copy has already arrived (if any) and starts the next one."""
        st = self._status
        if st is None:
            return None
        if st.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return None       # inside a HIP-graph capture (utils/graphed.py): the flags accumulate over the replays, the loop checks at its end
        if st.device.type != 'cuda':
            return st
        if wait:
            self._pending = None
            return st.cpu()                   # (ordered behind every rollout enqueued on, or joined into, the current stream)
        host = None
        pend = self._pending
        if pend is not None and pend[1].query():
            host, self._pending = pend[0], None
        if self._pending is None:
            snap = torch.empty(tuple(st.shape), dtype=torch.int32).pin_memory()
            snap.copy_(st, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(st.device))
            self._pending = (snap, ev)
        return host

    def failed_scenes(self, wait=True):
        """{scene: [limit names]} of the scenes whose plan is NaN because SOME rollout since reset() hit a capacity / range limit in
        them (the cases in which the reference's planner raises).  The flags are per scene and sticky; the other scenes' plans are
        unaffected.  Does not clear anything."""
        host = self._read_status(wait)
        if host is None:
            return {}
        host = host.reshape(-1, len(STATUS_NAMES))
        return {b: [STATUS_NAMES[i] for i in range(len(STATUS_NAMES)) if int(host[b, i]) != 0]
                for b in range(host.shape[0]) if bool((host[b] != 0).any())}

    def check(self, wait=True, on_error=None):
        """What became of the rollouts since reset(): ``on_error='raise'`` (default, ``self.on_error``) raises StriveHipError naming
        the failed scenes and limits and clears the flags -- a failure costs the caller the whole batch, which is what a raise
        inside the reference's per-scene numpy loop costs it (:178-276 called from adv_gen_optim.py:133-139); ``'report'`` returns
        ``failed_scenes()`` and leaves the flags set, for callers that quarantine the scenes (run_adv_gen_optim).
        The device kernels only set flags in one persistent tensor, so nothing is lost between rollouts: ``wait=True`` reads it back
        (one synchronisation; the loops call it once, after their last iteration); ``wait=False`` never blocks -- which is how a
        failure surfaces early inside a synchronisation-free optimisation loop."""
        bad = self.failed_scenes(wait)
        mode = self.on_error if on_error is None else on_error
        if mode not in ('raise', 'report'):
            raise ValueError("on_error must be 'raise' or 'report'")
        if bad and mode == 'raise':
            self._status.zero_()
            self.alive.fill_(1)
            self._pending = None
            names = sorted(set(n for v in bad.values() for n in v))
            raise L.StriveHipError('HardcodeNuscPlanner.rollout exceeded a limit of the device planner in scene(s) %s: %s' % (
                ', '.join(str(b) for b in sorted(bad)), '; '.join(names)))
        return bad

    # ---- rollout (reference :178-276) -------------------------------------------------------------------------------------
    def rollout(self, agent_obs, agent_t, agent_ptr, planner_t, init_state=None, control_all=False, viz=None, coll_t=None,
                on_error=None):
        """agent_obs (NA-B, T, 4) UNNORMALISED futures of the non-ego agents (device tensor or numpy array), agent_t (T)
        their times, agent_ptr (B+1) scene offsets into agent_obs, planner_t (T') times at which the planner pose is returned
        -> float64 tensor (B, T', 4) of (x, y, cos h, sin h): on the device for a device ``agent_obs``, on the host for numpy."""
        if self._world is None or self.B is None:
            raise RuntimeError('HardcodeNuscPlanner.rollout: call reset() first')
        if init_state is not None or control_all or agent_obs is None:
            raise NotImplementedError('only the closed-loop attack mode of adv_gen_optim is implemented (observed other agents)')
        if viz is not None:
            raise NotImplementedError('planner visualisation (viz=...) is outside the hot path; render the returned plan instead')
        self.check(wait=False, on_error=on_error)     # (``on_error``: this call's answer to failed scenes; None = self.on_error)
        w = self._world
        dev = w['dev']
        from_numpy = not torch.is_tensor(agent_obs)
        obs = torch.from_numpy(np.ascontiguousarray(agent_obs)) if from_numpy else agent_obs.detach()
        obs = obs.to(dev).to(torch.float64).contiguous()     # widened exactly, as numpy does when stacking onto the float64 world state
        agent_t = np.asarray(agent_t, dtype=np.float64)
        planner_t = np.asarray(planner_t, dtype=np.float64)
        assert obs.dim() == 3 and obs.shape[1] == agent_t.shape[0] and obs.shape[2] == 4
        row_obj, row_scene, NR = self._row_maps(np.asarray(agent_ptr).reshape(-1))
        assert obs.shape[0] == NR
        cfg = self.cfg
        nstep = int(planner_t[-1] / cfg.dt)
        t_out = np.linspace(cfg.dt, cfg.dt * nstep, nstep + 1)      # (sic: the reference labels its nstep+1 poses like this)
        lib = ops._lib_for(obs)
        desc = self._descriptor(row_obj, row_scene, NR)
        nbytes = lib.query('strive_planner_workspace_bytes', C.byref(desc), nstep, int(self.traj_cap))
        if nbytes == 0:
            raise L.StriveHipError('strive_planner_workspace_bytes: ' + lib.query('strive_last_error').decode())
        ws = ops._workspace(dev, nbytes, tag='planner')
        self._ws, self._ws_prof_offset = ws, nbytes - 192        # (tools/planner_bench.py: the option planner_prof's counters)
        if os.environ.get('STRIVE_POISON_WS') == '1':
            ws.fill_(0x25)                # debug aid (the tests set it): a kernel that reads workspace it did not write sees garbage
        TP = planner_t.shape[0]
        plan = torch.empty((self.B, TP, 4), dtype=torch.float64, device=dev)
        status, alive = self._status, self.alive
        at, to, pt = self._table(agent_t), self._table(t_out), self._table(planner_t)
        if NR == 0:
            obs = torch.zeros((1, max(1, agent_t.shape[0]), 4), dtype=torch.float64, device=dev)
        lib.call('strive_planner_rollout', C.byref(desc), L.ptr(obs), L.ptr(at), int(agent_t.shape[0]), L.ptr(to), nstep, L.ptr(pt),
                 TP, int(self.traj_cap), L.ptr(plan), L.ptr(status), L.ptr(alive), L.ptr(ws), ws.numel(), L.stream_ptr(obs))
        defer = (dev.type == 'cuda') if self.defer_check is None else bool(self.defer_check)
        if from_numpy or not defer:
            self.check(wait=True)
        return plan.cpu() if from_numpy else plan
