"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's rule-based lane-following planner
(reference src/planners/hardcode_goalcond_nusc.py), the planner ``adv_gen_rule_based.cfg`` attacks in closed loop
(src/utils/adv_gen_optim.py:133-139: one ``rollout`` per optimisation iteration).  Host-side numpy, scene by scene and
planner step by planner step like the reference, restated around three small building blocks:

  * ``LinearPath``      piecewise-linear map s -> R^d with strict bounds (what the reference gets from scipy's interp1d);
  * ``LaneGraph``       the per-map graph (node positions, successor / predecessor lists, directed edge table);
  * ``VehicleState``    one world object; the world is a dict id -> VehicleState.

Pinned by fixture g10 (tests/golden/make_golden.py::g10_planner: the reference's own ``HardcodeNuscPlanner.rollout`` for both
shipped configurations, equal to 1e-9) and, in closed loop, by g6h.  The product planner
(strive_amd/planners/hardcode_goalcond_nusc.py -> strive_amd/csrc/planner.hip) is checked against this file; nothing in
strive_amd/ imports it.

Per planner step (dt = 0.2 s): every object is matched to the lane edges it could be following (heading within
``cdistang`` degrees, lateral distance below ``xydistmax``), matches that are connected through other matches are
clustered and the closest one of each cluster kept (:298-347); from each kept edge all routes at least as long as the
object could travel are enumerated breadth-first, successor lists in order (:379-414), resampled every 0.4 m, blended so
that they pass through the object's own position (:487-556) and turned into an arc-length path.  The ego then follows ITS
FIRST route: 25 two-phase speed profiles (:804-826) are scored by the probability of coming close to any predicted
trajectory of any other object (5-circle box distance, :860-897; tanh score growing with time, :724-728), the
fastest profile below ``col_plim`` wins (the safest if none) and its first speed is applied for one step (:829-857).
"""
from collections import deque

import numpy as np
import torch



class PlannerConfig(object):
    """(reference src/planners/planner.py)"""
    def __init__(self, **kwargs):
        for key, val in kwargs.items():
            setattr(self, key, val)


class PlannerNusc(object):
    def __init__(self, map_env, cfg):
        self.map_env = map_env
        self.cfg = cfg

DEF_CONFIG = {
    'dt': 0.2, 'preddt': 0.2, 'nsteps': 25, 'cdistang': 20.0, 'xydistmax': 2.0, 'smax': 15.0, 'accmax': 3.0,
    'predsfacs': [0.5, 1.0], 'predafacs': [0.5], 'interacdist': 70.0, 'planaccfacs': [1.0], 'plannspeeds': 5,
    'col_plim': 0.1, 'score_wmin': 0.7, 'score_wfac': 0.05,
}
# tuned on generated validation scenarios (reference :43-59)
TUNED_VAL_FINAL_1 = dict(DEF_CONFIG, smax=20.0, accmax=4.0, score_wmin=0.3, score_wfac=0.02)
CONFIG_DICT = {'default': DEF_CONFIG, 'final_tuned_val_1': TUNED_VAL_FINAL_1}

LANE_DS, LANE_SIG, SBUFFER = 0.4, 3.5, 4.0      # constants of rollout() in the reference (:211-213)


class LinearPath(object):
    """y(t) by linear interpolation between knots, error outside [t0, tN] -- scipy.interpolate.interp1d(kind='linear',
    bounds_error=True, assume_sorted=True) evaluated with the same arithmetic (slope * (t - t_lo) + y_lo)."""

    def __init__(self, t, y):
        self.t = np.asarray(t, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        if np.any(q < self.t[0]) or np.any(q > self.t[-1]):
            raise ValueError('LinearPath: query outside [%g, %g]' % (self.t[0], self.t[-1]))
        hi = np.clip(np.searchsorted(self.t, q), 1, len(self.t) - 1)
        lo = hi - 1
        tail = (1,) * (self.y.ndim - 1)
        slope = (self.y[hi] - self.y[lo]) / (self.t[hi] - self.t[lo]).reshape(q.shape + tail)
        return slope * (q - self.t[lo]).reshape(q.shape + tail) + self.y[lo]


class VehicleState(object):
    __slots__ = ('x', 'y', 'h', 's', 'l', 'w', 'match_edges', 'match_points', 'routes', 'control')

    def __init__(self, x, y, h, s, l, w):
        self.x, self.y, self.h, self.s, self.l, self.w = float(x), float(y), float(h), float(s), float(l), float(w)
        self.match_edges = self.match_points = self.routes = self.control = None


class LaneGraph(object):
    def __init__(self, lg):
        self.xy = np.asarray(lg['xy'], dtype=np.float64)
        self.succ, self.pred = lg['out_edges'], lg['in_edges']
        e = np.asarray(lg['edges'], dtype=np.float64)
        self.e_xy, self.e_dir, self.e_len = e[:, 0:2], e[:, 2:4], e[:, 4]
        self.e_ix = np.asarray(lg['edgeixes'], dtype=np.int64)

    # -- which edges could this pose be on?  (reference get_lane_matches / edge_closest_point, :298-359)
    def match(self, x, y, h, cdistmax, xydistmax):
        cdist = 1.0 - self.e_dir[:, 0] * np.cos(h) - self.e_dir[:, 1] * np.sin(h)
        keep = np.nonzero(cdist < cdistmax)[0]
        if keep.size == 0:
            return np.empty((0, 2), dtype=np.int64), np.empty((0, 2))
        pts, dist = closest_on_segments(self.e_xy[keep], self.e_dir[keep], self.e_len[keep], np.array([x, y]))
        near = dist < xydistmax
        return self.e_ix[keep][near], pts[near]

    # -- one representative per group of matches connected through matches  (cluster_matches_combine / cluster_bfs, :324-376)
    def cluster(self, x, y, edges, points):
        if len(points) == 0:
            return edges, points
        order = np.argsort(np.linalg.norm(np.array([[x, y]]) - points, axis=1))
        done = {(int(a), int(b)): False for a, b in edges}
        kept_e, kept_p = [], []
        for k in order:
            key = (int(edges[k, 0]), int(edges[k, 1]))
            if done[key]:
                continue
            kept_e.append(list(key))
            kept_p.append(points[k])
            for forward in (True, False):
                todo = deque([key])
                while todo:
                    a, b = todo.popleft()
                    done[(a, b)] = True
                    if forward:
                        nxt = [(b, c) for c in self.succ[b]]
                    else:
                        nxt = [(c, a) for c in self.pred[a]]
                    for cand in nxt:
                        if cand in done and not done[cand]:
                            todo.append(cand)
        return np.array(kept_e), np.array(kept_p)

    # -- all node chains from v of length > mindist (or ending at a terminal node), breadth first with the successor
    #    lists in order: the first chain always takes the first connection  (expand_verts, :379-414)
    def chains(self, v, table, mindist):
        todo = deque([([v], 0.0)])
        out = []
        while todo:
            verts, length = todo.popleft()
            verts = list(verts)
            while length <= mindist:
                cur = verts[-1]
                conn = table[cur]
                if len(conn) == 0:
                    break
                for other in conn[1:]:
                    todo.append((verts + [other], length + float(np.linalg.norm(self.xy[other] - self.xy[cur]))))
                first = conn[0]
                length = length + float(np.linalg.norm(self.xy[first] - self.xy[cur]))
                verts.append(first)
            out.append((verts, length))
        return out


def closest_on_segments(p0, direction, length, query):
    along = (query[None, 0] - p0[:, 0]) * direction[:, 0] + (query[None, 1] - p0[:, 1]) * direction[:, 1]
    along = np.minimum(np.maximum(along, 0.0), length)
    pts = p0 + along[:, None] * direction
    return pts, np.linalg.norm(query[None, :] - pts, axis=1)


def straight_route(xy, h, back, fwd):
    """no lane nearby: keep the heading  (constant_heading_spline, :477-484)"""
    c, s = np.cos(h), np.sin(h)
    return LinearPath(np.array([-back, fwd]), np.array([[xy[0] - back * c, xy[1] - back * s, c, s],
                                                         [xy[0] + fwd * c, xy[1] + fwd * s, c, s]]))


def routes_through(graph, edges, points, back, fwd, xydistmax, xy, h):
    """Arc-length paths (x, y, cos, sin)(s), s = 0 at the object, one per (kept match, forward chain, backward chain)
    (get_prediction_splines / local_lane_closest / xy2spline, :433-556)."""
    if len(edges) == 0:
        return [straight_route(xy, h, back, fwd)]
    out = []
    need_f, need_b = fwd + SBUFFER + xydistmax, back + SBUFFER + xydistmax
    nb, nf = int((back + SBUFFER) / LANE_DS) + 1, int((fwd + SBUFFER) / LANE_DS) + 1
    s_eval = np.concatenate((np.linspace(-back - SBUFFER, 0.0, nb + 1)[:-1], np.linspace(0.0, fwd + SBUFFER, nf)), 0)
    for (v0, v1), _ in zip(edges, points):
        fchains = graph.chains(int(v1), graph.succ, need_f)
        bchains = graph.chains(int(v0), graph.pred, need_b)
        for fverts, flen in fchains:
            for bverts, blen in bchains:
                pts = np.concatenate((graph.xy[bverts[::-1]], graph.xy[fverts]), axis=0)
                i0 = len(bverts) - 1
                if flen <= need_f:                                   # dead end ahead: extend straight
                    d = pts[-1] - pts[-2]
                    d = d / np.linalg.norm(d)
                    pts = np.concatenate((pts, (pts[-1] + d * (1.0 + need_f - flen))[None]), axis=0)
                if blen <= need_b:
                    d = pts[0] - pts[1]
                    d = d / np.linalg.norm(d)
                    pts = np.concatenate(((pts[0] + d * (1.0 + need_b - blen))[None], pts), axis=0)
                    i0 += 1
                # the locally closest point of the chain to the object, walking downhill from the matched edge
                seg = pts[1:] - pts[:-1]
                seglen = np.linalg.norm(seg, axis=1)
                cp, cd = closest_on_segments(pts[:-1], seg / seglen[:, None], seglen, xy)
                k = i0
                while k - 1 >= 0 and cd[k - 1] < cd[k]:
                    k -= 1
                while k + 1 < len(cd) and cd[k + 1] < cd[k]:
                    k += 1
                anchor = cp[k]
                s_nodes = np.zeros(len(pts))
                s_nodes[1:] = np.cumsum(seglen)
                s_nodes = s_nodes - s_nodes[k] - np.linalg.norm(anchor - pts[k])
                lane = LinearPath(s_nodes, pts)(s_eval)
                # blend from the lane to the object's own position around s = 0
                lane = lane + (xy - anchor)[None, :] * np.exp(-np.square(s_eval) / LANE_SIG ** 2)[:, None]
                d = lane[1:] - lane[:-1]
                dl = np.linalg.norm(d, axis=1)
                head = d / dl[:, None]
                head = np.concatenate((head, head[[-1]]), 0)
                knots = np.concatenate((lane, head), 1)
                knots[nb, 2], knots[nb, 3] = np.cos(h), np.sin(h)      # pass through the object's heading exactly
                s = np.zeros(len(lane))
                s[1:] = np.cumsum(dl)
                s -= s[nb]
                if not (s[0] < -back and s[-1] > fwd):
                    raise AssertionError('route does not cover [%g, %g]: [%g, %g]' % (-back, fwd, s[0], s[-1]))
                out.append(LinearPath(s, knots))
    return out


def signed_speed(x0, y0, x1, y1, h1, dt):
    mag = np.sqrt((x1 - x0) ** 2 + (y1 - y0) ** 2) / dt
    return mag if (x1 - x0) * np.cos(h1) + (y1 - y0) * np.sin(h1) >= 0 else -mag


def speed_ramp(s, target, acc, n, dt):
    """n+1 speeds starting at s, moving towards `target` by at most acc*dt per step  (compute_speed_profile, :670-683)"""
    k = np.arange(n + 1)
    if target > s:
        return np.minimum(s + k * acc * dt, target)
    if target < s:
        return np.maximum(s - k * acc * dt, target)
    return s + np.zeros(n + 1)


def travelled(speeds, dt):
    d = np.zeros(len(speeds))
    d[1:] = np.cumsum(speeds[1:] * dt)
    return d


def candidate_profiles(s0, dt, nsteps, accfacs, accmax, smax, nspeeds):
    """two-phase speed profiles: reach one of `nspeeds` speeds after half the horizon, then one of `nspeeds` again
    (gen_sprofiles, :804-826)"""
    n1 = nsteps // 2
    n2 = nsteps - n1
    out = []
    for fac in accfacs:
        acc = fac * accmax
        for s1 in np.linspace(max(0.0, s0 - n1 * dt * acc), min(smax, s0 + n1 * dt * acc), nspeeds):
            first = speed_ramp(s0, s1, acc, n1, dt)
            for s2 in np.linspace(max(0.0, first[-1] - n2 * dt * acc), min(smax, first[-1] + n2 * dt * acc), nspeeds):
                prof = np.concatenate((first, speed_ramp(first[-1], s2, acc, n2, dt)[1:]))
                out.append({'speeds': prof, 'dist': travelled(prof, dt), 'acc': acc, 's1': s1, 's2': s2})
    return out


def box_circles(b):
    """(T, N, 5) boxes (x, y, h, l, w) -> (T, N, 5, 3) circles (x, y, r): four of radius w/4 towards the corners and one of
    radius w/2 in the middle  (boxes2circles, :860-882)"""
    xy, hh, li, wi = b[:, :, 0:2], b[:, :, 2], b[:, :, 3], b[:, :, 4]
    L, W = np.maximum(li, wi), np.minimum(li, wi)
    H = np.where(li < wi, hh + np.pi / 2.0, hh)
    a0 = ((L - W) / 2 + W / 4)[:, :, None] * np.stack((np.cos(H), np.sin(H)), 2)
    a1 = (W / 4)[:, :, None] * np.stack((-np.sin(H), np.cos(H)), 2)
    c = np.empty(b.shape[:2] + (5, 3))
    c[:, :, 0, :2] = xy + a0 + a1
    c[:, :, 1, :2] = xy - a0 + a1
    c[:, :, 2, :2] = xy - a0 - a1
    c[:, :, 3, :2] = xy + a0 - a1
    c[:, :, 4, :2] = xy
    c[:, :, 4, 2] = W / 2
    c[:, :, :4, 2] = W[:, :, None] / 4
    return c


def box_gap(b0, b1):
    """(T, N0, 5), (T, N1, 5) -> (T, N0): smallest circle-to-circle gap to any box of b1  (approx_bbox_distance, :885-897)"""
    T, N0, _ = b0.shape
    N1 = b1.shape[1]
    c0 = box_circles(b0).reshape((T, N0, 5, 1, 1, 3))
    c1 = box_circles(b1).reshape((T, 1, 1, N1, 5, 3))
    gap = np.linalg.norm(c1[..., 0:2] - c0[..., 0:2], axis=5) - c0[..., 2] - c1[..., 2]
    return np.amin(gap, axis=(2, 3, 4))


class HardcodeNuscPlanner(PlannerNusc):
    def __init__(self, map_env, cfg):
        super(HardcodeNuscPlanner, self).__init__(map_env, cfg)
        self.lane_graphs = self.map_env.lane_graphs
        self._graphs = {}
        self.init_world = None
        self.batch_mask = self.B = self.batch_maps = None
        self.ego_idx = 0

    # ---- world construction --------------------------------------------------------------------
    @staticmethod
    def _name(i):
        return '%04d' % i

    def _world_of(self, state, att):
        """(n,6) unnormalised (x,y,hx,hy,s,hdot) + (n,2) (l,w) of ONE scene -> {id: VehicleState}  (state_conv, :80-98)"""
        state = state.detach().cpu().numpy()
        att = att.detach().cpu().numpy()
        world = {}
        for i in range(state.shape[0]):
            x, y, hc, hs, s, _ = state[i]
            world['ego' if i == self.ego_idx else self._name(i)] = VehicleState(x, y, np.arctan2(hs, hc), s, att[i, 0], att[i, 1])
        return world

    def reset(self, init_state, vehicle_atts, batch_mask, batch_size, map_idx, ego_idx=0):
        """(reference :109-127)"""
        self.ego_idx = ego_idx
        self.B = batch_size
        self.batch_mask = batch_mask
        self.init_world = [self._world_of(init_state[batch_mask == b], vehicle_atts[batch_mask == b]) for b in range(batch_size)]
        self.batch_maps = [self.map_env.map_list[int(map_idx[b])] for b in range(batch_size)]

    def _graph(self, name):
        g = self._graphs.get(name)
        if g is None:
            g = LaneGraph(self.lane_graphs[name])
            self._graphs[name] = g
        return g

    def _observations(self, world, obs, obs_t):
        """per non-ego object: its observed future as a path over time, up to the first NaN frame  (create_other_agents, :140-176)"""
        paths = {}
        for k in range(obs.shape[0]):
            oid = self._name(k + 1 if k >= self.ego_idx else k)
            o = world[oid]
            states = np.concatenate([np.array([[o.x, o.y, np.cos(o.h), np.sin(o.h)]]), obs[k]], axis=0)
            bad = np.nonzero(np.isnan(states.sum(axis=1)))[0]
            n = states.shape[0] if bad.size == 0 else int(bad[0])
            if n == 1:
                paths[oid] = (0.0, 0.0, None)
                continue
            t = np.append(np.array([0.0]), obs_t[:n - 1])
            paths[oid] = (0.0, float(t[-1]), LinearPath(t, states[:n]))
        return paths

    # ---- one planner step ------------------------------------------------------------------------
    def _plan_routes(self, world, graph):
        cfg = self.cfg
        tmax = cfg.nsteps * cfg.preddt
        cdistmax = 1.0 - np.cos(np.radians(cfg.cdistang))
        for o in world.values():
            e, p = graph.match(o.x, o.y, o.h, cdistmax, cfg.xydistmax)
            o.match_edges, o.match_points = graph.cluster(o.x, o.y, e, p)
            back = 1.0 if o.s > 0 else 1.0 + abs(o.s) * tmax
            fwd = 1.0 + cfg.smax * tmax if o.s < 0 else max(1.0 + cfg.smax * tmax, 1.0 + o.s * tmax)
            o.routes = routes_through(graph, o.match_edges, o.match_points, back, fwd, cfg.xydistmax, np.array([o.x, o.y]), o.h)

    def _predict_others(self, world, ego):
        cfg = self.cfg
        trajs = []
        for oid, o in world.items():
            if o is ego or np.sqrt((ego.x - o.x) ** 2 + (ego.y - o.y) ** 2) > cfg.interacdist:
                continue
            dists = [travelled(speed_ramp(o.s, o.s * sf, cfg.accmax * af, cfg.nsteps, cfg.preddt), cfg.preddt)
                     for sf in cfg.predsfacs for af in cfg.predafacs]
            for route in o.routes:
                for d in dists:
                    q = route(d)
                    tr = np.empty((cfg.nsteps + 1, 5))
                    tr[:, :2] = q[:, :2]
                    tr[:, 2] = np.arctan2(q[:, 3], q[:, 2])
                    tr[:, 3], tr[:, 4] = o.l, o.w
                    trajs.append(tr)
        if not trajs:
            return np.empty((cfg.nsteps + 1, 0, 5))
        return np.transpose(np.array(trajs), (1, 0, 2))

    def _choose_profile(self, ego, profiles, others, prefer_stop):
        """(plot_plan_info, :768-801 and score_dists, :724-728)"""
        cfg = self.cfg
        if others.shape[1] == 0:
            return profiles[int(np.argmax([p['dist'][-1] for p in profiles]))]
        route = ego.routes[0]
        box = np.empty((cfg.nsteps + 1, 1, 5))
        box[:, :, 3], box[:, :, 4] = ego.l, ego.w
        w = cfg.score_wmin + np.arange(cfg.nsteps + 1) * cfg.score_wfac
        risk = []
        for p in profiles:
            q = route(p['dist'])
            box[:, 0, :2] = q[:, :2]
            box[:, 0, 2] = np.arctan2(q[:, 3], q[:, 2])
            gap = box_gap(box, others)[:, 0]
            pr = 1.0 + np.tanh(-gap * w)
            pr[gap < 0] = 1.0
            risk.append(1.0 - np.prod(1.0 - pr))
        ok = [i for i in range(len(profiles)) if risk[i] < cfg.col_plim]
        if not ok:
            return profiles[int(np.argmin(risk))]
        reach = [profiles[i]['dist'][-1] for i in ok]
        return profiles[ok[int(np.argmin(reach) if prefer_stop else np.argmax(reach))]]

    def _act(self, world):
        """(compute_action, :829-857 and postprocess_act_for_speed, :642-666)"""
        cfg = self.cfg
        ego = world['ego']
        profiles = candidate_profiles(ego.s, cfg.preddt, cfg.nsteps, cfg.planaccfacs, cfg.accmax, cfg.smax, cfg.plannspeeds)
        others = self._predict_others(world, ego)
        best = self._choose_profile(ego, profiles, others, prefer_stop=len(ego.match_points) == 0)
        s_next = speed_ramp(ego.s, best['s1'], best['acc'], 1, cfg.dt)[1]
        nx, ny, nc, ns = ego.routes[0](cfg.dt * s_next)
        nh = np.arctan2(ns, nc)
        # place the new pose so that the step's signed speed is exactly s_next
        if np.sign(signed_speed(ego.x, ego.y, nx, ny, nh, cfg.dt)) != np.sign(s_next):
            px, py, ph = ego.x + np.cos(ego.h) * s_next * cfg.dt, ego.y + np.sin(ego.h) * s_next * cfg.dt, ego.h
        else:
            d = np.array([nx - ego.x, ny - ego.y])
            dn = np.linalg.norm(d)
            if dn == 0.0:
                assert s_next == 0.0
                px, py, ph = ego.x + np.cos(ego.h) * s_next * cfg.dt, ego.y + np.sin(ego.h) * s_next * cfg.dt, ego.h
            else:
                d = d / dn
                px, py, ph = ego.x + d[0] * abs(s_next) * cfg.dt, ego.y + d[1] * abs(s_next) * cfg.dt, nh
        assert abs(signed_speed(ego.x, ego.y, px, py, ph, cfg.dt) - s_next) < 1e-6
        ego.control = (px, py, ph)

    def _advance(self, world, t0, paths):
        """(update_wstate, :601-621): the ego moves to its control, the others to their observed pose; objects without an
        observation at the new time leave the world"""
        dt = self.cfg.dt
        t1 = t0 + dt
        new = {}
        for oid, o in world.items():
            if o.control is not None:
                x, y, h = o.control
                new[oid] = VehicleState(x, y, h, signed_speed(o.x, o.y, x, y, h, dt), o.l, o.w)
            else:
                ta, tb, path = paths[oid]
                if ta <= t1 <= tb:
                    x, y, hc, hs = path(t1)
                    h = np.arctan2(hs, hc)
                    new[oid] = VehicleState(x, y, h, signed_speed(o.x, o.y, x, y, h, dt), o.l, o.w)
        return new, t1

    # ---- rollout -------------------------------------------------------------------------------------
    def rollout(self, agent_obs, agent_t, agent_ptr, planner_t, init_state=None, control_all=False, viz=None, coll_t=None):
        """(reference :178-276)  agent_obs (NA-B, T, 4) unnormalised futures of the non-ego agents, agent_t (T) their times,
        agent_ptr (B+1) scene offsets into agent_obs, planner_t (T') times at which the planner pose is returned.
        -> float64 tensor (B, T', 4) of (x, y, cos h, sin h)."""
        if self.init_world is None or self.B is None:
            raise RuntimeError('HardcodeNuscPlanner.rollout: call reset() first')
        if init_state is not None or control_all or agent_obs is None:
            raise NotImplementedError('only the closed-loop attack mode of adv_gen_optim is implemented (observed other agents)')
        cfg = self.cfg
        assert agent_obs.shape[1] == agent_t.shape[0]
        nstep = int(planner_t[-1] / cfg.dt)
        t_out = np.linspace(cfg.dt, cfg.dt * nstep, nstep + 1)      # (sic: the reference labels its nstep+1 poses like this)
        result = []
        for b in range(self.B):
            world = {k: VehicleState(o.x, o.y, o.h, o.s, o.l, o.w) for k, o in self.init_world[b].items()}
            obs = agent_obs[agent_ptr[b]:agent_ptr[b + 1]]
            assert obs.shape[0] == len(world) - 1
            paths = self._observations(world, obs, agent_t)
            graph = self._graph(self.batch_maps[b])
            t = 0.0
            poses = []
            for k in range(nstep + 1):
                if k > 0:
                    world, t = self._advance(world, t, paths)
                self._plan_routes(world, graph)
                self._act(world)
                x, y, h = world['ego'].control
                poses.append([x, y, np.cos(h), np.sin(h)])
            result.append(np.array(poses))
        plan = LinearPath(t_out, np.stack(result, axis=1))(np.asarray(planner_t, dtype=np.float64))      # (T', B, 4)
        return torch.from_numpy(np.ascontiguousarray(np.transpose(plan, (1, 0, 2))))
