// Rule-based lane-following planner on the device (reference src/planners/hardcode_goalcond_nusc.py).
//
// The reference rolls the planner out scene by scene in numpy, once per optimisation iteration of the closed-loop attack
// (src/utils/adv_gen_optim.py:133-139): 31 planner steps per scene, every step re-derives the lane routes of EVERY object and
// scores 25 ego speed profiles against every predicted trajectory.  Only the ego's pose chains the steps together; the other
// objects replay their observed futures.  So here
//
//   planner_world_kernel    one thread per non-ego object: its pose / signed speed / presence at every planner step
//                           (create_other_agents :140-176, update_wstate :601-621);
//   planner_routes_kernel   one wave per (non-ego object, planner step), all steps at once: lane matches through a uniform
//                           grid over the edges (:298-322), clustering (:324-376), breadth-first chains (:379-414), blended
//                           arc-length routes (:433-556), constant-heading fallback (:477-484), predicted trajectories
//                           (:686-721) appended to the (scene, step) list;
//   per planner step k:
//     planner_ego_kernel    one wave per scene: risk of step k-1's profiles from the partial gap minima, profile choice and
//                           action (:768-857, :642-666), world update of the ego, its first route, the candidate speed
//                           profiles (:804-826) and their 5-circle boxes (:860-882);
//     planner_gap_kernel    (scene, trajectory chunk): 5-circle gaps (:885-897) of all profiles x times against the chunk's
//                           trajectories -- one square root per radius class (the minimum commutes with the monotone
//                           square root and subtractions);
//   planner_interp_kernel   the planner poses interpolated to the requested times (:270-272).
//
// float64 throughout, operations in numpy's order and without fused multiply-adds, so that the discrete decisions (matches,
// arg-mins, the profile choice) are the reference's; serial recurrences (cumulative sums, chain walks) run on lane 0.
#include "common.h"
#include <math.h>
#include <stdlib.h>

#pragma clang fp contract(off)

namespace strive_planner {

constexpr int MAXM = 96;        // lane matches of one pose
constexpr int MAXKEEP = 16;     // clusters of matches
constexpr int MAXCHF = 48;      // forward chains of one match
constexpr int MAXCHB = 16;      // backward chains
constexpr int POOLF = 640;
constexpr int POOLB = 160;
constexpr int MAXP = 320;       // nodes of one backward + forward chain (+ 2 extensions)
constexpr int MAXK = 384;       // resampled knots of a route
static_assert(3 * MAXK <= 4 * MAXP, "RouteLds: the knots' arc lengths and headings reuse the node arrays");
constexpr int MAXNT = 32;       // nsteps + 1
constexpr int MAXPRED = 8;      // predicted speed profiles per route
constexpr int MAXPROF = 64;     // ego speed profiles
constexpr int PBLK = 25;        // profiles whose running minima a gap thread keeps in registers
constexpr int NCHUNK = 32;      // most trajectory chunks per scene in the gap kernel (the launch chooses Work::nchunk <= this)

constexpr double LANE_DS = 0.4, LANE_SIG2 = 3.5 * 3.5, SBUFFER = 4.0;   // constants of rollout() (:211-213)

enum { ST_MATCH = 0, ST_KEEP = 1, ST_CHAIN = 2, ST_NODES = 3, ST_KNOTS = 4, ST_RANGE = 5, ST_TRAJ = 6, ST_ACT = 7 };
constexpr int NSTATUS = STRIVE_PLANNER_NSTATUS;

template <int NCH, int NPOOL>
struct ChainTab {
    double len[NCH];
    int first[NCH];
    short parent[NCH], fork[NCH], nown[NCH], ownoff[NCH];
    int pool[NPOOL];
    int n, npool;
};

struct RouteLds {
    double m_px[MAXM], m_py[MAXM], m_d[MAXM];
    int m_v0[MAXM], m_v1[MAXM];
    short m_order[MAXM];
    unsigned char m_done[MAXM];
    short kept[MAXKEEP];
    short queue[MAXM];
    ChainTab<MAXCHF, POOLF> cf;
    ChainTab<MAXCHB, POOLB> cb;
    // One route's working set.  The node arrays (positions, distances to the pose, arc lengths) are dead once the knots are
    // resampled from them, and the knots' headings and arc lengths are only written after that (assemble_route: a barrier lies
    // between): both live in `a`, which takes RouteLds from 35 to 25 KB -- six waves per CU instead of four for a kernel whose
    // every phase is a latency chain.
    alignas(16) double a[4 * MAXP];
    alignas(16) double kx[MAXK], ky[MAXK];
    __device__ __forceinline__ double* px() { return a; }
    __device__ __forceinline__ double* py() { return a + MAXP; }
    __device__ __forceinline__ double* sn() { return a + 2 * MAXP; }
    __device__ __forceinline__ double* cd() { return a + 3 * MAXP; }
    __device__ __forceinline__ double* ks() { return a; }
    __device__ __forceinline__ double* khx() { return a + MAXK; }
    __device__ __forceinline__ double* khy() { return a + 2 * MAXK; }
    __device__ __forceinline__ const double* ks() const { return a; }
    __device__ __forceinline__ const double* khx() const { return a + MAXK; }
    __device__ __forceinline__ const double* khy() const { return a + 2 * MAXK; }
    double bc[8];                 // broadcast slots
    int nm, nkept, np, nk, bad, flag;
};

struct Pose { double x, y, h, s; };

__device__ __forceinline__ int lane_id() { return (int)threadIdx.x & 63; }
// The route-building functions below are written for ONE wave.  A workgroup may run them with several waves (the ego kernel
// does, to have 256 threads for its data-parallel parts): every wave then computes the same values redundantly -- pure
// functions of inputs that only change between barriers -- and the serial, in-place sections (cumulative sums, the insertion
// sort, table writes) are executed by the workgroup's first thread only.
__device__ __forceinline__ bool leader() { return threadIdx.x == 0; }

__device__ __forceinline__ double norm2(double dx, double dy) { return sqrt(dx * dx + dy * dy); }

__device__ __forceinline__ void flag(int32_t* status, int which) { atomicMax(status + which, 1); }

// connection j of a node (list order)
__device__ __forceinline__ void conn_at(const StriveLaneNode& r, const int32_t* cptr, const int32_t* cidx, const double* clen,
                                        int cur, int j, int& node, double& len) {
    if (j < 4) { node = r.node[j]; len = r.len[j]; }
    else { const int p = cptr[cur] + j; node = cidx[p]; len = clen[p]; }
}

// LinearPath (scipy interp1d, linear, assume_sorted): value of component arrays at q; lo/hi by searchsorted(side='left')
__device__ __forceinline__ int knot_hi(const double* t, int n, double q) {
    int lo = 0, hi = n;                 // first index with t[i] >= q
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (t[mid] < q) lo = mid + 1; else hi = mid;
    }
    return lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
}
__device__ __forceinline__ double lerp_at(const double* t, const double* y, int lo, int hi, double q) {
    const double slope = (y[hi] - y[lo]) / (t[hi] - t[lo]);
    return slope * (q - t[lo]) + y[lo];
}

// a[0] = 0, a[i] = a[i-1] + a[i] for i = 1 .. n-1: numpy's sequential cumulative sum, same additions in the same order, by the calling
// thread.  What is serial is the chain of float64 additions and nothing else: the terms come from LDS sixteen at a time (as pairs:
// `a` is 16-byte aligned and the batches start at even indices), the next batch while the current one is added up, and every sum is
// written into the register its term came in (no copies).  214 knots: 26 k cycles as a load-add-store loop, 11 k with batched loads
// and a running register, ~3 k like this (routes kernel: 225 of 1081 us were this function, option planner_dbg 8192).
__device__ __forceinline__ void seq_cumsum(double* a, int n) {
    constexpr int CB = 16;
    a[0] = 0.0;
    if (n < 2) return;
    double acc = a[1];                       // 0.0 + a[1] is a[1] (the sum starts from +0.0; -0.0 + 0.0 = +0.0 either way)
    acc = 0.0 + acc;
    a[1] = acc;
    int i = 2;
    if (i + CB <= n) {
        double2 v[CB / 2], nx[CB / 2];
        const double2* src = reinterpret_cast<const double2*>(a);
        double2* dst = reinterpret_cast<double2*>(a);
#pragma unroll
        for (int j = 0; j < CB / 2; ++j) v[j] = src[i / 2 + j];
        for (;;) {
            const bool more = i + 2 * CB <= n;
            if (more) {
#pragma unroll
                for (int j = 0; j < CB / 2; ++j) nx[j] = src[(i + CB) / 2 + j];
            }
#pragma unroll
            for (int j = 0; j < CB / 2; ++j) {
                v[j].x = acc + v[j].x;
                v[j].y = v[j].x + v[j].y;
                acc = v[j].y;
            }
#pragma unroll
            for (int j = 0; j < CB / 2; ++j) dst[i / 2 + j] = v[j];
            i += CB;
            if (!more) break;
#pragma unroll
            for (int j = 0; j < CB / 2; ++j) v[j] = nx[j];
        }
    }
    for (; i < n; ++i) { acc = acc + a[i]; a[i] = acc; }
}

// wave-uniform helpers of the chain walk: lane `idx`'s copy of a value, and a condition every lane agrees on as a scalar
__device__ __forceinline__ int lane_val(int v, int idx) { return __builtin_amdgcn_readlane(v, idx); }
__device__ __forceinline__ double lane_val(double v, int idx) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), idx), __builtin_amdgcn_readlane(__double2loint(v), idx));
}
__device__ __forceinline__ bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// all chains from v: breadth first, connection lists in order (expand_verts, :379-414).  WHOLE WAVE, uniform control flow: every
// lane walks the same chain on the same values, the workgroup's first thread writes the tables.  The nodes of a lane are consecutive
// in the reference's lane graph, so a walk that would be one dependent L2 round trip per node keeps a WINDOW of 64 consecutive node
// records (count, first connection, its length) in registers, one record per lane, refilled by one coalesced load per 64 nodes;
// a step of the walk is three v_readlane and one float64 addition (round 6: the window used to live in LDS -- three dependent LDS
// reads per node, 77 k of the routes kernel's 335 k ticks per pose).
// dir = +1 walks successors (the window reaches forward), -1 predecessors (backward).
template <int NCH, int NPOOL>
__device__ void build_chains(ChainTab<NCH, NPOOL>& c, const StriveLaneNode* rec, int N, const int32_t* cptr,
                             const int32_t* cidx, const double* clen, int v, double mindist, bool first_only, int dir, int32_t* status) {
    const int lane = lane_id();
    int n = 1, npool = 0;
    __syncthreads();
    if (leader()) { c.parent[0] = -1; c.fork[0] = 0; c.first[0] = v; c.len[0] = 0.0; }
    __syncthreads();
    int wb = -1000;                      // lane l holds the record of node wb + l
    int w_n = 0, w_node0 = 0;
    double w_len0 = 0.0;
    unsigned long long plain = 0ull;
    for (int ci = 0; ci < n; ++ci) {
        double length = c.len[ci];
        int cur = __builtin_amdgcn_readfirstlane(c.first[ci]);
        int total = __builtin_amdgcn_readfirstlane((int)c.fork[ci]) + 1, cnt = 0;
        const int ownoff = npool;
        while (uniform(length <= mindist)) {
            if (cur < wb || cur >= wb + 64) {
                wb = dir > 0 ? cur : cur - 63;
                const int i = wb + lane;
                w_n = 0;
                if (i >= 0 && i < N) { w_n = rec[i].n; w_node0 = rec[i].node[0]; w_len0 = rec[i].len[0]; }
                // nodes whose only step (first_only: whose first step) leads to the next node of the window
                plain = __ballot((first_only ? w_n >= 1 : w_n == 1) && w_node0 == i + dir);
            }
            const int idx = cur - wb;
            {
                // a run of plain nodes from cur on: nothing to look at but the lengths -- two v_readlane and one addition per node
                const unsigned long long m = dir > 0 ? (plain >> idx) : (plain << (63 - idx));
                const int run = ~m == 0ull ? 64 : (dir > 0 ? __builtin_ctzll(~m) : __builtin_clzll(~m));
                int steps = 0;
                while (steps < run && npool + steps < NPOOL && total + steps < MAXP - 4 && uniform(length <= mindist)) {
                    length = length + lane_val(w_len0, idx + dir * steps);
                    ++steps;
                }
                if (steps > 0) {
                    if (threadIdx.x < 64 && lane < steps) c.pool[npool + lane] = cur + dir * (lane + 1);
                    cur += dir * steps; npool += steps; cnt += steps; total += steps;
                    continue;
                }
            }
            const int rn = lane_val(w_n, idx);
            if (rn == 0) break;
            if (!first_only && rn > 1) {
                const StriveLaneNode r = rec[cur];
                for (int j = 1; j < r.n; ++j) {
                    int nd; double ln;
                    conn_at(r, cptr, cidx, clen, cur, j, nd, ln);
                    if (n < NCH) {
                        if (leader()) { c.parent[n] = (short)ci; c.fork[n] = (short)total; c.first[n] = nd; c.len[n] = length + ln; }
                        ++n;
                    } else if (leader()) flag(status, ST_CHAIN);
                }
            }
            if (npool >= NPOOL || total >= MAXP - 4) { if (leader()) flag(status, ST_NODES); break; }
            length = length + lane_val(w_len0, idx);
            cur = lane_val(w_node0, idx);
            if (leader()) c.pool[npool] = cur;
            ++npool; ++cnt; ++total;
        }
        if (leader()) { c.ownoff[ci] = (short)ownoff; c.nown[ci] = (short)cnt; c.len[ci] = length; }
        if (first_only) break;
        __syncthreads();                 // the next chain's start (written by the first thread above) is read by every lane
    }
    if (leader()) { c.n = n; c.npool = npool; }
    __syncthreads();
}

template <int NCH, int NPOOL>
__device__ __forceinline__ int chain_nodes(const ChainTab<NCH, NPOOL>& c, int ci) { return c.fork[ci] + 1 + c.nown[ci]; }

// node ids of chain ci into nid at index base + sign * position (whole wave; LDS to LDS)
template <int NCH, int NPOOL>
__device__ void chain_ids(const ChainTab<NCH, NPOOL>& c, int ci, int* nid, int base, int sign) {
    int cur = ci, hi = chain_nodes(c, ci);
    while (cur >= 0) {
        const int lo = c.fork[cur];
        for (int pos = lo + lane_id(); pos < hi; pos += 64)
            nid[base + sign * pos] = pos == lo ? c.first[cur] : c.pool[c.ownoff[cur] + pos - lo - 1];
        hi = lo;
        cur = c.parent[cur];
    }
}

// positions of the nodes nid[first .. first + n) into px / py (whole wave): the ids are collected first so that every position load
// of a route is in flight together -- gathered level by level of the chain tree (LDS read -> global load -> LDS store per level,
// for both chains in turn) the positions were half of assemble_route
__device__ void gather_nodes(const int* nid, const double* xy, double* px, double* py, int first, int n) {
    constexpr int U = (MAXP + 63) / 64;
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;       // (the workgroup shares the positions: one wave takes five each, eleven one)
    int v[U];
    double x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = tid + nthr * u;
        v[u] = p < n ? nid[first + p] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (v[u] >= 0) { x[u] = xy[2 * v[u]]; y[u] = xy[2 * v[u] + 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int p = tid + nthr * u;
        if (v[u] >= 0) { px[first + p] = x[u]; py[first + p] = y[u]; }
    }
}

struct RouteGeom { double back, fwd, need_f, need_b; int nb, nf; };

__device__ __forceinline__ RouteGeom route_geom(const StrivePlannerCfg& cfg, double s) {
    RouteGeom g;
    g.back = s > 0 ? 1.0 : 1.0 + fabs(s) * cfg.tmax;
    const double f0 = 1.0 + cfg.smax * cfg.tmax;
    g.fwd = s < 0 ? f0 : fmax(f0, 1.0 + s * cfg.tmax);
    g.need_f = g.fwd + SBUFFER + cfg.xydistmax;
    g.need_b = g.back + SBUFFER + cfg.xydistmax;
    g.nb = (int)((g.back + SBUFFER) / LANE_DS) + 1;
    g.nf = (int)((g.fwd + SBUFFER) / LANE_DS) + 1;
    return g;
}

// np.concatenate((np.linspace(-back - SBUFFER, 0, nb + 1)[:-1], np.linspace(0, fwd + SBUFFER, nf)))[i]
__device__ __forceinline__ double s_eval_at(const RouteGeom& g, int i) {
    if (i < g.nb) {
        const double start = -g.back - SBUFFER;
        const double step = (0.0 - start) / (double)g.nb;
        return (double)i * step + start;
    }
    const int j = i - g.nb;
    const double stop = g.fwd + SBUFFER;
    if (j == g.nf - 1) return stop;
    const double step = stop / (double)(g.nf - 1);
    return (double)j * step + 0.0;
}

// lane matches of a pose, in edge order (get_lane_matches / edge_closest_point, :298-359), and their clusters
// (cluster_matches_combine, :324-376).  Whole wave; R.nm, R.nkept, R.kept valid afterwards.
__device__ void match_and_cluster(RouteLds& R, const StrivePlannerMap& mp, const StrivePlannerCfg& cfg, const Pose& o, int32_t* status) {
    const int lane = lane_id();
    int nm = 0;
    const double ch = cos(o.h), sh = sin(o.h);
    const double fx = floor((o.x - mp.gx0) / mp.gcell), fy = floor((o.y - mp.gy0) / mp.gcell);
    if (fx >= 0.0 && fy >= 0.0 && fx < (double)mp.gnx && fy < (double)mp.gny) {
        const int cell = (int)fy * mp.gnx + (int)fx;
        const int c0 = mp.cell_ptr[cell], c1 = mp.cell_ptr[cell + 1];
        for (int base = c0; base < c1; base += 64) {
            const int i = base + lane;
            bool ok = false;
            double ptx = 0, pty = 0, dist = 0;
            int e = 0;
            if (i < c1) {
                e = mp.cell_edges[i];
                const double* ed = mp.edges + 5 * (size_t)e;
                const double dx = ed[2], dy = ed[3];
                const double cdist = 1.0 - dx * ch - dy * sh;
                if (cdist < cfg.cdistmax) {
                    double along = (o.x - ed[0]) * dx + (o.y - ed[1]) * dy;
                    along = fmin(fmax(along, 0.0), ed[4]);
                    ptx = ed[0] + along * dx;
                    pty = ed[1] + along * dy;
                    dist = norm2(o.x - ptx, o.y - pty);
                    ok = dist < cfg.xydistmax;
                }
            }
            const unsigned long long m = __ballot(ok);
            const int pos = nm + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (ok) {
                if (pos < MAXM) {
                    R.m_px[pos] = ptx; R.m_py[pos] = pty; R.m_d[pos] = dist;
                    R.m_v0[pos] = mp.edge_ix[2 * e]; R.m_v1[pos] = mp.edge_ix[2 * e + 1];
                } else flag(status, ST_MATCH);
            }
            nm += __builtin_popcountll(m);
        }
    }
    if (nm > MAXM) nm = MAXM;
    __syncthreads();
    if (leader()) {
        R.nm = nm;
        // stable order by distance (np.argsort on <= 16 entries is an insertion sort)
        for (int i = 0; i < nm; ++i) {
            int j = i;
            while (j > 0 && R.m_d[R.m_order[j - 1]] > R.m_d[i]) { R.m_order[j] = R.m_order[j - 1]; --j; }
            R.m_order[j] = (short)i;
        }
        for (int i = 0; i < nm; ++i) R.m_done[i] = 0;
        int nkept = 0;
        for (int oi = 0; oi < nm; ++oi) {
            const int k = R.m_order[oi];
            if (R.m_done[k]) continue;
            if (nkept < MAXKEEP) R.kept[nkept++] = (short)k; else flag(status, ST_KEEP);
            // breadth-first over the matches connected to k through matches, forward then backward (cluster_bfs, :349-376).  Every
            // match IS a graph edge, so "(b, c) for c in out_edges[b] that is a match" is simply "a match that starts at b" (and a
            // predecessor match one that ends at a): the walk needs the match list only, not the node records
            for (int dir = 0; dir < 2; ++dir) {
                int qh = 0, qt = 0;
                R.queue[qt++] = (short)k;
                R.m_done[k] = 1;
                while (qh < qt) {
                    const int cur = R.queue[qh++];
                    const int a = R.m_v0[cur], b = R.m_v1[cur];
                    for (int q = 0; q < nm; ++q) {
                        if (R.m_done[q]) continue;
                        if (dir == 0 ? R.m_v0[q] == b : R.m_v1[q] == a) { R.m_done[q] = 1; R.queue[qt++] = (short)q; }
                    }
                }
            }
        }
        R.nkept = nkept;
    }
    __syncthreads();
}

// constant-heading route when no lane is near (constant_heading_spline, :477-484)
__device__ void straight_route(RouteLds& R, const Pose& o, const RouteGeom& g) {
    if (leader()) {
        const double c = cos(o.h), s = sin(o.h);
        R.ks()[0] = -g.back; R.kx[0] = o.x - g.back * c; R.ky[0] = o.y - g.back * s; R.khx()[0] = c; R.khy()[0] = s;
        R.ks()[1] = g.fwd;   R.kx[1] = o.x + g.fwd * c;  R.ky[1] = o.y + g.fwd * s;  R.khx()[1] = c; R.khy()[1] = s;
        R.nk = 2;
        R.bad = 0;
    }
    __syncthreads();
}

// one route through (backward chain bi, forward chain fi) of the current match (local_lane_closest / xy2spline, :433-556).
// Whole wave; leaves the knots in R.ks() / kx / ky / khx / khy (R.nk) and R.bad != 0 if the route could not be built.
__device__ void assemble_route(RouteLds& R, const StrivePlannerMap& mp, const Pose& o, const RouteGeom& g, int fi, int bi, int32_t* status,
                               unsigned long long* tp = nullptr, int dbg = 0) {
    // (option planner_dbg, measurement only: 256 / 512 / 1024 / 2048 run the node gather / the closest-point loop / the resampling /
    //  the heading loop twice, 4096 / 8192 add a second cumulative sum of the node / of 128 knot terms: the differences are the phases)
    const int lane = lane_id();
    long long ta = tp ? (long long)clock64() : 0;
    auto tka = [&](int id) { if (tp) { const long long n = (long long)clock64(); tp[id] += (unsigned long long)(n - ta); ta = n; } };
    const int nbv = chain_nodes(R.cb, bi), nfv = chain_nodes(R.cf, fi);
    const double flen = R.cf.len[fi], blen = R.cb.len[bi];
    const bool ext_f = flen <= g.need_f, ext_b = blen <= g.need_b;
    const int shift = ext_b ? 1 : 0;
    const int np = nbv + nfv + shift + (ext_f ? 1 : 0);
    const int nk = g.nb + g.nf;
    if (np > MAXP || nk > MAXK) {
        if (leader()) { flag(status, np > MAXP ? ST_NODES : ST_KNOTS); R.bad = 1; }
        __syncthreads();
        return;
    }
    for (int rep = 0; rep < ((dbg & 256) ? 2 : 1); ++rep) {
        int* nid = reinterpret_cast<int*>(R.cd());        // (the distances that live there are written further down)
        __syncthreads();
        chain_ids(R.cb, bi, nid, shift + nbv - 1, -1);
        chain_ids(R.cf, fi, nid, shift + nbv, +1);
        __syncthreads();
        gather_nodes(nid, mp.xy, R.px(), R.py(), shift, nbv + nfv);
    }
    __syncthreads();
    tka(0);                 // node positions
    if (leader()) {
        R.bad = 0;
        if (ext_f) {                    // dead end ahead: extend straight
            const int last = shift + nbv + nfv - 1;
            double dx = R.px()[last] - R.px()[last - 1], dy = R.py()[last] - R.py()[last - 1];
            const double n = norm2(dx, dy);
            dx = dx / n; dy = dy / n;
            const double ext = 1.0 + g.need_f - flen;
            R.px()[last + 1] = R.px()[last] + dx * ext;
            R.py()[last + 1] = R.py()[last] + dy * ext;
        }
        if (ext_b) {
            double dx = R.px()[1] - R.px()[2], dy = R.py()[1] - R.py()[2];
            const double n = norm2(dx, dy);
            dx = dx / n; dy = dy / n;
            const double ext = 1.0 + g.need_b - blen;
            R.px()[0] = R.px()[1] + dx * ext;
            R.py()[0] = R.py()[1] + dy * ext;
        }
        R.np = np;
    }
    __syncthreads();
    const int ns = np - 1;
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;       // (loops over nodes and knots are shared by the workgroup's waves)
    for (int rep = 0; rep < ((dbg & 512) ? 2 : 1); ++rep)
    for (int i = tid; i < ns; i += nthr) {
        const double sx = R.px()[i + 1] - R.px()[i], sy = R.py()[i + 1] - R.py()[i];
        const double sl = norm2(sx, sy);
        const double dx = sx / sl, dy = sy / sl;
        double along = (o.x - R.px()[i]) * dx + (o.y - R.py()[i]) * dy;
        along = fmin(fmax(along, 0.0), sl);
        const double cx = R.px()[i] + along * dx, cy = R.py()[i] + along * dy;
        R.cd()[i] = norm2(o.x - cx, o.y - cy);
        R.sn()[i + 1] = sl;
    }
    __syncthreads();
    if (leader()) {
        // the locally closest point, walking downhill from the matched edge
        int k = nbv - 1 + shift;
        while (k - 1 >= 0 && R.cd()[k - 1] < R.cd()[k]) --k;
        while (k + 1 < ns && R.cd()[k + 1] < R.cd()[k]) ++k;
        const double sx = R.px()[k + 1] - R.px()[k], sy = R.py()[k + 1] - R.py()[k];
        const double sl = R.sn()[k + 1];
        const double dx = sx / sl, dy = sy / sl;
        double along = (o.x - R.px()[k]) * dx + (o.y - R.py()[k]) * dy;
        along = fmin(fmax(along, 0.0), sl);
        const double ax = R.px()[k] + along * dx, ay = R.py()[k] + along * dy;
        if (dbg & 4096) seq_cumsum(R.cd(), np);
        seq_cumsum(R.sn(), np);
        R.bc[0] = ax; R.bc[1] = ay;
        R.flag = 0;
        R.bc[2] = R.sn()[k];
        R.bc[3] = norm2(ax - R.px()[k], ay - R.py()[k]);
    }
    __syncthreads();
    {
        const double off1 = R.bc[2], off2 = R.bc[3];
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += blockDim.x) R.sn()[i] = R.sn()[i] - off1 - off2;      // (in place: once per element)
    }
    __syncthreads();
    const double ax = R.bc[0], ay = R.bc[1];
    tka(1);                 // closest point, arc lengths
    bool range_bad = false;
    for (int rep = 0; rep < ((dbg & 1024) ? 2 : 1); ++rep)
    for (int i = tid; i < nk; i += nthr) {
        const double q = s_eval_at(g, i);
        if (q < R.sn()[0] || q > R.sn()[np - 1]) { range_bad = true; R.kx[i] = 0; R.ky[i] = 0; continue; }
        const int hi = knot_hi(R.sn(), np, q), lo = hi - 1;
        const double e = exp(-(q * q) / LANE_SIG2);
        R.kx[i] = lerp_at(R.sn(), R.px(), lo, hi, q) + (o.x - ax) * e;
        R.ky[i] = lerp_at(R.sn(), R.py(), lo, hi, q) + (o.y - ay) * e;
    }
    if (range_bad) R.flag = 1;              // (cleared by the first thread two barriers ago)
    __syncthreads();
    if (R.flag) {
        if (leader()) { flag(status, ST_RANGE); R.bad = 1; }
        __syncthreads();
        return;
    }
    tka(2);                 // resampling + blend
    for (int rep = 0; rep < ((dbg & 2048) ? 2 : 1); ++rep)
    for (int i = tid; i < nk - 1; i += nthr) {
        const double dx = R.kx[i + 1] - R.kx[i], dy = R.ky[i + 1] - R.ky[i];
        const double dl = norm2(dx, dy);
        R.khx()[i] = dx / dl;
        R.khy()[i] = dy / dl;
        R.ks()[i + 1] = dl;
    }
    __syncthreads();
    if (leader()) {
        R.khx()[nk - 1] = R.khx()[nk - 2];
        R.khy()[nk - 1] = R.khy()[nk - 2];
        R.khx()[g.nb] = cos(o.h);             // the route passes through the object's heading exactly
        R.khy()[g.nb] = sin(o.h);
        if (dbg & 8192) seq_cumsum(R.a + 3 * MAXK, 4 * MAXP - 3 * MAXK);
        seq_cumsum(R.ks(), nk);
        R.bc[4] = R.ks()[g.nb];
        R.nk = nk;
    }
    __syncthreads();
    {
        const double s0 = R.bc[4];
        __syncthreads();
        for (int i = threadIdx.x; i < nk; i += blockDim.x) R.ks()[i] = R.ks()[i] - s0;               // (in place: once per element)
    }
    __syncthreads();
    if (leader() && !(R.ks()[0] < -g.back && R.ks()[nk - 1] > g.fwd)) { flag(status, ST_RANGE); R.bad = 1; }
    __syncthreads();
    tka(3);                 // headings, knot arc lengths
}

// (x, y, heading angle) of the route in R at arc length q; false outside the route
__device__ __forceinline__ bool route_pose(const RouteLds& R, double q, double& x, double& y, double& h) {
    if (q < R.ks()[0] || q > R.ks()[R.nk - 1]) return false;
    const int hi = knot_hi(R.ks(), R.nk, q), lo = hi - 1;
    x = lerp_at(R.ks(), R.kx, lo, hi, q);
    y = lerp_at(R.ks(), R.ky, lo, hi, q);
    h = atan2(lerp_at(R.ks(), R.khy(), lo, hi, q), lerp_at(R.ks(), R.khx(), lo, hi, q));
    return true;
}

// speeds moving from s towards target by at most acc*dt per step (compute_speed_profile, :670-683): entry k
__device__ __forceinline__ double ramp_at(double s, double target, double acc, int k, double dt) {
    if (target > s) return fmin(s + (double)k * acc * dt, target);
    if (target < s) return fmax(s - (double)k * acc * dt, target);
    return s;
}

__device__ __forceinline__ double signed_speed(double x0, double y0, double x1, double y1, double h1, double dt) {
    const double dx = x1 - x0, dy = y1 - y0;
    const double mag = sqrt(dx * dx + dy * dy) / dt;
    return dx * cos(h1) + dy * sin(h1) >= 0 ? mag : -mag;
}

// np.linspace(a, b, n)[i]
__device__ __forceinline__ double linspace_at(double a, double b, int n, int i) {
    if (n == 1) return a;
    if (i == n - 1) return b;
    const double step = (b - a) / (double)(n - 1);
    return (double)i * step + a;
}

struct Work {
    double* wstate;         // (NR, K, 4) pose + speed of the non-ego objects
    int8_t* present;        // (NR, K)
    double* traj;           // (B*K, cap, ENT) predicted trajectories: object x, y, l, w, duplicate mask, then NT x (x, y, h)
    int32_t* traj_cnt;      // (B*K)
    double* ego;            // (B, 8) x, y, h, s, l, w
    double* route;          // (B, 5, MAXK) s, x, y, cos, sin of the ego's route
    int32_t* route_nk;      // (B)
    int32_t* prefer_stop;   // (B)
    double* prof;           // (B, MAXPROF, 3) s1, acc, final distance
    double* circ;           // (B, P, NT, 10) circle centres of the ego's boxes
    double* part;           // (B, NCHUNK, P, NT) partial gap minima
    int32_t* part_cnt;      // (B, NCHUNK) trajectories within the interaction distance
    double* poses;          // (B, K, 4)
    int32_t* scene_bad;     // (B)
    int K, cap, ENT, P, NT;
    int nchunk;             // trajectory chunks per scene of this launch (<= NCHUNK)
    int dbg;                // option planner_dbg (measurement only): 1 gap: no circle pairs, 4 no staging arithmetic,
                            // 8 gap: prologue only; 16 ego: no route / profiles / circles; 32 routes: no emission, 64 routes: no assembly, 128 routes: matches only
    size_t nzero;
    unsigned long long* tprof;   // measurement only (STRIVE_PLANNER_PROF): clock sums of the ego kernel's phases, in the workspace's spare tail
};

__global__ void planner_world_kernel(StrivePlanner pl, Work w, const double* __restrict__ obs, const double* __restrict__ agent_t, int T) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= pl.NR) return;
    const double* in = pl.init + 6 * (size_t)pl.row_obj[r];
    double x = in[0], y = in[1], h = in[2], s = in[3];
    const double* ob = obs + (size_t)r * T * 4;
    int n = T + 1;                                  // frames up to the first NaN one (frame 0 = the initial pose)
    for (int t = 0; t < T; ++t) {
        const double sum = ob[4 * t] + ob[4 * t + 1] + ob[4 * t + 2] + ob[4 * t + 3];
        if (sum != sum) { n = t + 1; break; }
    }
    const double c0 = cos(h), s0 = sin(h);
    const double x0 = x, y0 = y;
    const double tb = n == 1 ? 0.0 : agent_t[n - 2];
    double* ws = w.wstate + (size_t)r * w.K * 4;
    int8_t* pr = w.present + (size_t)r * w.K;
    ws[0] = x; ws[1] = y; ws[2] = h; ws[3] = s;
    pr[0] = 1;
    bool here = true;
    double t = 0.0;
    for (int k = 1; k < w.K; ++k) {
        const double t1 = t + pl.cfg.dt;
        if (here && n > 1 && 0.0 <= t1 && t1 <= tb) {
            // knots: time 0 (initial pose) and agent_t[0 .. n-2]
            int hi = 1;
            while (hi < n - 1 && agent_t[hi - 1] < t1) ++hi;          // first knot index >= 1 with time >= t1, clipped to n-1
            const int lo = hi - 1;
            const double tlo = lo == 0 ? 0.0 : agent_t[lo - 1], thi = agent_t[hi - 1];
            double v[4];
            for (int c = 0; c < 4; ++c) {
                const double ylo = lo == 0 ? (c == 0 ? x0 : c == 1 ? y0 : c == 2 ? c0 : s0) : ob[4 * (lo - 1) + c];
                const double yhi = ob[4 * (hi - 1) + c];
                const double slope = (yhi - ylo) / (thi - tlo);
                v[c] = slope * (t1 - tlo) + ylo;
            }
            const double h1 = atan2(v[3], v[2]);
            s = signed_speed(x, y, v[0], v[1], h1, pl.cfg.dt);
            x = v[0]; y = v[1]; h = h1;
        } else {
            here = false;
        }
        ws[4 * k] = x; ws[4 * k + 1] = y; ws[4 * k + 2] = h; ws[4 * k + 3] = s;
        pr[k] = here ? 1 : 0;
        t = t1;
    }
}

// routes of one pose: `begin_group(n)` announces the n routes of the next match (or the single constant-heading route), then
// `on_route(i)` is called by the whole wave for i = 0..n-1 with the knots in R (R.bad != 0: the route could not be built)
template <bool FIRST_ONLY, class G, class F>
__device__ int for_each_route(RouteLds& R, const StrivePlannerMap& mp, const StrivePlannerCfg& cfg, const Pose& o, int32_t* status,
                              G&& begin_group, F&& on_route, unsigned long long* tp = nullptr, int dbg = 0) {
    const RouteGeom g = route_geom(cfg, o.s);
    long long t0 = tp ? (long long)clock64() : 0;
    auto tk = [&](int id) { if (tp) { const long long n = (long long)clock64(); tp[id] += (unsigned long long)(n - t0); t0 = n; } };
    match_and_cluster(R, mp, cfg, o, status);
    tk(0);
    const int nkept = R.nkept;
    if (nkept == 0) {
        begin_group(1);
        straight_route(R, o, g);
        on_route(0);
        __syncthreads();
        return 0;
    }
    const int nmatch = (dbg & 128) ? 0 : (FIRST_ONLY ? 1 : nkept);          // (128: matches only)
    for (int mi = 0; mi < nmatch; ++mi) {
        const int m = R.kept[mi];
        build_chains(R.cf, mp.succ, mp.N, mp.succ_ptr, mp.succ_idx, mp.succ_len, R.m_v1[m], g.need_f, FIRST_ONLY, +1, status);
        build_chains(R.cb, mp.pred, mp.N, mp.pred_ptr, mp.pred_idx, mp.pred_len, R.m_v0[m], g.need_b, FIRST_ONLY, -1, status);
        tk(1);
        const int nf = FIRST_ONLY ? 1 : R.cf.n, nb = FIRST_ONLY ? 1 : R.cb.n;
        begin_group(nf * nb);
        for (int fi = 0; fi < nf; ++fi)
            for (int bi = 0; bi < nb; ++bi) {
                if (!(dbg & 64)) assemble_route(R, mp, o, g, fi, bi, status, (tp && !FIRST_ONLY) ? tp + 12 : nullptr, dbg);
                on_route(fi * nb + bi);
                __syncthreads();
            }
        tk(2);
    }
    return nkept;
}

// `status` of the rollout kernels is (B, 8): a kernel offsets it to its scene's row before anything can set a flag, so a capacity
// or range violation poisons (and is reported for) the scene it happened in and no other
__global__ void __launch_bounds__(64) planner_routes_kernel(StrivePlanner pl, Work w, int32_t* status_all) {
    __shared__ RouteLds R;
    __shared__ double pd[MAXPRED][MAXNT];
    __shared__ int slot_base;
    __shared__ int dmask[MAXPRED];
    const int lane = lane_id();
    const int r = blockIdx.x / w.K, k = blockIdx.x % w.K;
    if (!w.present[(size_t)r * w.K + k]) return;
    const double* ws = w.wstate + ((size_t)r * w.K + k) * 4;
    const Pose o = {ws[0], ws[1], ws[2], ws[3]};
    const int b = pl.row_scene[r];
    int32_t* status = status_all + NSTATUS * (size_t)b;
    const double* in = pl.init + 6 * (size_t)pl.row_obj[r];
    const double l = in[4], wd = in[5];
    const StrivePlannerCfg& cfg = pl.cfg;
    const StrivePlannerMap& mp = pl.maps[pl.scene_map[b]];
    const int npred = cfg.npredsfacs * cfg.npredafacs;
    // distances travelled under the predicted speed profiles (:686-700)
    if (lane < npred) {
        const double sf = cfg.predsfacs[lane / cfg.npredafacs], af = cfg.predafacs[lane % cfg.npredafacs];
        const double target = o.s * sf, acc = cfg.accmax * af;
        double d = 0.0;
        pd[lane][0] = 0.0;
        for (int t = 1; t < w.NT; ++t) {
            d = d + ramp_at(o.s, target, acc, t, cfg.preddt) * cfg.preddt;
            pd[lane][t] = d;
        }
    }
    __syncthreads();
    const int slot = b * w.K + k;
    double* tbase = w.traj + (size_t)slot * w.cap * w.ENT;
    // (measurement only, option planner_prof: clock sums over ALL waves -- slots 8.. of the profile tail: match, chains, assemble +
    //  emit, emit alone, poses, routes)
    unsigned long long prof[16] = {0};           // (summed per wave, added to the global slots once at the end: sixteen hot addresses
    unsigned long long* tp = (w.tprof && lane == 0) ? prof : nullptr;      //  under atomics from 1500 waves distorted the first profile)
    if (tp) tp[4] += 1ull;
    auto begin_group = [&](int n) {
        __syncthreads();
        if (lane == 0) slot_base = atomicAdd(w.traj_cnt + slot, n * npred);
        if (tp) tp[5] += (unsigned long long)n;
        __syncthreads();
    };
    auto on_route = [&](int ri) {
        if (w.dbg & 32) return;
        const long long t_emit = tp ? (long long)clock64() : 0;
        struct EmitTimer {
            unsigned long long* tp; long long t0;
            __device__ ~EmitTimer() { if (tp) tp[3] += (unsigned long long)((long long)clock64() - t0); }
        } emit_timer{tp, t_emit};
        const int base = slot_base + ri * npred;
        const bool bad = R.bad != 0;
        if (base + npred > w.cap) { if (lane == 0) { flag(status, ST_TRAJ); w.scene_bad[b] = 1; } return; }
        if (bad && lane == 0) w.scene_bad[b] = 1;
        // Routes of one match that share their backward chain and a prefix of their forward chain (breadth-first tree: chain fi
        // continues its parent up to the fork) are IDENTICAL up to the fork, and so are the trajectory points that do not reach
        // it -- most of them: an object travels ~25 m in the 5 s horizon, crossings are further apart.  Points that equal the
        // parent route's bit for bit are marked; the gap kernel skips them (the minimum over duplicates is the same).
        int par_base = -1;
        if (R.nkept > 0 && !bad) {
            const int nb = R.cb.n, fi = ri / nb, bi = ri - fi * nb;
            const int pf = R.cf.parent[fi];
            if (pf >= 0) par_base = slot_base + (pf * nb + bi) * npred;
        }
        if (lane < MAXPRED) dmask[lane] = 0;
        __syncthreads();
        for (int it = lane; it < npred * w.NT; it += 64) {
            const int p = it / w.NT, t = it % w.NT;
            double* ent = tbase + (size_t)(base + p) * w.ENT;
            double x = NAN, y = NAN, h = NAN;
            if (!bad && !route_pose(R, pd[p][t], x, y, h)) { flag(status, ST_RANGE); w.scene_bad[b] = 1; }
            if (t == 0) { ent[0] = o.x; ent[1] = o.y; ent[2] = l; ent[3] = wd; }
            ent[5 + 3 * t] = x; ent[6 + 3 * t] = y; ent[7 + 3 * t] = h;
            if (par_base >= 0) {
                const double* pe = tbase + (size_t)(par_base + p) * w.ENT;
                if (pe[5 + 3 * t] == x && pe[6 + 3 * t] == y && pe[7 + 3 * t] == h) atomicOr(&dmask[p], 1 << t);
            }
        }
        __syncthreads();
        if (lane < npred) tbase[(size_t)(base + lane) * w.ENT + 4] = (double)(unsigned)dmask[lane];
    };
    for_each_route<false>(R, mp, cfg, o, status, begin_group, on_route, tp, w.dbg);
    if (tp)
        for (int i = 0; i < 16; ++i) atomicAdd(w.tprof + 8 + i, prof[i]);
}

// circle centres of a box (boxes2circles, :860-882): c[0..3] towards the corners, c[4] the centre
__device__ __forceinline__ void box_circles(double x, double y, double h, double l, double w, double* cx, double* cy) {
    const double L = fmax(l, w), W = fmin(l, w);
    const double H = l < w ? h + M_PI / 2.0 : h;
    const double ch = cos(H), sh = sin(H);
    const double k0 = (L - W) / 2 + W / 4, k1 = W / 4;
    const double a0x = k0 * ch, a0y = k0 * sh, a1x = k1 * -sh, a1y = k1 * ch;
    cx[0] = x + a0x + a1x; cy[0] = y + a0y + a1y;
    cx[1] = x - a0x + a1x; cy[1] = y - a0y + a1y;
    cx[2] = x - a0x - a1x; cy[2] = y - a0y - a1y;
    cx[3] = x + a0x - a1x; cy[3] = y + a0y - a1y;
    cx[4] = x; cy[4] = y;
}

__global__ void __launch_bounds__(1024) planner_ego_kernel(StrivePlanner pl, Work w, int k, int32_t* status_all) {
    __shared__ RouteLds R;
    __shared__ double pdist[MAXPROF][MAXNT];        // distances of the profiles; before that, (1 - pr) of step k-1's profiles
    __shared__ double risk[MAXPROF], pfd[MAXPROF];
    const int lane = lane_id(), tid = threadIdx.x, nthr = blockDim.x;
    const int b = blockIdx.x;
    int32_t* status = status_all + NSTATUS * (size_t)b;
    const StrivePlannerCfg& cfg = pl.cfg;
    const StrivePlannerMap& mp = pl.maps[pl.scene_map[b]];
    double* eg = w.ego + 8 * (size_t)b;
    const int P = w.P, NT = w.NT;
    long long tlast = w.tprof ? (long long)clock64() : 0;
    auto tick = [&](int id) {
        if (w.tprof && tid == 0 && b == 0) {
            const long long now = (long long)clock64();
            atomicAdd(w.tprof + id, (unsigned long long)(now - tlast));
            tlast = now;
        }
    };
    auto sub = [&](int id) {        // (cumulative since the kernel's start, slots 14..19: where inside the first phase)
        if (w.tprof && tid == 0 && b == 0) atomicAdd(w.tprof + id, (unsigned long long)((long long)clock64() - tlast));
    };
    if (k == 0 && tid == 0) {
        const double* in = pl.init + 6 * (size_t)(pl.ptr[b] + pl.ego_idx);
        for (int c = 0; c < 6; ++c) eg[c] = in[c];
    }
    __syncthreads();
    if (k > 0) {
        // the route of step k-1 (what the action below is read from) back into LDS with one coalesced load: its binary search used
        // to be eight dependent round trips to global memory by thread 0, ~5 of the kernel's 48 us
        {
            const double* rt = w.route + (size_t)b * 5 * MAXK;
            const int nk = w.route_nk[b];
            for (int i = tid; i < nk; i += nthr) {
                R.ks()[i] = rt[i]; R.kx[i] = rt[MAXK + i]; R.ky[i] = rt[2 * MAXK + i]; R.khx()[i] = rt[3 * MAXK + i]; R.khy()[i] = rt[4 * MAXK + i];
            }
            if (tid < P) pfd[tid] = w.prof[((size_t)b * MAXPROF + tid) * 3 + 2];      // (final distances of the profiles, for the choice)
        }
        // risk of the profiles of step k-1 (score_dists, :724-728; plot_plan_info, :768-801): all threads take the minimum over
        // the trajectory chunks and the tanh score of one (profile, time) each, then one lane per profile multiplies in time order
        int nother = 0;
        for (int c = 0; c < w.nchunk; ++c) nother += w.part_cnt[b * NCHUNK + c];
        sub(14);
        for (int it = tid; it < P * NT; it += nthr) {
            const int p = it / NT, t = it % NT;
            double gap = INFINITY;
            for (int c = 0; c < w.nchunk; ++c) gap = fmin(gap, w.part[(((size_t)b * NCHUNK + c) * P + p) * NT + t]);
            const double wt = cfg.score_wmin + (double)t * cfg.score_wfac;
            double pr = 1.0 + tanh(-gap * wt);
            if (gap < 0) pr = 1.0;
            pdist[p][t] = 1.0 - pr;
        }
        __syncthreads();
        sub(15);
        if (tid < P) {
            double prod = 1.0;
            for (int t = 0; t < NT; ++t) prod = prod * pdist[tid][t];
            risk[tid] = 1.0 - prod;
        }
        __syncthreads();
        sub(16);
        if (tid == 0) {
            const double* pf = w.prof + (size_t)b * MAXPROF * 3;
            int best = 0;
            if (nother == 0) {
                for (int p = 1; p < P; ++p) if (pfd[p] > pfd[best]) best = p;
            } else {
                int first_ok = -1;
                for (int p = 0; p < P; ++p) if (risk[p] < cfg.col_plim) { first_ok = p; break; }
                if (first_ok < 0) {
                    for (int p = 1; p < P; ++p) if (risk[p] < risk[best]) best = p;
                } else {
                    best = first_ok;
                    const bool stop = w.prefer_stop[b] != 0;
                    for (int p = first_ok + 1; p < P; ++p) {
                        if (!(risk[p] < cfg.col_plim)) continue;
                        if (stop ? pfd[p] < pfd[best] : pfd[p] > pfd[best]) best = p;
                    }
                }
            }
            sub(17);
            // compute_action (:829-857) / postprocess_act_for_speed (:642-666)
            const double x = eg[0], y = eg[1], h = eg[2], s = eg[3];
            const double s_next = ramp_at(s, pf[3 * best], pf[3 * best + 1], 1, cfg.dt);
            const double* rt = R.ks();                 // (the route of step k-1, loaded above)
            const int nk = w.route_nk[b];
            const double q = cfg.dt * s_next;
            double px, py, ph;
            bool ok = nk >= 2 && !(q < rt[0] || q > rt[nk - 1]);
            if (ok) {
                const int hi = knot_hi(rt, nk, q), lo = hi - 1;
                const double nx = lerp_at(rt, R.kx, lo, hi, q), ny = lerp_at(rt, R.ky, lo, hi, q);
                const double nh = atan2(lerp_at(rt, R.khy(), lo, hi, q), lerp_at(rt, R.khx(), lo, hi, q));
                const double sp = signed_speed(x, y, nx, ny, nh, cfg.dt);
                const int sg0 = sp > 0 ? 1 : (sp < 0 ? -1 : 0), sg1 = s_next > 0 ? 1 : (s_next < 0 ? -1 : 0);
                const double dx = nx - x, dy = ny - y;
                const double dn = norm2(dx, dy);
                if (sg0 != sg1 || dn == 0.0) {
                    px = x + cos(h) * s_next * cfg.dt; py = y + sin(h) * s_next * cfg.dt; ph = h;
                } else {
                    px = x + dx / dn * fabs(s_next) * cfg.dt; py = y + dy / dn * fabs(s_next) * cfg.dt; ph = nh;
                }
                if (!(fabs(signed_speed(x, y, px, py, ph, cfg.dt) - s_next) < 1e-6)) { flag(status, ST_ACT); ok = false; }
            } else {
                flag(status, ST_RANGE);
            }
            if (!ok || w.scene_bad[b]) { px = py = ph = NAN; w.scene_bad[b] = 1; }
            double* po = w.poses + ((size_t)b * w.K + (k - 1)) * 4;
            po[0] = px; po[1] = py; po[2] = cos(ph); po[3] = sin(ph);
            // update_wstate (:601-621)
            const double sn = signed_speed(x, y, px, py, ph, cfg.dt);
            eg[0] = px; eg[1] = py; eg[2] = ph; eg[3] = sn;
        }
        __syncthreads();
    }
    tick(0);                                          // risk scores, choice, action
    if (k >= w.K) return;
    if (w.dbg & 16) return;
    if (w.scene_bad[b]) return;                       // (NaN state: nothing more to plan for this scene)
    const Pose o = {eg[0], eg[1], eg[2], eg[3]};
    const double el = eg[4], ew = eg[5];
    double* rt = w.route + (size_t)b * 5 * MAXK;
    auto on_route = [&](int) {
        for (int i = tid; i < R.nk; i += nthr) {
            rt[i] = R.ks()[i]; rt[MAXK + i] = R.kx[i]; rt[2 * MAXK + i] = R.ky[i]; rt[3 * MAXK + i] = R.khx()[i]; rt[4 * MAXK + i] = R.khy()[i];
        }
        if (tid == 0) { w.route_nk[b] = R.bad ? 0 : R.nk; if (R.bad) w.scene_bad[b] = 1; }
    };
    const int nkept = for_each_route<true>(R, mp, cfg, o, status, [](int) {}, on_route, w.tprof && tid == 0 && b == 0 ? w.tprof + 4 : nullptr, w.dbg & ~(64 | 128));
    __syncthreads();
    tick(1);                                          // match + chains + route
    if (tid == 0) w.prefer_stop[b] = nkept == 0;
    if (R.bad) return;
    // candidate speed profiles (gen_sprofiles, :804-826)
    if (tid < P) {
        const int ns = cfg.plannspeeds;
        const int i2 = tid % ns, i1 = (tid / ns) % ns, fi = tid / (ns * ns);
        const double acc = cfg.planaccfacs[fi] * cfg.accmax;
        const int n1 = cfg.nsteps / 2, n2 = cfg.nsteps - n1;
        const double dt = cfg.preddt;
        const double s0 = o.s;
        const double r1 = (double)n1 * dt * acc;
        const double s1 = linspace_at(fmax(0.0, s0 - r1), fmin(cfg.smax, s0 + r1), ns, i1);
        const double f_last = ramp_at(s0, s1, acc, n1, dt);
        const double r2 = (double)n2 * dt * acc;
        const double s2 = linspace_at(fmax(0.0, f_last - r2), fmin(cfg.smax, f_last + r2), ns, i2);
        double d = 0.0;
        pdist[tid][0] = 0.0;
        for (int t = 1; t < NT; ++t) {
            const double sp = t <= n1 ? ramp_at(s0, s1, acc, t, dt) : ramp_at(f_last, s2, acc, t - n1, dt);
            d = d + sp * dt;
            pdist[tid][t] = d;
        }
        double* pf = w.prof + ((size_t)b * MAXPROF + tid) * 3;
        pf[0] = s1; pf[1] = acc; pf[2] = d;
    }
    __syncthreads();
    tick(2);                                          // speed profiles
    double* cc = w.circ + (size_t)b * P * NT * 10;
    for (int it = tid; it < P * NT; it += nthr) {
        const int p = it / NT, t = it % NT;
        double x, y, h;
        double cx[5], cy[5];
        if (!route_pose(R, pdist[p][t], x, y, h)) { flag(status, ST_RANGE); w.scene_bad[b] = 1; x = y = h = NAN; }
        box_circles(x, y, h, el, ew, cx, cy);
        for (int c = 0; c < 5; ++c) { cc[(size_t)it * 10 + 2 * c] = cx[c]; cc[(size_t)it * 10 + 2 * c + 1] = cy[c]; }
    }
    tick(3);                                          // circles
}

// gaps between the ego's profile boxes and one chunk of the scene's predicted trajectories (approx_bbox_distance, :885-897).
// One thread per (profile, time): its ego box (5 circle centres) stays in registers, the chunk's trajectory boxes are staged in
// LDS once (one sincos per trajectory point instead of one per profile), and the thread's running minimum culls every later
// box whose centre is too far to matter -- every circle of a box lies within  m = max(|corner offset| + W/4, W/2)  of its
// centre, so no gap to a box whose centre is D away can be below D - m_e - m_o (an exact cull, the minimum is unchanged).
// Trajectory points marked as bit-for-bit duplicates by the routes kernel are not staged at all.
constexpr int GAP_TJ = 16;          // trajectories staged per round
static size_t gap_lds_bytes(int NT) { return ((size_t)(2 * GAP_TJ * NT * 5 + 2 * GAP_TJ) * 8 + (size_t)GAP_TJ * NT + 15) / 16 * 16; }
__global__ void __launch_bounds__(1024) planner_gap_kernel(StrivePlanner pl, Work w, int k) {
    // staged boxes, sized by the launch's NT (gap_lds_bytes): 14.5 KB at 11 times instead of 41 KB at the cap of 32, so that all
    // (scene, chunk) workgroups of a step are resident together
    HIP_DYNAMIC_SHARED(double, gap_lds)
    __shared__ int cnt;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int P = w.P, NT = w.NT;
    double* const ocx = gap_lds;                        // (GAP_TJ, NT, 5)
    double* const ocy = ocx + GAP_TJ * NT * 5;
    double* const om = ocy + GAP_TJ * NT * 5;           // (GAP_TJ)
    double* const or4 = om + GAP_TJ;
    unsigned char* const ook = reinterpret_cast<unsigned char*>(or4 + GAP_TJ);      // (GAP_TJ, NT)
    const int tid = threadIdx.x, nthr = blockDim.x;
    const bool active = tid < P * NT;
    const int p = active ? tid / NT : 0, t = active ? tid % NT : 0;
    const double* eg = w.ego + 8 * (size_t)b;
    const double ex = eg[0], ey = eg[1];
    const double We = fmin(eg[4], eg[5]), Le = fmax(eg[4], eg[5]);
    const double re4 = We / 4, re2 = We / 2;
    const double k0e = (Le - We) / 2 + We / 4, k1e = We / 4;
    const double m_e = fmax(sqrt(k0e * k0e + k1e * k1e) + We / 4, We / 2);
    double e[10];
    {
        const double* cc = w.circ + ((size_t)b * P * NT + (size_t)p * NT + t) * 10;
#pragma unroll
        for (int c = 0; c < 10; ++c) e[c] = active ? cc[c] : 0.0;
    }
    const int slot = b * w.K + k;
    int J = w.traj_cnt[slot];
    if (J > w.cap) J = w.cap;
    const int per = (J + w.nchunk - 1) / w.nchunk;
    const int j0 = chunk * per, j1 = (j0 + per < J) ? j0 + per : J;
    const double* tb = w.traj + (size_t)slot * w.cap * w.ENT;
    if (tid == 0) cnt = 0;
    double gm = INFINITY;
    if (w.dbg & 8) { if (active) w.part[(((size_t)b * NCHUNK + chunk) * P + p) * NT + t] = e[0]; return; }
    for (int jb = j0; jb < j1; jb += GAP_TJ) {
        const int nj = (j1 - jb) < GAP_TJ ? (j1 - jb) : GAP_TJ;
        __syncthreads();
        // stage: circle centres of (trajectory, time); trajectories beyond the interaction distance and duplicates are left out
        for (int it = tid; it < nj * NT; it += nthr) {
            const int jj = it / NT, tt = it % NT;
            const double* ent = tb + (size_t)(jb + jj) * w.ENT;
            const double ox = ent[0], oy = ent[1], ol = ent[2], ow = ent[3];
            const bool near = !(sqrt((ex - ox) * (ex - ox) + (ey - oy) * (ey - oy)) > pl.cfg.interacdist);
            const bool dup = (((unsigned)ent[4]) >> tt) & 1u;
            if (tt == 0) {
                const double Wo = fmin(ol, ow), Lo = fmax(ol, ow);
                const double k0o = (Lo - Wo) / 2 + Wo / 4, k1o = Wo / 4;
                om[jj] = fmax(sqrt(k0o * k0o + k1o * k1o) + Wo / 4, Wo / 2);
                or4[jj] = Wo / 4;
                if (near) atomicAdd(&cnt, 1);
            }
            ook[jj * NT + tt] = (near && !dup) ? 1 : 0;
            if (near && !dup && !(w.dbg & 4)) box_circles(ent[5 + 3 * tt], ent[6 + 3 * tt], ent[7 + 3 * tt], ol, ow, ocx + (jj * NT + tt) * 5, ocy + (jj * NT + tt) * 5);
        }
        __syncthreads();
        if (active) {
            for (int jj = 0; jj < ((w.dbg & 1) ? 0 : nj); ++jj) {
                if (!ook[jj * NT + t]) continue;
                const double* cx = ocx + (jj * NT + t) * 5;
                const double* cy = ocy + (jj * NT + t) * 5;
                const double cdx = cx[4] - e[8], cdy = cy[4] - e[9];
                const double m22 = cdx * cdx + cdy * cdy;
                if (sqrt(m22) - m_e - om[jj] - 1e-6 > gm) continue;
                double m44 = INFINITY, m42 = INFINITY, m24 = INFINITY;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const double eax = e[2 * a], eay = e[2 * a + 1];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double dx = cx[c] - eax, dy = cy[c] - eay;
                        m44 = fmin(m44, dx * dx + dy * dy);
                    }
                    const double dx = cx[4] - eax, dy = cy[4] - eay;
                    m42 = fmin(m42, dx * dx + dy * dy);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double dx = cx[c] - e[8], dy = cy[c] - e[9];
                    m24 = fmin(m24, dx * dx + dy * dy);
                }
                const double ro4 = or4[jj], ro2 = 2 * ro4;
                double g = sqrt(m44) - re4 - ro4;
                g = fmin(g, sqrt(m42) - re4 - ro2);
                g = fmin(g, sqrt(m24) - re2 - ro4);
                g = fmin(g, sqrt(m22) - re2 - ro2);
                gm = fmin(gm, g);
            }
        }
    }
    if (active) w.part[(((size_t)b * NCHUNK + chunk) * P + p) * NT + t] = gm;
    __syncthreads();
    if (tid == 0) w.part_cnt[b * NCHUNK + chunk] = cnt;
}

__global__ void planner_interp_kernel(Work w, int B, const double* __restrict__ t_out, const double* __restrict__ planner_t, int TP,
                                      double* __restrict__ plan, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * TP) return;
    const int b = i / TP, j = i % TP;
    const double q = planner_t[j];
    double* out = plan + (size_t)i * 4;
    if (q < t_out[0] || q > t_out[w.K - 1]) { flag(status + NSTATUS * (size_t)b, ST_RANGE); out[0] = out[1] = out[2] = out[3] = NAN; return; }
    const int hi = knot_hi(t_out, w.K, q), lo = hi - 1;
    const double* po = w.poses + (size_t)b * w.K * 4;
    for (int c = 0; c < 4; ++c) {
        const double slope = (po[4 * hi + c] - po[4 * lo + c]) / (t_out[hi] - t_out[lo]);
        out[c] = slope * (q - t_out[lo]) + po[4 * lo + c];
    }
}

// alive[b] = no flag of scene b is set -- by this rollout or any earlier one since the caller zeroed the status (the flags are
// sticky).  The optimisation loop masks a scene with alive == 0 out of its losses without a host round trip.
__global__ void planner_alive_kernel(int B, const int32_t* __restrict__ status, uint8_t* __restrict__ alive) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int any = 0;
    for (int i = 0; i < NSTATUS; ++i) any |= status[NSTATUS * (size_t)b + i];
    alive[b] = any == 0;
}

__global__ void __launch_bounds__(64) planner_routes_debug_kernel(StrivePlanner pl, int mapix, const double* __restrict__ pose4, int maxr, int maxk,
                                                                   int32_t* nroutes, int32_t* nk, double* knots, int32_t* status) {
    __shared__ RouteLds R;
    const Pose o = {pose4[0], pose4[1], pose4[2], pose4[3]};
    const int lane = lane_id();
    int base = 0, group = 0;
    auto begin_group = [&](int n) { base += group; group = n; };
    auto on_route = [&](int gi) {
        const int ri = base + gi;
        if (ri >= maxr) return;
        if (lane == 0) nk[ri] = R.bad ? -1 : R.nk;
        if (R.bad) return;
        for (int i = lane; i < R.nk && i < maxk; i += 64) {
            double* kn = knots + ((size_t)ri * maxk + i) * 5;
            kn[0] = R.ks()[i]; kn[1] = R.kx[i]; kn[2] = R.ky[i]; kn[3] = R.khx()[i]; kn[4] = R.khy()[i];
        }
    };
    for_each_route<false>(R, pl.maps[mapix], pl.cfg, o, status, begin_group, on_route);
    if (lane == 0) *nroutes = base + group;
}

struct Layout {
    size_t total;
    Work w;
};

Layout carve(const StrivePlanner* pl, int nstep, int traj_cap, void* ws, size_t ws_bytes) {
    Layout L;
    Work& w = L.w;
    const StrivePlannerCfg& c = pl->cfg;
    w.K = nstep + 1;
    w.cap = traj_cap;
    w.NT = c.nsteps + 1;
    w.ENT = 5 + 3 * w.NT;
    w.P = c.nplanaccfacs * c.plannspeeds * c.plannspeeds;
    StriveArena a(ws ? ws : (void*)64, ws ? ws_bytes : (size_t)-1 / 2);
    w.wstate = a.take<double>((size_t)pl->NR * w.K * 4);
    w.present = a.take<int8_t>((size_t)pl->NR * w.K);
    w.traj = a.take<double>((size_t)pl->B * w.K * w.cap * w.ENT);
    // small integer state, zeroed by ONE memset at the start of a rollout: trajectory counts, scene flags, route lengths
    // (a scene that is given up keeps route length 0 instead of whatever the workspace held), stop preferences, chunk counts
    w.nzero = (size_t)pl->B * w.K + 3 * (size_t)pl->B + (size_t)pl->B * NCHUNK;
    w.traj_cnt = a.take<int32_t>(w.nzero);
    w.ego = a.take<double>((size_t)pl->B * 8);
    w.route = a.take<double>((size_t)pl->B * 5 * MAXK);
    w.scene_bad = w.traj_cnt ? w.traj_cnt + (size_t)pl->B * w.K : nullptr;
    w.route_nk = w.traj_cnt ? w.scene_bad + pl->B : nullptr;
    w.prefer_stop = w.traj_cnt ? w.route_nk + pl->B : nullptr;
    w.part_cnt = w.traj_cnt ? w.prefer_stop + pl->B : nullptr;
    w.prof = a.take<double>((size_t)pl->B * MAXPROF * 3);
    w.circ = a.take<double>((size_t)pl->B * w.P * w.NT * 10);
    w.part = a.take<double>((size_t)pl->B * NCHUNK * w.P * w.NT);
    w.poses = a.take<double>((size_t)pl->B * w.K * 4);
    L.total = a.ok() ? a.off : 0;
    return L;
}

int check_cfg(const StrivePlanner* pl, int nstep, int traj_cap) {
    const StrivePlannerCfg& c = pl->cfg;
    STRIVE_CHECK_ARG(pl->nmaps >= 1 && pl->nmaps <= STRIVE_PLANNER_MAXMAPS, "1..4 maps");
    STRIVE_CHECK_ARG(c.nsteps >= 2 && c.nsteps + 1 <= MAXNT, "nsteps + 1 must be <= 32");
    STRIVE_CHECK_ARG(c.npredsfacs >= 1 && c.npredafacs >= 1 && c.npredsfacs <= 4 && c.npredafacs <= 4 &&
                     c.npredsfacs * c.npredafacs <= MAXPRED, "at most 8 predicted speed profiles");
    STRIVE_CHECK_ARG(c.nplanaccfacs >= 1 && c.nplanaccfacs <= 4 && c.plannspeeds >= 1 &&
                     c.nplanaccfacs * c.plannspeeds * c.plannspeeds <= MAXPROF, "at most 64 ego speed profiles");
    STRIVE_CHECK_ARG(c.nplanaccfacs * c.plannspeeds * c.plannspeeds * (c.nsteps + 1) <= 1024,
                     "profiles x times exceed one workgroup of the gap kernel");
    STRIVE_CHECK_ARG(nstep >= 0 && traj_cap >= 1, "nstep >= 0, traj_cap >= 1");
    STRIVE_CHECK_ARG(pl->B >= 1 && pl->NR >= 0 && pl->NO == pl->NR + pl->B, "one ego per scene");
    return 0;
}

}  // namespace strive_planner
using namespace strive_planner;

extern "C" size_t strive_planner_workspace_bytes(const StrivePlanner* pl, int32_t nstep, int32_t traj_cap) {
    if (check_cfg(pl, nstep, traj_cap)) return 0;
    return carve(pl, nstep, traj_cap, nullptr, 0).total + 256;
}

extern "C" int strive_planner_rollout(const StrivePlanner* pl, const double* agent_obs, const double* agent_t, int32_t T,
                                      const double* t_out, int32_t nstep, const double* planner_t, int32_t TP, int32_t traj_cap,
                                      double* plan, int32_t* status, uint8_t* alive, void* ws, size_t ws_bytes,
                                      strive_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (check_cfg(pl, nstep, traj_cap)) return -1;
    STRIVE_CHECK_ARG(plan && status && ws && t_out && planner_t && TP >= 1, "null argument");
    STRIVE_CHECK_ARG(pl->NR == 0 || (agent_obs && agent_t && T >= 1), "agent_obs / agent_t missing");
    Layout L = carve(pl, nstep, traj_cap, ws, ws_bytes);
    STRIVE_CHECK_ARG(L.total != 0, "workspace too small");
    Work& w = L.w;
    w.tprof = nullptr;
    w.dbg = strive_tuning().planner_dbg;
    if (strive_tuning().planner_prof) {      // (measurement only: the last 64 bytes of the workspace's spare tail, accumulated over rollouts)
        w.tprof = reinterpret_cast<unsigned long long*>((char*)ws + ((L.total + 7) / 8) * 8 + 64);
    }
    hipMemsetAsync(w.traj_cnt, 0, sizeof(int32_t) * w.nzero, stream);
    if (pl->NR > 0) {
        hipLaunchKernelGGL(planner_world_kernel, dim3((pl->NR + 63) / 64), dim3(64), 0, stream, *pl, w, agent_obs, agent_t, (int)T);
        hipLaunchKernelGGL(planner_routes_kernel, dim3(pl->NR * w.K), dim3(64), 0, stream, *pl, w, status);
    }
    const int gap_threads = ((w.P * w.NT + 63) / 64) * 64;
    // one thread per (profile, time) of the risk scores and the circles where that fits (25 x 26 -> 704: one pass)
    const int ego_threads = gap_threads < 256 ? 256 : gap_threads;
    // chunks per scene: as many as keep ALL (scene, chunk) workgroups of a step resident at once -- a workgroup of P x NT threads
    // is 11 waves at the default 25 x 26, two of them fit a CU, 512 the chip; 36 scenes x 16 chunks = 576 workgroups ran as one full
    // round and a second one of 64
    {
        const int waves = gap_threads / 64, per_cu = 32 / waves < 1 ? 1 : 32 / waves;
        const int nc = (256 * per_cu) / (int)pl->B;
        w.nchunk = nc < 4 ? 4 : (nc > NCHUNK ? NCHUNK : nc);
    }
    for (int k = 0; k <= w.K; ++k) {
        hipLaunchKernelGGL(planner_ego_kernel, dim3(pl->B), dim3(ego_threads), 0, stream, *pl, w, k, status);
        if (k < w.K)
            hipLaunchKernelGGL(planner_gap_kernel, dim3(pl->B, w.nchunk), dim3(gap_threads), gap_lds_bytes(w.NT), stream, *pl, w, k);
    }
    hipLaunchKernelGGL(planner_interp_kernel, dim3((pl->B * TP + 63) / 64), dim3(64), 0, stream, w, (int)pl->B, t_out, planner_t, (int)TP,
                       plan, status);
    if (alive) hipLaunchKernelGGL(planner_alive_kernel, dim3((pl->B + 63) / 64), dim3(64), 0, stream, (int)pl->B, (const int32_t*)status, alive);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_planner_routes(const StrivePlanner* pl, int32_t mapix, const double* pose4, int32_t maxr, int32_t maxk,
                                     int32_t* nroutes, int32_t* nk, double* knots, int32_t* status, strive_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    STRIVE_CHECK_ARG(pl && mapix >= 0 && mapix < pl->nmaps && pose4 && nroutes && nk && knots && status, "bad argument");
    hipLaunchKernelGGL(planner_routes_debug_kernel, dim3(1), dim3(64), 0, stream, *pl, (int)mapix, pose4, (int)maxr, (int)maxk, nroutes, nk,
                       knots, status);
    STRIVE_CHECK_LAUNCH();
    return 0;
}
