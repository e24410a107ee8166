// Decoder rollout: forward over FT autoregressive steps and the explicit reverse-time backward to dL/dz.
// (reference src/models/traffic_model.py:589-704; gradient dependency map: SURVEY.md Appendix A)
//
// Per step t the forward enqueues
//   node1  : [past_feat_t | map_feat_t | sem | z | lw] -> mlp_in -> x, edge partials P, Q
//   edge   : per target node, factorised edge MLP over its scene, running max / arg-max
//   node2r : update MLP -> mlp_out -> (a, ddh) -> kinematic bicycle step -> local-frame pose -> 3-layer GRU
//   map CNN: fused crop + 6 MFMA convolutions at the new (detached) poses   [t < FT-1]
// and writes every step's inputs into the tape, so the backward re-derives activations by recomputation
// (cheap: the GNN/GRU are ~1.6 MFLOP per agent-step against 300 MFLOP for the CNN, which needs no backward
// at all because the reference crops at pos.detach()).  The backward is four launches per step:
// node1 (recompute), gru_bwd, node2_bwd, edge_bwd, node1_bwd.
#include <atomic>
#include "gnn_bwd_kernels.h"

struct DynParams {
    float smean[6], sstd[6];
    float amean0, astd0;     // vehicle length normaliser (l)
    float a_mean, a_std, ddh_mean, ddh_std, dt, max_hdot, max_s;
};

struct GRUDev {
    const float* wih[3];
    const float* whh[3];
    const float* wih_t[3];
    const float* whh_t[3];
    const float* bih[3];
    const float* bhh[3];
};

// Weight gradients of the GRU memory (training path): flat buffer in named_parameters() order, per layer
// weight_ih (192, in_l) | weight_hh (192, 64) | bias_ih (192) | bias_hh (192); null = no weight gradients.
struct GRUGradDev {
    float* wih[3];
    float* whh[3];
    float* bih[3];
    float* bhh[3];
    bool on;
    WJobTable* jobs;        // deferred weight gradients (see mlp_dev.h), or null
};

static inline size_t gru_param_count() {
    size_t n = 0;
    for (int l = 0; l < 3; ++l) n += (size_t)192 * (l == 0 ? 4 : 64) + (size_t)192 * 64 + 2 * 192;
    return n;
}

static inline GRUGradDev gru_grad_dev(float* flat) {
    GRUGradDev d;
    d.on = flat != nullptr;
    d.jobs = nullptr;
    float* p = flat;
    for (int l = 0; l < 3; ++l) {
        const int in = l == 0 ? 4 : 64;
        d.wih[l] = p; if (p) p += (size_t)192 * in;
        d.whh[l] = p; if (p) p += (size_t)192 * 64;
        d.bih[l] = p; if (p) p += 192;
        d.bhh[l] = p; if (p) p += 192;
    }
    return d;
}

struct Tape {
    float *pf, *mf, *pos, *state, *mem, *A, *loc;
    float *X, *P, *Q;       // node embeddings and edge layer-0 partials of every step (kept: the reverse sweep reads them
                            // instead of re-running the node-1 kernel)
    // pre-LayerNorm outputs of every hidden layer, decoder outputs and GRU gates of every step: what the reverse sweep used
    // to recompute (5 + 3 + 2 dense layers per node, 2 per edge, 6 GRU products).  1.9 KB per node-step + 1 KB per
    // edge-step: 150 MB for the 512-agent, 16-step closure -- nothing next to 288 GB, and a third of the backward's time
    float *PRE_IN, *PRE_E, *PRE_U, *PRE_O, *DEC, *GATES;
    int32_t* ARG;
    size_t R;
    int max_n;
    __host__ __device__ float* PRE_IN_t(int t) const { return PRE_IN + (size_t)t * R * 2 * STRIVE_HID; }
    __host__ __device__ float* PRE_E_t(int t) const { return PRE_E + (size_t)t * R * max_n * 2 * STRIVE_HID; }
    __host__ __device__ float* PRE_U_t(int t) const { return PRE_U + (size_t)t * R * STRIVE_HID; }
    __host__ __device__ float* PRE_O_t(int t) const { return PRE_O + (size_t)t * R * 2 * STRIVE_HID; }
    __host__ __device__ float* DEC_t(int t) const { return DEC + (size_t)t * R * 4; }
    __host__ __device__ float* GATES_t(int t) const { return GATES + (size_t)t * R * 3 * 256; }
    __host__ __device__ float* X_t(int t) const { return X + (size_t)t * R * 64; }
    __host__ __device__ float* P_t(int t) const { return P + (size_t)t * R * STRIVE_HID; }
    __host__ __device__ float* Q_t(int t) const { return Q + (size_t)t * R * STRIVE_HID; }
    __host__ __device__ float* pf_t(int t) const { return pf + (size_t)t * R * 64; }
    __host__ __device__ float* mf_t(int t) const { return mf + (size_t)t * R * 64; }
    __host__ __device__ float* pos_t(int t) const { return pos + (size_t)t * R * 4; }
    __host__ __device__ float* state_t(int t) const { return state + (size_t)t * R * 8; }
    __host__ __device__ float* mem_t(int t) const { return mem + (size_t)t * R * 192; }
    __host__ __device__ float* A_t(int t) const { return A + (size_t)t * R * 64; }
    __host__ __device__ float* loc_t(int t) const { return loc + (size_t)t * R * 4; }
    __host__ __device__ int32_t* ARG_t(int t) const { return ARG + (size_t)t * R * 64; }
};

static size_t tape_bytes_for(size_t R, int FT, int max_n) {
    const size_t per = 64 + 64 + 4 + 8 + 192 + 64 + 4 + 64 + 64 + 2 * STRIVE_HID +
                       2 * STRIVE_HID + (size_t)max_n * 2 * STRIVE_HID + STRIVE_HID + 2 * STRIVE_HID + 4 + 3 * 256;
    return strive_align_up(R * FT * per * 4 + 17 * 256, 256);
}

static Tape carve_tape(void* p, size_t bytes, size_t R, int FT, int max_n) {
    StriveArena ar(p, bytes);
    Tape t;
    t.R = R;
    t.max_n = max_n;
    t.pf = ar.take<float>(R * FT * 64);
    t.mf = ar.take<float>(R * FT * 64);
    t.pos = ar.take<float>(R * FT * 4);
    t.state = ar.take<float>(R * FT * 8);
    t.mem = ar.take<float>(R * FT * 192);
    t.A = ar.take<float>(R * FT * 64);
    t.loc = ar.take<float>(R * FT * 4);
    t.ARG = ar.take<int32_t>(R * FT * 64);
    t.X = ar.take<float>(R * FT * 64);
    t.P = ar.take<float>(R * FT * STRIVE_HID);
    t.Q = ar.take<float>(R * FT * STRIVE_HID);
    t.PRE_IN = ar.take<float>(R * FT * 2 * STRIVE_HID);
    t.PRE_E = ar.take<float>(R * FT * (size_t)max_n * 2 * STRIVE_HID);
    t.PRE_U = ar.take<float>(R * FT * STRIVE_HID);
    t.PRE_O = ar.take<float>(R * FT * 2 * STRIVE_HID);
    t.DEC = ar.take<float>(R * FT * 4);
    t.GATES = ar.take<float>(R * FT * 3 * 256);
    return t;
}

// ---------------------------------------------------------------------------------------------
// bicycle step on one row (reference src/models/common.py:47-68, src/utils/transforms.py:8-29,
// src/models/traffic_model.py:645-650) -- forward values + the partials the backward needs.
// ---------------------------------------------------------------------------------------------
struct BikeFwd {
    float su[6];        // unnormalised input state
    float h, nh, ns, nhd, sn, cs, len;
    float pre_s, pre_hd;
    float out[6];       // normalised output state
};

__device__ __forceinline__ void bike_forward(const DynParams& p, const float* st, float dec0, float dec1, float lw0, BikeFwd& b) {
    for (int i = 0; i < 6; ++i) b.su[i] = unnorm1(st[i], p.smean[i], p.sstd[i]);
    const float a = __fadd_rn(__fmul_rn(dec0, p.a_std), p.a_mean);
    const float ddh = __fadd_rn(__fmul_rn(dec1, p.ddh_std), p.ddh_mean);
    b.len = unnorm1(lw0, p.amean0, p.astd0);
    b.h = atan2f(b.su[3], b.su[2]);
    b.pre_hd = b.su[5] + ddh * p.dt;
    b.nhd = fminf(fmaxf(b.pre_hd, -p.max_hdot), p.max_hdot);
    b.nh = b.h + p.dt * fabsf(b.su[4]) / b.len * b.nhd;
    b.pre_s = b.su[4] + a * p.dt;
    b.ns = fminf(fmaxf(b.pre_s, 0.0f), p.max_s);
    b.sn = sinf(b.nh);
    b.cs = cosf(b.nh);
    const float ny = b.su[1] + b.ns * b.sn * p.dt;
    const float nx = b.su[0] + b.ns * b.cs * p.dt;
    const float u[6] = {nx, ny, b.cs, b.sn, b.ns, b.nhd};
    for (int i = 0; i < 6; ++i) b.out[i] = norm1(u[i], p.smean[i], p.sstd[i]);
}

// adjoint: gout (6, w.r.t. normalised output) -> gst (6, w.r.t. normalised input state), gdec (2)
__device__ __forceinline__ void bike_backward(const DynParams& p, const BikeFwd& b, const float* gout, float* gst, float* gdec) {
    float gu[6];
    for (int i = 0; i < 6; ++i) gu[i] = gout[i] / p.sstd[i];
    const float g_nx = gu[0], g_ny = gu[1];
    float g_nh = gu[2] * (-b.sn) + gu[3] * b.cs + g_nx * (-b.ns * b.sn * p.dt) + g_ny * (b.ns * b.cs * p.dt);
    float g_ns = gu[4] + g_nx * b.cs * p.dt + g_ny * b.sn * p.dt;
    const float m_s = (b.pre_s >= 0.0f && b.pre_s <= p.max_s) ? 1.f : 0.f;
    g_ns *= m_s;
    float g_s = g_ns;
    const float g_a = g_ns * p.dt;
    // new_h = h + (dt*|s|/len) * new_hdot
    const float coef = p.dt * fabsf(b.su[4]) / b.len;
    const float g_h = g_nh;
    const float sgn = (b.su[4] > 0.f) ? 1.f : ((b.su[4] < 0.f) ? -1.f : 0.f);
    g_s += g_nh * (p.dt / b.len) * b.nhd * sgn;
    float g_nhd = gu[5] + g_nh * coef;
    const float m_h = (b.pre_hd >= -p.max_hdot && b.pre_hd <= p.max_hdot) ? 1.f : 0.f;
    g_nhd *= m_h;
    const float g_hd = g_nhd;
    const float g_ddh = g_nhd * p.dt;
    const float den = b.su[2] * b.su[2] + b.su[3] * b.su[3];
    const float g_hx = g_h * (-b.su[3] / den);
    const float g_hy = g_h * (b.su[2] / den);
    const float gsu[6] = {g_nx, g_ny, g_hx, g_hy, g_s, g_hd};
    for (int i = 0; i < 6; ++i) gst[i] = gsu[i] * p.sstd[i];
    gdec[0] = g_a * p.a_std;
    gdec[1] = g_ddh * p.ddh_std;
}

// ---------------------------------------------------------------------------------------------
// GRU helpers (one time step, 3 layers, hidden 64; torch.nn.GRU gate order r,z,n)
// ---------------------------------------------------------------------------------------------
#define GLD 192
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// x: LDS [RB_NODE][xld] (layer input), h: LDS [RB_NODE][64] (layer hidden), gi/gh: LDS [RB_NODE][GLD] scratch.
// Writes the new hidden into hn [RB_NODE][64]; if gates != null stores (r,z,n,gh_n) at gates[RB_NODE][4*64].
// ggates (optional, global [RB_NODE rows][256] of THIS layer, row stride gstride floats): the same four values kept for the
// reverse sweep; rows >= nrows are not written.
__device__ __forceinline__ void gru_layer_lds(const GRUDev& g, int l, const float* x, int xld, int xin, const float* h,
                                              float* gi, float* gh, float* hn, float* gates, int tid,
                                              float* ggates = nullptr, int gstride = 0, int nrows = 0) {
    dense_lds<RB_NODE, false>(x, xld, xin, g.wih_t[l], GLD, g.bih[l], gi, GLD, GLD, tid, 256);
    dense_lds<RB_NODE, false>(h, 64, 64, g.whh_t[l], GLD, g.bhh[l], gh, GLD, GLD, tid, 256);
    __syncthreads();
    for (int i = tid; i < RB_NODE * 64; i += 256) {
        const int rr = i >> 6, c = i & 63;
        const float r = sigmoidf_(gi[rr * GLD + c] + gh[rr * GLD + c]);
        const float z = sigmoidf_(gi[rr * GLD + 64 + c] + gh[rr * GLD + 64 + c]);
        const float ghn = gh[rr * GLD + 128 + c];
        const float n = tanhf(gi[rr * GLD + 128 + c] + r * ghn);
        hn[rr * 64 + c] = (1.0f - z) * n + z * h[rr * 64 + c];
        if (gates) {
            gates[rr * 256 + c] = r;
            gates[rr * 256 + 64 + c] = z;
            gates[rr * 256 + 128 + c] = n;
            gates[rr * 256 + 192 + c] = ghn;
        }
        if (ggates && rr < nrows) {
            float* gq = ggates + (size_t)rr * gstride;
            gq[c] = r;
            gq[64 + c] = z;
            gq[128 + c] = n;
            gq[192 + c] = ghn;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// forward step kernel: node2 + dynamics + GRU.   grid = ceil(R/RB_NODE)
// ---------------------------------------------------------------------------------------------
struct StepArgs {
    int t, FT, R, NS;
    const float* X;            // (R, 64) from node1
    const float* sem;          // (NA, NC)
    const float* lw;           // (NA, 2) normalised
    const float* ext;          // (B, FT, 4) or null
    const int32_t* ptr;        // scene offsets (ego = first agent of a scene)
    const int32_t* scene_of;
    float* traj;               // (R, FT, 4)
};

static __global__ __launch_bounds__(256) void rollout_node2_kernel(GNNDev g, GRUDev gru, DynParams dp, StepArgs a, Tape tp) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int in_ld = ld4(2 * g.D + g.NC);
    Node2Lds L(smem, in_ld);
    float* s_x = L.out + RB_NODE * HLD;        // [RB_NODE][4]  GRU input (local pose)
    float* s_h = s_x + RB_NODE * 4;            // [RB_NODE][64]    hidden input of the current layer
    float* s_hn = s_h + RB_NODE * 64;          // [2][RB_NODE][64] layer outputs, ping-pong (layer l+1 reads layer l's as input)
    float* s_gi = s_hn + 2 * RB_NODE * 64;     // [RB_NODE][GLD]
    float* s_gh = s_gi + RB_NODE * GLD;        // [RB_NODE][GLD]
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE, t = a.t;
    node2_forward(g, a.NS, a.X, tp.A_t(t), a.sem, r0, a.R, L, in_ld, tid);
    const bool more = t < a.FT - 1;
    // keep the pre-activations of update (1 layer) and mlp_out (2 layers) for the reverse sweep
    for (int i = tid; i < RB_NODE * 3 * STRIVE_HID; i += 256) {
        const int rr = i / (3 * STRIVE_HID), rem = i - rr * 3 * STRIVE_HID;
        if (r0 + rr < a.R) {
            if (rem < STRIVE_HID) tp.PRE_U_t(t)[(size_t)(r0 + rr) * STRIVE_HID + rem] = L.pre_u[rr * HLD + rem];
            else {
                const int q = rem - STRIVE_HID, l = q / STRIVE_HID, c = q - l * STRIVE_HID;
                tp.PRE_O_t(t)[(size_t)(r0 + rr) * 2 * STRIVE_HID + q] = L.pre_o[(size_t)l * RB_NODE * HLD + rr * HLD + c];
            }
        }
    }
    if (tid < RB_NODE) {
        const int r = r0 + tid;
        float loc[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < a.R) {
            const int ag = r / a.NS;
            const float* st = tp.state_t(t) + (size_t)r * 8;
            BikeFwd b;
            bike_forward(dp, st, L.out[tid * HLD + 0], L.out[tid * HLD + 1], a.lw[ag * 2], b);
            tp.DEC_t(t)[(size_t)r * 4 + 0] = L.out[tid * HLD + 0];
            tp.DEC_t(t)[(size_t)r * 4 + 1] = L.out[tid * HLD + 1];
            float* tr = a.traj + ((size_t)r * a.FT + t) * 4;
            for (int i = 0; i < 4; ++i) tr[i] = b.out[i];
            float gin[4] = {b.out[0], b.out[1], b.out[2], b.out[3]};
            if (a.ext) {
                const int sc = a.scene_of[ag];
                if (a.ptr[sc] == ag) {
                    const float* e = a.ext + ((size_t)sc * a.FT + t) * 4;
                    for (int i = 0; i < 4; ++i) gin[i] = e[i];
                }
            }
            rel_pose(st, gin, loc);
            float* lo = tp.loc_t(t) + (size_t)r * 4;
            for (int i = 0; i < 4; ++i) lo[i] = loc[i];
            if (more) {
                float* ns = tp.state_t(t + 1) + (size_t)r * 8;
                for (int i = 0; i < 6; ++i) ns[i] = b.out[i];
                float* np = tp.pos_t(t + 1) + (size_t)r * 4;
                for (int i = 0; i < 4; ++i) np[i] = gin[i];
            }
        }
        for (int i = 0; i < 4; ++i) s_x[tid * 4 + i] = loc[i];
    }
    if (!more) return;
    // ---- GRU memory step (reference traffic_model.py:684-688) ----
    const float* x = s_x;
    int xld = 4, xin = 4;
    for (int l = 0; l < 3; ++l) {
        __syncthreads();
        for (int i = tid; i < RB_NODE * 64; i += 256) {
            const int rr = i >> 6, c = i & 63;
            const int r = r0 + rr;
            s_h[i] = (r < a.R) ? tp.mem_t(t)[((size_t)r * 3 + l) * 64 + c] : 0.f;
        }
        __syncthreads();
        float* hn = s_hn + (size_t)(l & 1) * RB_NODE * 64;
        const int nrows = (a.R - r0) < RB_NODE ? (a.R - r0) : RB_NODE;
        gru_layer_lds(gru, l, x, xld, xin, s_h, s_gi, s_gh, hn, nullptr, tid, tp.GATES_t(t) + ((size_t)r0 * 3 + l) * 256, 3 * 256,
                      nrows);
        for (int i = tid; i < RB_NODE * 64; i += 256) {
            const int rr = i >> 6, c = i & 63;
            const int r = r0 + rr;
            if (r < a.R) {
                tp.mem_t(t + 1)[((size_t)r * 3 + l) * 64 + c] = hn[i];
                if (l == 2) tp.pf_t(t + 1)[(size_t)r * 64 + c] = hn[i];
            }
        }
        x = hn;
        xld = 64;
        xin = 64;
    }
}

// ---------------------------------------------------------------------------------------------
// small helper kernels
// ---------------------------------------------------------------------------------------------
static __global__ void rollout_init_kernel(Tape tp, const float* __restrict__ past_last, const float* __restrict__ past_feat,
                                           const float* __restrict__ map_feat, const int32_t* __restrict__ mapix,
                                           int32_t* __restrict__ mapix_rows, int R, int NS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * 64) return;
    const int r = i >> 6, c = i & 63, ag = r / NS;
    const float pf = past_feat[(size_t)ag * 64 + c];
    tp.pf_t(0)[i] = pf;
    tp.mf_t(0)[i] = map_feat[(size_t)ag * 64 + c];
    for (int l = 0; l < 3; ++l) tp.mem_t(0)[((size_t)r * 3 + l) * 64 + c] = pf;
    if (c < 6) tp.state_t(0)[(size_t)r * 8 + c] = past_last[(size_t)ag * 6 + c];
    if (c < 4) tp.pos_t(0)[(size_t)r * 4 + c] = past_last[(size_t)ag * 6 + c];
    if (c == 0) mapix_rows[r] = mapix[ag];
}

#include "scene_rollout.h"

// =============================================================================================
// host orchestration: forward
// =============================================================================================
namespace {

// The scene-resident kernels (scene_rollout.h) serve single-sample rollouts of scenes with <= 16 agents whose weight packs
// carry matrix-core fragments; option scene_kernels = 0 keeps the launch-per-phase kernels (A/B measurements).
bool scene_kernels_on(const StriveDecoder* dec, const StriveScenes* sc) {
    return strive_tuning().scene_kernels != 0 && scn::supported(*dec, *sc);   // (read per call: tests and A/B runs switch it inside one process)
}

// Batches with scenes of more than 16 agents (single-sample): the node-level phases of the FORWARD step on the scene kernel in
// 16-row tiles (option scene_tiles, default 1), the edge rows and the reverse sweep on the launch-per-phase kernels.
bool scene_tiles_on(const StriveDecoder* dec, const StriveScenes* sc) {
    return strive_tuning().scene_kernels != 0 && strive_tuning().scene_tiles != 0 && sc->max_n > scn::NR && scn::supported(*dec, *sc, true);
}

// more than 64 KB of LDS per workgroup needs the attribute, once per device
int scene_kernels_prepare() {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (done.load(std::memory_order_acquire) & (1ull << dev)) return 0;
    if (hipFuncSetAttribute((const void*)scn::scene_fwd_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scn::FwdLds::BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)scn::scene_fwd_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scn::FwdLds::BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)scn::scene_bwd_sweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scn::BwdLds::BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)scn::scene_bwd_sweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scn::BwdLds::BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)(scn::scene_bwd_sweep_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scn::BwdLds::BYTES) != hipSuccess) {
        strive_set_error("rollout: the scene kernels' LDS request was refused");
        return -1;
    }
    done.fetch_or(1ull << dev, std::memory_order_release);
    return 0;
}

struct FwdWs {
    GnnBuffers gb;       // X, P, Q (A/ARG live in the tape)
    int32_t* mapix_rows;
    void* cnn_ws;
    size_t cnn_bytes;
};

// elements of each of the two partial-maxima arrays the scene kernels' K-workgroup edge walk writes: (B, K <= 4, NR, 64), and none
// when the batch can never run on the scene kernels (multi-sample, > 16 agents per scene, weights without fragments).  Sized from
// what the batch supports, not from the options of the moment, so that an option set between the size query and the call cannot
// outgrow the workspace.
size_t fwd_part_elems(const StriveDecoder* dec, const StriveScenes* sc) {
    return dec && sc && scn::supported(*dec, *sc) ? (size_t)sc->B * 4 * scn::NR * 64 : 0;
}

size_t fwd_ws_bytes(size_t R, size_t part_elems) {
    size_t b = 0;
    b += strive_align_up(R * 64 * 4, 256);
    b += 2 * strive_align_up(R * STRIVE_HID * 4, 256);
    b += strive_align_up(R * 4, 256);
    b += strive_align_up(strive_map_cnn_workspace_bytes((int32_t)R), 256);
    b += 2 * strive_align_up(part_elems * 4, 256);
    return b;
}

DynParams dyn_params(const StriveDecoder& d) {
    DynParams p;
    for (int i = 0; i < 6; ++i) { p.smean[i] = d.state_mean[i]; p.sstd[i] = d.state_std[i]; }
    p.amean0 = d.att_mean[0];
    p.astd0 = d.att_std[0];
    p.a_mean = d.a_mean; p.a_std = d.a_std; p.ddh_mean = d.ddh_mean; p.ddh_std = d.ddh_std;
    p.dt = d.dt; p.max_hdot = d.max_hdot; p.max_s = d.max_s;
    return p;
}

GRUDev gru_dev(const StriveGRU& g) {
    GRUDev d;
    for (int l = 0; l < 3; ++l) {
        d.wih[l] = g.wih[l]; d.whh[l] = g.whh[l]; d.wih_t[l] = g.wih_t[l]; d.whh_t[l] = g.whh_t[l];
        d.bih[l] = g.bih[l]; d.bhh[l] = g.bhh[l];
    }
    return d;
}

FeatSrc decoder_features(const Tape& tp, int t, const float* sem, const float* z, const float* lw, int NC) {
    FeatSrc f;
    f.n = 5;
    f.p[0] = tp.pf_t(t); f.w[0] = 64; f.per_agent[0] = 0;
    f.p[1] = tp.mf_t(t); f.w[1] = 64; f.per_agent[1] = 0;
    f.p[2] = sem;        f.w[2] = NC; f.per_agent[2] = 1;
    f.p[3] = z;          f.w[3] = STRIVE_ZDIM; f.per_agent[3] = 0;
    f.p[4] = lw;         f.w[4] = 2;  f.per_agent[4] = 1;
    return f;
}

size_t node2r_lds_bytes(int in_ld) { return (Node2Lds::floats(in_ld) + RB_NODE * 4 + 3 * RB_NODE * 64 + 2 * RB_NODE * GLD) * 4; }

int check_decoder(const StriveDecoder* dec, const StriveScenes* sc, int FT) {
    if (gnn_check(dec->gnn)) return -1;
    if (dec->gnn.D != 64 || dec->gnn.mlp_out.dims[3] != 2) { strive_set_error("rollout: decoder_net must have D=64 and 2 outputs"); return -1; }
    if (dec->gnn.mlp_in.dims[0] != 64 + 64 + dec->gnn.NC + STRIVE_ZDIM + 2) { strive_set_error("rollout: decoder_net input width mismatch"); return -1; }
    if (FT < 1 || sc->NA < 0 || sc->NS < 1) { strive_set_error("rollout: bad sizes"); return -1; }
    return 0;
}

}  // namespace

extern "C" int strive_rollout_scene_resident(const StriveDecoder* dec, const StriveScenes* sc) {
    if (!dec || !sc) return 0;
    return scene_kernels_on(dec, sc) ? 1 : (scene_tiles_on(dec, sc) ? 2 : 0);
}

extern "C" size_t strive_rollout_tape_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT) {
    if (!sc) return 0;
    return tape_bytes_for((size_t)sc->NA * sc->NS, FT, sc->max_n > 0 ? sc->max_n : 1);
}

extern "C" size_t strive_rollout_keep_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT) {
    (void)dec;
    if (!sc || FT < 2) return strive_map_cnn_keep_bytes(0);
    return strive_map_cnn_keep_bytes((int32_t)((size_t)(FT - 1) * sc->NA * sc->NS));
}

// `kept` (optional): the map CNN's activations of the FT - 1 re-encoded steps are kept for strive_rollout_bwd_train_kept
// (rows (t - 1) R .. t R of the kept arrays = the crops at pos_t, t = 1 .. FT - 1: the order strive_rollout_bwd_train walks them in)
static int rollout_forward(const StriveDecoder* dec, const StriveScenes* sc, const float* past_last,
                           const float* lw, const float* sem, const float* past_feat, const float* map_feat,
                           const float* z, const int32_t* mapix, const float* ext_future, int32_t FT, float* traj,
                           void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* kept, size_t kept_bytes,
                           strive_stream_t stream_) {
    STRIVE_CHECK_ARG(dec && sc && past_last && lw && sem && past_feat && map_feat && z && mapix && traj && tape && ws,
                     "null argument");
    if (check_decoder(dec, sc, FT)) return -1;
    STRIVE_CHECK_ARG(!kept || kept_bytes >= strive_rollout_keep_bytes(dec, sc, FT), "kept-activation buffer too small");
    STRIVE_CHECK_ARG(!(ext_future && sc->NS != 1), "ext_future with multiple samples is not supported");
    const size_t R = (size_t)sc->NA * sc->NS;
    if (R == 0) return 0;
    STRIVE_CHECK_ARG(tape_bytes >= tape_bytes_for(R, FT, sc->max_n > 0 ? sc->max_n : 1), "tape too small");
    STRIVE_CHECK_ARG(ws_bytes >= strive_rollout_workspace_bytes(dec, sc, FT), "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    Tape tp = carve_tape(tape, tape_bytes, R, FT, sc->max_n > 0 ? sc->max_n : 1);
    StriveArena ar(ws, ws_bytes);
    FwdWs w;
    w.gb.X = ar.take<float>(R * 64);
    w.gb.P = ar.take<float>(R * STRIVE_HID);
    w.gb.Q = ar.take<float>(R * STRIVE_HID);
    w.mapix_rows = ar.take<int32_t>(R);
    w.cnn_bytes = strive_map_cnn_workspace_bytes((int32_t)R);
    w.cnn_ws = ar.take<char>(w.cnn_bytes);
    const size_t part_n = fwd_part_elems(dec, sc);
    float* part_a = ar.take<float>(part_n);
    int32_t* part_arg = ar.take<int32_t>(part_n);
    STRIVE_CHECK_ARG(ar.ok(), "workspace arena overflow");

    const GNNDev gd = gnn_dev(dec->gnn);
    const GRUDev gr = gru_dev(dec->gru);
    const DynParams dp = dyn_params(*dec);
    const ScenesDev sd = scenes_dev(*sc);
    const int NC = dec->gnn.NC;
    const int in_ld1 = ld4(dec->gnn.mlp_in.dims[0]), xs_ld = ld4(64 + NC), in_ld2 = ld4(128 + NC);
    const int nb = (int)((R + RB_NODE - 1) / RB_NODE);

    const bool scene = scene_kernels_on(dec, sc);
    const bool scene_tiles = !scene && scene_tiles_on(dec, sc);
    if ((scene || scene_tiles) && scene_kernels_prepare()) return -1;
    const scn::GRUFrag gf = scn::gru_frag(dec->gru);
    const bool scene_prof = scene && strive_tuning().scene_prof != 0;
    // scenes of >= option scene_split agents (default 12 = three 64-row edge chunks; 0 = never): their 130-240 edge rows are 3-4
    // chunks on ONE CU inside the one-launch scene step.  Round 4 moved them to gnn_edge_kernel (one workgroup per target) from 15
    // agents on (profiles/r04_ab_scene_split.txt); round 5 shares the chunks among K workgroups of the scene kernel itself (below):
    // a wash against the per-target kernel at 16 agents per scene, -2 % / -4 % of the refine closure at 14 / 12 agents where the
    // per-target kernel did not pay (profiles/r05_ab_fwd_k.json)
    const int split_min = strive_tuning().scene_split;
    const bool scene_split = scene && split_min > 0 && sc->max_n >= split_min;
    // option scene_fwd_k = workgroups per scene for the edge chunks of such scenes (default: one per chunk, <= 4; 0 = the round-4
    // form: scene kernel | one workgroup per target in gnn_edge_kernel | scene kernel)
    int fwd_k = (sc->max_n * (sc->max_n - 1) + scn::EC - 1) / scn::EC;
    if (strive_tuning().scene_fwd_k >= 0) fwd_k = strive_tuning().scene_fwd_k;
    fwd_k = fwd_k > 4 ? 4 : fwd_k;
    if (scene_prof) fwd_k = 0;
    hipLaunchKernelGGL(rollout_init_kernel, dim3((unsigned)((R * 64 + 255) / 256)), dim3(256), 0, stream, tp, past_last,
                       past_feat, map_feat, mapix, w.mapix_rows, (int)R, sc->NS);
    // map feature of step t + 1 at the pose step t produced (reference traffic_model.py:694-695)
    auto encode_step = [&](int t) -> int {
        if (kept)
            return strive_map_cnn_fwd_keep(&dec->map, &dec->cnn, tp.pos_t(t + 1), dec->state_mean, dec->state_std, w.mapix_rows,
                                           (int32_t)R, tp.mf_t(t + 1), w.cnn_ws, w.cnn_bytes, kept, kept_bytes,
                                           (int32_t)((size_t)(FT - 1) * R), (int32_t)((size_t)t * R), stream_);
        return strive_map_cnn_fwd(&dec->map, &dec->cnn, tp.pos_t(t + 1), dec->state_mean, dec->state_std, w.mapix_rows, (int32_t)R,
                                  tp.mf_t(t + 1), w.cnn_ws, w.cnn_bytes, stream_);
    };
    for (int t = 0; t < FT; ++t) {
        if (scene) {
            scn::StepArgsS a;
            a.t = t; a.FT = FT; a.NC = NC; a.max_n = sc->max_n; a.sem = sem; a.lw = lw; a.z = z; a.ext = ext_future; a.ptr = sc->ptr;
            a.par = dec->scene_par; a.traj = traj; a.KW = 0; a.part_a = part_a; a.part_arg = part_arg;
            auto launch_scene = [&](int mode, int ky = 1) {
                a.mode = mode;
                if (scene_prof)       // (tools/scene_phase_probe.py: phase ticks of workgroup 0 at the start of the workspace)
                    hipLaunchKernelGGL(scn::scene_fwd_step_kernel<true>, dim3((unsigned)sc->B, (unsigned)ky), dim3(scn::NTHR), scn::FwdLds::BYTES, stream, gd,
                                       gr, gf, dp, a, tp, (unsigned long long*)ws);
                else
                    hipLaunchKernelGGL(scn::scene_fwd_step_kernel<false>, dim3((unsigned)sc->B, (unsigned)ky), dim3(scn::NTHR), scn::FwdLds::BYTES, stream, gd,
                                       gr, gf, dp, a, tp, (unsigned long long*)nullptr);
            };
            if (scene_split && fwd_k > 0) {
                // large scenes, round 5: K workgroups of the scene kernel per scene share the edge chunks (node phases repeated, partial
                // maxima folded by the second launch): two launches per step instead of three, the 64-row chunk form for the edges
                a.KW = fwd_k;
                launch_scene(3, fwd_k);
                launch_scene(4);
                a.KW = 0;
            } else if (scene_split) {
                // large scenes: node phases per scene, the edge rows on one workgroup per target node in between (gnn_kernels.h)
                GnnBuffers gb = w.gb;
                gb.A = tp.A_t(t); gb.ARG = tp.ARG_t(t); gb.X = tp.X_t(t); gb.P = tp.P_t(t); gb.Q = tp.Q_t(t);
                gb.PRE_IN = tp.PRE_IN_t(t); gb.PRE_E = tp.PRE_E_t(t);
                launch_scene(1);
                hipLaunchKernelGGL(gnn_edge_kernel, dim3((unsigned)R), dim3(256), EdgeLds::bytes(), stream, gd, sd, tp.pos_t(t), gb);
                launch_scene(4);
            } else {
                launch_scene(7);
            }
            if (t < FT - 1) {
                int rc = encode_step(t);
                if (rc) return rc;
            }
            continue;
        }
        GnnBuffers gb = w.gb;
        gb.A = tp.A_t(t);
        gb.ARG = tp.ARG_t(t);
        gb.X = tp.X_t(t);
        gb.P = tp.P_t(t);
        gb.Q = tp.Q_t(t);
        gb.PRE_IN = tp.PRE_IN_t(t);
        gb.PRE_E = tp.PRE_E_t(t);
        if (scene_tiles) {
            // scenes of more than 16 agents: mlp_in .. P / Q and update MLP .. GRU .. dynamics per 16-row tile of a scene on the scene
            // kernel (grid (B, 1, tiles): matrix tiles of 16 rows instead of 4-row blocks, one launch each), the edge rows in between
            // on one workgroup per target node
            scn::StepArgsS a;
            a.t = t; a.FT = FT; a.NC = NC; a.max_n = sc->max_n; a.sem = sem; a.lw = lw; a.z = z; a.ext = ext_future; a.ptr = sc->ptr;
            a.par = dec->scene_par; a.traj = traj; a.KW = 0; a.part_a = nullptr; a.part_arg = nullptr;
            const dim3 grid((unsigned)sc->B, 1, (unsigned)((sc->max_n + scn::NR - 1) / scn::NR));
            a.mode = 1;
            hipLaunchKernelGGL(scn::scene_fwd_step_kernel<false>, grid, dim3(scn::NTHR), scn::FwdLds::BYTES, stream, gd, gr, gf, dp, a, tp,
                               (unsigned long long*)nullptr);
            hipLaunchKernelGGL(gnn_edge_kernel, dim3((unsigned)R), dim3(256), EdgeLds::bytes(), stream, gd, sd, tp.pos_t(t), gb);
            a.mode = 4;
            hipLaunchKernelGGL(scn::scene_fwd_step_kernel<false>, grid, dim3(scn::NTHR), scn::FwdLds::BYTES, stream, gd, gr, gf, dp, a, tp,
                               (unsigned long long*)nullptr);
            if (t < FT - 1) {
                int rc = encode_step(t);
                if (rc) return rc;
            }
            continue;
        }
        FeatSrc f = decoder_features(tp, t, sem, z, lw, NC);
        hipLaunchKernelGGL(gnn_node1_kernel, dim3(nb), dim3(256), Node1Lds::bytes(in_ld1, xs_ld), stream, gd, sc->NS, f, sem,
                           gb, (int)R);
        hipLaunchKernelGGL(gnn_edge_kernel, dim3((unsigned)R), dim3(256), EdgeLds::bytes(), stream, gd, sd, tp.pos_t(t), gb);
        StepArgs a;
        a.t = t; a.FT = FT; a.R = (int)R; a.NS = sc->NS; a.X = gb.X; a.sem = sem; a.lw = lw; a.ext = ext_future;
        a.ptr = sc->ptr; a.scene_of = sc->scene_of; a.traj = traj;
        hipLaunchKernelGGL(rollout_node2_kernel, dim3(nb), dim3(256), node2r_lds_bytes(in_ld2), stream, gd, gr, dp, a, tp);
        if (t < FT - 1) {
            int rc = encode_step(t);
            if (rc) return rc;
        }
    }
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_rollout_fwd(const StriveDecoder* dec, const StriveScenes* sc, const float* past_last,
                                  const float* lw, const float* sem, const float* past_feat, const float* map_feat,
                                  const float* z, const int32_t* mapix, const float* ext_future, int32_t FT, float* traj,
                                  void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    return rollout_forward(dec, sc, past_last, lw, sem, past_feat, map_feat, z, mapix, ext_future, FT, traj, tape, tape_bytes, ws,
                           ws_bytes, nullptr, 0, stream_);
}

extern "C" int strive_rollout_fwd_keep(const StriveDecoder* dec, const StriveScenes* sc, const float* past_last,
                                       const float* lw, const float* sem, const float* past_feat, const float* map_feat,
                                       const float* z, const int32_t* mapix, const float* ext_future, int32_t FT, float* traj,
                                       void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* kept, size_t kept_bytes,
                                       strive_stream_t stream_) {
    STRIVE_CHECK_ARG(kept, "null argument");
    return rollout_forward(dec, sc, past_last, lw, sem, past_feat, map_feat, z, mapix, ext_future, FT, traj, tape, tape_bytes, ws,
                           ws_bytes, kept, kept_bytes, stream_);
}

// =============================================================================================
// backward kernels
// =============================================================================================

// ---- GRU backward for rows r0.. : consumes g_mem (adjoint of mem_{t+1}) and g_pf (adjoint of past_feat_{t+1}),
//      produces g_mem (adjoint of mem_t) and d_loc (adjoint of the local pose fed to the GRU).  grid = ceil(R/RB_NODE)
template <bool WG>
static __global__ __launch_bounds__(256) void gru_bwd_kernel(GRUDev gru, GRUGradDev gg, Tape tp, int t, int R,
                                                               const float* __restrict__ g_pf, float* __restrict__ g_mem,
                                                               float* __restrict__ d_loc) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* s_x = smem;                      // [RB_NODE][4]
    float* s_h = s_x + RB_NODE * 4;              // [3][RB_NODE][64]   layer hidden inputs (mem_t)
    float* s_hn = s_h + 3 * RB_NODE * 64;        // [3][RB_NODE][64]   layer outputs
    float* s_gi = s_hn + 3 * RB_NODE * 64;       // [RB_NODE][GLD]
    float* s_gh = s_gi + RB_NODE * GLD;          // [RB_NODE][GLD]
    float* s_g = s_gh + RB_NODE * GLD;           // [3][RB_NODE][256]  gates r,z,n,gh_n
    float* s_dx = s_g + 3 * RB_NODE * 256;       // [RB_NODE][64]      adjoint flowing to the layer below
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE;
    for (int i = tid; i < RB_NODE * 4; i += 256) {
        const int r = r0 + (i >> 2);
        s_x[i] = (r < R) ? tp.loc_t(t)[(size_t)r * 4 + (i & 3)] : 0.f;
    }
    for (int i = tid; i < 3 * RB_NODE * 64; i += 256) {
        const int l = i / (RB_NODE * 64), rem = i - l * RB_NODE * 64;
        const int r = r0 + (rem >> 6), c = rem & 63;
        s_h[i] = (r < R) ? tp.mem_t(t)[((size_t)r * 3 + l) * 64 + c] : 0.f;
    }
    __syncthreads();
    // the forward sweep kept the gates (r, z, n, W_hn h + b_hn) of this step: nothing to recompute
    for (int i = tid; i < 3 * RB_NODE * 256; i += 256) {
        const int l = i / (RB_NODE * 256), rem = i - l * RB_NODE * 256, rr = rem >> 8, c = rem & 255;
        s_g[i] = (r0 + rr < R) ? tp.GATES_t(t)[((size_t)(r0 + rr) * 3 + l) * 256 + c] : 0.f;
    }
    if (WG) {
        // the weight gradients also need every layer's INPUT: the output of the layer below = that layer's memory of step t + 1,
        // which the forward sweep left on the tape (this kernel only runs for t < FT - 1).  Round 5: the <WG> instantiation used to
        // recompute the three GRU layers for them (77 against 22 us per launch).
        for (int i = tid; i < 2 * RB_NODE * 64; i += 256) {
            const int l = i / (RB_NODE * 64), rem = i - l * RB_NODE * 64;
            const int r = r0 + (rem >> 6), c = rem & 63;
            s_hn[i] = (r < R) ? tp.mem_t(t + 1)[((size_t)r * 3 + l) * 64 + c] : 0.f;
        }
    }
    __syncthreads();
    // backward, top layer first
    for (int l = 2; l >= 0; --l) {
        float* gates = s_g + (size_t)l * RB_NODE * 256;
        const float* h = s_h + (size_t)l * RB_NODE * 64;
        // gate pre-activation adjoints into s_gi (d gi) and s_gh (d gh)
        for (int i = tid; i < RB_NODE * 64; i += 256) {
            const int rr = i >> 6, c = i & 63;
            const int r = r0 + rr;
            float dh = 0.f;
            if (r < R) {
                dh = g_mem[((size_t)r * 3 + l) * 64 + c];
                if (l == 2) dh += g_pf[(size_t)r * 64 + c];
                else dh += s_dx[rr * 64 + c];
            }
            const float rg = gates[rr * 256 + c], z = gates[rr * 256 + 64 + c], n = gates[rr * 256 + 128 + c];
            const float ghn = gates[rr * 256 + 192 + c];
            const float dn = dh * (1.0f - z);
            const float dz = dh * (h[rr * 64 + c] - n);
            const float dpn = dn * (1.0f - n * n);
            const float dr = dpn * ghn;
            const float dpz = dz * z * (1.0f - z);
            const float dpr = dr * rg * (1.0f - rg);
            s_gi[rr * GLD + c] = dpr;
            s_gi[rr * GLD + 64 + c] = dpz;
            s_gi[rr * GLD + 128 + c] = dpn;
            s_gh[rr * GLD + c] = dpr;
            s_gh[rr * GLD + 64 + c] = dpz;
            s_gh[rr * GLD + 128 + c] = dpn * rg;
            // direct path h' = ... + z*h
            s_hn[(size_t)l * RB_NODE * 64 + i] = dh * z;
        }
        __syncthreads();
        const int xin = (l == 0) ? 4 : 64;
        if (WG) {
            // dW_ih = d gi^T . x_l,  dW_hh = d gh^T . h_l  (x_l = the layer below's forward output, still intact in s_hn[l-1])
            const int nrows = (R - r0) < RB_NODE ? (R - r0) : RB_NODE;
            const float* xl = (l == 0) ? s_x : s_hn + (size_t)(l - 1) * RB_NODE * 64;
            wgrad_lds(s_gi, GLD, GLD, xl, l == 0 ? 4 : 64, xin, gg.wih[l], xin, gg.bih[l], nrows, tid, 256, gg.jobs);
            wgrad_lds(s_gh, GLD, GLD, h, 64, 64, gg.whh[l], 64, gg.bhh[l], nrows, tid, 256, gg.jobs);
        }
        // adjoint of the hidden input: dh*z + dgh * W_hh ; adjoint of the layer input: dgi * W_ih
        dense_lds<RB_NODE, true>(s_gh, GLD, GLD, gru.whh[l], 64, nullptr, s_hn + (size_t)l * RB_NODE * 64, 64, 64, tid, 256);
        __syncthreads();     // both calls split k and share dense_lds' partial-sum buffer: the first must have drained it
        dense_lds<RB_NODE, false>(s_gi, GLD, GLD, gru.wih[l], xin, nullptr, s_dx, 64, xin, tid, 256);
        __syncthreads();
        for (int i = tid; i < RB_NODE * 64; i += 256) {
            const int r = r0 + (i >> 6), c = i & 63;
            if (r < R) g_mem[((size_t)r * 3 + l) * 64 + c] = s_hn[(size_t)l * RB_NODE * 64 + i];
        }
        __syncthreads();
    }
    for (int i = tid; i < RB_NODE * 4; i += 256) {
        const int rr = i >> 2, r = r0 + rr;
        if (r < R) d_loc[(size_t)r * 4 + (i & 3)] = s_dx[rr * 64 + (i & 3)];
    }
}

static size_t gru_bwd_lds_bytes() { return (size_t)(RB_NODE * 4 + 6 * RB_NODE * 64 + 2 * RB_NODE * GLD + 3 * RB_NODE * 256 + RB_NODE * 64) * 4; }

// ---- node2 backward: dynamics + mlp_out + update.  grid = ceil(R/RB_NODE)
struct Node2BwdArgs {
    int t, FT, R, NS;
    const float* X;
    const float* sem;
    const float* lw;
    const float* ext;
    const int32_t* ptr;
    const int32_t* scene_of;
    const float* g_traj;     // (R, FT, 4)
    const float* g_pos;      // (R, 4)  adjoint of pos_{t+1}
    const float* d_loc;      // (R, 4)
    float* g_state;          // (R, 8)  in: adjoint of state_{t+1}; out: adjoint of state_t
    float* dX;               // (R, 64) out: update-MLP part of dL/dx
    float* dA;               // (R, 64) out: dL/d(aggregated message)
};

template <bool WG>
static __global__ __launch_bounds__(256) void node2_bwd_kernel(GNNDev g, GNNGradDev gr, DynParams dp, Node2BwdArgs a, Tape tp) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int in_ld = ld4(2 * g.D + g.NC);
    Node2Lds L(smem, in_ld);
    float* s_go = L.out + RB_NODE * HLD;      // [RB_NODE][4]  gradient w.r.t. decoder output (2 used)
    float* s_ga = s_go + RB_NODE * 4;         // [RB_NODE][HLD]
    float* s_gb = s_ga + RB_NODE * HLD;       // [RB_NODE][HLD]
    float* s_gx = s_gb + RB_NODE * HLD;       // [RB_NODE][HLD] gradient w.r.t. x'
    float* s_gin = s_gx + RB_NODE * HLD;      // [RB_NODE][in_ld]
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE, t = a.t;
    {
        // pre-activations and decoder outputs of this step come from the tape
        for (int i = tid; i < RB_NODE * 3 * STRIVE_HID; i += 256) {
            const int rr = i / (3 * STRIVE_HID), rem = i - rr * 3 * STRIVE_HID;
            const bool live = r0 + rr < a.R;
            if (rem < STRIVE_HID) L.pre_u[rr * HLD + rem] = live ? tp.PRE_U_t(t)[(size_t)(r0 + rr) * STRIVE_HID + rem] : 0.f;
            else {
                const int q = rem - STRIVE_HID, l = q / STRIVE_HID, c = q - l * STRIVE_HID;
                L.pre_o[(size_t)l * RB_NODE * HLD + rr * HLD + c] = live ? tp.PRE_O_t(t)[(size_t)(r0 + rr) * 2 * STRIVE_HID + q] : 0.f;
            }
        }
        if (tid < RB_NODE * 2) {
            const int rr = tid >> 1, c = tid & 1;
            L.out[rr * HLD + c] = (r0 + rr < a.R) ? tp.DEC_t(t)[(size_t)(r0 + rr) * 4 + c] : 0.f;
        }
        if (WG) {
            // the weight gradients also need the two networks' INPUTS: (x | aggregated messages | sem) is assembled from the tape,
            // x' = the update network's output is its last layer applied to the taped pre-activation.  Round 5: this instantiation
            // used to recompute all five dense layers of the step (node2_forward: 44 against 18 us per launch).
            const int D = g.D, NC = g.NC;
            for (int i = tid; i < RB_NODE * in_ld; i += 256) {
                const int rr = i / in_ld, k = i - rr * in_ld;
                const int r = r0 + rr;
                float v = 0.f;
                if (r < a.R) {
                    if (k < D) v = a.X[(size_t)r * D + k];
                    else if (k < 2 * D) v = tp.A_t(t)[(size_t)r * D + (k - D)];
                    else if (k < 2 * D + NC) v = a.sem[(size_t)(r / a.NS) * NC + (k - 2 * D)];
                }
                L.in[i] = v;
            }
        }
        __syncthreads();
        if (WG) mlp_forward_lds<RB_NODE>(g.update, L.in, in_ld, L.pre_u, L.act, L.xp, HLD, /*first_done=*/true, tid, 256);
    }
    const bool more = t < a.FT - 1;
    if (tid < RB_NODE) {
        const int r = r0 + tid;
        float gdec[2] = {0.f, 0.f};
        if (r < a.R) {
            const int ag = r / a.NS;
            const float* st = tp.state_t(t) + (size_t)r * 8;
            BikeFwd b;
            bike_forward(dp, st, L.out[tid * HLD + 0], L.out[tid * HLD + 1], a.lw[ag * 2], b);
            bool forced = false;
            float gin[4] = {b.out[0], b.out[1], b.out[2], b.out[3]};
            if (a.ext) {
                const int sc = a.scene_of[ag];
                if (a.ptr[sc] == ag) {
                    forced = true;
                    const float* e = a.ext + ((size_t)sc * a.FT + t) * 4;
                    for (int i = 0; i < 4; ++i) gin[i] = e[i];
                }
            }
            float gbike[6];
            const float* gt = a.g_traj + ((size_t)r * a.FT + t) * 4;
            for (int i = 0; i < 4; ++i) gbike[i] = gt[i];
            gbike[4] = gbike[5] = 0.f;
            float gfr[4] = {0.f, 0.f, 0.f, 0.f};
            if (more) {
                float gpo[4] = {0.f, 0.f, 0.f, 0.f};
                rel_pose_bwd(st, gin, a.d_loc + (size_t)r * 4, gfr, gpo);
                const float* gs = a.g_state + (size_t)r * 8;
                for (int i = 0; i < 6; ++i) gbike[i] += gs[i];
                if (!forced)
                    for (int i = 0; i < 4; ++i) gbike[i] += gpo[i] + a.g_pos[(size_t)r * 4 + i];
            }
            float gst[6];
            bike_backward(dp, b, gbike, gst, gdec);
            float* go = a.g_state + (size_t)r * 8;
            for (int i = 0; i < 6; ++i) go[i] = gst[i] + (i < 4 ? gfr[i] : 0.f);
        }
        s_go[tid * 4 + 0] = gdec[0];
        s_go[tid * 4 + 1] = gdec[1];
        s_go[tid * 4 + 2] = 0.f;
        s_go[tid * 4 + 3] = 0.f;
    }
    __syncthreads();
    const int nrows = (a.R - r0) < RB_NODE ? (a.R - r0) : RB_NODE;
    mlp_backward_lds<RB_NODE, WG>(g.mlp_out, L.pre_o, s_go, 4, s_ga, s_gb, s_gx, HLD, false, tid, 256,
                                  WG ? &gr.mlp_out : nullptr, L.act, L.xp, HLD, nrows);
    mlp_backward_lds<RB_NODE, WG>(g.update, L.pre_u, s_gx, HLD, s_ga, s_gb, s_gin, in_ld, false, tid, 256,
                                  WG ? &gr.update : nullptr, L.act, L.in, in_ld, nrows);
    const int D = g.D;
    for (int i = tid; i < RB_NODE * D; i += 256) {
        const int rr = i / D, c = i - rr * D;
        if (r0 + rr < a.R) {
            a.dX[(size_t)(r0 + rr) * D + c] = s_gin[rr * in_ld + c];
            a.dA[(size_t)(r0 + rr) * D + c] = s_gin[rr * in_ld + D + c];
        }
    }
}

static size_t node2_bwd_lds_bytes(int in_ld) { return (Node2Lds::floats(in_ld) + RB_NODE * 4 + 3 * RB_NODE * HLD + RB_NODE * in_ld) * 4; }

// =============================================================================================
// host orchestration: backward
// =============================================================================================
namespace {
size_t bwd_ws_bytes(size_t R, int max_n) {
    size_t b = 0;
    b += strive_align_up(R * 8 * 4, 256);                  // g_state
    b += 2 * strive_align_up(R * 4 * 4, 256);              // g_pos, d_loc
    b += 2 * strive_align_up(R * 64 * 4, 256);             // g_pf, g_mf
    b += strive_align_up(R * 192 * 4, 256);                // g_mem
    b += gnn_bwd_buffers_bytes(R, 64, max_n);
    return b;
}

// What the training path wants on top of dL/dz (all device pointers; reference src/train_traffic.py:103-131 needs the
// gradients of every parameter the rollout touches and of the encoder outputs it starts from).
struct TrainOut {
    float* d_past_feat;    // (NA, 64)
    float* d_map_feat;     // (NA, 64)
    float* d_gnn;          // flat decoder_net gradients, accumulated
    float* d_gru;          // flat decoder_memory gradients, accumulated
    float* d_cnn;          // flat map_conv + map_feature gradients, accumulated
    const int32_t* mapix;  // (NA)
    const void* kept;      // the forward's map-CNN activations (strive_rollout_fwd_keep) or null: recompute
    size_t kept_bytes;
    // The adjoints of map_feat_t of ALL steps are kept ((FT, R, 64): step 0 is the encoder's map feature) and the CNN
    // backward runs over the (FT - 1) R crops in a few large calls, not one per step: the crop is data (pos_t.detach()), so nothing
    // in the sweep waits for it, and a call over >= 256 samples fills the chip where 11 calls of R = 64 samples were latency-bound
    // (one call after the sweep, or -- round 5, kept activations -- groups of steps on a side stream beside the sweep).
    float* g_mf_all;       // (FT, R, 64)
    int32_t* mapix_all;    // (FT - 1, R)
    WJobTable* jobs;       // deferred weight gradients of decoder_net / decoder_memory (mlp_dev.h), device copy ...
    float* wtape;          // ... and its row tapes
    void* cnn_ws;
    size_t cnn_ws_bytes;
};

static __global__ void tile_mapix_kernel(const int32_t* __restrict__ mapix, int32_t* __restrict__ out, int R, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = mapix[i % R];
}

static __global__ void rollout_init_bwd_kernel(const float* __restrict__ g_pf, const float* __restrict__ g_mem,
                                               const float* __restrict__ g_mf, float* __restrict__ d_past_feat,
                                               float* __restrict__ d_map_feat, int R) {
    // past_feat feeds past_feat_0 and all three GRU layers' initial memory (reference traffic_model.py:625)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * 64) return;
    const int r = i >> 6, c = i & 63;
    float v = g_pf[i];
    for (int l = 0; l < 3; ++l) v += g_mem[((size_t)r * 3 + l) * 64 + c];
    d_past_feat[i] = v;
    d_map_feat[i] = g_mf[i];
}

// rows: every node job gets R rows per step, every edge job one row per directed edge and step: the exact count n_edges when the
// caller's StriveScenes has it (ABI 14), else the bound R max_n (one 60-agent scene in a batch of small ones used to size the
// five edge tapes for R x 60 rows: GBs)
static WJobsPlan wjobs_plan(const StriveGNN& g, const GNNGradDev& gr, const GRUGradDev& gg, float* tape, size_t R, int max_n, int FT,
                            long long n_edges) {
    WJobsPlan p;
    memset(&p.t, 0, sizeof(p.t));
    p.tape_floats = 0; p.max_in = 1; p.max_out = 1;
    p.dropped = false;
    const size_t node_rows = R * (size_t)FT;
    const size_t edge_rows = (n_edges > 0 ? (size_t)n_edges : R * (size_t)(max_n > 1 ? max_n : 1)) * (size_t)FT;
    p.too_large = node_rows > 0x7fffffffull || edge_rows > 0x7fffffffull;
    const int node_cap = p.too_large ? 1 : (int)node_rows, edge_cap = p.too_large ? 1 : (int)(edge_rows > 0 ? edge_rows : 1);
    wjobs_add_gnn(p, tape, g, gr, node_cap, edge_cap);
    for (int l = 0; l < 3; ++l) {
        const int xin = l == 0 ? 4 : 64;
        wjobs_add(p, tape, gg.wih[l], gg.bih[l], GLD, xin, xin, node_cap);
        wjobs_add(p, tape, gg.whh[l], gg.bhh[l], GLD, 64, 64, node_cap);
    }
    wjobs_finish(p.t);
    return p;
}

static size_t wjobs_tape_floats(const StriveGNN& g, size_t R, int max_n, int FT, long long n_edges) {
    float* fake = reinterpret_cast<float*>(uintptr_t(1) << 20);      // (planning only: nothing is dereferenced)
    const GNNGradDev gr = gnn_grad_dev(g, fake);
    const GRUGradDev gg = gru_grad_dev(fake);
    return wjobs_plan(g, gr, gg, fake, R, max_n, FT, n_edges).tape_floats;
}


// a library-owned stream per device for work that overlaps the caller's stream inside ONE call (forked and joined with events
// before the call returns: the caller never sees it)
struct SideStream {
    static constexpr int NEV = 64;
    hipStream_t s;
    hipEvent_t ev[NEV], done;
    bool ok;                          // every create succeeded; otherwise the overlap is off on this device
    std::atomic<int> in_use;          // one strive_rollout_bwd_train_kept at a time per device (host threads): the second one is refused
};
// the guard of one call: taken before the first hand-over, released on every way out
struct SideStreamUse {
    SideStream* s;
    explicit SideStreamUse(SideStream* s_) : s(s_) {}
    ~SideStreamUse() { if (s) s->in_use.store(0, std::memory_order_release); }
};
static SideStream* side_stream() {
    static SideStream table[64];
    static PerDeviceOnce once;
    static std::atomic<int> busy{0};
    const int dev = once.device();
    if (!once.is_done(dev)) {
        int expect = 0;
        while (!busy.compare_exchange_weak(expect, 1)) expect = 0;      // (first use per device only)
        if (!once.is_done(dev)) {
            SideStream& t = table[dev];
            bool ok = hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) == hipSuccess;      // (a lowest-priority stream was measured: no difference)
            for (int i = 0; i < SideStream::NEV; ++i) ok = (hipEventCreateWithFlags(&t.ev[i], hipEventDisableTiming) == hipSuccess) && ok;
            ok = (hipEventCreateWithFlags(&t.done, hipEventDisableTiming) == hipSuccess) && ok;
            (void)hipGetLastError();
            t.ok = ok;
            t.in_use.store(0);
            once.set_done(dev);
        }
        busy.store(0);
    }
    return table[dev].ok ? &table[dev] : nullptr;      // (no side stream: the caller runs the CNN backward after the sweep, on its own stream)
}

template <bool WG>
int rollout_backward(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem, const float* z,
                     const float* ext_future, int32_t FT, const float* d_traj, float* dz, const void* tape, size_t tape_bytes,
                     void* ws, size_t ws_bytes, strive_stream_t stream_, const TrainOut* tr) {
    const size_t R = (size_t)sc->NA * sc->NS;
    hipStream_t stream = (hipStream_t)stream_;
    Tape tp = carve_tape(const_cast<void*>(tape), tape_bytes, R, FT, sc->max_n > 0 ? sc->max_n : 1);
    StriveArena ar(ws, ws_bytes);
    float* g_state = ar.take<float>(R * 8);
    float* g_pos = ar.take<float>(R * 4);
    float* d_loc = ar.take<float>(R * 4);
    float* g_pf = ar.take<float>(R * 64);
    float* g_mf = ar.take<float>(R * 64);
    (void)g_mf;             // (the training path keeps the map-feature adjoints of every step in TrainOut::g_mf_all)
    float* g_mem = ar.take<float>(R * 192);
    GnnBwdBuffers bw = gnn_bwd_buffers_take(ar, R, 64, sc->max_n);
    if (!ar.ok()) { strive_set_error("rollout_bwd: workspace arena overflow"); return -1; }

    const GNNDev gd = gnn_dev(dec->gnn);
    const GRUDev gr = gru_dev(dec->gru);
    GNNGradDev ggn = gnn_grad_dev(dec->gnn, tr ? tr->d_gnn : nullptr);
    GRUGradDev ggr = gru_grad_dev(tr ? tr->d_gru : nullptr);
    WJobsPlan plan;
    plan.t.n = 0;
    if (tr && tr->jobs) {
        plan = wjobs_plan(dec->gnn, ggn, ggr, tr->wtape, R, sc->max_n, FT, sc->n_edges);
        if (plan.too_large || plan.dropped) {
            strive_set_error(plan.too_large ? "rollout_bwd_train: more than 2^31 tape rows (split the batch)" :
                                              "rollout_bwd_train: more deferred weight blocks than STRIVE_WJOBS_MAX");
            return -1;
        }
        hipLaunchKernelGGL(wjobs_upload_kernel, dim3(1), dim3(64), 0, stream, tr->jobs, plan.t);
        ggn.mlp_in.jobs = ggn.edge.jobs = ggn.update.jobs = ggn.mlp_out.jobs = tr->jobs;
        ggr.jobs = tr->jobs;
    }
    const DynParams dp = dyn_params(*dec);
    const ScenesDev sd = scenes_dev(*sc);
    const int NC = dec->gnn.NC;
    const int in_ld1 = ld4(dec->gnn.mlp_in.dims[0]), xs_ld = ld4(64 + NC), in_ld2 = ld4(128 + NC);
    const int nb = (int)((R + RB_NODE - 1) / RB_NODE);

    // Training, kept activations: the map CNN's backward of the crops of step t needs nothing but the map-feature adjoints node1_bwd(t)
    // leaves -- and the sweep is a chain of small kernels on <= R workgroups while the CNN backward fills the chip.  The crops of
    // `grp` consecutive steps (>= 256 samples) are handed to a library-owned side stream as soon as the sweep has passed them; the
    // caller's stream joins at the end.  option train_overlap = 0: one call after the sweep (the round-4 form).
    // (both switches are read per call: the tests run the forms side by side in one process)
    const bool overlap_on = strive_tuning().train_overlap != 0;
    const bool overlap = WG && tr && tr->kept && FT > 1 && overlap_on;
    SideStream* side = overlap ? side_stream() : nullptr;
    if (side && side->in_use.exchange(1, std::memory_order_acquire) != 0) {
        strive_set_error("rollout_bwd_train_kept: another call is using this device's side stream (one such call at a time per device)");
        return -3;
    }
    SideStreamUse side_guard(side);
    const int total_crops = (int)((size_t)(FT > 1 ? FT - 1 : 0) * R);
    const int grp_rows = strive_tuning().train_overlap_rows;
    int grp = (int)(((size_t)(grp_rows > 0 ? grp_rows : 256) + R - 1) / R);
    grp = grp < 1 ? 1 : grp;
    int n_handed = 0, t_hi = FT - 1;
    if (tr && FT > 1)
        hipLaunchKernelGGL(tile_mapix_kernel, dim3((total_crops + 255) / 256), dim3(256), 0, stream, tr->mapix, tr->mapix_all, (int)R, total_crops);

    hipMemsetAsync(g_state, 0, R * 8 * 4, stream);
    hipMemsetAsync(g_pos, 0, R * 4 * 4, stream);
    hipMemsetAsync(d_loc, 0, R * 4 * 4, stream);
    hipMemsetAsync(g_pf, 0, R * 64 * 4, stream);
    hipMemsetAsync(g_mem, 0, R * 192 * 4, stream);
    hipMemsetAsync(dz, 0, R * STRIVE_ZDIM * 4, stream);

    for (int t = FT - 1; t >= 0; --t) {
        GnnBuffers g2;
        g2.A = tp.A_t(t);
        g2.ARG = tp.ARG_t(t);
        g2.X = tp.X_t(t);          // x, P, Q of step t as the forward sweep left them in the tape
        g2.P = tp.P_t(t);
        g2.Q = tp.Q_t(t);
        g2.PRE_IN = tp.PRE_IN_t(t);
        g2.PRE_E = tp.PRE_E_t(t);
        FeatSrc f = decoder_features(tp, t, sem, z, lw, NC);
        if (t < FT - 1)
            hipLaunchKernelGGL(gru_bwd_kernel<WG>, dim3(nb), dim3(256), gru_bwd_lds_bytes(), stream, gr, ggr, tp, t, (int)R, g_pf,
                               g_mem, d_loc);
        Node2BwdArgs a2;
        a2.t = t; a2.FT = FT; a2.R = (int)R; a2.NS = sc->NS; a2.X = g2.X; a2.sem = sem; a2.lw = lw; a2.ext = ext_future;
        a2.ptr = sc->ptr; a2.scene_of = sc->scene_of; a2.g_traj = d_traj; a2.g_pos = g_pos; a2.d_loc = d_loc;
        a2.g_state = g_state; a2.dX = bw.dX; a2.dA = bw.dA;
        hipLaunchKernelGGL(node2_bwd_kernel<WG>, dim3(nb), dim3(256), node2_bwd_lds_bytes(in_ld2), stream, gd, ggn, dp, a2, tp);
        EdgeBwdArgs ae;
        ae.dA = bw.dA; ae.ARG = tp.ARG_t(t); ae.dP = bw.dP; ae.DE1 = bw.DE1; ae.DPJ = bw.DPJ; ae.gpos_tgt = bw.gpos_tgt;
        hipLaunchKernelGGL(edge_bwd_kernel<WG>, dim3((unsigned)R), dim3(256), edge_bwd_lds_bytes(), stream, gd, ggn, sd, tp.pos_t(t),
                           g2, ae);
        Node1BwdArgs a1;
        a1.t = t; a1.R = (int)R; a1.dX = bw.dX; a1.dP = bw.dP; a1.DE1 = bw.DE1; a1.DPJ = bw.DPJ; a1.gpos_tgt = bw.gpos_tgt;
        a1.sem = sem; a1.PRE_IN = tp.PRE_IN_t(t); a1.X = g2.X; a1.g_pos = g_pos; a1.g_full = nullptr; a1.g_pf = g_pf; a1.g_mf = tr ? tr->g_mf_all + (size_t)t * R * 64 : nullptr; a1.dz = dz;
        hipLaunchKernelGGL(node1_bwd_kernel<WG>, dim3(nb), dim3(256), node1_bwd_lds_bytes(in_ld1, xs_ld), stream, gd, ggn, sd, f, a1);
        if (side && t >= 1 && ((t_hi - t + 1 >= grp && n_handed < SideStream::NEV - 1) || t == 1)) {
            // steps t .. t_hi are final: their crops are rows (t - 1) R .. t_hi R of the kept arrays (with more groups than events --
            // FT > 64 at one step per group -- the last group takes all remaining steps)
            hipEvent_t ev = side->ev[n_handed++];
            hipEventRecord(ev, stream);
            hipStreamWaitEvent(side->s, ev, 0);
            const size_t off = (size_t)(t - 1) * R, cnt = (size_t)(t_hi - t + 1) * R;
            int rc = strive_map_cnn_bwd_kept_range(&dec->map, &dec->cnn, tp.pos_t(1) + off * 4, dec->state_mean, dec->state_std,
                                                   tr->mapix_all + off, (int32_t)cnt, tr->g_mf_all + R * 64 + off * 64, tr->d_cnn, tr->kept,
                                                   tr->kept_bytes, (int32_t)total_crops, (int32_t)off, tr->cnn_ws, tr->cnn_ws_bytes,
                                                   (strive_stream_t)side->s);
            if (rc) {
                hipEventRecord(side->done, side->s);            // (join before reporting: nothing of this call stays in flight)
                hipStreamWaitEvent(stream, side->done, 0);
                return rc;
            }
            t_hi = t - 1;
        }
    }
    if (side) {
        hipEventRecord(side->done, side->s);
        hipStreamWaitEvent(stream, side->done, 0);
    }
    if (tr && tr->jobs && plan.t.n > 0)
        hipLaunchKernelGGL(wjobs_gemm_kernel, dim3((plan.max_in + 63) / 64, (plan.max_out + 63) / 64, plan.t.ztotal), dim3(256), 0,
                           stream, tr->jobs);
    if (tr && FT > 1 && !side) {
        // map_feat_t = CNN(crop(pos_t.detach())), t = 1 .. FT-1 (reference traffic_model.py:694-695): the adjoints reach the CNN
        // weights; positions (FT, R, 4) and adjoints (FT, R, 64) are contiguous over the steps
        const int total = total_crops;
        int rc = tr->kept ? strive_map_cnn_bwd_kept(&dec->map, &dec->cnn, tp.pos_t(1), dec->state_mean, dec->state_std, tr->mapix_all,
                                                    (int32_t)total, tr->g_mf_all + R * 64, tr->d_cnn, tr->kept, tr->kept_bytes, tr->cnn_ws,
                                                    tr->cnn_ws_bytes, stream_)
                          : strive_map_cnn_bwd(&dec->map, &dec->cnn, tp.pos_t(1), dec->state_mean, dec->state_std, tr->mapix_all,
                                               (int32_t)total, tr->g_mf_all + R * 64, tr->d_cnn, tr->cnn_ws, tr->cnn_ws_bytes, stream_);
        if (rc) return rc;
    }
    if (tr)
        hipLaunchKernelGGL(rollout_init_bwd_kernel, dim3((unsigned)((R * 64 + 255) / 256)), dim3(256), 0, stream, g_pf, g_mem, tr->g_mf_all,
                           tr->d_past_feat, tr->d_map_feat, (int)R);
    return 0;
}
}  // namespace

extern "C" size_t strive_rollout_workspace_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT) {
    if (!sc) return 0;
    const size_t R = (size_t)sc->NA * sc->NS;
    const size_t f = fwd_ws_bytes(R, fwd_part_elems(dec, sc)), b = bwd_ws_bytes(R, sc->max_n > 0 ? sc->max_n : 1);
    return (f > b ? f : b) + 4096;
}

extern "C" int strive_rollout_bwd(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                                  const float* z, const float* ext_future, int32_t FT, const float* d_traj, float* dz,
                                  const void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(dec && sc && lw && sem && z && d_traj && dz && tape && ws, "null argument");
    if (check_decoder(dec, sc, FT)) return -1;
    const size_t R = (size_t)sc->NA * sc->NS;
    if (R == 0) return 0;
    STRIVE_CHECK_ARG(tape_bytes >= tape_bytes_for(R, FT, sc->max_n > 0 ? sc->max_n : 1), "tape too small");
    STRIVE_CHECK_ARG(ws_bytes >= strive_rollout_workspace_bytes(dec, sc, FT), "workspace too small");
    STRIVE_CHECK_ARG(sc->max_n >= 1, "max_n not set");
    if (scene_kernels_on(dec, sc)) {
        // the whole reverse sweep of a scene in one launch (scene_rollout.h)
        if (scene_kernels_prepare()) return -1;
        Tape tp = carve_tape(const_cast<void*>(tape), tape_bytes, R, FT, sc->max_n);
        scn::SweepArgs a;
        a.FT = FT; a.NC = dec->gnn.NC; a.max_n = sc->max_n; a.sem = sem; a.lw = lw; a.ext = ext_future; a.ptr = sc->ptr;
        a.par = dec->scene_par; a.g_traj = d_traj; a.dz = dz;
        // option scene_prof = 1 (tools/scene_phase_probe.py): workgroup 0 adds the core-clock ticks of every phase to 16 counters at the
        // start of the workspace, which this path does not use otherwise
        const bool prof = strive_tuning().scene_prof != 0;
        // Stepwise form: one launch per reverse step, K workgroups per scene sharing the scene's edge chunks (scene_rollout.h).
        // Default from 3 chunks per scene on (>= 12 agents: 132 edge rows); option sweep_step = 0 (never) | K (1..4, forced).
        const int chunks = (sc->max_n * (sc->max_n - 1) + scn::EC - 1) / scn::EC;
        int K = chunks >= 3 ? chunks : 0;
        if (strive_tuning().sweep_step >= 0) K = strive_tuning().sweep_step;
        if (K > 4) K = 4;
        const size_t need = (size_t)sc->B * (K > 0 ? K : 1) * (2 * scn::SWEEP_PART_FLOATS + scn::SWEEP_STATE_FLOATS) * 4 + 512;
        if (K >= 1 && !prof && ws_bytes >= need) {
            a.K = K;
            a.part = reinterpret_cast<float*>((char*)ws + 256);
            a.state = a.part + 2 * (size_t)sc->B * K * scn::SWEEP_PART_FLOATS;
            const GNNDev gd = gnn_dev(dec->gnn);
            const GRUDev gr = gru_dev(dec->gru);
            const scn::GRUFrag gf = scn::gru_frag(dec->gru);
            const DynParams dp = dyn_params(*dec);
            for (int t = FT - 1; t >= -1; --t) {
                a.t = t;
                hipLaunchKernelGGL((scn::scene_bwd_sweep_kernel<false, true>), dim3((unsigned)sc->B, (unsigned)K), dim3(scn::NTHR), scn::BwdLds::BYTES,
                                   (hipStream_t)stream_, gd, gr, gf, dp, a, tp, (unsigned long long*)nullptr);
            }
            STRIVE_CHECK_LAUNCH();
            return 0;
        }
        a.t = 0; a.K = 1; a.part = nullptr; a.state = nullptr;
        if (prof)
            hipLaunchKernelGGL(scn::scene_bwd_sweep_kernel<true>, dim3((unsigned)sc->B), dim3(scn::NTHR), scn::BwdLds::BYTES, (hipStream_t)stream_,
                               gnn_dev(dec->gnn), gru_dev(dec->gru), scn::gru_frag(dec->gru), dyn_params(*dec), a, tp, (unsigned long long*)ws + 32);
        else
            hipLaunchKernelGGL(scn::scene_bwd_sweep_kernel<false>, dim3((unsigned)sc->B), dim3(scn::NTHR), scn::BwdLds::BYTES, (hipStream_t)stream_,
                               gnn_dev(dec->gnn), gru_dev(dec->gru), scn::gru_frag(dec->gru), dyn_params(*dec), a, tp, (unsigned long long*)nullptr);
        STRIVE_CHECK_LAUNCH();
        return 0;
    }
    int rc = rollout_backward<false>(dec, sc, lw, sem, z, ext_future, FT, d_traj, dz, tape, tape_bytes, ws, ws_bytes, stream_, nullptr);
    if (rc) return rc;
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t strive_rollout_train_workspace_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT) {
    if (!sc) return 0;
    const size_t R = (size_t)sc->NA * sc->NS, steps = FT > 1 ? (size_t)(FT - 1) : 1;
    return strive_rollout_workspace_bytes(dec, sc, FT) + strive_align_up(strive_map_cnn_bwd_workspace_bytes((int32_t)(steps * R)), 256) +
           strive_align_up((size_t)FT * R * 64 * 4, 256) + strive_align_up(steps * R * 4, 256) +
           strive_align_up(sizeof(WJobTable), 256) + strive_align_up(wjobs_tape_floats(dec->gnn, R, sc->max_n, FT, sc->n_edges) * 4, 256);
}

extern "C" size_t strive_gnn_param_count(const StriveGNN* gnn) { return gnn ? gnn_param_count(*gnn) : 0; }
extern "C" size_t strive_gru_param_count(void) { return gru_param_count(); }

static int rollout_backward_train(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                                  const float* z, const float* ext_future, const int32_t* mapix, int32_t FT,
                                  const float* d_traj, float* dz, float* d_past_feat, float* d_map_feat, float* d_gnn,
                                  float* d_gru, float* d_cnn, const void* tape, size_t tape_bytes, const void* kept, size_t kept_bytes,
                                  void* ws, size_t ws_bytes, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(dec && sc && lw && sem && z && mapix && d_traj && dz && d_past_feat && d_map_feat && d_gnn && d_gru && d_cnn &&
                     tape && ws, "null argument");
    if (check_decoder(dec, sc, FT)) return -1;
    STRIVE_CHECK_ARG(sc->NS == 1, "the training backward takes 2-D latents (one sample per agent)");
    const size_t R = (size_t)sc->NA;
    if (R == 0) return 0;
    STRIVE_CHECK_ARG(tape_bytes >= tape_bytes_for(R, FT, sc->max_n > 0 ? sc->max_n : 1), "tape too small");
    STRIVE_CHECK_ARG(ws_bytes >= strive_rollout_train_workspace_bytes(dec, sc, FT), "workspace too small");
    STRIVE_CHECK_ARG(sc->max_n >= 1, "max_n not set");
    const size_t base = strive_rollout_workspace_bytes(dec, sc, FT);
    STRIVE_CHECK_ARG(!kept || kept_bytes >= strive_rollout_keep_bytes(dec, sc, FT), "kept-activation buffer too small");
    TrainOut tr;
    tr.d_past_feat = d_past_feat; tr.d_map_feat = d_map_feat; tr.d_gnn = d_gnn; tr.d_gru = d_gru; tr.d_cnn = d_cnn;
    tr.mapix = mapix;
    tr.kept = kept; tr.kept_bytes = kept_bytes;
    {
        const size_t steps = FT > 1 ? (size_t)(FT - 1) : 1;
        char* p = (char*)ws + base;
        tr.g_mf_all = (float*)p;
        p += strive_align_up((size_t)FT * R * 64 * 4, 256);
        tr.mapix_all = (int32_t*)p;
        p += strive_align_up(steps * R * 4, 256);
        const bool atomics_only = strive_tuning().wgrad_atomics != 0;      // A/B switch: no deferred weight gradients
        tr.jobs = atomics_only ? nullptr : (WJobTable*)p;
        p += strive_align_up(sizeof(WJobTable), 256);
        tr.wtape = (float*)p;
        p += strive_align_up(wjobs_tape_floats(dec->gnn, R, sc->max_n, FT, sc->n_edges) * 4, 256);
        tr.cnn_ws = p;
        tr.cnn_ws_bytes = ws_bytes - (size_t)(p - (char*)ws);
    }
    int rc = rollout_backward<true>(dec, sc, lw, sem, z, ext_future, FT, d_traj, dz, tape, tape_bytes, ws, base, stream_, &tr);
    if (rc) return rc;
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_rollout_bwd_train(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                                        const float* z, const float* ext_future, const int32_t* mapix, int32_t FT,
                                        const float* d_traj, float* dz, float* d_past_feat, float* d_map_feat, float* d_gnn,
                                        float* d_gru, float* d_cnn, const void* tape, size_t tape_bytes, void* ws,
                                        size_t ws_bytes, strive_stream_t stream_) {
    return rollout_backward_train(dec, sc, lw, sem, z, ext_future, mapix, FT, d_traj, dz, d_past_feat, d_map_feat, d_gnn, d_gru, d_cnn,
                                  tape, tape_bytes, nullptr, 0, ws, ws_bytes, stream_);
}

extern "C" int strive_rollout_bwd_train_kept(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                                             const float* z, const float* ext_future, const int32_t* mapix, int32_t FT,
                                             const float* d_traj, float* dz, float* d_past_feat, float* d_map_feat, float* d_gnn,
                                             float* d_gru, float* d_cnn, const void* tape, size_t tape_bytes, const void* kept,
                                             size_t kept_bytes, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(kept, "null argument");
    return rollout_backward_train(dec, sc, lw, sem, z, ext_future, mapix, FT, d_traj, dz, d_past_feat, d_map_feat, d_gnn, d_gru, d_cnn,
                                  tape, tape_bytes, kept, kept_bytes, ws, ws_bytes, stream_);
}

// =============================================================================================
// operator-level views of the in-register building blocks (parity tests call these; the rollout kernels call the same
// device functions): one kinematic bicycle step and the rigid frame change
// =============================================================================================
static __global__ void bicycle_step_kernel(DynParams dp, const float* __restrict__ state, const float* __restrict__ dec,
                                           const float* __restrict__ lw0, const float* __restrict__ g_out, float* __restrict__ out,
                                           float* __restrict__ g_state, float* __restrict__ g_dec, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    BikeFwd b;
    bike_forward(dp, state + (size_t)i * 6, dec[2 * i], dec[2 * i + 1], lw0[i], b);
    for (int c = 0; c < 6; ++c) out[(size_t)i * 6 + c] = b.out[c];
    if (g_out) {
        float gs[6], gd[2];
        bike_backward(dp, b, g_out + (size_t)i * 6, gs, gd);
        for (int c = 0; c < 6; ++c) g_state[(size_t)i * 6 + c] = gs[c];
        g_dec[2 * i] = gd[0];
        g_dec[2 * i + 1] = gd[1];
    }
}

extern "C" int strive_bicycle_step(const StriveDecoder* dec, const float* state, const float* dec_out, const float* lw0,
                                   const float* g_out, float* out, float* g_state, float* g_dec, int32_t N, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(dec && state && dec_out && lw0 && out && N >= 0, "null argument");
    STRIVE_CHECK_ARG(!g_out || (g_state && g_dec), "g_state / g_dec missing");
    if (N == 0) return 0;
    hipLaunchKernelGGL(bicycle_step_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream_, dyn_params(*dec), state, dec_out, lw0,
                       g_out, out, g_state, g_dec, (int)N);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

static __global__ void rel_pose_kernel(const float* __restrict__ frame, const float* __restrict__ poses, const float* __restrict__ g_out,
                                       float* __restrict__ out, float* __restrict__ g_frame, float* __restrict__ g_poses, int N, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float gf[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < M; ++j) {
        const size_t o = ((size_t)i * M + j) * 4;
        float r[4];
        rel_pose(frame + (size_t)i * 4, poses + o, r);
        for (int c = 0; c < 4; ++c) out[o + c] = r[c];
        if (g_out) {
            float gp[4] = {0.f, 0.f, 0.f, 0.f};
            rel_pose_bwd(frame + (size_t)i * 4, poses + o, g_out + o, gf, gp);
            for (int c = 0; c < 4; ++c) g_poses[o + c] = gp[c];
        }
    }
    if (g_out)
        for (int c = 0; c < 4; ++c) g_frame[(size_t)i * 4 + c] = gf[c];
}

extern "C" int strive_rel_pose(const float* frame, const float* poses, const float* g_out, float* out, float* g_frame, float* g_poses,
                               int32_t N, int32_t M, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(frame && poses && out && N >= 0 && M >= 1, "null argument");
    STRIVE_CHECK_ARG(!g_out || (g_frame && g_poses), "g_frame / g_poses missing");
    if (N == 0) return 0;
    hipLaunchKernelGGL(rel_pose_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream_, frame, poses, g_out, out, g_frame, g_poses,
                       (int)N, (int)M);
    STRIVE_CHECK_LAUNCH();
    return 0;
}
