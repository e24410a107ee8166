// Backward kernels of the scene interaction network (reference src/models/interaction_net.py:16-218 under autograd),
// shared by the decoder rollout's reverse sweep (rollout.hip) and the stand-alone network backward (mlp_gnn.hip).
//
// Input gradients are deterministic (per-edge adjoints are written to slots and reduced in a fixed order).  Weight
// gradients -- needed by the training path only (reference src/train_traffic.py:103-131) -- are accumulated with fp32
// atomics into a flat buffer in the parameter order of the reference module; the latent-optimisation path passes null
// gradient pointers and executes none of that code.
#pragma once
#include "gnn_kernels.h"

struct GNNGradDev {
    MLPGradDev mlp_in, edge, update, mlp_out;
    bool on;
};

static inline size_t gnn_param_count(const StriveGNN& g) {
    return mlp_param_count(g.mlp_in) + mlp_param_count(g.edge) + mlp_param_count(g.update) + mlp_param_count(g.mlp_out);
}

// flat buffer in named_parameters() order: mlp_in | msg.0.edge_mlp | msg.0.update_mlp | mlp_out
static inline GNNGradDev gnn_grad_dev(const StriveGNN& g, float* flat) {
    GNNGradDev d;
    float* p = flat;
    d.on = flat != nullptr;
    d.mlp_in = mlp_grad_dev(g.mlp_in, flat ? &p : nullptr);
    d.edge = mlp_grad_dev(g.edge, flat ? &p : nullptr);
    d.update = mlp_grad_dev(g.update, flat ? &p : nullptr);
    d.mlp_out = mlp_grad_dev(g.mlp_out, flat ? &p : nullptr);
    return d;
}

// ---------------------------------------------------------------------------------------------
// node2 backward (stand-alone network): d_out -> mlp_out -> update -> dX (update part), dA.   grid = ceil(R/RB_NODE)
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void gnn_node2_bwd_kernel(GNNDev g, GNNGradDev gr, int NS, const float* __restrict__ sem,
                                                                     GnnBuffers gb, const float* __restrict__ d_out,
                                                                     float* __restrict__ dX, float* __restrict__ dA, int R) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int in_ld = ld4(2 * g.D + g.NC);
    Node2Lds L(smem, in_ld);
    float* s_go = L.out + RB_NODE * HLD;      // [RB_NODE][HLD] gradient w.r.t. the network output
    float* s_ga = s_go + RB_NODE * HLD;       // [RB_NODE][HLD]
    float* s_gb = s_ga + RB_NODE * HLD;       // [RB_NODE][HLD]
    float* s_gx = s_gb + RB_NODE * HLD;       // [RB_NODE][HLD] gradient w.r.t. x'
    float* s_gin = s_gx + RB_NODE * HLD;      // [RB_NODE][in_ld]
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE;
    node2_forward(g, NS, gb.X, gb.A, sem, r0, R, L, in_ld, tid);
    const int O = g.mlp_out.dims[g.mlp_out.nlayers];
    const int nrows = (R - r0) < RB_NODE ? (R - r0) : RB_NODE;
    for (int i = tid; i < RB_NODE * HLD; i += 256) {
        const int rr = i / HLD, c = i - rr * HLD;
        s_go[i] = (rr < nrows && c < O) ? d_out[(size_t)(r0 + rr) * O + c] : 0.f;
    }
    __syncthreads();
    mlp_backward_lds<RB_NODE, true>(g.mlp_out, L.pre_o, s_go, HLD, s_ga, s_gb, s_gx, HLD, false, tid, 256,
                                    gr.on ? &gr.mlp_out : nullptr, L.act, L.xp, HLD, nrows);
    mlp_backward_lds<RB_NODE, true>(g.update, L.pre_u, s_gx, HLD, s_ga, s_gb, s_gin, in_ld, false, tid, 256,
                                    gr.on ? &gr.update : nullptr, L.act, L.in, in_ld, nrows);
    const int D = g.D;
    for (int i = tid; i < RB_NODE * D; i += 256) {
        const int rr = i / D, c = i - rr * D;
        if (r0 + rr < R) {
            dX[(size_t)(r0 + rr) * D + c] = s_gin[rr * in_ld + c];
            dA[(size_t)(r0 + rr) * D + c] = s_gin[rr * in_ld + D + c];
        }
    }
}

static inline size_t gnn_node2_bwd_lds_bytes(int in_ld) { return (Node2Lds::floats(in_ld) + 4 * RB_NODE * HLD + RB_NODE * in_ld) * 4; }

// ---------------------------------------------------------------------------------------------
// edge backward: one workgroup per target row.  grid = R
// ---------------------------------------------------------------------------------------------
struct EdgeBwdArgs {
    const float* dA;       // (R, D)
    const int32_t* ARG;    // (R, D)
    float* dP;             // (R, 128)
    float* DE1;            // (R*max_n, 128)  per-edge layer-0 adjoint, slot = target_row*max_n + local source index
    float* DPJ;            // (R*max_n, 4)    per-edge adjoint of the SOURCE pose
    float* gpos_tgt;       // (R, 4)          adjoint of the TARGET pose (frame), summed over its edges
};

template <bool WG>
static __global__ __launch_bounds__(256) void edge_bwd_kernel(GNNDev g, GNNGradDev gr, ScenesDev sc, const float* __restrict__ pos,
                                                                GnnBuffers gb, EdgeBwdArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    EdgeLds L(smem);
    float* s_ga = (float*)(L.src + RB_EDGE);       // [RB_EDGE][HLD]
    float* s_gb = s_ga + RB_EDGE * HLD;            // [RB_EDGE][HLD]
    float* s_grel = s_gb + RB_EDGE * HLD;          // [RB_EDGE][4]
    float* s_gfr = s_grel + RB_EDGE * 4;           // [RB_EDGE][4]
    unsigned* s_nan = (unsigned*)(s_gfr + RB_EDGE * 4);   // [RB_EDGE]
    const int r = blockIdx.x, tid = threadIdx.x, D = g.D, H = STRIVE_HID;
    const int ag = r / sc.NS;
    const int b = sc.scene_of[ag];
    const int lo = sc.ptr[b];
    const int nsrc = sc.ptr[b + 1] - lo - 1;
    const int nchunks = (nsrc + RB_EDGE - 1) / RB_EDGE;
    const int EIN = g.edge.dims[0];
    const float* Wrel = g.edge.wt[0] + (size_t)(2 * D + 2 * g.NC) * H;
    float dp_acc = 0.f;                  // thread c < 128: sum over sources of d e1[.][c]
    float gfr_acc[4] = {0.f, 0.f, 0.f, 0.f};   // thread 0
    for (int ch = 0; ch < nchunks; ++ch) {
        // the hidden layers' pre-activations come from the forward pass's table when it kept one (no recompute)
        const int nv = edge_chunk_setup(g, sc, pos, gb, r, ch, L, s_nan, tid, gb.PRE_E);
        if (!gb.PRE_E) mlp_forward_lds<RB_EDGE>(g.edge, nullptr, 0, L.pre, L.act, L.m, HLD, true, tid, 256);
        // route d(aggregate) to the arg-max edge of every channel
        for (int i = tid; i < RB_EDGE * D; i += 256) {
            const int jr = i / D, c = i - jr * D;
            float v = 0.f;
            if (jr < nv && a.ARG[(size_t)r * D + c] == L.src[jr]) v = a.dA[(size_t)r * D + c];
            L.m[jr * HLD + c] = v;
        }
        __syncthreads();
        mlp_backward_lds<RB_EDGE, WG>(g.edge, L.pre, L.m, HLD, s_ga, s_gb, nullptr, 0, true, tid, 256,
                                      WG ? &gr.edge : nullptr, L.act, nullptr, 0, nv);   // d e1 -> s_ga
        if (WG) {
            // layer 0 is factorised: the relative-pose columns of its weight get d e1^T . rel here (NaN components were
            // replaced by 0 in L.rel, like the forward); the node columns and the bias follow in node1_bwd from dP / dQ
            wgrad_lds(s_ga, HLD, H, L.rel, 4, 4, gr.edge.w[0] + (2 * D + 2 * g.NC), EIN, nullptr, nv, tid, 256, gr.edge.jobs);
        }
        // per-edge outputs
        for (int i = tid; i < RB_EDGE * H; i += 256) {
            const int jr = i / H, c = i - jr * H;
            if (jr < nv) {
                const int jl = L.src[jr] / sc.NS - lo;
                a.DE1[((size_t)r * sc.max_n + jl) * H + c] = s_ga[jr * HLD + c];
            }
        }
        if (tid < H) {
            for (int jr = 0; jr < nv; ++jr) dp_acc += s_ga[jr * HLD + tid];
        }
        // d rel = d e1 . W_rel^T : one wave per edge row, lanes over channels
        {
            const int wave = tid >> 6, lane = tid & 63;
            for (int jr = wave; jr < RB_EDGE; jr += 4) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                for (int c = lane; c < H; c += 64) {
                    const float ge = s_ga[jr * HLD + c];
                    for (int d = 0; d < 4; ++d) v[d] = fmaf(ge, Wrel[d * H + c], v[d]);
                }
                for (int d = 0; d < 4; ++d) v[d] = wave_sum(v[d]);
                if (lane == 0)
                    for (int d = 0; d < 4; ++d) s_grel[jr * 4 + d] = v[d];
            }
        }
        __syncthreads();
        if (tid < RB_EDGE) {
            float gfr[4] = {0.f, 0.f, 0.f, 0.f}, gpo[4] = {0.f, 0.f, 0.f, 0.f};
            if (tid < nv) {
                const int srow = L.src[tid];
                float gr4[4];
                for (int d = 0; d < 4; ++d) gr4[d] = (s_nan[tid] >> d) & 1u ? 0.f : s_grel[tid * 4 + d];
                rel_pose_bwd(pos + (size_t)r * 4, pos + (size_t)srow * 4, gr4, gfr, gpo);
                const int jl = srow / sc.NS - lo;
                float* o = a.DPJ + ((size_t)r * sc.max_n + jl) * 4;
                for (int d = 0; d < 4; ++d) o[d] = gpo[d];
            }
            for (int d = 0; d < 4; ++d) s_gfr[tid * 4 + d] = gfr[d];
        }
        __syncthreads();
        if (tid == 0)
            for (int jr = 0; jr < nv; ++jr)
                for (int d = 0; d < 4; ++d) gfr_acc[d] += s_gfr[jr * 4 + d];
        __syncthreads();
    }
    if (tid < H) a.dP[(size_t)r * H + tid] = dp_acc;
    if (tid == 0)
        for (int d = 0; d < 4; ++d) a.gpos_tgt[(size_t)r * 4 + d] = gfr_acc[d];
}

static inline size_t edge_bwd_lds_bytes() { return EdgeLds::bytes() + (size_t)(2 * RB_EDGE * HLD + 2 * RB_EDGE * 4 + RB_EDGE) * 4; }

// ---------------------------------------------------------------------------------------------
// node1 backward: gather source-side adjoints, back through the edge layer-0 partials and mlp_in.  grid = ceil(R/RB_NODE)
// ---------------------------------------------------------------------------------------------
struct Node1BwdArgs {
    int t, R;
    const float* dX;        // (R, D)
    const float* dP;        // (R, 128)
    const float* DE1;
    const float* DPJ;
    const float* gpos_tgt;
    const float* sem;       // (NA, NC)   needed for the weight gradients only
    const float* PRE_IN;    // (R, 2, 128) mlp_in pre-activations kept by the forward pass, or null (then recomputed)
    const float* X;         // (R, D) node embeddings (needed with PRE_IN for nothing but the weight gradients)
    float* g_pos;           // (R, 4)  out: adjoint of pos
    // adjoint of the node features: any of these may be null
    float* g_full;          // (R, F)  the whole feature row (stand-alone network)
    float* g_pf;            // (R, 64) columns [0, 64)      (rollout: past_feat_t)
    float* g_mf;            // (R, 64) columns [64, 128)    (rollout, training: map_feat_t)
    float* dz;              // (R, 32) columns [128+NC, 128+NC+32), ACCUMULATED   (rollout: the latents)
};

template <bool WG>
static __global__ __launch_bounds__(256) void node1_bwd_kernel(GNNDev g, GNNGradDev gr, ScenesDev sc, FeatSrc f, Node1BwdArgs a) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int F = g.mlp_in.dims[0], D = g.D, H = STRIVE_HID;
    const int in_ld = ld4(F), xs_ld = ld4(D + g.NC);
    Node1Lds L(smem, in_ld, xs_ld);
    float* s_dp = L.po + RB_NODE * HLD;       // [RB_NODE][HLD]  dP rows
    float* s_dq = s_dp + RB_NODE * HLD;       // [RB_NODE][HLD]  dQ rows
    float* s_gx = s_dq + RB_NODE * HLD;       // [RB_NODE][HLD]  adjoint of x (D wide)
    float* s_gb = s_gx + RB_NODE * HLD;       // [RB_NODE][HLD]
    float* s_gin = s_gb + RB_NODE * HLD;      // [RB_NODE][in_ld]
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE, NS = sc.NS;
    const int nrows = (a.R - r0) < RB_NODE ? (a.R - r0) : RB_NODE;
    if (a.PRE_IN) {
        // the forward pass kept mlp_in's pre-activations: no dense layer is recomputed.  The weight gradients also need the
        // network's input (the features, gathered again: loads only) and its output x (on the tape as well).  Round 5: the <WG>
        // instantiation used to re-run the three layers of mlp_in for them (63 against 30 us per launch).
        for (int i = tid; i < RB_NODE * 2 * H; i += 256) {
            const int rr = i / (2 * H), rem = i - rr * 2 * H, l = rem / H, c = rem - l * H;
            L.pre[(size_t)l * RB_NODE * HLD + rr * HLD + c] = (r0 + rr < a.R) ? a.PRE_IN[(size_t)(r0 + rr) * 2 * H + rem] : 0.f;
        }
        if (WG) {
            gather_features<RB_NODE>(f, r0, a.R, NS, L.in, in_ld, tid, 256);
            for (int i = tid; i < RB_NODE * D; i += 256) {
                const int rr = i / D, c = i - rr * D;
                L.xs[rr * xs_ld + c] = (r0 + rr < a.R) ? a.X[(size_t)(r0 + rr) * D + c] : 0.f;
            }
        }
    } else {
        // forward recompute of mlp_in (pre-activations)
        gather_features<RB_NODE>(f, r0, a.R, NS, L.in, in_ld, tid, 256);
        __syncthreads();
        mlp_forward_lds<RB_NODE>(g.mlp_in, L.in, in_ld, L.pre, L.act, L.xs, xs_ld, false, tid, 256);
    }
    // gather dP, dQ = sum over targets of the per-edge adjoints, and the source-pose adjoint
    for (int i = tid; i < RB_NODE * H; i += 256) {
        const int rr = i / H, c = i - rr * H;
        const int r = r0 + rr;
        float vp = 0.f, vq = 0.f;
        if (r < a.R) {
            vp = a.dP[(size_t)r * H + c];
            const int ag = r / NS, s = r - ag * NS;
            const int b = sc.scene_of[ag];
            const int lo = sc.ptr[b], hi = sc.ptr[b + 1];
            const int jl = ag - lo;
            for (int ia = lo; ia < hi; ++ia) {
                if (ia == ag) continue;
                vq += a.DE1[((size_t)(ia * NS + s) * sc.max_n + jl) * H + c];
            }
        }
        s_dp[rr * HLD + c] = vp;
        s_dq[rr * HLD + c] = vq;
    }
    if (tid < RB_NODE * 4) {
        const int rr = tid >> 2, d = tid & 3;
        const int r = r0 + rr;
        if (r < a.R) {
            float v = a.gpos_tgt[(size_t)r * 4 + d];
            const int ag = r / NS, s = r - ag * NS;
            const int b = sc.scene_of[ag];
            const int lo = sc.ptr[b], hi = sc.ptr[b + 1];
            const int jl = ag - lo;
            for (int ia = lo; ia < hi; ++ia) {
                if (ia == ag) continue;
                v += a.DPJ[((size_t)(ia * NS + s) * sc.max_n + jl) * 4 + d];
            }
            a.g_pos[(size_t)r * 4 + d] = v;
        }
    }
    const int EIN = g.edge.dims[0];
    if (WG) {
        // node columns and bias of the factorised edge layer 0:  e1_ij = W[:, x_i] x_i + W[:, x_j] x_j + W[:, s_i] sem_i +
        // W[:, s_j] sem_j + W[:, rel] rel_ij + b,  so  dW[:, x_i] = sum_i dP_i x_i^T,  dW[:, x_j] = sum_j dQ_j x_j^T, ...
        for (int i = tid; i < RB_NODE * (xs_ld - D); i += 256) {
            const int rr = i / (xs_ld - D), k = i - rr * (xs_ld - D);
            const int r = r0 + rr;
            L.xs[rr * xs_ld + D + k] = (r < a.R && k < g.NC) ? a.sem[(size_t)(r / NS) * g.NC + k] : 0.f;
        }
        __syncthreads();
        float* w0 = gr.edge.w[0];
        wgrad_lds(s_dp, HLD, H, L.xs, xs_ld, D, w0, EIN, gr.edge.b[0], nrows, tid, 256, gr.edge.jobs);
        wgrad_lds(s_dq, HLD, H, L.xs, xs_ld, D, w0 + D, EIN, nullptr, nrows, tid, 256, gr.edge.jobs);
        wgrad_lds(s_dp, HLD, H, L.xs + D, xs_ld, g.NC, w0 + 2 * D, EIN, nullptr, nrows, tid, 256, gr.edge.jobs);
        wgrad_lds(s_dq, HLD, H, L.xs + D, xs_ld, g.NC, w0 + 2 * D + g.NC, EIN, nullptr, nrows, tid, 256, gr.edge.jobs);
    }
    __syncthreads();
    // adjoint of x: dP . W_e0[:, 0:D] + dQ . W_e0[:, D:2D] + update-MLP part
    dense_lds<RB_NODE, false>(s_dp, HLD, H, g.edge.w[0], EIN, nullptr, s_gx, HLD, D, tid, 256);
    __syncthreads();
    dense_lds<RB_NODE, true>(s_dq, HLD, H, g.edge.w[0] + D, EIN, nullptr, s_gx, HLD, D, tid, 256);
    __syncthreads();
    for (int i = tid; i < RB_NODE * D; i += 256) {
        const int rr = i / D, c = i - rr * D;
        if (r0 + rr < a.R) s_gx[rr * HLD + c] += a.dX[(size_t)(r0 + rr) * D + c];
    }
    __syncthreads();
    mlp_backward_lds<RB_NODE, WG>(g.mlp_in, L.pre, s_gx, HLD, s_dp, s_gb, s_gin, in_ld, false, tid, 256,
                                  WG ? &gr.mlp_in : nullptr, L.act, L.in, in_ld, nrows);
    if (a.g_full) {
        for (int i = tid; i < RB_NODE * F; i += 256) {
            const int rr = i / F, c = i - rr * F;
            if (r0 + rr < a.R) a.g_full[(size_t)(r0 + rr) * F + c] = s_gin[rr * in_ld + c];
        }
    }
    if (a.g_pf || a.g_mf) {
        for (int i = tid; i < RB_NODE * 64; i += 256) {
            const int rr = i >> 6, c = i & 63;
            if (r0 + rr < a.R) {
                if (a.g_pf) a.g_pf[(size_t)(r0 + rr) * 64 + c] = s_gin[rr * in_ld + c];
                if (a.g_mf) a.g_mf[(size_t)(r0 + rr) * 64 + c] = s_gin[rr * in_ld + 64 + c];
            }
        }
    }
    if (a.dz) {
        const int zoff = 128 + g.NC;
        for (int i = tid; i < RB_NODE * STRIVE_ZDIM; i += 256) {
            const int rr = i / STRIVE_ZDIM, c = i - rr * STRIVE_ZDIM;
            if (r0 + rr < a.R) a.dz[(size_t)(r0 + rr) * STRIVE_ZDIM + c] += s_gin[rr * in_ld + zoff + c];
        }
    }
}

static inline size_t node1_bwd_lds_bytes(int in_ld, int xs_ld) { return Node1Lds::bytes(in_ld, xs_ld) + (size_t)(4 * RB_NODE * HLD + RB_NODE * in_ld) * 4; }

// workspace of the per-step backward buffers shared by the rollout sweep and the stand-alone network backward
struct GnnBwdBuffers {
    float *dX, *dA, *dP, *gpos_tgt, *DE1, *DPJ;
};

static inline size_t gnn_bwd_buffers_bytes(size_t R, int D, int max_n) {
    size_t b = 0;
    b += 2 * strive_align_up(R * D * 4, 256);                          // dX, dA
    b += strive_align_up(R * STRIVE_HID * 4, 256);                     // dP
    b += strive_align_up(R * 4 * 4, 256);                              // gpos_tgt
    b += strive_align_up(R * (size_t)max_n * STRIVE_HID * 4, 256);     // DE1
    b += strive_align_up(R * (size_t)max_n * 4 * 4, 256);              // DPJ
    return b;
}

static inline GnnBwdBuffers gnn_bwd_buffers_take(StriveArena& ar, size_t R, int D, int max_n) {
    GnnBwdBuffers w;
    w.dX = ar.take<float>(R * D);
    w.dA = ar.take<float>(R * D);
    w.dP = ar.take<float>(R * STRIVE_HID);
    w.gpos_tgt = ar.take<float>(R * 4);
    w.DE1 = ar.take<float>(R * (size_t)max_n * STRIVE_HID);
    w.DPJ = ar.take<float>(R * (size_t)max_n * 4);
    return w;
}

// =============================================================================================
// Job table of the deferred weight gradients (mlp_dev.h WJobTable): one job per weight block a backward pass touches.
// Shared by the rollout's training sweep (rollout.hip) and the stand-alone network backward (strive_gnn_bwd).
// =============================================================================================
struct WJobsPlan {
    WJobTable t;
    size_t tape_floats;
    int max_in, max_out;
    bool dropped, too_large;
};

static void wjobs_add(WJobsPlan& p, float* tape, float* dW, float* db, int OUT, int IN, int ldw, int cap) {
    if (!dW || OUT <= 0 || IN <= 0) return;
    if (p.t.n >= STRIVE_WJOBS_MAX) { p.dropped = true; return; }      // (27 blocks today; a dropped block would silently fall back to atomics)
    const int j = p.t.n++;
    p.t.count[j] = 0;
    p.t.OUT[j] = OUT; p.t.IN[j] = IN; p.t.ldw[j] = ldw; p.t.cap[j] = cap;
    p.t.dW[j] = dW; p.t.db[j] = db;
    p.t.G[j] = tape + p.tape_floats; p.tape_floats += (size_t)cap * OUT;
    p.t.A[j] = tape + p.tape_floats; p.tape_floats += (size_t)cap * IN;
    p.max_in = IN > p.max_in ? IN : p.max_in;
    p.max_out = OUT > p.max_out ? OUT : p.max_out;
}

static void wjobs_add_mlp(WJobsPlan& p, float* tape, const StriveMLP& m, const MLPGradDev& g, int cap, bool skip_first) {
    for (int l = skip_first ? 1 : 0; l < m.nlayers; ++l) wjobs_add(p, tape, g.w[l], g.b[l], m.dims[l + 1], m.dims[l], m.dims[l], cap);
}


// the blocks of one SceneInteractionNet: node jobs get node_cap rows, edge jobs edge_cap
static void wjobs_add_gnn(WJobsPlan& p, float* tape, const StriveGNN& g, const GNNGradDev& gr, int node_cap, int edge_cap) {
    wjobs_add_mlp(p, tape, g.mlp_in, gr.mlp_in, node_cap, false);
    wjobs_add_mlp(p, tape, g.update, gr.update, node_cap, false);
    wjobs_add_mlp(p, tape, g.mlp_out, gr.mlp_out, node_cap, false);
    wjobs_add_mlp(p, tape, g.edge, gr.edge, edge_cap, true);
    // the factorised layer 0 of the edge network: [x_i | x_j | sem_i | sem_j | rel] column blocks (gnn_bwd_kernels.h)
    const int D = g.D, NC = g.NC, H = STRIVE_HID, EIN = g.edge.dims[0];
    float* w0 = gr.edge.w[0];
    if (w0) {
        wjobs_add(p, tape, w0, gr.edge.b[0], H, D, EIN, node_cap);
        wjobs_add(p, tape, w0 + D, nullptr, H, D, EIN, node_cap);
        wjobs_add(p, tape, w0 + 2 * D, nullptr, H, NC, EIN, node_cap);
        wjobs_add(p, tape, w0 + 2 * D + NC, nullptr, H, NC, EIN, node_cap);
        wjobs_add(p, tape, w0 + 2 * D + 2 * NC, nullptr, H, 4, EIN, edge_cap);
    }
}

static __global__ void wjobs_upload_kernel(WJobTable* dst, WJobTable src) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = src;
}
