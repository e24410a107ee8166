// Forward kernels of the scene interaction network, shared by mlp_gnn.hip and rollout.hip.
// (kernels live in a header so both translation units can launch them without relocatable device code)
#pragma once
#include "gnn_dev.h"

struct GnnBuffers {
    float* X;       // (R, D)   node embeddings after mlp_in
    float* P;       // (R, 128) target-side partial of edge layer 0 (bias included)
    float* Q;       // (R, 128) source-side partial of edge layer 0
    float* A;       // (R, D)   max-aggregated messages
    int32_t* ARG;   // (R, D)   source row that attained the max (-1: no incoming edge)
    // optional (null = not kept): pre-LayerNorm outputs of the hidden layers, kept for the backward instead of being
    // recomputed there (HBM is 288 GB; these are 1 KB per node and 1 KB per edge)
    float* PRE_IN;  // (R, 2, 128)          mlp_in layers 0, 1
    float* PRE_E;   // (R * max_n, 2, 128)  edge MLP layers 0, 1; slot = target_row * max_n + local source index
};

// LDS carve-ups (floats)
__host__ __device__ static inline int ld4(int n) { return (n + 3) & ~3; }

struct Node1Lds {
    float *in, *pre, *act, *xs, *po;
    __device__ Node1Lds(float* base, int in_ld, int xs_ld) {
        in = base;
        pre = in + RB_NODE * in_ld;
        act = pre + 2 * RB_NODE * HLD;
        xs = act + RB_NODE * HLD;
        po = xs + RB_NODE * xs_ld;
    }
    static size_t bytes(int in_ld, int xs_ld) { return (size_t)(RB_NODE * in_ld + 4 * RB_NODE * HLD + RB_NODE * xs_ld) * 4; }
};

// ---------------------------------------------------------------------------------------------
// node kernel 1: features -> mlp_in -> x ; edge layer-0 partials P, Q.   grid = ceil(R/RB_NODE)
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void gnn_node1_kernel(GNNDev g, int NS, FeatSrc f, const float* __restrict__ sem,
                                                                 GnnBuffers gb, int R) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int F = g.mlp_in.dims[0], D = g.D, NC = g.NC;
    const int in_ld = ld4(F), xs_ld = ld4(D + NC);
    Node1Lds L(smem, in_ld, xs_ld);
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE;
    gather_features<RB_NODE>(f, r0, R, NS, L.in, in_ld, tid, 256);
    __syncthreads();
    mlp_forward_lds<RB_NODE>(g.mlp_in, L.in, in_ld, L.pre, L.act, L.xs, xs_ld, false, tid, 256);
    if (gb.PRE_IN) {
        for (int i = tid; i < RB_NODE * 2 * STRIVE_HID; i += 256) {
            const int rr = i / (2 * STRIVE_HID), rem = i - rr * 2 * STRIVE_HID, l = rem / STRIVE_HID, c = rem - l * STRIVE_HID;
            if (r0 + rr < R) gb.PRE_IN[(size_t)(r0 + rr) * 2 * STRIVE_HID + rem] = L.pre[(size_t)l * RB_NODE * HLD + rr * HLD + c];
        }
    }
    // append sem, publish x
    for (int i = tid; i < RB_NODE * (xs_ld - D); i += 256) {
        const int rr = i / (xs_ld - D), k = i - rr * (xs_ld - D);
        const int r = r0 + rr;
        L.xs[rr * xs_ld + D + k] = (r < R && k < NC) ? sem[(size_t)(r / NS) * NC + k] : 0.f;
    }
    for (int i = tid; i < RB_NODE * D; i += 256) {
        const int rr = i / D, c = i - rr * D;
        if (r0 + rr < R) gb.X[(size_t)(r0 + rr) * D + c] = L.xs[rr * xs_ld + c];
    }
    __syncthreads();
    // edge layer 0, rows of the transposed weight: [x_i (D) | x_j (D) | sem_i (NC) | sem_j (NC) | rel (4)] x 128
    const float* Wt = g.edge.wt[0];
    const int H = STRIVE_HID;
    dense_lds<RB_NODE, false>(L.xs, xs_ld, D, Wt, H, g.edge.b[0], L.po, HLD, H, tid, 256);
    __syncthreads();
    dense_lds<RB_NODE, true>(L.xs + D, xs_ld, NC, Wt + (size_t)(2 * D) * H, H, nullptr, L.po, HLD, H, tid, 256);
    __syncthreads();
    for (int i = tid; i < RB_NODE * H; i += 256) {
        const int rr = i / H, c = i - rr * H;
        if (r0 + rr < R) gb.P[(size_t)(r0 + rr) * H + c] = L.po[rr * HLD + c];
    }
    __syncthreads();
    dense_lds<RB_NODE, false>(L.xs, xs_ld, D, Wt + (size_t)D * H, H, nullptr, L.po, HLD, H, tid, 256);
    __syncthreads();
    dense_lds<RB_NODE, true>(L.xs + D, xs_ld, NC, Wt + (size_t)(2 * D + NC) * H, H, nullptr, L.po, HLD, H, tid, 256);
    __syncthreads();
    for (int i = tid; i < RB_NODE * H; i += 256) {
        const int rr = i / H, c = i - rr * H;
        if (r0 + rr < R) gb.Q[(size_t)(r0 + rr) * H + c] = L.po[rr * HLD + c];
    }
}

// ---------------------------------------------------------------------------------------------
// edge kernel: one workgroup per target row; sources streamed in chunks of RB.   grid = R
// ---------------------------------------------------------------------------------------------
struct EdgeLds {
    float *rel, *pre, *act, *m;
    int* src;
    __device__ EdgeLds(float* base) {
        rel = base;                    // [RB_EDGE][4]
        pre = rel + RB_EDGE * 4;            // [2][RB_EDGE][HLD]
        act = pre + 2 * RB_EDGE * HLD;      // [RB_EDGE][HLD]
        m = act + RB_EDGE * HLD;            // [RB_EDGE][HLD]
        src = (int*)(m + RB_EDGE * HLD);    // [RB_EDGE]
    }
    static size_t bytes() { return (size_t)(RB_EDGE * 4 + 4 * RB_EDGE * HLD + RB_EDGE) * 4; }
};

// Fill one chunk: source rows, relative poses (NaN -> 0), and the factorised edge layer 0 into pre[0].
// Returns the number of valid sources in the chunk (block-uniform).
// `stored` (optional): the PRE_E table a forward pass kept -- then BOTH hidden layers' pre-activations of the chunk are
// loaded into L.pre and the caller skips the edge MLP forward.
__device__ __forceinline__ int edge_chunk_setup(const GNNDev& g, const ScenesDev& sc, const float* __restrict__ pos,
                                                const GnnBuffers& gb, int r, int chunk, EdgeLds& L, unsigned* nanmask_out,
                                                int tid, const float* __restrict__ stored = nullptr) {
    const int NS = sc.NS;
    const int a = r / NS, s = r - a * NS;
    const int b = sc.scene_of[a];
    const int lo = sc.ptr[b], n = sc.ptr[b + 1] - lo;
    const int nsrc = n - 1;
    const int j0 = chunk * RB_EDGE;
    const int nv = (nsrc - j0) < RB_EDGE ? (nsrc - j0) : RB_EDGE;
    if (tid < RB_EDGE) {
        int srow = -1;
        float rel[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned nm = 0;
        if (tid < nv) {
            int jl = j0 + tid;              // index among the scene's agents, skipping the target itself
            if (jl >= a - lo) jl += 1;
            srow = (lo + jl) * NS + s;
            rel_pose(pos + (size_t)r * 4, pos + (size_t)srow * 4, rel);
            for (int d = 0; d < 4; ++d)
                if (rel[d] != rel[d]) { rel[d] = 0.f; nm |= 1u << d; }   // interaction_net.py:162
        }
        L.src[tid] = srow;
        for (int d = 0; d < 4; ++d) L.rel[tid * 4 + d] = rel[d];
        if (nanmask_out) nanmask_out[tid] = nm;
    }
    __syncthreads();
    const int H = STRIVE_HID;
    if (stored) {
        for (int i = tid; i < RB_EDGE * 2 * H; i += 256) {
            const int jr = i / (2 * H), rem = i - jr * 2 * H, l = rem / H, c = rem - l * H;
            float v = 0.f;
            if (jr < nv) v = stored[((size_t)r * sc.max_n + (L.src[jr] / NS - lo)) * 2 * H + rem];
            L.pre[(size_t)l * RB_EDGE * HLD + jr * HLD + c] = v;
        }
        __syncthreads();
        return nv;
    }
    const float* Wrel = g.edge.wt[0] + (size_t)(2 * g.D + 2 * g.NC) * H;
    for (int i = tid; i < RB_EDGE * H; i += 256) {
        const int jr = i / H, c = i - jr * H;
        float v = 0.f;
        if (jr < nv) {
            const int srow = L.src[jr];
            v = gb.P[(size_t)r * H + c] + gb.Q[(size_t)srow * H + c];
            v = fmaf(L.rel[jr * 4 + 0], Wrel[c], v);
            v = fmaf(L.rel[jr * 4 + 1], Wrel[H + c], v);
            v = fmaf(L.rel[jr * 4 + 2], Wrel[2 * H + c], v);
            v = fmaf(L.rel[jr * 4 + 3], Wrel[3 * H + c], v);
        }
        L.pre[jr * HLD + c] = v;
    }
    __syncthreads();
    return nv;
}

static __global__ __launch_bounds__(256) void gnn_edge_kernel(GNNDev g, ScenesDev sc, const float* __restrict__ pos,
                                                                GnnBuffers gb) {
    HIP_DYNAMIC_SHARED(float, smem)
    EdgeLds L(smem);
    const int r = blockIdx.x, tid = threadIdx.x, D = g.D;
    const int a = r / sc.NS;
    const int b = sc.scene_of[a];
    const int nsrc = sc.ptr[b + 1] - sc.ptr[b] - 1;
    float best = 0.f;
    int arg = -1;
    const int nchunks = (nsrc + RB_EDGE - 1) / RB_EDGE;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int nv = edge_chunk_setup(g, sc, pos, gb, r, ch, L, nullptr, tid);
        mlp_forward_lds<RB_EDGE>(g.edge, nullptr, 0, L.pre, L.act, L.m, HLD, true, tid, 256);
        if (gb.PRE_E) {
            const int lo = sc.ptr[b];
            for (int i = tid; i < RB_EDGE * 2 * STRIVE_HID; i += 256) {
                const int jr = i / (2 * STRIVE_HID), rem = i - jr * 2 * STRIVE_HID, l = rem / STRIVE_HID, c = rem - l * STRIVE_HID;
                if (jr < nv) {
                    const int jl = L.src[jr] / sc.NS - lo;
                    gb.PRE_E[((size_t)r * sc.max_n + jl) * 2 * STRIVE_HID + rem] = L.pre[(size_t)l * RB_EDGE * HLD + jr * HLD + c];
                }
            }
        }
        if (tid < D) {
            for (int jr = 0; jr < nv; ++jr) {
                const float v = L.m[jr * HLD + tid];
                if (arg < 0 || v > best) { best = v; arg = L.src[jr]; }
            }
        }
        __syncthreads();
    }
    if (tid < D) {
        gb.A[(size_t)r * D + tid] = (arg < 0) ? 0.f : best;
        gb.ARG[(size_t)r * D + tid] = arg;
    }
}

// ---------------------------------------------------------------------------------------------
// node kernel 2: [x, aggr, sem] -> update MLP -> x' -> mlp_out -> out.   grid = ceil(R/RB)
// ---------------------------------------------------------------------------------------------
struct Node2Lds {
    float *in, *pre_u, *act, *xp, *pre_o, *out;
    __device__ Node2Lds(float* base, int in_ld) {
        in = base;                       // [RB_NODE][in_ld]   (x | aggr | sem)
        pre_u = in + RB_NODE * in_ld;         // [1][RB_NODE][HLD]
        act = pre_u + RB_NODE * HLD;          // [RB_NODE][HLD]
        xp = act + RB_NODE * HLD;             // [RB_NODE][HLD]    x' (D wide)
        pre_o = xp + RB_NODE * HLD;           // [2][RB_NODE][HLD]
        out = pre_o + 2 * RB_NODE * HLD;      // [RB_NODE][HLD]
    }
    static size_t floats(int in_ld) { return (size_t)(RB_NODE * in_ld + 6 * RB_NODE * HLD); }
};

// Leaves update/mlp_out pre-activations and outputs in LDS (the rollout backward re-uses them).
__device__ __forceinline__ void node2_forward(const GNNDev& g, int NS, const float* __restrict__ X,
                                              const float* __restrict__ A, const float* __restrict__ sem, int r0, int R,
                                              Node2Lds& L, int in_ld, int tid) {
    const int D = g.D, NC = g.NC;
    for (int i = tid; i < RB_NODE * in_ld; i += 256) {
        const int rr = i / in_ld, k = i - rr * in_ld;
        const int r = r0 + rr;
        float v = 0.f;
        if (r < R) {
            if (k < D) v = X[(size_t)r * D + k];
            else if (k < 2 * D) v = A[(size_t)r * D + (k - D)];
            else if (k < 2 * D + NC) v = sem[(size_t)(r / NS) * NC + (k - 2 * D)];
        }
        L.in[i] = v;
    }
    __syncthreads();
    mlp_forward_lds<RB_NODE>(g.update, L.in, in_ld, L.pre_u, L.act, L.xp, HLD, false, tid, 256);
    mlp_forward_lds<RB_NODE>(g.mlp_out, L.xp, HLD, L.pre_o, L.act, L.out, HLD, false, tid, 256);
}

static __global__ __launch_bounds__(256) void gnn_node2_kernel(GNNDev g, int NS, const float* __restrict__ sem, GnnBuffers gb,
                                                                 float* __restrict__ out, int R) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int in_ld = ld4(2 * g.D + g.NC);
    Node2Lds L(smem, in_ld);
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE;
    node2_forward(g, NS, gb.X, gb.A, sem, r0, R, L, in_ld, tid);
    const int O = g.mlp_out.dims[g.mlp_out.nlayers];
    for (int i = tid; i < RB_NODE * O; i += 256) {
        const int rr = i / O, c = i - rr * O;
        if (r0 + rr < R) out[(size_t)(r0 + rr) * O + c] = L.out[rr * HLD + c];
    }
}

static inline int gnn_check(const StriveGNN& g) {
    const StriveMLP* ms[4] = {&g.mlp_in, &g.edge, &g.update, &g.mlp_out};
    for (int i = 0; i < 4; ++i)
        for (int l = 1; l < ms[i]->nlayers; ++l)
            if (ms[i]->dims[l] != STRIVE_HID) { strive_set_error("gnn: hidden width must be 128"); return -1; }
    if (g.mlp_in.nlayers != 3 || g.edge.nlayers != 3 || g.update.nlayers != 2 || g.mlp_out.nlayers != 3) {
        strive_set_error("gnn: unexpected MLP depths");
        return -1;
    }
    if ((g.D != 64 && g.D != 128) || g.NC < 1 || g.NC > 16) { strive_set_error("gnn: unsupported D/NC"); return -1; }
    if (g.mlp_in.dims[0] > 512) { strive_set_error("gnn: input feature too wide"); return -1; }
    return 0;
}

static inline int gnn_forward_launch(const StriveGNN& g, const StriveScenes& sc, const FeatSrc& f, const float* pos,
                                     const float* sem, const GnnBuffers& gb, float* out, hipStream_t stream) {
    if (gnn_check(g)) return -1;
    const int R = sc.NA * sc.NS;
    const GNNDev gd = gnn_dev(g);
    const ScenesDev sd = scenes_dev(sc);
    const int in_ld = ld4(g.mlp_in.dims[0]), xs_ld = ld4(g.D + g.NC);
    const int nb = (R + RB_NODE - 1) / RB_NODE;
    hipLaunchKernelGGL(gnn_node1_kernel, dim3(nb), dim3(256), Node1Lds::bytes(in_ld, xs_ld), stream, gd, sc.NS, f, sem, gb, R);
    hipLaunchKernelGGL(gnn_edge_kernel, dim3(R), dim3(256), EdgeLds::bytes(), stream, gd, sd, pos, gb);
    if (out) {
        const int in2 = ld4(2 * g.D + g.NC);
        hipLaunchKernelGGL(gnn_node2_kernel, dim3(nb), dim3(256), Node2Lds::floats(in2) * 4, stream, gd, sc.NS, sem, gb, out, R);
    }
    return 0;
}
