// Device-side pieces of the scene interaction network (reference src/models/interaction_net.py:16-218)
// shared by the stand-alone GNN forward (mlp_gnn.hip) and the decoder rollout (rollout.hip).
//
// The reference gathers six per-edge tensors with index_select, runs the edge MLP on E = sum n(n-1) rows of
// width 2(D+NC)+4, and reduces with torch_scatter's scatter-max.  Scenes are cliques, so here:
//   * layer 0 of the edge MLP is factorised into per-node partials
//         e1_ij = P_i + Q_j + W_rel * rel_ij,   P_i = W_xi x_i + W_si sem_i + b,   Q_j = W_xj x_j + W_sj sem_j
//     (exact up to fp32 re-association), computed once per node by the node kernel;
//   * one workgroup per TARGET node streams its sources in chunks of RB rows through the remaining edge
//     layers with activations in LDS and keeps the running max / arg-max in registers -- no edge tensor
//     ever exists in HBM, and isolated nodes get the aggregate 0 the reference relies on (:188).
#pragma once
#include "mlp_dev.h"

struct GNNDev {
    MLPDev mlp_in, edge, update, mlp_out;
    int D, NC;
};

static inline GNNDev gnn_dev(const StriveGNN& g) {
    GNNDev d;
    d.mlp_in = mlp_dev(g.mlp_in);
    d.edge = mlp_dev(g.edge);
    d.update = mlp_dev(g.update);
    d.mlp_out = mlp_dev(g.mlp_out);
    d.D = g.D;
    d.NC = g.NC;
    return d;
}

struct ScenesDev {
    int NA, NS, B, max_n;
    const int32_t* ptr;
    const int32_t* scene_of;
};

static inline ScenesDev scenes_dev(const StriveScenes& s) {
    ScenesDev d;
    d.NA = s.NA;
    d.NS = s.NS;
    d.B = s.B;
    d.max_n = s.max_n;
    d.ptr = s.ptr;
    d.scene_of = s.scene_of;
    return d;
}

// Node features assembled from up to 5 column segments; a segment is per row (R = NA*NS rows) or per agent.
struct FeatSrc {
    const float* p[5];
    int w[5];
    int per_agent[5];
    int n;
};

__device__ __forceinline__ int feat_width(const FeatSrc& f) {
    int t = 0;
    for (int i = 0; i < f.n; ++i) t += f.w[i];
    return t;
}

// rows r0..r0+RB-1 of the concatenated feature -> LDS [RB][ld] (rows >= R are zero filled)
template <int RB>
__device__ __forceinline__ void gather_features(const FeatSrc& f, int r0, int R, int NS, float* dst, int ld, int tid,
                                                int nthreads) {
    int col0 = 0;
    for (int s = 0; s < f.n; ++s) {
        const int w = f.w[s];
        for (int i = tid; i < RB * w; i += nthreads) {
            const int rr = i / w, k = i - rr * w;
            const int r = r0 + rr;
            float v = 0.f;
            if (r < R) {
                const int src = f.per_agent[s] ? (r / NS) : r;
                v = f.p[s][(size_t)src * w + k];
            }
            dst[rr * ld + col0 + k] = v;
        }
        col0 += w;
    }
    // zero the alignment padding so float4 reads past the feature width are defined
    const int F = col0;
    for (int i = tid; i < RB * (ld - F); i += nthreads) {
        const int rr = i / (ld - F), k = i - rr * (ld - F);
        dst[rr * ld + F + k] = 0.f;
    }
}

// pose (px,py,pc,ps) expressed in frame (fx,fy,c,s): reference src/utils/transforms.py:78-139, forward branch.
__device__ __forceinline__ void rel_pose(const float* fr, const float* po, float* out) {
    const float dx = po[0] - fr[0], dy = po[1] - fr[1];
    const float c = fr[2], s = fr[3];
    out[0] = c * dx + s * dy;
    out[1] = -s * dx + c * dy;
    out[2] = po[2] * c + po[3] * s;
    out[3] = po[3] * c - po[2] * s;
}

// Adjoint of rel_pose: accumulates d(frame) and d(pose) given d(out).
__device__ __forceinline__ void rel_pose_bwd(const float* fr, const float* po, const float* g, float* dfr, float* dpo) {
    const float dx = po[0] - fr[0], dy = po[1] - fr[1];
    const float c = fr[2], s = fr[3];
    const float gdx = c * g[0] - s * g[1];
    const float gdy = s * g[0] + c * g[1];
    dpo[0] += gdx;
    dpo[1] += gdy;
    dfr[0] -= gdx;
    dfr[1] -= gdy;
    // d/dc, d/ds
    dfr[2] += g[0] * dx + g[1] * dy + g[2] * po[2] + g[3] * po[3];
    dfr[3] += g[0] * dy - g[1] * dx + g[2] * po[3] - g[3] * po[2];
    dpo[2] += g[2] * c - g[3] * s;
    dpo[3] += g[2] * s + g[3] * c;
}
