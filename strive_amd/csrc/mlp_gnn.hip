// Stand-alone MLP forward and scene-interaction-network forward (used by embed(): past/future encoders,
// prior and posterior networks), plus the three forward kernels the decoder rollout shares.
#include "gnn_kernels.h"

// ---------------------------------------------------------------------------------------------
// MLP forward on a (rows, F) matrix
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MLPDev m, const float* __restrict__ x, int rows,
                                                        float* __restrict__ y) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int F = m.dims[0], O = m.dims[m.nlayers];
    const int in_ld = (F + 3) & ~3;
    float* s_in = smem;
    float* s_pre = s_in + RB_NODE * in_ld;
    float* s_act = s_pre + (STRIVE_MAX_LAYERS - 1) * RB_NODE * HLD;
    float* s_out = s_act + RB_NODE * HLD;
    const int tid = threadIdx.x, r0 = blockIdx.x * RB_NODE;
    FeatSrc f;
    f.n = 1;
    f.p[0] = x;
    f.w[0] = F;
    f.per_agent[0] = 0;
    gather_features<RB_NODE>(f, r0, rows, 1, s_in, in_ld, tid, 256);
    __syncthreads();
    mlp_forward_lds<RB_NODE>(m, s_in, in_ld, s_pre, s_act, s_out, HLD, false, tid, 256);
    for (int i = tid; i < RB_NODE * O; i += 256) {
        const int rr = i / O, c = i - rr * O;
        if (r0 + rr < rows) y[(size_t)(r0 + rr) * O + c] = s_out[rr * HLD + c];
    }
}

extern "C" int strive_mlp_fwd(const StriveMLP* mlp, const float* x, int32_t rows, float* y, strive_stream_t stream) {
    STRIVE_CHECK_ARG(mlp && x && y, "null argument");
    STRIVE_CHECK_ARG(mlp->nlayers >= 2 && mlp->nlayers <= STRIVE_MAX_LAYERS, "unsupported layer count");
    for (int l = 1; l < mlp->nlayers; ++l) STRIVE_CHECK_ARG(mlp->dims[l] == STRIVE_HID, "hidden width must be 128");
    STRIVE_CHECK_ARG(mlp->dims[mlp->nlayers] <= STRIVE_HID && mlp->dims[0] <= 512, "layer too wide");
    if (rows <= 0) return 0;
    const int in_ld = (mlp->dims[0] + 3) & ~3;
    const size_t lds = (size_t)(RB_NODE * in_ld + (STRIVE_MAX_LAYERS - 1) * RB_NODE * HLD + 2 * RB_NODE * HLD) * 4;
    hipLaunchKernelGGL(mlp_fwd_kernel, dim3((rows + RB_NODE - 1) / RB_NODE), dim3(256), lds, (hipStream_t)stream, mlp_dev(*mlp), x,
                       rows, y);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// SceneInteractionNet forward
// ---------------------------------------------------------------------------------------------
extern "C" size_t strive_gnn_workspace_bytes(const StriveGNN* gnn, const StriveScenes* sc) {
    if (!gnn || !sc) return 0;
    const size_t R = (size_t)sc->NA * sc->NS;
    size_t b = 0;
    b += strive_align_up(R * gnn->D * 4, 256);          // X
    b += 2 * strive_align_up(R * STRIVE_HID * 4, 256);  // P, Q
    b += strive_align_up(R * gnn->D * 4, 256);          // A
    b += strive_align_up(R * gnn->D * 4, 256);          // ARG
    return b;
}

extern "C" int strive_gnn_fwd(const StriveGNN* gnn, const StriveScenes* sc, const float* x, const float* pos,
                              const float* sem, float* out, void* ws, size_t ws_bytes, strive_stream_t stream) {
    STRIVE_CHECK_ARG(gnn && sc && x && pos && sem && out && ws, "null argument");
    STRIVE_CHECK_ARG(ws_bytes >= strive_gnn_workspace_bytes(gnn, sc), "workspace too small");
    const int R = sc->NA * sc->NS;
    if (R == 0) return 0;
    StriveArena ar(ws, ws_bytes);
    GnnBuffers gb;
    gb.X = ar.take<float>((size_t)R * gnn->D);
    gb.P = ar.take<float>((size_t)R * STRIVE_HID);
    gb.Q = ar.take<float>((size_t)R * STRIVE_HID);
    gb.A = ar.take<float>((size_t)R * gnn->D);
    gb.ARG = ar.take<int32_t>((size_t)R * gnn->D);
    FeatSrc f;
    f.n = 1;
    f.p[0] = x;
    f.w[0] = gnn->mlp_in.dims[0];
    f.per_agent[0] = 0;
    int rc = gnn_forward_launch(*gnn, *sc, f, pos, sem, gb, out, (hipStream_t)stream);
    if (rc) return rc;
    STRIVE_CHECK_LAUNCH();
    return 0;
}
