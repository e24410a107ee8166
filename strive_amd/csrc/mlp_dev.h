// Workgroup-cooperative building blocks for the small dense networks on the hot path
// (MLP = Linear -> LayerNorm -> ReLU -> Linear ..., reference src/models/common.py:8-44).
//
// A workgroup pushes a block of RB rows through a layer with the activations resident in LDS and the
// weights streamed from L2 (they are shared by every workgroup of a launch and stay cache resident):
//   thread item = (output channel c, group of 4 rows); per k it issues one coalesced weight load
//   (consecutive c across lanes) and broadcast LDS reads of the 4 rows' activations (read as float4
//   over k).  These layers are tiny (<= 136x128) and latency/launch bound at the agent counts of
//   interest; the map CNN (map_cnn.hip) carries 98 % of the path's FLOPs.
#pragma once
#include "common.h"

#define RB_EDGE 16     // source rows per chunk in the per-target edge kernels
#define RB_NODE 4      // rows per workgroup in the per-node kernels: 128 workgroups for 512 agents (see dense_lds)
#define RPT 4          // rows per thread item
#define LN_EPS 1e-5f

// Scratch for the k-split partial sums of dense_lds (one per kernel: not a template).
#define KSPLIT_CAP (4 * 192)
__device__ __forceinline__ float* ksplit_buf() {
    __shared__ float s_part[KSPLIT_CAP];
    return s_part;
}

// out[r][c] (+)= bias[c] + sum_k in[r][k] * Wt[k*ldw + c],  r < RB, c < OUT.
// in_ld, out_ld multiples of 4; `in` 16-byte aligned; `out` must not alias `in`; called by all threads of the workgroup.
// Thread item = (output channel c, group of 4 rows, k half).  A workgroup of RB rows has only OUT * RB/4 (channel, row
// group) items; when that leaves half of the threads idle (RB = 4, OUT <= 128) the k range is split in two and the two
// partial sums are added through LDS: the per-CU weight traffic (every item streams its weight column through the L1)
// and the FMA count per thread are both halved -- these layers are bound by exactly those two, on the 64-128 CUs
// that a batch of 512 agents occupies.
template <int RB, bool ACCUM>
__device__ __forceinline__ void dense_lds(const float* in, int in_ld, int IN, const float* __restrict__ Wt, int ldw,
                                          const float* __restrict__ bias, float* out, int out_ld, int OUT, int tid,
                                          int nthreads) {
    const int base_items = OUT * (RB / RPT);
    const bool split = (2 * base_items <= nthreads) && (RB * OUT <= KSPLIT_CAP) && IN >= 32;
    const int items = split ? 2 * base_items : base_items;
    const int kmid = split ? ((IN / 2) & ~15) : IN;
    float* s_part = ksplit_buf();
    for (int item = tid; item < items; item += nthreads) {
        const int kh = item / base_items;               // 0, or 1 = upper k half
        const int it = item - kh * base_items;
        const int c = it % OUT;
        const int r0 = (it / OUT) * RPT;
        const int kbeg = kh ? kmid : 0, kend = kh ? IN : kmid;
        float acc[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            float v = 0.f;
            if (kh == 0) {
                v = bias ? bias[c] : 0.f;
                if (ACCUM) v += out[(r0 + i) * out_ld + c];
            }
            acc[i] = v;
        }
        int k = kbeg;
        // 16 weight loads are issued back to back before any is used
        for (; k + 15 < kend; k += 16) {
            float w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = Wt[(size_t)(k + q) * ldw + c];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const float4 a = *reinterpret_cast<const float4*>(&in[(r0 + i) * in_ld + k + 4 * q4]);
                    acc[i] = fmaf(a.x, w[4 * q4 + 0], acc[i]);
                    acc[i] = fmaf(a.y, w[4 * q4 + 1], acc[i]);
                    acc[i] = fmaf(a.z, w[4 * q4 + 2], acc[i]);
                    acc[i] = fmaf(a.w, w[4 * q4 + 3], acc[i]);
                }
            }
        }
        for (; k + 3 < kend; k += 4) {
            const float w0 = Wt[(size_t)(k + 0) * ldw + c];
            const float w1 = Wt[(size_t)(k + 1) * ldw + c];
            const float w2 = Wt[(size_t)(k + 2) * ldw + c];
            const float w3 = Wt[(size_t)(k + 3) * ldw + c];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(&in[(r0 + i) * in_ld + k]);
                acc[i] = fmaf(a.x, w0, acc[i]);
                acc[i] = fmaf(a.y, w1, acc[i]);
                acc[i] = fmaf(a.z, w2, acc[i]);
                acc[i] = fmaf(a.w, w3, acc[i]);
            }
        }
        for (; k < kend; ++k) {
            const float w0 = Wt[(size_t)k * ldw + c];
#pragma unroll
            for (int i = 0; i < RPT; ++i) acc[i] = fmaf(in[(r0 + i) * in_ld + k], w0, acc[i]);
        }
        if (!split) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) out[(r0 + i) * out_ld + c] = acc[i];
        } else if (kh == 1) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) s_part[(r0 + i) * OUT + c] = acc[i];
        } else {
#pragma unroll
            for (int i = 0; i < RPT; ++i) out[(r0 + i) * out_ld + c] = acc[i];      // lower half; the upper half is added below
        }
    }
    if (split) {
        __syncthreads();
        for (int it = tid; it < RB * OUT; it += nthreads) {
            const int r = it / OUT, c = it - r * OUT;
            out[r * out_ld + c] += s_part[it];
        }
    }
}

// y = relu(layer_norm(x)) row-wise over N channels (one wave per row, two-pass mean/variance).
template <int RB>
__device__ __forceinline__ void ln_relu_rows(const float* x, int x_ld, float* y, int y_ld, int N,
                                             const float* __restrict__ g, const float* __restrict__ b, int tid,
                                             int nthreads) {
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
    for (int r = wave; r < RB; r += nw) {
        float s = 0.f;
        for (int c = lane; c < N; c += 64) s += x[r * x_ld + c];
        const float mean = wave_sum(s) / (float)N;
        float v = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float d = x[r * x_ld + c] - mean;
            v = fmaf(d, d, v);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)N + LN_EPS);
        for (int c = lane; c < N; c += 64) {
            const float t = (x[r * x_ld + c] - mean) * rstd * g[c] + b[c];
            y[r * y_ld + c] = fmaxf(t, 0.f);
        }
    }
}

// Backward of y = relu(layer_norm(x)):  dx from dy, recomputing the normalisation from x (pre-LN).
// dx may alias dy.
template <int RB>
__device__ __forceinline__ void ln_relu_bwd_rows(const float* x, int x_ld, const float* dy, int dy_ld, float* dx,
                                                 int dx_ld, int N, const float* __restrict__ g,
                                                 const float* __restrict__ b, int tid, int nthreads) {
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
    for (int r = wave; r < RB; r += nw) {
        float s = 0.f;
        for (int c = lane; c < N; c += 64) s += x[r * x_ld + c];
        const float mean = wave_sum(s) / (float)N;
        float v = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float d = x[r * x_ld + c] - mean;
            v = fmaf(d, d, v);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)N + LN_EPS);
        float m1 = 0.f, m2 = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float xh = (x[r * x_ld + c] - mean) * rstd;
            const float pre = xh * g[c] + b[c];
            const float gg = (pre > 0.f ? dy[r * dy_ld + c] : 0.f) * g[c];
            m1 += gg;
            m2 = fmaf(gg, xh, m2);
        }
        m1 = wave_sum(m1) / (float)N;
        m2 = wave_sum(m2) / (float)N;
        for (int c = lane; c < N; c += 64) {
            const float xh = (x[r * x_ld + c] - mean) * rstd;
            const float pre = xh * g[c] + b[c];
            const float gg = (pre > 0.f ? dy[r * dy_ld + c] : 0.f) * g[c];
            dx[r * dx_ld + c] = rstd * (gg - m1 - xh * m2);
        }
    }
}

// Device view of a StriveMLP (passed by value into kernels).
struct MLPDev {
    int nlayers;
    int dims[STRIVE_MAX_LAYERS + 1];
    const float* w[STRIVE_MAX_LAYERS];
    const float* wt[STRIVE_MAX_LAYERS];
    const float* b[STRIVE_MAX_LAYERS];
    const float* ln_g[STRIVE_MAX_LAYERS];
    const float* ln_b[STRIVE_MAX_LAYERS];
};

static inline MLPDev mlp_dev(const StriveMLP& m) {
    MLPDev d;
    d.nlayers = m.nlayers;
    for (int i = 0; i <= STRIVE_MAX_LAYERS; ++i) d.dims[i] = m.dims[i];
    for (int i = 0; i < STRIVE_MAX_LAYERS; ++i) {
        d.w[i] = m.w[i];
        d.wt[i] = m.wt[i];
        d.b[i] = m.b[i];
        d.ln_g[i] = m.ln_g[i];
        d.ln_b[i] = m.ln_b[i];
    }
    return d;
}

#define HLD 132   // leading dimension of 128-wide hidden buffers (padded, multiple of 4)

// Forward through an MLP whose hidden widths are all 128.
//   in  : LDS [RB][in_ld]  (layer-0 input)
//   pre : LDS [nlayers-1][RB][HLD]  pre-LayerNorm outputs of the hidden layers (kept for the backward)
//   act : LDS [RB][HLD]   scratch for the post-ReLU activations (overwritten layer by layer)
//   out : LDS [RB][out_ld]
// first_done: layer 0's linear output is already in pre[0] (used by the factorised edge layer).
template <int RB>
__device__ __forceinline__ void mlp_forward_lds(const MLPDev& m, const float* in, int in_ld, float* pre, float* act,
                                                float* out, int out_ld, bool first_done, int tid, int nthreads) {
    const int L = m.nlayers;
    if (!first_done) {
        dense_lds<RB, false>(in, in_ld, m.dims[0], m.wt[0], m.dims[1], m.b[0], (L == 1) ? out : pre, (L == 1) ? out_ld : HLD,
                         m.dims[1], tid, nthreads);
        __syncthreads();
    }
    for (int l = 1; l < L; ++l) {
        float* p = pre + (size_t)(l - 1) * RB * HLD;
        ln_relu_rows<RB>(p, HLD, act, HLD, m.dims[l], m.ln_g[l - 1], m.ln_b[l - 1], tid, nthreads);
        __syncthreads();
        const bool last = (l == L - 1);
        dense_lds<RB, false>(act, HLD, m.dims[l], m.wt[l], m.dims[l + 1], m.b[l], last ? out : pre + (size_t)l * RB * HLD,
                         last ? out_ld : HLD, m.dims[l + 1], tid, nthreads);
        __syncthreads();
    }
}

// Backward (input gradient only) through the same MLP, given the `pre` buffers of a forward pass.
//   dout : LDS [RB][dout_ld] gradient w.r.t. the MLP output
//   ga, gb : LDS [RB][HLD] scratch
//   din  : LDS [RB][din_ld] gradient w.r.t. the layer-0 input; if skip_first, the gradient w.r.t. layer 0's
//          linear OUTPUT (pre[0]) is left in `ga` instead and din is untouched.
template <int RB>
__device__ __forceinline__ void mlp_backward_lds(const MLPDev& m, const float* pre, const float* dout, int dout_ld,
                                                 float* ga, float* gb, float* din, int din_ld, bool skip_first, int tid,
                                                 int nthreads) {
    const int L = m.nlayers;
    const float* g = dout;
    int g_ld = dout_ld;
    for (int l = L - 1; l >= 1; --l) {
        // gradient w.r.t. the post-ReLU activation feeding layer l: gb = g * W_l   (W_l torch layout (out,in))
        dense_lds<RB, false>(g, g_ld, m.dims[l + 1], m.w[l], m.dims[l], nullptr, gb, HLD, m.dims[l], tid, nthreads);
        __syncthreads();
        const float* p = pre + (size_t)(l - 1) * RB * HLD;
        ln_relu_bwd_rows<RB>(p, HLD, gb, HLD, ga, HLD, m.dims[l], m.ln_g[l - 1], m.ln_b[l - 1], tid, nthreads);
        __syncthreads();
        g = ga;
        g_ld = HLD;
    }
    if (!skip_first) {
        dense_lds<RB, false>(g, g_ld, m.dims[1], m.w[0], m.dims[0], nullptr, din, din_ld, m.dims[0], tid, nthreads);
        __syncthreads();
    }
}
