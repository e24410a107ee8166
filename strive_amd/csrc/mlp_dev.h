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
// dgam / dbet (optional, global memory): LayerNorm weight / bias gradients, accumulated with atomics over the first
// `nrows` rows (training path only; the latent-optimisation path passes nullptr).
template <int RB>
__device__ __forceinline__ void ln_relu_bwd_rows(const float* x, int x_ld, const float* dy, int dy_ld, float* dx,
                                                 int dx_ld, int N, const float* __restrict__ g,
                                                 const float* __restrict__ b, int tid, int nthreads,
                                                 float* dgam = nullptr, float* dbet = nullptr, int nrows = RB) {
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
            const float dn = pre > 0.f ? dy[r * dy_ld + c] : 0.f;
            const float gg = dn * g[c];
            m1 += gg;
            m2 = fmaf(gg, xh, m2);
            if (dgam && r < nrows && dn != 0.f) {
                unsafeAtomicAdd(&dgam[c], dn * xh);
                unsafeAtomicAdd(&dbet[c], dn);
            }
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

// Weight gradients of one MLP as device pointers into a flat fp32 buffer laid out in the parameter order of the
// reference module (torch named_parameters()): for every layer  W_l (dims[l+1], dims[l]) | b_l | and, for hidden layers,
// LayerNorm gamma_l | beta_l.  All null = the latent-optimisation path (no weight gradients).
struct MLPGradDev {
    float* w[STRIVE_MAX_LAYERS];
    float* b[STRIVE_MAX_LAYERS];
    float* ln_g[STRIVE_MAX_LAYERS];
    float* ln_b[STRIVE_MAX_LAYERS];
};

static inline size_t mlp_param_count(const StriveMLP& m) {
    size_t n = 0;
    for (int l = 0; l < m.nlayers; ++l) {
        n += (size_t)m.dims[l + 1] * m.dims[l] + m.dims[l + 1];
        if (l < m.nlayers - 1) n += 2 * (size_t)m.dims[l + 1];
    }
    return n;
}

// carve `flat` (may be null: all-null result); advances *flat past this MLP's parameters
static inline MLPGradDev mlp_grad_dev(const StriveMLP& m, float** flat) {
    MLPGradDev g;
    for (int l = 0; l < STRIVE_MAX_LAYERS; ++l) g.w[l] = g.b[l] = g.ln_g[l] = g.ln_b[l] = nullptr;
    if (!flat || !*flat) return g;
    float* p = *flat;
    for (int l = 0; l < m.nlayers; ++l) {
        g.w[l] = p; p += (size_t)m.dims[l + 1] * m.dims[l];
        g.b[l] = p; p += m.dims[l + 1];
        if (l < m.nlayers - 1) {
            g.ln_g[l] = p; p += m.dims[l + 1];
            g.ln_b[l] = p; p += m.dims[l + 1];
        }
    }
    *flat = p;
    return g;
}

// dW[o * ldw + i] += sum_{r < nrows} g[r][o] * a[r][i]   (o < OUT, i < IN; dW in torch (out, in) layout, possibly a column
// block of a wider matrix: ldw = its full row length);  db[o] += sum_r g[r][o]  (db may be null).
// Lanes run over consecutive i of one output row, so the atomics of a wave hit consecutive addresses.
__device__ __forceinline__ void wgrad_lds(const float* g, int g_ld, int OUT, const float* a, int a_ld, int IN, float* dW,
                                          int ldw, float* db, int nrows, int tid, int nthreads) {
    for (int item = tid; item < OUT * IN; item += nthreads) {
        const int o = item / IN, i = item - o * IN;
        float s = 0.f;
        for (int r = 0; r < nrows; ++r) s = fmaf(g[r * g_ld + o], a[r * a_ld + i], s);
        if (s != 0.f) unsafeAtomicAdd(&dW[(size_t)o * ldw + i], s);
    }
    if (db) {
        for (int o = tid; o < OUT; o += nthreads) {
            float s = 0.f;
            for (int r = 0; r < nrows; ++r) s += g[r * g_ld + o];
            if (s != 0.f) unsafeAtomicAdd(&db[o], s);
        }
    }
}

// Backward through the same MLP, given the `pre` buffers of a forward pass.
//   dout : LDS [RB][dout_ld] gradient w.r.t. the MLP output
//   ga, gb : LDS [RB][HLD] scratch
//   din  : LDS [RB][din_ld] gradient w.r.t. the layer-0 input; if skip_first, the gradient w.r.t. layer 0's
//          linear OUTPUT (pre[0]) is left in `ga` instead and din is untouched.
// Weight gradients (training path): pass `grads` (+ `act`: LDS [RB][HLD] scratch for the re-derived layer inputs, `in`:
// the layer-0 input rows, and `nrows` = valid rows of the block); they are accumulated with atomics.  With skip_first the
// layer-0 weight gradient is the caller's business (factorised edge layer).
// WG is a compile-time switch: the latent-optimisation kernels instantiate WG = false and carry none of the weight-gradient
// code (it costs registers and time in these latency-bound kernels even when it is branched over).
template <int RB, bool WG = false>
__device__ __forceinline__ void mlp_backward_lds(const MLPDev& m, const float* pre, const float* dout, int dout_ld,
                                                 float* ga, float* gb, float* din, int din_ld, bool skip_first, int tid,
                                                 int nthreads, const MLPGradDev* grads = nullptr, float* act = nullptr,
                                                 const float* in = nullptr, int in_ld = 0, int nrows = RB) {
    const int L = m.nlayers;
    const float* g = dout;
    int g_ld = dout_ld;
    const bool wg = WG && grads && grads->w[L - 1];
    for (int l = L - 1; l >= 1; --l) {
        const float* p = pre + (size_t)(l - 1) * RB * HLD;
        if (wg) {
            // this layer's input = relu(layer_norm(pre[l-1])), re-derived into `act`
            ln_relu_rows<RB>(p, HLD, act, HLD, m.dims[l], m.ln_g[l - 1], m.ln_b[l - 1], tid, nthreads);
            __syncthreads();
            wgrad_lds(g, g_ld, m.dims[l + 1], act, HLD, m.dims[l], grads->w[l], m.dims[l], grads->b[l], nrows, tid, nthreads);
        }
        // gradient w.r.t. the post-ReLU activation feeding layer l: gb = g * W_l   (W_l torch layout (out,in))
        dense_lds<RB, false>(g, g_ld, m.dims[l + 1], m.w[l], m.dims[l], nullptr, gb, HLD, m.dims[l], tid, nthreads);
        __syncthreads();
        ln_relu_bwd_rows<RB>(p, HLD, gb, HLD, ga, HLD, m.dims[l], m.ln_g[l - 1], m.ln_b[l - 1], tid, nthreads,
                             wg ? grads->ln_g[l - 1] : nullptr, wg ? grads->ln_b[l - 1] : nullptr, nrows);
        __syncthreads();
        g = ga;
        g_ld = HLD;
    }
    if (!skip_first) {
        if (wg) wgrad_lds(g, g_ld, m.dims[1], in, in_ld, m.dims[0], grads->w[0], m.dims[0], grads->b[0], nrows, tid, nthreads);
        if (din) {
            dense_lds<RB, false>(g, g_ld, m.dims[1], m.w[0], m.dims[0], nullptr, din, din_ld, m.dims[0], tid, nthreads);
            __syncthreads();
        }
    }
}
