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

// ---------------------------------------------------------------------------------------------------------------------
// The same product on the matrix cores, fp32-accurate: out[r][c] (+)= bias[c] + sum_k in[r][k] * W[c][k].
//
// Measured (tools/dense_probe.hip, 10 chained 128x128 layers on 4 rows, 128 workgroups): 2.5 us per layer with dense_lds,
// 0.97 us here.  The VALU form is bound by its weight stream -- 64 dependent-latency 4-byte loads per thread and layer --
// not by arithmetic; as matrix operands the same 64 KB arrive as 16 coalesced 16-byte loads per lane, all in flight at
// once, and the multiply-adds cost 24 instructions per wave.
//
// Both operands are split into two fp16 pieces (x = x0 + x1 up to 2^-22 |x|, x0 = fp16(x) to nearest, x1 = fp16(x - x0))
// and three products are accumulated in fp32 by v_mfma_f32_16x16x32_f16 (x0 w0 + x0 w1 + x1 w0; x1 w1 <= 2^-22 is dropped).
// The weights are split and laid out in fragment order on the host (`frag`, scaled by the power of two `wscale`); every
// activation ROW is scaled by its own power of two (largest magnitude into [2^14, 2^15)) before the split, so rows of any
// magnitude -- loss gradients of 1e-9 next to 1e+3 in the backward kernels -- keep 22 significant bits relative to their own
// largest entry, which is what an fp32 dot product keeps.
//   frag: [n-tile nt = c / 16][k-step ks = k / 32][piece][lane][8 x fp16]; lane l holds channel 16 nt + (l & 15),
//         k = 32 ks + 8 (l >> 4) + j; rows / columns beyond (OUT, IN) are zero.
//   D of one instruction: lane l holds channels 16 nt + 4 (l >> 4) + {0..3} of activation row l & 15.
// Called by all threads; `out` must not alias `in`; the caller synchronises afterwards (as for dense_lds).
typedef __attribute__((ext_vector_type(8))) _Float16 mf_f16x8;
typedef float mf_f32x4 __attribute__((ext_vector_type(4)));

template <int RB>
struct MfmaScratch {
    static constexpr int MAXK = RB <= 4 ? 288 : 128;             // padded K the scratch is sized for
    static constexpr int BYTES = 2 * RB * (2 * MAXK + 16);
};
template <int RB>
__device__ __forceinline__ unsigned char* mfma_scratch() {
    __shared__ __attribute__((aligned(16))) unsigned char s_mf[MfmaScratch<RB>::BYTES + RB * 4];
    return s_mf;
}

__device__ __forceinline__ bool dense_mfma_ok(const void* frag, int RB, int IN, int OUT) {
    return frag != nullptr && IN >= 32 && OUT >= 32 && ((IN + 31) & ~31) <= (RB <= 4 ? 288 : 128);
}

template <int RB, bool ACCUM>
__device__ __forceinline__ void dense_mfma(const float* in, int in_ld, int IN, const uint4* __restrict__ frag, float wscale,
                                           const float* __restrict__ bias, float* out, int out_ld, int OUT, int tid,
                                           int nthreads) {
    static_assert(RB <= 16, "one 16-row tile");
    const int KS = (IN + 31) >> 5, KP = KS * 32, NTL = (OUT + 15) >> 4;
    const int BROW = 2 * KP + 16;                                 // bytes per row and piece (+16: rows land in distinct banks)
    unsigned char* sb = mfma_scratch<RB>();
    float* s_rs = reinterpret_cast<float*>(sb + MfmaScratch<RB>::BYTES);
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    // the weight fragments do not depend on the activations: the first batch (two 16-channel tiles x 4 k-steps = 16 loads per
    // lane -- the whole layer when it is 128 x 128 and the workgroup has 4 waves) is requested BEFORE the split, so the L2 round
    // trip runs under it
    auto load_a = [&](int nt0, int nt1, int ks0, uint4 (&a)[2][4][2]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ks = ks0 + q < KS ? ks0 + q : KS - 1;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                a[0][q][pl] = frag[((size_t)(nt0 * KS + ks) * 2 + pl) * 64 + lane];
                a[1][q][pl] = frag[((size_t)(nt1 * KS + ks) * 2 + pl) * 64 + lane];
            }
        }
    };
    uint4 a[2][4][2];
    {
        const int f0 = 2 * wave < NTL ? 2 * wave : 0;
        load_a(f0, f0 + 1 < NTL ? f0 + 1 : f0, 0, a);
    }
    // ---- 1. per-row power-of-two scale, split into the two pieces ----
    for (int r = wave; r < RB; r += nw) {
        float mx = 0.f;
        for (int k = lane; k < IN; k += 64) mx = fmaxf(mx, fabsf(in[r * in_ld + k]));
        mx = wave_max(mx);
        float sc = 1.f;
        if (mx > 0.f && mx < 3.0e38f) {
            int e = 127 + 14 - ((__float_as_int(mx) >> 23) & 255) + 127;       // biased exponent of 2^(14 - floor(log2 mx))
            e = e < 1 ? 1 : (e > 254 ? 254 : e);
            sc = __int_as_float(e << 23);
        }
        for (int k2 = 2 * lane; k2 < KP; k2 += 128) {
            const float v0 = k2 < IN ? in[r * in_ld + k2] * sc : 0.f;
            const float v1 = k2 + 1 < IN ? in[r * in_ld + k2 + 1] * sc : 0.f;
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            uint16_t a0, a1, b0, b1;
            __builtin_memcpy(&a0, &h0, 2);
            __builtin_memcpy(&a1, &h1, 2);
            __builtin_memcpy(&b0, &l0, 2);
            __builtin_memcpy(&b1, &l1, 2);
            *reinterpret_cast<uint32_t*>(sb + r * BROW + k2 * 2) = a0 | ((uint32_t)a1 << 16);
            *reinterpret_cast<uint32_t*>(sb + (RB + r) * BROW + k2 * 2) = b0 | ((uint32_t)b1 << 16);
        }
        if (lane == 0) s_rs[r] = 1.0f / (sc * wscale);            // powers of two: exact
    }
    __syncthreads();
    // ---- 2. two 16-channel tiles per wave at a time ----
    const int row = lane & 15, g = lane >> 4;
    const int row_eff = row < RB ? row : 0;                       // the D columns of the rows that do not exist are never stored
    const unsigned char* bp0 = sb + row_eff * BROW + g * 16;
    const float rs = s_rs[row_eff];
    bool first = true;
    for (int nt0 = 2 * wave; nt0 < NTL; nt0 += 2 * nw) {
        const bool two = nt0 + 1 < NTL;
        const int nt1 = two ? nt0 + 1 : nt0;
        mf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int ks0 = 0; ks0 < KS; ks0 += 4) {
            if (!first) load_a(nt0, nt1, ks0, a);
            first = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ks0 + q < KS) {
                    const unsigned char* bp = bp0 + (ks0 + q) * 64;
                    const mf_f16x8 b0 = *reinterpret_cast<const mf_f16x8*>(bp);
                    const mf_f16x8 b1 = *reinterpret_cast<const mf_f16x8*>(bp + RB * BROW);
                    mf_f16x8 w0, w1;
                    __builtin_memcpy(&w0, &a[0][q][0], 16);
                    __builtin_memcpy(&w1, &a[0][q][1], 16);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, b0, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b1, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b0, acc0, 0, 0, 0);
                    __builtin_memcpy(&w0, &a[1][q][0], 16);
                    __builtin_memcpy(&w1, &a[1][q][1], 16);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, b0, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b1, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b0, acc1, 0, 0, 0);
                }
            }
        }
        if (row < RB) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
                const mf_f32x4 acc = t ? acc1 : acc0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * (t ? nt1 : nt0) + 4 * g + r;
                    if (c < OUT) {
                        float v = fmaf(acc[r], rs, bias ? bias[c] : 0.f);
                        if (ACCUM) v += out[row * out_ld + c];
                        out[row * out_ld + c] = v;
                    }
                }
            }
        }
    }
}

// dense_mfma when the layer has fragments and fits, dense_lds (fp32 VALU on Wt) otherwise
template <int RB, bool ACCUM>
__device__ __forceinline__ void dense_any(const float* in, int in_ld, int IN, const float* __restrict__ Wt, int ldw,
                                          const uint4* __restrict__ frag, float wscale, const float* __restrict__ bias,
                                          float* out, int out_ld, int OUT, int tid, int nthreads) {
    if (dense_mfma_ok(frag, RB, IN, OUT)) dense_mfma<RB, ACCUM>(in, in_ld, IN, frag, wscale, bias, out, out_ld, OUT, tid, nthreads);
    else dense_lds<RB, ACCUM>(in, in_ld, IN, Wt, ldw, bias, out, out_ld, OUT, tid, nthreads);
}

// y = relu(layer_norm(x)) row-wise over N channels (one wave per row, two-pass mean/variance).
// N <= 128 (the hidden width of every MLP on the path): a lane keeps its (up to) two channels, the LayerNorm parameters and,
// in the backward, the incoming gradient in registers, so a row costs ONE pass over LDS instead of three (five in the
// backward); the arithmetic and its order are those of the plain loops over c = lane, lane + 64.
template <int RB>
__device__ __forceinline__ void ln_relu_rows(const float* x, int x_ld, float* y, int y_ld, int N,
                                             const float* __restrict__ g, const float* __restrict__ b, int tid,
                                             int nthreads) {
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;
    const int c0 = lane, c1 = lane + 64;
    const bool h0 = c0 < N, h1 = c1 < N;
    const float g0 = h0 ? g[c0] : 0.f, g1 = h1 ? g[c1] : 0.f, b0 = h0 ? b[c0] : 0.f, b1 = h1 ? b[c1] : 0.f;
    for (int r = wave; r < RB; r += nw) {
        const float x0 = h0 ? x[r * x_ld + c0] : 0.f, x1 = h1 ? x[r * x_ld + c1] : 0.f;
        float s = 0.f;
        if (h0) s += x0;
        if (h1) s += x1;
        const float mean = wave_sum(s) / (float)N;
        float v = 0.f;
        if (h0) { const float d = x0 - mean; v = fmaf(d, d, v); }
        if (h1) { const float d = x1 - mean; v = fmaf(d, d, v); }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)N + LN_EPS);
        if (h0) y[r * y_ld + c0] = fmaxf((x0 - mean) * rstd * g0 + b0, 0.f);
        if (h1) y[r * y_ld + c1] = fmaxf((x1 - mean) * rstd * g1 + b1, 0.f);
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
    const int cc[2] = {lane, lane + 64};
    const bool hh[2] = {cc[0] < N, cc[1] < N};
    float gq[2], bq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { gq[q] = hh[q] ? g[cc[q]] : 0.f; bq[q] = hh[q] ? b[cc[q]] : 0.f; }
    for (int r = wave; r < RB; r += nw) {
        float xq[2], dq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            xq[q] = hh[q] ? x[r * x_ld + cc[q]] : 0.f;
            dq[q] = hh[q] ? dy[r * dy_ld + cc[q]] : 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (hh[q]) s += xq[q];
        const float mean = wave_sum(s) / (float)N;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (hh[q]) { const float d = xq[q] - mean; v = fmaf(d, d, v); }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)N + LN_EPS);
        float m1 = 0.f, m2 = 0.f, xh[2] = {0.f, 0.f}, gg[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (hh[q]) {
                xh[q] = (xq[q] - mean) * rstd;
                const float pre = xh[q] * gq[q] + bq[q];
                const float dn = pre > 0.f ? dq[q] : 0.f;
                gg[q] = dn * gq[q];
                m1 += gg[q];
                m2 = fmaf(gg[q], xh[q], m2);
                if (dgam && r < nrows && dn != 0.f) {
                    unsafeAtomicAdd(&dgam[cc[q]], dn * xh[q]);
                    unsafeAtomicAdd(&dbet[cc[q]], dn);
                }
            }
        }
        m1 = wave_sum(m1) / (float)N;
        m2 = wave_sum(m2) / (float)N;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (hh[q]) dx[r * dx_ld + cc[q]] = rstd * (gg[q] - m1 - xh[q] * m2);
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
    const uint4* wf[STRIVE_MAX_LAYERS];      // matrix-core fragments of W_l (forward) and of W_l^T (input gradient), or null
    const uint4* wbf[STRIVE_MAX_LAYERS];
    float wsc[STRIVE_MAX_LAYERS];            // their power-of-two scale
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
        d.wf[i] = reinterpret_cast<const uint4*>(m.wf[i]);
        d.wbf[i] = reinterpret_cast<const uint4*>(m.wbf[i]);
        d.wsc[i] = m.wsc[i];
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
        dense_any<RB, false>(in, in_ld, m.dims[0], m.wt[0], m.dims[1], m.wf[0], m.wsc[0], m.b[0], (L == 1) ? out : pre,
                             (L == 1) ? out_ld : HLD, m.dims[1], tid, nthreads);
        __syncthreads();
    }
    for (int l = 1; l < L; ++l) {
        float* p = pre + (size_t)(l - 1) * RB * HLD;
        ln_relu_rows<RB>(p, HLD, act, HLD, m.dims[l], m.ln_g[l - 1], m.ln_b[l - 1], tid, nthreads);
        __syncthreads();
        const bool last = (l == L - 1);
        dense_any<RB, false>(act, HLD, m.dims[l], m.wt[l], m.dims[l + 1], m.wf[l], m.wsc[l], m.b[l],
                             last ? out : pre + (size_t)l * RB * HLD, last ? out_ld : HLD, m.dims[l + 1], tid, nthreads);
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
    struct WJobTable* jobs;       // deferred weight gradients (training rollout), or null: atomics
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
    g.jobs = nullptr;
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

// ---------------------------------------------------------------------------------------------------------------------
// Deferred weight gradients (training rollout).  A 128 x 128 weight block costs a workgroup 64 atomic wave instructions per
// layer whatever its row count, and at ~100 cycles each those were 2/3 of the reverse-sweep kernels' time (4 rows per
// workgroup).  With a job table the kernels instead APPEND their (dL/d pre-activation, layer input) rows to a per-weight-block
// tape -- two coalesced row copies -- and one product per block  dW += G^T A  over the rows of ALL steps follows the sweep
// (wjobs_gemm_kernel).  Jobs are keyed by the dW pointer the call site passes; a site without a job keeps the atomic path.
// ---------------------------------------------------------------------------------------------------------------------
#define STRIVE_WJOBS_MAX 40
struct WJobTable {
    int n, overflow;
    int count[STRIVE_WJOBS_MAX];          // rows appended so far
    int OUT[STRIVE_WJOBS_MAX], IN[STRIVE_WJOBS_MAX], ldw[STRIVE_WJOBS_MAX], cap[STRIVE_WJOBS_MAX];
    float* dW[STRIVE_WJOBS_MAX];
    float* db[STRIVE_WJOBS_MAX];
    float* G[STRIVE_WJOBS_MAX];           // (cap, OUT)
    float* A[STRIVE_WJOBS_MAX];           // (cap, IN)
    // wjobs_gemm_kernel's grid.z: job j owns slices z0[j] .. z0[j] + ks[j] - 1 of its rows (ks ~ rows / WJOBS_ROWS_PER_WG: the edge
    // jobs have 15 x the rows of the node jobs; with one split count for all, 16 workgroups walked 1440 rows each while the rest idled)
    int ks[STRIVE_WJOBS_MAX], z0[STRIVE_WJOBS_MAX], ztotal;
};

// dW[o * ldw + i] += sum_{r < nrows} g[r][o] * a[r][i]   (o < OUT, i < IN; dW in torch (out, in) layout, possibly a column
// block of a wider matrix: ldw = its full row length);  db[o] += sum_r g[r][o]  (db may be null).
// Lanes run over consecutive i of one output row, so the atomics of a wave hit consecutive addresses.
// Called by all threads of the workgroup (with `jobs` it synchronises).
__device__ __forceinline__ void wgrad_lds(const float* g, int g_ld, int OUT, const float* a, int a_ld, int IN, float* dW,
                                          int ldw, float* db, int nrows, int tid, int nthreads, WJobTable* jobs = nullptr) {
    if (jobs) {
        __shared__ int s_job[2];
        if (tid < 64) {
            // one table entry per lane (a serial scan by one thread was ~0.5 us per entry: 70 us of the GRU kernel's 85)
            const bool hit = tid < jobs->n && jobs->dW[tid] == dW;
            const unsigned long long m = __ballot(hit);
            if (tid == 0) {
                const int j = m ? __ffsll((long long)m) - 1 : -1;
                int row0 = 0;
                if (j >= 0 && nrows > 0) {
                    row0 = atomicAdd(&jobs->count[j], nrows);
                    if (row0 + nrows > jobs->cap[j]) { jobs->overflow = 1; row0 = -1; }      // cannot happen by construction; loud if it does
                }
                s_job[0] = j;
                s_job[1] = row0;
            }
        }
        __syncthreads();
        const int j = s_job[0], row0 = s_job[1];
        __syncthreads();
        if (j >= 0) {
            if (row0 >= 0) {
                float* G = jobs->G[j] + (size_t)row0 * OUT;
                float* A = jobs->A[j] + (size_t)row0 * IN;
                for (int i = tid; i < nrows * OUT; i += nthreads) { const int r = i / OUT, o = i - r * OUT; G[i] = g[r * g_ld + o]; }
                for (int i = tid; i < nrows * IN; i += nthreads) { const int r = i / IN, c = i - r * IN; A[i] = a[r * a_ld + c]; }
            }
            return;
        }
    }
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

// dW += G^T A, db += column sums of G for every job: 64 x 64 tile of dW per workgroup, the rows of job j split ks[j] ways.
// grid = (ceil(maxIN / 64), ceil(maxOUT / 64), ztotal)
#define WJOBS_ROWS_PER_WG 256
#define WJOBS_KSPLIT_MAX 96
static inline void wjobs_finish(WJobTable& t) {
    int z = 0;
    for (int j = 0; j < t.n; ++j) {
        int k = (t.cap[j] + WJOBS_ROWS_PER_WG - 1) / WJOBS_ROWS_PER_WG;
        k = k < 1 ? 1 : (k > WJOBS_KSPLIT_MAX ? WJOBS_KSPLIT_MAX : k);
        t.ks[j] = k;
        t.z0[j] = z;
        z += k;
    }
    t.ztotal = z;
}
static __global__ __launch_bounds__(256) void wjobs_gemm_kernel(const WJobTable* __restrict__ T) {
    __shared__ float Gs[16][68];
    __shared__ float As[16][68];
    int job = 0;
    while (job + 1 < T->n && (int)blockIdx.z >= T->z0[job + 1]) ++job;
    const int ks = (int)blockIdx.z - T->z0[job], WJOBS_KSPLIT = T->ks[job];
    if (job >= T->n || ks >= WJOBS_KSPLIT) return;
    const int OUT = T->OUT[job], IN = T->IN[job], ldw = T->ldw[job];
    const int o0 = blockIdx.y * 64, i0 = blockIdx.x * 64;
    if (o0 >= OUT || i0 >= IN) return;
    int rows = T->count[job];
    rows = rows < T->cap[job] ? rows : T->cap[job];
    const int per = ((rows + WJOBS_KSPLIT - 1) / WJOBS_KSPLIT + 15) / 16 * 16;
    const int k_begin = ks * per, k_end = (k_begin + per) < rows ? (k_begin + per) : rows;
    if (k_begin >= k_end) return;
    const float* G = T->G[job];
    const float* A = T->A[job];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[4][4], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int mm = e & 63, kk = e >> 6;
            const int k = k0 + kk;
            Gs[kk][mm] = (k < k_end && o0 + mm < OUT) ? G[(size_t)k * OUT + o0 + mm] : 0.f;
            As[kk][mm] = (k < k_end && i0 + mm < IN) ? A[(size_t)k * IN + i0 + mm] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float gv[4], av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gv[i] = Gs[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = As[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bsum[i] += gv[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(gv[i], av[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
    float* dW = T->dW[job];
    const bool bad = T->overflow != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = o0 + ty + 16 * i;
        if (o >= OUT) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = i0 + tx + 16 * j;
            if (c < IN) {
                const float v = bad ? __int_as_float(0x7fc00000) : acc[i][j];
                if (v != 0.f) unsafeAtomicAdd(&dW[(size_t)o * ldw + c], v);
            }
        }
        if (T->db[job] && blockIdx.x == 0 && tx == 0 && bsum[i] != 0.f) unsafeAtomicAdd(&T->db[job][o], bsum[i]);
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
            wgrad_lds(g, g_ld, m.dims[l + 1], act, HLD, m.dims[l], grads->w[l], m.dims[l], grads->b[l], nrows, tid, nthreads, grads->jobs);
        }
        // gradient w.r.t. the post-ReLU activation feeding layer l: gb = g * W_l   (W_l torch layout (out,in))
        dense_any<RB, false>(g, g_ld, m.dims[l + 1], m.w[l], m.dims[l], m.wbf[l], m.wsc[l], nullptr, gb, HLD, m.dims[l], tid,
                             nthreads);
        __syncthreads();
        ln_relu_bwd_rows<RB>(p, HLD, gb, HLD, ga, HLD, m.dims[l], m.ln_g[l - 1], m.ln_b[l - 1], tid, nthreads,
                             wg ? grads->ln_g[l - 1] : nullptr, wg ? grads->ln_b[l - 1] : nullptr, nrows);
        __syncthreads();
        g = ga;
        g_ld = HLD;
    }
    if (!skip_first) {
        if (wg) wgrad_lds(g, g_ld, m.dims[1], in, in_ld, m.dims[0], grads->w[0], m.dims[0], grads->b[0], nrows, tid, nthreads, grads->jobs);
        if (din) {
            dense_any<RB, false>(g, g_ld, m.dims[1], m.w[0], m.dims[0], m.wbf[0], m.wsc[0], nullptr, din, din_ld, m.dims[0], tid,
                                 nthreads);
            __syncthreads();
        }
    }
}
