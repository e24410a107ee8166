// Scene-resident decoder kernels: ONE workgroup per scene runs a whole decoder step (forward) or the whole reverse-time
// sweep (backward) with the scene's node and edge rows resident in LDS.
// (reference src/models/traffic_model.py:589-704, src/models/interaction_net.py:52-218; gradient map: SURVEY.md Appendix A)
//
// The per-kernel chain of rollout.hip (node1 -> edge -> node2 forward, gru_bwd -> node2_bwd -> edge_bwd -> node1_bwd in the
// sweep) costs 7 dependent launches per step whose duration is the same for 8 agents as for 512 (profiles/r03_1x8_kernel_stats.txt):
// every workgroup pushes 4 rows through a layer, so a 16-row matrix instruction runs three quarters empty, every layer streams
// its weights once per 4 rows, and every phase boundary is a kernel boundary.  Here a scene's n <= 16 node rows are ONE
// 16-row tile of v_mfma_f32_16x16x32_f16, its n (n - 1) edge rows are processed in chunks of 64 rows (4 tiles), phases are
// separated by workgroup barriers instead of launches, and the backward loops over all FT steps inside one launch with the
// adjoint state (g_state, g_pos, d_loc, g_pf, g_mem, dz) in LDS.  Arithmetic: the fp16 x 3 scheme of mlp_dev.h (every activation
// row scaled by its own power of two, two fp16 pieces per operand, three products, fp32 accumulate); LayerNorm and the
// splits run with 16 lanes per row (four rows per wave instruction, DPP reductions inside a row of lanes).
//
// The tape layout is the one of rollout.hip, so either direction can be paired with the launch-per-phase kernels
// (option scene_kernels = 0 keeps those; the training sweep, multi-sample rollouts and scenes of more than 16 agents use them).
#pragma once

namespace scn {

constexpr int NTHR = 512;          // 8 waves: two per SIMD
constexpr int NWAVE = NTHR / 64;
constexpr int NR = 16;             // node rows of a scene (one matrix tile)
constexpr int EC = 64;             // edge rows per chunk (four tiles)
constexpr int XLD = 68;            // leading dimension of 64-wide LDS rows
constexpr int GLDS = 196;          // ... of 192-wide rows (GRU gate pre-activations, mlp_in features)

// Compiler-only fence.  The sweep kernel is one long loop over steps and edge chunks; without it the loop-invariant parameter
// loads of EVERY phase (LayerNorm affine terms, W_rel, the last layer's rows: ~200 registers) are hoisted to the kernel's
// entry and spilled (256 registers + 420 bytes of scratch per lane before, see profiles/r04_scene_kernels_resources.txt).
// The thread index is passed through it as an opaque value, so the per-lane addresses a phase derives from it (one per weight
// table, split buffer and tape section: ~100 registers) are formed where they are used instead of once per kernel.
// SCN_SYNC = workgroup barrier + that fence: every barrier-delimited stage of the sweep keeps its own temporaries only.
#define SCN_SYNC(tid)     \
    do {                  \
        __syncthreads();  \
        SCN_PHASE(tid);   \
    } while (0)
#if defined(__AMDGCN__)
#define SCN_PHASE(tid) asm volatile("" : "+v"(tid) : : "memory")
#else
#define SCN_PHASE(tid) asm volatile("" : "+r"(tid) : : "memory")
#endif

struct GRUFrag {
    const uint4* whh_f[3];
    const uint4* wih_f[3];
    const uint4* whh_bf[3];
    const uint4* wih_bf[3];
    float hh_sc[3], ih_sc[3];
};

static inline GRUFrag gru_frag(const StriveGRU& g) {
    GRUFrag f;
    for (int l = 0; l < 3; ++l) {
        f.whh_f[l] = reinterpret_cast<const uint4*>(g.whh_f[l]);
        f.wih_f[l] = reinterpret_cast<const uint4*>(g.wih_f[l]);
        f.whh_bf[l] = reinterpret_cast<const uint4*>(g.whh_bf[l]);
        f.wih_bf[l] = reinterpret_cast<const uint4*>(g.wih_bf[l]);
        f.hh_sc[l] = g.hh_sc[l];
        f.ih_sc[l] = g.ih_sc[l];
    }
    return f;
}

// Are the matrix-core operands this path needs all there (weights packed with fragments, GRU included)?
// (`any_size`: the node-level phases of the forward step run in 16-row tiles of a scene of any size -- scene_fwd_step_kernel modes
//  1 and 4 on a (B, 1, tiles) grid; everything else needs the whole scene in one tile)
static inline bool supported(const StriveDecoder& d, const StriveScenes& sc, bool any_size = false) {
    if (sc.NS != 1 || (sc.max_n > NR && !any_size) || sc.max_n < 1 || !d.scene_par) return false;
    if (d.gnn.D != 64 || d.gnn.NC > 8) return false;
    // k-step counts the kernels are written for: mlp_in 162 + NC -> 6, edge layer 0 (132 + 2 NC: sem_i, sem_j and the relative
    // pose share the 5th step), update 128 + NC -> 5
    if (((d.gnn.mlp_in.dims[0] + 31) >> 5) != 6 || ((d.gnn.edge.dims[0] + 31) >> 5) != 5 || ((d.gnn.update.dims[0] + 31) >> 5) != 5) return false;
    const StriveMLP* ms[4] = {&d.gnn.mlp_in, &d.gnn.edge, &d.gnn.update, &d.gnn.mlp_out};
    for (int i = 0; i < 4; ++i)
        for (int l = 0; l < ms[i]->nlayers; ++l) {
            const bool small = ms[i]->dims[l + 1] < 32 || ms[i]->dims[l] < 32;
            if (!small && (!ms[i]->wf[l] || !ms[i]->wbf[l])) return false;
        }
    for (int l = 0; l < 3; ++l) {
        if (!d.gru.whh_f[l] || !d.gru.whh_bf[l]) return false;
        if (l > 0 && (!d.gru.wih_f[l] || !d.gru.wih_bf[l])) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-lane row groups: lane = 16 * sub + q; a wave instruction works on 4 rows, lane q of a group on 8 consecutive channels
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float grp_sum(float v) {
    v += dpp_move<STRIVE_DPP_QUAD_XOR1>(v);
    v += dpp_move<STRIVE_DPP_QUAD_XOR2>(v);
    v += dpp_move<STRIVE_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move<STRIVE_DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float grp_max(float v) {
    v = fmaxf(v, dpp_move<STRIVE_DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_move<STRIVE_DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_move<STRIVE_DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_move<STRIVE_DPP_ROW_MIRROR>(v));
    return v;
}

// power of two that brings mx into [2^14, 2^15) (mlp_dev.h dense_mfma)
__device__ __forceinline__ float pow2_scale(float mx) {
    float sc = 1.f;
    if (mx > 0.f && mx < 3.0e38f) {
        int e = 127 + 14 - ((__float_as_int(mx) >> 23) & 255) + 127;
        e = e < 1 ? 1 : (e > 254 ? 254 : e);
        sc = __int_as_float(e << 23);
    }
    return sc;
}

// Split buffer: two fp16 pieces of up to `cap` rows of KP (multiple of 32) values, row pitch BROW = 2 KP + 16 bytes (rows
// land in distinct banks), and the reciprocal of every row's scale.
struct SB {
    unsigned char* base;
    float* rs;
    int BROW, pstride;
    __device__ __forceinline__ SB() {}
    __device__ __forceinline__ SB(unsigned char* b, float* r, int KP, int cap) : base(b), rs(r), BROW(2 * KP + 16), pstride(cap * (2 * KP + 16)) {}
    static constexpr int bytes(int KP, int cap) { return 2 * cap * (2 * KP + 16); }
};

__device__ __forceinline__ void store_pieces8(const SB& sb, int r, int k, const float (&v)[8]) {
    mf_f16x8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = (_Float16)v[i];
        l[i] = (_Float16)(v[i] - (float)h[i]);
    }
    unsigned char* p = sb.base + (size_t)r * sb.BROW + k * 2;
    *reinterpret_cast<mf_f16x8*>(p) = h;
    *reinterpret_cast<mf_f16x8*>(p + sb.pstride) = l;
}

// rows [0, nrows) of a row source -> the two pieces; rows [nrows, rows_pad) and columns [IN, KP) are zero.
// src(r, k) returns element k of row r (k < IN).  KP <= 256.
template <typename Src>
__device__ __forceinline__ void split16(Src src, int IN, int KP, int nrows, int rows_pad, const SB& sb, int tid) {
    const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
    for (int r0 = 4 * wave; r0 < rows_pad; r0 += 4 * NWAVE) {
        const int r = r0 + sub;
        const bool live = r < nrows;
        float v[2][8];
        float mx = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int k0 = 8 * q + 128 * m;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = 0.f;
                if (live && k0 + i < IN) x = src(r, k0 + i);
                v[m][i] = x;
                mx = fmaxf(mx, fabsf(x));
            }
        }
        mx = grp_max(mx);
        const float sc = pow2_scale(mx);
        if (r < rows_pad) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int k0 = 8 * q + 128 * m;
                if (k0 < KP) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[m][i] *= sc;
                    store_pieces8(sb, r, k0, v[m]);
                }
            }
            if (q == 0) sb.rs[r] = 1.0f / sc;
        }
    }
}

// y = relu(layer_norm(x)) over 128 channels (eps 1e-5, affine) for rows [0, nrows), split into the two pieces; xrow(r) points at
// the row's 128 pre-LayerNorm values (LDS or global).  `f32` (optional): also write y as fp32 rows of pitch f32_ld.
template <typename RowPtr>
__device__ __forceinline__ void ln_relu_split16(RowPtr xrow, int nrows, int rows_pad, const float* __restrict__ gam,
                                                const float* __restrict__ bet, const SB& sb, int tid, float* f32 = nullptr,
                                                int f32_ld = 0) {
    const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
    float g8[8], b8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { g8[i] = gam[8 * q + i]; b8[i] = bet[8 * q + i]; }
    for (int r0 = 4 * wave; r0 < rows_pad; r0 += 4 * NWAVE) {
        const int r = r0 + sub;
        const bool live = r < nrows;
        float x[8];
        if (live) {
            const float4* p = reinterpret_cast<const float4*>(xrow(r) + 8 * q);
            const float4 a = p[0], b = p[1];
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        const float mean = grp_sum(s) / 128.0f;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = x[i] - mean; v = fmaf(d, d, v); }
        const float rstd = 1.0f / sqrtf(grp_sum(v) / 128.0f + LN_EPS);
        float y[8], mx = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            y[i] = live ? fmaxf((x[i] - mean) * rstd * g8[i] + b8[i], 0.f) : 0.f;
            mx = fmaxf(mx, y[i]);
        }
        mx = grp_max(mx);
        const float sc = pow2_scale(mx);
        if (r < rows_pad) {
            if (f32 && live) {
                float4* o = reinterpret_cast<float4*>(f32 + (size_t)r * f32_ld + 8 * q);
                o[0] = make_float4(y[0], y[1], y[2], y[3]);
                o[1] = make_float4(y[4], y[5], y[6], y[7]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] *= sc;
            store_pieces8(sb, r, 8 * q, y);
            if (q == 0) sb.rs[r] = 1.0f / sc;
        }
    }
}

// Pre-activation rows of a LayerNorm backward pass requested ahead of time (they come from the tape in HBM / L2; the request is
// made before the product that precedes the pass, so the round trip runs under it).  MAXIT = passes of a wave: 1 for the 16
// node rows, 2 for a chunk of 64 edge rows.
template <int MAXIT>
struct XR {
    float4 a[MAXIT], b[MAXIT];
};
template <int MAXIT, typename RowPtr>
__device__ __forceinline__ void xr_load(XR<MAXIT>& xr, RowPtr xrow, int nrows, int tid) {
    const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int r = 4 * wave + it * 4 * NWAVE + sub;
        if (r < nrows) {
            const float4* p = reinterpret_cast<const float4*>(xrow(r) + 8 * q);
            xr.a[it] = p[0];
            xr.b[it] = p[1];
        } else {
            xr.a[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            xr.b[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// Backward of y = relu(layer_norm(x)) over 128 channels: dx from dy (mlp_dev.h ln_relu_bwd_rows), for rows [0, nrows),
// rows_pad <= 32 MAXIT.  xr: the rows' pre-LayerNorm values (xr_load); dy(r, k0, d): fills d[0..7] with dL/dy of channels
// k0 .. k0+7.  Outputs: the two pieces of dx in `sb` (when sb != null) and / or fp32 rows (when out != null).
template <int MAXIT, typename DyFn>
__device__ __forceinline__ void ln_relu_bwd16(const XR<MAXIT>& xr, DyFn dy, int nrows, int rows_pad, const float* __restrict__ gam,
                                              const float* __restrict__ bet, const SB* sb, float* out, int out_ld, int tid) {
    const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
    float g8[8], b8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { g8[i] = gam[8 * q + i]; b8[i] = bet[8 * q + i]; }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int r0 = 4 * wave + it * 4 * NWAVE;
        if (r0 >= rows_pad) break;
        const int r = r0 + sub;
        const bool live = r < nrows;
        const float x[8] = {xr.a[it].x, xr.a[it].y, xr.a[it].z, xr.a[it].w, xr.b[it].x, xr.b[it].y, xr.b[it].z, xr.b[it].w};
        float d[8];
        if (live) dy(r, 8 * q, d);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        const float mean = grp_sum(s) / 128.0f;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float dd = x[i] - mean; v = fmaf(dd, dd, v); }
        const float rstd = 1.0f / sqrtf(grp_sum(v) / 128.0f + LN_EPS);
        float xh[8], gg[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xh[i] = (x[i] - mean) * rstd;
            const float pre = xh[i] * g8[i] + b8[i];
            const float dn = pre > 0.f ? d[i] : 0.f;
            gg[i] = dn * g8[i];
            m1 += gg[i];
            m2 = fmaf(gg[i], xh[i], m2);
        }
        m1 = grp_sum(m1) / 128.0f;
        m2 = grp_sum(m2) / 128.0f;
        float dx[8], mx = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dx[i] = live ? rstd * (gg[i] - m1 - xh[i] * m2) : 0.f;
            mx = fmaxf(mx, fabsf(dx[i]));
        }
        mx = grp_max(mx);
        if (out && live) {
            float4* o = reinterpret_cast<float4*>(out + (size_t)r * out_ld + 8 * q);
            o[0] = make_float4(dx[0], dx[1], dx[2], dx[3]);
            o[1] = make_float4(dx[4], dx[5], dx[6], dx[7]);
        }
        if (sb) {
            const float sc = pow2_scale(mx);
#pragma unroll
            for (int i = 0; i < 8; ++i) dx[i] *= sc;
            store_pieces8(*sb, r, 8 * q, dx);
            if (q == 0) sb->rs[r] = 1.0f / sc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The matrix product over 16 x 16 (channel, row) tiles: D[c][r] = sum over the listed k-steps of W[c][k] X[r][k].
//   frag : fragment table of W * wscale ([channel tile][k-step of KSW][piece][lane]), mlp_dev.h dense_mfma
//   wks / xks : for step i < NK the k-step of the weight table and of the split buffer (they differ when the activation
//               matrix holds only some column blocks of the layer's input: the factorised edge layer 0)
//   tiles T = nt + NTL * rt, nt < NTL channel tiles (nt0 = first channel tile of the table to use), rt < nrt row tiles;
//   wave w takes T = w, w + 8, ... and reloads its weight fragments only when the channel tile changes
//   epi(row, c0, v): v[0..3] = products of channels c0 .. c0+3 (c0 relative to nt0 * 16 ... see below) of activation row `row`,
//                    already divided by the row and weight scales
// ---------------------------------------------------------------------------------------------------------------------
// Weight fragments of one channel tile held in registers.  A product's first tile is known before its activations are: the
// callers request it (af_first) BEFORE the LayerNorm / split pass and the barrier that produce the activations, so that the L2
// round trip (~1-2 us, as long as the pass itself) runs under them; the edge loops keep their tile across all chunks.
template <int NK>
struct AF {
    uint4 a[NK][2];
    int nt;           // channel tile of the table the registers hold, -1 = none
};

template <int NK>
__device__ __forceinline__ void af_load(AF<NK>& f, const uint4* __restrict__ frag, int KSW, int nt_abs, const int (&wks)[NK], int lane) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        f.a[i][0] = frag[((size_t)(nt_abs * KSW + wks[i]) * 2 + 0) * 64 + lane];
        f.a[i][1] = frag[((size_t)(nt_abs * KSW + wks[i]) * 2 + 1) * 64 + lane];
    }
    f.nt = nt_abs;
}

// the first tile wave `tid >> 6` will take in mma_tiles(frag, KSW, ., nt0, NTL, nrt, wks, ...)
template <int NK>
__device__ __forceinline__ void af_first(AF<NK>& f, const uint4* __restrict__ frag, int KSW, int nt0, int NTL, int nrt, const int (&wks)[NK],
                                         int tid) {
    // (unconditional: a wave without a tile requests one it never uses -- no control flow around the register definitions)
    (void)nrt;
    af_load(f, frag, KSW, nt0 + (tid >> 6) % NTL, wks, tid & 63);
}
template <int NK>
__device__ __forceinline__ void af_first(AF<NK>& f, const uint4* __restrict__ frag, int nt0, int NTL, int nrt, int tid) {
    int ks[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) ks[i] = i;
    af_first<NK>(f, frag, NK, nt0, NTL, nrt, ks, tid);
}

// `pre` (optional): a tile requested ahead of time; it is only READ here -- a tile the loop has to fetch itself lives in the
// loop's own registers (handing the caller's registers back as the loop's cache kept 32-48 registers per product alive from
// product to product: 256 registers and 1 KB of scratch per lane in the sweep kernel).
template <int NK, typename Epi>
__device__ __forceinline__ void mma_tiles(const uint4* __restrict__ frag, int KSW, float wscale, int nt0, int NTL, int nrt,
                                          const int (&wks)[NK], const int (&xks)[NK], const SB& sb, int tid, Epi epi,
                                          const AF<NK>* pre = nullptr) {
    const int lane = tid & 63, wave = tid >> 6, row16 = lane & 15, g = lane >> 4;
    const float inv_w = 1.0f / wscale;
    const int ntiles = NTL * nrt;
    AF<NK> own;
    own.nt = -1;
    for (int T = wave; T < ntiles; T += NWAVE) {
        const int nt = T % NTL, rt = T / NTL;
        const int row = 16 * rt + row16;
        const unsigned char* bp = sb.base + (size_t)row * sb.BROW + g * 16;
        // one accumulator per product term: three independent chains of NK matrix instructions instead of one of 3 NK
        mf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        auto tile = [&](const uint4 (&a)[NK][2]) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const mf_f16x8 b0 = *reinterpret_cast<const mf_f16x8*>(bp + xks[i] * 64);
                const mf_f16x8 b1 = *reinterpret_cast<const mf_f16x8*>(bp + sb.pstride + xks[i] * 64);
                mf_f16x8 w0, w1;
                __builtin_memcpy(&w0, &a[i][0], 16);
                __builtin_memcpy(&w1, &a[i][1], 16);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, b0, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b1, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b0, acc, 0, 0, 0);
            }
        };
        if (pre && pre->nt == nt0 + nt) tile(pre->a);
        else {
            if (own.nt != nt0 + nt) af_load(own, frag, KSW, nt0 + nt, wks, lane);
            tile(own.a);
        }
        const float rs = sb.rs[row] * inv_w;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((acc1[r] + acc2[r]) + acc[r]) * rs;
        epi(row, 16 * (nt0 + nt) + 4 * g, v);
    }
}

// identity k-step lists
template <int NK> struct KS { int v[NK]; };
#define SCN_KLIST(name, NK) int name[NK]; _Pragma("unroll") for (int i_ = 0; i_ < NK; ++i_) name[i_] = i_

// A whole dense layer out[r][c] = bias[c] + sum_k X[r][k] W[c][k] over row tiles [0, nrt): NK = k-steps of the layer.
template <int NK>
__device__ __forceinline__ void dense_rows(const uint4* __restrict__ frag, float wscale, const float* __restrict__ bias, int OUT,
                                           int nrt, int nrows, const SB& sb, float* out, int out_ld, int tid,
                                           const AF<NK>* pre = nullptr) {
    int ks[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) ks[i] = i;
    mma_tiles<NK>(frag, NK, wscale, 0, (OUT + 15) >> 4, nrt, ks, ks, sb, tid, [&](int row, int c0, const float (&v)[4]) {
        if (row < nrows) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c0 + r < OUT) out[(size_t)row * out_ld + c0 + r] = v[r] + (bias ? bias[c0 + r] : 0.f);
        }
    }, pre);
}

// ---------------------------------------------------------------------------------------------------------------------
// The small parameters (biases, LayerNorm affine terms, W_rel, the 2 x 128 last layer, the GRU's biases and 4-wide input
// matrix: 22 KB) are copied to LDS once per launch.  A stage that reads them from global memory starts with an L2 round trip
// (~1 us, as long as the stage's arithmetic: tools/scene_phase_probe.py before / after in profiles/r04_scene_phase_*.txt).
// ---------------------------------------------------------------------------------------------------------------------
struct Par {
    enum {
        IN_B0 = 0, IN_B1 = 128, IN_B2 = 256, IN_G0 = 320, IN_E0 = 448, IN_G1 = 576, IN_E1 = 704,
        E_B0 = 832, E_B1 = 960, E_B2 = 1088, E_G0 = 1152, E_E0 = 1280, E_G1 = 1408, E_E1 = 1536, E_WREL = 1664,
        U_B0 = 2176, U_B1 = 2304, U_G0 = 2368, U_E0 = 2496,
        O_B0 = 2624, O_B1 = 2752, O_B2 = 2880, O_G0 = 2884, O_E0 = 3012, O_G1 = 3140, O_E1 = 3268, O_W2 = 3396,
        R_BIH = 3652, R_BHH = 4228, R_WIH0T = 4804, R_WIH0 = 5572, FLOATS = 6340
    };
};

// The block is laid out on the host (StriveDecoder.scene_par, strive_amd/params.py): one coalesced copy, every load in flight.
__device__ __forceinline__ void par_stage(float* par, const float* __restrict__ src, int tid) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(par);
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + k * NTHR;
        if (i < Par::FLOATS / 4) v[k] = s4[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + k * NTHR;
        if (i < Par::FLOATS / 4) d4[i] = v[k];
    }
}
static_assert(Par::FLOATS % 4 == 0 && Par::FLOATS / 4 <= 4 * NTHR, "par_stage copies four float4 per thread");

// ---------------------------------------------------------------------------------------------------------------------
// LDS map (bytes).  Persistent regions first, then one scratch region that the phases carve differently.
// ---------------------------------------------------------------------------------------------------------------------
struct FwdLds {
    // persistent over the step
    float* x;        // [NR][XLD]      node embeddings x (64)
    float* P;        // [NR][HLD]      edge layer-0 partials
    float* Q;        // [NR][HLD]
    float* A;        // [NR][64]       running max of the messages
    int* ARG;        // [NR][64]
    float* pos;      // [NR][4]
    float* rs;       // [EC]           row scales of the split buffer
    float* rel;      // [EC][4]
    int* ei;         // [EC]           target / source local index of the chunk's edge rows
    int* ej;
    unsigned char* sb;   // split buffer: max(2 pieces x EC rows x 128, 2 x NR x 192)
    float* U;        // scratch
    float* par;      // [Par::FLOATS]
    static constexpr int SB_BYTES = SB::bytes(128, EC) > SB::bytes(192, NR) ? SB::bytes(128, EC) : SB::bytes(192, NR);
    static constexpr int U_FLOATS = NR * GLDS + 3 * NR * HLD > EC * HLD + EC * XLD ? NR * GLDS + 3 * NR * HLD : EC * HLD + EC * XLD;
    static constexpr size_t BYTES = (size_t)(NR * XLD + 2 * NR * HLD + 2 * NR * 64 + NR * 4 + EC + EC * 4 + 2 * EC) * 4 + SB_BYTES +
                                    (size_t)(U_FLOATS + Par::FLOATS) * 4 + 64;
    __device__ FwdLds(float* base) {
        x = base;
        P = x + NR * XLD;
        Q = P + NR * HLD;
        A = Q + NR * HLD;
        ARG = reinterpret_cast<int*>(A + NR * 64);
        pos = reinterpret_cast<float*>(ARG + NR * 64);
        rs = pos + NR * 4;
        rel = rs + EC;
        ei = reinterpret_cast<int*>(rel + EC * 4);
        ej = ei + EC;
        sb = reinterpret_cast<unsigned char*>(ej + EC);
        U = reinterpret_cast<float*>(sb + SB_BYTES);
        par = U + U_FLOATS;
    }
};

struct StepArgsS {
    int t, FT, NC, max_n;
    int mode;                  // bit 0: features .. edge partials (x, P, Q into the tape); bit 1: the edge chunks (running max);
                               // bit 2: update MLP .. GRU (x and the aggregate come from the tape when bits 0 / 1 are off).  7 = a whole
                               // step in one launch.  Scenes of >= 15 agents run 1 | edge kernel of gnn_kernels.h | 4: their 130-240 edge
                               // rows are 3-4 chunks on ONE CU here and one workgroup per target there
    const float* sem;          // (NA, NC)
    const float* lw;           // (NA, 2) normalised
    const float* z;            // (NA, 32)
    const float* ext;          // (B, FT, 4) or null
    const int32_t* ptr;        // (B + 1)
    const float* par;          // StriveDecoder.scene_par
    float* traj;               // (NA, FT, 4)
    // round 5, K workgroups per scene for the edge chunks (grid (B, K), mode 3 | mode 4 instead of 1 | gnn_edge_kernel | 4): every
    // workgroup repeats the node-level phases of mode 1 (workgroup 0 writes the tape), walks chunks k, k + K, ... and leaves ITS
    // running max / arg-max in `part_a / part_arg` (B, K, NR, 64); the mode-4 launch folds the K partial maxima in workgroup order
    // with the rule of the chunk loop (a later source only wins when strictly larger): the aggregate of the one-workgroup walk
    int KW;                    // 0 = off
    float* part_a;
    int* part_arg;
};

// =====================================================================================================================
// forward: one decoder step of one scene.   grid = B (x K, see StepArgsS::KW), block = 512
// =====================================================================================================================
template <bool PROF>
static __global__ __launch_bounds__(NTHR) void scene_fwd_step_kernel(GNNDev g, GRUDev gru, GRUFrag gf, DynParams dp, StepArgsS a,
                                                                     Tape tp, unsigned long long* prof) {
    HIP_DYNAMIC_SHARED(float, smem)
    FwdLds L(smem);
    const int tid = threadIdx.x, b = blockIdx.x, t = a.t, NC = a.NC, H = STRIVE_HID;
    // Rows of this workgroup: the whole scene (<= NR agents, grid (B, K)), or -- round 6, scenes of any size -- the 16-row node tile
    // blockIdx.z of it (grid (B, 1, tiles), modes 1 and 4 only: the node-level phases are row-wise, the edge rows between them run
    // on gnn_edge_kernel, one workgroup per target node).  lo / n below are the tile's.
    const int lo_s = a.ptr[b], n_s = a.ptr[b + 1] - lo_s;
    const int tile = (int)blockIdx.z;
    const int lo = lo_s + tile * NR;
    const int n = n_s - tile * NR < NR ? n_s - tile * NR : NR;
    if (n <= 0) return;
    if (n_s > NR && (a.mode & 2)) return;        // (the edge chunks need the whole scene in one tile: never launched like this)
    const int kw = (a.KW > 0 && (a.mode & 2)) ? (int)blockIdx.y : 0, KW = (a.KW > 0 && (a.mode & 2)) ? a.KW : 1;
    const bool tape_w = kw == 0;                 // the K workgroups of a scene hold the same node-level values: one of them stores
    long long tick = PROF ? clock64() : 0;
#define SCN_TICK(id)                                                        \
    do {                                                                    \
        if (PROF && threadIdx.x == 0 && blockIdx.x == 0) {                  \
            const long long now_ = clock64();                               \
            prof[id] += (unsigned long long)(now_ - tick);                  \
            tick = now_;                                                    \
        }                                                                   \
    } while (0)
    const int F = 64 + 64 + NC + STRIVE_ZDIM + 2;                 // decoder_net input (traffic_model.py:628-629)
    float* s_in = L.U;                                            // [NR][GLDS]
    float* s_pre = L.U + NR * GLDS;                               // [3][NR][HLD]
    // requested first: the weight tiles of the first product and of the edge loop (kept across all chunks), and the GRU's
    // hidden inputs of this step
    AF<6> f_in0;
    af_first<6>(f_in0, g.mlp_in.wf[0], 0, 8, 1, tid);
    AF<4> f_e1, f_e2;
    af_first<4>(f_e1, g.edge.wf[1], 0, 8, 4, tid);
    af_first<4>(f_e2, g.edge.wf[2], 0, 4, 4, tid);
    float h_in[3][2];
    if (t < a.FT - 1) {
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = tid + k * NTHR;
                h_in[l][k] = i < n * 64 ? tp.mem_t(t)[((size_t)(lo + (i >> 6)) * 3 + l) * 64 + (i & 63)] : 0.f;
            }
    }

    if (a.mode & 1) {
    // ---- features [past_feat_t | map_feat_t | sem | z | lw], poses ----
    for (int i = tid; i < n * F; i += NTHR) {
        const int rr = i / F, k = i - rr * F, r = lo + rr;
        float v;
        if (k < 64) v = tp.pf_t(t)[(size_t)r * 64 + k];
        else if (k < 128) v = tp.mf_t(t)[(size_t)r * 64 + (k - 64)];
        else if (k < 128 + NC) v = a.sem[(size_t)r * NC + (k - 128)];
        else if (k < 128 + NC + STRIVE_ZDIM) v = a.z[(size_t)r * STRIVE_ZDIM + (k - 128 - NC)];
        else v = a.lw[(size_t)r * 2 + (k - 128 - NC - STRIVE_ZDIM)];
        s_in[rr * GLDS + k] = v;
    }
    if (tid < n * 4) L.pos[tid] = tp.pos_t(t)[(size_t)lo * 4 + tid];
    for (int i = tid; i < NR * 64; i += NTHR) { L.A[i] = 0.f; L.ARG[i] = -1; }
    par_stage(L.par, a.par, tid);
    __syncthreads();

    // ---- mlp_in: F -> 128 -> 128 -> 64 ----
    {
        SB sb(L.sb, L.rs, 192, NR);
        split16([&](int r, int k) { return s_in[r * GLDS + k]; }, F, 192, n, NR, sb, tid);
        __syncthreads();
        dense_rows<6>(g.mlp_in.wf[0], g.mlp_in.wsc[0], (L.par + Par::IN_B0), H, 1, n, sb, s_pre, HLD, tid, &f_in0);
        __syncthreads();
    }
    {
        SB sb(L.sb, L.rs, 128, NR);
        AF<4> f1, f2;
        af_first<4>(f1, g.mlp_in.wf[1], 0, 8, 1, tid);
        ln_relu_split16([&](int r) { return s_pre + r * HLD; }, n, NR, (L.par + Par::IN_G0), (L.par + Par::IN_E0), sb, tid);
        __syncthreads();
        dense_rows<4>(g.mlp_in.wf[1], g.mlp_in.wsc[1], (L.par + Par::IN_B1), H, 1, n, sb, s_pre + NR * HLD, HLD, tid, &f1);
        af_first<4>(f2, g.mlp_in.wf[2], 0, 4, 1, tid);
        // keep the pre-activations of both hidden layers for the reverse sweep
        for (int i = tid; i < n * H; i += NTHR) {
            const int rr = i >> 7, c = i & 127;
            if (tape_w) tp.PRE_IN_t(t)[(size_t)(lo + rr) * 2 * H + c] = s_pre[rr * HLD + c];
        }
        __syncthreads();
        ln_relu_split16([&](int r) { return s_pre + NR * HLD + r * HLD; }, n, NR, (L.par + Par::IN_G1), (L.par + Par::IN_E1), sb, tid);
        for (int i = tid; i < n * H; i += NTHR) {
            const int rr = i >> 7, c = i & 127;
            if (tape_w) tp.PRE_IN_t(t)[(size_t)(lo + rr) * 2 * H + H + c] = s_pre[NR * HLD + rr * HLD + c];
        }
        __syncthreads();
        dense_rows<4>(g.mlp_in.wf[2], g.mlp_in.wsc[2], (L.par + Par::IN_B2), 64, 1, n, sb, L.x, XLD, tid, &f2);
        __syncthreads();
    }
    if (tape_w)
        for (int i = tid; i < n * 64; i += NTHR) tp.X_t(t)[(size_t)lo * 64 + i] = L.x[(i >> 6) * XLD + (i & 63)];
    SCN_TICK(0);

    // ---- edge layer 0, factorised: P_i = W[:, x_i | sem_i] [x_i, sem_i] + b,  Q_j = W[:, x_j | sem_j] [x_j, sem_j] ----
    // the activation matrix is [x (64) | sem at the columns it has inside the layer's 5th k-step as sem_i (32) | as sem_j (32)]
    {
        SB sb(L.sb, L.rs, 128, NR);
        const int wksP[3] = {0, 1, 4}, xksP[3] = {0, 1, 2};
        const int wksQ[3] = {2, 3, 4}, xksQ[3] = {0, 1, 3};
        AF<3> fP, fQ;
        af_first<3>(fP, g.edge.wf[0], 5, 0, 8, 1, wksP, tid);
        af_first<3>(fQ, g.edge.wf[0], 5, 0, 8, 1, wksQ, tid);
        split16([&](int r, int k) {
            if (k < 64) return L.x[r * XLD + k];
            const int kk = k - 64;                       // 0..31: the sem_i copy, 32..63: the sem_j copy
            if (kk < 32) return kk < NC ? a.sem[(size_t)(lo + r) * NC + kk] : 0.f;
            const int kj = kk - 32 - NC;
            return (kj >= 0 && kj < NC) ? a.sem[(size_t)(lo + r) * NC + kj] : 0.f;
        }, 128, 128, n, NR, sb, tid);
        __syncthreads();
        const float* b0 = (L.par + Par::E_B0);
        mma_tiles<3>(g.edge.wf[0], 5, g.edge.wsc[0], 0, 8, 1, wksP, xksP, sb, tid, [&](int row, int c0, const float (&v)[4]) {
            if (row < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) L.P[row * HLD + c0 + r] = v[r] + b0[c0 + r];
            }
        }, &fP);
        mma_tiles<3>(g.edge.wf[0], 5, g.edge.wsc[0], 0, 8, 1, wksQ, xksQ, sb, tid, [&](int row, int c0, const float (&v)[4]) {
            if (row < n) {
#pragma unroll
                for (int r = 0; r < 4; ++r) L.Q[row * HLD + c0 + r] = v[r];
            }
        }, &fQ);
        __syncthreads();
    }
    if (tape_w)
        for (int i = tid; i < n * H; i += NTHR) {
            const int rr = i >> 7, c = i & 127;
            tp.P_t(t)[(size_t)(lo + rr) * H + c] = L.P[rr * HLD + c];
            tp.Q_t(t)[(size_t)(lo + rr) * H + c] = L.Q[rr * HLD + c];
        }
    SCN_TICK(1);
    } else if (a.KW > 0) {
        // the node embeddings are in the tape; the aggregated messages arrive as the K workgroups' partial maxima: folded in workgroup
        // order (= ascending source order), a later one only wins when strictly larger -- the rule of the chunk loop below
        const int E_ = n * (n - 1), nchunk = (E_ + EC - 1) / EC, kb = nchunk < a.KW ? nchunk : a.KW;
        for (int i = tid; i < n * 64; i += NTHR) {
            L.x[(i >> 6) * XLD + (i & 63)] = tp.X_t(t)[(size_t)lo * 64 + i];
            float best = 0.f;
            int arg = -1;
            for (int k = 0; k < kb; ++k) {
                const size_t o = ((size_t)b * a.KW + k) * NR * 64 + i;
                const int pa = a.part_arg[o];
                const float pv = a.part_a[o];
                if (pa >= 0 && (arg < 0 || pv > best)) { best = pv; arg = pa; }
            }
            const float v = arg < 0 ? 0.f : best;            // isolated node: aggregate 0 (interaction_net.py:188)
            L.A[i] = v;
            tp.A_t(t)[(size_t)lo * 64 + i] = v;
            tp.ARG_t(t)[(size_t)lo * 64 + i] = arg;
        }
        par_stage(L.par, a.par, tid);
        __syncthreads();
    } else {
        // the node embeddings and the aggregated messages of this step are in the tape (written by the launches before this one)
        for (int i = tid; i < n * 64; i += NTHR) {
            L.x[(i >> 6) * XLD + (i & 63)] = tp.X_t(t)[(size_t)lo * 64 + i];
            L.A[i] = tp.A_t(t)[(size_t)lo * 64 + i];
        }
        par_stage(L.par, a.par, tid);
        __syncthreads();
    }
    if (!(a.mode & 6)) return;

    // ---- edges in chunks of EC rows: e = i (n - 1) + jj, source jl = jj + (jj >= i) ----
    if (a.mode & 2) {
        float* s_e = L.U;                  // [EC][HLD]
        float* s_m = L.U + EC * HLD;       // [EC][XLD]
        const int E = n * (n - 1);
        const float* Wrel = L.par + Par::E_WREL;
        const int c_own = tid & 127;
        const float wr0 = Wrel[c_own], wr1 = Wrel[H + c_own], wr2 = Wrel[2 * H + c_own], wr3 = Wrel[3 * H + c_own];
        SB sb(L.sb, L.rs, 128, EC);
        for (int e0 = kw * EC; e0 < E; e0 += KW * EC) {
            const int ne = (E - e0) < EC ? (E - e0) : EC;
            const int ne_pad = (ne + 15) & ~15, nrt = ne_pad >> 4;
            if (tid < EC) {
                int i = 0, jl = 0;
                float rel[4] = {0.f, 0.f, 0.f, 0.f};
                if (tid < ne) {
                    const int e = e0 + tid;
                    i = e / (n - 1);
                    const int jj = e - i * (n - 1);
                    jl = jj + (jj >= i ? 1 : 0);
                    rel_pose(L.pos + i * 4, L.pos + jl * 4, rel);
                    for (int d = 0; d < 4; ++d)
                        if (rel[d] != rel[d]) rel[d] = 0.f;                    // interaction_net.py:162
                }
                L.ei[tid] = i;
                L.ej[tid] = jl;
                for (int d = 0; d < 4; ++d) L.rel[tid * 4 + d] = rel[d];
            }
            __syncthreads();
            for (int jr = tid >> 7; jr < ne; jr += NTHR >> 7) {
                const int i = L.ei[jr], jl = L.ej[jr];
                float v = L.P[i * HLD + c_own] + L.Q[jl * HLD + c_own];
                v = fmaf(L.rel[jr * 4 + 0], wr0, v);
                v = fmaf(L.rel[jr * 4 + 1], wr1, v);
                v = fmaf(L.rel[jr * 4 + 2], wr2, v);
                v = fmaf(L.rel[jr * 4 + 3], wr3, v);
                s_e[jr * HLD + c_own] = v;
                tp.PRE_E_t(t)[((size_t)(lo + i) * a.max_n + jl) * 2 * H + c_own] = v;
            }
            __syncthreads();
            SCN_TICK(2);
            ln_relu_split16([&](int r) { return s_e + r * HLD; }, ne, ne_pad, (L.par + Par::E_G0), (L.par + Par::E_E0), sb, tid);
            __syncthreads();
            SCN_TICK(3);
            dense_rows<4>(g.edge.wf[1], g.edge.wsc[1], (L.par + Par::E_B1), H, nrt, ne, sb, s_e, HLD, tid, &f_e1);
            __syncthreads();
            SCN_TICK(4);
            for (int jr = tid >> 7; jr < ne; jr += NTHR >> 7)
                tp.PRE_E_t(t)[((size_t)(lo + L.ei[jr]) * a.max_n + L.ej[jr]) * 2 * H + H + c_own] = s_e[jr * HLD + c_own];
            ln_relu_split16([&](int r) { return s_e + r * HLD; }, ne, ne_pad, (L.par + Par::E_G1), (L.par + Par::E_E1), sb, tid);
            __syncthreads();
            SCN_TICK(5);
            dense_rows<4>(g.edge.wf[2], g.edge.wsc[2], (L.par + Par::E_B2), 64, nrt, ne, sb, s_m, XLD, tid, &f_e2);
            __syncthreads();
            SCN_TICK(6);
            // running max / arg-max per (target, channel), sources in ascending order: ties keep the first (scatter-max rule)
            {
                const int i_first = e0 / (n - 1), i_last = (e0 + ne - 1) / (n - 1);
                for (int idx = tid; idx < (i_last - i_first + 1) * 64; idx += NTHR) {
                    const int i = i_first + (idx >> 6), c = idx & 63;
                    int eb = i * (n - 1), ee = eb + (n - 1);
                    eb = eb < e0 ? e0 : eb;
                    ee = ee > e0 + ne ? e0 + ne : ee;
                    float best = L.A[i * 64 + c];
                    int arg = L.ARG[i * 64 + c];
                    for (int e = eb; e < ee; ++e) {
                        const float v = s_m[(e - e0) * XLD + c];
                        if (arg < 0 || v > best) { best = v; arg = lo + L.ej[e - e0]; }
                    }
                    L.A[i * 64 + c] = best;
                    L.ARG[i * 64 + c] = arg;
                }
            }
            __syncthreads();
            SCN_TICK(7);
        }
    if (a.KW > 0) {
        // this workgroup's running maxima over ITS chunks; the mode-4 launch folds them
        for (int i = tid; i < n * 64; i += NTHR) {
            const size_t o = ((size_t)b * a.KW + kw) * NR * 64 + i;
            a.part_a[o] = L.A[i];
            a.part_arg[o] = L.ARG[i];
        }
    } else {
    for (int i = tid; i < n * 64; i += NTHR) {
        const int arg = L.ARG[i];
        const float v = arg < 0 ? 0.f : L.A[i];          // isolated node: aggregate 0 (interaction_net.py:188)
        L.A[i] = v;
        tp.A_t(t)[(size_t)lo * 64 + i] = v;
        tp.ARG_t(t)[(size_t)lo * 64 + i] = arg;
    }
    }
    __syncthreads();
    }
    if (!(a.mode & 4)) return;

    SCN_TICK(8);
    // ---- update MLP [x | aggr | sem] -> 128 -> 64, mlp_out 64 -> 128 -> 128 -> 2 ----
    float* s_xp = L.U;                          // [NR][XLD]   x'
    float* s_pu = L.U + NR * GLDS;              // [NR][HLD]   update pre-activation (s_pre[0])
    float* s_po = s_pu + NR * HLD;              // [2][NR][HLD]
    float* s_dec = L.P;                         // [NR][4]     (P is dead)
    AF<5> f_u0;
    AF<4> f_u1, f_o1;
    AF<2> f_o0;
    {
        SB sb(L.sb, L.rs, 160, NR);
        af_first<5>(f_u0, g.update.wf[0], 0, 8, 1, tid);
        af_first<4>(f_u1, g.update.wf[1], 0, 4, 1, tid);
        split16([&](int r, int k) {
            if (k < 64) return L.x[r * XLD + k];
            if (k < 128) return L.A[r * 64 + (k - 64)];
            return a.sem[(size_t)(lo + r) * NC + (k - 128)];
        }, 128 + NC, 160, n, NR, sb, tid);
        __syncthreads();
        dense_rows<5>(g.update.wf[0], g.update.wsc[0], (L.par + Par::U_B0), H, 1, n, sb, s_pu, HLD, tid, &f_u0);
        af_first<2>(f_o0, g.mlp_out.wf[0], 0, 8, 1, tid);
        __syncthreads();
    }
    {
        SB sb(L.sb, L.rs, 128, NR);
        ln_relu_split16([&](int r) { return s_pu + r * HLD; }, n, NR, (L.par + Par::U_G0), (L.par + Par::U_E0), sb, tid);
        for (int i = tid; i < n * H; i += NTHR) tp.PRE_U_t(t)[(size_t)lo * H + i] = s_pu[(i >> 7) * HLD + (i & 127)];
        __syncthreads();
        dense_rows<4>(g.update.wf[1], g.update.wsc[1], (L.par + Par::U_B1), 64, 1, n, sb, s_xp, XLD, tid, &f_u1);
        af_first<4>(f_o1, g.mlp_out.wf[1], 0, 8, 1, tid);
        __syncthreads();
    }
    {
        SB sb(L.sb, L.rs, 64, NR);
        split16([&](int r, int k) { return s_xp[r * XLD + k]; }, 64, 64, n, NR, sb, tid);
        __syncthreads();
        dense_rows<2>(g.mlp_out.wf[0], g.mlp_out.wsc[0], (L.par + Par::O_B0), H, 1, n, sb, s_po, HLD, tid, &f_o0);
        __syncthreads();
    }
    {
        SB sb(L.sb, L.rs, 128, NR);
        ln_relu_split16([&](int r) { return s_po + r * HLD; }, n, NR, (L.par + Par::O_G0), (L.par + Par::O_E0), sb, tid);
        __syncthreads();
        dense_rows<4>(g.mlp_out.wf[1], g.mlp_out.wsc[1], (L.par + Par::O_B1), H, 1, n, sb, s_po + NR * HLD, HLD, tid, &f_o1);
        for (int i = tid; i < n * H; i += NTHR) {
            const int rr = i >> 7, c = i & 127;
            tp.PRE_O_t(t)[(size_t)(lo + rr) * 2 * H + c] = s_po[rr * HLD + c];
        }
        __syncthreads();
    }
    // last layer 128 -> 2 on the vector ALUs, fused with the LayerNorm of its input (16 lanes per row)
    {
        const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
        const float* gam = (L.par + Par::O_G1);
        const float* bet = (L.par + Par::O_E1);
        const float* W2 = (L.par + Par::O_W2);             // (2, 128) torch layout
        for (int r0 = 4 * wave; r0 < NR; r0 += 4 * NWAVE) {
            const int r = r0 + sub;
            const bool live = r < n;
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = live ? s_po[NR * HLD + r * HLD + 8 * q + i] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += x[i];
            const float mean = grp_sum(s) / 128.0f;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = x[i] - mean; v = fmaf(d, d, v); }
            const float rstd = 1.0f / sqrtf(grp_sum(v) / 128.0f + LN_EPS);
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 8 * q + i;
                const float y = fmaxf((x[i] - mean) * rstd * gam[c] + bet[c], 0.f);
                d0 = fmaf(y, W2[c], d0);
                d1 = fmaf(y, W2[H + c], d1);
            }
            d0 = grp_sum(d0);
            d1 = grp_sum(d1);
            if (live && q == 0) {
                s_dec[r * 4 + 0] = d0 + (L.par + Par::O_B2)[0];
                s_dec[r * 4 + 1] = d1 + (L.par + Par::O_B2)[1];
            }
        }
        for (int i = tid; i < n * H; i += NTHR) {
            const int rr = i >> 7, c = i & 127;
            tp.PRE_O_t(t)[(size_t)(lo + rr) * 2 * H + H + c] = s_po[NR * HLD + rr * HLD + c];
        }
    }
    __syncthreads();

    SCN_TICK(9);
    // ---- kinematic bicycle step, local pose (traffic_model.py:645-680) ----
    const bool more = t < a.FT - 1;
    float* s_loc = L.Q;                    // [NR][4]   (Q is dead)
    if (tid < NR) {
        float loc[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < n) {
            const int r = lo + tid;
            const float* st = tp.state_t(t) + (size_t)r * 8;
            BikeFwd bk;
            bike_forward(dp, st, s_dec[tid * 4 + 0], s_dec[tid * 4 + 1], a.lw[r * 2], bk);
            tp.DEC_t(t)[(size_t)r * 4 + 0] = s_dec[tid * 4 + 0];
            tp.DEC_t(t)[(size_t)r * 4 + 1] = s_dec[tid * 4 + 1];
            float* tr = a.traj + ((size_t)r * a.FT + t) * 4;
            for (int i = 0; i < 4; ++i) tr[i] = bk.out[i];
            float gin[4] = {bk.out[0], bk.out[1], bk.out[2], bk.out[3]};
            if (a.ext && tid == 0 && tile == 0) {                     // the ego is the first agent of its scene
                const float* e = a.ext + ((size_t)b * a.FT + t) * 4;
                for (int i = 0; i < 4; ++i) gin[i] = e[i];
            }
            rel_pose(st, gin, loc);
            float* lo_ = tp.loc_t(t) + (size_t)r * 4;
            for (int i = 0; i < 4; ++i) lo_[i] = loc[i];
            if (more) {
                float* ns = tp.state_t(t + 1) + (size_t)r * 8;
                for (int i = 0; i < 6; ++i) ns[i] = bk.out[i];
                float* np = tp.pos_t(t + 1) + (size_t)r * 4;
                for (int i = 0; i < 4; ++i) np[i] = gin[i];
            }
        }
        for (int i = 0; i < 4; ++i) s_loc[tid * 4 + i] = loc[i];
    }
    SCN_TICK(10);
    if (!more) return;

    // ---- 3-layer GRU memory step (traffic_model.py:684-688): gates on the matrix cores ----
    float* s_gi = L.U;                                                // [NR][GLDS]   (the MLP scratch is dead)
    float* s_gh = s_gi + NR * GLDS;                                   // [NR][GLDS]
    float* s_h = L.P + NR * 4;                                        // [NR][XLD]   hidden input of the layer
    float* s_hn = L.x;                                                // [NR][XLD]   layer output (x is dead)
    SB sbh(L.sb, L.rs, 64, NR);
    SB sbx(L.sb + SB::bytes(64, NR), L.rs + NR, 64, NR);
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        AF<2> f_hh, f_ih;
        af_first<2>(f_hh, gf.whh_f[l], 0, 12, 1, tid);
        if (l > 0) af_first<2>(f_ih, gf.wih_f[l], 0, 12, 1, tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * NTHR;
            if (i < n * 64) s_h[(i >> 6) * XLD + (i & 63)] = h_in[l][k];
        }
        __syncthreads();
        split16([&](int r, int k) { return s_h[r * XLD + k]; }, 64, 64, n, NR, sbh, tid);
        if (l > 0) split16([&](int r, int k) { return s_hn[r * XLD + k]; }, 64, 64, n, NR, sbx, tid);
        __syncthreads();
        dense_rows<2>(gf.whh_f[l], gf.hh_sc[l], (L.par + Par::R_BHH + l * GLD), GLD, 1, n, sbh, s_gh, GLDS, tid, &f_hh);
        if (l > 0) dense_rows<2>(gf.wih_f[l], gf.ih_sc[l], (L.par + Par::R_BIH + l * GLD), GLD, 1, n, sbx, s_gi, GLDS, tid, &f_ih);
        else {
            for (int i = tid; i < n * GLD; i += NTHR) {
                const int rr = i / GLD, c = i - rr * GLD;
                float v = L.par[Par::R_BIH + c];
#pragma unroll
                for (int k = 0; k < 4; ++k) v = fmaf(s_loc[rr * 4 + k], L.par[Par::R_WIH0T + k * GLD + c], v);
                s_gi[rr * GLDS + c] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < n * 64; i += NTHR) {
            const int rr = i >> 6, c = i & 63, r = lo + rr;
            const float rg = sigmoidf_(s_gi[rr * GLDS + c] + s_gh[rr * GLDS + c]);
            const float zg = sigmoidf_(s_gi[rr * GLDS + 64 + c] + s_gh[rr * GLDS + 64 + c]);
            const float ghn = s_gh[rr * GLDS + 128 + c];
            const float ng = tanhf(s_gi[rr * GLDS + 128 + c] + rg * ghn);
            const float hn = (1.0f - zg) * ng + zg * s_h[rr * XLD + c];
            float* gq = tp.GATES_t(t) + ((size_t)r * 3 + l) * 256;
            gq[c] = rg;
            gq[64 + c] = zg;
            gq[128 + c] = ng;
            gq[192 + c] = ghn;
            tp.mem_t(t + 1)[((size_t)r * 3 + l) * 64 + c] = hn;
            if (l == 2) tp.pf_t(t + 1)[(size_t)r * 64 + c] = hn;
            s_hn[rr * XLD + c] = hn;
        }
    }
    SCN_TICK(11);
#undef SCN_TICK
}


// =====================================================================================================================
// backward: the whole reverse-time sweep of one scene in one launch.   grid = B, block = 512
//   d_traj (NA, FT, 4) -> dz (NA, 32); reads the tape of the forward sweep (pre-activations, gates, arg-max, poses).
// =====================================================================================================================
struct BwdLds {
    // adjoint state carried from step t + 1 to step t
    float* g_state;  // [NR][8]
    float* g_pos;    // [NR][4]    adjoint of pos_{t+1}
    float* d_loc;    // [NR][4]    adjoint of the local pose fed to the GRU
    float* g_pf;     // [NR][XLD]  adjoint of past_feat_{t+1}
    float* g_mem;    // [3][NR][XLD]
    float* dz;       // [NR][32]
    // per step
    float* pos;      // [NR][4]
    float* dP;       // [NR][HLD]
    float* dQ;       // [NR][HLD]
    float* gpos_n;   // [NR][4]    adjoint of pos_t being assembled
    float* gin2;     // [NR][HLD]  adjoint of [x | aggr | sem]
    int* ARG;        // [NR][64]
    float* gdec;     // [NR][4]
    float* rs;       // [EC]
    float* rel;      // [EC][4]
    int* ei;         // [EC]
    int* ej;         // [EC]
    unsigned* nan;   // [EC]
    unsigned char* sb;
    float* U;
    float* par;      // [Par::FLOATS]
    static constexpr int SB_BYTES = SB::bytes(128, EC) > 2 * SB::bytes(192, NR) ? SB::bytes(128, EC) : 2 * SB::bytes(192, NR);
    static constexpr int U_FLOATS = EC * HLD + 3 * EC * 4 > 2 * NR * GLDS + NR * XLD ? EC * HLD + 3 * EC * 4 : 2 * NR * GLDS + NR * XLD;
    static constexpr int P_FLOATS = NR * 8 + NR * 4 + NR * 4 + NR * XLD + 3 * NR * XLD + NR * 32 + NR * 4 + 2 * NR * HLD + NR * 4 + NR * HLD +
                                    NR * 64 + NR * 4 + EC + EC * 4 + 3 * EC;
    static constexpr size_t BYTES = (size_t)P_FLOATS * 4 + SB_BYTES + (size_t)(U_FLOATS + Par::FLOATS) * 4 + 64;
    __device__ BwdLds(float* base) {
        g_state = base;
        g_pos = g_state + NR * 8;
        d_loc = g_pos + NR * 4;
        g_pf = d_loc + NR * 4;
        g_mem = g_pf + NR * XLD;
        dz = g_mem + 3 * NR * XLD;
        pos = dz + NR * 32;
        dP = pos + NR * 4;
        dQ = dP + NR * HLD;
        gpos_n = dQ + NR * HLD;
        gin2 = gpos_n + NR * 4;
        ARG = reinterpret_cast<int*>(gin2 + NR * HLD);
        gdec = reinterpret_cast<float*>(ARG + NR * 64);
        rs = gdec + NR * 4;
        rel = rs + EC;
        ei = reinterpret_cast<int*>(rel + EC * 4);
        ej = ei + EC;
        nan = reinterpret_cast<unsigned*>(ej + EC);
        sb = reinterpret_cast<unsigned char*>(nan + EC);
        U = reinterpret_cast<float*>(sb + SB_BYTES);
        par = U + U_FLOATS;
    }
};
static_assert(BwdLds::P_FLOATS % 4 == 0, "split buffer must stay 16-byte aligned");

struct SweepArgs {
    int FT, NC, max_n;
    const float* sem;
    const float* lw;
    const float* ext;
    const int32_t* ptr;
    const float* par;          // StriveDecoder.scene_par
    const float* g_traj;       // (NA, FT, 4)
    float* dz;                 // (NA, 32)
    // stepwise form only (scene_bwd_sweep_kernel<PROF, true>): the step of this launch, workgroups per scene, and the two global
    // buffers that carry a scene's sweep from launch to launch
    int t, K;
    float* part;               // (2, B, K, PART_FLOATS)   dP | dQ | gpos_n of the edge chunks workgroup k walked, by step parity:
                               // the launch of step t reads the sums of step t + 1 while its workgroups write those of step t
    float* state;              // (B, K, STATE_FLOATS)  g_state | g_mem | dz | gin2 of workgroup k (all K hold the same values)
};
constexpr int SWEEP_PART_FLOATS = 2 * NR * HLD + NR * 4;
constexpr int SWEEP_STATE_FLOATS = NR * 8 + 3 * NR * XLD + NR * 32 + NR * HLD;

// one product of the sweep with a caller-supplied epilogue (identity k-step lists)
template <int NK, typename Epi>
__device__ __forceinline__ void dense_rows_epi(const uint4* __restrict__ frag, float wscale, int nt0, int NTL, int nrt, const SB& sb,
                                               int tid, Epi epi, const AF<NK>* pre = nullptr) {
    int ks[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) ks[i] = i;
    mma_tiles<NK>(frag, NK, wscale, nt0, NTL, nrt, ks, ks, sb, tid, epi, pre);
}

// STEP = false: grid = B, the whole sweep in one launch (above).
// STEP = true (round 5): grid = (B, K), ONE reverse step per launch, FT + 1 launches.  At 16 agents per scene two thirds of the
// one-launch sweep are the 240 edge rows of a scene walked as four 64-row chunks by ONE workgroup, on 32 of the chip's 256 CUs
// (profiles/r04_scene_phase_32x16.txt: 1.1 of 1.7 ms).  Here workgroup k of a scene walks chunks k, k + K, ... and every one of
// the K workgroups repeats the scene's node-level phases (GRU, dynamics, mlp_out, update; the edge partials and mlp_in) --
// deterministic, so all K hold the same adjoint state.  What crosses workgroups are the chunk sums dP_i, dQ_j and the pose
// adjoints: each workgroup leaves its partial sums in `part`, the launch boundary is the barrier, and the NEXT launch adds the
// partials in workgroup order before it finishes that step's node1 / mlp_in phase and starts the following step.  Launch for step
// t = [finish step t + 1] -> GRU(t) -> dynamics .. update(t) -> own edge chunks of t; the launch with t = -1 finishes step 0 and
// writes dz.  With one chunk per workgroup (n (n - 1) <= 64 K) every sum is formed in the order of the one-launch sweep: the two
// forms give the same bits (tests/test_emu_kernels.py).
template <bool PROF, bool STEP = false>
static __global__ __launch_bounds__(NTHR) void scene_bwd_sweep_kernel(GNNDev g, GRUDev gru, GRUFrag gf, DynParams dp, SweepArgs a,
                                                                      Tape tp, unsigned long long* prof) {
    HIP_DYNAMIC_SHARED(float, smem)
    BwdLds L(smem);
    int tid = threadIdx.x;
    const int b = blockIdx.x, NC = a.NC, H = STRIVE_HID, FT = a.FT;
    const int kw = STEP ? (int)blockIdx.y : 0, KW = STEP ? a.K : 1;
    const int lo = a.ptr[b], n = a.ptr[b + 1] - lo;
    if (n <= 0) return;
    const int E = n * (n - 1);
    const float* Wrel = L.par + Par::E_WREL;       // (4, 128)
    long long tick = PROF ? clock64() : 0;
#define SCN_TICK(id)                                                        \
    do {                                                                    \
        if (PROF && threadIdx.x == 0 && blockIdx.x == 0) {                  \
            const long long now_ = clock64();                               \
            prof[id] += (unsigned long long)(now_ - tick);                  \
            tick = now_;                                                    \
        }                                                                   \
    } while (0)

    for (int i = tid; i < NR * 8; i += NTHR) L.g_state[i] = 0.f;
    for (int i = tid; i < NR * 4; i += NTHR) { L.g_pos[i] = 0.f; L.d_loc[i] = 0.f; L.gpos_n[i] = 0.f; }
    for (int i = tid; i < NR * XLD; i += NTHR) L.g_pf[i] = 0.f;
    for (int i = tid; i < 3 * NR * XLD; i += NTHR) L.g_mem[i] = 0.f;
    for (int i = tid; i < NR * 32; i += NTHR) L.dz[i] = 0.f;
    for (int i = tid; i < NR * HLD; i += NTHR) { L.dP[i] = 0.f; L.dQ[i] = 0.f; }
    par_stage(L.par, a.par, tid);
    SCN_SYNC(tid);

    // (tape rows and weight tiles are requested one stage ahead of their use: before the product or pass that precedes it)
    // ================= GRU memory step backward (consumes g_mem, g_pf; produces g_mem, d_loc) =================
    auto sec_gru = [&](const int t) {
        SCN_PHASE(tid);
        const bool more = t < FT - 1;
        if (more) {
            float* s_dgi = L.U;                     // [NR][GLDS]
            float* s_dgh = L.U + NR * GLDS;         // [NR][GLDS]
            float* s_dx = L.U + 2 * NR * GLDS;      // [NR][XLD]   adjoint flowing to the layer below
            SB sbh(L.sb, L.rs, 192, NR);
            SB sbi(L.sb + SB::bytes(192, NR), L.rs + NR, 192, NR);
            // gates (r, z, n, W_hn h + b_hn) and hidden input of a layer: requested one layer ahead
            float gq[2][5], gq_next[2][5];
            auto load_gates = [&](int l, float (&o)[2][5]) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = tid + k * NTHR;
                    if (i < n * 64) {
                        const int r = lo + (i >> 6), c = i & 63;
                        const float* gp = tp.GATES_t(t) + ((size_t)r * 3 + l) * 256;
                        o[k][0] = gp[c];
                        o[k][1] = gp[64 + c];
                        o[k][2] = gp[128 + c];
                        o[k][3] = gp[192 + c];
                        o[k][4] = tp.mem_t(t)[((size_t)r * 3 + l) * 64 + c];
                    }
                }
            };
            load_gates(2, gq_next);
#pragma unroll
            for (int l = 2; l >= 0; --l) {
                SCN_PHASE(tid);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 5; ++j) gq[k][j] = gq_next[k][j];
                if (l > 0) load_gates(l - 1, gq_next);
                AF<6> f_hh;
                af_first<6>(f_hh, gf.whh_bf[l], 0, 4, 1, tid);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = tid + k * NTHR;
                    if (i < n * 64) {
                        const int rr = i >> 6, c = i & 63;
                        float dh = L.g_mem[(l * NR + rr) * XLD + c];
                        dh += (l == 2) ? L.g_pf[rr * XLD + c] : s_dx[rr * XLD + c];
                        const float rg = gq[k][0], z = gq[k][1], nn = gq[k][2], ghn = gq[k][3], h = gq[k][4];
                        const float dn = dh * (1.0f - z);
                        const float dzg = dh * (h - nn);
                        const float dpn = dn * (1.0f - nn * nn);
                        const float dr = dpn * ghn;
                        const float dpz = dzg * z * (1.0f - z);
                        const float dpr = dr * rg * (1.0f - rg);
                        s_dgi[rr * GLDS + c] = dpr;
                        s_dgi[rr * GLDS + 64 + c] = dpz;
                        s_dgi[rr * GLDS + 128 + c] = dpn;
                        s_dgh[rr * GLDS + c] = dpr;
                        s_dgh[rr * GLDS + 64 + c] = dpz;
                        s_dgh[rr * GLDS + 128 + c] = dpn * rg;
                        L.g_mem[(l * NR + rr) * XLD + c] = dh * z;            // direct path h' = ... + z h
                    }
                }
                SCN_SYNC(tid);
                split16([&](int r, int k) { return s_dgh[r * GLDS + k]; }, GLD, 192, n, NR, sbh, tid);
                if (l > 0) split16([&](int r, int k) { return s_dgi[r * GLDS + k]; }, GLD, 192, n, NR, sbi, tid);
                SCN_SYNC(tid);
                // adjoint of the hidden input += d gh . W_hh ; adjoint of the layer input = d gi . W_ih
                dense_rows_epi<6>(gf.whh_bf[l], gf.hh_sc[l], 0, 4, 1, sbh, tid, [&](int row, int c0, const float (&v)[4]) {
                    if (row < n) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) L.g_mem[(l * NR + row) * XLD + c0 + r] += v[r];
                    }
                }, &f_hh);
                if (l > 0) {
                    dense_rows_epi<6>(gf.wih_bf[l], gf.ih_sc[l], 0, 4, 1, sbi, tid, [&](int row, int c0, const float (&v)[4]) {
                        if (row < n) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) s_dx[row * XLD + c0 + r] = v[r];
                        }
                    });
                } else {
                    // layer 0's input is the 4-wide local pose: d_loc = d gi . W_ih on the vector ALUs, 16 lanes per row
                    const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
                    for (int r0 = 4 * wave; r0 < NR; r0 += 4 * NWAVE) {
                        const int rr = r0 + sub;
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        if (rr < n) {
#pragma unroll
                            for (int j = 0; j < 12; ++j) {
                                const int c = q + 16 * j;
                                const float ge = s_dgi[rr * GLDS + c];
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = fmaf(ge, L.par[Par::R_WIH0 + c * 4 + k], v[k]);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = grp_sum(v[k]);
                        if (rr < n && q == 0)
                            for (int k = 0; k < 4; ++k) L.d_loc[rr * 4 + k] = v[k];
                    }
                }
                SCN_SYNC(tid);
            }
        }
        SCN_TICK(0);
    };

    // ================= dynamics + mlp_out + update backward =================
    auto sec_node2 = [&](const int t) {
        const bool more = t < FT - 1;
        SCN_PHASE(tid);
        AF<4> f_o1, f_o0, f_u0;
        AF<2> f_u1;
        XR<1> x_o1, x_o0, x_u;
        xr_load<1>(x_o1, [&](int r) { return tp.PRE_O_t(t) + (size_t)(lo + r) * 2 * H + H; }, n, tid);
        af_first<4>(f_o1, g.mlp_out.wbf[1], 0, 8, 1, tid);
        if (tid < n * 4) L.pos[tid] = tp.pos_t(t)[(size_t)lo * 4 + tid];
        for (int i = tid; i < n * 64; i += NTHR) L.ARG[i] = tp.ARG_t(t)[(size_t)lo * 64 + i];
        if (tid < NR) {
            float gdec[2] = {0.f, 0.f};
            if (tid < n) {
                const int r = lo + tid;
                const float* st = tp.state_t(t) + (size_t)r * 8;
                const float* dc = tp.DEC_t(t) + (size_t)r * 4;
                BikeFwd bk;
                bike_forward(dp, st, dc[0], dc[1], a.lw[r * 2], bk);
                bool forced = false;
                float gin[4] = {bk.out[0], bk.out[1], bk.out[2], bk.out[3]};
                if (a.ext && tid == 0) {
                    forced = true;
                    const float* e = a.ext + ((size_t)b * FT + t) * 4;
                    for (int i = 0; i < 4; ++i) gin[i] = e[i];
                }
                float gbike[6];
                const float* gt = a.g_traj + ((size_t)r * FT + t) * 4;
                for (int i = 0; i < 4; ++i) gbike[i] = gt[i];
                gbike[4] = gbike[5] = 0.f;
                float gfr[4] = {0.f, 0.f, 0.f, 0.f};
                if (more) {
                    float gpo[4] = {0.f, 0.f, 0.f, 0.f};
                    rel_pose_bwd(st, gin, L.d_loc + tid * 4, gfr, gpo);
                    for (int i = 0; i < 6; ++i) gbike[i] += L.g_state[tid * 8 + i];
                    if (!forced)
                        for (int i = 0; i < 4; ++i) gbike[i] += gpo[i] + L.g_pos[tid * 4 + i];
                }
                float gst[6];
                bike_backward(dp, bk, gbike, gst, gdec);
                for (int i = 0; i < 6; ++i) L.g_state[tid * 8 + i] = gst[i] + (i < 4 ? gfr[i] : 0.f);
            }
            L.gdec[tid * 4 + 0] = gdec[0];
            L.gdec[tid * 4 + 1] = gdec[1];
        }
        SCN_SYNC(tid);
        SCN_TICK(1);
        {
            float* s_ga = L.U;                  // [NR][HLD]
            float* s_gxp = L.U + NR * HLD;      // [NR][XLD]
            SB sb(L.sb, L.rs, 128, NR);
            const float* W2 = (L.par + Par::O_W2);   // (2, 128)
            // mlp_out layer 2 (2 <- 128) and the LayerNorm before it
            ln_relu_bwd16<1>(x_o1,
                             [&](int r, int k0, float (&d)[8]) {
                                 const float g0 = L.gdec[r * 4 + 0], g1 = L.gdec[r * 4 + 1];
#pragma unroll
                                 for (int i = 0; i < 8; ++i) d[i] = g0 * W2[k0 + i] + g1 * W2[H + k0 + i];
                             },
                             n, NR, (L.par + Par::O_G1), (L.par + Par::O_E1), &sb, nullptr, 0, tid);
            SCN_SYNC(tid);
            xr_load<1>(x_o0, [&](int r) { return tp.PRE_O_t(t) + (size_t)(lo + r) * 2 * H; }, n, tid);
            dense_rows_epi<4>(g.mlp_out.wbf[1], g.mlp_out.wsc[1], 0, 8, 1, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_ga[row * HLD + c0 + r] = v[r];
            }, &f_o1);
            af_first<4>(f_o0, g.mlp_out.wbf[0], 0, 4, 1, tid);
            SCN_SYNC(tid);
            ln_relu_bwd16<1>(x_o0,
                             [&](int r, int k0, float (&d)[8]) {
#pragma unroll
                                 for (int i = 0; i < 8; ++i) d[i] = s_ga[r * HLD + k0 + i];
                             },
                             n, NR, (L.par + Par::O_G0), (L.par + Par::O_E0), &sb, nullptr, 0, tid);
            SCN_SYNC(tid);
            dense_rows_epi<4>(g.mlp_out.wbf[0], g.mlp_out.wsc[0], 0, 4, 1, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_gxp[row * XLD + c0 + r] = v[r];
            }, &f_o0);
            af_first<2>(f_u1, g.update.wbf[1], 0, 8, 1, tid);
            SCN_SYNC(tid);
            SCN_TICK(2);
            SCN_PHASE(tid);
            // update MLP: layer 1 (64 <- 128), LayerNorm, layer 0 (128 <- 128 + NC)
            SB sb64(L.sb, L.rs, 64, NR);
            split16([&](int r, int k) { return s_gxp[r * XLD + k]; }, 64, 64, n, NR, sb64, tid);
            SCN_SYNC(tid);
            xr_load<1>(x_u, [&](int r) { return tp.PRE_U_t(t) + (size_t)(lo + r) * H; }, n, tid);
            dense_rows_epi<2>(g.update.wbf[1], g.update.wsc[1], 0, 8, 1, sb64, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_ga[row * HLD + c0 + r] = v[r];
            }, &f_u1);
            af_first<4>(f_u0, g.update.wbf[0], 0, 8, 1, tid);
            SCN_SYNC(tid);
            ln_relu_bwd16<1>(x_u,
                             [&](int r, int k0, float (&d)[8]) {
#pragma unroll
                                 for (int i = 0; i < 8; ++i) d[i] = s_ga[r * HLD + k0 + i];
                             },
                             n, NR, (L.par + Par::U_G0), (L.par + Par::U_E0), &sb, nullptr, 0, tid);
            SCN_SYNC(tid);
            // columns [0, 64) = update part of dL/dx, [64, 128) = dL/d(aggregated message); the sem columns are not needed
            dense_rows_epi<4>(g.update.wbf[0], g.update.wsc[0], 0, 8, 1, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) L.gin2[row * HLD + c0 + r] = v[r];
            }, &f_u0);
            SCN_SYNC(tid);
            SCN_TICK(3);
        }
    };

    // ================= edge backward, chunks of EC rows: e_first, e_first + e_stride, ... =================
    auto sec_edges = [&](const int t, const int e_first, const int e_stride) {
        {
            float* s_g = L.U;                         // [EC][HLD]
            float* s_grel = L.U + EC * HLD;           // [EC][4]
            float* s_gfr = s_grel + EC * 4;           // [EC][4]
            float* s_gpo = s_gfr + EC * 4;            // [EC][4]
            SB sb(L.sb, L.rs, 128, EC);
            SB sb64(L.sb, L.rs, 64, EC);
            // the wave's weight tiles of both products, kept across the chunks
            AF<2> f_e2;
            AF<4> f_e1;
            if (E > e_first) {
                af_first<2>(f_e2, g.edge.wbf[2], 0, 8, 4, tid);
                af_first<4>(f_e1, g.edge.wbf[1], 0, 8, 4, tid);
            }
            for (int e0 = e_first; e0 < E; e0 += e_stride) {
                SCN_PHASE(tid);
                const int ne = (E - e0) < EC ? (E - e0) : EC;
                const int ne_pad = (ne + 15) & ~15, nrt = ne_pad >> 4;
                if (tid < EC) {
                    int i = 0, jl = 0;
                    float rel[4] = {0.f, 0.f, 0.f, 0.f};
                    unsigned nm = 0;
                    if (tid < ne) {
                        const int e = e0 + tid;
                        i = e / (n - 1);
                        const int jj = e - i * (n - 1);
                        jl = jj + (jj >= i ? 1 : 0);
                        rel_pose(L.pos + i * 4, L.pos + jl * 4, rel);
                        for (int d = 0; d < 4; ++d)
                            if (rel[d] != rel[d]) { rel[d] = 0.f; nm |= 1u << d; }
                    }
                    L.ei[tid] = i;
                    L.ej[tid] = jl;
                    L.nan[tid] = nm;
                    for (int d = 0; d < 4; ++d) L.rel[tid * 4 + d] = rel[d];
                }
                SCN_SYNC(tid);
                // the chunk's pre-activation rows (both hidden layers) from the tape, requested before the first product
                XR<2> x_e1, x_e0;
                xr_load<2>(x_e1, [&](int r) { return tp.PRE_E_t(t) + ((size_t)(lo + L.ei[r]) * a.max_n + L.ej[r]) * 2 * H + H; }, ne, tid);
                xr_load<2>(x_e0, [&](int r) { return tp.PRE_E_t(t) + ((size_t)(lo + L.ei[r]) * a.max_n + L.ej[r]) * 2 * H; }, ne, tid);
                // d(aggregate) goes to the arg-max edge of every channel
                split16([&](int r, int k) {
                    const int i = L.ei[r];
                    return (L.ARG[i * 64 + k] == lo + L.ej[r]) ? L.gin2[i * HLD + 64 + k] : 0.f;
                }, 64, 64, ne, ne_pad, sb64, tid);
                SCN_SYNC(tid);
                SCN_TICK(4);
                dense_rows_epi<2>(g.edge.wbf[2], g.edge.wsc[2], 0, 8, nrt, sb64, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_g[row * HLD + c0 + r] = v[r];
                }, &f_e2);
                SCN_SYNC(tid);
                SCN_TICK(5);
                ln_relu_bwd16<2>(x_e1,
                                 [&](int r, int k0, float (&d)[8]) {
#pragma unroll
                                     for (int i = 0; i < 8; ++i) d[i] = s_g[r * HLD + k0 + i];
                                 },
                                 ne, ne_pad, (L.par + Par::E_G1), (L.par + Par::E_E1), &sb, nullptr, 0, tid);
                SCN_SYNC(tid);
                SCN_TICK(6);
                dense_rows_epi<4>(g.edge.wbf[1], g.edge.wsc[1], 0, 8, nrt, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_g[row * HLD + c0 + r] = v[r];
                }, &f_e1);
                SCN_SYNC(tid);
                SCN_TICK(7);
                // d e1 (adjoint of the factorised layer 0's output) in place
                ln_relu_bwd16<2>(x_e0,
                                 [&](int r, int k0, float (&d)[8]) {
#pragma unroll
                                     for (int i = 0; i < 8; ++i) d[i] = s_g[r * HLD + k0 + i];
                                 },
                                 ne, ne_pad, (L.par + Par::E_G0), (L.par + Par::E_E0), nullptr, s_g, HLD, tid);
                SCN_SYNC(tid);
                SCN_TICK(8);
                // dP_i += sum_j d e1_ij, dQ_j += sum_i d e1_ij: thread = (channel, quarter of the targets / sources); the rows of a
                // target are contiguous, the row of (target i, source j) is i (n - 1) + j - (j > i); sums in ascending row order
                {
                    const int c = tid & 127, part = tid >> 7;
                    const int i_first = e0 / (n - 1), i_last = (e0 + ne - 1) / (n - 1);
                    for (int i = i_first + part; i <= i_last; i += 4) {
                        int eb = i * (n - 1), ee = eb + (n - 1);
                        eb = eb < e0 ? e0 : eb;
                        ee = ee > e0 + ne ? e0 + ne : ee;
                        float acc = 0.f;
                        for (int e = eb; e < ee; ++e) acc += s_g[(e - e0) * HLD + c];
                        L.dP[i * HLD + c] += acc;
                    }
                    for (int j = part; j < n; j += 4) {
                        float acc = 0.f;
                        for (int i = i_first; i <= i_last; ++i) {
                            if (i == j) continue;
                            const int e = i * (n - 1) + j - (j > i ? 1 : 0);
                            if (e >= e0 && e < e0 + ne) acc += s_g[(e - e0) * HLD + c];
                        }
                        L.dQ[j * HLD + c] += acc;
                    }
                }
                // d rel = d e1 . W_rel^T
                {
                    const int lane = tid & 63, wave = tid >> 6, q = lane & 15, sub = lane >> 4;
                    float w8[4][8];
#pragma unroll
                    for (int d = 0; d < 4; ++d)
#pragma unroll
                        for (int i = 0; i < 8; ++i) w8[d][i] = Wrel[d * H + 8 * q + i];
                    for (int r0 = 4 * wave; r0 < ne_pad; r0 += 4 * NWAVE) {
                        const int r = r0 + sub;
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        if (r < ne) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float ge = s_g[r * HLD + 8 * q + i];
#pragma unroll
                                for (int d = 0; d < 4; ++d) v[d] = fmaf(ge, w8[d][i], v[d]);
                            }
                        }
#pragma unroll
                        for (int d = 0; d < 4; ++d) v[d] = grp_sum(v[d]);
                        if (r < ne && q == 0)
                            for (int d = 0; d < 4; ++d) s_grel[r * 4 + d] = v[d];
                    }
                }
                SCN_SYNC(tid);
                SCN_TICK(9);
                if (tid < ne) {
                    float gfr[4] = {0.f, 0.f, 0.f, 0.f}, gpo[4] = {0.f, 0.f, 0.f, 0.f}, gr4[4];
                    for (int d = 0; d < 4; ++d) gr4[d] = (L.nan[tid] >> d) & 1u ? 0.f : s_grel[tid * 4 + d];
                    rel_pose_bwd(L.pos + L.ei[tid] * 4, L.pos + L.ej[tid] * 4, gr4, gfr, gpo);
                    for (int d = 0; d < 4; ++d) { s_gfr[tid * 4 + d] = gfr[d]; s_gpo[tid * 4 + d] = gpo[d]; }
                }
                SCN_SYNC(tid);
                if (tid < n * 4) {
                    // adjoint of pos_k: as a frame (its own rows, contiguous) and as a source (one row per target of the chunk)
                    const int k = tid >> 2, d = tid & 3;
                    const int i_first = e0 / (n - 1), i_last = (e0 + ne - 1) / (n - 1);
                    float acc = 0.f;                  // the chunk's own sum first (like dP / dQ): the same association whether the
                    int eb = k * (n - 1), ee = eb + (n - 1);      // chunks are walked by one workgroup or by several
                    eb = eb < e0 ? e0 : eb;
                    ee = ee > e0 + ne ? e0 + ne : ee;
                    for (int e = eb; e < ee; ++e) acc += s_gfr[(e - e0) * 4 + d];
                    for (int i = i_first; i <= i_last; ++i) {
                        if (i == k) continue;
                        const int e = i * (n - 1) + k - (k > i ? 1 : 0);
                        if (e >= e0 && e < e0 + ne) acc += s_gpo[(e - e0) * 4 + d];
                    }
                    L.gpos_n[tid] += acc;
                }
                SCN_SYNC(tid);
                SCN_TICK(10);
            }
        }
    };

    // ================= node features backward: edge layer-0 partials, mlp_in =================
    auto sec_node1 = [&](const int t) {
        SCN_PHASE(tid);
        {
            float* s_gx = L.U;                  // [NR][XLD]
            float* s_ga = L.U + NR * XLD;       // [NR][HLD]
            SB sbP(L.sb, L.rs, 128, NR);
            SB sbQ(L.sb + SB::bytes(128, NR), L.rs + NR, 128, NR);
            AF<4> f_p, f_q, f_i1, f_i0a, f_i0b;
            AF<2> f_i2;
            af_first<4>(f_p, g.edge.wbf[0], 0, 4, 1, tid);
            af_first<4>(f_q, g.edge.wbf[0], 4, 4, 1, tid);
            XR<1> x_i1, x_i0;
            split16([&](int r, int k) { return L.dP[r * HLD + k]; }, H, 128, n, NR, sbP, tid);
            split16([&](int r, int k) { return L.dQ[r * HLD + k]; }, H, 128, n, NR, sbQ, tid);
            SCN_SYNC(tid);
            // adjoint of x: dP . W_e0[:, x_i] + dQ . W_e0[:, x_j] + the update MLP's part
            dense_rows_epi<4>(g.edge.wbf[0], g.edge.wsc[0], 0, 4, 1, sbP, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_gx[row * XLD + c0 + r] = v[r] + L.gin2[row * HLD + c0 + r];
            }, &f_p);
            SCN_SYNC(tid);
            dense_rows_epi<4>(g.edge.wbf[0], g.edge.wsc[0], 4, 4, 1, sbQ, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_gx[row * XLD + (c0 - 64) + r] += v[r];
            }, &f_q);
            af_first<2>(f_i2, g.mlp_in.wbf[2], 0, 8, 1, tid);
            SCN_SYNC(tid);
            SCN_TICK(11);
            SCN_PHASE(tid);
            xr_load<1>(x_i1, [&](int r) { return tp.PRE_IN_t(t) + (size_t)(lo + r) * 2 * H + H; }, n, tid);
            for (int i = tid; i < NR * HLD; i += NTHR) { L.dP[i] = 0.f; L.dQ[i] = 0.f; }
            if (tid < NR * 4) { L.g_pos[tid] = L.gpos_n[tid]; L.gpos_n[tid] = 0.f; }
            SB sb64(L.sb, L.rs, 64, NR);
            SB sb(L.sb, L.rs, 128, NR);
            split16([&](int r, int k) { return s_gx[r * XLD + k]; }, 64, 64, n, NR, sb64, tid);
            SCN_SYNC(tid);
            dense_rows_epi<2>(g.mlp_in.wbf[2], g.mlp_in.wsc[2], 0, 8, 1, sb64, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_ga[row * HLD + c0 + r] = v[r];
            }, &f_i2);
            af_first<4>(f_i1, g.mlp_in.wbf[1], 0, 8, 1, tid);
            SCN_SYNC(tid);
            ln_relu_bwd16<1>(x_i1,
                             [&](int r, int k0, float (&d)[8]) {
#pragma unroll
                                 for (int i = 0; i < 8; ++i) d[i] = s_ga[r * HLD + k0 + i];
                             },
                             n, NR, (L.par + Par::IN_G1), (L.par + Par::IN_E1), &sb, nullptr, 0, tid);
            SCN_SYNC(tid);
            xr_load<1>(x_i0, [&](int r) { return tp.PRE_IN_t(t) + (size_t)(lo + r) * 2 * H; }, n, tid);
            dense_rows_epi<4>(g.mlp_in.wbf[1], g.mlp_in.wsc[1], 0, 8, 1, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_ga[row * HLD + c0 + r] = v[r];
            }, &f_i1);
            af_first<4>(f_i0a, g.mlp_in.wbf[0], 0, 4, 1, tid);
            af_first<4>(f_i0b, g.mlp_in.wbf[0], 8, 3, 1, tid);
            SCN_SYNC(tid);
            ln_relu_bwd16<1>(x_i0,
                             [&](int r, int k0, float (&d)[8]) {
#pragma unroll
                                 for (int i = 0; i < 8; ++i) d[i] = s_ga[r * HLD + k0 + i];
                             },
                             n, NR, (L.par + Par::IN_G0), (L.par + Par::IN_E0), &sb, nullptr, 0, tid);
            SCN_SYNC(tid);
            // layer 0 (F <- 128): only the past_feat columns [0, 64) and the latent columns [128 + NC, 160 + NC) are wanted
            dense_rows_epi<4>(g.mlp_in.wbf[0], g.mlp_in.wsc[0], 0, 4, 1, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) L.g_pf[row * XLD + c0 + r] = v[r];
            }, &f_i0a);
            const int zoff = 128 + NC;
            dense_rows_epi<4>(g.mlp_in.wbf[0], g.mlp_in.wsc[0], 8, 3, 1, sb, tid, [&](int row, int c0, const float (&v)[4]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + r - zoff;
                    if (c >= 0 && c < STRIVE_ZDIM) L.dz[row * 32 + c] += v[r];
                }
            }, &f_i0b);
            SCN_SYNC(tid);
            SCN_TICK(12);
        }
    };

    if (!STEP) {
        for (int t = FT - 1; t >= 0; --t) {
            sec_gru(t);
            sec_node2(t);
            sec_edges(t, 0, EC);
            sec_node1(t);
        }
        for (int i = tid; i < n * STRIVE_ZDIM; i += NTHR) a.dz[(size_t)lo * STRIVE_ZDIM + i] = L.dz[i];
    } else {
        const int t = a.t;
        float* st = a.state + ((size_t)b * KW + kw) * SWEEP_STATE_FLOATS;
        if (t < FT - 1) {
            // this workgroup's state of the previous launch ...
            for (int i = tid; i < NR * 8; i += NTHR) L.g_state[i] = st[i];
            for (int i = tid; i < 3 * NR * XLD; i += NTHR) L.g_mem[i] = st[NR * 8 + i];
            for (int i = tid; i < NR * 32; i += NTHR) L.dz[i] = st[NR * 8 + 3 * NR * XLD + i];
            for (int i = tid; i < NR * HLD; i += NTHR) L.gin2[i] = st[NR * 8 + 3 * NR * XLD + NR * 32 + i];
            // ... and the chunk sums of step t + 1 of ALL workgroups of the scene that had chunks, added in workgroup order
            const int nchunk = (E + EC - 1) / EC, kb = nchunk < KW ? nchunk : KW;
            const float* part_prev = a.part + (size_t)((t + 1) & 1) * gridDim.x * KW * SWEEP_PART_FLOATS;
            for (int i = tid; i < NR * HLD; i += NTHR) {
                float p = 0.f, q = 0.f;
                for (int k = 0; k < kb; ++k) {
                    const float* pk = part_prev + ((size_t)b * KW + k) * SWEEP_PART_FLOATS;
                    p += pk[i];
                    q += pk[NR * HLD + i];
                }
                L.dP[i] = p;
                L.dQ[i] = q;
            }
            if (tid < NR * 4) {
                float v = 0.f;
                for (int k = 0; k < kb; ++k) v += part_prev[((size_t)b * KW + k) * SWEEP_PART_FLOATS + 2 * NR * HLD + tid];
                L.gpos_n[tid] = v;
            }
            SCN_SYNC(tid);
            sec_node1(t + 1);
        }
        if (t >= 0) {
            sec_gru(t);
            sec_node2(t);
            sec_edges(t, kw * EC, KW * EC);
            float* pk = a.part + ((size_t)(t & 1) * gridDim.x * KW + (size_t)b * KW + kw) * SWEEP_PART_FLOATS;
            for (int i = tid; i < NR * HLD; i += NTHR) { pk[i] = L.dP[i]; pk[NR * HLD + i] = L.dQ[i]; }
            if (tid < NR * 4) pk[2 * NR * HLD + tid] = L.gpos_n[tid];
            for (int i = tid; i < NR * 8; i += NTHR) st[i] = L.g_state[i];
            for (int i = tid; i < 3 * NR * XLD; i += NTHR) st[NR * 8 + i] = L.g_mem[i];
            for (int i = tid; i < NR * 32; i += NTHR) st[NR * 8 + 3 * NR * XLD + i] = L.dz[i];
            for (int i = tid; i < NR * HLD; i += NTHR) st[NR * 8 + 3 * NR * XLD + NR * 32 + i] = L.gin2[i];
        } else if (kw == 0) {
            for (int i = tid; i < n * STRIVE_ZDIM; i += NTHR) a.dz[(size_t)lo * STRIVE_ZDIM + i] = L.dz[i];
        }
    }
#undef SCN_TICK
}

}  // namespace scn
