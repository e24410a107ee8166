// Convolution gradients of the map CNN on the matrix cores (training path; included from map_cnn_bwd.h).
//
// Number format: gradients span many decades (dL/dy of the first layers is ~1e-7 of the last one's), so the operands are split
// into two bf16 pieces -- bf16 keeps fp32's exponent, no per-tensor scaling pass is needed -- v = hi + lo up to 2^-17 |v|, and
// three products (hi hi + hi lo + lo hi) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16: every product term is accurate
// to ~2^-16, i.e. a TF32-class (the reference's cuDNN default) result at half the error.  Fragment layout of the instruction
// (tools/mfma_probe.hip): lane l carries k = 8 (l >> 5) + 0..7 of row (A) / column (B) l & 31; D: column l & 31, rows
// (r & 3) + 8 (r >> 2) + 4 (l >> 5).
#pragma once

namespace cnnbwd {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t bf16_bits(float v) {          // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bf16_split(float v, uint32_t& hi, uint32_t& lo) {
    hi = bf16_bits(v);
    lo = bf16_bits(v - __uint_as_float(hi << 16));                  // the difference is exact in fp32
}
__device__ __forceinline__ void bf16_split8(const float v[8], uint4& hi, uint4& lo) {
    uint32_t h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bf16_split(v[i], h[i], l[i]);
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}
__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// three-product accumulation of one fragment pair
__device__ __forceinline__ void mfma3(f32x16& acc, const uint4& a_hi, const uint4& a_lo, const uint4& b_hi, const uint4& b_lo) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(b_hi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(b_lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(b_hi), acc, 0, 0, 0);
}

// =============================================================================================
// Data gradient of layer L (1..5):  G_{L-1}[n][ci][iy][ix] = sum_{co,ky,kx} dy[n][co][(iy-ky)/2][(ix-kx)/2] W[co][ci][ky][kx].
// The stride-2 transposed convolution falls apart into the four input-pixel parity classes (py, px) = (iy & 1, ix & 1): class
// pixel (iy2, ix2) gathers dy[iy2 - a][ix2 - b] over the taps (ky, kx) = (py + 2a, px + 2b) -- a stride-1 correlation with
// ceil((KS - py) / 2) x ceil((KS - px) / 2) taps, no structurally zero products.
// As a matrix product per class: rows = input channels (32 per block), columns = class pixels (32 per tile), k = 16 output
// channels of one tap per instruction.  A workgroup takes (S samples, a band of RB class rows, all four classes): the dy rows
// the band needs, all COUT channels, are split into bf16 pieces ONCE on the way into LDS as [piece][sample][co / 8][row][col]
// [8 x bf16] with PAD zero rows / columns around the image (16-byte entries: a fragment is one aligned ds_read_b128 at the
// lane's own pixel offset minus the tap offset).  The weight fragments come pre-packed from dgrad_pack_kernel (L2-resident,
// 1 KB coalesced per wave load) and are reused over the PT pixel tiles of the wave.
// =============================================================================================
template <int L>
struct DgM {
    static constexpr int CIN = LC_IN[L], COUT = LC_OUT[L], KS = LKS[L], IH = LIH[L], OH = LOH[L];
    static constexpr int PAD = (KS - 1) / 2, NYC = (IH + 1) / 2;          // zero border of the dy tile; class rows / columns (max)
    static constexpr int CB = (CIN + 31) / 32, NG = COUT / 16, NOCT = COUT / 8;
    static constexpr int S = L == 5 ? 8 : L == 4 ? 2 : 1;               // samples per unit (small images)
    static constexpr int RB = L <= 2 ? 4 : NYC;                          // class rows per band
    static constexpr int NBAND = (NYC + RB - 1) / RB;
    static constexpr int ROWS = RB + PAD, PW = OH + 2 * PAD;
    static constexpr int ENTRIES = S * NOCT * ROWS * PW, PIECE_B = ENTRIES * 16;
    static constexpr size_t LDS_BYTES = 2 * (size_t)PIECE_B;
    // pixel tiles: both column-parity classes of a row parity share ONE linearisation (sample, row, ix2 < NYC), so that a lane
    // holds horizontally adjacent input pixels (2 ix2, 2 ix2 + 1) and stores them back to back: the stride-2 stores of one
    // class alone left half-written 128-byte lines for the other class to complete much later (layer 1 writes 256 MB per 256
    // samples; that, not the matrix work, bounded it)
    static constexpr int NTILE = (S * RB * NYC + 31) / 32;
    // which tiles a wave takes: every 4th tile of all four classes (all four waves then stream ALL weight fragments: 400 KB of
    // L2 reads per workgroup for layer 2, that kernel's bound), or -- WAVE_CLASS -- all tiles of ONE class (each fragment is read
    // by one wave; the classes have 9 / 6 / 6 / 4 taps, so the matrix work is less balanced, and the two column parities are
    // stored by different waves).  Measured per 256 samples: layer 2 185 us -> 105 us (137 us with one row parity per wave pair);
    // layer 1, bound by its 256 MB of output, 236 -> 271 us with WAVE_CLASS and 214 us with the paired stores.
    static constexpr bool WAVE_CLASS = L == 2;
    static constexpr int PT = WAVE_CLASS ? NTILE : (NTILE + 3) / 4;
    static constexpr int STEP_Q = CB * 2 * 64;                           // uint4 per (tap, 16 output channels): [cb][piece][lane]
    // the small layers have few (sample group, band) units: their workgroups also split over the parity classes and the blocks
    // of 32 input channels (grid.y), so that 64 samples still make > 100 workgroups
    static constexpr bool CLS_SPLIT = L >= 4;
    static constexpr int CBW = L == 5 ? 1 : CB, NSPLIT = (CLS_SPLIT ? 4 : 1) * (CB / CBW);
    static constexpr int Q00 = ((KS + 1) / 2) * ((KS + 1) / 2), Q01 = ((KS + 1) / 2) * (KS / 2), Q10 = Q01;   // taps per class
    static constexpr size_t WFRAG_Q = (size_t)KS * KS * NG * STEP_Q;
    static_assert(LDS_BYTES <= 66 * 1024 && COUT % 16 == 0, "tile budget");
};

// uint4 offset of a layer's fragments in the packed buffer
static inline size_t dgrad_frag_offset(int l) {
    size_t off = 0;
    for (int k = 1; k < l; ++k) off += (size_t)LKS[k] * LKS[k] * (LC_OUT[k] / 16) * ((LC_IN[k] + 31) / 32) * 128;
    return off;
}
static inline size_t dgrad_frag_total() { return dgrad_frag_offset(6); }

// fragments of ALL layers in one launch: [layer][class][a][b][co / 16][ci / 32][piece][lane] x 8 bf16 (k = co)
struct DgradPackArgs { const float* w[6]; };
static __global__ __launch_bounds__(256) void dgrad_pack_kernel(DgradPackArgs args, uint4* __restrict__ out, int total) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    int l = 1, base = 0, cin = 16, cout = 32, ks = 5;
    for (;; ++l) {
        switch (l) {                                     // (literals: the layer table is a host-side constexpr)
            case 1: cin = 16; cout = 32; ks = 5; break;
            case 2: cin = 32; cout = 64; ks = 5; break;
            case 3: cin = 64; cout = 64; ks = 3; break;
            case 4: cin = 64; cout = 128; ks = 3; break;
            default: cin = 128; cout = 128; ks = 3; break;
        }
        const int cnt = ks * ks * (cout / 16) * ((cin + 31) / 32) * 128;
        if (l == 5 || q < base + cnt) break;
        base += cnt;
    }
    const int ng = cout / 16, cbn = (cin + 31) / 32;
    int r = q - base;
    const int lane = r & 63; r >>= 6;
    const int piece = r & 1; r >>= 1;
    const int cb = r % cbn; r /= cbn;
    const int g = r % ng; r /= ng;                       // r = tap index over the classes in order
    int py = 0, px = 0, a = 0, b = 0;
    for (int cls = 0; cls < 4; ++cls) {
        py = cls >> 1; px = cls & 1;
        const int na = (ks - py + 1) / 2, nb = (ks - px + 1) / 2;
        if (r < na * nb) { a = r / nb; b = r - a * nb; break; }
        r -= na * nb;
    }
    const int ci = cb * 32 + (lane & 31), co0 = g * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        v[j] = ci < cin ? args.w[l][(((size_t)(co0 + j) * cin + ci) * ks + (py + 2 * a)) * ks + (px + 2 * b)] : 0.f;
    uint4 hi, lo;
    bf16_split8(v, hi, lo);
    out[q] = piece ? lo : hi;
}

// matrix steps of one parity class for the wave's tiles
template <int L, int PY, int PX>
__device__ __forceinline__ void dgrad_accum(f32x16 (&acc)[DgM<L>::PT][DgM<L>::CBW], const int (&ent0)[DgM<L>::PT],
                                            const unsigned char* __restrict__ s_dy, const uint4* __restrict__ wl, int tfirst,
                                            int tstride) {
    using T = DgM<L>;
    constexpr int NA = (T::KS - PY + 1) / 2, NB = (T::KS - PX + 1) / 2;
#pragma unroll
    for (int i = 0; i < T::PT; ++i)
#pragma unroll
        for (int c = 0; c < T::CBW; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    // one loop over (tap, 16 output channels), unrolled by four: the weight-fragment loads of four steps go out together (an L2
    // round trip per step, ~1 us under load, is what a rolled loop costs)
#pragma unroll 4
    for (int st = 0; st < NA * NB * T::NG; ++st) {
        const int tap = st / T::NG, g = st - tap * T::NG;
        const int a = tap / NB, b = tap - a * NB;
        const uint4* ws = wl + (size_t)st * T::STEP_Q;
        uint4 wa[T::CBW][2];
#pragma unroll
        for (int c = 0; c < T::CBW; ++c) {
            wa[c][0] = ws[(c * 2 + 0) * 64];
            wa[c][1] = ws[(c * 2 + 1) * 64];
        }
#pragma unroll
        for (int i = 0; i < T::PT; ++i) {
            if (tfirst + i * tstride < T::NTILE) {               // (scalar)
                const int ent = ent0[i] + 2 * g * T::ROWS * T::PW - (a * T::PW + b);
                const uint4 b0 = *reinterpret_cast<const uint4*>(s_dy + (size_t)ent * 16);
                const uint4 b1 = *reinterpret_cast<const uint4*>(s_dy + T::PIECE_B + (size_t)ent * 16);
#pragma unroll
                for (int c = 0; c < T::CBW; ++c) mfma3(acc[i][c], wa[c][0], wa[c][1], b0, b1);
            }
        }
    }
}

// both column parities (pxmask: bit 0 = even columns, bit 1 = odd) of input rows of parity PY, for tiles tfirst + i tstride
template <int L, int PY, int pxmask>
__device__ __forceinline__ void dgrad_rows(const unsigned char* __restrict__ s_dy, const uint4* __restrict__ wfrag, float* __restrict__ gin,
                                           int n0, int ns, int r0, int lane, int cb0, int tfirst, int tstride) {
    using T = DgM<L>;
    constexpr int NY = (T::IH - PY + 1) / 2, NX0 = (T::IH + 1) / 2, NX1 = T::IH / 2;
    constexpr int NPIX = T::S * T::RB * T::NYC;
    if (tfirst >= T::NTILE) return;                               // (scalar)
    const int h = lane >> 5, j = lane & 31;
    int ent0[T::PT], gofs[T::PT];
    bool valid0[T::PT], valid1[T::PT];
#pragma unroll
    for (int i = 0; i < T::PT; ++i) {
        const int p = (tfirst + i * tstride) * 32 + j;
        const int ss = p / (T::RB * T::NYC), rem = p - ss * (T::RB * T::NYC);
        const int ry = rem / T::NYC, ix2 = rem - ry * T::NYC;
        const bool v = p < NPIX && ss < ns && r0 + ry < NY;
        valid0[i] = v && ix2 < NX0 && (pxmask & 1);
        valid1[i] = v && ix2 < NX1 && (pxmask & 2);
        const int ss_c = v ? ss : 0, ry_c = v ? ry : 0, ix_c = v ? ix2 : 0;
        ent0[i] = ((ss_c * T::NOCT + h) * T::ROWS + (ry_c + T::PAD)) * T::PW + (ix_c + T::PAD);
        gofs[i] = (((n0 + ss_c) * T::CIN) * T::IH + 2 * (r0 + ry_c) + PY) * T::IH + 2 * ix_c;
    }
    const uint4* w0 = wfrag + (size_t)(PY ? T::Q00 + T::Q01 : 0) * T::NG * T::STEP_Q + cb0 * 128 + lane;
    const uint4* w1 = w0 + (size_t)(PY ? T::Q10 : T::Q00) * T::NG * T::STEP_Q;
    f32x16 acc0[(pxmask & 1) ? T::PT : 1][T::CBW], acc1[(pxmask & 2) ? T::PT : 1][T::CBW];
    if constexpr ((pxmask & 1) != 0) dgrad_accum<L, PY, 0>(acc0, ent0, s_dy, w0, tfirst, tstride);
    if constexpr ((pxmask & 2) != 0) dgrad_accum<L, PY, 1>(acc1, ent0, s_dy, w1, tfirst, tstride);
#pragma unroll
    for (int i = 0; i < T::PT; ++i) {
#pragma unroll
        for (int c = 0; c < T::CBW; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (cb0 + c) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (ci < T::CIN) {
                    float* o = gin + (size_t)gofs[i] + (size_t)ci * T::IH * T::IH;
                    if constexpr ((pxmask & 1) != 0) { if (valid0[i]) o[0] = acc0[i][c][r]; }
                    if constexpr ((pxmask & 2) != 0) { if (valid1[i]) o[1] = acc1[i][c][r]; }
                }
            }
    }
}

template <int L>
static __global__ __launch_bounds__(256) void dgrad_mfma_kernel(const float* __restrict__ dy, const uint4* __restrict__ wfrag,
                                                                 float* __restrict__ gin, int NS) {
    using T = DgM<L>;
    HIP_DYNAMIC_SHARED(unsigned char, s_dy)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // in a scalar register: "this wave has tile i" is a scalar branch
    const int ngroup = (NS + T::S - 1) / T::S;
    for (int unit = blockIdx.x; unit < ngroup * T::NBAND; unit += gridDim.x) {
        const int grp = unit / T::NBAND, band = unit - grp * T::NBAND;
        const int n0 = grp * T::S, ns = (NS - n0) < T::S ? (NS - n0) : T::S, r0 = band * T::RB;
        __syncthreads();                                         // the previous unit's readers are done
        // ---- dy tile: rows r0 - PAD .. r0 + RB - 1, all columns, zero outside the image ----
#pragma unroll 2
        for (int e = tid; e < T::ENTRIES; e += 256) {
            const int col = e % T::PW, t1 = e / T::PW;
            const int row = t1 % T::ROWS, t2 = t1 / T::ROWS;
            const int oct = t2 % T::NOCT, ss = t2 / T::NOCT;
            const int oy = r0 - T::PAD + row, ox = col - T::PAD;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = 0.f;
            if (ss < ns && oy >= 0 && oy < T::OH && ox >= 0 && ox < T::OH) {
                const float* src = dy + (((size_t)(n0 + ss) * T::COUT + oct * 8) * T::OH + oy) * T::OH + ox;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = src[(size_t)c * T::OH * T::OH];
            }
            uint4 hi, lo;
            bf16_split8(v, hi, lo);
            *reinterpret_cast<uint4*>(s_dy + (size_t)e * 16) = hi;
            *reinterpret_cast<uint4*>(s_dy + T::PIECE_B + (size_t)e * 16) = lo;
        }
        __syncthreads();
        const int cb0 = (int)(T::CLS_SPLIT ? blockIdx.y >> 2 : blockIdx.y) * T::CBW;
        if (T::CLS_SPLIT || T::WAVE_CLASS) {
            // one class per workgroup (small images: the workgroups of a unit split over the four classes) or per wave
            const int cls = T::CLS_SPLIT ? (int)(blockIdx.y & 3) : wave;
            const int tf = T::CLS_SPLIT ? wave : 0, ts = T::CLS_SPLIT ? 4 : 1;
            switch (cls) {                                       // (scalar)
                case 0: dgrad_rows<L, 0, 1>(s_dy, wfrag, gin, n0, ns, r0, lane, cb0, tf, ts); break;
                case 1: dgrad_rows<L, 0, 2>(s_dy, wfrag, gin, n0, ns, r0, lane, cb0, tf, ts); break;
                case 2: dgrad_rows<L, 1, 1>(s_dy, wfrag, gin, n0, ns, r0, lane, cb0, tf, ts); break;
                default: dgrad_rows<L, 1, 2>(s_dy, wfrag, gin, n0, ns, r0, lane, cb0, tf, ts); break;
            }
        } else {
            dgrad_rows<L, 0, 3>(s_dy, wfrag, gin, n0, ns, r0, lane, cb0, wave, 4);
            dgrad_rows<L, 1, 3>(s_dy, wfrag, gin, n0, ns, r0, lane, cb0, wave, 4);
        }
    }
}

template <int L>
static inline void launch_dgrad_mfma(const float* dy, const uint4* wfrag_all, float* gin, int NS, hipStream_t stream) {
    using T = DgM<L>;
    const int units = ((NS + T::S - 1) / T::S) * T::NBAND;
    int grid = units < 1024 ? units : 1024;
    if (grid < 1) grid = 1;
    static PerDeviceOnce once;
    const int dev_ = once.device();
    if (!once.is_done(dev_)) {
        hipFuncSetAttribute((const void*)dgrad_mfma_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        once.set_done(dev_);
    }
    hipLaunchKernelGGL(dgrad_mfma_kernel<L>, dim3(grid, T::NSPLIT), dim3(256), T::LDS_BYTES, stream, dy, wfrag_all + dgrad_frag_offset(L), gin, NS);
}

static inline void launch_dgrad_mfma_layer(int l, const float* dy, const uint4* wfrag_all, float* gin, int NS, hipStream_t stream) {
    switch (l) {
        case 1: launch_dgrad_mfma<1>(dy, wfrag_all, gin, NS, stream); break;
        case 2: launch_dgrad_mfma<2>(dy, wfrag_all, gin, NS, stream); break;
        case 3: launch_dgrad_mfma<3>(dy, wfrag_all, gin, NS, stream); break;
        case 4: launch_dgrad_mfma<4>(dy, wfrag_all, gin, NS, stream); break;
        default: launch_dgrad_mfma<5>(dy, wfrag_all, gin, NS, stream); break;
    }
}

}  // namespace cnnbwd

namespace cnnbwd {

// =============================================================================================
// Weight gradient of layer L (0..5):  dW[co][ci][ky][kx] += sum_{n,oy,ox} dy[n][co][oy][ox] a[n][ci][2 oy + ky][2 ox + kx].
// As a matrix product: rows = output channels (32), columns = (ci, ky) pairs of ONE kx (32 per block), k = 16 consecutive
// pixels of an output row.  The k values of a lane must be 8 CONSECUTIVE 2-byte elements at a 16-byte aligned LDS address for
// both operands; with a stride-2 window that is arranged by the substitution u = ox + (kx >> 1):
//     sum_ox dy[ox] a[2 ox + kx]  =  sum_u dy[u - (kx >> 1)] a[2 u + (kx & 1)]
// i.e. the input band is stored as two column-parity planes a_par[u] = a[2 u + par] (always read at aligned u), and the band of
// dy is stored (KS + 1) / 2 times, shifted by 0, 1, ... elements, so that dy[u - shift] is aligned too; u outside the row
// reads a zero of the dy copy.  A workgroup takes (S samples x RH output rows) units for its block of 32 output channels and
// CI_WG input channels (grid.y enumerates the blocks): dy band and input band -- GroupNorm + ReLU applied -- are split into
// bf16 pieces once on the way into LDS.  The (column block, kx) tiles x RS step subsets are dealt to the four waves; the
// partial sums of all units of the workgroup stay in the accumulators and are written ONCE, as they lie in the registers, to the
// workgroup's slot of a partial buffer; wgrad_reduce_kernel adds the slots in a fixed order into dW (deterministic, and
// measured: the first version's fp32 atomics -- 512 workgroups x 28 K values onto the 3 K addresses of layer 0 -- cost more
// than all the matrix work, 430 of 730 us).
// Pitches are padded to odd multiples of 16 bytes so that the 32 rows / columns of a fragment read hit distinct banks.
// =============================================================================================
template <int L>
struct WgM {
    static constexpr int CIN = LC_IN[L], COUT = LC_OUT[L], KS = LKS[L], IH = LIH[L], OH = LOH[L];
    static constexpr int NSH = (KS + 1) / 2;                             // shifted copies of dy
    static constexpr int UW = (OH + NSH - 1 + 15) / 16 * 16, NUG = UW / 16;
    static constexpr int CO_R = COUT < 32 ? COUT : 32, CO_SPLIT = (COUT + 31) / 32;
    static constexpr int CI_WG = L == 0 ? 4 : L <= 2 ? 16 : 32, CI_SPLIT = CIN / CI_WG;
    static constexpr int NSPLIT = CO_SPLIT * CI_SPLIT;
    static constexpr int NCOL = CI_WG * KS, NCOLB = (NCOL + 31) / 32, NT = NCOLB * KS;
    static constexpr int S = L >= 4 ? 2 : 1, RH = L <= 1 ? 1 : L == 2 ? 2 : L == 3 ? 4 : 2;
    static constexpr int NR = S * RH, IR = 2 * RH + KS - 2, NBAND = (OH + RH - 1) / RH;
    static constexpr int NSTEP = NR * NUG;
    static constexpr int RS = L <= 2 ? (L == 0 ? 2 : 1) : 2;            // step subsets per tile (items ~ multiple of 4, accumulators <= 80)
    static constexpr int ITEMS = NT * RS, IPW = (ITEMS + 3) / 4;
    static constexpr bool A_LO = L > 0;                                   // the raster bytes of layer 0 are exact in one bf16
    // byte pitches
    static constexpr int DY_CO = NR * UW * 2 + 16, DY_PIECE = CO_R * DY_CO, DY_SH = 2 * DY_PIECE, DY_BYTES = NSH * DY_SH;
    static constexpr int A_ROW = 4 * UW + 16;
    static constexpr int A_CI_RAW = S * IR * A_ROW;
    static constexpr int A_CI = A_CI_RAW + ((KS * 16 - A_CI_RAW) % 128 + 128) % 128;   // (ci, ky) -> consecutive 16-byte slots mod 128
    static constexpr int A_PIECE = CI_WG * A_CI, A_BYTES = (A_LO ? 2 : 1) * A_PIECE;
    static constexpr size_t LDS_BYTES = (size_t)DY_BYTES + A_BYTES;
    static_assert(LDS_BYTES <= 80 * 1024 && NSTEP % RS == 0 && CIN % CI_WG == 0 && UW % 16 == 0 && ITEMS <= 18, "tile budget");
};

template <int L>
static __global__ __launch_bounds__(256) void wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ act_in,
                                                                 const uint8_t* __restrict__ crop, const float2* __restrict__ mr_in,
                                                                 const float* __restrict__ gam_in, const float* __restrict__ bet_in,
                                                                 float* __restrict__ partial, int NS, int dbg) {
    using T = WgM<L>;
    HIP_DYNAMIC_SHARED(unsigned char, s_dy)           // [shift][piece][co][row][UW]
    unsigned char* s_a = s_dy + T::DY_BYTES;          // [piece][ci][sample row][parity][UW]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cob = (int)blockIdx.y % T::CO_SPLIT, cib = (int)blockIdx.y / T::CO_SPLIT;
    const int co_base = cob * 32, ci_base = cib * T::CI_WG;

    // ---- this wave's (tile, step subset) items ----
    int a_off[T::IPW], b_off[T::IPW], rs_[T::IPW];
    f32x16 acc[T::IPW];
#pragma unroll
    for (int s = 0; s < T::IPW; ++s) {
        const int item = wave + 4 * s;
        const int item_c = item < T::ITEMS ? item : wave;          // (a wave short of items repeats its first one; not stored)
        const int tile = item_c / T::RS;
        rs_[s] = item_c - tile * T::RS;
        const int colb = tile / T::KS, kx = tile - colb * T::KS;
        int q = colb * 32 + j;
        q = q < T::NCOL ? q : T::NCOL - 1;
        const int ci = q / T::KS, ky = q - ci * T::KS;
        a_off[s] = (kx >> 1) * T::DY_SH + (j % T::CO_R) * T::DY_CO + h * 16;
        b_off[s] = ci * T::A_CI + ky * T::A_ROW + (kx & 1) * (T::UW * 2) + h * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    }

    // GroupNorm scale / shift source of this workgroup's input channels
    __shared__ float s_gb[2][32];
    if (L > 0 && tid < T::CI_WG) { s_gb[0][tid] = gam_in[ci_base + tid]; s_gb[1][tid] = bet_in[ci_base + tid]; }

    // ---- staging is a two-stage pipeline: the raw values of the NEXT unit are requested (all loads of a unit together, into
    // registers) before the matrix steps of the current one, and converted / split / written to LDS when their turn comes.
    // (One load per loop trip -- the first version -- made a unit cost ~10 dependent memory round trips: 36 us per unit.)
    constexpr int DY_ITEMS = T::CO_R * T::NR * (T::UW / 2), DY_IT = (DY_ITEMS + 255) / 256;
    constexpr int NOCT = T::CI_WG / 8 > 0 ? T::CI_WG / 8 : 1;
    constexpr int A_ITEMS = L == 0 ? T::CI_WG * T::S * T::IR * (T::UW / 2) : NOCT * T::S * T::IR * T::UW, A_IT = (A_ITEMS + 255) / 256;
    float dyr[DY_IT][2];
    uint32_t a0r[L == 0 ? A_IT : 1];
    float4 ar[L == 0 ? 1 : A_IT][4];
    float2 amr[L == 0 ? 1 : A_IT];
    const int ngroup = (NS + T::S - 1) / T::S, nunit = ngroup * T::NBAND;

    auto load_unit = [&](int unit) {
        const int grp = unit / T::NBAND, band = unit - grp * T::NBAND;
        const int n0 = grp * T::S, oy0 = band * T::RH;
#pragma unroll
        for (int k = 0; k < DY_IT; ++k) {
            const int it = tid + 256 * k;
            const int e2 = it % (T::UW / 2), t1 = it / (T::UW / 2);
            const int row = t1 % T::NR, co = t1 / T::NR;
            const int ss = row / T::RH, oy = oy0 + (row - ss * T::RH);
            const bool rv = it < DY_ITEMS && n0 + ss < NS && oy < T::OH;
            const float* src = dy + (((size_t)(n0 + ss) * T::COUT + co_base + co) * T::OH + oy) * T::OH;
            dyr[k][0] = (rv && 2 * e2 < T::OH) ? src[2 * e2] : 0.f;
            dyr[k][1] = (rv && 2 * e2 + 1 < T::OH) ? src[2 * e2 + 1] : 0.f;
        }
        if (L == 0) {
#pragma unroll
            for (int k = 0; k < A_IT; ++k) {
                const int it = tid + 256 * k;
                const int u2 = it % (T::UW / 2), t1 = it / (T::UW / 2);
                const int srow = t1 % (T::S * T::IR), ci = t1 / (T::S * T::IR);
                const int ss = srow / T::IR, iy = 2 * oy0 + (srow - ss * T::IR);
                a0r[k] = 0u;
                if (it < A_ITEMS && n0 + ss < NS && iy < T::IH)
                    a0r[k] = *reinterpret_cast<const uint32_t*>(crop + (((size_t)(n0 + ss) * 4 + ci) * 256 + iy) * 256 + 4 * u2);
            }
        } else {
#pragma unroll
            for (int k = 0; k < A_IT; ++k) {
                const int it = tid + 256 * k;
                const int u2 = it % (T::UW / 2), t0 = it / (T::UW / 2);
                const int par = t0 & 1, t1 = t0 >> 1;
                const int srow = t1 % (T::S * T::IR), oct = t1 / (T::S * T::IR);
                const int ss = srow / T::IR, iy = 2 * oy0 + (srow - ss * T::IR);
                const int x0 = 4 * u2 + par, x1 = x0 + 2;
                const bool rv = it < A_ITEMS && n0 + ss < NS && iy < T::IH;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                ar[k][0] = z; ar[k][1] = z; ar[k][2] = z; ar[k][3] = z;
                amr[k] = make_float2(0.f, 0.f);                  // scale 0, shift 0: relu(0) = 0 outside the image
                if (rv) {
                    const float* src = act_in + ((((size_t)(n0 + ss) * (T::CIN / 8) + ((ci_base >> 3) + oct)) * T::IH + iy) * T::IH) * 8;
                    amr[k] = mr_in[n0 + ss];
                    if (x0 < T::IH) {
                        ar[k][0] = *reinterpret_cast<const float4*>(src + (size_t)x0 * 8);
                        ar[k][1] = *reinterpret_cast<const float4*>(src + (size_t)x0 * 8 + 4);
                    }
                    if (x1 < T::IH) {
                        ar[k][2] = *reinterpret_cast<const float4*>(src + (size_t)x1 * 8);
                        ar[k][3] = *reinterpret_cast<const float4*>(src + (size_t)x1 * 8 + 4);
                    }
                }
            }
        }
    };

    auto store_unit = [&](int unit) {
        const int grp = unit / T::NBAND, band = unit - grp * T::NBAND;
        const int n0 = grp * T::S, oy0 = band * T::RH;
        // ---- dy band, NSH shifted copies: element e of copy sh = dy[e - sh]; this thread owns ox = 2 e2, 2 e2 + 1 ----
#pragma unroll
        for (int k = 0; k < DY_IT; ++k) {
            const int it = tid + 256 * k;
            if (it >= DY_ITEMS) continue;
            const int e2 = it % (T::UW / 2), t1 = it / (T::UW / 2);
            const int row = t1 % T::NR, co = t1 / T::NR;
            uint32_t h0, l0, h1, l1;
            bf16_split(dyr[k][0], h0, l0);
            bf16_split(dyr[k][1], h1, l1);
            unsigned char* base = s_dy + co * T::DY_CO + row * (T::UW * 2);
#pragma unroll
            for (int sh = 0; sh < T::NSH; ++sh) {
                unsigned char* c0 = base + sh * T::DY_SH;
                const int e = 2 * e2 + sh;                       // element of ox = 2 e2
                if ((sh & 1) == 0) {
                    if (e + 1 < T::UW) {
                        *reinterpret_cast<uint32_t*>(c0 + e * 2) = h0 | (h1 << 16);
                        *reinterpret_cast<uint32_t*>(c0 + T::DY_PIECE + e * 2) = l0 | (l1 << 16);
                    }
                } else {
                    if (e < T::UW) {
                        *reinterpret_cast<uint16_t*>(c0 + e * 2) = (uint16_t)h0;
                        *reinterpret_cast<uint16_t*>(c0 + T::DY_PIECE + e * 2) = (uint16_t)l0;
                    }
                    if (e + 1 < T::UW) {
                        *reinterpret_cast<uint16_t*>(c0 + (e + 1) * 2) = (uint16_t)h1;
                        *reinterpret_cast<uint16_t*>(c0 + T::DY_PIECE + (e + 1) * 2) = (uint16_t)l1;
                    }
                }
                if (e2 == 0) {                                   // elements 0 .. sh-1 of the copy: ox < 0
                    for (int z = 0; z < sh; ++z) {
                        *reinterpret_cast<uint16_t*>(c0 + z * 2) = 0;
                        *reinterpret_cast<uint16_t*>(c0 + T::DY_PIECE + z * 2) = 0;
                    }
                }
            }
        }
        // ---- input band: two column-parity planes per (channel, row) ----
        if (L == 0) {
#pragma unroll
            for (int k = 0; k < A_IT; ++k) {
                const int it = tid + 256 * k;
                if (it >= A_ITEMS) continue;
                const int u2 = it % (T::UW / 2), t1 = it / (T::UW / 2);
                const int srow = t1 % (T::S * T::IR), ci = t1 / (T::S * T::IR);
                const uint32_t v = a0r[k];
                // pixels 4 u2 .. 4 u2 + 3: parity 0 holds (p0, p2) at u = 2 u2, 2 u2 + 1; parity 1 (p1, p3); a byte is exact in bf16
                const uint32_t p0 = bf16_bits((float)(v & 255u)), p1 = bf16_bits((float)((v >> 8) & 255u));
                const uint32_t p2 = bf16_bits((float)((v >> 16) & 255u)), p3 = bf16_bits((float)(v >> 24));
                unsigned char* dst = s_a + ci * T::A_CI + srow * T::A_ROW + u2 * 4;
                *reinterpret_cast<uint32_t*>(dst) = p0 | (p2 << 16);
                *reinterpret_cast<uint32_t*>(dst + T::UW * 2) = p1 | (p3 << 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < A_IT; ++k) {
                const int it = tid + 256 * k;
                if (it >= A_ITEMS) continue;
                const int u2 = it % (T::UW / 2), t0 = it / (T::UW / 2);
                const int par = t0 & 1, t1 = t0 >> 1;
                const int srow = t1 % (T::S * T::IR), oct = t1 / (T::S * T::IR);
                const float2 m = amr[k];
                const float ra[8] = {ar[k][0].x, ar[k][0].y, ar[k][0].z, ar[k][0].w, ar[k][1].x, ar[k][1].y, ar[k][1].z, ar[k][1].w};
                const float rb[8] = {ar[k][2].x, ar[k][2].y, ar[k][2].z, ar[k][2].w, ar[k][3].x, ar[k][3].y, ar[k][3].z, ar[k][3].w};
                const int x0 = 4 * u2 + par, x1 = x0 + 2;
                unsigned char* dst = s_a + (oct * 8) * T::A_CI + srow * T::A_ROW + par * (T::UW * 2) + u2 * 4;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float sc = m.y * s_gb[0][oct * 8 + c];
                    const float sf = m.y == 0.f ? 0.f : s_gb[1][oct * 8 + c] - m.x * sc;      // (rows outside the image: exact zeros)
                    const float va = x0 < T::IH ? fmaxf(fmaf(ra[c], sc, sf), 0.f) : 0.f;
                    const float vb = x1 < T::IH ? fmaxf(fmaf(rb[c], sc, sf), 0.f) : 0.f;
                    uint32_t ah, al, bh, bl;
                    bf16_split(va, ah, al);
                    bf16_split(vb, bh, bl);
                    *reinterpret_cast<uint32_t*>(dst + c * T::A_CI) = ah | (bh << 16);
                    *reinterpret_cast<uint32_t*>(dst + c * T::A_CI + T::A_PIECE) = al | (bl << 16);
                }
            }
        }
        (void)n0; (void)oy0;
    };

    // dbg (STRIVE_WGRAD_DBG, measurement only): 1 = no loads after the first unit, 2 = no LDS stores after the first unit,
    // 4 = no matrix steps
    if ((int)blockIdx.x < nunit) load_unit(blockIdx.x);
    for (int unit = blockIdx.x; unit < nunit; unit += gridDim.x) {
        __syncthreads();                                         // the previous unit's readers are done (and s_gb is there)
        if (!(dbg & 2) || unit == (int)blockIdx.x) store_unit(unit);
        __syncthreads();
        if (unit + (int)gridDim.x < nunit && !(dbg & 1)) load_unit(unit + gridDim.x);      // in flight during the matrix steps
        if (dbg & 4) continue;
        // ---- matrix steps: sigma = (row, 16-pixel group).  No branches in here (a wave short of items repeats its first one into
        // an accumulator nobody stores), so that the fragment reads of an item are scheduled under the matrix work of the others ----
#pragma unroll 2
        for (int sp = 0; sp < T::NSTEP / T::RS; ++sp) {
#pragma unroll
            for (int s = 0; s < T::IPW; ++s) {
                const int sigma = sp * T::RS + rs_[s];
                const int row = sigma / T::NUG, ug = sigma - row * T::NUG;
                const int ss = row / T::RH, oyl = row - ss * T::RH;
                const unsigned char* pa = s_dy + a_off[s] + row * (T::UW * 2) + ug * 32;
                const unsigned char* pb = s_a + b_off[s] + (ss * T::IR + 2 * oyl) * T::A_ROW + ug * 32;
                const uint4 a_hi = *reinterpret_cast<const uint4*>(pa), a_lo = *reinterpret_cast<const uint4*>(pa + T::DY_PIECE);
                const uint4 b_hi = *reinterpret_cast<const uint4*>(pb);
                if (T::A_LO) {
                    const uint4 b_lo = *reinterpret_cast<const uint4*>(pb + T::A_PIECE);
                    mfma3(acc[s], a_hi, a_lo, b_hi, b_lo);
                } else {
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_lo), as_bf16x8(b_hi), acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a_hi), as_bf16x8(b_hi), acc[s], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial sums of this workgroup: [workgroup][split][item][register][lane] ----
    float* pout = partial + ((size_t)blockIdx.x * T::NSPLIT + blockIdx.y) * T::ITEMS * 1024 + lane;
#pragma unroll
    for (int s = 0; s < T::IPW; ++s) {
        const int item = wave + 4 * s;
        if (item >= T::ITEMS) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) pout[(size_t)item * 1024 + r * 64] = acc[s][r];
    }
}

// dW[co][ci][ky][kx] += sum over workgroups and step subsets, in a fixed order.  block = one (split, tile, register) row of 64
// lanes x 16 phases: phase p adds workgroups p, p + 16, ... (eight loads in flight), the phases meet in LDS in index order.
template <int L>
static __global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int gx) {
    using T = WgM<L>;
    __shared__ float s_p[16][64];
    const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
    const int r = blockIdx.x & 15, t1 = blockIdx.x >> 4;
    const int tile = t1 % T::NT, split = t1 / T::NT;
    float sum = 0.f;
    const float* src = partial + ((size_t)split * T::ITEMS + (size_t)tile * T::RS) * 1024 + r * 64 + lane;
    const size_t wg_stride = (size_t)T::NSPLIT * T::ITEMS * 1024;
#pragma unroll 8
    for (int wg = phase; wg < gx; wg += 16) {
#pragma unroll
        for (int rs = 0; rs < T::RS; ++rs) sum += src[(size_t)wg * wg_stride + (size_t)rs * 1024];
    }
    s_p[phase][lane] = sum;
    __syncthreads();
    if (phase != 0) return;
#pragma unroll
    for (int p2 = 1; p2 < 16; ++p2) sum += s_p[p2][lane];
    const int colb = tile / T::KS, kx = tile - colb * T::KS;
    const int q = colb * 32 + (lane & 31);
    const int co = (split % T::CO_SPLIT) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (q >= T::NCOL || co >= T::COUT) return;
    const int ci = (split / T::CO_SPLIT) * T::CI_WG + q / T::KS, ky = q % T::KS;
    dW[(((size_t)co * T::CIN + ci) * T::KS + ky) * T::KS + kx] += sum;
}

template <int L>
static inline void launch_wgrad_mfma(const float* dy, const float* act_in, const uint8_t* crop, const float2* mr_in, const float* gam_in,
                                     const float* bet_in, float* dW, float* partial, int NS, hipStream_t stream) {
    using T = WgM<L>;
    const int units = ((NS + T::S - 1) / T::S) * T::NBAND;
    int gx = 512 / T::NSPLIT;                 // ~2 workgroups per CU in total
    if (gx > units) gx = units;
    if (gx < 1) gx = 1;
    static PerDeviceOnce once;
    const int dev_ = once.device();
    if (!once.is_done(dev_)) {
        hipFuncSetAttribute((const void*)wgrad_mfma_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        once.set_done(dev_);
    }
    const int dbg = strive_tuning().wgrad_dbg;
    hipLaunchKernelGGL(wgrad_mfma_kernel<L>, dim3(gx, T::NSPLIT), dim3(256), T::LDS_BYTES, stream, dy, act_in, crop, mr_in, gam_in,
                       bet_in, partial, NS, dbg);
    hipLaunchKernelGGL(wgrad_reduce_kernel<L>, dim3(T::NSPLIT * T::NT * 16), dim3(1024), 0, stream, partial, dW, gx);
}

// floats of the partial buffer: <= 512 workgroups x <= 18 items x 1024
static inline size_t wgrad_partial_floats() { return (size_t)512 * 18 * 1024; }

static inline void launch_wgrad_mfma_layer(int l, const float* dy, const float* act_in, const uint8_t* crop, const float2* mr_in,
                                           const float* gam_in, const float* bet_in, float* dW, float* partial, int NS, hipStream_t stream) {
    switch (l) {
        case 0: launch_wgrad_mfma<0>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, partial, NS, stream); break;
        case 1: launch_wgrad_mfma<1>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, partial, NS, stream); break;
        case 2: launch_wgrad_mfma<2>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, partial, NS, stream); break;
        case 3: launch_wgrad_mfma<3>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, partial, NS, stream); break;
        case 4: launch_wgrad_mfma<4>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, partial, NS, stream); break;
        default: launch_wgrad_mfma<5>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, partial, NS, stream); break;
    }
}

}  // namespace cnnbwd
