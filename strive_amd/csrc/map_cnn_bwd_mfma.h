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
    static constexpr int NTILE = (S * RB * NYC + 31) / 32, PT = (NTILE + 3) / 4;
    static constexpr int STEP_Q = CB * 2 * 64;                           // uint4 per (tap, 16 output channels): [cb][piece][lane]
    // the small layers have few (sample group, band) units: their workgroups also split over the parity classes and the blocks
    // of 32 input channels (grid.y), so that 64 samples still make > 100 workgroups
    static constexpr bool CLS_SPLIT = L >= 4;
    static constexpr int CBW = L == 5 ? 1 : CB, NSPLIT = (CLS_SPLIT ? 4 : 1) * (CB / CBW);
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

template <int L, int PY, int PX>
__device__ __forceinline__ void dgrad_class(const unsigned char* __restrict__ s_dy, const uint4* __restrict__ wf, float* __restrict__ gin,
                                            int n0, int ns, int r0, int lane, int wave, int cb0) {
    using T = DgM<L>;
    constexpr int NY = (T::IH - PY + 1) / 2, NX = (T::IH - PX + 1) / 2, NA = (T::KS - PY + 1) / 2, NB = (T::KS - PX + 1) / 2;
    constexpr int NPIX = T::S * T::RB * NX, NTILE = (NPIX + 31) / 32;
    const int h = lane >> 5, j = lane & 31;
    // this lane's pixel in each of the wave's tiles
    int ent0[T::PT], gofs[T::PT];
    bool valid[T::PT];
#pragma unroll
    for (int i = 0; i < T::PT; ++i) {
        const int p = (wave + 4 * i) * 32 + j;
        const int ss = p / (T::RB * NX), rem = p - ss * (T::RB * NX);
        const int ry = rem / NX, ix2 = rem - ry * NX;
        valid[i] = p < NPIX && ss < ns && r0 + ry < NY;
        const int ss_c = valid[i] ? ss : 0, ry_c = valid[i] ? ry : 0, ix_c = valid[i] ? ix2 : 0;
        ent0[i] = ((ss_c * T::NOCT + h) * T::ROWS + (ry_c + T::PAD)) * T::PW + (ix_c + T::PAD);
        gofs[i] = (((n0 + ss_c) * T::CIN) * T::IH + 2 * (r0 + ry_c) + PY) * T::IH + 2 * ix_c + PX;
    }
    if (wave >= NTILE) return;                                   // (wave-uniform: nothing to do for this class)
    f32x16 acc[T::PT][T::CBW];
#pragma unroll
    for (int i = 0; i < T::PT; ++i)
#pragma unroll
        for (int c = 0; c < T::CBW; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    const uint4* wl = wf + cb0 * 128 + lane;
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
            if (wave + 4 * i < NTILE) {
                const int ent = ent0[i] + 2 * g * T::ROWS * T::PW - (a * T::PW + b);
                const uint4 b0 = *reinterpret_cast<const uint4*>(s_dy + (size_t)ent * 16);
                const uint4 b1 = *reinterpret_cast<const uint4*>(s_dy + T::PIECE_B + (size_t)ent * 16);
#pragma unroll
                for (int c = 0; c < T::CBW; ++c) mfma3(acc[i][c], wa[c][0], wa[c][1], b0, b1);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < T::PT; ++i) {
        if (!valid[i]) continue;
#pragma unroll
        for (int c = 0; c < T::CBW; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = (cb0 + c) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (ci < T::CIN) gin[(size_t)gofs[i] + (size_t)ci * T::IH * T::IH] = acc[i][c][r];
            }
    }
}

template <int L>
static __global__ __launch_bounds__(256) void dgrad_mfma_kernel(const float* __restrict__ dy, const uint4* __restrict__ wfrag,
                                                                 float* __restrict__ gin, int NS) {
    using T = DgM<L>;
    HIP_DYNAMIC_SHARED(unsigned char, s_dy)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        constexpr int Q00 = ((T::KS + 1) / 2) * ((T::KS + 1) / 2), Q01 = ((T::KS + 1) / 2) * (T::KS / 2), Q10 = Q01;
        const int cls_sel = T::CLS_SPLIT ? (int)(blockIdx.y & 3) : -1;
        const int cb0 = (int)(T::CLS_SPLIT ? blockIdx.y >> 2 : blockIdx.y) * T::CBW;
        const uint4* wf = wfrag;
        if (cls_sel < 0 || cls_sel == 0) dgrad_class<L, 0, 0>(s_dy, wf, gin, n0, ns, r0, lane, wave, cb0);
        wf += (size_t)Q00 * T::NG * T::STEP_Q;
        if (cls_sel < 0 || cls_sel == 1) dgrad_class<L, 0, 1>(s_dy, wf, gin, n0, ns, r0, lane, wave, cb0);
        wf += (size_t)Q01 * T::NG * T::STEP_Q;
        if (cls_sel < 0 || cls_sel == 2) dgrad_class<L, 1, 0>(s_dy, wf, gin, n0, ns, r0, lane, wave, cb0);
        wf += (size_t)Q10 * T::NG * T::STEP_Q;
        if (cls_sel < 0 || cls_sel == 3) dgrad_class<L, 1, 1>(s_dy, wf, gin, n0, ns, r0, lane, wave, cb0);
    }
}

template <int L>
static inline void launch_dgrad_mfma(const float* dy, const uint4* wfrag_all, float* gin, int NS, hipStream_t stream) {
    using T = DgM<L>;
    const int units = ((NS + T::S - 1) / T::S) * T::NBAND;
    int grid = units < 1024 ? units : 1024;
    if (grid < 1) grid = 1;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)dgrad_mfma_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        attr_set = true;
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
