// Map CNN forward: 6 x [Conv2d(stride 2, pad 0) -> GroupNorm(1 group) -> ReLU] -> Linear(512,64)
// (reference src/models/traffic_model.py:69-87, 437-440), with the raster crop fused into the
// first convolution so the (N,4,256,256) crop never exists in HBM.
//
// Every convolution is an implicit GEMM on the fp16 matrix cores at fp32 accuracy,
//     D[co][pixel] += W[co][k] * patch[k][pixel]
// with A = weights (M = output channels) and B = input patches (N = output pixels).  An fp32 operand x is carried as TWO fp16
// pieces, x0 = fp16(x) rounded to nearest and x1 = fp16(x - x0): because x0 is rounded to nearest the remainder needs 12 bits,
// so x0 + x1 = x up to 2^-24 |x|, and x w = x0 w0 + x0 w1 + x1 w0 (the dropped x1 w1 is <= 2^-24 |x w|) -- three products, each
// exact in fp32 (11 x 11 significand bits), fp32 accumulation inside the MFMA.  fp16's narrow exponent range is met with exact
// power-of-two scales (weights: wscale[l] on the host; activations: xscale[l] folded into the GroupNorm affine map), undone in
// the epilogue.
//   layer 1 (u8 crop, exact in fp16; Cout 16):  v_mfma_f32_16x16x32_f16, weights in 2 pieces = 2 products
//   layers 2-4 (fp32 in):                       v_mfma_f32_32x32x16_f16, both operands in 2 pieces = 3 products
//   layers 5-6 + Linear:                        one fused kernel (map_cnn_tail.h), same scheme
// Input tiles are staged once per workgroup into LDS with the previous layer's GroupNorm + ReLU applied
// on the way in (so normalised activations never exist in HBM either); columns are stored
// de-interleaved by parity so the stride-2 window reads of consecutive output pixels hit consecutive
// banks.  GroupNorm statistics (sum, sum of squares per sample, float64) are accumulated in the
// epilogue of the producing convolution, per tile, and added in tile order by the consumer.
#include "common.h"
#include "crop_dev.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GN_EPS 1e-5

struct GNStats { double sum, sq; };

// Per-sample GroupNorm(1) moments from the producing layer's per-tile partial sums, added in tile order
// (no atomics anywhere: the CNN is bitwise reproducible run to run, which matters because the rollout
// re-samples the raster at poses that depend on these features).
__device__ __forceinline__ void gn_moments(const GNStats* __restrict__ st, int n, int nparts, double count, float& mean,
                                           float& rstd) {
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nparts; ++i) {
        s += st[(size_t)n * nparts + i].sum;
        q += st[(size_t)n * nparts + i].sq;
    }
    const double m = s / count;
    double var = q / count - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + GN_EPS));
}

// =============================================================================================
// Layer 1 on the fp16 matrix cores at fp32 accuracy.
// The crop is uint8 (exactly representable in fp16) and every fp32 weight, scaled by the power of two wscale[0], is carried as
// two fp16 pieces (w = w0 + w1 up to 2^-24 |w|, w0 rounded to nearest), so  sum_k in_k * w_k  =  sum_k in_k*w0_k + in_k*w1_k
// with every product exact in fp32 and fp32 accumulation inside v_mfma_f32_16x16x32_f16: an fp32 dot product in a different
// summation order, at the fp16 matrix rate: 14 MFMAs per 16-pixel tile (7 window rows x 2 weight pieces).
// k is ordered (ky, column pair g = 0..3, column parity, channel): lane group g of an MFMA holds the 8 values
// of window columns 2g, 2g+1 (x 4 layers) which are 16 contiguous bytes of the [row][col][layer] bf16 LDS
// tile; the 8th window column is padding with zero weights.
// Workgroup = 4 waves, 32 (x) x 16 (y) output pixels = 32 pixel tiles of 16, 8 per wave.
// =============================================================================================
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

namespace l1b {
constexpr int CIN = 4, COUT = 16, KS = 7, IH = 256, OH = 125;
constexpr int TOX = 32, TOY = 16;
constexpr int ITW = 2 * TOX + KS - 2 + 1;   // 70 columns (69 used + the zero-weight pad column of the last pair)
constexpr int ITH = 2 * TOY + KS - 2;       // 37 rows
constexpr int TILES_X = (OH + TOX - 1) / TOX, TILES_Y = (OH + TOY - 1) / TOY;   // 4 x 8
constexpr int NPIX = ITH * ITW;             // 2590 staged pixels (8 bytes each)
constexpr int WFRAGS = KS * 2;              // (ky, piece) fragments, 64 lanes x 16 B each
constexpr int WBYTES = WFRAGS * 64 * 16;    // 14336
constexpr int NPART = TILES_X * TILES_Y;    // 32 statistics slots per sample
}  // namespace l1b

__device__ __forceinline__ uint32_t f16_bits(float v) {
    const _Float16 h = (_Float16)v;             // round to nearest even
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return (uint32_t)b;
}

__device__ __forceinline__ uint32_t pack_f16_pair_from_bytes(uint32_t word, int lo_byte) {
    // two layers (bytes lo_byte, lo_byte+1 of `word`) -> two fp16 in one dword; integers <= 255 are exact in fp16
    const float f0 = (float)((word >> (8 * lo_byte)) & 0xffu);
    const float f1 = (float)((word >> (8 * lo_byte + 8)) & 0xffu);
    return f16_bits(f0) | (f16_bits(f1) << 16);
}

// Persistent over one row of tiles (TILES_X = 4 tiles of 32 x 16 outputs) with two LDS input buffers: while the
// matrix cores work on tile t, the same waves have already derived the raster pixels of tile t+1 and have its
// gathers in flight; the gathered words are converted and written to the other buffer after the MFMA phase.
// The weight fragments are loaded once per workgroup.  8 waves: wave w owns output rows 2w, 2w+1 (4 pixel tiles).
constexpr int C1_NT = 512;

// Round 6: no global load is consumed inside the tile loop except the raster gathers.  The bias and the crop's lwise / wwise tables
// come from LDS: hipcc's wait-count pass cannot count the predicated gathers, so the wait it placed in front of every use of a
// globally loaded value in the loop was s_waitcnt vmcnt(0) -- in front of each of the four output stores of a tile (the bias), i.e.
// every store waited for the previous one to be acknowledged and for the next tile's gathers, which were meant to stay in flight
// until deposit().  The GroupNorm sums of a lane's 16 outputs of a tile are formed in fp32 (conv_bf6_kernel's rule), then float64:
// 2 conversions + 2 additions per 16 values instead of 16 + 32 (fp64 runs at half rate).
template <bool FUSED_CROP, int DBG = 0>
__global__ __launch_bounds__(C1_NT, 4) void conv1b_kernel(StriveMap map, const float* __restrict__ pos, Float4Host pmean,
                                                         Float4Host pstd, const int32_t* __restrict__ mapix,
                                                         const uint8_t* __restrict__ crop, const uint32_t* __restrict__ wfrag,
                                                         float unscale, const float* __restrict__ bias, float* __restrict__ out,
                                                         GNStats* __restrict__ stats) {
    using namespace l1b;
    __shared__ __attribute__((aligned(16))) uint32_t s_in[2][NPIX * 2];     // [buffer][row][col][4 x fp16]
    __shared__ __attribute__((aligned(16))) uint32_t s_w[WBYTES / 4];
    __shared__ double s_red[2][16];
    __shared__ __attribute__((aligned(16))) float s_bias[COUT];
    __shared__ float s_lw[2 * IH];                                    // lwise | wwise
    const int n = blockIdx.z;
    const int ty = blockIdx.x;
    const int oy0 = ty * TOY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;       // k lane group = window column pair
    const int j = lane & 15;       // pixel within the tile (B) / output channel (A)

    for (int i = tid; i < WBYTES / 16; i += C1_NT)
        reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(wfrag)[i];
    if (tid < COUT) s_bias[tid] = bias[tid];
    if (FUSED_CROP && tid < 2 * IH) s_lw[tid] = tid < IH ? map.lwise[tid] : map.wwise[tid - IH];

    CropFrame fr;
    const uint32_t* pk = nullptr;
    if (FUSED_CROP) {
        fr = load_crop_frame(map, pos, pmean.v, pstd.v, mapix, n);
        pk = map.raster_px4 + (size_t)mapix[n] * map.H * map.W;
    }
    // Lanes of a wave cover an 8 x 8 patch of crop pixels (not 64 pixels of one row): a patch spans ~10 x 10
    // raster pixels whatever the heading, i.e. ~10 cache lines per gather instruction instead of up to 64.
    constexpr int NBX = (ITW + 7) / 8, NBY = (ITH + 7) / 8;       // 9 x 5 patches
    constexpr int NSLOT = NBX * NBY * 64;
    constexpr int NIT = (NSLOT + C1_NT - 1) / C1_NT;              // 6 samples per thread per tile
    uint32_t word[NIT];
    // per-tile tables of the rounded products l*cos, l*sin (rows) and w*cos, w*sin (columns): the reference's
    // fl(fl(l*c) - fl(w*s)) + x needs each product once per row / column, not once per pixel
    __shared__ float s_tab[2][2 * ITH + 2 * ITW];

    auto tables = [&](int tx, int tb) {
        const int ox0 = tx * TOX;
        float* T = s_tab[tb];
        for (int i = tid; i < 2 * ITH + 2 * ITW; i += C1_NT) {
            float v = 0.f;
            if (i < 2 * ITH) {
                const int r = i >> 1, l = 2 * oy0 + r;
                if (l < IH) v = __fmul_rn(s_lw[l], (i & 1) ? fr.hs : fr.hc);
            } else {
                const int c = (i - 2 * ITH) >> 1, w = 2 * ox0 + c;
                if (w < IH) v = __fmul_rn(s_lw[IH + w], (i & 1) ? fr.hs : fr.hc);
            }
            T[i] = v;
        }
    };
    auto slot_rc = [&](int k, int& r, int& c) {
        const int slot = tid + k * C1_NT;
        const int blk = slot >> 6, li = slot & 63;
        r = (blk / NBX) * 8 + (li >> 3);
        c = (blk % NBX) * 8 + (li & 7);
        return slot < NSLOT && r < ITH && c < ITW;
    };
    auto gather = [&](int tx, int tb) {      // derive pixel indices of tile tx and issue the gathers
        const int ox0 = tx * TOX;
        const float* T = s_tab[tb];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            int r, c;
            word[k] = 0u;
            if (slot_rc(k, r, c)) {
                const int l = 2 * oy0 + r, w = 2 * ox0 + c;
                if (l < IH && w < IH) {
                    if (FUSED_CROP) {
                        float gx = __fadd_rn(__fsub_rn(T[2 * r], T[2 * ITH + 2 * c + 1]), fr.x);
                        float gy = __fadd_rn(__fadd_rn(T[2 * r + 1], T[2 * ITH + 2 * c]), fr.y);
                        gx = (gx != gx) ? 0.0f : gx;
                        gy = (gy != gy) ? 0.0f : gy;
                        int px, py;
                        if (DBG == 2) { px = w + (int)gx % 3; py = l; }                 // timing probe: no fp64 pixel math
                        else world_to_pixel(fr, gx, gy, px, py);
                        if (DBG == 1) word[k] = (uint32_t)(px ^ py) & 0x01010101u;      // timing probe: no gather
                        else word[k] = pk[(size_t)py * map.W + px];
                    } else {
                        const uint8_t* src = crop + (size_t)n * CIN * IH * IH + (size_t)l * IH + w;
                        word[k] = (uint32_t)src[0] | ((uint32_t)src[(size_t)IH * IH] << 8) |
                                  ((uint32_t)src[(size_t)2 * IH * IH] << 16) | ((uint32_t)src[(size_t)3 * IH * IH] << 24);
                    }
                }
            }
        }
    };
    auto deposit = [&](int buf) {    // gathered words -> 4 x fp16 -> LDS
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            int r, c;
            if (slot_rc(k, r, c)) {
                uint2 v;
                v.x = pack_f16_pair_from_bytes(word[k], 0);
                v.y = pack_f16_pair_from_bytes(word[k], 2);
                *reinterpret_cast<uint2*>(&s_in[buf][(r * ITW + c) * 2]) = v;
            }
        }
    };

    // gridDim.y workgroups share a row of tiles: each walks TILES_X / gridDim.y of them.  One (the whole row, pipelined) when the
    // batch fills the chip; four (one tile each) for small batches, where the row's 4-tile chain is what a launch waits for
    // (8 samples: 17.7 -> ~8 us, profiles/r04_ab_cnn_small_batch.txt).  Same tiles, same arithmetic, same statistics slots.
    const int tpb = TILES_X / (int)gridDim.y;
    const int tx_begin = (int)blockIdx.y * tpb, tx_end = tx_begin + tpb;
    __syncthreads();                                                 // s_lw, s_bias
    if (FUSED_CROP) tables(tx_begin, tx_begin & 1);
    __syncthreads();
    gather(tx_begin, tx_begin & 1);
    if (FUSED_CROP && tx_begin + 1 < tx_end) tables(tx_begin + 1, (tx_begin + 1) & 1);
    deposit(tx_begin & 1);
    __syncthreads();
    const f16x8* wl = reinterpret_cast<const f16x8*>(s_w) + lane;
    for (int tx = tx_begin; tx < tx_end; ++tx) {
        const int buf = tx & 1;
        if (tx + 1 < tx_end) gather(tx + 1, (tx + 1) & 1);
        // statistics of the previous tile (its per-wave partials were published before the last barrier)
        if (tx > tx_begin && tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < 8; ++w) { a += s_red[(tx - 1) & 1][2 * w]; b += s_red[(tx - 1) & 1][2 * w + 1]; }
            GNStats& o = stats[(size_t)n * NPART + ty * TILES_X + (tx - 1)];
            o.sum = a;
            o.sq = b;
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            f16x8 bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int oyl = wave * 2 + (i >> 1), oxl = (i & 1) * 16 + j;
                const int pix = (2 * oyl + ky) * ITW + 2 * oxl + 2 * g;
                bfr[i] = *reinterpret_cast<const f16x8*>(&s_in[buf][pix * 2]);
            }
#pragma unroll
            for (int term = (DBG == 3 ? 0 : 1); term >= 0; --term) {      // the small piece first
                const f16x8 a = wl[(ky * 2 + term) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bfr[i], acc[i], 0, 0, 0);
            }
        }
        // epilogue: D column = lane&15 = pixel, row = (lane>>4)*4 + r = output channel.  Output layout = channel
        // octets as planes, [n][c/8][y][x][c%8] ("octet-planar"): the consumer stages 8 channels per pass and reads
        // 32 contiguous bytes per pixel; one 16-byte store per lane and pixel tile
        double lsum = 0.0, lsq = 0.0;
        float fsum = 0.f, fsq = 0.f;
        const int ox0 = tx * TOX;
        const float4 bv = *reinterpret_cast<const float4*>(s_bias + g * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = oy0 + wave * 2 + (i >> 1), ox = ox0 + (i & 1) * 16 + j;
            float4 v;
            v.x = fmaf(acc[i][0], unscale, bv.x);      // unscale = 2^-k exactly: one rounding, like acc + bias
            v.y = fmaf(acc[i][1], unscale, bv.y);
            v.z = fmaf(acc[i][2], unscale, bv.z);
            v.w = fmaf(acc[i][3], unscale, bv.w);
            if (oy < OH && ox < OH) {
                if (DBG != 4)
                    *reinterpret_cast<float4*>(out + ((((size_t)n * (COUT / 8) + (g >> 1)) * OH + oy) * OH + ox) * 8 + (g & 1) * 4) = v;
                fsum += (v.x + v.y) + (v.z + v.w);
                fsq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fsq))));
            }
        }
        lsum = (double)fsum;
        lsq = (double)fsq;
        lsum = wave_sum_d(lsum);
        lsq = wave_sum_d(lsq);
        if (lane == 0) { s_red[buf][2 * wave] = lsum; s_red[buf][2 * wave + 1] = lsq; }
        if (tx + 1 < tx_end) deposit(buf ^ 1);
        if (FUSED_CROP && tx + 2 < tx_end) tables(tx + 2, tx & 1);   // slot tx&1 was last read by gather(tx)
        __syncthreads();
    }
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 8; ++w) { a += s_red[(tx_end - 1) & 1][2 * w]; b += s_red[(tx_end - 1) & 1][2 * w + 1]; }
        GNStats& o = stats[(size_t)n * NPART + ty * TILES_X + (tx_end - 1)];
        o.sum = a;
        o.sq = b;
    }
}

// =============================================================================================
// Layers 2-4 on the bf16 matrix cores at fp32 accuracy ("bf16 x 6").
// Both operands are split EXACTLY into three bf16 pieces (x = x0 + x1 + x2, 8 mantissa bits each; the weights on
// the host, the activations while they are staged into LDS, by truncation: x0 = x & 0xffff0000, x1 likewise of
// x - x0, x2 = the rest) and the six products down to 2^-16 of the leading one are accumulated in fp32 inside
// v_mfma_f32_32x32x16_bf16:  a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0.  The dropped terms are <= 2^-23 of
// a product, i.e. the size of one fp32 rounding of that product: the result is an fp32 dot product in another
// summation order (measured: 1.6x torch-CPU-fp32's own error against float64, tools/cnn_accuracy.py).
// The 32x32x16 shape is the one that reaches the bf16 peak on gfx950 (measured 2.29 PFLOP/s; the 16x16x32 shape
// issues at the same 14 ns for half the work: tools/mfma_rate_probe.hip), i.e. 2.4x the fp32 matrix rate after
// the factor 6.
// Activations are fp32 in HBM in octet-planar order [n][c/8][y][x][c%8] (a pass reads 32 contiguous bytes per pixel);
// the previous layer's GroupNorm + ReLU is applied on the way into LDS.
// LDS input layout: [piece 3][row][column parity][column/2][8 x bf16]: one 16-byte slot per pixel (8 input channels
// per pass), consecutive output pixels (stride 2 in the input) in consecutive slots.
// k order: an MFMA step covers two window taps x 8 channels; lane half h = lane / 32 holds tap h of the step.
//   steps 0-9:  row ky = s / 2, columns kx = (s & 1) + 2h   (same row, same parity: the two 512-byte windows overlap)
//   steps 10-11: column 4 of rows 2 (s - 10) + h;   step 12: tap (4, 4) and one zero-weight slot.
// Workgroup: 4 waves, 8 x 32 output pixels x 32 output channels; wave w owns output rows 2w, 2w+1 (two pixel tiles of
// 32) : per MFMA step it reads 3 + 6 fragments from LDS for 12 MFMAs.  Input channels are staged 8 at a time
// (CIN / 8 passes, next pass prefetched into registers); the weight fragments of a pass (13 x 3 KB) are requested up
// front, parked in registers and fed through three rotating LDS buffers, and the fragment reads of step s+1 are
// interleaved with the matrix work of step s.
// =============================================================================================
template <int CIN_, int COUT_, int KS_, int IH_, int OH_, int NPART_IN_, bool OUT_OCT_, int PT_ = 2, int WGS_PER_CU_ = 2,
          bool ROWS2_ = false, int CBW_ = 1>
struct BfCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, KS = KS_, IH = IH_, OH = OH_, NPART_IN = NPART_IN_;
    static constexpr bool OUT_OCT = OUT_OCT_;
    static constexpr int PT = PT_;                                  // pixel tiles of 32 per wave
    static constexpr bool ROWS2 = ROWS2_;                           // pixel tile = 2 rows x 16 columns (small images) instead of 1 x 32
    static constexpr int TILE_ROWS = ROWS2 ? 2 : 1;
    static constexpr int WGS_PER_CU = WGS_PER_CU_;                  // residency target (LDS and register budget)
    static constexpr int NT = 256, NW = 4, TH = NW * PT * TILE_ROWS, TW = ROWS2 ? 16 : 32;
    // a workgroup computes CBW blocks of 32 output channels from ONE staging of the input tile (the matrix steps of a pass
    // run once per block): the input is fetched, normalised and split COUT / (32 CBW) times instead of COUT / 32 times
    static constexpr int CBW = CBW_, COUT_WG = 32 * CBW, CSPLIT = COUT / COUT_WG;
    static constexpr int PASS_CH = 8, NPASS = CIN / PASS_CH;
    static constexpr int ITH = 2 * TH + KS - 2, ITW = 2 * TW + KS - 2, HW = (ITW + 1) / 2;
    static constexpr int HALF_B = HW * 16, ROW_B = 2 * HALF_B, PIECE_B = ITH * ROW_B;
    static constexpr int NPIECE = 2;                                          // fp16 pieces per value
    static constexpr int IN_B = (NPIECE * PIECE_B + 255) / 256 * 256;         // weight fragments start 256-byte aligned
    static constexpr int NKS = (KS * KS + 1) / 2;                            // MFMA steps per pass (two taps each)
    static constexpr int WSTEP_B = CBW * 2 * 64 * 16;                        // one matrix step: [block][piece][lane][16 B]
    static constexpr int TILES_X = (OH + TW - 1) / TW, TILES_Y = (OH + TH - 1) / TH;
    static constexpr int NPART_OUT = TILES_X * TILES_Y * CSPLIT;
    static constexpr int UNITS = ITH * ITW, UITERS = (UNITS + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = (size_t)IN_B + 3 * WSTEP_B + (size_t)CIN * 8 + NW * 16 + 16 + COUT_WG * 4;
    static constexpr size_t WFRAG_BYTES = (size_t)NPASS * NKS * (COUT / 32) * 2048;
    static_assert(CIN % PASS_CH == 0 && COUT % COUT_WG == 0 && CIN <= NT, "channel tiling");
    static_assert(LDS_BYTES * WGS_PER_CU <= 160 * 1024, "LDS budget of the residency target");
    static_assert((KS == 5 && NKS == 13) || (KS == 3 && NKS == 5), "tap orders exist for 5x5 and 3x3 windows");
    static_assert(WSTEP_B == 16 * 128 * CBW && WSTEP_B / 16 <= NT, "weight step = one 16-byte piece for each of the first 128 CBW threads");
    // weight steps of a pass requested before its staging; the rest is requested at matrix step W_LATE_AT, when the registers
    // of the first steps have been handed to LDS.  Three workgroups per CU leave 168 registers per lane: with all 13 steps
    // parked (52 registers) conv2 spilled 11 of them to scratch -- 92 MB written and 92 MB read back per 512-agent launch
    // (rocprofv3 WRITE_SIZE / FETCH_SIZE), a sixth of the kernel's HBM traffic.
    static constexpr int W_UPFRONT = (WGS_PER_CU >= 3 && NKS > 8) ? 7 : NKS, W_LATE_AT = 2;
};

// v = p0 + p1 up to 2^-24 |v| (p0 = fp16(v) rounded to nearest, p1 = fp16(v - p0)); v is pre-scaled into fp16's range.
// Two values per conversion (v_cvt_pk_f16_f32 on gfx950, round to nearest even like the scalar form: same bits, 6 instead of 8
// instructions per pair); the subtraction stays scalar (no packed fp32 arithmetic: DESIGN.md 8.1).
typedef float split_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 split_f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_f16x2(const float v[8], uint4& p0, uint4& p1) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const split_f32x2 v2 = {v[2 * i], v[2 * i + 1]};
        const split_f16x2v a = __builtin_convertvector(v2, split_f16x2v);
        const float r0 = v[2 * i] - (float)a[0];            // exact
        const float r1 = v[2 * i + 1] - (float)a[1];
        const split_f32x2 r2 = {r0, r1};
        const split_f16x2v c = __builtin_convertvector(r2, split_f16x2v);
        __builtin_memcpy(&h[i], &a, 4);
        __builtin_memcpy(&l[i], &c, 4);
    }
    p0 = make_uint4(h[0], h[1], h[2], h[3]);
    p1 = make_uint4(l[0], l[1], l[2], l[3]);
}

// TIMING: phase timestamps (s_memtime) of every workgroup summed into `tprof` (measurement hook only)
// DBG (measurement hook only, results invalid): 1 = no matrix steps, 2 = no output stores, 3 = no input loads, 4 = 1 + 2
template <class Cfg, bool TIMING = false, int DBG = 0>
__global__ __launch_bounds__(Cfg::NT, Cfg::WGS_PER_CU) void conv_bf6_kernel(const float* __restrict__ in, const GNStats* __restrict__ st_in,
                                                                const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                                const uint32_t* __restrict__ wfrag, const float* __restrict__ bias,
                                                                float* __restrict__ out, GNStats* __restrict__ st_out, int N,
                                                                float xscale, float unscale,
                                                                unsigned long long* __restrict__ tprof = nullptr) {
    long long tstamp[8];
    int nstamp = 0;
    auto stamp = [&]() { if (TIMING) tstamp[nstamp++] = clock64(); };
    stamp();
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, IH = Cfg::IH, OH = Cfg::OH, NT = Cfg::NT;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, ITW = Cfg::ITW, PT = Cfg::PT;
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* s_in = reinterpret_cast<unsigned char*>(smem);
    unsigned char* s_w = s_in + Cfg::IN_B;
    float* s_gn = (float*)(s_w + 3 * Cfg::WSTEP_B);              // [CIN][2] scale, shift
    double* s_red = (double*)(s_gn + 2 * CIN);
    float* s_mr = (float*)(s_red + 2 * Cfg::NW);
    float* s_bias = s_mr + 4;                                      // [COUT_WG]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    // Workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  All tiles of a sample
    // are therefore given ids of the same residue mod 8 (id = 8 (tile + T (n / 8)) + n % 8; the grid's z extent is N rounded up
    // to a multiple of 8 and the surplus workgroups leave at once): neighbouring tiles then share their halo rows / columns in
    // ONE L2 instead of fetching them twice from HBM, and the 128-byte lines that straddle tile edges (output rows are not
    // multiples of a line: conv2 61 px x 32 B) are completed in that L2 instead of being evicted half written
    // (rocprofv3 FETCH_SIZE of conv2: 644 -> 602 MB per 512-agent launch with x-tile pairs co-located; all tiles: see
    // profiles/r02_traffic.json).
    int bx, by, n;
    {
        const unsigned T = gridDim.x * gridDim.y;
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned q = lin >> 3, tile = q % T;
        n = (int)(8u * (q / T) + (lin & 7u));
        bx = (int)(tile % gridDim.x);
        by = (int)(tile / gridDim.x);
        if (n >= N) return;
    }
    const int tile_x = bx % Cfg::TILES_X, cb = bx / Cfg::TILES_X;
    const int oy0 = by * TH, ox0 = tile_x * TW;
    const int iy0 = 2 * oy0, ix0 = 2 * ox0;

    // ---- raw input loads of pass 0 go out first: they do not depend on the statistics ----
    const float* in_n = in + (size_t)n * IH * IH * CIN;          // [c/8][y][x][c%8]
    float4 raw[Cfg::UITERS][2];
    auto issue_loads = [&](int pass) {
#pragma unroll
        for (int k = 0; k < Cfg::UITERS; ++k) {
            const int idx = tid + k * NT;
            raw[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            raw[k][1] = raw[k][0];
            if (idx < Cfg::UNITS && DBG != 3) {
                const int col = idx % ITW, r = idx / ITW;
                const int iy = iy0 + r, ix = ix0 + col;
                if (iy < IH && ix < IH) {
                    const float4* src = reinterpret_cast<const float4*>(in_n + (((size_t)pass * IH + iy) * IH + ix) * 8);
                    raw[k][0] = src[0];
                    raw[k][1] = src[1];
                }
            }
        }
    };
    issue_loads(0);
    const float my_g = tid < CIN ? gn_g[tid] : 0.f, my_b = tid < CIN ? gn_b[tid] : 0.f;   // in flight during the reduction
    // The epilogue takes the bias from LDS (round 6).  A global load in the epilogue is waited for with vmcnt(0) (the wait-count pass
    // merges the divergent store blocks conservatively), and on gfx9 stores count in vmcnt too: every bias load waited for the
    // stores issued before it -- 16 serial store round trips per wave in conv3's epilogue.
    if (tid < Cfg::COUT_WG) s_bias[tid] = bias[cb * Cfg::COUT_WG + tid];

    // ---- GroupNorm moments of the input sample: the producer's per-tile partial sums, one per lane, reduced in a
    // fixed (butterfly) order ----
    if (wave == 0) {
        double ps = 0.0, pq = 0.0;
        for (int i = lane; i < Cfg::NPART_IN; i += 64) {
            ps += st_in[(size_t)n * Cfg::NPART_IN + i].sum;
            pq += st_in[(size_t)n * Cfg::NPART_IN + i].sq;
        }
        ps = wave_sum_d(ps);
        pq = wave_sum_d(pq);
        if (lane == 0) {
            const double cnt = (double)CIN * IH * IH;
            const double mu = ps / cnt;
            double var = pq / cnt - mu * mu;
            var = var < 0.0 ? 0.0 : var;
            s_mr[0] = (float)mu;
            s_mr[1] = (float)(1.0 / sqrt(var + GN_EPS));
        }
    }
    __syncthreads();
    if (tid < CIN) {
        // xscale = 2^k folded into the affine map: relu(2^k (a x + b)) = 2^k relu(a x + b), exact
        const float sc = s_mr[1] * my_g;
        s_gn[2 * tid] = sc * xscale;
        s_gn[2 * tid + 1] = (my_b - s_mr[0] * sc) * xscale;
    }
    stamp();

    constexpr int CBW = Cfg::CBW, NKS = Cfg::NKS;
    f32x16 acc[CBW][PT];
#pragma unroll
    for (int c = 0; c < CBW; ++c)
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

    // this lane's pixel inside a pixel tile (row, column) and the byte offset of its window origin in the wave's first
    // tile; tile i is 2 * TILE_ROWS * i input rows further
    const int prow = Cfg::ROWS2 ? (j >> 4) : 0, pcol = Cfg::ROWS2 ? (j & 15) : j;
    const int lane_base = (2 * (Cfg::TILE_ROWS * PT * wave + prow)) * Cfg::ROW_B + pcol * 16;

    const uint4* wsrc = reinterpret_cast<const uint4*>(wfrag);
    constexpr int WQ = Cfg::WSTEP_B / 16;                         // 128 CBW x 16 B per matrix step
    // fragments are stored [pass][tap pair][co / 32][piece][lane]: the CBW blocks of this workgroup are adjacent
    auto wstep_src = [&](int pass, int t) { return wsrc + ((size_t)(pass * NKS + t) * (COUT / 32) + cb * CBW) * 128; };
    const bool wmover = tid < WQ;                                 // the first 2 CBW waves move the weight fragments

    for (int pass = 0; pass < Cfg::NPASS; ++pass) {
        // ALL weight fragments of the pass (13 x 16 bytes per moving thread) are requested up front and parked in
        // registers: an L2 round trip under load (~1 us) is longer than two MFMA steps, a shallower prefetch paces
        // the whole pipeline at latency / depth (measured with the s_memtime phase profile, tools/conv_phase_profile.py)
        uint4 wq[NKS];
#pragma unroll
        for (int t = 0; t < NKS; ++t) wq[t] = make_uint4(0u, 0u, 0u, 0u);
        if (wmover) {
#pragma unroll
            for (int t = 0; t < Cfg::W_UPFRONT; ++t) wq[t] = wstep_src(pass, t)[tid];
        }
        __syncthreads();        // s_gn ready (pass 0) / every wave is done with the previous pass's tiles
        // ---- 8 input channels of the (2TH+KS-2) x (2TW+KS-2) window: GroupNorm + ReLU (pre-scaled), two-piece fp16 split ----
#pragma unroll
        for (int k = 0; k < Cfg::UITERS; ++k) {
            const int idx = tid + k * NT;
            if (idx < Cfg::UNITS) {
                const int col = idx % ITW, r = idx / ITW;
                const int iy = iy0 + r, ix = ix0 + col;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;             // exact zero outside the image
                if (iy < IH && ix < IH) {
                    const float4 a = raw[k][0], b = raw[k][1];
                    const float4* gn = reinterpret_cast<const float4*>(s_gn + 2 * pass * Cfg::PASS_CH);
                    const float4 g0 = gn[0], g1 = gn[1], g2 = gn[2], g3 = gn[3];
                    v[0] = fmaxf(fmaf(a.x, g0.x, g0.y), 0.f);
                    v[1] = fmaxf(fmaf(a.y, g0.z, g0.w), 0.f);
                    v[2] = fmaxf(fmaf(a.z, g1.x, g1.y), 0.f);
                    v[3] = fmaxf(fmaf(a.w, g1.z, g1.w), 0.f);
                    v[4] = fmaxf(fmaf(b.x, g2.x, g2.y), 0.f);
                    v[5] = fmaxf(fmaf(b.y, g2.z, g2.w), 0.f);
                    v[6] = fmaxf(fmaf(b.z, g3.x, g3.y), 0.f);
                    v[7] = fmaxf(fmaf(b.w, g3.z, g3.w), 0.f);
                }
                uint4 p0, p1;
                split_f16x2(v, p0, p1);
                unsigned char* dst = s_in + r * Cfg::ROW_B + (col & 1) * Cfg::HALF_B + (col >> 1) * 16;
                *reinterpret_cast<uint4*>(dst) = p0;
                *reinterpret_cast<uint4*>(dst + Cfg::PIECE_B) = p1;
            }
        }
        // ---- weight fragments: step t lives in LDS buffer t % 3.  Steps 0 and 1 go straight in; step t is written at the
        // end of step t-2 (that buffer was last read during step t-4: two barriers earlier) and read into the fragment
        // registers during step t-1 ----
        if (wmover) {
            reinterpret_cast<uint4*>(s_w)[tid] = wq[0];
            reinterpret_cast<uint4*>(s_w + Cfg::WSTEP_B)[tid] = wq[1];
        }
        __syncthreads();
        if (pass < 2) stamp();
        if (pass + 1 < Cfg::NPASS) issue_loads(pass + 1);      // consumed after this pass's matrix work

        // Software pipeline over the MFMA steps: while the matrix cores work on step s, the A/B fragments of step
        // s+1 are read from LDS into the other register set; one barrier per step.
        f16x8 fa[2][CBW][2], fb[2][PT][2];
        auto load_frags = [&](int t, int set) {
            int ky, kx;
            if (Cfg::KS == 5) {
                if (t < 10) { ky = t >> 1; kx = (t & 1) + 2 * h; }
                else { ky = 2 * (t - 10) + h; kx = 4; ky = ky > 4 ? 4 : ky; }     // (row 5 does not exist: zero-weight slot)
            } else {        // 3x3: steps 0-2 = row t, columns 0 and 2; step 3 = column 1 of rows 0, 1; step 4 = (2, 1) + zero slot
                if (t < 3) { ky = t; kx = 2 * h; }
                else if (t == 3) { ky = h; kx = 1; }
                else { ky = 2; kx = 1; }
            }
            const int off = ky * Cfg::ROW_B + (kx & 1) * Cfg::HALF_B + (kx >> 1) * 16;
            const unsigned char* wb = s_w + (t % 3) * Cfg::WSTEP_B + lane * 16;
#pragma unroll
            for (int c = 0; c < CBW; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fa[set][c][pl] = *reinterpret_cast<const f16x8*>(wb + (c * 2 + pl) * 1024);
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fb[set][i][pl] = *reinterpret_cast<const f16x8*>(s_in + pl * Cfg::PIECE_B + 2 * Cfg::TILE_ROWS * i * Cfg::ROW_B + lane_base + off);
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < ((DBG == 1 || DBG == 4) ? 0 : NKS); ++s) {
            const int cur = s & 1;
            if (s + 1 < NKS) load_frags(s + 1, cur ^ 1);
            // three products per (channel block, pixel tile) (w1 x0, w0 x1, w0 x0: the small ones first; w1 x1 is below 2^-24
            // of the leading product); the accumulation chains alternate so that an MFMA never waits for the one issued just
            // before it
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int c = 0; c < CBW; ++c)
#pragma unroll
                    for (int i = 0; i < PT; ++i)
                        acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c][TA[term]], fb[cur][i][TB[term]], acc[c][i], 0, 0, 0);
            // issue order: one LDS fragment read of step s+1 behind each MFMA of step s (issuing the reads up front stalls
            // the wave on the LDS queue before the matrix pipe gets any work)
            if (s + 1 < NKS) {
                constexpr int NRD = 2 * CBW + 2 * PT, NMF = 3 * PT * CBW;
#pragma unroll
                for (int q = 0; q < (NRD < NMF ? NRD : NMF); ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
                }
                if (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
                if (NRD > NMF) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMF, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < NKS && wmover) reinterpret_cast<uint4*>(s_w + ((s + 2) % 3) * Cfg::WSTEP_B)[tid] = wq[s + 2];
            if (s == Cfg::W_LATE_AT && Cfg::W_UPFRONT < NKS && wmover) {
#pragma unroll
                for (int t = Cfg::W_UPFRONT; t < NKS; ++t) wq[t] = wstep_src(pass, t)[tid];
            }
            if (s + 1 < NKS) __syncthreads();
        }
        if (pass < 2) stamp();
    }

    // ---- epilogue: D column = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel within the block ----
    // GroupNorm moments: the 16 outputs of one accumulator tile (one pixel x 16 channels) are summed in fp32, everything above that
    // in float64.  The fp32 unit is the same set of values in the same order for every tiling of the layer (CBW, PT are per-form
    // parameters: DESIGN.md 4.10), so the forms differ only in the grouping of float64 additions: 1e-16, i.e. the same fp32 mean and
    // rstd -- a scene decoded alone and inside a large batch gets the same map features.
    double dsum = 0.0, dsq = 0.0;
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int oy = oy0 + Cfg::TILE_ROWS * (PT * wave + i) + prow, ox = ox0 + pcol;
            const bool valid = oy < OH && ox < OH;
            float fsum = 0.f, fsq = 0.f;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int co = (cb * CBW + c) * 32 + 8 * rg + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>(s_bias + c * 32 + 8 * rg + 4 * h);
                float4 v;
                v.x = fmaf(acc[c][i][4 * rg + 0], unscale, bv.x);     // unscale = 2^-k exactly: one rounding, like acc + bias
                v.y = fmaf(acc[c][i][4 * rg + 1], unscale, bv.y);
                v.z = fmaf(acc[c][i][4 * rg + 2], unscale, bv.z);
                v.w = fmaf(acc[c][i][4 * rg + 3], unscale, bv.w);
                if (valid) {
                    if (DBG == 2 || DBG == 4) {
                        // (no stores)
                    } else if (Cfg::OUT_OCT) {     // octet-planar, [n][c/8][y][x][c%8]
                        *reinterpret_cast<float4*>(out + ((((size_t)n * (COUT / 8) + (co >> 3)) * OH + oy) * OH + ox) * 8 + (co & 7)) = v;
                    } else {                // NCHW
                        float* o = out + (((size_t)n * COUT + co) * OH + oy) * OH + ox;
                        o[0] = v.x;
                        o[(size_t)OH * OH] = v.y;
                        o[(size_t)2 * OH * OH] = v.z;
                        o[(size_t)3 * OH * OH] = v.w;
                    }
                    fsum += (v.x + v.y) + (v.z + v.w);
                    fsq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fsq))));
                }
            }
            dsum += (double)fsum;
            dsq += (double)fsq;
        }
    }
    const double lsum = wave_sum_d(dsum), lsq = wave_sum_d(dsq);
    if (lane == 0) { s_red[2 * wave] = lsum; s_red[2 * wave + 1] = lsq; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < Cfg::NW; ++w) { a += s_red[2 * w]; b += s_red[2 * w + 1]; }
        GNStats& o = st_out[(size_t)n * Cfg::NPART_OUT + (by * Cfg::TILES_X + tile_x) * Cfg::CSPLIT + cb];
        o.sum = a;
        o.sq = b;
    }
    stamp();
    if (TIMING && tid == 0) {
        // [0] = workgroups, [1 + k] = sum of (stamp k+1 - stamp k): statistics+first loads, staging 0, steps 0,
        // staging 1, steps 1, (remaining passes +) epilogue
        atomicAdd(tprof, 1ull);
        for (int k = 0; k + 1 < nstamp; ++k) atomicAdd(tprof + 1 + k, (unsigned long long)(tstamp[k + 1] - tstamp[k]));
    }
}

// =============================================================================================
// conv_bf6_kernel with SPECIALISED waves (round 3).  tools/conv_floor_probe.py: the matrix loop of conv2 is 42 % of the kernel,
// its memory phases the rest, and the two overlap only across the 3 workgroups of a CU.  Here a persistent workgroup of 8 waves
// (one per CU) splits the roles: waves 4-7 PRODUCE -- they fetch the 8-channel slice of the next (tile, pass) unit, apply
// GroupNorm + ReLU, split and write it to one of two LDS input buffers -- while waves 0-3 CONSUME the other buffer: the 13 matrix
// steps of the previous unit (same instruction order as conv_bf6_kernel: the results are bit-identical) and, after a tile's last
// pass, its epilogue.  One barrier per unit.  The whole layer's weight fragments stay in LDS (52 KB for conv2), so the matrix
// loop has no ring and no barrier of its own.  A workgroup walks a contiguous range of tiles (all tiles of a sample in a row:
// halo rows come from its own XCD's L2).
// =============================================================================================
template <class Cfg>
struct WsCfg {
    static constexpr int NCONS_W = 8, NT = 1024, NPROD = 512;        // 8 consumer waves (one output row each), 8 producer waves
    static constexpr int UITERS = (Cfg::UNITS + NPROD - 1) / NPROD;
    static constexpr int W_B = Cfg::NPASS * Cfg::NKS * Cfg::WSTEP_B;
    static constexpr size_t LDS_BYTES = 2 * (size_t)Cfg::IN_B + W_B + 2 * (size_t)Cfg::CIN * 8 + 2 * 8 * 16 + 64 + Cfg::COUT * 4;
    static_assert(Cfg::CBW == 1 && Cfg::CSPLIT == 1 && Cfg::OUT_OCT && !Cfg::ROWS2 && Cfg::TH == NCONS_W, "built for conv2's shape");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// the (tile, pass) units of a workgroup and where unit u lives
template <class Cfg>
struct WsUnits {
    int t_begin, nunit;
    __device__ __forceinline__ void tile(int u, int& n, int& ty, int& tx, int& pass) const {
        constexpr int TPS = Cfg::TILES_X * Cfg::TILES_Y;
        const int t = t_begin + u / Cfg::NPASS;
        pass = u - (u / Cfg::NPASS) * Cfg::NPASS;
        n = t / TPS;
        const int r = t - n * TPS;
        // down a column of tiles first: vertical neighbours share 3 rows x 67 columns of halo (12.9 KB per pass, horizontal ones
        // 3.6 KB), and with the next tile one unit away those rows are still in the XCD's L2 (row-major order: 1.13 x the
        // algorithmic bytes fetched)
        tx = r / Cfg::TILES_Y;
        ty = r - tx * Cfg::TILES_Y;
    }
};

// dbg (STRIVE_CONV_WS_DBG, measurement only, results invalid): 1 = consumers skip the matrix steps, 2 = producers skip the staging,
// 4 = producers skip the input loads, 8 = every unit loads the same (L2-resident) tile, 16 = consumers skip the epilogue.
// tprof (bench-layer code 81): clock sums of consumer wave 0 (matrix steps, epilogue, barrier wait) and of the first producer wave
// (request, stage, barrier wait).  What they showed in round 6 (profiles/r06_conv_ws_decomposition.txt): matrix steps alone 115 us,
// + epilogue 40 (in series with them), producers alone 117 us (memory), all together 190: two pipelines of about the same service
// time coupled by a barrier per unit -- the sum of the per-unit maxima, not the maximum of the sums.
// ---- producer waves (8-15): own function = own register allocation (raw prefetch sets; the consumers keep accumulators) ----
template <class Cfg>
__device__ __forceinline__ void ws_producer(const float* __restrict__ in, const GNStats* __restrict__ st_in, const float* __restrict__ gn_g,
                                         const float* __restrict__ gn_b, float xscale, unsigned char* s_in, float* s_gn,
                                         WsUnits<Cfg> un, int dbg, unsigned long long* __restrict__ tprof) {
    using W = WsCfg<Cfg>;
    constexpr int CIN = Cfg::CIN, IH = Cfg::IH, TH = Cfg::TH, TW = Cfg::TW, ITW = Cfg::ITW;
    long long tp_req = 0, tp_stage = 0, tp_bar = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptid = tid - 64 * W::NCONS_W;
    const int nunit = un.nunit;
    // Two register sets: the loads of unit u + 1 are requested before unit u is staged, a whole unit ahead of their use.  The loads
    // are UNCONDITIONAL (addresses clamped into the tensor; stage() zeroes what lies outside): with predicated loads the compiler
    // cannot count what is outstanding and waits for ALL of it (s_waitcnt vmcnt(0)) before staging, i.e. also for the loads it
    // has just issued -- the period was then load latency + staging, whatever the prefetch distance.  (A third set, loads two units
    // ahead, changed nothing in round 6: 194 against 193 us.)
    float4 raw0[W::UITERS][2], raw1[W::UITERS][2];
    auto issue_loads = [&](int u, float4 (&raw)[W::UITERS][2]) {
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const float* in_n = in + (size_t)((dbg & 8) ? 0 : n) * IH * IH * CIN;
        const int iy0 = (dbg & 8) ? 0 : 2 * ty * TH, ix0 = (dbg & 8) ? 0 : 2 * tx * TW;
#pragma unroll
        for (int k = 0; k < W::UITERS; ++k) {
            int idx = ptid + k * W::NPROD;
            idx = idx < Cfg::UNITS ? idx : Cfg::UNITS - 1;
            const int col = idx % ITW, r = idx / ITW;
            int iy = iy0 + r, ix = ix0 + col;
            iy = iy < IH ? iy : IH - 1;
            ix = ix < IH ? ix : IH - 1;
            const float4* src = reinterpret_cast<const float4*>(in_n + (((size_t)pass * IH + iy) * IH + ix) * 8);
            raw[k][0] = src[0];
            raw[k][1] = src[1];
        }
    };
    // GroupNorm scale / shift of sample n into s_gn[n & 1] (first producer wave; fixed butterfly order like conv_bf6_kernel)
    auto sample_moments = [&](int n) {
        double ps = 0.0, pq = 0.0;
        for (int i = lane; i < Cfg::NPART_IN; i += 64) {
            ps += st_in[(size_t)n * Cfg::NPART_IN + i].sum;
            pq += st_in[(size_t)n * Cfg::NPART_IN + i].sq;
        }
        ps = wave_sum_d(ps);
        pq = wave_sum_d(pq);
        const double cnt = (double)CIN * IH * IH;
        const double mu = ps / cnt;
        double var = pq / cnt - mu * mu;
        var = var < 0.0 ? 0.0 : var;
        const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + GN_EPS));
        if (lane < CIN) {
            const float sc = rstd * gn_g[lane];
            float* g = s_gn + (n & 1) * CIN * 2;
            g[2 * lane] = sc * xscale;
            g[2 * lane + 1] = (gn_b[lane] - mean * sc) * xscale;
        }
    };
    auto stage = [&](int u, const float4 (&raw)[W::UITERS][2]) {   // raw -> GroupNorm + ReLU -> two fp16 pieces -> s_in[u & 1]
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const int iy0 = 2 * ty * TH, ix0 = 2 * tx * TW;
        unsigned char* buf = s_in + (u & 1) * Cfg::IN_B;
        const float4* gn = reinterpret_cast<const float4*>(s_gn + (n & 1) * CIN * 2 + 2 * pass * Cfg::PASS_CH);
        const float4 g0 = gn[0], g1 = gn[1], g2 = gn[2], g3 = gn[3];
#pragma unroll
        for (int k = 0; k < W::UITERS; ++k) {
            const int idx = ptid + k * W::NPROD;
            if (idx < Cfg::UNITS) {
                const int col = idx % ITW, r = idx / ITW;
                const int iy = iy0 + r, ix = ix0 + col;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;             // exact zero outside the image
                if (iy < IH && ix < IH) {
                    const float4 a = raw[k][0], b = raw[k][1];
                    v[0] = fmaxf(fmaf(a.x, g0.x, g0.y), 0.f);
                    v[1] = fmaxf(fmaf(a.y, g0.z, g0.w), 0.f);
                    v[2] = fmaxf(fmaf(a.z, g1.x, g1.y), 0.f);
                    v[3] = fmaxf(fmaf(a.w, g1.z, g1.w), 0.f);
                    v[4] = fmaxf(fmaf(b.x, g2.x, g2.y), 0.f);
                    v[5] = fmaxf(fmaf(b.y, g2.z, g2.w), 0.f);
                    v[6] = fmaxf(fmaf(b.z, g3.x, g3.y), 0.f);
                    v[7] = fmaxf(fmaf(b.w, g3.z, g3.w), 0.f);
                }
                uint4 p0, p1;
                split_f16x2(v, p0, p1);
                unsigned char* dst = buf + r * Cfg::ROW_B + (col & 1) * Cfg::HALF_B + (col >> 1) * 16;
                *reinterpret_cast<uint4*>(dst) = p0;
                *reinterpret_cast<uint4*>(dst + Cfg::PIECE_B) = p1;
            }
        }
    };
    int cur_sample;
    auto request = [&](int u, float4 (&raw)[W::UITERS][2]) {       // loads of unit u (+ its sample's scale / shift when it is a new one)
        if (u >= nunit) return;
        if (!(dbg & 4)) issue_loads(u, raw);
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        if (n != cur_sample) {                                    // (its slot s_gn[n & 1] was last read while staging sample n - 2)
            if (wave == W::NCONS_W) sample_moments(n);
            cur_sample = n;
        }
    };
    cur_sample = -1;
    request(0, raw0);
    __syncthreads();                                              // (1) weights, bias and the first sample's scale / shift are in LDS
    auto iteration = [&](int u, float4 (&raw_cur)[W::UITERS][2], float4 (&raw_next)[W::UITERS][2]) {
        const long long t0 = tprof ? clock64() : 0;
        request(u + 1, raw_next);
        const long long t1 = tprof ? clock64() : 0;
        if (u < nunit && !(dbg & 2)) stage(u, raw_cur);
        const long long t2 = tprof ? clock64() : 0;
        __syncthreads();
        if (tprof) { tp_req += t1 - t0; tp_stage += t2 - t1; tp_bar += clock64() - t2; }
    };
    for (int u = 0; u <= nunit; u += 2) {
        iteration(u, raw0, raw1);
        if (u + 1 <= nunit) iteration(u + 1, raw1, raw0);
    }
    if (tprof && ptid == 0) {
        atomicAdd(tprof + 4, (unsigned long long)tp_req);
        atomicAdd(tprof + 5, (unsigned long long)tp_stage);
        atomicAdd(tprof + 6, (unsigned long long)tp_bar);
    }
}

// ---- consumer waves (0-7): wave w owns output row w of the 8-row tile ----
template <class Cfg>
__device__ __forceinline__ void ws_consumer(const unsigned char* s_in, const unsigned char* s_w, double* s_red, const float* s_bias,
                                         float* __restrict__ out, GNStats* __restrict__ st_out, float unscale, WsUnits<Cfg> un, int dbg,
                                         unsigned long long* __restrict__ tprof) {
    constexpr int COUT = Cfg::COUT, OH = Cfg::OH, TH = Cfg::TH, TW = Cfg::TW, NKS = Cfg::NKS, NPASS = Cfg::NPASS;
    long long tc_mat = 0, tc_epi = 0, tc_bar = 0, tc_mark = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int nunit = un.nunit;
    f32x16 acc0;
    const int lane_base = (2 * wave) * Cfg::ROW_B + j * 16;
    auto matrix_steps = [&](int u) {
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const unsigned char* buf = s_in + (u & 1) * Cfg::IN_B;
        if (pass == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        }
        f16x8 fa[2][2], fb[2][2];
        auto load_frags = [&](int t, int set) {
            int ky, kx;
            if (Cfg::KS == 5) {
                if (t < 10) { ky = t >> 1; kx = (t & 1) + 2 * h; }
                else { ky = 2 * (t - 10) + h; kx = 4; ky = ky > 4 ? 4 : ky; }
            } else {
                if (t < 3) { ky = t; kx = 2 * h; }
                else if (t == 3) { ky = h; kx = 1; }
                else { ky = 2; kx = 1; }
            }
            const int off = ky * Cfg::ROW_B + (kx & 1) * Cfg::HALF_B + (kx >> 1) * 16;
            const unsigned char* wb = s_w + (size_t)(pass * NKS + t) * Cfg::WSTEP_B + lane * 16;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                fa[set][pl] = *reinterpret_cast<const f16x8*>(wb + pl * 1024);
                fb[set][pl] = *reinterpret_cast<const f16x8*>(buf + pl * Cfg::PIECE_B + lane_base + off);
            }
        };
        load_frags(0, 0);
#pragma unroll
        for (int s2 = 0; s2 < NKS; ++s2) {
            const int cur = s2 & 1;
            if (s2 + 1 < NKS) load_frags(s2 + 1, cur ^ 1);
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][TA[term]], fb[cur][TB[term]], acc0, 0, 0, 0);
            // issue order pinned (round 6): one fragment read of step s+1 behind each matrix instruction of step s.  Left to itself the
            // scheduler (which does not know that the dynamic LDS allocation admits one workgroup per CU, and so minimises registers)
            // re-reads each weight fragment into ONE register set just before its use: ds_read, s_waitcnt lgkmcnt(0), two matrix
            // instructions, ds_read ... -- an exposed LDS round trip per pair of matrix instructions.
            if (s2 + 1 < NKS) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tprof) tc_mark = clock64();
        if (pass + 1 < NPASS || (dbg & 16)) return;
        // ---- epilogue of the tile: the bias comes from LDS (see conv_bf6_kernel) ----
        const int oy = ty * TH + wave, ox = tx * TW + j;
        const bool valid = oy < OH && ox < OH;
        float fsum = 0.f, fsq = 0.f;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int co = 8 * rg + 4 * h;
            const float4 bv = *reinterpret_cast<const float4*>(s_bias + co);
            float4 v;
            v.x = fmaf(acc0[4 * rg + 0], unscale, bv.x);
            v.y = fmaf(acc0[4 * rg + 1], unscale, bv.y);
            v.z = fmaf(acc0[4 * rg + 2], unscale, bv.z);
            v.w = fmaf(acc0[4 * rg + 3], unscale, bv.w);
            if (valid) {
                *reinterpret_cast<float4*>(out + ((((size_t)n * (COUT / 8) + (co >> 3)) * OH + oy) * OH + ox) * 8 + (co & 7)) = v;
                fsum += (v.x + v.y) + (v.z + v.w);
                fsq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fsq))));
            }
        }
        const double lsum = wave_sum_d((double)fsum), lsq = wave_sum_d((double)fsq);
        const int tl = (u / NPASS) & 1;
        if (lane == 0) { s_red[(tl * 8 + wave) * 2] = lsum; s_red[(tl * 8 + wave) * 2 + 1] = lsq; }
    };
    auto publish_stats = [&](int u) {                              // thread 0, one barrier after the tile's epilogue
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const int tl = (u / NPASS) & 1;
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 8; ++w) { a += s_red[(tl * 8 + w) * 2]; b += s_red[(tl * 8 + w) * 2 + 1]; }
        GNStats& o = st_out[(size_t)n * Cfg::NPART_OUT + (ty * Cfg::TILES_X + tx)];
        o.sum = a;
        o.sq = b;
    };
    __syncthreads();                                              // (1)
    for (int u = 0; u <= nunit; ++u) {
        const long long t0 = tprof ? clock64() : 0;
        tc_mark = t0;
        if (u >= 1 && !(dbg & 1)) matrix_steps(u - 1);
        if (tid == 0 && u >= 2 && ((u - 2) % NPASS) == NPASS - 1) publish_stats(u - 2);
        const long long t1 = tprof ? clock64() : 0;
        __syncthreads();
        if (tprof) { tc_mat += tc_mark - t0; tc_epi += t1 - tc_mark; tc_bar += clock64() - t1; }
    }
    if (tid == 0 && ((nunit - 1) % NPASS) == NPASS - 1) publish_stats(nunit - 1);
    if (tprof && tid == 0) {
        atomicAdd(tprof + 0, 1ull);
        atomicAdd(tprof + 1, (unsigned long long)tc_mat);
        atomicAdd(tprof + 2, (unsigned long long)tc_epi);
        atomicAdd(tprof + 3, (unsigned long long)tc_bar);
    }
}

template <class Cfg>
__global__ __launch_bounds__(1024, 1) void conv_ws_kernel(const float* __restrict__ in, const GNStats* __restrict__ st_in,
                                                           const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                           const uint32_t* __restrict__ wfrag, const float* __restrict__ bias,
                                                           float* __restrict__ out, GNStats* __restrict__ st_out, int N, float xscale,
                                                           float unscale, int dbg, unsigned long long* __restrict__ tprof) {
    using W = WsCfg<Cfg>;
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* s_in = reinterpret_cast<unsigned char*>(smem);          // [2][IN_B]
    unsigned char* s_w = s_in + 2 * Cfg::IN_B;                              // [pass][step][piece][lane][16 B]
    float* s_gn = (float*)(s_w + W::W_B);                                   // [2][CIN][2] scale, shift of the sample (parity)
    double* s_red = (double*)(s_gn + 2 * Cfg::CIN * 2);                     // [2][8][2]
    float* s_bias = (float*)(s_red + 2 * 8 * 2) + 16;                       // [COUT]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this workgroup's tiles: a contiguous range of the (sample, tile) list
    const int total = N * Cfg::TILES_X * Cfg::TILES_Y;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    WsUnits<Cfg> un;
    un.t_begin = (int)blockIdx.x * per;
    const int t_end = (un.t_begin + per) < total ? (un.t_begin + per) : total;
    un.nunit = (t_end > un.t_begin ? t_end - un.t_begin : 0) * Cfg::NPASS;
    for (int i = tid; i < W::W_B / 16; i += W::NT) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(wfrag)[i];
    if (tid < Cfg::COUT) s_bias[tid] = bias[tid];
    if (un.nunit == 0) return;
    // both roles execute the same sequence of barriers: (1), then one per unit
    if (wave >= W::NCONS_W) ws_producer<Cfg>(in, st_in, gn_g, gn_b, xscale, s_in, s_gn, un, dbg, tprof);
    else ws_consumer<Cfg>(s_in, s_w, s_red, s_bias, out, st_out, unscale, un, dbg, tprof);
}

// =============================================================================================
// Specialised waves for the layers whose weight fragments do NOT fit LDS (conv3: 208 KB, conv4: 147 KB) -- round 6.
// conv_bf6_kernel's phases (fetch -> GroupNorm + split -> matrix steps -> epilogue) run one after the other inside a workgroup and
// overlap only by chance across the two workgroups of a CU (tools/conv_floor_probe.py: conv3 = 84 us without matrix steps + 71 us of
// matrix steps).  Here ONE persistent workgroup per CU of 12 waves walks a contiguous range of (tile, 8-channel pass) units:
//   waves 0-3   CONSUME: the matrix steps of conv_bf6_kernel on the same pixel tiles in the same order (bit-identical outputs and
//               moments), reading the unit's input from one of two LDS buffers and its weight fragments from one of two LDS
//               weight buffers; the tile's epilogue after its last pass;
//   waves 4-10  PRODUCE: fetch the next unit's 8-channel slice a whole unit ahead into two register sets (unconditional loads:
//               countable), GroupNorm + ReLU + two-piece fp16 split, write the other input buffer;
//   wave 11     STREAMS the weights: a unit's 13 matrix steps are two SUB-UNITS of 7 + 6 steps (28 + 24 KB of fragments; the 5 steps
//               of a 3x3 layer are one), loaded into the weight buffer the consumers are not reading.
// One barrier per sub-unit: none inside the matrix loop, no per-step weight ring.
// =============================================================================================
template <class Cfg, int NPW_ = 7>
struct WxCfg {
    static constexpr int NCW = Cfg::NW, GW = Cfg::NW, NPW = NPW_, NT = 64 * (NCW + NPW + 1), NPROD = 64 * NPW;
    static constexpr int NSUB = Cfg::NKS > 7 ? 2 : 1;                          // weight sub-units per unit
    static constexpr int SUB_STEPS = (Cfg::NKS + NSUB - 1) / NSUB;             // 7 (13 = 7 + 6) / 5
    static constexpr int WSUB_B = SUB_STEPS * Cfg::WSTEP_B;
    static constexpr int WCHUNKS = WSUB_B / 1024;                              // 1 KB (64 lanes x 16 B) pieces of a sub-unit
    static constexpr int UITERS = (Cfg::UNITS + NPROD - 1) / NPROD;
    static constexpr size_t LDS_BYTES = 2 * (size_t)Cfg::IN_B + 2 * (size_t)WSUB_B + 2 * (size_t)Cfg::CIN * 8 + 2 * GW * 16 + Cfg::COUT * 4 + 64;
    static_assert(Cfg::CSPLIT == 1 && Cfg::OUT_OCT && Cfg::NW == 4, "one workgroup computes all output channels of its tile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NT <= 1024, "waves");
};

template <class Cfg, class W>
__device__ __forceinline__ void wx_producer(const float* __restrict__ in, const GNStats* __restrict__ st_in, const float* __restrict__ gn_g,
                                         const float* __restrict__ gn_b, float xscale, unsigned char* s_in, float* s_gn, WsUnits<Cfg> un,
                                         int dbg) {
    // dbg (STRIVE_CONV_WS_DBG, timing probes, results invalid): 1 = no matrix steps, 2 = no staging, 4 = no input loads, 8 = no weight stream,
    // 16 = no epilogue
    constexpr int CIN = Cfg::CIN, IH = Cfg::IH, TH = Cfg::TH, TW = Cfg::TW, ITW = Cfg::ITW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ptid = tid - 64 * W::NCW;
    const int nunit = un.nunit;
    float4 raw0[W::UITERS][2], raw1[W::UITERS][2];
    auto issue_loads = [&](int u, float4 (&raw)[W::UITERS][2]) {
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const float* in_n = in + (size_t)n * IH * IH * CIN;
        const int iy0 = 2 * ty * TH, ix0 = 2 * tx * TW;
#pragma unroll
        for (int k = 0; k < W::UITERS; ++k) {
            int idx = ptid + k * W::NPROD;
            idx = idx < Cfg::UNITS ? idx : Cfg::UNITS - 1;
            const int col = idx % ITW, r = idx / ITW;
            int iy = iy0 + r, ix = ix0 + col;
            iy = iy < IH ? iy : IH - 1;
            ix = ix < IH ? ix : IH - 1;
            const float4* src = reinterpret_cast<const float4*>(in_n + (((size_t)pass * IH + iy) * IH + ix) * 8);
            raw[k][0] = src[0];
            raw[k][1] = src[1];
        }
    };
    auto sample_moments = [&](int n) {                            // first producer wave; fixed butterfly order like conv_bf6_kernel
        double ps = 0.0, pq = 0.0;
        for (int i = lane; i < Cfg::NPART_IN; i += 64) {
            ps += st_in[(size_t)n * Cfg::NPART_IN + i].sum;
            pq += st_in[(size_t)n * Cfg::NPART_IN + i].sq;
        }
        ps = wave_sum_d(ps);
        pq = wave_sum_d(pq);
        const double cnt = (double)CIN * IH * IH;
        const double mu = ps / cnt;
        double var = pq / cnt - mu * mu;
        var = var < 0.0 ? 0.0 : var;
        const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + GN_EPS));
        if (lane < CIN) {
            const float sc = rstd * gn_g[lane];
            float* g = s_gn + (n & 1) * CIN * 2;
            g[2 * lane] = sc * xscale;
            g[2 * lane + 1] = (gn_b[lane] - mean * sc) * xscale;
        }
    };
    // iterations [k0, k1) of the staging of unit u: raw -> GroupNorm + ReLU -> two fp16 pieces -> s_in[u & 1]
    auto stage = [&](int u, const float4 (&raw)[W::UITERS][2], int chunk) {
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const int iy0 = 2 * ty * TH, ix0 = 2 * tx * TW;
        unsigned char* buf = s_in + (u & 1) * Cfg::IN_B;
        const float4* gn = reinterpret_cast<const float4*>(s_gn + (n & 1) * CIN * 2 + 2 * pass * Cfg::PASS_CH);
        const float4 g0 = gn[0], g1 = gn[1], g2 = gn[2], g3 = gn[3];
#pragma unroll
        for (int k = 0; k < W::UITERS; ++k) {
            if (chunk >= 0 && (k * W::NSUB) / W::UITERS != chunk) continue;
            const int idx = ptid + k * W::NPROD;
            if (idx < Cfg::UNITS) {
                const int col = idx % ITW, r = idx / ITW;
                const int iy = iy0 + r, ix = ix0 + col;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;             // exact zero outside the image
                if (iy < IH && ix < IH) {
                    const float4 a = raw[k][0], b = raw[k][1];
                    v[0] = fmaxf(fmaf(a.x, g0.x, g0.y), 0.f);
                    v[1] = fmaxf(fmaf(a.y, g0.z, g0.w), 0.f);
                    v[2] = fmaxf(fmaf(a.z, g1.x, g1.y), 0.f);
                    v[3] = fmaxf(fmaf(a.w, g1.z, g1.w), 0.f);
                    v[4] = fmaxf(fmaf(b.x, g2.x, g2.y), 0.f);
                    v[5] = fmaxf(fmaf(b.y, g2.z, g2.w), 0.f);
                    v[6] = fmaxf(fmaf(b.z, g3.x, g3.y), 0.f);
                    v[7] = fmaxf(fmaf(b.w, g3.z, g3.w), 0.f);
                }
                uint4 p0, p1;
                split_f16x2(v, p0, p1);
                unsigned char* dst = buf + r * Cfg::ROW_B + (col & 1) * Cfg::HALF_B + (col >> 1) * 16;
                *reinterpret_cast<uint4*>(dst) = p0;
                *reinterpret_cast<uint4*>(dst + Cfg::PIECE_B) = p1;
            }
        }
    };
    int cur_sample = -1;
    auto request = [&](int u, float4 (&raw)[W::UITERS][2]) {       // loads of unit u (+ its sample's scale / shift when it is a new one)
        if (u >= nunit) return;
        if (!(dbg & 4)) issue_loads(u, raw);
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        if (n != cur_sample) {                                    // (slot s_gn[n & 1] was last read while staging sample n - 2)
            if (wave == W::NCW) sample_moments(n);
            cur_sample = n;
        }
    };
    request(0, raw0);
    __syncthreads();                                              // (1) the first sample's scale / shift
    // interval 0: unit 0 whole
    request(1, raw1);
    if (!(dbg & 2)) stage(0, raw0, -1);
    __syncthreads();
    // intervals of unit us - 1 (consumers): stage unit us, chunk by chunk; the loads of unit us + 1 go out first
    auto unit = [&](int us, float4 (&raw_cur)[W::UITERS][2], float4 (&raw_next)[W::UITERS][2]) {
#pragma unroll
        for (int h = 0; h < W::NSUB; ++h) {
            if (h == 0) request(us + 1, raw_next);
            if (us < nunit && !(dbg & 2)) stage(us, raw_cur, h);
            __syncthreads();
        }
    };
    for (int us = 1; us <= nunit; us += 2) {                      // (us == nunit: the consumers' last unit, nothing left to stage)
        unit(us, raw1, raw0);
        if (us + 1 <= nunit) unit(us + 1, raw0, raw1);
    }
}

template <class Cfg, class W>
__device__ __forceinline__ void wx_streamer(const uint32_t* __restrict__ wfrag, unsigned char* s_w, WsUnits<Cfg> un, int dbg) {
    const int lane = threadIdx.x & 63;
    const int nunit = un.nunit;
    const uint4* wsrc = reinterpret_cast<const uint4*>(wfrag);
    // sub-unit ks = (unit, half): steps [h SUB_STEPS, ...) of the unit's pass, contiguous in the fragment array
    auto fill = [&](int ks) {
        const int u = ks / W::NSUB, h = ks - u * W::NSUB;
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const int s0 = h * W::SUB_STEPS;
        const int nst = (Cfg::NKS - s0) < W::SUB_STEPS ? (Cfg::NKS - s0) : W::SUB_STEPS;
        const uint4* src = wsrc + (size_t)(pass * Cfg::NKS + s0) * (Cfg::WSTEP_B / 16);
        uint4* dst = reinterpret_cast<uint4*>(s_w + (ks & 1) * W::WSUB_B);
        const int nchunk = nst * (Cfg::WSTEP_B / 1024);
        uint4 w[W::WCHUNKS];
#pragma unroll
        for (int i = 0; i < W::WCHUNKS; ++i) w[i] = src[(i < nchunk ? i : nchunk - 1) * 64 + lane];   // unconditional: countable
#pragma unroll
        for (int i = 0; i < W::WCHUNKS; ++i)
            if (i < nchunk) dst[i * 64 + lane] = w[i];
    };
    const int K = nunit * W::NSUB;
    __syncthreads();                                              // (1)
    fill(0);
    __syncthreads();
    for (int k = 1; k <= K; ++k) {
        if (k < K && !(dbg & 8)) fill(k);
        __syncthreads();
    }
}

template <class Cfg, class W>
__device__ __forceinline__ void wx_consumer(const unsigned char* s_in, const unsigned char* s_w, double* s_red, const float* s_bias,
                                         float* __restrict__ out, GNStats* __restrict__ st_out, float unscale, WsUnits<Cfg> un, int dbg) {
    constexpr int COUT = Cfg::COUT, OH = Cfg::OH, TH = Cfg::TH, TW = Cfg::TW, NKS = Cfg::NKS, NPASS = Cfg::NPASS, PT = Cfg::PT, CBW = Cfg::CBW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int nunit = un.nunit;
    f32x16 acc[CBW][PT];
    const int gwave = wave;
    const int prow = Cfg::ROWS2 ? (j >> 4) : 0, pcol = Cfg::ROWS2 ? (j & 15) : j;
    const int lane_base = (2 * (Cfg::TILE_ROWS * PT * gwave + prow)) * Cfg::ROW_B + pcol * 16;

    // matrix steps [S0, S1) of unit u: conv_bf6_kernel's step body (same products, same order)
    auto matrix_sub = [&](int u, auto HC) {
        constexpr int HS = decltype(HC)::value;
        constexpr int S0 = HS * W::SUB_STEPS, S1 = (S0 + W::SUB_STEPS) < NKS ? (S0 + W::SUB_STEPS) : NKS;
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const unsigned char* buf = s_in + (u & 1) * Cfg::IN_B;
        const unsigned char* wbuf = s_w + ((u * W::NSUB + HS) & 1) * W::WSUB_B;
        if (pass == 0 && HS == 0) {
#pragma unroll
            for (int c = 0; c < CBW; ++c)
#pragma unroll
                for (int i = 0; i < PT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;
        }
        f16x8 fa[2][CBW][2], fb[2][PT][2];
        auto load_frags = [&](int t, int set) {
            int ky, kx;
            if (Cfg::KS == 5) {
                if (t < 10) { ky = t >> 1; kx = (t & 1) + 2 * h; }
                else { ky = 2 * (t - 10) + h; kx = 4; ky = ky > 4 ? 4 : ky; }
            } else {
                if (t < 3) { ky = t; kx = 2 * h; }
                else if (t == 3) { ky = h; kx = 1; }
                else { ky = 2; kx = 1; }
            }
            const int off = ky * Cfg::ROW_B + (kx & 1) * Cfg::HALF_B + (kx >> 1) * 16;
            const unsigned char* wb = wbuf + (t - S0) * Cfg::WSTEP_B + lane * 16;
#pragma unroll
            for (int c = 0; c < CBW; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fa[set][c][pl] = *reinterpret_cast<const f16x8*>(wb + (c * 2 + pl) * 1024);
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fb[set][i][pl] = *reinterpret_cast<const f16x8*>(buf + pl * Cfg::PIECE_B + 2 * Cfg::TILE_ROWS * i * Cfg::ROW_B + lane_base + off);
        };
        load_frags(S0, 0);
#pragma unroll
        for (int s = S0; s < S1; ++s) {
            const int cur = (s - S0) & 1;
            if (s + 1 < S1) load_frags(s + 1, cur ^ 1);
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int c = 0; c < CBW; ++c)
#pragma unroll
                    for (int i = 0; i < PT; ++i)
                        acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c][TA[term]], fb[cur][i][TB[term]], acc[c][i], 0, 0, 0);
            if (s + 1 < S1) {
                constexpr int NRD = 2 * CBW + 2 * PT, NMF = 3 * PT * CBW;
#pragma unroll
                for (int q = 0; q < (NRD < NMF ? NRD : NMF); ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
                if (NRD > NMF) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMF, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // epilogue of the tile whose last unit is u, accumulator tiles [q0, q1) (q = c PT + i): conv_bf6_kernel's sums in the same order
    // (per lane: fp32 over the 16 values of an accumulator tile, float64 above).  All bias values of a slice come from LDS up front,
    // addresses are a uniform 64-bit sample base + a 32-bit lane offset.  After the last slice: wave sums -> s_red.
    double dsum = 0.0, dsq = 0.0;
    auto epilogue = [&](int u, int q0, int q1) {
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const int oy0 = ty * TH, ox0 = tx * TW;
        float* out_n = out + (size_t)n * COUT * OH * OH;                  // [c/8][y][x][c%8]
        constexpr int PLANE = OH * OH * 8;
        if (q0 == 0) { dsum = 0.0; dsq = 0.0; }
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                if (c * PT + i < q0 || c * PT + i >= q1) continue;
                float4 bv[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) bv[rg] = *reinterpret_cast<const float4*>(s_bias + c * 32 + 8 * rg + 4 * h);
                const int oy = oy0 + Cfg::TILE_ROWS * (PT * gwave + i) + prow, ox = ox0 + pcol;
                const bool valid = oy < OH && ox < OH;
                const int loff = (oy * OH + ox) * 8 + 4 * h;
                float fsum = 0.f, fsq = 0.f;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float4 v;
                    v.x = fmaf(acc[c][i][4 * rg + 0], unscale, bv[rg].x);
                    v.y = fmaf(acc[c][i][4 * rg + 1], unscale, bv[rg].y);
                    v.z = fmaf(acc[c][i][4 * rg + 2], unscale, bv[rg].z);
                    v.w = fmaf(acc[c][i][4 * rg + 3], unscale, bv[rg].w);
                    if (valid) {
                        if (!(dbg & 64)) *reinterpret_cast<float4*>(out_n + (loff + (c * 4 + rg) * PLANE)) = v;
                        fsum += (v.x + v.y) + (v.z + v.w);
                        fsq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fsq))));
                    }
                }
                dsum += (double)fsum;
                dsq += (double)fsq;
            }
        }
        if (q1 == CBW * PT) {
            const double lsum = wave_sum_d(dsum), lsq = wave_sum_d(dsq);
            const int tl = (u / NPASS) & 1;
            if (lane == 0) { s_red[(tl * W::GW + gwave) * 2] = lsum; s_red[(tl * W::GW + gwave) * 2 + 1] = lsq; }
        }
    };
    auto publish_stats = [&](int u) {                              // thread 0, at least one barrier after the tile's last epilogue slice
        int n, ty, tx, pass;
        un.tile(u, n, ty, tx, pass);
        const int tl = (u / NPASS) & 1;
        double a = 0.0, b = 0.0;
        for (int w = 0; w < W::GW; ++w) { a += s_red[(tl * W::GW + w) * 2]; b += s_red[(tl * W::GW + w) * 2 + 1]; }
        GNStats& o = st_out[(size_t)n * Cfg::NPART_OUT + (ty * Cfg::TILES_X + tx)];
        o.sum = a;
        o.sq = b;
    };
    constexpr int NQ = CBW * PT;
    __syncthreads();                                              // (1)
    __syncthreads();                                              // interval 0: unit 0 staged, sub-unit 0's weights in LDS
    for (int u = 0; u < nunit; ++u) {
        if (!(dbg & 1)) matrix_sub(u, std::integral_constant<int, 0>());
        if (tid == 0 && u >= 1 && ((u - 1) % NPASS) == NPASS - 1) publish_stats(u - 1);
        if (W::NSUB == 1 && (u % NPASS) == NPASS - 1 && !(dbg & 16)) epilogue(u, 0, NQ);
        __syncthreads();
        if (W::NSUB == 2) {
            if (!(dbg & 1)) matrix_sub(u, std::integral_constant<int, W::NSUB - 1>());
            if ((u % NPASS) == NPASS - 1 && !(dbg & 16)) epilogue(u, 0, NQ);
            __syncthreads();
        }
    }
    if (tid == 0 && nunit >= 1 && ((nunit - 1) % NPASS) == NPASS - 1) publish_stats(nunit - 1);
}

template <class Cfg, int NPW = 7>
__global__ __launch_bounds__(64 * (Cfg::NW + NPW + 1), 1) void conv_wsx_kernel(const float* __restrict__ in, const GNStats* __restrict__ st_in,
                                                           const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                           const uint32_t* __restrict__ wfrag, const float* __restrict__ bias,
                                                           float* __restrict__ out, GNStats* __restrict__ st_out, int N, float xscale,
                                                           float unscale, int dbg) {
    using W = WxCfg<Cfg, NPW>;
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* s_in = reinterpret_cast<unsigned char*>(smem);          // [2][IN_B]
    unsigned char* s_w = s_in + 2 * Cfg::IN_B;                              // [2][WSUB_B]
    float* s_gn = (float*)(s_w + 2 * W::WSUB_B);                            // [2][CIN][2] scale, shift of the sample (parity)
    double* s_red = (double*)(s_gn + 2 * Cfg::CIN * 2);                     // [2][GW][2]
    float* s_bias = (float*)(s_red + 2 * W::GW * 2);                        // [COUT]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = N * Cfg::TILES_X * Cfg::TILES_Y;
    const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    WsUnits<Cfg> un;
    un.t_begin = (int)blockIdx.x * per;
    const int t_end = (un.t_begin + per) < total ? (un.t_begin + per) : total;
    un.nunit = (t_end > un.t_begin ? t_end - un.t_begin : 0) * Cfg::NPASS;
    if (un.nunit == 0) return;
    if (tid < Cfg::COUT) s_bias[tid] = bias[tid];
    // every role executes the same sequence of barriers: (1), interval 0, then one per sub-unit
    if (wave < W::NCW) wx_consumer<Cfg, W>(s_in, s_w, s_red, s_bias, out, st_out, unscale, un, dbg);
    else if (wave < W::NCW + W::NPW) wx_producer<Cfg, W>(in, st_in, gn_g, gn_b, xscale, s_in, s_gn, un, dbg);
    else wx_streamer<Cfg, W>(wfrag, s_w, un, dbg);
}

// =============================================================================================
// The same bf16 x 6 scheme for SMALL images (conv5: 64 -> 128 channels, 14 x 14 -> 6 x 6; conv6: 128 -> 128, 6 x 6 -> 2 x 2;
// both 3x3): a workgroup takes S whole samples, the 32-pixel MFMA tiles are filled with the linearised pixels (sample, y, x) of those samples (S = 7: 252 of
// 256 lanes carry a pixel), and every lane keeps the LDS offset of its own window origin.  Input octet-planar, the 8
// channels of a pass for all S samples staged together ([piece][sample][row][parity][column/2][8 x bf16]); GroupNorm
// scale / shift per (sample, channel); output statistics per sample reduced in pixel order through LDS.
// =============================================================================================
template <int CIN_, int COUT_, int IH_, int OH_, int S_, int NPART_IN_, bool OUT_OCT_, int PT_ = 2, int CBW_ = 1>
struct BfsCfg {
    static constexpr int CIN = CIN_, COUT = COUT_, KS = 3, IH = IH_, OH = OH_, S = S_, NPART_IN = NPART_IN_;
    static constexpr bool OUT_OCT = OUT_OCT_;
    static constexpr int NT = 256, NW = 4, PT = PT_;
    static constexpr int PPS = OH * OH, NPIX = S * PPS;                    // pixels per sample / per workgroup
    static constexpr int CBW = CBW_, COUT_WG = 32 * CBW, CSPLIT = COUT / COUT_WG;      // CBW 32-channel blocks from one staging
    static constexpr int PASS_CH = 8, NPASS = CIN / PASS_CH;
    static constexpr int HW = (IH + 1) / 2, HALF_B = HW * 16, ROW_B = 2 * HALF_B, SAMPLE_B = IH * ROW_B, PIECE_B = S * SAMPLE_B;
    static constexpr int IN_B = (2 * PIECE_B + 255) / 256 * 256;
    static constexpr int NKS = (KS * KS + 1) / 2;
    static constexpr int WSTEP_B = CBW * 2 * 64 * 16;
    static constexpr int NPART_OUT = CSPLIT;
    static constexpr int UNITS = S * IH * IH, UITERS = (UNITS + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = (size_t)IN_B + 3 * WSTEP_B + (size_t)CIN * 8 + S * 8 + 16;
    static_assert(NW * PT * 64 * 8 <= IN_B, "the epilogue's partial sums reuse the input tile");
    static_assert(NPIX <= NW * PT * 32, "pixel tiles of the workgroup");
    static_assert(CIN % PASS_CH == 0 && COUT % COUT_WG == 0, "channel tiling");
    static_assert(LDS_BYTES * 2 <= 160 * 1024, "two workgroups per CU");
    static_assert(2 * (OH - 1) + KS <= IH, "valid convolution");
    static_assert(WSTEP_B / 16 <= NT, "one 16-byte weight piece per thread and matrix step");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 2) void conv_bf6s_kernel(const float* __restrict__ in, const GNStats* __restrict__ st_in,
                                                                 const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                                 const uint32_t* __restrict__ wfrag, const float* __restrict__ bias,
                                                                 float* __restrict__ out, GNStats* __restrict__ st_out, int N,
                                                                 float xscale, float unscale) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, IH = Cfg::IH, OH = Cfg::OH, NT = Cfg::NT, S = Cfg::S, PT = Cfg::PT;
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* s_in = reinterpret_cast<unsigned char*>(smem);
    unsigned char* s_w = s_in + Cfg::IN_B;
    float* s_gb = (float*)(s_w + 3 * Cfg::WSTEP_B);              // [CIN][2] GroupNorm gamma, beta
    float* s_mr = s_gb + CIN * 2;                                 // [S][2] mean, rstd of the input samples
    float* s_part = reinterpret_cast<float*>(s_in);               // [NW*PT*64][2] per-lane partial sums (epilogue: the tile is dead)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int cb = blockIdx.x % Cfg::CSPLIT;
    const int n0 = (blockIdx.x / Cfg::CSPLIT) * S;

    // ---- raw input loads of pass 0 first ----
    float4 raw[Cfg::UITERS][2];
    auto issue_loads = [&](int pass) {
#pragma unroll
        for (int k = 0; k < Cfg::UITERS; ++k) {
            const int idx = tid + k * NT;
            raw[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            raw[k][1] = raw[k][0];
            if (idx < Cfg::UNITS) {
                const int a = idx / (IH * IH), r = idx - a * (IH * IH);
                if (n0 + a < N) {
                    const float4* src = reinterpret_cast<const float4*>(in + ((((size_t)(n0 + a) * (CIN / 8) + pass) * IH * IH) + r) * 8);
                    raw[k][0] = src[0];
                    raw[k][1] = src[1];
                }
            }
        }
    };
    issue_loads(0);

    // ---- GroupNorm moments per sample (partials added in slot order), then scale / shift per (sample, channel) ----
    if (tid < S) {
        float mean = 0.f, rstd = 0.f;
        if (n0 + tid < N) gn_moments(st_in, n0 + tid, Cfg::NPART_IN, (double)CIN * IH * IH, mean, rstd);
        s_mr[2 * tid] = mean;
        s_mr[2 * tid + 1] = rstd;
    }
    for (int c = tid; c < CIN; c += NT) {
        s_gb[2 * c] = gn_g[c];
        s_gb[2 * c + 1] = gn_b[c];
    }

    // ---- this lane's pixels: tile i of the wave holds linear pixels 32 (PT wave + i) + j ----
    int lane_base[PT];
    int pix_a[PT], pix_q[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        int g = 32 * (PT * wave + i) + j;
        const bool live = g < Cfg::NPIX;
        g = live ? g : 0;
        const int a = g / Cfg::PPS, q = g - a * Cfg::PPS;
        const int oy = q / OH, ox = q - oy * OH;
        pix_a[i] = live ? a : -1;
        pix_q[i] = q;
        lane_base[i] = a * Cfg::SAMPLE_B + (2 * oy) * Cfg::ROW_B + ox * 16;
    }
    constexpr int CBW = Cfg::CBW;
    f32x16 acc[CBW][PT];
#pragma unroll
    for (int c = 0; c < CBW; ++c)
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

    const uint4* wsrc = reinterpret_cast<const uint4*>(wfrag);
    constexpr int WQ = Cfg::WSTEP_B / 16;
    auto wstep_src = [&](int pass, int s) { return wsrc + ((size_t)(pass * Cfg::NKS + s) * (COUT / 32) + cb * CBW) * 128; };
    const bool wmover = tid < WQ;

    for (int pass = 0; pass < Cfg::NPASS; ++pass) {
        uint4 wq[Cfg::NKS];
#pragma unroll
        for (int t = 0; t < Cfg::NKS; ++t) wq[t] = make_uint4(0u, 0u, 0u, 0u);
        if (wmover) {
#pragma unroll
            for (int t = 0; t < Cfg::NKS; ++t) wq[t] = wstep_src(pass, t)[tid];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < Cfg::UITERS; ++k) {
            const int idx = tid + k * NT;
            if (idx < Cfg::UNITS) {
                const int a = idx / (IH * IH), r = idx - a * (IH * IH);
                const int row = r / IH, col = r - row * IH;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (n0 + a < N) {
                    const float4 x0 = raw[k][0], x1 = raw[k][1];
                    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    const float mean = s_mr[2 * a], rstd = s_mr[2 * a + 1];
                    const float4* gb = reinterpret_cast<const float4*>(s_gb + 2 * pass * Cfg::PASS_CH);
                    const float4 q0 = gb[0], q1 = gb[1], q2 = gb[2], q3 = gb[3];
                    const float gam[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
                    const float bet[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sc = rstd * gam[e];                    // scale / shift exactly as the other conv kernels form them
                        v[e] = fmaxf(fmaf(x[e], sc * xscale, (bet[e] - mean * sc) * xscale), 0.f);
                    }
                }
                uint4 p0, p1;
                split_f16x2(v, p0, p1);
                unsigned char* dst = s_in + a * Cfg::SAMPLE_B + row * Cfg::ROW_B + (col & 1) * Cfg::HALF_B + (col >> 1) * 16;
                *reinterpret_cast<uint4*>(dst) = p0;
                *reinterpret_cast<uint4*>(dst + Cfg::PIECE_B) = p1;
            }
        }
        if (wmover) {
            reinterpret_cast<uint4*>(s_w)[tid] = wq[0];
            reinterpret_cast<uint4*>(s_w + Cfg::WSTEP_B)[tid] = wq[1];
        }
        __syncthreads();
        if (pass + 1 < Cfg::NPASS) issue_loads(pass + 1);

        f16x8 fa[2][CBW][2], fb[2][PT][2];
        auto load_frags = [&](int t, int set) {       // 3x3 tap order of conv_bf6_kernel
            int ky, kx;
            if (t < 3) { ky = t; kx = 2 * h; }
            else if (t == 3) { ky = h; kx = 1; }
            else { ky = 2; kx = 1; }
            const int off = ky * Cfg::ROW_B + (kx & 1) * Cfg::HALF_B + (kx >> 1) * 16;
            const unsigned char* wb = s_w + (t % 3) * Cfg::WSTEP_B + lane * 16;
#pragma unroll
            for (int c = 0; c < CBW; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fa[set][c][pl] = *reinterpret_cast<const f16x8*>(wb + (c * 2 + pl) * 1024);
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    fb[set][i][pl] = *reinterpret_cast<const f16x8*>(s_in + pl * Cfg::PIECE_B + lane_base[i] + off);
        };
        load_frags(0, 0);
#pragma unroll
        for (int s = 0; s < Cfg::NKS; ++s) {
            const int cur = s & 1;
            if (s + 1 < Cfg::NKS) load_frags(s + 1, cur ^ 1);
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int c = 0; c < CBW; ++c)
#pragma unroll
                    for (int i = 0; i < PT; ++i)
                        acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c][TA[term]], fb[cur][i][TB[term]], acc[c][i], 0, 0, 0);
            if (s + 1 < Cfg::NKS) {
                constexpr int NRD = 2 * CBW + 2 * PT, NMF = 3 * PT * CBW;
#pragma unroll
                for (int q = 0; q < (NRD < NMF ? NRD : NMF); ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
                if (NRD > NMF) __builtin_amdgcn_sched_group_barrier(0x100, NRD - NMF, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < Cfg::NKS && wmover) reinterpret_cast<uint4*>(s_w + ((s + 2) % 3) * Cfg::WSTEP_B)[tid] = wq[s + 2];
            if (s + 1 < Cfg::NKS) __syncthreads();
        }
    }

    // ---- epilogue: D column = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel within the block ----
    __syncthreads();            // every wave has read its last fragments: the input tile becomes the partial-sum scratch
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int a = pix_a[i], q = pix_q[i];
        const bool valid = a >= 0 && n0 + a < N;
        const int n = n0 + (a < 0 ? 0 : a);
        float fsum = 0.f, fsq = 0.f;
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int co = (cb * CBW + c) * 32 + 8 * rg + 4 * h;
                const float4 bv = *reinterpret_cast<const float4*>(bias + co);
                float4 v;
                v.x = fmaf(acc[c][i][4 * rg + 0], unscale, bv.x);     // unscale = 2^-k exactly: one rounding, like acc + bias
                v.y = fmaf(acc[c][i][4 * rg + 1], unscale, bv.y);
                v.z = fmaf(acc[c][i][4 * rg + 2], unscale, bv.z);
                v.w = fmaf(acc[c][i][4 * rg + 3], unscale, bv.w);
                if (valid) {
                    if (Cfg::OUT_OCT) {
                        *reinterpret_cast<float4*>(out + (((size_t)n * (COUT / 8) + (co >> 3)) * Cfg::PPS + q) * 8 + (co & 7)) = v;
                    } else {
                        float* o = out + ((size_t)n * COUT + co) * Cfg::PPS + q;
                        o[0] = v.x;
                        o[(size_t)Cfg::PPS] = v.y;
                        o[(size_t)2 * Cfg::PPS] = v.z;
                        o[(size_t)3 * Cfg::PPS] = v.w;
                    }
                    fsum += (v.x + v.y) + (v.z + v.w);
                    fsq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fsq))));
                }
            }
        }
        // slot = (linear pixel, channel half): the per-sample reduction below walks them in a fixed order
        const int slot = (32 * (PT * wave + i) + j) * 2 + h;
        s_part[2 * slot] = valid ? fsum : 0.f;
        s_part[2 * slot + 1] = valid ? fsq : 0.f;
    }
    __syncthreads();
    if (tid < S && n0 + tid < N) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < Cfg::PPS; ++q)
            for (int hh = 0; hh < 2; ++hh) {
                const int slot = ((tid * Cfg::PPS + q) * 2 + hh);
                a += (double)s_part[2 * slot];
                b += (double)s_part[2 * slot + 1];
            }
        GNStats& o = st_out[(size_t)(n0 + tid) * Cfg::NPART_OUT + cb];
        o.sum = a;
        o.sq = b;
    }
}

template <class Cfg>
static int launch_bf6s(const float* in, const GNStats* st_in, const float* g, const float* b, const uint32_t* wfrag,
                       const float* bias, float* out, GNStats* st_out, int N, float xscale, float wscale, hipStream_t stream) {
    dim3 grid(((N + Cfg::S - 1) / Cfg::S) * Cfg::CSPLIT);
    static PerDeviceOnce once;
    const int dev = once.device();
    if (!once.is_done(dev)) {
        hipFuncSetAttribute((const void*)conv_bf6s_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        once.set_done(dev);
    }
    hipLaunchKernelGGL(conv_bf6s_kernel<Cfg>, grid, dim3(Cfg::NT), Cfg::LDS_BYTES, stream, in, st_in, g, b, wfrag, bias, out,
                       st_out, N, xscale, 1.0f / (xscale * wscale));
    return 0;
}

typedef BfCfg<16, 32, 5, 125, 61, l1b::NPART, true, 2, 3> Bf2;     // conv2: octet-planar in and out
typedef BfCfg<32, 64, 5, 61, 29, Bf2::NPART_OUT, true, 2, 2, false, 2> Bf3;   // conv3: octet-planar in and out; both 32-channel blocks from one staging
// conv4 (3x3, 29 -> 14): the whole 14 x 14 image is one workgroup tile of 16 x 16 (pixel tiles of 2 rows x 16 columns);
typedef BfCfg<64, 64, 3, 29, 14, Bf3::NPART_OUT, true, 2, 2, true, 2> Bf4;
// small batches (<= CNN_SMALL_BATCH samples): one 32-channel block per workgroup and one pixel tile per wave for conv3 / conv4 --
// four times the workgroups, a quarter of the matrix work per step in each: a launch is then a few per-workgroup latency chains,
// not throughput (8 samples: conv3 23.7 -> 12.4 us, conv4 26.2 -> 15.7 us, profiles/r04_ab_cnn_small_batch.txt).  Same products
// in the same order per output; the GroupNorm moments are float64 partial sums per (tile, block) added in a fixed order, so the
// normalisation agrees to 1e-16.
typedef BfCfg<32, 64, 5, 61, 29, Bf2::NPART_OUT, true, 1, 2, false, 1> Bf3s;
typedef BfCfg<64, 64, 3, 29, 14, Bf3s::NPART_OUT, true, 1, 2, true, 1> Bf4s;
typedef BfsCfg<64, 128, 14, 6, 7, Bf4::NPART_OUT, true, 2, 2> Bfs5;       // conv5: 7 samples (252 pixels) x 32 channels per workgroup
typedef BfsCfg<128, 128, 6, 2, 32, Bfs5::NPART_OUT, false, 1, 2> Bfs6;  // conv6: 32 samples (128 pixels) x 32 channels per workgroup; NCHW out (fc)

// StriveCNN.conv2_plain: the caller runs small latency-bound kernels of ANOTHER stream under the CNN (the rule-based planner
// behind one rollout while the second rollout's CNN runs).  conv2 then stays on conv_bf6_kernel: the persistent conv_ws_kernel
// is faster alone and in the two-stream open loop (17.9 against 18.35 ms) but costs the closed loop 1 ms (22.8 against 21.9 ms;
// leaving 16 or 32 CUs free changes nothing: profiles/r03_ab_conv_ws_closed_loop.json).

// the specialised-wave form (conv_ws_kernel): one persistent workgroup per CU; option conv_ws = 0 keeps conv_bf6_kernel (A/B)
template <class Cfg>
static int launch_ws(const float* in, const GNStats* st_in, const float* g, const float* b, const uint32_t* wfrag,
                     const float* bias, float* out, GNStats* st_out, int N, float xscale, float wscale, hipStream_t stream,
                     unsigned long long* tprof = nullptr) {
    using W = WsCfg<Cfg>;
    static PerDeviceOnce once;
    const int dev = once.device();
    if (!once.is_done(dev)) {
        hipFuncSetAttribute((const void*)conv_ws_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS_BYTES);
        int v = 0;
        if (!(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)) v = 256;
        once.value[dev].store(v, std::memory_order_relaxed);
        once.set_done(dev);
    }
    const int ncu = once.value[dev].load(std::memory_order_relaxed);
    const int total = N * Cfg::TILES_X * Cfg::TILES_Y;
    const int grid = total < ncu ? total : ncu;
    hipLaunchKernelGGL(conv_ws_kernel<Cfg>, dim3(grid), dim3(W::NT), W::LDS_BYTES, stream, in, st_in, g, b, wfrag, bias, out, st_out, N,
                       xscale, 1.0f / (xscale * wscale), strive_tuning().conv_ws_dbg, tprof);
    return 0;
}

template <class Cfg, int NPW = 7>
static int launch_wsx(const float* in, const GNStats* st_in, const float* g, const float* b, const uint32_t* wfrag,
                      const float* bias, float* out, GNStats* st_out, int N, float xscale, float wscale, hipStream_t stream) {
    using W = WxCfg<Cfg, NPW>;
    static PerDeviceOnce once;
    const int dev = once.device();
    if (!once.is_done(dev)) {
        hipFuncSetAttribute((const void*)conv_wsx_kernel<Cfg, NPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS_BYTES);
        int v = 0;
        if (!(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)) v = 256;
        once.value[dev].store(v, std::memory_order_relaxed);
        once.set_done(dev);
    }
    const int ncu = once.value[dev].load(std::memory_order_relaxed);
    const int total = N * Cfg::TILES_X * Cfg::TILES_Y;
    const int grid = total < ncu ? total : ncu;
    const int dbg = strive_tuning().conv_ws_dbg;
    hipLaunchKernelGGL((conv_wsx_kernel<Cfg, NPW>), dim3(grid), dim3(W::NT), W::LDS_BYTES, stream, in, st_in, g, b, wfrag, bias, out, st_out, N,
                       xscale, 1.0f / (xscale * wscale), dbg);
    return 0;
}

template <class Cfg>
static int launch_bf6(const float* in, const GNStats* st_in, const float* g, const float* b, const uint32_t* wfrag,
                      const float* bias, float* out, GNStats* st_out, int N, float xscale, float wscale, hipStream_t stream) {
    dim3 grid(Cfg::TILES_X * Cfg::CSPLIT, Cfg::TILES_Y, (N + 7) / 8 * 8);      // z rounded up: see the id -> (sample, tile) map in the kernel
    static PerDeviceOnce once;
    const int dev = once.device();
    if (!once.is_done(dev)) {
        hipFuncSetAttribute((const void*)conv_bf6_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        once.set_done(dev);
    }
    // (round 5, tools/conv_dephase_probe.py at commit "dephase probe": holding back one of every two co-resident workgroups by a
    // fraction of a staging + matrix period -- by wave slot parity or by id -- changes conv2 / conv3 / conv4 by less than the
    // run-to-run noise, profiles/r05_conv_dephase_probe.txt: the workgroups of a CU do not run in lock step)
    hipLaunchKernelGGL((conv_bf6_kernel<Cfg>), grid, dim3(Cfg::NT), Cfg::LDS_BYTES, stream, in, st_in, g, b, wfrag, bias, out,
                       st_out, N, xscale, 1.0f / (xscale * wscale), (unsigned long long*)nullptr);
    return 0;
}

// per-layer configurations            CIN COUT KS  IH  OH  TH  TW  S  CC NWP NWM NPW MTW NPART_IN

// =============================================================================================
// GroupNorm6 + ReLU + flatten + Linear(512 -> 64): 4 samples per workgroup.
// thread = (output o, k quarter): 128 weights per thread in 4 batches of 32 loads (the weight stream from L2 is the
// critical path: 512 k in batches of 8 cost 64 serial round trips), each weight used for the 4 samples; the four k
// quarters are added through LDS in a fixed order.
// =============================================================================================
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ in, const GNStats* __restrict__ st,
                                                   const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                   const float* __restrict__ wt, const float* __restrict__ bias,
                                                   float* __restrict__ feat, int N) {
    __shared__ __attribute__((aligned(16))) float s_a[4][512];
    __shared__ float s_p[4][4][64];
    __shared__ float s_mr[4][2];
    const int n0 = blockIdx.x * 4, tid = threadIdx.x;
    if (tid < 4) {
        float mean = 0.f, rstd = 0.f;
        if (n0 + tid < N) gn_moments(st, n0 + tid, Bfs6::NPART_OUT, 512.0, mean, rstd);
        s_mr[tid][0] = mean;
        s_mr[tid][1] = rstd;
    }
    __syncthreads();
    for (int i = tid; i < 4 * 512; i += 256) {
        const int s = i >> 9, k = i & 511;
        float v = 0.f;
        if (n0 + s < N) {
            const int c = k >> 2;
            const float sc = s_mr[s][1] * gn_g[c];
            v = fmaxf(fmaf(in[(size_t)(n0 + s) * 512 + k], sc, gn_b[c] - s_mr[s][0] * sc), 0.f);
        }
        s_a[s][k] = v;
    }
    __syncthreads();
    const int o = tid & 63, kq = tid >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
        const int k0 = kq * 128 + kb * 32;
        float w[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q] = wt[(size_t)(k0 + q) * 64 + o];
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 a = *reinterpret_cast<const float4*>(&s_a[s][k0 + 4 * q4]);
                acc[s] = fmaf(a.x, w[4 * q4 + 0], acc[s]);
                acc[s] = fmaf(a.y, w[4 * q4 + 1], acc[s]);
                acc[s] = fmaf(a.z, w[4 * q4 + 2], acc[s]);
                acc[s] = fmaf(a.w, w[4 * q4 + 3], acc[s]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) s_p[kq][s][o] = acc[s];
    __syncthreads();
    const int s = tid >> 6;
    if (n0 + s < N) feat[(size_t)(n0 + s) * 64 + o] = ((s_p[0][s][o] + s_p[1][s][o]) + (s_p[2][s][o] + s_p[3][s][o])) + bias[o];
}

#include "map_cnn_tail.h"

// =============================================================================================
// host side
// =============================================================================================
namespace {
constexpr size_t L_OUT[6] = {16u * 125 * 125, 32u * 61 * 61, 64u * 29 * 29, 64u * 14 * 14, 128u * 6 * 6, 128u * 2 * 2};
constexpr int CNN_CHUNK_MAX = 1024;   // workspace is sized for this many agents per pass
// up to here a launch is a few latency chains: conv1 gives every tile its own workgroup, conv3 / conv4 run Bf3s / Bf4s.  Measured
// (profiles/r04_ab_cnn_small_batch.txt, refine closures): 48 samples -9.5 %, 64 -5.7 %, 96 -4.5 %, 128 -0.6 %, 192 +3 %.
constexpr int CNN_SMALL_BATCH = 96;
// the fused tail takes ONE sample per workgroup up to here (bit-identical either way): -0.2 ms per closure at 192 samples,
// -0.12 at 256, nothing at 512 (where the four-sample form keeps the weight streams per CU lower)
constexpr int CNN_TAIL_ONE_SAMPLE = 256;
// option cnn_small_batch moves the threshold (0: never): the tests and A/B runs exercise both chains in one process
static int cnn_small_batch() { return strive_tuning().cnn_small_batch; }
// agents pushed through the layer stack together (option cnn_chunk)
static int cnn_chunk() {
    int v = strive_tuning().cnn_chunk;
    if (v < 8) v = 8;
    if (v > CNN_CHUNK_MAX) v = CNN_CHUNK_MAX;
    return v;
}
constexpr int NPARTS[6] = {l1b::NPART, Bf2::NPART_OUT, Bf3::NPART_OUT, Bf4::NPART_OUT, Bfs5::NPART_OUT, Bfs6::NPART_OUT};
// slots reserved per layer: the small-batch chain (Bf3s / Bf4s) writes four times as many partial moments for conv3 / conv4
constexpr int NPMAX[6] = {l1b::NPART, Bf2::NPART_OUT, Bf3s::NPART_OUT, Bf4s::NPART_OUT, Bfs5::NPART_OUT, Bfs6::NPART_OUT};
constexpr int STAT_SLOTS = NPMAX[0] + NPMAX[1] + NPMAX[2] + NPMAX[3] + NPMAX[4] + NPMAX[5];
static_assert(Bf3s::NPART_OUT >= Bf3::NPART_OUT && Bf4s::NPART_OUT >= Bf4::NPART_OUT, "reserved statistics slots cover both chains");
// the statistics block of `per` samples carved into the six layers' slot arrays (one carve-up for every user of the workspace)
static inline void stat_slots(GNStats* stats, size_t per, GNStats* (&st)[6]) {
    size_t off = 0;
    for (int l = 0; l < 6; ++l) { st[l] = stats + off; off += per * NPMAX[l]; }
}
static_assert(Bfs5::NPART_IN == Bf4::NPART_OUT &&
              Bfs6::NPART_IN == Bfs5::NPART_OUT, "statistics slot chain");

size_t per_agent_floats() {
    size_t t = 0;
    for (int l = 0; l < 6; ++l) t += L_OUT[l];
    return t;
}
}  // namespace

extern "C" size_t strive_map_cnn_workspace_bytes(int32_t N) {
    const size_t ch = (size_t)(N < CNN_CHUNK_MAX ? (N > 0 ? N : 1) : CNN_CHUNK_MAX);
    size_t bytes = 0;
    for (int l = 0; l < 6; ++l) bytes += strive_align_up(ch * L_OUT[l] * 4, 256);
    bytes += strive_align_up(ch * STAT_SLOTS * sizeof(GNStats), 256);
    return bytes;
}

// ---------------------------------------------------------------------------------------------
// Activations kept for the training backward (strive_map_cnn_fwd_keep -> strive_map_cnn_bwd_kept): the raw outputs of all six
// convolutions and their GroupNorm partial sums of EVERY sample of a training forward (1.76 MB per sample: 2.5 GB for the 1408 crops
// of a 2 x 64-agent rollout of 12 steps -- nothing against 288 GB), so that the backward does not run the layers again (conv1 .. conv4
// were 2.2 of the 28 ms of a training step, conv5 / conv6 0.6: the fused tail writes their raw outputs on its way).  The crop (conv1's
// input for its weight gradient) never exists in HBM in the forward and is gathered again.
// ---------------------------------------------------------------------------------------------
struct CnnKeep {
    float* act[6];
    GNStats* st[6];
};
static inline size_t cnn_keep_bytes(size_t N) {
    size_t b = 256;
    for (int l = 0; l < 6; ++l) b += strive_align_up(N * L_OUT[l] * 4, 256) + strive_align_up(N * NPARTS[l] * sizeof(GNStats), 256);
    return b;
}
static inline bool cnn_keep_carve(void* p, size_t bytes, size_t N, CnnKeep& k) {
    StriveArena ar(p, bytes);
    for (int l = 0; l < 6; ++l) k.act[l] = ar.take<float>(N * L_OUT[l]);
    for (int l = 0; l < 6; ++l) k.st[l] = ar.take<GNStats>(N * NPARTS[l]);
    return ar.ok();
}

static int cnn_run(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pmean, const float* pstd,
                   const int32_t* mapix, const uint8_t* crop, int32_t N, float* feat, void* ws, size_t ws_bytes,
                   hipStream_t stream, bool keep_tail_activations = false, const CnnKeep* keep = nullptr, size_t keep_off = 0) {
    // keep_tail_activations: run conv5, conv6 and the Linear layer as separate kernels that leave their outputs and moments
    // in `ws` (the training backward reads them); otherwise the fused tail kernel (map_cnn_tail.h)
    if (N == 0) return 0;
    if (ws_bytes < strive_map_cnn_workspace_bytes(N)) {
        strive_set_error("map_cnn: workspace too small (%zu < %zu)", ws_bytes, strive_map_cnn_workspace_bytes(N));
        return -1;
    }
    const int ch = N < cnn_chunk() ? N : cnn_chunk();
    StriveArena ar(ws, ws_bytes);
    float* act[6];
    for (int l = 0; l < 6; ++l) act[l] = ar.take<float>((size_t)ch * L_OUT[l]);
    GNStats* stats = ar.take<GNStats>((size_t)ch * STAT_SLOTS);
    if (!ar.ok()) { strive_set_error("map_cnn: workspace arena overflow"); return -1; }
    Float4Host m, s;
    StriveMap mp;
    memset(&mp, 0, sizeof(mp));
    if (map) {
        memcpy(m.v, pmean, 16);
        memcpy(s.v, pstd, 16);
        mp = *map;
    } else {
        memset(&m, 0, sizeof(m));
        memset(&s, 0, sizeof(s));
    }
    const int small_batch = cnn_small_batch();
    const int tail_force = strive_tuning().cnn_tail_s;         // samples per workgroup of the fused tail (A/B option: 1, 2 or 4; 0: by size)
    for (int n0 = 0; n0 < N; n0 += ch) {
        const int n = (N - n0) < ch ? (N - n0) : ch;
        GNStats* st[6];
        stat_slots(stats, (size_t)ch, st);
        if (keep) {
            // every layer writes this chunk's rows of the kept arrays instead of the (reused) workspace (conv5 / conv6: the fused tail,
            // TailKeep); the standard chain only: the kept statistics have NPARTS slots per sample
            for (int l = 0; l < 6; ++l) {
                act[l] = keep->act[l] + (keep_off + (size_t)n0) * L_OUT[l];
                st[l] = keep->st[l] + (keep_off + (size_t)n0) * NPARTS[l];
            }
        }
        TailKeep tk;
        tk.y5 = act[4]; tk.y6 = act[5]; tk.st5 = st[4]; tk.st6 = st[5]; tk.np5 = NPARTS[4]; tk.np6 = NPARTS[5];
        dim3 g1(l1b::TILES_Y, n <= small_batch ? l1b::TILES_X : 1, n);
        // (above 256 samples: TWO samples per workgroup since round 5 -- 256 workgroups for a 512-sample chunk instead of 128 on 256
        // CUs: refine closure 12.08 -> 12.01 ms, three alternations, profiles/r05_ab_sweep_step.json; bit-identical for every S)
        const int tail_s = tail_force ? tail_force : (n <= CNN_TAIL_ONE_SAMPLE ? 1 : 2);
        if (map) {
            hipLaunchKernelGGL(conv1b_kernel<true>, g1, dim3(C1_NT), 0, stream, mp, pos + (size_t)n0 * 4, m, s, mapix + n0,
                               (const uint8_t*)nullptr, cnn->w1_frag, 1.0f / cnn->wscale[0], (const float*)cnn->b[0], act[0], st[0]);
        } else {
            hipLaunchKernelGGL(conv1b_kernel<false>, g1, dim3(C1_NT), 0, stream, mp, (const float*)nullptr, m, s,
                               (const int32_t*)nullptr, crop + (size_t)n0 * 4 * 256 * 256, cnn->w1_frag, 1.0f / cnn->wscale[0],
                               (const float*)cnn->b[0], act[0], st[0]);
        }
        // conv2: specialised producer / consumer waves (bit-identical to conv_bf6_kernel; option conv_ws = 0 switches back)
        const bool conv_ws = strive_tuning().conv_ws != 0;
        if (conv_ws && !cnn->conv2_plain)
            launch_ws<Bf2>(act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], n, cnn->xscale[1], cnn->wscale[1], stream);
        else
            launch_bf6<Bf2>(act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], n, cnn->xscale[1], cnn->wscale[1], stream);
        if (!keep_tail_activations && !keep && n <= small_batch) {
            launch_bf6<Bf3s>(act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], n, cnn->xscale[2], cnn->wscale[2], stream);
            launch_bf6<Bf4s>(act[2], st[2], cnn->gn_g[2], cnn->gn_b[2], cnn->w4_frag, cnn->b[3], act[3], st[3], n, cnn->xscale[3], cnn->wscale[3], stream);
            launch_cnn_tail(cnn, act[3], st[3], Bf4s::NPART_OUT, feat + (size_t)n0 * 64, n, stream, nullptr, tail_s);
            continue;
        }
        // conv3 / conv4: specialised waves with the pass weights double-buffered through LDS (round 6; bit-identical to conv_bf6_kernel,
        // which option conv_wsx = 0 and conv2_plain callers keep: a persistent 12-wave workgroup per CU leaves no room for another
        // stream's small kernels)
        const bool conv_wsx = strive_tuning().conv_wsx != 0;
        if (conv_wsx && !cnn->conv2_plain) {
            launch_wsx<Bf3>(act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], n, cnn->xscale[2], cnn->wscale[2], stream);
            launch_wsx<Bf4>(act[2], st[2], cnn->gn_g[2], cnn->gn_b[2], cnn->w4_frag, cnn->b[3], act[3], st[3], n, cnn->xscale[3], cnn->wscale[3], stream);
        } else {
            launch_bf6<Bf3>(act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], n, cnn->xscale[2], cnn->wscale[2], stream);
            launch_bf6<Bf4>(act[2], st[2], cnn->gn_g[2], cnn->gn_b[2], cnn->w4_frag, cnn->b[3], act[3], st[3], n, cnn->xscale[3], cnn->wscale[3], stream);
        }
        if (!keep_tail_activations) {
            launch_cnn_tail(cnn, act[3], st[3], NPARTS[3], feat + (size_t)n0 * 64, n, stream, nullptr, tail_s, keep ? &tk : nullptr);
            continue;
        }
        launch_bf6s<Bfs5>(act[3], st[3], cnn->gn_g[3], cnn->gn_b[3], cnn->w5_frag, cnn->b[4], act[4], st[4], n, cnn->xscale[4], cnn->wscale[4], stream);
        launch_bf6s<Bfs6>(act[4], st[4], cnn->gn_g[4], cnn->gn_b[4], cnn->w6_frag, cnn->b[5], act[5], st[5], n, cnn->xscale[5], cnn->wscale[5], stream);
        hipLaunchKernelGGL(fc_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, act[5], st[5], cnn->gn_g[5], cnn->gn_b[5],
                           cnn->fc_wt, cnn->fc_b, feat + (size_t)n0 * 64, n);
    }
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_map_cnn_fwd(const StriveMap* map, const StriveCNN* cnn, const float* pos,
                                  const float* pos_mean4_host, const float* pos_std4_host, const int32_t* mapix, int32_t N,
                                  float* feat, void* ws, size_t ws_bytes, strive_stream_t stream) {
    STRIVE_CHECK_ARG(map && cnn && pos && mapix && feat && ws && pos_mean4_host && pos_std4_host, "null argument");
    STRIVE_CHECK_ARG(map->C == 4 && map->L == 256 && map->Wc == 256, "the HIP map CNN supports the default 4x256x256 crop only");
    return cnn_run(map, cnn, pos, pos_mean4_host, pos_std4_host, mapix, nullptr, N, feat, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" size_t strive_map_cnn_keep_bytes(int32_t N) { return cnn_keep_bytes((size_t)(N > 0 ? N : 0)); }

extern "C" int strive_map_cnn_fwd_keep(const StriveMap* map, const StriveCNN* cnn, const float* pos,
                                       const float* pos_mean4_host, const float* pos_std4_host, const int32_t* mapix, int32_t N,
                                       float* feat, void* ws, size_t ws_bytes, void* kept, size_t kept_bytes, int32_t kept_total,
                                       int32_t kept_offset, strive_stream_t stream) {
    STRIVE_CHECK_ARG(map && cnn && pos && mapix && feat && ws && pos_mean4_host && pos_std4_host && kept, "null argument");
    STRIVE_CHECK_ARG(map->C == 4 && map->L == 256 && map->Wc == 256, "the HIP map CNN supports the default 4x256x256 crop only");
    STRIVE_CHECK_ARG(N >= 0 && kept_offset >= 0 && (int64_t)kept_offset + N <= (int64_t)kept_total, "rows outside the kept arrays");
    STRIVE_CHECK_ARG(kept_bytes >= cnn_keep_bytes((size_t)kept_total), "kept-activation buffer too small");
    CnnKeep k;
    STRIVE_CHECK_ARG(cnn_keep_carve(kept, kept_bytes, (size_t)kept_total, k), "kept-activation arena overflow");
    return cnn_run(map, cnn, pos, pos_mean4_host, pos_std4_host, mapix, nullptr, N, feat, ws, ws_bytes, (hipStream_t)stream, false, &k,
                   (size_t)kept_offset);
}

extern "C" int strive_map_cnn_fwd_from_crop(const StriveCNN* cnn, const uint8_t* crop, int32_t N, float* feat, void* ws,
                                            size_t ws_bytes, strive_stream_t stream) {
    STRIVE_CHECK_ARG(cnn && crop && feat && ws, "null argument");
    return cnn_run(nullptr, cnn, nullptr, nullptr, nullptr, nullptr, crop, N, feat, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Measurement hook: launch ONE layer of the stack on activations left in `ws` by a previous
// strive_map_cnn_fwd over the same N (<= 256) poses, so bench.py can time a single kernel with events
// on the launching stream.  layer 0 = fused crop + conv1, 1..3 = conv2..conv4, 7 = the fused tail (conv5 + conv6 + Linear: what
// strive_map_cnn_fwd runs); 4, 5, 6 = the separate conv5 / conv6 / GroupNorm + Linear kernels of the training recompute (4 first:
// 5 and 6 read what the previous one wrote).
// ---------------------------------------------------------------------------------------------
extern "C" int strive_map_cnn_bench_layer(const StriveMap* map, const StriveCNN* cnn, int32_t layer, const float* pos,
                                          const float* pos_mean4_host, const float* pos_std4_host, const int32_t* mapix,
                                          int32_t N, float* feat, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(map && cnn && pos && mapix && feat && ws, "null argument");
    STRIVE_CHECK_ARG(N > 0 && N <= CNN_CHUNK_MAX && layer >= 0 && layer <= 81, "bad layer / N");
    STRIVE_CHECK_ARG(ws_bytes >= strive_map_cnn_workspace_bytes(N), "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    StriveArena ar(ws, ws_bytes);
    float* act[6];
    for (int l = 0; l < 6; ++l) act[l] = ar.take<float>((size_t)N * L_OUT[l]);
    GNStats* stats = ar.take<GNStats>((size_t)N * STAT_SLOTS);
    GNStats* st[6];
    stat_slots(stats, (size_t)N, st);
    Float4Host m, s;
    memcpy(m.v, pos_mean4_host, 16);
    memcpy(s.v, pos_std4_host, 16);
    switch (layer) {
        case 0:
            hipLaunchKernelGGL(conv1b_kernel<true>, dim3(l1b::TILES_Y, 1, N), dim3(C1_NT), 0, stream, *map, pos, m, s,
                               mapix, (const uint8_t*)nullptr, cnn->w1_frag, 1.0f / cnn->wscale[0], (const float*)cnn->b[0], act[0], st[0]);
            break;
        case 21: case 22: {   // phase profile of conv2 / conv3: sums of s_memtime deltas land in `feat` (>= 64 bytes, zeroed here)
            hipMemsetAsync(feat, 0, 64, stream);
            if (layer == 21) {
                hipFuncSetAttribute((const void*)conv_bf6_kernel<Bf2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Bf2::LDS_BYTES);
                hipLaunchKernelGGL((conv_bf6_kernel<Bf2, true>), dim3(Bf2::TILES_X * Bf2::CSPLIT, Bf2::TILES_Y, (N + 7) / 8 * 8), dim3(Bf2::NT), Bf2::LDS_BYTES,
                                   stream, act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], N,
                                   cnn->xscale[1], 1.0f / (cnn->xscale[1] * cnn->wscale[1]), reinterpret_cast<unsigned long long*>(feat));
            } else {
                hipFuncSetAttribute((const void*)conv_bf6_kernel<Bf3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Bf3::LDS_BYTES);
                hipLaunchKernelGGL((conv_bf6_kernel<Bf3, true>), dim3(Bf3::TILES_X * Bf3::CSPLIT, Bf3::TILES_Y, (N + 7) / 8 * 8), dim3(Bf3::NT), Bf3::LDS_BYTES,
                                   stream, act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], N,
                                   cnn->xscale[2], 1.0f / (cnn->xscale[2] * cnn->wscale[2]), reinterpret_cast<unsigned long long*>(feat));
            }
            break;
        }
        case 11: case 12: case 13: case 14: {   // timing probes of the layer-0 kernel (results are NOT valid): no gather / no fp64 / 1/3 MFMA
            dim3 gg(l1b::TILES_Y, 1, N);
            if (layer == 11) hipLaunchKernelGGL((conv1b_kernel<true, 1>), gg, dim3(C1_NT), 0, stream, *map, pos, m, s, mapix, (const uint8_t*)nullptr, cnn->w1_frag, 1.0f / cnn->wscale[0], (const float*)cnn->b[0], act[0], st[0]);
            if (layer == 12) hipLaunchKernelGGL((conv1b_kernel<true, 2>), gg, dim3(C1_NT), 0, stream, *map, pos, m, s, mapix, (const uint8_t*)nullptr, cnn->w1_frag, 1.0f / cnn->wscale[0], (const float*)cnn->b[0], act[0], st[0]);
            if (layer == 14) hipLaunchKernelGGL((conv1b_kernel<true, 4>), gg, dim3(C1_NT), 0, stream, *map, pos, m, s, mapix, (const uint8_t*)nullptr, cnn->w1_frag, 1.0f / cnn->wscale[0], (const float*)cnn->b[0], act[0], st[0]);
            if (layer == 13) hipLaunchKernelGGL((conv1b_kernel<true, 3>), gg, dim3(C1_NT), 0, stream, *map, pos, m, s, mapix, (const uint8_t*)nullptr, cnn->w1_frag, 1.0f / cnn->wscale[0], (const float*)cnn->b[0], act[0], st[0]);
            break;
        }
        case 31: case 32: case 33: case 34: case 41: case 42: case 43: case 44: {   // timing probes of conv2 (3x) / conv3 (4x): DBG 1..4, results invalid
            const int dbg = layer % 10;
#define STRIVE_DBG_LAUNCH(CFG, D, IN, STI, G, B, W, BIAS, OUT, STO, L)                                                                   \
    hipFuncSetAttribute((const void*)conv_bf6_kernel<CFG, false, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CFG::LDS_BYTES);   \
    hipLaunchKernelGGL((conv_bf6_kernel<CFG, false, D>), dim3(CFG::TILES_X * CFG::CSPLIT, CFG::TILES_Y, (N + 7) / 8 * 8), dim3(CFG::NT),  \
                       CFG::LDS_BYTES, stream, IN, STI, G, B, W, BIAS, OUT, STO, N, cnn->xscale[L], 1.0f / (cnn->xscale[L] * cnn->wscale[L]), \
                       (unsigned long long*)nullptr)
            if (layer < 40) {
                if (dbg == 1) { STRIVE_DBG_LAUNCH(Bf2, 1, act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], 1); }
                if (dbg == 2) { STRIVE_DBG_LAUNCH(Bf2, 2, act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], 1); }
                if (dbg == 3) { STRIVE_DBG_LAUNCH(Bf2, 3, act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], 1); }
                if (dbg == 4) { STRIVE_DBG_LAUNCH(Bf2, 4, act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], 1); }
            } else {
                if (dbg == 1) { STRIVE_DBG_LAUNCH(Bf3, 1, act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], 2); }
                if (dbg == 2) { STRIVE_DBG_LAUNCH(Bf3, 2, act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], 2); }
                if (dbg == 3) { STRIVE_DBG_LAUNCH(Bf3, 3, act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], 2); }
                if (dbg == 4) { STRIVE_DBG_LAUNCH(Bf3, 4, act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], 2); }
            }
#undef STRIVE_DBG_LAUNCH
            break;
        }
        case 1: launch_bf6<Bf2>(act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], N, cnn->xscale[1], cnn->wscale[1], stream); break;
        case 51: launch_ws<Bf2>(act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], N, cnn->xscale[1], cnn->wscale[1], stream); break;   // conv2, specialised waves
        case 81: {   // phase profile of the specialised-wave conv2: clock sums of consumer wave 0 / the first producer wave land in `feat`
            unsigned long long* tp = reinterpret_cast<unsigned long long*>(feat);
            hipMemsetAsync(tp, 0, 64, stream);
            launch_ws<Bf2>(act[0], st[0], cnn->gn_g[0], cnn->gn_b[0], cnn->w2_frag, cnn->b[1], act[1], st[1], N, cnn->xscale[1], cnn->wscale[1], stream, tp);
            break;
        }
        case 52: launch_wsx<Bf3>(act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], N, cnn->xscale[2], cnn->wscale[2], stream); break;   // conv3, specialised waves + streamed weights
        case 53: launch_wsx<Bf4>(act[2], st[2], cnn->gn_g[2], cnn->gn_b[2], cnn->w4_frag, cnn->b[3], act[3], st[3], N, cnn->xscale[3], cnn->wscale[3], stream); break;
        case 2: launch_bf6<Bf3>(act[1], st[1], cnn->gn_g[1], cnn->gn_b[1], cnn->w3_frag, cnn->b[2], act[2], st[2], N, cnn->xscale[2], cnn->wscale[2], stream); break;
        case 3: launch_bf6<Bf4>(act[2], st[2], cnn->gn_g[2], cnn->gn_b[2], cnn->w4_frag, cnn->b[3], act[3], st[3], N, cnn->xscale[3], cnn->wscale[3], stream); break;
        case 4: launch_bf6s<Bfs5>(act[3], st[3], cnn->gn_g[3], cnn->gn_b[3], cnn->w5_frag, cnn->b[4], act[4], st[4], N, cnn->xscale[4], cnn->wscale[4], stream); break;
        case 5: launch_bf6s<Bfs6>(act[4], st[4], cnn->gn_g[4], cnn->gn_b[4], cnn->w6_frag, cnn->b[5], act[5], st[5], N, cnn->xscale[5], cnn->wscale[5], stream); break;
        case 7: launch_cnn_tail(cnn, act[3], st[3], NPARTS[3], feat, N, stream, nullptr, N <= CNN_TAIL_ONE_SAMPLE ? 1 : 2); break;   // conv5 + conv6 + Linear, fused: the form strive_map_cnn_fwd launches for N samples
        case 27: {  // its phase profile: clock sums land in `ws` beyond the activations conv4 left (results in feat stay valid)
            unsigned long long* tp = reinterpret_cast<unsigned long long*>(act[4]);
            hipMemsetAsync(tp, 0, 64, stream);
            launch_cnn_tail(cnn, act[3], st[3], NPARTS[3], feat, N, stream, tp);
            hipMemcpyAsync(feat, tp, 64, hipMemcpyDeviceToDevice, stream);
            break;
        }
        default:
            hipLaunchKernelGGL(fc_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, act[5], st[5], cnn->gn_g[5], cnn->gn_b[5],
                               cnn->fc_wt, cnn->fc_b, feat, N);
    }
    STRIVE_CHECK_LAUNCH();
    return 0;
}

#include "map_cnn_bwd.h"
