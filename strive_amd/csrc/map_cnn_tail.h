// conv5 + GroupNorm + ReLU + conv6 + GroupNorm + ReLU + flatten + Linear(512 -> 64) in ONE kernel
// (reference src/models/traffic_model.py:69-87, 437-451: the last two convolution blocks and the output layer of the map CNN).
//
// The three separate kernels were latency chains, not throughput work: conv5 and conv6 walk 40 + 80 barrier-separated
// matrix steps per workgroup with their weights staged through LDS (40 + 56 us for 512 agents on 74 / 32 workgroups), and
// each is a launch plus an HBM round trip of activations that are 18 KB / 2 KB per agent.  Here a workgroup owns S = 4
// whole samples, so both GroupNorm(1) reductions are local, and the roles of the waves are swapped: the four waves split
// the 128 OUTPUT CHANNELS (one 32-channel block each) and share the pixels.  The input tile therefore sits in LDS once for
// all waves, while every wave reads ITS weight fragments straight from global memory into registers (the fragment tables
// are already in lane order: one coalesced 1 KB load per matrix operand), one pass ahead -- no weight staging, and no
// barrier inside conv6 at all.  conv5's output never leaves the chip: it is normalised, split into its two fp16 pieces
// and written to LDS in conv6's input layout; conv6's output feeds the Linear layer from LDS.
//
// Same arithmetic as the unfused kernels (fp16 x 3 products with fp32 accumulation, the same tap order and term order for
// conv5); conv6 keeps one accumulator per product term (three independent MFMA chains instead of one of 240) and adds
// them smallest first.  No atomics: bitwise reproducible.
#pragma once
#include <type_traits>

//
// S = samples per workgroup is a template parameter: 4 is the throughput form; with S = 1 a batch of <= 32 samples (the one-scene
// operating point of the shipped .cfg files) spreads over N workgroups instead of N / 4, each walking 2 pixel tiles per conv5 step
// instead of 5 -- the kernel is a per-workgroup latency chain there (43 us for 8 samples as for 512).  Per sample the arithmetic
// and the order of every sum are the same for every S: the results are bit-identical.
namespace tail {
constexpr int NT = 256, NW = 4, NKS = 5;
// conv5: 64 -> 128 channels, 14 x 14 -> 6 x 6
constexpr int C5I = 64, IH5 = 14, OH5 = 6, PPS5 = OH5 * OH5;
constexpr int HW5 = (IH5 + 1) / 2, HALF5 = HW5 * 16, ROW5 = 2 * HALF5, SAMPLE5 = IH5 * ROW5;
constexpr int NPASS5 = C5I / 8;
// conv6: 128 -> 128 channels, 6 x 6 -> 2 x 2; all 16 channel octets of the input resident
constexpr int C6I = 128, IH6 = 6, OH6 = 2, PPS6 = OH6 * OH6;
constexpr int HW6 = (IH6 + 1) / 2, HALF6 = HW6 * 16, ROW6 = 2 * HALF6, SAMPLE6 = IH6 * ROW6;
constexpr int NPASS6 = C6I / 8;
constexpr int COUT = 128;
constexpr int GB_B = (C5I + COUT + COUT) * 8;

template <int S_>
struct Cfg {
    static constexpr int S = S_;
    static constexpr int NPIX5 = S * PPS5, PT5 = (NPIX5 + 31) / 32;
    static constexpr int PIECE5 = S * SAMPLE5, IN5_B = 2 * PIECE5;
    static constexpr int UNITS5 = S * IH5 * IH5, UIT5 = (UNITS5 + NT - 1) / NT;
    static constexpr int NPIX6 = S * PPS6;
    static constexpr int PIECE6 = S * SAMPLE6, OCT6_B = 2 * PIECE6, IN6_B = NPASS6 * OCT6_B;
    // LDS: [tile: max(two conv5 input buffers, conv6 input)] [partials] [fc input] [fc partial sums] [gamma/beta x3] [moments x3]
    static constexpr int TILE_B = (2 * IN5_B > IN6_B ? 2 * IN5_B : IN6_B);
    static constexpr int PART_B = PT5 * 32 * NW * 2 * 8;                  // (pixel, wave, half) -> (sum, sum of squares)
    static constexpr int Y_B = S * 512 * 4, FCP_B = 4 * S * 64 * 4;
    static constexpr int MR_B = 3 * S * 8;
    static constexpr size_t LDS_BYTES = (size_t)TILE_B + PART_B + Y_B + FCP_B + GB_B + MR_B + 64;
    static_assert(NPIX6 <= 32 && S >= 1 && S <= NW && PPS6 * NW * 2 <= 64, "conv6 fills (part of) one 32-pixel tile; wave s reduces sample s");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(IN5_B % 16 == 0 && OCT6_B % 16 == 0 && TILE_B % 16 == 0, "16-byte fragment reads");
};
}  // namespace tail

// window offset of matrix step t for lane half h (3x3 tap order of conv_bf6_kernel / conv_bf6s_kernel)
template <int ROW_B, int HALF_B>
__device__ __forceinline__ int tail_tap_offset(int t, int h) {
    int ky, kx;
    if (t < 3) { ky = t; kx = 2 * h; }
    else if (t == 3) { ky = h; kx = 1; }
    else { ky = 2; kx = 1; }
    return ky * ROW_B + (kx & 1) * HALF_B + (kx >> 1) * 16;
}

// 4 fp32 values (pre-scaled into fp16's range) -> their two fp16 pieces, 8 bytes each
__device__ __forceinline__ void split_f16x4(const float v[4], uint2& p0, uint2& p1) {
    uint32_t hh[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 a = (_Float16)v[i];
        const float r = v[i] - (float)a;
        const _Float16 c = (_Float16)r;
        uint16_t ab, cb;
        __builtin_memcpy(&ab, &a, 2);
        __builtin_memcpy(&cb, &c, 2);
        hh[i] = ab;
        ll[i] = cb;
    }
    p0 = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
    p1 = make_uint2(ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16));
}

struct TailArgs {
    const float* in;            // conv4 output, octet-planar (N, 8, 14*14, 8), pre-GroupNorm
    const GNStats* st_in;       // its per-tile partial moments, npart_in per sample
    int npart_in;
    const float *g4, *b4;       // GroupNorm of conv4's output (64)
    const uint32_t* w5;         // conv5 fragments [pass 8][step 5][block 4][piece 2][lane 64][16 B]
    const float* bias5;
    const float *g5, *b5;       // GroupNorm of conv5's output (128)
    const uint32_t* w6;         // conv6 fragments [pass 16][step 5][block 4][piece 2][lane 64][16 B]
    const float* bias6;
    const float *g6, *b6;       // GroupNorm of conv6's output (128)
    const float* fc_wt;         // (512, 64)
    const float* fc_b;
    float xs5, un5, xs6, un6;   // operand pre-scales (powers of two) and the matching result scales
    float* feat;                // (N, 64)
    int N;
    // optional (training forward, CnnKeep): the raw outputs of conv5 (octet-planar (N, 16, 36, 8)) and conv6 (NCHW (N, 128, 2, 2)) and
    // their GroupNorm sums in slot 0 of np5 / np6 slots per sample (the other slots zero) -- what conv_bf6s_kernel leaves for the backward
    float* y5;
    float* y6;
    GNStats* st5;
    GNStats* st6;
    int np5, np6;
};

// TIMING: clock64 stamps of the phases summed over workgroups into `tprof` (measurement hook only)
template <int S_, bool TIMING = false>
__global__ __launch_bounds__(tail::NT, 1) void cnn_tail_kernel(TailArgs A, unsigned long long* __restrict__ tprof = nullptr) {
    using namespace tail;
    using C = tail::Cfg<S_>;
    constexpr int S = C::S, NPIX5 = C::NPIX5, PT5 = C::PT5, PIECE5 = C::PIECE5, IN5_B = C::IN5_B, UNITS5 = C::UNITS5, UIT5 = C::UIT5;
    constexpr int NPIX6 = C::NPIX6, PIECE6 = C::PIECE6, OCT6_B = C::OCT6_B, TILE_B = C::TILE_B, PART_B = C::PART_B;
    long long tstamp[8];
    int nstamp = 0;
    auto stamp = [&]() { if (TIMING) tstamp[nstamp++] = clock64(); };
    stamp();
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* s_tile = reinterpret_cast<unsigned char*>(smem);
    float* s_part = reinterpret_cast<float*>(s_tile + TILE_B);                 // [pixel][wave][half][2]
    float* s_y = reinterpret_cast<float*>(s_tile + TILE_B + PART_B);           // [S][512]
    float* s_p = s_y + S * 512;                                                // [4][S][64]
    float* s_gb4 = s_p + 4 * S * 64;                                           // [64][2]
    float* s_gb5 = s_gb4 + 2 * C5I;                                            // [128][2]
    float* s_gb6 = s_gb5 + 2 * COUT;                                           // [128][2]
    float* s_mr4 = s_gb6 + 2 * COUT;                                           // [S][2] mean, rstd of conv4's output
    float* s_mr5 = s_mr4 + 2 * S;
    float* s_mr6 = s_mr5 + 2 * S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int n0 = blockIdx.x * S;
    const int N = A.N;

    // ---- raw input loads of pass 0 and the first weight fragments go out first ----
    float4 raw[UIT5][2];
    auto issue_loads = [&](int pass) {
#pragma unroll
        for (int k = 0; k < UIT5; ++k) {
            const int idx = tid + k * NT;
            raw[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            raw[k][1] = raw[k][0];
            if (idx < UNITS5) {
                const int a = idx / (IH5 * IH5), r = idx - a * (IH5 * IH5);
                if (n0 + a < N) {
                    const float4* src = reinterpret_cast<const float4*>(A.in + ((((size_t)(n0 + a) * (C5I / 8) + pass) * IH5 * IH5) + r) * 8);
                    raw[k][0] = src[0];
                    raw[k][1] = src[1];
                }
            }
        }
    };
    const uint4* w5 = reinterpret_cast<const uint4*>(A.w5);
    const uint4* w6 = reinterpret_cast<const uint4*>(A.w6);
    auto load_w = [&](const uint4* base, int pass, uint4 (&wq)[NKS][2]) {
#pragma unroll
        for (int t = 0; t < NKS; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wq[t][pl] = base[((size_t)(pass * NKS + t) * 4 + wave) * 128 + pl * 64 + lane];
    };
    issue_loads(0);
    uint4 wq[NKS][2], wqn[NKS][2];
    load_w(w5, 0, wq);

    if (tid < S) {
        float mean = 0.f, rstd = 0.f;
        if (n0 + tid < N) gn_moments(A.st_in, n0 + tid, A.npart_in, (double)C5I * IH5 * IH5, mean, rstd);
        s_mr4[2 * tid] = mean;
        s_mr4[2 * tid + 1] = rstd;
    }
    for (int c = tid; c < C5I; c += NT) { s_gb4[2 * c] = A.g4[c]; s_gb4[2 * c + 1] = A.b4[c]; }
    for (int c = tid; c < COUT; c += NT) {
        s_gb5[2 * c] = A.g5[c]; s_gb5[2 * c + 1] = A.b5[c];
        s_gb6[2 * c] = A.g6[c]; s_gb6[2 * c + 1] = A.b6[c];
    }
    __syncthreads();

    // one staging unit = 8 channels of one input pixel: GroupNorm + ReLU + pre-scale, split into the two fp16 pieces, 2 x 16 B
    // into the tile.  Units 0 .. UFULL-1 exist for every thread (no branch: they are interleaved with the matrix steps);
    // samples beyond N stage finite filler (their loads were skipped) that nothing ever stores.
    constexpr int UFULL = UNITS5 / NT;
    auto stage_unit = [&](int k, int pass, unsigned char* buf) {
        const int idx = tid + k * NT;
        const int a = idx / (IH5 * IH5), r = idx - a * (IH5 * IH5);
        const int row = r / IH5, col = r - row * IH5;
        const float4 x0 = raw[k][0], x1 = raw[k][1];
        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float mean = s_mr4[2 * a], rstd = s_mr4[2 * a + 1];
        const float4* gb = reinterpret_cast<const float4*>(s_gb4 + 2 * pass * 8);
        const float4 q0 = gb[0], q1 = gb[1], q2 = gb[2], q3 = gb[3];
        const float gam[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
        const float bet[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = rstd * gam[e];                    // scale / shift exactly as the other conv kernels form them
            v[e] = fmaxf(fmaf(x[e], sc * A.xs5, (bet[e] - mean * sc) * A.xs5), 0.f);
        }
        uint4 p0, p1;
        split_f16x2(v, p0, p1);
        unsigned char* dst = buf + a * SAMPLE5 + row * ROW5 + (col & 1) * HALF5 + (col >> 1) * 16;
        *reinterpret_cast<uint4*>(dst) = p0;
        *reinterpret_cast<uint4*>(dst + PIECE5) = p1;
    };
    auto stage_rest = [&](int pass, unsigned char* buf) {
#pragma unroll
        for (int k = UFULL; k < UIT5; ++k)
            if (tid + k * NT < UNITS5) stage_unit(k, pass, buf);
    };
#pragma unroll
    for (int k = 0; k < UFULL; ++k) stage_unit(k, 0, s_tile);
    stage_rest(0, s_tile);
    issue_loads(1);

    // ---- this lane's conv5 pixels: tile i holds linear pixels 32 i + j of the workgroup's S samples ----
    int base5[PT5];
#pragma unroll
    for (int i = 0; i < PT5; ++i) {
        int g = 32 * i + j;
        g = g < NPIX5 ? g : 0;
        const int a = g / PPS5, q = g - a * PPS5;
        const int oy = q / OH5, ox = q - oy * OH5;
        base5[i] = a * SAMPLE5 + (2 * oy) * ROW5 + ox * 16;
    }
    f32x16 acc[PT5];
#pragma unroll
    for (int i = 0; i < PT5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    stamp();

    // ================= conv5: 8 passes of 8 input channels, one barrier per pass =================
    // One wave per SIMD: nothing else hides latencies, so the staging of the NEXT pass (VALU) and the fragment reads of the
    // next step (LDS) are interleaved with the matrix instructions of the current step by hand.
    f16x8 fb[2][PT5][2];
    auto load_fb = [&](const unsigned char* buf, int t, int set) {
        const int off = tail_tap_offset<ROW5, HALF5>(t, h);
#pragma unroll
        for (int i = 0; i < PT5; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[set][i][pl] = *reinterpret_cast<const f16x8*>(buf + pl * PIECE5 + base5[i] + off);
    };
    auto conv5_pass = [&](int pass, auto stage_next) {
        constexpr bool STAGE = decltype(stage_next)::value;
        const unsigned char* buf = s_tile + (pass & 1) * IN5_B;
        unsigned char* nbuf = s_tile + ((pass + 1) & 1) * IN5_B;
        load_fb(buf, 0, 0);
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
            const int cur = t & 1;
            if (t + 1 < NKS) load_fb(buf, t + 1, cur ^ 1);
            f16x8 fa[2];
            __builtin_memcpy(&fa[0], &wq[t][0], 16);
            __builtin_memcpy(&fa[1], &wq[t][1], 16);
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < PT5; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[TA[term]], fb[cur][i][TB[term]], acc[i], 0, 0, 0);
            const bool staged = STAGE && t < UFULL;
            if (staged) stage_unit(t, pass + 1, nbuf);
            // per matrix instruction: one fragment read of the next step and a slice of the staging arithmetic
            constexpr int NMF = 3 * PT5, NRD = 2 * PT5;
#pragma unroll
            for (int q = 0; q < NMF; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (t + 1 < NKS && q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (staged) __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STAGE) stage_rest(pass + 1, nbuf);
    };
#pragma unroll 1
    for (int pass = 0; pass < NPASS5; ++pass) {
        if (pass + 1 < NPASS5) load_w(w5, pass + 1, wqn);
        else load_w(w6, 0, wqn);                                    // conv6's first pass rides on conv5's last
        if (pass + 1 < NPASS5) {
            conv5_pass(pass, std::true_type());
            if (pass + 2 < NPASS5) issue_loads(pass + 2);
        } else {
            conv5_pass(pass, std::false_type());
        }
        __syncthreads();          // the next buffer is complete, and nobody reads this one any more
#pragma unroll
        for (int t = 0; t < NKS; ++t) { wq[t][0] = wqn[t][0]; wq[t][1] = wqn[t][1]; }
    }
    stamp();
    // wq now holds conv6's pass 0; start pass 1 as well (the epilogue below hides the latency)
    load_w(w6, 1, wqn);

    // ---- conv5 epilogue: bias, GroupNorm(1) over each sample, ReLU, fp16 split, into conv6's input layout ----
    // D column = lane&31 = pixel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel within the wave's block
#pragma unroll
    for (int i = 0; i < PT5; ++i) {
        const int g = 32 * i + j;
        const int a = g / PPS5;
        const bool valid = g < NPIX5 && n0 + a < N;
        float fsum = 0.f, fsq = 0.f;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int co = wave * 32 + 8 * rg + 4 * h;
            const float4 bv = *reinterpret_cast<const float4*>(A.bias5 + co);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = fmaf(acc[i][4 * rg + k], A.un5, bb[k]);     // un5 = 2^-k exactly: one rounding, like acc + bias
                acc[i][4 * rg + k] = v;
                fsum += v;
                fsq = fmaf(v, v, fsq);
            }
            if (A.y5 && valid) {
                const int q = g - a * PPS5;
                *reinterpret_cast<float4*>(A.y5 + ((((size_t)(n0 + a) * (COUT / 8) + (co >> 3)) * PPS5) + q) * 8 + (co & 7)) =
                    make_float4(acc[i][4 * rg], acc[i][4 * rg + 1], acc[i][4 * rg + 2], acc[i][4 * rg + 3]);
            }
        }
        float2* o = reinterpret_cast<float2*>(s_part) + ((size_t)g * NW + wave) * 2 + h;
        *o = make_float2(valid ? fsum : 0.f, valid ? fsq : 0.f);
    }
    __syncthreads();              // (also: every wave is done with the conv5 input buffers)
    if (wave < S) {   // wave = sample: lanes stride over its partials, then a shuffle tree -- a fixed order, like everything else
        double a = 0.0, b = 0.0;
        const float2* p = reinterpret_cast<const float2*>(s_part) + (size_t)wave * PPS5 * NW * 2;
        for (int q = lane; q < PPS5 * NW * 2; q += 64) { a += (double)p[q].x; b += (double)p[q].y; }
        a = wave_sum_d(a);
        b = wave_sum_d(b);
        if (lane == 0) {
            const double cnt = (double)COUT * PPS5;
            const double m = a / cnt;
            double var = b / cnt - m * m;
            var = var < 0.0 ? 0.0 : var;
            s_mr5[2 * wave] = (float)m;
            s_mr5[2 * wave + 1] = (float)(1.0 / sqrt(var + GN_EPS));
            if (A.st5 && n0 + wave < N)
                for (int p = 0; p < A.np5; ++p) {
                    GNStats& o = A.st5[(size_t)(n0 + wave) * A.np5 + p];
                    o.sum = p == 0 ? a : 0.0;
                    o.sq = p == 0 ? b : 0.0;
                }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PT5; ++i) {
        const int g = 32 * i + j;
        if (g < NPIX5) {
            const int a = g / PPS5, q = g - a * PPS5;
            const int row = q / OH5, col = q - row * OH5;
            const float mean = s_mr5[2 * a], rstd = s_mr5[2 * a + 1];
            unsigned char* dst = s_tile + a * SAMPLE6 + row * ROW6 + (col & 1) * HALF6 + (col >> 1) * 16 + 8 * h;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int co = wave * 32 + 8 * rg + 4 * h;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sc = rstd * s_gb5[2 * (co + k)];
                    v[k] = fmaxf(fmaf(acc[i][4 * rg + k], sc * A.xs6, (s_gb5[2 * (co + k) + 1] - mean * sc) * A.xs6), 0.f);
                }
                uint2 p0, p1;
                split_f16x4(v, p0, p1);
                unsigned char* d = dst + (wave * 4 + rg) * OCT6_B;
                *reinterpret_cast<uint2*>(d) = p0;
                *reinterpret_cast<uint2*>(d + PIECE6) = p1;
            }
        }
    }
    __syncthreads();
    stamp();

    // ================= conv6: 16 resident passes, no barrier; weights two passes ahead =================
    int base6;
    {
        const int g = j < NPIX6 ? j : 0;
        const int a = g / PPS6, q = g - a * PPS6;
        const int oy = q / OH6, ox = q - oy * OH6;
        base6 = a * SAMPLE6 + (2 * oy) * ROW6 + ox * 16;
    }
    f32x16 c_hl, c_lh, c_hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c_hl[r] = 0.f; c_lh[r] = 0.f; c_hh[r] = 0.f; }
    uint4 wqf[NKS][2];
#pragma unroll 1
    for (int pass = 0; pass < NPASS6; ++pass) {
        if (pass + 2 < NPASS6) load_w(w6, pass + 2, wqf);
        const unsigned char* buf = s_tile + pass * OCT6_B + base6;
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
            const int off = tail_tap_offset<ROW6, HALF6>(t, h);
            f16x8 fa0, fa1;
            __builtin_memcpy(&fa0, &wq[t][0], 16);
            __builtin_memcpy(&fa1, &wq[t][1], 16);
            const f16x8 fb0 = *reinterpret_cast<const f16x8*>(buf + off);
            const f16x8 fb1 = *reinterpret_cast<const f16x8*>(buf + PIECE6 + off);
            c_lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb0, c_lh, 0, 0, 0);
            c_hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb1, c_hl, 0, 0, 0);
            c_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb0, c_hh, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
            wq[t][0] = wqn[t][0]; wq[t][1] = wqn[t][1];
            wqn[t][0] = wqf[t][0]; wqn[t][1] = wqf[t][1];
        }
    }

    stamp();
    // ---- conv6 epilogue: bias, GroupNorm(1), ReLU -> the Linear layer's input in NCHW order (k = 4 channel + pixel) ----
    {
        const int a = j / PPS6, q = j - a * PPS6;
        const bool valid = j < NPIX6 && n0 + a < N;
        float fsum = 0.f, fsq = 0.f;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int co = wave * 32 + 8 * rg + 4 * h;
            const float4 bv = *reinterpret_cast<const float4*>(A.bias6 + co);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float s3 = (c_lh[4 * rg + k] + c_hl[4 * rg + k]) + c_hh[4 * rg + k];
                const float v = fmaf(s3, A.un6, bb[k]);
                c_hh[4 * rg + k] = v;
                fsum += v;
                fsq = fmaf(v, v, fsq);
                if (A.y6 && valid) A.y6[((size_t)(n0 + a) * COUT + co + k) * PPS6 + q] = v;
            }
        }
        float2* o = reinterpret_cast<float2*>(s_part) + ((size_t)j * NW + wave) * 2 + h;
        *o = make_float2(valid ? fsum : 0.f, valid ? fsq : 0.f);
        __syncthreads();
        if (wave < S) {
            double sa = 0.0, sb = 0.0;
            const float2* p = reinterpret_cast<const float2*>(s_part) + (size_t)wave * PPS6 * NW * 2;
            if (lane < PPS6 * NW * 2) { sa = (double)p[lane].x; sb = (double)p[lane].y; }
            sa = wave_sum_d(sa);
            sb = wave_sum_d(sb);
            if (lane == 0) {
                const double cnt = (double)COUT * PPS6;
                const double m = sa / cnt;
                double var = sb / cnt - m * m;
                var = var < 0.0 ? 0.0 : var;
                s_mr6[2 * wave] = (float)m;
                s_mr6[2 * wave + 1] = (float)(1.0 / sqrt(var + GN_EPS));
                if (A.st6 && n0 + wave < N)
                    for (int p = 0; p < A.np6; ++p) {
                        GNStats& o = A.st6[(size_t)(n0 + wave) * A.np6 + p];
                        o.sum = p == 0 ? sa : 0.0;
                        o.sq = p == 0 ? sb : 0.0;
                    }
            }
        }
        __syncthreads();
        if (j < NPIX6) {
            const float mean = s_mr6[2 * a], rstd = s_mr6[2 * a + 1];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int co = wave * 32 + 8 * rg + 4 * h;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sc = rstd * s_gb6[2 * (co + k)];                                   // as fc_kernel forms it
                    const float y = fmaxf(fmaf(c_hh[4 * rg + k], sc, s_gb6[2 * (co + k) + 1] - mean * sc), 0.f);
                    s_y[a * 512 + (co + k) * PPS6 + q] = valid ? y : 0.f;
                }
            }
        }
        __syncthreads();
    }

    stamp();
    // ================= Linear(512 -> 64): thread = (output o, k quarter), 4 samples (as fc_kernel) =================
    const int o = tid & 63, kq = tid >> 6;
    float fc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) fc[s] = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
        const int k0 = kq * 128 + kb * 32;
        float w[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q] = A.fc_wt[(size_t)(k0 + q) * 64 + o];
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_y[s * 512 + k0 + 4 * q4]);
                fc[s] = fmaf(a4.x, w[4 * q4 + 0], fc[s]);
                fc[s] = fmaf(a4.y, w[4 * q4 + 1], fc[s]);
                fc[s] = fmaf(a4.z, w[4 * q4 + 2], fc[s]);
                fc[s] = fmaf(a4.w, w[4 * q4 + 3], fc[s]);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) s_p[(kq * S + s) * 64 + o] = fc[s];
    __syncthreads();
    const int s = tid >> 6;
    if (s < S && n0 + s < N)
        A.feat[(size_t)(n0 + s) * 64 + o] = ((s_p[(0 * S + s) * 64 + o] + s_p[(1 * S + s) * 64 + o]) +
                                             (s_p[(2 * S + s) * 64 + o] + s_p[(3 * S + s) * 64 + o])) + A.fc_b[o];
    stamp();
    if (TIMING && tid == 0) {
        // [0] = workgroups, [1 + k] = sum of (stamp k+1 - stamp k): prologue (moments, first staging), conv5 passes, conv5
        // epilogue (GroupNorm + split), conv6 passes, conv6 epilogue, Linear
        atomicAdd(tprof, 1ull);
        for (int k = 0; k + 1 < nstamp; ++k) atomicAdd(tprof + 1 + k, (unsigned long long)(tstamp[k + 1] - tstamp[k]));
    }
}

template <int S_>
static void launch_cnn_tail_s(const TailArgs& a, int N, hipStream_t stream, unsigned long long* tprof) {
    using C = tail::Cfg<S_>;
    static PerDeviceOnce once;
    const int dev_ = once.device();
    if (!once.is_done(dev_)) {
        (void)hipFuncSetAttribute((const void*)cnn_tail_kernel<S_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)cnn_tail_kernel<S_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        once.set_done(dev_);
    }
    const dim3 grid((N + S_ - 1) / S_);
    if (tprof) hipLaunchKernelGGL((cnn_tail_kernel<S_, true>), grid, dim3(tail::NT), C::LDS_BYTES, stream, a, tprof);
    else hipLaunchKernelGGL((cnn_tail_kernel<S_, false>), grid, dim3(tail::NT), C::LDS_BYTES, stream, a, (unsigned long long*)nullptr);
}

// samples_per_wg: 4 = throughput form; 1 (or 2) for small batches -- see the note on S above
struct TailKeep {          // (see TailArgs::y5 ..)
    float* y5;
    float* y6;
    GNStats* st5;
    GNStats* st6;
    int np5, np6;
};
static int launch_cnn_tail(const StriveCNN* cnn, const float* act4, const GNStats* st4, int npart_in, float* feat, int N,
                           hipStream_t stream, unsigned long long* tprof = nullptr, int samples_per_wg = 4,
                           const TailKeep* keep = nullptr) {
    TailArgs a;
    a.y5 = keep ? keep->y5 : nullptr; a.y6 = keep ? keep->y6 : nullptr;
    a.st5 = keep ? keep->st5 : nullptr; a.st6 = keep ? keep->st6 : nullptr;
    a.np5 = keep ? keep->np5 : 0; a.np6 = keep ? keep->np6 : 0;
    a.in = act4; a.st_in = st4; a.npart_in = npart_in;
    a.g4 = cnn->gn_g[3]; a.b4 = cnn->gn_b[3];
    a.w5 = cnn->w5_frag; a.bias5 = cnn->b[4]; a.g5 = cnn->gn_g[4]; a.b5 = cnn->gn_b[4];
    a.w6 = cnn->w6_frag; a.bias6 = cnn->b[5]; a.g6 = cnn->gn_g[5]; a.b6 = cnn->gn_b[5];
    a.fc_wt = cnn->fc_wt; a.fc_b = cnn->fc_b;
    a.xs5 = cnn->xscale[4]; a.un5 = 1.0f / (cnn->xscale[4] * cnn->wscale[4]);
    a.xs6 = cnn->xscale[5]; a.un6 = 1.0f / (cnn->xscale[5] * cnn->wscale[5]);
    a.feat = feat; a.N = N;
    switch (samples_per_wg) {
        case 1: launch_cnn_tail_s<1>(a, N, stream, tprof); break;
        case 2: launch_cnn_tail_s<2>(a, N, stream, tprof); break;
        default: launch_cnn_tail_s<4>(a, N, stream, tprof); break;
    }
    return 0;
}
