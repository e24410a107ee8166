// Probe: latency of a chain of 128x128 dense layers on 4 rows per workgroup (the shape of the per-node GNN kernels):
//   A  fp32 VALU, weights streamed k-major from L2 in batches of 16 loads (mlp_dev.h dense_lds)
//   B  fp16 x 3 on the matrix cores (v_mfma_f32_16x16x32_f16), weight fragments straight from global memory
//   C  B + the next layer's fragments requested before the current layer is computed
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I strive_amd/csrc tools/dense_probe.hip -o gpurun_out/dense_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../strive_amd/csrc/mlp_dev.h"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NL 10
#define H 128
#define XLD 132

__global__ __launch_bounds__(256) void chain_valu(const float* __restrict__ wt, const float* __restrict__ x, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float s_a[2][RB_NODE * XLD];
    const int tid = threadIdx.x;
    for (int i = tid; i < RB_NODE * H; i += 256) s_a[0][(i / H) * XLD + (i % H)] = x[(size_t)blockIdx.x * RB_NODE * H + i];
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < NL; ++l) {
        dense_lds<RB_NODE, false>(s_a[cur], XLD, H, wt + (size_t)l * H * H, H, nullptr, s_a[cur ^ 1], XLD, H, tid, 256);
        __syncthreads();
        for (int i = tid; i < RB_NODE * H; i += 256) { float& v = s_a[cur ^ 1][(i / H) * XLD + (i % H)]; v = fmaxf(v, 0.f) + 0.01f; }
        __syncthreads();
        cur ^= 1;
    }
    for (int i = tid; i < RB_NODE * H; i += 256) y[(size_t)blockIdx.x * RB_NODE * H + i] = s_a[cur][(i / H) * XLD + (i % H)];
}

// fragments: [layer][ntile 8][kstep 4][piece 2][lane 64][8 x fp16]: lane l = channel 16 nt + (l & 15), k = 32 ks + 8 (l >> 4) + j
#define FRAG_L (8 * 4 * 2 * 64)      // uint4 per layer
#define BROW 272                     // bytes per activation row and piece in LDS: 128 fp16 + 16 pad
template <bool PREFETCH>
__global__ __launch_bounds__(256) void chain_mfma(const uint4* __restrict__ wf, const float* __restrict__ x, float* __restrict__ y,
                                                    float wscale) {
    __shared__ __attribute__((aligned(16))) float s_a[RB_NODE * XLD];
    __shared__ __attribute__((aligned(16))) unsigned char s_b[2 * 16 * BROW];      // [piece][row 16][k 128] fp16; rows >= RB stay zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * BROW / 4; i += 256) reinterpret_cast<uint32_t*>(s_b)[i] = 0u;
    for (int i = tid; i < RB_NODE * H; i += 256) s_a[(i / H) * XLD + (i % H)] = x[(size_t)blockIdx.x * RB_NODE * H + i];
    __syncthreads();
    uint4 wq[2][4][2], wn[2][4][2];
    auto load_w = [&](int l, uint4 (&q)[2][4][2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    q[t][ks][pl] = wf[(size_t)l * FRAG_L + ((((2 * wave + t) * 4 + ks) * 2 + pl) * 64) + lane];
    };
    if (PREFETCH) load_w(0, wq);
    for (int l = 0; l < NL; ++l) {
        if (PREFETCH) { if (l + 1 < NL) load_w(l + 1, wn); }
        else load_w(l, wq);
        // split the 4 x 128 activations into their fp16 pieces: thread = (row, pair of k)
        {
            const int r = tid >> 6, k2 = (tid & 63) * 2;
            const float v0 = s_a[r * XLD + k2], v1 = s_a[r * XLD + k2 + 1];
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            uint16_t a0, a1, b0, b1;
            __builtin_memcpy(&a0, &h0, 2); __builtin_memcpy(&a1, &h1, 2); __builtin_memcpy(&b0, &l0, 2); __builtin_memcpy(&b1, &l1, 2);
            *reinterpret_cast<uint32_t*>(s_b + r * BROW + k2 * 2) = a0 | ((uint32_t)a1 << 16);
            *reinterpret_cast<uint32_t*>(s_b + 16 * BROW + r * BROW + k2 * 2) = b0 | ((uint32_t)b1 << 16);
        }
        __syncthreads();
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned char* bp = s_b + (lane & 15) * BROW + ks * 64 + (lane >> 4) * 16;
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(bp), b1 = *reinterpret_cast<const f16x8*>(bp + 16 * BROW);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f16x8 a0, a1;
                __builtin_memcpy(&a0, &wq[t][ks][0], 16);
                __builtin_memcpy(&a1, &wq[t][ks][1], 16);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[t], 0, 0, 0);
            }
        }
        // D[channel 4 (lane >> 4) + r][row lane & 15]
        if ((lane & 15) < RB_NODE) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 16 * (2 * wave + t) + 4 * (lane >> 4) + r;
                    s_a[(lane & 15) * XLD + ch] = fmaxf(acc[t][r] * wscale, 0.f) + 0.01f;
                }
        }
        __syncthreads();
        if (PREFETCH) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { wq[t][ks][0] = wn[t][ks][0]; wq[t][ks][1] = wn[t][ks][1]; }
        }
    }
    for (int i = tid; i < RB_NODE * H; i += 256) y[(size_t)blockIdx.x * RB_NODE * H + i] = s_a[(i / H) * XLD + (i % H)];
}

static uint16_t f2h(float v) { _Float16 h = (_Float16)v; uint16_t b; memcpy(&b, &h, 2); return b; }
static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }

int main() {
    const float WS = 4096.f;
    std::vector<float> w((size_t)NL * H * H), wt((size_t)NL * H * H);
    srand(1);
    for (auto& v : w) v = ((rand() % 2001) - 1000) * 1e-4f;          // W[l][c][k]
    for (int l = 0; l < NL; ++l)
        for (int c = 0; c < H; ++c)
            for (int k = 0; k < H; ++k) wt[((size_t)l * H + k) * H + c] = w[((size_t)l * H + c) * H + k];
    std::vector<uint16_t> fr((size_t)NL * FRAG_L * 8);
    for (int l = 0; l < NL; ++l)
        for (int nt = 0; nt < 8; ++nt)
            for (int ks = 0; ks < 4; ++ks)
                for (int ln = 0; ln < 64; ++ln)
                    for (int j = 0; j < 8; ++j) {
                        const int c = 16 * nt + (ln & 15), k = 32 * ks + 8 * (ln >> 4) + j;
                        const float v = w[((size_t)l * H + c) * H + k] * WS;
                        const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
                        const size_t base = ((size_t)l * FRAG_L + (((size_t)(nt * 4 + ks) * 2 + 0) * 64 + ln)) * 8 + j;
                        fr[base] = hi;
                        fr[base + 64 * 8] = lo;
                    }
    const int GMAX = 128;
    std::vector<float> x((size_t)GMAX * RB_NODE * H);
    for (auto& v : x) v = (rand() % 1000) * 1e-3f;
    float *d_wt, *d_x, *d_y;
    uint4* d_fr;
    hipMalloc(&d_wt, wt.size() * 4); hipMalloc(&d_x, x.size() * 4); hipMalloc(&d_y, x.size() * 4); hipMalloc(&d_fr, fr.size() * 2);
    hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_fr, fr.data(), fr.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> ya(x.size()), yb(x.size());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {2, 128}) {
        for (int variant = 0; variant < 3; ++variant) {
            auto launch = [&]() {
                if (variant == 0) hipLaunchKernelGGL(chain_valu, dim3(G), dim3(256), 0, 0, d_wt, d_x, d_y);
                if (variant == 1) hipLaunchKernelGGL(chain_mfma<false>, dim3(G), dim3(256), 0, 0, d_fr, d_x, d_y, 1.0f / WS);
                if (variant == 2) hipLaunchKernelGGL(chain_mfma<true>, dim3(G), dim3(256), 0, 0, d_fr, d_x, d_y, 1.0f / WS);
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(variant == 0 ? ya.data() : yb.data(), d_y, (size_t)G * RB_NODE * H * 4, hipMemcpyDeviceToHost);
            double err = 0, mag = 0;
            if (variant) for (size_t i = 0; i < (size_t)G * RB_NODE * H; ++i) { err = fmax(err, fabs(ya[i] - yb[i])); mag = fmax(mag, fabs(ya[i])); }
            printf("grid %3d  %-28s %7.2f us per kernel = %5.2f us per layer   max|diff| vs VALU %.3g (max|y| %.3g)\n", G,
                   variant == 0 ? "VALU fp32 (dense_lds)" : variant == 1 ? "MFMA fp16x3" : "MFMA fp16x3 + prefetch", ms * 1e3 / 50,
                   ms * 1e3 / 50 / NL, err, mag);
        }
    }
    return 0;
}
