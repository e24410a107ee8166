// Probe: sustained issue rate of the bf16 / f32 MFMAs used by the map CNN on gfx950 (independent accumulators).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256) void probe(int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((lane + j) & 7); b[j] = (__bf16)(float)((lane - j) & 3); }
    float r = 0.f;
    if (KIND == 0) {
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) c[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
        for (int k = 0; k < 8; ++k) r += c[k][0];
    } else if (KIND == 1) {
        f32x16 c[4];
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) c[k][e] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
        for (int k = 0; k < 4; ++k) r += c[k][0];
    } else {
        f32x16 c[4];
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) c[k][e] = 0.f;
        const float fa = (float)lane, fb = 1.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c[k], 0, 0, 0);
        for (int k = 0; k < 4; ++k) r += c[k][0];
    }
    if (r == 12345.f) sink[0] = r;
}

template <int KIND>
static void run(const char* name, int per_iter, double flop, float* sink) {
    for (int wg_per_cu : {1, 2}) {
        const int iters = 20000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<KIND><<<256 * wg_per_cu, 256>>>(100, sink);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        probe<KIND><<<256 * wg_per_cu, 256>>>(iters, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)256 * wg_per_cu * 4 * iters * per_iter;       // wave-level MFMAs
        printf("%-28s %d wave/SIMD: %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", name, wg_per_cu, ms * 1e6 / (n / 1024), n * flop / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float* sink; (void)hipMalloc(&sink, 4);
    run<0>("v_mfma_f32_16x16x32_bf16", 8, 16384.0, sink);
    run<1>("v_mfma_f32_32x32x16_bf16", 4, 32768.0, sink);
    run<2>("v_mfma_f32_32x32x2_f32", 4, 4096.0, sink);
    return 0;
}
