// Probe (gfx950): v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 -- fragment layout (assumed = the bf16 variants'),
// handling of fp16 SUBNORMAL inputs (kept or flushed?), exactness of products, and issue rate next to the bf16 shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe32(const float* A /*32x16*/, const float* B /*16x32*/, float* D /*32x32*/) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = (l >> 5) * 8 + j;
        a[j] = (_Float16)A[(l & 31) * 16 + k];
        b[j] = (_Float16)B[k * 32 + (l & 31)];
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void probe16(const float* A /*16x32*/, const float* B /*32x16*/, float* D /*16x16*/) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = (l >> 4) * 8 + j;
        a[j] = (_Float16)A[(l & 15) * 32 + k];
        b[j] = (_Float16)B[k * 16 + (l & 15)];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

template <bool F16>
__global__ void rate(float* out, int iters) {
    f16x8 a, b;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)0.5f; ab[j] = (__bf16)0.25f; bb[j] = (__bf16)0.5f; }
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        if (F16) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    // ---- layouts ----
    {
        float hA[32 * 16], hB[16 * 32], hD[1024], ref[1024];
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 7 + k * 3) % 13 - 6);
        for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 11) % 17 - 8);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
        float *dA, *dB, *dD;
        hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        float e = 0; for (int i = 0; i < 1024; ++i) e = fmaxf(e, fabsf(hD[i] - ref[i]));
        printf("mfma_f32_32x32x16_f16 layout: max |D - ref| = %g (%s)\n", e, e == 0.f ? "as the bf16 variant" : "MISMATCH");
        // subnormal inputs: A[0][0] = 2^-20 (fp16 subnormal), B[0][0] = 1024 -> D[0][0] = 2^-10 if kept, 0 if flushed
        for (int i = 0; i < 32 * 16; ++i) hA[i] = 0.f;
        for (int i = 0; i < 16 * 32; ++i) hB[i] = 0.f;
        hA[0] = ldexpf(1.f, -20); hB[0] = 1024.f;
        hA[1 * 16 + 1] = ldexpf(3.f, -24); hB[1 * 32 + 1] = 2048.f;     // smallest subnormals: 3 * 2^-24 * 2^11 = 3 * 2^-13
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        printf("fp16 subnormal inputs: D[0][0] = %g (kept: %g), D[1][1] = %g (kept: %g) -> %s\n", hD[0], ldexpf(1.f, -10), hD[33],
               ldexpf(3.f, -13), (hD[0] == ldexpf(1.f, -10) && hD[33] == ldexpf(3.f, -13)) ? "subnormals are KEPT" : "subnormals are FLUSHED");
        // exactness + fp32 accumulation of large scaled products: (2047 * 2^4) * (2047 * 2^4) summed 16 times
        for (int i = 0; i < 32 * 16; ++i) hA[i] = 0.f;
        for (int i = 0; i < 16 * 32; ++i) hB[i] = 0.f;
        for (int k = 0; k < 16; ++k) { hA[k] = 2047.f * 16.f; hB[k * 32] = 2047.f * 16.f - 16.f * k; }
        double want = 0; for (int k = 0; k < 16; ++k) want += (double)hA[k] * (double)hB[k * 32];
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        printf("16 products of 11-bit x 11-bit significands: D = %.1f, float64 sum = %.1f, rel err %.3g\n", hD[0], want, fabs(hD[0] - want) / want);
    }
    {
        float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) hA[i * 32 + k] = (float)((i * 7 + k * 3) % 13 - 6);
        for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 5 + j * 11) % 17 - 8);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
        float *dA, *dB, *dD;
        hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        float e = 0; for (int i = 0; i < 256; ++i) e = fmaxf(e, fabsf(hD[i] - ref[i]));
        printf("mfma_f32_16x16x32_f16 layout: max |D - ref| = %g (%s)\n", e, e == 0.f ? "as the bf16 variant" : "MISMATCH");
    }
    // ---- rate: 256 CUs x 4 waves x 4 independent chains ----
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    for (int f16 = 0; f16 < 2; ++f16) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) hipLaunchKernelGGL(rate<true>, dim3(1024), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(rate<false>, dim3(1024), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 1024.0 * 4 * iters * 4 * 32768.0;
        printf("%s 32x32x16: %.1f TFLOP/s\n", f16 ? "f16 " : "bf16", flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
