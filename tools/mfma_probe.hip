// Probe: fragment layout of v_mfma_f32_16x16x32_bf16 on gfx950 (asymmetric operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __bf16 to_bf16(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint16_t h = (uint16_t)(v.u >> 16);
    __bf16 r; __builtin_memcpy(&r, &h, 2); return r;
}

__global__ void probe(const float* A /*16x32*/, const float* B /*32x16*/, float* D /*16x16*/) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = (l >> 4) * 8 + j;
        a[j] = to_bf16(A[(l & 15) * 32 + k]);
        b[j] = to_bf16(B[k * 16 + (l & 15)]);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) hA[i * 32 + k] = (float)((i * 7 + k * 3) % 13 - 6);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 5 + j * 11) % 17 - 8);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += hA[i * 32 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    float e = 0; for (int i = 0; i < 256; ++i) e = fmaxf(e, fabsf(hD[i] - ref[i]));
    printf("mfma_f32_16x16x32_bf16 layout probe: max |D - ref| = %g (%s)\n", e, e == 0.f ? "layout as assumed" : "MISMATCH");
    return 0;
}
