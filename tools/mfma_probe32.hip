// Probe: fragment layout of v_mfma_f32_32x32x16_bf16 on gfx950 (asymmetric operands).
// Assumed: lane l holds k = 8*(l>>5)+j of A row / B column l&31; D col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __bf16 to_bf16(float f) { union { float f; uint32_t u; } v; v.f = f; uint16_t h = (uint16_t)(v.u >> 16); __bf16 r; __builtin_memcpy(&r, &h, 2); return r; }
__global__ void probe(const float* A /*32x16*/, const float* B /*16x32*/, float* D /*32x32*/) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = (l >> 5) * 8 + j;
        a[j] = to_bf16(A[(l & 31) * 16 + k]);
        b[j] = to_bf16(B[k * 32 + (l & 31)]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[1024], ref[1024];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (float)((i * 7 + k * 3) % 13 - 6);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((k * 5 + j * 11) % 9 - 4);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dD, sizeof(hD));
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (hD[i] != ref[i]);
    printf("v_mfma_f32_32x32x16_bf16 layout %s (%d mismatches)\n", bad ? "DIFFERS from the assumption" : "as assumed", bad);
    return 0;
}
