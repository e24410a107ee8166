// Calibrate clock64() (s_memtime) against wall time and against the 32x32x16 bf16 MFMA issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void probe(int iters, long long* ticks, float* sink) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((lane + j) & 7); b[j] = (__bf16)(float)((lane - j) & 3); }
    f32x16 c[4];
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) c[k][e] = 0.f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
    const long long t1 = clock64();
    float r = 0.f;
    for (int k = 0; k < 4; ++k) r += c[k][0];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (r == 12345.f) sink[0] = r;
}
int main() {
    long long* d; float* sink;
    (void)hipMalloc(&d, 8 * 512); (void)hipMalloc(&sink, 4);
    for (int wgs : {256, 512}) {
        const int iters = 20000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<<<wgs, 256>>>(100, d, sink); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        probe<<<wgs, 256>>>(iters, d, sink);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[512]; (void)hipMemcpy(h, d, 8 * wgs, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < wgs; ++i) s += (double)h[i];
        printf("%d workgroups: kernel %.1f us, mean %.0f ticks per wave -> %.3f ticks/ns; %.2f ticks per MFMA issued by the wave\n",
               wgs, ms * 1e3, s / wgs, s / wgs / (ms * 1e6), s / wgs / (iters * 4.0));
    }
    return 0;
}
