// Probe: cost of ds_read_b128 on gfx950 as a function of where the four 16-lane groups of a wave point.
// Every lane reads 16 bytes at  base[g] + j*16  (g = lane/16, j = lane%16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void probe(const int* bases /*4*/, int jstride, int iters, long long* cycles, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int addr = bases[g] + j * jstride + (threadIdx.x >> 6) * 8192;
    uint4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
        const uint32_t a = (uint32_t)(uintptr_t)(lds) + (uint32_t)addr;   // LDS byte address
        asm volatile(
            "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
            "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
            "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)\n\t"
            : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
        acc.x ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3] ^ v4[0] ^ v5[1] ^ v6[2] ^ v7[3];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u && acc.y == 1 && acc.z == 2 && acc.w == 3) sink[0] = 1.f;
}

int main() {
    int* d_b; long long* d_c; float* sink;
    (void)hipMalloc(&d_b, 16); (void)hipMalloc(&d_c, 8 * 1024); (void)hipMalloc(&sink, 4);
    struct { const char* name; int b[4]; int js; } cases[] = {
        {"all groups same 256 B (broadcast)", {0, 0, 0, 0}, 16},
        {"groups shifted by 16 B (conv1 pattern)", {0, 16, 32, 48}, 16},
        {"groups 256 B apart (contiguous 1 KB)", {0, 256, 512, 768}, 16},
        {"groups 288 B apart (parity halves)", {0, 288, 576, 864}, 16},
        {"groups at 0,288,16,304 (taps kx..kx+3)", {0, 288, 16, 304}, 16},
        {"groups 272 B apart", {0, 272, 544, 816}, 16},
        {"groups 320 B apart", {0, 320, 640, 960}, 16},
        {"groups 576 B apart (rows)", {0, 576, 1152, 1728}, 16},
        {"lanes 32 B apart, groups 16 B apart", {0, 16, 512, 528}, 32},
        // r03: one 16-byte slot per lane at a padded row pitch (fragment rows = lanes, the transposed wgrad layout)
        {"lanes 144 B apart, halves +16 B", {0, 2304, 16, 2320}, 144},
        {"lanes 272 B apart, halves +16 B", {0, 4352, 16, 4368}, 272},
        {"lanes 528 B apart, halves +16 B", {0, 8448, 16, 8464}, 528},
        {"lanes 80 B apart, halves +16 B", {0, 1280, 16, 1296}, 80},
        {"lanes 48 B apart, halves +16 B", {0, 768, 16, 784}, 48},
        {"lanes 32 B apart, halves +16 B", {0, 512, 16, 528}, 32},
    };
    const int iters = 2000;
    for (auto& c : cases) {
        (void)hipMemcpy(d_b, c.b, 16, hipMemcpyHostToDevice);
        probe<<<256, 256>>>(d_b, c.js, iters, d_c, sink);
        (void)hipDeviceSynchronize();
        long long h[256]; (void)hipMemcpy(h, d_c, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("%-44s %.2f clk per ds_read_b128 per wave (4 waves/CU issuing)\n", c.name, s / 256 / (iters * 8.0));
    }
    return 0;
}
