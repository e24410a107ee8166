// Probe: do co-resident workgroups keep private LDS when the per-workgroup footprint is close to 64 KB?
// Every workgroup fills its LDS with a workgroup-unique pattern, does some unrelated work, and verifies it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int STATIC_BYTES>
__global__ __launch_bounds__(512) void probe_static(int spin, unsigned* bad, float* sink) {
    __shared__ uint32_t s[STATIC_BYTES / 4];
    const int n = STATIC_BYTES / 4;
    const uint32_t tag = (blockIdx.x + 1) * 0x9E3779B1u;
    for (int i = threadIdx.x; i < n; i += 512) s[i] = tag ^ (uint32_t)i;
    __syncthreads();
    float acc = (float)threadIdx.x;
    for (int k = 0; k < spin; ++k) acc = acc * 1.0000001f + 0.5f;
    __syncthreads();
    unsigned nb = 0;
    for (int i = threadIdx.x; i < n; i += 512) nb += (s[i] != (tag ^ (uint32_t)i));
    if (nb) atomicAdd(bad, nb);
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(512) void probe_dyn(int nbytes, int spin, unsigned* bad, float* sink) {
    extern __shared__ uint32_t sd[];
    const int n = nbytes / 4;
    const uint32_t tag = (blockIdx.x + 1) * 0x9E3779B1u;
    for (int i = threadIdx.x; i < n; i += 512) sd[i] = tag ^ (uint32_t)i;
    __syncthreads();
    float acc = (float)threadIdx.x;
    for (int k = 0; k < spin; ++k) acc = acc * 1.0000001f + 0.5f;
    __syncthreads();
    unsigned nb = 0;
    for (int i = threadIdx.x; i < n; i += 512) nb += (sd[i] != (tag ^ (uint32_t)i));
    if (nb) atomicAdd(bad, nb);
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    unsigned* bad; float* sink;
    hipMalloc(&bad, 4); hipMalloc(&sink, 4);
    const int sizes[] = {32768, 60000, 63200, 64000, 64912, 65536, 70000, 81920};
    for (int s : sizes) {
        hipMemset(bad, 0, 4);
        hipFuncSetAttribute((const void*)probe_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, s);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe_dyn, 512, s);
        probe_dyn<<<2048, 512, s>>>(s, 20000, bad, sink);
        hipError_t e = hipDeviceSynchronize();
        unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("dynamic %6d B: occupancy %d WG/CU, err %d, corrupted words %u\n", s, occ, (int)e, h);
    }
    hipMemset(bad, 0, 4);
    probe_static<64912><<<2048, 512>>>(20000, bad, sink);
    hipError_t e = hipDeviceSynchronize();
    unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("static  64912 B: err %d, corrupted words %u\n", (int)e, h);
    hipMemset(bad, 0, 4);
    probe_static<63200><<<2048, 512>>>(20000, bad, sink);
    e = hipDeviceSynchronize();
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("static  63200 B: err %d, corrupted words %u\n", (int)e, h);
    return 0;
}
