// TEST INFRASTRUCTURE ONLY -- a tiny host-side model of the HIP execution model.
//
// There is no GPU in the build container, so kernel *logic* (indexing, LDS staging, barriers,
// wave shuffles, MFMA fragment layouts, atomics) would otherwise first run at round end on the
// MI355X box.  tests/hipemu/build.py compiles the SAME strive_amd/csrc/*.hip sources, unmodified,
// as host C++ against this header: every workgroup runs as a set of cooperative fibers on one OS
// thread, __syncthreads() and the wave-collective builtins are rendezvous points, and the two f32
// MFMA builtins are modelled with the fragment layout documented for gfx950
// (/opt/skills/guides/cdna_hip_programming.md §3).  The resulting library is loaded ONLY by
// tests (tests/test_emu_*.py); the product loader (strive_amd/_lib.py) never looks for it, and
// nothing here is a fallback: no performance claim and no parity claim rests on it.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __noinline__ __attribute__((noinline))
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::g.dyn_smem);

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(8) float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct alignas(8) uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline long long clock64() { return 0; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// streams / events: launches are synchronous in the emulator, so forking and joining streams is a no-op
struct hipEvent_emu;
typedef hipEvent_emu* hipEvent_t;
static const unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }   // 3 "CUs": persistent kernels loop
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// graphs: capture is refused, so the library takes its plain-launch path
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 1; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

namespace hipemu {

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
};

struct Wave {
    float f[64];
    float f2[64];
    double d[64];
    long long i[64];
    int arrived = 0;
    unsigned gen = 0;
    int nlanes = 64;
};

struct Globals {
    std::vector<Fiber> fibers;
    ucontext_t main_ctx;
    int cur = 0;
    int nthreads = 0;
    int alive = 0;
    int barrier_arrived = 0;
    unsigned barrier_gen = 0;
    std::vector<Wave> waves;
    std::function<void()> body;
    unsigned char* dyn_smem = nullptr;
    size_t dyn_cap = 0;
    uint3_emu tid[1024];
};
extern Globals g;
static const size_t STACK_BYTES = 512 * 1024;

void yield_();
void block_barrier();
void wave_barrier();
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem);

}  // namespace hipemu

extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))

inline void __syncthreads() { hipemu::block_barrier(); }
inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }      // callers pass wave-uniform values
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
inline void __threadfence() {}
inline void __threadfence_block() {}

inline int hipemu_lane() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63; }
inline hipemu::Wave& hipemu_wave() {
    int lin = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    return hipemu::g.waves[lin >> 6];
}

template <typename T> inline T hipemu_exchange(T v, int src_lane) {
    hipemu::Wave& w = hipemu_wave();
    int lane = hipemu_lane();
    static_assert(sizeof(T) <= 8, "exchange");
    long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    w.i[lane] = bits;
    hipemu::wave_barrier();
    long long r = w.i[(src_lane >= 0 && src_lane < w.nlanes) ? src_lane : lane];
    hipemu::wave_barrier();
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) { return hipemu_exchange(v, hipemu_lane() ^ mask); }
template <typename T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = hipemu_lane();
    int src = lane + (int)delta;
    if ((src / width) != (lane / width)) src = lane;
    return hipemu_exchange(v, src);
}
template <typename T> inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int lane = hipemu_lane();
    int src = lane - (int)delta;
    if (src < 0 || (src / width) != (lane / width)) src = lane;
    return hipemu_exchange(v, src);
}
// data-parallel-primitive moves used by common.h's wave reductions (row = 16 lanes): quad_perm xor 1 / xor 2, row_half_mirror,
// row_mirror; readlane = broadcast of one lane
inline int __builtin_amdgcn_update_dpp(int /*old*/, int src, int ctrl, int, int, bool) {
    const int l = hipemu_lane();
    int from = l;
    if (ctrl == 0xB1) from = l ^ 1;
    else if (ctrl == 0x4E) from = l ^ 2;
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
    else { fprintf(stderr, "hipemu: unsupported dpp control 0x%x\n", ctrl); abort(); }
    return hipemu_exchange(src, from);
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu_exchange(v, lane); }
inline int __double2hiint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b >> 32); }
inline int __double2loint(double d) { long long b; memcpy(&b, &d, 8); return (int)(b & 0xffffffffll); }
inline double __hiloint2double(int hi, int lo) {
    const long long b = ((long long)hi << 32) | (long long)(unsigned)lo;
    double d; memcpy(&d, &b, 8); return d;
}
template <typename T> inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu_lane();
    return hipemu_exchange(v, (lane / width) * width + (src % width));
}
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline unsigned long long __ballot(int pred) {
    hipemu::Wave& w = hipemu_wave();
    int lane = hipemu_lane();
    w.i[lane] = pred ? 1 : 0;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    for (int l = 0; l < w.nlanes; ++l) if (w.i[l]) m |= (1ull << l);
    hipemu::wave_barrier();
    return m;
}

// ---- atomics (single OS thread: plain read-modify-write) ----
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T unsafeAtomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
inline int atomicCAS(int* p, int c, int v) { int o = *p; if (o == c) *p = v; return o; }

// ---- math intrinsics: round-to-nearest single ops (the build uses -ffp-contract=off) ----
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline double __dmul_rn(double a, double b) { volatile double r = a * b; return r; }
inline double __ddiv_rn(double a, double b) { volatile double r = a / b; return r; }
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }

// ---- f32 MFMA builtins, gfx950 fragment layout ----
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

// 16x16x4: A[i][k] held by lane k*16+i, B[k][j] by lane k*16+j; D col = lane&15, row = (lane>>4)*4 + reg
inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c, int, int, int) {
    hipemu::Wave& w = hipemu_wave();
    int lane = hipemu_lane();
    w.f[lane] = a;
    w.f2[lane] = b;
    hipemu::wave_barrier();
    hipemu_f32x4 d = c;
    int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.f[k * 16 + i], w.f2[k * 16 + j], acc);
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
// 32x32x2: A[i][k] held by lane k*32+i, B[k][j] by lane k*32+j; D col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c, int, int, int) {
    hipemu::Wave& w = hipemu_wave();
    int lane = hipemu_lane();
    w.f[lane] = a;
    w.f2[lane] = b;
    hipemu::wave_barrier();
    hipemu_f32x16 d = c;
    int j = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.f[k * 32 + i], w.f2[k * 32 + j], acc);
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
// 16x16x32 bf16: lane l holds k = 8*(l>>4) + j (j = 0..7) of row/column l&15 for BOTH operands (verified on
// gfx950 with tools/mfma_probe.hip); D as for 16x16x4.  Products are exact in fp32 (8 x 8 significand bits);
// the hardware's internal summation order is not modelled -- plain fp32 adds in k order.
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
inline float hipemu_bf16_to_float(__bf16 v) {
    uint16_t h;
    memcpy(&h, &v, 2);
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
struct hipemu_bf16_wave { float a[64][8]; float b[64][8]; };
inline hipemu_bf16_wave& hipemu_bf16_buf() {
    static std::vector<hipemu_bf16_wave> bufs(16);
    int lin = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    return bufs[lin >> 6];
}
inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
    hipemu_bf16_wave& w = hipemu_bf16_buf();
    int lane = hipemu_lane();
    for (int j = 0; j < 8; ++j) {
        w.a[lane][j] = hipemu_bf16_to_float(a[j]);
        w.b[lane][j] = hipemu_bf16_to_float(b[j]);
    }
    hipemu::wave_barrier();
    hipemu_f32x4 d = c;
    int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) acc += w.a[g * 16 + row][j] * w.b[g * 16 + col][j];
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
// 32x32x16 bf16: lane l holds k = 8*(l>>5) + j (j = 0..7) of row/column l&31 for both operands; D as for 32x32x2
// (verified on gfx950 with tools/mfma_probe.hip)
inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
    hipemu_bf16_wave& w = hipemu_bf16_buf();
    int lane = hipemu_lane();
    for (int j = 0; j < 8; ++j) {
        w.a[lane][j] = hipemu_bf16_to_float(a[j]);
        w.b[lane][j] = hipemu_bf16_to_float(b[j]);
    }
    hipemu::wave_barrier();
    hipemu_f32x16 d = c;
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) acc += w.a[h * 32 + row][j] * w.b[h * 32 + col][j];
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
// fp16 variants: same fragment layout as the bf16 ones (verified on gfx950 with tools/mfma_f16_probe.hip; fp16 subnormals
// are kept, products of two fp16 values are exact in fp32)
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
inline hipemu_f32x4 hipemu_mfma_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c, int, int, int) {
    hipemu_bf16_wave& w = hipemu_bf16_buf();
    int lane = hipemu_lane();
    for (int j = 0; j < 8; ++j) {
        w.a[lane][j] = (float)a[j];
        w.b[lane][j] = (float)b[j];
    }
    hipemu::wave_barrier();
    hipemu_f32x4 d = c;
    int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 8; ++j) acc += w.a[g * 16 + row][j] * w.b[g * 16 + col][j];
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
inline hipemu_f32x16 hipemu_mfma_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c, int, int, int) {
    hipemu_bf16_wave& w = hipemu_bf16_buf();
    int lane = hipemu_lane();
    for (int j = 0; j < 8; ++j) {
        w.a[lane][j] = (float)a[j];
        w.b[lane][j] = (float)b[j];
    }
    hipemu::wave_barrier();
    hipemu_f32x16 d = c;
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 8; ++j) acc += w.a[h * 32 + row][j] * w.b[h * 32 + col][j];
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu_mfma_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu_mfma_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_32x32x2
