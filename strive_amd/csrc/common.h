// Shared host/device helpers for libstrive_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/strive_hip.h"

#define STRIVE_WAVE 64

void strive_set_error(const char* fmt, ...);

#define STRIVE_CHECK_ARG(cond, msg)                                   \
    do {                                                              \
        if (!(cond)) {                                                \
            strive_set_error("%s: %s", __func__, msg);                \
            return -1;                                                \
        }                                                             \
    } while (0)

#define STRIVE_CHECK_LAUNCH()                                                             \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) {                                                          \
            strive_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
            return -2;                                                                    \
        }                                                                                 \
    } while (0)

static inline size_t strive_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace.
struct StriveArena {
    char* base;
    size_t cap, off;
    StriveArena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
    template <typename T> T* take(size_t count) {
        size_t bytes = strive_align_up(count * sizeof(T), 256);
        if (off + bytes > cap) { off = cap + 1; return nullptr; }
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= cap; }
};

// ---- wave-level reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// pos*std + mean exactly as MeanStdNormalizer.unnormalize evaluates it in fp32
// (reference src/datasets/utils.py:89-101): one rounded multiply, one rounded add.
__device__ __forceinline__ float unnorm1(float v, float mean, float std) { return __fadd_rn(__fmul_rn(v, std), mean); }
// (v - mean) / std (reference src/datasets/utils.py:58-73)
__device__ __forceinline__ float norm1(float v, float mean, float std) { return __fdiv_rn(__fsub_rn(v, mean), std); }

struct Float4Host { float v[4]; };
