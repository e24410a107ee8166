// Shared host/device helpers for libstrive_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/strive_hip.h"

#define STRIVE_WAVE 64

void strive_set_error(const char* fmt, ...);

// Tuning / measurement options of the library (strive_set_option / strive_get_option, include/strive_hip.h): process-wide plain
// ints with the shipped behaviour as defaults.  The library reads NO environment variable (tests/test_abi.py greps for getenv);
// the Python host maps STRIVE_<NAME> environment variables onto these options (strive_amd/_lib.py).
struct StriveTuning {
    int cnn_small_batch, cnn_chunk, cnn_tail_s, conv_ws, conv_wsx, conv_ws_dbg;
    int scene_kernels, scene_tiles, scene_prof, scene_split, scene_fwd_k, sweep_step;
    int train_overlap, train_overlap_rows;
    int wgrad_atomics, dgrad_igemm, wgrad_igemm, wgrad_tile, wgrad_dbg;
    int planner_prof;
    int planner_dbg;
};
StriveTuning& strive_tuning();

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

// Per-device one-time set-up (a dynamic-LDS attribute, the CU count): a process may drive several GPUs, and the attribute and
// the count belong to the device that is current when the launch is made.
struct PerDeviceOnce {
    std::atomic<unsigned long long> done{0};
    std::atomic<int> value[64];
    int device() const {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
        return dev;
    }
    bool is_done(int dev) const { return (done.load(std::memory_order_acquire) >> dev) & 1ull; }
    void set_done(int dev) { done.fetch_or(1ull << dev, std::memory_order_release); }
};

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
// __shfl_xor compiles to ds_bpermute_b32: an LDS-crossbar round trip (~100 cycles) per step, six dependent steps per
// reduction -- and the small dense kernels are chains of such reductions (two per LayerNorm row, four in its backward;
// tools/mlp_phase_probe.hip: a LayerNorm of 4 rows costs as much as a 128 x 128 layer).  The first four steps stay inside a
// row of 16 lanes, where the data-parallel-primitive modifiers move data in the VALU itself (quad_perm, row_half_mirror,
// row_mirror); the four row totals are then read through scalar registers.  Every lane gets the same value, added in one
// fixed order.
#define STRIVE_DPP_QUAD_XOR1 0xB1        // quad_perm [1,0,3,2]
#define STRIVE_DPP_QUAD_XOR2 0x4E        // quad_perm [2,3,0,1]
#define STRIVE_DPP_ROW_HALF_MIRROR 0x141 // lane i <- lane 7 - i within each 8
#define STRIVE_DPP_ROW_MIRROR 0x140      // lane i <- lane 15 - i within each 16
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<STRIVE_DPP_QUAD_XOR1>(v);
    v += dpp_move<STRIVE_DPP_QUAD_XOR2>(v);
    v += dpp_move<STRIVE_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move<STRIVE_DPP_ROW_MIRROR>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<STRIVE_DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_move<STRIVE_DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_move<STRIVE_DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_move<STRIVE_DPP_ROW_MIRROR>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_move_d<STRIVE_DPP_QUAD_XOR1>(v);
    v += dpp_move_d<STRIVE_DPP_QUAD_XOR2>(v);
    v += dpp_move_d<STRIVE_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_move_d<STRIVE_DPP_ROW_MIRROR>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// pos*std + mean exactly as MeanStdNormalizer.unnormalize evaluates it in fp32
// (reference src/datasets/utils.py:89-101): one rounded multiply, one rounded add.
__device__ __forceinline__ float unnorm1(float v, float mean, float std) { return __fadd_rn(__fmul_rn(v, std), mean); }
// (v - mean) / std (reference src/datasets/utils.py:58-73)
__device__ __forceinline__ float norm1(float v, float mean, float std) { return __fdiv_rn(__fsub_rn(v, mean), std); }

struct Float4Host { float v[4]; };
