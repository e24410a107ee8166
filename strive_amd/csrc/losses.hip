// Vehicle-vehicle collision penalty with the 5-circle approximation, forward and backward.
// (reference src/losses/adv_gen_nusc.py:405-512; training variant src/losses/traffic_model.py:166-238)
//
// The reference expands every ordered pair of the WHOLE batch at every time sample into two
// (T*NA*NA, 5, 2) tensors, runs cdist and only then masks out cross-scene pairs; here only the in-scene
// blocks exist: sum_b n_b^2 slots per time sample instead of NA^2 (32x fewer at 32 scenes of 16).
// One thread owns (agent i, time t) and walks the members j of i's scene; the scene's circle centres
// are a few hundred bytes, so everything stays in registers / L1.
#include "common.h"

#define NCIRC 5

struct VehArgs {
    int NA, T, P;
    const int32_t* ptr;
    const int32_t* scene_of;
    const int32_t* pair_off;
    const float* traj;     // (NA, T, 4) unnormalised (x, y, hx, hy)
    const float* cent_x;   // (NA, 5)
    const float* rad;      // (NA)
    float buffer;
};

// world-frame circle centres of agent a at time t: transform2frame(inverse) of (cx, 0)
// (reference src/utils/transforms.py:113-133): (c*cx + x, s*cx + y)
__device__ __forceinline__ void circle_centres(const VehArgs& a, int ag, int t, float* cx, float* cy) {
    const float* p = a.traj + ((size_t)ag * a.T + t) * 4;
    for (int k = 0; k < NCIRC; ++k) {
        const float c0 = a.cent_x[ag * NCIRC + k];
        cx[k] = p[2] * c0 + p[0];
        cy[k] = p[3] * c0 + p[1];
    }
}

// A group of VG consecutive lanes owns one (agent i, time t): lane g of the group handles the scene members
// jl = g, g + VG, ... so a 16-agent scene is one pass, and the backward reduces the group's partial gradients
// with shuffles (deterministic, no atomics).
#define VG 16

__global__ __launch_bounds__(256) void veh_coll_fwd_kernel(VehArgs a, float* __restrict__ pen, uint8_t* __restrict__ hit,
                                                             uint8_t* __restrict__ amin) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / VG;
    const int sub = threadIdx.x & (VG - 1);
    if (gid >= a.NA * a.T) return;
    const int i = gid / a.T, t = gid - i * a.T;
    const int b = a.scene_of[i];
    const int lo = a.ptr[b], n = a.ptr[b + 1] - lo;
    float ax[NCIRC], ay[NCIRC];
    circle_centres(a, i, t, ax, ay);
    const float ri = a.rad[i];
    const size_t base = (size_t)t * a.P + a.pair_off[i];
    for (int jl = sub; jl < n; jl += VG) {
        const int j = lo + jl;
        float bx[NCIRC], by[NCIRC];
        circle_centres(a, j, t, bx, by);
        float dmin = 3.0e38f;
        int am = 0;
        for (int p = 0; p < NCIRC; ++p)
            for (int q = 0; q < NCIRC; ++q) {
                const float dx = ax[p] - bx[q], dy = ay[p] - by[q];
                const float d = sqrtf(dx * dx + dy * dy);
                if (d < dmin) { dmin = d; am = p * NCIRC + q; }
            }
        const float pd = (ri + a.rad[j]) + a.buffer;
        pen[base + jl] = 1.0f - dmin / pd;
        hit[base + jl] = (j != i && dmin <= pd) ? 1 : 0;
        amin[base + jl] = (uint8_t)am;
    }
}

// d_traj[i][t] += sum_j [ d_pen(i,j) * dpen(i,j)/dpose_i  +  d_pen(j,i) * dpen(j,i)/dpose_i ]
__global__ __launch_bounds__(256) void veh_coll_bwd_kernel(VehArgs a, const float* __restrict__ d_pen,
                                                             const uint8_t* __restrict__ amin, float* __restrict__ d_traj) {
    const int gid_raw = (blockIdx.x * blockDim.x + threadIdx.x) / VG;
    const int sub = threadIdx.x & (VG - 1);
    const bool live = gid_raw < a.NA * a.T;      // whole groups are live or not; every lane stays for the shuffles
    const int gid = live ? gid_raw : 0;
    const int i = gid / a.T, t = gid - i * a.T;
    const int b = a.scene_of[i];
    const int lo = a.ptr[b], n = a.ptr[b + 1] - lo;
    float ax[NCIRC], ay[NCIRC];
    circle_centres(a, i, t, ax, ay);
    const float ri = a.rad[i];
    const int il = i - lo;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    for (int jl = sub; live && jl < n; jl += VG) {
        const int j = lo + jl;
        if (j == i) continue;
        float bx[NCIRC], by[NCIRC];
        circle_centres(a, j, t, bx, by);
        const float pd = (ri + a.rad[j]) + a.buffer;
        // pair (i, j): i is the first member
        {
            const size_t s = (size_t)t * a.P + a.pair_off[i] + jl;
            const float gp = d_pen[s];
            if (gp != 0.f) {
                const int am = amin[s];
                const int p = am / NCIRC, q = am - p * NCIRC;
                const float dx = ax[p] - bx[q], dy = ay[p] - by[q];
                const float d = sqrtf(dx * dx + dy * dy);
                if (d > 0.f) {
                    const float k = -gp / (pd * d);
                    const float gx = k * dx, gy = k * dy;
                    const float c0 = a.cent_x[i * NCIRC + p];
                    g[0] += gx; g[1] += gy; g[2] += gx * c0; g[3] += gy * c0;
                }
            }
        }
        // pair (j, i): i is the second member
        {
            const size_t s = (size_t)t * a.P + a.pair_off[j] + il;
            const float gp = d_pen[s];
            if (gp != 0.f) {
                const int am = amin[s];
                const int p = am / NCIRC, q = am - p * NCIRC;   // p indexes j's circles, q indexes i's
                const float dx = bx[p] - ax[q], dy = by[p] - ay[q];
                const float d = sqrtf(dx * dx + dy * dy);
                if (d > 0.f) {
                    const float k = gp / (pd * d);
                    const float gx = k * dx, gy = k * dy;
                    const float c0 = a.cent_x[i * NCIRC + q];
                    g[0] += gx; g[1] += gy; g[2] += gx * c0; g[3] += gy * c0;
                }
            }
        }
    }
#pragma unroll
    for (int m = VG / 2; m >= 1; m >>= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] += __shfl_xor(g[k], m);
    if (live && sub == 0) {
        float* o = d_traj + ((size_t)i * a.T + t) * 4;
        for (int k = 0; k < 4; ++k) o[k] += g[k];
    }
}

static VehArgs veh_args(const StriveScenes* sc, const int32_t* pair_off, int P, const float* traj, int T,
                        const float* cent_x, const float* rad, float buffer) {
    VehArgs a;
    a.NA = sc->NA; a.T = T; a.P = P; a.ptr = sc->ptr; a.scene_of = sc->scene_of; a.pair_off = pair_off;
    a.traj = traj; a.cent_x = cent_x; a.rad = rad; a.buffer = buffer;
    return a;
}

extern "C" int strive_veh_coll_fwd(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj, int32_t T,
                                   const float* cent_x, const float* rad, float buffer, float* pen, uint8_t* hit,
                                   uint8_t* amin, strive_stream_t stream) {
    STRIVE_CHECK_ARG(sc && pair_off && traj && cent_x && rad && pen && hit && amin, "null argument");
    STRIVE_CHECK_ARG(sc->NS == 1, "collision losses take one trajectory per agent");
    const long long n = (long long)sc->NA * T * VG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(veh_coll_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       veh_args(sc, pair_off, P, traj, T, cent_x, rad, buffer), pen, hit, amin);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_veh_coll_bwd(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj, int32_t T,
                                   const float* cent_x, const float* rad, float buffer, const float* d_pen,
                                   const uint8_t* amin, float* d_traj, strive_stream_t stream) {
    STRIVE_CHECK_ARG(sc && pair_off && traj && cent_x && rad && d_pen && amin && d_traj, "null argument");
    STRIVE_CHECK_ARG(sc->NS == 1, "collision losses take one trajectory per agent");
    const long long n = (long long)sc->NA * T * VG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(veh_coll_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       veh_args(sc, pair_off, P, traj, T, cent_x, rad, buffer), d_pen, amin, d_traj);
    STRIVE_CHECK_LAUNCH();
    return 0;
}
