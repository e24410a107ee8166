// Device-side coordinate pipeline shared by the crop, the fused crop->conv1 and the collision-point
// kernels.  Restates gen_car_coords + the pixel conversion of get_map_obs
// (reference src/datasets/nuscenes_utils.py:205-232, 250-263) per sample, in registers.
#pragma once
#include "common.h"

struct CropFrame {
    float x, y, hc, hs;      // unnormalised pose
    double dx0, dx1;         // metres per pixel for x (dx[m][0]) and y (dx[m][1])
    int H, W;
    const uint8_t* base;     // raster + m*C*H*W (channel 0 of the agent's map)
};

__device__ __forceinline__ CropFrame load_crop_frame(const StriveMap& map, const float* __restrict__ pos,
                                                     const float* pmean, const float* pstd,
                                                     const int32_t* __restrict__ mapix, int n) {
    CropFrame fr;
    fr.x = unnorm1(pos[n * 4 + 0], pmean[0], pstd[0]);
    fr.y = unnorm1(pos[n * 4 + 1], pmean[1], pstd[1]);
    fr.hc = unnorm1(pos[n * 4 + 2], pmean[2], pstd[2]);
    fr.hs = unnorm1(pos[n * 4 + 3], pmean[3], pstd[3]);
    const int m = mapix[n];
    fr.dx0 = map.dx[m * 2 + 0];
    fr.dx1 = map.dx[m * 2 + 1];
    fr.H = map.H;
    fr.W = map.W;
    fr.base = map.raster + (size_t)m * map.C * map.H * map.W;
    return fr;
}

// world = (l*cos - w*sin) + x ; (l*sin + w*cos) + y, each operation rounded separately
__device__ __forceinline__ void crop_world(const CropFrame& fr, float lwise, float wwise, float& gx, float& gy) {
    gx = __fadd_rn(__fsub_rn(__fmul_rn(lwise, fr.hc), __fmul_rn(wwise, fr.hs)), fr.x);
    gy = __fadd_rn(__fadd_rn(__fmul_rn(lwise, fr.hs), __fmul_rn(wwise, fr.hc)), fr.y);
}

// float64 divide, round half to even, out-of-bounds (either axis) -> pixel (0,0)
__device__ __forceinline__ void world_to_pixel(const CropFrame& fr, float gx, float gy, int& px, int& py) {
    const double qx = rint(__ddiv_rn((double)gx, fr.dx0));
    const double qy = rint(__ddiv_rn((double)gy, fr.dx1));
    const bool inside = (qy >= 0.0) && (qy < (double)fr.H) && (qx >= 0.0) && (qx < (double)fr.W);
    px = inside ? (int)qx : 0;
    py = inside ? (int)qy : 0;
}

__device__ __forceinline__ void crop_pixel(const CropFrame& fr, float lwise, float wwise, bool nan_to_zero,
                                           int& px, int& py) {
    float gx, gy;
    crop_world(fr, lwise, wwise, gx, gy);
    if (nan_to_zero) {
        gx = (gx != gx) ? 0.0f : gx;
        gy = (gy != gy) ? 0.0f : gy;
    }
    world_to_pixel(fr, gx, gy, px, py);
}
