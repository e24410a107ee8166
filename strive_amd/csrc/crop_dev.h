// Device-side coordinate pipeline shared by the crop, the fused crop->conv1 and the collision-point
// kernels.  Restates gen_car_coords + the pixel conversion of get_map_obs
// (reference src/datasets/nuscenes_utils.py:205-232, 250-263) per sample, in registers.
#pragma once
#include "common.h"

struct CropFrame {
    float x, y, hc, hs;      // unnormalised pose
    double dx0, dx1;         // metres per pixel for x (dx[m][0]) and y (dx[m][1])
    double inv0, inv1;       // 1/dx, used only to PREDICT the quotient (see world_to_pixel)
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
    fr.inv0 = __ddiv_rn(1.0, fr.dx0);
    fr.inv1 = __ddiv_rn(1.0, fr.dx1);
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

// rint(double(g) / dx), bit-identical to the IEEE division the reference performs, without paying for one per
// sample: q' = g * (1/dx) differs from the exact quotient by a few ulps (|q'| < 2^24 here, so < 1e-8 absolute);
// unless q' lies within 1e-6 of a rounding boundary (k + 0.5) both round to the same integer.  The rare
// boundary cases take the exact division.  Non-finite inputs fall through to the division as well.
__device__ __forceinline__ double quotient_rint(double g, double dx, double inv) {
    const double q = g * inv;
    const double r = rint(q);
    const double d = fabs(q - r);                 // in [0, 0.5]
    if (d < 0.499999 && fabs(q) < 16777216.0) return r;
    return rint(__ddiv_rn(g, dx));
}

// float64 divide, round half to even, out-of-bounds (either axis) -> pixel (0,0)
__device__ __forceinline__ void world_to_pixel(const CropFrame& fr, float gx, float gy, int& px, int& py) {
    const double qx = quotient_rint((double)gx, fr.dx0, fr.inv0);
    const double qy = quotient_rint((double)gy, fr.dx1, fr.inv1);
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
