// Device-side coordinate pipeline shared by the crop, the fused crop->conv1 and the collision-point
// kernels.  Restates gen_car_coords + the pixel conversion of get_map_obs
// (reference src/datasets/nuscenes_utils.py:205-232, 250-263) per sample, in registers.
#pragma once
#include "common.h"

struct CropFrame {
    float x, y, hc, hs;      // unnormalised pose
    double dx0, dx1;         // metres per pixel for x (dx[m][0]) and y (dx[m][1])
    double inv0, inv1;       // 1/dx, used only to PREDICT the quotient (see world_to_pixel)
    float finv0, finv1;      // 1/dx in fp32, exact when `pow2`
    bool pow2;               // both dx are powers of two (pix_per_m = 4 or 8 in the reference): g/dx is exact in fp32
    int H, W;
    const uint8_t* base;     // raster + m*C*H*W (channel 0 of the agent's map)
};

// dx = 2^e with 2^-100 <= dx <= 2^100: 1/dx is an exact normal fp32 and g * (1/dx) is the exact quotient
__device__ __forceinline__ bool dx_is_pow2(double dx) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(dx);
    const int e = (int)((b >> 52) & 0x7ff) - 1023;
    return (b >> 63) == 0 && (b & 0x000fffffffffffffull) == 0 && e >= -100 && e <= 100;
}

__device__ __forceinline__ void set_crop_scale(CropFrame& fr, double dx0, double dx1) {
    fr.dx0 = dx0;
    fr.dx1 = dx1;
    fr.inv0 = __ddiv_rn(1.0, dx0);
    fr.inv1 = __ddiv_rn(1.0, dx1);
    fr.finv0 = (float)fr.inv0;
    fr.finv1 = (float)fr.inv1;
    fr.pow2 = dx_is_pow2(dx0) && dx_is_pow2(dx1);
}

__device__ __forceinline__ CropFrame load_crop_frame(const StriveMap& map, const float* __restrict__ pos,
                                                     const float* pmean, const float* pstd,
                                                     const int32_t* __restrict__ mapix, int n) {
    CropFrame fr;
    fr.x = unnorm1(pos[n * 4 + 0], pmean[0], pstd[0]);
    fr.y = unnorm1(pos[n * 4 + 1], pmean[1], pstd[1]);
    fr.hc = unnorm1(pos[n * 4 + 2], pmean[2], pstd[2]);
    fr.hs = unnorm1(pos[n * 4 + 3], pmean[3], pstd[3]);
    const int m = mapix[n];
    set_crop_scale(fr, map.dx[m * 2 + 0], map.dx[m * 2 + 1]);
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
    if (fr.pow2) {
        // scaling by a power of two is exact, so rint(fp32 quotient) IS rint(double(g) / dx); overflow to inf and
        // NaN fail the bounds / ordered tests exactly like the out-of-range doubles do
        const float qx = rintf(__fmul_rn(gx, fr.finv0)), qy = rintf(__fmul_rn(gy, fr.finv1));
        const int ix = (int)qx, iy = (int)qy;
        const bool ordered = (qx == qx) & (qy == qy);
        const bool inside = ordered & ((unsigned)ix < (unsigned)fr.W) & ((unsigned)iy < (unsigned)fr.H);
        px = inside ? ix : 0;
        py = inside ? iy : 0;
        return;
    }
    const double qx = quotient_rint((double)gx, fr.dx0, fr.inv0);
    const double qy = quotient_rint((double)gy, fr.dx1, fr.inv1);
    // qx, qy are integer valued (or NaN / inf): test the bounds on the saturating int conversion (NaN -> 0 is
    // excluded by the explicit ordered test; -0.0 is inside, as `pix >= 0` is in the reference)
    const int ix = (int)qx, iy = (int)qy;
    const bool ordered = (qx == qx) & (qy == qy);
    const bool inside = ordered & ((unsigned)ix < (unsigned)fr.W) & ((unsigned)iy < (unsigned)fr.H);
    px = inside ? ix : 0;
    py = inside ? iy : 0;
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
