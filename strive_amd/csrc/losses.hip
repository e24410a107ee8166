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
    // agent fastest: the slots of consecutive agents at one time sample are contiguous in pen / hit / amin, so a wave's
    // stores (4 groups x 16 lanes) form one contiguous run instead of four runs P floats apart
    const int t = gid / a.NA, i = gid - t * a.NA;
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
        // min over the 25 centre distances, first index on ties like torch.min.  The correctly rounded square root is
        // monotonic, so min sqrt(d2) = sqrt(min d2): one square root instead of 25.  Two different d2 can round to the same
        // distance, and the reference then reports the EARLIER index: the (rare) candidates within 2 ulp of the minimum are
        // re-checked with their own square root.
        float d2[NCIRC * NCIRC];
        float m2 = 3.0e38f;
#pragma unroll
        for (int p = 0; p < NCIRC; ++p)
#pragma unroll
            for (int q = 0; q < NCIRC; ++q) {
                const float dx = ax[p] - bx[q], dy = ay[p] - by[q];
                const float v = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                d2[p * NCIRC + q] = v;
                m2 = fminf(m2, v);
            }
        const float dmin = m2 < 3.0e38f ? sqrtf(m2) : 3.0e38f;      // NaN poses: nothing compares below the sentinel
        const float thr = m2 * 1.0000005f;
        int am = NCIRC * NCIRC;
#pragma unroll
        for (int k = NCIRC * NCIRC - 1; k >= 0; --k)
            if (d2[k] <= thr && (d2[k] == m2 || sqrtf(d2[k]) == dmin)) am = k;
        if (am == NCIRC * NCIRC) am = 0;
        const float pd = (ri + a.rad[j]) + a.buffer;
        pen[base + jl] = 1.0f - dmin / pd;
        hit[base + jl] = (j != i && dmin <= pd) ? 1 : 0;
        amin[base + jl] = (uint8_t)am;
    }
}

// Gradient of the masked MEAN of the penalties without a d_pen tensor (fused AvoidCollLoss): d_pen(slot) = g for the
// slots that are colliding and valid, g = d_loss * w / max(#such slots, 1) read from the forward's sums.
struct VehImplicit {
    const uint8_t* hit;       // (T,P)
    const uint8_t* valid;     // (P)
    const double* sums;       // [1] = number of colliding valid slots
    const float* d_loss;      // device scalar
    float w;
};

// d_traj[i][t] += sum_j [ d_pen(i,j) * dpen(i,j)/dpose_i  +  d_pen(j,i) * dpen(j,i)/dpose_i ]
template <bool IMPL>
__global__ __launch_bounds__(256) void veh_coll_bwd_kernel(VehArgs a, const float* __restrict__ d_pen,
                                                             const uint8_t* __restrict__ amin, float* __restrict__ d_traj,
                                                             VehImplicit im) {
    const int gid_raw = (blockIdx.x * blockDim.x + threadIdx.x) / VG;
    const int sub = threadIdx.x & (VG - 1);
    const bool live = gid_raw < a.NA * a.T;      // whole groups are live or not; every lane stays for the shuffles
    const int gid = live ? gid_raw : 0;
    const int t = gid / a.NA, i = gid - t * a.NA;      // agent fastest (see the forward kernel)
    const int b = a.scene_of[i];
    const int lo = a.ptr[b], n = a.ptr[b + 1] - lo;
    float ax[NCIRC], ay[NCIRC];
    circle_centres(a, i, t, ax, ay);
    const float ri = a.rad[i];
    const int il = i - lo;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    float gs = 0.f;
    if (IMPL) {
        const double c = im.sums[1];
        gs = (float)((double)im.d_loss[0] * (double)im.w / (c < 1.0 ? 1.0 : c));
    }
    for (int jl = sub; live && jl < n; jl += VG) {
        const int j = lo + jl;
        if (j == i) continue;
        float bx[NCIRC], by[NCIRC];
        circle_centres(a, j, t, bx, by);
        const float pd = (ri + a.rad[j]) + a.buffer;
        // pair (i, j): i is the first member
        {
            const size_t s = (size_t)t * a.P + a.pair_off[i] + jl;
            const float gp = IMPL ? ((im.hit[s] && im.valid[a.pair_off[i] + jl]) ? gs : 0.f) : d_pen[s];
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
            const float gp = IMPL ? ((im.hit[s] && im.valid[a.pair_off[j] + il]) ? gs : 0.f) : d_pen[s];
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
    hipLaunchKernelGGL(veh_coll_bwd_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       veh_args(sc, pair_off, P, traj, T, cent_x, rad, buffer), d_pen, amin, d_traj, VehImplicit());
    STRIVE_CHECK_LAUNCH();
    return 0;
}

// =============================================================================================
// interp_traj (reference src/losses/adv_gen_nusc.py:625-644): linear up-sampling in time of (x,y,hx,hy) by an
// integer factor (F.interpolate(mode='linear', align_corners=False)) followed by renormalisation of the heading.
// torch's generic upsample backward costs 1.9 ms per call here (atomics); this pair is a thread per output /
// per input element with the tap tables (i0, i1, w0, w1 per output step, computed on the host in fp32 exactly
// like ATen's area_pixel_compute_source_index) passed in.
// =============================================================================================
__global__ __launch_bounds__(256) void interp_traj_fwd_kernel(const float* __restrict__ in, int N, int T, int TO,
                                                                const int32_t* __restrict__ i0, const int32_t* __restrict__ i1,
                                                                const float* __restrict__ w0, const float* __restrict__ w1,
                                                                float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * TO) return;
    const int n = idx / TO, j = idx - n * TO;
    const float* a = in + ((size_t)n * T + i0[j]) * 4;
    const float* b = in + ((size_t)n * T + i1[j]) * 4;
    float u[4];
    for (int c = 0; c < 4; ++c) u[c] = __fadd_rn(__fmul_rn(w0[j], a[c]), __fmul_rn(w1[j], b[c]));
    const float nrm = sqrtf(u[2] * u[2] + u[3] * u[3]);
    float* o = out + (size_t)idx * 4;
    o[0] = u[0];
    o[1] = u[1];
    o[2] = u[2] / nrm;
    o[3] = u[3] / nrm;
}

// d_in[n][t] = sum over outputs j that tap t of  w * d_u[j],  d_u = adjoint of the pre-normalisation lerp
__global__ __launch_bounds__(256) void interp_traj_bwd_kernel(const float* __restrict__ in, const float* __restrict__ d_out,
                                                                int N, int T, int TO, int scale,
                                                                const int32_t* __restrict__ i0, const int32_t* __restrict__ i1,
                                                                const float* __restrict__ w0, const float* __restrict__ w1,
                                                                float* __restrict__ d_in) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * T) return;
    const int n = idx / T, t = idx - n * T;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    int jlo = (t - 1) * scale - scale, jhi = (t + 1) * scale + scale;
    jlo = jlo < 0 ? 0 : jlo;
    jhi = jhi > TO - 1 ? TO - 1 : jhi;
    for (int j = jlo; j <= jhi; ++j) {
        const int a0 = i0[j], a1 = i1[j];
        float w = 0.f;
        if (a0 == t) w += w0[j];
        if (a1 == t) w += w1[j];
        if (w == 0.f) continue;
        const float* a = in + ((size_t)n * T + a0) * 4;
        const float* b = in + ((size_t)n * T + a1) * 4;
        const float u2 = __fadd_rn(__fmul_rn(w0[j], a[2]), __fmul_rn(w1[j], b[2]));
        const float u3 = __fadd_rn(__fmul_rn(w0[j], a[3]), __fmul_rn(w1[j], b[3]));
        const float nrm = sqrtf(u2 * u2 + u3 * u3);
        const float o2 = u2 / nrm, o3 = u3 / nrm;
        const float* go = d_out + ((size_t)n * TO + j) * 4;
        const float dot = o2 * go[2] + o3 * go[3];
        g[0] += w * go[0];
        g[1] += w * go[1];
        g[2] += w * (go[2] - o2 * dot) / nrm;
        g[3] += w * (go[3] - o3 * dot) / nrm;
    }
    float* o = d_in + (size_t)idx * 4;
    for (int c = 0; c < 4; ++c) o[c] = g[c];
}

extern "C" int strive_interp_traj_fwd(const float* in, int32_t N, int32_t T, int32_t TO, const int32_t* i0, const int32_t* i1,
                                      const float* w0, const float* w1, float* out, strive_stream_t stream) {
    STRIVE_CHECK_ARG(in && i0 && i1 && w0 && w1 && out, "null argument");
    if (N <= 0 || TO <= 0) return 0;
    hipLaunchKernelGGL(interp_traj_fwd_kernel, dim3((N * TO + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, N, T, TO, i0,
                       i1, w0, w1, out);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_interp_traj_bwd(const float* in, const float* d_out, int32_t N, int32_t T, int32_t TO, int32_t scale,
                                      const int32_t* i0, const int32_t* i1, const float* w0, const float* w1, float* d_in,
                                      strive_stream_t stream) {
    STRIVE_CHECK_ARG(in && d_out && i0 && i1 && w0 && w1 && d_in, "null argument");
    STRIVE_CHECK_ARG(scale >= 1, "bad scale");
    if (N <= 0 || T <= 0) return 0;
    hipLaunchKernelGGL(interp_traj_bwd_kernel, dim3((N * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, d_out, N, T, TO,
                       scale, i0, i1, w0, w1, d_in);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

// =============================================================================================
// Rotated-rectangle IoU of vehicle boxes (the success / collision-metric tests of the optimisation loops:
// reference src/losses/adv_gen_nusc.py:517-623 builds shapely polygons from get_corners
// (src/datasets/nuscenes_utils.py:416-428) and evaluates intersection.area / union.area per (agent, step) in Python).
// One thread per box pair, float64: corners = R(atan2(hy, hx)) * (+-l/2, +-w/2) + (x, y); the first quadrilateral is
// clipped against the four edges of the second (Sutherland-Hodgman, <= 8 vertices) and the areas come from the
// shoelace formula.  A pair with a NaN in either pose yields NaN (the reference skips such frames).
// =============================================================================================
__device__ __forceinline__ void box_corners(const float* b, const float* lw, double cx[4], double cy[4]) {
    const double hl = 0.5 * (double)lw[0], hw = 0.5 * (double)lw[1];
    const double h = atan2((double)b[3], (double)b[2]);
    const double c = cos(h), s = sin(h);
    const double lx[4] = {-hl, hl, hl, -hl}, ly[4] = {-hw, -hw, hw, hw};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cx[i] = lx[i] * c - ly[i] * s + (double)b[0];
        cy[i] = lx[i] * s + ly[i] * c + (double)b[1];
    }
}

__device__ __forceinline__ double poly_area(const double* x, const double* y, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a += x[i] * y[j] - x[j] * y[i];
    }
    return 0.5 * fabs(a);
}

__global__ __launch_bounds__(256) void rect_iou_kernel(const float* __restrict__ box_a, const float* __restrict__ lw_a,
                                                         const float* __restrict__ box_b, const float* __restrict__ lw_b, int P,
                                                         double* __restrict__ iou) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* a = box_a + (size_t)p * 4;
    const float* b = box_b + (size_t)p * 4;
    bool bad = false;
    for (int i = 0; i < 4; ++i) bad = bad || (a[i] != a[i]) || (b[i] != b[i]);
    if (bad) {
        iou[p] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    double ax[4], ay[4], bx[4], by[4];
    box_corners(a, lw_a + (size_t)p * 2, ax, ay);
    box_corners(b, lw_b + (size_t)p * 2, bx, by);
    double px[10], py[10], qx[10], qy[10];
    int n = 4;
    for (int i = 0; i < 4; ++i) { px[i] = ax[i]; py[i] = ay[i]; }
    // both quadrilaterals are counter-clockwise: "inside" of edge (e0 -> e1) is the left side
    for (int e = 0; e < 4 && n > 0; ++e) {
        const double ex = bx[e], ey = by[e];
        const double dx = bx[(e + 1) & 3] - ex, dy = by[(e + 1) & 3] - ey;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const double si = dx * (py[i] - ey) - dy * (px[i] - ex);
            const double sj = dx * (py[j] - ey) - dy * (px[j] - ex);
            if (si >= 0.0) { qx[m] = px[i]; qy[m] = py[i]; ++m; }
            if ((si >= 0.0) != (sj >= 0.0)) {
                const double t = si / (si - sj);
                qx[m] = px[i] + t * (px[j] - px[i]);
                qy[m] = py[i] + t * (py[j] - py[i]);
                ++m;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
    }
    const double inter = n >= 3 ? poly_area(px, py, n) : 0.0;
    const double uni = poly_area(ax, ay, 4) + poly_area(bx, by, 4) - inter;
    iou[p] = inter / uni;
}

extern "C" int strive_rect_iou(const float* box_a, const float* lw_a, const float* box_b, const float* lw_b, int32_t P,
                               double* iou, strive_stream_t stream) {
    STRIVE_CHECK_ARG(box_a && lw_a && box_b && lw_b && iou, "null argument");
    if (P <= 0) return 0;
    hipLaunchKernelGGL(rect_iou_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, box_a, lw_a, box_b, lw_b, P, iou);
    STRIVE_CHECK_LAUNCH();
    return 0;
}


// =============================================================================================
// AvoidCollLoss in one call per direction (reference src/losses/adv_gen_nusc.py:264-341).
//   forward : interp_traj_fwd -> veh_coll_fwd -> coll_point (rows) -> avoid_partial -> avoid_final     (5 launches)
//   backward: avoid_grad (environment term, d_z) -> veh_coll_bwd<implicit> -> interp_traj_bwd            (3 launches)
// against ~60 + ~45 elementwise torch operators.  All sums are float64 partials per workgroup added in a fixed order
// (deterministic); the means divide by max(count, 1) like the reference's "[0.] when nothing collides" sentinel does.
// =============================================================================================
extern "C" int strive_coll_point_rows(const StriveMap* map, const float* fine, int32_t TO, const int32_t* agent_of,
                                      const float* lw, const int32_t* mapix, int32_t NE, int32_t gl, int32_t gw,
                                      const float* lin_l, const float* lin_w, float* out_pt, int32_t* out_cnt,
                                      strive_stream_t stream);

#define AV_BLOCKS 1024
#define NT_AV 256
#define AV_TERMS 6          // veh sum, veh count, env sum, env count, prior NLL sum, init-z sum

struct AvoidWs {
    float* fine;            // (NA,TO,4)
    float* pen;             // (TO,P)
    uint8_t* hit;           // (TO,P)
    uint8_t* amin;          // (TO,P)
    float* pt;              // (NE*TO,2)
    int32_t* cnt;           // (NE*TO)
    double* partial;        // (AV_BLOCKS, AV_TERMS)
    double* sums;           // (AV_TERMS)
    float* d_fine;          // (NA,TO,4) backward scratch
};

static size_t avoid_ws_bytes(size_t NA, size_t TO, size_t P, size_t NE) {
    size_t b = 0;
    b += strive_align_up(NA * TO * 16, 256) * 2;
    b += strive_align_up(TO * P * 4, 256) + 2 * strive_align_up(TO * P, 256);
    b += strive_align_up(NE * TO * 8, 256) + strive_align_up(NE * TO * 4, 256);
    b += strive_align_up((size_t)AV_BLOCKS * AV_TERMS * 8, 256) + 256;
    return b + 256;
}

static AvoidWs avoid_carve(void* p, size_t bytes, size_t NA, size_t TO, size_t P, size_t NE) {
    StriveArena ar(p, bytes);
    AvoidWs w;
    w.fine = ar.take<float>(NA * TO * 4);
    w.d_fine = ar.take<float>(NA * TO * 4);
    w.pen = ar.take<float>(TO * P);
    w.hit = ar.take<uint8_t>(TO * P);
    w.amin = ar.take<uint8_t>(TO * P);
    w.pt = ar.take<float>(NE * TO * 2);
    w.cnt = ar.take<int32_t>(NE * TO);
    w.partial = ar.take<double>((size_t)AV_BLOCKS * AV_TERMS);
    w.sums = ar.take<double>(AV_TERMS);
    return w;
}

struct AvoidArgs {
    int NA, TO, P, NE, NZ, D;
    float prior_den, init_den;
    const float* fine;
    const float* pen;
    const uint8_t* hit;
    const uint8_t* valid;
    const int32_t* env_agent;
    const float* pt;
    const float* pdist;
    const float *z, *mu, *var, *init_z;
    float w_veh, w_env, w_prior, w_init;
};

// off-road penalty of row (e, t): 1 - |c - p| / r for rows with a collision point (reference :384-403)
__device__ __forceinline__ bool env_row(const AvoidArgs& a, int row, float& dx, float& dy, float& d, float& pd) {
    const float px = a.pt[(size_t)row * 2], py = a.pt[(size_t)row * 2 + 1];
    if (!(px + py == px + py)) return false;      // NaN point = no collision point
    const int e = row / a.TO, t = row - e * a.TO;
    const float* c = a.fine + ((size_t)a.env_agent[e] * a.TO + t) * 4;
    dx = c[0] - px;
    dy = c[1] - py;
    d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    pd = a.pdist[e];
    return true;
}

__global__ __launch_bounds__(256) void avoid_partial_kernel(AvoidArgs a, double* __restrict__ partial) {
    __shared__ double s_red[4][AV_TERMS];
    const int tid = threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long g0 = (long long)blockIdx.x * blockDim.x + tid;
    double acc[AV_TERMS] = {0, 0, 0, 0, 0, 0};
    if (a.w_veh > 0.f) {
        // (t, slot) walked with 32-bit indices: slot fastest, so pen / hit reads are contiguous and `valid` stays in L1
        const int per_t = (a.P + NT_AV - 1) / NT_AV;                 // slot chunks of one time sample
        const long long nchunk = (long long)a.TO * per_t;
        for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
            const int t = (int)(ch / per_t), slot = (int)(ch - (long long)t * per_t) * NT_AV + tid;
            if (slot < a.P) {
                const size_t i = (size_t)t * a.P + slot;
                if (a.hit[i] && a.valid[slot]) { acc[0] += (double)a.pen[i]; acc[1] += 1.0; }
            }
        }
    }
    if (a.w_env > 0.f) {
        const long long n = (long long)a.NE * a.TO;
        for (long long i = g0; i < n; i += stride) {
            float dx, dy, d, pd;
            if (env_row(a, (int)i, dx, dy, d, pd)) { acc[2] += (double)(1.0f - d / pd); acc[3] += 1.0; }
        }
    }
    if (a.w_prior > 0.f || a.w_init > 0.f) {
        const long long n = (long long)a.NZ * a.D;
        for (long long i = g0; i < n; i += stride) {
            const float z = a.z[i];
            if (a.w_prior > 0.f) {
                // -log N(z; mu, var) per element as losses/common.py:26-41 evaluates it in fp32
                const float m = a.mu[i], v = a.var[i];
                const float dz = z - m;
                const float lp = -logf(sqrtf(v)) - 0.91893853320467267f - (dz * dz) / (2.0f * v);
                acc[4] -= (double)lp;
            }
            if (a.w_init > 0.f) {
                const float di = a.init_z[i] - z;
                acc[5] += (double)(di * di);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < AV_TERMS; ++k) acc[k] = wave_sum_d(acc[k]);
    if ((tid & 63) == 0)
        for (int k = 0; k < AV_TERMS; ++k) s_red[tid >> 6][k] = acc[k];
    __syncthreads();
    if (tid < AV_TERMS) partial[(size_t)blockIdx.x * AV_TERMS + tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
}

__global__ __launch_bounds__(64) void avoid_final_kernel(AvoidArgs a, const double* __restrict__ partial, int nblocks,
                                                           double* __restrict__ sums, float* __restrict__ out) {
    __shared__ double s[AV_TERMS];
    const int tid = threadIdx.x;
    // lanes stride over the workgroup partials, then a shuffle tree: a fixed order (deterministic), 16 steps instead of 1024
    for (int k = 0; k < AV_TERMS; ++k) {
        double v = 0.0;
        for (int b = tid; b < nblocks; b += 64) v += partial[(size_t)b * AV_TERMS + k];
        v = wave_sum_d(v);
        if (tid == 0) { s[k] = v; sums[k] = v; }
    }
    __syncthreads();
    if (tid == 0) {
        const double veh = s[0] / (s[1] < 1.0 ? 1.0 : s[1]);
        const double env = s[2] / (s[3] < 1.0 ? 1.0 : s[3]);
        const double pri = a.prior_den > 0.f ? s[4] / (double)a.prior_den : 0.0;
        const double ini = a.init_den > 0.f ? s[5] / (double)a.init_den : 0.0;
        double loss = 0.0;
        if (a.w_veh > 0.f) loss += (double)a.w_veh * veh;
        if (a.w_env > 0.f) loss += (double)a.w_env * env;
        if (a.w_prior > 0.f) loss += (double)a.w_prior * pri;
        if (a.w_init > 0.f) loss += (double)a.w_init * ini;
        out[0] = (float)loss;
        out[1] = (float)veh;
        out[2] = (float)env;
        out[3] = (float)pri;
        out[4] = (float)ini;
        out[5] = (float)s[1];
        out[6] = (float)s[3];
        out[7] = 0.f;
    }
}

// d_fine of the environment term -- every (agent, t) row is written, zeros where the agent takes no environment term or
// has no collision point, so the vehicle term can accumulate on top without a memset -- and d_z
__global__ __launch_bounds__(256) void avoid_grad_kernel(AvoidArgs a, const int32_t* __restrict__ env_of_agent,
                                                           const double* __restrict__ sums, const float* __restrict__ d_loss,
                                                           float* __restrict__ d_fine, float* __restrict__ d_z) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_rows = (long long)a.NA * a.TO;
    const float gl = d_loss[0];
    if (idx < n_rows) {
        const int ag = (int)(idx / a.TO), t = (int)(idx - (long long)ag * a.TO);
        float gx = 0.f, gy = 0.f;
        const int e = (a.w_env > 0.f && env_of_agent) ? env_of_agent[ag] : -1;
        if (e >= 0) {
            float dx, dy, d, pd;
            if (env_row(a, e * a.TO + t, dx, dy, d, pd) && d > 0.f) {
                const double c = sums[3];
                const float gs = (float)((double)gl * (double)a.w_env / (c < 1.0 ? 1.0 : c));
                const float k = -gs / (d * pd);      // pen = 1 - d / pd  ->  d pen / d c = -(c - p) / (d * pd)
                gx = k * dx;
                gy = k * dy;
            }
        }
        float* o = d_fine + (size_t)idx * 4;
        o[0] = gx; o[1] = gy; o[2] = 0.f; o[3] = 0.f;
        return;
    }
    const long long i = idx - n_rows;
    if (i < (long long)a.NZ * a.D) {
        const float z = a.z[i];
        float g = 0.f;
        if (a.w_prior > 0.f) g += gl * a.w_prior / a.prior_den * ((z - a.mu[i]) / a.var[i]);
        if (a.w_init > 0.f) g += gl * a.w_init / a.init_den * (2.0f * (z - a.init_z[i]));
        d_z[i] = g;
    }
}

static AvoidArgs avoid_args(const StriveScenes* sc, const StriveAvoidColl* h, const AvoidWs& w, int TO, const float* z,
                            const float* mu, const float* var) {
    AvoidArgs a;
    a.NA = sc->NA; a.TO = TO; a.P = h->P; a.NE = h->NE; a.NZ = h->NZ; a.D = h->D;
    a.prior_den = h->prior_den; a.init_den = h->init_den;
    a.fine = w.fine; a.pen = w.pen; a.hit = w.hit; a.valid = h->pair_valid;
    a.env_agent = h->env_agent; a.pt = w.pt; a.pdist = h->env_pdist;
    a.z = z; a.mu = mu; a.var = var; a.init_z = h->init_z;
    a.w_veh = h->w_veh; a.w_env = h->w_env; a.w_prior = h->w_prior; a.w_init = h->w_init;
    return a;
}

static int avoid_check(const StriveScenes* sc, const StriveAvoidColl* h, int T) {
    STRIVE_CHECK_ARG(sc && h, "null argument");
    STRIVE_CHECK_ARG(sc->NS == 1, "collision losses take one trajectory per agent");
    STRIVE_CHECK_ARG(T > 0 && h->scale >= 1 && h->P >= 0 && h->NE >= 0 && h->NZ >= 0 && h->D >= 0, "bad sizes");
    STRIVE_CHECK_ARG(h->i0 && h->i1 && h->w0 && h->w1, "null interpolation taps");
    if (h->w_veh > 0.f) STRIVE_CHECK_ARG(h->pair_off && h->cent_x && h->rad && h->pair_valid, "null vehicle-term constants");
    if (h->w_env > 0.f && h->NE > 0)
        STRIVE_CHECK_ARG(h->env_agent && h->env_of_agent && h->env_lw && h->env_mapix && h->env_pdist && h->lin_l && h->lin_w &&
                             h->gl > 0 && h->gw > 0,
                         "null environment-term constants");
    if (h->w_init > 0.f) STRIVE_CHECK_ARG(h->init_z, "null init_z");
    return 0;
}

extern "C" size_t strive_avoid_coll_workspace_bytes(const StriveScenes* sc, const StriveAvoidColl* h, int32_t T) {
    if (!sc || !h || T <= 0 || h->scale < 1) return 0;
    return avoid_ws_bytes((size_t)sc->NA, (size_t)T * h->scale, (size_t)h->P, (size_t)h->NE);
}

extern "C" int strive_avoid_coll_fwd(const StriveScenes* sc, const StriveMap* map, const StriveAvoidColl* h, const float* traj,
                                     int32_t T, const float* z, const float* mu, const float* var, float* out, void* ws,
                                     size_t ws_bytes, strive_stream_t stream) {
    if (int rc = avoid_check(sc, h, T)) return rc;
    STRIVE_CHECK_ARG(traj && out && ws, "null argument");
    if (h->w_prior > 0.f) STRIVE_CHECK_ARG(z && mu && var, "null latent tensors");
    if (h->w_init > 0.f) STRIVE_CHECK_ARG(z, "null latent tensors");
    if (h->w_env > 0.f && h->NE > 0) STRIVE_CHECK_ARG(map, "null map");
    const int NA = sc->NA, TO = T * h->scale;
    STRIVE_CHECK_ARG(ws_bytes >= avoid_ws_bytes(NA, TO, h->P, h->NE), "workspace too small");
    AvoidWs w = avoid_carve(ws, ws_bytes, NA, TO, h->P, h->NE);
    hipStream_t st = (hipStream_t)stream;
    if (NA > 0) {
        if (int rc = strive_interp_traj_fwd(traj, NA, T, TO, h->i0, h->i1, h->w0, h->w1, w.fine, stream)) return rc;
        if (h->w_veh > 0.f && h->P > 0)
            if (int rc = strive_veh_coll_fwd(sc, h->pair_off, h->P, w.fine, TO, h->cent_x, h->rad, h->buffer, w.pen, w.hit, w.amin,
                                             stream))
                return rc;
        if (h->w_env > 0.f && h->NE > 0)
            if (int rc = strive_coll_point_rows(map, w.fine, TO, h->env_agent, h->env_lw, h->env_mapix, h->NE, h->gl, h->gw,
                                                h->lin_l, h->lin_w, w.pt, w.cnt, stream))
                return rc;
    }
    AvoidArgs a = avoid_args(sc, h, w, TO, z, mu, var);
    if (h->P == 0) a.w_veh = 0.f;          // nothing to sum; the weight is re-applied below (mean of nothing = 0)
    if (h->NE == 0) a.w_env = 0.f;
    long long work = (long long)TO * h->P;
    if ((long long)h->NE * TO > work) work = (long long)h->NE * TO;
    if ((long long)h->NZ * h->D > work) work = (long long)h->NZ * h->D;
    int nb = (int)((work + 2047) / 2048);
    nb = nb < 1 ? 1 : (nb > AV_BLOCKS ? AV_BLOCKS : nb);
    hipLaunchKernelGGL(avoid_partial_kernel, dim3(nb), dim3(256), 0, st, a, w.partial);
    STRIVE_CHECK_LAUNCH();
    hipLaunchKernelGGL(avoid_final_kernel, dim3(1), dim3(64), 0, st, a, (const double*)w.partial, nb, w.sums, out);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_avoid_coll_bwd(const StriveScenes* sc, const StriveAvoidColl* h, const float* traj, int32_t T,
                                     const float* z, const float* mu, const float* var, const float* d_loss, void* ws,
                                     size_t ws_bytes, float* d_traj, float* d_z, strive_stream_t stream) {
    if (int rc = avoid_check(sc, h, T)) return rc;
    STRIVE_CHECK_ARG(traj && d_loss && ws && d_traj, "null argument");
    const int NA = sc->NA, TO = T * h->scale;
    STRIVE_CHECK_ARG(ws_bytes >= avoid_ws_bytes(NA, TO, h->P, h->NE), "workspace too small");
    if (NA == 0) return 0;
    AvoidWs w = avoid_carve(ws, ws_bytes, NA, TO, h->P, h->NE);
    hipStream_t st = (hipStream_t)stream;
    AvoidArgs a = avoid_args(sc, h, w, TO, z, mu, var);
    if (h->NE == 0) a.w_env = 0.f;
    const bool latent = (h->w_prior > 0.f || h->w_init > 0.f) && h->D > 0;
    if (latent) STRIVE_CHECK_ARG(z && d_z && (h->w_prior <= 0.f || (mu && var)), "null latent tensors");
    if (!latent) a.D = 0;
    const long long n = (long long)NA * TO + (long long)a.NZ * a.D;
    hipLaunchKernelGGL(avoid_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, h->env_of_agent,
                       (const double*)w.sums, d_loss, w.d_fine, d_z);
    STRIVE_CHECK_LAUNCH();
    if (h->w_veh > 0.f && h->P > 0) {
        VehImplicit im;
        im.hit = w.hit; im.valid = h->pair_valid; im.sums = w.sums; im.d_loss = d_loss; im.w = h->w_veh;
        const long long nv = (long long)NA * TO * VG;
        hipLaunchKernelGGL(veh_coll_bwd_kernel<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st,
                           veh_args(sc, h->pair_off, h->P, w.fine, TO, h->cent_x, h->rad, h->buffer), (const float*)nullptr,
                           (const uint8_t*)w.amin, w.d_fine, im);
        STRIVE_CHECK_LAUNCH();
    }
    return strive_interp_traj_bwd(traj, w.d_fine, NA, T, TO, h->scale, h->i0, h->i1, h->w0, h->w1, d_traj, stream);
}
