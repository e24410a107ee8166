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
    int cull;              // fused AvoidCollLoss only (pen / amin stay inside the call): a pair whose vehicle centres are further apart than
                           // the circles can reach is written as hit = 0 without its 25 circle distances, pen and amin left unwritten --
                           // every consumer reads them behind hit.  Exact: the cull distance carries 1 mm of slack over the bound
    const uint8_t* alive;  // (B) or null: pairs of a scene with alive == 0 are written as "not colliding" (AdvGenLoss with a
                           // quarantined scene, StriveAdvGen.scene_alive): they then enter neither a sum, a count nor a gradient
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
    // how far a circle centre of agent i lies from its pose: max |offset| x |heading vector| (the interpolated headings are not unit)
    const float* pi = a.traj + ((size_t)i * a.T + t) * 4;
    float reach_i = 0.f;
    if (a.cull) {
        for (int k = 0; k < NCIRC; ++k) reach_i = fmaxf(reach_i, fabsf(a.cent_x[i * NCIRC + k]));
        reach_i *= sqrtf(pi[2] * pi[2] + pi[3] * pi[3]);
    }
    for (int jl = sub; jl < n; jl += VG) {
        const int j = lo + jl;
        if (a.cull) {
            // the refine loop's loss has no scene structure (reference refine_traffic_optim.py: all NA^2 pairs): at 512 agents in 32
            // scenes 97 % of the pairs are hundreds of metres apart
            const float* pj = a.traj + ((size_t)j * a.T + t) * 4;
            float reach_j = 0.f;
            for (int k = 0; k < NCIRC; ++k) reach_j = fmaxf(reach_j, fabsf(a.cent_x[j * NCIRC + k]));
            reach_j *= sqrtf(pj[2] * pj[2] + pj[3] * pj[3]);
            const float far = ((ri + a.rad[j]) + a.buffer) + reach_i + reach_j + 1e-3f;
            const float dxc = pi[0] - pj[0], dyc = pi[1] - pj[1];
            if (dxc * dxc + dyc * dyc > far * far * 1.00001f) {       // (NaN poses compare false and take the full path)
                hit[base + jl] = 0;
                continue;
            }
        }
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
        hit[base + jl] = (j != i && dmin <= pd && !(a.alive && !a.alive[b])) ? 1 : 0;
        amin[base + jl] = (uint8_t)am;
    }
}

// Gradient of the masked MEAN of the penalties without a d_pen tensor (fused AvoidCollLoss): d_pen(slot) = g for the
// slots that are colliding and valid, g = d_loss * w / max(#such slots, 1) read from the forward's sums.
struct VehImplicit {
    const uint8_t* hit;       // (T,P)
    const uint8_t* valid;     // (P)
    const double* sums;       // [1] = number of colliding valid slots ([7] = of planner slots, see slot_ne)
    const float* d_loss;      // device scalar
    float w;
    // AdvGenLoss: slots that involve the scene's ego form a second masked mean, weighted per slot by rew of the non-ego
    // member (reference :192-204); slot_ne = that member's non-ego row, -1 for the other slots; null = one mean over all
    const int32_t* slot_ne;
    const float* rew;
    float w2;
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
    float gs = 0.f, gs2 = 0.f;
    if (IMPL) {
        const double c = im.sums[1];
        gs = (float)((double)im.d_loss[0] * (double)im.w / (c < 1.0 ? 1.0 : c));
        if (im.slot_ne) {
            const double c2 = im.sums[7];
            gs2 = (float)((double)im.d_loss[0] * (double)im.w2 / (c2 < 1.0 ? 1.0 : c2));
        }
    }
    auto slot_grad = [&](size_t s, int slot) -> float {
        if (!(im.hit[s] && im.valid[slot])) return 0.f;
        if (im.slot_ne) {
            const int ne = im.slot_ne[slot];
            if (ne >= 0) return gs2 * im.rew[ne];
        }
        return gs;
    };
    for (int jl = sub; live && jl < n; jl += VG) {
        const int j = lo + jl;
        if (j == i) continue;
        const size_t s_ij = (size_t)t * a.P + a.pair_off[i] + jl, s_ji = (size_t)t * a.P + a.pair_off[j] + il;
        const float gp_ij = IMPL ? slot_grad(s_ij, a.pair_off[i] + jl) : d_pen[s_ij];
        const float gp_ji = IMPL ? slot_grad(s_ji, a.pair_off[j] + il) : d_pen[s_ji];
        if (gp_ij == 0.f && gp_ji == 0.f) continue;            // (most pairs: the other agent's circles are not needed)
        float bx[NCIRC], by[NCIRC];
        circle_centres(a, j, t, bx, by);
        const float pd = (ri + a.rad[j]) + a.buffer;
        // pair (i, j): i is the first member
        {
            const size_t s = s_ij;
            const float gp = gp_ij;
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
            const size_t s = s_ji;
            const float gp = gp_ji;
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
    a.traj = traj; a.cent_x = cent_x; a.rad = rad; a.buffer = buffer; a.alive = nullptr; a.cull = 0;
    return a;
}

static int veh_coll_fwd_masked(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj, int32_t T,
                               const float* cent_x, const float* rad, float buffer, const uint8_t* scene_alive, float* pen,
                               uint8_t* hit, uint8_t* amin, strive_stream_t stream, int cull = 0) {
    STRIVE_CHECK_ARG(sc && pair_off && traj && cent_x && rad && pen && hit && amin, "null argument");
    STRIVE_CHECK_ARG(sc->NS == 1, "collision losses take one trajectory per agent");
    const long long n = (long long)sc->NA * T * VG;
    if (n <= 0) return 0;
    VehArgs a = veh_args(sc, pair_off, P, traj, T, cent_x, rad, buffer);
    a.alive = scene_alive;
    a.cull = cull;
    hipLaunchKernelGGL(veh_coll_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, pen, hit, amin);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_veh_coll_fwd(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj, int32_t T,
                                   const float* cent_x, const float* rad, float buffer, float* pen, uint8_t* hit,
                                   uint8_t* amin, strive_stream_t stream) {
    return veh_coll_fwd_masked(sc, pair_off, P, traj, T, cent_x, rad, buffer, nullptr, pen, hit, amin, stream);
}

extern "C" int strive_veh_coll_bwd(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj, int32_t T,
                                   const float* cent_x, const float* rad, float buffer, const float* d_pen,
                                   const uint8_t* amin, float* d_traj, strive_stream_t stream) {
    STRIVE_CHECK_ARG(sc && pair_off && traj && cent_x && rad && d_pen && amin && d_traj, "null argument");
    STRIVE_CHECK_ARG(sc->NS == 1, "collision losses take one trajectory per agent");
    const long long n = (long long)sc->NA * T * VG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(veh_coll_bwd_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       veh_args(sc, pair_off, P, traj, T, cent_x, rad, buffer), d_pen, amin, d_traj, VehImplicit{});
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
    const uint8_t* scene_alive;     // (B) or null; AdvGenLoss only (a quarantined scene's rows count as "no collision point")
    const int32_t* scene_of;        // (NA)
};

// off-road penalty of row (e, t): 1 - |c - p| / r for rows with a collision point (reference :384-403)
__device__ __forceinline__ bool env_row(const AvoidArgs& a, int row, float& dx, float& dy, float& d, float& pd) {
    const float px = a.pt[(size_t)row * 2], py = a.pt[(size_t)row * 2 + 1];
    if (!(px + py == px + py)) return false;      // NaN point = no collision point
    const int e = row / a.TO, t = row - e * a.TO;
    if (a.scene_alive && !a.scene_alive[a.scene_of[a.env_agent[e]]]) return false;
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
    a.scene_alive = nullptr; a.scene_of = sc->scene_of;
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
            if (int rc = veh_coll_fwd_masked(sc, h->pair_off, h->P, w.fine, TO, h->cent_x, h->rad, h->buffer, nullptr, w.pen, w.hit, w.amin,
                                             stream, /*cull=*/1))
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
        im.slot_ne = nullptr; im.rew = nullptr; im.w2 = 0.f;
        const long long nv = (long long)NA * TO * VG;
        hipLaunchKernelGGL(veh_coll_bwd_kernel<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st,
                           veh_args(sc, h->pair_off, h->P, w.fine, TO, h->cent_x, h->rad, h->buffer), (const float*)nullptr,
                           (const uint8_t*)w.amin, w.d_fine, im);
        STRIVE_CHECK_LAUNCH();
    }
    return strive_interp_traj_bwd(traj, w.d_fine, NA, T, TO, h->scale, h->i0, h->i1, h->w0, h->w1, d_traj, stream);
}


// =============================================================================================
// AdvGenLoss in one call per direction (reference src/losses/adv_gen_nusc.py:53-262).
//   forward : interp -> veh_coll_fwd -> coll_point (non-ego rows) -> adv_crash (one workgroup per scene: distances to the
//             target, behind test, per-scene soft-min, crash term, prior re-weights) -> adv_partial -> adv_final   (6 launches)
//   backward: adv_grad (environment term, d_z) -> veh_coll_bwd<implicit, per-slot weights> -> interp_bwd -> adv_crash_bwd
// against ~140 + ~110 elementwise / scatter torch operators.  The "every attacker is always behind its target" escape
// (:120-123) is a property of the whole batch: adv_crash evaluates the soft-min with and without the behind mask and leaves a
// per-scene flag; the kernels after it form the batch-wide AND themselves and pick the variant.
// =============================================================================================
#define ADV_TERMS 8         // veh sum, veh count, env sum, env count, weighted prior sum, weighted init sum, planner sum, planner count
#define ADV_SUMS 10         // sums[]: the terms above, then [8] latent rows and [9] scenes that are alive (all of them without a mask)
#define ADV_MAXE 1024       // (non-ego agents of a scene) x (crash time samples) the crash kernel keeps in LDS

struct AdvWs {
    AvoidWs av;
    double* partial;        // (AV_BLOCKS, ADV_TERMS)
    double* sums;           // (ADV_TERMS)
    float* soft2;           // (2, NE, NT)   variant 0 = with the behind mask, 1 = without
    float* rew2;            // (2, NE)
    float* crash2;          // (2, B)
    int32_t* scene_flag;    // (B) 1 = every non-ego agent of the scene is always behind
    int32_t* sel;           // (1) chosen variant
    float* rew_sel;         // (NE) re-weights of the chosen variant (the backward kernels read these)
};

static size_t adv_ws_bytes(size_t NA, size_t TO, size_t P, size_t NE, size_t NT, size_t B) {
    return avoid_ws_bytes(NA, TO, P, NE) + strive_align_up((size_t)AV_BLOCKS * ADV_TERMS * 8, 256) + 256 +
           strive_align_up(2 * NE * NT * 4, 256) + strive_align_up(2 * NE * 4, 256) + strive_align_up(2 * B * 4, 256) +
           strive_align_up(B * 4, 256) + strive_align_up(NE * 4, 256) + 256 + 256;
}

static AdvWs adv_carve(void* p, size_t bytes, size_t NA, size_t TO, size_t P, size_t NE, size_t NT, size_t B) {
    const size_t ab = avoid_ws_bytes(NA, TO, P, NE);
    AdvWs w;
    w.av = avoid_carve(p, ab, NA, TO, P, NE);
    StriveArena ar((char*)p + ab, bytes - ab);
    w.partial = ar.take<double>((size_t)AV_BLOCKS * ADV_TERMS);
    w.sums = ar.take<double>(ADV_SUMS);
    w.soft2 = ar.take<float>(2 * NE * NT);
    w.rew2 = ar.take<float>(2 * NE);
    w.crash2 = ar.take<float>(2 * B);
    w.scene_flag = ar.take<int32_t>(B);
    w.sel = ar.take<int32_t>(1);
    w.rew_sel = ar.take<float>(NE);
    return w;
}

struct AdvArgs {
    AvoidArgs a;
    int B, T, t0, NT, use_infront;
    float infront;
    const float* traj;          // (NA,T,4)
    const float* tgt;           // (B,T,4)
    const int32_t* ne_ptr;
    const int32_t* nonego;      // (NE) agent index
    const int32_t* slot_ne;
    const uint8_t* atk_mask;
    const uint8_t* scene_alive;     // (B) or null
    float w_crash, w_plan, w_prior_atk, w_init_atk;
    float* soft2;
    float* rew2;
    float* crash2;
    int32_t* scene_flag;
    float* rew_sel;
};

// block-wide reductions of a 256-thread workgroup in a fixed order (wave tree, then waves 0..3)
__device__ __forceinline__ double block_sum_d(double v, double* s_red, int tid) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ float block_max_f(float v, float* s_red, int tid) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

// one workgroup per scene
__global__ __launch_bounds__(256) void adv_crash_kernel(AdvArgs A) {
    __shared__ float s_d[ADV_MAXE];
    __shared__ float s_s[ADV_MAXE];
    __shared__ int s_nb[64];             // per non-ego agent: time samples at which it is NOT behind the target
    __shared__ double s_redd[4];
    __shared__ float s_redf[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int a0 = A.ne_ptr[b], n = A.ne_ptr[b + 1] - a0, NT = A.NT, E = n * NT;
    if (A.scene_alive && !A.scene_alive[b]) {
        // quarantined scene (its planner failed, StriveAdvGen.scene_alive): no crash term, no soft-min weights, re-weights 1, and
        // like an empty scene it does not veto the batch-wide "everybody is behind" test
        for (int v = 0; v < 2; ++v) {
            for (int e = tid; e < E; e += 256) A.soft2[((size_t)v * A.a.NE + a0) * NT + e] = 0.f;
            for (int al = tid; al < n; al += 256) A.rew2[(size_t)v * A.a.NE + a0 + al] = 1.0f;
            if (tid == 0) A.crash2[(size_t)v * A.B + b] = 0.f;
        }
        if (tid == 0) A.scene_flag[b] = 1;
        return;
    }
    if (tid < 64) s_nb[tid] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        const int al = e / NT, t = e - al * NT;
        const float* p = A.traj + ((size_t)A.nonego[a0 + al] * A.T + A.t0 + t) * 4;
        const float* q = A.tgt + ((size_t)b * A.T + A.t0 + t) * 4;
        const float dx = p[0] - q[0], dy = p[1] - q[1];
        const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
        s_d[e] = d;
        if (A.use_infront) {
            // behind <=> (atk - tgt) / |atk - tgt| . heading(tgt) < threshold   (reference :646-673); NaN compares false
            const float dot = __fadd_rn(__fmul_rn(dx / d, q[2]), __fmul_rn(dy / d, q[3]));
            if (!(dot < A.infront)) atomicAdd(&s_nb[al], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int all_always = A.use_infront ? 1 : 0;
        for (int al = 0; al < n; ++al) all_always &= (s_nb[al] == 0);
        A.scene_flag[b] = all_always;            // (an empty scene leaves 1: it does not veto the batch-wide test)
    }
    for (int v = 0; v < 2; ++v) {
        float mx = -3.0e38f;
        bool any = false;
        for (int e = tid; e < E; e += 256) {
            const int al = e / NT;
            const bool masked = (v == 0 && A.use_infront && s_nb[al] == 0) || (A.atk_mask && !A.atk_mask[a0 + al]);
            const float d = s_d[e];
            if (!masked && !(d != d)) { mx = fmaxf(mx, -d); any = true; }
        }
        mx = block_max_f(mx, s_redf, tid);
        double den = 0.0;
        for (int e = tid; e < E; e += 256) {
            const int al = e / NT;
            const bool masked = (v == 0 && A.use_infront && s_nb[al] == 0) || (A.atk_mask && !A.atk_mask[a0 + al]);
            const float ex = masked ? 0.f : expf(-s_d[e] - mx);
            s_s[e] = ex;
            den += (double)ex;
        }
        (void)any;
        den = block_sum_d(den, s_redd, tid);
        const bool dead = !(den > 0.0) || mx <= -3.0e38f;          // nothing unmasked (or NaN distances): all weights 0 (:134-135)
        double cr = 0.0;
        for (int e = tid; e < E; e += 256) {
            const float sft = dead ? 0.f : (float)((double)s_s[e] / den);
            s_s[e] = sft;
            const float d = s_d[e];
            cr += (double)(sft * (d * d));
            A.soft2[((size_t)v * A.a.NE + a0) * NT + e] = sft;
        }
        cr = block_sum_d(cr, s_redd, tid);
        if (tid == 0) A.crash2[(size_t)v * A.B + b] = (float)cr;
        for (int al = tid; al < n; al += 256) {
            float sm = 0.f;
            for (int t = 0; t < NT; ++t) sm += s_s[al * NT + t];
            A.rew2[(size_t)v * A.a.NE + a0 + al] = 1.0f - sm;
        }
        __syncthreads();
    }
}

// batch-wide "everybody is always behind" test -> variant 1 (no behind mask); every workgroup that needs it forms it itself
__device__ __forceinline__ int adv_select(const int32_t* scene_flag, int B, int use_infront, int tid) {
    __shared__ int s_all;
    if (tid == 0) s_all = use_infront ? 1 : 0;
    __syncthreads();
    int all = 1;
    for (int b = tid; b < B; b += blockDim.x) all &= scene_flag[b];
    if (!all) s_all = 0;                 // (every writer stores the same value)
    __syncthreads();
    return s_all;
}

__global__ __launch_bounds__(256) void adv_partial_kernel(AdvArgs A, double* __restrict__ partial) {
    __shared__ double s_red[4][ADV_TERMS];
    const AvoidArgs& a = A.a;
    const int tid = threadIdx.x;
    const int sel = adv_select(A.scene_flag, A.B, A.use_infront, tid);
    const float* rew = A.rew2 + (size_t)sel * a.NE;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long g0 = (long long)blockIdx.x * blockDim.x + tid;
    double acc[ADV_TERMS] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a.w_veh > 0.f || A.w_plan > 0.f) {
        const int per_t = (a.P + NT_AV - 1) / NT_AV;
        const long long nchunk = (long long)a.TO * per_t;
        for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
            const int t = (int)(ch / per_t), slot = (int)(ch - (long long)t * per_t) * NT_AV + tid;
            if (slot < a.P) {
                const size_t i = (size_t)t * a.P + slot;
                if (a.hit[i] && a.valid[slot]) {
                    const int ne = A.slot_ne[slot];
                    if (ne >= 0) {
                        if (A.w_plan > 0.f) { acc[6] += (double)(a.pen[i] * rew[ne]); acc[7] += 1.0; }
                    } else if (a.w_veh > 0.f) {
                        acc[0] += (double)a.pen[i];
                        acc[1] += 1.0;
                    }
                }
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
            const int r = (int)(i / a.D);
            if (A.scene_alive && !A.scene_alive[a.scene_of[A.nonego[r]]]) continue;
            const float rw = rew[r];
            const float z = a.z[i];
            if (a.w_prior > 0.f) {
                const float m = a.mu[i], v = a.var[i];
                const float dz = z - m;
                const float lp = -logf(sqrtf(v)) - 0.91893853320467267f - (dz * dz) / (2.0f * v);
                acc[4] -= (double)(lp * (rw * a.w_prior + (1.0f - rw) * A.w_prior_atk));
            }
            if (a.w_init > 0.f) {
                const float di = a.init_z[i] - z;
                acc[5] += (double)((di * di) * (rw * a.w_init + (1.0f - rw) * A.w_init_atk));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < ADV_TERMS; ++k) acc[k] = wave_sum_d(acc[k]);
    if ((tid & 63) == 0)
        for (int k = 0; k < ADV_TERMS; ++k) s_red[tid >> 6][k] = acc[k];
    __syncthreads();
    if (tid < ADV_TERMS) partial[(size_t)blockIdx.x * ADV_TERMS + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

__global__ __launch_bounds__(64) void adv_final_kernel(AdvArgs A, const double* __restrict__ partial, int nblocks,
                                                         double* __restrict__ sums, int32_t* __restrict__ sel_out,
                                                         float* __restrict__ out, float* __restrict__ soft,
                                                         float* __restrict__ rew) {
    __shared__ double s[ADV_TERMS + 3];
    const AvoidArgs& a = A.a;
    const int tid = threadIdx.x;
    const int sel = adv_select(A.scene_flag, A.B, A.use_infront, tid);
    for (int k = 0; k < ADV_TERMS; ++k) {
        double v = 0.0;
        for (int b = tid; b < nblocks; b += 64) v += partial[(size_t)b * ADV_TERMS + k];
        v = wave_sum_d(v);
        if (tid == 0) { s[k] = v; sums[k] = v; }
    }
    {
        double v = 0.0, rows = 0.0, scenes = 0.0;
        for (int b = tid; b < A.B; b += 64) {
            v += (double)A.crash2[(size_t)sel * A.B + b];
            if (!A.scene_alive || A.scene_alive[b]) { scenes += 1.0; rows += (double)(A.ne_ptr[b + 1] - A.ne_ptr[b]); }
        }
        v = wave_sum_d(v);
        rows = wave_sum_d(rows);
        scenes = wave_sum_d(scenes);
        if (tid == 0) { s[ADV_TERMS] = v; s[ADV_TERMS + 1] = rows; s[ADV_TERMS + 2] = scenes; sums[8] = rows; sums[9] = scenes; }
    }
    // the selected soft-min weights and re-weights, for the caller (return_mins, logging)
    for (long long i = tid; i < (long long)a.NE * A.NT; i += 64) soft[i] = A.soft2[(size_t)sel * a.NE * A.NT + i];
    for (int i = tid; i < a.NE; i += 64) {
        const float rv = A.rew2[(size_t)sel * a.NE + i];
        rew[i] = rv;
        A.rew_sel[i] = rv;
    }
    __syncthreads();
    if (tid == 0) {
        const double veh = s[0] / (s[1] < 1.0 ? 1.0 : s[1]);
        const double env = s[2] / (s[3] < 1.0 ? 1.0 : s[3]);
        const double plan = s[6] / (s[7] < 1.0 ? 1.0 : s[7]);
        // means over the latent rows / scenes that are in the batch: all of them, or the alive ones (= the batch rebuilt
        // without the quarantined scenes, which is what the reference's batch_size-1 runs amount to)
        const double pri = s[ADV_TERMS + 1] > 0.0 ? s[4] / s[ADV_TERMS + 1] : 0.0;
        const double ini = s[5];
        const double crash = s[ADV_TERMS + 2] > 0.0 ? s[ADV_TERMS] / s[ADV_TERMS + 2] : 0.0;
        double loss = 0.0;
        if (a.w_init > 0.f) loss += ini;
        if (a.w_prior > 0.f) loss += pri;
        if (a.w_veh > 0.f) loss += (double)a.w_veh * veh;
        if (A.w_plan > 0.f) loss += (double)A.w_plan * plan;
        if (a.w_env > 0.f) loss += (double)a.w_env * env;
        if (A.w_crash > 0.f) loss += (double)A.w_crash * crash;
        sel_out[0] = sel;
        out[0] = (float)loss;
        out[1] = (float)veh;
        out[2] = (float)plan;
        out[3] = (float)env;
        out[4] = (float)pri;
        out[5] = (float)ini;
        out[6] = (float)crash;
        out[7] = (float)s[1];
        out[8] = (float)s[7];
        out[9] = (float)s[3];
        out[10] = (float)sel;
        for (int k = 11; k < 16; ++k) out[k] = 0.f;
    }
}

// d_fine of the environment term (all rows written, like avoid_grad_kernel) and d_z
__global__ __launch_bounds__(256) void adv_grad_kernel(AdvArgs A, const int32_t* __restrict__ env_of_agent,
                                                         const double* __restrict__ sums, const int32_t* __restrict__ sel_p,
                                                         const float* __restrict__ d_loss, float* __restrict__ d_fine,
                                                         float* __restrict__ d_z) {
    const AvoidArgs& a = A.a;
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
                const float k = -gs / (d * pd);
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
        const int r = (int)(i / a.D);
        const float rw = A.rew_sel[r];
        const float z = a.z[i];
        float g = 0.f;
        if (!(A.scene_alive && !A.scene_alive[a.scene_of[A.nonego[r]]])) {
            if (a.w_prior > 0.f) g += gl * (rw * a.w_prior + (1.0f - rw) * A.w_prior_atk) / (float)sums[8] * ((z - a.mu[i]) / a.var[i]);
            if (a.w_init > 0.f) g += gl * (rw * a.w_init + (1.0f - rw) * A.w_init_atk) * (2.0f * (z - a.init_z[i]));
        }
        d_z[i] = g;
    }
}

// crash term: L = w/B sum_b sum_e s_e d_e^2 with s = softmin(d) over the scene's unmasked entries
//   dL/dd_e = w/B * s_e * (2 d_e + C_b - d_e^2),  C_b = sum_e s_e d_e^2;   dd/d(atk xy) = (atk - tgt) / d = -dd/d(tgt xy)
// one workgroup per scene: adds into d_traj (non-ego rows, time >= t0; interp_bwd has written them) and writes d_tgt
__global__ __launch_bounds__(256) void adv_crash_bwd_kernel(AdvArgs A, const int32_t* __restrict__ sel_p, const double* __restrict__ sums,
                                                              const float* __restrict__ d_loss, float* __restrict__ d_traj,
                                                              float* __restrict__ d_tgt) {
    __shared__ float s_gx[ADV_MAXE], s_gy[ADV_MAXE];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int sel = sel_p[0];
    const int a0 = A.ne_ptr[b], n = A.ne_ptr[b + 1] - a0, NT = A.NT, E = n * NT;
    const float C = A.crash2[(size_t)sel * A.B + b];
    const float gw = A.w_crash > 0.f ? d_loss[0] * A.w_crash / (float)sums[9] : 0.f;      // (a quarantined scene has soft == 0 throughout)
    for (int e = tid; e < E; e += 256) {
        const int al = e / NT, t = e - al * NT;
        float* gp = d_traj + ((size_t)A.nonego[a0 + al] * A.T + A.t0 + t) * 4;
        const float* p = A.traj + ((size_t)A.nonego[a0 + al] * A.T + A.t0 + t) * 4;
        const float* q = A.tgt + ((size_t)b * A.T + A.t0 + t) * 4;
        const float dx = p[0] - q[0], dy = p[1] - q[1];
        const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
        const float sft = A.soft2[((size_t)sel * A.a.NE + a0) * NT + e];
        float gx = 0.f, gy = 0.f;
        if (sft != 0.f && d > 0.f) {
            const float gd = gw * sft * (2.0f * d + C - d * d);
            gx = gd * dx / d;
            gy = gd * dy / d;
        }
        gp[0] += gx;
        gp[1] += gy;
        s_gx[e] = gx;
        s_gy[e] = gy;
    }
    __syncthreads();
    for (int t = tid; t < A.T; t += 256) {
        float sx = 0.f, sy = 0.f;
        if (t >= A.t0)
            for (int al = 0; al < n; ++al) { sx += s_gx[al * NT + t - A.t0]; sy += s_gy[al * NT + t - A.t0]; }
        float* o = d_tgt + ((size_t)b * A.T + t) * 4;
        o[0] = -sx; o[1] = -sy; o[2] = 0.f; o[3] = 0.f;
    }
}

static int adv_check(const StriveScenes* sc, const StriveAdvGen* h, int T) {
    STRIVE_CHECK_ARG(sc && h, "null argument");
    if (int rc = avoid_check(sc, &h->base, T)) return rc;
    STRIVE_CHECK_ARG(h->ne_ptr && h->slot_ne && h->base.env_agent && h->base.env_of_agent, "null scene tables");
    if (h->w_plan > 0.f) STRIVE_CHECK_ARG(h->base.pair_off && h->base.cent_x && h->base.rad && h->base.pair_valid, "null vehicle-term constants");
    STRIVE_CHECK_ARG(h->base.NE == sc->NA - sc->B && h->base.NZ == h->base.NE, "one latent row per non-ego agent");
    STRIVE_CHECK_ARG(h->t0 >= 0 && h->t0 < T, "bad crash_loss_min_time");
    STRIVE_CHECK_ARG(h->w_crash > 0.f, "the fused AdvGenLoss needs the crash term (its soft-min defines the re-weights)");
    STRIVE_CHECK_ARG(sc->max_n >= 1 && (sc->max_n - 1) * (T - h->t0) <= ADV_MAXE && sc->max_n - 1 <= 64,
                     "scene too large for the crash kernel's LDS tables");
    return 0;
}

static AdvArgs adv_args(const StriveScenes* sc, const StriveAdvGen* h, const AdvWs& w, int T, const float* traj, const float* tgt,
                        const float* z, const float* mu, const float* var) {
    AdvArgs A;
    A.a = avoid_args(sc, &h->base, w.av, T * h->base.scale, z, mu, var);
    A.B = sc->B; A.T = T; A.t0 = h->t0; A.NT = T - h->t0; A.use_infront = h->use_infront; A.infront = h->infront;
    A.traj = traj; A.tgt = tgt; A.ne_ptr = h->ne_ptr; A.nonego = h->base.env_agent; A.slot_ne = h->slot_ne; A.atk_mask = h->atk_mask;
    A.scene_alive = h->scene_alive; A.a.scene_alive = h->scene_alive;
    A.w_crash = h->w_crash; A.w_plan = h->w_plan; A.w_prior_atk = h->w_prior_atk; A.w_init_atk = h->w_init_atk;
    A.soft2 = w.soft2; A.rew2 = w.rew2; A.crash2 = w.crash2; A.scene_flag = w.scene_flag; A.rew_sel = w.rew_sel;
    return A;
}

extern "C" size_t strive_adv_gen_workspace_bytes(const StriveScenes* sc, const StriveAdvGen* h, int32_t T) {
    if (!sc || !h || T <= 0 || h->base.scale < 1 || h->t0 < 0 || h->t0 >= T) return 0;
    return adv_ws_bytes((size_t)sc->NA, (size_t)T * h->base.scale, (size_t)h->base.P, (size_t)h->base.NE, (size_t)(T - h->t0),
                        (size_t)sc->B);
}

extern "C" int strive_adv_gen_fwd(const StriveScenes* sc, const StriveMap* map, const StriveAdvGen* h, const float* traj,
                                  const float* tgt, int32_t T, const float* z, const float* mu, const float* var, float* out,
                                  float* soft, float* rew, void* ws, size_t ws_bytes, strive_stream_t stream) {
    if (int rc = adv_check(sc, h, T)) return rc;
    STRIVE_CHECK_ARG(traj && tgt && out && soft && rew && ws && z && mu && var, "null argument");
    const StriveAvoidColl* hb = &h->base;
    if (hb->w_env > 0.f && hb->NE > 0) STRIVE_CHECK_ARG(map, "null map");
    const int NA = sc->NA, TO = T * hb->scale, NT = T - h->t0;
    STRIVE_CHECK_ARG(ws_bytes >= adv_ws_bytes(NA, TO, hb->P, hb->NE, NT, sc->B), "workspace too small");
    AdvWs w = adv_carve(ws, ws_bytes, NA, TO, hb->P, hb->NE, NT, sc->B);
    hipStream_t st = (hipStream_t)stream;
    const bool veh = (hb->w_veh > 0.f || h->w_plan > 0.f) && hb->P > 0;
    if (NA > 0) {
        if (int rc = strive_interp_traj_fwd(traj, NA, T, TO, hb->i0, hb->i1, hb->w0, hb->w1, w.av.fine, stream)) return rc;
        if (veh)
            if (int rc = veh_coll_fwd_masked(sc, hb->pair_off, hb->P, w.av.fine, TO, hb->cent_x, hb->rad, hb->buffer, h->scene_alive,
                                             w.av.pen, w.av.hit, w.av.amin, stream))
                return rc;
        if (hb->w_env > 0.f && hb->NE > 0)
            if (int rc = strive_coll_point_rows(map, w.av.fine, TO, hb->env_agent, hb->env_lw, hb->env_mapix, hb->NE, hb->gl, hb->gw,
                                                hb->lin_l, hb->lin_w, w.av.pt, w.av.cnt, stream))
                return rc;
    }
    AdvArgs A = adv_args(sc, h, w, T, traj, tgt, z, mu, var);
    if (!veh) { A.a.w_veh = 0.f; A.w_plan = 0.f; }
    if (hb->NE == 0) A.a.w_env = 0.f;
    if (sc->B > 0) {
        hipLaunchKernelGGL(adv_crash_kernel, dim3(sc->B), dim3(256), 0, st, A);
        STRIVE_CHECK_LAUNCH();
    }
    long long work = (long long)TO * hb->P;
    if ((long long)hb->NE * TO > work) work = (long long)hb->NE * TO;
    if ((long long)hb->NZ * hb->D > work) work = (long long)hb->NZ * hb->D;
    int nb = (int)((work + 2047) / 2048);
    nb = nb < 1 ? 1 : (nb > AV_BLOCKS ? AV_BLOCKS : nb);
    hipLaunchKernelGGL(adv_partial_kernel, dim3(nb), dim3(256), 0, st, A, w.partial);
    STRIVE_CHECK_LAUNCH();
    hipLaunchKernelGGL(adv_final_kernel, dim3(1), dim3(64), 0, st, A, (const double*)w.partial, nb, w.sums, w.sel, out, soft, rew);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_adv_gen_bwd(const StriveScenes* sc, const StriveAdvGen* h, const float* traj, const float* tgt, int32_t T,
                                  const float* z, const float* mu, const float* var, const float* d_loss, void* ws,
                                  size_t ws_bytes, float* d_traj, float* d_tgt, float* d_z, strive_stream_t stream) {
    if (int rc = adv_check(sc, h, T)) return rc;
    STRIVE_CHECK_ARG(traj && tgt && d_loss && ws && d_traj && d_tgt && d_z && z && mu && var, "null argument");
    const StriveAvoidColl* hb = &h->base;
    const int NA = sc->NA, TO = T * hb->scale, NT = T - h->t0;
    STRIVE_CHECK_ARG(ws_bytes >= adv_ws_bytes(NA, TO, hb->P, hb->NE, NT, sc->B), "workspace too small");
    if (NA == 0) return 0;
    AdvWs w = adv_carve(ws, ws_bytes, NA, TO, hb->P, hb->NE, NT, sc->B);
    hipStream_t st = (hipStream_t)stream;
    AdvArgs A = adv_args(sc, h, w, T, traj, tgt, z, mu, var);
    const bool veh = (hb->w_veh > 0.f || h->w_plan > 0.f) && hb->P > 0;
    if (hb->NE == 0) A.a.w_env = 0.f;
    const long long n = (long long)NA * TO + (long long)A.a.NZ * A.a.D;
    hipLaunchKernelGGL(adv_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, hb->env_of_agent,
                       (const double*)w.sums, (const int32_t*)w.sel, d_loss, w.av.d_fine, d_z);
    STRIVE_CHECK_LAUNCH();
    if (veh) {
        VehImplicit im;
        im.hit = w.av.hit; im.valid = hb->pair_valid; im.sums = w.sums; im.d_loss = d_loss; im.w = hb->w_veh;
        im.slot_ne = h->slot_ne; im.rew = w.rew_sel; im.w2 = h->w_plan;
        const long long nv = (long long)NA * TO * VG;
        hipLaunchKernelGGL(veh_coll_bwd_kernel<true>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st,
                           veh_args(sc, hb->pair_off, hb->P, w.av.fine, TO, hb->cent_x, hb->rad, hb->buffer), (const float*)nullptr,
                           (const uint8_t*)w.av.amin, w.av.d_fine, im);
        STRIVE_CHECK_LAUNCH();
    }
    if (int rc = strive_interp_traj_bwd(traj, w.av.d_fine, NA, T, TO, hb->scale, hb->i0, hb->i1, hb->w0, hb->w1, d_traj, stream)) return rc;
    if (sc->B > 0) {
        hipLaunchKernelGGL(adv_crash_bwd_kernel, dim3(sc->B), dim3(256), 0, st, A, (const int32_t*)w.sel, (const double*)w.sums, d_loss, d_traj,
                           d_tgt);
        STRIVE_CHECK_LAUNCH();
    }
    return 0;
}
