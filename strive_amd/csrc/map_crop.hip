// Rasterised-map lookups: per-agent crop (get_map_obs) and off-road collision point (get_coll_point).
//
// Both are HBM/L2 gathers of single bytes addressed through a rotated grid.  The coordinate pipeline
// follows the reference bit for bit (reference src/datasets/nuscenes_utils.py:205-264, 334-390):
//   world = (l*cos - w*sin) + x   in fp32, every product/sum individually rounded (no FMA contraction)
//   NaN   -> 0                    (crop only)
//   pixel = rint(double(world) / dx)   in float64, round half to even
//   outside the raster -> pixel (0,0)
// but never materialises the (N,C,L,W,2) fp32/fp64/int64 coordinate tensors the reference builds
// (10 MiB per agent per step); each thread derives its pixel index in registers and reads C bytes.
#include "common.h"
#include "crop_dev.h"

// ---------------------------------------------------------------------------------------------
// crop: grid (ceil(L/ROWS), N), block 256 = one thread per crop column w, looping over ROWS rows.
// Writes are coalesced along w; the raster reads follow the rotated grid (a 0.3 m step = 1.2 px).
// ---------------------------------------------------------------------------------------------
#define CROP_ROWS 16

__global__ __launch_bounds__(256) void map_crop_u8_kernel(StriveMap map, const float* __restrict__ pos,
                                                            Float4Host pmean, Float4Host pstd,
                                                            const int32_t* __restrict__ mapix, uint8_t* __restrict__ out) {
    const int n = blockIdx.y;
    const int l0 = blockIdx.x * CROP_ROWS;
    CropFrame fr = load_crop_frame(map, pos, pmean.v, pstd.v, mapix, n);
    const size_t plane = (size_t)map.H * map.W;
    for (int w = threadIdx.x; w < map.Wc; w += blockDim.x) {
        const float ww = map.wwise[w];
        for (int l = l0; l < l0 + CROP_ROWS && l < map.L; ++l) {
            int px, py;
            crop_pixel(fr, map.lwise[l], ww, true, px, py);
            const uint8_t* src = fr.base + (size_t)py * map.W + px;
            for (int c = 0; c < map.C; ++c)
                out[(((size_t)n * map.C + c) * map.L + l) * map.Wc + w] = src[c * plane];
        }
    }
}

// The same crop from the pixel-interleaved raster (one 32-bit word = the 4 layers of a pixel): a thread takes 4 consecutive
// columns of a row -- 4 word loads instead of 16 byte loads -- and writes one 32-bit word per layer instead of 16 bytes one by
// one (the byte-wise kernel above moves 67 MB for 256 crops at 0.24 TB/s).  grid = (L / 4, N), 256 threads = 4 rows x 64.
__global__ __launch_bounds__(256) void map_crop_u8_px4_kernel(StriveMap map, const float* __restrict__ pos, Float4Host pmean,
                                                                Float4Host pstd, const int32_t* __restrict__ mapix,
                                                                uint8_t* __restrict__ out) {
    const int n = blockIdx.y;
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), w0 = (threadIdx.x & 63) * 4;
    if (l >= map.L) return;
    CropFrame fr = load_crop_frame(map, pos, pmean.v, pstd.v, mapix, n);
    const uint32_t* px4 = reinterpret_cast<const uint32_t*>(map.raster_px4) + (size_t)mapix[n] * map.H * map.W;
    const float ll = map.lwise[l];
    for (int w = w0; w < map.Wc; w += 256) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int px, py;
            crop_pixel(fr, ll, map.wwise[w + k], true, px, py);
            v[k] = px4[(size_t)py * map.W + px];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t word = ((v[0] >> (8 * c)) & 255u) | (((v[1] >> (8 * c)) & 255u) << 8) | (((v[2] >> (8 * c)) & 255u) << 16) |
                                  (((v[3] >> (8 * c)) & 255u) << 24);
            *reinterpret_cast<uint32_t*>(out + (((size_t)n * 4 + c) * map.L + l) * map.Wc + w) = word;
        }
    }
}

extern "C" int strive_map_crop_u8(const StriveMap* map, const float* pos, const float* pos_mean4_host,
                                  const float* pos_std4_host, const int32_t* mapix, int32_t N, uint8_t* out,
                                  strive_stream_t stream) {
    STRIVE_CHECK_ARG(map && pos && mapix && out && pos_mean4_host && pos_std4_host, "null argument");
    STRIVE_CHECK_ARG(N >= 0 && map->L > 0 && map->Wc > 0 && map->C > 0, "bad sizes");
    if (N == 0) return 0;
    Float4Host m, s;
    memcpy(m.v, pos_mean4_host, 16);
    memcpy(s.v, pos_std4_host, 16);
    if (map->C == 4 && map->raster_px4 && map->Wc % 4 == 0) {
        hipLaunchKernelGGL(map_crop_u8_px4_kernel, dim3((map->L + 3) / 4, N), dim3(256), 0, (hipStream_t)stream, *map, pos, m, s, mapix,
                           out);
    } else {
        dim3 grid((map->L + CROP_ROWS - 1) / CROP_ROWS, N);
        hipLaunchKernelGGL(map_crop_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, *map, pos, m, s, mapix, out);
    }
    STRIVE_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// collision point: one workgroup per car, threads over the gl x gw samples inside the car box.
// Sums are taken in float64 (the reference's fp32 torch.sum differs from any other order by
// ~1e-4 m at map coordinates of 1e3 m; float64 is the faithful side of that noise).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coll_point_kernel(StriveMap map, const float* __restrict__ cars,
                                                           const float* __restrict__ lw, const int32_t* __restrict__ mapix,
                                                           int gl, int gw, const float* __restrict__ lin_l,
                                                           const float* __restrict__ lin_w, float* __restrict__ out_pt,
                                                           int32_t* __restrict__ out_cnt, int TO,
                                                           const int32_t* __restrict__ agent_of) {
    __shared__ double s_x[4], s_y[4];
    __shared__ int s_n[4];
    const int n = blockIdx.x;
    // agent_of: rows are (e, t) of an up-sampled trajectory tensor; the car sits at fine[agent_of[e]*TO + t] and the
    // attributes are per e (strive_coll_point_rows)
    int crow = n, arow = n;
    if (agent_of) {
        arow = n / TO;
        crow = agent_of[arow] * TO + (n - arow * TO);
    }
    CropFrame fr;
    fr.x = cars[(size_t)crow * 4 + 0];
    fr.y = cars[(size_t)crow * 4 + 1];
    fr.hc = cars[(size_t)crow * 4 + 2];
    fr.hs = cars[(size_t)crow * 4 + 3];
    const int m = mapix[arow];
    set_crop_scale(fr, map.dx[m * 2 + 0], map.dx[m * 2 + 1]);
    fr.H = map.H;
    fr.W = map.W;
    fr.base = map.raster + (size_t)m * map.C * map.H * map.W;   // layer 0
    const float ls = lw[arow * 2 + 0], ws = lw[arow * 2 + 1];
    double sx = 0.0, sy = 0.0;
    int cnt = 0;
    for (int i = threadIdx.x; i < gl * gw; i += blockDim.x) {
        const int a = i / gw, b = i - a * gw;
        // (linspace * ls) / 2   (reference nuscenes_utils.py:221-222)
        const float lwise = __fmul_rn(lin_l[a], ls) * 0.5f;
        const float wwise = __fmul_rn(lin_w[b], ws) * 0.5f;
        int px, py;
        float gx, gy;
        crop_world(fr, lwise, wwise, gx, gy);
        world_to_pixel(fr, gx, gy, px, py);
        if (fr.base[(size_t)py * map.W + px] == 0) {
            sx += (double)gx;
            sy += (double)gy;
            cnt++;
        }
    }
    sx = wave_sum_d(sx);
    sy = wave_sum_d(sy);
    for (int msk = 32; msk >= 1; msk >>= 1) cnt += __shfl_xor(cnt, msk);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_x[wv] = sx; s_y[wv] = sy; s_n[wv] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tx = 0, ty = 0;
        int tn = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { tx += s_x[w]; ty += s_y[w]; tn += s_n[w]; }
        out_cnt[n] = tn;
        if (tn == 0 || tn == gl * gw) {
            out_pt[n * 2 + 0] = __int_as_float(0x7fc00000);
            out_pt[n * 2 + 1] = __int_as_float(0x7fc00000);
        } else {
            out_pt[n * 2 + 0] = (float)(tx / (double)tn);
            out_pt[n * 2 + 1] = (float)(ty / (double)tn);
        }
    }
}

extern "C" int strive_coll_point(const StriveMap* map, const float* cars, const float* lw, const int32_t* mapix,
                                 int32_t N, int32_t gl, int32_t gw, const float* lin_l, const float* lin_w,
                                 float* out_pt, int32_t* out_cnt, strive_stream_t stream) {
    STRIVE_CHECK_ARG(map && cars && lw && mapix && lin_l && lin_w && out_pt && out_cnt, "null argument");
    STRIVE_CHECK_ARG(N >= 0 && gl > 0 && gw > 0, "bad sizes");
    if (N == 0) return 0;
    hipLaunchKernelGGL(coll_point_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, *map, cars, lw, mapix, gl, gw,
                       lin_l, lin_w, out_pt, out_cnt, 1, (const int32_t*)nullptr);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_coll_point_rows(const StriveMap* map, const float* fine, int32_t TO, const int32_t* agent_of,
                                      const float* lw, const int32_t* mapix, int32_t NE, int32_t gl, int32_t gw,
                                      const float* lin_l, const float* lin_w, float* out_pt, int32_t* out_cnt,
                                      strive_stream_t stream) {
    STRIVE_CHECK_ARG(map && fine && agent_of && lw && mapix && lin_l && lin_w && out_pt && out_cnt, "null argument");
    STRIVE_CHECK_ARG(NE >= 0 && TO > 0 && gl > 0 && gw > 0, "bad sizes");
    if (NE == 0) return 0;
    hipLaunchKernelGGL(coll_point_kernel, dim3(NE * TO), dim3(256), 0, (hipStream_t)stream, *map, fine, lw, mapix, gl, gw,
                       lin_l, lin_w, out_pt, out_cnt, TO, agent_of);
    STRIVE_CHECK_LAUNCH();
    return 0;
}
