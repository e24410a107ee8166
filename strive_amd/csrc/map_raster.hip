// Rasterisation of map geometry into the uint8 layers the map crop reads.
// (reference src/datasets/map_env.py:79-166: NuScenesMapEnv.__init__ binarises every layer of every map once and keeps
//  nusc_raster (M, C, H, W) / nusc_dx (M, 2) on the device; the polygon and line geometry comes from the nuscenes devkit there
//  -- NuScenesMap.get_map_mask -> cv2.fillPoly / cv2.polylines -- which is not available here, see include/strive_hip.h)
//
// Rule (this library's, stated because no fixture of the devkit's rasteriser can exist here): pixel (row r, column c) of a
// layer is the value of the layer AT THE POINT THE CROP LOOKS IT UP -- get_map_obs reads raster[round(y / dx_y), round(x / dx_x)]
// (src/datasets/nuscenes_utils.py:250-263), so the pixel stands for the world point (c dx_x, r dx_y).  It is 1 when that point
// lies inside or on the boundary of a polygon of the layer (even-odd over the polygon's rings: holes), or within half a line
// width of a polyline of the layer (boundary included).  float64 throughout, no fused multiply-adds: on geometry whose
// coordinates are multiples of 2^-k the tests are exact, which is what the rational-arithmetic oracle (oracle/raster.py) pins.
//
// Work decomposition: a workgroup per 32 x 32 pixel tile; the host has listed, per tile, the shapes whose bounding boxes touch
// it (CSR); the tile's threads (one per 4 pixels of a row) walk that list, the edges of a shape staged through LDS in chunks.
#include "common.h"

#pragma clang fp contract(off)

namespace {
constexpr int RT = 32;            // tile edge in pixels
constexpr int ECH = 256;          // edges staged per chunk

struct Edge { double x0, y0, x1, y1; };

__global__ __launch_bounds__(256) void raster_tile_kernel(StriveRasterJob job, uint8_t* __restrict__ out) {
    __shared__ Edge s_e[ECH];
    const int tx = blockIdx.x, ty = blockIdx.y, tid = threadIdx.x;
    const int tiles_x = (job.W + RT - 1) / RT;
    const int tile = ty * tiles_x + tx;
    const int row = ty * RT + (tid >> 3), col0 = tx * RT + (tid & 7) * 4;
    const double py = (double)row * job.dx_y;
    double px[4];
    for (int k = 0; k < 4; ++k) px[k] = (double)(col0 + k) * job.dx_x;
    unsigned set = 0;                                   // bit k: pixel k of this thread is 1
    const double hw = job.half_width, hw2 = hw * hw;
    for (int li = job.tile_ptr[tile]; li < job.tile_ptr[tile + 1]; ++li) {
        const int s = job.tile_shapes[li];
        const int v0 = job.shape_ptr[s], v1 = job.shape_ptr[s + 1];     // rings of the shape: ring_ptr[v0 .. v1]
        const bool is_line = job.shape_kind[s] != 0;
        unsigned cross = 0, on = 0;
        for (int r = v0; r < v1; ++r) {
            const int a = job.ring_ptr[r], b = job.ring_ptr[r + 1], nv = b - a;
            const int ne = is_line ? nv - 1 : nv;                           // polygons close their rings
            for (int e0 = 0; e0 < ne; e0 += ECH) {
                __syncthreads();
                const int cnt = (ne - e0) < ECH ? (ne - e0) : ECH;
                for (int i = tid; i < cnt; i += 256) {
                    const int i0 = a + e0 + i, i1 = (e0 + i + 1 < nv) ? i0 + 1 : a;
                    s_e[i].x0 = job.verts[2 * i0]; s_e[i].y0 = job.verts[2 * i0 + 1];
                    s_e[i].x1 = job.verts[2 * i1]; s_e[i].y1 = job.verts[2 * i1 + 1];
                }
                __syncthreads();
                for (int i = 0; i < cnt; ++i) {
                    const Edge e = s_e[i];
                    const double ex = e.x1 - e.x0, ey = e.y1 - e.y0;
                    for (int k = 0; k < 4; ++k) {
                        const double qx = px[k] - e.x0, qy = py - e.y0;
                        if (is_line) {
                            // squared distance to the segment <= half_width^2 (projection clamped to the segment)
                            const double len2 = ex * ex + ey * ey;
                            double t = len2 > 0.0 ? (qx * ex + qy * ey) / len2 : 0.0;
                            t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
                            const double dx = qx - t * ex, dy = qy - t * ey;
                            if (dx * dx + dy * dy <= hw2) on |= 1u << k;
                        } else {
                            // on the edge: collinear and inside its bounding box; else the half-open crossing rule
                            const double cr = ex * qy - ey * qx;
                            const double lox = e.x0 < e.x1 ? e.x0 : e.x1, hix = e.x0 < e.x1 ? e.x1 : e.x0;
                            const double loy = e.y0 < e.y1 ? e.y0 : e.y1, hiy = e.y0 < e.y1 ? e.y1 : e.y0;
                            if (cr == 0.0 && px[k] >= lox && px[k] <= hix && py >= loy && py <= hiy) on |= 1u << k;
                            const bool up = e.y0 <= py && py < e.y1, down = e.y1 <= py && py < e.y0;
                            if ((up && cr > 0.0) || (down && cr < 0.0)) cross ^= 1u << k;
                        }
                    }
                }
            }
        }
        set |= on | (is_line ? 0u : cross);
    }
    if (row < job.H) {
        for (int k = 0; k < 4; ++k) {
            if (col0 + k < job.W && ((set >> k) & 1u)) {
                const int orow = job.flip_rows ? (job.H - 1 - row) : row;
                out[(size_t)orow * job.out_pitch + col0 + k] = 1;          // OR into the layer (several jobs may share one: road layers)
            }
        }
    }
}
}  // namespace

extern "C" int strive_map_rasterize(const StriveRasterJob* job, uint8_t* out_layer, strive_stream_t stream) {
    STRIVE_CHECK_ARG(job && out_layer, "null argument");
    STRIVE_CHECK_ARG(job->H > 0 && job->W > 0 && job->out_pitch >= job->W && job->dx_x > 0.0 && job->dx_y > 0.0, "bad raster size");
    STRIVE_CHECK_ARG(job->tile_ptr && job->shape_ptr && job->ring_ptr && job->verts && job->shape_kind, "missing geometry tables");
    const int tiles_x = (job->W + RT - 1) / RT, tiles_y = (job->H + RT - 1) / RT;
    hipLaunchKernelGGL(raster_tile_kernel, dim3(tiles_x, tiles_y), dim3(256), 0, (hipStream_t)stream, *job, out_layer);
    STRIVE_CHECK_LAUNCH();
    return 0;
}
