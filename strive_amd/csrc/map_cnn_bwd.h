// Map CNN backward for the TRAINING path (reference src/train_traffic.py:103-131 back-propagates the reconstruction / KL /
// collision losses into map_conv.* and map_feature.*; the latent-optimisation loops never come here: the reference crops at
// pos.detach(), traffic_model.py:694, so dL/dz needs no CNN backward).
//
// Included at the end of map_cnn.hip (it re-uses the forward's workspace carve-up, GroupNorm statistics and layer table).
// One call takes ALL crops of a rollout's reverse sweep (the crop is data: nothing in the sweep waits for these gradients) in
// chunks of BWD_CHUNK samples.  Per chunk the forward is re-run to regenerate the raw convolution outputs y_l and their
// GroupNorm moments (nothing is kept from the rollout's forward sweep: 1.76 MB per agent and step), then per layer, top down:
//     GroupNorm(1)+ReLU backward in place on G_l   (moments from the forward's float64 partial sums; gn_bwd_*_oct_kernel walk
//                                                   the octet-planar activations in their own order)
//     weight gradient   dW_l[co][ci][ky][kx] += sum_{n,oy,ox} dy_l[n][co][oy][ox] * a_{l-1}[n][ci][2 oy + ky][2 ox + kx]
//     data gradient     G_{l-1}[n][ci][iy][ix] = sum_{co,ky,kx} dy_l[n][co][(iy-ky)/2][(ix-kx)/2] * W_l[co][ci][ky][kx]
// Both convolution gradients run on the matrix cores with two-piece bf16 operand splits (map_cnn_bwd_mfma.h).
// The round-2 forms stay in this file as A/B switches: one fp32 64 x 64 x 16 LDS-tiled implicit GEMM whose operand tiles are
// fetched through small functors (STRIVE_DGRAD_IGEMM / STRIVE_WGRAD_IGEMM), and the fp32 weight gradient with operands staged
// once per tile (STRIVE_WGRAD_TILE).
#pragma once

namespace cnnbwd {

constexpr int LC_IN[6] = {4, 16, 32, 64, 64, 128};
constexpr int LC_OUT[6] = {16, 32, 64, 64, 128, 128};
constexpr int LKS[6] = {7, 5, 5, 3, 3, 3};
constexpr int LIH[6] = {256, 125, 61, 29, 14, 6};
constexpr int LOH[6] = {125, 61, 29, 14, 6, 2};
constexpr bool LOCT[6] = {true, true, true, true, true, false};   // layout of the forward's raw output of layer l

struct LayerDesc {
    int cin, cout, ks, ih, oh;
    bool in_oct, out_oct;    // layouts of the forward's activations: layer input (= output of l-1) and output
};

static inline LayerDesc layer_desc(int l) {
    LayerDesc d;
    d.cin = LC_IN[l]; d.cout = LC_OUT[l]; d.ks = LKS[l]; d.ih = LIH[l]; d.oh = LOH[l];
    d.in_oct = l > 0 ? LOCT[l - 1] : false;
    d.out_oct = LOCT[l];
    return d;
}

// element (n, c, y, x) of a forward activation tensor with C channels and H x H pixels
__device__ __forceinline__ size_t act_index(bool oct, int C, int H, int n, int c, int y, int x) {
    if (oct) return ((((size_t)n * (C >> 3) + (c >> 3)) * H + y) * H + x) * 8 + (c & 7);
    return (((size_t)n * C + c) * H + y) * H + x;
}

// ---- per-sample GroupNorm moments of all six layers in one launch: MR[l][n] = (mean, rstd); grid = (ceil(N / 64), 6) ----
struct MomentsArgs {
    const GNStats* st[6];
    int nparts[6];
    double count[6];
};
static __global__ void moments_kernel(MomentsArgs a, float2* __restrict__ mr, int ch, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.y;
    if (n >= N) return;
    float mean, rstd;
    gn_moments(a.st[l], n, a.nparts[l], a.count[l], mean, rstd);
    mr[(size_t)l * ch + n] = make_float2(mean, rstd);
}

// ---- Linear(512, 64) + GroupNorm6/ReLU input: G5 <- d a6 = W^T d feat;  dW += d feat^T a6;  db += sum d feat ----
// grid = (ceil(N / FCB_S), 512 / FCB_K): a workgroup takes FCB_S samples x FCB_K of the 512 input columns -- every weight-gradient
// atomic carries 32 samples, the weight slice sits in LDS.  (Round 5: the first form gave a workgroup 8 samples and ALL columns:
// 32 workgroups per 256-sample chunk with 128 serial atomics per thread, 75 us per chunk.)
constexpr int FCB_S = 32, FCB_K = 64;
static __global__ __launch_bounds__(256) void fc_bwd_kernel(const float* __restrict__ y6, const float2* __restrict__ mr,
                                                              const float* __restrict__ gn_g, const float* __restrict__ gn_b,
                                                              const float* __restrict__ wt /* (512,64) */, const float* __restrict__ d_feat,
                                                              float* __restrict__ G5, float* __restrict__ dW /* (64,512) */,
                                                              float* __restrict__ db, int N) {
    __shared__ float s_a[FCB_S][FCB_K + 1];
    __shared__ float s_d[FCB_S][64 + 1];
    __shared__ float s_w[FCB_K][64 + 1];
    const int n0 = blockIdx.x * FCB_S, k0 = blockIdx.y * FCB_K, tid = threadIdx.x;
    const int ns = (N - n0) < FCB_S ? (N - n0) : FCB_S;
    for (int i = tid; i < FCB_S * FCB_K; i += 256) {
        const int s = i / FCB_K, kk = i - s * FCB_K, k = k0 + kk;
        float v = 0.f;
        if (s < ns) {
            const int c = k >> 2;
            const float2 m = mr[n0 + s];
            const float xh = (y6[(size_t)(n0 + s) * 512 + k] - m.x) * m.y;
            v = fmaxf(xh * gn_g[c] + gn_b[c], 0.f);
        }
        s_a[s][kk] = v;
    }
    for (int i = tid; i < FCB_S * 64; i += 256) {
        const int s = i >> 6, o = i & 63;
        s_d[s][o] = s < ns ? d_feat[(size_t)(n0 + s) * 64 + o] : 0.f;
    }
    for (int i = tid; i < FCB_K * 64; i += 256) {
        const int kk = i >> 6, o = i & 63;
        s_w[kk][o] = wt[(size_t)(k0 + kk) * 64 + o];
    }
    __syncthreads();
    // data gradient (w.r.t. the post-ReLU activation a6; the ReLU / GroupNorm part follows in the layer-5 GN backward)
    for (int i = tid; i < FCB_S * FCB_K; i += 256) {
        const int s = i / FCB_K, kk = i - s * FCB_K;
        if (s < ns) {
            float acc = 0.f;
#pragma unroll 8
            for (int o = 0; o < 64; ++o) acc = fmaf(s_d[s][o], s_w[kk][o], acc);
            G5[(size_t)(n0 + s) * 512 + k0 + kk] = acc;
        }
    }
    // weight gradient: lanes over k (consecutive addresses of one output row)
    for (int i = tid; i < 64 * FCB_K; i += 256) {
        const int o = i / FCB_K, kk = i - o * FCB_K;
        float acc = 0.f;
        for (int s = 0; s < ns; ++s) acc = fmaf(s_d[s][o], s_a[s][kk], acc);
        if (acc != 0.f) unsafeAtomicAdd(&dW[(size_t)o * 512 + k0 + kk], acc);
    }
    if (blockIdx.y == 0 && tid < 64) {
        float acc = 0.f;
        for (int s = 0; s < ns; ++s) acc += s_d[s][tid];
        if (acc != 0.f) unsafeAtomicAdd(&db[tid], acc);
    }
}

// ---- GroupNorm(1) + ReLU backward, pass 1: G <- dn = da * [pre > 0];  S[n] += (sum dn*gamma, sum dn*gamma*xhat);
//      dgamma[c] += sum dn*xhat, dbeta[c] += sum dn.   grid = (blocks per sample, N); G is NCHW.
static __global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(LayerDesc d, const float* __restrict__ y, const float2* __restrict__ mr,
                                                                     const float* __restrict__ gam, const float* __restrict__ bet,
                                                                     float* __restrict__ G, double* __restrict__ S,
                                                                     float* __restrict__ dgam, float* __restrict__ dbet) {
    __shared__ float s_dg[128], s_db[128];
    __shared__ double s_s[2][4];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int HW = d.oh * d.oh, M = d.cout * HW;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per, e1 = (e0 + per) < M ? (e0 + per) : M;
    if (tid < 128) { s_dg[tid] = 0.f; s_db[tid] = 0.f; }
    __syncthreads();
    const float2 m = mr[n];
    double s1 = 0.0, s2 = 0.0;
    for (int e = e0 + tid; e < e1; e += 256) {
        const int c = e / HW, p = e - c * HW;
        const int yy = p / d.oh, xx = p - yy * d.oh;
        const float xh = (y[act_index(d.out_oct, d.cout, d.oh, n, c, yy, xx)] - m.x) * m.y;
        const float pre = xh * gam[c] + bet[c];
        const size_t gi = (size_t)n * M + e;
        const float dn = pre > 0.f ? G[gi] : 0.f;
        G[gi] = dn;
        if (dn != 0.f) {
            const float dg = dn * gam[c];
            s1 += (double)dg;
            s2 += (double)dg * (double)xh;
            atomicAdd(&s_dg[c], dn * xh);
            atomicAdd(&s_db[c], dn);
        }
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((tid & 63) == 0) { s_s[0][tid >> 6] = s1; s_s[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        unsafeAtomicAdd(&S[2 * n + 0], (s_s[0][0] + s_s[0][1]) + (s_s[0][2] + s_s[0][3]));
        unsafeAtomicAdd(&S[2 * n + 1], (s_s[1][0] + s_s[1][1]) + (s_s[1][2] + s_s[1][3]));
    }
    if (tid < d.cout) {
        if (s_dg[tid] != 0.f) unsafeAtomicAdd(&dgam[tid], s_dg[tid]);
        if (s_db[tid] != 0.f) unsafeAtomicAdd(&dbet[tid], s_db[tid]);
    }
}

// ---- pass 2: G <- dy = rstd * (dn*gamma - S1/M - xhat*S2/M);  conv bias gradient db[c] += sum dy ----
static __global__ __launch_bounds__(256) void gn_bwd_apply_kernel(LayerDesc d, const float* __restrict__ y, const float2* __restrict__ mr,
                                                                    const float* __restrict__ gam, float* __restrict__ G,
                                                                    const double* __restrict__ S, float* __restrict__ dbias) {
    __shared__ float s_db[128];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int HW = d.oh * d.oh, M = d.cout * HW;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per, e1 = (e0 + per) < M ? (e0 + per) : M;
    if (tid < 128) s_db[tid] = 0.f;
    __syncthreads();
    const float2 m = mr[n];
    const float a1 = (float)(S[2 * n + 0] / (double)M), a2 = (float)(S[2 * n + 1] / (double)M);
    for (int e = e0 + tid; e < e1; e += 256) {
        const int c = e / HW, p = e - c * HW;
        const int yy = p / d.oh, xx = p - yy * d.oh;
        const float xh = (y[act_index(d.out_oct, d.cout, d.oh, n, c, yy, xx)] - m.x) * m.y;
        const size_t gi = (size_t)n * M + e;
        const float dy = m.y * (G[gi] * gam[c] - a1 - xh * a2);
        G[gi] = dy;
        atomicAdd(&s_db[c], dy);
    }
    __syncthreads();
    if (tid < d.cout && s_db[tid] != 0.f) unsafeAtomicAdd(&dbias[tid], s_db[tid]);
}

// ---- the same two passes for the octet-planar layers (l = 0..4), walked in the activation's own order: a thread takes the 8
// channels of one pixel (32 contiguous bytes of y -- the NCHW walk above uses 4 bytes of every 32-byte sector it fetches) and
// the 8 matching G values (one coalesced row per channel).  Its channels are fixed, so the per-channel sums stay in registers
// and meet in LDS once per wave.  grid = (octets x pixel blocks, N), GN_PPB pixels per block.
constexpr int GN_PPB = 2048;

static __global__ __launch_bounds__(256) void gn_bwd_reduce_oct_kernel(int C, int HW, const float* __restrict__ y,
                                                                        const float2* __restrict__ mr, const float* __restrict__ gam,
                                                                        const float* __restrict__ bet, const float* __restrict__ G,
                                                                        double* __restrict__ S, float* __restrict__ dgam,
                                                                        float* __restrict__ dbet) {
    __shared__ float s_c[16];
    __shared__ double s_s[2];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int npb = (HW + GN_PPB - 1) / GN_PPB;
    const int oct = blockIdx.x / npb, pb = blockIdx.x - oct * npb;
    const int p1 = (pb + 1) * GN_PPB < HW ? (pb + 1) * GN_PPB : HW;
    if (tid < 16) s_c[tid] = 0.f;
    if (tid < 2) s_s[tid] = 0.0;
    __syncthreads();
    const float2 m = mr[n];
    float ga[8], be[8], dg[8], db[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ga[c] = gam[oct * 8 + c]; be[c] = bet[oct * 8 + c]; dg[c] = 0.f; db[c] = 0.f; }
    double s1 = 0.0, s2 = 0.0;
    const float* yb = y + ((size_t)n * (C >> 3) + oct) * HW * 8;
    const float* gb = G + ((size_t)n * C + oct * 8) * HW;
    for (int p = pb * GN_PPB + tid; p < p1; p += 256) {
        const float4 y0 = *reinterpret_cast<const float4*>(yb + (size_t)p * 8), y1 = *reinterpret_cast<const float4*>(yb + (size_t)p * 8 + 4);
        const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
        float f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xh = (yy[c] - m.x) * m.y;
            const float pre = xh * ga[c] + be[c];
            const float dn = pre > 0.f ? gb[(size_t)c * HW + p] : 0.f;      // (read only: the apply pass masks again, see there)
            const float w = dn * ga[c];
            f1 += w;
            f2 = fmaf(w, xh, f2);
            dg[c] = fmaf(dn, xh, dg[c]);
            db[c] += dn;
        }
        s1 += (double)f1;
        s2 += (double)f2;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
#pragma unroll
    for (int c = 0; c < 8; ++c) { dg[c] = wave_sum(dg[c]); db[c] = wave_sum(db[c]); }
    if ((tid & 63) == 0) {
        unsafeAtomicAdd(&s_s[0], s1);
        unsafeAtomicAdd(&s_s[1], s2);
#pragma unroll
        for (int c = 0; c < 8; ++c) { atomicAdd(&s_c[c], dg[c]); atomicAdd(&s_c[8 + c], db[c]); }
    }
    __syncthreads();
    if (tid == 0) {
        unsafeAtomicAdd(&S[2 * n + 0], s_s[0]);
        unsafeAtomicAdd(&S[2 * n + 1], s_s[1]);
    }
    if (tid < 8) { if (s_c[tid] != 0.f) unsafeAtomicAdd(&dgam[oct * 8 + tid], s_c[tid]); }
    else if (tid < 16) { if (s_c[tid] != 0.f) unsafeAtomicAdd(&dbet[oct * 8 + tid - 8], s_c[tid]); }
}

// (Round 5: the reduce pass used to store the ReLU-masked gradient back into G -- a scattered partial-line write of about half the
//  elements; this pass reads y anyway and masks again with the same comparison, so the reduce pass is read-only now.)
static __global__ __launch_bounds__(256) void gn_bwd_apply_oct_kernel(int C, int HW, const float* __restrict__ y,
                                                                       const float2* __restrict__ mr, const float* __restrict__ gam,
                                                                       const float* __restrict__ bet, float* __restrict__ G,
                                                                       const double* __restrict__ S, float* __restrict__ dbias) {
    __shared__ float s_c[8];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int npb = (HW + GN_PPB - 1) / GN_PPB;
    const int oct = blockIdx.x / npb, pb = blockIdx.x - oct * npb;
    const int p1 = (pb + 1) * GN_PPB < HW ? (pb + 1) * GN_PPB : HW;
    if (tid < 8) s_c[tid] = 0.f;
    __syncthreads();
    const float2 m = mr[n];
    const double M = (double)C * HW;
    const float a1 = (float)(S[2 * n + 0] / M), a2 = (float)(S[2 * n + 1] / M);
    float ga[8], be[8], db[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ga[c] = gam[oct * 8 + c]; be[c] = bet[oct * 8 + c]; db[c] = 0.f; }
    const float* yb = y + ((size_t)n * (C >> 3) + oct) * HW * 8;
    float* gb = G + ((size_t)n * C + oct * 8) * HW;
    for (int p = pb * GN_PPB + tid; p < p1; p += 256) {
        const float4 y0 = *reinterpret_cast<const float4*>(yb + (size_t)p * 8), y1 = *reinterpret_cast<const float4*>(yb + (size_t)p * 8 + 4);
        const float yy[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xh = (yy[c] - m.x) * m.y;
            float* gp = gb + (size_t)c * HW + p;
            const float pre = xh * ga[c] + be[c];                  // (the reduce pass's expression: the same mask)
            const float dn = pre > 0.f ? *gp : 0.f;
            const float dyv = m.y * (dn * ga[c] - a1 - xh * a2);
            *gp = dyv;
            db[c] += dyv;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) db[c] = wave_sum(db[c]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) atomicAdd(&s_c[c], db[c]);
    }
    __syncthreads();
    if (tid < 8 && s_c[tid] != 0.f) unsafeAtomicAdd(&dbias[oct * 8 + tid], s_c[tid]);
}

// ---------------------------------------------------------------------------------------------
// fp32 implicit GEMM:  C[m][j] (+)= sum_k A(m, k) * B(k, j),  64 x 64 tile per workgroup, K in steps of 16, 256 threads,
// 4 x 4 outputs per thread (rows ty + 16 i, columns tx + 16 j).  P supplies M, N, the K range of this workgroup and the
// element functors; out-of-range elements read as 0.
// ---------------------------------------------------------------------------------------------
template <class P>
static __global__ __launch_bounds__(256) void igemm64_kernel(P p) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    typename P::Ctx cx = p.context(blockIdx.z);
    if (j0 >= cx.N) return;          // block-uniform
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = cx.k_begin; k0 < cx.k_end; k0 += 16) {
        // A tile: lanes over k (A is contiguous along k for the weight gradient)
        for (int e = tid; e < 64 * 16; e += 256) {
            const int kk = e & 15, mm = e >> 4;
            const int m = m0 + mm, k = k0 + kk;
            As[kk][mm] = (m < p.M && k < cx.k_end) ? p.loadA(cx, m, k) : 0.f;
        }
        // B tile: lanes over j
        for (int e = tid; e < 64 * 16; e += 256) {
            const int jj = e & 63, kk = e >> 6;
            const int j = j0 + jj, k = k0 + kk;
            Bs[kk][jj] = (j < cx.N && k < cx.k_end) ? p.loadB(cx, k, j) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jg = j0 + tx + 16 * j;
            if (jg < cx.N) p.store(cx, m, jg, acc[i][j]);
        }
    }
}

// ---- weight gradient of layer l:  M = cout, N = cin*ks*ks, K = samples * oh*oh, split over grid.z ----
struct WgradProb {
    LayerDesc d;
    int M, NS, kchunk;
    const float* dy;            // G_l, NCHW (NS, cout, oh, oh)
    const float* act_in;        // forward raw output of layer l-1 (l > 0) ...
    const uint8_t* crop;        // ... or the uint8 crop (NS, 4, 256, 256) for l = 0
    const float2* mr_in;        // moments of layer l-1
    const float* gam_in;
    const float* bet_in;
    float* dW;                  // (cout, cin, ks, ks), accumulated
    struct Ctx { int k_begin, k_end, N; };
    __device__ Ctx context(int z) const {
        Ctx c;
        const int K = NS * d.oh * d.oh;
        c.k_begin = z * kchunk;
        c.k_end = (c.k_begin + kchunk) < K ? (c.k_begin + kchunk) : K;
        c.N = d.cin * d.ks * d.ks;
        return c;
    }
    __device__ float loadA(const Ctx&, int m, int k) const {
        const int HW = d.oh * d.oh;
        const int n = k / HW, rem = k - n * HW;
        return dy[((size_t)n * d.cout + m) * HW + rem];
    }
    __device__ float loadB(const Ctx&, int k, int j) const {
        const int HW = d.oh * d.oh, KK = d.ks * d.ks;
        const int n = k / HW, rem = k - n * HW;
        const int oy = rem / d.oh, ox = rem - oy * d.oh;
        const int ci = j / KK, t = j - ci * KK;
        const int ky = t / d.ks, kx = t - ky * d.ks;
        const int iy = 2 * oy + ky, ix = 2 * ox + kx;
        if (crop) return (float)crop[(((size_t)n * 4 + ci) * 256 + iy) * 256 + ix];
        const float2 m = mr_in[n];
        const float v = act_in[act_index(d.in_oct, d.cin, d.ih, n, ci, iy, ix)];
        return fmaxf((v - m.x) * m.y * gam_in[ci] + bet_in[ci], 0.f);
    }
    __device__ void store(const Ctx& c, int m, int j, float v) const {
        if (v != 0.f) unsafeAtomicAdd(&dW[(size_t)m * c.N + j], v);
    }
};

// ---- data gradient of layer l (l > 0): per sample and input-pixel parity class (grid.z = 4 n + class) ----
//      M = cin, N = pixels of the class, K = cout * taps of the class
struct DgradProb {
    LayerDesc d;
    int M;
    const float* dy;            // G_l, NCHW
    const float* w;             // torch layout (cout, cin, ks, ks)
    float* gin;                 // G_{l-1}, NCHW (NS, cin, ih, ih): every element is written exactly once
    struct Ctx { int k_begin, k_end, N, n, py, px, ny, nx, na, nb; };
    __device__ Ctx context(int z) const {
        Ctx c;
        c.n = z >> 2;
        c.py = (z >> 1) & 1;
        c.px = z & 1;
        c.ny = (d.ih - c.py + 1) / 2;
        c.nx = (d.ih - c.px + 1) / 2;
        c.na = (d.ks - c.py + 1) / 2;
        c.nb = (d.ks - c.px + 1) / 2;
        c.N = c.ny * c.nx;
        c.k_begin = 0;
        c.k_end = d.cout * c.na * c.nb;
        return c;
    }
    __device__ float loadA(const Ctx& c, int m, int k) const {
        const int T = c.na * c.nb;
        const int co = k / T, t = k - co * T;
        const int a = t / c.nb, b = t - a * c.nb;
        return w[(((size_t)co * d.cin + m) * d.ks + (c.py + 2 * a)) * d.ks + (c.px + 2 * b)];
    }
    __device__ float loadB(const Ctx& c, int k, int j) const {
        const int T = c.na * c.nb;
        const int co = k / T, t = k - co * T;
        const int a = t / c.nb, b = t - a * c.nb;
        const int iy2 = j / c.nx, ix2 = j - iy2 * c.nx;
        const int oy = iy2 - a, ox = ix2 - b;
        if (oy < 0 || oy >= d.oh || ox < 0 || ox >= d.oh) return 0.f;
        return dy[(((size_t)c.n * d.cout + co) * d.oh + oy) * d.oh + ox];
    }
    __device__ void store(const Ctx& c, int m, int j, float v) const {
        const int iy2 = j / c.nx, ix2 = j - iy2 * c.nx;
        gin[(((size_t)c.n * d.cin + m) * d.ih + (2 * iy2 + c.py)) * d.ih + (2 * ix2 + c.px)] = v;
    }
};

// ---------------------------------------------------------------------------------------------
// Weight gradient with the operands staged ONCE per tile (round 3).  The implicit-GEMM form above fetches every operand
// element through a functor: run-time index divisions, a GroupNorm + ReLU evaluation per USE of an input value (25 uses for a
// 5 x 5 window) and 4-byte gathers 32-64 bytes apart -- profiles/r03_train_kernel_stats.txt: 48 % of the training step.
// Here a workgroup owns one block of 8 input channels (the octet of the forward's activation layout: 32 contiguous bytes per
// pixel; layer 0: the 4 raster layers) and walks (sample, band of output rows) units: the band of dy (all output channels)
// and the matching band of the input -- normalised and rectified once per element on the way in -- go to LDS with coalesced
// row loads, then thread (ci, ky, group of output channels) slides along the band with its KS window values in registers (two new
// ones per output pixel): COG x KS multiply-adds per 2 + COG LDS reads.  All layer dimensions are template constants.
// The partial sums of all units of the workgroup stay in registers and are added to dW once, with atomics.
// ---------------------------------------------------------------------------------------------
template <int L>
struct WgTile {
    static constexpr int CIN = LC_IN[L], COUT = LC_OUT[L], KS = LKS[L], IH = LIH[L], OH = LOH[L];
    static constexpr int CB = CIN < 8 ? CIN : 8, NCB = CIN / CB;
    static constexpr int RH = L == 0 ? 2 : L == 1 ? 2 : L == 2 ? 4 : L == 3 ? 7 : L == 4 ? 6 : 2;      // output rows per band
    static constexpr int NBAND = (OH + RH - 1) / RH, IR = 2 * RH + KS - 2;
    static constexpr int IWP = IH | 1;                                   // odd row pitch: (ci, ky) rows land in different banks
    static constexpr int OWP = OH;
    static constexpr int PAIRS = CB * KS, NG = 256 / PAIRS, COG = (COUT + NG - 1) / NG;
    static constexpr int A_FLOATS = CB * IR * IWP, DY_FLOATS = COUT * RH * OWP;
    static constexpr size_t LDS_BYTES = (size_t)(A_FLOATS + DY_FLOATS) * 4;
    static_assert(LDS_BYTES <= 64 * 1024 && NG >= 1 && COG * KS <= 64, "tile budget");
};

template <int L>
static __global__ __launch_bounds__(256) void wgrad_tile_kernel(const float* __restrict__ dy, const float* __restrict__ act_in,
                                                                 const uint8_t* __restrict__ crop, const float2* __restrict__ mr_in,
                                                                 const float* __restrict__ gam_in, const float* __restrict__ bet_in,
                                                                 float* __restrict__ dW, int NS) {
    using T = WgTile<L>;
    HIP_DYNAMIC_SHARED(float, smem)
    float* s_a = smem;                       // [CB][IR][IWP]   input band, GroupNorm + ReLU applied
    float* s_dy = smem + T::A_FLOATS;        // [COUT][RH][OWP] output-gradient band
    const int tid = threadIdx.x, cb = blockIdx.y;
    const int grp = tid / T::PAIRS, pair = tid - grp * T::PAIRS;
    const int ci = pair / T::KS, ky = pair - ci * T::KS;
    const bool worker = grp < T::NG;
    const int co0 = grp * T::COG;
    float acc[T::COG][T::KS];
#pragma unroll
    for (int c = 0; c < T::COG; ++c)
#pragma unroll
        for (int k = 0; k < T::KS; ++k) acc[c][k] = 0.f;
    const int units = NS * T::NBAND;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int n = u / T::NBAND, band = u - n * T::NBAND;
        const int oy0 = band * T::RH;
        const int rh = (T::OH - oy0) < T::RH ? (T::OH - oy0) : T::RH;
        const int ir = 2 * rh + T::KS - 2;
        __syncthreads();                     // the previous unit's readers are done
        // ---- dy band: rows oy0 .. oy0+rh of every output channel are contiguous in NCHW ----
        // (the staging loops are unrolled so that several loads are in flight: with one load per trip these kernels spent most
        //  of their time waiting for 30-50 dependent-latency loads per unit)
        {
            const int cnt = T::COUT * rh * T::OH;
#pragma unroll 4
            for (int e = tid; e < cnt; e += 256) {
                const int co = e / (rh * T::OH), rem = e - co * (rh * T::OH);
                s_dy[co * (T::RH * T::OWP) + rem] = dy[(((size_t)n * T::COUT + co) * T::OH + oy0) * T::OH + rem];
            }
        }
        // ---- input band of this channel block ----
        if (L == 0) {
            // four raster pixels per load (rows of the crop are 256-byte aligned)
            const int cnt = T::CB * ir * (T::IH / 4);
#pragma unroll 4
            for (int e = tid; e < cnt; e += 256) {
                const int c = e / (ir * (T::IH / 4)), rem = e - c * (ir * (T::IH / 4));
                const int row = rem / (T::IH / 4), x4 = rem - row * (T::IH / 4);
                const uint32_t v = *reinterpret_cast<const uint32_t*>(crop + (((size_t)n * 4 + c) * 256 + 2 * oy0 + row) * 256 + 4 * x4);
                float* dst = s_a + (c * T::IR + row) * T::IWP + 4 * x4;
                dst[0] = (float)(v & 255u); dst[1] = (float)((v >> 8) & 255u); dst[2] = (float)((v >> 16) & 255u); dst[3] = (float)(v >> 24);
            }
        } else {
            // four channels of a pixel per load (octet-planar rows are 32-byte aligned)
            const float2 m = mr_in[n];
            const float4* src = reinterpret_cast<const float4*>(act_in + (((size_t)n * T::NCB + cb) * T::IH + 2 * oy0) * T::IH * 8);
            const int cnt = ir * T::IH * 2;
#pragma unroll 4
            for (int e = tid; e < cnt; e += 256) {
                const int c4 = (e & 1) * 4, px = e >> 1;
                const int row = px / T::IH, x = px - row * T::IH;
                const float4 v = src[e];
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cg = cb * 8 + c4 + q;
                    s_a[((c4 + q) * T::IR + row) * T::IWP + x] = fmaxf((vv[q] - m.x) * m.y * gam_in[cg] + bet_in[cg], 0.f);
                }
            }
        }
        __syncthreads();
        if (worker) {
            for (int r = 0; r < rh; ++r) {
                const float* arow = s_a + (ci * T::IR + 2 * r + ky) * T::IWP;
                const float* drow = s_dy + (size_t)co0 * (T::RH * T::OWP) + r * T::OWP;
                // the KS window values slide by two per output pixel: two LDS reads per step instead of KS
                float a[T::KS];
#pragma unroll
                for (int k = 2; k < T::KS; ++k) a[k] = arow[k - 2];
                for (int ox = 0; ox < T::OH; ++ox) {
#pragma unroll
                    for (int k = 0; k + 2 < T::KS; ++k) a[k] = a[k + 2];
                    a[T::KS - 2] = arow[2 * ox + T::KS - 2];
                    a[T::KS - 1] = arow[2 * ox + T::KS - 1];
#pragma unroll
                    for (int c = 0; c < T::COG; ++c) {
                        if (co0 + c < T::COUT) {
                            const float d = drow[c * (T::RH * T::OWP) + ox];
#pragma unroll
                            for (int k = 0; k < T::KS; ++k) acc[c][k] = fmaf(d, a[k], acc[c][k]);
                        }
                    }
                }
            }
        }
    }
    if (worker) {
#pragma unroll
        for (int c = 0; c < T::COG; ++c) {
            if (co0 + c < T::COUT) {
                float* w = dW + (((size_t)(co0 + c) * T::CIN + cb * T::CB + ci) * T::KS + ky) * T::KS;
#pragma unroll
                for (int k = 0; k < T::KS; ++k)
                    if (acc[c][k] != 0.f) unsafeAtomicAdd(&w[k], acc[c][k]);
            }
        }
    }
}

template <int L>
static inline void launch_wgrad_tile(const float* dy, const float* act_in, const uint8_t* crop, const float2* mr_in, const float* gam_in,
                                     const float* bet_in, float* dW, int NS, hipStream_t stream) {
    using T = WgTile<L>;
    const int units = NS * T::NBAND;
    int per_cb = 768 / T::NCB;                 // ~3 workgroups per CU in total
    if (per_cb > units) per_cb = units;
    if (per_cb < 1) per_cb = 1;
    static PerDeviceOnce once;
    const int dev_ = once.device();
    if (!once.is_done(dev_)) {
        hipFuncSetAttribute((const void*)wgrad_tile_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
        once.set_done(dev_);
    }
    hipLaunchKernelGGL(wgrad_tile_kernel<L>, dim3(per_cb, T::NCB), dim3(256), T::LDS_BYTES, stream, dy, act_in, crop, mr_in, gam_in,
                       bet_in, dW, NS);
}

static inline void launch_wgrad_tile_layer(int l, const float* dy, const float* act_in, const uint8_t* crop, const float2* mr_in,
                                           const float* gam_in, const float* bet_in, float* dW, int NS, hipStream_t stream) {
    switch (l) {
        case 0: launch_wgrad_tile<0>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, NS, stream); break;
        case 1: launch_wgrad_tile<1>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, NS, stream); break;
        case 2: launch_wgrad_tile<2>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, NS, stream); break;
        case 3: launch_wgrad_tile<3>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, NS, stream); break;
        case 4: launch_wgrad_tile<4>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, NS, stream); break;
        default: launch_wgrad_tile<5>(dy, act_in, crop, mr_in, gam_in, bet_in, dW, NS, stream); break;
    }
}

}  // namespace cnnbwd
#include "map_cnn_bwd_mfma.h"
namespace cnnbwd {

constexpr int BWD_CHUNK = 256;    // samples pushed through forward-recompute + backward together (3.8 MB of workspace each)

static inline size_t grad_floats_per_sample() {
    size_t t = 0;
    for (int l = 0; l < 6; ++l) t += L_OUT[l];
    return t;
}

// flat gradient layout: per layer conv W (co,ci,k,k) | conv b | GN gamma | GN beta ; then fc W (64,512) | fc b
struct CnnGradPtrs {
    float *w[6], *b[6], *g[6], *be[6], *fcw, *fcb;
};
static inline size_t cnn_param_count() {
    size_t n = 0;
    for (int l = 0; l < 6; ++l) n += (size_t)LC_OUT[l] * LC_IN[l] * LKS[l] * LKS[l] + 3 * (size_t)LC_OUT[l];
    return n + 64 * 512 + 64;
}
static inline CnnGradPtrs cnn_grad_ptrs(float* flat) {
    CnnGradPtrs p;
    for (int l = 0; l < 6; ++l) {
        p.w[l] = flat; flat += (size_t)LC_OUT[l] * LC_IN[l] * LKS[l] * LKS[l];
        p.b[l] = flat; flat += LC_OUT[l];
        p.g[l] = flat; flat += LC_OUT[l];
        p.be[l] = flat; flat += LC_OUT[l];
    }
    p.fcw = flat; flat += 64 * 512;
    p.fcb = flat;
    return p;
}

}  // namespace cnnbwd

extern "C" size_t strive_map_cnn_param_count(void) { return cnnbwd::cnn_param_count(); }

extern "C" size_t strive_map_cnn_bwd_workspace_bytes(int32_t N) {
    const size_t ch = (size_t)(N < cnnbwd::BWD_CHUNK ? (N > 0 ? N : 1) : cnnbwd::BWD_CHUNK);
    size_t b = 0;
    b += strive_align_up(strive_map_cnn_workspace_bytes((int32_t)ch), 256);     // forward activations + statistics
    b += strive_align_up(ch * cnnbwd::grad_floats_per_sample() * 4, 256);       // G_0 .. G_5
    b += strive_align_up(ch * 4 * 256 * 256, 256);                              // uint8 crop (conv1's input)
    b += strive_align_up(ch * 6 * sizeof(float2), 256);                         // moments
    b += strive_align_up(ch * 6 * 2 * sizeof(double), 256);                     // GroupNorm backward sums, per layer
    b += strive_align_up(ch * 64 * 4, 256);                                     // feature scratch of the recomputed forward
    b += strive_align_up(cnnbwd::dgrad_frag_total() * 16, 256);                 // bf16 weight fragments of the data gradient
    b += strive_align_up(cnnbwd::wgrad_partial_floats() * 4, 256);              // per-workgroup partial weight gradients
    return b + 1024;
}

namespace cnnbwd {
// the workspace of one backward chunk of `ch` samples (strive_map_cnn_bwd_workspace_bytes lists the same blocks)
struct BwdArena {
    char* fwd_ws;
    size_t fwd_bytes;
    float* G[6];
    uint8_t* crop;
    float2* mr;
    double* S;
    float* feat;
    uint4* dfrag;
    float* wpart;
    bool ok;
};
static inline BwdArena carve_bwd(void* ws, size_t ws_bytes, int ch) {
    BwdArena a;
    StriveArena ar(ws, ws_bytes);
    a.fwd_bytes = strive_map_cnn_workspace_bytes(ch);
    a.fwd_ws = ar.take<char>(a.fwd_bytes);
    for (int l = 0; l < 6; ++l) a.G[l] = ar.take<float>((size_t)ch * L_OUT[l]);
    a.crop = ar.take<uint8_t>((size_t)ch * 4 * 256 * 256);
    a.mr = ar.take<float2>((size_t)ch * 6);
    a.S = ar.take<double>((size_t)ch * 6 * 2);
    a.feat = ar.take<float>((size_t)ch * 64);
    a.dfrag = ar.take<uint4>(dgrad_frag_total());
    a.wpart = ar.take<float>(wgrad_partial_floats());
    a.ok = ar.ok();
    return a;
}
}  // namespace cnnbwd

extern "C" int strive_map_cnn_bwd_bench_dgrad(int32_t layer, int32_t N, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    using namespace cnnbwd;
    STRIVE_CHECK_ARG(ws && layer >= 1 && layer <= 5 && N > 0 && N <= BWD_CHUNK, "bad layer / N");
    STRIVE_CHECK_ARG(ws_bytes >= strive_map_cnn_bwd_workspace_bytes(N), "workspace too small");
    const BwdArena a = carve_bwd(ws, ws_bytes, N);
    STRIVE_CHECK_ARG(a.ok, "workspace arena overflow");
    launch_dgrad_mfma_layer(layer, a.G[layer], a.dfrag, a.G[layer - 1], N, (hipStream_t)stream_);
    STRIVE_CHECK_LAUNCH();
    return 0;
}

// d_feat (N,64) -> CNN weight gradients at the N poses `pos`, ACCUMULATED into d_params (flat, see strive_hip.h).
// `keep`: the activations of these N samples as strive_map_cnn_fwd_keep left them (conv1 .. conv4 are then not run again).
static int cnn_backward(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                        const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat,
                        float* d_params, void* ws, size_t ws_bytes, strive_stream_t stream_, const CnnKeep* keep, size_t keep_off = 0) {
    using namespace cnnbwd;
    STRIVE_CHECK_ARG(map && cnn && pos && mapix && d_feat && d_params && ws && pos_mean4_host && pos_std4_host, "null argument");
    STRIVE_CHECK_ARG(map->C == 4 && map->L == 256 && map->Wc == 256, "the HIP map CNN supports the default 4x256x256 crop only");
    STRIVE_CHECK_ARG(cnn->w_torch[0] && cnn->fc_wt, "map_cnn_bwd needs the torch-layout weights (StriveCNN.w_torch)");
    if (N <= 0) return 0;
    STRIVE_CHECK_ARG(ws_bytes >= strive_map_cnn_bwd_workspace_bytes(N), "workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int ch = N < BWD_CHUNK ? N : BWD_CHUNK;
    const BwdArena arena = carve_bwd(ws, ws_bytes, ch);
    STRIVE_CHECK_ARG(arena.ok, "workspace arena overflow");
    char* fwd_ws = arena.fwd_ws;
    const size_t fwd_bytes = arena.fwd_bytes;
    float* const* G = arena.G;
    uint8_t* crop = arena.crop;
    float2* mr = arena.mr;
    double* S = arena.S;
    float* feat = arena.feat;
    uint4* dfrag = arena.dfrag;
    float* wpart = arena.wpart;
    const bool dgrad_igemm = strive_tuning().dgrad_igemm != 0;     // A/B switch: the fp32 implicit-GEMM form
    if (!dgrad_igemm) {
        DgradPackArgs pa;
        for (int l = 0; l < 6; ++l) pa.w[l] = cnn->w_torch[l];
        const int total = (int)dgrad_frag_total();
        hipLaunchKernelGGL(dgrad_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, pa, dfrag, total);
    }
    const CnnGradPtrs gp = cnn_grad_ptrs(d_params);

    for (int n0 = 0; n0 < N; n0 += ch) {
        const int n = (N - n0) < ch ? (N - n0) : ch;
        // forward recompute of this chunk: raw convolution outputs and GroupNorm partial sums land in fwd_ws
        int rc = 0;
        if (!keep) {
            rc = cnn_run(map, cnn, pos + (size_t)n0 * 4, pos_mean4_host, pos_std4_host, mapix + n0, nullptr, n, feat, fwd_ws,
                         fwd_bytes, stream, /*keep_tail_activations=*/true);
            if (rc) return rc;
        }
        rc = strive_map_crop_u8(map, pos + (size_t)n0 * 4, pos_mean4_host, pos_std4_host, mapix + n0, n, crop, stream_);
        if (rc) return rc;
        // the same carve-up cnn_run used for n samples
        StriveArena fa(fwd_ws, fwd_bytes);
        float* act[6];
        for (int l = 0; l < 6; ++l) act[l] = fa.take<float>((size_t)n * L_OUT[l]);
        GNStats* stats = fa.take<GNStats>((size_t)n * STAT_SLOTS);
        GNStats* st[6];
        stat_slots(stats, (size_t)n, st);
        if (keep) {
            // all six layers' raw outputs of these samples are on the kept arrays (the forward ran the standard chain and the fused
            // tail wrote conv5 / conv6's on its way: NPARTS slots per sample)
            for (int l = 0; l < 6; ++l) {
                act[l] = keep->act[l] + (keep_off + (size_t)n0) * L_OUT[l];
                st[l] = keep->st[l] + (keep_off + (size_t)n0) * NPARTS[l];
            }
        }
        {
            MomentsArgs ma;
            for (int l = 0; l < 6; ++l) { ma.st[l] = st[l]; ma.nparts[l] = NPARTS[l]; ma.count[l] = (double)L_OUT[l]; }
            hipLaunchKernelGGL(moments_kernel, dim3((n + 63) / 64, 6), dim3(64), 0, stream, ma, mr, ch, n);
        }
        hipMemsetAsync(S, 0, (size_t)6 * ch * 2 * sizeof(double), stream);      // GroupNorm-backward sums of all six layers
        hipLaunchKernelGGL(fc_bwd_kernel, dim3((n + FCB_S - 1) / FCB_S, 512 / FCB_K), dim3(256), 0, stream, act[5], mr + (size_t)5 * ch, cnn->gn_g[5],
                           cnn->gn_b[5], cnn->fc_wt, d_feat + (size_t)n0 * 64, G[5], gp.fcw, gp.fcb, n);
        for (int l = 5; l >= 0; --l) {
            const LayerDesc d = layer_desc(l);
            const int M = d.cout * d.oh * d.oh;
            int nblk = (M + 8191) / 8192;
            if (nblk < 1) nblk = 1;
            double* Sl = S + (size_t)l * ch * 2;
            if (d.out_oct) {
                const int HW = d.oh * d.oh;
                const dim3 grid((d.cout / 8) * ((HW + GN_PPB - 1) / GN_PPB), n);
                hipLaunchKernelGGL(gn_bwd_reduce_oct_kernel, grid, dim3(256), 0, stream, d.cout, HW, act[l], mr + (size_t)l * ch,
                                   cnn->gn_g[l], cnn->gn_b[l], G[l], Sl, gp.g[l], gp.be[l]);
                hipLaunchKernelGGL(gn_bwd_apply_oct_kernel, grid, dim3(256), 0, stream, d.cout, HW, act[l], mr + (size_t)l * ch,
                                   cnn->gn_g[l], cnn->gn_b[l], G[l], Sl, gp.b[l]);
            } else {
                hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(nblk, n), dim3(256), 0, stream, d, act[l], mr + (size_t)l * ch, cnn->gn_g[l],
                                   cnn->gn_b[l], G[l], Sl, gp.g[l], gp.be[l]);
                hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nblk, n), dim3(256), 0, stream, d, act[l], mr + (size_t)l * ch, cnn->gn_g[l],
                                   G[l], Sl, gp.b[l]);
            }
            // weight gradient
            WgradProb wp;
            wp.d = d; wp.M = d.cout; wp.NS = n; wp.dy = G[l];
            wp.act_in = l > 0 ? act[l - 1] : nullptr;
            wp.crop = l > 0 ? nullptr : crop;
            wp.mr_in = l > 0 ? mr + (size_t)(l - 1) * ch : nullptr;
            wp.gam_in = l > 0 ? cnn->gn_g[l - 1] : nullptr;
            wp.bet_in = l > 0 ? cnn->gn_b[l - 1] : nullptr;
            wp.dW = gp.w[l];
            const bool use_igemm = strive_tuning().wgrad_igemm != 0;      // A/B switches: the round-2 implicit-GEMM form,
            const bool use_tile = strive_tuning().wgrad_tile != 0;        // the fp32 LDS-tile form
            if (use_igemm) {
                const int K = n * d.oh * d.oh, NN = d.cin * d.ks * d.ks;
                wp.kchunk = 2048;
                hipLaunchKernelGGL(igemm64_kernel<WgradProb>, dim3((NN + 63) / 64, (d.cout + 63) / 64, (K + wp.kchunk - 1) / wp.kchunk),
                                   dim3(256), 0, stream, wp);
            } else if (use_tile) {
                launch_wgrad_tile_layer(l, G[l], wp.act_in, wp.crop, wp.mr_in, wp.gam_in, wp.bet_in, gp.w[l], n, stream);
            } else {
                launch_wgrad_mfma_layer(l, G[l], wp.act_in, wp.crop, wp.mr_in, wp.gam_in, wp.bet_in, gp.w[l], wpart, n, stream);
            }
            if (l > 0 && !dgrad_igemm) {
                launch_dgrad_mfma_layer(l, G[l], dfrag, G[l - 1], n, stream);
            } else if (l > 0) {
                // (a staged fp32 data-gradient kernel in the style of wgrad_tile_kernel -- thread = input pixel, weights through
                // the scalar cache -- was measured slower than this form: 2.19 vs 1.36 ms per 64-sample call)
                DgradProb dp;
                dp.d = d; dp.M = d.cin; dp.dy = G[l]; dp.w = cnn->w_torch[l]; dp.gin = G[l - 1];
                const int nmax = ((d.ih + 1) / 2) * ((d.ih + 1) / 2);
                hipLaunchKernelGGL(igemm64_kernel<DgradProb>, dim3((nmax + 63) / 64, (d.cin + 63) / 64, 4 * n), dim3(256), 0, stream, dp);
            }
        }
    }
    STRIVE_CHECK_LAUNCH();
    return 0;
}

extern "C" int strive_map_cnn_bwd(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                                  const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat,
                                  float* d_params, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    return cnn_backward(map, cnn, pos, pos_mean4_host, pos_std4_host, mapix, N, d_feat, d_params, ws, ws_bytes, stream_, nullptr);
}

extern "C" int strive_map_cnn_bwd_kept(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                                       const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat,
                                       float* d_params, const void* kept, size_t kept_bytes, void* ws, size_t ws_bytes,
                                       strive_stream_t stream_) {
    STRIVE_CHECK_ARG(kept, "null argument");
    STRIVE_CHECK_ARG(N >= 0 && kept_bytes >= cnn_keep_bytes((size_t)N), "kept-activation buffer too small");
    CnnKeep k;
    STRIVE_CHECK_ARG(cnn_keep_carve(const_cast<void*>(kept), kept_bytes, (size_t)N, k), "kept-activation arena overflow");
    return cnn_backward(map, cnn, pos, pos_mean4_host, pos_std4_host, mapix, N, d_feat, d_params, ws, ws_bytes, stream_, &k);
}

// ... over rows [kept_offset, kept_offset + N) of a kept buffer sized for kept_total crops (pos / mapix / d_feat point at the first of
// these rows): the training rollout's backward hands the crops of a few steps at a time to a side stream while its sweep continues
extern "C" int strive_map_cnn_bwd_kept_range(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                                             const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat,
                                             float* d_params, const void* kept, size_t kept_bytes, int32_t kept_total,
                                             int32_t kept_offset, void* ws, size_t ws_bytes, strive_stream_t stream_) {
    STRIVE_CHECK_ARG(kept, "null argument");
    STRIVE_CHECK_ARG(N >= 0 && kept_offset >= 0 && (int64_t)kept_offset + N <= (int64_t)kept_total, "rows outside the kept arrays");
    STRIVE_CHECK_ARG(kept_bytes >= cnn_keep_bytes((size_t)kept_total), "kept-activation buffer too small");
    CnnKeep k;
    STRIVE_CHECK_ARG(cnn_keep_carve(const_cast<void*>(kept), kept_bytes, (size_t)kept_total, k), "kept-activation arena overflow");
    return cnn_backward(map, cnn, pos, pos_mean4_host, pos_std4_host, mapix, N, d_feat, d_params, ws, ws_bytes, stream_, &k,
                        (size_t)kept_offset);
}
