/* strive_hip.h -- C ABI of libstrive_hip.so: STRIVE's latent-optimisation hot path on MI355X (gfx950).
 *
 * The reference (nv-tlabs/STRIVE) is pure Python/PyTorch and has no FFI; its boundary for this path is
 * the Python surface of src/models/traffic_model.py, src/models/interaction_net.py,
 * src/datasets/nuscenes_utils.py and src/losses/adv_gen_nusc.py.  Each entry point below names the
 * reference function (file:line under /root/reference) whose arithmetic it replaces; INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add at those lines.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the comment says "host";
 *   - fp32 tensors are row-major contiguous; index tensors are int32; the raster is uint8, metres-per-
 *     pixel is float64 (as the reference keeps it, src/datasets/map_env.py:166);
 *   - nothing is allocated or freed: the caller passes outputs and a workspace whose size comes from the
 *     matching *_bytes() query; workspaces are scratch (contents undefined on return) except "tape";
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), no hidden synchronisation;
 *   - return 0 on success, a negative code on failure; strive_last_error() (thread local) explains it;
 *   - re-entrant; no global mutable state besides the error string and the options below.  One exception (round 5): strive_rollout_bwd_train_kept forks
 *     the map CNN's backward onto a library-owned side stream (one per device, created at first use) and joins it back on `stream`
 *     with events before it returns -- no host synchronisation, the caller's stream order is what it would be without it; one
 *     such call at a time per device: a second host thread entering while the first is still enqueueing gets -3 (ABI 17; it was
 *     a documented rule only), and a device on which the stream or its events cannot be created runs without the overlap.
 */
#ifndef STRIVE_HIP_H
#define STRIVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STRIVE_ABI_VERSION 17
#define STRIVE_HID 128        /* hidden width of every MLP in the reference (models/common.py, interaction_net.py:32,41) */
#define STRIVE_MAX_LAYERS 4
#define STRIVE_ZDIM 32
#define STRIVE_FEAT 64        /* map / past feature size and GRU hidden size (traffic_model.py:25-27) */

typedef void* strive_stream_t;

int strive_abi_version(void);
const char* strive_last_error(void);

/* Options (ABI 17).  Up to round 5 the library read 19 undeclared environment variables inside its entry points; it reads none
 * any more.  Every switch is a named process-wide integer with the shipped behaviour as its default; a caller that never touches
 * them gets exactly the documented behaviour.  Set them between calls (a call reads them when it starts); they are plain ints, not
 * synchronised against calls running on other threads.  Reference: none (the reference has no native side to tune); the Python host
 * (strive_amd/_lib.py) maps an environment variable STRIVE_<NAME> onto option <name> when the library is loaded and on
 * sync_options_from_env().
 *   algorithm choices (results agree to rounding, mostly bit for bit -- the A/B switches of DESIGN.md section 4):
 *     cnn_small_batch (96)   map CNN: a launch of <= this many samples runs the small-batch kernel forms (0: never)
 *     cnn_chunk (512)        samples pushed through the CNN layer stack together (8 .. 1024)
 *     cnn_tail_s (0)         samples per workgroup of the fused CNN tail: 0 = by size, or 1 / 2 / 4
 *     conv_ws (1)            conv2 on specialised producer / consumer waves (0: conv_bf6_kernel; bit-identical)
 *     conv_wsx (1)           conv3 / conv4 on specialised waves with streamed weights (0: conv_bf6_kernel; bit-identical)
 *     scene_kernels (1)      scene-resident decoder kernels where they apply (0: launch-per-phase kernels)
 *     scene_tiles (1)        batches with scenes of > 16 agents: the forward step's node-level phases on the scene kernel in 16-row
 *                            tiles (0: launch-per-phase kernels; results agree to fp32 rounding)
 *     scene_split (12)       scenes of >= this many agents share their edge chunks among K workgroups (0: never)
 *     scene_fwd_k (-1)       K of the forward step (-1: one workgroup per 64-row edge chunk, <= 4)
 *     sweep_step (-1)        reverse sweep as one launch per step on K workgroups per scene (-1: from 3 chunks on; 0: never; 1..4)
 *     train_overlap (1)      strive_rollout_bwd_train_kept: CNN backward of finished steps on the library's side stream
 *     train_overlap_rows (256)  samples per hand-over to that stream
 *     wgrad_atomics, dgrad_igemm, wgrad_igemm, wgrad_tile (0)   earlier forms of the training backward kept for A/B
 *   measurement hooks (results of the affected call are INVALID or the workspace tail is written; never set in production):
 *     conv_ws_dbg (bit mask), wgrad_dbg, scene_prof, planner_prof, planner_dbg (bit mask)
 * strive_set_option / strive_get_option return 0, or -1 for an unknown name or a value outside the option's range
 * (strive_last_error() names it); strive_option_name(i), 0 <= i < strive_option_count(), enumerates the names. */
int strive_set_option(const char* name_host, int64_t value);
int strive_get_option(const char* name_host, int64_t* value_host);
int strive_reset_options(void);
int32_t strive_option_count(void);
const char* strive_option_name(int32_t i);

/* ------------------------------------------------------------------------------------------------
 * Shared descriptors (host structs holding device pointers)
 * ---------------------------------------------------------------------------------------------- */

/* Linear -> (LayerNorm -> ReLU -> Linear)* ; reference src/models/common.py:8-44.
 * w[l]  : torch layout (dims[l+1], dims[l])      -- used by the backward (input-gradient) kernels
 * wt[l] : transposed   (dims[l],   dims[l+1])    -- used by the forward kernels
 * ln_g/ln_b[l] : LayerNorm(dims[l+1]) applied to the output of layer l, l < nlayers-1 (eps 1e-5). */
typedef struct StriveMLP {
    int32_t nlayers;
    int32_t dims[STRIVE_MAX_LAYERS + 1];
    const float* w[STRIVE_MAX_LAYERS];
    const float* wt[STRIVE_MAX_LAYERS];
    const float* b[STRIVE_MAX_LAYERS];
    const float* ln_g[STRIVE_MAX_LAYERS];
    const float* ln_b[STRIVE_MAX_LAYERS];
    /* optional (NULL = the layer runs on the vector ALUs from w / wt): W_l * wsc[l] (wf) and its transpose (wbf) split into two
     * fp16 pieces (w0 = fp16(w) to nearest, w1 = fp16(w - w0)) in the operand order of v_mfma_f32_16x16x32_f16:
     * [row tile = row / 16][k-step = k / 32][piece][lane 0..63][8 x fp16], lane l = row 16 tile + (l & 15),
     * k = 32 step + 8 (l >> 4) + j, zero beyond the matrix; rows = outputs and k = inputs for wf, the reverse for wbf.
     * wsc[l] is a power of two. */
    const void* wf[STRIVE_MAX_LAYERS];
    const void* wbf[STRIVE_MAX_LAYERS];
    float wsc[STRIVE_MAX_LAYERS];
} StriveMLP;

/* One message-passing round over per-scene cliques; reference src/models/interaction_net.py:16-218
 * (k=1, MLP update, max aggregation).  edge.dims[0] = 2*(D+NC)+4, update.dims[0] = 2*D+NC. */
typedef struct StriveGNN {
    StriveMLP mlp_in, edge, update, mlp_out;
    int32_t D;   /* node embedding size during message passing: 64 (decoder) or 128 (prior/posterior) */
    int32_t NC;  /* semantic classes */
} StriveGNN;

/* 3-layer GRU memory, hidden 64, input 4; reference src/models/traffic_model.py:151-156.
 * wih_t[l]: (in_l, 192) with in_0 = 4, in_1 = in_2 = 64;  whh_t[l]: (64, 192); gate order r,z,n.
 * wih[l], whh[l]: torch layouts (192, in_l), (192, 64) for the backward kernels.
 * Optional matrix-core operands (ABI 14; NULL = the scene-resident rollout kernels are not used): whh_f[l] / wih_f[l] = the
 * two-piece fp16 fragments (StriveMLP.wf layout) of whh[l] * hh_sc[l] / wih[l] * ih_sc[l] (rows = the 192 gate outputs),
 * whh_bf[l] / wih_bf[l] those of the transposes (rows = the 64 inputs: the input-gradient products of the reverse sweep).
 * Layer 0's input is 4 wide: wih_f[0] / wih_bf[0] stay NULL (that product runs on the vector ALUs). */
typedef struct StriveGRU {
    const float* wih[3];
    const float* whh[3];
    const float* wih_t[3];
    const float* whh_t[3];
    const float* bih[3];
    const float* bhh[3];
    const void* whh_f[3];
    const void* wih_f[3];
    const void* whh_bf[3];
    const void* wih_bf[3];
    float hh_sc[3];
    float ih_sc[3];
} StriveGRU;

/* Rasterised maps + crop geometry; reference src/datasets/map_env.py:50-61,165-166 and
 * src/datasets/nuscenes_utils.py:205-232.  lwise/wwise are the fp32 linspace tables over the crop
 * bounds ([-17,60] m along the heading, [-38.5,38.5] m across), computed by the caller. */
typedef struct StriveMap {
    const uint8_t* raster;   /* (M, C, H, W) */
    const double* dx;        /* (M, 2) metres per pixel: [m][0] divides x, [m][1] divides y */
    int32_t M, C, H, W;
    const float* lwise;      /* (L)  */
    const float* wwise;      /* (Wc) */
    int32_t L, Wc;
    const uint32_t* raster_px4;   /* optional (M, H, W) copy with the 4 layers of a pixel packed into one little-endian
                                     word (byte c = layer c), or NULL; lets the fused crop gather 4 layers with ONE load */
} StriveMap;

/* Map CNN: 6 x [Conv2d(stride 2, pad 0) -> GroupNorm(1 group) -> ReLU] + Linear(512, 64), default
 * architecture only (kernels 7,5,5,3,3,3; channels 4->16->32->64->64->128->128; 256x256 input);
 * reference src/models/traffic_model.py:69-87, 437-440.
 * w[l]: not read by the kernels any more (all six convolutions take the fp16 fragment tables w1_frag .. w6_frag below; the
 * host side points it at w_torch[l]); kept so that the struct layout stays put;
 * fc_wt: (512, 64) transposed Linear weight. */
typedef struct StriveCNN {
    const float* w[6];
    const float* b[6];
    const float* gn_g[6];
    const float* gn_b[6];
    const float* fc_wt;
    const float* fc_b;
    const uint32_t* w1_frag;   /* layer-0 weights * wscale[0] split into two fp16 pieces (w0 = fp16(w) rounded to nearest,
                                  w1 = fp16(w - w0): w0 + w1 = w up to 2^-24 |w|) in MFMA fragment order
                                  [ky][piece][lane 0..63][8 x fp16], 8 values = window columns 2g,2g+1 x 4 layers for
                                  lane group g = lane/16, output channel = lane%16; 14336 bytes */
    const uint32_t* w2_frag;   /* layer-1 (16->32, 5x5), layer-2 (32->64, 5x5) and layer-3 (64->64, 3x3) weights, each * wscale[l] */
    const uint32_t* w3_frag;   /* and split into two fp16 pieces, in the fragment order of conv_bf6_kernel: [pass = ci/8][step s][co/32] */
    const uint32_t* w4_frag;   /* (w5_frag, w6_frag: layers 4 and 5, 64->128 and 128->128, 3x3, same format)  [piece 2][lane 64][8 x fp16]; lane half h = lane/32 holds one window tap of the step, element
                                  e = input channel 8*pass + e, output channel = 32*(co/32) + lane%32.
                                  5x5 (13 steps): tap (s/2, (s&1)+2h) for s < 10, (2(s-10)+h, 4) for s = 10, 11, (4, 4) or zero
                                  for s = 12.  3x3 (5 steps): (s, 2h) for s < 3, (h, 1) for s = 3, (2, 1) or zero for s = 4.
                                  53248 / 212992 / 163840 / 327680 / 655360 bytes */
    const uint32_t* w5_frag;
    const uint32_t* w6_frag;
    const float* w_torch[6];   /* the convolution weights in torch layout (co, ci, ky, kx): read by the training backward's
                                  data-gradient kernel only (may be NULL on the latent-optimisation path) */
    float wscale[6];           /* power of two the fp16 weight pieces of layer l were multiplied by (so that both pieces sit
                                  in fp16's normal range) */
    float xscale[6];           /* power of two the GroupNorm+ReLU input of layer l (l >= 1) is multiplied by before its fp16
                                  split; chosen from the bound |gamma| sqrt(C H W) + |beta| so that it cannot overflow */
    int32_t conv2_plain;       /* (ABI 14) conv2 normally runs as ONE persistent workgroup per CU (specialised producer / consumer
                                  waves).  A caller that runs small latency-bound kernels on a second stream under the CNN (the
                                  planner rollout behind one decoder rollout while the other rollout's CNN runs, reference
                                  src/utils/adv_gen_optim.py:133-139) passes a descriptor with conv2_plain != 0 for those calls:
                                  the ordinary conv2 kernel.  The two kernels write bit-identical activations; their GroupNorm
                                  partial sums are added in a different order (8 rows against 4 row pairs per tile), so later
                                  layers agree to fp32 rounding, not bit for bit.  A field of the caller's descriptor, not a
                                  switch inside the library: no global mutable state. */
} StriveCNN;

/* Scene structure of a batch: agents of scene b are rows ptr[b] .. ptr[b+1]-1, ego first
 * (reference src/datasets/nuscenes_dataset.py:678-702, PyG Batch.ptr).  With NS > 1 every per-agent
 * tensor has rows r = agent * NS + sample (the reference's (NA, NS, .) layout flattened). */
typedef struct StriveScenes {
    int32_t NA, NS, B;
    int32_t max_n;            /* largest scene size (host-known; sizes per-edge scratch) */
    int64_t n_edges;          /* (ABI 14) sum over scenes of n (n - 1) NS directed edge rows (host-known; sizes the training sweep's
                                 per-edge row tapes exactly instead of NA NS max_n) */
    const int32_t* ptr;       /* (B+1) */
    const int32_t* scene_of;  /* (NA)  */
} StriveScenes;

/* Everything the decoder rollout needs besides per-call tensors. */
typedef struct StriveDecoder {
    StriveGNN gnn;
    StriveGRU gru;
    StriveCNN cnn;
    StriveMap map;
    float state_mean[6], state_std[6];   /* (x,y,hx,hy,s,hdot) normaliser, datasets/utils.py:44-113 */
    float att_mean[2], att_std[2];       /* (l,w) normaliser */
    float a_mean, a_std, ddh_mean, ddh_std, dt, max_hdot, max_s;   /* NUSC_BIKE_PARAMS, datasets/utils.py:121-127 */
    /* optional (ABI 14; NULL = the scene-resident kernels are not used): the small parameters of decoder_net / decoder_memory in
     * one block of 6340 floats that those kernels copy to LDS -- offsets in floats (csrc/scene_rollout.h Par):
     * mlp_in  b0 0, b1 128, b2 256, ln_g0 320, ln_b0 448, ln_g1 576, ln_b1 704;
     * edge    b0 832, b1 960, b2 1088, ln_g0 1152, ln_b0 1280, ln_g1 1408, ln_b1 1536, W_rel^T (4,128) = rows 128+2NC.. of wt[0] 1664;
     * update  b0 2176, b1 2304, ln_g0 2368, ln_b0 2496;
     * mlp_out b0 2624, b1 2752, b2 2880 (2 + 2 pad), ln_g0 2884, ln_b0 3012, ln_g1 3140, ln_b1 3268, w[2] (2,128) 3396;
     * GRU     b_ih (3,192) 3652, b_hh (3,192) 4228, wih_t[0] (4,192) 4804, wih[0] (192,4) 5572. */
    const float* scene_par;
} StriveDecoder;

/* ------------------------------------------------------------------------------------------------
 * Map raster lookups
 * ---------------------------------------------------------------------------------------------- */

/* get_map_obs (reference src/datasets/nuscenes_utils.py:234-264) via NuScenesMapEnv.get_map_crop
 * (src/datasets/map_env.py:168-203).  pos (N,4) is used as pos*pos_std + pos_mean (pass 1/0 for
 * already-unnormalised frames); mapix (N) selects the map.  out: (N, C, L, Wc) uint8, bit-exact. */
int strive_map_crop_u8(const StriveMap* map, const float* pos, const float* pos_mean4_host,
                       const float* pos_std4_host, const int32_t* mapix, int32_t N, uint8_t* out,
                       strive_stream_t stream);

/* NuScenesMapEnv.__init__'s rasterisation (reference src/datasets/map_env.py:79-166): one layer of one map from polygon / line
 * geometry.  THE REFERENCE TAKES THE GEOMETRY AND ITS RASTERISER FROM THE nuscenes DEVKIT (NuScenesMap.get_map_mask ->
 * cv2.fillPoly / cv2.polylines), which is not installed here and cannot be: no fixture of its output can exist, so this entry
 * point is pinned to its own stated rule by an exact rational-arithmetic oracle (oracle/raster.py), not to the devkit.
 * Rule: pixel (r, c) = the layer's value at the world point the crop looks it up for, (c dx_x, r dx_y) (get_map_obs reads
 * raster[round(y / dx_y), round(x / dx_x)], nuscenes_utils.py:250-263): 1 inside or on the boundary of a polygon (even-odd over the
 * rings of a shape = holes) or within half_width of a polyline.  Sets bytes to 1, never clears: the caller zero-fills the layer, and
 * several jobs may target one layer (the reference collapses the road layers into channel 0, :108-113).
 * Tables (device, built by the host): shapes s = rings ring_ptr[shape_ptr[s] .. shape_ptr[s+1]]; ring r = vertices
 * verts[ring_ptr[r] .. ring_ptr[r+1]] (x, y in metres, float64; polygon rings close implicitly); shape_kind[s] 0 = polygon, 1 = line;
 * tile t = 32 x 32 pixels, row-major over ceil(H/32) x ceil(W/32): the shapes whose bounding box (grown by half_width) touches it are
 * tile_shapes[tile_ptr[t] .. tile_ptr[t+1]].  flip_rows: write row H-1-r (the reference flips the Singapore maps about the x axis,
 * :125-127).  out_layer (H, out_pitch) uint8. */
typedef struct StriveRasterJob {
    const double* verts;
    const int32_t* ring_ptr;
    const int32_t* shape_ptr;
    const uint8_t* shape_kind;
    const int32_t* tile_ptr;
    const int32_t* tile_shapes;
    int32_t H, W, out_pitch, flip_rows;
    double dx_x, dx_y, half_width;
} StriveRasterJob;

int strive_map_rasterize(const StriveRasterJob* job, uint8_t* out_layer, strive_stream_t stream);

/* get_coll_point (reference src/datasets/nuscenes_utils.py:334-390) on raster layer 0.
 * cars (N,4) unnormalised, lw (N,2); gl/gw = grid size (host computes it from the batch mean like the
 * reference, lines 351-354); lin_l (gl), lin_w (gw) = fp32 linspace(-1,1,.) tables.
 * out_pt (N,2): collision point or NaN; out_cnt (N) int32: number of non-drivable samples. */
int strive_coll_point(const StriveMap* map, const float* cars, const float* lw, const int32_t* mapix,
                      int32_t N, int32_t gl, int32_t gw, const float* lin_l, const float* lin_w,
                      float* out_pt, int32_t* out_cnt, strive_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Map CNN (crop fused into the first convolution)
 * ---------------------------------------------------------------------------------------------- */

size_t strive_map_cnn_workspace_bytes(int32_t N);

/* encode_map (reference src/models/traffic_model.py:416-451): feat (N,64) for N poses. */
int strive_map_cnn_fwd(const StriveMap* map, const StriveCNN* cnn, const float* pos,
                       const float* pos_mean4_host, const float* pos_std4_host, const int32_t* mapix,
                       int32_t N, float* feat, void* ws, size_t ws_bytes, strive_stream_t stream);

/* The same CNN on an explicit crop (N,4,256,256) uint8 -- for parity tests of the convolution stack. */
int strive_map_cnn_fwd_from_crop(const StriveCNN* cnn, const uint8_t* crop, int32_t N, float* feat,
                                 void* ws, size_t ws_bytes, strive_stream_t stream);

/* Measurement hook for bench.py: launch ONE kernel of the stack (layer 0 = fused crop+conv1, 1..3 = conv2..4,
 * 7 = the fused conv5 + conv6 + Linear kernel strive_map_cnn_fwd runs; 4, 5, 6 = the separate conv5 / conv6 /
 * GroupNorm+Linear kernels of the training recompute, in that order) on the activations a previous
 * strive_map_cnn_fwd over the same N poses left in `ws`, so a single kernel can be timed with events on the
 * launching stream. */
int strive_map_cnn_bench_layer(const StriveMap* map, const StriveCNN* cnn, int32_t layer, const float* pos,
                               const float* pos_mean4_host, const float* pos_std4_host, const int32_t* mapix,
                               int32_t N, float* feat, void* ws, size_t ws_bytes, strive_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Small operators
 * ---------------------------------------------------------------------------------------------- */

/* MLP.forward (reference src/models/common.py:41-44): y (rows, dims[nlayers]). */
int strive_mlp_fwd(const StriveMLP* mlp, const float* x, int32_t rows, float* y, strive_stream_t stream);

size_t strive_gnn_workspace_bytes(const StriveGNN* gnn, const StriveScenes* sc);

/* SceneInteractionNet.forward (reference src/models/interaction_net.py:52-77) on clique scenes.
 * x (R, mlp_in.dims[0]), pos (R,4) [NaN poses give a zero relative pose, :162], sem (NA,NC);
 * out (R, mlp_out out dim).  R = NA*NS. */
int strive_gnn_fwd(const StriveGNN* gnn, const StriveScenes* sc, const float* x, const float* pos,
                   const float* sem, float* out, void* ws, size_t ws_bytes, strive_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decoder rollout (forward, and backward to the latents)
 * ---------------------------------------------------------------------------------------------- */

size_t strive_rollout_tape_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT);
size_t strive_rollout_workspace_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT);

/* 1 when strive_rollout_fwd / strive_rollout_bwd will run this batch on the scene-resident kernels (one workgroup per scene:
 * a decoder step is ONE launch, the reverse sweep over all FT steps is ONE launch; csrc/scene_rollout.h) -- single-sample
 * rollouts, scenes of <= 16 agents, weight packs with matrix-core fragments (StriveMLP.wf / StriveGRU.whh_f) -- else 0: the
 * launch-per-phase kernels.  Same arithmetic scheme, same tape layout; results agree to fp32 rounding.  Option
 * scene_kernels = 0 (read per call) forces 0.  2 (round 6): a single-sample batch with scenes of more than 16 agents -- the node-level
 * phases of the forward step (mlp_in .. edge partials; update MLP .. GRU .. dynamics) run on the scene kernel in 16-row tiles of
 * every scene, the edge rows and the reverse sweep on the launch-per-phase kernels (option scene_tiles = 0: all per-phase). */
int strive_rollout_scene_resident(const StriveDecoder* dec, const StriveScenes* sc);

/* TrafficModel.autoregressive_decoder (reference src/models/traffic_model.py:589-704).
 * past_last (NA,6) normalised last past state; lw (NA,2) normalised; sem (NA,NC); past_feat, map_feat
 * (NA,64); z (R,32); mapix (NA); ext_future (B,FT,4) normalised or NULL (ego rows teacher-forced,
 * lines 667-675).  traj (R,FT,4): normalised global (x,y,hx,hy).  tape keeps what the backward needs. */
int strive_rollout_fwd(const StriveDecoder* dec, const StriveScenes* sc, const float* past_last,
                       const float* lw, const float* sem, const float* past_feat, const float* map_feat,
                       const float* z, const int32_t* mapix, const float* ext_future, int32_t FT,
                       float* traj, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                       strive_stream_t stream);

/* d(loss)/dz given d(loss)/d(traj): the reverse-time sweep of SURVEY.md Appendix A (no gradient flows
 * through the map crop / CNN: the reference crops at pos.detach(), traffic_model.py:694). */
int strive_rollout_bwd(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                       const float* z, const float* ext_future, int32_t FT, const float* d_traj,
                       float* dz, const void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                       strive_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Collision penalties
 * ---------------------------------------------------------------------------------------------- */

/* VehCollLoss.forward (reference src/losses/adv_gen_nusc.py:464-512) restricted to in-scene pairs.
 * traj (NA,T,4) unnormalised; cent_x (NA,5) circle centres on the length axis; rad (NA).
 * Pair enumeration: for each t, for each agent i (global order), for each j in scene(i) (ascending,
 * j == i included as an always-invalid slot): index  t*P + pair_off[i] + (j - ptr[scene(i)]),
 * P = sum_b n_b^2.  pen (T,P) = 1 - dmin/(r_i+r_j+buffer); hit (T,P) uint8 = dmin <= that distance and
 * i != j; amin (T,P) uint8 = argmin over the 25 circle pairs (ci*5+cj). */
int strive_veh_coll_fwd(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj,
                        int32_t T, const float* cent_x, const float* rad, float buffer, float* pen,
                        uint8_t* hit, uint8_t* amin, strive_stream_t stream);

/* d_traj (NA,T,4) += sum over pairs of d_pen * d(pen)/d(traj) (both members of each pair). */
int strive_veh_coll_bwd(const StriveScenes* sc, const int32_t* pair_off, int32_t P, const float* traj,
                        int32_t T, const float* cent_x, const float* rad, float buffer, const float* d_pen,
                        const uint8_t* amin, float* d_traj, strive_stream_t stream);

/* interp_traj (reference src/losses/adv_gen_nusc.py:625-644): linear x`scale` up-sampling in time of (x,y,hx,hy)
 * followed by heading renormalisation.  in (N,T,4) -> out (N,TO,4), TO = T*scale.  i0,i1 (TO) int32 and w0,w1 (TO) fp32
 * are the two taps of every output step as F.interpolate(mode='linear', align_corners=False) computes them. */
int strive_interp_traj_fwd(const float* in, int32_t N, int32_t T, int32_t TO, const int32_t* i0, const int32_t* i1,
                           const float* w0, const float* w1, float* out, strive_stream_t stream);
int strive_interp_traj_bwd(const float* in, const float* d_out, int32_t N, int32_t T, int32_t TO, int32_t scale,
                           const int32_t* i0, const int32_t* i1, const float* w0, const float* w1, float* d_in,
                           strive_stream_t stream);

/* AvoidCollLoss (reference src/losses/adv_gen_nusc.py:264-341) as ONE call per direction: up-sampling (:625-644), the
 * in-scene circle penalties (:405-512), the off-road collision points and their penalty (:366-403), the prior NLL
 * (:343-364), the init-z distance and the weighted masked means -- what the reference evaluates with ~60 elementwise
 * torch operators per closure.  Constants of one batch: */
typedef struct StriveAvoidColl {
    const int32_t* pair_off;     /* (NA)  first in-scene pair slot of every agent, P slots in all */
    int32_t P;
    const float* cent_x;         /* (NA,5) circle offsets along the heading */
    const float* rad;            /* (NA)   circle radii */
    float buffer;                /* veh_coll_buffer */
    const uint8_t* pair_valid;   /* (P)    1 = the slot counts (i != j, and the single-agent filter of :281-286) */
    const int32_t* i0;           /* (TO)   up-sampling taps as for strive_interp_traj_fwd */
    const int32_t* i1;
    const float* w0;
    const float* w1;
    int32_t scale;               /* TO = T * scale */
    int32_t NE;                  /* agents that take the environment term (all, or one per scene) */
    const int32_t* env_agent;    /* (NE)   their agent index */
    const int32_t* env_of_agent; /* (NA)   inverse: row e of an agent, -1 if it takes no environment term */
    const float* env_lw;         /* (NE,2) unnormalised length, width */
    const int32_t* env_mapix;    /* (NE)   */
    const float* env_pdist;      /* (NE)   sqrt(l^2/4 + w^2/4) (:373) */
    int32_t gl, gw;              /* collision-point grid (nuscenes_utils.py:351-354) */
    const float* lin_l;          /* (gl) linspace(-1,1,gl) */
    const float* lin_w;          /* (gw) */
    const float* init_z;         /* (NZ,D) */
    int32_t NZ, D;               /* latent rows (NA, or one per scene when only the first agent of a scene is optimised) */
    float prior_den, init_den;   /* denominators of the two latent means: NZ and NZ for a (NZ,D) latent; NZ and NZ*D for a
                                    (NZ,1,D) one, where the reference's sum(dim=1) runs over the singleton axis (:327-330) */
    float w_veh, w_env, w_prior, w_init;   /* loss weights; a term with weight <= 0 is skipped like the reference does */
} StriveAvoidColl;

size_t strive_avoid_coll_workspace_bytes(const StriveScenes* sc, const StriveAvoidColl* h, int32_t T);

/* traj (NA,T,4) unnormalised, z / mu / var (NZ,D).  out (8 floats): loss, mean colliding-pair penalty, mean off-road
 * penalty, mean prior NLL, mean init-z distance, #colliding valid slots, #rows with a collision point, 0.
 * `ws` keeps what the backward needs and must stay untouched until it ran. */
int strive_avoid_coll_fwd(const StriveScenes* sc, const StriveMap* map, const StriveAvoidColl* h, const float* traj,
                          int32_t T, const float* z, const float* mu, const float* var, float* out, void* ws,
                          size_t ws_bytes, strive_stream_t stream);

/* d_loss: one float ON THE DEVICE.  d_traj (NA,T,4) and d_z (NZ,D) are overwritten. */
int strive_avoid_coll_bwd(const StriveScenes* sc, const StriveAvoidColl* h, const float* traj, int32_t T, const float* z,
                          const float* mu, const float* var, const float* d_loss, void* ws, size_t ws_bytes,
                          float* d_traj, float* d_z, strive_stream_t stream);

/* AdvGenLoss (reference src/losses/adv_gen_nusc.py:53-262) as one call per direction.  `base` carries what it shares with
 * AvoidCollLoss: the pair penalties (pair_valid = i != j), the up-sampling taps, the environment term over the NON-EGO agents
 * (env_agent = their agent indices, NE = NA - B; the latent tensors z / mu / var / init_z have one row per non-ego agent in
 * the same order, so base.NZ = NE) and the latent size.  base.w_veh / w_env / w_prior / w_init are the weights 'coll_veh',
 * 'coll_env', 'motion_prior', 'init_z'; base.prior_den = NE, base.init_den is unused (the init term is a plain sum, :216-223). */
typedef struct StriveAdvGen {
    StriveAvoidColl base;
    const int32_t* ne_ptr;       /* (B+1)  offsets of every scene's non-ego agents in the non-ego order */
    const int32_t* slot_ne;      /* (P)    non-ego row of the non-ego member of a pair slot that involves the scene's ego, else -1 */
    const uint8_t* atk_mask;     /* (NE)   1 = may be the attacker (attack_agt_idx, :126-130), or NULL = everybody */
    const uint8_t* scene_alive;  /* (B) or NULL.  0 = the scene has left the batch (closed loop: its planner rollout failed,
                                    strive_planner_rollout's `alive`): it contributes to no sum and no count -- colliding pairs,
                                    off-road rows, prior / init rows, crash term, the 1/NE and 1/B of the two means -- and receives
                                    zero gradients; the other scenes get exactly the values of the batch rebuilt without it
                                    (the reference's remedy for scenes it gives up, adv_scenario_gen.py:323-356) */
    int32_t t0;                  /* crash_loss_min_time */
    int32_t use_infront;         /* crash_loss_min_infront given? */
    float infront;
    float w_crash, w_plan, w_prior_atk, w_init_atk;   /* 'adv_crash', 'coll_veh_plan', 'motion_prior_atk', 'init_z_atk' */
} StriveAdvGen;

size_t strive_adv_gen_workspace_bytes(const StriveScenes* sc, const StriveAdvGen* h, int32_t T);

/* traj (NA,T,4) and tgt (B,T,4) unnormalised; z / mu / var (NE,D).  out (16 floats): loss, mean colliding non-ego pair
 * penalty, mean weighted planner-pair penalty, mean off-road penalty, mean weighted prior NLL, weighted init-z sum, mean crash
 * term, the four counts (pairs, planner pairs, off-road rows, 1 if every attacker was always behind its target), 0...
 * soft (NE, T - t0): the per-scene soft-min weights (:133-135), rew (NE): 1 - their sum per agent. */
int strive_adv_gen_fwd(const StriveScenes* sc, const StriveMap* map, const StriveAdvGen* h, const float* traj, const float* tgt,
                       int32_t T, const float* z, const float* mu, const float* var, float* out, float* soft, float* rew,
                       void* ws, size_t ws_bytes, strive_stream_t stream);

/* d_loss: one float on the device.  d_traj (NA,T,4), d_tgt (B,T,4) and d_z (NE,D) are overwritten. */
int strive_adv_gen_bwd(const StriveScenes* sc, const StriveAdvGen* h, const float* traj, const float* tgt, int32_t T,
                       const float* z, const float* mu, const float* var, const float* d_loss, void* ws, size_t ws_bytes,
                       float* d_traj, float* d_tgt, float* d_z, strive_stream_t stream);

/* strive_coll_point over the rows (e, t) of an up-sampled trajectory tensor without expanding the per-agent attributes:
 * car = fine[agent_of[e]*TO + t], size lw[e], map mapix[e]; out_pt (NE*TO,2), out_cnt (NE*TO). */
int strive_coll_point_rows(const StriveMap* map, const float* fine, int32_t TO, const int32_t* agent_of, const float* lw,
                           const int32_t* mapix, int32_t NE, int32_t gl, int32_t gw, const float* lin_l,
                           const float* lin_w, float* out_pt, int32_t* out_cnt, strive_stream_t stream);

/* Rotated-rectangle IoU of P box pairs (x, y, hx, hy) + (l, w), float64 out; NaN where a pose contains NaN.
 * Replaces the shapely polygon loop of check_single_veh_coll / check_pairwise_veh_coll
 * (reference src/losses/adv_gen_nusc.py:517-623; corners as src/datasets/nuscenes_utils.py:416-428). */
int strive_rect_iou(const float* box_a, const float* lw_a, const float* box_b, const float* lw_b, int32_t P, double* iou,
                    strive_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training backward (weight gradients) -- reference src/train_traffic.py:103-131 calls loss.backward() through
 * TrafficModel.forward (src/models/traffic_model.py:178-225).
 *
 * Gradient buffers are flat fp32 arrays in the parameter order of the reference module (torch named_parameters()),
 * ACCUMULATED into (the caller zeroes them):
 *   MLP  (models/common.py:8-44):      per layer l:  W_l (dims[l+1], dims[l]) | b_l | and for hidden layers LN gamma_l | beta_l
 *   GNN  (models/interaction_net.py):  mlp_in | msg.0.edge_mlp | msg.0.update_mlp | mlp_out
 *   GRU  (nn.GRU(4, 64, 3)):           per layer: weight_ih (192, in_l) | weight_hh (192, 64) | bias_ih | bias_hh
 *   CNN  (traffic_model.py:69-87):     per layer: conv W (co, ci, k, k) | conv b | GroupNorm gamma | beta;  then Linear W (64, 512) | b
 * Weight gradients are summed with fp32 atomics (their summation order is not fixed run to run); input gradients are
 * deterministic.
 * ---------------------------------------------------------------------------------------------- */
size_t strive_mlp_param_count(const StriveMLP* mlp);
size_t strive_gnn_param_count(const StriveGNN* gnn);
size_t strive_gru_param_count(void);
size_t strive_map_cnn_param_count(void);

/* MLP.forward under autograd (reference src/models/common.py:41-44): dy (rows, out) -> d_params (+=) and, if dx != NULL,
 * dx (rows, in).  The forward is recomputed from x. */
int strive_mlp_bwd(const StriveMLP* mlp, const float* x, const float* dy, int32_t rows, float* dx, float* d_params,
                   strive_stream_t stream);

/* Weight pack of one Linear layer (reference src/models/common.py:26-39 stores W (out, in)): w (M, K) fp32 on the device ->
 * wt (K, M) fp32, wf / wbf = the two-piece fp16 fragments of w * scale / w^T * scale (StriveMLP.wf / .wbf layout;
 * (M+15)/16 * (K+31)/32 * 2 KiB and (K+15)/16 * (M+31)/32 * 2 KiB).  Any output may be NULL.  One launch: the training step
 * re-packs every layer after each optimiser step. */
int strive_pack_dense(const float* w, int32_t M, int32_t K, float scale, float* wt, void* wf, void* wbf, strive_stream_t stream);

/* Any two-piece fp16 fragment table of a weight tensor in one launch: out[i] (n_out x fp16) = piece idx[i] / n_w of the split of
 * w[idx[i] % n_w] * scale, or 0 where idx[i] == 2 n_w.  The index table IS the operand layout (the convolution layouts of
 * StriveCNN.w1_frag .. w6_frag); the training step re-packs the six convolutions after each optimiser step. */
int strive_pack_split_gather(const float* w, int32_t n_w, const int32_t* idx, int32_t n_out, float scale, void* out,
                             strive_stream_t stream);

size_t strive_gnn_bwd_workspace_bytes(const StriveGNN* gnn, const StriveScenes* sc);

/* SceneInteractionNet.forward under autograd (reference src/models/interaction_net.py:52-77): d_out (R, out) ->
 * dx (R, mlp_in.dims[0]) and d_params (+=).  pos carries no gradient here (the encoders are called at data poses). */
int strive_gnn_bwd(const StriveGNN* gnn, const StriveScenes* sc, const float* x, const float* pos, const float* sem,
                   const float* d_out, float* dx, float* d_params, void* ws, size_t ws_bytes, strive_stream_t stream);

size_t strive_map_cnn_bwd_workspace_bytes(int32_t N);

/* encode_map under autograd (reference src/models/traffic_model.py:416-451; the crop is data): d_feat (N,64) at the N
 * poses `pos` -> d_params (+=).  The forward is recomputed in chunks inside. */
int strive_map_cnn_bwd(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                       const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat, float* d_params,
                       void* ws, size_t ws_bytes, strive_stream_t stream);

/* Measurement hook for bench.py's training line (the counterpart of strive_map_cnn_bench_layer): launch ONE kernel of the CNN
 * backward -- the data gradient of conv layer `layer` (1..5: d conv2 .. d conv6 input) -- on the buffers a previous
 * strive_map_cnn_bwd over the same N (<= 256, one backward chunk) samples left in `ws`, so it can be timed with events on the
 * launching stream.  It overwrites the gradient buffer of layer - 1 in `ws` (scratch of the finished call). */
int strive_map_cnn_bwd_bench_dgrad(int32_t layer, int32_t N, void* ws, size_t ws_bytes, strive_stream_t stream);

/* Kept activations (round 5): the training forward may keep the raw outputs of the six convolutions and their GroupNorm partial sums
 * of every crop it encodes (1.76 MB per crop) so that the backward does not run the layers again -- the reference's autograd keeps
 * ALL activations of map_conv (src/models/traffic_model.py:69-87 under loss.backward()); only the crop is gathered again.  strive_map_cnn_fwd_keep = strive_map_cnn_fwd that also writes rows [kept_offset, kept_offset + N) of a kept buffer
 * sized for kept_total crops (strive_map_cnn_keep_bytes(kept_total)); strive_map_cnn_bwd_kept = strive_map_cnn_bwd over the N =
 * kept_total crops of such a buffer, in the buffer's row order. */
size_t strive_map_cnn_keep_bytes(int32_t N);
int strive_map_cnn_fwd_keep(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                            const float* pos_std4_host, const int32_t* mapix, int32_t N, float* feat, void* ws, size_t ws_bytes,
                            void* kept, size_t kept_bytes, int32_t kept_total, int32_t kept_offset, strive_stream_t stream);
int strive_map_cnn_bwd_kept(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                            const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat, float* d_params,
                            const void* kept, size_t kept_bytes, void* ws, size_t ws_bytes, strive_stream_t stream);

/* ... over rows [kept_offset, kept_offset + N) of a kept buffer sized for kept_total crops; pos / mapix / d_feat point at the first
 * of these rows (strive_rollout_bwd_train_kept hands the crops of a few steps at a time to a side stream while its sweep continues). */
int strive_map_cnn_bwd_kept_range(const StriveMap* map, const StriveCNN* cnn, const float* pos, const float* pos_mean4_host,
                                  const float* pos_std4_host, const int32_t* mapix, int32_t N, const float* d_feat, float* d_params,
                                  const void* kept, size_t kept_bytes, int32_t kept_total, int32_t kept_offset, void* ws,
                                  size_t ws_bytes, strive_stream_t stream);

size_t strive_rollout_train_workspace_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT);

/* strive_rollout_fwd keeping the map CNN's activations of its FT - 1 re-encoded steps in `kept` (strive_rollout_keep_bytes), and
 * strive_rollout_bwd_train reading them instead of recomputing (same results; the tape and `kept` belong to one forward call). */
size_t strive_rollout_keep_bytes(const StriveDecoder* dec, const StriveScenes* sc, int32_t FT);
int strive_rollout_fwd_keep(const StriveDecoder* dec, const StriveScenes* sc, const float* past_last, const float* lw,
                            const float* sem, const float* past_feat, const float* map_feat, const float* z, const int32_t* mapix,
                            const float* ext_future, int32_t FT, float* traj, void* tape, size_t tape_bytes, void* ws,
                            size_t ws_bytes, void* kept, size_t kept_bytes, strive_stream_t stream);
int strive_rollout_bwd_train_kept(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                                  const float* z, const float* ext_future, const int32_t* mapix, int32_t FT, const float* d_traj,
                                  float* dz, float* d_past_feat, float* d_map_feat, float* d_gnn, float* d_gru, float* d_cnn,
                                  const void* tape, size_t tape_bytes, const void* kept, size_t kept_bytes, void* ws,
                                  size_t ws_bytes, strive_stream_t stream);

/* autoregressive_decoder under autograd with parameter gradients (reference src/models/traffic_model.py:589-704 as used
 * by forward(), :178-225): like strive_rollout_bwd, plus d_past_feat, d_map_feat (NA,64) -- the adjoints of the encoder
 * outputs the rollout starts from -- and the gradients of decoder_net (d_gnn), decoder_memory (d_gru) and of the map CNN
 * (d_cnn: every step t >= 1 re-encodes the map at the detached pose, :694-695), all (+=).  NS must be 1. */
int strive_rollout_bwd_train(const StriveDecoder* dec, const StriveScenes* sc, const float* lw, const float* sem,
                             const float* z, const float* ext_future, const int32_t* mapix, int32_t FT, const float* d_traj,
                             float* dz, float* d_past_feat, float* d_map_feat, float* d_gnn, float* d_gru, float* d_cnn,
                             const void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, strive_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rule-based lane-following planner (reference src/planners/hardcode_goalcond_nusc.py), the planner that
 * adv_gen_rule_based.cfg attacks in closed loop: src/utils/adv_gen_optim.py:133-139 calls
 * HardcodeNuscPlanner.rollout once per optimisation iteration.  All arithmetic is float64 like the reference's numpy.
 * ---------------------------------------------------------------------------------------------- */
#define STRIVE_PLANNER_MAXMAPS 4
#define STRIVE_PLANNER_NSTATUS 8

/* Connections of one lane-graph node in list order (out_edges / in_edges of the reference's lane-graph dict,
 * src/datasets/nuscenes_utils.py:50-123), the first four inline; len = |xy[node] - xy[this]|. */
typedef struct StriveLaneNode {
    int32_t n;
    int32_t node[4];
    int32_t pad[3];
    double len[4];
} StriveLaneNode;

/* One map's lane graph + a uniform grid over its directed edges (each edge is listed, in ascending order, in every cell
 * its bounding box grown by `xydistmax` touches, so that get_lane_matches (:298-322) reads one cell instead of all edges). */
typedef struct StrivePlannerMap {
    const double* xy;                 /* (N,2) node positions */
    const StriveLaneNode* succ;       /* (N) */
    const StriveLaneNode* pred;       /* (N) */
    const int32_t* succ_ptr;          /* (N+1)  CSR of the same lists (used beyond the fourth connection) */
    const int32_t* succ_idx;
    const double* succ_len;
    const int32_t* pred_ptr;
    const int32_t* pred_idx;
    const double* pred_len;
    const double* edges;              /* (M,5) x0, y0, unit direction, length */
    const int32_t* edge_ix;           /* (M,2) node pair */
    const int32_t* cell_ptr;          /* (gnx*gny+1) */
    const int32_t* cell_edges;
    int32_t N, M, gnx, gny;
    double gx0, gy0, gcell;
} StrivePlannerMap;

/* PlannerConfig of the reference (:25-59) + two derived values evaluated on the host with numpy's arithmetic. */
typedef struct StrivePlannerCfg {
    double dt, preddt, xydistmax, smax, accmax, interacdist, col_plim, score_wmin, score_wfac;
    double cdistmax;                  /* 1 - cos(radians(cdistang)) */
    double tmax;                      /* nsteps * preddt */
    double predsfacs[4], predafacs[4], planaccfacs[4];
    int32_t nsteps, npredsfacs, npredafacs, nplanaccfacs, plannspeeds;
} StrivePlannerCfg;

/* The planner after reset() (:109-127): initial world of every scene, scene-sorted, the ego at position `ego_idx` of
 * its scene.  init (NO,6) = x, y, heading angle, signed speed, length, width.  Rows of `agent_obs` (the non-ego objects
 * in scene order) are described by row_obj (index into init) and row_scene. */
typedef struct StrivePlanner {
    StrivePlannerCfg cfg;
    int32_t nmaps;
    StrivePlannerMap maps[STRIVE_PLANNER_MAXMAPS];
    int32_t B, NO, NR, ego_idx;
    const int32_t* ptr;               /* (B+1) object offsets */
    const int32_t* scene_map;         /* (B) map of each scene */
    const double* init;               /* (NO,6) */
    const int32_t* row_obj;           /* (NR) */
    const int32_t* row_scene;         /* (NR) */
} StrivePlanner;

/* nstep = int(planner_t[-1] / dt) planner steps after the initial one; traj_cap = predicted trajectories kept per scene
 * and planner step (other objects x routes x speed profiles). */
size_t strive_planner_workspace_bytes(const StrivePlanner* pl, int32_t nstep, int32_t traj_cap);

/* HardcodeNuscPlanner.rollout(agent_obs, agent_t, agent_ptr, planner_t) (:178-276) for all scenes: agent_obs (NR,T,4)
 * fp32 UNNORMALISED (x, y, cos, sin) futures of the non-ego objects (NaN from the first unobserved frame on), agent_t (T),
 * t_out (nstep+1) = linspace(dt, dt*nstep, nstep+1) and planner_t (TP) float64 -> plan (B,TP,4) float64.
 * Per planner step: lane matching / clustering / breadth-first route enumeration / blended arc-length routes for every
 * object (compute_splines, :559-598), 5-circle gaps between the ego's candidate speed profiles and every predicted
 * trajectory (compute_action, :829-857), world update (update_wstate, :601-621).
 * status (B, STRIVE_PLANNER_NSTATUS) int32 on the device, zeroed by the caller once and then only ever SET by the kernels
 * (sticky over rollouts): a non-zero entry of row b names a capacity or range violation in scene b (0 matches, 1 clusters,
 * 2 chains, 3 chain nodes, 4 knots, 5 route range, 6 trajectory cap, 7 action check) -- the cases in which the reference's
 * numpy planner raises (interp1d bounds, :411-428; the speed assertion, :659-666), which there ends the run of that scene
 * (adv_scenario_gen.py:540-543 with the shipped batch_size 1).  The plan of an affected scene is NaN; every other scene of
 * the batch is planned as if the failing one were not there (scenes share nothing but this call).
 * alive (B) uint8 or NULL: written last, alive[b] = (row b of status is all zero) -- the device-side mask the closed loop
 * hands to strive_adv_gen_fwd (StriveAdvGen.scene_alive) so that a failed scene leaves the losses without a host round trip. */
int strive_planner_rollout(const StrivePlanner* pl, const double* agent_obs, const double* agent_t, int32_t T,
                           const double* t_out, int32_t nstep, const double* planner_t, int32_t TP, int32_t traj_cap,
                           double* plan, int32_t* status, uint8_t* alive, void* ws, size_t ws_bytes, strive_stream_t stream);

/* Debug view used by the parity tests: routes of one object pose (x, y, h, s) on map `mapix` as the planner builds them.
 * Outputs: nroutes (1), nk (maxr) knots per route, knots (maxr, maxk, 5) = s, x, y, cos, sin. */
int strive_planner_routes(const StrivePlanner* pl, int32_t mapix, const double* pose4, int32_t maxr, int32_t maxk,
                          int32_t* nroutes, int32_t* nk, double* knots, int32_t* status, strive_stream_t stream);

/* Operator-level views of two building blocks the rollout kernels evaluate in registers (the kernels call the same device
 * functions); used by the parity tests against the reference's own outputs.
 *
 * strive_bicycle_step: TrafficModel.sim_traj -> car_dynamics for one step (reference src/models/traffic_model.py:645-650,
 * 714-733, src/models/common.py:47-68, src/utils/transforms.py:8-29) with dec's normaliser statistics and bicycle
 * parameters: state (N,6) normalised, dec_out (N,2) the decoder's (acceleration, yaw acceleration) output, lw0 (N) the
 * normalised vehicle length -> out (N,6) normalised next state; with g_out (N,6) also the adjoints g_state (N,6) and
 * g_dec (N,2) (zero through an active clamp).
 * strive_rel_pose: transform2frame(frame, poses) (reference src/utils/transforms.py:78-139, forward branch): frame (N,4),
 * poses (N,M,4) -> out (N,M,4); with g_out also g_frame (N,4) (summed over M) and g_poses (N,M,4). */
int strive_bicycle_step(const StriveDecoder* dec, const float* state, const float* dec_out, const float* lw0, const float* g_out,
                        float* out, float* g_state, float* g_dec, int32_t N, strive_stream_t stream);
int strive_rel_pose(const float* frame, const float* poses, const float* g_out, float* out, float* g_frame, float* g_poses,
                    int32_t N, int32_t M, strive_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STRIVE_HIP_H */
