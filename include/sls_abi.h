/*
 * sls_abi.h — C-ABI of libsls_hip.so, the MI355X (gfx950) implementation of
 * the hot path behind Splat-LOAM's renderer.
 *
 * Every entry point replaces one native call the reference makes through
 * its (un-vendored) CUDA extensions; the Python binding a maintainer adds is
 * shown in INTEGRATION.md and shipped in splat_loam_amd/_abi.py.
 *
 *   reference call site (file:line under /root/reference)      replaced by
 *   ----------------------------------------------------------  -------------------------
 *   GaussianRasterizer.forward, native forward                  sls_forward_stage1/2
 *     gaussian_renderer/__init__.py:26,40-47
 *   loss.backward() -> native backward  slam/mapper.py:201      sls_backward
 *   optimizer.step() (Adam, 4 groups)   slam/mapper.py:204,     sls_adam_step
 *     scene/gaussian_model.py:97-121
 *   distCUDA2(points)  slam/mapper.py:113-115,                  sls_knn_dist2
 *     scene/gaussian_model.py:77-81
 *   GaussianRasterizer.markVisible (lineage API, unused in tree) sls_mark_visible
 *   one iteration of Mapper.optimize  slam/mapper.py:150-204     sls_mapping_step
 *   GSAligner.set_query/set_reference/align                      sls_aligner_normals/_align
 *     slam/tracker.py:141-197 ("next" row 3)
 *   render() post-processing + depth_to_normal + mapper loss     sls_consumer_fwd_bwd
 *     gaussian_renderer/__init__.py:48-93,                       ("next" row 1)
 *     utils/graphic_utils.py:26-88, slam/mapper.py:158-187
 *
 * Conventions
 *   - plain C, no torch types; all pointers are DEVICE pointers to
 *     contiguous float32/int32/uint32/uint64 arrays unless named *_host;
 *   - the CALLER owns every buffer (torch's caching allocator stays the only
 *     device allocator); the library never allocates device memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises unless stated;
 *   - return 0 on success, a negative SLS_E_* code on failure; the message
 *     is available from sls_last_error() (thread-local); nothing throws.
 */
#ifndef SLS_ABI_H
#define SLS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLS_OK 0
#define SLS_E_ARG (-1)      /* bad argument (null pointer, negative size, ...) */
#define SLS_E_HIP (-2)      /* a HIP runtime call or kernel launch failed      */
#define SLS_E_SCRATCH (-3)  /* scratch buffer too small                         */
#define SLS_E_UNSUPPORTED (-4) /* this entry point does not serve the configuration: use the one it names */

/* Camera of one LiDAR keyframe.  Filled by sls_camera_from_matrices from the
 * two matrices the reference passes in GaussianRasterizationSettings
 * (gaussian_renderer/__init__.py:16-24; conventions scene/cameras.py:43-50). */
typedef struct SlsCamera {
    int32_t H, W;           /* image_height, image_width                        */
    int32_t wrap;           /* 1: azimuth wraps (360 deg image), D5             */
    int32_t tile_cull_min;  /* D10: the binning emits only the instances of a surfel's tile rectangle whose tile the
                             * footprint can reach (include/sls_det_math.h: sls_tile_outside), for rectangles of at
                             * least this many (and at most 64) tiles.  0: the default (SLS_TILE_CULL_MIN_DEFAULT: off since
                             * round 3, include/sls_spec.h); 1: test off, every tile of the rectangle; k >= 2: threshold k */
    float fx, fy, cx, cy;   /* K = projmatrix[:3,:3]^T : u = fx*az+cx, v = fy*el+cy */
    float scale_modifier;
    float near_cut, far_cut;
    uint32_t flags;         /* SLS_CAM_LEAN_ALLMAP (bit 0): the caller neither reads allmap's median / distortion planes
                             * (5, 6) nor sends a gradient into them — true for the reference's mapper and tracker, which
                             * run at depth_ratio = 0 and use rend_dist only when meshing
                             * (gaussian_renderer/__init__.py:65-86, scene/postprocessing.py:167-170).  The staged forward
                             * then writes zeros there and sls_backward ignores dL/dallmap[5:7]: the kernels
                             * sls_mapping_step runs.  0 (set by sls_camera_from_matrices): all seven planes. */
    float Rvw[9];           /* row-major: p_view = Rvw * p_world + tvw          */
    float tvw[3];
    float pix_offset[2];    /* D1: pixel (c, r) has image coordinate (c + pix_offset[0], r + pix_offset[1]) in
                             * K-space.  (0, 0) = the lineage convention `pixf = (float)pix` (default, set by
                             * sls_camera_from_matrices); (-0.5, -0.5) = the convention of the reference's own
                             * back-projection and projector (utils/graphic_utils.py:46-49,
                             * scene/preprocessing.py:42-64).  Honoured by the pixel rays (sls_ray_tables), the
                             * centre pixel and with it the tile rectangle, the low-pass term and every cull:
                             * the rasterizer works with the principal point (cx - pix_offset[0],
                             * cy - pix_offset[1]), each rounded once in float. */
} SlsCamera;
#define SLS_CAM_LEAN_ALLMAP 1u

const char *sls_last_error(void);
int sls_version(void);

/* Compile-time tile size of the binning/render kernels (D8). */
int sls_tile_w(void);
int sls_tile_h(void);
/* floats per surfel record / gradient record (include/sls_spec.h). */
int sls_rec_stride(void);
int sls_grec_stride(void);

/* Host helper.  view/proj: row-major 4x4 float, exactly the tensors in
 * GaussianRasterizationSettings.viewmatrix / .projmatrix. */
int sls_camera_from_matrices(const float *view_host, const float *proj_host, int H, int W,
                             float scale_modifier, SlsCamera *out);

/* Host helper: per-column (cos az, sin az) and per-row (cos el, sin el) of the
 * pixel rays at the camera's pix_offset, evaluated in double and rounded once.
 * col_cs_host: 2*W floats, row_cs_host: 2*H floats.  The caller uploads them
 * (they depend on K and pix_offset only). */
int sls_ray_tables(const SlsCamera *cam, float *col_cs_host, float *row_cs_host);
/* Same at image coordinate (c + col_offset, r + row_offset) whatever cam->pix_offset
 * says; the reference's back-projection uses (-0.5, -0.5) (utils/graphic_utils.py:46-49),
 * which is what sls_consumer_fwd_bwd / sls_mapping_step expect as col_cs_half / row_cs_half. */
int sls_ray_tables_at(const SlsCamera *cam, float col_offset, float row_offset, float *col_cs_host,
                      float *row_cs_host);

/* ---- forward, stage 1: preprocess + depth order + scan ---------------------
 * col_cs / row_cs: the DEVICE copies of sls_ray_tables (the tile test of D10 takes its tile-centre directions
 * from them; needed only while that test is on: cam->tile_cull_min >= 2, or 0 with a non-zero default).
 * rec: N*20 floats, radii: N int32, rect: N*4 int32 {txlo,ncols,tylo,nrows},
 * tiles_touched: N uint32 = number of tile instances the surfel emits,
 * tile_mask: N uint64 — bit k set: the k-th tile of the rectangle (row-major, the emission order) is emitted;
 *   rectangles of fewer than cam->tile_cull_min or more than 64 tiles are not tested and emit every tile (mask =
 *   all ones below the tile count); may be null while the test is off,
 * block_box (optional, may be null): N uint32 — the surfel's support box as two ranges over the image's 8x2 pixel
 *   blocks (one word; images up to 4096 x 128).  Handed to stage 2, it lets the tile sort deliver the sorted list
 *   as (surfel, mask of the tile's sixteen pixel blocks the surfel can reach) pairs, which the forward tile kernel
 *   scans 256 at a time instead of staging every record (its "dense rounds", the kernel sls_mapping_step runs),
 * depth: N floats (range of the centre, the sort key),
 * order: N uint32 = surfel index at each position of the (range, index) order (ALL surfels,
 *        culled ones included at their range; they have tiles_touched = 0 and emit nothing),
 * offsets: N uint32 = inclusive scan of tiles_touched[order[.]],
 * total_out: 1 uint32 on the DEVICE = number of tile instances R.
 * The caller reads total_out (one D2H sync, as the lineage does) to size the
 * stage-2 buffers (sls_mapping_step has no such sync). */
size_t sls_stage1_scratch_bytes(int N);
int sls_forward_stage1(const SlsCamera *cam, int N,
                       const float *means3D, const float *scales, const float *rotations,
                       const float *opacities, const float *col_cs, const float *row_cs,
                       float *rec, int32_t *radii, int32_t *rect, uint32_t *tiles_touched, uint64_t *tile_mask,
                       uint32_t *block_box,
                       float *depth, uint32_t *order, uint32_t *offsets, uint32_t *total_out,
                       void *scratch, size_t scratch_bytes, void *stream);

/* ---- forward, stage 2: binning, stable sort by tile, ranges, per-tile render
 * total_dev: the device word stage 1 wrote (= R).  tile_keys/vals and the _tmp
 * pair: R uint32 each (ping-pong).  keys64_out (optional, R uint64): the 64-bit keys
 * (tile << 32 | depth bits) the list is ordered by (asks for the two-array sort: no pairs then).
 * The sorted list of surfel indices comes back as (*sorted_list, *sorted_stride): entry j is
 * (*sorted_list)[j * *sorted_stride] — stride 1: a plain array (vals or vals_tmp; *sorted_in_tmp says which); stride 2: the
 * (surfel, block mask) pairs inside sort_scratch (block_box given, lists long enough or list_pairs = 1, at most
 * 2048 tiles): sort_scratch then has to stay alive as long as the list is used (the backward, the caller).
 * list_pairs: 0 = pairs where R >= 1500 T (the rule of sls_mapping_step), 1 = whenever possible, 2 = never.
 * tile_keys / tile_keys_tmp are SCRATCH: the sorted tile ids are NOT an output.  Where the direct binning serves the
 * image (<= 512 tiles, D10 off — every size the reference meets) nothing is written to them, nor to vals_tmp, and
 * tiles_touched / offsets / total_dev / tile_mask are not read; a caller that wants the keys the list is ordered by
 * passes keys64_out (or derives them from ranges + list + depth, as splat_loam_amd/rasterizer.py:_list_and_keys does).
 * ranges: T*2 uint32 with T = ceil(W/tw)*ceil(H/th).  allmap: 7*H*W floats;
 * pix_state: H*W float4 {T_final, M1, M2, 0}; pix_contrib: H*W uint2
 * {n_contrib, median_contrib}; tile_consumed: T uint32 (list entries consumed
 * before the tile finished, the R_eff of SURVEY §8d).
 * block_masks (optional, may be null; sls_block_mask_bytes(R, H, W) bytes = 128 bytes per
 * instance of capacity + a header: room for every list entry in each of a tile's 16 pixel
 * blocks, of which only what contributes is written or read): forward -> backward hand-over,
 * per pixel block the compact list of the (list position, surfel) pairs that reached one of
 * its pixels; with it sls_backward walks exactly those, 64 per round, instead of re-deriving
 * them with a box test over the tile's whole list (the name is round 2's, when the hand-over
 * was a bit mask per 64 list entries).  *block_masks_shape (may be null): which pixel-block shape wrote it
 * (0: none, 2: 4x4, 3: 8x2) — hand it to sls_backward together with the buffer. */
size_t sls_sort_scratch_bytes(uint64_t R);
size_t sls_block_mask_bytes(uint64_t R, int H, int W);
int sls_forward_stage2(const SlsCamera *cam, int N, uint64_t R,
                       const float *rec, const int32_t *rect, const uint32_t *tiles_touched,
                       const uint64_t *tile_mask, const uint32_t *block_box,
                       const float *depth, const uint32_t *order, const uint32_t *offsets,
                       const uint32_t *total_dev,
                       uint32_t *tile_keys, uint32_t *vals, uint32_t *tile_keys_tmp, uint32_t *vals_tmp,
                       void *sort_scratch, size_t sort_scratch_bytes, int *sorted_in_tmp,
                       uint64_t *keys64_out, int list_pairs,
                       const uint32_t **sorted_list, int *sorted_stride,
                       uint32_t *ranges, const float *col_cs, const float *row_cs,
                       float *allmap, float *pix_state, uint32_t *pix_contrib,
                       uint32_t *tile_consumed, uint64_t *block_masks, int *block_masks_shape, void *stream);

/* ---- backward ----------------------------------------------------------
 * vals_sorted (+ vals_stride: 1 or 2, as stage 2 returned them)/ranges/rec/pix_* are the forward's buffers
 * (never allmap: the caller may have overwritten it in place).  block_masks + block_masks_shape: the forward's
 * hand-over and the block shape that wrote it; the compact lists are walked only if that shape is the backward
 * kernel's own (otherwise, or with null / 0, the backward culls the tile's list itself).  grec: N*16 floats of
 * scratch (zeroed by the call).  Outputs: dL/dmeans3D (N*3), dL/dscales (N*2),
 * dL/drotations (N*4, w.r.t. the normalised quaternion as passed in),
 * dL/dopacities (N). */
int sls_backward(const SlsCamera *cam, int N, uint64_t R,
                 const float *means3D, const float *scales, const float *rotations,
                 const int32_t *radii, const float *rec,
                 const uint32_t *ranges, const uint32_t *vals_sorted, int vals_stride,
                 const float *col_cs, const float *row_cs,
                 const float *pix_state, const uint32_t *pix_contrib,
                 const float *dL_dallmap, float *grec,
                 float *dL_dmeans3D, float *dL_dscales, float *dL_drotations,
                 float *dL_dopacities, const uint64_t *block_masks, int block_masks_shape, void *stream);

/* The same backward with DETERMINISTIC accumulation of the per-surfel gradient records: integer atomics
 * instead of float atomics (first launch: the largest |contribution| per surfel and field; second launch:
 * every contribution scaled by 2^(39 - unbiased exponent of that maximum), rounded, added as a 64-bit integer:
 * 39 fractional bits below the largest term, |q| < 2^40, 23 bits of headroom for the sum), so
 * that two runs on the same inputs return the same bits.  No `grec` buffer; det_scratch: DEVICE scratch of
 * sls_backward_det_scratch_bytes(N). */
size_t sls_backward_det_scratch_bytes(int N);
int sls_backward_det(const SlsCamera *cam, int N, uint64_t R,
                     const float *means3D, const float *scales, const float *rotations, const int32_t *radii,
                     const float *rec, const uint32_t *ranges, const uint32_t *vals_sorted, int vals_stride,
                     const float *col_cs, const float *row_cs, const float *pix_state, const uint32_t *pix_contrib,
                     const float *dL_dallmap,
                     float *dL_dmeans3D, float *dL_dscales, float *dL_drotations, float *dL_dopacities,
                     const uint64_t *block_masks, int block_masks_shape, void *det_scratch, size_t det_scratch_bytes,
                     void *stream);

/* ---- forward / backward in ONE call each, against a capacity (no host read of R) ---------------------------
 * What `GaussianRasterizer` runs by default (gaussian_renderer/__init__.py:26,40-47; slam/mapper.py:201): the kernels
 * of sls_mapping_step behind the staged interface's contract.  The instance buffers are sized for `R_capacity` (the
 * caller's guess: last call's R with head room); the status block (R, bit 0 of `overflow`: capacity too small, bit 1:
 * the repaired depth order was not exact) reaches `status_mirror` (pinned HOST memory, optional) from the first
 * workgroup of the binning kernel, a few microseconds into it, words 0..6 before word 7: the caller arms words 0 and 7 with a value
 * the device never writes (0xFFFFFFFF) and polls them while the binning and the tile forward run; a non-zero `overflow` means the
 * outputs are void — repeat with more room resp. with reuse_rounds = 0.
 * depth_order (N uint32, caller-kept PER CAMERA, may be null with reuse_rounds = 0): receives the depth order;
 * reuse_rounds 1..4 repairs the order found there (from this camera's previous call) instead of sorting from scratch
 * — verified exact on the device, bit 1 otherwise.  list_pairs as in stage 2.
 * workspace: sls_forward_ws_bytes(N, H, W, R_capacity) bytes, 256-byte aligned, alive until the backward has run;
 * workspace_ready = 0 on its first use (a region of it must start from zero; forward and backward leave it so).
 * want_backward = 0 (render() under no_grad: Mapper.densify, the tracker): no forward -> backward hand-over is written.
 * radii (N int32), allmap (7*H*W floats): tensors of the caller's own.  *sorted_list / *sorted_stride /
 * *block_masks_shape: what sls_backward_ws wants back (pointers into the workspace).
 * Serves what the direct binning serves (<= 512 tiles, images the block boxes describe, D10 off): otherwise
 * SLS_E_UNSUPPORTED, and the staged forward is the way. */
struct SlsMappingStatus;
size_t sls_forward_ws_bytes(int N, int H, int W, uint64_t R_capacity);
int sls_forward_ws(const SlsCamera *cam, int N, const float *means3D, const float *scales, const float *rotations,
                   const float *opacities, const float *col_cs, const float *row_cs, uint64_t R_capacity,
                   uint32_t *depth_order, int reuse_rounds, int list_pairs, int workspace_ready, int want_backward,
                   int32_t *radii, float *allmap, void *workspace, size_t workspace_bytes, struct SlsMappingStatus *status_dev,
                   struct SlsMappingStatus *status_mirror, const uint32_t **sorted_list, int *sorted_stride,
                   int *block_masks_shape, void *stream);
/* Blocks until words 0 and 7 of the pinned status mirror sls_forward_ws was given differ from `sentinel` (the caller
 * arms both with a value the device never writes): a bounded spin on the calling thread — no interpreter lock is held by
 * a ctypes caller meanwhile — then a drain of the stream as a last resort. */
int sls_wait_status_mirror(const void *mirror_host, uint32_t sentinel, void *stream);
/* The backward of that forward: tile backward (it marks the surfels it reaches) + the projection's backward, which
 * reads — and clears — only the marked surfels' gradient records: no 64 N-byte memset per call.  block_order (optional;
 * sls_block_order_bytes(H, W) bytes, zero-initialised, caller-kept PER CAMERA): the tile backward walks its pixel blocks
 * most expensive first in the order this camera's previous backward left there, and leaves the next one.  Float atomics
 * (sls_backward_det on the staged buffers is the bit-reproducible alternative). */
int sls_backward_ws(const SlsCamera *cam, int N, const float *means3D, const float *scales, const float *rotations,
                    const int32_t *radii, const float *col_cs, const float *row_cs, const float *dL_dallmap,
                    uint64_t R_capacity, void *workspace, size_t workspace_bytes, const uint32_t *sorted_list,
                    int sorted_stride, int block_masks_shape, uint32_t *block_order, float *dL_dmeans3D,
                    float *dL_dscales, float *dL_drotations, float *dL_dopacities, void *stream);

/* ---- fused consumer of allmap: render() post-processing + mapper loss ------
 * Computes, from allmap (7*H*W, NOT modified), the three per-pixel loss terms of
 * slam/mapper.py:174-187 on top of the maps of gaussian_renderer/__init__.py:48-93
 * and writes dL/dallmap (7*H*W) for a loss weight of 1.
 *   loss_sums (4 floats, device): [sum_valid |s-gt|, sum_valid (1-<n_hat,n_surf*alpha>),
 *                                  sum_valid BCE(alpha,1), total]
 *   total = sums[0]/(H*W) + lambda_normal*sums[1]/n_valid + lambda_alpha*sums[2]/n_valid
 * gt_depth: H*W floats, valid: H*W uint8 (1 = valid), col_cs_half/row_cs_half: ray
 * tables at offset (-0.5,-0.5) (device), n_valid: number of valid pixels (host). */
size_t sls_consumer_scratch_bytes(int H, int W);
int sls_consumer_fwd_bwd(int H, int W, const float *allmap, const float *gt_depth, const uint8_t *valid,
                         const float *col_cs_half, const float *row_cs_half, float depth_ratio,
                         float lambda_normal, float lambda_alpha, int n_valid, float *loss_sums,
                         float *dL_dallmap, void *scratch, size_t scratch_bytes, void *stream);

/* render()'s post-processing alone — what the reference's NO-GRAD callers of gaussian_renderer.render read
 * (gaussian_renderer/__init__.py:48-93 + utils/graphic_utils.py:26-88; callers: slam/mapper.py:52-54,
 * slam/tracker.py:173-175, slam/slam.py:81-82, scene/postprocessing.py:162): one launch instead of ~30 torch kernels.
 *   view_rot9: HOST pointer, world_view_transform[:3,:3] row-major (as a matrix: view frame -> world);
 *   out: rend_normal (3*H*W, world frame), surf_depth (H*W), surf_normal (3*H*W, world frame, x alpha, zero border);
 *   rend_alpha / rend_dist are planes 1 / 6 of allmap themselves. */
int sls_render_maps(int H, int W, const float *allmap, const float *view_rot9, const float *col_cs_half,
                    const float *row_cs_half, float depth_ratio, float *rend_normal, float *surf_depth,
                    float *surf_normal, void *stream);

/* Mapper.densify's candidates and sampling weights (slam/mapper.py:51-102 with densify_threshold_egeom <= 0;
 * utils/graphic_utils.py:91-106): weights_out[H*W] = |central differences of log(depth)| at the valid pixels rendered
 * with alpha <= threshold_opacity (rend_alpha = NULL: at every valid pixel — a model's first keyframe), 0 elsewhere;
 * stats_out (4 words, device): [#candidates, float bits of the gradient's maximum over the image, float sum of the
 * weights, 0].  valid: uint8, 1 = valid. */
int sls_densify_weights(int H, int W, const float *image_depth, const uint8_t *valid, const float *rend_alpha,
                        float threshold_opacity, float *weights_out, uint32_t *stats_out, void *stream);

/* Mapper.densify's new rows (slam/mapper.py:104-137) for n drawn pixels (row-major indices, int64 as torch's nonzero
 * leaves them): centres = the measured points in the model frame, rotations = unit quaternions (w,x,y,z; w >= 0) whose
 * third axis is the measured normal (utils/general_utils.py:85-187).  All pointers DEVICE; cam_to_model16 =
 * inv(world_view_transform^T), model_T_frame16 = the keyframe's pose in the model, both row-major 4x4.  Scales (3-NN,
 * sls_knn_dist2) and opacity (0.9) are the caller's. */
int sls_densify_rows(int n, int H, int W, const int64_t *pixels, const float *image_depth, const float *image_normal,
                     const float *col_cs_half, const float *row_cs_half, const float *cam_to_model16,
                     const float *model_T_frame16, float *xyz_out, float *quat_out, void *stream);

/* ---- one whole mapping iteration, enqueued without any host sync -----------
 * slam/mapper.py:150-204 for one keyframe: activations (scene/gaussian_model.py:
 * 39-44) -> rasterizer forward -> sls_consumer_fwd_bwd -> rasterizer backward ->
 * activation backward + scale regulariser (slam/mapper.py:190-195) -> Adam.
 * Parameters are the RAW (pre-activation) tensors of the model.  grads /
 * exp_avg / exp_avg_sq are flat 10*N buckets laid out [xyz 3N | opacity N |
 * scaling 2N | rotation 4N] (the optimiser's group order) — one contiguous
 * buffer, i.e. one RCCL all-reduce in the keyframe-parallel mode.
 * The instance count R stays on the device: buffers are sized for R_capacity;
 * if R exceeds it the excess is dropped, status.overflow is set and the Adam
 * update is skipped (the caller grows the capacity and repeats the iteration).
 * status_dev lives in DEVICE memory; read it after the stream has drained.
 * *allmap_out (optional) receives the address of allmap inside the workspace; with
 * depth_ratio == 0 the loss neither reads nor differentiates planes 5 (median) and 6
 * (distortion) — gaussian_renderer/__init__.py:79-86 — and this call leaves them zero
 * (sls_forward_stage2 always fills all seven). */
struct SlsMappingStatus;
typedef struct SlsMappingConfig {
    float lambda_alpha, lambda_normal, scaling_max, scaling_max_penalty, depth_ratio;
    float lr_xyz, lr_opacity, lr_scaling, lr_rotation;
    int32_t apply_adam;   /* 0: gradients only (all-reduce them, then sls_adam_step) */
    int32_t reuse_depth_order; /* 0: sort from scratch.  2 (up to 4): as 1 with that many repair rounds, each of which
                                * lets a surfel travel one more window of 1024 positions (one more launch, +9 us, per extra round).
                                * 1: the workspace still holds the depth order of the previous iteration on the
                                * SAME keyframe and surfel set: repair it (windowed re-sort + verification)
                                * instead of sorting from scratch.  If the repair does not reach the exact
                                * order, bit 1 of status.overflow is set, Adam is skipped, repeat with 0. */
    int32_t keep_grads;   /* apply_adam = 1 only: 1 = also write the gradients to `grads` (with 0 they are
                           * consumed where they are produced and `grads` is left untouched) */
    int32_t workspace_ready; /* 0 on the first call with a (new) workspace, 1 afterwards: the call keeps the
                              * accumulation buffers inside the workspace zeroed for its successor */
    double beta1, beta2, eps;
    uint32_t *depth_order;   /* optional DEVICE buffer of N uint32: where this keyframe's depth order is kept
                              * (read with reuse_depth_order >= 1, always written); null: inside the workspace */
    struct SlsMappingStatus *status_mirror; /* optional, HOST-visible (pinned, device-mapped) memory: the last
                              * kernel of the iteration copies *status_dev there, so the caller can read the
                              * status after an event/stream wait without enqueuing a device->host copy */
    float *void_flags_out;   /* optional DEVICE pointer to 2 floats (keyframe-parallel mode: the two words after the
                              * gradient bucket): [0] = 1.0 if bit 0 of status.overflow is set, [1] = 1.0 if any
                              * other bit is, else 0.0 — summed over ranks by the gradient all-reduce and then
                              * handed to sls_adam_step_reduced */
    uint32_t grad_chunk;     /* 0: `grads` is the flat bucket [xyz 3N | opacity N | scaling 2N | rotation 4N].
                              * C > 0 (a multiple of 4, N even, apply_adam = 0): reduce-scatter layout for
                              * `grad_ranks` ranks — flat element e lives at e + 4 * (e / C), i.e. rank g's chunk of C
                              * elements is followed by 4 words whose first two receive the void flags (as
                              * void_flags_out, which is ignored then): after a SUM reduce-scatter every rank finds the
                              * group's verdict behind its own chunk.  `grads` holds grad_ranks * (C + 4) floats. */
    uint32_t grad_ranks;
    int32_t deterministic;   /* 1: the backward tile kernel accumulates the gradient records with integer atomics
                              * (two launches: per-field maximum, then fixed-point sum scaled by it): bit-identical
                              * gradients from run to run, about one extra tile-backward per iteration.  0: float
                              * atomics, whose order — and so the last bits of the sums — changes between runs.
                              * 2: the same in ONE launch — every (surfel, field)'s scale is predicted from its sum in
                              * the keyframe's previous iteration (`det_prev`, needed), or from the field's default where
                              * there is no history (defaults are set by iterations run with 1 on the same workspace:
                              * run the first one with 1).  A prediction off by more than 2^22 sets bit 3 of
                              * status.overflow: the iteration is void, repeat it with 1.  Needs the default 8x2 tile
                              * kernels.  Deterministic: the scales depend on earlier (deterministic) results only. */
    int32_t block_masks;     /* the forward tile kernel's dense rounds (tile sort delivers (surfel, block mask) pairs,
                              * DESIGN.md section 4): 0 = where the lists are long enough to pay (capacity >= 1500
                              * instances per tile), 1 = always, 2 = never.  Same results either way. */
    uint64_t *grad_bitmap;   /* optional DEVICE buffer of sls_grad_bitmap_words(N) uint64 (apply_adam = 0, flat `grads`):
                              * bit i of word i / 64 is set iff surfel i has a non-zero gradient in this iteration
                              * (the keyframe reached it, or the scale regulariser pushes on it); the two words behind
                              * the bitmap receive the void flags (non-zero: bit 0 resp. any other bit of
                              * status.overflow) — OR-reduced over the ranks they are the group's verdict
                              * (sls_grad_compact / sls_adam_step_sparse) */
    uint8_t *det_prev;       /* optional DEVICE buffer of 16 N bytes, zero-initialised by the caller, one per keyframe
                              * (deterministic = 1 or 2): the predicted scale of every (surfel, field), rewritten by each
                              * iteration that is not void */
    int32_t phase;           /* 0: the whole iteration.  1: up to and including the tile backward; 2: the rest (backward
                              * of the projection [+ Adam]) of the iteration phase 1 started — same arguments, same
                              * workspace.  Between the two the caller can start a collective that needs nothing of
                              * phase 2: with grad_bitmap, phase 1 ends by writing the bitmap EARLY — the surfels the
                              * tile backward reached or the scale regulariser may push on, a superset of the non-zero
                              * gradients (zero rows change no sum) — so that the ranks' bitmaps can be all-gathered
                              * while phase 2 runs (MappingEngine.overlap) */
    int32_t reserved;
    uint32_t *block_order;   /* optional DEVICE buffer of sls_block_order_bytes(H, W) bytes, zero-initialised by the caller, one
                              * per keyframe: the launch order of the keyframe's tile backward (most expensive pixel blocks
                              * first), written by every iteration for the keyframe's next one.  With it the loss stage
                              * has no launch of its own: the tile backward computes the per-pixel loss terms and their
                              * gradient itself (default 8x2 kernel, depth_ratio = 0; otherwise the buffer is ignored).
                              * Same gradients bit for bit; loss_sums are added up in another order (last bits) */
    const uint64_t *union_bitmap;  /* phase 2 of the touched-set exchange (below; apply_adam = 1, N even): the OR of the ranks'
                              * bitmaps (sls_grad_union's output).  The backward of the projection then WRITES the gradient
                              * row of every surfel of the union into its slot of `grad_compact` (no flat bucket, no packing
                              * launch) and leaves its parameters alone; every other surfel — zero gradient on every rank —
                              * gets its Adam update right there, as on one GPU.  After the SUM of the rows,
                              * sls_adam_step_union updates the union's surfels.  A non-zero gradient outside
                              * the union (a bitmap that was no superset) sets bit 5 of status.overflow. */
    const uint32_t *union_prefix;  /* sls_grad_union's word_prefix */
    float *grad_compact;     /* capacity x 10 floats */
    uint32_t *grad_compact_index;  /* capacity uint32: the surfel of every slot (for sls_adam_step_union) */
    uint32_t grad_compact_capacity;
    uint32_t reserved2;
} SlsMappingConfig;
typedef struct SlsMappingStatus {
    uint32_t R;           /* tile instances of this iteration */
    uint32_t overflow;    /* bit 0: R > R_capacity; bit 1: depth-order repair failed (reuse_depth_order); bit 2: the
                           * sparse exchange's compact buffer was too small (sls_grad_union / sls_grad_compact); bit 3: a predicted scale
                           * of the one-pass deterministic accumulation was off (deterministic = 2: repeat with 1).
                           * Non-zero: results of this iteration are void, Adam was skipped */
    float loss_sums[4];   /* sums of the three pixel terms, pixel-loss total */
    float loss_reg;       /* scale regulariser */
    uint32_t exchange_count;  /* sparse exchange only: number of surfels in the union of the ranks' touched sets */
} SlsMappingStatus;
size_t sls_mapping_workspace_bytes(int N, int H, int W, uint64_t R_capacity);
size_t sls_block_order_bytes(int H, int W);   /* SlsMappingConfig.block_order */
/* The same for ONE configuration: without cfg->deterministic the two fixed-point accumulators (192 B per
 * surfel, about as much as the rest of the per-surfel workspace) are not reserved.  A workspace sized by
 * sls_mapping_workspace_bytes fits every configuration. */
size_t sls_mapping_workspace_bytes_cfg(int N, int H, int W, uint64_t R_capacity, const struct SlsMappingConfig *cfg);
int sls_mapping_step(const SlsCamera *cam, int N,
                     float *xyz, float *scaling_raw, float *rotation_raw, float *opacity_raw,
                     float *grads, float *exp_avg, float *exp_avg_sq, int64_t adam_step,
                     const float *gt_depth, const uint8_t *valid, int n_valid,
                     const float *col_cs, const float *row_cs,
                     const float *col_cs_half, const float *row_cs_half,
                     const SlsMappingConfig *cfg, uint64_t R_capacity,
                     void *workspace, size_t workspace_bytes,
                     SlsMappingStatus *status_dev, float **allmap_out, void *stream);

/* ---- frame-to-keyframe registration on spherical range images -------
 * The job of the reference's `gsaligner` extension (slam/tracker.py:141-197: set_query /
 * set_reference / align(iguess) -> (T, fitness, _)).  That submodule is not vendored, so the
 * algorithm is this repository's own: projective nearest-pixel association
 * under the spherical model of `projmatrix`, point-to-plane + range-image residuals with Huber
 * weights, Gauss-Newton on SE(3) (left perturbation), the 6x6 system solved on the device.
 *   depth: H*W floats (0 / <= depth_min = invalid), points: H*W*3 floats in the frame's own
 *   coordinates (utils/graphic_utils.py:26-66 with transform_in_world=False), normals: H*W*3.
 *   T_host: row-major 4x4 ref_T_query initial guess (HOST); workspace: DEVICE scratch. */
typedef struct SlsAlignerParams {
    int32_t num_iterations;   /* Gauss-Newton iterations */
    int32_t min_inliers;      /* fewer associated pixels: the pose is left as it is */
    float max_distance;       /* association gate |T p - q| (m) */
    float min_cos_angle;      /* gate on cos(angle between reference normal and viewing ray) */
    float huber_delta;        /* point-to-plane residual (m) */
    float range_weight;       /* weight of the range-image term (0: off) */
    float range_huber;        /* (m) */
    float depth_min, depth_max;
    float damping;            /* added to the diagonal of the normal equations */
} SlsAlignerParams;
typedef struct SlsAlignerResult {
    float pose[12];           /* row-major 3x4 [R|t] of ref_T_query */
    float fitness;            /* associated / valid query pixels at the final pose */
    float chi2;               /* weighted squared error at the final pose */
    float last_step;          /* |xi| of the last update (-1: system was not positive definite) */
    int32_t inliers, valid_query, iterations;
} SlsAlignerResult;
size_t sls_aligner_workspace_bytes(void);
int sls_aligner_normals(const SlsCamera *cam, const float *depth, const float *points, float depth_min,
                        float *normals, void *stream);
/* one linearisation at T_host; sys_out (DEVICE, 32 doubles): H upper triangle (21) | b (6) |
 * chi2 | inliers | valid query pixels | 0 0   (tests compare it with the checker) */
int sls_aligner_linearize(const SlsCamera *cam, const SlsAlignerParams *prm, const float *ref_depth,
                          const float *ref_points, const float *ref_normals, const float *query_depth,
                          const float *query_points, const float *T_host, void *workspace, double *sys_out,
                          void *stream);
/* num_iterations iterations + one evaluation, enqueued without a host sync; result_dev: DEVICE */
int sls_aligner_align(const SlsCamera *cam, const SlsAlignerParams *prm, const float *ref_depth,
                      const float *ref_points, const float *ref_normals, const float *query_depth,
                      const float *query_points, const float *T_host, void *workspace,
                      SlsAlignerResult *result_dev, void *stream);

/* ---- spherical projector (scene/preprocessing.py:42-64, the job of `pyprojections`) ----------
 * A LiDAR scan (n,3) f32 -> the images a keyframe is built from.  Convention (pinned by
 * utils/graphic_utils.py:41-59): pixel (c, r) holds the directions with floor(fx*az + cx + 1) = c,
 * floor(fy*el + cy + 1) = r; a 360-degree image wraps in azimuth; the nearest return wins a pixel.
 * scratch: sls_projector_scratch_bytes(H, W), 8-byte aligned, initialised ONCE by sls_projector_prepare;
 * every later call leaves it ready for the next scan.  Nothing synchronises with the host. */
size_t sls_projector_scratch_bytes(int H, int W);
int sls_projector_prepare(int H, int W, void *scratch, size_t scratch_bytes, void *stream);
/* pyp.calculate_spherical_intrinsics (scene/preprocessing.py:42-44): intrinsics_out = 12 DEVICE floats:
 * K row-major (9), vfov, hfov, 1.0 if the cloud had any point off the origin. */
int sls_projector_intrinsics(int n, const float *cloud, int H, int W, float full_azimuth_threshold_deg,
                             float *intrinsics_out, void *scratch, size_t scratch_bytes, void *stream);
/* pyp.Camera(...).project + the image gathering of scene/preprocessing.py:45-64.  K_dev: 9 DEVICE floats.
 * lut (H*W int32, -1 = empty; may be null), range_image (H*W; 0 where empty), normals_image (H*W*3,
 * -p/|p| , scene/preprocessing.py:112; may be null), valid (H*W bytes).  Points with range outside
 * (depth_min, depth_max] are dropped. */
int sls_projector_project(int n, const float *cloud, const float *K_dev, int H, int W, float depth_min,
                          float depth_max, int32_t *lut, float *range_image, float *normals_image,
                          uint8_t *valid, void *scratch, size_t scratch_bytes, void *stream);

/* ---- fused Adam over up to 8 parameter tensors in one launch ------------
 * torch.optim.Adam semantics (no weight decay, no amsgrad); step is 1-based
 * and shared by all groups. */
typedef struct SlsAdamGroup {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t numel;
    float lr;
    float pad;
} SlsAdamGroup;
int sls_adam_step(const SlsAdamGroup *groups_host, int ngroups, double beta1, double beta2,
                  double eps, int64_t step, void *stream);
/* Same, but the update is skipped on the device when *skip_flag_dev != 0 (the
 * overflow word of SlsMappingStatus, possibly OR-reduced over ranks). */
int sls_adam_step_guarded(const SlsAdamGroup *groups_host, int ngroups, double beta1, double beta2,
                          double eps, int64_t step, const uint32_t *skip_flag_dev, void *stream);
/* Keyframe-parallel form: void_flags_dev points at the two floats SlsMappingConfig.void_flags_out
 * produced, AFTER the SUM all-reduce.  The update is skipped if either is > 0 (some rank voided the
 * iteration), and the reduced bits (bit 0 / bit 1) are stored to status_dev->overflow (status_dev may be
 * null) so that the caller's status read sees the group's verdict; with status_mirror (HOST-visible, as in
 * SlsMappingConfig) the status block is also copied there by this, the iteration's last, kernel. */
int sls_adam_step_reduced(const SlsAdamGroup *groups_host, int ngroups, double beta1, double beta2,
                          double eps, int64_t step, const float *void_flags_dev,
                          struct SlsMappingStatus *status_dev, struct SlsMappingStatus *status_mirror,
                          void *stream);

/* ---- keyframe-parallel exchange of the touched set only -----------------------
 * A keyframe reaches ~10 % of a local model's surfels; only they (and the few the scale regulariser pushes on)
 * carry a non-zero gradient.  Instead of all-reducing the dense 40 B x N bucket:
 *   sls_mapping_step(apply_adam = 0, cfg->grad_bitmap = B)      B: sls_grad_bitmap_words(N) uint64
 *   all-gather(B) -> G bitmaps                                   G x N / 8 bytes (RCCL has no bitwise-OR reduction)
 *   sls_grad_compact(N, bitmaps, G, U, grads, compact, capacity, prefix, status)
 *        U (sls_grad_bitmap_words(N) uint64, out) = OR of the G bitmaps: the union of the touched sets + the verdict;
 *        compact[slot][10] = the gradient of the slot-th surfel of the union, [xyz 3 | opacity | scaling 2 |
 *        rotation 4]; status->exchange_count = K = size of the union; K > capacity sets bit 2 of
 *        status->overflow (void: repeat with more room); status->overflow = the group's verdict
 *   all-reduce(compact[0 .. 10 * K_send), SUM)                   K_send <= capacity chosen by the host
 *        (grads_flat / compact null: the union and its prefix only — sls_grad_union)
 *   sls_adam_step_sparse(...)                                    Adam on every surfel, gradient from its slot or 0;
 *        skipped when the iteration is void; copies the status block to status_mirror (HOST-visible) if given.
 *        `part`: 0 = every surfel in one launch; 1 = only the surfels OUTSIDE the union (their gradient is zero on
 *        every rank: their update needs nothing from the collective, so this launch can run while the rows are being
 *        reduced); 2 = only the union's surfels (after the reduction; mirrors the status) — 1 then 2 give the bits of 0.
 * word_prefix: DEVICE scratch of (N + 63) / 64 uint32. */
/* The form MappingEngine runs (dp_mode "sparse"): the rows never exist as a flat bucket.
 *   sls_mapping_step(apply_adam = 0, grad_bitmap = B, phase = 1)  up to the tile backward; B = the surfels it reached or
 *                                                                 the regulariser may push on (a superset) + the verdict
 *   all-gather(B);  sls_grad_union(N, bitmaps, G, U, capacity, prefix, status)      U, prefix, K, the group's verdict
 *   sls_mapping_step(apply_adam = 1, phase = 2, union_bitmap = U, union_prefix = prefix, grad_compact = compact,
 *                    grad_compact_capacity = capacity)            rows of the union -> their slots; Adam everywhere else
 *   all-reduce(compact[0 .. 10 * K_send), SUM);  sls_adam_step_union(...)           Adam on the union; mirrors the status */
int sls_grad_union(int N, const uint64_t *bitmaps, int n_bitmaps, uint64_t *union_bitmap, uint32_t capacity,
                   uint32_t *word_prefix, struct SlsMappingStatus *status_dev, void *stream);
/* Adam on the union's surfels only, a thread per slot (compact_index[slot] = the surfel; status->exchange_count slots
 * are live); skipped when the iteration is void; copies the status block to status_mirror (HOST-visible) if given. */
int sls_adam_step_union(int N, float *xyz, float *opacity_raw, float *scaling_raw, float *rotation_raw,
                        const uint32_t *compact_index, const float *compact_reduced, uint32_t capacity, float *exp_avg,
                        float *exp_avg_sq, float lr_xyz, float lr_opacity, float lr_scaling, float lr_rotation, double beta1,
                        double beta2, double eps, int64_t step, struct SlsMappingStatus *status_dev,
                        struct SlsMappingStatus *status_mirror, void *stream);
size_t sls_grad_bitmap_words(int N);
int sls_grad_compact(int N, const uint64_t *bitmaps, int n_bitmaps, uint64_t *union_bitmap, const float *grads_flat,
                     float *compact, uint32_t capacity, uint32_t *word_prefix, struct SlsMappingStatus *status_dev,
                     void *stream);
int sls_adam_step_sparse(int N, float *xyz, float *opacity_raw, float *scaling_raw, float *rotation_raw,
                         const uint64_t *union_bitmap, const uint32_t *word_prefix, const float *compact_reduced,
                         float *exp_avg, float *exp_avg_sq, float lr_xyz, float lr_opacity, float lr_scaling,
                         float lr_rotation, double beta1, double beta2, double eps, int64_t step, int part,
                         struct SlsMappingStatus *status_dev, struct SlsMappingStatus *status_mirror, void *stream);

/* ---- simple-knn ---------------------------------------------------------
 * out[i] = mean of squared distances from point i to its 3 nearest other
 * points. */
size_t sls_knn_scratch_bytes(int M);
int sls_knn_dist2(int M, const float *xyz, float *out, void *scratch, size_t scratch_bytes,
                  void *stream);
/* The same for the FIRST M_first points only (out: M_first floats), neighbours searched among all M: what Mapper.densify
 * keeps of `distCUDA2(cat(new, existing))` (slam/mapper.py:109-117 reads `[:n_new]`).  Same values as sls_knn_dist2's
 * first M_first; the search runs only for the waves of the sorted curve that hold a query. */
int sls_knn_dist2_first(int M, int M_first, const float *xyz, float *out, void *scratch, size_t scratch_bytes,
                        void *stream);

/* visible[i] = 1 if surfel centre i survives the near cut (radii would be >0
 * unless it is off-image). */
int sls_mark_visible(const SlsCamera *cam, int N, const float *means3D, uint8_t *visible,
                     void *stream);

/* Optional per-kernel timing with HIP events recorded on the launch stream
 * (bench.py's roofline figures).  sls_timing_collect synchronises on the
 * recorded events, fills total_ms[slot] / counts[slot] for slot <
 * sls_timing_slots() and resets the recorder; returns 1 if the event pool ran
 * out (later launches untimed). */
int sls_timing_slots(void);
const char *sls_timing_name(int slot);
int sls_timing_enable(int mode);   /* 0 off, 1 every launch, 2 the two tile kernels only, 3 render_bwd only,
                                    * 4 every 8th render_bwd launch (a timed region that keeps its own clock) */
int sls_timing_collect(double *total_ms_host, int64_t *counts_host);

/* Diagnostic: when non-null, the tile kernels write the shader-clock cycles each
 * wave spent (T * waves_per_tile uint32 each, DEVICE pointers) — used to look at
 * load balance across tiles.  Pass nulls to switch it off (the default). */
int sls_debug_wave_cycles(uint32_t *fwd_cycles, uint32_t *bwd_cycles);

/* Tuning/diagnostic: choose the tile kernels' pixel-block shape: 2 = 4x4, 3 = 8x2 (the default);
 * negative = keep.  Both produce the same results; the tests run both.  Like sls_timing_* and
 * sls_debug_wave_cycles this is a PROCESS-wide diagnostic switch (torch runs backward nodes on its autograd
 * device thread, which must see what the calling thread chose); the data path itself keeps no mutable state. */
int sls_debug_variant(int fwd_variant, int bwd_variant);

/* Device self-test of the wave64 primitives (DPP reduction, ballot ranking).
 * Returns 0 if they behave as the kernels assume.  Synchronises. */
int sls_selftest(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SLS_ABI_H */
