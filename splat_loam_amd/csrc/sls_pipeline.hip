// sls_pipeline.hip — host-side orchestration behind the C-ABI: the two-stage
// forward and the backward of the drop-in rasterizer interface, and
// sls_mapping_step, which enqueues one WHOLE mapping iteration
// (slam/mapper.py:150-204: render -> loss -> backward -> Adam) on a stream with
// no device->host sync and no torch op in between.
#include <stdlib.h>
#include <string.h>

#include "sls_consumer_dev.hpp"
#include "sls_bin.hpp"

namespace sls {

// launchers implemented in the other translation units
int launch_preprocess_fwd(const DevCam &cam, int raw, float smax, float pen, float *reg_out, int N,
                          const float *means, const float *scales, const float *rots, const float *opac, float *rec,
                          int32_t *radii, int32_t *rect, uint32_t *tiles, float *depth, uint32_t *order_keys,
                          uint32_t *order_vals, uint32_t *n_dev, hipStream_t st, uint32_t *status_clear = nullptr,
                          const float *col_cs = nullptr, const float *row_cs = nullptr, uint64_t *tile_mask = nullptr,
                          int32_t *erec = nullptr, const uint32_t *resort_prev_order = nullptr,
                          uint64_t *resort_comp = nullptr, uint32_t *sbox = nullptr, int erec_box = 0,
                          uint32_t *zero_words = nullptr, int n_zero_words = 0);
uint64_t *resort_comp_buffer(int N, void *scratch);
void depth_order_key_buffers(int N, void *scratch, uint32_t *order, uint32_t **keys, uint32_t **vals0,
                             uint32_t **n_dev);
int launch_preprocess_bwd(const DevCam &cam, int raw, float smax, float pen, int N, const float *means,
                          const float *scales, const float *rots, const float *opac, const int32_t *radii,
                          const float *grec, float *dmeans, float *dscales, float *drots, float *dopac,
                          hipStream_t st, const AdamFuse *fuse = nullptr);
size_t sort_scratch_bytes(uint64_t cap);
size_t order_scratch_bytes(int N);
int launch_bin_sort(const DevCam &cam, int N, const uint32_t *count_ptr, uint32_t cap, const uint32_t *order,
                    const int32_t *rect, const uint32_t *tiles, const uint64_t *tile_mask, const int32_t *erec,
                    const float *depth, const uint32_t *offsets,
                    uint32_t *tkeys, uint32_t *vals, uint32_t *tkeys_tmp, uint32_t *vals_tmp, void *scratch,
                    size_t scratch_bytes, int *sorted_in_tmp, uint32_t *ranges, uint64_t *keys64_out,
                    uint32_t *overflow, hipStream_t st, const ScanHandoff *handoff = nullptr,
                    uint32_t *total_out = nullptr, const uint32_t *sbox = nullptr, const uint2 **bmask_out = nullptr,
                    int bmask_mode = 0);
int launch_render_fwd(const DevCam &, const uint32_t *, const uint32_t *, const float *, const float *,
                      const float *, float *, float *, uint32_t *, uint32_t *, hipStream_t, bool consumed_zeroed = false,
                      uint64_t *block_masks = nullptr, bool no_median_dist = false, uint32_t *block_cost = nullptr,
                      const uint2 *bmask = nullptr, bool order_in_handover = false);
int launch_render_bwd(const DevCam &, const uint32_t *, const uint32_t *, const float *, const float *,
                      const float *, const float *, const uint32_t *, const float *, float *, hipStream_t,
                      const uint64_t *block_masks = nullptr, bool no_median_dist_grad = false,
                      uint8_t *touched = nullptr, const struct ConsumerArgs *fused_consumer = nullptr,
                      uint32_t *det_max = nullptr, unsigned long long *det_acc = nullptr,
                      const uint32_t *block_order = nullptr, int vals_stride = 1, int block_masks_shape = -1,
                      bool order_in_handover = false, const uint8_t *det_prev = nullptr, const uint32_t *det_gex = nullptr,
                      uint32_t *det_flag = nullptr, bool consumer_b_inline = false, uint32_t order_tag = 0u);
size_t block_mask_bytes(uint64_t cap, int T);
size_t consumer_scratch_bytes(int H, int W);
int launch_consumer(int H, int W, const float *allmap, const float *gt_depth, const uint8_t *valid,
                    const float *col_h, const float *row_h, float depth_ratio, float lambda_n, float lambda_a,
                    int n_valid, float *sums, float *dL_dallmap, void *scratch, size_t scratch_bytes,
                    hipStream_t st, bool sums_zeroed = false, struct ConsumerArgs *args_out_skip_c = nullptr,
                    int order_tiles = 0, const uint32_t *block_cost = nullptr, uint32_t *block_order = nullptr,
                    bool no_launch = false);
int launch_touched_bitmap(int N, const uint8_t *touched, const float *scaling_raw, float smax, float pen,
                          const uint32_t *status_block, uint64_t *bitmap, hipStream_t st);
int launch_adam(const SlsAdamGroup *groups, int ngroups, double beta1, double beta2, double eps, int64_t step,
                const uint32_t *skip_flag, hipStream_t stream, const float *void_flags = nullptr,
                uint32_t *status_block = nullptr, uint32_t *status_mirror = nullptr);

// ---------------------------------------------------------------------------
// workspace of sls_mapping_step: one caller-owned buffer, carved here
// ---------------------------------------------------------------------------
struct MapWs {
    float *rec; int32_t *radii; int32_t *rect; uint32_t *tiles; uint64_t *tmask; int32_t *erec; int32_t *serec; uint32_t *sbox; float *depth; uint32_t *order; uint32_t *offsets;
    void *order_scratch; size_t order_scratch_bytes;
    uint32_t *tkeys, *vals, *tkeys_tmp, *vals_tmp; void *sort_scratch; size_t sort_scratch_bytes;
    uint32_t *ranges; float *allmap; float *pix_state; uint32_t *pix_contrib; uint32_t *tile_consumed;
    float *dL_dallmap; void *consumer_scratch; size_t consumer_scratch_bytes; float *grec; size_t zero_bytes; uint64_t *block_masks; float *reg_accum; uint8_t *touched;
    uint32_t *det_max; unsigned long long *det_acc; size_t det_bytes;    // deterministic accumulation (zeroed per iteration when used)
    uint32_t *det_gex;                                                   // one-pass variant: the fields' default scales (16 words)
    uint32_t *block_cost, *block_order;                                  // backward blocks: cost from the forward, order (most expensive first)
    size_t total;
};

static MapWs carve(int N, int H, int W, uint64_t cap, void *base, bool deterministic)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    MapWs w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { void *r = (void *)(p + off); off += al(bytes); return r; };
    const size_t n = (size_t)(N > 0 ? N : 1), P = (size_t)H * W, c = (size_t)(cap > 0 ? cap : 1);
    const int GX = (W + kTileW - 1) / kTileW, GY = (H + kTileH - 1) / kTileH;
    const size_t T = (size_t)GX * GY;
    w.rec = (float *)take(n * SLS_REC_STRIDE * 4);
    w.radii = (int32_t *)take(n * 4);
    w.rect = (int32_t *)take(n * 16);
    w.tiles = (uint32_t *)take(n * 4);
    w.tmask = (uint64_t *)take(n * 8);
    w.erec = (int32_t *)take(n * 16);
    w.serec = (int32_t *)take(n * 8);          // direct binning: the emission records by depth position
    w.sbox = (uint32_t *)take(n * 4);          // block box per surfel (sls_common.hpp: make_block_box)
    w.depth = (float *)take(n * 4);
    w.order = (uint32_t *)take(n * 4);
    w.offsets = (uint32_t *)take(n * 4);
    w.order_scratch_bytes = order_scratch_bytes(N);
    w.order_scratch = take(w.order_scratch_bytes);
    w.tkeys = (uint32_t *)take(c * 4);
    w.vals = (uint32_t *)take(c * 4);
    w.tkeys_tmp = (uint32_t *)take(c * 4);
    w.vals_tmp = (uint32_t *)take(c * 4);
    w.sort_scratch_bytes = sort_scratch_bytes(cap);
    w.sort_scratch = take(w.sort_scratch_bytes);
    w.ranges = (uint32_t *)take(T * 8);
    w.allmap = (float *)take(P * 7 * 4);
    w.pix_state = (float *)take(P * 16);
    w.pix_contrib = (uint32_t *)take(P * 8);
    w.dL_dallmap = (float *)take(P * 7 * 4);
    w.consumer_scratch_bytes = consumer_scratch_bytes(H, W);
    w.consumer_scratch = take(w.consumer_scratch_bytes);
    w.block_masks = (uint64_t *)take(block_mask_bytes(cap, (int)T));
    // zeroed together on the first use of a workspace: [reg_accum | tile_consumed | touched | grec]
    w.reg_accum = (float *)take(4);
    w.det_gex = (uint32_t *)take(16 * 4);
    w.tile_consumed = (uint32_t *)take(T * 4);
    w.touched = (uint8_t *)take(n);
    w.grec = (float *)take(n * SLS_GREC_STRIDE * 4);
    w.zero_bytes = (size_t)((char *)w.grec - (char *)w.reg_accum) + n * SLS_GREC_STRIDE * 4;
    w.block_cost = (uint32_t *)take(T * (kTilePix / 16) * 4);
    w.block_order = (uint32_t *)take(T * (kTilePix / 16) * 4);
    // deterministic accumulation only (192 B per surfel: as much again as everything per-surfel above)
    w.det_max = nullptr; w.det_acc = nullptr; w.det_bytes = 0;
    if (deterministic) {
        w.det_max = (uint32_t *)take(n * SLS_GREC_STRIDE * 4);
        w.det_acc = (unsigned long long *)take(n * SLS_GREC_STRIDE * 8);
        w.det_bytes = (size_t)((char *)w.det_acc - (char *)w.det_max) + n * SLS_GREC_STRIDE * 8;
    }
    w.total = off;
    return w;
}

}  // namespace sls

using namespace sls;

extern "C" {

size_t sls_stage1_scratch_bytes(int N) { return order_scratch_bytes(N); }

int sls_forward_stage1(const SlsCamera *cam, int N, const float *means3D, const float *scales,
                       const float *rotations, const float *opacities, const float *col_cs, const float *row_cs,
                       float *rec, int32_t *radii, int32_t *rect,
                       uint32_t *tiles_touched, uint64_t *tile_mask, uint32_t *block_box, float *depth, uint32_t *order,
                       uint32_t *offsets, uint32_t *total_out, void *scratch, size_t scratch_bytes, void *stream)
{
    SLS_REQUIRE(cam && total_out, "null pointer");
    SLS_REQUIRE(N >= 0, "negative N");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        SLS_HIP_CHECK(hipMemsetAsync(total_out, 0, sizeof(uint32_t), st));
        return SLS_OK;
    }
    SLS_REQUIRE(means3D && scales && rotations && opacities && rec && radii && rect && tiles_touched && depth &&
                    order && offsets && scratch,
                "null pointer");
    const DevCam dc = make_devcam(*cam);
    // (the ray tables and the mask buffer are D10's: required only while the tile-level footprint test is on)
    SLS_REQUIRE(dc.tile_cull == 0 || (col_cs && row_cs && tile_mask),
                "the tile-level footprint test (SlsCamera.tile_cull_min >= 2) needs the ray tables and a tile_mask buffer");
    if (scratch_bytes < order_scratch_bytes(N)) {
        set_error("stage1 scratch too small");
        return SLS_E_SCRATCH;
    }
    // without the tables the preprocess writes no masks: every tile of every rectangle is emitted
    if (tile_mask && !(col_cs && row_cs)) SLS_HIP_CHECK(hipMemsetAsync(tile_mask, 0xFF, sizeof(uint64_t) * (size_t)N, st));
    uint32_t *okeys, *ovals, *n_dev;
    depth_order_key_buffers(N, scratch, order, &okeys, &ovals, &n_dev);
    int rc = launch_preprocess_fwd(dc, 0, 0.0f, 0.0f, nullptr, N, means3D, scales, rotations, opacities, rec, radii,
                                   rect, tiles_touched, depth, okeys, ovals, n_dev, st, nullptr, col_cs, row_cs, tile_mask,
                                   nullptr, nullptr, nullptr, block_box);
    if (rc) return rc;
    return launch_depth_order_scan(N, depth, tiles_touched, order, offsets, total_out, scratch, scratch_bytes, 1, st);
}

size_t sls_sort_scratch_bytes(uint64_t R) { return sort_scratch_bytes(R); }
size_t sls_block_mask_bytes(uint64_t R, int H, int W)
{
    return block_mask_bytes(R, ((W + kTileW - 1) / kTileW) * ((H + kTileH - 1) / kTileH));
}

int sls_forward_stage2(const SlsCamera *cam, int N, uint64_t R, const float *rec, const int32_t *rect,
                       const uint32_t *tiles_touched, const uint64_t *tile_mask, const uint32_t *block_box,
                       const float *depth, const uint32_t *order,
                       const uint32_t *offsets, const uint32_t *total_dev, uint32_t *tkeys, uint32_t *vals,
                       uint32_t *tkeys_tmp, uint32_t *vals_tmp, void *sort_scratch, size_t sort_scratch_bytes_,
                       int *sorted_in_tmp, uint64_t *keys64_out, int list_pairs, const uint32_t **sorted_list,
                       int *sorted_stride, uint32_t *ranges, const float *col_cs,
                       const float *row_cs, float *allmap, float *pix_state, uint32_t *pix_contrib,
                       uint32_t *tile_consumed, uint64_t *block_masks, int *block_masks_shape, void *stream)
{
    SLS_REQUIRE(cam && sorted_in_tmp && sorted_list && sorted_stride && ranges && col_cs && row_cs && allmap &&
                    pix_state && pix_contrib,
                "null pointer");
    SLS_REQUIRE(R == 0 || (rec && rect && tiles_touched && depth && order && offsets && total_dev && tkeys && vals &&
                           tkeys_tmp && vals_tmp && sort_scratch),
                "null pointer");
    SLS_REQUIRE(R < (1ull << 32), "more than 2^32 tile instances");
    SLS_REQUIRE(list_pairs >= 0 && list_pairs <= 2, "list_pairs: 0 auto, 1 whenever possible, 2 never");
    hipStream_t st = (hipStream_t)stream;
    const DevCam dc = make_devcam(*cam);
    // With the surfels' block boxes the tile sort delivers (surfel, block mask) pairs in list order and the forward
    // runs its dense rounds — the kernels of sls_mapping_step, under the same long-list rule (R is exact here)
    const uint2 *bmask = nullptr;
    const bool default_kernels = debug_state().fwd_variant == 3;
    const uint32_t *boxes = (default_kernels && !keys64_out) ? block_box : nullptr;
    int rc;
    if (!keys64_out && R > 0 && bin_direct_possible(dc, N, (uint32_t)R)) {
        // direct binning (sls_sort.hip), as in sls_mapping_step; the records are gathered from rect / block_box
        *sorted_in_tmp = 0;
        if (sort_scratch_bytes_ < sort_scratch_bytes(R)) {
            set_error("sort scratch too small: %zu < %zu", sort_scratch_bytes_, sort_scratch_bytes(R));
            return SLS_E_SCRATCH;
        }
        const DirectBin db = make_direct_bin(dc, N, sort_scratch, nullptr, false);
        rc = launch_bin_direct(dc, N, (uint32_t)R, db, false, order, nullptr, rect, boxes, sort_scratch, vals, ranges,
                               nullptr, nullptr, 0, nullptr, &bmask, list_pairs, st);
    } else {
        rc = launch_bin_sort(dc, N, total_dev, (uint32_t)R, order, rect, tiles_touched,
                             dc.tile_cull ? tile_mask : nullptr, nullptr, depth, offsets, tkeys, vals,
                             tkeys_tmp, vals_tmp, sort_scratch, sort_scratch_bytes_, sorted_in_tmp, ranges,
                             keys64_out, nullptr, st, nullptr, nullptr, boxes, &bmask, list_pairs);
    }
    if (rc) return rc;
    const uint32_t *sorted_vals = *sorted_in_tmp ? vals_tmp : vals;
    *sorted_list = bmask ? (const uint32_t *)bmask : sorted_vals;
    *sorted_stride = bmask ? 2 : 1;
    if (block_masks_shape) *block_masks_shape = block_masks ? (int)debug_state().fwd_variant : 0;
    return launch_render_fwd(dc, ranges, sorted_vals, rec, col_cs, row_cs, allmap, pix_state, pix_contrib,
                             tile_consumed, st, false, block_masks, (cam->flags & SLS_CAM_LEAN_ALLMAP) != 0, nullptr, bmask,
                             true);       // (+ the backward's launch order into the hand-over buffer)
}

int sls_backward(const SlsCamera *cam, int N, uint64_t R, const float *means3D, const float *scales,
                 const float *rotations, const int32_t *radii, const float *rec, const uint32_t *ranges,
                 const uint32_t *vals_sorted, int vals_stride, const float *col_cs, const float *row_cs,
                 const float *pix_state,
                 const uint32_t *pix_contrib, const float *dL_dallmap, float *grec, float *dL_dmeans3D,
                 float *dL_dscales, float *dL_drotations, float *dL_dopacities, const uint64_t *block_masks,
                 int block_masks_shape, void *stream)
{
    SLS_REQUIRE(cam, "null pointer");
    SLS_REQUIRE(N >= 0, "negative N");
    if (N == 0) return SLS_OK;
    SLS_REQUIRE(means3D && scales && rotations && radii && grec && dL_dmeans3D && dL_dscales && dL_drotations &&
                    dL_dopacities,
                "null pointer");
    SLS_REQUIRE(vals_stride == 1 || vals_stride == 2, "vals_stride: 1 (plain list) or 2 ((surfel, block mask) pairs)");
    hipStream_t st = (hipStream_t)stream;
    const DevCam dc = make_devcam(*cam);
    {
        ScopedTimer tm(T_GREC_MEMSET, st);
        SLS_HIP_CHECK(hipMemsetAsync(grec, 0, sizeof(float) * (size_t)N * SLS_GREC_STRIDE, st));
    }
    if (R > 0) {
        SLS_REQUIRE(rec && ranges && vals_sorted && col_cs && row_cs && pix_state && pix_contrib && dL_dallmap,
                    "null pointer");
        int rc = launch_render_bwd(dc, ranges, vals_sorted, rec, col_cs, row_cs, pix_state, pix_contrib, dL_dallmap,
                                   grec, st, block_masks_shape ? block_masks : nullptr, (cam->flags & SLS_CAM_LEAN_ALLMAP) != 0,
                                   nullptr, nullptr, nullptr, nullptr, nullptr, vals_stride, block_masks_shape, true);
        if (rc) return rc;
    }
    return launch_preprocess_bwd(dc, 0, 0.0f, 0.0f, N, means3D, scales, rotations, nullptr, radii, grec, dL_dmeans3D,
                                 dL_dscales, dL_drotations, dL_dopacities, st);
}

size_t sls_backward_det_scratch_bytes(int N) { return N > 0 ? (size_t)N * SLS_GREC_STRIDE * 12 + 256 : 256; }

int sls_backward_det(const SlsCamera *cam, int N, uint64_t R, const float *means3D, const float *scales,
                     const float *rotations, const int32_t *radii, const float *rec, const uint32_t *ranges,
                     const uint32_t *vals_sorted, int vals_stride, const float *col_cs, const float *row_cs,
                     const float *pix_state,
                     const uint32_t *pix_contrib, const float *dL_dallmap, float *dL_dmeans3D, float *dL_dscales,
                     float *dL_drotations, float *dL_dopacities, const uint64_t *block_masks, int block_masks_shape,
                     void *det_scratch, size_t det_scratch_bytes, void *stream)
{
    SLS_REQUIRE(cam, "null pointer");
    SLS_REQUIRE(N >= 0, "negative N");
    if (N == 0) return SLS_OK;
    SLS_REQUIRE(means3D && scales && rotations && radii && det_scratch && dL_dmeans3D && dL_dscales && dL_drotations &&
                    dL_dopacities,
                "null pointer");
    SLS_REQUIRE(vals_stride == 1 || vals_stride == 2, "vals_stride: 1 (plain list) or 2 ((surfel, block mask) pairs)");
    if (det_scratch_bytes < sls_backward_det_scratch_bytes(N)) {
        set_error("deterministic-backward scratch too small");
        return SLS_E_SCRATCH;
    }
    hipStream_t st = (hipStream_t)stream;
    const DevCam dc = make_devcam(*cam);
    unsigned long long *acc = (unsigned long long *)(((uintptr_t)det_scratch + 255) & ~(uintptr_t)255);
    uint32_t *mx = (uint32_t *)(acc + (size_t)N * SLS_GREC_STRIDE);
    SLS_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)N * SLS_GREC_STRIDE * 12, st));
    if (R > 0) {
        SLS_REQUIRE(rec && ranges && vals_sorted && col_cs && row_cs && pix_state && pix_contrib && dL_dallmap,
                    "null pointer");
        int rc = launch_render_bwd(dc, ranges, vals_sorted, rec, col_cs, row_cs, pix_state, pix_contrib, dL_dallmap,
                                   nullptr, st, block_masks_shape ? block_masks : nullptr,
                                   (cam->flags & SLS_CAM_LEAN_ALLMAP) != 0, nullptr, nullptr, mx, acc,
                                   nullptr, vals_stride, block_masks_shape, true);
        if (rc) return rc;
    }
    AdamFuse fuse;
    memset(&fuse, 0, sizeof(fuse));
    fuse.det_max = mx;
    fuse.det_acc = (const long long *)acc;
    return launch_preprocess_bwd(dc, 0, 0.0f, 0.0f, N, means3D, scales, rotations, nullptr, radii, nullptr, dL_dmeans3D,
                                 dL_dscales, dL_drotations, dL_dopacities, st, &fuse);
}

// ---- the drop-in forward without the host read of R --------------------------------------------------------------
// (the status block — R, void bits — reaches the caller's pinned host mirror from the first workgroup of bin_direct)
size_t sls_forward_ws_bytes(int N, int H, int W, uint64_t R_capacity)
{
    if (N < 0 || H <= 0 || W <= 0) return 0;
    return carve(N, H, W, R_capacity, nullptr, false).total;
}

int sls_forward_ws(const SlsCamera *cam, int N, const float *means3D, const float *scales, const float *rotations,
                   const float *opacities, const float *col_cs, const float *row_cs, uint64_t R_capacity,
                   uint32_t *depth_order, int reuse_rounds, int list_pairs, int workspace_ready, int want_backward,
                   int32_t *radii, float *allmap, void *workspace, size_t workspace_bytes, SlsMappingStatus *status_dev,
                   SlsMappingStatus *status_mirror, const uint32_t **sorted_list, int *sorted_stride,
                   int *block_masks_shape, void *stream)
{
    SLS_REQUIRE(cam && status_dev && workspace && sorted_list && sorted_stride && block_masks_shape, "null pointer");
    SLS_REQUIRE(N > 0, "N must be positive");
    SLS_REQUIRE(means3D && scales && rotations && opacities && col_cs && row_cs && radii && allmap, "null pointer");
    SLS_REQUIRE(R_capacity > 0 && R_capacity < (1ull << 32), "bad instance capacity");
    SLS_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    SLS_REQUIRE(reuse_rounds >= 0 && reuse_rounds <= 4, "reuse_rounds: 0 from scratch, 1..4 repair rounds");
    SLS_REQUIRE(reuse_rounds == 0 || depth_order, "a repair needs the camera's previous depth order");
    SLS_REQUIRE(list_pairs >= 0 && list_pairs <= 2, "list_pairs: 0 auto, 1 whenever possible, 2 never");
    const int H = cam->H, W = cam->W;
    const DevCam dc = make_devcam(*cam);
    const uint32_t cap = (uint32_t)R_capacity;
    if (!bin_direct_possible(dc, N, cap) || debug_state().fwd_variant != 3 || debug_state().bwd_variant != 3) {
        set_error("sls_forward_ws serves what the direct binning serves (<= 512 tiles, D10 off, default tile kernels): use the staged forward");
        return SLS_E_UNSUPPORTED;
    }
    const MapWs w = carve(N, H, W, R_capacity, workspace, false);
    if (workspace_bytes < w.total) {
        set_error("forward workspace too small: %zu < %zu", workspace_bytes, w.total);
        return SLS_E_SCRATCH;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!workspace_ready) {     // the gradient records and the touched marks start from zero; the backward leaves them so
        ScopedTimer tm(T_GREC_MEMSET, st);
        SLS_HIP_CHECK(hipMemsetAsync(w.reg_accum, 0, w.zero_bytes, st));
    }
    uint32_t *okeys, *ovals, *n_dev;
    uint32_t *order = depth_order ? depth_order : w.order;
    depth_order_key_buffers(N, w.order_scratch, order, &okeys, &ovals, &n_dev);
    constexpr bool no_merge = false;
    const bool merged_sort = reuse_rounds >= 1 && !no_merge;
    DirectBin db = make_direct_bin(dc, N, w.sort_scratch, (uint2 *)w.serec, reuse_rounds >= 1, true);
    // (as in sls_mapping_step: the direct binning reads the emission records only)
    int rc = launch_preprocess_fwd(dc, 0, 0.0f, 0.0f, nullptr, N, means3D, scales, rotations, opacities, w.rec, radii,
                                   nullptr, nullptr, nullptr, merged_sort ? nullptr : okeys, ovals, n_dev, st,
                                   (uint32_t *)status_dev, col_cs, row_cs, w.tmask, w.erec, merged_sort ? order : nullptr,
                                   merged_sort ? resort_comp_buffer(N, w.order_scratch) : nullptr, nullptr, 1, db.coarse,
                                   (int)direct_coarse_words(dc, N));
    if (rc) return rc;
    ScanHandoff handoff = { nullptr, 0, nullptr, 0 };
    rc = launch_depth_order_scan(N, w.depth, w.tiles, order, w.offsets, &status_dev->R, w.order_scratch,
                                 w.order_scratch_bytes, 1, st, reuse_rounds, &status_dev->overflow, &handoff, merged_sort,
                                 &db, (const int4 *)w.erec, dc.GX);
    if (rc) return rc;
    const uint2 *bmask = nullptr;
    rc = launch_bin_direct(dc, N, cap, db, handoff.counted != 0, order, w.erec, nullptr, nullptr, w.sort_scratch, w.vals,
                           w.ranges, &status_dev->R, &status_dev->overflow, handoff.resort_windows, handoff.resort_edges,
                           &bmask, list_pairs, st, (uint32_t *)status_mirror);      // (the status block leaves from its first workgroup)
    if (rc) return rc;
    const uint32_t *list = bmask ? (const uint32_t *)bmask : w.vals;
    *sorted_list = list;
    *sorted_stride = bmask ? 2 : 1;
    // a forward nobody differentiates (render() under no_grad: Mapper.densify, the tracker) writes no hand-over; one
    // that is leaves the backward its blocks' compact lists and their costs (sorted into the camera's launch order by
    // passengers of the backward's last kernel, for the camera's NEXT backward: as sls_mapping_step does)
    *block_masks_shape = want_backward ? (int)debug_state().fwd_variant : 0;
    return launch_render_fwd(dc, w.ranges, list, w.rec, col_cs, row_cs, allmap, w.pix_state, w.pix_contrib, nullptr, st,
                             true, want_backward ? w.block_masks : nullptr, (cam->flags & SLS_CAM_LEAN_ALLMAP) != 0,
                             want_backward ? w.block_cost : nullptr, bmask, false);
}

int sls_backward_ws(const SlsCamera *cam, int N, const float *means3D, const float *scales, const float *rotations,
                    const int32_t *radii, const float *col_cs, const float *row_cs, const float *dL_dallmap,
                    uint64_t R_capacity, void *workspace, size_t workspace_bytes, const uint32_t *sorted_list,
                    int sorted_stride, int block_masks_shape, uint32_t *block_order, float *dL_dmeans3D,
                    float *dL_dscales, float *dL_drotations, float *dL_dopacities, void *stream)
{
    SLS_REQUIRE(cam && workspace && sorted_list, "null pointer");
    SLS_REQUIRE(N > 0, "N must be positive");
    SLS_REQUIRE(means3D && scales && rotations && radii && col_cs && row_cs && dL_dallmap && dL_dmeans3D && dL_dscales &&
                    dL_drotations && dL_dopacities,
                "null pointer");
    SLS_REQUIRE(sorted_stride == 1 || sorted_stride == 2, "sorted_stride: 1 (plain list) or 2 ((surfel, block mask) pairs)");
    const MapWs w = carve(N, cam->H, cam->W, R_capacity, workspace, false);
    if (workspace_bytes < w.total) {
        set_error("forward workspace too small: %zu < %zu", workspace_bytes, w.total);
        return SLS_E_SCRATCH;
    }
    hipStream_t st = (hipStream_t)stream;
    const DevCam dc = make_devcam(*cam);
    // the tile backward marks the surfels it reaches; the projection's backward reads — and clears — only their records:
    // no 64 N-byte memset per call
    // the blocks most expensive first per XCD, in the order this camera's PREVIOUS backward left in the caller's buffer
    // (its tag word says whether one has: a first visit walks the natural order)
    const int T = dc.GX * dc.GY;
    const bool order_bwd = block_order != nullptr && block_masks_shape == 3 && debug_state().bwd_variant == 3 &&
                           T % 32 == 0 && kTileW == 16 && kTileH == 16;
    // (a tile-kernel variant switched between forward and backward — sls_debug_variant — is no lost gradient: the launcher
    //  walks the forward's compact lists only when block_masks_shape names the backward's own shape, otherwise the
    //  backward culls the tiles' lists itself; the kernel's tag check is a last guard against a FOREIGN buffer, which the
    //  workspace lease rules out on this path)
    int rc = launch_render_bwd(dc, w.ranges, sorted_list, w.rec, col_cs, row_cs, w.pix_state, w.pix_contrib, dL_dallmap,
                               w.grec, st, block_masks_shape ? w.block_masks : nullptr,
                               (cam->flags & SLS_CAM_LEAN_ALLMAP) != 0, w.touched, nullptr, nullptr, nullptr,
                               order_bwd ? block_order : nullptr, sorted_stride, block_masks_shape, false, nullptr, nullptr,
                               nullptr, false, order_bwd ? block_order_tag(T) : 0u);
    if (rc) return rc;
    AdamFuse fuse;
    memset(&fuse, 0, sizeof(fuse));
    fuse.clear_grec = 1;
    fuse.touched = w.touched;
    if (order_bwd) { fuse.order_T = T; fuse.order_cost = w.block_cost; fuse.order_out = block_order; }
    return launch_preprocess_bwd(dc, 0, 0.0f, 0.0f, N, means3D, scales, rotations, nullptr, radii, w.grec, dL_dmeans3D,
                                 dL_dscales, dL_drotations, dL_dopacities, st, &fuse);
}

size_t sls_mapping_workspace_bytes(int N, int H, int W, uint64_t R_capacity)
{
    if (N < 0 || H <= 0 || W <= 0) return 0;
    return carve(N, H, W, R_capacity, nullptr, true).total;      // fits either setting of cfg->deterministic
}

size_t sls_block_order_bytes(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    const size_t T = (size_t)((W + kTileW - 1) / kTileW) * (size_t)((H + kTileH - 1) / kTileH);
    return sizeof(uint32_t) * (1 + T * (size_t)(kTilePix / 16));
}

size_t sls_mapping_workspace_bytes_cfg(int N, int H, int W, uint64_t R_capacity, const SlsMappingConfig *cfg)
{
    if (N < 0 || H <= 0 || W <= 0) return 0;
    return carve(N, H, W, R_capacity, nullptr, !cfg || cfg->deterministic != 0).total;
}

int sls_mapping_step(const SlsCamera *cam, int N, float *xyz, float *scaling_raw, float *rotation_raw,
                     float *opacity_raw, float *grads, float *exp_avg, float *exp_avg_sq, int64_t adam_step,
                     const float *gt_depth, const uint8_t *valid, int n_valid, const float *col_cs,
                     const float *row_cs, const float *col_cs_half, const float *row_cs_half,
                     const SlsMappingConfig *cfg, uint64_t R_capacity, void *workspace, size_t workspace_bytes,
                     SlsMappingStatus *status_dev, float **allmap_out, void *stream)
{
    SLS_REQUIRE(cam && cfg && status_dev && workspace, "null pointer");
    SLS_REQUIRE(N > 0, "N must be positive");
    SLS_REQUIRE(xyz && scaling_raw && rotation_raw && opacity_raw && grads && gt_depth && valid && col_cs && row_cs &&
                    col_cs_half && row_cs_half,
                "null pointer");
    SLS_REQUIRE(!cfg->apply_adam || (exp_avg && exp_avg_sq && adam_step >= 1), "Adam state missing");
    SLS_REQUIRE(R_capacity > 0 && R_capacity < (1ull << 32), "bad instance capacity");
    SLS_REQUIRE(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    // (every argument check before the first launch: an argument error must not leave half an iteration on the stream)
    SLS_REQUIRE(cfg->phase >= 0 && cfg->phase <= 2, "phase: 0 whole iteration, 1 up to the tile backward, 2 the rest");
    SLS_REQUIRE(!cfg->grad_bitmap || ((!cfg->apply_adam || cfg->union_bitmap) && !cfg->grad_chunk),
                "the gradient bitmap belongs to apply_adam = 0 with the flat bucket");
    SLS_REQUIRE(!cfg->union_bitmap || (cfg->phase == 2 && cfg->apply_adam && cfg->union_prefix && cfg->grad_compact && cfg->grad_compact_index &&
                                       !cfg->grad_chunk && !cfg->keep_grads && (N % 2) == 0 &&
                                       ((((uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0)),
                "union_bitmap: phase 2 with apply_adam = 1, the union's prefix and row buffer, even N, 16-byte aligned moments");
    SLS_REQUIRE(!cfg->grad_chunk ||
                    (!cfg->apply_adam && cfg->grad_ranks >= 1 && (cfg->grad_chunk % 4u) == 0 && (N % 2) == 0 &&
                     (uint64_t)cfg->grad_chunk * cfg->grad_ranks >= (uint64_t)10 * (uint64_t)N &&
                     (uint64_t)10 * (uint64_t)N < (1ull << 32)),
                "reduce-scatter gradient layout: apply_adam = 0, even N, chunk a multiple of 4 covering 10 N");
    const int H = cam->H, W = cam->W;
    const MapWs w = carve(N, H, W, R_capacity, workspace, cfg->deterministic != 0);
    if (workspace_bytes < w.total) {
        set_error("mapping workspace too small: %zu < %zu", workspace_bytes, w.total);
        return SLS_E_SCRATCH;
    }
    hipStream_t st = (hipStream_t)stream;
    const DevCam dc = make_devcam(*cam);
    const uint32_t cap = (uint32_t)R_capacity;
    if (allmap_out) *allmap_out = w.allmap;
    uint8_t *touched = w.touched;   // (the backward tile kernel marks the surfels it reaches)
    const bool det = cfg->deterministic != 0;
    // one launch with predicted scales (cfg->deterministic = 2) where the default tile kernels hand compact lists over
    const bool det_one = cfg->deterministic == 2 && cfg->det_prev != nullptr && debug_state().bwd_variant == 3 &&
                         debug_state().fwd_variant == 3;
    SLS_REQUIRE(cfg->deterministic >= 0 && cfg->deterministic <= 2, "deterministic: 0 off, 1 two launches, 2 one launch with predicted scales");
    SLS_REQUIRE(cfg->deterministic != 2 || cfg->det_prev, "deterministic = 2 needs the keyframe's det_prev buffer");
    // With the default backward kernel and depth_ratio = 0 the consumer's second kernel is folded into the
    // backward tile kernel: every pixel block computes its dL/dallmap from kernel B's planes itself ...
    const bool fuse_c = debug_state().bwd_variant == 3 && cfg->depth_ratio == 0.0f;
    // ... and with the keyframe's launch-order buffer kernel B's work too
    constexpr bool no_fused_b = false;
    const bool fuse_b = fuse_c && cfg->block_order != nullptr && !no_fused_b;
    // the backward's blocks are launched most expensive first (cost recorded by the forward, sorted per XCD by eight
    // passenger workgroups — of the consumer's launch, or with fuse_b of the previous iteration's last launch): 8x2
    // kernels, XCD-interleaved tile mapping (T % 32 == 0)
    const bool order_bwd = debug_state().bwd_variant == 3 && debug_state().fwd_variant == 3 &&
                           (dc.GX * dc.GY) % 32 == 0 && kTileW == 16 && kTileH == 16;
    // cfg->phase: 0 = the whole iteration; 1 = up to the tile backward (+ the early gradient bitmap); 2 = the rest
    auto front = [&]() -> int {
        // (the status block is zeroed by thread 0 of preprocess_fwd, the iteration's first kernel)
        if (!cfg->workspace_ready) {
            // first use of this workspace: the gradient records must start from zero; afterwards the backward
            // of the projection leaves them zeroed behind itself (no 64*N-byte memset per iteration)
            ScopedTimer tm(T_GREC_MEMSET, st);
            SLS_HIP_CHECK(hipMemsetAsync(w.reg_accum, 0, w.zero_bytes, st));
        }

        // ---- forward ---------------------------------------------------------------
        uint32_t *okeys, *ovals, *n_dev;
        // the depth order lives in the workspace, or in a caller-owned buffer (one per keyframe, so that every
        // keyframe of a window can repair ITS order when the mapper samples keyframes at random)
        uint32_t *order = cfg->depth_order ? cfg->depth_order : w.order;
        depth_order_key_buffers(N, w.order_scratch, order, &okeys, &ovals, &n_dev);
        // Repairing the previous order: its first step (sorting windows of the old order by the new keys) rides in the
        // preprocess launch — it needs nothing the preprocess produces
        constexpr bool no_merge = false;
        const bool merged_sort = cfg->reuse_depth_order >= 1 && !no_merge;
        // Direct binning (sls_sort.hip) where it applies: no unsorted instance array, no scan of tiles_touched; the preprocess
        // then leaves the emission records in the form its first kernel gathers (rectangle + block box)
        const bool direct = bin_direct_possible(dc, N, cap);
        // (the staged API scans the count table's rows with a launch of its own instead)
        constexpr bool no_coarse = false;
        DirectBin db;
        if (direct) db = make_direct_bin(dc, N, w.sort_scratch, (uint2 *)w.serec, cfg->reuse_depth_order >= 1, !no_coarse);
        // (the direct binning reads the emission records only — not the rectangles, the tile counts, the depths or the
        //  block boxes as arrays of their own; a repair whose window sort rides in the preprocess launch computes its
        //  keys itself: 32 bytes per surfel that are not written)
        const bool trim = direct;
        int rc = launch_preprocess_fwd(dc, 1, cfg->scaling_max, cfg->scaling_max_penalty, w.reg_accum, N, xyz,
                                       scaling_raw, rotation_raw, opacity_raw, w.rec, w.radii, trim ? nullptr : w.rect,
                                       trim ? nullptr : w.tiles, trim ? nullptr : w.depth,
                                       (trim && merged_sort) ? nullptr : okeys, ovals, n_dev, st,
                                       (uint32_t *)status_dev, col_cs, row_cs, w.tmask, w.erec,
                                       merged_sort ? order : nullptr,
                                       merged_sort ? resort_comp_buffer(N, w.order_scratch) : nullptr, trim ? nullptr : w.sbox, direct ? 1 : 0,
                                       (direct && db.coarse) ? db.coarse : nullptr,
                                       (direct && db.coarse) ? (int)direct_coarse_words(dc, N) : 0);
        if (rc) return rc;
        ScanHandoff handoff = { nullptr, 0, nullptr, 0 };   // the binning finishes (or does not need) the scan of tiles_touched
        rc = launch_depth_order_scan(N, w.depth, w.tiles, order, w.offsets, &status_dev->R, w.order_scratch,
                                     w.order_scratch_bytes, 1, st, cfg->reuse_depth_order, &status_dev->overflow, &handoff,
                                     merged_sort, direct ? &db : nullptr, (const int4 *)w.erec, dc.GX);
        if (rc) return rc;
        int in_tmp = 0;
        const uint2 *bmask = nullptr;
        // (pairs instead of values only if both tile kernels are the default 8x2 ones: no other reads them)
        const bool pairs_ok = debug_state().fwd_variant == 3 && debug_state().bwd_variant == 3;
        if (direct) {
            rc = launch_bin_direct(dc, N, cap, db, handoff.counted != 0, order, w.erec, nullptr, nullptr, w.sort_scratch, w.vals,
                                   w.ranges, &status_dev->R, &status_dev->overflow, handoff.resort_windows,
                                   handoff.resort_edges, pairs_ok ? &bmask : nullptr, cfg->block_masks, st);
        } else {
            rc = launch_bin_sort(dc, N, &status_dev->R, cap, order, w.rect, w.tiles, dc.tile_cull ? w.tmask : nullptr,
                                 (dc.GX < 65536 && dc.GY < 65536) ? w.erec : nullptr, w.depth,
                                 w.offsets, w.tkeys, w.vals,
                                 w.tkeys_tmp, w.vals_tmp, w.sort_scratch, w.sort_scratch_bytes, &in_tmp, w.ranges, nullptr,
                                 &status_dev->overflow, st, &handoff, &status_dev->R,
                                 pairs_ok ? w.sbox : nullptr, &bmask, cfg->block_masks);
        }
        if (rc) return rc;
        // (with the pairs the plain value arrays are not written: the list IS the pairs, two words apart)
        const uint32_t *sorted_vals = bmask ? (const uint32_t *)bmask : (in_tmp ? w.vals_tmp : w.vals);
        const int vals_stride = bmask ? 2 : 1;
        rc = launch_render_fwd(dc, w.ranges, sorted_vals, w.rec, col_cs, row_cs, w.allmap, w.pix_state, w.pix_contrib,
                               nullptr, st, true, w.block_masks,    // (nobody reads the consumed counters here)
                               cfg->depth_ratio == 0.0f,            // (nor, then, the median / distortion planes: not tracked)
                               w.block_cost, bmask);
        if (rc) return rc;
        // ---- loss + dL/dallmap --------------------------------------------------------
        // With the keyframe's own launch-order buffer kernel B is folded in as well (fuse_b, above): the loss stage has no
        // launch; the order the backward walks is the one the keyframe's previous iteration left.
        ConsumerArgs cargs;
        rc = launch_consumer(H, W, w.allmap, gt_depth, valid, col_cs_half, row_cs_half, cfg->depth_ratio,
                             cfg->lambda_normal, cfg->lambda_alpha, n_valid, status_dev->loss_sums, w.dL_dallmap,
                             w.consumer_scratch, w.consumer_scratch_bytes, st, true, fuse_c ? &cargs : nullptr,
                             order_bwd ? dc.GX * dc.GY : 0, w.block_cost, w.block_order, fuse_b);
        if (rc) return rc;
        // ---- backward -----------------------------------------------------------------
        // (two launches: both accumulators start from zero; one launch: det_acc is left zeroed by every deterministic
        //  iteration's preprocess_bwd where it was written — and the first deterministic iteration on a workspace is a
        //  two-launch one, which also sets the fields' default scales)
        if (det && !det_one) SLS_HIP_CHECK(hipMemsetAsync(w.det_max, 0, w.det_bytes, st));
        const uint32_t *block_order = order_bwd ? (fuse_b ? cfg->block_order : w.block_order) : nullptr;
        rc = launch_render_bwd(dc, w.ranges, sorted_vals, w.rec, col_cs, row_cs, w.pix_state, w.pix_contrib, w.dL_dallmap,
                               w.grec, st, w.block_masks, cfg->depth_ratio == 0.0f, touched,    // the consumer's dL/d(median, distortion) are 0 then
                               fuse_c ? &cargs : nullptr, (det && !det_one) ? w.det_max : nullptr, det ? w.det_acc : nullptr, block_order,
                               vals_stride, (int)debug_state().fwd_variant, false,
                               det_one ? cfg->det_prev : nullptr, w.det_gex, &status_dev->overflow, fuse_b,
                               (fuse_b && order_bwd) ? block_order_tag(dc.GX * dc.GY) : 0u);
        if (rc) return rc;
        if (cfg->phase == 1 && cfg->grad_bitmap)      // the bitmap EARLY: an all-gather of it can overlap phase 2
            return launch_touched_bitmap(N, touched, scaling_raw, cfg->scaling_max, cfg->scaling_max_penalty,
                                         (const uint32_t *)status_dev, cfg->grad_bitmap, st);
        return SLS_OK;
    };
    int rc = SLS_OK;
    if (cfg->phase != 2) rc = front();
    if (rc || cfg->phase == 1) return rc;
    // flat gradient bucket: [xyz 3N | opacity N | scaling 2N | rotation 4N] (optimizer group order)
    float *g_xyz = grads, *g_op = grads + (size_t)3 * N, *g_sc = grads + (size_t)4 * N, *g_rot = grads + (size_t)6 * N;
    // ---- backward of the projection + optimiser -------------------------------------------
    // One keyframe per step: the Adam update is applied to each surfel right where its gradient is
    // produced (no gradient bucket round trip, no second pass over the parameters).
    AdamFuse fuse;
    memset(&fuse, 0, sizeof(fuse));
    fuse.clear_grec = 1;
    static_assert(sizeof(SlsMappingStatus) == 32, "the mirror copy moves 8 words");
    fuse.status_src = (uint32_t *)status_dev;
    fuse.reg_accum = w.reg_accum;
    fuse.touched = touched;
    fuse.status_mirror = (uint32_t *)cfg->status_mirror;
    fuse.void_flags = cfg->void_flags_out;
    fuse.void_count = 1; fuse.void_stride = 0;
    if (det) {
        fuse.det_max = w.det_max; fuse.det_acc = (const long long *)w.det_acc;
        fuse.det_prev = cfg->det_prev; fuse.det_gex = w.det_gex; fuse.det_onepass = det_one ? 1 : 0;
    }
    if (fuse_b) {
        // the tile backward's blocks left the loss terms (sls_consumer.hip: launch_consumer, no_launch) and the forward
        // its blocks' costs: this launch sums the first and sorts the second for the keyframe's next iteration
        fuse.loss_partials = (const float *)w.consumer_scratch;
        fuse.n_loss_partials = dc.GX * dc.GY * (kTilePix / 16);
        fuse.loss_w[0] = 1.0f / ((float)H * (float)W);
        fuse.loss_w[1] = n_valid > 0 ? cfg->lambda_normal * (1.0f / (float)n_valid) : 0.0f;
        fuse.loss_w[2] = n_valid > 0 ? cfg->lambda_alpha * (1.0f / (float)n_valid) : 0.0f;
        if (order_bwd) { fuse.order_T = dc.GX * dc.GY; fuse.order_cost = w.block_cost; fuse.order_out = cfg->block_order; }
    }
    if (cfg->grad_bitmap && cfg->phase != 2) {      // (phase 2: phase 1 wrote the bitmap early — a superset, left alone)
        fuse.grad_bitmap = cfg->grad_bitmap;
        fuse.grad_bitmap_words = (N + 63) / 64;
    }
    if (cfg->grad_chunk) {
        fuse.gchunk = cfg->grad_chunk;
        fuse.gbase = grads;
        fuse.void_flags = grads + cfg->grad_chunk;
        fuse.void_count = (int)cfg->grad_ranks;
        fuse.void_stride = (int)cfg->grad_chunk + 4;
    }
    const bool aligned = (N % 2 == 0) && ((((uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0);
    if (cfg->apply_adam && aligned) {
        fuse.enabled = 1;
        fuse.write_grads = cfg->keep_grads;
        fuse.c = make_adam_coef(cfg->beta1, cfg->beta2, cfg->eps, adam_step);
        fuse.lr_xyz = cfg->lr_xyz; fuse.lr_opacity = cfg->lr_opacity;
        fuse.lr_scaling = cfg->lr_scaling; fuse.lr_rotation = cfg->lr_rotation;
        fuse.exp_avg = exp_avg; fuse.exp_avg_sq = exp_avg_sq;
        fuse.skip_flag = &status_dev->overflow;
    }
    if (cfg->union_bitmap) {
        fuse.union_bitmap = cfg->union_bitmap; fuse.union_prefix = cfg->union_prefix;
        fuse.compact = cfg->grad_compact; fuse.compact_idx = cfg->grad_compact_index; fuse.compact_cap = cfg->grad_compact_capacity;
    }
    rc = launch_preprocess_bwd(dc, 1, cfg->scaling_max, cfg->scaling_max_penalty, N, xyz, scaling_raw, rotation_raw,
                               opacity_raw, w.radii, w.grec, g_xyz, g_sc, g_rot, g_op, st, &fuse);
    if (rc) return rc;
    if (cfg->apply_adam && !fuse.enabled) {
        SlsAdamGroup grp[4];
        memset(grp, 0, sizeof(grp));
        float *params[4] = { xyz, opacity_raw, scaling_raw, rotation_raw };
        const size_t offs[4] = { 0, (size_t)3 * N, (size_t)4 * N, (size_t)6 * N };
        const int64_t numel[4] = { (int64_t)3 * N, (int64_t)N, (int64_t)2 * N, (int64_t)4 * N };
        const float lrs[4] = { cfg->lr_xyz, cfg->lr_opacity, cfg->lr_scaling, cfg->lr_rotation };
        for (int k = 0; k < 4; ++k) {
            grp[k].param = params[k];
            grp[k].grad = grads + offs[k];
            grp[k].exp_avg = exp_avg + offs[k];
            grp[k].exp_avg_sq = exp_avg_sq + offs[k];
            grp[k].numel = numel[k];
            grp[k].lr = lrs[k];
        }
        rc = launch_adam(grp, 4, cfg->beta1, cfg->beta2, cfg->eps, adam_step, &status_dev->overflow, st);
        if (rc) return rc;
    }
    return SLS_OK;
}

}  // extern "C"
