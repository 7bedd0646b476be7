// sls_render.hip — dispatch of the tile kernels (A6 forward blend, A7 backward; SURVEY.md §8a, maths in
// DESIGN.md §2).  The kernels live in sls_render_block.hip: one wave per 16-pixel block x 4 list entries
// per step.  Two block shapes are instantiated — 8x2 (the default) and 4x4 (kept as an independent
// cross-check: `sls_debug_variant`, tests/test_gpu_parity.py::test_tile_kernel_variants_agree_with_checker);
// the two earlier generations (workgroup per tile, wave per 8x8 sub-tile) were removed in round 2.
// No MFMA: there is no dense contraction here (BASELINE.json north_star).
#include "sls_tile.hpp"

namespace sls {

// kernel variants (sls_debug_variant): 2 / 3 = one wave per 4x4 / 8x2 pixel block x 4 surfel slots
int launch_render_fwd_block(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                            const float *col_cs, const float *row_cs, float *allmap, float *pix_state,
                            uint32_t *pix_contrib, uint32_t *tile_consumed, uint64_t *block_masks, int shape,
                            hipStream_t st, bool lean, uint32_t *block_cost, const uint2 *bmask, bool order_in_handover);
bool handover_has_order(int T);
const uint32_t *handover_block_order(const uint64_t *block_masks, int T);
size_t block_mask_bytes(uint64_t cap, int T);
int launch_render_bwd_block(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                            const float *col_cs, const float *row_cs, const float *pix_state,
                            const uint32_t *pix_contrib, const float *dL_dallmap, float *grec,
                            const uint64_t *block_masks, int shape, hipStream_t st, bool lean, uint8_t *touched,
                            const struct ConsumerArgs *fused_consumer, uint32_t *det_max, unsigned long long *det_acc,
                            const uint32_t *block_order, int vals_stride, bool dense, const uint8_t *det_prev,
                            const uint32_t *det_gex, uint32_t *det_flag, bool consumer_b_inline, uint32_t order_tag);
// Diagnostic switches are per PROCESS (sls_common.hpp: DebugState): torch runs a backward node on its autograd
// device thread, not on the thread that called sls_debug_variant / sls_debug_wave_cycles, so per-thread state
// would silently not reach sls_backward under loss.backward().  They are tuning / test aids only: the data path
// keeps no mutable state of its own.
DebugState &debug_state()
{
    static DebugState s;
    return s;
}

// ---------------------------------------------------------------------------
int launch_render_fwd(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                      const float *col_cs, const float *row_cs, float *allmap, float *pix_state,
                      uint32_t *pix_contrib, uint32_t *tile_consumed, hipStream_t st, bool consumed_zeroed,
                      uint64_t *block_masks, bool no_median_dist, uint32_t *block_cost, const uint2 *bmask,
                      bool order_in_handover)
{
    const int T = cam.GX * cam.GY;
    // the block kernels combine their blocks' counters with atomicMax: start from zero
    if (tile_consumed && !consumed_zeroed)
        SLS_HIP_CHECK(hipMemsetAsync(tile_consumed, 0, sizeof(uint32_t) * (size_t)T, st));
    return launch_render_fwd_block(cam, ranges, vals, rec, col_cs, row_cs, allmap, pix_state, pix_contrib,
                                   tile_consumed, block_masks, debug_state().fwd_variant - 2, st, no_median_dist,
                                   debug_state().fwd_variant == 3 ? block_cost : nullptr, bmask, order_in_handover);
}

int launch_render_bwd(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                      const float *col_cs, const float *row_cs, const float *pix_state,
                      const uint32_t *pix_contrib, const float *dL_dallmap, float *grec, hipStream_t st,
                      const uint64_t *block_masks, bool no_median_dist_grad, uint8_t *touched,
                      const struct ConsumerArgs *fused_consumer, uint32_t *det_max, unsigned long long *det_acc,
                      const uint32_t *block_order, int vals_stride, int block_masks_shape, bool order_in_handover,
                      const uint8_t *det_prev, const uint32_t *det_gex, uint32_t *det_flag, bool consumer_b_inline,
                      uint32_t order_tag)
{
    // vals_stride: 1 = plain list of surfel indices, 2 = the tile sort's (surfel, block mask) pairs.
    // block_masks_shape: the pixel-block shape (sls_debug_variant numbering: 2 = 4x4, 3 = 8x2) of the forward that
    // wrote the compact lists in `block_masks`; they are walked only by a backward of the same shape — any other
    // combination culls the tile's list itself (the buffer's tag word is the kernel's last guard).
    const int variant = (det_max || det_prev) ? 3 : (int)debug_state().bwd_variant;
    const bool dense = block_masks != nullptr && block_masks_shape == variant;
    // no order from the caller: the staged forward left one in the hand-over buffer (block_order_kernel)
    if (!block_order && order_in_handover && dense && variant == 3 && handover_has_order(cam.GX * cam.GY))
        block_order = handover_block_order(block_masks, cam.GX * cam.GY);
    return launch_render_bwd_block(cam, ranges, vals, rec, col_cs, row_cs, pix_state, pix_contrib, dL_dallmap,
                                   grec, block_masks, variant - 2, st, no_median_dist_grad,
                                   touched, fused_consumer, det_max, det_acc,
                                   (debug_state().bwd_variant == 3 || det_max || det_prev) ? block_order : nullptr, vals_stride, dense,
                                   det_prev, det_gex, det_flag, consumer_b_inline, order_tag);
}

}  // namespace sls
