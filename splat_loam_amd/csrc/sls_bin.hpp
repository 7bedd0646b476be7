// sls_bin.hpp — what the depth-order stage, the binning and the orchestration (sls_sort.hip, sls_pipeline.hip) share.
#pragma once
#include "sls_common.hpp"

namespace sls {

// What the depth-order stage hands to the binning that follows it
struct ScanHandoff {
    const uint32_t *block_sums;     // sums of tiles_touched over 256-blocks of depth-order positions (the emission
                                    // finishes the scan itself; null: it reads precomputed offsets)
    int resort_windows;             // > 0: the order was repaired, check these window edges (resort_verify)
    const uint64_t *resort_edges;
    int counted;                    // direct binning: the repair's merge already filled the count table
};

constexpr int kDirectChunk = 1024;     // depth positions per chunk of the direct binning (= the repair's window)
constexpr int kDirectMaxBins = 512;    // tiles it serves

// Direct binning (sls_sort.hip): the count table over chunks of 1024 depth positions and the per-position records
struct DirectBin {
    uint32_t *cnt;          // count table cnt[tile][chunk]: bins x nchunks words
    uint32_t *totals;       // bins words
    uint2 *serec;           // optional: per depth position the surfel's emission record {rectangle in one word, block box}
    int bins, nchunks, pos0;    // pos0: first depth position of chunk 0 (0, or -512: the repair's shifted windows)
    int stride;             // words per row of cnt (>= nchunks; a multiple of kDirectGroup with the coarse table: a group's
                            // counts of one tile are then one aligned 64-byte line)
    // optional (sls_mapping_step): coarse[group][tile] = the tile's instances in the chunks of group g (kDirectGroup
    // chunks each), summed with atomics by the counting kernels and ZEROED by the iteration's first kernel.  With it
    // bin_direct sums what lies in front of its chunk itself — the groups in front + the chunks of its own group — and
    // the row-scan launch disappears (one dependent launch less); cnt then keeps the RAW counts.
    uint32_t *coarse;
};
constexpr int kDirectGroup = 16;

bool bin_direct_possible(const DevCam &cam, int N, uint32_t cap);
DirectBin make_direct_bin(const DevCam &cam, int N, void *sort_scratch, uint2 *serec, bool repaired, bool coarse = false);
size_t direct_coarse_words(const DevCam &cam, int N);
int launch_bin_direct(const DevCam &cam, int N, uint32_t cap, const DirectBin &db, bool counted, const uint32_t *order,
                      const int32_t *erec_box, const int32_t *rect, const uint32_t *sbox, void *scratch, uint32_t *vals_out,
                      uint32_t *ranges, uint32_t *total_out, uint32_t *overflow, int resort_windows,
                      const uint64_t *resort_edges, const uint2 **bmask_out, int bmask_mode, hipStream_t st,
                      uint32_t *status_mirror = nullptr);     // (pinned host memory: the status block, early — sls_forward_ws)
int launch_depth_order_scan(int N, const float *depth, const uint32_t *tiles, uint32_t *order, uint32_t *offsets,
                            uint32_t *total_out, void *scratch, size_t scratch_bytes, int keys_prefilled,
                            hipStream_t st, int reuse_order = 0, uint32_t *fail_flag = nullptr,
                            ScanHandoff *handoff = nullptr, bool window_sort_done = false,
                            const DirectBin *direct = nullptr, const int4 *erec_box = nullptr, int GX = 0);

}  // namespace sls
