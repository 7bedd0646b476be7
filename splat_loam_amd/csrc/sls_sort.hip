// sls_sort.hip — tile binning (A3), sorting (A4) and per-tile range detection
// (A5).  SURVEY.md §8a; all integer work, checked bit-exactly.
//
// The sorted list is defined by the 64-bit key (tile << 32 | depth bits) with
// ties broken by surfel index (D3, D9).  It is produced WITHOUT ever sorting
// 64-bit keys over the R tile instances:
//   1. the N surfels are sorted by (29-bit depth key, index) — 3 passes of 10-bit digits
//      over N (u32 key, u32 index) pairs, or the previous iteration's order is REPAIRED
//      (windowed bitonic re-sort + exactness check, see "Temporal re-sort" below);
//   2. instances are emitted in that order                — so within any tile
//      the emission order already is the final order;
//   3. one STABLE pass over the R instances by tile id (9-bit digit for T <= 512, up to
//      11 bits; two passes beyond 2048 tiles); when tile id and surfel index fit 32 bits
//      together an instance is ONE packed word.  The pass's digit bases are the tile ranges.
// LSD radix principle (least-significant part first, stable passes after); the
// result is bit-identical to a stable 64-bit sort and costs ~24 B of HBM traffic
// per instance instead of ~190 B.
//
// Radix sort building block (wave64-native, no vendor library):
//   * 8..11-bit digits (sort_plan); the unit of work is ONE WAVE owning kSortWaveItems =
//     1024 consecutive items (16 rounds of 64): no workgroup barrier inside a pass, each
//     wave keeps its 2^BITS running bucket cursors in a private LDS slice; a workgroup is
//     SortBlock<BITS>::kWaves waves (8 up to 10-bit digits, 4 for 11) so that the count
//     table is written and read back in runs of that many chunks per digit;
//   * stable ranking inside a round by BITS ballots (the set of lanes holding my
//     digit), rank = popcount(peers below me);
//   * per pass: histogram -> per-digit row scan over chunks -> scatter;
//   * the item count is read from DEVICE memory (count_ptr), grids are sized for
//     a host-side capacity, so a whole iteration can be enqueued without a
//     device->host sync (sls_mapping_step).
#include <cstdlib>
#include "sls_common.hpp"
#include "sls_resort.hpp"
#include "sls_bin.hpp"

namespace sls {

// (build-time knobs for A/B runs; measured on the bench scene, A/B inside one box, iteration time with
//  rounds x waves per block = 16x16: 0.2652 ms, 16x8: 0.2613 (scatter 18.6 -> 15.8 us: twice the workgroups for
//  the 1260 wave chunks of the tile sort), 16x4: 0.266 (the stray-word table access), 24x8: 0.2618, 32x8: 0.264,
//  8x16: 0.268 and 8x8: 0.269 (row scan +4 us), 4x16: 0.280)
#ifndef SLS_SORT_ROUNDS
#define SLS_SORT_ROUNDS 16
#endif
#ifndef SLS_SORT_WAVES
#define SLS_SORT_WAVES 8
#endif
constexpr int kSortRounds = SLS_SORT_ROUNDS;
constexpr int kSortWaveItems = kWave * kSortRounds;  // 1024 items per wave
constexpr int kSortMaxBins = 2048;   // 11-bit digits at most
constexpr int kDepthKeyBits = 29;    // compressed depth key, see depth_order_key()

__device__ __forceinline__ uint32_t load_count(const uint32_t *count_ptr, uint32_t cap)
{
    const uint32_t c = *count_ptr;
    return c < cap ? c : cap;
}

__device__ __forceinline__ void resort_verify(int nwin, const uint64_t *__restrict__ edges, uint32_t *__restrict__ flag);

// The tile sort's list with block masks (launch_bin_sort, the mapping iteration): per instance, in list order, the
// pair (surfel, mask of the sixteen 8x2 pixel blocks of its tile that the surfel's support box can reach) INSTEAD of
// the bare surfel index.  An instance is then a 64-bit word: the packed (tile, surfel) word below, the surfel's block
// box (sls_common.hpp: make_block_box, one word per surfel from the preprocess, fetched by the emission with the
// gather it makes anyway) above; the pass sorts on the same digit, and the scatter — tile and box in registers —
// stores the pair with the ONE scattered store it would spend on the value.  The forward tile kernel reads the masks
// instead of the records to decide what to stage (render_fwd_dense_kernel).  (Measured on the way: masks computed in
// the emission from the records' support boxes — a gather from the 40 MB record array — +12...17 us there at 500 k
// surfels and +4.7 us in the scatter that carried them; computed in the scatter from a per-surfel gather but stored
// as a second, 2-byte array: +8.1 us in the scatter at 500 k — a scattered store instruction costs what it costs,
// whatever its width; the gather alone, with the pair store: +6.3 us at 500 k, +3.2 us at 50 k — a round trip in a
// chain of latencies.)
struct BlockMaskArgs { uint2 *out; int GX; float invGX; int NC; };
// bit 2*by + bx of the mask of an instance of `tile`; box = the surfel's block box
__device__ __forceinline__ uint32_t block_mask_of(const BlockMaskArgs &a, uint32_t tile, uint32_t box)
{
    const int ty = (int)(((float)tile + 0.5f) * a.invGX), tx = (int)tile - ty * a.GX;
    const int bc_lo = (int)(box & 511u), bc_n = (int)((box >> 9) & 1023u) - 1;       // columns [bc_lo, bc_lo + bc_n] mod NC
    const int br_lo = (int)((box >> 19) & 63u), br_hi = (int)(box >> 25) - 1;        // rows [br_lo, br_hi]
    const int r0 = max(br_lo - ty * (kTileH / 2), 0), r1 = min(br_hi - ty * (kTileH / 2), kTileH / 2 - 1);
    const uint32_t ym = r0 <= r1 ? (((4u << (2 * r1)) - 1u) & ~((1u << (2 * r0)) - 1u)) : 0u;
    int d0 = 2 * tx - bc_lo;
    d0 += d0 < 0 ? a.NC : 0;
    const int d1 = d0 + 1 >= a.NC ? d0 + 1 - a.NC : d0 + 1;
    const uint32_t xm = (d0 <= bc_n ? 0x5555u : 0u) | (d1 <= bc_n ? 0xAAAAu : 0u);
    return ym & xm;
}

// ---------------------------------------------------------------------------
// Direct binning (round 4): the tile instances are never materialised unsorted.
//
// The list order inside a tile is the depth order of its surfels, so an instance's place in the sorted list is
//     base[tile] + (instances of that tile emitted by earlier CHUNKS of depth positions) + (its rank inside its chunk),
// all of which follow from a count table cnt[tile][chunk] over chunks of 1024 depth positions:
//   1. the kernel that leaves the depth order — the repair's merge (resort_merge_kernel<true>), or gather_count_kernel
//      after a from-scratch sort — gathers, per position, the surfel's emission record {rectangle, block box}, writes it
//      to an array indexed by POSITION (the third kernel reads it coalesced) and counts its chunk's tiles in LDS:
//      one column of the table;
//   2. sort_rowscan_kernel: exclusive scan of every tile's row + the tile totals;
//   3. bin_direct_kernel, one workgroup per chunk: digit bases (= tile ranges, R), per-wave counts -> cursors, then every
//      wave deals its instances out 64 at a time (owner by binary search in the wave's scan, tile from the rectangle),
//      ranks them with the BITS ballots of the radix scatter and stores (surfel[, block mask]) at the final position.
// Against emission + histogram + row scan + scatter: two launches and the unsorted instance array (written once, read
// twice) less; the scattered 16-byte gathers of the emission move into the latency-bound merge, which made a 4-byte
// gather per position anyway.  Tiles <= 512 (every size the reference's mapper meets and BASELINE config 3), D10 off;
// anything else takes the emission + radix pass below.
// ---------------------------------------------------------------------------
#ifdef SLS_TRACE
// Experiment build only (tools/build_variant.sh trace ... -DSLS_TRACE; read back with sls_debug_read_bin_trace, profiles/r04f_bin_trace.txt): every wave of bin_direct_kernel
// and every workgroup of the counting merge records the 100 MHz wall clock at its phases.
__device__ uint32_t g_bin_trace[32768 * 8];      // per wave: start, loads done, counted, cursors ready, end, rounds, S, (pad)
__device__ uint32_t g_merge_trace[1024 * 4];     // per workgroup: start, network done, counted + stored, end
extern "C" int sls_debug_read_bin_trace(uint32_t *host_bin, uint32_t *host_merge)
{
    int rc = (int)hipMemcpyFromSymbol(host_bin, HIP_SYMBOL(g_bin_trace), sizeof(g_bin_trace));
    if (rc == 0) rc = (int)hipMemcpyFromSymbol(host_merge, HIP_SYMBOL(g_merge_trace), sizeof(g_merge_trace));
    return rc;
}
#define SLS_BT(k_) do { if (lane == 0) { const int wi_ = (int)blockIdx.x * WAVES + w; if (wi_ < 32768) g_bin_trace[8 * wi_ + (k_)] = (uint32_t)wall_clock64(); } } while (0)
#define SLS_BTV(k_, v_) do { if (lane == 0) { const int wi_ = (int)blockIdx.x * WAVES + w; if (wi_ < 32768) g_bin_trace[8 * wi_ + (k_)] = (uint32_t)(v_); } } while (0)
#define SLS_MT(k_) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_merge_trace[4 * blockIdx.x + (k_)] = (uint32_t)wall_clock64(); } while (0)
#else
#define SLS_BT(k_)
#define SLS_BTV(k_, v_)
#define SLS_MT(k_)
#endif

// The emission record of a surfel, 8 bytes: x = its tile rectangle in ONE word — txlo (9 bits) | ncols (10) | tylo (6) |
// nrows (7), zero: nothing emitted; grids up to 512 x 64 tiles (bin_direct_possible) — y = its block box.
__host__ __device__ inline uint32_t pack_rect32(int txlo, int ncols, int tylo, int nrows)
{
    return (uint32_t)txlo | ((uint32_t)ncols << 9) | ((uint32_t)tylo << 19) | ((uint32_t)nrows << 25);
}
__device__ __forceinline__ uint2 load_emit_record(const uint2 *__restrict__ erec_box, const int4 *__restrict__ rect,
                                                 const uint32_t *__restrict__ sbox, uint32_t g)
{
    if (erec_box) return erec_box[g];
    const int4 rc = rect[g];                               // {txlo, ncols, tylo, nrows}
    return make_uint2(pack_rect32(rc.x, rc.y, rc.z, rc.w), sbox ? sbox[g] : 0u);
}
// one LDS count per tile of the rectangle (row-major, x wrapping modulo the grid width: D5 / D9)
__device__ __forceinline__ void count_rect_tiles(uint32_t r32, int GX, uint32_t *s_hist)
{
    const int txlo = (int)(r32 & 511u), ncols = (int)((r32 >> 9) & 1023u), tylo = (int)((r32 >> 19) & 63u), nrows = (int)(r32 >> 25);
    for (int y = 0; y < nrows; ++y) {
        const int row = (tylo + y) * GX;
        for (int k = 0; k < ncols; ++k) {
            int tx = txlo + k;
            if (tx >= GX) tx -= GX;
            atomicAdd(&s_hist[row + tx], 1u);
        }
    }
}

// The same for a whole wave (call it from uniform control flow, r32 = 0 for lanes without a surfel).  A lane walks its
// own rectangle while that is small; rectangles of more than kBigRect tiles — a surfel within a metre of the sensor covers
// a quarter of the image and more: 420 of 500 k at the bench scene's last keyframe, all of them in the first chunks of the
// depth order — are walked by the 64 lanes TOGETHER, one rectangle after the other: a lane that walks 512 tiles alone
// keeps its wave for 512 dependent LDS atomics (20 us) while 63 lanes wait, and its neighbours' rectangles hit the
// same words at the same moment (profiles/r05a_bin_tail.txt).
constexpr uint32_t kBigRect = 128;
__device__ __forceinline__ void count_rect_tiles_wave(uint32_t r32, int GX, uint32_t *s_hist)
{
    const uint32_t ncols = (r32 >> 9) & 1023u, t = ncols * (r32 >> 25);
    uint64_t big = __ballot(t > kBigRect);
    if (t <= kBigRect) count_rect_tiles(r32, GX, s_hist);
    const int lane = (int)(threadIdx.x & 63u);
    while (big) {
        const int src = (int)__builtin_ctzll(big);
        big &= big - 1ull;
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)r32, src);
        const uint32_t nc = (r >> 9) & 1023u, n = nc * (r >> 25);
        const float inc = __builtin_amdgcn_rcpf((float)nc);
        for (uint32_t k = (uint32_t)lane; k < n; k += 64u) {
            const uint32_t ky = (uint32_t)(((float)k + 0.5f) * inc);      // (k / ncols: see bin_direct_kernel)
            int tx = (int)(r & 511u) + (int)(k - ky * nc);
            if (tx >= GX) tx -= GX;
            atomicAdd(&s_hist[((int)((r >> 19) & 63u) + (int)ky) * GX + tx], 1u);
        }
    }
}

// Counting a whole workgroup's rectangles through per-row DIFFERENCE counts (the counting merge and gather_count: one
// histogram per workgroup).  A rectangle row of ncols >= kDiffCols tiles is two LDS atomics — +1 where it begins, -1
// behind its end (three when it wraps) — instead of ncols; narrower rows add to their tiles directly.  A surfel within a
// metre of the sensor has rows of 30 ... 128 tiles, and a chunk at the front of the depth order holds 70 k instances of
// them (profiles/r05a_bin_tail.txt): the merge took 25-33 us at the bench window's last keyframes against 13 at its first.
// s_diff: (GX + 1) words per tile row, zeroed by the caller; diff_finish() turns them into counts and adds them to s_hist.
// Tile rows must be whole waves of the finishing loop: GX % 64 == 0 (else the callers count the old way).
constexpr int kDiffCols = 8;
// (*s_wide, zeroed by the caller: set where a row went into the difference counts — a workgroup of far surfels, whose
//  rectangles are one to three tiles wide, then skips diff_finish and its barrier)
__device__ __forceinline__ void count_rect_rows_diff(uint32_t r32, int GX, uint32_t *s_hist, int *s_diff, int *s_wide)
{
    const int txlo = (int)(r32 & 511u), ncols = (int)((r32 >> 9) & 1023u), tylo = (int)((r32 >> 19) & 63u), nrows = (int)(r32 >> 25);
    if (ncols < kDiffCols) { count_rect_tiles(r32, GX, s_hist); return; }
    *s_wide = 1;
    const int end = txlo + ncols;
    for (int y = 0; y < nrows; ++y) {
        int *row = s_diff + (tylo + y) * (GX + 1);
        atomicAdd(&row[txlo], 1);
        if (end <= GX) atomicAdd(&row[end], -1);            // (slot GX: behind the row, never read)
        else { atomicAdd(&row[0], 1); atomicAdd(&row[end - GX], -1); }
    }
}
// every thread of the workgroup; bins = GY * GX tiles, threads a multiple of 64; between two __syncthreads of the caller
__device__ __forceinline__ void diff_finish(int GX, int bins, uint32_t *s_hist, const int *s_diff, int tid, int nthreads)
{
    const int lane = tid & 63;
    for (int d0 = (tid >> 6) * 64; d0 < bins; d0 += nthreads) {      // a wave takes 64 consecutive tiles of ONE row
        const int d = d0 + lane, row = d0 / GX, x0 = d0 - row * GX;
        const int *r = s_diff + row * (GX + 1);
        int carry = 0;
        for (int x = lane; x < x0; x += 64) carry += r[x];            // (the row's tiles in front of this wave's 64)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) carry += __shfl_xor(carry, off, 64);
        int v = r[x0 + lane];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off, 64);
            if (lane >= off) v += u;
        }
        s_hist[d] += (uint32_t)(carry + v);
    }
}

// ---------------------------------------------------------------------------
// step 1 of a pass: per-wave-chunk digit histogram -> cnt[digit][chunk]
// ---------------------------------------------------------------------------
// A block is WAVES chunks: the count table cnt[digit][chunk] is written (and read back by the scatter)
// in runs of WAVES consecutive chunks per digit — 32 contiguous bytes with 8 waves — instead of one
// stray word per (digit, wave).  The LDS rows are padded by one word so that the transposed access
// (lanes = consecutive waves of one digit) spreads over the banks.
template <int BITS> struct SortBlock { static constexpr int kWaves = BITS <= 9 ? SLS_SORT_WAVES : (BITS == 10 ? (SLS_SORT_WAVES < 8 ? SLS_SORT_WAVES : 8) : 4); };

template <typename KeyT, int BITS>
__global__ __launch_bounds__(64 * SortBlock<BITS>::kWaves) void sort_hist_kernel(
    const KeyT *__restrict__ keys, const uint32_t *__restrict__ count_ptr, uint32_t cap, int shift,
    uint32_t *__restrict__ cnt, int nchunks_cap)
{
    constexpr int BINS = 1 << BITS, WAVES = SortBlock<BITS>::kWaves, STR = BINS + 1;
    __shared__ uint32_t s_hist[WAVES * STR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk0 = blockIdx.x * WAVES, chunk = chunk0 + wave;
    const uint32_t R = load_count(count_ptr, cap);
    const int nchunks = (int)((R + kSortWaveItems - 1) / kSortWaveItems);
    if (chunk0 >= nchunks) return;             // whole block beyond the data
#pragma unroll
    for (int k = 0; k < BINS / 64; ++k) s_hist[wave * STR + lane + 64 * k] = 0;
    __builtin_amdgcn_wave_barrier();
    if (chunk < nchunks) {
        const uint32_t base = (uint32_t)chunk * kSortWaveItems + lane;
        KeyT k[kSortRounds];
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) k[r] = keys[min(base + (uint32_t)r * 64, R - 1u)];   // (chunk < nchunks: R >= 1)
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const uint32_t idx = base + (uint32_t)r * 64;
            if (idx < R) atomicAdd(&s_hist[wave * STR + ((uint32_t)(k[r] >> shift) & (uint32_t)(BINS - 1))], 1u);
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BINS * WAVES; idx += 64 * WAVES) {
        const int d = idx / WAVES, w = idx % WAVES;
        if (chunk0 + w < nchunks_cap) cnt[(size_t)d * nchunks_cap + chunk0 + w] = s_hist[w * STR + d];
    }
}

// step 2: block d scans row d of cnt[][] exclusively in place, writes the row total.
// Thread t owns P = ceil(nchunks / 256) CONSECUTIVE entries: one pass to sum them, one wave scan + one barrier for
// the 256 partial sums, one pass to write the running prefix (a loop over 256-entry slabs with two barriers each
// cost 2 us more on the 1260-chunk rows of the tile sort: the kernel is nothing but its chain of latencies).
// nchunks_fixed > 0: the rows have that many entries (chunks of the emission, see emit_tiles_kernel)
__global__ __launch_bounds__(256) void sort_rowscan_kernel(uint32_t *__restrict__ cnt,
                                                           const uint32_t *__restrict__ count_ptr, uint32_t cap,
                                                           int nchunks_cap, uint32_t *__restrict__ totals,
                                                           int nchunks_fixed)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t R = nchunks_fixed > 0 ? 0u : load_count(count_ptr, cap);
    const int nchunks = nchunks_fixed > 0 ? nchunks_fixed : (int)((R + kSortWaveItems - 1) / kSortWaveItems);
    uint32_t *row = cnt + (size_t)blockIdx.x * nchunks_cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = (nchunks + 255) / 256, i0 = (int)threadIdx.x * P, i1 = min(i0 + P, nchunks);
    constexpr int kKeep = 8;                 // entries kept in registers between the two passes (P <= 8 up to 2 M items)
    uint32_t keep[kKeep];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) { keep[j] = (i0 + j < i1) ? row[i0 + j] : 0u; sum += keep[j]; }
    for (int i = i0 + kKeep; i < i1; ++i) sum += row[i];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
        if (i0 + j < i1) row[i0 + j] = run;
        run += keep[j];
    }
    for (int i = i0 + kKeep; i < i1; ++i) { const uint32_t v = row[i]; row[i] = run; run += v; }
    if (threadIdx.x == 255) totals[blockIdx.x] = run;
}

// step 3: stable scatter
template <typename KeyT, int BITS>
__global__ __launch_bounds__(64 * SortBlock<BITS>::kWaves) void sort_scatter_kernel(const KeyT *__restrict__ keys_in,
                                                           const uint32_t *__restrict__ vals_in,
                                                           KeyT *__restrict__ keys_out,
                                                           uint32_t *__restrict__ vals_out,
                                                           const uint32_t *__restrict__ count_ptr, uint32_t cap,
                                                           int shift, const uint32_t *__restrict__ cnt,
                                                           const uint32_t *__restrict__ totals, int nchunks_cap,
                                                           uint2 *__restrict__ ranges_out, int nranges,
                                                           uint32_t packed_val_mask, BlockMaskArgs bm)
{
    constexpr int BINS = 1 << BITS, PER = BINS / 256;   // digit totals handled per thread (of the first 256)
    constexpr int WAVES = SortBlock<BITS>::kWaves, STR = BINS + 1;
    __shared__ uint32_t s_cursor[WAVES * STR];
    __shared__ uint32_t s_digit_base[BINS];
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk0 = blockIdx.x * WAVES, chunk = chunk0 + wave;
    // Everything the block needs from memory is requested up front — its items, its rows of the count
    // table, the digit totals, the item count — so the waits below overlap into ONE round trip (with one
    // wave per SIMD nothing else would hide them).  Loads are guarded by the buffers' capacity, validity
    // (idx < R) is applied afterwards.
    const uint32_t base = (uint32_t)chunk * kSortWaveItems + lane;
    KeyT k[kSortRounds];
    uint32_t v[kSortRounds];
    // (clamped addresses instead of predicated loads: a conditional load per round compiles into a branch
    //  and a full wait per round — sixteen serialised round trips)
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) k[r] = keys_in[min(base + (uint32_t)r * 64, cap - 1u)];
    if (vals_in) {
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) v[r] = vals_in[min(base + (uint32_t)r * 64, cap - 1u)];
    } else {   // packed mode: the value is the low part of the key
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) v[r] = (uint32_t)k[r] & packed_val_mask;
    }
    constexpr int CPT = BINS / 64;             // count-table entries per thread: BINS * WAVES / (64 * WAVES)
    uint32_t my_cnt[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = threadIdx.x + q * 64 * WAVES, d = idx / WAVES, w = idx % WAVES;
        my_cnt[q] = (chunk0 + w < nchunks_cap) ? cnt[(size_t)d * nchunks_cap + chunk0 + w] : 0u;
    }
    const bool scanner = threadIdx.x < 256;
    uint32_t tot[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) tot[q] = scanner ? totals[threadIdx.x * PER + q] : 0u;
    const uint32_t R = load_count(count_ptr, cap);
    const int nchunks = (int)((R + kSortWaveItems - 1) / kSortWaveItems);
    if (chunk0 >= nchunks) return;   // whole block beyond the data
    {   // exclusive scan of the BINS digit totals (PER consecutive ones per thread, first 256 threads)
        uint32_t loc[PER];
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { loc[k] = v; v += tot[k]; }
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (scanner && lane == 63) s_wave[wave] = incl;
        __syncthreads();
        if (scanner) {
            uint32_t wave_prefix = 0;
            for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
#pragma unroll
            for (int k = 0; k < PER; ++k) s_digit_base[threadIdx.x * PER + k] = wave_prefix + incl - v + loc[k];
            // single-pass sort by tile id: the digit bases ARE the tile ranges (A5)
            if (ranges_out && blockIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const int d = threadIdx.x * PER + k;
                    const uint32_t b0 = wave_prefix + incl - v + loc[k], c = tot[k];
                    if (d < nranges) ranges_out[d] = c ? make_uint2(b0, b0 + c) : make_uint2(0u, 0u);
                }
            }
        }
        __syncthreads();
    }
    // cursors of the block's WAVES chunks: runs of WAVES consecutive counts per digit (see sort_hist_kernel)
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = threadIdx.x + q * 64 * WAVES, d = idx / WAVES, w = idx % WAVES;
        s_cursor[w * STR + d] = s_digit_base[d] + my_cnt[q];
    }
    __syncthreads();
    if (chunk >= nchunks) return;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint32_t idx = base + (uint32_t)r * 64;
        const bool valid = idx < R;
        const uint32_t digit = (uint32_t)(k[r] >> shift) & (uint32_t)(BINS - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t count = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = s_cursor[wave * STR + digit] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            if (keys_out) keys_out[pos] = k[r];     // (a caller that only wants the permutation passes null in the last pass)
            if (sizeof(KeyT) == 8 && bm.out) bm.out[pos] = make_uint2(v[r], block_mask_of(bm, digit, (uint32_t)((uint64_t)k[r] >> 32)));
            else vals_out[pos] = v[r];
            if (rank == count - 1) s_cursor[wave * STR + digit] = pos + 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// step 3 when the chunks are the EMISSION's (emit_tiles_kernel counted the digits of the instances each of its
// workgroups wrote: no histogram launch): chunk c = instances [chunk_start[c], chunk_start[c + 1]), a few hundred
// to a few thousand; one wave per chunk, 16 rounds of 64 in registers at a time.  Same positions as the pass over
// fixed 1024-item chunks: a stable partition does not care where the chunk boundaries are.
template <typename KeyT, int BITS>
__global__ __launch_bounds__(64 * SortBlock<BITS>::kWaves) void sort_scatter_chunks_kernel(
    const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, KeyT *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ chunk_start, uint32_t cap, int shift,
    const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ totals, int nchunks,
    uint2 *__restrict__ ranges_out, int nranges, uint32_t packed_val_mask, BlockMaskArgs bm)
{
    constexpr int BINS = 1 << BITS, PER = BINS / 256;
    constexpr int WAVES = SortBlock<BITS>::kWaves, STR = BINS + 1;
    __shared__ uint32_t s_cursor[WAVES * STR];
    __shared__ uint32_t s_digit_base[BINS];
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk0 = blockIdx.x * WAVES, chunk = chunk0 + wave;
    // (the chunk's bounds, its rows of the count table and the digit totals travel together; the items follow)
    const uint32_t cs = min(chunk_start[min(chunk, nchunks)], cap), ce = min(chunk_start[min(chunk + 1, nchunks)], cap);
    KeyT k[kSortRounds];
    uint32_t v[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) k[r] = keys_in[min(cs + (uint32_t)(r * 64 + lane), cap - 1u)];
    if (vals_in) {
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) v[r] = vals_in[min(cs + (uint32_t)(r * 64 + lane), cap - 1u)];
    }
    constexpr int CPT = BINS / 64;
    uint32_t my_cnt[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = threadIdx.x + q * 64 * WAVES, d = idx / WAVES, w = idx % WAVES;
        my_cnt[q] = (chunk0 + w < nchunks) ? cnt[(size_t)d * nchunks + chunk0 + w] : 0u;
    }
    const bool scanner = threadIdx.x < 256;
    uint32_t tot[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) tot[q] = scanner ? totals[threadIdx.x * PER + q] : 0u;
    {   // exclusive scan of the BINS digit totals (PER consecutive ones per thread, first 256 threads)
        uint32_t loc[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { loc[q] = sum; sum += tot[q]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (scanner && lane == 63) s_wave[wave] = incl;
        __syncthreads();
        if (scanner) {
            uint32_t wave_prefix = 0;
            for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
#pragma unroll
            for (int q = 0; q < PER; ++q) s_digit_base[threadIdx.x * PER + q] = wave_prefix + incl - sum + loc[q];
            if (ranges_out && blockIdx.x == 0) {      // the digit bases ARE the tile ranges (A5)
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int d = threadIdx.x * PER + q;
                    const uint32_t b0 = wave_prefix + incl - sum + loc[q], c = tot[q];
                    if (d < nranges) ranges_out[d] = c ? make_uint2(b0, b0 + c) : make_uint2(0u, 0u);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = threadIdx.x + q * 64 * WAVES, d = idx / WAVES, w = idx % WAVES;
        s_cursor[w * STR + d] = s_digit_base[d] + my_cnt[q];
    }
    __syncthreads();
    if (chunk >= nchunks) return;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (uint32_t b0 = cs; b0 < ce; b0 += (uint32_t)kSortWaveItems) {
        if (b0 != cs) {     // (a chunk of more than 1024 instances: the next 16 rounds)
#pragma unroll
            for (int r = 0; r < kSortRounds; ++r) k[r] = keys_in[min(b0 + (uint32_t)(r * 64 + lane), cap - 1u)];
            if (vals_in) {
#pragma unroll
                for (int r = 0; r < kSortRounds; ++r) v[r] = vals_in[min(b0 + (uint32_t)(r * 64 + lane), cap - 1u)];
            }
        }
        if (!vals_in) {     // packed mode: the value is the low part of the key
#pragma unroll
            for (int r = 0; r < kSortRounds; ++r) v[r] = (uint32_t)k[r] & packed_val_mask;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < kSortRounds; ++r) {
            const uint32_t idx = b0 + (uint32_t)(r * 64 + lane);
            if (b0 + (uint32_t)(r * 64) >= ce) break;       // (wave-uniform)
            const bool valid = idx < ce;
            const uint32_t digit = (uint32_t)(k[r] >> shift) & (uint32_t)(BINS - 1);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const bool bit = (digit >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            const uint32_t count = (uint32_t)__popcll(peers);
            uint32_t pos = 0;
            if (valid) pos = s_cursor[wave * STR + digit] + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                if (keys_out) keys_out[pos] = k[r];
                if (sizeof(KeyT) == 8 && bm.out) bm.out[pos] = make_uint2(v[r], block_mask_of(bm, digit, (uint32_t)((uint64_t)k[r] >> 32)));
                else vals_out[pos] = v[r];
                if (rank == count - 1) s_cursor[wave * STR + digit] = pos + 1;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// rows of the emission's count table + scatter over its chunks (the histogram was the emission's)
template <typename KeyT, int BITS>
static int tile_sort_emit_chunks(const KeyT *kin, const uint32_t *vin, KeyT *kout, uint32_t *vout,
                                 const uint32_t *chunk_start, uint32_t cap, int shift, uint32_t *cnt, uint32_t *totals,
                                 int nchunks, uint2 *ranges_out, int nranges, uint32_t packed_val_mask, hipStream_t st,
                                 const BlockMaskArgs &bm)
{
    {
        ScopedTimer tm(T_SORT_ROWSCAN, st);
        hipLaunchKernelGGL(sort_rowscan_kernel, dim3(1 << BITS), dim3(256), 0, st, cnt, (const uint32_t *)nullptr, cap, nchunks,
                           totals, nchunks);
    }
    SLS_LAUNCH_CHECK("sort_rowscan_kernel");
    {
        ScopedTimer tm(T_SORT_SCATTER, st);
        const int nblocks = (nchunks + SortBlock<BITS>::kWaves - 1) / SortBlock<BITS>::kWaves;
        hipLaunchKernelGGL((sort_scatter_chunks_kernel<KeyT, BITS>), dim3(nblocks), dim3(64 * SortBlock<BITS>::kWaves), 0, st,
                           kin, vin, kout, vout, chunk_start, cap, shift, (const uint32_t *)cnt, (const uint32_t *)totals,
                           nchunks, ranges_out, nranges, packed_val_mask, bm);
    }
    SLS_LAUNCH_CHECK("sort_scatter_chunks_kernel");
    return SLS_OK;
}

// ---------------------------------------------------------------------------
static size_t sort_core_bytes(uint64_t cap)
{
    const uint64_t nchunks = (cap + kSortWaveItems - 1) / kSortWaveItems;
    return (size_t)(kSortMaxBins * (nchunks ? nchunks : 1) + kSortMaxBins) * sizeof(uint32_t);
}
// (count table + digit totals, and behind them the list of (surfel, block mask) pairs and the 64-bit instances it
//  is sorted from: sort_bmask_buffer, sort_wide_buffer)
size_t sort_scratch_bytes(uint64_t cap)
{
    return sort_core_bytes(cap) + 2 * sizeof(uint64_t) * (size_t)cap + 64;
}
uint2 *sort_bmask_buffer(void *scratch, uint64_t cap)
{
    return (uint2 *)((char *)scratch + ((sort_core_bytes(cap) + 15) & ~(size_t)15));
}
uint64_t *sort_wide_buffer(void *scratch, uint64_t cap)
{
    return (uint64_t *)(sort_bmask_buffer(scratch, cap) + cap);
}

// digit width: as few passes as 11-bit digits allow, then the narrowest digit that still fits
static void sort_plan(int nbits, int &npasses, int &bits)
{
    npasses = (nbits + 10) / 11;
    bits = (nbits + npasses - 1) / npasses;
    if (bits < 8) bits = 8;
}
int sort_passes(int nbits)
{
    int np, b;
    sort_plan(nbits, np, b);
    return np;
}

template <typename KeyT, int BITS>
static int radix_pass(const KeyT *kin, const uint32_t *vin, KeyT *kout, uint32_t *vout, const uint32_t *count_ptr,
                      uint32_t cap, int shift, uint32_t *cnt, uint32_t *totals, int nchunks, int /*unused*/,
                      uint2 *ranges_out, int nranges, uint32_t packed_val_mask, hipStream_t st,
                      const BlockMaskArgs &bm)
{
    const int nblocks = (nchunks + SortBlock<BITS>::kWaves - 1) / SortBlock<BITS>::kWaves;
    {
        ScopedTimer tm(T_SORT_HIST, st);
        hipLaunchKernelGGL((sort_hist_kernel<KeyT, BITS>), dim3(nblocks), dim3(64 * SortBlock<BITS>::kWaves), 0, st, kin,
                           count_ptr, cap, shift, cnt, nchunks);
    }
    SLS_LAUNCH_CHECK("sort_hist_kernel");
    {
        ScopedTimer tm(T_SORT_ROWSCAN, st);
        hipLaunchKernelGGL(sort_rowscan_kernel, dim3(1 << BITS), dim3(256), 0, st, cnt, count_ptr, cap, nchunks, totals, 0);
    }
    SLS_LAUNCH_CHECK("sort_rowscan_kernel");
    {
        ScopedTimer tm(T_SORT_SCATTER, st);
        hipLaunchKernelGGL((sort_scatter_kernel<KeyT, BITS>), dim3(nblocks), dim3(64 * SortBlock<BITS>::kWaves), 0, st, kin, vin, kout, vout,
                           count_ptr, cap, shift, (const uint32_t *)cnt, (const uint32_t *)totals, nchunks, ranges_out,
                           nranges, packed_val_mask, bm);
    }
    SLS_LAUNCH_CHECK("sort_scatter_kernel");
    return SLS_OK;
}

// Stable LSD radix sort of (key, u32 value) pairs on the low `nbits` key bits.
// Item count = min(*count_ptr, cap), read on the device.  Ping-pongs between
// (keys, vals) and (keys_tmp, vals_tmp); *result_in_tmp says where the sorted
// data ended up.  ranges_out (single-pass sorts only): [start, end) of every key
// value < nranges in the sorted output ((0,0) for absent values).  drop_sorted_keys: the
// last pass writes the permuted values only.
// (Counting the next pass's digits inside the scatter / the producer with global
// atomics was measured and is slower than the histogram kernel: +35 us per pass.)
template <typename KeyT>
static int radix_sort_pairs_t(KeyT *keys, uint32_t *vals, KeyT *keys_tmp, uint32_t *vals_tmp,
                              const uint32_t *count_ptr, uint32_t cap, int nbits, void *scratch,
                              size_t scratch_bytes, int *result_in_tmp, hipStream_t st,
                              uint2 *ranges_out = nullptr, int nranges = 0, bool drop_sorted_keys = false,
                              int base_shift = 0, BlockMaskArgs bm = BlockMaskArgs{ nullptr, 1, 1.0f, 1 })
{
    *result_in_tmp = 0;
    if (cap == 0 || nbits <= 0) return SLS_OK;
    if (scratch_bytes < sort_scratch_bytes(cap)) {
        set_error("sort scratch too small: %zu < %zu", scratch_bytes, sort_scratch_bytes(cap));
        return SLS_E_SCRATCH;
    }
    const int nchunks = (int)(((uint64_t)cap + kSortWaveItems - 1) / kSortWaveItems);
    uint32_t *cnt = (uint32_t *)scratch;
    int npasses, bits;
    sort_plan(nbits, npasses, bits);
    if (ranges_out && npasses != 1) {
        set_error("internal: ranges need a single-pass sort");
        return SLS_E_ARG;
    }
    uint32_t *totals = cnt + ((size_t)1 << bits) * nchunks;
    KeyT *kb[2] = { keys, keys_tmp };
    uint32_t *vb[2] = { vals, vals_tmp };
    for (int p = 0; p < npasses; ++p) {
        const int shift = base_shift + bits * p;
        const int src = p & 1, dst = src ^ 1;
        // packed single-pass sort (vals == null): the value travels in the key's low base_shift bits
        const uint32_t packed_val_mask = (vals == nullptr && base_shift > 0) ? ((1u << base_shift) - 1u) : 0u;
        int rc;
        KeyT *kout = (drop_sorted_keys && p + 1 == npasses) ? nullptr : kb[dst];
#define SLS_PASS(B) radix_pass<KeyT, B>(kb[src], vb[src], kout, vb[dst], count_ptr, cap, shift, cnt, totals, nchunks, \
                                        0, ranges_out, nranges, packed_val_mask, st,                           \
                                        (npasses == 1 ? bm : BlockMaskArgs{ nullptr, 1, 1.0f, 1 }))
        switch (bits) {
        case 8: rc = SLS_PASS(8); break;
        case 9: rc = SLS_PASS(9); break;
        case 10: rc = SLS_PASS(10); break;
        default: rc = SLS_PASS(11); break;
        }
#undef SLS_PASS
        if (rc) return rc;
    }
    *result_in_tmp = npasses & 1;
    return SLS_OK;
}

int radix_sort_pairs_u32(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                         const uint32_t *count_ptr, uint32_t cap, int nbits, void *scratch, size_t scratch_bytes,
                         int *result_in_tmp, hipStream_t st)
{
    return radix_sort_pairs_t<uint32_t>(keys, vals, keys_tmp, vals_tmp, count_ptr, cap, nbits, scratch,
                                        scratch_bytes, result_in_tmp, st);
}

// ---------------------------------------------------------------------------
// Temporal re-sort of the depth order.  Between two mapping iterations on the
// same keyframe a surfel moves by at most ~100 positions in the depth order
// (measured in round 2: HISTORY.md), so instead of three radix passes the previous
// permutation is repaired:
//   A. windows of kResortWindow positions of the OLD order are sorted by
//      (new key, surfel index) with a bitonic network in LDS;
//   B. windows shifted by half a window are merged (each is two sorted halves);
//   C. the window edges are compared: the result is a permutation that is sorted
//      inside every shifted window, so strictly increasing edges <=> the exact
//      (key, index) order.  Otherwise bit 1 of the iteration's overflow word is set:
//      the Adam update is skipped on the device and the caller repeats the
//      iteration with the full radix sort (same protocol as a capacity overflow).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kResortThreads) void resort_sort_kernel(int N, const uint32_t *__restrict__ prev_order,
                                                                     const uint32_t *__restrict__ keys_by_surfel,
                                                                     uint64_t *__restrict__ comp)
{
    __shared__ __attribute__((aligned(16))) ulonglong2 s_pairs[kResortThreads];
    resort_sort_window((int)blockIdx.x, N, prev_order, [&](uint32_t g) { return keys_by_surfel[g]; }, comp, s_pairs);
}

// window b covers positions [b*W - W/2, b*W + W/2): second half of sorted window b-1, first half of b
// Also produces level 1 of the scan of tiles_touched (the sums of the four aligned 256-blocks a
// window covers), which saves the gather_block_sums launch.
// PRE (a further repair round in ONE launch): `comp` is sorted inside every SHIFTED window (the previous round wrote the
// merged pairs back, comp_out), so the two aligned windows b-1 and b this window straddles are two sorted halves each:
// the workgroup merges both itself (10 stages each instead of the 55 of a sort; every aligned window is merged by the
// two workgroups that need it), hands them over in LDS and goes on with its own window — the second half of b-1, the
// first of b.  Every round lets a surfel travel another window, at one dependent launch per round.
// comp_out: the merged (key, surfel) pairs by position, for the round that follows (in place where !PRE: a workgroup
// reads and writes its own window only; another buffer where PRE: the neighbours read what this one would overwrite).
template <bool DIRECT, bool PRE>
__global__ __launch_bounds__(kResortThreads) void resort_merge_kernel(int N, const uint64_t *__restrict__ comp,
                                                                      uint32_t *__restrict__ order,
                                                                      uint64_t *__restrict__ edges,
                                                                      const uint32_t *__restrict__ tiles,
                                                                      uint32_t *__restrict__ block_sums, int GX,
                                                                      const uint2 *__restrict__ erec_box, DirectBin db,
                                                                      uint64_t *comp_out)
{
    // DIRECT: instead of level 1 of the scan, step 1 of the direct binning (above): the window is a chunk
    static_assert(kResortWindow == 1024 && kResortThreads == 512, "a 256-block of positions = two waves of pairs");
    __shared__ __attribute__((aligned(16))) ulonglong2 s_pairs[kResortThreads];
    __shared__ uint32_t s_part[4][2];          // [256-block of the window][wave inside it]
    __shared__ uint32_t s_hist[DIRECT ? kDirectMaxBins : 1];
    __shared__ int s_diff[DIRECT ? kDirectMaxBins + 64 : 1];      // per tile row GX + 1 difference counts (count_rect_rows_diff)
    __shared__ int s_wide;
    __shared__ uint64_t s_win[PRE ? 2 * kResortWindow : 1];
    const bool use_diff = DIRECT && GX % 64 == 0 && db.bins % GX == 0 && db.bins / GX <= 64;
    if (DIRECT) {
        for (int d = threadIdx.x; d < db.bins; d += kResortThreads) s_hist[d] = 0u;   // (the network's barriers come before its use)
        for (int d = threadIdx.x; d < kDirectMaxBins + 64; d += kResortThreads) s_diff[d] = 0;
        if (threadIdx.x == 0) s_wide = 0;
    }
    const int base = blockIdx.x * kResortWindow - kResortWindow / 2, o0 = 2 * (int)threadIdx.x;
    SLS_MT(0);
    uint64_t e[2];
    if (PRE) {
        uint64_t f[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int abase = ((int)blockIdx.x - 1 + a) * kResortWindow;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pos = abase + bitonic_src(o0 + q);
                const uint64_t c = comp[min(max(pos, 0), N - 1)];
                f[a][q] = pos < 0 ? 0ull : (pos < N ? c : ~0ull);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            bitonic_pairs<kResortWindow>(f[a][0], f[a][1], s_pairs);
            s_win[a * kResortWindow + o0] = f[a][0];
            s_win[a * kResortWindow + o0 + 1] = f[a][1];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = bitonic_src(o0 + q);        // position inside this (shifted) window
            e[q] = o < kResortWindow / 2 ? s_win[kResortWindow / 2 + o] : s_win[kResortWindow + o - kResortWindow / 2];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = base + bitonic_src(o0 + q);
            const uint64_t c = comp[min(max(pos, 0), N - 1)];
            e[q] = pos < 0 ? 0ull : (pos < N ? c : ~0ull);
        }
    }
    bitonic_pairs<kResortWindow>(e[0], e[1], s_pairs);
    SLS_MT(1);
    if (comp_out) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = base + o0 + q;
            if (pos >= 0 && pos < N) comp_out[pos] = e[q];
        }
    }
    uint32_t v = 0;
    if (DIRECT) {
        uint2 er[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = base + o0 + q;
            const bool real = pos >= 0 && pos < N;
            const uint32_t g = real ? (uint32_t)e[q] : 0u;
            er[q] = erec_box[g];                          // (surfel 0 for the padding: a valid address, masked below)
            if (!real) er[q].x = 0u;
            if (real) order[pos] = g;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = base + o0 + q;
            if (pos >= 0 && pos < N) db.serec[pos] = er[q];
            if (use_diff) count_rect_rows_diff(er[q].x, GX, s_hist, s_diff, &s_wide);
            else count_rect_tiles_wave(er[q].x, GX, s_hist);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = base + o0 + q;
            const bool real = pos >= 0 && pos < N;
            const uint32_t g = real ? (uint32_t)e[q] : 0u;
            const uint32_t tv = tiles[g];                 // (surfel 0 for the padding: a valid address, masked below)
            v += real ? tv : 0u;
            if (real) order[pos] = g;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 7][(threadIdx.x >> 6) & 1] = v;
    }
    // smallest / largest real element of the window (it holds at least one)
    const int lo = base < 0 ? -base : 0, hi = min(kResortWindow, N - base) - 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (o0 + q == lo) edges[2 * blockIdx.x + 0] = e[q];
        if (o0 + q == hi) edges[2 * blockIdx.x + 1] = e[q];
    }
    __syncthreads();
    SLS_MT(2);
    if (DIRECT && use_diff && s_wide) {          // (workgroup-uniform: read behind the barrier)
        diff_finish(GX, db.bins, s_hist, s_diff, (int)threadIdx.x, kResortThreads);
        __syncthreads();
    }
    if (DIRECT) {
        for (int d = threadIdx.x; d < db.bins; d += kResortThreads) {
            const uint32_t c = s_hist[d];
            db.cnt[(size_t)d * db.stride + blockIdx.x] = c;
            if (db.coarse && c) atomicAdd(&db.coarse[(size_t)(blockIdx.x / kDirectGroup) * db.bins + d], c);
        }
        SLS_MT(3);
    } else if (threadIdx.x < 4) {
        const int blk = (base + (int)threadIdx.x * 256) / 256;      // aligned 256-block of positions
        if (base + (int)threadIdx.x * 256 >= 0 && blk * 256 < N)
            block_sums[blk] = s_part[threadIdx.x][0] + s_part[threadIdx.x][1];
    }
}

// step 1 of the direct binning after a from-scratch depth sort (and in the staged API): chunks of 1024 positions
__global__ __launch_bounds__(kDirectChunk) void gather_count_kernel(int N, int GX, const uint32_t *__restrict__ order,
                                                                    const uint2 *__restrict__ erec_box,
                                                                    const int4 *__restrict__ rect,
                                                                    const uint32_t *__restrict__ sbox, DirectBin db)
{
    __shared__ uint32_t s_hist[kDirectMaxBins];
    __shared__ int s_diff[kDirectMaxBins + 64];
    __shared__ int s_wide;
    const bool use_diff = GX % 64 == 0 && db.bins % GX == 0 && db.bins / GX <= 64;
    if ((int)threadIdx.x < db.bins) s_hist[threadIdx.x] = 0u;
    for (int d = threadIdx.x; d < kDirectMaxBins + 64; d += kDirectChunk) s_diff[d] = 0;
    if (threadIdx.x == 0) s_wide = 0;
    __syncthreads();
    const int pos = blockIdx.x * kDirectChunk + (int)threadIdx.x;
    uint2 er = make_uint2(0u, 0u);
    if (pos < N) {
        er = load_emit_record(erec_box, rect, sbox, order[pos]);
        if (db.serec) db.serec[pos] = er;
    }
    if (use_diff) count_rect_rows_diff(er.x, GX, s_hist, s_diff, &s_wide);
    else count_rect_tiles_wave(er.x, GX, s_hist);
    __syncthreads();
    if (use_diff && s_wide) {
        diff_finish(GX, db.bins, s_hist, s_diff, (int)threadIdx.x, kDirectChunk);
        __syncthreads();
    }
    if ((int)threadIdx.x < db.bins) {
        const uint32_t c = s_hist[threadIdx.x];
        db.cnt[(size_t)threadIdx.x * db.stride + blockIdx.x] = c;
        if (db.coarse && c) atomicAdd(&db.coarse[(size_t)(blockIdx.x / kDirectGroup) * db.bins + threadIdx.x], c);
    }
}

// step C, run by block 0 of the scan's first kernel
__device__ __forceinline__ void resort_verify(int nwin, const uint64_t *__restrict__ edges, uint32_t *__restrict__ flag)
{
    for (int b = threadIdx.x; b + 1 < nwin; b += 256)
        if (edges[2 * b + 1] >= edges[2 * (b + 1)]) atomicOr(flag, kResortFailed);
}

// inclusive max-scan over the 64 lanes of a wave (DPP: shifts inside the rows of 16, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_max_scan(uint32_t v)
{
#define SLS_MAXDPP(ctrl_, rows_) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl_, rows_, 0xF, false))
    SLS_MAXDPP(0x111, 0xF);     // row_shr:1   (lanes without a source keep their own value: the identity of max)
    SLS_MAXDPP(0x112, 0xF);     // row_shr:2
    SLS_MAXDPP(0x114, 0xF);     // row_shr:4
    SLS_MAXDPP(0x118, 0xF);     // row_shr:8
    SLS_MAXDPP(0x142, 0xA);     // row_bcast:15 into rows 1, 3
    SLS_MAXDPP(0x143, 0xC);     // row_bcast:31 into rows 2, 3
#undef SLS_MAXDPP
    return v;
}

// step 3 of the direct binning.  A chunk of 1024 depth positions is served by SPLIT workgroups of 1024 / SPLIT threads:
// the chunks at the front of the depth order hold the near surfels — ten and more tiles each, 10-15 k instances per
// chunk against 1.5 k at the far end — and a launch lasts as long as its heaviest workgroup.  Workgroup `sub` of a chunk
// first counts the rectangles of the sub-chunks in front of it (LDS counts only: a tenth of what emitting them costs)
// to know where its own instances start.
// A workgroup one of whose waves holds kHeavyWave instances or more (near surfels: rectangles of a hundred tiles and more,
// all at the front of the depth order — 14 k instances in ONE wave at the bench window's last keyframes, 222 rounds where
// the launch's median wave has two: profiles/r05a_bin_tail.txt) leaves the wave-by-wave rounds: see "heavy" below.
constexpr uint32_t kHeavyWave = 1536;
constexpr int kHeavyParts = 4, kHeavyChunks = 16;
template <int BITS, bool PAIRS, int SPLIT>
__global__ __launch_bounds__(kDirectChunk / SPLIT) void bin_direct_kernel(int N, int GX, DirectBin db,
                                                                  const uint32_t *__restrict__ order,
                                                                  const uint2 *__restrict__ erec_box,
                                                                  const int4 *__restrict__ rect,
                                                                  const uint32_t *__restrict__ sbox, uint32_t cap,
                                                                  uint32_t *__restrict__ vals_out, BlockMaskArgs bm,
                                                                  uint2 *__restrict__ ranges_out, int nranges,
                                                                  uint32_t *__restrict__ total_out,
                                                                  uint32_t *__restrict__ overflow, int resort_windows,
                                                                  const uint64_t *__restrict__ resort_edges,
                                                                  uint32_t *__restrict__ fail_flag,
                                                                  uint32_t *__restrict__ status_mirror)
{
    constexpr int BINS = 1 << BITS, TPB = kDirectChunk / SPLIT, WAVES = TPB / 64;
    constexpr int PER = BINS > TPB ? BINS / TPB : 1;          // tiles per thread in the per-tile steps
    static_assert(BINS <= kDirectMaxBins && TPB % 64 == 0 && (BINS % TPB == 0 || BINS < TPB), "tiles dealt evenly to the threads");
    __shared__ uint32_t s_cur[WAVES * BINS];     // per wave and tile: first the instance counts, then the running cursors
    __shared__ uint32_t s_pre[BINS];             // per tile: instances of the chunk's sub-chunks in front of this one
    __shared__ uint4 s_lane[WAVES][64];          // per wave and lane: {rectangle, block box, surfel, first instance of the lane}
    __shared__ uint32_t s_mark[WAVES][64];       // per wave: which lane's instances start at each slot of the current round
    __shared__ uint32_t s_part[WAVES];
    __shared__ uint32_t s_ws[WAVES];             // the waves' instance counts (which workgroup is "heavy")
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Workgroups in front of the grid's chunks (SPLIT = 1): HELPERS of the first kHeavyChunks chunks — where the near
    // surfels sit — kHeavyParts - 1 each.  A helper looks at its chunk; if that is not heavy (almost always) it leaves at
    // once, else it takes one quarter of the chunk's tiles (pass `part` below) while the chunk's own workgroup takes
    // quarter 0: a heavy chunk's 70 k scattered 8-byte stores are what ONE compute unit's address pipeline takes 35 us for
    // (measured with the stores compiled out), its instances' arithmetic as much again — four compute units share both.
    const int nhelp = SPLIT == 1 ? min(kHeavyChunks, db.nchunks) * (kHeavyParts - 1) : 0;
    const bool helper = (int)blockIdx.x < nhelp;
    const int chunk = helper ? (int)blockIdx.x / (kHeavyParts - 1) : ((int)blockIdx.x - nhelp) / SPLIT;
    const int sub = helper ? 0 : ((int)blockIdx.x - nhelp) % SPLIT;
    const int part = helper ? 1 + (int)blockIdx.x % (kHeavyParts - 1) : 0;
    const bool lead = !helper && chunk == 0 && sub == 0;      // the launch's first chunk: ranges, R, the void bits, the status mirror
    SLS_BT(0);
    if (resort_windows > 0 && lead) {
        for (int b = tid; b + 1 < resort_windows; b += TPB)
            if (resort_edges[2 * b + 1] >= resort_edges[2 * (b + 1)]) atomicOr(fail_flag, kResortFailed);
    }
    // everything the workgroup needs from memory is requested up front: its positions' records (and the rectangles of
    // the sub-chunks in front), its column of the count table, the tile totals
    const int pos0 = db.pos0 + chunk * kDirectChunk;
    const int pos = pos0 + sub * TPB + tid;
    const bool real = pos >= 0 && pos < N;
    uint2 er = make_uint2(0u, 0u);
    uint32_t g = 0u;
    if (real) {
        g = order[pos];
        er = db.serec ? db.serec[pos] : load_emit_record(erec_box, rect, sbox, g);
    }
    uint32_t front[SPLIT > 1 ? SPLIT - 1 : 1];
#pragma unroll
    for (int m = 0; m < SPLIT - 1; ++m) {
        const int p2 = pos0 + m * TPB + tid;
        front[m] = 0u;
        if (m < sub && p2 >= 0 && p2 < N) front[m] = db.serec ? db.serec[p2].x : load_emit_record(erec_box, rect, sbox, order[p2]).x;
    }
    uint32_t tot[PER], ccol[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int d = tid * PER + q;
        tot[q] = 0u; ccol[q] = 0u;
        if (d < BINS) {
            if (db.coarse) {
                // no row scan ran: the tile's total = the sum of its groups, what lies in front of this chunk = the groups
                // in front + the chunks of its own group in front (raw counts; all loads independent)
                const int ngroups = (db.nchunks + kDirectGroup - 1) / kDirectGroup, g0 = chunk / kDirectGroup;
                uint32_t all = 0u, front_g = 0u;
                // (sixteen loads in flight at a time: written as a plain loop the compiler waits for every pair of
                //  them — sixteen L2 round trips one after the other at BASELINE config 3, 5 us of the launch)
                for (int gb = 0; gb < ngroups; gb += 16) {
                    uint32_t v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {     // (uniform row, lane offset d: one address register for all)
                        const int grow = __builtin_amdgcn_readfirstlane(min(gb + k, ngroups - 1));
                        v[k] = (db.coarse + (size_t)grow * BINS)[d];
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const uint32_t w = gb + k < ngroups ? v[k] : 0u;
                        all += w;
                        front_g += gb + k < g0 ? w : 0u;
                    }
                }
                // (the group's counts of this tile: one aligned 64-byte line, four 16-byte loads)
                const uint4 *row = reinterpret_cast<const uint4 *>(db.cnt + (size_t)d * db.stride + g0 * kDirectGroup);
                const int nin = chunk - g0 * kDirectGroup;          // chunks of the own group in front: 0..15
                uint32_t front_c = 0u;
#pragma unroll
                for (int c = 0; c < kDirectGroup / 4; ++c) {
                    const uint4 v = row[c];
                    front_c += (4 * c < nin ? v.x : 0u) + (4 * c + 1 < nin ? v.y : 0u) + (4 * c + 2 < nin ? v.z : 0u) + (4 * c + 3 < nin ? v.w : 0u);
                }
                tot[q] = all;
                ccol[q] = front_g + front_c;
            } else {
                tot[q] = db.totals[d];
                ccol[q] = db.cnt[(size_t)d * db.stride + chunk];
            }
        }
    }
    for (int i = tid; i < WAVES * BINS; i += TPB) s_cur[i] = 0u;
    for (int i = tid; i < BINS; i += TPB) s_pre[i] = 0u;
    const uint32_t t = ((er.x >> 9) & 1023u) * (er.x >> 25);
    uint32_t incl = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    const uint32_t S = (uint32_t)__shfl((int)incl, 63, 64);
    const uint32_t first = incl - t;
    s_lane[w][lane] = make_uint4(er.x, er.y, g, first);
    if (lane == 63) s_ws[w] = S;
    __syncthreads();
    SLS_BT(1);
    bool heavy = false;                          // (workgroup-uniform)
    if (SPLIT == 1) {
#pragma unroll
        for (int k = 0; k < WAVES; ++k) heavy = heavy || s_ws[k] >= kHeavyWave;
    }
    if (helper && !heavy) return;                // (workgroup-uniform: nothing to help with)
    // per-wave counts (the order inside a wave does not matter for counting: every lane walks its own rectangle)
    if (!heavy) count_rect_tiles_wave(er.x, GX, s_cur + w * BINS);
#pragma unroll
    for (int m = 0; m < SPLIT - 1; ++m) count_rect_tiles_wave(front[m], GX, s_pre);
    SLS_BT(2);
    // digit bases: exclusive scan of the tile totals (PER consecutive ones per thread)
    uint32_t loc[PER], dsum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = dsum; dsum += tot[q]; }
    uint32_t dinc = dsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(dinc, off, 64);
        if (lane >= off) dinc += u;
    }
    if (lane == 63) s_part[w] = dinc;
    __syncthreads();
    {
        uint32_t wp = 0;
        for (int k = 0; k < w; ++k) wp += s_part[k];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int d = tid * PER + q;
            if (d >= BINS) break;
            const uint32_t dbase = wp + dinc - dsum + loc[q];
            if (lead) {
                // the digit bases ARE the tile ranges (A5), clipped to the buffers' capacity
                if (d < nranges) ranges_out[d] = tot[q] ? make_uint2(min(dbase, cap), min(dbase + tot[q], cap)) : make_uint2(0u, 0u);
                if (d == BINS - 1) {
                    const uint32_t R = dbase + tot[q];
                    if (total_out) *total_out = R;
                    if (R > cap && overflow) atomicOr(overflow, 1u);     // too small: flagged, every slot below cap still filled
                }
            }
            if (heavy) { s_pre[d] = dbase + ccol[q]; continue; }      // (the tile's first slot for this chunk; no per-wave cursors)
            uint32_t c[WAVES];
#pragma unroll
            for (int k = 0; k < WAVES; ++k) c[k] = s_cur[k * BINS + d];
            uint32_t run = dbase + ccol[q] + s_pre[d];
#pragma unroll
            for (int k = 0; k < WAVES; ++k) { s_cur[k * BINS + d] = run; run += c[k]; }
        }
    }
    __syncthreads();
    // The drop-in forward (sls_forward_ws): workgroup 0 has just published R and — its threads alone — the two void bits
    // (capacity too small, repaired order inexact): the status block goes to the caller's pinned host mirror HERE, a
    // few microseconds into the launch, and the host knows whether the forward stands while the binning and the tile
    // forward are still running.  (total_out = word 0 of the block; the launch's other workgroups never write it.)
    // INVARIANT this snapshot rests on (words 2..6 go out as zeros): on the drop-in path every void bit is raised IN FRONT
    // of this point — bit 0 three lines up by this workgroup, bit 1 by the counting merge, a launch earlier — and nothing
    // behind it (the other workgroups of this launch, the tile forward) writes the status block.  A check added later
    // must either sit in front of it or be read from the device's block (rasterize_forward_ws re-reads that block under
    // settings.debug and raises on a difference).
    if (status_mirror && lead && tid == 0) {
        __threadfence();
        // (words written a moment ago by other threads of this workgroup: read where atomics live, not through this CU's L1)
        const uint32_t w0 = __hip_atomic_load(total_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t w1 = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_nontemporal_store(w0, status_mirror + 0);
        __builtin_nontemporal_store(w1, status_mirror + 1);
#pragma unroll
        for (int k = 2; k < 7; ++k) __builtin_nontemporal_store(0u, status_mirror + k);
        __threadfence_system();
        __builtin_nontemporal_store(0u, status_mirror + 7);      // (the host polls words 0 and 7: sls_common.hpp, mirror_status_block)
    }
    SLS_BT(3);
    SLS_BTV(5, (S + 63u) / 64u);
    SLS_BTV(6, S);
    // The wave's S instances, 64 at a time in emission order (lane-major, then the rectangle row-major).  Owner of slot q:
    // the last lane whose first instance is <= q — every lane marks the slot its instances start at, a max-scan over the
    // round's 64 slots (carried on from the previous round) names the owner: one LDS round trip and six DPP steps.
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (SPLIT == 1 && heavy) {
        // ---- heavy workgroup: every INSTANCE of the chunk goes to a thread, whatever surfel it belongs to.
        // An instance's place is   first slot of its tile for this chunk + set bits of the tile's membership mask below
        // its depth position,   the mask M[wave][tile] = one bit per lane of the wave whose rectangle holds the tile.  1024
        // positions x 512 tiles are 64 KB of masks; taken a QUARTER of the tiles at a time (16 KB + 8 KB of per-wave
        // prefixes) they fit into the cursor table the rounds would have used.  Four passes; in each the pass's instances
        // are dealt out in equal contiguous shares, one per thread (the owner of a share's first instance by binary
        // search in the prefix of the positions' counts, then rectangles are walked and owners advanced), twice:
        //   once to set the masks' bits (LDS atomics: neighbouring instances are neighbouring tiles, different words),
        //   once — behind the per-tile prefix over the waves' bit counts — to place and store them.
        // No step depends on a cursor another step wrote, and no thread has more than S / 1024 + 1 instances per walk.
        // (Masks by ballot instead — every wave asks "does my rectangle hold tile t" for the pass's 128 tiles — cost 8 k
        //  instructions per wave whatever the instance count: 99 us at keyframe 5 where the rounds took 37.)
        constexpr int QT = BINS / 4;                                   // tiles per pass
        uint64_t *const Mq = reinterpret_cast<uint64_t *>(s_cur);     // [WAVES][QT]
        uint32_t *const pre = s_cur + 2 * WAVES * QT;                 // [WAVES][QT]: set bits of the tile in the waves in front
        uint32_t *const cpre = &s_mark[0][0];                         // [TPB]: inclusive prefix of the positions' counts in the pass
        uint32_t *const wsum = s_part;                                // [WAVES]
        static_assert(3 * WAVES * QT <= WAVES * BINS, "the heavy passes' masks and prefixes live in the cursor table");
        const int onc = (int)((er.x >> 9) & 1023u), oty = (int)((er.x >> 19) & 63u), onr = (int)(er.x >> 25);
        static_assert(kHeavyParts == 4, "a helper per quarter of the tiles");
        const bool shared = chunk < kHeavyChunks;     // (this chunk has helpers: its own workgroup takes quarter 0 only)
        for (int qp = shared ? part : 0; qp < (shared ? part + 1 : 4); ++qp) {
            const int t_lo = qp * QT, t_hi = min(t_lo + QT, nranges);    // tiles of the pass; tile rows that can hold one:
            if (t_lo >= t_hi) break;
            const int ra = t_lo / GX, rb = (t_hi - 1) / GX;
            const int y0 = max(ra, oty), y1 = min(rb, oty + onr - 1);  // rows of MY rectangle among them
            const uint32_t cq = (onc > 0 && y1 >= y0) ? (uint32_t)(onc * (y1 - y0 + 1)) : 0u;
            for (int i = tid; i < 2 * WAVES * QT; i += TPB) s_cur[i] = 0u;
            uint32_t inc2 = cq;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(inc2, off, 64);
                if (lane >= off) inc2 += u;
            }
            if (lane == 63) wsum[w] = inc2;
            __syncthreads();
            uint32_t wp2 = 0, Sq = 0;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) { const uint32_t v = wsum[k]; wp2 += k < w ? v : 0u; Sq += v; }
            cpre[tid] = wp2 + inc2;
            __syncthreads();
            const uint32_t share = (Sq + (uint32_t)TPB - 1u) / (uint32_t)TPB;
            const uint32_t i_begin = (uint32_t)tid * share, i_end = min(Sq, i_begin + share);
            // fn(position, tile, record) for every instance of my share
            auto walk = [&](auto fn) {
                if (i_begin >= i_end) return;
                int lo = 0, hi = TPB - 1;                              // the first position whose inclusive prefix exceeds i_begin
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (cpre[mid] > i_begin) hi = mid; else lo = mid + 1; }
                uint32_t upto = cpre[lo];
                uint4 o = s_lane[lo >> 6][lo & 63];                   // {rectangle, block box, surfel, -}
                int ptx = (int)(o.x & 511u), pnc = (int)((o.x >> 9) & 1023u);
                const uint32_t k = i_begin - (lo > 0 ? cpre[lo - 1] : 0u);      // my first instance: its k-th among the pass's rows
                const uint32_t ky = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)pnc));     // (k / ncols: see the rounds)
                int kx = (int)(k - ky * (uint32_t)pnc);
                int ty = max(ra, (int)((o.x >> 19) & 63u)) + (int)ky;
                int tx = ptx + kx;
                if (tx >= GX) tx -= GX;
                // ONE loop over the share (the same trip count in every lane); a lane whose surfel is used up moves on to
                // the next position that has instances in this pass and starts its rows from the left
                for (uint32_t i = i_begin; i < i_end; ++i) {
                    if (i >= upto) {
                        do { ++lo; upto = cpre[lo]; } while (upto <= i);
                        o = s_lane[lo >> 6][lo & 63];
                        ptx = (int)(o.x & 511u); pnc = (int)((o.x >> 9) & 1023u);
                        kx = 0; ty = max(ra, (int)((o.x >> 19) & 63u)); tx = ptx;
                    }
                    const int tile = ty * GX + tx;
                    if (tile >= t_lo && tile < t_hi) fn(lo, tile, o);     // (a tile row may straddle two passes)
                    ++kx; ++tx;
                    if (tx == GX) tx = 0;
                    if (kx == pnc) { kx = 0; ++ty; tx = ptx; }
                }
            };
            walk([&](int pp, int tile, const uint4 &) {
                atomicOr((unsigned long long *)&Mq[(pp >> 6) * QT + (tile - t_lo)], 1ull << (pp & 63));
            });
            __syncthreads();
            if (tid < t_hi - t_lo) {
                uint32_t run = 0;
#pragma unroll
                for (int k = 0; k < WAVES; ++k) { pre[k * QT + tid] = run; run += (uint32_t)__popcll(Mq[k * QT + tid]); }
            }
            __syncthreads();
            walk([&](int pp, int tile, const uint4 &o) {
                const int tl = tile - t_lo, pw = pp >> 6;
                const uint32_t place = s_pre[tile] + pre[pw * QT + tl] + (uint32_t)__popcll(Mq[pw * QT + tl] & ((1ull << (pp & 63)) - 1ull));
                if (place < cap) {
                    if (PAIRS) bm.out[place] = make_uint2(o.z, block_mask_of(bm, (uint32_t)tile, o.y));
                    else vals_out[place] = o.z;
                }
            });
            __syncthreads();
        }
        SLS_BT(4);
        return;
    }
    uint32_t *const cur = s_cur + w * BINS;
    uint32_t carry = 0u;                          // (owner of the slot before this round) + 1
    const bool long_wave = S >= 1024u;            // (a wave of far surfels never meets the case below: spare it the test)
    for (uint32_t q0 = 0; q0 < S; q0 += 64u) {
        if (long_wave && carry) {
            // Rounds whose 64 slots all belong to the surfel the previous round ended in — most rounds of a rectangle of
            // hundreds of tiles — need no owner search and no ranking: consecutive tiles of ONE rectangle are different
            // tiles, each takes its tile's cursor as it stands.  And since they are different tiles, four such rounds
            // read their cursors together before any of them is written back: one LDS round trip per four rounds.
            // (The near end of the depth order at the bench scene's last keyframes: 222 rounds in the heaviest wave, 9
            //  elsewhere — profiles/r05a_bin_tail.txt.)
            const uint4 oc = s_lane[w][carry - 1u];
            const uint32_t ncc = (oc.x >> 9) & 1023u, end = oc.w + ncc * (oc.x >> 25);
            if (end >= q0 + 64u) {
                const float inc = __builtin_amdgcn_rcpf((float)ncc);
                const int otx = (int)(oc.x & 511u), oty = (int)((oc.x >> 19) & 63u);
                auto tile_at = [&](uint32_t k) -> uint32_t {
                    const uint32_t ky = (uint32_t)(((float)k + 0.5f) * inc);
                    int tx = otx + (int)(k - ky * ncc);
                    if (tx >= GX) tx -= GX;
                    return (uint32_t)((oty + (int)ky) * GX + tx);
                };
                auto put = [&](uint32_t p, uint32_t tile) {
                    if (p < cap) {
                        if (PAIRS) bm.out[p] = make_uint2(oc.z, block_mask_of(bm, tile, oc.y));
                        else vals_out[p] = oc.z;
                    }
                };
                uint32_t k = q0 + (uint32_t)lane - oc.w;
                for (; q0 + 256u <= end; q0 += 256u, k += 256u) {
                    const uint32_t t0 = tile_at(k), t1 = tile_at(k + 64u), t2 = tile_at(k + 128u), t3 = tile_at(k + 192u);
                    const uint32_t p0 = cur[t0], p1 = cur[t1], p2 = cur[t2], p3 = cur[t3];
                    cur[t0] = p0 + 1u; cur[t1] = p1 + 1u; cur[t2] = p2 + 1u; cur[t3] = p3 + 1u;
                    put(p0, t0); put(p1, t1); put(p2, t2); put(p3, t3);
                }
                for (; q0 + 64u <= end; q0 += 64u, k += 64u) {
                    const uint32_t t0 = tile_at(k);
                    const uint32_t p0 = cur[t0];
                    cur[t0] = p0 + 1u;
                    put(p0, t0);
                }
                __builtin_amdgcn_wave_barrier();
                q0 -= 64u;           // (the loop's own increment follows)
                continue;
            }
        }
        s_mark[w][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (t != 0u && first - q0 < 64u) s_mark[w][first - q0] = (uint32_t)lane + 1u;     // (unsigned: first >= q0 too)
        __builtin_amdgcn_wave_barrier();
        const uint32_t own1 = max(wave_max_scan(s_mark[w][lane]), carry);
        carry = (uint32_t)__builtin_amdgcn_readlane((int)own1, 63);
        const uint32_t q = q0 + (uint32_t)lane;
        const bool valid = q < S;
        const uint4 o = s_lane[w][valid ? own1 - 1u : 0u];     // {rectangle, block box, surfel, first}
        const uint32_t k = q - o.w;
        const uint32_t onc = (o.x >> 9) & 1023u;
        // k / ncols through the reciprocal: (k + 0.5) / ncols lies at least 1 / (2 ncols) >= 2^-11 from an integer and is
        // below nrows <= 2^7 (k < ncols nrows), so the float error (rcp 1 ulp + one rounding) stays below 2^-15
        const uint32_t ky = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)onc));
        const uint32_t kx = k - ky * onc;
        int tx = (int)(o.x & 511u) + (int)kx;
        if (tx >= GX) tx -= GX;
        const uint32_t tile = valid ? (uint32_t)(((int)((o.x >> 19) & 63u) + (int)ky) * GX + tx) : 0u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const bool bit = (tile >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t count = (uint32_t)__popcll(peers);
        uint32_t p = 0;
        if (valid) p = cur[tile] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            if (p < cap) {
                if (PAIRS) bm.out[p] = make_uint2(o.z, block_mask_of(bm, tile, o.y));
                else vals_out[p] = o.z;
            }
            if (rank == count - 1) cur[tile] = p + 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    SLS_BT(4);
}

// A2 on the depth-ordered surfels, level 1: per-block sums of tiles[order[i]]
__global__ __launch_bounds__(256) void gather_block_sums_kernel(int N, const uint32_t *__restrict__ order,
                                                                const uint32_t *__restrict__ tiles,
                                                                uint32_t *__restrict__ block_sums)
{
    __shared__ uint32_t s_part[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t v = (i < N) ? tiles[order[i]] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// level 2: inclusive scan inside each 256-surfel block (depth order) on top of the
// sum of the preceding blocks' totals, which every block adds up itself (a few KB
// of L2 reads; no separate single-block scan launch).  The last block publishes R.
// offsets[] is indexed by DEPTH-ORDER position.
__global__ __launch_bounds__(256) void gather_scan_final_kernel(int N, const uint32_t *__restrict__ order,
                                                                const uint32_t *__restrict__ tiles,
                                                                const uint32_t *__restrict__ block_sums,
                                                                uint32_t *__restrict__ offsets,
                                                                uint32_t *__restrict__ total_out, int resort_windows,
                                                                const uint64_t *__restrict__ resort_edges,
                                                                uint32_t *__restrict__ fail_flag)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_pre[4];
    if (resort_windows > 0 && blockIdx.x == 0) resort_verify(resort_windows, resort_edges, fail_flag);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) pre += block_sums[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    const uint32_t v = (i < N) ? tiles[order[i]] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    if (lane == 0) s_pre[wave] = pre;
    __syncthreads();
    uint32_t wave_prefix = 0;
    for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
    const uint32_t block_prefix = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
    if (i < N) offsets[i] = block_prefix + wave_prefix + incl;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total_out = block_prefix + wave_prefix + incl;
}

// ---------------------------------------------------------------------------
// A3: thread i handles the surfel at depth-order position i and walks its tile
// rectangle (row-major: y outer, x inner, x wrapping modulo the grid width in
// 360-degree mode, D5/D9).  Instances beyond `cap` are dropped and flagged.
// ---------------------------------------------------------------------------
template <int EB>
__global__ __launch_bounds__(EB) void emit_tiles_kernel(int N, int GX, const uint32_t *__restrict__ order,
                                                         const int4 *__restrict__ rect,
                                                         const uint32_t *__restrict__ tiles,
                                                         const uint64_t *__restrict__ tile_mask,
                                                         const int4 *__restrict__ erec,
                                                         const uint32_t *__restrict__ offsets, uint32_t cap,
                                                         uint32_t *__restrict__ tkeys, uint32_t *__restrict__ vals,
                                                         uint32_t *__restrict__ overflow, int pack_shift,
                                                         ScanHandoff fused, uint32_t *__restrict__ total_out,
                                                         uint32_t *__restrict__ fail_flag,
                                                         uint32_t *__restrict__ hist_cnt, int hist_bins,
                                                         uint32_t *__restrict__ chunk_start,
                                                         const uint32_t *__restrict__ sbox, uint64_t *__restrict__ wide)
{
    // wide != null (packed mode): an instance is 64 bits — the surfel's block box above the (tile, surfel) word
    // hist_cnt != null: the workgroup is a CHUNK of the tile sort that follows — it counts the tiles of the instances
    // it writes (LDS) and stores column blockIdx of the count table cnt[tile][chunk] and the chunk's first instance:
    // the sort needs no histogram launch (sort_scatter_chunks_kernel)
    __shared__ uint32_t s_hist[kSortMaxBins];
    if (hist_cnt) {
        for (int d = threadIdx.x; d < hist_bins; d += EB) s_hist[d] = 0u;
        __syncthreads();
    }
    // pack_shift > 0 (vals == null): one word per instance, (tile << pack_shift) | surfel
    constexpr int EW = EB / 64;                // waves of the workgroup = of a chunk of EB depth positions
    const int i = blockIdx.x * EB + threadIdx.x;
    // ONE dependent round trip per surfel: the rectangle (all zeros for a culled surfel) and the mask of its tiles
    // that the footprint can reach (D10; null: the whole rectangle) give the tile count
    // (erec: the same two things packed by preprocess into one 16-byte word)
    const uint32_t g = order[min(i, N - 1)];
    int4 rc;
    uint64_t mask;
    if (erec) {
        const int4 e = erec[g];
        rc = make_int4(e.x & 0xFFFF, e.y & 0xFFFF, (int)((uint32_t)e.x >> 16), (int)((uint32_t)e.y >> 16));
        mask = ((uint64_t)(uint32_t)e.w << 32) | (uint64_t)(uint32_t)e.z;
    } else {
        rc = rect[g];
        mask = (tile_mask && (uint32_t)(rc.y * rc.w) <= 64u) ? tile_mask[g] : ~0ull;
    }
    const uint64_t box_hi = wide ? ((uint64_t)sbox[g] << 32) : 0ull;
    const uint32_t nrect = (uint32_t)(rc.y * rc.w);
    if (nrect > 64u) mask = ~0ull;
    const uint32_t t = (i < N) ? (nrect <= 64u ? (uint32_t)__popcll(mask & (nrect >= 64u ? ~0ull : ((1ull << nrect) - 1ull))) : nrect) : 0u;
    uint32_t end;
    if (fused.block_sums) {
        // level 2 of the scan of tiles_touched done here (no scan launch, no offsets array): prefix of the
        // preceding 256-blocks' sums + inclusive scan inside the block; the last block publishes R
        __shared__ uint32_t s_wave[EW];
        __shared__ uint32_t s_pre[EW];
        if (fused.resort_windows > 0 && blockIdx.x == 0) resort_verify(fused.resort_windows, fused.resort_edges, fail_flag);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t pre = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x * (EB / 256); b += EB) pre += fused.block_sums[b];   // (sums of 256 positions)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
        uint32_t incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        if (lane == 63) s_wave[wave] = incl;
        if (lane == 0) s_pre[wave] = pre;
        __syncthreads();
        uint32_t wave_prefix = 0;
        for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
        uint32_t block_prefix = 0;
#pragma unroll
        for (int w = 0; w < EW; ++w) block_prefix += s_pre[w];
        end = block_prefix + wave_prefix + incl;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == EB - 1) *total_out = end;
    } else {
        end = (i < N) ? offsets[i] : 0u;
    }
    if (chunk_start && threadIdx.x == 0) {
        chunk_start[blockIdx.x] = end - t;                      // (thread 0 of a launched block is a surfel: i < N)
        if (blockIdx.x == gridDim.x - 1 && !fused.block_sums) chunk_start[gridDim.x] = offsets[N - 1];
    }
    if (chunk_start && fused.block_sums && blockIdx.x == gridDim.x - 1 && threadIdx.x == EB - 1) chunk_start[gridDim.x] = end;
    bool emit = i < N && t != 0u;
    uint32_t off = end - t;
    if (emit && end > cap) {
        // The buffers are too small: flag it (the caller repeats the iteration with more room) but
        // still fill every slot below cap, so that nothing downstream reads an uninitialised entry.
        if (overflow) atomicOr(overflow, 1u);
        if (off >= cap) emit = false;
    }
    if (emit) {
        uint32_t bit = 0;
        for (int y = 0; y < rc.w; ++y) {
            const uint32_t row = (uint32_t)(rc.z + y) * (uint32_t)GX;
            for (int k = 0; k < rc.y; ++k, ++bit) {
                if (bit < 64u && !((mask >> bit) & 1ull)) continue;      // the footprint cannot reach this tile
                int tx = rc.x + k;
                if (tx >= GX) tx -= GX;
                if (off < cap) {
                    const uint32_t tile = row + (uint32_t)tx;
                    if (wide) wide[off] = box_hi | (uint64_t)((tile << pack_shift) | g);
                    else if (vals) { tkeys[off] = tile; vals[off] = g; }
                    else tkeys[off] = (tile << pack_shift) | g;
                    if (hist_cnt) atomicAdd(&s_hist[tile], 1u);
                }
                ++off;
            }
        }
    }
    if (hist_cnt) {
        __syncthreads();
        for (int d = threadIdx.x; d < hist_bins; d += EB) hist_cnt[(size_t)d * gridDim.x + blockIdx.x] = s_hist[d];
    }
}

// ---------------------------------------------------------------------------
// A5: [start, end) of every tile in the sorted list; optionally materialises
// the 64-bit keys (tile << 32 | depth bits) the list is ordered by.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t *__restrict__ tkeys,
                                                          const uint32_t *__restrict__ vals,
                                                          const uint32_t *__restrict__ count_ptr, uint32_t cap,
                                                          const float *__restrict__ depth, uint2 *__restrict__ ranges,
                                                          uint64_t *__restrict__ keys64)
{
    const uint32_t R = load_count(count_ptr, cap);
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= R) return;
    const uint32_t t = tkeys[j];
    if (j == 0 || tkeys[j - 1] != t) ranges[t].x = j;
    if (j + 1 == R || tkeys[j + 1] != t) ranges[t].y = j + 1;
    if (keys64) keys64[j] = ((uint64_t)t << 32) | (uint64_t)__float_as_uint(depth[vals[j]]);
}

static int bits_for(uint32_t max_value)
{
    int b = 0;
    while (b < 32 && (max_value >> b) != 0) ++b;
    return b;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Depth order of the surfels + scan of tiles_touched in that order.
//   order   : N u32, surfel index at each depth-order position
//   offsets : N u32, inclusive scan of tiles_touched[order[.]]
//   total   : device u32 = R
// scratch: order keys (N) | tmp keys (N) | tmp vals (N) | 2 pad | block sums | N on device | sort scratch
size_t order_scratch_bytes(int N)
{
    const size_t n = (size_t)(N > 0 ? N : 1);
    // (+ the repair's second pair buffer behind its window edges inside the sort's scratch: 8 n + 16 (n / 1024 + 3) bytes)
    return sizeof(uint32_t) * (3 * n + (n + 255) / 256 + 64) + sort_scratch_bytes(n) + 16 + n / 32 + 128;
}

// where preprocess may write the sort input directly (saves the depth_keys launch)
void depth_order_key_buffers(int N, void *scratch, uint32_t *order, uint32_t **keys, uint32_t **vals0,
                             uint32_t **n_dev)
{
    const int nb = (N + 255) / 256;
    *keys = (uint32_t *)scratch;
    // with an odd number of passes the sorted values land in the "other" buffer, so the
    // identity permutation starts in the scratch buffer and the result ends in `order`
    *vals0 = (sort_passes(kDepthKeyBits) & 1) ? *keys + 2 * (size_t)N : order;
    *n_dev = *keys + 3 * (size_t)N + 2 + nb;   // (2 words of padding: see launch_depth_order_scan)
}

// reuse_order != 0: `order` holds the permutation of the previous iteration (same surfels, same
// keyframe) and the keys were written by preprocess (keys_prefilled): temporal re-sort, failure is
// reported in *fail_flag (see above).
// where the repair keeps its (key, surfel) pairs: N u64 over the sort's two temporary arrays
uint64_t *resort_comp_buffer(int N, void *scratch)
{
    uint32_t *keys_tmp = (uint32_t *)scratch + N;
    return (uint64_t *)(((uintptr_t)keys_tmp + 7) & ~(uintptr_t)7);
}

// window_sort_done: step A of the repair (resort_sort) already ran — merged into the preprocess launch
int launch_depth_order_scan(int N, const float *depth, const uint32_t *tiles, uint32_t *order, uint32_t *offsets,
                            uint32_t *total_out, void *scratch, size_t scratch_bytes, int keys_prefilled,
                            hipStream_t st, int reuse_order, uint32_t *fail_flag, ScanHandoff *handoff,
                            bool window_sort_done, const DirectBin *direct, const int4 *erec_box, int GX)
{
    // direct (+ handoff): the direct binning follows — the repair's last merge fills the count table (handoff->counted),
    // no scan of tiles_touched is needed at all
    if (scratch_bytes < order_scratch_bytes(N)) {
        set_error("depth-order scratch too small: %zu < %zu", scratch_bytes, order_scratch_bytes(N));
        return SLS_E_SCRATCH;
    }
    const int nb = (N + 255) / 256;
    uint32_t *keys = (uint32_t *)scratch;
    uint32_t *keys_tmp = keys + N;
    uint32_t *vals_tmp = keys_tmp + N;
    uint32_t *block_sums = vals_tmp + N + 2;    // padding: the 8-byte aligned u64 view of the temporaries may end one word late
    uint32_t *n_dev = block_sums + nb;          // device copy of N for the count_ptr protocol
    void *sort_scratch = (void *)(n_dev + 32);
    const size_t sort_bytes = sort_scratch_bytes((uint64_t)N);
    const bool odd = (sort_passes(kDepthKeyBits) & 1) != 0;
    uint32_t *v0 = odd ? vals_tmp : order, *v1 = odd ? order : vals_tmp;
    if (!keys_prefilled) {
        set_error("internal: the depth keys are written by preprocess");
        return SLS_E_ARG;
    }
    int resort_windows = 0;
    const uint64_t *resort_edges = nullptr;
    if (reuse_order && keys_prefilled && fail_flag) {
        // comp: N u64 over the two temporary arrays (8-byte aligned), edges in the sort's count table
        uint64_t *comp = resort_comp_buffer(N, scratch);
        uint64_t *edges = (uint64_t *)(((uintptr_t)sort_scratch + 7) & ~(uintptr_t)7);
        const int nA = (N + kResortWindow - 1) / kResortWindow;
        const int nB = (N + kResortWindow / 2 + kResortWindow - 1) / kResortWindow;   // windows that hold a real element
        ScopedTimer tm(T_RESORT, st);
        if (!window_sort_done) {
            hipLaunchKernelGGL(resort_sort_kernel, dim3(nA), dim3(kResortThreads), 0, st, N, (const uint32_t *)order,
                               (const uint32_t *)keys, comp);
            SLS_LAUNCH_CHECK("resort_sort_kernel");
        }
        const DirectBin no_db = { nullptr, nullptr, nullptr, 0, 0, 0 };
        const bool count_here = direct != nullptr && handoff != nullptr;
        // the pairs of the rounds that follow ping-pong between comp and a second buffer behind the edges
        uint64_t *comp2 = edges + 2 * (size_t)nB + 2;
        // (the LAST merge counts: after it the order is final; every earlier one hands its merged pairs on)
#define SLS_MERGE(PRE_, last_, in_, out_)                                                                                    \
        do {                                                                                                                 \
            if (count_here && (last_))                                                                                       \
                hipLaunchKernelGGL((resort_merge_kernel<true, PRE_>), dim3(nB), dim3(kResortThreads), 0, st, N,               \
                                   (const uint64_t *)(in_), order, edges, tiles, block_sums, GX, (const uint2 *)erec_box,      \
                                   *direct, (uint64_t *)(out_));                                                              \
            else                                                                                                             \
                hipLaunchKernelGGL((resort_merge_kernel<false, PRE_>), dim3(nB), dim3(kResortThreads), 0, st, N,              \
                                   (const uint64_t *)(in_), order, edges, tiles, block_sums, GX, (const uint2 *)erec_box,      \
                                   no_db, (uint64_t *)(out_));                                                                \
        } while (0)
        SLS_MERGE(false, reuse_order <= 1, comp, reuse_order > 1 ? comp : nullptr);
        SLS_LAUNCH_CHECK("resort_merge_kernel");
        uint64_t *cin = comp, *cout = comp2;
        for (int round = 1; round < reuse_order; ++round) {   // (reuse_order = 2: one more round, twice the reach)
            const bool last = round + 1 == reuse_order;
            SLS_MERGE(true, last, cin, last ? nullptr : cout);
            SLS_LAUNCH_CHECK("resort_merge_kernel (further round)");
            uint64_t *t = cin; cin = cout; cout = t;
        }
#undef SLS_MERGE
        if (count_here) {
            if (direct->nchunks != nB || direct->pos0 != -kResortWindow / 2) {
                set_error("internal: the direct binning's chunks are not the repair's windows");
                return SLS_E_ARG;
            }
            handoff->counted = 1;
        }
        resort_windows = nB;
        resort_edges = edges;
    } else {
        int which = 0;
        int rc = radix_sort_pairs_t<uint32_t>(keys, v0, keys_tmp, v1, n_dev, (uint32_t)N, kDepthKeyBits, sort_scratch,
                                              sort_bytes, &which, st, nullptr, 0, true);   // only the order is used
        if (rc) return rc;
        if ((which != 0) != odd) {
            set_error("internal: depth order ended in the wrong buffer");
            return SLS_E_ARG;
        }
    }
    if (handoff && direct && resort_windows == 0) {      // from scratch + direct binning: nothing to scan
        handoff->block_sums = nullptr;
        handoff->resort_windows = 0;
        handoff->resort_edges = nullptr;
        handoff->counted = 0;
        return SLS_OK;
    }
    if (handoff && resort_windows > 0) {   // the emission finishes the scan and checks the repaired order
        handoff->block_sums = block_sums;
        handoff->resort_windows = resort_windows;
        handoff->resort_edges = resort_edges;
        return SLS_OK;
    }
    ScopedTimer tm(T_SCAN, st);
    if (resort_windows == 0) {   // (the merge kernel of the repair already summed the blocks)
        hipLaunchKernelGGL(gather_block_sums_kernel, dim3(nb), dim3(256), 0, st, N, (const uint32_t *)order, tiles,
                           block_sums);
        SLS_LAUNCH_CHECK("gather_block_sums_kernel");
    }
    if (handoff) {
        handoff->block_sums = block_sums;
        handoff->resort_windows = 0;
        handoff->resort_edges = nullptr;
        return SLS_OK;
    }
    hipLaunchKernelGGL(gather_scan_final_kernel, dim3(nb), dim3(256), 0, st, N, (const uint32_t *)order, tiles,
                       (const uint32_t *)block_sums, offsets, total_out, resort_windows, resort_edges, fail_flag);
    SLS_LAUNCH_CHECK("gather_scan_final_kernel");
    return SLS_OK;
}

// Emission in depth order + stable sort by tile + ranges.
//   tkeys/vals, tkeys_tmp/vals_tmp : cap u32 each (ping-pong)
//   count_ptr: device R; cap: host-side capacity of the buffers (>= R, or the
//   overflow flag is raised and the excess instances are dropped)
int launch_bin_sort(const DevCam &cam, int N, const uint32_t *count_ptr, uint32_t cap, const uint32_t *order,
                    const int32_t *rect, const uint32_t *tiles, const uint64_t *tile_mask, const int32_t *erec,
                    const float *depth, const uint32_t *offsets,
                    uint32_t *tkeys, uint32_t *vals, uint32_t *tkeys_tmp, uint32_t *vals_tmp, void *scratch,
                    size_t scratch_bytes, int *sorted_in_tmp, uint32_t *ranges, uint64_t *keys64_out,
                    uint32_t *overflow, hipStream_t st, const ScanHandoff *handoff, uint32_t *total_out,
                    const uint32_t *sbox, const uint2 **bmask_out, int bmask_mode)
{
    // sbox + bmask_out (optional): the sorted list comes out as (surfel, mask of reachable 8x2 pixel blocks) pairs in
    // *bmask_out (sort_bmask_buffer) and the plain value arrays are NOT written; *bmask_out stays null — and the values
    // are written as ever — where that is not possible (more than one sort pass, other tile sizes, switched off)
    const int T = cam.GX * cam.GY;
    *sorted_in_tmp = 0;
    if (bmask_out) *bmask_out = nullptr;
    if (cap == 0 || N == 0) {
        SLS_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T, st));
        return SLS_OK;
    }
    if (scratch_bytes < sort_scratch_bytes(cap)) {
        set_error("sort scratch too small: %zu < %zu", scratch_bytes, sort_scratch_bytes(cap));
        return SLS_E_SCRATCH;
    }
    const int tile_bits = bits_for((uint32_t)(T - 1));
    // one pass over the tile ids (T <= 2048): the sort's digit bases are the ranges
    const bool fused_ranges = sort_passes(tile_bits) == 1 && keys64_out == nullptr;
    if (!fused_ranges) SLS_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T, st));
    // ... and if tile id and surfel index fit one word together, an instance IS one word:
    // half the traffic in emit, histogram and scatter
    const int idx_bits = bits_for((uint32_t)(N - 1)) > 0 ? bits_for((uint32_t)(N - 1)) : 1;
    const bool packed = fused_ranges && tile_bits + idx_bits <= 32;
    // The emission's workgroups as the chunks of the (single) sort pass: they count their instances' tiles themselves,
    // the pass is row scan + scatter — one dependent launch less.
    // Needs the count table (tiles x workgroups) and the chunk starts to fit the sort's scratch.
    constexpr bool no_emit_hist = false;
    // Only while the count table (tiles x chunks words, written, scanned and read back as scattered words) stays small:
    // at 500 k surfels / 64 x 2048 its 512 x 1954 words cost 14 us more than the launch saves, and chunks of 512
    // positions (half the table) make the scatter's waves too few and too long (+16 us); measured gains: -3.3 us per
    // iteration at 50 k / 64 x 1024, -3.2 us at 170 k.
    const int bins = 1 << (tile_bits < 8 ? 8 : tile_bits);
    constexpr int eb = 256;
    const int nemit = (N + eb - 1) / eb;
    uint32_t *cnt = (uint32_t *)scratch, *totals = cnt + (size_t)bins * nemit, *chunk_start = totals + bins;
    const bool emit_hist = fused_ranges && !no_emit_hist && (size_t)bins * nemit <= (size_t)400000 &&
                           ((size_t)bins * nemit + bins + nemit + 1) * sizeof(uint32_t) <= sort_core_bytes(cap);
    // Block masks (the forward's dense rounds) only where the tiles' lists are long enough for the forward to gain more than the binning pays:
    // capacity per tile as the host-side proxy (bmask_mode = SlsMappingConfig.block_masks: 0 auto, 1 always, 2 never)
    const bool long_lists = bmask_mode == 1 || (bmask_mode == 0 && (uint64_t)cap >= 1500ull * (uint64_t)T);
    BlockMaskArgs bm = { nullptr, cam.GX, 1.0f / (float)cam.GX, (cam.GX * kTileW) / 8 };
    uint64_t *wide = nullptr;
    if (sbox && bmask_out && packed && kTileW == 16 && kTileH == 16 && long_lists &&
        block_box_fits(cam.GX * kTileW, cam.H)) {
        bm.out = sort_bmask_buffer(scratch, cap);
        wide = sort_wide_buffer(scratch, cap);
        *bmask_out = bm.out;
    }
    {
        ScopedTimer tm(T_EMIT_KEYS, st);
#define SLS_EMIT(EB_) hipLaunchKernelGGL(emit_tiles_kernel<EB_>, dim3(nemit), dim3(EB_), 0, st, N, cam.GX, order,                 \
                           (const int4 *)rect, tiles, tile_mask, (const int4 *)erec, offsets, cap, tkeys, packed ? (uint32_t *)nullptr : vals, overflow, \
                           packed ? idx_bits : 0, handoff ? *handoff : ScanHandoff{ nullptr, 0, nullptr, 0 }, total_out,  \
                           overflow, emit_hist ? cnt : (uint32_t *)nullptr, bins, emit_hist ? chunk_start : (uint32_t *)nullptr, \
                           sbox, wide)
        SLS_EMIT(eb);
#undef SLS_EMIT
    }
    SLS_LAUNCH_CHECK("emit_tiles_kernel");
    int which = 0;
    int rc;
    if (emit_hist) {
        const uint32_t *vin = packed ? nullptr : vals;
        const int shift = packed ? idx_bits : 0;
        const uint32_t pmask = packed ? ((1u << idx_bits) - 1u) : 0u;
#define SLS_CASE(K_, B) case B: rc = tile_sort_emit_chunks<K_, B>(wide ? (const K_ *)wide : (const K_ *)tkeys, vin, (K_ *)nullptr, vals_tmp, chunk_start, \
                                                              cap, shift, cnt, totals, nemit, (uint2 *)ranges, T, pmask, st, bm); break;
        if (wide) {
            switch (tile_bits < 8 ? 8 : tile_bits) { SLS_CASE(uint64_t, 8) SLS_CASE(uint64_t, 9) SLS_CASE(uint64_t, 10) default: SLS_CASE(uint64_t, 11) }
        } else {
            switch (tile_bits < 8 ? 8 : tile_bits) { SLS_CASE(uint32_t, 8) SLS_CASE(uint32_t, 9) SLS_CASE(uint32_t, 10) default: SLS_CASE(uint32_t, 11) }
        }
#undef SLS_CASE
        which = 1;
    } else if (wide) {
        rc = radix_sort_pairs_t<uint64_t>(wide, nullptr, nullptr, vals_tmp, count_ptr, cap, tile_bits, scratch,
                                          scratch_bytes, &which, st, (uint2 *)ranges, T, true, idx_bits, bm);
    } else if (packed) {
        rc = radix_sort_pairs_t<uint32_t>(tkeys, nullptr, tkeys_tmp, vals_tmp, count_ptr, cap, tile_bits, scratch,
                                          scratch_bytes, &which, st, (uint2 *)ranges, T, true, idx_bits);
    } else {
        rc = radix_sort_pairs_t<uint32_t>(tkeys, vals, tkeys_tmp, vals_tmp, count_ptr, cap, tile_bits, scratch,
                                          scratch_bytes, &which, st, fused_ranges ? (uint2 *)ranges : nullptr, T,
                                          fused_ranges);   // nobody reads the sorted tile ids then
    }
    if (rc) return rc;
    *sorted_in_tmp = which;
    if (!fused_ranges) {
        ScopedTimer tm(T_TILE_RANGES, st);
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((cap + 255) / 256), dim3(256), 0, st,
                           (const uint32_t *)(which ? tkeys_tmp : tkeys), (const uint32_t *)(which ? vals_tmp : vals),
                           count_ptr, cap, depth, (uint2 *)ranges, keys64_out);
        SLS_LAUNCH_CHECK("tile_ranges_kernel");
    }
    return SLS_OK;
}


// ---------------------------------------------------------------------------
// Direct binning, host side.
// ---------------------------------------------------------------------------
static int direct_bins(const DevCam &cam)
{
    const int tb = bits_for((uint32_t)(cam.GX * cam.GY - 1));
    return 1 << (tb < 8 ? 8 : tb);
}
// Can the direct binning serve this camera / size / capacity?  (tiles <= 512, D10 off, 16-bit rectangle fields, the
// count table inside the sort's scratch; otherwise the emission + radix pass)
// words per row of the count table that serve either chunking, with or without the coarse table
static size_t direct_max_stride(int N)
{
    const size_t nchunks = (size_t)(N + kDirectChunk / 2 + kDirectChunk - 1) / kDirectChunk;
    return (nchunks + kDirectGroup - 1) / kDirectGroup * kDirectGroup;
}
bool bin_direct_possible(const DevCam &cam, int N, uint32_t cap)
{
    if (N <= 0 || cap == 0 || cam.tile_cull != 0 || cam.GX > 512 || cam.GY > 64) return false;    // (pack_rect32's fields)
    if (cam.GX * cam.GY > kDirectMaxBins) return false;
    // (its emission records carry the surfel's block box: images the box can describe)
    if (!block_box_fits(cam.GX * kTileW, cam.H)) return false;
    const size_t bins = (size_t)direct_bins(cam);
    return (bins * direct_max_stride(N) + bins + direct_coarse_words(cam, N)) * sizeof(uint32_t) <= sort_core_bytes(cap);
}
// words of the coarse table (groups of kDirectGroup chunks x tiles), for either chunking
size_t direct_coarse_words(const DevCam &cam, int N)
{
    const size_t nchunks = (size_t)(N + kDirectChunk / 2 + kDirectChunk - 1) / kDirectChunk;
    return ((nchunks + kDirectGroup - 1) / kDirectGroup) * (size_t)direct_bins(cam);
}
// the table's place in the sort's scratch; repaired: the chunks are the repair's shifted windows
DirectBin make_direct_bin(const DevCam &cam, int N, void *sort_scratch, uint2 *serec, bool repaired, bool coarse)
{
    DirectBin db;
    db.bins = direct_bins(cam);
    db.nchunks = repaired ? (N + kResortWindow / 2 + kResortWindow - 1) / kResortWindow : (N + kDirectChunk - 1) / kDirectChunk;
    db.pos0 = repaired ? -kResortWindow / 2 : 0;
    db.stride = coarse ? (db.nchunks + kDirectGroup - 1) / kDirectGroup * kDirectGroup : db.nchunks;
    db.cnt = (uint32_t *)sort_scratch;
    db.totals = db.cnt + (size_t)db.bins * db.stride;
    db.serec = serec;
    // (the coarse table sits at a place that does not depend on the chunking: the iteration's first kernel zeroes it
    //  before the depth-order stage decides between repair and radix sort)
    db.coarse = coarse ? (uint32_t *)sort_scratch + (size_t)db.bins * direct_max_stride(N) + db.bins : nullptr;
    return db;
}

// order (+ erec_box, or rect + sbox) -> sorted list (vals_out, or (surfel, block mask) pairs in *bmask_out), ranges, R.
// counted: the count table was filled by the repair's merge (launch_depth_order_scan).
int launch_bin_direct(const DevCam &cam, int N, uint32_t cap, const DirectBin &db, bool counted, const uint32_t *order,
                      const int32_t *erec_box, const int32_t *rect, const uint32_t *sbox, void *scratch, uint32_t *vals_out,
                      uint32_t *ranges, uint32_t *total_out, uint32_t *overflow, int resort_windows,
                      const uint64_t *resort_edges, const uint2 **bmask_out, int bmask_mode, hipStream_t st,
                      uint32_t *status_mirror)
{
    const int T = cam.GX * cam.GY;
    if (bmask_out) *bmask_out = nullptr;
    if (!counted) {
        ScopedTimer tm(T_BIN_COUNT, st);
        hipLaunchKernelGGL(gather_count_kernel, dim3(db.nchunks), dim3(kDirectChunk), 0, st, N, cam.GX, order,
                           (const uint2 *)erec_box, (const int4 *)rect, sbox, db);
        SLS_LAUNCH_CHECK("gather_count_kernel");
    }
    if (!db.coarse) {       // (with the coarse table bin_direct sums what lies in front of a chunk itself)
        ScopedTimer tm(T_SORT_ROWSCAN, st);
        hipLaunchKernelGGL(sort_rowscan_kernel, dim3(db.bins), dim3(256), 0, st, db.cnt, (const uint32_t *)nullptr, cap,
                           db.nchunks, db.totals, db.nchunks);
        SLS_LAUNCH_CHECK("sort_rowscan_kernel");
    }
    // (surfel, block mask) pairs under the rule of launch_bin_sort: long lists (or always / never)
    const bool long_lists = bmask_mode == 1 || (bmask_mode == 0 && (uint64_t)cap >= 1500ull * (uint64_t)T);
    const bool have_box = erec_box != nullptr || sbox != nullptr;
    BlockMaskArgs bm = { nullptr, cam.GX, 1.0f / (float)cam.GX, (cam.GX * kTileW) / 8 };
    if (bmask_out && have_box && long_lists && kTileW == 16 && kTileH == 16 && block_box_fits(cam.GX * kTileW, cam.H)) {
        bm.out = sort_bmask_buffer(scratch, cap);
        *bmask_out = bm.out;
    }
    {
        ScopedTimer tm(T_BIN_DIRECT, st);
        // (every chunk split over 2 / 4 workgroups was measured in round 4 — 24.0 / 26.6 against 24.8 us, profiles/r04f_bin_trace.txt:
        //  the launch is its chain of phases, not its heaviest chunk — the template keeps the parameter, nothing launches it;
        //  what IS split, since round 6, are the heavy chunks at the front of the depth order: the helper workgroups)
        const int nhelp = (db.nchunks < kHeavyChunks ? db.nchunks : kHeavyChunks) * (kHeavyParts - 1);
#define SLS_DIRECT(B_, P_) hipLaunchKernelGGL((bin_direct_kernel<B_, P_, 1>), dim3(nhelp + db.nchunks), dim3(kDirectChunk), 0, st, N, cam.GX, db, \
                               order, (const uint2 *)erec_box, (const int4 *)rect, sbox, cap, vals_out, bm, (uint2 *)ranges, T, \
                               total_out, overflow, resort_windows, resort_edges, overflow, status_mirror)
        if (db.bins == 256) { if (bm.out) SLS_DIRECT(8, true); else SLS_DIRECT(8, false); }
        else { if (bm.out) SLS_DIRECT(9, true); else SLS_DIRECT(9, false); }
#undef SLS_DIRECT
        SLS_LAUNCH_CHECK("bin_direct_kernel");
    }
    return SLS_OK;
}

}  // namespace sls
