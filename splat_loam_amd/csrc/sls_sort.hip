// sls_sort.hip — tile binning (A3), LSD radix sort of (tile<<32 | depth bits)
// keys with surfel-index payload (A4) and per-tile range detection (A5).
// SURVEY.md §8a; all integer work, checked bit-exactly.
//
// Radix sort design (wave64-native, no vendor library):
//   * 8-bit digits, only the ceil((32 + tile_bits)/8) passes that carry
//     information;
//   * the unit of work is ONE WAVE owning 1024 consecutive items (16 rounds of
//     64): no workgroup barrier is needed inside a pass, each wave keeps its
//     256 running bucket cursors in a private LDS slice;
//   * stable ranking inside a round by 8 ballots (the set of lanes holding my
//     digit), rank = popcount(peers below me);
//   * per pass: histogram -> per-digit row scan over chunks -> scatter.
// HBM-bound: a pass reads 8 B/item (histogram) + 12 B/item and writes 12 B/item.
#include "sls_common.hpp"

namespace sls {

constexpr int kSortRounds = 16;
constexpr int kSortWaveItems = kWave * kSortRounds;  // 1024 items per wave
constexpr int kSortWavesPerBlock = 4;

// ---------------------------------------------------------------------------
// A3: one thread per surfel walks its tile rectangle (row-major: y outer,
// x inner, x wrapping modulo the grid width in 360-degree mode, D5/D9).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emit_keys_kernel(int N, int GX, const int4 *__restrict__ rect,
                                                        const uint32_t *__restrict__ tiles,
                                                        const uint32_t *__restrict__ offsets,
                                                        const float *__restrict__ depth,
                                                        uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t t = tiles[i];
    if (!t) return;
    uint32_t off = offsets[i] - t;
    const int4 rc = rect[i];
    const uint64_t db = (uint64_t)__float_as_uint(depth[i]);
    for (int y = 0; y < rc.w; ++y) {
        const uint32_t row = (uint32_t)(rc.z + y) * (uint32_t)GX;
        for (int k = 0; k < rc.y; ++k) {
            int tx = rc.x + k;
            if (tx >= GX) tx -= GX;
            keys[off] = ((uint64_t)(row + (uint32_t)tx) << 32) | db;
            vals[off] = (uint32_t)i;
            ++off;
        }
    }
}

// ---------------------------------------------------------------------------
// A4 pass, step 1: per-wave-chunk digit histogram -> cnt[digit][chunk].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_hist_kernel(const uint64_t *__restrict__ keys, uint64_t R, int shift,
                                                        uint32_t *__restrict__ cnt, int nchunks)
{
    __shared__ uint32_t s_hist[kSortWavesPerBlock][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x * kSortWavesPerBlock + wave;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_hist[wave][lane + 64 * k] = 0;
    if (chunk >= nchunks) return;
    const uint64_t base = (uint64_t)chunk * kSortWaveItems + lane;
    uint64_t k[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint64_t idx = base + (uint64_t)r * 64;
        k[r] = (idx < R) ? keys[idx] : ~0ull;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint64_t idx = base + (uint64_t)r * 64;
        if (idx < R) atomicAdd(&s_hist[wave][(uint32_t)(k[r] >> shift) & 255u], 1u);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = lane + 64 * q;
        cnt[(size_t)d * nchunks + chunk] = s_hist[wave][d];
    }
}

// A4 pass, step 2: block d scans row d of cnt[][] exclusively in place and
// writes the row total.
__global__ __launch_bounds__(256) void sort_rowscan_kernel(uint32_t *__restrict__ cnt, int nchunks,
                                                           uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    uint32_t *row = cnt + (size_t)blockIdx.x * nchunks;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < nchunks) ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
        for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
        const uint32_t carry = s_carry;
        if (i < nchunks) row[i] = carry + wave_prefix + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + wave_prefix + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}

// A4 pass, step 3: stable scatter.
__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint64_t *__restrict__ keys_in,
                                                           const uint32_t *__restrict__ vals_in,
                                                           uint64_t *__restrict__ keys_out,
                                                           uint32_t *__restrict__ vals_out, uint64_t R, int shift,
                                                           const uint32_t *__restrict__ cnt,
                                                           const uint32_t *__restrict__ totals, int nchunks)
{
    __shared__ uint32_t s_cursor[kSortWavesPerBlock][256];
    __shared__ uint32_t s_digit_base[256];
    __shared__ uint32_t s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // exclusive scan of the 256 digit totals (one per thread)
        const uint32_t v = totals[threadIdx.x];
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
        for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
        s_digit_base[threadIdx.x] = wave_prefix + incl - v;
        __syncthreads();
    }
    const int chunk = blockIdx.x * kSortWavesPerBlock + wave;
    if (chunk >= nchunks) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = lane + 64 * q;
        s_cursor[wave][d] = s_digit_base[d] + cnt[(size_t)d * nchunks + chunk];
    }
    const uint64_t base = (uint64_t)chunk * kSortWaveItems + lane;
    uint64_t k[kSortRounds];
    uint32_t v[kSortRounds];
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint64_t idx = base + (uint64_t)r * 64;
        const bool valid = idx < R;
        k[r] = valid ? keys_in[idx] : ~0ull;
        v[r] = valid ? vals_in[idx] : 0u;
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < kSortRounds; ++r) {
        const uint64_t idx = base + (uint64_t)r * 64;
        const bool valid = idx < R;
        const uint32_t digit = (uint32_t)(k[r] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t count = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = s_cursor[wave][digit] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[pos] = k[r];
            vals_out[pos] = v[r];
            if (rank == count - 1) s_cursor[wave][digit] = pos + 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------
// A5: [start, end) of every tile in the sorted list.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint64_t *__restrict__ keys, uint64_t R,
                                                          uint2 *__restrict__ ranges)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= R) return;
    const uint32_t t = (uint32_t)(keys[j] >> 32);
    if (j == 0 || (uint32_t)(keys[j - 1] >> 32) != t) ranges[t].x = (uint32_t)j;
    if (j + 1 == R || (uint32_t)(keys[j + 1] >> 32) != t) ranges[t].y = (uint32_t)(j + 1);
}

// ---------------------------------------------------------------------------
size_t sort_scratch_bytes(uint64_t R)
{
    const uint64_t nchunks = (R + kSortWaveItems - 1) / kSortWaveItems;
    return (size_t)(256 * (nchunks ? nchunks : 1) + 256) * sizeof(uint32_t);
}

static int bits_for(uint32_t max_value)
{
    int b = 0;
    while (b < 32 && (max_value >> b) != 0) ++b;
    return b;
}

// Stable LSD radix sort of (u64 key, u32 value) pairs on the low `nbits` key
// bits.  Ping-pongs between (keys, vals) and (keys_tmp, vals_tmp); *result_in_tmp
// says where the sorted data ended up.
int radix_sort_pairs(uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp, uint64_t R,
                     int nbits, void *scratch, size_t scratch_bytes, int *result_in_tmp, hipStream_t st)
{
    *result_in_tmp = 0;
    if (R == 0) return SLS_OK;
    if (scratch_bytes < sort_scratch_bytes(R)) {
        set_error("sort scratch too small: %zu < %zu", scratch_bytes, sort_scratch_bytes(R));
        return SLS_E_SCRATCH;
    }
    const int nchunks = (int)((R + kSortWaveItems - 1) / kSortWaveItems);
    const int nblocks = (nchunks + kSortWavesPerBlock - 1) / kSortWavesPerBlock;
    uint32_t *cnt = (uint32_t *)scratch;
    uint32_t *totals = cnt + (size_t)256 * nchunks;
    const int npasses = (nbits + 7) / 8;
    uint64_t *kb[2] = { keys, keys_tmp };
    uint32_t *vb[2] = { vals, vals_tmp };
    for (int p = 0; p < npasses; ++p) {
        const int shift = 8 * p;
        const int src = p & 1, dst = src ^ 1;
        {
            ScopedTimer tm(T_SORT_HIST, st);
            hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(256), 0, st, kb[src], R, shift, cnt, nchunks);
        }
        SLS_LAUNCH_CHECK("sort_hist_kernel");
        {
            ScopedTimer tm(T_SORT_ROWSCAN, st);
            hipLaunchKernelGGL(sort_rowscan_kernel, dim3(256), dim3(256), 0, st, cnt, nchunks, totals);
        }
        SLS_LAUNCH_CHECK("sort_rowscan_kernel");
        {
            ScopedTimer tm(T_SORT_SCATTER, st);
            hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(256), 0, st, kb[src], vb[src], kb[dst],
                               vb[dst], R, shift, cnt, totals, nchunks);
        }
        SLS_LAUNCH_CHECK("sort_scatter_kernel");
    }
    *result_in_tmp = npasses & 1;
    return SLS_OK;
}

int launch_bin_sort(const DevCam &cam, int N, uint64_t R, const int32_t *rect, const uint32_t *tiles,
                    const float *depth, const uint32_t *offsets, uint64_t *keys, uint32_t *vals,
                    uint64_t *keys_tmp, uint32_t *vals_tmp, void *scratch, size_t scratch_bytes,
                    int *sorted_in_tmp, uint32_t *ranges, hipStream_t st)
{
    const int T = cam.GX * cam.GY;
    SLS_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T, st));
    *sorted_in_tmp = 0;
    if (R == 0) return SLS_OK;
    {
        ScopedTimer tm(T_EMIT_KEYS, st);
        hipLaunchKernelGGL(emit_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, cam.GX, (const int4 *)rect,
                           tiles, offsets, depth, keys, vals);
    }
    SLS_LAUNCH_CHECK("emit_keys_kernel");

    const int total_bits = 32 + bits_for((uint32_t)(T - 1));
    int which = 0;
    int rc = radix_sort_pairs(keys, vals, keys_tmp, vals_tmp, R, total_bits, scratch, scratch_bytes, &which, st);
    if (rc) return rc;
    uint64_t *kb[2] = { keys, keys_tmp };
    *sorted_in_tmp = which;
    {
        ScopedTimer tm(T_TILE_RANGES, st);
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                           kb[which], R, (uint2 *)ranges);
    }
    SLS_LAUNCH_CHECK("tile_ranges_kernel");
    return SLS_OK;
}

}  // namespace sls
