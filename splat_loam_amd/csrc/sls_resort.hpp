// sls_resort.hpp — the bitonic network of the depth-order repair on register-held pairs (sls_sort.hip: "Temporal
// re-sort of the depth order"), shared by the kernels of sls_sort.hip and by the merged preprocess + window-sort
// launch of sls_preprocess.hip.
#pragma once
#include "sls_common.hpp"

namespace sls {

constexpr int kResortWindow = 1024;
constexpr uint32_t kResortFailed = 2u;            // bit in SlsMappingStatus.overflow

// The bitonic network of a window, from width K0 up to the full window, on elements held in REGISTERS:
// thread t of the 512 owns elements 2t and 2t+1 of the window.
//   * partner distance 1: inside the thread;
//   * distances 2 .. 64 (thread ^ 1 .. 32): inside the wave — DPP (quad_perm, row_shl/shr:4, row_ror:8) for
//     thread distances 1, 2, 4, 8, ds_bpermute for 16 and 32: no LDS storage, no barrier;
//   * distances 128, 256, 512 (thread ^ 64, 128, 256): the only stages that go through LDS (one 128-bit write and
//     one 128-bit read of the partner thread's pair, two workgroup barriers): 6 of the 55 stages of a sort, 3 of
//     the 10 of a merge.
// (The same network with every stage of distance >= 2 in LDS — 45 round trips with bank conflicts on the 64-bit
//  elements — took 16.5 + 7.8 us per repair on C1 against what this one takes)
// Branch-free compare-exchange throughout; all elements are distinct (the surfel index is part of the key).
constexpr int kResortThreads = 512;
static_assert(kResortWindow == 2 * kResortThreads, "one pair of elements per thread");

template <int M>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v)
{
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "lane distance inside a wave");
    if (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    if (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    if (M == 4) {
        int a = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);                    // row_shl:4 -> banks 0, 2
        a = __builtin_amdgcn_update_dpp(a, (int)v, 0x114, 0xF, 0xA, false);                        // row_shr:4 -> banks 1, 3
        return (uint32_t)a;
    }
    if (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    return (uint32_t)__shfl_xor((int)v, M, 64);                                                    // ds_bpermute_b32
}
template <int M>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v)
{
    return ((uint64_t)lane_xor_u32<M>((uint32_t)(v >> 32)) << 32) | lane_xor_u32<M>((uint32_t)v);
}
// e <- min(e, p) if keep_min else max(e, p)
__device__ __forceinline__ void keep64(uint64_t &e, uint64_t p, bool keep_min)
{
    e = ((p < e) == keep_min) ? p : e;
}
// one stage with the partner thread at distance M (element distance 2M) of the width-k step
template <int M>
__device__ __forceinline__ void bitonic_stage(uint64_t &e0, uint64_t &e1, bool asc, ulonglong2 *s_pairs)
{
    const int t = threadIdx.x;
    const bool keep_min = ((t & M) == 0) == asc;
    uint64_t p0, p1;
    if constexpr (M >= 64) {
        s_pairs[t] = make_ulonglong2(e0, e1);
        __syncthreads();
        const ulonglong2 pp = s_pairs[t ^ M];
        p0 = pp.x; p1 = pp.y;
        __syncthreads();
    } else {
        p0 = lane_xor_u64<M>(e0); p1 = lane_xor_u64<M>(e1);
    }
    keep64(e0, p0, keep_min);
    keep64(e1, p1, keep_min);
}
template <int K>
__device__ __forceinline__ void bitonic_width(uint64_t &e0, uint64_t &e1, ulonglong2 *s_pairs)
{
    // direction of the width-K blocks (the last width sorts ascending throughout)
    const bool asc = (K == kResortWindow) || ((2 * (int)threadIdx.x) & K) == 0;
    if constexpr (K >= 1024) bitonic_stage<256>(e0, e1, asc, s_pairs);
    if constexpr (K >= 512) bitonic_stage<128>(e0, e1, asc, s_pairs);
    if constexpr (K >= 256) bitonic_stage<64>(e0, e1, asc, s_pairs);
    if constexpr (K >= 128) bitonic_stage<32>(e0, e1, asc, s_pairs);
    if constexpr (K >= 64) bitonic_stage<16>(e0, e1, asc, s_pairs);
    if constexpr (K >= 32) bitonic_stage<8>(e0, e1, asc, s_pairs);
    if constexpr (K >= 16) bitonic_stage<4>(e0, e1, asc, s_pairs);
    if constexpr (K >= 8) bitonic_stage<2>(e0, e1, asc, s_pairs);
    if constexpr (K >= 4) bitonic_stage<1>(e0, e1, asc, s_pairs);
    // distance 1: the two elements of the thread
    const bool sw = (e0 > e1) == asc;
    const uint64_t lo = sw ? e1 : e0, hi = sw ? e0 : e1;
    e0 = lo; e1 = hi;
}
// full sort (K0 = 2) or merge of a bitonic window (K0 = kResortWindow)
template <int K0>
__device__ __forceinline__ void bitonic_pairs(uint64_t &e0, uint64_t &e1, ulonglong2 *s_pairs)
{
    if constexpr (K0 <= 2) bitonic_width<2>(e0, e1, s_pairs);
    if constexpr (K0 <= 4) bitonic_width<4>(e0, e1, s_pairs);
    if constexpr (K0 <= 8) bitonic_width<8>(e0, e1, s_pairs);
    if constexpr (K0 <= 16) bitonic_width<16>(e0, e1, s_pairs);
    if constexpr (K0 <= 32) bitonic_width<32>(e0, e1, s_pairs);
    if constexpr (K0 <= 64) bitonic_width<64>(e0, e1, s_pairs);
    if constexpr (K0 <= 128) bitonic_width<128>(e0, e1, s_pairs);
    if constexpr (K0 <= 256) bitonic_width<256>(e0, e1, s_pairs);
    if constexpr (K0 <= 512) bitonic_width<512>(e0, e1, s_pairs);
    bitonic_width<1024>(e0, e1, s_pairs);
}
// window position of element q (0, 1) of thread t when the window is loaded as a bitonic sequence: the second
// half back to front (ascending + descending)
__device__ __forceinline__ int bitonic_src(int o)
{
    return o < kResortWindow / 2 ? o : (kResortWindow + kResortWindow / 2 - 1 - o);
}


// Step A of the repair for one window: thread t of the 512 takes positions 2t, 2t+1 of the OLD order, keys them with
// the NEW depth keys, sorts the window.  KeyOf(g) -> u32 key of surfel g.
template <typename KeyOf>
__device__ __forceinline__ void resort_sort_window(int window, int N, const uint32_t *__restrict__ prev_order, KeyOf key_of,
                                                   uint64_t *__restrict__ comp, ulonglong2 *s_pairs)
{
    const int pos0 = window * kResortWindow + 2 * (int)threadIdx.x;
    // (unconditional loads at clamped addresses, all of a level before the next: two dependent round trips
    //  instead of one branch + full wait per element)
    const uint32_t g0 = min(prev_order[min(pos0, N - 1)], (uint32_t)(N - 1));        // (memory-safe whatever the caller kept)
    const uint32_t g1 = min(prev_order[min(pos0 + 1, N - 1)], (uint32_t)(N - 1));
    const uint32_t k0 = key_of(g0), k1 = key_of(g1);
    uint64_t e0 = pos0 < N ? (((uint64_t)k0 << 32) | g0) : ~0ull;                    // padding behind the end sorts last
    uint64_t e1 = pos0 + 1 < N ? (((uint64_t)k1 << 32) | g1) : ~0ull;
    bitonic_pairs<2>(e0, e1, s_pairs);
    if (pos0 < N) comp[pos0] = e0;
    if (pos0 + 1 < N) comp[pos0 + 1] = e1;
}

}  // namespace sls
