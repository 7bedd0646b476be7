// tile_sort.hip — experiment for the "binning without a global depth order" lead: how long does it
// take to sort every tile's list by (depth key, surfel index) inside LDS, one workgroup per tile?  Lists of the lengths
// given on stdin (one per tile: the bench scene's, written by tools/tile_sort.sh), random 48-bit keys + a 32-bit value
// (the block mask), rocPRIM's block radix sort as a stand-in for a hand-written one (1024 threads, 2 / 4 / 8 / 16 items
// per thread by list length).  Prints JSON: launch time (HIP events, best of 20), the longest list, and a check.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tile_sort.hip -o gpurun_tmp_tile_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int IPT>
__device__ __forceinline__ void sort_tile(uint64_t *keys, uint32_t *vals, int n, void *lds)
{
    using Sort = rocprim::block_radix_sort<uint64_t, 1024, IPT, uint32_t>;
    typename Sort::storage_type &st = *reinterpret_cast<typename Sort::storage_type *>(lds);
    uint64_t k[IPT];
    uint32_t v[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {     // blocked arrangement: thread t holds items t * IPT ..
        const int i = (int)threadIdx.x * IPT + j;
        k[j] = i < n ? keys[i] : ~0ull;
        v[j] = i < n ? vals[i] : 0u;
    }
    Sort().sort(k, v, st, 0, 48);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int i = (int)threadIdx.x * IPT + j;
        if (i < n) { keys[i] = k[j]; vals[i] = v[j]; }
    }
}

__global__ __launch_bounds__(1024) void tile_sort_kernel(uint64_t *keys, uint32_t *vals, const uint2 *ranges)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint2 r = ranges[blockIdx.x];
    const int n = (int)(r.y - r.x);
    uint64_t *k = keys + r.x;
    uint32_t *v = vals + r.x;
    if (n <= 2048) sort_tile<2>(k, v, n, lds);
    else if (n <= 4096) sort_tile<4>(k, v, n, lds);
    else if (n <= 8192) sort_tile<8>(k, v, n, lds);
    else sort_tile<16>(k, v, min(n, 16384), lds);
}

int main()
{
    std::vector<uint32_t> len;
    unsigned x;
    while (scanf("%u", &x) == 1) len.push_back(x);
    if (len.empty()) { fprintf(stderr, "tile lengths on stdin\n"); return 1; }
    const int T = (int)len.size();
    std::vector<uint2> ranges(T);
    uint32_t R = 0, longest = 0;
    for (int t = 0; t < T; ++t) { ranges[t] = make_uint2(R, R + len[t]); R += len[t]; longest = std::max(longest, len[t]); }
    std::vector<uint64_t> hk(R);
    std::vector<uint32_t> hv(R);
    std::mt19937_64 rng(1);
    for (uint32_t i = 0; i < R; ++i) { hk[i] = rng() & ((1ull << 48) - 1); hv[i] = (uint32_t)i; }
    uint64_t *dk, *dk0; uint32_t *dv, *dv0; uint2 *dr;
    CHECK(hipMalloc(&dk, R * 8)); CHECK(hipMalloc(&dk0, R * 8)); CHECK(hipMalloc(&dv, R * 4)); CHECK(hipMalloc(&dv0, R * 4));
    CHECK(hipMalloc(&dr, T * sizeof(uint2)));
    CHECK(hipMemcpy(dk0, hk.data(), R * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dv0, hv.data(), R * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dr, ranges.data(), T * sizeof(uint2), hipMemcpyHostToDevice));
    size_t lds = std::max({ sizeof(rocprim::block_radix_sort<uint64_t, 1024, 2, uint32_t>::storage_type),
                            sizeof(rocprim::block_radix_sort<uint64_t, 1024, 4, uint32_t>::storage_type),
                            sizeof(rocprim::block_radix_sort<uint64_t, 1024, 8, uint32_t>::storage_type),
                            sizeof(rocprim::block_radix_sort<uint64_t, 1024, 16, uint32_t>::storage_type) });
    CHECK(hipFuncSetAttribute((const void *)tile_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 20; ++it) {
        CHECK(hipMemcpyAsync(dk, dk0, R * 8, hipMemcpyDeviceToDevice, 0)); CHECK(hipMemcpyAsync(dv, dv0, R * 4, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(tile_sort_kernel, dim3(T), dim3(1024), lds, 0, dk, dv, dr);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CHECK(hipMemcpy(hk.data(), dk, R * 8, hipMemcpyDeviceToHost));
    bool sorted = true;
    for (int t = 0; t < T && sorted; ++t)
        for (uint32_t i = ranges[t].x + 1; i < ranges[t].y && i < ranges[t].x + 16384; ++i)
            if (hk[i - 1] > hk[i]) { sorted = false; break; }
    printf("{\"tiles\": %d, \"instances\": %u, \"longest_list\": %u, \"lds_bytes\": %zu, \"launch_us\": %.2f, \"sorted\": %s}\n",
           T, R, longest, lds, best * 1000.0f, sorted ? "true" : "false");
    return sorted ? 0 : 2;
}
