// mem_calib.hip — what the TCC traffic counters (FETCH_SIZE / WRITE_SIZE) report for access patterns with a KNOWN
// number of accesses: wide and narrow streaming reads, scattered narrow gathers (one access per 128-byte line, every
// line visited once, working set far beyond L2 + MALL), scattered narrow stores.  VERDICT r03 item 3: the guide's
// gfx950 correction (FETCH_SIZE x2) was calibrated on 16 B / lane streaming only (adam_kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mem_calib.hip -o gpurun_tmp_mem_calib
//   tools/mem_calib.sh (on the GPU box: plain run + two rocprofv3 --pmc passes) -> gpurun_out/mem_calibration.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename T>
__global__ void stream_read(const T *__restrict__ a, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = a[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// one access of sizeof(T) per 128-byte line; the lines in a pseudo-random order (odd multiplier modulo a power of two)
template <typename T>
__global__ void gather_lines(const unsigned char *__restrict__ a, size_t nlines, size_t mult, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (i * mult) & (nlines - 1);
        const T v = *reinterpret_cast<const T *>(a + line * 128);
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename T>
__global__ void scatter_lines(unsigned char *__restrict__ a, size_t nlines, size_t mult)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (i * mult) & (nlines - 1);
        T v;
        uint32_t *w = reinterpret_cast<uint32_t *>(&v);
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) w[k] = (uint32_t)i + k;
        *reinterpret_cast<T *>(a + line * 128) = v;
    }
}
// every 8-byte slot of the array written exactly once, in a pseudo-random order: what a radix scatter does
__global__ void scatter_all8(uint2 *__restrict__ a, size_t n, size_t mult)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[(i * mult) & (n - 1)] = make_uint2((uint32_t)i, 7u);
}
__global__ void stream_write16(uint4 *__restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main()
{
    const size_t bytes = (size_t)2 << 30;            // 2 GiB: far beyond L2 (32 MB) + MALL (256 MB)
    const size_t nlines = bytes / 128;
    unsigned char *a;
    uint32_t *sink;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, bytes));
    const size_t mult = 0x9E3779B1ull | 1ull;
    const dim3 grid(4096), block(256);
    const size_t n_small = (size_t)64 << 20;         // 64 MiB working sets for the dense cases
    printf("{\"lines\": %zu, \"cases\": [\n", nlines);
#define RUN(name_, accesses_, bytes_each_, ...) do { __VA_ARGS__; CHECK(hipDeviceSynchronize()); \
        printf("  {\"kernel\": \"%s\", \"accesses\": %zu, \"bytes_per_access\": %d},\n", name_, (size_t)(accesses_), (int)(bytes_each_)); } while (0)
    RUN("stream_read<uint4>", bytes / 16, 16, hipLaunchKernelGGL(stream_read<uint4>, grid, block, 0, 0, (const uint4 *)a, bytes / 16, sink));
    RUN("stream_read<uint2>", bytes / 8, 8, hipLaunchKernelGGL(stream_read<uint2>, grid, block, 0, 0, (const uint2 *)a, bytes / 8, sink));
    RUN("stream_read<uint32_t>", bytes / 4, 4, hipLaunchKernelGGL(stream_read<uint32_t>, grid, block, 0, 0, (const uint32_t *)a, bytes / 4, sink));
    RUN("gather_lines<uint32_t>", nlines, 4, hipLaunchKernelGGL(gather_lines<uint32_t>, grid, block, 0, 0, a, nlines, mult, sink));
    RUN("gather_lines<uint2>", nlines, 8, hipLaunchKernelGGL(gather_lines<uint2>, grid, block, 0, 0, a, nlines, mult, sink));
    RUN("gather_lines<uint4>", nlines, 16, hipLaunchKernelGGL(gather_lines<uint4>, grid, block, 0, 0, a, nlines, mult, sink));
    RUN("scatter_lines<uint32_t>", nlines, 4, hipLaunchKernelGGL(scatter_lines<uint32_t>, grid, block, 0, 0, a, nlines, mult));
    RUN("scatter_lines<uint2>", nlines, 8, hipLaunchKernelGGL(scatter_lines<uint2>, grid, block, 0, 0, a, nlines, mult));
    RUN("scatter_all8", n_small / 8, 8, hipLaunchKernelGGL(scatter_all8, grid, block, 0, 0, (uint2 *)a, n_small / 8, mult));
    RUN("stream_write16", bytes / 16, 16, hipLaunchKernelGGL(stream_write16, grid, block, 0, 0, (uint4 *)a, bytes / 16));
    printf("  {\"kernel\": \"end\", \"accesses\": 0, \"bytes_per_access\": 0}]}\n");
    return 0;
}
