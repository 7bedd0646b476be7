// Micro-benchmark: issue rate of packed FP32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against their
// scalar forms on gfx950, 4 waves per SIMD, independent accumulators (no dependency stalls).  Measured on MI355X:
// v_fma_f32 0.655 ms, v_mul_f32 0.601 ms, v_pk_fma_f32 1.121 ms, v_pk_mul_f32 1.106 ms, v_pk_add_f32 1.081 ms for the
// same number of instructions: a packed instruction holds the issue port 1.7-1.8x as long as a scalar one.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s, int iters)
{
    v2f a0 = {s, s + 1}, a1 = {s + 2, s + 3}, a2 = {s + 4, s + 5}, a3 = {s + 6, s + 7};
    v2f a4 = {s + 8, s + 9}, a5 = {s + 10, s + 11}, a6 = {s + 12, s + 13}, a7 = {s + 14, s + 15};
    const v2f m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 16 scalar fma per group
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 1) { // 8 packed fma per group (16 fma of work)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 2) { // 8 packed mul
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 3) { // 8 packed add
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else {                // 8 scalar mul
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y) : "v"(m.x));)
        }
    }
    const v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y;
}
template <int MODE>
static void run(const char *name, float *d)
{
    const int iters = 2000, blocks = 256 * 4;     // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x 64 instructions per iteration (8 asm statements of 8 instructions)
    const double insts = (double)iters * 64.0 * 4.0;
    printf("%-14s %8.3f ms  -> %.2f ns per instruction per SIMD\n", name, ms, ms * 1e6 / insts);
}
int main()
{
    float *d;
    hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d); run<4>("v_mul_f32", d);
    return 0;
}
