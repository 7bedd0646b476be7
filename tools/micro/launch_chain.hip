// launch_chain.hip — what does a dependent kernel launch cost on this GPU / runtime?  A chain of kernels on one
// stream, each writing what the next reads; per-kernel time from HIP events over many launches.  Run under
// different runtime settings (HIP_FORCE_DEV_KERNARG, AMD_OPT_FLUSH, ...): see tools/launch_chain.sh.
//   hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Big { float v[60]; };   // 240-byte by-value argument, like the tile kernels' camera
__global__ void k_small(const float *in, float *out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = in[0] + 1.0f; }
__global__ void k_big(Big b, const float *in, float *out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = in[0] + b.v[3]; }
__global__ void k_work(const float *in, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int chain = 11, reps = argc > 1 ? atoi(argv[1]) : 300;
    float *a, *b;
    const int n = 1 << 20;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Big big; for (int i = 0; i < 60; ++i) big.v[i] = 1.0f;
    const char *names[] = { "1 block, pointer args", "1 block, 240-byte struct arg", "4096 blocks x 64, pointer args", "8192 blocks x 64 touching 4 MB" };
    for (int mode = 0; mode < 4; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r)
                for (int k = 0; k < chain; ++k) {
                    float *in = (k & 1) ? b : a, *out = (k & 1) ? a : b;
                    if (mode == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, in, out);
                    else if (mode == 1) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, big, in, out);
                    else if (mode == 2) hipLaunchKernelGGL(k_small, dim3(4096), dim3(64), 0, st, in, out);
                    else hipLaunchKernelGGL(k_work, dim3(8192), dim3(128), 0, st, in, out, n);
                }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 1) printf("  %-34s %.2f us per dependent kernel\n", names[mode], 1e3 * ms / (reps * chain));
        }
    }
    return 0;
}
