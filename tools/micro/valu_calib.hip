// valu_calib.hip — what does a gfx950 SIMD issue per clock, and what do the SQ counters read when it does?
//
// VERDICT r02 / DESIGN.md section 4: bench.py's `valu_frac` was SQ_ACTIVE_INST_VALU over the SIMDs' quad-cycles,
// which reads ~0.95 for the tile kernels although MI355X_MICROARCH.md gives a wave64 v_fma_f32 at 2 clocks
// (SIMD-32).  This program measures, per instruction class of the tile kernels' step loops,
//     clocks per wave-instruction per SIMD at 1 / 2 / 4 / 8 resident waves per SIMD
// with independent operands (issue rate) and — for a few — as a dependent chain (latency), taking the clock from
// s_memtime (shader clock) against the 100 MHz wall clock, so that the actual shader frequency is part of the
// result.  Run once plain and once under `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
// SQ_WAVE_CYCLES` (tools/valu_calib.sh): the counters at a KNOWN issue rate calibrate what "VALU busy" means.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_calib.hip -o gpurun_tmp_valu_calib
//   ./gpurun_tmp_valu_calib [json-out]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum Op {
    OP_FMA = 0,        // v_fma_f32, 8 independent accumulators
    OP_MUL,            // v_mul_f32
    OP_PK_FMA,         // v_pk_fma_f32 (2 lanes-worth per instruction)
    OP_EXP,            // v_exp_f32 (transcendental)
    OP_RCP,            // v_rcp_f32
    OP_DPP_MOV,        // v_mov_b32_dpp quad_perm
    OP_ADD_DPP,        // v_add_f32_dpp quad_perm (DPP folded into an add)
    OP_PERMLANE32,     // v_permlane32_swap_b32
    OP_CNDMASK,        // v_cndmask_b32 (vcc)
    OP_CMP,            // v_cmp_lt_f32 vcc (VALU -> SGPR pair)
    OP_DS_READ128,     // ds_read_b128, 8 in flight, one wait per group
    OP_FMA_CHAIN,      // v_fma_f32 dependent chain (latency)
    OP_EXP_CHAIN,      // v_exp_f32 dependent chain
    OP_DPP_CHAIN,      // v_mov_b32_dpp dependent chain (the quad transmittance chain)
    OP_BALLOT_BRANCH,  // v_cmp + s_and + s_cbranch (ballot -> uniform branch round trip)
    OP_STEP_MIX,       // the forward step's class mix (counts from the disassembly, see kMix below)
    OP_CNDMASK_SGPR,   // v_cndmask_b32_e64 with an SGPR-pair mask (the form the tile kernels mostly use)
    OP_CNDMASK_VCC_W,  // v_cmp writes vcc, v_cndmask reads it (pairs, as a select compiles)
    OP_CNDMASK_VCC_S,  // s_and_b64 writes vcc (a combined condition), four v_cndmask_b32_e32 read it: as the backward step does
    OP_COUNT
};
static const char *kOpName[OP_COUNT] = {
    "v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_mov_b32_dpp", "v_add_f32_dpp",
    "v_permlane32_swap", "v_cndmask_b32", "v_cmp_lt_f32", "ds_read_b128", "v_fma_f32 chain", "v_exp_f32 chain",
    "v_mov_b32_dpp chain", "ballot+branch", "fwd step mix", "v_cndmask_b32 sgpr", "v_cmp+v_cndmask", "s_and vcc + 4 v_cndmask"
};
// wave-instructions per loop iteration of each kernel (what the time is divided by)
static const int kPerIter[OP_COUNT] = { 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 8, 64, 64, 64, 8, 96, 64, 64, 80 };
// VALU wave-instructions per loop iteration (for the SQ_INSTS_VALU cross-check)
static const int kValuPerIter[OP_COUNT] = { 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 0, 64, 64, 64, 8, 84, 64, 64, 64 };

#define REP8(x) x x x x x x x x
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define GROUP8(INS, TAIL)                                                                                           \
    asm volatile(INS " %0, %0" TAIL "\n" INS " %1, %1" TAIL "\n" INS " %2, %2" TAIL "\n" INS " %3, %3" TAIL "\n"     \
                 INS " %4, %4" TAIL "\n" INS " %5, %5" TAIL "\n" INS " %6, %6" TAIL "\n" INS " %7, %7" TAIL "\n"     \
                 : A8 : "v"(m), "v"(c))

template <int OP>
__global__ __launch_bounds__(1024) void calib_kernel(float *out, int iters, uint64_t *clk)
{
    extern __shared__ float4 s_buf[];
    const float s = (float)(threadIdx.x & 7) * 1e-3f + 1.0f;
    float a0 = s, a1 = s + 1, a2 = s + 2, a3 = s + 3, a4 = s + 4, a5 = s + 5, a6 = s + 6, a7 = s + 7;
    v2f p0 = {s, s}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    const v2f pm = {1.0001f, 0.9999f}, pc = {1e-3f, -1e-3f};
    const float m = 0.99991f, c = 1e-4f;
    if (OP == OP_DS_READ128 || OP == OP_STEP_MIX) {
        for (int i = threadIdx.x; i < 64 * 5 * 16; i += blockDim.x) s_buf[i] = make_float4(s, s, s, s);
        __syncthreads();
    }
    const uint32_t lds_addr = (uint32_t)((threadIdx.x & 63) >> 2) * 80u + (threadIdx.x >> 6) * 5120u;   // 4 addresses per quad-group, as the step
    const uint64_t t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == OP_FMA) { REP8(GROUP8("v_fma_f32", ", %8, %9");) }
        else if (OP == OP_MUL) { REP8(GROUP8("v_mul_f32", ", %8");) }
        else if (OP == OP_PK_FMA) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        }
        else if (OP == OP_EXP) {   // sources constant: independent of each other
            REP8(asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %8\n v_exp_f32 %2, %8\n v_exp_f32 %3, %8\n"
                              "v_exp_f32 %4, %8\n v_exp_f32 %5, %8\n v_exp_f32 %6, %8\n v_exp_f32 %7, %8\n" : A8 : "v"(m));)
        }
        else if (OP == OP_RCP) {
            REP8(asm volatile("v_rcp_f32 %0, %8\n v_rcp_f32 %1, %8\n v_rcp_f32 %2, %8\n v_rcp_f32 %3, %8\n"
                              "v_rcp_f32 %4, %8\n v_rcp_f32 %5, %8\n v_rcp_f32 %6, %8\n v_rcp_f32 %7, %8\n" : A8 : "v"(m));)
        }
        else if (OP == OP_DPP_MOV) {
#define DPPQ " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
            REP8(asm volatile("v_mov_b32_dpp %0, %8" DPPQ "\n v_mov_b32_dpp %1, %8" DPPQ "\n v_mov_b32_dpp %2, %8" DPPQ "\n v_mov_b32_dpp %3, %8" DPPQ "\n"
                              "v_mov_b32_dpp %4, %8" DPPQ "\n v_mov_b32_dpp %5, %8" DPPQ "\n v_mov_b32_dpp %6, %8" DPPQ "\n v_mov_b32_dpp %7, %8" DPPQ "\n" : A8 : "v"(m));)
        }
        else if (OP == OP_ADD_DPP) {
            REP8(asm volatile("v_add_f32_dpp %0, %8, %0" DPPQ "\n v_add_f32_dpp %1, %8, %1" DPPQ "\n v_add_f32_dpp %2, %8, %2" DPPQ "\n v_add_f32_dpp %3, %8, %3" DPPQ "\n"
                              "v_add_f32_dpp %4, %8, %4" DPPQ "\n v_add_f32_dpp %5, %8, %5" DPPQ "\n v_add_f32_dpp %6, %8, %6" DPPQ "\n v_add_f32_dpp %7, %8, %7" DPPQ "\n" : A8 : "v"(m));)
        }
        else if (OP == OP_PERMLANE32) {
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n" : A8);)
        }
        else if (OP == OP_CNDMASK) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : A8 : "v"(m) : "vcc");)
        }
        else if (OP == OP_CNDMASK_SGPR) {
            const uint64_t msk = 0x5555555555555555ull ^ (uint64_t)blockIdx.x;
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n" : A8 : "v"(m), "s"(msk));)
        }
        else if (OP == OP_CNDMASK_VCC_W) {
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n" : A8 : "v"(m) : "vcc");)
        }
        else if (OP == OP_CNDMASK_VCC_S) {
            const uint64_t m0 = 0x5555555555555555ull ^ (uint64_t)blockIdx.x, m1 = 0x3333333333333333ull ^ (uint64_t)i;
            REP8(asm volatile("s_and_b64 vcc, %8, %9\n v_cndmask_b32 %0, %0, %10, vcc\n v_cndmask_b32 %1, %1, %10, vcc\n v_cndmask_b32 %2, %2, %10, vcc\n v_cndmask_b32 %3, %3, %10, vcc\n"
                              "s_and_b64 vcc, %9, %8\n v_cndmask_b32 %4, %4, %10, vcc\n v_cndmask_b32 %5, %5, %10, vcc\n v_cndmask_b32 %6, %6, %10, vcc\n v_cndmask_b32 %7, %7, %10, vcc\n" : A8 : "s"(m0), "s"(m1), "v"(m) : "vcc");)
        }
        else if (OP == OP_CMP) {
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n" : A8 : "v"(m) : "vcc");)
        }
        else if (OP == OP_DS_READ128) {
            v4f r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:1280\n ds_read_b128 %6, %8 offset:1296\n ds_read_b128 %7, %8 offset:1312\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(lds_addr));
            a0 += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
        }
        else if (OP == OP_FMA_CHAIN) { REP8(REP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)) }
        else if (OP == OP_EXP_CHAIN) { REP8(REP8(asm volatile("v_exp_f32 %0, %0" : "+v"(a0));)) }
        else if (OP == OP_DPP_CHAIN) { REP8(REP8(asm volatile("v_mov_b32_dpp %0, %0" DPPQ "\n s_nop 1" : "+v"(a0));)) }
        else if (OP == OP_BALLOT_BRANCH) {
            // v_cmp -> s_and/s_cmp -> branch on the result, as `if (!__ballot(live)) continue;` does
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool live = a0 > -1.0e30f;
                if (!__ballot(live)) { a1 += 1.0f; asm volatile("s_nop 0"); }
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));
            }
        }
        else if (OP == OP_STEP_MIX) {
            // the forward step's classes in the proportions of the compiled loop (per step: ~79 VALU of which 2
            // transcendental, 9 DPP, 6 v_cmp, 8 v_cndmask, 3 packed; 6 LDS reads; 3 s_and/branch sequences)
            v4f r0, r1, r2, r3, r4;
            float l0;
            asm volatile("ds_read_b32 %5, %6\n ds_read_b128 %0, %6\n ds_read_b128 %1, %6 offset:16\n ds_read_b128 %2, %6 offset:32\n"
                         "ds_read_b128 %3, %6 offset:48\n ds_read_b128 %4, %6 offset:64\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(l0) : "v"(lds_addr));
            a0 = r0.x * a0 + r1.x; a1 = r2.x * a1 + r3.x; a2 = r4.x * a2 + l0;
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %0, %8, %4\n v_fma_f32 %5, %1, %8, %5\n" : A8 : "v"(m), "v"(c));)          // 48 dependent-ish fma
            asm volatile("v_rcp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(a6), "+v"(a7));                                  // 2
            REP8(asm volatile("v_mov_b32_dpp %0, %1" DPPQ "\n" : "+v"(a5) : "v"(a6));)                                  // 8 DPP, dependent on the rcp
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_mul_f32 %0, %0, %1\n" : "+v"(p0), "+v"(p1) : "v"(pm), "v"(pc));   // 3
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(a4) : "v"(a5) : "vcc");)   // 16
#pragma unroll
            for (int k = 0; k < 3; ++k) {                                                                               // 3 + 3 branch round trips
                const bool live = a4 > -1.0e30f;
                if (!__ballot(live)) { a1 += 1.0f; asm volatile("s_nop 0"); }
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
            }
            a0 += p0.x + p1.y;                                                                                          // ~4
        }
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    float t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    const v2f tp = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    t += tp.x + tp.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

struct Result { int op, waves; double ms, ns_per_inst, clk_per_inst, mhz; };

template <int OP>
static Result run(int waves_per_simd, float *d_out, uint64_t *d_clk, int cus)
{
    // `waves_per_simd` resident waves on every SIMD: one workgroup of 4 * w waves per CU (w <= 4), two for w = 8
    const int wg_per_cu = waves_per_simd > 4 ? 2 : 1;
    const int threads = 64 * 4 * (waves_per_simd / wg_per_cu);
    const int blocks = cus * wg_per_cu;
    const int iters = (OP == OP_DS_READ128 || OP == OP_BALLOT_BRANCH) ? 4000 : 1000;
    const size_t lds = (OP == OP_DS_READ128 || OP == OP_STEP_MIX) ? 64 * 5 * 16 * sizeof(float4) : 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(calib_kernel<OP>, dim3(blocks), dim3(threads), lds, 0, d_out, 20, d_clk);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib_kernel<OP>, dim3(blocks), dim3(threads), lds, 0, d_out, iters, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * blocks);
    hipMemcpy(h.data(), d_clk, sizeof(uint64_t) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)h[2 * b]; wall += (double)h[2 * b + 1]; }
    cyc /= blocks; wall /= blocks;
    Result r;
    r.op = OP; r.waves = waves_per_simd; r.ms = ms;
    const double insts_per_simd = (double)iters * kPerIter[OP] * waves_per_simd;
    r.mhz = wall > 0 ? cyc / (wall * 10.0e-3) : 0.0;          // wall clock ticks at 100 MHz: 10 ns each
    r.ns_per_inst = (double)ms * 1e6 / insts_per_simd;         // HIP-event time of the launch / instructions per SIMD
    // clocks per wave-instruction per SIMD: the event time at the shader frequency the blocks measured themselves
    // (a block's own clock span is NOT the launch's: with 16-wave workgroups the blocks of a launch do not all run
    // side by side, so block clocks under-count; the frequency ratio inside a block is sound)
    r.clk_per_inst = r.ns_per_inst * r.mhz * 1e-3;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

template <int OP>
static void sweep(std::vector<Result> &all, float *d_out, uint64_t *d_clk, int cus)
{
    for (int w : { 1, 2, 4, 8 }) all.push_back(run<OP>(w, d_out, d_clk, cus));
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *d_out; uint64_t *d_clk;
    hipMalloc(&d_out, sizeof(float) * 1024 * 2 * cus);
    hipMalloc(&d_clk, sizeof(uint64_t) * 4 * cus);
    std::vector<Result> all;
    sweep<OP_FMA>(all, d_out, d_clk, cus);
    sweep<OP_MUL>(all, d_out, d_clk, cus);
    sweep<OP_PK_FMA>(all, d_out, d_clk, cus);
    sweep<OP_EXP>(all, d_out, d_clk, cus);
    sweep<OP_RCP>(all, d_out, d_clk, cus);
    sweep<OP_DPP_MOV>(all, d_out, d_clk, cus);
    sweep<OP_ADD_DPP>(all, d_out, d_clk, cus);
    sweep<OP_PERMLANE32>(all, d_out, d_clk, cus);
    sweep<OP_CNDMASK>(all, d_out, d_clk, cus);
    sweep<OP_CMP>(all, d_out, d_clk, cus);
    sweep<OP_DS_READ128>(all, d_out, d_clk, cus);
    sweep<OP_FMA_CHAIN>(all, d_out, d_clk, cus);
    sweep<OP_EXP_CHAIN>(all, d_out, d_clk, cus);
    sweep<OP_DPP_CHAIN>(all, d_out, d_clk, cus);
    sweep<OP_BALLOT_BRANCH>(all, d_out, d_clk, cus);
    sweep<OP_STEP_MIX>(all, d_out, d_clk, cus);
    sweep<OP_CNDMASK_SGPR>(all, d_out, d_clk, cus);
    sweep<OP_CNDMASK_VCC_W>(all, d_out, d_clk, cus);
    printf("%-22s %5s %9s %12s %12s %8s\n", "class", "w/SIMD", "ms", "ns/inst/SIMD", "clk/inst/SIMD", "MHz");
    for (const Result &r : all)
        printf("%-22s %5d %9.3f %12.3f %12.3f %8.0f\n", kOpName[r.op], r.waves, r.ms, r.ns_per_inst, r.clk_per_inst, r.mhz);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"clock_khz_reported\": %d, \"rows\": [\n", prop.gcnArchName, cus, prop.clockRate);
        for (size_t i = 0; i < all.size(); ++i) {
            const Result &r = all[i];
            fprintf(f, " {\"class\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_inst_per_simd\": %.4f, "
                       "\"clk_per_inst_per_simd\": %.4f, \"shader_mhz\": %.1f, \"insts_per_iter\": %d, \"valu_per_iter\": %d}%s\n",
                    kOpName[r.op], r.waves, r.ms, r.ns_per_inst, r.clk_per_inst, r.mhz, kPerIter[r.op], kValuPerIter[r.op],
                    i + 1 < all.size() ? "," : "");
        }
        fprintf(f, "]}\n");
        fclose(f);
    }
    return 0;
}
