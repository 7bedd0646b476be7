// sls_common.hpp — shared by the HIP translation units of libsls_hip.so.
// gfx950 only: wave = 64 lanes, DPP row_bcast available (GFX9 family).
#pragma once
#include <cmath>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sls_abi.h"
#include "../../include/sls_spec.h"

#ifndef SLS_TILE_W
#define SLS_TILE_W 16
#endif
#ifndef SLS_TILE_H
#define SLS_TILE_H 16
#endif
static_assert((SLS_TILE_W * SLS_TILE_H) % 64 == 0, "a tile must be a whole number of wave64s");
static_assert(SLS_TILE_W % 8 == 0 && SLS_TILE_H % 8 == 0, "tiles are made of 8x8 wave sub-tiles");

namespace sls {

constexpr int kWave = 64;
constexpr int kTileW = SLS_TILE_W;
constexpr int kTileH = SLS_TILE_H;
constexpr int kTilePix = kTileW * kTileH;
constexpr int kRec4 = SLS_REC_STRIDE / 4;    // float4 per surfel record
constexpr int kGrec = SLS_GREC_STRIDE;

// thread-local error text behind sls_last_error()
void set_error(const char *fmt, ...);

// Optional per-kernel timing with HIP events recorded on the launch stream
// (sls_timing_enable / sls_timing_collect, used by bench.py for the roofline
// figures).  Disabled: zero cost beyond one branch per launch.
enum TimerSlot {
    T_PREPROCESS_FWD = 0, T_SCAN, T_EMIT_KEYS, T_SORT_HIST, T_SORT_ROWSCAN, T_SORT_SCATTER, T_TILE_RANGES,
    T_RENDER_FWD, T_GREC_MEMSET, T_RENDER_BWD, T_PREPROCESS_BWD, T_ADAM, T_KNN, T_CONSUMER, T_RESORT, T_COUNT
};
// Debug / tuning switches of the CALLING THREAD (sls_debug_variant, sls_debug_wave_cycles, sls_timing_*):
// thread-local, so that the library has no process-global mutable state.
struct DebugState {
    int fwd_variant = 3, bwd_variant = 3;                      // 2: 4x4 pixel blocks, 3: 8x2 (default)
    uint32_t *dbg_fwd_cycles = nullptr, *dbg_bwd_cycles = nullptr;
};
DebugState &debug_state();
void timer_begin(int slot, hipStream_t st);
void timer_end(int slot, hipStream_t st);
struct ScopedTimer {
    int slot; hipStream_t st;
    ScopedTimer(int s, hipStream_t t) : slot(s), st(t) { timer_begin(slot, st); }
    ~ScopedTimer() { timer_end(slot, st); }
};

#define SLS_HIP_CHECK(expr)                                                         \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            ::sls::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                   \
            return SLS_E_HIP;                                                       \
        }                                                                           \
    } while (0)

#define SLS_LAUNCH_CHECK(name)                                                              \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            ::sls::set_error("launch of %s failed: %s", name, hipGetErrorString(e_));       \
            return SLS_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

#define SLS_REQUIRE(cond, msg)                       \
    do {                                             \
        if (!(cond)) {                               \
            ::sls::set_error("%s: %s", __func__, msg); \
            return SLS_E_ARG;                        \
        }                                            \
    } while (0)

// Device copy of the camera, passed by value as a kernel argument.
struct DevCam {
    int H, W, wrap, GX, GY;
    float fx, fy, cx, cy, mod, near_c, far_c;
    float R[9];
    float t[3];
};

inline DevCam make_devcam(const SlsCamera &c)
{
    DevCam d;
    d.H = c.H; d.W = c.W; d.wrap = c.wrap;
    d.GX = (c.W + kTileW - 1) / kTileW;
    d.GY = (c.H + kTileH - 1) / kTileH;
    d.fx = c.fx; d.fy = c.fy; d.cx = c.cx; d.cy = c.cy;
    d.mod = c.scale_modifier; d.near_c = c.near_cut; d.far_c = c.far_cut;
    for (int i = 0; i < 9; ++i) d.R[i] = c.Rvw[i];
    for (int i = 0; i < 3; ++i) d.t[i] = c.tvw[i];
    return d;
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One DPP move: returns src permuted by CTRL; lanes whose row is masked off
// or whose source is invalid receive 0.0f.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 ONLY.
// 6 DPP-modified adds, no LDS traffic (GFX9 row_bcast15/31).
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp_mov0<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov0<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov0<0x124, 0xF>(v);  // row_ror:4
    v += dpp_mov0<0x128, 0xF>(v);  // row_ror:8  -> every lane holds its row's sum
    v += dpp_mov0<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
    v += dpp_mov0<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3
    return v;
}

// Reduce-scatter of 16 per-lane values over the wave: returns, in EVERY lane l,
// the wave-wide sum of component reduce16_component(l) = l >> 2.  Each level
// halves the number of live values by exchanging one half with a partner lane:
//   lanes l / l^32 and rows r / r^1 with the gfx950 v_permlane32_swap /
//   v_permlane16_swap (one swap serves two values, no selects), then row_mirror
//   and row_half_mirror DPP inside a row, then a quad all-reduce.
// 35 VALU instructions for the whole 16 x 64 reduction.
__device__ __forceinline__ int reduce16_component(int lane) { return lane >> 2; }

// a <- [a.lo | b.lo], b <- [a.hi | b.hi] (halves of 32 lanes) for four / two register pairs.
// The leading s_nop covers the VALU-write -> permlane-swap-read hazard (inline asm is opaque
// to the compiler's hazard recogniser).
#define SLS_SWAP4(op, a0, b0, a1, b1, a2, b2, a3, b3)                                              \
    asm volatile("s_nop 1\n\t" op " %0, %1\n\t" op " %2, %3\n\t" op " %4, %5\n\t" op " %6, %7"      \
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3))
__device__ __forceinline__ float wave_reduce16(const float (&x)[16], int lane)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x[i];
    // lanes < 32 keep components 0..7, lanes >= 32 components 8..15
    SLS_SWAP4("v_permlane32_swap_b32", a[0], a[8], a[1], a[9], a[2], a[10], a[3], a[11]);
    SLS_SWAP4("v_permlane32_swap_b32", a[4], a[12], a[5], a[13], a[6], a[14], a[7], a[15]);
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = a[i] + a[i + 8];
    // even rows keep y[0..3], odd rows y[4..7]
    SLS_SWAP4("v_permlane16_swap_b32", y[0], y[4], y[1], y[5], y[2], y[6], y[3], y[7]);
    float z[4], w[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = y[i] + y[i + 4];
    const bool s3 = (lane & 8) != 0, s2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) w[i] = (s3 ? z[i + 2] : z[i]) + dpp_mov0<0x140, 0xF>(s3 ? z[i] : z[i + 2]);   // row_mirror
    float v = (s2 ? w[1] : w[0]) + dpp_mov0<0x141, 0xF>(s2 ? w[0] : w[1]);                                    // row_half_mirror
    v += dpp_mov0<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov0<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    return v;
}

// lane <- lane ^ 4 inside a row (two bank-masked DPP moves).
__device__ __forceinline__ float dpp_xor4(float v)
{
    int a = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);   // row_shl:4 -> banks 0,2
    a = __builtin_amdgcn_update_dpp(a, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);       // row_shr:4 -> banks 1,3
    return __builtin_bit_cast(float, a);
}

// Block variant (lane = 4 * pixel + slot): reduce-scatter of 16 per-lane values over the 16
// PIXELS of each slot.  Lane (pixel p, slot s) receives the sum over the 16 lanes of slot s of
// component p.  Levels: lane^32 and lane^16 with permlane swaps, lane^8 (row_ror:8), lane^4.
__device__ __forceinline__ float block_reduce16(const float (&x)[16], int lane)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x[i];
    SLS_SWAP4("v_permlane32_swap_b32", a[0], a[8], a[1], a[9], a[2], a[10], a[3], a[11]);
    SLS_SWAP4("v_permlane32_swap_b32", a[4], a[12], a[5], a[13], a[6], a[14], a[7], a[15]);
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = a[i] + a[i + 8];
    SLS_SWAP4("v_permlane16_swap_b32", y[0], y[4], y[1], y[5], y[2], y[6], y[3], y[7]);
    float z[4], w[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = y[i] + y[i + 4];
    const bool s3 = (lane & 8) != 0, s2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) w[i] = (s3 ? z[i + 2] : z[i]) + dpp_mov0<0x128, 0xF>(s3 ? z[i] : z[i + 2]);   // row_ror:8
    return (s2 ? w[1] : w[0]) + dpp_xor4(s2 ? w[0] : w[1]);
}

// Bounding box (in lane coordinates of an 8x8 sub-tile, lane = y*8 + x) of the
// lanes set in `m`; all scalar work.  Empty mask -> returns false.
__device__ __forceinline__ bool mask_bbox8x8(uint64_t m, int &xmin, int &xmax, int &ymin, int &ymax)
{
    if (m == 0) return false;
    uint64_t c = m | (m >> 32);
    c |= c >> 16;
    c |= c >> 8;
    const uint32_t cols = (uint32_t)c & 0xFFu;
    uint64_t t = (m | (m >> 4)) & 0x0F0F0F0F0F0F0F0Full;
    t = (t | (t >> 2)) & 0x0303030303030303ull;
    t = (t | (t >> 1)) & 0x0101010101010101ull;
    xmin = __builtin_ctz(cols);
    xmax = 31 - __builtin_clz(cols);
    ymin = __builtin_ctzll(t) >> 3;
    ymax = (63 - __builtin_clzll(t)) >> 3;
    return true;
}

__device__ __forceinline__ float readlane63(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sort key of the depth ordering: positive floats order like their bit patterns; subtracting
// the pattern of 2^-3 (below the 0.2 m near cut) leaves 29 significant bits for any finite LiDAR
// range.  Monotone, so the order of the visible surfels equals the order of their raw depth bits
// (everything below 2^-3 collapses to 0, NaN / inf / huge to the largest key: culled anyway).
__host__ __device__ inline uint32_t depth_order_key(float range)
{
    constexpr uint32_t kBase = 0x3E000000u;             // bits of 0.125f
    constexpr uint32_t kMax = (1u << 29) - 1u;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t b = __float_as_uint(range) & 0x7FFFFFFFu;
#else
    uint32_t b; __builtin_memcpy(&b, &range, 4); b &= 0x7FFFFFFFu;
#endif
    const uint32_t k = b > kBase ? b - kBase : 0u;
    return k < kMax ? k : kMax;
}

// torch.optim.Adam update of one element (no weight decay, no amsgrad); shared by adam_kernel and
// by the update fused into preprocess_bwd so that both produce the same bits.
struct AdamCoef {
    float w1, b2, w2, eps, bc2_sqrt, bc1;   // 1-beta1, beta2, 1-beta2, eps, sqrt(1-beta2^t), 1-beta1^t
};
inline AdamCoef make_adam_coef(double beta1, double beta2, double eps, int64_t step)
{
    AdamCoef c;
    c.w1 = (float)(1.0 - beta1);
    c.b2 = (float)beta2;
    c.w2 = (float)(1.0 - beta2);
    c.eps = (float)eps;
    c.bc1 = (float)(1.0 - pow(beta1, (double)step));
    c.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    return c;
}
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float step_size, const AdamCoef &a)
{
    m = m + (g - m) * a.w1;
    v = v * a.b2 + (a.w2 * g) * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - step_size * (m / denom);
}
// Adam fused into preprocess_bwd (one keyframe per step, sls_mapping_step): moments in the flat
// [xyz 3N | opacity N | scaling 2N | rotation 4N] layout of the gradient bucket.
struct AdamFuse {
    int enabled, write_grads;
    int clear_grec;                       // zero every gradient record after reading it (ready for the next iteration)
    AdamCoef c;
    float lr_xyz, lr_opacity, lr_scaling, lr_rotation;
    float *exp_avg, *exp_avg_sq;          // 10*N floats each
    const uint32_t *skip_flag;            // the iteration's overflow word: non-zero -> no update
    // optional: the iteration's status block (8 words, complete before this last kernel starts) is
    // copied to a host-visible mirror by thread 0, which saves the device->host copy kernel
    uint32_t *status_src;
    uint32_t *status_mirror;
    uint8_t *touched;       // optional: per surfel, set by the backward tile kernel if its gradient record was written
    float *void_flags;      // optional: 2 floats, the void bits for the keyframe-parallel all-reduce
    int void_count, void_stride;   // the flags are written void_count times, void_stride floats apart (one copy per rank's chunk)
    uint32_t gchunk;        // > 0: gradients leave in the reduce-scatter layout (SlsMappingConfig.grad_chunk) ...
    float *gbase;           // ... relative to this base (the flat layout's element 0)
    // deterministic accumulation (render_bwd DET): the gradient record is det_acc * 2^(exponent(det_max) - 166)
    const uint32_t *det_max;
    const long long *det_acc;
    float *reg_accum;                     // optional: workspace scalar holding this iteration's regulariser sum
};

// XCD-aware block -> tile remap: the dispatcher places block b on XCD b % 8
// (speed only, never correctness); give each XCD a contiguous run of tiles so
// neighbouring tiles, which share surfel records, share an L2.
__host__ __device__ inline int xcd_remap(int b, int n)
{
    constexpr int kXcd = 8;
    if (n % kXcd != 0) return b;
    const int per = n / kXcd;
    return (b % kXcd) * per + b / kXcd;
}

}  // namespace sls
