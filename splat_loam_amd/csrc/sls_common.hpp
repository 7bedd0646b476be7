// sls_common.hpp — shared by the HIP translation units of libsls_hip.so.
// gfx950 only: wave = 64 lanes, DPP row_bcast available (GFX9 family).
#pragma once
#include <atomic>
#include <cmath>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sls_abi.h"
#include "../../include/sls_spec.h"
#include "../../include/sls_det_math.h"

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
    T_RENDER_FWD, T_GREC_MEMSET, T_RENDER_BWD, T_PREPROCESS_BWD, T_ADAM, T_KNN, T_CONSUMER, T_RESORT, T_BIN_COUNT, T_BIN_DIRECT,
    T_COUNT
};
// Debug / tuning switches (sls_debug_variant, sls_debug_wave_cycles, sls_timing_*): ONE set per process, relaxed
// atomics — a backward reached through torch autograd runs on the autograd engine's device thread and must see
// what the Python thread chose.  Diagnostics only; the data path itself keeps no mutable state.
struct DebugState {
    std::atomic<int> fwd_variant{3}, bwd_variant{3};           // 2: 4x4 pixel blocks, 3: 8x2 (default)
    std::atomic<uint32_t *> dbg_fwd_cycles{nullptr}, dbg_bwd_cycles{nullptr};
};
DebugState &debug_state();
void timer_begin(int slot, hipStream_t st);
void timer_end(int slot, hipStream_t st);
struct ScopedTimer {
    int slot; hipStream_t st;
    bool open = true;
    ScopedTimer(int s, hipStream_t t) : slot(s), st(t) { timer_begin(slot, st); }
    void end_now() { if (open) { timer_end(slot, st); open = false; } }
    ~ScopedTimer() { end_now(); }
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
    SlsTileCullCam tc;      // D10: constants of the tile-level footprint test in the binning (sls_det_math.h)
    int tile_cull;          // D10: rectangles of at least this many (and at most 64) tiles are tested, instances that
                            // cannot contribute are not emitted; 0: off, the whole rectangle is emitted
};

// Host: the per-camera constants of the tile test, in double and rounded once — the checker computes the same
// expressions with the same libm (its own copy of this function).
inline SlsTileCullCam make_tile_cull_cam(float fx, float fy, int W, int wrap)
{
    SlsTileCullCam c;
    const double hx = 0.5 * (double)(kTileW - 1), hy = 0.5 * (double)(kTileH - 1);
    const double ax = 0.5 / (double)fx, ay = 0.5 / (double)fy;
    c.chx = (float)cos(ax); c.shx = (float)sin(ax);
    c.chy = (float)cos(ay); c.shy = (float)sin(ay);
    const double kx = hx / (double)fx, ky = hy / (double)fy;
    c.kx = (float)kx; c.ky = (float)ky;
    const double span = fabs(kx) + fabs(ky);
    c.eps = (float)(0.5 * span * span * 1.01 + 4.0e-6);
    c.hx = (float)hx; c.hy = (float)hy;
    c.wrapW = wrap ? (float)W : 0.0f;
    c.invW = wrap ? 1.0f / (float)W : 0.0f;
    return c;
}

inline DevCam make_devcam(const SlsCamera &c)
{
    DevCam d;
    d.H = c.H; d.W = c.W; d.wrap = c.wrap;
    d.GX = (c.W + kTileW - 1) / kTileW;
    d.GY = (c.H + kTileH - 1) / kTileH;
    d.fx = c.fx; d.fy = c.fy;
    // D1 as a parameter: pixel (c, r) sits at image coordinate (c + ox, r + oy), i.e. the kernels see the
    // principal point (cx - ox, cy - oy) — one float subtraction, the same the checker makes
    d.cx = c.cx - c.pix_offset[0]; d.cy = c.cy - c.pix_offset[1];
    d.mod = c.scale_modifier; d.near_c = c.near_cut; d.far_c = c.far_cut;
    for (int i = 0; i < 9; ++i) d.R[i] = c.Rvw[i];
    for (int i = 0; i < 3; ++i) d.t[i] = c.tvw[i];
    d.tc = make_tile_cull_cam(d.fx, d.fy, d.W, d.wrap);
    // SlsCamera.tile_cull_min: 0 = the default threshold, 1 = test off, k >= 2 = rectangles of k..64 tiles are tested
    d.tile_cull = c.tile_cull_min == 0 ? SLS_TILE_CULL_MIN_DEFAULT : (c.tile_cull_min == 1 ? 0 : c.tile_cull_min);
    return d;
}

// Block box: a surfel's support box (the record's centre + half extents, what the tile kernels' cull_pass tests) as two
// integer ranges over the image's 8x2 pixel blocks — block columns [lo, lo + n] modulo the NC columns of the image (a
// block column c passes cull_pass iff |8c + 3.5 - cx| <= ex + 3.5), block rows [lo, hi] (|2r + 0.5 - cy| <= ey + 0.5);
// 0.01 pixels of slack.  ONE word: column lo (9 bits) | n + 1 (10 bits, 0: nothing) | row lo (6 bits) | hi + 1 (7 bits)
// — images up to 4096 x 128 (block_box_fits).  Written per surfel by the preprocess; the emission puts it into the
// upper half of every instance word, the tile sort's scatter turns it into the mask of the blocks of the instance's
// tile that the surfel can reach (sls_sort.hip: block_mask_of).
__host__ __device__ inline bool block_box_fits(int W, int H) { return W <= 4096 && H <= 128; }
__device__ __forceinline__ uint32_t make_block_box(float cx, float cy, float ex, float ey, int NC)
{
    if (!(ex >= 0.0f && ey >= 0.0f)) return 0u;                            // nothing of the surfel can be seen
    const float lo = ceilf((cx - ex - 7.01f) * 0.125f), hi = floorf((cx + ex + 0.01f) * 0.125f);
    const int n = (int)fminf(hi - lo, (float)(NC - 1));                    // (NC - 1: every column)
    int c0 = (int)fmaxf(fminf(lo, 1.0e6f), -1.0e6f) % NC;
    c0 += c0 < 0 ? NC : 0;
    const int r0 = (int)fminf(fmaxf(ceilf((cy - ey - 1.01f) * 0.5f), 0.0f), 63.0f);
    const int r1 = (int)fminf(floorf((cy + ey + 0.01f) * 0.5f), 63.0f);
    if (n < 0 || r1 < r0) return 0u;
    return (uint32_t)c0 | ((uint32_t)(n + 1) << 9) | ((uint32_t)r0 << 19) | ((uint32_t)(r1 + 1) << 25);
}

// Ballot / all over the wave straight from the condition's lane mask.  HIP's __ballot() goes through an integer
// compare of the zero-extended predicate: where the predicate is the result of scalar mask logic the compiler
// materialises it as 0/1 in a VGPR and compares again — two half-rate VALU instructions per ballot in the tile
// kernels' step loops (v_cndmask_b32 + v_cmp_ne_u32, ~4 clocks each on gfx950: profiles/r03a_valu_calibration.json).
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

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

// lane <- lane ^ 4 inside a row (two bank-masked DPP moves).
__device__ __forceinline__ float dpp_xor4(float v)
{
    int a = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);   // row_shl:4 -> banks 0,2
    a = __builtin_amdgcn_update_dpp(a, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);       // row_shr:4 -> banks 1,3
    return __builtin_bit_cast(float, a);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }

// a <- [a.lo | b.lo], b <- [a.hi | b.hi] (halves of 32 lanes / of 16 lanes in each half): the gfx950
// v_permlane32_swap / v_permlane16_swap through the compiler's builtins — it can use the halves of a register
// pair as operands directly and places the hazard no-ops itself (inline asm cost a v_mov per half and fixed
// s_nops).
__device__ __forceinline__ void permlane32_swap(float &a, float &b)
{
    const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
__device__ __forceinline__ void permlane16_swap(float &a, float &b)
{
    const v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}

// Reduce-scatter of 16 per-lane values over the 16 PIXELS of each slot of a pixel block (lane = 4 * pixel +
// slot): lane (pixel p, slot s) receives the sum over the 16 lanes of slot s of component p.  The values come
// as 8 adjacent pairs (component 2j, 2j+1 = x[j]).  Each level halves the number of live values by exchanging
// one half with a partner lane: lane^32 and lane^16 with the swaps (one swap serves two values, no selects; the
// adds of these two levels are packed: 6 v_pk_add_f32), lane^8 (row_ror:8) and lane^4 with selects + DPP.
__device__ __forceinline__ float block_reduce16_pk(const v2f (&x)[8], int lane)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[2 * i] = x[i].x; a[2 * i + 1] = x[i].y; }
#pragma unroll
    for (int i = 0; i < 8; ++i) permlane32_swap(a[i], a[i + 8]);
    v2f y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = mk2(a[2 * i], a[2 * i + 1]) + mk2(a[2 * i + 8], a[2 * i + 9]);
    float b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[2 * i] = y[i].x; b[2 * i + 1] = y[i].y; }
#pragma unroll
    for (int i = 0; i < 4; ++i) permlane16_swap(b[i], b[i + 4]);
    const v2f z01 = mk2(b[0], b[1]) + mk2(b[4], b[5]), z23 = mk2(b[2], b[3]) + mk2(b[6], b[7]);
    const float z[4] = { z01.x, z01.y, z23.x, z23.y };
    // The last two levels (partners lane ^ 8 and lane ^ 4 inside a row of 16): which half of its values a lane keeps is
    // a property of its BANK (lanes 4b .. 4b + 3 of a row) — bit 3 of the lane is set in banks 2, 3, bit 2 in banks 1, 3 —
    // so each level is two bank-masked DPP adds per output, `kept + partner's` written only where the bank keeps that
    // value: 6 instructions for both levels where selects + DPP moves took 12 (the same sums, operands swapped at most).
    // (ONE statement: a DPP operand written by one of the two instructions in front needs the wait states the s_nops
    //  supply — the compiler does not see inside — and six separate statements cost eleven of them)
    float w0 = z[0], w1 = z[1];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"      // banks 0, 1 keep values 0, 1 ...
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"      // ... banks 2, 3 values 2, 3
        "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"      // banks 0, 2 keep w0 (partner: lane + 4)
        "v_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"           // banks 1, 3 keep w1 (partner: lane - 4)
        : "+v"(w0), "+v"(w1) : "v"(z[2]), "v"(z[3]));
    const float r = w0;
    (void)lane;
    return r;
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

// Predicted scale of the one-pass deterministic accumulation: biased exponent of |sum| + 12, one byte (0: none).
// With it the fixed point keeps 27 bits below the previous sum's magnitude and has room for terms 2^22 times larger
// (|q| < 2^50 is checked per contribution; beyond it the iteration is void and repeated with the two-pass scheme).
__device__ __forceinline__ uint32_t det_predict(float sum)
{
    const uint32_t e = (__float_as_uint(sum) >> 23) & 0xFFu;
    return sum == 0.0f ? 0u : min(max(e + 12u, 1u), 254u);
}
constexpr uint32_t kDetMispredicted = 8u;        // bit 3 of SlsMappingStatus.overflow
// The scale a one-launch iteration uses for a (surfel, field): its own prediction, but never more than 24 bits below
// the field's default (the largest prediction any surfel had in a two-launch iteration).  A sum that all but cancelled
// last time says nothing about the size of its terms: without the floor such elements overflowed their fixed point a
// few times per hundred iterations at 500 k surfels (8 voided iterations in 250); with it a contribution has to reach
// half the field's largest SUM to do so.  gex = 0 (no two-launch iteration yet): no scale, every contribution flags.
__device__ __forceinline__ int det_scale_exp(uint32_t own, uint32_t gex)
{
    return (int)max(own, gex > 24u ? gex - 24u : (gex ? 1u : 0u));
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
    // (hardware 1-ulp square root and reciprocals: an error of 1e-7 of the step, i.e. of lr-sized changes)
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(a.bc2_sqrt) + a.eps;
    p = p - step_size * (m * __builtin_amdgcn_rcpf(denom));
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
    // one-pass variant (DET = 3): the scale of every (surfel, field) is PREDICTED — det_prev[surfel][field], one byte, is
    // the biased exponent of the field's sum in the keyframe's previous iteration + 12 (0: no history: the field's
    // default det_gex[field], set by two-pass iterations only).  det_onepass = 1: det_acc is scaled back with the
    // prediction and cleared where read; in every deterministic mode the new prediction is written (iteration not void).
    uint8_t *det_prev;
    uint32_t *det_gex;
    int det_onepass;
    float *reg_accum;                     // optional: workspace scalar holding this iteration's regulariser sum
    uint64_t *grad_bitmap;                // optional (sparse exchange): bit per surfel with a non-zero gradient + 2 verdict words
    int grad_bitmap_words;                // (N + 63) / 64
    // kernel B inside the backward tile kernel (sls_render_block.hip, FUSED = 2): its blocks' loss terms, three floats per
    // block, are summed here — one wave of block 0, a fixed order — into words 2..5 of the status block before thread 0
    // publishes it (loss_w: 1 / pixels, lambda_n / valid pixels, lambda_a / valid pixels)
    const float *loss_partials;
    int n_loss_partials;
    float loss_w[3];
    // and the launch order of the keyframe's NEXT tile backward (sls_consumer_dev.hpp: order_blocks_by_cost), by eight
    // passenger workgroups behind the surfels': order_out[0] = the tag, the T * 16 block indices behind it
    int order_T;
    const uint32_t *order_cost;
    uint32_t *order_out;
    int publisher;          // set by launch_preprocess_bwd: a workgroup of its own does the status duties above
    // touched-set exchange, second half (SlsMappingConfig.union_bitmap): a surfel of the union writes its 10 gradient values
    // into its slot of `compact` and keeps its parameters; every other surfel takes the fused Adam update (enabled = 1)
    const uint64_t *union_bitmap;
    const uint32_t *union_prefix;
    float *compact;
    uint32_t *compact_idx;
    uint32_t compact_cap;
};

// Copy of an iteration's finished status block (8 words) into its pinned host mirror, by ONE lane.  The host polls
// words 0 and 7 (engine.py: it arms them with a value the device never writes): words 0..6 are made visible at
// system scope BEFORE word 7 is stored, so a host that sees word 7 sees the whole row.
__device__ __forceinline__ void mirror_status_block(const uint32_t *src, uint32_t *mirror, bool override1 = false,
                                                    uint32_t word1 = 0u)
{
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = src[k];
    if (override1) w[1] = word1;
#pragma unroll
    for (int k = 0; k < 7; ++k) __builtin_nontemporal_store(w[k], mirror + k);
    __threadfence_system();
    __builtin_nontemporal_store(w[7], mirror + 7);
}

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
