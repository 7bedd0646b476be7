// sls_exchange.hip — keyframe-parallel exchange of the TOUCHED SET only.
//
// A keyframe's pixels reach ~10 % of the surfels of a local model; only those (plus the few whose scale the
// regulariser pushes on) have a non-zero gradient.  The dense exchange moves 40 B x N per rank and iteration all
// the same.  Here:
//   1. preprocess_bwd leaves, next to the flat gradient bucket, a BITMAP of the surfels with a non-zero gradient
//      (one ballot per wave: N / 8 bytes) followed by two verdict words (non-zero: this rank voids the iteration);
//   2. the bitmaps are all-gathered (G x N / 8 bytes; RCCL offers no bitwise-OR reduction) and OR-ed here,
//      inside sls_grad_compact's first kernel;
//   3. sls_grad_compact packs the 10 gradient values of every surfel of the UNION, in surfel order, into
//      compact[slot][10] — the same slots on every rank, because the union bitmap is the same;
//   4. the first K_send slots are SUM-reduced (K_send is a host-side capacity; K itself stays on the device:
//      K > K_send sets bit 2 of status.overflow, the iteration is void and repeated with more room — the protocol
//      of the instance buffers);
//   5. sls_adam_step_sparse applies torch.optim.Adam to EVERY surfel (moments decay where the gradient is zero),
//      reading the gradient from its slot or taking zero.
// The form MappingEngine runs (round 6) never has the flat bucket: step 1's bitmap is the EARLY one (touched_bitmap_kernel,
// behind the tile backward: a superset), steps 2-3 shrink to the OR + prefix (sls_grad_union), and the projection's backward
// itself writes the union's rows into their slots and applies Adam to every surfel OUTSIDE the union (zero gradient on
// every rank: SlsMappingConfig.union_bitmap); step 5 then only touches the union (part = 2).
// Every rank ends with bit-identical parameters: the collective's result is identical everywhere and the update
// is a pure function of it.  Results equal the dense all-reduce path's to the bit (same operands per element).
#include <string.h>

#include "sls_common.hpp"

namespace sls {

constexpr uint32_t kExchangeTooSmall = 4u;      // bit 2 of SlsMappingStatus.overflow

// exclusive prefix of popcount(bitmap[w]) over the words; ONE workgroup (N / 64 words: 7.8 k at 500 k surfels).  The
// union's words are formed with coalesced loads into LDS (62 KB at 500 k), every thread then owns a run of consecutive
// words there — one scan over the threads, two barriers — and the words and their prefixes leave coalesced again:
// two memory round trips in all (round 5's form walked the words 1024 at a time with three barriers per step: 11.6 us
// at 500 k).  Models beyond kPrefixLdsWords * 64 surfels take the same steps through the output arrays instead of LDS.
// Also publishes the union's size and the group's verdict.
constexpr int kPrefixLdsWords = 12288;       // 104 KB of words + 52 KB of prefixes (padded): 786 k surfels
__global__ __launch_bounds__(1024) void exchange_prefix_kernel(int nwords, const uint64_t *__restrict__ maps, int n_maps,
                                                               uint64_t *__restrict__ bitmap,
                                                               uint32_t *__restrict__ word_prefix, uint32_t capacity,
                                                               uint32_t *__restrict__ status_block)
{
    // maps: n_maps bitmaps of nwords + 2 words each (the ranks' all-gathered bitmaps); bitmap: their OR (out)
    extern __shared__ uint64_t s_dyn[];
    __shared__ uint32_t s_wave[16];
    const bool in_lds = nwords <= kPrefixLdsWords;
    const int per = (nwords + 1023) / 1024, w0 = (int)threadIdx.x * per, w1 = min(w0 + per, nwords);
    // (in LDS a thread's run is followed by one pad word when its length is even: the threads' runs then start an odd
    //  number of 8-byte words apart and their reads do not pile onto four banks)
    const int pad = (in_lds && (per & 1) == 0) ? 1 : 0;
    uint64_t *const words = in_lds ? s_dyn : bitmap;
    uint32_t *const pref = in_lds ? reinterpret_cast<uint32_t *>(s_dyn + nwords + 1024) : word_prefix;
    // (w / per without a division — 40 instructions each, 36 of them per thread made this kernel 10 us: w < 2^20, per <= 2^10)
    const uint32_t magic = (uint32_t)((0x100000000ull + (uint64_t)per - 1ull) / (uint64_t)per);
#define SLS_PW(w_) ((w_) + (pad ? (int)__umulhi((uint32_t)(w_), magic) : 0))
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)(nwords + 2);
    // (every load of a batch in flight before the first is used: a single workgroup pays a full memory round trip per
    //  dependent step — the plain loop over the words took eight of them)
    constexpr int kBatch = 12;        // words per thread and batch: kPrefixLdsWords / 1024
    for (int base = 0; base < nwords; base += 1024 * kBatch) {
        uint64_t acc[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) acc[k] = 0ull;
        for (int m = 0; m < n_maps; m += 2) {
            uint64_t v0[kBatch], v1[kBatch];
            const bool two = m + 1 < n_maps;
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int w = base + k * 1024 + (int)threadIdx.x;
                v0[k] = w < nwords ? maps[(size_t)m * stride + (size_t)w] : 0ull;
                v1[k] = (two && w < nwords) ? maps[(size_t)(m + 1) * stride + (size_t)w] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) acc[k] |= v0[k] | v1[k];
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int w = base + k * 1024 + (int)threadIdx.x;
            if (w < nwords) words[SLS_PW(w)] = acc[k];
        }
    }
    if (!in_lds) __threadfence();
    __syncthreads();
    uint32_t c = 0u;
    const int own = w0 + (pad ? (int)threadIdx.x : 0);          // SLS_PW(w0): the run is contiguous in LDS
    for (int w = w0; w < w1; ++w) c += (uint32_t)__popcll(words[own + (w - w0)]);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - c;
    for (int k = 0; k < wave; ++k) pre += s_wave[k];
    for (int w = w0; w < w1; ++w) {
        pref[own + (w - w0)] = pre;
        pre += (uint32_t)__popcll(words[own + (w - w0)]);
    }
    if (threadIdx.x == 1023) {
        const uint32_t K = pre;
        uint64_t v0 = 0ull, v1 = 0ull;
        for (int m = 0; m < n_maps; ++m) {
            v0 |= maps[(size_t)m * stride + (size_t)nwords];
            v1 |= maps[(size_t)m * stride + (size_t)nwords + 1];
        }
        bitmap[nwords] = v0; bitmap[nwords + 1] = v1;
        const uint32_t void_bits = (v0 ? 1u : 0u) | (v1 ? 2u : 0u);   // the group's verdict
        status_block[7] = K;                                          // SlsMappingStatus.exchange_count
        status_block[1] = void_bits | (K > capacity ? kExchangeTooSmall : 0u);
    }
    if (in_lds) {
        __syncthreads();
        for (int w = threadIdx.x; w < nwords; w += 1024) { bitmap[w] = words[SLS_PW(w)]; word_prefix[w] = pref[SLS_PW(w)]; }
    }
#undef SLS_PW
}

// thread per surfel: the surfels of the union copy their 10 gradient values into their slot
__global__ __launch_bounds__(256) void exchange_compact_kernel(int N, const uint64_t *__restrict__ bitmap,
                                                               const uint32_t *__restrict__ word_prefix,
                                                               const float *__restrict__ grads, float *__restrict__ compact,
                                                               uint32_t capacity)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint64_t word = bitmap[i >> 6];
    const int b = i & 63;
    if (!((word >> b) & 1ull)) return;
    const uint32_t slot = word_prefix[i >> 6] + (uint32_t)__popcll(word & ((1ull << b) - 1ull));
    if (slot >= capacity) return;
    const size_t n = (size_t)N;
    float *o = compact + (size_t)slot * 10;
    o[0] = grads[3 * (size_t)i]; o[1] = grads[3 * (size_t)i + 1]; o[2] = grads[3 * (size_t)i + 2];
    o[3] = grads[3 * n + i];
    o[4] = grads[4 * n + 2 * (size_t)i]; o[5] = grads[4 * n + 2 * (size_t)i + 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[6 + k] = grads[6 * n + 4 * (size_t)i + k];
}

// The bitmap EARLY (SlsMappingConfig.phase = 1): before the projection's backward has produced a single gradient, the
// set of surfels that CAN get one is known — those the tile backward reached (its `touched` marks) and those whose larger
// raw scale is within a margin of the regulariser's threshold (the backward decides that with its own exp(); the margin of
// 1e-3 in log scale is four orders above its rounding).  A superset of the non-zero gradients: the extra rows are zeros.
__global__ __launch_bounds__(256) void touched_bitmap_kernel(int N, const uint8_t *__restrict__ touched,
                                                             const float2 *__restrict__ scaling_raw, float log_smax_margin,
                                                             int reg_on, const uint32_t *__restrict__ status_block,
                                                             uint64_t *__restrict__ bitmap, int nwords)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool on = false;
    if (i < N) {
        on = touched[i] != 0;
        if (reg_on) { const float2 s = scaling_raw[i]; on = on || fmaxf(s.x, s.y) >= log_smax_margin; }
    }
    const uint64_t word = __ballot(on);
    if ((threadIdx.x & 63) == 0 && i < N) bitmap[i >> 6] = word;
    if (i == 0) {      // this rank's verdict behind the bitmap (all of it is known since the binning)
        const uint32_t bits = status_block[1];
        bitmap[nwords] = (bits & 1u) ? 1ull : 0ull;
        bitmap[nwords + 1] = (bits & ~1u) ? 1ull : 0ull;
    }
}
int launch_touched_bitmap(int N, const uint8_t *touched, const float *scaling_raw, float smax, float pen,
                          const uint32_t *status_block, uint64_t *bitmap, hipStream_t st)
{
    hipLaunchKernelGGL(touched_bitmap_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, touched, (const float2 *)scaling_raw,
                       logf(fmaxf(smax, 1e-30f)) - 1e-3f, pen != 0.0f ? 1 : 0, status_block, bitmap, (N + 63) / 64);
    SLS_LAUNCH_CHECK("touched_bitmap_kernel");
    return SLS_OK;
}

struct SparseAdamArgs {
    float *xyz, *opacity, *scaling, *rotation;
    float *exp_avg, *exp_avg_sq;          // flat buckets [xyz 3N | opacity N | scaling 2N | rotation 4N]
    float lr_xyz, lr_opacity, lr_scaling, lr_rotation;
    AdamCoef c;
};

// torch.optim.Adam on every surfel; the gradient comes from the reduced compact buffer (union surfels) or is zero.
// Skipped as a whole when the group voided the iteration; the last kernel of the iteration: mirrors the status.
// part: 0 = every surfel, 1 = only those outside the union (zero gradient on every rank), 2 = only the union's
__global__ __launch_bounds__(256) void exchange_adam_kernel(int N, SparseAdamArgs a, const uint64_t *__restrict__ bitmap,
                                                            const uint32_t *__restrict__ word_prefix,
                                                            const float *__restrict__ compact,
                                                            const uint32_t *__restrict__ status_block,
                                                            uint32_t *__restrict__ status_mirror, int part)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && status_mirror && part != 1) mirror_status_block(status_block, status_mirror);
    if (status_block[1] != 0u) return;        // void: the verdict of the group or an exchange buffer too small
    if (i >= N) return;
    float g[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) g[k] = 0.0f;
    const uint64_t word = bitmap[i >> 6];
    const int b = i & 63;
    const bool in_union = (word >> b) & 1ull;
    if ((part == 1 && in_union) || (part == 2 && !in_union)) return;
    if (in_union) {
        const uint32_t slot = word_prefix[i >> 6] + (uint32_t)__popcll(word & ((1ull << b) - 1ull));
        const float *s = compact + (size_t)slot * 10;
#pragma unroll
        for (int k = 0; k < 10; ++k) g[k] = s[k];
    }
    const size_t n = (size_t)N, ix = 3 * (size_t)i, io = 3 * n + i, is = 4 * n + 2 * (size_t)i, ir = 6 * n + 4 * (size_t)i;
    float *M = a.exp_avg, *V = a.exp_avg_sq;
    const float ibc1 = __builtin_amdgcn_rcpf(a.c.bc1);
    const float st_x = a.lr_xyz * ibc1, st_o = a.lr_opacity * ibc1, st_s = a.lr_scaling * ibc1, st_r = a.lr_rotation * ibc1;
    float p, m, v;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p = a.xyz[ix + k]; m = M[ix + k]; v = V[ix + k];
        adam_one(p, g[k], m, v, st_x, a.c);
        a.xyz[ix + k] = p; M[ix + k] = m; V[ix + k] = v;
    }
    p = a.opacity[i]; m = M[io]; v = V[io];
    adam_one(p, g[3], m, v, st_o, a.c);
    a.opacity[i] = p; M[io] = m; V[io] = v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        p = a.scaling[2 * (size_t)i + k]; m = M[is + k]; v = V[is + k];
        adam_one(p, g[4 + k], m, v, st_s, a.c);
        a.scaling[2 * (size_t)i + k] = p; M[is + k] = m; V[is + k] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p = a.rotation[4 * (size_t)i + k]; m = M[ir + k]; v = V[ir + k];
        adam_one(p, g[6 + k], m, v, st_r, a.c);
        a.rotation[4 * (size_t)i + k] = p; M[ir + k] = m; V[ir + k] = v;
    }
}

// The same update for the union ONLY, a thread per SLOT: surfel = index[slot] (written next to the row by the projection's
// backward).  58 k fully used threads at 500 k surfels / one keyframe per rank instead of 500 k of which one in nine works.
__global__ __launch_bounds__(256) void exchange_adam_union_kernel(uint32_t capacity, size_t n, SparseAdamArgs a,
                                                                  const uint32_t *__restrict__ index,
                                                                  const float *__restrict__ compact,
                                                                  const uint32_t *__restrict__ status_block,
                                                                  uint32_t *__restrict__ status_mirror)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot == 0 && status_mirror) mirror_status_block(status_block, status_mirror);
    if (status_block[1] != 0u) return;        // void: the verdict of the group or an exchange buffer too small
    if (slot >= min(capacity, status_block[7])) return;
    const size_t i = index[slot];
    const size_t off[4] = { 3 * i, 3 * n + i, 4 * n + 2 * i, 6 * n + 4 * i };     // the surfel's place in the flat buckets
    float *const par[4] = { a.xyz + 3 * i, a.opacity + i, a.scaling + 2 * i, a.rotation + 4 * i };
    constexpr int kW[4] = { 3, 1, 2, 4 }, kG[4] = { 0, 3, 4, 6 };
    const float *row = compact + (size_t)slot * 10;
    // every load first (parameters, moments and the row alias as far as the compiler can tell: interleaved with the stores
    // they became four dependent round trips — 15 us for 58 k surfels)
    float g[10], p[10], m[10], v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) g[k] = row[k];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < kW[q]; ++k) {
            p[kG[q] + k] = par[q][k]; m[kG[q] + k] = a.exp_avg[off[q] + k]; v[kG[q] + k] = a.exp_avg_sq[off[q] + k];
        }
    const float ibc1 = __builtin_amdgcn_rcpf(a.c.bc1);
    const float stp[4] = { a.lr_xyz * ibc1, a.lr_opacity * ibc1, a.lr_scaling * ibc1, a.lr_rotation * ibc1 };
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < kW[q]; ++k) adam_one(p[kG[q] + k], g[kG[q] + k], m[kG[q] + k], v[kG[q] + k], stp[q], a.c);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < kW[q]; ++k) {
            par[q][k] = p[kG[q] + k]; a.exp_avg[off[q] + k] = m[kG[q] + k]; a.exp_avg_sq[off[q] + k] = v[kG[q] + k];
        }
}

}  // namespace sls

using namespace sls;

extern "C" {

size_t sls_grad_bitmap_words(int N) { return N > 0 ? (size_t)((N + 63) / 64) + 2 : 2; }

int sls_grad_union(int N, const uint64_t *bitmaps, int n_bitmaps, uint64_t *union_bitmap, uint32_t capacity,
                   uint32_t *word_prefix, SlsMappingStatus *status_dev, void *stream)
{
    return sls_grad_compact(N, bitmaps, n_bitmaps, union_bitmap, nullptr, nullptr, capacity, word_prefix, status_dev, stream);
}

int sls_grad_compact(int N, const uint64_t *bitmaps, int n_bitmaps, uint64_t *union_bitmap, const float *grads_flat,
                     float *compact, uint32_t capacity, uint32_t *word_prefix, SlsMappingStatus *status_dev, void *stream)
{
    SLS_REQUIRE(N > 0 && bitmaps && n_bitmaps >= 1 && union_bitmap && word_prefix && status_dev && (!grads_flat == !compact),
                "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nwords = (N + 63) / 64;
    const size_t lds = nwords <= kPrefixLdsWords ? (size_t)(nwords + 1024) * 12 + 16 : 0;
    static bool lds_allowed = false;
    if (lds > 0 && !lds_allowed) {
        SLS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(exchange_prefix_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (kPrefixLdsWords + 1024) * 12 + 16));
        lds_allowed = true;
    }
    hipLaunchKernelGGL(exchange_prefix_kernel, dim3(1), dim3(1024), lds, st, nwords, bitmaps, n_bitmaps, union_bitmap,
                       word_prefix, capacity, (uint32_t *)status_dev);
    SLS_LAUNCH_CHECK("exchange_prefix_kernel");
    if (!grads_flat) return SLS_OK;       // (sls_grad_union: the projection's backward writes the rows into their slots itself)
    hipLaunchKernelGGL(exchange_compact_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, (const uint64_t *)union_bitmap,
                       (const uint32_t *)word_prefix, grads_flat, compact, capacity);
    SLS_LAUNCH_CHECK("exchange_compact_kernel");
    return SLS_OK;
}

int sls_adam_step_sparse(int N, float *xyz, float *opacity, float *scaling, float *rotation,
                         const uint64_t *union_bitmap, const uint32_t *word_prefix, const float *compact_reduced,
                         float *exp_avg, float *exp_avg_sq, float lr_xyz, float lr_opacity, float lr_scaling,
                         float lr_rotation, double beta1, double beta2, double eps, int64_t step, int part,
                         SlsMappingStatus *status_dev, SlsMappingStatus *status_mirror, void *stream)
{
    SLS_REQUIRE(part >= 0 && part <= 2, "part: 0 every surfel, 1 outside the union, 2 the union");
    SLS_REQUIRE(N > 0 && xyz && opacity && scaling && rotation && union_bitmap && word_prefix && compact_reduced &&
                    exp_avg && exp_avg_sq && status_dev,
                "bad argument");
    SLS_REQUIRE(step >= 1, "step is 1-based");
    SparseAdamArgs a;
    memset(&a, 0, sizeof(a));
    a.xyz = xyz; a.opacity = opacity; a.scaling = scaling; a.rotation = rotation;
    a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq;
    a.lr_xyz = lr_xyz; a.lr_opacity = lr_opacity; a.lr_scaling = lr_scaling; a.lr_rotation = lr_rotation;
    a.c = make_adam_coef(beta1, beta2, eps, step);
    ScopedTimer tm(T_ADAM, (hipStream_t)stream);
    hipLaunchKernelGGL(exchange_adam_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, a, union_bitmap,
                       word_prefix, compact_reduced, (const uint32_t *)status_dev, (uint32_t *)status_mirror, part);
    SLS_LAUNCH_CHECK("exchange_adam_kernel");
    return SLS_OK;
}

int sls_adam_step_union(int N, float *xyz, float *opacity, float *scaling, float *rotation, const uint32_t *compact_index,
                        const float *compact_reduced, uint32_t capacity, float *exp_avg, float *exp_avg_sq, float lr_xyz,
                        float lr_opacity, float lr_scaling, float lr_rotation, double beta1, double beta2, double eps,
                        int64_t step, SlsMappingStatus *status_dev, SlsMappingStatus *status_mirror, void *stream)
{
    SLS_REQUIRE(N > 0 && xyz && opacity && scaling && rotation && compact_index && compact_reduced && capacity > 0 &&
                    exp_avg && exp_avg_sq && status_dev,
                "bad argument");
    SLS_REQUIRE(step >= 1, "step is 1-based");
    SparseAdamArgs a;
    memset(&a, 0, sizeof(a));
    a.xyz = xyz; a.opacity = opacity; a.scaling = scaling; a.rotation = rotation;
    a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq;
    a.lr_xyz = lr_xyz; a.lr_opacity = lr_opacity; a.lr_scaling = lr_scaling; a.lr_rotation = lr_rotation;
    a.c = make_adam_coef(beta1, beta2, eps, step);
    ScopedTimer tm(T_ADAM, (hipStream_t)stream);
    hipLaunchKernelGGL(exchange_adam_union_kernel, dim3((capacity + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, capacity,
                       (size_t)N, a, compact_index, compact_reduced, (const uint32_t *)status_dev, (uint32_t *)status_mirror);
    SLS_LAUNCH_CHECK("exchange_adam_union_kernel");
    return SLS_OK;
}

}  // extern "C"
