// sls_knn.hip — simple-knn's distCUDA2 for MI355X:
// out[i] = mean of the squared distances from point i to its 3 nearest
// other points (slam/mapper.py:113-115, scene/gaussian_model.py:77-81).
//
// Exact 3-NN, built for LiDAR-like (very non-uniform) clouds:
//   1. bounding cube, 30-bit Hilbert-curve index per point;
//   2. wave64 LSD radix sort of (code, index) — the sorter of sls_sort.hip;
//   3. points gathered in curve order (float4: xyz + original index) and
//      axis-aligned boxes over runs of 256 consecutive points;
//   4. one wave per 64 consecutive points, one lane per point: the own box first for a
//      tight bound, then boxes / their 32-point sub-boxes whose gap to the wave's points
//      beats the wave's worst third-best are listed (64 tests per step), and each listed
//      sub-box is staged through LDS if it can still improve some lane's own bound.
// Squared distances use the fixed expression fma(dz,dz,fma(dy,dy,dx*dx)), so
// the result is reproducible bit for bit by a CPU brute force.
#include <float.h>

#include "sls_common.hpp"

namespace sls {

size_t sort_scratch_bytes(uint64_t R);
int radix_sort_pairs_u32(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                         const uint32_t *count_ptr, uint32_t cap, int nbits, void *scratch, size_t scratch_bytes,
                         int *result_in_tmp, hipStream_t st);

constexpr int kKnnBox = 256;
constexpr int kKnnSub = 32;    // points per sub-box (8 per box)

__device__ __forceinline__ uint32_t f2ord(float f)
{   // monotone float -> uint mapping
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

__global__ void knn_init_bbox_kernel(uint32_t *bbox, uint32_t M)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xFFFFFFFFu;       // min (ordered domain)
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;           // max
    else if (threadIdx.x == 6) bbox[6] = M;                     // item count for the sorter
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(int M, const float *__restrict__ xyz, uint32_t *bbox)
{
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * i + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    }
    // block combine in LDS, then 6 global atomics per block (same-address global atomics serialise)
    __shared__ uint32_t s_box[6];
    if (threadIdx.x < 6) s_box[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            atomicMin(&s_box[k], f2ord(mn[k]));
            atomicMax(&s_box[3 + k], f2ord(mx[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bbox[threadIdx.x], s_box[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bbox[threadIdx.x], s_box[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{   // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(int M, const float *__restrict__ xyz,
                                                         const uint32_t *__restrict__ bbox,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, int key_bits)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float mnx = ord2f(bbox[0]), mny = ord2f(bbox[1]), mnz = ord2f(bbox[2]);
    const float ext = fmaxf(fmaxf(ord2f(bbox[3]) - mnx, ord2f(bbox[4]) - mny), fmaxf(ord2f(bbox[5]) - mnz, 1e-30f));
    const float s = 1024.0f / ext;
    const uint32_t qx = (uint32_t)fminf(fmaxf((xyz[3 * i] - mnx) * s, 0.0f), 1023.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((xyz[3 * i + 1] - mny) * s, 0.0f), 1023.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf((xyz[3 * i + 2] - mnz) * s, 0.0f), 1023.0f);
    // Hilbert index of the cell (Skilling's axes-to-transpose form, 10 bits per axis): unlike the Morton
    // order it has no jumps, so runs of consecutive points (the boxes below) stay compact in space.
    uint32_t X[3] = { qx, qy, qz };
#pragma unroll
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (X[a] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    uint32_t t = 0;
#pragma unroll
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    // small clouds are sorted on the top 21 bits only (two 11-bit radix passes instead of three 10-bit ones;
    // points of one 2^-7 cell stay in input order): measured faster up to ~200 k points, slower beyond
    keys[i] = ((spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2])) >> (30 - key_bits);
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void knn_gather_boxes_kernel(int M, const float *__restrict__ xyz,
                                                               const uint32_t *__restrict__ sorted_idx,
                                                               float4 *__restrict__ pts, float4 *__restrict__ boxes,
                                                               float4 *__restrict__ subboxes)
{
    // one block per box of kKnnBox == blockDim.x points
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int j = blockIdx.x * kKnnBox + threadIdx.x;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (j < M) {
        const uint32_t i = sorted_idx[j];
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        pts[j] = make_float4(x, y, z, __uint_as_float(i));
        mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    }
    if ((threadIdx.x & (kKnnSub - 1)) == 0) {   // tight boxes over runs of 32 points (empty run: min > max)
        const size_t sb = (size_t)blockIdx.x * (kKnnBox / kKnnSub) + threadIdx.x / kKnnSub;
        subboxes[2 * sb] = make_float4(mn[0], mn[1], mn[2], 0.0f);
        subboxes[2 * sb + 1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        mn[k] = fminf(mn[k], __shfl_xor(mn[k], 32, 64));
        mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], 32, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s_mn[wave][k] = mn[k]; s_mx[wave][k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k)
            for (int w = 1; w < 4; ++w) { s_mn[0][k] = fminf(s_mn[0][k], s_mn[w][k]); s_mx[0][k] = fmaxf(s_mx[0][k], s_mx[w][k]); }
        boxes[2 * blockIdx.x] = make_float4(s_mn[0][0], s_mn[0][1], s_mn[0][2], 0.0f);
        boxes[2 * blockIdx.x + 1] = make_float4(s_mx[0][0], s_mx[0][1], s_mx[0][2], 0.0f);
    }
}

__device__ __forceinline__ void best3_update(float d2, float &b0, float &b1, float &b2)
{   // insertion into the sorted triple b0 <= b1 <= b2, branch-free (nested ifs end up as an indexed
    // stack array here: scratch traffic in the innermost loop)
    const float u0 = fmaxf(b0, d2);
    b0 = fminf(b0, d2);
    const float u1 = fmaxf(b1, u0);
    b1 = fminf(b1, u0);
    b2 = fminf(b2, u1);
}

// One WAVE per 64 curve-consecutive query points (lane = point), no block barriers: every wave runs
// its own search at its own pace and 8 of them share a SIMD.
//   a. the 8 runs of 32 points of the own box (256 points): a tight third-best b2 per lane;
//   b. boxes (256 points) are tested 64 at a time, lane = box: gap between the box and the bounding
//      box of each group of 8 consecutive query points against that group's worst b2 (isolated
//      points and sparse stretches of the curve must not widen everybody's search); for the survivors the
//      8 sub-boxes (32 points, much tighter: even a Hilbert run of 256 is loose) are tested the same
//      way, 8 boxes per step; surviving sub-boxes go to a wave-private LDS list.  Chunks of 64 boxes
//      are visited outwards from the own box, so b2 shrinks early;
//   c. every listed sub-box is re-tested per lane against the lane's own b2 (point-to-box gap);
//      if any lane can still improve, its 32 points go through LDS (broadcast reads) to all lanes.
// All bounds are conservative (relative slack 1e-5), so the result is the exact 3-NN.
template <bool SELF>
__device__ __forceinline__ void knn_scan32(const float4 *s_run, const float4 me, int self, float &b0, float &b1, float &b2)
{
#pragma unroll 8
    for (int k = 0; k < kKnnSub; ++k) {
        const float4 p = s_run[k];
        const float dx = p.x - me.x, dy = p.y - me.y, dz = p.z - me.z;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        best3_update((SELF && k == self) ? FLT_MAX : d2, b0, b1, b2);
    }
}

__device__ __forceinline__ float box_gap2(const float4 amn, const float4 amx, const float4 bmn, const float4 bmx)
{
    const float gx = fmaxf(fmaxf(bmn.x - amx.x, amn.x - bmx.x), 0.0f);
    const float gy = fmaxf(fmaxf(bmn.y - amx.y, amn.y - bmx.y), 0.0f);
    const float gz = fmaxf(fmaxf(bmn.z - amx.z, amn.z - bmx.z), 0.0f);
    return fmaf(gz, gz, fmaf(gy, gy, gx * gx));
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(64) void knn_query_kernel(int M, int nboxes, const float4 *__restrict__ pts,
                                                       const float4 *__restrict__ boxes,
                                                       const float4 *__restrict__ subboxes, float *__restrict__ out, uint32_t Mq)
{
    constexpr int kSubs = kKnnBox / kKnnSub;            // 8 sub-boxes per box
    __shared__ float4 s_run[kKnnSub];                   // wave-private: the workgroup IS one wave
    __shared__ uint32_t s_box[64];
    __shared__ uint32_t s_sub[64 * kSubs];
    const int lane = threadIdx.x;
    const int q = blockIdx.x * 64 + lane;
    const float4 me = pts[q < M ? q : M - 1];
    // Mq < M (sls_knn_dist2_first): only the points whose ORIGINAL index is below Mq are queries — Mapper.densify asks
    // for the new surfels' distances among new + existing centres (slam/mapper.py:109-117) and drops the rest.  A wave
    // of 64 neighbours on the curve without one leaves; the others search exactly as they would
    const bool live = q < M && __float_as_uint(me.w) < Mq;
    if (!__ballot(live)) return;
    const int own_box = (blockIdx.x * 64) / kKnnBox, own_sub0 = (blockIdx.x * 64) / kKnnSub;   // my runs: own_sub0, +1
    const int nsub = (M + kKnnSub - 1) / kKnnSub;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;

    auto stage = [&](int sb) {      // 32 points of run sb -> LDS (far-away sentinels beyond M)
        __builtin_amdgcn_wave_barrier();
        if (lane < kKnnSub) {
            const int k = sb * kKnnSub + lane;
            s_run[lane] = k < M ? pts[k] : make_float4(1.0e30f, 1.0e30f, 1.0e30f, 0.0f);
        }
        __syncthreads();
    };

    // a. own box: my two runs first, then the other six
    for (int t = 0; t < kSubs; ++t) {
        const int sb = own_box * kSubs + ((t + (own_sub0 % kSubs)) % kSubs);
        if (sb >= nsub) continue;
        stage(sb);
        const int self = (live && (q / kKnnSub) == sb) ? (q % kKnnSub) : -1;
        knn_scan32<true>(s_run, me, self, b0, b1, b2);
    }

    // bounding box of the wave's own points
    float4 wmn = subboxes[2 * (size_t)own_sub0], wmx = subboxes[2 * (size_t)own_sub0 + 1];
    if (own_sub0 + 1 < nsub) {
        const float4 mn1 = subboxes[2 * (size_t)(own_sub0 + 1)], mx1 = subboxes[2 * (size_t)(own_sub0 + 1) + 1];
        wmn = make_float4(fminf(wmn.x, mn1.x), fminf(wmn.y, mn1.y), fminf(wmn.z, mn1.z), 0.0f);
        wmx = make_float4(fmaxf(wmx.x, mx1.x), fmaxf(wmx.y, mx1.y), fmaxf(wmx.z, mx1.z), 0.0f);
    }

    // bounding boxes of the 8 groups of 8 consecutive points (dead lanes: empty)
    auto rl = [](float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); };
    float4 gmn = live ? me : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.0f);
    float4 gmx = live ? me : make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.0f);
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        gmn.x = fminf(gmn.x, __shfl_xor(gmn.x, off, 64)); gmn.y = fminf(gmn.y, __shfl_xor(gmn.y, off, 64));
        gmn.z = fminf(gmn.z, __shfl_xor(gmn.z, off, 64));
        gmx.x = fmaxf(gmx.x, __shfl_xor(gmx.x, off, 64)); gmx.y = fmaxf(gmx.y, __shfl_xor(gmx.y, off, 64));
        gmx.z = fmaxf(gmx.z, __shfl_xor(gmx.z, off, 64));
    }
    const int nchunks = (nboxes + 63) / 64, c0 = own_box / 64;
    for (int it = 0; it < nchunks; ++it) {
        // chunks outwards from the own one: c0, c0+1, c0-1, c0+2, ... (those inside [0, nchunks))
        int chunk;
        {
            const int near = min(c0, nchunks - 1 - c0);            // both sides available for 2*near steps
            if (it <= 2 * near) chunk = c0 + ((it & 1) ? (it + 1) / 2 : -(it / 2));
            else chunk = (c0 < nchunks - 1 - c0) ? it : nchunks - 1 - it;
        }
        // Which boxes can reach the wave?  One bound for all 64 points would be ruined by a single isolated
        // point (its b2 can be 1000 x the others') or by a sparse stretch of the curve (a bounding box that
        // swallows the dense regions in between), so the wave is split into 8 groups of 8 consecutive points:
        // a box is a candidate if its gap to a group's bounding box beats that group's worst b2.
        float gB = live ? b2 : 0.0f;
        gB = fmaxf(gB, __shfl_xor(gB, 1, 64)); gB = fmaxf(gB, __shfl_xor(gB, 2, 64)); gB = fmaxf(gB, __shfl_xor(gB, 4, 64));
        const float B2 = wave_max(gB);
        auto reaches = [&](const float4 bmn, const float4 bmx) -> bool {
            if (!(box_gap2(wmn, wmx, bmn, bmx) * 0.99999f <= B2)) return false;
            bool r = false;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 mn = make_float4(rl(gmn.x, 8 * g), rl(gmn.y, 8 * g), rl(gmn.z, 8 * g), 0.0f);
                const float4 mx = make_float4(rl(gmx.x, 8 * g), rl(gmx.y, 8 * g), rl(gmx.z, 8 * g), 0.0f);
                r = r || (box_gap2(mn, mx, bmn, bmx) * 0.99999f <= rl(gB, 8 * g));
            }
            return r;
        };
        const int b = chunk * 64 + lane;
        bool c = false;
        if (b < nboxes && b != own_box) c = reaches(boxes[2 * b], boxes[2 * b + 1]);
        const uint64_t cm = __ballot(c);
        const int nb = __builtin_popcountll(cm);
        if (nb == 0) continue;
        __builtin_amdgcn_wave_barrier();
        if (c) s_box[__builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u))] = (uint32_t)b;
        __syncthreads();
        // their sub-boxes, 8 boxes per step: lane = (box slot, sub-box)
        int ns = 0;
        for (int g = 0; g < nb; g += 8) {
            const int slot = g + (lane >> 3);
            bool cs = false;
            uint32_t sb = 0;
            if (slot < nb) {
                sb = s_box[slot] * kSubs + (uint32_t)(lane & 7);
                if ((int)sb < nsub) cs = reaches(subboxes[2 * (size_t)sb], subboxes[2 * (size_t)sb + 1]);
            }
            const uint64_t sm = __ballot(cs);
            if (cs) s_sub[ns + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u))] = sb;
            ns += __builtin_popcountll(sm);
        }
        __syncthreads();
        // c. 64 listed sub-boxes at a time, lane = candidate: the exact test — can it improve ANY of the 64
        //    query points? — runs for all 64 candidates at once (the queries are broadcast with v_readlane);
        //    the survivors are scanned one after the other, the next one's points are fetched meanwhile.
        for (int base = 0; base < ns; base += 64) {
            const int nk = min(64, ns - base);
            const uint32_t sbl = s_sub[base + min(lane, nk - 1)];
            const float4 mnl = subboxes[2 * (size_t)sbl], mxl = subboxes[2 * (size_t)sbl + 1];
            const float qb2 = live ? b2 : -1.0f;
            bool need = false;
            for (int qi = 0; qi < 64; ++qi) {
                const float4 qp = make_float4(rl(me.x, qi), rl(me.y, qi), rl(me.z, qi), 0.0f);
                need = need || (box_gap2(qp, qp, mnl, mxl) * 0.99999f <= rl(qb2, qi));
            }
            uint64_t todo = __ballot(need && lane < nk);
            auto fetch = [&](int k) -> float4 {     // lanes 0..31: the points of candidate k (sentinels beyond M)
                const int i = __builtin_amdgcn_readlane((int)sbl, k) * kKnnSub + (lane & (kKnnSub - 1));
                return i < M ? pts[i] : make_float4(1.0e30f, 1.0e30f, 1.0e30f, 0.0f);
            };
            float4 pre = make_float4(0, 0, 0, 0);
            if (todo) pre = fetch(__builtin_ctzll(todo));
            while (todo) {
                todo &= todo - 1;
                const float4 cur = pre;
                if (todo) pre = fetch(__builtin_ctzll(todo));
                __builtin_amdgcn_wave_barrier();
                if (lane < kKnnSub) s_run[lane] = cur;
                __syncthreads();
                knn_scan32<false>(s_run, me, -1, b0, b1, b2);    // (a lane that did not need it cannot be changed by it)
            }
        }
    }
    if (live) out[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

// scratch layout (all 256-byte aligned)
struct KnnScratch {
    uint32_t *bbox;
    uint32_t *keys, *keys_tmp;
    uint32_t *vals, *vals_tmp;
    float4 *pts, *boxes, *subboxes;
    void *sort;
    size_t sort_bytes, total;
};

static KnnScratch knn_layout(int M, void *base)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    KnnScratch s;
    char *p = (char *)base;
    size_t off = 0;
    const int nboxes = (M + kKnnBox - 1) / kKnnBox;
    s.bbox = (uint32_t *)(p + off); off += al(8 * sizeof(uint32_t));
    s.keys = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.keys_tmp = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.vals = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.vals_tmp = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.pts = (float4 *)(p + off); off += al(sizeof(float4) * (size_t)M);
    s.boxes = (float4 *)(p + off); off += al(sizeof(float4) * 2 * (size_t)nboxes);
    s.subboxes = (float4 *)(p + off); off += al(sizeof(float4) * 2 * (size_t)nboxes * (kKnnBox / kKnnSub));
    s.sort = (void *)(p + off);
    s.sort_bytes = sort_scratch_bytes((uint64_t)M);
    off += al(s.sort_bytes);
    s.total = off;
    return s;
}

size_t knn_scratch_bytes(int M) { return M > 0 ? knn_layout(M, nullptr).total : 0; }

int launch_knn(int M, const float *xyz, float *out, void *scratch, size_t scratch_bytes, hipStream_t st, int Mq)
{
    const KnnScratch s = knn_layout(M, scratch);
    if (scratch_bytes < s.total) {
        set_error("knn scratch too small: %zu < %zu", scratch_bytes, s.total);
        return SLS_E_SCRATCH;
    }
    if (((uintptr_t)scratch & 255) != 0) {
        set_error("knn scratch must be 256-byte aligned");
        return SLS_E_ARG;
    }
    const int nb = (M + 255) / 256;
    const int key_bits = M <= 200000 ? 21 : 30;
    ScopedTimer tm(T_KNN, st);
    hipLaunchKernelGGL(knn_init_bbox_kernel, dim3(1), dim3(64), 0, st, s.bbox, (uint32_t)M);
    SLS_LAUNCH_CHECK("knn_init_bbox_kernel");
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 256 ? nb : 256), dim3(256), 0, st, M, xyz, s.bbox);
    SLS_LAUNCH_CHECK("knn_bbox_kernel");
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, st, M, xyz, s.bbox, s.keys, s.vals, key_bits);
    SLS_LAUNCH_CHECK("knn_morton_kernel");
    int which = 0;
    int rc = radix_sort_pairs_u32(s.keys, s.vals, s.keys_tmp, s.vals_tmp, s.bbox + 6, (uint32_t)M, key_bits, s.sort,
                                  s.sort_bytes, &which, st);
    if (rc) return rc;
    const int nboxes = (M + kKnnBox - 1) / kKnnBox;
    hipLaunchKernelGGL(knn_gather_boxes_kernel, dim3(nboxes), dim3(kKnnBox), 0, st, M, xyz,
                       which ? s.vals_tmp : s.vals, s.pts, s.boxes, s.subboxes);
    SLS_LAUNCH_CHECK("knn_gather_boxes_kernel");
    static_assert(kKnnBox == 256 && kKnnSub == 32, "the query kernel's lane mappings are written for 8 runs of 32");
    hipLaunchKernelGGL(knn_query_kernel, dim3((M + 63) / 64), dim3(64), 0, st, M, nboxes, s.pts, s.boxes, s.subboxes, out,
                       (uint32_t)(Mq < 0 || Mq > M ? M : Mq));
    SLS_LAUNCH_CHECK("knn_query_kernel");
    return SLS_OK;
}

}  // namespace sls
