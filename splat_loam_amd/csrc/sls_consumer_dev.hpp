// sls_consumer_dev.hpp — device pieces of the allmap consumer shared by sls_consumer.hip (kernels B and C) and
// the backward tile kernel, which can compute a pixel's dL/dallmap itself instead of reading what kernel C
// wrote (one launch and a 7-plane round trip less per iteration).  Maths: sls_consumer.hip's header.
#pragma once
#include "sls_common.hpp"

namespace sls {

struct ConsumerArgs {
    int H, W;
    float depth_ratio, lambda_n, lambda_a;
    float inv_P, inv_nv;      // 1/(H*W), 1/n_valid (0 if n_valid == 0)
    const float *allmap, *gt_depth;
    const uint8_t *valid;
    const float2 *col_h, *row_h;   // half-pixel ray tables
    float4 *du, *dv, *ns;          // scratch: dL/du, dL/dv, (n_surf, dot)
    float *sums;                   // [geom, normal, alpha, total]
    float *partials;               // scratch: 3 floats per block of kernel B (no same-address atomics)
    float *dL_dallmap;
    // optional passenger of kernel B's launch (it runs between the two tile kernels): the backward's blocks sorted by
    // the cost the forward recorded, most expensive first, per XCD — 8 extra workgroups, one counting sort each
    int order_tiles;               // T (a multiple of 32), 0: off
    const uint32_t *block_cost;    // T * 16 quantised costs (0..255)
    uint32_t *block_order;         // out: per XCD, T * 2 block indices (tile slot * 16 + block)
};

__device__ __forceinline__ float surf_depth_of(const ConsumerArgs &a, float al, float D, float med)
{
    const float Dh = (al > 0.0f) ? D / al : D;
    return Dh * (1.0f - a.depth_ratio) + med * a.depth_ratio;
}

__device__ __forceinline__ float3 surf_point(const ConsumerArgs &a, int r, int c, float &s_out)
{
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    const float al = a.allmap[SLS_CH_ALPHA * P + pix];
    const float D = a.allmap[SLS_CH_DEPTH * P + pix];
    const float med = a.allmap[SLS_CH_MEDIAN * P + pix];
    const float s = surf_depth_of(a, al, D, med);
    const float2 cc = a.col_h[c], rr = a.row_h[r];
    s_out = s;
    // (products rounded one by one, never contracted into the differences taken of them: the point is the same bits
    //  whether it reaches its user in registers or through LDS — the tile backward's inline stage)
    return make_float3(__fmul_rn(__fmul_rn(s, cc.x), rr.x), __fmul_rn(__fmul_rn(s, cc.y), rr.x), __fmul_rn(s, rr.y));
}

// dL/dallmap of pixel (r, c) for a loss weight of 1: out = [depth, alpha, n0, n1, n2, median, distortion].
// Needs kernel B's pieces of the pixel (nsd) and of its four neighbours: dL/du of the pixels above and below (tu, td),
// dL/dv of the pixels left and right (tl, tr); a neighbour outside the image is masked here, whatever was passed.
// (the pixel's own inputs are loaded apart, so that a caller can have them in flight while it produces the pieces)
struct ConsumerOwn { float al, D, N0, N1, N2, med, gt; float2 cc, rr; bool valid; };
__device__ __forceinline__ ConsumerOwn consumer_own_load(const ConsumerArgs &a, int r, int c)
{
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    ConsumerOwn o;
    o.valid = a.valid[pix] == 1;
    o.al = a.allmap[SLS_CH_ALPHA * P + pix];
    o.D = a.allmap[SLS_CH_DEPTH * P + pix];
    o.N0 = a.allmap[(SLS_CH_NORMAL + 0) * P + pix];
    o.N1 = a.allmap[(SLS_CH_NORMAL + 1) * P + pix];
    o.N2 = a.allmap[(SLS_CH_NORMAL + 2) * P + pix];
    o.med = a.allmap[SLS_CH_MEDIAN * P + pix];
    o.gt = a.gt_depth[pix];
    o.cc = a.col_h[c]; o.rr = a.row_h[r];
    return o;
}
__device__ __forceinline__ void consumer_pixel_grad_core(const ConsumerArgs &a, int r, int c, const ConsumerOwn &own,
                                                         const float4 tu, const float4 td, const float4 tl, const float4 tr,
                                                         const float4 nsd, float (&out)[7])
{
    const bool valid = own.valid;
    const float al = own.al, D = own.D, N0 = own.N0, N1 = own.N1, N2 = own.N2;
    const bool hit = al > 0.0f;
    const float inv = hit ? 1.0f / al : 1.0f;
    const float s = surf_depth_of(a, al, D, own.med);
    // gather the stencil adjoint: P(r,c) enters u(r-1,c) with +, u(r+1,c) with -, v(r,c-1) with +, v(r,c+1) with -
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    const bool up = r > 0, down = r < a.H - 1, left = c > 0, right = c < a.W - 1;
    g0 += up ? tu.x : 0.0f; g1 += up ? tu.y : 0.0f; g2 += up ? tu.z : 0.0f;
    g0 -= down ? td.x : 0.0f; g1 -= down ? td.y : 0.0f; g2 -= down ? td.z : 0.0f;
    g0 += left ? tl.x : 0.0f; g1 += left ? tl.y : 0.0f; g2 += left ? tl.z : 0.0f;
    g0 -= right ? tr.x : 0.0f; g1 -= right ? tr.y : 0.0f; g2 -= right ? tr.z : 0.0f;
    const float2 cc = own.cc, rr = own.rr;
    float ds = g0 * cc.x * rr.x + g1 * cc.y * rr.x + g2 * rr.y;
    float da = 0.0f, dn0 = 0.0f, dn1 = 0.0f, dn2 = 0.0f;
    if (valid) {
        const float diff = s - own.gt;
        ds += ((diff > 0.0f) ? 1.0f : ((diff < 0.0f) ? -1.0f : 0.0f)) * a.inv_P;
        const float k = -a.lambda_n * a.inv_nv;
        dn0 = k * al * nsd.x; dn1 = k * al * nsd.y; dn2 = k * al * nsd.z;
        da = k * nsd.w + a.lambda_a * a.inv_nv * (al - 1.0f) / fmaxf((1.0f - al) * al, 1e-12f);   // torch BCE backward
    }
    const float dDh = (1.0f - a.depth_ratio) * ds;
    if (hit) da -= (dDh * D + dn0 * N0 + dn1 * N1 + dn2 * N2) * inv * inv;
    out[0] = dDh * inv; out[1] = da; out[2] = dn0 * inv; out[3] = dn1 * inv; out[4] = dn2 * inv;
    out[5] = a.depth_ratio * ds; out[6] = 0.0f;
}

// The same from kernel B's planes (du, dv, ns) in memory.
__device__ __forceinline__ void consumer_pixel_grad(const ConsumerArgs &a, int r, int c, float (&out)[7])
{
    const size_t pix = (size_t)r * a.W + c;
    // (four unconditional loads at clamped addresses, masked afterwards: a load under a condition compiles into a
    //  branch with a full wait each — four serialised round trips in a kernel that is nothing but latency)
    const bool up = r > 0, down = r < a.H - 1, left = c > 0, right = c < a.W - 1;
    const float4 tu = a.du[up ? pix - a.W : pix], td = a.du[down ? pix + a.W : pix];
    const float4 tl = a.dv[left ? pix - 1 : pix], tr = a.dv[right ? pix + 1 : pix];
    consumer_pixel_grad_core(a, r, c, consumer_own_load(a, r, c), tu, td, tl, tr, a.ns[pix], out);
}

// Kernel B's work for ONE pixel from its own inputs (valid, alpha, the raw normal planes, surface depth s, target gt) and,
// for an interior pixel, the surface points below / above / right / left of it: the stencil's adjoint pieces dL/du, dL/dv,
// (n_surf, <n_hat, n_surf>) and the pixel's three loss terms (zero where the pixel is not valid).
__device__ __forceinline__ void consumer_b_core(const ConsumerArgs &a, bool valid, float al, float N0, float N1, float N2,
                                                float s, float gt, bool interior, float3 pu, float3 pd, float3 pr, float3 pl,
                                                float4 &du, float4 &dv, float4 &nsd, float &lg, float &ln, float &la)
{
    const bool hit = al > 0.0f;
    const float inv = hit ? 1.0f / al : 1.0f;
    const float n0 = N0 * inv, n1 = N1 * inv, n2 = N2 * inv;
    du = make_float4(0, 0, 0, 0); dv = du; nsd = du;
    lg = 0.0f; ln = 0.0f; la = 0.0f;
    if (interior) {
        const float u0 = pu.x - pd.x, u1 = pu.y - pd.y, u2 = pu.z - pd.z;
        const float v0 = pr.x - pl.x, v1 = pr.y - pl.y, v2 = pr.z - pl.z;
        const float c0 = u1 * v2 - u2 * v1, c1 = u2 * v0 - u0 * v2, c2 = u0 * v1 - u1 * v0;
        const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
        const float rden = 1.0f / fmaxf(len, 1e-12f);            // F.normalize eps
        const float s0 = c0 * rden, s1 = c1 * rden, s2 = c2 * rden;
        const float dot = n0 * s0 + n1 * s1 + n2 * s2;
        nsd = make_float4(s0, s1, s2, dot);
        if (valid) {
            // dL/dn_surf = -lambda_n/Nv * alpha * n_hat ; through normalize ; through the cross product
            const float k = -a.lambda_n * a.inv_nv * al;
            float g0 = k * n0, g1 = k * n1, g2 = k * n2;
            if (len > 1e-12f) {
                const float gd = g0 * s0 + g1 * s1 + g2 * s2;
                g0 = (g0 - gd * s0) * rden; g1 = (g1 - gd * s1) * rden; g2 = (g2 - gd * s2) * rden;
            } else {
                g0 *= rden; g1 *= rden; g2 *= rden;
            }
            // cr = u x v : dL/du = v x g, dL/dv = g x u
            du = make_float4(v1 * g2 - v2 * g1, v2 * g0 - v0 * g2, v0 * g1 - v1 * g0, 0.0f);
            dv = make_float4(g1 * u2 - g2 * u1, g2 * u0 - g0 * u2, g0 * u1 - g1 * u0, 0.0f);
        }
    }
    if (valid) {
        lg = fabsf(s - gt);
        ln = 1.0f - al * nsd.w;
        la = -fmaxf(logf(al), -100.0f);                            // torch BCE clamps log at -100
    }
}

// The same for pixel (r, c) of the image, everything from memory.
__device__ __forceinline__ void consumer_b_pixel(const ConsumerArgs &a, int r, int c, float4 &du, float4 &dv, float4 &nsd,
                                                 float &lg, float &ln, float &la)
{
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    const bool valid = a.valid[pix] == 1;
    const float al = a.allmap[SLS_CH_ALPHA * P + pix];
    const float N0 = a.allmap[(SLS_CH_NORMAL + 0) * P + pix];
    const float N1 = a.allmap[(SLS_CH_NORMAL + 1) * P + pix];
    const float N2 = a.allmap[(SLS_CH_NORMAL + 2) * P + pix];
    const float gt = a.gt_depth[pix];
    float s, t;
    (void)surf_point(a, r, c, s);
    const bool interior = (r > 0) && (r < a.H - 1) && (c > 0) && (c < a.W - 1);
    float3 pu = make_float3(0, 0, 0), pd = pu, pr = pu, pl = pu;
    if (interior) {
        pu = surf_point(a, r + 1, c, t); pd = surf_point(a, r - 1, c, t);
        pr = surf_point(a, r, c + 1, t); pl = surf_point(a, r, c - 1, t);
    }
    consumer_b_core(a, valid, al, N0, N1, N2, s, gt, interior, pu, pd, pr, pl, du, dv, nsd, lg, ln, la);
}

// The per-block partial sums of kernel B -> the three sums + the total, by ONE wave (lane = threadIdx % 64).
__device__ __forceinline__ void consumer_reduce_partials_wave(const ConsumerArgs &a, int nb, int lane)
{
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    for (int b = lane; b < nb; b += 64) { t0 += a.partials[3 * b]; t1 += a.partials[3 * b + 1]; t2 += a.partials[3 * b + 2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t0 += __shfl_down(t0, off, 64); t1 += __shfl_down(t1, off, 64); t2 += __shfl_down(t2, off, 64);
    }
    if (lane == 0) {
        a.sums[0] = t0; a.sums[1] = t1; a.sums[2] = t2;
        a.sums[3] = t0 * a.inv_P + a.lambda_n * a.inv_nv * t1 + a.lambda_a * a.inv_nv * t2;
    }
}

// n items in the order of their keys (0..255), largest first, by one workgroup of 256 threads: a counting sort.  A
// permutation of 0..n-1 whatever the keys are.  The keys of the first 2048 items are fetched ONCE, all loads in flight
// together, and kept in registers: the sort is a chain of latencies (it rides in launches that are short themselves),
// and a loop of load -> LDS atomic pays a memory round trip per iteration.
template <class KeyFn>
__device__ __forceinline__ void order_by_key_desc(int n, KeyFn key, uint32_t *__restrict__ out)
{
    constexpr int kHeld = 8;
    __shared__ uint32_t s_hist[256];
    const int tid = threadIdx.x;
    uint32_t kc[kHeld];
#pragma unroll
    for (int j = 0; j < kHeld; ++j) {
        const int i = tid + j * 256;
        kc[j] = i < n ? min(key(i), 255u) : 0u;
    }
    s_hist[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kHeld; ++j)
        if (tid + j * 256 < n) atomicAdd(&s_hist[255u - kc[j]], 1u);
    for (int i = tid + kHeld * 256; i < n; i += 256) atomicAdd(&s_hist[255u - min(key(i), 255u)], 1u);
    __syncthreads();
    // exclusive scan of the 256 bins (wave 0)
    if (tid < 64) {
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = s_hist[tid * 4 + k]; sum += v[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if (tid >= off) inc += t; }
        uint32_t base = inc - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_hist[tid * 4 + k] = base; base += v[k]; }
    }
    __syncthreads();
    // (items of equal key in index order inside a thread's share only: any order of equals is a valid order)
#pragma unroll
    for (int j = 0; j < kHeld; ++j)
        if (tid + j * 256 < n) out[atomicAdd(&s_hist[255u - kc[j]], 1u)] = (uint32_t)(tid + j * 256);
    for (int i = tid + kHeld * 256; i < n; i += 256) out[atomicAdd(&s_hist[255u - min(key(i), 255u)], 1u)] = (uint32_t)i;
}

// Blocks of one XCD (index i = tile slot * 16 + block, tile = ((slot >> 2) * 8 + xcd) * 4 + (slot & 3), the mapping
// of tile_of_block) in the order of their backward cost, most expensive first: the launch drains when the queue is
// empty, and it drains for as long as the last-started waves run — they should be the cheap ones.  The order is a
// permutation whatever the costs are; only speed depends on it.  One workgroup of 256 threads per XCD.
__device__ __forceinline__ void order_blocks_by_cost(int T, const uint32_t *__restrict__ block_cost,
                                                     uint32_t *__restrict__ block_order, int xcd)
{
    order_by_key_desc(T * 2, [&](int i) {                      // T * 16 / 8 blocks per XCD
        const int ts = i >> 4;
        const int tile = ((ts >> 2) * 8 + xcd) * 4 + (ts & 3);
        return block_cost[tile * 16 + (i & 15)];
    }, block_order + (size_t)xcd * (T * 2));
}

// A keyframe's own launch-order buffer (SlsMappingConfig.block_order): word 0 = this tag once an iteration has filled
// the T * 16 words behind it (a zero-initialised or differently sized buffer is not walked).
__host__ __device__ inline uint32_t block_order_tag(int T) { return 0x424F0000u + (uint32_t)T; }

}  // namespace sls
