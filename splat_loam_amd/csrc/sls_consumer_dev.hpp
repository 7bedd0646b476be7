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

__device__ __forceinline__ float3 surf_point(const ConsumerArgs &a, int r, int c, float &s_out)
{
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    const float al = a.allmap[SLS_CH_ALPHA * P + pix];
    const float D = a.allmap[SLS_CH_DEPTH * P + pix];
    const float med = a.allmap[SLS_CH_MEDIAN * P + pix];
    const float Dh = (al > 0.0f) ? D / al : D;
    const float s = Dh * (1.0f - a.depth_ratio) + med * a.depth_ratio;
    const float2 cc = a.col_h[c], rr = a.row_h[r];
    s_out = s;
    return make_float3(s * cc.x * rr.x, s * cc.y * rr.x, s * rr.y);
}

// dL/dallmap of pixel (r, c) for a loss weight of 1: out = [depth, alpha, n0, n1, n2, median, distortion].
// Needs kernel B's planes (du, dv, ns) of the pixel and its four neighbours.
__device__ __forceinline__ void consumer_pixel_grad(const ConsumerArgs &a, int r, int c, float (&out)[7])
{
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    const bool valid = a.valid[pix] == 1;
    const float al = a.allmap[SLS_CH_ALPHA * P + pix];
    const float D = a.allmap[SLS_CH_DEPTH * P + pix];
    const float N0 = a.allmap[(SLS_CH_NORMAL + 0) * P + pix];
    const float N1 = a.allmap[(SLS_CH_NORMAL + 1) * P + pix];
    const float N2 = a.allmap[(SLS_CH_NORMAL + 2) * P + pix];
    const bool hit = al > 0.0f;
    const float inv = hit ? 1.0f / al : 1.0f;
    float s;
    (void)surf_point(a, r, c, s);
    // gather the stencil adjoint: P(r,c) enters u(r-1,c) with +, u(r+1,c) with -, v(r,c-1) with +, v(r,c+1) with -
    // (four unconditional loads at clamped addresses, masked afterwards: a load under a condition compiles into a
    //  branch with a full wait each — four serialised round trips in a kernel that is nothing but latency)
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    const bool up = r > 0, down = r < a.H - 1, left = c > 0, right = c < a.W - 1;
    const float4 tu = a.du[up ? pix - a.W : pix], td = a.du[down ? pix + a.W : pix];
    const float4 tl = a.dv[left ? pix - 1 : pix], tr = a.dv[right ? pix + 1 : pix];
    g0 += up ? tu.x : 0.0f; g1 += up ? tu.y : 0.0f; g2 += up ? tu.z : 0.0f;
    g0 -= down ? td.x : 0.0f; g1 -= down ? td.y : 0.0f; g2 -= down ? td.z : 0.0f;
    g0 += left ? tl.x : 0.0f; g1 += left ? tl.y : 0.0f; g2 += left ? tl.z : 0.0f;
    g0 -= right ? tr.x : 0.0f; g1 -= right ? tr.y : 0.0f; g2 -= right ? tr.z : 0.0f;
    const float2 cc = a.col_h[c], rr = a.row_h[r];
    float ds = g0 * cc.x * rr.x + g1 * cc.y * rr.x + g2 * rr.y;
    const float4 nsd = a.ns[pix];
    float da = 0.0f, dn0 = 0.0f, dn1 = 0.0f, dn2 = 0.0f;
    if (valid) {
        const float diff = s - a.gt_depth[pix];
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

}  // namespace sls
