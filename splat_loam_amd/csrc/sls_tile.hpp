// sls_tile.hpp — device helpers shared by the tile kernels (sls_render.hip,
// sls_render_block.hip): the (pixel, surfel) evaluation, the support-box cull
// and the wave-private LDS staging of a tile's list.  Maths in DESIGN.md §2.
#pragma once
#include "sls_common.hpp"

namespace sls {

// Pairs of values that are produced and consumed together live in 2-vectors (v2f, sls_common.hpp): their
// arithmetic becomes packed FP32 instructions (v_pk_add/mul/fma_f32) without any packing moves, because the pairs
// are adjacent from where they are made ((x, y) halves of a ds_read_b128, (d0, d1) of the pixel's ray, (hu, hv),
// (dx, dy)).  Measured on gfx950 (tools/micro/pk_rate.hip): a packed op occupies the issue port 1.7x as long as
// a scalar one, so a pair is worth 15 % of two scalar ops — and nothing once a v_mov is needed to form it, which
// is why the auto-vectoriser is off for the tile kernels (-fno-slp-vectorize: it paired values that were NOT
// adjacent and paid more moves than it saved: 167 -> 160 VALU instructions per backward step without it).
struct Eval {
    v2f dl01, huv, uv, dxy;       // (dl0, dl1), (hu, hv), (u, v), (dx, dy)
    float dl2, rinv, t, depth, G, og, alpha;
    bool use3d, skip;
};

// One (pixel, surfel) evaluation; identical in forward and backward.  d01 = (d0, d1) of the pixel's ray,
// pcr = (column, row) of the pixel.
__device__ __forceinline__ void eval_surfel(const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                            const float4 q4, v2f d01, float d2, v2f pcr,
                                            float wrapW, float invW, float near_c, Eval &e)
{
    e.dl01 = d01 - mk2(q3.x, q3.y); e.dl2 = d2 - q3.z;
    const float nd = q2.x * d01.x + q2.y * d01.y + q2.z * d2;
    const bool valid3d = nd < 0.0f;
    e.rinv = __builtin_amdgcn_rcpf(nd);
    e.huv = mk2(q0.x * e.dl01.x + q0.y * e.dl01.y + q0.z * e.dl2, q1.x * e.dl01.x + q1.y * e.dl01.y + q1.z * e.dl2);
    e.uv = e.huv * e.rinv;
    e.t = q0.w * e.rinv;
    const float rho3 = e.uv.x * e.uv.x + e.uv.y * e.uv.y;
    // D5 wrapped azimuth difference, branch-free: wrapW = W (360-degree image) or 0
    e.dxy = pcr - mk2(q4.x, q4.y);
    e.dxy.x = e.dxy.x - wrapW * __builtin_rintf(e.dxy.x * invW);
    const float rho2 = SLS_FILTER_INV_SQUARE * (e.dxy.x * e.dxy.x + e.dxy.y * e.dxy.y);
    e.use3d = valid3d && (rho3 <= rho2);
    const float rho = e.use3d ? rho3 : rho2;
    e.depth = e.use3d ? e.t : q1.w;
    e.G = __builtin_amdgcn_exp2f(rho * -0.72134752044448170f);   // exp(-rho / 2) = 2^(-rho log2(e) / 2)
    e.og = q2.w * e.G;
    e.alpha = fminf(SLS_ALPHA_MAX, e.og);
    e.skip = (e.depth < near_c) || (e.alpha < SLS_ALPHA_MIN);
}

// Conservative test: can the surfel (centre q4.xy, support half-extents q4.zw)
// reach a pixel of the box centred (bcx, bcy) with half-extents (bhx, bhy)?
__device__ __forceinline__ bool cull_pass(const float4 q4, float bcx, float bcy, float bhx, float bhy,
                                          float wrapW, float invW)
{
    const float dx0 = bcx - q4.x;
    const float dxc = dx0 - wrapW * __builtin_rintf(dx0 * invW);
    return (fabsf(dxc) <= q4.z + bhx) && (fabsf(bcy - q4.y) <= q4.w + bhy);
}

// ---------------------------------------------------------------------------
// Footprint test of a surfel against a whole pixel block (forward cull).
//
// A pixel ray d can only receive alpha >= 1/255 from the surfel's 3D branch if
//     G(d) = |(Hu.d, Hv.d)| + kc (n.d) <= 0          (rho3 <= kc^2 and n.d < 0; Hu.dc = Hv.dc = 0)
// and G is CONVEX in d (a norm of linear forms plus a linear form).  So for any d0
//     G(d) >= G(d0) + grad G(d0) . (d - d0),
// and with the block's rays written as d = d0 + x Dx + y Dy + r, |x|,|y| <= 1, |r| <= eps (Taylor
// remainder of the unit sphere's parametrisation), the right-hand side is bounded below over the
// whole block by  G(d0) - |gradG.Dx| - |gradG.Dy| - eps |gradG|_1.  If that is positive no pixel of
// the block lies in the 3D footprint: a separating-line test whose axis is the footprint's own
// boundary normal at the block centre — nearly exact for footprints larger than the block, which
// are the near surfels that dominate the consumed part of every list (a CPU simulation of the lanes' use in round 3, HISTORY.md:
// 31 % fewer evaluated (block, surfel) pairs than the support box alone, 2.5 % above the exact
// count, nothing missed).  The 2D (low-pass) branch reaches sqrt(kc^2/2) pixels from the centre:
// tested as the distance from the centre to the block's pixel box.
// Everything is multiplied through by |(a,b)| so that no division is needed.
__device__ __forceinline__ float uniform_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
struct BlockCone {
    float d0[3], Dx[3], Dy[3], eps;
};
__device__ __forceinline__ BlockCone make_block_cone(const DevCam &cam, float pcx, float pcy, float hx, float hy)
{
    BlockCone c;
    float sa, ca, se, ce;
    sincosf((pcx - cam.cx) / cam.fx, &sa, &ca);
    sincosf((pcy - cam.cy) / cam.fy, &se, &ce);
    c.d0[0] = ca * ce; c.d0[1] = sa * ce; c.d0[2] = se;
    const float kx = hx / cam.fx, ky = hy / cam.fy;
    c.Dx[0] = -kx * sa * ce; c.Dx[1] = kx * ca * ce; c.Dx[2] = 0.0f;
    c.Dy[0] = -ky * ca * se; c.Dy[1] = -ky * sa * se; c.Dy[2] = ky * ce;
    const float span = fabsf(kx) + fabsf(ky);
    c.eps = 0.5f * span * span * 1.01f + 4.0e-6f;     // second-order remainder + float32 slop of d0 / Dx / Dy
    // wave-uniform by construction (one block per wave), but sincosf leaves them in vector registers:
    // move them to scalar registers (10 VGPRs less in the tile kernel)
#pragma unroll
    for (int k = 0; k < 3; ++k) { c.d0[k] = uniform_f(c.d0[k]); c.Dx[k] = uniform_f(c.Dx[k]); c.Dy[k] = uniform_f(c.Dy[k]); }
    c.eps = uniform_f(c.eps);
    return c;
}
// true: the 3D footprint cannot reach any pixel of the block.
__device__ __forceinline__ bool cone_outside(const BlockCone &c, const float4 q0, const float4 q1, const float4 q2,
                                             const float4 q3)
{
    const float l0 = c.d0[0] - q3.x, l1 = c.d0[1] - q3.y, l2 = c.d0[2] - q3.z;   // d0 - dc: exact cancellation
    const float a = q0.x * l0 + q0.y * l1 + q0.z * l2;
    const float b = q1.x * l0 + q1.y * l1 + q1.z * l2;
    const float e = q2.x * c.d0[0] + q2.y * c.d0[1] + q2.z * c.d0[2];
    const float n2 = a * a + b * b, nrm = __builtin_amdgcn_sqrtf(n2), kn = q3.w * nrm;
    const float g0 = a * q0.x + b * q1.x + kn * q2.x;      // |(a,b)| * grad G
    const float g1 = a * q0.y + b * q1.y + kn * q2.y;
    const float g2 = a * q0.z + b * q1.z + kn * q2.z;
    const float tx = g0 * c.Dx[0] + g1 * c.Dx[1];
    const float ty = g0 * c.Dy[0] + g1 * c.Dy[1] + g2 * c.Dy[2];
    const float reach = fabsf(tx) + fabsf(ty) + c.eps * (fabsf(g0) + fabsf(g1) + fabsf(g2));
    // 2e-4 relative slack on both terms of G: rounding of a, b, e, the approximate sqrt, and the
    // tile kernels' own approximate rcp / exp when they decide alpha >= 1/255 at the boundary
    return n2 + kn * e - reach > 2.0e-4f * (n2 + kn * fabsf(e));
}
// true: the low-pass disc (radius kc / sqrt 2 pixels around the centre) reaches the pixel box.
__device__ __forceinline__ bool disc_reaches(const float4 q3, const float4 q4, float bcx, float bcy, float bhx,
                                             float bhy, float wrapW, float invW)
{
    const float dx0 = bcx - q4.x;
    const float dxc = dx0 - wrapW * __builtin_rintf(dx0 * invW);
    const float ex = fmaxf(fabsf(dxc) - bhx, 0.0f), ey = fmaxf(fabsf(bcy - q4.y) - bhy, 0.0f);
    const float r = q3.w * 0.70781f + 0.05f;               // kc / sqrt 2, +0.1 % and 0.05 px as in (ex, ey)
    return ex * ex + ey * ey <= r * r;
}

// Box of the active lanes of an 8x8 sub-tile at (x0, y0); false if none.
__device__ __forceinline__ bool active_box(uint64_t m, int x0, int y0, float &bcx, float &bcy, float &bhx,
                                           float &bhy)
{
    int xa, xb, ya, yb;
    if (!mask_bbox8x8(m, xa, xb, ya, yb)) return false;
    bcx = (float)x0 + 0.5f * (float)(xa + xb);
    bhx = 0.5f * (float)(xb - xa);
    bcy = (float)y0 + 0.5f * (float)(ya + yb);
    bhy = 0.5f * (float)(yb - ya);
    return true;
}

// Workgroup -> (tile, sub-tile).  The dispatcher places block b on XCD b % 8.  All sub-tiles
// of a tile stay on one XCD (they share the tile's list and records through its L2), and
// each XCD gets runs of 4 neighbouring tiles taken round-robin from the WHOLE image, so that
// the expensive image rows (long lists) are spread over all XCDs instead of filling one.
template <int PER_TILE>
__device__ __forceinline__ void tile_of_block(int b, int T, int &tile, int &sub)
{
    if (T % 32 == 0) {
        const int xcd = b % 8, i = b / 8;
        const int ts = i / PER_TILE;                 // tile slot inside this XCD
        tile = ((ts >> 2) * 8 + xcd) * 4 + (ts & 3);
        sub = i % PER_TILE;
    } else {
        tile = b / PER_TILE;
        sub = b % PER_TILE;
    }
}
// Wave-private two-deep staging of 64 list entries per round (macros so that the
// small arrays stay in registers): lane k of load i fetches float4 #(k mod 5) of
// entry #(k div 5), the LDS image is written with stride-1 ds_write_b128.
// `first` is the list offset of the tile, `limit` the number of usable entries.
static_assert(kRec4 == 5, "staging macros are written out for 5 float4 per record");
#define SLS_STAGE_DECL float4 sp0, sp1, sp2, sp3, sp4; uint32_t si0, si1, si2, si3, si4;
#define SLS_WIDX1(i_, first, r_, limit) vals[(first) + (uint32_t)min((r_) * 64 + ((i_) * 64 + lane) / kRec4, (limit) - 1)]
#define SLS_WSTAGE_LOAD_IDX(first, r_, limit)                                                  \
    si0 = SLS_WIDX1(0, first, r_, limit); si1 = SLS_WIDX1(1, first, r_, limit);                 \
    si2 = SLS_WIDX1(2, first, r_, limit); si3 = SLS_WIDX1(3, first, r_, limit);                 \
    si4 = SLS_WIDX1(4, first, r_, limit);
#define SLS_WREC1(i_, idx_) rec[(size_t)(idx_) * kRec4 + (((i_) * 64 + lane) % kRec4)]
#define SLS_WSTAGE_LOAD_REC()                                                                  \
    sp0 = SLS_WREC1(0, si0); sp1 = SLS_WREC1(1, si1); sp2 = SLS_WREC1(2, si2);                  \
    sp3 = SLS_WREC1(3, si3); sp4 = SLS_WREC1(4, si4);
#define SLS_WSTAGE_STORE()                                                                     \
    s_rec[0 * 64 + lane] = sp0; s_rec[1 * 64 + lane] = sp1; s_rec[2 * 64 + lane] = sp2;         \
    s_rec[3 * 64 + lane] = sp3; s_rec[4 * 64 + lane] = sp4;

}  // namespace sls
