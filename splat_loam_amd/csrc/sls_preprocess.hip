// sls_preprocess.hip — per-surfel kernels (N threads): 3D -> spherical
// projection and record build (A1), the tiles-touched scan (A2), the
// gradient chain back to means/scales/rotations/opacities (A8) and
// mark_visible (A9).  SURVEY.md §8a; maths in DESIGN.md §2.
//
// HBM-bound, trivially parallel.  Built with -ffp-contract=off and only
// exactly-rounded operations (include/sls_det_math.h) so that every INTEGER
// this stage emits — tile rectangle, tiles_touched, radii, depth-key bits —
// is reproducible bit for bit by a CPU checker.
#include <cstring>
#include "sls_common.hpp"
#include "sls_consumer_dev.hpp"
#include "sls_resort.hpp"

namespace sls {

__device__ __forceinline__ float dot3(const float *a, const float *b)
{
    return fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2]));
}
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
__device__ __forceinline__ void matvec(const float *M, const float *v, float *o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = dot3(M + 3 * i, v);
}
__device__ __forceinline__ void matTvec(const float *M, const float *v, float *o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = fmaf(M[i], v[0], fmaf(M[3 + i], v[1], M[6 + i] * v[2]));
}
__device__ __forceinline__ int floordiv_i(int a, int b)
{
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}
__device__ __forceinline__ int posmod_i(int a, int b)
{
    int m = a % b;
    return m < 0 ? m + b : m;
}
__device__ __forceinline__ int to_int_clamped(float v)
{
    v = fminf(fmaxf(v, -1.0e9f), 1.0e9f);
    return (int)v;
}
// quaternion (w,x,y,z) -> columns of R(q) (utils/general_utils.py:13-37, no
// re-normalisation: callers pass F.normalize'd rotations).
__device__ __forceinline__ void quat_axes(const float4 q, float *tu, float *tv, float *tn)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    tu[0] = 1.0f - 2.0f * fmaf(y, y, z * z);
    tu[1] = 2.0f * fmaf(x, y, r * z);
    tu[2] = 2.0f * fmaf(x, z, -(r * y));
    tv[0] = 2.0f * fmaf(x, y, -(r * z));
    tv[1] = 1.0f - 2.0f * fmaf(x, x, z * z);
    tv[2] = 2.0f * fmaf(y, z, r * x);
    tn[0] = 2.0f * fmaf(x, z, r * y);
    tn[1] = 2.0f * fmaf(y, z, -(r * x));
    tn[2] = 1.0f - 2.0f * fmaf(x, x, y * y);
}
// Angular half-extents (elevation theta, azimuth daz) of a ball of radius rad
// centred at range rho / horizontal range rxy, seen from the origin (D4).
// EXACT: the rectangle of tiles comes out of these (integers: the checker's arithmetic, sls_det_math.h).  The
// approximate variant (hardware rcp / sqrt, same polynomial) serves the support extents of the cull, which only
// have to be conservative and carry their own margins.
__device__ __forceinline__ float asin01_approx(float s)
{
    // atan2(s, sqrt(1 - s^2)) for s in [0, 1): both arguments non-negative
    const float c = __builtin_amdgcn_sqrtf(fmaf(-s, s, 1.0f));
    const float mx = fmaxf(s, c), mn = fminf(s, c);
    const float r = sls_atan_unit(mn * __builtin_amdgcn_rcpf(mx));
    return s > c ? SLS_PIO2 - r : r;
}
template <bool EXACT>
__device__ __forceinline__ void ball_extent(float rad, float rho, float rxy, float &theta, float &daz)
{
    if (!(rad < rho)) { theta = SLS_PI; daz = SLS_PI; return; }
    if (EXACT) {
        theta = sls_asin01(rad / rho);
        const float q = rad / rxy;
        if (!(q < 1.0f)) daz = SLS_PI;
        else daz = sls_asin01(q);
    } else {
        theta = asin01_approx(rad * __builtin_amdgcn_rcpf(rho));
        const float q = rad * __builtin_amdgcn_rcpf(rxy);
        if (!(q < 0.9999f)) daz = SLS_PI;
        else daz = asin01_approx(q);
    }
}

__device__ __forceinline__ uint32_t s_rect_cnt(const int4 rc) { return (uint32_t)(rc.y * rc.w); }

struct SurfelGeom {
    float p[3], rho, rho2, rxy, rxy2;
    float Tu[3], Tv[3], Tn[3], n[3], A[3], B[3], Hu[3], Hv[3];
    float su, sv, sig, c;
};

// Correctly rounded division and square root (-fhip-fp32-correctly-rounded-divide-sqrt) cost ~14 VALU instructions
// each.  They are kept where an INTEGER output depends on them — the centre's projection and the support extents
// that become the tile rectangle, the depth key — because those must match the checker bit for bit.  Everything that
// only feeds float outputs (the record's tangent rows, the whole backward, the optimiser) uses the hardware's
// 1-ulp v_rcp_f32 / v_sqrt_f32 / v_exp_f32: preprocess_bwd 1279 -> ~500 VALU instructions per surfel.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Shared by forward and backward.  EXACT_ROOTS: rho and rxy correctly rounded (forward: they enter the projection).
template <bool EXACT_ROOTS>
__device__ __forceinline__ void surfel_geom(const DevCam &cam, const float *m, const float2 s, const float4 q,
                                            SurfelGeom &g)
{
#pragma unroll
    for (int k = 0; k < 3; ++k)
        g.p[k] = fmaf(cam.R[3 * k], m[0], fmaf(cam.R[3 * k + 1], m[1], fmaf(cam.R[3 * k + 2], m[2], cam.t[k])));
    g.rxy2 = fmaf(g.p[0], g.p[0], g.p[1] * g.p[1]);
    g.rho2 = fmaf(g.p[2], g.p[2], g.rxy2);
    g.rho = EXACT_ROOTS ? sqrtf(g.rho2) : fsqrt(g.rho2);
    g.rxy = EXACT_ROOTS ? sqrtf(g.rxy2) : fsqrt(g.rxy2);
    float tu[3], tv[3], tn[3];
    quat_axes(q, tu, tv, tn);
    matvec(cam.R, tu, g.Tu);
    matvec(cam.R, tv, g.Tv);
    matvec(cam.R, tn, g.Tn);
    g.su = s.x * cam.mod;
    g.sv = s.y * cam.mod;
    g.c = dot3(g.Tn, g.p);
    g.sig = (g.c > 0.0f) ? -1.0f : 1.0f;
    const float isu = g.sig * frcp(g.su), isv = -g.sig * frcp(g.sv);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.n[k] = g.sig * g.Tn[k];
        g.A[k] = g.Tv[k] * isu;
        g.B[k] = g.Tu[k] * isv;
    }
    cross3(g.A, g.p, g.Hu);
    cross3(g.B, g.p, g.Hv);
}

// Parameter activations of the reference model (scene/gaussian_model.py:39-44):
// scaling exp, opacity sigmoid, rotation F.normalize (eps 1e-12).  Used by the
// raw-parameter entry points (sls_mapping_step) so that no torch op sits
// between the optimiser state and the rasterizer.
struct RegArgs {
    int raw;            // 1: scales/rots/opac are raw (pre-activation) parameters
    float smax, pen;    // slam/mapper.py:190-195 scale regulariser (pen == 0: off)
    float *reg_out;     // device scalar accumulating pen * sum relu(max_axis_scale - smax)
    uint32_t *status_clear;   // optional: 8 status words of the iteration, zeroed by thread 0 of the FIRST kernel
                              // (nothing else touches them before this kernel has finished)
    uint32_t *zero_words;     // optional: a small table the iteration's first kernel zeroes (the direct binning's coarse
    int n_zero_words;         // counts, summed with atomics two kernels later): surfel thread i clears words i, i + N, ...
};

// The scales go through the library expf in BOTH directions: forward they enter the extents of the tile rectangle
// (integers), and the regulariser's test `max(s) >= scaling_max` (slam/mapper.py:190-195) must come out the same in the
// forward's loss value, in the backward's gradient and in torch's own exp — Mapper.densify clamps new scales AT
// scaling_max (slam/mapper.py:113-117), so whole generations of surfels sit on that edge (golden G7: 2408 of 2453;
// exp(log(0.1f)) is one ulp below 0.1f in the library, on it with the hardware exp2 — with the fast form here the backward
// priced them all while the forward and the reference priced none).  `EXACT` is kept for the call sites' documentation.
template <bool EXACT>
__device__ __forceinline__ void activate(const RegArgs &ra, float2 &s, float4 &q, float &o)
{
    if (!ra.raw) return;
    const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    s.x = expf(s.x); s.y = expf(s.y);
    // (opacity and rotation reach float outputs only)
    o = frcp(1.0f + __expf(-o));
    const float inv = frcp(fmaxf(fsqrt(n2), 1e-12f));
    q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
}

// ---------------------------------------------------------------------------
// A1 forward preprocess.  One thread per surfel; blocks of WAVES waves, the block's first surfel is `first`.
// s_mem: WAVES wave-private slices of kPreSliceBytes (tile tests, then the staging of the records).
// ---------------------------------------------------------------------------
struct PreFwdArgs {
    const float *means; const float2 *scales; const float4 *rots; const float *opac;
    float4 *rec; int *radii; int4 *rect; uint32_t *tiles; float *depth; uint32_t *order_keys, *order_vals, *n_dev;
    const float2 *col_cs, *row_cs; uint64_t *tile_mask; int4 *erec;
    uint32_t *sbox;   // optional: the surfels' block boxes (make_block_box) for the tile sort's block masks
    int erec_box;     // 1: erec is an array of uint2 {rectangle in one word (sls_sort.hip: pack_rect32), block box} — what the
                      // direct binning gathers (no D10 mask)
};
constexpr int kPreCullBytes = 64 * (int)(sizeof(SlsTileCullSurfel) + sizeof(int4) + 4 * sizeof(uint32_t)) + 16;
constexpr int kPreSliceBytes = kPreCullBytes > 64 * kRec4 * 16 ? kPreCullBytes : 64 * kRec4 * 16;

template <int WAVES>
__device__ __forceinline__ void preprocess_fwd_body(const DevCam &cam, const RegArgs &ra, int N, const PreFwdArgs &pa,
                                                    int first, unsigned char *s_mem, float *s_part)
{
    const float *__restrict__ means = pa.means; const float2 *__restrict__ scales = pa.scales;
    const float4 *__restrict__ rots = pa.rots; const float *__restrict__ opac = pa.opac;
    float4 *__restrict__ rec = pa.rec; int *__restrict__ radii = pa.radii; int4 *__restrict__ rect = pa.rect;
    uint32_t *__restrict__ tiles = pa.tiles; float *__restrict__ depth = pa.depth;
    uint32_t *__restrict__ order_keys = pa.order_keys, *__restrict__ order_vals = pa.order_vals, *__restrict__ n_dev = pa.n_dev;
    const float2 *__restrict__ col_cs = pa.col_cs, *__restrict__ row_cs = pa.row_cs;
    uint64_t *__restrict__ tile_mask = pa.tile_mask; int4 *__restrict__ erec = pa.erec;
    const int i = first + (int)threadIdx.x;
    uint32_t my_tiles = 0;
    // D10 (sls_det_math.h): which tiles of the rectangle the footprint can reach — bit k = k-th tile in emission
    // order (row-major).  Tested for rectangles of cam.tile_cull .. 64 tiles; others keep the whole rectangle.
    uint64_t my_mask = 0;
    bool tested = false;
    SlsTileCullSurfel cull;
    int4 my_rc = make_int4(0, 0, 0, 0);
    float my_reg = 0.0f;
    if (i == 0 && n_dev) *n_dev = (uint32_t)N;
    if (ra.zero_words && i < N) {
        for (int k = i; k < ra.n_zero_words; k += N) ra.zero_words[k] = 0u;
    }
    if (i == 0 && ra.status_clear) {
#pragma unroll
        for (int k = 0; k < 8; ++k) ra.status_clear[k] = 0u;
    }
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0, q4 = q0;
    if (i < N) {
        int r_out = 0;
        int4 rc = make_int4(0, 0, 0, 0);
        float dep = 0.0f;

        const float m[3] = { means[3 * i], means[3 * i + 1], means[3 * i + 2] };
        float2 s = scales[i];
        float4 q = rots[i];
        float o = opac[i];
        activate<true>(ra, s, q, o);
        if (ra.pen != 0.0f) my_reg = ra.pen * fmaxf(fmaxf(s.x, s.y) - ra.smax, 0.0f);
        SurfelGeom g;
        surfel_geom<true>(cam, m, s, q, g);
        bool vis = (g.rho >= cam.near_c) && (g.rho < 1.0e18f);  // D2 near cut; NaN/inf fail
        if (vis) {
            const float az = sls_atan2(g.p[1], g.p[0]);
            const float el = sls_atan2(g.p[2], g.rxy);
            const float cpx = fmaf(cam.fx, az, cam.cx), cpy = fmaf(cam.fy, el, cam.cy);
            const float smax = fmaxf(g.su, g.sv);
            float theta, daz;
            ball_extent<true>(SLS_CUTOFF * smax, g.rho, g.rxy, theta, daz);
            const float rx = fmaxf(fabsf(cam.fx) * daz, SLS_RMIN_PX);
            const float ry = fmaxf(fabsf(cam.fy) * theta, SLS_RMIN_PX);
            int xlo = to_int_clamped(floorf(cpx - rx + 0.5f));
            int xhi = to_int_clamped(floorf(cpx + rx + 0.5f));
            int ylo = to_int_clamped(floorf(cpy - ry + 0.5f));
            int yhi = to_int_clamped(floorf(cpy + ry + 0.5f));
            ylo = max(ylo, 0);
            yhi = min(yhi, cam.H - 1);
            int txlo = 0, ncols = 0;
            if (ylo > yhi) vis = false;
            if (vis) {
                if (cam.wrap) {
                    if ((long long)xhi - (long long)xlo + 1 >= (long long)cam.W) { txlo = 0; ncols = cam.GX; }
                    else {
                        const int a = floordiv_i(xlo, kTileW), b = floordiv_i(xhi, kTileW);
                        ncols = min(b - a + 1, cam.GX);
                        txlo = posmod_i(a, cam.GX);
                    }
                } else {
                    xlo = max(xlo, 0);
                    xhi = min(xhi, cam.W - 1);
                    if (xlo > xhi) vis = false;
                    else { txlo = xlo / kTileW; ncols = xhi / kTileW - txlo + 1; }
                }
            }
            if (vis) {
                const int tylo = ylo / kTileH, nrows = yhi / kTileH - tylo + 1;
                rc = make_int4(txlo, ncols, tylo, nrows);
                my_tiles = (uint32_t)(ncols * nrows);
                my_rc = rc;
                // (correctly rounded: the tile kernels subtract this unit vector from the pixel's — a 1-ulp difference
                //  here is 1e-4 of (d - dc) for a pixel next to the centre, and the checker rounds it this way)
                const float dcv[3] = { g.p[0] / g.rho, g.p[1] / g.rho, g.p[2] / g.rho };
                my_mask = my_tiles >= 64u ? ~0ull : ((1ull << my_tiles) - 1ull);
                tested = cam.tile_cull > 0 && tile_mask && my_tiles >= (uint32_t)cam.tile_cull && my_tiles <= 64u;
                if (tested) sls_tile_cull_surfel(g.Tu, g.Tv, g.n, g.p, dcv, g.su, g.sv, o, cpx, cpy, &cull);
                r_out = to_int_clamped(ceilf(fmaxf(rx, ry)));
                dep = g.rho;
                // Conservative support half-extents for the wave-level cull in the
                // tile kernels: a pixel can only receive alpha >= 1/255 if
                // rho <= rho_max = 2 ln(255 o); in the 3D branch the hit lies inside
                // the ball of radius sqrt(rho_max)*smax around p, in the 2D branch
                // within sqrt(rho_max/2) px of the centre.  Never changes a result.
                float ex = -1.0e30f, ey = -1.0e30f, kc = 0.0f;
                const float lo = 255.0f * o;
                if (lo > 1.0f) {
                    // (hardware-approximate rcp / sqrt / log here: these extents only have to be
                    //  conservative, the margins below swallow an ulp; nothing exact depends on them)
                    const float rho_max = 2.0f * __logf(lo) * 1.001f + 1.0e-3f;
                    float th2, daz2;
                    const float kk = __builtin_amdgcn_sqrtf(rho_max), rad = kk * smax;
                    kc = kk * 1.0001f;
                    ball_extent<false>(rad, g.rho, g.rxy, th2, daz2);
                    const float r2 = __builtin_amdgcn_sqrtf(0.5f * rho_max);
                    // Tighter bound for the 3D branch: the hit point is p + u su Tu + v sv Tv with
                    // u^2+v^2 <= rho_max, an ellipse; its (az, el) extent to first order, plus a bound
                    // on the second-order remainder of atan2 (|Hessian| <= 1/r^2 on the segment).
                    // min(ball bound, ellipse bound) is still conservative.
                    if (g.rxy > 2.0f * rad && g.rxy > 0.5f * g.rho) {
                        const float irxy2 = __builtin_amdgcn_rcpf(g.rxy2), irho2 = __builtin_amdgcn_rcpf(g.rho2);
                        const float au = (g.p[0] * g.Tu[1] - g.p[1] * g.Tu[0]) * irxy2;
                        const float av = (g.p[0] * g.Tv[1] - g.p[1] * g.Tv[0]) * irxy2;
                        const float rat_xy = rad * __builtin_amdgcn_rcpf(g.rxy - rad);
                        const float az_ell = kk * __builtin_amdgcn_sqrtf(g.su * g.su * au * au + g.sv * g.sv * av * av) + 0.75f * rat_xy * rat_xy;
                        const float zr = g.p[2] * __builtin_amdgcn_rcpf(g.rxy) * irho2;
                        const float eu = -zr * (g.p[0] * g.Tu[0] + g.p[1] * g.Tu[1]) + g.rxy * irho2 * g.Tu[2];
                        const float ev = -zr * (g.p[0] * g.Tv[0] + g.p[1] * g.Tv[1]) + g.rxy * irho2 * g.Tv[2];
                        const float rat = rad * __builtin_amdgcn_rcpf(g.rho - rad);
                        const float el_ell = kk * __builtin_amdgcn_sqrtf(g.su * g.su * eu * eu + g.sv * g.sv * ev * ev) + 1.5f * rat * rat;
                        daz2 = fminf(daz2, az_ell * 1.02f);
                        th2 = fminf(th2, el_ell * 1.02f);
                    }
                    ex = fmaxf(fabsf(cam.fx) * daz2, r2) * 1.001f + 0.05f;
                    ey = fmaxf(fabsf(cam.fy) * th2, r2) * 1.001f + 0.05f;
                }
                q0 = make_float4(g.Hu[0], g.Hu[1], g.Hu[2], g.sig * g.c);
                q1 = make_float4(g.Hv[0], g.Hv[1], g.Hv[2], g.rho);
                q2 = make_float4(g.n[0], g.n[1], g.n[2], o);
                q3 = make_float4(dcv[0], dcv[1], dcv[2], kc);
                q4 = make_float4(cpx, cpy, ex, ey);
            }
        }
        radii[i] = r_out;
        // (optional outputs: what the iteration's binning does not read is not written — sls_pipeline.hip)
        if (rect) rect[i] = rc;
        if (depth) depth[i] = dep;
        if (order_keys) {
            // Input of the depth-order sort.  Culled surfels are keyed by their range as well (they
            // emit nothing): a surfel that flips between visible and culled then keeps its place in
            // the order, which keeps the order repairable from one iteration to the next.
            order_keys[i] = depth_order_key(g.rho);
            if (order_vals) order_vals[i] = (uint32_t)i;
        }
    }
    // ---- D10: the tile tests of a wave's surfels, dealt out evenly over its lanes (a surfel's rectangle has 1 to 18
    // tiles on the bench scene: lane-per-surfel loops would run at the pace of the largest rectangle of the wave)
    // One wave-private LDS slice serves the tile tests and, afterwards, the staging of the records.
    if (__ballot(tested)) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int4 *s_rect_w = reinterpret_cast<int4 *>(s_mem + (size_t)wave * kPreSliceBytes);
        SlsTileCullSurfel *s_cull_w = reinterpret_cast<SlsTileCullSurfel *>(s_rect_w + 64);
        uint32_t *s_scan_w = reinterpret_cast<uint32_t *>(s_cull_w + 64);
        uint32_t *s_drop_w = s_scan_w + 64;                       // [64][2]
        uint32_t *s_seg_w = s_drop_w + 128;                       // [64]: lane of the r-th tested surfel
        uint32_t *s_ends_w = s_seg_w + 64;                        // [2]
        if (tested) s_cull_w[lane] = cull;
        s_rect_w[lane] = tested ? my_rc : make_int4(0, 1, 0, 1);     // (every lane: lanes beyond the last pair read owner 0)
        uint32_t incl = tested ? my_tiles : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        s_scan_w[lane] = incl;
        s_drop_w[2 * lane] = 0u; s_drop_w[2 * lane + 1] = 0u;
        {   // the tested lanes in lane order
            const uint64_t tb = __ballot(tested);
            if (tested) s_seg_w[__popcll(tb & ((1ull << lane) - 1ull))] = (uint32_t)lane;
            if (lane == 0) { s_ends_w[0] = 0u; s_ends_w[1] = 0u; }
        }
        int seg_base = 0;
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t j = base + (uint32_t)lane;
            const bool valid = j < total;
            // owner = first lane whose inclusive count exceeds j = number of segment ends <= j.  The ends of the
            // tested lanes are distinct (each has at least one pair): the ends that fall into this chunk as a 64-bit
            // mask (one LDS atomic per tested lane), the ends before it as a running count
            if (tested && incl > base && incl <= base + 64u) atomicOr(&s_ends_w[(incl - 1u - base) >> 5], 1u << ((incl - 1u - base) & 31u));
            __builtin_amdgcn_wave_barrier();
            const uint64_t ends = ((uint64_t)s_ends_w[1] << 32) | (uint64_t)s_ends_w[0];    // bit p: a segment's LAST pair is base + p
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { s_ends_w[0] = 0u; s_ends_w[1] = 0u; }
            const int seg = seg_base + (int)__popcll(ends & ((1ull << lane) - 1ull));           // segments that end before pair j
            seg_base += (int)__popcll(ends);
            const int owner = valid ? (int)s_seg_w[min(seg, 63)] : 0;
            const uint32_t idx = valid ? j - (s_scan_w[owner] - s_rect_cnt(s_rect_w[owner])) : 0u;
            const int4 orc = s_rect_w[owner];
            const int ncols = max(orc.y, 1);
            const int ky = (int)idx / ncols, kx = (int)idx - ky * ncols;
            int tx = orc.x + kx;
            if (tx >= cam.GX) tx -= cam.GX;
            const int ty = orc.z + ky;
            const int x0 = tx * kTileW, y0 = ty * kTileH;
            const float2 cc = col_cs[min(max(x0 + kTileW / 2 - 1, 0), cam.W - 1)], rr = row_cs[min(max(y0 + kTileH / 2 - 1, 0), cam.H - 1)];
            const int out = sls_tile_outside(&cam.tc, &s_cull_w[owner], (float)x0, (float)y0, cc.x, cc.y, rr.x, rr.y);
            if (valid && out) atomicOr(&s_drop_w[2 * owner + (int)(idx >> 5)], 1u << (idx & 31u));
        }
        __builtin_amdgcn_wave_barrier();
        if (tested) {
            const uint64_t drop = ((uint64_t)s_drop_w[2 * lane + 1] << 32) | (uint64_t)s_drop_w[2 * lane];
            my_mask &= ~drop;
            my_tiles = (uint32_t)__popcll(my_mask);
        }
        __builtin_amdgcn_wave_barrier();      // (the slice is reused below)
    }
    if (i < N) {
        if (tiles) tiles[i] = my_tiles;
        if (pa.sbox) pa.sbox[i] = my_tiles ? make_block_box(q4.x, q4.y, q4.z, q4.w, (cam.GX * kTileW) / 8) : 0u;
        if (tile_mask) tile_mask[i] = my_mask;
        // what the emission reads, in ONE 16-byte gather: the rectangle (16-bit fields) and the mask
        // (direct binning, sls_sort.hip: 8 bytes instead — the rectangle in one word and the block box; D10 is off there)
        if (erec && pa.erec_box)
            reinterpret_cast<uint2 *>(erec)[i] = make_uint2((uint32_t)my_rc.x | ((uint32_t)my_rc.y << 9) | ((uint32_t)my_rc.z << 19) | ((uint32_t)my_rc.w << 25),
                                                            my_tiles ? make_block_box(q4.x, q4.y, q4.z, q4.w, (cam.GX * kTileW) / 8) : 0u);
        else if (erec) erec[i] = make_int4(my_rc.x | (my_rc.z << 16), my_rc.y | (my_rc.w << 16), (int)(uint32_t)my_mask, (int)(uint32_t)(my_mask >> 32));
    }
    {   // the 80-byte records leave through LDS so that every store instruction writes 1 KB of
        // consecutive addresses (a direct store would touch 40 cache lines per instruction)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float4 *t = reinterpret_cast<float4 *>(s_mem + (size_t)wave * kPreSliceBytes);
        t[lane * kRec4 + 0] = q0; t[lane * kRec4 + 1] = q1; t[lane * kRec4 + 2] = q2;
        t[lane * kRec4 + 3] = q3; t[lane * kRec4 + 4] = q4;
        __builtin_amdgcn_wave_barrier();
        const size_t base = ((size_t)first + (size_t)wave * 64) * kRec4, end = (size_t)N * kRec4;
#pragma unroll
        for (int k = 0; k < kRec4; ++k) {
            const size_t idx = base + (size_t)(k * 64 + lane);
            if (idx < end) rec[idx] = t[k * 64 + lane];
        }
    }
    if (ra.pen != 0.0f && ra.reg_out) {   // block reduction of the regulariser, one atomic per block
        float v = my_reg;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.0f;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) tot += s_part[k];
            if (tot != 0.0f) atomicAdd(ra.reg_out, tot);
        }
    }
}

__global__ __launch_bounds__(256) void preprocess_fwd_kernel(DevCam cam, RegArgs ra, int N, PreFwdArgs pa)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[4 * kPreSliceBytes];
    __shared__ float s_part[4];
    preprocess_fwd_body<4>(cam, ra, N, pa, (int)blockIdx.x * 256, s_mem, s_part);
}

// The same in ONE launch with step A of the depth-order repair (sls_resort.hpp): the first `n_windows` workgroups
// sort windows of the previous order by the NEW depth keys — which they compute themselves from the centres, with
// the arithmetic of surfel_geom<true> — the others preprocess 512 surfels each.  The window sort is a chain of
// latencies (gathers, LDS stages), the preprocess is VALU bound: side by side they cost little more than the
// longer of the two, and the iteration has one dependent launch less.
// (six waves per SIMD = three workgroups per CU: at the 83 registers the compiler would take, a CU holds two — 2.9
//  generations of workgroups at 500 k surfels instead of 1.9; the cap costs two spilled words)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void preprocess_fwd_resort_kernel(DevCam cam, RegArgs ra, int N, PreFwdArgs pa, int n_windows,
                                                                    const uint32_t *__restrict__ prev_order,
                                                                    uint64_t *__restrict__ comp)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[8 * kPreSliceBytes];
    __shared__ float s_part[8];
    static_assert(8 * kPreSliceBytes >= (int)(kResortThreads * sizeof(ulonglong2)), "the window sort's pairs fit the slices");
    if ((int)blockIdx.x < n_windows) {
        const float *__restrict__ means = pa.means;
        resort_sort_window((int)blockIdx.x, N, prev_order, [&](uint32_t g) {
            const float m0 = means[3 * (size_t)g], m1 = means[3 * (size_t)g + 1], m2 = means[3 * (size_t)g + 2];
            float p[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = fmaf(cam.R[3 * k], m0, fmaf(cam.R[3 * k + 1], m1, fmaf(cam.R[3 * k + 2], m2, cam.t[k])));
            const float rho2 = fmaf(p[2], p[2], fmaf(p[0], p[0], p[1] * p[1]));
            return depth_order_key(sqrtf(rho2));
        }, comp, reinterpret_cast<ulonglong2 *>(s_mem));
        return;
    }
    preprocess_fwd_body<8>(cam, ra, N, pa, ((int)blockIdx.x - n_windows) * 512, s_mem, s_part);
}

// ---------------------------------------------------------------------------
// A8 preprocess backward: gradient record (sls_spec.h) -> input gradients.
// ---------------------------------------------------------------------------
// With ra.raw the inputs are raw parameters, the activations are re-applied here
// and the outputs are gradients w.r.t. the RAW parameters (exp / sigmoid /
// normalize backward + the scale regulariser's gradient), opacity needed too.
// With af.enabled (raw mode only) the Adam update of the surfel's ten parameters follows at once:
// the gradients never travel through HBM (af.write_grads = 0) and no separate optimiser launch
// reads the parameters again.  The parameter pointers are therefore NOT restrict-qualified.
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    DevCam cam, RegArgs ra, AdamFuse af, int N, float *means, float2 *scales, float4 *rots, float *opac,
    const int *__restrict__ radii, float4 *grec, float *__restrict__ dmeans,
    float2 *__restrict__ dscales, float4 *__restrict__ drots, float *__restrict__ dopac)
{
    // Passenger workgroups come FIRST in the grid: they start with the launch and run beside the surfels' blocks.  Each
    // is a chain of latencies (cold loads, barriers, a system-scope fence) with little work: at the end of the grid — or
    // in front of a block's surfel work — it would outlast the launch at the reference's sizes (+3 us at 50 k surfels).
    //   block 0 (af.publisher): the iteration's status — loss sums, regulariser, void flags, the pinned host mirror;
    //   8 blocks (af.order_out): the launch order of the keyframe's next tile backward, one counting sort per XCD.
    const int n_pub = af.publisher ? 1 : 0, n_ord = af.order_out ? 8 : 0;
    if ((int)blockIdx.x < n_pub) {
        if (af.loss_partials) {
            // the tile backward's blocks left their pixels' loss terms (three arrays of n_loss_partials floats, a multiple
            // of 16): the iteration's three sums and the pixel-loss total, in a fixed order — every thread a strided
            // share of 16-byte loads (all in flight at once), then the wave, then the block's four waves
            __shared__ float s_loss[3][4];
            const int n4 = af.n_loss_partials / 4;
            float t[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 *src = reinterpret_cast<const float4 *>(af.loss_partials + (size_t)k * af.n_loss_partials);
#pragma unroll 8
                for (int b = threadIdx.x; b < n4; b += 256) {
                    const float4 v = src[b];
                    t[k] += (v.x + v.y) + (v.z + v.w);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                t[0] += __shfl_down(t[0], off, 64); t[1] += __shfl_down(t[1], off, 64); t[2] += __shfl_down(t[2], off, 64);
            }
            if ((threadIdx.x & 63) == 0) { s_loss[0][threadIdx.x >> 6] = t[0]; s_loss[1][threadIdx.x >> 6] = t[1]; s_loss[2][threadIdx.x >> 6] = t[2]; }
            __syncthreads();
            if (threadIdx.x == 0) {
                float r[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) r[k] = (s_loss[k][0] + s_loss[k][1]) + (s_loss[k][2] + s_loss[k][3]);
                af.status_src[2] = __float_as_uint(r[0]); af.status_src[3] = __float_as_uint(r[1]); af.status_src[4] = __float_as_uint(r[2]);
                af.status_src[5] = __float_as_uint(r[0] * af.loss_w[0] + af.loss_w[1] * r[1] + af.loss_w[2] * r[2]);
            }
        }
        if (threadIdx.x == 0) {
            if (af.reg_accum) {   // the regulariser was summed in the workspace: publish it, leave zero for the next iteration
                af.status_src[6] = __float_as_uint(*af.reg_accum);     // (same type as the mirror's reads below: no aliasing games)
                *af.reg_accum = 0.0f;
            }
            if (af.void_flags) {   // keyframe-parallel mode: the void bits as two floats that can ride a SUM collective
                const uint32_t bits = af.status_src[1];
                for (int g = 0; g < (af.void_count > 0 ? af.void_count : 1); ++g) {
                    af.void_flags[(size_t)g * af.void_stride + 0] = (bits & 1u) ? 1.0f : 0.0f;
                    af.void_flags[(size_t)g * af.void_stride + 1] = (bits & ~1u) ? 1.0f : 0.0f;
                }
            }
            if (af.status_mirror) mirror_status_block(af.status_src, af.status_mirror);
            if (af.grad_bitmap) {   // sparse exchange: this rank's verdict behind the bitmap (OR-reduced with the bitmap)
                const uint32_t bits = af.status_src[1];
                af.grad_bitmap[af.grad_bitmap_words] = (bits & 1u) ? 1ull : 0ull;
                af.grad_bitmap[af.grad_bitmap_words + 1] = (bits & ~1u) ? 1ull : 0ull;
            }
        }
        return;
    }
    if ((int)blockIdx.x < n_pub + n_ord) {
        const int xcd = (int)blockIdx.x - n_pub;
        order_blocks_by_cost(af.order_T, af.order_cost, af.order_out + 1, xcd);
        if (xcd == 0 && threadIdx.x == 0) af.order_out[0] = block_order_tag(af.order_T);
        return;
    }
    const int i = ((int)blockIdx.x - n_pub - n_ord) * 256 + threadIdx.x;
    if (i >= N) return;
    float dm[3] = { 0, 0, 0 };
    float2 ds = make_float2(0, 0);
    float4 dq = make_float4(0, 0, 0, 0);
    float dop = 0.0f;
    const float2 s_in = scales[i];
    float2 s = s_in;
    const float4 q_in = rots[i];
    float4 q = q_in;
    const float o_in = ra.raw ? opac[i] : 0.0f;
    float o = o_in;
    activate<false>(ra, s, q, o);
    const float m[3] = { means[3 * i], means[3 * i + 1], means[3 * i + 2] };
    // (an unmarked surfel's record is all zeros and so is its gradient: nothing to read, nothing to clear)
    const bool marked = !af.touched || af.touched[i] != 0;
    if (marked && af.touched) af.touched[i] = 0;
    if (radii[i] > 0 && marked) {
        SurfelGeom g;
        surfel_geom<false>(cam, m, s, q, g);
        float4 g0, g1, g2, g3;
        if (af.det_acc) {
            // deterministic accumulation: the record is the fixed-point sum scaled back — by the exact maximum of the
            // two-pass scheme, or by the predicted scale of the one-pass scheme
            float gv[16];
            uint32_t prev[4] = { 0u, 0u, 0u, 0u };
            if (af.det_onepass) {
                const uint4 pv = reinterpret_cast<const uint4 *>(af.det_prev)[i];
                prev[0] = pv.x; prev[1] = pv.y; prev[2] = pv.z; prev[3] = pv.w;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int ex;
                if (af.det_onepass) {
                    const uint32_t pb = (prev[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                    ex = det_scale_exp(pb, af.det_gex[k]);
                } else {
                    ex = (int)((af.det_max[(size_t)i * 16 + k] >> 23) & 0xFFu);
                }
                const long long acc = af.det_acc[(size_t)i * 16 + k];
                gv[k] = (float)ldexp((double)acc, ex - 166);
                // (cleared where read: a one-launch iteration that follows finds the accumulators at zero)
                if (af.det_onepass || af.clear_grec) const_cast<long long *>(af.det_acc)[(size_t)i * 16 + k] = 0;
            }
            if (af.det_prev && (!af.status_src || af.status_src[1] == 0u)) {
                // the scale the keyframe's NEXT iteration will use (never from a void iteration's sums)
                uint32_t pw[4];
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    pw[w4] = 0u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) pw[w4] |= det_predict(gv[4 * w4 + b]) << (8 * b);
                }
                reinterpret_cast<uint4 *>(af.det_prev)[i] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
                if (!af.det_onepass && af.det_gex) {
                    // the fields' defaults: raised by two-pass iterations only (a one-pass iteration reads them)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const uint32_t pe = (pw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                        if (pe > af.det_gex[k]) atomicMax(&af.det_gex[k], pe);
                    }
                }
            }
            g0 = make_float4(gv[0], gv[1], gv[2], gv[3]); g1 = make_float4(gv[4], gv[5], gv[6], gv[7]);
            g2 = make_float4(gv[8], gv[9], gv[10], gv[11]); g3 = make_float4(gv[12], gv[13], gv[14], gv[15]);
        } else {
            g0 = grec[(size_t)i * 4 + 0]; g1 = grec[(size_t)i * 4 + 1];
            g2 = grec[(size_t)i * 4 + 2]; g3 = grec[(size_t)i * 4 + 3];
        }
        if (af.clear_grec && !af.det_acc) {   // only records of visible surfels are ever touched by the tile kernel
            const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            grec[(size_t)i * 4 + 0] = z; grec[(size_t)i * 4 + 1] = z; grec[(size_t)i * 4 + 2] = z; grec[(size_t)i * 4 + 3] = z;
        }
        const float gHu[3] = { g0.x, g0.y, g0.z }, gHv[3] = { g1.x, g1.y, g1.z }, gn[3] = { g2.x, g2.y, g2.z };
        const float gnpv = g0.w, grhoc = g1.w, go = g2.w, Su = g3.x, Sv = g3.y, gcpx = g3.z, gcpy = g3.w;
        float dc[3];
        const float irho = frcp(g.rho), isu = frcp(g.su), isv = frcp(g.sv);
#pragma unroll
        for (int k = 0; k < 3; ++k) dc[k] = g.p[k] * irho;
        float dp[3], dA[3], dB[3], t1[3], t2[3];
        cross3(g.p, gHu, dA); cross3(gHu, g.A, t1);
        cross3(g.p, gHv, dB); cross3(gHv, g.B, t2);
        float dTu[3], dTv[3], dTn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dp[k] = t1[k] + t2[k];
            dTv[k] = g.sig * dA[k] * isu;
            dTu[k] = -g.sig * dB[k] * isv;
        }
        const float dsu = -dot3(dA, g.A) * isu, dsv = -dot3(dB, g.B) * isv;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dTn[k] = g.sig * (gn[k] + gnpv * g.p[k]);
            dp[k] += gnpv * g.n[k];
            dp[k] += grhoc * dc[k];
        }
        float gdc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) gdc[k] = -(Su * g.Hu[k] + Sv * g.Hv[k]);
        const float gd = dot3(gdc, dc);
#pragma unroll
        for (int k = 0; k < 3; ++k) dp[k] += (gdc[k] - gd * dc[k]) * irho;
        if (g.rxy2 > 1e-30f) {
            const float gaz = gcpx * cam.fx, gel = gcpy * cam.fy;
            const float irxy2 = frcp(g.rxy2), irr = frcp(g.rxy * g.rho2), irho2 = irho * irho;
            dp[0] += gaz * (-g.p[1] * irxy2) + gel * (-g.p[2] * g.p[0] * irr);
            dp[1] += gaz * (g.p[0] * irxy2) + gel * (-g.p[2] * g.p[1] * irr);
            dp[2] += gel * (g.rxy * irho2);
        }
        matTvec(cam.R, dp, dm);
        ds = make_float2(cam.mod * dsu, cam.mod * dsv);
        dop = go;
        float G0[3], G1[3], G2[3];
        matTvec(cam.R, dTu, G0); matTvec(cam.R, dTv, G1); matTvec(cam.R, dTn, G2);
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        // G[i][j] = dL/dR_ij, column j in {tu, tv, tn}: Gj[i]
        dq.x = 2.0f * (-z * G1[0] + y * G2[0] + z * G0[1] - x * G2[1] - y * G0[2] + x * G1[2]);
        dq.y = 2.0f * (y * G1[0] + z * G2[0] + y * G0[1] - 2.0f * x * G1[1] - r * G2[1] + z * G0[2] + r * G1[2] - 2.0f * x * G2[2]);
        dq.z = 2.0f * (-2.0f * y * G0[0] + x * G1[0] + r * G2[0] + x * G0[1] + z * G2[1] - r * G0[2] + z * G1[2] - 2.0f * y * G2[2]);
        dq.w = 2.0f * (-2.0f * z * G0[0] - r * G1[0] + x * G2[0] + r * G0[1] - 2.0f * z * G1[1] + y * G2[1] + x * G0[2] + y * G1[2]);
    }
    if (ra.pen != 0.0f) {   // d/ds of pen * relu(max(s.x, s.y) - smax); torch.max picks the first index on ties
        const bool first = s.x >= s.y;
        if ((first ? s.x : s.y) >= ra.smax) { if (first) ds.x += ra.pen; else ds.y += ra.pen; }
    }
    if (ra.raw) {
        ds.x *= s.x; ds.y *= s.y;                                   // exp backward
        dop *= o * (1.0f - o);                                      // sigmoid backward
        const float nrm = fsqrt(q_in.x * q_in.x + q_in.y * q_in.y + q_in.z * q_in.z + q_in.w * q_in.w);
        const float inv = frcp(fmaxf(nrm, 1e-12f));                 // F.normalize backward
        const float dd = dq.x * q.x + dq.y * q.y + dq.z * q.z + dq.w * q.w;
        dq.x = (dq.x - dd * q.x) * inv; dq.y = (dq.y - dd * q.y) * inv;
        dq.z = (dq.z - dd * q.z) * inv; dq.w = (dq.w - dd * q.w) * inv;
    }
    if (af.grad_bitmap) {   // which surfels have anything to exchange (one 64-bit word per wave)
        const bool nz = dm[0] != 0.0f || dm[1] != 0.0f || dm[2] != 0.0f || dop != 0.0f || ds.x != 0.0f || ds.y != 0.0f ||
                        dq.x != 0.0f || dq.y != 0.0f || dq.z != 0.0f || dq.w != 0.0f;
        const uint64_t bal = __ballot(nz);
        if ((threadIdx.x & 63) == 0) af.grad_bitmap[i >> 6] = bal;
    }
    if (af.gchunk) {
        // reduce-scatter layout: flat element e of [xyz 3N | opacity N | scaling 2N | rotation 4N] lives at
        // e + 4 * (e / C).  C is a multiple of 4 and N even, so a float2 / float4 never straddles a chunk end.
        const uint32_t C = af.gchunk, n = (uint32_t)N, ui = (uint32_t)i;
        uint32_t e = 3u * ui, qd = e / C, lim = (qd + 1u) * C;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (e + (uint32_t)k >= lim) { ++qd; lim += C; }
            af.gbase[(size_t)(e + (uint32_t)k) + 4u * (size_t)qd] = dm[k];
        }
        e = 3u * n + ui;      af.gbase[(size_t)e + 4u * (size_t)(e / C)] = dop;
        e = 4u * n + 2u * ui; *reinterpret_cast<float2 *>(af.gbase + (size_t)e + 4u * (size_t)(e / C)) = ds;
        e = 6u * n + 4u * ui; *reinterpret_cast<float4 *>(af.gbase + (size_t)e + 4u * (size_t)(e / C)) = dq;
    } else if (!af.enabled || af.write_grads) {
        dmeans[3 * i] = dm[0]; dmeans[3 * i + 1] = dm[1]; dmeans[3 * i + 2] = dm[2];
        dscales[i] = ds;
        drots[i] = dq;
        dopac[i] = dop;
    }
    if (af.union_bitmap) {
        // keyframe-parallel, touched-set exchange: the union's surfels hand their rows to the collective (slot = rank of
        // the surfel inside the union, the same on every rank); the others have a zero gradient on EVERY rank and are
        // updated below like any surfel of the one-GPU iteration
        const uint64_t word = af.union_bitmap[i >> 6];
        const int b = i & 63;
        if ((word >> b) & 1ull) {
            const uint32_t slot = af.union_prefix[i >> 6] + (uint32_t)__popcll(word & ((1ull << b) - 1ull));
            if (slot < af.compact_cap) {
                float *o = af.compact + (size_t)slot * 10;
                o[0] = dm[0]; o[1] = dm[1]; o[2] = dm[2]; o[3] = dop; o[4] = ds.x; o[5] = ds.y;
                o[6] = dq.x; o[7] = dq.y; o[8] = dq.z; o[9] = dq.w;
                af.compact_idx[slot] = (uint32_t)i;
            }
            return;
        }
        // (the early bitmap is a superset of the non-zero gradients by construction: say so if it ever is not)
        if (dm[0] != 0.0f || dm[1] != 0.0f || dm[2] != 0.0f || dop != 0.0f || ds.x != 0.0f || ds.y != 0.0f ||
            dq.x != 0.0f || dq.y != 0.0f || dq.z != 0.0f || dq.w != 0.0f)
            if (af.status_src) atomicOr(af.status_src + 1, 32u);
    }
    if (af.enabled && *af.skip_flag == 0u) {
        // moments in the bucket layout [xyz 3N | opacity N | scaling 2N | rotation 4N]
        const size_t n = (size_t)N, ix = 3 * (size_t)i, io = 3 * n + i, is = 4 * n + 2 * (size_t)i, ir = 6 * n + 4 * (size_t)i;
        float *M = af.exp_avg, *V = af.exp_avg_sq;
        float p, mm, vv;
        const float ibc1 = frcp(af.c.bc1);
        const float st_x = af.lr_xyz * ibc1, st_o = af.lr_opacity * ibc1;
        const float st_s = af.lr_scaling * ibc1, st_r = af.lr_rotation * ibc1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p = m[k]; mm = M[ix + k]; vv = V[ix + k];
            adam_one(p, dm[k], mm, vv, st_x, af.c);
            means[3 * i + k] = p; M[ix + k] = mm; V[ix + k] = vv;
        }
        p = o_in; mm = M[io]; vv = V[io];
        adam_one(p, dop, mm, vv, st_o, af.c);
        opac[i] = p; M[io] = mm; V[io] = vv;
        float2 ps = s_in, ms = *reinterpret_cast<float2 *>(M + is), vs = *reinterpret_cast<float2 *>(V + is);
        adam_one(ps.x, ds.x, ms.x, vs.x, st_s, af.c);
        adam_one(ps.y, ds.y, ms.y, vs.y, st_s, af.c);
        scales[i] = ps; *reinterpret_cast<float2 *>(M + is) = ms; *reinterpret_cast<float2 *>(V + is) = vs;
        // (4N + 2i and 6N + 4i floats: 8- resp. 16-byte aligned whenever the bucket is 16-byte aligned and N is even;
        //  the launcher falls back to the separate optimiser kernel otherwise)
        float4 pq = q_in, mq = *reinterpret_cast<float4 *>(M + ir), vq = *reinterpret_cast<float4 *>(V + ir);
        adam_one(pq.x, dq.x, mq.x, vq.x, st_r, af.c);
        adam_one(pq.y, dq.y, mq.y, vq.y, st_r, af.c);
        adam_one(pq.z, dq.z, mq.z, vq.z, st_r, af.c);
        adam_one(pq.w, dq.w, mq.w, vq.w, st_r, af.c);
        rots[i] = pq; *reinterpret_cast<float4 *>(M + ir) = mq; *reinterpret_cast<float4 *>(V + ir) = vq;
    }
}

__global__ __launch_bounds__(256) void mark_visible_kernel(DevCam cam, int N, const float *__restrict__ means,
                                                           uint8_t *__restrict__ visible)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        p[k] = fmaf(cam.R[3 * k], means[3 * i], fmaf(cam.R[3 * k + 1], means[3 * i + 1], fmaf(cam.R[3 * k + 2], means[3 * i + 2], cam.t[k])));
    const float rho = sqrtf(fmaf(p[2], p[2], fmaf(p[0], p[0], p[1] * p[1])));
    visible[i] = (rho >= cam.near_c && rho < 1.0e18f) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// host-side launchers used by sls_api.hip
// ---------------------------------------------------------------------------
int launch_preprocess_fwd(const DevCam &cam, int raw, float smax, float pen, float *reg_out, int N,
                          const float *means, const float *scales, const float *rots, const float *opac, float *rec,
                          int32_t *radii, int32_t *rect, uint32_t *tiles, float *depth, uint32_t *order_keys,
                          uint32_t *order_vals, uint32_t *n_dev, hipStream_t st, uint32_t *status_clear,
                          const float *col_cs, const float *row_cs, uint64_t *tile_mask, int32_t *erec,
                          const uint32_t *resort_prev_order, uint64_t *resort_comp, uint32_t *sbox, int erec_box,
                          uint32_t *zero_words, int n_zero_words)
{
    RegArgs ra;
    ra.raw = raw; ra.smax = smax; ra.pen = pen; ra.reg_out = reg_out; ra.status_clear = status_clear;
    ra.zero_words = zero_words; ra.n_zero_words = n_zero_words;
    PreFwdArgs pa;
    pa.means = means; pa.scales = (const float2 *)scales; pa.rots = (const float4 *)rots; pa.opac = opac;
    pa.rec = (float4 *)rec; pa.radii = radii; pa.rect = (int4 *)rect; pa.tiles = tiles; pa.depth = depth;
    pa.order_keys = order_keys; pa.order_vals = order_vals; pa.n_dev = n_dev;
    pa.col_cs = (const float2 *)col_cs; pa.row_cs = (const float2 *)row_cs;
    pa.tile_mask = (col_cs && row_cs) ? tile_mask : nullptr;
    pa.erec = (cam.GX < 65536 && cam.GY < 65536) ? (int4 *)erec : nullptr;
    pa.sbox = block_box_fits(cam.GX * kTileW, cam.H) ? sbox : nullptr;
    pa.erec_box = (erec_box && pa.erec && block_box_fits(cam.GX * kTileW, cam.H)) ? 1 : 0;
    ScopedTimer tm(T_PREPROCESS_FWD, st);
    if (resort_prev_order && resort_comp) {
        // merged with the repair's window sort (which overwrites the sort's identity permutation: not written here)
        pa.order_vals = nullptr;
        const int nw = (N + kResortWindow - 1) / kResortWindow, nb = (N + 511) / 512;
        hipLaunchKernelGGL(preprocess_fwd_resort_kernel, dim3(nw + nb), dim3(512), 0, st, cam, ra, N, pa, nw, resort_prev_order,
                           resort_comp);
        SLS_LAUNCH_CHECK("preprocess_fwd_resort_kernel");
        return SLS_OK;
    }
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, cam, ra, N, pa);
    SLS_LAUNCH_CHECK("preprocess_fwd_kernel");
    return SLS_OK;
}

int launch_preprocess_bwd(const DevCam &cam, int raw, float smax, float pen, int N, const float *means,
                          const float *scales, const float *rots, const float *opac, const int32_t *radii,
                          const float *grec, float *dmeans, float *dscales, float *drots, float *dopac,
                          hipStream_t st, const AdamFuse *fuse)
{
    const int nb = (N + 255) / 256;
    RegArgs ra;
    ra.raw = raw; ra.smax = smax; ra.pen = pen; ra.reg_out = nullptr; ra.status_clear = nullptr;
    ra.zero_words = nullptr; ra.n_zero_words = 0;
    AdamFuse af;
    memset(&af, 0, sizeof(af));
    if (fuse) af = *fuse;
    // (a block of its own for the status duties, if there are any: see the kernel)
    af.publisher = (af.loss_partials || af.reg_accum || af.void_flags || af.status_mirror || af.grad_bitmap) ? 1 : 0;
    ScopedTimer tm(T_PREPROCESS_BWD, st);
    // (parameters are only written when af.enabled, which the caller sets for its own mutable tensors)
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(nb + af.publisher + (af.order_out ? 8 : 0)), dim3(256), 0, st, cam, ra, af, N, const_cast<float *>(means),
                       (float2 *)const_cast<float *>(scales), (float4 *)const_cast<float *>(rots),
                       const_cast<float *>(opac), radii, (float4 *)const_cast<float *>(grec), dmeans, (float2 *)dscales,
                       (float4 *)drots, dopac);
    SLS_LAUNCH_CHECK("preprocess_bwd_kernel");
    return SLS_OK;
}

int launch_mark_visible(const DevCam &cam, int N, const float *means, uint8_t *visible, hipStream_t st)
{
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(mark_visible_kernel, dim3(nb), dim3(256), 0, st, cam, N, means, visible);
    SLS_LAUNCH_CHECK("mark_visible_kernel");
    return SLS_OK;
}

}  // namespace sls
