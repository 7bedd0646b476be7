/*
 * sls_det_math.h — deterministic elementary functions shared by the HIP
 * preprocess kernel and by any CPU checker that wants to reproduce its
 * INTEGER outputs (tile rectangles, tiles_touched, radii, depth-key bits)
 * bit for bit.
 *
 * Why this exists: the north-star bar is "tile/key integers bit-exact".
 * The tile rectangle of a surfel is floor() of a pixel coordinate that
 * comes out of atan2/asin of the view-space centre.  Vendor libm atan2f on
 * the host and the device's ocml atan2f differ in the last ulp, and one ulp
 * is enough to move a floor() across a tile boundary.  So the projection
 * uses ONLY operations that IEEE-754 defines exactly (+ - * / sqrt fma,
 * compare/select); atan2 is a fixed odd polynomial evaluated with fmaf.
 * Max abs error of sls_atan2 vs. the real atan2 is < 4e-7 rad in float
 * (polynomial 1.1e-7 + one ulp at |angle| ~ pi), i.e. < 1.4e-4 px at
 * |fx| = 326 px/rad), see tests/test_oracle.py::test_det_atan2_accuracy.
 *
 * Rules for users of this header (both sides):
 *   - compile with -ffp-contract=off (no implicit fma), no fast-math;
 *   - HIP: keep -fhip-fp32-correctly-rounded-divide-sqrt (the default);
 *   - write every multiply-add explicitly with SLS_FMA.
 *
 * The header is plain C99 / HIP device compatible.  `sls_real` is float
 * unless SLS_REAL_IS_DOUBLE is defined (the float64 build of the CPU
 * checker uses the same polynomial so both precisions share one spec).
 */
#ifndef SLS_DET_MATH_H
#define SLS_DET_MATH_H

#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define SLS_HD __host__ __device__ __forceinline__
#else
#define SLS_HD static inline
#endif

#ifdef SLS_REAL_IS_DOUBLE
typedef double sls_real;
#define SLS_FMA(a, b, c) fma((a), (b), (c))
#define SLS_SQRT(a) sqrt(a)
#define SLS_FABS(a) fabs(a)
#define SLS_FMAX(a, b) fmax((a), (b))
#define SLS_FMIN(a, b) fmin((a), (b))
#define SLS_FLOOR(a) floor(a)
#define SLS_CEIL(a) ceil(a)
#define SLS_RINT(a) rint(a)
#define SLS_R(x) x
#else
typedef float sls_real;
#define SLS_FMA(a, b, c) fmaf((a), (b), (c))
#define SLS_SQRT(a) sqrtf(a)
#define SLS_FABS(a) fabsf(a)
#define SLS_FMAX(a, b) fmaxf((a), (b))
#define SLS_FMIN(a, b) fminf((a), (b))
#define SLS_FLOOR(a) floorf(a)
#define SLS_CEIL(a) ceilf(a)
#define SLS_RINT(a) rintf(a)
#define SLS_R(x) x##f
#endif

#define SLS_PI SLS_R(3.14159265358979323846)
#define SLS_PIO2 SLS_R(1.57079632679489661923)

/* atan(a) for a in [0,1]:  a * P(a*a), P = degree-8 Chebyshev fit of
 * atan(sqrt(s))/sqrt(s) on s in [0,1] (a least-squares fit made in round 1; its accuracy is pinned by tests/test_oracle.py::test_det_atan2_accuracy). */
SLS_HD sls_real sls_atan_unit(sls_real a)
{
    const sls_real s = a * a;
    sls_real r = SLS_R(0.0028340641874819994);
    r = SLS_FMA(r, s, SLS_R(-0.016005029901862144));
    r = SLS_FMA(r, s, SLS_R(0.042587608098983765));
    r = SLS_FMA(r, s, SLS_R(-0.07495445758104324));
    r = SLS_FMA(r, s, SLS_R(0.10636754333972931));
    r = SLS_FMA(r, s, SLS_R(-0.14202570915222168));
    r = SLS_FMA(r, s, SLS_R(0.19992484152317047));
    r = SLS_FMA(r, s, SLS_R(-0.3333306610584259));
    r = SLS_FMA(r, s, SLS_R(1.0));
    return r * a;
}

/* Full-range atan2 with the usual quadrant conventions; (0,0) -> 0.
 * NaN inputs are not supported (callers cull non-finite centres first). */
SLS_HD sls_real sls_atan2(sls_real y, sls_real x)
{
    const sls_real ax = SLS_FABS(x), ay = SLS_FABS(y);
    const sls_real mx = SLS_FMAX(ax, ay), mn = SLS_FMIN(ax, ay);
    if (!(mx > SLS_R(0.0))) return SLS_R(0.0);
    sls_real r = sls_atan_unit(mn / mx);
    if (ay > ax) r = SLS_PIO2 - r;
    if (x < SLS_R(0.0)) r = SLS_PI - r;
    if (y < SLS_R(0.0)) r = -r;
    return r;
}

/* asin for s in [0,1) through atan2(s, sqrt(1-s*s)). */
SLS_HD sls_real sls_asin01(sls_real s)
{
    return sls_atan2(s, SLS_SQRT(SLS_FMA(-s, s, SLS_R(1.0))));
}


/* ------------------------------------------------------------------------------------------------
 * D10: footprint test of a surfel against a whole TILE in the binning (which instances exist at all).
 *
 * A pixel ray d can only receive alpha >= 1/255 from a surfel's 3D branch if
 *     G(d) = |(Hu.d, Hv.d)| + kc (n.d) <= 0      (rho3 <= kc^2 = 2 ln(255 o) and n.d < 0),
 * and G is convex in d.  With the tile's rays written as d0 + x Dx + y Dy + r, |x|,|y| <= 1, |r| <= eps
 * (second-order remainder of the sphere's parametrisation), G over the tile is bounded below by
 *     G(d0) - |gradG.Dx| - |gradG.Dy| - eps |gradG|_1 ;
 * if that is positive no pixel of the tile lies in the 3D footprint.  The low-pass (2D) branch reaches
 * kc / sqrt 2 pixels from the centre pixel: tested as the distance to the tile's pixel box.  An instance
 * (surfel, tile) of the surfel's tile rectangle is emitted iff the 3D footprint may reach the tile OR the
 * disc does.  Everything is multiplied through by su sv |(a, b)| so that no division is needed, and only
 * exactly-rounded operations in a fixed order are used (see the rules at the top of this file): the HIP
 * kernel and the CPU checker reach the same verdict for every instance, so tiles_touched, the sorted lists
 * and the tile ranges stay bit-exact.  Conservative by construction (margins: 0.1 % + 1e-3 on rho_max,
 * 2e-4 relative on G, 0.05 px on the disc); tests/test_tile_cull.py checks on whole scenes that the
 * rendered image does not change by a single bit when the test is switched off.
 * ------------------------------------------------------------------------------------------------ */

/* Upper bound of ln(x) for a finite x >= 1 (x = 255 o <= 255): x = 2^e m, m in [1,2) straight from the bit pattern,
 * ln m = t P(t), t = m - 1, P = degree-5 fit of ln(1+t)/t on [0,1] (float32 Horner evaluation within 6.1e-6 of ln m);
 * 2e-5 is added.  No division, no loop: ~15 instructions. */
SLS_HD sls_real sls_log_upper(sls_real x)
{
#ifdef SLS_REAL_IS_DOUBLE
    int e = 0;
    sls_real m = x;
    for (int k = 0; k < 1100 && m >= SLS_R(2.0); ++k) { m = m * SLS_R(0.5); ++e; }
#else
    unsigned int bits;
    __builtin_memcpy(&bits, &x, 4);
    const int e = (int)(bits >> 23) - 127;
    bits = (bits & 0x007FFFFFu) | 0x3F800000u;
    sls_real m;
    __builtin_memcpy(&m, &bits, 4);
#endif
    const sls_real t = m - SLS_R(1.0);
    sls_real p = SLS_R(-0.02397957257926464);
    p = SLS_FMA(p, t, SLS_R(0.10150004923343658));
    p = SLS_FMA(p, t, SLS_R(-0.2102936953306198));
    p = SLS_FMA(p, t, SLS_R(0.3252951502799988));
    p = SLS_FMA(p, t, SLS_R(-0.49937260150909424));
    p = SLS_FMA(p, t, SLS_R(0.9999918341636658));
    return SLS_FMA((sls_real)e, SLS_R(0.69314718055994531), p * t) + SLS_R(2.0e-5);
}

/* Per-surfel inputs of the tile test (all from exactly reproducible quantities). */
typedef struct SlsTileCullSurfel {
    sls_real Pu[3], Pv[3];   /* sv (Tv x p), su (Tu x p): su sv times the record's Hu, Hv up to sign */
    sls_real n[3], dc[3];    /* sensor-facing normal, unit centre direction p / |p| */
    sls_real K2;             /* (kc su sv)^2 with kc^2 = rho_max = 2 ln(255 o) (+ margins) */
    sls_real rd2;            /* squared reach of the low-pass disc in pixels: >= (kc / sqrt 2 (+0.1 %) + 0.05)^2 */
    sls_real cpx, cpy;       /* centre pixel */
} SlsTileCullSurfel;

/* Per-camera constants of the tile test (host: sls_tile_cull_consts / the checker's copy of it). */
typedef struct SlsTileCullCam {
    sls_real chx, shx, chy, shy;   /* cos / sin of half a pixel in azimuth (0.5 / fx) and elevation (0.5 / fy) */
    sls_real kx, ky;               /* half extent of a tile in radians: (tile_w - 1) / 2 / fx, (tile_h - 1) / 2 / fy */
    sls_real eps;                  /* bound of the second-order remainder + float32 slop */
    sls_real hx, hy;               /* (tile_w - 1) / 2, (tile_h - 1) / 2 pixels */
    sls_real wrapW, invW;          /* W and 1 / W in 360-degree mode, else 0 */
} SlsTileCullCam;

/* dc: p / |p|, exactly rounded (the record's unit centre direction) */
SLS_HD void sls_tile_cull_surfel(const sls_real *Tu, const sls_real *Tv, const sls_real *n, const sls_real *p,
                                 const sls_real *dc, sls_real su, sls_real sv, sls_real opacity, sls_real cpx,
                                 sls_real cpy, SlsTileCullSurfel *s)
{
    /* W = T x p with the fma form used for Hu / Hv */
    const sls_real Wu0 = SLS_FMA(Tv[1], p[2], -(Tv[2] * p[1])), Wu1 = SLS_FMA(Tv[2], p[0], -(Tv[0] * p[2])),
                   Wu2 = SLS_FMA(Tv[0], p[1], -(Tv[1] * p[0]));
    const sls_real Wv0 = SLS_FMA(Tu[1], p[2], -(Tu[2] * p[1])), Wv1 = SLS_FMA(Tu[2], p[0], -(Tu[0] * p[2])),
                   Wv2 = SLS_FMA(Tu[0], p[1], -(Tu[1] * p[0]));
    s->Pu[0] = sv * Wu0; s->Pu[1] = sv * Wu1; s->Pu[2] = sv * Wu2;
    s->Pv[0] = su * Wv0; s->Pv[1] = su * Wv1; s->Pv[2] = su * Wv2;
    for (int k = 0; k < 3; ++k) { s->n[k] = n[k]; s->dc[k] = dc[k]; }
    const sls_real lo = SLS_R(255.0) * opacity;
    /* lo <= 1: no pixel can reach alpha >= 1/255 at all (alpha <= o): rho_max = 0 keeps the maths finite.
     * rho_max: 0.1 % + 1e-3 as the block-level cull's, times 1.0002 (its kc carries 1.0001) */
    const sls_real rho_max = lo > SLS_R(1.0)
        ? (SLS_R(2.0) * sls_log_upper(lo) * SLS_R(1.001) + SLS_R(1.0e-3)) * SLS_R(1.0002) : SLS_R(0.0);
    const sls_real ss = su * sv;
    s->K2 = rho_max * (ss * ss);
    /* (c kc + 0.05)^2 with c = 0.70781 and kc <= (kc^2 + 1) / 2 */
    s->rd2 = (SLS_R(0.5009949961) * rho_max + SLS_R(0.0353905) * (rho_max + SLS_R(1.0))) + SLS_R(0.0025);
    s->cpx = cpx; s->cpy = cpy;
}

/* 1: the surfel cannot contribute to any pixel of the tile whose first pixel is (x0, y0); col_c / row_c:
 * (cos, sin) of the pixel column min(x0 + (tile_w - 1) / 2, W - 1) rounded down / of the corresponding row, from
 * the rasterizer's own ray tables (the tile centre lies half a pixel further). */
SLS_HD int sls_tile_outside(const SlsTileCullCam *c, const SlsTileCullSurfel *s, sls_real x0, sls_real y0,
                            sls_real col_cos, sls_real col_sin, sls_real row_cos, sls_real row_sin)
{
    /* tile-centre direction: the pixel's angles advanced by half a pixel */
    const sls_real cc = col_cos * c->chx - col_sin * c->shx, sc = col_sin * c->chx + col_cos * c->shx;
    const sls_real cr = row_cos * c->chy - row_sin * c->shy, sr = row_sin * c->chy + row_cos * c->shy;
    const sls_real d0 = cc * cr, d1 = sc * cr, d2 = sr;
    const sls_real Dx0 = -(c->kx * d1), Dx1 = c->kx * d0;
    const sls_real Dy0 = -(c->ky * (cc * sr)), Dy1 = -(c->ky * (sc * sr)), Dy2 = c->ky * cr;
    const sls_real l0 = d0 - s->dc[0], l1 = d1 - s->dc[1], l2 = d2 - s->dc[2];
    const sls_real a = (s->Pu[0] * l0 + s->Pu[1] * l1) + s->Pu[2] * l2;
    const sls_real b = (s->Pv[0] * l0 + s->Pv[1] * l1) + s->Pv[2] * l2;
    const sls_real e = (s->n[0] * d0 + s->n[1] * d1) + s->n[2] * d2;
    const sls_real n2 = a * a + b * b;
    const sls_real kn = SLS_SQRT(s->K2 * n2);          /* kc su sv |(a, b)| */
    const sls_real g0 = (a * s->Pu[0] + b * s->Pv[0]) + kn * s->n[0];
    const sls_real g1 = (a * s->Pu[1] + b * s->Pv[1]) + kn * s->n[1];
    const sls_real g2 = (a * s->Pu[2] + b * s->Pv[2]) + kn * s->n[2];
    const sls_real tx = g0 * Dx0 + g1 * Dx1;
    const sls_real ty = (g0 * Dy0 + g1 * Dy1) + g2 * Dy2;
    const sls_real reach = (SLS_FABS(tx) + SLS_FABS(ty)) + c->eps * ((SLS_FABS(g0) + SLS_FABS(g1)) + SLS_FABS(g2));
    const int outside3d = ((n2 + kn * e) - reach) > SLS_R(2.0e-4) * (n2 + kn * SLS_FABS(e));
    if (!outside3d) return 0;
    /* low-pass disc against the tile's pixel box */
    const sls_real dxr = (x0 + c->hx) - s->cpx;
    const sls_real dxc = dxr - c->wrapW * SLS_RINT(dxr * c->invW);
    const sls_real ex = SLS_FMAX(SLS_FABS(dxc) - c->hx, SLS_R(0.0));
    const sls_real ey = SLS_FMAX(SLS_FABS((y0 + c->hy) - s->cpy) - c->hy, SLS_R(0.0));
    return (ex * ex + ey * ey) > s->rd2;
}

#endif /* SLS_DET_MATH_H */
