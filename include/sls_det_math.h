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
#define SLS_R(x) x##f
#endif

#define SLS_PI SLS_R(3.14159265358979323846)
#define SLS_PIO2 SLS_R(1.57079632679489661923)

/* atan(a) for a in [0,1]:  a * P(a*a), P = degree-8 Chebyshev fit of
 * atan(sqrt(s))/sqrt(s) on s in [0,1] (tools/fit_atan.py). */
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

#endif /* SLS_DET_MATH_H */
