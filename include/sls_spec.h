/*
 * sls_spec.h — numeric constants and buffer layouts of the spherical surfel
 * rasterizer (SURVEY.md §8a decisions D1–D9).  Part of the
 * public boundary: the HIP kernels, the C-ABI and any checker agree on
 * these and nothing else.
 */
#ifndef SLS_SPEC_H
#define SLS_SPEC_H

/* D2 near cut on range; D7 near/far of the distortion mapping. */
#define SLS_NEAR 0.2f
#define SLS_FAR 100.0f
/* D4: binning cutoff (sigma) and minimum pixel radius = cutoff * low-pass
 * filter size sqrt(2)/2 (the filter is G2d = exp(-|dpix|^2), i.e.
 * FilterInvSquare = 2). */
#define SLS_CUTOFF 3.0f
#define SLS_RMIN_PX 2.1213203435596424f
#define SLS_FILTER_INV_SQUARE 2.0f
/* D10: the binning can test a surfel's tile rectangle tile by tile (include/sls_det_math.h) when it has at least
 * SlsCamera.tile_cull_min tiles (and at most 64).  On the bench scene rectangles of >= 3 tiles are 28 % of the
 * surfels, 57 % of the instances and 97 % of what the test removes (-12 % instances).  Default: 0 = OFF since both
 * tile kernels run dense rounds (round 3): while they walked the tiles' lists in rounds of 64 entries, threshold 3
 * gave 0.2676 ms per iteration against 0.2701 without the test (profiles/r03e: +6.9 us in the VALU-bound preprocess
 * kernel for -8.1 us in the tile kernels); with compact lists for the backward and a scanned list for the forward
 * the shorter lists are worth 1.6 us in the forward and nothing in the backward: 0.2321 ms with threshold 3 or 6,
 * 0.2286 ms without at 500 k surfels / 64 x 2048, no difference at 170 k / 50 k (tools/gpu_r03w.sh). */
#define SLS_TILE_CULL_MIN_DEFAULT 0
/* blending thresholds */
#define SLS_ALPHA_MAX 0.99f
#define SLS_ALPHA_MIN (1.0f / 255.0f)
#define SLS_T_MIN 1.0e-4f

/* Per-surfel record written by preprocess, read by the tile kernels:
 * 5 x float4 = 80 B.
 *   q0: Hu.xyz , npv      Hu =  sigma (Tv x p)/su   (u = Hu.(d-dc)/nd)
 *   q1: Hv.xyz , rho_c    Hv = -sigma (Tu x p)/sv   (v = Hv.(d-dc)/nd)
 *   q2: n.xyz  , opacity  n = sigma*Tn faces the sensor, npv = n.p <= 0
 *   q3: dc.xyz , kc       dc = p/|p|; kc >= sqrt(2 ln(255 o)): no pixel further than kc
 *                         sigma from the centre (on the surfel plane) reaches alpha >= 1/255
 *   q4: cpx, cpy, ex, ey  centre pixel; conservative support half-extent
 *                         (kc, ex, ey are a kernel-side culling aid only; they
 *                          never change a result and are not checked).
 */
#define SLS_REC_STRIDE 20
#define SLS_REC_HU 0
#define SLS_REC_NPV 3
#define SLS_REC_HV 4
#define SLS_REC_RHOC 7
#define SLS_REC_N 8
#define SLS_REC_OPAC 11
#define SLS_REC_DC 12
#define SLS_REC_KC 15
#define SLS_REC_CPX 16
#define SLS_REC_CPY 17
#define SLS_REC_EX 18
#define SLS_REC_EY 19

/* Per-surfel gradient record accumulated by the tile backward kernel and
 * consumed by preprocess-backward: 4 x float4 = 64 B.
 *   g0: dL/dHu.xyz , dL/dnpv
 *   g1: dL/dHv.xyz , dL/drho_c
 *   g2: dL/dn.xyz  , dL/dopacity
 *   g3: Su, Sv, dL/dcpx, dL/dcpy     (dL/ddc = -(Su*Hu + Sv*Hv))
 */
#define SLS_GREC_STRIDE 16

/* Per-pixel state saved by the forward for the backward (the caller may
 * overwrite allmap in place — gaussian_renderer/__init__.py:61-62,70-71 —
 * so the backward never reads allmap):  float4 {T_final, M1, M2, 0} and
 * uint2 {n_contrib, median_contrib}. */

/* allmap channels (gaussian_renderer/__init__.py:51-79) */
#define SLS_CH_DEPTH 0
#define SLS_CH_ALPHA 1
#define SLS_CH_NORMAL 2
#define SLS_CH_MEDIAN 5
#define SLS_CH_DIST 6
#define SLS_NUM_CH 7

#endif /* SLS_SPEC_H */
