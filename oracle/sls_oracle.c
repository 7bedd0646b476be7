/*
 * sls_oracle.c — CPU restatement of the spherical 2D-Gaussian-surfel
 * rasterizer hot path (preprocess -> tile keys -> stable sort -> tile
 * ranges -> per-tile front-to-back blend -> per-pixel backward ->
 * preprocess backward), plus Adam and brute-force 3-NN.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (splat_loam_amd/)
 * never imports, links or executes anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned".  The reference's rasterizer is the
 * un-vendored git submodule diff-surfel-spherical-rasterization 0.0.1
 * (/root/reference/.gitmodules:1-3, pixi.lock:1258-1261), its source is
 * not in /root/reference, there are no reference tests or golden vectors
 * for it.  What IS pinned by the reference tree and
 * is followed here:
 *   - camera conventions  scene/cameras.py:43-50  (viewmatrix = inv(T)^T,
 *     projmatrix[:3,:3] = K^T), utils/graphic_utils.py:26-66 (spherical
 *     ray model: az = K^-1 col, el = K^-1 row, ray = (cos az cos el,
 *     sin az cos el, sin el));
 *   - quaternion order (w,x,y,z) and R(q)  utils/general_utils.py:13-37;
 *   - surfel frame: tangents = R[:,0], R[:,1], normal = R[:,2]
 *     scene/gaussian_model.py:22-38;
 *   - output contract: allmap (7,H,W) = [sum w*depth, alpha, sum w*n (view
 *     frame) x3, median depth, distortion], depth == RANGE along the ray,
 *     radii > 0 <=> visible   gaussian_renderer/__init__.py:40-79;
 *   - Adam spec  scene/gaussian_model.py:97-121 (eps 1e-15, 4 groups);
 *   - distCUDA2 usage  slam/mapper.py:109-117.
 * Everything inside the rasterizer follows SURVEY.md §8a decisions D1-D9
 * as refined in DESIGN.md §2 (the written spec of this build).
 *
 * The file compiles twice: float (liboracle_f32.so, the parity checker for
 * the HIP kernels) and -DSLS_REAL_IS_DOUBLE (liboracle_f64.so, used to
 * verify the analytic backward against autograd / finite differences).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/sls_det_math.h"
#include "../include/sls_spec.h"

typedef sls_real real;
#ifdef SLS_REAL_IS_DOUBLE
#define R_EXP(x) exp(x)
#else
#define R_EXP(x) expf(x)
#endif
#define RC(x) ((real)(x))

/* icam: H, W, tile_w, tile_h, wrap
 * fcam: fx, fy, cx, cy, scale_modifier, near, far, Rvw[9] (row-major), tvw[3] */
enum { IC_H = 0, IC_W, IC_TW, IC_TH, IC_WRAP, IC_TILECULL, IC_COUNT };   /* IC_TILECULL: D10 threshold (tiles), 0 = off */
enum { FC_FX = 0, FC_FY, FC_CX, FC_CY, FC_MOD, FC_NEAR, FC_FAR, FC_R = 7, FC_T = 16, FC_COUNT = 19 };

int or_real_bytes(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------ */
/* Ray tables.  Pixel (c, r) has image coordinate (c, r) (D1); with a    */
/* pixel-centre offset (SlsCamera.pix_offset, oracle.py:Camera) the     */
/* caller passes the principal point (cx - ox, cy - oy) in FC_CX/FC_CY, */
/* which moves rays, centre pixel, rectangle and low-pass term alike:   */
/*   az = (c - cx)/fx, el = (r - cy)/fy                                */
/*   ray = (cos az cos el, sin az cos el, sin el)                      */
/* utils/graphic_utils.py:46-59 (without its -0.5 offset, see D1).     */
/* Tables are evaluated in double and rounded once.                    */
/* ------------------------------------------------------------------ */
void or_ray_tables(const int32_t *ic, const real *fc, real *col_cs, real *row_cs)
{
    const int H = ic[IC_H], W = ic[IC_W];
    for (int c = 0; c < W; ++c) {
        double a = ((double)c - (double)fc[FC_CX]) / (double)fc[FC_FX];
        col_cs[2 * c + 0] = (real)cos(a);
        col_cs[2 * c + 1] = (real)sin(a);
    }
    for (int r = 0; r < H; ++r) {
        double e = ((double)r - (double)fc[FC_CY]) / (double)fc[FC_FY];
        row_cs[2 * r + 0] = (real)cos(e);
        row_cs[2 * r + 1] = (real)sin(e);
    }
}

static inline real dot3(const real *a, const real *b)
{ /* fixed fma chain: ((a2*b2) + a1*b1) + a0*b0 */
    return SLS_FMA(a[0], b[0], SLS_FMA(a[1], b[1], a[2] * b[2]));
}
static inline void cross3(const real *a, const real *b, real *o)
{
    o[0] = SLS_FMA(a[1], b[2], -(a[2] * b[1]));
    o[1] = SLS_FMA(a[2], b[0], -(a[0] * b[2]));
    o[2] = SLS_FMA(a[0], b[1], -(a[1] * b[0]));
}
static inline void matvec(const real *M, const real *v, real *o)
{
    for (int i = 0; i < 3; ++i) o[i] = dot3(M + 3 * i, v);
}
static inline void matTvec(const real *M, const real *v, real *o)
{
    for (int i = 0; i < 3; ++i)
        o[i] = SLS_FMA(M[i], v[0], SLS_FMA(M[3 + i], v[1], M[6 + i] * v[2]));
}
static inline int floordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; }
static inline int posmod(int a, int b) { int m = a % b; return m < 0 ? m + b : m; }
static inline int to_int_clamped(real v)
{
    if (v > RC(1.0e9)) v = RC(1.0e9);
    if (v < RC(-1.0e9)) v = RC(-1.0e9);
    return (int)v;
}

/* quaternion (w,x,y,z) -> columns tu, tv, tn of R(q).  No re-normalisation
 * (the caller passes F.normalize'd rotations, scene/gaussian_model.py:63). */
static inline void quat_axes(const real *q, real *tu, real *tv, real *tn)
{
    const real r = q[0], x = q[1], y = q[2], z = q[3];
    const real two = RC(2.0), one = RC(1.0);
    tu[0] = one - two * SLS_FMA(y, y, z * z);
    tu[1] = two * SLS_FMA(x, y, r * z);
    tu[2] = two * SLS_FMA(x, z, -(r * y));
    tv[0] = two * SLS_FMA(x, y, -(r * z));
    tv[1] = one - two * SLS_FMA(x, x, z * z);
    tv[2] = two * SLS_FMA(y, z, r * x);
    tn[0] = two * SLS_FMA(x, z, r * y);
    tn[1] = two * SLS_FMA(y, z, -(r * x));
    tn[2] = one - two * SLS_FMA(x, x, y * y);
}

/* Angular half-extents of a ball of radius rad around a point at range rho,
 * horizontal range rxy: theta (elevation) and daz (azimuth). D4. */
static inline void ball_extent(real rad, real rho, real rxy, real *theta, real *daz)
{
    if (!(rad < rho)) { *theta = SLS_PI; *daz = SLS_PI; return; }
    *theta = sls_asin01(rad / rho);
    const real q = rad / rxy; /* rxy == 0 -> +inf */
    if (!(q < RC(1.0))) *daz = SLS_PI;
    else *daz = sls_asin01(q);
}

/* D10 — per-camera constants of the tile-level footprint test (include/sls_det_math.h), in double and rounded
 * once, exactly as the library's make_tile_cull_cam does (same expressions, same libm). */
static SlsTileCullCam tile_cull_consts(const int32_t *ic, const real *fc)
{
    SlsTileCullCam c;
    const double hx = 0.5 * (double)(ic[IC_TW] - 1), hy = 0.5 * (double)(ic[IC_TH] - 1);
    const double ax = 0.5 / (double)fc[FC_FX], ay = 0.5 / (double)fc[FC_FY];
    c.chx = (real)cos(ax); c.shx = (real)sin(ax);
    c.chy = (real)cos(ay); c.shy = (real)sin(ay);
    const double kx = hx / (double)fc[FC_FX], ky = hy / (double)fc[FC_FY];
    c.kx = (real)kx; c.ky = (real)ky;
    const double span = fabs(kx) + fabs(ky);
    c.eps = (real)(0.5 * span * span * 1.01 + 4.0e-6);
    c.hx = (real)hx; c.hy = (real)hy;
    c.wrapW = ic[IC_WRAP] ? (real)ic[IC_W] : RC(0.0);
    c.invW = ic[IC_WRAP] ? RC(1.0) / (real)ic[IC_W] : RC(0.0);
    return c;
}

/* ------------------------------------------------------------------ */
/* A1 preprocess (called inside                     */
/* gaussian_renderer/__init__.py:40-47).                               */
/* rect[4*i..] = {txlo, ncols, tylo, nrows} (tile units, txlo already  */
/* reduced modulo the tile-grid width in wrap mode).                   */
/* ------------------------------------------------------------------ */
void or_preprocess(const int32_t *ic, const real *fc, int N,
                   const real *means, const real *scales, const real *rots, const real *opac,
                   const real *col_cs, const real *row_cs,
                   real *rec, int32_t *radii, int32_t *rect, uint32_t *tiles, uint64_t *tmask, real *depth)
{
    const SlsTileCullCam tcc = tile_cull_consts(ic, fc);
    const int tile_cull = (col_cs && row_cs) ? ic[IC_TILECULL] : 0;   /* rectangles of tile_cull .. 64 tiles are tested */
    const int H = ic[IC_H], W = ic[IC_W], TW = ic[IC_TW], TH = ic[IC_TH], wrap = ic[IC_WRAP];
    const int GX = (W + TW - 1) / TW;
    const real fx = fc[FC_FX], fy = fc[FC_FY], cx = fc[FC_CX], cy = fc[FC_CY];
    const real mod = fc[FC_MOD], near_c = fc[FC_NEAR];
    const real *Rvw = fc + FC_R, *tvw = fc + FC_T;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        real *rc = rec + (size_t)i * SLS_REC_STRIDE;
        for (int k = 0; k < SLS_REC_STRIDE; ++k) rc[k] = 0;
        radii[i] = 0; tiles[i] = 0; depth[i] = 0; tmask[i] = 0;
        rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;

        const real *m = means + 3 * i;
        real p[3];
        for (int k = 0; k < 3; ++k)
            p[k] = SLS_FMA(Rvw[3 * k], m[0], SLS_FMA(Rvw[3 * k + 1], m[1], SLS_FMA(Rvw[3 * k + 2], m[2], tvw[k])));
        const real rxy2 = SLS_FMA(p[0], p[0], p[1] * p[1]);
        const real rho2 = SLS_FMA(p[2], p[2], rxy2);
        const real rho = SLS_SQRT(rho2), rxy = SLS_SQRT(rxy2);
        if (!(rho >= near_c) || !(rho < RC(1.0e18))) continue; /* D2 near cut, non-finite */

        const real az = sls_atan2(p[1], p[0]);
        const real el = sls_atan2(p[2], rxy);
        const real cpx = SLS_FMA(fx, az, cx), cpy = SLS_FMA(fy, el, cy);

        real tu[3], tv[3], tn[3], Tu[3], Tv[3], Tn[3];
        quat_axes(rots + 4 * i, tu, tv, tn);
        matvec(Rvw, tu, Tu); matvec(Rvw, tv, Tv); matvec(Rvw, tn, Tn);
        const real su = scales[2 * i] * mod, sv = scales[2 * i + 1] * mod;
        const real c = dot3(Tn, p);
        const real sig = (c > RC(0.0)) ? RC(-1.0) : RC(1.0);
        real n[3], A[3], B[3], Hu[3], Hv[3];
        for (int k = 0; k < 3; ++k) {
            n[k] = sig * Tn[k];
            A[k] = (sig * Tv[k]) / su;
            B[k] = (-sig * Tu[k]) / sv;
        }
        cross3(A, p, Hu); cross3(B, p, Hv);

        /* D4 extent */
        const real smax = SLS_FMAX(su, sv);
        real theta, daz;
        ball_extent(SLS_CUTOFF * smax, rho, rxy, &theta, &daz);
        const real rx = SLS_FMAX(SLS_FABS(fx) * daz, RC(SLS_RMIN_PX));
        const real ry = SLS_FMAX(SLS_FABS(fy) * theta, RC(SLS_RMIN_PX));
        int xlo = to_int_clamped(SLS_FLOOR(cpx - rx + RC(0.5)));
        int xhi = to_int_clamped(SLS_FLOOR(cpx + rx + RC(0.5)));
        int ylo = to_int_clamped(SLS_FLOOR(cpy - ry + RC(0.5)));
        int yhi = to_int_clamped(SLS_FLOOR(cpy + ry + RC(0.5)));
        if (ylo < 0) ylo = 0;
        if (yhi > H - 1) yhi = H - 1;
        if (ylo > yhi) continue;
        int txlo, ncols;
        if (wrap) {
            if ((int64_t)xhi - (int64_t)xlo + 1 >= (int64_t)W) { txlo = 0; ncols = GX; }
            else {
                const int a = floordiv(xlo, TW), b = floordiv(xhi, TW);
                ncols = b - a + 1; if (ncols > GX) ncols = GX;
                txlo = posmod(a, GX);
            }
        } else {
            if (xlo < 0) xlo = 0;
            if (xhi > W - 1) xhi = W - 1;
            if (xlo > xhi) continue;
            txlo = xlo / TW; ncols = xhi / TW - txlo + 1;
        }
        const int tylo = ylo / TH, nrows = yhi / TH - tylo + 1;

        rect[4 * i] = txlo; rect[4 * i + 1] = ncols; rect[4 * i + 2] = tylo; rect[4 * i + 3] = nrows;
        /* D10: which tiles of the rectangle (row-major, the emission order) the footprint can reach */
        const int nrect = ncols * nrows;
        uint64_t mask = nrect >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << nrect) - 1);
        if (tile_cull > 0 && nrect >= tile_cull && nrect <= 64) {
            SlsTileCullSurfel cs;
            const real dcv[3] = { p[0] / rho, p[1] / rho, p[2] / rho };
            sls_tile_cull_surfel(Tu, Tv, n, p, dcv, su, sv, opac[i], cpx, cpy, &cs);
            for (int idx = 0; idx < nrect; ++idx) {
                const int ky = idx / ncols, kx = idx - ky * ncols;
                int txx = txlo + kx;
                if (txx >= GX) txx -= GX;
                const int x0 = txx * TW, y0 = (tylo + ky) * TH;
                int ci = x0 + TW / 2 - 1, ri = y0 + TH / 2 - 1;
                if (ci > W - 1) ci = W - 1;
                if (ri > H - 1) ri = H - 1;
                if (sls_tile_outside(&tcc, &cs, (real)x0, (real)y0, col_cs[2 * ci], col_cs[2 * ci + 1],
                                     row_cs[2 * ri], row_cs[2 * ri + 1]))
                    mask &= ~((uint64_t)1 << idx);
            }
        }
        tmask[i] = mask;
        tiles[i] = nrect <= 64 ? (uint32_t)__builtin_popcountll(mask) : (uint32_t)nrect;
        radii[i] = to_int_clamped(SLS_CEIL(SLS_FMAX(rx, ry)));
        depth[i] = rho;

        for (int k = 0; k < 3; ++k) {
            rc[SLS_REC_HU + k] = Hu[k];
            rc[SLS_REC_HV + k] = Hv[k];
            rc[SLS_REC_N + k] = n[k];
            rc[SLS_REC_DC + k] = p[k] / rho;
        }
        rc[SLS_REC_NPV] = sig * c;
        rc[SLS_REC_RHOC] = rho;
        rc[SLS_REC_OPAC] = opac[i];
        rc[SLS_REC_CPX] = cpx;
        rc[SLS_REC_CPY] = cpy;
    }
}

/* ------------------------------------------------------------------ */
/* A2-A5: instance emission, stable sort by (tile, depth bits), ranges */
/* key = (tile_id << 32) | bits(float depth)  (D3, D9).  Each surfel    */
/* appears at most once per tile, so "stable" == tie-break on the      */
/* surfel index.                                                       */
/* ------------------------------------------------------------------ */
uint64_t or_count_instances(int N, const uint32_t *tiles)
{
    uint64_t r = 0;
    for (int i = 0; i < N; ++i) r += tiles[i];
    return r;
}

typedef struct { uint64_t k; uint32_t v; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    if (x->v != y->v) return x->v < y->v ? -1 : 1;
    return 0;
}

/* depth32: IEEE-754 binary32 bit patterns of the float depth (the f64
 * build rounds its depth to float first; the caller passes the bits). */
void or_emit_sort(const int32_t *ic, int N, const int32_t *rect, const uint32_t *tiles, const uint64_t *tmask,
                  const uint32_t *depth_bits, uint64_t R,
                  uint64_t *keys_unsorted, uint32_t *vals_unsorted,
                  uint64_t *keys, uint32_t *vals, uint32_t *ranges)
{
    const int H = ic[IC_H], W = ic[IC_W], TW = ic[IC_TW], TH = ic[IC_TH];
    const int GX = (W + TW - 1) / TW, GY = (H + TH - 1) / TH;
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (R ? R : 1));
    uint64_t off = 0;
    for (int i = 0; i < N; ++i) {
        if (!tiles[i]) continue;
        const int txlo = rect[4 * i], ncols = rect[4 * i + 1], tylo = rect[4 * i + 2], nrows = rect[4 * i + 3];
        int bit = 0;
        for (int y = 0; y < nrows; ++y)
            for (int k = 0; k < ncols; ++k, ++bit) {
                if (bit < 64 && !((tmask[i] >> bit) & 1)) continue;     /* D10: the footprint cannot reach this tile */
                const int tx = (txlo + k) % GX;
                const uint64_t tile = (uint64_t)(tylo + y) * GX + tx;
                kv[off].k = (tile << 32) | depth_bits[i];
                kv[off].v = (uint32_t)i;
                ++off;
            }
    }
    for (uint64_t j = 0; j < R; ++j) { keys_unsorted[j] = kv[j].k; vals_unsorted[j] = kv[j].v; }
    qsort(kv, R, sizeof(kv_t), kv_cmp);
    for (uint64_t j = 0; j < R; ++j) { keys[j] = kv[j].k; vals[j] = kv[j].v; }
    free(kv);
    const int T = GX * GY;
    for (int t = 0; t < 2 * T; ++t) ranges[t] = 0;
    for (uint64_t j = 0; j < R; ++j) {
        const uint32_t t = (uint32_t)(keys[j] >> 32);
        if (j == 0 || (uint32_t)(keys[j - 1] >> 32) != t) ranges[2 * t] = (uint32_t)j;
        if (j + 1 == R || (uint32_t)(keys[j + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(j + 1);
    }
}

/* ------------------------------------------------------------------ */
/* Shared per-(pixel, surfel) evaluation.                              */
/* ------------------------------------------------------------------ */
typedef struct {
    real delta[3], nd, rinv, hu, hv, u, v, t, dx, dy, rho3, rho2, rho, depth, G, og, alpha;
    int valid3d, use3d, skip;
} eval_t;

static inline void eval_surfel(const real *rc, const real *d, real pc, real pr, int wrap, real Wf,
                               real near_c, eval_t *e)
{
    for (int k = 0; k < 3; ++k) e->delta[k] = d[k] - rc[SLS_REC_DC + k];
    const real *n = rc + SLS_REC_N;
    e->nd = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    e->valid3d = e->nd < RC(0.0);
    e->rinv = RC(1.0) / e->nd;
    e->hu = rc[SLS_REC_HU] * e->delta[0] + rc[SLS_REC_HU + 1] * e->delta[1] + rc[SLS_REC_HU + 2] * e->delta[2];
    e->hv = rc[SLS_REC_HV] * e->delta[0] + rc[SLS_REC_HV + 1] * e->delta[1] + rc[SLS_REC_HV + 2] * e->delta[2];
    e->u = e->hu * e->rinv; e->v = e->hv * e->rinv;
    e->t = rc[SLS_REC_NPV] * e->rinv;
    e->rho3 = e->u * e->u + e->v * e->v;
    real dx = pc - rc[SLS_REC_CPX];
    if (wrap) { /* D5 wrapped azimuth difference */
        if (dx > RC(0.5) * Wf) dx -= Wf;
        else if (dx < RC(-0.5) * Wf) dx += Wf;
    }
    e->dx = dx; e->dy = pr - rc[SLS_REC_CPY];
    e->rho2 = RC(SLS_FILTER_INV_SQUARE) * (e->dx * e->dx + e->dy * e->dy);
    e->use3d = e->valid3d && (e->rho3 <= e->rho2);
    e->rho = e->use3d ? e->rho3 : e->rho2;
    e->depth = e->use3d ? e->t : rc[SLS_REC_RHOC]; /* D6 */
    e->skip = 1;
    e->G = 0; e->og = 0; e->alpha = 0;
    if (e->depth < near_c) return;
    e->G = R_EXP(RC(-0.5) * e->rho);
    e->og = rc[SLS_REC_OPAC] * e->G;
    e->alpha = SLS_FMIN(RC(SLS_ALPHA_MAX), e->og);
    if (e->alpha < RC(SLS_ALPHA_MIN)) return;
    e->skip = 0;
}

static inline int near_rel(real a, real b, real tol) { return SLS_FABS(a - b) <= tol * SLS_FMAX(SLS_FABS(a), SLS_FABS(b)); }

/* ------------------------------------------------------------------ */
/* A6 per-tile forward render.                                         */
/* fragile[p] != 0: some discrete decision at pixel p sat within       */
/* `frag_tol` (relative) of its threshold; a bit-different exp/rcp may  */
/* legitimately decide the other way there.                            */
/* ------------------------------------------------------------------ */
void or_render_fwd(const int32_t *ic, const real *fc, const real *col_cs, const real *row_cs,
                   const uint32_t *ranges, const uint32_t *vals, const real *rec,
                   real *allmap, real *pixT, uint32_t *pixN, uint32_t *pixMed, real *pixM1, real *pixM2,
                   uint8_t *fragile, uint32_t *tile_consumed, double frag_tol)
{
    const int H = ic[IC_H], W = ic[IC_W], TW = ic[IC_TW], TH = ic[IC_TH], wrap = ic[IC_WRAP];
    const int GX = (W + TW - 1) / TW, GY = (H + TH - 1) / TH;
    const real near_c = fc[FC_NEAR], far_c = fc[FC_FAR];
    const real mscale = far_c / (far_c - near_c);
    const real ftol = (real)frag_tol;
    const size_t P = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < GX * GY; ++tile) {
        const int ty = tile / GX, tx = tile % GX;
        const uint32_t s = ranges[2 * tile], en = ranges[2 * tile + 1];
        uint32_t tile_max = 0;
        for (int py = ty * TH; py < (ty + 1) * TH && py < H; ++py)
            for (int px = tx * TW; px < (tx + 1) * TW && px < W; ++px) {
                const real d[3] = { col_cs[2 * px] * row_cs[2 * py], col_cs[2 * px + 1] * row_cs[2 * py], row_cs[2 * py + 1] };
                real T = 1, D = 0, Nn[3] = { 0, 0, 0 }, M1 = 0, M2 = 0, dist = 0, med = 0;
                uint32_t medc = 0, last = 0, contributor = 0, consumed = en - s;
                uint8_t frag = 0;
                for (uint32_t j = s; j < en; ++j) {
                    ++contributor;
                    const real *rc = rec + (size_t)vals[j] * SLS_REC_STRIDE;
                    eval_t e;
                    eval_surfel(rc, d, (real)px, (real)py, wrap, (real)W, near_c, &e);
                    if (ftol > 0) {
                        if (SLS_FABS(e.nd) <= RC(1e-6)) frag = 1;
                        if (e.valid3d && near_rel(e.rho3, e.rho2, ftol)) frag = 1;
                        if (near_rel(e.depth, near_c, ftol)) frag = 1;
                        if (!(e.depth < near_c)) {
                            if (near_rel(e.og, RC(SLS_ALPHA_MIN), ftol)) frag = 1;
                            if (near_rel(e.og, RC(SLS_ALPHA_MAX), ftol)) frag = 1;
                        }
                    }
                    if (e.skip) continue;
                    const real testT = T * (RC(1.0) - e.alpha);
                    if (ftol > 0 && near_rel(testT, RC(SLS_T_MIN), ftol)) frag = 1;
                    if (testT < RC(SLS_T_MIN)) { consumed = contributor; break; }
                    const real w = e.alpha * T;
                    const real A = RC(1.0) - T;
                    const real m = mscale * (RC(1.0) - near_c / e.depth);
                    dist += (m * m * A + M2 - RC(2.0) * m * M1) * w;
                    D += e.depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                    if (ftol > 0 && near_rel(T, RC(0.5), ftol)) frag = 1;
                    if (T > RC(0.5)) { med = e.depth; medc = contributor; }
                    for (int k = 0; k < 3; ++k) Nn[k] += rc[SLS_REC_N + k] * w;
                    T = testT;
                    last = contributor;
                }
                const size_t pix = (size_t)py * W + px;
                allmap[SLS_CH_DEPTH * P + pix] = D;
                allmap[SLS_CH_ALPHA * P + pix] = RC(1.0) - T;
                for (int k = 0; k < 3; ++k) allmap[(SLS_CH_NORMAL + k) * P + pix] = Nn[k];
                allmap[SLS_CH_MEDIAN * P + pix] = med;
                allmap[SLS_CH_DIST * P + pix] = dist;
                pixT[pix] = T; pixN[pix] = last; pixMed[pix] = medc; pixM1[pix] = M1; pixM2[pix] = M2;
                if (fragile) fragile[pix] = frag;
                if (consumed > tile_max) tile_max = consumed;
            }
        if (tile_consumed) tile_consumed[tile] = tile_max;
    }
}

/* ------------------------------------------------------------------ */
/* A7 per-tile backward render.  grec: N x 16 (layout sls_spec.h),     */
/* gabs: same shape, sum of |terms| (tolerance scale for a checker     */
/* whose accumulation order differs), may be NULL.                     */
/* Sequential over tiles => deterministic accumulation order.          */
/* ------------------------------------------------------------------ */
static inline void acc(real *g, real *ga, int k, real v)
{
    g[k] += v;
    if (ga) ga[k] += SLS_FABS(v);
}

void or_render_bwd(const int32_t *ic, const real *fc, const real *col_cs, const real *row_cs,
                   const uint32_t *ranges, const uint32_t *vals, const real *rec,
                   const real *pixT, const uint32_t *pixN, const uint32_t *pixMed,
                   const real *pixM1, const real *pixM2, const real *dL_dallmap,
                   int N, real *grec, real *gabs, int threads)
{
    const int H = ic[IC_H], W = ic[IC_W], TW = ic[IC_TW], TH = ic[IC_TH], wrap = ic[IC_WRAP];
    const int GX = (W + TW - 1) / TW, GY = (H + TH - 1) / TH;
    const real near_c = fc[FC_NEAR], far_c = fc[FC_FAR];
    const real mscale = far_c / (far_c - near_c);
    const size_t P = (size_t)H * W;
    memset(grec, 0, sizeof(real) * (size_t)N * SLS_GREC_STRIDE);
    if (gabs) memset(gabs, 0, sizeof(real) * (size_t)N * SLS_GREC_STRIDE);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) if (threads > 1)
    for (int tile = 0; tile < GX * GY; ++tile) {
        const int ty = tile / GX, tx = tile % GX;
        const uint32_t s = ranges[2 * tile];
        for (int py = ty * TH; py < (ty + 1) * TH && py < H; ++py)
            for (int px = tx * TW; px < (tx + 1) * TW && px < W; ++px) {
                const size_t pix = (size_t)py * W + px;
                const uint32_t last = pixN[pix];
                if (!last) continue;
                const real d[3] = { col_cs[2 * px] * row_cs[2 * py], col_cs[2 * px + 1] * row_cs[2 * py], row_cs[2 * py + 1] };
                const real dD = dL_dallmap[SLS_CH_DEPTH * P + pix];
                const real dA = dL_dallmap[SLS_CH_ALPHA * P + pix];
                const real dN[3] = { dL_dallmap[(SLS_CH_NORMAL)*P + pix], dL_dallmap[(SLS_CH_NORMAL + 1) * P + pix],
                                     dL_dallmap[(SLS_CH_NORMAL + 2) * P + pix] };
                const real dMed = dL_dallmap[SLS_CH_MEDIAN * P + pix];
                const real dDist = dL_dallmap[SLS_CH_DIST * P + pix];
                const real Tf = pixT[pix], Af = RC(1.0) - Tf, M1 = pixM1[pix], M2 = pixM2[pix];
                const uint32_t medc = pixMed[pix];
                real T = Tf, S = 0;
                for (uint32_t c = last; c >= 1; --c) {
                    const uint32_t j = s + c - 1;
                    const uint32_t gi = vals[j];
                    const real *rc = rec + (size_t)gi * SLS_REC_STRIDE;
                    eval_t e;
                    eval_surfel(rc, d, (real)px, (real)py, wrap, (real)W, near_c, &e);
                    if (e.skip) continue;
                    const real om = RC(1.0) - e.alpha;
                    T = T / om;
                    const real w = e.alpha * T;
                    const real m = mscale * (RC(1.0) - near_c / e.depth);
                    const real dm_dd = mscale * near_c / (e.depth * e.depth);
                    const real *n = rc + SLS_REC_N;
                    const real gk = dD * e.depth + (dN[0] * n[0] + dN[1] * n[1] + dN[2] * n[2]) + dA +
                                    dDist * (M2 + m * m * Af - RC(2.0) * m * M1);
                    const real dL_dalpha = T * gk - S / om;
                    S += w * gk;
                    real dL_ddepth = w * dD + dDist * RC(2.0) * w * (m * Af - M1) * dm_dd;
                    if (c == medc) dL_ddepth += dMed;
                    real dL_do = 0, dL_dG = 0;
                    if (e.og < RC(SLS_ALPHA_MAX)) { dL_do = dL_dalpha * e.G; dL_dG = dL_dalpha * rc[SLS_REC_OPAC]; }
                    const real dL_drho = RC(-0.5) * e.G * dL_dG;
                    real *g, *ga;
                    real gl[SLS_GREC_STRIDE];
                    for (int k = 0; k < SLS_GREC_STRIDE; ++k) gl[k] = 0;
                    for (int k = 0; k < 3; ++k) gl[8 + k] = w * dN[k];
                    gl[11] = dL_do;
                    if (e.use3d) {
                        const real dL_du = dL_drho * RC(2.0) * e.u, dL_dv = dL_drho * RC(2.0) * e.v;
                        const real dL_dt = dL_ddepth;
                        const real dL_dhu = dL_du * e.rinv, dL_dhv = dL_dv * e.rinv;
                        const real dL_drinv = dL_du * e.hu + dL_dv * e.hv + dL_dt * rc[SLS_REC_NPV];
                        const real dL_dnd = -dL_drinv * e.rinv * e.rinv;
                        for (int k = 0; k < 3; ++k) {
                            gl[0 + k] = dL_dhu * e.delta[k];
                            gl[4 + k] = dL_dhv * e.delta[k];
                            gl[8 + k] += dL_dnd * d[k];
                        }
                        gl[3] = dL_dt * e.rinv;
                        gl[12] = dL_dhu;
                        gl[13] = dL_dhv;
                    } else {
                        gl[7] = dL_ddepth;
                        gl[14] = dL_drho * RC(SLS_FILTER_INV_SQUARE) * RC(2.0) * e.dx * RC(-1.0);
                        gl[15] = dL_drho * RC(SLS_FILTER_INV_SQUARE) * RC(2.0) * e.dy * RC(-1.0);
                    }
                    g = grec + (size_t)gi * SLS_GREC_STRIDE;
                    ga = gabs ? gabs + (size_t)gi * SLS_GREC_STRIDE : NULL;
                    if (threads > 1) {
                        for (int k = 0; k < SLS_GREC_STRIDE; ++k) {
                            if (gl[k] == 0) continue;
#pragma omp atomic
                            g[k] += gl[k];
                            if (ga) {
                                const real av = SLS_FABS(gl[k]);
#pragma omp atomic
                                ga[k] += av;
                            }
                        }
                    } else {
                        for (int k = 0; k < SLS_GREC_STRIDE; ++k) acc(g, ga, k, gl[k]);
                    }
                }
            }
    }
}

/* ------------------------------------------------------------------ */
/* A8 preprocess backward: gradient record -> dL/d(means3D, scales,    */
/* rotations, opacities).  R(q) is differentiated as the polynomial it */
/* is (no re-normalisation).                                           */
/* ------------------------------------------------------------------ */
void or_preprocess_bwd(const int32_t *ic, const real *fc, int N,
                       const real *means, const real *scales, const real *rots,
                       const int32_t *radii, const real *grec,
                       real *dmeans, real *dscales, real *drots, real *dopac)
{
    (void)ic;
    const real fx = fc[FC_FX], fy = fc[FC_FY], mod = fc[FC_MOD];
    const real *Rvw = fc + FC_R, *tvw = fc + FC_T;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) dmeans[3 * i + k] = 0;
        dscales[2 * i] = dscales[2 * i + 1] = 0;
        for (int k = 0; k < 4; ++k) drots[4 * i + k] = 0;
        dopac[i] = 0;
        if (radii[i] <= 0) continue;
        const real *g = grec + (size_t)i * SLS_GREC_STRIDE;
        const real *m = means + 3 * i;
        real p[3];
        for (int k = 0; k < 3; ++k)
            p[k] = SLS_FMA(Rvw[3 * k], m[0], SLS_FMA(Rvw[3 * k + 1], m[1], SLS_FMA(Rvw[3 * k + 2], m[2], tvw[k])));
        const real rxy2 = SLS_FMA(p[0], p[0], p[1] * p[1]);
        const real rho2 = SLS_FMA(p[2], p[2], rxy2);
        const real rho = SLS_SQRT(rho2), rxy = SLS_SQRT(rxy2);
        real tu[3], tv[3], tn[3], Tu[3], Tv[3], Tn[3];
        const real *q = rots + 4 * i;
        quat_axes(q, tu, tv, tn);
        matvec(Rvw, tu, Tu); matvec(Rvw, tv, Tv); matvec(Rvw, tn, Tn);
        const real su = scales[2 * i] * mod, sv = scales[2 * i + 1] * mod;
        const real c = dot3(Tn, p);
        const real sig = (c > RC(0.0)) ? RC(-1.0) : RC(1.0);
        real n[3], A[3], B[3], Hu[3], Hv[3], dc[3];
        for (int k = 0; k < 3; ++k) {
            n[k] = sig * Tn[k];
            A[k] = (sig * Tv[k]) / su;
            B[k] = (-sig * Tu[k]) / sv;
            dc[k] = p[k] / rho;
        }
        cross3(A, p, Hu); cross3(B, p, Hv);

        const real *gHu = g + 0, *gHv = g + 4, *gn = g + 8;
        const real gnpv = g[3], grhoc = g[7], go = g[11], Su = g[12], Sv = g[13], gcpx = g[14], gcpy = g[15];
        real dp[3] = { 0, 0, 0 }, dA[3], dB[3], t1[3], t2[3];
        /* Hu = A x p, Hv = B x p */
        cross3(p, gHu, dA); cross3(gHu, A, t1);
        cross3(p, gHv, dB); cross3(gHv, B, t2);
        for (int k = 0; k < 3; ++k) dp[k] += t1[k] + t2[k];
        real dTu[3], dTv[3], dTn[3];
        for (int k = 0; k < 3; ++k) {
            dTv[k] = sig * dA[k] / su;
            dTu[k] = -sig * dB[k] / sv;
        }
        const real dsu = -dot3(dA, A) / su, dsv = -dot3(dB, B) / sv;
        /* n = sig Tn, npv = n.p */
        for (int k = 0; k < 3; ++k) {
            dTn[k] = sig * (gn[k] + gnpv * p[k]);
            dp[k] += gnpv * n[k];
        }
        /* rho_c = |p| */
        for (int k = 0; k < 3; ++k) dp[k] += grhoc * dc[k];
        /* dc = p/rho, dL/ddc = -(Su Hu + Sv Hv) */
        real gdc[3];
        for (int k = 0; k < 3; ++k) gdc[k] = -(Su * Hu[k] + Sv * Hv[k]);
        const real gd = dot3(gdc, dc);
        for (int k = 0; k < 3; ++k) dp[k] += (gdc[k] - gd * dc[k]) / rho;
        /* cpx = fx*atan2(y,x)+cx ; cpy = fy*atan2(z,rxy)+cy */
        if (rxy2 > RC(1e-30)) {
            const real gaz = gcpx * fx, gel = gcpy * fy;
            dp[0] += gaz * (-p[1] / rxy2) + gel * (-p[2] * p[0] / (rxy * rho2));
            dp[1] += gaz * (p[0] / rxy2) + gel * (-p[2] * p[1] / (rxy * rho2));
            dp[2] += gel * (rxy / rho2);
        }
        matTvec(Rvw, dp, dmeans + 3 * i);
        dscales[2 * i] = mod * dsu;
        dscales[2 * i + 1] = mod * dsv;
        dopac[i] = go;
        real G0[3], G1[3], G2[3]; /* dL/dtu, dL/dtv, dL/dtn (world) */
        matTvec(Rvw, dTu, G0); matTvec(Rvw, dTv, G1); matTvec(Rvw, dTn, G2);
        const real r = q[0], x = q[1], y = q[2], z = q[3];
        /* G[i][j] = dL/dR_ij with column j in {tu,tv,tn} */
#define GG(i, j) ((j) == 0 ? G0[i] : ((j) == 1 ? G1[i] : G2[i]))
        drots[4 * i + 0] = RC(2.0) * (-z * GG(0, 1) + y * GG(0, 2) + z * GG(1, 0) - x * GG(1, 2) - y * GG(2, 0) + x * GG(2, 1));
        drots[4 * i + 1] = RC(2.0) * (y * GG(0, 1) + z * GG(0, 2) + y * GG(1, 0) - RC(2.0) * x * GG(1, 1) - r * GG(1, 2) + z * GG(2, 0) + r * GG(2, 1) - RC(2.0) * x * GG(2, 2));
        drots[4 * i + 2] = RC(2.0) * (-RC(2.0) * y * GG(0, 0) + x * GG(0, 1) + r * GG(0, 2) + x * GG(1, 0) + z * GG(1, 2) - r * GG(2, 0) + z * GG(2, 1) - RC(2.0) * y * GG(2, 2));
        drots[4 * i + 3] = RC(2.0) * (-RC(2.0) * z * GG(0, 0) - r * GG(0, 1) + x * GG(0, 2) + r * GG(1, 0) - RC(2.0) * z * GG(1, 1) + y * GG(1, 2) + x * GG(2, 0) + y * GG(2, 1));
#undef GG
    }
}

/* ------------------------------------------------------------------ */
/* P6 Adam, one parameter tensor, torch.optim.Adam single-tensor op    */
/* order (scene/gaussian_model.py:121: lr per group, eps 1e-15,        */
/* betas (0.9, 0.999), no weight decay, no amsgrad).  step is 1-based. */
/* ------------------------------------------------------------------ */
void or_adam(int64_t n, real *p, const real *g, real *m, real *v,
             double lr, double b1, double b2, double eps, int64_t step)
{
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const real step_size = (real)(lr / bc1);
    const real bc2_sqrt = (real)sqrt(bc2);
    const real w1 = (real)(1.0 - b1), fb2 = (real)b2, w2 = (real)(1.0 - b2), feps = (real)eps;
    for (int64_t i = 0; i < n; ++i) {
        const real gi = g[i];
        m[i] = m[i] + (gi - m[i]) * w1; /* lerp_ */
        v[i] = v[i] * fb2 + gi * gi * w2; /* mul_().addcmul_() */
        const real denom = SLS_SQRT(v[i]) / bc2_sqrt + feps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}

/* ------------------------------------------------------------------ */
/* K1 distCUDA2: mean squared distance to the 3 nearest OTHER points   */
/* (slam/mapper.py:113-115, scene/gaussian_model.py:77-81).  Brute     */
/* force.  Fewer than 4 points: missing neighbours count as FLT_MAX    */
/* (lineage behaviour of the best-3 list initialised to FLT_MAX).      */
/* ------------------------------------------------------------------ */
void or_knn_dist2(int M, const float *pts, float *out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M; ++i) {
        float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        for (int j = 0; j < M; ++j) {
            if (j == i) continue;
            const float dx = pts[3 * j] - x, dy = pts[3 * j + 1] - y, dz = pts[3 * j + 2] - z;
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d2 < b2) {
                if (d2 < b1) {
                    b2 = b1;
                    if (d2 < b0) { b1 = b0; b0 = d2; } else b1 = d2;
                } else b2 = d2;
            }
        }
        out[i] = (b0 + b1 + b2) / 3.0f;
    }
}

/* test hook: the shared polynomial atan2 (include/sls_det_math.h) */
void or_atan2(int n, const real *y, const real *x, real *out)
{
    for (int i = 0; i < n; ++i) out[i] = sls_atan2(y[i], x[i]);
}

int or_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void or_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
