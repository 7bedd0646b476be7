// sls_render.hip — per-tile front-to-back blend (A6) and its per-pixel
// backward (A7).  SURVEY.md §8a rows A6/A7; maths in DESIGN.md §2.
//
// Mapping to CDNA4:
//   * one workgroup = one TILE_W x TILE_H tile = 4 waves; each wave owns an 8x8
//     pixel sub-tile and keeps its own "all 64 pixels saturated" early-out;
//   * the tile's depth-sorted surfel list is staged through LDS in batches of
//     256 records (80 B each, 5 x float4).  Staging is lane-linear: thread k
//     of round i fetches float4 #(k mod 5) of record #(k div 5), so adjacent
//     lanes read adjacent 16-B pieces of one record and the LDS image is
//     written with stride-1 ds_write_b128.  The next batch is prefetched into
//     registers while the current one is consumed;
//   * wave64 ballot culling: lane j tests record j of the batch against the
//     wave's sub-tile (conservative support box written by preprocess); the
//     64-bit ballot is then walked with scalar ff1 — surfels that cannot touch
//     the sub-tile cost no VALU work at all, the survivors are evaluated with
//     wave-uniform (broadcast) LDS reads;
//   * backward: per-surfel gradient = sum over the wave's pixels by 6 DPP adds
//     (row_ror / row_bcast, no LDS traffic), cross-wave accumulation with LDS
//     float atomics, one global float atomic per touched (tile, surfel, field).
// No MFMA: there is no dense contraction here (BASELINE.json north_star).
#include "sls_common.hpp"

namespace sls {

constexpr int kThreads = kTilePix;          // one thread per tile pixel
constexpr int kBatch = kThreads;            // records staged per batch (one index per thread)
constexpr int kSubX = kTileW / 8;           // wave sub-tiles per tile row

struct Eval {
    float dl0, dl1, dl2, rinv, hu, hv, u, v, t, dx, dy, depth, G, og, alpha;
    bool use3d, skip;
};

// One (pixel, surfel) evaluation; identical in forward and backward.
__device__ __forceinline__ void eval_surfel(const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                            const float4 q4, float d0, float d1, float d2, float pc, float pr,
                                            int wrap, float Wf, float near_c, Eval &e)
{
    e.dl0 = d0 - q3.x; e.dl1 = d1 - q3.y; e.dl2 = d2 - q3.z;
    const float nd = q2.x * d0 + q2.y * d1 + q2.z * d2;
    const bool valid3d = nd < 0.0f;
    e.rinv = __builtin_amdgcn_rcpf(nd);
    e.hu = q0.x * e.dl0 + q0.y * e.dl1 + q0.z * e.dl2;
    e.hv = q1.x * e.dl0 + q1.y * e.dl1 + q1.z * e.dl2;
    e.u = e.hu * e.rinv;
    e.v = e.hv * e.rinv;
    e.t = q0.w * e.rinv;
    const float rho3 = e.u * e.u + e.v * e.v;
    float dx = pc - q4.x;
    if (wrap) {
        if (dx > 0.5f * Wf) dx -= Wf;
        else if (dx < -0.5f * Wf) dx += Wf;
    }
    e.dx = dx;
    e.dy = pr - q4.y;
    const float rho2 = SLS_FILTER_INV_SQUARE * (e.dx * e.dx + e.dy * e.dy);
    e.use3d = valid3d && (rho3 <= rho2);
    const float rho = e.use3d ? rho3 : rho2;
    e.depth = e.use3d ? e.t : q1.w;
    e.G = __expf(-0.5f * rho);
    e.og = q2.w * e.G;
    e.alpha = fminf(SLS_ALPHA_MAX, e.og);
    e.skip = (e.depth < near_c) || (e.alpha < SLS_ALPHA_MIN);
}

__device__ __forceinline__ bool cull_pass(const float4 q4, float wcx, float wcy, int wrap, float Wf)
{
    float dxc = wcx - q4.x;
    if (wrap) {
        if (dxc > 0.5f * Wf) dxc -= Wf;
        else if (dxc < -0.5f * Wf) dxc += Wf;
    }
    return (fabsf(dxc) <= q4.z + 3.5f) && (fabsf(wcy - q4.y) <= q4.w + 3.5f);
}

// ---------------------------------------------------------------------------
// A6 forward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void render_fwd_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    float *__restrict__ allmap, float4 *__restrict__ pix_state, uint2 *__restrict__ pix_contrib,
    uint32_t *__restrict__ tile_consumed)
{
    __shared__ float4 s_rec[kBatch * kRec4];
    __shared__ uint32_t s_consumed;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = cam.GX * cam.GY;
    const int tile = xcd_remap(blockIdx.x, T);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int sub_x = wave % kSubX, sub_y = wave / kSubX;
    const int x0 = tx * kTileW + sub_x * 8, y0 = ty * kTileH + sub_y * 8;
    const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wcx = (float)x0 + 3.5f, wcy = (float)y0 + 3.5f;
    const float Wf = (float)cam.W;

    float d0 = 1.0f, d1 = 0.0f, d2 = 0.0f;
    if (inside) {
        const float2 c = col_cs[px], r = row_cs[py];
        d0 = c.x * r.x; d1 = c.y * r.x; d2 = r.y;
    }
    const float pc = (float)px, pr = (float)py;
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);

    float Tr = 1.0f, D = 0.0f, N0 = 0.0f, N1 = 0.0f, N2 = 0.0f, M1 = 0.0f, M2 = 0.0f, dist = 0.0f, med = 0.0f;
    uint32_t medc = 0, last = 0, consumed = inside ? (uint32_t)n : 0u;
    bool done = !inside;
    bool wave_done = __all(done);
    if (tid == 0) s_consumed = 0;

    const int nb = (n + kBatch - 1) / kBatch;
    float4 pre[kRec4];
    auto prefetch = [&](int b) {
#pragma unroll
        for (int i = 0; i < kRec4; ++i) {
            const int k = i * kThreads + tid;
            const int j = k / kRec4, q = k - j * kRec4;
            const int gj = b * kBatch + j;
            if (gj < n) {
                const uint32_t idx = vals[range.x + gj];
                pre[i] = rec[(size_t)idx * kRec4 + q];
            } else pre[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    if (nb > 0) prefetch(0);

    for (int b = 0; b < nb; ++b) {
        if (__syncthreads_and(wave_done ? 1 : 0)) break;
#pragma unroll
        for (int i = 0; i < kRec4; ++i) s_rec[i * kThreads + tid] = pre[i];
        __syncthreads();
        if (b + 1 < nb) prefetch(b + 1);
        const int cnt = min(kBatch, n - b * kBatch);
        if (!wave_done) {
            for (int r = 0; r < kBatch / 64 && !wave_done; ++r) {
                const int jl = r * 64 + lane;
                bool pass = false;
                if (jl < cnt) pass = cull_pass(s_rec[jl * kRec4 + 4], wcx, wcy, cam.wrap, Wf);
                uint64_t mask = __ballot(pass);
                while (mask) {
                    const int jj = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int j = r * 64 + jj;
                    const uint32_t contributor = (uint32_t)(b * kBatch + j + 1);
                    const float4 q0 = s_rec[j * kRec4 + 0], q1 = s_rec[j * kRec4 + 1], q2 = s_rec[j * kRec4 + 2];
                    const float4 q3 = s_rec[j * kRec4 + 3], q4 = s_rec[j * kRec4 + 4];
                    Eval e;
                    eval_surfel(q0, q1, q2, q3, q4, d0, d1, d2, pc, pr, cam.wrap, Wf, cam.near_c, e);
                    if (!done && !e.skip) {
                        const float testT = Tr * (1.0f - e.alpha);
                        if (testT < SLS_T_MIN) {
                            done = true;
                            consumed = contributor;
                        } else {
                            const float w = e.alpha * Tr;
                            const float A = 1.0f - Tr;
                            const float m = mscale * (1.0f - cam.near_c * __builtin_amdgcn_rcpf(e.depth));
                            dist += (m * m * A + M2 - 2.0f * m * M1) * w;
                            D += e.depth * w;
                            M1 += m * w;
                            M2 += m * m * w;
                            if (Tr > 0.5f) { med = e.depth; medc = contributor; }
                            N0 += q2.x * w; N1 += q2.y * w; N2 += q2.z * w;
                            Tr = testT;
                            last = contributor;
                        }
                    }
                    if (__all(done)) { wave_done = true; break; }
                }
            }
        }
    }

    if (inside) {
        const size_t P = (size_t)cam.H * cam.W;
        const size_t pix = (size_t)py * cam.W + px;
        allmap[SLS_CH_DEPTH * P + pix] = D;
        allmap[SLS_CH_ALPHA * P + pix] = 1.0f - Tr;
        allmap[(SLS_CH_NORMAL + 0) * P + pix] = N0;
        allmap[(SLS_CH_NORMAL + 1) * P + pix] = N1;
        allmap[(SLS_CH_NORMAL + 2) * P + pix] = N2;
        allmap[SLS_CH_MEDIAN * P + pix] = med;
        allmap[SLS_CH_DIST * P + pix] = dist;
        pix_state[pix] = make_float4(Tr, M1, M2, 0.0f);
        pix_contrib[pix] = make_uint2(last, medc);
    }
    if (tile_consumed) {
        // wave max -> LDS max -> one store per tile
        uint32_t c = consumed;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c = max(c, (uint32_t)__shfl_down(c, off, 64));
        __syncthreads();
        if (lane == 0) atomicMax(&s_consumed, c);
        __syncthreads();
        if (tid == 0) tile_consumed[tile] = s_consumed;
    }
}

// ---------------------------------------------------------------------------
// A7 backward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void render_bwd_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    const float4 *__restrict__ pix_state, const uint2 *__restrict__ pix_contrib,
    const float *__restrict__ dL_dallmap, float *__restrict__ grec)
{
    __shared__ float4 s_rec[kBatch * kRec4];
    __shared__ float s_grad[kBatch * kGrec];
    __shared__ uint32_t s_max;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = cam.GX * cam.GY;
    const int tile = xcd_remap(blockIdx.x, T);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int sub_x = wave % kSubX, sub_y = wave / kSubX;
    const int x0 = tx * kTileW + sub_x * 8, y0 = ty * kTileH + sub_y * 8;
    const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wcx = (float)x0 + 3.5f, wcy = (float)y0 + 3.5f;
    const float Wf = (float)cam.W;
    const float pc = (float)px, pr = (float)py;
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);

    float d0 = 1.0f, d1 = 0.0f, d2 = 0.0f;
    uint32_t last = 0, medc = 0;
    float Tf = 1.0f, M1 = 0.0f, M2 = 0.0f;
    float dD = 0, dA = 0, dN0 = 0, dN1 = 0, dN2 = 0, dMed = 0, dDist = 0;
    if (inside) {
        const float2 c = col_cs[px], r = row_cs[py];
        d0 = c.x * r.x; d1 = c.y * r.x; d2 = r.y;
        const size_t P = (size_t)cam.H * cam.W;
        const size_t pix = (size_t)py * cam.W + px;
        const uint2 pcn = pix_contrib[pix];
        last = pcn.x; medc = pcn.y;
        const float4 ps = pix_state[pix];
        Tf = ps.x; M1 = ps.y; M2 = ps.z;
        dD = dL_dallmap[SLS_CH_DEPTH * P + pix];
        dA = dL_dallmap[SLS_CH_ALPHA * P + pix];
        dN0 = dL_dallmap[(SLS_CH_NORMAL + 0) * P + pix];
        dN1 = dL_dallmap[(SLS_CH_NORMAL + 1) * P + pix];
        dN2 = dL_dallmap[(SLS_CH_NORMAL + 2) * P + pix];
        dMed = dL_dallmap[SLS_CH_MEDIAN * P + pix];
        dDist = dL_dallmap[SLS_CH_DIST * P + pix];
    }
    const float Af = 1.0f - Tf;

    // wave / tile maxima of n_contrib
    uint32_t wmax = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor(wmax, off, 64));
    if (tid == 0) s_max = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_max, wmax);
    __syncthreads();
    const int tmax = (int)s_max;
    if (tmax == 0) return;

    const int nb = (tmax + kBatch - 1) / kBatch;
    float4 pre[kRec4];
    uint32_t pre_idx = 0, cur_idx = 0;
    auto prefetch = [&](int b) {
#pragma unroll
        for (int i = 0; i < kRec4; ++i) {
            const int k = i * kThreads + tid;
            const int j = k / kRec4, q = k - j * kRec4;
            const int gj = b * kBatch + j;
            if (gj < tmax) {
                const uint32_t idx = vals[range.x + gj];
                pre[i] = rec[(size_t)idx * kRec4 + q];
            } else pre[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        const int gt = b * kBatch + tid;
        pre_idx = (gt < tmax) ? vals[range.x + gt] : 0u;
    };
    prefetch(nb - 1);

    float Tr = Tf, S = 0.0f;
    for (int b = nb - 1; b >= 0; --b) {
        __syncthreads();   // previous batch fully consumed and flushed
#pragma unroll
        for (int i = 0; i < kRec4; ++i) s_rec[i * kThreads + tid] = pre[i];
        cur_idx = pre_idx;
        {
            float4 *z = reinterpret_cast<float4 *>(s_grad + tid * kGrec);
            z[0] = z[1] = z[2] = z[3] = make_float4(0, 0, 0, 0);
        }
        __syncthreads();
        if (b > 0) prefetch(b - 1);
        const int cnt = min(kBatch, tmax - b * kBatch);
        for (int r = kBatch / 64 - 1; r >= 0; --r) {
            const int jl = r * 64 + lane;
            bool pass = false;
            if (jl < cnt && (uint32_t)(b * kBatch + jl + 1) <= wmax)
                pass = cull_pass(s_rec[jl * kRec4 + 4], wcx, wcy, cam.wrap, Wf);
            uint64_t mask = __ballot(pass);
            while (mask) {
                const int jj = 63 - __builtin_clzll(mask);
                mask &= ~(1ull << jj);
                const int j = r * 64 + jj;
                const uint32_t contributor = (uint32_t)(b * kBatch + j + 1);
                const float4 q0 = s_rec[j * kRec4 + 0], q1 = s_rec[j * kRec4 + 1], q2 = s_rec[j * kRec4 + 2];
                const float4 q3 = s_rec[j * kRec4 + 3], q4 = s_rec[j * kRec4 + 4];
                Eval e;
                eval_surfel(q0, q1, q2, q3, q4, d0, d1, d2, pc, pr, cam.wrap, Wf, cam.near_c, e);
                const bool act = inside && (contributor <= last) && !e.skip;
                const uint64_t any = __ballot(act);
                if (!any) continue;
                float gl[kGrec];
#pragma unroll
                for (int k = 0; k < kGrec; ++k) gl[k] = 0.0f;
                bool act3 = false, act2 = false;
                if (act) {
                    const float om = 1.0f - e.alpha;
                    const float rom = __builtin_amdgcn_rcpf(om);
                    Tr = Tr * rom;
                    const float w = e.alpha * Tr;
                    const float rdep = __builtin_amdgcn_rcpf(e.depth);
                    const float m = mscale * (1.0f - cam.near_c * rdep);
                    const float dm_dd = mscale * cam.near_c * rdep * rdep;
                    const float gk = dD * e.depth + (dN0 * q2.x + dN1 * q2.y + dN2 * q2.z) + dA +
                                     dDist * (M2 + m * m * Af - 2.0f * m * M1);
                    const float dL_dalpha = Tr * gk - S * rom;
                    S += w * gk;
                    float dL_ddepth = w * dD + dDist * 2.0f * w * (m * Af - M1) * dm_dd;
                    if (contributor == medc) dL_ddepth += dMed;
                    float dL_do = 0.0f, dL_dG = 0.0f;
                    if (e.og < SLS_ALPHA_MAX) { dL_do = dL_dalpha * e.G; dL_dG = dL_dalpha * q2.w; }
                    const float dL_drho = -0.5f * e.G * dL_dG;
                    gl[8] = w * dN0; gl[9] = w * dN1; gl[10] = w * dN2;
                    gl[11] = dL_do;
                    if (e.use3d) {
                        act3 = true;
                        const float dL_du = dL_drho * 2.0f * e.u, dL_dv = dL_drho * 2.0f * e.v;
                        const float dL_dhu = dL_du * e.rinv, dL_dhv = dL_dv * e.rinv;
                        const float dL_drinv = dL_du * e.hu + dL_dv * e.hv + dL_ddepth * q0.w;
                        const float dL_dnd = -dL_drinv * e.rinv * e.rinv;
                        gl[0] = dL_dhu * e.dl0; gl[1] = dL_dhu * e.dl1; gl[2] = dL_dhu * e.dl2;
                        gl[3] = dL_ddepth * e.rinv;
                        gl[4] = dL_dhv * e.dl0; gl[5] = dL_dhv * e.dl1; gl[6] = dL_dhv * e.dl2;
                        gl[8] += dL_dnd * d0; gl[9] += dL_dnd * d1; gl[10] += dL_dnd * d2;
                        gl[12] = dL_dhu;
                        gl[13] = dL_dhv;
                    } else {
                        act2 = true;
                        gl[7] = dL_ddepth;
                        gl[14] = -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) * e.dx;
                        gl[15] = -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) * e.dy;
                    }
                }
                const bool any3 = __ballot(act3) != 0, any2 = __ballot(act2) != 0;
                float *sg = s_grad + j * kGrec;
#pragma unroll
                for (int k = 0; k < kGrec; ++k) {
                    const bool is3 = (k <= 6) || k == 12 || k == 13;
                    const bool is2 = (k == 7) || (k == 14) || (k == 15);
                    if (is3 && !any3) continue;
                    if (is2 && !any2) continue;
                    const float tot = wave_sum_to_lane63(gl[k]);
                    if (lane == 63) atomicAdd(&sg[k], tot);
                }
            }
        }
        __syncthreads();
        if (tid < cnt) {
            const float *sg = s_grad + tid * kGrec;
            float *g = grec + (size_t)cur_idx * kGrec;
#pragma unroll
            for (int k = 0; k < kGrec; ++k) {
                const float v = sg[k];
                if (v != 0.0f) atomicAdd(&g[k], v);
            }
        }
    }
}

// ---------------------------------------------------------------------------
int launch_render_fwd(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                      const float *col_cs, const float *row_cs, float *allmap, float *pix_state,
                      uint32_t *pix_contrib, uint32_t *tile_consumed, hipStream_t st)
{
    const int T = cam.GX * cam.GY;
    ScopedTimer tm(T_RENDER_FWD, st);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(kThreads), 0, st, cam, (const uint2 *)ranges, vals,
                       (const float4 *)rec, (const float2 *)col_cs, (const float2 *)row_cs, allmap,
                       (float4 *)pix_state, (uint2 *)pix_contrib, tile_consumed);
    SLS_LAUNCH_CHECK("render_fwd_kernel");
    return SLS_OK;
}

int launch_render_bwd(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                      const float *col_cs, const float *row_cs, const float *pix_state,
                      const uint32_t *pix_contrib, const float *dL_dallmap, float *grec, hipStream_t st)
{
    const int T = cam.GX * cam.GY;
    ScopedTimer tm(T_RENDER_BWD, st);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(T), dim3(kThreads), 0, st, cam, (const uint2 *)ranges, vals,
                       (const float4 *)rec, (const float2 *)col_cs, (const float2 *)row_cs,
                       (const float4 *)pix_state, (const uint2 *)pix_contrib, dL_dallmap, grec);
    SLS_LAUNCH_CHECK("render_bwd_kernel");
    return SLS_OK;
}

}  // namespace sls
