// sls_render.hip — per-tile front-to-back blend (A6) and its per-pixel
// backward (A7).  SURVEY.md §8a rows A6/A7; maths in DESIGN.md §2.
//
// Mapping to CDNA4:
//   * one workgroup = one TILE_W x TILE_H tile; each wave owns an 8x8 pixel
//     sub-tile and keeps its own "all 64 pixels saturated" early-out;
//   * the tile's depth-sorted surfel list is staged through LDS in batches of
//     kBatch records (80 B each, 5 x float4).  Staging is lane-linear: thread k
//     of round i fetches float4 #(k mod 5) of record #(k div 5), so adjacent
//     lanes read adjacent 16-B pieces of one record and the LDS image is
//     written with stride-1 ds_write_b128.  Two-deep software pipeline: while
//     batch b is consumed, the records of b+1 and the list indices of b+2 are
//     in flight (no branch between the loads, so they all overlap);
//   * wave64 ballot culling: lane j tests record j of the batch against the
//     bounding box of the wave's still-active pixels (conservative support box
//     from preprocess); the 64-bit ballot is walked with scalar ff1 — surfels
//     that cannot touch an active pixel cost no VALU work, the survivors are
//     evaluated with wave-uniform (broadcast) LDS reads.  The active box
//     shrinks as pixels saturate, which is what bounds the straggler waves;
//   * backward: the 16 per-surfel gradient fields are reduced over the wave's
//     pixels with ONE 16-way DPP reduce-scatter (sls_common.hpp), accumulated
//     across waves with LDS float atomics (16 lanes, 16 distinct addresses) and
//     flushed with one global float atomic per touched (tile, surfel, field).
// No MFMA: there is no dense contraction here (BASELINE.json north_star).
#include "sls_tile.hpp"

namespace sls {

constexpr int kThreads = kTilePix;          // one thread per tile pixel
constexpr int kBatch = kThreads;            // records staged per batch (one index per thread)
constexpr int kRounds = kBatch / 64;
constexpr int kSubX = kTileW / 8;           // wave sub-tiles per tile row

// Two-deep staging pipeline shared by both kernels (macros so that the small
// arrays stay in registers).  `first` is the list offset of the tile, `limit`
// the number of usable entries (>= 1 whenever used).
#define SLS_IDX1(i_, first, b, limit) vals[(first) + (uint32_t)min((b) * kBatch + ((i_) * kThreads + tid) / kRec4, (limit) - 1)]
#define SLS_STAGE_LOAD_IDX(first, b, limit)                                                    \
    si0 = SLS_IDX1(0, first, b, limit); si1 = SLS_IDX1(1, first, b, limit);                     \
    si2 = SLS_IDX1(2, first, b, limit); si3 = SLS_IDX1(3, first, b, limit);                     \
    si4 = SLS_IDX1(4, first, b, limit);
#define SLS_REC1(i_, idx_) rec[(size_t)(idx_) * kRec4 + (((i_) * kThreads + tid) % kRec4)]
#define SLS_STAGE_LOAD_REC()                                                                   \
    sp0 = SLS_REC1(0, si0); sp1 = SLS_REC1(1, si1); sp2 = SLS_REC1(2, si2);                     \
    sp3 = SLS_REC1(3, si3); sp4 = SLS_REC1(4, si4);
#define SLS_STAGE_STORE()                                                                      \
    s_rec[0 * kThreads + tid] = sp0; s_rec[1 * kThreads + tid] = sp1;                           \
    s_rec[2 * kThreads + tid] = sp2; s_rec[3 * kThreads + tid] = sp3;                           \
    s_rec[4 * kThreads + tid] = sp4;

// ---------------------------------------------------------------------------
// A6 forward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void render_fwd_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    float *__restrict__ allmap, float4 *__restrict__ pix_state, uint2 *__restrict__ pix_contrib,
    uint32_t *__restrict__ tile_consumed, uint32_t *__restrict__ dbg_cycles)
{
    __shared__ float4 s_rec[kBatch * kRec4];
    __shared__ uint32_t s_consumed;
    const uint64_t t_start = dbg_cycles ? clock64() : 0;

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
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;

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
    SLS_STAGE_DECL
    if (nb > 0) {
        SLS_STAGE_LOAD_IDX(range.x, 0, n)
        SLS_STAGE_LOAD_REC()
        if (nb > 1) { SLS_STAGE_LOAD_IDX(range.x, 1, n) }
    }

    for (int b = 0; b < nb; ++b) {
        if (__syncthreads_and(wave_done ? 1 : 0)) break;
        SLS_STAGE_STORE()
        __syncthreads();
        if (b + 1 < nb) {
            SLS_STAGE_LOAD_REC()
            if (b + 2 < nb) { SLS_STAGE_LOAD_IDX(range.x, b + 2, n) }
        }
        const int cnt = min(kBatch, n - b * kBatch);
        for (int r = 0; r < kRounds && !wave_done; ++r) {
            float bcx, bcy, bhx, bhy;
            if (!active_box(__ballot(!done), x0, y0, bcx, bcy, bhx, bhy)) { wave_done = true; break; }
            const int jl = r * 64 + lane;
            bool pass = false;
            if (jl < cnt) pass = cull_pass(s_rec[jl * kRec4 + 4], bcx, bcy, bhx, bhy, wrapW, invW);
            uint64_t mask = __ballot(pass);
            while (mask) {
                const int jj = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int j = r * 64 + jj;
                const uint32_t contributor = (uint32_t)(b * kBatch + j + 1);
                const float4 q0 = s_rec[j * kRec4 + 0], q1 = s_rec[j * kRec4 + 1], q2 = s_rec[j * kRec4 + 2];
                const float4 q3 = s_rec[j * kRec4 + 3], q4 = s_rec[j * kRec4 + 4];
                Eval e;
                eval_surfel(q0, q1, q2, q3, q4, d0, d1, d2, pc, pr, wrapW, invW, cam.near_c, e);
                // fully predicated blend (no divergent branches): w == 0 leaves every accumulator unchanged
                const float testT = Tr * (1.0f - e.alpha);
                const bool live = !done && !e.skip;
                const bool term = live && (testT < SLS_T_MIN);
                const bool upd = live && !term;
                const float w = upd ? e.alpha * Tr : 0.0f;
                const float dep = upd ? e.depth : 1.0f;
                const float A = 1.0f - Tr;
                const float m = mscale * (1.0f - cam.near_c * __builtin_amdgcn_rcpf(dep));
                dist += (m * m * A + M2 - 2.0f * m * M1) * w;
                D += dep * w;
                M1 += m * w;
                M2 += m * m * w;
                const bool is_med = upd && (Tr > 0.5f);
                med = is_med ? dep : med;
                medc = is_med ? contributor : medc;
                N0 += q2.x * w; N1 += q2.y * w; N2 += q2.z * w;
                Tr = upd ? testT : Tr;
                last = upd ? contributor : last;
                consumed = term ? contributor : consumed;
                done = done || term;
                if (__all(done)) { wave_done = true; break; }
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
    if (dbg_cycles && lane == 0) dbg_cycles[tile * (kThreads / 64) + wave] = (uint32_t)(clock64() - t_start);
}

// ---------------------------------------------------------------------------
// A6 forward, wave-independent variant: ONE WAVE per workgroup owns an 8x8
// sub-tile and walks the tile's list on its own, 64 entries per round, staging
// them in a private LDS slice.  No workgroup barrier anywhere: a wave never
// waits for a sibling with more surfels to evaluate (in the 4-wave kernel above
// ~30 % of the wave-cycles are barrier waits), at the price of each of the 4
// sub-tiles of a tile fetching the records itself (L2-resident).
// ---------------------------------------------------------------------------
constexpr int kSubPerTile = kTilePix / 64;

__device__ __forceinline__ void wave_tile_of_block(int b, int T, int &tile, int &sub)
{
    tile_of_block<kSubPerTile>(b, T, tile, sub);
}

__global__ __launch_bounds__(64) void render_fwd_wave_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    float *__restrict__ allmap, float4 *__restrict__ pix_state, uint2 *__restrict__ pix_contrib,
    uint32_t *__restrict__ tile_consumed, uint32_t *__restrict__ dbg_cycles)
{
    __shared__ float4 s_rec[64 * kRec4];
    const uint64_t t_start = dbg_cycles ? clock64() : 0;
    const int lane = threadIdx.x;
    const int T = cam.GX * cam.GY;
    int tile, sub;
    wave_tile_of_block(blockIdx.x, T, tile, sub);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int x0 = tx * kTileW + (sub % kSubX) * 8, y0 = ty * kTileH + (sub / kSubX) * 8;
    const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;

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

    uint32_t st_staged = 0, st_pass = 0, st_contrib = 0, st_lanes = 0;   // diagnostics only
    const int nr = (n + 63) / 64;
    SLS_STAGE_DECL
    if (nr > 0 && !wave_done) {
        SLS_WSTAGE_LOAD_IDX(range.x, 0, n)
        SLS_WSTAGE_LOAD_REC()
        if (nr > 1) { SLS_WSTAGE_LOAD_IDX(range.x, 1, n) }
    }
    for (int r = 0; r < nr && !wave_done; ++r) {
        float bcx, bcy, bhx, bhy;
        if (!active_box(__ballot(!done), x0, y0, bcx, bcy, bhx, bhy)) break;
        // LDS of a single wave: program order is enough (no barrier), the previous
        // round's reads are complete before these writes are issued
        SLS_WSTAGE_STORE()
        if (r + 1 < nr) {
            SLS_WSTAGE_LOAD_REC()
            if (r + 2 < nr) { SLS_WSTAGE_LOAD_IDX(range.x, r + 2, n) }
        }
        const int cnt = min(64, n - r * 64);
        bool pass = false;
        if (lane < cnt) pass = cull_pass(s_rec[lane * kRec4 + 4], bcx, bcy, bhx, bhy, wrapW, invW);
        uint64_t mask = __ballot(pass);
        if (dbg_cycles) { st_staged += (uint32_t)cnt; st_pass += (uint32_t)__builtin_popcountll(mask); }
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t contributor = (uint32_t)(r * 64 + j + 1);
            const float4 q0 = s_rec[j * kRec4 + 0], q1 = s_rec[j * kRec4 + 1], q2 = s_rec[j * kRec4 + 2];
            const float4 q3 = s_rec[j * kRec4 + 3], q4 = s_rec[j * kRec4 + 4];
            Eval e;
            eval_surfel(q0, q1, q2, q3, q4, d0, d1, d2, pc, pr, wrapW, invW, cam.near_c, e);
            const bool live = !done && !e.skip;
            if (!__ballot(live)) continue;          // inside the support box but below 1/255 everywhere
            const float testT = Tr * (1.0f - e.alpha);
            const bool term = live && (testT < SLS_T_MIN);
            const bool upd = live && !term;
            if (dbg_cycles) {
                const uint64_t um = __ballot(upd);
                st_contrib += um ? 1u : 0u;
                st_lanes += (uint32_t)__builtin_popcountll(um);
            }
            const float w = upd ? e.alpha * Tr : 0.0f;
            const float dep = upd ? e.depth : 1.0f;
            const float A = 1.0f - Tr;
            const float m = mscale * (1.0f - cam.near_c * __builtin_amdgcn_rcpf(dep));
            dist += (m * m * A + M2 - 2.0f * m * M1) * w;
            D += dep * w;
            M1 += m * w;
            M2 += m * m * w;
            const bool is_med = upd && (Tr > 0.5f);
            med = is_med ? dep : med;
            medc = is_med ? contributor : medc;
            N0 += q2.x * w; N1 += q2.y * w; N2 += q2.z * w;
            Tr = upd ? testT : Tr;
            last = upd ? contributor : last;
            consumed = term ? contributor : consumed;
            done = done || term;
            if (__all(done)) { wave_done = true; break; }
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
    if (tile_consumed) {   // tile value = max over its sub-tiles (buffer zeroed by the launcher)
        uint32_t c = consumed;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c = max(c, (uint32_t)__shfl_down(c, off, 64));
        if (lane == 0) atomicMax(&tile_consumed[tile], c);
    }
    if (dbg_cycles && lane == 0) {
        dbg_cycles[tile * kSubPerTile + sub] = (uint32_t)(clock64() - t_start);
        uint32_t *st = dbg_cycles + (size_t)T * kSubPerTile;   // 4 counters after the per-wave cycles
        atomicAdd(&st[0], st_staged); atomicAdd(&st[1], st_pass);
        atomicAdd(&st[2], st_contrib); atomicAdd(&st[3], st_lanes);
    }
}

// ---------------------------------------------------------------------------
// A7 backward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void render_bwd_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    const float4 *__restrict__ pix_state, const uint2 *__restrict__ pix_contrib,
    const float *__restrict__ dL_dallmap, float *__restrict__ grec, uint32_t *__restrict__ dbg_cycles)
{
    __shared__ float4 s_rec[kBatch * kRec4];
    __shared__ float s_grad[kBatch * kGrec];
    __shared__ uint32_t s_max;
    const uint64_t t_start = dbg_cycles ? clock64() : 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = cam.GX * cam.GY;
    const int tile = xcd_remap(blockIdx.x, T);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int sub_x = wave % kSubX, sub_y = wave / kSubX;
    const int x0 = tx * kTileW + sub_x * 8, y0 = ty * kTileH + sub_y * 8;
    const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;
    const float pc = (float)px, pr = (float)py;
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);
    const int my_comp = reduce16_component(lane);

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
    SLS_STAGE_DECL
    SLS_STAGE_LOAD_IDX(range.x, nb - 1, tmax)
    SLS_STAGE_LOAD_REC()
    // surfel index of list entry (b*kBatch + tid), for the flush
    uint32_t next_idx = vals[range.x + (uint32_t)min((nb - 1) * kBatch + tid, tmax - 1)];
    if (nb > 1) { SLS_STAGE_LOAD_IDX(range.x, nb - 2, tmax) }

    float Tr = Tf, S = 0.0f;
    for (int b = nb - 1; b >= 0; --b) {
        __syncthreads();   // previous batch fully consumed and flushed
        SLS_STAGE_STORE()
        const uint32_t cur_idx = next_idx;
        {
            float4 *z = reinterpret_cast<float4 *>(s_grad + tid * kGrec);
            z[0] = z[1] = z[2] = z[3] = make_float4(0, 0, 0, 0);
        }
        __syncthreads();
        if (b > 0) {
            SLS_STAGE_LOAD_REC()
            next_idx = vals[range.x + (uint32_t)((b - 1) * kBatch + tid)];
            if (b > 1) { SLS_STAGE_LOAD_IDX(range.x, b - 2, tmax) }
        }
        const int cnt = min(kBatch, tmax - b * kBatch);
        for (int r = kRounds - 1; r >= 0; --r) {
            const uint32_t c_lo = (uint32_t)(b * kBatch + r * 64 + 1);   // smallest contributor of this round
            if (c_lo > wmax) continue;
            float bcx, bcy, bhx, bhy;
            if (!active_box(__ballot(inside && last >= c_lo), x0, y0, bcx, bcy, bhx, bhy)) continue;
            const int jl = r * 64 + lane;
            bool pass = false;
            if (jl < cnt && (uint32_t)(b * kBatch + jl + 1) <= wmax)
                pass = cull_pass(s_rec[jl * kRec4 + 4], bcx, bcy, bhx, bhy, wrapW, invW);
            uint64_t mask = __ballot(pass);
            while (mask) {
                const int jj = 63 - __builtin_clzll(mask);
                mask &= ~(1ull << jj);
                const int j = r * 64 + jj;
                const uint32_t contributor = (uint32_t)(b * kBatch + j + 1);
                const float4 q0 = s_rec[j * kRec4 + 0], q1 = s_rec[j * kRec4 + 1], q2 = s_rec[j * kRec4 + 2];
                const float4 q3 = s_rec[j * kRec4 + 3], q4 = s_rec[j * kRec4 + 4];
                Eval e;
                eval_surfel(q0, q1, q2, q3, q4, d0, d1, d2, pc, pr, wrapW, invW, cam.near_c, e);
                const bool act = inside && (contributor <= last) && !e.skip;
                if (!__ballot(act)) continue;
                // predicated gradient (no divergent branches); selects, not products, gate the
                // 3D/2D fields so an infinite rinv of an inactive lane cannot leak a NaN
                const float om = act ? 1.0f - e.alpha : 1.0f;
                const float rom = __builtin_amdgcn_rcpf(om);
                Tr = Tr * rom;
                const float w = act ? e.alpha * Tr : 0.0f;
                const float dep = act ? e.depth : 1.0f;
                const float rdep = __builtin_amdgcn_rcpf(dep);
                const float m = mscale * (1.0f - cam.near_c * rdep);
                const float dm_dd = mscale * cam.near_c * rdep * rdep;
                const float gk = dD * dep + (dN0 * q2.x + dN1 * q2.y + dN2 * q2.z) + dA +
                                 dDist * (M2 + m * m * Af - 2.0f * m * M1);
                const float dL_dalpha = act ? Tr * gk - S * rom : 0.0f;
                S += w * gk;
                float dL_ddepth = w * dD + dDist * 2.0f * w * (m * Af - M1) * dm_dd;
                dL_ddepth += (act && contributor == medc) ? dMed : 0.0f;
                const bool unclamped = e.og < SLS_ALPHA_MAX;
                const float dL_do = unclamped ? dL_dalpha * e.G : 0.0f;
                const float dL_drho = unclamped ? -0.5f * e.G * dL_dalpha * q2.w : 0.0f;
                const bool a3 = act && e.use3d, a2 = act && !e.use3d;
                const float dL_du = dL_drho * 2.0f * e.u, dL_dv = dL_drho * 2.0f * e.v;
                const float dL_dhu = a3 ? dL_du * e.rinv : 0.0f, dL_dhv = a3 ? dL_dv * e.rinv : 0.0f;
                const float dL_drinv = dL_du * e.hu + dL_dv * e.hv + dL_ddepth * q0.w;
                const float dL_dnd = a3 ? -dL_drinv * e.rinv * e.rinv : 0.0f;
                float gl[kGrec];
                gl[0] = dL_dhu * e.dl0; gl[1] = dL_dhu * e.dl1; gl[2] = dL_dhu * e.dl2;
                gl[3] = a3 ? dL_ddepth * e.rinv : 0.0f;
                gl[4] = dL_dhv * e.dl0; gl[5] = dL_dhv * e.dl1; gl[6] = dL_dhv * e.dl2;
                gl[7] = a2 ? dL_ddepth : 0.0f;
                gl[8] = w * dN0 + dL_dnd * d0; gl[9] = w * dN1 + dL_dnd * d1; gl[10] = w * dN2 + dL_dnd * d2;
                gl[11] = dL_do;
                gl[12] = dL_dhu;
                gl[13] = dL_dhv;
                gl[14] = a2 ? -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) * e.dx : 0.0f;
                gl[15] = a2 ? -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) * e.dy : 0.0f;
                const float tot = wave_reduce16(gl, lane);
                if ((lane & 3) == 0 && tot != 0.0f) atomicAdd(&s_grad[j * kGrec + my_comp], tot);
            }
        }
        __syncthreads();
        if (tid < cnt) {
            const float *sgr = s_grad + tid * kGrec;
            float *g = grec + (size_t)cur_idx * kGrec;
#pragma unroll
            for (int k = 0; k < kGrec; ++k) {
                const float v = sgr[k];
                if (v != 0.0f) atomicAdd(&g[k], v);
            }
        }
    }
    if (dbg_cycles && lane == 0) dbg_cycles[tile * (kThreads / 64) + wave] = (uint32_t)(clock64() - t_start);
}

// ---------------------------------------------------------------------------
// A7 backward, wave-independent variant: one wave per 8x8 sub-tile, private LDS
// staging, no workgroup barrier.  The 16 reduced gradient fields of a surfel end
// up in lanes 0..15 (one field each), which add them to the surfel's 64-byte
// gradient record with ONE global_atomic_add_f32 instruction (16 lanes, one
// cache line) — no LDS accumulation, no flush phase.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void render_bwd_wave_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    const float4 *__restrict__ pix_state, const uint2 *__restrict__ pix_contrib,
    const float *__restrict__ dL_dallmap, float *__restrict__ grec, uint32_t *__restrict__ dbg_cycles)
{
    __shared__ float4 s_rec[64 * kRec4];
    const uint64_t t_start = dbg_cycles ? clock64() : 0;
    const int lane = threadIdx.x;
    const int T = cam.GX * cam.GY;
    int tile, sub;
    wave_tile_of_block(blockIdx.x, T, tile, sub);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int x0 = tx * kTileW + (sub % kSubX) * 8, y0 = ty * kTileH + (sub / kSubX) * 8;
    const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;
    const float pc = (float)px, pr = (float)py;
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);
    const int my_comp = reduce16_component(lane);

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
    uint32_t wmax = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor(wmax, off, 64));
    const int tmax = (int)wmax;
    if (tmax > 0) {
        const int nr = (tmax + 63) / 64;
        SLS_STAGE_DECL
        SLS_WSTAGE_LOAD_IDX(range.x, nr - 1, tmax)
        SLS_WSTAGE_LOAD_REC()
        uint32_t next_idx = vals[range.x + (uint32_t)min((nr - 1) * 64 + lane, tmax - 1)];   // surfel of entry (r*64 + lane)
        if (nr > 1) { SLS_WSTAGE_LOAD_IDX(range.x, nr - 2, tmax) }
        float Tr = Tf, S = 0.0f;
        for (int r = nr - 1; r >= 0; --r) {
            SLS_WSTAGE_STORE()
            const uint32_t my_idx = next_idx;
            if (r > 0) {
                SLS_WSTAGE_LOAD_REC()
                next_idx = vals[range.x + (uint32_t)((r - 1) * 64 + lane)];
                if (r > 1) { SLS_WSTAGE_LOAD_IDX(range.x, r - 2, tmax) }
            }
            const int cnt = min(64, tmax - r * 64);
            const uint32_t c_lo = (uint32_t)(r * 64 + 1);
            float bcx, bcy, bhx, bhy;
            if (!active_box(__ballot(inside && last >= c_lo), x0, y0, bcx, bcy, bhx, bhy)) continue;
            bool pass = false;
            if (lane < cnt) pass = cull_pass(s_rec[lane * kRec4 + 4], bcx, bcy, bhx, bhy, wrapW, invW);
            uint64_t mask = __ballot(pass);
            while (mask) {
                const int j = 63 - __builtin_clzll(mask);
                mask &= ~(1ull << j);
                const uint32_t contributor = (uint32_t)(r * 64 + j + 1);
                const float4 q0 = s_rec[j * kRec4 + 0], q1 = s_rec[j * kRec4 + 1], q2 = s_rec[j * kRec4 + 2];
                const float4 q3 = s_rec[j * kRec4 + 3], q4 = s_rec[j * kRec4 + 4];
                Eval e;
                eval_surfel(q0, q1, q2, q3, q4, d0, d1, d2, pc, pr, wrapW, invW, cam.near_c, e);
                const bool act = inside && (contributor <= last) && !e.skip;
                if (!__ballot(act)) continue;
                const float om = act ? 1.0f - e.alpha : 1.0f;
                const float rom = __builtin_amdgcn_rcpf(om);
                Tr = Tr * rom;
                const float w = act ? e.alpha * Tr : 0.0f;
                const float dep = act ? e.depth : 1.0f;
                const float rdep = __builtin_amdgcn_rcpf(dep);
                const float m = mscale * (1.0f - cam.near_c * rdep);
                const float dm_dd = mscale * cam.near_c * rdep * rdep;
                const float gk = dD * dep + (dN0 * q2.x + dN1 * q2.y + dN2 * q2.z) + dA +
                                 dDist * (M2 + m * m * Af - 2.0f * m * M1);
                const float dL_dalpha = act ? Tr * gk - S * rom : 0.0f;
                S += w * gk;
                float dL_ddepth = w * dD + dDist * 2.0f * w * (m * Af - M1) * dm_dd;
                dL_ddepth += (act && contributor == medc) ? dMed : 0.0f;
                const bool unclamped = e.og < SLS_ALPHA_MAX;
                const float dL_do = unclamped ? dL_dalpha * e.G : 0.0f;
                const float dL_drho = unclamped ? -0.5f * e.G * dL_dalpha * q2.w : 0.0f;
                const bool a3 = act && e.use3d, a2 = act && !e.use3d;
                const float dL_du = dL_drho * 2.0f * e.u, dL_dv = dL_drho * 2.0f * e.v;
                const float dL_dhu = a3 ? dL_du * e.rinv : 0.0f, dL_dhv = a3 ? dL_dv * e.rinv : 0.0f;
                const float dL_drinv = dL_du * e.hu + dL_dv * e.hv + dL_ddepth * q0.w;
                const float dL_dnd = a3 ? -dL_drinv * e.rinv * e.rinv : 0.0f;
                float gl[kGrec];
                gl[0] = dL_dhu * e.dl0; gl[1] = dL_dhu * e.dl1; gl[2] = dL_dhu * e.dl2;
                gl[3] = a3 ? dL_ddepth * e.rinv : 0.0f;
                gl[4] = dL_dhv * e.dl0; gl[5] = dL_dhv * e.dl1; gl[6] = dL_dhv * e.dl2;
                gl[7] = a2 ? dL_ddepth : 0.0f;
                gl[8] = w * dN0 + dL_dnd * d0; gl[9] = w * dN1 + dL_dnd * d1; gl[10] = w * dN2 + dL_dnd * d2;
                gl[11] = dL_do;
                gl[12] = dL_dhu;
                gl[13] = dL_dhv;
                gl[14] = a2 ? -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) * e.dx : 0.0f;
                gl[15] = a2 ? -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) * e.dy : 0.0f;
                const float tot = wave_reduce16(gl, lane);
                const uint32_t gidx = (uint32_t)__builtin_amdgcn_readlane((int)my_idx, j);
                if ((lane & 3) == 0 && tot != 0.0f) atomicAdd(&grec[(size_t)gidx * kGrec + my_comp], tot);
            }
        }
    }
    if (dbg_cycles && lane == 0) dbg_cycles[tile * kSubPerTile + sub] = (uint32_t)(clock64() - t_start);
}

// kernel variants (sls_debug_variant): 0 = one workgroup per tile, 1 = one wave per 8x8 sub-tile,
// 2 / 3 = one wave per 4x4 / 8x2 pixel block x 4 surfel slots (sls_render_block.hip)
int launch_render_fwd_block(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                            const float *col_cs, const float *row_cs, float *allmap, float *pix_state,
                            uint32_t *pix_contrib, uint32_t *tile_consumed, uint64_t *block_masks, int shape,
                            hipStream_t st);
size_t block_mask_bytes(uint64_t cap, int T);
int launch_render_bwd_block(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                            const float *col_cs, const float *row_cs, const float *pix_state,
                            const uint32_t *pix_contrib, const float *dL_dallmap, float *grec,
                            const uint64_t *block_masks, int shape, hipStream_t st, bool lean, uint8_t *touched,
                            const struct ConsumerArgs *fused_consumer);
int g_fwd_variant = 3, g_bwd_variant = 3;
// unused dynamic LDS requested at launch (caps the workgroups resident per CU)
int g_pad_lds_fwd = 0, g_pad_lds_bwd = 0;

// diagnostic: per-wave shader-clock counts (sls_debug_wave_cycles)
uint32_t *g_dbg_fwd_cycles = nullptr, *g_dbg_bwd_cycles = nullptr;

// ---------------------------------------------------------------------------
int launch_render_fwd(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                      const float *col_cs, const float *row_cs, float *allmap, float *pix_state,
                      uint32_t *pix_contrib, uint32_t *tile_consumed, hipStream_t st, bool consumed_zeroed,
                      uint64_t *block_masks)
{
    const int T = cam.GX * cam.GY;
    // only the block kernels fill the contribution masks: invalidate the tag otherwise
    if (g_fwd_variant < 2 && block_masks) SLS_HIP_CHECK(hipMemsetAsync(block_masks, 0, sizeof(uint64_t), st));
    // the wave / block kernels combine their sub-tiles with atomicMax: start from zero
    if (g_fwd_variant >= 1 && tile_consumed && !consumed_zeroed)
        SLS_HIP_CHECK(hipMemsetAsync(tile_consumed, 0, sizeof(uint32_t) * (size_t)T, st));
    if (g_fwd_variant >= 2)
        return launch_render_fwd_block(cam, ranges, vals, rec, col_cs, row_cs, allmap, pix_state, pix_contrib,
                                       tile_consumed, block_masks, g_fwd_variant - 2, st);
    if (g_fwd_variant == 1) {
        ScopedTimer tm(T_RENDER_FWD, st);
        hipLaunchKernelGGL(render_fwd_wave_kernel, dim3(T * kSubPerTile), dim3(64), g_pad_lds_fwd, st, cam, (const uint2 *)ranges,
                           vals, (const float4 *)rec, (const float2 *)col_cs, (const float2 *)row_cs, allmap,
                           (float4 *)pix_state, (uint2 *)pix_contrib, tile_consumed, g_dbg_fwd_cycles);
        SLS_LAUNCH_CHECK("render_fwd_wave_kernel");
        return SLS_OK;
    }
    ScopedTimer tm(T_RENDER_FWD, st);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(T), dim3(kThreads), 0, st, cam, (const uint2 *)ranges, vals,
                       (const float4 *)rec, (const float2 *)col_cs, (const float2 *)row_cs, allmap,
                       (float4 *)pix_state, (uint2 *)pix_contrib, tile_consumed, g_dbg_fwd_cycles);
    SLS_LAUNCH_CHECK("render_fwd_kernel");
    return SLS_OK;
}

int launch_render_bwd(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                      const float *col_cs, const float *row_cs, const float *pix_state,
                      const uint32_t *pix_contrib, const float *dL_dallmap, float *grec, hipStream_t st,
                      const uint64_t *block_masks, bool no_median_dist_grad, uint8_t *touched,
                      const struct ConsumerArgs *fused_consumer)
{
    const int T = cam.GX * cam.GY;
    if (g_bwd_variant >= 2)
        return launch_render_bwd_block(cam, ranges, vals, rec, col_cs, row_cs, pix_state, pix_contrib, dL_dallmap,
                                       grec, block_masks, g_bwd_variant - 2, st, no_median_dist_grad, touched,
                                       fused_consumer);
    SLS_REQUIRE(!touched && !fused_consumer, "only the block kernels mark the touched surfels / fuse the consumer");
    ScopedTimer tm(T_RENDER_BWD, st);
    if (g_bwd_variant == 1) {
        hipLaunchKernelGGL(render_bwd_wave_kernel, dim3(T * kSubPerTile), dim3(64), g_pad_lds_bwd, st, cam,
                           (const uint2 *)ranges, vals, (const float4 *)rec, (const float2 *)col_cs,
                           (const float2 *)row_cs, (const float4 *)pix_state, (const uint2 *)pix_contrib, dL_dallmap,
                           grec, g_dbg_bwd_cycles);
        SLS_LAUNCH_CHECK("render_bwd_wave_kernel");
        return SLS_OK;
    }
    hipLaunchKernelGGL(render_bwd_kernel, dim3(T), dim3(kThreads), 0, st, cam, (const uint2 *)ranges, vals,
                       (const float4 *)rec, (const float2 *)col_cs, (const float2 *)row_cs,
                       (const float4 *)pix_state, (const uint2 *)pix_contrib, dL_dallmap, grec, g_dbg_bwd_cycles);
    SLS_LAUNCH_CHECK("render_bwd_kernel");
    return SLS_OK;
}

}  // namespace sls
