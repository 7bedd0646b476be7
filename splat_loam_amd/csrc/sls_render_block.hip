// sls_render_block.hip — tile kernels, "pixel block x surfel slot" variant.
// SURVEY.md §8a rows A6/A7; maths in DESIGN.md §2; same inputs, outputs and
// per-pixel state as the kernels of sls_render.hip (the variants are
// interchangeable: forward of one, backward of the other).
//
// Why: with one pixel per lane a wave evaluates ONE surfel per ~110-instruction
// step and the kernel time is the serial instruction stream of the longest
// wave (2048 waves for 1024 SIMDs, nothing to backfill).  Here a wave owns a
// BW x BH = 16 pixel block and evaluates FOUR consecutive list entries per
// step: lane = 4 * pixel + slot.  The front-to-back dependence between the four
// slots of a pixel is resolved inside the quad with DPP (exclusive transmittance
// product, exclusive sums of the distortion moments, first-terminating-slot
// mask), which costs ~35 instructions per step, so
//   * the serial chain per wave is ~4x shorter and there are 4x more waves
//     (8 per SIMD): the SIMDs stay busy and the image rows balance;
//   * the support-box cull works on 4x4 pixels instead of 8x8, so far fewer
//     lanes evaluate a surfel that cannot touch them;
//   * backward: the 16 gradient fields are reduced over the 16 pixels of a slot
//     with ONE reduce-scatter per step (4 surfels) instead of one per surfel,
//     and one 64-lane float atomic instruction flushes 4 surfels x 16 fields.
// Survivors of the cull are compacted into a wave-private LDS list (ballot +
// mbcnt), a step takes the next four.
#include <cstring>
#include "sls_tile.hpp"
#include "sls_consumer_dev.hpp"

namespace sls {

#ifdef SLS_TRACE
// Experiment build only (make FAST='$(COMMON) -munsafe-fp-atomics -DSLS_TRACE'; read back with sls_debug_wave_cycles, profiles/r03l_wave_trace*.txt): every wave of
// the tile kernels records when it ran (100 MHz wall clock), where (HW_ID) and how many rounds / steps it did.
__device__ uint32_t g_trace[2][8192 * 4];
// forward: shader clocks a wave spent waiting for the staged records (+ LDS store) / in cull + compaction / in the
// steps / at the round's end
__device__ uint32_t g_trace_phase[8192 * 4];
__device__ uint32_t g_trace_phase_b[8192 * 4];    // backward: wait for the staged records + store / list / steps / (unused)
// marks: wall-clock offsets (10 ns units) from the wave's start at up to four points of its life, taken when the
// value passed (the result of the loads the point waits for) is in a register
__device__ uint32_t g_trace_mark[2][8192 * 4];
#define SLS_MARK(kern_, k_, v_) { asm volatile("" :: "v"(v_)); const uint32_t m_ = (uint32_t)(wall_clock64() - trace_t0); \
                                  if (threadIdx.x == 0 && blockIdx.x < 8192) g_trace_mark[kern_][4 * blockIdx.x + (k_)] = m_; }
#define SLS_PHASE_DECL() uint64_t ph_t = clock64(); uint32_t ph_acc[4] = { 0, 0, 0, 0 }
#define SLS_PHASE(k_) { const uint64_t now_ = clock64(); ph_acc[k_] += (uint32_t)(now_ - ph_t); ph_t = now_; }
#define SLS_PHASE_RESET() ph_t = clock64()
#define SLS_PHASE_END() if (threadIdx.x == 0 && blockIdx.x < 8192) { for (int k_ = 0; k_ < 4; ++k_) g_trace_phase[4 * blockIdx.x + k_] = ph_acc[k_]; }
#define SLS_PHASE_END_B() if (threadIdx.x == 0 && blockIdx.x < 8192) { for (int k_ = 0; k_ < 4; ++k_) g_trace_phase_b[4 * blockIdx.x + k_] = ph_acc[k_]; }
#define SLS_TRACE_BEGIN() const uint64_t trace_t0 = wall_clock64(); uint32_t trace_rounds = 0, trace_steps = 0, trace_sparse = 0
#define SLS_TRACE_ACTIVE(m_) { const int na_ = __builtin_popcountll((m_) & 0x1111111111111111ull); trace_sparse += (na_ <= 1 ? 1u : 0u) + (na_ <= 2 ? 1u << 10 : 0u) + (na_ <= 4 ? 1u << 20 : 0u); }
#define SLS_TRACE_ROUND() ++trace_rounds
#define SLS_TRACE_STEP() ++trace_steps
#define SLS_TRACE_END(k_)                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 8192) {                                                          \
        uint32_t *t = g_trace[k_] + 4 * blockIdx.x;                                                       \
        t[0] = (uint32_t)trace_t0; t[1] = (uint32_t)wall_clock64();                                       \
        t[2] = (k_) == 0 ? trace_sparse : __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       \
        t[3] = (trace_rounds << 16) | (trace_steps & 0xFFFFu);                                            \
    }
#else
#define SLS_MARK(kern_, k_, v_)
#define SLS_TRACE_BEGIN()
#define SLS_TRACE_ACTIVE(m_)
#define SLS_TRACE_ROUND()
#define SLS_TRACE_STEP()
#define SLS_TRACE_END(k_)
#define SLS_PHASE_DECL()
#define SLS_PHASE(k_)
#define SLS_PHASE_RESET()
#define SLS_PHASE_END()
#define SLS_PHASE_END_B()
#endif

template <int CTRL>
__device__ __forceinline__ float dppq(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dppq_u(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kQuadShr1 = 0x90;   // quad_perm [0,0,1,2]: slot s reads slot s-1 (slot 0 itself)
constexpr int kQuadXor1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;   // quad_perm [2,3,0,1]
__device__ __forceinline__ float quad_sum(float v)
{
    v += dppq<kQuadXor1>(v);
    v += dppq<kQuadXor2>(v);
    return v;
}
__device__ __forceinline__ uint32_t quad_max(uint32_t v)
{
    v = max(v, dppq_u<kQuadXor1>(v));
    v = max(v, dppq_u<kQuadXor2>(v));
    return v;
}
// Exclusive prefix over the 4 slots of a quad (k1,k2,k3 = 1.0f where slot >= 1,2,3 else 0)
// and the quad total, summed in slot order.
__device__ __forceinline__ void quad_excl_total(float x, float k1, float k2, float k3, float &excl, float &total)
{
    excl = k1 * dppq<0x00>(x);
    excl = fmaf(k2, dppq<0x55>(x), excl);
    excl = fmaf(k3, dppq<0xAA>(x), excl);
    total = dppq<0xFF>(excl + x);
}

// Forward -> backward hand-over: per pixel block the COMPACT list of the tile-list entries that contributed to at
// least one of its pixels — (list position, surfel) pairs in list order — so that the backward evaluates exactly
// those, 64 per round (round 3: it used to be one 64-bit word per (tile, 64 list positions, block); a backward round
// then staged 64 records for the ~12 that were marked, and its rounds — wait for the records 2000-2400 shader clocks,
// mask + list 1000-1200, against 1600 per step — were 28 % of a wave's life at C3 and 40 % at the mapper's real
// sizes, profiles/r03l_wave_trace.txt).
// Layout in 8-byte words: [0] tag naming the producer's block shape; [1, 1 + T*8) the blocks' entry counts (u32,
// T*16 of them); [1 + T*8, 1 + T*16) the launch order of the backward's blocks, most expensive first (u32, written by
// block_order_kernel behind the staged forward; sls_mapping_step keeps its own, made by the consumer's launch); then
// the entries: block `sub` of a tile whose list is [first, first + n) owns
// [first*16 + sub*n, first*16 + (sub+1)*n) — room for every entry of the tile, so 16 words per instance of capacity
// (only what contributes is ever written or read).
__host__ __device__ inline uint64_t block_mask_tag(int bw) { return 0x534C4C4953540000ull | (uint64_t)bw; }
__host__ __device__ inline size_t block_list_order_word(int T, int per_tile) { return 1 + ((size_t)T * per_tile + 1) / 2; }
__host__ __device__ inline size_t block_list_entries_word(int T, int per_tile) { return 1 + 2 * (((size_t)T * per_tile + 1) / 2); }
__host__ __device__ inline uint32_t block_backward_cost(uint32_t entries)
{
    // steps, a round of 64 entries ~1.3 steps' worth + its latency (one byte: the order is a counting sort)
    return min(255u, 2u * ((entries + 63u) / 64u) + (entries + 3u) / 4u);
}

// The backward's blocks of one XCD (index i = tile slot * 16 + block; tile = ((slot >> 2) * 8 + xcd) * 4 + (slot & 3),
// the mapping of tile_of_block) ordered by their cost, most expensive first, from the forward's entry counts: a counting
// sort per XCD, eight workgroups.  A permutation whatever the counts are: only speed depends on it.
__global__ __launch_bounds__(256) void block_order_kernel(uint64_t *__restrict__ blk_mask, int T)
{
    constexpr int kPerTile = kTilePix / 16;
    __shared__ uint32_t s_hist[256];
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(blk_mask + 1);
    uint32_t *order = reinterpret_cast<uint32_t *>(blk_mask + block_list_order_word(T, kPerTile));
    const int xcd = blockIdx.x, n_x = T * kPerTile / 8, tid = threadIdx.x;
    s_hist[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < n_x; i += 256) {
        const int ts = i / kPerTile;
        const int tile = ((ts >> 2) * 8 + xcd) * 4 + (ts & 3);
        atomicAdd(&s_hist[255u - block_backward_cost(counts[tile * kPerTile + (i % kPerTile)])], 1u);
    }
    __syncthreads();
    if (tid < 64) {      // exclusive scan of the 256 bins
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
    for (int i = tid; i < n_x; i += 256) {
        const int ts = i / kPerTile;
        const int tile = ((ts >> 2) * 8 + xcd) * 4 + (ts & 3);
        const uint32_t bin = 255u - block_backward_cost(counts[tile * kPerTile + (i % kPerTile)]);
        order[(size_t)xcd * n_x + atomicAdd(&s_hist[bin], 1u)] = (uint32_t)i;
    }
}
size_t block_mask_bytes(uint64_t cap, int T)
{
    return sizeof(uint64_t) * (block_list_entries_word(T, kTilePix / 16) + (size_t)cap * (size_t)(kTilePix / 16));
}

// Box (pixel coordinates, centre + half extents) of the pixels whose slot-0 lane
// is set in `m`, for a BW-wide block at (x0, y0); all scalar work.
template <int BW, int BH>
__device__ __forceinline__ bool block_active_box(uint64_t m, int x0, int y0, float &bcx, float &bcy, float &bhx,
                                                 float &bhy)
{
    m &= 0x1111111111111111ull;
    if (m == 0) return false;
    constexpr int kRowBits = 4 * BW;
    uint64_t c = m;
#pragma unroll
    for (int r = 1; r < BH; ++r) c |= m >> (kRowBits * r);
    const uint32_t cols = (uint32_t)(c & ((kRowBits == 32) ? 0xFFFFFFFFull : ((1ull << (kRowBits & 31)) - 1ull)));
    const int xa = __builtin_ctz(cols) >> 2, xb = (31 - __builtin_clz(cols)) >> 2;
    const int ya = __builtin_ctzll(m) / kRowBits, yb = (63 - __builtin_clzll(m)) / kRowBits;
    bcx = (float)x0 + 0.5f * (float)(xa + xb);
    bhx = 0.5f * (float)(xb - xa);
    bcy = (float)y0 + 0.5f * (float)(ya + yb);
    bhy = 0.5f * (float)(yb - ya);
    return true;
}

// ---------------------------------------------------------------------------
// A6 forward
// ---------------------------------------------------------------------------
// LEAN (sls_mapping_step with depth_ratio = 0): the consumer neither reads the median / distortion channels
// nor sends a gradient into them (gaussian_renderer/__init__.py:79-86 at depth_ratio 0; dL/dallmap[5:7] == 0),
// and the LEAN backward does not read the two distortion moments or the median contributor: they are not
// tracked (planes 5 and 6 are written as zeros) — one reciprocal and ~11 instructions less per step.
#ifndef SLS_FWD_WAVES
#define SLS_FWD_WAVES 4
#endif
template <int BW, int BH, bool DBG, bool LEAN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SLS_FWD_WAVES, SLS_FWD_WAVES))) void render_fwd_block_kernel(uint64_t *__restrict__ blk_mask,
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    float *__restrict__ allmap, float4 *__restrict__ pix_state, uint2 *__restrict__ pix_contrib,
    uint32_t *__restrict__ tile_consumed, uint32_t *__restrict__ dbg_cycles, uint32_t *__restrict__ block_cost)
{
    static_assert(BW * BH == 16 && kTileW % BW == 0 && kTileH % BH == 0, "16-pixel blocks tiling a tile");
    constexpr int kPerTile = kTilePix / 16, kBX = kTileW / BW;
    // (record 64 is all zeros — opacity 0, range 0: never live — and pads the compacted list to a multiple of four,
    //  so that a step needs no "is my slot beyond the list" test)
    __shared__ float4 s_rec[65 * kRec4];
    __shared__ uint32_t s_list[64 + 4];
    __shared__ uint32_t s_flag[65];
    uint32_t ccnt = 0;                            // entries of this block's compact list so far
    const uint64_t t_start = DBG ? clock64() : 0;
    SLS_TRACE_BEGIN();
    const int lane = threadIdx.x, slot = lane & 3, p = lane >> 2;
    const int T = cam.GX * cam.GY;
    int tile, sub;
    tile_of_block<kPerTile>(blockIdx.x, T, tile, sub);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int x0 = tx * kTileW + (sub % kBX) * BW, y0 = ty * kTileH + (sub / kBX) * BH;
    const int px = x0 + (p % BW), py = y0 + (p / BW);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;

    v2f d01 = mk2(1.0f, 0.0f);
    float d2 = 0.0f;
    if (inside) {
        const float2 c = col_cs[px], r = row_cs[py];
        d01 = mk2(c.x * r.x, c.y * r.x); d2 = r.y;
    }
    const v2f pcr = mk2((float)px, (float)py);
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);
    const uint32_t below = (1u << slot) - 1u, upto = (2u << slot) - 1u;   // quad bits of the earlier slots (and self)
    const bool sge1 = slot >= 1, sge2 = slot >= 2, sge3 = slot >= 3;

    // replicated over the quad: Tr, done.  Per-lane partial sums: D, N*, M1, M2.
    float Tr = 1.0f, M1 = 0.0f, M2 = 0.0f;
    float D = 0.0f, N2 = 0.0f, med = 0.0f;
    v2f N01 = mk2(0.0f, 0.0f);
    uint32_t medc = 0, last = 0, cons = 0;
    bool done = !inside;
    bool wave_done = wave_all(done);
    uint32_t st_staged = 0, st_pass = 0, st_steps = 0, st_lanes = 0, st_slots = 0, st_geom = 0;   // diagnostics only

    // footprint test against the whole block (sls_tile.hpp): rays of the block = d0 + x Dx + y Dy
    const BlockCone cone = make_block_cone(cam, (float)x0 + 0.5f * (float)(BW - 1), (float)y0 + 0.5f * (float)(BH - 1),
                                           0.5f * (float)(BW - 1), 0.5f * (float)(BH - 1));
    const int nr = (n + 63) / 64;
    SLS_MARK(0, 0, n);
    SLS_MARK(0, 1, d2);
    if (lane < kRec4) s_rec[64 * kRec4 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    SLS_STAGE_DECL
    if (nr > 0 && !wave_done) {
        SLS_WSTAGE_LOAD_IDX(range.x, 0, n)
        SLS_WSTAGE_LOAD_REC()
        if (nr > 1) { SLS_WSTAGE_LOAD_IDX(range.x, 1, n) }
    }
    if (blk_mask && blockIdx.x == 0 && lane == 0) blk_mask[0] = block_mask_tag(BW);
    // the compact list of this block and the surfel of list entry (r * 64 + lane), requested a round ahead
    uint2 *const clist = blk_mask ? reinterpret_cast<uint2 *>(blk_mask + block_list_entries_word(T, kPerTile))
                                        + ((size_t)range.x * kPerTile + (size_t)sub * (size_t)n) : nullptr;
    uint32_t my_idx_next = (blk_mask && nr > 0 && !wave_done) ? vals[range.x + (uint32_t)min(lane, n - 1)] : 0u;
    SLS_PHASE_DECL();
    for (int r = 0; r < nr && !wave_done; ++r) {
        float bcx, bcy, bhx, bhy;
        if (!block_active_box<BW, BH>(wave_ballot(!done), x0, y0, bcx, bcy, bhx, bhy)) break;
        SLS_PHASE_RESET();
        SLS_TRACE_ROUND();
        SLS_TRACE_ACTIVE(wave_ballot(!done));
        if (blk_mask) s_flag[lane] = 0u;   // entries of this round that reach at least one pixel of the block
        // single wave: LDS operations complete in program order, no barrier needed
        SLS_WSTAGE_STORE()
        if (r == 0) SLS_MARK(0, 2, sp4.x);
        const uint32_t my_idx = my_idx_next;
        if (r + 1 < nr) {
            SLS_WSTAGE_LOAD_REC()
            if (blk_mask) my_idx_next = vals[range.x + (uint32_t)min((r + 1) * 64 + lane, n - 1)];
            if (r + 2 < nr) { SLS_WSTAGE_LOAD_IDX(range.x, r + 2, n) }
        }
        const int cnt = min(64, n - r * 64);
        __builtin_amdgcn_wave_barrier();
        SLS_PHASE(0);
        bool pass = false;
        if (lane < cnt) {
            const float4 c4 = s_rec[lane * kRec4 + 4];
            pass = cull_pass(c4, bcx, bcy, bhx, bhy, wrapW, invW);
            if (pass) {
                const float4 c3 = s_rec[lane * kRec4 + 3];
                pass = !cone_outside(cone, s_rec[lane * kRec4 + 0], s_rec[lane * kRec4 + 1], s_rec[lane * kRec4 + 2], c3) ||
                       disc_reaches(c3, c4, bcx, bcy, bhx, bhy, wrapW, invW);
            }
        }
        const uint64_t mask = wave_ballot(pass);
        const int npass = __builtin_popcountll(mask);
        if (pass) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint32_t)lane;
        if (lane < 3) s_list[npass + lane] = 64u;          // pad to a multiple of four with the empty record
        __builtin_amdgcn_wave_barrier();
        SLS_PHASE(1);
        if (DBG) { st_staged += (uint32_t)cnt; st_pass += (uint32_t)npass; }
        for (int k = 0; k < npass; k += 4) {
            const int j = (int)s_list[k + slot];
            const uint32_t contributor = (uint32_t)(r * 64 + j + 1);
            const float4 *sr = s_rec + __umul24((unsigned)j, (unsigned)kRec4);
            const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], q3 = sr[3], q4 = sr[4];
            Eval e;
            eval_surfel(q0, q1, q2, q3, q4, d01, d2, pcr, wrapW, invW, cam.near_c, e);
            const bool live = !done && !e.skip;
            if (DBG) {
                const uint64_t lb = wave_ballot(live);
                st_steps += 1u; st_lanes += (uint32_t)__builtin_popcountll(lb);
                const uint64_t gb = wave_ballot(inside && !e.skip);      // ignoring finished pixels
                for (int q = 0; q < 4; ++q) {
                    st_slots += (lb & (0x1111111111111111ull << q)) ? 1u : 0u;
                    st_geom += (gb & (0x1111111111111111ull << q)) ? 1u : 0u;
                }
            }
#ifdef SLS_FWD_SKIP_DEAD
            // (a step none of whose 64 (pixel, entry) pairs is live: measured rare enough that testing for it — a
            //  ballot of mask logic costs two half-rate VALU instructions — does not pay; kept as a build switch)
            if (!wave_ballot(live)) continue;
#endif
            SLS_TRACE_STEP();
            // alpha of the lanes that take part (0: the entry passes through, f = 1)
            const float a = live ? e.alpha : 0.0f;
            const float f = 1.0f - a;
            // transmittance in front of each slot, multiplied up in list order: E = Tr * prod_{k<slot} f_k — every lane
            // forms the three running products (the quad's f broadcast by DPP) and picks its own
            // (the DPP moves must execute in all lanes: never inside a conditional expression)
            const float P1 = Tr * dppq<0x00>(f);
            const float P2 = P1 * dppq<0x55>(f);
            const float P3 = P2 * dppq<0xAA>(f);
            float E = sge1 ? P1 : Tr;
            E = sge2 ? P2 : E;
            E = sge3 ? P3 : E;
            const float I = E * f;
            // the transmittance behind the four entries (a finished pixel has f = 1 in every slot: Tr stays)
            const float I3 = dppq<0xFF>(I);
            bool upd = live;
            float w = a * E;
            // T only falls along a pixel's slots and stays >= T_MIN while the pixel is alive, so some slot of a quad
            // terminates iff the quad's last value is below the threshold: ONE compare on a VGPR feeds the ballot
            if (wave_ballot(I3 < SLS_T_MIN)) {
                // some pixel of the block terminates in this step (at most once per pixel): cut its quad at the
                // first terminating slot
                const bool term = live && (I < SLS_T_MIN);
                const uint64_t tb = wave_ballot(term);
                const uint32_t nib = (uint32_t)(tb >> (lane & 60)) & 15u;   // terminating slots of my pixel
                const bool first_term = term && !(nib & below);
                upd = live && !(nib & upto);
                w = upd ? w : 0.0f;
                cons = first_term ? contributor : cons;
                const float Tt = quad_sum(first_term ? E : 0.0f);          // in front of the first terminating slot
                Tr = nib ? Tt : I3;
                done = done || (nib != 0u);
                if (wave_all(done)) wave_done = true;
            } else {
                Tr = I3;
            }
            if (blk_mask && upd) s_flag[j] = 1u;   // (same value from every lane: plain LDS store)
            // (w = 0 where the lane does not take part and the depth of an evaluated pair is finite: no select needed;
            //  the distortion's 1 / depth below wants a harmless value there)
            const float dep = LEAN ? e.depth : (upd ? e.depth : 1.0f);
            D += dep * w;
            N01 += mk2(q2.x, q2.y) * w; N2 += q2.z * w;
            last = upd ? contributor : last;
            if (!LEAN) {
                // Distortion: sum_i w_i (m_i^2 A_i + M2_i - 2 m_i M1_i) over the exclusive prefixes is the
                // pairwise form sum_{j<i} w_i w_j (m_i - m_j)^2 = A * M2 - M1^2 of the TOTALS (A = sum w = 1 - T):
                // only the two moments are accumulated (per lane), no prefix over the slots, no running term.
                const float m = mscale * (1.0f - cam.near_c * __builtin_amdgcn_rcpf(dep));
                const float mw = m * w;
                M1 += mw;
                M2 += m * mw;
                const bool is_med = upd && (E > 0.5f);
                med = is_med ? dep : med;
                medc = is_med ? contributor : medc;
            }
            if (wave_done) break;
        }
        SLS_PHASE(2);
        if (blk_mask) {
            __builtin_amdgcn_wave_barrier();
            const bool fl = s_flag[lane] != 0u;
            const uint64_t rmask = wave_ballot(fl);
            if (fl) clist[ccnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rmask, 0u))] =
                        make_uint2((uint32_t)(r * 64 + lane), my_idx);
            ccnt += (uint32_t)__builtin_popcountll(rmask);
        }
        SLS_PHASE(3);
    }

    SLS_MARK(0, 3, Tr);
    // combine the four slots of a pixel
    D = quad_sum(D); const float N0 = quad_sum(N01.x), N1 = quad_sum(N01.y); N2 = quad_sum(N2);
    M1 = quad_sum(M1); M2 = quad_sum(M2);
    const float dist = (1.0f - Tr) * M2 - M1 * M1;
    last = quad_max(last);
    const uint32_t medc_q = quad_max(medc);
    med = quad_sum((medc == medc_q && medc != 0u) ? med : 0.0f);
    if (inside && slot == 0) {
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
        pix_contrib[pix] = make_uint2(last, medc_q);
    }
    if (blk_mask && lane == 0) reinterpret_cast<uint32_t *>(blk_mask + 1)[tile * kPerTile + sub] = ccnt;
    // cost of this block in the backward, for its longest-first launch order: steps, a round of 64 entries ~1.3 steps' worth + its latency
    if (block_cost && lane == 0) block_cost[tile * kPerTile + sub] = block_backward_cost(ccnt);
    if (tile_consumed) {   // tile value = max over its pixels (buffer zeroed by the launcher)
        uint32_t c = inside ? (done ? cons : (uint32_t)n) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c = max(c, (uint32_t)__shfl_down(c, off, 64));
        if (lane == 0) atomicMax(&tile_consumed[tile], c);
    }
    SLS_TRACE_END(0);
    SLS_PHASE_END();
    if (DBG && lane == 0) {
        dbg_cycles[tile * kPerTile + sub] = (uint32_t)(clock64() - t_start);
        uint32_t *st = dbg_cycles + (size_t)T * kPerTile;
        atomicAdd(&st[0], st_staged); atomicAdd(&st[1], st_pass);
        atomicAdd(&st[2], st_steps); atomicAdd(&st[3], st_lanes); atomicAdd(&st[4], st_slots); atomicAdd(&st[5], st_geom);
    }
}

// ---------------------------------------------------------------------------
// A6 forward, dense rounds (8x2 blocks, when the tile sort delivered the instances' block masks).  A wave of the
// kernel above spends a third of its life on per-round overhead (wait for the 64 staged records 1230, cull +
// compaction 780, round end 280 shader clocks, against 825 per step) and the tile's list is the TILE's: of the 64
// entries of a round only 25-50 % can reach this block's 16 pixels at all.  Here the list is read twice: a SCAN tests
// 256 entries per round — four coalesced 8-byte loads per lane: the (surfel, block mask) pairs the tile sort's
// scatter stored in list order (sls_sort.hip: block_mask_of, from the surfels' block boxes) — and queues the
// survivors; a ROUND stages, culls and blends 64 queued survivors.  Rounds per block 5.3 -> 3.3 at BASELINE config 3,
// 14.6 -> 5.0 at 170 k surfels / 64 x 1024, 6.2 -> 2.9 at 50 k.  The first 64 entries go straight into round 0, so
// that the first records are requested as early as before.  Everything a pixel
// accumulates is unchanged (same entries, same order, same arithmetic: the mask is the box test the round's cull
// repeats, with 0.01 pixels of slack) and so is the hand-over to the backward: the block's compact list.
// (The scan was first built on gathers of the records' support boxes: 256 scattered line requests per chunk, 4300-4800
//  clocks of waiting per round, slower than the kernel above at every size.)
// ---------------------------------------------------------------------------
constexpr int kScanChunk = 256;      // list entries tested per scan: four consecutive ones per lane
constexpr int kQueueCap = 320;       // ring of survivors waiting for a round: <= 64 before a scan + 256 from it
template <int BW, int BH, bool DBG, bool LEAN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SLS_FWD_WAVES, SLS_FWD_WAVES))) void render_fwd_dense_kernel(uint64_t *__restrict__ blk_mask,
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    float *__restrict__ allmap, float4 *__restrict__ pix_state, uint2 *__restrict__ pix_contrib,
    uint32_t *__restrict__ tile_consumed, uint32_t *__restrict__ dbg_cycles, uint32_t *__restrict__ block_cost,
    const uint2 *__restrict__ bmask)
{
    static_assert(BW == 8 && BH == 2 && kTileW == 16 && kTileH == 16, "the block masks name the 8x2 blocks of a 16x16 tile");
    constexpr int kPerTile = kTilePix / 16, kBX = kTileW / BW;
    __shared__ float4 s_rec[65 * kRec4];          // (record 64: all zeros, pads a round's list to a multiple of four)
    __shared__ uint32_t s_list[64 + 4];           // (list position << 7) | slot in s_rec
    __shared__ uint32_t s_flag[65];
    __shared__ uint32_t s_qpos[kQueueCap], s_qidx[kQueueCap];   // queued survivors: list position, surfel
    __shared__ uint32_t s_rpos[2][64];            // list positions of the entries of the current / the next round
    const uint64_t t_start = DBG ? clock64() : 0;
    SLS_TRACE_BEGIN();
    const int lane = threadIdx.x, slot = lane & 3, p = lane >> 2;
    const int T = cam.GX * cam.GY;
    int tile, sub;
    tile_of_block<kPerTile>(blockIdx.x, T, tile, sub);
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const int x0 = tx * kTileW + (sub % kBX) * BW, y0 = ty * kTileH + (sub / kBX) * BH;
    const int px = x0 + (p % BW), py = y0 + (p / BW);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;

    v2f d01 = mk2(1.0f, 0.0f);
    float d2 = 0.0f;
    if (inside) {
        const float2 c = col_cs[px], r = row_cs[py];
        d01 = mk2(c.x * r.x, c.y * r.x); d2 = r.y;
    }
    const v2f pcr = mk2((float)px, (float)py);
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);
    const uint32_t below = (1u << slot) - 1u, upto = (2u << slot) - 1u;   // quad bits of the earlier slots (and self)
    const bool sge1 = slot >= 1, sge2 = slot >= 2, sge3 = slot >= 3;

    // replicated over the quad: Tr, done.  Per-lane partial sums: D, N*, M1, M2.
    float Tr = 1.0f, M1 = 0.0f, M2 = 0.0f;
    float D = 0.0f, N2 = 0.0f, med = 0.0f;
    v2f N01 = mk2(0.0f, 0.0f);
    uint32_t medc = 0, last = 0, cons = 0;
    bool done = !inside;
    bool wave_done = wave_all(done);
    uint32_t st_staged = 0, st_pass = 0, st_steps = 0, st_lanes = 0, st_slots = 0, st_geom = 0;   // diagnostics only
    uint32_t ccnt = 0;                            // entries of this block's compact list so far
    uint32_t hs_n = 0, hs_l = 0, hs_r = 0, hs_q[4] = { 0, 0, 0, 0 };   // diagnostics only

    const BlockCone cone = make_block_cone(cam, (float)x0 + 0.5f * (float)(BW - 1), (float)y0 + 0.5f * (float)(BH - 1),
                                           0.5f * (float)(BW - 1), 0.5f * (float)(BH - 1));
    if (lane < kRec4) s_rec[64 * kRec4 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s_rpos[0][lane] = (uint32_t)lane;
    // scan state (wave-uniform scalars); ONE chunk of 256 (surfel, mask) pairs is in flight
    const int n_scan = max(n - 64, 0);
    const int nchunks = (n_scan + kScanChunk - 1) / kScanChunk;
    int sc_c = 0, qh = 0, qn = 0;
    // the chunk in flight: surfels and block masks of list entries 64 + 256 c + 4 lane + (0..3)
    uint32_t ci0 = 0, ci1 = 0, ci2 = 0, ci3 = 0, cm0 = 0, cm1 = 0, cm2 = 0, cm3 = 0;
#define SLS_SCAN_POS(c_, e_) (range.x + (uint32_t)min(64 + (c_) * kScanChunk + 4 * lane + (e_), n - 1))
#define SLS_SCAN_LOAD(c_)                                                                                              \
    { const uint2 e0_ = bmask[SLS_SCAN_POS(c_, 0)], e1_ = bmask[SLS_SCAN_POS(c_, 1)], e2_ = bmask[SLS_SCAN_POS(c_, 2)],  \
                  e3_ = bmask[SLS_SCAN_POS(c_, 3)];                                                                    \
      ci0 = e0_.x; cm0 = e0_.y; ci1 = e1_.x; cm1 = e1_.y; ci2 = e2_.x; cm2 = e2_.y; ci3 = e3_.x; cm3 = e3_.y; }
    // (the list is the tile sort's (surfel, block mask) pairs: `vals` is not written in this mode)
#define SLS_EIDX1(i_, first, r_, limit) bmask[(first) + (uint32_t)min((r_) * 64 + ((i_) * 64 + lane) / kRec4, (limit) - 1)].x
    SLS_STAGE_DECL
    if (n > 0 && !wave_done) {
        si0 = SLS_EIDX1(0, range.x, 0, n); si1 = SLS_EIDX1(1, range.x, 0, n); si2 = SLS_EIDX1(2, range.x, 0, n);
        si3 = SLS_EIDX1(3, range.x, 0, n); si4 = SLS_EIDX1(4, range.x, 0, n);
        if (nchunks > 0) { SLS_SCAN_LOAD(0) }
        SLS_WSTAGE_LOAD_REC()
    }
#undef SLS_EIDX1
    if (blk_mask && blockIdx.x == 0 && lane == 0) blk_mask[0] = block_mask_tag(BW);
    // the compact list of this block (the hand-over to the backward) and, for round 0, the surfels of its entries
    uint2 *const clist = blk_mask ? reinterpret_cast<uint2 *>(blk_mask + block_list_entries_word(T, kPerTile))
                                        + ((size_t)range.x * kPerTile + (size_t)sub * (size_t)n) : nullptr;
    __shared__ uint32_t s_ridx[2][64];            // surfels of the entries of the current / the next round
    if (blk_mask && n > 0 && !wave_done) s_ridx[0][lane] = bmask[range.x + (uint32_t)min(lane, n - 1)].x;
    int cur = 0, cur_k = min(n, 64);
    SLS_PHASE_DECL();
    while (cur_k > 0 && !wave_done) {
        float bcx, bcy, bhx, bhy;
        if (!block_active_box<BW, BH>(wave_ballot(!done), x0, y0, bcx, bcy, bhx, bhy)) break;
        SLS_PHASE_RESET();
        SLS_TRACE_ROUND();
        SLS_TRACE_ACTIVE(wave_ballot(!done));
        if (blk_mask) s_flag[lane] = 0u;   // entries of this round that reach at least one pixel of the block
        // single wave: LDS operations complete in program order, no barrier needed
        SLS_WSTAGE_STORE()
        // --- scan: top the queue up from the chunk whose pairs arrived (requested a round ago); goes on
        //     (waiting for its loads) only while the queue is empty
        while (sc_c < nchunks && qn <= 64) {
            const int p0 = 64 + sc_c * kScanChunk + 4 * lane;
            const bool t0 = (p0 + 0 < n) && ((cm0 >> sub) & 1u), t1 = (p0 + 1 < n) && ((cm1 >> sub) & 1u);
            const bool t2 = (p0 + 2 < n) && ((cm2 >> sub) & 1u), t3 = (p0 + 3 < n) && ((cm3 >> sub) & 1u);
            const uint64_t b0 = wave_ballot(t0), b1 = wave_ballot(t1), b2 = wave_ballot(t2), b3 = wave_ballot(t3);
            // list order = lane-major: the survivors of the lower lanes come first
            uint32_t off = (uint32_t)(qh + qn);
            off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, off));
            off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, off));
            off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, off));
            off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b3, off));
#define SLS_SCAN_PUSH(t_, e_, ci_)                                                              \
            if (t_) { const uint32_t o_ = off >= (uint32_t)kQueueCap ? off - kQueueCap : off;   \
                      s_qpos[o_] = (uint32_t)(p0 + (e_)); s_qidx[o_] = ci_; ++off; }
            SLS_SCAN_PUSH(t0, 0, ci0) SLS_SCAN_PUSH(t1, 1, ci1) SLS_SCAN_PUSH(t2, 2, ci2) SLS_SCAN_PUSH(t3, 3, ci3)
#undef SLS_SCAN_PUSH
            qn += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) + __builtin_popcountll(b3);
            ++sc_c;
            if (sc_c < nchunks) { SLS_SCAN_LOAD(sc_c) }
            if (qn > 0) break;
        }
        __builtin_amdgcn_wave_barrier();
        // --- the next round: up to 64 queued entries
        const int nb = cur ^ 1;
        const int nk = min(qn, 64);
        if (nk > 0) {
            const int qa = qh + lane, ql = qa >= kQueueCap ? qa - kQueueCap : qa;
            if (lane < nk) { s_rpos[nb][lane] = s_qpos[ql]; if (blk_mask) s_ridx[nb][lane] = s_qidx[ql]; }
#define SLS_QIDX(i_) s_qidx[(qh + min(((i_) * 64 + lane) / kRec4, nk - 1)) >= kQueueCap ? (qh + min(((i_) * 64 + lane) / kRec4, nk - 1)) - kQueueCap : (qh + min(((i_) * 64 + lane) / kRec4, nk - 1))]
            si0 = SLS_QIDX(0); si1 = SLS_QIDX(1); si2 = SLS_QIDX(2); si3 = SLS_QIDX(3); si4 = SLS_QIDX(4);
#undef SLS_QIDX
            SLS_WSTAGE_LOAD_REC()
            qh += nk; qh = qh >= kQueueCap ? qh - kQueueCap : qh;
            qn -= nk;
        }
        const int cnt = cur_k;
        __builtin_amdgcn_wave_barrier();
        SLS_PHASE(0);
        bool pass = false;
        if (lane < cnt) {
            const float4 c4 = s_rec[lane * kRec4 + 4];
            pass = cull_pass(c4, bcx, bcy, bhx, bhy, wrapW, invW);
            if (pass) {
                const float4 c3 = s_rec[lane * kRec4 + 3];
                pass = !cone_outside(cone, s_rec[lane * kRec4 + 0], s_rec[lane * kRec4 + 1], s_rec[lane * kRec4 + 2], c3) ||
                       disc_reaches(c3, c4, bcx, bcy, bhx, bhy, wrapW, invW);
            }
        }
        const uint64_t mask = wave_ballot(pass);
        const int npass = __builtin_popcountll(mask);
        if (pass) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint32_t)lane | (s_rpos[cur][lane] << 7);
        if (lane < 3) s_list[npass + lane] = 64u;          // pad to a multiple of four with the empty record
        __builtin_amdgcn_wave_barrier();
        SLS_PHASE(1);
        if (DBG) { st_staged += (uint32_t)cnt; st_pass += (uint32_t)npass; }
        for (int k = 0; k < npass; k += 4) {
            const uint32_t lv = s_list[k + slot];
            const int j = (int)(lv & 127u);
            const uint32_t contributor = (lv >> 7) + 1u;
            const float4 *sr = s_rec + __umul24((unsigned)j, (unsigned)kRec4);
            const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], q3 = sr[3], q4 = sr[4];
            Eval e;
            eval_surfel(q0, q1, q2, q3, q4, d01, d2, pcr, wrapW, invW, cam.near_c, e);
            const bool live = !done && !e.skip;
            if (DBG) {
                const uint64_t lb = wave_ballot(live);
                st_steps += 1u; st_lanes += (uint32_t)__builtin_popcountll(lb);
                const uint64_t gb = wave_ballot(inside && !e.skip);      // ignoring finished pixels
                for (int q = 0; q < 4; ++q) {
                    st_slots += (lb & (0x1111111111111111ull << q)) ? 1u : 0u;
                    st_geom += (gb & (0x1111111111111111ull << q)) ? 1u : 0u;
                }
            }
            SLS_TRACE_STEP();
            // alpha of the lanes that take part (0: the entry passes through, f = 1)
            const float a = live ? e.alpha : 0.0f;
            const float f = 1.0f - a;
            // transmittance in front of each slot, multiplied up in list order: E = Tr * prod_{k<slot} f_k — every lane
            // forms the three running products (the quad's f broadcast by DPP) and picks its own
            // (the DPP moves must execute in all lanes: never inside a conditional expression)
            const float P1 = Tr * dppq<0x00>(f);
            const float P2 = P1 * dppq<0x55>(f);
            const float P3 = P2 * dppq<0xAA>(f);
            float E = sge1 ? P1 : Tr;
            E = sge2 ? P2 : E;
            E = sge3 ? P3 : E;
            const float I = E * f;
            // the transmittance behind the four entries (a finished pixel has f = 1 in every slot: Tr stays)
            const float I3 = dppq<0xFF>(I);
            bool upd = live;
            float w = a * E;
            // T only falls along a pixel's slots and stays >= T_MIN while the pixel is alive, so some slot of a quad
            // terminates iff the quad's last value is below the threshold: ONE compare on a VGPR feeds the ballot
            if (wave_ballot(I3 < SLS_T_MIN)) {
                // some pixel of the block terminates in this step (at most once per pixel): cut its quad at the
                // first terminating slot
                const bool term = live && (I < SLS_T_MIN);
                const uint64_t tb = wave_ballot(term);
                const uint32_t nib = (uint32_t)(tb >> (lane & 60)) & 15u;   // terminating slots of my pixel
                const bool first_term = term && !(nib & below);
                upd = live && !(nib & upto);
                w = upd ? w : 0.0f;
                cons = first_term ? contributor : cons;
                const float Tt = quad_sum(first_term ? E : 0.0f);          // in front of the first terminating slot
                Tr = nib ? Tt : I3;
                done = done || (nib != 0u);
                if (wave_all(done)) wave_done = true;
            } else {
                Tr = I3;
            }
            if (blk_mask && upd) s_flag[j] = 1u;   // (same value from every lane: plain LDS store)
            if (DBG) {     // what finer blocks would save: entries that reach only one 4x2 half / one 2x2 quarter of the block
                const uint64_t ub = wave_ballot(upd);
                for (int q = 0; q < 4; ++q) {
                    const uint64_t m = (ub >> q) & 0x1111111111111111ull;
                    if (!m) continue;
                    ++hs_n;
                    hs_l += (m & 0x0000FFFF0000FFFFull) ? 1u : 0u; hs_r += (m & 0xFFFF0000FFFF0000ull) ? 1u : 0u;
                    for (int c = 0; c < 4; ++c) hs_q[c] += (m & (0x000000FF000000FFull << (8 * c))) ? 1u : 0u;
                }
            }
            // (w = 0 where the lane does not take part and the depth of an evaluated pair is finite: no select needed;
            //  the distortion's 1 / depth below wants a harmless value there)
            const float dep = LEAN ? e.depth : (upd ? e.depth : 1.0f);
            D += dep * w;
            N01 += mk2(q2.x, q2.y) * w; N2 += q2.z * w;
            last = upd ? contributor : last;
            if (!LEAN) {
                // Distortion: sum_i w_i (m_i^2 A_i + M2_i - 2 m_i M1_i) over the exclusive prefixes is the
                // pairwise form sum_{j<i} w_i w_j (m_i - m_j)^2 = A * M2 - M1^2 of the TOTALS (A = sum w = 1 - T):
                // only the two moments are accumulated (per lane), no prefix over the slots, no running term.
                const float m = mscale * (1.0f - cam.near_c * __builtin_amdgcn_rcpf(dep));
                const float mw = m * w;
                M1 += mw;
                M2 += m * mw;
                const bool is_med = upd && (E > 0.5f);
                med = is_med ? dep : med;
                medc = is_med ? contributor : medc;
            }
            if (wave_done) break;
        }
        SLS_PHASE(2);
        if (blk_mask) {
            __builtin_amdgcn_wave_barrier();
            const bool fl = lane < cnt && s_flag[lane] != 0u;
            const uint64_t rmask = wave_ballot(fl);
            if (fl) clist[ccnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rmask, 0u))] =
                        make_uint2(s_rpos[cur][lane], s_ridx[cur][lane]);
            ccnt += (uint32_t)__builtin_popcountll(rmask);
        }
        cur = nb; cur_k = nk;
        SLS_PHASE(3);
    }
#undef SLS_SCAN_POS
#undef SLS_SCAN_LOAD

    // combine the four slots of a pixel
    D = quad_sum(D); const float N0 = quad_sum(N01.x), N1 = quad_sum(N01.y); N2 = quad_sum(N2);
    M1 = quad_sum(M1); M2 = quad_sum(M2);
    const float dist = (1.0f - Tr) * M2 - M1 * M1;
    last = quad_max(last);
    const uint32_t medc_q = quad_max(medc);
    med = quad_sum((medc == medc_q && medc != 0u) ? med : 0.0f);
    if (inside && slot == 0) {
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
        pix_contrib[pix] = make_uint2(last, medc_q);
    }
    if (blk_mask && lane == 0) reinterpret_cast<uint32_t *>(blk_mask + 1)[tile * kPerTile + sub] = ccnt;
    // cost of this block in the backward, for its longest-first launch order: steps, a round of 64 entries ~1.3 steps' worth + its latency
    if (block_cost && lane == 0) block_cost[tile * kPerTile + sub] = block_backward_cost(ccnt);
    if (tile_consumed) {   // tile value = max over its pixels (buffer zeroed by the launcher)
        uint32_t c = inside ? (done ? cons : (uint32_t)n) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c = max(c, (uint32_t)__shfl_down(c, off, 64));
        if (lane == 0) atomicMax(&tile_consumed[tile], c);
    }
    SLS_TRACE_END(0);
    SLS_PHASE_END();
    if (DBG && lane == 0) {
        dbg_cycles[tile * kPerTile + sub] = (uint32_t)(clock64() - t_start);
        uint32_t *st = dbg_cycles + (size_t)T * kPerTile;
        atomicAdd(&st[0], st_staged); atomicAdd(&st[1], st_pass);
        atomicAdd(&st[2], st_steps); atomicAdd(&st[3], st_lanes); atomicAdd(&st[4], st_slots); atomicAdd(&st[5], st_geom);
        const uint32_t qmax = max(max(hs_q[0], hs_q[1]), max(hs_q[2], hs_q[3]));
        atomicAdd(&st[6], hs_n); atomicAdd(&st[7], hs_l + hs_r); atomicAdd(&st[8], hs_q[0] + hs_q[1] + hs_q[2] + hs_q[3]);
        atomicAdd(&st[9], (hs_n + 3u) / 4u); atomicAdd(&st[10], (max(hs_l, hs_r) + 3u) / 4u); atomicAdd(&st[11], (qmax + 3u) / 4u);
        atomicMax(&st[12], (hs_n + 3u) / 4u); atomicMax(&st[13], (max(hs_l, hs_r) + 3u) / 4u); atomicMax(&st[14], (qmax + 3u) / 4u);
    }
}

// ---------------------------------------------------------------------------
// A7 backward.  Back to front, four list entries per step: slot 0 holds the
// LAST entry of the four.  T_i = T_{i+1} / (1 - alpha_i) is multiplied up in
// that order inside the quad, S (the suffix sum of w g) is an exclusive quad
// prefix.  The 16 gradient fields are reduced over the 16 pixels of a slot with
// block_reduce16_pk and flushed with one 64-lane global float atomic per step.
// ---------------------------------------------------------------------------
// LEAN: the caller guarantees dL/d(median) = dL/d(distortion) = 0 (the mapper's loss at
// depth_ratio = 0, gaussian_renderer/__init__.py:79-86): their terms are compiled out.
// FUSED = 1 (sls_mapping_step, LEAN only): dL/dallmap is not read but computed per pixel from the consumer's
// kernel-B planes (sls_consumer_dev.hpp) — consumer kernel C is not launched; block 0 also turns kernel B's
// per-block loss partials into the iteration's loss sums.
// FUSED = 2: kernel B is not launched either.  The block runs kernel B's per-pixel work itself for its 16 pixels and
// the 2 (BW + BH) pixels around them — 36 lanes of the wave, one pixel each, the same device function — hands the
// pieces over in LDS, and leaves its pixels' three loss terms in ca.partials (one array per term, summed by preprocess_bwd).
// Same gradients bit for bit: one dependent launch and the three planes' round trip less.
// DET (deterministic accumulation, SlsMappingConfig.deterministic / sls_backward_det): float atomics add in an
// order that changes from run to run, so the last bits of the gradients do too.  Integer atomics commute:
//   DET = 1: first launch — per (surfel, field) the LARGEST |contribution| (atomicMax on the float's bit
//            pattern, order-independent), into det_max;
//   DET = 2: second launch — every contribution is scaled by 2^(39 - unbiased exponent of that maximum) (=
//            2^(166 - biased exponent)), rounded to an integer (a pure function of the contribution) and added
//            with a 64-bit integer atomic into det_acc: 39 fractional bits below the largest term, |q| < 2^40,
//            23 bits of headroom for the sum.  preprocess_bwd scales back by 2^(biased exponent - 166).
//   DET = 3: ONE launch (DENSE only) — the scale of a (surfel, field) is predicted instead of measured: the byte
//            det_prev[surfel][field] (biased exponent of the field's sum in the keyframe's previous iteration + 12,
//            written by preprocess_bwd) or, without history, the field's default det_gex[field].  The exponent bytes of
//            a round's 64 surfels are fetched a round ahead (16 bytes per entry) into LDS.  A contribution whose
//            fixed-point value reaches 2^50 — the prediction was off by more than 2^22 — sets bit 3 of the iteration's
//            overflow word: the iteration is void and the caller repeats it with the two launches above.
// Same kernel otherwise: the result does not depend on the order of the blocks or of the atomics.
// SLS_ABL_EXTRA (experiment builds only, tools/build_variant.sh ... -DSLS_ABL_EXTRA=k): behind the engine's tile backward a
// SECOND launch of the same kernel, timed in a slot of its own ("knn" in bench.py's kernel table), with one ingredient
// changed — k = 1: no atomics (what they cost), k = 2: unchanged, its gradient records into a scratch copy (the control:
// the production launch's time under the same bracketing).  The iteration's results are not touched.
#ifdef SLS_ABL_EXTRA
#define SLS_ABL_PARAM , int abl
#define SLS_ABL_ARG(v_) , v_
#else
#define SLS_ABL_PARAM
#define SLS_ABL_ARG(v_)
#endif
#ifdef SLS_BWD_WAVES      // (experiment builds: a register cap for that many waves per SIMD; HISTORY #75)
#define SLS_BWD_OCC __attribute__((amdgpu_waves_per_eu(SLS_BWD_WAVES, SLS_BWD_WAVES)))
#else
#define SLS_BWD_OCC
#endif
template <int BW, int BH, bool LEAN, int FUSED, int DET, bool DENSE>
__global__ __launch_bounds__(64) SLS_BWD_OCC void render_bwd_block_kernel(
    DevCam cam, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ vals,
    const float4 *__restrict__ rec, const float2 *__restrict__ col_cs, const float2 *__restrict__ row_cs,
    const float4 *__restrict__ pix_state, const uint2 *__restrict__ pix_contrib,
    const float *__restrict__ dL_dallmap, float *__restrict__ grec, const uint64_t *__restrict__ blk_mask,
    uint8_t *__restrict__ touched, uint32_t *__restrict__ dbg_cycles, ConsumerArgs ca, int consumer_blocks,
    uint32_t *__restrict__ det_max, unsigned long long *__restrict__ det_acc, const uint32_t *__restrict__ block_order,
    int vstride, const uint8_t *__restrict__ det_prev, const uint32_t *__restrict__ det_gex, uint32_t *__restrict__ det_flag,
    uint32_t order_tag SLS_ABL_PARAM)
{
    static_assert(DET != 3 || DENSE, "the one-pass deterministic accumulation walks the forward's compact lists");
    static_assert(!FUSED || LEAN, "the fused consumer gradient has no median / distortion channel");
    SLS_TRACE_BEGIN();
    if (FUSED == 1 && blockIdx.x == 0) consumer_reduce_partials_wave(ca, consumer_blocks, (int)threadIdx.x);
    static_assert(BW * BH == 16 && kTileW % BW == 0 && kTileH % BH == 0, "16-pixel blocks tiling a tile");
    constexpr int kPerTile = kTilePix / 16, kBX = kTileW / BW;
    // (record 64: all zeros, never active — pads the compacted list to a multiple of four, as in the forward)
    __shared__ float4 s_rec[65 * kRec4];
    __shared__ uint32_t s_list[64 + 4];          // culling path: the round's survivors; DENSE: its entries' list positions
    __shared__ uint32_t s_gidx[65];
    __shared__ uint4 s_ex[DET == 3 ? 65 : 1];    // DET = 3: the predicted-scale bytes (16 fields) of the round's entries
    // DENSE: blk_mask is the compact-list hand-over of a forward with the same block shape (the launcher checks what
    // it can, the tag settles it: another producer's buffer is not walked)
    const int T = cam.GX * cam.GY;
    // The three words that decide where this block works — the hand-over's tag, the launch order's tag (order_tag: the
    // order is the keyframe's own buffer, walked only once an iteration has filled it; it has its full size either
    // way) and the block's entry in it — are requested TOGETHER: one memory round trip at the head of the block, not
    // three one after the other.
    const int oxcd = blockIdx.x % 8;
    const uint64_t tagw = DENSE ? blk_mask[0] : 0ull;
    // (unconditional loads — from `ranges`, always there, where there is no order: a load under a condition is a
    //  branch with a wait of its own)
    const uint32_t *ob = block_order ? block_order : reinterpret_cast<const uint32_t *>(ranges);
    const uint32_t otag_w = ob[0];
    const uint32_t oidx = ob[block_order ? (order_tag ? 1 : 0) + oxcd * (T * kPerTile / 8) + (int)blockIdx.x / 8 : 0];
    const uint32_t otag = (block_order && order_tag) ? otag_w : order_tag;
    if (DENSE && tagw != block_mask_tag(BW)) {
        // the hand-over was written by a forward of another block shape (debug variants switched between the two
        // launches): nothing here can be trusted — say so (bit 4 of the iteration's void word: the engine raises) instead
        // of dropping the block's gradients and loss terms silently
        if (det_flag && threadIdx.x == 0) atomicOr(det_flag, 16u);
        return;
    }
    const uint64_t t_start = dbg_cycles ? clock64() : 0;
    const int lane = threadIdx.x, slot = lane & 3, p = lane >> 2;
    int tile, sub;
    if (block_order && otag == order_tag) {
        // the blocks of this XCD, most expensive first (bwd_block_order_kernel): b -> (XCD b % 8, rank b / 8)
        const int xcd = oxcd;
        const int i = (int)oidx;
        const int ts = i / kPerTile;
        tile = ((ts >> 2) * 8 + xcd) * 4 + (ts & 3);
        sub = i % kPerTile;
    } else {
        tile_of_block<kPerTile>(blockIdx.x, T, tile, sub);
    }
    const int ty = tile / cam.GX, tx = tile - ty * cam.GX;
    SLS_MARK(1, 0, tile);
    const uint2 range = ranges[tile];
    const int x0 = tx * kTileW + (sub % kBX) * BW, y0 = ty * kTileH + (sub / kBX) * BH;
    const int px = x0 + (p % BW), py = y0 + (p / BW);
    const bool inside = (px < cam.W) && (py < cam.H);
    const float wrapW = cam.wrap ? (float)cam.W : 0.0f, invW = cam.wrap ? 1.0f / (float)cam.W : 0.0f;
    const v2f pcr = mk2((float)px, (float)py);
    const float mscale = cam.far_c / (cam.far_c - cam.near_c);
    const float k1 = slot >= 1 ? 1.0f : 0.0f, k2 = slot >= 2 ? 1.0f : 0.0f, k3 = slot >= 3 ? 1.0f : 0.0f;

    v2f d01 = mk2(1.0f, 0.0f), dN01 = mk2(0.0f, 0.0f);
    float d2 = 0.0f;
    uint32_t last = 0, medc = 0;
    float Tf = 1.0f, M1 = 0.0f, M2 = 0.0f;
    float dD = 0, dA = 0, dN2 = 0, dMed = 0, dDist = 0;
    // the 16 gradient fields are reduced in an order that keeps pairs adjacent (below): position p of the
    // reduction -> field of the gradient record
    const int field = p < 8 ? (int)((0x73625410u >> (4 * p)) & 15u) : p;
    const uint32_t gex_field = DET == 3 ? det_gex[field] : 0u;          // the field's default scale (no history)
    constexpr int kRing = 16 + 2 * BW + 2 * BH;
    static_assert(kRing <= 64, "one lane per pixel of the block and its ring");
    __shared__ float4 s_cb[FUSED == 2 ? 3 * kRing : 1];
    ConsumerOwn cown;
    if (FUSED == 2) {
        static_assert(FUSED != 2 || (BW == 8 && BH == 2), "the inline stage's lane maps are written out for 8x2 blocks");
        // (the pixel's own inputs of the gradient below: requested first, in flight during the stage)
        cown = consumer_own_load(ca, min(py, cam.H - 1), min(px, cam.W - 1));
        // stage 2 (below) = kernel B's pieces for the block's pixels [0, 16) and the ring around them — the row above, the
        // row below, the column left, the column right — one pixel per lane; its own inputs are requested here, before
        // stage 1 computes
        int qr, qc;
        if (lane < 16) { qr = y0 + lane / BW; qc = x0 + lane % BW; }
        else if (lane < 16 + BW) { qr = y0 - 1; qc = x0 + lane - 16; }
        else if (lane < 16 + 2 * BW) { qr = y0 + BH; qc = x0 + lane - 16 - BW; }
        else if (lane < 16 + 2 * BW + BH) { qr = y0 + lane - 16 - 2 * BW; qc = x0 - 1; }
        else { qr = y0 + lane - 16 - 2 * BW - BH; qc = x0 + BW; }
        const bool has = lane < kRing && qr >= 0 && qr < cam.H && qc >= 0 && qc < cam.W;
        uint32_t bvalid = 0u;      // (the byte as loaded: a comparison here would wait for it before stage 1's loads leave)
        float bal = 0.0f, bN0 = 0.0f, bN1 = 0.0f, bN2 = 0.0f, bgt = 0.0f;
        if (has) {
            const size_t P = (size_t)cam.H * cam.W, pix = (size_t)qr * cam.W + qc;
            bvalid = ca.valid[pix];
            bal = ca.allmap[SLS_CH_ALPHA * P + pix];
            bN0 = ca.allmap[(SLS_CH_NORMAL + 0) * P + pix];
            bN1 = ca.allmap[(SLS_CH_NORMAL + 1) * P + pix];
            bN2 = ca.allmap[(SLS_CH_NORMAL + 2) * P + pix];
            bgt = ca.gt_depth[pix];
        }
        // stage 1: the surface points of the 60 pixels within two steps of the block — rows y0-2 .. y0+3 with 8, 10, 12,
        // 12, 10, 8 pixels — one per lane, into a 6 x 12 grid in LDS (a point is needed by up to four stencils: computed
        // once, with its division, instead of once per stencil)
        __shared__ float4 s_pt[6 * 12];
        {
            int dr, dc;
            if (lane < 8) { dr = -2; dc = lane; }
            else if (lane < 18) { dr = -1; dc = lane - 9; }
            else if (lane < 30) { dr = 0; dc = lane - 20; }
            else if (lane < 42) { dr = 1; dc = lane - 32; }
            else if (lane < 52) { dr = 2; dc = lane - 43; }
            else { dr = 3; dc = lane - 52; }
            const int r = y0 + dr, c = x0 + dc;
            float4 pt = make_float4(0, 0, 0, 0);
            if (lane < 60 && r >= 0 && r < cam.H && c >= 0 && c < cam.W) {
                float sd;
                const float3 pp = surf_point(ca, r, c, sd);
                pt = make_float4(pp.x, pp.y, pp.z, sd);
            }
            if (lane < 60) s_pt[(dr + 2) * 12 + dc + 2] = pt;
        }
        // (keeps the byte a register until here: the compiler otherwise compares it where it is loaded — and waits for
        //  it, a full round trip, before stage 1's loads are issued)
        asm volatile("" : "+v"(bvalid));
        __syncthreads();
        float4 bu = make_float4(0, 0, 0, 0), bv = bu, bn = bu;
        float lg = 0.0f, ln = 0.0f, la = 0.0f;
        if (has) {
            const int pi = (qr - y0 + 2) * 12 + (qc - x0 + 2);
            const float4 po = s_pt[pi], pu = s_pt[pi + 12], pd = s_pt[pi - 12], pr = s_pt[pi + 1], pl = s_pt[pi - 1];
            const bool interior = (qr > 0) && (qr < cam.H - 1) && (qc > 0) && (qc < cam.W - 1);
            consumer_b_core(ca, bvalid == 1u, bal, bN0, bN1, bN2, po.w, bgt, interior, make_float3(pu.x, pu.y, pu.z),
                            make_float3(pd.x, pd.y, pd.z), make_float3(pr.x, pr.y, pr.z), make_float3(pl.x, pl.y, pl.z),
                            bu, bv, bn, lg, ln, la);
        }
        if (lane < kRing) { s_cb[lane] = bu; s_cb[kRing + lane] = bv; s_cb[2 * kRing + lane] = bn; }
        if (lane >= 16) { lg = 0.0f; ln = 0.0f; la = 0.0f; }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            lg += __shfl_down(lg, off, 64); ln += __shfl_down(ln, off, 64); la += __shfl_down(la, off, 64);
        }
        if (lane == 0) {      // (three arrays, one per term: preprocess_bwd sums them with 16-byte loads)
            const size_t nb = (size_t)T * kPerTile, blk = (size_t)(tile * kPerTile + sub);
            ca.partials[blk] = lg; ca.partials[nb + blk] = ln; ca.partials[2 * nb + blk] = la;
        }
        __syncthreads();
    }
    if (inside) {
        const float2 c = col_cs[px], r = row_cs[py];
        d01 = mk2(c.x * r.x, c.y * r.x); d2 = r.y;
        const size_t P = (size_t)cam.H * cam.W;
        const size_t pix = (size_t)py * cam.W + px;
        const uint2 pcn = pix_contrib[pix];
        last = pcn.x; medc = pcn.y;
        const float4 ps = pix_state[pix];
        Tf = ps.x; M1 = ps.y; M2 = ps.z;
        if (FUSED) {
            float gpix[7];
            if (FUSED == 2) {
                const int pr = p / BW, pc = p % BW;
                const int iu = pr > 0 ? p - BW : 16 + pc, id = pr < BH - 1 ? p + BW : 16 + BW + pc;
                const int il = pc > 0 ? p - 1 : 16 + 2 * BW + pr, ir = pc < BW - 1 ? p + 1 : 16 + 2 * BW + BH + pr;
                consumer_pixel_grad_core(ca, py, px, cown, s_cb[iu], s_cb[id], s_cb[kRing + il], s_cb[kRing + ir], s_cb[2 * kRing + p], gpix);
            } else {
                consumer_pixel_grad(ca, py, px, gpix);
            }
            dD = gpix[0]; dA = gpix[1]; dN01 = mk2(gpix[2], gpix[3]); dN2 = gpix[4];
        } else {
            dD = dL_dallmap[SLS_CH_DEPTH * P + pix];
            dA = dL_dallmap[SLS_CH_ALPHA * P + pix];
            dN01 = mk2(dL_dallmap[(SLS_CH_NORMAL + 0) * P + pix], dL_dallmap[(SLS_CH_NORMAL + 1) * P + pix]);
            dN2 = dL_dallmap[(SLS_CH_NORMAL + 2) * P + pix];
        }
        if (!LEAN) {
            dMed = dL_dallmap[SLS_CH_MEDIAN * P + pix];
            dDist = dL_dallmap[SLS_CH_DIST * P + pix];
        }
    }
    const float Af = 1.0f - Tf;
    uint32_t wmax = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor(wmax, off, 64));
    const int tmax = (int)wmax;
    SLS_MARK(1, 1, tmax);
    SLS_MARK(1, 2, dD + dA + dN2);
    float Tr = Tf, S = 0.0f;   // replicated over the quad
    // One step: the list entries in slots j (LDS record slot of my quad lane; 64 = the empty padding record) with
    // contributor numbers `contributor`, back to front: slot 0 of a quad holds the LAST entry of the four.
    // (two halves: step_sums does everything up to the reduce-scatter — the chain over Tr and S runs through it — and
    //  step_flush the atomics.  Two steps per loop iteration with both flushes behind them — one basic block of 252 VALU
    //  instructions, 112 registers — changed nothing at any size: the compiler schedules them one after the other, HISTORY #77)
    auto step_sums = [&](const int j, const uint32_t contributor) -> float {
        const float4 *sr = s_rec + __umul24((unsigned)j, (unsigned)kRec4);
        const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], q3 = sr[3], q4 = sr[4];
        Eval e;
        eval_surfel(q0, q1, q2, q3, q4, d01, d2, pcr, wrapW, invW, cam.near_c, e);
        // (bitwise: three lane masks ANDed, no short-circuit control flow in the step; the compact list holds only
        //  entries that reached a pixel of this block, and on the culling path a step whose 64 pairs are all inactive
        //  is rare — no dead-step test: its ballot costs more than it saves, as in the forward)
        const bool act = inside & (contributor <= last) & !e.skip;
        SLS_TRACE_STEP();
        const float a = act ? e.alpha : 0.0f;          // 0: the entry passes through (1 - a = 1, w = 0)
        const float om = 1.0f - a;
        const float rom = __builtin_amdgcn_rcpf(om);
        // T in front of each entry: Ti = Tr * prod_{slots <= mine} rom, multiplied up in slot order
        // (the DPP moves must execute in all lanes: never inside a conditional expression)
        float Ti = Tr * rom, sh;
        sh = dppq<kQuadShr1>(Ti) * rom; Ti = slot >= 1 ? sh : Ti;
        sh = dppq<kQuadShr1>(Ti) * rom; Ti = slot >= 2 ? sh : Ti;
        sh = dppq<kQuadShr1>(Ti) * rom; Ti = slot >= 3 ? sh : Ti;
        Tr = dppq<0xFF>(Ti);
        const float w = a * Ti;
        // (the depth of an evaluated pair is finite and meets w = 0 where the lane is inactive; the
        //  distortion's 1 / depth wants a harmless value there)
        const float dep = LEAN ? e.depth : (act ? e.depth : 1.0f);
        float gdist = 0.0f, ddist = 0.0f;     // distortion terms of g_k and of dL/ddepth
        if (!LEAN) {
            const float rdep = __builtin_amdgcn_rcpf(dep);
            const float m = mscale * (1.0f - cam.near_c * rdep);
            const float dm_dd = mscale * cam.near_c * rdep * rdep;
            gdist = dDist * (M2 + m * m * Af - 2.0f * m * M1);
            ddist = dDist * 2.0f * (m * Af - M1) * dm_dd;
        }
        float gk = dD * dep + (dN01.x * q2.x + dN01.y * q2.y + dN2 * q2.z) + dA;
        if (!LEAN) gk += gdist;             // (x + 0.0f is not x for the compiler: an instruction of its own)
        float Se, St;
        quad_excl_total(w * gk, k1, k2, k3, Se, St);
        const float dL_dalpha = act ? Ti * gk - (S + Se) * rom : 0.0f;
        S += St;
        float dL_ddepth = w * dD;
        if (!LEAN) {
            dL_ddepth += w * ddist;
            dL_ddepth += (act && contributor == medc) ? dMed : 0.0f;
        }
        const bool unclamped = e.og < SLS_ALPHA_MAX;
        const float dL_do = unclamped ? dL_dalpha * e.G : 0.0f;
        const float dL_drho = unclamped ? -0.5f * e.G * dL_dalpha * q2.w : 0.0f;
        const bool a3 = act && e.use3d, a2 = act && !e.use3d;
        const v2f dL_duv = (dL_drho * 2.0f) * e.uv;                     // dL/d(u, v)
        const v2f dL_dhuv0 = dL_duv * e.rinv;
        const v2f dL_dhuv = mk2(a3 ? dL_dhuv0.x : 0.0f, a3 ? dL_dhuv0.y : 0.0f);   // dL/d(hu, hv)
        const float dL_drinv = dL_duv.x * e.huv.x + dL_duv.y * e.huv.y + dL_ddepth * q0.w;
        const float dL_dnd = a3 ? -dL_drinv * e.rinv * e.rinv : 0.0f;
        const float lp = a2 ? -dL_drho * (2.0f * SLS_FILTER_INV_SQUARE) : 0.0f;
        // fields of the gradient record, in the order `field` names: pairs that one packed instruction makes
        v2f gl[kGrec / 2];
        gl[0] = dL_dhuv.x * e.dl01;                                     // fields 0, 1
        gl[1] = dL_dhuv.y * e.dl01;                                     // fields 4, 5
        gl[2] = dL_dhuv * e.dl2;                                        // fields 2, 6
        gl[3] = mk2(a3 ? dL_ddepth * e.rinv : 0.0f, a2 ? dL_ddepth : 0.0f);   // fields 3, 7
        gl[4] = w * dN01 + dL_dnd * d01;                                // fields 8, 9
        gl[5] = mk2(w * dN2 + dL_dnd * d2, dL_do);                      // fields 10, 11
        gl[6] = dL_dhuv;                                                // fields 12, 13
        gl[7] = lp * e.dxy;                                             // fields 14, 15
        return block_reduce16_pk(gl, lane);  // field `field` of the surfel in my slot
    };
    auto step_flush = [&](const int j, const float tot) {
        const uint32_t gidx = s_gidx[j];
        // (the padding entry is never active: its sums are exact zeros)
        if (DET == 0) {
#ifdef SLS_ABL_EXTRA
            if (abl == 1) asm volatile("" :: "v"(tot), "v"(gidx));
            else
#endif
            if (tot != 0.0f) atomicAdd(&grec[(size_t)gidx * kGrec + field], tot);
        } else if (DET == 1) {
            if (tot != 0.0f) atomicMax(&det_max[(size_t)gidx * kGrec + field], __float_as_uint(fabsf(tot)));
        } else if (DET == 2) {
            if (tot != 0.0f) {
                const int ex = (int)((det_max[(size_t)gidx * kGrec + field] >> 23) & 0xFFu);   // |tot| < 2^(ex - 126)
                const long long q = __float2ll_rn(ldexpf(tot, 166 - ex));                   // |q| < 2^40
                atomicAdd(&det_acc[(size_t)gidx * kGrec + field], (unsigned long long)q);
            }
        } else if (tot != 0.0f) {
            const uint32_t pb = reinterpret_cast<const uint8_t *>(s_ex)[16 * j + field];
            const int ex = det_scale_exp(pb, gex_field);
            const float scaled = ldexpf(tot, 166 - ex);
            // (|q| < 2^50: 13 bits of headroom for the sum; beyond — or without any scale — the iteration is void)
            if (!(fabsf(scaled) < 1125899906842624.0f)) atomicOr(det_flag, kDetMispredicted);
            else atomicAdd(&det_acc[(size_t)gidx * kGrec + field], (unsigned long long)__float2ll_rn(scaled));
        }
    };
    auto blend_step = [&](const int j, const uint32_t contributor) { step_flush(j, step_sums(j, contributor)); };
    if (DENSE) {
        // ---- rounds of 64 entries of the forward's compact list (every one of them reached a pixel of this block)
        const int n = (int)(range.y - range.x);
        const uint32_t cc = tmax > 0 ? reinterpret_cast<const uint32_t *>(blk_mask + 1)[tile * kPerTile + sub] : 0u;
        const uint2 *const clist = reinterpret_cast<const uint2 *>(blk_mask + block_list_entries_word(T, kPerTile))
                                   + ((size_t)range.x * kPerTile + (size_t)sub * (size_t)n);
        if (cc > 0u) {
            const int nr = (int)((cc + 63u) / 64u);
            if (lane < kRec4) s_rec[64 * kRec4 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            SLS_STAGE_DECL
            // The round's entries are STAGED in descending list order (LDS slot j = the round's entry cnt - 1 - j): a step
            // takes slots k .. k + 3 as they lie — no index arithmetic or padding test in the step.  Only the first
            // round processed (the list's tail) can be short: its three slots behind the last entry are cleared.
#define SLS_CENTRY(r_, e_) ((uint32_t)((r_) * 64 + max((int)min(64u, cc - (uint32_t)((r_) * 64)) - 1 - (e_), 0)))
#define SLS_CIDX1(i_, r_) clist[SLS_CENTRY(r_, ((i_) * 64 + lane) / kRec4)].y
#define SLS_CSTAGE_LOAD_IDX(r_) si0 = SLS_CIDX1(0, r_); si1 = SLS_CIDX1(1, r_); si2 = SLS_CIDX1(2, r_); si3 = SLS_CIDX1(3, r_); si4 = SLS_CIDX1(4, r_);
            SLS_CSTAGE_LOAD_IDX(nr - 1)
            uint2 mine_next = clist[SLS_CENTRY(nr - 1, lane)];      // (list position, surfel) of the entry in slot `lane`
            SLS_WSTAGE_LOAD_REC()
            if (nr > 1) { SLS_CSTAGE_LOAD_IDX(nr - 2) }
            // DET = 3: the entries' predicted-scale bytes travel one round ahead of their use, like the records: the entry
            // of slot `lane` two rounds ahead (mine_next2) names the surfel whose 16 bytes are requested a round ahead
            uint2 mine_next2 = make_uint2(0u, 0u);
            uint4 ex_next = make_uint4(0u, 0u, 0u, 0u);
            if (DET == 3) {
                ex_next = reinterpret_cast<const uint4 *>(det_prev)[mine_next.y];
                if (nr > 1) mine_next2 = clist[SLS_CENTRY(nr - 2, lane)];
                if (lane == 0) s_ex[64] = make_uint4(0u, 0u, 0u, 0u);
            }
            SLS_PHASE_DECL();
            for (int r = nr - 1; r >= 0; --r) {
                SLS_PHASE_RESET();
                SLS_WSTAGE_STORE()
                if (r == nr - 1) SLS_MARK(1, 3, sp4.x);
                SLS_TRACE_ROUND();
                const uint2 mine = mine_next;
                s_gidx[lane] = mine.y;
                s_list[lane] = mine.x + 1u;              // (the contributor numbers of the round's entries)
                if (DET == 3) s_ex[lane] = ex_next;
                if (r > 0) {
                    SLS_WSTAGE_LOAD_REC()
                    if (DET == 3) {
                        mine_next = mine_next2;
                        ex_next = reinterpret_cast<const uint4 *>(det_prev)[mine_next.y];
                        if (r > 1) mine_next2 = clist[SLS_CENTRY(r - 2, lane)];
                    } else {
                        mine_next = clist[SLS_CENTRY(r - 1, lane)];
                    }
                    if (r > 1) { SLS_CSTAGE_LOAD_IDX(r - 2) }
                }
                const int cnt = (int)min(64u, cc - (uint32_t)(r * 64));
                if (lane < 3 * kRec4 && cnt * kRec4 + lane < 64 * kRec4) s_rec[cnt * kRec4 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                // only surfels listed here can have a non-zero gradient record: preprocess_bwd reads (and clears)
                // the records of the others not at all.  One byte store per entry; same value from every block.
                if (touched && lane < cnt) touched[mine.y] = 1;
                __builtin_amdgcn_wave_barrier();
                SLS_PHASE(0);
                SLS_PHASE(1);
                for (int k = 0; k < cnt; k += 4) blend_step(k + slot, s_list[k + slot]);
                SLS_PHASE(2);
            }
#undef SLS_CENTRY
#undef SLS_CIDX1
#undef SLS_CSTAGE_LOAD_IDX
            SLS_PHASE_END_B();
        }
    } else if (tmax > 0) {
        // ---- no list from a forward of this block shape: rounds of 64 consecutive list entries, culled here
        // (the list: surfel indices `vstride` words apart — 2 where the tile sort delivered (surfel, block mask) pairs)
        const int nr = (tmax + 63) / 64;
        if (lane < kRec4) s_rec[64 * kRec4 + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane == 0) s_gidx[64] = 0u;
        SLS_STAGE_DECL
#define SLS_SIDX1(i_, r_) vals[(size_t)(range.x + (uint32_t)min((r_) * 64 + ((i_) * 64 + lane) / kRec4, tmax - 1)) * (size_t)vstride]
#define SLS_SSTAGE_LOAD_IDX(r_) si0 = SLS_SIDX1(0, r_); si1 = SLS_SIDX1(1, r_); si2 = SLS_SIDX1(2, r_); si3 = SLS_SIDX1(3, r_); si4 = SLS_SIDX1(4, r_);
        SLS_SSTAGE_LOAD_IDX(nr - 1)
        SLS_WSTAGE_LOAD_REC()
        uint32_t next_idx = vals[(size_t)(range.x + (uint32_t)min((nr - 1) * 64 + lane, tmax - 1)) * (size_t)vstride];   // surfel of entry (r*64 + lane)
        if (nr > 1) { SLS_SSTAGE_LOAD_IDX(nr - 2) }
        for (int r = nr - 1; r >= 0; --r) {
            SLS_WSTAGE_STORE()
            SLS_TRACE_ROUND();
            const uint32_t my_idx = next_idx;
            s_gidx[lane] = my_idx;
            if (r > 0) {
                SLS_WSTAGE_LOAD_REC()
                next_idx = vals[(size_t)(range.x + (uint32_t)((r - 1) * 64 + lane)) * (size_t)vstride];
                if (r > 1) { SLS_SSTAGE_LOAD_IDX(r - 2) }
            }
            const int cnt = min(64, tmax - r * 64);
            const uint32_t c_lo = (uint32_t)(r * 64 + 1);
            float bcx, bcy, bhx, bhy;
            if (!block_active_box<BW, BH>(wave_ballot(inside && last >= c_lo), x0, y0, bcx, bcy, bhx, bhy)) continue;
            __builtin_amdgcn_wave_barrier();
            bool pass = false;
            if (lane < cnt) pass = cull_pass(s_rec[lane * kRec4 + 4], bcx, bcy, bhx, bhy, wrapW, invW);
            const uint64_t mask = wave_ballot(pass);
            if (mask == 0) continue;
            const int npass = __builtin_popcountll(mask);
            if (touched && pass) touched[my_idx] = 1;
            // survivors in DESCENDING list order
            if (pass) s_list[npass - 1 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint32_t)lane;
            if (lane < 3) s_list[npass + lane] = 64u;
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < npass; k += 4) {
                const int j = (int)s_list[k + slot];
                blend_step(j, (uint32_t)(r * 64 + j + 1));
            }
        }
#undef SLS_SIDX1
#undef SLS_SSTAGE_LOAD_IDX
    }

    if (dbg_cycles && lane == 0) dbg_cycles[tile * kPerTile + sub] = (uint32_t)(clock64() - t_start);
    SLS_TRACE_END(1);
}

// ---------------------------------------------------------------------------
// the hand-over buffer carries a launch order for the backward (8x2 blocks, XCD-interleaved tile mapping)
bool handover_has_order(int T) { return T % 32 == 0 && kTileW == 16 && kTileH == 16; }
const uint32_t *handover_block_order(const uint64_t *block_masks, int T)
{
    return reinterpret_cast<const uint32_t *>(block_masks + block_list_order_word(T, kTilePix / 16));
}

#ifdef SLS_TRACE
extern "C" int sls_debug_read_trace(uint32_t *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(g_trace));
}
extern "C" int sls_debug_read_trace_phases(uint32_t *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_phase), sizeof(g_trace_phase));
}
extern "C" int sls_debug_read_trace_phases_b(uint32_t *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_phase_b), sizeof(g_trace_phase_b));
}
extern "C" int sls_debug_read_trace_marks(uint32_t *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_mark), sizeof(g_trace_mark));
}
#endif

int launch_render_fwd_block(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                            const float *col_cs, const float *row_cs, float *allmap, float *pix_state,
                            uint32_t *pix_contrib, uint32_t *tile_consumed, uint64_t *block_masks, int shape,
                            hipStream_t st, bool lean, uint32_t *block_cost, const uint2 *bmask, bool order_in_handover)
{
    const int T = cam.GX * cam.GY;
    ScopedTimer tm(T_RENDER_FWD, st);
    const dim3 grid(T * (kTilePix / 16)), block(64);
    uint32_t *const g_dbg_fwd_cycles = debug_state().dbg_fwd_cycles;
    SLS_REQUIRE(shape == 0 || shape == 1, "tile-kernel variant must be 2 (4x4 blocks) or 3 (8x2 blocks)");
#define SLS_FWD_ARGS grid, block, 0, st, block_masks, cam, (const uint2 *)ranges, vals, (const float4 *)rec,     \
                     (const float2 *)col_cs, (const float2 *)row_cs, allmap, (float4 *)pix_state,                \
                     (uint2 *)pix_contrib, tile_consumed, g_dbg_fwd_cycles, block_cost
#define SLS_FWD_BLOCK(BW_, BH_, DBG_, LEAN_) hipLaunchKernelGGL((render_fwd_block_kernel<BW_, BH_, DBG_, LEAN_>), SLS_FWD_ARGS)
#define SLS_FWD_DENSE(DBG_, LEAN_) hipLaunchKernelGGL((render_fwd_dense_kernel<8, 2, DBG_, LEAN_>), SLS_FWD_ARGS, bmask)
    // the instances' block masks in list order (a passenger of the tile sort): dense rounds, 8x2 blocks only
    if (bmask && shape == 1) {
        if (g_dbg_fwd_cycles) SLS_FWD_DENSE(true, false); else if (lean) SLS_FWD_DENSE(false, true); else SLS_FWD_DENSE(false, false);
    } else
    if (g_dbg_fwd_cycles) { if (shape == 1) SLS_FWD_BLOCK(8, 2, true, false); else SLS_FWD_BLOCK(4, 4, true, false); }
    else if (lean) { if (shape == 1) SLS_FWD_BLOCK(8, 2, false, true); else SLS_FWD_BLOCK(4, 4, false, true); }
    else { if (shape == 1) SLS_FWD_BLOCK(8, 2, false, false); else SLS_FWD_BLOCK(4, 4, false, false); }
#undef SLS_FWD_BLOCK
#undef SLS_FWD_DENSE
#undef SLS_FWD_ARGS
    SLS_LAUNCH_CHECK("render_fwd_block_kernel");
    if (order_in_handover && block_masks && shape == 1 && handover_has_order(T)) {
        // the staged API: the order the backward launches its blocks in, from the counts the forward just wrote
        hipLaunchKernelGGL(block_order_kernel, dim3(8), dim3(256), 0, st, block_masks, T);
        SLS_LAUNCH_CHECK("block_order_kernel");
    }
    return SLS_OK;
}

int launch_render_bwd_block(const DevCam &cam, const uint32_t *ranges, const uint32_t *vals, const float *rec,
                            const float *col_cs, const float *row_cs, const float *pix_state,
                            const uint32_t *pix_contrib, const float *dL_dallmap, float *grec,
                            const uint64_t *block_masks, int shape, hipStream_t st, bool lean, uint8_t *touched,
                            const ConsumerArgs *fused_consumer, uint32_t *det_max, unsigned long long *det_acc,
                            const uint32_t *block_order, int vals_stride, bool dense, const uint8_t *det_prev,
                            const uint32_t *det_gex, uint32_t *det_flag, bool consumer_b_inline, uint32_t order_tag)
{
    const int T = cam.GX * cam.GY;
    ScopedTimer tm(T_RENDER_BWD, st);
    const dim3 grid(T * (kTilePix / 16)), block(64);
    uint32_t *const g_dbg_bwd_cycles = debug_state().dbg_bwd_cycles;
    SLS_REQUIRE(shape == 0 || shape == 1, "tile-kernel variant must be 2 (4x4 blocks) or 3 (8x2 blocks)");
    ConsumerArgs ca;
    memset(&ca, 0, sizeof(ca));
    int cblocks = 0;
#ifdef SLS_ABL_EXTRA
    int abl_mode = 0;
#endif
    if (fused_consumer) {
        ca = *fused_consumer;
        cblocks = ((ca.W + 63) / 64) * ((ca.H + 3) / 4);
    }
    // `dense`: the forward's compact lists come from a forward of THIS block shape (the caller carries the producer's
    // shape next to the buffer; the kernel checks the buffer's tag as a last guard); otherwise the backward culls.
#define SLS_BWD_LAUNCH(BW_, BH_, LEAN_, FUSED_, DET_, DENSE_)                                                        \
    hipLaunchKernelGGL((render_bwd_block_kernel<BW_, BH_, LEAN_, FUSED_, DET_, DENSE_>), grid, block, 0, st, cam, (const uint2 *)ranges, \
                       vals, (const float4 *)rec, (const float2 *)col_cs, (const float2 *)row_cs,                    \
                       (const float4 *)pix_state, (const uint2 *)pix_contrib, dL_dallmap, grec, block_masks,         \
                       touched, g_dbg_bwd_cycles, ca, cblocks, det_max, det_acc, block_order, vals_stride, det_prev, det_gex, det_flag, order_tag SLS_ABL_ARG(abl_mode))
#define SLS_BWD_BLOCK(BW_, BH_, LEAN_, FUSED_, DET_)                                                                 \
    do { if (dense) SLS_BWD_LAUNCH(BW_, BH_, LEAN_, FUSED_, DET_, true); else SLS_BWD_LAUNCH(BW_, BH_, LEAN_, FUSED_, DET_, false); } while (0)
    if (det_prev) {
        // deterministic accumulation in ONE launch: predicted scales (8x2 kernel on the forward's compact lists only)
        SLS_REQUIRE(det_acc && det_gex && det_flag && shape == 1 && dense, "the one-pass deterministic accumulation needs the 8x2 kernel on the forward's compact lists");
        if (fused_consumer) { SLS_REQUIRE(lean, "the fused consumer gradient exists for the lean kernel only");
                              if (consumer_b_inline) SLS_BWD_LAUNCH(8, 2, true, 2, 3, true); else SLS_BWD_LAUNCH(8, 2, true, 1, 3, true); }
        else if (lean) SLS_BWD_LAUNCH(8, 2, true, 0, 3, true);
        else SLS_BWD_LAUNCH(8, 2, false, 0, 3, true);
    } else if (det_max) {
        // deterministic accumulation: two launches of the 8x2 kernel (maximum, then fixed-point sum)
        SLS_REQUIRE(det_acc && shape == 1, "deterministic accumulation exists for the 8x2 kernel");
        if (fused_consumer) { SLS_REQUIRE(lean, "the fused consumer gradient exists for the lean kernel only");
                              if (consumer_b_inline) { SLS_BWD_BLOCK(8, 2, true, 2, 1); SLS_BWD_BLOCK(8, 2, true, 2, 2); }
                              else { SLS_BWD_BLOCK(8, 2, true, 1, 1); SLS_BWD_BLOCK(8, 2, true, 1, 2); } }
        else if (lean) { SLS_BWD_BLOCK(8, 2, true, 0, 1); SLS_BWD_BLOCK(8, 2, true, 0, 2); }
        else { SLS_BWD_BLOCK(8, 2, false, 0, 1); SLS_BWD_BLOCK(8, 2, false, 0, 2); }
    } else if (fused_consumer) {
        SLS_REQUIRE(lean && shape == 1, "the fused consumer gradient exists for the lean 8x2 kernel only");
        if (consumer_b_inline) SLS_BWD_BLOCK(8, 2, true, 2, 0); else SLS_BWD_BLOCK(8, 2, true, 1, 0);
#ifdef SLS_ABL_EXTRA
        if (consumer_b_inline && dense) {
            tm.end_now();
            ScopedTimer tm2(T_KNN, st);
            abl_mode = SLS_ABL_EXTRA;
            static float *abl_grec = nullptr;
            if (abl_mode == 2) {
                if (!abl_grec) { if (hipMalloc(&abl_grec, (size_t)1 << 28) != hipSuccess) return SLS_E_HIP; (void)hipMemsetAsync(abl_grec, 0, (size_t)1 << 28, st); }
                grec = abl_grec;
            }
            SLS_BWD_LAUNCH(8, 2, true, 2, 0, true);
        }
#endif
    } else if (lean) { if (shape == 1) SLS_BWD_BLOCK(8, 2, true, 0, 0); else SLS_BWD_BLOCK(4, 4, true, 0, 0); }
    else { if (shape == 1) SLS_BWD_BLOCK(8, 2, false, 0, 0); else SLS_BWD_BLOCK(4, 4, false, 0, 0); }
#undef SLS_BWD_LAUNCH
#undef SLS_BWD_BLOCK
    SLS_LAUNCH_CHECK("render_bwd_block_kernel");
    return SLS_OK;
}

}  // namespace sls
