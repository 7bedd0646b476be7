// sls_api.hip — the extern "C" boundary (include/sls_abi.h), host helpers,
// fused Adam (P6) and the device self-test.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <mutex>

#include "sls_common.hpp"

namespace sls {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------
// per-kernel timing (HIP events on the launch stream)
// ---------------------------------------------------------------------------
static const char *kTimerNames[T_COUNT] = {
    "preprocess_fwd", "scan", "emit_keys", "sort_hist", "sort_rowscan", "sort_scatter", "tile_ranges",
    "render_fwd", "grec_memset", "render_bwd", "preprocess_bwd", "adam", "knn", "consumer", "resort", "bin_count", "bin_direct"
};
constexpr int kTimerPool = 8192;
struct TimerState {
    bool enabled = false;
    int mode = 0;          // 1: every launch, 2: the two tile kernels only, 3: render_bwd only, 4: every 8th render_bwd
    int seen = 0;          // mode 4: render_bwd launches met since the mode was set
    int used = 0;
    int open = -1;         // index of the pair whose start was recorded and whose stop is pending
    int created = 0;
    hipEvent_t start[kTimerPool], stop[kTimerPool];
    int slot[kTimerPool];
};
// ONE recorder per process (allocated on first use), guarded by a mutex: a backward that torch's autograd engine
// runs on its own device thread records into the recorder the Python thread enabled (a per-thread recorder would
// silently miss render_bwd / preprocess_bwd in every autograd-driven mode).  Launches of one stream are enqueued
// in order, so begin/end pairs of different threads do not interleave inside a stream.
static TimerState *t_timer = nullptr;
static std::mutex g_timer_mutex;
static std::atomic<bool> g_timer_on{false};
#define g_timer (*t_timer)

static inline bool timer_wants(int slot)
{
    if (!t_timer || !g_timer.enabled) return false;
    if (g_timer.mode == 1) return true;
    if (g_timer.mode == 2) return slot == T_RENDER_FWD || slot == T_RENDER_BWD;
    return slot == T_RENDER_BWD;
}
// begin reserves an event pair and hands its index back through the ScopedTimer (slot field reused: -1 = untimed)
void timer_begin(int slot, hipStream_t st)
{
    if (!g_timer_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_timer_mutex);
    if (!timer_wants(slot) || g_timer.used >= g_timer.created) return;
    // mode 4 samples the launches: an event pair per launch costs the host a few microseconds of an iteration that
    // is host-bound, so a timed region that wants an undisturbed clock AND a live duration brackets one launch in 8
    if (g_timer.mode == 4 && (g_timer.seen++ & 7) != 0) return;
    g_timer.slot[g_timer.used] = slot;
    (void)hipEventRecord(g_timer.start[g_timer.used], st);
    g_timer.open = g_timer.used;
}
void timer_end(int slot, hipStream_t st)
{
    if (!g_timer_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_timer_mutex);
    if (!timer_wants(slot) || g_timer.used >= g_timer.created || g_timer.open != g_timer.used) return;
    (void)hipEventRecord(g_timer.stop[g_timer.used], st);
    g_timer.open = -1;
    ++g_timer.used;
}

// launchers implemented in the other translation units
int launch_mark_visible(const DevCam &, int, const float *, uint8_t *, hipStream_t);
size_t consumer_scratch_bytes(int H, int W);
int launch_consumer(int H, int W, const float *allmap, const float *gt_depth, const uint8_t *valid,
                    const float *col_h, const float *row_h, float depth_ratio, float lambda_n, float lambda_a,
                    int n_valid, float *sums, float *dL_dallmap, void *scratch, size_t scratch_bytes,
                    hipStream_t st, bool sums_zeroed = false, struct ConsumerArgs *args_out_skip_c = nullptr,
                    int order_tiles = 0, const uint32_t *block_cost = nullptr, uint32_t *block_order = nullptr,
                    bool no_launch = false);
int launch_render_maps(int H, int W, const float *allmap, const float *rot9, const float *col_h, const float *row_h,
                       float depth_ratio, float *rend_normal, float *surf_depth, float *surf_normal, hipStream_t st);
int launch_densify_weights(int H, int W, const float *depth, const uint8_t *valid, const float *alpha, float thr, float *w_out,
                           uint32_t *stats, hipStream_t st);
int launch_densify_rows(int n, int H, int W, const int64_t *pix, const float *depth, const float *normal, const float *col_h,
                        const float *row_h, const float *c2w, const float *mTf, float *xyz, float *quat, hipStream_t st);
size_t knn_scratch_bytes(int M);
int launch_knn(int M, const float *xyz, float *out, void *scratch, size_t scratch_bytes, hipStream_t st, int Mq = -1);

// ---------------------------------------------------------------------------
// P6 fused Adam: every parameter tensor of the model in ONE launch.
// HBM-bound: 16 B read + 12 B written per element.
// ---------------------------------------------------------------------------
constexpr int kMaxAdamGroups = 8;
struct AdamArgs {
    SlsAdamGroup g[kMaxAdamGroups];
    int64_t unit_end[kMaxAdamGroups];   // prefix of ceil(numel/4) units
    int ngroups;
    AdamCoef c;
};

// void_flags (keyframe-parallel mode): two floats that rode the gradient all-reduce, > 0 if ANY rank
// voided the iteration (instance buffers too small / another reason); the reduced bits are also
// written to *status_word so that the caller's one status read sees them.
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a, int64_t total_units,
                                                   const uint32_t *__restrict__ skip_flag,
                                                   const float *__restrict__ void_flags,
                                                   uint32_t *status_block, uint32_t *status_mirror)
{
    if (void_flags) {
        const uint32_t bits = (void_flags[0] > 0.0f ? 1u : 0u) | (void_flags[1] > 0.0f ? 2u : 0u);
        if (blockIdx.x == 0 && threadIdx.x == 0 && status_block) {
            status_block[1] = bits;                       // SlsMappingStatus.overflow: the group's verdict
            if (status_mirror) mirror_status_block(status_block, status_mirror, true, bits);
        }
        if (bits) return;
    }
    if (skip_flag && *skip_flag) return;   // e.g. the instance buffers overflowed: gradients are incomplete
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total_units; u += (int64_t)gridDim.x * 256) {
        int gi = 0;
#pragma unroll
        for (int k = 0; k < kMaxAdamGroups - 1; ++k)
            if (k < a.ngroups - 1 && u >= a.unit_end[k]) gi = k + 1;
        const SlsAdamGroup grp = a.g[gi];
        const int64_t e0 = (u - (gi ? a.unit_end[gi - 1] : 0)) * 4;
        const float step_size = grp.lr * __builtin_amdgcn_rcpf(a.c.bc1);   // (as the fused update in preprocess_bwd)
        const bool vec = (e0 + 4 <= grp.numel) &&
                         ((((uintptr_t)grp.param | (uintptr_t)grp.grad | (uintptr_t)grp.exp_avg | (uintptr_t)grp.exp_avg_sq) & 15) == 0);
        if (vec) {
            float4 p = *reinterpret_cast<float4 *>(grp.param + e0);
            const float4 g = *reinterpret_cast<const float4 *>(grp.grad + e0);
            float4 m = *reinterpret_cast<float4 *>(grp.exp_avg + e0);
            float4 v = *reinterpret_cast<float4 *>(grp.exp_avg_sq + e0);
            adam_one(p.x, g.x, m.x, v.x, step_size, a.c);
            adam_one(p.y, g.y, m.y, v.y, step_size, a.c);
            adam_one(p.z, g.z, m.z, v.z, step_size, a.c);
            adam_one(p.w, g.w, m.w, v.w, step_size, a.c);
            *reinterpret_cast<float4 *>(grp.param + e0) = p;
            *reinterpret_cast<float4 *>(grp.exp_avg + e0) = m;
            *reinterpret_cast<float4 *>(grp.exp_avg_sq + e0) = v;
        } else {
            for (int64_t e = e0; e < e0 + 4 && e < grp.numel; ++e) {
                float p = grp.param[e], m = grp.exp_avg[e], v = grp.exp_avg_sq[e];
                adam_one(p, grp.grad[e], m, v, step_size, a.c);
                grp.param[e] = p; grp.exp_avg[e] = m; grp.exp_avg_sq[e] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// device self-test of the wave64 primitives the kernels rely on
// ---------------------------------------------------------------------------
__global__ void selftest_kernel(int *out)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    // DPP sum: lanes hold (lane+1) * 0.5 -> total 1040 at lane 63
    const float tot = wave_sum_to_lane63(0.5f * (float)(lane + 1));
    if (lane == 63 && tot != 1040.0f) bad |= 1;
    // asymmetric pattern: only lanes 5 and 40 non-zero
    const float t2 = wave_sum_to_lane63(lane == 5 ? 3.0f : (lane == 40 ? 11.0f : 0.0f));
    if (lane == 63 && t2 != 14.0f) bad |= 2;
    // ballot / popcount ranking
    const uint64_t bal = __ballot((lane % 3) == 0);
    if (__popcll(bal) != 22) bad |= 4;
    if (lane_id() != lane) bad |= 8;
    const uint64_t lt = (1ull << lane) - 1ull;
    if ((lane % 3) == 0 && (int)__popcll(bal & lt) != lane / 3) bad |= 16;
    // 16-way reduce-scatter of the backward tile kernel: x[k] = (lane+1)(k+1)/4; lane (pixel p, slot s) gets
    // component p summed over the 16 lanes of slot s
    v2f x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = mk2(0.25f * (float)((lane + 1) * (2 * k + 1)), 0.25f * (float)((lane + 1) * (2 * k + 2)));
    const float b16 = block_reduce16_pk(x, lane);
    if (b16 != (float)(124 + 4 * (lane & 3)) * (float)((lane >> 2) + 1)) bad |= 256;
    if (dpp_xor4((float)lane) != (float)(lane ^ 4)) bad |= 512;
    int xa, xb, ya, yb;
    const uint64_t mm = __ballot(lane == 10 || lane == 29 || lane == 52);   // (2,1) (5,3) (4,6)
    if (!mask_bbox8x8(mm, xa, xb, ya, yb) || xa != 2 || xb != 5 || ya != 1 || yb != 6) bad |= 128;
    if (bad) atomicOr(out, bad);
}

}  // namespace sls

using namespace sls;

extern "C" {

const char *sls_last_error(void) { return g_err; }
int sls_version(void) { return 100; }
int sls_tile_w(void) { return kTileW; }
int sls_tile_h(void) { return kTileH; }
int sls_rec_stride(void) { return SLS_REC_STRIDE; }
int sls_grec_stride(void) { return SLS_GREC_STRIDE; }

int sls_camera_from_matrices(const float *view, const float *proj, int H, int W, float scale_modifier,
                             SlsCamera *out)
{
    SLS_REQUIRE(view && proj && out, "null pointer");
    SLS_REQUIRE(H > 0 && W > 0, "image size must be positive");
    memset(out, 0, sizeof(*out));
    out->H = H; out->W = W;
    // K = proj[:3,:3]^T  (scene/cameras.py:47-50), row-major proj[r*4+c]
    const float k01 = proj[1 * 4 + 0], k10 = proj[0 * 4 + 1];
    SLS_REQUIRE(k01 == 0.0f && k10 == 0.0f, "skewed intrinsics are not supported");
    out->fx = proj[0 * 4 + 0];
    out->fy = proj[1 * 4 + 1];
    out->cx = proj[2 * 4 + 0];
    out->cy = proj[2 * 4 + 1];
    SLS_REQUIRE(out->fx != 0.0f && out->fy != 0.0f, "fx and fy must be non-zero");
    // p_view = R_vw p + t, R_vw = view[:3,:3]^T, t = view[3,:3]  (scene/cameras.py:43-46)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out->Rvw[3 * r + c] = view[c * 4 + r];
    for (int c = 0; c < 3; ++c) out->tvw[c] = view[3 * 4 + c];
    out->scale_modifier = scale_modifier;
    out->near_cut = SLS_NEAR;
    out->far_cut = SLS_FAR;
    // D5: 360-degree image <=> |fx| * 2pi == W (within a pixel) and whole tiles per row
    const double period = fabs((double)out->fx) * 2.0 * 3.14159265358979323846;
    out->wrap = (fabs(period - (double)W) <= 1.0 && (W % kTileW) == 0) ? 1 : 0;
    return SLS_OK;
}

// rays of pixel (c, r) at image coordinate (c + oc, r + orow) for a principal point (cx, cy)
static void ray_tables_for(const SlsCamera *cam, double oc, double orow, double cx, double cy, float *col_cs, float *row_cs)
{
    for (int c = 0; c < cam->W; ++c) {
        const double a = ((double)c + oc - cx) / (double)cam->fx;
        col_cs[2 * c] = (float)cos(a);
        col_cs[2 * c + 1] = (float)sin(a);
    }
    for (int r = 0; r < cam->H; ++r) {
        const double e = ((double)r + orow - cy) / (double)cam->fy;
        row_cs[2 * r] = (float)cos(e);
        row_cs[2 * r + 1] = (float)sin(e);
    }
}

int sls_ray_tables(const SlsCamera *cam, float *col_cs, float *row_cs)
{
    SLS_REQUIRE(cam && col_cs && row_cs, "null pointer");
    // the rasterizer's own rays: the principal point the kernels use (make_devcam), rounded to float as there
    const DevCam dc = make_devcam(*cam);
    ray_tables_for(cam, 0.0, 0.0, (double)dc.cx, (double)dc.cy, col_cs, row_cs);
    return SLS_OK;
}

int sls_ray_tables_at(const SlsCamera *cam, float col_offset, float row_offset, float *col_cs, float *row_cs)
{
    SLS_REQUIRE(cam && col_cs && row_cs, "null pointer");
    ray_tables_for(cam, (double)col_offset, (double)row_offset, (double)cam->cx, (double)cam->cy, col_cs, row_cs);
    return SLS_OK;
}

}  // extern "C"

namespace sls {
int launch_adam(const SlsAdamGroup *groups, int ngroups, double beta1, double beta2, double eps, int64_t step,
                const uint32_t *skip_flag, hipStream_t stream, const float *void_flags, uint32_t *status_block,
                uint32_t *status_mirror)
{
    SLS_REQUIRE(groups && ngroups > 0 && ngroups <= kMaxAdamGroups, "1..8 groups");
    SLS_REQUIRE(step >= 1, "step is 1-based");
    AdamArgs a;
    memset(&a, 0, sizeof(a));
    int64_t units = 0;
    for (int i = 0; i < ngroups; ++i) {
        SLS_REQUIRE(groups[i].numel >= 0, "negative numel");
        SLS_REQUIRE(groups[i].numel == 0 || (groups[i].param && groups[i].grad && groups[i].exp_avg && groups[i].exp_avg_sq),
                    "null tensor");
        a.g[i] = groups[i];
        units += (groups[i].numel + 3) / 4;
        a.unit_end[i] = units;
    }
    for (int i = ngroups; i < kMaxAdamGroups; ++i) a.unit_end[i] = units;
    a.ngroups = ngroups;
    a.c = make_adam_coef(beta1, beta2, eps, step);
    if (units == 0 && !void_flags) return SLS_OK;   // (with void flags the verdict is still published)
    int64_t blocks = (units + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;   // 8 blocks per CU, grid-stride beyond
    {
        ScopedTimer tm(T_ADAM, (hipStream_t)stream);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, units, skip_flag,
                           void_flags, status_block, status_mirror);
    }
    SLS_LAUNCH_CHECK("adam_kernel");
    return SLS_OK;
}
}  // namespace sls

extern "C" {

int sls_adam_step(const SlsAdamGroup *groups, int ngroups, double beta1, double beta2, double eps, int64_t step,
                  void *stream)
{
    return launch_adam(groups, ngroups, beta1, beta2, eps, step, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

int sls_adam_step_guarded(const SlsAdamGroup *groups, int ngroups, double beta1, double beta2, double eps,
                          int64_t step, const uint32_t *skip_flag_dev, void *stream)
{
    return launch_adam(groups, ngroups, beta1, beta2, eps, step, skip_flag_dev, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

int sls_adam_step_reduced(const SlsAdamGroup *groups, int ngroups, double beta1, double beta2, double eps,
                          int64_t step, const float *void_flags_dev, SlsMappingStatus *status_dev,
                          SlsMappingStatus *status_mirror, void *stream)
{
    SLS_REQUIRE(void_flags_dev, "null pointer");
    SLS_REQUIRE(!status_mirror || status_dev, "status_mirror needs status_dev");
    static_assert(sizeof(SlsMappingStatus) == 32, "the mirror copy moves 8 words");
    return launch_adam(groups, ngroups, beta1, beta2, eps, step, nullptr, (hipStream_t)stream, void_flags_dev,
                       (uint32_t *)status_dev, (uint32_t *)status_mirror);
}

size_t sls_consumer_scratch_bytes(int H, int W) { return (H > 0 && W > 0) ? consumer_scratch_bytes(H, W) : 0; }

int sls_consumer_fwd_bwd(int H, int W, const float *allmap, const float *gt_depth, const uint8_t *valid,
                         const float *col_cs_half, const float *row_cs_half, float depth_ratio,
                         float lambda_normal, float lambda_alpha, int n_valid, float *loss_sums,
                         float *dL_dallmap, void *scratch, size_t scratch_bytes, void *stream)
{
    SLS_REQUIRE(H > 0 && W > 0 && n_valid >= 0, "bad size");
    SLS_REQUIRE(allmap && gt_depth && valid && col_cs_half && row_cs_half && loss_sums && dL_dallmap && scratch,
                "null pointer");
    return launch_consumer(H, W, allmap, gt_depth, valid, col_cs_half, row_cs_half, depth_ratio, lambda_normal,
                           lambda_alpha, n_valid, loss_sums, dL_dallmap, scratch, scratch_bytes,
                           (hipStream_t)stream);
}

int sls_render_maps(int H, int W, const float *allmap, const float *view_rot9, const float *col_cs_half,
                    const float *row_cs_half, float depth_ratio, float *rend_normal, float *surf_depth,
                    float *surf_normal, void *stream)
{
    SLS_REQUIRE(H > 0 && W > 0, "bad size");
    SLS_REQUIRE(allmap && view_rot9 && col_cs_half && row_cs_half && rend_normal && surf_depth && surf_normal, "null pointer");
    return launch_render_maps(H, W, allmap, view_rot9, col_cs_half, row_cs_half, depth_ratio, rend_normal, surf_depth,
                              surf_normal, (hipStream_t)stream);
}

int sls_densify_weights(int H, int W, const float *image_depth, const uint8_t *valid, const float *rend_alpha,
                        float threshold_opacity, float *weights_out, uint32_t *stats_out, void *stream)
{
    SLS_REQUIRE(H > 0 && W > 0, "bad size");
    SLS_REQUIRE(image_depth && valid && weights_out && stats_out, "null pointer");
    return launch_densify_weights(H, W, image_depth, valid, rend_alpha, threshold_opacity, weights_out, stats_out,
                                  (hipStream_t)stream);
}

int sls_densify_rows(int n, int H, int W, const int64_t *pixels, const float *image_depth, const float *image_normal,
                     const float *col_cs_half, const float *row_cs_half, const float *cam_to_model16,
                     const float *model_T_frame16, float *xyz_out, float *quat_out, void *stream)
{
    SLS_REQUIRE(n >= 0 && H > 0 && W > 0, "bad size");
    if (n == 0) return SLS_OK;
    SLS_REQUIRE(pixels && image_depth && image_normal && col_cs_half && row_cs_half && cam_to_model16 && model_T_frame16 &&
                    xyz_out && quat_out, "null pointer");
    return launch_densify_rows(n, H, W, pixels, image_depth, image_normal, col_cs_half, row_cs_half, cam_to_model16,
                               model_T_frame16, xyz_out, quat_out, (hipStream_t)stream);
}

size_t sls_knn_scratch_bytes(int M) { return knn_scratch_bytes(M); }

int sls_knn_dist2(int M, const float *xyz, float *out, void *scratch, size_t scratch_bytes, void *stream)
{
    SLS_REQUIRE(M >= 0, "negative M");
    if (M == 0) return SLS_OK;
    SLS_REQUIRE(xyz && out && scratch, "null pointer");
    return launch_knn(M, xyz, out, scratch, scratch_bytes, (hipStream_t)stream);
}

int sls_wait_status_mirror(const void *mirror_host, uint32_t sentinel, void *stream)
{
    // The drop-in forward's status (R, void bits) arrives in the caller's pinned host block a few microseconds into the
    // binning; the caller must know it before it hands the image on.  Spun HERE (the binding releases the interpreter's
    // lock around a foreign call) with the CPU's pause hint; if the words have not arrived after ~2 ms of spinning the
    // stream is drained — a lost mirror write must not hang the caller.
    SLS_REQUIRE(mirror_host, "null pointer");
    const volatile uint32_t *w = (const volatile uint32_t *)mirror_host;
    for (int spin = 0; spin < 400000; ++spin) {
        if (w[7] != sentinel && w[0] != sentinel) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return SLS_OK; }
        __builtin_ia32_pause();
    }
    SLS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (w[7] != sentinel && w[0] != sentinel) return SLS_OK;
    set_error("the forward's status never reached its host mirror");
    return SLS_E_HIP;
}

int sls_knn_dist2_first(int M, int M_first, const float *xyz, float *out, void *scratch, size_t scratch_bytes, void *stream)
{
    SLS_REQUIRE(M >= 0 && M_first >= 0 && M_first <= M, "bad sizes");
    if (M == 0 || M_first == 0) return SLS_OK;
    SLS_REQUIRE(xyz && out && scratch, "null pointer");
    return launch_knn(M, xyz, out, scratch, scratch_bytes, (hipStream_t)stream, M_first);
}

int sls_mark_visible(const SlsCamera *cam, int N, const float *means3D, uint8_t *visible, void *stream)
{
    SLS_REQUIRE(cam && N >= 0, "bad argument");
    if (N == 0) return SLS_OK;
    SLS_REQUIRE(means3D && visible, "null pointer");
    return launch_mark_visible(make_devcam(*cam), N, means3D, visible, (hipStream_t)stream);
}

int sls_debug_wave_cycles(uint32_t *fwd_cycles, uint32_t *bwd_cycles)
{
    debug_state().dbg_fwd_cycles = fwd_cycles;
    debug_state().dbg_bwd_cycles = bwd_cycles;
    return SLS_OK;
}

int sls_debug_variant(int fwd_variant, int bwd_variant)
{
    SLS_REQUIRE((fwd_variant < 0 || fwd_variant == 2 || fwd_variant == 3) &&
                    (bwd_variant < 0 || bwd_variant == 2 || bwd_variant == 3),
                "variants: 2 = 4x4 pixel blocks, 3 = 8x2 pixel blocks (negative: leave as it is)");
    if (fwd_variant >= 0) debug_state().fwd_variant = fwd_variant;
    if (bwd_variant >= 0) debug_state().bwd_variant = bwd_variant;
    return SLS_OK;
}

int sls_timing_slots(void) { return T_COUNT; }
const char *sls_timing_name(int slot) { return (slot >= 0 && slot < T_COUNT) ? kTimerNames[slot] : ""; }

int sls_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_timer_mutex);
    if (!t_timer) {
        if (!on) return SLS_OK;
        t_timer = new TimerState();
    }
    if (on && g_timer.created == 0) {
        for (int i = 0; i < kTimerPool; ++i) {
            if (hipEventCreate(&g_timer.start[i]) != hipSuccess || hipEventCreate(&g_timer.stop[i]) != hipSuccess) break;
            g_timer.created = i + 1;
        }
    }
    g_timer.enabled = on != 0;
    g_timer.mode = on;
    g_timer.used = 0;
    g_timer.seen = 0;
    g_timer.open = -1;
    g_timer_on.store(on != 0, std::memory_order_relaxed);
    return SLS_OK;
}

int sls_timing_collect(double *total_ms, int64_t *counts)
{
    SLS_REQUIRE(total_ms && counts, "null pointer");
    for (int s = 0; s < T_COUNT; ++s) { total_ms[s] = 0.0; counts[s] = 0; }
    std::lock_guard<std::mutex> lk(g_timer_mutex);
    if (!t_timer) return SLS_OK;
    for (int i = 0; i < g_timer.used; ++i) {
        SLS_HIP_CHECK(hipEventSynchronize(g_timer.stop[i]));
        float ms = 0.0f;
        SLS_HIP_CHECK(hipEventElapsedTime(&ms, g_timer.start[i], g_timer.stop[i]));
        total_ms[g_timer.slot[i]] += (double)ms;
        counts[g_timer.slot[i]] += 1;
    }
    const int dropped = (g_timer.used >= g_timer.created) ? 1 : 0;
    g_timer.used = 0;
    g_timer.open = -1;
    return dropped ? 1 : SLS_OK;   // 1: pool exhausted, later launches were not timed
}

int sls_selftest(void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int *d = nullptr;
    SLS_HIP_CHECK(hipMalloc(&d, sizeof(int)));   // test-only entry point: the one place the library allocates
    int h = 0;
    hipError_t e = hipMemsetAsync(d, 0, sizeof(int), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_kernel, dim3(2), dim3(128), 0, st, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess) {
        set_error("selftest: %s", hipGetErrorString(e));
        return SLS_E_HIP;
    }
    if (h != 0) {
        set_error("selftest: wave primitive mismatch, flags 0x%x", h);
        return SLS_E_HIP;
    }
    return SLS_OK;
}

}  // extern "C"
