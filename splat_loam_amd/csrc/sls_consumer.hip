// sls_consumer.hip — the consumer of allmap fused into two small kernels:
// render()'s post-processing (gaussian_renderer/__init__.py:48-93), the
// spherical back-projection + central-difference normals it calls
// (utils/graphic_utils.py:26-88) and the per-pixel terms of the mapper's loss
// (slam/mapper.py:158-187), forward AND gradient w.r.t. allmap.
// in the reference this stage is ~85 tiny torch kernels plus
// host syncs per iteration and costs more than the rasterizer itself.
//
//   loss = mean_all |valid (s - gt)|                              (geom, :174-175)
//        + lambda_n * mean_valid (1 - <n_hat, n_surf * alpha>)    (normal, :177-180)
//        + lambda_a * mean_valid BCE(alpha, 1)                    (alpha, :182-187)
//   s      = D/alpha * (1 - depth_ratio) + median * depth_ratio   (surf_depth)
//   n_hat  = N/alpha (view frame; the loss is rotation invariant, so nothing is
//            rotated to the world frame here)
//   n_surf = normalize((P[r+1,c]-P[r-1,c]) x (P[r,c+1]-P[r,c-1])), P = s * ray(c-.5, r-.5),
//            zero on the 1-pixel border.
// Kernel B (per pixel): loss terms (block-reduced, one atomic per block and
// term) + the stencil's adjoint pieces dL/du, dL/dv; kernel C (per pixel):
// gathers the four neighbours' pieces and writes dL/dallmap.  HBM-bound,
// ~100 B/pixel.
#include "sls_consumer_dev.hpp"

namespace sls {

__global__ __launch_bounds__(256) void consumer_b_kernel(ConsumerArgs a, int consumer_blocks_x, int consumer_blocks)
{
    if ((int)blockIdx.x >= consumer_blocks) {          // the passenger workgroups
        order_blocks_by_cost(a.order_tiles, a.block_cost, a.block_order, (int)blockIdx.x - consumer_blocks);
        return;
    }
    const int bx = (int)blockIdx.x % consumer_blocks_x, by = (int)blockIdx.x / consumer_blocks_x;
    const int c = bx * 64 + (threadIdx.x & 63);
    const int r = by * 4 + (threadIdx.x >> 6);
    float lg = 0.0f, ln = 0.0f, la = 0.0f;
    if (c < a.W && r < a.H) {
        const size_t pix = (size_t)r * a.W + c;
        float4 du, dv, nsd;
        consumer_b_pixel(a, r, c, du, dv, nsd, lg, ln, la);
        a.du[pix] = du; a.dv[pix] = dv; a.ns[pix] = nsd;
    }
    // block reduction -> one atomic per block and term
    __shared__ float s_part[3][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lg += __shfl_down(lg, off, 64);
        ln += __shfl_down(ln, off, 64);
        la += __shfl_down(la, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_part[0][wave] = lg; s_part[1][wave] = ln; s_part[2][wave] = la; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = s_part[threadIdx.x][0] + s_part[threadIdx.x][1] + s_part[threadIdx.x][2] + s_part[threadIdx.x][3];
        a.partials[blockIdx.x * 3 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void consumer_c_kernel(ConsumerArgs a)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && blockIdx.y == 0) {   // the per-block partial sums of kernel B -> the three sums + the total
        __shared__ float s_red[3][4];
        const int nb = gridDim.x * gridDim.y;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        for (int b = threadIdx.x; b < nb; b += 256) { t0 += a.partials[3 * b]; t1 += a.partials[3 * b + 1]; t2 += a.partials[3 * b + 2]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            t0 += __shfl_down(t0, off, 64); t1 += __shfl_down(t1, off, 64); t2 += __shfl_down(t2, off, 64);
        }
        if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = t0; s_red[1][threadIdx.x >> 6] = t1; s_red[2][threadIdx.x >> 6] = t2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float g = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
            const float n = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
            const float al = s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3];
            a.sums[0] = g; a.sums[1] = n; a.sums[2] = al;
            a.sums[3] = g * a.inv_P + a.lambda_n * a.inv_nv * n + a.lambda_a * a.inv_nv * al;
        }
    }
    if (c >= a.W || r >= a.H) return;
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    float g[7];
    consumer_pixel_grad(a, r, c, g);
    a.dL_dallmap[SLS_CH_DEPTH * P + pix] = g[0];
    a.dL_dallmap[SLS_CH_ALPHA * P + pix] = g[1];
    a.dL_dallmap[(SLS_CH_NORMAL + 0) * P + pix] = g[2];
    a.dL_dallmap[(SLS_CH_NORMAL + 1) * P + pix] = g[3];
    a.dL_dallmap[(SLS_CH_NORMAL + 2) * P + pix] = g[4];
    a.dL_dallmap[SLS_CH_MEDIAN * P + pix] = g[5];
    a.dL_dallmap[SLS_CH_DIST * P + pix] = g[6];
}

size_t consumer_scratch_bytes(int H, int W)
{
    const size_t nblocks = (size_t)((W + 63) / 64) * (size_t)((H + 3) / 4);
    return sizeof(float4) * 3 * (size_t)H * (size_t)W + sizeof(float) * 3 * nblocks;
}

int launch_consumer(int H, int W, const float *allmap, const float *gt_depth, const uint8_t *valid,
                    const float *col_h, const float *row_h, float depth_ratio, float lambda_n, float lambda_a,
                    int n_valid, float *sums, float *dL_dallmap, void *scratch, size_t scratch_bytes,
                    hipStream_t st, bool sums_zeroed, ConsumerArgs *args_out_skip_c, int order_tiles,
                    const uint32_t *block_cost, uint32_t *block_order, bool no_launch)
{
    if (scratch_bytes < consumer_scratch_bytes(H, W)) {
        set_error("consumer scratch too small");
        return SLS_E_SCRATCH;
    }
    ConsumerArgs a;
    a.H = H; a.W = W;
    a.depth_ratio = depth_ratio; a.lambda_n = lambda_n; a.lambda_a = lambda_a;
    a.inv_P = 1.0f / ((float)H * (float)W);
    a.inv_nv = n_valid > 0 ? 1.0f / (float)n_valid : 0.0f;
    a.allmap = allmap; a.gt_depth = gt_depth; a.valid = valid;
    a.col_h = (const float2 *)col_h; a.row_h = (const float2 *)row_h;
    a.du = (float4 *)scratch;
    a.dv = a.du + (size_t)H * W;
    a.ns = a.dv + (size_t)H * W;
    a.partials = (float *)(a.ns + (size_t)H * W);
    a.sums = sums;
    a.dL_dallmap = dL_dallmap;
    a.order_tiles = (order_tiles > 0 && order_tiles % 32 == 0 && block_cost && block_order && kTilePix / 16 == 16) ? order_tiles : 0;
    a.block_cost = block_cost;
    a.block_order = block_order;
    if (no_launch) {
        // kernel B runs inside the backward tile kernel (FUSED = 2 there): no plane is written, the blocks' loss terms —
        // three floats per 16-pixel block — take the planes' place at the head of the scratch
        if (!args_out_skip_c) { set_error("internal: the inline consumer needs the fused backward"); return SLS_E_ARG; }
        a.partials = (float *)scratch;
        a.order_tiles = 0;
        *args_out_skip_c = a;
        return SLS_OK;
    }
    ScopedTimer tm(T_CONSUMER, st);
    (void)sums_zeroed;   // (the sums are written, not accumulated)
    const dim3 grid((W + 63) / 64, (H + 3) / 4);
    const int cb = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(consumer_b_kernel, dim3(cb + (a.order_tiles ? 8 : 0)), dim3(256), 0, st, a, (int)grid.x, cb);
    SLS_LAUNCH_CHECK("consumer_b_kernel");
    if (args_out_skip_c) {          // kernel C's work is done by the backward tile kernel (FUSED)
        *args_out_skip_c = a;
        return SLS_OK;
    }
    hipLaunchKernelGGL(consumer_c_kernel, grid, dim3(256), 0, st, a);
    SLS_LAUNCH_CHECK("consumer_c_kernel");
    return SLS_OK;
}

// ---------------------------------------------------------------------------
// render()'s post-processing WITHOUT the loss (gaussian_renderer/__init__.py:48-93, utils/graphic_utils.py:26-88): what
// the reference's no-grad callers of render() read — Mapper.densify (slam/mapper.py:52-54), the tracker
// (slam/tracker.py:173-175), the logger (slam/slam.py:81-82), meshing (scene/postprocessing.py:162).  One launch, a
// thread per pixel:
//   rend_normal = R (N / alpha)            R = world_view_transform[:3,:3] as a matrix: view frame -> world (rot, 9 floats row-major)
//   surf_depth  = D / alpha (1 - ratio) + median ratio
//   surf_normal = alpha R normalize((P(r+1,c) - P(r-1,c)) x (P(r,c+1) - P(r,c-1))),  P = surf_depth x ray at (c-.5, r-.5); 0 on the border
// (the reference differences WORLD points; a rotation commutes with the cross product and the translation cancels, so
//  the sensor-frame differences rotated once are the same normal without the translation's rounding).
// rend_alpha and rend_dist are planes 1 and 6 of allmap as they are.
// ---------------------------------------------------------------------------
struct RenderMapsArgs {
    int H, W;
    float depth_ratio;
    float rot[9];
    const float *allmap;
    const float2 *col_h, *row_h;
    float *rend_normal, *surf_depth, *surf_normal;
};

__global__ __launch_bounds__(256) void render_maps_kernel(RenderMapsArgs a)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= a.W || r >= a.H) return;
    const size_t P = (size_t)a.H * a.W, pix = (size_t)r * a.W + c;
    ConsumerArgs ca;
    ca.H = a.H; ca.W = a.W; ca.depth_ratio = a.depth_ratio; ca.allmap = a.allmap; ca.col_h = a.col_h; ca.row_h = a.row_h;
    const float al = a.allmap[SLS_CH_ALPHA * P + pix];
    const bool hit = al > 0.0f;
    const float N0 = a.allmap[(SLS_CH_NORMAL + 0) * P + pix], N1 = a.allmap[(SLS_CH_NORMAL + 1) * P + pix],
                N2 = a.allmap[(SLS_CH_NORMAL + 2) * P + pix];
    // (torch: the rotation first, then the division)
    const float w0 = a.rot[0] * N0 + a.rot[1] * N1 + a.rot[2] * N2, w1 = a.rot[3] * N0 + a.rot[4] * N1 + a.rot[5] * N2,
                w2 = a.rot[6] * N0 + a.rot[7] * N1 + a.rot[8] * N2;
    a.rend_normal[pix] = hit ? w0 / al : w0;
    a.rend_normal[P + pix] = hit ? w1 / al : w1;
    a.rend_normal[2 * P + pix] = hit ? w2 / al : w2;
    float s;
    (void)surf_point(ca, r, c, s);
    a.surf_depth[pix] = s;
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
    if (r > 0 && r < a.H - 1 && c > 0 && c < a.W - 1) {
        float t;
        const float3 pd = surf_point(ca, r + 1, c, t), pu = surf_point(ca, r - 1, c, t);
        const float3 pr = surf_point(ca, r, c + 1, t), pl = surf_point(ca, r, c - 1, t);
        const float ux = pd.x - pu.x, uy = pd.y - pu.y, uz = pd.z - pu.z;      // d_row
        const float vx = pr.x - pl.x, vy = pr.y - pl.y, vz = pr.z - pl.z;      // d_col
        const float cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
        const float inv = al / fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);        // (F.normalize's eps, then x alpha)
        const float sx = cx * inv, sy = cy * inv, sz = cz * inv;
        n0 = a.rot[0] * sx + a.rot[1] * sy + a.rot[2] * sz;
        n1 = a.rot[3] * sx + a.rot[4] * sy + a.rot[5] * sz;
        n2 = a.rot[6] * sx + a.rot[7] * sy + a.rot[8] * sz;
    }
    a.surf_normal[pix] = n0; a.surf_normal[P + pix] = n1; a.surf_normal[2 * P + pix] = n2;
}

int launch_render_maps(int H, int W, const float *allmap, const float *rot9, const float *col_h, const float *row_h,
                       float depth_ratio, float *rend_normal, float *surf_depth, float *surf_normal, hipStream_t st)
{
    RenderMapsArgs a;
    a.H = H; a.W = W; a.depth_ratio = depth_ratio;
    for (int k = 0; k < 9; ++k) a.rot[k] = rot9[k];
    a.allmap = allmap; a.col_h = (const float2 *)col_h; a.row_h = (const float2 *)row_h;
    a.rend_normal = rend_normal; a.surf_depth = surf_depth; a.surf_normal = surf_normal;
    ScopedTimer tm(T_CONSUMER, st);
    hipLaunchKernelGGL(render_maps_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, a);
    SLS_LAUNCH_CHECK("render_maps_kernel");
    return SLS_OK;
}

// ---------------------------------------------------------------------------
// Mapper.densify's new rows (slam/mapper.py:104-137) for the pixels drawn: centre = the measured point in the model
// frame (utils/graphic_utils.py:26-66 at (c - .5, r - .5)), rotation = the quaternion (w, x, y, z; w >= 0) of the frame
// [d x h | d x (d x h) | d], d = the measured normal rotated into the model frame, h = e_x (e_y where d is within 1e-3
// of the x axis) — create_rotation_matrix_from_direction_vector_batch + matrix_to_quaternion
// (utils/general_utils.py:85-187), read off through the largest of the four components.  One thread per new surfel
// instead of ~65 torch kernels; fused_mapper.normal_aligned_quaternions is the torch form golden G3 / G7 pin.
//   c2w: 16 floats, inv(world_view_transform^T) row-major (DEVICE: torch made it);  mTf: 16 floats, model_T_frame (DEVICE)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void densify_rows_kernel(int n, int W, const int64_t *__restrict__ pix, const float *__restrict__ depth,
                                                           const float *__restrict__ normal, size_t P, const float2 *__restrict__ col_h,
                                                           const float2 *__restrict__ row_h, const float *__restrict__ c2w,
                                                           const float *__restrict__ mTf, float *__restrict__ xyz, float *__restrict__ quat)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t px = pix[i];
    const int r = (int)(px / W), c = (int)(px - (int64_t)r * W);
    const float2 cc = col_h[c], rr = row_h[r];
    const float rng = depth[px];
    const float r0 = cc.x * rr.x, r1 = cc.y * rr.x, r2 = rr.y;
    // rays @ R^T, x range, + t   (products and sums in torch's order: a row of the matmul, then the scale, then the offset)
    float q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        q[k] = __fadd_rn(__fmul_rn(rng, __fadd_rn(__fadd_rn(__fmul_rn(r0, c2w[4 * k + 0]), __fmul_rn(r1, c2w[4 * k + 1])), __fmul_rn(r2, c2w[4 * k + 2]))), c2w[4 * k + 3]);
    xyz[3 * i + 0] = q[0]; xyz[3 * i + 1] = q[1]; xyz[3 * i + 2] = q[2];
    const float n0 = normal[px], n1 = normal[P + px], n2 = normal[2 * P + px];
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = mTf[4 * k + 0] * n0 + mTf[4 * k + 1] * n1 + mTf[4 * k + 2] * n2;
    const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= dn; d[1] /= dn; d[2] /= dn;
    const bool near_x = fabsf(d[1]) < 1e-3f && fabsf(d[2]) < 1e-3f;
    const float h0 = near_x ? 0.0f : 1.0f, h1 = near_x ? 1.0f : 0.0f;
    float a[3] = { d[1] * 0.0f - d[2] * h1, d[2] * h0 - d[0] * 0.0f, d[0] * h1 - d[1] * h0 };      // d x h
    const float an = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] /= an; a[1] /= an; a[2] /= an;
    float b[3] = { d[1] * a[2] - d[2] * a[1], d[2] * a[0] - d[0] * a[2], d[0] * a[1] - d[1] * a[0] };   // d x a
    const float bn = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    b[0] /= bn; b[1] /= bn; b[2] /= bn;
    // R = [a | b | d] (columns)
    const float r00 = a[0], r11 = b[1], r22 = d[2];
    const float r01 = b[0], r02 = d[0], r10 = a[1], r12 = d[1], r20 = a[2], r21 = b[2];
    const float fs[4] = { 1.0f + r00 + r11 + r22, 1.0f + r00 - r11 - r22, 1.0f - r00 + r11 - r22, 1.0f - r00 - r11 + r22 };
    int k = 0;
    float best = -1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float m = sqrtf(fmaxf(fs[j], 0.0f)); if (m > best) { best = m; k = j; } }   // (first maximum, as argmax)
    float w, x, y, z;
    if (k == 0) { w = fs[0]; x = r21 - r12; y = r02 - r20; z = r10 - r01; }
    else if (k == 1) { w = r21 - r12; x = fs[1]; y = r10 + r01; z = r02 + r20; }
    else if (k == 2) { w = r02 - r20; x = r10 + r01; y = fs[2]; z = r12 + r21; }
    else { w = r10 - r01; x = r20 + r02; y = r21 + r12; z = fs[3]; }
    const float den = 2.0f * fmaxf(best, 0.1f);
    w /= den; x /= den; y /= den; z /= den;
    if (w < 0.0f) { w = -w; x = -x; y = -y; z = -z; }
    quat[4 * i + 0] = w; quat[4 * i + 1] = x; quat[4 * i + 2] = y; quat[4 * i + 3] = z;
}

// ---------------------------------------------------------------------------
// Where Mapper.densify may add surfels and with what weight (slam/mapper.py:51-102 at densify_threshold_egeom <= 0):
// candidate = valid pixel the model renders with alpha <= threshold (every valid pixel for a first keyframe: alpha =
// null); weight = the magnitude of the central differences of log(depth) (utils/graphic_utils.py:91-106: non-finite
// logs -> 0, each difference masked by the validity of its two end points, zero on the border) at the candidates, 0
// elsewhere.  stats: [#candidates, bits of max over the WHOLE image of the gradient, sum of the candidates' weights].
// One launch for compute_depth_gradient + the masks + three reductions (~25 torch kernels and three host syncs).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void densify_weights_kernel(int H, int W, const float *__restrict__ depth, const uint8_t *__restrict__ valid,
                                                              const float *__restrict__ alpha, float thr, float *__restrict__ w_out,
                                                              uint32_t *__restrict__ stats)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    float g = 0.0f, w = 0.0f;
    uint32_t cand = 0u;
    if (c < W && r < H) {
        const size_t pix = (size_t)r * W + c;
        auto logd = [&](size_t p) { const float l = logf(depth[p]); return (l == l && fabsf(l) != INFINITY) ? l : 0.0f; };
        if (r > 0 && r < H - 1 && c > 0 && c < W - 1) {
            const float dx = (logd(pix + W) - logd(pix - W)) * ((valid[pix + W] == 1 && valid[pix - W] == 1) ? 1.0f : 0.0f);
            const float dy = (logd(pix + 1) - logd(pix - 1)) * ((valid[pix + 1] == 1 && valid[pix - 1] == 1) ? 1.0f : 0.0f);
            g = sqrtf(dx * dx + dy * dy);
        }
        cand = (valid[pix] == 1 && (!alpha || alpha[pix] <= thr)) ? 1u : 0u;
        w = cand ? g : 0.0f;
        // (a candidate without gradient keeps a weight above every non-candidate's 0: torch.multinomial without replacement
        //  falls back to zero-weight entries once the positive ones are used up — they must be candidates, as in the
        //  reference, whose weight vector holds nothing else)
        w_out[pix] = cand ? fmaxf(g, 1.0e-30f) : 0.0f;
    }
    // workgroup sums -> three atomics per workgroup
    __shared__ float s_sum[4], s_max[4];
    __shared__ uint32_t s_cnt[4];
    float sum = w, mx = g;
    uint32_t cnt = cand;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); cnt += __shfl_xor(cnt, off, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wv] = sum; s_max[wv] = mx; s_cnt[wv] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[0], s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
        atomicMax(&stats[1], __float_as_uint(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]))));   // (>= 0: bit order = value order)
        atomicAdd(reinterpret_cast<float *>(&stats[2]), (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
    }
}

int launch_densify_weights(int H, int W, const float *depth, const uint8_t *valid, const float *alpha, float thr, float *w_out,
                           uint32_t *stats, hipStream_t st)
{
    SLS_HIP_CHECK(hipMemsetAsync(stats, 0, 4 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(densify_weights_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, H, W, depth, valid, alpha, thr,
                       w_out, stats);
    SLS_LAUNCH_CHECK("densify_weights_kernel");
    return SLS_OK;
}

int launch_densify_rows(int n, int H, int W, const int64_t *pix, const float *depth, const float *normal, const float *col_h,
                        const float *row_h, const float *c2w, const float *mTf, float *xyz, float *quat, hipStream_t st)
{
    if (n == 0) return SLS_OK;
    hipLaunchKernelGGL(densify_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, W, pix, depth, normal, (size_t)H * W,
                       (const float2 *)col_h, (const float2 *)row_h, c2w, mTf, xyz, quat);
    SLS_LAUNCH_CHECK("densify_rows_kernel");
    return SLS_OK;
}

}  // namespace sls
