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

}  // namespace sls
