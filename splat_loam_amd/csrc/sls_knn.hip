// sls_knn.hip — simple-knn's distCUDA2 for MI355X:
// out[i] = mean of the squared distances from point i to its 3 nearest
// other points (slam/mapper.py:113-115, scene/gaussian_model.py:77-81).
//
// Exact 3-NN, built for LiDAR-like (very non-uniform) clouds:
//   1. bounding cube, 30-bit Morton code per point;
//   2. wave64 LSD radix sort of (code, index) — the sorter of sls_sort.hip;
//   3. points gathered in Morton order (float4: xyz + original index) and
//      axis-aligned boxes over runs of 256 consecutive points;
//   4. one thread per point (neighbouring lanes = neighbouring points): scan
//      the own box first for a tight bound, then walk all boxes with a
//      wave-uniform box index (box bounds come through the scalar cache) and
//      scan only boxes whose distance lower bound beats the current third-best.
// Squared distances use the fixed expression fma(dz,dz,fma(dy,dy,dx*dx)), so
// the result is reproducible bit for bit by a CPU brute force.
#include <float.h>

#include "sls_common.hpp"

namespace sls {

size_t sort_scratch_bytes(uint64_t R);
int radix_sort_pairs_u32(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                         const uint32_t *count_ptr, uint32_t cap, int nbits, void *scratch, size_t scratch_bytes,
                         int *result_in_tmp, hipStream_t st);

constexpr int kKnnBox = 256;

__device__ __forceinline__ uint32_t f2ord(float f)
{   // monotone float -> uint mapping
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}

__global__ void knn_init_bbox_kernel(uint32_t *bbox, uint32_t M)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xFFFFFFFFu;       // min (ordered domain)
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;           // max
    else if (threadIdx.x == 6) bbox[6] = M;                     // item count for the sorter
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(int M, const float *__restrict__ xyz, uint32_t *bbox)
{
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * i + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            atomicMin(&bbox[k], f2ord(mn[k]));
            atomicMax(&bbox[3 + k], f2ord(mx[k]));
        }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{   // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(int M, const float *__restrict__ xyz,
                                                         const uint32_t *__restrict__ bbox,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float mnx = ord2f(bbox[0]), mny = ord2f(bbox[1]), mnz = ord2f(bbox[2]);
    const float ext = fmaxf(fmaxf(ord2f(bbox[3]) - mnx, ord2f(bbox[4]) - mny), fmaxf(ord2f(bbox[5]) - mnz, 1e-30f));
    const float s = 1024.0f / ext;
    const uint32_t qx = (uint32_t)fminf(fmaxf((xyz[3 * i] - mnx) * s, 0.0f), 1023.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((xyz[3 * i + 1] - mny) * s, 0.0f), 1023.0f);
    const uint32_t qz = (uint32_t)fminf(fmaxf((xyz[3 * i + 2] - mnz) * s, 0.0f), 1023.0f);
    keys[i] = spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2);
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void knn_gather_boxes_kernel(int M, const float *__restrict__ xyz,
                                                               const uint32_t *__restrict__ sorted_idx,
                                                               float4 *__restrict__ pts, float4 *__restrict__ boxes)
{
    // one block per box of kKnnBox == blockDim.x points
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int j = blockIdx.x * kKnnBox + threadIdx.x;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (j < M) {
        const uint32_t i = sorted_idx[j];
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        pts[j] = make_float4(x, y, z, __uint_as_float(i));
        mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s_mn[wave][k] = mn[k]; s_mx[wave][k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k)
            for (int w = 1; w < 4; ++w) { s_mn[0][k] = fminf(s_mn[0][k], s_mn[w][k]); s_mx[0][k] = fmaxf(s_mx[0][k], s_mx[w][k]); }
        boxes[2 * blockIdx.x] = make_float4(s_mn[0][0], s_mn[0][1], s_mn[0][2], 0.0f);
        boxes[2 * blockIdx.x + 1] = make_float4(s_mx[0][0], s_mx[0][1], s_mx[0][2], 0.0f);
    }
}

__device__ __forceinline__ void best3_update(float d2, float &b0, float &b1, float &b2)
{
    if (d2 < b2) {
        if (d2 < b1) {
            b2 = b1;
            if (d2 < b0) { b1 = b0; b0 = d2; } else b1 = d2;
        } else b2 = d2;
    }
}

__global__ __launch_bounds__(256) void knn_query_kernel(int M, int nboxes, const float4 *__restrict__ pts,
                                                        const float4 *__restrict__ boxes, float *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < M;
    const float4 me = live ? pts[j] : make_float4(0, 0, 0, 0);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int own = j / kKnnBox;
    if (live) {
        const int lo = own * kKnnBox, hi = min(lo + kKnnBox, M);
        for (int k = lo; k < hi; ++k) {
            if (k == j) continue;
            const float4 p = pts[k];
            const float dx = p.x - me.x, dy = p.y - me.y, dz = p.z - me.z;
            best3_update(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), b0, b1, b2);
        }
    }
    for (int b = 0; b < nboxes; ++b) {   // wave-uniform walk over all boxes
        const float4 mn = boxes[2 * b], mx = boxes[2 * b + 1];
        const float ex = fmaxf(fmaxf(mn.x - me.x, me.x - mx.x), 0.0f);
        const float ey = fmaxf(fmaxf(mn.y - me.y, me.y - mx.y), 0.0f);
        const float ez = fmaxf(fmaxf(mn.z - me.z, me.z - mx.z), 0.0f);
        const float lb = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
        if (live && b != own && lb * 0.99999f <= b2) {
            const int lo = b * kKnnBox, hi = min(lo + kKnnBox, M);
            for (int k = lo; k < hi; ++k) {
                const float4 p = pts[k];
                const float dx = p.x - me.x, dy = p.y - me.y, dz = p.z - me.z;
                best3_update(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), b0, b1, b2);
            }
        }
    }
    if (live) out[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

// scratch layout (all 256-byte aligned)
struct KnnScratch {
    uint32_t *bbox;
    uint32_t *keys, *keys_tmp;
    uint32_t *vals, *vals_tmp;
    float4 *pts, *boxes;
    void *sort;
    size_t sort_bytes, total;
};

static KnnScratch knn_layout(int M, void *base)
{
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    KnnScratch s;
    char *p = (char *)base;
    size_t off = 0;
    const int nboxes = (M + kKnnBox - 1) / kKnnBox;
    s.bbox = (uint32_t *)(p + off); off += al(8 * sizeof(uint32_t));
    s.keys = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.keys_tmp = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.vals = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.vals_tmp = (uint32_t *)(p + off); off += al(sizeof(uint32_t) * (size_t)M);
    s.pts = (float4 *)(p + off); off += al(sizeof(float4) * (size_t)M);
    s.boxes = (float4 *)(p + off); off += al(sizeof(float4) * 2 * (size_t)nboxes);
    s.sort = (void *)(p + off);
    s.sort_bytes = sort_scratch_bytes((uint64_t)M);
    off += al(s.sort_bytes);
    s.total = off;
    return s;
}

size_t knn_scratch_bytes(int M) { return M > 0 ? knn_layout(M, nullptr).total : 0; }

int launch_knn(int M, const float *xyz, float *out, void *scratch, size_t scratch_bytes, hipStream_t st)
{
    const KnnScratch s = knn_layout(M, scratch);
    if (scratch_bytes < s.total) {
        set_error("knn scratch too small: %zu < %zu", scratch_bytes, s.total);
        return SLS_E_SCRATCH;
    }
    if (((uintptr_t)scratch & 255) != 0) {
        set_error("knn scratch must be 256-byte aligned");
        return SLS_E_ARG;
    }
    const int nb = (M + 255) / 256;
    ScopedTimer tm(T_KNN, st);
    hipLaunchKernelGGL(knn_init_bbox_kernel, dim3(1), dim3(64), 0, st, s.bbox, (uint32_t)M);
    SLS_LAUNCH_CHECK("knn_init_bbox_kernel");
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, M, xyz, s.bbox);
    SLS_LAUNCH_CHECK("knn_bbox_kernel");
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, st, M, xyz, s.bbox, s.keys, s.vals);
    SLS_LAUNCH_CHECK("knn_morton_kernel");
    int which = 0;
    int rc = radix_sort_pairs_u32(s.keys, s.vals, s.keys_tmp, s.vals_tmp, s.bbox + 6, (uint32_t)M, 30, s.sort,
                                  s.sort_bytes, &which, st);
    if (rc) return rc;
    const int nboxes = (M + kKnnBox - 1) / kKnnBox;
    hipLaunchKernelGGL(knn_gather_boxes_kernel, dim3(nboxes), dim3(kKnnBox), 0, st, M, xyz,
                       which ? s.vals_tmp : s.vals, s.pts, s.boxes);
    SLS_LAUNCH_CHECK("knn_gather_boxes_kernel");
    hipLaunchKernelGGL(knn_query_kernel, dim3(nb), dim3(256), 0, st, M, nboxes, s.pts, s.boxes, out);
    SLS_LAUNCH_CHECK("knn_query_kernel");
    return SLS_OK;
}

}  // namespace sls
