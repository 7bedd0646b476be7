// sls_aligner.hip — frame-to-keyframe registration on spherical range images
// (the job of the reference's `gsaligner` submodule,
// slam/tracker.py:141-197; un-vendored, so everything below the call interface is
// THIS repository's specification; the tests hold a float64 NumPy
// restatement of it).
//
// Query frame Q (depth + back-projected points in the query frame), reference
// frame R (rendered keyframe: depth + points + normals in the reference frame),
// pose T = ref_T_query.  One Gauss-Newton iteration:
//   linearize : thread per query pixel: p' = T p, project p' into R (nearest pixel,
//               spherical model u = fx*az + cx, v = fy*el + cy, pixel = floor(.+1)),
//               gate (valid, |p'-q| <= max_distance, normal not grazing), then
//                 geometric  e_g = n . (p' - q),              J_g = [n, p' x n]
//                 range      e_r = |p'| - D_r(pixel) with the image gradient of D_r,
//                            J_r = [rhat - gu*du/dp - gv*dv/dp] [I, -[p']x]
//               Huber weights; 21 + 6 + 2 sums reduced in DOUBLE over the block (LDS) and
//               added to the system with one f64 atomic per block and term;
//   solve     : ONE thread: (H + damping I) xi = -b by Cholesky in double,
//               T <- exp(xi) T, statistics of the iteration, accumulators cleared.
// A whole align() is enqueued without a host sync; the caller reads the 72-byte result.
// HBM/latency bound: ~30 B per query pixel per iteration, P <= 131k pixels.
#include <cstring>
#include "sls_common.hpp"

namespace sls {

struct AlignCam {
    int H, W, wrap;
    float fx, fy, cx, cy;
};

static AlignCam make_aligncam(const SlsCamera &c)
{
    AlignCam a;
    a.H = c.H; a.W = c.W;
    a.fx = c.fx; a.fy = c.fy; a.cx = c.cx; a.cy = c.cy;
    const double period = fabs((double)c.fx) * 2.0 * 3.14159265358979323846;
    a.wrap = fabs(period - (double)c.W) <= 1.0 ? 1 : 0;
    return a;
}

constexpr int kSysTerms = 32;   // 21 (H upper) + 6 (b) + chi2 + inliers + valid query pixels + 2 spare

// ---------------------------------------------------------------------------
// Reference normals from the reference points: normalised cross product of the central
// differences (the construction of utils/graphic_utils.py:69-88), oriented towards the
// sensor, zero where any of the 4 neighbours or the pixel itself is invalid.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aligner_normals_kernel(AlignCam cam, const float *__restrict__ depth,
                                                              const float *__restrict__ points, float depth_min,
                                                              float *__restrict__ normals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int P = cam.H * cam.W;
    if (i >= P) return;
    const int r = i / cam.W, c = i - r * cam.W;
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
    int cl = c - 1, cr = c + 1;
    if (cam.wrap) { cl = (cl + cam.W) % cam.W; cr = cr % cam.W; }
    if (r > 0 && r < cam.H - 1 && cl >= 0 && cr < cam.W) {
        const int iu = (r + 1) * cam.W + c, id = (r - 1) * cam.W + c, il = r * cam.W + cl, ir = r * cam.W + cr;
        if (depth[i] > depth_min && depth[iu] > depth_min && depth[id] > depth_min && depth[il] > depth_min &&
            depth[ir] > depth_min) {
            const float u0 = points[3 * iu] - points[3 * id], u1 = points[3 * iu + 1] - points[3 * id + 1],
                        u2 = points[3 * iu + 2] - points[3 * id + 2];
            const float v0 = points[3 * ir] - points[3 * il], v1 = points[3 * ir + 1] - points[3 * il + 1],
                        v2 = points[3 * ir + 2] - points[3 * il + 2];
            float c0 = u1 * v2 - u2 * v1, c1 = u2 * v0 - u0 * v2, c2 = u0 * v1 - u1 * v0;
            const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
            if (len > 1e-12f) {
                const float inv = 1.0f / len;
                c0 *= inv; c1 *= inv; c2 *= inv;
                const float d = c0 * points[3 * i] + c1 * points[3 * i + 1] + c2 * points[3 * i + 2];
                const float s = d > 0.0f ? -1.0f : 1.0f;       // face the sensor at the origin
                n0 = s * c0; n1 = s * c1; n2 = s * c2;
            }
        }
    }
    normals[3 * i] = n0; normals[3 * i + 1] = n1; normals[3 * i + 2] = n2;
}

__device__ __forceinline__ float huber_w(float e, float delta)
{
    const float a = fabsf(e);
    return a <= delta ? 1.0f : delta / a;
}

// ---------------------------------------------------------------------------
// linearize: pose = 12 floats on the device (row-major 3x4 [R|t]), sys = kSysTerms doubles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aligner_linearize_kernel(
    AlignCam cam, SlsAlignerParams prm, const float *__restrict__ ref_depth, const float *__restrict__ ref_points,
    const float *__restrict__ ref_normals, const float *__restrict__ q_depth, const float *__restrict__ q_points,
    const float *__restrict__ pose, double *__restrict__ sys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int P = cam.H * cam.W;
    float acc[kSysTerms];
#pragma unroll
    for (int k = 0; k < kSysTerms; ++k) acc[k] = 0.0f;
    if (i < P) {
        const float dq = q_depth[i];
        if (dq > prm.depth_min && dq <= prm.depth_max) {
            acc[29] = 1.0f;   // valid query pixel
            const float px = q_points[3 * i], py = q_points[3 * i + 1], pz = q_points[3 * i + 2];
            const float x = pose[0] * px + pose[1] * py + pose[2] * pz + pose[3];
            const float y = pose[4] * px + pose[5] * py + pose[6] * pz + pose[7];
            const float z = pose[8] * px + pose[9] * py + pose[10] * pz + pose[11];
            const float rxy2 = x * x + y * y, rho2 = rxy2 + z * z;
            const float rxy = sqrtf(rxy2), rho = sqrtf(rho2);
            if (rho > prm.depth_min && rxy > 1e-6f) {
                const float az = atan2f(y, x), el = atan2f(z, rxy);
                const float u = cam.fx * az + cam.cx, v = cam.fy * el + cam.cy;
                int c = (int)floorf(u + 1.0f), r = (int)floorf(v + 1.0f);
                if (cam.wrap) c = ((c % cam.W) + cam.W) % cam.W;
                if (c >= 0 && c < cam.W && r >= 0 && r < cam.H) {
                    const int j = r * cam.W + c;
                    const float dr = ref_depth[j];
                    const float n0 = ref_normals[3 * j], n1 = ref_normals[3 * j + 1], n2 = ref_normals[3 * j + 2];
                    const bool has_n = (n0 != 0.0f) || (n1 != 0.0f) || (n2 != 0.0f);
                    if (dr > prm.depth_min && dr <= prm.depth_max && has_n) {
                        const float d0 = x - ref_points[3 * j], d1 = y - ref_points[3 * j + 1], d2 = z - ref_points[3 * j + 2];
                        const float dist2 = d0 * d0 + d1 * d1 + d2 * d2;
                        const float inv_rho = 1.0f / rho;
                        const float cosang = -(n0 * x + n1 * y + n2 * z) * inv_rho;   // normal vs viewing ray
                        if (dist2 <= prm.max_distance * prm.max_distance && cosang >= prm.min_cos_angle) {
                            float J[6], e, w;
                            // ---- geometric (point to plane) ----
                            e = n0 * d0 + n1 * d1 + n2 * d2;
                            w = huber_w(e, prm.huber_delta);
                            J[0] = n0; J[1] = n1; J[2] = n2;
                            J[3] = y * n2 - z * n1; J[4] = z * n0 - x * n2; J[5] = x * n1 - y * n0;
                            int t = 0;
#pragma unroll
                            for (int a = 0; a < 6; ++a) {
#pragma unroll
                                for (int b = a; b < 6; ++b) acc[t++] += w * J[a] * J[b];
                                acc[21 + a] += w * J[a] * e;
                            }
                            acc[27] += w * e * e;
                            acc[28] += 1.0f;
                            // ---- range image term ----
                            if (prm.range_weight > 0.0f) {
                                int cl = c - 1, cr = c + 1;
                                if (cam.wrap) { cl = (cl + cam.W) % cam.W; cr = cr % cam.W; }
                                float gu = 0.0f, gv = 0.0f;
                                if (cl >= 0 && cr < cam.W) {
                                    const float a = ref_depth[r * cam.W + cl], b = ref_depth[r * cam.W + cr];
                                    if (a > prm.depth_min && b > prm.depth_min) gu = 0.5f * (b - a);
                                }
                                if (r > 0 && r < cam.H - 1) {
                                    const float a = ref_depth[(r - 1) * cam.W + c], b = ref_depth[(r + 1) * cam.W + c];
                                    if (a > prm.depth_min && b > prm.depth_min) gv = 0.5f * (b - a);
                                }
                                const float er = rho - dr;
                                const float wr = prm.range_weight * huber_w(er, prm.range_huber);
                                // d rho/dp = p/rho ; du/dp = fx (-y, x, 0)/rxy^2 ; dv/dp = fy (-xz, -yz, rxy^2)/(rxy rho^2)
                                const float iu = cam.fx / rxy2, iv = cam.fy / (rxy * rho2);
                                const float g0 = x * inv_rho - gu * (-y * iu) - gv * (-x * z * iv);
                                const float g1 = y * inv_rho - gu * (x * iu) - gv * (-y * z * iv);
                                const float g2 = z * inv_rho - gv * (rxy2 * iv);
                                J[0] = g0; J[1] = g1; J[2] = g2;
                                J[3] = y * g2 - z * g1; J[4] = z * g0 - x * g2; J[5] = x * g1 - y * g0;
                                t = 0;
#pragma unroll
                                for (int a = 0; a < 6; ++a) {
#pragma unroll
                                    for (int b = a; b < 6; ++b) acc[t++] += wr * J[a] * J[b];
                                    acc[21 + a] += wr * J[a] * er;
                                }
                                acc[27] += wr * er * er;
                            }
                        }
                    }
                }
            }
        }
    }
    // block reduction in double: lanes -> wave (shuffles) -> block (LDS) -> one f64 atomic per term
    __shared__ double s_part[4][kSysTerms];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 30; ++k) {
        double v = (double)acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_part[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 30) {
        const double v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (v != 0.0) atomicAdd(&sys[threadIdx.x], v);
    }
}

// ---------------------------------------------------------------------------
// solve + pose update, one thread.  result: SlsAlignerResult on the device.
// ---------------------------------------------------------------------------
__global__ void aligner_solve_kernel(SlsAlignerParams prm, double *__restrict__ sys, float *__restrict__ pose,
                                     SlsAlignerResult *__restrict__ res, int iteration, int update_pose)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double H[6][6], b[6];
    int t = 0;
    for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) { H[a][c] = sys[t]; H[c][a] = sys[t]; ++t; }
    for (int a = 0; a < 6; ++a) b[a] = sys[21 + a];
    const double chi2 = sys[27], inl = sys[28], nval = sys[29];
    res->chi2 = (float)chi2;
    res->inliers = (int32_t)inl;
    res->valid_query = (int32_t)nval;
    res->fitness = nval > 0.0 ? (float)(inl / nval) : 0.0f;
    res->iterations = update_pose ? iteration + 1 : iteration;
    for (int k = 0; k < kSysTerms; ++k) sys[k] = 0.0;
    if (update_pose && inl >= (double)prm.min_inliers) {
        // Cholesky of H + damping * I (lower), then two triangular solves for xi = -(H)^-1 b
        double L[6][6];
        bool ok = true;
        for (int a = 0; a < 6; ++a) H[a][a] += (double)prm.damping;
        for (int a = 0; a < 6 && ok; ++a)
            for (int c = 0; c <= a; ++c) {
                double s = H[a][c];
                for (int k = 0; k < c; ++k) s -= L[a][k] * L[c][k];
                if (a == c) {
                    if (s <= 0.0) { ok = false; break; }
                    L[a][a] = sqrt(s);
                } else {
                    L[a][c] = s / L[c][c];
                }
            }
        if (ok) {
            double yv[6], xi[6];
            for (int a = 0; a < 6; ++a) {
                double s = -b[a];
                for (int k = 0; k < a; ++k) s -= L[a][k] * yv[k];
                yv[a] = s / L[a][a];
            }
            for (int a = 5; a >= 0; --a) {
                double s = yv[a];
                for (int k = a + 1; k < 6; ++k) s -= L[k][a] * xi[k];
                xi[a] = s / L[a][a];
            }
            // exp(xi) for xi = (v, w): R = Rodrigues(w), t = V(w) v
            const double wx = xi[3], wy = xi[4], wz = xi[5];
            const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
            double A, B, C;
            if (th < 1e-6) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
            else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; C = (1.0 - A) / th2; }
            const double K[3][3] = { { 0, -wz, wy }, { wz, 0, -wx }, { -wy, wx, 0 } };
            double K2[3][3], dR[3][3], V[3][3];
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) {
                    K2[a][c] = 0.0;
                    for (int k = 0; k < 3; ++k) K2[a][c] += K[a][k] * K[k][c];
                }
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) {
                    const double I = a == c ? 1.0 : 0.0;
                    dR[a][c] = I + A * K[a][c] + B * K2[a][c];
                    V[a][c] = I + B * K[a][c] + C * K2[a][c];
                }
            double dt[3];
            for (int a = 0; a < 3; ++a) dt[a] = V[a][0] * xi[0] + V[a][1] * xi[1] + V[a][2] * xi[2];
            double Rn[3][3], tn[3];
            for (int a = 0; a < 3; ++a) {
                for (int c = 0; c < 3; ++c) {
                    Rn[a][c] = 0.0;
                    for (int k = 0; k < 3; ++k) Rn[a][c] += dR[a][k] * (double)pose[4 * k + c];
                }
                tn[a] = dt[a];
                for (int k = 0; k < 3; ++k) tn[a] += dR[a][k] * (double)pose[4 * k + 3];
            }
            for (int a = 0; a < 3; ++a) {
                for (int c = 0; c < 3; ++c) pose[4 * a + c] = (float)Rn[a][c];
                pose[4 * a + 3] = (float)tn[a];
            }
            double n2 = 0.0;
            for (int a = 0; a < 6; ++a) n2 += xi[a] * xi[a];
            res->last_step = (float)sqrt(n2);
        } else {
            res->last_step = -1.0f;      // system not positive definite: pose kept
        }
    }
    for (int k = 0; k < 12; ++k) res->pose[k] = pose[k];
}

}  // namespace sls

using namespace sls;

extern "C" {

size_t sls_aligner_workspace_bytes(void) { return sizeof(double) * kSysTerms + sizeof(float) * 16; }

int sls_aligner_normals(const SlsCamera *cam, const float *depth, const float *points, float depth_min,
                        float *normals, void *stream)
{
    SLS_REQUIRE(cam && depth && points && normals, "null pointer");
    const AlignCam ac = make_aligncam(*cam);
    const int P = ac.H * ac.W;
    hipLaunchKernelGGL(aligner_normals_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, ac, depth,
                       points, depth_min, normals);
    SLS_LAUNCH_CHECK("aligner_normals_kernel");
    return SLS_OK;
}

// One linearisation at the HOST pose T (row-major 4x4 ref_T_query): sys_out gets 32 doubles
// [H upper (21) | b (6) | chi2 | inliers | valid query pixels | 0 0] on the device.
int sls_aligner_linearize(const SlsCamera *cam, const SlsAlignerParams *prm, const float *ref_depth,
                          const float *ref_points, const float *ref_normals, const float *query_depth,
                          const float *query_points, const float *T_host, void *workspace, double *sys_out,
                          void *stream)
{
    SLS_REQUIRE(cam && prm && ref_depth && ref_points && ref_normals && query_depth && query_points && T_host &&
                    workspace && sys_out,
                "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const AlignCam ac = make_aligncam(*cam);
    const int P = ac.H * ac.W;
    float *pose = (float *)((double *)workspace + kSysTerms);
    SLS_HIP_CHECK(hipMemcpyAsync(pose, T_host, sizeof(float) * 12, hipMemcpyHostToDevice, st));
    SLS_HIP_CHECK(hipMemsetAsync(sys_out, 0, sizeof(double) * kSysTerms, st));
    hipLaunchKernelGGL(aligner_linearize_kernel, dim3((P + 255) / 256), dim3(256), 0, st, ac, *prm, ref_depth,
                       ref_points, ref_normals, query_depth, query_points, (const float *)pose, sys_out);
    SLS_LAUNCH_CHECK("aligner_linearize_kernel");
    return SLS_OK;
}

// prm->num_iterations Gauss-Newton iterations from T_host, all enqueued back to back;
// result_dev (device) holds the final pose, fitness and statistics of the LAST linearisation.
int sls_aligner_align(const SlsCamera *cam, const SlsAlignerParams *prm, const float *ref_depth,
                      const float *ref_points, const float *ref_normals, const float *query_depth,
                      const float *query_points, const float *T_host, void *workspace,
                      SlsAlignerResult *result_dev, void *stream)
{
    SLS_REQUIRE(cam && prm && ref_depth && ref_points && ref_normals && query_depth && query_points && T_host &&
                    workspace && result_dev,
                "null pointer");
    SLS_REQUIRE(prm->num_iterations >= 0 && prm->num_iterations <= 1000, "bad iteration count");
    hipStream_t st = (hipStream_t)stream;
    const AlignCam ac = make_aligncam(*cam);
    const int P = ac.H * ac.W;
    double *sys = (double *)workspace;
    float *pose = (float *)(sys + kSysTerms);
    SLS_HIP_CHECK(hipMemcpyAsync(pose, T_host, sizeof(float) * 12, hipMemcpyHostToDevice, st));
    SLS_HIP_CHECK(hipMemsetAsync(sys, 0, sizeof(double) * kSysTerms, st));
    SLS_HIP_CHECK(hipMemsetAsync(result_dev, 0, sizeof(SlsAlignerResult), st));
    // the last pass only evaluates (fitness / chi2 of the final pose)
    for (int it = 0; it <= prm->num_iterations; ++it) {
        hipLaunchKernelGGL(aligner_linearize_kernel, dim3((P + 255) / 256), dim3(256), 0, st, ac, *prm, ref_depth,
                           ref_points, ref_normals, query_depth, query_points, (const float *)pose, sys);
        hipLaunchKernelGGL(aligner_solve_kernel, dim3(1), dim3(1), 0, st, *prm, sys, pose, result_dev, it,
                           it < prm->num_iterations ? 1 : 0);
    }
    SLS_LAUNCH_CHECK("aligner kernels");
    return SLS_OK;
}

}  // extern "C"
