// Pose optimisation: the per-ray camera transform of pc_nerf/ba_pipeline.py:85-92 and the gradient of the view embedding.
//
// configs/bup20/best.yaml trains the camera extrinsics in EVERY step (optimize_extrinsics, extrinsics_epoch_end 900 > epochs 800:
// pc_nerf/trainer.py:308), so every step maps the camera-frame base rays of its 6 images to the world through the current parameters
// (`transform_rays`: kaolin's inv_transform_rays on the 'matrix_6dof_rotation' camera backend, then re-normalised directions) and sends
// d loss / d origins, d loss / d dirs back through that map.  As tensor ops this is ~25 launches forward and ~60 backward on [N,3]
// tensors (N = 24 576): 0.8 ms of host-paced 5 us kernels around a 3 ms post-prune step.  Here: one launch each way.
//
// Parametrisation (restated from the public kaolin sources, PARITY UNPINNED - pagnerf_amd/ba_pipeline.py):
//   params [C,9] = (a1[3], a2[3], t[3]);  b1 = a1/|a1|, b2 = normalise(a2 - (b1.a2) b1), b3 = b1 x b2;  R rows = (b1, b2, b3)
//   origins_w = R^T (o_c - t) = sum_k (o_c - t)[k] R[k],   dirs_w = normalise(sum_k d_c[k] R[k])
// Products and sums in the order of the tensor-op form (ba_pipeline.transform_rays_indexed: three scaled rows); compiled with
// -ffp-contract=off, so no FMA contraction changes them.
#include "common.h"

namespace {

struct Rot {
    float b[3][3];        // rows b1, b2, b3
    float n1, n2, s;      // |a1|, |a2 - s b1|, s = b1 . a2
};

__device__ __forceinline__ float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// rotation_6d_to_matrix of pagnerf_amd/ba_pipeline.py (Gram-Schmidt, Zhou et al. 2019)
__device__ __forceinline__ Rot rotation(const float *p) {
    Rot r;
    r.n1 = __fsqrt_rn(dot3(p, p));
#pragma unroll
    for (int j = 0; j < 3; ++j) r.b[0][j] = p[j] / r.n1;
    r.s = dot3(r.b[0], p + 3);
    float q[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) q[j] = p[3 + j] - r.s * r.b[0][j];
    r.n2 = __fsqrt_rn(dot3(q, q));
#pragma unroll
    for (int j = 0; j < 3; ++j) r.b[1][j] = q[j] / r.n2;
    cross3(r.b[0], r.b[1], r.b[2]);
    return r;
}

__global__ __launch_bounds__(256) void pose_rays_fwd_kernel(const float *__restrict__ params, int64_t C, const int32_t *__restrict__ cam, int64_t rays_per_entry,
                                                            const float *__restrict__ oc, const float *__restrict__ dc, int64_t N, float *__restrict__ ow,
                                                            float *__restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int64_t c = cam[i / rays_per_entry];
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    const float *p = params + c * 9;
    const Rot r = rotation(p);
    float v[3], d[3], u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = oc[i * 3 + k] - p[6 + k];
        d[k] = dc[i * 3 + k];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ow[i * 3 + j] = (v[0] * r.b[0][j] + v[1] * r.b[1][j]) + v[2] * r.b[2][j];
        u[j] = (d[0] * r.b[0][j] + d[1] * r.b[1][j]) + d[2] * r.b[2][j];
    }
    const float n = __fsqrt_rn(dot3(u, u));
#pragma unroll
    for (int j = 0; j < 3; ++j) dw[i * 3 + j] = u[j] / n;
}

// utils/outlier_rejection.py:74-97 (`rays_to_3d_points`, pc_nerf/trainer.py:508-518): the step's base rays unprojected by the rendered depth and mapped to the
// world by the same inverse camera transform - points = R^T (o_c - t) + R^T (d_c depth) = sum_k (o_c - t + d_c depth)[k] R[k], in the op order of
// ba_pipeline.rays_to_3d_points_indexed (as tensor ops: ~20 launches on [N,3] tensors in front of the assignment's cost matrix).
__global__ __launch_bounds__(256) void pose_points_kernel(const float *__restrict__ params, int64_t C, const int32_t *__restrict__ cam, int64_t rays_per_entry,
                                                          const float *__restrict__ oc, const float *__restrict__ dc, const float *__restrict__ depth, int64_t N,
                                                          float *__restrict__ points) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int64_t c = cam[i / rays_per_entry];
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    const float *p = params + c * 9;
    const Rot r = rotation(p);
    const float t = depth[i];
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = (oc[i * 3 + k] - p[6 + k]) + dc[i * 3 + k] * t;
#pragma unroll
    for (int j = 0; j < 3; ++j) points[i * 3 + j] = (v[0] * r.b[0][j] + v[1] * r.b[1][j]) + v[2] * r.b[2][j];
}

// Two launches, fixed summation order (bitwise reproducible).  Stage 1: workgroup (camera c, slice s) walks slice s of the rays (N / POSE_SLICES
// consecutive rays) and sums the contributions of camera c's rays to d R (9) and d t (3): thread-strided partial sums, then a tree over the
// workgroup -> part[c][s][12].  Stage 2: one thread per camera adds its slices in order and applies the chain through the Gram-Schmidt
// construction once.  Cameras without a ray in the batch get a zero row.  (One workgroup per camera over ALL rays - 6 workgroups on the chip for
// a best.yaml step - took 67 us; C x N camera-index reads, 42 x 24 576 on BUP20, are noise either way.)
constexpr int POSE_SLICES = 32;
__global__ __launch_bounds__(256) void pose_rays_bwd_kernel(const float *__restrict__ params, const int32_t *__restrict__ cam, int64_t rays_per_entry,
                                                            const float *__restrict__ oc, const float *__restrict__ dc, int64_t N,
                                                            const float *__restrict__ g_o, const float *__restrict__ g_d, float *__restrict__ part) {
    const int c = blockIdx.x, sl = blockIdx.y;
    const float *p = params + (int64_t)c * 9;
    const Rot r = rotation(p);
    float acc[12];        // d R[k][j] at 3k + j, d t at 9..11
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0f;
    const int64_t per = (N + POSE_SLICES - 1) / POSE_SLICES;
    const int64_t lo = sl * per, hi = lo + per < N ? lo + per : N;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        // the forward's clamp: an index outside [0, C) renders through camera 0 / C - 1, so its gradient belongs to that camera too
        int cc = cam[i / rays_per_entry];
        cc = cc < 0 ? 0 : (cc >= (int)gridDim.x ? (int)gridDim.x - 1 : cc);
        if (cc != c) continue;
        float v[3], d[3], u[3], go[3], gd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] = oc[i * 3 + k] - p[6 + k];
            d[k] = dc[i * 3 + k];
            go[k] = g_o ? g_o[i * 3 + k] : 0.0f;
            gd[k] = g_d ? g_d[i * 3 + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) u[j] = (d[0] * r.b[0][j] + d[1] * r.b[1][j]) + d[2] * r.b[2][j];
        const float n = __fsqrt_rn(dot3(u, u));
        float w[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) w[j] = u[j] / n;
        const float wg = dot3(w, gd);
        float gu[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) gu[j] = (gd[j] - w[j] * wg) / n;          // through dirs = u / |u|
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[3 * k + j] += v[k] * go[j] + d[k] * gu[j];
            acc[9 + k] -= dot3(r.b[k], go);                                    // v = o_c - t
        }
    }
    __shared__ float red[256][13];
#pragma unroll
    for (int q = 0; q < 12; ++q) red[threadIdx.x][q] = acc[q];
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int q = 0; q < 12; ++q) red[threadIdx.x][q] += red[threadIdx.x + s][q];
        __syncthreads();
    }
    if (threadIdx.x < 12) part[((int64_t)c * POSE_SLICES + sl) * 12 + threadIdx.x] = red[0][threadIdx.x];
}

__global__ __launch_bounds__(64) void pose_rays_bwd_finish_kernel(const float *__restrict__ params, int64_t C, const float *__restrict__ part,
                                                                  float *__restrict__ d_params) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float *p = params + c * 9;
    const Rot r = rotation(p);
    float G[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) G[q] = 0.0f;
    for (int s = 0; s < POSE_SLICES; ++s)
#pragma unroll
        for (int q = 0; q < 12; ++q) G[q] += part[(c * POSE_SLICES + s) * 12 + q];
    float gb1[3] = {G[0], G[1], G[2]}, gb2[3] = {G[3], G[4], G[5]}, gb3[3] = {G[6], G[7], G[8]};
    float t1[3], t2[3];
    cross3(r.b[1], gb3, t1);           // b3 = b1 x b2:  d b1 . (b2 x g3),  d b2 . (g3 x b1)
    cross3(gb3, r.b[0], t2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gb1[j] += t1[j];
        gb2[j] += t2[j];
    }
    const float b2g = dot3(r.b[1], gb2);
    float gp[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gp[j] = (gb2[j] - r.b[1][j] * b2g) / r.n2;     // b2 = q / |q|
    const float b1gp = dot3(r.b[0], gp);
    float *o = d_params + c * 9;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o[3 + j] = gp[j] - r.b[0][j] * b1gp;                                  // q = a2 - (b1 . a2) b1
        gb1[j] += -(p[3 + j] * b1gp) - r.s * gp[j];
    }
    const float b1g = dot3(r.b[0], gb1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o[j] = (gb1[j] - r.b[0][j] * b1g) / r.n1;                              // b1 = a1 / |a1|
        o[6 + j] = G[9 + j];
    }
}

// d loss / d dirs through pag_view_embed: out = (-d, sin(-d 2^k), cos(-d 2^k)), frequency-major.
__global__ __launch_bounds__(256) void view_embed_bwd_kernel(const float *__restrict__ dirs, int64_t R, int n_freq, int width, const float *__restrict__ g,
                                                             float *__restrict__ d_dirs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * 3) return;
    const int64_t r = i / 3;
    const int ax = (int)(i - r * 3);
    const float *gr = g + r * width;
    const float x = dirs[i];
    float a = -gr[ax];
    for (int k = 0; k < n_freq; ++k) {
        const float sc = (float)(1 << k);
        const float w = -x * sc;
        a += sc * (sinf(w) * gr[3 + 3 * n_freq + 3 * k + ax] - cosf(w) * gr[3 + 3 * k + ax]);
    }
    d_dirs[i] = a;
}
}  // namespace

static int pose_check(const char *name, const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *oc, const float *dc, int64_t N) {
    PAG_CHECK_ARG(N >= 0 && C >= 1 && C <= 65535 && rays_per_entry >= 1, "%s: N %lld < 0, cameras %lld not in [1,65535] or rays_per_entry %lld < 1", name,
                  (long long)N, (long long)C, (long long)rays_per_entry);
    PAG_CHECK_ARG(N == 0 || (params && cam && oc && dc), "%s: NULL input", name);
    return PAG_OK;
}

extern "C" int pag_pose_rays_fwd(const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *origins_c, const float *dirs_c,
                                 int64_t N, float *origins_w, float *dirs_w, void *stream) {
    int rc = pose_check("pag_pose_rays_fwd", params, C, cam, rays_per_entry, origins_c, dirs_c, N);
    if (rc) return rc;
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(origins_w && dirs_w, "pag_pose_rays_fwd: NULL output");
    hipLaunchKernelGGL(pose_rays_fwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, C, cam, rays_per_entry, origins_c,
                       dirs_c, N, origins_w, dirs_w);
    PAG_CHECK_LAUNCH("pag_pose_rays_fwd");
    return PAG_OK;
}

extern "C" int pag_pose_points(const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *origins_c, const float *dirs_c,
                               const float *depth, int64_t N, float *points, void *stream) {
    int rc = pose_check("pag_pose_points", params, C, cam, rays_per_entry, origins_c, dirs_c, N);
    if (rc) return rc;
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(depth && points, "pag_pose_points: NULL depth / output");
    hipLaunchKernelGGL(pose_points_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, C, cam, rays_per_entry, origins_c, dirs_c,
                       depth, N, points);
    PAG_CHECK_LAUNCH("pag_pose_points");
    return PAG_OK;
}

extern "C" int64_t pag_pose_rays_bwd_workspace_bytes(int64_t C) { return C > 0 ? C * POSE_SLICES * 12 * (int64_t)sizeof(float) : 0; }

extern "C" int pag_pose_rays_bwd(const float *params, int64_t C, const int32_t *cam, int64_t rays_per_entry, const float *origins_c, const float *dirs_c,
                                 int64_t N, const float *g_origins, const float *g_dirs, float *d_params, void *workspace, int64_t workspace_bytes,
                                 void *stream) {
    int rc = pose_check("pag_pose_rays_bwd", params, C, cam, rays_per_entry, origins_c, dirs_c, N);
    if (rc) return rc;
    PAG_CHECK_ARG(d_params && params, "pag_pose_rays_bwd: NULL params / d_params");
    PAG_CHECK_ARG(workspace && workspace_bytes >= pag_pose_rays_bwd_workspace_bytes(C), "pag_pose_rays_bwd: workspace smaller than pag_pose_rays_bwd_workspace_bytes(C)");
    hipLaunchKernelGGL(pose_rays_bwd_kernel, dim3((unsigned)C, POSE_SLICES), dim3(256), 0, (hipStream_t)stream, params, cam, rays_per_entry, origins_c, dirs_c, N,
                       g_origins, g_dirs, (float *)workspace);
    hipLaunchKernelGGL(pose_rays_bwd_finish_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, params, C, (const float *)workspace, d_params);
    PAG_CHECK_LAUNCH("pag_pose_rays_bwd");
    return PAG_OK;
}

extern "C" int pag_view_embed_bwd(const float *dirs, int64_t R, int n_freq, int width, const float *g_out, float *d_dirs, void *stream) {
    PAG_CHECK_ARG(R >= 0 && n_freq >= 0 && n_freq <= 16 && width >= 3 + 6 * n_freq, "pag_view_embed_bwd: R %lld, n_freq %d, width %d", (long long)R, n_freq, width);
    if (R == 0) return PAG_OK;
    PAG_CHECK_ARG(dirs && g_out && d_dirs, "pag_view_embed_bwd: NULL input/output");
    hipLaunchKernelGGL(view_embed_bwd_kernel, dim3((unsigned)((R * 3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dirs, R, n_freq, width, g_out, d_dirs);
    PAG_CHECK_LAUNCH("pag_view_embed_bwd");
    return PAG_OK;
}
