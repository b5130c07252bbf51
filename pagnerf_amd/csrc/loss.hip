// Per-ray training loss of the rendered buffers in two launches (forward: value, backward: gradients) instead of the ~30
// element-wise / reduction launches the same arithmetic costs as separate tensor ops on 4096-ray batches:
//   rgb term   : weight * mean |rgb - gt|                                  (pc_nerf/trainer.py:443-446)
//   NLL terms  : weight * mean_n( -log(p[n, target_n] + eps) * inv_temperature * conf_n )
//                (semantics :459-465 - reduction 'none' then mean over ALL rays; the instance term after the linear
//                 assignment, loss/lin_assignment_things.py:80 - F.nll_loss mean over the VALID rays)
// Rows whose target is outside [0, C) (F.nll_loss's ignore_index = -100 included) contribute nothing.
//
// Determinism: every block reduces its rows in a fixed order and writes one partial; the last block to finish (ticket
// counter, reset for the next call) adds the partials in block order.  No float atomics.
#include "common.h"

namespace {

constexpr int RL_MAX_BLOCKS = 128;

struct NllTerm {
    const float *prob;        // [N, C] or NULL (term absent)
    const int64_t *target;    // [N]
    const float *conf;        // [N] or NULL
    int C;
    float weight, inv_temperature;
    int mean_over_all;        // 1: divide by N, 0: divide by the number of valid rows
};

struct LossArgs {
    const float *rgb, *rgb_gt;   // [N,3] or NULL
    int64_t N;
    float rgb_weight, eps;
    NllTerm t[2];
};

__device__ __forceinline__ float block_sum(float v, float *scratch /*[4]*/) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

__global__ __launch_bounds__(256) void render_loss_fwd_kernel(LossArgs a, float *__restrict__ partials, int *__restrict__ ticket,
                                                              float *__restrict__ out) {
    __shared__ float scratch[4];
    __shared__ int last;
    float s[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};          // |rgb| sum, nll A, count A, nll B, count B
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < a.N; n += (int64_t)gridDim.x * 256) {
        if (a.rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s[0] += fabsf(a.rgb[n * 3 + c] - a.rgb_gt[n * 3 + c]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const NllTerm &t = a.t[k];
            if (!t.prob) continue;
            const int64_t lab = t.target[n];
            if (lab < 0 || lab >= t.C) continue;
            float l = -logf(t.prob[n * t.C + lab] + a.eps) * t.inv_temperature;
            if (t.conf) l *= t.conf[n];
            s[1 + 2 * k] += l;
            s[2 + 2 * k] += 1.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float v = block_sum(s[q], scratch);
        if (threadIdx.x == 0) partials[blockIdx.x * 5 + q] = v;
    }
    __threadfence();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;                          // workgroup-uniform
    __threadfence();
    // the last workgroup adds the partial sums: thread b holds workgroup b's (gridDim.x <= RL_MAX_BLOCKS <= 256), then the fixed tree of block_sum - one
    // thread walking them (5 x gridDim.x dependent L2 round trips) was 20 of the launch's 25 us
    float tot[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float v = threadIdx.x < gridDim.x ? __hip_atomic_load(partials + threadIdx.x * 5 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        tot[q] = block_sum(v, scratch);
    }
    if (threadIdx.x != 0) return;
    const float n_f = (float)a.N;
    const float rgb_term = a.rgb ? a.rgb_weight * (tot[0] / (3.0f * n_f)) : 0.0f;
    float term[2], den[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        den[k] = a.t[k].mean_over_all ? n_f : tot[2 + 2 * k];
        term[k] = a.t[k].prob ? a.t[k].weight * (tot[1 + 2 * k] / den[k]) : 0.0f;
    }
    out[0] = (rgb_term + term[0]) + term[1];
    out[1] = rgb_term;
    out[2] = term[0];
    out[3] = term[1];
    out[4] = den[0];
    out[5] = den[1];
    *ticket = 0;
}

__global__ __launch_bounds__(256) void render_loss_bwd_kernel(LossArgs a, const float *__restrict__ g_ptr, const float *__restrict__ fwd_out,
                                                              float *__restrict__ d_rgb, float *__restrict__ d_a, float *__restrict__ d_b) {
    const float g = g_ptr ? *g_ptr : 1.0f;
    const int64_t n_rgb = d_rgb ? a.N * 3 : 0;
    const int64_t n_a = d_a ? a.N * a.t[0].C : 0;
    const int64_t n_b = d_b ? a.N * a.t[1].C : 0;
    const float rgb_scale = g * a.rgb_weight / (3.0f * (float)a.N);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rgb + n_a + n_b; i += (int64_t)gridDim.x * 256) {
        if (i < n_rgb) {
            const float d = a.rgb[i] - a.rgb_gt[i];
            d_rgb[i] = d > 0.0f ? rgb_scale : (d < 0.0f ? -rgb_scale : (d == 0.0f ? 0.0f : d));     // sgn(0) = 0, NaN stays NaN
            continue;
        }
        const int k = i < n_rgb + n_a ? 0 : 1;
        const int64_t j = i - n_rgb - (k ? n_a : 0);
        const NllTerm &t = a.t[k];
        const int64_t n = j / t.C;
        const int c = (int)(j - n * t.C);
        float v = 0.0f;
        if (t.target[n] == c) {
            v = -g * t.weight * t.inv_temperature / (fwd_out[4 + k] * (t.prob[j] + a.eps));
            if (t.conf) v *= t.conf[n];
        }
        (k ? d_b : d_a)[j] = v;
    }
}

int fill_args(LossArgs &a, const float *rgb, const float *rgb_gt, int64_t N, float rgb_weight, const float *prob_a, int C_a,
              const int64_t *target_a, const float *conf_a, float weight_a, float inv_temp_a, int all_a, const float *prob_b, int C_b,
              const int64_t *target_b, const float *conf_b, float weight_b, float inv_temp_b, int all_b, float eps, const char *who) {
    PAG_CHECK_ARG(N >= 0, "%s: N < 0", who);
    PAG_CHECK_ARG((rgb == nullptr) == (rgb_gt == nullptr), "%s: rgb and rgb_gt go together", who);
    PAG_CHECK_ARG(!prob_a || (C_a >= 1 && target_a), "%s: term A needs C >= 1 and targets", who);
    PAG_CHECK_ARG(!prob_b || (C_b >= 1 && target_b), "%s: term B needs C >= 1 and targets", who);
    a = LossArgs{rgb, rgb_gt, N, rgb_weight, eps,
                 {NllTerm{prob_a, target_a, conf_a, C_a, weight_a, inv_temp_a, all_a}, NllTerm{prob_b, target_b, conf_b, C_b, weight_b, inv_temp_b, all_b}}};
    return PAG_OK;
}

}  // namespace

extern "C" int64_t pag_render_loss_workspace_bytes(void) { return (int64_t)(RL_MAX_BLOCKS * 5 * sizeof(float) + 64); }

extern "C" int pag_render_loss_fwd(const float *rgb, const float *rgb_gt, int64_t N, float rgb_weight, const float *prob_a, int C_a,
                                   const int64_t *target_a, const float *conf_a, float weight_a, float inv_temp_a, int all_a,
                                   const float *prob_b, int C_b, const int64_t *target_b, const float *conf_b, float weight_b,
                                   float inv_temp_b, int all_b, float eps, void *workspace, float *out, void *stream) {
    LossArgs a;
    int rc = fill_args(a, rgb, rgb_gt, N, rgb_weight, prob_a, C_a, target_a, conf_a, weight_a, inv_temp_a, all_a, prob_b, C_b, target_b,
                       conf_b, weight_b, inv_temp_b, all_b, eps, "pag_render_loss_fwd");
    if (rc != PAG_OK) return rc;
    PAG_CHECK_ARG(workspace && out, "pag_render_loss_fwd: NULL workspace/out");
    int64_t blocks = (N + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > RL_MAX_BLOCKS ? RL_MAX_BLOCKS : blocks);
    int *ticket = reinterpret_cast<int *>(workspace);                    // zero on first use, reset by the kernel
    float *partials = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + 64);
    hipLaunchKernelGGL(render_loss_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, partials, ticket, out);
    PAG_CHECK_LAUNCH("pag_render_loss_fwd");
    return PAG_OK;
}

extern "C" int pag_render_loss_bwd(const float *g, const float *fwd_out, const float *rgb, const float *rgb_gt, int64_t N, float rgb_weight,
                                   const float *prob_a, int C_a, const int64_t *target_a, const float *conf_a, float weight_a,
                                   float inv_temp_a, int all_a, const float *prob_b, int C_b, const int64_t *target_b, const float *conf_b,
                                   float weight_b, float inv_temp_b, int all_b, float eps, float *d_rgb, float *d_a, float *d_b,
                                   void *stream) {
    LossArgs a;
    int rc = fill_args(a, rgb, rgb_gt, N, rgb_weight, prob_a, C_a, target_a, conf_a, weight_a, inv_temp_a, all_a, prob_b, C_b, target_b,
                       conf_b, weight_b, inv_temp_b, all_b, eps, "pag_render_loss_bwd");
    if (rc != PAG_OK) return rc;
    PAG_CHECK_ARG(fwd_out, "pag_render_loss_bwd: NULL fwd_out");
    PAG_CHECK_ARG((!d_rgb || rgb) && (!d_a || prob_a) && (!d_b || prob_b), "pag_render_loss_bwd: gradient requested for an absent term");
    const int64_t items = (d_rgb ? N * 3 : 0) + (d_a ? N * C_a : 0) + (d_b ? N * C_b : 0);
    if (items == 0) return PAG_OK;
    int64_t blocks = (items + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(render_loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, g, fwd_out, d_rgb, d_a, d_b);
    PAG_CHECK_LAUNCH("pag_render_loss_bwd");
    return PAG_OK;
}
