// Adam for the grid tables and the decoders (the caller side of the train step: pc_nerf/trainer.py:583 `scaler.step(optimizer)` with
// config_parser.py:667-673 `torch.optim.Adam(params, eps=1e-15)`).  Why a kernel of our own: the update is a pure stream - per
// element 16 bytes read (p, g, m, v) and 12 written - over two 50.3 MB tables, 704 MB per step whatever the batch; torch's fused
// multi-tensor kernel moves that at 3.6 TB/s on MI355X (196 us per step: 5 % of the full step, 16 % of the post-prune step).  One lane
// handles four float4 groups a full grid-stride apart (four independent 16-byte loads per tensor in flight), plain loads (the gradient
// was written by the reduce pass a moment ago and is still in L2 / MALL), non-temporal stores for the state.
//
// Arithmetic = torch.optim.Adam's single-tensor formula, op for op (torch/optim/adam.py::_single_tensor_adam, maximize / amsgrad off):
//     g     = grad (+ weight_decay * p)
//     m     = m + (g - m) * (1 - beta1)                       exp_avg.lerp_(grad, 1 - beta1)
//     v     = v * beta2 + (1 - beta2) * g * g                  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//     denom = sqrt(v) / sqrt(1 - beta2^t) + eps
//     p     = p - (lr / (1 - beta1^t)) * (m / denom)           param.addcdiv_(exp_avg, denom, value=-step_size)
// with the two bias corrections formed on the host in double precision and handed over as floats.
#include <algorithm>
#include <math.h>
#include "common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AdamTensor {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
};
constexpr int ADAM_MAX_TENSORS = 48;       // tensors of one launch (the two tables + the decoders: 24)
struct AdamScalars {
    float step_size, beta1, beta2, one_minus_beta1, one_minus_beta2, bc2_sqrt, eps, weight_decay;
};

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, const AdamScalars &s) {
    if (s.weight_decay != 0.0f) g = fmaf(p, s.weight_decay, g);                     // grad.add(param, alpha=weight_decay)
    m = fmaf(g - m, s.one_minus_beta1, m);                                           // lerp (weight < 0.5 form: a + w (b - a))
    v = fmaf(s.one_minus_beta2 * g, g, v * s.beta2);                                 // addcmul: v*beta2 + (1 - beta2) g g
    const float denom = __fsqrt_rn(v) / s.bc2_sqrt + s.eps;
    p = p - s.step_size * (m / denom);
}

// ONE launch for every tensor of the call: block b works on the tensor whose block range [first[t], first[t + 1]) holds it (a scan over at
// most 48 scalars); the range is sized by the host - a streaming tensor gets thousands of blocks, a decoder matrix one or two.  Within a
// tensor: float4 groups, four per lane a stride apart (four independent 16-byte loads per array in flight), the n % 4 tail by block 0;
// tensors that are not 16-byte aligned take the element loop.
struct AdamAll {
    AdamTensor t[ADAM_MAX_TENSORS];
    int first[ADAM_MAX_TENSORS + 1];
    unsigned char vec[ADAM_MAX_TENSORS];
    int count;
};
__global__ __launch_bounds__(256) void adam_kernel(AdamAll b, AdamScalars s) {
    int ti = 0;
    for (int k = 1; k < b.count; ++k) ti = (int)blockIdx.x >= b.first[k] ? k : ti;
    const AdamTensor t = b.t[ti];
    const int64_t nblk = b.first[ti + 1] - b.first[ti], blk = (int)blockIdx.x - b.first[ti];
    const int64_t stride = nblk * 256;
    const int64_t i0 = blk * 256 + threadIdx.x;
    if (!b.vec[ti]) {
        for (int64_t i = i0; i < t.n; i += stride) {
            float p = t.p[i], m = t.m[i], v = t.v[i];
            adam_elem(p, t.g[i], m, v, s);
            t.p[i] = p;
            t.m[i] = m;
            t.v[i] = v;
        }
        return;
    }
    const int64_t n4 = t.n >> 2;
    f32x4 *p4 = reinterpret_cast<f32x4 *>(t.p), *m4 = reinterpret_cast<f32x4 *>(t.m), *v4 = reinterpret_cast<f32x4 *>(t.v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(t.g);
    for (int64_t base = i0; base < n4; base += 4 * stride) {
        f32x4 pp[4], gg[4], mm[4], vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * stride;
            if (i < n4) {
                pp[k] = p4[i];
                gg[k] = g4[i];
                mm[k] = m4[i];
                vv[k] = v4[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = base + k * stride;
            if (i < n4) {
                // a group whose gradient and both moments are exactly zero (table rows no sample has ever touched: most rows of the coarse
                // levels) does not change: g = 0 -> m = 0, v = 0, p - step * (0 / (0 + eps)) = p.  Its three stores are skipped (12 of the
                // 28 bytes per element); with weight decay or eps = 0 the update is not the identity and nothing is skipped.
                bool idle = s.weight_decay == 0.0f && s.eps > 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) idle = idle && gg[k][e] == 0.0f && mm[k][e] == 0.0f && vv[k][e] == 0.0f;
                if (idle) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pp[k][e], me = mm[k][e], ve = vv[k][e];
                    adam_elem(pe, gg[k][e], me, ve, s);
                    pp[k][e] = pe;
                    mm[k][e] = me;
                    vv[k][e] = ve;
                }
                __builtin_nontemporal_store(pp[k], p4 + i);
                __builtin_nontemporal_store(mm[k], m4 + i);
                __builtin_nontemporal_store(vv[k], v4 + i);
            }
        }
    }
    const int64_t tail = n4 << 2;
    if (blk == 0 && threadIdx.x < (t.n - tail)) {
        const int64_t i = tail + threadIdx.x;
        float p = t.p[i], m = t.m[i], v = t.v[i];
        adam_elem(p, t.g[i], m, v, s);
        t.p[i] = p;
        t.m[i] = m;
        t.v[i] = v;
    }
}
}  // namespace

extern "C" int pag_adam_step(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                             const int64_t *numel, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                             void *stream) {
    PAG_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel)), "pag_adam_step: NULL tensor list");
    PAG_CHECK_ARG(step >= 1, "pag_adam_step: step %lld must be >= 1 (the step count AFTER this update, as torch counts it)", (long long)step);
    PAG_CHECK_ARG(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0,
                  "pag_adam_step: lr %g, betas (%g, %g), eps %g, weight_decay %g out of range", lr, beta1, beta2, eps, weight_decay);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars s{(float)(lr / bc1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)weight_decay};
    if (n_tensors == 0) return PAG_OK;
    hipStream_t st = (hipStream_t)stream;
    AdamAll all{};
    all.count = 0;
    all.first[0] = 0;
    bool launched = false;
    auto flush = [&]() {
        if (all.count == 0) return;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)all.first[all.count]), dim3(256), 0, st, all, s);
        launched = true;
        all.count = 0;
        all.first[0] = 0;
    };
    for (int i = 0; i < n_tensors; ++i) {
        if (numel[i] == 0) continue;
        PAG_CHECK_ARG(numel[i] > 0 && params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], "pag_adam_step: tensor %d: NULL pointer or negative size", i);
        AdamTensor t{params[i], grads[i], exp_avg[i], exp_avg_sq[i], numel[i]};
        const bool aligned = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
        // blocks of this tensor: one per 4096 elements (a lane's four float4 groups), at most 4096 - enough to fill the chip several times
        // over, few enough that every lane of a streaming tensor walks several batches
        const int64_t want = aligned ? (numel[i] + 4095) / 4096 : (numel[i] + 1023) / 1024;
        const int nb = (int)std::min<int64_t>(4096, std::max<int64_t>(1, want));
        all.t[all.count] = t;
        all.vec[all.count] = aligned ? 1 : 0;
        all.first[all.count + 1] = all.first[all.count] + nb;
        if (++all.count == ADAM_MAX_TENSORS) flush();
    }
    flush();
    if (!launched) return PAG_OK;
    PAG_CHECK_LAUNCH("pag_adam_step");
    return PAG_OK;
}
