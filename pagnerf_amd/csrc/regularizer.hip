// segment_consistency_regularizer (loss/regularizers.py:5-35, called at pc_nerf/trainer.py:525-527 in every step of configs/bup20/best.yaml from epoch 601 on)
// on the rendered instance probabilities, where they already live: four launches forward, one backward, no host synchronisation.
//
// The reference loops over images and segments on the host (unique -> .cpu() -> tensor_split, then one bincount / arg-max / nll_loss per segment); the tensor-op
// restatement in pagnerf_amd/loss.py needs ~60 launches forward and as many backward - in the late-training step that is ~0.5 ms of 5 us kernels on the
// stream and ~0.8 ms of launch overhead on the host between the two halves of the backward.  What is computed (quirks included, see oracle/regularizers.py):
//   per image, EVERY distinct value of `labels` is a segment (:11-18); per segment the histogram of its rays' arg-max column (:22); a segment whose rays all
//   predict column 0 is skipped (:24-25); label = first most frequent column among 1.. (:27), 0 when bins[0] * 0.5 > bins[label] (:29-30); term = mean over the
//   segment's rays of -log p[ray, label] (:32); after an image's segments the RUNNING total is divided by that image's segment count (:33); finally / B (:35).
//
//   segreg_slots_kernel   one workgroup per image: the distinct ids through an LDS hash set, sorted by a bitonic network (the order of torch.unique), every
//                         ray's slot = rank of its id by binary search
//   segreg_argmax_kernel  one wave per ray: first arg-max of p + eps over the columns (torch.argmax: NaN counts as the maximum)
//   segreg_terms_kernel   one workgroup per (image, slot): histogram of the slot's rays in LDS (integer atomics), the label, the mean of -log(p[ray, label] + eps) -
//                         lane-strided sums and a fixed butterfly: bitwise reproducible
//   segreg_finish_kernel  one wave: the running normalisation over the images, the value, and coef[b] = d value / d (sum of image b's terms)
//   segreg_bwd_kernel     one wave per ray: the ray's row of d p - zeros and -g coef[b] / (count (p + eps)) in the label's column
#include "common.h"

namespace {

constexpr int SEG_SLOTS = 2048;            // distinct ids per image the set has room for (more: the value is NaN, as pagnerf_amd.loss documents)
constexpr int SEG_SET = 4096;              // hash set slots (load factor <= 1/2)
constexpr long long SEG_EMPTY = (long long)0x8000000000000000ull;      // INT64_MIN marks an empty slot; a ray carrying it counts as "too many ids"

struct SegWs {
    int32_t *slot, *arg;       // [B,P]
    int32_t *nseg, *over;      // [B]
    int32_t *chosen, *cnt;     // [B,SEG_SLOTS]
    float *term;               // [B,SEG_SLOTS]
    float *coef;               // [B]
};

__host__ __device__ inline int64_t seg_align(int64_t x) { return (x + 63) / 64 * 64; }
__host__ inline int64_t seg_ws_bytes(int64_t B, int64_t P) {
    return 2 * seg_align(B * P * 4) + 2 * seg_align(B * 4) + 3 * seg_align(B * SEG_SLOTS * 4) + seg_align(B * 4);
}
__host__ inline SegWs seg_ws(void *base, int64_t B, int64_t P) {
    unsigned char *p = reinterpret_cast<unsigned char *>(base);
    SegWs w;
    w.slot = reinterpret_cast<int32_t *>(p);   p += seg_align(B * P * 4);
    w.arg = reinterpret_cast<int32_t *>(p);    p += seg_align(B * P * 4);
    w.nseg = reinterpret_cast<int32_t *>(p);   p += seg_align(B * 4);
    w.over = reinterpret_cast<int32_t *>(p);   p += seg_align(B * 4);
    w.chosen = reinterpret_cast<int32_t *>(p); p += seg_align(B * SEG_SLOTS * 4);
    w.cnt = reinterpret_cast<int32_t *>(p);    p += seg_align(B * SEG_SLOTS * 4);
    w.term = reinterpret_cast<float *>(p);     p += seg_align(B * SEG_SLOTS * 4);
    w.coef = reinterpret_cast<float *>(p);
    return w;
}

__global__ __launch_bounds__(1024) void segreg_slots_kernel(const int64_t *__restrict__ labels, int64_t P, int32_t *__restrict__ slot, int32_t *__restrict__ nseg,
                                                            int32_t *__restrict__ over_out) {
    __shared__ long long set[SEG_SET];
    __shared__ long long ids[SEG_SLOTS];
    __shared__ int32_t over, count;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    labels += b * P;
    slot += b * P;
    for (int e = tid; e < SEG_SET; e += 1024) set[e] = SEG_EMPTY;
    for (int e = tid; e < SEG_SLOTS; e += 1024) ids[e] = 0x7fffffffffffffffll;
    if (tid == 0) over = 0, count = 0;
    __syncthreads();
    for (int64_t p = tid; p < P; p += 1024) {
        const long long id = labels[p];
        if (id == SEG_EMPTY) { over = 1; continue; }
        unsigned h = (unsigned)(((unsigned long long)id * 0x9E3779B97F4A7C15ull) >> 52);      // 12 bits
        bool done = false;
        for (int probe = 0; probe < SEG_SET && !done; ++probe) {
            const long long prev = (long long)atomicCAS(reinterpret_cast<unsigned long long *>(&set[h]), (unsigned long long)SEG_EMPTY, (unsigned long long)id);
            done = prev == SEG_EMPTY || prev == id;
            h = (h + 1) & (SEG_SET - 1);
        }
        if (!done) over = 1;
    }
    __syncthreads();
    for (int e = tid; e < SEG_SET; e += 1024) {
        const long long v = set[e];
        if (v != SEG_EMPTY) {
            const int at = atomicAdd(&count, 1);
            if (at < SEG_SLOTS) ids[at] = v;
        }
    }
    __syncthreads();
    if (count > SEG_SLOTS) {
        if (tid == 0) over = 1;
    }
    // bitonic network over the 2048 entries (1024 comparators per stage); empty entries hold INT64_MAX and sort to the end
    for (int k = 2; k <= SEG_SLOTS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((tid / j) * 2 * j) + (tid % j), partner = i + j;
            const long long a = ids[i], c = ids[partner];
            const bool up = (i & k) == 0;
            if ((a > c) == up) { ids[i] = c; ids[partner] = a; }
            __syncthreads();
        }
    const int n = count < SEG_SLOTS ? count : SEG_SLOTS;
    for (int64_t p = tid; p < P; p += 1024) {
        const long long id = labels[p];
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ids[mid] < id) lo = mid + 1; else hi = mid;
        }
        slot[p] = lo < n ? lo : (n > 0 ? n - 1 : 0);           // ids the set could not hold (over): any valid slot, the value is NaN anyway
    }
    if (tid == 0) {
        nseg[b] = n;
        over_out[b] = over;
    }
}

__global__ __launch_bounds__(256) void segreg_argmax_kernel(const float *__restrict__ prob, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps,
                                                            int32_t *__restrict__ arg_out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (ray >= P) return;
    const float *row = prob + b * image_stride + ray * row_stride;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    bool any_nan = false;
    for (int c = lane; c < n_cols; c += 64) {
        const float x = __fadd_rn(row[c], eps);
        if (x != x) { if (!any_nan) { any_nan = true; arg = c; } }           // torch.argmax: NaN is the maximum, the first one wins
        else if (!any_nan && (x > best || arg == 0x7fffffff)) { best = x; arg = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oa = __shfl_xor(arg, d);
        const bool on = __shfl_xor((int)any_nan, d) != 0;
        bool take;
        if (any_nan != on) take = on;                       // a NaN beats every number
        else if (any_nan) take = oa < arg;                  // both NaN: the lower column
        else take = ob > best || (ob == best && oa < arg);
        if (take) { best = ob; arg = oa; any_nan = on; }
    }
    if (lane == 0) arg_out[b * P + ray] = arg;
}

__global__ __launch_bounds__(256) void segreg_terms_kernel(const float *__restrict__ prob, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps,
                                                           const int32_t *__restrict__ slot, const int32_t *__restrict__ arg, const int32_t *__restrict__ nseg,
                                                           int32_t *__restrict__ chosen, int32_t *__restrict__ cnt, float *__restrict__ term) {
    extern __shared__ int32_t bins[];                        // [n_cols]
    __shared__ int32_t w_total[4], w_bc[4], w_ba[4];
    __shared__ float w_acc[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.y;
    const int s = blockIdx.x;
    if (s >= nseg[b]) return;                                // workgroup-uniform
    slot += b * P;
    arg += b * P;
    for (int c = tid; c < n_cols; c += 256) bins[c] = 0;
    __syncthreads();
    for (int64_t p = tid; p < P; p += 256)
        if (slot[p] == s) atomicAdd(&bins[arg[p]], 1);      // :22
    __syncthreads();
    // total, and the first most frequent column among 1 .. (:27)
    int total = 0, bc = -1, ba = 0x7fffffff;
    for (int c = tid; c < n_cols; c += 256) {
        const int v = bins[c];
        total += v;
        if (c >= 1 && v > bc) { bc = v; ba = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        total += __shfl_xor(total, d);
        const int oc = __shfl_xor(bc, d), oa = __shfl_xor(ba, d);
        if (oc > bc || (oc == bc && oa < ba)) { bc = oc; ba = oa; }
    }
    if (lane == 0) w_total[wave] = total, w_bc[wave] = bc, w_ba[wave] = ba;
    __syncthreads();
    total = w_total[0] + w_total[1] + w_total[2] + w_total[3];
    bc = w_bc[0], ba = w_ba[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (w_bc[w] > bc || (w_bc[w] == bc && w_ba[w] < ba)) bc = w_bc[w], ba = w_ba[w];
    const int b0 = bins[0];
    int label = -1;
    if (n_cols > 1 && total - b0 > 0) {                      // :24-25: skipped when every ray predicts column 0
        label = ba;
        if (__fmul_rn((float)b0, 0.5f) > (float)bc) label = 0;      // :29-30
    }
    float acc = 0.0f;
    if (label >= 0) {                                        // workgroup-uniform
        const float *col = prob + b * image_stride + label;
        for (int64_t p = tid; p < P; p += 256)
            if (slot[p] == s) acc = __fadd_rn(acc, -logf(__fadd_rn(col[p * row_stride], eps)));      // :32
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, d));
        if (lane == 0) w_acc[wave] = acc;
        __syncthreads();
        acc = __fadd_rn(__fadd_rn(w_acc[0], w_acc[1]), __fadd_rn(w_acc[2], w_acc[3]));
    }
    if (tid == 0) {
        chosen[b * SEG_SLOTS + s] = label;
        cnt[b * SEG_SLOTS + s] = total;
        term[b * SEG_SLOTS + s] = label >= 0 ? __fdiv_rn(acc, (float)total) : 0.0f;
    }
}

__global__ __launch_bounds__(64) void segreg_finish_kernel(int B, const int32_t *__restrict__ nseg, const int32_t *__restrict__ over, const float *__restrict__ term,
                                                           float *__restrict__ coef, float *__restrict__ out) {
    const int lane = threadIdx.x;
    float reg = 0.0f;
    bool bad = false;
    for (int b = 0; b < B; ++b) {
        const int n = nseg[b];
        float acc = 0.0f;
        for (int s = lane; s < n; s += 64) acc = __fadd_rn(acc, term[(int64_t)b * SEG_SLOTS + s]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, d));
        reg = __fdiv_rn(__fadd_rn(reg, acc), (float)n);      // :33 - the running total, earlier images included
        bad = bad || over[b] != 0;
    }
    reg = __fdiv_rn(reg, (float)B);                           // :35
    if (lane == 0) {
        out[0] = bad ? __builtin_nanf("") : reg;
        float c = bad ? 0.0f : __fdiv_rn(1.0f, (float)B);
        for (int b = B - 1; b >= 0; --b) {                    // image b's terms are divided by every LATER image's segment count too
            c = __fdiv_rn(c, (float)nseg[b]);
            coef[b] = c;
        }
    }
}

__global__ __launch_bounds__(256) void segreg_bwd_kernel(const float *__restrict__ prob, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps,
                                                         const int32_t *__restrict__ slot, const int32_t *__restrict__ chosen, const int32_t *__restrict__ cnt,
                                                         const float *__restrict__ coef, const float *__restrict__ grad, float *__restrict__ d_prob) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (ray >= P) return;
    const int s = slot[b * P + ray];
    const int label = chosen[b * SEG_SLOTS + s];
    float v = 0.0f;
    if (label >= 0) {
        const float p = __fadd_rn(prob[b * image_stride + ray * row_stride + label], eps);
        v = -__fdiv_rn(__fmul_rn(grad[0], coef[b]), __fmul_rn((float)cnt[b * SEG_SLOTS + s], p));
    }
    float *row = d_prob + (b * P + ray) * n_cols;
    for (int c = lane; c < n_cols; c += 64) row[c] = c == label ? v : 0.0f;
}

}      // namespace

extern "C" int64_t pag_segment_reg_workspace_bytes(int B, int64_t P) { return (B < 0 || P < 0) ? 0 : seg_ws_bytes(B, P); }

extern "C" int pag_segment_reg_fwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps, const int64_t *labels,
                                   void *workspace, int64_t workspace_bytes, float *out, void *stream) {
    PAG_CHECK_ARG(B >= 1 && B <= 65535 && P >= 1 && n_cols >= 1 && n_cols <= 4096 && row_stride >= n_cols, "pag_segment_reg_fwd: sizes (B %d, P %lld, n_cols %d)", B,
                  (long long)P, n_cols);
    PAG_CHECK_ARG(prob && labels && workspace && out, "pag_segment_reg_fwd: NULL input/output");
    PAG_CHECK_ARG(workspace_bytes >= seg_ws_bytes(B, P), "pag_segment_reg_fwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)seg_ws_bytes(B, P));
    hipStream_t st = (hipStream_t)stream;
    const SegWs w = seg_ws(workspace, B, P);
    hipLaunchKernelGGL(segreg_slots_kernel, dim3(B), dim3(1024), 0, st, labels, P, w.slot, w.nseg, w.over);
    hipLaunchKernelGGL(segreg_argmax_kernel, dim3((unsigned)((P + 3) / 4), B), dim3(256), 0, st, prob, P, image_stride, row_stride, n_cols, eps, w.arg);
    const int64_t slots = P < SEG_SLOTS ? P : SEG_SLOTS;
    hipLaunchKernelGGL(segreg_terms_kernel, dim3((unsigned)slots, B), dim3(256), n_cols * sizeof(int32_t), st, prob, P, image_stride, row_stride, n_cols,
                       eps, w.slot, w.arg, w.nseg, w.chosen, w.cnt, w.term);
    hipLaunchKernelGGL(segreg_finish_kernel, dim3(1), dim3(64), 0, st, B, w.nseg, w.over, w.term, w.coef, out);
    PAG_CHECK_LAUNCH("pag_segment_reg_fwd");
    return PAG_OK;
}

extern "C" int pag_segment_reg_bwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, float eps, const void *workspace,
                                   int64_t workspace_bytes, const float *grad, float *d_prob, void *stream) {
    PAG_CHECK_ARG(B >= 1 && B <= 65535 && P >= 1 && n_cols >= 1 && row_stride >= n_cols, "pag_segment_reg_bwd: sizes");
    PAG_CHECK_ARG(prob && workspace && grad && d_prob, "pag_segment_reg_bwd: NULL input/output");
    PAG_CHECK_ARG(workspace_bytes >= seg_ws_bytes(B, P), "pag_segment_reg_bwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)seg_ws_bytes(B, P));
    const SegWs w = seg_ws(const_cast<void *>(workspace), B, P);
    hipLaunchKernelGGL(segreg_bwd_kernel, dim3((unsigned)((P + 3) / 4), B), dim3(256), 0, (hipStream_t)stream, prob, P, image_stride, row_stride, n_cols, eps, w.slot,
                       w.chosen, w.cnt, w.coef, grad, d_prob);
    PAG_CHECK_LAUNCH("pag_segment_reg_bwd");
    return PAG_OK;
}
