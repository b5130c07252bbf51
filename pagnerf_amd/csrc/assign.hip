// Per-label sums of per-ray values: the cost-matrix half of the linear-assignment instance loss
// (loss/lin_assignment_things.py:31-33, loss/lin_assignment.py:19-21) and the per-id centre sums of
// utils/outlier_rejection.py:56-71, computed where the rendered `inst_embedding` already lives so that only
// the [K, I] matrix (K gt labels in the image) crosses PCIe for SciPy's Hungarian solver instead of
// the [4096, 200] probabilities.
//
// One workgroup per label.  The rows of that label are compacted IN RAY ORDER into LDS 256 at a time
// (ballot + prefix), then every lane adds its column of those rows sequentially: the fp32 sum has the
// same order as a sequential reduction over dim 0, independent of launch geometry (deterministic).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void label_sums_kernel(const T *__restrict__ values, int64_t P, int64_t row_stride, int col0, int C,
                                                         const int64_t *__restrict__ labels_gt, const uint8_t *__restrict__ row_mask,
                                                         const int64_t *__restrict__ label_list, float *__restrict__ sums,
                                                         int32_t *__restrict__ counts) {
    __shared__ int32_t rows[256];
    __shared__ int32_t wave_n[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t mine = label_list[k];
    const int ncol = (C + 255) / 256;          // columns per lane (C <= 1024)
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int total = 0;
    for (int64_t base = 0; base < P; base += 256) {
        const int64_t p = base + tid;
        const bool hit = p < P && labels_gt[p] == mine && (row_mask == nullptr || row_mask[p] != 0);
        const uint64_t m = __ballot(hit);
        if (lane == 0) wave_n[wave] = __popcll(m);
        __syncthreads();
        int off = 0, n = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            off += w < wave ? wave_n[w] : 0;
            n += wave_n[w];
        }
        if (hit) rows[off + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)(p - base);
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const T *row = values + (base + rows[j]) * row_stride + col0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = tid + 256 * q;
                if (q < ncol && c < C) acc[q] += pag_ld(row + c);
            }
        }
        total += n;
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = tid + 256 * q;
        if (q < ncol && c < C) sums[(int64_t)k * C + c] = acc[q];
    }
    if (tid == 0) counts[k] = total;
}

}  // namespace

extern "C" int pag_label_sums(const void *values, int value_dtype, int64_t P, int64_t row_stride, int col0, int C,
                              const int64_t *labels_gt, const uint8_t *row_mask, const int64_t *label_list, int K, float *sums,
                              int32_t *counts, void *stream) {
    PAG_CHECK_ARG(P >= 0 && K >= 0, "pag_label_sums: negative size");
    PAG_CHECK_ARG(C >= 1 && C <= 1024 && col0 >= 0, "pag_label_sums: C %d not in [1,1024] or col0 < 0", C);
    PAG_CHECK_ARG(value_dtype == PAG_F32 || value_dtype == PAG_BF16, "pag_label_sums: values must be F32 or BF16");
    PAG_CHECK_ARG(row_stride >= col0 + C, "pag_label_sums: row_stride smaller than col0 + C");
    if (K == 0) return PAG_OK;
    PAG_CHECK_ARG(label_list && sums && counts && (P == 0 || (values && labels_gt)), "pag_label_sums: NULL input/output");
    hipStream_t st = (hipStream_t)stream;
    if (value_dtype == PAG_F32)
        hipLaunchKernelGGL(label_sums_kernel<float>, dim3(K), dim3(256), 0, st, (const float *)values, P, row_stride, col0, C, labels_gt,
                           row_mask, label_list, sums, counts);
    else
        hipLaunchKernelGGL(label_sums_kernel<bf16_t>, dim3(K), dim3(256), 0, st, (const bf16_t *)values, P, row_stride, col0, C, labels_gt,
                           row_mask, label_list, sums, counts);
    PAG_CHECK_LAUNCH("pag_label_sums");
    return PAG_OK;
}
