// Touched-rows exchange of a table gradient (pagnerf_amd/shard.py::SparseRows): the four passes around the collective.
//
// After the first prune (configs/bup20/best.yaml:187, pc_nerf/trainer.py:362-366) a step's samples touch a small part of the coarse and middle
// lattice levels' rows; the reduce pass of the table gradient writes every row (pag_*_encode_bwd_set), so an untouched row is an exact zero on
// every rank.  Only the UNION over the ranks of the non-zero rows has to cross the links:
//   mask    grad f32 [L][T][F] -> bits u32 [L][W], W = ceil(T / 32): bit r of word w of level l = any(grad[l][32 w + r][:] != 0)
//           (all_gather + OR of the ranks' bits happens between mask and plan, on the host side: 786 KB for 24 x 2^18 rows)
//   plan    union bits -> word_prefix i32 [L][W] = union rows of the level before word w, counts i64 [L + 1] = union rows per level and, in [L], the
//           rows that do not fit their level's slots (caps i32 [L]; caps[l] >= T = the level travels whole: prefix 32 w, every row a member)
//   pack    member rows in row order -> buf f32 [slots][F] at offs[l] + rank (rank < caps[l]); the caller zero-fills buf first (unused slots)
//   unpack  every row of grad rewritten: member and rank < caps[l] ? buf[offs[l] + rank] : 0
// ~170 MB of traffic for a 50 MB table instead of the ~25 tensor-op launches over 6.3 M-element index tensors of the torch form (1.3 ms, measured).
#include "common.h"

namespace {

// one lane per row, one wave per two words
__global__ __launch_bounds__(256) void sparse_mask_kernel(const float *__restrict__ grad, int64_t T, int F, int64_t W, uint32_t *__restrict__ bits) {
    const int l = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool any = false;
    if (row < T) {
        const float *g = grad + ((int64_t)l * T + row) * F;
        if (F == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(g);
            any = v.x != 0.0f || v.y != 0.0f;
        } else {
            for (int f = 0; f < F; ++f) any |= g[f] != 0.0f;
        }
    }
    const unsigned long long m = __ballot(any);
    const int lane = threadIdx.x & 63;
    const int64_t w0 = row >> 5;                      // word of this lane's row; lanes 0 and 32 of a wave write
    if ((lane & 31) == 0 && w0 < W) bits[(int64_t)l * W + w0] = (uint32_t)(m >> (lane & 32));
}

// one workgroup per level: exclusive prefix of the words' popcounts
__global__ __launch_bounds__(1024) void sparse_plan_kernel(const uint32_t *__restrict__ bits, int64_t T, int64_t W, const int32_t *__restrict__ caps,
                                                           int32_t *__restrict__ word_prefix, unsigned long long *__restrict__ counts, int L) {
    __shared__ int32_t wave_sum[16];
    __shared__ int32_t carry_s;
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool whole = caps[l] >= T;
    bits += (int64_t)l * W;
    word_prefix += (int64_t)l * W;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < W; base += 1024) {
        const int64_t w = base + tid;
        uint32_t word = w < W ? bits[w] : 0u;
        if (w == W - 1 && (T & 31)) word &= (1u << (T & 31)) - 1u;          // bits past the last row do not count
        const int c = __popc(word);
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int k = 0; k < wave; ++k) before += wave_sum[k];
        if (w < W) word_prefix[w] = whole ? (int32_t)(w * 32) : before + incl - c;
        __syncthreads();
        if (tid == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        const long long n = carry_s;
        counts[l] = (unsigned long long)n;
        const long long over = whole ? 0 : n - (long long)caps[l];
        if (over > 0) atomicAdd(&counts[L], (unsigned long long)over);
    }
}

template <bool PACK>
__global__ __launch_bounds__(256) void sparse_move_kernel(float *__restrict__ grad, int64_t T, int F, int64_t W, const uint32_t *__restrict__ bits,
                                                          const int32_t *__restrict__ word_prefix, const int32_t *__restrict__ caps,
                                                          const int64_t *__restrict__ offs, float *__restrict__ buf) {
    const int l = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= T) return;
    const int cap = caps[l];
    const bool whole = cap >= T;
    const int64_t w = row >> 5;
    const int r = (int)(row & 31);
    const uint32_t word = bits[(int64_t)l * W + w];
    const bool member = whole || ((word >> r) & 1u);
    const int64_t rank = whole ? row : (int64_t)word_prefix[(int64_t)l * W + w] + __popc(word & ((1u << r) - 1u));
    const bool in = member && rank < cap;
    float *g = grad + ((int64_t)l * T + row) * F;
    float *b = buf + (offs[l] + rank) * F;
    if (PACK) {
        if (!in) return;
        if (F == 2) *reinterpret_cast<float2 *>(b) = *reinterpret_cast<const float2 *>(g);
        else
            for (int f = 0; f < F; ++f) b[f] = g[f];
    } else {
        if (F == 2) *reinterpret_cast<float2 *>(g) = in ? *reinterpret_cast<const float2 *>(b) : float2{0.0f, 0.0f};
        else
            for (int f = 0; f < F; ++f) g[f] = in ? b[f] : 0.0f;
    }
}

int sparse_check(const char *name, const void *grad, int L, int64_t T, int F, const void *bits) {
    PAG_CHECK_ARG(L >= 1 && L <= 65535 && T >= 1 && T <= ((int64_t)1 << 31) && F >= 1 && F <= PAG_MAX_FEATS, "%s: L %d not in [1,65535], T %lld not in [1,2^31] or F %d not in [1,%d]", name, L,
                  (long long)T, F, PAG_MAX_FEATS);
    PAG_CHECK_ARG(grad && bits, "%s: NULL grad / bits", name);
    return PAG_OK;
}

}  // namespace

extern "C" int pag_sparse_rows_mask(const float *grad, int L, int64_t T, int F, uint32_t *bits, void *stream) {
    int rc = sparse_check("pag_sparse_rows_mask", grad, L, T, F, bits);
    if (rc != PAG_OK) return rc;
    const int64_t W = (T + 31) / 32;
    hipLaunchKernelGGL(sparse_mask_kernel, dim3((unsigned)((W * 32 + 255) / 256), L), dim3(256), 0, (hipStream_t)stream, grad, T, F, W, bits);
    PAG_CHECK_LAUNCH("pag_sparse_rows_mask");
    return PAG_OK;
}

extern "C" int pag_sparse_rows_plan(const uint32_t *bits, int L, int64_t T, const int32_t *caps, int32_t *word_prefix, int64_t *counts, void *stream) {
    PAG_CHECK_ARG(L >= 1 && L <= 65535 && T >= 1 && T <= ((int64_t)1 << 31), "pag_sparse_rows_plan: L %d not in [1,65535] or T %lld not in [1,2^31]", L, (long long)T);
    PAG_CHECK_ARG(bits && caps && word_prefix && counts, "pag_sparse_rows_plan: NULL input/output");
    const int64_t W = (T + 31) / 32;
    hipError_t e = hipMemsetAsync(counts + L, 0, sizeof(int64_t), (hipStream_t)stream);
    PAG_CHECK_ARG(e == hipSuccess, "pag_sparse_rows_plan: hipMemsetAsync failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sparse_plan_kernel, dim3(L), dim3(1024), 0, (hipStream_t)stream, bits, T, W, caps, word_prefix, reinterpret_cast<unsigned long long *>(counts), L);
    PAG_CHECK_LAUNCH("pag_sparse_rows_plan");
    return PAG_OK;
}

extern "C" int pag_sparse_rows_pack(const float *grad, int L, int64_t T, int F, const uint32_t *bits, const int32_t *word_prefix, const int32_t *caps, const int64_t *offs,
                                    float *buf, void *stream) {
    int rc = sparse_check("pag_sparse_rows_pack", grad, L, T, F, bits);
    if (rc != PAG_OK) return rc;
    PAG_CHECK_ARG(word_prefix && caps && offs && buf, "pag_sparse_rows_pack: NULL word_prefix / caps / offs / buf");
    const int64_t W = (T + 31) / 32;
    hipLaunchKernelGGL(sparse_move_kernel<true>, dim3((unsigned)((T + 255) / 256), L), dim3(256), 0, (hipStream_t)stream, const_cast<float *>(grad), T, F, W, bits, word_prefix, caps, offs,
                       buf);
    PAG_CHECK_LAUNCH("pag_sparse_rows_pack");
    return PAG_OK;
}

extern "C" int pag_sparse_rows_unpack(const float *buf, int L, int64_t T, int F, const uint32_t *bits, const int32_t *word_prefix, const int32_t *caps, const int64_t *offs,
                                      float *grad, void *stream) {
    int rc = sparse_check("pag_sparse_rows_unpack", grad, L, T, F, bits);
    if (rc != PAG_OK) return rc;
    PAG_CHECK_ARG(word_prefix && caps && offs && buf, "pag_sparse_rows_unpack: NULL word_prefix / caps / offs / buf");
    const int64_t W = (T + 31) / 32;
    hipLaunchKernelGGL(sparse_move_kernel<false>, dim3((unsigned)((T + 255) / 256), L), dim3(256), 0, (hipStream_t)stream, grad, T, F, W, bits, word_prefix, caps, offs,
                       const_cast<float *>(buf));
    PAG_CHECK_LAUNCH("pag_sparse_rows_unpack");
    return PAG_OK;
}
