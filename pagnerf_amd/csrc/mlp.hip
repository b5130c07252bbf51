// Fused tiny-MLP decoders for gfx950 (wisp BasicDecoder: Linear+ReLU stacks, width 64).
//
// MFMA path (PAG_MLP_MFMA_BF16) - the whole decoder chain of a 32-sample tile stays in one wave's
// registers:
//   * everything is computed TRANSPOSED, H^T[neurons x samples] = W[out x in] . X^T[in x samples],
//     with v_mfma_f32_32x32x16_bf16: A = the nn.Linear weight exactly as torch stores it ([out][in],
//     read from LDS as one ds_read_b128 per fragment), B = activations with the SAMPLE on the lane
//     (col = lane & 31) - so a layer's 32x32 fp32 accumulator block, after bias/ReLU and a pairwise
//     v_cvt_pk_bf16_f32, IS the next layer's B operand: no LDS round trip, no shuffles.
//   * accumulator register q of lane (r, h = lane >> 5) holds row rho(q,h) = (q&3) + 8(q>>2) + 4h of
//     its block.  Used as a B fragment, element j of k-step s is therefore input 16s + 8(j>>2) + 4h +
//     (j&3), not 16s + 8h + j: the weights of every layer fed from registers are staged into LDS
//     with bits 2 and 3 of the input index swapped, which makes the hardware's k order match.
//   * biases are the accumulators' initial value; softmax / sigmoid run on the accumulators
//     (softmax over the registers of a lane + one cross-half shuffle).
//   * one 256-thread workgroup = 4 waves x 32-sample tiles, grid-strided; weights are converted to
//     bf16 and staged once per workgroup (<= 51 KiB LDS, rows padded by 16 B against bank conflicts).
// The backward kernel runs the same structure on the transposed weights and emits the per-layer
// pre-activation gradients dz (bf16) and dx; weight gradients are dz^T . input, a plain GEMM that is
// left to the BLAS library.
//
// FP32 path (PAG_MLP_FP32): one lane per sample, fp32 FMA chains in k order with the weights
// broadcast from LDS - the parity path (tolerance 1e-5 against the fp32 oracle).
#include "common.h"
#include "blocktime.h"
#include <algorithm>
#include <type_traits>

namespace {

typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PAG_WIDE_BWD_WAVES
#define PAG_WIDE_BWD_WAVES 8      // waves per workgroup of mlp_bwd_wide_mfma (8 or 16; one workgroup per CU)
#endif
constexpr int RS = 72;          // LDS row stride (bf16 elements) of a 64-wide weight row: 144 B
constexpr int HID = 64;

__device__ __forceinline__ int rho(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }
__device__ __forceinline__ int swap23(int a) { return (a & ~12) | ((a & 4) << 1) | ((a & 8) >> 1); }

// staged position (8*g + e) of the XCD8 layout -> column level*F + f of the [M, L*F] feature row, or -1 (padding)
// (a / d for the wave-uniform divisors of the staging loops: a shift when d is a power of two - the feature width and the pad sizes always
// are - instead of the ~40-instruction software division; the loops below were bound by those: 25 k of a launch's clocks per workgroup)
__device__ __forceinline__ int udiv_uniform(int a, int d) { return (d & (d - 1)) == 0 ? a >> (31 - __clz(d)) : a / d; }
__device__ __forceinline__ int grp_col(int pos, int L, int F) {
    const int g = pos >> 3, e = pos & 7;
    const int j = udiv_uniform(e, F), f = e - j * F;
    const int level = xcd8_level(g, j);
    return (j < (L + 7) / 8 && level < L) ? level * F + f : -1;
}

struct FwdParams {
    const void *x1;
    const float *x2;
    const int32_t *x2_index;
    int k1, k2p, in_dim, in_pad, out_dim, act;
    const float *W[3];
    const float *b[3];
    void *out;
    void *hsave[2];
    int64_t M;
    int grp_L, grp_F;      // x1 in PAG_LAYOUT_XCD8 (bf16 [8][M][8]) when grp_L > 0
    float *stats;          // optional f32 [M,2]: (max logit * log2e, 1 / sum exp) of the output softmax
    float *col0_relu;      // optional f32 [M]: relu(x1[m][0]) (strided bf16 x1) - the density read off the colour decoder's input
    // companion head of mlp_fwd_wide_stats<.., PAIR = true>: a two-layer softmax decoder (out2_dim <= 8, bf16 out2 [M, out2_dim]) on the same x1
    const float *W2[2];
    const float *b2[2];
    void *out2;
    int out2_dim;
};

struct BwdParams {
    const void *grad_out;      // same dtype as `out` (autograd hands back the output's dtype); unused in rank-1 mode
    // rank-1 upstream gradient (composited panoptic heads): g[m][c] = g_scale[m] * g_ray[g_index[m]][c]
    const float *g_ray;
    const float *g_scale;
    const float *g_ray_scale;  // optional per-ray factor: g[m][c] = g_scale[m] * g_ray_scale[g_index[m]] * g_ray[g_index[m]][c]
    const int32_t *g_index;
    const void *out;
    int k1, in_dim, in_pad, out_dim, act;
    const float *W[3];
    const void *hsave[2];
    void *dz[3];
    void *dx1;
    int64_t M;
    int grp_L, grp_F;      // dx1 (and layer-0 weight columns) in PAG_LAYOUT_XCD8 order when grp_L > 0
    const float *stats;    // forward's softmax statistics + last-layer bias: the wide kernel recomputes the probabilities
    const float *b_last;
    int dx1_acc;           // XCD8 dx1: add to the existing contents instead of overwriting
    const float *dx_col0;  // strided dx1: f32 [M] added to column 0
    const float *dx_col0_gate;   // optional f32 [M]: the addend counts only where gate[m] > 0 (relu of that column)
    // fused weight gradients (mlp_bwd_mfma<.., FUSE = true>): the forward's layer-0 input and one slab set per layer
    const void *x1;            // bf16: [8][M][8] (grp_L > 0) or [M, k1]
    const float *x2;           // optional f32 [R, k2p] gathered through x2_index (view embedding)
    const int32_t *x2_index;
    int k2p;
    float *slabs[3];           // per layer: f32 [4 * gridDim.x][rows_pad][WG_SLAB_COLS]  (cols 0..63 dW, col 64 db)
    const float *b[3];         // biases of the hidden layers: mlp_bwd_fused recomputes the hidden activations instead of reading saved ones
    float *dz0_slots;          // colour-like fused kernel, DZ0 == 2: f32 [(ceil(M / 32) + R)][64] column sums of dz_0 per (tile, ray), row = tile + ray
};

// Stage W [n_out x n_in] f32 row-major into LDS as bf16 [rows_pad][stride]; zero padding;
// optional bit-2/3 swap of the column index (see header).
// Every decoder kernel starts with 3 - 8 of these matrices.  As a loop of one element per iteration (a predicated load, then its store) the
// prologue ran at one L2 round trip per element - 80 dependent round trips per thread, 23 - 27 k clocks = 10 us per launch whatever the batch
// (-DPAG_FUSED_PROF).  Here eight loads are issued back to back - unconditionally, from clamped addresses, so that no branch separates them -
// and the padding is applied to the values afterwards.
constexpr int STAGE_BATCH = 8;      // 16: slower (registers / code size)
__device__ void stage_weight(bf16_t *dst, int stride, int rows_pad, int cols_pad, const float *W, int n_out, int n_in,
                             bool permute, int grp_L = 0, int grp_F = 0) {
    const int total = rows_pad * cols_pad, step = (int)blockDim.x, last = n_out * n_in - 1;
    for (int e0 = threadIdx.x; e0 < total; e0 += STAGE_BATCH * step) {
        float v[STAGE_BATCH];
        int at[STAGE_BATCH];
#pragma unroll
        for (int k = 0; k < STAGE_BATCH; ++k) {
            const int e = e0 + k * step;
            const int o = udiv_uniform(e, cols_pad), a = e - o * cols_pad;
            const int col = grp_L ? grp_col(a, grp_L, grp_F) : a;
            const bool ok = e < total && o < n_out && col >= 0 && col < n_in;
            at[k] = e < total ? o * stride + (permute ? swap23(a) : a) : -1;
            v[k] = W[min(max(o * n_in + col, 0), last)];
            v[k] = ok ? v[k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < STAGE_BATCH; ++k)
            if (at[k] >= 0) dst[at[k]] = (bf16_t)v[k];
    }
}
// Stage W^T: dst[row = input a][col = output o (permuted)]
__device__ void stage_weight_t(bf16_t *dst, int stride, int rows_pad, int cols_pad, const float *W, int n_out, int n_in,
                               int grp_L = 0, int grp_F = 0) {
    const int total = rows_pad * cols_pad, step = (int)blockDim.x, last = n_out * n_in - 1;
    for (int e0 = threadIdx.x; e0 < total; e0 += STAGE_BATCH * step) {
        float v[STAGE_BATCH];
        int at[STAGE_BATCH];
#pragma unroll
        for (int k = 0; k < STAGE_BATCH; ++k) {
            const int e = e0 + k * step;
            const int o = udiv_uniform(e, rows_pad), a = e - o * rows_pad;      // W's own row-major order (input index fastest): whole cache lines per
                                                                                // wave load, the transposition happens in the 2-byte LDS stores
            const int col = grp_L ? grp_col(a, grp_L, grp_F) : a;
            const bool ok = e < total && o < n_out && col >= 0 && col < n_in;
            at[k] = e < total ? a * stride + swap23(o) : -1;
            v[k] = W[min(max(o * n_in + col, 0), last)];
            v[k] = ok ? v[k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < STAGE_BATCH; ++k)
            if (at[k] >= 0) dst[at[k]] = (bf16_t)v[k];
    }
}

// Both images of one matrix from ONE pass over it: the natural image (as stage_weight: dst_s[o][a], optionally with the bit-2/3 swap) and the
// transposed one (as stage_weight_t: dst_t[a][o permuted]).  The backward kernels need W_0 (and W_1) both ways - recomputed forward and W^T chain.
__device__ void stage_weight_both(bf16_t *dst_s, int stride_s, bool permute_s, bf16_t *dst_t, int stride_t, int out_pad, int in_pad, const float *W,
                                  int n_out, int n_in, int grp_L = 0, int grp_F = 0) {
    const int total = out_pad * in_pad, step = (int)blockDim.x, last = n_out * n_in - 1;
    for (int e0 = threadIdx.x; e0 < total; e0 += STAGE_BATCH * step) {
        float v[STAGE_BATCH];
        int as[STAGE_BATCH], at[STAGE_BATCH];
#pragma unroll
        for (int k = 0; k < STAGE_BATCH; ++k) {
            const int e = e0 + k * step;
            const int o = udiv_uniform(e, in_pad), a = e - o * in_pad;
            const int col = grp_L ? grp_col(a, grp_L, grp_F) : a;
            const bool ok = e < total && o < n_out && col >= 0 && col < n_in;
            as[k] = e < total ? o * stride_s + (permute_s ? swap23(a) : a) : -1;
            at[k] = a * stride_t + swap23(o);
            v[k] = W[min(max(o * n_in + col, 0), last)];
            v[k] = ok ? v[k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < STAGE_BATCH; ++k)
            if (as[k] >= 0) {
                dst_s[as[k]] = (bf16_t)v[k];
                dst_t[at[k]] = (bf16_t)v[k];
            }
    }
}

__device__ __forceinline__ bf16x8 load8(const float *p) {
    f32x4 a = *reinterpret_cast<const f32x4 *>(p);
    f32x4 b = *reinterpret_cast<const f32x4 *>(p + 4);
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (bf16_t)a[j];
        r[j + 4] = (bf16_t)b[j];
    }
    return r;
}
__device__ __forceinline__ bf16x8 load8(const bf16_t *p) { return *reinterpret_cast<const bf16x8 *>(p); }

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)0.0f;
    return r;
}

// accumulator block -> two B fragments (k-steps 2*blk, 2*blk+1)
__device__ __forceinline__ void pack_block(const f32x16 &acc, bf16x8 &lo, bf16x8 &hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = (bf16_t)acc[j];
        hi[j] = (bf16_t)acc[8 + j];
    }
}

template <typename T>
__device__ __forceinline__ void store_block_full(T *row_ptr, int ch_base, int h, const f32x16 &acc);

// store one 32-row accumulator block of sample m as 4 groups of 4 consecutive channels
template <typename T>
__device__ __forceinline__ void store_block(T *row_ptr, int ch_base, int h, const f32x16 &acc, int n_valid, bool vec_ok) {
    if (vec_ok && ch_base + 32 <= n_valid) {
        store_block_full(row_ptr, ch_base, h, acc);
    } else if (vec_ok) {      // n_valid % 4 == 0: whole groups only
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = ch_base + 8 * g + 4 * h;
            if (c0 < n_valid) {
                if constexpr (sizeof(T) == 4) {
                    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(row_ptr + c0) = v;
                } else {
                    bf16x4 v = {(bf16_t)acc[4 * g], (bf16_t)acc[4 * g + 1], (bf16_t)acc[4 * g + 2], (bf16_t)acc[4 * g + 3]};
                    *reinterpret_cast<bf16x4 *>(row_ptr + c0) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = ch_base + 8 * g + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < n_valid) pag_st(row_ptr + c0 + j, acc[4 * g + j]);
        }
    }
}

template <typename T>
__device__ __forceinline__ void load_block(const T *row_ptr, int ch_base, int h, f32x16 &acc, int n_valid, bool vec_ok) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int c0 = ch_base + 8 * g + 4 * h;
        if (vec_ok && c0 + 3 < n_valid) {
            if constexpr (sizeof(T) == 4) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(row_ptr + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
            } else {
                bf16x4 v = *reinterpret_cast<const bf16x4 *>(row_ptr + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * g + j] = (float)v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = (c0 + j < n_valid) ? pag_ld(row_ptr + c0 + j) : 0.0f;
        }
    }
}

// two-phase variant of load_block: issue every load of a block first (raw registers), convert later, so that
// the compiler can keep all of a tile's loads in flight instead of waiting on each one before its conversion
template <typename T> struct RawVec { typedef f32x4 type; };
template <> struct RawVec<bf16_t> { typedef bf16x4 type; };

template <typename T>
__device__ __forceinline__ void load_block_raw(const T *row_ptr, int ch_base, int h, typename RawVec<T>::type (&raw)[4], int n_valid,
                                               bool vec_ok) {
    typedef typename RawVec<T>::type V;
    if (vec_ok && ch_base + 32 <= n_valid) {      // block wholly in range (all but the last block): plain vector loads
#pragma unroll
        for (int g = 0; g < 4; ++g) raw[g] = *reinterpret_cast<const V *>(row_ptr + ch_base + 8 * g + 4 * h);
    } else if (vec_ok) {      // n_valid % 4 == 0: a group is wholly in range or wholly out; branch-free (clamped address + select)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = ch_base + 8 * g + 4 * h;
            const bool ok = c0 < n_valid;
            V v = *reinterpret_cast<const V *>(row_ptr + (ok ? c0 : 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : (T)0.0f;
            raw[g] = v;
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = ch_base + 8 * g + 4 * h;
            V v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (c0 + j < n_valid) ? row_ptr[c0 + j] : (T)0.0f;
            raw[g] = v;
        }
    }
}
// whole 64-wide rows (hidden activations): no range checks at all
template <typename T>
__device__ __forceinline__ void load_block_full(const T *row_ptr, int ch_base, int h, f32x16 &acc) {
    typename RawVec<T>::type raw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) raw[g] = *reinterpret_cast<const typename RawVec<T>::type *>(row_ptr + ch_base + 8 * g + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = (float)raw[g][j];
}
template <typename T>
__device__ __forceinline__ void store_block_full(T *row_ptr, int ch_base, int h, const f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        typename RawVec<T>::type v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (T)acc[4 * g + j];
        *reinterpret_cast<typename RawVec<T>::type *>(row_ptr + ch_base + 8 * g + 4 * h) = v;
    }
}

template <typename V>
__device__ __forceinline__ void raw_to_block(const V (&raw)[4], f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = (float)raw[g][j];
}

// ---------------------------------------------------------------------------------------- LDS staging
// In the accumulator layout every lane owns ONE sample row, so a direct global access touches 32 different rows per
// wave-instruction: rocprofv3 showed the texture-address unit busy 75-85 % of these kernels' time (TA_BUSY_avr vs
// GRBM_GUI_ACTIVE).  Row-major [M,64] bf16 tiles (32 rows = 4 KiB contiguous) and 32-column blocks of the wide
// [M,W] tensors therefore go through a wave-private LDS buffer: global side = fully coalesced 16-byte-per-lane
// accesses, register side = ds_read/ds_write_b64 in accumulator layout.
constexpr int ST_RS = 72;                         // staging row stride in bf16 (144 B: conflict-light for b64 and b128)
constexpr int ST_BYTES = 32 * ST_RS * 2;          // 4608 B per wave

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// accumulator blocks acc[0..1] (64 columns) of a 32-row tile -> row-major [M,64] bf16 at gtile (= tensor + tile*32*64)
__device__ __forceinline__ void tile64_store(bf16_t *stg, bf16_t *gtile, int rows_valid, int lane, int r, int h, const f32x16 (&acc)[2]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v = {(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
            *reinterpret_cast<bf16x4 *>(stg + r * ST_RS + 32 * mb + 8 * g + 4 * h) = v;
        }
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(stg + row * ST_RS + c);
        if (row < rows_valid) *reinterpret_cast<bf16x8 *>(gtile + row * HID + c) = v;
    }
    wave_lds_sync();
}
// row-major [M,64] bf16 tile -> accumulator-layout raw pieces raw[mb][g]
__device__ __forceinline__ void tile64_load(bf16_t *stg, const bf16_t *gtile, int rows_valid, int lane, int r, int h, bf16x4 (&raw)[2][4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        bf16x8 v = zero8();
        if (row < rows_valid) v = *reinterpret_cast<const bf16x8 *>(gtile + row * HID + c);
        *reinterpret_cast<bf16x8 *>(stg + row * ST_RS + c) = v;
    }
    wave_lds_sync();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) raw[mb][g] = *reinterpret_cast<const bf16x4 *>(stg + r * ST_RS + 32 * mb + 8 * g + 4 * h);
    wave_lds_sync();
}
// tile64_load in two phases so that the global request of the NEXT tile can be in flight during the current tile's work
// (these kernels run 2 waves per SIMD: an exposed round trip per tile costs more than anything else in them)
__device__ __forceinline__ void tile64_fetch(const bf16_t *gtile, int rows_valid, int lane, bf16x8 (&v)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        v[it] = zero8();
        if (row < rows_valid) v[it] = *reinterpret_cast<const bf16x8 *>(gtile + row * HID + c);
    }
}
__device__ __forceinline__ void tile64_unstage(bf16_t *stg, const bf16x8 (&v)[4], int lane, int r, int h, bf16x4 (&raw)[2][4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        *reinterpret_cast<bf16x8 *>(stg + row * ST_RS + c) = v[it];
    }
    wave_lds_sync();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) raw[mb][g] = *reinterpret_cast<const bf16x4 *>(stg + r * ST_RS + 32 * mb + 8 * g + 4 * h);
    wave_lds_sync();
}
// columns [col0, col0+32) of rows [0,32) of a row-major [M,W] bf16 tensor (W % 8 == 0) at gtile (= tensor + tile*32*W):
// staged into LDS columns [lcol, lcol+32) (two tensors can share the buffer: lcol = 0 / 32)
__device__ __forceinline__ void block32_stage_in(bf16_t *stg, int lcol, const bf16_t *gtile, int W, int col0, int rows_valid, int lane) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = it * 16 + (lane >> 2), c = (lane & 3) * 8;
        bf16x8 v = zero8();
        if (row < rows_valid && col0 + c < W) v = *reinterpret_cast<const bf16x8 *>(gtile + (int64_t)row * W + col0 + c);
        *reinterpret_cast<bf16x8 *>(stg + row * ST_RS + lcol + c) = v;
    }
}
__device__ __forceinline__ void block32_read(const bf16_t *stg, int lcol, int r, int h, bf16x4 (&raw)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) raw[g] = *reinterpret_cast<const bf16x4 *>(stg + r * ST_RS + lcol + 8 * g + 4 * h);
}
// accumulator block -> columns [col0, col0+32) of the row-major [M,W] bf16 tensor
__device__ __forceinline__ void block32_store(bf16_t *stg, bf16_t *gtile, int W, int col0, int rows_valid, int lane, int r, int h, const f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 v = {(bf16_t)acc[4 * g], (bf16_t)acc[4 * g + 1], (bf16_t)acc[4 * g + 2], (bf16_t)acc[4 * g + 3]};
        *reinterpret_cast<bf16x4 *>(stg + r * ST_RS + 8 * g + 4 * h) = v;
    }
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = it * 16 + (lane >> 2), c = (lane & 3) * 8;
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(stg + row * ST_RS + c);
        if (row < rows_valid && col0 + c < W) *reinterpret_cast<bf16x8 *>(gtile + (int64_t)row * W + col0 + c) = v;
    }
    wave_lds_sync();
}

// upstream gradient block in rank-1 form: z[q] = scale * g_row[32ob + rho(q,h)] (g_row = this sample's ray row, f32 [n])
__device__ __forceinline__ void rank1_block(const float *g_row, float scale, int ch_base, int h, int n_valid, f32x16 &z) {
    const bool vec = (n_valid & 3) == 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = ch_base + 8 * g + 4 * h;
        if (vec) {
            const bool ok = c0 < n_valid;
            f32x4 v = *reinterpret_cast<const f32x4 *>(g_row + (ok ? c0 : 0));
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * g + j] = ok ? scale * v[j] : 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * g + j] = (c0 + j < n_valid) ? scale * g_row[c0 + j] : 0.0f;
        }
    }
}

// same, when every lane of the wave belongs to ONE ray: the row pointer is wave-uniform, so the 8 floats of a group
// (both lane halves) come from scalar loads and each lane picks its half - no vector-memory traffic at all
__device__ __forceinline__ void rank1_block_uniform(const float *g_row_u, float scale, int ch_base, int h, int n_valid, f32x16 &z) {
    // read-only for the whole launch: address space 4 (constant) lets the compiler use s_load for the uniform address
    typedef const float __attribute__((address_space(4))) cfloat;
    cfloat *gc = (cfloat *)(uintptr_t)g_row_u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = ch_base + 8 * g;                     // uniform
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c0 + j < n_valid) ? gc[c0 + j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) z[4 * g + j] = scale * (h ? v[4 + j] : v[j]);
    }
}

// hidden layer: acc[2] = bias + W(64 x K) . frags ; K = 16 * nks
template <int NKS>
__device__ __forceinline__ void hidden_layer(const bf16_t *Ws, const float *bs, const bf16x8 (&frag)[NKS], int nks, int r, int h,
                                             f32x16 (&acc)[2]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mb][q] = bs[32 * mb + rho(q, h)];
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            if (s < nks) {
                bf16x8 a = *reinterpret_cast<const bf16x8 *>(Ws + (32 * mb + r) * RS + 16 * s + 8 * h);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag[s], acc[mb], 0, 0, 0);
            }
        }
    }
}

// The same for the one-wave-per-SIMD fused backward kernels, with the instruction order PINNED.  Left to itself the scheduler emits
// "ds_read fragment, s_waitcnt lgkmcnt(0), v_mfma" once per k-step - a full LDS round trip in front of every MFMA, and a lone wave
// has nothing else to run meanwhile (20 such steps were half of a tile's time).  Here: every read of the first two k-steps first,
// then per k-step its two MFMAs (independent accumulators) while the reads two steps ahead are in flight.
template <int MBS, int NS, int LEAD_READS>
__device__ __forceinline__ void mfma_chain_pinned(const bf16_t *Wt, int stride, const bf16x8 (&frag)[NS], int r, int h, f32x16 (&acc)[2]) {
    bf16x8 a[MBS][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mb = 0; mb < MBS; ++mb) a[mb][s] = *reinterpret_cast<const bf16x8 *>(Wt + (32 * mb + r) * stride + 16 * s + 8 * h);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mb = 0; mb < MBS; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][s], frag[s], acc[mb], 0, 0, 0);
    constexpr int AHEAD = NS < 2 ? NS : 2;
    __builtin_amdgcn_sched_group_barrier(0x100, LEAD_READS + AHEAD * MBS, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, MBS, 0);
        if (s + AHEAD < NS) __builtin_amdgcn_sched_group_barrier(0x100, MBS, 0);
    }
}
template <int NS>
__device__ __forceinline__ void hidden_layer_pinned(const bf16_t *Ws, const float *bs, const bf16x8 (&frag)[NS], int r, int h, f32x16 (&acc)[2]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mb][q] = bs[32 * mb + rho(q, h)];
    mfma_chain_pinned<2, NS, 8>(Ws, RS, frag, r, h, acc);          // 8 leading reads: the bias blocks (4 x 16 bytes per block)
    __builtin_amdgcn_sched_barrier(0);
}
// acc = W^T-layout weight . fragments, from zero
template <int MBS, int NS>
__device__ __forceinline__ void wt_chain_pinned(const bf16_t *Wt, int stride, const bf16x8 (&frag)[NS], int r, int h, f32x16 (&acc)[2]) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < MBS; ++mb)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
    mfma_chain_pinned<MBS, NS, 0>(Wt, stride, frag, r, h, acc);
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------- swizzled tiles for the fused weight gradients
// dW = dz^T . a sums over the SAMPLES, which sit on the lanes in every register layout of the backward kernel - both MFMA
// operands need one transpose.  A wave keeps its 32-sample tiles ([32 samples][64 channels] bf16, 128-byte rows, 4 KiB) in LDS
// and reads operand fragments with ds_read_b64_tr_b16 (a 4-row x 16-column block per 16 lanes, delivered column-major).
// 8-byte chunk `c` of row `s` lives at chunk position c ^ tw_f(s): the accumulator-layout stores (lane = sample, 16 lanes =
// 16 rows, one chunk column) land in 16 different bank pairs, and the 4 rows of a transposed block use 4 different chunk
// groups, so neither access pattern has bank conflicts (the 32-lane ds_read_b64 of a whole column is 2-way).
typedef short i16x4 __attribute__((ext_vector_type(4)));
constexpr int TW_ELEMS = 32 * 64;
__device__ __forceinline__ int tw_f(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int tw_off(int row, int chunk) { return row * 64 + ((chunk ^ tw_f(row)) << 2); }
// row-major global pieces (tile64_fetch: piece `it` of a lane = row it*8 + lane/8, channels 8*(lane%8)..+7)
__device__ __forceinline__ void tw_put_rows(bf16_t *T, const bf16x8 (&v)[4], int lane) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), ch = (lane & 7) * 2;
        bf16x4 lo = {v[it][0], v[it][1], v[it][2], v[it][3]}, hi = {v[it][4], v[it][5], v[it][6], v[it][7]};
        *reinterpret_cast<bf16x4 *>(T + tw_off(row, ch)) = lo;
        *reinterpret_cast<bf16x4 *>(T + tw_off(row, ch + 1)) = hi;
    }
}
// forward-style B fragments of this lane's sample: xf[s] = channels 16s + 8h .. +7
__device__ __forceinline__ void tw_put_frags(bf16_t *T, const bf16x8 (&xf)[4], int r, int h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ch = 4 * s + 2 * h;
        bf16x4 lo = {xf[s][0], xf[s][1], xf[s][2], xf[s][3]}, hi = {xf[s][4], xf[s][5], xf[s][6], xf[s][7]};
        *reinterpret_cast<bf16x4 *>(T + tw_off(r, ch)) = lo;
        *reinterpret_cast<bf16x4 *>(T + tw_off(r, ch + 1)) = hi;
    }
}
// accumulator block mb (rows rho(q,h) + 32 mb of sample r)
__device__ __forceinline__ void tw_put_block(bf16_t *T, int mb, int r, int h, const f32x16 &acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 v = {(bf16_t)acc[4 * g], (bf16_t)acc[4 * g + 1], (bf16_t)acc[4 * g + 2], (bf16_t)acc[4 * g + 3]};
        *reinterpret_cast<bf16x4 *>(T + tw_off(r, 8 * mb + 2 * g + h)) = v;
    }
}
// dz of a hidden layer from a W^T chain's accumulators: ReLU mask (the layer's saved bf16 activation > 0, read from its image Th) and
// bf16 rounding on PAIRS - a post-ReLU half-word is 0 or positive, so min(max(h as i16, 0), 1) is its 0 / 1 mask (a -0 counts as not
// positive, like the float compare) and a 16-bit multiply applies it: 4 instructions per pair instead of 7, and ONE conversion serves
// both outputs - the next chain's B fragments hb[] and the transposed-read image Tz (blocks 0, 1).  Bit-identical to mask-in-fp32.
__device__ __forceinline__ void relu_mask_pack_put(const bf16_t *Th, bf16_t *Tz, int r, int h, bool live, const f32x16 (&acc)[2], bf16x8 (&hb)[4]) {
    typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        unsigned zp[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x2_t hv = *reinterpret_cast<const u32x2_t *>(Th + tw_off(r, 8 * mb + 2 * g + h));
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const bf16x2_t a2 = {(bf16_t)acc[mb][4 * g + 2 * d], (bf16_t)acc[mb][4 * g + 2 * d + 1]};
                unsigned m, o;
                asm("v_pk_max_i16 %0, %1, %2\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(m) : "v"(hv[d]), "s"(0u), "s"(0x00010001u));
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o) : "v"(__builtin_bit_cast(unsigned, a2)), "v"(m));
                zp[2 * g + d] = live ? o : 0u;
            }
            *reinterpret_cast<u32x2_t *>(Tz + tw_off(r, 8 * mb + 2 * g + h)) = u32x2_t{zp[2 * g], zp[2 * g + 1]};
        }
        hb[2 * mb] = __builtin_bit_cast(bf16x8, u32x4_t{zp[0], zp[1], zp[2], zp[3]});
        hb[2 * mb + 1] = __builtin_bit_cast(bf16x8, u32x4_t{zp[4], zp[5], zp[6], zp[7]});
    }
}
__device__ __forceinline__ void tw_get_raw(const bf16_t *T, int r, int h, bf16x4 (&raw)[2][4]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) raw[mb][g] = *reinterpret_cast<const bf16x4 *>(T + tw_off(r, 8 * mb + 2 * g + h));
}
// MFMA 32x32x16 operand fragment with k = sample: lane (c = lane & 31, h) gets samples 16ks + 8h .. +7 of channel 32 blk + c
__device__ __forceinline__ bf16x8 tw_frag(const bf16_t *T, int blk, int ks, int lane) {
    typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
    const int li = lane & 15, q = li >> 2, pp = li & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
    const int chunk = 8 * blk + 4 * g1 + pp, row0 = 16 * ks + 8 * h + q;
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(T + tw_off(row0, chunk)));
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4 *)(T + tw_off(row0 + 4, chunk)));
    union { i16x4 v[2]; bf16x8 f; } u;
    u.v[0] = lo;
    u.v[1] = hi;
    return u.f;
}
constexpr int WG_SLAB_COLS_F = 96;      // = WG_SLAB_COLS (declared with the weight-gradient kernels below)
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ------------------------------------------------------------------------------------------ forward
template <typename X1T, typename OutT, int NL, int OBMAX>
#ifndef PAG_FWD_WAVES_WIDE
#define PAG_FWD_WAVES_WIDE 3
#endif
#ifndef PAG_FWD_WAVES_NARROW
#define PAG_FWD_WAVES_NARROW 1
#endif
__global__ __launch_bounds__(256, (OBMAX > 2 ? PAG_FWD_WAVES_WIDE : PAG_FWD_WAVES_NARROW)) void mlp_fwd_mfma(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OB = (p.out_dim + 31) / 32;
    bf16_t *W0s = reinterpret_cast<bf16_t *>(smem);                  // [64][RS] natural k
    bf16_t *W1s = W0s + 64 * RS;                                     // [64][RS] permuted k (NL == 3)
    bf16_t *WLs = W1s + (NL == 3 ? 64 * RS : 0);                     // [OB*32][RS] permuted k
    float *b0s = reinterpret_cast<float *>(WLs + OB * 32 * RS);
    float *b1s = b0s + 64;
    float *bLs = b1s + 64;
    bf16_t *stg = reinterpret_cast<bf16_t *>(bLs + OB * 32) + (threadIdx.x >> 6) * (ST_BYTES / 2);      // wave-private staging tile

    stage_weight(W0s, RS, 64, 64, p.W[0], HID, p.in_dim, false, p.grp_L, p.grp_F);
    if (NL == 3) stage_weight(W1s, RS, 64, 64, p.W[1], HID, HID, true);
    stage_weight(WLs, RS, OB * 32, 64, p.W[NL - 1], p.out_dim, HID, true);
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0s[e] = p.b[0][e];
        b1s[e] = NL == 3 ? p.b[1][e] : 0.0f;
    }
    for (int e = threadIdx.x; e < OB * 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b[NL - 1][e] : 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int r = lane & 31, h = lane >> 5;
    const int nks0 = p.in_pad / 16;
    const int64_t ntiles = (p.M + 31) / 32;
    const bool vec_out = (p.out_dim % 4) == 0;
    const X1T *x1 = reinterpret_cast<const X1T *>(p.x1);
    OutT *out = reinterpret_cast<OutT *>(p.out);

    // layer-0 B fragments straight from memory (features 16s + 8h .. +7 of sample m); the NEXT tile's fragments are
    // requested before the current tile is computed so that their HBM latency hides under the MFMA / epilogue work
    auto load_x = [&](int64_t tile, bf16x8 (&xf)[4]) {
        const int64_t m = tile * 32 + r;
        const bool live = tile < ntiles && m < p.M;
        const int64_t mc = live ? m : 0;
        const int32_t ray = (p.x2 && live) ? p.x2_index[mc] : 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int f0 = 16 * s + 8 * h;
            if (p.grp_L && live)
                xf[s] = load8(reinterpret_cast<const bf16_t *>(p.x1) + ((int64_t)(2 * s + h) * p.M + mc) * 8);
            else if (s < nks0 && live && f0 < p.k1)
                xf[s] = load8(x1 + mc * p.k1 + f0);
            else if (s < nks0 && live && f0 < p.k1 + p.k2p)
                xf[s] = load8(p.x2 + (int64_t)ray * p.k2p + (f0 - p.k1));
            else
                xf[s] = zero8();
        }
    };
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    bf16x8 xnext[4];
    load_x((int64_t)blockIdx.x * 4 + wave, xnext);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += tile_step) {
        const int64_t m = tile * 32 + r;
        const bool live = m < p.M;
        bf16x8 xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = xnext[s];
        load_x(tile + tile_step, xnext);
        if constexpr (OBMAX <= 2) {     // narrow decoders only (the colour decoder): density = relu(column 0 of its input)
            if (p.col0_relu && h == 0 && live) p.col0_relu[m] = fmaxf((float)xb[0][0], 0.0f);
        }
        if constexpr (OBMAX > 2) {      // keep lane-constant addresses from being hoisted out of the tile loop and spilled
            asm volatile("" : "+v"(r), "+v"(h));
        }
        f32x16 acc[2];
        bf16x8 hb[4];
        hidden_layer<4>(W0s, b0s, xb, nks0, r, h, acc);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = fmaxf(acc[mb][q], 0.0f);
            pack_block(acc[mb], hb[2 * mb], hb[2 * mb + 1]);
        }
        const int rows_valid = (int)min((int64_t)32, p.M - tile * 32);
        if (p.hsave[0]) tile64_store(stg, reinterpret_cast<bf16_t *>(p.hsave[0]) + tile * 32 * HID, rows_valid, lane, r, h, acc);
        if (NL == 3) {
            hidden_layer<4>(W1s, b1s, hb, 4, r, h, acc);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][q] = fmaxf(acc[mb][q], 0.0f);
                pack_block(acc[mb], hb[2 * mb], hb[2 * mb + 1]);
            }
            if (p.hsave[1]) tile64_store(stg, reinterpret_cast<bf16_t *>(p.hsave[1]) + tile * 32 * HID, rows_valid, lane, r, h, acc);
        }
        // ---- output layer
        auto out_block = [&](int ob, f32x16 &o) __attribute__((always_inline)) {      // bias + W_L[32ob .. 32ob+31, :] . h : 4 MFMAs
#pragma unroll
            for (int q = 0; q < 16; ++q) o[q] = bLs[32 * ob + rho(q, h)];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a = *reinterpret_cast<const bf16x8 *>(WLs + (32 * ob + r) * RS + 16 * s + 8 * h);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[s], o, 0, 0, 0);
            }
        };
        constexpr float LOG2E = 1.4426950408889634f;
        if constexpr (OBMAX > 2) {
            // Wide heads (e.g. 200 instance logits): ONE 32-channel block is live at a time.  Softmax is computed online
            // (running max / sum over the blocks), then the blocks are recomputed (4 MFMAs each - the matrix cores are
            // idle anyway) to be normalised and stored.  Holding all 7 blocks took 256 VGPRs = 1 wave per SIMD with every
            // latency exposed; this form fits 128 VGPRs = 4 waves per SIMD.
            float M_ = 0.0f, inv = 1.0f;
            if (p.act == PAG_ACT_SOFTMAX) {
                float mrun = -INFINITY, srun = 0.0f;
                for (int ob = 0; ob < OB; ++ob) {
                    f32x16 o;
                    out_block(ob, o);
                    const bool lastb = ob == OB - 1;
                    float bm = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (!lastb || 32 * ob + rho(q, h) < p.out_dim) bm = fmaxf(bm, o[q]);
                    const float mn = fmaxf(mrun, bm);
                    float bs = 0.0f;
                    const float mns = mn * LOG2E;
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (!lastb || 32 * ob + rho(q, h) < p.out_dim) bs += __builtin_amdgcn_exp2f(fmaf(o[q], LOG2E, -mns));
                    srun = srun * __builtin_amdgcn_exp2f((mrun - mn) * LOG2E) + bs;
                    mrun = mn;
                }
                const float mo = __shfl_xor(mrun, 32), so = __shfl_xor(srun, 32);
                M_ = fmaxf(mrun, mo);
                const float S_ = srun * __builtin_amdgcn_exp2f((mrun - M_) * LOG2E) + so * __builtin_amdgcn_exp2f((mo - M_) * LOG2E);
                inv = 1.0f / S_;
            }
            const float Ms = M_ * LOG2E;
            if (p.stats && p.act == PAG_ACT_SOFTMAX && live && h == 0) {
                float2 st2 = {Ms, inv};
                *reinterpret_cast<float2 *>(p.stats + 2 * m) = st2;
            }
            for (int ob = 0; ob < (p.out ? OB : 0); ++ob) {      // out == NULL: statistics only (pag_head_composite_fwd rebuilds)
                f32x16 o;
                out_block(ob, o);
                if (p.act == PAG_ACT_SOFTMAX) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) o[q] = __builtin_amdgcn_exp2f(fmaf(o[q], LOG2E, -Ms)) * inv;
                } else if (p.act == PAG_ACT_SIGMOID) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) o[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * o[q]));
                }
                if constexpr (sizeof(OutT) == 2) {
                    if ((p.out_dim & 7) == 0)
                        block32_store(stg, reinterpret_cast<bf16_t *>(p.out) + tile * 32 * p.out_dim, p.out_dim, 32 * ob, rows_valid, lane, r, h, o);
                    else if (live)
                        store_block(out + m * p.out_dim, 32 * ob, h, o, p.out_dim, vec_out);
                } else {
                    if (live) store_block(out + m * p.out_dim, 32 * ob, h, o, p.out_dim, vec_out);
                }
            }
        } else {
            f32x16 o[OBMAX];
#pragma unroll
            for (int ob = 0; ob < OBMAX; ++ob)
                if (ob < OB) out_block(ob, o[ob]);
            // Output activations on the accumulators.  The bf16 path uses the hardware exp2 / rcp (v_exp_f32, v_rcp_f32:
            // ~1 ulp) - an accurate expf() and a true division per element made this epilogue 5400 VALU instructions per
            // 32-sample tile and the kernel VALU-bound (rocprofv3 SQ_INSTS_VALU); only the last block can hold padding rows.
            if (p.act == PAG_ACT_SIGMOID) {
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q) o[ob][q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * o[ob][q]));
            } else if (p.act == PAG_ACT_SOFTMAX) {
                float mx = -INFINITY;
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if (ob < OB - 1 || 32 * ob + rho(q, h) < p.out_dim) mx = fmaxf(mx, o[ob][q]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mxs = mx * LOG2E;
                float sum = 0.0f;
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float e = (ob < OB - 1 || 32 * ob + rho(q, h) < p.out_dim) ? __builtin_amdgcn_exp2f(fmaf(o[ob][q], LOG2E, -mxs)) : 0.0f;
                            o[ob][q] = e;
                            sum += e;
                        }
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q) o[ob][q] *= inv;
            }
            if (live) {
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB) store_block(out + m * p.out_dim, 32 * ob, h, o[ob], p.out_dim, vec_out);
            }
        }
    }
}

// ------------------------------------------------------------------ forward, production decoder shapes: straight-line tile loops
// mlp_fwd_mfma above serves every layout / dtype / activation through run-time branches: 62 branches per tile, and past every join the
// compiler waits for ALL loads in flight (it no longer counts them) - the next tile's prefetch was waited for right after it was
// issued; 45 - 58 % VALU-busy, the rest stalls.  The decoder shapes of the panoptic nef (pc_nerf/panoptic_nef.py:114-164: density-like,
// colour-like, semantic-like, the wide instance head) get dedicated kernels, like their backward (mlp_bwd_fused): no branch in the tile
// loop at all.  Every global access goes through a buffer descriptor - out-of-range lanes (rows past M, channels past out_dim, the
// half-waves that hold no output channel) get an offset past the end: their loads return 0, their stores are dropped - the weight chains
// use the pinned read-ahead order, and the element-wise arithmetic is written on pairs (packed fp32).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned BUF_OOB = 0x80000000u;         // added to an offset < 2^30: past any descriptor's range (tensors here are <= 2^30 bytes:
constexpr unsigned BUF_OOB_ROW = 0x40000000u;     // M <= 2^24 rows of <= 64 bytes).  Dead PIECE + dead ROW = 0xC0000000: the sum must not wrap to 0

__device__ __forceinline__ void out_block_pinned(const bf16_t *WLs, const float *bLs, int ob, const bf16x8 (&hb)[4], int r, int h, f32x16 &o) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) o[q] = bLs[32 * ob + rho(q, h)];
    bf16x8 a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const bf16x8 *>(WLs + (32 * ob + r) * RS + 16 * s + 8 * h);
#pragma unroll
    for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], hb[s], o, 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);          // 4 bias reads + 4 weight fragments, then the chain
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
}
// relu + bf16 B fragments of the next layer
__device__ __forceinline__ void relu_pack(f32x16 (&acc)[2], bf16x8 (&hb)[4]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mb][q] = fmaxf(acc[mb][q], 0.0f);
        pack_block(acc[mb], hb[2 * mb], hb[2 * mb + 1]);
    }
}
// accumulator blocks (64 columns, 32 rows) -> row-major [M,64] bf16 through the wave's staging buffer; rows past M are dropped
template <typename RS_T>
__device__ __forceinline__ void tile64_store_buf(bf16_t *stg, RS_T rs, unsigned tile_off, int lane, int r, int h, const f32x16 (&acc)[2]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v = {(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
            *reinterpret_cast<bf16x4 *>(stg + r * ST_RS + 32 * mb + 8 * g + 4 * h) = v;
        }
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stg + row * ST_RS + c);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, tile_off + row * (HID * 2) + c * 2, 0, 0);
    }
    wave_lds_sync();
}

// the same with only the first rows_valid rows written (tiles that end with their ray)
template <typename RS_T>
__device__ __forceinline__ void tile64_store_buf_rows(bf16_t *stg, RS_T rs, unsigned tile_off, int rows_valid, int lane, int r, int h, const f32x16 (&acc)[2]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v = {(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
            *reinterpret_cast<bf16x4 *>(stg + r * ST_RS + 32 * mb + 8 * g + 4 * h) = v;
        }
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c = (lane & 7) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stg + row * ST_RS + c);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, row < rows_valid ? tile_off + row * (HID * 2) + c * 2 : BUF_OOB, 0, 0);
    }
    wave_lds_sync();
}

// KIND 0: XCD8 bf16 input, no output activation, bf16 out, out_dim % 4 == 0 (<= 32)                 (density decoder)
// KIND 1: bf16 x1 [M,16] + f32 x2 [R,32] through x2_index, sigmoid, f32 out, out_dim <= 4, col0_relu   (colour decoder)
// KIND 2: XCD8 bf16 input, softmax, bf16 out, out_dim <= 8                                             (semantic head)
// SAVE: write the hidden activations (hsave[0], and hsave[1] with three layers) - the backward that recomputes them passes none.
template <int NL, int KIND, bool SAVE>
__global__ __launch_bounds__(256, 2) void mlp_fwd_fast(FwdParams p) {
    PAG_BLOCK_TIMER(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool GRP = KIND != 1;
    constexpr int NKS0 = GRP ? 4 : 3;
    constexpr float LOG2E = 1.4426950408889634f;
    bf16_t *W0s = reinterpret_cast<bf16_t *>(smem);                  // [64][RS] natural k
    bf16_t *W1s = W0s + 64 * RS;                                     // [64][RS] permuted k (NL == 3)
    bf16_t *WLs = W1s + (NL == 3 ? 64 * RS : 0);                     // [32][RS] permuted k
    float *b0s = reinterpret_cast<float *>(WLs + 32 * RS);
    float *b1s = b0s + 64;
    float *bLs = b1s + 64;
    bf16_t *stg = reinterpret_cast<bf16_t *>(bLs + 32) + (threadIdx.x >> 6) * (ST_BYTES / 2);      // wave-private staging tile
    stage_weight(W0s, RS, 64, 64, p.W[0], HID, p.in_dim, false, p.grp_L, p.grp_F);
    if (NL == 3) stage_weight(W1s, RS, 64, 64, p.W[1], HID, HID, true);
    stage_weight(WLs, RS, 32, 64, p.W[NL - 1], p.out_dim, HID, true);
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0s[e] = p.b[0][e];
        b1s[e] = NL == 3 ? p.b[1][e] : 0.0f;
    }
    for (int e = threadIdx.x; e < 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b[NL - 1][e] : 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t M = p.M, ntiles = (M + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    const auto rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x1), 0, (int)(GRP ? M * 128 : M * 32), 0x00020000);
    const auto rs_xi = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(KIND == 1 ? p.x2_index : nullptr), 0, (int)(M * 4), 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(M * p.out_dim * (KIND == 1 ? 4 : 2)), 0x00020000);
    const auto rs_c0 = __builtin_amdgcn_make_buffer_rsrc(KIND == 1 ? p.col0_relu : nullptr, 0, (int)(M * 4), 0x00020000);
    const auto rs_h0 = __builtin_amdgcn_make_buffer_rsrc(SAVE ? p.hsave[0] : nullptr, 0, (int)(M * HID * 2), 0x00020000);
    const auto rs_h1 = __builtin_amdgcn_make_buffer_rsrc(SAVE && NL == 3 ? p.hsave[1] : nullptr, 0, (int)(M * HID * 2), 0x00020000);
    // loop-invariant lane offsets
    unsigned xoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xoff[s] = GRP ? (unsigned)(((int64_t)(2 * s + h) * M + r) * 16) : (unsigned)(r * 32 + h * 16);
    unsigned ooff[4];            // per output piece of this lane; BUF_OOB when the piece is past out_dim
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (KIND == 0) ooff[g] = (8 * g + 4 * h < p.out_dim) ? (unsigned)((r * p.out_dim + 8 * g + 4 * h) * 2) : BUF_OOB;
        else ooff[g] = (4 * h + g < p.out_dim) ? (unsigned)((r * p.out_dim + 4 * h + g) * (KIND == 1 ? 4 : 2)) : BUF_OOB;
    }
    const unsigned c0off = h == 0 ? (unsigned)(r * 4) : BUF_OOB;

    // the next tile's layer-0 fragments are requested before the current tile is computed (colour: the view-embedding row index
    // runs one tile further ahead, so that the gather it addresses never waits for it inside an iteration)
    auto row_live = [&](int64_t tile) __attribute__((always_inline)) { return tile * 32 + r < M; };
    auto load_x = [&](int64_t tile, int ray, bf16x8 (&xf)[4]) __attribute__((always_inline)) {
        const unsigned base = row_live(tile) ? (unsigned)(tile * 32) * (GRP ? 16u : 32u) : BUF_OOB;
        if constexpr (GRP) {
#pragma unroll
            for (int s = 0; s < 4; ++s) xf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, xoff[s] + base, 0, 0));
        } else {
            xf[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, xoff[0] + base, 0, 0));
            const float *row = p.x2 + (int64_t)ray * p.k2p + 8 * h;          // ray is a valid index for every lane (0 for rows past M)
#pragma unroll
            for (int s = 1; s < 3; ++s) xf[s] = load8(row + 16 * (s - 1));
            xf[3] = zero8();
        }
    };
    auto load_ray = [&](int64_t tile) __attribute__((always_inline)) {
        if constexpr (KIND != 1) return 0;
        else return (int)__builtin_amdgcn_raw_buffer_load_b32(rs_xi, row_live(tile) ? (unsigned)(tile * 32 + r) * 4u : BUF_OOB, 0, 0);
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    int ray1 = load_ray(tile + tile_step);
    bf16x8 xn[4];
    load_x(tile, load_ray(tile), xn);

    for (; tile < ntiles; tile += tile_step) {
        const bool live = row_live(tile);
        const unsigned row0 = (unsigned)(tile * 32);
        bf16x8 xb[NKS0 < 4 ? 3 : 4];
#pragma unroll
        for (int s = 0; s < NKS0; ++s) xb[s] = xn[s];
        load_x(tile + tile_step, ray1, xn);
        ray1 = load_ray(tile + 2 * tile_step);
        if constexpr (KIND == 1)      // density = relu(column 0 of the colour decoder's input)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf((float)xb[0][0], 0.0f)), rs_c0, live ? c0off + row0 * 4 : BUF_OOB, 0, 0);
        f32x16 acc[2];
        bf16x8 hb[4];
        hidden_layer_pinned<NKS0>(W0s, b0s, xb, r, h, acc);
        relu_pack(acc, hb);
        if constexpr (SAVE) tile64_store_buf(stg, rs_h0, row0 * (HID * 2), lane, r, h, acc);
        if constexpr (NL == 3) {
            hidden_layer_pinned<4>(W1s, b1s, hb, r, h, acc);
            relu_pack(acc, hb);
            if constexpr (SAVE) tile64_store_buf(stg, rs_h1, row0 * (HID * 2), lane, r, h, acc);
        }
        f32x16 o;
        out_block_pinned(WLs, bLs, 0, hb, r, h, o);
        const unsigned obase = live ? row0 * (unsigned)(p.out_dim * (KIND == 1 ? 4 : 2)) : BUF_OOB_ROW;
        if constexpr (KIND == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
                const u32x2 v = {__builtin_bit_cast(unsigned, bf16x2{(bf16_t)o[4 * g], (bf16_t)o[4 * g + 1]}),
                                 __builtin_bit_cast(unsigned, bf16x2{(bf16_t)o[4 * g + 2], (bf16_t)o[4 * g + 3]})};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs_out, ooff[g] + obase, 0, 0);
            }
        } else if constexpr (KIND == 1) {
            // hardware exp2 / rcp (~1 ulp), as the generic kernel
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * o[j]));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs_out, ooff[j] + obase, 0, 0);
            }
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * h + j < p.out_dim) mx = fmaxf(mx, o[j]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mxs = mx * LOG2E;
            float e[4], sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = (4 * h + j < p.out_dim) ? __builtin_amdgcn_exp2f(fmaf(o[j], LOG2E, -mxs)) : 0.0f;
                sum += e[j];
            }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t y = (bf16_t)(e[j] * inv);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, y), rs_out, ooff[j] + obase, 0, 0);
            }
        }
    }
}

// Density decoder and colour decoder in ONE launch (pag_mlp_fwd_args.x1_producer): the colour decoder's x1 is the density decoder's [M,16] output
// (pc_nerf/panoptic_delta_nef.py:184 -> :198-203).  As two launches the second re-reads what the first just wrote and each pays its own ramp and tail
// (81 + 58 us per 2.1 M samples).  Here a wave runs mlp_fwd_fast<2, 0> and then mlp_fwd_fast<3, 1> on its tile: the density features are written as
// before (the backward reads them) and handed to the colour decoder in registers - as the bf16 values the store rounds to, brought from the accumulator
// layout (a lane holds channels 0-3, 8-11 or 4-7, 12-15 of its sample) to the natural k order of the standalone launch (8 consecutive channels per
// half) by two v_permlane32_swap: same fragments, same MFMA sequence, bit-identical outputs - the backward's recomputation stays consistent.
#ifndef PAG_CD_WAVES
#define PAG_CD_WAVES 2      // waves per SIMD asked of the compiler (146 VGPRs: three resident, grid cap 768 = one round; asking for 4 = 128 VGPRs: 123 us against 98; caps 512 / 1024: 101 / 105)
#endif
__global__ __launch_bounds__(256, PAG_CD_WAVES) void mlp_fwd_density_colour(FwdParams pd, FwdParams pc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr float LOG2E = 1.4426950408889634f;
    bf16_t *W0d = reinterpret_cast<bf16_t *>(smem);                  // density: [64][RS] natural k, [32][RS] permuted k
    bf16_t *WLd = W0d + 64 * RS;
    bf16_t *W0c = WLd + 32 * RS;                                     // colour: [64][RS] natural k, [64][RS] permuted k, [32][RS] permuted k
    bf16_t *W1c = W0c + 64 * RS;
    bf16_t *WLc = W1c + 64 * RS;
    float *b0d = reinterpret_cast<float *>(WLc + 32 * RS);
    float *bLd = b0d + 64;
    float *b0c = bLd + 32;
    float *b1c = b0c + 64;
    float *bLc = b1c + 64;
    stage_weight(W0d, RS, 64, 64, pd.W[0], HID, pd.in_dim, false, pd.grp_L, pd.grp_F);
    stage_weight(WLd, RS, 32, 64, pd.W[1], pd.out_dim, HID, true);
    stage_weight(W0c, RS, 64, 64, pc.W[0], HID, pc.in_dim, false);
    stage_weight(W1c, RS, 64, 64, pc.W[1], HID, HID, true);
    stage_weight(WLc, RS, 32, 64, pc.W[2], pc.out_dim, HID, true);
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0d[e] = pd.b[0][e];
        b0c[e] = pc.b[0][e];
        b1c[e] = pc.b[1][e];
    }
    for (int e = threadIdx.x; e < 32; e += blockDim.x) {
        bLd[e] = e < pd.out_dim ? pd.b[1][e] : 0.0f;
        bLc[e] = e < pc.out_dim ? pc.b[2][e] : 0.0f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t M = pd.M, ntiles = (M + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(pd.x1), 0, (int)(M * 128), 0x00020000);
    const auto rs_xi = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(pc.x2_index), 0, (int)(M * 4), 0x00020000);
    const auto rs_od = __builtin_amdgcn_make_buffer_rsrc(pd.out, 0, (int)(M * pd.out_dim * 2), 0x00020000);
    const auto rs_oc = __builtin_amdgcn_make_buffer_rsrc(pc.out, 0, (int)(M * pc.out_dim * 4), 0x00020000);
    const auto rs_c0 = __builtin_amdgcn_make_buffer_rsrc(pc.col0_relu, 0, (int)(M * 4), 0x00020000);
    unsigned xoff[4], ooff_d[4], ooff_c[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xoff[s] = (unsigned)(((int64_t)(2 * s + h) * M + r) * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        ooff_d[g] = (8 * g + 4 * h < pd.out_dim) ? (unsigned)((r * pd.out_dim + 8 * g + 4 * h) * 2) : BUF_OOB;
        ooff_c[g] = (4 * h + g < pc.out_dim) ? (unsigned)((r * pc.out_dim + 4 * h + g) * 4) : BUF_OOB;
    }
    const unsigned c0off = h == 0 ? (unsigned)(r * 4) : BUF_OOB;
    auto row_live = [&](int64_t tile) __attribute__((always_inline)) { return tile * 32 + r < M; };
    auto load_x = [&](int64_t tile, int ray, bf16x8 (&xf)[4], bf16x8 (&pe)[2]) __attribute__((always_inline)) {
        const unsigned base = row_live(tile) ? (unsigned)(tile * 32) * 16u : BUF_OOB;
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[s] + base, 0, 0));
        const float *row = pc.x2 + (int64_t)ray * pc.k2p + 8 * h;          // ray is a valid index for every lane (0 for rows past M)
        pe[0] = load8(row);
        pe[1] = load8(row + 16);
    };
    auto load_ray = [&](int64_t tile) __attribute__((always_inline)) {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rs_xi, row_live(tile) ? (unsigned)(tile * 32 + r) * 4u : BUF_OOB, 0, 0);
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    int ray1 = load_ray(tile + tile_step);
    bf16x8 xn[4], pen[2];
    load_x(tile, load_ray(tile), xn, pen);
    for (; tile < ntiles; tile += tile_step) {
        const bool live = row_live(tile);
        const unsigned row0 = (unsigned)(tile * 32);
        bf16x8 xb[4], xc[3];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = xn[s];
        xc[1] = pen[0];
        xc[2] = pen[1];
        load_x(tile + tile_step, ray1, xn, pen);
        ray1 = load_ray(tile + 2 * tile_step);
        // ---- density decoder (mlp_fwd_fast<2, 0, false>)
        f32x16 acc[2], o;
        bf16x8 hb[4];
        hidden_layer_pinned<4>(W0d, b0d, xb, r, h, acc);
        relu_pack(acc, hb);
        out_block_pinned(WLd, bLd, 0, hb, r, h, o);
        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
        unsigned dw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dw[k] = __builtin_bit_cast(unsigned, bf16x2{(bf16_t)o[2 * k], (bf16_t)o[2 * k + 1]});
        const unsigned obase_d = live ? row0 * (unsigned)(pd.out_dim * 2) : BUF_OOB_ROW;
#pragma unroll
        for (int g = 0; g < 4; ++g) __builtin_amdgcn_raw_buffer_store_b64(u32x2{dw[2 * g], dw[2 * g + 1]}, rs_od, ooff_d[g] + obase_d, 0, 0);
        // ---- its 16 output channels as the colour decoder's first k-step, natural order: half 0 = channels 0-7, half 1 = channels 8-15
        {
            const auto s02 = __builtin_amdgcn_permlane32_swap(dw[0], dw[2], false, false);      // dw[0] upper half <-> dw[2] lower half
            const auto s13 = __builtin_amdgcn_permlane32_swap(dw[1], dw[3], false, false);
            const u32x4 v = {s02[0], s13[0], s02[1], s13[1]};
            xc[0] = __builtin_bit_cast(bf16x8, v);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf((float)xc[0][0], 0.0f)), rs_c0, live ? c0off + row0 * 4 : BUF_OOB, 0, 0);
        // ---- colour decoder (mlp_fwd_fast<3, 1, false>)
        hidden_layer_pinned<3>(W0c, b0c, xc, r, h, acc);
        relu_pack(acc, hb);
        hidden_layer_pinned<4>(W1c, b1c, hb, r, h, acc);
        relu_pack(acc, hb);
        out_block_pinned(WLc, bLc, 0, hb, r, h, o);
        const unsigned obase_c = live ? row0 * (unsigned)(pc.out_dim * 4) : BUF_OOB_ROW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * o[j]));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs_oc, ooff_c[j] + obase_c, 0, 0);
        }
    }
}

// Wide softmax head (three layers, XCD8 input, 192 < out_dim <= 224), statistics only: the forward of the instance head writes the
// per-sample (max logit * log2e, 1 / sum exp) and the last hidden layer; pag_head_composite_fwd and the backward rebuild the
// probabilities from them.  One 32-channel block is live at a time (online softmax over the blocks); block k+1's MFMAs are issued
// before block k's exponentials so that the matrix pipe runs under them.
// PAIR: a second, narrow softmax decoder on the same input (the semantic head next to the instance head: both read the panoptic
// features) is evaluated on the tile while it is in registers - its own launch was one more 268 MB read of the features.  The weights
// of both decoders then take 66 KiB: 8 waves share them (512 threads, one workgroup per CU) instead of two 4-wave workgroups.
template <bool SAVE0, bool PAIR>
#ifndef PAG_WIDE_FWD_PAIR_THREADS
#define PAG_WIDE_FWD_PAIR_THREADS 1024
#endif
__global__ __launch_bounds__(PAIR ? PAG_WIDE_FWD_PAIR_THREADS : 256, PAIR ? 1 : 2) void mlp_fwd_wide_stats(FwdParams p) {
    PAG_BLOCK_TIMER(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int OB = 7;
    constexpr float LOG2E = 1.4426950408889634f;
    const int NW = blockDim.x >> 6;
    bf16_t *W0s = reinterpret_cast<bf16_t *>(smem);
    bf16_t *W1s = W0s + 64 * RS;
    bf16_t *WLs = W1s + 64 * RS;                                     // [OB*32][RS] permuted k
    bf16_t *W0s2 = WLs + OB * 32 * RS;                               // PAIR: [64][RS] natural k, [32][RS] permuted k
    bf16_t *WLs2 = W0s2 + (PAIR ? 64 * RS : 0);
    float *b0s = reinterpret_cast<float *>(WLs2 + (PAIR ? 32 * RS : 0));
    float *b1s = b0s + 64;
    float *bLs = b1s + 64;
    float *b0s2 = bLs + OB * 32;
    float *bLs2 = b0s2 + (PAIR ? 64 : 0);
    bf16_t *stg = reinterpret_cast<bf16_t *>(bLs2 + (PAIR ? 32 : 0)) + (threadIdx.x >> 6) * (ST_BYTES / 2);
    stage_weight(W0s, RS, 64, 64, p.W[0], HID, p.in_dim, false, p.grp_L, p.grp_F);
    stage_weight(W1s, RS, 64, 64, p.W[1], HID, HID, true);
    stage_weight(WLs, RS, OB * 32, 64, p.W[2], p.out_dim, HID, true);
    if constexpr (PAIR) {
        stage_weight(W0s2, RS, 64, 64, p.W2[0], HID, p.in_dim, false, p.grp_L, p.grp_F);
        stage_weight(WLs2, RS, 32, 64, p.W2[1], p.out2_dim, HID, true);
        for (int e = threadIdx.x; e < 64; e += blockDim.x) b0s2[e] = p.b2[0][e];
        for (int e = threadIdx.x; e < 32; e += blockDim.x) bLs2[e] = e < p.out2_dim ? p.b2[1][e] : 0.0f;
    }
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0s[e] = p.b[0][e];
        b1s[e] = p.b[1][e];
    }
    // padding channels: a bias of -1e30 makes their exponential exactly 0 and never the maximum
    for (int e = threadIdx.x; e < OB * 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b[2][e] : -1e30f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t M = p.M, ntiles = (M + 31) / 32;
    const int64_t tile_step = (int64_t)gridDim.x * NW;
    const auto rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x1), 0, (int)(M * 128), 0x00020000);
    const auto rs_st = __builtin_amdgcn_make_buffer_rsrc(p.stats, 0, (int)(M * 8), 0x00020000);
    const auto rs_o2 = __builtin_amdgcn_make_buffer_rsrc(PAIR ? p.out2 : nullptr, 0, (int)(M * (PAIR ? p.out2_dim : 0) * 2), 0x00020000);
    unsigned ooff2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ooff2[j] = (PAIR && 4 * h + j < p.out2_dim) ? (unsigned)((r * p.out2_dim + 4 * h + j) * 2) : BUF_OOB;
    const auto rs_h0 = __builtin_amdgcn_make_buffer_rsrc(SAVE0 ? p.hsave[0] : nullptr, 0, (int)(M * HID * 2), 0x00020000);
    const auto rs_h1 = __builtin_amdgcn_make_buffer_rsrc(p.hsave[1], 0, (int)(M * HID * 2), 0x00020000);
    unsigned xoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xoff[s] = (unsigned)(((int64_t)(2 * s + h) * M + r) * 16);
    const unsigned soff = h == 0 ? (unsigned)(r * 8) : BUF_OOB;
    auto row_live = [&](int64_t tile) __attribute__((always_inline)) { return tile * 32 + r < M; };
    auto load_x = [&](int64_t tile, bf16x8 (&xf)[4]) __attribute__((always_inline)) {
        const unsigned base = row_live(tile) ? (unsigned)(tile * 32) * 16u : BUF_OOB;
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, xoff[s] + base, 0, 0));
    };
    int64_t tile = (int64_t)blockIdx.x * NW + wave;
    bf16x8 xn[4];
    load_x(tile, xn);
    for (; tile < ntiles; tile += tile_step) {
        const bool live = row_live(tile);
        const unsigned row0 = (unsigned)(tile * 32);
        bf16x8 xb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xb[s] = xn[s];
        load_x(tile + tile_step, xn);
        f32x16 acc[2];
        bf16x8 hb[4];
        if constexpr (PAIR) {      // the companion head: hidden layer, <= 8 logits, softmax (as mlp_fwd_fast<2, 2, false>)
            f32x16 o2;
            hidden_layer_pinned<4>(W0s2, b0s2, xb, r, h, acc);
            relu_pack(acc, hb);
            out_block_pinned(WLs2, bLs2, 0, hb, r, h, o2);
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * h + j < p.out2_dim) mx = fmaxf(mx, o2[j]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mxs = mx * LOG2E;
            float e[4], sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = (4 * h + j < p.out2_dim) ? __builtin_amdgcn_exp2f(fmaf(o2[j], LOG2E, -mxs)) : 0.0f;
                sum += e[j];
            }
            sum += __shfl_xor(sum, 32);
            const float inv2 = 1.0f / sum;
            const unsigned obase = live ? row0 * (unsigned)(p.out2_dim * 2) : BUF_OOB_ROW;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t y = (bf16_t)(e[j] * inv2);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, y), rs_o2, ooff2[j] + obase, 0, 0);
            }
        }
        hidden_layer_pinned<4>(W0s, b0s, xb, r, h, acc);
        relu_pack(acc, hb);
        if constexpr (SAVE0) tile64_store_buf(stg, rs_h0, row0 * (HID * 2), lane, r, h, acc);
        hidden_layer_pinned<4>(W1s, b1s, hb, r, h, acc);
        relu_pack(acc, hb);
        tile64_store_buf(stg, rs_h1, row0 * (HID * 2), lane, r, h, acc);
        // online softmax statistics over the OB blocks
        float mrun = -INFINITY, srun = 0.0f;
        f32x16 o, on;
        out_block_pinned(WLs, bLs, 0, hb, r, h, o);
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            if (ob + 1 < OB) out_block_pinned(WLs, bLs, ob + 1, hb, r, h, on);
            float bm = fmaxf(o[0], o[1]);
#pragma unroll
            for (int q = 2; q < 16; ++q) bm = fmaxf(bm, o[q]);
            const float mn = fmaxf(mrun, bm);
            const f32x2 l2 = {LOG2E, LOG2E}, nm2 = {-mn * LOG2E, -mn * LOG2E};
            f32x2 bs2 = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x2 t = f32x2{o[2 * q], o[2 * q + 1]} * l2 + nm2;
                bs2 += f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            }
            srun = srun * __builtin_amdgcn_exp2f((mrun - mn) * LOG2E) + (bs2[0] + bs2[1]);
            mrun = mn;
            if (ob + 1 < OB) o = on;
        }
        const float mo = __shfl_xor(mrun, 32), so = __shfl_xor(srun, 32);
        const float M_ = fmaxf(mrun, mo);
        const float S_ = srun * __builtin_amdgcn_exp2f((mrun - M_) * LOG2E) + so * __builtin_amdgcn_exp2f((mo - M_) * LOG2E);
        const u32x2 st2 = {__builtin_bit_cast(unsigned, M_ * LOG2E), __builtin_bit_cast(unsigned, 1.0f / S_)};
        __builtin_amdgcn_raw_buffer_store_b64(st2, rs_st, live ? soff + row0 * 8 : BUF_OOB, 0, 0);
    }
}

// ----------------------------------------------------------------------------------------- backward
template <typename OutT, typename DxT, int NL, int OBMAX>
// 2 waves per SIMD: without the bound the 3-layer narrow variants took 252 VGPRs + 36 AGPRs (1 wave per SIMD); asking for 2 makes them fit 254 with no
// scratch.  The 33..64-output variants (OBMAX 2) would spill 150 - 350 B and stay at 1.
#ifndef PAG_BWD_WAVES
#define PAG_BWD_WAVES 2
#endif
__global__ __launch_bounds__(256, (OBMAX == 2 ? 1 : PAG_BWD_WAVES)) void mlp_bwd_mfma(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OB = (p.out_dim + 31) / 32;
    const int RSL = OB * 32 + 8;
    bf16_t *WLt = reinterpret_cast<bf16_t *>(smem);              // [64 in][RSL]  k = output channel (permuted)
    bf16_t *W1t = WLt + 64 * RSL;                                // [64][RS]      (NL == 3)
    bf16_t *W0t = W1t + (NL == 3 ? 64 * RS : 0);                 // [64 in-feature rows][RS]
    stage_weight_t(WLt, RSL, 64, OB * 32, p.W[NL - 1], p.out_dim, HID);
    if (NL == 3) stage_weight_t(W1t, RS, 64, 64, p.W[1], HID, HID);
    if (p.dx1) stage_weight_t(W0t, RS, 64, 64, p.W[0], HID, p.in_dim, p.grp_L, p.grp_F);
    bf16_t *stg = W0t + 64 * RS + (threadIdx.x >> 6) * (ST_BYTES / 2);      // wave-private staging tile
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (p.M + 31) / 32;
    const bool vec_out = (p.out_dim % 4) == 0;
    const OutT *outp = reinterpret_cast<const OutT *>(p.out);

    typedef typename RawVec<OutT>::type RawO;
    constexpr bool PREFETCH = OBMAX <= 2;          // registers allow keeping the next tile's gradients in flight
    auto load_g = [&](int64_t tile, RawO (&rz)[OBMAX][4], RawO (&ry)[OBMAX][4]) {
        const int64_t m = tile * 32 + r;
        const int64_t mc = (tile < ntiles && m < p.M) ? m : p.M - 1;
        const OutT *gop = reinterpret_cast<const OutT *>(p.grad_out) + mc * p.out_dim;
        if (!p.g_ray) {
#pragma unroll
            for (int ob = 0; ob < OBMAX; ++ob)
                if (ob < OB) load_block_raw(gop, 32 * ob, h, rz[ob], p.out_dim, vec_out);
        }
        if (p.act != PAG_ACT_NONE) {
#pragma unroll
            for (int ob = 0; ob < OBMAX; ++ob)
                if (ob < OB) load_block_raw(outp + mc * p.out_dim, 32 * ob, h, ry[ob], p.out_dim, vec_out);
        }
    };
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    RawO nz[PREFETCH ? OBMAX : 1][4], ny[PREFETCH ? OBMAX : 1][4];
    if constexpr (PREFETCH) load_g((int64_t)blockIdx.x * 4 + wave, nz, ny);
    bf16x8 hnext[PREFETCH ? NL - 1 : 1][4];      // next tile's hidden rows, in flight during the current tile
    auto fetch_h = [&](int64_t tile, bf16x8 (&hn)[PREFETCH ? NL - 1 : 1][4]) __attribute__((always_inline)) {
        if (tile < ntiles) {
            const int rv = (int)min((int64_t)32, p.M - tile * 32);
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
                tile64_fetch(reinterpret_cast<const bf16_t *>(p.hsave[l]) + tile * 32 * HID, rv, lane, hn[l]);
        }
    };
    if constexpr (PREFETCH) fetch_h((int64_t)blockIdx.x * 4 + wave, hnext);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += tile_step) {
        if constexpr (OBMAX > 2) {      // keep lane-constant addresses from being hoisted out of the tile loop and spilled
            asm volatile("" : "+v"(r), "+v"(h));
        }
        const int64_t m = tile * 32 + r;
        const bool live = m < p.M;
        const int64_t mc = live ? m : p.M - 1;
        // ReLU masks of the hidden layers: requested first, consumed after the first MFMA chain
        const int rows_valid = (int)min((int64_t)32, p.M - tile * 32);
        bf16x4 hraw[NL - 1][2][4];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int l = 0; l < NL - 1; ++l) tile64_unstage(stg, hnext[l], lane, r, h, hraw[l]);
            fetch_h(tile + tile_step, hnext);
        } else {
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
                tile64_load(stg, reinterpret_cast<const bf16_t *>(p.hsave[l]) + tile * 32 * HID, rows_valid, lane, r, h, hraw[l]);
        }
        // dx1_accumulate: the other decoder's gradient pieces are requested now and added at the end of the tile
        bf16x4 oldx[2][4];
        if (p.dx1_acc) {
            const bf16_t *dg = reinterpret_cast<const bf16_t *>(p.dx1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    oldx[mb][g] = *reinterpret_cast<const bf16x4 *>(dg + ((int64_t)(4 * mb + g) * p.M + mc) * 8 + 4 * h);
        }
        // ---- dz of the output layer -> bf16 B fragments (k-steps of the first backward MFMA chain W_L^T . dz_L)
        bf16x8 zb[OBMAX > 2 ? 2 : 2 * OBMAX];          // wide heads consume each block's two fragments immediately
        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
        const bool tile_full = (tile + 1) * 32 <= p.M;      // wave-uniform: only the last tile has dead lanes
        auto finish_block = [&](int ob, f32x16 &zz) __attribute__((always_inline)) {      // zero dead lanes, store dz, pack
            if (!tile_full) {
#pragma unroll
                for (int q = 0; q < 16; ++q) zz[q] = live ? zz[q] : 0.0f;
            }
            const int zi = OBMAX > 2 ? 0 : 2 * ob;
            pack_block(zz, zb[zi], zb[zi + 1]);
            if (OBMAX > 2 && (p.out_dim & 7) == 0)
                block32_store(stg, reinterpret_cast<bf16_t *>(p.dz[NL - 1]) + tile * 32 * p.out_dim, p.out_dim, 32 * ob, rows_valid, lane, r, h, zz);
            else if (live)
                store_block(reinterpret_cast<bf16_t *>(p.dz[NL - 1]) + m * p.out_dim, 32 * ob, h, zz, p.out_dim, vec_out);
            if constexpr (OBMAX > 2) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        bf16x8 a = *reinterpret_cast<const bf16x8 *>(WLt + (32 * mb + r) * RSL + 16 * (2 * ob + half) + 8 * h);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zb[half], acc[mb], 0, 0, 0);
                    }
            }
        };
        if constexpr (PREFETCH) {
            f32x16 z[OBMAX];
            RawO rz[OBMAX][4], ry[OBMAX][4];
#pragma unroll
            for (int ob = 0; ob < OBMAX; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    rz[ob][g] = nz[ob][g];
                    ry[ob][g] = ny[ob][g];
                }
            load_g(tile + tile_step, nz, ny);
#pragma unroll
            for (int ob = 0; ob < OBMAX; ++ob)
                if (ob < OB) raw_to_block(rz[ob], z[ob]);
            if (p.g_ray) {
                const int gi = p.g_index[mc], gi0 = __builtin_amdgcn_readfirstlane(gi);
                const float sc = p.g_ray_scale ? __fmul_rn(p.g_scale[mc], p.g_ray_scale[gi]) : p.g_scale[mc];
                if (__all(gi == gi0)) {
                    const float *g_row_u = p.g_ray + (int64_t)gi0 * p.out_dim;
#pragma unroll
                    for (int ob = 0; ob < OBMAX; ++ob)
                        if (ob < OB) rank1_block_uniform(g_row_u, sc, 32 * ob, h, p.out_dim, z[ob]);
                } else {
                    const float *g_row = p.g_ray + (int64_t)gi * p.out_dim;
#pragma unroll
                    for (int ob = 0; ob < OBMAX; ++ob)
                        if (ob < OB) rank1_block(g_row, sc, 32 * ob, h, p.out_dim, z[ob]);
                }
            }
            if (p.act == PAG_ACT_SIGMOID) {
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float y = (float)ry[ob][q >> 2][q & 3];
                            z[ob][q] = z[ob][q] * y * (1.0f - y);
                        }
            } else if (p.act == PAG_ACT_SOFTMAX) {
                float dot = 0.0f;
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q) dot += z[ob][q] * (float)ry[ob][q >> 2][q & 3];
                dot += __shfl_xor(dot, 32);
#pragma unroll
                for (int ob = 0; ob < OBMAX; ++ob)
                    if (ob < OB)
#pragma unroll
                        for (int q = 0; q < 16; ++q) z[ob][q] = (float)ry[ob][q >> 2][q & 3] * (z[ob][q] - dot);
            }
#pragma unroll
            for (int ob = 0; ob < OBMAX; ++ob) {
                if (ob < OB) {
                    finish_block(ob, z[ob]);
                } else {
                    zb[2 * ob] = zero8();
                    zb[2 * ob + 1] = zero8();
                }
            }
        } else {
            // wide heads: two passes over the gradient / probability rows (dot product for the softmax backward, then dz)
            // keep only ONE 32-channel block live at a time instead of all of them (256 VGPRs -> 1 wave per SIMD before).
            // The passes form one software pipeline over 2*OB steps: the global loads of step s+1 (16 bytes per lane,
            // fully coalesced) are in flight while step s goes through the LDS staging tile and the MFMAs - profiling
            // showed this kernel parked on memory 69 % of the time with one exposed round trip per block.
            const int g_idx1 = p.g_ray ? p.g_index[mc] : 0;
            const int g_idx0 = __builtin_amdgcn_readfirstlane(g_idx1);
            const bool g_uni = __all(g_idx1 == g_idx0);      // whole tile inside one ray (the common case)
            const float *g_row1 = p.g_ray ? p.g_ray + (int64_t)g_idx1 * p.out_dim : nullptr;
            const float *g_row_u = p.g_ray ? p.g_ray + (int64_t)g_idx0 * p.out_dim : nullptr;
            const float g_sc1 = p.g_ray ? (p.g_ray_scale ? __fmul_rn(p.g_scale[mc], p.g_ray_scale[g_idx1]) : p.g_scale[mc]) : 0.0f;
            const bool r1 = p.g_ray != nullptr;
            const bool need_y = p.act != PAG_ACT_NONE;
            const int n1 = p.act == PAG_ACT_SOFTMAX ? OB : 0, n_steps = n1 + OB;
            float dot = 0.0f;
            bool fast = false;
            if constexpr (sizeof(OutT) == 2) fast = (p.out_dim & 7) == 0;
            if (fast) {
                const bf16_t *gtile_g = reinterpret_cast<const bf16_t *>(p.grad_out) + tile * 32 * p.out_dim;
                const bf16_t *gtile_y = reinterpret_cast<const bf16_t *>(p.out) + tile * 32 * p.out_dim;
                bf16x8 cz[2], cy[2], nzr[2], nyr[2];
                auto gfetch = [&](int ob, bf16x8 (&fz)[2], bf16x8 (&fy)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 16 + (lane >> 2), c = 32 * ob + (lane & 3) * 8;
                        const bool ok = row < rows_valid && c < p.out_dim;
                        fz[it] = (ok && !r1) ? *reinterpret_cast<const bf16x8 *>(gtile_g + (int64_t)row * p.out_dim + c) : zero8();
                        fy[it] = (ok && need_y) ? *reinterpret_cast<const bf16x8 *>(gtile_y + (int64_t)row * p.out_dim + c) : zero8();
                    }
                };
                gfetch(0, cz, cy);
                for (int step = 0; step < n_steps; ++step) {
                    const int ob = step < n1 ? step : step - n1;
                    if (step + 1 < n_steps) gfetch(step + 1 < n1 ? step + 1 : step + 1 - n1, nzr, nyr);
                    bf16x4 rz1[4], ry1[4];
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 16 + (lane >> 2), c = (lane & 3) * 8;
                        *reinterpret_cast<bf16x8 *>(stg + row * ST_RS + c) = cz[it];
                        *reinterpret_cast<bf16x8 *>(stg + row * ST_RS + 32 + c) = cy[it];
                    }
                    wave_lds_sync();
                    block32_read(stg, 0, r, h, rz1);
                    block32_read(stg, 32, r, h, ry1);
                    wave_lds_sync();
                    f32x16 zz;
                    if (r1 && g_uni) rank1_block_uniform(g_row_u, g_sc1, 32 * ob, h, p.out_dim, zz);
                    else if (r1) rank1_block(g_row1, g_sc1, 32 * ob, h, p.out_dim, zz);
                    else raw_to_block(rz1, zz);
                    if (step < n1) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) dot += zz[q] * (float)ry1[q >> 2][q & 3];
                        if (step == n1 - 1) dot += __shfl_xor(dot, 32);
                    } else {
                        if (p.act == PAG_ACT_SIGMOID) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                const float y = (float)ry1[q >> 2][q & 3];
                                zz[q] = zz[q] * y * (1.0f - y);
                            }
                        } else if (p.act == PAG_ACT_SOFTMAX) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) zz[q] = (float)ry1[q >> 2][q & 3] * (zz[q] - dot);
                        }
                        finish_block(ob, zz);
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        cz[it] = nzr[it];
                        cy[it] = nyr[it];
                    }
                }
            } else {      // generic wide path (fp32 outputs or out_dim % 8 != 0): direct per-lane row accesses
                const OutT *gop = reinterpret_cast<const OutT *>(p.grad_out) + mc * p.out_dim;
                const OutT *yop = outp + mc * p.out_dim;
                for (int step = 0; step < n_steps; ++step) {
                    const int ob = step < n1 ? step : step - n1;
                    RawO rz1[4], ry1[4];
                    if (!r1) load_block_raw(gop, 32 * ob, h, rz1, p.out_dim, vec_out);
                    if (need_y) load_block_raw(yop, 32 * ob, h, ry1, p.out_dim, vec_out);
                    f32x16 zz;
                    if (r1) rank1_block(g_row1, g_sc1, 32 * ob, h, p.out_dim, zz);
                    else raw_to_block(rz1, zz);
                    if (step < n1) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) dot += zz[q] * (float)ry1[q >> 2][q & 3];
                        if (step == n1 - 1) dot += __shfl_xor(dot, 32);
                    } else {
                        if (p.act == PAG_ACT_SIGMOID) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                const float y = (float)ry1[q >> 2][q & 3];
                                zz[q] = zz[q] * y * (1.0f - y);
                            }
                        } else if (p.act == PAG_ACT_SOFTMAX) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) zz[q] = (float)ry1[q >> 2][q & 3] * (zz[q] - dot);
                        }
                        finish_block(ob, zz);
                    }
                }
            }
        }
        // ---- back through the output layer: dA = W_L^T . dz_L (already accumulated per block for wide heads),
        //      masked by the saved ReLU output
        bf16x8 hb[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if constexpr (OBMAX <= 2) {
#pragma unroll
                for (int s = 0; s < 2 * OBMAX; ++s) {
                    if (s < 2 * OB) {
                        bf16x8 a = *reinterpret_cast<const bf16x8 *>(WLt + (32 * mb + r) * RSL + 16 * s + 8 * h);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zb[s], acc[mb], 0, 0, 0);
                    }
                }
            }
            f32x16 hv;
            raw_to_block(hraw[NL - 2][mb], hv);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = (hv[q] > 0.0f && live) ? acc[mb][q] : 0.0f;
            pack_block(acc[mb], hb[2 * mb], hb[2 * mb + 1]);
        }
        tile64_store(stg, reinterpret_cast<bf16_t *>(p.dz[NL - 2]) + tile * 32 * HID, rows_valid, lane, r, h, acc);
        if (NL == 3) {
            bf16x8 hb2[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 a = *reinterpret_cast<const bf16x8 *>(W1t + (32 * mb + r) * RS + 16 * s + 8 * h);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[s], acc[mb], 0, 0, 0);
                }
                f32x16 hv;
                raw_to_block(hraw[0][mb], hv);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][q] = (hv[q] > 0.0f && live) ? acc[mb][q] : 0.0f;
                pack_block(acc[mb], hb2[2 * mb], hb2[2 * mb + 1]);
            }
            tile64_store(stg, reinterpret_cast<bf16_t *>(p.dz[0]) + tile * 32 * HID, rows_valid, lane, r, h, acc);
#pragma unroll
            for (int s = 0; s < 4; ++s) hb[s] = hb2[s];
        }
        // ---- dx1 = (W_0^T . dz_0)[0:k1]
        if (p.dx1) {
            DxT *dx = reinterpret_cast<DxT *>(p.dx1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (32 * mb < p.k1) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        bf16x8 a = *reinterpret_cast<const bf16x8 *>(W0t + (32 * mb + r) * RS + 16 * s + 8 * h);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[s], acc[mb], 0, 0, 0);
                    }
                    if (live && !p.grp_L) {
                        // extra gradient of input column 0 (the density read off the density decoder's first output,
                        // pc_nerf/panoptic_delta_nef.py:188): added here instead of a zero-padded [M,k1] tensor + add pass
                        if (p.dx_col0 && mb == 0 && h == 0 && (!p.dx_col0_gate || p.dx_col0_gate[m] > 0.0f)) acc[0][0] += p.dx_col0[m];
                        store_block(dx + m * p.k1, 32 * mb, h, acc[mb], p.k1, true);
                    }
                    if (live && p.grp_L) {      // XCD8: row 32mb + 8g + 4h + j of dx^T -> piece [4mb + g][m][4h + j]
                        bf16_t *dg = reinterpret_cast<bf16_t *>(p.dx1);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 *dst = reinterpret_cast<bf16x4 *>(dg + ((int64_t)(4 * mb + g) * p.M + m) * 8 + 4 * h);
                            if (p.dx1_acc) {      // second decoder on the same input: add to the first one's gradient in place
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[mb][4 * g + j] += (float)oldx[mb][g][j];
                            }
                            bf16x4 v = {(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
                            *dst = v;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------ backward-data + weight gradients, narrow decoders (one launch)
// The separate weight-gradient kernel re-read every dz and every layer input ([M,64] bf16 each: 536 MB per 64 x 64 layer at
// M = 2.1 M) at 4.7 TB/s - the practical HBM rate, so only removing the bytes helps.  Here every wave keeps dW of ALL layers
// as MFMA accumulators for the whole launch (6 - 10 blocks of 16 registers + one block for the biases; one wave per SIMD,
// 512 registers) and feeds them from its own 32-sample tiles: dz (just computed) and the layer inputs (the saved activations it
// needs anyway for the ReLU masks + the layer-0 input) live as swizzled LDS images and are read back TRANSPOSED
// (ds_read_b64_tr_b16: the sum runs over the samples).  No shared tiles, no block barriers, no dz tensors in memory.
// The three decoder shapes of the model are separate instantiations so that the tile loop is straight-line code: every
// global load of tile t+1 is issued unconditionally (clamped addresses) at the top of tile t - with one wave per SIMD an
// exposed round trip costs a whole tile, and branches make the compiler drain vmcnt to 0 at each join.
//   KIND 0  density-like : XCD8 bf16 input, dense bf16 gradient (out_dim % 4 == 0, <= 32), no activation, XCD8 bf16 dx
//   KIND 1  colour-like  : bf16 x1 [M,16] + per-ray f32 x2 [R,32], dense f32 gradient, sigmoid (out_dim <= 4), bf16 dx [M,16] with
//                          the gated column-0 addend (the density read off x1[:,0])
//   KIND 2  semantic-like: XCD8 bf16 input, rank-1 gradient, softmax with saved bf16 probabilities (out_dim <= 8), XCD8 bf16 dx
//                          (DXACC: added to the other head's gradient in place)
// Full 32-sample tiles run in the main loop; a ragged last tile runs once after it with predicated stores.
// DZ0 = 1: also write dz_0 (the gradient at the first hidden layer's pre-activation) as a bf16 [M,64] tensor (p.dz[0]) - what the caller sums per ray
// and multiplies by W_0[:, k1:] for the gradient of the per-ray x2 (the view embedding: pose optimisation, pc_nerf/ba_pipeline.py:89-90).
// DZ0 = 2 (x2_index non-decreasing: samples packed ray by ray): the per-ray sum starts HERE - the tile's dz_0 rows are summed per ray on the matrix
// cores (dz_0^T . S with S[sample][j] = 1 where the sample belongs to the tile's j-th ray: the A fragments are the ones layer 0's weight gradient
// reads anyway) and one 64-float row per (tile, ray) leaves the kernel (p.dz0_slots, row = tile + ray: strictly increasing along the samples, so
// every pair owns a row): 8 B per sample on 512-sample rays instead of 128 written and 128 read again by the per-ray sum.
#ifdef PAG_FUSED_PROF      // debug build: where a workgroup of mlp_bwd_fused spends its clocks outside the tile loop (printed per launch by pag_mlp_bwd)
__device__ unsigned long long g_fused_prof[8];      // shader clocks summed over workgroups: [0] staging, [1] tile loop, [2] cross-wave sum, [4] workgroups
#endif
// Experiment switches (scripts/build_variant.sh; never set in the shipped build): PAG_EXP_NO_WGRAD drops the weight-gradient MFMAs (wrong
// results, timing only: what the backward-data chain alone costs), PAG_EXP_FUSED_WAVES = minimum waves per SIMD asked of the compiler and
// workgroups per CU launched - together they measure the producer half of a producer / consumer split at two waves per SIMD.
#ifndef PAG_EXP_FUSED_WAVES
#define PAG_EXP_FUSED_WAVES 1
#endif
template <int NL, int KIND, bool DXACC, int OBL = 1 /* 32-row blocks of the output layer; 2 only with KIND 0 */, int DZ0 = 0 /* 1: dz_0 rows, 2: per-(tile, ray) sums */>
__global__ __launch_bounds__(256, PAG_EXP_FUSED_WAVES) void mlp_bwd_fused(BwdParams p) {
    PAG_BLOCK_TIMER(2);
#ifdef PAG_FUSED_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    static_assert(OBL == 1 || KIND == 0, "a 64-wide output layer exists for the dense-gradient form only");
    constexpr bool GRP = KIND != 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RSL = OBL * 32 + 8;
    bf16_t *WLt = reinterpret_cast<bf16_t *>(smem);              // [64 in][RSL]  k = output channel (permuted)
    bf16_t *W1t = WLt + 64 * RSL;                                // [64][RS]      (NL == 3)
    bf16_t *W0t = W1t + (NL == 3 ? 64 * RS : 0);                 // [64 in-feature rows][RS]
    stage_weight_t(WLt, RSL, 64, OBL * 32, p.W[NL - 1], p.out_dim, HID);
    // (W_1 and W_0 are staged below, both images from one pass each)
    // the forward's own images of the hidden layers: their activations are RECOMPUTED here (8 - 16 MFMAs per tile on idle matrix
    // cores, same fragments and instruction sequence as mlp_fwd_mfma: bit-identical) instead of being written by the forward and
    // read back - 268 MB each way per hidden layer at M = 2.1 M, in kernels that run at the HBM rate
    bf16_t *W0s = W0t + 64 * RS;                                 // [64][RS] natural k
    bf16_t *W1s = W0s + 64 * RS;                                 // [64][RS] permuted k (NL == 3)
    float *b0s = reinterpret_cast<float *>(W1s + (NL == 3 ? 64 * RS : 0));
    float *b1s = b0s + 64;
    stage_weight_both(W0s, RS, false, W0t, RS, 64, 64, p.W[0], HID, p.in_dim, p.grp_L, p.grp_F);
    if (NL == 3) stage_weight_both(W1s, RS, true, W1t, RS, 64, 64, p.W[1], HID, HID);
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0s[e] = p.b[0][e];
        b1s[e] = NL == 3 ? p.b[1][e] : 0.0f;
    }
    bf16_t *Tx = reinterpret_cast<bf16_t *>(b1s + 64) + (threadIdx.x >> 6) * ((NL + 1) * TW_ELEMS);      // wave-private swizzled tiles
    bf16_t *Th0 = Tx + TW_ELEMS, *Th1 = Th0 + (NL == 3 ? TW_ELEMS : 0), *Tz = Th1 + TW_ELEMS;
    __syncthreads();
#ifdef PAG_FUSED_PROF
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t M = p.M, ntiles = (M + 31) / 32, nfull = M / 32;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    const bf16_t *x1b = reinterpret_cast<const bf16_t *>(p.x1);
    const int32_t *ridx = KIND == 1 ? p.x2_index : p.g_index;      // per-sample row of the per-ray tables
    const auto rs_dz0 = __builtin_amdgcn_make_buffer_rsrc(DZ0 == 1 ? p.dz[0] : nullptr, 0, (int)(DZ0 == 1 ? M * HID * 2 : 0), 0x00020000);
    // the masked dz_0 of the tile sits in the Tz image (row = sample, 4-unit chunks in natural order): whole rows leave as 16-byte pieces;
    // rows past M fall outside the buffer's bounds and are dropped
    auto store_dz0 = [&](int64_t tile) __attribute__((always_inline)) {
        if constexpr (DZ0 == 1) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const unsigned voff = (unsigned)(tile * (32 * HID * 2)) + (unsigned)((lane >> 3) * (HID * 2) + (lane & 7) * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rowl = k * 8 + (lane >> 3), ch = (lane & 7) * 2;
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(Tz + tw_off(rowl, ch));
                const u32x2 hi = *reinterpret_cast<const u32x2 *>(Tz + tw_off(rowl, ch + 1));
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, rs_dz0, voff + k * (8 * HID * 2), 0, 0);
            }
        }
    };

    // DZ0 == 2: wave-private scratch behind the tiles - the local ray number of each of the tile's 32 samples and the ray of each local number
    int *lray_s = reinterpret_cast<int *>(reinterpret_cast<bf16_t *>(b1s + 64) + 4 * ((NL + 1) * TW_ELEMS)) + (threadIdx.x >> 6) * 64, *rid_s = lray_s + 32;
    int ray_c = 0;             // x2 row (= ray) of this lane's sample in the CURRENT tile (DZ0 == 2)
    unsigned heads_c = 0u;     // bit s: sample s of the current tile starts a new ray
    auto note_rays = [&](int ray_here) __attribute__((always_inline)) {      // called at the top of a tile, before its first wave_lds_sync
        if constexpr (DZ0 == 2) {
            const int prev = __shfl_up(ray_here, 1, 32);
            const bool head = r == 0 || prev != ray_here;
            heads_c = (unsigned)__ballot(head);                      // lanes 0..31 and 32..63 hold the same rows: the low word
            const int lr = __builtin_popcount(heads_c & (0xFFFFFFFFu >> (31 - r))) - 1;
            if (h == 0) {
                lray_s[r] = lr;
                if (head) rid_s[lr] = ray_here;
            }
        }
    };
    auto store_dz0_slots = [&](int64_t tile) __attribute__((always_inline)) {
        if constexpr (DZ0 == 2) {
            f32x16 d[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 l0 = *reinterpret_cast<const i32x4 *>(lray_s + 16 * ks + 8 * h), l1 = *reinterpret_cast<const i32x4 *>(lray_s + 16 * ks + 8 * h + 4);
                bf16x8 sf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sf[e] = (bf16_t)(l0[e] == r ? 1.0f : 0.0f);
                    sf[4 + e] = (bf16_t)(l1[e] == r ? 1.0f : 0.0f);
                }
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const bf16x8 afr = tw_frag(Tz, ob, ks, lane);
                    if (ks == 0) {
                        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        d[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, sf, zero, 0, 0, 0);
                    } else {
                        d[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, sf, d[ob], 0, 0, 0);
                    }
                }
            }
            if (r < __builtin_popcount(heads_c)) {      // lane (j, h): the tile's j-th ray, channels 32 ob + 8 g + 4 h .. + 3
                float *o = p.dz0_slots + ((int64_t)tile + rid_s[r]) * 64;
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4 *>(o + 32 * ob + 8 * g + 4 * h) = f32x4{d[ob][4 * g], d[ob][4 * g + 1], d[ob][4 * g + 2], d[ob][4 * g + 3]};
            }
        }
    };

    // ---- registers of the NEXT tile (requested one tile ahead; the per-ray row index two tiles ahead)
    bf16x8 xn[4];
    bf16x4 gz[OBL][KIND == 0 ? 4 : 1], oldx[DXACC ? 2 : 1][4];
    float gs[4], gy[4], gsc = 0.0f, c0a = 0.0f, c0g = 0.0f;      // KIND 1: gradient / output; KIND 2: g_ray row / probabilities
    int ray1 = 0, ray2 = 0;                                        // row index of the tile after next (and the one after that)
    auto row_of = [&](int64_t tile) __attribute__((always_inline)) { return min(min(tile, ntiles - 1) * 32 + r, M - 1); };
    auto prefetch = [&](int64_t tile_raw, int ray) __attribute__((always_inline)) {
        const int64_t tile = min(tile_raw, ntiles - 1);
        const int64_t m = min(tile * 32 + r, M - 1);
        if constexpr (GRP) {
#pragma unroll
            for (int s = 0; s < 4; ++s) xn[s] = load8(x1b + ((int64_t)(2 * s + h) * M + m) * 8);
        } else {      // k1 = 16, k2p = 32: x1 | x2[ray][0:16] | x2[ray][16:32] | 0
            xn[0] = load8(x1b + m * 16 + 8 * h);
            xn[1] = load8(p.x2 + (int64_t)ray * 32 + 8 * h);
            xn[2] = load8(p.x2 + (int64_t)ray * 32 + 16 + 8 * h);
            xn[3] = zero8();
        }
        if constexpr (KIND == 0) {
            const bf16_t *gop = reinterpret_cast<const bf16_t *>(p.grad_out) + m * p.out_dim;
#pragma unroll
            for (int ob = 0; ob < OBL; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * ob + 8 * g + 4 * h;
                    gz[ob][g] = *reinterpret_cast<const bf16x4 *>(gop + (c0 < p.out_dim ? c0 : 0));
                }
        } else if constexpr (KIND == 1) {
            const float *gop = reinterpret_cast<const float *>(p.grad_out) + m * p.out_dim;
            const float *yop = reinterpret_cast<const float *>(p.out) + m * p.out_dim;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = j < p.out_dim ? j : 0;
                gs[j] = gop[c];
                gy[j] = yop[c];
            }
            c0a = p.dx_col0[m];
            c0g = p.dx_col0_gate[m];
        } else {
            gsc = __fmul_rn(p.g_scale[m], p.g_ray_scale[ray]);
            const float *grow = p.g_ray + (int64_t)ray * p.out_dim;
            const bf16_t *yop = reinterpret_cast<const bf16_t *>(p.out) + m * p.out_dim;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (4 * h + j) < p.out_dim ? (4 * h + j) : 0;
                gs[j] = grow[c];
                gy[j] = (float)yop[c];
            }
        }
        if constexpr (DXACC) {
            const bf16_t *dg = reinterpret_cast<const bf16_t *>(p.dx1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    oldx[mb][g] = *reinterpret_cast<const bf16x4 *>(dg + ((int64_t)(4 * mb + g) * M + m) * 8 + 4 * h);
        }
    };

    // ---- weight-gradient accumulators (the whole launch) and their feeding
    f32x16 awL[OBL][2], awM[NL == 3 ? 2 : 1][2], aw0[2][2], dbacc;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        dbacc[q] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int ob = 0; ob < OBL; ++ob) awL[ob][i][q] = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                aw0[i][j][q] = 0.0f;
                if constexpr (NL == 3) awM[i][j][q] = 0.0f;
            }
        }
    }
    // B fragment "1 in column j": dbacc[:, j] += row sums of the A operand (bias gradients of the layers fed by hidden activations)
    auto ones_col = [&](int j) __attribute__((always_inline)) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(r == j ? 1.0f : 0.0f);
        return o;
    };
    // dW[ob][ib] += dz(Tz block ob)^T . input(Tin block ib) over this tile's 32 samples; dbcol >= 0: bias gradients into dbacc
    auto wgrad_tile = [&](const bf16_t *Tin, auto &aw, int dbcol) __attribute__((always_inline)) {      // aw: f32x16 [out blocks][2]
        constexpr int NOB = (int)(sizeof(aw) / sizeof(aw[0]));
#ifdef PAG_EXP_NO_WGRAD
        return;
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[2], afr[NOB];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) bfr[ib] = tw_frag(Tin, ib, ks, lane);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) afr[ob] = tw_frag(Tz, ob, ks, lane);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) aw[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ob], bfr[ib], aw[ob][ib], 0, 0, 0);
                if (dbcol >= 0) dbacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ob], ones_col(dbcol + ob), dbacc, 0, 0, 0);
            }
        }
    };

    // dx of tile t is STORED at the top of tile t+1 (after that tile's prefetched registers have been consumed, before the next
    // prefetch is issued): vmcnt counts loads and stores in issue order, so stores issued at the end of a tile would have to drain
    // before the next tile could touch its prefetched data - a full write round trip per tile with one wave per SIMD.  The very
    // first flush writes zeros to the wave's first tile; the real values follow from the same wave (stores of one wave stay ordered).
    constexpr int NPB = GRP ? 2 : 1, NPG = GRP ? 4 : 2;
    bf16x4 pend[NPB][NPG];
#pragma unroll
    for (int mb = 0; mb < NPB; ++mb)
#pragma unroll
        for (int g = 0; g < NPG; ++g) pend[mb][g] = bf16x4{(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
    int64_t pend_m = min((int64_t)blockIdx.x * 4 * 32 + (int64_t)wave * 32 + r, M - 1);
    auto flush = [&](bool pred) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < NPB; ++mb)
#pragma unroll
            for (int g = 0; g < NPG; ++g) {
                bf16_t *dst = GRP ? reinterpret_cast<bf16_t *>(p.dx1) + ((int64_t)(4 * mb + g) * M + pend_m) * 8 + 4 * h
                                  : reinterpret_cast<bf16_t *>(p.dx1) + pend_m * 16 + 8 * g + 4 * h;
                if (pred) *reinterpret_cast<bf16x4 *>(dst) = pend[mb][g];
            }
    };
    auto body = [&](int64_t tile, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t m = tile * 32 + r;
        const bool live = FULL || m < M;
        // ---- this tile's layer-0 input and the hidden activations recomputed from it become LDS images (kept for the whole tile)
        wave_lds_sync();
        note_rays(ray_c);
        {
            bf16x8 xb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) xb[s] = xn[s];
            if (h == 1) xb[3][7] = (bf16_t)1.0f;      // column 63 := 1 (its weight is 0 in the forward): its dW column is the layer-0 bias gradient
            tw_put_frags(Tx, xb, r, h);
            f32x16 ha[2];
            bf16x8 hf[4];
            hidden_layer_pinned<4>(W0s, b0s, xb, r, h, ha);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) ha[mb][q] = fmaxf(ha[mb][q], 0.0f);
                tw_put_block(Th0, mb, r, h, ha[mb]);
                if constexpr (NL == 3) pack_block(ha[mb], hf[2 * mb], hf[2 * mb + 1]);
            }
            if constexpr (NL == 3) {
                hidden_layer_pinned<4>(W1s, b1s, hf, r, h, ha);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) ha[mb][q] = fmaxf(ha[mb][q], 0.0f);
                    tw_put_block(Th1, mb, r, h, ha[mb]);
                }
            }
        }
        // ---- upstream gradient of this tile -> dz of the output layer (OBL 32-row blocks)
        f32x16 zv[OBL];
#pragma unroll
        for (int ob = 0; ob < OBL; ++ob)
#pragma unroll
            for (int q = 0; q < 16; ++q) zv[ob][q] = 0.0f;
        f32x16 &z = zv[0];
        if constexpr (KIND == 0) {
#pragma unroll
            for (int ob = 0; ob < OBL; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool ok = 32 * ob + 8 * g + 4 * h < p.out_dim;
#pragma unroll
                    for (int j = 0; j < 4; ++j) zv[ob][4 * g + j] = ok ? (float)gz[ob][g][j] : 0.0f;
                }
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = gy[j];
                z[j] = (h == 0 && j < p.out_dim) ? gs[j] * y * (1.0f - y) : 0.0f;
            }
        } else {
            float zr[4], yv[4], dot = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = 4 * h + j < p.out_dim;
                zr[j] = ok ? gsc * gs[j] : 0.0f;
                yv[j] = ok ? gy[j] : 0.0f;
                dot += zr[j] * yv[j];
            }
            dot += __shfl_xor(dot, 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = yv[j] * (zr[j] - dot);
        }
        const float col0 = (KIND == 1 && h == 0 && c0g > 0.0f) ? c0a : 0.0f;
        bf16x4 ox[DXACC ? 2 : 1][4];
        if constexpr (DXACC) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) ox[mb][g] = oldx[mb][g];
        }
        // ---- the previous tile's dx goes out, then everything of the next tile is requested (its row index was loaded one tile earlier)
        flush(true);
        prefetch(tile + tile_step, ray1);
        ray_c = ray1;              // (the next tile's; this tile's local ray numbers are in LDS by now)
        ray1 = ray2;
        if constexpr (KIND != 0) ray2 = ridx[row_of(tile + 3 * tile_step)];
        bf16x8 zb[2 * OBL];
#pragma unroll
        for (int ob = 0; ob < OBL; ++ob) {
            if constexpr (!FULL) {
#pragma unroll
                for (int q = 0; q < 16; ++q) zv[ob][q] = live ? zv[ob][q] : 0.0f;
            }
            pack_block(zv[ob], zb[2 * ob], zb[2 * ob + 1]);
            tw_put_block(Tz, ob, r, h, zv[ob]);
        }
        wave_lds_sync();
        // ---- back through the output layer, masked by the saved ReLU output (read from its LDS image).  Every W^T chain (fragments in
        // registers, weights from LDS) is issued BEFORE the weight-gradient MFMAs of the same dz: their transposed reads of the image just
        // written then land under the chain instead of being waited for by the only wave of the SIMD.
        f32x16 acc[2];
        bf16x8 hb[4];
        wt_chain_pinned<2, 2 * OBL>(WLt, RSL, zb, r, h, acc);
        wgrad_tile(NL == 3 ? Th1 : Th0, awL, 0);
        wave_lds_sync();
        relu_mask_pack_put(NL == 3 ? Th1 : Th0, Tz, r, h, live, acc, hb);
        wave_lds_sync();
        if constexpr (NL == 3) {
            bf16x8 hb2[4];
            wt_chain_pinned<2, 4>(W1t, RS, hb, r, h, acc);
            wgrad_tile(Th0, awM, OBL);
            wave_lds_sync();
            relu_mask_pack_put(Th0, Tz, r, h, live, acc, hb2);
            wave_lds_sync();
#pragma unroll
            for (int s = 0; s < 4; ++s) hb[s] = hb2[s];
        }
        store_dz0(tile);
        store_dz0_slots(tile);
        // ---- dx1 = (W_0^T . dz_0)[0:k1], then layer 0's weight gradient
        wt_chain_pinned<(GRP ? 2 : 1), 4>(W0t, RS, hb, r, h, acc);
        wgrad_tile(Tx, aw0, -1);
#pragma unroll
        for (int mb = 0; mb < (GRP ? 2 : 1); ++mb) {
            if constexpr (GRP) {      // XCD8: row 32mb + 8g + 4h + j of dx^T -> piece [4mb + g][m][4h + j]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (DXACC) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[mb][4 * g + j] += (float)ox[mb][g][j];
                    }
                    pend[mb][g] = bf16x4{(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
                }
            } else {                  // [M,16] bf16: columns 8g + 4h + j, g < 2; the density gradient joins column 0
                acc[0][0] += col0;
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    pend[0][g] = bf16x4{(bf16_t)acc[0][4 * g], (bf16_t)acc[0][4 * g + 1], (bf16_t)acc[0][4 * g + 2], (bf16_t)acc[0][4 * g + 3]};
            }
        }
        pend_m = min(m, M - 1);
    };

    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    int ray0 = 0;
    if constexpr (KIND != 0) {
        ray0 = ridx[row_of(tile)];
        ray1 = ridx[row_of(tile + tile_step)];
        ray2 = ridx[row_of(tile + 2 * tile_step)];
    }
    prefetch(tile, ray0);
    ray_c = ray0;
    const bool any_tile = tile < ntiles;
    bool last_ragged = false;
    for (; tile < nfull; tile += tile_step) body(tile, std::true_type{});
    if (tile < ntiles) {      // the ragged last tile (one wave of the launch)
        body(tile, std::false_type{});
        last_ragged = true;
    }
    if (any_tile) flush(!last_ragged || (nfull * 32 + r) < M);

    // ---- the four waves' accumulators are summed through LDS (tiles and weights are dead), then ONE slab per workgroup and layer:
    //      [rows_pad][96] (cols 0..63 dW - staged positions for XCD8 inputs -, col 64 db), summed over workgroups by wgrad_finish_kernel
    constexpr int BM = 2 * OBL, B0 = BM + (NL == 3 ? 4 : 0), NBLK = B0 + 4 + 1;      // blocks: last layer | middle layer | layer 0 | db
    auto blk = [&](auto bi) -> const f32x16 & {
        constexpr int bb = decltype(bi)::value;
        if constexpr (bb < BM) return awL[bb >> 1][bb & 1];
        else if constexpr (bb < B0) return awM[(bb - BM) >> 1][(bb - BM) & 1];
        else if constexpr (bb < NBLK - 1) return aw0[(bb - B0) >> 1][(bb - B0) & 1];
        else return dbacc;
    };
    __syncthreads();
#ifdef PAG_FUSED_PROF
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    float *red = reinterpret_cast<float *>(smem);                      // [NBLK][4][64 lanes] x 4 floats
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            static_for<NBLK>([&](auto bi) {
                constexpr int bb = decltype(bi)::value;
                const f32x16 &a = blk(bi);
                // four values per LDS access (the sum was 2 x 176 four-byte accesses per lane and wave: 42 k clocks of a launch); a block's four
                // reads are issued together, then its four writes (one select per access made each a dependent round trip)
                f32x4 *dst = reinterpret_cast<f32x4 *>(red) + (bb * 4) * 64 + lane;
                f32x4 v[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) v[q4] = f32x4{a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
                if (w != 0) {
                    f32x4 t[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) t[q4] = dst[q4 * 64];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) v[q4] = t[q4] + v[q4];
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) dst[q4 * 64] = v[q4];
            });
        }
        __syncthreads();
    }
#ifdef PAG_FUSED_PROF
    if (threadIdx.x == 0) {
        const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_fused_prof[0], pt1 - pt0);
        atomicAdd(&g_fused_prof[1], pt2 - pt1);
        atomicAdd(&g_fused_prof[2], pt3 - pt2);
        atomicAdd(&g_fused_prof[4], 1ull);
    }
#endif
    static_for<NBLK>([&](auto bi) {
        constexpr int bb = decltype(bi)::value;
        if ((bb & 3) != wave) return;
        f32x4 sv[4];      // this lane's 16 values of block bb
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) sv[q4] = (reinterpret_cast<const f32x4 *>(red) + (bb * 4 + q4) * 64 + lane)[0];
        auto srcv = [&](int q) __attribute__((always_inline)) { return sv[q >> 2][q & 3]; };
        if constexpr (bb < BM) {
            float *sl = p.slabs[NL - 1] + (int64_t)blockIdx.x * (OBL * 32) * WG_SLAB_COLS_F;
#pragma unroll
            for (int q = 0; q < 16; ++q) sl[(32 * (bb >> 1) + rho(q, h)) * WG_SLAB_COLS_F + 32 * (bb & 1) + r] = srcv(q);
        } else if constexpr (bb < B0) {
            float *sl = p.slabs[1] + (int64_t)blockIdx.x * 64 * WG_SLAB_COLS_F;
#pragma unroll
            for (int q = 0; q < 16; ++q) sl[(32 * ((bb - BM) >> 1) + rho(q, h)) * WG_SLAB_COLS_F + 32 * ((bb - BM) & 1) + r] = srcv(q);
        } else if constexpr (bb < NBLK - 1) {
            constexpr int ob = (bb - B0) >> 1, ib = (bb - B0) & 1;
            float *sl = p.slabs[0] + (int64_t)blockIdx.x * 64 * WG_SLAB_COLS_F;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = srcv(q);
                sl[(32 * ob + rho(q, h)) * WG_SLAB_COLS_F + 32 * ib + r] = v;
                if (ib == 1 && r == 31) sl[(32 * ob + rho(q, h)) * WG_SLAB_COLS_F + 64] = v;      // input column 63 (the ones column) = db of layer 0
            }
        } else {      // dbacc: column ob -> last layer's block ob; column OBL + ob -> middle layer's block ob
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = srcv(q);
                if (r < OBL) p.slabs[NL - 1][((int64_t)blockIdx.x * (OBL * 32) + 32 * r + rho(q, h)) * WG_SLAB_COLS_F + 64] = v;
                if (NL == 3 && r >= OBL && r < OBL + 2) p.slabs[1][((int64_t)blockIdx.x * 64 + 32 * (r - OBL) + rho(q, h)) * WG_SLAB_COLS_F + 64] = v;
            }
        }
    });
}

// ------------------------------------------------ two decoders on the same XCD8 input, backward in one launch (panoptic heads)
// The semantic head and the two layers below the instance head's output layer read the same panoptic features and their input
// gradients are summed.  As two launches (stage B of the instance head writes dx, the semantic head reads it back and adds) the
// pair moved 1.6 GB at 3.5 - 4.4 TB/s; here a wave runs both decoders on ITS tile - one read of the features, one write of the
// summed gradient (396 B per sample instead of 780) - with both decoders' weight gradients as accumulators (240 registers).
//   .i  two layers, dense bf16 upstream gradient [M,64] (the hidden gradient stage A wrote), 64-wide "output" layer, no activation
//   .s  two layers, rank-1 upstream gradient, softmax with saved bf16 probabilities, out_dim <= 8
// Same tile images, fragments and instruction sequences as mlp_bwd_fused<2, 0, false, 2> and <2, 2, *>: the weight gradients are
// bit-identical to the two launches, dx differs by one bf16 rounding less (the sum is formed in fp32).
struct PairParams {
    BwdParams i, s;
};
__global__ __launch_bounds__(256, 1) void mlp_bwd_pair(PairParams pp) {
    PAG_BLOCK_TIMER(3);
#ifdef PAG_FUSED_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    const BwdParams &pi = pp.i, &ps = pp.s;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RSLI = 72, RSLS = 40;
    bf16_t *WLtI = reinterpret_cast<bf16_t *>(smem);             // [64][72]  k = the 64 outputs of .i's upper layer (permuted)
    bf16_t *W0tI = WLtI + 64 * RSLI;                             // [64 input positions][RS]
    bf16_t *W0sI = W0tI + 64 * RS;                               // [64][RS] natural k (forward recompute)
    bf16_t *WLtS = W0sI + 64 * RS;                               // [64][40]
    bf16_t *W0tS = WLtS + 64 * RSLS;
    bf16_t *W0sS = W0tS + 64 * RS;
    float *b0I = reinterpret_cast<float *>(W0sS + 64 * RS);
    float *b0S = b0I + 64;
    stage_weight_t(WLtI, RSLI, 64, 64, pi.W[1], pi.out_dim, HID);
    stage_weight_both(W0sI, RS, false, W0tI, RS, 64, 64, pi.W[0], HID, pi.in_dim, pi.grp_L, pi.grp_F);
    stage_weight_t(WLtS, RSLS, 64, 32, ps.W[1], ps.out_dim, HID);
    stage_weight_both(W0sS, RS, false, W0tS, RS, 64, 64, ps.W[0], HID, ps.in_dim, ps.grp_L, ps.grp_F);
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0I[e] = pi.b[0][e];
        b0S[e] = ps.b[0][e];
    }
    bf16_t *Tx = reinterpret_cast<bf16_t *>(b0S + 64) + (threadIdx.x >> 6) * (4 * TW_ELEMS);      // wave-private swizzled tiles
    bf16_t *ThI = Tx + TW_ELEMS, *ThS = ThI + TW_ELEMS, *Tz = ThS + TW_ELEMS;
    __syncthreads();
#ifdef PAG_FUSED_PROF
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t M = pi.M, ntiles = (M + 31) / 32, nfull = M / 32;
    const int64_t tile_step = (int64_t)gridDim.x * 4;
    const bf16_t *x1b = reinterpret_cast<const bf16_t *>(pi.x1);
    const int32_t *ridx = ps.g_index;

    // ---- registers of the NEXT tile (requested one tile ahead; the per-ray row index two tiles ahead)
    bf16x8 xn[4];
    bf16x4 gz[2][4];
    float gs[4], gy[4], gsc = 0.0f;
    int ray1 = 0, ray2 = 0;
    auto row_of = [&](int64_t tile) __attribute__((always_inline)) { return min(min(tile, ntiles - 1) * 32 + r, M - 1); };
    auto prefetch = [&](int64_t tile_raw, int ray) __attribute__((always_inline)) {
        const int64_t tile = min(tile_raw, ntiles - 1);
        const int64_t m = min(tile * 32 + r, M - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) xn[s] = load8(x1b + ((int64_t)(2 * s + h) * M + m) * 8);
        const bf16_t *gop = reinterpret_cast<const bf16_t *>(pi.grad_out) + m * HID;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g) gz[ob][g] = *reinterpret_cast<const bf16x4 *>(gop + 32 * ob + 8 * g + 4 * h);
        gsc = __fmul_rn(ps.g_scale[m], ps.g_ray_scale[ray]);
        const float *grow = ps.g_ray + (int64_t)ray * ps.out_dim;
        const bf16_t *yop = reinterpret_cast<const bf16_t *>(ps.out) + m * ps.out_dim;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (4 * h + j) < ps.out_dim ? (4 * h + j) : 0;
            gs[j] = grow[c];
            gy[j] = (float)yop[c];
        }
    };

    // ---- weight-gradient accumulators (the whole launch): .i upper layer [64 x 64], .i layer 0, .s output layer [32 x 64], .s layer 0, biases
    f32x16 awLI[2][2], aw0I[2][2], awLS[1][2], aw0S[2][2], dbacc;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        dbacc[q] = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            awLS[0][i][q] = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) awLI[i][j][q] = aw0I[i][j][q] = aw0S[i][j][q] = 0.0f;
        }
    }
    auto ones_col = [&](int j) __attribute__((always_inline)) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(r == j ? 1.0f : 0.0f);
        return o;
    };
    auto wgrad_tile = [&](const bf16_t *Tin, auto &aw, int dbcol) __attribute__((always_inline)) {      // aw: f32x16 [out blocks][2]
        constexpr int NOB = (int)(sizeof(aw) / sizeof(aw[0]));
#ifdef PAG_EXP_NO_WGRAD
        return;
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[2], afr[NOB];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) bfr[ib] = tw_frag(Tin, ib, ks, lane);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) afr[ob] = tw_frag(Tz, ob, ks, lane);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) aw[ob][ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ob], bfr[ib], aw[ob][ib], 0, 0, 0);
                if (dbcol >= 0) dbacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ob], ones_col(dbcol + ob), dbacc, 0, 0, 0);
            }
        }
    };
    // dx of tile t is stored at the top of tile t+1 (see mlp_bwd_fused)
    bf16x4 pend[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) pend[mb][g] = bf16x4{(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f};
    int64_t pend_m = min((int64_t)blockIdx.x * 4 * 32 + (int64_t)wave * 32 + r, M - 1);
    auto flush = [&](bool pred) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16_t *dst = reinterpret_cast<bf16_t *>(pi.dx1) + ((int64_t)(4 * mb + g) * M + pend_m) * 8 + 4 * h;
                if (pred) *reinterpret_cast<bf16x4 *>(dst) = pend[mb][g];
            }
    };
    // masked by the saved ReLU output (read from its LDS image), then: fragments of the next chain + the transposed image for dW
    auto mask_pack_put = [&](const bf16_t *Th, bool live, f32x16 (&acc)[2], bf16x8 (&hb)[4]) __attribute__((always_inline)) {
        // (the packed form of relu_mask_pack_put, and issuing the chains before the weight-gradient MFMAs as mlp_bwd_fused does, cost this
        // kernel 8 / 120 bytes of scratch for no gain: 256 + 256 registers are all taken)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x4 hv = *reinterpret_cast<const bf16x4 *>(Th + tw_off(r, 8 * mb + 2 * g + h));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mb][4 * g + j] = ((float)hv[j] > 0.0f && live) ? acc[mb][4 * g + j] : 0.0f;
            }
            pack_block(acc[mb], hb[2 * mb], hb[2 * mb + 1]);
        }
        tw_put_block(Tz, 0, r, h, acc[0]);
        tw_put_block(Tz, 1, r, h, acc[1]);
    };
    auto body = [&](int64_t tile, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t m = tile * 32 + r;
        const bool live = FULL || m < M;
        // ---- the tile's input and both decoders' recomputed hidden activations become LDS images (kept for the whole tile)
        wave_lds_sync();
        {
            bf16x8 xb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) xb[s] = xn[s];
            if (h == 1) xb[3][7] = (bf16_t)1.0f;      // column 63 := 1 (weight 0 in both forwards): its dW columns are the layer-0 bias gradients
            tw_put_frags(Tx, xb, r, h);
            f32x16 ha[2];
            hidden_layer_pinned<4>(W0sI, b0I, xb, r, h, ha);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) ha[mb][q] = fmaxf(ha[mb][q], 0.0f);
                tw_put_block(ThI, mb, r, h, ha[mb]);
            }
            hidden_layer_pinned<4>(W0sS, b0S, xb, r, h, ha);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) ha[mb][q] = fmaxf(ha[mb][q], 0.0f);
                tw_put_block(ThS, mb, r, h, ha[mb]);
            }
        }
        // ---- upstream gradients of this tile
        f32x16 zi[2];
        float zs4[4];          // the softmax head's dz: 4 channels per lane, kept small until its phase
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) zi[ob][4 * g + j] = (float)gz[ob][g][j];
        {
            float zr[4], yv[4], dot = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = 4 * h + j < ps.out_dim;
                zr[j] = ok ? gsc * gs[j] : 0.0f;
                yv[j] = ok ? gy[j] : 0.0f;
                dot += zr[j] * yv[j];
            }
            dot += __shfl_xor(dot, 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) zs4[j] = (FULL || live) ? yv[j] * (zr[j] - dot) : 0.0f;
        }
        // ---- the previous tile's dx goes out, then everything of the next tile is requested
        flush(true);
        prefetch(tile + tile_step, ray1);
        ray1 = ray2;
        ray2 = ridx[row_of(tile + 3 * tile_step)];
        f32x16 acc[2];
        bf16x8 hb[4];
        // ================================================================ .i : upper layer (64 wide), layer 0
        {
            bf16x8 zb[4];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                if constexpr (!FULL) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) zi[ob][q] = live ? zi[ob][q] : 0.0f;
                }
                pack_block(zi[ob], zb[2 * ob], zb[2 * ob + 1]);
                tw_put_block(Tz, ob, r, h, zi[ob]);
            }
            wave_lds_sync();
            wgrad_tile(ThI, awLI, 0);
            wave_lds_sync();
            wt_chain_pinned<2, 4>(WLtI, RSLI, zb, r, h, acc);
        }
        mask_pack_put(ThI, live, acc, hb);
        wave_lds_sync();
        wgrad_tile(Tx, aw0I, -1);
        wave_lds_sync();
        wt_chain_pinned<2, 4>(W0tI, RS, hb, r, h, acc);
        // .i's input gradient waits in bf16 (the rounding the two-launch form applies when it stores it): 8 registers instead of 32
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pend[mb][g] = bf16x4{(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
        // ================================================================ .s : softmax output layer, layer 0
        {
            bf16x8 zb[2];
            f32x16 zs;
#pragma unroll
            for (int q = 0; q < 16; ++q) zs[q] = q < 4 ? zs4[q] : 0.0f;
            pack_block(zs, zb[0], zb[1]);
            tw_put_block(Tz, 0, r, h, zs);
            wave_lds_sync();
            wgrad_tile(ThS, awLS, 2);
            wave_lds_sync();
            wt_chain_pinned<2, 2>(WLtS, RSLS, zb, r, h, acc);
        }
        mask_pack_put(ThS, live, acc, hb);
        wave_lds_sync();
        wgrad_tile(Tx, aw0S, -1);
        wave_lds_sync();
        wt_chain_pinned<2, 4>(W0tS, RS, hb, r, h, acc);
        // ---- dx = both decoders' input gradients: row 32mb + 8g + 4h + j of dx^T -> piece [4mb + g][m][4h + j]
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sum = {acc[mb][4 * g] + (float)pend[mb][g][0], acc[mb][4 * g + 1] + (float)pend[mb][g][1], acc[mb][4 * g + 2] + (float)pend[mb][g][2],
                                   acc[mb][4 * g + 3] + (float)pend[mb][g][3]};
                pend[mb][g] = bf16x4{(bf16_t)sum[0], (bf16_t)sum[1], (bf16_t)sum[2], (bf16_t)sum[3]};
            }
        pend_m = min(m, M - 1);
    };

    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int ray0 = ridx[row_of(tile)];
    ray1 = ridx[row_of(tile + tile_step)];
    ray2 = ridx[row_of(tile + 2 * tile_step)];
    prefetch(tile, ray0);
    const bool any_tile = tile < ntiles;
    bool last_ragged = false;
    for (; tile < nfull; tile += tile_step) body(tile, std::true_type{});
    if (tile < ntiles) {      // the ragged last tile (one wave of the launch)
        body(tile, std::false_type{});
        last_ragged = true;
    }
    if (any_tile) flush(!last_ragged || (nfull * 32 + r) < M);

    // ---- the four waves' accumulators are summed through LDS, then one slab per workgroup and layer (see mlp_bwd_fused)
    constexpr int NBLK = 4 + 4 + 2 + 4 + 1;      // .i upper | .i layer 0 | .s output | .s layer 0 | db
    auto blk = [&](auto bi) -> const f32x16 & {
        constexpr int bb = decltype(bi)::value;
        if constexpr (bb < 4) return awLI[bb >> 1][bb & 1];
        else if constexpr (bb < 8) return aw0I[(bb - 4) >> 1][(bb - 4) & 1];
        else if constexpr (bb < 10) return awLS[0][bb - 8];
        else if constexpr (bb < 14) return aw0S[(bb - 10) >> 1][(bb - 10) & 1];
        else return dbacc;
    };
    __syncthreads();
#ifdef PAG_FUSED_PROF
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    float *red = reinterpret_cast<float *>(smem);                      // [NBLK][4][64 lanes] x 4 floats
    // wave 0 stores, waves 1 - 3 add in turn (same order of summation as a read-modify-write by all four); a block's four 16-byte reads are issued
    // together.  (Written as two code paths: the single loop with `w == 0 ? v : *dst + v` ran one dependent LDS round trip per access, and the form
    // mlp_bwd_fused uses crashes hipcc 7.2's 'AMDGPU Rewrite AGPR-Copy-MFMA' pass on this kernel.)
    if (wave == 0) {
        static_for<NBLK>([&](auto bi) {
            constexpr int bb = decltype(bi)::value;
            const f32x16 &a = blk(bi);
            f32x4 *dst = reinterpret_cast<f32x4 *>(red) + (bb * 4) * 64 + lane;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) dst[q4 * 64] = f32x4{a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
        });
    }
    __syncthreads();
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
            static_for<NBLK>([&](auto bi) {
                constexpr int bb = decltype(bi)::value;
                const f32x16 &a = blk(bi);
                f32x4 *dst = reinterpret_cast<f32x4 *>(red) + (bb * 4) * 64 + lane;
                f32x4 t[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) t[q4] = dst[q4 * 64];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) dst[q4 * 64] = t[q4] + f32x4{a[4 * q4], a[4 * q4 + 1], a[4 * q4 + 2], a[4 * q4 + 3]};
            });
        }
        __syncthreads();
    }
#ifdef PAG_FUSED_PROF
    if (threadIdx.x == 0) {
        const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_fused_prof[0], pt1 - pt0);
        atomicAdd(&g_fused_prof[1], pt2 - pt1);
        atomicAdd(&g_fused_prof[2], pt3 - pt2);
        atomicAdd(&g_fused_prof[4], 1ull);
    }
#endif
    static_for<NBLK>([&](auto bi) {
        constexpr int bb = decltype(bi)::value;
        if ((bb & 3) != wave) return;
        f32x4 sv[4];      // this lane's 16 values of block bb
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) sv[q4] = (reinterpret_cast<const f32x4 *>(red) + (bb * 4 + q4) * 64 + lane)[0];
        auto srcv = [&](int q) __attribute__((always_inline)) { return sv[q >> 2][q & 3]; };
        if constexpr (bb < 4 || (bb >= 8 && bb < 10)) {               // output-layer blocks: rows = output channels
            constexpr bool S = bb >= 8;
            constexpr int ob = S ? 0 : (bb >> 1), ib = S ? (bb - 8) : (bb & 1);
            float *sl = (S ? ps.slabs[1] + (int64_t)blockIdx.x * 32 * WG_SLAB_COLS_F : pi.slabs[1] + (int64_t)blockIdx.x * 64 * WG_SLAB_COLS_F);
#pragma unroll
            for (int q = 0; q < 16; ++q) sl[(32 * ob + rho(q, h)) * WG_SLAB_COLS_F + 32 * ib + r] = srcv(q);
        } else if constexpr (bb < 14) {                               // layer-0 blocks; input column 63 (the ones column) = db of layer 0
            constexpr bool S = bb >= 10;
            constexpr int k = S ? bb - 10 : bb - 4, ob = k >> 1, ib = k & 1;
            float *sl = (S ? ps.slabs[0] : pi.slabs[0]) + (int64_t)blockIdx.x * 64 * WG_SLAB_COLS_F;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = srcv(q);
                sl[(32 * ob + rho(q, h)) * WG_SLAB_COLS_F + 32 * ib + r] = v;
                if (ib == 1 && r == 31) sl[(32 * ob + rho(q, h)) * WG_SLAB_COLS_F + 64] = v;
            }
        } else {      // dbacc: columns 0, 1 -> .i upper layer's blocks; column 2 -> .s output layer
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = srcv(q);
                if (r < 2) pi.slabs[1][((int64_t)blockIdx.x * 64 + 32 * r + rho(q, h)) * WG_SLAB_COLS_F + 64] = v;
                if (r == 2) ps.slabs[1][((int64_t)blockIdx.x * 32 + rho(q, h)) * WG_SLAB_COLS_F + 64] = v;
            }
        }
    });
}

// -------------------------------------------- wide softmax head: output layer backward + its weight gradient (one launch)
// Stage A of the fused backward of a decoder whose last layer is wide (the 200-way instance head): per 32-sample tile the
// probabilities are rebuilt from the saved last hidden layer and the forward's softmax statistics (as mlp_bwd_wide_mfma does),
// dz_L is formed block by block from the rank-1 upstream gradient and consumed at once - by the W_L^T chain (-> the hidden
// layer's gradient, written as a [M,64] bf16 tensor for stage B: mlp_bwd_fused<.., KIND 0, OBL 2> on the remaining layers) and by
// the dW_L accumulators.  The [M, out_dim] softmax gradient (839 MB at M = 2.1 M, written once and read once before) never exists.
//
// The first form of this kernel gave every wave whole tiles and ALL dW_L blocks: 240 accumulator registers = one wave per SIMD,
// where nothing hides a wave's waits - 0.70 ms, VALU 41 % / waiting 38 %, with both passes rebuilding the probabilities for lack
// of registers.  Here the OUTPUT BLOCKS are spread over the waves of a workgroup instead: wave ob (< OB) owns the 32 channels of
// block ob for EVERY tile of the workgroup - it rebuilds that block once per tile (kept in registers between the passes), and
// accumulates dW_L[32 ob .. +31][:] (2 blocks + the bias block: 48 accumulator registers) for the whole launch; one helper wave
// runs the W_L^T chain on the dz fragments the block waves leave in LDS (W_L^T itself stays in its registers), applies the ReLU
// mask, writes the hidden gradient and stages the coming tiles.  What crosses waves per tile: the partial <p, g> of each block
// (256 B per wave) and the dz B-fragments (2 KiB per wave).
//
// The three steps of a tile are SKEWED over three iterations so that ONE workgroup barrier per iteration orders everything and
// every wave has two independent instruction streams between barriers:
//     iteration it:   block waves   finish(it-1): <p,g> from the partials, dz -> LDS fragments, dW_L MFMAs
//                                   rebuild(it) : probabilities of tile it, z = gradient rows, partial <p,g> -> LDS
//                     helper wave   tiles it+1 (registers -> LDS) and it+2 (global -> registers) staged,
//                                   chain(it-2) : W_L^T . dz, ReLU mask, hidden gradient of tile it-2 -> global
// LDS rings: activations 4 deep (tiles it-2 .. it+1), dz fragments / partial dots / gradient rows 2 deep.  Tiles that do not exist
// (the two drain iterations, tiles past the end) run with scale 0: their dz is zero, their stores go to a dump tile.
// OB + 1 = 8 waves per workgroup, one workgroup per CU: two waves per SIMD.  0.43 ms; per-wave barrier waits (-DPAG_WB_PROF): the
// block waves that share a SIMD with another block wave wait 4-30 %, the helper 5 % - the roles are balanced, what is left is the
// issue time of two waves per SIMD.
constexpr int WR_RS = 256;         // floats per staged gradient row
#ifdef PAG_WB_PROF
__device__ unsigned long long g_wb_prof[2][8];      // [0] cycles at the barrier, [1] loop cycles (slots 0-3: helper segments on top); summed over workgroups
#endif
constexpr int WB_RMAX = 4;         // rays whose gradient rows are staged per tile; tiles spanning more read their rows from global
template <int OB>
__global__ __launch_bounds__((OB + 1) * 64) void mlp_bwd_wide_blocks(BwdParams p) {
    PAG_BLOCK_TIMER(4);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RSL = OB * 32 + 8;
    constexpr float LOG2E = 1.4426950408889634f;
    bf16_t *WLt = reinterpret_cast<bf16_t *>(smem);              // [64 hidden][RSL]   k = output channel (permuted)
    bf16_t *WLs = WLt + 64 * RSL;                                // [OB*32 channels][RS] permuted k (forward layout)
    float *bLs = reinterpret_cast<float *>(WLs + OB * 32 * RS);  // [OB*32]
    bf16_t *Th = reinterpret_cast<bf16_t *>(bLs + OB * 32);      // [4][tile]  activations, ring over tiles
    bf16_t *Tp = Th + 4 * TW_ELEMS;                              // helper's transpose buffer for the hidden-gradient store
    bf16_t *TzAll = Tp + TW_ELEMS;                               // [OB][tile] dz block of each block wave (block 0 of its tile)
    float *grow = reinterpret_cast<float *>(TzAll + OB * TW_ELEMS);      // [2][WB_RMAX][WR_RS]
    float *dotbuf = grow + 2 * WB_RMAX * WR_RS;                  // [2][OB][64]
    bf16x8 *zbuf = reinterpret_cast<bf16x8 *>(dotbuf + 2 * OB * 64);     // [2][OB][2][64]
    stage_weight_both(WLs, RS, true, WLt, RSL, OB * 32, 64, p.W[0], p.out_dim, HID);
    for (int e = threadIdx.x; e < OB * 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b_last[e] : -1e30f;      // padding channels: p = exp2(-huge) = 0
    for (int e = threadIdx.x; e < 2 * OB * 64; e += blockDim.x) dotbuf[e] = 0.0f;
    for (int e = threadIdx.x; e < 2 * OB * 2 * 64 * 4; e += blockDim.x) reinterpret_cast<float *>(zbuf)[e] = 0.0f;
    for (int e = threadIdx.x; e < 4 * TW_ELEMS / 2; e += blockDim.x) reinterpret_cast<float *>(Th)[e] = 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t M = p.M, ntiles = (M + 31) / 32;
    const int64_t tile_step = gridDim.x;
    const bf16_t *hsrc = reinterpret_cast<const bf16_t *>(p.hsave[0]);
    bf16_t *dzh = reinterpret_cast<bf16_t *>(p.dz[0]);
    const bool is_block = wave < OB;
    const int ob = is_block ? wave : 0;
    bf16_t *Tz = TzAll + ob * TW_ELEMS;
    auto row_of = [&](int64_t tile) __attribute__((always_inline)) { return min(min(tile, ntiles - 1) * 32 + r, M - 1); };
    // The helper wave stages through registers: loads are issued a whole iteration before the LDS writes that consume them.  Its wave
    // shares a SIMD with a block wave, so every VALU instruction it saves is issue time for both: all its global traffic goes through
    // buffer descriptors (scalar base + loop-invariant lane offset, out-of-range lanes read 0 / are not written) instead of per-lane
    // 64-bit address arithmetic with clamps.
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(hsrc), 0, (int)min(M * HID * 2, (int64_t)0xffffffffll), 0x00020000);
    const auto rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(p.g_index), 0, (int)min(M * 4, (int64_t)0xffffffffll), 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(dzh, 0, (int)min((ntiles + 1) * 32 * HID * 2, (int64_t)0xffffffffll), 0x00020000);
    const int voff_tile = (lane >> 3) * (HID * 2) + (lane & 7) * 16;          // + 8 rows per load
    auto load_rows = [&](float (&v)[4 * WB_RMAX], int ray_first, int ray_last) __attribute__((always_inline)) {      // WB_RMAX gradient rows
#pragma unroll
        for (int k = 0; k < WB_RMAX; ++k) {
            const int rr = min(ray_first + k, ray_last);
            const auto rs_row = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.g_ray) + (int64_t)rr * p.out_dim, 0, p.out_dim * 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * k + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_row, lane * 4, 256 * j, 0));
        }
    };
    auto put_rows = [&](float *dst, const float (&v)[4 * WB_RMAX]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4 * WB_RMAX; ++k) dst[(k >> 2) * WR_RS + 64 * (k & 3) + lane] = v[k];
    };
    auto load_tile = [&](bf16x8 (&v)[4], int64_t tile_raw) __attribute__((always_inline)) {                          // activations of a tile
        const int voff = voff_tile + (int)min(tile_raw, ntiles) * (32 * HID * 2);
#pragma unroll
        for (int it = 0; it < 4; ++it) v[it] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_h, voff + it * (8 * HID * 2), 0, 0));
    };
    auto load_ray = [&](int64_t tile_raw) __attribute__((always_inline)) {                                           // ray of this lane's sample (clamped row)
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rs_i, (int)row_of(tile_raw) * 4, 0, 0);
    };
    const int64_t tile0 = blockIdx.x;
    const int n_own = tile0 < ntiles ? (int)((ntiles - 1 - tile0) / tile_step) + 1 : 0;      // tiles of this workgroup
    // The two roles run their own loops (same number of barriers) so that neither carries the other's registers.  The barrier is
    // the bare instruction behind an LDS-only wait: __syncthreads() would also drain the global loads that are meant to stay in flight.
#ifdef PAG_WB_PROF
    unsigned long long prof_wait = 0, prof_t0 = __builtin_amdgcn_s_memtime(), prof_seg[4] = {0, 0, 0, 0};
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_barrier" ::: "memory");
        prof_wait += __builtin_amdgcn_s_memtime() - t0;
    };
#else
    auto wg_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
#endif
    if (!is_block) {
        // =========================================================================================== helper wave
        bf16x8 vt[4];
        float vr[4 * WB_RMAX];
        load_tile(vt, tile0);
        int ray = load_ray(tile0);
        load_rows(vr, __builtin_amdgcn_readfirstlane(ray), __builtin_amdgcn_readlane(ray, 31));
        tw_put_rows(Th, vt, lane);
        put_rows(grow, vr);
        load_tile(vt, tile0 + tile_step);                                   // tile 1 waits in registers
        ray = load_ray(tile0 + tile_step);
        load_rows(vr, __builtin_amdgcn_readfirstlane(ray), __builtin_amdgcn_readlane(ray, 31));
        int ray_nn = load_ray(tile0 + 2 * tile_step);                       // ray of this lane's sample two tiles ahead
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] = acc[1][q] = 0.0f;
        // W_L^T as MFMA A fragments, resident for the whole launch: rows = hidden units 32 mb + r, k = channels 16 s2 + 8 h ..
        bf16x8 wt[2][2 * OB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int s2 = 0; s2 < 2 * OB; ++s2)
                wt[mb][s2] = *reinterpret_cast<const bf16x8 *>(WLt + (32 * mb + r) * RSL + 16 * s2 + 8 * h);
        wg_barrier();
        for (int it = 0; it < n_own + 2; ++it) {
            const int64_t tile = tile0 + (int64_t)it * tile_step;
            // ---------------- tile it+1 registers -> LDS, tile it+2 global -> registers (in flight for a whole iteration).  The ring slot
            // written here was last read by this wave's own ReLU mask one iteration ago.
#ifdef PAG_WB_PROF
            const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
            tw_put_rows(Th + ((it + 1) & 3) * TW_ELEMS, vt, lane);
            put_rows(grow + ((it + 1) & 1) * (WB_RMAX * WR_RS), vr);
            asm volatile("" ::: "memory");
#ifdef PAG_WB_PROF
            const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
            prof_seg[0] += ts1 - ts0;
#endif
            load_tile(vt, tile + 2 * tile_step);
            load_rows(vr, __builtin_amdgcn_readfirstlane(ray_nn), __builtin_amdgcn_readlane(ray_nn, 31));
            ray_nn = load_ray(tile + 3 * tile_step);
            asm volatile("" ::: "memory");
#ifdef PAG_WB_PROF
            const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
            prof_seg[1] += ts2 - ts1;
#endif
            // ---------------- chain(it-2): dA = W_L^T . dz_L masked by the saved ReLU output -> global.  The stores are unconditional (the
            // buffer has a padding tile and a dump tile): a predicated store is a branch, and past a branch the compiler no longer counts
            // the operations in flight - it would wait for ALL of them, stores included, before the next iteration's LDS writes.
            const bf16x8 *zbp = zbuf + (it & 1) * (OB * 2 * 64);
            const bf16_t *Thq = Th + ((it + 2) & 3) * TW_ELEMS;
            // every dz fragment is requested before the first MFMA (W_L^T stays in registers): one LDS latency per tile, not one per MFMA
            bf16x8 zf[OB];          // rolling window: fragment s2 + OB is requested when fragment s2 has been consumed
#pragma unroll
            for (int s2 = 0; s2 < OB; ++s2) zf[s2] = zbp[s2 * 64 + lane];
#pragma unroll
            for (int s2 = 0; s2 < 2 * OB; ++s2) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[mb][s2], zf[s2 % OB], acc[mb], 0, 0, 0);
                if (s2 < OB) zf[s2] = zbp[(s2 + OB) * 64 + lane];
            }
            // the order above is the order wanted: OB reads, then (2 MFMAs, 1 read) x OB, then the remaining MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, OB, 0);
#pragma unroll
            for (int s2 = 0; s2 < OB; ++s2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * OB, 0);
            asm volatile("" ::: "memory");
#ifdef PAG_WB_PROF
            asm volatile("s_nop 0" : "+v"(acc[0]), "+v"(acc[1]));
            const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
            prof_seg[2] += ts3 - ts2;
#endif
            // ReLU mask and bf16 rounding on PAIRS: the saved activation is 0 or positive, so min(max(h as i16, 0), 1) is the 0 / 1
            // mask of a half-word and a 16-bit multiply applies it - 4 instructions per pair instead of 7
            typedef short i16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x2 hv = *reinterpret_cast<const u32x2 *>(Thq + tw_off(r, 8 * mb + 2 * g + h));
                    u32x2 o;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
                        const bf16x2 a2 = {(bf16_t)acc[mb][4 * g + 2 * d], (bf16_t)acc[mb][4 * g + 2 * d + 1]};
                        unsigned int m;
                        asm("v_pk_max_i16 %0, %1, %2\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(m) : "v"(hv[d]), "s"(0u), "s"(0x00010001u));
                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(o[d]) : "v"(__builtin_bit_cast(unsigned int, a2)), "v"(m));
                    }
                    *reinterpret_cast<u32x2 *>(Tp + tw_off(r, 8 * mb + 2 * g + h)) = o;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
            }
            wave_lds_sync();
            {
                const int voff = voff_tile + (int)(it >= 2 ? tile - 2 * tile_step : ntiles) * (32 * HID * 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rowl = k * 8 + (lane >> 3), ch = (lane & 7) * 2;
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(Tp + tw_off(rowl, ch));
                    const u32x2 hi = *reinterpret_cast<const u32x2 *>(Tp + tw_off(rowl, ch + 1));
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, rs_d, voff + k * (8 * HID * 2), 0, 0);
                }
            }
#ifdef PAG_WB_PROF
            prof_seg[3] += __builtin_amdgcn_s_memtime() - ts3;
#endif
            wg_barrier();
        }
#ifdef PAG_WB_PROF
        if (lane == 0)
            for (int k = 0; k < 4; ++k) atomicAdd(&g_wb_prof[1][k], prof_seg[k]);
        if (lane == 0) {
            atomicAdd(&g_wb_prof[0][wave], prof_wait);
            atomicAdd(&g_wb_prof[1][wave], __builtin_amdgcn_s_memtime() - prof_t0);
        }
#endif
        return;
    }
    // =============================================================================================== block waves
    f32x16 aw[2], dbacc;        // dW_L of the own block (2 in-blocks) + bias block
#pragma unroll
    for (int q = 0; q < 16; ++q) aw[0][q] = aw[1][q] = dbacc[q] = 0.0f;
    bf16x8 ones0;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones0[e] = (bf16_t)(r == 0 ? 1.0f : 0.0f);
    // per-lane scalars of the tile the NEXT iteration rebuilds; the ray index runs one tile further ahead so that the load of the
    // per-ray scale it addresses never waits for it inside an iteration
    int ray_c = p.g_index[row_of(tile0)], ray_n1 = p.g_index[row_of(tile0 + tile_step)];
    float2 st_n = *reinterpret_cast<const float2 *>(p.stats + 2 * row_of(tile0));
    float gsa_n = p.g_scale[row_of(tile0)], grs_n = p.g_ray_scale[ray_c];
    // carried from rebuild(it) to finish(it) one iteration later: probabilities, the ray's raw gradient slice, its scale (0 = dead row).
    // The element-wise arithmetic is written on pairs: packed fp32 instructions do two elements per issue slot.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pf_c[8], v_c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) pf_c[q] = v_c[q] = f32x2{0.0f, 0.0f};
    float gsc_c = 0.0f;
    wg_barrier();

    // One iteration = finish(it-1) and rebuild(it), two independent instruction streams.  They are INTERLEAVED in program order (the
    // LDS queue is in order, so program order decides what a wait waits for): every LDS round trip of one stream runs under arithmetic
    // of the other.
    for (int it = 0; it < n_own + 2; ++it) {
        const int64_t tile = tile0 + (int64_t)it * tile_step;
        asm volatile("" ::: "memory");          // keep the loop-invariant weight fragments in LDS, not hoisted into registers
        const bf16_t *Thc = Th + (it & 3) * TW_ELEMS, *Thp = Th + ((it + 3) & 3) * TW_ELEMS;
        const bool live = tile * 32 + r < M && it < n_own;
        const float Ms = st_n.x, inv = st_n.y, g_sc = live ? __fmul_rn(gsa_n, grs_n) : 0.0f;
        const int ray = ray_c;
        const int ray_first = __builtin_amdgcn_readfirstlane(ray), ray_last = __builtin_amdgcn_readlane(ray, 31);
        // ---- rebuild(it), reads: this lane's slice of its ray's gradient row - from the staged rows, or from global when the tile spans
        //      too many rays (the only branch of the loop, kept first: a join drains every load still in flight)
        f32x16 z;
        if (ray_last - ray_first < WB_RMAX) {
            const float *grow_l = grow + (it & 1) * (WB_RMAX * WR_RS) + (ray - ray_first) * WR_RS;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(grow_l + 32 * ob + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[4 * g + j] = v[j];
            }
        } else {
            const float *gr = p.g_ray + (int64_t)ray * p.out_dim;
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = gr[min(32 * ob + rho(q, h), p.out_dim - 1)];
        }
        //      ... the forward's B operand of the output layer (exact: the saved bf16 activations), bias, weight fragments
        bf16x8 hbL[4], wa[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(Thc + tw_off(r, 8 * mb + 4 * half + h));
                const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(Thc + tw_off(r, 8 * mb + 4 * half + 2 + h));
                hbL[2 * mb + half] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        f32x16 pf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bLs + 32 * ob + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[4 * g + j] = b4[j];
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) wa[s2] = *reinterpret_cast<const bf16x8 *>(WLs + (32 * ob + r) * RS + 16 * s2 + 8 * h);
        // ---- finish(it-1): <p, g> from the partials, dz = p (g - <p, g>) with g = scale * row:  p_bf16 * fma(row, scale, -scale <p, row>)
        {
            const float *dotp = dotbuf + ((it + 1) & 1) * (OB * 64);
            bf16x8 *zbp = zbuf + ((it + 1) & 1) * (OB * 2 * 64);
            float dot = 0.0f;
#pragma unroll
            for (int w = 0; w < OB; ++w) dot += dotp[w * 64 + lane];
            dot += __shfl_xor(dot, 32);
            const f32x2 sc2 = {gsc_c, gsc_c}, nd2 = {-gsc_c * dot, -gsc_c * dot};
            typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            unsigned int zp[8];          // dz rounded to bf16, channel pairs (2q, 2q+1): the helper's B fragments and the own transposed tile
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x2 pb = {(float)(bf16_t)pf_c[q][0], (float)(bf16_t)pf_c[q][1]};
                const f32x2 t = pb * (v_c[q] * sc2 + nd2);
                zp[q] = __builtin_bit_cast(unsigned int, bf16x2{(bf16_t)t[0], (bf16_t)t[1]});
            }
            *reinterpret_cast<u32x4 *>(zbp + (ob * 2 + 0) * 64 + lane) = u32x4{zp[0], zp[1], zp[2], zp[3]};
            *reinterpret_cast<u32x4 *>(zbp + (ob * 2 + 1) * 64 + lane) = u32x4{zp[4], zp[5], zp[6], zp[7]};
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2 *>(Tz + tw_off(r, 2 * g + h)) = u32x2{zp[2 * g], zp[2 * g + 1]};
        }
        // ---- rebuild(it): logits
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) pf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s2], hbL[s2], pf, 0, 0, 0);
        // ---- finish(it-1): operand fragments of dW_L (k = sample) come back transposed from LDS
        wave_lds_sync();
        bf16x8 fz[2], fh[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            fz[ks] = tw_frag(Tz, 0, ks, lane);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) fh[ks][ib] = tw_frag(Thp, ib, ks, lane);
        }
        // ---- rebuild(it): next tile's scalars (consumed next iteration), probabilities, partial <p, row>
        {
            const int64_t m1 = row_of(tile + tile_step);
            st_n = *reinterpret_cast<const float2 *>(p.stats + 2 * m1);
            gsa_n = p.g_scale[m1];
            grs_n = p.g_ray_scale[ray_n1];
            ray_c = ray_n1;
            ray_n1 = p.g_index[row_of(tile + 2 * tile_step)];
            const f32x2 l2 = {LOG2E, LOG2E}, nm2 = {-Ms, -Ms}, inv2 = {inv, inv};
            f32x2 dp2 = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x2 t = f32x2{pf[2 * q], pf[2 * q + 1]} * l2 + nm2;
                const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} * inv2;
                const f32x2 v = {z[2 * q], z[2 * q + 1]};
                dp2 = e * v + dp2;
                pf_c[q] = e;
                v_c[q] = v;
            }
            dotbuf[(it & 1) * (OB * 64) + ob * 64 + lane] = dp2[0] + dp2[1];
            gsc_c = g_sc;
        }
        // ---- finish(it-1): dW_L and its bias column
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) aw[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz[ks], fh[ks][ib], aw[ib], 0, 0, 0);
            dbacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz[ks], ones0, dbacc, 0, 0, 0);
        }
        wg_barrier();
    }
#ifdef PAG_WB_PROF
    if (lane == 0) {
        atomicAdd(&g_wb_prof[0][wave], prof_wait);
        atomicAdd(&g_wb_prof[1][wave], __builtin_amdgcn_s_memtime() - prof_t0);
    }
#endif
    // ---- every block wave owns its 32 rows of the workgroup's slab [OB*32][96]: cols 0..63 dW_L, col 64 db
    {
        float *sl = p.slabs[0] + (int64_t)blockIdx.x * (OB * 32) * WG_SLAB_COLS_F;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = 32 * ob + rho(q, h);
            sl[row * WG_SLAB_COLS_F + r] = aw[0][q];
            sl[row * WG_SLAB_COLS_F + 32 + r] = aw[1][q];
            if (r == 0) sl[row * WG_SLAB_COLS_F + 64] = dbacc[q];
        }
    }
}

// ------------------------------------------------------------------- backward of wide softmax heads
// The 200-way instance head dominated the decoder backward: its [M,200] probabilities were streamed through twice
// (dot product of the softmax backward, then dz) - 1.7 GB per launch.  With the forward's per-sample softmax statistics
// (max*log2e, 1/sum) the probabilities of a 32-channel block are instead REBUILT from the saved last hidden layer with
// the forward's own instruction sequence (bias + 4 MFMAs on the idle matrix cores + exp2), so the kernel reads only the
// two [M,64] hidden tensors.  Both layouts of W_L are needed in LDS (rows = channels for the rebuild, rows = hidden
// units for W_L^T . dz), 62 KiB - the workgroup therefore has NW = 8 or 16 waves sharing one copy instead of two
// 4-wave workgroups per CU.
template <typename DxT, int NL, int NW>
__global__ __launch_bounds__(NW * 64) void mlp_bwd_wide_mfma(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OB = (p.out_dim + 31) / 32;
    const int RSL = OB * 32 + 8;
    bf16_t *WLt = reinterpret_cast<bf16_t *>(smem);              // [64 hidden][RSL]   k = output channel (permuted)
    bf16_t *WLs = WLt + 64 * RSL;                                // [OB*32 channels][RS] permuted k (forward layout)
    bf16_t *W1t = WLs + OB * 32 * RS;                            // [64][RS]      (NL == 3)
    bf16_t *W0t = W1t + (NL == 3 ? 64 * RS : 0);                 // [64 in-feature rows][RS]
    float *bLs = reinterpret_cast<float *>(W0t + 64 * RS);       // [OB*32]
    bf16_t *stg = reinterpret_cast<bf16_t *>(bLs + OB * 32) + (threadIdx.x >> 6) * (ST_BYTES / 2);
    float *grow = reinterpret_cast<float *>(reinterpret_cast<bf16_t *>(bLs + OB * 32) + NW * (ST_BYTES / 2)) +
                  (threadIdx.x >> 6) * (OB * 32);              // wave-private copy of the tile's upstream gradient row [OB*32]
    stage_weight_t(WLt, RSL, 64, OB * 32, p.W[NL - 1], p.out_dim, HID);
    stage_weight(WLs, RS, OB * 32, 64, p.W[NL - 1], p.out_dim, HID, true);
    if (NL == 3) stage_weight_t(W1t, RS, 64, 64, p.W[1], HID, HID);
    if (p.dx1) stage_weight_t(W0t, RS, 64, 64, p.W[0], HID, p.in_dim, p.grp_L, p.grp_F);
    for (int e = threadIdx.x; e < OB * 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b_last[e] : 0.0f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (p.M + 31) / 32;
    const bool vec_out = (p.out_dim % 4) == 0;
    constexpr float LOG2E = 1.4426950408889634f;
    const int64_t tile_step = (int64_t)gridDim.x * NW;
    // the next tile's hidden rows are requested as soon as the probability blocks of the current one are dead (after the
    // second pass), so their round trip hides under the rest of the layer chain instead of opening the next tile
    bf16x8 hnext[NL - 1][4];
    auto fetch_h = [&](int64_t tile) __attribute__((always_inline)) {
        if (tile < ntiles) {
            const int rv = (int)min((int64_t)32, p.M - tile * 32);
#pragma unroll
            for (int l = 0; l < NL - 1; ++l)
                tile64_fetch(reinterpret_cast<const bf16_t *>(p.hsave[l]) + tile * 32 * HID, rv, lane, hnext[l]);
        }
    };
    fetch_h((int64_t)blockIdx.x * NW + wave);
    // ... and so are its per-sample scalars and (when the tile lies inside one ray) the ray's gradient row
    const bool r1 = p.g_ray != nullptr;
    float2 st_n = {0.0f, 1.0f};
    int gi_n = 0;
    float gs_n = 0.0f, grow_n[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool uni_n = false;
    auto fetch_s = [&](int64_t tile) __attribute__((always_inline)) {
        if (tile < ntiles) {
            const int64_t m_ = tile * 32 + r;
            const int64_t mc_ = m_ < p.M ? m_ : p.M - 1;
            st_n = *reinterpret_cast<const float2 *>(p.stats + 2 * mc_);
            if (r1) {
                gi_n = p.g_index[mc_];
                gs_n = p.g_ray_scale ? __fmul_rn(p.g_scale[mc_], p.g_ray_scale[gi_n]) : p.g_scale[mc_];
                const int g0 = __builtin_amdgcn_readfirstlane(gi_n);
                uni_n = __all(gi_n == g0);
                if (uni_n) {
                    const float *row = p.g_ray + (int64_t)g0 * p.out_dim;
#pragma unroll
                    for (int k = 0; k < 4; ++k) grow_n[k] = (lane + 64 * k < p.out_dim) ? row[lane + 64 * k] : 0.0f;
                }
            }
        }
    };
    fetch_s((int64_t)blockIdx.x * NW + wave);
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += tile_step) {
        asm volatile("" : "+v"(r), "+v"(h));      // keep lane-constant addresses from being hoisted and spilled
        const int64_t m = tile * 32 + r;
        const bool live = m < p.M;
        const int64_t mc = live ? m : p.M - 1;
        const int rows_valid = (int)min((int64_t)32, p.M - tile * 32);
        const bool tile_full = (tile + 1) * 32 <= p.M;
        bf16x4 hraw[NL - 1][2][4];
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) tile64_unstage(stg, hnext[l], lane, r, h, hraw[l]);
        // the forward's B operand of the output layer: the saved bf16 activations, re-packed (exact)
        bf16x8 hbL[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x16 hv;
            raw_to_block(hraw[NL - 2][mb], hv);
            pack_block(hv, hbL[2 * mb], hbL[2 * mb + 1]);
        }
        const float Ms = st_n.x, inv = st_n.y;
        auto prob_block = [&](int ob, f32x16 &o) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // rows rho(4g..4g+3, h) = 8g + 4h + 0..3: one 16-byte LDS read
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bLs + 32 * ob + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[4 * g + j] = b4[j];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a = *reinterpret_cast<const bf16x8 *>(WLs + (32 * ob + r) * RS + 16 * s + 8 * h);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hbL[s], o, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) o[q] = __builtin_amdgcn_exp2f(fmaf(o[q], LOG2E, -Ms)) * inv;
            if (ob == OB - 1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) o[q] = (32 * ob + rho(q, h) < p.out_dim) ? o[q] : 0.0f;
            }
        };
        // upstream gradient of the probabilities: rank-1 (scale_m * G[ray_m]) or a dense bf16 [M,out_dim] tensor
        const int g_idx1 = gi_n;
        const bool g_uni = uni_n;
        const float *g_row1 = r1 ? p.g_ray + (int64_t)g_idx1 * p.out_dim : nullptr;
        const float g_sc1 = gs_n;
        const bf16_t *gtile_g = reinterpret_cast<const bf16_t *>(p.grad_out) + tile * 32 * p.out_dim;
        if (r1 && g_uni) {      // whole tile inside one ray (the common case): its gradient row goes to LDS once - the
            // per-block scalar loads of rank1_block_uniform each exposed a full round trip with 2 waves per SIMD
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k < OB * 32) grow[lane + 64 * k] = grow_n[k];
            wave_lds_sync();
        }
        auto grad_block = [&](int ob, f32x16 &z) __attribute__((always_inline)) {
            if (r1 && g_uni) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(grow + 32 * ob + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[4 * g + j] = g_sc1 * v[j];
                }
            } else if (r1) {
                rank1_block(g_row1, g_sc1, 32 * ob, h, p.out_dim, z);
            } else {
                bf16x4 rz[4];
                block32_stage_in(stg, 0, gtile_g, p.out_dim, 32 * ob, rows_valid, lane);
                wave_lds_sync();
                block32_read(stg, 0, r, h, rz);
                wave_lds_sync();
                raw_to_block(rz, z);
            }
        };
        // pass 1 keeps the rebuilt probability blocks in registers (7 x 16 floats: the 8-wave workgroup runs 2 waves per
        // SIMD, i.e. a 256-VGPR budget) - the exp2 of the rebuild is a quarter-rate instruction and was the largest item
        float dot = 0.0f;
        bf16x8 pr[7][2];      // kept as bf16 (the precision the stored probabilities had): 56 registers instead of 112
#pragma unroll
        for (int ob = 0; ob < 7; ++ob) {
            if (ob < OB) {
                f32x16 z, pf;
                prob_block(ob, pf);
                grad_block(ob, z);
#pragma unroll
                for (int q = 0; q < 16; ++q) dot += pf[q] * z[q];
                pack_block(pf, pr[ob][0], pr[ob][1]);
            }
        }
        dot += __shfl_xor(dot, 32);
        f32x16 acc[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
#pragma unroll
        for (int ob = 0; ob < 7; ++ob) {
            if (ob >= OB) break;
            f32x16 zz, z;
            grad_block(ob, z);
#pragma unroll
            for (int q = 0; q < 16; ++q) zz[q] = (float)pr[ob][q >> 3][q & 7] * (z[q] - dot);
            if (!tile_full) {
#pragma unroll
                for (int q = 0; q < 16; ++q) zz[q] = live ? zz[q] : 0.0f;
            }
            bf16x8 zb[2];
            pack_block(zz, zb[0], zb[1]);
            if ((p.out_dim & 7) == 0)
                block32_store(stg, reinterpret_cast<bf16_t *>(p.dz[NL - 1]) + tile * 32 * p.out_dim, p.out_dim, 32 * ob, rows_valid, lane, r, h, zz);
            else if (live)
                store_block(reinterpret_cast<bf16_t *>(p.dz[NL - 1]) + m * p.out_dim, 32 * ob, h, zz, p.out_dim, vec_out);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    bf16x8 a = *reinterpret_cast<const bf16x8 *>(WLt + (32 * mb + r) * RSL + 16 * (2 * ob + half) + 8 * h);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, zb[half], acc[mb], 0, 0, 0);
                }
        }
        fetch_h(tile + tile_step);
        fetch_s(tile + tile_step);
        // ---- dA = W_L^T . dz_L masked by the saved ReLU output, then down the chain exactly as mlp_bwd_mfma
        bf16x8 hb[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f32x16 hv;
            raw_to_block(hraw[NL - 2][mb], hv);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mb][q] = (hv[q] > 0.0f && live) ? acc[mb][q] : 0.0f;
            pack_block(acc[mb], hb[2 * mb], hb[2 * mb + 1]);
        }
        tile64_store(stg, reinterpret_cast<bf16_t *>(p.dz[NL - 2]) + tile * 32 * HID, rows_valid, lane, r, h, acc);
        if (NL == 3) {
            bf16x8 hb2[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 a = *reinterpret_cast<const bf16x8 *>(W1t + (32 * mb + r) * RS + 16 * s + 8 * h);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[s], acc[mb], 0, 0, 0);
                }
                f32x16 hv;
                raw_to_block(hraw[0][mb], hv);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[mb][q] = (hv[q] > 0.0f && live) ? acc[mb][q] : 0.0f;
                pack_block(acc[mb], hb2[2 * mb], hb2[2 * mb + 1]);
            }
            tile64_store(stg, reinterpret_cast<bf16_t *>(p.dz[0]) + tile * 32 * HID, rows_valid, lane, r, h, acc);
#pragma unroll
            for (int s = 0; s < 4; ++s) hb[s] = hb2[s];
        }
        if (p.dx1) {
            DxT *dx = reinterpret_cast<DxT *>(p.dx1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (32 * mb < p.k1) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[mb][q] = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        bf16x8 a = *reinterpret_cast<const bf16x8 *>(W0t + (32 * mb + r) * RS + 16 * s + 8 * h);
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[s], acc[mb], 0, 0, 0);
                    }
                    if (live && !p.grp_L) store_block(dx + m * p.k1, 32 * mb, h, acc[mb], p.k1, true);
                    if (live && p.grp_L) {
                        bf16_t *dg = reinterpret_cast<bf16_t *>(p.dx1);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 *dst = reinterpret_cast<bf16x4 *>(dg + ((int64_t)(4 * mb + g) * p.M + m) * 8 + 4 * h);
                            if (p.dx1_acc) {
                                const bf16x4 o = *dst;
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[mb][4 * g + j] += (float)o[j];
                            }
                            bf16x4 v = {(bf16_t)acc[mb][4 * g], (bf16_t)acc[mb][4 * g + 1], (bf16_t)acc[mb][4 * g + 2], (bf16_t)acc[mb][4 * g + 3]};
                            *dst = v;
                        }
                    }
                }
            }
        }
    }
}

// --------------------------------------------------- wide softmax head + per-ray weighted sum, forward
// out[ray][c] = alpha[ray] * sum_i w_i * softmax(W_L h_i + b_L)[c]   (tracer :197-205 on the instance head) WITHOUT the
// [M,200] probability tensor: the decoder forward only writes the per-sample softmax statistics, and this kernel -
// one workgroup per ray, waves taking 32-sample tiles of that ray - rebuilds each tile's probabilities from the saved
// last hidden layer exactly as mlp_bwd_wide_mfma does (same fragments, same MFMA sequence as the forward), scales
// them by the sample weight and keeps the 7 x 16 per-lane partial sums in registers until the ray is finished.  The
// sum over a tile's 32 samples is a sum over lanes: DPP row_shr 1/2/4/8 + row_bcast:15 put it on lanes 31 / 63, LDS
// combines the four waves.  Reads 128 B per sample instead of 400 B, and the 839 MB tensor is never written.
#ifndef PAG_HC_PER_WAVE_P
#define PAG_HC_PER_WAVE_P 2048
#endif
struct HeadCompParams {
    const int64_t *pack_start;
    const int32_t *ray_of_pack;
    int64_t P;
    const bf16_t *hidden;      // [M,64] last hidden layer (hidden_save of the forward)
    const float *W, *b;        // [out_dim,64], [out_dim]
    int out_dim;
    const float *stats;        // [M,2]
    int per_wave;              // 1: one WAVE per pack (short packs: the voxel regime has ~80 samples per ray), 0: one workgroup per pack
    const float *weights;      // [M]
    const float *alpha;        // [N]
    float *out;                // [N,out_dim]
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}

// OBT = 7 (193 - 224 outputs: the 200-instance head): the block loop is compile-time, so the output blocks' MFMA chains are issued one block
// AHEAD of the exponentials that consume them (the run-time `ob < OB` guards of the generic form, OBT = 0, end a scheduling region per block
// and every block ran as reads -> 4 chained MFMAs -> 16 exponentials, nothing overlapping at two waves per SIMD).
template <int OBT>
__global__ __launch_bounds__(256, 2) void head_composite_fwd_kernel(HeadCompParams p) {      // two waves per SIMD: 256 registers
    PAG_BLOCK_TIMER(5);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OB = OBT ? OBT : (p.out_dim + 31) / 32;
    bf16_t *WLs = reinterpret_cast<bf16_t *>(smem);                  // [OB*32][RS] permuted k (forward layout)
    float *bLs = reinterpret_cast<float *>(WLs + OB * 32 * RS);      // [OB*32]
    float *red = bLs + OB * 32;                                      // [4][OB*32]
    bf16_t *stg = reinterpret_cast<bf16_t *>(red + 4 * OB * 32) + (threadIdx.x >> 6) * (ST_BYTES / 2);
    stage_weight(WLs, RS, OB * 32, 64, p.W, p.out_dim, HID, true);
    for (int e = threadIdx.x; e < OB * 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b[e] : 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int r = lane & 31, h = lane >> 5;
    constexpr float LOG2E = 1.4426950408889634f;
    // One workgroup per pack (its four waves take every fourth 32-sample tile and their partial sums meet in LDS) suits 512-sample
    // rays; with ~80 samples per ray (voxel march after the first prune) two of the waves have no tile at all and the per-pack
    // reduction + two block barriers are most of the time (0.090 ms for 325 k samples against 0.165 ms for 2.1 M).  Short packs
    // therefore go one per WAVE: four packs of a workgroup proceed independently, no barrier, each wave reduces its own sums.
    const int pw = p.per_wave;
    const int64_t pk0 = pw ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x, pk_step = pw ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x;
    const int t_first = pw ? 0 : wave, t_step = pw ? 1 : 4;
    for (int64_t pk = pk0; pk < p.P; pk += pk_step) {
        asm volatile("" : "+v"(r), "+v"(h));
        const int64_t beg = p.pack_start[pk], end = p.pack_start[pk + 1];
        const int64_t ntile = (end - beg + 31) / 32;
        f32x16 acc[7];
#pragma unroll
        for (int ob = 0; ob < 7; ++ob)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ob][q] = 0.0f;
        bf16x8 hnext[4];
        float2 stn = {0.0f, 0.0f};
        float wn = 0.0f;
        auto fetch = [&](int64_t t) __attribute__((always_inline)) {      // next tile's rows / statistics / weight: in flight during this tile
            if (t < ntile) {
                const int64_t base = beg + 32 * t;
                const int rv = (int)min((int64_t)32, end - base);
                tile64_fetch(p.hidden + base * HID, rv, lane, hnext);
                const int64_t mc = r < rv ? base + r : base;
                stn = *reinterpret_cast<const float2 *>(p.stats + 2 * mc);
                wn = r < rv ? p.weights[mc] : 0.0f;
            }
        };
        fetch(t_first);
        for (int64_t t = t_first; t < ntile; t += t_step) {
            bf16x4 hraw[2][4];
            tile64_unstage(stg, hnext, lane, r, h, hraw);
            const float2 st2 = stn;
            const float wcur = wn;
            fetch(t + t_step);
            bf16x8 hb[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 hv;
                raw_to_block(hraw[mb], hv);
                pack_block(hv, hb[2 * mb], hb[2 * mb + 1]);
            }
            const float Ms = st2.x, sw = st2.y * wcur;       // 1/sum folded into the sample weight (0 for lanes past the pack's end)
            auto logits = [&](int ob) __attribute__((always_inline)) {
                f32x16 o;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bLs + 32 * ob + 8 * g + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * g + j] = b4[j];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 a = *reinterpret_cast<const bf16x8 *>(WLs + (32 * ob + r) * RS + 16 * s + 8 * h);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[s], o, 0, 0, 0);
                }
                return o;
            };
            if constexpr (OBT != 0) {
                f32x16 o_next = logits(0);
#pragma unroll
                for (int ob = 0; ob < OBT; ++ob) {
                    const f32x16 o = o_next;
                    if (ob + 1 < OBT) o_next = logits(ob + 1);        // in the matrix pipe while the exponentials below issue
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[ob][q] = fmaf(sw, __builtin_amdgcn_exp2f(fmaf(o[q], LOG2E, -Ms)), acc[ob][q]);
                }
            } else {
#pragma unroll
                for (int ob = 0; ob < 7; ++ob) {
                    if (ob < OB) {
                        const f32x16 o = logits(ob);
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[ob][q] = fmaf(sw, __builtin_amdgcn_exp2f(fmaf(o[q], LOG2E, -Ms)), acc[ob][q]);
                    }
                }
            }
        }
        // ---- sum over the 32 sample lanes of each half, then over the four waves
#pragma unroll
        for (int ob = 0; ob < 7; ++ob) {
            if (ob < OB) {
                // step by step over all 16 values: consecutive DPP adds are independent (a chain per value would put two wait states
                // between every pair - 459 s_nop in the first listing of this epilogue)
                // (written as asm blocks of 16 v_add_f32_dpp: left to itself the compiler pairs the values into v_pk_add_f32 fed by two DPP moves
                // each - 1.5 instructions per value and step instead of 1.  Inside a block every DPP source was written 16 instructions earlier.)
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = acc[ob][q];
#define PAG_DPP_ADD16(ctl)                                                                                                         \
                asm("s_nop 1\n\t"                                                                                                   \
                    "v_add_f32_dpp %0, %0, %0 " ctl "\n\tv_add_f32_dpp %1, %1, %1 " ctl "\n\tv_add_f32_dpp %2, %2, %2 " ctl "\n\t"   \
                    "v_add_f32_dpp %3, %3, %3 " ctl "\n\tv_add_f32_dpp %4, %4, %4 " ctl "\n\tv_add_f32_dpp %5, %5, %5 " ctl "\n\t"   \
                    "v_add_f32_dpp %6, %6, %6 " ctl "\n\tv_add_f32_dpp %7, %7, %7 " ctl "\n\tv_add_f32_dpp %8, %8, %8 " ctl "\n\t"   \
                    "v_add_f32_dpp %9, %9, %9 " ctl "\n\tv_add_f32_dpp %10, %10, %10 " ctl "\n\tv_add_f32_dpp %11, %11, %11 " ctl "\n\t" \
                    "v_add_f32_dpp %12, %12, %12 " ctl "\n\tv_add_f32_dpp %13, %13, %13 " ctl "\n\tv_add_f32_dpp %14, %14, %14 " ctl "\n\t" \
                    "v_add_f32_dpp %15, %15, %15 " ctl                                                                             \
                    : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),     \
                      "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))
                PAG_DPP_ADD16("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
                PAG_DPP_ADD16("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
                PAG_DPP_ADD16("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
                PAG_DPP_ADD16("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");      // lane 15 of every 16-lane row holds the row sum
                PAG_DPP_ADD16("row_bcast:15 row_mask:0xa bank_mask:0xf");                // into rows 1, 3: lanes 31 / 63 hold the half's sum
#undef PAG_DPP_ADD16
                if (r == 31) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) red[wave * OB * 32 + 32 * ob + rho(q, h)] = v[q];
                }
            }
        }
        const int32_t ray = p.ray_of_pack[pk];
        const float al = p.alpha[ray];
        if (pw) {           // this wave's own section of `red`: LDS accesses of one wave complete in program order, no barrier
            for (int c = lane; c < p.out_dim; c += 64) p.out[(int64_t)ray * p.out_dim + c] = al * red[wave * OB * 32 + c];
        } else {
            __syncthreads();
            for (int c = threadIdx.x; c < p.out_dim; c += blockDim.x)
                p.out[(int64_t)ray * p.out_dim + c] = al * (red[c] + red[OB * 32 + c] + red[2 * OB * 32 + c] + red[3 * OB * 32 + c]);
            __syncthreads();
        }
    }
}

// --------------------------------------------------- wide softmax head: decoder + per-ray weighted sum in ONE pass over the logits
// mlp_fwd_wide_stats + head_composite_fwd_kernel form every logit twice (once for the softmax statistics, once - from the saved hidden layer - for the
// weighted sum) and take 224 exponentials per sample where 112 are needed.  Here a wave owns whole 32-sample tiles OF ONE RAY: hidden layers (and
// the companion narrow head, PAIR) as in mlp_fwd_wide_stats, then all seven 32-channel logit blocks stay in registers (112) while their maximum, the
// exponentials and their sum are formed once; (max * log2e, 1 / sum) and the last hidden layer are written for the backward exactly where the two-launch
// form writes them, and the exponentials - scaled by w_i / sum - go straight into the ray's 112 per-lane partial sums (the epilogue of
// head_composite_fwd_kernel).  224 accumulator registers: one wave per SIMD; what hides latency is the next tile's features in flight and the
// independent MFMA chains of the seven blocks.  Same logits bit for bit as the two-launch form (same fragments, same MFMA order); the sum of the
// exponentials is formed directly instead of online over the blocks, so 1 / sum - and with it the outputs - may differ in the last bit.
template <bool PAIR>
__global__ __launch_bounds__(256, 1) void head_fwd_once_kernel(FwdParams p, HeadCompParams c) {
    PAG_BLOCK_TIMER(7);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int OB = 7;
    constexpr float LOG2E = 1.4426950408889634f;
    bf16_t *W0s = reinterpret_cast<bf16_t *>(smem);
    bf16_t *W1s = W0s + 64 * RS;
    bf16_t *WLs = W1s + 64 * RS;                                     // [OB*32][RS] permuted k
    bf16_t *W0s2 = WLs + OB * 32 * RS;                               // PAIR: [64][RS] natural k, [32][RS] permuted k
    bf16_t *WLs2 = W0s2 + (PAIR ? 64 * RS : 0);
    float *b0s = reinterpret_cast<float *>(WLs2 + (PAIR ? 32 * RS : 0));
    float *b1s = b0s + 64;
    float *bLs = b1s + 64;
    float *b0s2 = bLs + OB * 32;
    float *bLs2 = b0s2 + (PAIR ? 64 : 0);
    float *red = bLs2 + (PAIR ? 32 : 0);                             // [4][OB*32]
    bf16_t *stg = reinterpret_cast<bf16_t *>(red + 4 * OB * 32) + (threadIdx.x >> 6) * (ST_BYTES / 2);
    stage_weight(W0s, RS, 64, 64, p.W[0], HID, p.in_dim, false, p.grp_L, p.grp_F);
    stage_weight(W1s, RS, 64, 64, p.W[1], HID, HID, true);
    stage_weight(WLs, RS, OB * 32, 64, p.W[2], p.out_dim, HID, true);
    if constexpr (PAIR) {
        stage_weight(W0s2, RS, 64, 64, p.W2[0], HID, p.in_dim, false, p.grp_L, p.grp_F);
        stage_weight(WLs2, RS, 32, 64, p.W2[1], p.out2_dim, HID, true);
        for (int e = threadIdx.x; e < 64; e += blockDim.x) b0s2[e] = p.b2[0][e];
        for (int e = threadIdx.x; e < 32; e += blockDim.x) bLs2[e] = e < p.out2_dim ? p.b2[1][e] : 0.0f;
    }
    for (int e = threadIdx.x; e < 64; e += blockDim.x) {
        b0s[e] = p.b[0][e];
        b1s[e] = p.b[1][e];
    }
    // padding channels: a bias of -1e30 makes their exponential exactly 0 and never the maximum
    for (int e = threadIdx.x; e < OB * 32; e += blockDim.x) bLs[e] = e < p.out_dim ? p.b[2][e] : -1e30f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int r = lane & 31, h = lane >> 5;
    const int64_t M = p.M;
    const auto rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x1), 0, (int)(M * 128), 0x00020000);
    const auto rs_st = __builtin_amdgcn_make_buffer_rsrc(p.stats, 0, (int)(M * 8), 0x00020000);
    const auto rs_o2 = __builtin_amdgcn_make_buffer_rsrc(PAIR ? p.out2 : nullptr, 0, (int)(M * (PAIR ? p.out2_dim : 0) * 2), 0x00020000);
    const auto rs_h1 = __builtin_amdgcn_make_buffer_rsrc(p.hsave[1], 0, (int)(M * HID * 2), 0x00020000);
    const int pw = c.per_wave;
    const int64_t pk0 = pw ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x, pk_step = pw ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x;
    const int t_first = pw ? 0 : wave, t_step = pw ? 1 : 4;
    // One wave per SIMD: a round trip to memory that is not already in flight is paid in full.  The next tile's features and weights are requested at the
    // top of every tile - across the end of a ray too: the first tile of the wave's NEXT ray is requested during the last tile of the current one
    // (its sample range was read a whole ray earlier).
    // (straight-line: every request is issued unconditionally with an out-of-range offset for lanes - or whole tiles - that do not exist; a branch
    // around the loads would make the compiler drain vmcnt at the join, i.e. wait for the prefetch right where it was issued)
    bf16x8 xn[4];
    float wn = 0.0f;
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(c.weights), 0, (int)(M * 4), 0x00020000);
    unsigned xoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xoff[s] = (unsigned)(((int64_t)(2 * s + h) * M + r) * 16);
    auto fetch = [&](int64_t fbeg, int64_t fend, int64_t t) __attribute__((always_inline)) {
        const int64_t base = fbeg + 32 * t;
        const bool lv = base + r < fend;                      // false for every lane when the tile does not exist: zeros, weight 0
        const unsigned off = lv ? (unsigned)base * 16u : BUF_OOB;
#pragma unroll
        for (int s = 0; s < 4; ++s) xn[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x1, xoff[s] + off, 0, 0));
        wn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, lv ? (unsigned)(base + r) * 4u : BUF_OOB, 0, 0));
    };
    int64_t beg = 0, end = 0;
    if (pk0 < c.P) {
        beg = c.pack_start[pk0];
        end = c.pack_start[pk0 + 1];
        fetch(beg, end, t_first);
    }
    for (int64_t pk = pk0; pk < c.P; pk += pk_step) {
        asm volatile("" : "+v"(r), "+v"(h));
        const int64_t ntile = (end - beg + 31) / 32;
        int64_t nbeg = 0, nend = 0;
        if (pk + pk_step < c.P) {
            nbeg = c.pack_start[pk + pk_step];
            nend = c.pack_start[pk + pk_step + 1];
        }
        unsigned ooff2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ooff2[j] = (PAIR && 4 * h + j < p.out2_dim) ? (unsigned)((r * p.out2_dim + 4 * h + j) * 2) : BUF_OOB;
        f32x16 acc[7];
#pragma unroll
        for (int ob = 0; ob < 7; ++ob)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ob][q] = 0.0f;
        if (t_first >= ntile) fetch(nbeg, nend, t_first);      // no tile of this ray for this wave (wave-uniform): keep the chain of requests going
        for (int64_t t = t_first; t < ntile; t += t_step) {
            const int64_t base = beg + 32 * t;
            const bool live = base + r < end;
            const int rv = (int)min((int64_t)32, end - base);
            const unsigned row0 = (unsigned)base;
            bf16x8 xb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) xb[s] = xn[s];
            const float wcur = wn;
            {
                const bool same = t + t_step < ntile;
                fetch(same ? beg : nbeg, same ? end : nend, same ? t + t_step : (int64_t)t_first);
            }
            f32x16 a2[2];
            bf16x8 hb[4];
            if constexpr (PAIR) {      // the companion head: hidden layer, <= 8 logits, softmax (as mlp_fwd_wide_stats<.., true>)
                f32x16 o2;
                hidden_layer_pinned<4>(W0s2, b0s2, xb, r, h, a2);
                relu_pack(a2, hb);
                out_block_pinned(WLs2, bLs2, 0, hb, r, h, o2);
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * h + j < p.out2_dim) mx = fmaxf(mx, o2[j]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mxs = mx * LOG2E;
                float e[4], sum = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[j] = (4 * h + j < p.out2_dim) ? __builtin_amdgcn_exp2f(fmaf(o2[j], LOG2E, -mxs)) : 0.0f;
                    sum += e[j];
                }
                sum += __shfl_xor(sum, 32);
                const float inv2 = 1.0f / sum;
                const unsigned obase = live ? row0 * (unsigned)(p.out2_dim * 2) : BUF_OOB_ROW;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16_t y = (bf16_t)(e[j] * inv2);
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, y), rs_o2, ooff2[j] + obase, 0, 0);
                }
            }
            hidden_layer_pinned<4>(W0s, b0s, xb, r, h, a2);
            relu_pack(a2, hb);
            hidden_layer_pinned<4>(W1s, b1s, hb, r, h, a2);
            relu_pack(a2, hb);
            tile64_store_buf_rows(stg, rs_h1, row0 * (HID * 2), rv, lane, r, h, a2);
            // all seven logit blocks, then ONE pass: maximum, exponentials, sum
            f32x16 o[7];
#pragma unroll
            for (int ob = 0; ob < 7; ++ob) out_block_pinned(WLs, bLs, ob, hb, r, h, o[ob]);
            float mx = o[0][0];
#pragma unroll
            for (int ob = 0; ob < 7; ++ob)
#pragma unroll
                for (int q = 0; q < 16; ++q) mx = fmaxf(mx, o[ob][q]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float Ms = mx * LOG2E;
            const f32x2 l2 = {LOG2E, LOG2E}, nm2 = {-Ms, -Ms};
            f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
            for (int ob = 0; ob < 7; ++ob)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x2 tt = f32x2{o[ob][2 * q], o[ob][2 * q + 1]} * l2 + nm2;
                    const f32x2 ee = {__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])};
                    o[ob][2 * q] = ee[0];
                    o[ob][2 * q + 1] = ee[1];
                    s2 += ee;
                }
            float sm = s2[0] + s2[1];
            sm += __shfl_xor(sm, 32);
            const float inv = 1.0f / sm;
            const u32x2 st2 = {__builtin_bit_cast(unsigned, Ms), __builtin_bit_cast(unsigned, inv)};
            __builtin_amdgcn_raw_buffer_store_b64(st2, rs_st, (live && h == 0) ? row0 * 8u + (unsigned)(r * 8) : BUF_OOB, 0, 0);
            const float sw = inv * wcur;      // 0 for lanes past the ray's end
#pragma unroll
            for (int ob = 0; ob < 7; ++ob)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ob][q] = fmaf(sw, o[ob][q], acc[ob][q]);
        }
        // ---- sum over the 32 sample lanes of each half, then over the four waves (as head_composite_fwd_kernel)
#pragma unroll
        for (int ob = 0; ob < 7; ++ob) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = acc[ob][q];
#define PAG_DPP_ADD16(ctl)                                                                                                         \
            asm("s_nop 1\n\t"                                                                                                   \
                "v_add_f32_dpp %0, %0, %0 " ctl "\n\tv_add_f32_dpp %1, %1, %1 " ctl "\n\tv_add_f32_dpp %2, %2, %2 " ctl "\n\t"   \
                "v_add_f32_dpp %3, %3, %3 " ctl "\n\tv_add_f32_dpp %4, %4, %4 " ctl "\n\tv_add_f32_dpp %5, %5, %5 " ctl "\n\t"   \
                "v_add_f32_dpp %6, %6, %6 " ctl "\n\tv_add_f32_dpp %7, %7, %7 " ctl "\n\tv_add_f32_dpp %8, %8, %8 " ctl "\n\t"   \
                "v_add_f32_dpp %9, %9, %9 " ctl "\n\tv_add_f32_dpp %10, %10, %10 " ctl "\n\tv_add_f32_dpp %11, %11, %11 " ctl "\n\t" \
                "v_add_f32_dpp %12, %12, %12 " ctl "\n\tv_add_f32_dpp %13, %13, %13 " ctl "\n\tv_add_f32_dpp %14, %14, %14 " ctl "\n\t" \
                "v_add_f32_dpp %15, %15, %15 " ctl                                                                             \
                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),     \
                  "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]))
            PAG_DPP_ADD16("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
            PAG_DPP_ADD16("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
            PAG_DPP_ADD16("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
            PAG_DPP_ADD16("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
            PAG_DPP_ADD16("row_bcast:15 row_mask:0xa bank_mask:0xf");
#undef PAG_DPP_ADD16
            if (r == 31) {
#pragma unroll
                for (int q = 0; q < 16; ++q) red[wave * OB * 32 + 32 * ob + rho(q, h)] = v[q];
            }
        }
        const int32_t ray = c.ray_of_pack[pk];
        const float al = c.alpha[ray];
        if (pw) {           // this wave's own section of `red`: LDS accesses of one wave complete in program order, no barrier
            for (int cc = lane; cc < p.out_dim; cc += 64) c.out[(int64_t)ray * p.out_dim + cc] = al * red[wave * OB * 32 + cc];
        } else {
            __syncthreads();
            for (int cc = threadIdx.x; cc < p.out_dim; cc += blockDim.x)
                c.out[(int64_t)ray * p.out_dim + cc] = al * (red[cc] + red[OB * 32 + cc] + red[2 * OB * 32 + cc] + red[3 * OB * 32 + cc]);
            __syncthreads();
        }
        beg = nbeg;
        end = nend;
    }
    // Samples past the last pack (the filler samples of a padded batch, pagnerf_amd/graphs.py) belong to no ray: what the backward reads of them - always
    // scaled by a zero weight - must be finite.  statistics (huge maximum, 1 / sum = 0): rebuilt probabilities are exactly 0; hidden layer and the
    // companion head's output: zeros.
    for (int64_t row = c.pack_start[c.P] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < M; row += (int64_t)gridDim.x * blockDim.x) {
        *reinterpret_cast<float2 *>(p.stats + 2 * row) = float2{1e30f, 0.0f};
        u32x4 *hrow = reinterpret_cast<u32x4 *>(reinterpret_cast<bf16_t *>(p.hsave[1]) + row * HID);
#pragma unroll
        for (int k = 0; k < 8; ++k) hrow[k] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (PAIR) {
            bf16_t *orow = reinterpret_cast<bf16_t *>(p.out2) + row * p.out2_dim;
            for (int k = 0; k < p.out2_dim; ++k) orow[k] = (bf16_t)0.0f;
        }
    }
}

// ---------------------------------------------------------- one affine map of the XCD8 features (decoder without activations)
// pc_nerf/panoptic_dd_nef.py:49-56 `decoder_delta_density` has no activation: any number of its layers compose to one [n_out, in_dim]
// matrix (n_out = 1).  out[m][o] = b[o] + sum_p W[o][col(p)] x[p / 8][m][p % 8] on the encoders' bf16 [8][M][8] layout, one lane per
// sample, 128 bytes in / 4 n_out bytes out per sample; backward-data is the transposed product, written back in the same layout.
// (Weight gradients: pag_mlp_wgrad_batch with the upstream gradient as `dz`.)
constexpr int AFF_MAX_OUT = 8;
__global__ __launch_bounds__(256) void affine_xcd8_fwd_kernel(const bf16_t *__restrict__ x8, int64_t M, int grp_L, int grp_F, const float *__restrict__ W,
                                                              const float *__restrict__ b, int n_out, int in_dim, float *__restrict__ out) {
    __shared__ float Ws[AFF_MAX_OUT][64];
    for (int e = threadIdx.x; e < n_out * 64; e += blockDim.x) {
        const int o = e >> 6, col = grp_col(e & 63, grp_L, grp_F);
        Ws[o][e & 63] = (col >= 0 && col < in_dim) ? W[(int64_t)o * in_dim + col] : 0.0f;
    }
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float acc[AFF_MAX_OUT];
#pragma unroll
    for (int o = 0; o < AFF_MAX_OUT; ++o) acc[o] = o < n_out ? b[o] : 0.0f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x8 + ((int64_t)g * M + m) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = (float)v[e];
#pragma unroll
            for (int o = 0; o < AFF_MAX_OUT; ++o)
                if (o < n_out) acc[o] = fmaf(xv, Ws[o][8 * g + e], acc[o]);
        }
    }
    for (int o = 0; o < n_out; ++o) out[m * n_out + o] = acc[o];
}
__global__ __launch_bounds__(256) void affine_xcd8_bwd_dx_kernel(const float *__restrict__ gout, int64_t M, int grp_L, int grp_F, const float *__restrict__ W,
                                                                 int n_out, int in_dim, bf16_t *__restrict__ dx8) {
    __shared__ float Ws[AFF_MAX_OUT][64];
    for (int e = threadIdx.x; e < n_out * 64; e += blockDim.x) {
        const int o = e >> 6, col = grp_col(e & 63, grp_L, grp_F);
        Ws[o][e & 63] = (col >= 0 && col < in_dim) ? W[(int64_t)o * in_dim + col] : 0.0f;
    }
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float g[AFF_MAX_OUT];
#pragma unroll
    for (int o = 0; o < AFF_MAX_OUT; ++o) g[o] = o < n_out ? gout[m * n_out + o] : 0.0f;
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.0f;
#pragma unroll
            for (int o = 0; o < AFF_MAX_OUT; ++o)
                if (o < n_out) a = fmaf(g[o], Ws[o][8 * gp + e], a);
            v[e] = (bf16_t)a;
        }
        *reinterpret_cast<bf16x8 *>(dx8 + ((int64_t)gp * M + m) * 8) = v;
    }
}

// ------------------------------------------------------------------------------------ FP32 parity path
// One lane per sample.  Weights transposed in LDS ([k][j]) so the 64 outputs of a layer are 16
// broadcast ds_read_b128; the per-sample activation column lives in LDS ([k][lane]).
constexpr int PT = 128;   // threads per block on this path

struct F32Fwd {
    FwdParams p;
    float *hsave32[2];
};

__device__ void stage_f32_t(float *dst, const float *W, int n_out, int n_in, int out_pad) {   // dst[k][out_pad]
    for (int e = threadIdx.x; e < n_in * out_pad; e += blockDim.x) {
        int k = e / out_pad, j = e - k * out_pad;
        dst[e] = j < n_out ? W[(int64_t)j * n_in + k] : 0.0f;
    }
}

template <typename X1T, typename OutT>
__global__ __launch_bounds__(PT) void mlp_fwd_f32(FwdParams p, int n_layers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *xs = reinterpret_cast<float *>(smem);          // [64][PT] activation columns
    float *wt = xs + 64 * PT;                             // current layer's W^T [k][out_pad]
    const int t = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * PT + t;
    const bool live = m < p.M;
    const int64_t mc = live ? m : p.M - 1;
    const X1T *x1 = reinterpret_cast<const X1T *>(p.x1);
    OutT *out = reinterpret_cast<OutT *>(p.out);
    // input column
    const int32_t ray = p.x2 ? p.x2_index[mc] : 0;
    for (int k = 0; k < p.in_dim; ++k)
        xs[k * PT + t] = k < p.k1 ? pag_ld(x1 + mc * p.k1 + k) : p.x2[(int64_t)ray * p.k2p + (k - p.k1)];
    int n_in = p.in_dim;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = l == n_layers - 1;
        const int n_out = last ? p.out_dim : HID;
        const int chunks = (n_out + 63) / 64;
        for (int c = 0; c < chunks; ++c) {
            __syncthreads();
            const int rows = min(64, n_out - 64 * c);
            stage_f32_t(wt, p.W[l] + (int64_t)64 * c * n_in, rows, n_in, 64);
            __syncthreads();
            float acc[64];
#pragma unroll
            for (int j = 0; j < 64; ++j) acc[j] = j < rows ? p.b[l][64 * c + j] : 0.0f;
            for (int k = 0; k < n_in; ++k) {
                const float xk = xs[k * PT + t];
                const f32x4 *wrow = reinterpret_cast<const f32x4 *>(wt + k * 64);
#pragma unroll
                for (int j4 = 0; j4 < 16; ++j4) {
                    f32x4 w = wrow[j4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * j4 + j] = fmaf(w[j], xk, acc[4 * j4 + j]);
                }
            }
            if (!last) {
                __syncthreads();   // everyone finished reading xs of the previous layer
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    float v = fmaxf(acc[j], 0.0f);
                    xs[j * PT + t] = v;
                    if (p.hsave[l] && live) reinterpret_cast<float *>(p.hsave[l])[m * HID + j] = v;
                }
            } else if (live) {
#pragma unroll
                for (int j = 0; j < 64; ++j)
                    if (j < rows) {
                        float v = acc[j];
                        if (p.act == PAG_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                        pag_st(out + m * p.out_dim + 64 * c + j, v);
                    }
            }
        }
        n_in = HID;
    }
    if (p.act == PAG_ACT_SOFTMAX && live) {   // second pass over this sample's logits
        OutT *row = out + m * p.out_dim;
        float mx = -INFINITY;
        for (int j = 0; j < p.out_dim; ++j) mx = fmaxf(mx, pag_ld(row + j));
        float sum = 0.0f;
        for (int j = 0; j < p.out_dim; ++j) sum += expf(pag_ld(row + j) - mx);
        for (int j = 0; j < p.out_dim; ++j) pag_st(row + j, expf(pag_ld(row + j) - mx) / sum);
    }
}

__device__ void stage_f32_n(float *dst, const float *W, int row0, int rows, int n_in) {   // dst[j][64] = W[row0+j][0:n_in]
    for (int e = threadIdx.x; e < rows * 64; e += blockDim.x) {
        int j = e / 64, k = e - j * 64;
        dst[e] = k < n_in ? W[(int64_t)(row0 + j) * n_in + k] : 0.0f;
    }
}

template <typename OutT, typename DxT>
__global__ __launch_bounds__(PT) void mlp_bwd_f32(BwdParams p, int n_layers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *zs = reinterpret_cast<float *>(smem);          // [224][PT] dz column of the current layer
    float *wt = zs + 224 * PT;                            // [64 rows j][64 k]
    const int t = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * PT + t;
    const bool live = m < p.M;
    const int64_t mc = live ? m : p.M - 1;
    const OutT *y = reinterpret_cast<const OutT *>(p.out) + mc * p.out_dim;
    const OutT *g = reinterpret_cast<const OutT *>(p.grad_out) + mc * p.out_dim;
    float dot = 0.0f;
    if (p.act == PAG_ACT_SOFTMAX)
        for (int j = 0; j < p.out_dim; ++j) dot += pag_ld(g + j) * pag_ld(y + j);
    for (int j = 0; j < p.out_dim; ++j) {
        float v = pag_ld(g + j);
        if (p.act == PAG_ACT_SIGMOID) {
            float yy = pag_ld(y + j);
            v = v * yy * (1.0f - yy);
        } else if (p.act == PAG_ACT_SOFTMAX) {
            v = pag_ld(y + j) * (v - dot);
        }
        v = live ? v : 0.0f;
        zs[j * PT + t] = v;
        if (live) reinterpret_cast<float *>(p.dz[n_layers - 1])[m * p.out_dim + j] = v;
    }
    int n_out = p.out_dim;
    for (int l = n_layers - 1; l >= 0; --l) {
        const int n_in = l == 0 ? p.in_dim : HID;
        if (l == 0 && !p.dx1) break;
        float acc[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) acc[k] = 0.0f;
        const int chunks = (n_out + 63) / 64;
        for (int c = 0; c < chunks; ++c) {
            __syncthreads();
            const int rows = min(64, n_out - 64 * c);
            stage_f32_n(wt, p.W[l], 64 * c, rows, n_in);
            __syncthreads();
            for (int j = 0; j < rows; ++j) {
                const float zj = zs[(64 * c + j) * PT + t];
                const f32x4 *wrow = reinterpret_cast<const f32x4 *>(wt + j * 64);
#pragma unroll
                for (int k4 = 0; k4 < 16; ++k4) {
                    f32x4 w = wrow[k4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[4 * k4 + k] = fmaf(w[k], zj, acc[4 * k4 + k]);
                }
            }
        }
        __syncthreads();
        if (l > 0) {
            const float *hv = reinterpret_cast<const float *>(p.hsave[l - 1]) + mc * HID;
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                float v = (live && hv[k] > 0.0f) ? acc[k] : 0.0f;
                zs[k * PT + t] = v;
                if (live) reinterpret_cast<float *>(p.dz[l - 1])[m * HID + k] = v;
            }
            n_out = HID;
        } else if (live) {
            DxT *dx = reinterpret_cast<DxT *>(p.dx1) + m * p.k1;
#pragma unroll
            for (int k = 0; k < 64; ++k)
                if (k < p.k1) pag_st(dx + k, acc[k]);
        }
    }
}


// ------------------------------------------------------------------------------- weight gradients
// dW[out][in] = sum_m dz[m][out] * a[m][in]  and  db[out] = sum_m dz[m][out]: a GEMM whose reduction
// runs over the M ~ 2e6 samples with both operands K-major in memory, which BLAS libraries handle
// badly (2.5 ms per layer measured).  Here each workgroup walks 64-sample chunks: the [64 x n_out] dz
// tile and the [64 x n_in] input tile are transposed into LDS (lane = sample, so the ds_write_b16
// stores are conflict-free), every wave owns up to 6 of the 32x32 (out-block, in-block) pairs and
// feeds them with ds_read_b128 fragments; an extra in-block whose B fragment is the constant
// "1 in column 0" yields db for free.  Partial sums are written as per-workgroup fp32 slabs
// [blocks][OB*32][96] (cols 0..63 = dW, col 64 = db) and summed by the caller - deterministic, no atomics.
struct WgradParams {
    const bf16_t *dz;
    int dz_cols, n_out;
    const void *a1;
    int k1;
    const float *a2;
    int k2p;
    const int32_t *a2_index;
    int n_in;
    float *slabs;
    int64_t M;
    int a1_grouped;        // a1 is bf16 [8][M][8] (PAG_LAYOUT_XCD8); slab columns are then staged positions
};
constexpr int WG_MAX_BATCH = 6;
struct WgradBatch {
    WgradParams p[WG_MAX_BATCH];
};
constexpr int WG_RS = 72;       // LDS row stride (bf16) of the transposed tiles: 64 samples + 8 pad
constexpr int WG_SLAB_COLS = 96;

template <typename A1T, int APW /* accumulator blocks per wave */, int NWV = 4 /* waves per workgroup */>
// narrow variant (APW 2, 4 waves): asking for 5 waves per SIMD keeps every accumulator in VGPRs (no AGPR copies) under 102
// registers.  Wide layers (up to 224 outputs = 21 block pairs): 8 waves x 3 pairs instead of 4 x 6 - 48 accumulator
// registers per wave leave room for the prefetch and for 4 waves per SIMD (the 4 x 6 form ran 2 waves per SIMD, no prefetch).
__global__ __launch_bounds__(NWV * 64, (APW == 2 ? 5 : (APW == 3 ? 4 : 1))) void mlp_wgrad_kernel(WgradBatch batch) {
    PAG_BLOCK_TIMER(6);
    const WgradParams &p = batch.p[blockIdx.y];       // blockIdx.y = layer: the layers of one decoder share a launch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OB = (p.n_out + 31) / 32;
    const int IB = (p.n_in + 31) / 32;                 // 1 or 2
    bf16_t *Zt = reinterpret_cast<bf16_t *>(smem);      // [OB*32][WG_RS]
    bf16_t *At = Zt + OB * 32 * WG_RS;                  // [IB*32][WG_RS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int npairs = OB * (IB + 1);
    f32x16 acc[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.0f;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)(r == 0 ? 1.0f : 0.0f);
    const A1T *a1 = reinterpret_cast<const A1T *>(p.a1);
    const bool dz_vec = (p.dz_cols % 8) == 0 && (p.n_out % 8) == 0;
    const int64_t nchunks = (p.M + 63) / 64;
    // global -> register fetch of one 8-column piece of this lane's sample row (dz tile / input tile)
    auto fetch_z = [&](int64_t chunk, int cg) __attribute__((always_inline)) {
        const int64_t m = chunk * 64 + lane;
        const bool live = m < p.M;
        const int64_t mc = live ? m : p.M - 1;
        const int c0 = 8 * cg;
        bf16x8 v = zero8();
        if (live && c0 < p.n_out) {      // n_out <= dz_cols: dz may point at a band of columns of a wider row
            if (dz_vec) {
                v = load8(p.dz + mc * p.dz_cols + c0);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + j < p.n_out) v[j] = p.dz[mc * p.dz_cols + c0 + j];
            }
        }
        return v;
    };
    auto fetch_a = [&](int64_t chunk, int cg) __attribute__((always_inline)) {
        const int64_t m = chunk * 64 + lane;
        const bool live = m < p.M;
        const int64_t mc = live ? m : p.M - 1;
        const int c0 = 8 * cg;
        bf16x8 v = zero8();
        if (live && p.a1_grouped)
            v = load8(reinterpret_cast<const bf16_t *>(p.a1) + ((int64_t)cg * p.M + mc) * 8);
        else if (live && c0 < p.k1)
            v = load8(a1 + mc * p.k1 + c0);
        else if (live && p.a2 && c0 < p.k1 + p.k2p)
            v = load8(p.a2 + (int64_t)p.a2_index[mc] * p.k2p + (c0 - p.k1));
        return v;
    };
    // APW == 2 (<= 64 x 64 layers, 76 VGPRs): the next chunk's four 16-byte pieces are fetched into registers while the
    // current chunk goes through LDS and the MFMAs - the kernel sat waiting on memory 77 % of its wave cycles (SQ_WAIT_ANY)
    // with nothing in flight between the two barriers.  The wide variant has no registers to spare for this.
    constexpr bool PF = APW <= 3;
    constexpr int ZG = (NWV == 8) ? 4 : 2, AG = (NWV == 8) ? 1 : 2;      // 16-byte pieces per wave: dz (<= 224 / 64 cols), input (64 cols)
    bf16x8 pz[ZG], pa[AG];
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < ZG; ++i) {
            const int cg = wave + NWV * i;
            pz[i] = (blockIdx.x < nchunks && cg < OB * 4) ? fetch_z(blockIdx.x, cg) : zero8();
        }
#pragma unroll
        for (int i = 0; i < AG; ++i) {
            const int cg = wave + NWV * i;
            pa[i] = (blockIdx.x < nchunks && cg < IB * 4) ? fetch_a(blockIdx.x, cg) : zero8();
        }
    }
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // ---- dz tile, transposed: Zt[col][sample]
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < ZG; ++i) {
                const int cg = wave + NWV * i, c0 = 8 * cg;
                if (cg < OB * 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) Zt[(c0 + j) * WG_RS + lane] = pz[i][j];
                }
            }
#pragma unroll
            for (int i = 0; i < AG; ++i) {
                const int cg = wave + NWV * i, c0 = 8 * cg;
                if (cg < IB * 4) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) At[(c0 + j) * WG_RS + lane] = (c0 + j < p.n_in) ? pa[i][j] : (bf16_t)0.0f;
                }
            }
        } else {
            for (int cg = wave; cg < OB * 4; cg += NWV) {
                const int c0 = 8 * cg;
                const bf16x8 v = fetch_z(chunk, cg);
#pragma unroll
                for (int j = 0; j < 8; ++j) Zt[(c0 + j) * WG_RS + lane] = v[j];
            }
            // ---- input tile, transposed: At[col][sample]
            for (int cg = wave; cg < IB * 4; cg += NWV) {
                const int c0 = 8 * cg;
                const bf16x8 v = fetch_a(chunk, cg);
#pragma unroll
                for (int j = 0; j < 8; ++j) At[(c0 + j) * WG_RS + lane] = (c0 + j < p.n_in) ? v[j] : (bf16_t)0.0f;
            }
        }
        __syncthreads();
        if constexpr (PF) {
            const int64_t next = chunk + gridDim.x;
            if (next < nchunks) {
#pragma unroll
                for (int i = 0; i < ZG; ++i) {
                    const int cg = wave + NWV * i;
                    if (cg < OB * 4) pz[i] = fetch_z(next, cg);
                }
#pragma unroll
                for (int i = 0; i < AG; ++i) {
                    const int cg = wave + NWV * i;
                    if (cg < IB * 4) pa[i] = fetch_a(next, cg);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int pr = wave + NWV * i;
            if (pr < npairs) {
                const int ob = pr / (IB + 1), ib = pr - ob * (IB + 1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 a = *reinterpret_cast<const bf16x8 *>(Zt + (32 * ob + r) * WG_RS + 16 * ks + 8 * h);
                    bf16x8 b = ones;
                    if (ib < IB) b = *reinterpret_cast<const bf16x8 *>(At + (32 * ib + r) * WG_RS + 16 * ks + 8 * h);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float *slab = p.slabs + (int64_t)blockIdx.x * OB * 32 * WG_SLAB_COLS;
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int pr = wave + NWV * i;
        if (pr < npairs) {
            const int ob = pr / (IB + 1), ib = pr - ob * (IB + 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) slab[(32 * ob + rho(q, h)) * WG_SLAB_COLS + (ib < IB ? 32 * ib : 64) + r] = acc[i][q];
        }
    }
}

inline unsigned mlp_grid(int64_t M) {
    int64_t tiles = (M + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    int64_t cap = 256 * 6;   // a few workgroups per CU; tiles are grid-strided
    return (unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

}  // namespace

#define MLP_FWD_LAUNCH(X1T, OutT, NL_, OBM)                                                            \
    hipLaunchKernelGGL((mlp_fwd_mfma<X1T, OutT, NL_, OBM>), dim3(mlp_grid(M)), dim3(256), lds, st, p)
#define MLP_FWD_OB(X1T, OutT, NL_)                                   \
    do {                                                             \
        if (OB <= 1) MLP_FWD_LAUNCH(X1T, OutT, NL_, 1);              \
        else if (OB <= 2) MLP_FWD_LAUNCH(X1T, OutT, NL_, 2);         \
        else if (OB <= 4) MLP_FWD_LAUNCH(X1T, OutT, NL_, 4);         \
        else MLP_FWD_LAUNCH(X1T, OutT, NL_, 7);                      \
    } while (0)
#define MLP_FWD_NL(X1T, OutT)                                        \
    do {                                                             \
        if (a->n_layers == 2) MLP_FWD_OB(X1T, OutT, 2);              \
        else MLP_FWD_OB(X1T, OutT, 3);                               \
    } while (0)

extern "C" int pag_mlp_fwd_pair_supported(const pag_mlp_fwd_args *a, const pag_mlp_fwd_args *b) {
    if (!a || !b || b->pair) return 0;
    const bool wide = a->mode == PAG_MLP_MFMA_BF16 && a->x1_dtype == PAG_BF16 && a->x1_layout == PAG_LAYOUT_XCD8 && !a->out && a->n_layers == 3 &&
                      a->softmax_stats && a->out_act == PAG_ACT_SOFTMAX && a->out_dim > 192 && a->out_dim <= 224 && a->hidden_save[1] && !a->hidden_save[0];
    const bool narrow = b->mode == PAG_MLP_MFMA_BF16 && b->x1 == a->x1 && b->x1_dtype == PAG_BF16 && b->x1_layout == PAG_LAYOUT_XCD8 && b->x1_levels == a->x1_levels &&
                        b->x1_feats == a->x1_feats && b->in_dim == a->in_dim && b->n_layers == 2 && b->out && b->out_dtype == PAG_BF16 &&
                        b->out_act == PAG_ACT_SOFTMAX && b->out_dim >= 1 && b->out_dim <= 8 && !b->hidden_save[0] && !b->hidden_save[1] && !b->x2 &&
                        !b->x1_col0_relu && b->W[0] && b->W[1] && b->b[0] && b->b[1];
    return wide && narrow ? 1 : 0;
}

#ifndef PAG_FAST_FWD_GRID_CAP
#define PAG_FAST_FWD_GRID_CAP 768      // three workgroups per CU are resident: one round, tiles grid-strided (1536 ran two rounds with a ragged second: colour forward 63 -> 57 us)
#endif
extern "C" int pag_mlp_fwd_producer_supported(const pag_mlp_fwd_args *a, const pag_mlp_fwd_args *d, int64_t M) {
    static const bool no_fast = getenv("PAG_NO_FAST_FWD") != nullptr;
    if (!a || !d || no_fast || M < 1 || M > PAG_MLP_FUSED_WIDE_MAX_M) return 0;
    const bool density = d->mode == PAG_MLP_MFMA_BF16 && d->x1_dtype == PAG_BF16 && d->x1_layout == PAG_LAYOUT_XCD8 && d->k1 == 64 && !d->x2 && d->n_layers == 2 &&
                         d->out && d->out_dtype == PAG_BF16 && d->out_act == PAG_ACT_NONE && d->out_dim == 16 && !d->hidden_save[0] && !d->hidden_save[1] &&
                         !d->softmax_stats && !d->x1_col0_relu && !d->pair && !d->composite && !d->x1_producer && d->W[0] && d->W[1] && d->b[0] && d->b[1] &&
                         d->in_dim == d->x1_levels * d->x1_feats && d->x1_feats >= 1 && ((d->x1_levels + 7) / 8) * d->x1_feats <= 8;
    const bool colour = a->mode == PAG_MLP_MFMA_BF16 && a->x1_dtype == PAG_BF16 && a->x1_layout != PAG_LAYOUT_XCD8 && a->k1 == 16 && a->x1 == d->out && a->x2 &&
                        a->k2p == 32 && a->x2_index && a->in_dim <= 48 && a->n_layers == 3 && a->out && a->out_dtype == PAG_F32 && a->out_act == PAG_ACT_SIGMOID &&
                        a->out_dim <= 4 && a->x1_col0_relu && !a->hidden_save[0] && !a->hidden_save[1] && !a->pair && !a->composite;
    return density && colour ? 1 : 0;
}

extern "C" int pag_mlp_fwd_composite_supported(const pag_mlp_fwd_args *a, int64_t M) {
    static const bool no_fast = getenv("PAG_NO_FAST_FWD") != nullptr;
    if (!a || no_fast || M < 1 || M > PAG_MLP_FUSED_WIDE_MAX_M) return 0;
    const bool wide = a->mode == PAG_MLP_MFMA_BF16 && a->x1_dtype == PAG_BF16 && a->x1_layout == PAG_LAYOUT_XCD8 && !a->out && a->n_layers == 3 &&
                      a->softmax_stats && a->out_act == PAG_ACT_SOFTMAX && a->out_dim > 192 && a->out_dim <= 224 && a->hidden_save[1] && !a->hidden_save[0];
    return wide && (!a->pair || pag_mlp_fwd_pair_supported(a, a->pair) == 1) ? 1 : 0;
}

extern "C" int pag_mlp_fwd(const pag_mlp_fwd_args *a, int64_t M, void *stream) {
    PAG_CHECK_ARG(a, "pag_mlp_fwd: args is NULL");
    PAG_CHECK_ARG(M >= 0, "pag_mlp_fwd: M < 0");
    PAG_CHECK_ARG(a->n_layers == 2 || a->n_layers == 3, "pag_mlp_fwd: n_layers %d not in {2,3}", a->n_layers);
    PAG_CHECK_ARG(a->k1 > 0 && a->k1 % 8 == 0, "pag_mlp_fwd: k1 %d must be a positive multiple of 8", a->k1);
    PAG_CHECK_ARG(a->x2 == nullptr || (a->k2p > 0 && a->k2p % 8 == 0 && a->x2_index), "pag_mlp_fwd: x2 needs k2p %% 8 == 0 and x2_index");
    const int k2p = a->x2 ? a->k2p : 0;
    PAG_CHECK_ARG(a->in_dim > 0 && a->in_dim <= a->k1 + k2p && a->in_dim <= 64 && a->k1 + k2p <= 64,
                  "pag_mlp_fwd: in_dim %d / k1+k2p %d out of range (<= 64)", a->in_dim, a->k1 + k2p);
    PAG_CHECK_ARG(a->out_dim >= 1 && a->out_dim <= 224, "pag_mlp_fwd: out_dim %d not in [1,224]", a->out_dim);
    PAG_CHECK_ARG(a->x1_dtype == PAG_F32 || a->x1_dtype == PAG_BF16, "pag_mlp_fwd: x1 dtype must be F32 or BF16");
    PAG_CHECK_ARG(a->out_dtype == PAG_F32 || a->out_dtype == PAG_BF16, "pag_mlp_fwd: out dtype must be F32 or BF16");
    PAG_CHECK_ARG(a->out_act >= PAG_ACT_NONE && a->out_act <= PAG_ACT_SOFTMAX, "pag_mlp_fwd: bad out_act %d", a->out_act);
    PAG_CHECK_ARG(a->mode == PAG_MLP_MFMA_BF16 || a->mode == PAG_MLP_FP32, "pag_mlp_fwd: bad mode %d", a->mode);
    if (M == 0) return PAG_OK;
    for (int l = 0; l < a->n_layers; ++l) PAG_CHECK_ARG(a->W[l] && a->b[l], "pag_mlp_fwd: NULL weight/bias of layer %d", l);
    PAG_CHECK_ARG(a->x1, "pag_mlp_fwd: NULL x1");
    PAG_CHECK_ARG(a->out || (a->softmax_stats && a->mode == PAG_MLP_MFMA_BF16 && a->out_dim > 64 && a->out_act == PAG_ACT_SOFTMAX),
                  "pag_mlp_fwd: NULL out (allowed only for wide softmax heads that write softmax_stats)");
    FwdParams p;
    for (int l = 0; l < 3; ++l) p.b[l] = a->b[l];
    p.x1 = a->x1;
    p.x2 = a->x2;
    p.x2_index = a->x2_index;
    p.k1 = a->k1;
    p.k2p = k2p;
    p.in_dim = a->in_dim;
    p.in_pad = ((a->k1 + k2p + 15) / 16) * 16;
    p.out_dim = a->out_dim;
    p.act = a->out_act;
    p.stats = (a->mode == PAG_MLP_MFMA_BF16 && a->out_dim > 64) ? a->softmax_stats : nullptr;
    PAG_CHECK_ARG(!a->x1_col0_relu || (a->mode == PAG_MLP_MFMA_BF16 && a->x1_dtype == PAG_BF16 && a->x1_layout != PAG_LAYOUT_XCD8 && a->out_dim <= 64),
                  "pag_mlp_fwd: x1_col0_relu needs MFMA mode, a strided bf16 x1 and out_dim <= 64");
    p.col0_relu = a->x1_col0_relu;
    p.W2[0] = p.W2[1] = p.b2[0] = p.b2[1] = nullptr;
    p.out2 = nullptr;
    p.out2_dim = 0;
    for (int l = 0; l < 3; ++l) {
        p.W[l] = l < a->n_layers ? a->W[l] : nullptr;
        p.b[l] = l < a->n_layers ? a->b[l] : nullptr;
    }
    p.out = a->out;
    p.hsave[0] = a->hidden_save[0];
    p.hsave[1] = a->n_layers == 3 ? a->hidden_save[1] : nullptr;
    p.M = M;
    p.grp_L = a->x1_layout == PAG_LAYOUT_XCD8 ? a->x1_levels : 0;
    p.grp_F = a->x1_feats;
    if (p.grp_L) {
        PAG_CHECK_ARG(a->mode == PAG_MLP_MFMA_BF16 && a->x1_dtype == PAG_BF16 && a->k1 == 64 && a->x2 == nullptr && p.grp_F >= 1 &&
                          ((p.grp_L + 7) / 8) * p.grp_F <= 8 && a->in_dim == p.grp_L * p.grp_F,
                      "pag_mlp_fwd: XCD8 input needs MFMA mode, bf16, k1 = 64, no x2 and in_dim = levels*feats");
        p.in_pad = 64;
    }
    hipStream_t st = (hipStream_t)stream;
    // ---- the panoptic nef's decoder shapes on the bf16 path: dedicated straight-line kernels (PAG_NO_FAST_FWD: the generic one, for A/B runs)
    static const bool no_fast = getenv("PAG_NO_FAST_FWD") != nullptr;
    if (!no_fast && a->mode == PAG_MLP_MFMA_BF16 && a->x1_dtype == PAG_BF16 && M <= PAG_MLP_FUSED_WIDE_MAX_M) {
        const bool grp = p.grp_L > 0;
        const bool save_all = a->hidden_save[0] && (a->n_layers == 2 || a->hidden_save[1]);
        const bool save_none = !a->hidden_save[0] && !a->hidden_save[1];
        int kind = -1;
        if (grp && a->out && a->out_dtype == PAG_BF16 && a->out_act == PAG_ACT_NONE && a->out_dim % 4 == 0 && a->out_dim <= 32) kind = 0;
        else if (!grp && a->k1 == 16 && a->x2 && a->k2p == 32 && a->in_dim <= 48 && a->out && a->out_dtype == PAG_F32 && a->out_act == PAG_ACT_SIGMOID &&
                 a->out_dim <= 4 && a->x1_col0_relu)
            kind = 1;
        else if (grp && a->out && a->out_dtype == PAG_BF16 && a->out_act == PAG_ACT_SOFTMAX && a->out_dim <= 8) kind = 2;
        else if (grp && !a->out && a->n_layers == 3 && p.stats && a->out_act == PAG_ACT_SOFTMAX && a->out_dim > 192 && a->hidden_save[1]) kind = 3;
        PAG_CHECK_ARG(!a->pair || (kind == 3 && pag_mlp_fwd_pair_supported(a, a->pair) == 1), "pag_mlp_fwd: pair is not supported for these arguments (pag_mlp_fwd_pair_supported)");
        PAG_CHECK_ARG(!a->composite || kind == 3, "pag_mlp_fwd: composite rides only in the statistics-only wide softmax head (pag_mlp_fwd_composite_supported)");
        if (kind == 3) {
            static bool attr = false;
            if (!attr) {
                hipFuncSetAttribute((const void *)mlp_fwd_wide_stats<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute((const void *)mlp_fwd_wide_stats<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipFuncSetAttribute((const void *)mlp_fwd_wide_stats<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr = true;
            }
            if (a->composite) {      // decoder + per-ray weighted sum in one pass over the logits (head_fwd_once_kernel)
                const pag_head_composite_args *hc = a->composite;
                PAG_CHECK_ARG(hc->P >= 0 && (hc->P == 0 || (hc->pack_start && hc->ray_of_pack && hc->weights && hc->alpha && hc->out)),
                              "pag_mlp_fwd: composite: NULL input/output");
                PAG_CHECK_ARG(!a->hidden_save[0] && a->hidden_save[1] && a->softmax_stats, "pag_mlp_fwd: composite needs softmax_stats and hidden_save[1] only");
                if (hc->P == 0) return PAG_OK;
                HeadCompParams c{hc->pack_start, hc->ray_of_pack, hc->P, nullptr, nullptr, nullptr, a->out_dim, nullptr, 0, hc->weights, hc->alpha, hc->out};
                c.per_wave = ((hc->n_samples > 0 && hc->n_samples < 160 * hc->P) || hc->P >= PAG_HC_PER_WAVE_P) ? 1 : 0;
                static bool attr_once = false;
                if (!attr_once) {
                    hipFuncSetAttribute((const void *)head_fwd_once_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    hipFuncSetAttribute((const void *)head_fwd_once_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr_once = true;
                }
#ifndef PAG_HEAD_ONCE_GRID
#define PAG_HEAD_ONCE_GRID 256
#endif
                const unsigned grid = (unsigned)std::min<int64_t>(c.per_wave ? (hc->P + 3) / 4 : hc->P, PAG_HEAD_ONCE_GRID);
                if (a->pair) {
                    const pag_mlp_fwd_args *b = a->pair;
                    p.W2[0] = b->W[0];
                    p.W2[1] = b->W[1];
                    p.b2[0] = b->b[0];
                    p.b2[1] = b->b[1];
                    p.out2 = b->out;
                    p.out2_dim = b->out_dim;
                    const size_t lds = (size_t)(128 + 224 + 96) * RS * sizeof(bf16_t) + (size_t)(128 + 224 + 96 + 4 * 224) * sizeof(float) + 4 * ST_BYTES;
                    hipLaunchKernelGGL((head_fwd_once_kernel<true>), dim3(grid), dim3(256), lds, st, p, c);
                } else {
                    const size_t lds = (size_t)(128 + 224) * RS * sizeof(bf16_t) + (size_t)(128 + 224 + 4 * 224) * sizeof(float) + 4 * ST_BYTES;
                    hipLaunchKernelGGL((head_fwd_once_kernel<false>), dim3(grid), dim3(256), lds, st, p, c);
                }
                PAG_CHECK_LAUNCH("pag_mlp_fwd (wide head, decoder + per-ray sum)");
                return PAG_OK;
            }
            if (a->pair) {
                const pag_mlp_fwd_args *b = a->pair;
                p.W2[0] = b->W[0];
                p.W2[1] = b->W[1];
                p.b2[0] = b->b[0];
                p.b2[1] = b->b[1];
                p.out2 = b->out;
                p.out2_dim = b->out_dim;
                constexpr int NWP = PAG_WIDE_FWD_PAIR_THREADS / 64;      // waves per workgroup (one workgroup per CU shares the 66 KiB of weights)
                const size_t lds = (size_t)(128 + 224 + 96) * RS * sizeof(bf16_t) + (128 + 224 + 96) * sizeof(float) + NWP * ST_BYTES;
                // one workgroup per CU is all that fits (LDS): a grid of 1.5 x 256 ran a full round and a half-empty one (220 us, of which
                // the second round's 110 us kept 128 CUs idle: scripts/block_timeline.py) - at most ONE round, tiles grid-strided
#ifndef PAG_WIDE_FWD_PAIR_GRID
#define PAG_WIDE_FWD_PAIR_GRID 256
#endif
                const unsigned grid = std::min<unsigned>((mlp_grid(M) * 4 + NWP - 1) / NWP, PAG_WIDE_FWD_PAIR_GRID);
                hipLaunchKernelGGL((mlp_fwd_wide_stats<false, true>), dim3(grid), dim3(PAG_WIDE_FWD_PAIR_THREADS), lds, st, p);
            } else {
                const size_t lds = (size_t)(128 + 224) * RS * sizeof(bf16_t) + (128 + 224) * sizeof(float) + 4 * ST_BYTES;
                if (a->hidden_save[0]) hipLaunchKernelGGL((mlp_fwd_wide_stats<true, false>), dim3(mlp_grid(M)), dim3(256), lds, st, p);
                else hipLaunchKernelGGL((mlp_fwd_wide_stats<false, false>), dim3(mlp_grid(M)), dim3(256), lds, st, p);
            }
            PAG_CHECK_LAUNCH("pag_mlp_fwd (wide head statistics)");
            return PAG_OK;
        }
        PAG_CHECK_ARG(!a->x1_producer || (kind == 1 && pag_mlp_fwd_producer_supported(a, a->x1_producer, M) == 1),
                      "pag_mlp_fwd: x1_producer is not supported for these arguments (pag_mlp_fwd_producer_supported)");
        if (a->x1_producer) {      // density decoder + colour decoder in one launch (mlp_fwd_density_colour)
            const pag_mlp_fwd_args *d = a->x1_producer;
            FwdParams pd = p;
            pd.x1 = d->x1;
            pd.x2 = nullptr;
            pd.x2_index = nullptr;
            pd.k1 = 64;
            pd.k2p = 0;
            pd.in_dim = d->in_dim;
            pd.in_pad = 64;
            pd.out_dim = d->out_dim;
            pd.act = d->out_act;
            pd.stats = nullptr;
            pd.col0_relu = nullptr;
            for (int l = 0; l < 3; ++l) {
                pd.W[l] = l < 2 ? d->W[l] : nullptr;
                pd.b[l] = l < 2 ? d->b[l] : nullptr;
            }
            pd.out = d->out;
            pd.hsave[0] = pd.hsave[1] = nullptr;
            pd.grp_L = d->x1_levels;
            pd.grp_F = d->x1_feats;
            const size_t lds = (size_t)(64 + 32 + 64 + 64 + 32) * RS * sizeof(bf16_t) + (64 + 32 + 64 + 64 + 32) * sizeof(float);
            hipLaunchKernelGGL(mlp_fwd_density_colour, dim3(std::min<unsigned>(mlp_grid(M), PAG_FAST_FWD_GRID_CAP)), dim3(256), lds, st, pd, p);
            PAG_CHECK_LAUNCH("pag_mlp_fwd (density + colour)");
            return PAG_OK;
        }
        if (kind >= 0 && (save_all || save_none)) {
            const size_t lds = (size_t)(64 + (a->n_layers == 3 ? 64 : 0) + 32) * RS * sizeof(bf16_t) + (128 + 32) * sizeof(float) + 4 * ST_BYTES;
#define FWD_FAST(NL_, K_, S_) hipLaunchKernelGGL((mlp_fwd_fast<NL_, K_, S_>), dim3(std::min<unsigned>(mlp_grid(M), PAG_FAST_FWD_GRID_CAP)), dim3(256), lds, st, p)
#define FWD_FAST_K(K_)                                                        \
    do {                                                                      \
        if (a->n_layers == 2) { if (save_all) FWD_FAST(2, K_, true); else FWD_FAST(2, K_, false); } \
        else { if (save_all) FWD_FAST(3, K_, true); else FWD_FAST(3, K_, false); }                 \
    } while (0)
            if (kind == 0) FWD_FAST_K(0);
            else if (kind == 1) FWD_FAST_K(1);
            else FWD_FAST_K(2);
#undef FWD_FAST_K
#undef FWD_FAST
            PAG_CHECK_LAUNCH("pag_mlp_fwd (straight-line)");
            return PAG_OK;
        }
    }
    PAG_CHECK_ARG(!a->composite, "pag_mlp_fwd: composite rides only in the straight-line wide-head launch (pag_mlp_fwd_composite_supported)");
    PAG_CHECK_ARG(!a->x1_producer, "pag_mlp_fwd: x1_producer rides only in the straight-line colour launch (pag_mlp_fwd_producer_supported)");
    PAG_CHECK_ARG(!a->pair, "pag_mlp_fwd: pair rides only in the straight-line wide-head launch (M <= %lld, PAG_NO_FAST_FWD unset)", (long long)PAG_MLP_FUSED_WIDE_MAX_M);
    if (a->mode == PAG_MLP_MFMA_BF16) {
        const int OB = (a->out_dim + 31) / 32;
        const size_t lds = (size_t)(64 + (a->n_layers == 3 ? 64 : 0) + OB * 32) * RS * sizeof(bf16_t) + (128 + OB * 32) * sizeof(float) + 4 * ST_BYTES;
        if (a->x1_dtype == PAG_F32 && a->out_dtype == PAG_F32) MLP_FWD_NL(float, float);
        else if (a->x1_dtype == PAG_F32) MLP_FWD_NL(float, bf16_t);
        else if (a->out_dtype == PAG_F32) MLP_FWD_NL(bf16_t, float);
        else MLP_FWD_NL(bf16_t, bf16_t);
    } else {
        const size_t lds = (size_t)(64 * PT + 64 * 64) * sizeof(float);
        dim3 grid((unsigned)((M + PT - 1) / PT)), block(PT);
        if (a->x1_dtype == PAG_F32 && a->out_dtype == PAG_F32)
            hipLaunchKernelGGL((mlp_fwd_f32<float, float>), grid, block, lds, st, p, a->n_layers);
        else if (a->x1_dtype == PAG_F32)
            hipLaunchKernelGGL((mlp_fwd_f32<float, bf16_t>), grid, block, lds, st, p, a->n_layers);
        else if (a->out_dtype == PAG_F32)
            hipLaunchKernelGGL((mlp_fwd_f32<bf16_t, float>), grid, block, lds, st, p, a->n_layers);
        else
            hipLaunchKernelGGL((mlp_fwd_f32<bf16_t, bf16_t>), grid, block, lds, st, p, a->n_layers);
    }
    PAG_CHECK_LAUNCH("pag_mlp_fwd");
    return PAG_OK;
}

#define MLP_BWD_LAUNCH(OutT, DxT, NL_, OBM)                                                            \
    hipLaunchKernelGGL((mlp_bwd_mfma<OutT, DxT, NL_, OBM>), dim3(mlp_grid(M)), dim3(256), lds, st, p)
#define MLP_BWD_OB(OutT, DxT, NL_)                                   \
    do {                                                             \
        if (OB <= 1) MLP_BWD_LAUNCH(OutT, DxT, NL_, 1);              \
        else if (OB <= 2) MLP_BWD_LAUNCH(OutT, DxT, NL_, 2);         \
        else if (OB <= 4) MLP_BWD_LAUNCH(OutT, DxT, NL_, 4);         \
        else MLP_BWD_LAUNCH(OutT, DxT, NL_, 7);                      \
    } while (0)
#define MLP_BWD_NL(OutT, DxT)                                        \
    do {                                                             \
        if (a->n_layers == 2) MLP_BWD_OB(OutT, DxT, 2);              \
        else MLP_BWD_OB(OutT, DxT, 3);                               \
    } while (0)

// Sum the per-workgroup slabs of pag_mlp_wgrad into the final dW [n_out][n_in] / db [n_out] (fixed order: deterministic).
// One workgroup per output row, 4 slab quarters x 96 columns; XCD8 inputs: slab column p is a staged position and lands in
// feature column grp_col(p).
constexpr int WF_SPLITS = 10;      // slab range split over 10 x 96 threads of the row's workgroup
struct FinishParams {
    const float *slabs;
    int n_blocks, n_out, rows_pad, n_in, grp_L, grp_F;
    float *dW, *db;
};
struct FinishBatch {
    FinishParams p[WG_MAX_BATCH];
};
__global__ __launch_bounds__(WF_SPLITS * WG_SLAB_COLS) void wgrad_finish_kernel(FinishBatch batch) {
    __shared__ float part[WF_SPLITS][WG_SLAB_COLS];
    const FinishParams &fp = batch.p[blockIdx.y];     // blockIdx.y = layer
    if ((int)blockIdx.x >= fp.n_out) return;
    const float *__restrict__ slabs = fp.slabs;
    float *__restrict__ dW = fp.dW, *__restrict__ db = fp.db;
    const int n_blocks = fp.n_blocks, rows_pad = fp.rows_pad, n_in = fp.n_in, grp_L = fp.grp_L, grp_F = fp.grp_F;
    const int o = blockIdx.x, c = threadIdx.x % WG_SLAB_COLS, q = threadIdx.x / WG_SLAB_COLS;
    const int64_t stride = (int64_t)rows_pad * WG_SLAB_COLS;
    const float *src = slabs + (int64_t)o * WG_SLAB_COLS + c;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int b = q;
    for (; b + 3 * WF_SPLITS < n_blocks; b += 4 * WF_SPLITS) {
        a0 += src[(int64_t)b * stride];
        a1 += src[(int64_t)(b + WF_SPLITS) * stride];
        a2 += src[(int64_t)(b + 2 * WF_SPLITS) * stride];
        a3 += src[(int64_t)(b + 3 * WF_SPLITS) * stride];
    }
    for (; b < n_blocks; b += WF_SPLITS) a0 += src[(int64_t)b * stride];
    part[q][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < WF_SPLITS; ++k) v += part[k][c];
        if (c == 64) {
            db[o] = v;
        } else if (c < 64) {
            const int col = grp_L ? grp_col(c, grp_L, grp_F) : c;
            if (col >= 0 && col < n_in) dW[(int64_t)o * n_in + col] = v;
        }
    }
}

static unsigned fused_grid(int64_t M) {
    const int64_t tiles = (M + 31) / 32;
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((tiles + 3) / 4, 256 * PAG_EXP_FUSED_WAVES));      // one 4-wave workgroup per CU, tiles grid-strided
}

// which mlp_bwd_fused instantiation serves these arguments: 0 density-like, 1 colour-like, 2 semantic-like, -1 none
static int fused_kind(const pag_mlp_bwd_args *a) {
    if (!a || a->mode != PAG_MLP_MFMA_BF16 || a->out_dim < 1 || a->out_dim > 224 || !a->dx1 || a->dx1_dtype != PAG_BF16) return -1;
    if (a->n_layers != 2 && a->n_layers != 3) return -1;
    if (!a->x1 || a->x1_dtype != PAG_BF16) return -1;
    const bool rank1 = a->g_ray != nullptr;
    if (a->out_dim > 32) {      // wide softmax head: stage A (output layer, mlp_bwd_wide_blocks) + stage B (the two layers below it)
        if (a->out_dim <= 192 || a->out_dim > 224 || a->n_layers != 3 || !a->g_ray || !a->g_scale || !a->g_index || !a->g_ray_scale) return -1;
        if (a->out_act != PAG_ACT_SOFTMAX || !a->softmax_stats || !a->b_last || a->out_dtype != PAG_BF16 || !a->dx1 || a->dx1_dtype != PAG_BF16) return -1;
        if (!a->x1 || a->x1_dtype != PAG_BF16 || a->x1_layout != PAG_LAYOUT_XCD8 || a->k1 != 64 || a->x2 || a->dx1_col0_add || a->dx1_accumulate) return -1;
        if (a->x1_levels < 1 || a->x1_feats < 1 || ((a->x1_levels + 7) / 8) * a->x1_feats > 8) return -1;
        const int j = 7 / a->x1_feats;
        if (j < (a->x1_levels + 7) / 8 && xcd8_level(7, j) < a->x1_levels) return -1;
        return 3;
    }
    if (a->x1_layout == PAG_LAYOUT_XCD8) {
        if (a->k1 != 64 || a->x1_levels < 1 || a->x1_feats < 1 || ((a->x1_levels + 7) / 8) * a->x1_feats > 8) return -1;
        const int j = 7 / a->x1_feats;                            // staged position 63 = group 7, element 7: must be padding
        if (j < (a->x1_levels + 7) / 8 && xcd8_level(7, j) < a->x1_levels) return -1;
        if (a->x2 || a->dx1_col0_add) return -1;
        if (!rank1 && a->grad_out && a->out_dtype == PAG_BF16 && a->out_act == PAG_ACT_NONE && a->out_dim % 4 == 0 && !a->dx1_accumulate) return 0;
        if (rank1 && a->g_scale && a->g_index && a->g_ray_scale && a->out && a->out_dtype == PAG_BF16 && a->out_act == PAG_ACT_SOFTMAX &&
            a->out_dim <= 8)
            return 2;
        return -1;
    }
    if (a->k1 == 16 && a->x2 && a->k2p == 32 && a->x2_index && a->in_dim <= 48 && !rank1 && a->grad_out && a->out && a->out_dtype == PAG_F32 &&
        a->out_act == PAG_ACT_SIGMOID && a->out_dim <= 4 && a->dx1_col0_add && a->dx1_col0_gate && !a->dx1_accumulate)
        return 1;
    return -1;
}

extern "C" int pag_mlp_bwd_fused_supported(const pag_mlp_bwd_args *a) { return fused_kind(a) >= 0 ? 1 : 0; }

extern "C" int pag_mlp_bwd_pair_supported(const pag_mlp_bwd_args *a, const pag_mlp_bwd_args *b) {
    if (!a || !b || fused_kind(a) != 3 || fused_kind(b) != 2 || b->n_layers != 2) return 0;
    return (b->x1 == a->x1 && b->x1_layout == PAG_LAYOUT_XCD8 && b->x1_levels == a->x1_levels && b->x1_feats == a->x1_feats && b->in_dim == a->in_dim &&
            a->dx1 && b->dx1 == a->dx1 && b->dx1_accumulate && !a->dx1_accumulate)
               ? 1
               : 0;
}

extern "C" int64_t pag_mlp_bwd_fused_workspace_bytes(const pag_mlp_bwd_args *a, int64_t M) {
    const int kind = fused_kind(a);
    if (kind < 0 || M < 1) return 0;
    const int64_t grid = fused_grid(M);
    if (kind == 3)      // stage A: one [224][96] slab per workgroup; stage B: per-wave slabs of two 64-row layers; + the [M,64] bf16 hidden gradient
        return (grid * 224 + grid * 128) * WG_SLAB_COLS * (int64_t)sizeof(float) + ((M + 31) / 32 + 1) * 32 * HID * 2;      // + a dump tile
    return grid * ((int64_t)(a->n_layers - 1) * 64 + 32) * WG_SLAB_COLS * (int64_t)sizeof(float);
}

extern "C" int pag_mlp_bwd(const pag_mlp_bwd_args *a, int64_t M, void *stream) {
    PAG_CHECK_ARG(a, "pag_mlp_bwd: args is NULL");
    PAG_CHECK_ARG(M >= 0, "pag_mlp_bwd: M < 0");
    PAG_CHECK_ARG(a->n_layers == 2 || a->n_layers == 3, "pag_mlp_bwd: n_layers %d not in {2,3}", a->n_layers);
    PAG_CHECK_ARG(a->k1 > 0 && a->k1 % 8 == 0 && a->k1 <= 64, "pag_mlp_bwd: k1 %d must be a multiple of 8 in (0,64]", a->k1);
    PAG_CHECK_ARG(a->in_dim > 0 && a->in_dim <= 64, "pag_mlp_bwd: in_dim %d out of range", a->in_dim);
    PAG_CHECK_ARG(a->out_dim >= 1 && a->out_dim <= 224, "pag_mlp_bwd: out_dim %d not in [1,224]", a->out_dim);
    PAG_CHECK_ARG(a->out_act >= PAG_ACT_NONE && a->out_act <= PAG_ACT_SOFTMAX, "pag_mlp_bwd: bad out_act %d", a->out_act);
    PAG_CHECK_ARG(a->out_act == PAG_ACT_NONE || a->out || (a->softmax_stats && a->b_last),
                  "pag_mlp_bwd: activated output (or softmax_stats + b_last) needed for sigmoid/softmax");
    PAG_CHECK_ARG(a->out_dtype == PAG_F32 || a->out_dtype == PAG_BF16, "pag_mlp_bwd: out dtype must be F32 or BF16");
    PAG_CHECK_ARG(a->dx1 == nullptr || a->dx1_dtype == PAG_F32 || a->dx1_dtype == PAG_BF16, "pag_mlp_bwd: dx1 dtype must be F32 or BF16");
    PAG_CHECK_ARG(a->mode == PAG_MLP_MFMA_BF16 || a->mode == PAG_MLP_FP32, "pag_mlp_bwd: bad mode %d", a->mode);
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(a->grad_out || (a->g_ray && a->g_scale && a->g_index), "pag_mlp_bwd: NULL grad_out (and no rank-1 gradient)");
    PAG_CHECK_ARG(!a->g_ray || (a->mode == PAG_MLP_MFMA_BF16 && (a->out || a->softmax_stats)),
                  "pag_mlp_bwd: rank-1 gradients need MFMA mode and the saved output");
    const bool fuse = a->wgrad_workspace != nullptr;
    const int kind = fuse ? fused_kind(a) : -1;
    if (fuse) {
        PAG_CHECK_ARG(kind >= 0, "pag_mlp_bwd: this decoder shape has no fused weight-gradient kernel (pag_mlp_bwd_fused_supported)");
        PAG_CHECK_ARG(a->wgrad_workspace_bytes >= pag_mlp_bwd_fused_workspace_bytes(a, M), "pag_mlp_bwd: wgrad_workspace too small");
        for (int l = 0; l < a->n_layers; ++l) PAG_CHECK_ARG(a->dW[l] && a->db[l], "pag_mlp_bwd: NULL dW/db of layer %d", l);
        for (int l = 0; l + 1 < a->n_layers; ++l) PAG_CHECK_ARG(a->b[l], "pag_mlp_bwd: fused mode recomputes the hidden activations: NULL bias b[%d]", l);
    }
    for (int l = 0; l < a->n_layers; ++l) PAG_CHECK_ARG(a->W[l] && (fuse || a->dz[l]), "pag_mlp_bwd: NULL weight/dz of layer %d", l);
    for (int l = 0; l + 1 < a->n_layers; ++l)
        PAG_CHECK_ARG(a->hidden_save[l] || (a->wgrad_workspace && (a->out_dim <= 32 || l + 2 < a->n_layers)), "pag_mlp_bwd: NULL hidden_save[%d]", l);
    BwdParams p;
    p.g_ray = a->g_ray;
    p.g_scale = a->g_scale;
    p.g_ray_scale = a->g_ray_scale;
    p.g_index = a->g_index;
    p.grad_out = a->grad_out ? a->grad_out : (const void *)a->out;      // never dereferenced in rank-1 mode
    p.out = a->out ? a->out : (const void *)a->grad_out;
    p.k1 = a->k1;
    p.in_dim = a->in_dim;
    p.in_pad = 64;
    p.out_dim = a->out_dim;
    p.act = a->out_act;
    for (int l = 0; l < 3; ++l) {
        p.W[l] = l < a->n_layers ? a->W[l] : nullptr;
        p.dz[l] = l < a->n_layers ? a->dz[l] : nullptr;
    }
    p.hsave[0] = a->hidden_save[0];
    p.hsave[1] = a->n_layers == 3 ? a->hidden_save[1] : nullptr;
    p.dx1 = a->dx1;
    p.M = M;
    p.grp_L = a->x1_layout == PAG_LAYOUT_XCD8 ? a->x1_levels : 0;
    p.grp_F = a->x1_feats;
    if (p.grp_L)
        PAG_CHECK_ARG(a->mode == PAG_MLP_MFMA_BF16 && (a->dx1 == nullptr || a->dx1_dtype == PAG_BF16) && a->k1 == 64 &&
                          ((p.grp_L + 7) / 8) * p.grp_F <= 8 && a->in_dim == p.grp_L * p.grp_F,
                      "pag_mlp_bwd: XCD8 dx1 needs MFMA mode, bf16, k1 = 64 and in_dim = levels*feats");
    hipStream_t st = (hipStream_t)stream;
    const bool out_f32 = a->out_dtype == PAG_F32;      // dtype of grad_out and of the saved activated output
    const bool dx_f32 = a->dx1 == nullptr || a->dx1_dtype == PAG_F32;
    p.stats = a->softmax_stats;
    p.b_last = a->b_last;
    p.dx1_acc = a->dx1_accumulate;
    p.dx_col0 = a->dx1_col0_add;
    p.dx_col0_gate = a->dx1_col0_gate;
    PAG_CHECK_ARG(!a->dx1_col0_gate || a->dx1_col0_add, "pag_mlp_bwd: dx1_col0_gate without dx1_col0_add");
    PAG_CHECK_ARG(!a->dx1_col0_add || (a->dx1 && !p.grp_L && a->mode == PAG_MLP_MFMA_BF16 && a->out_dim <= 64),
                  "pag_mlp_bwd: dx1_col0_add needs a strided dx1, MFMA mode and out_dim <= 64");
    PAG_CHECK_ARG(!a->dx1_accumulate || (p.grp_L && a->dx1), "pag_mlp_bwd: dx1_accumulate needs an XCD8 dx1");
    for (int l = 0; l < 3; ++l) p.b[l] = a->b[l];
    p.x1 = a->x1;
    p.x2 = a->x2;
    p.x2_index = a->x2_index;
    p.k2p = a->x2 ? a->k2p : 0;
    for (int l = 0; l < 3; ++l) p.slabs[l] = nullptr;
    if (fuse && kind == 3) {
        // ---- wide softmax head: stage A (output layer + dW_L) then stage B (layers 0, 1 as a 2-layer decoder whose upstream gradient is
        //      the hidden gradient stage A wrote)
        PAG_CHECK_ARG(M <= PAG_MLP_FUSED_WIDE_MAX_M, "pag_mlp_bwd: the fused wide-head backward addresses [M,64] bf16 tensors with 32-bit byte offsets: M %lld > %lld, split the batch",
                      (long long)M, (long long)PAG_MLP_FUSED_WIDE_MAX_M);
        const unsigned grid = fused_grid(M);
        float *ws = a->wgrad_workspace;
        float *slabA = ws;
        ws += (int64_t)grid * 224 * WG_SLAB_COLS;
        float *slabB0 = ws;
        ws += (int64_t)grid * 64 * WG_SLAB_COLS;
        float *slabB1 = ws;
        ws += (int64_t)grid * 64 * WG_SLAB_COLS;
        bf16_t *dzh = reinterpret_cast<bf16_t *>(ws);
        BwdParams pa = p;
        pa.W[0] = a->W[2];
        pa.hsave[0] = a->hidden_save[1];
        pa.dz[0] = dzh;
        pa.slabs[0] = slabA;
        constexpr int OBW = 7;
        const size_t ldsA = (size_t)(64 * (OBW * 32 + 8) + OBW * 32 * RS) * sizeof(bf16_t) + (size_t)OBW * 32 * sizeof(float) +
                            (size_t)(5 + OBW) * TW_ELEMS * sizeof(bf16_t) + (size_t)2 * WB_RMAX * WR_RS * sizeof(float) +
                            (size_t)2 * OBW * 64 * sizeof(float) + (size_t)2 * OBW * 2 * 64 * 16;
        static bool attrA = false;
        if (!attrA) {
            hipFuncSetAttribute((const void *)mlp_bwd_wide_blocks<OBW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void *)mlp_bwd_fused<2, 0, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attrA = true;
        }
        // ---- a companion head on the same input (args->pair): validated before anything is launched
        const pag_mlp_bwd_args *b = a->pair;
        BwdParams ps{};
        float *slabS0 = nullptr, *slabS1 = nullptr;
        if (b) {
            PAG_CHECK_ARG(b->wgrad_workspace && fused_kind(b) == 2 && b->n_layers == 2 && !b->pair, "pag_mlp_bwd: pair must be a two-layer narrow softmax head with a fused workspace");
            PAG_CHECK_ARG(b->x1 == a->x1 && b->x1_layout == PAG_LAYOUT_XCD8 && b->x1_levels == a->x1_levels && b->x1_feats == a->x1_feats && b->in_dim == a->in_dim,
                          "pag_mlp_bwd: pair must read the same XCD8 input");
            PAG_CHECK_ARG(b->dx1 == a->dx1 && b->dx1_accumulate && a->dx1 && !a->dx1_accumulate, "pag_mlp_bwd: pair must accumulate into this decoder's dx1");
            PAG_CHECK_ARG(b->wgrad_workspace_bytes >= pag_mlp_bwd_fused_workspace_bytes(b, M), "pag_mlp_bwd: pair wgrad_workspace too small");
            PAG_CHECK_ARG(b->W[0] && b->W[1] && b->b[0] && b->dW[0] && b->dW[1] && b->db[0] && b->db[1] && b->out && b->g_ray && b->g_scale && b->g_index && b->g_ray_scale,
                          "pag_mlp_bwd: pair: NULL weight / bias / gradient / saved output");
            ps.g_ray = b->g_ray;
            ps.g_scale = b->g_scale;
            ps.g_ray_scale = b->g_ray_scale;
            ps.g_index = b->g_index;
            ps.out = b->out;
            ps.grad_out = b->out;
            ps.k1 = 64;
            ps.in_dim = b->in_dim;
            ps.in_pad = 64;
            ps.out_dim = b->out_dim;
            ps.act = PAG_ACT_SOFTMAX;
            ps.W[0] = b->W[0];
            ps.W[1] = b->W[1];
            ps.b[0] = b->b[0];
            ps.M = M;
            ps.grp_L = p.grp_L;
            ps.grp_F = p.grp_F;
            ps.x1 = a->x1;
            ps.dx1 = a->dx1;
            slabS0 = b->wgrad_workspace;
            slabS1 = slabS0 + (int64_t)grid * 64 * WG_SLAB_COLS;
            ps.slabs[0] = slabS0;
            ps.slabs[1] = slabS1;
        }
        hipLaunchKernelGGL((mlp_bwd_wide_blocks<OBW>), dim3(grid), dim3((OBW + 1) * 64), ldsA, st, pa);
        PAG_CHECK_LAUNCH("pag_mlp_bwd (fused, wide head)");
#ifdef PAG_WB_PROF
        {
            unsigned long long hp[2][8];
            hipStreamSynchronize(st);
            hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_wb_prof), sizeof(hp));
            fprintf(stderr, "[wb_prof] grid %u:", grid);
            for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d %.0f/%.0f", w, (double)hp[0][w] / grid, (double)hp[1][w] / grid);
            fprintf(stderr, "\n");
            for (auto &row : hp) for (auto &v : row) v = 0;
            hipMemcpyToSymbol(HIP_SYMBOL(g_wb_prof), hp, sizeof(hp));
        }
#endif
        BwdParams pb = p;
        pb.grad_out = dzh;
        pb.out = dzh;
        pb.g_ray = nullptr;
        pb.out_dim = HID;
        pb.act = PAG_ACT_NONE;
        pb.W[0] = a->W[0];
        pb.W[1] = a->W[1];
        pb.W[2] = nullptr;
        pb.b[0] = a->b[0];
        pb.b[1] = nullptr;
        pb.hsave[0] = a->hidden_save[0];
        pb.hsave[1] = nullptr;
        pb.slabs[0] = slabB0;
        pb.slabs[1] = slabB1;
        FinishBatch fb{};
        fb.p[0] = FinishParams{slabB0, (int)grid, HID, 64, a->in_dim, p.grp_L, p.grp_F, a->dW[0], a->db[0]};
        fb.p[1] = FinishParams{slabB1, (int)grid, HID, 64, HID, 0, 0, a->dW[1], a->db[1]};
        fb.p[2] = FinishParams{slabA, (int)grid, a->out_dim, 224, HID, 0, 0, a->dW[2], a->db[2]};
        int n_fin = 3;
        if (b) {
            // the companion head runs in the same launch as the layers below the wide head: one read of the input, one write of the summed gradient
            const size_t ldsP = (size_t)(64 * 72 * 5 + 64 * 40) * sizeof(bf16_t) + 128 * sizeof(float) + (size_t)4 * 4 * TW_ELEMS * sizeof(bf16_t);
            static bool attrP = false;
            if (!attrP) {
                hipFuncSetAttribute((const void *)mlp_bwd_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attrP = true;
            }
            PairParams pp{pb, ps};
            hipLaunchKernelGGL(mlp_bwd_pair, dim3(grid), dim3(256), ldsP, st, pp);
            PAG_CHECK_LAUNCH("pag_mlp_bwd (fused, layers below the wide head + companion head)");
#ifdef PAG_FUSED_PROF
            {
                unsigned long long hp[8];
                hipStreamSynchronize(st);
                hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_fused_prof), sizeof(hp));
                const double n = hp[4] ? (double)hp[4] : 1.0;
                fprintf(stderr, "[fused_prof] pair M %lld grid %u: staging %.0f  tiles %.0f  cross-wave sum %.0f  (shader clocks per workgroup)\n", (long long)M, grid,
                        hp[0] / n, hp[1] / n, hp[2] / n);
                for (auto &v : hp) v = 0;
                hipMemcpyToSymbol(HIP_SYMBOL(g_fused_prof), hp, sizeof(hp));
            }
#endif
            fb.p[3] = FinishParams{slabS0, (int)grid, HID, 64, b->in_dim, p.grp_L, p.grp_F, b->dW[0], b->db[0]};
            fb.p[4] = FinishParams{slabS1, (int)grid, b->out_dim, 32, HID, 0, 0, b->dW[1], b->db[1]};
            n_fin = 5;
        } else {
            const size_t ldsB = (size_t)(64 * (64 + 8) + 2 * 64 * RS) * sizeof(bf16_t) + 128 * sizeof(float) + (size_t)4 * 3 * TW_ELEMS * sizeof(bf16_t);
            hipLaunchKernelGGL((mlp_bwd_fused<2, 0, false, 2>), dim3(grid), dim3(256), ldsB, st, pb);
            PAG_CHECK_LAUNCH("pag_mlp_bwd (fused, layers below the wide head)");
        }
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(a->out_dim, n_fin), dim3(WF_SPLITS * WG_SLAB_COLS), 0, st, fb);
        PAG_CHECK_LAUNCH("pag_mlp_bwd (fused, finish)");
        return PAG_OK;
    }
    if (fuse) {
        // one workgroup per CU, one slab each (its four waves are summed in LDS): [grid][rows_pad][96] f32 per layer, hidden layers first
        const unsigned grid = fused_grid(M);
        float *ws = a->wgrad_workspace;
        for (int l = 0; l < a->n_layers; ++l) {
            p.slabs[l] = ws;
            ws += (int64_t)grid * (l + 1 < a->n_layers ? 64 : 32) * WG_SLAB_COLS;
        }
        const size_t lds = (size_t)(64 * (32 + 8) + 2 * (a->n_layers == 3 ? 64 * RS : 0) + 2 * 64 * RS) * sizeof(bf16_t) + 128 * sizeof(float) +
                           (size_t)4 * (a->n_layers + 1) * TW_ELEMS * sizeof(bf16_t);
#define MLP_BWD_FUSED(NL_, KIND_, ACC_)                                                                                              \
    do {                                                                                                                              \
        static bool attr_done = false;                                                                                                \
        if (!attr_done) {                                                                                                             \
            hipFuncSetAttribute((const void *)mlp_bwd_fused<NL_, KIND_, ACC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                                         \
        }                                                                                                                             \
        hipLaunchKernelGGL((mlp_bwd_fused<NL_, KIND_, ACC_>), dim3(grid), dim3(256), lds, st, p);                                     \
    } while (0)
        const bool acc = a->dx1_accumulate != 0;
        p.dz[0] = kind == 1 ? a->dz[0] : nullptr;
        p.dz0_slots = kind == 1 ? a->dz0_slots : nullptr;
        if (kind == 1 && a->dz0_slots) {      // colour-like, per-ray input's gradient requested, samples packed ray by ray: per-(tile, ray) sums of dz_0
            static bool attr2s = false, attr3s = false;
            const size_t lds_s = lds + 4 * 64 * sizeof(int);
            if (a->n_layers == 2) {
                if (!attr2s) hipFuncSetAttribute((const void *)mlp_bwd_fused<2, 1, false, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr2s = true;
                hipLaunchKernelGGL((mlp_bwd_fused<2, 1, false, 1, 2>), dim3(grid), dim3(256), lds_s, st, p);
            } else {
                if (!attr3s) hipFuncSetAttribute((const void *)mlp_bwd_fused<3, 1, false, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr3s = true;
                hipLaunchKernelGGL((mlp_bwd_fused<3, 1, false, 1, 2>), dim3(grid), dim3(256), lds_s, st, p);
            }
        } else if (kind == 1 && a->dz[0]) {      // colour-like with the per-ray input's gradient requested: dz_0 is written as well
            static bool attr2 = false, attr3 = false;
            if (a->n_layers == 2) {
                if (!attr2) hipFuncSetAttribute((const void *)mlp_bwd_fused<2, 1, false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr2 = true;
                hipLaunchKernelGGL((mlp_bwd_fused<2, 1, false, 1, 1>), dim3(grid), dim3(256), lds, st, p);
            } else {
                if (!attr3) hipFuncSetAttribute((const void *)mlp_bwd_fused<3, 1, false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr3 = true;
                hipLaunchKernelGGL((mlp_bwd_fused<3, 1, false, 1, 1>), dim3(grid), dim3(256), lds, st, p);
            }
        } else if (a->n_layers == 2) {
            if (kind == 0) MLP_BWD_FUSED(2, 0, false);
            else if (kind == 1) MLP_BWD_FUSED(2, 1, false);
            else if (acc) MLP_BWD_FUSED(2, 2, true);
            else MLP_BWD_FUSED(2, 2, false);
        } else {
            if (kind == 0) MLP_BWD_FUSED(3, 0, false);
            else if (kind == 1) MLP_BWD_FUSED(3, 1, false);
            else if (acc) MLP_BWD_FUSED(3, 2, true);
            else MLP_BWD_FUSED(3, 2, false);
        }
#undef MLP_BWD_FUSED
        PAG_CHECK_LAUNCH("pag_mlp_bwd (fused)");
#ifdef PAG_FUSED_PROF
        {
            unsigned long long hp[8];
            hipStreamSynchronize(st);
            hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_fused_prof), sizeof(hp));
            const double n = hp[4] ? (double)hp[4] : 1.0;
            fprintf(stderr, "[fused_prof] kind %d M %lld grid %u: staging %.0f  tiles %.0f  cross-wave sum %.0f  (shader clocks per workgroup)\n", kind, (long long)M,
                    grid, hp[0] / n, hp[1] / n, hp[2] / n);
            for (auto &v : hp) v = 0;
            hipMemcpyToSymbol(HIP_SYMBOL(g_fused_prof), hp, sizeof(hp));
        }
#endif
        FinishBatch fb{};
        int max_out = 0;
        for (int l = 0; l < a->n_layers; ++l) {
            const int n_out = l + 1 < a->n_layers ? HID : a->out_dim;
            const int n_in = l == 0 ? a->in_dim : HID;
            fb.p[l] = FinishParams{p.slabs[l], (int)grid, n_out, (n_out + 31) / 32 * 32, n_in, l == 0 ? p.grp_L : 0, p.grp_F, a->dW[l], a->db[l]};
            max_out = std::max(max_out, n_out);
        }
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(max_out, a->n_layers), dim3(WF_SPLITS * WG_SLAB_COLS), 0, st, fb);
        PAG_CHECK_LAUNCH("pag_mlp_bwd (fused, finish)");
        return PAG_OK;
    }
    const bool wide_rebuild = a->mode == PAG_MLP_MFMA_BF16 && a->out_dim > 64 && a->out_act == PAG_ACT_SOFTMAX && a->softmax_stats &&
                              a->b_last && !out_f32 && (a->g_ray || (a->grad_out && a->out_dim % 8 == 0));
    if (wide_rebuild) {
        constexpr int NW = PAG_WIDE_BWD_WAVES;
        const int OB = (a->out_dim + 31) / 32;
        const size_t lds = (size_t)(64 * (OB * 32 + 8) + OB * 32 * RS + (a->n_layers == 3 ? 64 * RS : 0) + 64 * RS) * sizeof(bf16_t) +
                           (size_t)OB * 32 * sizeof(float) + (size_t)NW * ST_BYTES + (size_t)NW * OB * 32 * sizeof(float);
        const int64_t tiles = (M + 31) / 32;
        const unsigned grid = (unsigned)std::min<int64_t>((tiles + NW - 1) / NW, 256);      // one workgroup per CU, tiles grid-strided
#define MLP_BWD_WIDE(DxT, NL_)                                                                                              \
    do {                                                                                                                    \
        static bool attr_done = false;                                                                                      \
        if (!attr_done) {                                                                                                   \
            hipFuncSetAttribute((const void *)mlp_bwd_wide_mfma<DxT, NL_, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                                               \
        }                                                                                                                   \
        hipLaunchKernelGGL((mlp_bwd_wide_mfma<DxT, NL_, NW>), dim3(grid), dim3(NW * 64), lds, st, p);                     \
    } while (0)
        if (dx_f32 && a->n_layers == 2) MLP_BWD_WIDE(float, 2);
        else if (dx_f32) MLP_BWD_WIDE(float, 3);
        else if (a->n_layers == 2) MLP_BWD_WIDE(bf16_t, 2);
        else MLP_BWD_WIDE(bf16_t, 3);
#undef MLP_BWD_WIDE
    } else if (a->mode == PAG_MLP_MFMA_BF16) {
        const int OB = (a->out_dim + 31) / 32;
        const size_t lds = (size_t)(64 * (OB * 32 + 8) + (a->n_layers == 3 ? 64 * RS : 0) + 64 * RS) * sizeof(bf16_t) + 4 * ST_BYTES;
        if (out_f32 && dx_f32) MLP_BWD_NL(float, float);
        else if (out_f32) MLP_BWD_NL(float, bf16_t);
        else if (dx_f32) MLP_BWD_NL(bf16_t, float);
        else MLP_BWD_NL(bf16_t, bf16_t);
    } else {
        const size_t lds = (size_t)(224 * PT + 64 * 64) * sizeof(float);
        dim3 grid((unsigned)((M + PT - 1) / PT)), block(PT);
        static bool attr_set = false;   // > 64 KiB of dynamic LDS must be opted into once per kernel
        if (!attr_set) {
            hipFuncSetAttribute((const void *)mlp_bwd_f32<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void *)mlp_bwd_f32<float, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void *)mlp_bwd_f32<bf16_t, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void *)mlp_bwd_f32<bf16_t, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (out_f32 && dx_f32) hipLaunchKernelGGL((mlp_bwd_f32<float, float>), grid, block, lds, st, p, a->n_layers);
        else if (out_f32) hipLaunchKernelGGL((mlp_bwd_f32<float, bf16_t>), grid, block, lds, st, p, a->n_layers);
        else if (dx_f32) hipLaunchKernelGGL((mlp_bwd_f32<bf16_t, float>), grid, block, lds, st, p, a->n_layers);
        else hipLaunchKernelGGL((mlp_bwd_f32<bf16_t, bf16_t>), grid, block, lds, st, p, a->n_layers);
    }
    PAG_CHECK_LAUNCH("pag_mlp_bwd");
    return PAG_OK;
}

static void wgrad_launch(const WgradBatch &b, int count, bool a1_f32, bool small, int n_blocks, size_t lds, hipStream_t st) {
    const dim3 grid(n_blocks, count);
    if (a1_f32) {
        if (small) hipLaunchKernelGGL((mlp_wgrad_kernel<float, 2>), grid, dim3(256), lds, st, b);
        else hipLaunchKernelGGL((mlp_wgrad_kernel<float, 6>), grid, dim3(256), lds, st, b);
    } else {
        if (small) hipLaunchKernelGGL((mlp_wgrad_kernel<bf16_t, 2>), grid, dim3(256), lds, st, b);
        else hipLaunchKernelGGL((mlp_wgrad_kernel<bf16_t, 3, 8>), grid, dim3(512), lds, st, b);
    }
}

extern "C" int pag_mlp_wgrad_blocks(int64_t M) {
    int64_t chunks = (M + 63) / 64;
    return (int)(chunks < 1024 ? (chunks > 0 ? chunks : 1) : 1024);
}

extern "C" int pag_mlp_wgrad(const void *dz, int dz_cols, int n_out, const void *a1, int a1_dtype, int a1_layout, int k1,
                             const float *a2, int k2p, const int32_t *a2_index, int n_in, float *slabs, int n_blocks, int64_t M,
                             void *stream) {
    PAG_CHECK_ARG(M >= 0, "pag_mlp_wgrad: M < 0");
    PAG_CHECK_ARG(n_out >= 1 && n_out <= 224 && dz_cols >= n_out, "pag_mlp_wgrad: n_out %d / dz_cols %d out of range", n_out, dz_cols);
    PAG_CHECK_ARG(k1 > 0 && k1 % 8 == 0, "pag_mlp_wgrad: k1 %d must be a positive multiple of 8", k1);
    PAG_CHECK_ARG(a2 == nullptr || (k2p > 0 && k2p % 8 == 0 && a2_index), "pag_mlp_wgrad: a2 needs k2p %% 8 == 0 and a2_index");
    PAG_CHECK_ARG(n_in >= 1 && n_in <= 64 && n_in <= k1 + (a2 ? k2p : 0), "pag_mlp_wgrad: n_in %d out of range", n_in);
    PAG_CHECK_ARG(a1_dtype == PAG_F32 || a1_dtype == PAG_BF16, "pag_mlp_wgrad: a1 dtype must be F32 or BF16");
    PAG_CHECK_ARG(n_blocks >= 1, "pag_mlp_wgrad: n_blocks < 1");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(dz && a1 && slabs, "pag_mlp_wgrad: NULL dz/a1/slabs");
    PAG_CHECK_ARG(a1_layout == PAG_LAYOUT_STRIDED || (a1_dtype == PAG_BF16 && k1 == 64 && n_in == 64 && a2 == nullptr),
                  "pag_mlp_wgrad: XCD8 a1 needs bf16, k1 = n_in = 64 and no a2");
    WgradParams p{(const bf16_t *)dz, dz_cols, n_out, a1, k1, a2, a2 ? k2p : 0, a2_index, n_in, slabs, M, a1_layout == PAG_LAYOUT_XCD8};
    const int OB = (n_out + 31) / 32, IB = (n_in + 31) / 32;
    const size_t lds = (size_t)(OB + IB) * 32 * WG_RS * sizeof(bf16_t);
    const bool small = OB * (IB + 1) <= 8;      // fewer accumulators -> fewer VGPRs -> more resident workgroups
    WgradBatch b{};
    b.p[0] = p;
    wgrad_launch(b, 1, a1_dtype == PAG_F32, small, n_blocks, lds, (hipStream_t)stream);
    PAG_CHECK_LAUNCH("pag_mlp_wgrad");
    return PAG_OK;
}

extern "C" int pag_mlp_wgrad_finish(const float *slabs, int n_blocks, int n_out, int n_in, int a1_layout, int a1_levels, int a1_feats,
                                    float *dW, float *db, void *stream) {
    PAG_CHECK_ARG(n_blocks >= 1 && n_out >= 1 && n_out <= 224 && n_in >= 1 && n_in <= 64, "pag_mlp_wgrad_finish: size out of range");
    PAG_CHECK_ARG(slabs && dW && db, "pag_mlp_wgrad_finish: NULL slabs/dW/db");
    const int grouped = a1_layout == PAG_LAYOUT_XCD8;
    PAG_CHECK_ARG(!grouped || (a1_levels >= 1 && a1_feats >= 1 && a1_levels * a1_feats == n_in && ((a1_levels + 7) / 8) * a1_feats <= 8),
                  "pag_mlp_wgrad_finish: XCD8 needs n_in = levels*feats");
    FinishBatch fb{};
    fb.p[0] = FinishParams{slabs, n_blocks, n_out, (n_out + 31) / 32 * 32, n_in, grouped ? a1_levels : 0, a1_feats, dW, db};
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(n_out, 1), dim3(WF_SPLITS * WG_SLAB_COLS), 0, (hipStream_t)stream, fb);
    PAG_CHECK_LAUNCH("pag_mlp_wgrad_finish");
    return PAG_OK;
}

// All weight gradients of one decoder: the layers that share a kernel variant (input dtype, narrow / wide) go into ONE slab
// launch (grid.y = layer) and ONE finish launch sums every layer's slabs - 2-3 launches per decoder instead of 2 per layer.
extern "C" int pag_mlp_wgrad_batch(const pag_wgrad_layer *layers, int n_layers, int64_t M, void *stream) {
    PAG_CHECK_ARG(layers && n_layers >= 1 && n_layers <= WG_MAX_BATCH, "pag_mlp_wgrad_batch: n_layers %d not in [1,%d]", n_layers, WG_MAX_BATCH);
    PAG_CHECK_ARG(M >= 1, "pag_mlp_wgrad_batch: M < 1 (callers zero the gradients of an empty batch themselves)");
    hipStream_t st = (hipStream_t)stream;
    bool done[WG_MAX_BATCH] = {};
    FinishBatch fb{};
    int max_out = 0;
    for (int l = 0; l < n_layers; ++l) {
        const pag_wgrad_layer &y = layers[l];
        PAG_CHECK_ARG(y.n_out >= 1 && y.n_out <= 224 && y.dz_cols >= y.n_out, "pag_mlp_wgrad_batch: layer %d n_out %d / dz_cols %d out of range", l, y.n_out, y.dz_cols);
        PAG_CHECK_ARG(y.k1 > 0 && y.k1 % 8 == 0, "pag_mlp_wgrad_batch: layer %d k1 %d must be a positive multiple of 8", l, y.k1);
        PAG_CHECK_ARG(y.a2 == nullptr || (y.k2p > 0 && y.k2p % 8 == 0 && y.a2_index), "pag_mlp_wgrad_batch: layer %d a2 needs k2p %% 8 == 0 and a2_index", l);
        PAG_CHECK_ARG(y.n_in >= 1 && y.n_in <= 64 && y.n_in <= y.k1 + (y.a2 ? y.k2p : 0), "pag_mlp_wgrad_batch: layer %d n_in %d out of range", l, y.n_in);
        PAG_CHECK_ARG(y.a1_dtype == PAG_F32 || y.a1_dtype == PAG_BF16, "pag_mlp_wgrad_batch: layer %d a1 dtype must be F32 or BF16", l);
        PAG_CHECK_ARG(y.n_blocks >= 1 && y.dz && y.a1 && y.slabs && y.dW && y.db, "pag_mlp_wgrad_batch: layer %d NULL pointer or n_blocks < 1", l);
        const bool grouped = y.a1_layout == PAG_LAYOUT_XCD8;
        PAG_CHECK_ARG(!grouped || (y.a1_dtype == PAG_BF16 && y.k1 == 64 && y.n_in == 64 && y.a2 == nullptr && y.a1_levels >= 1 && y.a1_feats >= 1 &&
                                   ((y.a1_levels + 7) / 8) * y.a1_feats <= 8),
                      "pag_mlp_wgrad_batch: layer %d XCD8 a1 needs bf16, k1 = n_in = 64 (staged positions), levels*feats <= 64 and no a2", l);
        // XCD8: the slab columns are the 64 staged positions, dW has levels*feats feature columns
        fb.p[l] = FinishParams{y.slabs, y.n_blocks, y.n_out, (y.n_out + 31) / 32 * 32, grouped ? y.a1_levels * y.a1_feats : y.n_in,
                               grouped ? y.a1_levels : 0, y.a1_feats, y.dW, y.db};
        max_out = std::max(max_out, y.n_out);
    }
    for (int l = 0; l < n_layers; ++l) {
        if (done[l]) continue;
        const pag_wgrad_layer &y = layers[l];
        const bool f32 = y.a1_dtype == PAG_F32;
        const bool small = ((y.n_out + 31) / 32) * ((y.n_in + 31) / 32 + 1) <= 8;
        WgradBatch b{};
        int count = 0;
        size_t lds = 0;
        for (int k = l; k < n_layers; ++k) {
            const pag_wgrad_layer &z = layers[k];
            const int OB = (z.n_out + 31) / 32, IB = (z.n_in + 31) / 32;
            if (done[k] || (z.a1_dtype == PAG_F32) != f32 || (OB * (IB + 1) <= 8) != small || z.n_blocks != y.n_blocks) continue;
            b.p[count++] = WgradParams{(const bf16_t *)z.dz, z.dz_cols, z.n_out, z.a1, z.k1, z.a2, z.a2 ? z.k2p : 0, z.a2_index, z.n_in, z.slabs, M,
                                       z.a1_layout == PAG_LAYOUT_XCD8};
            lds = std::max(lds, (size_t)(OB + IB) * 32 * WG_RS * sizeof(bf16_t));
            done[k] = true;
        }
        wgrad_launch(b, count, f32, small, y.n_blocks, lds, st);
        PAG_CHECK_LAUNCH("pag_mlp_wgrad_batch (slabs)");
    }
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(max_out, n_layers), dim3(WF_SPLITS * WG_SLAB_COLS), 0, st, fb);
    PAG_CHECK_LAUNCH("pag_mlp_wgrad_batch (finish)");
    return PAG_OK;
}

extern "C" int pag_affine_xcd8_fwd(const void *x, int64_t M, int x_levels, int x_feats, const float *W, const float *b, int n_out, int in_dim,
                                   float *out, void *stream) {
    PAG_CHECK_ARG(M >= 0, "pag_affine_xcd8_fwd: M < 0");
    PAG_CHECK_ARG(n_out >= 1 && n_out <= AFF_MAX_OUT, "pag_affine_xcd8_fwd: n_out %d not in [1,%d]", n_out, AFF_MAX_OUT);
    PAG_CHECK_ARG(x_feats >= 1 && ((x_levels + 7) / 8) * x_feats <= 8 && in_dim == x_levels * x_feats, "pag_affine_xcd8_fwd: in_dim %d != levels %d * feats %d (or more than 8 values per group)",
                  in_dim, x_levels, x_feats);
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(x && W && b && out, "pag_affine_xcd8_fwd: NULL input/output");
    hipLaunchKernelGGL(affine_xcd8_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, M, x_levels, x_feats, W, b,
                       n_out, in_dim, out);
    PAG_CHECK_LAUNCH("pag_affine_xcd8_fwd");
    return PAG_OK;
}

extern "C" int pag_affine_xcd8_bwd_dx(const float *grad_out, int64_t M, int x_levels, int x_feats, const float *W, int n_out, int in_dim, void *dx,
                                      void *stream) {
    PAG_CHECK_ARG(M >= 0, "pag_affine_xcd8_bwd_dx: M < 0");
    PAG_CHECK_ARG(n_out >= 1 && n_out <= AFF_MAX_OUT, "pag_affine_xcd8_bwd_dx: n_out %d not in [1,%d]", n_out, AFF_MAX_OUT);
    PAG_CHECK_ARG(x_feats >= 1 && ((x_levels + 7) / 8) * x_feats <= 8 && in_dim == x_levels * x_feats, "pag_affine_xcd8_bwd_dx: in_dim %d != levels %d * feats %d (or more than 8 values per group)",
                  in_dim, x_levels, x_feats);
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(grad_out && W && dx, "pag_affine_xcd8_bwd_dx: NULL input/output");
    hipLaunchKernelGGL(affine_xcd8_bwd_dx_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_out, M, x_levels, x_feats, W, n_out, in_dim,
                       (bf16_t *)dx);
    PAG_CHECK_LAUNCH("pag_affine_xcd8_bwd_dx");
    return PAG_OK;
}

// seg[ray][c] = sum of the ray's (tile, ray) rows of mlp_bwd_fused<.., DZ0 = 2>: one wave per ray, lane = channel, rows tile + ray for the tiles the ray's
// samples span (fixed order: bitwise reproducible); rays without samples get zeros
namespace {
__global__ __launch_bounds__(256) void dz0_slots_sum_kernel(const int64_t *__restrict__ pack_start, int64_t R, const float *__restrict__ slots, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    const int64_t beg = pack_start[ray], end = pack_start[ray + 1];
    float a = 0.0f;
    if (end > beg)
        for (int64_t t = beg >> 5; t <= ((end - 1) >> 5); ++t) a += slots[(t + ray) * 64 + lane];
    out[ray * 64 + lane] = a;
}
}  // namespace
extern "C" int64_t pag_mlp_dz0_slots_bytes(int64_t M, int64_t R) { return (M > 0 && R > 0) ? ((M + 31) / 32 + R) * 64 * (int64_t)sizeof(float) : 0; }
extern "C" int pag_mlp_dz0_slots_sum(const int64_t *pack_start, int64_t R, const float *slots, float *out, void *stream) {
    PAG_CHECK_ARG(R >= 0, "pag_mlp_dz0_slots_sum: R < 0");
    if (R == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && slots && out, "pag_mlp_dz0_slots_sum: NULL input/output");
    hipLaunchKernelGGL(dz0_slots_sum_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pack_start, R, slots, out);
    PAG_CHECK_LAUNCH("pag_mlp_dz0_slots_sum");
    return PAG_OK;
}

extern "C" int pag_head_composite_fwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const void *hidden,
                                      const float *W_last, const float *b_last, int out_dim, const float *softmax_stats,
                                      const float *weights, const float *alpha, float *out, int64_t samples_hint, void *stream) {
    PAG_CHECK_ARG(P >= 0, "pag_head_composite_fwd: P < 0");
    PAG_CHECK_ARG(out_dim > 64 && out_dim <= 224, "pag_head_composite_fwd: out_dim %d not in (64,224]", out_dim);
    if (P == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && ray_of_pack && hidden && W_last && b_last && softmax_stats && weights && alpha && out,
                  "pag_head_composite_fwd: NULL input/output");
    HeadCompParams p{pack_start, ray_of_pack, P, (const bf16_t *)hidden, W_last, b_last, out_dim, softmax_stats, 0, weights, alpha, out};
#ifndef PAG_HC_PER_WAVE_P
#define PAG_HC_PER_WAVE_P 2048
#endif
    // one wave per pack when packs are short (fewer than ~5 tiles on average) - and whenever there are enough packs to fill the chip that way
    // (2048 = 512 workgroups of 4 waves): the lane reduction of the 7 x 16 partial sums at the end of a pack costs as much as two tiles, and four
    // waves sharing a pack each pay it
    p.per_wave = ((samples_hint > 0 && samples_hint < 160 * P) || P >= PAG_HC_PER_WAVE_P) ? 1 : 0;
    const int OB = (out_dim + 31) / 32;
    const size_t lds = (size_t)OB * 32 * RS * sizeof(bf16_t) + (size_t)5 * OB * 32 * sizeof(float) + 4 * ST_BYTES;
#ifndef PAG_HC_GRID
#define PAG_HC_GRID 512
#endif
    const unsigned grid = (unsigned)std::min<int64_t>(p.per_wave ? (P + 3) / 4 : P, PAG_HC_GRID);
    if (OB == 7) hipLaunchKernelGGL(head_composite_fwd_kernel<7>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(head_composite_fwd_kernel<0>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    PAG_CHECK_LAUNCH("pag_head_composite_fwd");
    return PAG_OK;
}

PAG_BLOCK_TIMING_EXPORT(mlp)
