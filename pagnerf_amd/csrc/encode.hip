// Grid feature interpolation for gfx950: multiresolution hash grid and permutohedral lattice.
//
// Launch geometry (both encoders, forward and backward):
//   work item = (tile of 256 samples, XCD group g = blockIdx % 8); the block walks levels
//   g, g+8, g+16, ... for its samples.  Workgroups are dealt round-robin over the 8 XCDs, so each
//   XCD's private 4 MiB L2 only ever sees ceil(L/8) of the L level tables (2 MiB each at T = 2^18,
//   F = 2, fp32) instead of all of them: the random per-vertex gathers are then served from L2
//   rather than from the Infinity Cache.  This is a speed-only assumption: any other placement is
//   still correct.
//   One lane = one sample; per level it has 4 (permuto) or 8 (hash) independent F-wide gathers in
//   flight, times LPX levels unrolled.
//
// Numerics: all fp32 arithmetic is written with explicit round-to-nearest intrinsics in the op
// order of the oracle (oracle/hash_encode.py, oracle/permuto_encode.py); the file is compiled with
// -ffp-contract=off.  With fp32 tables and fp32 output the result is bit-identical to the oracle.
#include "encode_common.h"
#include "blocktime.h"

using namespace pag_enc;

namespace {

// XCD-grouped feature layout (PAG_LAYOUT_XCD8): out[g][m][8] bf16, element e = j * F + f holds level xcd8_level(g, j) (common.h)
// (zero padded): each lane stores ONE aligned 16-byte piece holding all the levels it computed, a wave
// stores 1 KiB contiguously - instead of LPX*F scattered 2-byte pieces per sample in a [M, L*F] row
// (measured 8x write amplification).  The decoders read the same pieces as MFMA B fragments.
// addend (optional, same layout): the piece stored is bf16(addend + bf16(value)) - exactly what a separate bf16 tensor add
// of the two feature tensors yields (pc_nerf/panoptic_delta_nef.py:226 `feats.detach() + delta`), without the pass.
typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
// the addend piece is requested at the top of the kernel so that its latency hides under the lattice arithmetic
__device__ __forceinline__ bf16x8_t load_addend(const bf16_t *addend, int64_t M, int g, int64_t i) {
    bf16x8_t a;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (bf16_t)0.0f;
    if (addend) a = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t *>(addend + ((int64_t)g * M + i) * 8));   // streamed once: keep the tables in L2
    return a;
}
template <int N>
__device__ __forceinline__ void store_grouped(bf16_t *out, int64_t M, int g, int64_t i, const float (&v)[N], bool has_addend,
                                              const bf16x8_t &a) {
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(e < N ? v[e < N ? e : 0] : 0.0f);
    if (has_addend) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)a[e] + (float)o[e]);
    }
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8_t *>(out + ((int64_t)g * M + i) * 8));
}
__device__ __forceinline__ void store_grouped(float *, int64_t, int, int64_t, ...) {}

// Workgroup size of the per-sample, XCD-pinned kernels (forward gathers, position gradients).  A workgroup of the permutohedral forward lives ~5 us (12 gathers per thread, combine, store) and the launch has
// 8 * M / threads of them: at 256 threads the dispatcher handed a CU a new workgroup only every ~0.5 us and the CUs held 3.3 workgroups
// on average where their registers allow 6 (scripts/block_timeline.py) - half of the gathers that could be in flight were not.  512
// threads halve the number of workgroups: 399 -> 328 us on the bench workload (two 256-sample tiles per workgroup in a loop measure the
// same but cost 10 VGPRs; 1024 threads lose a little again: 344).
#ifndef PAG_ENC_FWD_THREADS
#define PAG_ENC_FWD_THREADS 512
#endif
template <typename TableT, typename OutT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS) void hash_fwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                       const TableT *__restrict__ tables, HashParams p,
                                                       OutT *__restrict__ out, int64_t sm, int64_t sc, int grouped,
                                                       const bf16_t *__restrict__ addend) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * PAG_ENC_FWD_THREADS + threadIdx.x;
    if (i >= M) return;
    float x[3];
    load_xyz(xyz, i, p.half_coords, x);
    const bf16x8_t addv = load_addend(addend, M, g, i);
    const int64_t T = (int64_t)1 << p.log2T;
    float e[LPX][8][F];
    float w[LPX][3];
    float gvals[LPX * F <= 8 ? LPX * F : 1];
#pragma unroll
    for (int q = 0; q < (LPX * F <= 8 ? LPX * F : 1); ++q) gvals[q] = 0.0f;
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = xcd8_level(g, j);
        int le = l < p.L ? l : p.L - 1;
        uint32_t idx[8];
        hash_cell(x, p.res[le], p.log2T, idx, w[j]);
        const TableT *tab = tables + (int64_t)le * T * F;
        // The fine levels are bound by the L2's request rate (one request per lane and vertex, no two lanes on a line), not by
        // bytes.  The hash is an XOR with the x cell index, so when that index is even the two x-corners of a cell are rows 2q
        // and 2q+1 (only bit 0 differs): those lanes fetch each corner pair with ONE 16-byte request (F = 2, fp32 rows) - 6
        // instead of 8 requests per sample and level on average: 0.754 -> 0.625 ms per launch (L = 16, T = 2^19).
        bool paired = false;
        if constexpr (F == 2 && sizeof(TableT) == 4) {
            if (p.pair_loads && (idx[0] ^ idx[4]) == 1u) {
                paired = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 v = *reinterpret_cast<const float4 *>(tab + (int64_t)(idx[k] & ~1u) * 2);
                    const bool odd = idx[k] & 1u;
                    e[j][k][0] = odd ? v.z : v.x;
                    e[j][k][1] = odd ? v.w : v.y;
                    e[j][k + 4][0] = odd ? v.x : v.z;
                    e[j][k + 4][1] = odd ? v.y : v.w;
                }
            }
        }
        if (!paired) {
#pragma unroll
            for (int k = 0; k < 8; ++k) gather<F>(tab + (int64_t)idx[k] * F, e[j][k]);
        }
    }
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = xcd8_level(g, j);
        if (l >= p.L) break;
        const float wx = w[j][0], wy = w[j][1], wz = w[j][2];
        const float ox = __fsub_rn(1.0f, wx), oy = __fsub_rn(1.0f, wy), oz = __fsub_rn(1.0f, wz);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float c00 = lerp_ref(e[j][0][f], e[j][4][f], wx, ox);
            float c01 = lerp_ref(e[j][1][f], e[j][5][f], wx, ox);
            float c10 = lerp_ref(e[j][2][f], e[j][6][f], wx, ox);
            float c11 = lerp_ref(e[j][3][f], e[j][7][f], wx, ox);
            float c0 = lerp_ref(c00, c10, wy, oy);
            float c1 = lerp_ref(c01, c11, wy, oy);
            float v = lerp_ref(c0, c1, wz, oz);
            if (p.has_scale) v = __fmul_rn(v, p.scale[l * F + f]);
            if (!grouped) pag_st(out + i * sm + (int64_t)(l * F + f) * sc, v);
            if constexpr (LPX * F <= 8) gvals[j * F + f] = v;
        }
    }
    if constexpr (LPX * F <= 8 && sizeof(OutT) == 2)
        if (grouped) store_grouped<LPX * F>(out, M, g, i, gvals, addend != nullptr, addv);
}

template <typename GradT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS) void hash_bwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                       const GradT *__restrict__ go, int64_t sm, int64_t sc,
                                                       HashParams p, float *__restrict__ gtab) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * PAG_ENC_FWD_THREADS + threadIdx.x;
    if (i >= M) return;
    float x[3];
    load_xyz(xyz, i, p.half_coords, x);
    const int64_t T = (int64_t)1 << p.log2T;
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = xcd8_level(g, j);
        if (l >= p.L) break;
        uint32_t idx[8];
        float w[3];
        hash_cell(x, p.res[l], p.log2T, idx, w);
        float gv[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            gv[f] = pag_ld(go + i * sm + (int64_t)(l * F + f) * sc);
            if (p.has_scale) gv[f] *= p.scale[l * F + f];
        }
        float *tab = gtab + (int64_t)l * T * F;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float wc = ((k & 4) ? w[0] : 1.0f - w[0]) * ((k & 2) ? w[1] : 1.0f - w[1]) * ((k & 1) ? w[2] : 1.0f - w[2]);
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(tab + (int64_t)idx[k] * F + f, gv[f] * wc);
        }
    }
}

#ifdef PAG_EXP_JAC
__device__ bf16_t *g_exp_jac = nullptr;
#endif
template <typename TableT, typename OutT, int F, int LPX>
__device__ __forceinline__ void permuto_fwd_body(const float *__restrict__ xyz, int64_t M, const TableT *__restrict__ tables, const PermutoParams &p,
                                                 OutT *__restrict__ out, int64_t sm, int64_t sc, int grouped, const bf16_t *__restrict__ addend) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * PAG_ENC_FWD_THREADS + threadIdx.x;
    if (i >= M) return;
    float x[3];
    load_xyz(xyz, i, p.half_coords, x);
    const bf16x8_t addv = load_addend(addend, M, g, i);
    float e[LPX][4][F];
    float bary[LPX][4];
    float gvals[LPX * F <= 8 ? LPX * F : 1];
#pragma unroll
    for (int q = 0; q < (LPX * F <= 8 ? LPX * F : 1); ++q) gvals[q] = 0.0f;
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = xcd8_level(g, j);
        int le = l < p.L ? l : p.L - 1;
        uint32_t idx[4];
#ifdef PAG_DBG_ONLY_J      // experiment: only the j-th level of every XCD group (what a level-phased launch would run per phase)
        if (j != PAG_DBG_ONLY_J) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bary[j][r] = 0.0f;
#pragma unroll
                for (int f = 0; f < F; ++f) e[j][r][f] = 0.0f;
            }
            continue;
        }
#endif
        permuto_simplex(x, p.shift[le], p.sf[le], p.capacity, p.pow2mask, idx, bary[j]);
        const TableT *tab = tables + (int64_t)le * p.capacity * F;
#ifdef PAG_DBG_HOTIDX      // experiment: all gathers hit 256 hot rows (isolates the arithmetic + store cost)
#pragma unroll
        for (int r = 0; r < 4; ++r) idx[r] &= 0xFFu;
#endif
#ifdef PAG_DBG_CHEAPIDX    // experiment: random rows from a 2-instruction hash (isolates the gather cost)
#pragma unroll
        for (int r = 0; r < 4; ++r) idx[r] = (((uint32_t)i * 4u + r + le * 77u) * 2654435761u) >> 14;
#pragma unroll
        for (int r = 0; r < 4; ++r) bary[j][r] = 0.25f;
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gather_row<F>(tab, idx[r], e[j][r]);
        }
    }
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = xcd8_level(g, j);
        if (l >= p.L) break;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __fadd_rn(acc, __fmul_rn(e[j][r][f], bary[j][r]));
            if (p.has_scale) acc = __fmul_rn(acc, p.scale[l * F + f]);
            if (!grouped) pag_st(out + i * sm + (int64_t)(l * F + f) * sc, acc);
            if constexpr (LPX * F <= 8) gvals[j * F + f] = acc;
        }
    }
    if constexpr (LPX * F <= 8 && sizeof(OutT) == 2)
        if (grouped) store_grouped<LPX * F>(out, M, g, i, gvals, addend != nullptr, addv);
#ifdef PAG_EXP_JAC      // experiment (scripts/exp_jac_store.py): what would it cost the forward to write d feat / d xyz (6 values per level) next to the features?
    if (g_exp_jac != nullptr && addend == nullptr) {
        bf16_t jv[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) jv[q] = (bf16_t)0.0f;
#pragma unroll
        for (int j = 0; j < LPX; ++j) {
            const int l = xcd8_level(g, j), le = l < p.L ? l : p.L - 1;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float d01 = e[j][0][f] - e[j][1][f], d12 = e[j][1][f] - e[j][2][f], d23 = e[j][2][f] - e[j][3][f], d30 = e[j][3][f] - e[j][0][f];
                if (j * 6 + f * 3 + 2 < 24) {
                    jv[j * 6 + f * 3 + 0] = (bf16_t)(0.25f * (d01 - d12) * p.sf[le][0]);
                    jv[j * 6 + f * 3 + 1] = (bf16_t)(0.25f * (d01 + d12 - 2.0f * d23) * p.sf[le][1]);
                    jv[j * 6 + f * 3 + 2] = (bf16_t)(0.25f * (d01 + d12 + d23 - 3.0f * d30) * p.sf[le][2]);
                }
            }
        }
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t *dst = reinterpret_cast<u32x4_t *>(g_exp_jac + ((int64_t)g * M + i) * PAG_EXP_JAC);
        const u32x4_t *srcv = reinterpret_cast<const u32x4_t *>(jv);
#pragma unroll
        for (int q = 0; q < PAG_EXP_JAC / 8; ++q) dst[q] = srcv[q];
    }
#endif
}

// Two kernel symbols for the same body: the plain launch (the roofline kernel of bench.py) and the `_add` launch of the delta grid, so that
// rocprofv3's per-kernel statistics and PMC counters keep them apart.
#ifndef PAG_ENC_FWD_WAVES
#define PAG_ENC_FWD_WAVES 1      // minimum waves per SIMD asked of the compiler for the permutohedral forward (experiments: 8 = 64 VGPRs)
#endif
template <typename TableT, typename OutT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS, PAG_ENC_FWD_WAVES) void permuto_fwd_kernel(const float *__restrict__ xyz, int64_t M, const TableT *__restrict__ tables, PermutoParams p,
                                                          OutT *__restrict__ out, int64_t sm, int64_t sc, int grouped) {
    PAG_BLOCK_TIMER(1);
    permuto_fwd_body<TableT, OutT, F, LPX>(xyz, M, tables, p, out, sm, sc, grouped, nullptr);
}
template <typename TableT, typename OutT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS, PAG_ENC_FWD_WAVES) void permuto_fwd_add_kernel(const float *__restrict__ xyz, int64_t M, const TableT *__restrict__ tables, PermutoParams p,
                                                              OutT *__restrict__ out, int64_t sm, int64_t sc, int grouped,
                                                              const bf16_t *__restrict__ addend) {
    PAG_BLOCK_TIMER(2);
    permuto_fwd_body<TableT, OutT, F, LPX>(xyz, M, tables, p, out, sm, sc, grouped, addend);
}

template <typename GradT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS) void permuto_bwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                          const GradT *__restrict__ go, int64_t sm, int64_t sc,
                                                          PermutoParams p, float *__restrict__ gtab) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * PAG_ENC_FWD_THREADS + threadIdx.x;
    if (i >= M) return;
    float x[3];
    load_xyz(xyz, i, p.half_coords, x);
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = xcd8_level(g, j);
        if (l >= p.L) break;
        uint32_t idx[4];
        float bary[4];
        permuto_simplex(x, p.shift[l], p.sf[l], p.capacity, p.pow2mask, idx, bary);
        float gv[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            gv[f] = pag_ld(go + i * sm + (int64_t)(l * F + f) * sc);
            if (p.has_scale) gv[f] *= p.scale[l * F + f];
        }
        float *tab = gtab + (int64_t)l * p.capacity * F;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(tab + (int64_t)idx[r] * F + f, gv[f] * bary[r]);
    }
}


// ------------------------------------------------------------------------ d loss / d xyz (pose optimisation)
// pc_nerf/ba_pipeline.py:85-92 makes the ray origins/directions functions of the camera extrinsics, so the samples
// o + t*d carry a gradient and the encoders must return d loss / d xyz.  Within a cell / simplex the features are
// (tri)linear in xyz, so the kernel is a second gather pass with the forward's geometry: per (sample, level) it
// fetches the same 8 / 4 rows, contracts them with the incoming gradient and chains through the weight
// derivatives.  Same XCD-pinned launch as the forward; group g writes its partial sum to part[g][m][3] and a tiny
// second kernel adds the 8 groups (deterministic, no atomics).
// lane-crossing helpers of the segmented scans (position gradient per ray; the bin pass's merge of adjacent lanes)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int src) {
    return __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, 0xF, true);   // masked / out-of-row sources read as 0 = the scan identity
}
template <int F, int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(float (&v)[F], int &f) {
    // element = (f: a run head lies between the source lane (exclusive) and this lane (inclusive), v: sum since that head)
    const int fp = dpp_i<CTRL, ROW_MASK>(f);
#pragma unroll
    for (int k = 0; k < F; ++k) {
        const float vp = __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v[k])));
        v[k] = f ? v[k] : v[k] + vp;
    }
    f |= fp;
}
// any of the eight bf16 values of a gradient piece other than +-0 (NaN counts as non-zero)
typedef bf16_t bf16x8_piece __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bool piece_nonzero(const bf16x8_piece &t) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 b = __builtin_bit_cast(u32x4, t);
    return ((b[0] | b[1] | b[2] | b[3]) & 0x7FFF7FFFu) != 0u;
}

// d loss / d xyz of sample i from the levels of XCD group g: dx[3] (zero, without a single row request, when the sample's gradient piece is exactly zero)
template <int KIND /*0 hash, 1 permuto*/, typename TableT, typename GradT, int F, int LPX>
__device__ __forceinline__ void xyz_grad_sample(const float *__restrict__ xyz, int64_t i, int64_t M, int g, const TableT *__restrict__ tables,
                                                const GradT *__restrict__ go, int64_t sm, int64_t sc, int grouped, const HashParams &hp,
                                                const PermutoParams &pp, float (&dx)[3]) {
    constexpr int NV = KIND == 0 ? 8 : 4;
    dx[0] = 0.0f, dx[1] = 0.0f, dx[2] = 0.0f;
    const int L = KIND == 0 ? hp.L : pp.L;
    const float *scale = KIND == 0 ? hp.scale : pp.scale;
    const bool has_scale = KIND == 0 ? hp.has_scale : pp.has_scale;
    const int64_t rows = KIND == 0 ? ((int64_t)1 << hp.log2T) : (int64_t)pp.capacity;
    float x[3];
    load_xyz(xyz, i, KIND == 0 ? hp.half_coords : pp.half_coords, x);
    float gpiece[8];
    if (grouped) {
        typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t t = *reinterpret_cast<const bf16x8_t *>(reinterpret_cast<const bf16_t *>(go) + ((int64_t)g * M + i) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) gpiece[e] = (float)t[e];
        // A sample whose gradient piece is exactly zero (+-0: empty space of a trained scene - sigma = relu(pre) = 0, so neither the colour nor the
        // density path sends anything back -, filler samples of a padded batch) gets d xyz = 0 without its 4 x LPX row requests: the same value
        // the products with zero give (scripts/zero_weight_tiles.py: 93 % of the samples of the trained analytic scene).
        if (!piece_nonzero(t)) return;
    }
    float e[LPX][NV][F];
    float w[LPX][4];         // hash: wx, wy, wz, -   permuto: unused
    float dw[LPX][3];        // hash: d w / d x per axis
    int slot[LPX][4];        // permuto: 3 - rank
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        const int l = xcd8_level(g, j);
        const int le = l < L ? l : L - 1;
        const TableT *tab = tables + (int64_t)le * rows * F;
        if constexpr (KIND == 0) {
            uint32_t idx[8];
            float w3[3];
            hash_cell(x, hp.res[le], hp.log2T, idx, w3, dw[j]);
            w[j][0] = w3[0], w[j][1] = w3[1], w[j][2] = w3[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) gather<F>(tab + (int64_t)idx[k] * F, e[j][k]);
        } else {
            uint32_t idx[4];
            float bary[4];
            permuto_simplex(x, pp.shift[le], pp.sf[le], pp.capacity, pp.pow2mask, idx, bary, slot[j]);
#pragma unroll
            for (int k = 0; k < 4; ++k) gather_row<F>(tab, idx[k], e[j][k]);      // scalar base + 32-bit offset, as the forward
        }
    }
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        const int l = xcd8_level(g, j);
        if (l >= L) break;
        float gv[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            gv[f] = grouped ? gpiece[(j * F + f) & 7] : pag_ld(go + i * sm + (int64_t)(l * F + f) * sc);
            if (has_scale) gv[f] *= scale[l * F + f];
        }
        if constexpr (KIND == 0) {
            const float wx = w[j][0], wy = w[j][1], wz = w[j][2];
            const float ox = 1.0f - wx, oy = 1.0f - wy, oz = 1.0f - wz;
            float gx = 0.0f, gy = 0.0f, gz = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float e0 = e[j][0][f], e1 = e[j][1][f], e2 = e[j][2][f], e3 = e[j][3][f];
                const float e4 = e[j][4][f], e5 = e[j][5][f], e6 = e[j][6][f], e7 = e[j][7][f];
                const float c00 = e0 * ox + e4 * wx, c01 = e1 * ox + e5 * wx, c10 = e2 * ox + e6 * wx, c11 = e3 * ox + e7 * wx;
                const float c0 = c00 * oy + c10 * wy, c1 = c01 * oy + c11 * wy;
                gx += gv[f] * (oz * (oy * (e4 - e0) + wy * (e6 - e2)) + wz * (oy * (e5 - e1) + wy * (e7 - e3)));
                gy += gv[f] * (oz * (c10 - c00) + wz * (c11 - c01));
                gz += gv[f] * (c1 - c0);
            }
            dx[0] += gx * dw[j][0];
            dx[1] += gy * dw[j][1];
            dx[2] += gz * dw[j][2];
        } else {
            float gb[5];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = 0.0f;
#pragma unroll
                for (int f = 0; f < F; ++f) a += gv[f] * e[j][k][f];
                gb[k] = a;
            }
            gb[4] = gb[0];
            float gE[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float hi = 0.0f, lo = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {     // predicated select: no runtime-indexed private array
                    hi = (k == slot[j][a]) ? gb[k] : hi;
                    lo = (k == slot[j][a]) ? gb[k + 1] : lo;
                }
                gE[a] = 0.25f * (hi - lo);
            }
            // E0 = cf0+cf1+cf2, E1 = cf2+cf1-cf0, E2 = cf2-2cf1, E3 = -3cf2
            dx[0] += (gE[0] - gE[1]) * pp.sf[l][0];
            dx[1] += (gE[0] + gE[1] - 2.0f * gE[2]) * pp.sf[l][1];
            dx[2] += (gE[0] + gE[1] + gE[2] - 3.0f * gE[3]) * pp.sf[l][2];
        }
    }
}

template <int KIND, typename TableT, typename GradT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS) void xyz_grad_kernel(const float *__restrict__ xyz, int64_t M, const TableT *__restrict__ tables,
                                                       const GradT *__restrict__ go, int64_t sm, int64_t sc, int grouped,
                                                       HashParams hp, PermutoParams pp, float *__restrict__ part) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * PAG_ENC_FWD_THREADS + threadIdx.x;
    if (i >= M) return;
    float dx[3];
    xyz_grad_sample<KIND, TableT, GradT, F, LPX>(xyz, i, M, g, tables, go, sm, sc, grouped, hp, pp, dx);
    float *o = part + ((int64_t)g * M + i) * 3;
    o[0] = dx[0], o[1] = dx[1], o[2] = dx[2];
}

// The same pass reduced PER RAY before anything is written (pose optimisation: what the position gradient is wanted for is d loss / d origin = sum of
// d xyz over a ray's samples and d loss / d dir = sum of d xyz * depth, pc_nerf/ba_pipeline.py:85-92 through samples = origin + dir * depth).  The
// per-sample form writes 8 partial planes [M,3] (96 B per sample), adds them (108 B more) and sums per ray (16 B more): 1.4 GB and two further
// launches for a dense 24 576-ray step.  Here a wave - 64 consecutive samples, sorted by ray - forms (dx, dx * depth) per lane, sums the lanes of
// each ray with a segmented DPP scan and its tail lanes write one 6-float slot per (group, wave, ray): slot row = wave + ray (strictly increasing
// along the samples: every (wave, ray) pair has a row of its own, `waves + N` rows per group) - 3 B per sample on 512-sample rays.  ray_slots_sum_kernel
// adds a ray's rows (fixed order: bitwise reproducible).
template <int KIND, typename TableT, typename GradT, int F, int LPX>
__global__ __launch_bounds__(PAG_ENC_FWD_THREADS) void xyz_grad_rays_kernel(const float *__restrict__ xyz, int64_t M, const TableT *__restrict__ tables,
                                                       const GradT *__restrict__ go, int64_t sm, int64_t sc, int grouped,
                                                       HashParams hp, PermutoParams pp, const int32_t *__restrict__ ridx,
                                                       const float *__restrict__ depths, float *__restrict__ slots, int64_t slot_rows) {
    const int g = blockIdx.x & 7;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * PAG_ENC_FWD_THREADS + threadIdx.x;
    const int64_t wave = i >> 6;
    if ((wave << 6) >= M) return;                    // whole wave past the end (wave-uniform)
    const bool in_range = i < M;
    const int64_t ic = in_range ? i : M - 1;
    float dx[3] = {0.0f, 0.0f, 0.0f};
    if (in_range) xyz_grad_sample<KIND, TableT, GradT, F, LPX>(xyz, i, M, g, tables, go, sm, sc, grouped, hp, pp, dx);
    const int ray = ridx[ic];
    const float dep = depths[ic];
    float v[6] = {dx[0], dx[1], dx[2], dx[0] * dep, dx[1] * dep, dx[2] * dep};
    const int prev = dpp_i<0x138, 0xF>(ray);         // wave_shr:1 (lane 0 reads 0: it is a head regardless)
    int f = (lane == 0 || prev != ray) ? 1 : 0;      // first lane of a ray's run
    const unsigned long long heads = __ballot(f != 0);
    seg_step<6, 0x111, 0xF>(v, f);                   // segmented inclusive scan: row_shr 1 / 2 / 4 / 8, then across the 16-lane rows
    seg_step<6, 0x112, 0xF>(v, f);
    seg_step<6, 0x114, 0xF>(v, f);
    seg_step<6, 0x118, 0xF>(v, f);
    seg_step<6, 0x142, 0xA>(v, f);
    seg_step<6, 0x143, 0xC>(v, f);
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (tail) {
        float *o = slots + ((int64_t)g * slot_rows + wave + ray) * 6;
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] = v[c];
    }
}

// out[ray] = (d origin | d dir) = sum over the ray's slot rows: groups x (waves its samples span); one wave per ray, lane = (group, wave) pair
__global__ __launch_bounds__(256) void ray_slots_sum_kernel(const int64_t *__restrict__ pack_start, int64_t N, const float *__restrict__ slots, int64_t slot_rows,
                                                            int groups, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;
    const int64_t beg = pack_start[ray], end = pack_start[ray + 1];
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (end > beg) {
        const int64_t w0 = beg >> 6, nw = ((end - 1) >> 6) - w0 + 1;
        for (int64_t q = lane; q < nw * groups; q += 64) {
            const int64_t gq = q / nw, wq = w0 + q % nw;
            const float *o = slots + (gq * slot_rows + wq + ray) * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) a[c] += o[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) a[c] += __shfl_xor(a[c], d);
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 6; ++c) out[ray * 6 + c] = a[c];
}


__global__ __launch_bounds__(256) void xyz_grad_sum_kernel(const float *__restrict__ part, int64_t n, int groups, float *__restrict__ out) {
    // four consecutive floats per lane and group (16-byte loads: eight of them in flight per lane instead of eight 4-byte ones - the pass moves 108 B per
    // sample, 1.4 GB for a dense 24 576-ray step); the planes start 4 n bytes apart: 16-byte aligned when n % 4 == 0, else the scalar tail loop takes all
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int64_t n4 = (n % 4 == 0) ? n / 4 : 0;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n4) {
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int g = 0; g < groups; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(part + (int64_t)g * n + 4 * t);
            a += v;
        }
        *reinterpret_cast<f32x4 *>(out + 4 * t) = a;
    }
    for (int64_t i = 4 * n4 + t; i < n; i += (int64_t)gridDim.x * 256) {
        float a = 0.0f;
        for (int g = 0; g < groups; ++g) a += part[(int64_t)g * n + i];
        out[i] = a;
    }
}

// ------------------------------------------------------------------- binned (atomic-free) backward
// d loss / d tables without global atomics.  Scattered fp32 global atomics run at ~6 G adds/s on
// MI355X (they execute at the memory side, one 64-B request per lane), which made the scatter 77 % of
// a train step.  Instead:
//   pass 1 (bin_kernel)   one workgroup = (tile of TS consecutive samples, level).  Every lane
//       recomputes its sample's vertices, the wave merges runs of equal vertex ids in adjacent lanes
//       (consecutive samples of a ray share their simplex on the coarse levels), and the surviving
//       (row, weighted gradient) entries are counting-sorted by table SLICE (slice = row >> shift,
//       2^shift rows = 64 KiB of fp32 accumulators) into the tile's fixed-size region of the workspace;
//       the per-slice offsets go to a header.  No global atomics, no counting pre-pass.
//   pass 2 (reduce_kernel) one workgroup = (level, slice): it owns that slice of the gradient
//       table in LDS, walks every tile's segment for its slice (coalesced reads, each entry read
//       exactly once), accumulates in 64-bit fixed point with ds_add_u64 (order-independent, hence
//       bitwise reproducible) and finally adds the slice to the table with plain coalesced stores.
#ifndef PAG_TS
#define PAG_TS 1024
#endif
#ifndef PAG_TS_HASH
#define PAG_TS_HASH 512
#endif
// samples per pass-1 tile (= threads per workgroup): 1024 with 4 vertices per level (permutohedral), 512 with 8 (hash) - the hash
// variant needs ~100 VGPRs, and two workgroups per CU (block barriers!) as well as a 32 KiB staging tile only fit with the smaller tile
constexpr int tile_samples(int nv) { return nv == 8 ? PAG_TS_HASH : PAG_TS; }
constexpr int SLICE_SHIFT = 13;   // upper bound; bin_plan() shrinks it so a slice's int64 accumulators fit 64 KiB
constexpr int NS_MAX = 256;       // slices per level supported (T <= 2^21)

struct BinLayout {
    uint32_t *keys;      // [L][ntiles][TS*NV]      row index inside its slice
    float *vals;         // [L][ntiles][TS*NV][F]   (PACK: u64 [L][ntiles][TS*NV], see pack_entry)
    uint32_t *header;    // [L][NS+1][ntiles]       exclusive offsets of each slice inside the tile region
    uint32_t *tile_max;  // [L][ntiles] bit pattern of max |gradient| of each (level, tile): no atomics, pass 2 reduces it
    int64_t ntiles;
    int NS, shift;
};

// bf16 gradients with F = 2 (the production path): one 8-byte entry = two 32-bit words, each a 26-bit float (fp32 with the low
// 6 mantissa bits rounded away - 2^-18 relative, far below the bf16 inputs' 2^-9) with half of the 12-bit row in the freed bits,
// instead of a 4-byte key + 8-byte value: a third less pass-1 -> pass-2 traffic and one store / load per entry instead of two.
// Word form (round 4; before: row | a << 12 | b << 38 in one 64-bit integer - the 64-bit shifts and ors were ~10 VALU instructions per
// entry in the bin pass and ~8 in the reduce pass): pack = 2 adds + 1 shift + 2 v_bfi_b32, unpack = 3 ands + 1 v_lshl_or_b32.  Same values.
#ifdef PAG_PACK_U64
__device__ __forceinline__ uint64_t pack_entry(uint32_t key12, float v0, float v1) {
    const uint64_t a = (uint64_t)((__float_as_uint(v0) + 0x20u) >> 6), b = (uint64_t)((__float_as_uint(v1) + 0x20u) >> 6);
    return (uint64_t)key12 | (a << 12) | (b << 38);
}
__device__ __forceinline__ void unpack_entry(uint64_t e, uint32_t &key12, float &v0, float &v1) {
    key12 = (uint32_t)e & 0xFFFu;
    v0 = __uint_as_float(((uint32_t)(e >> 12) & 0x3FFFFFFu) << 6);
    v1 = __uint_as_float((uint32_t)(e >> 38) << 6);
}
#else
__device__ __forceinline__ uint64_t pack_entry(uint32_t key12, float v0, float v1) {
    const uint32_t a = __float_as_uint(v0) + 0x20u, b = __float_as_uint(v1) + 0x20u;
    const uint32_t lo = (a & ~63u) | (key12 & 63u), hi = (b & ~63u) | (key12 >> 6);       // key12 < 4096: v_bfi_b32
    return (uint64_t)lo | ((uint64_t)hi << 32);                                          // a register pair: no instruction
}
__device__ __forceinline__ void unpack_entry(uint64_t e, uint32_t &key12, float &v0, float &v1) {
    const uint32_t lo = (uint32_t)e, hi = (uint32_t)(e >> 32);
    key12 = (lo & 63u) | ((hi & 63u) << 6);
    v0 = __uint_as_float(lo & ~63u);
    v1 = __uint_as_float(hi & ~63u);
}
#endif

// merge runs of equal keys in adjacent lanes: on return `emit` is set on the last lane of every run
// and that lane's v[] holds the run's sum.  Skipped (wave-uniformly) when the wave has few repeats.
// The segmented inclusive scan runs on DPP moves (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast:15 and
// row_bcast:31 to carry the row totals across) - VALU-rate lane crossings instead of ds_bpermute through the LDS.
// max over the wave on DPP moves; the result is valid in lane 63 (row_shr inside the 16-lane rows, row_bcast across them)
__device__ __forceinline__ uint32_t wave_max_to_lane63(uint32_t v) {
    v = max(v, (uint32_t)dpp_i<0x111, 0xF>((int)v));
    v = max(v, (uint32_t)dpp_i<0x112, 0xF>((int)v));
    v = max(v, (uint32_t)dpp_i<0x114, 0xF>((int)v));
    v = max(v, (uint32_t)dpp_i<0x118, 0xF>((int)v));
    v = max(v, (uint32_t)dpp_i<0x142, 0xA>((int)v));
    v = max(v, (uint32_t)dpp_i<0x143, 0xC>((int)v));
    return v;
}
template <int F>
// pair = this lane and its predecessor both carry an entry (live && lane > 0; consecutive lanes are consecutive samples, so a live lane's
// predecessor is live), pairmask = its ballot: hoisted by the caller.  The ballot of the bare compare is the compare's own result
// register; the ballot of a conjunction costs a select and a second compare.
__device__ __forceinline__ void run_combine(uint32_t key, bool pair, unsigned long long pairmask, bool live, float (&v)[F], bool &emit, int lane) {
    const uint32_t prev = (uint32_t)dpp_i<0x138, 0xF>((int)key);             // wave_shr:1
    const bool eq = prev == key;
    const unsigned long long m = __ballot(eq) & pairmask;
    const bool same = eq && pair;
    emit = live;
    // 32-bit counts: the scalar unit has no ordered 64-bit compare, and a 64-bit popcount compare ends up on the vector unit
    if (__builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32)) < 8) return;
#ifdef PAG_BIN_FAKE_COMBINE
    {
        const bool ns = (m >> ((lane + 1) & 63)) & 1ull;
        emit = live && (lane == 63 || !ns);
        return;
    }
#endif
    if constexpr (F == 2) {
        // The production width.  The head flag travels as nf = 1.0 (no head between source and this lane) / 0.0, which turns one step
        // into v += dpp(v) * nf ; nf *= dpp(nf): three DPP-form instructions (v_fmac_f32_dpp / v_mul_f32_dpp) instead of a DPP move,
        // a compare and an add + select per feature.  The multiplier is exactly 1 or 0, so the sums round like the plain additions
        // of seg_step.  No bound_ctrl: a lane whose source is outside its row (or masked rows of the row_bcast steps) keeps v and nf.
        // Every DPP source was written at least three instructions earlier (the two wait states the hardware wants); the leading
        // s_nop covers values the compiler produced right before the block.
        float nf = same ? 1.0f : 0.0f;
#define PAG_SEG_STEP(ctl) \
        "v_fmac_f32_dpp %0, %0, %2 " ctl "\n\tv_fmac_f32_dpp %1, %1, %2 " ctl "\n\tv_mul_f32_dpp %2, %2, %2 " ctl "\n\t"
        asm("s_nop 4\n\t"
            PAG_SEG_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
            PAG_SEG_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
            PAG_SEG_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
            PAG_SEG_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
            PAG_SEG_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
            "v_fmac_f32_dpp %0, %0, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
            "v_fmac_f32_dpp %1, %1, %2 row_bcast:31 row_mask:0xc bank_mask:0xf"
            : "+v"(v[0]), "+v"(v[1]), "+v"(nf));
#undef PAG_SEG_STEP
        const bool next_same2 = (m >> ((lane + 1) & 63)) & 1ull;
        emit = live && (lane == 63 || !next_same2);
        return;
    }
    int f = same ? 0 : 1;                 // run head
    seg_step<F, 0x111, 0xF>(v, f);        // row_shr:1
    seg_step<F, 0x112, 0xF>(v, f);        // row_shr:2
    seg_step<F, 0x114, 0xF>(v, f);        // row_shr:4
    seg_step<F, 0x118, 0xF>(v, f);        // row_shr:8
    seg_step<F, 0x142, 0xA>(v, f);        // row_bcast:15 -> rows 1, 3
    seg_step<F, 0x143, 0xC>(v, f);        // row_bcast:31 -> rows 2, 3
    const bool next_same = (m >> ((lane + 1) & 63)) & 1ull;
    emit = live && (lane == 63 || !next_same);
}

#ifdef PAG_BIN_TIMING      // experiment builds only (scripts/bin_phases.py): wave 0 / wave 15 of every workgroup stamp the phases into LDS, flushed at the end
__device__ unsigned long long pag_bin_times[32768 * 32];
#define PAG_BSTAMP(k) do { if (lane == 0 && (wave == 0 || wave == 15)) pag_bstamps[wave ? 1 : 0][k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int pag_debug_bin_times(void *dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pag_bin_times), bytes); }
#else
#define PAG_BSTAMP(k)
#endif

template <int KIND /*0 hash, 1 permuto*/, typename GradT, int F, int LPX, bool PACK>
// TWO 1024-thread workgroups per CU (8 waves per SIMD, <= 64 VGPRs): with one, every block barrier of the counting sort / staging
// stalls the whole CU - nothing else is resident to run.  The permutohedral variant fits 64 VGPRs with 48 B of scratch and the encode
// backward drops from 2.18 to 1.75 ms per step; the hash variant (8 vertices) would spill 140 - 300 B and is measured separately.
#ifndef PAG_BIN_WAVES
#define PAG_BIN_WAVES 8
#endif
#ifndef PAG_BIN_WAVES_HASH
#define PAG_BIN_WAVES_HASH 4
#endif
__global__ __launch_bounds__(tile_samples(KIND == 0 ? 8 : 4), (KIND == 1 ? PAG_BIN_WAVES : PAG_BIN_WAVES_HASH)) void bin_kernel(const float *__restrict__ xyz, int64_t M, const GradT *__restrict__ go,
                                                 int64_t sm, int64_t sc, int grouped, HashParams hp, PermutoParams pp, BinLayout lay) {
    PAG_BLOCK_TIMER(0);
    constexpr int NV = KIND == 0 ? 8 : 4;
    constexpr int TS = tile_samples(NV);
#ifndef PAG_BIN_NO_STAGE
    constexpr bool STAGE = PACK && TS * NV * 8 <= 32768;      // packed 8-byte entries, tile fits 32 KiB of LDS (permutohedral: 4 vertices)
#else
    constexpr bool STAGE = false;
#endif
    __shared__ uint64_t stage[STAGE ? TS * NV : 1];
    __shared__ uint32_t cnt[LPX][NS_MAX + 2];     // [.][NS_MAX + 1] = max |g| of this (tile, level)
    __shared__ uint32_t offs[LPX][NS_MAX + 1];
    const int L = KIND == 0 ? hp.L : pp.L;
    const int64_t tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PAG_BIN_TIMING
    __shared__ unsigned long long pag_bstamps[2][16];
    if (lane == 0 && (wave == 0 || wave == 15)) pag_bstamps[wave ? 1 : 0][13] = __builtin_amdgcn_s_memrealtime();
#endif
    PAG_BSTAMP(0);
    const int64_t i = tile * TS + tid;
    const bool live = i < M;
    const int64_t ic = live ? i : M - 1;
    float x[3];
    load_xyz(xyz, ic, KIND == 0 ? hp.half_coords : pp.half_coords, x);
    // grouped (LPX = ceil(L/8)): blockIdx.y = XCD group g, levels g, g+8, ... share ONE counting sort and the 16-byte
    // gradient piece is read once.  strided (LPX = 1): blockIdx.y = level.
    float gpiece[8];
    bool wave_live = true;
    if (grouped) {
        typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t t = *reinterpret_cast<const bf16x8_t *>(reinterpret_cast<const bf16_t *>(go) + ((int64_t)blockIdx.y * M + ic) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) gpiece[e] = (float)t[e];
        // A wave whose 64 samples all carry an exactly-zero gradient piece (empty space of a trained scene: sigma = relu(pre) = 0, so neither the
        // colour nor the density path sends anything back; the filler samples of a padded batch) would emit entries whose values are all +-0: they
        // add nothing in pass 2 (fixed-point sums).  Such a wave skips the lattice, the merges and the counting of all its levels and only keeps
        // the workgroup's barriers (scripts/zero_weight_tiles.py: 85 % of the 32-sample tiles of the trained analytic scene).  Wave-uniform on purpose:
        // the merge of adjacent lanes relies on consecutive lanes being consecutive entries.
        wave_live = __ballot(live && piece_nonzero(t)) != 0ull;
    }
    const float *scale = KIND == 0 ? hp.scale : pp.scale;
    const bool has_scale = KIND == 0 ? hp.has_scale : pp.has_scale;
    for (int s = tid; s < LPX * (NS_MAX + 2); s += TS) (&cnt[0][0])[s] = 0;
    __syncthreads();
#ifdef PAG_BIN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(gpiece[0]) : : "memory");
#endif
    PAG_BSTAMP(1);
    uint32_t idx[LPX][NV];
    float ev[LPX][NV][F];
    bool emit[LPX][NV];
    uint32_t rank[LPX][NV];
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        const int level = grouped ? xcd8_level((int)blockIdx.y, j) : (int)blockIdx.y;
        const bool lv = level < L;
        const int lc = lv ? level : L - 1;
        if (!wave_live) {
#pragma unroll
            for (int k = 0; k < NV; ++k) emit[j][k] = false;
            continue;
        }
        float w[NV];
        float gv[F];
        if (KIND == 0) {
            float w3[3];
            uint32_t id8[8];
            hash_cell(x, hp.res[lc], hp.log2T, id8, w3);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                idx[j][k] = id8[k & 7];
                w[k] = ((k & 4) ? w3[0] : 1.0f - w3[0]) * ((k & 2) ? w3[1] : 1.0f - w3[1]) * ((k & 1) ? w3[2] : 1.0f - w3[2]);
            }
        } else {
            uint32_t id4[4];
            float b4[4];
#ifndef PAG_BIN_NO_LDS_SORT
            if constexpr (STAGE)      // the staging tile is idle until the placement phase: 32 bytes of it per lane serve the rank sort
                permuto_simplex_lds(x, pp.shift[lc], pp.sf[lc], pp.capacity, pp.pow2mask, id4, b4, reinterpret_cast<float *>(stage) + tid * 8);
            else
#endif
                permuto_simplex(x, pp.shift[lc], pp.sf[lc], pp.capacity, pp.pow2mask, id4, b4);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                idx[j][k] = id4[k & 3];
                w[k] = b4[k & 3];
            }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if (grouped) {
                gv[f] = gpiece[(j * F + f) & 7];
            } else {
                gv[f] = pag_ld(go + ic * sm + (int64_t)(lc * F + f) * sc);
            }
            if (has_scale) gv[f] *= scale[lc * F + f];
        }
        {   // per-(tile, level) max |g| (positive floats order like their bit patterns): feeds the fixed-point scale of pass 2
            float mx = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) mx = fmaxf(mx, (live && lv) ? fabsf(gv[f]) : 0.0f);
            uint32_t mb = __float_as_uint(mx);
            if (mx != mx) mb = 0x7FC00000u;   // NaN poisons the level
            mb = wave_max_to_lane63(mb);
            if (lane == 63 && mb) atomicMax(&cnt[j][NS_MAX + 1], mb);    // LDS, one per wave
        }
        const bool pair_l = live && lv && lane > 0;
        const unsigned long long pair_m = __ballot(pair_l);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int f = 0; f < F; ++f) ev[j][k][f] = gv[f] * w[k];
            const bool lk = live && lv;
#ifdef PAG_BIN_NO_COMBINE
            emit[j][k] = lk;
#else
            run_combine<F>(idx[j][k], pair_l, pair_m, lk, ev[j][k], emit[j][k], lane);
#endif
            if (emit[j][k]) rank[j][k] = atomicAdd(&cnt[j][idx[j][k] >> lay.shift], 1u);      // read only where emit is set
        }
        PAG_BSTAMP(2 + j);
    }
    __syncthreads();
    PAG_BSTAMP(6);
    if (wave < LPX) {   // wave j: exclusive prefix over level j's NS slice counters
        uint32_t carry = 0;
        for (int s0 = 0; s0 < lay.NS; s0 += 64) {
            const int s = s0 + lane;
            uint32_t c = s < lay.NS ? cnt[wave][s] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (s < lay.NS) offs[wave][s] = carry + incl - c;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) offs[wave][lay.NS] = carry;
    }
    __syncthreads();
    PAG_BSTAMP(7);
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        const int level = grouped ? xcd8_level((int)blockIdx.y, j) : (int)blockIdx.y;
        if (level >= L) break;
        const int64_t region = ((int64_t)level * lay.ntiles + tile) * (TS * NV);
        if constexpr (STAGE) {
            // The sorted entries of this (tile, level) occupy one contiguous run of the region, but a lane's entry lands anywhere in
            // it: written straight to memory that is one 8-byte partial-line request per entry (the L2 takes ~16 requests per clock
            // and XCD - the write phase was about half of this kernel).  Staged through LDS the run leaves as full 512-byte wave stores.
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (emit[j][k])
                    stage[offs[j][idx[j][k] >> lay.shift] + rank[j][k]] =
                        pack_entry(idx[j][k] & ((1u << lay.shift) - 1u), ev[j][k][0], ev[j][k][F - 1]);
            __syncthreads();
            if (j == 0) PAG_BSTAMP(8);
            const uint32_t total = offs[j][lay.NS];
            uint64_t *dst = reinterpret_cast<uint64_t *>(lay.vals) + region;
            for (uint32_t q = tid; q < total; q += TS) dst[q] = stage[q];
            if (j == 0) PAG_BSTAMP(9);
            __syncthreads();      // the next level reuses the staging tile
            if (j == 0) PAG_BSTAMP(10);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!STAGE && emit[j][k]) {
                const uint32_t s = idx[j][k] >> lay.shift;
                const int64_t pos = region + offs[j][s] + rank[j][k];
                if constexpr (PACK) {
                    reinterpret_cast<uint64_t *>(lay.vals)[pos] = pack_entry(idx[j][k] & ((1u << lay.shift) - 1u), ev[j][k][0], ev[j][k][F - 1]);
                } else {
                    lay.keys[pos] = idx[j][k] & ((1u << lay.shift) - 1u);
#pragma unroll
                    for (int f = 0; f < F; ++f) lay.vals[pos * F + f] = ev[j][k][f];
                }
            }
        }
        for (int s = tid; s <= lay.NS; s += TS) lay.header[((int64_t)level * (lay.NS + 1) + s) * lay.ntiles + tile] = offs[j][s];
        if (tid == 0) lay.tile_max[(int64_t)level * lay.ntiles + tile] = cnt[j][NS_MAX + 1];
    }
#ifdef PAG_BIN_TIMING
    PAG_BSTAMP(11);
    if (lane == 0 && (wave == 0 || wave == 15)) {
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.y;
        pag_bstamps[wave ? 1 : 0][14] = __builtin_amdgcn_s_memrealtime();
        if (b < 32768) for (int k = 0; k < 16; ++k) pag_bin_times[b * 32 + (wave ? 16 : 0) + k] = pag_bstamps[wave ? 1 : 0][k];
    }
#endif
}

// Add one wave's 64 (key, value) entries into the LDS slice.  LDS float atomics (ds_add_f32) serialise
// at ~3 cycles per active lane on gfx950 (194 cycles per wave-instruction measured) while 64-bit integer
// adds run at ~12 cycles per wave-instruction, so the slice is accumulated in 2^-S fixed point with
// ds_add_u64: S is chosen per level from the largest |gradient| so that 2^23 worst-case addends cannot
// overflow.  Integer addition is associative: the result does not depend on arrival order (bitwise
// reproducible).  Rows that several lanes share (coarse levels) are first summed across the wave.
template <int F>
__device__ __forceinline__ void lds_accumulate(unsigned long long *acc, uint32_t key, const long long (&val)[F], bool valid, int lane, bool distinct) {
    unsigned long long active = __ballot(valid);
#pragma unroll 1
    for (int it = 0; it < 8 && active && !distinct; ++it) {
        const int leader = __ffsll((long long)active) - 1;
        const uint32_t lk = __shfl(key, leader);
        const unsigned long long m = __ballot(valid && key == lk) & active;
        if (__popcll(m) < 8) break;
        const bool mine = (m >> lane) & 1ull;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            long long v = mine ? val[f] : 0ll;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            if (lane == leader) atomicAdd(&acc[lk * F + f], (unsigned long long)v);
        }
        active &= ~m;
    }
    if ((active >> lane) & 1ull) {
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(&acc[key * F + f], (unsigned long long)val[f]);
    }
}

// One entry as it sits in registers while its load is in flight: the packed form stays raw (decoding it would wait for the load)
template <int F, bool PACK>
struct RawEntry {
    uint64_t e;
    uint32_t key_;
    float val_[PACK ? 1 : F];
    __device__ __forceinline__ void load(const BinLayout &lay, int64_t pos) {
        if constexpr (PACK) {
            e = reinterpret_cast<const uint64_t *>(lay.vals)[pos];
        } else {
            key_ = lay.keys[pos];
#pragma unroll
            for (int f = 0; f < F; ++f) val_[f] = lay.vals[pos * F + f];
        }
    }
    __device__ __forceinline__ void decode(uint32_t &key, float (&val)[F]) const {
        if constexpr (PACK) {
            unpack_entry(e, key, val[0], val[F - 1]);
        } else {
            key = key_;
#pragma unroll
            for (int f = 0; f < F; ++f) val[f] = val_[f];
        }
    }
};

// value -> 2^-S fixed point.  Exact 64-bit conversion on the fp32-gradient path; on the packed bf16 path (entries already
// rounded to 18 mantissa bits) a 32-bit convert of v * 2^(S-14) followed by a 14-bit shift: the quantum becomes
// max|g| * 2^-24 per addend - far below the inputs' bf16 noise - and the software float -> int64 sequence, a third of this
// kernel's VALU instructions, disappears.
template <bool PACK>
__device__ __forceinline__ long long to_fixed(float v, int S) {
    if constexpr (PACK) return (long long)__float2int_rn(ldexpf(v, S - 14)) << 14;
    else return __float2ll_rn(ldexpf(v, S));
}

#ifdef PAG_REDUCE_TIMING      // experiment builds only (scripts/reduce_phases.py): lane 0 of wave 0 / wave 15 of every block stamps the phases.
// The stamps (shader clock; slots 13 / 14 the 100 MHz clock all CUs share) wait in LDS until the end of the kernel: kept in registers
// they push the kernel past 64 VGPRs and halve its occupancy, stored to memory as they are taken they join the vmcnt waits.
__device__ unsigned long long pag_dbg_times[8192 * 32];
#define PAG_STAMP(k) do { if (lane == 0 && (wave == 0 || wave == 15)) pag_stamps[wave ? 1 : 0][k] = __builtin_amdgcn_s_memtime(); } while (0)
#define PAG_STAMP_RT(k) do { if (lane == 0 && (wave == 0 || wave == 15)) pag_stamps[wave ? 1 : 0][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int pag_debug_reduce_times(void *dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(pag_dbg_times), bytes); }
#else
#define PAG_STAMP(k)
#define PAG_STAMP_RT(k)
#endif
// Experiment switches (scripts/build_variant.sh; profiles/README.md round 4 has the A/B numbers):
//   PAG_RED_SGPRS  SGPR budget.  With 81 SGPRs (what the compiler takes when left alone: next_free_sgpr 75 + 6) a gfx950 SIMD holds 7 waves of this
//                  kernel, not 8 - the runtime's occupancy query still answers "2 workgroups per CU", the hardware runs ONE 1024-thread workgroup per
//                  CU (scripts/exp/lds_occupancy.hip: the step is between next_free_sgpr 74 and 75).  80 costs no spill and no VGPR.
//   PAG_RED_ORDER  0 = blocks in (level, slice) order, 1 = finest level first, 2 = + each XCD takes 8 neighbouring slices, 3 = + the coarsest
//                  PAG_RED_HEAD levels stay in front.
#ifndef PAG_RED_SGPRS
#define PAG_RED_SGPRS 80
#endif
#ifndef PAG_RED_ORDER
#define PAG_RED_ORDER 3
#endif
#ifndef PAG_RED_HEAD
#define PAG_RED_HEAD 8
#endif

template <int F, int NV, bool PACK>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(PAG_RED_SGPRS))) void reduce_kernel(BinLayout lay, int64_t rows_per_level, float *__restrict__ gtab, int overwrite) {
    PAG_BLOCK_TIMER(3);
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];      // [2^shift][F] fixed point
    constexpr int TS = tile_samples(NV);
    // Block -> (level, slice).  Blocks are handed out in blockIdx order and run for very different times: the fine levels hold most of the
    // entries (a block of level 20 streams 1 MB and runs ~100 us next to its neighbours, one of level 8 ~20 us), and a few slices of the
    // COARSEST levels - the ones that hold the rows every ray touches - run 60 - 100 us although their level's average is 12 us.  In
    // (level, slice) order the launch ended on a tail of the longest blocks (the last 60 us ran half empty).  Now: the coarsest levels
    // first (their long blocks start at t = 0, in the shadow of everything else), then the remaining levels from the finest down, so that
    // the launch ends on the short blocks of the middle levels: 179 -> 153 us per launch on the bench workload.
    // Workgroups are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8): each XCD takes 1/8 of a level's slices as ONE contiguous
    // run, because neighbouring slices' segments are neighbours in every tile region (~450 B each, unaligned) and a 128-byte line they
    // share is then fetched into one L2 once instead of into two.
    int level, slice;
    {
        const int nl = (int)(gridDim.x / lay.NS);
        int q;                                                  // position in the launch order -> q-th (level, slice) pair, slices fastest
        if (PAG_RED_ORDER >= 2 && lay.NS % 8 == 0) {
            const int per = lay.NS / 8, x = blockIdx.x & 7, j = blockIdx.x >> 3;
            q = (j / per) * lay.NS + x * per + j % per;
        } else {
            q = blockIdx.x;
        }
        const int k = q / lay.NS;
        slice = q % lay.NS;
        const int head = PAG_RED_ORDER == 3 ? min(PAG_RED_HEAD, nl / 3) : 0;      // the coarsest levels keep their place at the front
        level = PAG_RED_ORDER == 0 ? k : k < head ? k : nl - 1 - (k - head);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int slice_rows = 1 << lay.shift;
    __shared__ uint32_t lvl_max;
#ifdef PAG_REDUCE_TIMING
    __shared__ unsigned long long pag_stamps[2][16];
#endif
    PAG_STAMP_RT(13);
    PAG_STAMP(0);
    // The level's largest |gradient| (the fixed-point scale) from the bin pass's per-tile maxima.  The accumulators are cleared while those loads
    // are in flight and under the same barrier that publishes lvl_max = 0: one barrier and one exposed load round trip less per workgroup than
    // max-then-clear (the fixed part of a workgroup is what the small batches of the post-prune regime pay: 1536 workgroups whatever M is).
    if (tid == 0) lvl_max = 0;
    {
        uint32_t mb = 0;
        for (int64_t t = tid; t < lay.ntiles; t += blockDim.x) mb = max(mb, lay.tile_max[(int64_t)level * lay.ntiles + t]);
        for (int j = tid; j < slice_rows * F; j += blockDim.x) acc[j] = 0ull;
        __syncthreads();
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, d));
        if (lane == 0 && mb) atomicMax(&lvl_max, mb);
    }
    __syncthreads();
    const float maxabs = __uint_as_float(lvl_max);
    if (maxabs == 0.0f) {                             // no gradient reached this level
        if (overwrite) {                              // the caller did not zero the table: this slice's rows are ours to clear
            const int64_t row0z = (int64_t)slice * slice_rows;
            float *dz = gtab + ((int64_t)level * rows_per_level + row0z) * F;
            const int64_t nz = min((int64_t)slice_rows, rows_per_level - row0z) * F;
            for (int64_t j = tid; j < nz; j += blockDim.x) dz[j] = 0.0f;
        }
        return;
    }
    const bool poisoned = !(maxabs <= 3.4e38f);       // inf / NaN upstream: propagate NaN
    int ex;
    frexpf(poisoned ? 1.0f : maxabs, &ex);            // maxabs < 2^ex
    const int S = 38 - ex;                            // |val| * 2^S < 2^38 ; 2^23 addends stay below 2^61
    PAG_STAMP(1);
    PAG_STAMP(2);
    const uint32_t *hb = lay.header + ((int64_t)level * (lay.NS + 1) + slice) * lay.ntiles;
    const uint32_t *he = hb + lay.ntiles;
    // A (tile, slice) segment holds ~20 entries on average (the fine levels fill all 64 slices evenly), so walking one
    // segment per wave-iteration left two thirds of the lanes idle.  Instead the segments of 64 consecutive tiles are
    // treated as ONE concatenated stream: lane l takes stream element 64*c + l, finds its tile by a 6-step binary search
    // over the wave's exclusive prefix of segment lengths (shuffles) and reads that tile's region at the right offset.
    // With few tiles (small batches: the post-prune voxel regime has ~300) groups of 64 would leave most of the 16 waves without any:
    // the group shrinks so that every wave gets tiles (lanes >= G hold empty segments; the search below is unchanged).
    // Where the segments are long (the fine levels, which hold most of the entries: ~64 per segment) a 64-element chunk of the stream
    // crosses at most a few tile boundaries, and the search collapses to three compares against the next boundaries, read with
    // wave-uniform LDS addresses from a per-wave copy of the prefix (s_excl) and of  tile * region + begin - prefix  (s_adj: position
    // + s_adj = offset of the element inside the group's regions).  `ts` (uniform) is a tile at or before the chunk's first
    // element; chunks with a fourth boundary (short segments: the coarse levels) take the binary search.
    __shared__ uint32_t s_excl[16][72], s_adj[16][72];
    const int G = (int)min((int64_t)64, max((int64_t)1, (lay.ntiles + nwaves - 1) / nwaves));
    for (int64_t t0 = (int64_t)wave * G; t0 < lay.ntiles; t0 += (int64_t)nwaves * G) {
        const int64_t tl = t0 + lane;
        const bool has = lane < G && tl < lay.ntiles;
        const uint32_t mb = has ? hb[tl] : 0u, me = has ? he[tl] : 0u;
        uint32_t incl = me - mb;
#ifdef PAG_REDUCE_TIMING
        const bool first_group = t0 == (int64_t)wave * G;      // the in-loop stamps describe a wave's first group of tiles
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(incl) : : "memory");
        if (first_group) PAG_STAMP(7);
#endif
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += up;
        }
        const uint32_t excl = incl - (me - mb);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total == 0u) continue;          // nothing of this slice in these tiles (small batches: most groups) - before any load is issued
        const int64_t region0 = ((int64_t)level * lay.ntiles + t0) * (TS * NV);
        s_excl[wave][lane] = excl;
        s_adj[wave][lane] = (uint32_t)lane * (uint32_t)(TS * NV) + mb - excl;      // modulo 2^32: only position + s_adj is used
        if (lane < 8) s_excl[wave][64 + lane] = 0xFFFFFFFFu;                        // "no further boundary"
        // The wave-wide pre-summation of shared rows (lds_accumulate's leader loop) pays where MANY lanes of a 64-entry chunk carry one row: the
        // coarse levels, whose (tile, slice) segments hold a few entries each, so that a chunk gathers the same hot rows from dozens of tiles.
        // Where the segments average >= 12 entries a chunk spans ~5 tiles and - adjacent repeats having been merged inside each tile by the bin
        // pass - a row appears a handful of times at most: the loop's test (a leader key through the LDS crossbar, a ballot, a count per chunk)
        // can only fail there, and is skipped.  (Before: only levels still holding >= 15/16 of a tile's entries took this path - level 23 alone
        // on the bench rays; 155 -> 152 us.)  The sums are integers: identical either way.
#ifndef PAG_RED_DISTINCT_SEG
#define PAG_RED_DISTINCT_SEG 12
#endif
        const int tiles_here = (int)min((int64_t)G, lay.ntiles - t0);
        const bool distinct = total >= (uint32_t)(PAG_RED_DISTINCT_SEG) * (uint32_t)tiles_here;
        int ts = 0;
        auto fetch = [&](uint32_t c0, RawEntry<F, PACK> &raw) __attribute__((always_inline)) {
            const uint32_t i = c0 + lane;
            const bool ok = i < total;
            const uint32_t *pe = &s_excl[wave][ts];
            const uint32_t b1 = pe[1], b2 = pe[2], b3 = pe[3], b4 = pe[4];           // identical in every lane
            int64_t pos;
            int t;
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)b4) >= c0 + 64u) {
                const uint32_t *pa = &s_adj[wave][ts];
                uint32_t a = pa[0];
                a = i >= b1 ? pa[1] : a;
                a = i >= b2 ? pa[2] : a;
                a = i >= b3 ? pa[3] : a;
                t = ts + (i >= b1 ? 1 : 0) + (i >= b2 ? 1 : 0) + (i >= b3 ? 1 : 0);
                pos = region0 + (int64_t)(uint32_t)(i + a);
            } else {
                t = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int cand = t + step;
                    const uint32_t pc = (uint32_t)__shfl((int)excl, cand & 63);
                    if (pc <= i) t = cand;                   // cand <= 63 always: t + step never exceeds 63
                }
                const uint32_t bt = (uint32_t)__shfl((int)mb, t), et = (uint32_t)__shfl((int)excl, t);
                pos = region0 + (int64_t)t * (TS * NV) + bt + (i - et);
            }
            ts = __builtin_amdgcn_readlane(t, 63);           // a tile at or before the next chunk's first element (only used when one follows)
            raw.load(lay, ok ? pos : region0);               // lanes past the end read the group's first entry (always inside the workspace) and ignore it: no branch around the load
            return ok;
        };
        // TWO chunks in flight: the kernel is latency-bound (VALU ~45 % busy at the full 8 waves per SIMD, 70 % of the wave cycles waiting),
        // and one iteration of accumulate work did not cover the latency of the next chunk's load.  Ping-pong between two named
        // registers sets (a rotation by copies would wait for the newer load at the copy) and every refill is issued unconditionally
        // (past the end it re-reads the group's first entry): the wait counts stay exact at every join.
        RawEntry<F, PACK> raw_a, raw_b;
        bool ok_a = fetch(0, raw_a);
        bool ok_b = fetch(64, raw_b);
#ifdef PAG_REDUCE_TIMING
        if (first_group) PAG_STAMP(3);
#endif
        auto stage = [&](RawEntry<F, PACK> &raw, bool &ok_r, uint32_t c_next) __attribute__((always_inline)) {
            uint32_t key;
            float fv[F];
            raw.decode(key, fv);
            const bool ok = ok_r;
            // decode first, for real (the empty asm pins the decoded values and, with its memory clobber, keeps the refill below it):
            // the refill then reuses the raw registers instead of landing in new ones that are copied - after a full wait - at the loop edge
            asm volatile("" : "+v"(key) : : "memory");
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("" : "+v"(fv[f]) : : "memory");
            ok_r = fetch(c_next, raw);
            long long val[F];
#pragma unroll
            for (int f = 0; f < F; ++f) val[f] = ok ? to_fixed<PACK>(fv[f], S) : 0ll;
            lds_accumulate<F>(acc, key, val, ok, lane, distinct);
        };
        for (uint32_t c0 = 0; c0 < total; c0 += 128) {
            stage(raw_a, ok_a, c0 + 128);
            stage(raw_b, ok_b, c0 + 192);          // past the end: ok_b is false, nothing is added
        }
#ifdef PAG_REDUCE_TIMING
        if (first_group) {
            PAG_STAMP(9);
            if (lane == 0 && (wave == 0 || wave == 15)) pag_stamps[wave ? 1 : 0][10] = total;
        }
#endif
    }
    PAG_STAMP(4);
    __syncthreads();
    PAG_STAMP(5);
    const int64_t row0 = (int64_t)slice * slice_rows;
    float *dst = gtab + ((int64_t)level * rows_per_level + row0) * F;
    const int64_t valid = min((int64_t)slice_rows, rows_per_level - row0) * F;
    for (int64_t j = tid; j < valid; j += blockDim.x) {
        const float v = poisoned ? __uint_as_float(0x7FC00000u) : (float)ldexp((double)(long long)acc[j], -S);
        dst[j] = overwrite ? v : dst[j] + v;
    }
#ifdef PAG_REDUCE_TIMING
    PAG_STAMP(6);
    PAG_STAMP_RT(14);
    if (lane == 0 && (wave == 0 || wave == 15) && blockIdx.x < 8192) {
        pag_stamps[wave ? 1 : 0][11] = (unsigned long long)level;
        pag_stamps[wave ? 1 : 0][12] = (unsigned long long)slice;
        pag_stamps[wave ? 1 : 0][15] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);      // HW_ID, XCC_ID
        for (int k = 0; k < 16; ++k) pag_dbg_times[blockIdx.x * 32 + (wave ? 16 : 0) + k] = pag_stamps[wave ? 1 : 0][k];
    }
#endif
}

#ifdef PAG_REDUCE_TIMING
extern "C" int pag_debug_reduce_occupancy(int threads, int dyn_bytes) {
    int occ = -1;
    hipFuncSetAttribute((const void *)reduce_kernel<2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_bytes);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reduce_kernel<2, 4, true>, threads, dyn_bytes);
    return occ;
}
#endif

struct BinPlan {
    int64_t ntiles, keys_bytes, vals_bytes, header_bytes, total;
    int NS, shift;
};
inline BinPlan bin_plan(int64_t M, int L, int F, int NV, int64_t rows) {
    BinPlan b;
    const int TS = tile_samples(NV);
    b.shift = SLICE_SHIFT;
    while (b.shift > 0 && ((int64_t)1 << b.shift) * F * 8 > 65536) --b.shift;      // int64 accumulators, 64 KiB of LDS
    b.NS = (int)((rows + ((int64_t)1 << b.shift) - 1) >> b.shift);
    b.ntiles = (M + TS - 1) / TS;
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    b.keys_bytes = up((int64_t)L * b.ntiles * TS * NV * 4);
    b.vals_bytes = up((int64_t)L * b.ntiles * TS * NV * F * 4);
    b.header_bytes = up((int64_t)L * (b.NS + 1) * b.ntiles * 4);
    b.total = b.keys_bytes + b.vals_bytes + b.header_bytes + up((int64_t)L * b.ntiles * 4);   // + tile_max[L][ntiles]
    return b;
}

template <int KIND>
int launch_binned(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t sm, int64_t sc, int grouped, int L, int F,
                  int64_t rows, const HashParams &hp, const PermutoParams &pp, float *gtab, void *workspace,
                  int64_t workspace_bytes, hipStream_t st, const char *name, bool overwrite = false) {
    constexpr int NV = KIND == 0 ? 8 : 4;
    const BinPlan b = bin_plan(M, L, F, NV, rows);
    PAG_CHECK_ARG(workspace_bytes >= b.total, "%s: workspace %lld B < required %lld B", name, (long long)workspace_bytes, (long long)b.total);
    PAG_CHECK_ARG(b.NS <= NS_MAX, "%s: table too large for the binned backward (%d slices > %d)", name, b.NS, NS_MAX);
    PAG_CHECK_ARG(F == 2 || F == 4 || F == 1, "%s: n_feat", name);
    BinLayout lay;
    char *wsp = (char *)workspace;
    lay.keys = (uint32_t *)wsp;
    lay.vals = (float *)(wsp + b.keys_bytes);
    lay.header = (uint32_t *)(wsp + b.keys_bytes + b.vals_bytes);
    lay.tile_max = (uint32_t *)(wsp + b.keys_bytes + b.vals_bytes + b.header_bytes);
    lay.ntiles = b.ntiles;
    lay.NS = b.NS;
    lay.shift = b.shift;
    dim3 g1((unsigned)b.ntiles, (unsigned)(grouped ? (L < 8 ? L : 8) : L)), g2((unsigned)(L * b.NS));
    const size_t lds = ((size_t)1 << b.shift) * F * sizeof(unsigned long long);
    const int lpx = grouped ? (L + 7) / 8 : 1;
#define BIN_LAUNCH1(GT, F_, LPX_)                                                                                       \
    hipLaunchKernelGGL((bin_kernel<KIND, GT, F_, LPX_, (sizeof(GT) == 2 && F_ == 2)>), g1, dim3(tile_samples(NV)), 0, st, xyz, M, (const GT *)grad_out, sm, sc, grouped, hp, pp, lay)
#define BIN_LAUNCH(GT, F_)                                                                                              \
    do {                                                                                                                \
        if (lpx == 1) BIN_LAUNCH1(GT, F_, 1);                                                                           \
        else if (lpx == 2) BIN_LAUNCH1(GT, F_, 2);                                                                      \
        else if (lpx == 3) BIN_LAUNCH1(GT, F_, 3);                                                                      \
        else BIN_LAUNCH1(GT, F_, 4);                                                                                    \
        hipLaunchKernelGGL((reduce_kernel<F_, NV, (sizeof(GT) == 2 && F_ == 2)>), g2, dim3(1024), lds, st, lay, rows, gtab, overwrite ? 1 : 0); \
    } while (0)
    if (grad_dtype == PAG_F32) {
        if (F == 2) BIN_LAUNCH(float, 2);
        else if (F == 4) BIN_LAUNCH(float, 4);
        else BIN_LAUNCH(float, 1);
    } else {
        if (F == 2) BIN_LAUNCH(bf16_t, 2);
        else if (F == 4) BIN_LAUNCH(bf16_t, 4);
        else BIN_LAUNCH(bf16_t, 1);
    }
#undef BIN_LAUNCH
#undef BIN_LAUNCH1
    return PAG_OK;
}

inline unsigned encode_grid(int64_t M) { return (unsigned)(((M + PAG_ENC_FWD_THREADS - 1) / PAG_ENC_FWD_THREADS) * 8); }

// ---- dispatch helpers: (table dtype, out dtype, F, LPX) -> kernel instantiation
#define PAG_DISPATCH_F_LPX(F_, LPX_, CALL)                 \
    if (n_feat == F_ && lpx == LPX_) {                     \
        constexpr int F = F_;                              \
        constexpr int LPX = LPX_;                          \
        CALL;                                              \
        launched = true;                                   \
    }
#define PAG_DISPATCH_ALL(CALL)          \
    PAG_DISPATCH_F_LPX(2, 1, CALL)      \
    PAG_DISPATCH_F_LPX(2, 2, CALL)      \
    PAG_DISPATCH_F_LPX(2, 3, CALL)      \
    PAG_DISPATCH_F_LPX(2, 4, CALL)      \
    PAG_DISPATCH_F_LPX(4, 1, CALL)      \
    PAG_DISPATCH_F_LPX(4, 2, CALL)      \
    PAG_DISPATCH_F_LPX(4, 3, CALL)      \
    PAG_DISPATCH_F_LPX(4, 4, CALL)      \
    PAG_DISPATCH_F_LPX(1, 1, CALL)      \
    PAG_DISPATCH_F_LPX(1, 2, CALL)      \
    PAG_DISPATCH_F_LPX(1, 3, CALL)      \
    PAG_DISPATCH_F_LPX(1, 4, CALL)

int check_common(const char *name, const void *xyz, int64_t M, int n_levels, int n_feat) {
    PAG_CHECK_ARG(M >= 0, "%s: M < 0", name);
    PAG_CHECK_ARG(M == 0 || xyz != nullptr, "%s: xyz is NULL", name);
    PAG_CHECK_ARG(n_levels >= 1 && n_levels <= PAG_MAX_LEVELS, "%s: n_levels %d not in [1,%d]", name, n_levels, PAG_MAX_LEVELS);
    PAG_CHECK_ARG(n_feat == 1 || n_feat == 2 || n_feat == 4, "%s: n_feat %d not in {1,2,4}", name, n_feat);
    PAG_CHECK_ARG(n_levels * n_feat <= PAG_MAX_FEATS, "%s: n_levels*n_feat %d > %d", name, n_levels * n_feat, PAG_MAX_FEATS);
    return PAG_OK;
}

}  // namespace

static int hash_encode_fwd_impl(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                int n_feat, int log2_T, const float *resolutions_host, const float *feat_scale_host,
                                void *out, int out_dtype, int64_t out_stride_m, int64_t out_stride_c, int layout,
                                const void *addend_p, int flags, void *stream) {
    const bf16_t *addend = (const bf16_t *)addend_p;
    PAG_CHECK_ARG(!addend || layout == PAG_LAYOUT_XCD8, "pag_hash_encode_fwd_add: the addend needs the XCD8 layout");
    int rc = check_common("pag_hash_encode_fwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(log2_T >= 1 && log2_T <= 30, "pag_hash_encode_fwd: log2_T %d not in [1,30]", log2_T);
    PAG_CHECK_ARG(resolutions_host, "pag_hash_encode_fwd: resolutions_host is NULL");
    PAG_CHECK_ARG(table_dtype == PAG_F32 || table_dtype == PAG_F16, "pag_hash_encode_fwd: table dtype must be F32 or F16");
    PAG_CHECK_ARG(out_dtype == PAG_F32 || out_dtype == PAG_BF16, "pag_hash_encode_fwd: out dtype must be F32 or BF16");
    PAG_CHECK_ARG(layout == PAG_LAYOUT_STRIDED || (layout == PAG_LAYOUT_XCD8 && out_dtype == PAG_BF16 && ((n_levels + 7) / 8) * n_feat <= 8),
                  "pag_hash_encode_fwd: XCD8 layout needs bf16 output and ceil(L/8)*F <= 8");
    const int grouped = layout == PAG_LAYOUT_XCD8;
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(tables && out, "pag_hash_encode_fwd: NULL tables/out");
    HashParams p;
    p.L = n_levels;
    p.log2T = log2_T;
    p.has_scale = feat_scale_host != nullptr;
    p.pair_loads = (reinterpret_cast<uintptr_t>(tables) & 15) == 0 && log2_T >= 1;
    p.half_coords = (flags & PAG_ENC_HALF_COORDS) ? 1 : 0;
    for (int l = 0; l < n_levels; ++l) p.res[l] = resolutions_host[l];
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = feat_scale_host ? feat_scale_host[c] : 1.0f;
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(encode_grid(M)), block(PAG_ENC_FWD_THREADS);
    bool launched = false;
    if (table_dtype == PAG_F32 && out_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((hash_fwd_kernel<float, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, p, (float *)out, out_stride_m, out_stride_c, grouped, addend)))
    } else if (table_dtype == PAG_F32 && out_dtype == PAG_BF16) {
        PAG_DISPATCH_ALL((hash_fwd_kernel<float, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, p, (bf16_t *)out, out_stride_m, out_stride_c, grouped, addend)))
    } else if (table_dtype == PAG_F16 && out_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((hash_fwd_kernel<__half, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, p, (float *)out, out_stride_m, out_stride_c, grouped, addend)))
    } else {
        PAG_DISPATCH_ALL((hash_fwd_kernel<__half, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, p, (bf16_t *)out, out_stride_m, out_stride_c, grouped, addend)))
    }
    PAG_CHECK_ARG(launched, "pag_hash_encode_fwd: unsupported (n_feat=%d, n_levels=%d)", n_feat, n_levels);
    PAG_CHECK_LAUNCH("pag_hash_encode_fwd");
    return PAG_OK;
}

extern "C" int pag_hash_encode_fwd(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                   int n_feat, int log2_T, const float *resolutions_host, const float *feat_scale_host,
                                   void *out, int out_dtype, int64_t out_stride_m, int64_t out_stride_c, int layout, int flags, void *stream) {
    return hash_encode_fwd_impl(xyz, M, tables, table_dtype, n_levels, n_feat, log2_T, resolutions_host, feat_scale_host, out, out_dtype,
                                out_stride_m, out_stride_c, layout, nullptr, flags, stream);
}

extern "C" int pag_hash_encode_fwd_add(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                       int n_feat, int log2_T, const float *resolutions_host, const float *feat_scale_host,
                                       const void *addend, void *out, int flags, void *stream) {
    return hash_encode_fwd_impl(xyz, M, tables, table_dtype, n_levels, n_feat, log2_T, resolutions_host, feat_scale_host, out, PAG_BF16, 0,
                                0, PAG_LAYOUT_XCD8, addend, flags, stream);
}

static int hash_encode_bwd_impl(bool overwrite, const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                int64_t g_stride_c, int layout, int n_levels, int n_feat, int log2_T, const float *resolutions_host,
                                const float *feat_scale_host, float *grad_tables, void *workspace,
                                int64_t workspace_bytes, int flags, void *stream) {
    int rc = check_common("pag_hash_encode_bwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(!overwrite || workspace, "pag_hash_encode_bwd_set: the overwriting form needs the binned algorithm (a workspace)");
    PAG_CHECK_ARG(log2_T >= 1 && log2_T <= 30, "pag_hash_encode_bwd: log2_T %d not in [1,30]", log2_T);
    PAG_CHECK_ARG(resolutions_host, "pag_hash_encode_bwd: resolutions_host is NULL");
    PAG_CHECK_ARG(grad_dtype == PAG_F32 || grad_dtype == PAG_BF16, "pag_hash_encode_bwd: grad dtype must be F32 or BF16");
    const int grouped = layout == PAG_LAYOUT_XCD8;
    PAG_CHECK_ARG(!grouped || (workspace && grad_dtype == PAG_BF16 && ((n_levels + 7) / 8) * n_feat <= 8),
                  "pag_hash_encode_bwd: XCD8 layout needs bf16 gradients, a workspace and ceil(L/8)*F <= 8");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(grad_out && grad_tables, "pag_hash_encode_bwd: NULL grad_out/grad_tables");
    HashParams p;
    p.L = n_levels;
    p.log2T = log2_T;
    p.has_scale = feat_scale_host != nullptr;
    p.half_coords = (flags & PAG_ENC_HALF_COORDS) ? 1 : 0;
    for (int l = 0; l < n_levels; ++l) p.res[l] = resolutions_host[l];
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = feat_scale_host ? feat_scale_host[c] : 1.0f;
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    if (workspace) {
        PermutoParams unused{};
        int r2 = launch_binned<0>(xyz, M, grad_out, grad_dtype, g_stride_m, g_stride_c, grouped, n_levels, n_feat, (int64_t)1 << log2_T, p,
                                  unused, grad_tables, workspace, workspace_bytes, st, "pag_hash_encode_bwd", overwrite);
        if (r2) return r2;
        PAG_CHECK_LAUNCH("pag_hash_encode_bwd");
        return PAG_OK;
    }
    dim3 grid(encode_grid(M)), block(PAG_ENC_FWD_THREADS);
    bool launched = false;
    if (grad_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((hash_bwd_kernel<float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)grad_out, g_stride_m, g_stride_c, p, grad_tables)))
    } else {
        PAG_DISPATCH_ALL((hash_bwd_kernel<bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const bf16_t *)grad_out, g_stride_m, g_stride_c, p, grad_tables)))
    }
    PAG_CHECK_ARG(launched, "pag_hash_encode_bwd: unsupported (n_feat=%d, n_levels=%d)", n_feat, n_levels);
    PAG_CHECK_LAUNCH("pag_hash_encode_bwd");
    return PAG_OK;
}

static int fill_permuto(PermutoParams &p, int n_levels, int n_feat, uint32_t capacity, const float *sf, const float *sh,
                        const float *scale, int flags) {
    p.L = n_levels;
    p.half_coords = (flags & PAG_ENC_HALF_COORDS) ? 1 : 0;
    p.capacity = capacity;
    p.pow2mask = (capacity & (capacity - 1)) == 0 ? capacity - 1 : 0;
    if (capacity == 1) p.pow2mask = 0;
    p.has_scale = scale != nullptr;
    for (int l = 0; l < n_levels; ++l)
        for (int a = 0; a < 3; ++a) {
            p.sf[l][a] = sf[l * 3 + a];
            p.shift[l][a] = sh[l * 3 + a];
        }
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = scale ? scale[c] : 1.0f;
    return 0;
}

static int permuto_encode_fwd_impl(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                   int n_feat, uint32_t capacity, const float *scale_factor_host, const float *shift_host,
                                   const float *feat_scale_host, void *out, int out_dtype, int64_t out_stride_m,
                                   int64_t out_stride_c, int layout, const void *addend_p, int flags, void *stream) {
    const bf16_t *addend = (const bf16_t *)addend_p;
    PAG_CHECK_ARG(!addend || layout == PAG_LAYOUT_XCD8, "pag_permuto_encode_fwd_add: the addend needs the XCD8 layout");
    int rc = check_common("pag_permuto_encode_fwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(capacity >= 1, "pag_permuto_encode_fwd: capacity is 0");
    PAG_CHECK_ARG(scale_factor_host && shift_host, "pag_permuto_encode_fwd: NULL scale_factor/shift");
    PAG_CHECK_ARG(table_dtype == PAG_F32 || table_dtype == PAG_F16, "pag_permuto_encode_fwd: table dtype must be F32 or F16");
    PAG_CHECK_ARG(out_dtype == PAG_F32 || out_dtype == PAG_BF16, "pag_permuto_encode_fwd: out dtype must be F32 or BF16");
    PAG_CHECK_ARG(layout == PAG_LAYOUT_STRIDED || (layout == PAG_LAYOUT_XCD8 && out_dtype == PAG_BF16 && ((n_levels + 7) / 8) * n_feat <= 8),
                  "pag_permuto_encode_fwd: XCD8 layout needs bf16 output and ceil(L/8)*F <= 8");
    const int grouped = layout == PAG_LAYOUT_XCD8;
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(tables && out, "pag_permuto_encode_fwd: NULL tables/out");
    PermutoParams p;
    fill_permuto(p, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host, flags);
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(encode_grid(M)), block(PAG_ENC_FWD_THREADS);
    bool launched = false;
#define PFWD(TT, OT)                                                                                                                 \
    PAG_DISPATCH_ALL((permuto_fwd_kernel<TT, OT, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const TT *)tables, p, (OT *)out, out_stride_m, \
                                                                                   out_stride_c, grouped)))
#define PFWD_ADD(TT, OT)                                                                                                             \
    PAG_DISPATCH_ALL((permuto_fwd_add_kernel<TT, OT, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const TT *)tables, p, (OT *)out, out_stride_m, \
                                                                                       out_stride_c, grouped, addend)))
    if (table_dtype == PAG_F32 && out_dtype == PAG_F32) {
        PFWD(float, float)
    } else if (table_dtype == PAG_F32 && out_dtype == PAG_BF16) {
        if (addend) { PFWD_ADD(float, bf16_t) } else { PFWD(float, bf16_t) }
    } else if (table_dtype == PAG_F16 && out_dtype == PAG_F32) {
        PFWD(__half, float)
    } else {
        if (addend) { PFWD_ADD(__half, bf16_t) } else { PFWD(__half, bf16_t) }
    }
#undef PFWD
#undef PFWD_ADD
    PAG_CHECK_ARG(launched, "pag_permuto_encode_fwd: unsupported (n_feat=%d, n_levels=%d)", n_feat, n_levels);
    PAG_CHECK_LAUNCH("pag_permuto_encode_fwd");
    return PAG_OK;
}

extern "C" int pag_permuto_encode_fwd(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                      int n_feat, uint32_t capacity, const float *scale_factor_host, const float *shift_host,
                                      const float *feat_scale_host, void *out, int out_dtype, int64_t out_stride_m,
                                      int64_t out_stride_c, int layout, int flags, void *stream) {
    return permuto_encode_fwd_impl(xyz, M, tables, table_dtype, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host,
                                   out, out_dtype, out_stride_m, out_stride_c, layout, nullptr, flags, stream);
}

extern "C" int pag_permuto_encode_fwd_add(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                          int n_feat, uint32_t capacity, const float *scale_factor_host, const float *shift_host,
                                          const float *feat_scale_host, const void *addend, void *out, int flags, void *stream) {
    return permuto_encode_fwd_impl(xyz, M, tables, table_dtype, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host,
                                   out, PAG_BF16, 0, 0, PAG_LAYOUT_XCD8, addend, flags, stream);
}

static int permuto_encode_bwd_impl(bool overwrite, const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                   int64_t g_stride_c, int layout, int n_levels, int n_feat, uint32_t capacity,
                                   const float *scale_factor_host, const float *shift_host, const float *feat_scale_host,
                                   float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    int rc = check_common("pag_permuto_encode_bwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(!overwrite || workspace, "pag_permuto_encode_bwd_set: the overwriting form needs the binned algorithm (a workspace)");
    PAG_CHECK_ARG(capacity >= 1, "pag_permuto_encode_bwd: capacity is 0");
    PAG_CHECK_ARG(scale_factor_host && shift_host, "pag_permuto_encode_bwd: NULL scale_factor/shift");
    PAG_CHECK_ARG(grad_dtype == PAG_F32 || grad_dtype == PAG_BF16, "pag_permuto_encode_bwd: grad dtype must be F32 or BF16");
    const int grouped = layout == PAG_LAYOUT_XCD8;
    PAG_CHECK_ARG(!grouped || (workspace && grad_dtype == PAG_BF16 && ((n_levels + 7) / 8) * n_feat <= 8),
                  "pag_permuto_encode_bwd: XCD8 layout needs bf16 gradients, a workspace and ceil(L/8)*F <= 8");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(grad_out && grad_tables, "pag_permuto_encode_bwd: NULL grad_out/grad_tables");
    PermutoParams p;
    fill_permuto(p, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host, flags);
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    if (workspace) {
        HashParams unused{};
        int r2 = launch_binned<1>(xyz, M, grad_out, grad_dtype, g_stride_m, g_stride_c, grouped, n_levels, n_feat, (int64_t)capacity, unused,
                                  p, grad_tables, workspace, workspace_bytes, st, "pag_permuto_encode_bwd", overwrite);
        if (r2) return r2;
        PAG_CHECK_LAUNCH("pag_permuto_encode_bwd");
        return PAG_OK;
    }
    dim3 grid(encode_grid(M)), block(PAG_ENC_FWD_THREADS);
    bool launched = false;
    if (grad_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((permuto_bwd_kernel<float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)grad_out, g_stride_m, g_stride_c, p, grad_tables)))
    } else {
        PAG_DISPATCH_ALL((permuto_bwd_kernel<bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const bf16_t *)grad_out, g_stride_m, g_stride_c, p, grad_tables)))
    }
    PAG_CHECK_ARG(launched, "pag_permuto_encode_bwd: unsupported (n_feat=%d, n_levels=%d)", n_feat, n_levels);
    PAG_CHECK_LAUNCH("pag_permuto_encode_bwd");
    return PAG_OK;
}

struct RaysArgs {            // per-ray reduction of the position gradient (launch_xyz_grad): NULL ridx = the per-sample form
    const int32_t *ridx = nullptr;
    const float *depths = nullptr;
    const int64_t *pack_start = nullptr;
    int64_t N = 0;
};
static int64_t rays_slot_rows(int64_t M, int64_t N) { return (M + 63) / 64 + N; }

template <int KIND>
static int launch_xyz_grad(const char *name, const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out,
                           int grad_dtype, int64_t sm, int64_t sc, int layout, int n_levels, int n_feat, const HashParams &hp,
                           const PermutoParams &pp, float *d_xyz, void *workspace, int64_t workspace_bytes, hipStream_t st, const RaysArgs &ra = RaysArgs()) {
    PAG_CHECK_ARG(table_dtype == PAG_F32 || table_dtype == PAG_F16, "%s: table dtype must be F32 or F16", name);
    PAG_CHECK_ARG(grad_dtype == PAG_F32 || grad_dtype == PAG_BF16, "%s: grad dtype must be F32 or BF16", name);
    const int grouped = layout == PAG_LAYOUT_XCD8;
    PAG_CHECK_ARG(!grouped || (grad_dtype == PAG_BF16 && ((n_levels + 7) / 8) * n_feat <= 8),
                  "%s: XCD8 layout needs bf16 gradients and ceil(L/8)*F <= 8", name);
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(tables && grad_out && d_xyz && workspace, "%s: NULL tables/grad_out/d_xyz/workspace", name);
    const int groups = n_levels < 8 ? n_levels : 8;
    const int lpx = (n_levels + 7) / 8;
    dim3 grid(encode_grid(M)), block(PAG_ENC_FWD_THREADS);
    bool launched = false;
    if (ra.ridx) {       // reduced per ray inside the pass: d_xyz = out f32 [N,6] (d origin | d dir)
        PAG_CHECK_ARG(ra.depths && ra.pack_start && ra.N >= 1, "%s: per-ray form needs ridx, depths, pack_start and N >= 1", name);
        const int64_t rows_ = rays_slot_rows(M, ra.N);
        PAG_CHECK_ARG(workspace_bytes >= 8 * rows_ * 6 * (int64_t)sizeof(float), "%s: workspace smaller than 8 * (ceil(M / 64) + N) * 6 floats", name);
        float *slots = (float *)workspace;
#define PAG_XR(TT, GT) PAG_DISPATCH_ALL((xyz_grad_rays_kernel<KIND, TT, GT, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const TT *)tables, (const GT *)grad_out, sm, sc, grouped, hp, pp, ra.ridx, ra.depths, slots, rows_)))
        if (table_dtype == PAG_F32 && grad_dtype == PAG_F32) { PAG_XR(float, float) }
        else if (table_dtype == PAG_F32 && grad_dtype == PAG_BF16) { PAG_XR(float, bf16_t) }
        else if (table_dtype == PAG_F16 && grad_dtype == PAG_F32) { PAG_XR(__half, float) }
        else { PAG_XR(__half, bf16_t) }
#undef PAG_XR
        PAG_CHECK_ARG(launched, "%s: unsupported (n_feat=%d, n_levels=%d)", name, n_feat, n_levels);
        ray_slots_sum_kernel<<<dim3((unsigned)((ra.N + 3) / 4)), dim3(256), 0, st>>>(ra.pack_start, ra.N, slots, rows_, groups, d_xyz);
        PAG_CHECK_LAUNCH(name);
        return PAG_OK;
    }
    PAG_CHECK_ARG(workspace_bytes >= (int64_t)8 * M * 3 * (int64_t)sizeof(float), "%s: workspace smaller than 8*M*3 floats", name);
    float *part = (float *)workspace;
    if (table_dtype == PAG_F32 && grad_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((xyz_grad_kernel<KIND, float, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, (const float *)grad_out, sm, sc, grouped, hp, pp, part)))
    } else if (table_dtype == PAG_F32 && grad_dtype == PAG_BF16) {
        PAG_DISPATCH_ALL((xyz_grad_kernel<KIND, float, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, (const bf16_t *)grad_out, sm, sc, grouped, hp, pp, part)))
    } else if (table_dtype == PAG_F16 && grad_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((xyz_grad_kernel<KIND, __half, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, (const float *)grad_out, sm, sc, grouped, hp, pp, part)))
    } else {
        PAG_DISPATCH_ALL((xyz_grad_kernel<KIND, __half, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, (const bf16_t *)grad_out, sm, sc, grouped, hp, pp, part)))
    }
    PAG_CHECK_ARG(launched, "%s: unsupported (n_feat=%d, n_levels=%d)", name, n_feat, n_levels);
    const int64_t n = M * 3;
    xyz_grad_sum_kernel<<<dim3((unsigned)((((n % 4 == 0) ? n / 4 : n) + 255) / 256)), dim3(256), 0, st>>>(part, n, groups, d_xyz);
    PAG_CHECK_LAUNCH(name);
    return PAG_OK;
}

extern "C" int pag_hash_encode_bwd_xyz(const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out,
                                       int grad_dtype, int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat,
                                       int log2_T, const float *resolutions_host, const float *feat_scale_host, float *d_xyz,
                                       void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    int rc = check_common("pag_hash_encode_bwd_xyz", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(log2_T >= 1 && log2_T <= 30, "pag_hash_encode_bwd_xyz: log2_T %d not in [1,30]", log2_T);
    PAG_CHECK_ARG(resolutions_host, "pag_hash_encode_bwd_xyz: resolutions_host is NULL");
    HashParams p;
    p.L = n_levels;
    p.log2T = log2_T;
    p.has_scale = feat_scale_host != nullptr;
    p.half_coords = (flags & PAG_ENC_HALF_COORDS) ? 1 : 0;
    for (int l = 0; l < n_levels; ++l) p.res[l] = resolutions_host[l];
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = feat_scale_host ? feat_scale_host[c] : 1.0f;
    PermutoParams unused{};
    return launch_xyz_grad<0>("pag_hash_encode_bwd_xyz", xyz, M, tables, table_dtype, grad_out, grad_dtype, g_stride_m, g_stride_c, layout,
                              n_levels, n_feat, p, unused, d_xyz, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int pag_permuto_encode_bwd_xyz(const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out,
                                          int grad_dtype, int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels,
                                          int n_feat, uint32_t capacity, const float *scale_factor_host, const float *shift_host,
                                          const float *feat_scale_host, float *d_xyz, void *workspace, int64_t workspace_bytes,
                                          int flags, void *stream) {
    int rc = check_common("pag_permuto_encode_bwd_xyz", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(capacity >= 1, "pag_permuto_encode_bwd_xyz: capacity is 0");
    PAG_CHECK_ARG(scale_factor_host && shift_host, "pag_permuto_encode_bwd_xyz: NULL scale_factor/shift");
    PermutoParams p;
    fill_permuto(p, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host, flags);
    HashParams unused{};
    return launch_xyz_grad<1>("pag_permuto_encode_bwd_xyz", xyz, M, tables, table_dtype, grad_out, grad_dtype, g_stride_m, g_stride_c,
                              layout, n_levels, n_feat, unused, p, d_xyz, workspace, workspace_bytes, (hipStream_t)stream);
}

// Position gradient reduced per ray (d origin | d dir, f32 [N,6]) in the gather pass itself: see xyz_grad_rays_kernel.
extern "C" int64_t pag_encode_bwd_rays_workspace_bytes(int64_t M, int64_t N) {
    return (M > 0 && N > 0) ? 8 * rays_slot_rows(M, N) * 6 * (int64_t)sizeof(float) : 0;
}
static int rays_check(const char *name, const int32_t *ridx, const float *depths, const int64_t *pack_start, int64_t N, int64_t M) {
    PAG_CHECK_ARG(N >= 1, "%s: N %lld < 1", name, (long long)N);
    PAG_CHECK_ARG(M == 0 || (ridx && depths && pack_start), "%s: NULL ridx / depths / pack_start", name);
    return PAG_OK;
}
extern "C" int pag_hash_encode_bwd_rays(const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out, int grad_dtype,
                                        int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat, int log2_T,
                                        const float *resolutions_host, const float *feat_scale_host, const int32_t *ridx, const float *depths,
                                        const int64_t *pack_start, int64_t N, float *out, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    int rc = check_common("pag_hash_encode_bwd_rays", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    rc = rays_check("pag_hash_encode_bwd_rays", ridx, depths, pack_start, N, M);
    if (rc) return rc;
    PAG_CHECK_ARG(log2_T >= 1 && log2_T <= 30, "pag_hash_encode_bwd_rays: log2_T %d not in [1,30]", log2_T);
    PAG_CHECK_ARG(resolutions_host, "pag_hash_encode_bwd_rays: resolutions_host is NULL");
    HashParams p;
    p.L = n_levels;
    p.log2T = log2_T;
    p.has_scale = feat_scale_host != nullptr;
    p.half_coords = (flags & PAG_ENC_HALF_COORDS) ? 1 : 0;
    for (int l = 0; l < n_levels; ++l) p.res[l] = resolutions_host[l];
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = feat_scale_host ? feat_scale_host[c] : 1.0f;
    PermutoParams unused{};
    RaysArgs ra{ridx, depths, pack_start, N};
    if (M == 0) return hipMemsetAsync(out, 0, (size_t)N * 6 * sizeof(float), (hipStream_t)stream) == hipSuccess ? PAG_OK : PAG_ERR_LAUNCH;
    return launch_xyz_grad<0>("pag_hash_encode_bwd_rays", xyz, M, tables, table_dtype, grad_out, grad_dtype, g_stride_m, g_stride_c, layout, n_levels, n_feat,
                              p, unused, out, workspace, workspace_bytes, (hipStream_t)stream, ra);
}
extern "C" int pag_permuto_encode_bwd_rays(const float *xyz, int64_t M, const void *tables, int table_dtype, const void *grad_out, int grad_dtype,
                                           int64_t g_stride_m, int64_t g_stride_c, int layout, int n_levels, int n_feat, uint32_t capacity,
                                           const float *scale_factor_host, const float *shift_host, const float *feat_scale_host, const int32_t *ridx,
                                           const float *depths, const int64_t *pack_start, int64_t N, float *out, void *workspace, int64_t workspace_bytes,
                                           int flags, void *stream) {
    int rc = check_common("pag_permuto_encode_bwd_rays", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    rc = rays_check("pag_permuto_encode_bwd_rays", ridx, depths, pack_start, N, M);
    if (rc) return rc;
    PAG_CHECK_ARG(capacity >= 1 && scale_factor_host && shift_host, "pag_permuto_encode_bwd_rays: capacity 0 or NULL scale_factor / shift");
    PermutoParams p;
    fill_permuto(p, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host, flags);
    HashParams unused{};
    RaysArgs ra{ridx, depths, pack_start, N};
    if (M == 0) return hipMemsetAsync(out, 0, (size_t)N * 6 * sizeof(float), (hipStream_t)stream) == hipSuccess ? PAG_OK : PAG_ERR_LAUNCH;
    return launch_xyz_grad<1>("pag_permuto_encode_bwd_rays", xyz, M, tables, table_dtype, grad_out, grad_dtype, g_stride_m, g_stride_c, layout, n_levels, n_feat,
                              unused, p, out, workspace, workspace_bytes, (hipStream_t)stream, ra);
}

extern "C" int64_t pag_encode_bwd_workspace_bytes(int64_t M, int n_levels, int n_feat, int n_vertices, int64_t rows_per_level) {
    if (M <= 0 || n_levels <= 0 || (n_vertices != 4 && n_vertices != 8) || rows_per_level <= 0) return 0;
    return bin_plan(M, n_levels, n_feat, n_vertices, rows_per_level).total;
}

// Exported forms: *_bwd ACCUMULATES into grad_tables (caller zeroes it); *_bwd_set OVERWRITES every row (binned algorithm only) -
// no zero fill of the 50 MB table before the call and no read of it in the reduce pass.
extern "C" int pag_hash_encode_bwd(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                   int64_t g_stride_c, int layout, int n_levels, int n_feat, int log2_T, const float *resolutions_host,
                                   const float *feat_scale_host, float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    return hash_encode_bwd_impl(false, xyz, M, grad_out, grad_dtype, g_stride_m, g_stride_c, layout, n_levels, n_feat, log2_T, resolutions_host,
                                feat_scale_host, grad_tables, workspace, workspace_bytes, flags, stream);
}
extern "C" int pag_hash_encode_bwd_set(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                       int64_t g_stride_c, int layout, int n_levels, int n_feat, int log2_T, const float *resolutions_host,
                                       const float *feat_scale_host, float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    return hash_encode_bwd_impl(true, xyz, M, grad_out, grad_dtype, g_stride_m, g_stride_c, layout, n_levels, n_feat, log2_T, resolutions_host,
                                feat_scale_host, grad_tables, workspace, workspace_bytes, flags, stream);
}
extern "C" int pag_permuto_encode_bwd(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                      int64_t g_stride_c, int layout, int n_levels, int n_feat, uint32_t capacity,
                                      const float *scale_factor_host, const float *shift_host, const float *feat_scale_host,
                                      float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    return permuto_encode_bwd_impl(false, xyz, M, grad_out, grad_dtype, g_stride_m, g_stride_c, layout, n_levels, n_feat, capacity, scale_factor_host,
                                   shift_host, feat_scale_host, grad_tables, workspace, workspace_bytes, flags, stream);
}
extern "C" int pag_permuto_encode_bwd_set(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                          int64_t g_stride_c, int layout, int n_levels, int n_feat, uint32_t capacity,
                                          const float *scale_factor_host, const float *shift_host, const float *feat_scale_host,
                                          float *grad_tables, void *workspace, int64_t workspace_bytes, int flags, void *stream) {
    return permuto_encode_bwd_impl(true, xyz, M, grad_out, grad_dtype, g_stride_m, g_stride_c, layout, n_levels, n_feat, capacity, scale_factor_host,
                                   shift_host, feat_scale_host, grad_tables, workspace, workspace_bytes, flags, stream);
}

PAG_BLOCK_TIMING_EXPORT(encode)

#ifdef PAG_EXP_JAC
extern "C" int pag_debug_set_jac(void *ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_jac), &ptr, sizeof(ptr)); }
#endif
