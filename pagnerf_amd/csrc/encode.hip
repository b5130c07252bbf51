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
#include "common.h"

namespace {

struct HashParams {
    float res[PAG_MAX_LEVELS];
    float scale[PAG_MAX_FEATS];
    int L, log2T, has_scale;
};

struct PermutoParams {
    float sf[PAG_MAX_LEVELS][3];
    float shift[PAG_MAX_LEVELS][3];
    float scale[PAG_MAX_FEATS];
    int L, has_scale;
    uint32_t capacity, pow2mask;   // pow2mask = capacity-1 when capacity is a power of two, else 0
};

template <typename T, int F> struct Vec;
template <> struct Vec<float, 1> { typedef float type; };
template <> struct Vec<float, 2> { typedef float2 type; };
template <> struct Vec<float, 4> { typedef float4 type; };
template <> struct Vec<__half, 1> { typedef __half type; };
template <> struct Vec<__half, 2> { typedef __half2 type; };
template <> struct Vec<__half, 4> { typedef uint2 type; };

template <int F> __device__ __forceinline__ void gather(const float *row, float (&v)[F]) {
    typename Vec<float, F>::type t = *reinterpret_cast<const typename Vec<float, F>::type *>(row);
    const float *p = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int f = 0; f < F; ++f) v[f] = p[f];
}
template <int F> __device__ __forceinline__ void gather(const __half *row, float (&v)[F]) {
    typename Vec<__half, F>::type t = *reinterpret_cast<const typename Vec<__half, F>::type *>(row);
    const __half *p = reinterpret_cast<const __half *>(&t);
#pragma unroll
    for (int f = 0; f < F; ++f) v[f] = __half2float(p[f]);
}

// ------------------------------------------------------------------------------------ hash grid
// grids/hash_grid_torch.py:26-46 (cell lookup) and :69-77 (weights), one level.
__device__ __forceinline__ void hash_cell(const float (&x)[3], float res, int log2T, uint32_t (&idx)[8], float (&w)[3]) {
    const float cell = __fdiv_rn(2.0f, res);
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float xc = fminf(fmaxf(x[a], -1.0f), 1.0f);
        float t = __fdiv_rn(__fadd_rn(xc, 1.0f), cell);
        int bl = (int)floorf(t);
        float vmin = __fadd_rn(__fmul_rn((float)bl, cell), -1.0f);
        float vmax = __fadd_rn(vmin, cell);
        w[a] = __fdiv_rn(__fsub_rn(x[a], vmin), __fsub_rn(vmax, vmin));
        c[a] = (uint32_t)bl;
    }
    const uint32_t mask = (1u << log2T) - 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // corner k = 4i + 2j + kk (hash_grid_torch.py:10)
        uint32_t cx = c[0] + ((k >> 2) & 1), cy = c[1] + ((k >> 1) & 1), cz = c[2] + (k & 1);
        idx[k] = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & mask;
    }
}

__device__ __forceinline__ float lerp_ref(float a, float b, float w, float omw) {
    return __fadd_rn(__fmul_rn(a, omw), __fmul_rn(b, w));
}

template <typename TableT, typename OutT, int F, int LPX>
__global__ __launch_bounds__(256) void hash_fwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                       const TableT *__restrict__ tables, HashParams p,
                                                       OutT *__restrict__ out, int64_t sm, int64_t sc) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= M) return;
    float x[3] = {xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    const int64_t T = (int64_t)1 << p.log2T;
    float e[LPX][8][F];
    float w[LPX][3];
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = g + 8 * j;
        int le = l < p.L ? l : p.L - 1;
        uint32_t idx[8];
        hash_cell(x, p.res[le], p.log2T, idx, w[j]);
        const TableT *tab = tables + (int64_t)le * T * F;
#pragma unroll
        for (int k = 0; k < 8; ++k) gather<F>(tab + (int64_t)idx[k] * F, e[j][k]);
    }
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = g + 8 * j;
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
            pag_st(out + i * sm + (int64_t)(l * F + f) * sc, v);
        }
    }
}

template <typename GradT, int F, int LPX>
__global__ __launch_bounds__(256) void hash_bwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                       const GradT *__restrict__ go, int64_t sm, int64_t sc,
                                                       HashParams p, float *__restrict__ gtab) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= M) return;
    float x[3] = {xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    const int64_t T = (int64_t)1 << p.log2T;
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = g + 8 * j;
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

// --------------------------------------------------------------------------- permutohedral lattice
// oracle/permuto_encode.py lattice_simplex() + vertex_indices(), one level, d = 3.
__device__ __forceinline__ void permuto_simplex(const float (&x)[3], const float (&sh)[3], const float (&sf)[3],
                                                uint32_t capacity, uint32_t pow2mask, uint32_t (&idx)[4], float (&bary)[4]) {
    float cf[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) cf[a] = __fmul_rn(__fadd_rn(x[a], sh[a]), sf[a]);
    float E[4];
    float s = 0.0f;
    E[3] = __fsub_rn(s, __fmul_rn(3.0f, cf[2]));
    s = __fadd_rn(s, cf[2]);
    E[2] = __fsub_rn(s, __fmul_rn(2.0f, cf[1]));
    s = __fadd_rn(s, cf[1]);
    E[1] = __fsub_rn(s, cf[0]);
    s = __fadd_rn(s, cf[0]);
    E[0] = s;

    int rem0[4], rank[4] = {0, 0, 0, 0};
    float resid[4];
    int sum = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float v = E[a] * 0.25f;
        float up = ceilf(v) * 4.0f, dn = floorf(v) * 4.0f;
        float r = (__fsub_rn(up, E[a]) < __fsub_rn(E[a], dn)) ? up : dn;
        rem0[a] = (int)r;
        sum += rem0[a];
        resid[a] = __fsub_rn(E[a], r);
    }
    sum >>= 2;   // exact: every rem0 is a multiple of 4
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            int lt = resid[a] < resid[b];
            rank[a] += lt;
            rank[b] += 1 - lt;
        }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        rank[a] += sum;
        if (rank[a] < 0) {
            rank[a] += 4;
            rem0[a] += 4;
        } else if (rank[a] > 3) {
            rank[a] -= 4;
            rem0[a] -= 4;
        }
    }
    float b5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float delta = __fsub_rn(E[a], (float)rem0[a]) * 0.25f;
        int slot = 3 - rank[a];
#pragma unroll
        for (int k = 0; k < 5; ++k) {   // predicated: no runtime-indexed private array
            b5[k] = (k == slot) ? __fadd_rn(b5[k], delta) : b5[k];
            b5[k] = (k == slot + 1) ? __fsub_rn(b5[k], delta) : b5[k];
        }
    }
    b5[0] = __fadd_rn(b5[0], __fadd_rn(1.0f, b5[4]));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bary[r] = b5[r];
        uint32_t k = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int key = rem0[a] + r - ((rank[a] > 3 - r) ? 4 : 0);
            k = (k + (uint32_t)key) * 2531011u;
        }
        idx[r] = pow2mask ? (k & pow2mask) : (k % capacity);
    }
}

template <typename TableT, typename OutT, int F, int LPX>
__global__ __launch_bounds__(256) void permuto_fwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                          const TableT *__restrict__ tables, PermutoParams p,
                                                          OutT *__restrict__ out, int64_t sm, int64_t sc) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= M) return;
    float x[3] = {xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    float e[LPX][4][F];
    float bary[LPX][4];
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = g + 8 * j;
        int le = l < p.L ? l : p.L - 1;
        uint32_t idx[4];
        permuto_simplex(x, p.shift[le], p.sf[le], p.capacity, p.pow2mask, idx, bary[j]);
        const TableT *tab = tables + (int64_t)le * p.capacity * F;
#pragma unroll
        for (int r = 0; r < 4; ++r) gather<F>(tab + (int64_t)idx[r] * F, e[j][r]);
    }
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = g + 8 * j;
        if (l >= p.L) break;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __fadd_rn(acc, __fmul_rn(e[j][r][f], bary[j][r]));
            if (p.has_scale) acc = __fmul_rn(acc, p.scale[l * F + f]);
            pag_st(out + i * sm + (int64_t)(l * F + f) * sc, acc);
        }
    }
}

template <typename GradT, int F, int LPX>
__global__ __launch_bounds__(256) void permuto_bwd_kernel(const float *__restrict__ xyz, int64_t M,
                                                          const GradT *__restrict__ go, int64_t sm, int64_t sc,
                                                          PermutoParams p, float *__restrict__ gtab) {
    const int g = blockIdx.x & 7;
    const int64_t i = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (i >= M) return;
    float x[3] = {xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2]};
#pragma unroll
    for (int j = 0; j < LPX; ++j) {
        int l = g + 8 * j;
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

inline unsigned encode_grid(int64_t M) { return (unsigned)(((M + 255) / 256) * 8); }

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

extern "C" int pag_hash_encode_fwd(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                   int n_feat, int log2_T, const float *resolutions_host, const float *feat_scale_host,
                                   void *out, int out_dtype, int64_t out_stride_m, int64_t out_stride_c, void *stream) {
    int rc = check_common("pag_hash_encode_fwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(log2_T >= 1 && log2_T <= 30, "pag_hash_encode_fwd: log2_T %d not in [1,30]", log2_T);
    PAG_CHECK_ARG(resolutions_host, "pag_hash_encode_fwd: resolutions_host is NULL");
    PAG_CHECK_ARG(table_dtype == PAG_F32 || table_dtype == PAG_F16, "pag_hash_encode_fwd: table dtype must be F32 or F16");
    PAG_CHECK_ARG(out_dtype == PAG_F32 || out_dtype == PAG_BF16, "pag_hash_encode_fwd: out dtype must be F32 or BF16");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(tables && out, "pag_hash_encode_fwd: NULL tables/out");
    HashParams p;
    p.L = n_levels;
    p.log2T = log2_T;
    p.has_scale = feat_scale_host != nullptr;
    for (int l = 0; l < n_levels; ++l) p.res[l] = resolutions_host[l];
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = feat_scale_host ? feat_scale_host[c] : 1.0f;
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(encode_grid(M)), block(256);
    bool launched = false;
    if (table_dtype == PAG_F32 && out_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((hash_fwd_kernel<float, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, p, (float *)out, out_stride_m, out_stride_c)))
    } else if (table_dtype == PAG_F32 && out_dtype == PAG_BF16) {
        PAG_DISPATCH_ALL((hash_fwd_kernel<float, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, p, (bf16_t *)out, out_stride_m, out_stride_c)))
    } else if (table_dtype == PAG_F16 && out_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((hash_fwd_kernel<__half, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, p, (float *)out, out_stride_m, out_stride_c)))
    } else {
        PAG_DISPATCH_ALL((hash_fwd_kernel<__half, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, p, (bf16_t *)out, out_stride_m, out_stride_c)))
    }
    PAG_CHECK_ARG(launched, "pag_hash_encode_fwd: unsupported (n_feat=%d, n_levels=%d)", n_feat, n_levels);
    PAG_CHECK_LAUNCH("pag_hash_encode_fwd");
    return PAG_OK;
}

extern "C" int pag_hash_encode_bwd(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                   int64_t g_stride_c, int n_levels, int n_feat, int log2_T, const float *resolutions_host,
                                   const float *feat_scale_host, float *grad_tables, void *stream) {
    int rc = check_common("pag_hash_encode_bwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(log2_T >= 1 && log2_T <= 30, "pag_hash_encode_bwd: log2_T %d not in [1,30]", log2_T);
    PAG_CHECK_ARG(resolutions_host, "pag_hash_encode_bwd: resolutions_host is NULL");
    PAG_CHECK_ARG(grad_dtype == PAG_F32 || grad_dtype == PAG_BF16, "pag_hash_encode_bwd: grad dtype must be F32 or BF16");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(grad_out && grad_tables, "pag_hash_encode_bwd: NULL grad_out/grad_tables");
    HashParams p;
    p.L = n_levels;
    p.log2T = log2_T;
    p.has_scale = feat_scale_host != nullptr;
    for (int l = 0; l < n_levels; ++l) p.res[l] = resolutions_host[l];
    for (int c = 0; c < n_levels * n_feat; ++c) p.scale[c] = feat_scale_host ? feat_scale_host[c] : 1.0f;
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(encode_grid(M)), block(256);
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
                        const float *scale) {
    p.L = n_levels;
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

extern "C" int pag_permuto_encode_fwd(const float *xyz, int64_t M, const void *tables, int table_dtype, int n_levels,
                                      int n_feat, uint32_t capacity, const float *scale_factor_host, const float *shift_host,
                                      const float *feat_scale_host, void *out, int out_dtype, int64_t out_stride_m,
                                      int64_t out_stride_c, void *stream) {
    int rc = check_common("pag_permuto_encode_fwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(capacity >= 1, "pag_permuto_encode_fwd: capacity is 0");
    PAG_CHECK_ARG(scale_factor_host && shift_host, "pag_permuto_encode_fwd: NULL scale_factor/shift");
    PAG_CHECK_ARG(table_dtype == PAG_F32 || table_dtype == PAG_F16, "pag_permuto_encode_fwd: table dtype must be F32 or F16");
    PAG_CHECK_ARG(out_dtype == PAG_F32 || out_dtype == PAG_BF16, "pag_permuto_encode_fwd: out dtype must be F32 or BF16");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(tables && out, "pag_permuto_encode_fwd: NULL tables/out");
    PermutoParams p;
    fill_permuto(p, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host);
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(encode_grid(M)), block(256);
    bool launched = false;
    if (table_dtype == PAG_F32 && out_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((permuto_fwd_kernel<float, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, p, (float *)out, out_stride_m, out_stride_c)))
    } else if (table_dtype == PAG_F32 && out_dtype == PAG_BF16) {
        PAG_DISPATCH_ALL((permuto_fwd_kernel<float, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const float *)tables, p, (bf16_t *)out, out_stride_m, out_stride_c)))
    } else if (table_dtype == PAG_F16 && out_dtype == PAG_F32) {
        PAG_DISPATCH_ALL((permuto_fwd_kernel<__half, float, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, p, (float *)out, out_stride_m, out_stride_c)))
    } else {
        PAG_DISPATCH_ALL((permuto_fwd_kernel<__half, bf16_t, F, LPX><<<grid, block, 0, st>>>(xyz, M, (const __half *)tables, p, (bf16_t *)out, out_stride_m, out_stride_c)))
    }
    PAG_CHECK_ARG(launched, "pag_permuto_encode_fwd: unsupported (n_feat=%d, n_levels=%d)", n_feat, n_levels);
    PAG_CHECK_LAUNCH("pag_permuto_encode_fwd");
    return PAG_OK;
}

extern "C" int pag_permuto_encode_bwd(const float *xyz, int64_t M, const void *grad_out, int grad_dtype, int64_t g_stride_m,
                                      int64_t g_stride_c, int n_levels, int n_feat, uint32_t capacity,
                                      const float *scale_factor_host, const float *shift_host, const float *feat_scale_host,
                                      float *grad_tables, void *stream) {
    int rc = check_common("pag_permuto_encode_bwd", xyz, M, n_levels, n_feat);
    if (rc) return rc;
    PAG_CHECK_ARG(capacity >= 1, "pag_permuto_encode_bwd: capacity is 0");
    PAG_CHECK_ARG(scale_factor_host && shift_host, "pag_permuto_encode_bwd: NULL scale_factor/shift");
    PAG_CHECK_ARG(grad_dtype == PAG_F32 || grad_dtype == PAG_BF16, "pag_permuto_encode_bwd: grad dtype must be F32 or BF16");
    if (M == 0) return PAG_OK;
    PAG_CHECK_ARG(grad_out && grad_tables, "pag_permuto_encode_bwd: NULL grad_out/grad_tables");
    PermutoParams p;
    fill_permuto(p, n_levels, n_feat, capacity, scale_factor_host, shift_host, feat_scale_host);
    const int lpx = (n_levels + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(encode_grid(M)), block(256);
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
