// Per-level vertex lookup shared by the encode forward / backward kernels (gfx950).
// Numerics: explicit round-to-nearest intrinsics in the op order of the oracle
// (oracle/hash_encode.py, oracle/permuto_encode.py); translation units that include this are
// compiled with -ffp-contract=off.
#pragma once
#include "common.h"

namespace pag_enc {

struct HashParams {
    float res[PAG_MAX_LEVELS];
    float scale[PAG_MAX_FEATS];
    int L, log2T, has_scale;
    int pair_loads = 0;      // forward only: tables are 16-byte aligned and have >= 2 rows - x-corner pairs may be fetched as one float4
    int half_coords = 0;     // PAG_ENC_HALF_COORDS
};

struct PermutoParams {
    float sf[PAG_MAX_LEVELS][3];
    float shift[PAG_MAX_LEVELS][3];
    float scale[PAG_MAX_FEATS];
    int L, has_scale;
    uint32_t capacity, pow2mask;   // pow2mask = capacity-1 when capacity is a power of two, else 0
    int half_coords = 0;           // PAG_ENC_HALF_COORDS
};

// One sample's coordinates; with PAG_ENC_HALF_COORDS rounded through fp16 first (round-to-nearest-even, as torch's .half()):
// grids/permuto_grid.py:65,71 - the encoder sees float(half(coords)) under the trainer's autocast.
__device__ __forceinline__ void load_xyz(const float *__restrict__ xyz, int64_t i, int half_coords, float (&x)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = xyz[i * 3 + a];
        x[a] = half_coords ? __half2float(__float2half_rn(v)) : v;
    }
}

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

// Row `row` of a table whose base is wave-uniform: the byte offset stays in 32 bits (a level's table is far below 4 GiB), so the load takes
// the scalar-base + 32-bit-vector-offset form and its address costs one VGPR instead of two and no 64-bit add.
template <int F, typename T> __device__ __forceinline__ void gather_row(const T *base, uint32_t row, float (&v)[F]) {
    gather<F>(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + row * (uint32_t)(F * sizeof(T))), v);
}

// ------------------------------------------------------------------------------------ hash grid
// grids/hash_grid_torch.py:26-46 (cell lookup) and :69-77 (weights), one level.
// dwdx[a] = d w[a] / d x[a] = 1 / (vmax - vmin)  (the clamp of :34-36 is local to the cell lookup, :100,105)
__device__ __forceinline__ void hash_cell(const float (&x)[3], float res, int log2T, uint32_t (&idx)[8], float (&w)[3],
                                          float (&dwdx)[3]) {
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
        dwdx[a] = __fdiv_rn(1.0f, __fsub_rn(vmax, vmin));
        c[a] = (uint32_t)bl;
    }
    const uint32_t mask = (1u << log2T) - 1u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // corner k = 4i + 2j + kk (hash_grid_torch.py:10)
        uint32_t cx = c[0] + ((k >> 2) & 1), cy = c[1] + ((k >> 1) & 1), cz = c[2] + (k & 1);
        idx[k] = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & mask;
    }
}

__device__ __forceinline__ void hash_cell(const float (&x)[3], float res, int log2T, uint32_t (&idx)[8], float (&w)[3]) {
    float unused[3];
    hash_cell(x, res, log2T, idx, w, unused);
}

__device__ __forceinline__ float lerp_ref(float a, float b, float w, float omw) {
    return __fadd_rn(__fmul_rn(a, omw), __fmul_rn(b, w));
}

// --------------------------------------------------------------------------- permutohedral lattice
// oracle/permuto_encode.py lattice_simplex() + vertex_indices(), one level, d = 3.
// slot[a] = 3 - rank[a]: coordinate a of the elevated point adds +delta_a to bary[slot] and -delta_a to bary[slot+1]
// (bary[4] folds into bary[0]) - what d bary / d x needs.
// LDS_SORT (the VALU-bound bin pass): the two rank-indexed selections below - the barycentric deltas sorted by rank and the hash
// constant of the coordinate that holds each rank, 12 compares + 21 selects of the expensive VOP3 kind - become four 2-dword LDS
// writes at slot rank[a] of a lane-private 32-byte record and two 16-byte reads: the same values moved, not recomputed.
template <bool LDS_SORT = false>
__device__ __forceinline__ void permuto_simplex(const float (&x)[3], const float (&sh)[3], const float (&sf)[3],
                                                uint32_t capacity, uint32_t pow2mask, uint32_t (&idx)[4], float (&bary)[4],
                                                int (&slot_out)[4], float *lane_slots = nullptr) {
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

    // Nearest remainder-0 point, per coordinate "up if strictly closer to up, else down" (oracle: up = ceil(v)*4, dn = floor(v)*4).
    // up is written dn + 4: when v is an integer the oracle's up equals dn and both differences are 0 (-> dn); with dn + 4 the
    // test reads 4 < 0 (-> dn) - the same choice, one instruction less, and dn + 4 is exact.  The residual E - r is exact
    // (|E - r| <= 2 and r is E's nearest multiple of 4: Sterbenz), so it is -(up - E) or (E - dn): a select instead of a subtraction.
    int rem0[4], rank[4];
    float resid[4];
    int sum = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float dn = floorf(E[a] * 0.25f) * 4.0f, up = dn + 4.0f;
        const float d_up = __fsub_rn(up, E[a]), d_dn = __fsub_rn(E[a], dn);
        const bool take_up = d_up < d_dn;
        const float rf = take_up ? up : dn;         // the chosen remainder-0 point, exact in fp32
        rem0[a] = (int)rf;
        sum += rem0[a];
        resid[a] = __fsub_rn(E[a], rf);             // = -d_up or d_dn bit for bit (fl(E - up) = -fl(up - E)): one subtraction instead of a select + add
    }
    sum >>= 2;   // exact: every rem0 is a multiple of 4
    // rank_a = #{b : resid_b > resid_a} + #{b < a : resid_b == resid_a} (the oracle's pairwise rule: for a < b, `lt = resid_a <
    // resid_b` adds to rank_a, its negation to rank_b).  lt is the sign bit of resid_a - resid_b: the difference of two floats is
    // exactly 0 only when they are equal (gradual underflow), so n_ab = (resid_a - resid_b) >> 31 is -1 or 0 = -lt.
    {
        const int n01 = __float_as_int(__fsub_rn(resid[0], resid[1])) >> 31, n02 = __float_as_int(__fsub_rn(resid[0], resid[2])) >> 31;
        const int n03 = __float_as_int(__fsub_rn(resid[0], resid[3])) >> 31, n12 = __float_as_int(__fsub_rn(resid[1], resid[2])) >> 31;
        const int n13 = __float_as_int(__fsub_rn(resid[1], resid[3])) >> 31, n23 = __float_as_int(__fsub_rn(resid[2], resid[3])) >> 31;
        rank[0] = -(n01 + n02 + n03);
        rank[1] = 1 + n01 - (n12 + n13);
        rank[2] = 2 + n02 + n12 - n23;
        rank[3] = 3 + n03 + n13 + n23;
    }
    // rank += sum, wrapped into 0..3 with rem0 moved by the same amount.  |sum| <= 2 (four roundings of < 2 each on
    // coordinates that add up to 0), so rank + sum lies in [-2, 5] and one +-4 step is the whole wrap: t & 3.
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int t = rank[a] + sum;
        rank[a] = t & 3;
        rem0[a] += rank[a] - t;
    }
    // Barycentric weights.  The oracle adds +delta_a to slot 3-rank_a and -delta_a to slot 4-rank_a in coordinate order;
    // the ranks are a permutation of 0..3, so every slot receives exactly one +delta and/or one -delta and the first of
    // the two lands on 0.0 exactly: slot s = fl(delta[rank 3-s] - delta[rank 4-s]) whatever the order, and
    // slot 0 = fl(delta[rank 3] + fl(1 - delta[rank 0])).  Sorting the four deltas by rank (16 selects) replaces the
    // 40 predicated slot updates and is bit-identical (the kernels were VALU-bound: 86 % VALUBusy on the forward).
    float d[4], dr[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        d[a] = __fsub_rn(E[a], (float)rem0[a]) * 0.25f;
        slot_out[a] = 3 - rank[a];
    }
    constexpr uint32_t m1 = 2531011u, m2 = m1 * m1, m3 = m2 * m1;
    constexpr uint32_t step = m3 + m2 + m1;
    constexpr uint32_t A[4] = {4u * m3, 4u * m2, 4u * m1, 0u};
    uint32_t B[4];      // LDS_SORT: holds step - 4*m^(3-a) (the increment itself), else 4*m^(3-a)
    if constexpr (LDS_SORT) {
        // record: dwords 0-3 = delta of rank 0..3, dwords 4-7 = 4*m^(3-a) of the coordinate a with rank 0..3 (0 for a = 3)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            lane_slots[rank[a]] = d[a];
            lane_slots[4 + rank[a]] = __uint_as_float(step - A[a]);
        }
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v0 = *reinterpret_cast<const f4 *>(lane_slots), v1 = *reinterpret_cast<const f4 *>(lane_slots + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dr[r] = v0[r];
            B[r] = __float_as_uint(v1[r]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) dr[r] = rank[0] == r ? d[0] : (rank[1] == r ? d[1] : (rank[2] == r ? d[2] : d[3]));
    }
    bary[0] = __fadd_rn(dr[3], __fsub_rn(1.0f, dr[0]));
    bary[1] = __fsub_rn(dr[2], dr[3]);
    bary[2] = __fsub_rn(dr[1], dr[2]);
    bary[3] = __fsub_rn(dr[0], dr[1]);
    // Vertex hashes.  k = ((key0*m + key1)*m + key2)*m is linear in the keys (mod 2^32) and
    // key_a(r) = rem0_a + r - 4*[rank_a > 3-r], so k(r) = k(rem0) + r*(m^3+m^2+m) - sum_a [rank_a >= 4-r] * 4*m^(3-a):
    // three multiplies per level instead of twelve.
    const uint32_t h0 = (((uint32_t)rem0[0] * m1 + (uint32_t)rem0[1]) * m1 + (uint32_t)rem0[2]) * m1;
    // the sets {a: rank_a >= 4-r} are nested in r and at most one coordinate holds each rank: B[q] = 4*m^(3-a) of the
    // coordinate a < 3 with rank q (the compares are the ones the delta sort above already made)
    if constexpr (!LDS_SORT) {
#pragma unroll
        for (int q = 1; q < 4; ++q) B[q] = rank[0] == q ? A[0] : (rank[1] == q ? A[1] : (rank[2] == q ? A[2] : 0u));
    }
    idx[0] = h0;
#pragma unroll
    for (int r = 1; r < 4; ++r) idx[r] = idx[r - 1] + (LDS_SORT ? B[4 - r] : step - B[4 - r]);
    if (pow2mask) {      // one wave-uniform branch for the four vertices
#pragma unroll
        for (int r = 0; r < 4; ++r) idx[r] &= pow2mask;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) idx[r] %= capacity;
    }
}

__device__ __forceinline__ void permuto_simplex(const float (&x)[3], const float (&sh)[3], const float (&sf)[3],
                                                uint32_t capacity, uint32_t pow2mask, uint32_t (&idx)[4], float (&bary)[4]) {
    int unused[4];
    permuto_simplex<false>(x, sh, sf, capacity, pow2mask, idx, bary, unused);
}
__device__ __forceinline__ void permuto_simplex_lds(const float (&x)[3], const float (&sh)[3], const float (&sf)[3], uint32_t capacity,
                                                    uint32_t pow2mask, uint32_t (&idx)[4], float (&bary)[4], float *lane_slots) {
    int unused[4];
    permuto_simplex<true>(x, sh, sf, capacity, pow2mask, idx, bary, unused, lane_slots);
}

}  // namespace pag_enc
