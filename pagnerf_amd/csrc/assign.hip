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
                                                         int32_t *__restrict__ counts, int64_t image_stride = 0) {
    __shared__ int32_t rows[256];
    __shared__ int32_t wave_n[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // blockIdx.y = image of a batch (pag_assign_cost): values advance by image_stride elements, every other array is [B, ...] contiguous
        const int64_t b = blockIdx.y, K = gridDim.x;
        values += b * image_stride;
        labels_gt += b * P;
        label_list += b * K;
        sums += b * K * C;
        counts += b * K;
    }
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
        // eight rows requested before the first is added (a row at a time left one 800-byte request in flight per workgroup: 70 us for the 20 MB of a
        // six-image batch); the additions keep their order - row after row - so the sums are the same bits
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const T *row = values + (base + rows[j + u]) * row_stride + col0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = tid + 256 * q;
                    v[u][q] = (q < ncol && c < C) ? pag_ld(row + c) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += v[u][q];
        }
        for (; j < n; ++j) {
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

// ---- the instance term of the late-training step with ONE host synchronisation (pag_assign_cost / pag_assign_nll_*; pagnerf_amd/loss.py) ----------------
// loss/lin_assignment_things.py:23-54 asks the device three times per image before SciPy can run (the sorted unique gt ids, one masked sum per id) and
// the loss that follows is a dozen small tensor ops forward and as many backward: ~0.9 ms on a 3.5 ms step, nearly all of it host latency.
//
// assign_unique_kernel (one workgroup): the distinct positive gt ids of the image through an LDS hash set (1024 slots, 64-bit compare-and-swap), sorted
// ascending by a bitonic network -> labels[0 .. n) = the reference's `sorted(torch.unique(things_gt))[:max_rows]` (:29), the rest of labels[] a sentinel no
// ray carries.  info[0] = n, info[1] = 1 when the image has more distinct ids than the set holds (the caller then takes the general path).
constexpr int ASSIGN_SET = 1024;
constexpr long long ASSIGN_NONE = -(1ll << 62);
__global__ __launch_bounds__(1024) void assign_unique_kernel(const int64_t *__restrict__ labels_gt, int64_t P, int max_rows, int32_t *__restrict__ info,
                                                             int64_t *__restrict__ labels) {
    __shared__ unsigned long long set[ASSIGN_SET];      // 0 = empty (only ids > 0 are inserted)
    __shared__ int32_t over, count;
    const int tid = threadIdx.x;
    labels_gt += (int64_t)blockIdx.x * P;               // one workgroup per image
    info += (int64_t)blockIdx.x * 2;
    labels += (int64_t)blockIdx.x * max_rows;
    set[tid] = 0ull;
    if (tid == 0) over = 0, count = 0;
    __syncthreads();
    for (int64_t p = tid; p < P; p += 1024) {
        const int64_t id = labels_gt[p];
        if (id <= 0) continue;
        const unsigned long long key = (unsigned long long)id;
        unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 54);      // 10 bits
        bool done = false;
        for (int probe = 0; probe < ASSIGN_SET && !done; ++probe) {
            const unsigned long long prev = atomicCAS(&set[h], 0ull, key);
            done = prev == 0ull || prev == key;
            h = (h + 1) & (ASSIGN_SET - 1);
        }
        if (!done) over = 1;                                               // set full: more than 1024 distinct ids
    }
    __syncthreads();
    // empty slots sort to the end: +inf
    unsigned long long v = set[tid] ? set[tid] : ~0ull;
    if (set[tid]) atomicAdd(&count, 1);
    __syncthreads();
    set[tid] = v;
    __syncthreads();
    for (int k = 2; k <= ASSIGN_SET; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int partner = tid ^ j;
            if (partner > tid) {
                const unsigned long long a = set[tid], b = set[partner];
                const bool up = (tid & k) == 0;
                if ((a > b) == up) { set[tid] = b; set[partner] = a; }
            }
            __syncthreads();
        }
    const int n = count < max_rows ? count : max_rows;
    if (tid < max_rows) labels[tid] = tid < n ? (int64_t)set[tid] : (int64_t)ASSIGN_NONE;
    if (tid == 0) {
        info[0] = n;
        info[1] = over;
    }
}
// cost[r, :] = -(sum / (count + 1e-4)) of labels[r] (:31-33: int64 count + python float -> fp32, fp32 division); rows past info[0] are left alone
__global__ __launch_bounds__(256) void assign_cost_kernel(const float *__restrict__ sums, const int32_t *__restrict__ counts, int C, const int32_t *__restrict__ info,
                                                          float *__restrict__ cost) {
    const int r = blockIdx.x;
    const int64_t b = blockIdx.y, R = gridDim.x;
    if (r >= info[b * 2]) return;
    const float den = __fadd_rn((float)counts[b * R + r], 1e-4f);
    for (int c = threadIdx.x; c < C; c += 256) cost[(b * R + r) * C + c] = -__fdiv_rn(sums[(b * R + r) * C + c], den);
}

// Outlier rejection (utils/outlier_rejection.py:8-51 on the per-id centres of :56-71): the range [lo, hi] of ids an instance at centre x may take, with the
// tensor ops' own fp32 arithmetic - centre = sum_x / count, x = (-centre + 1) / 2, lo = (int64) clamp(slope * remainder(x, x_limit), 0, n_ids - 1) (torch's
// remainder: fmod, moved by the divisor when the signs differ), hi = clamp(lo + margin, 0, n_ids - 1).  One workgroup per image, a lane per label.
__global__ __launch_bounds__(256) void assign_id_range_kernel(const float *__restrict__ psums, const int32_t *__restrict__ pcounts, int R, float slope, float x_limit,
                                                              int id_margin, int n_ids, int32_t *__restrict__ lo_hi) {
    const int64_t b = blockIdx.x;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float centre = __fdiv_rn(psums[(b * R + r) * 3], (float)pcounts[b * R + r]);
        const float x = __fdiv_rn(__fadd_rn(-centre, 1.0f), 2.0f);
        float m = fmodf(x, x_limit);
        if (m != 0.0f && ((x_limit < 0.0f) != (m < 0.0f))) m = __fadd_rn(m, x_limit);
        const float v = fminf(fmaxf(__fmul_rn(slope, m), 0.0f), (float)(n_ids - 1));
        const long long lo = (long long)v;                                      // rows without rays: NaN centre, never read by the caller
        long long hi = lo + id_margin;
        hi = hi < 0 ? 0 : (hi > n_ids - 1 ? n_ids - 1 : hi);
        lo_hi[(b * R + r) * 2] = (int32_t)lo;
        lo_hi[(b * R + r) * 2 + 1] = (int32_t)hi;
    }
}

// One wave per ray: valid = stuff | gt > 0 (:60), virtual label = targets[r] for gt == labels[r] (the assignment's relabelling, :47-53; ids without a row
// take `deflt`), else 0; first arg-max of the ray's probabilities (torch.argmax: lowest index of the maximum); nll = -log(p[virtual] + 1e-27) (:80);
// *wrong |= valid && virtual != arg-max (:79).  assign_nll_finish_kernel then keeps nll where the image has a wrong ray.
__global__ __launch_bounds__(256) void assign_nll_fwd_kernel(const float *__restrict__ prob, int64_t P, int64_t row_stride, int n_cols,
                                                             const int64_t *__restrict__ labels_gt, const uint8_t *__restrict__ stuff,
                                                             const int64_t *__restrict__ labels, const int64_t *__restrict__ targets, const int32_t *__restrict__ info,
                                                             int64_t deflt, int64_t *__restrict__ virt, float *__restrict__ nll, uint8_t *__restrict__ valid,
                                                             int32_t *__restrict__ wrong, int64_t image_stride, int max_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= P) return;
    {   // blockIdx.y = image
        const int64_t b = blockIdx.y;
        prob += b * image_stride;
        labels_gt += b * P;
        if (stuff) stuff += b * P;
        labels += b * max_rows;
        targets += b * max_rows;
        info += b * 2;
        virt += b * P;
        nll += b * P;
        valid += b * P;
        wrong += b;
    }
    const int64_t gt = labels_gt[ray];
    const bool things = gt > 0;
    const bool ok = things || (stuff && stuff[ray] != 0);
    int64_t v = 0;
    if (things) {                                      // the assignment's relabelling: labels[] is sorted, ids it does not hold keep `deflt`
        int lo = 0, hi = info[0];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (labels[mid] < gt) lo = mid + 1; else hi = mid;
        }
        v = (lo < info[0] && labels[lo] == gt) ? targets[lo] : deflt;
    }
    const float *row = prob + ray * row_stride;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    bool any_nan = false;
    for (int c = lane; c < n_cols; c += 64) {
        const float x = row[c];
        if (x != x) { if (!any_nan) { any_nan = true; arg = c; } }           // torch.argmax: NaN is the maximum, the first one wins
        else if (!any_nan && x > best) { best = x; arg = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d);
        const int oa = __shfl_xor(arg, d);
        const int on = __shfl_xor((int)any_nan, d);
        const bool take = (on && !any_nan) || (on == (int)any_nan && ((!any_nan && (ob > best || (ob == best && oa < arg))) || (any_nan && oa < arg)));
        if (take) { best = ob; arg = oa; any_nan = on != 0; }
    }
    if (lane == 0) {
        virt[ray] = v;
        valid[ray] = ok ? 1 : 0;
        const float pv = (v >= 0 && v < n_cols) ? row[v] : 0.0f;
        nll[ray] = -logf(__fadd_rn(pv, 1e-27f));
        // the image's `any wrong` flag: a plain store of 1 by whoever finds it still 0 (benign race: every writer stores the same value).  One
        // atomicOr per wrong ray - 24 576 of them on six words in an untrained step - serialised in the L2: 0.2 ms of a 0.01 ms kernel.
        if (ok && v != (int64_t)arg && __hip_atomic_load(wrong, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(wrong, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(256) void assign_nll_finish_kernel(float *__restrict__ nll, const uint8_t *__restrict__ valid, const int32_t *__restrict__ wrong, int64_t P) {
    const int64_t ray = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ray >= P) return;
    const int64_t b = blockIdx.y;
    if (!(valid[b * P + ray] && wrong[b])) nll[b * P + ray] = 0.0f;
}
// d loss_ray / d prob[ray, :] : -g / (p + 1e-27) in the virtual label's column of the rays the finish pass kept, zero everywhere else (rows written whole)
__global__ __launch_bounds__(256) void assign_nll_bwd_kernel(const float *__restrict__ prob, int64_t P, int64_t row_stride, int n_cols, const int64_t *__restrict__ virt,
                                                             const uint8_t *__restrict__ valid, const int32_t *__restrict__ wrong, const float *__restrict__ grad,
                                                             float *__restrict__ d_prob, int64_t image_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= P) return;
    {   // blockIdx.y = image
        const int64_t b = blockIdx.y;
        prob += b * image_stride;
        virt += b * P;
        valid += b * P;
        wrong += b;
        grad += b * P;
        d_prob += b * P * n_cols;
    }
    const bool on = valid[ray] && *wrong;
    const int64_t v = virt[ray];
    const float g = on && v >= 0 && v < n_cols ? -__fdiv_rn(grad[ray], __fadd_rn(prob[ray * row_stride + v], 1e-27f)) : 0.0f;
    for (int c = lane; c < n_cols; c += 64) d_prob[ray * n_cols + c] = (on && c == v) ? g : 0.0f;
}


// ------------------------------------------------------------------------------------------------- the assignment itself, on the device
// scipy.optimize.linear_sum_assignment(np.nan_to_num(cost)) of loss/lin_assignment_things.py:45 (loss/lin_assignment.py:22) without the host: one WAVE per image runs
// SciPy's own algorithm (rectangular_lsap.cpp: shortest augmenting paths after Crouse 2016, restated sequentially in oracle/lin_assign.py::lsap_jv, which the CPU
// suite pins against the installed SciPy) in float64 with SciPy's operation order, so that the assigned columns are the same integers:
//   per row `cur`: columns not yet scanned live in `remaining` (filled in reverse order, removed by swap-with-last); every pass over them relaxes
//   spc[j] = min(spc[j], (minVal + c[i][j] - u[i]) - v[j]) and selects the column of the lowest spc - among equal ones a column WITHOUT a row wins, the last such in
//   scan order, else the first of the minimum (SciPy's sequential `<` / `==` rule, evaluated here as (min, first position of the min, last unassigned position of
//   the min) over the lanes); dual update; augmentation along path[].
// The 64 lanes share a pass (position `it` = lane + 64 k); everything else is as sequential as SciPy's loop.  cost is the fp32 matrix pag_assign_cost wrote, widened to
// float64 as `.astype(np.float64)` does, with the outlier-rejection mask (10000 outside [lo, hi]) and nan_to_num applied on the fly.
constexpr int SOLVE_MAX = 256;          // rows and columns one wave handles (I <= 257: BUP20 has 200)
constexpr int SOLVE_STAGE = 12288;      // fp32 entries of the cost matrix kept in LDS (48 KiB: 61 labels x 199 columns); larger ones are read from memory

// Register-resident form: lane l owns columns l, l + 64, l + 128, l + 192 (spc, v, row4col, path and the column's POSITION in SciPy's `remaining` array live in
// registers; a removal moves the tail column into the freed position: two compares per lane) and rows l, l + 64, ... (u, col4row).  One pass = one LDS read
// of the cost row, the relaxation, and TWO wave reductions on DPP moves: the minimum of the float64 path costs (compared as float64, as SciPy does), then - among the lanes
// that hold it - the minimum of q, which encodes SciPy's tie rule in one number: a column without a row gets q = 255 - position (the LAST free one in scan order
// wins), a column with a row q = 256 + position (only if no free column shares the minimum; the FIRST in scan order wins).  ~3x fewer cycles per pass than the
// first form (every per-column array in LDS, six rounds of shuffles on a three-field record): 0.44 -> see profiles/README.md round 6.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t solve_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);      // lanes without a source keep their own value
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void solve_minf64_step(double &v) {      // float64 compare itself (as SciPy's `<`): two DPP moves, one compare, two selects
    const long long b = __double_as_longlong(v);
    const uint32_t ohi = solve_dpp<CTRL, ROW_MASK>((uint32_t)(b >> 32)), olo = solve_dpp<CTRL, ROW_MASK>((uint32_t)b);
    const double o = __longlong_as_double((long long)(((uint64_t)ohi << 32) | olo));
    v = o < v ? o : v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void solve_min32_step(uint32_t &v) {
    const uint32_t o = solve_dpp<CTRL, ROW_MASK>(v);
    v = o < v ? o : v;
}
template <typename T>
__device__ __forceinline__ T solve_pick4(const T (&a)[4], int k) {   // k is wave-uniform
    return k == 0 ? a[0] : (k == 1 ? a[1] : (k == 2 ? a[2] : a[3]));
}

__global__ __launch_bounds__(64) void assign_solve_kernel(const float *__restrict__ cost, int R, int C, const int32_t *__restrict__ info,
                                                          const int32_t *__restrict__ lo_hi, int64_t *__restrict__ targets, int32_t *__restrict__ status) {
    constexpr int K = SOLVE_MAX / 64;
    static_assert(K == 4, "the pass combines its four columns pairwise");
    // the cost matrix as fp32 in LDS (mask and NaN -> 0 applied; an infinity is kept as such and becomes +-DBL_MAX when it is widened): 61 labels x 199 columns
    __shared__ float stage[SOLVE_STAGE];
    double *stage64 = reinterpret_cast<double *>(stage);       // the same 48 KiB as float64 when n * C <= SOLVE_STAGE / 2 (30 labels x 199): widened once, not per pass
    __shared__ double spc_l[SOLVE_MAX], u_l[SOLVE_MAX];
    __shared__ int32_t path_l[SOLVE_MAX], row4col_l[SOLVE_MAX], col4row_l[SOLVE_MAX];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = info[b * 2], over = info[b * 2 + 1];
    cost += (int64_t)b * R * C;
    targets += (int64_t)b * R;
    if (lo_hi) lo_hi += (int64_t)b * R * 2;
    for (int r = lane; r < R; r += 64) targets[r] = 1;          // ids that get no column keep 0 + 1 (:47-53)
    if (over || n > R || n > C) {                                // more distinct ids than pag_assign_cost's set holds (or a shape SciPy would transpose): not solved here
        if (lane == 0) status[b] = 1;
        return;
    }
    if (lane == 0) status[b] = 0;
    if (n <= 0) return;
    auto widen = [&](int i, int j) -> double {
        double d = (double)cost[(int64_t)i * C + j];
        if (lo_hi && !(lo_hi[i * 2] <= j && j <= lo_hi[i * 2 + 1])) d = 10000.0;       // utils/outlier_rejection.py:8-51
        if (d != d) d = 0.0;                                                          // np.nan_to_num
        else if (d == INFINITY) d = 1.7976931348623157e308;
        else if (d == -INFINITY) d = -1.7976931348623157e308;
        return d;
    };
    const bool staged = n * C <= SOLVE_STAGE;
    const bool staged64 = 2 * n * C <= SOLVE_STAGE;
    if (staged) {
        // four rows per trip: 16 independent loads in flight per lane (a row per trip cost one memory round trip per label: ~45 us of a 55 us solve)
        for (int r0 = 0; r0 < n; r0 += 4) {
            float f[4][K];
            int lo[4], hi[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int r = r0 + a < n ? r0 + a : n - 1;
                lo[a] = lo_hi ? lo_hi[r * 2] : 0;
                hi[a] = lo_hi ? lo_hi[r * 2 + 1] : C;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int j = lane + 64 * k;
                    f[a][k] = cost[(int64_t)r * C + (j < C ? j : C - 1)];
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (r0 + a >= n) break;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int j = lane + 64 * k;
                    float g = f[a][k];
                    if (!(lo[a] <= j && j <= hi[a])) g = 10000.0f;
                    g = g != g ? 0.0f : g;
                    if (j < C) {
                        if (staged64) stage64[(r0 + a) * C + j] = g == INFINITY ? 1.7976931348623157e308 : (g == -INFINITY ? -1.7976931348623157e308 : (double)g);
                        else stage[(r0 + a) * C + j] = g;
                    }
                }
            }
        }
    }
    double spc[K], v[K], u[K];
    int r4c[K], pos[K], pth[K], c4r[K];
    uint32_t qk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = 0.0, u[k] = 0.0, r4c[k] = -1, pth[k] = -1, c4r[k] = -1, spc[k] = INFINITY, pos[k] = -2;
        const int j = lane + 64 * k;
        if (j < C) row4col_l[j] = -1;
        if (j < n) col4row_l[j] = -1, u_l[j] = 0.0;
    }
    __syncthreads();
    for (int cur = 0; cur < n; ++cur) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = lane + 64 * k;
            spc[k] = INFINITY;
            pos[k] = j < C ? C - 1 - j : -2;                    // SciPy fills `remaining` in reverse order; -1 = scanned (SC), -2 = no such column
            // the tie code of an unscanned column (see above): kept in a register, touched only when the column's position changes
            qk[k] = j < C ? (r4c[k] == -1 ? (uint32_t)(255 - pos[k]) : (uint32_t)(256 + pos[k])) : 0xFFFFFFFFu;
        }
        unsigned sr_bits = 0;
        int n_rem = C, i = cur, sink = -1;
        double min_val = 0.0;
        while (sink < 0) {
            const int irow_lane = i & 63, irow_k = i >> 6;
            if (lane == irow_lane) sr_bits |= 1u << irow_k;                       // SR[i] = true
            const double ui = u_l[i];                                            // an LDS broadcast (the register copy sat behind a four-way branch on i >> 6)
            double bm = INFINITY;
            uint32_t bq = 0xFFFFFFFFu;
            if (staged) {          // wave-uniform; the pass itself is straight-line code: selects, no branches
                double sk[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int j = lane + 64 * k;
                    const int e = i * C + (j < C ? j : 0);
                    double c;
                    if (staged64) {
                        c = stage64[e];
                    } else {
                        const float cf = stage[e];
                        c = (double)cf;
                        c = cf == INFINITY ? 1.7976931348623157e308 : (cf == -INFINITY ? -1.7976931348623157e308 : c);
                    }
                    const double r = ((min_val + c) - ui) - v[k];
                    const bool act = qk[k] != 0xFFFFFFFFu;                       // unscanned and existing
                    const bool better = act & (r < spc[k]);          // bitwise on purpose: `&&` / `||` become exec-mask branches in this loop
                    spc[k] = better ? r : spc[k];
                    pth[k] = better ? i : pth[k];
                    sk[k] = act ? spc[k] : (double)INFINITY;
                }
                // pairwise: (0,1), (2,3), then the two winners - two dependent compares instead of four
                const bool t01 = (sk[1] < sk[0]) | ((sk[1] == sk[0]) & (qk[1] < qk[0]));
                const bool t23 = (sk[3] < sk[2]) | ((sk[3] == sk[2]) & (qk[3] < qk[2]));
                const double m01 = t01 ? sk[1] : sk[0], m23 = t23 ? sk[3] : sk[2];
                const uint32_t q01 = t01 ? qk[1] : qk[0], q23 = t23 ? qk[3] : qk[2];
                const bool tt = (m23 < m01) | ((m23 == m01) & (q23 < q01));
                bm = tt ? m23 : m01;
                bq = tt ? q23 : q01;
            } else {
#pragma unroll 1
                for (int k = 0; k < K; ++k) {
                    if (pos[k] >= 0) {
                        const int j = lane + 64 * k;
                        const double r = ((min_val + widen(i, j)) - ui) - v[k];
                        if (r < spc[k]) {
                            spc[k] = r;
                            pth[k] = i;
                        }
                        const uint32_t q = qk[k];
                        if (spc[k] < bm || (spc[k] == bm && q < bq)) bm = spc[k], bq = q;
                    }
                }
            }
            double wm = bm;
            solve_minf64_step<0x111, 0xF>(wm);
            solve_minf64_step<0x112, 0xF>(wm);
            solve_minf64_step<0x114, 0xF>(wm);
            solve_minf64_step<0x118, 0xF>(wm);
            solve_minf64_step<0x142, 0xA>(wm);
            solve_minf64_step<0x143, 0xC>(wm);
            {
                const long long bits = __double_as_longlong(wm);
                const int lo = __builtin_amdgcn_readlane((int)bits, 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
                min_val = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
            }
            if (!(min_val < INFINITY)) {                                        // infeasible (SciPy raises): leave the defaults, report
                if (lane == 0) status[b] = 2;
                return;
            }
            uint32_t q = bm == min_val ? bq : 0xFFFFFFFFu;
            solve_min32_step<0x111, 0xF>(q);
            solve_min32_step<0x112, 0xF>(q);
            solve_min32_step<0x114, 0xF>(q);
            solve_min32_step<0x118, 0xF>(q);
            solve_min32_step<0x142, 0xA>(q);
            solve_min32_step<0x143, 0xC>(q);
            const int qmin = __builtin_amdgcn_readlane((int)q, 63);
            const bool free_col = qmin < 256;
            const int index = free_col ? 255 - qmin : qmin - 256;
            // the column at position `index`: exactly one (lane, k)
            int myk = -1, myr = -1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bool hit = pos[k] == index;
                myk = hit ? k : myk;
                myr = hit ? r4c[k] : myr;
            }
            const int jl = __builtin_ctzll(__ballot(myk >= 0));
            const int jsel = jl + 64 * __builtin_amdgcn_readlane(myk, jl), owner = __builtin_amdgcn_readlane(myr, jl);
            const int tail = n_rem - 1;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bool gone = pos[k] == index, moved = (pos[k] == tail) & !gone;      // SC[j] = true; remaining[index] = remaining[--n_rem]
                pos[k] = gone ? -1 : (moved ? index : pos[k]);
                // q = 255 - pos (free) or 256 + pos: the moved column's position drops from `tail` to `index`
                qk[k] = gone ? 0xFFFFFFFFu : (moved ? (qk[k] < 256u ? qk[k] + (uint32_t)(tail - index) : qk[k] - (uint32_t)(tail - index)) : qk[k]);
            }
            --n_rem;
            if (free_col) sink = jsel; else i = owner;
        }
        // dual variables, then the augmentation along path[] (sequential, as SciPy's)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = lane + 64 * k;
            if (j < C) spc_l[j] = spc[k], path_l[j] = pth[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int r = lane + 64 * k;
            if (r < n) {
                if (r == cur) u[k] += min_val;
                else if ((sr_bits >> k) & 1u) u[k] += min_val - spc_l[c4r[k]];
                u_l[r] = u[k];
            }
            if (pos[k] == -1) v[k] -= min_val - spc[k];
        }
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int r = path_l[j];
                row4col_l[j] = r;
                const int t = col4row_l[r];
                col4row_l[r] = j;
                j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = lane + 64 * k;
            if (j < C) r4c[k] = row4col_l[j];
            if (j < n) c4r[k] = col4row_l[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int r = lane + 64 * k;
        if (r < n) targets[r] = (int64_t)c4r[k] + 1;                           // :47-53: assigned column + 1
    }
}

}  // namespace

extern "C" int pag_assign_cost(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, int col0, const int64_t *labels_gt,
                               int max_rows, float *sums_ws, int32_t *counts_ws, int32_t *info, int64_t *labels, float *cost, const float *points, float id_slope,
                               float id_x_limit, int id_margin, float *psums_ws, int32_t *pcounts_ws, int32_t *id_lo_hi, void *stream) {
    PAG_CHECK_ARG(P >= 0 && B >= 0 && B <= 65535, "pag_assign_cost: P < 0 or B not in [0,65535]");
    PAG_CHECK_ARG(col0 >= 0 && n_cols - col0 >= 1 && n_cols - col0 <= 1024 && row_stride >= n_cols, "pag_assign_cost: columns");
    PAG_CHECK_ARG(max_rows >= 1 && max_rows <= 1024, "pag_assign_cost: max_rows %d not in [1,1024]", max_rows);
    if (B == 0) return PAG_OK;
    PAG_CHECK_ARG(sums_ws && counts_ws && info && labels && cost && (P == 0 || (prob && labels_gt)), "pag_assign_cost: NULL input/output");
    hipStream_t st = (hipStream_t)stream;
    const int C = n_cols - col0;
    hipLaunchKernelGGL(assign_unique_kernel, dim3(B), dim3(1024), 0, st, labels_gt, P, max_rows, info, labels);
    hipLaunchKernelGGL(label_sums_kernel<float>, dim3(max_rows, B), dim3(256), 0, st, prob, P, row_stride, col0, C, labels_gt, (const uint8_t *)nullptr,
                       (const int64_t *)labels, sums_ws, counts_ws, image_stride);
    hipLaunchKernelGGL(assign_cost_kernel, dim3(max_rows, B), dim3(256), 0, st, (const float *)sums_ws, (const int32_t *)counts_ws, C, (const int32_t *)info, cost);
    if (points) {
        PAG_CHECK_ARG(psums_ws && pcounts_ws && id_lo_hi, "pag_assign_cost: points without psums_ws / pcounts_ws / id_lo_hi");
        hipLaunchKernelGGL(label_sums_kernel<float>, dim3(max_rows, B), dim3(256), 0, st, points, P, (int64_t)3, 0, 3, labels_gt, (const uint8_t *)nullptr,
                           (const int64_t *)labels, psums_ws, pcounts_ws, P * 3);
        hipLaunchKernelGGL(assign_id_range_kernel, dim3(B), dim3(256), 0, st, (const float *)psums_ws, (const int32_t *)pcounts_ws, max_rows, id_slope, id_x_limit, id_margin,
                           C, id_lo_hi);
    }
    PAG_CHECK_LAUNCH("pag_assign_cost");
    return PAG_OK;
}

extern "C" int pag_assign_solve(const float *cost, int B, int max_rows, int n_ids, const int32_t *info, const int32_t *id_lo_hi, int64_t *targets, int32_t *status,
                                void *stream) {
    PAG_CHECK_ARG(B >= 0 && B <= 65535, "pag_assign_solve: B not in [0,65535]");
    PAG_CHECK_ARG(max_rows >= 1 && max_rows <= SOLVE_MAX && n_ids >= 1 && n_ids <= SOLVE_MAX, "pag_assign_solve: max_rows %d / n_ids %d not in [1,%d]", max_rows, n_ids,
                  SOLVE_MAX);
    if (B == 0) return PAG_OK;
    PAG_CHECK_ARG(cost && info && targets && status, "pag_assign_solve: NULL input/output");
    hipLaunchKernelGGL(assign_solve_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, cost, max_rows, n_ids, info, id_lo_hi, targets, status);
    PAG_CHECK_LAUNCH("pag_assign_solve");
    return PAG_OK;
}

extern "C" int pag_assign_nll_fwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, const int64_t *labels_gt,
                                  const uint8_t *stuff_mask, const int64_t *labels, const int64_t *targets, const int32_t *info, int max_rows, int64_t default_label,
                                  int64_t *virt, float *nll, uint8_t *valid, int32_t *wrong, void *stream) {
    PAG_CHECK_ARG(P >= 0 && B >= 0 && B <= 65535 && n_cols >= 1 && row_stride >= n_cols && max_rows >= 1, "pag_assign_nll_fwd: sizes");
    if (P == 0 || B == 0) return PAG_OK;
    PAG_CHECK_ARG(prob && labels_gt && labels && targets && info && virt && nll && valid && wrong, "pag_assign_nll_fwd: NULL input/output");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(assign_nll_fwd_kernel, dim3((unsigned)((P + 3) / 4), B), dim3(256), 0, st, prob, P, row_stride, n_cols, labels_gt, stuff_mask, labels, targets, info,
                       default_label, virt, nll, valid, wrong, image_stride, max_rows);
    hipLaunchKernelGGL(assign_nll_finish_kernel, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, st, nll, (const uint8_t *)valid, (const int32_t *)wrong, P);
    PAG_CHECK_LAUNCH("pag_assign_nll_fwd");
    return PAG_OK;
}

extern "C" int pag_assign_nll_bwd(const float *prob, int B, int64_t P, int64_t image_stride, int64_t row_stride, int n_cols, const int64_t *virt, const uint8_t *valid,
                                  const int32_t *wrong, const float *grad, float *d_prob, void *stream) {
    PAG_CHECK_ARG(P >= 0 && B >= 0 && B <= 65535 && n_cols >= 1 && row_stride >= n_cols, "pag_assign_nll_bwd: sizes");
    if (P == 0 || B == 0) return PAG_OK;
    PAG_CHECK_ARG(prob && virt && valid && wrong && grad && d_prob, "pag_assign_nll_bwd: NULL input/output");
    hipLaunchKernelGGL(assign_nll_bwd_kernel, dim3((unsigned)((P + 3) / 4), B), dim3(256), 0, (hipStream_t)stream, prob, P, row_stride, n_cols, virt, valid, wrong, grad,
                       d_prob, image_stride);
    PAG_CHECK_LAUNCH("pag_assign_nll_bwd");
    return PAG_OK;
}

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
