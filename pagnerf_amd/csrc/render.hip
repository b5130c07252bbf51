// Packed ray march ('ray' mode) and alpha compositing for gfx950.
//
// Compositing = a segmented exclusive scan of the optical thickness along each ray's packed samples
// plus per-ray sums.  One 64-lane wavefront owns one ray (pack): it walks the ray in 64-sample
// chunks, does the in-chunk prefix sum with DPP/ds_swizzle-lowered wave shuffles (6 steps) and
// carries the running total in a scalar, so there are no atomics and the summation order is fixed
// (bitwise reproducible, unlike the atomicAdd-based sum_reduce this replaces).  Wide per-sample
// features (semantic / instance probabilities, up to 224 channels) are reduced with the CHANNEL on
// the lane and the samples walked sequentially: every load is a fully coalesced row read.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------------------- ray march
// wisp OctreeAS.raymarch 'ray' mode as restated in oracle/render.py raymarch_ray():
//   depth = (t_s + jitter/S)^2 * (far - near) + near ;  sample = fma(d, depth, o) (torch.addcmul) ;
//   delta_s = depth_s - depth_{s-1} (delta_0 = depth_0 - near) ;
//   keep iff inside [-1,1]^3 and the occupancy bit of its 2^level cell is set.
struct MarchArgs {
    const float *origins, *dirs, *tvals, *jitter;
    const uint32_t *occ;
    int64_t N;
    int S, level;
    float dmin, dmax;
    int64_t *ridx64 = nullptr;      // optional second copy of the ray ids as int64 ('ray' mode pack pass)
};

__device__ __forceinline__ float march_depth(const MarchArgs &a, int64_t ray, int s) {
    float d = __fadd_rn(a.tvals[s], __fdiv_rn(a.jitter[ray * a.S + s], (float)a.S));
    d = __fmul_rn(d, d);
    d = __fmul_rn(d, __fsub_rn(a.dmax, a.dmin));
    return __fadd_rn(d, a.dmin);
}

__device__ __forceinline__ bool march_keep(const MarchArgs &a, const float (&p)[3], int32_t &cell) {
    const int R = 1 << a.level;
    const float half = (float)(R / 2) + (R == 1 ? 0.5f : 0.0f);
    bool inside = true;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        inside = inside && (p[k] >= -1.0f) && (p[k] <= 1.0f);
        int v = (int)floorf(__fmul_rn(__fadd_rn(p[k], 1.0f), half));
        c[k] = min(max(v, 0), R - 1);
    }
    cell = (c[0] * R + c[1]) * R + c[2];
    if (!inside) return false;
    if (a.occ == nullptr) return true;
    return (a.occ[cell >> 5] >> (cell & 31)) & 1u;
}

// one wave per ray; PACK = false: count only
template <bool PACK>
__global__ __launch_bounds__(256) void march_kernel(MarchArgs a, int32_t *counts, const int64_t *offsets, int32_t *ridx,
                                                    int32_t *pidx, float *samples, float *depths, float *deltas,
                                                    uint8_t *boundary) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= a.N) return;
    const float o[3] = {a.origins[ray * 3], a.origins[ray * 3 + 1], a.origins[ray * 3 + 2]};
    const float dr[3] = {a.dirs[ray * 3], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2]};
    int64_t base = PACK ? offsets[ray] : 0;
    int total = 0;
    for (int s0 = 0; s0 < a.S; s0 += 64) {
        const int s = s0 + lane;
        bool keep = false;
        float depth = 0.0f, prev = a.dmin, p[3] = {0.0f, 0.0f, 0.0f};
        int32_t cell = 0;
        if (s < a.S) {
            depth = march_depth(a, ray, s);
            if (PACK && s > 0) prev = march_depth(a, ray, s - 1);
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = __fmaf_rn(dr[k], depth, o[k]);   // torch.addcmul is a fused multiply-add
            keep = march_keep(a, p, cell);
        }
        const unsigned long long mask = __ballot(keep);
        if (PACK && keep) {
            const int64_t pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            ridx[pos] = (int32_t)ray;
            if (a.ridx64) a.ridx64[pos] = ray;
            pidx[pos] = cell;
            samples[pos * 3 + 0] = p[0];
            samples[pos * 3 + 1] = p[1];
            samples[pos * 3 + 2] = p[2];
            depths[pos] = depth;
            deltas[pos] = __fsub_rn(depth, prev);
            boundary[pos] = (total == 0 && (mask & ((1ull << lane) - 1ull)) == 0) ? 1 : 0;
        }
        const int n = __popcll(mask);
        base += n;
        total += n;
    }
    if (!PACK && lane == 0) counts[ray] = total;
}

// ------------------------------------------------------------------------------ voxel-mode ray march
// 3-D DDA through the 2^level occupancy grid (oracle/render.py raymarch_voxel(), same fp32 op order): every occupied
// cell a ray crosses is a nugget [t_in, t_out] that receives k samples at t_in + (t_out - t_in) (i + 0.5) / k.
// One lane per ray (rays are few - 4k..25k - and each walks <= 3R cells); PACK = false counts only.
//   max_travel (finite): the tracer's travel filter (tracers/panoptic_packed_rf_tracer.py:88-108) applied inside the walk - a nugget
//     is kept iff  depth of its first sample - depth of the ray's first sample < max_travel  (strict, fp32 subtraction as the
//     tensor expression).  Depths grow along the ray, so the first nugget that fails ends the walk.  INFINITY: no filter.
//   counts are in SAMPLES (nuggets * k): their exclusive scan (pag_pack_offsets) is the compositing kernels' pack table.
// The walk reads one occupancy word per step (a dependent L2 access, ~250 ns): a workgroup first ORs the bitfield down to a
// (R/4)^3 coarse grid in LDS (R >= 32) and only touches the fine word where the coarse cell has anything in it.
template <bool PACK>
__global__ __launch_bounds__(256) void voxel_march_kernel(MarchArgs a, int k, float max_travel, const uint32_t *coarse, int32_t *counts,
                                                          const int64_t *offsets, int32_t *ridx, int32_t *pidx, float *samples,
                                                          float *depths, float *deltas, uint8_t *boundary, int32_t *ridx_sample,
                                                          float2 *nug_t = nullptr, int32_t *nug_cell = nullptr) {
    extern __shared__ uint32_t coarse_lds[];
    const int R = 1 << a.level;
    const int RC = R >> 2;                                  // coarse cells per axis
    const bool use_coarse = coarse != nullptr;
    if (use_coarse) {
        const int words = (RC * RC * RC + 31) >> 5;
        for (int w = threadIdx.x; w < words; w += blockDim.x) coarse_lds[w] = coarse[w];
        __syncthreads();
    }
    const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= a.N) return;
    const float cs = __fdiv_rn(2.0f, (float)R);
    const float o[3] = {a.origins[ray * 3], a.origins[ray * 3 + 1], a.origins[ray * 3 + 2]};
    const float d[3] = {a.dirs[ray * 3], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2]};
    float t0 = a.dmin, t1 = a.dmax;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (d[ax] != 0.0f) {
            const float ta = __fdiv_rn(__fsub_rn(-1.0f, o[ax]), d[ax]), tb = __fdiv_rn(__fsub_rn(1.0f, o[ax]), d[ax]);
            const float lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
            t0 = t0 >= lo ? t0 : lo;
            t1 = t1 <= hi ? t1 : hi;
        } else if (o[ax] < -1.0f || o[ax] > 1.0f) {
            t1 = -1.0f;
        }
    }
    int n = 0;
    const int64_t base = PACK ? offsets[ray] / k : 0;       // first nugget of this ray
    if (t0 < t1) {
        const float tm = __fadd_rn(t0, __fmul_rn(__fsub_rn(t1, t0), 1e-6f));
        int c0, c1, c2, s0 = 0, s1 = 0, s2 = 0;
        float n0 = INFINITY, n1 = INFINITY, n2 = INFINITY, e0 = INFINITY, e1 = INFINITY, e2 = INFINITY;
        auto setup = [&](int ax, int &c, int &st, float &tn, float &td) {
            const float pa = __fadd_rn(__fmul_rn(d[ax], tm), o[ax]);
            int ci = (int)floorf(__fdiv_rn(__fadd_rn(pa, 1.0f), cs));
            c = min(max(ci, 0), R - 1);
            if (d[ax] > 0.0f) {
                st = 1;
                tn = __fdiv_rn(__fsub_rn(__fadd_rn(-1.0f, __fmul_rn((float)(c + 1), cs)), o[ax]), d[ax]);
                td = __fdiv_rn(cs, d[ax]);
            } else if (d[ax] < 0.0f) {
                st = -1;
                tn = __fdiv_rn(__fsub_rn(__fadd_rn(-1.0f, __fmul_rn((float)c, cs)), o[ax]), d[ax]);
                td = __fdiv_rn(cs, -d[ax]);
            }
        };
        setup(0, c0, s0, n0, e0);
        setup(1, c1, s1, n1, e1);
        setup(2, c2, s2, n2, e2);
        float t = t0;
        float first = 0.0f;                                  // depth of the ray's first sample (first kept nugget)
        const float fr0 = __fdiv_rn(0.5f, (float)k);
        for (int it = 0; it < 3 * R + 3; ++it) {
            const int ax = (n0 <= n1 && n0 <= n2) ? 0 : (n1 <= n2 ? 1 : 2);
            const float tn = ax == 0 ? n0 : (ax == 1 ? n1 : n2);
            const float tout = tn <= t1 ? tn : t1;
            const int lin = (c0 * R + c1) * R + c2;
            bool occ = tout > t;
            if (occ && use_coarse) {
                const int cl = ((c0 >> 2) * RC + (c1 >> 2)) * RC + (c2 >> 2);
                occ = (coarse_lds[cl >> 5] >> (cl & 31)) & 1u;
            }
            if (occ && a.occ) occ = (a.occ[lin >> 5] >> (lin & 31)) & 1u;
            if (occ) {
                const float span = __fsub_rn(tout, t);
                const float dep0 = __fadd_rn(t, __fmul_rn(span, fr0));
                if (n == 0) first = dep0;
                if (!(__fsub_rn(dep0, first) < max_travel)) break;          // travel filter: later nuggets are farther still
                if (!PACK && nug_t) {       // the walk is done ONCE: nugget n of every ray is kept ([n][ray]) for voxel_pack_nuggets_kernel
                    nug_t[(int64_t)n * a.N + ray] = float2{t, tout};
                    nug_cell[(int64_t)n * a.N + ray] = lin;
                }
                if (PACK) {
                    const int64_t g = base + n;
                    ridx[g] = (int32_t)ray;
                    if (a.ridx64) a.ridx64[g] = ray;
                    pidx[g] = lin;
                    const float dl = __fdiv_rn(span, (float)k);
                    for (int i = 0; i < k; ++i) {
                        const float fr = __fdiv_rn((float)i + 0.5f, (float)k);
                        const float dep = __fadd_rn(t, __fmul_rn(span, fr));
                        const int64_t q = g * k + i;
                        depths[q] = dep;
                        deltas[q] = dl;
                        boundary[q] = (n == 0 && i == 0) ? 1 : 0;
                        if (ridx_sample) ridx_sample[q] = (int32_t)ray;
#pragma unroll
                        for (int x = 0; x < 3; ++x) samples[q * 3 + x] = __fmaf_rn(d[x], dep, o[x]);
                    }
                }
                ++n;
            }
            if (tout >= t1) break;
            t = tout;
            if (ax == 0) { c0 += s0; n0 = __fadd_rn(n0, e0); if (c0 < 0 || c0 >= R) break; }
            else if (ax == 1) { c1 += s1; n1 = __fadd_rn(n1, e1); if (c1 < 0 || c1 >= R) break; }
            else { c2 += s2; n2 = __fadd_rn(n2, e2); if (c2 < 0 || c2 >= R) break; }
        }
    }
    if (!PACK) counts[ray] = n * k;
}

// ---- the walk in two phases (pag_raymarch_voxel_count_nuggets).  voxel_march_kernel's loop carries two chains: the DDA state (a few fp32
// adds and compares per step) and, behind every cell the coarse grid calls occupied, a dependent L2 read of the fine occupancy word - and a
// wave of 64 rays pays that round trip in every step in which ANY of its rays needs it: 0.10 ms for 4096 rays on 64 waves, most of it waiting.
// The occupancy does not steer the walk (it only decides what is emitted), so:
//   phase 1 (voxel_walk_kernel)    one lane per ray, arithmetic only: every step with t_out > t_in is recorded as a candidate (t_in, t_out, cell)
//                                  at [step][ray] of the nugget arrays (capacity 3R + 3 = the walk's iteration bound), candidates per ray -> counts;
//   phase 2 (voxel_select_kernel)  one wave per ray, a lane per candidate: occupancy bit (an independent gather), the travel filter against the
//                                  first kept candidate's depth, ordered in-place compaction (ballot prefix) -> the ray's nuggets and counts.
// Same fp32 expressions on the same operands in the same order as voxel_march_kernel: identical nuggets (tests/test_gpu_edges.py, test_gpu_parity.py
// compare both forms and the oracle).  The travel filter is monotone (depths grow along the ray), so "the first nugget that fails ends the walk"
// and "drop every candidate that fails" select the same set.
__global__ __launch_bounds__(64) void voxel_walk_kernel(MarchArgs a, int32_t *__restrict__ counts, float2 *__restrict__ cand_t,
                                                        int32_t *__restrict__ cand_cell) {
    const int R = 1 << a.level;
    const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= a.N) return;
    const float cs = __fdiv_rn(2.0f, (float)R);
    const float o[3] = {a.origins[ray * 3], a.origins[ray * 3 + 1], a.origins[ray * 3 + 2]};
    const float d[3] = {a.dirs[ray * 3], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2]};
    float t0 = a.dmin, t1 = a.dmax;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (d[ax] != 0.0f) {
            const float ta = __fdiv_rn(__fsub_rn(-1.0f, o[ax]), d[ax]), tb = __fdiv_rn(__fsub_rn(1.0f, o[ax]), d[ax]);
            const float lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
            t0 = t0 >= lo ? t0 : lo;
            t1 = t1 <= hi ? t1 : hi;
        } else if (o[ax] < -1.0f || o[ax] > 1.0f) {
            t1 = -1.0f;
        }
    }
    int n = 0;
    if (t0 < t1) {
        const float tm = __fadd_rn(t0, __fmul_rn(__fsub_rn(t1, t0), 1e-6f));
        int c0, c1, c2, s0 = 0, s1 = 0, s2 = 0;
        float n0 = INFINITY, n1 = INFINITY, n2 = INFINITY, e0 = INFINITY, e1 = INFINITY, e2 = INFINITY;
        auto setup = [&](int ax, int &c, int &st, float &tn, float &td) {
            const float pa = __fadd_rn(__fmul_rn(d[ax], tm), o[ax]);
            int ci = (int)floorf(__fdiv_rn(__fadd_rn(pa, 1.0f), cs));
            c = min(max(ci, 0), R - 1);
            if (d[ax] > 0.0f) {
                st = 1;
                tn = __fdiv_rn(__fsub_rn(__fadd_rn(-1.0f, __fmul_rn((float)(c + 1), cs)), o[ax]), d[ax]);
                td = __fdiv_rn(cs, d[ax]);
            } else if (d[ax] < 0.0f) {
                st = -1;
                tn = __fdiv_rn(__fsub_rn(__fadd_rn(-1.0f, __fmul_rn((float)c, cs)), o[ax]), d[ax]);
                td = __fdiv_rn(cs, -d[ax]);
            }
        };
        setup(0, c0, s0, n0, e0);
        setup(1, c1, s1, n1, e1);
        setup(2, c2, s2, n2, e2);
        // straight-line step (selects instead of the three-way branch of voxel_march_kernel: a divergent branch costs the wave both sides and
        // a dozen exec-mask instructions on the loop-carried path); a lane that has left the walk stays in the loop inactive
        float t = t0;
        bool active = true;
        int64_t off = ray;                    // [n][ray]
        for (int it = 0; it < 3 * R + 3; ++it) {
            const bool a0 = n0 <= n1 && n0 <= n2, a1 = !a0 && n1 <= n2, a2 = !a0 && !a1;
            const float tn = a0 ? n0 : (a1 ? n1 : n2);
            const float tout = tn <= t1 ? tn : t1;
            const bool rec = active && tout > t;
            if (rec) {
                cand_t[off] = float2{t, tout};
                cand_cell[off] = (c0 * R + c1) * R + c2;
            }
            off += rec ? a.N : 0;
            n += rec ? 1 : 0;
            const bool last = tout >= t1;
            t = tout;
            c0 += a0 ? s0 : 0;
            c1 += a1 ? s1 : 0;
            c2 += a2 ? s2 : 0;
            const float m0 = __fadd_rn(n0, e0), m1 = __fadd_rn(n1, e1), m2 = __fadd_rn(n2, e2);
            n0 = a0 ? m0 : n0;
            n1 = a1 ? m1 : n1;
            n2 = a2 ? m2 : n2;
            const int cm = a0 ? c0 : (a1 ? c1 : c2);
            active = active && !last && (unsigned)cm < (unsigned)R;
            if (__ballot(active) == 0ull) break;
        }
    }
    counts[ray] = n;
}

// Memory layout: the walk (a lane per ray) writes candidate n of ray r at [n][r] - consecutive lanes, consecutive addresses.  This kernel works a lane
// per CANDIDATE: read straight from [n][r] its lanes would be N x 8 bytes apart (64 separate lines per load: 65 us for 24 576 rays), so a workgroup takes
// 64 rays and moves 64 candidates of each through an LDS tile - loaded along the rays, read along the candidates.  The kept nuggets leave in RAY-MAJOR
// order ([r][slot], `cap` slots per ray): consecutive kept lanes write consecutive addresses, and the pack kernel (a lane per nugget) reads them the same way.
constexpr int SEL_RAYS = 64;
__global__ __launch_bounds__(256) void voxel_select_kernel(const uint32_t *__restrict__ occ, int64_t N, int k, float max_travel, int32_t *counts,
                                                           const float2 *__restrict__ cand_t, const int32_t *__restrict__ cand_cell, int64_t cap,
                                                           float2 *__restrict__ sel_t, int32_t *__restrict__ sel_cell) {
    __shared__ float2 tile_t[64][SEL_RAYS + 1];
    __shared__ int32_t tile_c[64][SEL_RAYS + 1];
    __shared__ int32_t s_cnt[SEL_RAYS], s_n[SEL_RAYS], s_have[SEL_RAYS];
    __shared__ float s_first[SEL_RAYS];
    __shared__ int32_t s_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * SEL_RAYS;
    if (tid == 0) s_max = 0;
    __syncthreads();
    if (tid < SEL_RAYS) {
        const int32_t c = (r0 + tid < N) ? counts[r0 + tid] : 0;
        s_cnt[tid] = c;
        s_n[tid] = 0;
        s_have[tid] = 0;
        s_first[tid] = 0.0f;
        int32_t m = c;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
        if (lane == 0) atomicMax(&s_max, m);
    }
    __syncthreads();
    const int ncand_max = s_max;
    const float fr0 = __fdiv_rn(0.5f, (float)k);
    for (int j0 = 0; j0 < ncand_max; j0 += 64) {
        // tile[jj][r] <- candidate j0 + jj of ray r0 + r: thread (wave, lane = r) loads rows jj = wave, wave + 4, ... (256-byte / 512-byte coalesced rows)
        const bool ray_ok = r0 + lane < N;
        const int cnt_l = s_cnt[lane];
#pragma unroll 4
        for (int jj = wave; jj < 64; jj += 4) {
            const int j = j0 + jj;
            if (ray_ok && j < cnt_l) {
                tile_t[jj][lane] = cand_t[(int64_t)j * N + r0 + lane];
                tile_c[jj][lane] = cand_cell[(int64_t)j * N + r0 + lane];
            }
        }
        __syncthreads();
        for (int rr = 0; rr < SEL_RAYS / 4; ++rr) {          // wave w owns rays w * 16 .. w * 16 + 15 of the group
            const int rl = wave * (SEL_RAYS / 4) + rr;
            const int64_t ray = r0 + rl;
            const int ncand = s_cnt[rl];
            if (ray >= N || j0 >= ncand) continue;            // wave-uniform
            const int j = j0 + lane;
            float2 tt = float2{0.0f, 0.0f};
            int32_t cell = 0;
            bool keep = false;
            if (j < ncand) {
                tt = tile_t[lane][rl];
                cell = tile_c[lane][rl];
                keep = occ ? ((occ[cell >> 5] >> (cell & 31)) & 1u) : true;
            }
            const float dep0 = __fadd_rn(tt.x, __fmul_rn(__fsub_rn(tt.y, tt.x), fr0));
            unsigned long long m = __ballot(keep);
            float first = s_first[rl];
            if (!s_have[rl] && m) {             // depth of the ray's first sample = first kept candidate (wave-uniform)
                first = __shfl(dep0, __ffsll((long long)m) - 1);
                if (lane == 0) {
                    s_first[rl] = first;
                    s_have[rl] = 1;
                }
            }
            keep = keep && (__fsub_rn(dep0, first) < max_travel);
            m = __ballot(keep);
            const int n = s_n[rl];
            if (keep) {
                const int64_t dst = ray * cap + n + __popcll(m & ((1ull << lane) - 1ull));
                sel_t[dst] = tt;
                sel_cell[dst] = cell;
            }
            if (lane == 0) s_n[rl] = n + __popcll(m);
        }
        __syncthreads();      // the tile is overwritten by the next chunk
    }
    if (tid < SEL_RAYS && r0 + tid < N) counts[r0 + tid] = s_n[tid] * k;
}

// The packed outputs of a ray from the nuggets its (single) walk recorded: one wave per ray, a lane per nugget - the second sequential walk of
// voxel_march_kernel<true> (0.12 ms for 4096 rays: 64 waves on the whole chip, every step a dependent occupancy read) becomes a parallel
// copy.  Same fp32 expressions on the same (t_in, t_out) -> the same bits.
__global__ __launch_bounds__(256) void voxel_pack_nuggets_kernel(const float *__restrict__ origins, const float *__restrict__ dirs, int64_t N, int k,
                                                                 const int64_t *__restrict__ offsets, const float2 *__restrict__ nug_t,
                                                                 const int32_t *__restrict__ nug_cell, int64_t cap, int32_t *ridx, int32_t *pidx, float *samples,
                                                                 float *depths, float *deltas, uint8_t *boundary, int32_t *ridx_sample, int64_t *ridx64) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;
    const int64_t base = offsets[ray] / k;
    const int cnt = (int)((offsets[ray + 1] - offsets[ray]) / k);
    const float o[3] = {origins[ray * 3], origins[ray * 3 + 1], origins[ray * 3 + 2]};
    const float d[3] = {dirs[ray * 3], dirs[ray * 3 + 1], dirs[ray * 3 + 2]};
    for (int n = lane; n < cnt; n += 64) {
        const float2 tt = nug_t[ray * cap + n];               // ray-major, as voxel_select_kernel leaves them
        const float t = tt.x, span = __fsub_rn(tt.y, tt.x);
        const int64_t g = base + n;
        ridx[g] = (int32_t)ray;
        if (ridx64) ridx64[g] = ray;
        pidx[g] = nug_cell[ray * cap + n];
        const float dl = __fdiv_rn(span, (float)k);
        for (int i = 0; i < k; ++i) {
            const float fr = __fdiv_rn((float)i + 0.5f, (float)k);
            const float dep = __fadd_rn(t, __fmul_rn(span, fr));
            const int64_t q = g * k + i;
            depths[q] = dep;
            deltas[q] = dl;
            boundary[q] = (n == 0 && i == 0) ? 1 : 0;
            if (ridx_sample) ridx_sample[q] = (int32_t)ray;
#pragma unroll
            for (int x = 0; x < 3; ++x) samples[q * 3 + x] = __fmaf_rn(d[x], dep, o[x]);
        }
    }
}

// OR the 2^level occupancy bitfield down by 4 per axis: coarse bit ((x/4)*RC + y/4)*RC + z/4 is set iff any of its 64 fine cells
// is.  One lane per coarse cell (RC^3 <= 32768 at level 7): 16 fine words of 32 z-bits each hold its 4x4 columns' nibbles.
__global__ __launch_bounds__(256) void occupancy_coarse_kernel(const uint32_t *__restrict__ bits, int level, uint32_t *__restrict__ coarse) {
    const int R = 1 << level, RC = R >> 2;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int total = RC * RC * RC;
    bool any = false;
    if (cell < total) {
        const int C2 = cell % RC, C1 = (cell / RC) % RC, C0 = cell / (RC * RC);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lin = ((C0 * 4 + i) * R + (C1 * 4 + j)) * R + C2 * 4;       // 4 consecutive z cells: one nibble of one word
                any = any || ((bits[lin >> 5] >> (lin & 31)) & 0xFu);
            }
    }
    const unsigned long long m = __ballot(any);
    const int lane = threadIdx.x & 63;
    const int word = (cell - lane) >> 5;
    const int words = (total + 31) >> 5;
    if (lane == 0 && word < words) coarse[word] = (uint32_t)m;
    if (lane == 32 && word + 1 < words) coarse[word + 1] = (uint32_t)(m >> 32);
}

// ------------------------------------------------------------------------------------- compositing
struct CompArgs {
    const int64_t *pack_start;
    const int32_t *ray_of_pack;
    int64_t P;
    const float *sigma, *deltas, *depths, *rgb;
    int bg;
    int64_t n_samples;      // > 0: the per-sample outputs hold n_samples elements and [pack_start[P], n_samples) - filler samples of a padded
                            // batch (pag_pad_packed) that belong to no pack - is zeroed by the extra workgroups of the launch
};
constexpr int COMP_TAIL_BLOCKS = 8;
// workgroups past the packs' zero the tail of the per-sample outputs (instead of a full-size fill launch per tensor before the call)
__device__ __forceinline__ void comp_zero_tail(const CompArgs &a, int64_t block, float *w0, float *w1_3) {
    const int64_t beg = a.pack_start[a.P];
    for (int64_t i = beg + block * 256 + threadIdx.x; i < a.n_samples; i += (int64_t)COMP_TAIL_BLOCKS * 256) {
        if (w0) w0[i] = 0.0f;
        if (w1_3) w1_3[3 * i] = 0.0f, w1_3[3 * i + 1] = 0.0f, w1_3[3 * i + 2] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs a, float *weights, float *out_alpha, float *out_rgb,
                                                            float *out_depth, uint8_t *out_hit) {
    const int lane = threadIdx.x & 63;
    const int64_t pack_blocks = (a.P + 3) / 4;
    if ((int64_t)blockIdx.x >= pack_blocks) {
        comp_zero_tail(a, (int64_t)blockIdx.x - pack_blocks, weights, nullptr);
        return;
    }
    const int64_t pk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pk >= a.P) return;
    const int64_t beg = a.pack_start[pk], end = a.pack_start[pk + 1];
    const int64_t ray = a.ray_of_pack[pk];
    float carry = 0.0f;                 // sum of tau before this chunk
    float s_w = 0.0f, s_r = 0.0f, s_g = 0.0f, s_b = 0.0f, s_d = 0.0f;
    for (int64_t i0 = beg; i0 < end; i0 += 64) {
        const int64_t i = i0 + lane;
        const bool live = i < end;
        const float tau = live ? a.sigma[i] * a.deltas[i] : 0.0f;
        const float incl = wave_incl_scan(tau, lane);
        const float excl = carry + (incl - tau);
        const float w = live ? expf(-excl) * (1.0f - expf(-tau)) : 0.0f;
        if (live) weights[i] = w;
        s_w += w;
        if (a.rgb && live) {
            s_r += w * a.rgb[i * 3 + 0];
            s_g += w * a.rgb[i * 3 + 1];
            s_b += w * a.rgb[i * 3 + 2];
        }
        if (a.depths && live) s_d += w * a.depths[i];
        carry += __shfl(incl, 63);
    }
    const float alpha = wave_sum(s_w);
    s_r = wave_sum(s_r);
    s_g = wave_sum(s_g);
    s_b = wave_sum(s_b);
    s_d = wave_sum(s_d);
    if (lane == 0) {
        out_alpha[ray] = alpha;
        if (out_hit) out_hit[ray] = alpha > 0.0f ? 1 : 0;
        if (a.rgb) {
            const float bgv = a.bg == PAG_BG_WHITE ? (1.0f - alpha) : 0.0f;   // tracer :163-168
            out_rgb[ray * 3 + 0] = bgv + alpha * s_r;
            out_rgb[ray * 3 + 1] = bgv + alpha * s_g;
            out_rgb[ray * 3 + 2] = bgv + alpha * s_b;
        }
        if (a.depths) out_depth[ray] = s_d;
    }
}

// Backward.  Per ray: C = sum w c, D = sum w d, alpha = sum w;
//   rgb_out = bg(1-alpha) + alpha*C ;  depth_out = D ; alpha_out = alpha
// upstream -> gC = g_rgb*alpha ; gA = sum_ch g_rgb_ch*(C_ch - bg) + g_alpha ; gD = g_depth
//   gw_i   = gA + gC.c_i + gD*d_i ;   d c_i = gC * w_i
//   dtau_i = gw_i * (T_i - w_i) - sum_{k>i} gw_k w_k ,  T_i = exp(-sum_{j<i} tau_j)
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs a, const float *weights, const float *out_alpha,
                                                            const float *g_rgb, const float *g_depth, const float *g_alpha,
                                                            float *d_sigma, float *d_rgb) {
    const int lane = threadIdx.x & 63;
    const int64_t pack_blocks = (a.P + 3) / 4;
    if ((int64_t)blockIdx.x >= pack_blocks) {
        comp_zero_tail(a, (int64_t)blockIdx.x - pack_blocks, d_sigma, d_rgb);
        return;
    }
    const int64_t pk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pk >= a.P) return;
    const int64_t beg = a.pack_start[pk], end = a.pack_start[pk + 1];
    const int64_t ray = a.ray_of_pack[pk];
    const float alpha = out_alpha[ray];
    const bool has_rgb = a.rgb && g_rgb;
    // pass 1: C (needed for gA)
    float c_r = 0.0f, c_g = 0.0f, c_b = 0.0f;
    if (has_rgb) {
        for (int64_t i = beg + lane; i < end; i += 64) {
            const float w = weights[i];
            c_r += w * a.rgb[i * 3 + 0];
            c_g += w * a.rgb[i * 3 + 1];
            c_b += w * a.rgb[i * 3 + 2];
        }
        c_r = wave_sum(c_r);
        c_g = wave_sum(c_g);
        c_b = wave_sum(c_b);
    }
    float gC[3] = {0.0f, 0.0f, 0.0f};
    float gA = g_alpha ? g_alpha[ray] : 0.0f;
    if (has_rgb) {
        const float bgv = a.bg == PAG_BG_WHITE ? 1.0f : 0.0f;
        const float gr = g_rgb[ray * 3 + 0], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
        gC[0] = gr * alpha;
        gC[1] = gg * alpha;
        gC[2] = gb * alpha;
        gA += gr * (c_r - bgv) + gg * (c_g - bgv) + gb * (c_b - bgv);
    }
    const float gD = (a.depths && g_depth) ? g_depth[ray] : 0.0f;
    // pass 2: total of gw*w
    float tot = 0.0f;
    for (int64_t i = beg + lane; i < end; i += 64) {
        const float w = weights[i];
        float gw = gA;
        if (has_rgb) gw += gC[0] * a.rgb[i * 3 + 0] + gC[1] * a.rgb[i * 3 + 1] + gC[2] * a.rgb[i * 3 + 2];
        if (gD != 0.0f) gw += gD * a.depths[i];
        tot += gw * w;
    }
    tot = wave_sum(tot);
    // pass 3: per-sample gradients
    float carry_tau = 0.0f, carry_gww = 0.0f;
    for (int64_t i0 = beg; i0 < end; i0 += 64) {
        const int64_t i = i0 + lane;
        const bool live = i < end;
        const float sg = live ? a.sigma[i] : 0.0f;
        const float dl = live ? a.deltas[i] : 0.0f;
        const float tau = sg * dl;
        const float w = live ? weights[i] : 0.0f;
        float gw = gA;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        if (has_rgb && live) {
            c0 = a.rgb[i * 3 + 0];
            c1 = a.rgb[i * 3 + 1];
            c2 = a.rgb[i * 3 + 2];
            gw += gC[0] * c0 + gC[1] * c1 + gC[2] * c2;
        }
        if (gD != 0.0f && live) gw += gD * a.depths[i];
        const float gww = live ? gw * w : 0.0f;
        const float incl_tau = wave_incl_scan(tau, lane);
        const float incl_gww = wave_incl_scan(gww, lane);
        const float T = expf(-(carry_tau + (incl_tau - tau)));
        const float suffix = tot - (carry_gww + incl_gww);          // sum_{k>i} gw_k w_k
        if (live) {
            const float dtau = gw * (T - w) - suffix;
            d_sigma[i] = dtau * dl;
            if (d_rgb) {
                d_rgb[i * 3 + 0] = gC[0] * w;
                d_rgb[i * 3 + 1] = gC[1] * w;
                d_rgb[i * 3 + 2] = gC[2] * w;
            }
        }
        carry_tau += __shfl(incl_tau, 63);
        carry_gww += __shfl(incl_gww, 63);
    }
}

// out[ray, c] = alpha[ray] * sum_i w_i f[i, c]; one workgroup (4 waves) per ray, the ray's samples split in 4 contiguous
// quarters, partials combined through LDS in wave order.  VEC: lane l owns channels 4l..4l+3 (one 8/16-byte load per
// lane and sample row - a 2-byte-per-lane load per channel group kept the texture-address unit busy instead of HBM).
template <typename FT, bool VEC>
__global__ __launch_bounds__(256) void composite_feats_fwd_kernel(const int64_t *pack_start, const int32_t *ray_of_pack,
                                                                  const float *weights, const float *alpha, const FT *feats, int C,
                                                                  float *out) {
    __shared__ float part[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pk = blockIdx.x;
    const int64_t beg = pack_start[pk], end = pack_start[pk + 1];
    const int64_t n = end - beg, q = (n + 3) / 4;
    const int64_t lo = beg + wave * q, hi = min(end, lo + q);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (VEC) {
        typedef FT vec4 __attribute__((ext_vector_type(4)));
        const bool on = 4 * lane < C;
        int64_t i = lo;
        for (; i + 4 <= hi; i += 4) {
            float wv[4];
            vec4 fv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                wv[u] = weights[i + u];
                if (on) fv[u] = *reinterpret_cast<const vec4 *>(feats + (i + u) * C + 4 * lane);
            }
            if (on) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] += wv[u] * (float)fv[u][k];
            }
        }
        for (; i < hi; ++i) {
            if (on) {
                const float w = weights[i];
                const vec4 f = *reinterpret_cast<const vec4 *>(feats + i * C + 4 * lane);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += w * (float)f[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) part[wave][(4 * lane + k) & 255] = acc[k];
    } else {
        for (int64_t i = lo; i < hi; ++i) {
            const float w = weights[i];
            const FT *row = feats + i * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                if (c < C) acc[k] += w * pag_ld(row + c);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) part[wave][lane + 64 * k] = acc[k];
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < C) {
        const int64_t ray = ray_of_pack[pk];
        const float s = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
        out[ray * C + c] = alpha[ray] * s;
    }
}

// The same sum for 64 bf16 channels (the per-ray sum of the colour decoder's dz_0 rows behind the view embedding's gradient, pose optimisation:
// 128 B per sample, 1.6 GB per dense 24 576-ray step).  The generic kernel above gives a lane 4 channels - 16 of 64 lanes busy on a 64-channel row,
// 2.4 TB/s.  Here a wave covers EIGHT rows per load instruction (lane = row l / 8, 16-byte piece l % 8: one full 1 KB request), four such groups
// in flight, fp32 partial sums per lane, the eight row-lanes combined by three shuffles at the end; one wave per ray quarter as before.
__global__ __launch_bounds__(256) void composite_feats64_bf16_fwd_kernel(const int64_t *pack_start, const int32_t *ray_of_pack, const float *weights,
                                                                         const float *alpha, const bf16_t *feats, float *out) {
    typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, piece = lane & 7;
    const int64_t pk = blockIdx.x;
    const int64_t beg = pack_start[pk], end = pack_start[pk + 1];
    const int64_t n = end - beg, q = (n + 3) / 4;
    const int64_t lo = beg + wave * q, hi = min(end, lo + q);
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t i0 = lo; i0 < hi; i0 += 32) {
        bf16x8 fv[4];
        float wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + 8 * u + sub;
            const bool ok = i < hi;
            const int64_t ic = ok ? i : lo;
            fv[u] = *reinterpret_cast<const bf16x8 *>(feats + ic * 64 + 8 * piece);
            wv[u] = ok ? weights[ic] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += wv[u] * (float)fv[u][k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = acc[k];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[k] = v;
    }
    if (sub == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) part[wave][8 * piece + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ray = ray_of_pack[pk];
        const float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        out[(int64_t)ray * 64 + threadIdx.x] = alpha[ray] * v;
    }
}


// narrow features (C <= 16, e.g. the semantic classes): one WAVE per ray with the SAMPLE on the lane - a workgroup per
// ray with the channel on the lane would leave 250 of 256 lanes idle and walk the ray one dependent load at a time
template <typename FT>
__global__ __launch_bounds__(256) void composite_feats_small_fwd_kernel(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P,
                                                                        const float *weights, const float *alpha, const FT *feats,
                                                                        int C, float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t pk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pk >= P) return;
    const int64_t beg = pack_start[pk], end = pack_start[pk + 1];
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    for (int64_t i = beg + lane; i < end; i += 64) {
        const float w = weights[i];
        const FT *row = feats + i * C;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) acc[c] += w * pag_ld(row + c);
    }
    const int64_t ray = ray_of_pack[pk];
    const float al = alpha[ray];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < C) {
            const float s = wave_sum(acc[c]);
            if (lane == 0) out[ray * C + c] = al * s;
        }
    }
}

template <typename FT>
__global__ __launch_bounds__(256) void composite_feats_bwd_kernel(const int64_t *pack_start, const int32_t *ray_of_pack,
                                                                  const float *weights, const float *alpha, const float *g_out,
                                                                  int C, FT *d_feats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pk = blockIdx.x;
    const int64_t beg = pack_start[pk], end = pack_start[pk + 1];
    const int64_t ray = ray_of_pack[pk];
    const float al = alpha[ray];
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;
        g[k] = c < C ? al * g_out[ray * C + c] : 0.0f;
    }
    for (int64_t i = beg + wave; i < end; i += 4) {
        const float w = weights[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + 64 * k;
            if (c < C) pag_st(d_feats + i * C + c, w * g[k]);
        }
    }
}

}  // namespace

static int march_check(const char *name, const float *origins, const float *dirs, int64_t N, int S, const float *tvals,
                       const float *jitter, int blas_level) {
    PAG_CHECK_ARG(N >= 0, "%s: N < 0", name);
    PAG_CHECK_ARG(S >= 1 && S <= 65536, "%s: S %d out of range", name, S);
    PAG_CHECK_ARG(blas_level >= 0 && blas_level <= 10, "%s: blas_level %d not in [0,10]", name, blas_level);
    PAG_CHECK_ARG(N == 0 || (origins && dirs && tvals && jitter), "%s: NULL input", name);
    return PAG_OK;
}

extern "C" int pag_raymarch_count(const float *origins, const float *dirs, int64_t N, int S, const float *tvals,
                                  const float *jitter, float dist_min, float dist_max, const uint32_t *occupancy_bits,
                                  int blas_level, int32_t *counts, void *stream) {
    int rc = march_check("pag_raymarch_count", origins, dirs, N, S, tvals, jitter, blas_level);
    if (rc) return rc;
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(counts, "pag_raymarch_count: counts is NULL");
    MarchArgs a{origins, dirs, tvals, jitter, occupancy_bits, N, S, blas_level, dist_min, dist_max};
    hipLaunchKernelGGL((march_kernel<false>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, counts,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    PAG_CHECK_LAUNCH("pag_raymarch_count");
    return PAG_OK;
}

namespace {
// Exclusive prefix sums of counts[0..N) for ONE workgroup of 16 waves, coalesced: wave w owns a contiguous segment (a multiple of 64 counts), sums it
// (pass 1: one 256-byte load per 64 rays), the sixteen totals meet in LDS, and pass 2 walks the segment again - from the L1 now - with a
// wave scan per 64 counts and a running carry.  emit(i, exclusive_prefix) is called for every i < N; returns the total.  Before: every THREAD
// summed a contiguous chunk of N / 1024 counts - lanes 96 bytes apart at 24 576 rays, 24 dependent uncoalesced loads per pass: 64 us of a 3 ms step.
template <typename Emit>
__device__ __forceinline__ int64_t block_exclusive_scan(const int32_t *__restrict__ counts, int64_t N, int64_t *wave_tot /* LDS [16] */, bool want_prefix, Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t seg = (((N + 15) / 16) + 63) / 64 * 64;
    const int64_t wlo = min(N, wave * seg), whi = min(N, wlo + seg);
    int64_t s = 0;
    for (int64_t i = wlo + lane; i < whi; i += 64) s += counts[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) wave_tot[wave] = s;
    __syncthreads();
    int64_t carry = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        carry += w < wave ? wave_tot[w] : 0;
        total += wave_tot[w];
    }
    if (!want_prefix) return total;
    for (int64_t i0 = wlo; i0 < whi; i0 += 64) {
        const int64_t i = i0 + lane;
        const int32_t c = i < whi ? counts[i] : 0;
        int32_t incl = c;              // a ray holds < 2^15 samples: 64 of them stay far below 2^31
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (i < whi) emit(i, carry + (int64_t)(incl - c));
        carry += (int64_t)__shfl(incl, 63);
    }
    return total;
}

// pack_start[i] = sum of counts[0..i), pack_start[N] = total: ONE workgroup, every thread owns a contiguous chunk (sum, block
// scan of the 1024 chunk sums, prefix write).  Replaces the cast / scan-init / scan / subtract / concatenate launches the same
// result costs as tensor ops at the head of every training step (N = 4096 rays: 4 counts per thread).
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const int32_t *__restrict__ counts, int64_t N, int64_t *__restrict__ pack_start,
                                                            int64_t *total_host) {
    __shared__ int64_t wave_tot[16];
    const int64_t total = block_exclusive_scan(counts, N, wave_tot, true, [&](int64_t i, int64_t ex) { pack_start[i] = ex; });
    if (threadIdx.x == 0) {
        pack_start[N] = total;
        // optional host-visible copy (pinned memory): the host polls it instead of a stream-synchronising read-back
        if (total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Pad a packed batch up to a fixed capacity (pagnerf_amd/graphs.py: HIP graphs need shapes that do not depend on device data).  The
// M = pack_start[N] real samples are followed by capacity - M filler samples that belong to NO pack (pack_start is left alone): the
// per-ray kernels never see them, the per-sample kernels (encoders, decoders) compute on them and their results are ignored; the caller
// zero-fills the per-sample gradient tensors the compositing backward produces, so the fillers carry exactly zero gradient.
// Coordinates (0,0,0) are inside the volume, ray index = the last ray.  One workgroup.  M > capacity: no filler is written (the caller
// sees the true count in its mailbox and falls back to exact shapes) - but the launches that were queued on the capacity-sized views
// before the host learnt M must stay inside them: pack_start_clamped[r] = min(pack_start[r], capacity) is the pack table THEY walk
// (identical to pack_start when the batch fits; a truncated batch, whose results the caller discards, when it does not).
// k = samples per nugget (voxel mode: the per-nugget arrays are padded up to capacity / k).
constexpr int PAD_WGS = 16;
struct PadArgs {
    int64_t capacity;
    int k;
    float *samples, *depths, *deltas;
    int32_t *ridx_sample, *ridx_nugget;
    int64_t *ridx64;
    int32_t *pidx;
    uint8_t *boundary;
};
// filler samples [M, capacity) of a padded batch (one workgroup; M > capacity: nothing to write)
__device__ __forceinline__ void pad_tail(const PadArgs &a, int64_t M, int64_t N) {
    if (M > a.capacity) return;
    const int32_t last = (int32_t)(N - 1);
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;      // every workgroup of the launch takes its share
    for (int64_t i = M + first; i < a.capacity; i += stride) {
        a.samples[i * 3] = 0.0f, a.samples[i * 3 + 1] = 0.0f, a.samples[i * 3 + 2] = 0.0f;
        a.depths[i] = 0.0f;
        a.deltas[i] = 0.0f;
        a.boundary[i] = 0;
        if (a.ridx_sample) a.ridx_sample[i] = last;
    }
    for (int64_t g = M / a.k + first; g < a.capacity / a.k; g += stride) {
        if (a.ridx_nugget) a.ridx_nugget[g] = last;
        if (a.ridx64) a.ridx64[g] = last;
        a.pidx[g] = 0;
    }
}
__global__ __launch_bounds__(1024) void pad_packed_kernel(const int64_t *__restrict__ pack_start, int64_t N, PadArgs a, int64_t *__restrict__ pack_start_clamped) {
    const int64_t M = pack_start[N];
    // the pack table the capacity-sized launches may walk: no pack reaches past `capacity`, whatever the march produced
    if (pack_start_clamped && blockIdx.x == 0)
        for (int64_t r = threadIdx.x; r <= N; r += blockDim.x) {
            const int64_t p = pack_start[r];
            pack_start_clamped[r] = p < a.capacity ? p : a.capacity;
        }
    pad_tail(a, M, N);
}

// pack_offsets_kernel + pad_packed_kernel + the copy of the ray directions into the graph's static buffer as ONE launch
// (pagnerf_amd/graphs.py: three 5 - 10 us launches at the head of every graph-replayed step; the fillers lie behind the samples the pack
// pass - queued after this kernel - writes, so the order does not matter).  PAD_WGS workgroups: every one repeats the (tiny) scan to learn
// the total, workgroup 0 writes the tables, and the filler stores - up to capacity - M ~ 10^4 samples x 25 B, which one workgroup took
// 10 us to write - and the direction copy are shared by all of them.
__global__ __launch_bounds__(1024) void pack_offsets_pad_kernel(const int32_t *__restrict__ counts, int64_t N, int64_t *__restrict__ pack_start, int64_t *total_host,
                                                                PadArgs a, int64_t *__restrict__ pack_start_clamped, const float *__restrict__ dirs_src,
                                                                float *__restrict__ dirs_dst) {
    __shared__ int64_t wave_tot[16];
    const int tid = threadIdx.x;
    const bool writer = blockIdx.x == 0;          // every workgroup repeats the (tiny) scan to learn the total; workgroup 0 writes the tables
    const int64_t total = block_exclusive_scan(counts, N, wave_tot, writer, [&](int64_t i, int64_t ex) {
        pack_start[i] = ex;
        pack_start_clamped[i] = ex < a.capacity ? ex : a.capacity;
    });
    if (writer && tid == 0) {
        pack_start[N] = total;
        pack_start_clamped[N] = total < a.capacity ? total : a.capacity;
        if (total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    pad_tail(a, total, N);
    if (dirs_dst)
        for (int64_t i = (int64_t)blockIdx.x * 1024 + tid; i < N * 3; i += (int64_t)gridDim.x * 1024) dirs_dst[i] = dirs_src[i];
}

// Several small device-to-device copies as ONE launch (pagnerf_amd/graphs.py: the copies of a replay's static outputs handed to the
// caller, and of the upstream gradients into the backward graph's static inputs - ten `hipMemcpyAsync`-style copies of 4 KB - 3 MB
// per step cost 4.7 us each as separate runtime copies).  blockIdx.y = copy, 16-byte pieces where both pointers allow, bytes otherwise.
constexpr int COPY_MAX = 16;
struct CopyBatch {
    void *dst[COPY_MAX];
    const void *src[COPY_MAX];
    int64_t nbytes[COPY_MAX];
};
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatch b) {
    const int c = blockIdx.y;
    const int64_t n = b.nbytes[c];
    unsigned char *d = reinterpret_cast<unsigned char *>(b.dst[c]);
    const unsigned char *s = reinterpret_cast<const unsigned char *>(b.src[c]);
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int64_t n16 = n >> 4;
        for (int64_t i = i0; i < n16; i += stride) reinterpret_cast<u32x4 *>(d)[i] = reinterpret_cast<const u32x4 *>(s)[i];
        for (int64_t i = (n16 << 4) + i0; i < n; i += stride) d[i] = s[i];
    } else {
        for (int64_t i = i0; i < n; i += stride) d[i] = s[i];
    }
}

// wisp PositionalEmbedder on the NEGATED ray directions (pc_nerf/panoptic_delta_nef.py:196-200): out[r] = (-d, sin(-d 2^k) k<F,
// cos(-d 2^k) k<F), frequency-major, zero padded to `width` columns.  One launch instead of the ten of the tensor-op form.
__global__ __launch_bounds__(256) void view_embed_kernel(const float *__restrict__ dirs, int64_t R, int n_freq, int width,
                                                         float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * width) return;
    const int64_t r = i / width;
    const int c = (int)(i - r * width);
    const int used = 3 + 6 * n_freq;
    float v = 0.0f;
    if (c < 3) {
        v = -dirs[r * 3 + c];
    } else if (c < used) {
        const int q = c - 3, is_cos = q >= 3 * n_freq;
        const int k = (is_cos ? q - 3 * n_freq : q) / 3, ax = (is_cos ? q - 3 * n_freq : q) % 3;
        const float w = -dirs[r * 3 + ax] * (float)(1 << k);          // exact scaling: same argument as x * 2.0^k
        v = is_cos ? cosf(w) : sinf(w);
    }
    out[i] = v;
}
}  // namespace

extern "C" int pag_pack_offsets(const int32_t *counts, int64_t N, int64_t *pack_start, int64_t *total_host, void *stream) {
    PAG_CHECK_ARG(N >= 0 && pack_start && (N == 0 || counts), "pag_pack_offsets: bad arguments");
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, N, pack_start, total_host);
    PAG_CHECK_LAUNCH("pag_pack_offsets");
    return PAG_OK;
}

extern "C" int pag_pad_packed(const int64_t *pack_start, int64_t N, int64_t capacity, int samples_per_entry, float *samples, float *depths, float *deltas,
                              int32_t *ridx_sample, int32_t *ridx_entry, int64_t *ridx64, int32_t *pidx, uint8_t *boundary, int64_t *pack_start_clamped,
                              void *stream) {
    PAG_CHECK_ARG(N >= 1 && capacity >= 0 && samples_per_entry >= 1 && capacity % samples_per_entry == 0,
                  "pag_pad_packed: N %lld, capacity %lld must be a multiple of samples_per_entry %d", (long long)N, (long long)capacity, samples_per_entry);
    PAG_CHECK_ARG(pack_start && samples && depths && deltas && pidx && boundary, "pag_pad_packed: NULL buffer");
    PadArgs a{capacity, samples_per_entry, samples, depths, deltas, ridx_sample, ridx_entry, ridx64, pidx, boundary};
    hipLaunchKernelGGL(pad_packed_kernel, dim3(PAD_WGS), dim3(1024), 0, (hipStream_t)stream, pack_start, N, a, pack_start_clamped);
    PAG_CHECK_LAUNCH("pag_pad_packed");
    return PAG_OK;
}

extern "C" int pag_pack_offsets_pad(const int32_t *counts, int64_t N, int64_t *pack_start, int64_t *total_host, int64_t capacity, int samples_per_entry,
                                    float *samples, float *depths, float *deltas, int32_t *ridx_sample, int32_t *ridx_entry, int64_t *ridx64, int32_t *pidx,
                                    uint8_t *boundary, int64_t *pack_start_clamped, const float *dirs_src, float *dirs_dst, void *stream) {
    PAG_CHECK_ARG(N >= 1 && counts && pack_start && pack_start_clamped, "pag_pack_offsets_pad: bad arguments");
    PAG_CHECK_ARG(capacity >= 0 && samples_per_entry >= 1 && capacity % samples_per_entry == 0,
                  "pag_pack_offsets_pad: capacity %lld must be a multiple of samples_per_entry %d", (long long)capacity, samples_per_entry);
    PAG_CHECK_ARG(samples && depths && deltas && pidx && boundary, "pag_pack_offsets_pad: NULL buffer");
    PAG_CHECK_ARG(!dirs_dst || dirs_src, "pag_pack_offsets_pad: dirs_dst without dirs_src");
    PadArgs a{capacity, samples_per_entry, samples, depths, deltas, ridx_sample, ridx_entry, ridx64, pidx, boundary};
    hipLaunchKernelGGL(pack_offsets_pad_kernel, dim3(PAD_WGS), dim3(1024), 0, (hipStream_t)stream, counts, N, pack_start, total_host, a, pack_start_clamped, dirs_src, dirs_dst);
    PAG_CHECK_LAUNCH("pag_pack_offsets_pad");
    return PAG_OK;
}

extern "C" int pag_copy_batch(int n, void *const *dst, const void *const *src, const int64_t *nbytes, void *stream) {
    PAG_CHECK_ARG(n >= 0 && n <= COPY_MAX, "pag_copy_batch: %d copies (at most %d per call)", n, COPY_MAX);
    if (n == 0) return PAG_OK;
    PAG_CHECK_ARG(dst && src && nbytes, "pag_copy_batch: NULL list");
    CopyBatch b{};
    int64_t longest = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        PAG_CHECK_ARG(nbytes[i] >= 0, "pag_copy_batch: copy %d has a negative size", i);
        if (nbytes[i] == 0) continue;
        PAG_CHECK_ARG(dst[i] && src[i], "pag_copy_batch: copy %d: NULL pointer", i);
        b.dst[m] = dst[i];
        b.src[m] = src[i];
        b.nbytes[m] = nbytes[i];
        longest = nbytes[i] > longest ? nbytes[i] : longest;
        ++m;
    }
    if (m == 0) return PAG_OK;
    const int64_t want = (longest / 16 + 255) / 256;
    const unsigned gx = (unsigned)(want < 1 ? 1 : (want > 256 ? 256 : want));
    hipLaunchKernelGGL(copy_batch_kernel, dim3(gx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, b);
    PAG_CHECK_LAUNCH("pag_copy_batch");
    return PAG_OK;
}

extern "C" int pag_view_embed(const float *dirs, int64_t R, int n_freq, int width, float *out, void *stream) {
    PAG_CHECK_ARG(R >= 0 && n_freq >= 0 && n_freq <= 16 && width >= 3 + 6 * n_freq, "pag_view_embed: n_freq %d / width %d out of range", n_freq, width);
    if (R == 0) return PAG_OK;
    PAG_CHECK_ARG(dirs && out, "pag_view_embed: NULL input/output");
    const int64_t items = R * width;
    hipLaunchKernelGGL(view_embed_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dirs, R, n_freq, width, out);
    PAG_CHECK_LAUNCH("pag_view_embed");
    return PAG_OK;
}

extern "C" int pag_raymarch_pack(const float *origins, const float *dirs, int64_t N, int S, const float *tvals,
                                 const float *jitter, float dist_min, float dist_max, const uint32_t *occupancy_bits,
                                 int blas_level, const int64_t *offsets, int32_t *ridx, int32_t *pidx, float *samples,
                                 float *depths, float *deltas, uint8_t *boundary, int64_t *ridx64, void *stream) {
    int rc = march_check("pag_raymarch_pack", origins, dirs, N, S, tvals, jitter, blas_level);
    if (rc) return rc;
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(offsets && ridx && pidx && samples && depths && deltas && boundary, "pag_raymarch_pack: NULL output");
    MarchArgs a{origins, dirs, tvals, jitter, occupancy_bits, N, S, blas_level, dist_min, dist_max, ridx64};
    hipLaunchKernelGGL((march_kernel<true>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, nullptr,
                       offsets, ridx, pidx, samples, depths, deltas, boundary);
    PAG_CHECK_LAUNCH("pag_raymarch_pack");
    return PAG_OK;
}

extern "C" int pag_composite_fwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const float *sigma,
                                 const float *deltas, const float *depths, const float *rgb, int bg_color, float *weights,
                                 float *out_alpha, float *out_rgb, float *out_depth, uint8_t *out_hit, int64_t n_samples, void *stream) {
    PAG_CHECK_ARG(P >= 0 && n_samples >= 0, "pag_composite_fwd: P < 0 or n_samples < 0");
    if (P == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && ray_of_pack && sigma && deltas && weights && out_alpha, "pag_composite_fwd: NULL input");
    PAG_CHECK_ARG(!rgb || out_rgb, "pag_composite_fwd: rgb given but out_rgb is NULL");
    PAG_CHECK_ARG(!depths || out_depth, "pag_composite_fwd: depths given but out_depth is NULL");
    PAG_CHECK_ARG(bg_color == PAG_BG_BLACK || bg_color == PAG_BG_WHITE, "pag_composite_fwd: bad bg_color %d", bg_color);
    CompArgs a{pack_start, ray_of_pack, P, sigma, deltas, depths, rgb, bg_color, n_samples};
    hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)((P + 3) / 4 + (n_samples > 0 ? COMP_TAIL_BLOCKS : 0))), dim3(256), 0, (hipStream_t)stream, a, weights,
                       out_alpha, out_rgb, out_depth, out_hit);
    PAG_CHECK_LAUNCH("pag_composite_fwd");
    return PAG_OK;
}

extern "C" int pag_composite_bwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const float *sigma,
                                 const float *deltas, const float *depths, const float *rgb, int bg_color, const float *weights,
                                 const float *out_alpha, const float *g_rgb, const float *g_depth, const float *g_alpha,
                                 float *d_sigma, float *d_rgb, int64_t n_samples, void *stream) {
    PAG_CHECK_ARG(P >= 0 && n_samples >= 0, "pag_composite_bwd: P < 0 or n_samples < 0");
    if (P == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && ray_of_pack && sigma && deltas && weights && out_alpha && d_sigma, "pag_composite_bwd: NULL input");
    PAG_CHECK_ARG(bg_color == PAG_BG_BLACK || bg_color == PAG_BG_WHITE, "pag_composite_bwd: bad bg_color %d", bg_color);
    CompArgs a{pack_start, ray_of_pack, P, sigma, deltas, depths, rgb, bg_color, n_samples};
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((P + 3) / 4 + (n_samples > 0 ? COMP_TAIL_BLOCKS : 0))), dim3(256), 0, (hipStream_t)stream, a, weights,
                       out_alpha, g_rgb, g_depth, g_alpha, d_sigma, d_rgb);
    PAG_CHECK_LAUNCH("pag_composite_bwd");
    return PAG_OK;
}

// d loss / d origins and d loss / d dirs of samples = origins[ray] + dirs[ray] * depth (pose optimisation, pc_nerf/ba_pipeline.py:85-92 through wisp's
// addcmul): per-ray sums of g and g * depth over the ray's pack.  One wave per pack, one pass over g [M,3] and depths [M] - the tensor-op form
// wrote a [M,6] concatenation, a weights-of-ones vector and ran the generic feature compositing on them.
__global__ __launch_bounds__(256) void ray_sample_grad_kernel(const int64_t *__restrict__ pack_start, const int32_t *__restrict__ ray_of_pack, int64_t P,
                                                              const float *__restrict__ g, const float *__restrict__ depths, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t pk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pk >= P) return;
    const int64_t beg = pack_start[pk], end = pack_start[pk + 1];
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t i = beg + lane; i < end; i += 64) {
        const float dep = depths[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = g[i * 3 + c];
            a[c] += v;
            a[3 + c] += v * dep;
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = wave_sum(a[c]);
    if (lane == 0) {
        float *o = out + (int64_t)ray_of_pack[pk] * 6;
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] = a[c];
    }
}

extern "C" int pag_ray_sample_grad(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const float *grad_samples, const float *depths,
                                   float *out, void *stream) {
    PAG_CHECK_ARG(P >= 0, "pag_ray_sample_grad: P < 0");
    if (P == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && ray_of_pack && grad_samples && depths && out, "pag_ray_sample_grad: NULL input/output");
    hipLaunchKernelGGL(ray_sample_grad_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pack_start, ray_of_pack, P, grad_samples,
                       depths, out);
    PAG_CHECK_LAUNCH("pag_ray_sample_grad");
    return PAG_OK;
}

extern "C" int pag_composite_feats_fwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const float *weights,
                                       const float *alpha, const void *feats, int feat_dtype, int C, float *out, void *stream) {
    PAG_CHECK_ARG(P >= 0, "pag_composite_feats_fwd: P < 0");
    PAG_CHECK_ARG(C >= 1 && C <= 256, "pag_composite_feats_fwd: C %d not in [1,256]", C);
    PAG_CHECK_ARG(feat_dtype == PAG_F32 || feat_dtype == PAG_BF16, "pag_composite_feats_fwd: feats dtype must be F32 or BF16");
    if (P == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && ray_of_pack && weights && alpha && feats && out, "pag_composite_feats_fwd: NULL input");
    if (C <= 16) {
        const dim3 g((unsigned)((P + 3) / 4));
        if (feat_dtype == PAG_F32)
            hipLaunchKernelGGL((composite_feats_small_fwd_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, pack_start, ray_of_pack, P,
                               weights, alpha, (const float *)feats, C, out);
        else
            hipLaunchKernelGGL((composite_feats_small_fwd_kernel<bf16_t>), g, dim3(256), 0, (hipStream_t)stream, pack_start, ray_of_pack, P,
                               weights, alpha, (const bf16_t *)feats, C, out);
    } else if (feat_dtype == PAG_BF16 && C == 64 && (reinterpret_cast<uintptr_t>(feats) & 15) == 0) {
        hipLaunchKernelGGL(composite_feats64_bf16_fwd_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pack_start, ray_of_pack, weights, alpha,
                           (const bf16_t *)feats, out);
    } else {
        const bool vec = (C % 4) == 0;
        const dim3 g((unsigned)P);
        hipStream_t st = (hipStream_t)stream;
        if (feat_dtype == PAG_F32 && vec)
            hipLaunchKernelGGL((composite_feats_fwd_kernel<float, true>), g, dim3(256), 0, st, pack_start, ray_of_pack, weights, alpha, (const float *)feats, C, out);
        else if (feat_dtype == PAG_F32)
            hipLaunchKernelGGL((composite_feats_fwd_kernel<float, false>), g, dim3(256), 0, st, pack_start, ray_of_pack, weights, alpha, (const float *)feats, C, out);
        else if (vec)
            hipLaunchKernelGGL((composite_feats_fwd_kernel<bf16_t, true>), g, dim3(256), 0, st, pack_start, ray_of_pack, weights, alpha, (const bf16_t *)feats, C, out);
        else
            hipLaunchKernelGGL((composite_feats_fwd_kernel<bf16_t, false>), g, dim3(256), 0, st, pack_start, ray_of_pack, weights, alpha, (const bf16_t *)feats, C, out);
    }
    PAG_CHECK_LAUNCH("pag_composite_feats_fwd");
    return PAG_OK;
}

extern "C" int pag_composite_feats_bwd(const int64_t *pack_start, const int32_t *ray_of_pack, int64_t P, const float *weights,
                                       const float *alpha, const float *g_out, int C, void *d_feats, int feat_dtype,
                                       void *stream) {
    PAG_CHECK_ARG(P >= 0, "pag_composite_feats_bwd: P < 0");
    PAG_CHECK_ARG(C >= 1 && C <= 256, "pag_composite_feats_bwd: C %d not in [1,256]", C);
    PAG_CHECK_ARG(feat_dtype == PAG_F32 || feat_dtype == PAG_BF16, "pag_composite_feats_bwd: feats dtype must be F32 or BF16");
    if (P == 0) return PAG_OK;
    PAG_CHECK_ARG(pack_start && ray_of_pack && weights && alpha && g_out && d_feats, "pag_composite_feats_bwd: NULL input");
    if (feat_dtype == PAG_F32)
        hipLaunchKernelGGL((composite_feats_bwd_kernel<float>), dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pack_start,
                           ray_of_pack, weights, alpha, g_out, C, (float *)d_feats);
    else
        hipLaunchKernelGGL((composite_feats_bwd_kernel<bf16_t>), dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pack_start,
                           ray_of_pack, weights, alpha, g_out, C, (bf16_t *)d_feats);
    PAG_CHECK_LAUNCH("pag_composite_feats_bwd");
    return PAG_OK;
}

static size_t voxel_coarse_lds(int level) {
    const int RC = (1 << level) >> 2;
    return (size_t)((RC * RC * RC + 31) / 32) * sizeof(uint32_t);
}

extern "C" int64_t pag_occupancy_coarse_bytes(int blas_level) {
    if (blas_level < 5 || blas_level > 8) return 0;       // below 32^3 the walk is short; above 256^3 the coarse grid outgrows LDS
    return (int64_t)voxel_coarse_lds(blas_level);
}

extern "C" int pag_occupancy_coarse(const uint32_t *occupancy_bits, int blas_level, uint32_t *coarse, void *stream) {
    PAG_CHECK_ARG(pag_occupancy_coarse_bytes(blas_level) > 0, "pag_occupancy_coarse: blas_level %d not in [5,8]", blas_level);
    PAG_CHECK_ARG(occupancy_bits && coarse, "pag_occupancy_coarse: NULL input/output");
    const int RC = (1 << blas_level) >> 2;
    hipLaunchKernelGGL(occupancy_coarse_kernel, dim3((unsigned)((RC * RC * RC + 255) / 256)), dim3(256), 0, (hipStream_t)stream, occupancy_bits,
                       blas_level, coarse);
    PAG_CHECK_LAUNCH("pag_occupancy_coarse");
    return PAG_OK;
}

extern "C" int pag_raymarch_voxel_count(const float *origins, const float *dirs, int64_t N, int samples_per_voxel, float dist_min,
                                        float dist_max, const uint32_t *occupancy_bits, const uint32_t *occupancy_coarse, int blas_level,
                                        float max_travel, int32_t *counts, void *stream) {
    PAG_CHECK_ARG(N >= 0, "pag_raymarch_voxel_count: N < 0");
    PAG_CHECK_ARG(samples_per_voxel >= 1 && samples_per_voxel <= 64, "pag_raymarch_voxel_count: samples_per_voxel %d not in [1,64]", samples_per_voxel);
    PAG_CHECK_ARG(blas_level >= 0 && blas_level <= 10, "pag_raymarch_voxel_count: blas_level %d not in [0,10]", blas_level);
    PAG_CHECK_ARG(!occupancy_coarse || (occupancy_bits && pag_occupancy_coarse_bytes(blas_level) > 0),
                  "pag_raymarch_voxel_count: a coarse grid needs the occupancy bits and blas_level in [5,8]");
    PAG_CHECK_ARG(!(max_travel != max_travel), "pag_raymarch_voxel_count: max_travel is NaN");
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(origins && dirs && counts, "pag_raymarch_voxel_count: NULL input");
    MarchArgs a{origins, dirs, nullptr, nullptr, occupancy_bits, N, 0, blas_level, dist_min, dist_max};
    const size_t lds = occupancy_coarse ? voxel_coarse_lds(blas_level) : 0;
    hipLaunchKernelGGL((voxel_march_kernel<false>), dim3((unsigned)((N + 63) / 64)), dim3(64), lds, (hipStream_t)stream, a, samples_per_voxel,
                       max_travel, occupancy_coarse, counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    PAG_CHECK_LAUNCH("pag_raymarch_voxel_count");
    return PAG_OK;
}

extern "C" int64_t pag_raymarch_voxel_nugget_capacity(int blas_level) { return 3 * ((int64_t)1 << blas_level) + 3; }

extern "C" int pag_raymarch_voxel_count_nuggets(const float *origins, const float *dirs, int64_t N, int samples_per_voxel, float dist_min,
                                                float dist_max, const uint32_t *occupancy_bits, const uint32_t *occupancy_coarse, int blas_level,
                                                float max_travel, int32_t *counts, float *nugget_t, int32_t *nugget_cell, void *stream) {
    PAG_CHECK_ARG(N >= 0, "pag_raymarch_voxel_count_nuggets: N < 0");
    PAG_CHECK_ARG(samples_per_voxel >= 1 && samples_per_voxel <= 64, "pag_raymarch_voxel_count_nuggets: samples_per_voxel %d not in [1,64]", samples_per_voxel);
    PAG_CHECK_ARG(blas_level >= 0 && blas_level <= 10, "pag_raymarch_voxel_count_nuggets: blas_level %d not in [0,10]", blas_level);
    PAG_CHECK_ARG(!occupancy_coarse || (occupancy_bits && pag_occupancy_coarse_bytes(blas_level) > 0),
                  "pag_raymarch_voxel_count_nuggets: a coarse grid needs the occupancy bits and blas_level in [5,8]");
    PAG_CHECK_ARG(!(max_travel != max_travel), "pag_raymarch_voxel_count_nuggets: max_travel is NaN");
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(origins && dirs && counts && nugget_t && nugget_cell, "pag_raymarch_voxel_count_nuggets: NULL input/output");
    MarchArgs a{origins, dirs, nullptr, nullptr, occupancy_bits, N, 0, blas_level, dist_min, dist_max};
    (void)occupancy_coarse;      // the walk no longer reads the occupancy: the coarse grid has nothing to shortcut
    // first half of the scratch: the walk's candidates [step][ray]; second half: the kept nuggets, ray-major [ray][slot] (voxel_select_kernel)
    const int64_t cap = pag_raymarch_voxel_nugget_capacity(blas_level);
    float2 *cand_t = reinterpret_cast<float2 *>(nugget_t), *sel_t = cand_t + cap * N;
    int32_t *cand_cell = nugget_cell, *sel_cell = nugget_cell + cap * N;
    hipLaunchKernelGGL(voxel_walk_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, counts, cand_t, cand_cell);
    hipLaunchKernelGGL(voxel_select_kernel, dim3((unsigned)((N + SEL_RAYS - 1) / SEL_RAYS)), dim3(256), 0, (hipStream_t)stream, occupancy_bits, N, samples_per_voxel,
                       max_travel, counts, (const float2 *)cand_t, (const int32_t *)cand_cell, cap, sel_t, sel_cell);
    PAG_CHECK_LAUNCH("pag_raymarch_voxel_count_nuggets");
    return PAG_OK;
}

extern "C" int pag_raymarch_voxel_pack_nuggets(const float *origins, const float *dirs, int64_t N, int samples_per_voxel, const int64_t *offsets,
                                               const float *nugget_t, const int32_t *nugget_cell, int blas_level, int32_t *ridx, int32_t *pidx, float *samples,
                                               float *depths, float *deltas, uint8_t *boundary, int32_t *ridx_sample, int64_t *ridx64, void *stream) {
    PAG_CHECK_ARG(N >= 0, "pag_raymarch_voxel_pack_nuggets: N < 0");
    PAG_CHECK_ARG(blas_level >= 0 && blas_level <= 10, "pag_raymarch_voxel_pack_nuggets: blas_level %d not in [0,10]", blas_level);
    PAG_CHECK_ARG(samples_per_voxel >= 1 && samples_per_voxel <= 64, "pag_raymarch_voxel_pack_nuggets: samples_per_voxel %d not in [1,64]", samples_per_voxel);
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(origins && dirs && offsets && nugget_t && nugget_cell && ridx && pidx && samples && depths && deltas && boundary,
                  "pag_raymarch_voxel_pack_nuggets: NULL input/output");
    const int64_t cap = pag_raymarch_voxel_nugget_capacity(blas_level);      // the kept nuggets sit ray-major in the second half of the scratch
    hipLaunchKernelGGL(voxel_pack_nuggets_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, origins, dirs, N, samples_per_voxel,
                       offsets, reinterpret_cast<const float2 *>(nugget_t) + cap * N, nugget_cell + cap * N, cap, ridx, pidx, samples, depths, deltas, boundary,
                       ridx_sample, ridx64);
    PAG_CHECK_LAUNCH("pag_raymarch_voxel_pack_nuggets");
    return PAG_OK;
}

extern "C" int pag_raymarch_voxel_pack(const float *origins, const float *dirs, int64_t N, int samples_per_voxel, float dist_min,
                                       float dist_max, const uint32_t *occupancy_bits, const uint32_t *occupancy_coarse, int blas_level,
                                       float max_travel, const int64_t *offsets, int32_t *ridx, int32_t *pidx, float *samples,
                                       float *depths, float *deltas, uint8_t *boundary, int32_t *ridx_sample, int64_t *ridx64,
                                       void *stream) {
    PAG_CHECK_ARG(N >= 0, "pag_raymarch_voxel_pack: N < 0");
    PAG_CHECK_ARG(samples_per_voxel >= 1 && samples_per_voxel <= 64, "pag_raymarch_voxel_pack: samples_per_voxel %d not in [1,64]", samples_per_voxel);
    PAG_CHECK_ARG(blas_level >= 0 && blas_level <= 10, "pag_raymarch_voxel_pack: blas_level %d not in [0,10]", blas_level);
    PAG_CHECK_ARG(!occupancy_coarse || (occupancy_bits && pag_occupancy_coarse_bytes(blas_level) > 0),
                  "pag_raymarch_voxel_pack: a coarse grid needs the occupancy bits and blas_level in [5,8]");
    PAG_CHECK_ARG(!(max_travel != max_travel), "pag_raymarch_voxel_pack: max_travel is NaN");
    if (N == 0) return PAG_OK;
    PAG_CHECK_ARG(origins && dirs && offsets && ridx && pidx && samples && depths && deltas && boundary, "pag_raymarch_voxel_pack: NULL input/output");
    MarchArgs a{origins, dirs, nullptr, nullptr, occupancy_bits, N, 0, blas_level, dist_min, dist_max, ridx64};
    const size_t lds = occupancy_coarse ? voxel_coarse_lds(blas_level) : 0;
    hipLaunchKernelGGL((voxel_march_kernel<true>), dim3((unsigned)((N + 63) / 64)), dim3(64), lds, (hipStream_t)stream, a, samples_per_voxel,
                       max_travel, occupancy_coarse, nullptr, offsets, ridx, pidx, samples, depths, deltas, boundary, ridx_sample);
    PAG_CHECK_LAUNCH("pag_raymarch_voxel_pack");
    return PAG_OK;
}

// ------------------------------------------------------------------------------------------- occupancy update (prune)
// panoptic_delta_nef.py:74-75,90-104: occupancy <- max(density, occupancy * decay); cell kept iff occupancy > min_density.
// One lane per cell; a wave's ballot is two words of the bitfield the march kernels read.
__global__ __launch_bounds__(256) void occupancy_update_kernel(const float *__restrict__ density, int64_t density_stride,
                                                               float *__restrict__ occupancy, uint32_t *__restrict__ bits,
                                                               int64_t num_cells, float decay, float min_density) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < num_cells) {
        float o = fmaxf(density[i * density_stride], occupancy[i] * decay);
        occupancy[i] = o;
        keep = o > min_density;
    }
    uint64_t m = __ballot(keep);
    int lane = threadIdx.x & 63;
    int64_t word = (i - lane) >> 5;
    int64_t words = (num_cells + 31) >> 5;
    if (lane == 0 && word < words) bits[word] = (uint32_t)m;
    if (lane == 32 && word + 1 < words) bits[word + 1] = (uint32_t)(m >> 32);
}

extern "C" int pag_occupancy_update(const float *density, int64_t density_stride, float *occupancy, uint32_t *occupancy_bits,
                                    int64_t num_cells, float decay, float min_density, void *stream) {
    PAG_CHECK_ARG(num_cells >= 0, "pag_occupancy_update: num_cells < 0");
    PAG_CHECK_ARG(density_stride >= 1, "pag_occupancy_update: density_stride < 1");
    if (num_cells == 0) return PAG_OK;
    PAG_CHECK_ARG(density && occupancy && occupancy_bits, "pag_occupancy_update: NULL input/output");
    hipLaunchKernelGGL(occupancy_update_kernel, dim3((unsigned)((num_cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, density,
                       density_stride, occupancy, occupancy_bits, num_cells, decay, min_density);
    PAG_CHECK_LAUNCH("pag_occupancy_update");
    return PAG_OK;
}
