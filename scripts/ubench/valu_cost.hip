// Microbenchmark: issue cost of the VALU instructions the encode kernels are made of (gfx950), per wave-instruction and SIMD,
// at 1 / 2 / 4 / 8 waves per SIMD.  Each test runs N_CH independent dependency chains of one instruction, unrolled, inside a loop;
// cycles from s_memtime around the loop (one workgroup per CU, waves-per-SIMD = blockDim / 256).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_cost.hip -o scripts/ubench/valu_cost && scripts/ubench/valu_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X X X X X X X X
#define ITERS 256

// 8 independent chains a0..a7 (VGPRs), operands b (VGPR), s (SGPR).  BODY uses %0..%7 as in/out, %8 = b, %9 = b2 (pairs use even regs)
#define KERNEL(NAME, BODY)                                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(float *out, long long *cyc, float bval) {                                   \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;   \
        float b = bval, b2 = bval + 1.0f;                                                                                    \
        long long t0 = __builtin_amdgcn_s_memtime();                                                                         \
        for (int it = 0; it < ITERS; ++it) {                                                                                 \
            asm volatile(REP8(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(b2) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); \
        }                                                                                                                    \
        long long t1 = __builtin_amdgcn_s_memtime();                                                                         \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                  \
    }

#define OP8(OP, SRC) OP " %0, %0, " SRC "\n" OP " %1, %1, " SRC "\n" OP " %2, %2, " SRC "\n" OP " %3, %3, " SRC "\n" OP " %4, %4, " SRC "\n" OP " %5, %5, " SRC "\n" OP " %6, %6, " SRC "\n" OP " %7, %7, " SRC "\n"
#define OP8U(OP) OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"
#define OP8T(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"

KERNEL(k_add_f32, OP8("v_add_f32", "%8"))
KERNEL(k_mul_f32, OP8("v_mul_f32", "%8"))
KERNEL(k_fma_f32, OP8T("v_fma_f32"))
KERNEL(k_add_u32, OP8("v_add_u32", "%8"))
KERNEL(k_and_b32, OP8("v_and_b32", "%8"))
KERNEL(k_ashr, OP8("v_ashrrev_i32", "%8"))
KERNEL(k_add3_u32, OP8T("v_add3_u32"))
KERNEL(k_lshl_add_u32, OP8T("v_lshl_add_u32"))
KERNEL(k_mul_lo_u32, OP8("v_mul_lo_u32", "%8"))
KERNEL(k_mul_hi_u32, OP8("v_mul_hi_u32", "%8"))
KERNEL(k_mul_u32_u24, OP8("v_mul_u32_u24", "%8"))
KERNEL(k_mad_u32_u24, OP8T("v_mad_u32_u24"))
KERNEL(k_floor, OP8U("v_floor_f32"))
KERNEL(k_fract, OP8U("v_fract_f32"))
KERNEL(k_rndne, OP8U("v_rndne_f32"))
KERNEL(k_cvt_i32_f32, OP8U("v_cvt_i32_f32"))
KERNEL(k_cvt_f32_i32, OP8U("v_cvt_f32_i32"))
KERNEL(k_cndmask, OP8("v_cndmask_b32", "%8, vcc"))
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %8, vcc\n")
KERNEL(k_cmp_sgpr_cnd, "v_cmp_lt_f32 s[20:21], %0, %8\nv_cndmask_b32 %0, %0, %8, s[20:21]\nv_cmp_lt_f32 s[22:23], %1, %8\nv_cndmask_b32 %1, %1, %8, s[22:23]\nv_cmp_lt_f32 s[24:25], %2, %8\nv_cndmask_b32 %2, %2, %8, s[24:25]\nv_cmp_lt_f32 s[26:27], %3, %8\nv_cndmask_b32 %3, %3, %8, s[26:27]\n")
KERNEL(k_cvt_pk_bf16, OP8("v_cvt_pk_bf16_f32", "%8"))
KERNEL(k_perm, OP8T("v_perm_b32"))
KERNEL(k_bfe, OP8T("v_bfe_u32"))
KERNEL(k_max_f32, OP8("v_max_f32", "%8"))
KERNEL(k_med3, OP8T("v_med3_f32"))
KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")

// packed fp32 and 64-bit forms work on register pairs: 4 chains of pairs
#define KERNEL2(NAME, BODY)                                                                                                  \
    __global__ __launch_bounds__(1024) void NAME(float *out, long long *cyc, float bval) {                                   \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                               \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;                                      \
        f2 b = {bval, bval + 1.0f};                                                                                          \
        long long t0 = __builtin_amdgcn_s_memtime();                                                                         \
        for (int it = 0; it < ITERS; ++it) {                                                                                 \
            asm volatile(REP8(BODY BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));                                 \
        }                                                                                                                    \
        long long t1 = __builtin_amdgcn_s_memtime();                                                                         \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                    \
        f2 s = a0 + a1 + a2 + a3;                                                                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                                              \
    }
#define P4(OP) OP " %0, %0, %4\n" OP " %1, %1, %4\n" OP " %2, %2, %4\n" OP " %3, %3, %4\n"
#define P4T(OP) OP " %0, %0, %4, %4\n" OP " %1, %1, %4, %4\n" OP " %2, %2, %4, %4\n" OP " %3, %3, %4, %4\n"
KERNEL2(k_pk_add_f32, P4("v_pk_add_f32"))
KERNEL2(k_pk_mul_f32, P4("v_pk_mul_f32"))
KERNEL2(k_pk_fma_f32, P4T("v_pk_fma_f32"))
KERNEL2(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %4\nv_lshl_add_u64 %1, %1, 3, %4\nv_lshl_add_u64 %2, %2, 3, %4\nv_lshl_add_u64 %3, %3, 3, %4\n")
// v_mad_u64_u32 vdst[2], sdst(carry), src0 (32), src1 (32), src2 (64): four independent accumulator pairs
__global__ __launch_bounds__(1024) void k_mad_u64_u32(float *out, long long *cyc, float bval) {
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned m0 = threadIdx.x * 3 + 1, m1 = (unsigned)bval + 2531011u;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(REP8("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\nv_mad_u64_u32 %1, s[20:21], %4, %5, %1\nv_mad_u64_u32 %2, s[20:21], %4, %5, %2\nv_mad_u64_u32 %3, s[20:21], %4, %5, %3\n"
                          "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\nv_mad_u64_u32 %1, s[20:21], %4, %5, %1\nv_mad_u64_u32 %2, s[20:21], %4, %5, %2\nv_mad_u64_u32 %3, s[20:21], %4, %5, %3\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m0), "v"(m1) : "s20", "s21");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}

typedef void (*kern_t)(float *, long long *, float);
struct Test { const char *name; kern_t k; int per_rep; };

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, 512 * 16 * 8);
    Test tests[] = {
        {"v_add_f32", k_add_f32, 8}, {"v_mul_f32", k_mul_f32, 8}, {"v_fma_f32", k_fma_f32, 8}, {"v_add_u32", k_add_u32, 8},
        {"v_and_b32", k_and_b32, 8}, {"v_ashrrev_i32", k_ashr, 8}, {"v_add3_u32", k_add3_u32, 8}, {"v_lshl_add_u32", k_lshl_add_u32, 8},
        {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8}, {"v_mul_u32_u24", k_mul_u32_u24, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8},
        {"v_floor_f32", k_floor, 8}, {"v_fract_f32", k_fract, 8}, {"v_rndne_f32", k_rndne, 8}, {"v_cvt_i32_f32", k_cvt_i32_f32, 8},
        {"v_cvt_f32_i32", k_cvt_f32_i32, 8}, {"v_cndmask_b32(vcc)", k_cndmask, 8}, {"v_cmp+v_cndmask (vcc) [pairs]", k_cmp_cnd, 4},
        {"v_cmp+v_cndmask (sgpr) [pairs]", k_cmp_sgpr_cnd, 4}, {"v_cvt_pk_bf16_f32", k_cvt_pk_bf16, 8}, {"v_perm_b32", k_perm, 8},
        {"v_bfe_u32", k_bfe, 8}, {"v_max_f32", k_max_f32, 8}, {"v_med3_f32", k_med3, 8}, {"v_mov_b32 dpp row_shr", k_dpp_mov, 8},
        {"v_add_f32 dpp row_shr", k_add_dpp, 8},
        {"v_pk_add_f32", k_pk_add_f32, 8}, {"v_pk_mul_f32", k_pk_mul_f32, 8}, {"v_pk_fma_f32", k_pk_fma_f32, 8},
        {"v_lshl_add_u64", k_lshl_add_u64, 8}, {"v_mad_u64_u32", k_mad_u64_u32, 8},
    };
    static long long h[512 * 16];
    printf("%-34s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD, s_memtime ticks; waves per SIMD = 1, 2, 4, 8)\n", "instruction", "w=1", "w=2", "w=4",
           "w=8");
    for (auto &t : tests) {
        printf("%-34s", t.name);
        for (int w : {1, 2, 4, 8}) {
            const int threads = 256 * w > 1024 ? 1024 : 256 * w;
            const int blocks = 256 * (256 * w > 1024 ? (256 * w) / 1024 : 1);      // w = 8: two 1024-thread workgroups per CU
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int b = 0; b < blocks; ++b)
                for (int wv = 0; wv < threads / 64; ++wv) { sum += (double)h[b * 16 + wv]; ++n; }
            const double per_wave = sum / n;                                          // cycles one wave spent in the loop
            const double instr = (double)ITERS * 8 * t.per_rep;                       // wave-instructions it issued
            // with w waves sharing the SIMD, the SIMD issued w * instr in per_wave cycles
            printf(" %8.2f", per_wave / (instr * w));
        }
        printf("\n");
    }
    return 0;
}
