// Shared helpers for the gfx950 kernels.  CDNA4 only: 64-wide wavefronts are hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pagnerf_hip.h"

// PAG_LAYOUT_XCD8: which level sits in slot j of XCD group g.  "Snake" order - even bands of 8 levels ascend with g, odd bands descend -
// so that every group (= every XCD: workgroup b of the encoders lands on XCD b % 8) holds a mix of cheap coarse and expensive fine
// levels.  With band j simply ascending, group 7 of the 24-level permutohedral grid held the finest level of every band and the forward
// waited for that one XCD: 0.458 -> 0.386 ms per launch (scripts/bench_encode_levels.py).  Levels >= n_levels are padding.
__host__ __device__ __forceinline__ int xcd8_level(int g, int j) { return (j & 1) ? 8 * j + 7 - g : 8 * j + g; }

#define PAG_WAVE 64

typedef __bf16 bf16_t;

void pag_set_error(const char *fmt, ...);

#define PAG_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            pag_set_error(__VA_ARGS__);          \
            return PAG_ERR_ARG;                  \
        }                                        \
    } while (0)

#define PAG_CHECK_LAUNCH(name)                                               \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            pag_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return PAG_ERR_LAUNCH;                                           \
        }                                                                    \
    } while (0)

// element load / store with conversion to / from f32
__device__ __forceinline__ float pag_ld(const float *p) { return *p; }
__device__ __forceinline__ float pag_ld(const bf16_t *p) { return (float)*p; }
__device__ __forceinline__ float pag_ld(const __half *p) { return __half2float(*p); }
__device__ __forceinline__ void pag_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void pag_st(bf16_t *p, float v) { *p = (bf16_t)v; }
