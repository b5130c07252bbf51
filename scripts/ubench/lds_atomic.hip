// Microbenchmark: LDS atomic throughput on gfx950 (float add vs int add vs 64-bit int add vs plain RMW).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_atomic.hip -o scripts/ubench/lds_atomic && scripts/ubench/lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t *keys, int iters, int active, float *out, long long *cyc) {
    extern __shared__ float acc[];
    uint32_t *acci = (uint32_t *)acc;
    unsigned long long *accl = (unsigned long long *)acc;
    for (int j = threadIdx.x; j < 16384; j += blockDim.x) acc[j] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t key = keys[blockIdx.x * 1024 + threadIdx.x];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        key = (key * 1664525u + 1013904223u);
        uint32_t a = (key >> 8) & 8191u;
        if (lane < active) {
            if (MODE == 0) atomicAdd(&acc[a * 2], 1.0f);
            if (MODE == 1) atomicAdd(&acci[a * 2], 1u);
            if (MODE == 2) atomicAdd(&accl[a], 1ull);
            if (MODE == 3) acc[a * 2] += 1.0f;
            if (MODE == 4) { atomicAdd(&acc[a * 2], 1.0f); atomicAdd(&acc[a * 2 + 1], 2.0f); }
            if (MODE == 5) atomicAdd(&acci[a], 1u);
        }
    }
    long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0;
    for (int j = threadIdx.x; j < 16384; j += blockDim.x) s += acc[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
int main() {
    const int blocks = 512, iters = 2000;
    uint32_t *keys; float *out; long long *cyc;
    hipMalloc(&keys, blocks * 1024 * 4); hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    uint32_t *h = (uint32_t *)malloc(blocks * 1024 * 4);
    for (int i = 0; i < blocks * 1024; ++i) h[i] = i * 2654435761u + 12345u;
    hipMemcpy(keys, h, blocks * 1024 * 4, hipMemcpyHostToDevice);
    const char *names[] = {"ds_add_f32", "ds_add_u32 (stride2)", "ds_add_u64", "plain rmw f32", "2x ds_add_f32", "ds_add_u32 (dense)"};
    for (int active : {64, 16, 4}) {
        for (int mode = 0; mode < 6; ++mode) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 65536, 0, keys, iters, active, out, cyc); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 65536, 0, keys, iters, active, out, cyc); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 65536, 0, keys, iters, active, out, cyc); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 65536, 0, keys, iters, active, out, cyc); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 65536, 0, keys, iters, active, out, cyc); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 65536, 0, keys, iters, active, out, cyc); break;
                }
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            // per CU: 2 blocks x 16 waves resident; wave-instructions per CU = blocks/256 * 16 * iters
            double winstr_per_cu = (double)blocks / 256 * 16 * iters * (mode == 4 ? 2 : 1);
            printf("active %2d %-22s %.3f ms  -> %.1f ns per wave-instr per CU (%.1f cycles @2.4GHz), %.2f G lane-ops/s chip\n", active,
                   names[mode], ms, ms * 1e6 / winstr_per_cu, ms * 1e6 / winstr_per_cu * 2.4,
                   (double)blocks * 1024 / 64 * active * iters * (mode == 4 ? 2 : 1) / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
