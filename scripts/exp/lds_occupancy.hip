// Experiment: how many 1024-thread workgroups with S bytes of static + D bytes of dynamic LDS does one gfx950 CU hold?
// hipcc --offload-arch=gfx950 -O2 scripts/exp/lds_occupancy.hip -o gpurun_out/lds_occupancy && ./gpurun_out/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int STATIC_WORDS>
__global__ __launch_bounds__(1024) void spin(unsigned long long *out, int ticks) {
    extern __shared__ unsigned long long dyn[];
    __shared__ unsigned int st[STATIC_WORDS];
    if (threadIdx.x < STATIC_WORDS) st[threadIdx.x] = threadIdx.x;
    dyn[threadIdx.x] = threadIdx.x;
#ifdef TOP_VGPR
#define STR2(x) #x
#define STR(x) STR2(x)
    asm volatile("v_mov_b32 v" STR(TOP_VGPR) ", 0" : : : "v" STR(TOP_VGPR));      // raises the kernel's VGPR allocation to TOP_VGPR + 1
#endif
#ifdef TOP_SGPR
    asm volatile("s_mov_b32 s" STR(TOP_SGPR) ", 0" : : : "s" STR(TOP_SGPR));
#endif
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = t0 + st[dyn[1] & 1] * 0;
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
template <int SW>
void run(int dyn_bytes, int threads) {
    const int nb = 2048;
    unsigned long long *d;
    hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void *)spin<SW>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_bytes);
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<SW>, threads, dyn_bytes);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(spin<SW>, dim3(nb), dim3(threads), dyn_bytes, 0, d, 2000);      // 20 us
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < nb; ++i) t0 = std::min(t0, h[2 * i]);
    int resident = 0;                                  // blocks running 10 us after the first one started
    for (int i = 0; i < nb; ++i) if (h[2 * i] <= t0 + 1000 && h[2 * i + 1] > t0 + 1000) ++resident;
    printf("static %5d B dynamic %6d B threads %4d: runtime says %d per CU, measured %d resident (%.2f per CU) %s\n", SW * 4, dyn_bytes, threads, occ, resident, resident / 256.0,
           e == hipSuccess ? "" : hipGetErrorString(e));
    hipFree(d);
}
int main() {
    for (int threads : {1024, 512}) {
        run<2308>(65536, threads);
        run<2308>(57344, threads);
        run<2308>(49152, threads);
        run<2308>(32768, threads);
        run<16>(65536, threads);
        run<16>(81920 - 64, threads);
        run<16>(73728, threads);
    }
    return 0;
}
