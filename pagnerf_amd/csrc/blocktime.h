// Experiment builds only (-DPAG_BLOCK_TIMING, scripts/block_timeline.py): every workgroup of an instrumented kernel records when it started
// and when its first thread left, on the 100 MHz clock all CUs share, and which XCD / CU it ran on.  The table is per translation unit
// and per slot (one slot per kernel; a later launch of the same kernel overwrites the earlier one).  Nothing of this exists in the
// regular build.
#pragma once
#ifdef PAG_BLOCK_TIMING
#include <hip/hip_runtime.h>
#define PAG_BT_SLOTS 8
#define PAG_BT_BLOCKS 32768
namespace {
__device__ unsigned long long pag_bt[PAG_BT_SLOTS][PAG_BT_BLOCKS][3];
__device__ unsigned int pag_bt_grid[PAG_BT_SLOTS][4];
struct PagBlockTimer {
    unsigned long long t0;
    int slot;
    __device__ __forceinline__ explicit PagBlockTimer(int s) : slot(s) { t0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ ~PagBlockTimer() {
        const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (threadIdx.x == 0 && b < PAG_BT_BLOCKS) {
            pag_bt[slot][b][0] = t0;
            pag_bt[slot][b][1] = __builtin_amdgcn_s_memrealtime();
            pag_bt[slot][b][2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);      // HW_ID, XCC_ID
            if (b == 0) {
                pag_bt_grid[slot][0] = gridDim.x; pag_bt_grid[slot][1] = gridDim.y; pag_bt_grid[slot][2] = gridDim.z; pag_bt_grid[slot][3] = blockDim.x;
            }
        }
    }
};
}
#define PAG_BLOCK_TIMER(slot) PagBlockTimer pag_bt_guard_(slot)
#define PAG_BLOCK_TIMING_EXPORT(tu)                                                                                        \
    extern "C" int pag_debug_block_times_##tu(void *times, void *grids) {                                                  \
        hipError_t e = hipMemcpyFromSymbol(times, HIP_SYMBOL(pag_bt), sizeof(unsigned long long) * PAG_BT_SLOTS * PAG_BT_BLOCKS * 3);   \
        if (e == hipSuccess) e = hipMemcpyFromSymbol(grids, HIP_SYMBOL(pag_bt_grid), sizeof(unsigned int) * PAG_BT_SLOTS * 4);          \
        void *p = nullptr;                                                                                                 \
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(pag_bt)) == hipSuccess) hipMemset(p, 0, sizeof(unsigned long long) * PAG_BT_SLOTS * PAG_BT_BLOCKS * 3); \
        return (int)e;                                                                                                     \
    }
#else
#define PAG_BLOCK_TIMER(slot)
#define PAG_BLOCK_TIMING_EXPORT(tu)
#endif
