// dma_gather.hpp - the per-lane gather form of LDS-DMA, shared by the kernels that stage their inputs through an LDS ring
// (costate_dma_kernel.hpp, mpc_dma_kernels.hpp).
#pragma once
#include <hip/hip_runtime.h>

namespace dmpc {

// One LDS-DMA instruction of the per-lane gather form: lane l copies the 16 bytes at its own global address to LDS at
// M0 + OFFSET + 16 l.  The instruction offset moves the global address as well, so the pointers carry -OFFSET.
// DMPC_DMA_NT (build flag, timing experiments): the nt cache policy on these loads.
#ifdef DMPC_DMA_NT
#define DMPC_DMA_POLICY " nt"
#else
#define DMPC_DMA_POLICY ""
#endif
template <int OFFSET>
__device__ __forceinline__ void dma16_gather(unsigned long long ptr) {
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" DMPC_DMA_POLICY ::"v"(ptr), "n"(OFFSET) : "memory");
}
__device__ __forceinline__ void set_m0(unsigned lds_dst) {  // + the wait state an LDS-DMA needs after an M0 write
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_dst) : "memory");
}

}  // namespace dmpc
