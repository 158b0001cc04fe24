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

// A contiguous run of `floats` >= 1 floats at `src` (wave-uniform, any 4-byte alignment, a run-time length) -> LDS byte address
// `dst` (wave-uniform), 64 floats per instruction: lane l copies float o + l.  Lanes past the end read the last float again;
// their LDS words are padding - the regions such runs land in are whole multiples of 64 floats.
__device__ __forceinline__ void dma_run_floats(const float *src, unsigned dst, int floats, int lane) {
  const unsigned long long base = reinterpret_cast<unsigned long long>(src);
  for (int o = 0; o < floats; o += 64) {   // uniform
    set_m0(dst + o * 4);
    const int e = o + lane;
    const unsigned voff = (unsigned)(e < floats ? e : floats - 1) * 4u;
    asm volatile("global_load_lds_dword %0, %1" DMPC_DMA_POLICY ::"v"(voff), "s"(base) : "memory");
  }
}

// s_waitcnt vmcnt takes an immediate: the largest of a few values that does not exceed n (waiting for more is always safe)
__device__ __forceinline__ void wait_vmcnt_at_most(int n) {   // n uniform
  if (n >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
  else if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else if (n >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace dmpc
