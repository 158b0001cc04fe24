// mpc_step_fused_kernel.hpp - MPCstep.forward (mpc/mpc_step.py:288-368) as ONE launch: the generated backward stream
// (mpc_asm_kernel.hpp: Riccati sweep + the box QP of every timestep, mpc_step.py:70-173) followed, in the same
// wavefront, by the generated line-search stream (mpc_fwd_asm_kernel.hpp, mpc_step.py:175-286).
//
// A wavefront's line search reads nothing but its own four trajectories' gains, so nothing has to wait for the rest of
// the grid: the launch boundary between the two kernels (the sweep's tail - wavefronts with more QP passes finish
// later - the launch gap, the line search's first DMA round trip on an idle chip) turns into wavefronts that move on
// while their neighbours still sweep, and the line search's first timesteps find C_0.., F_0.. in the L2 the sweep's last
// steps left them in.  The gains still go to HBM (they are outputs) and come back through the line search's DMA ring:
// the workgroup's stores are drained and the L1 invalidated in between (a 128-byte line of `ks` holds the rows of all
// four wavefronts of a workgroup).  Same two instruction streams, same results bit for bit.
#pragma once
#include "mpc_asm_kernel.hpp"
#include "mpc_fwd_asm_kernel.hpp"

namespace dmpc {

template <int NX, int NU>
constexpr size_t mpc_step_fused_lds_bytes() {
  return mpc_asm_lds_bytes<NX, NU>() > mpc_fwd_asm_lds_bytes<NX, NU>() ? mpc_asm_lds_bytes<NX, NU>()
                                                                      : mpc_fwd_asm_lds_bytes<NX, NU>();
}

template <int NX, int NU, bool HAS_F, bool EXPAND>
__global__ __launch_bounds__(256) void mpc_step_fused_asm_kernel(const MpcBackArgs ba, const MpcFwdArgs fa) {
  mpc_backward_asm_body<NX, NU, HAS_F, EXPAND>(ba, blockIdx.x);
  // gains out (and the ring's last DMAs in) before the line search's DMAs read them / reuse the ring.  Workgroup scope is
  // enough - and agent scope measurably wrong: its release / acquire write back and invalidate the L2, 1,024 times per
  // launch, 111 us against the two launches' 99.5 - because a wavefront reads back only what its own workgroup stored:
  // same CU, same (write-through) L1, same L2.  (the explicit wait: the compiler's counters know nothing of what the
  // stream issued)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#ifndef DMPC_FUSED_NO_BARRIER
  __syncthreads();
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  mpc_forward_asm_body<NX, NU>(fa, blockIdx.x);
}

}  // namespace dmpc
