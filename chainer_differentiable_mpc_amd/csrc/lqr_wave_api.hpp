// lqr_wave_api.hpp - launcher of the matrix-core backward sweep for the large shapes (lqr_wave_mfma.hpp), kept in its
// own translation unit (lqr_wave_api.hip) so that the kernel can be rebuilt without the generated streams of lqr_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "lqr_kernels.hpp"

namespace dmpc {

// Ks/ks of `a` receive the gains; with `rollout` the same launch goes on with the forward sweep (x, u of `a`).
// Returns 0, a hipError_t or DMPC_E_UNSUPPORTED for a shape without an instance.
int launch_lqr_wave_mfma_backward(int nx, int nu, bool masked, bool rollout, const LqrArgs &a, hipStream_t stream);
// MPCstep.backward_rec with the box QP inside the sweep (a.mpc_* set); (16,8) and (32,8); DMPC_E_UNSUPPORTED otherwise
int launch_mpc_wave_backward(int nx, int nu, const LqrArgs &a, hipStream_t stream);
int launch_mpc_wave_container_backward(int cnx, int cnu, const LqrArgs &a, hipStream_t stream);
// the sweep of a smaller problem (a.nx_log, a.nu_log) padded inside the (cnx, cnu) instance - (16,8) or (32,8)
int launch_lqr_wave_container_sweep(int cnx, int cnu, bool masked, const LqrArgs &a, hipStream_t stream);

// the rollout of a padded problem with its inputs staged through LDS; DMPC_E_UNSUPPORTED (nothing launched) where it does not apply
int launch_lqr_staged_forward(int nx, int nu, const LqrArgs &a, hipStream_t stream);

}  // namespace dmpc
