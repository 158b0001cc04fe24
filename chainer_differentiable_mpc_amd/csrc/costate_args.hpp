// costate_args.hpp - argument block and shape dispatch of the co-state / outer-product kernel
// (kernels in costate_kernels.hpp, instantiated once in kkt_api.hip; also called by mpc_api.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dmpc {

struct CostateArgs {
  int T, B;
  const float *C, *c, *F;
  const float *x, *u;    // tau   [T,B,nx], [T,B,nu]
  const float *dx, *du;  // dtau' [T,B,nx], [T,B,nu]  (solution of the second LQR solve)
  const float *r;        // affine term of the d_lambda recursion, rows of length ns (or r_cols): [T,B,ns]; only [:nx] is read
  float r_sign;          // d_lambda_t = r_sign * r_t[:nx] + ...      (+1 DiffLqr :115,124; -1 MPCstep :417,420)
  float out_sign;        // +1 DiffLqr, -1 MPCstep
  int dC_mode;           // 0: 0.5*dtau(x)tau + tau(x)dtau (differentiable_lqr.py:128); 1: 0.5*(dtau(x)tau + tau(x)dtau)
  int df_shift;          // 0: df[t] = d_lambda[t] (differentiable_lqr.py:133); 1: df[t] = d_lambda[t+1]
  float *dx0, *dC, *dc, *dF, *df;  // outputs (dC, dF, df may be nullptr)
  int r_cols = 0;        // row length of r; 0 = ns.  nx: r is the state part alone (DiffLqr: grad_x as it is)
  // Gradients of a cost that is ONE (C, c) tiled over time and batch (env_dx/il_env.py:119-129: the imitation loop's
  // learnable cost): dC_sum [ns,ns] += sum_{t,b} dC[t][b], dc_sum [ns] += sum_{t,b} dc[t][b] - what autograd's backward
  // of the tiling would reduce dC, dc to (326 K floats at config 4).  Accumulated per wavefront over its trajectories and
  // timesteps, one atomic add per element and wavefront; the caller zeroes them.  dC / dc may then be nullptr.
  float *dC_sum = nullptr, *dc_sum = nullptr;
  int nx_log = 0, nu_log = 0;   // container launches (costate_kernel<..., PAD>): the problem's own dimensions
};

// Shape dispatch (defined in kkt_api.hip).
int launch_costate(int nx, int nu, const CostateArgs &a, hipStream_t stream);
// dC_sum / dc_sum are formed by the LDS-DMA co-state kernel alone: does launch_costate take that kernel for this size?
// (callers refuse the sums with DMPC_E_UNSUPPORTED otherwise - the other kernels would leave them untouched)
bool costate_sums_available(int T, int B, int nx, int nu);

}  // namespace dmpc
