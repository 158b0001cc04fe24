// mpc_generic.hpp - MPCstep.backward_rec / forward_rec (mpc/mpc_step.py:70-286) for shapes without a
// register-resident specialisation: one wavefront per trajectory, matrices in LDS, lane j owns column j of the
// augmented matrices (the layout of lqr_generic.hpp, the runtime-dimension LQR solve).  The state dimension is a kernel argument; the number of
// controls stays a template parameter (1..8) because the projected-Newton QP and its LU live in registers
// (pnqp_device.hpp).  Covers any nx + nu + 1 <= 64 with nu <= 8, e.g. (32,8).  Completeness path, not the fast path.
#pragma once
#include "dma_gather.hpp"
#include "mpc_kernels.hpp"

namespace dmpc {

constexpr int kMpcGenericMaxNu = 8;

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int NU>
constexpr size_t mpc_generic_back_lds_bytes(int nx) {
  const int ns = nx + NU, nc = ns + 1;
  return (size_t)(3 * nx * nc + ns * nc + 2 * NU * nc + NU * NU + NU) * 4 + (size_t)2 * NU * 4;
}

template <int NU>
__global__ __launch_bounds__(64) void mpc_generic_backward_kernel(const MpcBackArgs a, const int nx) {
  if (a.done != nullptr && *a.done != 0) return;
  const int ns = nx + NU, nc = ns + 1;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;

  extern __shared__ float lds[];
  float *Vt = lds;                // [nx][nc]  (column ns = v)
  float *Ft = Vt + nx * nc;       // [nx][nc]
  float *Qt = Ft + nx * nc;       // [ns][nc]
  float *Wt = Qt + ns * nc;       // [nx][nc]
  float *Kt = Wt + nx * nc;       // [NU][nc]
  float *Rt = Kt + NU * nc;       // [NU][nc]
  float *fac_s = Rt + NU * nc;    // [NU][NU]  LU of the last free-set Hessian
  float *kt_s = fac_s + NU * NU;  // [NU]      k_t
  int *piv_s = reinterpret_cast<int *>(kt_s + NU);  // [NU]
  int *free_s = piv_s + NU;                         // [NU]
  const bool col = lane < nc;

  float kprev[NU];  // lane 0: warm start of the next (earlier) timestep's QP
#pragma unroll
  for (int m = 0; m < NU; ++m) kprev[m] = 0.f;
  int n_total = 0, info_bits = 0;
  QpTermination term;          // batch-coupled termination: every lane runs the QP (the grid-wide OR needs them all)
  term.slots = a.sync;
  term.n_blocks = gridDim.x;
  const bool coupled = a.sync != nullptr;

  for (int e = lane; e < nx * nc; e += 64) Vt[e] = 0.f;
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    const float *Cp = a.C + tb * ns * ns;
    for (int e = lane; e < ns * ns; e += 64) Qt[(e / ns) * nc + (e % ns)] = Cp[e];
    if (t < T - 1) {
      const float *Fp = a.F + tb * nx * ns;
      for (int e = lane; e < nx * ns; e += 64) Ft[(e / ns) * nc + (e % ns)] = Fp[e];
      for (int i = lane; i < nx; i += 64) Ft[i * nc + ns] = a.f ? a.f[tb * nx + i] : 0.f;
    }
    if (a.states != nullptr) {   // [x_t; u_t] into the (still unused) first row of W
      for (int j = lane; j < ns; j += 64) Wt[j] = j < nx ? a.states[tb * nx + j] : a.controls[tb * NU + (j - nx)];
    }
    __syncthreads();
    for (int i = lane; i < ns; i += 64) {
      float ci = a.c[tb * ns + i];
      if (a.states != nullptr) {   // need_expand inside the sweep: c_hat = C [x_t; u_t] + c, from the LDS copy of C   :305-317
#pragma unroll 8
        for (int j = 0; j < ns; ++j) ci = fmaf(Qt[i * nc + j], Wt[j], ci);
      }
      Qt[i * nc + ns] = ci;
    }
    __syncthreads();
    if (t < T - 1) {  // Q~ = C~ + F^T (V F~ + v e_aff)                                   mpc_step.py:110,116
      if (col) {
        for (int i = 0; i < nx; ++i) {
          float acc = (lane == ns) ? Vt[i * nc + ns] : 0.f;
#pragma unroll 8
          for (int k = 0; k < nx; ++k) acc = fmaf(Vt[i * nc + k], Ft[k * nc + lane], acc);
          Wt[i * nc + lane] = acc;
        }
      }
      __syncthreads();
      if (col) {
        for (int i = 0; i < ns; ++i) {
          float acc = Qt[i * nc + lane];
#pragma unroll 8
          for (int k = 0; k < nx; ++k) acc = fmaf(Ft[k * nc + i], Wt[k * nc + lane], acc);
          Qt[i * nc + lane] = acc;
        }
      }
      __syncthreads();
    }
    if (lane == 0 || coupled) {  // k_t: box QP on (Quu, qu), warm-started from the later timestep   :119-146
      float H[NU][NU], q[NU], lo[NU], hi[NU], kt[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
#pragma unroll
        for (int l = 0; l < NU; ++l) H[m][l] = Qt[(nx + m) * nc + nx + l];
        q[m] = Qt[(nx + m) * nc + ns];
        const float uc = a.controls[tb * NU + m];
        lo[m] = a.lower[tb * NU + m] - uc;                                              // :136-138
        hi[m] = a.upper[tb * NU + m] - uc;
        kt[m] = kprev[m];
      }
      PnqpResult<NU> qp;
      pnqp_solve<NU>(H, q, lo, hi, kt, /*warm=*/t != T - 1, a.n_qp_iter, qp, term);
      n_total += 1 + qp.it;
      if (!qp.converged) info_bits |= 4;
#pragma unroll
      for (int m = 0; m < NU; ++m) kprev[m] = kt[m];
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          kt_s[m] = kt[m];
          piv_s[m] = qp.piv[m];
          free_s[m] = qp.free_[m] ? 1 : 0;
#pragma unroll
          for (int l = 0; l < NU; ++l) fac_s[m * NU + l] = qp.fac[m][l];
        }
      }
    }
    __syncthreads();
    if (col) {  // K_t = -LU_free^-1 Qux with the rows of clamped controls zeroed                    :147-157
      float A[NU][NU], Kc[NU];
      int pv[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        pv[m] = piv_s[m];
        Kc[m] = free_s[m] ? Qt[(nx + m) * nc + lane] : 0.f;
#pragma unroll
        for (int l = 0; l < NU; ++l) A[m][l] = fac_s[m * NU + l];
      }
      if constexpr (NU == 1) {
        Kc[0] = -((1.0f / A[0][0]) * Kc[0]);
      } else {
        lu_solve_inplace<NU>(A, pv, Kc);
#pragma unroll
        for (int m = 0; m < NU; ++m) Kc[m] = -Kc[m];
      }
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        if (lane == ns) Kc[m] = kt_s[m];  // the affine column carries k_t
        Kt[m * nc + lane] = Kc[m];
        if (lane == ns) a.ks[tb * NU + m] = Kc[m];
        else if (lane < nx) a.Ks[(tb * NU + m) * nx + lane] = Kc[m];
      }
      // R = Qu. + Quu K~ from the UNMASKED blocks                                                  :165-166
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        float acc = Qt[(nx + m) * nc + lane];
#pragma unroll
        for (int l = 0; l < NU; ++l) acc = fmaf(Qt[(nx + m) * nc + nx + l], Kc[l], acc);
        Rt[m * nc + lane] = acc;
      }
    }
    __syncthreads();
    if (col && t > 0) {
#pragma unroll 4
      for (int i = 0; i < nx; ++i) {
        float acc = Qt[i * nc + lane];
#pragma unroll
        for (int m = 0; m < NU; ++m) acc = fmaf(Qt[i * nc + nx + m], Kt[m * nc + lane], acc);
#pragma unroll
        for (int m = 0; m < NU; ++m) acc = fmaf(Kt[m * nc + i], Rt[m * nc + lane], acc);
        Vt[i * nc + lane] = acc;
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    a.n_qp_total[b] = n_total;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

constexpr size_t mpc_generic_fwd_lds_bytes(int nx, int nu) { return (size_t)(nx + 2 * (nx + nu)) * 4; }

// forward_rec under a true LinDx / QuadCost: lanes m < nu evaluate the clamped feedback law, lanes i < ns one row of
// the cost each, lanes i < nx one row of the dynamics; per-trajectory backtracking as in mpc_forward_rec_kernel.
__global__ __launch_bounds__(64) void mpc_generic_forward_kernel(const MpcFwdArgs a, const int nx, const int nu) {
  if (a.done != nullptr && *a.done != 0) return;
  const int ns = nx + nu;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  extern __shared__ float lds[];
  float *xh = lds;         // [nx]  candidate state
  float *tau = xh + nx;    // [ns]  [new_x_t ; new_u_t]
  float *tau0 = tau + ns;  // [ns]  the iterate the step started from

  float alpha = 1.0f, cost = 0.f, old_cost = 0.f;
  int n_pass = 0;
  bool worse = true;
  while (worse && n_pass < a.ls_cap) {                                                   // mpc_step.py:196
    if (lane < nx) xh[lane] = a.states[(size_t)b * nx + lane];                           // :198
    __syncthreads();
    cost = 0.f;
    float delta = 0.f;   // current_cost - OLD_COST, per timestep and without cancellation (see mpc_forward_rec_kernel)
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      if (lane < nu) {
        const int m = lane;
        const float *Kr = a.Ks + (tb * nu + m) * nx;
        float v = alpha * a.ks[tb * nu + m];
#pragma unroll 8
        for (int i = 0; i < nx; ++i) v = fmaf(Kr[i], xh[i] - a.states[tb * nx + i], v);
        v += a.controls[tb * nu + m];                                                    // :209-219
        const float lb = a.lower[tb * nu + m], ub = a.upper[tb * nu + m];
        v = fminf(fmaxf(v, lb), ub);                                                     // :221
        v = (v - lb <= bound_tol(lb)) ? lb : v;
        v = (ub - v <= bound_tol(ub)) ? ub : v;
        tau[nx + m] = v;
        tau0[nx + m] = a.controls[tb * nu + m];
        a.u[tb * nu + m] = v;
        if (a.u_first != nullptr && n_pass == 0) a.u_first[tb * nu + m] = v;             // :260-263
      }
      if (lane < nx) {
        tau[lane] = xh[lane];
        tau0[lane] = a.states[tb * nx + lane];
        a.x[tb * nx + lane] = xh[lane];
      }
      __syncthreads();
      float part = 0.f, part0 = 0.f, partd = 0.f;
      if (lane < ns) {                                                                   // :246-251, util.py:162-198
        const float *Cr = a.C + (tb * ns + lane) * ns;
        float qi = 0.f, q0 = 0.f, qd = 0.f;
#pragma unroll 8
        for (int j = 0; j < ns; ++j) {   // (unrolled: the row's loads go out together instead of one per FMA)
          const float cij = Cr[j];
          qi = fmaf(cij, tau[j], qi);
          q0 = fmaf(cij, tau0[j], q0);
          qd = fmaf(cij, tau[j] - tau0[j], qd);
        }
        const float ci = a.c[tb * ns + lane];
        const float di = tau[lane] - tau0[lane];
        part = tau[lane] * fmaf(0.5f, qi, ci);
        part0 = tau0[lane] * fmaf(0.5f, q0, ci);
        partd = fmaf(di, fmaf(0.5f, qi, ci), 0.5f * tau0[lane] * qd);
      }
      const float obj = wave_sum64(part);
      cost += obj;
      delta += wave_sum64(partd);
      if (n_pass == 0) old_cost += wave_sum64(part0);                                    // :191
      if (a.objs != nullptr && lane == 0) a.objs[tb] = obj;
      float xn = 0.f;
      if (lane < nx && t < T - 1) {                                                      // :229-236
        const float *Fr = a.F + (tb * nx + lane) * ns;
        xn = a.f != nullptr ? a.f[tb * nx + lane] : 0.f;
#pragma unroll 8
        for (int j = 0; j < ns; ++j) xn = fmaf(Fr[j], tau[j], xn);
      }
      __syncthreads();
      if (lane < nx && t < T - 1) xh[lane] = xn;
      __syncthreads();
    }
    ++n_pass;
    worse = delta > 0.f;                 // :266  current_cost > OLD_COST
    if (worse) alpha *= a.ls_decay;      // :268
  }
  int info_bits = 0;
  if (worse) {                           // cap hit: the reference would still be looping; :274
    alpha /= a.ls_decay;
    info_bits |= 8;
  }
  if (!is_finite(cost)) info_bits |= 2;
  if (lane == 0) {
    a.costs[b] = cost;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

// ---- the same forward pass with its inputs staged through LDS (round 4).  mpc_generic_forward_kernel reads a row of C, F and K
// per lane straight from HBM - 4-byte accesses a row apart, every one a dependent trip - and spends 3.5 ms per call at (20,6),
// B = 4096, T = 50.  Here a timestep's blocks [C | c | F | f | K | k | u | lower | upper | x] travel to a two-slot LDS ring by
// LDS-DMA as the contiguous runs they are (dma_run_floats: any 4-byte alignment, run-time lengths), a step ahead of the
// wavefront that reads its rows from the slot.  The arithmetic, its order and the search's decisions are
// mpc_generic_forward_kernel's (and through it mpc_forward_rec_kernel's).  One wavefront per trajectory and workgroup.
struct MpcStagedFwdSlot {
  int C, c, F, f, K, k, u, lo, hi, x, floats;
};
__host__ __device__ inline MpcStagedFwdSlot mpc_staged_fwd_slot(int nx, int nu) {
  const int ns = nx + nu;
  MpcStagedFwdSlot s;
  int o = 0;
  auto region = [&](int n) { const int at = o; o += (n + 63) / 64 * 64; return at; };
  s.C = region(ns * ns); s.c = region(ns); s.F = region(nx * ns); s.f = region(nx); s.K = region(nu * nx); s.k = region(nu);
  s.u = region(nu); s.lo = region(nu); s.hi = region(nu); s.x = region(nx);
  s.floats = o;
  return s;
}
inline size_t mpc_staged_fwd_lds_bytes(int nx, int nu) { return (size_t)(2 * mpc_staged_fwd_slot(nx, nu).floats + 3 * 64) * 4; }

__global__ __launch_bounds__(64) void mpc_staged_forward_kernel(const MpcFwdArgs a, const int nx, const int nu) {
  if (a.done != nullptr && *a.done != 0) return;
  const int ns = nx + nu;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const MpcStagedFwdSlot L = mpc_staged_fwd_slot(nx, nu);
  extern __shared__ float lds[];
  float *xh = lds + 2 * L.floats;   // [64] candidate state
  float *tau = xh + 64;             // [64] [new_x_t ; new_u_t]
  float *tau0 = tau + 64;           // [64] the iterate the step started from
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char *)lds);
  auto issue = [&](int t, int slot) {
    const size_t tb = (size_t)t * B + b;
    const unsigned dst = ring_addr + (unsigned)(slot * L.floats) * 4u;
    dma_run_floats(a.C + tb * ns * ns, dst + L.C * 4, ns * ns, lane);
    dma_run_floats(a.c + tb * ns, dst + L.c * 4, ns, lane);
    if (t < T - 1) {   // uniform; there is no F_{T-1}
      dma_run_floats(a.F + tb * nx * ns, dst + L.F * 4, nx * ns, lane);
      if (has_f) dma_run_floats(a.f + tb * nx, dst + L.f * 4, nx, lane);
    }
    dma_run_floats(a.Ks + tb * nu * nx, dst + L.K * 4, nu * nx, lane);
    dma_run_floats(a.ks + tb * nu, dst + L.k * 4, nu, lane);
    dma_run_floats(a.controls + tb * nu, dst + L.u * 4, nu, lane);
    dma_run_floats(a.lower + tb * nu, dst + L.lo * 4, nu, lane);
    dma_run_floats(a.upper + tb * nu, dst + L.hi * 4, nu, lane);
    dma_run_floats(a.states + tb * nx, dst + L.x * 4, nx, lane);
  };

  float alpha = 1.0f, cost = 0.f, old_cost = 0.f;
  int n_pass = 0;
  bool worse = true;
  while (worse && n_pass < a.ls_cap) {                                                   // mpc_step.py:196
    issue(0, 0);
    if (lane < nx) xh[lane] = a.states[(size_t)b * nx + lane];                           // :198
    cost = 0.f;
    float delta = 0.f;   // current_cost - OLD_COST, per timestep and without cancellation (see mpc_forward_rec_kernel)
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      // the slot of this step has landed (requested a step ago), the other one has been read: it takes the next step's
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 1 < T) issue(t + 1, (t + 1) & 1);
      const float *S = lds + (t & 1) * L.floats;
      if (lane < nu) {
        const int m = lane;
        const float *Kr = S + L.K + m * nx;
        float v = alpha * S[L.k + m];
#pragma unroll 8
        for (int i = 0; i < nx; ++i) v = fmaf(Kr[i], xh[i] - S[L.x + i], v);
        const float u0 = S[L.u + m];
        v += u0;                                                                         // :209-219
        const float lb = S[L.lo + m], ub = S[L.hi + m];
        v = fminf(fmaxf(v, lb), ub);                                                     // :221
        v = (v - lb <= bound_tol(lb)) ? lb : v;
        v = (ub - v <= bound_tol(ub)) ? ub : v;
        tau[nx + m] = v;
        tau0[nx + m] = u0;
        a.u[tb * nu + m] = v;
        if (a.u_first != nullptr && n_pass == 0) a.u_first[tb * nu + m] = v;             // :260-263
      }
      if (lane < nx) {
        const float xl = xh[lane];
        tau[lane] = xl;
        tau0[lane] = S[L.x + lane];
        a.x[tb * nx + lane] = xl;
      }
      __syncthreads();
      float part = 0.f, part0 = 0.f, partd = 0.f;
      if (lane < ns) {                                                                   // :246-251, util.py:162-198
        const float *Cr = S + L.C + lane * ns;
        float qi = 0.f, q0 = 0.f, qd = 0.f;
#pragma unroll 8
        for (int j = 0; j < ns; ++j) {
          const float cij = Cr[j];
          qi = fmaf(cij, tau[j], qi);
          q0 = fmaf(cij, tau0[j], q0);
          qd = fmaf(cij, tau[j] - tau0[j], qd);
        }
        const float ci = S[L.c + lane];
        const float di = tau[lane] - tau0[lane];
        part = tau[lane] * fmaf(0.5f, qi, ci);
        part0 = tau0[lane] * fmaf(0.5f, q0, ci);
        partd = fmaf(di, fmaf(0.5f, qi, ci), 0.5f * tau0[lane] * qd);
      }
      const float obj = wave_sum64(part);
      cost += obj;
      delta += wave_sum64(partd);
      if (n_pass == 0) old_cost += wave_sum64(part0);                                    // :191
      if (a.objs != nullptr && lane == 0) a.objs[tb] = obj;
      float xn = 0.f;
      if (lane < nx && t < T - 1) {                                                      // :229-236
        const float *Fr = S + L.F + lane * ns;
        xn = has_f ? S[L.f + lane] : 0.f;
#pragma unroll 8
        for (int j = 0; j < ns; ++j) xn = fmaf(Fr[j], tau[j], xn);
      }
      __syncthreads();
      if (lane < nx && t < T - 1) xh[lane] = xn;
    }
    ++n_pass;
    worse = delta > 0.f;                 // :266  current_cost > OLD_COST
    if (worse) alpha *= a.ls_decay;      // :268
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                     // (the next pass refills slot 0 and rewrites xh)
  }
  int info_bits = 0;
  if (worse) {                           // cap hit: the reference would still be looping; :274
    alpha /= a.ls_decay;
    info_bits |= 8;
  }
  if (!is_finite(cost)) info_bits |= 2;
  if (lane == 0) {
    a.costs[b] = cost;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

}  // namespace dmpc
