// mpc_tiled.hpp - MPCstep.backward_rec / forward_rec and the projected-Newton box QP for ANY size (round 4).
//
// The reference has no size limit (mpc/mpc_step.py:70-286, mpc/pnqp.py:37-201 are numpy on whatever shapes they get); the
// kernels of mpc_kernels.hpp / mpc_generic.hpp keep the QP and its LU in registers (n_ctrl <= 8) and a trajectory's
// columns in one wavefront (n_state + n_ctrl + 1 <= 64), and `MPCstep` beyond that was refused.  Here, as in
// lqr_tiled.hpp: one workgroup of 256 threads per trajectory (per QP for the standalone solver), runtime dimensions, the
// matrices in a per-trajectory area of the caller's workspace, the QP's vectors in LDS, every phase spread over the
// threads by output element, `__syncthreads()` between phases.  Per-trajectory termination of the QP here; the
// batch-coupled form of the same pieces, for any size and any batch, is mpc_coupled.hpp.
// Same algorithm and operation order as pnqp_device.hpp / mpc_generic.hpp / the oracle; slow, and correct.
#pragma once
#include "lqr_tiled.hpp"
#include "mpc_kernels.hpp"
#include "pnqp_device.hpp"

namespace dmpc {

// LAPACK getf2 on an n x n matrix in (global or shared) memory, by the whole workgroup; piv 0-based.  s_p: one shared int.
__device__ __forceinline__ bool tiled_lu_factor(float *LU, int n, int *piv, int *s_p) {
  const int tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  bool singular = false;
  for (int k = 0; k < n; ++k) {
    if (tid == 0) {
      int p = k;
      float best = fabsf(LU[k * n + k]);
      for (int i = k + 1; i < n; ++i) {
        const float v = fabsf(LU[i * n + k]);
        if (v > best) { best = v; p = i; }
      }
      piv[k] = p;
      *s_p = p;
    }
    __syncthreads();
    const int p = *s_p;
    if (p != k)
      for (int c = tid; c < n; c += NT) {
        const float tmp = LU[k * n + c];
        LU[k * n + c] = LU[p * n + c];
        LU[p * n + c] = tmp;
      }
    __syncthreads();
    const float d = LU[k * n + k];
    singular = singular || (d == 0.f);
    const float r = 1.0f / d;
    for (int i = k + 1 + tid; i < n; i += NT)
      if (d != 0.f) LU[i * n + k] *= r;
    __syncthreads();
    const int rem = n - k - 1;
    for (int e = tid; e < rem * rem; e += NT) {
      const int i = k + 1 + e / rem, c = k + 1 + e % rem;
      LU[i * n + c] = fmaf(-LU[i * n + k], LU[k * n + c], LU[i * n + c]);
    }
    __syncthreads();
  }
  return singular;
}

// getrs on one right-hand side x[0..n) with stride xs, by ONE thread
__device__ __forceinline__ void tiled_lu_solve(const float *LU, const int *piv, int n, float *x, int xs) {
  for (int k = 0; k < n; ++k) {
    const int p = piv[k];
    if (p != k) {
      const float tmp = x[k * xs];
      x[k * xs] = x[p * xs];
      x[p * xs] = tmp;
    }
  }
  for (int k = 0; k < n; ++k)
    for (int i = k + 1; i < n; ++i) x[i * xs] = fmaf(-LU[i * n + k], x[k * xs], x[i * xs]);
  for (int k = n - 1; k >= 0; --k) {
    const float xk = x[k * xs] / LU[k * n + k];
    x[k * xs] = xk;
    for (int i = 0; i < k; ++i) x[i * xs] = fmaf(-LU[i * n + k], xk, x[i * xs]);
  }
}

// vectors of one QP in LDS: x, g, gf, dx, xh, d, hd, lo, hi, q (n floats each), free flags (n ints), piv (n ints)
__host__ __device__ inline size_t pnqp_tiled_lds_floats(int n) { return 12 * (size_t)n + 8; }

struct PnqpTiledOut {
  int it;
  bool converged;
};

// The vectors of one QP (pnqp_tiled_lds_floats(n) floats, in LDS or in the caller's workspace) and the pieces of a QP
// iteration on them.  Each piece is executed by the whole workgroup and ends behind a __syncthreads(); its verdict is
// workgroup-uniform.  pnqp_tiled (per-row termination) and the batch-coupled kernels of mpc_coupled.hpp are put together from
// the same pieces: same arithmetic, same order.
struct PnqpTiledVecs {
  float *x, *g, *gf, *dx, *xh, *d, *hd;
  const float *lo, *hi, *q;
  int *free_, *piv, *s_ctl;   // s_ctl: [0] pivot exchange, [1] verdict of the last piece, [2] = it, [3] = converged
  float *s_f;                 // [0] alpha, [1] != 0: this row still moves (batch-coupled search only)
  __device__ __forceinline__ PnqpTiledVecs(float *v, int n)
      : x(v), g(v + n), gf(v + 2 * n), dx(v + 3 * n), xh(v + 4 * n), d(v + 5 * n), hd(v + 6 * n), lo(v + 7 * n), hi(v + 8 * n),
        q(v + 9 * n), free_(reinterpret_cast<int *>(v + 10 * n)), piv(free_ + n), s_ctl(piv + n),
        s_f(reinterpret_cast<float *>(s_ctl + 4)) {}
};

// x_init = -H^-1 q unless warm (pnqp.py:75-83), projected on the box (:93); it = n_iter - 1, not converged
__device__ __forceinline__ void pnqp_tiled_start(const float *H, int ldh, int n, float *fac, const PnqpTiledVecs &w, bool warm,
                                                 int n_iter) {
  const int tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  if (!warm) {
    for (int e = tid; e < n * n; e += NT) fac[e] = H[(e / n) * ldh + (e % n)];
    for (int r = tid; r < n; r += NT) w.x[r] = w.q[r];
    __syncthreads();
    tiled_lu_factor(fac, n, w.piv, w.s_ctl);
    if (tid == 0) {
      tiled_lu_solve(fac, w.piv, n, w.x, 1);
      for (int r = 0; r < n; ++r) w.x[r] = -w.x[r];
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += NT) w.x[r] = fminf(fmaxf(w.x[r], w.lo[r]), w.hi[r]);   // :93
  if (tid == 0) { w.s_ctl[2] = n_iter - 1; w.s_ctl[3] = 0; }
  __syncthreads();
}

// gradient, free set, H_f = LU, Newton step dx (:98-136); true iff ||dx|| >= 1e-4 (:139-140).  alpha = 1 for the search.
__device__ __forceinline__ bool pnqp_tiled_newton(const float *H, int ldh, int n, float *fac, const PnqpTiledVecs &w) {
  const int tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  const float tol_sq = __builtin_bit_cast(float, kPnqpDxTolSqBits);
  for (int r = tid; r < n; r += NT) {   // grad = Hx + q                              :98
    float acc = w.q[r];
    for (int c = 0; c < n; ++c) acc = fmaf(H[r * ldh + c], w.x[c], acc);
    w.g[r] = acc;
    const bool cl = ((w.x[r] == w.lo[r]) && (acc > 0.f)) || ((w.x[r] == w.hi[r]) && (acc < 0.f));   // :110, exact equality
    w.gf[r] = cl ? 0.f : acc;
    w.free_[r] = cl ? 0 : 1;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += NT) {   // H_f = H on free x free, 0 elsewhere, + 1e-11 I      :124-129
    const int r = e / n, c = e % n;
    float val = (w.free_[r] && w.free_[c]) ? H[r * ldh + c] : 0.f;
    if (r == c) val += kPnqpReg;
    fac[e] = val;
  }
  for (int r = tid; r < n; r += NT) w.dx[r] = w.gf[r];
  __syncthreads();
  tiled_lu_factor(fac, n, w.piv, w.s_ctl);                                               // :136
  if (tid == 0) {
    tiled_lu_solve(fac, w.piv, n, w.dx, 1);
    float n2 = 0.f;
    for (int r = 0; r < n; ++r) {
      w.dx[r] = -w.dx[r];
      n2 = fmaf(w.dx[r], w.dx[r], n2);
    }
    const bool large = n2 >= tol_sq;                                                 // :139-140
    w.s_ctl[1] = large ? 1 : 0;
    w.s_f[0] = 1.0f;
    w.s_f[1] = large ? 1.0f : 0.0f;
  }
  __syncthreads();
  const bool large = w.s_ctl[1] != 0;
  __syncthreads();     // (s_ctl[1] is the next piece's verdict too)
  return large;
}

// one trial of the backtracking search (:172-186): xh = the projected step of length alpha, lhs = 1 + 0.5 d'Hd / g'd as in
// pnqp_device.hpp; true iff the trial FAILS (then alpha has been shortened).  `moving`: the row counts (a row of a coupled
// batch that has already converged carries GAMMA + 1e-6, :174: it passes, and keeps its alpha)
__device__ __forceinline__ bool pnqp_tiled_trial(const float *H, int ldh, int n, const PnqpTiledVecs &w, bool moving) {
  const int tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  const float alpha = w.s_f[0];
  for (int r = tid; r < n; r += NT) {
    const float xr = fminf(fmaxf(fmaf(alpha, w.dx[r], w.x[r]), w.lo[r]), w.hi[r]);         // :173
    w.xh[r] = xr;
    w.d[r] = xr - w.x[r];
  }
  __syncthreads();
  for (int r = tid; r < n; r += NT) {
    float acc = 0.f;
    for (int c = 0; c < n; ++c) acc = fmaf(H[r * ldh + c], w.d[c], acc);
    w.hd[r] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    float gd = 0.f, dHd = 0.f;
    for (int r = 0; r < n; ++r) {
      gd = fmaf(w.g[r], w.d[r], gd);
      dHd = fmaf(w.d[r], w.hd[r], dHd);
    }
    const float lhs = fmaf(0.5f * dHd, fast_rcp(gd), 1.0f);                        // :175-176
    const bool fails = moving && lhs <= kPnqpGamma;                                // false for NaN
    if (fails) w.s_f[0] = alpha * kPnqpDecay;                                      // :185-186
    w.s_ctl[1] = fails ? 1 : 0;
  }
  __syncthreads();
  const bool fails = w.s_ctl[1] != 0;
  __syncthreads();
  return fails;
}

__device__ __forceinline__ void pnqp_tiled_accept(int n, const PnqpTiledVecs &w) {     // :190
  for (int r = threadIdx.x; r < n; r += kTiledThreads) w.x[r] = w.xh[r];
  __syncthreads();
}

// Projected-Newton box QP (pnqp.py:37-201, per-row termination = pnqp_solve_rows) by the workgroup.
//   H: n x n, row stride ldh (read only).  fac: n x n scratch; on return the LU of the last free-set Hessian, piv its pivots.
//   v: LDS vectors (pnqp_tiled_lds_floats); on entry v[7n..8n) = lo, v[8n..9n) = hi, v[9n..10n) = q, v[0..n) = x (warm start).
__device__ __forceinline__ PnqpTiledOut pnqp_tiled(const float *H, int ldh, int n, float *fac, float *v, bool warm, int n_iter) {
  const PnqpTiledVecs w(v, n);
  pnqp_tiled_start(H, ldh, n, fac, w, warm, n_iter);
  for (int i = 0; i < n_iter; ++i) {
    if (!pnqp_tiled_newton(H, ldh, n, fac, w)) {                                         // :141-144
      if (threadIdx.x == 0) { w.s_ctl[2] = i; w.s_ctl[3] = 1; }
      __syncthreads();
      break;
    }
    for (int count = 0; count < kPnqpMaxLs; ++count)
      if (!pnqp_tiled_trial(H, ldh, n, w, true)) break;                                  // :172: the row passed
    pnqp_tiled_accept(n, w);
  }
  PnqpTiledOut out{w.s_ctl[2], w.s_ctl[3] != 0};
  __syncthreads();
  return out;
}

// ---- the standalone solver for n > 8: one workgroup per QP
struct PnqpTiledArgs {
  int B, n;
  const float *H, *q, *lower, *upper, *x_init;
  int n_iter;
  float *x_out, *fac;
  int32_t *piv;
  float *index_f;
  int32_t *n_iter_out, *info;
};

__global__ __launch_bounds__(kTiledThreads) void pnqp_tiled_kernel(const PnqpTiledArgs a) {
  extern __shared__ float v[];
  const int n = a.n, b = blockIdx.x, tid = threadIdx.x;
  for (int r = tid; r < n; r += kTiledThreads) {
    v[r] = a.x_init != nullptr ? a.x_init[(size_t)b * n + r] : 0.f;
    v[7 * n + r] = a.lower[(size_t)b * n + r];
    v[8 * n + r] = a.upper[(size_t)b * n + r];
    v[9 * n + r] = a.q[(size_t)b * n + r];
  }
  __syncthreads();
  float *fac = a.fac + (size_t)b * n * n;
  const PnqpTiledOut o = pnqp_tiled(a.H + (size_t)b * n * n, n, n, fac, v, a.x_init != nullptr, a.n_iter);
  const int *free_ = reinterpret_cast<const int *>(v + 10 * n);
  const int *piv = free_ + n;
  for (int r = tid; r < n; r += kTiledThreads) {
    a.x_out[(size_t)b * n + r] = v[r];
    a.index_f[(size_t)b * n + r] = free_[r] ? 1.0f : 0.0f;
    if (a.piv != nullptr) a.piv[(size_t)b * n + r] = piv[r] + 1;     // LAPACK's 1-based pivots
  }
  if (tid == 0) {
    a.n_iter_out[b] = o.it;
    if (!o.converged && a.info != nullptr) atomicOr(&a.info[b], 4);     // DMPC_INFO_QP_ITERCAP
  }
}

// ---- MPCstep.backward_rec (mpc_step.py:70-173): the tiled LQR sweep with the box QP in place of the gain solve
__host__ __device__ inline size_t mpc_tiled_scratch_floats(int nx, int nu) { return tiled_scratch_floats(nx, nu); }

// a trajectory's matrices in the workspace
struct MpcTiledMats {
  float *Vt, *Qt, *Wt, *LU, *Kt, *Rt;
  __device__ __forceinline__ MpcTiledMats(float *base, int nx, int nu) {
    const int ns = nx + nu, nc = ns + 1;
    Vt = base;
    Qt = Vt + (size_t)nx * nc;
    Wt = Qt + (size_t)ns * nc;
    LU = Wt + (size_t)nx * nc;
    Kt = LU + (size_t)nu * nu;
    Rt = Kt + (size_t)nu * nc;
  }
};

// timestep t up to the box QP: Q~_t (:110,116, with need_expand's re-centring :305-317) and the QP's data in v
// (lo, hi = bounds - u_t, q = qu; x stays the later timestep's solution: the warm start, :119-146).  tau: ns floats.
__device__ __forceinline__ void mpc_tiled_before_qp(const MpcBackArgs &a, int nx, int nu, int b, int t, const MpcTiledMats &m,
                                                    float *v, float *tau) {
  const int ns = nx + nu, nc = ns + 1, tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  const size_t B = (size_t)a.B, tb = (size_t)t * B + b;
  float *Vt = m.Vt, *Qt = m.Qt, *Wt = m.Wt;
  const float *Cp = a.C + tb * ns * ns;
  for (int e = tid; e < ns * ns; e += NT) Qt[(e / ns) * nc + (e % ns)] = Cp[e];
  if (a.states != nullptr)
    for (int j = tid; j < ns; j += NT) tau[j] = j < nx ? a.states[tb * nx + j] : a.controls[tb * nu + (j - nx)];
  __syncthreads();
  for (int i = tid; i < ns; i += NT) {
    float ci = a.c[tb * ns + i];
    if (a.states != nullptr)      // need_expand inside the sweep: c_hat = C [x_t; u_t] + c                  :305-317
      for (int j = 0; j < ns; ++j) ci = fmaf(Qt[i * nc + j], tau[j], ci);
    Qt[i * nc + ns] = ci;
  }
  __syncthreads();
  if (t < a.T - 1) {                // Q~ = C~ + F^T (V F~ + v e_aff)                                        :110,116
    const float *Fp = a.F + tb * nx * ns;
    const float *fp = a.f ? a.f + tb * nx : nullptr;
    for (int e = tid; e < nx * nc; e += NT) {
      const int i = e / nc, j = e % nc;
      float acc = (j == ns) ? Vt[i * nc + ns] : 0.f;
      if (j < ns) {
        for (int k = 0; k < nx; ++k) acc = fmaf(Vt[i * nc + k], Fp[(size_t)k * ns + j], acc);
      } else if (fp != nullptr) {
        for (int k = 0; k < nx; ++k) acc = fmaf(Vt[i * nc + k], fp[k], acc);
      }
      Wt[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < ns * nc; e += NT) {
      const int i = e / nc, j = e % nc;
      float acc = Qt[e];
      for (int k = 0; k < nx; ++k) acc = fmaf(Fp[(size_t)k * ns + i], Wt[k * nc + j], acc);
      Qt[e] = acc;
    }
    __syncthreads();
  }
  // k_t: box QP on (Quu, qu), bounds lower - u, upper - u, warm-started from the later timestep        :119-146
  for (int mm = tid; mm < nu; mm += NT) {
    const float uc = a.controls[tb * nu + mm];
    v[7 * nu + mm] = a.lower[tb * nu + mm] - uc;
    v[8 * nu + mm] = a.upper[tb * nu + mm] - uc;
    v[9 * nu + mm] = Qt[(nx + mm) * nc + ns];
  }
  __syncthreads();
}

// timestep t after the box QP (its solution and free set in v, the free-set LU in m.LU): gains (:147-157), R (:165-166),
// value function (:159-171)
__device__ __forceinline__ void mpc_tiled_after_qp(const MpcBackArgs &a, int nx, int nu, int b, int t, const MpcTiledMats &m,
                                                   const float *v) {
  const int ns = nx + nu, nc = ns + 1, tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  const size_t B = (size_t)a.B, tb = (size_t)t * B + b;
  float *Vt = m.Vt, *Qt = m.Qt, *LU = m.LU, *Kt = m.Kt, *Rt = m.Rt;
  const int *free_ = reinterpret_cast<const int *>(v + 10 * nu);
  const int *piv = free_ + nu;
  // K_t = -LU_free^-1 Qux with the rows of clamped controls zeroed; the affine column carries k_t      :147-157
  for (int j = tid; j < nc; j += NT) {
    if (j == ns) {
      for (int mm = 0; mm < nu; ++mm) Kt[mm * nc + j] = v[mm];
    } else {
      for (int mm = 0; mm < nu; ++mm) Kt[mm * nc + j] = free_[mm] ? Qt[(nx + mm) * nc + j] : 0.f;
      if (nu == 1) Kt[j] = Kt[j] / LU[0];
      else tiled_lu_solve(LU, piv, nu, Kt + j, nc);
      for (int mm = 0; mm < nu; ++mm) Kt[mm * nc + j] = -Kt[mm * nc + j];
    }
    if (j < nx || j == ns)
      for (int mm = 0; mm < nu; ++mm) {
        if (j == ns) a.ks[tb * nu + mm] = Kt[mm * nc + j];
        else a.Ks[(tb * nu + mm) * nx + j] = Kt[mm * nc + j];
      }
    for (int mm = 0; mm < nu; ++mm) {   // R = Qu. + Quu K~ from the UNMASKED blocks                             :165-166
      float acc = Qt[(nx + mm) * nc + j];
      for (int l = 0; l < nu; ++l) acc = fmaf(Qt[(nx + mm) * nc + nx + l], Kt[l * nc + j], acc);
      Rt[mm * nc + j] = acc;
    }
  }
  __syncthreads();
  if (t > 0)
    for (int e = tid; e < nx * nc; e += NT) {
      const int i = e / nc, j = e % nc;
      float acc = Qt[i * nc + j];
      for (int mm = 0; mm < nu; ++mm) acc = fmaf(Qt[i * nc + nx + mm], Kt[mm * nc + j], acc);
      for (int mm = 0; mm < nu; ++mm) acc = fmaf(Kt[mm * nc + i], Rt[mm * nc + j], acc);
      Vt[e] = acc;
    }
  __syncthreads();
}

__global__ __launch_bounds__(kTiledThreads) void mpc_tiled_backward_kernel(const MpcBackArgs a, const int nx, const int nu,
                                                                           float *scratch) {
  if (a.done != nullptr && *a.done != 0) return;
  const int ns = nx + nu, nc = ns + 1;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  constexpr int NT = kTiledThreads;
  const MpcTiledMats m(scratch + (size_t)b * mpc_tiled_scratch_floats(nx, nu), nx, nu);
  extern __shared__ float v[];                 // the QP's vectors, then [ns] tau for the re-centring
  float *tau = v + pnqp_tiled_lds_floats(nu);
  int n_total = 0, info_bits = 0;

  for (int e = tid; e < nx * nc; e += NT) m.Vt[e] = 0.f;
  for (int mm = tid; mm < nu; mm += NT) v[mm] = 0.f;     // warm start of the first QP is unused (cold)
  __syncthreads();
  for (int t = a.T - 1; t >= 0; --t) {
    mpc_tiled_before_qp(a, nx, nu, b, t, m, v, tau);
    const PnqpTiledOut qp = pnqp_tiled(m.Qt + (size_t)nx * nc + nx, nc, nu, m.LU, v, /*warm=*/t != a.T - 1, a.n_qp_iter);
    n_total += 1 + qp.it;
    if (!qp.converged) info_bits |= 4;
    mpc_tiled_after_qp(a, nx, nu, b, t, m, v);
  }
  if (tid == 0) {
    a.n_qp_total[b] = n_total;
    if (a.info != nullptr) {
      if (a.info_store) a.info[b] = info_bits;
      else if (info_bits != 0) atomicOr(&a.info[b], info_bits);
    }
  }
}

// ---- MPCstep.forward_rec under a true LinDx / QuadCost (mpc_step.py:175-286): mpc_generic_forward_kernel with the rows
// spread over 256 threads and block-wide sums
__device__ __forceinline__ float tiled_block_sum(float val, float *red) {   // red: 4 floats of LDS (one per wavefront)
  for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kTiledThreads) void mpc_tiled_forward_kernel(const MpcFwdArgs a, const int nx, const int nu) {
  if (a.done != nullptr && *a.done != 0) return;
  const int ns = nx + nu;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  constexpr int NT = kTiledThreads;
  extern __shared__ float lds[];
  float *xh = lds;          // [nx]  candidate state
  float *tau = xh + nx;     // [ns]
  float *tau0 = tau + ns;   // [ns]
  float *xn = tau0 + ns;    // [nx]  next state in the making
  float *red = xn + nx;     // [4]

  float alpha = 1.0f, cost = 0.f, old_cost = 0.f;
  int n_pass = 0;
  bool worse = true;
  while (worse && n_pass < a.ls_cap) {                                                   // mpc_step.py:196
    for (int i = tid; i < nx; i += NT) xh[i] = a.states[(size_t)b * nx + i];             // :198
    __syncthreads();
    cost = 0.f;
    float delta = 0.f;
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      for (int m = tid; m < nu; m += NT) {
        const float *Kr = a.Ks + (tb * nu + m) * nx;
        float val = alpha * a.ks[tb * nu + m];
        for (int i = 0; i < nx; ++i) val = fmaf(Kr[i], xh[i] - a.states[tb * nx + i], val);
        val += a.controls[tb * nu + m];                                                  // :209-219
        const float lb = a.lower[tb * nu + m], ub = a.upper[tb * nu + m];
        val = fminf(fmaxf(val, lb), ub);                                                 // :221
        val = (val - lb <= bound_tol(lb)) ? lb : val;
        val = (ub - val <= bound_tol(ub)) ? ub : val;
        tau[nx + m] = val;
        tau0[nx + m] = a.controls[tb * nu + m];
        a.u[tb * nu + m] = val;
        if (a.u_first != nullptr && n_pass == 0) a.u_first[tb * nu + m] = val;           // :260-263
      }
      for (int i = tid; i < nx; i += NT) {
        tau[i] = xh[i];
        tau0[i] = a.states[tb * nx + i];
        a.x[tb * nx + i] = xh[i];
      }
      __syncthreads();
      float part = 0.f, part0 = 0.f, partd = 0.f;
      for (int i = tid; i < ns; i += NT) {                                               // :246-251, util.py:162-198
        const float *Cr = a.C + (tb * ns + i) * ns;
        float qi = 0.f, q0 = 0.f, qd = 0.f;
        for (int j = 0; j < ns; ++j) {
          const float cij = Cr[j];
          qi = fmaf(cij, tau[j], qi);
          q0 = fmaf(cij, tau0[j], q0);
          qd = fmaf(cij, tau[j] - tau0[j], qd);
        }
        const float ci = a.c[tb * ns + i];
        const float di = tau[i] - tau0[i];
        part += tau[i] * fmaf(0.5f, qi, ci);
        part0 += tau0[i] * fmaf(0.5f, q0, ci);
        partd += fmaf(di, fmaf(0.5f, qi, ci), 0.5f * tau0[i] * qd);
      }
      const float obj = tiled_block_sum(part, red);
      cost += obj;
      delta += tiled_block_sum(partd, red);
      if (n_pass == 0) old_cost += tiled_block_sum(part0, red);                          // :191
      if (a.objs != nullptr && tid == 0) a.objs[tb] = obj;
      if (t < T - 1)
        for (int i = tid; i < nx; i += NT) {                                             // :229-236
          const float *Fr = a.F + (tb * nx + i) * ns;
          float acc = a.f != nullptr ? a.f[tb * nx + i] : 0.f;
          for (int j = 0; j < ns; ++j) acc = fmaf(Fr[j], tau[j], acc);
          xn[i] = acc;
        }
      __syncthreads();
      if (t < T - 1)
        for (int i = tid; i < nx; i += NT) xh[i] = xn[i];
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
  if (tid == 0) {
    a.costs[b] = cost;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr) {
      const int merged = info_bits | (a.info_in != nullptr ? a.info_in[b] : 0);
      if (merged != 0) atomicOr(&a.info[b], merged);
    }
  }
}

}  // namespace dmpc
