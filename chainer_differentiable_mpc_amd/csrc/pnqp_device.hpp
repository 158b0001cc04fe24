// pnqp_device.hpp - projected-Newton box QP in registers (device function).
//
//     min 1/2 x'Hx + q'x   s.t.  lo <= x <= hi            (Tassa, Mansard, Todorov 2014, Alg. 1)
//
// Follows PNQP, mpc/pnqp.py:37-201 of the reference.  The reference tests convergence and the Armijo
// condition over the whole batch (pnqp.py:139-144, 172, 187), which makes a row's result depend on its
// batch-mates.  Two termination modes (`QpTermination`):
//   * per row (default; what shards across GPUs): the reference called with a batch of one per row - pinned by
//     the per-row golden vectors (tests/golden/pnqp_n*.npz, *_row_* keys) and oracle.pnqp(batch_coupled=False);
//   * batch coupled: the reference's own semantics for an unsharded batch.  Both tests become ONE any-reduction
//     over the grid each ("is any row still moving", "did any row pass / is any row already converged"), done
//     with a grid barrier on a fresh slot per decision; the kernel must then be launched cooperatively (all
//     workgroups resident).  Pinned by the batched golden vectors and oracle.pnqp(batch_coupled=True).
//
// Everything is float32 (the reference computes in float64 but rounds every solve to float32,
// util.py:522-527).  The routine is executed redundantly by every lane that needs the result - inside
// the MPC-step kernel all lanes of a trajectory's group run it on identical data.
#pragma once
#include "colwise.hpp"

namespace dmpc {

constexpr float kPnqpGamma = 0.1f;   // pnqp.py:23
constexpr float kPnqpDecay = 0.1f;   // pnqp.py:163
constexpr float kPnqpDxTol = 1e-4f;  // pnqp.py:140
constexpr float kPnqpReg = 1e-11f;   // pnqp.py:73
constexpr int kPnqpMaxLs = 10;       // pnqp.py:172

// Where PNQP's two termination tests are reduced.  `slots == nullptr`: per row.  Otherwise a grid-wide OR: slot k
// (two dwords {arrivals, flag}, zeroed before the launch, never reused) serves the k-th decision of the launch; every
// thread of every workgroup must call any() the same number of times - true because all exits of the coupled
// algorithm are taken on grid-uniform values.
struct QpTermination {
  unsigned *slots = nullptr;
  unsigned n_blocks = 0;
  unsigned seq = 0;

  __device__ __forceinline__ bool any(bool v) {
    if (slots == nullptr) return v;
    __shared__ int flag;
    const int block_any = __syncthreads_or(v ? 1 : 0);
    if (threadIdx.x == 0) {
      unsigned *s = slots + 2 * (size_t)seq;
      if (block_any) __hip_atomic_fetch_or(&s[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&s[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&s[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < n_blocks) __builtin_amdgcn_s_sleep(2);
      flag = (int)__hip_atomic_load(&s[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // the next any() writes `flag` only behind its own __syncthreads_or: no second barrier needed
    ++seq;
    return flag != 0;
  }
};

// decisions one PNQP call can take: per QP iteration the convergence test and up to kPnqpMaxLs Armijo passes
__host__ __device__ constexpr size_t pnqp_sync_slots(int n_iter) { return (size_t)n_iter * (1 + 10); }

template <int N>
struct PnqpResult {
  float fac[N][N];  // LU of the last free-set Hessian H_f (N == 1: H_f itself)      pnqp.py:144,201
  float rinv[N];    // reciprocal pivots of that LU (N == 1: 1 / H_f): solves multiply, an IEEE divide is ~10 issue slots
  int piv[N];       // LAPACK 1-based pivots of that factorisation
  bool free_[N];    // Index_f
  int it;           // the reference's returned `i`
  bool converged;
};

template <int N>
__device__ __forceinline__ float pnqp_obj(const float (&H)[N][N], const float (&q)[N], const float (&x)[N]) {
  float quad = 0.f, lin = 0.f;  // 0.5 * x'Hx + q'x   (pnqp.py:26-33)
#pragma unroll
  for (int r = 0; r < N; ++r) {
    float hx = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c) hx = fmaf(H[r][c], x[c], hx);
    quad = fmaf(x[r], hx, quad);
    lin = fmaf(q[r], x[r], lin);
  }
  return fmaf(0.5f, quad, lin);
}

// sqrtf(n2) >= kPnqpDxTol  <=>  n2 >= kPnqpDxTolSq: sqrtf is correctly rounded and monotonic, and this is the smallest
// float32 whose root reaches the tolerance (0x322BCC76 = 9.99999905e-09; its predecessor's root is below 1e-4f).
// Same decisions, NaN included (both comparisons are false), without the ~20 instructions of an IEEE square root.
constexpr unsigned kPnqpDxTolSqBits = 0x322BCC76u;

template <int N, bool UNIFORM = false>
__device__ __forceinline__ void pnqp_cold_start(const float (&H)[N][N], const float (&q)[N], float (&x)[N]) {
  if constexpr (N == 1) {  // x_init = -H^-1 q                                        pnqp.py:75-83
    x[0] = -fast_rcp(H[0][0]) * q[0];
  } else {
    float A[N][N], ri[N];
    int piv[N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) A[r][c] = H[r][c];
    lu_factor_rinv<N, UNIFORM>(A, piv, ri);
#pragma unroll
    for (int r = 0; r < N; ++r) x[r] = q[r];
    lu_solve_rinv<N, UNIFORM>(A, piv, ri, x);
#pragma unroll
    for (int r = 0; r < N; ++r) x[r] = -x[r];
  }
}

// Per-row termination, written for a wavefront whose lanes carry different rows (or the same row redundantly): both
// loops are WAVE-UNIFORM (they run while any lane still needs them, decided by one ballot and a scalar branch) and the
// lanes that are finished keep their state by construction instead of being masked off -
//   * a row that has converged keeps its x, so every later pass recomputes the same gradient, free set and
//     factorisation from it: `res` ends up as the pass that converged left it, with no per-field selects;
//   * a row whose Armijo test has passed keeps its alpha, so every later trial recomputes the same candidate; the
//     trial counter is common to all rows, so rows that stop on the cap stop together with the loop.
// hipcc structurises the data-dependent loops of the straightforward form below (kept for the batch-coupled mode) into
// nested exec-mask bookkeeping that costs more than the arithmetic: 42-47 % of mpc_backward_rec's time went there
// (scripts/microbench/mpc_phases.hip).  Same arithmetic, same order, same results as that form with slots == nullptr.
// UNIFORM: all lanes of the wavefront run the SAME problem (lqr_wave_mfma_backward<..., MPC>): see lu_factor_rinv.
template <int N, bool UNIFORM = false>
__device__ __forceinline__ void pnqp_solve_rows(const float (&H)[N][N], const float (&q)[N], const float (&lo)[N],
                                                const float (&hi)[N], float (&x)[N], bool warm, int n_iter,
                                                PnqpResult<N> &res) {
  if (!warm) pnqp_cold_start<N, UNIFORM>(H, q, x);
#pragma unroll
  for (int r = 0; r < N; ++r) x[r] = fminf(fmaxf(x[r], lo[r]), hi[r]);  // :93
  res.converged = false;
  res.it = n_iter - 1;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    res.free_[r] = true;
    res.piv[r] = r + 1;
    res.rinv[r] = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c) res.fac[r][c] = 0.f;
  }
  const float tol_sq = __builtin_bit_cast(float, kPnqpDxTolSqBits);
  bool done = false;
  for (int i = 0; i < n_iter; ++i) {
    float g[N], gf[N], dx[N];
    bool clampd[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {  // grad = Hx + q                                    :98
      float acc = q[r];
#pragma unroll
      for (int c = 0; c < N; ++c) acc = fmaf(H[r][c], x[c], acc);
      g[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {  // exact float equality, as the reference           :110
      clampd[r] = ((x[r] == lo[r]) && (g[r] > 0.f)) || ((x[r] == hi[r]) && (g[r] < 0.f));
      gf[r] = clampd[r] ? 0.f : g[r];
      res.free_[r] = !clampd[r];
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) {  // H_f = H on free x free, 0 elsewhere, + 1e-11 I     :124-129
        float v = (clampd[r] || clampd[c]) ? 0.f : H[r][c];
        if (r == c) v += kPnqpReg;
        res.fac[r][c] = v;
      }
    if constexpr (N == 1) {
      res.rinv[0] = fast_rcp(res.fac[0][0]);
      dx[0] = -res.rinv[0] * gf[0];  // :134
    } else {
      lu_factor_rinv<N, UNIFORM>(res.fac, res.piv, res.rinv);  // :136
#pragma unroll
      for (int r = 0; r < N; ++r) dx[r] = gf[r];
      lu_solve_rinv<N, UNIFORM>(res.fac, res.piv, res.rinv, dx);
#pragma unroll
      for (int r = 0; r < N; ++r) dx[r] = -dx[r];
    }
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < N; ++r) n2 = fmaf(dx[r], dx[r], n2);
    const bool large = n2 >= tol_sq;               // :139-140 (a NaN norm counts as converged there too)
    if (!done && !large) {                         // :141-144: this row no longer moves
      res.it = i;
      res.converged = true;
    }
    done = done || !large;
    if (__builtin_amdgcn_ballot_w64(!done) == 0) return;
    // backtracking line search (:162-190), lhs as in the form below
    float alpha = 1.0f;
    float xh[N];
    bool searching = !done;
    int count = 0;
    do {
      float d[N];
      float gd = 0.f, dHd = 0.f;
#pragma unroll
      for (int r = 0; r < N; ++r) {
        xh[r] = fminf(fmaxf(fmaf(alpha, dx[r], x[r]), lo[r]), hi[r]);  // :173
        d[r] = xh[r] - x[r];
        gd = fmaf(g[r], d[r], gd);
      }
#pragma unroll
      for (int r = 0; r < N; ++r) {
        float hd = 0.f;
#pragma unroll
        for (int c = 0; c < N; ++c) hd = fmaf(H[r][c], d[c], hd);
        dHd = fmaf(d[r], hd, dHd);
      }
      const float lhs = fmaf(0.5f * dHd, fast_rcp(gd), 1.0f);   // :175-176
      const bool fails = searching && lhs <= kPnqpGamma;        // false for NaN, like numpy's max(nan) <= GAMMA
      alpha = fails ? alpha * kPnqpDecay : alpha;               // :185-186
      ++count;
      searching = fails && count < kPnqpMaxLs;                  // :172
    } while (__builtin_amdgcn_ballot_w64(searching) != 0);
#pragma unroll
    for (int r = 0; r < N; ++r) x[r] = done ? x[r] : xh[r];    // :190
  }
}

// x: in = warm start (if warm) ; out = solution.
template <int N>
__device__ __forceinline__ void pnqp_solve(const float (&H)[N][N], const float (&q)[N], const float (&lo)[N],
                                           const float (&hi)[N], float (&x)[N], bool warm, int n_iter,
                                           PnqpResult<N> &res, QpTermination &term) {
  if (term.slots == nullptr) {   // uniform
    pnqp_solve_rows<N>(H, q, lo, hi, x, warm, n_iter, res);
    return;
  }
  if (!warm) {  // x_init = -H^-1 q                                                   pnqp.py:75-83
    if constexpr (N == 1) {
      x[0] = -fast_rcp(H[0][0]) * q[0];
    } else {
      float A[N][N], ri[N];
      int piv[N];
#pragma unroll
      for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) A[r][c] = H[r][c];
      lu_factor_rinv<N>(A, piv, ri);
#pragma unroll
      for (int r = 0; r < N; ++r) x[r] = q[r];
      lu_solve_rinv<N>(A, piv, ri, x);
#pragma unroll
      for (int r = 0; r < N; ++r) x[r] = -x[r];
    }
  }
#pragma unroll
  for (int r = 0; r < N; ++r) x[r] = fminf(fmaxf(x[r], lo[r]), hi[r]);  // :93
  res.converged = false;
  res.it = n_iter - 1;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    res.free_[r] = true;
    res.piv[r] = r + 1;
    res.rinv[r] = 0.f;
#pragma unroll
    for (int c = 0; c < N; ++c) res.fac[r][c] = 0.f;
  }
  for (int i = 0; i < n_iter; ++i) {
    float g[N], gf[N], dx[N];
    bool clampd[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {  // grad = Hx + q                                    :98
      float acc = q[r];
#pragma unroll
      for (int c = 0; c < N; ++c) acc = fmaf(H[r][c], x[c], acc);
      g[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {  // exact float equality, as the reference           :110
      clampd[r] = ((x[r] == lo[r]) && (g[r] > 0.f)) || ((x[r] == hi[r]) && (g[r] < 0.f));
      gf[r] = clampd[r] ? 0.f : g[r];
      res.free_[r] = !clampd[r];
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) {  // H_f = H on free x free, 0 elsewhere, + 1e-11 I     :124-129
        float v = (clampd[r] || clampd[c]) ? 0.f : H[r][c];
        if (r == c) v += kPnqpReg;
        res.fac[r][c] = v;
      }
    if constexpr (N == 1) {
      res.rinv[0] = fast_rcp(res.fac[0][0]);
      dx[0] = -res.rinv[0] * gf[0];  // :134
    } else {
      lu_factor_rinv<N>(res.fac, res.piv, res.rinv);  // :136
#pragma unroll
      for (int r = 0; r < N; ++r) dx[r] = gf[r];
      lu_solve_rinv<N>(res.fac, res.piv, res.rinv, dx);
#pragma unroll
      for (int r = 0; r < N; ++r) dx[r] = -dx[r];
    }
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < N; ++r) n2 = fmaf(dx[r], dx[r], n2);
    const bool large = sqrtf(n2) >= kPnqpDxTol;  // :139-140 (a NaN norm counts as converged there too)
    if (!term.any(large)) {                      // :141-144: no row (of the batch / this row) still moves
      res.it = i;
      res.converged = true;
      return;
    }
    // backtracking line search                                                         :162-190
    // lhs = (J(x) - J(xh)) / (g'(x - xh)).  J is quadratic, so with d = xh - x this is exactly
    // 1 + 0.5 * d'Hd / g'd; evaluating it that way avoids the float32 cancellation in J(x) - J(xh) that
    // otherwise makes the Armijo test random near the optimum (the reference evaluates it in float64).
    float alpha = 1.0f;
    float xh[N];
    int count = 0;
    bool again;
    do {
      float d[N];
      float gd = 0.f, dHd = 0.f;
#pragma unroll
      for (int r = 0; r < N; ++r) {
        xh[r] = fminf(fmaxf(fmaf(alpha, dx[r], x[r]), lo[r]), hi[r]);  // :173
        d[r] = xh[r] - x[r];
        gd = fmaf(g[r], d[r], gd);
      }
#pragma unroll
      for (int r = 0; r < N; ++r) {
        float hd = 0.f;
#pragma unroll
        for (int c = 0; c < N; ++c) hd = fmaf(H[r][c], d[c], hd);
        dHd = fmaf(d[r], hd, dHd);
      }
      const float lhs = fmaf(0.5f * dHd, fast_rcp(gd), 1.0f);   // :175-176 (gd = 0 -> NaN or inf, as the quotient would be)
      // a row that has already converged carries GAMMA + 1e-6 (:174): it passes, and keeps its alpha
      const bool fails = large && lhs <= kPnqpGamma;        // false for NaN, like numpy's max(nan) <= GAMMA
      if (fails) alpha *= kPnqpDecay;                       // :185-186
      ++count;
      again = !term.any(!fails);                            // :172,187: the search ends once ANY row passes
    } while (again && count < kPnqpMaxLs);                  // :172
#pragma unroll
    for (int r = 0; r < N; ++r) x[r] = xh[r];               // :190
  }
}

template <int N>
__device__ __forceinline__ void pnqp_solve(const float (&H)[N][N], const float (&q)[N], const float (&lo)[N],
                                           const float (&hi)[N], float (&x)[N], bool warm, int n_iter,
                                           PnqpResult<N> &res) {
  QpTermination per_row;
  pnqp_solve<N>(H, q, lo, hi, x, warm, n_iter, res, per_row);
}

}  // namespace dmpc
