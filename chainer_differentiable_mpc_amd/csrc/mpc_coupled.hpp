// mpc_coupled.hpp - the reference's OWN termination of the projected-Newton box QP - "does any row of the batch still move",
// "has any row of the batch passed the Armijo test" (mpc/pnqp.py:139-144, 172, 187) - for ANY size and ANY batch (round 5).
//
// The register kernels take those two tests as grid-wide ORs at grid barriers (pnqp_device.hpp: QpTermination), which needs
// every row of the batch resident at once: a cooperative launch that the runtime refuses beyond ~4 K trajectories (MPCstep) and
// that the sizes of mpc_tiled.hpp (more than 8 controls / 64 columns) had no form of.  Here the barrier stays and the
// residency requirement goes: a FIXED grid that always fits (as many workgroups as the device holds, at most one per row), every
// workgroup walking its share of the rows piece by piece - start, Newton step, search trial, accept: the pieces of
// mpc_tiled.hpp - with each row's vectors and matrices kept in the caller's workspace between pieces, and ONE grid-wide OR per
// decision.  Runtime dimensions, a workgroup per row at a time: slow (as mpc_tiled.hpp), any n, any B, same decisions.
#pragma once
#include "mpc_tiled.hpp"

namespace dmpc {

// per row of the standalone solver: the QP's vectors
__host__ __device__ inline size_t pnqp_coupled_row_floats(int n) { return pnqp_tiled_lds_floats(n); }
// per trajectory of MPCstep.backward_rec: the tiled sweep's matrices, the QP's vectors, tau, {n_qp_total, info}
__host__ __device__ inline size_t mpc_coupled_traj_floats(int nx, int nu) {
  return mpc_tiled_scratch_floats(nx, nu) + pnqp_tiled_lds_floats(nu) + (size_t)(nx + nu) + 4;
}

// One box QP per row, batch-coupled: rows b = blockIdx.x, + gridDim.x, ... of this workgroup; `each(b, fn)` hands fn the row's
// (H, ldh, fac, vectors).  Returns the reference's `i` (grid-uniform) and whether the batch converged.
template <class Rows>
__device__ __forceinline__ PnqpTiledOut pnqp_coupled_rows(const Rows &rows, int n, int n_iter, QpTermination &term) {
  PnqpTiledOut out{n_iter - 1, false};
  for (int i = 0; i < n_iter; ++i) {
    bool moving = false;
    rows([&](const float *H, int ldh, float *fac, const PnqpTiledVecs &w) { moving = pnqp_tiled_newton(H, ldh, n, fac, w) || moving; });
    if (!term.any(moving)) {                                           // :141-144: no row of the batch still moves
      out.it = i;
      out.converged = true;
      break;
    }
    for (int count = 0; count < kPnqpMaxLs; ++count) {                 // :172
      bool passed = false;
      rows([&](const float *H, int ldh, float *, const PnqpTiledVecs &w) {
        const bool row_moves = w.s_f[1] != 0.0f;
        passed = !pnqp_tiled_trial(H, ldh, n, w, row_moves) || passed;
      });
      if (term.any(passed)) break;                                     // :172,187: the search ends once ANY row passes
    }
    rows([&](const float *, int, float *, const PnqpTiledVecs &w) { pnqp_tiled_accept(n, w); });   // :190, every row
  }
  return out;
}

// ---- the standalone solver (dmpc_pnqp, batch_coupled): vecs = [B][pnqp_coupled_row_floats(n)], slots zeroed
__global__ __launch_bounds__(kTiledThreads) void pnqp_coupled_kernel(const PnqpTiledArgs a, float *vecs, unsigned *slots) {
  const int n = a.n, tid = threadIdx.x;
  const size_t nv = pnqp_coupled_row_floats(n);
  QpTermination term;
  term.slots = slots;
  term.n_blocks = gridDim.x;
  auto rows = [&](auto fn) {
    for (int b = blockIdx.x; b < a.B; b += gridDim.x)
      fn(a.H + (size_t)b * n * n, n, a.fac + (size_t)b * n * n, PnqpTiledVecs(vecs + (size_t)b * nv, n));
  };
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    float *v = vecs + (size_t)b * nv;
    for (int r = tid; r < n; r += kTiledThreads) {
      v[r] = a.x_init != nullptr ? a.x_init[(size_t)b * n + r] : 0.f;
      v[7 * n + r] = a.lower[(size_t)b * n + r];
      v[8 * n + r] = a.upper[(size_t)b * n + r];
      v[9 * n + r] = a.q[(size_t)b * n + r];
    }
    __syncthreads();
    pnqp_tiled_start(a.H + (size_t)b * n * n, n, n, a.fac + (size_t)b * n * n, PnqpTiledVecs(v, n), a.x_init != nullptr, a.n_iter);
  }
  const PnqpTiledOut o = pnqp_coupled_rows(rows, n, a.n_iter, term);
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const PnqpTiledVecs w(vecs + (size_t)b * nv, n);
    for (int r = tid; r < n; r += kTiledThreads) {
      a.x_out[(size_t)b * n + r] = w.x[r];
      a.index_f[(size_t)b * n + r] = w.free_[r] ? 1.0f : 0.0f;
      if (a.piv != nullptr) a.piv[(size_t)b * n + r] = w.piv[r] + 1;     // LAPACK's 1-based pivots
    }
    if (tid == 0) {
      a.n_iter_out[b] = o.it;
      if (!o.converged && a.info != nullptr) atomicOr(&a.info[b], 4);     // DMPC_INFO_QP_ITERCAP
    }
  }
}

// ---- MPCstep.backward_rec (mpc_step.py:70-173), batch-coupled: scratch = [B][mpc_coupled_traj_floats(nx, nu)], a.sync zeroed
__global__ __launch_bounds__(kTiledThreads) void mpc_coupled_backward_kernel(const MpcBackArgs a, const int nx, const int nu,
                                                                             float *scratch) {
  if (a.done != nullptr && *a.done != 0) return;   // grid-uniform, before any barrier
  const int ns = nx + nu, nc = ns + 1, tid = threadIdx.x;
  constexpr int NT = kTiledThreads;
  const size_t per = mpc_coupled_traj_floats(nx, nu), nm = mpc_tiled_scratch_floats(nx, nu), nv = pnqp_tiled_lds_floats(nu);
  QpTermination term;
  term.slots = a.sync;
  term.n_blocks = gridDim.x;
  auto vecs_of = [&](int b) { return scratch + (size_t)b * per + nm; };
  auto rows = [&](auto fn) {
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
      const MpcTiledMats m(scratch + (size_t)b * per, nx, nu);
      fn(m.Qt + (size_t)nx * nc + nx, nc, m.LU, PnqpTiledVecs(vecs_of(b), nu));
    }
  };
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const MpcTiledMats m(scratch + (size_t)b * per, nx, nu);
    float *v = vecs_of(b);
    for (int e = tid; e < nx * nc; e += NT) m.Vt[e] = 0.f;
    for (int mm = tid; mm < nu; mm += NT) v[mm] = 0.f;
  }
  __syncthreads();
  int n_total = 0, info_bits = 0;     // (grid-uniform: every trajectory's QP runs the batch's number of iterations)
  for (int t = a.T - 1; t >= 0; --t) {
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
      const MpcTiledMats m(scratch + (size_t)b * per, nx, nu);
      float *v = vecs_of(b);
      mpc_tiled_before_qp(a, nx, nu, b, t, m, v, v + nv);
      pnqp_tiled_start(m.Qt + (size_t)nx * nc + nx, nc, nu, m.LU, PnqpTiledVecs(v, nu), /*warm=*/t != a.T - 1, a.n_qp_iter);
    }
    const PnqpTiledOut qp = pnqp_coupled_rows(rows, nu, a.n_qp_iter, term);
    n_total += 1 + qp.it;
    if (!qp.converged) info_bits |= 4;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x)
      mpc_tiled_after_qp(a, nx, nu, b, t, MpcTiledMats(scratch + (size_t)b * per, nx, nu), vecs_of(b));
  }
  if (tid == 0)
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
      a.n_qp_total[b] = n_total;
      if (a.info != nullptr) {
        if (a.info_store) a.info[b] = info_bits;
        else if (info_bits != 0) atomicOr(&a.info[b], info_bits);
      }
    }
}

}  // namespace dmpc
