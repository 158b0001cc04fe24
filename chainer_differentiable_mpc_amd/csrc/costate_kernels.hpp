// costate_kernels.hpp - co-state (lambda, d_lambda) backward sweeps and the outer products of the
// analytic KKT gradient, one lane group per trajectory.
//
// Follows DiffLqr.backward, lqr/differentiable_lqr.py:85-104 (lambda), :114-126 (d_lambda), :128-134
// (dC, dc, dF, df, d_x_init), and MPCstep.backward, mpc/mpc_step.py:383-446, which is the same
// computation with the opposite output sign, a symmetric dC and df = -d_lambda[1:].
//
// Layout per group of L lanes (colwise.hpp):
//   lane j < NS : tau_t[j], dtau_t[j]                 (element per lane)
//   lane i < NX : row i of C_t (contiguous in HBM), c_t[i], column i of F_t[:, :NX], lambda[i], dlambda[i]
//   lambda_t[i]  = c_t[i] + sum_j C_t[i][j] tau_t[j] + sum_k F_t[k][i] lambda_{t+1}[k]
//   dC_t row i (lane i < NS), dF_t row k (lane k < NX) are built with broadcast-multiplies and stored as
//   contiguous rows.
#pragma once
#include "colwise.hpp"
#include "costate_args.hpp"
#include "dpp_blocks_gen.hpp"
#include "lqr_kernels.hpp"
#include "riccati_blocks.hpp"

namespace dmpc {

template <int N>
__device__ __forceinline__ void store_contig(float *__restrict__ p, const float (&src)[N]) {
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i)
      reinterpret_cast<float4 *>(p)[i] = make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
  } else if constexpr (N % 2 == 0) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) reinterpret_cast<float2 *>(p)[i] = make_float2(src[2 * i], src[2 * i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = src[i];
  }
}

// PAD: container for a smaller problem (a.nx_log <= NX, a.nu_log <= NU; see lqr_kernel<..., PAD>): element i of tau lives in
// lane i (state) or NX + m (control), everything outside the problem is 0, rows are stored element by element at the
// problem's own strides.
template <int NX, int NU, int L, bool PAD = false>
__global__ __launch_bounds__(256) void costate_kernel(const CostateArgs a) {
  constexpr int NS = NX + NU;
  static_assert(NS <= L, "tau must fit the lane group");
  constexpr int GPB = 256 / L;
  using G = Group<L>;
  using Blk = RiccatiBlocks<NX, NU, L>;  // fused DPP broadcast-FMA blocks for the 16-lane shapes (dpp_blocks_gen.hpp)

  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = blockIdx.x * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;

  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int i) -> int { return i < NX ? (i < nx ? i : -1) : (i - NX < nu ? nx + (i - NX) : -1); };
  const int lrow = lane < NS ? logical(lane) : -1;
  const bool is_x = lane < nx;
  const bool is_tau = PAD ? lrow >= 0 : lane < NS;
  const int lane_x = is_x ? lane : nx - 1;    // clamped: rows/columns re-read by the idle lanes, never used
  const int lane_t = PAD ? (lrow >= 0 ? lrow : ns - 1) : (is_tau ? lane : NS - 1);   // index into [x; u]
  const float wa = 0.5f, wb = a.dC_mode == 0 ? 1.0f : 0.5f;

  float lam = 0.f, dlam = 0.f;  // lambda_{t+1}[lane], d_lambda_{t+1}[lane]  (lanes < NX)

  // Inputs of one timestep.  With one wavefront per SIMD nothing else hides HBM latency, so the loads of step
  // t-2 are issued before step t is computed (three banks, statically rotated - hipcc drains vmcnt at a loop
  // header, so the prefetch has to sit in the same iteration as the compute it overlaps).
  struct Slot {
    float tau, dtau, ci, ri;
    float Crow[NS], Fcol[NX];
  };
  auto load = [&](int t, Slot &s) __attribute__((always_inline)) {
    t = t < 0 ? 0 : t;  // the prefetch past t = 0 re-reads step 0 (never consumed)
    const size_t tb = (size_t)t * B + b;
    // (pointer selects, then ONE load each: `cond ? x[i] : u[j]` makes hipcc emit an exec-masked branch per load)
    const float *tp = lane_t < nx ? a.x + tb * nx + lane_t : a.u + tb * nu + (lane_t - nx);
    const float *dp = lane_t < nx ? a.dx + tb * nx + lane_t : a.du + tb * nu + (lane_t - nx);
    s.tau = *tp;
    s.dtau = *dp;
    if constexpr (PAD) {   // clamped addresses; step() discards what lies outside the problem
      const float *Cp = a.C + (tb * ns + lane_x) * ns;
      static_for<0, NS>([&](auto j) {
        const int lj = logical(j.value);
        s.Crow[j.value] = Cp[lj >= 0 ? lj : 0];
      });
      s.ci = a.c[tb * ns + lane_x];
      s.ri = a.r[tb * (a.r_cols ? a.r_cols : ns) + lane_x];
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const float *Fp = (T > 1 ? a.F : a.C) + ((size_t)tF * B + b) * nx * ns + lane_x;
      static_for<0, NX>([&](auto k) { s.Fcol[k.value] = Fp[(k.value < nx ? k.value : 0) * ns]; });
      return;
    }
    load_contig<NS>(a.C + (tb * NS + lane_x) * NS, s.Crow);
    s.ci = a.c[tb * NS + lane_x];
    s.ri = a.r[tb * (a.r_cols ? a.r_cols : NS) + lane_x];
    const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);  // there is no F_{T-1}
    const float *Fp = a.F + ((size_t)tF * B + b) * NX * NS + lane_x;  // column lane_x of F_t[:, :NX]
#pragma unroll
    for (int k = 0; k < NX; ++k) s.Fcol[k] = Fp[k * NS];
  };
  auto step = [&](int t, const Slot &s_in) __attribute__((always_inline)) {
    const size_t tb = (size_t)t * B + b;
    Slot sp;
    if constexpr (PAD) {
      sp = s_in;
      sp.tau = is_tau ? sp.tau : 0.f;
      sp.dtau = is_tau ? sp.dtau : 0.f;
      static_for<0, NS>([&](auto j) { sp.Crow[j.value] = logical(j.value) >= 0 ? sp.Crow[j.value] : 0.f; });
      static_for<0, NX>([&](auto k) { sp.Fcol[k.value] = k.value < nx ? sp.Fcol[k.value] : 0.f; });
    }
    const Slot &s = PAD ? sp : s_in;
    const float tau = s.tau, dtau = s.dtau;
    auto store_row = [&](float *row_base, const float (&row)[NS]) __attribute__((always_inline)) {   // PAD: the problem's columns of a row
      static_for<0, NS>([&](auto j) {
        const int lj = logical(j.value);
        if (lj >= 0) row_base[lj] = row[j.value];
      });
    };
    // ---- dF_t and df (they use lambda_{t+1}, d_lambda_{t+1})             differentiable_lqr.py:130-133
    if (t < T - 1) {
      if (a.dF != nullptr) {
        float row[NS];  // out_sign * (dlam (x) tau + lam (x) dtau), row `lane`
        Blk::outer2(row, tau, dtau, a.out_sign * dlam, a.out_sign * lam);
        if constexpr (PAD) { if (live && is_x) store_row(a.dF + (tb * nx + lane) * ns, row); }
        else if (live && is_x) store_contig<NS>(a.dF + (tb * NX + lane) * NS, row);
      }
      if (a.df != nullptr && a.df_shift == 1 && live && is_x) a.df[tb * nx + lane] = a.out_sign * dlam;
    }
    // ---- dC_t, dc_t                                                          :128-129
    if (a.dC != nullptr) {
      float row[NS];  // out_sign * (wa dtau (x) tau + wb tau (x) dtau), row `lane`
      Blk::outer2(row, tau, dtau, a.out_sign * wa * dtau, a.out_sign * wb * tau);
      if constexpr (PAD) { if (live && is_tau) store_row(a.dC + (tb * ns + lane_t) * ns, row); }
      else if (live && is_tau) store_contig<NS>(a.dC + (tb * NS + lane) * NS, row);
    }
    if (a.dc != nullptr && live && is_tau) a.dc[tb * ns + lane_t] = a.out_sign * dtau;

    // ---- lambda_t, d_lambda_t                                                 :92,102 / :115,124
    float nl = s.ci, ndl = a.r_sign * s.ri;
    Blk::dots2_ns(nl, ndl, s.Crow, tau, dtau);
    if (t < T - 1) Blk::dots2_nx(nl, ndl, s.Fcol, lam, dlam);
    lam = (!PAD || is_x) ? nl : 0.f;
    dlam = (!PAD || is_x) ? ndl : 0.f;
    if (a.df != nullptr && a.df_shift == 0 && t < T - 1 && live && is_x) a.df[tb * nx + lane] = a.out_sign * dlam;
  };

  Slot sa, sb, sc;  // two steps of loads in flight
  load(T - 1, sa);
  load(T - 2, sb);
  for (int t = T - 1; t >= 0; t -= 3) {
    load(t - 2, sc);
    step(t, sa);
    if (t - 1 >= 0) {
      load(t - 3, sa);
      step(t - 1, sb);
    }
    if (t - 2 >= 0) {
      load(t - 4, sb);
      step(t - 2, sc);
    }
  }
  if (a.dx0 != nullptr && live && is_x) a.dx0[(size_t)b * nx + lane] = a.out_sign * dlam;
}

// Runtime-dimension version: one wavefront per trajectory, vectors in LDS (completeness path).
struct CostateDims {
  int nx, nu;
};

__global__ __launch_bounds__(64) void costate_generic_kernel(const CostateArgs a, const CostateDims d) {
  const int nx = d.nx, nu = d.nu, ns = nx + nu;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  extern __shared__ float lds[];
  float *tau = lds, *dtau = tau + ns, *lam = dtau + ns, *dlam = lam + nx, *nlam = dlam + nx, *ndlam = nlam + nx;
  const float wa = 0.5f, wb = a.dC_mode == 0 ? 1.0f : 0.5f;
  for (int i = lane; i < nx; i += 64) { lam[i] = 0.f; dlam[i] = 0.f; }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    for (int j = lane; j < ns; j += 64) {
      tau[j] = j < nx ? a.x[tb * nx + j] : a.u[tb * nu + (j - nx)];
      dtau[j] = j < nx ? a.dx[tb * nx + j] : a.du[tb * nu + (j - nx)];
    }
    __syncthreads();
    if (t < T - 1) {
      if (a.dF != nullptr)
        for (int e = lane; e < nx * ns; e += 64) {
          const int k = e / ns, j = e % ns;
          a.dF[tb * nx * ns + e] = a.out_sign * fmaf(dlam[k], tau[j], lam[k] * dtau[j]);
        }
      if (a.df != nullptr && a.df_shift == 1)
        for (int i = lane; i < nx; i += 64) a.df[tb * nx + i] = a.out_sign * dlam[i];
    }
    if (a.dC != nullptr)
      for (int e = lane; e < ns * ns; e += 64) {
        const int i = e / ns, j = e % ns;
        a.dC[tb * ns * ns + e] = a.out_sign * fmaf(wa * dtau[i], tau[j], (wb * tau[i]) * dtau[j]);
      }
    if (a.dc != nullptr)
      for (int j = lane; j < ns; j += 64) a.dc[tb * ns + j] = a.out_sign * dtau[j];
    for (int i = lane; i < nx; i += 64) {
      float nl = a.c[tb * ns + i], ndl = a.r_sign * a.r[tb * (a.r_cols ? a.r_cols : ns) + i];
      const float *Cr = a.C + (tb * ns + i) * ns;
      for (int j = 0; j < ns; ++j) {
        nl = fmaf(Cr[j], tau[j], nl);
        ndl = fmaf(Cr[j], dtau[j], ndl);
      }
      if (t < T - 1) {
        const float *Fp = a.F + tb * nx * ns + i;
        for (int k = 0; k < nx; ++k) {
          nl = fmaf(Fp[k * ns], lam[k], nl);
          ndl = fmaf(Fp[k * ns], dlam[k], ndl);
        }
      }
      nlam[i] = nl;
      ndlam[i] = ndl;
    }
    __syncthreads();
    for (int i = lane; i < nx; i += 64) {
      lam[i] = nlam[i];
      dlam[i] = ndlam[i];
      if (a.df != nullptr && a.df_shift == 0 && t < T - 1) a.df[tb * nx + i] = a.out_sign * dlam[i];
    }
    __syncthreads();
  }
  if (a.dx0 != nullptr)
    for (int i = lane; i < nx; i += 64) a.dx0[(size_t)b * nx + i] = a.out_sign * dlam[i];
}

}  // namespace dmpc
