// f64_row_kernels.hpp - the LQR solve and its KKT gradient at the reference's own precision, on the column-per-lane layout.
//
// The reference computes this path in float64 (lqr/differentiable_lqr.py:169-172; numpy's default everywhere).  Until round 4
// float64 here meant one LANE per trajectory with its matrices in an HBM workspace (f64_api.hip: 18 ms at the headline size,
// 570 x the float32 stream).  These kernels are lqr_kernel / costate_kernel (lqr_kernels.hpp, costate_kernels.hpp) re-stated
// in double: a trajectory per group of L lanes (L = 16: a DPP row, `v_fmac_f64_dpp row_newbcast` blocks from
// dpp_blocks_f64_gen.hpp; L = 64: a wavefront, v_readlane broadcasts), matrices in registers (64-bit pairs), inputs
// prefetched one bank ahead, gains handed to the rollout through LDS (or the Ks / ks arrays when the horizon is long).
// Same operation order as the float32 kernels: (V F~) first, then F~^T (.), LAPACK getf2 pivoting, true divisions.
//
//   lqr/lqr_recursion.py:69-209 (LqrRecursion.backward + .forward), mpc/active_constrained_lqr.py:110-145 (MASKED),
//   lqr/differentiable_lqr.py:78-142 (co-states and outer products)
#pragma once
#include "dpp_blocks_f64_gen.hpp"
#include "f64_row_blocks.hpp"

namespace dmpc {

struct F64RowSolve {
  int T, B;
  const double *C, *c, *F, *f, *x_init;   // c: [T,B,ns], or - c_u set - its state part [T,B,nx]; x_init may be null (= 0)
  const double *c_u;                       // control part of c [T,B,nu] (the second solve of the gradient), or null
  const uint8_t *mask;
  double *Ks, *ks;                         // [T,B,nu,nx], [T,B,nu]: outputs, and the hand-over to the rollout when !k_lds
  double *x, *u;
  int32_t *info;
  int k_lds;                               // gains reach the rollout through LDS (T * NU * (NX+1) doubles per trajectory fit)
};

template <int NX, int NU, int L, bool MASKED>
__global__ __launch_bounds__(256) void lqr_f64_row_kernel(const F64RowSolve a) {
  constexpr int NS = NX + NU;
  static_assert(NS + 1 <= L, "a trajectory's augmented columns must fit its lane group");
  constexpr int GPB = 256 / L;
  constexpr int KROW = NX + 1;
  using G = Group64<L>;
  using Blk = RiccatiBlocks64<NX, NU, L>;

  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = blockIdx.x * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const bool split_c = a.c_u != nullptr;

  extern __shared__ double lds64[];
  double *kl = lds64 + (size_t)grp * (a.k_lds ? T * NU * KROW : 0);

  const bool col_aff = lane == NS;
  const int lane_c = lane < NS ? lane : NS - 1;
  const bool k_lane = lane < NX || col_aff;
  const int kidx = lane < NX ? lane : NX;
  int info_bits = 0;

  // ---------------------------------------------------------------- backward Riccati sweep
  // BANKS register banks of one timestep each: with two, the loads of step t-1 are in flight while step t is computed;
  // the widest shapes ((32,8): 72 doubles per bank next to a working set of 136) keep one
  {
    constexpr int BANKS = (2 * 2 * (NS + NX) + 2 * (NS + 3 * NX + NU * NU + 8)) <= 440 ? 2 : 1;
    double Qr[BANKS][NS], Fr[BANKS][NX];   // column `lane` of [C_t | c_t] and of [F_t | f_t]
    // ONE load per element and lane: lanes < NS walk a column of C_t / F_t (stride NS), the affine lane walks c_t / f_t
    // (stride 1) - a per-lane base pointer and stride instead of a second set of registers for the affine terms
    const double *Fsafe = T > 1 ? a.F : a.C;
    const size_t cstride = col_aff ? 1 : NS;
    auto issue = [&](int t, double (&Qn)[NS], double (&Fn)[NX]) __attribute__((always_inline)) {
      t = t < 0 ? 0 : t;
      const size_t tb = (size_t)t * B + b;
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const size_t tbF = (size_t)tF * B + b;
      const double *Cp = col_aff ? (a.c + tb * (split_c ? NX : NS)) : (a.C + tb * NS * NS + lane_c);
      const double *Cu = (col_aff && split_c) ? (a.c_u + tb * NU - NX) : Cp;    // rows >= NX of a split c
#pragma unroll
      for (int i = 0; i < NS; ++i) Qn[i] = (i < NX ? Cp : Cu)[i * cstride];
      const double *Fp = (col_aff && has_f) ? (a.f + tbF * NX) : (Fsafe + tbF * NX * NS + lane_c);
      const size_t fstride = (col_aff && has_f) ? 1 : NS;
#pragma unroll
      for (int k = 0; k < NX; ++k) Fn[k] = Fp[k * fstride];
    };

    double V[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) V[i] = 0.0;

    auto step = [&](int t, const double (&Qn)[NS], const double (&Fn)[NX]) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      double Q[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) Q[i] = Qn[i];
      if (t < T - 1) {
        double Fc[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) Fc[k] = (col_aff && !has_f) ? 0.0 : Fn[k];
        double W[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) W[i] = col_aff ? V[i] : 0.0;
        Blk::vf(W, V, Fc);       // W~ = V F~ (+ v)                         lqr_recursion.py:89,96
        Blk::ftw(Q, Fc, W);      // Q~ += F~^T W~
      }
      double Quu[NU][NU];
      static_for<0, NU>([&](auto l) {
#pragma unroll
        for (int m = 0; m < NU; ++m) Quu[m][l.value] = G::template bcast<NX + l.value>(Q[NX + m]);
      });
      double Kt[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) Kt[m] = Q[NX + m];
      double A[NU][NU];
      if constexpr (MASKED) {     // active_constrained_lqr.py:110-137
        bool act[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) act[m] = a.mask[tb * NU + m] != 0;
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          Kt[m] = act[m] ? 0.0 : Kt[m];
#pragma unroll
          for (int l = 0; l < NU; ++l) {
            double v = (act[m] || act[l]) ? 0.0 : Quu[m][l];
            if (m == l) v = act[m] ? (v + 1e-8) : v;
            A[m][l] = v;
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
          for (int l = 0; l < NU; ++l) A[m][l] = Quu[m][l];
      }
      if constexpr (NU == 1) {
        if (A[0][0] == 0.0) info_bits |= 1;
        Kt[0] = -(Kt[0] / A[0][0]);                                           // lqr_recursion.py:112-115
      } else {
        int piv[NU];
        if (lu_factor_inplace64<NU>(A, piv)) info_bits |= 1;                 // :116-120 (F.batch_inv) as an LU solve
        lu_solve_inplace64<NU>(A, piv, Kt);
#pragma unroll
        for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
      }
      if (k_lane) {
        if (a.k_lds) {
#pragma unroll
          for (int m = 0; m < NU; ++m) kl[(t * NU + m) * KROW + kidx] = Kt[m];
        }
        if (live && a.Ks != nullptr) {
#pragma unroll
          for (int m = 0; m < NU; ++m) {
            if (col_aff) a.ks[tb * NU + m] = Kt[m];
            else a.Ks[(tb * NU + m) * NX + lane] = Kt[m];
          }
        }
      }
      if (t > 0) {                                                              // :151-152, all four terms
        double R[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          R[m] = Q[NX + m];
#pragma unroll
          for (int l = 0; l < NU; ++l) R[m] = fma(Quu[m][l], Kt[l], R[m]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) V[i] = Q[i];
        Blk::vupd(V, Q, Kt, R);
      }
    };

    if constexpr (BANKS == 2) {
      issue(T - 1, Qr[0], Fr[0]);
      for (int t0 = T - 1; t0 >= 0; t0 -= 2) {
        issue(t0 - 1, Qr[1], Fr[1]);
        step(t0, Qr[0], Fr[0]);
        issue(t0 - 2, Qr[0], Fr[0]);
        if (t0 - 1 >= 0) step(t0 - 1, Qr[1], Fr[1]);
      }
    } else {
      for (int t = T - 1; t >= 0; --t) {
        issue(t, Qr[0], Fr[0]);
        step(t, Qr[0], Fr[0]);
      }
    }
  }

  // ---------------------------------------------------------------- forward rollout                    :160-200
  if (a.x != nullptr) {
    if (!a.k_lds) __threadfence_block();
    const bool row_x = lane < NX;
    const int lane_x = row_x ? lane : NX - 1;
    double Fr[2][NS], fr[2], Kr[2][NU];
    bool cl[2][NU];
    const double *Fsafe = T > 1 ? a.F : a.C;
    const double *fsafe = has_f ? a.f : a.C;
    auto issue_row = [&](int t, double (&Fn)[NS], double &fn, double (&Kn)[NU], bool (&cn)[NU]) __attribute__((always_inline)) {
      t = t < T ? t : T - 1;
      const size_t tb = (size_t)t * B + b;
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const size_t tbF = (size_t)tF * B + b;
      const double *Fp = Fsafe + (tbF * NX + lane_x) * NS;
#pragma unroll
      for (int j = 0; j < NS; ++j) Fn[j] = Fp[j];
      fn = fsafe[has_f ? tbF * NX + lane_x : 0];
      if (a.k_lds) {
#pragma unroll
        for (int m = 0; m < NU; ++m) Kn[m] = kl[(t * NU + m) * KROW + kidx];
      } else {
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          const double *p = col_aff ? (a.ks + tb * NU + m) : (a.Ks + (tb * NU + m) * NX + (kidx < NX ? kidx : 0));
          Kn[m] = *p;
        }
      }
      if constexpr (MASKED) {
#pragma unroll
        for (int m = 0; m < NU; ++m) cn[m] = a.mask[tb * NU + m] != 0;
      }
    };
    double xv = row_x ? (a.x_init ? a.x_init[(size_t)b * NX + lane] : 0.0) : (col_aff ? 1.0 : 0.0);
    bool bad = false;
    auto fstep = [&](int t, const double (&Fn)[NS], const double fn, const double (&Kn)[NU], const bool (&cn)[NU]) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      double u[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        u[m] = group_sum64<L>(k_lane ? Kn[m] * xv : 0.0);                    // :177
        if constexpr (MASKED) u[m] = cn[m] ? 0.0 : u[m];
        bad = bad || !(fabs(u[m]) <= 1.7e308);
      }
      bad = bad || !(fabs(xv) <= 1.7e308);
      if (live) {
        if (row_x) a.x[tb * NX + lane] = xv;
        if (lane < NU) {
          double uo = u[0];
#pragma unroll
          for (int m = 1; m < NU; ++m) uo = (lane == m) ? u[m] : uo;
          a.u[tb * NU + lane] = uo;
        }
      }
      if (t < T - 1) {
        double acc = has_f ? fn : 0.0;
        double M[NS + 1];
#pragma unroll
        for (int j = 0; j < NS; ++j) M[j] = Fn[j];
        M[NS] = 0.0;
        Blk::dot_x(acc, xv, M);                                                 // :189, state part
#pragma unroll
        for (int m = 0; m < NU; ++m) acc = fma(Fn[NX + m], u[m], acc);
        if (row_x) xv = acc;
      }
    };
    issue_row(0, Fr[0], fr[0], Kr[0], cl[0]);
    for (int t0 = 0; t0 < T; t0 += 2) {
      issue_row(t0 + 1, Fr[1], fr[1], Kr[1], cl[1]);
      fstep(t0, Fr[0], fr[0], Kr[0], cl[0]);
      issue_row(t0 + 2, Fr[0], fr[0], Kr[0], cl[0]);
      if (t0 + 1 < T) fstep(t0 + 1, Fr[1], fr[1], Kr[1], cl[1]);
    }
    if (bad) info_bits |= 2;
  }
  if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

struct F64RowCostate {
  int T, B;
  const double *C, *c, *F, *x, *u, *dx, *du, *gx;    // gx [T,B,nx]: the affine term of the d_lambda recursion
  int strict;
  double *dx0, *dC, *dc, *dF, *df;
};

// differentiable_lqr.py:85-104 (lambda), :114-126 (d_lambda), :128-134 (dC, dc, dF, df, d_x_init); costate_kernel in double
template <int NX, int NU, int L>
__global__ __launch_bounds__(256) void costate_f64_row_kernel(const F64RowCostate a) {
  constexpr int NS = NX + NU;
  static_assert(NS <= L, "tau must fit the lane group");
  constexpr int GPB = 256 / L;
  using Blk = RiccatiBlocks64<NX, NU, L>;
  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = blockIdx.x * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool is_x = lane < NX, is_tau = lane < NS;
  const int lane_x = is_x ? lane : NX - 1;
  const int lane_t = is_tau ? lane : NS - 1;
  const double wa = 0.5, wb = a.strict ? 0.5 : 1.0;
  const int df_shift = a.strict ? 1 : 0;
  double lam = 0.0, dlam = 0.0;

  struct Slot {
    double tau, dtau, ci, ri;
    double Crow[NS], Fcol[NX];
  };
  auto load = [&](int t, Slot &s) __attribute__((always_inline)) {
    t = t < 0 ? 0 : t;
    const size_t tb = (size_t)t * B + b;
    const double *tp = lane_t < NX ? a.x + tb * NX + lane_t : a.u + tb * NU + (lane_t - NX);
    const double *dp = lane_t < NX ? a.dx + tb * NX + lane_t : a.du + tb * NU + (lane_t - NX);
    s.tau = *tp;
    s.dtau = *dp;
    const double *Cp = a.C + (tb * NS + lane_x) * NS;
#pragma unroll
    for (int j = 0; j < NS; ++j) s.Crow[j] = Cp[j];
    s.ci = a.c[tb * NS + lane_x];
    s.ri = a.gx[tb * NX + lane_x];
    const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
    const double *Fp = (T > 1 ? a.F : a.C) + ((size_t)tF * B + b) * NX * NS + lane_x;
#pragma unroll
    for (int k = 0; k < NX; ++k) s.Fcol[k] = Fp[k * NS];
  };
  auto step = [&](int t, const Slot &s) __attribute__((always_inline)) {
    const size_t tb = (size_t)t * B + b;
    const double tau = s.tau, dtau = s.dtau;
    if (t < T - 1) {
      if (a.dF != nullptr) {
        double row[NS];
        Blk::outer2(row, tau, dtau, dlam, lam);                                // dlam (x) tau + lam (x) dtau      :130-131
        if (live && is_x) {
#pragma unroll
          for (int j = 0; j < NS; ++j) a.dF[(tb * NX + lane) * NS + j] = row[j];
        }
      }
      if (a.df != nullptr && df_shift == 1 && live && is_x) a.df[tb * NX + lane] = dlam;
    }
    if (a.dC != nullptr) {
      double row[NS];
      Blk::outer2(row, tau, dtau, wa * dtau, wb * tau);                        // :128 (and its symmetric variant)
      if (live && is_tau) {
#pragma unroll
        for (int j = 0; j < NS; ++j) a.dC[(tb * NS + lane) * NS + j] = row[j];
      }
    }
    if (a.dc != nullptr && live && is_tau) a.dc[tb * NS + lane] = dtau;
    double nl = s.ci, ndl = s.ri;
    Blk::dots2_ns(nl, ndl, s.Crow, tau, dtau);                                 // :92,102 / :115,124
    if (t < T - 1) Blk::dots2_nx(nl, ndl, s.Fcol, lam, dlam);
    lam = nl;
    dlam = ndl;
    if (a.df != nullptr && df_shift == 0 && t < T - 1 && live && is_x) a.df[tb * NX + lane] = dlam;   // the reference's index, :133
  };
  Slot sa, sb;
  load(T - 1, sa);
  for (int t = T - 1; t >= 0; t -= 2) {
    load(t - 1, sb);
    step(t, sa);
    load(t - 2, sa);
    if (t - 1 >= 0) step(t - 1, sb);
  }
  if (a.dx0 != nullptr && live && is_x) a.dx0[(size_t)b * NX + lane] = dlam;
}

}  // namespace dmpc
