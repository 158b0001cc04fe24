// lqr_kernels.hpp - fused time-varying LQR solve for gfx950: backward Riccati sweep + forward
// rollout in one launch, one lane group per trajectory (see colwise.hpp for the layout).
//
// Follows lqr/lqr_recursion.py:69-209 (LqrRecursion) and, with MASKED, the clamped-control
// variant mpc/active_constrained_lqr.py:67-202 (LQR_active) of the reference.
//
// Per timestep t (descending) a group holds, one column per lane:
//     Q~ = [C_t | c_t] + F~^T (V~ F~)         F~ = [F_t | f_t],  V~ = [V | v]
//     K~ = -Quu^-1 [Qux | Quu | qu]           (in-register LU, every lane its own column)
//     V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~)   (lqr_recursion.py:151-152, all terms kept)
// K~ goes to LDS (or to the Ks/ks arrays in HBM when T is too long for LDS); the forward
// sweep re-reads F_t one ROW per lane so that x_{t+1} = F_t [x;u] + f_t lands in lane i with
// nx + nu broadcast-FMAs and no reduction.
#pragma once
#include "colwise.hpp"

namespace dmpc {

struct LqrArgs {
  int T, B;
  const float *C, *c, *F, *f, *x_init;
  const uint8_t *mask;  // [T,B,nu] or nullptr
  float *Ks, *ks;       // gains requested by the caller ([T,B,nu,nx], [T,B,nu]) or nullptr
  float *wsK, *wsk;     // caller workspace used for the gains when they do not fit in LDS
  float *x, *u;
  int32_t *info;
};

enum LqrMode { kSolve = 0, kBackwardOnly = 1, kForwardOnly = 2 };

template <int N>
__device__ __forceinline__ void load_contig(const float *__restrict__ p, float (&dst)[N]) {
  // p is 4*N-byte aligned relative to a 16-byte aligned array base
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const float4 v = reinterpret_cast<const float4 *>(p)[i];
      dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
  } else if constexpr (N % 2 == 0) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float2 v = reinterpret_cast<const float2 *>(p)[i];
      dst[2 * i] = v.x; dst[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = p[i];
  }
}

// NX, NU: state / control dims.  L: lanes per trajectory (16 or 64).  MASKED: LQR_active.
// MODE: fused solve, gains only, or rollout only.  K_LDS: gains handed to the forward sweep
// through LDS (else through args.Ks/args.ks in HBM).
template <int NX, int NU, int L, bool MASKED, int MODE, bool K_LDS>
__global__ __launch_bounds__(256) void lqr_kernel(const LqrArgs a) {
  constexpr int NS = NX + NU;
  static_assert(NS + 1 <= L, "a trajectory's augmented columns must fit its lane group");
  constexpr int GPB = 256 / L;  // trajectories per 256-thread workgroup
  constexpr int KROW = NX + 1;  // [K_m | k_m]
  using G = Group<L>;

  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = blockIdx.x * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;  // keep the whole wave in lock step; only stores are suppressed
  const int T = a.T;
  const size_t B = (size_t)a.B;

  extern __shared__ float lds[];
  float *kl = lds + (size_t)grp * T * NU * KROW;  // this trajectory's gains [T][NU][KROW]

  const bool col_mat = lane < NS;   // lane owns a matrix column
  const bool col_aff = lane == NS;  // lane owns the affine column
  const int kidx = col_aff ? NX : lane;  // position inside a [K_m | k_m] row
  int info_bits = 0;

  if constexpr (MODE != kForwardOnly) {
    // ------------------------------------------------------------ backward Riccati sweep
    float V[NX];  // [V | v] columns; lanes NX..NS-1 carry junk that is never broadcast
#pragma unroll
    for (int i = 0; i < NX; ++i) V[i] = 0.f;

    for (int t = T - 1; t >= 0; --t) {
      const size_t tb = (size_t)t * B + b;
      float Q[NS];  // [C_t | c_t] column, then Q~
      {
        const float *Cp = a.C + tb * NS * NS + lane;
#pragma unroll
        for (int i = 0; i < NS; ++i) Q[i] = col_mat ? Cp[i * NS] : 0.f;
        if (col_aff) load_contig<NS>(a.c + tb * NS, Q);
      }
      if (t < T - 1) {
        float Fc[NX];  // [F_t | f_t] column
        const float *Fp = a.F + tb * NX * NS + lane;
#pragma unroll
        for (int k = 0; k < NX; ++k) Fc[k] = col_mat ? Fp[k * NS] : 0.f;
        if (col_aff && a.f != nullptr) load_contig<NX>(a.f + tb * NX, Fc);
        // W~ = V F~ (+ v in the affine column)     lqr_recursion.py:89,96: (F^T V) F, (F^T V) f + F^T v
        float W[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) W[i] = col_aff ? V[i] : 0.f;
        static_for<0, NX>([&](auto k) {
#pragma unroll
          for (int i = 0; i < NX; ++i) W[i] = fmaf(G::template bcast<k.value>(V[i]), Fc[k.value], W[i]);
        });
        // Q~ += F^T W~
        static_for<0, NS>([&](auto i) {
#pragma unroll
          for (int k = 0; k < NX; ++k) Q[i.value] = fmaf(G::template bcast<i.value>(Fc[k]), W[k], Q[i.value]);
        });
      }
      // every lane gets the full Quu                                lqr_recursion.py:102
      float Quu[NU][NU];
      static_for<0, NU>([&](auto l) {
#pragma unroll
        for (int m = 0; m < NU; ++m) Quu[m][l.value] = G::template bcast<NX + l.value>(Q[NX + m]);
      });
      // gains: K~ = -Quu^-1 * (own column of the u-rows)
      float Kt[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) Kt[m] = Q[NX + m];
      if constexpr (MASKED) {
        // active_constrained_lqr.py:110-137: zero q_u / Qux rows of clamped controls, zero Quu
        // outside free x free, 1e-8 on the clamped diagonal.
        bool act[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) act[m] = a.mask[tb * NU + m] != 0;
        float A[NU][NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          Kt[m] = act[m] ? 0.f : Kt[m];
#pragma unroll
          for (int l = 0; l < NU; ++l) {
            float v = (act[m] || act[l]) ? 0.f : Quu[m][l];
            if (m == l) v = act[m] ? (v + 1e-8f) : v;
            A[m][l] = v;
          }
        }
        if constexpr (NU == 1) {
          Kt[0] = -((1.0f / A[0][0]) * Kt[0]);  // :131-133
          if (A[0][0] == 0.f) info_bits |= 1;
        } else {
          int piv[NU];
          if (lu_factor_inplace<NU>(A, piv)) info_bits |= 1;
          lu_solve_inplace<NU>(A, piv, Kt);
#pragma unroll
          for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
        }
      } else {
        if constexpr (NU == 1) {
          Kt[0] = -((1.0f / Quu[0][0]) * Kt[0]);  // lqr_recursion.py:112-115
          if (Quu[0][0] == 0.f) info_bits |= 1;
        } else {
          float A[NU][NU];
#pragma unroll
          for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int l = 0; l < NU; ++l) A[m][l] = Quu[m][l];
          int piv[NU];
          if (lu_factor_inplace<NU>(A, piv)) info_bits |= 1;  // reference: F.batch_inv (LU + inverse), :116-120
          lu_solve_inplace<NU>(A, piv, Kt);
#pragma unroll
          for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
        }
      }
      // hand the gains to the forward sweep / the caller
      if (lane < NX || col_aff) {
        if constexpr (K_LDS) {
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
      if (t > 0) {
        // value update, UNMASKED blocks (lqr_recursion.py:151-152; active_constrained_lqr.py:143-145)
        float R[NU];  // (Qu. + Quu K~) column
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          R[m] = Q[NX + m];
#pragma unroll
          for (int l = 0; l < NU; ++l) R[m] = fmaf(Quu[m][l], Kt[l], R[m]);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) V[i] = Q[i];
        static_for<0, NU>([&](auto m) {
#pragma unroll
          for (int i = 0; i < NX; ++i) V[i] = fmaf(G::template bcast<NX + m.value>(Q[i]), Kt[m.value], V[i]);
        });
        static_for<0, NX>([&](auto i) {
#pragma unroll
          for (int m = 0; m < NU; ++m) V[i.value] = fmaf(G::template bcast<i.value>(Kt[m]), R[m], V[i.value]);
        });
      }
    }
  }

  if constexpr (MODE != kBackwardOnly) {
    // ------------------------------------------------------------ forward rollout
    if constexpr (MODE == kSolve) {
      if constexpr (K_LDS) __syncthreads();  // gains written column-wise, read row-wise
      else __threadfence_block();
    }
    const bool row_x = lane < NX;            // lane i: row i of [F_t | f_t]  -> x_{t+1}[i]
    const bool row_u = lane >= NX && lane < NS;  // lane nx+m: row m of [K_t | k_t] -> u_t[m]
    const int m_own = row_u ? lane - NX : 0;
    float xu = row_x ? a.x_init[(size_t)b * NX + lane] : 0.f;  // lane j<nx: x[j]; lane nx+m: u[m]
    bool bad = false;
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      float M[NS + 1];  // row of [F|f] (lanes < nx) or of [K|k] (lanes nx..ns-1); M[NS] = affine term
#pragma unroll
      for (int j = 0; j <= NS; ++j) M[j] = 0.f;
      if (row_x && t < T - 1) {
        float Fr[NS];
        load_contig<NS>(a.F + (tb * NX + lane) * NS, Fr);
#pragma unroll
        for (int j = 0; j < NS; ++j) M[j] = Fr[j];
        if (a.f != nullptr) M[NS] = a.f[tb * NX + lane];
      }
      bool clamp_u = false;
      if (row_u) {
        if constexpr (MODE == kSolve && K_LDS) {
#pragma unroll
          for (int j = 0; j < NX; ++j) M[j] = kl[(t * NU + m_own) * KROW + j];
          M[NS] = kl[(t * NU + m_own) * KROW + NX];
        } else {
#pragma unroll
          for (int j = 0; j < NX; ++j) M[j] = a.Ks[(tb * NU + m_own) * NX + j];
          M[NS] = a.ks[tb * NU + m_own];
        }
        if constexpr (MASKED) clamp_u = a.mask[tb * NU + m_own] != 0;
      }
      // lanes nx+m: u = k + K x          lanes i<nx: partial x' = f + Fx x     lqr_recursion.py:177,189
      float acc = M[NS];
      static_for<0, NX>([&](auto j) { acc = fmaf(G::template bcast<j.value>(xu), M[j.value], acc); });
      if (row_u) {
        if constexpr (MASKED) acc = clamp_u ? 0.f : acc;  // :179-183
        xu = acc;
      }
      bad = bad || !is_finite(xu);
      if (live) {  // x_t from lanes < nx, u_t from lanes nx..ns-1
        if (row_x) a.x[tb * NX + lane] = xu;
        else if (row_u) a.u[tb * NU + m_own] = xu;
      }
      // x' += Fu u
      static_for<0, NU>([&](auto m) {
        acc = fmaf(G::template bcast<NX + m.value>(xu), M[NX + m.value], acc);
      });
      if (row_x) xu = acc;
    }
    if (bad) info_bits |= 2;
  }

  if (a.info != nullptr) {
    // OR the group's bits into info[b]
    if (live && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

}  // namespace dmpc
