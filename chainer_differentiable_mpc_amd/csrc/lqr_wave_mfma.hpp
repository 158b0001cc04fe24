// lqr_wave_mfma.hpp - backward Riccati sweep for the large shapes (one wavefront per trajectory, ns + 1 <= 64,
// nx and ns multiples of 4 - (32,8) of BASELINE.json configs[4]) with every A^T B product on the matrix cores.
//
// Same recursion and column-per-lane layout as lqr_kernel<NX, NU, 64, ...> (lqr/lqr_recursion.py:69-158): lane j
// holds column j of [C|c], [F|f], [V|v]; the affine column is lane ns.  What changes is who multiplies:
//   v_mfma_f32_4x4x1_16b_f32 with cbsz:4 abid:I takes 4 lanes (block I) of the A register and all 64 lanes of the B
//   register and adds the outer product to a 4-row tile held in 4 consecutive registers:
//       D[4I + i][j] += A[lane 4I + i] * B[lane j]                        (one trajectory per wavefront)
//   i.e. it computes P^T R from column-per-lane P and R, one contraction index per instruction.  With the
//   reference's own association  Q~ = C~ + (F^T V) F~  (lqr_recursion.py:89,96) both products have that shape:
//       G  = V^^T F~        rows b = columns of [V|v] (tile 10, row 0 is the homogeneous row g1 = v^T F~)
//       Q~ += G^T F~ + g1 (x) e_aff
//   and so has K~^T (Q~u. + Quu K~) of the value update.  Qxu K~ (an A B product) gets its left factor as a row matrix XU = (F~^T G)[u rows] from the same pass.
// At (32,8) that is 288 + 330 + 64 MFMAs per timestep instead of ~4,600 readlane/FMA pairs; the HIP kernel it
// replaces spilled 1.5 KB of scratch per lane and ran at 0.03 of the HBM roof.
// The gains go to HBM (caller's Ks/ks or the workspace): at this size they do not fit in LDS; the rollout is the
// forward-only lqr_kernel, which is already bandwidth bound.
#pragma once
#include "colwise.hpp"
#include "lqr_kernels.hpp"

namespace dmpc {

typedef float f4v __attribute__((ext_vector_type(4)));

template <int I>
__device__ __forceinline__ f4v mfma_bcast(float a, float b, f4v c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, I, 0);  // cbsz = 4: block I of A feeds all 16 blocks
}

// LU with partial pivoting of a WAVE-UNIFORM matrix (every lane holds the same A - one trajectory per wavefront)
// applied to the lane's own right-hand side: the pivot row is a scalar, so the interchange (select chains, 600
// v_cndmask at n = 8 in lu_factor_inplace) sits behind a uniform branch that is taken only when a row really moves.
// LAPACK getf2/getrs order: first maximum wins, one interchange per column, scaling by the reciprocal pivot.
template <int N>
__device__ __forceinline__ bool lu_factor_solve_uniform(float (&A)[N][N], float (&x)[N]) {
  bool singular = false;
  static_for<0, N>([&](auto kc) {
    constexpr int k = kc.value;
    float best = fabsf(A[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const float v = fabsf(A[i][k]);
      const bool gt = v > best;
      best = gt ? v : best;
      p = gt ? i : p;
    }
    p = __builtin_amdgcn_readfirstlane(p);
    if (p != k) {  // uniform and rare for the well-conditioned Quu of an LQR: one branch around the select chain
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const float ak = A[k][c];
        float nk = ak;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
          const bool sw = (p == i);
          nk = sw ? A[i][c] : nk;
          A[i][c] = sw ? ak : A[i][c];
        }
        A[k][c] = nk;
      }
      const float xk = x[k];
      float nx_ = xk;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const bool sw = (p == i);
        nx_ = sw ? x[i] : nx_;
        x[i] = sw ? xk : x[i];
      }
      x[k] = nx_;
    }
    const float d = A[k][k];
    singular = singular || (d == 0.0f);
    const float r = fast_rcp(d);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const float l = A[i][k] * r;
      A[i][k] = l;
#pragma unroll
      for (int c = k + 1; c < N; ++c) A[i][c] = fmaf(-l, A[k][c], A[i][c]);
      x[i] = fmaf(-l, x[k], x[i]);  // forward substitution rides along (the interchanges of later columns
                                     // permute x and the stored multipliers together, as getrs does)
    }
    A[k][k] = r;  // keep the reciprocal pivot
  });
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    x[k] = x[k] * A[k][k];
#pragma unroll
    for (int i = 0; i < k; ++i) x[i] = fmaf(-A[i][k], x[k], x[i]);
  }
  return singular;
}

template <int NX, int NU>
__global__ __launch_bounds__(256, 2) void lqr_wave_mfma_backward(const LqrArgs a) {  // 2 waves per SIMD: <= 256 registers
  constexpr int NS = NX + NU, AFF = NS;
  static_assert(NX % 4 == 0 && NS % 4 == 0 && NS + 1 <= 64, "tiles of 4 rows, one wavefront per trajectory");
  constexpr int TX = NX / 4, TS = NS / 4, TA = AFF / 4;  // tiles: x rows, all rows, the tile whose row 0 is `aff`
  using G64 = Group<64>;

  const int lane = threadIdx.x & 63;
  int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const bool col_aff = lane == AFF;
  const int lane_c = lane < NS ? lane : NS - 1;
  const bool k_lane = lane < NX || col_aff;
  const float eaff = col_aff ? 1.f : 0.f;
  float *Ks = a.Ks != nullptr ? a.Ks : a.wsK;
  float *ks = a.Ks != nullptr ? a.ks : a.wsk;
  int info_bits = 0;

  f4v V4[TX];  // rows of [V | v]
#pragma unroll
  for (int I = 0; I < TX; ++I) V4[I] = f4v{0.f, 0.f, 0.f, 0.f};

  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    // ---- [C_t | c_t] rows, column-per-lane
    f4v Q4[TS];
    {  // one load per row through a per-lane base and stride (lane ns walks c, the others a column of C) - a
       // `col_aff ? c[i] : C[i][j]` per element makes hipcc emit an exec-masked branch diamond for every load
      const char *qp = reinterpret_cast<const char *>(col_aff ? a.c + tb * NS : a.C + tb * NS * NS + lane_c);
      const size_t qs = col_aff ? 4 : NS * 4;  // bytes to the next row (a running pointer: one 64-bit add per load)
#pragma unroll
      for (int I = 0; I < TS; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Q4[I][r] = *reinterpret_cast<const float *>(qp);
          qp += qs;
        }
    }
    // ---- XU[m] = column nx+m of Q~x. as a ROW (lane i = Q[i][nx+m]): the A operand of Qxu K~ in the value update
    constexpr bool kXU = NU % 4 == 0;
    constexpr int TU = kXU ? NU / 4 : 1;
    f4v XU4[TU];
    if constexpr (kXU) {
      float xu[NU];
      load_contig<NU>(a.C + (tb * NS + (lane < NX ? lane : NX - 1)) * NS + NX, xu);
#pragma unroll
      for (int m = 0; m < NU; ++m) XU4[m / 4][m % 4] = xu[m];
    }
    if (t < T - 1) {
      float Fc[NX];
      {
        const bool f_lane = col_aff && has_f;
        // lane ns without f: any column, zeroed below
        const char *fp = reinterpret_cast<const char *>(f_lane ? a.f + tb * NX : a.F + tb * NX * NS + lane_c);
        const size_t fs = f_lane ? 4 : NS * 4;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
          Fc[k] = *reinterpret_cast<const float *>(fp);
          fp += fs;
        }
        if (!has_f) {
#pragma unroll
          for (int k = 0; k < NX; ++k) Fc[k] = col_aff ? 0.f : Fc[k];
        }
      }
      // ---- G = [V|v]^T F~ : tiles 0..TX-1 (x columns of V) and TA (row 0 = v^T F~)
      f4v G4[TX + 1];
#pragma unroll
      for (int I = 0; I <= TX; ++I) G4[I] = f4v{0.f, 0.f, 0.f, 0.f};
      static_for<0, NX>([&](auto a_) {
        const float va = V4[a_.value / 4][a_.value % 4];
        const float fa = Fc[a_.value];
        static_for<0, TX>([&](auto I) { G4[I.value] = mfma_bcast<I.value>(va, fa, G4[I.value]); });
        G4[TX] = mfma_bcast<TA>(va, fa, G4[TX]);
      });
      // ---- Q~ += G^T F~ + g1 (x) e_aff
      static_for<0, NX>([&](auto b_) {
        const float gb = G4[b_.value / 4][b_.value % 4];
        const float fb = Fc[b_.value];
        static_for<0, TS>([&](auto I) { Q4[I.value] = mfma_bcast<I.value>(gb, fb, Q4[I.value]); });
        if constexpr (kXU)  // (F~^T G)[nx+m][i] = (G^T F~)[i][nx+m]: A = F~[b] block of the control columns, B = G[b]
          static_for<0, TU>([&](auto q) { XU4[q.value] = mfma_bcast<TX + q.value>(fb, gb, XU4[q.value]); });
      });
      {
        const float g1 = G4[TX][0];
        static_for<0, TS>([&](auto I) { Q4[I.value] = mfma_bcast<I.value>(g1, eaff, Q4[I.value]); });
      }
    }
    // ---- gains: every lane gets the full Quu (wave-uniform) and solves its own column  (:112-120)
    float Quu[NU][NU];
    static_for<0, NU>([&](auto l) {
#pragma unroll
      for (int m = 0; m < NU; ++m) Quu[m][l.value] = G64::template bcast<NX + l.value>(Q4[(NX + m) / 4][(NX + m) % 4]);
    });
    float Kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = Q4[(NX + m) / 4][(NX + m) % 4];
    {
      float A[NU][NU];
#pragma unroll
      for (int m = 0; m < NU; ++m)
#pragma unroll
        for (int l = 0; l < NU; ++l) A[m][l] = Quu[m][l];
      if (lu_factor_solve_uniform<NU>(A, Kt)) info_bits |= 1;
#pragma unroll
      for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
    }
    if (k_lane && live) {
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        if (col_aff) ks[tb * NU + m] = Kt[m];
        else Ks[(tb * NU + m) * NX + lane] = Kt[m];
      }
    }
    if (t > 0) {
      // ---- value update, all four terms (:151-152): V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~)
      float R[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        R[m] = Q4[(NX + m) / 4][(NX + m) % 4];
#pragma unroll
        for (int l = 0; l < NU; ++l) R[m] = fmaf(Quu[m][l], Kt[l], R[m]);
      }
#pragma unroll
      for (int I = 0; I < TX; ++I) V4[I] = Q4[I];
      if constexpr (kXU) {  // Qxu K~ = XU^T K~, the A^T B shape
        static_for<0, NU>([&](auto m) {
          const float xm = XU4[m.value / 4][m.value % 4];
          static_for<0, TX>([&](auto I) { V4[I.value] = mfma_bcast<I.value>(xm, Kt[m.value], V4[I.value]); });
        });
      } else {
        static_for<0, NU>([&](auto m) {  // Qxu K~ as an A B product (v_readlane broadcast of column nx+m)
#pragma unroll
          for (int I = 0; I < TX; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) V4[I][r] = fmaf(G64::template bcast<NX + m.value>(Q4[I][r]), Kt[m.value], V4[I][r]);
        });
      }
      static_for<0, NU>([&](auto m) {  // K~^T R
        static_for<0, TX>([&](auto I) { V4[I.value] = mfma_bcast<I.value>(Kt[m.value], R[m.value], V4[I.value]); });
      });
    }
  }
  if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
