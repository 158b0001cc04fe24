// f64_row_blocks.hpp - the column-per-lane primitives of colwise.hpp / riccati_blocks.hpp in float64.
//
// Same layout (register i of lane j holds M[i][j], a trajectory per group of L lanes), same products; a double is a
// 64-bit register pair, so a broadcast is two 32-bit DPP moves (L = 16) or two v_readlane (L = 64).  For the 16-lane
// shapes dpp_blocks_f64_gen.hpp (gen_dpp_blocks_f64.py) specialises `RiccatiBlocks64` with `v_fmac_f64_dpp ...
// row_newbcast:k` - gfx90a+ folds the broadcast into the double-precision FMA as well, one issue slot per FMA.
#pragma once
#include "colwise.hpp"

namespace dmpc {

template <int L>
struct Group64;

template <>
struct Group64<16> {
  template <int K>
  static __device__ __forceinline__ double bcast(double v) {
    static_assert(K >= 0 && K < 16, "row_newbcast lane out of range");
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x150 + K, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + K, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
  }
};

template <>
struct Group64<64> {
  template <int K>
  static __device__ __forceinline__ double bcast(double v) {
    static_assert(K >= 0 && K < 64, "readlane lane out of range");
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), K);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), K);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
  }
};

template <int L>
__device__ __forceinline__ double group_sum64(double v);

template <>
__device__ __forceinline__ double group_sum64<16>(double v) {
#define DMPC_ROR_ADD64(N)                                                                                        \
  {                                                                                                              \
    const long long b = __builtin_bit_cast(long long, v);                                                        \
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), 0x120 + N, 0xf, 0xf, true);           \
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 + N, 0xf, 0xf, true);                    \
    v += __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);                            \
  }
  DMPC_ROR_ADD64(8)
  DMPC_ROR_ADD64(4)
  DMPC_ROR_ADD64(2)
  DMPC_ROR_ADD64(1)
#undef DMPC_ROR_ADD64
  return v;
}

template <>
__device__ __forceinline__ double group_sum64<64>(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LAPACK getf2 / getrs in registers, float64 (colwise.hpp: lu_factor_inplace / lu_solve_inplace): first maximum wins, one
// row interchange per column, true divisions - the oracle's (and LAPACK's) operation order
template <int N>
__device__ __forceinline__ bool lu_factor_inplace64(double (&A)[N][N], int (&piv)[N]) {
  bool singular = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double best = fabs(A[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double v = fabs(A[i][k]);
      const bool gt = v > best;
      best = gt ? v : best;
      p = gt ? i : p;
    }
    piv[k] = p + 1;
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const double ak = A[k][c];
      double nk = ak;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const bool s = (p == i);
        nk = s ? A[i][c] : nk;
        A[i][c] = s ? ak : A[i][c];
      }
      A[k][c] = nk;
    }
    const double d = A[k][k];
    singular = singular || (d == 0.0);
    const double r = 1.0 / d;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double l = (d != 0.0) ? A[i][k] * r : A[i][k];
      A[i][k] = l;
#pragma unroll
      for (int c = k + 1; c < N; ++c) A[i][c] = fma(-l, A[k][c], A[i][c]);
    }
  }
  return singular;
}

template <int N>
__device__ __forceinline__ void lu_solve_inplace64(const double (&LU)[N][N], const int (&piv)[N], double (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int p = piv[k] - 1;
    const double xk = x[k];
    double nk = xk;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const bool s = (p == i);
      nk = s ? x[i] : nk;
      x[i] = s ? xk : x[i];
    }
    x[k] = nk;
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = k + 1; i < N; ++i) x[i] = fma(-LU[i][k], x[k], x[i]);
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    x[k] = x[k] / LU[k][k];
#pragma unroll
    for (int i = 0; i < k; ++i) x[i] = fma(-LU[i][k], x[k], x[i]);
  }
}

// the broadcast-FMA blocks, generic form (the compiler schedules them; dpp_blocks_f64_gen.hpp specialises L = 16)
template <int NX, int NU, int L>
struct RiccatiBlocks64 {
  static constexpr bool kAsm = false;
  static constexpr int NS = NX + NU;
  using G = Group64<L>;
  static __device__ __forceinline__ void vf(double (&W)[NX], const double (&V)[NX], const double (&Fc)[NX]) {
    static_for<0, NX>([&](auto k) {
#pragma unroll
      for (int i = 0; i < NX; ++i) W[i] = fma(G::template bcast<k.value>(V[i]), Fc[k.value], W[i]);
    });
  }
  static __device__ __forceinline__ void ftw(double (&Q)[NS], const double (&Fc)[NX], const double (&W)[NX]) {
    static_for<0, NS>([&](auto i) {
#pragma unroll
      for (int k = 0; k < NX; ++k) Q[i.value] = fma(G::template bcast<i.value>(Fc[k]), W[k], Q[i.value]);
    });
  }
  static __device__ __forceinline__ void vupd(double (&V)[NX], const double (&Q)[NS], const double (&Kt)[NU],
                                              const double (&R)[NU]) {
    static_for<0, NU>([&](auto m) {
#pragma unroll
      for (int i = 0; i < NX; ++i) V[i] = fma(G::template bcast<NX + m.value>(Q[i]), Kt[m.value], V[i]);
    });
    static_for<0, NX>([&](auto i) {
#pragma unroll
      for (int m = 0; m < NU; ++m) V[i.value] = fma(G::template bcast<i.value>(Kt[m]), R[m], V[i.value]);
    });
  }
  static __device__ __forceinline__ void dot_x(double &acc, const double xu, const double (&M)[NS + 1]) {
    static_for<0, NX>([&](auto j) { acc = fma(G::template bcast<j.value>(xu), M[j.value], acc); });
  }
  static __device__ __forceinline__ void dot_u(double &acc, const double xu, const double (&M)[NS + 1]) {
    static_for<0, NU>([&](auto m) { acc = fma(G::template bcast<NX + m.value>(xu), M[NX + m.value], acc); });
  }
  static __device__ __forceinline__ void outer2(double (&row)[NS], const double x, const double y, const double a,
                                                const double b) {
    static_for<0, NS>([&](auto j) {
      row[j.value] = fma(G::template bcast<j.value>(y), b, G::template bcast<j.value>(x) * a);
    });
  }
  static __device__ __forceinline__ void dots2_ns(double &p, double &q, const double (&M)[NS], const double x, const double y) {
    static_for<0, NS>([&](auto j) {
      p = fma(M[j.value], G::template bcast<j.value>(x), p);
      q = fma(M[j.value], G::template bcast<j.value>(y), q);
    });
  }
  static __device__ __forceinline__ void dots2_nx(double &p, double &q, const double (&M)[NX], const double x, const double y) {
    static_for<0, NX>([&](auto j) {
      p = fma(M[j.value], G::template bcast<j.value>(x), p);
      q = fma(M[j.value], G::template bcast<j.value>(y), q);
    });
  }
};

}  // namespace dmpc
