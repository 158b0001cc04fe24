// colwise.hpp - "one lane per matrix column" primitives for gfx950 (CDNA4, wave64).
//
// Data layout shared by every solver kernel in this directory.  A trajectory is owned by a
// GROUP of L lanes (L = 16: one DPP row, four trajectories per wavefront; L = 64: a whole
// wavefront).  A small matrix M (r x ncol, ncol <= L) lives in r VGPRs: register i of lane j
// holds M[i][j].  With that layout the three products the Riccati sweep needs are
//
//     (A*B)[i][j]   = sum_k A[i][k] B[k][j]  ->  c[i] += bcast<k>(a[i]) * b[k]
//     (A^T*B)[i][j] = sum_k A[k][i] B[k][j]  ->  c[i] += bcast<i>(a[k]) * b[k]
//
// where bcast<k>(v) = "v of lane k of my group, in every lane of the group":
//   L = 16 : DPP row_newbcast:k (gfx90a+; lives in the VALU operand path, no LDS traffic)
//   L = 64 : v_readlane_b32 -> SGPR operand
// Affine terms ride along as one extra column (lane ns): [C | c], [F | f], [V | v], [K | k].
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace dmpc {

template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<I + 1, N>(fn);
  }
}

template <int L>
struct Group;

template <>
struct Group<16> {
  static constexpr int kLanes = 16;
  template <int K>
  static __device__ __forceinline__ float bcast(float v) {
    static_assert(K >= 0 && K < 16, "row_newbcast lane out of range");
    // dpp_ctrl 0x150 + K = row_newbcast:K ; row_mask = bank_mask = 0xf ; bound_ctrl = 1
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, true));
  }
};

template <>
struct Group<64> {
  static constexpr int kLanes = 64;
  template <int K>
  static __device__ __forceinline__ float bcast(float v) {
    static_assert(K >= 0 && K < 64, "readlane lane out of range");
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), K));
  }
};

// "does any active lane of the wavefront want this?" - a wave-uniform value (one ballot), so the `if` around it is a scalar
// branch.  LAPACK's row interchange costs ~2 N selects per column and row candidate in every lane; a Quu that is close to
// diagonally dominant - every problem of the generators, most real ones - never needs one, so the selects sit behind this
// test and are skipped when no lane's pivot row differs from the diagonal.  Same arithmetic, same results: lanes that do
// not swap execute selects that select nothing.  (round 4: 83 of the 660 instructions of a (8,4) backward step)
__device__ __forceinline__ bool any_lane(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }

// ---------------------------------------------------------------------------------------
// In-register LU with partial pivoting (LAPACK getf2 semantics: first max |a| in the column,
// one row interchange k <-> p, scale by the reciprocal pivot).  Every lane of a group holds
// the same N x N matrix; `rhs` is the lane's own right-hand side column, so one factorisation
// solves L columns at once.  All indices are compile-time: nothing spills to scratch.
// piv (optional) receives LAPACK's 1-based pivot rows.  Returns true if a zero pivot was met.
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ bool lu_factor_inplace(float (&A)[N][N], int (&piv)[N]) {
  bool singular = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float best = fabsf(A[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const float v = fabsf(A[i][k]);
      const bool gt = v > best;  // strict: first maximum wins (idamax)
      best = gt ? v : best;
      p = gt ? i : p;
    }
    piv[k] = p + 1;
    if (any_lane(p != k)) {   // the interchange behind a wave-uniform branch: see any_lane
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const float ak = A[k][c];
        float nk = ak;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
          const bool s = (p == i);
          nk = s ? A[i][c] : nk;
          A[i][c] = s ? ak : A[i][c];
        }
        A[k][c] = nk;
      }
    }
    const float d = A[k][k];
    singular = singular || (d == 0.0f);
    const float r = 1.0f / d;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const float l = (d != 0.0f) ? A[i][k] * r : A[i][k];
      A[i][k] = l;
#pragma unroll
      for (int c = k + 1; c < N; ++c) A[i][c] = fmaf(-l, A[k][c], A[i][c]);
    }
  }
  return singular;
}

// getrs: row interchanges, unit-lower forward substitution, upper back substitution.
template <int N>
__device__ __forceinline__ void lu_solve_inplace(const float (&LU)[N][N], const int (&piv)[N],
                                                 float (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int p = piv[k] - 1;
    if (!any_lane(p != k)) continue;
    const float xk = x[k];
    float nk = xk;
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
    for (int i = k + 1; i < N; ++i) x[i] = fmaf(-LU[i][k], x[k], x[i]);
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    x[k] = x[k] / LU[k][k];
#pragma unroll
    for (int i = 0; i < k; ++i) x[i] = fmaf(-LU[i][k], x[k], x[i]);
  }
}

// Reciprocal to within ~1 ulp: v_rcp_f32 (1 ulp) + one Newton step.  The in-kernel solves multiply by the
// stored reciprocal pivots instead of dividing (LAPACK getf2 scales by the reciprocal pivot as well); an
// IEEE division costs ~10 VALU issue slots, this costs 3.
__device__ __forceinline__ float fast_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}

// LU as lu_factor_inplace, additionally returning the reciprocal pivots for lu_solve_rinv.
// UNIFORM: every lane of the wavefront holds the SAME matrix (the wavefront-per-trajectory kernels): the pivot row is then a
// wave-uniform value and the interchange - N selects per candidate row and column, most of this routine's instructions at
// N = 8 - sits behind a scalar branch that a well-conditioned matrix never takes.  Same arithmetic, same results.
template <int N, bool UNIFORM = false>
__device__ __forceinline__ bool lu_factor_rinv(float (&A)[N][N], int (&piv)[N], float (&rinv)[N]) {
  bool singular = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float best = fabsf(A[k][k]);
    int p = k;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const float v = fabsf(A[i][k]);
      const bool gt = v > best;
      best = gt ? v : best;
      p = gt ? i : p;
    }
    if constexpr (UNIFORM) p = __builtin_amdgcn_readfirstlane(p);
    piv[k] = p + 1;
    if (N > 1 && (UNIFORM ? p != k : any_lane(p != k))) {
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const float ak = A[k][c];
        float nk = ak;
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
          const bool s = (p == i);
          nk = s ? A[i][c] : nk;
          A[i][c] = s ? ak : A[i][c];
        }
        A[k][c] = nk;
      }
    }
    const float d = A[k][k];
    singular = singular || (d == 0.0f);
    const float r = fast_rcp(d);
    rinv[k] = r;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const float l = A[i][k] * r;
      A[i][k] = l;
#pragma unroll
      for (int c = k + 1; c < N; ++c) A[i][c] = fmaf(-l, A[k][c], A[i][c]);
    }
  }
  return singular;
}

template <int N, bool UNIFORM = false>
__device__ __forceinline__ void lu_solve_rinv(const float (&LU)[N][N], const int (&piv)[N], const float (&rinv)[N],
                                              float (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int p = piv[k] - 1;
    if (UNIFORM ? p == k : !any_lane(p != k)) continue;   // the interchange behind a scalar branch
    const float xk = x[k];
    float nk = xk;
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
    for (int i = k + 1; i < N; ++i) x[i] = fmaf(-LU[i][k], x[k], x[i]);
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    x[k] = x[k] * rinv[k];
#pragma unroll
    for (int i = 0; i < k; ++i) x[i] = fmaf(-LU[i][k], x[k], x[i]);
  }
}

// Sum of v over the lanes of a group, result in every lane of the group.
template <int L>
__device__ __forceinline__ float group_sum(float v);

template <>
__device__ __forceinline__ float group_sum<16>(float v) {
  // rotate-and-add inside the 16-lane DPP row: row_ror:8,4,2,1 (dpp_ctrl 0x120 + n)
#define DMPC_ROR_ADD(N) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true))
  DMPC_ROR_ADD(8);
  DMPC_ROR_ADD(4);
  DMPC_ROR_ADD(2);
  DMPC_ROR_ADD(1);
#undef DMPC_ROR_ADD
  return v;
}

template <>
__device__ __forceinline__ float group_sum<64>(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ bool is_finite(float v) { return fabsf(v) <= 3.402823466e+38f; }

}  // namespace dmpc
