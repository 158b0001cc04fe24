// lu_api.hip - standalone batched LU factor / solve (include/dmpc.h section D).
// Replaces xpbatch_lu_factor (util.py:462-482, torch.lu = LAPACK getrf, 1-based int32 pivots)
// and xpbatch_lu_solve (util.py:505-528, float32 torch.lu_solve = LAPACK getrs) of the reference.
// The solver kernels use the same device routines (colwise.hpp) in-register; these entry points
// exist so that callers of the reference's util functions have a drop-in.
#include <hip/hip_runtime.h>

#include <cxxabi.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/dmpc.h"
#include "api_util.hpp"
#include "colwise.hpp"

namespace dmpc {

// One lane per matrix, matrix in registers (n <= 8).
template <int N>
__global__ __launch_bounds__(256) void lu_factor_kernel(int B, const float *__restrict__ A,
                                                        float *__restrict__ LU, int32_t *__restrict__ piv,
                                                        int32_t *info) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float M[N][N];
  const float *Ab = A + (size_t)b * N * N;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) M[i][j] = Ab[i * N + j];
  int p[N];
  const bool sing = lu_factor_inplace<N>(M, p);
  float *Lb = LU + (size_t)b * N * N;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) Lb[i * N + j] = M[i][j];
    piv[(size_t)b * N + i] = p[i];
  }
  if (info != nullptr && sing) atomicOr(&info[b], DMPC_INFO_SINGULAR);
}

template <int N>
__global__ __launch_bounds__(256) void lu_solve_kernel(int B, int K, const float *__restrict__ LU,
                                                       const int32_t *__restrict__ piv,
                                                       const float *__restrict__ rhs, float *__restrict__ x) {
  // one lane per (matrix, right-hand-side column)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * K) return;
  const int b = idx / K, col = idx % K;
  float M[N][N];
  int p[N];
  float v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) M[i][j] = LU[((size_t)b * N + i) * N + j];
    p[i] = piv[(size_t)b * N + i];
    v[i] = rhs[((size_t)b * N + i) * K + col];
  }
  lu_solve_inplace<N>(M, p, v);
#pragma unroll
  for (int i = 0; i < N; ++i) x[((size_t)b * N + i) * K + col] = v[i];
}

// Any n: one lane per matrix, factorisation in place in the output array (HBM/L2).
__global__ __launch_bounds__(64) void lu_factor_generic_kernel(int B, int n, const float *__restrict__ A,
                                                               float *__restrict__ LU,
                                                               int32_t *__restrict__ piv, int32_t *info) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float *M = LU + (size_t)b * n * n;
  const float *Ab = A + (size_t)b * n * n;
  for (int e = 0; e < n * n; ++e) M[e] = Ab[e];
  bool sing = false;
  for (int k = 0; k < n; ++k) {
    int p = k;
    float best = fabsf(M[k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      const float v = fabsf(M[i * n + k]);
      if (v > best) { best = v; p = i; }
    }
    piv[(size_t)b * n + k] = p + 1;
    if (p != k)
      for (int c = 0; c < n; ++c) {
        const float t = M[k * n + c];
        M[k * n + c] = M[p * n + c];
        M[p * n + c] = t;
      }
    const float d = M[k * n + k];
    sing = sing || d == 0.f;
    const float r = 1.0f / d;
    for (int i = k + 1; i < n; ++i) {
      const float l = (d != 0.f) ? M[i * n + k] * r : M[i * n + k];
      M[i * n + k] = l;
      for (int c = k + 1; c < n; ++c) M[i * n + c] = fmaf(-l, M[k * n + c], M[i * n + c]);
    }
  }
  if (info != nullptr && sing) atomicOr(&info[b], DMPC_INFO_SINGULAR);
}

__global__ __launch_bounds__(64) void lu_solve_generic_kernel(int B, int n, int K, const float *__restrict__ LU,
                                                              const int32_t *__restrict__ piv,
                                                              const float *__restrict__ rhs,
                                                              float *__restrict__ x) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * K) return;
  const int b = idx / K, col = idx % K;
  const float *M = LU + (size_t)b * n * n;
  float *xb = x + (size_t)b * n * K + col;
  const float *rb = rhs + (size_t)b * n * K + col;
  for (int i = 0; i < n; ++i) xb[(size_t)i * K] = rb[(size_t)i * K];
  for (int k = 0; k < n; ++k) {
    const int p = piv[(size_t)b * n + k] - 1;
    if (p != k) {
      const float t = xb[(size_t)k * K];
      xb[(size_t)k * K] = xb[(size_t)p * K];
      xb[(size_t)p * K] = t;
    }
  }
  for (int k = 0; k < n; ++k)
    for (int i = k + 1; i < n; ++i) xb[(size_t)i * K] = fmaf(-M[i * n + k], xb[(size_t)k * K], xb[(size_t)i * K]);
  for (int k = n - 1; k >= 0; --k) {
    const float xk = xb[(size_t)k * K] / M[k * n + k];
    xb[(size_t)k * K] = xk;
    for (int i = 0; i < k; ++i) xb[(size_t)i * K] = fmaf(-M[i * n + k], xk, xb[(size_t)i * K]);
  }
}

}  // namespace dmpc

using namespace dmpc;

extern "C" {

#ifndef DMPC_SOURCE_HASH
#define DMPC_SOURCE_HASH "unknown"
#endif
// SHA-256 prefix of the source set this library was compiled from (csrc/build.py writes it)
const char *dmpc_source_hash(void) { return DMPC_SOURCE_HASH; }

int dmpc_last_kernel_name(char *buf, size_t buf_bytes) {
  if (buf == nullptr || buf_bytes == 0) return DMPC_E_BADARG;
  buf[0] = 0;
  if (dmpc::t_last_kernel == nullptr) return 0;
  const char *mangled = hipKernelNameRefByPtr(dmpc::t_last_kernel, nullptr);
  if (mangled == nullptr) return 0;
  int status = 0;
  char *plain = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  snprintf(buf, buf_bytes, "%s", status == 0 && plain != nullptr ? plain : mangled);
  free(plain);
  return (int)strlen(buf);
}


int dmpc_batch_lu_factor(int B, int n, const float *A, float *LU, int32_t *piv, int32_t *info,
                         dmpc_stream_t stream_) {
  if (B <= 0 || n <= 0 || !A || !LU || !piv) return DMPC_E_BADARG;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 block(256), grid((B + 255) / 256);
  switch (n) {
#define CASE(N) \
  case N: DMPC_LAUNCH_GGL((lu_factor_kernel<N>), grid, block, 0, stream, B, A, LU, piv, info); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    default:
      DMPC_LAUNCH_GGL(lu_factor_generic_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, B, n, A, LU, piv,
                         info);
  }
  return (int)hipGetLastError();
}

int dmpc_batch_lu_solve(int B, int n, int k, const float *LU, const int32_t *piv, const float *b, float *x,
                        dmpc_stream_t stream_) {
  if (B <= 0 || n <= 0 || k <= 0 || !LU || !piv || !b || !x) return DMPC_E_BADARG;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int total = B * k;
  const dim3 block(256), grid((total + 255) / 256);
  switch (n) {
#define CASE(N) \
  case N: DMPC_LAUNCH_GGL((lu_solve_kernel<N>), grid, block, 0, stream, B, k, LU, piv, b, x); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    default:
      DMPC_LAUNCH_GGL(lu_solve_generic_kernel, dim3((total + 63) / 64), dim3(64), 0, stream, B, n, k, LU,
                         piv, b, x);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
