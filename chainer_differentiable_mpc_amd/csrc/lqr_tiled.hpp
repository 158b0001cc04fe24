// lqr_tiled.hpp - the LQR solve for ANY (nx, nu): kernel family 5, the shapes beyond a wavefront's 64 columns.
//
// The reference has no size limit (lqr/lqr_recursion.py:69-209 is numpy on whatever shapes it is given); until round 4 a
// problem with nx + nu + 1 > 64 was refused with DMPC_E_UNSUPPORTED.  Here: one workgroup of 256 threads per trajectory,
// runtime dimensions, every matrix of the trajectory in a per-trajectory area of the caller's workspace (HBM/L2: the areas
// are written and read by the same compute unit, and `__syncthreads()` orders them), each phase of a timestep spread over
// the threads by output element.  Same algorithm, operation order and pivoting as lqr_generic.hpp / the oracle:
//     Q~ = [C|c] + F~^T (V~ F~)     LU of (masked) Quu in LAPACK getf2 order     K~ = -Quu^-1 [Qux|Quu|qu]
//     V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~)        rollout with the gains read back from HBM
// It is the completeness path for size - ~100 x slower per flop than the register-resident kernels, and correct.
#pragma once
#include "api_util.hpp"
#include "lqr_kernels.hpp"

namespace dmpc {

constexpr int kTiledThreads = 256;

struct TiledDims {
  int nx, nu, mode;
  float *scratch;          // [B][tiled_scratch_floats(nx, nu)]
};

__host__ __device__ inline size_t tiled_scratch_floats(int nx, int nu) {
  const size_t ns = (size_t)nx + nu, nc = ns + 1;
  // V~ [nx][nc], Q~ [ns][nc], W~ [nx][nc], LU [nu][nu], K~ [nu][nc], R [nu][nc], piv [nu] (as ints), flags [4]
  return 2 * (size_t)nx * nc + ns * nc + (size_t)nu * nu + 2 * (size_t)nu * nc + nu + 4;
}

template <int kInstance = 0>   // (a template so that the header may be included by several translation units)
__global__ __launch_bounds__(kTiledThreads) void lqr_tiled_kernel(const LqrArgs a, const TiledDims d) {
  const int nx = d.nx, nu = d.nu, ns = nx + nu, nc = ns + 1;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool masked = a.mask != nullptr;
  constexpr int NT = kTiledThreads;

  float *Vt = d.scratch + (size_t)b * tiled_scratch_floats(nx, nu);
  float *Qt = Vt + (size_t)nx * nc;
  float *Wt = Qt + (size_t)ns * nc;
  float *LU = Wt + (size_t)nx * nc;
  float *Kt = LU + (size_t)nu * nu;
  float *Rt = Kt + (size_t)nu * nc;
  int *piv = reinterpret_cast<int *>(Rt + (size_t)nu * nc);
  __shared__ int s_flags, s_p;
  extern __shared__ float xu[];   // [ns] the rollout's [x_t; u_t], then [nx] x_{t+1} in the making
  float *xnext = xu + ns;
  if (tid == 0) s_flags = 0;
  float *Ks = a.Ks != nullptr ? a.Ks : a.wsK;
  float *ks = a.Ks != nullptr ? a.ks : a.wsk;

  if (d.mode != kForwardOnly) {
    for (int e = tid; e < nx * nc; e += NT) Vt[e] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
      const size_t tb = (size_t)t * B + b;
      const float *Cp = a.C + tb * ns * ns;
      for (int e = tid; e < ns * ns; e += NT) Qt[(e / ns) * nc + (e % ns)] = Cp[e];
      for (int i = tid; i < ns; i += NT) Qt[i * nc + ns] = a.c[tb * ns + i];
      __syncthreads();
      if (t < T - 1) {
        const float *Fp = a.F + tb * nx * ns;
        const float *fp = a.f ? a.f + tb * nx : nullptr;
        // W~ = V F~ (+ v in the affine column)                                     lqr_recursion.py:89,96
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
        // Q~ += F^T W~
        for (int e = tid; e < ns * nc; e += NT) {
          const int i = e / nc, j = e % nc;
          float acc = Qt[e];
          for (int k = 0; k < nx; ++k) acc = fmaf(Fp[(size_t)k * ns + i], Wt[k * nc + j], acc);
          Qt[e] = acc;
        }
        __syncthreads();
      }
      // ---- LU of (masked) Quu, LAPACK getf2 order                   :112-120 / active_constrained_lqr.py:110-137
      for (int e = tid; e < nu * nu; e += NT) {
        const int m = e / nu, l = e % nu;
        float v = Qt[(nx + m) * nc + nx + l];
        if (masked) {
          const bool am = a.mask[tb * nu + m] != 0, al = a.mask[tb * nu + l] != 0;
          v = (am || al) ? 0.f : v;
          if (m == l && am) v += 1e-8f;
        }
        LU[e] = v;
      }
      __syncthreads();
      for (int k = 0; k < nu; ++k) {
        if (tid == 0) {
          int p = k;
          float best = fabsf(LU[k * nu + k]);
          for (int i = k + 1; i < nu; ++i) {
            const float v = fabsf(LU[i * nu + k]);
            if (v > best) { best = v; p = i; }     // strict: the first maximum wins (idamax)
          }
          piv[k] = p;
          s_p = p;
        }
        __syncthreads();
        const int p = s_p;
        if (p != k)
          for (int cidx = tid; cidx < nu; cidx += NT) {
            const float tmp = LU[k * nu + cidx];
            LU[k * nu + cidx] = LU[p * nu + cidx];
            LU[p * nu + cidx] = tmp;
          }
        __syncthreads();
        const float dpiv = LU[k * nu + k];
        if (dpiv == 0.f && tid == 0) s_flags |= 1;
        const float r = 1.0f / dpiv;
        for (int i = k + 1 + tid; i < nu; i += NT)
          if (dpiv != 0.f) LU[i * nu + k] *= r;
        __syncthreads();
        const int rem = nu - k - 1;
        for (int e = tid; e < rem * rem; e += NT) {
          const int i = k + 1 + e / rem, cidx = k + 1 + e % rem;
          LU[i * nu + cidx] = fmaf(-LU[i * nu + k], LU[k * nu + cidx], LU[i * nu + cidx]);
        }
        __syncthreads();
      }
      // ---- K~ = -Quu^-1 [Qux | Quu | qu]: a thread per right-hand-side column
      for (int j = tid; j < nc; j += NT) {
        for (int m = 0; m < nu; ++m) {
          float v = Qt[(nx + m) * nc + j];
          if (masked && a.mask[tb * nu + m] != 0) v = 0.f;
          Kt[m * nc + j] = v;
        }
        if (nu == 1) {
          Kt[j] = -((1.0f / LU[0]) * Kt[j]);
        } else {
          for (int k = 0; k < nu; ++k) {
            const int p = piv[k];
            if (p != k) {
              const float tmp = Kt[k * nc + j];
              Kt[k * nc + j] = Kt[p * nc + j];
              Kt[p * nc + j] = tmp;
            }
          }
          for (int k = 0; k < nu; ++k)
            for (int i = k + 1; i < nu; ++i) Kt[i * nc + j] = fmaf(-LU[i * nu + k], Kt[k * nc + j], Kt[i * nc + j]);
          for (int k = nu - 1; k >= 0; --k) {
            const float xk = Kt[k * nc + j] / LU[k * nu + k];
            Kt[k * nc + j] = xk;
            for (int i = 0; i < k; ++i) Kt[i * nc + j] = fmaf(-LU[i * nu + k], xk, Kt[i * nc + j]);
          }
          for (int m = 0; m < nu; ++m) Kt[m * nc + j] = -Kt[m * nc + j];
        }
        if (j < nx || j == ns) {
          for (int m = 0; m < nu; ++m) {
            const float kv = Kt[m * nc + j];
            if (j == ns) ks[tb * nu + m] = kv;
            else Ks[(tb * nu + m) * nx + j] = kv;
          }
        }
        // R = Qu. + Quu K~ (unmasked Quu)                                                           :151-152
        for (int m = 0; m < nu; ++m) {
          float acc = Qt[(nx + m) * nc + j];
          for (int l = 0; l < nu; ++l) acc = fmaf(Qt[(nx + m) * nc + nx + l], Kt[l * nc + j], acc);
          Rt[m * nc + j] = acc;
        }
      }
      __syncthreads();
      if (t > 0) {
        for (int e = tid; e < nx * nc; e += NT) {
          const int i = e / nc, j = e % nc;
          float acc = Qt[i * nc + j];
          for (int m = 0; m < nu; ++m) acc = fmaf(Qt[i * nc + nx + m], Kt[m * nc + j], acc);
          for (int m = 0; m < nu; ++m) acc = fmaf(Kt[m * nc + i], Rt[m * nc + j], acc);
          Vt[e] = acc;
        }
      }
      __syncthreads();
    }
  }

  if (d.mode != kBackwardOnly) {                                               // :160-200
    __threadfence_block();
    for (int j = tid; j < nx; j += NT) xu[j] = a.x_init[(size_t)b * nx + j];
    __syncthreads();
    bool bad = false;
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      for (int m = tid; m < nu; m += NT) {
        float acc = ks[tb * nu + m];
        const float *Kr = Ks + (tb * nu + m) * nx;
        for (int j = 0; j < nx; ++j) acc = fmaf(Kr[j], xu[j], acc);
        if (masked && a.mask[tb * nu + m] != 0) acc = 0.f;
        xu[nx + m] = acc;
        a.u[tb * nu + m] = acc;
        bad = bad || !is_finite(acc);
      }
      __syncthreads();
      for (int i = tid; i < nx; i += NT) {
        a.x[tb * nx + i] = xu[i];
        bad = bad || !is_finite(xu[i]);
      }
      // x_{t+1} = F_t [x_t; u_t] + f_t: every thread reads all of xu, so the new state is formed next to it
      if (t < T - 1) {
        for (int i = tid; i < nx; i += NT) {
          const float *Fr = a.F + (tb * nx + i) * ns;
          float acc = a.f ? a.f[tb * nx + i] : 0.f;
          for (int j = 0; j < ns; ++j) acc = fmaf(Fr[j], xu[j], acc);
          xnext[i] = acc;
        }
      }
      __syncthreads();
      if (t < T - 1)
        for (int i = tid; i < nx; i += NT) xu[i] = xnext[i];
      __syncthreads();
    }
    if (bad) atomicOr(&s_flags, 2);
  }
  __syncthreads();
  if (tid == 0 && a.info != nullptr && s_flags != 0) atomicOr(&a.info[b], s_flags);
}

static int launch_lqr_tiled(int mode, int nx, int nu, const LqrArgs &a, float *scratch, hipStream_t stream) {
  if (mode != kForwardOnly && scratch == nullptr) return DMPC_E_WORKSPACE;      // the sweep's matrices live there
  if (mode != kForwardOnly && a.Ks == nullptr && a.wsK == nullptr) return DMPC_E_WORKSPACE;
  if (mode == kForwardOnly && a.Ks == nullptr) return DMPC_E_BADARG;
  const size_t shmem = (size_t)(2 * nx + nu) * sizeof(float);
  if (shmem > 60 * 1024) return DMPC_E_UNSUPPORTED;       // (more than 15,000 states + controls)
  TiledDims d{nx, nu, mode, scratch};
  DMPC_LAUNCH_GGL(lqr_tiled_kernel<0>, dim3(a.B), dim3(kTiledThreads), shmem, stream, a, d);
  return (int)hipGetLastError();
}

}  // namespace dmpc
