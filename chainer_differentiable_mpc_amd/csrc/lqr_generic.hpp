// lqr_generic.hpp - runtime-dimension LQR solve (kernel family 3): one wavefront per
// trajectory, every matrix in LDS, lane j owns column j of the augmented matrices.  Same
// algorithm and column layout as lqr_kernels.hpp, but dimensions are kernel arguments, so it
// covers any (nx, nu) with nx + nu + 1 <= 64 that has no register-resident specialisation.
// It is the completeness path, not the fast path.
// Follows lqr/lqr_recursion.py:69-209 and mpc/active_constrained_lqr.py:67-202.
#pragma once
#include "api_util.hpp"
#include "lqr_kernels.hpp"

namespace dmpc {

constexpr int kGenericMaxCols = 64;

struct GenericDims {
  int nx, nu, mode, k_lds;
};

__global__ __launch_bounds__(64) void lqr_generic_kernel(const LqrArgs a, const GenericDims d) {
  const int nx = d.nx, nu = d.nu, ns = nx + nu, nc = ns + 1;  // nc columns: ns matrix + 1 affine
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool masked = a.mask != nullptr;

  extern __shared__ float lds[];
  float *Vt = lds;                 // [nx][nc]  (columns nx..ns-1 unused, column ns = v)
  float *Ft = Vt + nx * nc;        // [nx][nc]
  float *Qt = Ft + nx * nc;        // [ns][nc]
  float *Wt = Qt + ns * nc;        // [nx][nc]
  float *LU = Wt + nx * nc;        // [nu][nu]
  float *Kt = LU + nu * nu;        // [nu][nc]
  float *Rt = Kt + nu * nc;        // [nu][nc]
  float *xu = Rt + nu * nc;        // [nc]
  int *piv = reinterpret_cast<int *>(xu + nc);   // [nu]
  int *flags = piv + nu;                          // [1]
  float *kl = reinterpret_cast<float *>(flags + 1);  // [T][nu][nx+1] when d.k_lds
  const int krow = nx + 1;
  const bool col = lane < nc;
  if (lane == 0) flags[0] = 0;

  if (d.mode != kForwardOnly) {
    for (int e = lane; e < nx * nc; e += 64) Vt[e] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
      const size_t tb = (size_t)t * B + b;
      const float *Cp = a.C + tb * ns * ns;
      for (int e = lane; e < ns * ns; e += 64) Qt[(e / ns) * nc + (e % ns)] = Cp[e];
      for (int i = lane; i < ns; i += 64) Qt[i * nc + ns] = a.c[tb * ns + i];
      if (t < T - 1) {
        const float *Fp = a.F + tb * nx * ns;
        for (int e = lane; e < nx * ns; e += 64) Ft[(e / ns) * nc + (e % ns)] = Fp[e];
        for (int i = lane; i < nx; i += 64) Ft[i * nc + ns] = a.f ? a.f[tb * nx + i] : 0.f;
      }
      __syncthreads();
      if (t < T - 1) {
        if (col) {
          for (int i = 0; i < nx; ++i) {
            float acc = (lane == ns) ? Vt[i * nc + ns] : 0.f;
            for (int k = 0; k < nx; ++k) acc = fmaf(Vt[i * nc + k], Ft[k * nc + lane], acc);
            Wt[i * nc + lane] = acc;
          }
        }
        __syncthreads();
        if (col) {
          for (int i = 0; i < ns; ++i) {
            float acc = Qt[i * nc + lane];
            for (int k = 0; k < nx; ++k) acc = fmaf(Ft[k * nc + i], Wt[k * nc + lane], acc);
            Qt[i * nc + lane] = acc;
          }
        }
        __syncthreads();
      }
      // factor Quu (lane 0), LAPACK getf2 order
      if (lane == 0) {
        for (int m = 0; m < nu; ++m)
          for (int l = 0; l < nu; ++l) {
            float v = Qt[(nx + m) * nc + nx + l];
            if (masked) {
              const bool am = a.mask[tb * nu + m] != 0, al = a.mask[tb * nu + l] != 0;
              v = (am || al) ? 0.f : v;
              if (m == l && am) v += 1e-8f;
            }
            LU[m * nu + l] = v;
          }
        for (int k = 0; k < nu; ++k) {
          int p = k;
          float best = fabsf(LU[k * nu + k]);
          for (int i = k + 1; i < nu; ++i) {
            const float v = fabsf(LU[i * nu + k]);
            if (v > best) { best = v; p = i; }
          }
          piv[k] = p;
          if (p != k)
            for (int cidx = 0; cidx < nu; ++cidx) {
              const float tmp = LU[k * nu + cidx];
              LU[k * nu + cidx] = LU[p * nu + cidx];
              LU[p * nu + cidx] = tmp;
            }
          const float dpiv = LU[k * nu + k];
          if (dpiv == 0.f) flags[0] |= 1;
          const float r = 1.0f / dpiv;
          for (int i = k + 1; i < nu; ++i) {
            const float l = (dpiv != 0.f) ? LU[i * nu + k] * r : LU[i * nu + k];
            LU[i * nu + k] = l;
            for (int cidx = k + 1; cidx < nu; ++cidx)
              LU[i * nu + cidx] = fmaf(-l, LU[k * nu + cidx], LU[i * nu + cidx]);
          }
        }
      }
      __syncthreads();
      if (col) {
        // own right-hand side column, solved in place in Kt[.][lane]
        for (int m = 0; m < nu; ++m) {
          float v = Qt[(nx + m) * nc + lane];
          if (masked && a.mask[tb * nu + m] != 0) v = 0.f;
          Kt[m * nc + lane] = v;
        }
        if (nu == 1) {
          Kt[lane] = -((1.0f / LU[0]) * Kt[lane]);
        } else {
          for (int k = 0; k < nu; ++k) {
            const int p = piv[k];
            if (p != k) {
              const float tmp = Kt[k * nc + lane];
              Kt[k * nc + lane] = Kt[p * nc + lane];
              Kt[p * nc + lane] = tmp;
            }
          }
          for (int k = 0; k < nu; ++k)
            for (int i = k + 1; i < nu; ++i)
              Kt[i * nc + lane] = fmaf(-LU[i * nu + k], Kt[k * nc + lane], Kt[i * nc + lane]);
          for (int k = nu - 1; k >= 0; --k) {
            const float xk = Kt[k * nc + lane] / LU[k * nu + k];
            Kt[k * nc + lane] = xk;
            for (int i = 0; i < k; ++i)
              Kt[i * nc + lane] = fmaf(-LU[i * nu + k], xk, Kt[i * nc + lane]);
          }
          for (int m = 0; m < nu; ++m) Kt[m * nc + lane] = -Kt[m * nc + lane];
        }
        if (lane < nx || lane == ns) {
          const int kidx = lane == ns ? nx : lane;
          for (int m = 0; m < nu; ++m) {
            const float kv = Kt[m * nc + lane];
            if (d.k_lds) kl[(t * nu + m) * krow + kidx] = kv;
            if (a.Ks != nullptr) {
              if (lane == ns) a.ks[tb * nu + m] = kv;
              else a.Ks[(tb * nu + m) * nx + lane] = kv;
            }
          }
        }
        // R = Qu. + Quu K~  (unmasked Quu)
        for (int m = 0; m < nu; ++m) {
          float acc = Qt[(nx + m) * nc + lane];
          for (int l = 0; l < nu; ++l) acc = fmaf(Qt[(nx + m) * nc + nx + l], Kt[l * nc + lane], acc);
          Rt[m * nc + lane] = acc;
        }
      }
      __syncthreads();
      if (col && t > 0) {
        for (int i = 0; i < nx; ++i) {
          float acc = Qt[i * nc + lane];
          for (int m = 0; m < nu; ++m) acc = fmaf(Qt[i * nc + nx + m], Kt[m * nc + lane], acc);
          for (int m = 0; m < nu; ++m) acc = fmaf(Kt[m * nc + i], Rt[m * nc + lane], acc);
          Vt[i * nc + lane] = acc;
        }
      }
      __syncthreads();
    }
  }

  if (d.mode != kBackwardOnly) {
    __threadfence_block();
    for (int j = lane; j < nx; j += 64) xu[j] = a.x_init[(size_t)b * nx + j];
    __syncthreads();
    bool bad = false;
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      if (lane < nu) {
        const int m = lane;
        float acc;
        if (d.mode == kSolve && d.k_lds) {
          acc = kl[(t * nu + m) * krow + nx];
          for (int j = 0; j < nx; ++j) acc = fmaf(kl[(t * nu + m) * krow + j], xu[j], acc);
        } else {
          acc = a.ks[tb * nu + m];
          for (int j = 0; j < nx; ++j) acc = fmaf(a.Ks[(tb * nu + m) * nx + j], xu[j], acc);
        }
        if (masked && a.mask[tb * nu + m] != 0) acc = 0.f;
        xu[nx + m] = acc;
        a.u[tb * nu + m] = acc;
        bad = bad || !is_finite(acc);
      }
      __syncthreads();
      float xn = 0.f;
      if (lane < nx) {
        a.x[tb * nx + lane] = xu[lane];
        bad = bad || !is_finite(xu[lane]);
        if (t < T - 1) {
          const float *Fr = a.F + (tb * nx + lane) * ns;
          xn = a.f ? a.f[tb * nx + lane] : 0.f;
          for (int j = 0; j < ns; ++j) xn = fmaf(Fr[j], xu[j], xn);
        }
      }
      __syncthreads();
      if (lane < nx && t < T - 1) xu[lane] = xn;
      __syncthreads();
    }
    if (bad) atomicOr(&flags[0], 2);
  }
  __syncthreads();
  if (lane == 0 && a.info != nullptr && flags[0] != 0) atomicOr(&a.info[b], flags[0]);
}

static inline size_t lqr_generic_lds_bytes(int T, int nx, int nu, bool k_lds) {
  const int ns = nx + nu, nc = ns + 1;
  size_t fl = (size_t)nx * nc * 3 + (size_t)ns * nc + (size_t)nu * nu + (size_t)nu * nc * 2 + nc;
  size_t bytes = fl * sizeof(float) + (size_t)(nu + 1) * sizeof(int);
  if (k_lds) bytes += (size_t)T * nu * (nx + 1) * sizeof(float);
  return bytes;
}

static int launch_lqr_generic(int mode, int nx, int nu, const LqrArgs &a, hipStream_t stream) {
  const size_t base = lqr_generic_lds_bytes(a.T, nx, nu, false);
  const size_t with_k = lqr_generic_lds_bytes(a.T, nx, nu, true);
  LqrArgs args = a;
  GenericDims d{nx, nu, mode, 0};
  size_t shmem = base;
  if (mode == kSolve) {
    if (with_k <= 60 * 1024) {
      d.k_lds = 1;
      shmem = with_k;
    } else {
      if (args.Ks == nullptr) { args.Ks = args.wsK; args.ks = args.wsk; }
      if (args.Ks == nullptr) return DMPC_E_WORKSPACE;
    }
  }
  if (shmem > 64 * 1024) return DMPC_E_UNSUPPORTED;
  DMPC_LAUNCH_GGL(lqr_generic_kernel, dim3(a.B), dim3(64), shmem, stream, args, d);
  return (int)hipGetLastError();
}

}  // namespace dmpc
