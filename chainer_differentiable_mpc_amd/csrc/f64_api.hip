// f64_api.hip - reference-precision variants of the LQR solve and of its analytic gradient (`_f64` entry
// points, optional; the reference computes in float64 - lqr/differentiable_lqr.py:169-172, numpy's default everywhere).
//
// Two families behind the same entry points:
//   * f64_row_kernels.hpp (round 4): the register-resident column-per-lane kernels in double - a trajectory per 16 lanes
//     (`v_fmac_f64_dpp` blocks) or per wavefront - for the shapes of DMPC_F64_ROW_SHAPES; the fast path
//     (`dmpc_lqr_f64_path` says which; DMPC_NO_F64_ROW=1 takes it out);
//   * the kernels below: one lane per trajectory, every matrix of the trajectory in a caller workspace laid out
//     element-major / trajectory-minor (`ws[e * B + b]`, so the 64 lanes of a wavefront touch one run of HBM per access),
//     runtime dimensions, no cross-lane traffic - any shape at all; the completeness path.
// Same algorithm and operation order as the float32 kernels' runtime-dimension version (lqr_generic.hpp) and the oracle:
//   solve      lqr/lqr_recursion.py:69-209 (LqrRecursion.backward + .forward; LQR_active with `mask`,
//              mpc/active_constrained_lqr.py:110-145), LU with partial pivoting in LAPACK getf2 order for F.batch_inv
//   gradient   lqr/differentiable_lqr.py:78-142 (second solve on [grad_x; grad_u], co-state recursions, outer products)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "../../include/dmpc.h"
#include "api_util.hpp"
#include "f64_row_kernels.hpp"
#include "lqr_tile16_f64.hpp"

namespace dmpc {

struct F64Solve {
  int T, B, nx, nu;
  const double *C, *c, *F, *f, *x_init;
  const uint8_t *mask;
  double *Ks, *ks, *x, *u;   // gains [T,B,nu,nx], [T,B,nu] (always written: the rollout reads them back), x, u
  double *ws;
  int32_t *info;
  int c_cols;                // row length of c: ns, or - second solve - 0 = c is given as cx [T,B,nx] and cu [T,B,nu]
  const double *cu;
};

// workspace elements per trajectory of the solve: V~ [nx][nc], Q~ [ns][nc], W~ [nx][nc], LU [nu][nu], K~ [nu][nc], R [nu][nc],
// x [nx], piv [nu]
__host__ __device__ inline size_t f64_solve_ws_elems(int nx, int nu) {
  const size_t ns = nx + nu, nc = ns + 1;
  return 2 * (size_t)nx * nc + ns * nc + (size_t)nu * nu + 2 * (size_t)nu * nc + nx + nu;
}

__global__ __launch_bounds__(64) void lqr_f64_kernel(const F64Solve a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const int nx = a.nx, nu = a.nu, ns = nx + nu, nc = ns + 1, T = a.T;
  const size_t B = (size_t)a.B;
  double *w = a.ws + b;
#define AT(off, i) w[((size_t)(off) + (size_t)(i)) * B]
  const size_t oV = 0, oQ = oV + (size_t)nx * nc, oW = oQ + (size_t)ns * nc, oL = oW + (size_t)nx * nc,
               oK = oL + (size_t)nu * nu, oR = oK + (size_t)nu * nc, oX = oR + (size_t)nu * nc, oP = oX + nx;
  int flags = 0;
  for (int e = 0; e < nx * nc; ++e) AT(oV, e) = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    const double *Cp = a.C + tb * ns * ns;
    for (int i = 0; i < ns; ++i) {
      for (int j = 0; j < ns; ++j) AT(oQ, i * nc + j) = Cp[i * ns + j];
      double ci;
      if (a.c_cols != 0) ci = a.c[tb * ns + i];
      else ci = i < nx ? a.c[tb * nx + i] : a.cu[tb * nu + (i - nx)];
      AT(oQ, i * nc + ns) = ci;
    }
    if (t < T - 1) {
      const double *Fp = a.F + tb * nx * ns;
      const double *fp = a.f ? a.f + tb * nx : nullptr;
      // W~ = V F~ (+ v in the affine column)                                     lqr_recursion.py:89,96
      for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nc; ++j) {
          double acc = j == ns ? AT(oV, i * nc + ns) : 0.0;
          for (int k = 0; k < nx; ++k) {
            const double fkj = j < ns ? Fp[k * ns + j] : (fp ? fp[k] : 0.0);
            acc = fma(AT(oV, i * nc + k), fkj, acc);
          }
          AT(oW, i * nc + j) = acc;
        }
      // Q~ += F^T W~
      for (int i = 0; i < ns; ++i)
        for (int j = 0; j < nc; ++j) {
          double acc = AT(oQ, i * nc + j);
          for (int k = 0; k < nx; ++k) acc = fma(Fp[k * ns + i], AT(oW, k * nc + j), acc);
          AT(oQ, i * nc + j) = acc;
        }
    }
    // LU of (masked) Quu, LAPACK getf2 order                                      :112-120 / active_constrained_lqr.py:110-137
    for (int m = 0; m < nu; ++m)
      for (int l = 0; l < nu; ++l) {
        double v = AT(oQ, (nx + m) * nc + nx + l);
        if (a.mask != nullptr) {
          const bool am = a.mask[tb * nu + m] != 0, al = a.mask[tb * nu + l] != 0;
          v = (am || al) ? 0.0 : v;
          if (m == l && am) v += 1e-8;
        }
        AT(oL, m * nu + l) = v;
      }
    for (int k = 0; k < nu; ++k) {
      int p = k;
      double best = fabs(AT(oL, k * nu + k));
      for (int i = k + 1; i < nu; ++i) {
        const double v = fabs(AT(oL, i * nu + k));
        if (v > best) { best = v; p = i; }
      }
      AT(oP, k) = (double)p;
      if (p != k)
        for (int j = 0; j < nu; ++j) {
          const double tmp = AT(oL, k * nu + j);
          AT(oL, k * nu + j) = AT(oL, p * nu + j);
          AT(oL, p * nu + j) = tmp;
        }
      const double dpiv = AT(oL, k * nu + k);
      if (dpiv == 0.0) flags |= 1;
      const double r = 1.0 / dpiv;
      for (int i = k + 1; i < nu; ++i) {
        const double l = (dpiv != 0.0) ? AT(oL, i * nu + k) * r : AT(oL, i * nu + k);
        AT(oL, i * nu + k) = l;
        for (int j = k + 1; j < nu; ++j) AT(oL, i * nu + j) = fma(-l, AT(oL, k * nu + j), AT(oL, i * nu + j));
      }
    }
    // K~ = -Quu^-1 [Qux | Quu | qu], column by column
    for (int j = 0; j < nc; ++j) {
      for (int m = 0; m < nu; ++m) {
        double v = AT(oQ, (nx + m) * nc + j);
        if (a.mask != nullptr && a.mask[tb * nu + m] != 0) v = 0.0;
        AT(oK, m * nc + j) = v;
      }
      for (int k = 0; k < nu; ++k) {
        const int p = (int)AT(oP, k);
        if (p != k) {
          const double tmp = AT(oK, k * nc + j);
          AT(oK, k * nc + j) = AT(oK, p * nc + j);
          AT(oK, p * nc + j) = tmp;
        }
      }
      for (int k = 0; k < nu; ++k)
        for (int i = k + 1; i < nu; ++i) AT(oK, i * nc + j) = fma(-AT(oL, i * nu + k), AT(oK, k * nc + j), AT(oK, i * nc + j));
      for (int k = nu - 1; k >= 0; --k) {
        const double xk = AT(oK, k * nc + j) / AT(oL, k * nu + k);
        AT(oK, k * nc + j) = xk;
        for (int i = 0; i < k; ++i) AT(oK, i * nc + j) = fma(-AT(oL, i * nu + k), xk, AT(oK, i * nc + j));
      }
      for (int m = 0; m < nu; ++m) AT(oK, m * nc + j) = -AT(oK, m * nc + j);
    }
    for (int m = 0; m < nu; ++m) {
      for (int j = 0; j < nx; ++j) a.Ks[(tb * nu + m) * nx + j] = AT(oK, m * nc + j);
      a.ks[tb * nu + m] = AT(oK, m * nc + ns);
    }
    if (t > 0) {
      // R = Qu. + Quu K~ (unmasked Quu); V~ = Q~x. + Qxu K~ + K~^T R                :151-152
      for (int m = 0; m < nu; ++m)
        for (int j = 0; j < nc; ++j) {
          double acc = AT(oQ, (nx + m) * nc + j);
          for (int l = 0; l < nu; ++l) acc = fma(AT(oQ, (nx + m) * nc + nx + l), AT(oK, l * nc + j), acc);
          AT(oR, m * nc + j) = acc;
        }
      for (int i = 0; i < nx; ++i)
        for (int j = 0; j < nc; ++j) {
          double acc = AT(oQ, i * nc + j);
          for (int m = 0; m < nu; ++m) acc = fma(AT(oQ, i * nc + nx + m), AT(oK, m * nc + j), acc);
          for (int m = 0; m < nu; ++m) acc = fma(AT(oK, m * nc + i), AT(oR, m * nc + j), acc);
          AT(oV, i * nc + j) = acc;
        }
    }
  }
  // rollout                                                                       :160-200
  if (a.x != nullptr) {
    for (int j = 0; j < nx; ++j) AT(oX, j) = a.x_init ? a.x_init[(size_t)b * nx + j] : 0.0;
    bool bad = false;
    for (int t = 0; t < T; ++t) {
      const size_t tb = (size_t)t * B + b;
      for (int m = 0; m < nu; ++m) {
        double acc = a.ks[tb * nu + m];
        for (int j = 0; j < nx; ++j) acc = fma(a.Ks[(tb * nu + m) * nx + j], AT(oX, j), acc);
        if (a.mask != nullptr && a.mask[tb * nu + m] != 0) acc = 0.0;
        a.u[tb * nu + m] = acc;
        bad = bad || !(fabs(acc) <= 1.7e308);
      }
      for (int j = 0; j < nx; ++j) {
        a.x[tb * nx + j] = AT(oX, j);
        bad = bad || !(fabs(AT(oX, j)) <= 1.7e308);
      }
      if (t < T - 1) {
        const double *Fp = a.F + tb * nx * ns;
        for (int i = 0; i < nx; ++i) {
          double acc = a.f ? a.f[tb * nx + i] : 0.0;
          for (int j = 0; j < nx; ++j) acc = fma(Fp[i * ns + j], AT(oX, j), acc);
          for (int m = 0; m < nu; ++m) acc = fma(Fp[i * ns + nx + m], a.u[tb * nu + m], acc);
          AT(oW, i) = acc;   // (W is free here)
        }
        for (int i = 0; i < nx; ++i) AT(oX, i) = AT(oW, i);
      }
    }
    if (bad) flags |= 2;
  }
  if (a.info != nullptr && flags != 0) atomicOr(&a.info[b], flags);
#undef AT
}

struct F64Costate {
  int T, B, nx, nu;
  const double *C, *c, *F, *x, *u, *dx, *du, *gx;
  int strict;
  double *dx0, *dC, *dc, *dF, *df;
  double *ws;   // per trajectory: lam, dlam, nlam, ndlam [nx] each
};

__global__ __launch_bounds__(64) void costate_f64_kernel(const F64Costate a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  const int nx = a.nx, nu = a.nu, ns = nx + nu, T = a.T;
  const size_t B = (size_t)a.B;
  double *w = a.ws + b;
#define AT(off, i) w[((size_t)(off) + (size_t)(i)) * B]
  const size_t oL = 0, oD = nx, oNL = 2 * (size_t)nx, oND = 3 * (size_t)nx;
  const double wa = 0.5, wb = a.strict ? 0.5 : 1.0;   // differentiable_lqr.py:128 (and its symmetric variant)
  for (int i = 0; i < nx; ++i) { AT(oL, i) = 0.0; AT(oD, i) = 0.0; }
  auto tau = [&](size_t tb, int j) { return j < nx ? a.x[tb * nx + j] : a.u[tb * nu + (j - nx)]; };
  auto dtau = [&](size_t tb, int j) { return j < nx ? a.dx[tb * nx + j] : a.du[tb * nu + (j - nx)]; };
  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    if (t < T - 1) {
      if (a.dF != nullptr)
        for (int k = 0; k < nx; ++k)
          for (int j = 0; j < ns; ++j) a.dF[(tb * nx + k) * ns + j] = fma(AT(oD, k), tau(tb, j), AT(oL, k) * dtau(tb, j));
      if (a.df != nullptr && a.strict)
        for (int i = 0; i < nx; ++i) a.df[tb * nx + i] = AT(oD, i);
    }
    if (a.dC != nullptr)
      for (int i = 0; i < ns; ++i)
        for (int j = 0; j < ns; ++j)
          a.dC[(tb * ns + i) * ns + j] = fma(wa * dtau(tb, i), tau(tb, j), (wb * tau(tb, i)) * dtau(tb, j));
    if (a.dc != nullptr)
      for (int j = 0; j < ns; ++j) a.dc[tb * ns + j] = dtau(tb, j);
    for (int i = 0; i < nx; ++i) {
      double nl = a.c[tb * ns + i], ndl = a.gx[tb * nx + i];
      const double *Cr = a.C + (tb * ns + i) * ns;
      for (int j = 0; j < ns; ++j) {
        nl = fma(Cr[j], tau(tb, j), nl);
        ndl = fma(Cr[j], dtau(tb, j), ndl);
      }
      if (t < T - 1) {
        const double *Fp = a.F + tb * nx * ns + i;
        for (int k = 0; k < nx; ++k) {
          nl = fma(Fp[k * ns], AT(oL, k), nl);
          ndl = fma(Fp[k * ns], AT(oD, k), ndl);
        }
      }
      AT(oNL, i) = nl;
      AT(oND, i) = ndl;
    }
    for (int i = 0; i < nx; ++i) {
      AT(oL, i) = AT(oNL, i);
      AT(oD, i) = AT(oND, i);
      if (a.df != nullptr && !a.strict && t < T - 1) a.df[tb * nx + i] = AT(oD, i);
    }
  }
  if (a.dx0 != nullptr)
    for (int i = 0; i < nx; ++i) a.dx0[(size_t)b * nx + i] = AT(oD, i);
#undef AT
}

static size_t round256(size_t n) { return (n + 255) / 256 * 256; }

// ---- the register-resident float64 kernels (f64_row_kernels.hpp): shapes with an instantiation
// 16 lanes per trajectory (fused v_fmac_f64_dpp blocks) / a wavefront per trajectory
#define DMPC_F64_ROW_SHAPES(X)                                                                                         \
  X(1, 1, 16) X(2, 1, 16) X(3, 1, 16) X(2, 2, 16) X(3, 2, 16) X(4, 2, 16) X(6, 2, 16) X(8, 2, 16) X(4, 4, 16)         \
  X(8, 4, 16) X(12, 3, 16) X(16, 4, 64) X(16, 8, 64) X(32, 8, 64)

static constexpr size_t kMaxLds = 160 * 1024;

template <class K>
static void allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)   // above the default limit the runtime wants to be told (once per kernel; cheap to repeat)
    set_max_lds(reinterpret_cast<const void *>(kernel), (int)bytes);
}

// returns 0 / a HIP error, or DMPC_E_UNSUPPORTED when the shape has no instantiation (nothing launched)
static int launch_f64_row_solve(int nx, int nu, F64RowSolve a, hipStream_t stream) {
#define X(NX_, NU_, L_)                                                                                                \
  if (nx == NX_ && nu == NU_) {                                                                                        \
    constexpr int GPB = 256 / L_;                                                                                      \
    const size_t lds = (size_t)GPB * a.T * NU_ * (NX_ + 1) * sizeof(double);                                           \
    a.k_lds = lds <= kMaxLds ? 1 : 0;                                                                                  \
    const size_t shmem = a.k_lds ? lds : 0;                                                                            \
    const dim3 grid((a.B + GPB - 1) / GPB);                                                                            \
    if (a.mask != nullptr) {                                                                                           \
      allow_lds(lqr_f64_row_kernel<NX_, NU_, L_, true>, shmem);                                                        \
      DMPC_LAUNCH_GGL((lqr_f64_row_kernel<NX_, NU_, L_, true>), grid, dim3(256), shmem, stream, a);                    \
    } else {                                                                                                           \
      allow_lds(lqr_f64_row_kernel<NX_, NU_, L_, false>, shmem);                                                       \
      DMPC_LAUNCH_GGL((lqr_f64_row_kernel<NX_, NU_, L_, false>), grid, dim3(256), shmem, stream, a);                   \
    }                                                                                                                  \
    return (int)hipGetLastError();                                                                                     \
  }
  DMPC_F64_ROW_SHAPES(X)
#undef X
  return DMPC_E_UNSUPPORTED;
}

static int launch_f64_row_costate(int nx, int nu, const F64RowCostate &a, hipStream_t stream) {
#define X(NX_, NU_, L_)                                                                                                \
  if (nx == NX_ && nu == NU_) {                                                                                        \
    constexpr int GPB = 256 / L_;                                                                                      \
    DMPC_LAUNCH_GGL((costate_f64_row_kernel<NX_, NU_, L_>), dim3((a.B + GPB - 1) / GPB), dim3(256), 0, stream, a);    \
    return (int)hipGetLastError();                                                                                     \
  }
  DMPC_F64_ROW_SHAPES(X)
#undef X
  return DMPC_E_UNSUPPORTED;
}

// the plain fused solve of the large shapes on v_mfma_f64_16x16x4_f64 tiles (lqr_tile16_f64.hpp); DMPC_E_UNSUPPORTED - nothing
// launched - for other shapes, the clamped solve and the split-c second solve.  DMPC_NO_F64_TILE16=1: off (A/B timing).
static int launch_f64_tile16(int nx, int nu, const F64RowSolve &a, hipStream_t stream) {
  static const bool off = [] { const char *e = getenv("DMPC_NO_F64_TILE16"); return e && e[0] == '1'; }();
  if (off || a.mask != nullptr || a.Ks == nullptr || a.ks == nullptr) return DMPC_E_UNSUPPORTED;
  if (!aligned16(a.C) || !aligned16(a.c) || !aligned16(a.c_u) || !aligned16(a.F) || !aligned16(a.f)) return DMPC_E_UNSUPPORTED;
#define X(NX_, NU_)                                                                                                    \
  if (nx == NX_ && nu == NU_) {                                                                                        \
    constexpr size_t lds = Tile16F64Layout<NX_, NU_>::lds_bytes();                                                     \
    static_assert(lds <= 160 * 1024, "one workgroup per CU");                                                          \
    allow_lds(lqr_tile16_f64_kernel<NX_, NU_>, lds);                                                                   \
    DMPC_LAUNCH_GGL((lqr_tile16_f64_kernel<NX_, NU_>), dim3((a.B + 3) / 4), dim3(256), lds, stream, a);                \
    return (int)hipGetLastError();                                                                                     \
  }
  X(32, 8) X(16, 8) X(32, 4) X(16, 4)
#undef X
  return DMPC_E_UNSUPPORTED;
}

static bool f64_row_off() {
  static const bool off = [] { const char *e = getenv("DMPC_NO_F64_ROW"); return e && e[0] == '1'; }();
  return off;
}

}  // namespace dmpc

using namespace dmpc;

extern "C" {

int dmpc_lqr_f64_path(int nx, int nu) {
  if (nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!f64_row_off()) {
#define X(NX_, NU_, L_) if (nx == NX_ && nu == NU_) return L_ == 16 ? 1 : 2;
    DMPC_F64_ROW_SHAPES(X)
#undef X
  }
  return 0;
}

size_t dmpc_lqr_f64_workspace_bytes(int T, int B, int nx, int nu) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return 0;
  // solve: per-trajectory matrices; gradient: d_tau [T,B,ns], the second solve's gains [T,B,nu,nx+1], the solve's and
  // the co-state sweep's per-trajectory areas
  const size_t ns = nx + nu;
  const size_t per = f64_solve_ws_elems(nx, nu) + 4 * (size_t)nx;
  return round256(per * B * sizeof(double)) + round256((size_t)T * B * ns * sizeof(double)) +
         round256((size_t)T * B * nu * (nx + 1) * sizeof(double));
}

int dmpc_lqr_solve_f64(int T, int B, int nx, int nu, const double *C, const double *c, const double *F, const double *f,
                       const double *x_init, const uint8_t *u_zero_mask, double *Ks_out, double *ks_out, double *x_out,
                       double *u_out, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C || !c || !x_init || !x_out || !u_out || (T > 1 && !F) || !ws) return DMPC_E_BADARG;
  if ((Ks_out == nullptr) != (ks_out == nullptr)) return DMPC_E_BADARG;
  if (ws_bytes < dmpc_lqr_f64_workspace_bytes(T, B, nx, nu)) return DMPC_E_WORKSPACE;
  const size_t ns = nx + nu;
  char *p = static_cast<char *>(ws);
  double *area = reinterpret_cast<double *>(p);
  p += round256((f64_solve_ws_elems(nx, nu) + 4 * (size_t)nx) * B * sizeof(double)) + round256((size_t)T * B * ns * sizeof(double));
  double *gains = reinterpret_cast<double *>(p);
  if (!f64_row_off()) {   // the register-resident kernel where the shape has one; gains through the workspace only when LDS is short
    F64RowSolve r{T, B, C, c, F, f, x_init, nullptr, u_zero_mask, Ks_out, ks_out, x_out, u_out, info, 1};
    const int lanes = nx + nu + 1 <= 16 ? 16 : 64;
    if ((size_t)(256 / lanes) * T * nu * (nx + 1) * sizeof(double) > kMaxLds && Ks_out == nullptr) {
      r.Ks = gains;
      r.ks = gains + (size_t)T * B * nu * nx;
    }
    {   // the large shapes: 16x16x4 float64 tiles (gains through the caller's arrays or the workspace)
      F64RowSolve rt = r;
      if (rt.Ks == nullptr) {
        rt.Ks = gains;
        rt.ks = gains + (size_t)T * B * nu * nx;
      }
      rt.k_lds = 0;
      const int rct = launch_f64_tile16(nx, nu, rt, static_cast<hipStream_t>(stream));
      if (rct != DMPC_E_UNSUPPORTED) return rct;
    }
    const int rc = launch_f64_row_solve(nx, nu, r, static_cast<hipStream_t>(stream));
    if (rc != DMPC_E_UNSUPPORTED) return rc;
  }
  F64Solve a{T, B, nx, nu, C, c, F, f, x_init, u_zero_mask, Ks_out ? Ks_out : gains,
             ks_out ? ks_out : gains + (size_t)T * B * nu * nx, x_out, u_out, area, info, (int)ns, nullptr};
  DMPC_LAUNCH_GGL(lqr_f64_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

int dmpc_lqr_kkt_grad_f64(int T, int B, int nx, int nu, const double *C, const double *c, const double *F, const double *x,
                          const double *u, const double *grad_x, const double *grad_u, int strict_math, double *d_x_init,
                          double *dC, double *dc, double *dF, double *df, void *ws, size_t ws_bytes, int32_t *info,
                          dmpc_stream_t stream) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C || !c || !x || !u || !grad_x || !grad_u || !d_x_init || !ws || (T > 1 && !F)) return DMPC_E_BADARG;
  if (ws_bytes < dmpc_lqr_f64_workspace_bytes(T, B, nx, nu)) return DMPC_E_WORKSPACE;
  const size_t ns = nx + nu;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char *p = static_cast<char *>(ws);
  double *area = reinterpret_cast<double *>(p);
  p += round256((f64_solve_ws_elems(nx, nu) + 4 * (size_t)nx) * B * sizeof(double));
  double *dtau = reinterpret_cast<double *>(p);   // dx [T,B,nx] then du [T,B,nu]
  p += round256((size_t)T * B * ns * sizeof(double));
  double *gains = reinterpret_cast<double *>(p);
  double *dxs = dtau, *dus = dtau + (size_t)T * B * nx;
  // the second solve: same C, F; c = [grad_x; grad_u] (two arrays), f = 0, x_init = 0    differentiable_lqr.py:108-114
  if (!f64_row_off()) {
    F64RowSolve r{T, B, C, grad_x, F, nullptr, nullptr, grad_u, nullptr, nullptr, nullptr, dxs, dus, info, 1};
    const int lanes = nx + nu + 1 <= 16 ? 16 : 64;
    if ((size_t)(256 / lanes) * T * nu * (nx + 1) * sizeof(double) > kMaxLds) {
      r.Ks = gains;
      r.ks = gains + (size_t)T * B * nu * nx;
    }
    int rc;
    {   // the second solve on the float64 tile kernel where the shape has one (gains through the workspace)
      F64RowSolve rt = r;
      rt.Ks = gains;
      rt.ks = gains + (size_t)T * B * nu * nx;
      rt.k_lds = 0;
      rc = launch_f64_tile16(nx, nu, rt, s);
    }
    if (rc == DMPC_E_UNSUPPORTED) rc = launch_f64_row_solve(nx, nu, r, s);
    if (rc != DMPC_E_UNSUPPORTED) {
      if (rc != 0) return rc;
      F64RowCostate k{T, B, C, c, F, x, u, dxs, dus, grad_x, strict_math, d_x_init, dC, dc, dF, df};
      return launch_f64_row_costate(nx, nu, k, s);
    }
  }
  F64Solve a{T, B, nx, nu, C, grad_x, F, nullptr, nullptr, nullptr, gains, gains + (size_t)T * B * nu * nx, dxs, dus, area,
             info, 0, grad_u};
  DMPC_LAUNCH_GGL(lqr_f64_kernel, dim3((B + 63) / 64), dim3(64), 0, s, a);
  F64Costate k{T, B, nx, nu, C, c, F, x, u, dxs, dus, grad_x, strict_math, d_x_init, dC, dc, dF, df,
               area + f64_solve_ws_elems(nx, nu) * (size_t)B};
  DMPC_LAUNCH_GGL(costate_f64_kernel, dim3((B + 63) / 64), dim3(64), 0, s, k);
  return (int)hipGetLastError();
}

}  // extern "C"
