// kkt_api.hip - analytic KKT gradient of the LQR solution (include/dmpc.h section B).
// Replaces DiffLqr.backward, lqr/differentiable_lqr.py:78-142:
//   (1) d_tau  <- LqrRecursion(0, C, [grad_x;grad_u], F, 0).solve_recursion()        (:106-114)
//   (2) lambda, d_lambda backward sweeps + outer products                             (:85-104, :115-134)
// Step (1) is the fused solve kernel of lqr_api.hip, step (2) is costate_kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/dmpc.h"
#include "api_util.hpp"
#include "costate_dma_kernel.hpp"
#include "costate_wide_kernel.hpp"
#include "costate_kernels.hpp"
#include "costate_staged_kernel.hpp"

namespace dmpc {

// drl[t][b][:] = [grad_x[t][b][:], grad_u[t][b][:]] ; x0[b][:] = 0
__global__ __launch_bounds__(256) void concat_tau_kernel(size_t n_rows, int nx, int nu, const float *__restrict__ gx,
                                                         const float *__restrict__ gu, float *__restrict__ drl,
                                                         float *__restrict__ x0, size_t n_x0) {
  const int ns = nx + nu;
  const size_t total = n_rows * ns;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t row = e / ns;
    const int j = (int)(e % ns);
    drl[e] = j < nx ? gx[row * nx + j] : gu[row * nu + (j - nx)];
  }
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_x0; e += (size_t)gridDim.x * blockDim.x)
    x0[e] = 0.f;
}

#ifdef DMPC_EXPERIMENT_ONLY_8_2
#define DMPC_COSTATE_SHAPES(X) X(8, 2, 16)
#else
#define DMPC_COSTATE_SHAPES(X) \
  X(1, 1, 16) X(2, 1, 16) X(3, 1, 16) X(2, 2, 16) X(3, 2, 16) X(4, 2, 16) X(6, 2, 16) X(8, 2, 16) \
  X(4, 4, 16) X(8, 4, 16) X(12, 3, 16) X(32, 8, 64)
#endif

#ifdef DMPC_EXPERIMENT_ONLY_8_2
#define DMPC_COSTATE_CONTAINERS(X)
#define DMPC_COSTATE_WAVE_CONTAINERS(X)
#define DMPC_COSTATE_WIDE_SHAPES(X)
#define DMPC_COSTATE_WIDE_CONTAINERS(X)
#else
#define DMPC_COSTATE_WAVE_CONTAINERS(X) X(16, 8) X(32, 8)
#define DMPC_COSTATE_WIDE_SHAPES(X) X(16, 4) X(16, 8) X(12, 8)   /* (12,4): 16 elements of tau - the 16-lane kernels' size, no instance yet */
#define DMPC_COSTATE_WIDE_CONTAINERS(X) X(16, 4) X(12, 8)
#define DMPC_COSTATE_CONTAINERS(X) X(3, 1) X(4, 4) X(8, 2) X(5, 5) X(8, 4) X(14, 1) X(13, 2) X(12, 3) X(11, 4) X(10, 5) X(9, 6) X(8, 7) X(7, 8)
#endif

static bool costate_dma_disabled() {  // DMPC_NO_COSTATE_DMA=1: register-prefetch co-state kernel (A/B timing, debugging)
  static const bool off = [] { const char *e = getenv("DMPC_NO_COSTATE_DMA"); return e && e[0] == '1'; }();
  return off;
}
constexpr int kCostateDmaDepth = 4;

bool costate_sums_available(int T, int B, int nx, int nu) {
  if (B < 4 || B % 4 != 0 || T < 2 || costate_dma_disabled()) return false;
#define X(NX_, NU_, L_) \
  if (nx == NX_ && nu == NU_) return L_ == 16;
  DMPC_COSTATE_SHAPES(X)
#undef X
  return false;
}

int launch_costate(int nx, int nu, const CostateArgs &a, hipStream_t stream) {
  if (a.dC_sum != nullptr && !costate_sums_available(a.T, a.B, nx, nu)) return DMPC_E_UNSUPPORTED;
  // the LDS-DMA kernels move 16-byte chunks and store their rows as float4: every array they touch that way must be aligned
  // (the entry points check C, c, F, dC, dF; the rest are the caller's tensors - a misaligned view takes the other kernels)
  const bool al = aligned16(a.C) && aligned16(a.c) && aligned16(a.r) && aligned16(a.F) && aligned16(a.x) && aligned16(a.u) &&
                  aligned16(a.dx) && aligned16(a.du) && aligned16(a.dC) && aligned16(a.dF);
#define X(NX_, NU_, L_)                                                                                     \
  if (nx == NX_ && nu == NU_) {                                                                             \
    constexpr int GPB = 256 / L_;                                                                           \
    if constexpr (L_ == 16) { /* inputs staged through an LDS-DMA ring (costate_dma_kernel.hpp) */          \
      if (al && a.B >= 4 && a.B % 4 == 0 && a.T >= 2 && !costate_dma_disabled()) {                                          \
        using Lay = CostateDmaLayout<NX_, NU_, kCostateDmaDepth>;                                           \
        const int waves = (a.B + 3) / 4;                                                                    \
        DMPC_LAUNCH_GGL((costate_dma_kernel<NX_, NU_, kCostateDmaDepth>), dim3((waves + 3) / 4), dim3(256), \
                           Lay::lds_bytes(), stream, a);                                                    \
        return (int)hipGetLastError();                                                                      \
      }                                                                                                     \
    }                                                                                                       \
    DMPC_LAUNCH_GGL((costate_kernel<NX_, NU_, L_>), dim3((a.B + GPB - 1) / GPB), dim3(256), 0, stream, a); \
    return (int)hipGetLastError();                                                                          \
  }
  DMPC_COSTATE_SHAPES(X)
#undef X
  // 17 to 31 elements of tau, at most 16 states: four trajectories per wavefront with tau in two registers
  // (costate_wide_kernel.hpp; before, a wavefront per trajectory inside the (16,8) container).  DMPC_NO_WIDE=1: that path.
  {
    static const bool off = [] { const char *e = getenv("DMPC_NO_WIDE"); return e && e[0] == '1'; }();
    if (!off && al && a.dC_sum == nullptr && a.B >= 4 && a.B % 4 == 0 && a.T >= 2 && !costate_dma_disabled() &&
        (size_t)a.B * (nx + nu) * (nx + nu) * 4 < ((size_t)1 << 31)) {
#define X(NX_, NU_)                                                                                            \
  if (nx == NX_ && nu == NU_) {                                                                                \
    using Lay = CostateWideLayout<NX_, NU_, 2>;                                                                \
    static_assert(Lay::lds_bytes() <= 160 * 1024, "ring and staging beyond a CU's LDS");                      \
    if (Lay::lds_bytes() > 64 * 1024)                                                                          \
      set_max_lds(reinterpret_cast<const void *>(&costate_wide_kernel<NX_, NU_, 2>), \
                                (int)Lay::lds_bytes());           \
    DMPC_LAUNCH_GGL((costate_wide_kernel<NX_, NU_, 2>), dim3((a.B + 15) / 16), dim3(256), Lay::lds_bytes(), stream, a); \
    return (int)hipGetLastError();                                                                             \
  }
      DMPC_COSTATE_WIDE_SHAPES(X)
#undef X
      // ... and padded inside the (16,4), (12,8) or (16,8) instance: the shapes without a 16-lane container (nx + nu >= 16), and
      // the larger ones of those with one - the 16-lane container stores its rows of dC / dF element by element, this kernel
      // stages them: gradient at B = 4096, T = 50 (9,4) 400 -> 295 us, (11,4) 530 -> 309, (13,2) 568 -> 321; below 13
      // elements of tau the container wins ((6,3) 208 against 247 us).  DMPC_COSTATE_WIDE_MIN_NS moves the threshold.
      static const bool no_pad = [] { const char *e = getenv("DMPC_NO_CONTAINER"); return e && e[0] == '1'; }();
      static const int min_ns = [] { const char *e = getenv("DMPC_COSTATE_WIDE_MIN_NS"); return e ? atoi(e) : 13; }();
      if (!no_pad && nx + nu >= min_ns && nx >= 1 && nu >= 1) {
        CostateArgs p = a;
        p.nx_log = nx;
        p.nu_log = nu;
#define X(NX_, NU_)                                                                                            \
  if (nx <= NX_ && nu <= NU_) {                                                                                \
    using Lay = CostateWideLayout<NX_, NU_, 2, true>;                                                          \
    static_assert(Lay::lds_bytes() <= 160 * 1024, "ring and staging beyond a CU's LDS");                      \
    if (Lay::lds_bytes() > 64 * 1024)                                                                          \
      set_max_lds(reinterpret_cast<const void *>(&costate_wide_kernel<NX_, NU_, 2, true>), \
                                (int)Lay::lds_bytes());           \
    DMPC_LAUNCH_GGL((costate_wide_kernel<NX_, NU_, 2, true>), dim3((p.B + 15) / 16), dim3(256), Lay::lds_bytes(), stream, p); \
    return (int)hipGetLastError();                                                                             \
  }
        DMPC_COSTATE_WIDE_CONTAINERS(X)
#undef X
        if (nx <= 16 && nu <= 8) {   // 13+ states with 5+ controls: the (16,8) instance, three wavefronts per workgroup (LDS)
          using Lay = CostateWideLayout<16, 8, 2, true>;
          constexpr int kWaves = 3;
          static_assert(Lay::lds_bytes(kWaves) <= 160 * 1024, "ring and staging beyond a CU's LDS");
          set_max_lds(reinterpret_cast<const void *>(&costate_wide_kernel<16, 8, 2, true, kWaves>), (int)Lay::lds_bytes(kWaves));
          DMPC_LAUNCH_GGL((costate_wide_kernel<16, 8, 2, true, kWaves>), dim3((p.B + 4 * kWaves - 1) / (4 * kWaves)),
                          dim3(64 * kWaves), Lay::lds_bytes(kWaves), stream, p);
          return (int)hipGetLastError();
        }
      }
    }
  }
  {   // a problem without a specialisation padded inside the first container that holds it (the lists of lqr_api.hip)
    static const bool off = [] { const char *e = getenv("DMPC_NO_CONTAINER"); return e && e[0] == '1'; }();
    if (!off && a.dC_sum == nullptr) {
      CostateArgs p = a;
      p.nx_log = nx;
      p.nu_log = nu;
#define X(NX_, NU_)                                                                                          \
  if (nx <= NX_ && nu <= NU_) {                                                                              \
    DMPC_LAUNCH_GGL((costate_kernel<NX_, NU_, 16, true>), dim3((p.B + 15) / 16), dim3(256), 0, stream, p);  \
    return (int)hipGetLastError();                                                                           \
  }
      DMPC_COSTATE_CONTAINERS(X)
#undef X
      {   // wider (17+ states, ragged batches of the wide shapes): a wavefront per trajectory, the step's blocks through an
          // LDS ring at the problem's own dimensions (costate_staged_kernel.hpp); DMPC_NO_STAGED_COSTATE=1: the containers below
        static const bool staged_off = [] { const char *e = getenv("DMPC_NO_STAGED_COSTATE"); return e && e[0] == '1'; }();
        const size_t shmem = costate_staged_lds_bytes(nx, nu, a.r_cols);
        if (!staged_off && a.T >= 2 && nx + nu <= 63 && shmem <= 150 * 1024) {
          if (shmem > 64 * 1024)
            set_max_lds(reinterpret_cast<const void *>(&costate_staged_kernel), (int)shmem);
          DMPC_LAUNCH_GGL(costate_staged_kernel, dim3(a.B), dim3(64), shmem, stream, a, nx, nu);
          return (int)hipGetLastError();
        }
      }
#define X(NX_, NU_)                                                                                          \
  if (nx <= NX_ && nu <= NU_) {   /* wider: a wavefront per trajectory */                                     \
    DMPC_LAUNCH_GGL((costate_kernel<NX_, NU_, 64, true>), dim3((p.B + 3) / 4), dim3(256), 0, stream, p);     \
    return (int)hipGetLastError();                                                                           \
  }
      DMPC_COSTATE_WAVE_CONTAINERS(X)
#undef X
    }
  }
  // runtime dimensions, a wavefront per trajectory, vectors in LDS: any size the vectors fit (the reference has no limit)
  const size_t shmem = (size_t)(2 * (nx + nu) + 4 * nx) * sizeof(float);
  if (shmem > 64 * 1024) return DMPC_E_UNSUPPORTED;
  DMPC_LAUNCH_GGL(costate_generic_kernel, dim3(a.B), dim3(64), shmem, stream, a, CostateDims{nx, nu});
  return (int)hipGetLastError();
}

struct KktWs {
  size_t drl, x0, dx, du, lqr, total;
};
static KktWs kkt_layout(int T, int B, int nx, int nu) {
  const size_t ns = nx + nu;
  KktWs w;
  size_t off = 0;
  auto take = [&](size_t floats) {
    const size_t o = off;
    off += round_up(floats * sizeof(float), 256);
    return o;
  };
  w.drl = take((size_t)T * B * ns);
  w.x0 = take((size_t)B * nx);
  w.dx = take((size_t)T * B * nx);
  w.du = take((size_t)T * B * nu);
  w.lqr = off;
  off += round_up(dmpc_lqr_workspace_bytes(T, B, nx, nu), 256);
  w.total = off;
  return w;
}

}  // namespace dmpc

using namespace dmpc;

extern "C" {

size_t dmpc_lqr_kkt_workspace_bytes(int T, int B, int nx, int nu) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return 0;
  return kkt_layout(T, B, nx, nu).total;
}

// DiffLqr.backward.  Ks != nullptr: the gains of the forward solve are reused (dmpc_lqr_kkt_grad_saved).
static bool adjoint_disabled() {  // DMPC_NO_ADJOINT=1: the re-solve + co-state kernels instead of the one-pass gradient (A/B)
  static const bool off = [] { const char *e = getenv("DMPC_NO_ADJOINT"); return e && e[0] == '1'; }();
  return off;
}

static int kkt_grad(int T, int B, int nx, int nu, const float *C, const float *c, const float *F, const float *x,
                    const float *u, const float *Ks, const float *Quu, const float *Qxu, const float *Vv,
                    const float *grad_x, const float *grad_u, int strict_math, float *d_x_init, float *dC, float *dc,
                    float *dF, float *df, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C || !c || !F || !x || !u || !grad_x || !grad_u || !d_x_init || !dc || !ws) return DMPC_E_BADARG;
  if (!aligned16(C) || !aligned16(c) || !aligned16(F) || !aligned16(dC) || !aligned16(dF)) return DMPC_E_BADARG;
  if (Ks != nullptr && Vv != nullptr && dC != nullptr && dF != nullptr && df != nullptr && !adjoint_disabled() &&
      aligned16(grad_x) && aligned16(grad_u) && aligned16(Ks) && aligned16(Quu) && aligned16(Qxu) && aligned16(Vv) &&
      aligned16(x) && aligned16(u) && aligned16(dc) && aligned16(df)) {
    // one launch, no C: the affine re-solve whose rollout writes the gradients (lqr_adjoint, lqr_api.hip)
    const int rc1 = lqr_adjoint(T, B, nx, nu, F, grad_x, grad_u, Ks, Quu, Qxu, Vv, x, u, strict_math, d_x_init, dC, dc, dF, df,
                                info, static_cast<hipStream_t>(stream_));
    if (rc1 != DMPC_E_UNSUPPORTED) return rc1;
  }
  const KktWs w = kkt_layout(T, B, nx, nu);
  if (ws_bytes < w.total) return DMPC_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char *base = static_cast<char *>(ws);
  float *drl = reinterpret_cast<float *>(base + w.drl);
  float *x0 = reinterpret_cast<float *>(base + w.x0);
  float *dx = reinterpret_cast<float *>(base + w.dx);
  float *du = reinterpret_cast<float *>(base + w.du);
  // (1) second LQR solve: x_init = 0, c = [grad_x; grad_u], f = 0 (a NULL f is the same recursion, lqr_recursion.py:90-96).
  // The generated streams take the two gradient arrays as they are (and x_init = 0 without a buffer of zeros) ...
  const float *r = grad_x;
  int r_cols = nx;
  int rc = DMPC_E_UNSUPPORTED;
  if (aligned16(grad_x) && aligned16(grad_u) && (Ks == nullptr || (aligned16(Ks) && aligned16(Quu) && aligned16(Qxu))))
    rc = lqr_second_solve(T, B, nx, nu, C, grad_x, grad_u, F, Ks, Quu, Qxu, dx, du, info, stream);
  if (rc == DMPC_E_UNSUPPORTED) {
    if (Ks != nullptr) return rc;   // nothing has been launched: the caller goes on with dmpc_lqr_kkt_grad
    // ... the other kernels a concatenated copy
    const size_t rows = (size_t)T * B;
    const int blocks = (int)((rows * (nx + nu) + 255) / 256 > 4096 ? 4096 : (rows * (nx + nu) + 255) / 256);
    DMPC_LAUNCH_GGL(concat_tau_kernel, dim3(blocks), dim3(256), 0, stream, rows, nx, nu, grad_x, grad_u, drl, x0,
                       (size_t)B * nx);
    rc = dmpc_lqr_solve(T, B, nx, nu, C, drl, F, nullptr, x0, nullptr, nullptr, nullptr, dx, du, base + w.lqr,
                        w.total - w.lqr, info, stream_);
    r = drl;
    r_cols = 0;
  }
  if (rc != 0) return rc;
  // (2) co-state sweeps and outer products
  CostateArgs a{T, B, C, c, F, x, u, dx, du, r, 1.0f, 1.0f, strict_math ? 1 : 0, strict_math ? 1 : 0,
                d_x_init, dC, dc, dF, df};
  a.r_cols = r_cols;
  return launch_costate(nx, nu, a, stream);
}

int dmpc_lqr_kkt_grad(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                      const float *x, const float *u, const float *grad_x, const float *grad_u,
                      int strict_math, float *d_x_init, float *dC, float *dc, float *dF, float *df, void *ws,
                      size_t ws_bytes, int32_t *info, dmpc_stream_t stream) {
  return kkt_grad(T, B, nx, nu, C, c, F, x, u, nullptr, nullptr, nullptr, nullptr, grad_x, grad_u, strict_math, d_x_init, dC,
                  dc, dF, df, ws, ws_bytes, info, stream);
}

// DiffLqr.backward with the gains of the forward solve (dmpc_lqr_solve_saving): the second solve shares C and F with it,
// so K_t, Quu_t, Qxu_t are the same and only the affine recursion is redone (the `affine` stream of gen_lqr_asm.py).
// With Vv (the saving solve's value functions) the whole gradient is ONE launch that reads neither C nor c: co-states are
// value gradients, lambda_t = V_t x_t + v_t, d_lambda_t = V_t dx_t + v'_t (the `adj` stream of gen_lqr_asm.py).
int dmpc_lqr_kkt_grad_saved(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                            const float *x, const float *u, const float *Ks, const float *Quu, const float *Qxu,
                            const float *Vv, const float *grad_x, const float *grad_u, int strict_math, float *d_x_init,
                            float *dC, float *dc, float *dF, float *df, void *ws, size_t ws_bytes, int32_t *info,
                            dmpc_stream_t stream) {
  if (!Ks || !Quu || !Qxu) return DMPC_E_BADARG;
  return kkt_grad(T, B, nx, nu, C, c, F, x, u, Ks, Quu, Qxu, Vv, grad_x, grad_u, strict_math, d_x_init, dC, dc, dF, df, ws,
                  ws_bytes, info, stream);
}

}  // extern "C"
