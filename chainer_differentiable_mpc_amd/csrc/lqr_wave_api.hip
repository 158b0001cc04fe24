// lqr_wave_api.hip - instances and launcher of lqr_wave_mfma_backward (LqrRecursion.backward, lqr/lqr_recursion.py:69-158,
// and LQR_active's sweep, mpc/active_constrained_lqr.py:67-151, at the large shapes).
#include <hip/hip_runtime.h>

#include "../../include/dmpc.h"
#include "api_util.hpp"
#include "lqr_wave_api.hpp"
#include "lqr_staged_forward.hpp"
#include "lqr_wave_mfma.hpp"
#include "lqr_tile16.hpp"

namespace dmpc {

// DMPC_NO_TILE16=1: the plain sweep on the 4x4x1 outer-product kernel of lqr_wave_mfma.hpp (A/B timing)
static bool tile16_disabled() {
  static const bool off = [] { const char *e = getenv("DMPC_NO_TILE16"); return e && e[0] == '1'; }();
  return off;
}

// the plain (unclamped) sweep on 16x16x4 tiles (lqr_tile16.hpp); DMPC_E_UNSUPPORTED - nothing launched - for other shapes
static int launch_lqr_tile16(int nx, int nu, bool rollout, const LqrArgs &a, hipStream_t stream) {
  const dim3 grid((a.B + 3) / 4), block(256);
#define X(NX_, NU_)                                                                                                   \
  if (nx == NX_ && nu == NU_) {                                                                                       \
    constexpr size_t lds = Tile16Layout<NX_, NU_>::lds_bytes();                                                       \
    static_assert(DMPC_T16_OCC * lds <= 160 * 1024, "DMPC_T16_OCC workgroups per CU");                                                    \
    static const bool once = [] {                                                                                     \
      set_max_lds(reinterpret_cast<const void *>(&lqr_tile16_kernel<NX_, NU_, true>), \
                                (int)lds);                                \
      set_max_lds(reinterpret_cast<const void *>(&lqr_tile16_kernel<NX_, NU_, false>), \
                                (int)lds);                                \
      return true;                                                                                                    \
    }();                                                                                                              \
    (void)once;                                                                                                       \
    if (rollout) DMPC_LAUNCH_GGL((lqr_tile16_kernel<NX_, NU_, true>), grid, block, lds, stream, a);                   \
    else DMPC_LAUNCH_GGL((lqr_tile16_kernel<NX_, NU_, false>), grid, block, lds, stream, a);                          \
    return (int)hipGetLastError();                                                                                    \
  }
  // ((16,8) stays on the 4x4x1 kernel: 388 against 384 us at B = 4096, T = 50 - profiles/r05/tile16_shapes.txt)
  X(32, 8) X(24, 8) X(32, 4) X(24, 4)
#undef X
  return DMPC_E_UNSUPPORTED;
}

int launch_lqr_wave_mfma_backward(int nx, int nu, bool masked, bool rollout, const LqrArgs &a, hipStream_t stream) {
  const dim3 grid((a.B + 3) / 4), block(256);   // four wavefronts (= trajectories) per workgroup, one per SIMD
  if (!masked && !tile16_disabled()) {
    const int rt = launch_lqr_tile16(nx, nu, rollout, a, stream);
    if (rt != DMPC_E_UNSUPPORTED) return rt;
  }
#define X(NX_, NU_)                                                                                       \
  if (nx == NX_ && nu == NU_) {                                                                           \
    if (masked && rollout) DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, true, true>), grid, block, 0, stream, a);   \
    else if (masked) DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, true, false>), grid, block, 0, stream, a);        \
    else if (rollout) DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, false, true>), grid, block, 0, stream, a);       \
    else DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, false, false>), grid, block, 0, stream, a);                   \
    return (int)hipGetLastError();                                                                        \
  }
  // (32,8) is config 5's shape; the others are the sizes of the padded instances themselves, which then run the exact kernel
  // (compile-time strides, 16-byte LDS-DMA, the rollout in the same launch) instead of their own container: (24,8) at
  // B = 4096, T = 50 1.01 -> ~0.6 ms
  X(32, 8) X(24, 8) X(24, 4) X(32, 4) X(16, 8)
#undef X
  return DMPC_E_UNSUPPORTED;
}

// MPCstep.backward_rec at the wavefront-per-trajectory shapes: the sweep with the box QP inside (a.mpc_* set)
int launch_mpc_wave_backward(int nx, int nu, const LqrArgs &a, hipStream_t stream) {
  const dim3 grid((a.B + 3) / 4), block(256);
#define X(NX_, NU_)                                                                                                \
  if (nx == NX_ && nu == NU_) {                                                                                    \
    DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, false, false, false, true>), grid, block, 0, stream, a);     \
    return (int)hipGetLastError();                                                                                 \
  }
  X(16, 8) X(32, 8)
#undef X
  return DMPC_E_UNSUPPORTED;
}

// ... of a smaller problem (a.nx_log, a.nu_log) padded inside the (cnx, cnu) instance
int launch_mpc_wave_container_backward(int cnx, int cnu, const LqrArgs &a, hipStream_t stream) {
  const dim3 grid((a.B + 3) / 4), block(256);
#define X(NX_, NU_)                                                                                                \
  if (cnx == NX_ && cnu == NU_) {                                                                                  \
    DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, false, false, true, true>), grid, block, 0, stream, a);      \
    return (int)hipGetLastError();                                                                                 \
  }
  X(16, 8) X(32, 8)
#undef X
  return DMPC_E_UNSUPPORTED;
}

// The sweep runs in the smallest instance that holds the problem (fewest columns first), whichever container the caller's
// forward-only kernel is: the gains travel at the problem's own strides.  Measured and dropped (round 4, B = 4096, T = 50):
// a test of rows and tiles at run time instead (uniform branches around the matrix-core blocks of one (32,8) instance:
// (31,8) 1.43 -> 1.62 ms), and a rollout in the same launch with one 4-byte load per row element at the problem's
// strides ((20,6) 0.96 -> 1.18 ms; the forward-only container kernel of lqr_kernels.hpp stays a launch of its own).
int launch_lqr_wave_container_sweep(int cnx, int cnu, bool masked, const LqrArgs &a, hipStream_t stream) {
  const dim3 grid((a.B + 3) / 4), block(256);
  if (a.nx_log > cnx || a.nu_log > cnu) return DMPC_E_BADARG;
#define X(NX_, NU_)                                                                                                \
  if (a.nx_log <= NX_ && a.nu_log <= NU_) {                                                                        \
    if (masked) DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, true, false, true>), grid, block, 0, stream, a);  \
    else DMPC_LAUNCH_GGL((lqr_wave_mfma_backward<NX_, NU_, false, false, true>), grid, block, 0, stream, a);        \
    return (int)hipGetLastError();                                                                                 \
  }
  X(16, 8) X(24, 4) X(24, 8) X(32, 4) X(32, 8)
#undef X
  return DMPC_E_UNSUPPORTED;
}

// the rollout of a problem padded inside those instances (gains from a.Ks / a.ks, else the workspace), inputs staged through LDS
// (lqr_staged_forward.hpp).  DMPC_E_UNSUPPORTED - nothing launched - for the clamped rollout, T == 1 or DMPC_NO_STAGED_FWD=1.
int launch_lqr_staged_forward(int nx, int nu, const LqrArgs &a, hipStream_t stream) {
  static const bool off = [] { const char *e = getenv("DMPC_NO_STAGED_FWD"); return e && e[0] == '1'; }();
  const size_t shmem = lqr_staged_fwd_lds_bytes(nx, nu);
  if (off || a.mask != nullptr || a.T < 2 || nx + nu > 63 || shmem > 150 * 1024 || (a.Ks == nullptr && a.wsK == nullptr))
    return DMPC_E_UNSUPPORTED;
  if (shmem > 64 * 1024)
    set_max_lds(reinterpret_cast<const void *>(&lqr_staged_forward_kernel), (int)shmem);
  DMPC_LAUNCH_GGL(lqr_staged_forward_kernel, dim3(a.B), dim3(64), shmem, stream, a, nx, nu);
  return (int)hipGetLastError();
}

}  // namespace dmpc
