// mpc_api.hip - C-ABI entry points of the projected-Newton QP and of the MPC step
// (include/dmpc.h sections C and E).  Replaces PNQP (mpc/pnqp.py:37-201), MPCstep.forward /
// backward_rec / forward_rec / backward (mpc/mpc_step.py:70-460) of the reference.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>

#include "../../include/dmpc.h"
#include "api_util.hpp"
#include "costate_args.hpp"
#include "box_ddp_kernels.hpp"
#include "mpc_asm_kernel.hpp"
#include "mpc_dma_kernels.hpp"
#include "mpc_fwd_asm_kernel.hpp"
#include "mpc_step_fused_kernel.hpp"
#include "lqr_wide_kernel.hpp"
#include "mpc_wide_forward_kernel.hpp"
#include "lqr_wave_api.hpp"
#include "mpc_generic.hpp"
#include "mpc_tiled.hpp"
#include "mpc_coupled.hpp"
#include "mpc_kernels.hpp"

namespace dmpc {

// Standalone PNQP: one lane per QP, everything in registers.
struct PnqpArgs {
  int B;
  const float *H, *q, *lower, *upper, *x_init;
  int n_iter;
  float *x_out, *fac;
  int32_t *piv;
  float *index_f;
  int32_t *n_iter_out, *info;
  unsigned *sync;   // batch-coupled termination (pnqp_device.hpp): pnqp_sync_slots(n_iter) zeroed slots, or nullptr
};

template <int N>
__global__ __launch_bounds__(256) void pnqp_kernel(const PnqpArgs a) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < a.B;
  if (!live) {
    if (a.sync == nullptr) return;
    b = a.B - 1;   // coupled: padding lanes repeat the last row (neutral in the any-reductions) and take every barrier
  }
  float Hm[N][N], qv[N], lo[N], hi[N], x[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int c = 0; c < N; ++c) Hm[r][c] = a.H[((size_t)b * N + r) * N + c];
    qv[r] = a.q[(size_t)b * N + r];
    lo[r] = a.lower[(size_t)b * N + r];
    hi[r] = a.upper[(size_t)b * N + r];
    x[r] = a.x_init != nullptr ? a.x_init[(size_t)b * N + r] : 0.f;
  }
  PnqpResult<N> res;
  QpTermination term;
  term.slots = a.sync;
  term.n_blocks = gridDim.x;
  pnqp_solve<N>(Hm, qv, lo, hi, x, a.x_init != nullptr, a.n_iter, res, term);
  if (!live) return;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    a.x_out[(size_t)b * N + r] = x[r];
    a.index_f[(size_t)b * N + r] = res.free_[r] ? 1.0f : 0.0f;
    if (a.piv != nullptr) a.piv[(size_t)b * N + r] = res.piv[r];
#pragma unroll
    for (int c = 0; c < N; ++c) a.fac[((size_t)b * N + r) * N + c] = res.fac[r][c];
  }
  a.n_iter_out[b] = res.it;
  if (a.info != nullptr && !res.converged) atomicOr(&a.info[b], DMPC_INFO_QP_ITERCAP);
}

// A kernel whose workgroups meet at grid barriers needs all of them resident: cooperative launch (the runtime refuses a grid
// that does not fit instead of letting it deadlock).  The runtime's own limit - hipOccupancyMaxActiveBlocksPerMultiprocessor x
// CUs - is not safe on gfx950: mpc_coupled.hpp's kernel (7 workgroups per CU by that count) was ACCEPTED at 1,792 workgroups
// and deadlocked at its first barrier, and ran at 1,024 (round 5, scripts/coupled_grid_probe.py).  So the grids here stay at
// HALF the workgroups per CU the runtime counts (at least one).
static long long resident_workgroups(const void *kernel, int threads, size_t lds) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || cus <= 0 || per_cu <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  return (long long)cus * std::max(1, per_cu / 2);
}

static int launch_cooperative(const void *kernel, dim3 grid, dim3 block, void **args, size_t lds, hipStream_t stream) {
  if ((long long)grid.x > resident_workgroups(kernel, (int)block.x, lds)) return DMPC_E_UNSUPPORTED;   // -> mpc_coupled.hpp's fixed grid
  note_kernel(kernel);
  const hipError_t e = hipLaunchCooperativeKernel(kernel, grid, block, args, (unsigned)lds, stream);
  if (e == hipErrorCooperativeLaunchTooLarge) {
    (void)hipGetLastError();
    return DMPC_E_UNSUPPORTED;
  }
  return (int)e;
}

// DMPC_NO_COOP_REGISTER=1 (read at every call: tests switch it): batch-coupled problems skip the register kernels'
// whole-batch-resident launch and run on mpc_coupled.hpp's fixed grid, as they do when that launch does not fit
static bool coupled_register_disabled() {
  const char *e = getenv("DMPC_NO_COOP_REGISTER");
  return e && e[0] == '1';
}

// mpc_coupled.hpp's kernels: a grid that is resident whatever the batch - resident_workgroups() of this kernel, at most one per
// row; halved if the runtime still refuses
static int launch_fixed_grid(const void *kernel, int rows, void **args, hipStream_t stream) {
  long long cap = resident_workgroups(kernel, kTiledThreads, 0);
  if (cap <= 0) return DMPC_E_UNSUPPORTED;
  if (const char *e = getenv("DMPC_FIXED_GRID_MAX")) {   // (experiments; read at every call)
    const long long v = atoll(e);
    if (v >= 1) cap = v;
  }
  note_kernel(kernel);
  for (long long g = std::min<long long>(cap, rows); g >= 1; g /= 2) {
    const hipError_t e = hipLaunchCooperativeKernel(kernel, dim3((unsigned)g), dim3(kTiledThreads), args, 0, stream);
    if (e != hipErrorCooperativeLaunchTooLarge) return (int)e;
    (void)hipGetLastError();
  }
  return DMPC_E_UNSUPPORTED;
}

static size_t coupled_bytes(int T, int n_qp_iter) { return round_up((size_t)T * pnqp_sync_slots(n_qp_iter) * 2 * sizeof(unsigned), 256); }
// the fixed-grid form's rows, behind the decision slots
static size_t coupled_traj_bytes(int B, int nx, int nu) { return round_up((size_t)B * mpc_coupled_traj_floats(nx, nu) * sizeof(float), 256); }
static size_t pnqp_coupled_row_bytes(int B, int n) { return round_up((size_t)B * pnqp_coupled_row_floats(n) * sizeof(float), 256); }

// MPCstep.backward_rec, batch-coupled, any size and any batch (mpc_coupled.hpp); a.sync zeroed, a.tiled_scratch =
// [B][mpc_coupled_traj_floats]
static int launch_mpc_back_fixed_grid(int nx, int nu, const MpcBackArgs &a, hipStream_t stream) {
  if (a.sync == nullptr) return DMPC_E_BADARG;
  if (a.tiled_scratch == nullptr) return DMPC_E_WORKSPACE;
  MpcBackArgs aa = a;
  float *scratch = a.tiled_scratch;
  void *args[] = {&aa, &nx, &nu, &scratch};
  return launch_fixed_grid(reinterpret_cast<const void *>(&mpc_coupled_backward_kernel), a.B, args, stream);
}

#ifdef DMPC_EXPERIMENT_ONLY_8_2
#define DMPC_MPC_SHAPES(X) X(8, 2, 16)
#else
#define DMPC_MPC_SHAPES(X) \
  X(1, 1, 16) X(2, 1, 16) X(3, 1, 16) X(2, 2, 16) X(3, 2, 16) X(4, 2, 16) X(6, 2, 16) X(8, 2, 16) \
  X(4, 4, 16) X(8, 4, 16) X(12, 3, 16)
#endif

// Containers (mpc_backward_rec_kernel / mpc_forward_rec_kernel <..., PAD>): any smaller problem runs padded inside them
// instead of on the runtime-dimension kernels of mpc_generic.hpp; tried in this order (same list as lqr_api.hip)
#ifdef DMPC_EXPERIMENT_ONLY_8_2
#define DMPC_MPC_CONTAINERS(X)
#else
#define DMPC_MPC_CONTAINERS(X) X(3, 1) X(4, 4) X(8, 2) X(8, 4) X(14, 1) X(13, 2) X(12, 3) X(11, 4)
#endif
static bool mpc_container_disabled() {   // DMPC_NO_CONTAINER=1: the runtime-dimension kernels (A/B timing)
  static const bool off = [] { const char *e = getenv("DMPC_NO_CONTAINER"); return e && e[0] == '1'; }();
  return off;
}

static bool mpc_staged_forward_disabled() {   // DMPC_NO_MPC_STAGED_FWD=1: mpc_generic_forward_kernel (A/B timing, parity between them)
  static const bool off = [] { const char *e = getenv("DMPC_NO_MPC_STAGED_FWD"); return e && e[0] == '1'; }();
  return off;
}

// Which shapes run padded inside the wide row kernels' instances (sweep and line search alike): those without a 16-lane MPC
// container (16 and more elements of tau, or more than four controls) and, of those with one, three and four controls from
// 12 elements on - MPCstep.forward at B = 4096, T = 50: (9,4) 707 -> 605 us, (11,4) 769 -> 645, (10,3) 629 -> 578; with one
// or two controls the container's short QP wins ((13,2) 568 against 695 us, (14,1) 460 against 562).
static bool mpc_wide_padded(int nx, int nu) { return nx + nu >= 16 || nu > 4 || (nu >= 3 && nx + nu >= 12); }

static bool mpc_wave_disabled() {   // DMPC_NO_MPC_WAVE=1: wide MPC shapes on the runtime-dimension kernel (A/B timing)
  static const bool off = [] { const char *e = getenv("DMPC_NO_MPC_WAVE"); return e && e[0] == '1'; }();
  return off;
}
static bool mpc_dma_disabled() {
  static const bool off = [] { const char *e = getenv("DMPC_NO_MPC_DMA"); return e && e[0] == '1'; }();
  return off;
}
static bool mpc_asm_disabled() {
  static const bool off = [] { const char *e = getenv("DMPC_NO_MPC_ASM"); return e && e[0] == '1'; }();
  return off;
}
template <class... P>
static bool aligned16(const P *...p) {   // nullptr counts as aligned
  return ((reinterpret_cast<uintptr_t>(p) | ...) & 15u) == 0;
}

// workgroups that share the bookkeeping of one box-DDP iteration inside the fused sweep launch: 256 rows each
static int select_parts(int B) { const int n = (B + 255) / 256; return n < 1 ? 1 : (n > 8 ? 8 : n); }

// can launch_mpc_back stage this problem's inputs through the LDS-DMA ring?
static bool mpc_back_dma_ok(const MpcBackArgs &a) {
  return a.sync == nullptr && a.B >= 4 && a.B % 4 == 0 && !mpc_dma_disabled() &&
         aligned16(a.C, a.c, a.F, a.f, a.controls, a.lower, a.upper, a.states);
}

// sel != nullptr (nx = 3, nu = 1 and mpc_back_dma_ok only): one more workgroup runs box_ddp_select_body(*sel)
static int launch_mpc_back(int nx, int nu, const MpcBackArgs &a_in, hipStream_t stream, const DdpSelectArgs *sel = nullptr,
                           unsigned *sel_sync = nullptr) {
  MpcBackArgs a = a_in;
  if (a.sync != nullptr) {   // fresh decision slots for this launch
    const hipError_t e = hipMemsetAsync(a.sync, 0, coupled_bytes(a.T, a.n_qp_iter), stream);
    if (e != hipSuccess) return (int)e;
  }
  void *args1[] = {&a};
  const bool coop_off = a.sync != nullptr && coupled_register_disabled();
  // per-trajectory termination, whole wavefronts of four trajectories, 16-byte aligned runs: inputs through the LDS-DMA
  // ring of mpc_dma_kernels.hpp (DMPC_NO_MPC_DMA=1: the register-bank kernel, for A/B timing)
  const bool dma_ok = mpc_back_dma_ok(a);
  // The sweep as one generated instruction stream with the box QP inside (mpc_asm_kernel.hpp): shapes the generator
  // covers, c already re-centred, no bookkeeping riding along.  DMPC_NO_MPC_ASM=1: the HIP kernels (A/B timing).
  if (dma_ok && a.T >= 2 && !mpc_asm_disabled() && (sel == nullptr || sel_sync != nullptr) &&
      (a.states == nullptr || a.f == nullptr)) {
    const int n_sel = sel != nullptr ? select_parts(a.B) : 0;
    const dim3 grid((a.B + 15) / 16 + n_sel), block(256);
    const int variant = a.states != nullptr ? 2 : (a.f != nullptr ? 1 : 0);   // re-centring / with f_hat / plain
#define L_(K_, ...)                                                                                            \
  if (sel != nullptr) DMPC_LAUNCH_GGL((mpc_backward_asm_select_kernel<__VA_ARGS__>), grid, block, K_, stream, a, *sel, n_sel, sel_sync); \
  else DMPC_LAUNCH_GGL((mpc_backward_asm_kernel<__VA_ARGS__>), grid, block, K_, stream, a);
#define A(NX_, NU_)                                                                                            \
  if (nx == NX_ && nu == NU_) {                                                                                \
    constexpr size_t lds = mpc_asm_lds_bytes<NX_, NU_>();                                                      \
    if (variant == 2) { L_(lds, NX_, NU_, false, true) }                                                       \
    else if (variant == 1) { L_(lds, NX_, NU_, true, false) }                                                  \
    else { L_(lds, NX_, NU_, false, false) }                                                                   \
    return (int)hipGetLastError();                                                                             \
  }
    A(8, 2) A(3, 1) A(4, 2) A(2, 2) A(1, 1) A(2, 1) A(3, 2)
#undef A
#undef L_
  }
#define X(NX_, NU_, L_)                                                                                       \
  if (nx == NX_ && nu == NU_) {                                                                               \
    constexpr int GPB = 256 / L_;                                                                             \
    if constexpr (L_ == 16) {                                                                                 \
      constexpr int kD = MpcBackDmaLayout<NX_, NU_, 2>::kDma, DB_ = kD == 1 ? 8 : kD <= 4 ? 4 : 2;            \
      if constexpr (MpcBackDmaLayout<NX_, NU_, DB_>::lds_bytes() <= 65536) {                                  \
        if (dma_ok && sel != nullptr) {                                                                       \
          const int n_sel = select_parts(a.B);                                                                \
          DMPC_LAUNCH_GGL((mpc_backward_rec_dma_select_kernel<NX_, NU_, DB_>), dim3((a.B + 15) / 16 + n_sel), \
                             dim3(256), (MpcBackDmaLayout<NX_, NU_, DB_>::lds_bytes()), stream, a, *sel, n_sel, \
                             sel_sync);                                                                       \
          return (int)hipGetLastError();                                                                      \
        }                                                                                                     \
        if (dma_ok) {                                                                                         \
          DMPC_LAUNCH_GGL((mpc_backward_rec_dma_kernel<NX_, NU_, DB_>), dim3((a.B + 15) / 16), dim3(256),  \
                             (MpcBackDmaLayout<NX_, NU_, DB_>::lds_bytes()), stream, a);                      \
          return (int)hipGetLastError();                                                                      \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    if (a.sync != nullptr) {                                                                                  \
      const int rc = coop_off ? DMPC_E_UNSUPPORTED                                                            \
                              : launch_cooperative(reinterpret_cast<const void *>(&mpc_backward_rec_kernel<NX_, NU_, L_>), \
                                                   dim3((a.B + GPB - 1) / GPB), dim3(256), args1, 0, stream); \
      return rc == DMPC_E_UNSUPPORTED ? launch_mpc_back_fixed_grid(nx, nu, a, stream) : rc;                   \
    }                                                                                                         \
    DMPC_LAUNCH_GGL((mpc_backward_rec_kernel<NX_, NU_, L_>), dim3((a.B + GPB - 1) / GPB), dim3(256), 0, stream, \
                       a);                                                                                    \
    return (int)hipGetLastError();                                                                            \
  }
  if (sel != nullptr && !(dma_ok && sel_sync != nullptr)) return DMPC_E_BADARG;   // callers ask mpc_back_dma_ok first
  DMPC_MPC_SHAPES(X)
#undef X
  // (12,4), (16,4), (12,8), (16,8): the sweep on the wide row kernel with the box QP inside (lqr_wide_kernel<..., MPC>: four
  // trajectories per wavefront, matrix-core products; before, the runtime-dimension kernel - 2.0 ms at (12,4), B = 4096, T = 50 -
  // and, at (16,8), a wavefront per trajectory).  DMPC_NO_WIDE=1: those.
  {
    static const bool wide_off = [] { const char *e = getenv("DMPC_NO_WIDE"); return e && e[0] == '1'; }();
    if (!wide_off && a.sync == nullptr && sel == nullptr && a.B >= 4 && a.T >= 2 &&
        aligned16(a.C, a.c, a.F, a.f) && (size_t)a.B * (nx + nu) * (nx + nu) * 4 < ((size_t)1 << 31)) {
      LqrArgs s{a.T, a.B, a.C, a.c, a.F, a.f, nullptr, nullptr, a.Ks, a.ks, nullptr, nullptr, nullptr, nullptr, a.info};
      s.mpc_controls = a.controls;
      s.mpc_lower = a.lower;
      s.mpc_upper = a.upper;
      s.mpc_states = a.states;
      s.mpc_n_qp_iter = a.n_qp_iter;
      s.mpc_n_qp_total = a.n_qp_total;
      s.mpc_done = a.done;
      s.info_store = a.info_store != 0;
#define X(NX_, NU_)                                                                                            \
  if (nx == NX_ && nu == NU_) {                                                                                \
    constexpr size_t lds = LqrWideLayout<NX_, NU_, 2, 2>::lds_bytes();                                         \
    if (lds > 64 * 1024)                                                                                       \
      set_max_lds(reinterpret_cast<const void *>(&lqr_wide_kernel<NX_, NU_, 2, 2, false, false, true>), \
                                (int)lds);                         \
    DMPC_LAUNCH_GGL((lqr_wide_kernel<NX_, NU_, 2, 2, false, false, true>), dim3((s.B + 15) / 16), dim3(256), lds, stream, s); \
    return (int)hipGetLastError();                                                                             \
  }
      X(12, 4) X(16, 4) X(12, 8) X(16, 8)
#undef X
      // ... and padded inside the smallest of those instances: every other shape with at most 16 states, 8 controls and
      // no 16-lane container (nx + nu >= 16, or more than four controls: (5,5) 2.0 -> 1.3 ms per MPCstep.forward), whole wavefronts
      if (nx >= 1 && nu >= 1 && nx <= 16 && nu <= 8 && mpc_wide_padded(nx, nu) && a.B % 4 == 0 && !mpc_container_disabled()) {
        s.nx_log = nx;
        s.nu_log = nu;
#define X(NX_, NU_)                                                                                            \
  if (nx <= NX_ && nu <= NU_) {                                                                                \
    constexpr size_t lds = LqrWideLayout<NX_, NU_, 2, 2>::lds_bytes();                                         \
    if (lds > 64 * 1024)                                                                                       \
      set_max_lds(reinterpret_cast<const void *>(&lqr_wide_kernel<NX_, NU_, 2, 2, true, false, true>), \
                                (int)lds);                         \
    DMPC_LAUNCH_GGL((lqr_wide_kernel<NX_, NU_, 2, 2, true, false, true>), dim3((s.B + 15) / 16), dim3(256), lds, stream, s); \
    return (int)hipGetLastError();                                                                             \
  }
        X(12, 4) X(16, 4) X(12, 8) X(16, 8)
#undef X
      }
    }
  }
  // (16,8), (32,8): the matrix-core sweep with the box QP inside (lqr_wave_mfma_backward<..., MPC>); per-trajectory
  // termination, not inside the device-driven BoxDDP loop (no `done` flag there)
  if (a.sync == nullptr && a.done == nullptr && !a.info_store && ((nx == 16 && nu == 8) || (nx == 32 && nu == 8)) &&
      a.T >= 1 && !mpc_wave_disabled()) {
    LqrArgs s{a.T, a.B, a.C, a.c, a.F, a.f, nullptr, nullptr, a.Ks, a.ks, nullptr, nullptr, nullptr, nullptr, a.info};
    s.mpc_controls = a.controls;
    s.mpc_lower = a.lower;
    s.mpc_upper = a.upper;
    s.mpc_states = a.states;
    s.mpc_n_qp_iter = a.n_qp_iter;
    s.mpc_n_qp_total = a.n_qp_total;
    return launch_mpc_wave_backward(nx, nu, s, stream);
  }
  auto wave_container = [&](int cnx, int cnu) {   // wider than the 16-lane containers: padded inside a wave instance
    LqrArgs s{a.T, a.B, a.C, a.c, a.F, a.f, nullptr, nullptr, a.Ks, a.ks, nullptr, nullptr, nullptr, nullptr, a.info};
    s.mpc_controls = a.controls;
    s.mpc_lower = a.lower;
    s.mpc_upper = a.upper;
    s.mpc_states = a.states;
    s.mpc_n_qp_iter = a.n_qp_iter;
    s.mpc_n_qp_total = a.n_qp_total;
    s.nx_log = nx;
    s.nu_log = nu;
    return launch_mpc_wave_container_backward(cnx, cnu, s, stream);
  };
  const bool wave_ok = a.sync == nullptr && a.done == nullptr && !a.info_store && !mpc_wave_disabled() && !mpc_container_disabled();
  const bool small = nu <= 4 && nx + nu <= 15;     // (the 16-lane containers below take these)
  if (!mpc_container_disabled()) {   // a smaller problem inside the first container that holds it
    a.nx_log = nx;
    a.nu_log = nu;
#define X(NX_, NU_)                                                                                           \
  if (nx <= NX_ && nu <= NU_) {                                                                               \
    if (a.sync != nullptr) {                                                                                  \
      const int rc = coop_off ? DMPC_E_UNSUPPORTED                                                            \
                              : launch_cooperative(reinterpret_cast<const void *>(&mpc_backward_rec_kernel<NX_, NU_, 16, true>), \
                                                   dim3((a.B + 15) / 16), dim3(256), args1, 0, stream);       \
      a.nx_log = a.nu_log = 0;                                                                                \
      return rc == DMPC_E_UNSUPPORTED ? launch_mpc_back_fixed_grid(nx, nu, a, stream) : rc;                   \
    }                                                                                                         \
    DMPC_LAUNCH_GGL((mpc_backward_rec_kernel<NX_, NU_, 16, true>), dim3((a.B + 15) / 16), dim3(256), 0, stream, a); \
    return (int)hipGetLastError();                                                                            \
  }
    DMPC_MPC_CONTAINERS(X)
#undef X
  }
  // A padded wave instance costs what ITS shape costs (and fetches element by element, one wavefront per SIMD): measured at
  // B = 4096, T = 50 it beats the runtime-dimension kernel where the latter's QP in lane 0 is widest - (24,8) 9.2 against
  // 12.5 ms - and lost below - (20,6) 7.8 against 5.4 ms, (10,5) 5.4 against 2.9 ms.
  // (Round 4, the padded instance fetching through its LDS-DMA slot: (20,6) 3.4 against 3.8 ms - six controls and more.)
  // (Round 5, from 21 states on with three to five controls, from 24 with one or two: the runtime-dimension kernel's cost grows with nx^2 in one lane group -
  // (32,4) 7.7 ms, (32,2) 6.4, (23,4) 4.2, (24,4) 3.9 - where the padded (32,8) instance stays at ~3 ms: (32,4) 3.1, (32,2) 2.8, (24,4) 3.0.)
  if (wave_ok && !small && nu >= 6 && nx <= 16 && nu <= 8) return wave_container(16, 8);
  if (wave_ok && !small && (nu >= 6 || nx >= (nu >= 3 ? 21 : 24)) && nx <= 32 && nu <= 8) return wave_container(32, 8);
  // any other shape with nx + nu + 1 <= 64, nu <= 8: runtime-dimension kernel (mpc_generic.hpp)
  if (nx + nu + 1 <= 64 && nu <= kMpcGenericMaxNu) {
    void *args2[] = {&a, &nx};
#define G(NU_)                                                                                                    \
  case NU_:                                                                                                       \
    if (a.sync != nullptr) {                                                                                      \
      const int rc = coop_off ? DMPC_E_UNSUPPORTED                                                                \
                              : launch_cooperative(reinterpret_cast<const void *>(&mpc_generic_backward_kernel<NU_>), dim3(a.B), \
                                                   dim3(64), args2, mpc_generic_back_lds_bytes<NU_>(nx), stream); \
      return rc == DMPC_E_UNSUPPORTED ? launch_mpc_back_fixed_grid(nx, nu, a, stream) : rc;                       \
    }                                                                                                             \
    DMPC_LAUNCH_GGL((mpc_generic_backward_kernel<NU_>), dim3(a.B), dim3(64), mpc_generic_back_lds_bytes<NU_>(nx), \
                       stream, a, nx);                                                                            \
    return (int)hipGetLastError();
    switch (nu) { G(1) G(2) G(3) G(4) G(5) G(6) G(7) G(8) }
#undef G
  }
  // any other size (more than 8 controls, more than 64 columns): a workgroup per trajectory, matrices in the caller's
  // workspace (mpc_tiled.hpp; batch-coupled: mpc_coupled.hpp)
  if (a.sync != nullptr) {
    a.nx_log = a.nu_log = 0;
    return launch_mpc_back_fixed_grid(nx, nu, a, stream);
  }
  if (a.tiled_scratch != nullptr) {
    const size_t shmem = (pnqp_tiled_lds_floats(nu) + (size_t)(nx + nu)) * sizeof(float);
    if (shmem > 60 * 1024) return DMPC_E_UNSUPPORTED;
    DMPC_LAUNCH_GGL(mpc_tiled_backward_kernel, dim3(a.B), dim3(kTiledThreads), shmem, stream, a, nx, nu, a.tiled_scratch);
    return (int)hipGetLastError();
  }
  return DMPC_E_WORKSPACE;
}

// shapes that only the tiled kernels take
static bool mpc_needs_tiles(int nx, int nu) { return !(nx + nu + 1 <= 64 && nu <= kMpcGenericMaxNu); }

static bool spec_line_search_disabled() {  // DMPC_NO_SPEC_LS=1: sequential line search for the pendulum (A/B timing)
  static const bool off = [] { const char *e = getenv("DMPC_NO_SPEC_LS"); return e && e[0] == '1'; }();
  return off;
}

static bool spec4_disabled_early() {  // (DMPC_NO_SPEC4 also selects the lane-per-trajectory rollout)
  static const bool off = [] { const char *e = getenv("DMPC_NO_SPEC4"); return e && e[0] == '1'; }();
  return off;
}
// rollout + linearisation of the built-in pendulum: four lanes per trajectory when the 16-byte row accesses are aligned
static void launch_pendulum_rollout(const PendulumArgs &pa, hipStream_t stream) {
  if (aligned16(pa.F, pa.C) && !spec4_disabled_early())
    DMPC_LAUNCH_GGL(pendulum_rollout_linearize4_kernel, dim3((4 * pa.B + 255) / 256), dim3(256), 0, stream, pa);
  else
    DMPC_LAUNCH_GGL(pendulum_rollout_linearize_kernel, dim3((pa.B + 63) / 64), dim3(64), 0, stream, pa);
}

static bool spec4_disabled() {  // DMPC_NO_SPEC4=1: the lane-per-candidate speculative search (A/B timing, longer horizons' path)
  static const bool off = [] { const char *e = getenv("DMPC_NO_SPEC4"); return e && e[0] == '1'; }();
  return off;
}

static int launch_mpc_fwd(int nx, int nu, const MpcFwdArgs &a_in, hipStream_t stream) {
  MpcFwdArgs a = a_in;
  if (a.dyn_kind == 1 && nx == 3 && nu == 1 && !spec_line_search_disabled()) {
    // the pendulum's line search usually walks ten or more step sizes: 16 candidates per trajectory at once; every
    // candidate keeps its trajectory in LDS (T * 4 KB per workgroup) when that fits
    if (a.T <= kSpec4MaxT && !spec4_disabled()) {   // a wavefront per trajectory, all inputs and candidates in LDS
      DMPC_LAUNCH_GGL(mpc_forward_rec_pendulum_spec4_kernel, dim3((a.B + 3) / 4), dim3(256),
                         Spec4Layout::lds_bytes(a.T), stream, a);
      return (int)hipGetLastError();
    }
    const size_t lds = (size_t)a.T * 4 * 256 * sizeof(float);
    a.traj_in_lds = lds <= 96 * 1024 ? 1 : 0;
    const bool dma = a.B >= 4 && a.B % 4 == 0 && !mpc_dma_disabled() &&
                     aligned16(a.C, a.c, a.Ks, a.ks, a.controls, a.lower, a.upper, a.states);
    if (dma)
      DMPC_LAUNCH_GGL(mpc_forward_rec_pendulum_spec_kernel<true>, dim3((a.B + 15) / 16), dim3(256),
                         (a.traj_in_lds ? lds : 0) + SpecDmaLayout::lds_bytes(), stream, a);
    else
      DMPC_LAUNCH_GGL(mpc_forward_rec_pendulum_spec_kernel<false>, dim3((a.B + 15) / 16), dim3(256),
                         a.traj_in_lds ? lds : 0, stream, a);
    return (int)hipGetLastError();
  }
  // LinDx, whole wavefronts of four trajectories, 16-byte aligned runs: inputs through the LDS-DMA ring
  const bool fwd_dma = a.dyn_kind == 0 && a.B >= 4 && a.B % 4 == 0 && !mpc_dma_disabled() &&
                       aligned16(a.C, a.c, a.F, a.f, a.Ks, a.ks, a.controls, a.lower, a.upper, a.states);
  // the line search as one generated instruction stream (mpc_fwd_asm_kernel.hpp); DMPC_NO_MPC_ASM=1: the HIP kernels
  if (fwd_dma && a.ls_cap > 0 && !mpc_asm_disabled()) {
    const dim3 grid((a.B + 15) / 16), block(256);
#define A(NX_, NU_)                                                                                          \
  if (nx == NX_ && nu == NU_) {                                                                              \
    DMPC_LAUNCH_GGL((mpc_forward_asm_kernel<NX_, NU_>), grid, block, (mpc_fwd_asm_lds_bytes<NX_, NU_>()), stream, a); \
    return (int)hipGetLastError();                                                                           \
  }
    A(8, 2) A(3, 1) A(4, 2) A(6, 2) A(2, 2) A(1, 1) A(2, 1) A(3, 2)
#undef A
  }
#define X(NX_, NU_, L_)                                                                                      \
  if (nx == NX_ && nu == NU_) {                                                                              \
    constexpr int GPB = 256 / L_;                                                                            \
    if constexpr (L_ == 16 && MpcFwdDmaLayout<NX_, NU_>::lds_bytes() <= 96 * 1024) {                          \
      if (fwd_dma) {                                                                                         \
        DMPC_LAUNCH_GGL((mpc_forward_rec_kernel<NX_, NU_, L_, true>), dim3((a.B + GPB - 1) / GPB), dim3(256), \
                           (MpcFwdDmaLayout<NX_, NU_>::lds_bytes()), stream, a);                             \
        return (int)hipGetLastError();                                                                       \
      }                                                                                                      \
    }                                                                                                        \
    DMPC_LAUNCH_GGL((mpc_forward_rec_kernel<NX_, NU_, L_>), dim3((a.B + GPB - 1) / GPB), dim3(256), 0, stream, \
                       a);                                                                                   \
    return (int)hipGetLastError();                                                                           \
  }
  DMPC_MPC_SHAPES(X)
#undef X
  // 17 to 31 elements of tau, at most 16 states: the line search of the wide row layout (mpc_wide_forward_kernel.hpp: four
  // trajectories per wavefront; before, the runtime-dimension kernel - 0.68 ms at (12,4)).  DMPC_NO_WIDE=1: that one.
  {
    static const bool wide_off = [] { const char *e = getenv("DMPC_NO_WIDE"); return e && e[0] == '1'; }();
    if (!wide_off && a.dyn_kind == 0 && a.ls_cap > 0 && a.B >= 4 && a.T >= 2 && a.info_in == nullptr &&
        aligned16(a.C, a.c, a.F, a.f, a.Ks, a.ks, a.controls, a.lower, a.upper, a.states) &&
        (size_t)a.B * (nx + nu) * (nx + nu) * 4 < ((size_t)1 << 31)) {
#define X(NX_, NU_)                                                                                            \
  if (nx == NX_ && nu == NU_) {                                                                                \
    using Lay = MpcWideFwdLayout<NX_, NU_, 2>;                                                                 \
    static_assert(Lay::lds_bytes() <= 160 * 1024, "ring beyond a CU's LDS");                                   \
    if (Lay::lds_bytes() > 64 * 1024)                                                                          \
      set_max_lds(reinterpret_cast<const void *>(&mpc_wide_forward_kernel<NX_, NU_, 2>), \
                                (int)Lay::lds_bytes());           \
    DMPC_LAUNCH_GGL((mpc_wide_forward_kernel<NX_, NU_, 2>), dim3((a.B + 15) / 16), dim3(256), Lay::lds_bytes(), stream, a); \
    return (int)hipGetLastError();                                                                             \
  }
      X(12, 4) X(16, 4) X(12, 8) X(16, 8)
#undef X
      if (nx >= 1 && nu >= 1 && nx <= 16 && nu <= 8 && mpc_wide_padded(nx, nu) && a.B % 4 == 0 && !mpc_container_disabled()) {
        MpcFwdArgs p = a;     // ... and padded inside the smallest of those instances (as the sweep)
        p.nx_log = nx;
        p.nu_log = nu;
#define X(NX_, NU_)                                                                                            \
  if (nx <= NX_ && nu <= NU_) {                                                                                \
    using Lay = MpcWideFwdLayout<NX_, NU_, 2>;                                                                 \
    if (Lay::lds_bytes() > 64 * 1024)                                                                          \
      set_max_lds(reinterpret_cast<const void *>(&mpc_wide_forward_kernel<NX_, NU_, 2, true>), \
                                (int)Lay::lds_bytes());           \
    DMPC_LAUNCH_GGL((mpc_wide_forward_kernel<NX_, NU_, 2, true>), dim3((p.B + 15) / 16), dim3(256), Lay::lds_bytes(), stream, p); \
    return (int)hipGetLastError();                                                                             \
  }
        X(12, 4) X(16, 4) X(12, 8) X(16, 8)
#undef X
      }
    }
  }
  if (a.dyn_kind == 0 && !mpc_container_disabled()) {
    a.nx_log = nx;
    a.nu_log = nu;
#define X(NX_, NU_)                                                                                              \
  if (nx <= NX_ && nu <= NU_) {                                                                                  \
    DMPC_LAUNCH_GGL((mpc_forward_rec_kernel<NX_, NU_, 16, false, true>), dim3((a.B + 15) / 16), dim3(256), 0, stream, a); \
    return (int)hipGetLastError();                                                                               \
  }
    DMPC_MPC_CONTAINERS(X)
#undef X
  }
  if (a.dyn_kind == 0 && nx + nu + 1 <= 64 && nu <= kMpcGenericMaxNu && !mpc_staged_forward_disabled()) {
    // the inputs of a step through an LDS ring (mpc_staged_forward_kernel); DMPC_NO_MPC_STAGED_FWD=1: the kernel below
    const size_t shmem = mpc_staged_fwd_lds_bytes(nx, nu);
    if (shmem <= 150 * 1024) {
      if (shmem > 64 * 1024)
        set_max_lds(reinterpret_cast<const void *>(&mpc_staged_forward_kernel), (int)shmem);
      DMPC_LAUNCH_GGL(mpc_staged_forward_kernel, dim3(a.B), dim3(64), shmem, stream, a, nx, nu);
      return (int)hipGetLastError();
    }
  }
  if (a.dyn_kind == 0 && nx + nu + 1 <= 64 && nu <= kMpcGenericMaxNu) {
    DMPC_LAUNCH_GGL(mpc_generic_forward_kernel, dim3(a.B), dim3(64), mpc_generic_fwd_lds_bytes(nx, nu), stream, a,
                       nx, nu);
    return (int)hipGetLastError();
  }
  if (a.dyn_kind == 0) {   // any other size: mpc_tiled.hpp
    const size_t shmem = (size_t)(2 * nx + 2 * (nx + nu) + 4) * sizeof(float);
    if (shmem > 60 * 1024) return DMPC_E_UNSUPPORTED;
    DMPC_LAUNCH_GGL(mpc_tiled_forward_kernel, dim3(a.B), dim3(kTiledThreads), shmem, stream, a, nx, nu);
    return (int)hipGetLastError();
  }
  return DMPC_E_UNSUPPORTED;
}

struct MpcWs {
  size_t c_back, neg, x0, dx, du, mask, sync, tiled, lqr, total;
};
constexpr int kSyncQpIterMax = 64;   // the workspace queries do not know n_qp_iter_max: sized for up to this many
// Both generated streams in one launch (mpc_step_fused_kernel.hpp) when each of them would have been chosen on its own:
// the conditions of launch_mpc_back's and launch_mpc_fwd's stream branches.  DMPC_NO_MPC_FUSED=1: two launches (A/B).
static int launch_mpc_step_fused(int nx, int nu, const MpcBackArgs &ba, const MpcFwdArgs &fa, hipStream_t stream) {
  static const bool off = [] { const char *e = getenv("DMPC_NO_MPC_FUSED"); return e && e[0] == '1'; }();
  if (off || mpc_asm_disabled() || !mpc_back_dma_ok(ba) || ba.T < 2 || !(ba.states == nullptr || ba.f == nullptr))
    return DMPC_E_UNSUPPORTED;
  if (!(fa.dyn_kind == 0 && fa.ls_cap > 0 &&
        aligned16(fa.C, fa.c, fa.F, fa.f, fa.Ks, fa.ks, fa.controls, fa.lower, fa.upper, fa.states)))
    return DMPC_E_UNSUPPORTED;
  const dim3 grid((ba.B + 15) / 16), block(256);
  const int variant = ba.states != nullptr ? 2 : (ba.f != nullptr ? 1 : 0);   // re-centring / with f_hat / plain
#define A(NX_, NU_)                                                                                              \
  if (nx == NX_ && nu == NU_) {                                                                                  \
    constexpr size_t lds = mpc_step_fused_lds_bytes<NX_, NU_>();                                                 \
    if (variant == 2) DMPC_LAUNCH_GGL((mpc_step_fused_asm_kernel<NX_, NU_, false, true>), grid, block, lds, stream, ba, fa); \
    else if (variant == 1) DMPC_LAUNCH_GGL((mpc_step_fused_asm_kernel<NX_, NU_, true, false>), grid, block, lds, stream, ba, fa); \
    else DMPC_LAUNCH_GGL((mpc_step_fused_asm_kernel<NX_, NU_, false, false>), grid, block, lds, stream, ba, fa); \
    return (int)hipGetLastError();                                                                               \
  }
  A(8, 2) A(3, 1) A(4, 2) A(2, 2) A(1, 1) A(2, 1) A(3, 2)
#undef A
  return DMPC_E_UNSUPPORTED;
}

static MpcWs mpc_layout(int T, int B, int nx, int nu) {
  const size_t ns = nx + nu;
  MpcWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += round_up(bytes, 256);
    return o;
  };
  w.c_back = take((size_t)T * B * ns * sizeof(float));  // forward: re-centred c ; backward: unused
  w.neg = take((size_t)T * B * ns * sizeof(float));     // backward: -[dl_dx;dl_du]
  w.x0 = take((size_t)B * nx * sizeof(float));
  w.dx = take((size_t)T * B * nx * sizeof(float));
  w.du = take((size_t)T * B * nu * sizeof(float));
  w.mask = take((size_t)T * B * nu);
  w.sync = take(coupled_bytes(T, kSyncQpIterMax));
  w.tiled = take(coupled_traj_bytes(B, nx, nu));     // the tiled sweep's matrices (+ the QP's vectors of its batch-coupled form)
  w.lqr = off;
  off += round_up(dmpc_lqr_workspace_bytes(T, B, nx, nu), 256);
  w.total = off;
  return w;
}

// workspace of the device-driven box-DDP loop (dmpc_box_ddp)
struct DdpWs {
  size_t xs, F, f, c_back, Ks, ks, x_new, u_a, u_b, u_c, u1, u1_b, costs, costs_b, old, alphas, nqp, nls, keep, info_back, stage_a,
      stage_b, sel_sync, sync, tiled, total;
};
static DdpWs ddp_layout(int T, int B, int nx, int nu) {
  const size_t ns = nx + nu, TB = (size_t)T * B, fl = sizeof(float);
  DdpWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += round_up(bytes, 256);
    return o;
  };
  w.xs = take(TB * nx * fl);
  w.F = take(TB * nx * ns * fl);   // linearisation of a built-in dynamics (unused with a LinDx)
  w.f = take(TB * nx * fl);
  w.c_back = take(TB * ns * fl);
  w.Ks = take(TB * nu * nx * fl);
  w.ks = take(TB * nu * fl);
  w.x_new = take(TB * nx * fl);
  w.u_a = take(TB * nu * fl);
  w.u_b = take(TB * nu * fl);
  w.u_c = take(TB * nu * fl);     // (third control buffer, second u1 / costs, staging flags: the one-launch iterations)
  w.u1 = take(TB * nu * fl);
  w.u1_b = take(TB * nu * fl);
  w.costs = take((size_t)B * fl);
  w.costs_b = take((size_t)B * fl);
  w.old = take((size_t)B * fl);
  w.alphas = take((size_t)B * fl);
  w.nqp = take((size_t)B * sizeof(int32_t));
  w.nls = take((size_t)B * sizeof(int32_t));
  w.keep = take((size_t)B * sizeof(int32_t));
  w.info_back = take((size_t)B * sizeof(int32_t));
  w.stage_a = take((size_t)B * sizeof(int32_t));
  w.stage_b = take((size_t)B * sizeof(int32_t));
  w.sel_sync = take(4 * sizeof(unsigned));
  w.sync = take(coupled_bytes(T, kSyncQpIterMax));
  w.tiled = take(coupled_traj_bytes(B, nx, nu));
  w.total = off;
  return w;
}

// dmpc_mpc_step_status: one workgroup reduces the step's per-trajectory words (include/dmpc.h)
__global__ __launch_bounds__(1024) void mpc_step_status_kernel(int B, const int32_t *__restrict__ info, const int32_t *__restrict__ nqp,
                                                               const float *__restrict__ alphas, int32_t *__restrict__ status) {
  __shared__ int s_or[16], s_max[16], s_bad[16];
  __shared__ double s_sum[16];
  int o = 0, m = 0, bad = 0;
  double sum = 0.0;
  for (int b = threadIdx.x; b < B; b += 1024) {
    if (info != nullptr) {
      const int v = info[b];
      o |= v;
      bad += (v & DMPC_INFO_NONFINITE) != 0 ? 1 : 0;
    }
    if (nqp != nullptr) m = max(m, nqp[b]);
    if (alphas != nullptr) sum += (double)alphas[b];
  }
  for (int d = 32; d >= 1; d >>= 1) {
    o |= __shfl_xor(o, d);
    m = max(m, __shfl_xor(m, d));
    bad += __shfl_xor(bad, d);
    sum += __shfl_xor(sum, d);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_or[w] = o; s_max[w] = m; s_bad[w] = bad; s_sum[w] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) { o |= s_or[i]; m = max(m, s_max[i]); bad += s_bad[i]; sum += s_sum[i]; }
    const unsigned long long bits = __double_as_longlong(sum);
    status[0] = o; status[1] = m; status[2] = bad; status[3] = 0;
    status[4] = (int32_t)(bits & 0xffffffffull); status[5] = (int32_t)(bits >> 32); status[6] = 0; status[7] = 0;
  }
}

static int grid_for(size_t n) {
  const size_t blocks = (n + 255) / 256;
  return (int)(blocks > 8192 ? 8192 : (blocks == 0 ? 1 : blocks));
}

}  // namespace dmpc

using namespace dmpc;

extern "C" {

size_t dmpc_coupled_workspace_bytes(int T, int n_qp_iter_max) {
  if (T <= 0 || n_qp_iter_max <= 0) return 0;
  return coupled_bytes(T, n_qp_iter_max);
}

size_t dmpc_pnqp_workspace_bytes(int B, int n, int n_iter, int batch_coupled) {
  if (B <= 0 || n <= 0 || n_iter <= 0 || !batch_coupled) return 0;
  return coupled_bytes(1, n_iter) + pnqp_coupled_row_bytes(B, n);
}

int dmpc_pnqp(int B, int n, const float *H, const float *q, const float *lower, const float *upper,
              const float *x_init, int n_iter, int batch_coupled, float *x, float *fac, int32_t *piv, float *index_f,
              int32_t *n_iter_out, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream_) {
  if (B <= 0 || n <= 0 || n_iter <= 0 || !H || !q || !lower || !upper || !x || !fac || !index_f || !n_iter_out)
    return DMPC_E_BADARG;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 block(256), grid((B + 255) / 256);
  unsigned *sync = nullptr;
  if (batch_coupled) {
    if (!ws || ws_bytes < coupled_bytes(1, n_iter)) return DMPC_E_WORKSPACE;
    sync = static_cast<unsigned *>(ws);
    const hipError_t e = hipMemsetAsync(sync, 0, coupled_bytes(1, n_iter), stream);
    if (e != hipSuccess) return (int)e;
  }
  PnqpArgs a{B, H, q, lower, upper, x_init, n_iter, x, fac, piv, index_f, n_iter_out, info, sync};
  void *args[] = {&a};
  // batch-coupled on a grid that always fits (mpc_coupled.hpp): any n, any B; needs dmpc_pnqp_workspace_bytes
  auto fixed_grid = [&]() -> int {
    if (ws_bytes < dmpc_pnqp_workspace_bytes(B, n, n_iter, 1)) return DMPC_E_WORKSPACE;
    PnqpTiledArgs ta{B, n, H, q, lower, upper, x_init, n_iter, x, fac, piv, index_f, n_iter_out, info};
    float *vecs = reinterpret_cast<float *>(static_cast<char *>(ws) + coupled_bytes(1, n_iter));
    void *targs[] = {&ta, &vecs, &sync};
    return launch_fixed_grid(reinterpret_cast<const void *>(&pnqp_coupled_kernel), B, targs, stream);
  };
  switch (n) {
#define CASE(N)                                                                                                  \
  case N:                                                                                                        \
    if (sync != nullptr) {                                                                                       \
      const int rc = coupled_register_disabled()                                                                 \
                         ? DMPC_E_UNSUPPORTED                                                                    \
                         : launch_cooperative(reinterpret_cast<const void *>(&pnqp_kernel<N>), grid, block, args, 0, stream); \
      return rc == DMPC_E_UNSUPPORTED ? fixed_grid() : rc;                                                       \
    }                                                                                                            \
    DMPC_LAUNCH_GGL((pnqp_kernel<N>), grid, block, 0, stream, a);                                             \
    break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    default: {   // any n: a workgroup per QP (mpc_tiled.hpp; batch-coupled: mpc_coupled.hpp)
      if (sync != nullptr) return fixed_grid();
      const size_t shmem = pnqp_tiled_lds_floats(n) * sizeof(float);
      if (shmem > 60 * 1024) return DMPC_E_UNSUPPORTED;
      PnqpTiledArgs ta{B, n, H, q, lower, upper, x_init, n_iter, x, fac, piv, index_f, n_iter_out, info};
      DMPC_LAUNCH_GGL(pnqp_tiled_kernel, dim3(B), dim3(kTiledThreads), shmem, stream, ta);
    }
  }
  return (int)hipGetLastError();
}

size_t dmpc_mpc_backward_rec_workspace_bytes(int T, int B, int nx, int nu, int n_qp_iter_max, int batch_coupled) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return 0;
  if (batch_coupled) return coupled_bytes(T, n_qp_iter_max) + coupled_traj_bytes(B, nx, nu);
  return mpc_needs_tiles(nx, nu) ? (size_t)B * mpc_tiled_scratch_floats(nx, nu) * sizeof(float) : 0;
}

int dmpc_mpc_backward_rec(int T, int B, int nx, int nu, const float *C_hat, const float *c_hat,
                          const float *F_hat, const float *f_hat, const float *controls, const float *u_lower,
                          const float *u_upper, int n_qp_iter_max, int batch_coupled, float *Ks_out, float *ks_out,
                          int32_t *n_qp_iter, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0 || n_qp_iter_max <= 0) return DMPC_E_BADARG;
  if (!C_hat || !c_hat || !F_hat || !controls || !u_lower || !u_upper || !Ks_out || !ks_out || !n_qp_iter)
    return DMPC_E_BADARG;
  if (!aligned16(C_hat) || !aligned16(c_hat) || !aligned16(F_hat) || !aligned16(f_hat)) return DMPC_E_BADARG;
  if (batch_coupled && (!ws || ws_bytes < dmpc_mpc_backward_rec_workspace_bytes(T, B, nx, nu, n_qp_iter_max, 1))) return DMPC_E_WORKSPACE;
  MpcBackArgs ba{T, B, C_hat, c_hat, F_hat, f_hat, controls, u_lower, u_upper, n_qp_iter_max, Ks_out, ks_out,
                 n_qp_iter, info, nullptr, batch_coupled ? static_cast<unsigned *>(ws) : nullptr, nullptr};
  if (batch_coupled) ba.tiled_scratch = reinterpret_cast<float *>(static_cast<char *>(ws) + coupled_bytes(T, n_qp_iter_max));
  if (!batch_coupled && mpc_needs_tiles(nx, nu)) {
    if (!ws || ws_bytes < dmpc_mpc_backward_rec_workspace_bytes(T, B, nx, nu, n_qp_iter_max, 0)) return DMPC_E_WORKSPACE;
    ba.tiled_scratch = static_cast<float *>(ws);
  }
  return launch_mpc_back(nx, nu, ba, static_cast<hipStream_t>(stream_));
}

int dmpc_mpc_forward_rec(int T, int B, int nx, int nu, const float *Ks, const float *ks, const float *controls,
                         const float *states, const float *u_lower, const float *u_upper, const float *C_true,
                         const float *c_true, const float *F_true, const float *f_true, float ls_decay,
                         int max_ls_iter, float *x_out, float *u_out, float *costs, float *old_costs,
                         float *alphas, float *objs, float *u_first, int32_t *n_ls_iter, int32_t *info,
                         dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!Ks || !ks || !controls || !states || !u_lower || !u_upper || !C_true || !c_true || !F_true || !x_out ||
      !u_out || !costs || !alphas || !n_ls_iter)
    return DMPC_E_BADARG;
  if (!aligned16(C_true) || !aligned16(F_true)) return DMPC_E_BADARG;
  MpcFwdArgs fa{T, B, Ks, ks, controls, states, u_lower, u_upper, C_true, c_true, F_true, f_true, ls_decay,
                max_ls_iter, /*ls_cap=*/64, x_out, u_out, u_first, costs, old_costs, alphas, objs, n_ls_iter, info};
  return launch_mpc_fwd(nx, nu, fa, static_cast<hipStream_t>(stream_));
}

int dmpc_mpc_forward_rec_pendulum(int T, int B, const float *Ks, const float *ks, const float *controls,
                                  const float *states, const float *u_lower, const float *u_upper,
                                  const float *C_true, const float *c_true, float g, float m, float l, float dt,
                                  float max_torque, float ls_decay, int max_ls_iter, float *x_out, float *u_out,
                                  float *costs, float *old_costs, float *alphas, float *objs, float *u_first,
                                  int32_t *n_ls_iter, int32_t *info, dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0) return DMPC_E_BADARG;
  if (!Ks || !ks || !controls || !states || !u_lower || !u_upper || !C_true || !c_true || !x_out || !u_out || !costs ||
      !alphas || !n_ls_iter)
    return DMPC_E_BADARG;
  if (!aligned16(C_true)) return DMPC_E_BADARG;
  MpcFwdArgs fa{T, B, Ks, ks, controls, states, u_lower, u_upper, C_true, c_true, nullptr, nullptr, ls_decay,
                max_ls_iter, /*ls_cap=*/64, x_out, u_out, u_first, costs, old_costs, alphas, objs, n_ls_iter, info,
                /*dyn_kind=*/1, g, m, l, dt, max_torque};
  return launch_mpc_fwd(3, 1, fa, static_cast<hipStream_t>(stream_));
}

int dmpc_pendulum_rollout_linearize(int T, int B, const float *x_init, const float *u, float g, float m, float l,
                                    float dt, float max_torque, int clamp_grad_closed, float *x_out, float *F_out,
                                    float *f_out, dmpc_stream_t stream_) {
  if (T <= 0 || B <= 0 || !x_init || !u || !x_out) return DMPC_E_BADARG;
  if (f_out != nullptr && F_out == nullptr) return DMPC_E_BADARG;
  PendulumArgs pa{T, B, x_init, u, g, m, l, dt, max_torque, clamp_grad_closed, x_out, F_out, f_out};
  launch_pendulum_rollout(pa, static_cast<hipStream_t>(stream_));
  return (int)hipGetLastError();
}

size_t dmpc_mpc_step_workspace_bytes(int T, int B, int nx, int nu) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return 0;
  return mpc_layout(T, B, nx, nu).total;
}

int dmpc_mpc_step_forward(int T, int B, int nx, int nu, const float *C_hat, const float *c_hat,
                          const float *F_hat, const float *f_hat, const float *controls, const float *states,
                          const float *u_lower, const float *u_upper, const float *C_true, const float *c_true,
                          const float *F_true, const float *f_true, int need_expand, float ls_decay,
                          int max_ls_iter, int n_qp_iter_max, int batch_coupled, float *x_out, float *u_out,
                          float *Ks_out, float *ks_out, float *costs, float *old_costs, float *alphas, float *objs,
                          float *u_first, int32_t *n_qp_iter, int32_t *n_ls_iter, void *ws, size_t ws_bytes,
                          int32_t *info, dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0 || n_qp_iter_max <= 0) return DMPC_E_BADARG;
  if (batch_coupled && n_qp_iter_max > kSyncQpIterMax) return DMPC_E_UNSUPPORTED;
  if (!C_hat || !c_hat || !F_hat || !controls || !states || !u_lower || !u_upper || !C_true || !c_true || !F_true ||
      !x_out || !u_out || !Ks_out || !ks_out || !costs || !alphas || !n_qp_iter || !n_ls_iter || !ws)
    return DMPC_E_BADARG;
  if (!aligned16(C_hat) || !aligned16(c_hat) || !aligned16(F_hat) || !aligned16(f_hat) || !aligned16(C_true) ||
      !aligned16(F_true))
    return DMPC_E_BADARG;
  const MpcWs w = mpc_layout(T, B, nx, nu);
  if (ws_bytes < w.total) return DMPC_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char *base = static_cast<char *>(ws);
  // Taylor re-centring (c_hat <- C [x;u] + c, f_hat <- None, mpc_step.py:305-317) happens inside the backward sweep
  const float *c_use = c_hat;
  const float *f_use = need_expand ? nullptr : f_hat;
  MpcBackArgs ba{T, B, C_hat, c_use, F_hat, f_use, controls, u_lower, u_upper, n_qp_iter_max, Ks_out, ks_out,
                 n_qp_iter, info, nullptr, batch_coupled ? reinterpret_cast<unsigned *>(base + w.sync) : nullptr,
                 need_expand ? states : nullptr};
  if (mpc_needs_tiles(nx, nu) || batch_coupled) ba.tiled_scratch = reinterpret_cast<float *>(base + w.tiled);
  MpcFwdArgs fa{T, B, Ks_out, ks_out, controls, states, u_lower, u_upper, C_true, c_true, F_true, f_true, ls_decay,
                max_ls_iter, /*ls_cap=*/64, x_out, u_out, u_first, costs, old_costs, alphas, objs, n_ls_iter, info};
  {
    const int rc = launch_mpc_step_fused(nx, nu, ba, fa, stream);
    if (rc != DMPC_E_UNSUPPORTED) return rc;
  }
  int rc = launch_mpc_back(nx, nu, ba, stream);
  if (rc != 0) return rc;
  return launch_mpc_fwd(nx, nu, fa, stream);
}

int dmpc_lin_rollout(int T, int B, int nx, int nu, const float *x_init, const float *u, const float *F,
                     const float *f, float *x_out, dmpc_stream_t stream_) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0 || !x_init || !u || !x_out || (T > 1 && !F)) return DMPC_E_BADARG;
  DMPC_LAUNCH_GGL(lin_rollout_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream_), T, B, nx,
                     nu, x_init, u, F, f, x_out, nullptr, ChainClear{});
  return (int)hipGetLastError();
}

int dmpc_mpc_step_status(int B, const int32_t *info, const int32_t *n_qp_iter, const float *alphas, int32_t *status,
                         dmpc_stream_t stream_) {
  if (B <= 0 || !status) return DMPC_E_BADARG;
  // (not noted as "the kernel this thread launched last": dmpc_last_kernel_name() keeps naming the step's own kernel)
  hipLaunchKernelGGL(mpc_step_status_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream_), B, info, n_qp_iter, alphas,
                     status);
  return (int)hipGetLastError();
}

size_t dmpc_box_ddp_workspace_bytes(int T, int B, int nx, int nu) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return 0;
  return ddp_layout(T, B, nx, nu).total;
}

int dmpc_box_ddp(int T, int B, int nx, int nu, const float *x_init, const float *C, const float *c, const float *F,
                 const float *f, int dyn_kind, const float *dyn_params, const float *u_init, const float *u_lower,
                 const float *u_upper, float eps, int not_improved_lim, float ls_decay, int max_ls_iter,
                 float best_cost_eps, int max_iter, int n_qp_iter_max, int scrambled_norm, int batch_coupled,
                 float *x_best, float *u_best, float *costs_best, float *du_norm_best, float *du_norm_last,
                 int32_t *state, void *ws, size_t ws_bytes, int32_t *info, dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0 || max_iter <= 0 || n_qp_iter_max <= 0) return DMPC_E_BADARG;
  if (batch_coupled && n_qp_iter_max > kSyncQpIterMax) return DMPC_E_UNSUPPORTED;
  if (!x_init || !C || !c || !u_init || !u_lower || !u_upper || !x_best || !u_best || !costs_best || !du_norm_best ||
      !du_norm_last || !state || !ws)
    return DMPC_E_BADARG;
  if (dyn_kind == 0 && !F) return DMPC_E_BADARG;
  if (dyn_kind == 1 && (!dyn_params || nx != 3 || nu != 1)) return DMPC_E_BADARG;
  if (dyn_kind != 0 && dyn_kind != 1) return DMPC_E_UNSUPPORTED;
  // (round 5: shapes that run on the tiled kernels - more than 8 controls / 64 columns - too: their matrices live in w.tiled)
  if ((size_t)T * B * (nx + nu) >= ((size_t)1 << 31)) return DMPC_E_UNSUPPORTED;  // 32-bit indices in the bookkeeping
  if (!aligned16(C) || !aligned16(c) || !aligned16(F) || !aligned16(f) || !aligned16(ws)) return DMPC_E_BADARG;
  const DdpWs w = ddp_layout(T, B, nx, nu);
  if (ws_bytes < w.total) return DMPC_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char *base = static_cast<char *>(ws);
  auto fp = [&](size_t off) { return reinterpret_cast<float *>(base + off); };
  auto ip = [&](size_t off) { return reinterpret_cast<int32_t *>(base + off); };
  float *xs = fp(w.xs), *c_back = fp(w.c_back), *Ks = fp(w.Ks), *ks = fp(w.ks), *x_new = fp(w.x_new);
  float *u_buf[2] = {fp(w.u_a), fp(w.u_b)};
  float *u1 = fp(w.u1), *costs = fp(w.costs), *alphas = fp(w.alphas);
  const float *F_hat = dyn_kind == 0 ? F : fp(w.F);
  const float pg = dyn_kind == 1 ? dyn_params[0] : 0.f, pm = dyn_kind == 1 ? dyn_params[1] : 0.f,
              pl = dyn_kind == 1 ? dyn_params[2] : 0.f, pdt = dyn_kind == 1 ? dyn_params[3] : 0.f,
              pmax = dyn_kind == 1 ? dyn_params[4] : 0.f;
  const int pclosed = dyn_kind == 1 ? (dyn_params[5] != 0.f) : 0;
  const size_t rows = (size_t)T * B;
  // loop state, the bookkeeping workgroups' meeting words and the flags are cleared by the chain's first launch
  ChainClear clear;
  clear.p[0] = state; clear.n[0] = 8;
  clear.p[1] = ip(w.sel_sync); clear.n[1] = 4;
  clear.p[2] = info; clear.n[2] = info != nullptr ? B : 0;
  const int32_t *done = state + kDdpDone;
  // Pendulum: the trajectory the line search accepts IS the next iteration's nominal one, and the search writes its
  // linearisation and re-centred cost while it writes the trajectory - the rollout + linearisation kernel runs for
  // the first iteration only, the nominal states ping-pong between the two state buffers.
  const bool fuse_lin = dyn_kind == 1 && !spec_line_search_disabled();
  float *x_buf[2] = {xs, x_new};
  const bool copy_here = B <= kDdpCopyHereMaxB && rows * (size_t)(nx + nu) <= kDdpCopyHereMaxElems;
  bool fused_select = false;   // decided at the first sweep: the bookkeeping of iteration i rides in sweep i + 1's launch
  DdpSelectArgs sa{};
  // One launch per iteration (box_ddp_pendulum_iter_kernel) when the fused chain runs and the search is the wavefront-per-
  // trajectory one: the search then runs BESIDE the previous iteration's bookkeeping, so what that reads (the controls the
  // step started from, the first pass's controls, the costs) must not be what this search writes - three control buffers in
  // rotation, two of u_first / costs, and the iteration's flags through a staging word that the bookkeeping merges.
  // The first of these launches also rolls out and linearises the nominal trajectory (the chain's first launch otherwise).
  // DMPC_NO_DDP_ITER_FUSED=1 (read at every call): rollout, sweep and search as launches of their own (A/B timing; bit-identical).
  bool iter_fused = false;
  if (fuse_lin && nx == 3 && nu == 1 && copy_here) {   // (decided before the first launch: the first iteration's launch also rolls out)
    const MpcBackArgs ba0{T, B, C, c_back, F_hat, nullptr, u_init, u_lower, u_upper, n_qp_iter_max, Ks, ks, ip(w.nqp), info, done,
                          batch_coupled ? reinterpret_cast<unsigned *>(base + w.sync) : nullptr, nullptr, 0};
    const char *e = getenv("DMPC_NO_DDP_ITER_FUSED");
    iter_fused = mpc_back_dma_ok(ba0) && T >= 2 && T <= kSpec4MaxT && !spec4_disabled() && !mpc_asm_disabled() &&
                 aligned16(fp(w.F)) && !(e && e[0] == '1');
  }
  float *u_buf3[3] = {fp(w.u_a), fp(w.u_b), fp(w.u_c)};
  float *u1_2[2] = {u1, fp(w.u1_b)}, *costs_2[2] = {costs, fp(w.costs_b)};
  int32_t *stage_2[2] = {ip(w.stage_a), ip(w.stage_b)};
  for (int it = 0; it < max_iter; ++it) {
    // (iteration 0 names the same buffers in both rotations: the choice is made inside it)
    const float *u_cur = it == 0 ? u_init : (iter_fused ? u_buf3[it % 3] : u_buf[it & 1]);   // the first iteration reads the caller's controls in place
    float *u_new = iter_fused ? u_buf3[(it + 1) % 3] : u_buf[(it & 1) ^ 1];
    float *u1_it = iter_fused ? u1_2[it & 1] : u1, *costs_it = iter_fused ? costs_2[it & 1] : costs;
    float *xs_it = fuse_lin ? x_buf[it & 1] : xs;
    float *xn_it = fuse_lin ? x_buf[(it & 1) ^ 1] : x_new;
    // nominal trajectory and the Taylor models around it                                    box_ddp.py:123-171
    PendulumArgs pa_first{};     // one-launch iterations: the first iteration's launch rolls out and linearises itself (pa_first.T > 0)
    if (dyn_kind == 1) {
      if (it == 0 || !fuse_lin) {
        PendulumArgs pa{T, B, x_init, u_cur, pg, pm, pl, pdt, pmax, pclosed, xs_it, fp(w.F), fp(w.f), it == 0 ? nullptr : done, C, c,
                        c_back, it == 0 ? clear : ChainClear{}};
        if (iter_fused) pa_first = pa;
        else launch_pendulum_rollout(pa, stream);
      }
    } else {
      DMPC_LAUNCH_GGL(lin_rollout_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, T, B, nx, nu, x_init, u_cur, F,
                         f, xs_it, it == 0 ? nullptr : done, it == 0 ? clear : ChainClear{});
    }
    // MPCstep.forward with need_expand: f_hat = None                                        mpc_step.py:305-328
    // pendulum: c_back was re-centred by the rollout kernel / the previous line search; LinDx: the backward sweep
    // re-centres c itself
    MpcBackArgs ba{T, B, C, dyn_kind == 1 ? c_back : c, F_hat, nullptr, u_cur, u_lower, u_upper, n_qp_iter_max, Ks, ks,
                   ip(w.nqp), info, done, batch_coupled ? reinterpret_cast<unsigned *>(base + w.sync) : nullptr,
                   dyn_kind == 1 ? nullptr : xs_it, 0};
    if (mpc_needs_tiles(nx, nu) || batch_coupled) ba.tiled_scratch = fp(w.tiled);
    if (it == 0) fused_select = fuse_lin && nx == 3 && nu == 1 && copy_here && mpc_back_dma_ok(ba);
    if (fused_select && info != nullptr) {   // the sweep may run ahead of `done`: its flags count only if the search follows
      ba.info = ip(w.info_back);
      ba.info_store = 1;
    }
    MpcFwdArgs fa{T, B, Ks, ks, u_cur, xs_it, u_lower, u_upper, C, c, dyn_kind == 0 ? F : nullptr,
                  dyn_kind == 0 ? f : nullptr, ls_decay, max_ls_iter, /*ls_cap=*/64, xn_it, u_new, u1_it, costs_it,
                  /*old_costs: nobody reads them here*/ nullptr,
                  alphas, nullptr, ip(w.nls), info, dyn_kind, pg, pm, pl, pdt, pmax, pclosed, done,
                  fuse_lin ? fp(w.F) : nullptr, fuse_lin ? fp(w.f) : nullptr, fuse_lin ? c_back : nullptr, 0,
                  fused_select && info != nullptr ? ip(w.info_back) : nullptr};
    int rc;
    if (iter_fused) {
      if (info != nullptr) {     // the search may run ahead of `done` too: sweep + search flags to this iteration's staging word
        fa.info = stage_2[it & 1];
        fa.info_store = 1;
      }
      if (it == 0) ba.done = fa.done = nullptr;   // (this launch clears the flag; nothing can have stopped the loop yet)
      const int n_sel = it > 0 ? select_parts(B) : 0;
      const size_t lds = std::max(mpc_asm_lds_bytes<3, 1>(), Spec4Layout::lds_bytes(T));
      DMPC_LAUNCH_GGL(box_ddp_pendulum_iter_kernel, dim3(B / 4 + n_sel), dim3(256), lds, stream, ba, fa, sa, n_sel,
                      reinterpret_cast<unsigned *>(base + w.sel_sync), pa_first);
      rc = (int)hipGetLastError();
      if (rc != 0) return rc;
    } else {
      rc = launch_mpc_back(nx, nu, ba, stream, fused_select && it > 0 ? &sa : nullptr,
                           reinterpret_cast<unsigned *>(base + w.sel_sync));
      if (rc != 0) return rc;
      rc = launch_mpc_fwd(nx, nu, fa, stream);
      if (rc != 0) return rc;
    }
    // best-so-far update and stop tests of this iteration                                   box_ddp.py:200-230
    sa = DdpSelectArgs{it, T, B, nx, nu, max_iter, not_improved_lim, scrambled_norm, eps, best_cost_eps, u_cur, u1_it, costs_it,
                       costs_best, du_norm_best, du_norm_last, ip(w.keep), state, copy_here ? 1 : 0, xn_it, u_new,
                       x_best, u_best};
    if (iter_fused && info != nullptr) {
      sa.info_stage = stage_2[it & 1];
      sa.info = info;
    }
    if (fused_select) continue;   // rides in the next sweep's launch (the last one: in the summary launch)
    if (nx == 3 && nu == 1)
      DMPC_LAUNCH_GGL((box_ddp_select_kernel<3, 1>), dim3(1), dim3(kDdpSelectThreads), 0, stream, sa);
    else
      DMPC_LAUNCH_GGL((box_ddp_select_kernel<0, 0>), dim3(1), dim3(kDdpSelectThreads), 0, stream, sa);
    if (!copy_here)
      DMPC_LAUNCH_GGL(box_ddp_keep_kernel, dim3(grid_for(rows * nx)), dim3(256), 0, stream, T, B, nx, nu,
                         ip(w.keep), xn_it, u_new, x_best, u_best);
  }
  // fused chain: the last iteration's bookkeeping - with the summary when one workgroup does it, else as the fused
  // launch splits it
  const bool split_last = fused_select && select_parts(B) > 1;
  if (split_last)
    DMPC_LAUNCH_GGL((box_ddp_select_parts_kernel<3, 1>), dim3(select_parts(B)), dim3(256), 0, stream, sa,
                       reinterpret_cast<unsigned *>(base + w.sel_sync));
  DMPC_LAUNCH_GGL(box_ddp_summary_kernel, dim3(1), dim3(1024), 0, stream, rows * nu, B, u_init, u_lower, u_upper, info,
                     (int)DMPC_INFO_NONFINITE, du_norm_best, eps, state, sa, fused_select && !split_last ? 1 : 0);
  return (int)hipGetLastError();
}

int dmpc_mpc_step_backward(int T, int B, int nx, int nu, const float *C_hat, const float *c_hat,
                           const float *F_hat, const float *x, const float *u, const float *u_lower,
                           const float *u_upper, const float *grad_x, const float *grad_u, float *d_x_init,
                           float *dC, float *dc, float *dF, float *df, float *dC_sum, float *dc_sum,
                           const float *detach_norm, const int32_t *detach_flag, float detach_eps, void *ws,
                           size_t ws_bytes, int32_t *info, dmpc_stream_t stream_) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C_hat || !c_hat || !F_hat || !x || !u || !u_lower || !u_upper || !d_x_init || !ws) return DMPC_E_BADARG;
  if ((dC_sum == nullptr) != (dc_sum == nullptr) || (dc == nullptr && dc_sum == nullptr)) return DMPC_E_BADARG;
  // the sums are formed by the LDS-DMA co-state kernel only (whole wavefronts of four trajectories, 16-lane shapes)
  if (dC_sum != nullptr && !costate_sums_available(T, B, nx, nu)) return DMPC_E_UNSUPPORTED;   // (nothing launched)
  if (!aligned16(C_hat) || !aligned16(c_hat) || !aligned16(F_hat) || !aligned16(dC) || !aligned16(dF))
    return DMPC_E_BADARG;
  const MpcWs w = mpc_layout(T, B, nx, nu);
  if (ws_bytes < w.total) return DMPC_E_WORKSPACE;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char *base = static_cast<char *>(ws);
  float *neg = reinterpret_cast<float *>(base + w.neg);
  float *x0 = reinterpret_cast<float *>(base + w.x0);
  float *dx = reinterpret_cast<float *>(base + w.dx);
  float *du = reinterpret_cast<float *>(base + w.du);
  uint8_t *mask = reinterpret_cast<uint8_t *>(base + w.mask);
  const size_t rows = (size_t)T * B;
  DMPC_LAUNCH_GGL(active_mask_kernel, dim3(grid_for(rows * (nx + nu))), dim3(256), 0, stream, rows, nx, nu, u,
                     u_lower, u_upper, grad_x, grad_u, mask, neg, x0, (size_t)B * nx, dC_sum, dc_sum, B, detach_norm,
                     detach_flag, detach_eps);
  // LQR_active(0, C, -d_tau, F, None, u_zero_Index=active)                               mpc_step.py:374-376
  int rc = dmpc_lqr_solve(T, B, nx, nu, C_hat, neg, F_hat, nullptr, x0, mask, nullptr, nullptr, dx, du,
                          base + w.lqr, w.total - w.lqr, info, stream_);
  if (rc != 0) return rc;
  // d_lambda_t = C_xx dx + C_xu du - d_tau[:nx] + ...  ->  r = neg, r_sign = +1; outputs negated  :383-446
  CostateArgs a{T, B, C_hat, c_hat, F_hat, x, u, dx, du, neg, 1.0f, -1.0f, /*dC_mode=*/1, /*df_shift=*/1,
                d_x_init, dC, dc, dF, df};
  a.dC_sum = dC_sum;
  a.dc_sum = dc_sum;
  return launch_costate(nx, nu, a, stream);
}

}  // extern "C"
