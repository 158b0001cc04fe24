// lqr_api.hip - C-ABI entry points of the LQR solve family (include/dmpc.h section A).
// Replaces LqrRecursion.backward/forward/solve_recursion (lqr/lqr_recursion.py:69-209) and
// LQR_active (mpc/active_constrained_lqr.py:67-202) of the reference.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/dmpc.h"
#include "api_util.hpp"
#include "lqr_asm_kernel.hpp"
#include "lqr_dma_kernel.hpp"
#include "lqr_generic.hpp"
#include "lqr_tiled.hpp"
#include "lqr_kernels.hpp"
#include "lqr_wave_api.hpp"
#include "lqr_wide_kernel.hpp"

namespace dmpc {

// LDS a 256-thread workgroup may spend on gains before we spill them to HBM.
constexpr size_t kGainLdsBudget = 156 * 1024;   // (round 4: was 64 KB - (8,4) and (12,3) at T = 50 sent their gains through HBM)
// LDS-DMA path: ring depths (timesteps in flight per wave) and the LDS it may use (one workgroup per CU).
#ifndef DMPC_DMA_DEPTH_B
#define DMPC_DMA_DEPTH_B 4
#define DMPC_DMA_DEPTH_F 8
#endif
constexpr int kDmaDepthB = DMPC_DMA_DEPTH_B, kDmaDepthF = DMPC_DMA_DEPTH_F;
constexpr size_t kDmaLdsBudget = 156 * 1024;
constexpr size_t kAsmLdsBudget = 160 * 1024;
static bool asm_path_disabled() {  // DMPC_NO_ASM=1 forces the HIP kernels (A/B timing, debugging)
  static const bool off = [] { const char *e = getenv("DMPC_NO_ASM"); return e && e[0] == '1'; }();
  return off;
}
static bool stash_disabled() {  // DMPC_NO_STASH=1: forward sweep re-reads F by LDS-DMA (A/B timing)
  static const bool off = [] { const char *e = getenv("DMPC_NO_STASH"); return e && e[0] == '1'; }();
  return off;
}
static bool wave_mfma_disabled() {  // DMPC_NO_WAVE_MFMA=1: large shapes on the readlane HIP kernel (A/B timing)
  static const bool off = [] { const char *e = getenv("DMPC_NO_WAVE_MFMA"); return e && e[0] == '1'; }();
  return off;
}
static bool container_disabled() {  // DMPC_NO_CONTAINER=1: shapes without a specialisation on the runtime-dimension kernel
  static const bool off = [] { const char *e = getenv("DMPC_NO_CONTAINER"); return e && e[0] == '1'; }();
  return off;
}
static bool dma_path_disabled() {  // DMPC_NO_DMA=1 forces the register-prefetch kernel (A/B timing, debugging)
  static const bool off = [] { const char *e = getenv("DMPC_NO_DMA"); return e && e[0] == '1'; }();
  return off;
}

// 4 / 3: generated stream with / without the F stash; 6: generated stream, gain rows through the workspace (any horizon);
// 2: LDS-DMA HIP kernel; 1: register-prefetch HIP kernel; 7: a container (the register-prefetch kernel of a larger shape);
// 0: runtime-dimension kernel
template <int NX, int NU, int L>
static int solve_path(int T, int B) {
  if constexpr (L == 16 && LqrAsm<NX, NU, false, false>::kAvailable) {
    if (B >= 4 && T >= 2 && !asm_path_disabled()) {
      if constexpr (LqrAsm<NX, NU, false, true>::kAvailable) {
        if (T <= LqrAsm<NX, NU, false, true>::NSTASH && lqr_asm_lds_bytes<NX, NU, true>(T) <= kAsmLdsBudget &&
            !stash_disabled())
          return 4;
      }
      if (lqr_asm_lds_bytes<NX, NU, false>(T) <= kAsmLdsBudget) return 3;
      if constexpr (LqrAsm<NX, NU, false, false, false, true>::kAvailable) {
        if (lqr_asm_lds_bytes<NX, NU, false, false, true>(T) <= kAsmLdsBudget) return 6;
      }
    }
  }
  if constexpr (L == 16) {
    using Lay = LqrDmaLayout<NX, NU, kDmaDepthB, kDmaDepthF>;
    if (B >= 4 && T >= 2 && Lay::lds_bytes(T) <= kDmaLdsBudget && !dma_path_disabled()) return 2;
  }
  if constexpr (L == 64 && NX % 4 == 0 && NU % 4 == 0) {
    if (!wave_mfma_disabled()) return 5;
  }
  return 1;
}

// the saving solve: the stash form of the generated stream with room in LDS for the staging area of the saved blocks
template <int NX, int NU, int L>
static int saving_available(int T, int B) {
  if constexpr (L == 16 && LqrAsm<NX, NU, true, true, false, false, true>::kAvailable)
    return solve_path<NX, NU, L>(T, B) == 4 && lqr_asm_lds_bytes<NX, NU, true, true>(T) <= kAsmLdsBudget;
  return 0;
}

template <int NX, int NU, int L>
static int launch_lqr(int mode, const LqrArgs &a, hipStream_t stream) {
  constexpr int GPB = 256 / L;
  const dim3 grid((a.B + GPB - 1) / GPB), block(256);
  const bool masked = a.mask != nullptr;
  const size_t lds_gain = (size_t)GPB * a.T * NU * (NX + 1) * sizeof(float);
#define DMPC_LAUNCH(MASKED, MODE, KLDS, SHMEM) \
  DMPC_LAUNCH_GGL((lqr_kernel<NX, NU, L, MASKED, MODE, KLDS>), grid, block, SHMEM, stream, a)
  if (a.Vv_in == nullptr && (a.c_u != nullptr || (a.x_init == nullptr && a.x != nullptr))) {
    // c in two arrays / x_init = 0: forms only the generated streams take (lqr_second_solve below)
    bool ok = false;
    if constexpr (L == 16 && LqrAsm<NX, NU, false, false>::kAvailable)
      ok = mode == kSolve && !masked && a.Ks == nullptr && (solve_path<NX, NU, L>(a.T, a.B) == 3 || solve_path<NX, NU, L>(a.T, a.B) == 4);
    if (!ok) return DMPC_E_UNSUPPORTED;
  }
  if (a.Vv_in != nullptr) {
    // DiffLqr.backward in one launch (lqr_adjoint below): the affine re-solve whose rollout writes the gradients
    if constexpr (L == 16 && LqrAsm<NX, NU, false, true, false, false, false, true, true>::kAvailable) {
      if (mode == kSolve && !masked && a.f == nullptr && a.Ks == nullptr && a.Ks_in != nullptr &&
          solve_path<NX, NU, L>(a.T, a.B) == 4) {
        const int waves = (a.B + 3) / 4;
        const size_t shmem = lqr_asm_lds_bytes<NX, NU, true>(a.T);
        DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, false, true, false, false, false, true, true>), dim3((waves + 3) / 4),
                        block, shmem, stream, a);
        return (int)hipGetLastError();
      }
    }
    return DMPC_E_UNSUPPORTED;
  }
  if (a.Ks_in != nullptr) {
    // the re-solve from saved gains (dmpc_lqr_saved_solve): the affine form of the generated stream, F in the stash
    if constexpr (L == 16 && LqrAsm<NX, NU, false, true, false, false, false, true>::kAvailable) {
      if (mode == kSolve && !masked && a.f == nullptr && a.Ks == nullptr && solve_path<NX, NU, L>(a.T, a.B) == 4) {
        const int waves = (a.B + 3) / 4;
        const size_t shmem = lqr_asm_lds_bytes<NX, NU, true>(a.T);
        DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, false, true, false, false, false, true>), dim3((waves + 3) / 4),
                           block, shmem, stream, a);
        return (int)hipGetLastError();
      }
    }
    return DMPC_E_UNSUPPORTED;
  }
  if constexpr (L == 16 && LqrAsm<NX, NU, false, false>::kAvailable) {
    // fastest path: the whole solve as one generated instruction stream (lqr_asm_kernel.hpp); with the F stash
    // (no second read of F) when the horizon fits the stash registers
    if constexpr (LqrAsm<NX, NU, false, false, true>::kAvailable) {
      // LQR_active on the generated stream: same paths, flags fetched as dwords (B * nu must be a multiple of 4)
      const int mpath = (mode == kSolve && masked && a.Ks == nullptr && (a.B * NU) % 4 == 0 &&
                         (reinterpret_cast<uintptr_t>(a.mask) & 3u) == 0)
                            ? solve_path<NX, NU, L>(a.T, a.B) : 0;
      if (mpath == 3 || mpath == 4) {
        const int waves = (a.B + 3) / 4;
        const dim3 g((waves + 3) / 4);
        const bool has_f = a.f != nullptr;
#define DMPC_ASM_LAUNCH_M(STASH)                                                                                        \
  do {                                                                                                                  \
    const size_t shmem = lqr_asm_lds_bytes<NX, NU, STASH>(a.T);                                                         \
    if (has_f) DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, true, false, STASH, true>), g, block, shmem, stream, a);      \
    else DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, false, STASH, true>), g, block, shmem, stream, a);           \
    return (int)hipGetLastError();                                                                                      \
  } while (0)
        if constexpr (LqrAsm<NX, NU, false, true, true>::kAvailable) {
          if (mpath == 4) DMPC_ASM_LAUNCH_M(true);
        }
        DMPC_ASM_LAUNCH_M(false);
#undef DMPC_ASM_LAUNCH_M
      }
    }
    if (mode == kBackwardOnly && !masked && solve_path<NX, NU, L>(a.T, a.B) >= 3 && solve_path<NX, NU, L>(a.T, a.B) != 6 &&
        lqr_asm_lds_bytes<NX, NU, false>(a.T) <= kAsmLdsBudget) {
      // LqrRecursion.backward(): the generated stream's backward sweep with the gains written to HBM (x == nullptr)
      const int waves = (a.B + 3) / 4;
      const size_t shmem = lqr_asm_lds_bytes<NX, NU, false>(a.T);
      if (a.f != nullptr)
        DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, true, true, false>), dim3((waves + 3) / 4), block, shmem, stream, a);
      else
        DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, true, false>), dim3((waves + 3) / 4), block, shmem, stream, a);
      return (int)hipGetLastError();
    }
    const int path = (mode == kSolve && !masked) ? solve_path<NX, NU, L>(a.T, a.B) : 0;
    if constexpr (LqrAsm<NX, NU, false, false, false, true>::kAvailable) {
      // long horizons: the gain rows pass through the workspace instead of LDS (the caller's Ks / ks are another layout:
      // a solve that wants them goes to the kernels below)
      if (path == 6 && a.Ks == nullptr && a.wsK != nullptr && a.c_u == nullptr && a.x_init != nullptr) {
        const int waves = (a.B + 3) / 4;
        const size_t shmem = lqr_asm_lds_bytes<NX, NU, false, false, true>(a.T);
        if (a.f != nullptr) DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, true, false, false, false, true>), dim3((waves + 3) / 4), block, shmem, stream, a);
        else DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, false, false, false, true>), dim3((waves + 3) / 4), block, shmem, stream, a);
        return (int)hipGetLastError();
      }
    }
    if (path >= 3 && path != 6) {
      const int waves = (a.B + 3) / 4;
      const dim3 g((waves + 3) / 4);
      const bool has_f = a.f != nullptr, write_k = a.Ks != nullptr;
#define DMPC_ASM_LAUNCH(STASH)                                                                                   \
  do {                                                                                                           \
    const size_t shmem = lqr_asm_lds_bytes<NX, NU, STASH>(a.T);                                                  \
    if constexpr (LqrAsm<NX, NU, true, STASH, false, false, true>::kAvailable) {                                 \
      if (a.Quu_out != nullptr && write_k) { /* training form: Quu, Qxu, [V | v] saved too (staged in LDS) */     \
        const size_t shmem_s = lqr_asm_lds_bytes<NX, NU, STASH, true>(a.T);                                      \
        if (shmem_s > kAsmLdsBudget) return DMPC_E_UNSUPPORTED;                                                  \
        if (has_f) DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, true, true, STASH, false, false, true>), g, block, shmem_s, stream, a); \
        else DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, true, STASH, false, false, true>), g, block, shmem_s, stream, a); \
        return (int)hipGetLastError();                                                                           \
      }                                                                                                          \
    }                                                                                                            \
    if (a.Quu_out != nullptr) return DMPC_E_UNSUPPORTED;                                                         \
    if (has_f && !write_k) DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, true, false, STASH>), g, block, shmem, stream, a); \
    else if (has_f) DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, true, true, STASH>), g, block, shmem, stream, a); \
    else if (!write_k) DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, false, STASH>), g, block, shmem, stream, a); \
    else DMPC_LAUNCH_GGL((lqr_asm_kernel<NX, NU, false, true, STASH>), g, block, shmem, stream, a);           \
    return (int)hipGetLastError();                                                                               \
  } while (0)
      if constexpr (LqrAsm<NX, NU, false, true>::kAvailable) {
        if (path == 4) DMPC_ASM_LAUNCH(true);
      }
      DMPC_ASM_LAUNCH(false);
#undef DMPC_ASM_LAUNCH
    }
  }
  if constexpr (L == 64 && NX % 4 == 0 && NU % 4 == 0) {
    // large shapes: backward sweep on the matrix cores (lqr_wave_mfma.hpp, its own translation unit; plain and
    // LQR_active), gains through HBM, then the bandwidth-bound forward-only kernel
    if (mode != kForwardOnly && !wave_mfma_disabled()) {
      LqrArgs s = a;
      if (s.Ks == nullptr && s.wsK == nullptr) return DMPC_E_WORKSPACE;
      // solve_recursion: the wavefront that finished a trajectory's backward sweep rolls it out in the same launch
      return launch_lqr_wave_mfma_backward(NX, NU, masked, mode == kSolve, s, stream);
    }
  }
  if constexpr (L == 16) {
    // fast path: LDS-DMA staged inputs (lqr_dma_kernel.hpp) - plain solve, at least one full wave of
    // trajectories, gains + rings within the 160 KB of a CU
    using Lay = LqrDmaLayout<NX, NU, kDmaDepthB, kDmaDepthF>;
    if (mode == kSolve && !masked && a.B >= 4 && a.T >= 2 && Lay::lds_bytes(a.T) <= kDmaLdsBudget &&
        !dma_path_disabled()) {
      const int waves = (a.B + 3) / 4;
      DMPC_LAUNCH_GGL((lqr_dma_kernel<NX, NU, kDmaDepthB, kDmaDepthF>), dim3((waves + 3) / 4), block,
                         Lay::lds_bytes(a.T), stream, a);
      return (int)hipGetLastError();
    }
  }
  if (mode == kSolve) {
    const bool in_lds = lds_gain <= kGainLdsBudget;
    if (in_lds) {
      if (masked) DMPC_LAUNCH(true, kSolve, true, lds_gain);
      else DMPC_LAUNCH(false, kSolve, true, lds_gain);
    } else {
      if constexpr (L == 16) {
        // long horizon: the LDS-DMA kernel with its gain rows through the workspace (its rings alone fit a CU's LDS at any
        // horizon).  Measured against lqr_kernel (profiles/r04/khbm_vs_lqr_kernel.txt): behind it while the gains still fit
        // LDS ((8,4) T = 50: 96 against 84 us - hence only here), level at T = 100, ahead at T = 200 ((4,4): 221 against 267 us)
        using LayK = LqrDmaLayout<NX, NU, kDmaDepthB, kDmaDepthF, true>;
        if (!masked && a.B >= 4 && a.T >= 2 && a.wsK != nullptr && a.Ks == nullptr && LayK::lds_bytes(a.T) <= kDmaLdsBudget &&
            !dma_path_disabled()) {
          const int waves = (a.B + 3) / 4;
          DMPC_LAUNCH_GGL((lqr_dma_kernel<NX, NU, kDmaDepthB, kDmaDepthF, true>), dim3((waves + 3) / 4), block,
                             LayK::lds_bytes(a.T), stream, a);
          return (int)hipGetLastError();
        }
      }
      LqrArgs s = a;  // long horizon: gains go through HBM (caller's Ks/ks, else the workspace)
      if (s.Ks == nullptr) { s.Ks = s.wsK; s.ks = s.wsk; }
      if (s.Ks == nullptr) return DMPC_E_WORKSPACE;
      if (masked) DMPC_LAUNCH_GGL((lqr_kernel<NX, NU, L, true, kSolve, false>), grid, block, 0, stream, s);
      else DMPC_LAUNCH_GGL((lqr_kernel<NX, NU, L, false, kSolve, false>), grid, block, 0, stream, s);
    }
  } else if (mode == kBackwardOnly) {
    if (masked) DMPC_LAUNCH(true, kBackwardOnly, false, 0);
    else DMPC_LAUNCH(false, kBackwardOnly, false, 0);
  } else {
    if (masked) DMPC_LAUNCH(true, kForwardOnly, false, 0);
    else DMPC_LAUNCH(false, kForwardOnly, false, 0);
  }
#undef DMPC_LAUNCH
  return (int)hipGetLastError();
}

// The shapes with a register-resident specialisation.  Anything else (ns + 1 <= 64) goes to the
// runtime-dimension LDS kernel in lqr_generic.hpp.
#if defined(DMPC_EXPERIMENT_ONLY_32_8)  // quick single-shape builds for kernel experiments (scripts/*variants.sh)
#define DMPC_LQR_SHAPES(X) X(32, 8, 64)
#elif defined(DMPC_EXPERIMENT_ONLY_8_2)
#define DMPC_LQR_SHAPES(X) X(8, 2, 16)
#elif defined(DMPC_EXPERIMENT_ONLY_4_4)
#define DMPC_LQR_SHAPES(X) X(4, 4, 16)
#elif defined(DMPC_EXPERIMENT_ONLY_8_4)
#define DMPC_LQR_SHAPES(X) X(8, 4, 16)
#else
#define DMPC_LQR_SHAPES(X) \
  X(1, 1, 16) X(2, 1, 16) X(3, 1, 16) X(2, 2, 16) X(3, 2, 16) X(4, 2, 16) X(6, 2, 16) X(8, 2, 16) \
  X(4, 4, 16) X(8, 4, 16) X(12, 3, 16) X(32, 8, 64)
#endif

// Containers: register-resident kernels that also take any SMALLER problem (lqr_kernel<..., PAD>; the loads pad it in
// place), in the order they are tried - fewest columns first, fewest controls among equals.  Every shape with
// nx + nu <= 15 and nu <= 8 has one.
#if defined(DMPC_EXPERIMENT_ONLY_32_8) || defined(DMPC_EXPERIMENT_ONLY_8_2) || defined(DMPC_EXPERIMENT_ONLY_4_4) || \
    defined(DMPC_EXPERIMENT_ONLY_8_4)
#define DMPC_LQR_CONTAINERS(X)
#else
// (round 4: five to eight controls too - the gain solve on the rows needs no NU x NU LU per lane; before, (5,5) ran a
// wavefront per trajectory inside the (16,8) matrix-core kernel at 0.03 of the roof)
#define DMPC_LQR_CONTAINERS(X) X(3, 1) X(4, 4) X(8, 2) X(5, 5) X(8, 4) X(14, 1) X(13, 2) X(12, 3) X(11, 4) X(10, 5) X(9, 6) X(8, 7) X(7, 8)
#endif

// ... and the wavefront-per-trajectory kernels take what is larger, up to 32 states and 8 controls
#if defined(DMPC_EXPERIMENT_ONLY_32_8) || defined(DMPC_EXPERIMENT_ONLY_8_2) || defined(DMPC_EXPERIMENT_ONLY_4_4) || \
    defined(DMPC_EXPERIMENT_ONLY_8_4)
#define DMPC_LQR_WAVE_CONTAINERS(X)
#else
#define DMPC_LQR_WAVE_CONTAINERS(X) X(16, 8) X(32, 8)
#endif

// 17 to 32 augmented columns with at most 16 states: the wide 16-lane row kernel (lqr_wide_kernel.hpp; two registers per
// matrix row, four trajectories per wavefront) for the plain fused solve of exactly these shapes - round 4; before, a
// wavefront per trajectory inside the (16,8) matrix-core instance.  DMPC_NO_WIDE=1: that path (A/B timing).
#if (defined(DMPC_EXPERIMENT_ONLY_32_8) || defined(DMPC_EXPERIMENT_ONLY_8_2) || defined(DMPC_EXPERIMENT_ONLY_4_4) || \
     defined(DMPC_EXPERIMENT_ONLY_8_4)) && !defined(DMPC_EXPERIMENT_WITH_WIDE)
#define DMPC_LQR_WIDE_SHAPES(X)
#else
#define DMPC_LQR_WIDE_SHAPES(X) X(16, 4) X(16, 8) X(12, 4) X(12, 8)
/* ((8,4) as a one-register instance of the same kernel - the template takes it - measured 94 us against lqr_kernel's 84: not used) */
#endif
#define DMPC_LQR_WIDE_CONTAINERS(X) X(12, 4) X(16, 4) X(12, 8) X(16, 8)   /* fewest columns first */
static bool wide_disabled() {
  static const bool off = [] { const char *e = getenv("DMPC_NO_WIDE"); return e && e[0] == '1'; }();
  return off;
}
static bool wide_shape(int nx, int nu) {
#define X(NX_, NU_) \
  if (nx == NX_ && nu == NU_) return true;
  DMPC_LQR_WIDE_SHAPES(X)
#undef X
  return false;
}
// ... and, padded by its reads (lqr_wide_kernel<..., PAD>), of every other shape with at most 16 states and 8 controls that has
// no 16-lane container (nx + nu >= 16): smallest instance first.  Before, all of them took a wavefront per trajectory.
static bool wide_container_shape(int nx, int nu) {
  return nx >= 1 && nu >= 1 && nx <= 16 && nu <= 8 && nx + nu >= 16 && !wide_shape(nx, nu);
}
// can the wide kernel take this solve?  (whole wavefronts of four trajectories, a horizon with an F, 32-bit time strides;
// the gain rows travel through the caller's workspace, rows of the INSTANCE's width - dmpc_lqr_workspace_bytes allows for it)
// the geometric half of wide_ok - shape, batch, horizon and the 32-bit stride bound - shared with dmpc_lqr_solve_path, so that the
// path a caller is told (and sizes its workspace for) is the path dispatch_lqr takes
static bool wide_geometry_ok(int T, int B, int nx, int nu) {
  const bool exact = wide_shape(nx, nu), padded = wide_container_shape(nx, nu) && B % 4 == 0 && !container_disabled();
  return (exact || padded) && !wide_disabled() && B >= 4 && T >= 2 && (size_t)B * (nx + nu) * (nx + nu) * 4 < ((size_t)1 << 31);
}
static bool wide_ok(int mode, int nx, int nu, const LqrArgs &a) {
  // (the generated streams' own argument forms - c in two arrays, saved gains, x_init = 0 - are not its business)
  const bool plain = a.c_u == nullptr && a.Ks_in == nullptr && a.Vv_in == nullptr && a.Quu_out == nullptr && a.x_init != nullptr;
  return mode == kSolve && plain && wide_geometry_ok(a.T, a.B, nx, nu) && a.wsK != nullptr;
}
static int launch_lqr_wide(int nx, int nu, const LqrArgs &a, hipStream_t stream) {
  const dim3 grid((a.B + 15) / 16), block(256);
#define X(NX_, NU_)                                                                                        \
  if (nx == NX_ && nu == NU_) {                                                                            \
    constexpr int DB = 2, DF = 2;                                                                          \
    constexpr size_t lds = LqrWideLayout<NX_, NU_, DB, DF>::lds_bytes();                                   \
    static_assert(lds <= 160 * 1024, "rings beyond a CU's LDS");                                           \
    static const bool attr_once = [] {   /* the LDS request above 64 KB: set once per instantiation, not per launch */ \
      if (lds > 64 * 1024) {                                                                               \
        set_max_lds(reinterpret_cast<const void *>(&lqr_wide_kernel<NX_, NU_, DB, DF>), \
                                  (int)lds);                   \
        set_max_lds(reinterpret_cast<const void *>(&lqr_wide_kernel<NX_, NU_, DB, DF, false, true>), \
                                  (int)lds);                   \
      }                                                                                                    \
      return true;                                                                                         \
    }();                                                                                                   \
    (void)attr_once;                                                                                       \
    if (a.mask != nullptr) {                                                                               \
      DMPC_LAUNCH_GGL((lqr_wide_kernel<NX_, NU_, DB, DF, false, true>), grid, block, lds, stream, a);      \
    } else {                                                                                               \
      DMPC_LAUNCH_GGL((lqr_wide_kernel<NX_, NU_, DB, DF>), grid, block, lds, stream, a);                   \
    }                                                                                                      \
    return (int)hipGetLastError();                                                                         \
  }
  DMPC_LQR_WIDE_SHAPES(X)
#undef X
  LqrArgs p = a;
  p.nx_log = nx;
  p.nu_log = nu;
#define X(NX_, NU_)                                                                                        \
  if (nx <= NX_ && nu <= NU_) {                                                                            \
    constexpr int DB = 2, DF = 2;                                                                          \
    constexpr size_t lds = LqrWideLayout<NX_, NU_, DB, DF>::lds_bytes();                                   \
    if (lds > 64 * 1024)                                                                                   \
      set_max_lds(reinterpret_cast<const void *>(&lqr_wide_kernel<NX_, NU_, DB, DF, true>), \
                                (int)lds);                     \
    if (p.mask != nullptr) {                                                                               \
      if (lds > 64 * 1024)                                                                                 \
        set_max_lds(reinterpret_cast<const void *>(&lqr_wide_kernel<NX_, NU_, DB, DF, true, true>), \
                                  (int)lds);                   \
      DMPC_LAUNCH_GGL((lqr_wide_kernel<NX_, NU_, DB, DF, true, true>), grid, block, lds, stream, p);       \
    } else {                                                                                               \
      DMPC_LAUNCH_GGL((lqr_wide_kernel<NX_, NU_, DB, DF, true>), grid, block, lds, stream, p);             \
    }                                                                                                      \
    return (int)hipGetLastError();                                                                         \
  }
  DMPC_LQR_WIDE_CONTAINERS(X)
#undef X
  return DMPC_E_UNSUPPORTED;
}

static bool has_container(int nx, int nu) {
#define X(NX_, NU_) \
  if (nx <= NX_ && nu <= NU_) return true;
  DMPC_LQR_CONTAINERS(X)
  DMPC_LQR_WAVE_CONTAINERS(X)
#undef X
  return false;
}

// a wider problem (up to 32 states, 8 controls) inside a wavefront-per-trajectory instance: sweep on the matrix cores
// (lqr_wave_mfma_backward<..., PAD>), gains through HBM, then the forward-only container kernel
template <int NX, int NU>
static int launch_lqr_wave_container(int mode, int nx, int nu, const LqrArgs &a0, hipStream_t stream) {
  LqrArgs a = a0;
  a.nx_log = nx;
  a.nu_log = nu;
  const bool masked = a.mask != nullptr;
  if (mode != kForwardOnly) {
    if (a.Ks == nullptr) { a.Ks = a.wsK; a.ks = a.wsk; }
    if (a.Ks == nullptr) return DMPC_E_WORKSPACE;
    if (!wave_mfma_disabled()) {   // a size that IS one of the instances: the exact kernel (rollout in the same launch)
      const int rx = launch_lqr_wave_mfma_backward(nx, nu, masked, mode == kSolve, a, stream);
      if (rx != DMPC_E_UNSUPPORTED) return rx;
    }
    const int rc = launch_lqr_wave_container_sweep(NX, NU, masked, a, stream);
    if (rc != 0 || mode == kBackwardOnly) return rc;
  }
  {
    const int rc = launch_lqr_staged_forward(nx, nu, a, stream);   // (the rows through an LDS ring: lqr_staged_forward.hpp)
    if (rc != DMPC_E_UNSUPPORTED) return rc;
  }
  const dim3 grid((a.B + 3) / 4), block(256);
  if (masked) DMPC_LAUNCH_GGL((lqr_kernel<NX, NU, 64, true, kForwardOnly, false, true>), grid, block, 0, stream, a);
  else DMPC_LAUNCH_GGL((lqr_kernel<NX, NU, 64, false, kForwardOnly, false, true>), grid, block, 0, stream, a);
  return (int)hipGetLastError();
}

static int lqr_family(int nx, int nu) {
#define X(NX_, NU_, L_) \
  if (nx == NX_ && nu == NU_) return L_ == 16 ? 1 : 2;
  DMPC_LQR_SHAPES(X)
#undef X
  if (nx >= 1 && nu >= 1 && has_container(nx, nu)) return 4;
  if (nx >= 1 && nu >= 1 && nx + nu + 1 <= kGenericMaxCols && lqr_generic_lds_bytes(1, nx, nu, false) <= 64 * 1024) return 3;
  if (nx >= 1 && nu >= 1) return 5;     // any size: a workgroup per trajectory, matrices in the caller's workspace (lqr_tiled.hpp)
  return DMPC_E_UNSUPPORTED;
}

// a problem without a specialisation of its own, in the smallest container that holds it
template <int NX, int NU>
static int launch_lqr_container(int mode, int nx, int nu, const LqrArgs &a0, hipStream_t stream) {
  constexpr int GPB = 16;
  LqrArgs a = a0;
  a.nx_log = nx;
  a.nu_log = nu;
  const dim3 grid((a.B + GPB - 1) / GPB), block(256);
  const bool masked = a.mask != nullptr;
  const size_t lds_gain = (size_t)GPB * a.T * NU * (NX + 1) * sizeof(float);
#define DMPC_LAUNCH(MASKED, MODE, KLDS, SHMEM) \
  DMPC_LAUNCH_GGL((lqr_kernel<NX, NU, 16, MASKED, MODE, KLDS, true>), grid, block, SHMEM, stream, a)
  if (mode == kSolve) {
    if (lds_gain <= kGainLdsBudget) {
      if (masked) DMPC_LAUNCH(true, kSolve, true, lds_gain);
      else DMPC_LAUNCH(false, kSolve, true, lds_gain);
    } else {
      if (a.Ks == nullptr) { a.Ks = a.wsK; a.ks = a.wsk; }
      if (a.Ks == nullptr) return DMPC_E_WORKSPACE;
      if (masked) DMPC_LAUNCH(true, kSolve, false, 0);
      else DMPC_LAUNCH(false, kSolve, false, 0);
    }
  } else if (mode == kBackwardOnly) {
    if (masked) DMPC_LAUNCH(true, kBackwardOnly, false, 0);
    else DMPC_LAUNCH(false, kBackwardOnly, false, 0);
  } else {
    if (masked) DMPC_LAUNCH(true, kForwardOnly, false, 0);
    else DMPC_LAUNCH(false, kForwardOnly, false, 0);
  }
#undef DMPC_LAUNCH
  return (int)hipGetLastError();
}

static int dispatch_lqr(int mode, int nx, int nu, const LqrArgs &a, hipStream_t stream) {
  if (wide_ok(mode, nx, nu, a)) return launch_lqr_wide(nx, nu, a, stream);
#define X(NX_, NU_, L_) \
  if (nx == NX_ && nu == NU_) return launch_lqr<NX_, NU_, L_>(mode, a, stream);
  DMPC_LQR_SHAPES(X)
#undef X
  // (the generated streams' own argument forms - lqr_second_solve - stop here: nothing else reads them)
  if (a.c_u != nullptr || a.Ks_in != nullptr || a.Vv_in != nullptr || a.Quu_out != nullptr ||
      (a.x_init == nullptr && a.x != nullptr))
    return DMPC_E_UNSUPPORTED;
  if (lqr_family(nx, nu) == 4 && !container_disabled()) {
#define X(NX_, NU_) \
  if (nx <= NX_ && nu <= NU_) return launch_lqr_container<NX_, NU_>(mode, nx, nu, a, stream);
    DMPC_LQR_CONTAINERS(X)
#undef X
#define X(NX_, NU_) \
  if (nx <= NX_ && nu <= NU_) return launch_lqr_wave_container<NX_, NU_>(mode, nx, nu, a, stream);
    DMPC_LQR_WAVE_CONTAINERS(X)
#undef X
  }
  if (lqr_family(nx, nu) == 3 || lqr_family(nx, nu) == 4) {
    const int rc = launch_lqr_generic(mode, nx, nu, a, stream);
    if (rc != DMPC_E_UNSUPPORTED) return rc;      // (its matrices did not fit in LDS: the tiled kernel takes any size)
  }
  if (nx >= 1 && nu >= 1) return launch_lqr_tiled(mode, nx, nu, a, a.tiled_scratch, stream);
  return DMPC_E_UNSUPPORTED;
}

// DiffLqr.backward's second solve (differentiable_lqr.py:108-114) without a concatenated copy of its inputs: c = [cx; cu]
// as two arrays, x_init = 0, f = 0; with Ks != nullptr the re-solve from saved gains.  DMPC_E_UNSUPPORTED (nothing
// launched) unless a generated stream serves the size.
int lqr_second_solve(int T, int B, int nx, int nu, const float *C, const float *cx, const float *cu, const float *F,
                     const float *Ks, const float *Quu, const float *Qxu, float *x_out, float *u_out, int32_t *info,
                     hipStream_t stream) {
  LqrArgs a{T, B, C, cx, F, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x_out, u_out, info};
  a.c_u = cu;
  if (Ks != nullptr) {
    if (B % 4 != 0) return DMPC_E_UNSUPPORTED;
    a.Ks_in = Ks;
    a.Quu_in = Quu;
    a.Qxu_in = Qxu;
  }
  return dispatch_lqr(kSolve, nx, nu, a, stream);
}

// DiffLqr.backward (differentiable_lqr.py:78-142) in ONE launch that reads neither C nor c: the affine re-solve from the
// saving solve's K_t, Quu_t, Qxu_t on [grad_x; grad_u], whose rollout of d_tau forms lambda_t = V_t x_t + v_t and
// d_lambda_t = V_t dx_t + v'_t from the saved value functions Vv and writes dC, dc, dF, df, dx_init itself.
// DMPC_E_UNSUPPORTED (nothing launched) unless the generated stream serves the size.
int lqr_adjoint(int T, int B, int nx, int nu, const float *F, const float *grad_x, const float *grad_u, const float *Ks,
                const float *Quu, const float *Qxu, const float *Vv, const float *x, const float *u, int strict_math,
                float *d_x_init, float *dC, float *dc, float *dF, float *df, int32_t *info, hipStream_t stream) {
  if (B % 4 != 0) return DMPC_E_UNSUPPORTED;
  LqrArgs a{T, B, nullptr, grad_x, F, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, info};
  a.c_u = grad_u;
  a.Ks_in = Ks;
  a.Quu_in = Quu;
  a.Qxu_in = Qxu;
  a.Vv_in = Vv;
  a.tau_x = x;
  a.tau_u = u;
  a.dC = dC; a.dc = dc; a.dF = dF; a.df = df; a.dx0 = d_x_init;
  a.w_a = 0.5f;
  a.w_b = strict_math ? 0.5f : 1.0f;
  a.df_shift = strict_math ? 1 : 0;
  return dispatch_lqr(kSolve, nx, nu, a, stream);
}

}  // namespace dmpc

using namespace dmpc;

extern "C" {

int dmpc_version(void) { return DMPC_VERSION; }

int dmpc_lqr_kernel_family(int nx, int nu) { return lqr_family(nx, nu); }

int dmpc_lqr_solve_path(int T, int B, int nx, int nu) {
  if (T <= 0 || B <= 0) return DMPC_E_BADARG;
  if (wide_shape(nx, nu) && wide_geometry_ok(T, B, nx, nu)) return 9;   // lqr_wide_kernel (given the workspace)
#define X(NX_, NU_, L_) \
  if (nx == NX_ && nu == NU_) return solve_path<NX_, NU_, L_>(T, B);
  DMPC_LQR_SHAPES(X)
#undef X
  if (wide_container_shape(nx, nu) && wide_geometry_ok(T, B, nx, nu)) return 9;
  if (lqr_family(nx, nu) == 4 && !container_disabled()) return 7;   // a container kernel (lqr_kernel<..., PAD>)
  if (lqr_family(nx, nu) == 5) return 8;                              // lqr_tiled_kernel: any size
  return (lqr_family(nx, nu) == 3 || lqr_family(nx, nu) == 4) ? 0 : DMPC_E_UNSUPPORTED;
}

int dmpc_lqr_saving_available(int T, int B, int nx, int nu) {
  if (T <= 1 || B <= 0 || B % 4 != 0) return 0;
#define X(NX_, NU_, L_) \
  if (nx == NX_ && nu == NU_) return saving_available<NX_, NU_, L_>(T, B);
  DMPC_LQR_SHAPES(X)
#undef X
  return 0;
}

size_t dmpc_lqr_workspace_bytes(int T, int B, int nx, int nu) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return 0;
  // gains [T,B,nu,nx] + [T,B,nu], or - the generated stream at long horizons - rows of 12 floats [K_m | 0 | k_m | pad];
  // only touched when they do not fit in LDS (long horizons) or by the generic kernel
  // (rows of nx + nu + 1 floats for the LDS-DMA HIP kernel's workspace form)
  size_t bytes = (size_t)T * B * nu * (nx + nu + 1 > 12 ? nx + nu + 1 : 12) * sizeof(float);
  // (a shape padded inside an instance of the wide row kernel: gain rows of the INSTANCE's width - at most 8 rows of 25)
  if (wide_container_shape(nx, nu)) bytes = (size_t)T * B * 8 * 25 * sizeof(float);
  // the shapes beyond a wavefront's 64 columns (family 5) keep the matrices of every trajectory behind the gains
  if (lqr_family(nx, nu) == 5) bytes = round_up(bytes, 256) + (size_t)B * tiled_scratch_floats(nx, nu) * sizeof(float);
  return bytes;
}

static float *tiled_scratch_of(void *ws, int T, int B, int nx, int nu) {
  if (ws == nullptr || lqr_family(nx, nu) != 5) return nullptr;
  return reinterpret_cast<float *>(static_cast<char *>(ws) +
                                   round_up((size_t)T * B * nu * (nx + nu + 1 > 12 ? nx + nu + 1 : 12) * sizeof(float), 256));
}

int dmpc_lqr_solve(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                   const float *f, const float *x_init, const uint8_t *u_zero_mask, float *Ks_out,
                   float *ks_out, float *x_out, float *u_out, void *ws, size_t ws_bytes, int32_t *info,
                   dmpc_stream_t stream) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C || !c || !x_init || !x_out || !u_out || (T > 1 && !F)) return DMPC_E_BADARG;
  if ((Ks_out == nullptr) != (ks_out == nullptr)) return DMPC_E_BADARG;
  if (!aligned16(C) || !aligned16(c) || !aligned16(F) || !aligned16(f)) return DMPC_E_BADARG;
  LqrArgs a{T, B, C, c, F, f, x_init, u_zero_mask, Ks_out, ks_out, nullptr, nullptr, x_out, u_out, info};
  if (ws != nullptr) {
    if (ws_bytes < dmpc_lqr_workspace_bytes(T, B, nx, nu)) return DMPC_E_WORKSPACE;
    a.wsK = static_cast<float *>(ws);
    a.wsk = a.wsK + (size_t)T * B * nu * nx;
    a.tiled_scratch = tiled_scratch_of(ws, T, B, nx, nu);
  }
  return dispatch_lqr(kSolve, nx, nu, a, static_cast<hipStream_t>(stream));
}

int dmpc_lqr_solve_saving(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                          const float *f, const float *x_init, float *Ks_out, float *ks_out, float *Quu_out,
                          float *Qxu_out, float *Vv_out, float *x_out, float *u_out, int32_t *info, dmpc_stream_t stream) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C || !c || !F || !x_init || !x_out || !u_out || !Ks_out || !ks_out || !Quu_out || !Qxu_out || !Vv_out)
    return DMPC_E_BADARG;
  if (!aligned16(C) || !aligned16(c) || !aligned16(F) || !aligned16(f) || !aligned16(Quu_out)) return DMPC_E_BADARG;
  if (!dmpc_lqr_saving_available(T, B, nx, nu)) return DMPC_E_UNSUPPORTED;   // the stash form of the generated stream only
  LqrArgs a{T, B, C, c, F, f, x_init, nullptr, Ks_out, ks_out, nullptr, nullptr, x_out, u_out, info};
  a.Quu_out = Quu_out;
  a.Qxu_out = Qxu_out;
  a.Vv_out = Vv_out;
  a.info_store = true;
  return dispatch_lqr(kSolve, nx, nu, a, static_cast<hipStream_t>(stream));
}

int dmpc_lqr_saved_solve(int T, int B, int nx, int nu, const float *c, const float *F, const float *Ks,
                         const float *Quu, const float *Qxu, const float *x_init, float *x_out, float *u_out,
                         int32_t *info, dmpc_stream_t stream) {
  if (T <= 1 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!c || !F || !Ks || !Quu || !Qxu || !x_init || !x_out || !u_out) return DMPC_E_BADARG;
  if (!aligned16(c) || !aligned16(F) || !aligned16(Ks) || !aligned16(Quu) || !aligned16(Qxu)) return DMPC_E_BADARG;
  if (dmpc_lqr_solve_path(T, B, nx, nu) != 4 || B % 4 != 0) return DMPC_E_UNSUPPORTED;
  LqrArgs a{T, B, nullptr, c, F, nullptr, x_init, nullptr, nullptr, nullptr, nullptr, nullptr, x_out, u_out, info};
  a.Ks_in = Ks;
  a.Quu_in = Quu;
  a.Qxu_in = Qxu;
  return dispatch_lqr(kSolve, nx, nu, a, static_cast<hipStream_t>(stream));
}

int dmpc_lqr_backward_sweep_ws(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                               const float *f, const uint8_t *u_zero_mask, float *Ks_out, float *ks_out, void *ws,
                               size_t ws_bytes, int32_t *info, dmpc_stream_t stream) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!C || !c || !Ks_out || !ks_out || (T > 1 && !F)) return DMPC_E_BADARG;
  if (!aligned16(C) || !aligned16(c) || !aligned16(F) || !aligned16(f)) return DMPC_E_BADARG;
  LqrArgs a{T, B, C, c, F, f, nullptr, u_zero_mask, Ks_out, ks_out, nullptr, nullptr, nullptr, nullptr, info};
  if (ws != nullptr) {
    if (ws_bytes < dmpc_lqr_workspace_bytes(T, B, nx, nu)) return DMPC_E_WORKSPACE;
    a.tiled_scratch = tiled_scratch_of(ws, T, B, nx, nu);
  }
  return dispatch_lqr(kBackwardOnly, nx, nu, a, static_cast<hipStream_t>(stream));
}

int dmpc_lqr_backward_sweep(int T, int B, int nx, int nu, const float *C, const float *c, const float *F,
                            const float *f, const uint8_t *u_zero_mask, float *Ks_out, float *ks_out,
                            int32_t *info, dmpc_stream_t stream) {
  return dmpc_lqr_backward_sweep_ws(T, B, nx, nu, C, c, F, f, u_zero_mask, Ks_out, ks_out, nullptr, 0, info, stream);
}

int dmpc_lqr_forward_sweep(int T, int B, int nx, int nu, const float *Ks, const float *ks, const float *F,
                           const float *f, const float *x_init, const uint8_t *u_zero_mask, float *x_out,
                           float *u_out, int32_t *info, dmpc_stream_t stream) {
  if (T <= 0 || B <= 0 || nx <= 0 || nu <= 0) return DMPC_E_BADARG;
  if (!Ks || !ks || !x_init || !x_out || !u_out || (T > 1 && !F)) return DMPC_E_BADARG;
  if (!aligned16(F) || !aligned16(f)) return DMPC_E_BADARG;
  LqrArgs a{T, B, nullptr, nullptr, F, f, x_init, u_zero_mask, const_cast<float *>(Ks),
            const_cast<float *>(ks), nullptr, nullptr, x_out, u_out, info};
  return dispatch_lqr(kForwardOnly, nx, nu, a, static_cast<hipStream_t>(stream));
}

}  // extern "C"
