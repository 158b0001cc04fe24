// lqr_dma_kernel.hpp - fused LQR solve for the 16-lane row layout with the per-timestep blocks staged
// through LDS by LDS-DMA (global_load_lds_dwordx4), for B >= 4 and horizons whose gains fit in LDS.
//
// Same arithmetic as lqr_kernel (lqr_kernels.hpp; lqr/lqr_recursion.py:69-209); what changes is how the
// inputs reach the registers:
//   * a wavefront owns four CONSECUTIVE trajectories, so for one timestep its C, c, F, f blocks are four
//     contiguous runs of HBM (4*ns^2, 4*ns, 4*nx*ns, 4*nx floats).  Each run is copied to an LDS ring slot
//     by full-width 16-byte-per-lane DMA - every byte is fetched by exactly one coalesced request, no
//     VGPR is tied up, and the ring is DB (backward) / DF (forward) timesteps deep;
//   * the column-per-lane registers are then filled by ds_read_b32 with per-lane LDS indices computed once
//     (lane ns reads c / f directly: no merge of the affine column, no address arithmetic per step).
// The strided 4-byte global loads of lqr_kernel cost ~20 L1 tag lookups per instruction (64 lanes 40 B
// apart); that, not HBM latency, is what the register-prefetch version waits on.
//
// LDS-DMA is issued from inline asm, so hipcc neither tracks it in vmcnt nor drains it; completion is
// waited for with counted `s_waitcnt vmcnt(N)`: every wave issues exactly kDmaB (kDmaF) DMA instructions
// per timestep, in order, so "all but the youngest (D-1)*kDma operations" covers the slot being consumed.
// Out-of-range prefetches are clamped, never skipped, to keep that count exact; stores issued in between
// only make the wait more conservative.  A wave reads only the LDS it filled itself: no barrier.
#pragma once
#include "colwise.hpp"
#include "dma_gather.hpp"
#include "dpp_blocks_gen.hpp"
#include "lqr_kernels.hpp"
#include "riccati_blocks.hpp"

namespace dmpc {

__device__ __forceinline__ unsigned lds_byte_address(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

// One DMA instruction: 64 lanes x 16 bytes, global (sbase + voff) -> LDS (m0 + lane*16).  hipcc does not use M0
// in these kernels (no LDS-DMA builtin, no movrel), so it is written without save/restore; the instruction
// between the M0 write and the DMA provides the required wait state.
__device__ __forceinline__ void dma16(unsigned voff, unsigned lds_dst, const void *sbase) {
#ifdef DMPC_TIMING_NO_DMA      // timing experiments only
  return;
#endif
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(lds_dst), "s"(sbase)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field on gfx9");
#ifndef DMPC_TIMING_NO_WAIT   // timing experiments only (results are wrong without the wait)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// Copy a run of BYTES bytes (multiple of 16) starting at `src` (wave-uniform) into LDS at `dst`.
template <int BYTES>
__device__ __forceinline__ void dma_run(const char *src, unsigned dst, unsigned lane_off /* lane*16 */) {
  static_assert(BYTES % 16 == 0, "runs are whole 16-byte units");
  constexpr int FULL = BYTES / 1024, REM = BYTES % 1024;
#pragma unroll
  for (int q = 0; q < FULL; ++q) dma16(lane_off + q * 1024, dst + q * 1024, src);
  // Partial last chunk: let hipcc mask the lanes.  (Setting EXEC by hand inside the asm statement right in front
  // of the DMA was measured to corrupt the copy - an EXEC write needs a wait state before a VMEM instruction.)
  if constexpr (REM > 0) {
    if (lane_off < (unsigned)REM) dma16(lane_off + FULL * 1024, dst + FULL * 1024, src);
  }
}
template <int BYTES>
constexpr int dma_count() {
  return BYTES / 1024 + (BYTES % 1024 ? 1 : 0);
}

// KHBM: the gain rows [K_m | 0 | k_m] of every step go to the caller's workspace ([T,B,NU,NS+1] floats) instead of LDS and
// come back to the rollout through its ring, with F and f (the generated streams' `ghbm` form in HIP): what lets (8,4) and
// (12,3) use this kernel at T = 50 - their gain rows alone (166 / 154 KB) left no room for the rings.
template <int NX, int NU, int DB, int DF, bool KHBM = false>
struct LqrDmaLayout {
  static constexpr int NS = NX + NU;
  static constexpr int K_FL = KHBM ? 4 * NU * (NS + 1) : 0;     // gain rows of the wave's four trajectories, one step
  static constexpr int C_FL = 4 * NS * NS, c_FL = 4 * NS, F_FL = 4 * NX * NS, f_FL = 4 * NX;  // floats per wave-step
  static constexpr int OFF_C = 0, OFF_c = OFF_C + C_FL, OFF_F = OFF_c + c_FL, OFF_f = OFF_F + F_FL;
  // Round 4: the slot is filled by per-lane GATHER DMA (dma_gather.hpp; the scheme of costate_dma_kernel and of the generated
  // streams): lane l of instruction q copies chunk 64 q + l of [C | c | F | f] from its own 64-bit address.  Before, every
  // run was a DMA of its own from a uniform base - 4 (small shapes) to 7 instructions per step, each with its M0 write, a
  // 64-bit scalar pointer step and, for the partial last kilobyte, an exec-mask bracket: ~20 scalar instructions per DMA,
  // 42 % of the (4,4) kernel's instructions (profiles/r04/knobs_4_4.txt).
  static constexpr int CH_B = (OFF_f + f_FL) / 4;          // 16-byte chunks of a backward slot
  static constexpr int CH_F = (F_FL + f_FL + K_FL) / 4;    // ... of a forward slot [F | f] ([F | f | K rows] with KHBM)
  static constexpr int kDmaB = (CH_B + 63) / 64, kDmaF = (CH_F + 63) / 64;
  static constexpr int SLOT_B = kDmaB * 256;   // floats: whole kilobytes (the lanes past the last chunk re-fetch chunk 0)
  static constexpr int SLOT_F = kDmaF * 256;
  static constexpr int RING_FL = (DB * SLOT_B > DF * SLOT_F) ? DB * SLOT_B : DF * SLOT_F;  // per wave
  static constexpr size_t lds_bytes(int T) {
    return (size_t)4 * RING_FL * 4 + (KHBM ? 0 : (size_t)16 * T * NU * (NS + 1) * 4);  // rings + gain rows [K_m | 0 | k_m]
  }
};

template <int NX, int NU, int DB, int DF, bool KHBM = false>
__global__ __launch_bounds__(256) void lqr_dma_kernel(const LqrArgs a) {
  using Lay = LqrDmaLayout<NX, NU, DB, DF, KHBM>;
  constexpr int NS = NX + NU, L = 16, KROW = NS + 1;  // gain rows share the shape of an [F_i | f_i] row
  static_assert(NS + 1 <= L, "augmented columns must fit a DPP row");
  static_assert((DB - 1) * Lay::kDmaB <= 63 && (DF - 1) * Lay::kDmaF <= 63, "ring too deep for vmcnt");
  using G = Group<L>;
  using Blk = RiccatiBlocks<NX, NU, L>;

  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;   // trajectory within the wave
  const int lane = lane64 & 15;
  int b0 = ((int)blockIdx.x * 4 + wave) * 4;  // first trajectory of this wave
  if (b0 > a.B - 4) b0 = a.B - 4;             // last wave overlaps its neighbour instead of running ragged
  b0 = __builtin_amdgcn_readfirstlane(b0);
  const int b = b0 + r;

  extern __shared__ float lds[];
  float *ring = lds + wave * Lay::RING_FL;
  float *kl = lds + 4 * Lay::RING_FL + (size_t)(wave * 4 + r) * T * NU * KROW;  // gains of this trajectory (not KHBM)
  float *kw = KHBM ? a.wsK + (size_t)b * NU * KROW : nullptr;                   // KHBM: + t * B * NU * KROW in the workspace
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));
  const unsigned lane_off = (unsigned)lane64 * 16u;

  const bool col_aff = lane == NS;
  const int lane_c = lane < NS ? lane : NS - 1;
  const bool k_lane = lane < NX || col_aff;
  int info_bits = 0;

  // ------------------------------------------------------------------ backward Riccati sweep
  {
    // per-lane source pointers of the gather groups: chunk g = 64 q + lane64 of the slot [C | c | F | f].  Every array steps
    // back by one timestep per fetch; F (and f) have no slice T-1, so their lanes start at T-2 and sit out the first step.
    // A missing f is fetched from c (never used: step() zeroes the affine column of F~ then).
    unsigned long long ptr[Lay::kDmaB], str[Lay::kDmaB], str1[Lay::kDmaB];
#pragma unroll
    for (int q = 0; q < Lay::kDmaB; ++q) {
      const int g = q * 64 + lane64;
      const int gg = g < Lay::CH_B ? g : 0;          // padding lanes repeat chunk 0 of C (lands past the slot's last chunk)
      const char *base;
      size_t per;
      int g0;
      bool dyn = false;
      if (gg < Lay::OFF_c / 4) { base = (const char *)a.C; per = (size_t)NS * NS * 4; g0 = 0; }
      else if (gg < Lay::OFF_F / 4) { base = (const char *)a.c; per = (size_t)NS * 4; g0 = Lay::OFF_c / 4; }
      else if (gg < Lay::OFF_f / 4) { base = (const char *)(T > 1 ? a.F : a.C); per = (size_t)NX * NS * 4; g0 = Lay::OFF_F / 4; dyn = true; }
      else { base = (const char *)(has_f ? a.f : a.c); per = (size_t)NX * 4; g0 = Lay::OFF_f / 4; dyn = true; }
      const int t0 = dyn ? (T > 1 ? T - 2 : 0) : T - 1;
      ptr[q] = (unsigned long long)base + ((size_t)t0 * B + (size_t)b0) * per + (size_t)(gg - g0) * 16 -
               (unsigned long long)(q % 4) * 1024u;
      str[q] = 0ull - (unsigned long long)(B * per);
      str1[q] = dyn ? 0ull : str[q];
    }
    int ti = T - 1;  // timesteps still to step back over
    auto issue_next = [&](int slot) {
#ifndef DMPC_TIMING_NO_DMA
      const unsigned dst = ring_addr + (unsigned)slot * (Lay::SLOT_B * 4);
      static_for<0, Lay::kDmaB>([&](auto q) {  // the instruction offset is 13 bits signed: M0 moves every 4 KB
        if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
        dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
      });
#endif
      if (ti > 0) {  // past t = 0 the same blocks are fetched again (never consumed): the count per step stays exact
        const bool first = ti == T - 1;
#pragma unroll
        for (int q = 0; q < Lay::kDmaB; ++q) ptr[q] += first ? str1[q] : str[q];
        --ti;
      }
    };
    // per-lane LDS indices (floats, relative to a slot), computed once: lane ns reads c / f directly
    int iq[NS], ifc[NX];
#pragma unroll
    for (int i = 0; i < NS; ++i)
      iq[i] = col_aff ? (Lay::OFF_c + r * NS + i) : (Lay::OFF_C + r * NS * NS + i * NS + lane_c);
#pragma unroll
    for (int k = 0; k < NX; ++k)
      ifc[k] = col_aff ? (Lay::OFF_f + r * NX + k) : (Lay::OFF_F + r * NX * NS + k * NS + lane_c);
    auto read_slot = [&](const float *slot, float (&Qn)[NS], float (&Fn)[NX]) {
#ifdef DMPC_TIMING_NO_LDSREAD
#pragma unroll
      for (int i = 0; i < NS; ++i) Qn[i] = 1.0f + 0.01f * i + 0.001f * lane;
#pragma unroll
      for (int k = 0; k < NX; ++k) Fn[k] = 0.1f + 0.01f * k;
      return;
#endif
#pragma unroll
      for (int i = 0; i < NS; ++i) Qn[i] = slot[iq[i]];
#pragma unroll
      for (int k = 0; k < NX; ++k) Fn[k] = slot[ifc[k]];
    };

    float V[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) V[i] = 0.f;

    auto step = [&](int t, float (&Q)[NS], float (&Fc)[NX]) {
      const size_t tb = (size_t)t * B + b;
      if (t < T - 1) {
        if (!has_f) {
#pragma unroll
          for (int k = 0; k < NX; ++k) Fc[k] = col_aff ? 0.f : Fc[k];
        }
        float W[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) W[i] = col_aff ? V[i] : 0.f;
        Blk::vf(W, V, Fc);   // lqr_recursion.py:89,96
        Blk::ftw(Q, Fc, W);
      }
      float Kt[NU], R[NU];
      if constexpr (NU >= kRowGainsFromNu) {   // Gauss-Jordan on the rows where they lie (riccati_blocks.hpp)
        float Qu[NU];
        bool act[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          Qu[m] = Q[NX + m];
          act[m] = false;
        }
        if (gains_on_rows<NX, NU, L, false>(Qu, act, lane, Kt, R, t > 0)) info_bits |= 1;   // :112-120
      } else {
        float Quu[NU][NU];
        static_for<0, NU>([&](auto l) {
#pragma unroll
          for (int m = 0; m < NU; ++m) Quu[m][l.value] = G::template bcast<NX + l.value>(Q[NX + m]);
        });
#pragma unroll
        for (int m = 0; m < NU; ++m) Kt[m] = Q[NX + m];
        if constexpr (NU == 1) {
          Kt[0] = -(fast_rcp(Quu[0][0]) * Kt[0]);  // :112-115
          if (Quu[0][0] == 0.f) info_bits |= 1;
        } else {
          float A[NU][NU], rinv[NU];
          int piv[NU];
#pragma unroll
          for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int l = 0; l < NU; ++l) A[m][l] = Quu[m][l];
          if (lu_factor_rinv<NU>(A, piv, rinv)) info_bits |= 1;  // :116-120
          lu_solve_rinv<NU>(A, piv, rinv, Kt);
#pragma unroll
          for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
        }
        if (t > 0) {
#pragma unroll
          for (int m = 0; m < NU; ++m) {
            R[m] = Q[NX + m];
#pragma unroll
            for (int l = 0; l < NU; ++l) R[m] = fmaf(Quu[m][l], Kt[l], R[m]);
          }
        }
      }
      if (lane <= NS) {  // row m of the gains as [K_m (nx) | 0 (nu) | k_m]: the forward sweep reads it like an F row
        if constexpr (KHBM) {
          float *row = kw + (size_t)t * B * (NU * KROW) + lane;
#pragma unroll
          for (int m = 0; m < NU; ++m) row[m * KROW] = k_lane ? Kt[m] : 0.f;
        } else {
#pragma unroll
          for (int m = 0; m < NU; ++m) kl[(t * NU + m) * KROW + lane] = k_lane ? Kt[m] : 0.f;
        }
      }
      if (k_lane) {
        if (a.Ks != nullptr) {
#pragma unroll
          for (int m = 0; m < NU; ++m) {
            if (col_aff) a.ks[tb * NU + m] = Kt[m];
            else a.Ks[(tb * NU + m) * NX + lane] = Kt[m];
          }
        }
      }
      if (t > 0) {  // :151-152
#pragma unroll
        for (int i = 0; i < NX; ++i) V[i] = Q[i];
        Blk::vupd(V, Q, Kt, R);
      }
    };

    // Software pipeline (DB even): at the start of step t the DMA for step t-DB goes into the slot whose
    // contents (step t) were moved to registers one step ago, the slot of step t-1 is waited for and read into
    // the other register set, and only then step t is computed from its own set.
    static_assert(DB % 2 == 0 && DB >= 2, "two alternating register sets");
    float QA[NS], FA[NX], QB[NS], FB[NX];
    static_for<0, DB>([&](auto j) { issue_next(j.value); });
    wait_vmcnt<(DB - 1) * Lay::kDmaB>();
    read_slot(ring, QA, FA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // these reads are in before the loop's first fetch refills their slot (round 4:
    // a cache-resident refill was seen to overtake them in lqr_wide_kernel - tiny problems, wrong rows at the first step)
#ifdef DMPC_TIMING_SKIP_BWD
    for (int t0 = -1; t0 >= 0; t0 -= DB) {
#else
    for (int t0 = T - 1; t0 >= 0; t0 -= DB) {
#endif
      static_for<0, DB>([&](auto j) {
        const int t = t0 - j.value;
        if (t >= 0) {
          constexpr int nslot = (j.value + 1) % DB;
          issue_next(j.value);
          wait_vmcnt<(DB - 1) * Lay::kDmaB>();
          if constexpr (j.value % 2 == 0) {
            read_slot(ring + nslot * Lay::SLOT_B, QB, FB);
            step(t, QA, FA);
          } else {
            read_slot(ring + nslot * Lay::SLOT_B, QA, FA);
            step(t, QB, FB);
          }
        }
      });
    }
    wait_vmcnt<0>();  // the ring memory is reused by the forward sweep (KHBM: and this wave's gain rows have reached L2)
    if constexpr (KHBM) __threadfence();
  }

  // ------------------------------------------------------------------ forward rollout (lqr_recursion.py:160-200)
  {
    // gather pointers of the forward slot [F_t | f_t]; the arrays have T-1 slices: beyond that the last one is fetched again
    unsigned long long ptr[Lay::kDmaF], str[Lay::kDmaF];
    bool kstep[Lay::kDmaF];      // KHBM: lanes that fetch gain rows (T slices: they take the last pointer step alone)
#pragma unroll
    for (int q = 0; q < Lay::kDmaF; ++q) {
      const int g = q * 64 + lane64;
      const int gg = g < Lay::CH_F ? g : 0;
      const bool isk = KHBM && gg >= (Lay::F_FL + Lay::f_FL) / 4;      // the wave's gain rows of the step (workspace)
      const bool isf = !isk && gg >= Lay::F_FL / 4;
      const char *base = isk ? (const char *)a.wsK : isf ? (const char *)(has_f ? a.f : a.C) : (const char *)(T > 1 ? a.F : a.C);
      const size_t per = isk ? (size_t)NU * KROW * 4 : isf ? (size_t)NX * 4 : (size_t)NX * NS * 4;
      const int g0 = isk ? (Lay::F_FL + Lay::f_FL) / 4 : isf ? Lay::F_FL / 4 : 0;
      ptr[q] = (unsigned long long)base + (size_t)b0 * per + (size_t)(gg - g0) * 16 - (unsigned long long)(q % 4) * 1024u;
      str[q] = (unsigned long long)(B * per);
      kstep[q] = isk;
    }
    int ti = 0;
    auto issue_next = [&](int slot) {
#ifndef DMPC_TIMING_NO_DMA
      const unsigned dst = ring_addr + (unsigned)slot * (Lay::SLOT_F * 4);
      static_for<0, Lay::kDmaF>([&](auto q) {
        if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
        dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
      });
#endif
      if (ti < T - 2) {  // F/f have T-1 slices; beyond that the last one is fetched again (never consumed)
#pragma unroll
        for (int q = 0; q < Lay::kDmaF; ++q) ptr[q] += str[q];
        ++ti;
      } else if (KHBM && ti == T - 2) {   // the gain rows have a slice T-1: their lanes step once more
#pragma unroll
        for (int q = 0; q < Lay::kDmaF; ++q) ptr[q] += kstep[q] ? str[q] : 0ull;
        ++ti;
      }
    };
    // Lane i < NX owns row i of [F_t | f_t] (ring slot), lane NX+m owns row m of [K_t | 0 | k_t] (gain rows in
    // LDS); both are NS+1 floats, so ONE instruction stream serves both:
    //     acc  = M[NS] + sum_{j<NX} x[j] * M[j]        -> lanes NX+m: u_t[m] ; lanes < NX: f + Fx x
    //     acc += sum_m u[m] * M[NX+m]                   -> lanes < NX: x_{t+1}
    // with x[j], u[m] broadcast from lanes j, NX+m of the row (lqr_recursion.py:177,189).
    const bool row_x = lane < NX;
    const bool row_u = lane >= NX && lane < NS;
    const int lane_x = row_x ? lane : NX - 1;
    const int m_own = row_u ? lane - NX : 0;
    const float *xbase = ring + r * NX * NS + lane_x * NS;           // + slot * SLOT_F : row lane_x of F_t
    const float *xaff = ring + Lay::F_FL + r * NX + lane_x;          // + slot * SLOT_F : f_t[lane_x]
    const float *ubase = KHBM ? ring + Lay::F_FL + Lay::f_FL + (r * NU + m_own) * KROW   // + slot * SLOT_F: row m of the gains
                              : kl + m_own * KROW;                                      // + t * NU * KROW
    auto read_rows = [&](int t, int slot, float (&Mn)[NS + 1]) {
      const float *urow = KHBM ? ubase + slot * Lay::SLOT_F : ubase + (size_t)t * (NU * KROW);
      const float *row = row_u ? urow : xbase + slot * Lay::SLOT_F;
      const float *aff = row_u ? urow + NS : xaff + slot * Lay::SLOT_F;
#pragma unroll
      for (int j = 0; j < NS; ++j) Mn[j] = row[j];
      Mn[NS] = aff[0];
    };
    float xv = row_x ? a.x_init[(size_t)b * NX + lane] : 0.f;  // lane j < NX: x[j]; lane NX+m: u[m]
    auto fstep = [&](int t, const float (&Mc)[NS + 1]) {
      const size_t tb = (size_t)t * B + b;
      float M[NS + 1];
#pragma unroll
      for (int j = 0; j <= NS; ++j) M[j] = Mc[j];
      float acc = (row_u || has_f) ? M[NS] : 0.f;
      Blk::dot_x(acc, xv, M);
      if (row_u) xv = acc;
      if (row_x) a.x[tb * NX + lane] = xv;
      else if (row_u) a.u[tb * NU + m_own] = xv;
      Blk::dot_u(acc, xv, M);  // for t = T-1 this consumes a re-fetched F_{T-2}: the result is never used
      if (row_x) xv = acc;
    };
    static_assert(DF % 2 == 0 && DF >= 2, "two alternating register sets");
    float MA[NS + 1], MB[NS + 1];
    static_for<0, DF>([&](auto j) { issue_next(j.value); });
    wait_vmcnt<(DF - 1) * Lay::kDmaF>();
    read_rows(0, 0, MA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // these reads are in before the loop's first fetch refills their slot (round 4:
    // a cache-resident refill was seen to overtake them in lqr_wide_kernel - tiny problems, wrong rows at the first step)
#ifdef DMPC_TIMING_SKIP_FWD
    for (int t0 = T; t0 < T; t0 += DF) {
#else
    for (int t0 = 0; t0 < T; t0 += DF) {
#endif
      static_for<0, DF>([&](auto j) {
        const int t = t0 + j.value;
        if (t < T) {
          constexpr int nslot = (j.value + 1) % DF;
          issue_next(j.value);
          wait_vmcnt<(DF - 1) * Lay::kDmaF>();
          const int tn = t + 1 < T ? t + 1 : T - 1;
          if constexpr (j.value % 2 == 0) {
            read_rows(tn, nslot, MB);
            fstep(t, MA);
          } else {
            read_rows(tn, nslot, MA);
            fstep(t, MB);
          }
        }
      });
    }
    // NaN/Inf propagate through the recursion, so the state after the last step tells whether anything broke
    const bool bad = !is_finite(xv);
    if (bad) info_bits |= 2;
  }
  if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
