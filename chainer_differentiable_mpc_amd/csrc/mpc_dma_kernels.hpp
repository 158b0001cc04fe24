// mpc_dma_kernels.hpp - MPCstep.backward_rec (mpc/mpc_step.py:70-173) with its per-timestep inputs staged through an
// LDS ring by per-lane gather LDS-DMA (the scheme of costate_dma_kernel.hpp).  Same arithmetic as
// mpc_backward_rec_kernel (mpc_kernels.hpp), per-trajectory PNQP termination only.
//
// Why: with one wavefront per SIMD the sweep is a chain of T dependent steps, and the phase stamps of
// scripts/microbench/mpc_phases.hip put 29 % (nx=3, nu=1) / 16 % (8, 2) of a step into ISSUING the next inputs:
// 11 / 25 scalar loads with 64-bit per-lane address arithmetic, plus waits hipcc inserts at the loop header because
// its register scoreboard is merged over the back edge (the banks rotate, the waits do not).  Here a wavefront owns
// four consecutive trajectories, so every input array contributes one contiguous run per timestep; all of them are
// fetched by kDma (1 / 4) DMA instructions whose per-lane source pointers step back by one timestep, DB - 1 steps
// ahead of the arithmetic, and nothing the compiler tracks is in flight across the loop.
#pragma once
#include "box_ddp_kernels.hpp"
#include "dma_gather.hpp"
#include "lqr_dma_kernel.hpp"
#include "mpc_kernels.hpp"

namespace dmpc {

template <int NX, int NU, int DB>
struct MpcBackDmaLayout {
  static constexpr int NS = NX + NU;
  // 16-byte chunks of one wave-step (four trajectories): [C | c | F | f | u | lower | upper | x]
  static constexpr int CH_C = 0, CH_c = CH_C + NS * NS, CH_F = CH_c + NS, CH_f = CH_F + NX * NS, CH_u = CH_f + NX;
  static constexpr int CH_lo = CH_u + NU, CH_hi = CH_lo + NU, CH_x = CH_hi + NU, CH_END = CH_x + NX;
  static constexpr int OFF_C = CH_C * 4, OFF_c = CH_c * 4, OFF_F = CH_F * 4, OFF_f = CH_f * 4, OFF_u = CH_u * 4;
  static constexpr int OFF_lo = CH_lo * 4, OFF_hi = CH_hi * 4, OFF_x = CH_x * 4;   // in floats
  static constexpr int kDma = (CH_END + 63) / 64;   // gather DMAs per step; padding lanes repeat chunk 0 of C
  static constexpr int SLOT = kDma * 256;           // floats per wave and timestep (whole 1 KB pieces)
  static constexpr size_t lds_bytes() { return (size_t)4 * DB * SLOT * 4; }
};

// requires B % 4 == 0, 16-byte aligned arrays, a.sync == nullptr (checked by the launcher)
template <int NX, int NU, int DB>
__device__ __forceinline__ void mpc_backward_rec_dma_body(const MpcBackArgs &a, const int block) {
  using Lay = MpcBackDmaLayout<NX, NU, DB>;
  constexpr int NS = NX + NU, L = 16;
  static_assert(NS + 1 <= L, "augmented columns must fit the lane group");
  static_assert((DB - 1) * Lay::kDma <= 63, "ring too deep for vmcnt");
  static_assert(DB % 2 == 0 && DB >= 2, "two alternating register sets");
  using G = Group<L>;
  using Blk = RiccatiBlocks<NX, NU, L>;

  if (a.done != nullptr && *a.done != 0) return;  // uniform: the iLQR loop has stopped
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;  // trajectory within the wave
  const int lane = lane64 & 15;
  const int b0 = __builtin_amdgcn_readfirstlane((block * 4 + wave) * 4);
  if (b0 >= a.B) return;      // whole wavefront (B % 4 == 0); no workgroup barrier below
  const int b = b0 + r;
  const bool has_f = a.f != nullptr;
  const bool expand = a.states != nullptr;
  const bool col_aff = lane == NS;
  const int lane_c = lane < NS ? lane : NS - 1;

  extern __shared__ float lds[];
  float *ring = lds + wave * (DB * Lay::SLOT);
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));

  // per-lane source pointers of the gather groups: chunk g = 64 q + lane64 of the slot.  Every array steps back by one
  // timestep per fetch; F / f have no slice T-1, so their lanes start at T-2 and sit out the first step.  Arrays that
  // are absent (f, states) and padding lanes re-fetch chunk 0 of C (never read).
  unsigned long long ptr[Lay::kDma], str[Lay::kDma], str1[Lay::kDma];
#pragma unroll
  for (int q = 0; q < Lay::kDma; ++q) {
    const int g = q * 64 + lane64;
    const char *base = (const char *)a.C;
    size_t per = (size_t)NS * NS * 4;
    int g0 = Lay::CH_C;
    bool isF = false;
    if (g >= Lay::CH_END) { g0 = g; }
    else if (g >= Lay::CH_x) { if (expand) { base = (const char *)a.states; per = (size_t)NX * 4; g0 = Lay::CH_x; } else g0 = g; }
    else if (g >= Lay::CH_hi) { base = (const char *)a.upper; per = (size_t)NU * 4; g0 = Lay::CH_hi; }
    else if (g >= Lay::CH_lo) { base = (const char *)a.lower; per = (size_t)NU * 4; g0 = Lay::CH_lo; }
    else if (g >= Lay::CH_u) { base = (const char *)a.controls; per = (size_t)NU * 4; g0 = Lay::CH_u; }
    else if (g >= Lay::CH_f) { if (has_f && T > 1) { base = (const char *)a.f; per = (size_t)NX * 4; g0 = Lay::CH_f; isF = true; } else g0 = g; }
    else if (g >= Lay::CH_F) { if (T > 1) { base = (const char *)a.F; per = (size_t)NX * NS * 4; g0 = Lay::CH_F; isF = true; } else g0 = g; }
    else if (g >= Lay::CH_c) { base = (const char *)a.c; per = (size_t)NS * 4; g0 = Lay::CH_c; }
    const int t0 = isF ? T - 2 : T - 1;
    ptr[q] = (unsigned long long)base + ((size_t)t0 * B + (size_t)b0) * per + (size_t)(g - g0) * 16 - (unsigned long long)(q % 4) * 1024u;
    str[q] = 0ull - (unsigned long long)(B * per);
    str1[q] = isF ? 0ull : str[q];
  }
  int ti = T - 1;  // timesteps still to step back over
  auto issue_next = [&](int slot) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + (unsigned)slot * (Lay::SLOT * 4));
    static_for<0, Lay::kDma>([&](auto q) {  // the instruction offset is 13 bits signed: M0 moves every 4 KB
      if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
      dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
    });
    if (ti > 0) {  // past t = 0 the same blocks are fetched again (never consumed): the count per step stays exact
      const bool first = ti == T - 1;
#pragma unroll
      for (int q = 0; q < Lay::kDma; ++q) ptr[q] += first ? str1[q] : str[q];
      --ti;
    }
  };

  // per-lane LDS indices (floats, relative to a slot): lane j < ns walks column j of [C_t; F_t], lane ns walks c_t / f_t
  const int q_base = col_aff ? Lay::OFF_c + r * NS : Lay::OFF_C + r * NS * NS + lane_c;
  const int q_step = col_aff ? 1 : NS;
  const int f_base = col_aff ? Lay::OFF_f + r * NX : Lay::OFF_F + r * NX * NS + lane_c;
  const int i_tau = lane < NX ? Lay::OFF_x + r * NX + lane : Lay::OFF_u + r * NU + (lane_c - NX);
  const int i_u = Lay::OFF_u + r * NU, i_lo = Lay::OFF_lo + r * NU, i_hi = Lay::OFF_hi + r * NU;
  struct Slot {
    float Q[NS], Fc[NX];      // [C_t | c_t] rows, [F_t | f_t] rows
    float uc[NU], lb[NU], ub[NU];
    float tau;                // lane j < ns: [x_t; u_t][j] (need_expand), else 0
  };
  auto read_slot = [&](const float *slot, Slot &sl) {
#pragma unroll
    for (int i = 0; i < NS; ++i) sl.Q[i] = slot[q_base + i * q_step];
#pragma unroll
    for (int k = 0; k < NX; ++k) sl.Fc[k] = slot[f_base + k * q_step];
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      sl.uc[m] = slot[i_u + m];
      sl.lb[m] = slot[i_lo + m];
      sl.ub[m] = slot[i_hi + m];
    }
    sl.tau = slot[i_tau];
  };

  float V[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) V[i] = 0.f;
  float kprev[NU];
#pragma unroll
  for (int m = 0; m < NU; ++m) kprev[m] = 0.f;
  int n_total = 0;
  int info_bits = 0;

  auto step = [&](int t, const Slot &sl) {  // the step of mpc_backward_rec_body
    const size_t tb = (size_t)t * B + b;
    float Q[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) Q[i] = sl.Q[i];
    if (expand) {   // c_hat = C tau + c: row sums over the matrix columns land in the affine column   :305-317
      const float tau = lane < NS ? sl.tau : 0.f;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const float s = group_sum<L>(lane < NS ? Q[i] * tau : 0.f);
        Q[i] = col_aff ? Q[i] + s : Q[i];
      }
    }
    if (t < T - 1) {
      float Fc[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) Fc[k] = (col_aff && !has_f) ? 0.f : sl.Fc[k];
      float W[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) W[i] = col_aff ? V[i] : 0.f;
      Blk::vf(W, V, Fc);   // mpc_step.py:110,116
      Blk::ftw(Q, Fc, W);
    }
    // every lane gets Quu and qu                                             :119-124
    float Quu[NU][NU], qu[NU], lo[NU], hi[NU];
    static_for<0, NU>([&](auto l) {
#pragma unroll
      for (int m = 0; m < NU; ++m) Quu[m][l.value] = G::template bcast<NX + l.value>(Q[NX + m]);
    });
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      qu[m] = G::template bcast<NS>(Q[NX + m]);
      lo[m] = sl.lb[m] - sl.uc[m];  // :136-138
      hi[m] = sl.ub[m] - sl.uc[m];
    }
    // k_t: box QP, warm-started from the later timestep                        :141-146
    PnqpResult<NU> qp;
    float kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) kt[m] = kprev[m];
    pnqp_solve_rows<NU>(Quu, qu, lo, hi, kt, /*warm=*/t != T - 1, a.n_qp_iter, qp);
    n_total += 1 + qp.it;
    if (!qp.converged) info_bits |= 4;
#pragma unroll
    for (int m = 0; m < NU; ++m) kprev[m] = kt[m];
    // K_t = -LU_free^-1 Qux with the rows of clamped controls zeroed            :147-157
    float Kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = qp.free_[m] ? Q[NX + m] : 0.f;
    if constexpr (NU == 1) {
      Kt[0] = -(qp.rinv[0] * Kt[0]);
    } else {
      lu_solve_rinv<NU>(qp.fac, qp.piv, qp.rinv, Kt);
#pragma unroll
      for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
    }
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = col_aff ? kt[m] : Kt[m];  // affine column carries k_t
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      if (col_aff) a.ks[tb * NU + m] = Kt[m];
      else if (lane < NX) a.Ks[(tb * NU + m) * NX + lane] = Kt[m];
    }
    if (t > 0) {  // V, v from the UNMASKED blocks                                :165-166
      float R[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        R[m] = Q[NX + m];
#pragma unroll
        for (int l = 0; l < NU; ++l) R[m] = fmaf(Quu[m][l], Kt[l], R[m]);
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) V[i] = Q[i];
      Blk::vupd(V, Q, Kt, R);
    }
  };

  // Software pipeline of costate_dma_kernel: at step t the DMA for step t - DB goes into the slot whose contents went
  // to registers one step ago, the slot of step t - 1 is waited for and read into the other register set, then step t
  // is computed from its own set.  The gain stores issued in between only make the counted wait more conservative.
  Slot sa, sb;
  static_for<0, DB>([&](auto j) { issue_next(j.value); });
  wait_vmcnt<(DB - 1) * Lay::kDma>();
  read_slot(ring, sa);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // these reads are in before the loop's first fetch refills their slot (round 4:
  // a cache-resident refill was seen to overtake them in lqr_wide_kernel - tiny problems, wrong rows at the first step)
  for (int t0 = T - 1; t0 >= 0; t0 -= DB) {
    static_for<0, DB>([&](auto j) {
      const int t = t0 - j.value;
      if (t >= 0) {
        constexpr int nslot = (j.value + 1) % DB;
        issue_next(j.value);
        wait_vmcnt<(DB - 1) * Lay::kDma>();
        if constexpr (j.value % 2 == 0) {
          read_slot(ring + nslot * Lay::SLOT, sb);
          step(t, sa);
        } else {
          read_slot(ring + nslot * Lay::SLOT, sa);
          step(t, sb);
        }
      }
    });
  }
  wait_vmcnt<0>();
  if (lane == 0) {
    a.n_qp_total[b] = n_total;
    if (a.info != nullptr) {
      if (a.info_store) a.info[b] = info_bits;   // this sweep's flags alone (the caller merges them if the sweep counts)
      else if (info_bits != 0) atomicOr(&a.info[b], info_bits);
    }
  }
}

template <int NX, int NU, int DB>
__global__ __launch_bounds__(256) void mpc_backward_rec_dma_kernel(const MpcBackArgs a) {
  mpc_backward_rec_dma_body<NX, NU, DB>(a, blockIdx.x);
}

// The sweep of box-DDP iteration i + 1 with the bookkeeping of iteration i (box_ddp_select_body: best-so-far update, stop
// tests) in ONE launch: the last n_sel workgroups do the bookkeeping (256 rows each) while the others sweep.  The sweep
// needs nothing from it but the `done` flag, and a sweep that runs although the loop has just stopped only fills gains
// nobody reads (the line search that follows is a separate launch and sees the flag); its flags go to a side buffer
// for that reason.
template <int NX, int NU, int DB>
__global__ __launch_bounds__(256) void mpc_backward_rec_dma_select_kernel(const MpcBackArgs a, const DdpSelectArgs s,
                                                                          const int n_sel, unsigned *sel_sync) {
  const int n_back = (int)gridDim.x - n_sel;
  if ((int)blockIdx.x >= n_back) {
    box_ddp_select_body<256, NX, NU>(s, (int)blockIdx.x - n_back, n_sel, sel_sync);
    return;
  }
  mpc_backward_rec_dma_body<NX, NU, DB>(a, blockIdx.x);
}

// the bookkeeping workgroups of the fused launch on their own (the last iteration has no next sweep to ride in)
template <int NX, int NU>
__global__ __launch_bounds__(256) void box_ddp_select_parts_kernel(const DdpSelectArgs s, unsigned *sel_sync) {
  box_ddp_select_body<256, NX, NU>(s, blockIdx.x, gridDim.x, sel_sync);
}

}  // namespace dmpc
