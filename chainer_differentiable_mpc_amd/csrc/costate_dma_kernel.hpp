// costate_dma_kernel.hpp - the co-state / outer-product kernel of costate_kernels.hpp with its per-timestep inputs
// staged through LDS by LDS-DMA (the scheme of lqr_dma_kernel.hpp): a wavefront owns four consecutive trajectories,
// so C, c, r, F, x, u, dx, du of one timestep are eight contiguous runs of HBM; per-lane gather DMA (each lane copies the
// 16-byte chunk at its own address, as in the generated LQR stream) moves all of them with a handful of instructions
// into a ring of DB slots, DB - 1 timesteps ahead of the arithmetic.  The register-prefetch version keeps two timesteps
// in flight at best (hipcc drains vmcnt at its loop header) and the kernel is bandwidth/latency bound: ~150
// instructions per timestep against 1.6 KB of traffic.  Arithmetic and stores are the same code.
// Follows DiffLqr.backward, lqr/differentiable_lqr.py:85-134, and MPCstep.backward, mpc/mpc_step.py:383-446.
#pragma once
#include "costate_kernels.hpp"
#include "dma_gather.hpp"
#include "lqr_dma_kernel.hpp"

namespace dmpc {

// NCH 16-byte chunks staged in LDS (written row-wise by the lanes that own the rows) -> one contiguous run of HBM,
// a chunk per lane and instruction.  The LDS queue of a wavefront is in order, so the reads see the writes above.
template <int NCH>
__device__ __forceinline__ void store_chunks(const float *scr, float *dst, int lane64) {
#pragma unroll
  for (int q = 0; q < (NCH + 63) / 64; ++q) {
    const int ch = q * 64 + lane64;
    if (ch < NCH) reinterpret_cast<float4 *>(dst)[ch] = reinterpret_cast<const float4 *>(scr)[ch];
  }
}

template <int NX, int NU, int DB>
struct CostateDmaLayout {
  static constexpr int NS = NX + NU;
  // 16-byte chunks of one wave-step (four trajectories): [C | c | r | F | x | u | dx | du].  Only the state rows of C
  // enter the co-state recursions (differentiable_lqr.py:92,102,115,124: C[:nx, :]): where a trajectory's nx * ns
  // floats are whole chunks, the control rows are not fetched (a fifth of C: 16 MB of 331 at the headline size)
  static constexpr bool kStateRowsOnly = (NX * NS) % 4 == 0;
  static constexpr int kCRows = kStateRowsOnly ? NX : NS;       // rows of C_t per trajectory in the slot
  static constexpr int nC = kCRows * NS, nc = NS, nF = NX * NS, nx_ = NX, nu_ = NU;
  static constexpr int CH_C = 0, CH_c = CH_C + nC, CH_r = CH_c + nc, CH_F = CH_r + nc, CH_x = CH_F + nF;
  static constexpr int CH_u = CH_x + nx_, CH_dx = CH_u + nu_, CH_du = CH_dx + nx_, CH_END = CH_du + nu_;
  static constexpr int OFF_C = CH_C * 4, OFF_c = CH_c * 4, OFF_r = CH_r * 4, OFF_F = CH_F * 4, OFF_x = CH_x * 4;
  static constexpr int OFF_u = CH_u * 4, OFF_dx = CH_dx * 4, OFF_du = CH_du * 4;   // in floats
  static constexpr int kDma = (CH_END + 63) / 64;   // gather DMAs per step; padding lanes repeat chunk 0 of C
  static constexpr int SLOT = kDma * 256;           // floats per wave and timestep (whole 1 KB pieces)
  // output staging: the wave's dC (dF) rows of one timestep are one contiguous run of HBM - they go through LDS so that
  // the stores are whole 16-byte chunks, 64 lanes wide, instead of 8-byte pieces 40 bytes apart
  static constexpr int SCR = 4 * NS * NS + 4 * NX * NS;   // floats per wave
  static constexpr size_t lds_bytes() { return (size_t)4 * (DB * SLOT + SCR) * 4; }
};

template <int NX, int NU, int DB>
__global__ __launch_bounds__(256) void costate_dma_kernel(const CostateArgs a) {
  using Lay = CostateDmaLayout<NX, NU, DB>;
  constexpr int NS = NX + NU, L = 16;
  static_assert(NS <= L, "tau must fit the lane group");
  static_assert((DB - 1) * Lay::kDma <= 63, "ring too deep for vmcnt");
  static_assert(DB % 2 == 0 && DB >= 2, "two alternating register sets");
  static_assert(Lay::kDma <= 8, "gather groups");
  using Blk = RiccatiBlocks<NX, NU, L>;

  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;  // trajectory within the wave
  const int lane = lane64 & 15;
  int b0 = ((int)blockIdx.x * 4 + wave) * 4;
  if (b0 > a.B - 4) b0 = a.B - 4;  // last wave overlaps its neighbour instead of running ragged (same results twice)
  b0 = __builtin_amdgcn_readfirstlane(b0);
  const int b = b0 + r;

  extern __shared__ float lds[];
  float *ring = lds + wave * (DB * Lay::SLOT);
  float *scrC = lds + 4 * (DB * Lay::SLOT) + wave * Lay::SCR, *scrF = scrC + 4 * NS * NS;
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));

  const int rc = a.r_cols ? a.r_cols : NS;   // row length of a.r
  const bool is_x = lane < NX;
  const bool is_tau = lane < NS;
  const int lane_x = is_x ? lane : NX - 1;  // clamped: rows/columns re-read by the idle lanes, never used
  const int lane_t = is_tau ? lane : NS - 1;
  const float wa = 0.5f, wb = a.dC_mode == 0 ? 1.0f : 0.5f;

  // per-lane source pointers of the gather groups: chunk g = 64 q + lane64 of the slot.  Every array steps back by
  // one timestep per fetch; F has no slice T-1, so its lanes start at T-2 and sit out the first step.
  unsigned long long ptr[Lay::kDma], str[Lay::kDma], str1[Lay::kDma];
#pragma unroll
  for (int q = 0; q < Lay::kDma; ++q) {
    const int g = q * 64 + lane64;
    const bool pad = g >= Lay::CH_END;
    const int gg = pad ? 0 : g;
    const char *base;
    size_t per;
    int g0;
    bool isF = false;
    size_t skip = 0;   // C, state rows only: trajectory k of the wave starts k * (ns - nx) * ns floats further on
    if (gg < Lay::CH_c) {
      base = (const char *)a.C; per = (size_t)NS * NS * 4; g0 = Lay::CH_C;
      if constexpr (Lay::kStateRowsOnly) skip = (size_t)(gg / (NX * NS / 4)) * ((NS - NX) * NS * 4);
    }
    else if (gg < Lay::CH_r) { base = (const char *)a.c; per = (size_t)NS * 4; g0 = Lay::CH_c; }
    else if (gg < Lay::CH_F) {   // r: rows of r_cols floats - the chunks past the wave's 4 rows repeat its chunk 0
      base = (const char *)a.r; per = (size_t)rc * 4; g0 = gg - Lay::CH_r < rc ? Lay::CH_r : gg;
    }
    else if (gg < Lay::CH_x) { base = (const char *)a.F; per = (size_t)NX * NS * 4; g0 = Lay::CH_F; isF = true; }
    else if (gg < Lay::CH_u) { base = (const char *)a.x; per = (size_t)NX * 4; g0 = Lay::CH_x; }
    else if (gg < Lay::CH_dx) { base = (const char *)a.u; per = (size_t)NU * 4; g0 = Lay::CH_u; }
    else if (gg < Lay::CH_du) { base = (const char *)a.dx; per = (size_t)NX * 4; g0 = Lay::CH_dx; }
    else { base = (const char *)a.du; per = (size_t)NU * 4; g0 = Lay::CH_du; }
    const int t0 = isF ? T - 2 : T - 1;
    ptr[q] = (unsigned long long)base + ((size_t)t0 * B + (size_t)b0) * per + (size_t)(gg - g0) * 16 + skip - (unsigned long long)(q % 4) * 1024u;
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
  // per-lane LDS indices (floats, relative to a slot), computed once
  const int i_tau = lane_t < NX ? Lay::OFF_x + r * NX + lane_t : Lay::OFF_u + r * NU + (lane_t - NX);
  const int i_dtau = lane_t < NX ? Lay::OFF_dx + r * NX + lane_t : Lay::OFF_du + r * NU + (lane_t - NX);
  const int i_crow = Lay::OFF_C + (r * Lay::kCRows + lane_x) * NS;   // row lane_x of C_t
  const int i_c = Lay::OFF_c + r * NS + lane_x, i_r = Lay::OFF_r + r * rc + lane_x;
  const int i_fcol = Lay::OFF_F + r * NX * NS + lane_x;     // column lane_x of F_t[:, :NX]
  struct Slot {
    float tau, dtau, ci, ri;
    float Crow[NS], Fcol[NX];
  };
  auto read_slot = [&](const float *slot, Slot &s) {
    s.tau = slot[i_tau];
    s.dtau = slot[i_dtau];
    s.ci = slot[i_c];
    s.ri = slot[i_r];
#pragma unroll
    for (int j = 0; j < NS; ++j) s.Crow[j] = slot[i_crow + j];
#pragma unroll
    for (int k = 0; k < NX; ++k) s.Fcol[k] = slot[i_fcol + k * NS];
  };

  float lam = 0.f, dlam = 0.f;  // lambda_{t+1}[lane], d_lambda_{t+1}[lane]  (lanes < NX)
  const bool summed = a.dC_sum != nullptr;   // (uniform) the tiled-cost reduction: row `lane` of sum_t dC_t, sum_t dc_t[lane]
  float accC[NS], accc = 0.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) accC[j] = 0.f;
  auto step = [&](int t, const Slot &s) {  // the step of costate_kernel
    const size_t tb = (size_t)t * B + b;
    const float tau = s.tau, dtau = s.dtau;
    if (t < T - 1) {                                                          // differentiable_lqr.py:130-133
      if (a.dF != nullptr) {
        float row[NS];
        Blk::outer2(row, tau, dtau, a.out_sign * dlam, a.out_sign * lam);
        if (is_x) {
#pragma unroll
          for (int j = 0; j < NS; ++j) scrF[(r * NX + lane) * NS + j] = row[j];
        }
        store_chunks<NX * NS>(scrF, a.dF + ((size_t)t * B + b0) * (NX * NS), lane64);
      }
      if (a.df != nullptr && a.df_shift == 1 && is_x) a.df[tb * NX + lane] = a.out_sign * dlam;
    }
    if (a.dC != nullptr || summed) {                                          // :128-129
      float row[NS];
      Blk::outer2(row, tau, dtau, a.out_sign * wa * dtau, a.out_sign * wb * tau);
      if (summed) {
#pragma unroll
        for (int j = 0; j < NS; ++j) accC[j] += row[j];
        accc += a.out_sign * dtau;
      }
      if (a.dC != nullptr) {
        if (is_tau) {
#pragma unroll
          for (int j = 0; j < NS; ++j) scrC[(r * NS + lane) * NS + j] = row[j];
        }
        store_chunks<NS * NS>(scrC, a.dC + ((size_t)t * B + b0) * (NS * NS), lane64);
      }
    }
    if (a.dc != nullptr && is_tau) a.dc[tb * NS + lane] = a.out_sign * dtau;
    float nl = s.ci, ndl = a.r_sign * s.ri;                                   // :92,102 / :115,124
    Blk::dots2_ns(nl, ndl, s.Crow, tau, dtau);
    if (t < T - 1) Blk::dots2_nx(nl, ndl, s.Fcol, lam, dlam);
    lam = nl;
    dlam = ndl;
    if (a.df != nullptr && a.df_shift == 0 && t < T - 1 && is_x) a.df[tb * NX + lane] = a.out_sign * dlam;
  };

  // Software pipeline of lqr_dma_kernel: at step t the DMA for step t - DB goes into the slot whose contents went to
  // registers one step ago, the slot of step t - 1 is waited for and read into the other register set, then step t
  // is computed from its own set.  The stores issued in between only make the counted wait more conservative.
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
  if (a.dx0 != nullptr && is_x) a.dx0[(size_t)b * NX + lane] = a.out_sign * dlam;
  if (summed) {
    // the wave's four trajectories (the last wave of a ragged grid repeats trajectories of its neighbour: those rows are
    // left out) -> lanes of row 0 -> one atomic per element
    const bool mine = ((int)blockIdx.x * 4 + wave) * 4 + r == b;   // false for a repeated trajectory
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      float v = mine ? accC[j] : 0.f;
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane64 < NS) atomicAdd(&a.dC_sum[lane64 * NS + j], v);
    }
    float v = mine ? accc : 0.f;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane64 < NS && a.dc_sum != nullptr) atomicAdd(&a.dc_sum[lane64], v);
  }
}

}  // namespace dmpc
