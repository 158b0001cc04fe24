// costate_wide_kernel.hpp - the co-state / outer-product kernel (costate_dma_kernel.hpp; DiffLqr.backward,
// lqr/differentiable_lqr.py:85-134, and MPCstep.backward, mpc/mpc_step.py:383-446) for problems with 17 to 31 elements of
// tau = [x; u] (nx <= 16): the shapes of lqr_wide_kernel.hpp, which used to take a wavefront per trajectory inside the
// (16,8) instance of costate_kernel (0.78 ms of the 1.0 ms gradient at (16,4), B = 4096, T = 50).
//
// A trajectory keeps ONE DPP row of 16 lanes, four trajectories per wavefront:
//   lane j        : tau_t[j], dtau_t[j] in register 0 and tau_t[16 + j], dtau_t[16 + j] in register 1
//   lane i < NX   : row i of C_t[:nx, :], c_t[i], r_t[i], column i of F_t[:, :nx], lambda[i], d_lambda[i]
//   lambda_t[i]   = c_t[i] + sum_j C_t[i][j] tau_t[j] + sum_k F_t[k][i] lambda_{t+1}[k]           (and d_lambda likewise)
//   dF_t row k (lane k < NX), dC_t rows i and 16 + i (lane i, two passes): broadcast-multiplies, staged through LDS and stored
//   as the contiguous run of HBM the wave's four trajectories make (whole 16-byte chunks, 64 lanes wide).
// Inputs by per-lane gather DMA into a two-slot LDS ring and ONE register set (lqr_wide_kernel.hpp's pipeline).  The kernel
// moves 6.1 KB per timestep and trajectory at (16,4) for ~600 instructions per FOUR trajectories: it is bound by the memory
// system, so the products are plain `Group<16>::bcast` code (no generated blocks).  Needs B >= 4, 16-byte aligned arrays;
// the tiled-cost sums (dC_sum) are not formed here.
#pragma once
#include "costate_dma_kernel.hpp"
#include "lqr_wide_kernel.hpp"      // padded_read()

namespace dmpc {

// PAD: the kernel is a CONTAINER for a smaller problem (a.nx_log <= NX, a.nu_log <= NU; lqr_wide_kernel<..., PAD>): the arrays
// come into container-sized regions of the slot as they are (all of C: its state rows are whole chunks only by chance), the
// reads place element i of tau at lane i (state) or NX + m (control), everything outside the problem is 0, and the output
// rows are staged at the problem's own strides.  Needs B % 4 == 0.
template <int NX, int NU, int DB, bool PAD = false>
struct CostateWideLayout {
  static constexpr int NS = NX + NU;
  // 16-byte chunks of one wave-step (four trajectories): [C state rows (PAD: all rows) | c | r | F | x | u | dx | du]
  static constexpr int nC = (PAD ? NS : NX) * NS, nc = NS, nF = NX * NS;
  static constexpr int CH_C = 0, CH_c = CH_C + nC, CH_r = CH_c + nc, CH_F = CH_r + nc, CH_x = CH_F + nF;
  static constexpr int CH_u = CH_x + NX, CH_dx = CH_u + NU, CH_du = CH_dx + NX, CH_END = CH_du + NU;
  static constexpr int OFF_C = CH_C * 4, OFF_c = CH_c * 4, OFF_r = CH_r * 4, OFF_F = CH_F * 4, OFF_x = CH_x * 4;
  static constexpr int OFF_u = CH_u * 4, OFF_dx = CH_dx * 4, OFF_du = CH_du * 4;   // in floats
  static constexpr int kDma = (CH_END + 63) / 64;
  static constexpr int SLOT = kDma * 256;           // floats per wave and timestep (whole 1 KB pieces)
  static constexpr int SCR = 4 * NS * NS;           // output staging, floats per wave: dF rows, then dC rows (one buffer)
  // per workgroup of `waves` wavefronts (+ a zero per wave for the PAD reads).  The padded (16,8) instance does not fit a CU's
  // LDS with four wavefronts per workgroup (173 KB): it runs with three - the kernel is bound by memory, not by SIMDs
  static constexpr size_t lds_bytes(int waves = 4) { return (size_t)waves * (DB * SLOT + SCR) * 4 + 64; }
  static_assert((NX * NS) % 4 == 0, "a trajectory's state rows of C are whole 16-byte chunks");
};

template <int NX, int NU, int DB, bool PAD = false, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void costate_wide_kernel(const CostateArgs a) {
  using Lay = CostateWideLayout<NX, NU, DB, PAD>;
  using G = Group<16>;
  constexpr int NS = NX + NU, N1 = NS - 16;      // elements of tau in the second register
  static_assert(NX <= 16 && NS > 16 && NS <= 31, "tau in two registers, the state rows in the first");
  static_assert((DB - 1) * Lay::kDma <= 63, "ring too deep for vmcnt");

  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;  // trajectory within the wave
  const int lane = lane64 & 15;
  int b0 = ((int)blockIdx.x * WPB + wave) * 4;
  if (b0 > a.B - 4) b0 = a.B - 4;  // the last wave overlaps its neighbour instead of running ragged (same results twice)
  b0 = __builtin_amdgcn_readfirstlane(b0);
  const int b = b0 + r;

  extern __shared__ float lds[];
  float *ring = lds + wave * (DB * Lay::SLOT);
  float *scr = lds + WPB * (DB * Lay::SLOT) + wave * Lay::SCR;
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));
  float *zo = lds + WPB * (DB * Lay::SLOT + Lay::SCR) + wave * 4;   // PAD: a zero for the padded reads (this wave's own)
  if constexpr (PAD) {
    if (lane64 == 0) zo[0] = 0.f;
  }

  // the problem's own dimensions, and where container element e of tau lies in them (-1: padding)
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int e) -> int { return e < NX ? (e < nx ? e : -1) : (e - NX < nu ? nx + (e - NX) : -1); };
  const int rc = a.r_cols ? a.r_cols : ns;   // row length of a.r
  const bool is_x = lane < nx;
  const bool is_t1 = lane < N1;              // this lane holds an element of tau in its second register
  const int lane_x = is_x ? lane : nx - 1;   // clamped: rows / columns re-read by the idle lanes, never used
  const float wa = 0.5f, wb = a.dC_mode == 0 ? 1.0f : 0.5f;

  // per-lane source pointers of the gather groups (costate_dma_kernel.hpp; 32-bit time strides: the launcher checks them)
  unsigned long long ptr[Lay::kDma];
  unsigned str[Lay::kDma], dyn = 0;
#pragma unroll
  for (int q = 0; q < Lay::kDma; ++q) {
    const int g = q * 64 + lane64;
    const int gg = g >= Lay::CH_END ? 0 : g;
    const char *base;
    size_t per;
    int g0;
    bool isF = false;
    size_t skip = 0;   // C, state rows only: trajectory k of the wave starts k * (ns - nx) * ns floats further on
    size_t run;          // bytes of the wave's run of this array in the slot (PAD: what lies behind fetches chunk 0 again)
    if (gg < Lay::CH_c) {
      base = (const char *)a.C; per = (size_t)ns * ns * 4; g0 = Lay::CH_C; run = 4 * per;
      if constexpr (!PAD) skip = (size_t)(gg / (NX * NS / 4)) * ((NS - NX) * NS * 4);
    }
    else if (gg < Lay::CH_r) { base = (const char *)a.c; per = (size_t)ns * 4; g0 = Lay::CH_c; run = 4 * per; }
    else if (gg < Lay::CH_F) { base = (const char *)a.r; per = (size_t)rc * 4; g0 = Lay::CH_r; run = 4 * per; }
    else if (gg < Lay::CH_x) { base = (const char *)a.F; per = (size_t)nx * ns * 4; g0 = Lay::CH_F; isF = true; run = 4 * per; }
    else if (gg < Lay::CH_u) { base = (const char *)a.x; per = (size_t)nx * 4; g0 = Lay::CH_x; run = 4 * per; }
    else if (gg < Lay::CH_dx) { base = (const char *)a.u; per = (size_t)nu * 4; g0 = Lay::CH_u; run = 4 * per; }
    else if (gg < Lay::CH_du) { base = (const char *)a.dx; per = (size_t)nx * 4; g0 = Lay::CH_dx; run = 4 * per; }
    else { base = (const char *)a.du; per = (size_t)nu * 4; g0 = Lay::CH_du; run = 4 * per; }
    // (r: rows of r_cols floats - and, PAD, every array shorter than its region: the chunks past the run repeat its chunk 0)
    const int gc = (size_t)(gg - g0) * 16 < run || (!PAD && base != (const char *)a.r) ? gg - g0 : 0;
    const int t0 = isF ? T - 2 : T - 1;
    ptr[q] = (unsigned long long)base + ((size_t)t0 * B + (size_t)b0) * per + (size_t)gc * 16 + skip -
             (unsigned long long)(q % 4) * 1024u;
    str[q] = (unsigned)(B * per);
    dyn |= isF ? (1u << q) : 0u;
  }
  int ti = T - 1;  // timesteps still to step back over
  auto issue_next = [&](int slot) __attribute__((always_inline)) {
    const unsigned dst = ring_addr + (unsigned)slot * (Lay::SLOT * 4);
    static_for<0, Lay::kDma>([&](auto q) {
      if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
      dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
    });
    if (ti > 0) {  // past t = 0 the same blocks are fetched again (never consumed): the count per step stays exact
      if (ti == T - 1) {   // F has no slice T-1: its lanes start at T-2 and sit out the first step
#pragma unroll
        for (int q = 0; q < Lay::kDma; ++q) ptr[q] -= ((dyn >> q) & 1u) ? 0ull : (unsigned long long)str[q];
      } else {
#pragma unroll
        for (int q = 0; q < Lay::kDma; ++q) ptr[q] -= (unsigned long long)str[q];
      }
      --ti;
    }
  };
  // per-lane LDS indices (floats, relative to a slot), computed once
  const int l0 = PAD ? logical(lane) : lane;                                    // element `lane` of tau: where it lies (-1: none)
  const int e1 = is_t1 ? 16 + lane : NS - 1;                                    // element 16 + lane (clamped)
  const int l1 = is_t1 ? (PAD ? logical(e1) : e1) : (PAD ? -1 : e1);
  auto tau_index = [&](int l, int off_x, int off_u) { return l < nx ? off_x + r * nx + (l >= 0 ? l : 0) : off_u + r * nu + (l - nx); };
  const int i_tau0 = tau_index(l0, Lay::OFF_x, Lay::OFF_u), i_dtau0 = tau_index(l0, Lay::OFF_dx, Lay::OFF_du);
  const int i_tau1 = tau_index(l1, Lay::OFF_x, Lay::OFF_u), i_dtau1 = tau_index(l1, Lay::OFF_dx, Lay::OFF_du);
  const int i_crow = Lay::OFF_C + (r * (PAD ? ns : NX) + lane_x) * ns;   // row lane_x of C_t (not PAD: state rows only in the slot)
  const int i_c = Lay::OFF_c + r * ns + lane_x, i_r = Lay::OFF_r + r * rc + lane_x;
  const int i_fcol = Lay::OFF_F + r * nx * ns + lane_x;     // column lane_x of F_t[:, :nx]

  float tau[2], dtau[2], ci, ri, Crow[NS], Fcol[NX];
  auto read_slot = [&](const float *slot) __attribute__((always_inline)) {
    tau[0] = slot[i_tau0];
    dtau[0] = slot[i_dtau0];
    tau[1] = slot[i_tau1];
    dtau[1] = slot[i_dtau1];
    ci = slot[i_c];
    ri = slot[i_r];
    if constexpr (PAD) {   // elements / rows outside the problem: 0
      tau[0] = l0 >= 0 ? tau[0] : 0.f;
      dtau[0] = l0 >= 0 ? dtau[0] : 0.f;
      tau[1] = l1 >= 0 ? tau[1] : 0.f;
      dtau[1] = l1 >= 0 ? dtau[1] : 0.f;
      ci = is_x ? ci : 0.f;
      ri = is_x ? ri : 0.f;
    }
    static_for<0, NS>([&](auto j) {
      if constexpr (PAD) {
        const int lj = logical(j.value);   // uniform
        Crow[j.value] = padded_read(slot + i_crow + (lj >= 0 ? lj : 0), lj >= 0 && is_x, zo);
      } else {
        Crow[j.value] = slot[i_crow + j.value];
      }
    });
    static_for<0, NX>([&](auto k) {
      if constexpr (PAD) {
        const bool row = k.value < nx;   // uniform
        Fcol[k.value] = padded_read(slot + i_fcol + (row ? k.value : 0) * ns, row && is_x, zo);
      } else {
        Fcol[k.value] = slot[i_fcol + k.value * NS];
      }
    });
  };
  // the staged rows -> the wave's contiguous run of HBM: n_chunks 16-byte chunks, a chunk per lane and instruction
  auto store_run = [&](float *dst, int n_chunks) __attribute__((always_inline)) {
    if constexpr (PAD) {
      for (int ch = lane64; ch < n_chunks; ch += 64) reinterpret_cast<float4 *>(dst)[ch] = reinterpret_cast<const float4 *>(scr)[ch];
    } else {
      (void)n_chunks;
    }
  };
  // stage row[0..NS) of the problem's row `li` (trajectory r, rows of `ns` floats) at the problem's own strides
  auto stage_row = [&](const float (&row)[NS], int li, bool on) __attribute__((always_inline)) {
    static_for<0, NS>([&](auto j) {
      if constexpr (PAD) {
        const int lj = logical(j.value);   // uniform
        if (lj >= 0 && on) scr[(li) * ns + lj] = row[j.value];
      } else {
        if (on) scr[li * NS + j.value] = row[j.value];
      }
    });
  };
  // row[j] = tau[j] * ca + dtau[j] * cb, j < NS (the elements of tau broadcast from their lanes)
  auto outer2 = [&](float (&row)[NS], const float ca, const float cb) __attribute__((always_inline)) {
    static_for<0, NS>([&](auto j) {
      constexpr int h = j.value / 16, l = j.value % 16;
      row[j.value] = fmaf(G::template bcast<l>(dtau[h]), cb, G::template bcast<l>(tau[h]) * ca);
    });
  };

  float lam = 0.f, dlam = 0.f;  // lambda_{t+1}[lane], d_lambda_{t+1}[lane]  (lanes < NX)
  auto step = [&](int t) __attribute__((always_inline)) {
    const size_t tb = (size_t)t * B + b;
    if (t < T - 1) {                                                          // differentiable_lqr.py:130-133
      if (a.dF != nullptr) {   // dF_t[k][:] = d_lambda_{t+1}[k] tau + lambda_{t+1}[k] dtau
        float row[NS];
        outer2(row, a.out_sign * dlam, a.out_sign * lam);
        stage_row(row, r * nx + lane, is_x);
        if constexpr (PAD) store_run(a.dF + ((size_t)t * B + b0) * (nx * ns), nx * ns);
        else store_chunks<NX * NS>(scr, a.dF + ((size_t)t * B + b0) * (NX * NS), lane64);
      }
      if (a.df != nullptr && a.df_shift == 1 && is_x) a.df[tb * nx + lane] = a.out_sign * dlam;
    }
    if (a.dC != nullptr) {                                                    // :128-129: rows `lane` and 16 + lane
      float row[NS];
      outer2(row, a.out_sign * wa * dtau[0], a.out_sign * wb * tau[0]);
      stage_row(row, r * ns + (l0 >= 0 ? l0 : 0), l0 >= 0);
      outer2(row, a.out_sign * wa * dtau[1], a.out_sign * wb * tau[1]);
      stage_row(row, r * ns + (l1 >= 0 ? l1 : 0), is_t1 && l1 >= 0);
      if constexpr (PAD) store_run(a.dC + ((size_t)t * B + b0) * (ns * ns), ns * ns);
      else store_chunks<NS * NS>(scr, a.dC + ((size_t)t * B + b0) * (NS * NS), lane64);
    }
    if (a.dc != nullptr) {
      if (l0 >= 0) a.dc[tb * ns + l0] = a.out_sign * dtau[0];
      if (is_t1 && l1 >= 0) a.dc[tb * ns + l1] = a.out_sign * dtau[1];
    }
    float nl = ci, ndl = a.r_sign * ri;                                       // :92,102 / :115,124
    float nl2 = 0.f, ndl2 = 0.f;                                              // (two chains each)
    static_for<0, NS>([&](auto j) {
      constexpr int h = j.value / 16, l = j.value % 16;
      const float tj = G::template bcast<l>(tau[h]), dj = G::template bcast<l>(dtau[h]);
      if constexpr (j.value % 2 == 0) {
        nl = fmaf(tj, Crow[j.value], nl);
        ndl = fmaf(dj, Crow[j.value], ndl);
      } else {
        nl2 = fmaf(tj, Crow[j.value], nl2);
        ndl2 = fmaf(dj, Crow[j.value], ndl2);
      }
    });
    if (t < T - 1) {
      static_for<0, NX>([&](auto k) {
        const float lk = G::template bcast<k.value>(lam), dk = G::template bcast<k.value>(dlam);
        if constexpr (k.value % 2 == 0) {
          nl = fmaf(lk, Fcol[k.value], nl);
          ndl = fmaf(dk, Fcol[k.value], ndl);
        } else {
          nl2 = fmaf(lk, Fcol[k.value], nl2);
          ndl2 = fmaf(dk, Fcol[k.value], ndl2);
        }
      });
    }
    lam = nl + nl2;
    dlam = ndl + ndl2;
    if (a.df != nullptr && a.df_shift == 0 && t < T - 1 && is_x) a.df[tb * nx + lane] = a.out_sign * dlam;
  };

  // ONE register set (lqr_wide_kernel.hpp): the slot of step t is waited for and read, then - once the reads are in -
  // refilled with step t - DB, and the step runs while DB - 1 fetches are in flight.  The stores of a step are younger than
  // its refill: waiting for all but kDma operations is exact for the first step and conservative afterwards.
  static_for<0, DB>([&](auto j) { issue_next(j.value); });
  for (int t0 = T - 1; t0 >= 0; t0 -= DB) {
    static_for<0, DB>([&](auto j) {
      const int t = t0 - j.value;
      if (t >= 0) {
        wait_vmcnt<(DB - 1) * Lay::kDma>();
        read_slot(ring + j.value * Lay::SLOT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are in before it is refilled
        issue_next(j.value);
        step(t);
      }
    });
  }
  wait_vmcnt<0>();
  if (a.dx0 != nullptr && is_x) a.dx0[(size_t)b * nx + lane] = a.out_sign * dlam;
}

}  // namespace dmpc
