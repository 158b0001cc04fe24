// lqr_wide_kernel.hpp - the fused LQR solve (LqrRecursion.solve_recursion, lqr/lqr_recursion.py:69-209) for problems whose
// augmented matrices have 17 to 32 columns (nx <= 16, 16 <= nx + nu <= 31), on the 16-lane row layout with TWO registers per
// matrix row: register [i][h] of lane j holds M[i][16 h + j].  (The template also takes shapes of the plain 16-lane layout
// with four-row tiles - NR = 1, one register per row; (8,4) that way: 94 us against lqr_kernel's 84, not instantiated.)
//
// A trajectory keeps ONE DPP row of 16 lanes - four trajectories per wavefront - where these shapes used to take a whole
// wavefront each on the matrix-core kernel of the (32,8) class (inside its (16,8) instance, lqr_wave_mfma.hpp: ~2,000
// instructions per trajectory and step).  The products of the A^T B shape - G = V^T F~, Q~ += G^T F~ (the reference's own
// association (F^T V) F, lqr_recursion.py:89,96) and K~^T R of the value update - are outer products of registers and run on
// the matrix cores: `v_mfma_f32_4x4x1_16b_f32` with cbsz:2 takes four lanes (rows 4I..4I+3) of a trajectory's A register
// against all 16 lanes of its B register, one instruction per 4 x 16 block and inner index (the (8,2) stream's scheme,
// gen_lqr_asm.py).  The A B shapes (Qxu K~, Quu K~, v^T F~) stay chains of `v_fmac_f32_dpp ... row_newbcast`
// (dpp_blocks_wide_gen.hpp).  ~1,100 instructions per step and FOUR trajectories.  Everything else is lqr_dma_kernel.hpp's scheme: the wave's four
// consecutive trajectories make every input array one contiguous run per timestep, fetched into an LDS ring by per-lane
// gather DMA (16 bytes per lane), the registers filled by ds_read with per-lane indices, counted vmcnt waits, no barrier.
// The gain rows [K_m | 0 | k_m] go to the caller's workspace ([T,B,NU,NS+1] floats, the KHBM form) and come back to the
// rollout through its ring with F and f: at these sizes they do not fit next to the rings.
//
// Same arithmetic as lqr_kernel / lqr_dma_kernel with the row Gauss-Jordan of riccati_blocks.hpp (LAPACK's pivot choice).
// Needs B >= 4, 16-byte aligned arrays, the workspace.  Forms (template flags, see below): PAD - a container for every
// smaller shape without a 16-lane kernel; MASKED - LQR_active; MPC - MPCstep.backward_rec with the box QP inside (no rollout).
// The separate sweeps (LqrRecursion.backward() / forward()) of these shapes stay on the containers; the layout's co-state
// sweep is costate_wide_kernel.hpp, its line search mpc_wide_forward_kernel.hpp.
#pragma once
#include "colwise.hpp"
#include "dma_gather.hpp"
#include "dpp_blocks_wide_gen.hpp"
#include "lqr_dma_kernel.hpp"
#include "lqr_kernels.hpp"
#include "pnqp_device.hpp"

namespace dmpc {

typedef float f4w __attribute__((ext_vector_type(4)));
// A container's padded read: the element if it belongs to the problem, else a constant kept in LDS (0, or 1 for the diagonal
// of an unused control) - the ADDRESS is selected, not the value.  (Selecting the value lets hipcc sink the load into the
// select and wrap every padded element in its own branch: 5,600 instead of 3,100 instructions in
// lqr_wide_kernel<16, 4, ..., PAD>; pinning the value with an empty asm puts a wait behind every load.)
__device__ __forceinline__ float padded_read(const float *element, bool valid, const float *constant) {
  return *(valid ? element : constant);
}
// D[c] (lane l) += A[lane 16 (l / 16) + 4 I + c] * B[lane l], c < 4: rows 4I..4I+3 of an outer product, per 16-lane trajectory
template <int I>
__device__ __forceinline__ f4w mfma_rows(float a, float b, f4w c) {
  static_assert(I >= 0 && I < 4, "block of four lanes inside a 16-lane row");
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, I, 0);
}

template <int NX, int NU, int DB, int DF>
struct LqrWideLayout {
  static constexpr int NS = NX + NU, KROW = NS + 1;
  static constexpr int C_FL = 4 * NS * NS, c_FL = 4 * NS, F_FL = 4 * NX * NS, f_FL = 4 * NX, K_FL = 4 * NU * KROW;
  static constexpr int OFF_C = 0, OFF_c = OFF_C + C_FL, OFF_F = OFF_c + c_FL, OFF_f = OFF_F + F_FL;
  static constexpr int CH_B = (OFF_f + f_FL) / 4;          // 16-byte chunks of a backward slot [C | c | F | f]
  static constexpr int CH_F = (F_FL + f_FL + K_FL) / 4;    // ... of a forward slot [F | f | gain rows]
  static constexpr int kDmaB = (CH_B + 63) / 64, kDmaF = (CH_F + 63) / 64;
  static constexpr int SLOT_B = kDmaB * 256, SLOT_F = kDmaF * 256;   // floats: whole kilobytes
  static constexpr int RING_FL = (DB * SLOT_B > DF * SLOT_F) ? DB * SLOT_B : DF * SLOT_F;   // per wave
  static constexpr size_t lds_bytes() { return (size_t)4 * RING_FL * 4 + 64; }   // + [0, 1] per wave (the containers' padding)
};

// Gauss-Jordan on the rows of [Qux | Quu | qu] where they lie (gauss_jordan_rows of riccati_blocks.hpp, two registers per
// row): the multiplier of row i at pivot k is column NX + k of it - lane (NX + k) % 16 of register (NX + k) / 16.
template <int NX, int NU, int NR>
__device__ __forceinline__ bool gauss_jordan_rows_wide(float (&Kr)[NU][NR]) {
  using G = Group<16>;
  bool singular = false;
  static_for<0, NU>([&](auto kc) {
    constexpr int k = kc.value, kb = (NX + k) / 16, kl = (NX + k) % 16;
    float p = G::template bcast<kl>(Kr[k][kb]);
    float li[NU];
    float mx = 0.f;
    static_for<k + 1, NU>([&](auto ic) {
      li[ic.value] = G::template bcast<kl>(Kr[ic.value][kb]);
      mx = fmaxf(mx, fabsf(li[ic.value]));
    });
    if (k + 1 < NU && any_lane(mx > fabsf(p))) {   // rare: LAPACK's row interchange (the first largest entry)
      float best = fabsf(p);
      int pr = k;
      static_for<k + 1, NU>([&](auto ic) {
        const bool gt = fabsf(li[ic.value]) > best;
        best = gt ? fabsf(li[ic.value]) : best;
        pr = gt ? ic.value : pr;
      });
      static_for<k + 1, NU>([&](auto ic) {
        const bool sw = pr == ic.value;
#pragma unroll
        for (int h = 0; h < NR; ++h) {
          const float rk = Kr[k][h], ri = Kr[ic.value][h];
          Kr[k][h] = sw ? ri : rk;
          Kr[ic.value][h] = sw ? rk : ri;
        }
        const float pl = li[ic.value];
        li[ic.value] = sw ? p : pl;
        p = sw ? pl : p;
      });
    }
    singular = singular || (p == 0.f);
    const float r = fast_rcp(p);
#pragma unroll
    for (int h = 0; h < NR; ++h) Kr[k][h] *= r;
    static_for<0, NU>([&](auto ic) {
      constexpr int i = ic.value;
      if constexpr (i != k) {
        float l;
        if constexpr (i > k) l = li[i];
        else l = G::template bcast<kl>(Kr[i][kb]);
#pragma unroll
        for (int h = 0; h < NR; ++h) Kr[i][h] = fmaf(-l, Kr[k][h], Kr[i][h]);
      }
    });
  });
  return singular;
}

// PAD: the kernel is a CONTAINER for a smaller problem (a.nx_log <= NX states, a.nu_log <= NU controls; what lqr_kernel<..., PAD>
// is to the 16-lane shapes): the arrays in HBM keep the problem's own strides and come into the slot as they are; the reads
// place state i at row / column i and control m at NX + m, everything else of [C|c] and [F|f] is 0 and the unused controls
// get a unit diagonal in Quu (their gain rows come out exactly 0, LAPACK's pivot choice is unchanged).  Needs B % 4 == 0.
// MASKED: LQR_active (mpc/active_constrained_lqr.py:110-137): a.mask [T,B,nu] flags the clamped controls of every step - they
// leave the gain solve (their rows: 1e-8 on the diagonal, 0 elsewhere; their columns of the free rows: 0), so their gain rows
// come out exactly 0 and the rollout needs no change; the value update keeps the unmasked blocks (:143-145).
// MPC: MPCstep.backward_rec (mpc/mpc_step.py:70-173) - the sweep alone, with the feed-forward term k_t the solution of the box
// QP on (Quu, qu), lower - u <= k <= upper - u (projected Newton, mpc/pnqp.py:37-201, per-trajectory termination, warm start
// from the later step: pnqp_solve_rows in every lane on broadcast Quu, qu - mpc_kernels.hpp's scheme), K_t from the QP's
// own last factorisation with the clamped rows zeroed (:147-157), the value update from the unmasked blocks (:165-166);
// with a.mpc_states the re-centring c_hat = C [x_t; u_t] + c (need_expand, :305-317) happens on the slot's rows.  Gains to
// a.Ks / a.ks, sum_t (1 + i_t) to a.mpc_n_qp_total; no rollout.  (Before: the runtime-dimension kernel, 2.0 ms at (12,4).)
template <int NX, int NU, int DB, int DF, bool PAD = false, bool MASKED = false, bool MPC = false>
__global__ __launch_bounds__(256) void lqr_wide_kernel(const LqrArgs a) {
  using Lay = LqrWideLayout<NX, NU, DB, DF>;
  using Blk = RiccatiBlocksWide<NX, NU>;
  using G = Group<16>;
  constexpr int NS = NX + NU, KROW = NS + 1;
  constexpr int NR = NS + 1 <= 16 ? 1 : 2;    // registers per matrix row
  constexpr int AB = NS / 16;                 // the affine column lies in register AB (lane NS % 16)
  constexpr int NT = NS / 4;                  // tiles of four rows of Q~
  static_assert(Blk::kAvailable, "no generated blocks for this shape (gen_dpp_blocks_wide.py SHAPES)");
  static_assert(NX <= 16 && NS <= 31 && AB < NR, "at most 32 augmented columns, state columns in the first register");
  static_assert(NX % 4 == 0 && NU % 4 == 0, "tiles of four rows");
  static_assert((DB - 1) * Lay::kDmaB <= 63 && (DF - 1) * Lay::kDmaF <= 63, "ring too deep for vmcnt");
  static_assert(DF % 2 == 0, "two alternating register sets in the rollout");
  static_assert(!MPC || (!MASKED && NR == 2), "the MPC sweep: two-register shapes");

  if constexpr (MPC) {
    if (a.mpc_done != nullptr && *a.mpc_done != 0) return;  // uniform: the iLQR loop has stopped
  }
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;   // trajectory within the wave
  const int lane = lane64 & 15;
  int b0 = ((int)blockIdx.x * 4 + wave) * 4;  // first trajectory of this wave
  if (b0 > a.B - 4) b0 = a.B - 4;             // the last wave overlaps its neighbour instead of running ragged
  b0 = __builtin_amdgcn_readfirstlane(b0);
  const int b = b0 + r;

  extern __shared__ float lds[];
  float *ring = lds + wave * Lay::RING_FL;
  float *kw = a.wsK + (size_t)b * NU * KROW;    // + t * B * NU * KROW: this trajectory's gain rows in the workspace
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));
  float *zo = lds + 4 * Lay::RING_FL + wave * 4;   // PAD: zo[0] = 0, zo[1] = 1 (written below, read by this wave alone)
  if constexpr (PAD) {
    if (lane64 < 2) zo[lane64] = (float)lane64;
  }

  int info_bits = 0;
  // the problem's own dimensions, and where container row / column c lies in its arrays (-1: padding)
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int c) -> int { return c < NX ? (c < nx ? c : -1) : (c - NX < nu ? nx + (c - NX) : -1); };
  // register h of this lane holds column 16 h + lane: a column of C / F (lc[h]: where it lies in the arrays, -1: none), the
  // affine column, or nothing
  int lc[NR];
  bool aff[NR];
#pragma unroll
  for (int h = 0; h < NR; ++h) {
    const int col = 16 * h + lane;
    aff[h] = col == NS;
    lc[h] = col < NS ? (PAD ? logical(col) : col) : -1;
  }
  const bool affl = aff[AB];                    // this lane holds the affine column (in its register AB)

  // ------------------------------------------------------------------ backward Riccati sweep
  {
    // gather pointers (lqr_dma_kernel.hpp): chunk 64 q + lane64 of the slot [C | c | F | f] from its own address; every array
    // steps BACK one timestep per fetch (32-bit strides: the launcher checks B * ns^2 * 4 < 2^31); F and f have no slice
    // T-1: their lanes (`dyn`) start at T-2 and sit out the first step
    unsigned long long ptr[Lay::kDmaB];
    unsigned str[Lay::kDmaB], dyn = 0;
#pragma unroll
    for (int q = 0; q < Lay::kDmaB; ++q) {
      const int g = q * 64 + lane64;
      const int gg = g < Lay::CH_B ? g : 0;
      const char *base;
      size_t per;          // bytes per trajectory and timestep (PAD: the problem's own)
      int g0;
      bool d = false;
      if (gg < Lay::OFF_c / 4) { base = (const char *)a.C; per = (size_t)ns * ns * 4; g0 = 0; }
      else if (gg < Lay::OFF_F / 4) { base = (const char *)a.c; per = (size_t)ns * 4; g0 = Lay::OFF_c / 4; }
      else if (gg < Lay::OFF_f / 4) { base = (const char *)(T > 1 ? a.F : a.C); per = (size_t)nx * ns * 4; g0 = Lay::OFF_F / 4; d = true; }
      else { base = (const char *)(has_f ? a.f : a.c); per = (size_t)nx * 4; g0 = Lay::OFF_f / 4; d = true; }
      const int t0 = d ? (T > 1 ? T - 2 : 0) : T - 1;
      // (PAD: every array sits at the start of its container-sized region; the chunks behind its end fetch its chunk 0 again)
      const int gc = (!PAD || (size_t)(gg - g0) * 16 < 4 * per) ? gg - g0 : 0;
      ptr[q] = (unsigned long long)base + ((size_t)t0 * B + (size_t)b0) * per + (size_t)gc * 16 -
               (unsigned long long)(q % 4) * 1024u;
      str[q] = (unsigned)(B * per);
      dyn |= d ? (1u << q) : 0u;
    }
    int ti = T - 1;
    auto issue_next = [&](int slot) __attribute__((always_inline)) {
      const unsigned dst = ring_addr + (unsigned)slot * (Lay::SLOT_B * 4);
      static_for<0, Lay::kDmaB>([&](auto q) {
        if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
        dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
      });
      if (ti > 0) {
        if (ti == T - 1) {
#pragma unroll
          for (int q = 0; q < Lay::kDmaB; ++q) ptr[q] -= ((dyn >> q) & 1u) ? 0ull : (unsigned long long)str[q];
        } else {
#pragma unroll
          for (int q = 0; q < Lay::kDmaB; ++q) ptr[q] -= (unsigned long long)str[q];
        }
        --ti;
      }
    };
    // per-lane LDS indices (floats, relative to a slot): the first register reads column `lane`, the second column 16 + lane
    // of C / F - or c / f in the affine column's lane (row stride 1 there), or anything finite past it (never broadcast)
    int qb[NR], fb[NR], sh[NR];
    bool ok[NR];                 // PAD: this lane's column belongs to the problem
#pragma unroll
    for (int h = 0; h < NR; ++h) {
      qb[h] = aff[h] ? Lay::OFF_c + r * ns : Lay::OFF_C + r * ns * ns + (lc[h] >= 0 ? lc[h] : 0);
      fb[h] = aff[h] ? Lay::OFF_f + r * nx : Lay::OFF_F + r * nx * ns + (lc[h] >= 0 ? lc[h] : 0);
      sh[h] = aff[h] ? 1 : ns;   // c / f: consecutive entries; a column of C / F: a row apart
      ok[h] = aff[h] || lc[h] >= 0;
    }
    auto read_slot = [&](const float *slot, f4w (&Qn)[NT][NR], float (&Fn)[NX][NR]) __attribute__((always_inline)) {
      static_for<0, NT>([&](auto I) {
        static_for<0, 4>([&](auto cc) {
          constexpr int i = 4 * I.value + cc.value;
          static_for<0, NR>([&](auto h) {
            if constexpr (PAD) {
              const int li = logical(i);   // uniform
              // outside the problem: 0, and 1 on the diagonal of the unused controls (column i: lane i % 16 of register i / 16)
              const bool diag = li < 0 && i >= NX && i / 16 == h.value && lane == i % 16;
              Qn[I.value][h.value][cc.value] = padded_read(slot + qb[h.value] + (li >= 0 ? li : 0) * sh[h.value],
                                                           li >= 0 && ok[h.value], zo + (diag ? 1 : 0));
            } else {
              Qn[I.value][h.value][cc.value] = slot[qb[h.value] + i * sh[h.value]];
            }
          });
        });
      });
      static_for<0, NX>([&](auto k) {
        static_for<0, NR>([&](auto h) {
          if constexpr (PAD) {
            const bool row = k.value < nx;   // uniform
            Fn[k.value][h.value] = padded_read(slot + fb[h.value] + (row ? k.value : 0) * sh[h.value], row && ok[h.value], zo);
          } else {
            Fn[k.value][h.value] = slot[fb[h.value] + k.value * sh[h.value]];
          }
        });
      });
    };

    float V[NX][NR];         // [V | v] of the later step: row i = registers [i][0..NR-1]
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int h = 0; h < NR; ++h) V[i][h] = 0.f;
    const float eaff = affl ? 1.f : 0.f;     // unit vector of the affine column (in register AB)
    // MPC: the step's controls and bounds (the same in the 16 lanes of a trajectory) and this lane's elements of [x_t; u_t],
    // loaded one step ahead (older than the slot's refill: the counted wait lets that one fly)
    struct MpcIn {
      float uc[NU], lb[NU], ub[NU], tau[NR];
    };
    MpcIn mcur, mnxt;
    float kprev[NU];
    int n_total = 0;
    auto mpc_load = [&](int t, MpcIn &m) __attribute__((always_inline)) {
      if constexpr (MPC) {
        const size_t tbm = (size_t)(t < 0 ? 0 : t) * B + b;
#pragma unroll
        for (int q = 0; q < NU; ++q) {      // (PAD: an unused control reads control 0 - its box becomes [-1, 1] below)
          const int qc = (!PAD || q < nu) ? q : 0;
          m.uc[q] = a.mpc_controls[tbm * nu + qc];
          m.lb[q] = a.mpc_lower[tbm * nu + qc];
          m.ub[q] = a.mpc_upper[tbm * nu + qc];
        }
#pragma unroll
        for (int h = 0; h < NR; ++h) {
          m.tau[h] = 0.f;          // lc[h]: where this lane's column lies in [x; u] (-1: the affine column, padding)
          if (a.mpc_states != nullptr && lc[h] >= 0)
            m.tau[h] = lc[h] < nx ? a.mpc_states[tbm * nx + lc[h]] : a.mpc_controls[tbm * nu + (lc[h] - nx)];
        }
      }
    };
    if constexpr (MPC) {
#pragma unroll
      for (int q = 0; q < NU; ++q) kprev[q] = 0.f;
      mpc_load(T - 1, mcur);
    }

    auto step = [&](int t, f4w (&Q4)[NT][NR], float (&Fc)[NX][NR]) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      if constexpr (MPC) {
        if (a.mpc_states != nullptr) {   // c_hat = C tau + c: the row sums land in the affine column        mpc_step.py:305-317
          static_for<0, NS>([&](auto i) {
            float sum = 0.f;          // (tau is 0 in the lanes that hold no column of C)
#pragma unroll
            for (int h = 0; h < NR; ++h) sum = fmaf(Q4[i.value / 4][h][i.value % 4], aff[h] ? 0.f : mcur.tau[h], sum);
            sum = group_sum<16>(sum);
            Q4[i.value / 4][AB][i.value % 4] += affl ? sum : 0.f;
          });
        }
      }
      if (t < T - 1) {
        if (!has_f) {
#pragma unroll
          for (int k = 0; k < NX; ++k) Fc[k][AB] = affl ? 0.f : Fc[k][AB];
        }
        // Q~ += F~^T V^ F~ + [0 | F~^T v] as (V^T F~)^T F~ - two chains of outer products (lqr_recursion.py:89,96):
        //   G[b][j] = sum_a V[a][b] F~[a][j]        rows b = the state columns of V: tiles of 4, A = V[a] lanes 4I..4I+3
        //   g1[j]   = sum_a v[a] F~[a][j]           (broadcast-FMAs: v[a] is one lane of V[a])
        //   Q~[i][j] += sum_b G[b][i] F~[b][j]  and  Q~[i][aff] += g1[i]
        f4w G4[NX / 4][NR];
        static_for<0, NX / 4>([&](auto I) {
          static_for<0, NR>([&](auto h) { G4[I.value][h.value] = f4w{0.f, 0.f, 0.f, 0.f}; });
        });
        static_for<0, NX>([&](auto a_) {
          static_for<0, NX / 4>([&](auto I) {
            static_for<0, NR>([&](auto h) {
              G4[I.value][h.value] = mfma_rows<I.value>(V[a_.value][0], Fc[a_.value][h.value], G4[I.value][h.value]);
            });
          });
        });
        float ga[NR], gb[NR], g1[NR];
#pragma unroll
        for (int h = 0; h < NR; ++h) ga[h] = gb[h] = 0.f;
        Blk::g1(ga, gb, V, Fc);
#pragma unroll
        for (int h = 0; h < NR; ++h) g1[h] = ga[h] + gb[h];
        static_for<0, NX>([&](auto b_) {
          static_for<0, NT>([&](auto I) {
            constexpr int ib = (4 * I.value) / 16, il = ((4 * I.value) % 16) / 4;   // rows 4I..: register ib, lanes 4 il..
            const float gcol = G4[b_.value / 4][ib][b_.value % 4];                   // row b of G, the register with columns 4I..
            static_for<0, NR>([&](auto h) {
              Q4[I.value][h.value] = mfma_rows<il>(gcol, Fc[b_.value][h.value], Q4[I.value][h.value]);
            });
          });
        });
        static_for<0, NT>([&](auto I) {
          constexpr int ib = (4 * I.value) / 16, il = ((4 * I.value) % 16) / 4;
          Q4[I.value][AB] = mfma_rows<il>(g1[ib], eaff, Q4[I.value][AB]);
        });
      }
      // K~ = -Quu^-1 [Qux | Quu | qu] on the rows (:112-120)
      float Qu[NU][NR], Kt[NU][NR], R[NU][NR];
      static_for<0, NU>([&](auto m) {
        constexpr int i = NX + m.value;
        static_for<0, NR>([&](auto h) { Qu[m.value][h.value] = Kt[m.value][h.value] = Q4[i / 4][h.value][i % 4]; });
      });
      if constexpr (MASKED) {
        bool act[NU];
        static_for<0, NU>([&](auto m) { act[m.value] = (!PAD || m.value < nu) && a.mask[tb * nu + (m.value < nu ? m.value : 0)] != 0; });
        static_for<0, NR>([&](auto h) {
          const int col = 16 * h.value + lane;
          bool act_col = false;   // this lane's column is the column of a clamped control
          static_for<0, NU>([&](auto m) { act_col = act_col || (col == NX + m.value && act[m.value]); });
          static_for<0, NU>([&](auto m) {
            const float free_row = act_col ? 0.f : Kt[m.value][h.value];
            Kt[m.value][h.value] = act[m.value] ? ((col == NX + m.value) ? 1e-8f : 0.f) : free_row;
          });
        });
      }
      if constexpr (MPC) {
        // every lane gets Quu, qu and the QP's bounds                                              mpc_step.py:119-138
        float Quu[NU][NU], qu[NU], lo[NU], hi[NU], kt[NU];
        static_for<0, NU>([&](auto l) {
          constexpr int lb_ = (NX + l.value) / 16, ll = (NX + l.value) % 16;
          static_for<0, NU>([&](auto m) { Quu[m.value][l.value] = G::template bcast<ll>(Qu[m.value][lb_]); });
        });
        static_for<0, NU>([&](auto m) {
          qu[m.value] = G::template bcast<NS % 16>(Qu[m.value][AB]);
          const bool used = !PAD || m.value < nu;     // (an unused control: qu = 0 in the box [-1, 1] - its QP solution is 0)
          lo[m.value] = used ? mcur.lb[m.value] - mcur.uc[m.value] : -1.f;
          hi[m.value] = used ? mcur.ub[m.value] - mcur.uc[m.value] : 1.f;
          kt[m.value] = kprev[m.value];
        });
        PnqpResult<NU> qp;
        pnqp_solve_rows<NU>(Quu, qu, lo, hi, kt, /*warm=*/t != T - 1, a.mpc_n_qp_iter, qp);      // :141-146
        n_total += 1 + qp.it;
        if (!qp.converged) info_bits |= 4;
        // K_t = -LU_free^-1 Qux with the rows of clamped controls zeroed (:147-157); the affine column carries k_t
        static_for<0, NR>([&](auto h) {
          float col[NU];
#pragma unroll
          for (int m = 0; m < NU; ++m) col[m] = qp.free_[m] ? Qu[m][h.value] : 0.f;
          lu_solve_rinv<NU>(qp.fac, qp.piv, qp.rinv, col);
#pragma unroll
          for (int m = 0; m < NU; ++m) Kt[m][h.value] = (h.value == AB && affl) ? kt[m] : -col[m];
        });
#pragma unroll
        for (int m = 0; m < NU; ++m) kprev[m] = kt[m];
      } else {
        if (gauss_jordan_rows_wide<NX, NU, NR>(Kt)) info_bits |= 1;
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
          for (int h = 0; h < NR; ++h) Kt[m][h] = -Kt[m][h];
      }
      // gain rows [K_m | 0 | k_m] to the workspace (the rollout reads them like rows of F), and to the caller
      {
        if constexpr (!MPC) {
          float *row = kw + (size_t)t * B * (NU * KROW);
          if (lane < NX) {
#pragma unroll
            for (int m = 0; m < NU; ++m) row[m * KROW + lane] = Kt[m][0];
          }
          if (affl) {
#pragma unroll
            for (int m = 0; m < NU; ++m) row[m * KROW + NS] = Kt[m][AB];
          }
        }
        if (a.Ks != nullptr) {
          if (lane < nx) {
#pragma unroll
            for (int m = 0; m < NU; ++m)
              if (!PAD || m < nu) a.Ks[(tb * nu + m) * nx + lane] = Kt[m][0];
          }
          if (affl) {
#pragma unroll
            for (int m = 0; m < NU; ++m)
              if (!PAD || m < nu) a.ks[tb * nu + m] = Kt[m][AB];
          }
        }
      }
      if (t > 0) {  // V = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~), :151-152
#pragma unroll
        for (int m = 0; m < NU; ++m)
#pragma unroll
          for (int h = 0; h < NR; ++h) R[m][h] = Qu[m][h];
        Blk::rk(R, Qu, Kt);
        f4w V4[NX / 4][NR];
        static_for<0, NX / 4>([&](auto I) {
          static_for<0, NR>([&](auto h) { V4[I.value][h.value] = Q4[I.value][h.value]; });
        });
        static_for<0, NU>([&](auto m) {            // K~^T R: A = K~[m] lanes 4I..4I+3 (the state columns: first register)
          static_for<0, NX / 4>([&](auto I) {
            static_for<0, NR>([&](auto h) {
              V4[I.value][h.value] = mfma_rows<I.value>(Kt[m.value][0], R[m.value][h.value], V4[I.value][h.value]);
            });
          });
        });
        float Qx[NX][NR];
        static_for<0, NX>([&](auto i) {
          static_for<0, NR>([&](auto h) {
            V[i.value][h.value] = V4[i.value / 4][h.value][i.value % 4];
            Qx[i.value][h.value] = Q4[i.value / 4][h.value][i.value % 4];
          });
        });
        Blk::vq(V, Qx, Kt);                         // + Qxu K~
      }
    };

    // ONE register set (two would not fit next to V and W): at the top of step t its slot is waited for and read, then -
    // once the reads are in - refilled with step t - DB, and the step is computed while DB - 1 fetches are in flight.  The
    // gain-row stores of a step are younger than its refill: waiting for all but (DB - 1) kDmaB operations is exact for the
    // first step and conservative afterwards (the stores have had a whole step to complete; allowing for them - 2 NU per
    // step - would rely on stores and loads retiring in one order, and measured no faster).
    f4w Q[NT][NR];
    float Fc[NX][NR];
    static_for<0, DB>([&](auto j) { issue_next(j.value); });
    for (int t0 = T - 1; t0 >= 0; t0 -= DB) {
      static_for<0, DB>([&](auto j) {
        const int t = t0 - j.value;
        if (t >= 0) {
          wait_vmcnt<(DB - 1) * Lay::kDmaB>();
          read_slot(ring + j.value * Lay::SLOT_B, Q, Fc);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are in before it is refilled
          if constexpr (MPC) mpc_load(t - 1, mnxt);
          issue_next(j.value);
          step(t, Q, Fc);
          if constexpr (MPC) mcur = mnxt;
        }
      });
    }
    wait_vmcnt<0>();   // the ring is reused by the rollout, and this wave's gain rows have reached L2
    if constexpr (MPC) {
      if (lane == 0) {
        a.mpc_n_qp_total[b] = n_total;
        if (a.info != nullptr) {
          if (a.info_store) a.info[b] = info_bits;
          else if (info_bits != 0) atomicOr(&a.info[b], info_bits);
        }
      }
      return;
    }
    __threadfence();
  }

  // ------------------------------------------------------------------ forward rollout (lqr_recursion.py:160-200)
  {
    unsigned long long ptr[Lay::kDmaF], str[Lay::kDmaF];
    bool kstep[Lay::kDmaF];      // lanes that fetch gain rows (T slices: they take the last pointer step alone)
#pragma unroll
    for (int q = 0; q < Lay::kDmaF; ++q) {
      const int g = q * 64 + lane64;
      const int gg = g < Lay::CH_F ? g : 0;
      const bool isk = gg >= (Lay::F_FL + Lay::f_FL) / 4;
      const bool isf = !isk && gg >= Lay::F_FL / 4;
      const char *base = isk ? (const char *)a.wsK : isf ? (const char *)(has_f ? a.f : a.C) : (const char *)(T > 1 ? a.F : a.C);
      const size_t per = isk ? (size_t)NU * KROW * 4 : isf ? (size_t)nx * 4 : (size_t)nx * ns * 4;   // (the gain rows: container layout)
      const int g0 = isk ? (Lay::F_FL + Lay::f_FL) / 4 : isf ? Lay::F_FL / 4 : 0;
      const int gc = (!PAD || (size_t)(gg - g0) * 16 < 4 * per) ? gg - g0 : 0;     // PAD: behind the array's end, its chunk 0 again
      ptr[q] = (unsigned long long)base + (size_t)b0 * per + (size_t)gc * 16 - (unsigned long long)(q % 4) * 1024u;
      str[q] = (unsigned long long)(B * per);
      kstep[q] = isk;
    }
    int ti = 0;
    auto issue_next = [&](int slot) __attribute__((always_inline)) {
      const unsigned dst = ring_addr + (unsigned)slot * (Lay::SLOT_F * 4);
      static_for<0, Lay::kDmaF>([&](auto q) {
        if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
        dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
      });
      if (ti < T - 2) {
#pragma unroll
        for (int q = 0; q < Lay::kDmaF; ++q) ptr[q] += str[q];
        ++ti;
      } else if (ti == T - 2) {
#pragma unroll
        for (int q = 0; q < Lay::kDmaF; ++q) ptr[q] += kstep[q] ? str[q] : 0ull;
        ++ti;
      }
    };
    // lane i < NX owns row i of [F_t | f_t]; lane m < NU ALSO owns row m of the gains [K_t | . | k_t]:
    //     u[m]   = k[m] + sum_j K[m][j] x[j]                        (lanes m < NU; x[j] broadcast from lane j)
    //     x'[i]  = f[i] + sum_j F[i][j] x[j] + sum_m F[i][NX+m] u[m]  (lanes i < NX; u[m] broadcast from lane m)
    const int lane_x = lane < nx ? lane : nx - 1, lane_u = lane < NU ? lane : NU - 1;
    const int xrow = r * nx * ns + lane_x * ns, xaff = Lay::F_FL + r * nx + lane_x;
    const int urow = Lay::F_FL + Lay::f_FL + (r * NU + lane_u) * KROW;
    auto read_rows = [&](int slot, float (&Fr)[NS + 1], float (&Kr)[NX + 1]) __attribute__((always_inline)) {
      const float *s = ring + slot * Lay::SLOT_F;
      static_for<0, NS>([&](auto j) {
        if constexpr (PAD) {
          const int lj = logical(j.value);   // uniform
          Fr[j.value] = padded_read(s + xrow + (lj >= 0 ? lj : 0), lj >= 0, zo);
        } else {
          Fr[j.value] = s[xrow + j.value];
        }
      });
      Fr[NS] = s[xaff];
#pragma unroll
      for (int j = 0; j < NX; ++j) Kr[j] = s[urow + j];
      Kr[NX] = s[urow + NS];
    };
    float xv = lane < nx ? a.x_init[(size_t)b * nx + lane] : 0.f;
    auto fstep = [&](int t, const float (&Fr)[NS + 1], const float (&Kr)[NX + 1]) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      float ua = Kr[NX], ub = 0.f;
      static_for<0, NX>([&](auto j) {
        const float xj = G::template bcast<j.value>(xv);
        if constexpr (j.value % 2 == 0) ua = fmaf(xj, Kr[j.value], ua);
        else ub = fmaf(xj, Kr[j.value], ub);
      });
      const float uv = ua + ub;
      if (lane < nx) a.x[tb * nx + lane] = xv;
      if (lane < nu) a.u[tb * nu + lane] = uv;
      float xa = has_f ? Fr[NS] : 0.f, xb = 0.f;   // (for t = T-1 this consumes a re-fetched F_{T-2}: never used)
      static_for<0, NX>([&](auto j) {
        const float xj = G::template bcast<j.value>(xv);
        if constexpr (j.value % 2 == 0) xa = fmaf(xj, Fr[j.value], xa);
        else xb = fmaf(xj, Fr[j.value], xb);
      });
      static_for<0, NU>([&](auto m) {
        const float um = G::template bcast<m.value>(uv);
        if constexpr (m.value % 2 == 0) xa = fmaf(um, Fr[NX + m.value], xa);
        else xb = fmaf(um, Fr[NX + m.value], xb);
      });
      xv = (!PAD || lane < nx) ? xa + xb : 0.f;    // (PAD: the unused state lanes stay an exact 0 - they are broadcast)
    };
    float FA[NS + 1], KA[NX + 1], FB[NS + 1], KB[NX + 1];
    static_for<0, DF>([&](auto j) { issue_next(j.value); });
    wait_vmcnt<(DF - 1) * Lay::kDmaF>();
    read_rows(0, FA, KA);
    for (int t0 = 0; t0 < T; t0 += DF) {
      static_for<0, DF>([&](auto j) {
        const int t = t0 + j.value;
        if (t < T) {
          constexpr int nslot = (j.value + 1) % DF;
          // the rows read out of this slot (a step ago; for t = 0: just now) are in before it is refilled - a
          // cache-resident refill was seen to overtake the prologue's reads (wrong rows of F at the first step)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          issue_next(j.value);
          wait_vmcnt<(DF - 1) * Lay::kDmaF>();
          if constexpr (j.value % 2 == 0) {
            read_rows(nslot, FB, KB);
            fstep(t, FA, KA);
          } else {
            read_rows(nslot, FA, KA);
            fstep(t, FB, KB);
          }
        }
      });
    }
    wait_vmcnt<0>();
    if (!is_finite(xv)) info_bits |= 2;   // NaN / Inf propagate through the recursion
  }
  if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
