// lqr_wave_mfma.hpp - backward Riccati sweep for the large shapes (one wavefront per trajectory, ns + 1 <= 64,
// nx, nu and ns multiples of 4 - (32,8) of BASELINE.json configs[4]) with every A^T B product on the matrix cores.
//
// Same recursion and column-per-lane layout as lqr_kernel<NX, NU, 64, ...> (lqr/lqr_recursion.py:69-158): lane j
// holds column j of [C|c], [F|f], [V|v]; the affine column is lane ns.  Who multiplies:
//   v_mfma_f32_4x4x1_16b_f32 with cbsz:4 abid:I takes 4 lanes (block I) of the A register and all 64 lanes of the B
//   register and adds the outer product to a 4-row tile held in 4 consecutive registers:
//       D[4I + i][j] += A[lane 4I + i] * B[lane j]                        (one trajectory per wavefront)
//   i.e. it computes P^T R from column-per-lane P and R, one contraction index per instruction.  With the
//   reference's own association  Q~ = C~ + (F^T V) F~  (lqr_recursion.py:89,96) both products have that shape:
//       G  = V^^T F~        rows b = columns of [V|v] (tile TA, row 0 is the homogeneous row g1 = v^T F~)
//       Q~ += G^T F~ + g1 (x) e_aff
//   and so has Qxu K~ of the value update once Qxu is held as a row matrix XU (lane i = Q[i][nx+m]): the control lanes
//   write their columns of the finished x rows to LDS and lane i reads row i back (1 KB per wavefront).
//
// fp32 MFMA runs on the same FMA lanes as the VALU (their times add up, profiles/r01/microbench_mfma16_shadow.txt),
// so a timestep costs (MFMAs x 8 + other instructions x 4) cycles plus whatever latency is exposed.  Round 2 removes
// the "other" part and the exposed latency:
//   * inputs of step t-1 are fetched while step t computes (two register banks, one wavefront per SIMD with the
//     accumulation registers as the second half of the file), through buffer loads whose address is an SGPR
//     descriptor + a per-lane constant + an immediate - no 64-bit pointer steps on the VALU; lanes beyond the
//     matrix read out of range (= 0), lane ns gets c / f by one exec-masked block of 16-byte loads;
//   * the gains come from a Gauss-Jordan elimination carried out ON the rows of [Qux | Quu | qu] as they lie
//     (one readlane per multiplier, one FMA per row) instead of an 8 x 8 LU replicated in every lane; partial
//     pivoting (LAPACK's choice of row) sits behind a uniform branch taken only when a row really has to move;
//   * V~ = Q~x. + Qxu K~ : the fourth term K~^T (Q~u. + Quu K~) of lqr_recursion.py:151-152 multiplies the RESIDUAL of
//     the gain solve (Quu K~ = -Q~u. up to rounding) and is left out here; the masked variant (LQR_active,
//     mpc/active_constrained_lqr.py:110-145), where K~ solves the MASKED system and V~ uses the unmasked blocks,
//     computes it in full.
// The gains go to HBM (caller's Ks/ks or the workspace): at this size they do not fit in LDS; the rollout is the
// forward-only lqr_kernel, which is bandwidth bound.
#pragma once
#include "colwise.hpp"
#include "dma_gather.hpp"
#include "lqr_kernels.hpp"
#include "pnqp_device.hpp"

namespace dmpc {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int I>
__device__ __forceinline__ f4v mfma_bcast(float a, float b, f4v c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, I, 0);  // cbsz = 4: block I of A feeds all 16 blocks
}

// raw buffer descriptor (gfx9 layout: 48-bit base, stride 0, num_records in bytes, DATA_FORMAT 32): an access at or
// beyond num_records returns 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rsrc(const void *base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
// Cache policy knob of the sweep's loads of C and F (aux bit 1 = nt on gfx940+).  Measured (round 3, same box, B = 8192): nt
// 2.34 ms against 1.88 ms - the row loads (164 B every 160 B) share cache lines that nt gives up.  Off.
#ifndef DMPC_WAVE_NT
#define DMPC_WAVE_NT 0
#endif
template <int BYTE_OFF>
__device__ __forceinline__ float wave_load(__amdgpu_buffer_rsrc_t r, int voff) {
  // the immediate of a buffer load holds 12 bits: rows further down go through the scalar offset
  constexpr int kImm = BYTE_OFF % 4096, kS = BYTE_OFF - kImm;
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff + kImm, kS, DMPC_WAVE_NT ? 2 : 0));
}

constexpr int kOutOfRange = 0x40000000;

// DMPC_WAVE_DMA 1 (default): the inputs of step t-1 travel HBM -> LDS by LDS-DMA while step t computes, and the register
// bank is filled from LDS at the top of the step - a prefetch that costs no registers (the two-bank form above needs the
// whole file of a SIMD for one wavefront) and turns the sweep's 164-byte row loads into whole kilobytes.
//                0: buffer loads straight into the bank at the top of the step (rounds 1-2)
#ifndef DMPC_WAVE_DMA
#define DMPC_WAVE_DMA 1
#endif
#ifndef DMPC_WAVE_DMA_NT
#define DMPC_WAVE_DMA_NT 0
#endif
#ifndef DMPC_WAVE_COUNTED_WAIT
#define DMPC_WAVE_COUNTED_WAIT 1
#endif
// ROLLOUT only: wavefronts in an odd slot of their SIMD start DMPC_WAVE_STAGGER_TICKS (100 MHz ticks) late when the grid
// takes several rounds, so that the two wavefronts of a SIMD are not in the same phase: the rollout (waits on memory)
// of one then runs under the sweep (issue bound) of the other instead of both rolling out at once.
// Measured (round 3, B = 8192, same box): 0 -> 1.852 ms, 5,000 -> 1.85, 10,000 -> 1.79, 20,000 -> 1.81, 40,000 -> 1.88.
#ifndef DMPC_WAVE_STAGGER_TICKS
#define DMPC_WAVE_STAGGER_TICKS 10000
#endif
#ifndef DMPC_WAVE_JITTER_TICKS
#define DMPC_WAVE_JITTER_TICKS 0
#endif
#if DMPC_WAVE_DMA_NT
#define DMPC_WAVE_DMA_POLICY " nt"
#else
#define DMPC_WAVE_DMA_POLICY ""
#endif
// one LDS-DMA instruction, uniform base + per-lane byte offset: lane l copies 16 bytes from base + voff + OFF to LDS at
// M0 + OFF + 16 l (the instruction offset moves both addresses; 12 bits + sign)
template <int OFF>
__device__ __forceinline__ void wave_dma16(unsigned voff, unsigned long long base) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" DMPC_WAVE_DMA_POLICY ::"v"(voff), "s"(base), "n"(OFF) : "memory");
}
// the same for the first LANES lanes only (the tail of a region); every lane is active around it
template <int OFF, int LANES>
__device__ __forceinline__ void wave_dma16_tail(unsigned voff, unsigned long long base) {
  static_assert(LANES >= 1 && LANES <= 32, "the mask is a 32-bit literal");
  asm volatile("s_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1 offset:%2" DMPC_WAVE_DMA_POLICY "\n\ts_mov_b64 exec, -1"
               ::"v"(voff), "s"(base), "n"(OFF), "n"((1ll << LANES) - 1) : "memory");
}
// a contiguous region of BYTES bytes (a multiple of 16) at `src` (wave-uniform) -> LDS byte address `dst`
template <int BYTES>
__device__ __forceinline__ void wave_dma_region(const void *src, unsigned dst, unsigned voff16) {
  static_assert(BYTES % 16 == 0, "whole 16-byte chunks");
  constexpr int kChunks = BYTES / 16, kSeg = (kChunks + 255) / 256;     // a segment = 4 instructions = 4 KB
  static_for<0, kSeg>([&](auto sg) {
    constexpr int c0 = sg.value * 256;
    set_m0(dst + sg.value * 4096);
    const unsigned long long base = reinterpret_cast<unsigned long long>(src) + sg.value * 4096;
    static_for<0, 4>([&](auto q) {
      constexpr int first = c0 + q.value * 64, left = kChunks - first;
      if constexpr (left >= 64) wave_dma16<q.value * 1024>(voff16, base);
      else if constexpr (left > 32) {   // 33..63 lanes: two halves (the mask literal holds 32 bits)
        wave_dma16_tail<q.value * 1024, 32>(voff16, base);
        wave_dma16_tail<q.value * 1024 + 512, left - 32>(voff16, base);
      } else if constexpr (left > 0) wave_dma16_tail<q.value * 1024, left>(voff16, base);
    });
  });
}

// Closed-loop rollout (LqrRecursion.forward, lqr/lqr_recursion.py:160-200) by the wavefront that has just finished the
// backward sweep of the same trajectory: while it waits on memory here, the other wavefront of its SIMD is in the
// compute-bound sweep of ANOTHER trajectory, so the bandwidth-bound and the compute-bound halves of the solve overlap
// instead of running as two launches one after the other.  Row-per-lane: lane k < nx holds row k of [F_t | f_t], lane
// nx+m row m of [K_t | 0 | k_t]; ONE pass  acc = aff + sum_i w[i] x_t[i]  (x_t[i] by readlane) gives the state part of
// x_{t+1} in the F lanes and u_t in the K lanes, then  acc += w[nx+m] u_t[m]  completes x_{t+1}.  The gains were
// written by this wavefront (other lanes) moments ago: they are read back with glc (from L2, never a stale L1 line).
template <int NX, int NU, bool MASKED, int kRing = 4>
__device__ __forceinline__ void wave_rollout(const LqrArgs &a, const int b, const int lane, const bool live,
                                             const float *Ks, const float *ks, int &info_bits) {
  constexpr int NS = NX + NU, TS = NS / 4, TX = NX / 4;
  using G64 = Group<64>;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const bool f_lane = lane < NX, g_lane = lane >= NX && lane < NS;
  const int voff_f = f_lane ? lane * NS * 4 : kOutOfRange, voff_f1 = f_lane ? lane * 4 : kOutOfRange;
  const int voff_k = g_lane ? (lane - NX) * NX * 4 : kOutOfRange, voff_k1 = g_lane ? (lane - NX) * 4 : kOutOfRange;
  struct Row {   // NB: never __builtin_bit_cast a vector ELEMENT (hipcc 7.2 then reads element 0): whole vectors only
    f4v w[TS];
    float aff;
    unsigned act;
  };
  auto fetch = [&](int t, Row &r) {
    if (t >= T) return;   // uniform
    const size_t tb = (size_t)t * B + b;
    const __amdgpu_buffer_rsrc_t rK = wave_rsrc(Ks + tb * NU * NX, NU * NX * 4), rk = wave_rsrc(ks + tb * NU, NU * 4);
    u4v wk[TX];
#pragma unroll
    for (int q = 0; q < TX; ++q) wk[q] = __builtin_amdgcn_raw_buffer_load_b128(rK, voff_k + 16 * q, 0, 1);   // glc
    unsigned aff = __builtin_amdgcn_raw_buffer_load_b32(rk, voff_k1, 0, 1);
    u4v wf[TS];
    if (t < T - 1) {
      const __amdgpu_buffer_rsrc_t rF = wave_rsrc(a.F + tb * NX * NS, NX * NS * 4);
#pragma unroll
      for (int q = 0; q < TS; ++q) wf[q] = __builtin_amdgcn_raw_buffer_load_b128(rF, voff_f + 16 * q, 0, DMPC_WAVE_NT ? 2 : 0);
      if (has_f) aff |= __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(a.f + tb * NX, NX * 4), voff_f1, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < TS; ++q) wf[q] = u4v{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int q = 0; q < TS; ++q) {   // a lane is in one of the two row sets; the other load gave it 0
      const u4v m = q < TX ? (wf[q] | wk[q < TX ? q : 0]) : wf[q];
      r.w[q] = __builtin_bit_cast(f4v, m);
    }
    r.aff = __builtin_bit_cast(float, aff);
    if constexpr (MASKED) {
      unsigned bits = 0;
#pragma unroll
      for (int m = 0; m < NU; ++m) bits |= (a.mask[tb * NU + m] != 0 ? 1u : 0u) << m;
      r.act = __builtin_amdgcn_readfirstlane(bits);
    }
  };
  float xv = f_lane ? a.x_init[(size_t)b * NX + lane] : 0.f;
  bool bad = false;
  auto fstep = [&](int t, const Row &r) {
    const size_t tb = (size_t)t * B + b;
    if (f_lane && live) a.x[tb * NX + lane] = xv;
    // x_t into NX scalar registers first, the FMAs after: one scalar register reused for every broadcast puts a
    // wait state and a write-after-read stall between each readlane and its FMA
    float xb[NX];
    static_for<0, NX>([&](auto i) { xb[i.value] = G64::template bcast<i.value>(xv); });
    asm volatile("" ::: "memory");
    float acc[4] = {r.aff, 0.f, 0.f, 0.f};
    static_for<0, NX>([&](auto i) {
      const float w = r.w[i.value / 4][i.value % 4];
      acc[i.value % 4] = fmaf(w, xb[i.value], acc[i.value % 4]);   // :177 and the state part of :189
    });
    float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    float uo = s;
    if constexpr (MASKED) uo = (g_lane && ((r.act >> (lane - NX)) & 1u)) ? 0.f : s;       // :179-183
    if (g_lane && live) a.u[tb * NU + (lane - NX)] = uo;
    bad = bad || ((f_lane || g_lane) && !(fabsf(uo) <= 3.0e38f)) || (f_lane && !(fabsf(xv) <= 3.0e38f));
    static_for<0, NU>([&](auto m) {
      const float w = r.w[(NX + m.value) / 4][(NX + m.value) % 4];
      s = fmaf(w, G64::template bcast<NX + m.value>(uo), s);                              // control part of :189
    });
    if (f_lane) xv = s;
  };
  // a step is ~100 instructions, a trip to HBM several times that: rows are requested kRing steps ahead
  Row ring[kRing];
  static_for<0, kRing - 1>([&](auto j) { fetch(j.value, ring[j.value]); });
  for (int t0 = 0; t0 < T; t0 += kRing) {
    static_for<0, kRing>([&](auto j) {
      const int t = t0 + j.value;
      if (t < T) {   // uniform
        fetch(t + kRing - 1, ring[(j.value + kRing - 1) % kRing]);
        fstep(t, ring[j.value]);
      }
    });
  }
  if (bad) info_bits |= 2;
}


#ifndef DMPC_WAVE_PREFETCH
#define DMPC_WAVE_PREFETCH 0
#endif
// DMPC_WAVE_PREFETCH 1: one wavefront per SIMD (512 registers), inputs of the next step in a second register bank;
//                    0: two wavefronts per SIMD (256 registers each), each hides the other's latencies
// ROLLOUT: the wavefront goes on with the forward sweep of its trajectory (solve_recursion in ONE launch)
#ifndef DMPC_WAVE_OCC
#define DMPC_WAVE_OCC 2    // wavefronts per SIMD the register budget is set for (experiments: 3 -> 168 registers)
#endif
// PAD: container for a smaller problem (a.nx_log <= NX, a.nu_log <= NU; lqr_kernel<..., PAD> of lqr_kernels.hpp has the
// argument): rows by buffer loads at the problem's own strides, columns outside it out of range (= 0), a unit diagonal for
// the unused controls; sweep only (the rollout is the forward-only container kernel).
// MPC: MPCstep.backward_rec (mpc/mpc_step.py:70-173) - the feed-forward term k_t is the solution of the box QP on (Quu, qu)
// (pnqp_solve_rows of pnqp_device.hpp, executed by every lane on the same broadcast data, warm-started from the later
// step), K_t comes from the QP's own last factorisation with the rows of clamped controls zeroed (:147-157), the value update
// keeps the unmasked blocks (:165-166).  The QP's registers (Quu, its LU, the iterates) come on top of the sweep's: one
// wavefront per SIMD, and the LDS-DMA slot is what keeps the loads ahead of it.
template <int NX, int NU, bool MASKED, bool ROLLOUT, bool PAD = false, bool MPC = false>
__global__ __launch_bounds__(256, (DMPC_WAVE_PREFETCH || MPC) ? 1 : DMPC_WAVE_OCC) void lqr_wave_mfma_backward(const LqrArgs a) {
  static_assert(!(PAD && ROLLOUT), "containers roll out in a launch of their own");
  static_assert(!MPC || (!MASKED && !ROLLOUT && DMPC_WAVE_DMA && !DMPC_WAVE_PREFETCH), "MPC: the plain sweep (DMA form, or padded)");
  constexpr int NS = NX + NU, AFF = NS;
  static_assert(NX % 4 == 0 && NU % 4 == 0 && NS + 1 <= 64, "tiles of 4 rows, one wavefront per trajectory");
  constexpr int TX = NX / 4, TS = NS / 4, TA = AFF / 4, TU = NU / 4;
  using G64 = Group<64>;

  const int lane = threadIdx.x & 63;
  int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  b = __builtin_amdgcn_readfirstlane(b);
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const bool col_aff = lane == AFF;
  const bool k_lane = lane < NX || col_aff;
  const float eaff = col_aff ? 1.f : 0.f;
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int i) -> int { return i < NX ? (i < nx ? i : -1) : (i - NX < nu ? nx + (i - NX) : -1); };
  const int lcol = lane < NS ? logical(lane) : -1;
  const int voff_col = (PAD ? lcol >= 0 : lane < NS) ? (PAD ? lcol : lane) * 4 : kOutOfRange;   // this lane's column of a row of C / F
  float *Ks = a.Ks != nullptr ? a.Ks : a.wsK;
  float *ks = a.Ks != nullptr ? a.ks : a.wsk;
  int info_bits = 0;

  if constexpr (ROLLOUT && (DMPC_WAVE_STAGGER_TICKS > 0 || DMPC_WAVE_JITTER_TICKS > 0)) {
    // HW_ID (hwreg 4) bits 3:0 = the wavefront's slot in its SIMD; first round only (later rounds inherit the offset)
    const unsigned slot_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);
    constexpr unsigned kResident = 2 * 256;   // workgroups the chip holds at once: two per CU (LDS, registers)
    if (gridDim.x > kResident && blockIdx.x < kResident) {
      // DMPC_WAVE_JITTER_TICKS: besides the half-period offset of the odd slots, every wavefront starts at its own point of a
      // step (16 phases over JITTER ticks): 2,048 wavefronts that run in lock step request their 11.7 KB slots in the same
      // microsecond of every step - a 24 MB burst that the memory system needs most of a step to drain
      const unsigned phase = ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 7u) & 15u;
      const unsigned long long wait_ticks = ((slot_id & 1u) ? (unsigned long long)DMPC_WAVE_STAGGER_TICKS : 0ull) +
                                            (unsigned long long)(DMPC_WAVE_JITTER_TICKS) * phase / 16u;
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < wait_ticks) __builtin_amdgcn_s_sleep(8);
    }
  }
  float crow_pad[(MPC && PAD) ? NS : 1];   // MPC in a container: this lane's row of C_t for the re-centring
  struct Bank {
    f4v Q4[TS];    // rows of [C_t | c_t], column-per-lane
    float Fc[NX];  // rows of [F_t | f_t]
    unsigned act;  // MASKED: bit m = control m is clamped at this step
  };

  // The fetch of a step is issued in two parts (cost rows at the top of the previous step, dynamics rows after its
  // first product): at most 63 vector-memory instructions can be outstanding per wavefront, a 64th stalls the
  // wave until the oldest has returned.
  auto fetch_cost = [&](int t, Bank &k) {
    if (t < 0) return;   // uniform
    const size_t tb = (size_t)t * B + b;
    if constexpr (PAD) {
      const __amdgpu_buffer_rsrc_t rc = wave_rsrc(a.C + tb * ns * ns, ns * ns * 4);
      static_for<0, NS>([&](auto i) {
        const int li = logical(i.value);   // uniform
        float v = (i.value >= NX && lane == i.value) ? 1.f : 0.f;   // (row of an unused control: the unit diagonal)
        if (li >= 0) {
          v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, voff_col, li * ns * 4, 0));
          if (col_aff) v = a.c[tb * ns + li];
        }
        k.Q4[i.value / 4][i.value % 4] = v;
      });
      if constexpr (MASKED) {
        unsigned bits = 0;
#pragma unroll
        for (int m = 0; m < NU; ++m) bits |= (m < nu && a.mask[tb * nu + (m < nu ? m : 0)] != 0 ? 1u : 0u) << m;
        k.act = __builtin_amdgcn_readfirstlane(bits);
      }
      if constexpr (MPC) {
        if (a.mpc_states != nullptr) {   // the re-centring's row of C for this lane (container positions; 0 outside the problem)
          const float *rp = a.C + (tb * ns + (lcol >= 0 ? lcol : 0)) * ns;
          static_for<0, NS>([&](auto kk) {
            const int lk = logical(kk.value);   // uniform
            const float v = rp[lk >= 0 ? lk : 0];
            crow_pad[kk.value] = (lk >= 0 && lcol >= 0) ? v : 0.f;
          });
        }
      }
      return;
    }
    const __amdgpu_buffer_rsrc_t rc = wave_rsrc(a.C + tb * NS * NS, NS * NS * 4);
    static_for<0, NS>([&](auto i) { k.Q4[i.value / 4][i.value % 4] = wave_load<i.value * NS * 4>(rc, voff_col); });
    if (col_aff) {   // the affine column: c_t is a contiguous run, fetched by lane ns alone
      const f4v *cp = reinterpret_cast<const f4v *>(a.c + tb * NS);
#pragma unroll
      for (int I = 0; I < TS; ++I) k.Q4[I] = cp[I];
    }
    if constexpr (MASKED) {
      unsigned bits = 0;
#pragma unroll
      for (int m = 0; m < NU; ++m) bits |= (a.mask[tb * NU + m] != 0 ? 1u : 0u) << m;
      k.act = __builtin_amdgcn_readfirstlane(bits);
    }
  };
  auto fetch_dyn = [&](int t, Bank &k) {
    if (t < 0 || t >= T - 1) return;   // uniform; there is no F_{T-1}
    const size_t tb = (size_t)t * B + b;
    if constexpr (PAD) {
      const __amdgpu_buffer_rsrc_t rf = wave_rsrc(a.F + tb * nx * ns, nx * ns * 4);
      static_for<0, NX>([&](auto r) {
        float v = 0.f;
        if (r.value < nx) {   // uniform
          v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf, voff_col, r.value * ns * 4, 0));
          if (col_aff) v = has_f ? a.f[tb * nx + r.value] : 0.f;
        }
        k.Fc[r.value] = v;
      });
      return;
    }
    const __amdgpu_buffer_rsrc_t rf = wave_rsrc(a.F + tb * NX * NS, NX * NS * 4);
    static_for<0, NX>([&](auto r) { k.Fc[r.value] = wave_load<r.value * NS * 4>(rf, voff_col); });
    if (col_aff && has_f) {
      const f4v *fp = reinterpret_cast<const f4v *>(a.f + tb * NX);
#pragma unroll
      for (int I = 0; I < TX; ++I) {
        const f4v v = fp[I];
        k.Fc[4 * I] = v[0]; k.Fc[4 * I + 1] = v[1]; k.Fc[4 * I + 2] = v[2]; k.Fc[4 * I + 3] = v[3];
      }
    }
  };

  __shared__ float xu_all[4][NX * NU];   // per wavefront: Qxu of the current step, [i][m]
  float *xu_lds = xu_all[threadIdx.x >> 6];
#if DMPC_WAVE_DMA && !DMPC_WAVE_PREFETCH
  // ---- the input slot of this wavefront: [C_t | c_t | F_t | f_t] as they lie in HBM, filled by LDS-DMA a step ahead
  constexpr int kSlotC = 0, kSlotc = NS * NS, kSlotF = kSlotc + NS, kSlotf = kSlotF + NX * NS, kSlotFloats = kSlotf + NX;
  static_assert((NS * NS) % 4 == 0 && NS % 4 == 0 && (NX * NS) % 4 == 0 && NX % 4 == 0, "16-byte regions");
  // PAD: the same four regions at the PROBLEM's sizes and strides, each in its own multiple of 64 floats (dma_run_floats of dma_gather.hpp),
  // and a word of zero for the lanes and rows outside the problem
  constexpr int kPadC = 0, kPadc = (NS * NS + 63) / 64 * 64, kPadF = kPadc + 64, kPadf = kPadF + (NX * NS + 63) / 64 * 64,
                kPadZero = kPadf + 64, kPadFloats = kPadZero + 4;
  __shared__ __attribute__((aligned(16))) float slot_all[4][PAD ? kPadFloats : kSlotFloats];
  float *slot = slot_all[threadIdx.x >> 6];
  const unsigned slot_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot);   // LDS byte address (low 32 bits of the pointer)
  const unsigned voff16 = lane * 16;
  auto dma_issue = [&](int t) {   // the slot's reads are in (the caller has waited for them)
    const size_t tb = (size_t)t * B + b;
    wave_dma_region<NS * NS * 4>(a.C + tb * NS * NS, slot_addr + kSlotC * 4, voff16);
    wave_dma_region<NS * 4>(a.c + tb * NS, slot_addr + kSlotc * 4, voff16);
    if (t < T - 1) {   // uniform; there is no F_{T-1}
      wave_dma_region<NX * NS * 4>(a.F + tb * NX * NS, slot_addr + kSlotF * 4, voff16);
      if (has_f) wave_dma_region<NX * 4>(a.f + tb * NX, slot_addr + kSlotf * 4, voff16);
    }
  };
  // slot -> bank.  Lanes beyond the affine column are never written: they hold the zeros the bank starts with (their
  // columns of Q~ only ever add products with those zeros), as the out-of-range buffer loads of the other form give.
  f4v crow[MPC ? TS : 1];   // MPC with re-centring: row min(lane, NS-1) of C_t, taken from the slot before it is refilled
  auto read_bank = [&](int t, Bank &k) {
    if constexpr (MPC) {
      if (a.mpc_states != nullptr) {
        const f4v *rp = reinterpret_cast<const f4v *>(slot + kSlotC + (lane < NS ? lane : NS - 1) * NS);
#pragma unroll
        for (int q = 0; q < TS; ++q) crow[q] = rp[q];
      }
    }
    if (lane < NS) {
      static_for<0, NS>([&](auto i) { k.Q4[i.value / 4][i.value % 4] = slot[kSlotC + i.value * NS + lane]; });
      if (t < T - 1) static_for<0, NX>([&](auto r) { k.Fc[r.value] = slot[kSlotF + r.value * NS + lane]; });
    } else if (col_aff) {
      const f4v *cp = reinterpret_cast<const f4v *>(slot + kSlotc);
#pragma unroll
      for (int I = 0; I < TS; ++I) k.Q4[I] = cp[I];
      if (has_f && t < T - 1) {
        const f4v *fp = reinterpret_cast<const f4v *>(slot + kSlotf);
#pragma unroll
        for (int I = 0; I < TX; ++I) {
          const f4v v = fp[I];
          k.Fc[4 * I] = v[0]; k.Fc[4 * I + 1] = v[1]; k.Fc[4 * I + 2] = v[2]; k.Fc[4 * I + 3] = v[3];
        }
      }
    }
    if constexpr (MASKED) {
      const size_t tb = (size_t)t * B + b;
      unsigned bits = 0;
#pragma unroll
      for (int m = 0; m < NU; ++m) bits |= (a.mask[tb * NU + m] != 0 ? 1u : 0u) << m;
      k.act = __builtin_amdgcn_readfirstlane(bits);
    }
  };
  auto dma_issue_pad = [&](int t) {
    const size_t tb = (size_t)t * B + b;
    dma_run_floats(a.C + tb * ns * ns, slot_addr + kPadC * 4, ns * ns, lane);
    dma_run_floats(a.c + tb * ns, slot_addr + kPadc * 4, ns, lane);
    if (t < T - 1) {   // uniform
      dma_run_floats(a.F + tb * nx * ns, slot_addr + kPadF * 4, nx * ns, lane);
      if (has_f) dma_run_floats(a.f + tb * nx, slot_addr + kPadf * 4, nx, lane);
    }
  };
  // slot -> bank at the container's positions: a lane's element of logical row li is at base + li * stride, where a column
  // of the problem has (its column, ns), the affine lane (the c / f region, 1), every other lane (the zero word, 0) - the
  // ADDRESS is selected, so every lane executes the same loads
  auto read_bank_pad = [&](int t, Bank &k) {
    const int c_base = lcol >= 0 ? kPadC + lcol : (col_aff ? kPadc : kPadZero);
    const int f_base = lcol >= 0 ? kPadF + lcol : ((col_aff && has_f) ? kPadf : kPadZero);
    int stride = lcol >= 0 ? ns : (col_aff ? 1 : 0), f_stride = lcol >= 0 ? ns : ((col_aff && has_f) ? 1 : 0);
    // (opaque per step: the 72 addresses are loop invariants the compiler would otherwise keep in 72 registers)
    asm volatile("" : "+v"(stride), "+v"(f_stride));
    static_for<0, NS>([&](auto i) {
      const int li = logical(i.value);   // uniform
      float v = (i.value >= NX && lane == i.value) ? 1.f : 0.f;   // (row of an unused control: the unit diagonal)
      if (li >= 0) v = slot[c_base + li * stride];
      k.Q4[i.value / 4][i.value % 4] = v;
    });
    if (t < T - 1) {
      static_for<0, NX>([&](auto r) {
        float v = 0.f;
        if (r.value < nx) v = slot[f_base + r.value * f_stride];   // uniform
        k.Fc[r.value] = v;
      });
    }
    if constexpr (MPC && PAD) {
      if (a.mpc_states != nullptr) {   // the re-centring's row of C for this lane (container positions; 0 outside the problem)
        const int crow_base = lcol >= 0 ? kPadC + lcol * ns : kPadZero;
        int crow_step = lcol >= 0 ? 1 : 0;
        asm volatile("" : "+v"(crow_step));
        static_for<0, NS>([&](auto kk) {
          const int lk = logical(kk.value);   // uniform
          float v = 0.f;
          if (lk >= 0) v = slot[crow_base + lk * crow_step];
          crow_pad[kk.value] = v;
        });
      }
    }
    if constexpr (MASKED) {
      const size_t tb = (size_t)t * B + b;
      unsigned bits = 0;
#pragma unroll
      for (int m = 0; m < NU; ++m) bits |= (m < nu && a.mask[tb * nu + (m < nu ? m : 0)] != 0 ? 1u : 0u) << m;
      k.act = __builtin_amdgcn_readfirstlane(bits);
    }
  };
#endif
  f4v V4[TX];  // rows of [V | v]
#pragma unroll
  for (int I = 0; I < TX; ++I) V4[I] = f4v{0.f, 0.f, 0.f, 0.f};
#ifdef DMPC_WAVE_TIMING   // scripts/microbench/wave_phases.hip: s_memtime stamps, wave 0 reports through a.x
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define DMPC_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define DMPC_STAMP(i) do { } while (0)
#endif

  // ---- MPC: the box QP and what follows it (replaces the gain solve and the value update of the LQR sweep)
  float kprev[NU];
#pragma unroll
  for (int m = 0; m < NU; ++m) kprev[m] = 0.f;
  int n_qp_total = 0;
  auto mpc_gains = [&](int t, size_t tb, f4v(&Q4)[TS]) {
    float Kr[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kr[m] = Q4[(NX + m) / 4][(NX + m) % 4];
    // every lane gets Quu, qu and the QP's bounds                                            mpc_step.py:119-138
    float H[NU][NU], qu[NU], lo[NU], hi[NU], kt[NU];
    static_for<0, NU>([&](auto l) {
#pragma unroll
      for (int m = 0; m < NU; ++m) H[m][l.value] = G64::template bcast<NX + l.value>(Kr[m]);
    });
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      qu[m] = G64::template bcast<AFF>(Kr[m]);
      const bool real = !PAD || m < nu;    // an unused control of a container: qu = 0, unit diagonal, the box [-1, 1] around 0
      const int mc = real ? m : 0;
      const float uc = a.mpc_controls[tb * nu + mc];
      lo[m] = real ? a.mpc_lower[tb * nu + mc] - uc : -1.f;
      hi[m] = real ? a.mpc_upper[tb * nu + mc] - uc : 1.f;
      kt[m] = kprev[m];
    }
    PnqpResult<NU> qp;
    pnqp_solve_rows<NU, /*UNIFORM=*/true>(H, qu, lo, hi, kt, /*warm=*/t != T - 1, a.mpc_n_qp_iter, qp);   // :141-146
    n_qp_total += 1 + qp.it;
    if (!qp.converged) info_bits |= 4;
#pragma unroll
    for (int m = 0; m < NU; ++m) kprev[m] = kt[m];
    // K_t = -LU_free^-1 Qux, rows of clamped controls zero; the affine column carries k_t       :147-157
    float Kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = qp.free_[m] ? Kr[m] : 0.f;
    lu_solve_rinv<NU, /*UNIFORM=*/true>(qp.fac, qp.piv, qp.rinv, Kt);   // (the pivots are the QP's: uniform)
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = col_aff ? kt[m] : -Kt[m];
    if (k_lane && live && (!PAD || col_aff || lane < nx)) {
      float *kp = col_aff ? ks + tb * nu : Ks + tb * nu * nx + lane;
      const int kstride = col_aff ? 1 : nx;
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        if (PAD && m >= nu) break;   // uniform
        kp[m * kstride] = Kt[m];
      }
    }
    if (t > 0) {   // V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~), unmasked blocks                  :165-166
#pragma unroll
      for (int I = 0; I < TX; ++I) V4[I] = Q4[I];
      f4v XU4[TU];
      static_for<0, TU>([&](auto q) { XU4[q.value] = reinterpret_cast<const f4v *>(xu_lds + (lane < NX ? lane : 0) * NU)[q.value]; });
      static_for<0, NU>([&](auto m) {
        const float xm = XU4[m.value / 4][m.value % 4];
        static_for<0, TX>([&](auto I) { V4[I.value] = mfma_bcast<I.value>(xm, Kt[m.value], V4[I.value]); });
      });
      float R[NU];
      static_for<0, NU>([&](auto m) {
        R[m.value] = Kr[m.value];
        static_for<0, NU>([&](auto l) { R[m.value] = fmaf(H[m.value][l.value], Kt[l.value], R[m.value]); });
      });
      static_for<0, NU>([&](auto m) {
        static_for<0, TX>([&](auto I) { V4[I.value] = mfma_bcast<I.value>(Kt[m.value], R[m.value], V4[I.value]); });
      });
    }
  };

  auto step = [&](int t, Bank &k, Bank &next) {
    const size_t tb = (size_t)t * B + b;
    f4v(&Q4)[TS] = k.Q4;
    if constexpr (MPC) {
      if (a.mpc_states != nullptr) {   // c_hat = C [x_t; u_t] + c: lane i forms row i's sum, the affine lane of row i takes it   :305-317
        float tau = 0.f;
        if (lane < NX) { if (lane < nx) tau = a.mpc_states[tb * nx + lane]; }
        else if (lane < NS) { if (lane - NX < nu) tau = a.mpc_controls[tb * nu + (lane - NX)]; }
        float y = 0.f;
        static_for<0, NS>([&](auto kk) {
          const float ck = PAD ? crow_pad[PAD ? kk.value : 0] : crow[PAD ? 0 : kk.value / 4][kk.value % 4];
          y = fmaf(ck, G64::template bcast<kk.value>(tau), y);
        });
        static_for<0, NS>([&](auto i) {
          const float yi = G64::template bcast<i.value>(y);
          Q4[i.value / 4][i.value % 4] += col_aff ? yi : 0.f;
        });
      }
    }
    if (t < T - 1) {
      // ---- G = [V|v]^T F~ : tiles 0..TX-1 (x columns of V) and TA (row 0 = v^T F~)
      f4v G4[TX + 1];
#pragma unroll
      for (int I = 0; I <= TX; ++I) G4[I] = f4v{0.f, 0.f, 0.f, 0.f};
      static_for<0, NX>([&](auto a_) {
        const float va = V4[a_.value / 4][a_.value % 4];
        const float fa = k.Fc[a_.value];
        static_for<0, TX>([&](auto I) { G4[I.value] = mfma_bcast<I.value>(va, fa, G4[I.value]); });
        G4[TX] = mfma_bcast<TA>(va, fa, G4[TX]);
      });
      DMPC_STAMP(2);
      if constexpr (DMPC_WAVE_PREFETCH) fetch_dyn(t - 1, next);
      DMPC_STAMP(1);
      // ---- Q~ += G^T F~ + g1 (x) e_aff
      static_for<0, NX>([&](auto b_) {
        const float gb = G4[b_.value / 4][b_.value % 4];
        const float fb = k.Fc[b_.value];
        static_for<0, TS>([&](auto I) { Q4[I.value] = mfma_bcast<I.value>(gb, fb, Q4[I.value]); });
      });
      {
        const float g1 = G4[TX][0];
        static_for<0, TS>([&](auto I) { Q4[I.value] = mfma_bcast<I.value>(g1, eaff, Q4[I.value]); });
      }
    }
    DMPC_STAMP(3);
    // ---- Qxu as a row matrix for the value update (XU[m]: lane i = Q[i][nx+m]): the control lanes write their column
    // of the finished x rows to LDS, lane i reads row i back after the gain solve - a transpose through 1 KB of LDS per
    // wavefront instead of 64 more MFMAs per step accumulating (F~^T G)[u rows] (round 2, first version)
    if (t > 0 && lane >= NX && lane < NS) {
      static_for<0, NX>([&](auto i) { xu_lds[i.value * NU + (lane - NX)] = Q4[i.value / 4][i.value % 4]; });
    }
    if constexpr (MPC) {
      mpc_gains(t, tb, Q4);
      return;
    }
    // ---- gains (:112-120): Gauss-Jordan on the rows of [Qux | Quu | qu] where they lie.  Row m is one register
    // across the lanes; the multiplier of row i at pivot k is ONE lane of it (lane nx+k).
    float Kr[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kr[m] = Q4[(NX + m) / 4][(NX + m) % 4];
    if constexpr (MASKED) {   // active_constrained_lqr.py:110-126: clamped controls leave the system
      const unsigned act = k.act;
      const bool act_col = lane >= NX && lane < NS && ((act >> (lane - NX)) & 1u);
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        const bool am = (act >> m) & 1u;                                   // uniform
        const float free_row = act_col ? 0.f : Kr[m];                      // Quu_ off the free x free block = 0
        Kr[m] = am ? ((lane == NX + m) ? 1e-8f : 0.f) : free_row;          // active: 1e-8 on the diagonal, qu_ = Qux_ = 0
      }
    }
    static_for<0, NU>([&](auto kc) {
      constexpr int kk = kc.value;
      float p = G64::template bcast<NX + kk>(Kr[kk]);
      float li[NU];
      float mx = 0.f;
#pragma unroll
      for (int i = kk + 1; i < NU; ++i) {
        li[i] = G64::template bcast<NX + kk>(Kr[i]);
        mx = fmaxf(mx, fabsf(li[i]));
      }
      if (__builtin_expect(mx > fabsf(p), 0)) {   // uniform, rare: LAPACK's row interchange (first largest entry)
        float best = fabsf(p);
        int pr = kk;
#pragma unroll
        for (int i = kk + 1; i < NU; ++i) {
          const bool gt = fabsf(li[i]) > best;
          best = gt ? fabsf(li[i]) : best;
          pr = gt ? i : pr;
        }
        pr = __builtin_amdgcn_readfirstlane(pr);
#pragma unroll
        for (int i = kk + 1; i < NU; ++i) {
          if (pr == i) {
            const float tmp = Kr[kk];
            Kr[kk] = Kr[i];
            Kr[i] = tmp;
            li[i] = p;
            p = G64::template bcast<NX + kk>(Kr[kk]);
          }
        }
      }
      if (p == 0.f) info_bits |= 1;
      const float r = fast_rcp(p);
      Kr[kk] *= r;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (i == kk) continue;
        const float l = i > kk ? li[i] : G64::template bcast<NX + kk>(Kr[i]);
        Kr[i] = fmaf(-l, Kr[kk], Kr[i]);
      }
    });
    float Kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = -Kr[m];
    if (k_lane && live && (!PAD || col_aff || lane < nx)) {
      float *kp = col_aff ? ks + tb * nu : Ks + tb * nu * nx + lane;
      const int kstride = col_aff ? 1 : nx;
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        if (PAD && m >= nu) break;   // uniform
        kp[m * kstride] = Kt[m];
      }
    }
    DMPC_STAMP(4);
    if (t > 0) {
      // ---- value update (:151-152): V~ = Q~x. + Qxu K~ (+ K~^T (Q~u. + Quu K~) when the gains are masked)
#pragma unroll
      for (int I = 0; I < TX; ++I) V4[I] = Q4[I];
      f4v XU4[TU];
      static_for<0, TU>([&](auto q) { XU4[q.value] = reinterpret_cast<const f4v *>(xu_lds + (lane < NX ? lane : 0) * NU)[q.value]; });
      static_for<0, NU>([&](auto m) {  // Qxu K~ = XU^T K~, the A^T B shape
        const float xm = XU4[m.value / 4][m.value % 4];
        static_for<0, TX>([&](auto I) { V4[I.value] = mfma_bcast<I.value>(xm, Kt[m.value], V4[I.value]); });
      });
      if constexpr (MASKED) {
        float R[NU];
        static_for<0, NU>([&](auto m) {
          R[m.value] = Q4[(NX + m.value) / 4][(NX + m.value) % 4];
          static_for<0, NU>([&](auto l) {
            const float quu = G64::template bcast<NX + l.value>(Q4[(NX + m.value) / 4][(NX + m.value) % 4]);
            R[m.value] = fmaf(quu, Kt[l.value], R[m.value]);
          });
        });
        static_for<0, NU>([&](auto m) {  // K~^T R
          static_for<0, TX>([&](auto I) { V4[I.value] = mfma_bcast<I.value>(Kt[m.value], R[m.value], V4[I.value]); });
        });
      }
    }
  };

#if !DMPC_WAVE_PREFETCH
  if constexpr (PAD) {
    Bank kp;
#if DMPC_WAVE_DMA
#pragma unroll
    for (int r = 0; r < NX; ++r) kp.Fc[r] = 0.f;
    kp.act = 0;
    if (lane == 0) slot[kPadZero] = 0.f;
    dma_issue_pad(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the slot has landed (requested a whole step ago)
      read_bank_pad(t, kp);
      if (t > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and has been read: it can take the next step's inputs
        dma_issue_pad(t - 1);
      }
      step(t, kp, kp);
    }
#else
    for (int t = T - 1; t >= 0; --t) {
      fetch_cost(t, kp);
      fetch_dyn(t, kp);
      step(t, kp, kp);
    }
#endif
    if constexpr (MPC) {
      if (live && lane == 0) a.mpc_n_qp_total[b] = n_qp_total;
    }
    if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
    return;
  }
#endif
#if DMPC_WAVE_DMA && !DMPC_WAVE_PREFETCH
  Bank ka;
#pragma unroll
  for (int I = 0; I < TS; ++I) ka.Q4[I] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < NX; ++r) ka.Fc[r] = 0.f;
  ka.act = 0;
  dma_issue(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    // The slot has landed (requested a whole step ago).  The vector-memory counter retires in order and the only operations
    // behind the slot's DMA are the NU gain stores of the step in between: waiting for all but NU leaves their
    // acknowledgements (a trip to HBM each, right before this point) out of the critical path.
    if (DMPC_WAVE_COUNTED_WAIT && !MASKED && t < T - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DMPC_STAMP(0);
    read_bank(t, ka);
    if (t > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and has been read: it can take the next step's inputs
#ifndef DMPC_WAVE_KNOB_NO_FETCH   // timing knob: the sweep on whatever the slot holds (wrong results)
      dma_issue(t - 1);
#endif
    }
    DMPC_STAMP(1);
    step(t, ka, ka);
  }
#elif !DMPC_WAVE_PREFETCH
  Bank ka;
  for (int t = T - 1; t >= 0; --t) {
    fetch_cost(t, ka);
    fetch_dyn(t, ka);
    DMPC_STAMP(1);
    step(t, ka, ka);
  }
#else
  Bank ka, kb;
  fetch_cost(T - 1, ka);
  for (int t = T - 1; t >= 0; t -= 2) {
    DMPC_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // bank A has landed (it was requested a whole step ago) ...
    DMPC_STAMP(0);
    fetch_cost(t - 1, kb);                             // ... so the next requests queue behind nothing
    if (t == T - 1) fetch_dyn(t - 1, kb);              // the last step has no first product to hide behind
    DMPC_STAMP(1);
    step(t, ka, kb);
    if (t - 1 >= 0) {
      DMPC_STAMP(5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DMPC_STAMP(0);
      fetch_cost(t - 2, ka);
      DMPC_STAMP(1);
      step(t - 1, kb, ka);
    }
  }
#endif
#ifdef DMPC_WAVE_TIMING
  DMPC_STAMP(5);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long *out = reinterpret_cast<unsigned long long *>(a.x);
    unsigned long long tot = 0;
    for (int i = 0; i < 6; ++i) { out[i] = tacc[i]; tot += tacc[i]; }
    out[6] = tot;
  }
#endif
#ifdef DMPC_WAVE_KNOB_NO_ROLLOUT
  if constexpr (false) {
#else
  if constexpr (ROLLOUT) {
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gain stores of this wavefront have reached L2
#ifdef DMPC_DBG_FENCE
    __threadfence();
#endif
    wave_rollout<NX, NU, MASKED>(a, b, lane, live, Ks, ks, info_bits);
  }
  if constexpr (MPC) {
    if (live && lane == 0) a.mpc_n_qp_total[b] = n_qp_total;
  }
  if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
