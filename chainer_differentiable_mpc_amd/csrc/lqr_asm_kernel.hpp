// lqr_asm_kernel.hpp - fused LQR solve whose whole body is one generated gfx950 instruction stream
// (lqr_asm_gen.hpp, written by gen_lqr_asm.py).  This file is the C++ side: it lays out LDS, prepares the
// per-lane operands of the stream (DMA source pointers and strides, LDS read addresses, store pointers) and
// turns the stream's two outputs into the per-trajectory info flags.
//
// Arithmetic, layout and semantics are those of lqr_kernel / lqr_dma_kernel (lqr/lqr_recursion.py:69-209):
// a wavefront owns four consecutive trajectories, one 16-lane DPP row each, column-per-lane registers, the
// affine terms in lane ns.  What differs is that nothing is left to the compiler between the first DMA and
// the last store - with one wavefront per SIMD the kernel's time is its instruction count.
//
// Two variants of the forward sweep.  Ring: F and f are fetched a second time by LDS-DMA (any T whose gains fit
// in LDS).  Stash (T <= NSTASH): while the backward sweep has F_t in its ring slot it also parks the block in
// accumulation registers (AGPRs, 5 per step at (8,2) - the one on-chip store large enough: 251 KB per CU) in the half-row layout the
// forward sweep consumes (lane i: columns [0, H) of row i, lane 8+i: the rest, brought over by a DPP row rotation), f is
// fetched once more into LDS by a handful of DMAs in the prologue, and the forward sweep reads no F from memory at
// all - it is bandwidth bound otherwise (70 MB at the headline shape, a third of the kernel's traffic).
//
// LDS (dynamic), per 256-thread workgroup:
//   [0, 4*RING)                    one ring per wave: DB backward slots [C|c|F|f] of the wave's 4 trajectories,
//                                  later DF forward slots [F|f] (ring variant only)
//   [.., + 4*FAREA)                stash only: f of all timesteps of the wave's 4 trajectories
//   [.., + 16*T*NU*KROW*4)         gain rows [K_m | 0 | k_m | pad] per trajectory, time-major, zero-initialised
#pragma once
#include "api_util.hpp"
#include "colwise.hpp"
#include "lqr_asm_gen.hpp"
#include "lqr_dma_kernel.hpp"
#include "lqr_kernels.hpp"

namespace dmpc {

// What the f lanes of the DMA groups fetch when the caller passes no f: zeros from the code object's own data
// segment (the library allocates nothing), so that the stream itself has no "f is absent" case.
__device__ const float4 dmpc_zero_chunks[16] = {};

// Bytes of a wave's gain rows + (saving solve) the staging area of the saved blocks behind them: the stream zero-fills
// the gain rows in whole 1 KB pieces, and what that overshoots may be the staging area (scratch until the first step).
template <int NX, int NU, bool STASH, bool SAVE = false, bool GHBM = false>
constexpr size_t lqr_asm_gain_bytes(int T) {
  using G = LqrAsm<NX, NU, false, STASH>;
  if constexpr (GHBM) return 0;   // the gain rows live in the caller's workspace
  const size_t rows = (size_t)4 * T * NU * G::KROW * 4, filled = round_up(rows, 1024);
  if constexpr (SAVE) {
    const size_t both = round_up(rows, 16) + (size_t)LqrAsm<NX, NU, true, STASH, false, false, true>::SAVE_STAGE_BYTES;
    return both > filled ? both : filled;
  } else {
    return filled;
  }
}

template <int NX, int NU, bool STASH, bool SAVE = false, bool GHBM = false>
constexpr size_t lqr_asm_lds_bytes(int T) {
  using G = LqrAsm<NX, NU, false, STASH>;
  return (size_t)4 * G::RING_BYTES + (STASH ? (size_t)4 * G::FAREA_BYTES : 0) +
         4 * lqr_asm_gain_bytes<NX, NU, STASH, SAVE, GHBM>(T);
}

// Per-lane LDS-DMA sources of the backward groups: chunk g = q*64 + lane64 of the slot [C | c | F | f | padding]
// (shared by the LQR kernels and the MPC backward kernel of mpc_asm_kernel.hpp).
template <int NX, int NU, class G, bool HAS_F>
__device__ __forceinline__ void lqr_asm_backward_sources(LqrAsmIn<NX, NU> &in, const float *C, const float *c, const float *F,
                                                         const float *f, int T, size_t B, int b0, int lane64,
                                                         const float *c_u = nullptr) {
  constexpr int NS = NX + NU;
  constexpr int nC = NS * NS, nc = NS, nF = NX * NS, nf = NX;  // 16-byte chunks per wave-step (4 trajectories)
  const char *Cb = reinterpret_cast<const char *>(C), *cb = reinterpret_cast<const char *>(c);
  const char *Fb = reinterpret_cast<const char *>(F);
  const char *fb = HAS_F ? reinterpret_cast<const char *>(f) : reinterpret_cast<const char *>(dmpc_zero_chunks);
  constexpr size_t per_f = HAS_F ? (size_t)NX * 4 : 0;  // bytes of f per trajectory and timestep (0: the zero chunks)
  static_assert(nf <= 16, "dmpc_zero_chunks too small");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= G::NDB) {
      in.ptr[q] = in.str1[q] = in.str[q] = 0;
      continue;
    }
    // which array does chunk g of the slot [C | c | F | f | padding] belong to?  (selects, no branches: this
    // set-up runs once per wave but sits on the critical path of the first DMA)
    const int g = q * 64 + lane64;
    const bool isC = g < nC, isc = !isC && g < nC + nc, isF = !isC && !isc && g < nC + nc + nF;
    const bool isf = !isC && !isc && !isF && g < nC + nc + nF + nf;
    const bool dyn = isF || isf;  // arrays without a slice T-1
    // c in two arrays (c_u != nullptr): the c region is [c_x of the 4 trajectories (nx chunks) | c_u (nu chunks)]
    const bool split = c_u != nullptr, iscu = isc && split && g >= nC + NX;
    const uint64_t base = iscu ? reinterpret_cast<uint64_t>(c_u) : isc ? reinterpret_cast<uint64_t>(cb)
                          : isF ? reinterpret_cast<uint64_t>(Fb) : isf ? reinterpret_cast<uint64_t>(fb)
                          : reinterpret_cast<uint64_t>(Cb);
    const size_t per = iscu ? (size_t)NU * 4 : isc ? (split ? (size_t)NX * 4 : (size_t)NS * 4) : isF ? (size_t)NX * NS * 4
                       : isf ? per_f : (size_t)NS * NS * 4;
    const int g0 = isC ? 0 : iscu ? nC + NX : isc ? nC : isF ? nC + nc : isf ? nC + nc + nF : g;  // padding lanes: chunk 0 of C again
    const size_t off = (size_t)(g - g0) * 16;
    const int t0 = dyn ? T - 2 : T - 1;
    const uint64_t p = base + ((size_t)t0 * B + (size_t)b0) * per + off;
    in.ptr[q] = p - (uint64_t)q * 1024u;  // the instruction offset q*1024 moves the global address as well
    const uint64_t s = (uint64_t)0 - (uint64_t)(B * per);
    in.str[q] = s;
    in.str1[q] = dyn ? 0 : s;  // there is no F_{T-1}: the first group fetches F_{T-2} (unused), as does the second
  }
}

// AFFINE: the slot's C region holds [K_t | Qxu_t | Quu_t] of the wave's four trajectories (nu nx + nx nu + nu nu chunks)
// and nothing else - the remaining lanes of those groups fetch the resident zero chunk with a zero time stride.
template <int NX, int NU, class G>
__device__ __forceinline__ void lqr_asm_affine_sources(LqrAsmIn<NX, NU> &in, const float *Ks, const float *Qxu, const float *Quu,
                                                       const float *c, const float *F, int T, size_t B, int b0, int lane64,
                                                       const float *c_u = nullptr) {
  constexpr int NS = NX + NU;
  constexpr int nC = NS * NS, nc = NS, nF = NX * NS, nf = NX;
  constexpr int nK = NU * NX, nQx = NX * NU, nQu = NU * NU;
  static_assert(nK + nQx + nQu <= nC, "the saved blocks fit the C region");
  const uint64_t zero = reinterpret_cast<uint64_t>(dmpc_zero_chunks);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= G::NDB) {
      in.ptr[q] = in.str1[q] = in.str[q] = 0;
      continue;
    }
    const int g = q * 64 + lane64;
    const bool isK = g < nK, isQx = !isK && g < nK + nQx, isQu = !isK && !isQx && g < nK + nQx + nQu;
    const bool isc = g >= nC && g < nC + nc, isF = g >= nC + nc && g < nC + nc + nF;
    const bool split = c_u != nullptr, iscu = isc && split && g >= nC + NX;   // (as in lqr_asm_backward_sources)
    const uint64_t base = isK ? reinterpret_cast<uint64_t>(Ks) : isQx ? reinterpret_cast<uint64_t>(Qxu)
                          : isQu ? reinterpret_cast<uint64_t>(Quu) : iscu ? reinterpret_cast<uint64_t>(c_u)
                          : isc ? reinterpret_cast<uint64_t>(c) : isF ? reinterpret_cast<uint64_t>(F) : zero;
    const size_t per = isK ? (size_t)NU * NX * 4 : isQx ? (size_t)NX * NU * 4 : isQu ? (size_t)NU * NU * 4
                       : iscu ? (size_t)NU * 4 : isc ? (split ? (size_t)NX * 4 : (size_t)NS * 4) : isF ? (size_t)NX * NS * 4 : 0;
    const int g0 = isK ? 0 : isQx ? nK : isQu ? nK + nQx : iscu ? nC + NX : isc ? nC : isF ? nC + nc : g;   // zero lanes: chunk 0 of the zeros
    const int t0 = isF ? T - 2 : T - 1;
    const uint64_t p = base + ((size_t)t0 * B + (size_t)b0) * per + (size_t)(g - g0) * 16;
    in.ptr[q] = p - (uint64_t)q * 1024u;
    const uint64_t s = (uint64_t)0 - (uint64_t)(B * per);
    in.str[q] = s;
    in.str1[q] = isF ? 0 : s;
  }
}

// ... and its LDS read addresses: aq[0] lane j = c[j]; aq[1] lane i < nx = row i of Qxu; aq[2] Quu (every lane of the row);
// aq[3] lane j < nx = K[0][j] (row m at + m nx floats); af as in the full solve (f = 0: the zero chunks)
template <int NX, int NU, class G>
__device__ __forceinline__ void lqr_asm_affine_addresses(LqrAsmIn<NX, NU> &in, unsigned ring, int r, int lane, bool split_c = false) {
  constexpr int NS = NX + NU;
  static_assert(NS >= 4, "four address operands");
  const int lane_c = lane < NS ? lane : NS - 1;
  const int lane_x = lane < NX ? lane : NX - 1;
  const bool col_aff = lane == NS;
  in.aq[0] = ring + (unsigned)(G::OFF_c + (!split_c ? (r * NS + lane_c) * 4
                                           : lane_c < NX ? (r * NX + lane_c) * 4 : NX * 16 + (r * NU + lane_c - NX) * 4));
  in.aq[1] = ring + (unsigned)(G::OFF_C + NU * NX * 16 + (r * NX + lane_x) * NU * 4);
  in.aq[2] = ring + (unsigned)(G::OFF_C + (NU * NX + NX * NU) * 16 + r * NU * NU * 4);
  in.aq[3] = ring + (unsigned)(G::OFF_C + (r * NU * NX + lane_x) * 4);
#pragma unroll
  for (int i = 4; i < NS; ++i) in.aq[i] = 0;
  const unsigned f0 = ring + (col_aff ? (unsigned)(G::OFF_f + r * NX * 4) : (unsigned)(G::OFF_F + (r * NX * NS + lane_c) * 4));
  const unsigned st = col_aff ? 4u : (unsigned)(NS * 4);
#pragma unroll
  for (int k = 0; k < NX; ++k) in.af[k] = f0 + (unsigned)k * st;
}

// LDS read addresses (ring slot 0) of a lane's rows: lane j < ns walks column j of [C] / [F] (stride ns floats), lane ns
// walks c / f themselves (stride 1)
template <int NX, int NU, class G>
__device__ __forceinline__ void lqr_asm_row_addresses(LqrAsmIn<NX, NU> &in, unsigned ring, int r, int lane, bool split_c = false) {
  constexpr int NS = NX + NU;
  const int lane_c = lane < NS ? lane : NS - 1;  // lanes past the affine column duplicate column ns-1
  const bool col_aff = lane == NS;
  const unsigned q0 = ring + (col_aff ? (unsigned)(G::OFF_c + r * NS * 4) : (unsigned)(G::OFF_C + (r * NS * NS + lane_c) * 4));
  const unsigned f0 = ring + (col_aff ? (unsigned)(G::OFF_f + r * NX * 4) : (unsigned)(G::OFF_F + (r * NX * NS + lane_c) * 4));
  const unsigned st = col_aff ? 4u : (unsigned)(NS * 4);
#pragma unroll
  for (int i = 0; i < NS; ++i) in.aq[i] = q0 + (unsigned)i * st;
  if (split_c && col_aff) {   // c region = [c_x of the 4 trajectories | c_u of the 4 trajectories]
#pragma unroll
    for (int i = 0; i < NS; ++i)
      in.aq[i] = ring + (unsigned)(G::OFF_c + (i < NX ? (r * NX + i) * 4 : NX * 16 + (r * NU + i - NX) * 4));
  }
#pragma unroll
  for (int k = 0; k < NX; ++k) in.af[k] = f0 + (unsigned)k * st;
}

// MASKED: LQR_active (mpc/active_constrained_lqr.py) - a.mask [T,B,nu] uint8 marks the clamped controls; needs
// B * nu to be a multiple of 4 (the flags of a wave's four trajectories are fetched as whole dwords).
// a.x == nullptr: backward sweep only (LqrRecursion.backward(), gains to a.Ks / a.ks - WRITE_K).
// AFFINE: the re-solve from saved gains (a.Ks_in, a.Quu_in, a.Qxu_in; a.c the new affine term, no f, a.C not read).
// ADJ (with AFFINE): DiffLqr.backward in one launch - a.c / a.c_u are grad_x / grad_u, a.Vv_in the saving solve's value
// functions, a.tau_x / a.tau_u its solution; dC, dc, dF, df, dx0 are written by the rollout (a.x, a.u are not).
// GHBM (ring form): the gain rows pass through a.wsK ([T,B,nu,KROW] floats, rows [K_m | 0 | k_m | pad]) instead of LDS and
// come back to the rollout through its ring, with F and f: any horizon.
template <int NX, int NU, bool HAS_F, bool WRITE_K, bool STASH, bool MASKED = false, bool GHBM = false, bool SAVE = false,
          bool AFFINE = false, bool ADJ = false>
__global__ __launch_bounds__(256) void lqr_asm_kernel(const LqrArgs a) {
  using G = LqrAsm<NX, NU, WRITE_K, STASH, MASKED, GHBM, SAVE, AFFINE, ADJ>;
  static_assert(!GHBM || !(STASH || WRITE_K || MASKED || SAVE || AFFINE), "gains through HBM: the plain ring form only");
  static_assert(!ADJ || AFFINE, "the one-pass gradient is a form of the affine re-solve");
  static_assert(!AFFINE || (!HAS_F && STASH), "the affine re-solve has no f and keeps F in the stash");
  static_assert(G::kAvailable, "no generated instruction stream for this shape");
  constexpr int NS = NX + NU, AFF = NS, KROW = G::KROW;
  constexpr int nC = NS * NS, nc = NS, nF = NX * NS, nf = NX;  // 16-byte chunks per wave-step (4 trajectories)
  // (M0 carries the LDS-DMA target; it reaches all 160 KB - scripts/microbench/m0_range.hip)

#ifdef DMPC_ASM_TIMING_GEN
  const unsigned t_entry = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;  // trajectory within the wave
  const int lane = lane64 & 15;
  int b0 = ((int)blockIdx.x * 4 + wave) * 4;  // first trajectory of this wave
  if (b0 > a.B - 4) b0 = a.B - 4;             // last wave overlaps its neighbour instead of running ragged
  b0 = __builtin_amdgcn_readfirstlane(b0);
  const int b = b0 + r;

  extern __shared__ float lds[];
  const unsigned lds0 = lds_byte_address(lds);
  const unsigned ring = lds0 + (unsigned)wave * G::RING_BYTES;
  const unsigned farea = lds0 + 4u * G::RING_BYTES + (unsigned)wave * (STASH ? G::FAREA_BYTES : 0);
  const unsigned gain_wave_bytes = (unsigned)lqr_asm_gain_bytes<NX, NU, STASH, SAVE, GHBM>(T);
  const unsigned gain_wave = lds0 + 4u * G::RING_BYTES + (STASH ? 4u * G::FAREA_BYTES : 0u) + (unsigned)wave * gain_wave_bytes;
  const unsigned gain_traj = gain_wave + (unsigned)r * (unsigned)(T * NU * KROW * 4);

  LqrAsmIn<NX, NU> in;
  in.ring = __builtin_amdgcn_readfirstlane(ring);
  in.T = T;
  in.tf = 0;
  in.bwd_only = (a.x == nullptr && !ADJ) ? 1 : 0;   // (the one-pass gradient has no x, u outputs: its rollout writes dC ...)

  if constexpr (AFFINE) lqr_asm_affine_sources<NX, NU, G>(in, a.Ks_in, a.Qxu_in, a.Quu_in, a.c, a.F, T, B, b0, lane64, a.c_u);
  else lqr_asm_backward_sources<NX, NU, G, HAS_F>(in, a.C, a.c, a.F, a.f, T, B, b0, lane64, a.c_u);
  const char *Fb = reinterpret_cast<const char *>(a.F);
  const char *fb = HAS_F ? reinterpret_cast<const char *>(a.f) : reinterpret_cast<const char *>(dmpc_zero_chunks);
  constexpr size_t per_f = HAS_F ? (size_t)NX * 4 : 0;  // bytes of f per trajectory and timestep (0: the zero chunks)
  if constexpr (MASKED) {  // dword i < nu of the 4 * nu flag bytes of this wave and timestep; the other lanes repeat dword 0
    in.pm = reinterpret_cast<uint64_t>(a.mask) + ((size_t)(T - 1) * B + (size_t)b0) * NU + (size_t)(lane64 < NU ? lane64 : 0) * 4 -
            (uint64_t)G::PADM;  // the instruction offset that places the dwords in the slot padding moves the source too
    in.dm = (uint64_t)0 - (uint64_t)(B * NU);
    in.am = ring + (unsigned)(r * NU);
  } else {
    in.pm = in.dm = 0;
    in.am = 0;
  }
  // The first DB groups leave NOW; everything below (LDS read addresses, store pointers, the forward sweep's
  // operands) is computed while they are in flight.  No memory operation of this C++ code may follow: x_init
  // is loaded and x_0 stored by the stream itself.
  G::issue_first(in);
  in.gz = gain_wave + (unsigned)lane64 * 16u;  // the stream zero-fills the gain rows behind its first DMAs
  in.nz = GHBM ? 0 : (int)(round_up((size_t)4 * T * NU * KROW * 4, 1024) / 1024u);
  if constexpr (GHBM) {   // a whole gain row per trajectory and control: lane j <= ns stores column j (lane ns = the affine column)
    const size_t tb = (size_t)(T - 1) * B + (size_t)b;
    in.pgw = reinterpret_cast<uint64_t>(a.wsK + tb * NU * KROW + (lane <= NS ? lane : NS));
    in.dgw = (uint64_t)0 - (uint64_t)(B * NU * KROW * 4);
  } else {
    in.pgw = in.dgw = 0;
  }
  const int lane_c = lane < NS ? lane : NS - 1;  // lanes past the affine column duplicate column ns-1
  const bool col_aff = lane == AFF;
  if constexpr (AFFINE) lqr_asm_affine_addresses<NX, NU, G>(in, ring, r, lane, a.c_u != nullptr);
  else lqr_asm_row_addresses<NX, NU, G>(in, ring, r, lane, a.c_u != nullptr);
  in.ak = gain_traj + (unsigned)((T - 1) * NU * KROW * 4) + (unsigned)lane * 4u;
  in.eaff = col_aff ? 1.f : 0.f;
  if constexpr (WRITE_K) {
    const size_t tb = (size_t)(T - 1) * B + (size_t)b;
#pragma unroll
    for (int m = 0; m < NU; ++m)
      in.pk[m] = col_aff ? reinterpret_cast<uint64_t>(a.ks + tb * NU + m)
                         : reinterpret_cast<uint64_t>(a.Ks + (tb * NU + m) * NX + (lane < NX ? lane : 0));
    in.dk = (uint64_t)0 - (uint64_t)(col_aff ? B * NU * 4 : B * NU * NX * 4);
  } else {
#pragma unroll
    for (int m = 0; m < NU; ++m) in.pk[m] = 0;
    in.dk = 0;
  }
  if constexpr (SAVE) {
    // [Vv | Qxu | Quu] of the wave's four trajectories are staged behind its gain rows and leave as 16-byte chunks
    constexpr int nVv = NX * (NX + 1), nQx = NX * NU, nQu = NU * NU;
    const unsigned stg = gain_wave + (unsigned)round_up((size_t)4 * T * NU * KROW * 4, 16);
    const int m_ = lane >= NX && lane < NS ? lane - NX : 0;
    in.asv = stg + (unsigned)((r * NX * (NX + 1) + (lane < NX ? lane : NX)) * 4);   // lane ns (the affine column) owns column nx
    in.asq = stg + (unsigned)((r * NX * NU + m_) * 4);
    in.asu = stg + (unsigned)(r * NU * NU * 4);
    in.ach = stg + (unsigned)lane64 * 16u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = q * 64 + lane64;
      const bool isV = g < nVv, isQx = !isV && g < nVv + nQx, isQu = !isV && !isQx && g < nVv + nQx + nQu;
      const uint64_t base = isV ? reinterpret_cast<uint64_t>(a.Vv_out) : isQx ? reinterpret_cast<uint64_t>(a.Qxu_out)
                                                                              : reinterpret_cast<uint64_t>(a.Quu_out);
      const size_t per = isV ? (size_t)NX * (NX + 1) * 4 : isQx ? (size_t)NX * NU * 4 : (size_t)NU * NU * 4;
      const int g0 = isV ? 0 : isQx ? nVv : nVv + nQx;
      const bool any = isV || isQx || isQu;
      in.pso[q] = any ? base + ((size_t)(T - 1) * B + (size_t)b0) * per + (size_t)(g - g0) * 16 : 0;
      in.sso[q] = any ? (uint64_t)0 - (uint64_t)(B * per) : 0;
    }
  } else {
    in.asv = in.asq = in.asu = 0;
  }

  // lane i < nx: row i of [F_t | f_t]; lane nx+m: gain row m; the other lanes shadow the last gain row
  const bool row_x = lane < NX;
  const int m_own = row_x ? 0 : (lane < NS ? lane - NX : NU - 1);
  const unsigned arow_u = gain_traj + (unsigned)(m_own * KROW * 4);
#pragma unroll
  for (int q = 0; q < 2; ++q) in.fptr[q] = in.fstr[q] = in.fstrl[q] = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) in.fp[q] = 0;
  {  // stash: lane i < 8 of a row parks columns [0, H) of row i of F_t, lane 8 + i columns [H, ns)
    const int i8 = (lane & 7) < NX ? (lane & 7) : NX - 1;
    const unsigned half_row = (unsigned)(G::OFF_F + ((r * NX + i8) * NS + (lane >> 3) * G::HROW) * 4);
#pragma unroll
    for (int q = 0; q < 6; ++q) in.sr[q] = q < G::DEPTH_B ? ring + (unsigned)(q * G::SLOT_B) + half_row : 0u;
  }
  in.farea = __builtin_amdgcn_readfirstlane(farea);
  if constexpr (STASH) {
    // f of timestep tt (the wave's 4 trajectories, nx chunks) lives at farea + tt*nx*16: DMA q brings SPD steps
    static_assert(G::NFD <= 8 && 64 % NX == 0, "f area layout");
#pragma unroll
    for (int q = 0; q < G::NFD; ++q) {
      int tt = q * G::SPD + lane64 / NX;
      if (tt > T - 2) tt = T - 2;  // nothing past f_{T-2} exists (or is read)
      in.fp[q] = reinterpret_cast<uint64_t>(fb) + ((size_t)tt * B + (size_t)b0) * per_f + (size_t)(lane64 % NX) * 16;
    }
    // F comes out of the stash registers; LDS serves the gain rows (lanes nx..15 only - lanes < nx carry an address
    // they never use) and f_t / k_t
    in.arow = arow_u;
    in.aaff = row_x ? farea + (unsigned)((r * NX + lane) * 4) : arow_u + (unsigned)(NS * 4);
    in.drow = (unsigned)(NU * KROW * 4);
    in.drow2 = in.drow;
    in.daff = row_x ? (unsigned)(NX * 16) : (unsigned)(NU * KROW * 4);
    in.daff2 = in.daff;
  } else {
    // ---- forward DMA: chunk g of the slot [F | f]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q >= G::NDF) continue;
      const int g = q * 64 + lane64;
      const char *base = Fb;
      size_t per = (size_t)NX * NS * 4, off = 0;
      if (g < nF) {
        off = (size_t)g * 16;
      } else if (g < nF + nf) {
        base = fb; per = per_f; off = (size_t)(g - nF) * 16;
      } else if (GHBM && g < nF + nf + NU * KROW) {   // the wave's gain rows of the step: 4 trajectories x nu x KROW floats
        base = reinterpret_cast<const char *>(a.wsK); per = (size_t)NU * KROW * 4; off = (size_t)(g - nF - nf) * 16;
      }
      in.fptr[q] = reinterpret_cast<uint64_t>(base) + (size_t)b0 * per + off - (uint64_t)q * 1024u;
      in.fstr[q] = (uint64_t)(B * per);
      in.fstrl[q] = (GHBM && g >= nF + nf && g < nF + nf + NU * KROW) ? (uint64_t)(B * per) : 0;
    }
    const unsigned arow_g = GHBM ? ring + (unsigned)(G::FOFF_G + (r * NU + m_own) * KROW * 4) : arow_u;   // GHBM: in the slot
    in.arow = row_x ? ring + (unsigned)((r * NX + lane) * NS * 4) : arow_g;
    in.aaff = row_x ? ring + (unsigned)(G::FOFF_f + (r * NX + lane) * 4) : arow_g + (unsigned)(NS * 4);
    in.drow = (row_x || GHBM) ? (unsigned)G::SLOT_F : (unsigned)(NU * KROW * 4);
    in.drow2 = (row_x || GHBM) ? (unsigned)G::SLOT_F - (unsigned)(G::DEPTH_F * G::SLOT_F) : (unsigned)(NU * KROW * 4);
    in.daff = in.drow;
    in.daff2 = in.drow2;
  }
  in.pst = row_x ? reinterpret_cast<uint64_t>(a.x + (B + (size_t)b) * NX + lane)
                 : reinterpret_cast<uint64_t>(a.u + (size_t)b * NU + m_own);
  in.dst = row_x ? (uint64_t)(B * NX * 4) : (uint64_t)(B * NU * 4);
  in.pxi = a.x_init != nullptr ? reinterpret_cast<uint64_t>(a.x_init + (size_t)b * NX + (row_x ? lane : NX - 1))
                               : reinterpret_cast<uint64_t>(dmpc_zero_chunks);  // x_init = 0 (backward only: never used)
  in.px0 = reinterpret_cast<uint64_t>(a.x + (size_t)b * NX + (row_x ? lane : NX - 1));
  if constexpr (ADJ) {
    constexpr int nVv = NX * (NX + 1), nOutC = NS * NS, nOutc = NS, nOutF = NX * NS, nOutf = NX;   // 16-byte chunks per wave-step
    const int lane_x = row_x ? lane : NX - 1;
    in.px0 = reinterpret_cast<uint64_t>(a.dx0 + (size_t)b * NX + lane_x);     // dx_init = d_lambda_0 = v'_0
    in.avp = farea + (unsigned)((T - 1) * NX * 16 + (r * NX + lane_x) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) in.avr[q] = ring + (unsigned)(q * G::ADJ_SLOT + (r * NX + lane_x) * (NX + 1) * 4);
    in.atx = ring + (lane_c < NX ? (unsigned)(nVv * 16 + (r * NX + lane_c) * 4)
                                  : (unsigned)((nVv + NX) * 16 + (r * NU + lane_c - NX) * 4));
    in.awc = ring + (unsigned)((r * NS * NS + lane_c) * 4);
    in.awe = ring + (unsigned)((r * NS + lane_c) * 4);
    in.awf = ring + (unsigned)((r * NX * NS + lane_c) * 4);
    in.awd = ring + (unsigned)((r * NX + lane_x) * 4);
    in.ach = ring + (unsigned)lane64 * 16u;
    in.wa = a.w_a;
    in.wb = a.w_b;
    in.dfshift = a.df_shift;
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // DMA sources of chunk g of [Vv | x | u] at t = 0; padding lanes repeat chunk 0
      if (q >= G::ADJ_NDA) continue;
      const int g = q * 64 + lane64;
      const bool isV = g < nVv, isX = !isV && g < nVv + NX, isU = !isV && !isX && g < nVv + NX + NU;
      const uint64_t base = isX ? reinterpret_cast<uint64_t>(a.tau_x) : isU ? reinterpret_cast<uint64_t>(a.tau_u)
                                                                           : reinterpret_cast<uint64_t>(a.Vv_in);
      const size_t per = isX ? (size_t)NX * 4 : isU ? (size_t)NU * 4 : (size_t)NX * (NX + 1) * 4;
      const int g0 = isV ? 0 : isX ? nVv : isU ? nVv + NX : g;
      in.fptr[q] = base + (size_t)b0 * per + (size_t)(g - g0) * 16 - (uint64_t)q * 1024u;
      in.fstr[q] = (uint64_t)(B * per);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // store pointers of chunk g of [dC | dc | dF | df] at t = 0 (the stream masks the rest)
      const int g = q * 64 + lane64;
      const bool isC = g < nOutC, isc = !isC && g < nOutC + nOutc, isF = !isC && !isc && g < nOutC + nOutc + nOutF;
      const bool isf = !isC && !isc && !isF && g < nOutC + nOutc + nOutF + nOutf;
      const uint64_t base = isC ? reinterpret_cast<uint64_t>(a.dC) : isc ? reinterpret_cast<uint64_t>(a.dc)
                            : isF ? reinterpret_cast<uint64_t>(a.dF) : reinterpret_cast<uint64_t>(a.df);
      const size_t per = isC ? (size_t)NS * NS * 4 : isc ? (size_t)NS * 4 : isF ? (size_t)NX * NS * 4 : (size_t)NX * 4;
      const int g0 = isC ? 0 : isc ? nOutC : isF ? nOutC + nOutc : nOutC + nOutc + nOutF;
      const bool any = isC || isc || isF || isf;
      in.pso[q] = any ? base + (size_t)b0 * per + (size_t)(g - g0) * 16 : 0;
      in.sso[q] = any ? (uint64_t)(B * per) : 0;
    }
  } else {
    in.avp = in.atx = in.awc = in.awe = in.awf = in.awd = 0;
    if constexpr (!SAVE) in.ach = 0;
    in.wa = in.wb = 0.f;
    in.dfshift = 0;
  }

  float xvout, minpiv;
  G::run(in, xvout, minpiv);

#ifdef DMPC_ASM_TIMING_GEN  // timing builds: info[4*w + i] = cycles from the stream's start to phase boundary i
  if (a.info != nullptr && lane64 == 0) {
    const int w = (int)blockIdx.x * 4 + wave;
    if (4 * w + 3 < a.B) {
      a.info[4 * w + 0] = (int)(in.ts[0] - t_entry);  // operand set-up before the stream
      for (int i = 1; i < 4; ++i) a.info[4 * w + i] = (int)(in.ts[i] - in.ts[0]);
    }
  }
  return;
#endif
  if (a.info != nullptr) {
    int bits = 0;
    if (minpiv == 0.f) bits |= 1;                       // a zero pivot in some Quu (uniform over the row)
    if (lane < NS && !is_finite(xvout)) bits |= 2;      // NaN/Inf propagate to u_{T-1} through the recursion
    if (a.x == nullptr && !ADJ && !is_finite(minpiv)) bits |= 2;  // backward only: a NaN pivot is all there is to see
    if (a.info_store) {   // (uniform) one lane per trajectory writes the row's flags, zero included
      const unsigned long long nonfinite = __ballot((bits & 2) != 0);
      const int row_bits = (bits & 1) | ((((nonfinite >> (16 * r)) & 0xffffull) != 0) ? 2 : 0);
      if (lane == 0) a.info[b] = row_bits;
    } else if (bits != 0) {
      atomicOr(&a.info[b], bits);
    }
  }
}

}  // namespace dmpc
