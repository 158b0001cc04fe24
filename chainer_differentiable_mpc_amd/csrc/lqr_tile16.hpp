// lqr_tile16.hpp - the backward Riccati sweep of the large shapes ((32,8) of BASELINE.json configs[4] and its smaller
// instances) on v_mfma_f32_16x16x4_f32: one wavefront per trajectory, every matrix of a step held as 16 x 16 tiles in the
// matrix cores' own result layout.  Round 5; replaces the 4x4x1 outer-product sweep of lqr_wave_mfma.hpp for the plain solve
// (668 matrix instructions and ~1,060 vector instructions per step there, 144 + ~400 here).
//
// Recursion: LqrRecursion.backward, lqr/lqr_recursion.py:69-158; the rollout that follows in the same launch is
// wave_rollout of lqr_wave_mfma.hpp (lqr/lqr_recursion.py:160-200).
//
// TILE LAYOUT.  A matrix M is cut into 16 x 16 tiles; a tile is four registers: register r of lane l = 16 g + j holds
// M[4 g + r][j] of the tile - exactly what v_mfma_f32_16x16x4_f32 leaves in its result registers.  The instruction computes
// D[i][j] += sum_k A[i][k] B[k][j] with A[i][k] read from lane 16 k + i and B[k][j] from lane 16 k + j, so with register r of a
// tile of X as A and register r of a tile of Y as B it adds  sum_g X[4 g + r][i] Y[4 g + r][j] : four instructions (r = 0..3)
// contract the 16 ROWS of the two tiles, i.e. a tile of X^T Y - and the result is again in the tile layout.  Every product of
// the sweep is taken in that shape, so nothing is ever re-laid out between products:
//     G   = V^T F~  (+ v in the affine column)       lqr_recursion.py:89    (F^T V)^T, the reference's own association
//     Q~  = C~ + F~^T G                              lqr_recursion.py:85-96
//     V~' = Q~x. + Qxu K~                            lqr_recursion.py:151-152 (see below for the other two terms)
// The affine terms ride along as column ns of [C | c], [F | f], [Q | q], [K | k] and as the column ns of [V | v] (kept in
// its own registers `Vaff`: lanes j == ns % 16 of the tile column ns / 16).
//
// INPUTS.  The step's [C_t | c_t | F_t | f_t] come HBM -> LDS by per-lane gather LDS-DMA a step ahead, in the order the reads
// want them ([r][g][j]: a 16-byte chunk = four consecutive columns of one row, 64 chunks = one full tile), so a tile register is
// one conflict-free ds_read of consecutive floats and the bytes fetched are exactly the arrays' own (PackedImage below: a
// partial tile keeps only the rows and columns the matrix has).  11.8 KB of image + 1.5 KB of scratch per wavefront: three
// wavefronts per SIMD (fp32 MFMA and the vector ALU exclude each other on a SIMD - measured: two wavefronts overlap only 15 %
// - so the LDS, DMA-issue and scalar latencies of one wavefront need the others' arithmetic to hide behind).
//
// GAINS (lqr_recursion.py:112-120).  The control rows [Qux | Quu | qu] go through 1.5 KB of LDS into the column-per-lane
// layout (a row = one register across the lanes), where the Gauss-Jordan elimination of lqr_wave_mfma.hpp runs unchanged
// (LAPACK's pivot order); K~ returns through the same LDS rows in a COMPACT contraction layout (register r2 of lane 16 g + j
// = K~[4 r2 + g][j]), Qxu likewise (transposed through the same area), so Qxu K~ costs nu / 4 instructions per tile instead of four.
//
// As in lqr_wave_mfma.hpp the term K~^T (Q~u. + Quu K~) of lqr_recursion.py:151-152 multiplies the residual of the gain solve
// and is left out (measured against the float64 kernels on all 8,192 trajectories of a shard: tests/test_f64_gpu.py).
#pragma once
#include "lqr_wave_mfma.hpp"

namespace dmpc {

// Between an LDS write and the read of the same words by OTHER lanes of the wavefront: the hardware executes a wavefront's LDS
// operations in order, but the compiler sees one thread - without this it may hand a lane the value that lane itself stored
// to the address earlier (store-to-load forwarding; seen in the float64 kernel: the rows of lane group 3 came back as the control
// rows the group had written before the gain solve, not as the gains other lanes wrote after it).
__device__ __forceinline__ void lds_lanes_exchange() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ f4v mfma16(float a, float b, f4v c) {
#ifdef DMPC_T16_KNOB_NOMFMA   // timing knob (wrong results): the sweep without its matrix instructions
  c[0] += a * b;
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// NT: the non-temporal cache policy (C_t is read exactly once; F_t and the gains are read again by the rollout)
template <bool NT>
__device__ __forceinline__ void tile16_dma_full(unsigned voff, unsigned long long base) {
  if constexpr (NT) asm volatile("global_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base) : "memory");
  else asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base) : "memory");
}
template <bool NT>
__device__ __forceinline__ void tile16_dma_masked(unsigned voff, unsigned long long base, unsigned long long mask) {
  if constexpr (NT) asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(base), "s"(mask) : "memory");
  else asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(base), "s"(mask) : "memory");
}

// The PACKED LDS image of a ROWS x COLS row-major matrix (ROWS % 4 == 0, COLS % 4 == 0) cut into RT x CT tiles.  A tile keeps
// only what the matrix has: gv row groups (of 4 rows) and cv column chunks (of 4 columns); register r of the tile is the segment
// [g < gv][4 cv columns] - gv * 4 cv consecutive floats -, the four segments of a tile follow one another, the full tiles come
// first (they share one DMA lane pattern), then the others in row-major order.  The image is exactly ROWS * COLS floats: the
// LDS-DMA moves the arrays' own bytes and nothing else.  A 16-byte chunk of the image is four consecutive columns of one row.
// CPC: columns per 16-byte chunk (4 floats; 2 doubles - lqr_tile16_f64.hpp)
template <int ROWS, int COLS, int RT, int CT, int CPC = 4>
struct PackedImage {
  static_assert(ROWS % 4 == 0 && COLS % CPC == 0, "whole row groups and 16-byte chunks");
  static constexpr int kCols = CPC, kChunkCols = 16 / CPC;      // chunk columns of a full tile
  static constexpr int clampv(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
  static constexpr int gv(int ib) { return clampv((ROWS - 16 * ib) / 4, 4); }
  static constexpr int cv(int jb) { return clampv((COLS - 16 * jb) / CPC, kChunkCols); }
  static constexpr bool full(int ib, int jb) { return gv(ib) == 4 && cv(jb) == kChunkCols; }
  static constexpr int chunks(int ib, int jb) { return 4 * gv(ib) * cv(jb); }
  static constexpr int seg_floats(int ib, int jb) { return gv(ib) * CPC * cv(jb); }      // elements of one register's segment
  static constexpr int base_chunk(int ib, int jb) {
    int n = 0;
    const bool f = full(ib, jb);
    for (int i = 0; i < RT; ++i)
      for (int jj = 0; jj < CT; ++jj) {
        const bool fi = full(i, jj), earlier = i * CT + jj < ib * CT + jb;
        if (f ? (fi && earlier) : (fi || earlier)) n += chunks(i, jj);
      }
    return n;
  }
  static constexpr int n_full() {
    int n = 0;
    for (int i = 0; i < RT; ++i)
      for (int jj = 0; jj < CT; ++jj) n += full(i, jj) ? 1 : 0;
    return n;
  }
  static constexpr int total_chunks() { return ROWS * COLS / CPC; }
  static constexpr int tail_instrs() { return (total_chunks() - (256 / CPC) * n_full() + 63) / 64; }
  static constexpr int total_instrs() { return (total_chunks() + 63) / 64; }
  // (ib, jb) of the k-th full tile
  static constexpr int full_tile(int k) {
    int n = 0;
    for (int i = 0; i < RT; ++i)
      for (int jj = 0; jj < CT; ++jj)
        if (full(i, jj)) {
          if (n == k) return i * CT + jj;
          ++n;
        }
    return -1;
  }
  // element offset inside the matrix of image chunk n (a chunk of a partial tile); 0 past the end (those lanes are masked off)
  static __device__ __forceinline__ int tail_src_offset(int n) {
    int off = 0;
    static_for<0, RT * CT>([&](auto tl) {
      constexpr int ib = tl.value / CT, jb = tl.value % CT;
      if constexpr (!full(ib, jb) && chunks(ib, jb) > 0) {
        constexpr int base = base_chunk(ib, jb), cnt = chunks(ib, jb), gvv = gv(ib), cvv = cv(jb);
        if (n >= base && n < base + cnt) {
          const int q = n - base, r = q / (gvv * cvv), gg = (q / cvv) % gvv, cq = q % cvv;
          off = (16 * ib + 4 * gg + r) * COLS + 16 * jb + CPC * cq;
        }
      }
    });
    return off;
  }
  // the same for ANY chunk of the image, full tiles included (the float64 kernel keeps one lane-offset register per instruction)
  static __device__ __forceinline__ int src_offset(int n) {
    int off = 0;
    static_for<0, RT * CT>([&](auto tl) {
      constexpr int ib = tl.value / CT, jb = tl.value % CT;
      if constexpr (chunks(ib, jb) > 0) {
        constexpr int base = base_chunk(ib, jb), cnt = chunks(ib, jb), gvv = gv(ib), cvv = cv(jb);
        if (n >= base && n < base + cnt) {
          const int q = n - base, r = q / (gvv * cvv), gg = (q / cvv) % gvv, cq = q % cvv;
          off = (16 * ib + 4 * gg + r) * COLS + 16 * jb + CPC * cq;
        }
      }
    });
    return off;
  }
};

template <int NX, int NU>
struct Tile16Layout {
  static constexpr int NS = NX + NU;
  static constexpr int RX = (NX + 15) / 16;       // row tiles of F~, V, G (rows = states)
  static constexpr int RS = (NS + 15) / 16;       // row tiles of Q~ (rows = states and controls)
  static constexpr int CA = (NS + 16) / 16;       // column tiles of Q~, F~, G (columns 0 .. ns, the affine one included)
  static constexpr int SU = 16 * CA;              // row stride of the control-row scratch (floats)
  using ImgC = PackedImage<NS, NS, RS, CA>;
  using ImgF = PackedImage<NX, NS, RX, CA>;
  // [C image | c (read up to row 16 RS) | F image | f (up to row 16 RX) | nu rows of [Qux | Quu | qu], later of K~ | Qxu]
  static constexpr int kC = 0, kc = kC + NS * NS, kF = kc + 16 * RS, kf = kF + NX * NS, kU = kf + 16 * RX,
                       kX = kU + NU * SU, kFloats = (kX + NX * NU + 3) / 4 * 4;
  static constexpr size_t lds_bytes() { return (size_t)4 * kFloats * sizeof(float); }   // four wavefronts per workgroup
};

#ifndef DMPC_T16_OCC
#define DMPC_T16_OCC 2    // wavefronts per SIMD the registers and the LDS image are sized for
#endif
#ifndef DMPC_T16_NT
#define DMPC_T16_NT 0     // bit 0: C_t by non-temporal LDS-DMA, bit 1: F_t too
#endif
#ifndef DMPC_T16_STAGGER_TICKS
#define DMPC_T16_STAGGER_TICKS 0   // 100 MHz ticks
#endif
#ifndef DMPC_T16_RING
#define DMPC_T16_RING (DMPC_T16_OCC >= 3 ? 3 : 4)   // steps the rollout requests its rows ahead (42 registers each)
#endif

template <int NX, int NU, bool ROLLOUT>
__global__ __launch_bounds__(256, DMPC_T16_OCC) void lqr_tile16_kernel(const LqrArgs a) {
  using Lay = Tile16Layout<NX, NU>;
  using ImgC = typename Lay::ImgC;
  using ImgF = typename Lay::ImgF;
  constexpr int NS = NX + NU, AFF = NS;
  static_assert(NX % 4 == 0 && NU % 4 == 0, "rows in whole lane groups of four");
  static_assert(NX % 16 + NU <= 16, "the control rows / columns lie inside one tile");
  static_assert(NS + 1 <= 64, "the gain solve holds a row of [Qux | Quu | qu] across one wavefront");
  constexpr int RX = Lay::RX, RS = Lay::RS, CA = Lay::CA, SU = Lay::SU;
  constexpr int TA = AFF / 16, JA = AFF % 16;                  // the affine column: tile column TA, lane column JA
  constexpr int TU = NX / 16, JU = NX % 16, GU = JU / 4, RU = NU / 4;   // controls: tile TU, lane groups GU.., lane columns JU..
  constexpr bool kRaggedX = NX % 16 != 0;                      // the last state tile also holds control rows / columns
  using G64 = Group<64>;

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  int b = blockIdx.x * 4 + wv;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  b = __builtin_amdgcn_readfirstlane(b);
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const bool col_aff = lane == AFF;
  const bool k_lane = lane < NX || col_aff;
  float *Ks = a.Ks != nullptr ? a.Ks : a.wsK;
  float *ks = a.Ks != nullptr ? a.ks : a.wsk;
  int info_bits = 0;

  extern __shared__ __attribute__((aligned(16))) float tile16_lds[];
  float *slot = tile16_lds + wv * Lay::kFloats;
  const unsigned slot_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot);
  // (the padding behind c / f and the scratch rows are read before anything writes them: finite values, never used)
  for (int i = lane; i < Lay::kFloats; i += 64) slot[i] = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- LDS-DMA lane patterns: a full tile's chunk L = 16 r + 4 g + cq is columns 4 cq.. of row 4 g + r; the chunks of the
  // partial tiles follow the image order (PackedImage::tail_src_offset), one register per instruction
  const unsigned vpat = (unsigned)(((4 * ((lane >> 2) & 3) + (lane >> 4)) * NS + 4 * (lane & 3)) * 4);
  const unsigned voff16 = lane * 16;
  unsigned tailC[ImgC::tail_instrs() > 0 ? ImgC::tail_instrs() : 1], tailF[ImgF::tail_instrs() > 0 ? ImgF::tail_instrs() : 1];
  static_for<0, ImgC::tail_instrs()>([&](auto k) { tailC[k.value] = 4u * (unsigned)ImgC::tail_src_offset(64 * (ImgC::n_full() + k.value) + lane); });
  static_for<0, ImgF::tail_instrs()>([&](auto k) { tailF[k.value] = 4u * (unsigned)ImgF::tail_src_offset(64 * (ImgF::n_full() + k.value) + lane); });
  auto dma_image = [&](auto img, auto nt, const float *src, unsigned dst, const unsigned *tail) {
    using Img = decltype(img);
    constexpr bool NT = decltype(nt)::value;
    const unsigned long long base = reinterpret_cast<unsigned long long>(src);
    static_for<0, Img::n_full()>([&](auto k) {
      constexpr int tl = Img::full_tile(k.value), ib = tl / CA, jb = tl % CA;
      set_m0(__builtin_amdgcn_readfirstlane(dst + k.value * 1024));   // (uniform by construction; said so for -O1 builds)
      tile16_dma_full<NT>(vpat, base + (16 * ib * NS + 16 * jb) * 4);
    });
    static_for<0, Img::tail_instrs()>([&](auto k) {
      constexpr int left = Img::total_chunks() - 64 * (Img::n_full() + k.value);
      set_m0(__builtin_amdgcn_readfirstlane(dst + (Img::n_full() + k.value) * 1024));
      if constexpr (left >= 64) tile16_dma_full<NT>(tail[k.value], base);
      else tile16_dma_masked<NT>(tail[k.value], base, (1ull << left) - 1);
    });
  };
  auto dma_issue = [&](int t) {
    const size_t tb = (size_t)t * B + b;
    dma_image(ImgC{}, std::integral_constant<bool, (DMPC_T16_NT & 1) != 0>{}, a.C + tb * NS * NS, slot_addr + Lay::kC * 4, tailC);
    wave_dma_region<NS * 4>(a.c + tb * NS, slot_addr + Lay::kc * 4, voff16);
    if (t < T - 1) {   // uniform; there is no F_{T-1}
      dma_image(ImgF{}, std::integral_constant<bool, (DMPC_T16_NT & 2) != 0>{}, a.F + tb * NX * NS, slot_addr + Lay::kF * 4, tailF);
      if (has_f) wave_dma_region<NX * 4>(a.f + tb * NX, slot_addr + Lay::kf * 4, voff16);
    }
  };
  // register r of tile (ib, jb) of an image at float offset `at`: lanes outside the tile's rows / columns read a duplicate
  // (finite, never used: rows and columns beyond the matrix only ever feed rows and columns beyond the matrix)
  auto tile_read = [&](auto img, auto ib, auto jb, int at, f4v &dst) {
    using Img = decltype(img);
    constexpr int gvv = Img::gv(ib.value), cvv = Img::cv(jb.value);
    if constexpr (gvv * cvv == 0) {
      dst = f4v{0.f, 0.f, 0.f, 0.f};
    } else {
      constexpr int base = 4 * Img::base_chunk(ib.value, jb.value), seg = Img::seg_floats(ib.value, jb.value);
      const int lp = (g < gvv ? g : gvv - 1) * 4 * cvv + (j < 4 * cvv ? j : 4 * cvv - 1);
      static_for<0, 4>([&](auto r) { dst[r.value] = slot[at + base + r.value * seg + lp]; });
    }
  };

  if constexpr (ROLLOUT && DMPC_T16_STAGGER_TICKS > 0) {
    // as in lqr_wave_mfma_backward: the wavefronts in an odd slot of their SIMD start late in the first round, so that the
    // rollout (memory) of one wavefront of a SIMD runs beside the sweep (arithmetic) of the other instead of beside its rollout
    const unsigned slot_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_ID bits 3:0 = wave slot in the SIMD
    constexpr unsigned kResident = DMPC_T16_OCC * 256;
    if (gridDim.x > kResident && blockIdx.x < kResident && (slot_id & 1u)) {
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)DMPC_T16_STAGGER_TICKS) __builtin_amdgcn_s_sleep(8);
    }
  }
  f4v V[RX][RX], Vaff[RX];   // [V | v]: V[rho][bb] = rows 16 rho.., columns 16 bb..; Vaff[rho] = v in lanes j == JA
#pragma unroll
  for (int r = 0; r < RX; ++r) {
#pragma unroll
    for (int c = 0; c < RX; ++c) V[r][c] = f4v{0.f, 0.f, 0.f, 0.f};
    Vaff[r] = f4v{0.f, 0.f, 0.f, 0.f};
  }

#ifdef DMPC_T16_TIMING   // scripts/microbench/tile16_phases.hip: s_memtime stamps, every wavefront adds its phases into a.x
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long treal0 = __builtin_amdgcn_s_memrealtime(), tcyc0 = tlast;
#define T16_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define T16_STAMP(i) do { } while (0)
#endif
  // DMPC_T16_PRIO: issue priority of the wavefront outside its two matrix-core blocks.  Two wavefronts share a SIMD; the one
  // streaming MFMAs back to back otherwise wins most issue slots and the other's gain solve and LDS traffic crawl.
#ifndef DMPC_T16_PRIO
#define DMPC_T16_PRIO 0
#endif
#define T16_PRIO(p) do { if (DMPC_T16_PRIO > 0) __builtin_amdgcn_s_setprio(p); } while (0)
  T16_PRIO(DMPC_T16_PRIO);
  dma_issue(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    // the slot has landed (requested a whole step ago); the only younger operations are the NU gain stores in between
    if (t < T - 1 && live) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T16_STAMP(0);
    // ---- slot -> registers
    f4v Q[RS][CA], Ft[RX][CA];
    static_for<0, RS>([&](auto ib) {
      static_for<0, CA>([&](auto jb) { tile_read(ImgC{}, ib, jb, Lay::kC, Q[ib.value][jb.value]); });
      const f4v cv = *reinterpret_cast<const f4v *>(slot + Lay::kc + 16 * ib.value + 4 * g);
      static_for<0, 4>([&](auto r) { Q[ib.value][TA][r.value] = j == JA ? cv[r.value] : Q[ib.value][TA][r.value]; });
    });
    if (t < T - 1) {
      static_for<0, RX>([&](auto ib) {
        static_for<0, CA>([&](auto jb) { tile_read(ImgF{}, ib, jb, Lay::kF, Ft[ib.value][jb.value]); });
        const f4v fv = *reinterpret_cast<const f4v *>(slot + Lay::kf + 16 * ib.value + 4 * g);
        static_for<0, 4>([&](auto r) { Ft[ib.value][TA][r.value] = j == JA ? fv[r.value] : Ft[ib.value][TA][r.value]; });
      });
    }
    if (t > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and has been read: it can take the next step's inputs
#ifndef DMPC_T16_KNOB_NODMA   // timing knob (wrong results): the sweep on whatever the slot holds
      dma_issue(t - 1);
#endif
    }
    T16_STAMP(1);
    T16_PRIO(0);
    if (t < T - 1) {
      // ---- G = V^T F~, v added to the affine column (the accumulator starts from it)
      f4v G[RX][CA];
      static_for<0, RX>([&](auto bb) {
        static_for<0, CA>([&](auto jb) { G[bb.value][jb.value] = jb.value == TA ? Vaff[bb.value] : f4v{0.f, 0.f, 0.f, 0.f}; });
      });
      static_for<0, RX>([&](auto rho) {
        static_for<0, 4>([&](auto r) {
          static_for<0, RX>([&](auto bb) {
            static_for<0, CA>([&](auto jb) {
              G[bb.value][jb.value] = mfma16(V[rho.value][bb.value][r.value], Ft[rho.value][jb.value][r.value], G[bb.value][jb.value]);
            });
          });
        });
      });
      T16_STAMP(2);
      // ---- Q~ += F~^T G
      static_for<0, RX>([&](auto rho) {
        static_for<0, 4>([&](auto r) {
          static_for<0, RS>([&](auto ib) {
            static_for<0, CA>([&](auto jb) {
              Q[ib.value][jb.value] = mfma16(Ft[rho.value][ib.value][r.value], G[rho.value][jb.value][r.value], Q[ib.value][jb.value]);
            });
          });
        });
      });
    }
    T16_PRIO(DMPC_T16_PRIO);
    T16_STAMP(3);
    // ---- the control rows [Qux | Quu | qu] -> LDS rows -> one register per row across the lanes
    if (g >= GU && g < GU + RU) {
      static_for<0, CA>([&](auto jb) {
        static_for<0, 4>([&](auto r) { slot[Lay::kU + (4 * (g - GU) + r.value) * SU + 16 * jb.value + j] = Q[TU][jb.value][r.value]; });
      });
    }
    // (lanes beyond the row's 16 CA columns read a duplicate: finite, never used)
    float Kr[NU];
    auto read_rows = [&] {
      lds_lanes_exchange();
#pragma unroll
      for (int m = 0; m < NU; ++m) Kr[m] = slot[Lay::kU + m * SU + (lane < 16 * CA ? lane : 0)];
    };
    read_rows();
    // ---- Qxu through LDS: written row-major [i][m], read in the compact contraction layout (register r2 of lane 16 g + i =
    // Qxu[i][4 r2 + g])
    f4v Xc[RX];   // only the first RU registers of each are used
    if (t > 0) {
      if (j >= JU && j < JU + NU) {
        static_for<0, RX>([&](auto ib) {
          static_for<0, 4>([&](auto r) {
            const int i = 16 * ib.value + 4 * g + r.value;
            if (!kRaggedX || i < NX) slot[Lay::kX + i * NU + (j - JU)] = Q[ib.value][TU][r.value];
          });
        });
      }
      lds_lanes_exchange();
      static_for<0, RX>([&](auto ib) {
        static_for<0, RU>([&](auto r2) {
          const int i = 16 * ib.value + j;
          Xc[ib.value][r2.value] = (!kRaggedX || i < NX) ? slot[Lay::kX + (kRaggedX && i >= NX ? 0 : i) * NU + 4 * r2.value + g] : 0.f;
        });
      });
    }
    T16_STAMP(4);
    // ---- gains (:112-120): Gauss-Jordan on the rows where they lie; the multiplier of row i at pivot k is lane nx+k of it.
    // Fast path without row interchanges and without branches: LAPACK interchanges at pivot k exactly when some |a_ik| (i > k)
    // exceeds |a_kk| - lane nx+k holds that whole column, one compare there per pivot collects the verdict, and the rare
    // trajectory that does need an interchange repeats the solve on the rows (still in LDS) in LAPACK's order.
    unsigned long long need_swap = 0;
    unsigned zero_pivot = 0;
#ifdef DMPC_T16_KNOB_NOGJ   // timing knob (wrong results): no gain solve
    static_for<0, 0>([&](auto kc) {
#else
    static_for<0, NU>([&](auto kc) {
#endif
      constexpr int kk = kc.value;
      if constexpr (kk + 1 < NU) {
        float d = 0.f;
#pragma unroll
        for (int i = kk + 1; i < NU; ++i) d = fmaxf(d, fabsf(Kr[i]));
        need_swap |= __builtin_amdgcn_ballot_w64(d > fabsf(Kr[kk])) & (1ull << (NX + kk));
      }
      const float p = G64::template bcast<NX + kk>(Kr[kk]);
      float l[NU];
#pragma unroll
      for (int i = 0; i < NU; ++i)
        if (i != kk) l[i] = G64::template bcast<NX + kk>(Kr[i]);
      zero_pivot |= (__builtin_bit_cast(unsigned, p) << 1) == 0u ? 1u : 0u;
      Kr[kk] *= fast_rcp(p);
#pragma unroll
      for (int i = 0; i < NU; ++i)
        if (i != kk) Kr[i] = fmaf(-l[i], Kr[kk], Kr[i]);
    });
    if (__builtin_expect(need_swap != 0, 0)) {   // uniform, rare
      read_rows();
      zero_pivot = 0;
      static_for<0, NU>([&](auto kc) {
        constexpr int kk = kc.value;
        float p = G64::template bcast<NX + kk>(Kr[kk]);
        float li[NU];
        float best = fabsf(p);
        int pr = kk;
#pragma unroll
        for (int i = kk + 1; i < NU; ++i) {
          li[i] = G64::template bcast<NX + kk>(Kr[i]);
          const bool gt = fabsf(li[i]) > best;   // first largest entry
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
        if (p == 0.f) zero_pivot = 1;
        const float rp = fast_rcp(p);
        Kr[kk] *= rp;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          if (i == kk) continue;
          const float l = i > kk ? li[i] : G64::template bcast<NX + kk>(Kr[i]);
          Kr[i] = fmaf(-l, Kr[kk], Kr[i]);
        }
      });
    }
    if (zero_pivot) info_bits |= 1;
    float Kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = -Kr[m];
    if (k_lane && live) {
      float *kp = col_aff ? ks + tb * NU : Ks + tb * NU * NX + lane;
      const int kstride = col_aff ? 1 : NX;
#pragma unroll
      for (int m = 0; m < NU; ++m) kp[m * kstride] = Kt[m];
    }
    T16_STAMP(5);
    if (t > 0) {
      // ---- K~ back through the same LDS rows (zero in the control columns), read in the compact contraction layout
      if (lane < 16 * CA) {
#pragma unroll
        for (int m = 0; m < NU; ++m) slot[Lay::kU + m * SU + lane] = k_lane ? Kt[m] : 0.f;
      }
      lds_lanes_exchange();
      f4v Kc[CA];
      static_for<0, CA>([&](auto jb) {
        static_for<0, RU>([&](auto r2) { Kc[jb.value][r2.value] = slot[Lay::kU + (4 * r2.value + g) * SU + 16 * jb.value + j]; });
      });
      // ---- value update (:151-152): V~ = Q~x. + Qxu K~ on the state tiles and the affine tile column
      static_for<0, RX>([&](auto ib) {
        static_for<0, CA>([&](auto jb) {
          if constexpr (jb.value < RX || jb.value == TA) {
            f4v acc = Q[ib.value][jb.value];
            static_for<0, RU>([&](auto r2) { acc = mfma16(Xc[ib.value][r2.value], Kc[jb.value][r2.value], acc); });
            if constexpr (jb.value == TA) {
              f4v va;
              static_for<0, 4>([&](auto r) {
                const bool in = j == JA && (!kRaggedX || 16 * ib.value + 4 * g + r.value < NX);
                va[r.value] = in ? acc[r.value] : 0.f;
              });
              Vaff[ib.value] = va;
            }
            if constexpr (jb.value < RX) {
              if constexpr (kRaggedX) {   // rows and columns of the controls leave the value function
                static_for<0, 4>([&](auto r) {
                  const bool in = 16 * ib.value + 4 * g + r.value < NX && 16 * jb.value + j < NX;
                  acc[r.value] = in ? acc[r.value] : 0.f;
                });
              }
              V[ib.value][jb.value] = acc;
            }
          }
        });
      });
    }
    T16_STAMP(6);
  }
#ifdef DMPC_T16_TIMING
  {
    const unsigned long long tcyc1 = __builtin_amdgcn_s_memtime(), treal1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long *out = reinterpret_cast<unsigned long long *>(a.x);
    if (lane == 0) {
      for (int i = 0; i < 7; ++i) atomicAdd(&out[i], tacc[i]);
      atomicAdd(&out[7], tcyc1 - tcyc0);
      atomicAdd(&out[8], treal1 - treal0);
      atomicAdd(&out[9], 1ull);
    }
  }
#endif
  if constexpr (ROLLOUT) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gain stores of this wavefront have reached L2
    wave_rollout<NX, NU, false, DMPC_T16_RING>(a, b, lane, live, Ks, ks, info_bits);
  }
  if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
