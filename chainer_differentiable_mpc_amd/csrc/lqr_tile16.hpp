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
// INPUTS.  The step's [C_t | c_t | F_t | f_t] come HBM -> LDS by per-lane gather LDS-DMA a step ahead, each instruction
// filling ONE tile in the order the reads want it ([r][g][j]: a 16-byte chunk = four consecutive columns of one row, 64 chunks
// = one tile), so a tile register is one conflict-free ds_read of 64 consecutive floats and the bytes fetched are exactly the
// arrays' own (chunks outside a matrix are masked off and stay at the zero the LDS area starts with).
//
// GAINS (lqr_recursion.py:112-120).  The control rows [Qux | Quu | qu] go through 1.7 KB of LDS into the column-per-lane
// layout (a row = one register across the lanes), where the Gauss-Jordan elimination of lqr_wave_mfma.hpp runs unchanged
// (LAPACK's pivot order); K~ returns through the same LDS rows in a COMPACT contraction layout (register r2 of lane 16 g + j
// = K~[4 r2 + g][j]), Qxu likewise (transposed through 1 KB), so Qxu K~ costs nu / 4 instructions per tile instead of four.
//
// As in lqr_wave_mfma.hpp the term K~^T (Q~u. + Quu K~) of lqr_recursion.py:151-152 multiplies the residual of the gain solve
// and is left out (measured against the float64 kernels on all 8,192 trajectories of a shard: tests/test_f64_gpu.py).
#pragma once
#include "lqr_wave_mfma.hpp"

namespace dmpc {

__device__ __forceinline__ f4v mfma16(float a, float b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// lanes of the tile-filling LDS-DMA instruction that lie inside a ROWS x COLS matrix: lane L = 16 r + 4 g + cq carries
// columns 4 cq .. 4 cq + 3 of row 4 g + r of tile (ib, jb)
constexpr unsigned long long tile16_dma_mask(int ib, int jb, int rows, int cols) {
  unsigned long long m = 0;
  for (int L = 0; L < 64; ++L) {
    const int r = L / 16, g = (L / 4) % 4, cq = L % 4;
    if (16 * ib + 4 * g + r < rows && 16 * jb + 4 * cq < cols) m |= 1ull << L;
  }
  return m;
}

__device__ __forceinline__ void tile16_dma_full(unsigned vpat, unsigned long long base) {
  asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(vpat), "s"(base) : "memory");
}
__device__ __forceinline__ void tile16_dma_masked(unsigned vpat, unsigned long long base, unsigned long long mask) {
  asm volatile("s_mov_b64 exec, %2\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(vpat), "s"(base), "s"(mask) : "memory");
}

// a ROWS x COLS row-major matrix (COLS % 4 == 0) at `src` (wave-uniform) -> RT x CT tiles of 1 KB at LDS byte address dst;
// vpat = this lane's byte offset inside a tile's source rows: ((4 g + r) * COLS + 4 cq) * 4
template <int ROWS, int COLS, int RT, int CT>
__device__ __forceinline__ void tile16_dma(const float *src, unsigned dst, unsigned vpat) {
  static_assert(COLS % 4 == 0, "16-byte chunks");
  static_for<0, RT>([&](auto ib) {
    static_for<0, CT>([&](auto jb) {
      constexpr unsigned long long mask = tile16_dma_mask(ib.value, jb.value, ROWS, COLS);
      if constexpr (mask != 0) {
        set_m0(dst + (ib.value * CT + jb.value) * 1024);
        const unsigned long long base = reinterpret_cast<unsigned long long>(src) + (16 * ib.value * COLS + 16 * jb.value) * 4;
        if constexpr (mask == ~0ull) tile16_dma_full(vpat, base);
        else tile16_dma_masked(vpat, base, mask);
      }
    });
  });
}

template <int NX, int NU>
struct Tile16Layout {
  static constexpr int NS = NX + NU;
  static constexpr int RX = (NX + 15) / 16;       // row tiles of F~, V, G (rows = states)
  static constexpr int RS = (NS + 15) / 16;       // row tiles of Q~ (rows = states and controls)
  static constexpr int CA = (NS + 16) / 16;       // column tiles of Q~, F~, G (columns 0 .. ns, the affine one included)
  static constexpr int SU = 16 * CA + 4;          // row stride of the control-row scratch (floats)
  static constexpr int XS = NU + 1;               // row stride of the Qxu scratch
  static constexpr int kC = 0, kF = kC + RS * CA * 256, kc = kF + RX * CA * 256, kf = kc + 16 * RS, kU = kf + 16 * RX,
                       kX = kU + NU * SU, kFloats = (kX + NX * XS + 3) / 4 * 4;
  static constexpr size_t lds_bytes() { return (size_t)4 * kFloats * sizeof(float); }   // four wavefronts per workgroup
};

template <int NX, int NU, bool ROLLOUT>
__global__ __launch_bounds__(256, 2) void lqr_tile16_kernel(const LqrArgs a) {
  using Lay = Tile16Layout<NX, NU>;
  constexpr int NS = NX + NU, AFF = NS;
  static_assert(NX % 4 == 0 && NU % 4 == 0, "rows in whole lane groups of four");
  static_assert(NX % 16 + NU <= 16, "the control rows / columns lie inside one tile");
  static_assert(NS + 1 <= 64, "the gain solve holds a row of [Qux | Quu | qu] across one wavefront");
  constexpr int RX = Lay::RX, RS = Lay::RS, CA = Lay::CA, SU = Lay::SU, XS = Lay::XS;
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
  // the area starts at zero: chunks outside the matrices are never written and must read as zero
  for (int i = lane; i < Lay::kFloats; i += 64) slot[i] = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  const unsigned vpat = (unsigned)(((4 * ((lane >> 2) & 3) + (lane >> 4)) * NS + 4 * (lane & 3)) * 4);
  const unsigned voff16 = lane * 16;
  auto dma_issue = [&](int t) {
    const size_t tb = (size_t)t * B + b;
    tile16_dma<NS, NS, RS, CA>(a.C + tb * NS * NS, slot_addr + Lay::kC * 4, vpat);
    wave_dma_region<NS * 4>(a.c + tb * NS, slot_addr + Lay::kc * 4, voff16);
    if (t < T - 1) {   // uniform; there is no F_{T-1}
      tile16_dma<NX, NS, RX, CA>(a.F + tb * NX * NS, slot_addr + Lay::kF * 4, vpat);
      if (has_f) wave_dma_region<NX * 4>(a.f + tb * NX, slot_addr + Lay::kf * 4, voff16);
    }
  };

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
      static_for<0, CA>([&](auto jb) {
        static_for<0, 4>([&](auto r) { Q[ib.value][jb.value][r.value] = slot[Lay::kC + (ib.value * CA + jb.value) * 256 + 64 * r.value + lane]; });
      });
      const f4v cv = *reinterpret_cast<const f4v *>(slot + Lay::kc + 16 * ib.value + 4 * g);
      static_for<0, 4>([&](auto r) { Q[ib.value][TA][r.value] = j == JA ? cv[r.value] : Q[ib.value][TA][r.value]; });
    });
    if (t < T - 1) {
      static_for<0, RX>([&](auto ib) {
        static_for<0, CA>([&](auto jb) {
          static_for<0, 4>([&](auto r) { Ft[ib.value][jb.value][r.value] = slot[Lay::kF + (ib.value * CA + jb.value) * 256 + 64 * r.value + lane]; });
        });
        const f4v fv = *reinterpret_cast<const f4v *>(slot + Lay::kf + 16 * ib.value + 4 * g);
        static_for<0, 4>([&](auto r) { Ft[ib.value][TA][r.value] = j == JA ? fv[r.value] : Ft[ib.value][TA][r.value]; });
      });
    }
    if (t > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and has been read: it can take the next step's inputs
      dma_issue(t - 1);
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
    if (t > 0 && j >= JU && j < JU + NU) {   // Qxu, row-major [i][m], for the value update
      static_for<0, RX>([&](auto ib) {
        static_for<0, 4>([&](auto r) {
          const int i = 16 * ib.value + 4 * g + r.value;
          if (!kRaggedX || i < NX) slot[Lay::kX + i * XS + (j - JU)] = Q[ib.value][TU][r.value];
        });
      });
    }
    float Kr[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kr[m] = slot[Lay::kU + m * SU + (lane < 16 * CA ? lane : 0)];
    if (lane >= 16 * CA) {
#pragma unroll
      for (int m = 0; m < NU; ++m) Kr[m] = 0.f;
    }
    T16_STAMP(4);
    // ---- gains (:112-120): Gauss-Jordan on the rows where they lie; the multiplier of row i at pivot k is lane nx+k of it
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
      const float rp = fast_rcp(p);
      Kr[kk] *= rp;
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
      f4v Kc[CA], Xc[RX];   // only the first RU registers of each are used
      static_for<0, CA>([&](auto jb) {
        static_for<0, RU>([&](auto r2) { Kc[jb.value][r2.value] = slot[Lay::kU + (4 * r2.value + g) * SU + 16 * jb.value + j]; });
      });
      static_for<0, RX>([&](auto ib) {
        static_for<0, RU>([&](auto r2) {
          const int i = 16 * ib.value + j;
          Xc[ib.value][r2.value] = (!kRaggedX || i < NX) ? slot[Lay::kX + (kRaggedX && i >= NX ? 0 : i) * XS + 4 * r2.value + g] : 0.f;
        });
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
    wave_rollout<NX, NU, false>(a, b, lane, live, Ks, ks, info_bits);
  }
  if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
