// lqr_tile16_f64.hpp - the fused solve of the large shapes in the reference's own precision (float64: lqr/differentiable_lqr.py:
// 169-172) on v_mfma_f64_16x16x4_f64: lqr_tile16_kernel (lqr_tile16.hpp) with doubles.  A 16 x 16 tile is four 64-bit registers
// per lane, the three products keep the X^T Y shape, the step's inputs arrive as a packed LDS image by gather LDS-DMA (a 16-byte
// chunk is TWO columns here), the gain solve runs on the rows brought
// through LDS into the column-per-lane layout with LAPACK's pivot order (getf2: true divisions), the rollout in the same launch.
// One wavefront per trajectory and per SIMD (a step's image is 23.6 KB, its registers twice the float32 kernel's).
//
// ONE difference in the layout: v_mfma_f64_16x16x4_f64 leaves element (4 r + g, j) - not (4 g + r, j) - in register r of lane
// 16 g + j.  The algebra does not care (register r of X's and of Y's tile still hold the same four rows, and four instructions
// still contract all sixteen), but everything that names a ROW does: the image (PackedImageRG: a partial tile keeps its first
// rows / 4 REGISTERS, all four lane groups), the affine column's reads, the control rows' and Qxu's way through LDS.
//
// Replaces lqr_f64_row_kernel<32, 8, 64> (a v_readlane pair per scalar of every product: 15 ms per 8,192-trajectory shard of
// BASELINE.json configs[4]) for the plain solve and for the gradient's second solve (c in two arrays, x_init = 0); the clamped
// solve (LQR_active) stays there.  Recursion: lqr/lqr_recursion.py:69-209; as in the float32 tile kernel the term K~^T (Q~u. + Quu K~) of :151-152 - the
// residual of the gain solve, ~1e-16 relative here - is left out.
#pragma once
#include "f64_row_kernels.hpp"
#include "lqr_tile16.hpp"

namespace dmpc {

typedef double d4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d4v mfma16_f64(double a, double b, d4v c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// The packed LDS image of a ROWS x COLS matrix for tiles whose register r of lane 16 g + j is element (4 r + g, j): a tile keeps
// rv = rows / 4 registers (all four lane groups each) and cv chunk columns; register r's segment is [g < 4][CPC cv columns],
// the segments of a tile and the tiles (row-major) follow one another.  Exactly ROWS * COLS elements.
template <int ROWS, int COLS, int RT, int CT, int CPC>
struct PackedImageRG {
  static_assert(ROWS % 4 == 0 && COLS % CPC == 0, "whole registers and 16-byte chunks");
  static constexpr int clampv(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
  static constexpr int rv(int ib) { return clampv((ROWS - 16 * ib) / 4, 4); }
  static constexpr int cv(int jb) { return clampv((COLS - 16 * jb) / CPC, 16 / CPC); }
  static constexpr int chunks(int ib, int jb) { return rv(ib) * 4 * cv(jb); }
  static constexpr int seg_elems(int ib, int jb) { return 4 * CPC * cv(jb); }
  static constexpr int base_chunk(int ib, int jb) {
    int n = 0;
    for (int i = 0; i < RT; ++i)
      for (int jj = 0; jj < CT; ++jj)
        if (i * CT + jj < ib * CT + jb) n += chunks(i, jj);
    return n;
  }
  static constexpr int total_chunks() { return ROWS * COLS / CPC; }
  static constexpr int total_instrs() { return (total_chunks() + 63) / 64; }
  static __device__ __forceinline__ int src_offset(int n) {   // element offset inside the matrix of image chunk n (0 past the end)
    int off = 0;
    static_for<0, RT * CT>([&](auto tl) {
      constexpr int ib = tl.value / CT, jb = tl.value % CT;
      if constexpr (chunks(ib, jb) > 0) {
        constexpr int base = base_chunk(ib, jb), cnt = chunks(ib, jb), cvv = cv(jb);
        if (n >= base && n < base + cnt) {
          const int q = n - base, r = q / (4 * cvv), gg = (q / cvv) % 4, cq = q % cvv;
          off = (16 * ib + 4 * r + gg) * COLS + 16 * jb + CPC * cq;
        }
      }
    });
    return off;
  }
};

template <int NX, int NU>
struct Tile16F64Layout {
  static constexpr int NS = NX + NU;
  static constexpr int RX = (NX + 15) / 16, RS = (NS + 15) / 16, CA = (NS + 16) / 16;
  static constexpr int SU = 16 * CA;
  using ImgC = PackedImageRG<NS, NS, RS, CA, 2>;
  using ImgF = PackedImageRG<NX, NS, RX, CA, 2>;
  // doubles: [C image | c (read up to row 16 RS) | F image | f (up to row 16 RX) | nu rows of [Qux | Quu | qu], later of K~ | Qxu]
  static constexpr int kC = 0, kc = kC + NS * NS, kF = kc + 16 * RS, kf = kF + NX * NS, kU = kf + 16 * RX, kX = kU + NU * SU,
                       kDoubles = (kX + NX * NU + 1) / 2 * 2;
  static constexpr size_t lds_bytes() { return (size_t)4 * kDoubles * sizeof(double); }   // four wavefronts per workgroup
};

template <int NX, int NU>
__global__ __launch_bounds__(256, 1) void lqr_tile16_f64_kernel(const F64RowSolve a) {
  using Lay = Tile16F64Layout<NX, NU>;
  using ImgC = typename Lay::ImgC;
  using ImgF = typename Lay::ImgF;
  constexpr int NS = NX + NU, AFF = NS;
  static_assert(NX % 4 == 0 && NU % 4 == 0 && NX % 16 + NU <= 16 && NS + 1 <= 64, "as lqr_tile16_kernel");
  static_assert(NX % 16 == 0, "state tiles free of control rows (the float32 kernel's masks for the other case are not carried over)");
  constexpr int RX = Lay::RX, RS = Lay::RS, CA = Lay::CA, SU = Lay::SU;
  constexpr int TA = AFF / 16, JA = AFF % 16, TU = NX / 16, JU = NX % 16, RU = NU / 4;
  using G64 = Group64<64>;

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
  int info_bits = 0;

  extern __shared__ __attribute__((aligned(16))) double tile16_f64_lds[];
  double *slot = tile16_f64_lds + wv * Lay::kDoubles;
  const unsigned slot_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)slot);
  for (int i = lane; i < Lay::kDoubles; i += 64) slot[i] = 0.0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- LDS-DMA: one lane-offset register per instruction of the image (64 chunks of 16 bytes = 128 doubles each)
  constexpr int NCI = ImgC::total_instrs(), NFI = ImgF::total_instrs();
  unsigned vC[NCI], vF[NFI];
  static_for<0, NCI>([&](auto k) { vC[k.value] = 8u * (unsigned)ImgC::src_offset(64 * k.value + lane); });
  static_for<0, NFI>([&](auto k) { vF[k.value] = 8u * (unsigned)ImgF::src_offset(64 * k.value + lane); });
  const unsigned voff16 = lane * 16;
  auto dma_image = [&](auto img, const double *src, unsigned dst, const unsigned *voff) {
    using Img = decltype(img);
    const unsigned long long base = reinterpret_cast<unsigned long long>(src);
    static_for<0, Img::total_instrs()>([&](auto k) {
      constexpr int left = Img::total_chunks() - 64 * k.value;
      set_m0(__builtin_amdgcn_readfirstlane(dst + k.value * 1024));
      if constexpr (left >= 64) tile16_dma_full<false>(voff[k.value], base);
      else tile16_dma_masked<false>(voff[k.value], base, (1ull << left) - 1);
    });
  };
  auto dma_issue = [&](int t) {
    const size_t tb = (size_t)t * B + b;
    dma_image(ImgC{}, a.C + tb * NS * NS, slot_addr + Lay::kC * 8, vC);
    if (a.c_u == nullptr) {   // uniform.  c as one array [T,B,ns], or (the gradient's second solve) as its state and control parts
      wave_dma_region<NS * 8>(a.c + tb * NS, slot_addr + Lay::kc * 8, voff16);
    } else {
      wave_dma_region<NX * 8>(a.c + tb * NX, slot_addr + Lay::kc * 8, voff16);
      wave_dma_region<NU * 8>(a.c_u + tb * NU, slot_addr + (Lay::kc + NX) * 8, voff16);
    }
    if (t < T - 1) {   // uniform; there is no F_{T-1}
      dma_image(ImgF{}, a.F + tb * NX * NS, slot_addr + Lay::kF * 8, vF);
      if (has_f) wave_dma_region<NX * 8>(a.f + tb * NX, slot_addr + Lay::kf * 8, voff16);
    }
  };
  auto tile_read = [&](auto img, auto ib, auto jb, int at, d4v &dst) {
    using Img = decltype(img);
    constexpr int rvv = Img::rv(ib.value), cvv = Img::cv(jb.value);
    dst = d4v{0.0, 0.0, 0.0, 0.0};      // registers beyond the matrix's rows: zero (rows that only ever feed rows beyond the problem)
    if constexpr (rvv * cvv > 0) {
      constexpr int base = 2 * Img::base_chunk(ib.value, jb.value), seg = Img::seg_elems(ib.value, jb.value);
      const int lp = g * 2 * cvv + (j < 2 * cvv ? j : 2 * cvv - 1);   // lanes beyond the tile's columns read a finite duplicate
      static_for<0, rvv>([&](auto r) { dst[r.value] = slot[at + base + r.value * seg + lp]; });
    }
  };

  d4v V[RX][RX], Vaff[RX];
#pragma unroll
  for (int r = 0; r < RX; ++r) {
#pragma unroll
    for (int c = 0; c < RX; ++c) V[r][c] = d4v{0.0, 0.0, 0.0, 0.0};
    Vaff[r] = d4v{0.0, 0.0, 0.0, 0.0};
  }

  dma_issue(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    const size_t tb = (size_t)t * B + b;
    if (t < T - 1 && live) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU) : "memory");   // (the NU gain stores in between are younger)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    d4v Q[RS][CA], Ft[RX][CA];
    static_for<0, RS>([&](auto ib) {
      static_for<0, CA>([&](auto jb) { tile_read(ImgC{}, ib, jb, Lay::kC, Q[ib.value][jb.value]); });
      static_for<0, 4>([&](auto r) {
        const double cv = slot[Lay::kc + 16 * ib.value + 4 * r.value + g];
        Q[ib.value][TA][r.value] = j == JA ? cv : Q[ib.value][TA][r.value];
      });
    });
    if (t < T - 1) {
      static_for<0, RX>([&](auto ib) {
        static_for<0, CA>([&](auto jb) { tile_read(ImgF{}, ib, jb, Lay::kF, Ft[ib.value][jb.value]); });
        static_for<0, 4>([&](auto r) {
          const double fv = slot[Lay::kf + 16 * ib.value + 4 * r.value + g];
          Ft[ib.value][TA][r.value] = j == JA ? fv : Ft[ib.value][TA][r.value];
        });
      });
    }
    if (t > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot has been read: it can take the next step's inputs
      dma_issue(t - 1);
    }
    if (t < T - 1) {
      d4v G[RX][CA];
      static_for<0, RX>([&](auto bb) {
        static_for<0, CA>([&](auto jb) { G[bb.value][jb.value] = jb.value == TA ? Vaff[bb.value] : d4v{0.0, 0.0, 0.0, 0.0}; });
      });
      static_for<0, RX>([&](auto rho) {
        static_for<0, 4>([&](auto r) {
          static_for<0, RX>([&](auto bb) {
            static_for<0, CA>([&](auto jb) {
              G[bb.value][jb.value] = mfma16_f64(V[rho.value][bb.value][r.value], Ft[rho.value][jb.value][r.value], G[bb.value][jb.value]);
            });
          });
        });
      });
      static_for<0, RX>([&](auto rho) {
        static_for<0, 4>([&](auto r) {
          static_for<0, RS>([&](auto ib) {
            static_for<0, CA>([&](auto jb) {
              Q[ib.value][jb.value] = mfma16_f64(Ft[rho.value][ib.value][r.value], G[rho.value][jb.value][r.value], Q[ib.value][jb.value]);
            });
          });
        });
      });
    }
    // ---- the control rows [Qux | Quu | qu] -> LDS rows -> one register per row across the lanes
    static_for<0, CA>([&](auto jb) {   // (nx % 16 == 0: the control rows are rows 0 .. nu - 1 of tile TU = its registers r < nu / 4)
      static_for<0, RU>([&](auto r) { slot[Lay::kU + (4 * r.value + g) * SU + 16 * jb.value + j] = Q[TU][jb.value][r.value]; });
    });
    lds_lanes_exchange();
    double Kr[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kr[m] = slot[Lay::kU + m * SU + (lane < 16 * CA ? lane : 0)];
    d4v Xc[RX];   // the first RU registers: register r2 of lane 16 g + i = Qxu[i][4 r2 + g]
    if (t > 0) {
      if (j >= JU && j < JU + NU) {
        static_for<0, RX>([&](auto ib) {
          static_for<0, 4>([&](auto r) { slot[Lay::kX + (16 * ib.value + 4 * r.value + g) * NU + (j - JU)] = Q[ib.value][TU][r.value]; });
        });
      }
      lds_lanes_exchange();
      static_for<0, RX>([&](auto ib) {
        static_for<0, RU>([&](auto r2) { Xc[ib.value][r2.value] = slot[Lay::kX + (16 * ib.value + j) * NU + 4 * r2.value + g]; });
      });
    }
    // ---- gains (:112-120): Gauss-Jordan on the rows, LAPACK's pivot order (first largest entry), true divisions
    static_for<0, NU>([&](auto kc) {
      constexpr int kk = kc.value;
      double p = G64::template bcast<NX + kk>(Kr[kk]);
      double li[NU];
      double mx = 0.0;
#pragma unroll
      for (int i = kk + 1; i < NU; ++i) {
        li[i] = G64::template bcast<NX + kk>(Kr[i]);
        mx = fmax(mx, fabs(li[i]));
      }
      if (__builtin_expect(mx > fabs(p), 0)) {   // uniform, rare
        double best = fabs(p);
        int pr = kk;
#pragma unroll
        for (int i = kk + 1; i < NU; ++i) {
          const bool gt = fabs(li[i]) > best;
          best = gt ? fabs(li[i]) : best;
          pr = gt ? i : pr;
        }
        pr = __builtin_amdgcn_readfirstlane(pr);
#pragma unroll
        for (int i = kk + 1; i < NU; ++i) {
          if (pr == i) {
            const double tmp = Kr[kk];
            Kr[kk] = Kr[i];
            Kr[i] = tmp;
            li[i] = p;
            p = G64::template bcast<NX + kk>(Kr[kk]);
          }
        }
      }
      if (p == 0.0) info_bits |= 1;
      Kr[kk] = Kr[kk] / p;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (i == kk) continue;
        const double l = i > kk ? li[i] : G64::template bcast<NX + kk>(Kr[i]);
        Kr[i] = fma(-l, Kr[kk], Kr[i]);
      }
    });
    double Kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) Kt[m] = -Kr[m];
    if (k_lane && live) {
      double *kp = col_aff ? a.ks + tb * NU : a.Ks + tb * NU * NX + lane;
      const int kstride = col_aff ? 1 : NX;
#pragma unroll
      for (int m = 0; m < NU; ++m) kp[m * kstride] = Kt[m];
    }
    if (t > 0) {
      if (lane < 16 * CA) {
#pragma unroll
        for (int m = 0; m < NU; ++m) slot[Lay::kU + m * SU + lane] = k_lane ? Kt[m] : 0.0;
      }
      lds_lanes_exchange();
      d4v Kc[CA];
      static_for<0, CA>([&](auto jb) {
        static_for<0, RU>([&](auto r2) { Kc[jb.value][r2.value] = slot[Lay::kU + (4 * r2.value + g) * SU + 16 * jb.value + j]; });
      });
      static_for<0, RX>([&](auto ib) {
        static_for<0, CA>([&](auto jb) {
          if constexpr (jb.value < RX || jb.value == TA) {
            d4v acc = Q[ib.value][jb.value];
            static_for<0, RU>([&](auto r2) { acc = mfma16_f64(Xc[ib.value][r2.value], Kc[jb.value][r2.value], acc); });
            if constexpr (jb.value == TA) {
              d4v va;
              static_for<0, 4>([&](auto r) { va[r.value] = j == JA ? acc[r.value] : 0.0; });
              Vaff[ib.value] = va;
            }
            if constexpr (jb.value < RX) V[ib.value][jb.value] = acc;
          }
        });
      });
    }
  }

  // ---- rollout (:160-200), row per lane: lane k < nx holds row k of [F_t | f_t], lane nx + m row m of [K_t | . | k_t]
  if (a.x != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gain stores of this wavefront have reached L2
    __threadfence_block();
    const bool f_lane = lane < NX, g_lane = lane >= NX && lane < NS;
    struct Row {
      double w[NS];
      double aff;
    };
    auto fetch = [&](int t, Row &r) {
      if (t >= T) return;   // uniform
      const size_t tb = (size_t)t * B + b;
#pragma unroll
      for (int q = 0; q < NS; ++q) r.w[q] = 0.0;
      r.aff = 0.0;
      if (f_lane) {
        if (t < T - 1) {
          const d2v *fp = reinterpret_cast<const d2v *>(a.F + (tb * NX + lane) * NS);
#pragma unroll
          for (int q = 0; q < NS / 2; ++q) {
            const d2v v = fp[q];
            r.w[2 * q] = v[0];
            r.w[2 * q + 1] = v[1];
          }
          if (has_f) r.aff = a.f[tb * NX + lane];
        }
      } else if (g_lane) {
        // the gains were written by other lanes of this wavefront moments ago: read past the L1 (never a stale line)
        const double *kp = a.Ks + (tb * NU + (lane - NX)) * NX;
#pragma unroll
        for (int q = 0; q < NX; ++q) r.w[q] = __builtin_nontemporal_load(kp + q);
        r.aff = __builtin_nontemporal_load(a.ks + tb * NU + (lane - NX));
      }
    };
    double xv = f_lane ? (a.x_init != nullptr ? a.x_init[(size_t)b * NX + lane] : 0.0) : 0.0;
    bool bad = false;
    auto fstep = [&](int t, const Row &r) {
      const size_t tb = (size_t)t * B + b;
      if (f_lane && live) a.x[tb * NX + lane] = xv;
      double xb[NX];
      static_for<0, NX>([&](auto i) { xb[i.value] = G64::template bcast<i.value>(xv); });
      double acc[4] = {r.aff, 0.0, 0.0, 0.0};
      static_for<0, NX>([&](auto i) { acc[i.value % 4] = fma(r.w[i.value], xb[i.value], acc[i.value % 4]); });
      double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      const double uo = s;
      if (g_lane && live) a.u[tb * NU + (lane - NX)] = uo;
      bad = bad || ((f_lane || g_lane) && !(fabs(uo) <= 1.7e308)) || (f_lane && !(fabs(xv) <= 1.7e308));
      static_for<0, NU>([&](auto m) { s = fma(r.w[NX + m.value], G64::template bcast<NX + m.value>(uo), s); });
      if (f_lane) xv = s;
    };
    Row ring[2];
    fetch(0, ring[0]);
    for (int t0 = 0; t0 < T; t0 += 2) {
      fetch(t0 + 1, ring[1]);
      fstep(t0, ring[0]);
      if (t0 + 1 < T) {
        fetch(t0 + 2, ring[0]);
        fstep(t0 + 1, ring[1]);
      }
    }
    if (bad) info_bits |= 2;
  }
  if (a.info != nullptr && live && info_bits != 0) atomicOr(&a.info[b], info_bits);
}

}  // namespace dmpc
