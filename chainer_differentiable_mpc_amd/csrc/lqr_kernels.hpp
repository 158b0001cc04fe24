// lqr_kernels.hpp - fused time-varying LQR solve for gfx950: backward Riccati sweep + forward
// rollout in one launch, one lane group per trajectory (see colwise.hpp for the layout).
//
// Follows lqr/lqr_recursion.py:69-209 (LqrRecursion) and, with MASKED, the clamped-control
// variant mpc/active_constrained_lqr.py:67-202 (LQR_active) of the reference.
//
// Per timestep t (descending) a group holds, one column per lane:
//     Q~ = [C_t | c_t] + F~^T (V~ F~)         F~ = [F_t | f_t],  V~ = [V | v]
//     K~ = -Quu^-1 [Qux | Quu | qu]           (in-register LU, every lane its own column)
//     V~ = Q~x. + Qxu K~ + K~^T (Q~u. + Quu K~)   (lqr_recursion.py:151-152, all terms kept)
// K~ goes to LDS (or to the Ks/ks arrays in HBM when T is too long for LDS); the forward
// sweep re-reads F_t one ROW per lane so that x_{t+1} = F_t [x;u] + f_t lands in lane i with
// nx + nu broadcast-FMAs and no reduction.
#pragma once
#include "colwise.hpp"
#include "dpp_blocks_gen.hpp"
#include "riccati_blocks.hpp"

namespace dmpc {

struct LqrArgs {
  int T, B;
  const float *C, *c, *F, *f, *x_init;
  const uint8_t *mask;  // [T,B,nu] or nullptr
  float *Ks, *ks;       // gains requested by the caller ([T,B,nu,nx], [T,B,nu]) or nullptr
  float *wsK, *wsk;     // caller workspace used for the gains when they do not fit in LDS (the tiled kernel's per-trajectory
                        // scratch lies behind them: LqrArgs::tiled_scratch)
  float *x, *u;
  int32_t *info;
  // training form of the solve (generated stream only): Quu_t [T,B,nu,nu] and Qxu_t [T,B,nx,nu] next to Ks / ks, for
  // DiffLqr.backward's second solve (dmpc_lqr_saved_solve); nullptr: not wanted
  float *Quu_out = nullptr, *Qxu_out = nullptr;
  // generated streams only: c given as two arrays, `c` = its state part [T,B,nx] and `c_u` its control part [T,B,nu]
  // (DiffLqr.backward's second solve takes [grad_x; grad_u] as they are); there x_init == nullptr with x != nullptr
  // means x_init = 0
  const float *c_u = nullptr;
  bool info_store = false;   // info[b] is written (0 included) instead of or-ed into: the caller need not clear it
  // the re-solve with saved gains (lqr_asm_kernel<..., AFFINE>): K_t [T,B,nu,nx], Quu_t [T,B,nu,nu], Qxu_t [T,B,nx,nu] of
  // an earlier solve of the same C, F; `c` is the new affine cost term, C is not read
  const float *Ks_in = nullptr, *Quu_in = nullptr, *Qxu_in = nullptr;
  // saving solve: [V_t | v_t] of every step, [T,B,nx,nx+1] (row i = V_t[i][:], v_t[i]) - the value function whose gradient the
  // co-states are: lambda_t = V_t x_t + v_t (what the one-pass gradient reads instead of C)
  float *Vv_out = nullptr;
  // the one-pass gradient (lqr_asm_kernel<..., AFFINE, ADJ>; DiffLqr.backward, differentiable_lqr.py:78-142): with the
  // saved gains above, Vv_in [T,B,nx,nx+1] and the forward solution tau_x [T,B,nx], tau_u [T,B,nu]; `c` / `c_u` are grad_x /
  // grad_u; outputs dC [T,B,ns,ns], dc [T,B,ns], dF [T-1,B,nx,ns], df [T-1,B,nx], dx0 [B,nx];
  // dC = w_a dtau (x) tau + w_b tau (x) dtau; df[t] = d_lambda[t + df_shift]
  const float *Vv_in = nullptr, *tau_x = nullptr, *tau_u = nullptr;
  float *dC = nullptr, *dc = nullptr, *dF = nullptr, *df = nullptr, *dx0 = nullptr;
  float w_a = 0.5f, w_b = 1.0f;
  int df_shift = 0;
  // container launches (lqr_kernel<..., PAD>): the problem's own dimensions, nx_log <= NX and nu_log <= NU of the kernel
  int nx_log = 0, nu_log = 0;
  // MPCstep.backward_rec on the wavefront-per-trajectory kernel (lqr_wave_mfma_backward<..., MPC>): k_t is a box QP
  // on (Quu, qu) with the bounds lower - u, upper - u (mpc/mpc_step.py:119-146); with mpc_states the re-centring
  // c_hat = C [x_t; u_t] + c happens inside the sweep (:305-317); mpc_n_qp_total [B] receives sum_t (1 + i_t)
  const float *mpc_controls = nullptr, *mpc_lower = nullptr, *mpc_upper = nullptr, *mpc_states = nullptr;
  int mpc_n_qp_iter = 0;
  int32_t *mpc_n_qp_total = nullptr;
  const int32_t *mpc_done = nullptr;   // lqr_wide_kernel<..., MPC>: device flag of the BoxDDP loop (non-zero -> no-op)
  float *tiled_scratch = nullptr;   // lqr_tiled_kernel (any nx, nu): [B][tiled_scratch_floats(nx, nu)] of the caller's workspace
};

enum LqrMode { kSolve = 0, kBackwardOnly = 1, kForwardOnly = 2 };

template <int N>
__device__ __forceinline__ void load_contig(const float *__restrict__ p, float (&dst)[N]) {
  // p is 4*N-byte aligned relative to a 16-byte aligned array base
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const float4 v = reinterpret_cast<const float4 *>(p)[i];
      dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
  } else if constexpr (N % 2 == 0) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float2 v = reinterpret_cast<const float2 *>(p)[i];
      dst[2 * i] = v.x; dst[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = p[i];
  }
}

// Register ring depths: how many timesteps of inputs are in flight per lane group.  With one
// wavefront per SIMD nothing else hides HBM/L2 latency, so the sweeps prefetch explicitly; the depth
// is bounded by a VGPR budget (the wave kernel for large states keeps a single slot).
constexpr int ring_depth(int regs_per_slot, int budget, int max_depth) {
  int d = budget / regs_per_slot;
  return d < 1 ? 1 : (d > max_depth ? max_depth : d);
}

// NX, NU: state / control dims.  L: lanes per trajectory (16 or 64).  MASKED: LQR_active.
// MODE: fused solve, gains only, or rollout only.  K_LDS: gains handed to the forward sweep
// through LDS (else through args.Ks/args.ks in HBM).
// PAD: the kernel is a CONTAINER for a smaller problem (a.nx_log <= NX states, a.nu_log <= NU controls - the shapes without a
// specialisation of their own): the loads place the problem's rows and columns at the container's positions (state i at i,
// control m at NX + m), everything else of [C|c] and [F|f] is 0 and the unused controls get a unit diagonal in Quu - their
// gain rows come out exactly 0, the pivot search never picks their rows for a real column (LAPACK's choice is unchanged),
// and every added term of a dot product is an exact 0.  Arrays in HBM keep the problem's own strides.
template <int NX, int NU, int L, bool MASKED, int MODE, bool K_LDS, bool PAD = false>
__global__ __launch_bounds__(256) void lqr_kernel(const LqrArgs a) {
  constexpr int NS = NX + NU;
  static_assert(NS + 1 <= L, "a trajectory's augmented columns must fit its lane group");
  constexpr int GPB = 256 / L;  // trajectories per 256-thread workgroup
  constexpr int KROW = NX + 1;  // [K_m | k_m]
  using G = Group<L>;
  using Blk = RiccatiBlocks<NX, NU, L>;

  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = blockIdx.x * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;  // keep the whole wave in lock step; only stores are suppressed
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  // the problem's dimensions and where container index i lies in its arrays (-1: padding)
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int i) -> int { return i < NX ? (i < nx ? i : -1) : (i - NX < nu ? nx + (i - NX) : -1); };
  const int lcol = lane < NS ? logical(lane) : -1;   // this lane's column of C / F
  const int lcol_c = lcol >= 0 ? lcol : 0;

  extern __shared__ float lds[];
  float *kl = lds + (size_t)grp * T * NU * KROW;  // this trajectory's gains [T][NU][KROW]

  const bool col_aff = lane == NS;               // lane owns the affine column
  const int lane_c = lane < NS ? lane : NS - 1;  // lanes past the matrix re-read its last column (never used)
  const bool k_lane = lane < NX || col_aff;      // lanes that hold a gain entry
  const int kidx = lane < NX ? lane : NX;        // position inside a [K_m | k_m] row (k at NX)
  int info_bits = 0;

  if constexpr (MODE != kForwardOnly) {
    // ------------------------------------------------------------ backward Riccati sweep
    // Two banks of G ring slots, ping-ponged: the loads of the NEXT G timesteps are issued before the
    // current G are computed.  (hipcc drains vmcnt to 0 at the loop header, so whatever is issued last
    // in the loop body gets no latency cover - issuing a whole bank first gives every load G steps.)
#ifndef DMPC_BWD_BUDGET
#define DMPC_BWD_BUDGET 72
#define DMPC_BWD_MAXG 2
#endif
    constexpr int G = ring_depth(2 * (NS + NX), DMPC_BWD_BUDGET, DMPC_BWD_MAXG);
    // The matrix columns are loaded by every lane with the same instructions.  The affine column is
    // loaded by every lane too (all read the same c_t / f_t words - no extra HBM traffic) into
    // registers of its own and merged into lane NS when the slot is consumed: merging at load time,
    // or loading under `if (lane == NS)`, puts a wait for the load right behind it.
    float Qr[2][G][NS], Fr[2][G][NX];  // [C_t], [F_t] columns
    float cr[2][G][NS], fr[2][G][NX];  // c_t, f_t
    // Branch-free on purpose: hipcc can only emit a counted `s_waitcnt vmcnt(N)` for the bank being
    // consumed if the number of loads issued behind it is the same on every path.  Out-of-range
    // timesteps are clamped (the duplicate loads are never consumed), F_{T-1} does not exist and is
    // replaced by F_{T-2}, and a missing f reads c instead (both discarded by the merge in step()).
    const float *Fsafe = T > 1 ? a.F : a.C;
    const float *fsafe = has_f ? a.f : a.c;
    auto issue_loads = [&](int t, float (&Qn)[NS], float (&Fn)[NX], float (&cn)[NS], float (&fn)[NX]) __attribute__((always_inline)) {
      t = t < 0 ? 0 : t;
      const size_t tb = (size_t)t * B + b;
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const size_t tbF = (size_t)tF * B + b;
      if constexpr (PAD) {   // clamped addresses, the same loads on every path as below; step() discards the padding
        const float *Cp = a.C + tb * ns * ns + lcol_c;
        const float *cp = a.c + tb * ns;
        static_for<0, NS>([&](auto i) {
          const int li = logical(i.value);   // uniform
          Qn[i.value] = Cp[(li >= 0 ? li : 0) * ns];
          cn[i.value] = cp[li >= 0 ? li : 0];
        });
        const float *Fp = Fsafe + tbF * nx * ns + lcol_c;
        const float *fp = fsafe + tbF * nx;
        static_for<0, NX>([&](auto k) {
          Fn[k.value] = Fp[(k.value < nx ? k.value : 0) * ns];
          fn[k.value] = fp[k.value < nx ? k.value : 0];
        });
        return;
      }
      const float *Cp = a.C + tb * NS * NS + lane_c;
#pragma unroll
      for (int i = 0; i < NS; ++i) Qn[i] = Cp[i * NS];
      load_contig<NS>(a.c + tb * NS, cn);
      const float *Fp = Fsafe + tbF * NX * NS + lane_c;
#pragma unroll
      for (int k = 0; k < NX; ++k) Fn[k] = Fp[k * NS];
      load_contig<NX>(fsafe + tbF * NX, fn);
    };

    float V[NX];  // [V | v] columns; lanes NX..NS-1 carry junk that is never broadcast
#pragma unroll
    for (int i = 0; i < NX; ++i) V[i] = 0.f;

    auto step = [&](int t, const float (&Qn)[NS], const float (&Fn)[NX], const float (&cn)[NS],
                    const float (&fn)[NX]) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      float Q[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) Q[i] = col_aff ? cn[i] : Qn[i];
      if constexpr (PAD) {   // rows / columns outside the problem: 0, and 1 on the diagonal of the unused controls
        const bool col_ok = col_aff || lcol >= 0;
        static_for<0, NS>([&](auto i) {
          const bool row = logical(i.value) >= 0;
          Q[i.value] = (row && col_ok) ? Q[i.value] : ((i.value >= NX && lane == i.value) ? 1.f : 0.f);
        });
      }
      if (t < T - 1) {
        float Fc[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) Fc[k] = col_aff ? (has_f ? fn[k] : 0.f) : Fn[k];
        if constexpr (PAD) {
          const bool col_ok = col_aff || lcol >= 0;
          static_for<0, NX>([&](auto k) { Fc[k.value] = (k.value < nx && col_ok) ? Fc[k.value] : 0.f; });
        }
        // W~ = V F~ (+ v in the affine column)     lqr_recursion.py:89,96: (F^T V) F, (F^T V) f + F^T v
        float W[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) W[i] = col_aff ? V[i] : 0.f;
        Blk::vf(W, V, Fc);
        Blk::ftw(Q, Fc, W);  // Q~ += F^T W~
      }
      float Kt[NU], R[NU];
      if constexpr (NU >= kRowGainsFromNu) {
        // three and more controls: Gauss-Jordan on the rows of [Qux | Quu | qu] where they lie (riccati_blocks.hpp)
        float Qu[NU];
        bool act[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          Qu[m] = Q[NX + m];
          act[m] = false;
        }
        if constexpr (MASKED) {
          static_for<0, NU>([&](auto m) { act[m.value] = (!PAD || m.value < nu) && a.mask[tb * nu + (m.value < nu ? m.value : 0)] != 0; });
        }
        if (gains_on_rows<NX, NU, L, MASKED>(Qu, act, lane, Kt, R, t > 0)) info_bits |= 1;
      } else {
        // every lane gets the full Quu                                lqr_recursion.py:102
        float Quu[NU][NU];
        static_for<0, NU>([&](auto l) {
#pragma unroll
          for (int m = 0; m < NU; ++m) Quu[m][l.value] = G::template bcast<NX + l.value>(Q[NX + m]);
        });
        // gains: K~ = -Quu^-1 * (own column of the u-rows)
#pragma unroll
        for (int m = 0; m < NU; ++m) Kt[m] = Q[NX + m];
        float A[NU][NU];
        if constexpr (MASKED) {
          // active_constrained_lqr.py:110-137: zero q_u / Qux rows of clamped controls, zero Quu
          // outside free x free, 1e-8 on the clamped diagonal.
          bool act[NU];
          static_for<0, NU>([&](auto m) { act[m.value] = (!PAD || m.value < nu) && a.mask[tb * nu + (m.value < nu ? m.value : 0)] != 0; });
#pragma unroll
          for (int m = 0; m < NU; ++m) {
            Kt[m] = act[m] ? 0.f : Kt[m];
#pragma unroll
            for (int l = 0; l < NU; ++l) {
              float v = (act[m] || act[l]) ? 0.f : Quu[m][l];
              if (m == l) v = act[m] ? (v + 1e-8f) : v;
              A[m][l] = v;
            }
          }
        } else {
#pragma unroll
          for (int m = 0; m < NU; ++m)
#pragma unroll
            for (int l = 0; l < NU; ++l) A[m][l] = Quu[m][l];
        }
        if constexpr (NU == 1) {
          Kt[0] = -((1.0f / A[0][0]) * Kt[0]);  // lqr_recursion.py:112-115, active_constrained_lqr.py:131-133
          if (A[0][0] == 0.f) info_bits |= 1;
        } else {
          int piv[NU];
          if (lu_factor_inplace<NU>(A, piv)) info_bits |= 1;  // reference: F.batch_inv (:116-120) / torch.lu
          lu_solve_inplace<NU>(A, piv, Kt);
#pragma unroll
          for (int m = 0; m < NU; ++m) Kt[m] = -Kt[m];
        }
        if (t > 0) {   // (Qu. + Quu K~) column, UNMASKED Quu
#pragma unroll
          for (int m = 0; m < NU; ++m) {
            R[m] = Q[NX + m];
#pragma unroll
            for (int l = 0; l < NU; ++l) R[m] = fmaf(Quu[m][l], Kt[l], R[m]);
          }
        }
      }
      // hand the gains to the forward sweep / the caller
      if (k_lane) {
        if constexpr (K_LDS) {
#pragma unroll
          for (int m = 0; m < NU; ++m) kl[(t * NU + m) * KROW + kidx] = Kt[m];
        }
        if (live && a.Ks != nullptr && (!PAD || col_aff || lane < nx)) {
#pragma unroll
          for (int m = 0; m < NU; ++m) {
            if (PAD && m >= nu) break;   // uniform
            if (col_aff) a.ks[tb * nu + m] = Kt[m];
            else a.Ks[(tb * nu + m) * nx + lane] = Kt[m];
          }
        }
      }
      if (t > 0) {
        // value update, UNMASKED blocks (lqr_recursion.py:151-152; active_constrained_lqr.py:143-145)
#pragma unroll
        for (int i = 0; i < NX; ++i) V[i] = Q[i];
        Blk::vupd(V, Q, Kt, R);
      }
    };

    // prologue: bank 0 <- the last G timesteps
    static_for<0, G>([&](auto j) {
      issue_loads(T - 1 - j.value, Qr[0][j.value], Fr[0][j.value], cr[0][j.value], fr[0][j.value]);
    });
    for (int t0 = T - 1; t0 >= 0; t0 -= 2 * G) {
      static_for<0, 2>([&](auto h) {  // half h computes bank h while bank 1-h is being filled
        constexpr int cur = h.value, nxt = 1 - h.value;
        const int tb0 = t0 - h.value * G;
        static_for<0, G>([&](auto j) {
          issue_loads(tb0 - G - j.value, Qr[nxt][j.value], Fr[nxt][j.value], cr[nxt][j.value], fr[nxt][j.value]);
        });
        static_for<0, G>([&](auto j) {
          const int t = tb0 - j.value;
          if (t >= 0) step(t, Qr[cur][j.value], Fr[cur][j.value], cr[cur][j.value], fr[cur][j.value]);
        });
      });
    }
  }

  if constexpr (MODE != kBackwardOnly) {
    // ------------------------------------------------------------ forward rollout
    // Lane i < NX owns ROW i of [F_t | f_t] (contiguous in HBM) and x[i]; the gains stay in the column
    // layout of the backward sweep (lane j holds K[.][j], lane NS holds k), so u_t[m] is a group sum and
    // no transposition, barrier or second address space is needed:
    //     u[m]  = sum_lanes Kcol[m] * xv            (xv: x[j] in lanes < NX, 1 in lane NS, else 0)
    //     x'[i] = f[i] + sum_j F[i][j] x[j] + sum_m F[i][NX+m] u[m]
    if constexpr (MODE == kSolve && !K_LDS) __threadfence_block();
#ifndef DMPC_FWD_BUDGET
#define DMPC_FWD_BUDGET 56
#define DMPC_FWD_MAXG 4
#endif
    constexpr int G = ring_depth(NS + 1 + NU, DMPC_FWD_BUDGET, DMPC_FWD_MAXG);  // ping-pong banks as in the backward sweep
    const bool row_x = lane < nx;
    const int lane_x = row_x ? lane : nx - 1;  // other lanes re-read the last row (never used)
    float Fr[2][G][NS], fr[2][G], Kr[2][G][NU];
    bool cl[2][G][NU];
    const float *Fsafe = T > 1 ? a.F : a.x_init;  // T == 1: nothing is read through it that is used
    const float *fsafe = has_f ? a.f : a.x_init;
    auto issue_row = [&](int t, float (&Fn)[NS], float &fn, float (&Kn)[NU], bool (&cn)[NU]) __attribute__((always_inline)) {
      t = t < T ? t : T - 1;  // branch-free, see the backward sweep
      const size_t tb = (size_t)t * B + b;
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const size_t tbF = (size_t)tF * B + b;
      if constexpr (PAD) {
        const float *Fp = Fsafe + (tbF * nx + lane_x) * ns;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const int lj = logical(j);   // uniform
          if (T > 1) Fn[j] = Fp[lj >= 0 ? lj : 0];
        }
      } else {
        if (T > 1) load_contig<NS>(Fsafe + (tbF * NX + lane_x) * NS, Fn);
      }
      fn = fsafe[has_f ? tbF * nx + lane_x : 0];
      if constexpr (MODE == kSolve && K_LDS) {
#pragma unroll
        for (int m = 0; m < NU; ++m) Kn[m] = kl[(t * NU + m) * KROW + kidx];
      } else {
        // gains from HBM: lanes < NX read Ks[t][b][m][lane], lane NS reads ks[t][b][m]
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          const int mc = (!PAD || m < nu) ? m : 0;
          const float *p = col_aff ? (a.ks + tb * nu + mc) : (a.Ks + (tb * nu + mc) * nx + (kidx < nx ? kidx : 0));
          Kn[m] = *p;   // (PAD: fstep() discards what lies outside the problem)
        }
      }
      if constexpr (MASKED) {
#pragma unroll
        for (int m = 0; m < NU; ++m) cn[m] = (!PAD || m < nu) && a.mask[tb * nu + (m < nu ? m : 0)] != 0;
      }
    };
    float xv = row_x ? a.x_init[(size_t)b * nx + lane] : (col_aff ? 1.f : 0.f);
    bool bad = false;
    auto fstep = [&](int t, const float (&Fn)[NS], const float fn, const float (&Kn)[NU], const bool (&cn)[NU]) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      float u[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        const bool kv = PAD ? (m < nu && (col_aff || lane < nx)) : k_lane;
        u[m] = group_sum<L>(kv ? Kn[m] * xv : 0.f);  // lqr_recursion.py:177
        if constexpr (MASKED) u[m] = cn[m] ? 0.f : u[m];   // :179-183
        bad = bad || !is_finite(u[m]);
      }
      bad = bad || !is_finite(xv);
      if (live) {
        if (row_x) a.x[tb * nx + lane] = xv;
        if (lane < nu) {
          float uo = u[0];
#pragma unroll
          for (int m = 1; m < NU; ++m) uo = (lane == m) ? u[m] : uo;
          a.u[tb * nu + lane] = uo;
        }
      }
      if (t < T - 1) {
        float acc = has_f ? fn : 0.f;
        {
          float M[NS + 1];
#pragma unroll
          for (int j = 0; j < NS; ++j) M[j] = (!PAD || logical(j) >= 0) ? Fn[j] : 0.f;
          M[NS] = 0.f;
          Blk::dot_x(acc, xv, M);  // :189, state part
        }
#pragma unroll
        for (int m = 0; m < NU; ++m) acc = fmaf((!PAD || m < nu) ? Fn[NX + m] : 0.f, u[m], acc);  // control part (u is in every lane)
        if (row_x) xv = acc;
      }
    };
    static_for<0, G>([&](auto j) {
      issue_row(j.value, Fr[0][j.value], fr[0][j.value], Kr[0][j.value], cl[0][j.value]);
    });
    for (int t0 = 0; t0 < T; t0 += 2 * G) {
      static_for<0, 2>([&](auto h) {
        constexpr int cur = h.value, nxt = 1 - h.value;
        const int tb0 = t0 + h.value * G;
        static_for<0, G>([&](auto j) {
          issue_row(tb0 + G + j.value, Fr[nxt][j.value], fr[nxt][j.value], Kr[nxt][j.value], cl[nxt][j.value]);
        });
        static_for<0, G>([&](auto j) {
          const int t = tb0 + j.value;
          if (t < T) fstep(t, Fr[cur][j.value], fr[cur][j.value], Kr[cur][j.value], cl[cur][j.value]);
        });
      });
    }
    if (bad) info_bits |= 2;
  }

  if (a.info != nullptr) {
    if (live && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

}  // namespace dmpc
