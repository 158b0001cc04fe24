// mpc_kernels.hpp - one box-constrained iLQR step (MPCstep of the reference), one lane group per
// trajectory, column-per-lane layout of colwise.hpp.
//
//   mpc_backward_rec_kernel   MPCstep.backward_rec, mpc/mpc_step.py:70-173: Riccati sweep whose
//                             feed-forward term k_t is a box QP (PNQP, warm-started from t+1) and whose
//                             gain rows of clamped controls are zero.
//   mpc_forward_rec_kernel    MPCstep.forward_rec, mpc/mpc_step.py:175-286: clamped closed-loop rollout
//                             under the TRUE linear dynamics with a per-trajectory backtracking line
//                             search on the TRUE quadratic cost.
//   taylor_c_kernel           the need_expand re-centring c_hat_t = C_t [x_t;u_t] + c_t, :305-317.
//   active_mask_kernel        active = |u-lo| <= 1e-8 | |u-hi| <= 1e-8 (:363-364) and -[dl_dx;dl_du].
#pragma once
#include <type_traits>
#include "colwise.hpp"
#include "dma_gather.hpp"
#include "dpp_blocks_gen.hpp"
#include "lqr_dma_kernel.hpp"
#include "lqr_kernels.hpp"
#include "pnqp_device.hpp"
#include "riccati_blocks.hpp"

namespace dmpc {

// "Sits on its bound": the reference tests |u - bound| <= 1e-8 in float64 (mpc_step.py:363-364), where a
// clamped control differs from its bound by at most an ulp (1e-16).  In float32 an ulp at |bound| ~ 1 is
// 6e-8, so the same test needs a tolerance of a few float32 ulps; controls that close are also snapped
// onto the bound by the rollout so that `u == bound` holds exactly for callers.
__device__ __forceinline__ float bound_tol(float bound) { return 1e-8f + 4.0f * 1.1920929e-07f * fmaxf(1.0f, fabsf(bound)); }

struct MpcBackArgs {
  int T, B;
  const float *C, *c, *F, *f;           // model (c is already re-centred when need_expand); f may be nullptr
  const float *controls, *lower, *upper;  // [T,B,nu]
  int n_qp_iter;                        // PNQP iteration cap (reference: 20, mpc_step.py:142)
  float *Ks, *ks;                       // [T,B,nu,nx], [T,B,nu]
  int32_t *n_qp_total;                  // [B]  sum_t (1 + i_t)   (mpc_step.py:145)
  int32_t *info;
  const int32_t *done;                  // device flag of the BoxDDP loop (box_ddp_kernels.hpp): non-zero -> no-op
  // batch-coupled PNQP termination (pnqp.py:139-144,172,187): zeroed decision slots, T * pnqp_sync_slots(n_qp_iter) of
  // them; nullptr = per-trajectory termination.  With slots the kernel must be launched cooperatively.
  unsigned *sync;
  // need_expand (mpc_step.py:305-317) inside the sweep: with `states` [T,B,nx] given, `c` is the ORIGINAL linear term
  // and the kernel forms c_hat_t = C_t [x_t; u_t] + c_t from the C rows it holds anyway (nullptr: c is used as given)
  const float *states;
  int info_store;                       // != 0: info[b] = this sweep's flags (plain store) instead of an atomic OR
  // container launches (PAD kernels, see lqr_kernel): the problem's own dimensions, nx_log <= NX and nu_log <= NU
  int nx_log = 0, nu_log = 0;
  // mpc_tiled_backward_kernel (any size: more than 8 controls, or more than 64 columns): [B][mpc_tiled_scratch_floats]
  float *tiled_scratch = nullptr;
};

// `block` = the workgroup's index among the 256-thread workgroups that share the batch (blockIdx.x for the kernel below)
// PAD: container for a smaller problem, as lqr_kernel<..., PAD> (lqr_kernels.hpp): rows and columns outside the problem are
// 0, the unused controls get a unit diagonal in Quu, qu = 0 and the box [-1, 1] - their QP solution is exactly 0 and never
// clamped, their gain rows are 0, and they add exact zeros to every sum the QP's tests are made of.
template <int NX, int NU, int L, bool PAD = false>
__device__ __forceinline__ void mpc_backward_rec_body(const MpcBackArgs &a, const int block, const unsigned n_blocks) {
  constexpr int NS = NX + NU;
  static_assert(NS + 1 <= L, "augmented columns must fit the lane group");
  constexpr int GPB = 256 / L;
  using G = Group<L>;
  using Blk = RiccatiBlocks<NX, NU, L>;

  if (a.done != nullptr && *a.done != 0) return;  // uniform: the iLQR loop has stopped
  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = block * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const bool col_aff = lane == NS;
  const int lane_c = lane < NS ? lane : NS - 1;
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int i) -> int { return i < NX ? (i < nx ? i : -1) : (i - NX < nu ? nx + (i - NX) : -1); };
  const int lcol = lane < NS ? logical(lane) : -1;
  const int lcol_c = lcol >= 0 ? lcol : 0;
  const bool col_ok = col_aff || lcol >= 0;

  float V[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) V[i] = 0.f;
  float kprev[NU];
#pragma unroll
  for (int m = 0; m < NU; ++m) kprev[m] = 0.f;
  int n_total = 0;
  int info_bits = 0;
  QpTermination term;
  term.slots = a.sync;
  term.n_blocks = n_blocks;
#ifdef DMPC_MPC_TIMING   // scripts/microbench/mpc_phases.hip: s_memtime stamps (100 MHz), workgroup 0 / thread 0 reports through a.info
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define DMPC_MSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define DMPC_MSTAMP(i) do { } while (0)
#endif

  // Inputs of one timestep, column-per-lane.  One wavefront per SIMD: nothing else hides HBM latency, so the loads
  // of step t-2 are issued before step t is computed (three banks rotated statically - hipcc drains vmcnt at a loop
  // header, so the prefetch sits in the same iteration as the compute it overlaps).
  struct Slot {
    float Q[NS], Fc[NX];      // [C_t | c_t] rows, [F_t | f_t] rows
    float uc[NU], lb[NU], ub[NU];
    float tau;                // lane j < ns: [x_t; u_t][j] (need_expand), else 0
  };
  const bool expand = a.states != nullptr;
  auto load = [&](int t, Slot &sl) __attribute__((always_inline)) {
    t = t < 0 ? 0 : t;  // prefetch past t = 0: step 0 again (never consumed)
    const size_t tb = (size_t)t * B + b;
    if constexpr (PAD) {   // clamped addresses, the same loads on every path; step() discards what lies outside the problem
      const char *qp = reinterpret_cast<const char *>(col_aff ? a.c + tb * ns : a.C + tb * ns * ns + lcol_c);
      const size_t qs = col_aff ? 4 : (size_t)ns * 4;
      static_for<0, NS>([&](auto i) {
        const int li = logical(i.value);
        sl.Q[i.value] = *reinterpret_cast<const float *>(qp + (li >= 0 ? li : 0) * qs);
      });
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const size_t tbF = (size_t)tF * B + b;
      const bool f_lane = col_aff && has_f;
      const char *fp = reinterpret_cast<const char *>(f_lane ? a.f + tbF * nx : (T > 1 ? a.F : a.C) + tbF * nx * ns + lcol_c);
      const size_t fs = f_lane ? 4 : (size_t)ns * 4;
      static_for<0, NX>([&](auto k) { sl.Fc[k.value] = *reinterpret_cast<const float *>(fp + (k.value < nx ? k.value : 0) * fs); });
      static_for<0, NU>([&](auto m) {
        const int mc = m.value < nu ? m.value : 0;
        sl.uc[m.value] = a.controls[tb * nu + mc];
        sl.lb[m.value] = a.lower[tb * nu + mc];
        sl.ub[m.value] = a.upper[tb * nu + mc];
      });
      sl.tau = 0.f;
      if (expand && lcol >= 0) sl.tau = lane < NX ? a.states[tb * nx + lane] : a.controls[tb * nu + (lane - NX)];
      return;
    }
    // per-lane base and stride: lane ns walks c / f, the others a column of C / F (one load per element, no branches)
    const char *qp = reinterpret_cast<const char *>(col_aff ? a.c + tb * NS : a.C + tb * NS * NS + lane_c);
    const size_t qs = col_aff ? 4 : NS * 4;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      sl.Q[i] = *reinterpret_cast<const float *>(qp);
      qp += qs;
    }
    const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);  // there is no F_{T-1}
    const size_t tbF = (size_t)tF * B + b;
    const bool f_lane = col_aff && has_f;
    const char *fp = reinterpret_cast<const char *>(f_lane ? a.f + tbF * NX : (T > 1 ? a.F : a.C) + tbF * NX * NS + lane_c);
    const size_t fs = f_lane ? 4 : NS * 4;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      sl.Fc[k] = *reinterpret_cast<const float *>(fp);
      fp += fs;
    }
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      sl.uc[m] = a.controls[tb * NU + m];
      sl.lb[m] = a.lower[tb * NU + m];
      sl.ub[m] = a.upper[tb * NU + m];
    }
    sl.tau = 0.f;
    if (expand && lane < NS) sl.tau = lane < NX ? a.states[tb * NX + lane] : a.controls[tb * NU + (lane - NX)];
  };

  auto step = [&](int t, const Slot &sl) __attribute__((always_inline)) {
    const size_t tb = (size_t)t * B + b;
    DMPC_MSTAMP(6);
    float Q[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) Q[i] = sl.Q[i];
    if constexpr (PAD) {
      static_for<0, NS>([&](auto i) {
        const bool row = logical(i.value) >= 0;
        Q[i.value] = (row && col_ok) ? Q[i.value] : ((i.value >= NX && lane == i.value) ? 1.f : 0.f);
      });
    }
#ifdef DMPC_MPC_TIMING
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMPC_MPC_TIMING_VMCNT) : "memory");
    DMPC_MSTAMP(0);
#endif
    if (expand) {   // c_hat = C tau + c: row sums over the matrix columns land in the affine column   :305-317
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const float s = group_sum<L>(lane < NS ? Q[i] * sl.tau : 0.f);
        Q[i] = col_aff ? Q[i] + s : Q[i];
      }
    }
    if (t < T - 1) {
      float Fc[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) Fc[k] = (col_aff && !has_f) ? 0.f : sl.Fc[k];
      if constexpr (PAD) static_for<0, NX>([&](auto k) { Fc[k.value] = (k.value < nx && col_ok) ? Fc[k.value] : 0.f; });
      float W[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) W[i] = col_aff ? V[i] : 0.f;
      Blk::vf(W, V, Fc);   // mpc_step.py:110,116
      Blk::ftw(Q, Fc, W);
    }
    DMPC_MSTAMP(1);
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
    if constexpr (PAD) static_for<0, NU>([&](auto m) {
      lo[m.value] = m.value < nu ? lo[m.value] : -1.f;
      hi[m.value] = m.value < nu ? hi[m.value] : 1.f;
    });
    // k_t: box QP, warm-started from the later timestep                        :141-146
    PnqpResult<NU> qp;
    float kt[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) kt[m] = kprev[m];
    DMPC_MSTAMP(2);
    pnqp_solve<NU>(Quu, qu, lo, hi, kt, /*warm=*/t != T - 1, a.n_qp_iter, qp, term);
    DMPC_MSTAMP(3);
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
    if (live) {
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        if (PAD && m >= nu) break;   // uniform
        if (col_aff) a.ks[tb * nu + m] = Kt[m];
        else if (lane < nx) a.Ks[(tb * nu + m) * nx + lane] = Kt[m];
      }
    }
    DMPC_MSTAMP(4);
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
    DMPC_MSTAMP(5);
  };

  Slot sa, sb, sc;
  load(T - 1, sa);
  load(T - 2, sb);
  for (int t = T - 1; t >= 0; t -= 3) {
    load(t - 2, sc);
    step(t, sa);
    if (t - 1 >= 0) {
      load(t - 3, sa);
      step(t - 1, sb);
    }
    if (t - 2 >= 0) {
      load(t - 4, sb);
      step(t - 2, sc);
    }
  }
  if (live && lane == 0) {
    a.n_qp_total[b] = n_total;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
#ifdef DMPC_MPC_TIMING
  if (block == 0 && threadIdx.x == 0) {
    unsigned long long *out = reinterpret_cast<unsigned long long *>(a.info);
    for (int i = 0; i < 8; ++i) out[i] = tacc[i];
  }
#endif
}

template <int NX, int NU, int L, bool PAD = false>
__global__ __launch_bounds__(256) void mpc_backward_rec_kernel(const MpcBackArgs a) {
  mpc_backward_rec_body<NX, NU, L, PAD>(a, blockIdx.x, gridDim.x);
}

// The pendulum of env_dx/pendulum.py:84-98 (simple model), state (cos th, sin th, dth), one torque.  One definition
// for the nominal rollout and for both line-search kernels: the search ends when the candidate collapses onto the
// nominal trajectory and its cost EQUALS the old one (mpc_step.py:196), which needs bit-equal dynamics.
struct PendulumModel {
  float kg, ku, dt, max_torque;  // 3g/(2l), 3/(m l^2)
  // Derivative of the torque clamp AT the limits +-max_torque: 1 (closed interval) or 0 (open).  Box-DDP's bounds equal
  // the torque limit, so saturated controls sit exactly there and the choice decides d x_{t+1} / d u_t for them.  The
  // reference differentiates F.clip with chainer.grad; Chainer's ClipGrad is taken to be inclusive (DESIGN.md section 4
  // states this as an assumption) - callers pass PendulumDx.clamp_grad_closed, the one place where it is set.
  bool clamp_closed;
  __device__ __forceinline__ float clamp_grad(float u) const {
    const float a = fabsf(u);
    return (clamp_closed ? a <= max_torque : a < max_torque) ? 1.f : 0.f;
  }
};
__device__ __forceinline__ PendulumModel pendulum_model(float g, float m, float l, float dt, float max_torque,
                                                        int clamp_closed) {
  return PendulumModel{3.f * g / (2.f * l), 3.f / (m * (l * l)), dt, max_torque, clamp_closed != 0};
}
// sin and cos of a small angle.  The rotation below takes them of the per-step increment wn * dt (|.| < pi/4 for any
// angular velocity below 15 rad/s at dt = 0.05): no argument reduction is needed there, and the library routine's
// reduction and its branches are a quarter of a rollout step's instructions.  Minimax kernels of the Cephes sinf / cosf
// on [-pi/4, pi/4] (error below one ulp of the result); anything larger goes to the library.
__device__ __forceinline__ void sincos_increment(float x, float &sn, float &cs) {
  const float z = x * x;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  sn = fmaf(x * z, ps, x);
  cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const bool big = !(fabsf(x) <= 0.785398163f);
  if (__builtin_amdgcn_ballot_w64(big) != 0) {   // wave-uniform branch (almost never taken); per-lane result either way
    float sl, cl;
    sincosf(x, &sl, &cl);
    sn = big ? sl : sn;
    cs = big ? cl : cs;
  }
}

__device__ __forceinline__ void pendulum_next(const PendulumModel &p, float c, float s, float w, float u, float &cn,
                                              float &sn, float &wn, float &nth) {
  const float uc = fminf(fmaxf(u, -p.max_torque), p.max_torque);
  wn = w + p.dt * (p.kg * s + p.ku * uc);
  // (cos, sin)(atan2(s, c) + delta) = R(delta) (c, s) / |(c, s)|: the reference's atan2 -> add -> cos/sin
  // (pendulum.py:88-98) is a rotation of the normalised state by delta = wn * dt; no atan2, and the angle whose
  // sine and cosine are taken is the small increment, not the absolute angle.  atan2(0, 0) = 0 there: unit vector (1, 0).
  const float r2 = fmaf(c, c, s * s);
  const float ri = r2 > 0.f ? rsqrtf(r2) : 0.f;
  const float cu = r2 > 0.f ? c * ri : 1.f, su = s * ri;
  float sd, cd;
  sincos_increment(wn * p.dt, sd, cd);
  cn = fmaf(cu, cd, -su * sd);
  sn = fmaf(su, cd, cu * sd);
  nth = 0.f;   // the absolute angle is not formed any more (callers use cn, sn)
}

// F = d pendulum_next / d [c, s, w, u] (3 x 4, row-major at Fp) and f = next - F [c, s, w, u] (at fq, may be null):
// what linearize_dynamics (mpc/approximate.py:77-119) obtains through chainer.grad.  One definition for the rollout
// kernel and for the line search's accepted pass (which hands the next iLQR iteration its model).
__device__ __forceinline__ void pendulum_jacobian_store(const PendulumModel &p, float c, float s, float w, float u, float cn,
                                                        float sn, float nw, float *Fp, float *fq) {
  const float inside = p.clamp_grad(u);
  const float r2 = c * c + s * s;
  const float dnw[4] = {0.f, p.dt * p.kg, 1.f, p.dt * p.ku * inside};
  const float dnth[4] = {-s / r2 + p.dt * dnw[0], c / r2 + p.dt * dnw[1], p.dt * dnw[2], p.dt * dnw[3]};
  const float xin[4] = {c, s, w, u};
  float f0 = cn, f1 = sn, f2 = nw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float r0 = -sn * dnth[j], r1 = cn * dnth[j], r2_ = dnw[j];
    Fp[j] = r0;
    Fp[4 + j] = r1;
    Fp[8 + j] = r2_;
    f0 = fmaf(-r0, xin[j], f0);
    f1 = fmaf(-r1, xin[j], f1);
    f2 = fmaf(-r2_, xin[j], f2);
  }
  if (fq != nullptr) {
    fq[0] = f0;
    fq[1] = f1;
    fq[2] = f2;
  }
}

struct MpcFwdArgs {
  int T, B;
  const float *Ks, *ks;                  // gains from backward_rec
  const float *controls, *states;        // current iterate  [T,B,nu], [T,B,nx]
  const float *lower, *upper;            // absolute control bounds [T,B,nu]
  const float *C, *c, *F, *f;            // TRUE QuadCost / LinDx (f may be nullptr)
  float ls_decay;
  int max_ls_iter;                       // only guards the first pass in the reference (:196), kept for the record
  int ls_cap;                            // safety cap on line-search passes (the reference loop is unbounded)
  float *x, *u;                          // new trajectory
  float *u_first;                        // controls of the first (alpha = 1) pass, or nullptr      (:260-263)
  float *costs, *old_costs, *alphas;     // [B]
  float *objs;                           // [T,B] per-step cost of the accepted pass, or nullptr
  int32_t *n_ls;                         // [B] passes run
  int32_t *info;
  // TRUE dynamics other than LinDx, evaluated inside the line search (mpc_step.py:237-240 calls a Python callable):
  //   0 LinDx (F, f above)   1 the pendulum of env_dx/pendulum.py:65-102, simple model (nx = 3, nu = 1)
  int dyn_kind;
  float pend_g, pend_m, pend_l, pend_dt, pend_max_torque;
  int pend_clamp_closed;                 // derivative of the torque clamp at its limits (PendulumModel::clamp_grad)
  const int32_t *done;                   // device flag of the BoxDDP loop: non-zero -> no-op
  // The accepted trajectory IS the next iLQR iteration's nominal one (BoxDDP re-rolls it with get_traj and linearises
  // it, mpc/box_ddp.py:123-136): the speculative pendulum search can write that model while it writes the
  // trajectory - F_next [T-1,B,3,4], f_next [T-1,B,3], c_next [T,B,4] = C tau + c (mpc_step.py:305-317); nullptr = no.
  float *F_next, *f_next, *c_next;
  // speculative search: every candidate keeps its trajectory in LDS (T * 4 floats per lane, T * 4 KB per workgroup)
  // and the accepted one is copied out instead of being rolled out a second time; 0 = no such buffer was given
  int traj_in_lds;
  const int32_t *info_in;                // [B] flags to merge into info (the backward sweep's, when it ran ahead of `done`)
  int nx_log = 0, nu_log = 0;            // container launches (PAD): the problem's own dimensions
  int info_store = 0;                    // != 0: info[b] = this search's flags (plain store; pendulum spec4 search only)
};

// chunks of one wave-step of the forward kernel's LDS-DMA ring (DMA variant):
// [C | c | F | f | Ks | ks | u | lower | upper | x], four trajectories each
template <int NX, int NU>
struct MpcFwdDmaLayout {
  static constexpr int NS = NX + NU;
  static constexpr int CH_C = 0, CH_c = CH_C + NS * NS, CH_F = CH_c + NS, CH_f = CH_F + NX * NS, CH_K = CH_f + NX;
  static constexpr int CH_k = CH_K + NU * NX, CH_u = CH_k + NU, CH_lo = CH_u + NU, CH_hi = CH_lo + NU, CH_x = CH_hi + NU;
  static constexpr int CH_END = CH_x + NX;
  static constexpr int kDma = (CH_END + 63) / 64;   // gather DMAs per step; padding lanes repeat chunk 0 of C
  static constexpr int SLOT = kDma * 256;           // floats per wave and timestep
  static constexpr int DB = kDma == 1 ? 8 : 4;      // ring depth
  static constexpr size_t lds_bytes() { return (size_t)4 * DB * SLOT * 4; }
};

// DMA (L == 16, LinDx, B % 4 == 0, 16-byte aligned arrays): the inputs of a timestep come through an LDS ring filled by
// per-lane gather LDS-DMA, DB - 1 steps ahead (the scheme of mpc_dma_kernels.hpp) instead of three compiler-managed
// register banks.  Both variants run the line search as a WAVE-UNIFORM loop: a pass is executed by every lane while
// any trajectory of the wavefront still searches, and the trajectories that are done neither store nor commit.
// PAD (register-bank variant, LinDx): container for a smaller problem - element i of tau lives in lane i (state) or NX + m
// (control), rows and columns outside the problem are 0, the unused controls stay 0 inside the box [-1, 1].
template <int NX, int NU, int L, bool DMA = false, bool PAD = false>
__global__ __launch_bounds__(256) void mpc_forward_rec_kernel(const MpcFwdArgs a) {
  constexpr int NS = NX + NU;
  static_assert(NS + 1 <= L, "augmented columns must fit the lane group");
  static_assert(!DMA || L == 16, "the ring is laid out for four trajectories per wavefront");
  static_assert(!(DMA && PAD), "containers take their inputs through the register banks");
  constexpr int GPB = 256 / L;
  using G = Group<L>;
  using Lay = MpcFwdDmaLayout<NX, NU>;

  if (a.done != nullptr && *a.done != 0) return;  // uniform: the iLQR loop has stopped
  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  int b = blockIdx.x * GPB + grp;
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int i) -> int { return i < NX ? (i < nx ? i : -1) : (i - NX < nu ? nx + (i - NX) : -1); };
  const int lrow = lane < NS ? logical(lane) : -1;   // this lane's element of tau / row of C
  const bool is_x = lane < nx;
  const bool is_tau = PAD ? lrow >= 0 : lane < NS;
  const bool is_u = lane >= NX && lane - NX < nu;
  const bool col_aff = lane == NS;
  const bool k_lane = is_x || col_aff;
  const int lane_x = is_x ? lane : nx - 1;
  const int lane_t = PAD ? (lrow >= 0 ? lrow : ns - 1) : (is_tau ? lane : NS - 1);

  // OLD_COST of (states, controls) (mpc_step.py:191) is accumulated during the first pass from the same rows of C.
  // The test `current_cost > OLD_COST` (:196,266) is NOT taken on the two rounded totals: near a fixed point their
  // difference is far below float32's resolution of the totals (the reference decides it in float64).  Per timestep
  //     obj(tau') - obj(tau) = 1/2 d'(C tau') + 1/2 tau'(C d) + c'd ,   d = tau' - tau
  // is exact algebra for any (also non-symmetric) C and has no cancellation: its rounding error scales with |d|, not
  // with the cost.  A candidate that has collapsed onto the nominal trajectory gives d = 0, hence exactly 0.
  float old_cost = 0.f;

  // Inputs of one timestep of a pass.  Register-bank variant: the loads of step t+2 are issued before step t is
  // computed (see the backward kernel above).
  struct Slot {
    float xt, kv[NU], uc[NU], lb[NU], ub[NU];
    float Crow[NS + 1], ci, Frow[NS + 1], fi;  // (+1: the fused DPP blocks take rows in the [row | affine] shape)
  };
  using Blk = RiccatiBlocks<NX, NU, L>;
  const bool lin = a.dyn_kind == 0;
  auto load = [&](int t, Slot &sl) __attribute__((always_inline)) {
    t = t < T ? t : T - 1;  // prefetch past the horizon: the last step again (never consumed)
    const size_t tb = (size_t)t * B + b;
    if constexpr (PAD) {   // clamped addresses; step() discards what lies outside the problem
      sl.xt = a.states[tb * nx + lane_x];
      static_for<0, NU>([&](auto m) {
        const int mc = m.value < nu ? m.value : 0;
        const float *kp = col_aff ? (a.ks + tb * nu + mc) : (a.Ks + (tb * nu + mc) * nx + lane_x);
        sl.kv[m.value] = *kp;
        sl.uc[m.value] = a.controls[tb * nu + mc];
        sl.lb[m.value] = a.lower[tb * nu + mc];
        sl.ub[m.value] = a.upper[tb * nu + mc];
      });
      const float *Cp = a.C + (tb * ns + lane_t) * ns;
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);
      const size_t tbF = (size_t)tF * B + b;
      const float *Fp = (T > 1 ? a.F : a.C) + (tbF * nx + lane_x) * ns;
      static_for<0, NS>([&](auto j) {
        const int lj = logical(j.value);
        sl.Crow[j.value] = Cp[lj >= 0 ? lj : 0];
        sl.Frow[j.value] = Fp[lj >= 0 ? lj : 0];
      });
      sl.Crow[NS] = 0.f;
      sl.Frow[NS] = 0.f;
      sl.ci = a.c[tb * ns + lane_t];
      sl.fi = has_f ? a.f[tbF * nx + lane_x] : 0.f;
      return;
    }
    sl.xt = a.states[tb * NX + lane_x];
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      const float *kp = col_aff ? (a.ks + tb * NU + m) : (a.Ks + (tb * NU + m) * NX + lane_x);
      sl.kv[m] = *kp;
      sl.uc[m] = a.controls[tb * NU + m];
      sl.lb[m] = a.lower[tb * NU + m];
      sl.ub[m] = a.upper[tb * NU + m];
    }
    load_contig<NS>(a.C + (tb * NS + lane_t) * NS, reinterpret_cast<float (&)[NS]>(sl.Crow));
    sl.Crow[NS] = 0.f;
    sl.Frow[NS] = 0.f;
    sl.ci = a.c[tb * NS + lane_t];
    if (lin) {
      const int tF = t < T - 1 ? t : (T > 1 ? T - 2 : 0);  // there is no F_{T-1}
      const size_t tbF = (size_t)tF * B + b;
      load_contig<NS>(a.F + (tbF * NX + lane_x) * NS, reinterpret_cast<float (&)[NS]>(sl.Frow));
      sl.fi = has_f ? a.f[tbF * NX + lane_x] : 0.f;
    }
  };
  // ---- the ring (DMA only)
  extern __shared__ float fwd_lds[];
  constexpr int DB = Lay::DB;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r4 = lane64 >> 4;
  float *ring = fwd_lds + wave * (DB * Lay::SLOT);
  unsigned ring_addr = 0;
  unsigned long long ptr0[Lay::kDma], ptr[Lay::kDma], str[Lay::kDma], strl[Lay::kDma];
  if constexpr (DMA) {
    const int b0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.x * 4 + wave) * 4);
    if (b0 >= a.B) return;   // whole wavefront (B % 4 == 0); no workgroup barrier below
    ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));
#pragma unroll
    for (int q = 0; q < Lay::kDma; ++q) {
      const int g = q * 64 + lane64;
      const char *base_p = (const char *)a.C;
      size_t per = (size_t)NS * NS * 4;
      int g0 = g;   // absent arrays and padding lanes: chunk 0 of C again
      bool isF = false;
      if (g < Lay::CH_c) { g0 = Lay::CH_C; }
      else if (g < Lay::CH_F) { base_p = (const char *)a.c; per = (size_t)NS * 4; g0 = Lay::CH_c; }
      else if (g < Lay::CH_f) { if (T > 1) { base_p = (const char *)a.F; per = (size_t)NX * NS * 4; g0 = Lay::CH_F; isF = true; } }
      else if (g < Lay::CH_K) { if (has_f && T > 1) { base_p = (const char *)a.f; per = (size_t)NX * 4; g0 = Lay::CH_f; isF = true; } }
      else if (g < Lay::CH_k) { base_p = (const char *)a.Ks; per = (size_t)NU * NX * 4; g0 = Lay::CH_K; }
      else if (g < Lay::CH_u) { base_p = (const char *)a.ks; per = (size_t)NU * 4; g0 = Lay::CH_k; }
      else if (g < Lay::CH_lo) { base_p = (const char *)a.controls; per = (size_t)NU * 4; g0 = Lay::CH_u; }
      else if (g < Lay::CH_hi) { base_p = (const char *)a.lower; per = (size_t)NU * 4; g0 = Lay::CH_lo; }
      else if (g < Lay::CH_x) { base_p = (const char *)a.upper; per = (size_t)NU * 4; g0 = Lay::CH_hi; }
      else if (g < Lay::CH_END) { base_p = (const char *)a.states; per = (size_t)NX * 4; g0 = Lay::CH_x; }
      ptr0[q] = (unsigned long long)base_p + (size_t)b0 * per + (size_t)(g - g0) * 16 - (unsigned long long)(q % 4) * 1024u;
      str[q] = (unsigned long long)(B * per);
      strl[q] = isF ? 0ull : str[q];   // there is no F_{T-1}: the step t = T-1 fetches F_{T-2} again (never consumed)
    }
  }
  int ti = 0;  // timesteps the pointers may still advance
  auto issue_next = [&](int slot) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + (unsigned)slot * (Lay::SLOT * 4));
    static_for<0, Lay::kDma>([&](auto q) {  // the instruction offset is 13 bits signed: M0 moves every 4 KB
      if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
      dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
    });
    if (ti > 0) {  // past the horizon the last blocks are fetched again (never consumed): the count per step stays exact
      const bool last = ti == 1;
#pragma unroll
      for (int q = 0; q < Lay::kDma; ++q) ptr[q] += last ? strl[q] : str[q];
      --ti;
    }
  };
  // per-lane LDS indices (floats, relative to a slot)
  const int i_x = Lay::CH_x * 4 + r4 * NX + lane_x;
  const int i_k = col_aff ? Lay::CH_k * 4 + r4 * NU : Lay::CH_K * 4 + r4 * NU * NX + lane_x;
  const int k_step = col_aff ? 1 : NX;
  const int i_C = Lay::CH_C * 4 + (r4 * NS + lane_t) * NS, i_c = Lay::CH_c * 4 + r4 * NS + lane_t;
  const int i_F = Lay::CH_F * 4 + (r4 * NX + lane_x) * NS, i_f = Lay::CH_f * 4 + r4 * NX + lane_x;
  auto read_slot = [&](const float *slot, Slot &sl) {
    sl.xt = slot[i_x];
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      sl.kv[m] = slot[i_k + m * k_step];
      sl.uc[m] = slot[Lay::CH_u * 4 + r4 * NU + m];
      sl.lb[m] = slot[Lay::CH_lo * 4 + r4 * NU + m];
      sl.ub[m] = slot[Lay::CH_hi * 4 + r4 * NU + m];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      sl.Crow[j] = slot[i_C + j];
      sl.Frow[j] = slot[i_F + j];
    }
    sl.Crow[NS] = 0.f;
    sl.Frow[NS] = 0.f;
    sl.ci = slot[i_c];
    sl.fi = has_f ? slot[i_f] : 0.f;
  };
  auto row_dot = [&](const Slot &sl, float v) {   // (C v)[lane]: broadcast-FMAs fused into one DPP instruction each
    float q = 0.f;
    Blk::dot_x(q, v, sl.Crow);
    Blk::dot_u(q, v, sl.Crow);
    return q;
  };

  float alpha = 1.0f;
  float cost = 0.f;
  int n_pass = 0;
  bool worse = true;
  bool searching = a.ls_cap > 0;   // this trajectory's search goes on                      :196
  for (int pass_idx = 0; __any(searching); ++pass_idx) {
    float xh = 0.f;                                            // new_x[0] = states[0]     :198
    if constexpr (!DMA) xh = is_x ? a.states[(size_t)b * nx + lane] : 0.f;   // (DMA: from the ring's first slot, below)
    float cost_p = 0.f, old_p = 0.f;
    float delta = 0.f;                  // current_cost - OLD_COST, summed per timestep
    auto step = [&](int t, const Slot &sl_in) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      Slot slp;
      if constexpr (PAD) {   // the slot with everything outside the problem replaced: zeros, and the box [-1, 1] around 0
        slp = sl_in;
        static_for<0, NU>([&](auto m) {
          const bool ok = m.value < nu;
          slp.kv[m.value] = ok ? slp.kv[m.value] : 0.f;
          slp.uc[m.value] = ok ? slp.uc[m.value] : 0.f;
          slp.lb[m.value] = ok ? slp.lb[m.value] : -1.f;
          slp.ub[m.value] = ok ? slp.ub[m.value] : 1.f;
        });
        static_for<0, NS>([&](auto j) {
          const bool ok = logical(j.value) >= 0;
          slp.Crow[j.value] = ok ? slp.Crow[j.value] : 0.f;
          slp.Frow[j.value] = ok ? slp.Frow[j.value] : 0.f;
        });
        slp.xt = is_x ? slp.xt : 0.f;
      }
      const Slot &sl = PAD ? slp : sl_in;
      const float xt = is_x ? sl.xt : 0.f;
      const float z = is_x ? (xh - xt) : (col_aff ? alpha : 0.f);  // [dx ; alpha] against [K_t | k_t]
      float un[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        float v = group_sum<L>(k_lane ? sl.kv[m] * z : 0.f) + sl.uc[m];               // :209-219
        const float lb = sl.lb[m], ub = sl.ub[m];
        v = fminf(fmaxf(v, lb), ub);                                                  // :221
        v = (v - lb <= bound_tol(lb)) ? lb : v;
        un[m] = (ub - v <= bound_tol(ub)) ? ub : v;
      }
      float tau = xh;  // [new_x_t ; new_u_t], element per lane
#pragma unroll
      for (int m = 0; m < NU; ++m) tau = (lane == NX + m) ? un[m] : tau;
      float tau0 = sl.xt;  // the iterate the step started from
#pragma unroll
      for (int m = 0; m < NU; ++m) tau0 = (lane == NX + m) ? sl.uc[m] : tau0;
      const float dt_ = is_tau ? tau - tau0 : 0.f;
      const float qi = row_dot(sl, tau), qd = row_dot(sl, dt_);
      // per-lane partial sums over the timesteps; the lanes of the group are added up ONCE after the pass (only the
      // optional per-step output `objs` needs the sum of a single step)                          :246-251, util.py:162-198
      const float obj_l = is_tau ? tau * fmaf(0.5f, qi, sl.ci) : 0.f;
      cost_p += obj_l;
      delta += is_tau ? fmaf(dt_, fmaf(0.5f, qi, sl.ci), 0.5f * tau0 * qd) : 0.f;
      if (pass_idx == 0)   // cost of the iterate, from C tau = C tau' - C d                                 :191
        old_p += is_tau ? tau0 * fmaf(0.5f, qi - qd, sl.ci) : 0.f;
      float obj = 0.f;
      if (a.objs != nullptr) obj = group_sum<L>(obj_l);
      if (live && searching) {  // outputs are overwritten by later passes; the last one is the accepted one
        if (is_x) a.x[tb * nx + lane] = xh;
        else if (is_u) a.u[tb * nu + (lane - NX)] = tau;
        if (a.objs != nullptr && lane == 0) a.objs[tb] = obj;
        if (a.u_first != nullptr && pass_idx == 0 && is_u) a.u_first[tb * nu + (lane - NX)] = tau;
      }
      if (t < T - 1 && !lin) {  // built-in pendulum (cos th, sin th, dth), torque -> next   pendulum.py:84-98
        if constexpr (NX == 3 && NU == 1) {
          const float cs = G::template bcast<0>(tau), sn = G::template bcast<1>(tau), dth = G::template bcast<2>(tau);
          const float uu = G::template bcast<3>(tau);
          const PendulumModel pm = pendulum_model(a.pend_g, a.pend_m, a.pend_l, a.pend_dt, a.pend_max_torque, a.pend_clamp_closed);
          float cn, sn2, wn, nth;
          pendulum_next(pm, cs, sn, dth, uu, cn, sn2, wn, nth);
          xh = lane == 0 ? cn : (lane == 1 ? sn2 : (lane == 2 ? wn : 0.f));
        }
      } else if (t < T - 1) {  // new_x_{t+1} = F_t [new_x;new_u] + f_t under the TRUE dynamics   :229-236
        float acc = sl.fi;
        Blk::dot_x(acc, tau, sl.Frow);
        Blk::dot_u(acc, tau, sl.Frow);
        xh = is_x ? acc : 0.f;
      }
    };
    if constexpr (DMA) {
      // software pipeline of mpc_backward_rec_dma_body, forward in time; the stores of a pass only make the counted
      // wait more conservative
      Slot sa, sb;
#pragma unroll
      for (int q = 0; q < Lay::kDma; ++q) ptr[q] = ptr0[q];
      ti = T - 1;
      static_for<0, DB>([&](auto j) { issue_next(j.value); });
      wait_vmcnt<(DB - 1) * Lay::kDma>();
      read_slot(ring, sa);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // these reads are in before the loop's first fetch refills their slot (round 4:
      // a cache-resident refill was seen to overtake them in lqr_wide_kernel - tiny problems, wrong rows at the first step)
      xh = is_x ? sa.xt : 0.f;
      for (int t0 = 0; t0 < T; t0 += DB) {
        static_for<0, DB>([&](auto j) {
          const int t = t0 + j.value;
          if (t < T) {
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
      wait_vmcnt<0>();   // the ring is refilled from t = 0 by the next pass
    } else {
      Slot sa, sb, sc;
      load(0, sa);
      load(1, sb);
      for (int t = 0; t < T; t += 3) {
        load(t + 2, sc);
        step(t, sa);
        if (t + 1 < T) {
          load(t + 3, sa);
          step(t + 1, sb);
        }
        if (t + 2 < T) {
          load(t + 4, sb);
          step(t + 2, sc);
        }
      }
    }
    cost_p = group_sum<L>(cost_p);
    delta = group_sum<L>(delta);
    if (pass_idx == 0) old_cost = group_sum<L>(old_p);
    if (searching) {
      cost = cost_p;
      ++n_pass;
      worse = delta > 0.f;                 // :266  current_cost > OLD_COST
      if (worse) alpha *= a.ls_decay;      // :268
      searching = worse && n_pass < a.ls_cap;
    }
  }
  int info_bits = 0;
  if (worse) {                           // cap hit: the reference would still be looping; :274
    alpha /= a.ls_decay;
    info_bits |= 8;
  }
  if (!is_finite(cost)) info_bits |= 2;
  if (live && lane == 0) {
    a.costs[b] = cost;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}


// forward_rec for the pendulum with a SPECULATIVE line search: the step sizes the search of mpc_step.py:196-268 can
// visit are known in advance (alpha_p = ls_decay^p), and on the swing-up problem it usually walks ten or more of
// them before the candidate collapses onto the nominal trajectory.  One lane per (trajectory, candidate): the 16
// lanes of a group roll 16 consecutive step sizes out at once, the first one that is not worse than the old cost is
// the one the sequential search stops at, and one more pass writes its trajectory.  Two passes instead of p* + 1.
// DMA: the inputs of a timestep (the same for the 16 candidates of a trajectory, and one contiguous run per array for the
// four trajectories of a wavefront) come through a ring of LDS slots filled by ONE per-lane gather LDS-DMA per step,
// kSpecDmaDepth - 1 steps ahead (the scheme of mpc_dma_kernels.hpp); needs B % 4 == 0 and 16-byte aligned arrays.  The
// ring sits behind the candidates' trajectories in dynamic LDS.
constexpr int kSpecDmaDepth = 4;
struct SpecDmaLayout {  // 16-byte chunks of one wave-step: [C | c | Ks | ks | u | lower | upper | x]
  static constexpr int CH_C = 0, CH_c = 16, CH_K = 20, CH_k = 23, CH_u = 24, CH_lo = 25, CH_hi = 26, CH_x = 27, CH_END = 30;
  static constexpr int SLOT = 256;  // floats per wave and timestep
  static constexpr size_t lds_bytes() { return (size_t)4 * kSpecDmaDepth * SLOT * 4; }
};

template <bool DMA>
__device__ __forceinline__ void mpc_forward_rec_pendulum_spec_body(const MpcFwdArgs &a, const int block) {
  constexpr int NX = 3, NU = 1, NS = 4, NC = 16;
  if (a.done != nullptr && *a.done != 0) return;
  const int k = threadIdx.x & (NC - 1);
  const int base = (threadIdx.x & 63) & ~(NC - 1);  // first lane of the group within the wavefront
  int b = block * (256 / NC) + (threadIdx.x / NC);
  const bool live = b < a.B;
  if (!live) b = a.B - 1;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const PendulumModel pm = pendulum_model(a.pend_g, a.pend_m, a.pend_l, a.pend_dt, a.pend_max_torque, a.pend_clamp_closed);
  extern __shared__ float traj[];                            // [T][4][256] when a.traj_in_lds
  const bool keep_traj = a.traj_in_lds != 0 && a.objs == nullptr;

  struct Slot {  // inputs of one timestep (the same for every candidate of the trajectory)
    float xt[NX], K[NX], kk, uc, lb, ub, C[NS][NS], cc[NS];
  };
  auto load = [&](int t, Slot &sl) {
    t = t < T ? t : T - 1;
    const size_t tb = (size_t)t * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      sl.xt[i] = a.states[tb * NX + i];
      sl.K[i] = a.Ks[tb * NX + i];
    }
    sl.kk = a.ks[tb];
    sl.uc = a.controls[tb];
    sl.lb = a.lower[tb];
    sl.ub = a.upper[tb];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      load_contig<NS>(a.C + (tb * NS + i) * NS, sl.C[i]);
      sl.cc[i] = a.c[tb * NS + i];
    }
  };
  // ---- DMA ring (DMA only)
  using Lay = SpecDmaLayout;
  constexpr int DB = kSpecDmaDepth;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r4 = lane64 >> 4;
  const int b0 = __builtin_amdgcn_readfirstlane((block * 4 + wave) * 4);
  float *ring = traj + (a.traj_in_lds != 0 ? (size_t)T * NS * 256 : 0) + wave * (DB * Lay::SLOT);
  unsigned ring_addr = 0;
  unsigned long long ptr0 = 0, ptr = 0, str = 0;
  if constexpr (DMA) {
    if (b0 >= a.B) return;   // whole wavefront (B % 4 == 0); no workgroup barrier below
    ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));
    const int g = lane64;
    const char *base_p = (const char *)a.C;
    size_t per = 64;
    int g0 = g;   // padding lanes: chunk 0 of C again
    if (g < Lay::CH_c) { g0 = Lay::CH_C; }
    else if (g < Lay::CH_K) { base_p = (const char *)a.c; per = 16; g0 = Lay::CH_c; }
    else if (g < Lay::CH_k) { base_p = (const char *)a.Ks; per = 12; g0 = Lay::CH_K; }
    else if (g < Lay::CH_u) { base_p = (const char *)a.ks; per = 4; g0 = Lay::CH_k; }
    else if (g < Lay::CH_lo) { base_p = (const char *)a.controls; per = 4; g0 = Lay::CH_u; }
    else if (g < Lay::CH_hi) { base_p = (const char *)a.lower; per = 4; g0 = Lay::CH_lo; }
    else if (g < Lay::CH_x) { base_p = (const char *)a.upper; per = 4; g0 = Lay::CH_hi; }
    else if (g < Lay::CH_END) { base_p = (const char *)a.states; per = 12; g0 = Lay::CH_x; }
    ptr0 = (unsigned long long)base_p + (size_t)b0 * per + (size_t)(g - g0) * 16;
    str = (unsigned long long)(B * per);
  }
  int ti = 0;  // timesteps the pointers may still advance
  auto issue_next = [&](int slot) {
    set_m0(__builtin_amdgcn_readfirstlane(ring_addr + (unsigned)slot * (Lay::SLOT * 4)));
    dma16_gather<0>(ptr);
    if (ti > 0) {  // past the horizon the last blocks are fetched again (never consumed): the count per step stays exact
      ptr += str;
      --ti;
    }
  };
  auto read_slot = [&](const float *slot, Slot &sl) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      sl.xt[i] = slot[Lay::CH_x * 4 + r4 * NX + i];
      sl.K[i] = slot[Lay::CH_K * 4 + r4 * NX + i];
    }
    sl.kk = slot[Lay::CH_k * 4 + r4];
    sl.uc = slot[Lay::CH_u * 4 + r4];
    sl.lb = slot[Lay::CH_lo * 4 + r4];
    sl.ub = slot[Lay::CH_hi * 4 + r4];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
      for (int j = 0; j < NS; ++j) sl.C[i][j] = slot[Lay::CH_C * 4 + (r4 * NS + i) * NS + j];
      sl.cc[i] = slot[Lay::CH_c * 4 + r4 * NS + i];
    }
  };
  auto quad = [&](const Slot &sl, const float (&tau)[NS]) {  // 1/2 tau'C tau + c'tau                util.py:162-198
    float obj = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      float qi = 0.f;
#pragma unroll
      for (int j = 0; j < NS; ++j) qi = fmaf(sl.C[i][j], tau[j], qi);
      obj = fmaf(tau[i], fmaf(0.5f, qi, sl.cc[i]), obj);
    }
    return obj;
  };
  // obj(tau) - obj(tau0) without cancellation (see mpc_forward_rec_kernel): 1/2 d'(C tau) + 1/2 tau0'(C d) + c'd
  auto quad_diff = [&](const Slot &sl, const float (&tau)[NS], const float (&tau0)[NS]) {
    float d[NS], acc = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) d[i] = tau[i] - tau0[i];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      float qi = 0.f, qd = 0.f;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        qi = fmaf(sl.C[i][j], tau[j], qi);
        qd = fmaf(sl.C[i][j], d[j], qd);
      }
      acc = fmaf(d[i], fmaf(0.5f, qi, sl.cc[i]), acc);
      acc = fmaf(0.5f * tau0[i], qd, acc);
    }
    return acc;
  };
  // one rollout with step size alpha; mode 0: cost only, 1: also the cost of the nominal trajectory and (candidate 0)
  // the controls of the alpha = 1 pass, 3: as 1 without the nominal cost, 2: write the trajectory.  `delta` = cost - OLD_COST summed per timestep.
  float delta = 0.f;
  auto pass = [&](auto mode_c, float alpha, float &cost, float &old_cost) {   // the mode is a compile-time constant:
    constexpr int mode = decltype(mode_c)::value;                              // no mode tests inside the T steps
    float xh[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xh[i] = a.states[(size_t)b * NX + i];                       // :198
    cost = 0.f;
    delta = 0.f;
    if (mode == 1 || mode == 3) old_cost = 0.f;
    auto step = [&](int t, const Slot &sl) {
      const size_t tb = (size_t)t * B + b;
      float v = alpha * sl.kk;
#pragma unroll
      for (int i = NX - 1; i >= 0; --i) v = fmaf(sl.K[i], xh[i] - sl.xt[i], v);
      v += sl.uc;                                                                            // :209-219
      v = fminf(fmaxf(v, sl.lb), sl.ub);                                                     // :221
      v = (v - sl.lb <= bound_tol(sl.lb)) ? sl.lb : v;
      v = (sl.ub - v <= bound_tol(sl.ub)) ? sl.ub : v;
      const float tau[NS] = {xh[0], xh[1], xh[2], v};
      const float obj = quad(sl, tau);
      cost += obj;
      const float tau0[NS] = {sl.xt[0], sl.xt[1], sl.xt[2], sl.uc};
      if (mode != 2) delta += quad_diff(sl, tau, tau0);
      if (mode != 2 && keep_traj) {
#pragma unroll
        for (int i = 0; i < NS; ++i) traj[(t * NS + i) * 256 + threadIdx.x] = tau[i];
      }
      if (mode == 1) old_cost += quad(sl, tau0);                                             // :191
      if (mode == 1 || mode == 3) {
        if (k == 0 && live && a.u_first != nullptr) a.u_first[tb] = v;                       // :260-263
      }
      if (mode == 2 && k == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) a.x[tb * NX + i] = xh[i];
        a.u[tb] = v;
        if (a.objs != nullptr) a.objs[tb] = obj;
        if (a.c_next != nullptr) {
#pragma unroll
          for (int i = 0; i < NS; ++i) {
            float acc = sl.cc[i];
#pragma unroll
            for (int j = 0; j < NS; ++j) acc = fmaf(sl.C[i][j], tau[j], acc);
            a.c_next[tb * NS + i] = acc;
          }
        }
      }
      if (t < T - 1) {
        float cn, sn, wn, nth;
        pendulum_next(pm, xh[0], xh[1], xh[2], v, cn, sn, wn, nth);
        if (mode == 2 && k == 0 && live && a.F_next != nullptr)
          pendulum_jacobian_store(pm, xh[0], xh[1], xh[2], v, cn, sn, wn, a.F_next + tb * 12,
                                  a.f_next != nullptr ? a.f_next + tb * 3 : nullptr);
        xh[0] = cn;
        xh[1] = sn;
        xh[2] = wn;
      }
    };
    Slot sa, sb;
    if constexpr (DMA) {
      // the software pipeline of mpc_backward_rec_dma_kernel, forward in time; the stores of a pass only make the
      // counted wait more conservative
      ptr = ptr0;
      ti = T - 1;
      static_for<0, DB>([&](auto j) { issue_next(j.value); });
      wait_vmcnt<DB - 1>();
      read_slot(ring, sa);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // these reads are in before the loop's first fetch refills their slot (round 4:
      // a cache-resident refill was seen to overtake them in lqr_wide_kernel - tiny problems, wrong rows at the first step)
      for (int t0 = 0; t0 < T; t0 += DB) {
        static_for<0, DB>([&](auto j) {
          const int t = t0 + j.value;
          if (t < T) {
            constexpr int nslot = (j.value + 1) % DB;
            issue_next(j.value);
            wait_vmcnt<DB - 1>();
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
      wait_vmcnt<0>();   // the ring is refilled from t = 0 by the next pass
    } else {
      load(0, sa);
      for (int t = 0; t < T; t += 2) {
        load(t + 1, sb);
        step(t, sa);
        if (t + 1 < T) {
          load(t + 2, sa);
          step(t + 1, sb);
        }
      }
    }
  };

  const int rounds = (a.ls_cap + NC - 1) / NC;
  float old_cost = 0.f, cost = 0.f, alpha = 1.f;
  float alpha_sel = 1.f, cost_sel = 0.f;
  int p_sel = -1;
  for (int r = 0; r < rounds; ++r) {
    const bool searching = p_sel < 0;
    if (!__any(searching)) break;
    const int p = r * NC + k;
    if (searching) {
      alpha = 1.f;
      for (int i = 0; i < p; ++i) alpha *= a.ls_decay;                                       // :268, p times
      float oc = 0.f;
      if (r == 0 && a.old_costs != nullptr) pass(std::integral_constant<int, 1>{}, alpha, cost, oc);
      else if (r == 0) pass(std::integral_constant<int, 3>{}, alpha, cost, oc);   // nobody asked for OLD_COST itself
      else pass(std::integral_constant<int, 0>{}, alpha, cost, oc);
      if (r == 0) old_cost = oc;
    }
    const bool accept = searching && p < a.ls_cap && !(delta > 0.f);                         // :266  cost > OLD_COST
    const unsigned mask = (unsigned)((__ballot(accept) >> base) & 0xFFFFull);
    if (searching && mask != 0u) {
      const int ks = __ffs(mask) - 1;
      p_sel = r * NC + ks;
      alpha_sel = __shfl(alpha, base + ks);
      cost_sel = __shfl(cost, base + ks);
      if (keep_traj) {
        // the accepted candidate's trajectory is in LDS (lane ks of this group wrote it during the pass above; a
        // wavefront executes in lock step, so its writes are complete): the 16 lanes of the group share the steps
        const int src = (int)(threadIdx.x & ~(NC - 1)) + ks;
        for (int t = k; t < T; t += NC) {
          const size_t tb = (size_t)t * B + b;
          float tau[NS];
#pragma unroll
          for (int i = 0; i < NS; ++i) tau[i] = traj[(t * NS + i) * 256 + src];
          if (live) {
#pragma unroll
            for (int i = 0; i < NX; ++i) a.x[tb * NX + i] = tau[i];
            a.u[tb] = tau[NX];
            if (a.c_next != nullptr) {
#pragma unroll
              for (int i = 0; i < NS; ++i) {
                float crow[NS];
                load_contig<NS>(a.C + (tb * NS + i) * NS, crow);
                float acc = a.c[tb * NS + i];
#pragma unroll
                for (int j = 0; j < NS; ++j) acc = fmaf(crow[j], tau[j], acc);
                a.c_next[tb * NS + i] = acc;
              }
            }
            if (t < T - 1 && a.F_next != nullptr) {
              const float cn = traj[((t + 1) * NS + 0) * 256 + src], sn = traj[((t + 1) * NS + 1) * 256 + src],
                          wn = traj[((t + 1) * NS + 2) * 256 + src];
              pendulum_jacobian_store(pm, tau[0], tau[1], tau[2], tau[3], cn, sn, wn, a.F_next + tb * 12,
                                      a.f_next != nullptr ? a.f_next + tb * 3 : nullptr);
            }
          }
        }
      }
    }
  }
  int info_bits = 0;
  int n_pass;
  if (p_sel < 0) {  // cap hit: the reference would still be looping; its last pass is the result        :274
    const int last = (a.ls_cap - 1) % NC;
    alpha_sel = __shfl(alpha, base + last);
    cost_sel = __shfl(cost, base + last);
    alpha_sel = (alpha_sel * a.ls_decay) / a.ls_decay;
    n_pass = a.ls_cap;
    info_bits |= 8;
    float dummy = 0.f, c2 = 0.f;
    float al = 1.f;
    for (int i = 0; i < a.ls_cap - 1; ++i) al *= a.ls_decay;
    pass(std::integral_constant<int, 2>{}, al, c2, dummy);
  } else {
    n_pass = p_sel + 1;
    if (!keep_traj) {   // no LDS copy of the candidates: roll the accepted one out again and write it
      float dummy = 0.f, c2 = 0.f;
      pass(std::integral_constant<int, 2>{}, alpha_sel, c2, dummy);
    }
  }
  if (!is_finite(cost_sel)) info_bits |= 2;
  if (live && k == 0) {
    if (a.info_in != nullptr) info_bits |= a.info_in[b];
    a.costs[b] = cost_sel;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha_sel;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

template <bool DMA>
__global__ __launch_bounds__(256) void mpc_forward_rec_pendulum_spec_kernel(const MpcFwdArgs a) {
  mpc_forward_rec_pendulum_spec_body<DMA>(a, blockIdx.x);
}

// The speculative search with a wavefront per trajectory: 16 candidates x 4 lanes.  The four lanes of a candidate split
// what has four rows - the rows of C in C tau and C d, the elements of tau kept in LDS - and repeat the rest (control,
// clamp, pendulum step): ~110 instructions per timestep instead of ~290 when one lane does it all, and with one
// wavefront per SIMD (or fewer) a sweep's time is its instruction count.  All T timesteps of the trajectory's inputs
// (30 floats each) are brought into LDS once by T dword gather DMAs - every round of 16 candidates and the copy-out of
// the accepted one read them from there - and the candidates' trajectories stay in LDS until one is accepted, which
// also serves the reference's "still looping at the cap" case.  Needs T <= kSpec4MaxT (LDS).
constexpr int kSpec4MaxT = 32;
struct Spec4Layout {  // floats of one timestep's inputs in LDS (one dword per DMA lane, 64 lanes)
  static constexpr int C = 0, c = 16, K = 20, k = 23, u = 24, lo = 25, hi = 26, x = 27, END = 30, SLOT = 64;
  static constexpr size_t lds_bytes(int T) { return (size_t)4 * 2 * T * SLOT * 4; }   // inputs + trajectories, 4 wavefronts
};

// one wavefront: the search of trajectory b, its inputs and candidates in slot `wave` of the workgroup's LDS
__device__ __forceinline__ void mpc_forward_rec_pendulum_spec4_wave(const MpcFwdArgs &a, const int b, const int wave) {
  constexpr int NX = 3, NS = 4, NC = 16;
  using Lay = Spec4Layout;
  if (a.done != nullptr && *a.done != 0) return;
  const int lane64 = threadIdx.x & 63;
  const int cand = lane64 >> 2, sub = lane64 & 3;
  if (b >= a.B) return;   // whole wavefront; no workgroup barrier below
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const PendulumModel pm = pendulum_model(a.pend_g, a.pend_m, a.pend_l, a.pend_dt, a.pend_max_torque, a.pend_clamp_closed);
  extern __shared__ float spec4_lds[];
  float *in_ = spec4_lds + (size_t)wave * 2 * T * Lay::SLOT, *traj = in_ + (size_t)T * Lay::SLOT;

  {  // every timestep's inputs -> LDS: lane g < 30 carries one float of [C | c | K | k | u | lower | upper | x]
    const int g = lane64;
    const float *base = a.C;
    size_t per = 16;
    int j = 0;   // lanes >= 30: C[0] again (never read)
    if (g < Lay::c) { j = g; }
    else if (g < Lay::K) { base = a.c; per = 4; j = g - Lay::c; }
    else if (g < Lay::k) { base = a.Ks; per = 3; j = g - Lay::K; }
    else if (g < Lay::u) { base = a.ks; per = 1; }
    else if (g < Lay::lo) { base = a.controls; per = 1; }
    else if (g < Lay::hi) { base = a.lower; per = 1; }
    else if (g < Lay::x) { base = a.upper; per = 1; }
    else if (g < Lay::END) { base = a.states; per = 3; j = g - Lay::x; }
    unsigned long long ptr = reinterpret_cast<unsigned long long>(base + (size_t)b * per + j);
    const unsigned long long str = (unsigned long long)(B * per * 4);
    const unsigned in_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(in_));
    for (int t = 0; t < T; ++t) {
      set_m0(in_addr + (unsigned)t * (Lay::SLOT * 4));
      asm volatile("global_load_lds_dword %0, off" ::"v"(ptr) : "memory");
      ptr += str;
    }
    wait_vmcnt<0>();
  }

  const unsigned m0 = sub == 0 ? ~0u : 0u, m1 = sub == 1 ? ~0u : 0u, m2 = sub == 2 ? ~0u : 0u, m3 = sub == 3 ? ~0u : 0u;
  auto pick4 = [&](float e0, float e1, float e2, float e3) {
    const unsigned r = (__builtin_bit_cast(unsigned, e0) & m0) | (__builtin_bit_cast(unsigned, e1) & m1) |
                       (__builtin_bit_cast(unsigned, e2) & m2) | (__builtin_bit_cast(unsigned, e3) & m3);
    return __builtin_bit_cast(float, r);
  };
  // one rollout of this lane's candidate with step size alpha; WITH_OLD: also the cost of the nominal trajectory.
  // Per-lane partial sums (this lane's row of C) - the four lanes of a candidate are added up after the pass.
  auto pass = [&](auto with_old_c, float alpha, float &cost, float &delta, float &old_cost) {
    constexpr bool WITH_OLD = decltype(with_old_c)::value;
    float xh[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xh[i] = in_[Lay::x + i];                                    // :198
    cost = 0.f;
    delta = 0.f;
    old_cost = 0.f;
    for (int t = 0; t < T; ++t) {
      const float *sl = in_ + t * Lay::SLOT;
      const float4 Cr = *reinterpret_cast<const float4 *>(sl + Lay::C + sub * 4);   // row `sub` of C_t
      const float cs = sl[Lay::c + sub];
      const float4 Kk = *reinterpret_cast<const float4 *>(sl + Lay::K);             // K_t (3), k_t
      const float4 ub4 = *reinterpret_cast<const float4 *>(sl + Lay::u);            // u_t, lower, upper, x_t[0]
      const float2 x12 = *reinterpret_cast<const float2 *>(sl + Lay::x + 1);
      const float uc = ub4.x, lb = ub4.y, ub = ub4.z;
      const float xt[NX] = {ub4.w, x12.x, x12.y};
      float v = alpha * Kk.w;
      v = fmaf(Kk.z, xh[2] - xt[2], v);
      v = fmaf(Kk.y, xh[1] - xt[1], v);
      v = fmaf(Kk.x, xh[0] - xt[0], v);
      v += uc;                                                                               // :209-219
      v = fminf(fmaxf(v, lb), ub);                                                           // :221
      v = (v - lb <= bound_tol(lb)) ? lb : v;
      v = (ub - v <= bound_tol(ub)) ? ub : v;
      const float tau[NS] = {xh[0], xh[1], xh[2], v};
      const float d[NS] = {xh[0] - xt[0], xh[1] - xt[1], xh[2] - xt[2], v - uc};
      float qi = 0.f, qd = 0.f;   // (C tau)[sub], (C d)[sub]
      qi = fmaf(Cr.x, tau[0], qi); qi = fmaf(Cr.y, tau[1], qi); qi = fmaf(Cr.z, tau[2], qi); qi = fmaf(Cr.w, tau[3], qi);
      qd = fmaf(Cr.x, d[0], qd); qd = fmaf(Cr.y, d[1], qd); qd = fmaf(Cr.z, d[2], qd); qd = fmaf(Cr.w, d[3], qd);
      // element `sub` of tau / tau0 by bit masks (hipcc turns a chain of selects on `sub` into exec-mask branches)
      const float tau_s = pick4(tau[0], tau[1], tau[2], tau[3]);
      const float tau0_s = pick4(xt[0], xt[1], xt[2], uc);
      const float d_s = tau_s - tau0_s;
      const float lin = fmaf(0.5f, qi, cs);
      cost = fmaf(tau_s, lin, cost);                                                         // util.py:162-198
      // obj(tau) - obj(tau0) without cancellation (see mpc_forward_rec_kernel): 1/2 d'(C tau) + 1/2 tau0'(C d) + c'd
      delta += fmaf(d_s, lin, 0.5f * tau0_s * qd);
      if constexpr (WITH_OLD) old_cost = fmaf(tau0_s, fmaf(0.5f, qi - qd, cs), old_cost);    // :191, C tau0 = C tau - C d
      traj[t * Lay::SLOT + lane64] = tau_s;
      if (t < T - 1) {
        float cn, sn, wn, nth;
        pendulum_next(pm, xh[0], xh[1], xh[2], v, cn, sn, wn, nth);
        xh[0] = cn;
        xh[1] = sn;
        xh[2] = wn;
      }
    }
    // the four lanes of the candidate: rows of C
    auto sum4 = [](float v_) {
      v_ += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v_), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
      v_ += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v_), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
      return v_;
    };
    cost = sum4(cost);
    delta = sum4(delta);
    if constexpr (WITH_OLD) old_cost = sum4(old_cost);
  };

  const int rounds = (a.ls_cap + NC - 1) / NC;
  float old_cost = 0.f, cost = 0.f, delta = 0.f, alpha = 1.f;
  float alpha_sel = 1.f, cost_sel = 0.f;
  int p_sel = -1, ks = 0;
  for (int r = 0; r < rounds && p_sel < 0; ++r) {   // wave-uniform: the wavefront has ONE trajectory
    const int p = r * NC + cand;
    alpha = 1.f;
    for (int i = 0; i < p; ++i) alpha *= a.ls_decay;                                         // :268, p times
    float oc = 0.f;
    if (r == 0 && a.old_costs != nullptr) {
      pass(std::true_type{}, alpha, cost, delta, oc);
      old_cost = oc;
    } else {
      pass(std::false_type{}, alpha, cost, delta, oc);
    }
    if (r == 0 && a.u_first != nullptr && cand == 0 && sub == 3) {                           // :260-263
      for (int t = 0; t < T; ++t) a.u_first[(size_t)t * B + b] = traj[t * Lay::SLOT + 3];
    }
    const bool accept = p < a.ls_cap && !(delta > 0.f);                                      // :266  cost > OLD_COST
    const unsigned long long mask = __ballot(accept);
    if (mask != 0ull) {
      const int first = __ffsll((long long)mask) - 1;   // first lane of the first accepted candidate
      ks = first >> 2;
      p_sel = r * NC + ks;
      alpha_sel = __shfl(alpha, first);
      cost_sel = __shfl(cost, first);
    }
  }
  int info_bits = 0;
  int n_pass;
  if (p_sel < 0) {  // cap hit: the reference would still be looping; its last pass is the result        :274
    ks = (a.ls_cap - 1) % NC;   // still in LDS: a candidate of the last round
    alpha_sel = __shfl(alpha, ks * 4);
    cost_sel = __shfl(cost, ks * 4);
    alpha_sel = (alpha_sel * a.ls_decay) / a.ls_decay;
    n_pass = a.ls_cap;
    info_bits |= 8;
  } else {
    n_pass = p_sel + 1;
  }
  // the result's trajectory (a wavefront executes in lock step: the pass's LDS writes are complete), with the NEXT
  // iteration's model when asked for; the lanes share the timesteps
  for (int t = lane64; t < T; t += 64) {
    const size_t tb = (size_t)t * B + b;
    const float4 tau = *reinterpret_cast<const float4 *>(traj + t * Lay::SLOT + ks * 4);
    a.x[tb * NX + 0] = tau.x;
    a.x[tb * NX + 1] = tau.y;
    a.x[tb * NX + 2] = tau.z;
    a.u[tb] = tau.w;
    if (a.c_next != nullptr || a.objs != nullptr) {
      const float *sl = in_ + t * Lay::SLOT;
      const float tv[NS] = {tau.x, tau.y, tau.z, tau.w};
      float cn4[NS], obj = 0.f;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const float4 Cr = *reinterpret_cast<const float4 *>(sl + Lay::C + i * 4);
        const float ci = sl[Lay::c + i];
        float q = 0.f;
        q = fmaf(Cr.x, tau.x, q); q = fmaf(Cr.y, tau.y, q); q = fmaf(Cr.z, tau.z, q); q = fmaf(Cr.w, tau.w, q);
        obj = fmaf(tv[i], fmaf(0.5f, q, ci), obj);                                           // util.py:162-198
        float acc = ci;                                                                      // mpc_step.py:305-317
        acc = fmaf(Cr.x, tau.x, acc); acc = fmaf(Cr.y, tau.y, acc); acc = fmaf(Cr.z, tau.z, acc); acc = fmaf(Cr.w, tau.w, acc);
        cn4[i] = acc;
      }
      if (a.objs != nullptr) a.objs[tb] = obj;
      if (a.c_next != nullptr) {
#pragma unroll
        for (int i = 0; i < NS; ++i) a.c_next[tb * NS + i] = cn4[i];
      }
    }
    if (t < T - 1 && a.F_next != nullptr) {
      const float4 nx4 = *reinterpret_cast<const float4 *>(traj + (t + 1) * Lay::SLOT + ks * 4);
      pendulum_jacobian_store(pm, tau.x, tau.y, tau.z, tau.w, nx4.x, nx4.y, nx4.z, a.F_next + tb * 12,
                              a.f_next != nullptr ? a.f_next + tb * 3 : nullptr);
    }
  }
  if (!is_finite(cost_sel)) info_bits |= 2;
  if (lane64 == 0) {
    if (a.info_in != nullptr) info_bits |= a.info_in[b];
    a.costs[b] = cost_sel;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha_sel;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr) {
      if (a.info_store) a.info[b] = info_bits;
      else if (info_bits != 0) atomicOr(&a.info[b], info_bits);
    }
  }
}

__global__ __launch_bounds__(256) void mpc_forward_rec_pendulum_spec4_kernel(const MpcFwdArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  mpc_forward_rec_pendulum_spec4_wave(a, __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave), wave);
}

// Pendulum rollout and analytic linearisation in one pass, one lane per trajectory: x_{t+1} = pendulum(x_t, u_t)
// (env_dx/pendulum.py:84-98, simple model), F_t = d x_{t+1} / d [x_t; u_t], f_t = x_{t+1} - F_t [x_t; u_t] - what
// BoxDDP obtains from get_traj (util.py:201-277) followed by linearize_dynamics (mpc/approximate.py:77-119, there
// through chainer.grad).  The derivative of the torque clamp at its limits is the caller's choice (PendulumModel::clamp_grad).
// The first launch of a dmpc_box_ddp chain clears the words the chain accumulates into (a memset or fill launch of its
// own costs 2-5 us of GPU time each for a few bytes): thread g of the grid zeroes word g of each range.
struct ChainClear {
  int32_t *p[3] = {nullptr, nullptr, nullptr};
  int n[3] = {0, 0, 0};
  __device__ __forceinline__ void run(int g) const {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (p[k] != nullptr && g < n[k]) p[k][g] = 0;
  }
};

struct PendulumArgs {
  int T, B;
  const float *x_init, *u;   // [B,3], [T,B,1]
  float g, m, l, dt, max_torque;
  int clamp_closed;          // derivative of the torque clamp at its limits (PendulumModel::clamp_grad)
  float *x, *F, *f;          // [T,B,3], [T-1,B,3,4] or nullptr, [T-1,B,3] or nullptr
  const int32_t *done;       // device flag of the BoxDDP loop: non-zero -> no-op
  // optional Taylor re-centring of a QuadCost at the rolled-out trajectory (what taylor_c_kernel computes):
  // c_back[t][b] = C[t][b] [x_t; u_t] + c[t][b]                                            mpc_step.py:305-317
  const float *C, *c;        // [T,B,4,4], [T,B,4]
  float *c_back;             // [T,B,4] or nullptr
  ChainClear clear;          // first launch of a box-DDP chain: words to zero (loop state, meeting words, flags)
};

__device__ __forceinline__ void pendulum_rollout_linearize_body(const PendulumArgs &a, const int b) {
  a.clear.run(b);
  if (b >= a.B) return;
  if (a.done != nullptr && *a.done != 0) return;
  const size_t B = (size_t)a.B;
  float c = a.x_init[b * 3 + 0], s = a.x_init[b * 3 + 1], w = a.x_init[b * 3 + 2];
  const PendulumModel pm = pendulum_model(a.g, a.m, a.l, a.dt, a.max_torque, a.clamp_closed);
  const bool taylor = a.c_back != nullptr;
  // inputs of step t + 1 are fetched while step t runs its atan2 / sin / cos (one lane per trajectory: nothing else
  // hides the latency)
  float u_nx = a.u[b], C_nx[4][4], c_nx[4];
  if (taylor) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      load_contig<4>(a.C + ((size_t)b * 4 + i) * 4, C_nx[i]);
      c_nx[i] = a.c[(size_t)b * 4 + i];
    }
  }
  for (int t = 0; t < a.T; ++t) {
    const size_t tb = (size_t)t * B + b;
    const float ur = u_nx;
    float Cc[4][4], cc[4];
    if (taylor) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Cc[i][j] = C_nx[i][j];
        cc[i] = c_nx[i];
      }
    }
    if (t + 1 < a.T) {
      u_nx = a.u[tb + B];
      if (taylor) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          load_contig<4>(a.C + ((tb + B) * 4 + i) * 4, C_nx[i]);
          c_nx[i] = a.c[(tb + B) * 4 + i];
        }
      }
    }
    a.x[tb * 3 + 0] = c;
    a.x[tb * 3 + 1] = s;
    a.x[tb * 3 + 2] = w;
    if (taylor) {
      const float tau[4] = {c, s, w, ur};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float acc = cc[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(Cc[i][j], tau[j], acc);
        a.c_back[tb * 4 + i] = acc;
      }
    }
    if (t == a.T - 1) break;
    float cn, sn, nw, nth;
    pendulum_next(pm, c, s, w, ur, cn, sn, nw, nth);
    if (a.F != nullptr)
      pendulum_jacobian_store(pm, c, s, w, ur, cn, sn, nw, a.F + tb * 12, a.f != nullptr ? a.f + tb * 3 : nullptr);
    c = cn;
    s = sn;
    w = nw;
  }
}

__global__ __launch_bounds__(64) void pendulum_rollout_linearize_kernel(const PendulumArgs a) {
  pendulum_rollout_linearize_body(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// The same with four lanes per trajectory (needs 16-byte aligned F, C): the rollout is a chain of T dependent steps on
// B / 64 wavefronts, so its time is the instructions of a step - the four lanes repeat the state update and split
// what has rows: lane `sub` stores x_t[sub], row `sub` of F_t (one 16-byte store) with f_t[sub], and row `sub` of the
// re-centred cost.  Same values as pendulum_jacobian_store.
// lane `sub` of trajectory b (the caller has run a.clear)
__device__ __forceinline__ void pendulum_rollout_linearize4_lane(const PendulumArgs &a, const int b, const int sub) {
  if (b >= a.B) return;
  if (a.done != nullptr && *a.done != 0) return;
  const size_t B = (size_t)a.B;
  float c = a.x_init[b * 3 + 0], s = a.x_init[b * 3 + 1], w = a.x_init[b * 3 + 2];
  const PendulumModel pm = pendulum_model(a.g, a.m, a.l, a.dt, a.max_torque, a.clamp_closed);
  const bool taylor = a.c_back != nullptr;
  const unsigned m0 = sub == 0 ? ~0u : 0u, m1 = sub == 1 ? ~0u : 0u, m2 = sub == 2 ? ~0u : 0u, m3 = sub == 3 ? ~0u : 0u;
  auto pick4 = [&](float e0, float e1, float e2, float e3) {   // element `sub` (bit masks: no exec-mask branches)
    const unsigned r = (__builtin_bit_cast(unsigned, e0) & m0) | (__builtin_bit_cast(unsigned, e1) & m1) |
                       (__builtin_bit_cast(unsigned, e2) & m2) | (__builtin_bit_cast(unsigned, e3) & m3);
    return __builtin_bit_cast(float, r);
  };
  // the inputs of step t + 1 are fetched while step t computes
  float u_nx = a.u[b], c_nx = 0.f;
  float4 C_nx = {0.f, 0.f, 0.f, 0.f};
  if (taylor) {
    C_nx = *reinterpret_cast<const float4 *>(a.C + ((size_t)b * 4 + sub) * 4);
    c_nx = a.c[(size_t)b * 4 + sub];
  }
  for (int t = 0; t < a.T; ++t) {
    const size_t tb = (size_t)t * B + b;
    const float ur = u_nx, cs = c_nx;
    const float4 Cr = C_nx;
    if (t + 1 < a.T) {
      u_nx = a.u[tb + B];
      if (taylor) {
        C_nx = *reinterpret_cast<const float4 *>(a.C + ((tb + B) * 4 + sub) * 4);
        c_nx = a.c[(tb + B) * 4 + sub];
      }
    }
    if (sub < 3) a.x[tb * 3 + sub] = pick4(c, s, w, 0.f);
    if (taylor) {
      float acc = cs;
      acc = fmaf(Cr.x, c, acc); acc = fmaf(Cr.y, s, acc); acc = fmaf(Cr.z, w, acc); acc = fmaf(Cr.w, ur, acc);
      a.c_back[tb * 4 + sub] = acc;
    }
    if (t == a.T - 1) break;
    float cn, sn, nw, nth;
    pendulum_next(pm, c, s, w, ur, cn, sn, nw, nth);
    if (a.F != nullptr && sub < 3) {
      const float inside = pm.clamp_grad(ur);
      const float r2 = c * c + s * s;
      const float dnw[4] = {0.f, pm.dt * pm.kg, 1.f, pm.dt * pm.ku * inside};
      const float dnth[4] = {-s / r2 + pm.dt * dnw[0], c / r2 + pm.dt * dnw[1], pm.dt * dnw[2], pm.dt * dnw[3]};
      const float xin[4] = {c, s, w, ur};
      const float as = pick4(-sn, cn, 0.f, 0.f);
      float row[4], fs = pick4(cn, sn, nw, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned pr = __builtin_bit_cast(unsigned, as * dnth[j]), pw = __builtin_bit_cast(unsigned, dnw[j]);
        row[j] = __builtin_bit_cast(float, (m2 & pw) | (~m2 & pr));   // rows 0, 1: -sn / cn times dnth; row 2: dnw itself
        fs = fmaf(-row[j], xin[j], fs);
      }
      *reinterpret_cast<float4 *>(a.F + tb * 12 + sub * 4) = float4{row[0], row[1], row[2], row[3]};
      if (a.f != nullptr) a.f[tb * 3 + sub] = fs;
    }
    c = cn;
    s = sn;
    w = nw;
  }
}

__global__ __launch_bounds__(256) void pendulum_rollout_linearize4_kernel(const PendulumArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  a.clear.run(g);
  pendulum_rollout_linearize4_lane(a, g >> 2, g & 3);
}

// c_back[t][b][i] = sum_j C[t][b][i][j] tau[t][b][j] + c[t][b][i]        (mpc_step.py:305-317), one lane per (t,b,i)
__global__ __launch_bounds__(256) void taylor_c_kernel(size_t n_rows, int nx, int nu, const float *__restrict__ C,
                                                       const float *__restrict__ c, const float *__restrict__ states,
                                                       const float *__restrict__ controls, float *__restrict__ out) {
  const int ns = nx + nu;
  const size_t total = n_rows * ns;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t row = e / ns;  // (t, b)
    const float *Cr = C + e * ns;
    float acc = c[e];
    for (int j = 0; j < nx; ++j) acc = fmaf(Cr[j], states[row * nx + j], acc);
    for (int j = 0; j < nu; ++j) acc = fmaf(Cr[nx + j], controls[row * nu + j], acc);
    out[e] = acc;
  }
}

// active[t][b][m] = |u-lo| <= 1e-8 | |u-hi| <= 1e-8 (mpc_step.py:363-364);  neg[t][b][:] = -[gx;gu] (:374)
__global__ __launch_bounds__(256) void active_mask_kernel(size_t n_rows, int nx, int nu, const float *__restrict__ u,
                                                          const float *__restrict__ lo, const float *__restrict__ hi,
                                                          const float *__restrict__ gx, const float *__restrict__ gu,
                                                          uint8_t *__restrict__ active, float *__restrict__ neg,
                                                          float *__restrict__ x0, size_t n_x0, float *__restrict__ zero_a,
                                                          float *__restrict__ zero_b, int n_batch,
                                                          const float *__restrict__ detach_norm,
                                                          const int32_t *__restrict__ detach_flag, float detach_eps) {
  const int ns = nx + nu;
  // BoxDDP's detach mask (mpc/box_ddp.py:263-289) applied to the incoming gradient instead of to the solution: when the
  // solve ended with some trajectory above eps (*detach_flag != 0), trajectories whose last step was not below eps
  // (detach_norm[b] >= eps) receive no gradient.  All three live on the device: no read-back decides it.
  const bool gated = detach_norm != nullptr && (detach_flag == nullptr || *detach_flag != 0);
  const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the tiled-cost sums the co-state kernel adds into (CostateArgs::dC_sum / dc_sum): cleared by the chain's first launch
  if (zero_a != nullptr && tid < (size_t)(ns * ns)) zero_a[tid] = 0.f;
  if (zero_b != nullptr && tid < (size_t)ns) zero_b[tid] = 0.f;
  for (size_t e = tid; e < n_rows * nu; e += stride)
    active[e] = (fabsf(u[e] - lo[e]) <= bound_tol(lo[e])) || (fabsf(u[e] - hi[e]) <= bound_tol(hi[e]));
  for (size_t e = tid; e < n_rows * ns; e += stride) {
    const size_t row = e / ns;
    const int j = (int)(e % ns);
    const float v = j < nx ? (gx ? gx[row * nx + j] : 0.f) : (gu ? gu[row * nu + (j - nx)] : 0.f);
    const bool keep = !gated || detach_norm[row % (size_t)n_batch] < detach_eps;
    neg[e] = keep ? -v : 0.f;
  }
  for (size_t e = tid; e < n_x0; e += stride) x0[e] = 0.f;
}

}  // namespace dmpc
