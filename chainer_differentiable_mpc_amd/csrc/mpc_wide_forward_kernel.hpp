// mpc_wide_forward_kernel.hpp - MPCstep.forward_rec (mpc/mpc_step.py:175-286) for a LinDx and a QuadCost at the shapes of the
// wide row kernels (16 to 31 elements of tau = [x; u], nx <= 16): the clamped rollout under the gains of backward_rec and the
// per-trajectory line search on the true cost.  The arithmetic and the decisions are mpc_forward_rec_kernel's
// (mpc_kernels.hpp: the search decided on the cost DIFFERENCE summed per timestep, bounds snapped, a wave-uniform pass loop);
// what changes is the layout - four trajectories per wavefront with tau in two registers per lane:
//   lane j        : elements j and 16 + j of [new_x_t; new_u_t] and of the iterate, rows j and 16 + j of C_t, c_t
//   lane i < NX   : row i of [F_t | f_t], column i of K_t (the controls are sums over the 16 lanes, the same in all of them)
// and the inputs of a timestep come through a two-slot LDS ring by per-lane gather DMA into ONE register set
// (costate_wide_kernel.hpp's pipeline).  Before: the runtime-dimension kernel (0.68 ms per call at (12,4), B = 4096, T = 50).
// Needs B >= 4, 16-byte aligned arrays, T >= 2.
#pragma once
#include "dma_gather.hpp"
#include "lqr_dma_kernel.hpp"   // lds_byte_address, wait_vmcnt
#include "lqr_wide_kernel.hpp"  // padded_read
#include "mpc_kernels.hpp"

namespace dmpc {

template <int NX, int NU, int DB>
struct MpcWideFwdLayout {
  static constexpr int NS = NX + NU;
  // 16-byte chunks of one wave-step (four trajectories): [C | c | F | f | Ks | ks | u | lower | upper | x]
  static constexpr int CH_C = 0, CH_c = CH_C + NS * NS, CH_F = CH_c + NS, CH_f = CH_F + NX * NS, CH_K = CH_f + NX;
  static constexpr int CH_k = CH_K + NU * NX, CH_u = CH_k + NU, CH_lo = CH_u + NU, CH_hi = CH_lo + NU, CH_x = CH_hi + NU;
  static constexpr int CH_END = CH_x + NX;
  static constexpr int kDma = (CH_END + 63) / 64;   // gather DMAs per step; padding lanes repeat chunk 0 of C
  static constexpr int SLOT = kDma * 256;           // floats per wave and timestep
  static constexpr size_t lds_bytes() { return (size_t)4 * DB * SLOT * 4 + 64; }   // + a zero per wave (PAD)
};

// PAD: container for a smaller problem (a.nx_log <= NX, a.nu_log <= NU; lqr_wide_kernel<..., PAD>): the arrays come into
// container-sized regions of the slot as they are, the reads place element i of tau at lane i (state) or NX + m (control),
// everything outside the problem is 0 and the unused controls stay 0 inside the box [-1, 1].  Needs B % 4 == 0.
template <int NX, int NU, int DB, bool PAD = false>
__global__ __launch_bounds__(256) void mpc_wide_forward_kernel(const MpcFwdArgs a) {
  using Lay = MpcWideFwdLayout<NX, NU, DB>;
  using G = Group<16>;
  constexpr int NS = NX + NU, N1 = NS - 16;
  static_assert(NX <= 16 && NS >= 16 && NS <= 31, "tau in two registers (the second may be empty), the states in the first");
  static_assert((DB - 1) * Lay::kDma <= 63, "ring too deep for vmcnt");

  if (a.done != nullptr && *a.done != 0) return;  // uniform: the iLQR loop has stopped
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane64 = threadIdx.x & 63;
  const int r = lane64 >> 4;
  const int lane = lane64 & 15;
  int b0 = ((int)blockIdx.x * 4 + wave) * 4;
  if (b0 > a.B - 4) b0 = a.B - 4;  // the last wave overlaps its neighbour instead of running ragged (same results twice)
  b0 = __builtin_amdgcn_readfirstlane(b0);
  const int b = b0 + r;

  extern __shared__ float lds[];
  float *ring = lds + wave * (DB * Lay::SLOT);
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_address(ring));

  float *zo = lds + 4 * (DB * Lay::SLOT) + wave * 4;   // PAD: a zero for the padded reads (this wave's own)
  if constexpr (PAD) {
    if (lane64 == 0) zo[0] = 0.f;
  }
  // the problem's own dimensions, and where container element e of tau lies in them (-1: padding)
  const int nx = PAD ? a.nx_log : NX, nu = PAD ? a.nu_log : NU, ns = nx + nu;
  auto logical = [&](int e) -> int { return e < NX ? (e < nx ? e : -1) : (e - NX < nu ? nx + (e - NX) : -1); };
  const bool is_x = lane < nx;
  const bool is_t1 = lane < N1;                 // this lane holds an element of tau / a row of C in its second register
  const int lane_x = is_x ? lane : nx - 1;      // clamped: rows re-read by the idle lanes, never used
  const int e1 = is_t1 ? 16 + lane : NS - 1;    // element / row of the second register (clamped)
  const int le0 = PAD ? logical(lane) : lane, le1 = PAD ? (is_t1 ? logical(e1) : -1) : e1;   // ... in the problem's own numbering

  // per-lane source pointers of the gather groups at t = 0 (32-bit time strides: the launcher checks them); F and f have
  // T - 1 slices: on the last advance their lanes stay (the step t = T - 1 fetches slice T - 2 again, never consumed)
  unsigned long long ptr0[Lay::kDma], ptr[Lay::kDma];
  unsigned str[Lay::kDma], dynF = 0;
#pragma unroll
  for (int q = 0; q < Lay::kDma; ++q) {
    const int g = q * 64 + lane64;
    const char *base = (const char *)a.C;
    size_t per = (size_t)ns * ns * 4;
    int g0 = g;   // absent arrays and padding lanes: chunk 0 of C again
    bool isF = false;
    if (g < Lay::CH_c) { g0 = Lay::CH_C; }
    else if (g < Lay::CH_F) { base = (const char *)a.c; per = (size_t)ns * 4; g0 = Lay::CH_c; }
    else if (g < Lay::CH_f) { base = (const char *)a.F; per = (size_t)nx * ns * 4; g0 = Lay::CH_F; isF = true; }
    else if (g < Lay::CH_K) { if (has_f) { base = (const char *)a.f; per = (size_t)nx * 4; g0 = Lay::CH_f; isF = true; } }
    else if (g < Lay::CH_k) { base = (const char *)a.Ks; per = (size_t)nu * nx * 4; g0 = Lay::CH_K; }
    else if (g < Lay::CH_u) { base = (const char *)a.ks; per = (size_t)nu * 4; g0 = Lay::CH_k; }
    else if (g < Lay::CH_lo) { base = (const char *)a.controls; per = (size_t)nu * 4; g0 = Lay::CH_u; }
    else if (g < Lay::CH_hi) { base = (const char *)a.lower; per = (size_t)nu * 4; g0 = Lay::CH_lo; }
    else if (g < Lay::CH_x) { base = (const char *)a.upper; per = (size_t)nu * 4; g0 = Lay::CH_hi; }
    else if (g < Lay::CH_END) { base = (const char *)a.states; per = (size_t)nx * 4; g0 = Lay::CH_x; }
    // (PAD: every array sits at the start of its container-sized region; the chunks behind its end fetch its chunk 0 again)
    const int gc = (!PAD || (size_t)(g - g0) * 16 < 4 * per) ? g - g0 : 0;
    ptr0[q] = (unsigned long long)base + (size_t)b0 * per + (size_t)gc * 16 - (unsigned long long)(q % 4) * 1024u;
    str[q] = (unsigned)(B * per);
    dynF |= isF ? (1u << q) : 0u;
  }
  int ti = 0;  // timesteps the pointers may still advance
  auto issue_next = [&](int slot) __attribute__((always_inline)) {
    const unsigned dst = ring_addr + (unsigned)slot * (Lay::SLOT * 4);
    static_for<0, Lay::kDma>([&](auto q) {
      if constexpr (q.value % 4 == 0) set_m0(dst + (unsigned)q.value * 1024u);
      dma16_gather<(q.value % 4) * 1024>(ptr[q.value]);
    });
    if (ti > 0) {  // past the horizon the last blocks are fetched again (never consumed): the count per step stays exact
      if (ti == 1) {
#pragma unroll
        for (int q = 0; q < Lay::kDma; ++q) ptr[q] += ((dynF >> q) & 1u) ? 0ull : (unsigned long long)str[q];
      } else {
#pragma unroll
        for (int q = 0; q < Lay::kDma; ++q) ptr[q] += (unsigned long long)str[q];
      }
      --ti;
    }
  };
  // per-lane LDS indices (floats, relative to a slot)
  const int i_x = Lay::CH_x * 4 + r * nx + lane_x;
  const int i_K = Lay::CH_K * 4 + r * nu * nx + lane_x;                         // + m * nx: K[m][lane_x]
  const int r0 = le0 >= 0 ? le0 : 0, r1 = le1 >= 0 ? le1 : 0;                   // this lane's rows of C (clamped)
  const int i_C0 = Lay::CH_C * 4 + (r * ns + r0) * ns, i_C1 = Lay::CH_C * 4 + (r * ns + r1) * ns;
  const int i_c0 = Lay::CH_c * 4 + r * ns + r0, i_c1 = Lay::CH_c * 4 + r * ns + r1;
  const int i_F = Lay::CH_F * 4 + (r * nx + lane_x) * ns, i_f = Lay::CH_f * 4 + r * nx + lane_x;

  float xt, kvx[NU], ksv[NU], uc[NU], lb[NU], ub[NU], Crow0[NS], Crow1[NS], c0, c1, Frow[NS], fi;
  auto read_slot = [&](const float *slot) __attribute__((always_inline)) {
    xt = slot[i_x];
    static_for<0, NU>([&](auto m) {
      if constexpr (PAD) {   // an unused control: K = 0, k = 0, u = 0 in the box [-1, 1]
        const bool used = m.value < nu;   // uniform
        const int mc = used ? m.value : 0;
        kvx[m.value] = padded_read(slot + i_K + mc * nx, used, zo);
        ksv[m.value] = padded_read(slot + Lay::CH_k * 4 + r * nu + mc, used, zo);
        uc[m.value] = padded_read(slot + Lay::CH_u * 4 + r * nu + mc, used, zo);
        const float l_ = slot[Lay::CH_lo * 4 + r * nu + mc], u_ = slot[Lay::CH_hi * 4 + r * nu + mc];
        lb[m.value] = used ? l_ : -1.f;
        ub[m.value] = used ? u_ : 1.f;
      } else {
        kvx[m.value] = slot[i_K + m.value * NX];
        ksv[m.value] = slot[Lay::CH_k * 4 + r * NU + m.value];
        uc[m.value] = slot[Lay::CH_u * 4 + r * NU + m.value];
        lb[m.value] = slot[Lay::CH_lo * 4 + r * NU + m.value];
        ub[m.value] = slot[Lay::CH_hi * 4 + r * NU + m.value];
      }
    });
    static_for<0, NS>([&](auto j) {
      if constexpr (PAD) {
        const int lj = logical(j.value);   // uniform
        const int jc = lj >= 0 ? lj : 0;
        Crow0[j.value] = padded_read(slot + i_C0 + jc, lj >= 0 && le0 >= 0, zo);
        Crow1[j.value] = padded_read(slot + i_C1 + jc, lj >= 0 && le1 >= 0, zo);
        Frow[j.value] = padded_read(slot + i_F + jc, lj >= 0 && is_x, zo);
      } else {
        Crow0[j.value] = slot[i_C0 + j.value];
        Crow1[j.value] = slot[i_C1 + j.value];
        Frow[j.value] = slot[i_F + j.value];
      }
    });
    c0 = slot[i_c0];
    c1 = slot[i_c1];
    fi = has_f ? slot[i_f] : 0.f;
    if constexpr (PAD) {
      xt = is_x ? xt : 0.f;
      c0 = le0 >= 0 ? c0 : 0.f;
      c1 = le1 >= 0 ? c1 : 0.f;
      fi = is_x ? fi : 0.f;
    }
  };
  // (row . v): the elements of v broadcast from their lanes, two chains
  auto row_dot = [&](const float (&row)[NS], const float (&v)[2], float init) __attribute__((always_inline)) {
    float qa = init, qb = 0.f;
    static_for<0, NS>([&](auto j) {
      constexpr int h = j.value / 16, l = j.value % 16;
      const float vj = G::template bcast<l>(v[h]);
      if constexpr (j.value % 2 == 0) qa = fmaf(vj, row[j.value], qa);
      else qb = fmaf(vj, row[j.value], qb);
    });
    return qa + qb;
  };

  // OLD_COST and the test `current_cost > OLD_COST` on the per-timestep difference: see mpc_forward_rec_kernel
  float old_cost = 0.f, alpha = 1.0f, cost = 0.f;
  int n_pass = 0;
  bool worse = true;
  bool searching = a.ls_cap > 0;   // this trajectory's search goes on                      :196
  for (int pass_idx = 0; __any(searching); ++pass_idx) {
    float xh = 0.f;                // new_x_t[lane] (lanes < NX)
    float cost_p = 0.f, old_p = 0.f, delta = 0.f;
    auto step = [&](int t) __attribute__((always_inline)) {
      const size_t tb = (size_t)t * B + b;
      const float dxv = is_x ? xh - xt : 0.f;
      float un[NU];
#pragma unroll
      for (int m = 0; m < NU; ++m) {
        float v = fmaf(alpha, ksv[m], group_sum<16>(is_x ? kvx[m] * dxv : 0.f)) + uc[m];   // :209-219
        v = fminf(fmaxf(v, lb[m]), ub[m]);                                                   // :221
        v = (v - lb[m] <= bound_tol(lb[m])) ? lb[m] : v;
        un[m] = (ub[m] - v <= bound_tol(ub[m])) ? ub[m] : v;
      }
      // [new_x_t ; new_u_t] and the iterate [x_t ; u_t], elements `lane` and 16 + lane
      float tau[2], tau0[2], dt[2];
      bool valid[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = h == 0 ? lane : e1;
        valid[h] = PAD ? (h == 0 ? le0 >= 0 : le1 >= 0) : (h == 0 ? true : is_t1);   // this register holds an element of tau
        float tv = e < NX ? xh : 0.f, t0v = e < NX ? xt : 0.f;
        if (h == 1 && NX == 16) { tv = 0.f; t0v = 0.f; }   // (the second register holds controls only)
#pragma unroll
        for (int m = 0; m < NU; ++m) {
          tv = (e == NX + m) ? un[m] : tv;
          t0v = (e == NX + m) ? uc[m] : t0v;
        }
        tau[h] = valid[h] ? tv : 0.f;
        tau0[h] = valid[h] ? t0v : 0.f;
        dt[h] = tau[h] - tau0[h];
      }
      const float qi0 = row_dot(Crow0, tau, 0.f), qd0 = row_dot(Crow0, dt, 0.f);
      const float qi1 = row_dot(Crow1, tau, 0.f), qd1 = row_dot(Crow1, dt, 0.f);
      // per-lane partial sums over the timesteps; the lanes are added up once after the pass     :246-251, util.py:162-198
      const float obj_l = tau[0] * fmaf(0.5f, qi0, c0) + (valid[1] ? tau[1] * fmaf(0.5f, qi1, c1) : 0.f);
      cost_p += obj_l;
      delta += fmaf(dt[0], fmaf(0.5f, qi0, c0), 0.5f * tau0[0] * qd0) +
               (valid[1] ? fmaf(dt[1], fmaf(0.5f, qi1, c1), 0.5f * tau0[1] * qd1) : 0.f);
      if (pass_idx == 0)   // cost of the iterate, from C tau = C tau' - C d                                 :191
        old_p += tau0[0] * fmaf(0.5f, qi0 - qd0, c0) + (valid[1] ? tau0[1] * fmaf(0.5f, qi1 - qd1, c1) : 0.f);
      float obj = 0.f;
      if (a.objs != nullptr) obj = group_sum<16>(obj_l);
      if (searching) {  // outputs are overwritten by later passes; the last one is the accepted one
        const bool u0 = valid[0] && lane >= NX, u1 = valid[1] && e1 >= NX;      // this register holds a control: lanes NX.. / 16 + lane
        if (is_x) a.x[tb * nx + lane] = xh;
        if (u0) a.u[tb * nu + (lane - NX)] = tau[0];
        if (u1) a.u[tb * nu + (e1 - NX)] = tau[1];
        if (a.objs != nullptr && lane == 0) a.objs[tb] = obj;
        if (a.u_first != nullptr && pass_idx == 0) {
          if (u0) a.u_first[tb * nu + (lane - NX)] = tau[0];
          if (u1) a.u_first[tb * nu + (e1 - NX)] = tau[1];
        }
      }
      if (t < T - 1) {  // new_x_{t+1} = F_t [new_x; new_u] + f_t under the TRUE dynamics   :229-236
        const float acc = row_dot(Frow, tau, fi);
        xh = is_x ? acc : 0.f;
      }
    };
    // ONE register set (costate_wide_kernel.hpp): the slot of step t is waited for and read, then - once the reads are in -
    // refilled with step t + DB; the stores of a pass only make the counted wait more conservative
#pragma unroll
    for (int q = 0; q < Lay::kDma; ++q) ptr[q] = ptr0[q];
    ti = T - 1;
    static_for<0, DB>([&](auto j) { issue_next(j.value); });
    for (int t0 = 0; t0 < T; t0 += DB) {
      static_for<0, DB>([&](auto j) {
        const int t = t0 + j.value;
        if (t < T) {
          wait_vmcnt<(DB - 1) * Lay::kDma>();
          read_slot(ring + j.value * Lay::SLOT);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are in before it is refilled
          issue_next(j.value);
          if (t == 0) xh = is_x ? xt : 0.f;                     // new_x[0] = states[0]     :198
          step(t);
        }
      });
    }
    wait_vmcnt<0>();   // the ring is refilled from t = 0 by the next pass
    cost_p = group_sum<16>(cost_p);
    delta = group_sum<16>(delta);
    if (pass_idx == 0) old_cost = group_sum<16>(old_p);
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
  if (lane == 0) {
    a.costs[b] = cost;
    if (a.old_costs != nullptr) a.old_costs[b] = old_cost;
    a.alphas[b] = alpha;
    a.n_ls[b] = n_pass;
    if (a.info != nullptr && info_bits != 0) atomicOr(&a.info[b], info_bits);
  }
}

}  // namespace dmpc
