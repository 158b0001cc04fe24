// costate_staged_kernel.hpp - the co-state sweep and the outer products of the analytic KKT gradient (DiffLqr.backward,
// lqr/differentiable_lqr.py:85-134; MPCstep.backward, mpc/mpc_step.py:383-446) with run-time dimensions (nx + nu <= 63), one
// wavefront per trajectory and workgroup - the arithmetic of costate_generic_kernel (costate_kernels.hpp), what changes is
// how the data moves:
//   * a step's blocks [C | c | F | x | u | dx | du | r] travel to a three-slot LDS ring as the contiguous runs they are
//     (dma_run_floats: 4-byte LDS-DMA, run-time lengths, any alignment), two steps ahead; lane i < nx reads row i of C_t and
//     column i of F_t[:, :nx] from the slot.  Every step issues the same number of DMA instructions (past t = 0 the blocks of
//     step 0 are fetched again, never consumed; F_{T-1} does not exist: F_{T-2} twice), so the wait is a counted s_waitcnt;
//   * tau, dtau live in the lanes (element j in lane j), lambda, d_lambda in lanes < nx: every broadcast is a v_readlane;
//   * dC_t and dF_t are stored element per lane, 64 consecutive floats per instruction, with the row / column of an element
//     stepped incrementally.
// Before, the shapes with 17 to 32 states ran padded inside costate_kernel<32, 8, 64, PAD>: every lane fetched its row of C
// element by element from HBM and stored its row of dC / dF the same way - 1.39 ms per sweep at (20,6), B = 4096, T = 50.
#pragma once
#include "costate_args.hpp"
#include "dma_gather.hpp"   // dma_run_floats, wait_vmcnt_at_most

namespace dmpc {

struct CostateStagedSlot {
  int C, c, F, x, u, dx, du, r, floats, dmas;
};
__host__ __device__ inline CostateStagedSlot costate_staged_slot(int nx, int nu, int r_cols) {
  const int ns = nx + nu;
  CostateStagedSlot s;
  int o = 0, n = 0;
  auto region = [&](int len) { const int at = o; o += (len + 63) / 64 * 64; n += (len + 63) / 64; return at; };
  s.C = region(ns * ns); s.c = region(ns); s.F = region(nx * ns);
  s.x = region(nx); s.u = region(nu); s.dx = region(nx); s.du = region(nu);
  s.r = region(r_cols ? r_cols : ns);
  s.floats = o;
  s.dmas = n;
  return s;
}
constexpr int kCostateStagedDepth = 3;
inline size_t costate_staged_lds_bytes(int nx, int nu, int r_cols) {
  return (size_t)kCostateStagedDepth * costate_staged_slot(nx, nu, r_cols).floats * 4;
}

__device__ __forceinline__ float lane_value(float v, int l) {   // l uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__global__ __launch_bounds__(64) void costate_staged_kernel(const CostateArgs a, const int nx, const int nu) {
  const int ns = nx + nu;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const int rc = a.r_cols ? a.r_cols : ns;
  const CostateStagedSlot L = costate_staged_slot(nx, nu, a.r_cols);
  extern __shared__ float lds[];
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char *)lds);
  const float wa = 0.5f, wb = a.dC_mode == 0 ? 1.0f : 0.5f;
  // step t is the n-th fetch, n = T - 1 - t, into slot n % depth
  auto issue = [&](int n) {
    const int slot = n % kCostateStagedDepth;
    const int t = n < T ? T - 1 - n : 0;
    const int tf = t < T - 1 ? t : T - 2;   // (the launcher asks for T >= 2)
    const size_t tb = (size_t)t * B + b, tbf = (size_t)tf * B + b;
    const unsigned dst = ring_addr + (unsigned)(slot * L.floats) * 4u;
    dma_run_floats(a.C + tb * ns * ns, dst + L.C * 4, ns * ns, lane);
    dma_run_floats(a.c + tb * ns, dst + L.c * 4, ns, lane);
    dma_run_floats(a.F + tbf * nx * ns, dst + L.F * 4, nx * ns, lane);
    dma_run_floats(a.x + tb * nx, dst + L.x * 4, nx, lane);
    dma_run_floats(a.u + tb * nu, dst + L.u * 4, nu, lane);
    dma_run_floats(a.dx + tb * nx, dst + L.dx * 4, nx, lane);
    dma_run_floats(a.du + tb * nu, dst + L.du * 4, nu, lane);
    dma_run_floats(a.r + tb * rc, dst + L.r * 4, rc, lane);
  };
  const bool is_x = lane < nx, is_tau = lane < ns;
  const int lane_x = is_x ? lane : 0;        // clamped: the idle lanes re-read row / column 0 (never used)
  const int q64 = 64 / ns, r64 = 64 % ns;    // an element 64 further on: q64 rows down, r64 columns right (with carry)
  float lam = 0.f, dlam = 0.f;               // lambda_{t+1}[lane], d_lambda_{t+1}[lane]  (lanes < nx)
  // vector-memory operations retire in order: behind this step's fetch (issued two steps ago) lie the row stores of the two
  // steps since and the next step's fetch - that many may still be on their way when the slot is read (the dc / df stores
  // are left out of the count: waiting for a few more is safe, for fewer is not)
  const int stores_dF = a.dF != nullptr ? (nx * ns + 63) / 64 : 0, stores_dC = a.dC != nullptr ? (ns * ns + 63) / 64 : 0;
  int stores_1 = 0, stores_2 = 0;   // ... of the step before, and of the one before that
  issue(0);
  issue(1);
  for (int n = 0; n < T; ++n) {
    const int t = T - 1 - n;
    const size_t tb = (size_t)t * B + b;
    wait_vmcnt_at_most(L.dmas + stores_1 + stores_2);
    stores_2 = stores_1;
    stores_1 = (t < T - 1 ? stores_dF : 0) + stores_dC;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot of the step before has been read: it takes step n + 2
    issue(n + 2);
    const float *S = lds + (n % kCostateStagedDepth) * L.floats;
    const float tau = is_tau ? (is_x ? S[L.x + lane] : S[L.u + (lane - nx)]) : 0.f;
    const float dtau = is_tau ? (is_x ? S[L.dx + lane] : S[L.du + (lane - nx)]) : 0.f;
    // ---- dF_t and df (they use lambda_{t+1}, d_lambda_{t+1})             differentiable_lqr.py:130-133
    if (t < T - 1) {   // uniform
      if (a.dF != nullptr) {
        float *out = a.dF + tb * nx * ns;
        int k = lane / ns, j = lane % ns;
        for (int e = 0; e < nx * ns; e += 64) {   // uniform trip count: one store instruction per 64 elements
          // lam[k], dlam[k], tau[j], dtau[j] at per-lane indices: __shfl (ds_bpermute) of the lane-resident vectors
          const int kk = k < nx ? k : 0;
          const float lk = __shfl(lam, kk), dlk = __shfl(dlam, kk), tj = __shfl(tau, j), dtj = __shfl(dtau, j);
          if (e + lane < nx * ns) out[e + lane] = a.out_sign * fmaf(dlk, tj, lk * dtj);
          j += r64; k += q64;
          if (j >= ns) { j -= ns; ++k; }
        }
      }
      if (a.df != nullptr && a.df_shift == 1 && is_x) a.df[tb * nx + lane] = a.out_sign * dlam;
    }
    // ---- dC_t, dc_t                                                          :128-129
    if (a.dC != nullptr) {
      float *out = a.dC + tb * ns * ns;
      int i = lane / ns, j = lane % ns;
      for (int e = 0; e < ns * ns; e += 64) {
        const int ii = i < ns ? i : 0;
        const float ti = __shfl(tau, ii), dti = __shfl(dtau, ii), tj = __shfl(tau, j), dtj = __shfl(dtau, j);
        if (e + lane < ns * ns) out[e + lane] = a.out_sign * fmaf(wa * dti, tj, (wb * ti) * dtj);
        j += r64; i += q64;
        if (j >= ns) { j -= ns; ++i; }
      }
    }
    if (a.dc != nullptr && is_tau) a.dc[tb * ns + lane] = a.out_sign * dtau;
    // ---- lambda_t, d_lambda_t                                                 :92,102 / :115,124
    float nl = S[L.c + lane_x], ndl = a.r_sign * S[L.r + lane_x];
    const float *Cr = S + L.C + lane_x * ns;
#pragma unroll 4
    for (int j = 0; j < ns; ++j) {
      const float cij = Cr[j];
      nl = fmaf(cij, lane_value(tau, j), nl);
      ndl = fmaf(cij, lane_value(dtau, j), ndl);
    }
    if (t < T - 1) {   // uniform
      const float *Fp = S + L.F + lane_x;
#pragma unroll 4
      for (int k = 0; k < nx; ++k) {
        const float fki = Fp[k * ns];
        nl = fmaf(fki, lane_value(lam, k), nl);
        ndl = fmaf(fki, lane_value(dlam, k), ndl);
      }
    }
    lam = is_x ? nl : 0.f;
    dlam = is_x ? ndl : 0.f;
    if (a.df != nullptr && a.df_shift == 0 && t < T - 1 && is_x) a.df[tb * nx + lane] = a.out_sign * dlam;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the fetches past t = 0 have landed before the LDS goes back)
  if (a.dx0 != nullptr && is_x) a.dx0[(size_t)b * nx + lane] = a.out_sign * dlam;
}

}  // namespace dmpc
