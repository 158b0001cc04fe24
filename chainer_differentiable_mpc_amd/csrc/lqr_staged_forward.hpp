// lqr_staged_forward.hpp - LqrRecursion.forward (lqr/lqr_recursion.py:160-200) for the shapes that run padded inside the
// wavefront-per-trajectory instances (17 to 32 states, or ragged batches of the wide shapes): the closed-loop rollout
//     u_t = K_t x_t + k_t,   x_{t+1} = F_t [x_t; u_t] + f_t
// with run-time dimensions, one wavefront per trajectory and workgroup.  Row per lane at the PROBLEM's positions: lane i < nx
// holds row i of [F_t | f_t], lane nx + m row m of [K_t | k_t]; ONE pass  acc = aff + sum_j row[j] x_t[j]  gives u_t in the K
// lanes and the state part of x_{t+1} in the F lanes,  acc += row[nx + m] u_t[m]  completes x_{t+1}.
//
// What the forward-only container kernel (lqr_kernel<32, 8, 64, ..., kForwardOnly, ..., PAD>) did before: every lane loaded its
// row element by element from HBM and the eight controls were eight 64-lane reductions - 320-460 us at B = 4096, T = 50
// whatever the problem's size.  Here a step's blocks [F | f | K | k] travel to a three-slot LDS ring as the contiguous runs
// they are (dma_run_floats: 4-byte LDS-DMA, run-time lengths, any alignment), two steps ahead, and the rows are read from
// the slot.  Every step issues the same number of DMA instructions (past the horizon the last blocks are fetched again, never
// consumed), so "all but the youngest step's" is a counted s_waitcnt; the two stores of a step in between only make it
// stricter.  The clamped rollout (LQR_active) stays on the container kernel.
#pragma once
#include "dma_gather.hpp"
#include "lqr_kernels.hpp"

namespace dmpc {

struct LqrStagedFwdSlot {
  int F, f, K, k, floats, dmas;   // offsets in floats (regions are multiples of 64), DMA instructions per step
};
__host__ __device__ inline LqrStagedFwdSlot lqr_staged_fwd_slot(int nx, int nu, bool has_f) {
  const int ns = nx + nu;
  LqrStagedFwdSlot s;
  int o = 0, n = 0;
  auto region = [&](int len) { const int at = o; o += (len + 63) / 64 * 64; n += (len + 63) / 64; return at; };
  s.F = region(nx * ns);
  s.f = has_f ? region(nx) : 0;
  s.K = region(nu * nx);
  s.k = region(nu);
  s.floats = o;
  s.dmas = n;
  return s;
}
constexpr int kStagedFwdDepth = 3;
inline size_t lqr_staged_fwd_lds_bytes(int nx, int nu) { return (size_t)kStagedFwdDepth * lqr_staged_fwd_slot(nx, nu, true).floats * 4; }

__global__ __launch_bounds__(64) void lqr_staged_forward_kernel(const LqrArgs a, const int nx, const int nu) {
  const int ns = nx + nu;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int T = a.T;
  const size_t B = (size_t)a.B;
  const bool has_f = a.f != nullptr;
  const float *Ks = a.Ks != nullptr ? a.Ks : a.wsK;
  const float *ks = a.Ks != nullptr ? a.ks : a.wsk;
  const LqrStagedFwdSlot L = lqr_staged_fwd_slot(nx, nu, has_f);
  extern __shared__ float lds[];
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char *)lds);
  auto issue = [&](int t) {   // step t (clamped to the horizon) into slot t % depth
    const int slot = t % kStagedFwdDepth;
    const int tk = t < T ? t : T - 1;
    const int tf = tk < T - 1 ? tk : (T > 1 ? T - 2 : 0);   // there is no F_{T-1}: F_{T-2} again (never consumed)
    const size_t tbk = (size_t)tk * B + b, tbf = (size_t)tf * B + b;
    const unsigned dst = ring_addr + (unsigned)(slot * L.floats) * 4u;
    dma_run_floats(a.F + tbf * nx * ns, dst + L.F * 4, nx * ns, lane);   // (the launcher asks for T >= 2: F exists)
    if (has_f) dma_run_floats(a.f + tbf * nx, dst + L.f * 4, nx, lane);
    dma_run_floats(Ks + tbk * nu * nx, dst + L.K * 4, nu * nx, lane);
    dma_run_floats(ks + tbk * nu, dst + L.k * 4, nu, lane);
  };
  const bool f_lane = lane < nx, g_lane = lane >= nx && lane < ns;
  const int row_off = f_lane ? L.F + lane * ns : (g_lane ? L.K + (lane - nx) * nx : L.K);   // idle lanes re-read a gain row
  const int aff_off = f_lane ? (has_f ? L.f + lane : -1) : (g_lane ? L.k + (lane - nx) : L.k);
  float xv = f_lane ? a.x_init[(size_t)b * nx + lane] : 0.f;
  bool bad = false;
  issue(0);
  issue(1);
  for (int t = 0; t < T; ++t) {
    const size_t tb = (size_t)t * B + b;
    wait_vmcnt_at_most(L.dmas);                               // slot t has landed: only step t+1's requests may be on their way
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // slot t-1 has been read: it takes step t+2
    issue(t + 2);
    const float *S = lds + (t % kStagedFwdDepth) * L.floats;
    const float *row = S + row_off;
    float acc = aff_off >= 0 ? S[aff_off] : 0.f;
#pragma unroll 4
    for (int j = 0; j < nx; ++j)
      acc = fmaf(row[j], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv), j)), acc);   // :177, :189
    const float uo = acc;
    if (f_lane) a.x[tb * nx + lane] = xv;
    if (g_lane) a.u[tb * nu + (lane - nx)] = uo;
    bad = bad || (g_lane && !(fabsf(uo) <= 3.0e38f)) || (f_lane && !(fabsf(xv) <= 3.0e38f));
    if (t < T - 1) {   // uniform
      const float *frow = S + (f_lane ? row_off : L.F);   // (the other lanes: row 0 of F, never used)
      float s = acc;
#pragma unroll 4
      for (int m = 0; m < nu; ++m)
        s = fmaf(frow[nx + m], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uo), nx + m)), s);   // :189
      if (f_lane) xv = s;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the fetches past the horizon have landed before the LDS goes back)
  if (bad && a.info != nullptr) atomicOr(&a.info[b], 2);
}

}  // namespace dmpc
