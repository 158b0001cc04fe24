// box_ddp_kernels.hpp - the bookkeeping of the outer box-DDP loop (BoxDDP.forward, mpc/box_ddp.py:93-291) on the
// device, so that a whole solve (max_iter MPC steps) is one chain of launches without a host round trip:
//
//   lin_rollout_kernel       get_traj under a LinDx (util.py:239-277): x_{t+1} = F_t [x_t; u_t] + f_t
//   box_ddp_select_kernel    one workgroup: full_du_norm of the step (mpc_step.py:260-263, with the reference's
//                            reshape of a [T,nu,B] array to [B, T*nu] unless strict), per-sample "best so far" test
//                            (box_ddp.py:200-209), the stop tests (:223-230) and the loop state
//   box_ddp_keep_kernel      best x / u <- the step's x / u where the sample improved
//
// The loop state lives in `state[8]` (int32): [0] done, [1] n_iter, [2] status (1 Converged, 2 Not improved lim,
// 3 Not Converged), [3] n_not_improved; the last launch of the chain (box_ddp_summary_kernel) adds what the caller
// would otherwise reduce with a dozen small launches of its own: [4] NaN among the initial controls, [5] some
// lower > upper (MPCstep's input asserts, mpc_step.py:133-138), [6] trajectories flagged non-finite, [7] the best
// iterate's full_du_norm exceeds eps somewhere (box_ddp.py:263).  Once `done` is set every later launch of the chain returns at once (the
// MPC kernels test the same flag), so the host may enqueue max_iter iterations blindly and synchronise once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mpc_kernels.hpp"

namespace dmpc {

enum { kDdpDone = 0, kDdpIter = 1, kDdpStatus = 2, kDdpNotImproved = 3 };

__global__ __launch_bounds__(64) void lin_rollout_kernel(int T, int B, int nx, int nu, const float *__restrict__ x_init,
                                                         const float *__restrict__ u, const float *__restrict__ F,
                                                         const float *__restrict__ f, float *__restrict__ x,
                                                         const int32_t *__restrict__ done, const ChainClear clear) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  clear.run(b);
  if (b >= B) return;
  if (done != nullptr && *done != 0) return;
  const int ns = nx + nu;
  const size_t Bs = (size_t)B;
  for (int i = 0; i < nx; ++i) x[(size_t)b * nx + i] = x_init[(size_t)b * nx + i];
  for (int t = 0; t + 1 < T; ++t) {
    const size_t tb = (size_t)t * Bs + b;
    const float *xt = x + tb * nx, *ut = u + tb * nu, *Ft = F + tb * nx * ns;
    float *xn = x + (tb + Bs) * nx;
    for (int i = 0; i < nx; ++i) {
      float acc = f != nullptr ? f[tb * nx + i] : 0.f;
      for (int j = 0; j < nx; ++j) acc = fmaf(Ft[i * ns + j], xt[j], acc);
      for (int j = 0; j < nu; ++j) acc = fmaf(Ft[i * ns + nx + j], ut[j], acc);
      xn[i] = acc;
    }
  }
}

struct DdpSelectArgs {
  int it, T, B, nx, nu;
  int max_iter, not_improved_lim, scrambled;
  float eps, best_cost_eps;
  const float *u_old, *u_first;  // controls the step started from / of its first (alpha = 1) pass   [T,B,nu]
  const float *costs;            // [B] cost of the step's result
  float *best_costs, *best_norm, *last_norm;  // [B]
  int32_t *keep;                 // [B] 1 where the step's x / u replace the best ones
  int32_t *state;
  // small problems: the same workgroup also moves the kept trajectories (no box_ddp_keep_kernel launch)
  int copy_here;
  const float *x_new, *u_new;
  float *best_x, *best_u;
  // one-launch iterations (box_ddp_pendulum_iter_kernel): the flags this iteration's sweep and search left in a staging word
  // per trajectory (they may have run ahead of the stop flag) are merged into `info` here, i.e. only if the iteration counts
  const int32_t *info_stage = nullptr;
  int32_t *info = nullptr;
};

constexpr int kDdpCopyHereMaxB = 2048;
constexpr size_t kDdpCopyHereMaxElems = (size_t)128 * 1024;   // trajectory elements one workgroup moves per iteration
constexpr int kDdpSelectThreads = 1024;
constexpr int kDdpNormTile = 8192;   // floats of LDS the norm phase stages control differences in

// NT threads of a workgroup; NXC, NUC: nx, nu when known at compile time (0: the runtime values of `a`).
// n_parts > 1: workgroup `part` handles the rows [part * rpp, (part + 1) * rpp) - norms, "best so far", trajectory copy -
// and the batch-wide tests (max norm, any improvement, NaN) are combined through three words at `psync` (zero before
// the first iteration): atomic max / or, then a ticket; the workgroup that draws the last ticket has everyone's
// contribution, updates the loop state and clears the words for the next iteration.  max and or do not depend on the
// order of arrival, so the result is the single workgroup's.
template <int NT, int NXC = 0, int NUC = 0>
__device__ __forceinline__ void box_ddp_select_body(const DdpSelectArgs &a, const int part = 0, const int n_parts = 1,
                                                    unsigned *psync = nullptr) {
  constexpr int kDdpSelectThreads = NT;
  const int rpp = n_parts > 1 ? (((a.B + n_parts - 1) / n_parts + 3) & ~3) : a.B;   // rows per part: whole quads
  const int rb = part * rpp < a.B ? part * rpp : a.B, re = rb + rpp < a.B ? rb + rpp : a.B;
  __shared__ float s_max[kDdpSelectThreads / 64];
  __shared__ int s_any[kDdpSelectThreads / 64];
  __shared__ unsigned char s_keep[kDdpCopyHereMaxB];
  __shared__ float s_tile[kDdpNormTile];
  const int tid = threadIdx.x;
#ifdef DMPC_SELECT_TIMING   // scripts/microbench/select_phases.hip: s_memtime stamps of thread 0 into a.keep[B..]
  unsigned long long tst[6];
  int nst = 0;
#define DMPC_SSTAMP() do { __syncthreads(); tst[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DMPC_SSTAMP() do { } while (0)
#endif
  DMPC_SSTAMP();
  if (a.state[kDdpDone] != 0) {  // stopped in an earlier iteration: nothing moves any more
    for (int b = rb + tid; b < re; b += kDdpSelectThreads) a.keep[b] = 0;
    return;
  }
  if (a.info_stage != nullptr && a.info != nullptr)
    for (int b = rb + tid; b < re; b += kDdpSelectThreads) {
      const int v = a.info_stage[b];
      if (v != 0) atomicOr(&a.info[b], v);
    }
  const int row = a.T * a.nu;
  float vmax = -1.f;
  bool nan_seen = false;
  int any = 0;
  auto finish_row = [&](int b, float acc) {   // full_du_norm of row b and the per-sample "best so far" test
    const float nrm = sqrtf(acc);
    a.last_norm[b] = nrm;
    nan_seen = nan_seen || !(nrm == nrm);
    vmax = fmaxf(vmax, nrm);
    const bool better = a.it == 0 || a.costs[b] <= a.best_costs[b] + a.best_cost_eps;      // box_ddp.py:200-209
    if (better) {
      a.best_costs[b] = a.costs[b];
      a.best_norm[b] = nrm;
    }
    a.keep[b] = better ? 1 : 0;
    if (a.copy_here) s_keep[b - rb] = better ? 1 : 0;
    any |= (a.it > 0 && better) ? 1 : 0;
  };
  const int rows_per_tile = row <= kDdpNormTile ? (kDdpNormTile / row < NT ? kDdpNormTile / row : NT) : 0;
  if (a.scrambled && rows_per_tile > 0) {
    // the [T,nu,B] array read as [B, T*nu]: row b = flat [b*row, (b+1)*row) of that order          mpc_step.py:261-263
    // A lane per row would read 4 bytes out of every cache line it touches (rows lie `row` floats apart), and one CU
    // can only look up so many lines per microsecond: at B = 1024 that was 25 of the kernel's 28 us.  The differences are
    // read in flat order instead - consecutive lanes, consecutive floats - into an LDS tile, then every lane sums its own
    // row from the tile in the sequential order of the plain loop (same bits).
    for (int r0 = rb; r0 < re; r0 += rows_per_tile) {
      const int nr = re - r0 < rows_per_tile ? re - r0 : rows_per_tile;
      const unsigned f0 = (unsigned)r0 * (unsigned)row, n = (unsigned)nr * (unsigned)row;
      for (unsigned i0 = tid; i0 < n; i0 += 8u * NT) {   // eight independent pairs of loads in flight
        float uo[8], uf[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned i = i0 + (unsigned)q * NT;
          size_t idx = f0 + i;                           // nu == 1: the two orders coincide
          if (a.nu != 1) {
            const unsigned f = f0 + i, bb = f % (unsigned)a.B, tm = f / (unsigned)a.B;
            idx = ((size_t)(tm / (unsigned)a.nu) * a.B + bb) * a.nu + tm % (unsigned)a.nu;
          }
          const bool in = i < n;
          uo[q] = in ? a.u_old[idx] : 0.f;
          uf[q] = in ? a.u_first[idx] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned i = i0 + (unsigned)q * NT;
          if (i < n) s_tile[i] = uo[q] - uf[q];
        }
      }
      __syncthreads();
      if (tid < nr) {
        float acc = 0.f;
        for (int e = 0; e < row; ++e) {
          const float d = s_tile[tid * row + e];
          acc = fmaf(d, d, acc);
        }
        finish_row(r0 + tid, acc);
      }
      __syncthreads();
    }
  } else {
    for (int b = rb + tid; b < re; b += kDdpSelectThreads) {
      float acc = 0.f;
      if (a.scrambled) {   // rows longer than the tile: one lane per row
        const unsigned flat = (unsigned)b * (unsigned)row;
        unsigned bb = flat % (unsigned)a.B, tm = flat / (unsigned)a.B;
        unsigned m = tm % (unsigned)a.nu, t = tm / (unsigned)a.nu;
        for (int e = 0; e < row; ++e) {
          const size_t idx = ((size_t)t * a.B + bb) * a.nu + m;
          if (++bb == (unsigned)a.B) {
            bb = 0;
            if (++m == (unsigned)a.nu) {
              m = 0;
              ++t;
            }
          }
          const float d = a.u_old[idx] - a.u_first[idx];
          acc = fmaf(d, d, acc);
        }
      } else {
        for (int t = 0; t < a.T; ++t)
          for (int m = 0; m < a.nu; ++m) {
            const size_t idx = ((size_t)t * a.B + b) * a.nu + m;
            const float d = a.u_old[idx] - a.u_first[idx];
            acc = fmaf(d, d, acc);
          }
      }
      finish_row(b, acc);
    }
  }
  DMPC_SSTAMP();
  // wavefront reduction first (a serial pass of one lane over 256 LDS entries costs more than the norms themselves)
  int flags = (nan_seen ? 2 : 0) | any;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    flags |= __shfl_xor(flags, off);
  }
  if ((tid & 63) == 0) {
    s_max[tid >> 6] = vmax;
    s_any[tid >> 6] = flags;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = 0.f;   // norms are >= 0 (a NaN norm is carried by the flag)
    int fl = 0;
    for (int i = 0; i < kDdpSelectThreads / 64; ++i) {
      mx = fmaxf(mx, s_max[i]);
      fl |= s_any[i];
    }
    bool decide = true;
    if (n_parts > 1) {
      atomicMax(&psync[1], __float_as_uint(mx));   // non-negative floats order like their bit patterns
      atomicOr(&psync[2], (unsigned)fl);
      __threadfence();
      decide = atomicAdd(&psync[0], 1u) == (unsigned)(n_parts - 1);
      if (decide) {
        __threadfence();
        mx = __uint_as_float(atomicExch(&psync[1], 0u));
        fl = (int)atomicExch(&psync[2], 0u);
        atomicExch(&psync[0], 0u);
      }
    }
    if (decide) {
      const bool bad = (fl & 2) != 0;
      const int an = fl & 1;
      int n_not = a.state[kDdpNotImproved] + 1;
      if (an) n_not = 0;
      a.state[kDdpNotImproved] = n_not;
      a.state[kDdpIter] = a.it + 1;
      if (!bad && mx < a.eps) {                                                             // box_ddp.py:223-230
        a.state[kDdpStatus] = 1;
        a.state[kDdpDone] = 1;
      } else if (n_not > a.not_improved_lim) {
        a.state[kDdpStatus] = 2;
        a.state[kDdpDone] = 1;
      } else if (a.it == a.max_iter - 1) {
        a.state[kDdpStatus] = 3;
      }
    }
  }
  DMPC_SSTAMP();
  if (a.copy_here) {
    const float *__restrict__ xn = a.x_new, *__restrict__ un = a.u_new;
    float *__restrict__ bx = a.best_x, *__restrict__ bu = a.best_u;
    bool quads = false;
    if constexpr (NXC > 0)
      quads = a.B % 4 == 0 && rb % 4 == 0 && ((reinterpret_cast<uintptr_t>(xn) | reinterpret_cast<uintptr_t>(un) |
                               reinterpret_cast<uintptr_t>(bx) | reinterpret_cast<uintptr_t>(bu)) & 15u) == 0;
    if (quads) {
      // One workgroup moves the whole batch's best trajectories, and what bounds it is the number of memory
      // instructions one CU can issue (a dword access per lane costs the address pipeline what a 16-byte one does):
      // the rows of four consecutive trajectories at one timestep are NXC + NUC contiguous 16-byte chunks, loaded as
      // such, kQuadRows timesteps in flight; stored as such when all four trajectories improved (the usual case), else
      // row by row.  A thread owns quads of trajectories; with fewer quads than threads the spare threads share time.
      if constexpr (NXC > 0) {
        constexpr int kQuadRows = NT <= 256 ? 8 : 4;
        const int NQ = a.B / 4, NQL = (re - rb) / 4, q0 = rb / 4;   // quads per timestep, of this part, its first
        const int G = NQL > 0 && NQL < NT ? NT / NQL : 1;  // thread groups per quad
        const int tg = NQL > 0 ? tid / NQL : G;            // NQL >= NT: 0
        const float4 *__restrict__ xn4 = reinterpret_cast<const float4 *>(xn), *__restrict__ un4 = reinterpret_cast<const float4 *>(un);
        float4 *__restrict__ bx4 = reinterpret_cast<float4 *>(bx), *__restrict__ bu4 = reinterpret_cast<float4 *>(bu);
        for (int ql = G > 1 ? tid % NQL : tid; ql < NQL && tg < G; ql += NT) {
          const int qd = q0 + ql;
          const int k0 = s_keep[4 * ql], k1 = s_keep[4 * ql + 1], k2 = s_keep[4 * ql + 2], k3 = s_keep[4 * ql + 3];
          if (!(k0 | k1 | k2 | k3)) continue;
          const bool all = k0 & k1 & k2 & k3;
          for (int t0 = tg; t0 < a.T; t0 += kQuadRows * G) {
            float4 vx[kQuadRows][NXC], vu[kQuadRows][NUC];
#pragma unroll
            for (int q = 0; q < kQuadRows; ++q) {
              const int t = t0 + q * G;
              if (t < a.T) {
                const size_t r4 = (size_t)t * NQ + qd;   // index of the quad's first row / 4
#pragma unroll
                for (int i = 0; i < NXC; ++i) vx[q][i] = xn4[r4 * NXC + i];
#pragma unroll
                for (int i = 0; i < NUC; ++i) vu[q][i] = un4[r4 * NUC + i];
              }
            }
#pragma unroll
            for (int q = 0; q < kQuadRows; ++q) {
              const int t = t0 + q * G;
              if (t < a.T) {
                const size_t r4 = (size_t)t * NQ + qd;
                if (all) {
#pragma unroll
                  for (int i = 0; i < NXC; ++i) bx4[r4 * NXC + i] = vx[q][i];
#pragma unroll
                  for (int i = 0; i < NUC; ++i) bu4[r4 * NUC + i] = vu[q][i];
                } else {
                  const float *fx = reinterpret_cast<const float *>(&vx[q][0]), *fu = reinterpret_cast<const float *>(&vu[q][0]);
                  const int kk[4] = {k0, k1, k2, k3};
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    if (kk[j]) {
#pragma unroll
                      for (int i = 0; i < NXC; ++i) bx[(r4 * 4 + j) * NXC + i] = fx[j * NXC + i];
#pragma unroll
                      for (int i = 0; i < NUC; ++i) bu[(r4 * 4 + j) * NUC + i] = fu[j * NUC + i];
                    }
                }
              }
            }
          }
        }
      }
    } else {
      const int nb = re - rb;   // one (t, b) row per thread and trip
      for (int i = tid; i < a.T * nb; i += kDdpSelectThreads) {
        const int bl = i % nb;
        if (s_keep[bl]) {
          const size_t r = (size_t)(i / nb) * a.B + rb + bl;
          for (int j = 0; j < a.nx; ++j) bx[r * a.nx + j] = xn[r * a.nx + j];
          for (int j = 0; j < a.nu; ++j) bu[r * a.nu + j] = un[r * a.nu + j];
        }
      }
    }
  }
#ifdef DMPC_SELECT_TIMING
  DMPC_SSTAMP();
  if (tid == 0)
    for (int i = 0; i + 1 < nst; ++i) a.keep[a.B + i] = (int32_t)(tst[i + 1] - tst[i]);
#endif
}

// NXC = NUC = 0: any nx, nu; (3, 1): the pendulum of configs 2 and 4 (the host picks - one instantiation per kernel, its
// LDS tile is static)
template <int NXC, int NUC>
__global__ __launch_bounds__(kDdpSelectThreads) void box_ddp_select_kernel(const DdpSelectArgs a) {
  box_ddp_select_body<kDdpSelectThreads, NXC, NUC>(a);
}

// one workgroup; n_u = T * B * nu.  with_select: the bookkeeping of the last iteration first (its launch of its own in the
// plain chain; in the fused chain every other iteration's bookkeeping rides in the next backward sweep's launch)
__global__ __launch_bounds__(1024) void box_ddp_summary_kernel(size_t n_u, int B, const float *__restrict__ u_init,
                                                               const float *__restrict__ lower, const float *__restrict__ upper,
                                                               const int32_t *__restrict__ info, int nonfinite_bit,
                                                               const float *best_norm, float eps,
                                                               int32_t *state, const DdpSelectArgs sel, int with_select) {
  if (with_select) {   // fused chain: nx = 3, nu = 1
    box_ddp_select_body<1024, 3, 1>(sel);
    __syncthreads();   // best_norm / state were written by this workgroup
  }
  int nan_u = 0, bad_box = 0, n_bad = 0, above = 0;
  for (size_t e0 = threadIdx.x; e0 < n_u; e0 += (size_t)8 * blockDim.x) {   // eight independent triples in flight
    float v[8], lo[8], hi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const size_t e = e0 + (size_t)q * blockDim.x;
      const bool in = e < n_u;
      v[q] = in ? u_init[e] : 0.f;
      lo[q] = in ? lower[e] : 0.f;
      hi[q] = in ? upper[e] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      nan_u |= !(v[q] == v[q]);
      bad_box |= lo[q] > hi[q];
    }
  }
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (info != nullptr) n_bad += (info[b] & nonfinite_bit) != 0;
    above |= best_norm[b] > eps;
  }
  nan_u = __syncthreads_or(nan_u);
  bad_box = __syncthreads_or(bad_box);
  above = __syncthreads_or(above);
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  if (n_bad != 0) atomicAdd(&s_cnt, n_bad);
  __syncthreads();
  if (threadIdx.x == 0) {
    state[4] = nan_u != 0;
    state[5] = bad_box != 0;
    state[6] = s_cnt;
    state[7] = above != 0;
  }
}

__global__ __launch_bounds__(256) void box_ddp_keep_kernel(int T, int B, int nx, int nu, const int32_t *__restrict__ keep,
                                                           const float *__restrict__ x, const float *__restrict__ u,
                                                           float *__restrict__ best_x, float *__restrict__ best_u) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nxe = (size_t)T * B * nx, nue = (size_t)T * B * nu;
  for (size_t e = tid; e < nxe; e += stride)
    if (keep[(e / nx) % B]) best_x[e] = x[e];
  for (size_t e = tid; e < nue; e += stride)
    if (keep[(e / nu) % B]) best_u[e] = u[e];
}

}  // namespace dmpc
