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

namespace dmpc {

enum { kDdpDone = 0, kDdpIter = 1, kDdpStatus = 2, kDdpNotImproved = 3 };

__global__ __launch_bounds__(64) void lin_rollout_kernel(int T, int B, int nx, int nu, const float *__restrict__ x_init,
                                                         const float *__restrict__ u, const float *__restrict__ F,
                                                         const float *__restrict__ f, float *__restrict__ x,
                                                         const int32_t *__restrict__ done) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
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
};

constexpr int kDdpCopyHereMaxB = 2048;
constexpr int kDdpSelectThreads = 1024;

// NT threads of ONE workgroup
template <int NT>
__device__ __forceinline__ void box_ddp_select_body(const DdpSelectArgs &a) {
  constexpr int kDdpSelectThreads = NT;
  __shared__ float s_max[kDdpSelectThreads / 64];
  __shared__ int s_any[kDdpSelectThreads / 64];
  __shared__ unsigned char s_keep[kDdpCopyHereMaxB];
  const int tid = threadIdx.x;
  if (a.state[kDdpDone] != 0) {  // stopped in an earlier iteration: nothing moves any more
    for (int b = tid; b < a.B; b += kDdpSelectThreads) a.keep[b] = 0;
    return;
  }
  const int row = a.T * a.nu;
  float vmax = -1.f;
  bool nan_seen = false;
  int any = 0;
  for (int b = tid; b < a.B; b += kDdpSelectThreads) {
    float acc = 0.f;
    if (a.scrambled) {  // the [T,nu,B] array read as [B, T*nu]: row b = flat [b*row, (b+1)*row)      mpc_step.py:261-263
      const unsigned flat = (unsigned)b * (unsigned)row;
      unsigned bb = flat % (unsigned)a.B, tm = flat / (unsigned)a.B;
      unsigned m = tm % (unsigned)a.nu, t = tm / (unsigned)a.nu;
      auto next_idx = [&]() {
        const size_t idx = ((size_t)t * a.B + bb) * a.nu + m;
        if (++bb == (unsigned)a.B) {
          bb = 0;
          if (++m == (unsigned)a.nu) {
            m = 0;
            ++t;
          }
        }
        return idx;
      };
      int e = 0;
      for (; e + 8 <= row; e += 8) {  // eight independent pairs of loads in flight (one lane per row: latency bound)
        size_t idx[8];
        float uo[8], uf[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) idx[q] = next_idx();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          uo[q] = a.u_old[idx[q]];
          uf[q] = a.u_first[idx[q]];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float d = uo[q] - uf[q];
          acc = fmaf(d, d, acc);
        }
      }
      for (; e < row; ++e) {
        const size_t idx = next_idx();
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
    if (a.copy_here) s_keep[b] = better ? 1 : 0;
    any |= (a.it > 0 && better) ? 1 : 0;
  }
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
    float mx = -1.f;
    int fl = 0;
    for (int i = 0; i < kDdpSelectThreads / 64; ++i) {
      mx = fmaxf(mx, s_max[i]);
      fl |= s_any[i];
    }
    const bool bad = (fl & 2) != 0;
    const int an = fl & 1;
    int n_not = a.state[kDdpNotImproved] + 1;
    if (an) n_not = 0;
    a.state[kDdpNotImproved] = n_not;
    a.state[kDdpIter] = a.it + 1;
    if (!bad && mx < a.eps) {                                                               // box_ddp.py:223-230
      a.state[kDdpStatus] = 1;
      a.state[kDdpDone] = 1;
    } else if (n_not > a.not_improved_lim) {
      a.state[kDdpStatus] = 2;
      a.state[kDdpDone] = 1;
    } else if (a.it == a.max_iter - 1) {
      a.state[kDdpStatus] = 3;
    }
  }
  if (a.copy_here) {
    const int n_rows = a.T * a.B;  // one (t, b) row per thread and trip
    const float *__restrict__ xn = a.x_new, *__restrict__ un = a.u_new;
    float *__restrict__ bx = a.best_x, *__restrict__ bu = a.best_u;
    int b = tid % a.B;
    for (int r = tid; r < n_rows; r += kDdpSelectThreads) {
      if (s_keep[b]) {
        for (int i = 0; i < a.nx; ++i) bx[r * a.nx + i] = xn[r * a.nx + i];
        for (int i = 0; i < a.nu; ++i) bu[r * a.nu + i] = un[r * a.nu + i];
      }
      b = (b + kDdpSelectThreads) % a.B;
    }
  }
}

__global__ __launch_bounds__(kDdpSelectThreads) void box_ddp_select_kernel(const DdpSelectArgs a) {
  box_ddp_select_body<kDdpSelectThreads>(a);
}

// one workgroup; n_u = T * B * nu
__global__ __launch_bounds__(1024) void box_ddp_summary_kernel(size_t n_u, int B, const float *__restrict__ u_init,
                                                               const float *__restrict__ lower, const float *__restrict__ upper,
                                                               const int32_t *__restrict__ info, int nonfinite_bit,
                                                               const float *__restrict__ best_norm, float eps,
                                                               int32_t *__restrict__ state) {
  int nan_u = 0, bad_box = 0, n_bad = 0, above = 0;
  for (size_t e = threadIdx.x; e < n_u; e += blockDim.x) {
    const float v = u_init[e];
    nan_u |= !(v == v);
    bad_box |= lower[e] > upper[e];
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
