// What does a lone wavefront per SIMD pay per instruction?  Dependent-FMA chains, dependent loads (pointer chase
// through HBM-sized and L2-sized arrays), store -> vmcnt(0) round trips, LDS round trips and transcendental chains,
// at 1 / 32 / 1024 wavefronts in flight, as one long kernel and as 100 short back-to-back kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/lone_wave.hip -o /tmp/lone_wave && /tmp/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void fma_chain(float *out, int n, float a, float b) {
  float x = threadIdx.x;
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, a, b);
  }
  if (x == 123.456f) out[0] = x;
}

__global__ __launch_bounds__(64) void fma_indep(float *out, int n, float a, float b) {  // 4 independent chains
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x0 = __builtin_fmaf(x0, a, b);
      x1 = __builtin_fmaf(x1, a, b);
      x2 = __builtin_fmaf(x2, a, b);
      x3 = __builtin_fmaf(x3, a, b);
    }
  }
  if (x0 + x1 + x2 + x3 == 123.456f) out[0] = x0;
}

__global__ __launch_bounds__(64) void rcp_chain(float *out, int n, float a) {
  float x = 1.5f + threadIdx.x;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x = __builtin_amdgcn_rcpf(x) + a;
  }
  if (x == 123.456f) out[0] = x;
}

__global__ __launch_bounds__(64) void chase(const int *__restrict__ next, int *out, int n, int stride_blocks) {
  int p = (blockIdx.x * stride_blocks + threadIdx.x) ;
  for (int i = 0; i < n; ++i) p = next[p];
  if (p == -1) out[0] = p;
}

__global__ __launch_bounds__(64) void store_wait(float *buf, int n) {
  float *p = buf + (size_t)blockIdx.x * 64 * 1024 + threadIdx.x;
  float v = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    p[(i & 1023) * 64] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    v += 1.f;
  }
}

__global__ __launch_bounds__(64) void lds_chain(int *out, int n) {
  __shared__ int s[64];
  s[threadIdx.x] = (threadIdx.x * 17 + 5) & 63;
  __syncthreads();
  int p = threadIdx.x;
  for (int i = 0; i < n; ++i) p = s[p];
  if (p == -1) out[0] = p;
}

__global__ __launch_bounds__(64) void dpp_chain(float *out, int n) {
  float x = threadIdx.x;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) x = x + __shfl_xor(x, 1);
  }
  if (x == 123.456f) out[0] = x;
}

template <class Fn>
static float time_us(Fn fn, int reps = 3) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  float *out;
  hipMalloc(&out, 1024);
  const size_t big = (size_t)1 << 28;  // 1 GiB of ints: beyond L2 + MALL
  int *nextb, *nexts;
  hipMalloc(&nextb, big * 4);
  const size_t small = (size_t)1 << 18;  // 1 MiB: L2 resident
  hipMalloc(&nexts, small * 4);
  {
    std::vector<int> h(big);
    // a stride walk with a large odd multiplier: every hop lands in another DRAM page
    for (size_t i = 0; i < big; ++i) h[i] = (int)((i * 40503ull * 64 + 64 * 7919) % big);
    hipMemcpy(nextb, h.data(), big * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < small; ++i) h[i] = (int)((i * 40503ull * 64 + 64 * 7919) % small);
    hipMemcpy(nexts, h.data(), small * 4, hipMemcpyHostToDevice);
  }
  float *sbuf;
  hipMalloc(&sbuf, (size_t)1024 * 64 * 1024 * 4);
  const int grids[3] = {1, 32, 1024};
  for (int gi = 0; gi < 3; ++gi) {
    const int G = grids[gi];
    const int n = 100000;
    float t;
    t = time_us([&] { hipLaunchKernelGGL(fma_chain, dim3(G), dim3(64), 0, 0, out, n, 1.0001f, 0.5f); });
    printf("G=%4d  dependent fma      : %7.1f us  %.2f ns each\n", G, t, t * 1e3 / n);
    t = time_us([&] { hipLaunchKernelGGL(fma_indep, dim3(G), dim3(64), 0, 0, out, n, 1.0001f, 0.5f); });
    printf("G=%4d  4 independent fma  : %7.1f us  %.2f ns each\n", G, t, t * 1e3 / n);
    t = time_us([&] { hipLaunchKernelGGL(rcp_chain, dim3(G), dim3(64), 0, 0, out, n, 0.25f); });
    printf("G=%4d  rcp+add chain      : %7.1f us  %.2f ns per pair\n", G, t, t * 1e3 / n);
    t = time_us([&] { hipLaunchKernelGGL(dpp_chain, dim3(G), dim3(64), 0, 0, out, n); });
    printf("G=%4d  shfl_xor+add chain : %7.1f us  %.2f ns per pair\n", G, t, t * 1e3 / n);
    t = time_us([&] { hipLaunchKernelGGL(lds_chain, dim3(G), dim3(64), 0, 0, (int *)out, n); });
    printf("G=%4d  LDS chase          : %7.1f us  %.2f ns each\n", G, t, t * 1e3 / n);
    const int nc = 2000;
    t = time_us([&] { hipLaunchKernelGGL(chase, dim3(G), dim3(64), 0, 0, nextb, (int *)out, nc, 4099 * 64); });
    printf("G=%4d  HBM chase          : %7.1f us  %.1f ns each\n", G, t, t * 1e3 / nc);
    t = time_us([&] { hipLaunchKernelGGL(chase, dim3(G), dim3(64), 0, 0, nexts, (int *)out, nc, 64); });
    printf("G=%4d  L2 chase           : %7.1f us  %.1f ns each\n", G, t, t * 1e3 / nc);
    t = time_us([&] { hipLaunchKernelGGL(store_wait, dim3(G), dim3(64), 0, 0, sbuf, nc); });
    printf("G=%4d  store + vmcnt(0)   : %7.1f us  %.1f ns each\n", G, t, t * 1e3 / nc);
    // 100 short kernels back to back (each ~10k dependent FMAs)
    t = time_us([&] {
      for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(fma_chain, dim3(G), dim3(64), 0, 0, out, 10000, 1.0001f, 0.5f);
    });
    printf("G=%4d  100 x 10k-fma kernels: %7.1f us  %.2f us per kernel (%.2f ns per fma if all compute)\n", G, t, t / 100, t * 1e3 / 1e6);
  }
  return 0;
}
