// Which instruction classes of the SAME wavefront issue in the shadow of its own v_mfma_f32_16x16x4_f32 (32 cycles each)?
// One wavefront per SIMD; loop body = 8 x (MFMA + K fillers of one class); cycles per MFMA by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/mfma_same_wave.hip -o scripts/microbench/mfma_same_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *ticks, int iters, const float *src) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.f;
  __syncthreads();
  f4v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f4v{0.f, 0.f, 0.f, 0.f};
  const float av = threadIdx.x * 0.001f, bv = 1.f - av;
  float f0 = av, f1 = bv, f2 = 1.f, f3 = 2.f;
  int s = iters;
  const unsigned lds_dst = (unsigned)(size_t)(lds + 2048) + (threadIdx.x >> 6) * 1024 * 0;
  const unsigned voff = (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if constexpr (MODE == 0) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
        else if constexpr (MODE == 1) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(voff), "n"(256 * ((q) % 8))); f0 = v; }
        else if constexpr (MODE == 2) { f1 = fmaf(f1, 1.0001f, 0.5f); asm volatile("" : "+v"(f1)); }
        else if constexpr (MODE == 3) { int l; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(l) : "v"(f2)); s += l; }
        else if constexpr (MODE == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(src) : "memory");
        else if constexpr (MODE == 5) asm volatile("s_nop 0");
        else if constexpr (MODE == 6) asm volatile("ds_write_b32 %0, %1" ::"v"(voff), "v"(f3) : "memory");
      }
    }
    if constexpr (MODE == 1 || MODE == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = f0 + f1 + f2 + f3 + (float)s;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE, int K>
void run(const char *name) {
  float *out, *src; unsigned long long *ticks;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 64); hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  const int IT = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), 0, 0, out, ticks, IT, src);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-34s K=%d per MFMA: %.1f cycles per (MFMA + fillers)\n", name, K, (double)h / (IT * 8.0));
  hipFree(out); hipFree(ticks); hipFree(src);
}

int main() {
  run<5, 0>("MFMA only");
  run<0, 2>("s_add_u32"); run<0, 4>("s_add_u32"); run<0, 6>("s_add_u32");
  run<5, 2>("s_nop 0"); run<5, 6>("s_nop 0");
  run<1, 1>("ds_read_b32"); run<1, 2>("ds_read_b32"); run<1, 4>("ds_read_b32");
  run<6, 2>("ds_write_b32"); run<6, 4>("ds_write_b32");
  run<2, 2>("v_fma_f32"); run<2, 4>("v_fma_f32");
  run<3, 1>("v_readlane_b32"); run<3, 2>("v_readlane_b32"); run<3, 4>("v_readlane_b32");
  run<4, 1>("m0 + LDS-DMA 1 KB"); run<4, 2>("m0 + LDS-DMA 1 KB");
  return 0;
}
