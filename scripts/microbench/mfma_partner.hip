// What can a wavefront do while the OTHER wavefront of its SIMD streams v_mfma_f32_16x16x4_f32 back to back?
// 512-thread workgroups: waves 0-3 (one per SIMD) run the matrix stream, waves 4-7 a test stream of one instruction class.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/mfma_partner.hip -o scripts/microbench/mfma_partner
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 VALU fma (independent), 1 SALU, 2 LDS reads, 3 readlane + fma, 4 dependent VALU chain
__device__ __forceinline__ float test_stream(int iters, float seed, float *lds) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  int s = iters;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
        a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
      }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 64; ++k) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 += lds[threadIdx.x + 64 * 0 + k]; a1 += lds[threadIdx.x + 64 * 1 + k]; a2 += lds[threadIdx.x + 64 * 2 + k]; a3 += lds[threadIdx.x + 64 * 3 + k];
        a4 += lds[threadIdx.x + 64 * 4 + k]; a5 += lds[threadIdx.x + 64 * 5 + k]; a6 += lds[threadIdx.x + 64 * 6 + k]; a7 += lds[threadIdx.x + 64 * 7 + k];
      }
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const float l = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a0), k));
        a1 = fmaf(-l, a2, a1);
        a0 += 1.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 64; ++k) a0 = fmaf(a0, 1.0001f, 0.5f);
    }
  }
  return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)s;
}

template <int MODE, int GAP>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *ticks, int mfma_iters, int test_iters, int prio) {
  __shared__ float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = 1.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (wave < 4) {
    f4v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f4v{0.f, 0.f, 0.f, 0.f};
    const float av = threadIdx.x * 0.001f, bv = 1.f - av;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i & 7], 0, 0, 0);
        if constexpr (GAP == 1) { asm volatile("s_nop 7\n\ts_nop 7"); }
        else if constexpr (GAP == 2) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"); }
        else if constexpr (GAP == 3) { asm volatile("s_sleep 0"); }
        else if constexpr (GAP == 4) { asm volatile("s_nop 0"); }
        else if constexpr (GAP == 5) { asm volatile("s_nop 3\n\ts_nop 3\n\ts_nop 3\n\ts_nop 3\n\ts_nop 3\n\ts_nop 3"); }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    if (prio > 0) __builtin_amdgcn_s_setprio(3);
    r = test_stream<MODE>(test_iters, threadIdx.x * 0.01f, lds);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int GAP = 0>
void run(const char *name, int per_iter, int prio = 0) {
  float *out; unsigned long long *ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8 * 8);
  const int MI = 2000, TI = 2000;
  unsigned long long h[8];
  double res[3][2];
  for (int cfg = 0; cfg < 3; ++cfg) {   // 0: both, 1: matrix stream alone, 2: test stream alone
    const int mi = cfg == 2 ? 0 : MI, ti = cfg == 1 ? 0 : TI;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, GAP>), dim3(256), dim3(512), 0, 0, out, ticks, mi, ti, prio);
    hipDeviceSynchronize();
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    res[cfg][0] = (double)h[0] / (MI * 16);          // cycles per MFMA (wave 0)
    res[cfg][1] = (double)h[4] / ((double)TI * per_iter);   // cycles per test instruction (wave 4)
  }
  printf("%-28s  MFMA: alone %.1f, beside the test stream %.1f cycles each | test instr: alone %.2f, beside the MFMA stream %.2f cycles each\n",
         name, res[1][0], res[0][0], res[2][1], res[0][1]);
  hipFree(out); hipFree(ticks);
}

int main() {
  run<0>("v_fma_f32 x8 independent", 64);
  run<4>("v_fma_f32 dependent chain", 64);
  run<1>("s_add_u32", 64);
  run<2>("ds_read_b32 (+ v_add)", 64);
  run<3>("v_readlane + v_fma + v_add", 96);
  printf("test stream at s_setprio 3:\n");
  run<0>("v_fma_f32 x8 independent", 64, 1);
  run<1>("s_add_u32", 64, 1);
  run<2>("ds_read_b32 (+ v_add)", 64, 1);
  run<3>("v_readlane + v_fma + v_add", 96, 1);
#define GAPRUN(G, label)                                                       \
  printf("matrix stream with a gap after every MFMA (%s):\n", label);          \
  run<0, G>("v_fma_f32 x8 independent", 64);                                    \
  run<2, G>("ds_read_b32 (+ v_add)", 64);                                       \
  run<3, G>("v_readlane + v_fma + v_add", 96);
  GAPRUN(4, "s_nop 0") GAPRUN(1, "2 x s_nop 7") GAPRUN(2, "3 x s_nop 7") GAPRUN(5, "6 x s_nop 3") GAPRUN(3, "s_sleep 0")
  return 0;
}
