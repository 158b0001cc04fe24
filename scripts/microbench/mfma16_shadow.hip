// Can VALU work issue in the shadow of a long fp32 MFMA?  (gfx950; planning probe for the (32,8) sweep)
//   v_mfma_f32_16x16x4_f32 is 8 passes (32 cycles); v_mfma_f32_4x4x1_16b_f32 is 2 passes (8 cycles).  The (32,8)
//   backward sweep built from 4x4x1 MFMAs gets nothing from a second wavefront per SIMD: is that the
//   issue port, and does a longer MFMA leave room for VALU instructions of the same / another wavefront?
// Each kernel times ITER iterations of a block of 8 MFMAs (independent accumulators) with F independent v_fma per MFMA
// interleaved, one wavefront per SIMD (256 threads) or two (512 threads).
// hipcc --offload-arch=gfx950 -O3 mfma16_shadow.hip -o mfma16_shadow && ./mfma16_shadow
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

#define FMA1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));

template <int KIND, int F>  // KIND 0: 16x16x4, 1: 4x4x1, 2: no MFMA (VALU only)
__global__ void probe(float *out, int iters, unsigned long long *cyc) {
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  f4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f4{0, 0, 0, 0};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float k1 = 0.999f, k2 = 0.001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
      if (KIND == 1) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) FMA1(v[(i + f) & 7])
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int F>
void run(const char *name, int threads) {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<KIND, F>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
  }
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz: convert with the shader clock estimate printed by the 4x4x1 baseline if needed
  printf("%-34s %d wave(s)/SIMD: %7.2f memtime ticks per group of 8 MFMAs + %d FMAs\n", name, threads / 256, (double)h / iters, 8 * F);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<0, 0>("16x16x4 only", threads);
    run<0, 2>("16x16x4 + 2 fma each", threads);
    run<0, 4>("16x16x4 + 4 fma each", threads);
    run<0, 6>("16x16x4 + 6 fma each", threads);
    run<1, 0>("4x4x1 only", threads);
    run<1, 1>("4x4x1 + 1 fma each", threads);
    run<1, 2>("4x4x1 + 2 fma each", threads);
    run<2, 2>("valu only, 2 fma per slot", threads);
    run<2, 6>("valu only, 6 fma per slot", threads);
  }
  return 0;
}
