// Sustained rate of v_mfma_f32_4x4x1_16b_f32 with every SIMD busy (the regime of lqr_wave_mfma_backward): ns per MFMA per
// wavefront slot, i.e. 8 cycles / f - the clock the matrix pipe actually runs at under this load.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_clock.hip -o build_tmp/mfma_clock && build_tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float *out, int n) {
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
  }
  f4 s = a0 + a1 + a2 + a3;
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = s[0];
}
int main() {
  float *out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {           // wavefronts per SIMD
    for (int n : {20000, 200000, 2000000}) {     // 4 n MFMAs per wavefront: ~0.3 ms, 3 ms, 30 ms
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, out, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double per = best * 1e6 / (4.0 * n * wps);   // ns per MFMA issued on a SIMD
      printf("%d wave(s) per SIMD, %8d MFMAs per wave: %8.3f ms, %.3f ns per MFMA per SIMD -> %.2f GHz at 8 cycles each\n", wps, 4 * n, best, per, 8.0 / per);
    }
  }
  for (int n : {200000, 2000000}) {   // ONE workgroup of eight wavefronts (two per SIMD of one CU): the same code, the rest of the chip idle
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double per = best * 1e6 / (4.0 * n * 2);
    printf("one CU, 2 waves per SIMD, %8d MFMAs per wave: %8.3f ms, %.3f ns per MFMA per SIMD -> %.2f GHz at 8 cycles each\n", 4 * n, best, per, 8.0 / per);
  }
  return 0;
}
