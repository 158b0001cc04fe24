// Phase timing of lqr_wave_mfma_backward<32,8> for lone wavefronts (s_memtime stamps inside the kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DDMPC_WAVE_TIMING -I include -I chainer_differentiable_mpc_amd/csrc \
//         scripts/microbench/wave_phases.hip -o /tmp/wave_phases && /tmp/wave_phases [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dmpc.h"
#include "lqr_wave_mfma.hpp"

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, T = 50, NX = 32, NU = 8, NS = 40;
  const size_t nC = (size_t)T * B * NS * NS, nc = (size_t)T * B * NS, nF = (size_t)(T - 1) * B * NX * NS, nf = (size_t)(T - 1) * B * NX;
  std::vector<float> C(nC), c(nc), F(nF), f(nf);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (size_t i = 0; i < nC; ++i) C[i] = 0.1f * rnd();
  for (size_t tb = 0; tb < (size_t)T * B; ++tb)
    for (int i = 0; i < NS; ++i) C[tb * NS * NS + i * NS + i] += 2.0f;
  for (auto &v : c) v = rnd();
  for (size_t i = 0; i < nF; ++i) F[i] = 0.15f * rnd();
  for (size_t tb = 0; tb < (size_t)(T - 1) * B; ++tb)
    for (int i = 0; i < NX; ++i) F[tb * NX * NS + i * NS + i] += 1.0f;
  for (auto &v : f) v = 0.1f * rnd();
  float *dC, *dc, *dF, *df, *dK, *dk;
  unsigned long long *dt;
  hipMalloc(&dC, nC * 4); hipMalloc(&dc, nc * 4); hipMalloc(&dF, nF * 4); hipMalloc(&df, nf * 4);
  hipMalloc(&dK, (size_t)T * B * NU * NX * 4); hipMalloc(&dk, (size_t)T * B * NU * 4);
  hipMalloc(&dt, 16 * sizeof(unsigned long long)); hipMemset(dt, 0, 16 * 8);
  hipMemcpy(dC, C.data(), nC * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), nc * 4, hipMemcpyHostToDevice);
  hipMemcpy(dF, F.data(), nF * 4, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), nf * 4, hipMemcpyHostToDevice);
  dmpc::LqrArgs a{T, B, dC, dc, dF, df, nullptr, nullptr, dK, dk, nullptr, nullptr, reinterpret_cast<float *>(dt), nullptr, nullptr};
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(dt, 0, 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((dmpc::lqr_wave_mfma_backward<32, 8, false, false>), dim3((B + 3) / 4), dim3(256), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    printf("B=%d kernel %.1f us; wave 0, memtime ticks per step (100 MHz): wait %.1f fetch %.1f G %.1f Q %.1f gains %.1f value %.1f total %.1f\n", B,
           ms * 1e3, h[0] / 50.0, h[1] / 50.0, h[2] / 50.0, h[3] / 50.0, h[4] / 50.0, h[5] / 50.0, h[6] / 50.0);
  }
  return 0;
}
