// Phase timing of lqr_tile16_kernel<32,8,false> (s_memtime stamps inside the kernel, summed over every wavefront).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DDMPC_T16_TIMING -I include -I chainer_differentiable_mpc_amd/csrc \
//         scripts/microbench/tile16_phases.hip -o scripts/microbench/tile16_phases && scripts/microbench/tile16_phases [B]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dmpc.h"
#include "lqr_tile16.hpp"

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8192, T = 50, NX = 32, NU = 8, NS = 40;
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
  hipMalloc(&dt, 16 * sizeof(unsigned long long));
  hipMemcpy(dC, C.data(), nC * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), nc * 4, hipMemcpyHostToDevice);
  hipMemcpy(dF, F.data(), nF * 4, hipMemcpyHostToDevice); hipMemcpy(df, f.data(), nf * 4, hipMemcpyHostToDevice);
  dmpc::LqrArgs a{T, B, dC, dc, dF, df, nullptr, nullptr, dK, dk, nullptr, nullptr, reinterpret_cast<float *>(dt), nullptr, nullptr};
  const size_t lds = argc > 2 ? (size_t)atoi(argv[2]) : dmpc::Tile16Layout<32, 8>::lds_bytes();   // a larger request = fewer workgroups per CU
  hipFuncSetAttribute(reinterpret_cast<const void *>(&dmpc::lqr_tile16_kernel<32, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dmpc::lqr_tile16_kernel<32, 8, false>, 256, lds);
  printf("dynamic LDS %zu B per workgroup, occupancy query: %d workgroups per CU\n", lds, occ);
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(dt, 0, 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((dmpc::lqr_tile16_kernel<32, 8, false>), dim3((B + 3) / 4), dim3(256), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)h[9] * T;
    printf("B=%d kernel %.1f us; cycles per wavefront and step: wait %.0f read+dma %.0f G %.0f Q %.0f rows->lds %.0f gains %.0f value %.0f | total %.0f ; clock %.2f GHz\n",
           B, ms * 1e3, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n, h[7] / n, (double)h[7] / h[8] * 0.1);
  }
  return 0;
}
