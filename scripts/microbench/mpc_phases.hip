// Phase timing of mpc_backward_rec_kernel for lone wavefronts (s_memtime stamps inside the kernel, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DDMPC_MPC_TIMING -DDMPC_MPC_TIMING_VMCNT=26 -DNXV=3 -DNUV=1 \
//         -I include -I chainer_differentiable_mpc_amd/csrc scripts/microbench/mpc_phases.hip -o build_tmp/mpc_phases_31
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dmpc.h"
#include "mpc_kernels.hpp"
#ifdef USE_DMA
#include "mpc_dma_kernels.hpp"
#endif

int main(int argc, char **argv) {
  constexpr int NX = NXV, NU = NUV, NS = NX + NU;
  const int B = argc > 1 ? atoi(argv[1]) : 128, T = argc > 2 ? atoi(argv[2]) : 20;
  const size_t nC = (size_t)T * B * NS * NS, nc = (size_t)T * B * NS, nF = (size_t)(T - 1) * B * NX * NS, nf = (size_t)(T - 1) * B * NX;
  std::vector<float> C(nC), c(nc), F(nF), f(nf), u((size_t)T * B * NU), x((size_t)T * B * NX), lo((size_t)T * B * NU), hi((size_t)T * B * NU);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (size_t i = 0; i < nC; ++i) C[i] = 0.0f;
  for (size_t tb = 0; tb < (size_t)T * B; ++tb)
    for (int i = 0; i < NS; ++i) {
      for (int j = 0; j <= i; ++j) { const float v = 0.1f * rnd(); C[tb * NS * NS + i * NS + j] = v; C[tb * NS * NS + j * NS + i] = v; }
      C[tb * NS * NS + i * NS + i] += 1.0f;
    }
  for (auto &v : c) v = rnd();
  for (size_t i = 0; i < nF; ++i) F[i] = 0.3f * rnd();
  for (size_t tb = 0; tb < (size_t)(T - 1) * B; ++tb)
    for (int i = 0; i < NX; ++i) F[tb * NX * NS + i * NS + i] += 1.0f;
  for (auto &v : f) v = 0.1f * rnd();
  for (auto &v : u) v = 0.2f * rnd();
  for (auto &v : x) v = rnd();
  for (auto &v : lo) v = -0.3f;
  for (auto &v : hi) v = 0.3f;
  auto up = [](const std::vector<float> &h) { float *d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
  float *dC = up(C), *dc = up(c), *dF = up(F), *df = up(f), *du = up(u), *dx = up(x), *dlo = up(lo), *dhi = up(hi);
  float *dK, *dk; int32_t *nq; unsigned long long *dt;
  hipMalloc(&dK, (size_t)T * B * NU * NX * 4); hipMalloc(&dk, (size_t)T * B * NU * 4); hipMalloc(&nq, B * 4);
  hipMalloc(&dt, 16 * 8 + B * 4);
  dmpc::MpcBackArgs a{};
  a.T = T; a.B = B; a.C = dC; a.c = dc; a.F = dF; a.f = df; a.controls = du; a.lower = dlo; a.upper = dhi; a.n_qp_iter = 20;
  a.Ks = dK; a.ks = dk; a.n_qp_total = nq; a.info = reinterpret_cast<int32_t *>(dt); a.done = nullptr; a.sync = nullptr; a.states = dx;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(dt, 0, 16 * 8 + B * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
#ifdef USE_DMA
    hipLaunchKernelGGL((dmpc::mpc_backward_rec_dma_kernel<NX, NU, USE_DMA>), dim3((B + 15) / 16), dim3(256),
                       (dmpc::MpcBackDmaLayout<NX, NU, USE_DMA>::lds_bytes()), 0, a);
#else
    hipLaunchKernelGGL((dmpc::mpc_backward_rec_kernel<NX, NU, 16>), dim3((B + 15) / 16), dim3(256), 0, 0, a);
#endif
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    std::vector<int32_t> q(B); hipMemcpy(q.data(), nq, B * 4, hipMemcpyDeviceToHost);
    long tot = 0; for (int i = 0; i < B; ++i) tot += q[i];
    {
      std::vector<unsigned> hk((size_t)T * B * NU * NX), hs((size_t)T * B * NU);
      hipMemcpy(hk.data(), dK, hk.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), dk, hs.size() * 4, hipMemcpyDeviceToHost);
      unsigned long long ck = 0; for (size_t i = 0; i < hk.size(); ++i) ck = ck * 1000003ull + hk[i]; for (size_t i = 0; i < hs.size(); ++i) ck = ck * 1000003ull + hs[i];
      printf("checksum(Ks, ks) %016llx  ", ck);
    }
    double s = 0; for (int i = 0; i < 7; ++i) s += h[i];
    printf("(%d,%d) B=%d T=%d kernel %.1f us; QP passes per step %.2f; ns per step: issue-loads %.0f wait %.0f riccati %.0f bcast %.0f pnqp %.0f gains+store %.0f value %.0f | total %.0f\n",
           NX, NU, B, T, ms * 1e3, (double)tot / B / T, h[6] * 10.0 / T, h[0] * 10.0 / T, h[1] * 10.0 / T, h[2] * 10.0 / T, h[3] * 10.0 / T, h[4] * 10.0 / T, h[5] * 10.0 / T, s * 10.0 / T);
  }
  return 0;
}
