// Phase timing of box_ddp_select_body (s_memtime stamps between workgroup barriers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDMPC_SELECT_TIMING -I include -I chainer_differentiable_mpc_amd/csrc \
//         scripts/microbench/select_phases.hip -o build_tmp/select_phases && build_tmp/select_phases [B] [T] [NT]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dmpc.h"
#include "box_ddp_kernels.hpp"

template <int NT>
__global__ __launch_bounds__(NT) void sel_kernel(const dmpc::DdpSelectArgs a) { dmpc::box_ddp_select_body<NT, 3, 1>(a); }

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, T = argc > 2 ? atoi(argv[2]) : 20, NT = argc > 3 ? atoi(argv[3]) : 256;
  const int nx = 3, nu = 1;
  const size_t TB = (size_t)T * B;
  std::vector<float> uo(TB), uf(TB), xn(TB * 3), costs(B), bc(B);
  srand(2);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (auto &v : uo) v = rnd();
  for (auto &v : uf) v = rnd();
  for (auto &v : xn) v = rnd();
  for (int b = 0; b < B; ++b) { costs[b] = rnd(); bc[b] = costs[b] + (b % 3 == 0 ? -1.f : 1.f); }   // two thirds improve
  auto up = [](const std::vector<float> &h) { float *d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
  float *duo = up(uo), *duf = up(uf), *dxn = up(xn), *dun = up(uo), *dc = up(costs), *dbc = up(bc);
  float *bn, *ln, *bx, *bu; int32_t *keep, *state;
  hipMalloc(&bn, B * 4); hipMalloc(&ln, B * 4); hipMalloc(&bx, TB * 12); hipMalloc(&bu, TB * 4);
  hipMalloc(&keep, (B + 16) * 4); hipMalloc(&state, 32);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(state, 0, 32);
    hipMemcpy(dbc, bc.data(), B * 4, hipMemcpyHostToDevice);
    dmpc::DdpSelectArgs a{1, T, B, nx, nu, 10, 5, 1, 1e-3f, 1e-4f, duo, duf, dc, dbc, bn, ln, keep, state, 1, dxn, dun, bx, bu};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (NT == 256) hipLaunchKernelGGL(sel_kernel<256>, dim3(1), dim3(256), 0, 0, a);
    else hipLaunchKernelGGL(sel_kernel<1024>, dim3(1), dim3(1024), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int32_t h[8]; hipMemcpy(h, keep + B, sizeof(h), hipMemcpyDeviceToHost);
    printf("B=%d T=%d NT=%d kernel %.1f us; ticks: done-check+norm %d, reduce+state %d, (pre-copy sync) %d, copy %d\n", B, T, NT, ms * 1e3, h[0], h[1], h[2], h[3]);
  }
  return 0;
}
