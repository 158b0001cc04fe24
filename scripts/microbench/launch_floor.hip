// What a launch costs by itself on this device: K dependent (same-stream) launches of a kernel that does nothing, for the grid /
// workgroup / LDS shapes of the box-DDP chain's launches.    hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void nothing(int *p) {
  extern __shared__ float lds[];
  if (p != nullptr && threadIdx.x == 9999) p[0] = (int)lds[0];
}

static float chain(int grid, int block, size_t lds, int K, hipStream_t s) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(nothing, dim3(grid), dim3(block), lds, s, (int *)nullptr);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL(nothing, dim3(grid), dim3(block), lds, s, (int *)nullptr);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / K;
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  const int shapes[][3] = {{1, 64, 0}, {1, 1024, 0}, {32, 256, 0}, {32, 256, 40960}, {256, 256, 0}, {256, 256, 40960}, {1024, 256, 0}};
  for (auto &sh : shapes)
    printf("grid %4d x %4d threads, %5d B of LDS: %.2f us per launch (2,000 launches back to back on one stream)\n", sh[0], sh[1],
           sh[2], chain(sh[0], sh[1], (size_t)sh[2], 2000, s));
  return 0;
}
