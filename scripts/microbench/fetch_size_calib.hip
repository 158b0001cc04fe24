// fetch_size_calib.hip - what the FETCH_SIZE counter reports for a known number of bytes read, by load width per lane: 16 bytes
// (the solve's LDS-DMA / dwordx4), 8 bytes (the float64 row kernels' one double per lane) and 4 bytes.  MI355X_MICROARCH.md gives
// the x2 correction (64-byte requests counted as 32) for 16-byte-per-lane loads; the verdict of round 4 asked whether it holds for
// the float64 kernels' 8-byte loads.  1 GiB per kernel, read once, streaming.
//   hipcc --offload-arch=gfx950 -O3 fetch_size_calib.hip -o /tmp/fsc
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fsc
#include <hip/hip_runtime.h>
#include <stdio.h>

template <class V>
__global__ __launch_bounds__(256) void read_all(const V *p, size_t n, float *out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const V v = p[i];
    const float *f = reinterpret_cast<const float *>(&v);
    for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc += f[k];
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char *buf;
  float *out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_all<float4>, dim3(8192), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
    hipLaunchKernelGGL(read_all<double>, dim3(8192), dim3(256), 0, 0, (const double *)buf, bytes / 8, out);
    hipLaunchKernelGGL(read_all<float>, dim3(8192), dim3(256), 0, 0, (const float *)buf, bytes / 4, out);
  }
  hipDeviceSynchronize();
  printf("three kernels x 3, 1 GiB = 1048576 KiB read by each launch\n");
  return 0;
}
