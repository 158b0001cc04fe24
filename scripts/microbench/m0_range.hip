// Does the LDS-DMA target (M0) reach beyond 64 KB on gfx950?  One workgroup, 150 KB of dynamic LDS:
// DMA 1 KB of a pattern to LDS byte offsets 1000*16, 70000-ish and 140000-ish, read back with ds_read.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(const float4 *src, float *out, unsigned off) {
  extern __shared__ float lds[];
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)lds;
  for (int i = threadIdx.x; i < 150 * 256; i += 64) lds[i] = -1.f;
  __syncthreads();
  const unsigned dst = base + off;
  const unsigned long long p = (unsigned long long)(src + threadIdx.x);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(0)" ::"s"(dst), "v"(p) : "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = lds[off / 4 + threadIdx.x * 4 + i];
  // also report where else the data may have landed (aliasing at off & 0xffff)
  for (int i = 0; i < 4; ++i) out[256 + threadIdx.x * 4 + i] = lds[(off & 0xffff) / 4 + threadIdx.x * 4 + i];
}
int main() {
  float4 *src; float *out;
  hipMalloc(&src, 1024); hipMalloc(&out, 2048 * 4);
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)i;
  hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
  unsigned offs[3] = {16000, 70000 & ~15u, 140000 & ~15u};
  for (unsigned off : offs) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 150 * 1024, 0, src, out, off);
    float r[512]; hipMemcpy(r, out, 512 * 4, hipMemcpyDeviceToHost);
    int ok = 0, alias = 0; for (int i = 0; i < 256; ++i) { ok += r[i] == h[i]; alias += r[256 + i] == h[i]; }
    printf("LDS-DMA to offset %6u: %3d/256 words at the target, %3d/256 at (offset & 0xffff)  err=%s\n", off, ok, alias,
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
