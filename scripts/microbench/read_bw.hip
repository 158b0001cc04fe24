// read_bw.hip - what one MI355X box reads from HBM per second, by access pattern (1 GiB working set, beyond the 256 MiB
// Infinity Cache), next to its device-to-device copy rate:
//   vec     every wavefront of a full-occupancy grid walks the array with 16-byte loads (grid-stride), sums, writes nothing
//   dma N   the headline stream's pattern: N wavefronts per SIMD, each filling LDS slots of 3 KB with LDS-DMA
//           (global_load_lds_dwordx4, 3 instructions per slot, DEPTH slots in flight) from its own contiguous region
// Build: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip        Run on the GPU box: ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void vec_read(const float4 *p4, size_t n4, float *out) {
  const f4v *p = reinterpret_cast<const f4v *>(p4);
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f4v v = __builtin_nontemporal_load(p + i);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void dma_read(const char *p, size_t bytes_per_wave, int slots_per_wave) {
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char *src = p + gw * bytes_per_wave + lane * 16;
  const unsigned base = (unsigned)(size_t)(lds + wave * DEPTH * 3072);
  for (int s = 0; s < slots_per_wave; ++s) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + (s % DEPTH) * 3072);
    unsigned long long a = (unsigned long long)(src + (size_t)s * 3072);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory");
    if (NT) {
      asm volatile("global_load_lds_dwordx4 %0, off nt\n\tglobal_load_lds_dwordx4 %0, off offset:1024 nt\n\t"
                   "global_load_lds_dwordx4 %0, off offset:2048 nt" ::"v"(a) : "memory");
    } else {
      asm volatile("global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                   "global_load_lds_dwordx4 %0, off offset:2048" ::"v"(a) : "memory");
    }
    // keep DEPTH slots in flight: wait until at most (DEPTH-1) groups are outstanding
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    if (DEPTH == 6) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the headline stream's own gather: wave w, step t (descending) fetches the blocks of its four trajectories from four
// time-major arrays - C [T][B][100] floats (1,600 B per wave), c [T][B][10] (160 B), F [T][B][80] (1,280 B), f [T][B][8]
// (128 B) = 198 chunks of 16 bytes in 4 LDS-DMA instructions (58 padding lanes re-read chunk 0 of C), DEPTH steps in flight
typedef float f4w __attribute__((ext_vector_type(4)));
// WORK > 0: between the issue of a step's DMA group and the wait for the oldest one, WORK rounds of (8 dependent v_fma + 4
// independent 4x4x1 MFMAs) - the shape of the solve's own instruction mix; DMA = false: the work alone
// DENSE: the work is independent chains (4 MFMA accumulators, 4 FMA chains) issued back to back - the issue slots of the
// SIMD full, as in the solve's products - instead of one dependent chain
template <int DEPTH, bool NT, int WORK = 0, bool DMA = true, bool DENSE = false>
__global__ __launch_bounds__(256) void lqr_pattern(const char *C, const char *c, const char *F, const char *f, int T, int B, float *sink = nullptr) {
  float wx = 1.0f; f4w wm = {0.f, 0.f, 0.f, 0.f};
  float wy = 2.0f, wz = 3.0f, ww = 4.0f; f4w wn = wm, wo = wm, wp = wm;
  extern __shared__ char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const unsigned base = (unsigned)(size_t)(lds + wave * DEPTH * 4096);
  unsigned long long ptr[4], str[4];
  for (int q = 0; q < 4; ++q) {
    const int g = q * 64 + lane;
    const char *b0 = C; size_t per = 400; int g0 = 0;
    if (g < 100) { b0 = C; per = 400; g0 = 0; }
    else if (g < 110) { b0 = c; per = 40; g0 = 100; }
    else if (g < 190) { b0 = F; per = 320; g0 = 110; }
    else if (g < 198) { b0 = f; per = 32; g0 = 190; }
    else { g0 = g; }
    ptr[q] = (unsigned long long)(b0 + ((size_t)(T - 1) * B + gw * 4) * per + (size_t)(g - g0) * 16) - (unsigned long long)q * 1024;
    str[q] = (unsigned long long)((size_t)B * per);
  }
  for (int s = 0; s < T; ++s) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + (s % DEPTH) * 4096);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory");
    if (!DMA) {
    } else if (NT) {
      asm volatile("global_load_lds_dwordx4 %0, off nt\n\tglobal_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                   "global_load_lds_dwordx4 %2, off offset:2048 nt\n\tglobal_load_lds_dwordx4 %3, off offset:3072 nt"
                   ::"v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3]) : "memory");
    } else {
      asm volatile("global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                   "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %3, off offset:3072"
                   ::"v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3]) : "memory");
    }
    for (int q = 0; q < 4; ++q) ptr[q] -= str[q];
#pragma unroll
    for (int r = 0; r < WORK; ++r) {
      if (DENSE) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                     "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                     : "+v"(wx), "+v"(wy), "+v"(wz), "+v"(ww));
        wm = __builtin_amdgcn_mfma_f32_4x4x1f32(wx, wy, wm, 0, 0, 0);
        wn = __builtin_amdgcn_mfma_f32_4x4x1f32(wy, wz, wn, 0, 0, 0);
        wo = __builtin_amdgcn_mfma_f32_4x4x1f32(wz, ww, wo, 0, 0, 0);
        wp = __builtin_amdgcn_mfma_f32_4x4x1f32(ww, wx, wp, 0, 0, 0);
        wm = __builtin_amdgcn_mfma_f32_4x4x1f32(wx, wy, wm, 0, 0, 0);
        wn = __builtin_amdgcn_mfma_f32_4x4x1f32(wy, wz, wn, 0, 0, 0);
        wo = __builtin_amdgcn_mfma_f32_4x4x1f32(wz, ww, wo, 0, 0, 0);
        wp = __builtin_amdgcn_mfma_f32_4x4x1f32(ww, wx, wp, 0, 0, 0);
        continue;
      }
      asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\t"
                   "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(wx));
      wm = __builtin_amdgcn_mfma_f32_4x4x1f32(wx, wx, wm, 0, 0, 0);
      wm = __builtin_amdgcn_mfma_f32_4x4x1f32(wx, wx, wm, 0, 0, 0);
      wm = __builtin_amdgcn_mfma_f32_4x4x1f32(wx, wx, wm, 0, 0, 0);
      wm = __builtin_amdgcn_mfma_f32_4x4x1f32(wx, wx, wm, 0, 0, 0);
    }
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (WORK > 0 && wx + wy + wz + ww + wm[0] + wn[0] + wo[0] + wp[0] == 123.456f && sink) sink[0] = wx;
}

// shader clock of this box under a light load: a lone wavefront's dependent v_add chain (4 cycles each on the 16-lane SIMD)
__global__ void clock_probe(float *out, unsigned long long *ticks) {
  float x = out[1];
  const unsigned long long t0 = wall_clock64();   // 100 MHz
#pragma unroll 1
  for (int i = 0; i < 20000; ++i) {
    asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\t"
                 "v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(x));
  }
  const unsigned long long t1 = wall_clock64();
  out[0] = x;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <class F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char *a, *b; float *out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  {
    unsigned long long *ticks; CK(hipMalloc(&ticks, 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, out, ticks);
    unsigned long long h = 0; CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
    // 160,000 dependent adds, 4 cycles each (+ loop overhead ~2 %), h ticks of 10 ns
    printf("clock probe: %.3f ns per dependent v_add -> ~%.2f GHz if 4 cycles each\n", h * 10.0 / 160000, 4.0 / (h * 10.0 / 160000));
  }
  double ms = time_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
  printf("copy 1 GiB d2d          %.3f ms  %.0f GB/s (read + write)\n", ms, 2.0 * bytes / ms / 1e6);
  for (int wg_per_cu : {2, 4, 8}) {
    ms = time_ms([&] { hipLaunchKernelGGL(vec_read, dim3(256 * wg_per_cu), dim3(256), 0, 0, (const float4 *)a, bytes / 16, out); }, 10);
    printf("vec read, %d WG/CU       %.3f ms  %.0f GB/s\n", wg_per_cu, ms, bytes / ms / 1e6);
  }
#define DMA(DEPTH, NT, WPS)                                                                                          \
  {                                                                                                                  \
    const int waves = 1024 * WPS, slots = (int)(bytes / waves / 3072);                                               \
    const size_t per_wave = (size_t)slots * 3072;                                                                    \
    hipFuncSetAttribute((const void *)dma_read<DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEPTH * 3072); \
    ms = time_ms([&] { hipLaunchKernelGGL((dma_read<DEPTH, NT>), dim3(waves / 4), dim3(256), 4 * DEPTH * 3072, 0, a, per_wave, slots); }, 10); \
    printf("dma depth %d%s, %d wave/SIMD  %.3f ms  %.0f GB/s\n", DEPTH, NT ? " nt" : "   ", WPS, ms, (double)per_wave * waves / ms / 1e6); \
  }
  DMA(2, false, 1) DMA(3, false, 1) DMA(3, true, 1) DMA(4, true, 1) DMA(6, true, 1)
  DMA(3, true, 2) DMA(3, true, 4) DMA(6, true, 2) DMA(3, false, 4)
  {
    // eight problem sets of 162 MB in rotation (1.3 GB: nothing survives in the Infinity Cache)
    const int T = 50, B = 4096, NSET = 6;
    const size_t sC = (size_t)T * B * 400, sc = (size_t)T * B * 40, sF = (size_t)T * B * 320, sf = (size_t)T * B * 32;
    const size_t set = sC + sc + sF + sf;
    char *pool; CK(hipMalloc(&pool, set * NSET)); CK(hipMemset(pool, 3, set * NSET));
    int k = 0;
#define PAT(DEPTH, NT)                                                                                               \
    {                                                                                                                \
      hipFuncSetAttribute((const void *)lqr_pattern<DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEPTH * 4096); \
      ms = time_ms([&] { char *q = pool + (size_t)(k++ % NSET) * set;                                                \
                         hipLaunchKernelGGL((lqr_pattern<DEPTH, NT>), dim3(B / 16), dim3(256), 4 * DEPTH * 4096, 0, q, q + sC, q + sC + sc, q + sC + sc + sF, T, B); }, 24); \
      printf("lqr gather depth %d%s      %.4f ms  %.0f GB/s (162 MB per launch, %d sets in rotation)\n", DEPTH, NT ? " nt" : "   ", ms, (double)set / ms / 1e6, NSET); \
    }
    PAT(2, true) PAT(3, true) PAT(3, false) PAT(4, true)
#define PATW(DEPTH, WORK, DMA_)                                                                                       \
    {                                                                                                                \
      hipFuncSetAttribute((const void *)lqr_pattern<DEPTH, true, WORK, DMA_>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEPTH * 4096); \
      ms = time_ms([&] { char *q = pool + (size_t)(k++ % NSET) * set;                                                \
                         hipLaunchKernelGGL((lqr_pattern<DEPTH, true, WORK, DMA_>), dim3(B / 16), dim3(256), 4 * DEPTH * 4096, 0, q, q + sC, q + sC + sc, q + sC + sc + sF, T, B, out); }, 24); \
      printf("lqr gather depth %d nt + %2d rounds of work%s  %.4f ms\n", DEPTH, WORK, DMA_ ? "          " : " (no DMA) ", ms);   \
    }
#define PATD(DEPTH, WORK, DMA_)                                                                                       \
    {                                                                                                                \
      hipFuncSetAttribute((const void *)lqr_pattern<DEPTH, true, WORK, DMA_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DEPTH * 4096); \
      ms = time_ms([&] { char *q = pool + (size_t)(k++ % NSET) * set;                                                \
                         hipLaunchKernelGGL((lqr_pattern<DEPTH, true, WORK, DMA_, true>), dim3(B / 16), dim3(256), 4 * DEPTH * 4096, 0, q, q + sC, q + sC + sc, q + sC + sc + sF, T, B, out); }, 24); \
      printf("lqr gather depth %d nt + %2d rounds of DENSE work%s  %.4f ms\n", DEPTH, WORK, DMA_ ? "          " : " (no DMA) ", ms);   \
    }
    PATD(3, 6, false) PATD(3, 6, true) PATD(3, 9, false) PATD(3, 9, true) PATD(3, 12, false) PATD(3, 12, true)
    PATW(3, 6, false) PATW(3, 6, true) PATW(3, 10, false) PATW(3, 10, true) PATW(3, 14, false) PATW(3, 14, true) PATW(4, 10, true) PATW(2, 10, true)
  }
  CK(hipDeviceSynchronize());
  return 0;
}
