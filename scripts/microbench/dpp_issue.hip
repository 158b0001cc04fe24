// Microbenchmark: issue cost of v_fmac_f32_dpp (row_newbcast) vs plain v_fmac_f32 for ONE wave per SIMD
// and for 2 / 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 dpp_issue.hip -o dpp_issue && ./dpp_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  float b = 1.0001f, c = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 independent accumulators, DPP fused FMA
      asm volatile(REP8(
          "v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (MODE == 1) {  // 8 independent accumulators, plain FMA
      asm volatile(REP8(
          "v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\t"
          "v_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9\n\tv_fmac_f32 %6, %8, %9\n\tv_fmac_f32 %7, %8, %9\n\t")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (MODE == 2) {  // one dependent chain, DPP
      asm volatile(REP64("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
                   : "+v"(a0) : "v"(b), "v"(c));
    } else if (MODE == 3) {  // one dependent chain, plain
      asm volatile(REP64("v_fmac_f32 %0, %1, %2\n\t") : "+v"(a0) : "v"(b), "v"(c));
    } else if (MODE == 4) {  // v_mov_dpp + s_nop + v_fmac (what hipcc emits)
      asm volatile(REP8(
          "v_mov_b32_dpp %10, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %0, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %1, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %2, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %3, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %4, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %5, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %6, %10, %9\n\t"
          "v_mov_b32_dpp %10, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %7, %10, %9\n\t")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(a0));
    } else if (MODE == 5) {  // v_cndmask throughput
      asm volatile(REP8(
          "v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %8, %9, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\t"
          "v_cndmask_b32 %3, %8, %9, vcc\n\tv_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %8, %9, vcc\n\t"
          "v_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %8, %9, vcc\n\t")
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    } else if (MODE == 6) {  // v_pk_fma_f32: 2 FMAs per lane per instruction
      asm volatile(REP8(
          "v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\t"
          "v_pk_fma_f32 %3, %4, %5, %3\n\tv_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\t"
          "v_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3\n\t")
          : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
          : "v"(*(double*)&b), "v"(*(double*)&c));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int threads, int per_iter) {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  // s_memtime ticks at 100 MHz on gfx9 (constant clock): report ns per instruction from wall time too
  printf("%-34s blocks=%4d thr=%4d  s_memtime ticks/instr %.4f   wall ns/instr(per wave) %.3f\n", name, blocks, threads,
         avg / ((double)iters * per_iter), ms * 1e6 / ((double)iters * per_iter));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int thr : {256, 512, 1024}) {
    run<0>("fmac_dpp 8 indep", 256, thr, 64);
    run<1>("fmac plain 8 indep", 256, thr, 64);
    run<2>("fmac_dpp dependent chain", 256, thr, 64);
    run<3>("fmac plain dependent chain", 256, thr, 64);
    run<4>("mov_dpp + fmac (2 instr)", 256, thr, 128);
    run<5>("cndmask 8 indep", 256, thr, 64);
    run<6>("pk_fma_f32 4 indep", 256, thr, 64);
  }
  return 0;
}
