// Microbenchmark: per-instruction issue cost (shader cycles, s_memtime) of the VALU/LDS instructions the
// solver kernels lean on, for 1, 2 and 4 wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X

#define BODY8(INS) REP8(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7))

#define I_FMAC_DPP(n) "v_fmac_f32_dpp %" #n ", %8, %9 row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n\t"
#define I_CND32(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n\t"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %8, %9, %10\n\t"
#define I_BFI(n) "v_bfi_b32 %" #n ", %11, %8, %9\n\t"
#define I_MOVDPP_BANK(n) "v_mov_b32_dpp %" #n ", %8 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x8\n\t"
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %8 row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD_ROR(n) "v_add_f32_dpp %" #n ", %8, %9 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %8\n\t"
#define I_MULS(n) "v_mul_f32 %" #n ", %12, %9\n\t"
#define I_FMA3(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n\t"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n\t"
#define I_SNOP(n) "s_nop 0\n\t"
#define I_PERM(n) "v_permlane32_swap_b32 %" #n ", %8\n\t"

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *cyc, float sval) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  float b = 1.0001f, c = 0.5f;
  unsigned long long sm = __builtin_amdgcn_read_exec() & 0x0f0f0f0f0f0f0f0full;
  int vm = (threadIdx.x & 4) ? -1 : 0;
  float ss = __builtin_amdgcn_readfirstlane(sval);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define RUN(INS)                                                                                              \
  asm volatile(BODY8(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
               : "v"(b), "v"(c), "s"(sm), "v"(vm), "s"(ss) : "vcc")
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) RUN(I_FMAC_DPP);
    if (MODE == 1) RUN(I_FMAC);
    if (MODE == 2) RUN(I_CND32);
    if (MODE == 3) RUN(I_CND64);
    if (MODE == 4) RUN(I_BFI);
    if (MODE == 5) RUN(I_MOVDPP_BANK);
    if (MODE == 6) RUN(I_MOVDPP);
    if (MODE == 7) RUN(I_ADD_ROR);
    if (MODE == 8) RUN(I_RCP);
    if (MODE == 9) RUN(I_MULS);
    if (MODE == 10) RUN(I_FMA3);
    if (MODE == 11) RUN(I_MOV);
    if (MODE == 12) RUN(I_SNOP);
    if (MODE == 13) RUN(I_PERM);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads) {
  const int blocks = 256, iters = 1000, per_iter = 64;
  float *out; unsigned long long *cyc;
  (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc, 1.5f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc, 1.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  const int waves_per_simd = threads / 256;
  printf("%-28s waves/SIMD=%d  cycles/instr/wave %.2f  -> SIMD cycles per instr %.2f   (wall %.3f ns/instr/wave)\n", name,
         waves_per_simd, avg / ((double)iters * per_iter), avg / ((double)iters * per_iter) / waves_per_simd,
         ms * 1e6 / ((double)iters * per_iter));
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int thr : {256, 512, 1024}) {
    run<0>("v_fmac_f32_dpp newbcast", thr);
    run<1>("v_fmac_f32", thr);
    run<2>("v_cndmask_b32 (vcc)", thr);
    run<3>("v_cndmask_b32_e64 (sgpr)", thr);
    run<4>("v_bfi_b32", thr);
    run<5>("v_mov_b32_dpp bank_mask", thr);
    run<6>("v_mov_b32_dpp newbcast", thr);
    run<7>("v_add_f32_dpp row_ror", thr);
    run<8>("v_rcp_f32", thr);
    run<9>("v_mul_f32 sgpr operand", thr);
    run<10>("v_fma_f32 (vop3)", thr);
    run<11>("v_mov_b32", thr);
    run<12>("s_nop 0", thr);
    run<13>("v_permlane32_swap_b32", thr);
  }
  return 0;
}
