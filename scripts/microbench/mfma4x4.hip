// Microbenchmark / semantics probe for the building blocks of the MFMA Riccati kernel (gfx950):
//   1. v_mfma_f32_4x4x1_16b_f32 operand layout, incl. CBSZ/ABID broadcast of the A operand
//   2. its issue cost for one wave per SIMD, alone and with LDS-read / VALU fillers, and the dependent latency
//   3. global_load_lds_dword (4-byte gather DMA): per-lane global address, LDS destination, inst_offset
// hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4 && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void sem_kernel(const float *a_in, const float *b_in, float *d_out) {
  const int l = threadIdx.x;
  float4v c = {0.f, 0.f, 0.f, 0.f};
  float4v d = __builtin_amdgcn_mfma_f32_4x4x1f32(a_in[l], b_in[l], c, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) d_out[l * 4 + i] = d[i];
}

template <int CBSZ, int ABID>
void check_semantics() {
  std::vector<float> a(64), b(64), d(256);
  for (int l = 0; l < 64; ++l) { a[l] = 1.0f + l; b[l] = 100.0f + 3 * l; }
  float *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((sem_kernel<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  // hypothesis: D_blk[i][j] in lane 4*blk+j, element i;  A_blk[i] in lane 4*blk'+i with blk' = (blk & ~(2^CBSZ-1)) | ABID
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int blk = l / 4, j = l % 4;
      const int ablk = CBSZ ? ((blk & ~((1 << CBSZ) - 1)) | ABID) : blk;
      const float want = a[4 * ablk + i] * b[4 * blk + j];
      if (d[l * 4 + i] != want) ++bad;
    }
  printf("semantics cbsz=%d abid=%d : %s (lane5: %g %g %g %g ; a-lanes 4..7 = %g.., b lane5 = %g)\n", CBSZ, ABID,
         bad ? "MISMATCH" : "as hypothesised", d[20], d[21], d[22], d[23], a[4], b[5]);
  hipFree(da); hipFree(db); hipFree(dd);
}

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

// MODE 0: 6 independent accumulators back to back; 1: + one ds_read_b128 per MFMA; 2: + one v_fmac per MFMA;
// 3: one dependent accumulator chain; 4: two fillers (ds_read + v_fmac) per MFMA; 5: 16x v_fmac only (reference)
template <int MODE>
__global__ __launch_bounds__(256) void time_kernel(float *out, int iters, unsigned long long *cyc) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
  float f0 = 1, f1 = 2;
  float4v r = {0, 0, 0, 0};
  const float *lp = lds + (threadIdx.x & 63) * 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 3) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      } else if (MODE == 5) {
        asm volatile(REP4("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %3\n\t") "v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %3\n\tv_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %3\n\t"
                     : "+v"(f0), "+v"(f1) : "v"(a), "v"(b));
      } else {
#define ONE(C, AB)                                                          \
  C = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, C, 2, AB, 0);                \
  if (MODE == 1 || MODE == 4) {                                             \
    float4v t;                                                              \
    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float *)lp)); \
    r = t;                                                                  \
  }                                                                         \
  if (MODE == 2 || MODE == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f0) : "v"(a), "v"(b));
        ONE(c0, 0) ONE(c1, 1) ONE(c2, 2) ONE(c3, 0) ONE(c4, 1) ONE(c5, 2)
      }
    }
  }
  if (MODE == 1 || MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = f0 + f1 + r[0];
  for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run_time(const char *name, int blocks, int threads, int per_iter) {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  const int iters = 4000;
  hipLaunchKernelGGL(time_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(time_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s blocks=%4d thr=%4d  wall ns per unit %.3f  (= %.2f cycles at 2.4 GHz)\n", name, blocks, threads,
         ms * 1e6 / ((double)iters * per_iter), ms * 1e6 / ((double)iters * per_iter) * 2.4);
  hipFree(out); hipFree(cyc);
}

// ---- gather DMA: lane l fetches global dword src[perm(l)] into LDS[base + l]; then with inst_offset
__global__ void dma_kernel(const float *src, const int *perm, float *out) {
  __shared__ float lds[512];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = -1.f;
  __syncthreads();
  const unsigned voff = (unsigned)perm[l] * 4u;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)lds;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2" ::"v"(voff), "s"(base), "s"(src) : "memory");
  // second instruction: same M0, inst_offset 256 -> does it displace the LDS side, the global side, or both?
  asm volatile("global_load_lds_dword %0, %1 offset:256" ::"v"(voff), "s"(src) : "memory");
  // third: partial EXEC (lanes < 20) into LDS base + 1024
  if (l < 20) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2" ::"v"(voff), "s"(base + 1024u), "s"(src) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = lds[i];
}

void check_dma() {
  std::vector<float> src(4096);
  std::vector<int> perm(64);
  for (int i = 0; i < 4096; ++i) src[i] = (float)i;
  for (int l = 0; l < 64; ++l) perm[l] = (l * 37) % 1000;
  float *ds, *dout; int *dp;
  hipMalloc(&ds, 4096 * 4); hipMalloc(&dp, 256); hipMalloc(&dout, 2048);
  hipMemcpy(ds, src.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dp, perm.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(dma_kernel, dim3(1), dim3(64), 0, 0, ds, dp, dout);
  std::vector<float> out(512);
  hipMemcpy(out.data(), dout, 2048, hipMemcpyDeviceToHost);
  int bad1 = 0, lds_shift = 0, glob_shift = 0, bad3 = 0;
  for (int l = 0; l < 64; ++l) {
    if (out[l] != src[perm[l]]) ++bad1;
    if (out[64 + l] == src[perm[l] + 64]) ++glob_shift;       // LDS +256 B and global +256 B
    if (out[64 + l] == src[perm[l]]) ++lds_shift;              // LDS +256 B only
    const float want = l < 20 ? src[perm[l]] : -1.f;
    if (out[256 + l] != want) ++bad3;
  }
  printf("gather DMA dword: plain %s ; offset:256 -> LDS displaced: %s, global displaced too: %d/64, global not displaced: %d/64 ; "
         "partial exec %s\n", bad1 ? "MISMATCH" : "ok", (glob_shift == 64 || lds_shift == 64) ? "yes" : "no", glob_shift, lds_shift,
         bad3 ? "MISMATCH" : "ok");
  printf("   lds[64..67] = %g %g %g %g   (src[perm[0..3]] = %g %g %g %g)\n", out[64], out[65], out[66], out[67],
         src[perm[0]], src[perm[1]], src[perm[2]], src[perm[3]]);
  hipFree(ds); hipFree(dp); hipFree(dout);
}

int main() {
  check_semantics<0, 0>();
  check_semantics<2, 0>();
  check_semantics<2, 1>();
  check_semantics<2, 3>();
  check_semantics<1, 1>();
  check_dma();
  for (int thr : {256, 512}) {
    run_time<0>("mfma4x4x1 6 indep accumulators (per MFMA)", 256, thr, 24);
    run_time<1>("  + ds_read_b128 each (per MFMA)", 256, thr, 24);
    run_time<2>("  + v_fmac each (per MFMA)", 256, thr, 24);
    run_time<4>("  + ds_read_b128 + v_fmac each (per MFMA)", 256, thr, 24);
    run_time<3>("mfma4x4x1 dependent chain (per MFMA)", 256, thr, 24);
    run_time<5>("v_fmac x12 (per instr)", 256, thr, 48);
  }
  return 0;
}
