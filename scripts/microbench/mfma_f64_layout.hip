// Which element of D does register r of lane l hold after v_mfma_f64_16x16x4_f64, and which elements of A / B does a lane feed?
//   hipcc --offload-arch=gfx950 -O2 scripts/microbench/mfma_f64_layout.hip -o scripts/microbench/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4v __attribute__((ext_vector_type(4)));
__global__ void k(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  d4v c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
  double ha[64], hb[64], hd[256], *da, *db, *dd;
  // hypothesis: lane l feeds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]
  double A[16][4], Bm[4][16];
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) A[i][kk] = 1.0 + i + 0.01 * kk;
  for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) Bm[kk][j] = 2.0 + 0.5 * j + 10.0 * kk;
  for (int l = 0; l < 64; ++l) { ha[l] = A[l % 16][l / 16]; hb[l] = Bm[l / 16][l % 16]; }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 2048);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost);
  double D[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { D[i][j] = 0; for (int kk = 0; kk < 4; ++kk) D[i][j] += A[i][kk] * Bm[kk][j]; }
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int g = l / 16, j = l % 16;
    if (hd[l * 4 + r] != D[4 * g + r][j]) ok1 = 0;
    if (hd[l * 4 + r] != D[4 * r + g][j]) ok2 = 0;
  }
  printf("inputs as hypothesised; register r of lane 16 g + j = D[4 g + r][j]: %s ; = D[4 r + g][j]: %s\n", ok1 ? "yes" : "no", ok2 ? "yes" : "no");
  if (!ok1 && !ok2) {   // find where D[0][0..3], D[1][0], D[4][0] went
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; j += 5) for (int q = 0; q < 256; ++q)
      if (hd[q] == D[i][j]) printf("D[%d][%d] in lane %d register %d\n", i, j, q / 4, q % 4);
  }
  return 0;
}
