// Host-side sanitizer run of the C-ABI's argument checking and dispatch logic (SURVEY.md section 5's hook).
// Built by scripts/asan/run_asan_host.sh with -fsanitize=address,undefined on the HOST side of every *_api.hip and run on the
// CPU (no GPU needed, none used): workspace queries, dispatch tables, every entry point with NULL / inconsistent arguments
// (must return DMPC_E_BADARG or DMPC_E_WORKSPACE without touching memory), and the full host path of the solves with fake
// non-null device pointers for a sweep of shapes - the launch itself fails with "no device", the code before it has run.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "dmpc.h"

static int checks = 0, bad = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    ++checks;                                                          \
    if (!(cond)) { ++bad; std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

int main() {
  float *p = reinterpret_cast<float *>(0x1000);       // a 16-byte aligned "device pointer" the host never dereferences
  double *pd = reinterpret_cast<double *>(0x1000);
  int32_t *pi = reinterpret_cast<int32_t *>(0x2000);
  uint8_t *pm = reinterpret_cast<uint8_t *>(0x3000);
  void *ws = reinterpret_cast<void *>(0x4000);
  EXPECT(dmpc_version() == DMPC_VERSION);
  EXPECT(dmpc_mpc_step_status(0, pi, pi, p, pi, nullptr) == DMPC_E_BADARG);      // (argument checks return before any launch)
  EXPECT(dmpc_mpc_step_status(16, pi, pi, p, nullptr, nullptr) == DMPC_E_BADARG);
  const int shapes[][2] = {{1, 1}, {3, 1}, {8, 2}, {4, 4}, {8, 4}, {12, 3}, {6, 3}, {5, 5}, {3, 8}, {16, 4}, {20, 6}, {32, 8}, {16, 8},
                           {40, 4}, {10, 12}, {64, 16}, {100, 1}, {70, 3}, {300, 40}};
  for (auto &s : shapes) {
    const int nx = s[0], nu = s[1];
    for (int T : {1, 2, 20, 50, 51, 52, 75, 100, 400})
      for (int B : {1, 3, 4, 17, 4096}) {
        const int fam = dmpc_lqr_kernel_family(nx, nu);
        EXPECT(fam >= 1 && fam <= 5);
        const size_t wsb = dmpc_lqr_workspace_bytes(T, B, nx, nu);
        EXPECT(wsb >= (size_t)T * B * nu * (nx + 1) * 4);
        (void)dmpc_lqr_solve_path(T, B, nx, nu);
        (void)dmpc_lqr_saving_available(T, B, nx, nu);
        EXPECT(dmpc_lqr_kkt_workspace_bytes(T, B, nx, nu) >= wsb);
        EXPECT(dmpc_mpc_step_workspace_bytes(T, B, nx, nu) >= wsb);
        EXPECT(dmpc_lqr_f64_workspace_bytes(T, B, nx, nu) > 0);
        (void)dmpc_box_ddp_workspace_bytes(T, B, nx, nu);
        (void)dmpc_mpc_backward_rec_workspace_bytes(T, B, nx, nu, 20, 0);
        (void)dmpc_mpc_backward_rec_workspace_bytes(T, B, nx, nu, 20, 1);
        // the whole host path: dispatch, layout arithmetic, launch attempt (fails: no device) - must not crash or overrun
        (void)dmpc_lqr_solve(T, B, nx, nu, p, p, T > 1 ? p : nullptr, p, p, nullptr, nullptr, nullptr, p, p, ws, wsb, pi, nullptr);
        (void)dmpc_lqr_solve(T, B, nx, nu, p, p, T > 1 ? p : nullptr, nullptr, p, pm, p, p, p, p, ws, wsb, pi, nullptr);
        (void)dmpc_lqr_backward_sweep_ws(T, B, nx, nu, p, p, p, p, nullptr, p, p, ws, wsb, pi, nullptr);
        (void)dmpc_lqr_forward_sweep(T, B, nx, nu, p, p, p, p, p, nullptr, p, p, pi, nullptr);
        if (T > 1) {
          (void)dmpc_lqr_kkt_grad(T, B, nx, nu, p, p, p, p, p, p, p, 0, p, p, p, p, p, ws, dmpc_lqr_kkt_workspace_bytes(T, B, nx, nu), pi, nullptr);
          (void)dmpc_mpc_step_forward(T, B, nx, nu, p, p, p, p, p, p, p, p, p, p, p, p, 1, 0.2f, 5, 20, 0, p, p, p, p, p, p, p, p, p, pi, pi,
                                      ws, dmpc_mpc_step_workspace_bytes(T, B, nx, nu), pi, nullptr);
          (void)dmpc_lqr_solve_f64(T, B, nx, nu, pd, pd, pd, pd, pd, nullptr, nullptr, nullptr, pd, pd, ws,
                                   dmpc_lqr_f64_workspace_bytes(T, B, nx, nu), pi, nullptr);
        }
        // inconsistent arguments are refused before anything else happens
        EXPECT(dmpc_lqr_solve(T, B, nx, nu, nullptr, p, p, p, p, nullptr, nullptr, nullptr, p, p, ws, wsb, pi, nullptr) == DMPC_E_BADARG);
        EXPECT(dmpc_lqr_solve(T, B, nx, nu, p, p, p, p, p, nullptr, p, nullptr, p, p, ws, wsb, pi, nullptr) == DMPC_E_BADARG);
        EXPECT(dmpc_lqr_solve(T, B, nx, nu, p + 1, p, p, p, p, nullptr, nullptr, nullptr, p, p, ws, wsb, pi, nullptr) == DMPC_E_BADARG);
        if (wsb > 0) EXPECT(dmpc_lqr_solve(T, B, nx, nu, p, p, p, p, p, nullptr, nullptr, nullptr, p, p, ws, wsb - 1, pi, nullptr) == DMPC_E_WORKSPACE);
      }
  }
  EXPECT(dmpc_lqr_kernel_family(0, 3) == DMPC_E_UNSUPPORTED);
  EXPECT(dmpc_lqr_workspace_bytes(0, 1, 1, 1) == 0);
  EXPECT(dmpc_lqr_solve(0, 1, 1, 1, p, p, p, p, p, nullptr, nullptr, nullptr, p, p, nullptr, 0, nullptr, nullptr) == DMPC_E_BADARG);
  EXPECT(dmpc_pnqp(0, 2, p, p, p, p, nullptr, 20, 0, p, p, pi, p, pi, nullptr, 0, pi, nullptr) == DMPC_E_BADARG);
  EXPECT(dmpc_pnqp(4, 2, p, p, p, p, nullptr, 20, 1, p, p, pi, p, pi, nullptr, 0, pi, nullptr) == DMPC_E_WORKSPACE);
  // batch-coupled workspaces (round 5: the fixed-grid form's rows behind the decision slots)
  EXPECT(dmpc_pnqp_workspace_bytes(4, 2, 20, 0) == 0);
  EXPECT(dmpc_pnqp_workspace_bytes(0, 2, 20, 1) == 0);
  for (int n : {1, 2, 8, 9, 40})
    for (int B : {1, 256, 16384})
      EXPECT(dmpc_pnqp_workspace_bytes(B, n, 20, 1) >= dmpc_coupled_workspace_bytes(1, 20) + (size_t)B * 12 * n * sizeof(float));
  for (int nx : {3, 8, 60})
    for (int nu : {1, 2, 9})
      EXPECT(dmpc_mpc_backward_rec_workspace_bytes(10, 64, nx, nu, 20, 1) >
             dmpc_coupled_workspace_bytes(10, 20) + (size_t)64 * (nx + nu) * (nx + nu + 1) * sizeof(float));
  EXPECT(dmpc_mpc_backward_rec(10, 64, 8, 2, p, p, p, p, p, p, p, 20, 1, p, p, pi, p, dmpc_coupled_workspace_bytes(10, 20), pi, nullptr) ==
         DMPC_E_WORKSPACE);
  EXPECT(dmpc_batch_lu_factor(0, 2, p, p, pi, nullptr, nullptr) == DMPC_E_BADARG);
  char name[64];
  (void)dmpc_last_kernel_name(name, sizeof(name));
  (void)dmpc_last_kernel_name(name, 1);
  std::printf("asan_host_driver: %d checks, %d failed\n", checks, bad);
  return bad ? 1 : 0;
}
