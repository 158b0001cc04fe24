"""Exercise every row of SURVEY.md section 8 at its BASELINE.json sizes (for `rocprofv3 --kernel-trace --stats`):
    A  fused LQR solve            (8,2) B=4096 T=50 ; (3,1) B=1024 T=20 ; (32,8) B=8192 T=50 (one shard of config 5)
    B  KKT gradient (DiffLqr)      (8,2) B=4096 T=50 ; (3,1) B=1024 T=20 (config 4)
    C  PNQP                        n=2 and n=8, B=4096
    D  batched LU factor / solve   n=2 and n=8, B=4096
    E  MPC step forward/backward   (3,1) B=128 T=20 (config 2) ; (8,2) B=4096 T=50
    F  LQR_active                  (8,2) B=4096 T=50
Each op runs REPS times after a warm-up; durations come from the profiler, not from this script."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from chainer_differentiable_mpc_amd import LQR_active, LinDx, MPCstep, QuadCost, PNQP, synthetic, util
from chainer_differentiable_mpc_amd import differentiable_lqr as dl
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device

REPS = int(os.environ.get("REPS", "10"))
dev = torch.device("cuda")

def rep(fn, n=REPS):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    for _ in range(n): fn()
    torch.cuda.synchronize()

def lqr(B, T, nx, nu):
    p, d = bench.make_inputs(B, T, nx, nu, 0, dev)
    x = torch.empty((T, B, nx), device=dev); u = torch.empty((T, B, nu), device=dev)
    rep(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)))
    return d, x, u

print("A: LQR solves", flush=True)
d82, x82, u82 = lqr(4096, 50, 8, 2)
d31, x31, u31 = lqr(1024, 20, 3, 1)
lqr(8192, 50, 32, 8)
torch.cuda.empty_cache()

print("B: KKT gradient", flush=True)
for (d, x, u, T, nx, nu) in ((d82, x82, u82, 50, 8, 2), (d31, x31, u31, 20, 3, 1)):
    gx = torch.ones_like(x); gu = torch.ones_like(u)
    rep(lambda: dl.kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu))

print("C/D: PNQP, LU", flush=True)
rng = np.random.RandomState(0)
for n in (2, 8):
    L = rng.standard_normal((4096, n, n)); H = torch.as_tensor(L @ L.transpose(0, 2, 1) + n * np.eye(n), dtype=torch.float32, device=dev)
    q = torch.as_tensor(rng.standard_normal((4096, n)), dtype=torch.float32, device=dev)
    lo = -0.3 * torch.ones_like(q); hi = 0.3 * torch.ones_like(q)
    rep(lambda: PNQP(H, q, lo, hi))
    rep(lambda: util.batch_lu_solve(util.batch_lu_factor(H), q))

print("E: MPC step", flush=True)
for (B, T, nx, nu, bound) in ((128, 20, 3, 1, 2.0), (4096, 50, 8, 2, 0.5)):
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])
    u_nom = torch.zeros((T, B, nu), device=dev)
    x_nom = util.get_traj(T, u_nom, x0, LinDx(F, f))
    hi = bound * torch.ones((T, B, nu), device=dev); lo = -hi
    def fwd_bwd():
        step = MPCstep(u_nom, T, hi, lo, B, nx, nu, x_nom, QuadCost(C, c), LinDx(F, f), ls_decay=0.2, max_ls_iter=5, need_expand=True)
        x, u = step.forward((x0, C, c, F, f))
        step.backward((0, 1, 2, 3, 4), (torch.ones_like(x), torch.ones_like(u)))
    rep(fwd_bwd, n=max(2, REPS // 2))

print("F: LQR_active", flush=True)
mask = (torch.rand((50, 4096, 2), device=dev) < 0.3)
rep(lambda: LQR_active(d82["x_init"], d82["C"], d82["c"], d82["F"], d82["f"], 50, 8, 2, u_zero_Index=mask).solve_recursion())
print("done", flush=True)
