"""DiffLqr forward (fused solve) and backward (KKT gradient: second solve + co-state sweep) per shape, B = 4096, T = 50:
HIP-event time per call after a run-up, with the profiler-free kernel names of the gradient's launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import _lib
from chainer_differentiable_mpc_amd import differentiable_lqr as dl
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "8x2,8x4,12x3,6x3,9x4,11x4,13x2,16x4,16x8,13x3,20x6,24x8,32x8").split(",")]
B, T = int(os.environ.get("B", 4096)), 50
for nx, nu in shapes:
    Bq = B if nx < 32 else 2048
    p, d = bench.make_inputs(Bq, T, nx, nu, 0, torch.device("cuda"))
    x = torch.empty((T, Bq, nx), device="cuda"); u = torch.empty((T, Bq, nu), device="cuda")
    solve = lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
    ts = bench.event_time(solve, 20) * 1e6
    kn = _lib.last_kernel_name()[:70]
    gx, gu = torch.ones_like(x), torch.ones_like(u)
    grad = lambda: dl.kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu)
    tg = bench.event_time(grad, 20) * 1e6
    print("(%d,%d) B=%d: solve %.1f us [%s]; gradient %.1f us [last: %s]" % (nx, nu, Bq, ts, kn, tg, _lib.last_kernel_name()[:80]), flush=True)
    del p, d, x, u, gx, gu
    torch.cuda.empty_cache()
