"""Which kernel serves a shape, and how fast: the fused solve, DiffLqr forward + backward (autograd) and MPCstep.forward at
B = 4096, T = 50 over a list of (nx, nu) - shapes with a specialisation, shapes that run inside a container, and shapes
on the runtime-dimension kernels.   SHAPES="8x2,6x3,16x4" python scripts/shape_coverage_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import DiffLqr, LinDx, MPCstep, QuadCost, _lib, util
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
dev = torch.device("cuda")
B, T = int(os.environ.get("B", "4096")), int(os.environ.get("T", "50"))
shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "8x2,6x2,6x3,4x3,13x2,9x4,12x3,5x5,16x4").split(",")]


def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%8s %5s %12s %16s %16s" % ("shape", "path", "solve us", "DiffLqr f+b us", "MPCstep.fwd us"))
for nx, nu in shapes:
    _, d = bench.make_inputs(B, T, nx, nu, 0, dev)
    x = torch.empty((T, B, nx), device=dev); u = torch.empty((T, B, nu), device=dev)
    t_solve = timed(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)))
    Cg, cg, Fg, fg, xg = (d[k].clone().requires_grad_(True) for k in ("C", "c", "F", "f", "x_init"))
    node = DiffLqr(T, B, nx, nu)
    gx, gu = torch.randn((T, B, nx), device=dev), torch.randn((T, B, nu), device=dev)

    def difflqr():
        xs, us = node.apply((xg, Cg, cg, Fg, fg))
        torch.autograd.grad((xs, us), (xg, Cg, cg, Fg, fg), (gx, gu))
    t_diff = timed(difflqr, 5)
    torch.manual_seed(0)
    un = (0.5 * torch.randn((T, B, nu), device=dev)).clamp(-0.5, 0.5)
    xn = util.get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
    lo, hi = torch.full((T, B, nu), -0.5, device=dev), torch.full((T, B, nu), 0.5, device=dev)

    def mpc():
        step = MPCstep(un, T, hi, lo, B, nx, nu, xn, QuadCost(d["C"], d["c"]), LinDx(d["F"], d["f"]), 0.2, 5, need_expand=True)
        step.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    t_mpc = timed(mpc, 5)
    print("%8s %5d %12.1f %16.1f %16.1f" % ("(%d,%d)" % (nx, nu), _lib.load().dmpc_lqr_solve_path(T, B, nx, nu), t_solve, t_diff, t_mpc), flush=True)
    del d, x, u, Cg, cg, Fg, fg, xg, un, xn, lo, hi
    torch.cuda.empty_cache()
