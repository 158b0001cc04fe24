"""Secondary timings on one GPU (diagnostic): KKT gradient at the headline shape, (32,8) shard, pendulum shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import synthetic, _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
import bench

def timeit(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, (B, T, nx, nu) in (("headline", (4096, 50, 8, 2)), ("pendulum", (1024, 20, 3, 1)), ("pendulum128", (128, 20, 3, 1)),
                             ("cfg5-shard", (8192, 50, 32, 8))):
    p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
    x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
    us = timeit(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)), n=20 if B * nx > 100000 else 100)
    by = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu) * B * T
    print("%-12s solve  B=%5d T=%d (%d,%d): %9.1f us  %.3e ts/s  alg %.1f MB -> %.0f GB/s (%.2f of 8 TB/s) path %d" % (
        name, B, T, nx, nu, us, B * T / us * 1e6, by / 1e6, by / us / 1e3, by / us / 1e3 / 8000, _lib.load().dmpc_lqr_solve_path(T, B, nx, nu)), flush=True)

# KKT gradient (DiffLqr forward + backward) at the headline shape
from chainer_differentiable_mpc_amd import differentiable_lqr as dl
B, T, nx, nu = 4096, 50, 8, 2
p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
try:
    f = dl.DiffLqr(T, B, nx, nu)
    ins = [d["x_init"], d["C"], d["c"], d["F"], d["f"]]
    def fwd_bwd():
        xs = [t.detach().requires_grad_(True) for t in ins]
        x, u = f.apply(tuple(xs))
        (x.sum() + u.sum()).backward()
    us = timeit(fwd_bwd, n=20)
    print("DiffLqr fwd+bwd (autograd, headline): %.1f us -> %.3e ts/s" % (us, B * T / us * 1e6))
except Exception as e:
    print("DiffLqr timing failed:", repr(e))
