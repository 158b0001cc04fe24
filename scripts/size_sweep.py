"""Fused-solve time over batch sizes and horizons at (8,2) (HIP events around 100 back-to-back launches of ONE input set after
the device has been run up to its steady clocks, bench.settle; sizes whose kernel is shorter than the host's launch rate
are launch bound and marked)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
# SHAPES="4x4,8x4,12x3" sweeps shapes at B=4096 (or BATCH=...), T=50 instead (algorithmic bytes 4(ns^2 + ns + nx ns + nx + ns) per timestep)
shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "").split(",") if sh]
cases = [(int(os.environ.get("BATCH", 4096)), 50, a, b) for a, b in shapes] or [(B, T, 8, 2) for B, T in (
    (1024, 50), (2048, 50), (4096, 50), (8192, 50), (16384, 50), (65536, 50), (4096, 20), (4096, 51), (4096, 52), (4096, 74),
    (4096, 100), (4096, 200))]
print("%7s %4s %7s %5s %10s %12s %8s" % ("B", "T", "shape", "path", "us/solve", "ts/s", "frac"))
for B, T, nx, nu in cases:
    p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
    x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
    ws = None
    bench.settle(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)))   # steady clocks
    torch.cuda.synchronize()
    n = 100 if B <= 16384 else 10
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    ns = nx + nu
    frac = 4 * (ns * ns + ns + nx * ns + nx + ns) * B * T / us / 1e3 / 8000
    print("%7d %4d %7s %5d %10.1f %12.3e %8.3f%s" % (B, T, "(%d,%d)" % (nx, nu), _lib.load().dmpc_lqr_solve_path(T, B, nx, nu), us, B * T / us * 1e6, frac,
                                              "  (launch bound)" if us < 16 else ""), flush=True)
    del p, d, x, u
    torch.cuda.empty_cache()
