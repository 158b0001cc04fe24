"""Timing and sampled oracle parity of the fused solve at B = 65536 (16 rounds of workgroups, 2.7 GB of inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import lqr as olqr
import bench
B, T, nx, nu = 65536, 50, 8, 2
p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
info = torch.zeros(B, dtype=torch.int32, device="cuda")
x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, info=info)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("B=65536: %.1f us per solve, %.3e ts/s, %.0f GB/s alg; info max %d" % (ms * 1e3, B * T / ms * 1e3, 832 * B * T / ms / 1e6, int(info.max())))
idx = np.r_[0:8, 30000:30008, B - 8:B]
sl = lambda a, ax: np.take(a.cpu().numpy().astype(np.float64), idx, axis=ax)
xr, ur = olqr.lqr_solve(sl(d["x_init"], 0), sl(d["C"], 1), sl(d["c"], 1), sl(d["F"], 1), sl(d["f"], 1), T, nx, nu)
ex = np.max(np.abs(sl(x, 1) - xr) / np.maximum(1, np.abs(xr))); eu = np.max(np.abs(sl(u, 1) - ur) / np.maximum(1, np.abs(ur)))
print("sampled parity: x %.2e u %.2e" % (ex, eu))
