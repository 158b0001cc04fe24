"""Headline solve with inputs that are NOT Infinity-Cache resident: launches alternate between three input sets
(3 x 161 MB > 256 MiB), so every launch streams its inputs from HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
B, T, nx, nu = 4096, 50, 8, 2
sets = [bench.make_inputs(B, T, nx, nu, s, torch.device("cuda"))[1] for s in range(3)]
x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
def run(k):
    d = sets[k % len(sets)]
    solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
for name, nset in (("one input set (cache resident)", 1), ("three input sets in rotation (from HBM)", 3)):
    for i in range(12): run(i % nset)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(150): run(i % nset)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 150 * 1e3
    print("%-42s %.1f us per solve, %.3e ts/s, %.0f GB/s algorithmic (%.2f of 8 TB/s)" % (name, us, B * T / us * 1e6, 832 * B * T / us / 1e3, 832 * B * T / us / 1e3 / 8000))
