"""Repeat the fused solve at the headline shape and compare every run with the first (race detector)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
B, T, nx, nu = 4096, 50, 8, 2
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
x0, u0, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
x0, u0 = x0.clone(), u0.clone()
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(n):
    x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
    if not (torch.equal(x, x0) and torch.equal(u, u0)):
        bad += 1
        if bad <= 3:
            idx = (x != x0).nonzero()
            print("mismatch run", i, "n elems", len(idx), "first", idx[0].tolist() if len(idx) else None)
print("runs", n, "mismatching runs", bad)
