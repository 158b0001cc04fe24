import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from chainer_differentiable_mpc_amd import differentiable_lqr as dl
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
B, T, nx, nu = 4096, 50, 32, 8
p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
gx = torch.ones_like(x); gu = torch.ones_like(u)
for _ in range(2): dl.kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): dl.kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
by = 4 * (2 * 40 * 40 + 2 * 32 * 40 + 3 * 40 + 2 * 32) * B * T
print("KKT grad (32,8) B=%d: %.2f ms ; algorithmic %.1f GB -> %.0f GB/s" % (B, ms, by / 1e9, by / ms / 1e6))
