"""How much of the co-state kernel's time are its stores?  KKT gradient at the headline shape with / without dC, dF."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import differentiable_lqr as dl
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
B, T, nx, nu = 4096, 50, 8, 2
p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
gx, gu = torch.ones_like(x), torch.ones_like(u)
for name, kw in (("all outputs", {}), ("no dC", dict(need_dC=False)), ("no dF", dict(need_dF=False)),
                 ("no dC, no dF", dict(need_dC=False, need_dF=False))):
    f = lambda: dl.kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu, **kw)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("%-14s %.1f us per KKT gradient (second solve + co-state kernel)" % (name, e0.elapsed_time(e1) * 20))
