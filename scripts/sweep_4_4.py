import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from chainer_differentiable_mpc_amd import _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
for (B,T,nx,nu) in ((4096,20,4,4),(4096,30,4,4),(4096,50,4,4),(4096,100,4,4),(8192,50,4,4)):
    p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
    x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
    fn=lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
    bench.settle(fn); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    ns = nx + nu
    frac = 4 * (ns * ns + ns + nx * ns + nx + ns) * B * T / us / 1e3 / 8000
    print("B=%d T=%d (%d,%d) path %d: %.1f us frac %.3f  %s" % (B,T,nx,nu,_lib.load().dmpc_lqr_solve_path(T,B,nx,nu),us,frac,_lib.last_kernel_name()[:80]), flush=True)
