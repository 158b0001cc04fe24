"""Headline-size solve three ways, back to back: plain, with gains out, saving (gains + Quu + Qxu out); then the re-solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import saved_solve_device, solve_device, solve_saving_device
dev = torch.device("cuda")
B, T, nx, nu = 4096, 50, 8, 2
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])

def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

print("plain   %.1f us" % timeit(lambda: solve_device(C, c, F, f, x0, None, T, nx, nu)))
print("gains   %.1f us" % timeit(lambda: solve_device(C, c, F, f, x0, None, T, nx, nu, want_gains=True)))
print("saving  %.1f us" % timeit(lambda: solve_saving_device(C, c, F, f, x0, T, nx, nu)))
x, u, Ks, ks, Quu, Qxu, Vv = solve_saving_device(C, c, F, f, x0, T, nx, nu)
print("resolve %.1f us" % timeit(lambda: saved_solve_device(c, F, Ks, Quu, Qxu, x0, T, nx, nu)))
