"""(32,8) shard of config 5: backward sweep (matrix-core kernel), forward sweep and the fused solve, inputs drawn on
the device.  Usage: wave_mfma_timing.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T, nx, nu = 50, 32, 8
_, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
lib = _lib.load()
Ks = torch.empty((T, B, nu, nx), device="cuda"); ks = torch.empty((T, B, nu), device="cuda")
x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
st = _lib.stream_ptr(); P = _lib.ptr
def bwd():
    _lib.check(lib.dmpc_lqr_backward_sweep(T, B, nx, nu, P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), None, P(Ks), P(ks), None, st), "b")
def fwd():
    _lib.check(lib.dmpc_lqr_forward_sweep(T, B, nx, nu, P(Ks), P(ks), P(d["F"]), P(d["f"]), P(d["x_init"]), None, P(x), P(u), None, st), "f")
def solve():
    solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
for name, fn in (("backward_sweep", bwd), ("forward_sweep", fwd), ("fused solve", solve)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("B=%d %-16s %.1f us  (%.3e ts/s, %.2f of the 8 TB/s roof on 11,968 B per timestep-solve)" % (B, name, us, B * T / us * 1e6, 11968.0 * B * T / (us * 1e-6) / 8e12))
