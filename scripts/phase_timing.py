"""Time the backward-only, forward-only and fused kernels at the headline shape (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import synthetic, _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device

B, T, nx, nu = 4096, 50, 8, 2
if len(sys.argv) > 4:
    B, T, nx, nu = [int(v) for v in sys.argv[1:5]]
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
lib = _lib.load()
Ks = torch.empty((T, B, nu, nx), device="cuda"); ks = torch.empty((T, B, nu), device="cuda")
x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
st = _lib.stream_ptr()
P = _lib.ptr

def bwd():
    _lib.check(lib.dmpc_lqr_backward_sweep(T, B, nx, nu, P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), None, P(Ks), P(ks), None, st), "b")
def fwd():
    _lib.check(lib.dmpc_lqr_forward_sweep(T, B, nx, nu, P(Ks), P(ks), P(d["F"]), P(d["f"]), P(d["x_init"]), None, P(x), P(u), None, st), "f")
def solve():
    solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))

for name, fn in (("backward_sweep", bwd), ("forward_sweep", fwd), ("fused solve", solve)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-16s %.2f us" % (name, e0.elapsed_time(e1) * 10))
