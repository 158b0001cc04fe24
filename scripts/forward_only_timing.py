"""LqrRecursion.forward alone (dmpc_lqr_forward_sweep: gains given) per shape at B = 4096, T = 50: HIP-event time and kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "32x8,16x8,20x6").split(",")]
B, T = int(os.environ.get("B", 4096)), 50
lib = _lib.load()
for nx, nu in shapes:
    p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
    x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=True)
    x2, u2 = torch.empty_like(x), torch.empty_like(u)
    P = _lib.ptr

    def fwd():
        rc = lib.dmpc_lqr_forward_sweep(T, B, nx, nu, P(Ks), P(ks), P(d["F"]), P(d["f"]), P(d["x_init"]), None, P(x2), P(u2), None,
                                        _lib.stream_ptr(x.device))
        assert rc == 0, rc
    t = bench.event_time(fwd, 20)
    err = float((x2 - x).abs().max() / x.abs().max().clamp(min=1.0))
    print("(%d,%d) B=%d: forward sweep %.1f us [%s], max |dx| vs the fused solve %.2e" % (nx, nu, B, t * 1e6, _lib.last_kernel_name()[:70], err), flush=True)
    del p, d, x, u, Ks, ks, x2, u2
    torch.cuda.empty_cache()
