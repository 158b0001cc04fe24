"""MPCstep.forward (re-centring, sweep with the box QP of every step, line search) per shape at B = 4096 (or B=...), T = 50,
bounds +-0.5: HIP-event time per call and the kernels it ran (the last two launches' names)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost, _lib
from chainer_differentiable_mpc_amd.util import get_traj
shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "8x2,8x4,9x4,11x4,13x2,12x4,16x4,16x8,13x3,10x6,5x5,9x6,20x6").split(",")]
B, T = int(os.environ.get("B", 4096)), 50
dev = torch.device("cuda")
for nx, nu in shapes:
    p, d = bench.make_inputs(B, T, nx, nu, 0, dev)
    torch.manual_seed(0)
    un = (0.5 * torch.randn((T, B, nu), device=dev)).clamp(-0.5, 0.5)
    xn = get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
    lo, hi = torch.full((T, B, nu), -0.5, device=dev), torch.full((T, B, nu), 0.5, device=dev)
    step = MPCstep(un, T, hi, lo, B, nx, nu, xn, QuadCost(d["C"], d["c"]), LinDx(d["F"], d["f"]), 0.2, 5, need_expand=True)
    import warnings
    def fwd():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            step.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    ts = []                 # single calls, the median reported (a mean over a handful is at the mercy of one hiccup)
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fwd(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("(%d,%d) B=%d: MPCstep.forward median %.1f us, min %.1f, max %.1f over 9 calls (incl. the wrapper's host work) [last: %s]"
          % (nx, nu, B, ts[4], ts[0], ts[-1], _lib.last_kernel_name()[:70]), flush=True)
    del p, d, un, xn, lo, hi, step
    torch.cuda.empty_cache()
