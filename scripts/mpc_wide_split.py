"""MPCstep at 17-32 states: where a step's time goes - backward_rec (sweep with the box QP) and forward_rec (line search) timed
separately by HIP events, with the kernels they ran.   python scripts/mpc_wide_split.py   (on the GPU box)"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost, _lib  # noqa: E402
from chainer_differentiable_mpc_amd.util import get_traj  # noqa: E402

warnings.simplefilter("ignore")
B, T = int(os.environ.get("B", 4096)), 50
dev = torch.device("cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


SH = [tuple(int(v) for v in sh.split('x')) for sh in os.environ.get('SHAPES', '16x8,20x6,24x8,32x8,32x4').split(',')]
for nx, nu in SH:
    p, d = bench.make_inputs(B, T, nx, nu, 0, dev)
    torch.manual_seed(0)
    un = (0.5 * torch.randn((T, B, nu), device=dev)).clamp(-0.5, 0.5)
    xn = get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
    lo, hi = torch.full((T, B, nu), -0.5, device=dev), torch.full((T, B, nu), 0.5, device=dev)
    step = MPCstep(un, T, hi, lo, B, nx, nu, xn, QuadCost(d["C"], d["c"]), LinDx(d["F"], d["f"]), 0.2, 5, need_expand=True)
    tau = torch.cat((xn, un), dim=2)
    c_hat = (torch.einsum("tbij,tbj->tbi", d["C"], tau) + d["c"]).contiguous()     # need_expand: mpc_step.py:305-317
    tb = timed(lambda: step.backward_rec(d["C"], c_hat, d["F"], None))
    kb = _lib.last_kernel_name().split("(")[0]
    Ks, ks, bo = step.backward_rec(d["C"], c_hat, d["F"], None)
    tf = timed(lambda: step.forward_rec(Ks, ks, step.true_cost, step.true_dynamics, 0.2, 5))
    kf = _lib.last_kernel_name().split("(")[0]
    tw = timed(lambda: step.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"])))
    print("(%d,%d) B=%d: backward_rec %.0f us [%s]; forward_rec %.0f us [%s]; forward() %.0f us; QP passes per timestep %.2f; "
          "line-search passes %.2f" % (nx, nu, B, tb, kb, tf, kf, tw, float(step.n_qp_iter.float().mean()) / T,
                                       float(step.n_ls_iter.float().mean())), flush=True)
    del p, d, un, xn, lo, hi, step, Ks, ks
    torch.cuda.empty_cache()
