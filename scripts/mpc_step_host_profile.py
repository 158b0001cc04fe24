"""Where the host time of MPCstep.forward goes at config-3 size ((8,2), B = 4096, T = 50): cProfile over 300 calls."""
import cProfile
import os
import pstats
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost  # noqa: E402
from chainer_differentiable_mpc_amd.util import get_traj  # noqa: E402

warnings.simplefilter("ignore")
B, T, nx, nu = 4096, 50, 8, 2
dev = torch.device("cuda")
p, d = bench.make_inputs(B, T, nx, nu, 0, dev)
torch.manual_seed(0)
un = (0.5 * torch.randn((T, B, nu), device=dev)).clamp(-0.5, 0.5)
xn = get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
lo, hi = torch.full((T, B, nu), -0.5, device=dev), torch.full((T, B, nu), 0.5, device=dev)
step = MPCstep(un, T, hi, lo, B, nx, nu, xn, QuadCost(d["C"], d["c"]), LinDx(d["F"], d["f"]), 0.2, 5, need_expand=True)
inp = (d["x_init"], d["C"], d["c"], d["F"], d["f"])
for _ in range(5):
    step.forward(inp)
torch.cuda.synchronize()
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.forward(inp)
    ts.append(time.perf_counter() - t0)
print("MPCstep.forward: %.1f us wall per call (median of 50)" % (1e6 * sorted(ts)[25]))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step.forward(inp)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
