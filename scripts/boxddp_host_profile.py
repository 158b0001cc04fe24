"""Where the host time of one BoxDDP solve goes (config 2: pendulum, B = 128, T = 20): wall per call without waiting for the
device, then cProfile over 300 calls."""
import cProfile
import os
import pstats
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_differentiable_mpc_amd import BoxDDP, PendulumDx, QuadCost  # noqa: E402
from chainer_differentiable_mpc_amd.pendulum import sample_xinit  # noqa: E402

warnings.simplefilter("ignore")
B, T = 128, 20
dx = PendulumDx()
q, pp = dx.get_true_obj()
kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
x0 = torch.as_tensor(sample_xinit(B, seed=0), dtype=torch.float32, device="cuda")
Q = torch.as_tensor(np.tile(np.diag(q.numpy()), (T, B, 1, 1)), dtype=torch.float32, device="cuda")
pv = torch.as_tensor(np.tile(pp.numpy(), (T, B, 1)), dtype=torch.float32, device="cuda")
cost = QuadCost(Q, pv)
for graph in (False, True):
    solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=10, exit_unconverged=False, quiet=True, graph=graph, **kw)
    for _ in range(5):
        solver((x0, cost, dx))
    torch.cuda.synchronize()
    ts, tr = [], []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver((x0, cost, dx))
        t1 = time.perf_counter()
        ts.append(t1 - t0)
    print("graph=%s: %.1f us wall per call incl. its read-back (median of 50)" % (graph, 1e6 * sorted(ts)[25]))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        solver((x0, cost, dx))
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)
