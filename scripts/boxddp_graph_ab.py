"""Config 2 / config 4's solve (pendulum box-DDP, T = 20, 10 iLQR iterations, device-driven chain of 22-23 launches): the chain
launched as it is against the same chain captured ONCE in a hipGraph and replayed - device time per solve by HIP events around
20 solves back to back (no read-back inside), and wall time per solve with the one read-back a solve ends with.
    python scripts/boxddp_graph_ab.py      (on the GPU box)"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_differentiable_mpc_amd import BoxDDP, PendulumDx, QuadCost  # noqa: E402
from chainer_differentiable_mpc_amd.pendulum import sample_xinit  # noqa: E402

warnings.simplefilter("ignore")
T = 20
dx = PendulumDx()
q, pp = dx.get_true_obj()
kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
for B in (128, 1024):
    x0 = torch.as_tensor(sample_xinit(B, seed=0), dtype=torch.float32, device="cuda")
    Q = torch.as_tensor(np.tile(np.diag(q.numpy()), (T, B, 1, 1)), dtype=torch.float32, device="cuda")
    pv = torch.as_tensor(np.tile(pp.numpy(), (T, B, 1)), dtype=torch.float32, device="cuda")
    cost = QuadCost(Q, pv)
    solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=10, exit_unconverged=False, quiet=True, lazy_status=True, **kw)
    for _ in range(3):
        x, u, c = solver((x0, cost, dx))
        _ = solver.status
    torch.cuda.synchronize()

    def direct():
        return solver((x0, cost, dx))

    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        xg, ug, cg = solver((x0, cost, dx))
        _ = solver.status
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            xg, ug, cg = solver((x0, cost, dx))
    solver._pending = None       # (the captured call's deferred read-back belongs to the graph, not to a later call)
    torch.cuda.synchronize()
    xd, ud, cd = direct()
    _ = solver.status
    g.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(xd, xg)) and bool(torch.equal(ud, ug)) and bool(torch.equal(cd, cg))

    def dev_time(fn, reps=20, resolve=False):
        res = []
        for _ in range(7):
            fn()
            if resolve:
                _ = solver.status
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
                if resolve:
                    solver._pending = None
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
        return sorted(res)[3]

    def wall_time(fn, after, reps=5):
        res = []
        for _ in range(7):
            fn(); after(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
                after()
            res.append((time.perf_counter() - t0) / reps * 1e6)
        return sorted(res)[3]

    d_dev = dev_time(direct, resolve=True)
    g_dev = dev_time(g.replay)
    d_wall = wall_time(direct, lambda: solver.status)
    g_wall = wall_time(g.replay, lambda: cg[0].item())
    print("B=%-5d chain launched: %.1f us device per solve back to back, %.1f us wall per solve with its read-back | "
          "hipGraph replay: %.1f us device, %.1f us wall with a read-back | same results: %s" % (B, d_dev, d_wall, g_dev, g_wall, same), flush=True)
