"""scripts/headline_streams.py with the host taken out: N headline steps captured in ONE hipGraph per arrangement (a chain on one
stream; whole batches alternating over S parallel chains; every step cut into S launches of B / S trajectories, one chain per
part, optionally with the chains started a fraction of a launch apart), the graph replayed back to back.
    python scripts/headline_streams_graph.py      (on the GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device  # noqa: E402

B, T, nx, nu = 4096, 50, 8, 2
dev = torch.device("cuda:0")
N_SETS = 4
STEPS = 40          # per graph


def run(label, parts, chains, offset_T=0):
    Bp = B // parts
    sets = [bench.make_inputs(Bp, T, nx, nu, seed=9000 + 100 * parts + k, device=dev)[1] for k in range(N_SETS * parts)]
    outs = [(torch.empty((T, Bp, nx), device=dev), torch.empty((T, Bp, nu), device=dev)) for _ in range(max(chains, parts))]
    short = None
    if offset_T:
        short = bench.make_inputs(Bp, offset_T, nx, nu, seed=77, device=dev)[1]
        short_out = (torch.empty((offset_T, Bp, nx), device=dev), torch.empty((offset_T, Bp, nu), device=dev))
    side = [torch.cuda.Stream(dev) for _ in range(chains - 1)]

    def launches(streams):
        if short is not None:      # chain c > 0 starts c * offset_T steps of a solve late
            for c in range(1, chains):
                with torch.cuda.stream(streams[c]):
                    for _ in range(c):
                        solve_device(short["C"], short["c"], short["F"], short["f"], short["x_init"], None, offset_T, nx, nu, out=short_out)
        for k in range(STEPS * parts):
            e = sets[k % len(sets)]
            with torch.cuda.stream(streams[k % chains]):
                solve_device(e["C"], e["c"], e["F"], e["f"], e["x_init"], None, T, nx, nu, out=outs[k % len(outs)])

    cur = torch.cuda.current_stream(dev)
    launches([cur] + side)          # (first-call costs outside the capture)
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        for s in side:
            s.wait_stream(cap)
        launches([cap] + side)
        for s in side:
            cap.wait_stream(s)
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    res = []
    for _ in range(11):
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / (5 * STEPS) * 1e6)
    res.sort()
    print("%-72s median %.2f us per step (min %.2f, max %.2f) = %.3f of the HBM roof" % (
        label, res[5], res[0], res[-1], 170393600 / (res[5] * 1e-6) / 8e12), flush=True)


run("one chain", 1, 1)
run("whole batches over 2 chains", 1, 2)
run("whole batches over 2 chains, second started 25 steps late", 1, 2, 25)
run("whole batches over 3 chains", 1, 3)
run("each step as 2 launches of 2,048, 2 chains", 2, 2)
run("each step as 2 launches of 2,048, 2 chains, second started 25 steps late", 2, 2, 25)
run("each step as 2 launches of 2,048, 2 chains, second started 35 steps late", 2, 2, 35)
run("each step as 4 launches of 1,024, 4 chains", 4, 4)
run("each step as 4 launches of 1,024, 4 chains, started 12 steps apart", 4, 4, 12)
run("each step as 2 launches of 2,048, ONE chain", 2, 1)
