"""The headline step (B = 4096, T = 50, (8,2)) issued on ONE stream (every launch waits for the whole previous one: all 1,024
wavefronts walk sweep -> rollout in lockstep, and HBM idles through every rollout) against the same K steps issued round-robin on
S streams (a CU takes the next launch's workgroup as soon as its own finishes: phases drift apart, sweeps of one launch read
while another's rollouts compute), and against each step cut into S launches of B / S trajectories on S streams.
    python scripts/headline_streams.py [K]      (on the GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, T, nx, nu = 4096, 50, 8, 2
dev = torch.device("cuda:0")
N_SETS = 4


def make(Bp, n_sets, seed0):
    return [bench.make_inputs(Bp, T, nx, nu, seed=seed0 + k, device=dev)[1] for k in range(n_sets)]


def run(label, parts, n_streams, K):
    """parts: launches per step (each of B / parts trajectories); n_streams: streams the launches rotate over"""
    Bp = B // parts
    sets = make(Bp, N_SETS * parts, 7000 + 100 * parts)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream(dev)]
    outs = [(torch.empty((T, Bp, nx), device=dev), torch.empty((T, Bp, nu), device=dev)) for _ in range(max(n_streams, parts))]
    n = [0]

    def step():
        for _ in range(parts):
            k = n[0]
            n[0] += 1
            e = sets[k % len(sets)]
            with torch.cuda.stream(streams[k % n_streams]):
                solve_device(e["C"], e["c"], e["F"], e["f"], e["x_init"], None, T, nx, nu, out=outs[k % len(outs)])

    for _ in range(1500):
        step()
    torch.cuda.synchronize()
    res = []
    for _ in range(11):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / K * 1e6)
    res.sort()
    print("%-58s K=%d: median %.2f us per step (min %.2f, max %.2f) = %.3f of the HBM roof" % (
        label, K, res[5], res[0], res[-1], 170393600 / (res[5] * 1e-6) / 8e12), flush=True)


for K_ in (K, 20):
    run("one stream", 1, 1, K_)
    run("whole batches round-robin on 2 streams", 1, 2, K_)
    run("whole batches round-robin on 3 streams", 1, 3, K_)
    run("whole batches round-robin on 4 streams", 1, 4, K_)
    run("each step as 2 launches of 2,048 on 2 streams", 2, 2, K_)
    run("each step as 4 launches of 1,024 on 4 streams", 4, 4, K_)
    run("each step as 2 launches of 2,048 on ONE stream", 2, 1, K_)
