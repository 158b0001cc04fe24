"""How the headline solve's time depends on how long the GPU has been busy: blocks of 100 back-to-back HBM-streamed solves
(4 input sets in rotation), timed with HIP events, from a cold start (2 s idle) to several seconds of load."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
dev = torch.device("cuda")
B, T, nx, nu = 4096, 50, 8, 2
sets = [bench.make_inputs(B, T, nx, nu, s, dev)[1] for s in range(4)]
x = torch.empty((T, B, nx), device=dev); u = torch.empty((T, B, nu), device=dev)
solve_device(sets[0]["C"], sets[0]["c"], sets[0]["F"], sets[0]["f"], sets[0]["x_init"], None, T, nx, nu, out=(x, u))
torch.cuda.synchronize()
for trial in range(2):
    time.sleep(2.0)
    t_start = time.perf_counter()
    k = 0
    out = []
    while time.perf_counter() - t_start < float(os.environ.get("SECONDS", "4")):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20 if k < 10 else 100
        e0.record()
        for i in range(n):
            e = sets[(k + i) % 4]
            solve_device(e["C"], e["c"], e["F"], e["f"], e["x_init"], None, T, nx, nu, out=(x, u))
        e1.record()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t_start, e0.elapsed_time(e1) / n * 1e3))
        k += 1
    pick = [0, 1, 2, 4, 8, 12, 16, 24, 40, 80, 160, 320, 640, len(out) - 1]
    print("trial %d (after 2 s idle): " % trial + "  ".join("%.0f ms: %.1f us" % (out[i][0] * 1e3, out[i][1]) for i in pick if i < len(out)))
