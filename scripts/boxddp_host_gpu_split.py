"""Config 2 (pendulum box-DDP, B=128, T=20, 10 iterations): wall time per solve against GPU-busy time (HIP events
around the enqueued work) - how much of a solve is host-side glue."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import BoxDDP, PendulumDx, QuadCost
from chainer_differentiable_mpc_amd.pendulum import sample_xinit
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")
dx = PendulumDx()
q, pp = dx.get_true_obj()
Q = torch.diag(q).to(dev)[None, None].expand(20, B, -1, -1).contiguous()
pv = pp.to(dev)[None, None].expand(20, B, -1).contiguous()
x0 = torch.as_tensor(sample_xinit(B, seed=0), dtype=torch.float32, device=dev)
solver = BoxDDP(20, dx.lower, dx.upper, B, 3, 1, None, eps=dx.mpc_eps, max_iter=10, exit_unconverged=False,
                line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter, quiet=True)
warnings.simplefilter("ignore")
with torch.no_grad():
    for _ in range(5): solver((x0, QuadCost(Q, pv), dx))
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): solver((x0, QuadCost(Q, pv), dx))
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("B=%d: wall %.3f ms per solve, GPU span %.3f ms per solve, iterations %d" % (B, (t1 - t0) / reps * 1e3, e0.elapsed_time(e1) / reps, solver.n_iter))
    # host-only cost: time to ENQUEUE one solve's work without the final synchronisation is not separable here (the
    # solve ends with one .cpu()); profile the Python side instead
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): solver((x0, QuadCost(Q, pv), dx))
    pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
