"""Wall time of BASELINE.json configs[1]: pendulum box-DDP, batch=128, T=20 (host loop over the MPC-step kernels)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import BoxDDP, PendulumDx, QuadCost
from chainer_differentiable_mpc_amd.pendulum import sample_xinit
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 20)
dx = PendulumDx()
q, pp = dx.get_true_obj()
x0 = torch.as_tensor(sample_xinit(B, seed=0), dtype=torch.float32, device="cuda")
Q = torch.as_tensor(np.tile(np.diag(q.numpy()), (T, B, 1, 1)), dtype=torch.float32, device="cuda")
pv = torch.as_tensor(np.tile(pp.numpy(), (T, B, 1)), dtype=torch.float32, device="cuda")
kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
for device_loop in (True, False):
    for it in (1, 5, 10):
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=it, exit_unconverged=False, quiet=True,
                        device_loop=device_loop, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ts = []
            for rep in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                x, u, costs = solver((x0, QuadCost(Q, pv), dx))
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("BoxDDP pendulum B=%d T=%d max_iter=%2d %s loop: %.2f ms per call (status %s), mean cost %.4f" % (
            B, T, it, "device" if device_loop else "host  ", sorted(ts)[len(ts) // 2] * 1e3, solver.status,
            float(costs.mean())))
