import os, sys, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import BoxDDP, PendulumDx, QuadCost
from chainer_differentiable_mpc_amd.pendulum import sample_xinit
B, T = 128, 20
dx = PendulumDx(); q, pp = dx.get_true_obj()
x0 = torch.as_tensor(sample_xinit(B, seed=0), dtype=torch.float32, device="cuda")
Q = torch.as_tensor(np.tile(np.diag(q.numpy()), (T, B, 1, 1)), dtype=torch.float32, device="cuda")
pv = torch.as_tensor(np.tile(pp.numpy(), (T, B, 1)), dtype=torch.float32, device="cuda")
kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=10, exit_unconverged=False, quiet=True, **kw)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    solver((x0, QuadCost(Q, pv), dx))
import cProfile, pstats
pr = cProfile.Profile()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pr.enable()
    for _ in range(5): solver((x0, QuadCost(Q, pv), dx))
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
