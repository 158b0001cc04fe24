"""Randomised parity sweep of the MPC step (forward = backward_rec with PNQP + line search) over shapes, incl. the
runtime-dimension kernels, against the per-trajectory oracle.  Diagnostic, GPU box: python tests/tools/fuzz_mpc_generic.py [n] [seed]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost, synthetic
from oracle import mpc as ompc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda")
worst, forks = 0.0, 0
for case in range(n_cases):
    nu = int(rng.randint(1, 9))
    nx = int(rng.randint(1, min(40, 63 - nu) + 1))
    B, T = int(rng.randint(1, 7)), int(rng.randint(2, 9))
    bound = float(rng.choice([0.125, 0.25, 0.5, 1.0, 4.0]))
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(rng.randint(1 << 30)), with_f=True)
    lo, hi = -bound * np.ones((T, B, nu)), bound * np.ones((T, B, nu))
    u0 = np.zeros((T, B, nu))
    xs = [p["x_init"]]
    for t in range(T - 1):
        xs.append(np.einsum("bij,bj->bi", p["F"][t], np.concatenate((xs[t], u0[t]), axis=1)) + p["f"][t])
    x0 = np.stack(xs).astype(np.float32).astype(np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xr, ur, bo, fo, Ksr, ksr = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u0, x0, lo, hi, ompc.QuadCost(p["C"], p["c"]),
                                                    ompc.LinDx(p["F"], p["f"]), 0.2, 5, T, nx, nu, need_expand=True, batch_coupled=False)
        step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
                       LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=True)
        x, u = step.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    eu = np.abs(u.cpu().numpy() - ur) / np.maximum(1.0, np.abs(ur))
    ex = np.abs(x.cpu().numpy() - xr) / np.maximum(1.0, np.abs(xr))
    per_traj = np.maximum(eu.max(axis=(0, 2)), ex.max(axis=(0, 2)))
    bad = per_traj > 5e-4
    if bad.any():   # a different active set / step size on a float32 tie is a legitimate fork; count, do not hide
        forks += int(bad.sum())
        print("case %d (B=%d T=%d nx=%d nu=%d bound=%g): %d trajectories differ, worst %.2e" % (case, B, T, nx, nu, bound, int(bad.sum()), per_traj.max()))
    worst = max(worst, float(per_traj[~bad].max()) if (~bad).any() else 0.0)
print("fuzz done: %d cases, worst agreeing error %.2e, %d forked trajectories" % (n_cases, worst, forks))
