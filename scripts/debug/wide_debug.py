import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import synthetic, _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import lqr as olqr
for (nx, nu) in ((12, 4), (16, 4)):
    B, T = 8, 5
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1, with_f=True)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
    x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=True)
    print(nx, nu, _lib.last_kernel_name()[:60])
    Ks, ks, x, u = (a.cpu().numpy() for a in (Ks, ks, x, u))
    for t in range(T - 1, -1, -1):
        print(" t=%d  Ks err %.2e  ks err %.2e   x err %.2e  u err %.2e" % (t, np.abs(Ks[t] - Ksr[t]).max(), np.abs(ks[t] - ksr[t]).max(),
                                                                     np.abs(x[t] - xr[t]).max(), np.abs(u[t] - ur[t]).max()))
    t = T - 2
    print(" Ks[T-2] err per (b, m):\n", np.abs(Ks[t] - Ksr[t]).max(axis=2).round(4))
    print(" Ks[T-2][0] err per column:\n", np.abs(Ks[t][0] - Ksr[t][0]).round(4))
    if (nx, nu) == (12, 4):
        b = 0
        x0, u0 = xr[0][b], ur[0][b]
        F0, f0 = p["F"][0][b], p["f"][0][b]
        print(" x[1] got   ", x[1][b].round(3))
        print(" x[1] ref   ", xr[1][b].round(3))
        print(" Fx x0 + f  ", (F0[:, :nx] @ x0 + f0).round(3))
        print(" Fx x0      ", (F0[:, :nx] @ x0).round(3))
        print(" F tau      ", (F0 @ np.concatenate((x0, u0))).round(3))
        tau0 = np.concatenate((x0, u0))
        for tt in range(T - 1):
            for bb in range(B):
                cand = p["F"][tt][bb] @ tau0 + p["f"][tt][bb]
                if np.abs(cand - x[1][b]).max() < 1e-3: print(" MATCH F[%d][%d] with f" % (tt, bb))
                cand = p["F"][tt][bb] @ tau0 + p["f"][0][b]
                if np.abs(cand - x[1][b]).max() < 1e-3: print(" MATCH F[%d][%d] with own f" % (tt, bb))
        # per-row least squares: which 17-vector w gives got_i = w . [tau0; 1]?  compare with rows of F
        print(" got - ref  ", (x[1][b] - xr[1][b]).round(3))
        Fu = F0[:, nx:]
        print(" Fu u0      ", (Fu @ u0).round(3))
        print(" u0", u0.round(3), " x0", x0.round(3))
        for sh in range(0):
            Fu = np.roll(F0, sh, axis=1)[:, nx:]
            print(" shift", sh, (F0[:, :nx] @ x0 + f0 + Fu @ u0).round(3))
