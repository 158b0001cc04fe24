"""Quick parity check of the fused solve against the numpy oracle over shapes / options (diagnostic, GPU).
Lives under tests/ because it uses the oracle (test infrastructure); run as `python tests/tools/asm_check.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from chainer_differentiable_mpc_amd import synthetic, _lib
from chainer_differentiable_mpc_amd.lqr_recursion import LqrRecursion, solve_device
from oracle import lqr as olqr

def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1, np.abs(b))))

worst = 0.0
cases = [(64, 10, 8, 2), (4, 2, 8, 2), (5, 3, 8, 2), (7, 4, 8, 2), (9, 5, 8, 2), (33, 7, 8, 2), (130, 50, 8, 2),
         (16, 20, 3, 1), (6, 9, 4, 2), (11, 13, 6, 2), (8, 6, 2, 2), (8, 6, 1, 1), (8, 6, 2, 1), (8, 6, 3, 2), (4096, 50, 8, 2)]
for (B, T, nx, nu) in cases:
    for has_f in (True, False):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=B + T)
        d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
        f = d["f"] if has_f else None
        rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], f, T, nx, nu)
        x, u = rec.solve_recursion()
        Ks, ks = rec.backward()          # separate (HIP) backward sweep kernel
        x2, u2, K2, k2 = solve_device(d["C"], d["c"], d["F"], f, d["x_init"], None, T, nx, nu, want_gains=True)
        torch.cuda.synchronize()
        eK = rel(K2.cpu().numpy(), torch.stack(Ks).cpu().numpy()); ek = rel(k2.cpu().numpy(), torch.stack(ks).cpu().numpy())
        e2 = max(rel(x2.cpu().numpy(), x.cpu().numpy()), rel(u2.cpu().numpy(), u.cpu().numpy()))
        worst = max(worst, eK, ek, e2)
        fo = p["f"] if has_f else None
        nb = min(B, 256)
        sl = lambda a, ax: np.take(a, range(nb), axis=ax)
        xr, ur = olqr.lqr_solve(p["x_init"][:nb], sl(p["C"], 1), sl(p["c"], 1), sl(p["F"], 1),
                                None if fo is None else sl(fo, 1), T, nx, nu)
        ex, eu = rel(x.cpu().numpy()[:, :nb], xr), rel(u.cpu().numpy()[:, :nb], ur)
        worst = max(worst, ex, eu)
        print("B=%d T=%d (%d,%d) f=%d  err x %.2e u %.2e  K %.1e k %.1e x2 %.1e info %s" % (B, T, nx, nu, has_f, ex, eu, eK, ek, e2,
              None if rec.info is None else int(rec.info.abs().max())), flush=True)
print("worst", worst)
sys.exit(0 if worst < 1e-4 else 1)
