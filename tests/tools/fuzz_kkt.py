"""Randomised parity sweep of the analytic KKT gradient (second solve + co-state kernel, DMA and plain variants) over
shapes, horizons and batch sizes against the oracle.  Diagnostic, GPU box: python tests/tools/fuzz_kkt.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd import differentiable_lqr as dl
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import kkt as okkt, lqr as olqr

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SHAPES = [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2), (4, 2), (6, 2), (8, 2), (4, 4), (8, 4), (12, 3), (5, 2), (7, 3)]
dev = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda")
worst = {}
for case in range(n_cases):
    nx, nu = SHAPES[rng.randint(len(SHAPES))]
    T = int(rng.randint(2, 14))
    B = int(rng.choice([1, 3, 4, 5, 8, 12, 17, 32]))
    strict = bool(rng.randint(2))
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(rng.randint(1 << 30)), with_f=True)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    gx, gu = rng.standard_normal((T, B, nx)), rng.standard_normal((T, B, nu))
    ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
    d = {k: dev(v) for k, v in p.items()}
    x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
    got = dl.kkt_grad_device(d["C"], d["c"], d["F"], x, u, dev(gx), dev(gu), T, nx, nu, strict_math=strict)
    for g, w, key in zip(got, ref, ("d_x_init", "dC", "dc", "dF", "df")):
        if w is None:
            continue
        err = float(np.max(np.abs(g.cpu().numpy() - w) / np.maximum(1.0, np.abs(w))))
        worst[key] = max(worst.get(key, 0.0), err)
        if err > 5e-4:
            print("case %d (B=%d T=%d nx=%d nu=%d strict=%s): %s err %.2e" % (case, B, T, nx, nu, strict, key, err))
print("fuzz done: %d cases, worst %s" % (n_cases, {k: "%.1e" % v for k, v in worst.items()}))
