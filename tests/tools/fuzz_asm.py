"""Randomised parity sweep of the fused solve (generated stream, both forward variants, masked / gains / no-f
options, ragged batches, horizons across the variant limits) against the numpy oracle.  Diagnostic; GPU.
    python tests/tools/fuzz_asm.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from chainer_differentiable_mpc_amd import LQR_active, _lib, synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import lqr as olqr
from oracle import mpc as ompc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
shapes = [(8, 2), (3, 1), (4, 2), (6, 2), (2, 2), (1, 1), (2, 1), (3, 2)]
worst, paths = 0.0, {}
for case in range(n_cases):
    nx, nu = shapes[rng.randint(len(shapes))]
    B = int(rng.choice([4, 5, 7, 8, 12, 16, 33, 64, 100]))
    T = int(rng.choice([2, 3, 4, 5, 7, 20, 49, 50, 51, 52, 64, 65, 74, 75, 80]))
    with_f = bool(rng.randint(2))
    mode = rng.randint(3)          # 0 plain, 1 gains, 2 masked
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(rng.randint(1 << 30)), with_f=with_f)
    d = {k: (None if v is None else torch.as_tensor(v, dtype=torch.float32, device="cuda")) for k, v in p.items()}
    path = _lib.load().dmpc_lqr_solve_path(T, B, nx, nu)
    paths[path] = paths.get(path, 0) + 1
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1, np.abs(b))))
    if mode == 2:
        act = rng.rand(T, B, nu) < 0.35
        xr, ur = ompc.lqr_active_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], act, T, nx, nu)
        x, u = LQR_active(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu, u_zero_Index=torch.as_tensor(act).cuda()).solve_recursion()
        err = max(rel(x.cpu().numpy(), xr), rel(u.cpu().numpy(), ur))
        assert np.all(u.cpu().numpy()[act] == 0)
        tol = 2e-4
    else:
        info = torch.zeros(B, dtype=torch.int32, device="cuda")
        x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=(mode == 1), info=info)
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        err = max(rel(x.cpu().numpy(), xr), rel(u.cpu().numpy(), ur))
        if mode == 1:
            Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
            err = max(err, rel(Ks.cpu().numpy(), Ksr), rel(ks.cpu().numpy(), ksr))
        assert int(info.abs().max()) == 0
        tol = 1e-4 if T <= 52 else 5e-4
    worst = max(worst, err)
    if err > tol:
        print("FAIL case %d: (%d,%d) B=%d T=%d f=%d mode=%d path=%d err %.3e" % (case, nx, nu, B, T, with_f, mode, path, err))
        sys.exit(1)
print("fuzz ok: %d cases, worst %.2e, paths %s" % (n_cases, worst, dict(sorted(paths.items()))))
