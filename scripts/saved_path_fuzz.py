"""Randomised sweep of the saved-gains DiffLqr path against the full second solve (device vs device) and, on a subset,
the oracle: shapes with an affine stream, horizons 2..NSTASH+1, batch sizes 4..68."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import DiffLqr, synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import saving_solve_available
from oracle import kkt as okkt, lqr as olqr
rng = np.random.RandomState(11)
dev = torch.device("cuda")
worst = 0.0; n_saved = n_plain = 0
KEYS = ("d_x_init", "dC", "dc", "dF", "df")
for it in range(160):
    nx, nu = [(8, 2), (4, 2), (2, 2)][rng.randint(3)]
    T = int(rng.choice([2, 3, 4, 5, 7, 10, 17, 31, 49, 50, 51, 52, 53, 60]))
    B = int(4 * rng.randint(1, 18))
    with_f = bool(rng.randint(2))
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=100 + it, with_f=with_f)
    d = {k: (torch.as_tensor(v, dtype=torch.float32, device=dev) if v is not None else None) for k, v in p.items()}
    gx = torch.as_tensor(rng.randn(T, B, nx), dtype=torch.float32, device=dev)
    gu = torch.as_tensor(rng.randn(T, B, nu), dtype=torch.float32, device=dev)
    a = DiffLqr(T, B, nx, nu); b = DiffLqr(T, B, nx, nu, save_gains=False)
    xa, ua = a.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    xb, ub = b.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    used = a._retained["saved"] is not None
    assert used == saving_solve_available(T, B, nx, nu)
    n_saved += used; n_plain += not used
    oa = a.backward((0, 1, 2, 3, 4), (gx, gu)); ob = b.backward((0, 1, 2, 3, 4), (gx, gu))
    torch.cuda.synchronize()
    assert int(a.info.abs().max()) == 0
    for ga, gb, key in [(xa, xb, "x"), (ua, ub, "u")] + list(zip(oa, ob, KEYS)):
        ga, gb = ga.cpu().numpy(), gb.cpu().numpy()
        e = float(np.abs(ga - gb).max() / max(1.0, np.abs(gb).max()))
        worst = max(worst, e)
        assert e <= 2e-4, (nx, nu, T, B, with_f, key, e)
    if it % 16 == 0:    # the oracle on a subset
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx.cpu().numpy().astype(np.float64),
                                    gu.cpu().numpy().astype(np.float64), T, nx, nu)
        for g, want, key in zip(oa, ref, KEYS):
            e = float(np.abs(g.cpu().numpy() - want).max() / max(1.0, np.abs(want).max()))
            assert e <= 5e-4, ("oracle", nx, nu, T, B, key, e)
print("160 cases: %d on the saved path, %d on the plain one; worst saved-vs-full difference %.2e" % (n_saved, n_plain, worst))
