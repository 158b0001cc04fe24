"""CPU: pin the numpy oracle against (1) the notebook known answers the reference holds
and (2) golden vectors recorded from the unmodified reference (tests/golden/make_golden.py)."""
import glob
import os
import warnings

import numpy as np
import pytest

from chainer_differentiable_mpc_amd import synthetic
from oracle import kkt, linalg, lqr, mpc, pnqp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def checksum(d):
    return float(sum(np.abs(v).sum() for v in d.values() if v is not None))


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


# ----------------------------------------------------------------- notebook anchors
def test_anchor_one_variable_lqr():
    """examples/LQR_recursion_solver_one_variable.ipynb:226-245 (gains), :346-382 (states)."""
    T, nx, nu = 20, 2, 1
    F = np.tile(np.array([[1.0, 1.0, 0], [0, 1.0, 1.0]]), (T, 1, 1, 1))
    c = np.zeros((T, 1, 3))
    C = np.tile(np.array([[1.0, 0, 0], [0, 0, 0], [0, 0, 10]]), (T, 1, 1, 1))
    x0 = np.array([[1.0, 0.0]])
    Ks, ks = lqr.lqr_backward(C, c, F, None, T, nx, nu)
    np.testing.assert_allclose(Ks[0, 0, 0], [-0.21140641, -0.7644787], atol=5e-9)
    np.testing.assert_allclose(Ks[9, 0, 0], [-0.21102155, -0.76280464], atol=5e-9)
    np.testing.assert_allclose(Ks[16, 0, 0], [-0.19254658, -0.50931677], atol=5e-9)
    np.testing.assert_allclose(Ks[17, 0, 0], [-0.09090909, -0.18181818], atol=5e-9)
    assert np.all(Ks[18:] == 0)
    x, u = lqr.lqr_solve(x0, C, c, F, None, T, nx, nu)
    np.testing.assert_allclose(x[1, 0], [1.0, -2.11406412e-01], atol=5e-10)
    np.testing.assert_allclose(x[8, 0], [-3.96705017e-02, -1.28355152e-04], atol=5e-11)
    np.testing.assert_allclose(x[19, 0], [1.00328598e-03, -3.91198810e-04], atol=5e-12)
    a = load("anchors.npz")
    np.testing.assert_allclose(Ks, a["onevar_Ks"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(x, a["onevar_x"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(u, a["onevar_u"], rtol=0, atol=1e-13)


def test_anchor_boyd_lqr():
    """examples/Boyd_lqr.ipynb:508-558: steady-state gain and the trailing -0. gains."""
    T, nx, nu = 51, 3, 1
    F = np.tile(np.array([[1.0, 0, 0, 1], [1, 1.0, 0, 0], [0, 1, 1, 0]]), (T, 1, 1, 1))
    c = np.zeros((T, 1, 4))
    C = np.tile(np.diag([0, 0, 1.0, 1.0]), (T, 1, 1, 1))
    C[T - 1, 0, 3, 3] = 0.00000000000001
    x0 = np.array([[0.5428, 0.7633, 0.3504]])
    Ks, ks = lqr.lqr_backward(C, c, F, None, T, nx, nu)
    np.testing.assert_allclose(Ks[0, 0, 0], [-1.86152282, -1.34921019, -0.35888729], atol=5e-9)
    assert np.all(Ks[-3:] == 0)
    x, u = lqr.lqr_solve(x0, C, c, F, None, T, nx, nu)
    a = load("anchors.npz")
    np.testing.assert_allclose(Ks, a["boyd_Ks"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(x, a["boyd_x"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(u, a["boyd_u"], rtol=0, atol=1e-12)


def test_anchor_pnqp_notebook():
    """experiment_mpc/Projected_Newton_Quadratic_Programming.py:67-68 (mpc.pytorch's answer)."""
    a = load("anchors.npz")
    x, (LU, piv), idx_f, it = pnqp.pnqp(a["pnqp_H"], a["pnqp_q"], a["pnqp_lower"], a["pnqp_upper"])
    expect = np.array([[0.1239, -0.0063, 0.0277, -0.6669], [0.6205, 0.2703, 0.4023, -0.0121]])
    np.testing.assert_allclose(x, expect, atol=5e-5)
    np.testing.assert_allclose(x, a["pnqp_x"], atol=2e-6)
    assert it == int(a["pnqp_it"])
    np.testing.assert_array_equal(idx_f, a["pnqp_idx_f"])


def test_anchor_lqrnet_iteration0():
    """examples/LQRnet.ipynb:184: loss 0.661925 at iteration 0, dynamics mse 4.774785 after one
    RMSprop step (pins the sign pattern of dF).  Restated in tests/lqrnet_anchor.py."""
    from tests.lqrnet_anchor import run_iteration0
    loss0, mse1 = run_iteration0()
    assert abs(loss0 - 0.661925) < 5e-7
    assert abs(mse1 - 4.774785) < 5e-7


# ----------------------------------------------------------------- goldens rows A, B
LQR_FILES = sorted(glob.glob(os.path.join(GOLDEN, "lqr_*.npz")))


@pytest.mark.parametrize("path", LQR_FILES, ids=[os.path.basename(p) for p in LQR_FILES])
def test_lqr_and_kkt_golden(path):
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=bool(g["with_f"]))
    assert abs(checksum(p) - float(g["in_checksum"])) < 1e-9 * float(g["in_checksum"])
    Ks, ks = lqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    x, u = lqr.lqr_forward(Ks, ks, p["x_init"], p["F"], p["f"], T, nx, nu)
    tol = dict(rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(Ks, g["Ks"], **tol)
    np.testing.assert_allclose(ks, g["ks"], **tol)
    np.testing.assert_allclose(x, g["x"], **tol)
    np.testing.assert_allclose(u, g["u"], **tol)
    out = kkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], x, u, g["grad_x"], g["grad_u"], T, nx, nu)
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        np.testing.assert_allclose(got, g[key], rtol=1e-9, atol=1e-9, err_msg=key)


def test_kkt_gradient_matches_finite_differences():
    """dc, dx_init, dF are true gradients; dC diag is 1.5x FD and df is shifted (reference quirks)."""
    B, T, nx, nu = 1, 4, 2, 1
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=5, fp32_representable=False)
    rng = np.random.RandomState(6)
    gx, gu = rng.randn(T, B, nx), rng.randn(T, B, nu)

    def loss(**kw):
        q = dict(p)
        q.update(kw)
        x, u = lqr.lqr_solve(q["x_init"], q["C"], q["c"], q["F"], q["f"], T, nx, nu)
        return (gx * x).sum() + (gu * u).sum()

    x, u = lqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    dx0, dC, dc, dF, df = kkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], x, u, gx, gu, T, nx, nu)
    sx0, sC, sc, sF, sf = kkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], x, u, gx, gu, T, nx, nu,
                                               strict_math=True)
    eps = 1e-6

    def fd(name, idx):
        a = p[name].copy(); a[idx] += eps
        b = p[name].copy(); b[idx] -= eps
        return (loss(**{name: a}) - loss(**{name: b})) / (2 * eps)

    # +drl in the second solve and the un-negated outer products compensate: outputs ARE the gradients
    assert abs(dc[1, 0, 2] - fd("c", (1, 0, 2))) < 1e-6
    assert abs(dx0[0, 1] - fd("x_init", (0, 1))) < 1e-6
    assert abs(dF[1, 0, 1, 2] - fd("F", (1, 0, 1, 2))) < 1e-6
    assert abs(sf[1, 0, 0] - fd("f", (1, 0, 0))) < 1e-6          # strict: d_lambda[t+1]
    assert abs(df[2, 0, 0] - fd("f", (1, 0, 0))) < 1e-6          # quirk: reference df[t+1] holds it
    assert abs(sC[1, 0, 1, 1] - fd("C", (1, 0, 1, 1))) < 1e-6
    assert abs(dC[1, 0, 1, 1] - 1.5 * fd("C", (1, 0, 1, 1))) < 1e-6


# ----------------------------------------------------------------- golden row D
@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_lu_golden(n):
    g = load("lu_n%d.npz" % n)
    LU, piv = linalg.batch_lu_factor(g["A"])
    np.testing.assert_array_equal(piv, g["piv"])
    assert piv.dtype == np.int32
    np.testing.assert_allclose(LU, g["LU"], rtol=1e-12, atol=1e-13)
    x2 = linalg.batch_lu_solve((LU, piv), g["b2"])
    x3 = linalg.batch_lu_solve((LU, piv), g["b3"])
    assert x2.dtype == np.float32 and x3.dtype == np.float32
    np.testing.assert_allclose(x2, g["x2"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(x3, g["x3"], rtol=2e-5, atol=2e-6)


# ----------------------------------------------------------------- golden row C
@pytest.mark.parametrize("n", [1, 2, 4, 8])
@pytest.mark.parametrize("tag", ["cold", "warm"])
def test_pnqp_golden(n, tag):
    g = load("pnqp_n%d.npz" % n)
    B = int(g["B"])
    p = synthetic.make_box_qp(B, n, seed=int(g["seed"]), bound=0.5)
    x0 = None if tag == "cold" else g["warm"]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x, fac, idx_f, it = pnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], x_init=x0, n_iter=20)
    assert (len(w) > 0) == bool(g[tag + "_warned"])
    assert it == int(g[tag + "_it"])
    np.testing.assert_array_equal(idx_f, g[tag + "_idx_f"])
    np.testing.assert_allclose(x, g[tag + "_x"], rtol=1e-5, atol=2e-6)
    if n == 1:
        np.testing.assert_allclose(fac, g[tag + "_Hf"], rtol=1e-12)
    else:
        np.testing.assert_array_equal(fac[1], g[tag + "_piv"])
        np.testing.assert_allclose(fac[0], g[tag + "_LU"], rtol=1e-9, atol=1e-12)
    # per-row semantics (= the reference with a batch of one per row)
    xr, _, idxr, _, info = pnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], x_init=x0, n_iter=20,
                                     batch_coupled=False, return_info=True, warn=False)
    np.testing.assert_array_equal(info["iters"], g[tag + "_row_it"])
    np.testing.assert_array_equal(idxr, g[tag + "_row_idx_f"])
    np.testing.assert_allclose(xr, g[tag + "_row_x"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(~info["converged"], g[tag + "_row_warned"])


# ----------------------------------------------------------------- golden rows E, F
MPC_FILES = sorted(glob.glob(os.path.join(GOLDEN, "mpc_*.npz")))


@pytest.mark.parametrize("path", MPC_FILES, ids=[os.path.basename(p) for p in MPC_FILES])
def test_mpc_step_golden(path):
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    bound = float(g["bound"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    assert abs(checksum(p) - float(g["in_checksum"])) < 1e-9 * float(g["in_checksum"])
    lo = -bound * np.ones((T, B, nu))
    hi = bound * np.ones((T, B, nu))
    x, u, back_out, for_out, Ks, ks = mpc.mpc_forward(
        p["C"], p["c"], p["F"], p["f"], g["u_nom"], g["x_nom"], lo, hi, mpc.QuadCost(p["C"], p["c"]),
        mpc.LinDx(p["F"], p["f"]), 0.2, 5, T, nx, nu, need_expand=bool(g["need_expand"]))
    assert back_out.n_total_qp_iter == int(g["n_total_qp_iter"])
    tol = dict(rtol=2e-5, atol=2e-6)          # float32 solves inside (util.py:522-527)
    np.testing.assert_allclose(u, g["u"], **tol)
    np.testing.assert_allclose(x, g["x"], **tol)
    np.testing.assert_allclose(for_out.costs, g["costs"], rtol=1e-5)
    np.testing.assert_allclose(for_out.objs, g["objs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(for_out.full_du_norm, g["full_du_norm"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(for_out.alpha_du_norm, g["alpha_du_norm"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(for_out.mean_alphas, g["mean_alphas"], rtol=1e-12)
    # backward from the GOLDEN forward outputs, so that both sides see the same active set
    active = (np.abs(g["u"] - lo) <= 1e-8) | (np.abs(g["u"] - hi) <= 1e-8)
    np.testing.assert_array_equal(active, g["active"])
    adx, adu = mpc.lqr_active_solve(np.zeros((B, nx)), p["C"], -np.concatenate((g["grad_x"], g["grad_u"]), axis=2),
                                    p["F"], None, active, T, nx, nu)
    np.testing.assert_allclose(adx, g["active_dx"], rtol=5e-5, atol=5e-6)
    np.testing.assert_allclose(adu, g["active_du"], rtol=5e-5, atol=5e-6)
    out = mpc.mpc_backward(g["x_nom"][0], p["C"], p["c"], p["F"], p["f"], g["x"], g["u"], lo, hi,
                           g["grad_x"], g["grad_u"], T, nx, nu)
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        np.testing.assert_allclose(got, g[key], rtol=1e-4, atol=1e-5, err_msg=key)


@pytest.mark.parametrize("path", MPC_FILES, ids=[os.path.basename(p) for p in MPC_FILES])
def test_mpc_step_per_trajectory_golden(path):
    """oracle(batch_coupled=False) == the reference run one trajectory at a time (row_* keys)"""
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    bound = float(g["bound"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    lo = -bound * np.ones((T, B, nu))
    hi = bound * np.ones((T, B, nu))
    x, u, back_out, for_out, Ks, ks = mpc.mpc_forward(
        p["C"], p["c"], p["F"], p["f"], g["u_nom"], g["x_nom"], lo, hi, mpc.QuadCost(p["C"], p["c"]),
        mpc.LinDx(p["F"], p["f"]), 0.2, 5, T, nx, nu, need_expand=bool(g["need_expand"]), batch_coupled=False)
    np.testing.assert_allclose(u, g["row_u"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(x, g["row_x"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(for_out.costs, g["row_costs"], rtol=1e-5)
    out = mpc.mpc_backward(g["x_nom"][0], p["C"], p["c"], p["F"], p["f"], g["row_x"], g["row_u"], lo, hi,
                           g["grad_x"], g["grad_u"], T, nx, nu)
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        np.testing.assert_allclose(got, g["row_" + key], rtol=1e-4, atol=1e-5, err_msg=key)


def test_box_ddp_trace_golden():
    """BoxDDP + LinDx/QuadCost (mpc/box_ddp.py:93-291): final x, u, costs and the stop reason of the reference"""
    from oracle import box_ddp
    g = load("boxddp_trace.npz")
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    x, u, costs, status, n_iter, *_ = box_ddp.box_ddp(p["x_init"], mpc.QuadCost(p["C"], p["c"]), mpc.LinDx(p["F"], p["f"]),
                                                     T, -float(g["bound"]), float(g["bound"]), nx, nu)
    assert status in str(g["stdout"])
    np.testing.assert_allclose(u, g["u"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(x, g["x"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(costs, g["costs"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["wide", "tight"])
def test_mpcnet_experiment_golden(tag):
    """The reference's experiment_mpc/MpcNet.py (first training iteration, its own sizes: T=5, (3,3), B=128): the oracle's
    box-DDP under the expert's and the learner's (A, B), the imitation loss (:80-90), and d loss / d(A, B) as the sum over
    time and batch of the final no-op node's dF (box_ddp.py:247-259, mpc_step.py:330-460) - against the reference's run."""
    from oracle import box_ddp
    g = load("mpcnet_experiment.npz")
    T, B, nx, nu = int(g["T"]), int(g["B"]), int(g["nx"]), int(g["nu"])
    ns, bound = nx + nu, float(g[tag + "_bound"])
    cost = mpc.QuadCost(np.tile(np.eye(ns), (T, B, 1, 1)), np.tile(g[tag + "_p"], (T, B, 1)))
    res = {}
    for who, A, Bm in (("true", g[tag + "_A_exp"], g[tag + "_B_exp"]), ("pred", g[tag + "_A0"], g[tag + "_B0"])):
        Fm = np.tile(np.concatenate((A, Bm), axis=1), (T - 1, B, 1, 1))
        x, u, costs, status, *_ = box_ddp.box_ddp(g[tag + "_x_init"], cost, mpc.LinDx(Fm, np.zeros((T - 1, B, nx))),
                                                  T, -bound, bound, nx, nu)
        assert status == "Converged"
        np.testing.assert_allclose(x, g[tag + "_x_" + who], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(u, g[tag + "_u_" + who], rtol=2e-5, atol=2e-6)
        res[who] = (x, u, Fm)
    (xt, ut, _), (xp, up, Fm) = res["true"], res["pred"]
    loss = np.mean((ut - up) ** 2) + np.mean((xt - xp) ** 2)
    assert abs(loss - float(g[tag + "_loss"])) < 1e-5
    lo, hi = np.full((T, B, nu), -bound), np.full((T, B, nu), bound)
    out = mpc.mpc_backward(g[tag + "_x_init"], cost.C, cost.c, Fm, np.zeros((T - 1, B, nx)), xp, up, lo, hi,
                           -2.0 * (xt - xp) / xp.size, -2.0 * (ut - up) / up.size, T, nx, nu)
    dAB = out[3].sum(axis=(0, 1))
    np.testing.assert_allclose(dAB[:, :nx], g[tag + "_gA"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dAB[:, nx:], g[tag + "_gB"], rtol=2e-4, atol=2e-6)


def test_pendulum_jacobian_matches_finite_differences():
    from oracle import box_ddp
    rng = np.random.RandomState(0)
    T, B = 4, 5
    th = rng.uniform(-1.5, 1.5, B)
    x0 = np.stack((np.cos(th), np.sin(th), rng.uniform(-1, 1, B)), axis=1)
    u = rng.uniform(-1.5, 1.5, (T, B, 1))
    xs = [x0]
    for t in range(T - 1):
        xs.append(box_ddp.pendulum_step(xs[t], u[t]))
    x = np.stack(xs)
    F, f = box_ddp.pendulum_linearize(x, u)
    eps = 1e-6
    for t in range(T - 1):
        tau = np.concatenate((x[t], u[t]), axis=1)
        np.testing.assert_allclose(np.einsum("bij,bj->bi", F[t], tau) + f[t], x[t + 1], atol=1e-12)
        for j in range(4):
            d = np.zeros(4); d[j] = eps
            plus = box_ddp.pendulum_step(x[t] + d[:3], u[t] + d[3:])
            minus = box_ddp.pendulum_step(x[t] - d[:3], u[t] - d[3:])
            np.testing.assert_allclose(F[t][:, :, j], (plus - minus) / (2 * eps), atol=1e-6)


# ----------------------------------------------------------------- pendulum / box-DDP / imitation (config 2 and 4)
def test_oracle_pendulum_step_and_linearisation_match_the_reference():
    """env_dx/pendulum.py:65-102 (PendulumDx.forward) and mpc/approximate.py:77-119 around it, torques inside,
    exactly ON and beyond the clamp (d clip / du = 1 on the closed interval)"""
    from oracle import box_ddp as obox
    g = load("pendulum.npz")
    np.testing.assert_allclose(obox.pendulum_step(g["x"], g["u"]), g["next"], rtol=0, atol=1e-15)
    F, f = obox.pendulum_linearize(g["lin_x"], g["lin_u"])
    np.testing.assert_allclose(F, g["lin_F"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(f, g["lin_f"], rtol=0, atol=1e-13)
    assert np.allclose(g["lin_F"][:, :2, 2, 3], 0.15, atol=1e-15)      # d newdth / du at u = +-2 exactly: not zero


@pytest.mark.parametrize("name", ["pendulum_boxddp.npz", "pendulum_boxddp_b128.npz"], ids=["B16", "B128_config2"])
def test_oracle_box_ddp_around_the_pendulum_matches_the_reference(name):
    """BoxDDP.forward (mpc/box_ddp.py:93-291) with the non-linear pendulum, 1..4 outer iterations, at B=16 and at config
    2's own batch (B=128, BASELINE.json configs[1])"""
    from oracle import box_ddp as obox
    from oracle import imitation as oim
    g = load(name)
    B, T = int(g["B"]), int(g["T"])
    Q, pv = oim.tile_cost(g["q"], g["p"], T, B)
    for k in (1, 2, 3, 4):
        x, u, c, status, *_ = obox.box_ddp(g["x_init"], mpc.QuadCost(Q, pv), obox.pendulum_step, T, -2.0, 2.0, 3, 1,
                                           eps=1e-3, line_search_decay=0.2, max_line_search_iter=5, max_iter=k,
                                           linearize=obox.pendulum_linearize, batch_coupled=True)
        np.testing.assert_allclose(u, g["u_%d" % k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(x, g["x_%d" % k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(c, g["costs_%d" % k], rtol=0, atol=1e-9)
        assert status.strip() in str(g["stdout_%d" % k])


def test_oracle_imitation_chain_matches_the_reference():
    """config 4, small: Pendulum_Net_cost_logit -> IL_Env.mpc -> loss -> (d logit, d learn_p), incl. the detach mask
    for unconverged samples (env_dx/pendulum_net.py:27-39, il_env.py:104-158, il_exp.py:246-270)"""
    from oracle import imitation as oim
    g = load("imitation_16.npz")
    r = oim.imitation_grads(g["q_logit"], g["learn_p"], g["xinit"], g["expert_u"], int(g["T"]), int(g["lqr_iter"]))
    np.testing.assert_allclose(r["nom_u"], g["nom_u"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(r["nom_x"], g["nom_x"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(r["loss"], float(g["loss"]), rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["g_logit"], g["g_logit"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["g_p"], g["g_p"], rtol=0, atol=1e-12)
    assert np.abs(g["g_logit"]).max() > 1e-4


def test_oracle_imitation_loop_matches_the_reference():
    """config 4's loop: three RMSprop updates of learn_p with the evaluation pass's warm-start carry-over after each
    (env_dx/il_exp.py:213-302, :97-181), recorded from the reference's own pieces (tests/golden/make_golden.py)"""
    from oracle import imitation as oim
    g = load("imitation_loop_16.npz")
    hist = oim.imitation_loop(g["q_logit0"], g["learn_p0"], g["xinit"], g["expert_u"], int(g["T"]), int(g["lqr_iter"]),
                              int(g["K"]), float(g["lr"]), float(g["alpha"]), float(g["eps"]))
    # The first update is exact.  The reference rounds its box-QP solves to float32 (util.py:522-527), which leaves 1e-7
    # between two float64 runs whose parameters differ in the last digits, and ten iLQR iterations (with line-search
    # decisions) carry that to 1e-4 in a quarter of the third pass's controls: measured differences x ~5 as bounds.
    bounds = [dict(loss=1e-12, g=1e-12, learn_p=1e-12, nom_u=1e-12, eval_u=1e-6, eval_loss=1e-8),
              dict(loss=1e-8, g=5e-8, learn_p=1e-9, nom_u=1e-6, eval_u=1e-7, eval_loss=1e-9),
              dict(loss=1e-6, g=2e-6, learn_p=1e-7, nom_u=1e-4, eval_u=1e-3, eval_loss=1e-5)]
    for k, (h, bd) in enumerate(zip(hist, bounds)):
        np.testing.assert_allclose(h["loss"], float(g["loss_%d" % k]), rtol=0, atol=bd["loss"])
        np.testing.assert_allclose(h["g_p"], g["g_p_%d" % k], rtol=0, atol=bd["g"])
        np.testing.assert_allclose(h["g_logit"], g["g_logit_%d" % k], rtol=0, atol=bd["g"])
        np.testing.assert_allclose(h["learn_p"], g["learn_p_%d" % k], rtol=0, atol=bd["learn_p"])
        np.testing.assert_allclose(h["nom_u"], g["nom_u_%d" % k], rtol=0, atol=bd["nom_u"])
        np.testing.assert_allclose(h["eval_u"], g["eval_u_%d" % k], rtol=0, atol=bd["eval_u"])
        np.testing.assert_allclose(h["eval_loss"], float(g["eval_loss_%d" % k]), rtol=0, atol=bd["eval_loss"])


def test_oracle_imitation_step_b1024_matches_the_reference():
    """config 4 at full size (B=1024, T=20): one MPCstep from a common iterate + the gradient node"""
    from oracle import box_ddp as obox
    from oracle import imitation as oim
    g = load("imitation_step_1024.npz")
    pg = load("pendulum.npz")
    B, T, S = int(g["B"]), int(g["T"]), g["sample"]
    np.random.seed(0)
    th = np.random.rand(B) * np.pi - 0.5 * np.pi              # env_dx/il_env.py:55-69
    thdot = np.random.rand(B) * 2.0 - 1.0
    xinit = np.stack((np.cos(th), np.sin(th), thdot), axis=1)
    np.testing.assert_allclose(xinit[:64], pg["xinit1024_head"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(xinit.sum(axis=0), pg["xinit1024_sum"], rtol=0, atol=1e-10)
    xinit = xinit.astype(np.float32).astype(np.float64)
    u_k = g["u_k"].astype(np.float64)
    x_k = obox.get_traj(T, u_k, xinit, obox.pendulum_step)
    Fk, fk = obox.pendulum_linearize(x_k, u_k)
    np.testing.assert_allclose(x_k[:, S], g["x_k_s"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(Fk[:, S], g["F_k_s"], rtol=0, atol=1e-13)
    q, p = oim.cost_from_params(g["logit"], g["learn_p"])
    Q, pv = oim.tile_cost(q, p, T, B)
    lo, hi = np.full((T, B, 1), -2.0), np.full((T, B, 1), 2.0)
    x1, u1, bo, fo, _, _ = mpc.mpc_forward(Q, pv, Fk, fk, u_k, x_k, lo, hi, mpc.QuadCost(Q, pv), obox.pendulum_step,
                                           0.2, 5, T, 3, 1, need_expand=True, batch_coupled=True)
    np.testing.assert_allclose(u1, g["u1"], rtol=0, atol=2e-7)          # stored as float32
    np.testing.assert_allclose(x1[:, S], g["x1_s"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fo.costs, g["costs"], rtol=0, atol=1e-11)
    assert bo.n_total_qp_iter == int(g["n_total_qp_iter"])
    loss, dC, dc = oim.gradient_node(x1, u1, Q, pv, g["expert_u"].astype(np.float64))
    gl, gp = oim.param_grads(dC, dc, g["logit"], g["learn_p"])
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=0, atol=1e-13)
    np.testing.assert_allclose(dC[:, S], g["dC_s"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(dc[:, S], g["dc_s"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(gl, g["g_logit"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(gp, g["g_p"], rtol=0, atol=1e-14)


def test_oracle_imitation_step_1024_early_iterate_matches_the_reference():
    """the same step from the iterate after ONE box-DDP iteration (tests/golden/imitation_step_1024_it1.npz): the oracle
    reproduces the unmodified reference's x', u', costs, QP iteration total"""
    from oracle import box_ddp as obox
    from oracle import imitation as oim
    g = load("imitation_step_1024_it1.npz")
    B, T, S = int(g["B"]), int(g["T"]), g["sample"]
    np.random.seed(0)
    th = np.random.rand(B) * np.pi - 0.5 * np.pi
    thdot = np.random.rand(B) * 2.0 - 1.0
    xinit = np.stack((np.cos(th), np.sin(th), thdot), axis=1).astype(np.float32).astype(np.float64)
    u_k = g["u_k"].astype(np.float64)
    x_k = obox.get_traj(T, u_k, xinit, obox.pendulum_step)
    Fk, fk = obox.pendulum_linearize(x_k, u_k)
    np.testing.assert_allclose(x_k[:, S], g["x_k_s"], rtol=0, atol=1e-13)
    q, p = oim.cost_from_params(g["logit"], g["learn_p"])
    Q, pv = oim.tile_cost(q, p, T, B)
    lo, hi = np.full((T, B, 1), -2.0), np.full((T, B, 1), 2.0)
    x1, u1, bo, fo, _, _ = mpc.mpc_forward(Q, pv, Fk, fk, u_k, x_k, lo, hi, mpc.QuadCost(Q, pv), obox.pendulum_step,
                                           0.2, 5, T, 3, 1, need_expand=True, batch_coupled=True)
    np.testing.assert_allclose(u1, g["u1"], rtol=0, atol=2e-7)          # stored as float32
    np.testing.assert_allclose(x1, g["x1"], rtol=0, atol=2e-6)          # stored as float32 (|x| up to 8)
    np.testing.assert_allclose(fo.costs, g["costs"], rtol=0, atol=1e-10)
    assert bo.n_total_qp_iter == int(g["n_total_qp_iter"])
    assert abs(fo.mean_alphas - float(g["mean_alphas"])) < 1e-12


def test_oracle_pnqp_batch_coupling_fork_matches_the_reference():
    """n=8, B=256: under the batch-global termination (pnqp.py:139-144,172,187) a row ends far from
    its batch-of-one answer and the batch runs into the iteration cap; both modes of the oracle against the reference"""
    g = load("pnqp_n8_b256.npz")
    p = synthetic.make_box_qp(int(g["B"]), int(g["n"]), seed=int(g["seed"]), bound=float(g["bound"]), reg=float(g["reg"]))
    assert abs(checksum(p) - float(g["in_checksum"])) < 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, (LU, piv), idx_f, it = pnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], n_iter=20, batch_coupled=True)
        xr, _, _, itr = pnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], n_iter=20, batch_coupled=False)
    assert it == int(g["it"]) == 19
    np.testing.assert_allclose(x, g["x"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(idx_f, g["idx_f"])
    np.testing.assert_array_equal(piv, g["piv"])
    np.testing.assert_allclose(xr, g["row_x"], rtol=0, atol=1e-6)
    assert np.abs(g["x"] - g["row_x"]).max() > 1.0            # the fork is real


def test_pendulum_step_float32_calibration():
    """Calibration of TOL_STEP_PENDULUM (tests/helpers.py), in the form BASELINE.md section 3 / SURVEY 8d calibrated the LQR
    contract (the reference's own arithmetic run in float32 against float64): ONE MPC step of config 4 (B=1024, T=20;
    mpc/mpc_step.py:288-328 from the common iterate of tests/golden/imitation_step_1024.npz) with the oracle's pendulum
    rollout, linearisation and re-centred cost evaluated in float32 - and, as numpy promotes, the box-QP sweep and the
    line-search rollout still in float64 - against the all-float64 run: of the rows that are not line-search ties ONE
    moves by 9.8e-5, every other by less than 1e-5 (two tie rows move by 7e-5).  It is a MARGINAL row, not rounding growth:
    rounding nothing but the linearisation F_k to float32 (6e-8 relative) already moves it by 1.0e-4 - a box QP whose
    clamped set flips under a 1e-7 perturbation of its data.  So the LQR contract's 1e-4 has no margin for ANY float32 implementation of this step; 2e-4
    has.  The kernels measure 1.3e-4 worst on this batch (profiles/r04/parity_margins.txt)."""
    from oracle import box_ddp as obox
    from oracle import imitation as oim
    from oracle import mpc as ompc
    from tests.helpers import TOL_PRIMAL, TOL_STEP_PENDULUM, tie_rows
    g = np.load(os.path.join(GOLDEN, "imitation_step_1024.npz"))
    B, T = int(g["B"]), int(g["T"])
    np.random.seed(0)
    th = np.random.rand(B) * np.pi - 0.5 * np.pi
    thdot = np.random.rand(B) * 2.0 - 1.0
    x0 = np.stack((np.cos(th), np.sin(th), thdot), axis=1).astype(np.float32).astype(np.float64)
    uk = g["u_k"].astype(np.float64)
    qo, po = oim.cost_from_params(g["logit"], g["learn_p"])
    Qo, pvo = oim.tile_cost(qo, po, T, B)
    lo, hi = np.full((T, B, 1), -2.0), np.full((T, B, 1), 2.0)

    def step(dt, round_inputs_only=False):
        c = (lambda a: a.astype(np.float32).astype(np.float64)) if round_inputs_only else (lambda a: a.astype(dt))
        ukc = uk if round_inputs_only else c(uk)
        xk = obox.get_traj(T, ukc, x0 if round_inputs_only else c(x0), obox.pendulum_step)
        Fk, _ = obox.pendulum_linearize(xk, ukc)
        if round_inputs_only:
            xk, Fk = c(xk), c(Fk)
        Q, pv = c(Qo), c(pvo)
        tau = np.concatenate((xk, ukc), axis=2)
        c_hat = np.einsum("tbij,tbj->tbi", Q, tau) + pv
        Ks, ks, _, _ = ompc.mpc_backward_rec(Q, c_hat, Fk, None, ukc, lo, hi, T, 3, 1, batch_coupled=False)
        x1, u1, _ = ompc.ls_rollout(Ks, ks, ukc, xk, lo, hi, ompc.QuadCost(Q, pv), obox.pendulum_step, np.ones(B), T)
        return x1.astype(np.float64), u1.astype(np.float64)

    x64, u64 = step(np.float64)
    old = ompc.get_cost(T, uk, ompc.QuadCost(Qo, pvo), obox.get_traj(T, uk, x0, obox.pendulum_step))
    strict = ~tie_rows(old, g["costs"])
    err = lambda a, b: (np.abs(a - b) / np.maximum(1.0, np.abs(b)))[:, strict].max(axis=(0, 2))   # noqa: E731
    xr, ur = step(np.float64, round_inputs_only=True)
    er = np.maximum(err(ur, u64), err(xr, x64))
    assert er.max() > 0.5 * TOL_PRIMAL and (er > 1e-5).sum() <= 2      # rounded x_k, F_k, Q, p alone: one marginal row at ~1e-4
    x32, u32 = step(np.float32)
    e = np.maximum(err(u32, u64), err(x32, x64))
    print("config-4 step, float32 rollout + linearisation + re-centring vs float64: worst rows %s" % np.sort(e)[-4:])
    assert e.max() > 0.5 * TOL_PRIMAL          # the LQR contract's 1e-4 has no margin at this step ...
    assert e.max() <= TOL_STEP_PENDULUM        # ... 2e-4 has
    assert (e > 1e-5).sum() <= 8               # and it is a handful of ill-conditioned rows, not the batch
