"""GPU parity: MPCstep forward (backward_rec + forward_rec) and backward, LQR_active - through the C-ABI,
against golden vectors recorded from the reference and against the numpy oracle.  Rows E, F."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import LQR_active, LinDx, MPCstep, QuadCost, synthetic
from oracle import mpc as ompc
from tests.helpers import GOLDEN, TOL_COSTATE, TOL_PRIMAL, TOL_STEP, assert_close, npy

pytestmark = pytest.mark.gpu

MPC_FILES = sorted(glob.glob(os.path.join(GOLDEN, "mpc_*.npz")))
TOL = TOL_STEP  # 1e-4: the contract of BASELINE.md section 3 (measured worst case on these goldens: 2e-6, profiles/r04/parity_margins.txt)


def dev(a):
    return None if a is None else torch.as_tensor(a, dtype=torch.float32, device="cuda")


def setup(g):
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    bound = float(g["bound"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    lo = -bound * np.ones((T, B, nu))
    hi = bound * np.ones((T, B, nu))
    return B, T, nx, nu, p, lo, hi


def make_step(g, p, lo, hi, B, T, nx, nu, **kw):
    return MPCstep(dev(g["u_nom"]), T, dev(hi), dev(lo), B, nx, nu, dev(g["x_nom"]),
                   QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5,
                   need_expand=bool(g["need_expand"]), **kw)


@pytest.mark.parametrize("path", MPC_FILES, ids=[os.path.basename(p) for p in MPC_FILES])
def test_forward_matches_reference_golden(path):
    g = np.load(path)
    B, T, nx, nu, p, lo, hi = setup(g)
    step = make_step(g, p, lo, hi, B, T, nx, nu)
    x, u = step.forward((dev(g["x_nom"][0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    # (1) the reference run one trajectory at a time: the semantics of the fused kernels
    assert_close(npy(u), g["row_u"], TOL, "u vs per-trajectory reference")
    assert_close(npy(x), g["row_x"], TOL, "x vs per-trajectory reference")
    assert_close(npy(step.for_out.costs), g["row_costs"], TOL, "costs vs per-trajectory reference")
    np.testing.assert_array_equal(step.n_qp_iter.cpu().numpy(), g["row_n_qp"])
    un = npy(u)
    np.testing.assert_array_equal((un == lo) | (un == hi),                      # the reference's test, mpc_step.py:363-364
                                  (np.abs(g["row_u"] - lo) <= 1e-8) | (np.abs(g["row_u"] - hi) <= 1e-8))
    # (2) the reference run on the whole batch (batch-global PNQP termination, pnqp.py:139-144,172,187):
    # batch_coupled=True, same plain tolerance
    step = make_step(g, p, lo, hi, B, T, nx, nu, batch_coupled=True)
    x, u = step.forward((dev(g["x_nom"][0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert_close(npy(u), g["u"], TOL, "u")
    assert_close(npy(x), g["x"], TOL, "x")
    fo = step.for_out
    assert_close(npy(fo.costs), g["costs"], TOL, "costs")
    assert_close(npy(fo.objs), g["objs"], TOL, "objs")
    assert_close(npy(fo.full_du_norm), g["full_du_norm"], TOL, "full_du_norm")       # scrambled reshape quirk
    assert_close(npy(fo.alpha_du_norm), g["alpha_du_norm"], TOL, "alpha_du_norm")
    assert abs(fo.mean_alphas - float(g["mean_alphas"])) <= 1e-6
    # the same controls sit on their bounds - exactly on them
    un = npy(u)
    active = (un == lo) | (un == hi)
    np.testing.assert_array_equal(active, g["active"])
    assert step.back_out.n_total_qp_iter == int(g["n_total_qp_iter"])      # sum_t (1 + i_t) with the batch-global i_t
    assert bool((step.n_qp_iter == int(g["n_total_qp_iter"])).all())


@pytest.mark.parametrize("path", MPC_FILES, ids=[os.path.basename(p) for p in MPC_FILES])
def test_backward_matches_reference_golden(path):
    g = np.load(path)
    B, T, nx, nu, p, lo, hi = setup(g)
    step = make_step(g, p, lo, hi, B, T, nx, nu)
    step.forward((dev(g["x_nom"][0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    out = step.backward((0, 1, 2, 3, 4), (dev(g["grad_x"]), dev(g["grad_u"])))
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        assert_close(npy(got), g["row_" + key], TOL_PRIMAL if key in ("dC", "dc") else TOL_COSTATE, key + " vs per-trajectory reference")
    step = make_step(g, p, lo, hi, B, T, nx, nu, batch_coupled=True)
    step.forward((dev(g["x_nom"][0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    out = step.backward((0, 1, 2, 3, 4), (dev(g["grad_x"]), dev(g["grad_u"])))
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        assert_close(npy(got), g[key], TOL_PRIMAL if key in ("dC", "dc") else TOL_COSTATE, key + " vs the batched reference run")


@pytest.mark.parametrize("path", MPC_FILES[::2], ids=[os.path.basename(p) for p in MPC_FILES[::2]])
@pytest.mark.parametrize("coupled", [False, True], ids=["per_row", "batch_coupled"])
def test_backward_rec_and_forward_rec_against_oracle(path, coupled):
    g = np.load(path)
    B, T, nx, nu, p, lo, hi = setup(g)
    c_hat, f_hat = p["c"], p["f"]
    if bool(g["need_expand"]):
        tau = np.concatenate((g["x_nom"], g["u_nom"]), axis=2)
        c_hat = np.einsum("tbij,tbj->tbi", p["C"], tau) + p["c"]
        f_hat = None
    Ksr, ksr, bo, Ifree = ompc.mpc_backward_rec(p["C"], c_hat, p["F"], f_hat, g["u_nom"], lo, hi, T, nx, nu,
                                                batch_coupled=coupled)
    step = make_step(g, p, lo, hi, B, T, nx, nu, batch_coupled=coupled)
    Ks, ks, back_out = step.backward_rec(dev(p["C"]), dev(c_hat), dev(p["F"]), dev(f_hat))
    assert_close(npy(ks), ksr, TOL, "ks")
    assert_close(npy(Ks), Ksr, TOL, "Ks")
    if coupled:
        assert back_out.n_total_qp_iter == bo.n_total_qp_iter
    clamped = Ifree == 0
    assert np.all(npy(Ks)[clamped] == 0)             # gain rows of clamped controls are exactly zero
    xr, ur, fo, alphas, n_it = ompc.mpc_forward_rec(Ksr, ksr, g["u_nom"], g["x_nom"], lo, hi, ompc.QuadCost(p["C"], p["c"]),
                                                    ompc.LinDx(p["F"], p["f"]), 0.2, 5, T)
    x, u, for_out = step.forward_rec(Ks, ks, step.true_cost, step.true_dynamics, 0.2, 5)
    assert_close(npy(x), xr, TOL, "x")
    assert_close(npy(u), ur, TOL, "u")
    assert_close(npy(step.alphas), alphas, 1e-6, "alphas")
    assert_close(npy(for_out.costs), fo.costs, TOL, "costs")


@pytest.mark.parametrize("path", MPC_FILES[1::3], ids=[os.path.basename(p) for p in MPC_FILES[1::3]])
def test_lqr_active_class_golden(path):
    g = np.load(path)
    B, T, nx, nu, p, lo, hi = setup(g)
    la = LQR_active(torch.zeros(B, nx).cuda(), dev(p["C"]), -dev(np.concatenate((g["grad_x"], g["grad_u"]), axis=2)),
                    dev(p["F"]), None, T, nx, nu, u_zero_Index=torch.as_tensor(g["active"]).cuda())
    dx, du = la.solve_recursion()
    assert_close(npy(dx), g["active_dx"], TOL_PRIMAL, "dx")
    assert_close(npy(du), g["active_du"], TOL_PRIMAL, "du")


def test_callable_dynamics_and_cost_line_search():
    """non-linear true dynamics + callable cost: gains from the kernel, rollout in torch (mpc_step.py:237-253)"""
    B, T, nx, nu = 6, 7, 3, 1
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=41)
    rng = np.random.RandomState(42)
    u_nom = np.clip(0.3 * rng.randn(T, B, nu), -0.5, 0.5).astype(np.float32).astype(np.float64)
    lo, hi = -0.5 * np.ones((T, B, nu)), 0.5 * np.ones((T, B, nu))

    def dyn_np(x, u):
        return np.stack((x[:, 0] + 0.1 * np.sin(x[:, 1]), x[:, 1] + 0.1 * x[:, 2], 0.9 * x[:, 2] + 0.2 * u[:, 0]), axis=1)

    def dyn_t(x, u):
        return torch.stack((x[:, 0] + 0.1 * torch.sin(x[:, 1]), x[:, 1] + 0.1 * x[:, 2], 0.9 * x[:, 2] + 0.2 * u[:, 0]), dim=1)

    def cost_np(tau):
        return 0.5 * (tau ** 2).sum(axis=1) + 0.1 * tau[:, 0]

    def cost_t(tau):
        return 0.5 * (tau ** 2).sum(dim=1) + 0.1 * tau[:, 0]

    xs = [p["x_init"]]
    for t in range(T - 1):
        xs.append(dyn_np(xs[t], u_nom[t]))
    x_nom = np.stack(xs)
    Ksr, ksr, _, _ = ompc.mpc_backward_rec(p["C"], p["c"], p["F"], p["f"], u_nom, lo, hi, T, nx, nu, batch_coupled=False)
    xr, ur, fo, alphas, _ = ompc.mpc_forward_rec(Ksr, ksr, u_nom, x_nom, lo, hi, cost_np, dyn_np, 0.2, 5, T)
    step = MPCstep(dev(u_nom), T, dev(hi), dev(lo), B, nx, nu, dev(x_nom), cost_t, dyn_t, ls_decay=0.2, max_ls_iter=5)
    x, u = step.forward((dev(x_nom[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert_close(npy(u), ur, TOL, "u")
    assert_close(npy(x), xr, TOL, "x")
    assert_close(npy(step.for_out.costs), fo.costs, TOL, "costs")


def test_autograd_through_mpc_step_and_no_op_forward():
    g = np.load(MPC_FILES[2])
    B, T, nx, nu, p, lo, hi = setup(g)
    C = dev(p["C"]).requires_grad_(True)
    c = dev(p["c"]).requires_grad_(True)
    step = make_step(g, p, lo, hi, B, T, nx, nu)
    x, u = step.apply((dev(g["x_nom"][0]), C, c, dev(p["F"]), dev(p["f"])))
    (x * dev(g["grad_x"])).sum().add((u * dev(g["grad_u"])).sum()).backward()
    assert_close(npy(C.grad), g["dC"], TOL_PRIMAL, "dC")
    assert_close(npy(c.grad), g["dc"], TOL_PRIMAL, "dc")
    # no_op_forward returns the iterate it was given and still differentiates (box_ddp.py:247-259)
    xd, ud = x.detach(), u.detach()
    noop = MPCstep(ud, T, dev(hi), dev(lo), B, nx, nu, xd, step.true_cost, step.true_dynamics, 0.2, 5,
                   need_expand=True, no_op_forward=True)
    x2, u2 = noop.forward((xd[0], dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert torch.equal(x2, xd) and torch.equal(u2, ud)
    out = noop.backward((0, 1, 2, 3, 4), (dev(g["grad_x"]), dev(g["grad_u"])))
    assert_close(npy(out[1]), g["dC"], TOL_PRIMAL, "dC via no-op node")


def test_headline_shape_properties():
    """B=4096, T=50, nx=8, nu=2 with bounds: feasibility, descent, dynamics consistency"""
    B, T, nx, nu = 4096, 50, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d = {k: dev(v) for k, v in p.items()}
    u0 = torch.zeros((T, B, nu), device="cuda")
    xs = [d["x_init"]]
    for t in range(T - 1):
        xs.append(torch.einsum("bij,bj->bi", d["F"][t], torch.cat((xs[t], u0[t]), dim=1)) + d["f"][t])
    x0 = torch.stack(xs)
    lo = torch.full((T, B, nu), -0.5, device="cuda")
    hi = torch.full((T, B, nu), 0.5, device="cuda")
    step = MPCstep(u0, T, hi, lo, B, nx, nu, x0, QuadCost(d["C"], d["c"]), LinDx(d["F"], d["f"]), 0.2, 5, need_expand=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u = step.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    assert bool(((u >= lo) & (u <= hi)).all())
    tau = torch.cat((x, u), dim=2)
    nxt = torch.einsum("tbij,tbj->tbi", d["F"], tau[:-1]) + d["f"]
    assert float((nxt - x[1:]).abs().max()) <= 1e-3 * max(1.0, float(x.abs().max()))
    old = 0.5 * torch.einsum("tbi,tbij,tbj->b", torch.cat((x0, u0), 2), d["C"], torch.cat((x0, u0), 2)) + \
        (torch.cat((x0, u0), 2) * d["c"]).sum(dim=(0, 2))
    assert bool((step.for_out.costs <= old * (1 + 1e-5) + 1e-3).all())        # the line search never accepts a worse cost
    assert float((u == lo).float().mean() + (u == hi).float().mean()) > 0.05  # the box is active somewhere


@pytest.mark.parametrize("coupled", [False, True], ids=["per_row", "batch_coupled"])
@pytest.mark.parametrize("shape", [(6, 7, 5, 2, 0.25), (5, 6, 4, 3, 0.375), (4, 8, 7, 1, 0.5), (3, 5, 10, 5, 0.25),
                                   (3, 6, 32, 8, 0.25), (4, 6, 20, 6, 0.375), (5, 6, 3, 3, 0.25), (19, 5, 6, 3, 0.375),
                                   (4, 6, 13, 2, 0.25), (5, 5, 9, 4, 0.5), (3, 6, 13, 1, 0.25), (6, 5, 2, 4, 0.375),
                                   (5, 7, 16, 8, 0.25), (9, 5, 32, 8, 0.375), (6, 5, 5, 5, 0.25), (5, 6, 16, 4, 0.375), (3, 5, 31, 7, 0.25),
                                   (4, 5, 12, 4, 0.5), (16, 9, 12, 4, 0.25), (36, 7, 16, 4, 0.5),
                                   # padded inside the wide row kernels' instances (nx <= 16, nu <= 8, nx + nu >= 16, B % 4 == 0)
                                   (8, 6, 13, 3, 0.25), (4, 5, 10, 6, 0.375), (12, 6, 15, 7, 0.25), (8, 5, 14, 2, 0.5), (4, 6, 16, 3, 0.25),
                                   (8, 5, 9, 8, 0.25),
                                   # ... and below 16 columns with more than 4 controls (no 16-lane MPC container holds them)
                                   (8, 6, 5, 5, 0.25), (4, 5, 9, 6, 0.375), (12, 5, 3, 8, 0.25), (4, 6, 1, 5, 0.5),
                                   # ... and three / four controls from 12 elements of tau on (whole wavefronts; a container otherwise)
                                   (8, 5, 9, 4, 0.25), (4, 6, 11, 4, 0.375), (8, 5, 10, 3, 0.5), (4, 7, 9, 3, 0.25)])   # bounds exact in float32
def test_shapes_without_a_specialisation_against_the_oracle(shape, coupled):
    """shapes outside the register-resident list - padded inside a container kernel (nu <= 4, nx + nu <= 15; (3,3) is the
    shape of the reference's experiment_mpc/MpcNet.py:43-44), on the matrix-core sweep with the box QP inside ((16,8), (32,8),
    per-trajectory termination) or on the runtime-dimension kernels of mpc_generic.hpp:
    forward (backward_rec with PNQP + line search) and the analytic backward against the oracle, both termination modes"""
    B, T, nx, nu, bound = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=7, with_f=True)
    lo, hi = -bound * np.ones((T, B, nu)), bound * np.ones((T, B, nu))
    u0 = np.zeros((T, B, nu))
    xs = [p["x_init"]]
    for t in range(T - 1):
        xs.append(np.einsum("bij,bj->bi", p["F"][t], np.concatenate((xs[t], u0[t]), axis=1)) + p["f"][t])
    x0 = np.stack(xs).astype(np.float32).astype(np.float64)
    xr, ur, bo, fo, Ksr, ksr = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u0, x0, lo, hi,
                                                ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]), 0.2, 5,
                                                T, nx, nu, need_expand=True, batch_coupled=coupled)
    step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
                   LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=True, batch_coupled=coupled)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u = step.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    if coupled:
        assert step.back_out.n_total_qp_iter == bo.n_total_qp_iter
    assert_close(npy(step.ks), ksr, TOL, "ks")
    assert_close(npy(step.Ks), Ksr, TOL, "Ks")
    assert_close(npy(u), ur, TOL, "u")
    assert_close(npy(x), xr, TOL, "x")
    assert_close(npy(step.for_out.costs), fo.costs, TOL, "costs")
    active = (npy(u) == lo) | (npy(u) == hi)
    np.testing.assert_array_equal(active, (np.abs(ur - lo) <= 1e-8) | (np.abs(ur - hi) <= 1e-8))
    assert active.any()
    gx, gu = np.ones((T, B, nx)), np.ones((T, B, nu))
    ref = ompc.mpc_backward(x0[0], p["C"], p["c"], p["F"], None, xr, ur, lo, hi, gx, gu, T, nx, nu)
    out = step.backward((0, 1, 2, 3, 4), (dev(gx), dev(gu)))
    for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
        if want is None:
            continue
        assert_close(npy(got), want, TOL_PRIMAL if key in ("dC", "dc") else TOL_COSTATE, key)


@pytest.mark.parametrize("shape", [(4, 6, 10, 12, 0.25), (3, 5, 40, 12, 0.375), (3, 5, 64, 16, 0.25), (3, 5, 70, 3, 0.5)],
                         ids=lambda s: "%dx%d" % (s[2], s[3]))
def test_shapes_beyond_8_controls_or_64_columns_against_the_oracle(shape):
    """VERDICT r03 "any shape": `MPCstep` with more than 8 controls, or more than 64 columns, used to be refused; the
    reference has no limit (mpc/mpc_step.py:70-286, mpc/pnqp.py:37-201).  The tiled kernels (mpc_tiled.hpp: a workgroup
    per trajectory, the box QP with runtime dimensions) take them: forward (backward_rec with PNQP + line search) and the
    analytic backward against the oracle at the contract's tolerances, per-trajectory termination; and (round 5) the
    batch-coupled mode on mpc_coupled.hpp's fixed grid."""
    from chainer_differentiable_mpc_amd import _lib
    B, T, nx, nu, bound = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=7, with_f=True)
    lo, hi = -bound * np.ones((T, B, nu)), bound * np.ones((T, B, nu))
    u0 = np.zeros((T, B, nu))
    xs = [p["x_init"]]
    for t in range(T - 1):
        xs.append(np.einsum("bij,bj->bi", p["F"][t], np.concatenate((xs[t], u0[t]), axis=1)) + p["f"][t])
    x0 = np.stack(xs).astype(np.float32).astype(np.float64)
    xr, ur, bo, fo, Ksr, ksr = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u0, x0, lo, hi,
                                                ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]), 0.2, 5,
                                                T, nx, nu, need_expand=True, batch_coupled=False)
    step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
                   LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u = step.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert "tiled" in _lib.last_kernel_name()
    assert_close(npy(step.ks), ksr, TOL, "ks")
    assert_close(npy(step.Ks), Ksr, TOL, "Ks")
    assert_close(npy(u), ur, TOL, "u")
    assert_close(npy(x), xr, TOL, "x")
    assert_close(npy(step.for_out.costs), fo.costs, TOL, "costs")
    active = (npy(u) == lo) | (npy(u) == hi)
    np.testing.assert_array_equal(active, (np.abs(ur - lo) <= 1e-8) | (np.abs(ur - hi) <= 1e-8))
    assert active.any()
    # the separately callable halves (mpc_step.py:70-173, :175-286)
    tau = np.concatenate((x0, u0), axis=2)
    c_hat = np.einsum("tbij,tbj->tbi", p["C"], tau) + p["c"]
    Ks, ks, _ = step.backward_rec(dev(p["C"]), dev(c_hat), dev(p["F"]), None)
    assert_close(npy(Ks), Ksr, TOL, "Ks (backward_rec)")
    assert_close(npy(ks), ksr, TOL, "ks (backward_rec)")
    x2, u2, _ = step.forward_rec(Ks, ks, step.true_cost, step.true_dynamics, 0.2, 5)
    assert_close(npy(u2), ur, TOL, "u (forward_rec)")
    gx, gu = np.ones((T, B, nx)), np.ones((T, B, nu))
    ref = ompc.mpc_backward(x0[0], p["C"], p["c"], p["F"], None, xr, ur, lo, hi, gx, gu, T, nx, nu)
    out = step.backward((0, 1, 2, 3, 4), (dev(gx), dev(gu)))
    for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
        if want is None:
            continue
        assert_close(npy(got), want, TOL_PRIMAL if key in ("dC", "dc") else TOL_COSTATE, key)
    coupled = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
                      LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=True, batch_coupled=True)
    # batch-coupled at these sizes: refused until round 5, now mpc_coupled.hpp's fixed grid - against the oracle's coupled run
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xc, uc = coupled.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
        xcr, ucr, boc, foc, _, _ = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u0, x0, lo, hi, ompc.QuadCost(p["C"], p["c"]),
                                                   ompc.LinDx(p["F"], p["f"]), 0.2, 5, T, nx, nu, need_expand=True, batch_coupled=True)
    assert_close(npy(uc), ucr, TOL, "u (batch-coupled)")
    assert_close(npy(xc), xcr, TOL, "x (batch-coupled)")
    assert coupled.back_out.n_total_qp_iter == boc.n_total_qp_iter


def test_tiled_mpc_step_without_recentring_and_with_an_affine_model():
    """the any-size MPC kernels with `need_expand=False` (the Taylor model is used as given: c_hat and a non-zero f_hat enter
    the sweep, mpc_step.py:110-116) at 9 controls - one more than the register-resident QP holds"""
    B, T, nx, nu, bound = 5, 6, 7, 9, 0.25
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=17, with_f=True)
    lo, hi = -bound * np.ones((T, B, nu)), bound * np.ones((T, B, nu))
    u0 = np.clip(0.2 * np.random.RandomState(1).randn(T, B, nu), -bound, bound).astype(np.float32).astype(np.float64)
    xs = [p["x_init"]]
    for t in range(T - 1):
        xs.append(np.einsum("bij,bj->bi", p["F"][t], np.concatenate((xs[t], u0[t]), axis=1)) + p["f"][t])
    x0 = np.stack(xs).astype(np.float32).astype(np.float64)
    xr, ur, bo, fo, Ksr, ksr = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u0, x0, lo, hi,
                                                ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]), 0.2, 5,
                                                T, nx, nu, need_expand=False, batch_coupled=False)
    step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
                   LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u = step.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert_close(npy(step.ks), ksr, TOL, "ks")
    assert_close(npy(step.Ks), Ksr, TOL, "Ks")
    assert_close(npy(u), ur, TOL, "u")
    assert_close(npy(x), xr, TOL, "x")
    assert_close(npy(step.for_out.costs), fo.costs, TOL, "costs")
    assert int(step.back_out.n_total_qp_iter) >= T


def test_batch_coupled_takes_a_batch_that_is_not_resident_in_one_launch():
    """rounds 1-4 refused this with DMPC_E_UNSUPPORTED (the register kernels' grid-wide termination needs every workgroup
    resident); mpc_coupled.hpp's fixed grid takes any batch - parity: tests/test_coupled_gpu.py"""
    from chainer_differentiable_mpc_amd import PNQP, _lib
    B, n = 1 << 19, 2                       # 2048 workgroups of 256 QPs: more than the register kernel's launch holds
    p = synthetic.make_box_qp(B, n, seed=5)
    args = (dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, _, _, _ = PNQP(*args)
        assert bool(torch.isfinite(x).all())
        xc, _, _, i = PNQP(*args, batch_coupled=True)
    assert bool(torch.isfinite(xc).all()) and bool((PNQP.last_info["iters"] == i).all())


def test_box_ddp_device_loop_batch_coupled_matches_the_reference_trace():
    """`dmpc_box_ddp(batch_coupled=1)`: the reference's BoxDDP run (tests/golden/boxddp_trace.npz) is a batched run,
    i.e. with batch-global PNQP termination inside every step"""
    from chainer_differentiable_mpc_amd import BoxDDP
    g = np.load(os.path.join(GOLDEN, "boxddp_trace.npz"))
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    for device_loop in (True, False):
        solver = BoxDDP(T, -float(g["bound"]), float(g["bound"]), B, nx, nu, None, max_iter=10, quiet=True,
                        batch_coupled=True, device_loop=device_loop)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, costs = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
        assert solver.status in str(g["stdout"])
        assert_close(npy(u), g["u"], TOL_PRIMAL, "u")
        assert_close(npy(x), g["x"], TOL_PRIMAL, "x")
        assert_close(npy(costs), g["costs"], TOL_PRIMAL, "costs")
