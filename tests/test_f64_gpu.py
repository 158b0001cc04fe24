"""GPU parity of the float64 entry points (`dmpc_lqr_solve_f64`, `dmpc_lqr_kkt_grad_f64`; SURVEY.md 8b `_f64`): outputs at
the reference's own precision - the oracle (float64 numpy restatement of lqr/lqr_recursion.py:69-209 and
lqr/differentiable_lqr.py:78-142) to 1e-9 relative, and the golden vectors recorded from the reference."""
import glob
import os

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import DiffLqr, LqrRecursion, synthetic
from oracle import kkt as okkt
from oracle import lqr as olqr
from oracle import mpc as ompc
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
TOL64 = 1e-9     # relative to max(1, |reference|): float64 rounding through T Riccati steps, far below any float32 bound


def close64(got, want, what, tol=TOL64):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err <= tol, "%s: %.3e" % (what, err)


def dev64(a):
    return None if a is None else torch.as_tensor(np.asarray(a, dtype=np.float64)).cuda()


@pytest.mark.parametrize("shape", [(5, 6, 3, 1), (9, 7, 8, 2), (3, 5, 6, 3), (2, 4, 20, 6), (70, 5, 4, 2), (2, 3, 32, 8)])
@pytest.mark.parametrize("with_f", [True, False])
def test_solve_and_gradient_at_reference_precision(shape, with_f):
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=5, with_f=with_f)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    x, u = LqrRecursion(dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"]), T, nx, nu,
                        precision="float64").solve_recursion()
    assert x.dtype == torch.float64
    close64(x.cpu().numpy(), xr, "x")
    close64(u.cpu().numpy(), ur, "u")
    rng = np.random.RandomState(3)
    gx, gu = rng.randn(T, B, nx), rng.randn(T, B, nu)
    for strict in (False, True):
        ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
        node = DiffLqr(T, B, nx, nu, strict_math=strict, precision="float64")
        node.forward((dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"])))
        out = node.backward((0, 1, 2, 3, 4), (dev64(gx), dev64(gu)))
        for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
            close64(got.cpu().numpy(), want, key)


def test_clamped_controls_at_reference_precision():
    B, T, nx, nu = 6, 7, 5, 3
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=21, with_f=False)
    act = np.random.RandomState(5).rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(np.zeros((B, nx)), p["C"], p["c"], p["F"], None, act, T, nx, nu)
    x, u = LqrRecursion(dev64(np.zeros((B, nx))), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), None, T, nx, nu,
                        u_zero_Index=torch.as_tensor(act).cuda(), precision="float64").solve_recursion()
    # the clamped system carries 1e-8 on its diagonal (active_constrained_lqr.py:126): the reference multiplies by its
    # explicit inverse (F.batch_inv), which rounds at cond * eps ~ 1e-8; the kernel solves by LU - they agree to that
    close64(x.cpu().numpy(), xr, "x", 1e-6)
    close64(u.cpu().numpy(), ur, "u", 1e-6)
    assert np.all(u.cpu().numpy()[act] == 0)


LQR_FILES = sorted(glob.glob(os.path.join(GOLDEN, "lqr_*.npz")))


@pytest.mark.parametrize("path", LQR_FILES, ids=[os.path.basename(p) for p in LQR_FILES])
def test_reference_golden_vectors_at_reference_precision(path):
    """the vectors recorded from the unmodified reference (float64): solution and gradient to 1e-9"""
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=bool(g["with_f"]))   # the recorded run's inputs
    node = DiffLqr(T, B, nx, nu, precision="float64")
    x, u = node.forward((dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"])))
    close64(x.cpu().numpy(), g["x"], "x")
    close64(u.cpu().numpy(), g["u"], "u")
    out = node.backward((0, 1, 2, 3, 4), (dev64(g["grad_x"]), dev64(g["grad_u"])))
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        close64(got.cpu().numpy(), g[key], key)


def test_headline_batch_every_trajectory_against_the_float64_kernels():
    """BASELINE.json configs[2] at FULL size (B=4096, T=50, (8,2)): the float32 fast path against the float64 kernels
    (themselves held to the oracle above) on EVERY trajectory - solution at the north star's 1e-4, gradient at 5e-4 - not
    on a sample as the oracle tests have to"""
    from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device, kkt_grad_device_f64
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device, solve_device_f64
    from tests.helpers import TOL_COSTATE, TOL_PRIMAL, assert_close
    B, T, nx, nu = 4096, 50, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d32 = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in p.items() if isinstance(v, np.ndarray)}
    d64 = {k: v.double() for k, v in d32.items()}      # identical inputs: the float32 values up-cast
    x, u, _, _ = solve_device(d32["C"], d32["c"], d32["F"], d32["f"], d32["x_init"], None, T, nx, nu)
    x64, u64, _, _ = solve_device_f64(d64["C"], d64["c"], d64["F"], d64["f"], d64["x_init"], None, T, nx, nu)
    assert_close(x.cpu().numpy(), x64.cpu().numpy(), TOL_PRIMAL, "x")
    assert_close(u.cpu().numpy(), u64.cpu().numpy(), TOL_PRIMAL, "u")
    gx, gu = torch.ones((T, B, nx), device="cuda"), torch.ones((T, B, nu), device="cuda")
    out = kkt_grad_device(d32["C"], d32["c"], d32["F"], x, u, gx, gu, T, nx, nu)
    ref = kkt_grad_device_f64(d64["C"], d64["c"], d64["F"], x64, u64, gx.double(), gu.double(), T, nx, nu)
    for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
        assert_close(got.cpu().numpy(), want.cpu().numpy(), TOL_COSTATE if key in ("d_x_init", "dF", "df") else TOL_PRIMAL, key)


def test_config5_shard_sampled_against_the_float64_kernels():
    """(32,8): 512 trajectories of a config-5 shard, float32 matrix-core path against float64"""
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device, solve_device_f64
    from tests.helpers import assert_close
    B, T, nx, nu = 512, 50, 32, 8
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
    d32 = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in p.items() if isinstance(v, np.ndarray)}
    d64 = {k: v.double() for k, v in d32.items()}
    x, u, _, _ = solve_device(d32["C"], d32["c"], d32["F"], d32["f"], d32["x_init"], None, T, nx, nu)
    x64, u64, _, _ = solve_device_f64(d64["C"], d64["c"], d64["F"], d64["f"], d64["x_init"], None, T, nx, nu)
    assert_close(x.cpu().numpy(), x64.cpu().numpy(), 5e-4, "x")
    assert_close(u.cpu().numpy(), u64.cpu().numpy(), 5e-4, "u")
