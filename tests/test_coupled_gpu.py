"""GPU parity: the reference's batch-global termination of the box QP (mpc/pnqp.py:139-144,172,187) for ANY size and ANY batch -
mpc_coupled.hpp's fixed grid (a batch that is not resident in one launch of the register kernels, more than 8 variables /
controls, more than 64 columns; DMPC_NO_COOP_REGISTER=1 forces it for the others) against the golden vectors recorded from the
reference and against the numpy oracle, beside the register form.  Rows C, E of SURVEY.md section 8; verdict r04 item 8."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import LinDx, MPCstep, PNQP, QuadCost, _lib, synthetic
from oracle import mpc as ompc
from oracle import pnqp as opnqp
from tests.helpers import GOLDEN, TOL_STEP, assert_close, npy

pytestmark = pytest.mark.gpu
MPC_FILES = sorted(glob.glob(os.path.join(GOLDEN, "mpc_*.npz")))
TOL = TOL_STEP      # 1e-4


def dev(a):
    return None if a is None else torch.as_tensor(a, dtype=torch.float32, device="cuda")


@pytest.fixture(params=["register", "fixed_grid"])
def form(request, monkeypatch):
    if request.param == "fixed_grid":
        monkeypatch.setenv("DMPC_NO_COOP_REGISTER", "1")
    else:
        monkeypatch.delenv("DMPC_NO_COOP_REGISTER", raising=False)
    return request.param


def last_kernel():
    return _lib.last_kernel_name()


@pytest.mark.parametrize("n", [1, 2, 4, 8])
@pytest.mark.parametrize("tag", ["cold", "warm"])
def test_batched_golden_on_the_fixed_grid(n, tag, monkeypatch):
    """tests/golden/pnqp_n*.npz, plain keys = the reference called on the whole batch; same assertions as
    test_pnqp_gpu.py::test_batched_golden_with_batch_coupled_termination, which runs the register form"""
    monkeypatch.setenv("DMPC_NO_COOP_REGISTER", "1")
    g = np.load(os.path.join(GOLDEN, "pnqp_n%d.npz" % n))
    B = int(g["B"])
    p = synthetic.make_box_qp(B, n, seed=int(g["seed"]), bound=0.5)
    x0 = None if tag == "cold" else dev(g["warm"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x, fac, idx_f, i = PNQP(dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"]), x_init=x0, n_iter=20, batch_coupled=True)
    assert "pnqp_coupled_kernel" in last_kernel()
    np.testing.assert_array_equal(npy(idx_f), g[tag + "_idx_f"])
    assert_close(npy(x), g[tag + "_x"], 1e-4, "x")
    assert i == int(g[tag + "_it"])                                   # the batch-global iteration index, exactly
    assert bool((PNQP.last_info["iters"] == i).all())
    assert (len(w) > 0) == bool(g[tag + "_warned"])
    if n == 1:
        assert_close(npy(fac), g[tag + "_Hf"], 1e-4, "H_f")
    else:
        np.testing.assert_array_equal(fac[1].cpu().numpy(), g[tag + "_piv"])
        assert_close(npy(fac[0]), g[tag + "_LU"], 1e-4, "LU")


@pytest.mark.parametrize("tiles", [1, 64], ids=["B256", "B16384"])
def test_the_forking_batch_in_both_forms_and_tiled(form, tiles):
    """tests/golden/pnqp_n8_b256.npz (the batch whose rows fork under the batch-global tests), as recorded and tiled 64 times to
    B = 16,384: an OR over a batch and over 64 copies of it are the same decisions, so every copy must reproduce the fixture"""
    g = np.load(os.path.join(GOLDEN, "pnqp_n8_b256.npz"))
    B, n = int(g["B"]), int(g["n"])
    p = synthetic.make_box_qp(B, n, seed=int(g["seed"]), bound=float(g["bound"]), reg=float(g["reg"]))
    rep = lambda a: np.tile(a, (tiles,) + (1,) * (a.ndim - 1))  # noqa: E731
    args = tuple(dev(rep(p[k])) for k in ("H", "q", "lower", "upper"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x, (LU, piv), idx_f, i = PNQP(*args, n_iter=20, batch_coupled=True)
    assert ("pnqp_coupled_kernel" in last_kernel()) == (form == "fixed_grid")
    assert len(w) > 0 and bool(g["warned"]) and i == int(g["it"]) == 19
    forked = rep(np.abs(g["x"] - g["row_x"]).max(axis=1) > 1e-3)
    assert_close(npy(x)[~forked], rep(g["x"])[~forked], 1e-4, "x, rows that do not fork")
    assert_close(npy(x)[forked], rep(g["x"])[forked], 2e-3, "x, forked rows")      # (tolerance as in test_pnqp_gpu.py)
    np.testing.assert_array_equal(npy(idx_f), rep(g["idx_f"]))
    xs = npy(x).reshape(tiles, B, n)
    assert np.array_equal(xs, np.broadcast_to(xs[0], xs.shape))                    # every copy: bit-identical


@pytest.mark.parametrize("n", [9, 12, 40])
@pytest.mark.parametrize("tag", ["cold", "warm"])
def test_more_than_eight_variables_batch_coupled(n, tag):
    """n > 8 has no register form: DMPC_E_UNSUPPORTED until round 5.  Against the oracle's batch-coupled run: solution, free
    set, pivots, LU of the last free-set Hessian, and the batch-global iteration index exactly"""
    B = 48
    p = synthetic.make_box_qp(B, n, seed=300 + n, bound=0.3)
    rng = np.random.default_rng(n)
    x0 = None if tag == "cold" else np.clip(rng.standard_normal((B, n)) * 0.2, p["lower"], p["upper"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        xr, (LUr, pivr), idxr, ir = opnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], x_init=x0, n_iter=20, batch_coupled=True)
        xp, _, _, _ = opnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], x_init=x0, n_iter=20, batch_coupled=False)
        x, (LU, piv), idx_f, i = PNQP(dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"]), x_init=dev(x0), n_iter=20,
                                      batch_coupled=True)
    assert "pnqp_coupled_kernel" in last_kernel()
    assert i == ir
    np.testing.assert_array_equal(npy(idx_f), idxr)
    np.testing.assert_array_equal(piv.cpu().numpy(), pivr)
    assert_close(npy(x), xr, 1e-4, "x")
    assert_close(npy(LU), LUr, 1e-4, "LU")
    print("n=%d %s: batch-global i = %d; coupled vs per-row answers differ by %.2e" % (n, tag, i, np.abs(xr - xp).max()))


def test_a_batch_far_beyond_one_launch_of_the_register_kernel():
    """B = 2^20 rows at n = 2: 4,096 workgroups of the register kernel - more than the device holds at once, refused with
    DMPC_E_UNSUPPORTED until round 5; now the fixed grid takes it.  Against the oracle's batch-coupled run."""
    B, n = 1 << 20, 2
    p = synthetic.make_box_qp(B, n, seed=5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, (LU, piv), idx_f, i = PNQP(dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"]), batch_coupled=True)
        xr, _, idxr, ir = opnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], n_iter=20, batch_coupled=True)
    assert "pnqp_coupled_kernel" in last_kernel()
    assert i == ir
    assert_close(npy(x), xr, 1e-4, "x")
    assert float((npy(idx_f) != idxr).mean()) <= 1e-5      # (a free-set flag is an exact float comparison: ties among 2 M of them)


def make_step(g, p, lo, hi, B, T, nx, nu, **kw):
    return MPCstep(dev(g["u_nom"]), T, dev(hi), dev(lo), B, nx, nu, dev(g["x_nom"]),
                   QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5,
                   need_expand=bool(g["need_expand"]), **kw)


@pytest.mark.parametrize("path", MPC_FILES, ids=[os.path.basename(p) for p in MPC_FILES])
def test_mpc_step_golden_on_the_fixed_grid(path, monkeypatch):
    """tests/golden/mpc_*.npz, plain keys = the reference's MPCstep.forward on the whole batch; same assertions as
    test_mpc_step_gpu.py::test_forward_matches_reference_golden (2), which runs the register form"""
    monkeypatch.setenv("DMPC_NO_COOP_REGISTER", "1")
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    lo = -float(g["bound"]) * np.ones((T, B, nu))
    hi = -lo
    step = make_step(g, p, lo, hi, B, T, nx, nu, batch_coupled=True)
    x, u = step.forward((dev(g["x_nom"][0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert_close(npy(u), g["u"], TOL, "u")
    assert_close(npy(x), g["x"], TOL, "x")
    assert_close(npy(step.for_out.costs), g["costs"], TOL, "costs")
    un = npy(u)
    np.testing.assert_array_equal((un == lo) | (un == hi), g["active"])
    assert step.back_out.n_total_qp_iter == int(g["n_total_qp_iter"])      # sum_t (1 + i_t) with the batch-global i_t
    assert bool((step.n_qp_iter == int(g["n_total_qp_iter"])).all())


def backward_rec_case(B, T, nx, nu, seed, bound):
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed, with_f=True)
    rng = np.random.default_rng(seed)
    u_nom = np.clip(0.3 * rng.standard_normal((T, B, nu)), -bound, bound)
    x_nom = 0.3 * rng.standard_normal((T, B, nx))
    lo = -bound * np.ones((T, B, nu))
    g = dict(u_nom=u_nom, x_nom=x_nom, need_expand=False)
    return p, g, lo, -lo


@pytest.mark.parametrize("shape", [(5, 9), (60, 6)], ids=["9_controls", "67_columns"])
def test_backward_rec_batch_coupled_at_the_tiled_sizes(shape):
    """more than 8 controls / more than 64 columns: per-trajectory termination only until round 5"""
    nx, nu = shape
    B, T = 6, 5
    p, g, lo, hi = backward_rec_case(B, T, nx, nu, 41, 0.15)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Ksr, ksr, bo, Ifree = ompc.mpc_backward_rec(p["C"], p["c"], p["F"], p["f"], g["u_nom"], lo, hi, T, nx, nu, batch_coupled=True)
        step = make_step(g, p, lo, hi, B, T, nx, nu, batch_coupled=True)
        Ks, ks, back_out = step.backward_rec(dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"]))
    assert "mpc_coupled_backward_kernel" in last_kernel()
    assert back_out.n_total_qp_iter == bo.n_total_qp_iter
    assert_close(npy(ks), ksr, TOL, "ks")
    assert_close(npy(Ks), Ksr, TOL, "Ks")
    assert np.all(npy(Ks)[Ifree == 0] == 0)


def test_backward_rec_batch_coupled_beyond_the_resident_batch():
    """(8,2) at B = 8,192: 512 workgroups of the register kernel at one wavefront per SIMD - twice what the device holds; the
    cooperative launch refuses and the fixed grid takes over by itself (no environment switch)"""
    B, T, nx, nu = 8192, 6, 8, 2
    p, g, lo, hi = backward_rec_case(B, T, nx, nu, 43, 0.2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Ksr, ksr, bo, Ifree = ompc.mpc_backward_rec(p["C"], p["c"], p["F"], p["f"], g["u_nom"], lo, hi, T, nx, nu, batch_coupled=True)
        step = make_step(g, p, lo, hi, B, T, nx, nu, batch_coupled=True)
        Ks, ks, back_out = step.backward_rec(dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"]))
    print("kernel:", last_kernel())
    assert back_out.n_total_qp_iter == bo.n_total_qp_iter
    assert_close(npy(ks), ksr, TOL, "ks")
    assert_close(npy(Ks), Ksr, TOL, "Ks")
