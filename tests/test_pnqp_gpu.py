"""GPU parity: projected-Newton box QP (mpc/pnqp.py) through the C-ABI.  Row C of SURVEY.md section 8."""
import os
import warnings

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import PNQP, synthetic
from chainer_differentiable_mpc_amd.pnqp import pnqp_device
from oracle import pnqp as opnqp
from tests.helpers import GOLDEN, assert_close, npy

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(a, dtype=torch.float32, device="cuda")


def test_notebook_anchor():
    """experiment_mpc/Projected_Newton_Quadratic_Programming.py:67-68 (the answer pasted from mpc.pytorch)"""
    a = np.load(os.path.join(GOLDEN, "anchors.npz"))
    x, (LU, piv), idx_f, i = PNQP(dev(a["pnqp_H"]), dev(a["pnqp_q"]), dev(a["pnqp_lower"]), dev(a["pnqp_upper"]))
    expect = np.array([[0.1239, -0.0063, 0.0277, -0.6669], [0.6205, 0.2703, 0.4023, -0.0121]])
    np.testing.assert_allclose(npy(x), expect, atol=5e-5)
    np.testing.assert_allclose(npy(x), a["pnqp_x"], atol=5e-6)
    np.testing.assert_array_equal(npy(idx_f), a["pnqp_idx_f"])
    assert list(LU.shape) == [2, 4, 4] and piv.dtype == torch.int32


@pytest.mark.parametrize("n", [1, 2, 4, 8])
@pytest.mark.parametrize("tag", ["cold", "warm"])
def test_per_row_golden(n, tag):
    """the reference called with a batch of one per row (tests/golden/pnqp_n*.npz, *_row_* keys)"""
    g = np.load(os.path.join(GOLDEN, "pnqp_n%d.npz" % n))
    B = int(g["B"])
    p = synthetic.make_box_qp(B, n, seed=int(g["seed"]), bound=0.5)
    x0 = None if tag == "cold" else dev(g["warm"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x, fac, idx_f, i = PNQP(dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"]), x_init=x0, n_iter=20)
    np.testing.assert_array_equal(npy(idx_f), g[tag + "_row_idx_f"])          # identical active sets
    assert_close(npy(x), g[tag + "_row_x"], 1e-4, "x")
    iters = PNQP.last_info["iters"].cpu().numpy()
    assert np.mean(iters == g[tag + "_row_it"]) >= 0.85, (iters, g[tag + "_row_it"])
    assert (len(w) > 0) == bool(g[tag + "_row_warned"].any())
    # box feasibility is exact
    assert (npy(x) >= p["lower"] - 1e-7).all() and (npy(x) <= p["upper"] + 1e-7).all()


@pytest.mark.parametrize("n", [2, 3, 5, 8])
def test_factorisation_outputs_against_oracle(n):
    B = 12
    p = synthetic.make_box_qp(B, n, seed=90 + n, bound=0.4)
    xr, (LUr, pivr), idxr, ir, info = opnqp.pnqp(p["H"], p["q"], p["lower"], p["upper"], batch_coupled=False,
                                                return_info=True, warn=False)
    x, (LU, piv), idx_f, i = PNQP(dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"]))
    np.testing.assert_array_equal(npy(idx_f), idxr)
    np.testing.assert_array_equal(piv.cpu().numpy(), pivr)
    assert_close(npy(LU), LUr, 1e-4, "LU")
    assert_close(npy(x), xr, 1e-4, "x")


def test_kkt_conditions_on_a_large_batch():
    """size-independent property: first-order optimality of the box QP for every row"""
    B, n = 65536, 2
    p = synthetic.make_box_qp(B, n, seed=7, bound=0.5)
    H, q, lo, hi = dev(p["H"]), dev(p["q"]), dev(p["lower"]), dev(p["upper"])
    info = torch.zeros(B, dtype=torch.int32, device="cuda")
    x, fac, piv, idx_f, iters = pnqp_device(H, q, lo, hi, None, 20, info)
    assert int(info.max()) == 0
    g = torch.einsum("bij,bj->bi", H, x) + q
    at_lo, at_hi = x == lo, x == hi
    free = ~(at_lo | at_hi)
    assert float(g[free].abs().max()) <= 2e-3
    assert float(g[at_lo].min()) >= -2e-3 and float(g[at_hi].max()) <= 2e-3
    assert bool(((x >= lo) & (x <= hi)).all())


def test_rejects_inverted_bounds():
    p = synthetic.make_box_qp(2, 2, seed=1)
    with pytest.raises(AssertionError):
        PNQP(dev(p["H"]), dev(p["q"]), dev(p["upper"]), dev(p["lower"]))
