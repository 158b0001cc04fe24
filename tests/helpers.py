"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated fp32 parity tolerance (calibrated by running the reference in fp32 vs fp64):
#   |delta| <= 1e-4 * max(1, |ref|) for x, u, dC, dc;  5e-4 * max(1, |ref|) for dx_init, dF, df
TOL_PRIMAL = 1e-4
TOL_COSTATE = 5e-4


def assert_close(got, ref, tol, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s shape %s vs %s" % (what, got.shape, ref.shape)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    worst = float(err.max()) if err.size else 0.0
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    assert worst <= tol, "%s: max |delta|/max(1,|ref|) = %.3e > %.1e" % (what, worst, tol)
    return worst


def to_dev(p, device="cuda"):
    import torch
    return {k: (None if v is None else torch.as_tensor(v, dtype=torch.float32, device=device))
            for k, v in p.items()}


def npy(t):
    return t.detach().cpu().numpy().astype(np.float64)
