"""Synthetic LQR / box-QP problem generators (host side, numpy).

The reference publishes no datasets; its own initialisers are `A = I + 0.2*randn`,
`B = randn` (lqr/differentiable_lqr.py:168-172, mpc/mpc_net.py:60-64).  The benchmark
and the parity tests use the normalised generator fixed in SURVEY.md 8(d) /
BASELINE.md section 3 (legacy `numpy.random.RandomState(seed)`, stable across numpy
versions), drawn in this order in float64:

    L ~ N(0,1) [T,B,ns,ns]      C = (L L^T + ns I)/ns        (SPD, Quu well conditioned)
    c ~ N(0,1) [T,B,ns]
    A = I + (0.2/sqrt(nx)) N(0,1) [T-1,B,nx,nx]   Bm ~ N(0,1) [T-1,B,nx,nu]   F = [A|Bm]
    f = 0.1 N(0,1) [T-1,B,nx]   x_init ~ N(0,1) [B,nx]

`fp32_representable=True` rounds every array to float32 and returns it as float64, so
that the GPU path (float32) and the CPU oracle (float64) see identical input values.
"""
import numpy as np


def make_lqr_problem(n_batch, T, n_state, n_ctrl, seed=0, with_f=True,
                     reference_style_A=False, fp32_representable=True, dtype=np.float64):
    nx, nu = n_state, n_ctrl
    ns = nx + nu
    rng = np.random.RandomState(seed)
    L = rng.randn(T, n_batch, ns, ns)
    C = (np.matmul(L, np.transpose(L, (0, 1, 3, 2))) + ns * np.eye(ns)) / ns
    c = rng.randn(T, n_batch, ns)
    scale = 0.2 if reference_style_A else 0.2 / np.sqrt(nx)
    A = np.eye(nx) + scale * rng.randn(T - 1, n_batch, nx, nx)
    Bm = rng.randn(T - 1, n_batch, nx, nu)
    F = np.concatenate((A, Bm), axis=3)
    f = 0.1 * rng.randn(T - 1, n_batch, nx)
    x_init = rng.randn(n_batch, nx)
    out = dict(C=C, c=c, F=F, f=f if with_f else None, x_init=x_init)
    for k, v in out.items():
        if v is None:
            continue
        if fp32_representable:
            v = v.astype(np.float32)
        out[k] = np.ascontiguousarray(v.astype(dtype))
    return out


def make_box_qp(n_batch, n, seed=0, bound=0.5, fp32_representable=True, reg=None):
    """Random strictly convex box QPs with a mix of active and inactive bounds.  H = (L L' + reg I) / n with
    reg = n by default (well conditioned); a small `reg` with wide bounds gives QPs whose projected-Newton steps
    fail the Armijo test - the cases where the reference's batch-global termination makes rows fork."""
    rng = np.random.RandomState(seed)
    L = rng.randn(n_batch, n, n)
    H = (np.matmul(L, np.transpose(L, (0, 2, 1))) + (n if reg is None else reg) * np.eye(n)) / n
    q = 2.0 * rng.randn(n_batch, n)
    lower = -bound * (0.2 + rng.rand(n_batch, n))
    upper = bound * (0.2 + rng.rand(n_batch, n))
    out = dict(H=H, q=q, lower=lower, upper=upper)
    if fp32_representable:
        out = {k: v.astype(np.float32).astype(np.float64) for k, v in out.items()}
    return out


def lqr_algorithmic_bytes_per_timestep(n_state, n_ctrl, itemsize=4):
    """every input read once, every output written once:
    C + c + F + f + x + u = itemsize * (ns^2 + ns + nx*ns + nx + nx + nu)."""
    nx, nu = n_state, n_ctrl
    ns = nx + nu
    return itemsize * (ns * ns + ns + nx * ns + nx + nx + nu)


def kkt_algorithmic_bytes_per_timestep(n_state, n_ctrl, itemsize=4):
    """the KKT backward adds itemsize*(2 ns^2 + 2 nx ns + 3 ns + 2 nx)."""
    nx, nu = n_state, n_ctrl
    ns = nx + nu
    return itemsize * (2 * ns * ns + 2 * nx * ns + 3 * ns + 2 * nx)
