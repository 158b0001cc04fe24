"""Oracle (test infrastructure): batched small linear algebra of the reference.

Restates util.py of pfnet-research/chainer-differentiable-mpc on raw ndarrays:
  bmv/xpbmv      util.py:280-298      bger/xpbger    util.py:301-329
  bquad/xpbquad  util.py:332-358      bdot/xpbdot    util.py:411-434
  clamp/xpclamp  util.py:101-123      expand_*       util.py:361-408
  xpbatch_lu_factor util.py:462-482 (torch.lu  == LAPACK getrf, 1-based int32 pivots)
  xpbatch_lu_solve  util.py:505-528 (casts LU and b to float32, LAPACK getrs)

torch (pinned "*" in the reference Pipfile:17, tested with 1.2.0, README.md:26) is a
third-party dependency of the reference; its two call sites are restated here from
the published LAPACK algorithms (getf2: first-max partial pivoting, scale by the
reciprocal pivot; getrs: row interchanges, unit-lower forward substitution, upper
back substitution), so the oracle needs numpy only.
"""
import numpy as np


def bmv(a, x):
    """[B,n,m] @ [B,m] -> [B,n]  (util.py:280-298)"""
    assert a.shape[0] == x.shape[0], "batch mismatch"
    assert a.shape[2] == x.shape[1], "mat mul dim mismatch"
    assert x.ndim == 2, " x is not batch vector"
    return np.squeeze(np.matmul(a, np.expand_dims(x, 2)), axis=2)


def bger(x, y):
    """batched outer product [B,n],[B,m] -> [B,n,m]  (util.py:301-329)"""
    return np.expand_dims(x, 2) @ np.expand_dims(y, 1)


def bquad(x, Q):
    """x^T Q x per batch row  (util.py:332-358)"""
    assert x.shape[0] == Q.shape[0] and x.shape[1] == Q.shape[1] == Q.shape[2]
    xT = np.expand_dims(x, 1)
    x_ = np.expand_dims(x, 2)
    return np.squeeze(np.squeeze(xT @ Q @ x_, axis=1), axis=1)


def bdot(x, y):
    """per-row dot product  (util.py:411-434)"""
    assert x.shape == y.shape
    return np.squeeze(np.squeeze(np.expand_dims(x, 1) @ np.expand_dims(y, 2), axis=1), axis=1)


def clamp(x, lower, upper):
    """min(max(x, lower), upper); asserts lower<=upper  (util.py:117-123)"""
    assert x.shape == lower.shape == upper.shape
    assert (lower <= upper).all(), " lower is larger than upper"
    return np.minimum(np.maximum(x, lower), upper)


def expand_time_batch(m, time, n_batch):
    """tile [...] -> [time, n_batch, ...]  (util.py:361-377)"""
    m = np.asarray(m)
    return np.broadcast_to(m, (time, n_batch) + m.shape).copy()


def expand_batch(m, n_batch):
    """tile [...] -> [n_batch, ...]  (util.py:380-408)"""
    m = np.asarray(m)
    return np.broadcast_to(m, (n_batch,) + m.shape).copy()


def batch_lu_factor(A):
    """LAPACK getrf semantics, batched.  (util.py:462-482: torch.lu(torch.tensor(A)))

    Returns (LU [B,n,n] in A's dtype, pivots [B,n] int32, 1-based).  A zero pivot is
    left in place (LAPACK info>0) - the later solve then produces inf/nan, as
    LAPACK/torch do.
    """
    assert A.ndim == 3 and A.shape[1] == A.shape[2], "Actual" + str(A.shape)
    LU = np.array(A, copy=True)
    B, n, _ = LU.shape
    piv = np.zeros((B, n), dtype=np.int32)
    rows = np.arange(B)
    tiny = np.finfo(LU.dtype).tiny
    for k in range(n):
        p = k + np.argmax(np.abs(LU[:, k:, k]), axis=1)      # first max (idamax)
        piv[:, k] = p + 1
        rk = LU[rows, k, :].copy()
        rp = LU[rows, p, :].copy()
        LU[rows, k, :] = rp
        LU[rows, p, :] = rk
        d = LU[:, k, k]
        if k + 1 < n:
            nz = d != 0
            big = np.abs(d) >= tiny
            with np.errstate(divide="ignore", invalid="ignore"):
                scaled = np.where(big[:, None], LU[:, k + 1:, k] * (1.0 / d)[:, None],
                                  LU[:, k + 1:, k] / d[:, None])
            LU[:, k + 1:, k] = np.where(nz[:, None], scaled, LU[:, k + 1:, k])
            LU[:, k + 1:, k + 1:] -= LU[:, k + 1:, k, None] * LU[:, None, k, k + 1:]
    return LU, piv


def batch_lu_solve(lu_and_piv, b):
    """LAPACK getrs semantics in **float32**  (util.py:505-528).

    The reference casts `b` and `LU` to float32 before `torch.lu_solve`, so every
    projected-Newton / MPC-step solve is rounded to float32.  `b` may be [B,n]
    (pnqp.py:83,137; active_constrained_lqr.py:137 - torch<=1.2 `btrisolve`
    semantics) or [B,n,k] (mpc_step.py:157, active_constrained_lqr.py:136).
    Returns float32.
    """
    LU, piv = lu_and_piv
    LU = np.asarray(LU).astype(np.float32)
    x = np.array(b, copy=True).astype(np.float32)
    vec = x.ndim == 2
    if vec:
        x = x[:, :, None]
    B, n, _ = LU.shape
    rows = np.arange(B)
    for k in range(n):                       # laswp
        p = piv[:, k].astype(np.int64) - 1
        xk = x[rows, k, :].copy()
        xp_ = x[rows, p, :].copy()
        x[rows, k, :] = xp_
        x[rows, p, :] = xk
    for k in range(n):                       # L y = Pb (unit lower, column oriented)
        if k + 1 < n:
            x[:, k + 1:, :] -= LU[:, k + 1:, k, None] * x[:, None, k, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        for k in range(n - 1, -1, -1):       # U x = y (column oriented)
            x[:, k, :] = x[:, k, :] / LU[:, k, k, None]
            if k > 0:
                x[:, :k, :] -= LU[:, :k, k, None] * x[:, None, k, :]
    return x[:, :, 0] if vec else x
