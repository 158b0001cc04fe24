"""Oracle (test infrastructure): batched projected-Newton box QP
    min 1/2 x^T H x + q^T x   s.t.  lower <= x <= upper        (Tassa et al. 2014, Alg. 1)
Restates mpc/pnqp.py:26-201 of the reference on raw ndarrays, quirks included:

  * solves go through `xpbatch_lu_solve` and are therefore rounded to float32 when
    n > 1 (util.py:522-527); n == 1 is a float64 scalar divide (:77-78, :133-134);
  * clamped set uses exact float equality `x == lower` / `x == upper` (:110);
  * convergence (:139-144) and the Armijo loop (:172, :187) reduce over the WHOLE
    batch: the QP returns only when no row has |dx| >= 1e-4, and the backtracking loop
    ends as soon as the batch-max of lhs exceeds GAMMA (rows that already converged
    contribute GAMMA+1e-6, so it then ends after one pass).
`batch_coupled=False` applies the same code to every row on its own (= the reference
called with n_batch=1 per row); that is the semantics of the fused GPU kernels.

Returns the reference's 4-tuple: (x [B,n], H_f [B,1,1] if n==1 else (LU [B,n,n],
piv [B,n] int32), Index_f [B,n] float {0,1}, i).  `return_info=True` appends a dict
(converged flag, iterations) - the reference only warns (:192).
"""
import warnings

import numpy as np

from .linalg import batch_lu_factor, batch_lu_solve, bdot, bger, bmv, bquad, clamp

GAMMA = 0.1          # pnqp.py:23
DECAY = 0.1          # pnqp.py:163
DX_TOL = 1e-4        # pnqp.py:140
REG = 1e-11          # pnqp.py:73
MAX_LS = 10          # pnqp.py:172


def calc_obj(H, q, x):
    """pnqp.py:26-33"""
    return 0.5 * bquad(x, H) + bdot(q, x)


def _pnqp_coupled(H, q, lower, upper, x_init, n_iter, norm_log=None):
    assert (lower <= upper).all(), " lower is larger than upper"
    B, n = q.shape
    assert H.shape == (B, n, n) and lower.shape == (B, n) and upper.shape == (B, n)
    I_pnqp = np.broadcast_to(REG * np.eye(n), (B, n, n))                     # :73
    if x_init is None:
        if n == 1:
            x_init = -(1.0 / np.squeeze(H, axis=2)) * q                      # :78
        else:
            x_init = -batch_lu_solve(batch_lu_factor(H), q)                  # :80-83 (float32)
    else:
        x_init = np.array(x_init, copy=True)                                 # :86
    x = clamp(x_init, lower, upper)                                          # :93
    H_f = H_lu_f = Index_f = None
    i = -1
    for i in range(n_iter):                                                  # :95
        grad = bmv(H, x) + q                                                 # :98
        Index_c = ((x == lower) & (grad > 0.0)) | ((x == upper) & (grad < 0.0))   # :110
        Index_c = 1.0 * Index_c
        Index_f = 1.0 - Index_c
        Index_not_Hff = (1.0 - bger(Index_f, Index_f)).astype(bool)          # :115-116,126
        Index_c = Index_c.astype(bool)
        g_f = np.array(grad, copy=True)
        g_f[Index_c] = 0.0                                                   # :121
        H_f = np.array(H, copy=True)
        H_f[Index_not_Hff] = 0.0                                             # :127
        H_f = H_f + I_pnqp                                                   # :129
        if n == 1:
            dx = -(1.0 / np.squeeze(H_f, axis=2)) * g_f                      # :134
        else:
            H_lu_f = batch_lu_factor(H_f)                                    # :136
            dx = -batch_lu_solve(H_lu_f, g_f)                                # :137 (float32)
        norm = np.sqrt(np.sum(dx ** 2, axis=1))                              # :139
        if norm_log is not None:      # (test infrastructure: how close each pass came to the threshold)
            norm_log.append(np.array(norm, copy=True))
        batch_large = norm >= DX_TOL
        if np.sum(batch_large.astype(float)) == 0:                           # :143
            return x, (H_f if n == 1 else H_lu_f), Index_f, i, True
        alpha = np.ones(B, dtype=x.dtype)                                    # :162
        max_lhs = np.array(GAMMA)
        count = 0
        x_hat = x
        while max_lhs <= GAMMA and count < MAX_LS:                           # :172
            # xp.diagflat(alpha) @ dx  ==  alpha[:,None]*dx  (B x B product, :173)
            x_hat = clamp(x + alpha[:, None] * dx, lower, upper)
            lhs = (GAMMA + 1e-6) * np.ones(B, dtype=x.dtype)                 # :174
            with np.errstate(divide="ignore", invalid="ignore"):
                lhs[batch_large] = (calc_obj(H, q, x) - calc_obj(H, q, x_hat))[batch_large] \
                    / bdot(grad, x - x_hat)[batch_large]                     # :175-176
            I = lhs <= GAMMA
            alpha[I] *= DECAY                                                # :186
            max_lhs = np.max(lhs)                                            # :187
            count += 1
        x = x_hat                                                            # :190
    return x, (H_f if n == 1 else H_lu_f), Index_f, i, False


def pnqp(H, q, lower, upper, x_init=None, n_iter=20, batch_coupled=True,
         return_info=False, warn=True, norm_logs=None):
    """norm_logs (per-row mode only): a list that receives, per row, the |dx| of every pass (the 1e-4 test's operand)"""
    H = np.asarray(H)
    q = np.asarray(q)
    lower = np.asarray(lower)
    upper = np.asarray(upper)
    B, n = q.shape
    if batch_coupled:
        x, fac, idx_f, i, ok = _pnqp_coupled(H, q, lower, upper, x_init, n_iter)
        iters = np.full(B, i, dtype=np.int32)
        conv = np.full(B, ok, dtype=bool)
    else:
        xs, facs_a, facs_p, idxs = [], [], [], []
        iters = np.zeros(B, dtype=np.int32)
        conv = np.zeros(B, dtype=bool)
        for b in range(B):
            xi = None if x_init is None else np.asarray(x_init)[b:b + 1]
            log = [] if norm_logs is not None else None
            xb, fb, ib, it, ok = _pnqp_coupled(H[b:b + 1], q[b:b + 1], lower[b:b + 1],
                                               upper[b:b + 1], xi, n_iter, norm_log=log)
            if norm_logs is not None:
                norm_logs.append(np.array([float(v[0]) for v in log]))
            xs.append(xb)
            idxs.append(ib)
            if n == 1:
                facs_a.append(fb)
            else:
                facs_a.append(fb[0])
                facs_p.append(fb[1])
            iters[b] = it
            conv[b] = ok
        x = np.concatenate(xs, axis=0)
        idx_f = np.concatenate(idxs, axis=0)
        fac = np.concatenate(facs_a, axis=0) if n == 1 else \
            (np.concatenate(facs_a, axis=0), np.concatenate(facs_p, axis=0))
        i = int(iters.max())
    if warn and not conv.all():
        warnings.warn("Projected Newton Quadratic Programming warning: Did not converge")   # :192
    if return_info:
        return x, fac, idx_f, i, {"converged": conv, "iters": iters}
    return x, fac, idx_f, i
