"""Oracle (test infrastructure): analytic KKT gradient of the LQR solution.
Restates DiffLqr.backward, lqr/differentiable_lqr.py:78-142, on raw float64 ndarrays.

Quirks of the reference that the default mode reproduces:
  * the second LQR solve uses c := +[grad_x;grad_u] (:110-111), so d_tau is the
    negative of mpc.pytorch's and every output carries the compensating sign;
  * dC_t = 0.5*(d_tau (x) tau) + (tau (x) d_tau)  (:128) - only the first term is
    halved (operator precedence);
  * df = d_lambda[0:T-1]  (:133) - off by one (the true gradient is d_lambda[1:T]).
`strict_math=True` returns 0.5*(d_tau(x)tau + tau(x)d_tau) and d_lambda[1:T] instead.
"""
import numpy as np

from .linalg import bger, bmv
from .lqr import lqr_solve


def difflqr_backward(x_init, C, c, F, x, u, grad_x, grad_u, T, n_state, n_ctrl,
                     strict_math=False):
    """-> (d_x_init [B,nx], dC [T,B,ns,ns], dc [T,B,ns], dF [T-1,B,nx,ns], df [T-1,B,nx])"""
    nx, nu = n_state, n_ctrl
    B = C.shape[1]
    C_Tx = C[T - 1, :, :nx, :]
    c_Tx = c[T - 1, :, :nx]
    taus = np.concatenate((x, u), axis=2)                                   # :91
    lams = [bmv(C_Tx, taus[T - 1]) + c_Tx]                                  # :92
    for i in range(T - 2, -1, -1):                                          # :95-103
        lam_tp1 = lams[T - 2 - i]
        F_tx_T = np.transpose(F[i][:, :nx, :nx], (0, 2, 1))
        lams.append(bmv(F_tx_T, lam_tp1) + bmv(C[i][:, :nx, :], taus[i]) + c[i][:, :nx])
    lams.reverse()
    zero_init = np.zeros_like(x_init)                                       # :108
    zero_f = np.zeros((T - 1, B, nx))                                       # :109
    drl = np.concatenate((grad_x, grad_u), axis=2)                          # :110
    dx, du = lqr_solve(zero_init, C, drl, F, zero_f, T, nx, nu)             # :111-112
    d_taus = np.concatenate((dx, du), axis=2)                               # :114
    d_lams = [bmv(C_Tx, d_taus[T - 1]) + drl[T - 1][:, :nx]]                # :115
    for i in range(T - 2, -1, -1):                                          # :117-125
        d_lam_tp1 = d_lams[T - 2 - i]
        F_tx_T = np.transpose(F[i][:, :nx, :nx], (0, 2, 1))
        d_lams.append(bmv(F_tx_T, d_lam_tp1) + bmv(C[i][:, :nx, :], d_taus[i]) + drl[i][:, :nx])
    d_lams.reverse()
    if strict_math:
        dC = np.stack([0.5 * (bger(d_taus[t], taus[t]) + bger(taus[t], d_taus[t])) for t in range(T)], axis=0)
    else:
        dC = np.stack([0.5 * bger(d_taus[t], taus[t]) + bger(taus[t], d_taus[t]) for t in range(T)], axis=0)  # :128
    dc = np.stack([d_taus[t] for t in range(T)], axis=0)                    # :129
    dF = np.stack([bger(d_lams[t + 1], taus[t]) + bger(lams[t + 1], d_taus[t])
                   for t in range(T - 1)], axis=0)                          # :130-132
    if strict_math:
        df = np.stack(d_lams[1:T], axis=0)
    else:
        df = np.stack(d_lams[:T - 1], axis=0)                               # :133
    d_x_init = d_lams[0]                                                    # :134
    return d_x_init, dC, dc, dF, df
