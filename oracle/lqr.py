"""Oracle (test infrastructure): time-varying batched LQR, Riccati backward sweep and
forward rollout.  Restates lqr/lqr_recursion.py:69-209 (class LqrRecursion) of the
reference on raw float64 ndarrays.

Layout (time-major, C-contiguous): C [T,B,ns,ns], c [T,B,ns], F [T-1 or T,B,nx,ns]
(only F[t], t<T-1 is read - lqr_recursion.py:56-64 leaves F unchecked), f [T-1,B,nx]
or None, x_init [B,nx].  ns = nx + nu.
"""
import numpy as np

from .linalg import bmv


def lqr_backward(C, c, F, f, T, n_state, n_ctrl, blocks=None):
    """Riccati value recursion -> gains.  lqr_recursion.py:69-158.

    blocks: an optional dict that receives the control blocks of every step's Q-function, "Quu" [T,B,nu,nu] and
    "Qxu" [T,B,nx,nu] (:100,102) - what the build's saving solve leaves in HBM next to the gains - and the value
    function "V" [T,B,nx,nx], "v" [T,B,nx] of every step (:151-152).

    Returns Ks [T,B,nu,nx], ks [T,B,nu] in forward time order (the reference returns
    Python lists of the same per-step arrays, :156-158).
    The `u_zero_Index` branch (:121-145) is dead code in the reference (no caller
    passes it; MPCstep uses LQR_active) and is not restated.
    """
    nx, nu = n_state, n_ctrl
    B = C.shape[1]
    Ks = np.zeros((T, B, nu, nx), dtype=np.result_type(C, F))
    ks = np.zeros((T, B, nu), dtype=Ks.dtype)
    Vt = vt = None
    for t in range(T - 1, -1, -1):
        if t == T - 1:                                    # :81-83
            Qt, qt = C[t], c[t]
        else:
            Ft = F[t]
            Ft_T = np.transpose(Ft, (0, 2, 1))
            Qt = C[t] + np.matmul(np.matmul(Ft_T, Vt), Ft)          # :89
            if f is None:
                qt = c[t] + bmv(Ft_T, vt)                            # :92
            else:
                qt = c[t] + bmv(np.matmul(Ft_T, Vt), f[t]) + bmv(Ft_T, vt)   # :96
        Qt_xx = Qt[:, :nx, :nx]
        Qt_xu = Qt[:, :nx, nx:]
        Qt_ux = Qt[:, nx:, :nx]
        Qt_uu = Qt[:, nx:, nx:]
        qt_x = qt[:, :nx]
        qt_u = qt[:, nx:]
        if nu == 1:                                       # :112-115
            Kt = -(1.0 / Qt_uu) * Qt_ux
            kt = -(1.0 / np.squeeze(Qt_uu, axis=2)) * qt_u
        else:                                             # :116-120 (F.batch_inv)
            Qt_uu_inv = np.linalg.inv(Qt_uu)
            Kt = -np.matmul(Qt_uu_inv, Qt_ux)
            kt = -bmv(Qt_uu_inv, qt_u)
        Kt_T = np.transpose(Kt, (0, 2, 1))
        Ks[t] = Kt
        ks[t] = kt
        if blocks is not None:
            blocks.setdefault("Quu", np.zeros((T, B, nu, nu), dtype=Ks.dtype))[t] = Qt_uu
            blocks.setdefault("Qxu", np.zeros((T, B, nx, nu), dtype=Ks.dtype))[t] = Qt_xu
        # :151-152 - no symmetrisation, every term kept
        Vt = Qt_xx + np.matmul(Qt_xu, Kt) + np.matmul(Kt_T, Qt_ux) + np.matmul(np.matmul(Kt_T, Qt_uu), Kt)
        vt = qt_x + bmv(Qt_xu, kt) + bmv(Kt_T, qt_u) + bmv(np.matmul(Kt_T, Qt_uu), kt)
        if blocks is not None:
            blocks.setdefault("V", np.zeros((T, B, nx, nx), dtype=Ks.dtype))[t] = Vt
            blocks.setdefault("v", np.zeros((T, B, nx), dtype=Ks.dtype))[t] = vt
    return Ks, ks


def lqr_forward(Ks, ks, x_init, F, f, T, n_state, n_ctrl, u_zero_index=None):
    """Closed-loop rollout.  lqr_recursion.py:160-200.

    u_t = K_t x_t + k_t (zeroed where u_zero_index[t], :179-183);
    x_{t+1} = F_t [x_t;u_t] (+ f_t), t < T-1.
    """
    xs = [x_init]
    us = []
    for t in range(T):
        xt = xs[t]
        ut = bmv(Ks[t], xt) + ks[t]
        if u_zero_index is not None:
            ut = np.where(u_zero_index[t], np.zeros_like(ut), ut)
        us.append(ut)
        if t < T - 1:
            xu = np.concatenate((xt, ut), axis=1)
            x = bmv(F[t], xu)
            if f is not None:
                x = x + f[t]
            xs.append(x)
    return np.stack(xs, axis=0), np.stack(us, axis=0)


def lqr_solve(x_init, C, c, F, f, T, n_state, n_ctrl):
    """backward() then forward().  lqr_recursion.py:202-209.  One *timestep-solve* of
    the headline metric = one t of one trajectory through both sweeps."""
    Ks, ks = lqr_backward(C, c, F, f, T, n_state, n_ctrl)
    return lqr_forward(Ks, ks, x_init, F, f, T, n_state, n_ctrl)
