"""Oracle (test infrastructure): one differentiable box-iLQR step and the active-set LQR
used by its gradient.  Restates on raw ndarrays

  MPCstep.backward_rec   mpc/mpc_step.py:70-173
  MPCstep.forward_rec    mpc/mpc_step.py:175-286
  MPCstep.forward        mpc/mpc_step.py:288-328
  MPCstep.backward       mpc/mpc_step.py:330-460
  LQR_active             mpc/active_constrained_lqr.py:67-202
  xpget_cost             util.py:162-198

Reference quirks reproduced:
  * every nu>1 solve is rounded to float32 (util.py:522-527);
  * K_t rows of clamped controls are exactly 0, V/v are updated from the UNMASKED
    Q blocks (mpc_step.py:147-166, active_constrained_lqr.py:143-145);
  * the line-search loop condition (mpc_step.py:196) is
    `(n_iter < max_ls_iter and cost is None) or (cost > OLD).any()` - max_ls_iter
    only guards the first pass, alpha decays until no sample is worse;
  * full_du_norm / alpha_du_norm (mpc_step.py:261-263, 275-277) reshape a
    [T,nu,B] array to [B, T*nu]: for B > 1 row b is NOT trajectory b's step but a
    scrambled slice.  `du_norm(..., scrambled=True)` reproduces that;
  * PNQP's batch-global termination (see oracle/pnqp.py).  `batch_coupled=False`
    runs every trajectory as its own batch of one.
"""
from collections import namedtuple

import numpy as np

from .linalg import batch_lu_factor, batch_lu_solve, bdot, bger, bmv, bquad, clamp
from .pnqp import pnqp

LqrBackOut = namedtuple("lqrBackOut", "n_total_qp_iter")                     # mpc_step.py:25
LqrForOut = namedtuple("lqrForOut", "objs full_du_norm alpha_du_norm mean_alphas costs")   # :27-30
QuadCost = namedtuple("QuadCost", "C c", defaults=(None, None))              # util.py:25-32
LinDx = namedtuple("LinDx", "F f", defaults=(None, None))


def get_cost(T, u, cost, x):
    """sum_t 1/2 tau^T C tau + c^T tau (QuadCost) or sum_t cost(tau).  util.py:162-198"""
    objs = []
    for t in range(T):
        xut = np.concatenate((x[t], u[t]), axis=1)
        if isinstance(cost, QuadCost):
            objs.append(0.5 * bquad(xut, cost.C[t]) + bdot(xut, cost.c[t]))
        else:
            objs.append(cost(xut))
    return np.sum(np.stack(objs, axis=0), axis=0)


def du_norm(u_old, u_new, scrambled=True):
    """mpc_step.py:261-263 / 275-277."""
    du = u_old - u_new
    T, B, nu = du.shape
    if scrambled:
        du = np.transpose(du, (0, 2, 1)).reshape(B, T * nu)
    else:
        du = np.transpose(du, (1, 0, 2)).reshape(B, T * nu)
    return np.sqrt(np.sum(du ** 2, axis=1))


# --------------------------------------------------------------------------- E2
def mpc_backward_rec(C_hat, c_hat, F_hat, f_hat, controls, u_lower, u_upper, T, n_state,
                     n_ctrl, batch_coupled=True, n_qp_iter=20):
    """mpc_step.py:70-173 -> Ks [T,B,nu,nx], ks [T,B,nu] (float64 storage), LqrBackOut,
    plus per-step free-index masks (diagnostic, not returned by the reference)."""
    nx, nu = n_state, n_ctrl
    B = C_hat.shape[1]
    if F_hat.shape[0] == T:
        F_hat = F_hat[:T - 1]                                                # :83-84
    Ks = np.zeros((T, B, nu, nx))
    ks = np.zeros((T, B, nu))
    Ifree = np.zeros((T, B, nu))
    Vt = vt = None
    prev_kt = None
    n_total_qp_iter = 0
    for t in range(T - 1, -1, -1):
        if t == T - 1:
            Qt, qt = C_hat[t], c_hat[t]
        else:
            Ft = F_hat[t]
            Ft_T = np.transpose(Ft, (0, 2, 1))
            Qt = C_hat[t] + Ft_T @ Vt @ Ft                                   # :110
            if f_hat is None:
                qt = c_hat[t] + bmv(Ft_T, vt)
            else:
                qt = c_hat[t] + bmv(Ft_T @ Vt, f_hat[t]) + bmv(Ft_T, vt)     # :116
        Qt_xx, Qt_xu = Qt[:, :nx, :nx], Qt[:, :nx, nx:]
        Qt_ux, Qt_uu = Qt[:, nx:, :nx], Qt[:, nx:, nx:]
        qt_x, qt_u = qt[:, :nx], qt[:, nx:]
        lower_bound = u_lower[t] - controls[t]                               # :136
        upper_bound = u_upper[t] - controls[t]                               # :138
        kt, Quu_free_LU, Index_free, n_it = pnqp(Qt_uu, qt_u, lower_bound, upper_bound,
                                                 x_init=prev_kt, n_iter=n_qp_iter,
                                                 batch_coupled=batch_coupled, warn=False)   # :141
        n_total_qp_iter += 1 + n_it                                          # :145
        prev_kt = kt
        Qt_ux_copy = np.array(Qt_ux, copy=True)
        mask = np.repeat(np.expand_dims(1.0 - Index_free, axis=2), nx, axis=2).astype(bool)
        Qt_ux_copy[mask] = 0.0                                               # :147-150
        if nu == 1:
            Kt = -((1.0 / Quu_free_LU) * Qt_ux_copy)                         # :154
        else:
            Kt = -batch_lu_solve(Quu_free_LU, Qt_ux_copy)                    # :157 (float32)
        Kt_T = np.transpose(Kt, (0, 2, 1))
        Ks[t] = Kt
        ks[t] = kt
        Ifree[t] = Index_free
        Vt = Qt_xx + Qt_xu @ Kt + Kt_T @ Qt_ux + Kt_T @ Qt_uu @ Kt           # :165 (unmasked)
        vt = qt_x + bmv(Qt_xu, kt) + bmv(Kt_T, qt_u) + bmv(Kt_T @ Qt_uu, kt)  # :166
    return Ks, ks, LqrBackOut(n_total_qp_iter=n_total_qp_iter), Ifree


# --------------------------------------------------------------------------- E3
def ls_rollout(Ks, ks, controls, states, u_lower, u_upper, true_cost, true_dynamics, alphas, T):
    """one pass of the line search, mpc_step.py:198-256: clamped closed-loop rollout with step sizes `alphas` [B]
    under the true dynamics, per-step true cost -> (new_x [T,B,nx], new_u [T,B,nu], objs [T,B])"""
    new_x = [states[0]]
    new_u = []
    dx = [np.zeros_like(states[0])]
    objs = []
    for t in range(T):
        new_xt = new_x[t]
        new_ut = bmv(Ks[t], dx[t]) + controls[t]                             # :209
        new_ut = new_ut + alphas[:, None].astype(ks.dtype) * ks[t]           # :213-219 (diagflat(alpha) @ kt)
        new_ut = clamp(new_ut, u_lower[t], u_upper[t])                       # :221
        new_u.append(new_ut)
        new_xut = np.concatenate((new_xt, new_ut), axis=1)
        if t < T - 1:
            if isinstance(true_dynamics, LinDx):
                new_xtp1 = bmv(true_dynamics.F[t], new_xut)                  # :234
                if true_dynamics.f is not None:
                    new_xtp1 = new_xtp1 + true_dynamics.f[t]
            else:
                new_xtp1 = np.asarray(true_dynamics(new_xt, new_ut))         # :239
            new_x.append(new_xtp1)
            dx.append(new_xtp1 - states[t + 1])                              # :243
        if isinstance(true_cost, QuadCost):
            obj = 0.5 * bquad(new_xut, true_cost.C[t]) + bdot(new_xut, true_cost.c[t])   # :251
        else:
            obj = true_cost(new_xut)
        objs.append(obj)
    return np.stack(new_x, axis=0), np.stack(new_u, axis=0), np.stack(objs, axis=0)


def mpc_forward_rec(Ks, ks, controls, states, u_lower, u_upper, true_cost, true_dynamics,
                    ls_decay, max_ls_iter, T, per_sample=False, max_total_iter=200):
    """mpc_step.py:175-286.  Clamped rollout + line search on the TRUE cost.

    `per_sample=False` is the reference loop (batch-global count; per-sample results
    do not depend on batch-mates).  `max_total_iter` is a safety net only - the
    reference loop is unbounded."""
    B = controls.shape[1]
    alphas = np.ones(B, dtype=controls.dtype)
    OLD_COST = get_cost(T, controls, true_cost, x=states)                    # :191
    current_cost = None
    n_iter = 0
    full_du_norm = None
    while (n_iter < max_ls_iter and current_cost is None) or (current_cost > OLD_COST).any():   # :196
        new_x, new_u, objs = ls_rollout(Ks, ks, controls, states, u_lower, u_upper, true_cost, true_dynamics, alphas, T)
        current_cost = np.sum(objs, axis=0)
        if full_du_norm is None:
            full_du_norm = du_norm(controls, new_u)                          # :260-263
        index_decay = current_cost > OLD_COST
        alphas[index_decay] *= ls_decay                                      # :268
        n_iter += 1
        if n_iter >= max_total_iter:
            break
    alphas[current_cost > OLD_COST] /= ls_decay                              # :274
    alpha_du_norm = du_norm(controls, new_u)                                 # :275-277
    res = LqrForOut(objs, full_du_norm, alpha_du_norm, np.mean(alphas), current_cost)
    return new_x, new_u, res, alphas, n_iter


# --------------------------------------------------------------------------- E4
def mpc_forward(C_hat, c_hat, F_hat, f_hat, controls, current_states, u_lower, u_upper,
                true_cost, true_dynamics, ls_decay, max_ls_iter, T, n_state, n_ctrl,
                need_expand=False, no_op_forward=False, batch_coupled=True):
    """mpc_step.py:288-328 -> (x, u, LqrBackOut, LqrForOut, Ks, ks)."""
    if no_op_forward:
        return current_states, controls, None, None, None, None              # :297-299
    if need_expand:                                                          # :305-317
        c_back = []
        for t in range(T):
            xut = np.concatenate((current_states[t], controls[t]), axis=1)
            c_back.append(bmv(C_hat[t], xut) + c_hat[t])
        c_hat = np.stack(c_back)
        f_hat = None
    Ks, ks, back_out, _ = mpc_backward_rec(C_hat, c_hat, F_hat, f_hat, controls, u_lower, u_upper,
                                           T, n_state, n_ctrl, batch_coupled=batch_coupled)
    x, u, for_out, _, _ = mpc_forward_rec(Ks, ks, controls, current_states, u_lower, u_upper,
                                          true_cost, true_dynamics, ls_decay, max_ls_iter, T)
    return x, u, back_out, for_out, Ks, ks


# --------------------------------------------------------------------------- F1
def lqr_active_backward(C, c, F, f, u_zero_index, T, n_state, n_ctrl):
    """active_constrained_lqr.py:67-151 - Riccati sweep with clamped controls masked."""
    nx, nu = n_state, n_ctrl
    B = C.shape[1]
    Ks, ks = [], []
    Vt = vt = None
    for t in range(T - 1, -1, -1):
        if t == T - 1:
            Qt, qt = C[t], c[t]
        else:
            Ft = F[t]
            Ft_T = np.transpose(Ft, (0, 2, 1))
            Qt = C[t] + Ft_T @ Vt @ Ft
            if f is None:
                qt = c[t] + bmv(Ft_T, vt)
            else:
                qt = c[t] + bmv(Ft_T @ Vt, f[t]) + bmv(Ft_T, vt)
        Qt_xx, Qt_xu = Qt[:, :nx, :nx], Qt[:, :nx, nx:]
        Qt_ux, Qt_uu = Qt[:, nx:, :nx], Qt[:, nx:, nx:]
        qt_x, qt_u = qt[:, :nx], qt[:, nx:]
        index = u_zero_index[t]
        qt_u_ = np.array(qt_u, copy=True)
        qt_u_[index] = 0.0                                                   # :113-114
        Qt_uu_ = np.array(Qt_uu, copy=True)
        notI = 1.0 - index.astype(float)
        Qt_uu_[(1 - bger(notI, notI)).astype(bool)] = 0.0                    # :115-119
        index_qt_uu = np.array([np.diagflat(index[i]) for i in range(B)])    # :121
        Qt_uu_[index_qt_uu] += 1e-8                                          # :122
        Qt_ux_ = np.array(Qt_ux, copy=True)
        Qt_ux_[np.repeat(np.expand_dims(index, axis=2), nx, axis=2)] = 0.0   # :124-126
        if nu == 1:
            Kt = -(1.0 / Qt_uu_) * Qt_ux_                                    # :132
            kt = -(1.0 / np.squeeze(Qt_uu_, axis=2)) * qt_u_
        else:
            LU = batch_lu_factor(Qt_uu_)                                     # :135
            Kt = -batch_lu_solve(LU, Qt_ux_)                                 # float32
            kt = -batch_lu_solve(LU, qt_u_)
        Kt_T = np.transpose(Kt, (0, 2, 1))
        Ks.append(Kt)
        ks.append(kt)
        Vt = Qt_xx + np.matmul(Qt_xu, Kt) + np.matmul(Kt_T, Qt_ux) + np.matmul(np.matmul(Kt_T, Qt_uu), Kt)
        vt = qt_x + bmv(Qt_xu, kt) + bmv(Kt_T, qt_u) + bmv(np.matmul(Kt_T, Qt_uu), kt)
    Ks.reverse()
    ks.reverse()
    return np.stack(Ks), np.stack(ks)


def lqr_active_solve(x_init, C, c, F, f, u_zero_index, T, n_state, n_ctrl):
    """active_constrained_lqr.py:195-202"""
    from .lqr import lqr_forward
    Ks, ks = lqr_active_backward(C, c, F, f, u_zero_index, T, n_state, n_ctrl)
    return lqr_forward(Ks, ks, x_init, F, f, T, n_state, n_ctrl, u_zero_index=u_zero_index)


# --------------------------------------------------------------------------- E5
def mpc_backward(x_init, C_hat, c_hat, F_hat, f_hat, new_x, new_u, u_lower, u_upper,
                 dl_dx, dl_du, T, n_state, n_ctrl):
    """mpc_step.py:330-460 -> (dx_init, dC, dc, dF, df or None)."""
    nx, nu = n_state, n_ctrl
    B = C_hat.shape[1]
    if dl_dx is None:
        dl_dx = np.zeros((T, B, nx))
    if dl_du is None:
        dl_du = np.zeros((T, B, nu))
    d_taus = np.concatenate((dl_dx, dl_du), axis=2)                          # :357
    active = (np.absolute(new_u - u_lower) <= 1e-8) | (np.absolute(new_u - u_upper) <= 1e-8)   # :363
    dx, du = lqr_active_solve(np.zeros_like(x_init), C_hat, -d_taus, F_hat, None, active,
                              T, nx, nu)                                     # :374-376
    dxu = np.concatenate((dx, du), axis=2)
    xu = np.concatenate((new_x, new_u), axis=2)
    dC = np.zeros_like(C_hat)
    for t in range(T):
        dC[t] = -0.5 * (bger(dxu[t], xu[t]) + bger(xu[t], dxu[t]))           # :387
    dc = -dxu                                                                # :390
    lams = np.zeros((T, B, nx))
    prev = None
    for t in range(T - 1, -1, -1):                                           # :395-406
        lamt = bmv(C_hat[t, :, :nx, :nx], new_x[t]) + bmv(C_hat[t, :, :nx, nx:], new_u[t]) + c_hat[t, :, :nx]
        if prev is not None:
            lamt = lamt + bmv(np.transpose(F_hat[t, :, :, :nx], (0, 2, 1)), prev)
        lams[t] = lamt
        prev = lamt
    dlams = np.zeros_like(lams)
    prev = None
    for t in range(T - 1, -1, -1):                                           # :414-425
        dlamt = bmv(C_hat[t, :, :nx, :nx], dx[t]) + bmv(C_hat[t, :, :nx, nx:], du[t]) - d_taus[t, :, :nx]
        if prev is not None:
            dlamt = dlamt + bmv(np.transpose(F_hat[t, :, :, :nx], (0, 2, 1)), prev)
        dlams[t] = dlamt
        prev = dlamt
    dF = np.zeros_like(F_hat)
    for t in range(T - 1):
        dF[t] = -(bger(dlams[t + 1], xu[t]) + bger(lams[t + 1], dxu[t]))     # :434
    df = -dlams[1:] if f_hat is not None else None                           # :437-444
    dx_init = -dlams[0]                                                      # :446
    return dx_init, dC, dc, dF, df
