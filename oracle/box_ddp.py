"""Oracle (test infrastructure): the box-DDP outer loop, restating BoxDDP.forward, mpc/box_ddp.py:93-291,
on raw ndarrays over oracle/mpc.py.  For non-linear dynamics the reference linearises with chainer.grad
(mpc/approximate.py:77-119), which the numpy stand-in cannot run; here a `linearize(x, u) -> (F, f)` callable
is passed in (analytic Jacobian in the tests).  Also `pendulum_step` / `pendulum_linearize`: the forward model of
env_dx/pendulum.py:65-102 (simple model) and its Jacobian.

Only the forward computation is restated (the returned x, u, costs and the stop reason); the autograd
plumbing of the final no-op MPCstep node is exercised through oracle.mpc.mpc_backward in the tests.
"""
import numpy as np

from . import mpc as ompc
from .linalg import bmv


def get_traj(T, u, x_init, dynamics):
    """util.py:201-277"""
    xs = [x_init]
    for t in range(T - 1):
        if isinstance(dynamics, ompc.LinDx):
            nx = bmv(dynamics.F[t], np.concatenate((xs[t], u[t]), axis=1))
            if dynamics.f is not None:
                nx = nx + dynamics.f[t]
        else:
            nx = dynamics(xs[t], u[t])
        xs.append(nx)
    return np.stack(xs, axis=0)


def box_ddp(x_init, cost, dynamics, T, u_lower, u_upper, n_state, n_ctrl, u_init=None, eps=1e-5,
            not_improved_lim=5, line_search_decay=0.2, max_line_search_iter=10, best_cost_eps=1e-4, max_iter=10,
            linearize=None, batch_coupled=True):
    """-> (x, u, costs, status, n_iter, last_full_du_norm, best_full_du_norm)"""
    B = x_init.shape[0]
    nx, nu = n_state, n_ctrl
    if np.isscalar(u_lower):
        u_lower = np.full((T, B, nu), float(u_lower))
        u_upper = np.full((T, B, nu), float(u_upper))
    u = np.zeros((T, B, nu)) if u_init is None else np.array(u_init, copy=True)
    best = None
    n_not_improved = 0
    status = None
    for_out = None
    n_iter = 0
    for i in range(max_iter):
        x = get_traj(T, u, x_init, dynamics)                                   # :123
        if isinstance(dynamics, ompc.LinDx):
            Fm, fm = dynamics.F, dynamics.f
        else:
            Fm, fm = linearize(x, u)                                           # :131
        assert isinstance(cost, ompc.QuadCost)
        Cm, cm = cost.C, cost.c
        x, u, back_out, for_out, _, _ = ompc.mpc_forward(
            Cm, cm, Fm, fm, u, x, u_lower, u_upper, cost, dynamics, line_search_decay, max_line_search_iter,
            T, nx, nu, need_expand=True, batch_coupled=batch_coupled)          # :160-172
        n_not_improved += 1
        if best is None:
            best = dict(x=x.copy(), u=u.copy(), costs=for_out.costs.copy(), full_du_norm=for_out.full_du_norm.copy())
        else:
            for j in range(B):                                                 # :200-209
                if for_out.costs[j] <= best["costs"][j] + best_cost_eps:
                    n_not_improved = 0
                    best["x"][:, j] = x[:, j]
                    best["u"][:, j] = u[:, j]
                    best["costs"][j] = for_out.costs[j]
                    best["full_du_norm"][j] = for_out.full_du_norm[j]
        n_iter = i + 1
        if max(for_out.full_du_norm) < eps:                                    # :223-230
            status = "Converged"
            break
        if n_not_improved > not_improved_lim:
            status = "Not improved lim"
            break
        if i == max_iter - 1:
            status = "Not Converged"
    return best["x"], best["u"], best["costs"], status, n_iter, for_out.full_du_norm, best["full_du_norm"]


# ---- pendulum (env_dx/pendulum.py:65-102, simple model: params g, m, l = 10, 1, 1; dt = 0.05; |u| <= 2)
def pendulum_step(x, u, g=10.0, m=1.0, l=1.0, dt=0.05, max_torque=2.0):
    uc = np.clip(u, -max_torque, max_torque)[:, 0]
    cos_th, sin_th, dth = x[:, 0], x[:, 1], x[:, 2]
    th = np.arctan2(sin_th, cos_th)
    newdth = dth + dt * (-3.0 * g / (2.0 * l) * (-sin_th) + 3.0 * uc / (m * l ** 2))
    newth = th + newdth * dt
    return np.stack((np.cos(newth), np.sin(newth), newdth), axis=1)


def pendulum_linearize(x, u, g=10.0, m=1.0, l=1.0, dt=0.05, max_torque=2.0, clamp_grad_closed=True):
    """F_t = d step / d [x;u], f_t = step - F_t [x;u] along the trajectory re-rolled from x[0] (approximate.py:77-119).
    clamp_grad_closed: the derivative of F.clip AT the limits (1 on the closed interval, else 0) - the tests pass the
    product's one flag (PendulumDx.clamp_grad_closed); the fixture recorded through the stand-in's F.clip pins True."""
    T = x.shape[0]
    xs = [x[0]]
    Fs, fs = [], []
    for t in range(T - 1):
        xt, ut = xs[t], u[t]
        c, s, w = xt[:, 0], xt[:, 1], xt[:, 2]
        au = np.abs(ut[:, 0])
        inside = (au <= max_torque if clamp_grad_closed else au < max_torque).astype(xt.dtype)
        r2 = c * c + s * s
        new_x = pendulum_step(xt, ut, g, m, l, dt, max_torque)
        nth = np.arctan2(s, c) + new_x[:, 2] * dt
        one, zero = np.ones_like(c), np.zeros_like(c)
        dnw = np.stack((zero, dt * 3.0 * g / (2.0 * l) * one, one, dt * 3.0 / (m * l ** 2) * inside), axis=1)
        dnth = np.stack((-s / r2, c / r2, zero, zero), axis=1) + dt * dnw
        Ft = np.stack((-np.sin(nth)[:, None] * dnth, np.cos(nth)[:, None] * dnth, dnw), axis=1)
        Fs.append(Ft)
        fs.append(new_x - np.einsum("bij,bj->bi", Ft, np.concatenate((xt, ut), axis=1)))
        xs.append(new_x)
    return np.stack(Fs), np.stack(fs)
