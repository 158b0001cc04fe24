"""Restates the first training iteration of examples/LQRnet.ipynb (cells 4-10) on the oracle.

Known answers recorded in the notebook (examples/LQRnet.ipynb:184): imitation loss 0.661925 at
iteration 0 and dynamics mse 4.774785 after the first RMSprop step.  The second number depends on
dF from DiffLqr.backward (lqr/differentiable_lqr.py:130-132) summed over time and batch
(expand_time_batch's backward) and therefore pins its sign pattern.
"""
import numpy as np

from oracle import kkt, lqr
from oracle.linalg import expand_time_batch


def problem():
    T, nx, nu, B = 5, 3, 1, 128
    ns = nx + nu
    np.random.seed(42)                                   # expert_seed (cell 4)
    p = np.random.randn(ns)                              # cell 5, in dict order: Q, p, A, B
    A_e = np.eye(nx) + 0.2 * np.random.randn(nx, nx)
    B_e = np.random.randn(nx, nu)
    np.random.seed(2)                                    # LqrNet(..., train_seed=2): differentiable_lqr.py:167-172
    A = np.eye(nx) + 0.2 * np.random.randn(nx, nx)
    Bm = np.random.randn(nx, nu)
    x_init = np.random.randn(B, nx)                      # cell 10, first iteration
    C = expand_time_batch(np.eye(ns), T, B)
    c = expand_time_batch(p, T, B)
    return dict(T=T, nx=nx, nu=nu, B=B, A_e=A_e, B_e=B_e, A=A, Bm=Bm, x_init=x_init, C=C, c=c)


def imitation_grads(x_pred, u_pred, x_true, u_true):
    """d/d(x_pred,u_pred) of mean((u_true-u_pred)^2) + mean((x_true-x_pred)^2)."""
    gx = 2.0 * (x_pred - x_true) / x_pred.size
    gu = 2.0 * (u_pred - u_true) / u_pred.size
    return gx, gu


def run_iteration0(solve=None, grad=None):
    """`solve(x_init,C,c,F,f,T,nx,nu)->(x,u)` and `grad(...)->(dx0,dC,dc,dF,df)` default to the oracle;
    the GPU parity test passes the HIP path instead."""
    q = problem()
    T, nx, nu, B = q["T"], q["nx"], q["nu"], q["B"]
    if solve is None:
        solve = lqr.lqr_solve
    if grad is None:
        grad = kkt.difflqr_backward
    F_e = expand_time_batch(np.concatenate((q["A_e"], q["B_e"]), axis=1), T - 1, B)
    F_l = expand_time_batch(np.concatenate((q["A"], q["Bm"]), axis=1), T - 1, B)
    x_true, u_true = solve(q["x_init"], q["C"], q["c"], F_e, None, T, nx, nu)
    x_pred, u_pred = solve(q["x_init"], q["C"], q["c"], F_l, None, T, nx, nu)
    loss0 = np.mean((u_true - u_pred) ** 2) + np.mean((x_true - x_pred) ** 2)
    gx, gu = imitation_grads(x_pred, u_pred, x_true, u_true)
    _, _, _, dF, _ = grad(q["x_init"], q["C"], q["c"], F_l, x_pred, u_pred, gx, gu, T, nx, nu)
    g = np.asarray(dF).sum(axis=(0, 1))                  # expand_time_batch backward
    gA, gB = g[:, :nx], g[:, nx:]
    # chainer.optimizers.RMSprop(lr=1e-2): alpha=0.99, eps=1e-8, ms starts at 0
    lr, alpha, eps = 1e-2, 0.99, 1e-8
    A1 = q["A"] - lr * gA / (np.sqrt((1 - alpha) * gA * gA) + eps)
    B1 = q["Bm"] - lr * gB / (np.sqrt((1 - alpha) * gB * gB) + eps)
    mse1 = np.mean((A1 - q["A_e"]) ** 2) + np.mean((B1 - q["B_e"]) ** 2)
    return float(loss0), float(mse1)
