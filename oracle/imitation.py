"""Oracle (test infrastructure): the imitation step of config 4 - the learnable pendulum cost of
env_dx/pendulum_net.py:12-39 pushed through IL_Env.mpc (env_dx/il_env.py:104-158), the loss of
env_dx/il_exp.py:246-255 and its gradient with respect to (learn_q_logit, learn_p), restated on raw ndarrays over
oracle/box_ddp.py and oracle/mpc.py.  In the reference the gradient is carried by the final no-op MPCstep node of
BoxDDP (mpc/box_ddp.py:234-259) and masked for unconverged samples (:263-289); the chain below follows it by hand."""
import numpy as np

from . import box_ddp as obox
from . import mpc as ompc

MAX_TORQUE, LOWER, UPPER, MPC_EPS, LS_DECAY, MAX_LS_ITER = 2.0, -2.0, 2.0, 1e-3, 0.2, 5   # env_dx/pendulum.py:40-63


def cost_from_params(logit, learn_p):
    """q = sigmoid(logit), p = sqrt(q) * learn_p   (pendulum_net.py:33-34)"""
    q = 1.0 / (1.0 + np.exp(-logit))
    return q, np.sqrt(q) * learn_p


def tile_cost(q, p, T, B):
    """il_env.py:117-129"""
    return np.tile(np.diag(q), (T, B, 1, 1)), np.tile(p, (T, B, 1))


def param_grads(dC, dc, logit, learn_p, keep=None):
    """chain rule back from (dC [T,B,ns,ns], dc [T,B,ns]) to (d logit, d learn_p): the tiling (F.repeat) sums over
    time and batch, util.chainer_diag reads the diagonal, then p = sqrt(q) learn_p and q = sigmoid(logit)"""
    q, _ = cost_from_params(logit, learn_p)
    dQ = dC.sum(axis=(0, 1))
    dpv = dc.sum(axis=(0, 1))
    dq = np.diag(dQ) + dpv * learn_p / (2.0 * np.sqrt(q))
    return dq * q * (1.0 - q), dpv * np.sqrt(q)


def gradient_node(x, u, Q, pv, expert_u, keep=None):
    """loss = mean((expert_u - u)^2) (il_exp.py:254-255) and (dC, dc) from MPCstep.backward at the iterate (x, u)
    with the pendulum linearised there (box_ddp.py:234-259); `keep` [B] in {0,1} is the detach mask of :263-289"""
    T, B = u.shape[0], u.shape[1]
    Fm, fm = obox.pendulum_linearize(x, u)
    dl_du = -2.0 * (expert_u - u) / expert_u.size
    if keep is not None:
        dl_du = dl_du * keep[None, :, None]
    lo, hi = np.full((T, B, 1), LOWER), np.full((T, B, 1), UPPER)
    _, dC, dc, _, _ = ompc.mpc_backward(x[0], Q, pv, Fm, fm, x, u, lo, hi, None, dl_du, T, 3, 1)
    return float(np.mean((expert_u - u) ** 2)), dC, dc


def imitation_grads(logit, learn_p, xinit, expert_u, T, lqr_iter, u_init=None, batch_coupled=True):
    """-> dict(nom_x, nom_u, loss, g_logit, g_p, status): Pendulum_Net_cost_logit.forward + loss.backward()"""
    B = xinit.shape[0]
    q, p = cost_from_params(logit, learn_p)
    Q, pv = tile_cost(q, p, T, B)
    x, u, costs, status, n_iter, last_norm, best_norm = obox.box_ddp(
        xinit, ompc.QuadCost(Q, pv), obox.pendulum_step, T, LOWER, UPPER, 3, 1, u_init=u_init, eps=MPC_EPS,
        line_search_decay=LS_DECAY, max_line_search_iter=MAX_LS_ITER, max_iter=lqr_iter,
        linearize=obox.pendulum_linearize, batch_coupled=batch_coupled)
    keep = (last_norm < MPC_EPS).astype(float) if best_norm.max() > MPC_EPS else None
    loss, dC, dc = gradient_node(x, u, Q, pv, expert_u, keep)
    g_logit, g_p = param_grads(dC, dc, logit, learn_p)
    return dict(nom_x=x, nom_u=u, loss=loss, g_logit=g_logit, g_p=g_p, status=status, dC=dC, dc=dc, keep=keep)


def imitation_loop(logit, learn_p, xinit, expert_u, T, lqr_iter, K, lr=1e-2, alpha=0.5, eps=1e-8):
    """K consecutive updates of config 4's loop (env_dx/il_exp.py:213-302) with the evaluation pass of :97-181 after each:
    the training solve starts cold (the loop passes `train_warm_start`, :248, a buffer it never writes - it fills the
    differently spelled `train_warmstart`, :257); only learn_p moves while `cost_update_q` is False (:227,268-281);
    RMSprop(lr, alpha): ms <- alpha ms + (1 - alpha) g^2, p <- p - lr g / (sqrt(ms) + eps) (chainer.optimizers.RMSprop,
    :213); the evaluation solve is warm-started from the previous pass's prediction (`warmstart[idxs] = pred_u`, :122-124)."""
    B = xinit.shape[0]
    learn_p = np.array(learn_p, dtype=np.float64)
    ms = np.zeros_like(learn_p)
    warm = np.zeros((T, B, 1))
    hist = []
    for _ in range(K):
        r = imitation_grads(logit, learn_p, xinit, expert_u, T, lqr_iter, u_init=None)
        ms = alpha * ms + (1.0 - alpha) * r["g_p"] ** 2
        learn_p = learn_p - lr * r["g_p"] / (np.sqrt(ms) + eps)
        q, p = cost_from_params(logit, learn_p)
        Q, pv = tile_cost(q, p, T, B)
        _, pred_u, *_ = obox.box_ddp(xinit, ompc.QuadCost(Q, pv), obox.pendulum_step, T, LOWER, UPPER, 3, 1, u_init=warm,
                                     eps=MPC_EPS, line_search_decay=LS_DECAY, max_line_search_iter=MAX_LS_ITER,
                                     max_iter=lqr_iter, linearize=obox.pendulum_linearize, batch_coupled=True)
        warm = pred_u
        hist.append(dict(loss=r["loss"], g_logit=r["g_logit"], g_p=r["g_p"], nom_u=r["nom_u"], learn_p=learn_p.copy(),
                         eval_u=pred_u, eval_loss=float(np.mean((expert_u - pred_u) ** 2))))
    return hist
