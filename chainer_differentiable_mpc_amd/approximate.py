"""`linearize_dynamics` / `approximate_cost` - same signatures as mpc/approximate.py:18-119 of the
reference (there built on chainer.grad), here on torch.autograd.  Callers: `BoxDDP` for non-linear
dynamics / non-quadratic costs.  A dynamics object may provide `linearize(x, u) -> (F, f)` (analytic
Jacobian, e.g. `PendulumDx`) which is used instead of autograd."""
import torch

from .util import bmv


def linearize_dynamics(x, u, dynamics):
    """x [T,B,nx], u [T,B,nu] -> F [T-1,B,nx,nx+nu], f [T-1,B,nx] with F_t [x_t;u_t] + f_t = dynamics(x_t,u_t)
    (approximate.py:77-119; like the reference the trajectory is re-rolled from x[0]).  As there, the Jacobians
    R_t, S_t are constants of the graph while f_t = new_x - R_t x_t - S_t u_t stays differentiable through new_x
    (and the re-rolled states), so that MPCstep's `df` reaches learnable dynamics parameters."""
    assert x.shape[0] == u.shape[0] and x.shape[1] == u.shape[1]
    T, n_state = x.shape[0], x.shape[2]
    if hasattr(dynamics, "linearize"):
        return dynamics.linearize(x, u)
    ambient = torch.is_grad_enabled()
    xs = [x[0]]
    Fs, fs = [], []
    for t in range(T - 1):
        xt, ut = xs[t], u[t]
        with torch.enable_grad():
            xd = xt.detach().requires_grad_(True)
            ud = ut.detach().requires_grad_(True)
            nd = dynamics(xd, ud)
            Rt, St = [], []
            for j in range(n_state):
                Rj, Sj = torch.autograd.grad(nd[:, j].sum(), [xd, ud], retain_graph=True)
                Rt.append(Rj)
                St.append(Sj)
        Rt = torch.stack(Rt, dim=1)
        St = torch.stack(St, dim=1)
        new_x = nd.detach()
        if ambient:     # a second evaluation that carries the graph, only when something upstream wants a gradient
            live = dynamics(xt, ut)
            if live.requires_grad:
                new_x = live
        Fs.append(torch.cat((Rt, St), dim=2))
        fs.append(new_x - bmv(Rt, xt) - bmv(St, ut))
        xs.append(new_x)
    return torch.stack(Fs, 0), torch.stack(fs, 0)


def approximate_cost(x, u, Cf):
    """second-order Taylor model of a cost tau -> [B]: (hessians [T,B,ns,ns], grads - H tau [T,B,ns], costs [T,B])
    (approximate.py:18-54).  As in the reference (chainer.grad with enable_double_backprop for the first derivative
    only) the Hessians are constants, while `grads - H tau` and `costs` stay differentiable - MPCstep's `dc` reaches
    the parameters of a learnable non-quadratic cost."""
    assert x.shape[0] == u.shape[0] and x.shape[1] == u.shape[1]
    T = x.shape[0]
    ambient = torch.is_grad_enabled()
    tau = torch.cat((x, u), dim=2)
    costs, hessians, grads = [], [], []
    with torch.enable_grad():
        for t in range(T):
            tau_t = tau[t]
            keep = ambient
            if not (ambient and tau_t.requires_grad):
                # a local leaf to differentiate against; the results stay on the graph only if the cost itself
                # has something that wants a gradient (its parameters)
                keep = ambient and Cf(tau_t).requires_grad
                tau_t = tau_t.detach().requires_grad_(True)
            cost = Cf(tau_t)
            assert list(cost.shape) == [x.shape[1]]
            grad = torch.autograd.grad(cost.sum(), tau_t, create_graph=True)[0]
            hess = []
            for v_i in range(tau.shape[2]):
                if grad.requires_grad:
                    hess.append(torch.autograd.grad(grad[:, v_i].sum(), tau_t, retain_graph=True, allow_unused=True)[0])
                else:
                    hess.append(None)
            hess = [torch.zeros_like(tau_t) if h is None else h for h in hess]   # a cost linear in tau
            hessian = torch.stack(hess, dim=-1).detach()
            g0 = grad - bmv(hessian, tau_t)
            if not keep:
                cost, g0 = cost.detach(), g0.detach()
            costs.append(cost)
            grads.append(g0)
            hessians.append(hessian)
    return torch.stack(hessians), torch.stack(grads), torch.stack(costs)
