"""`linearize_dynamics` / `approximate_cost` - same signatures as mpc/approximate.py:18-119 of the
reference (there built on chainer.grad), here on torch.autograd.  Callers: `BoxDDP` for non-linear
dynamics / non-quadratic costs.  A dynamics object may provide `linearize(x, u) -> (F, f)` (analytic
Jacobian, e.g. `PendulumDx`) which is used instead of autograd."""
import torch

from .util import bmv


def linearize_dynamics(x, u, dynamics):
    """x [T,B,nx], u [T,B,nu] -> F [T-1,B,nx,nx+nu], f [T-1,B,nx] with F_t [x_t;u_t] + f_t = dynamics(x_t,u_t)
    (approximate.py:77-119; like the reference the trajectory is re-rolled from x[0])"""
    assert x.shape[0] == u.shape[0] and x.shape[1] == u.shape[1]
    T, n_state = x.shape[0], x.shape[2]
    if hasattr(dynamics, "linearize"):
        return dynamics.linearize(x, u)
    xs = [x[0]]
    Fs, fs = [], []
    with torch.enable_grad():
        for t in range(T - 1):
            xt = xs[t].detach().requires_grad_(True)
            ut = u[t].detach().requires_grad_(True)
            new_x = dynamics(xt, ut)
            Rt, St = [], []
            for j in range(n_state):
                Rj, Sj = torch.autograd.grad(new_x[:, j].sum(), [xt, ut], retain_graph=True)
                Rt.append(Rj)
                St.append(Sj)
            Rt = torch.stack(Rt, dim=1)
            St = torch.stack(St, dim=1)
            Fs.append(torch.cat((Rt, St), dim=2))
            fs.append(new_x.detach() - bmv(Rt, xt.detach()) - bmv(St, ut.detach()))
            xs.append(new_x.detach())
    return torch.stack(Fs, 0), torch.stack(fs, 0)


def approximate_cost(x, u, Cf):
    """second-order Taylor model of a cost tau -> [B]: (hessians [T,B,ns,ns], grads - H tau [T,B,ns], costs [T,B])
    (approximate.py:18-54)"""
    assert x.shape[0] == u.shape[0] and x.shape[1] == u.shape[1]
    T = x.shape[0]
    tau = torch.cat((x, u), dim=2)
    costs, hessians, grads = [], [], []
    with torch.enable_grad():
        for t in range(T):
            tau_t = tau[t].detach().requires_grad_(True)
            cost = Cf(tau_t)
            assert list(cost.shape) == [x.shape[1]]
            grad = torch.autograd.grad(cost.sum(), tau_t, create_graph=True)[0]
            hess = []
            for v_i in range(tau.shape[2]):
                hess.append(torch.autograd.grad(grad[:, v_i].sum(), tau_t, retain_graph=True)[0])
            hessian = torch.stack(hess, dim=-1)
            costs.append(cost.detach())
            grads.append(grad.detach() - bmv(hessian, tau_t.detach()))
            hessians.append(hessian)
    return torch.stack(hessians), torch.stack(grads), torch.stack(costs)
