"""`MPCstep` - one differentiable box-constrained iLQR step with the constructor, methods and
attributes of mpc/mpc_step.py:33-460 of the reference (chainer FunctionNode -> torch.autograd).

  forward   = Taylor re-centring (need_expand) + backward_rec (Riccati sweep with one projected-Newton
              box QP per timestep) + forward_rec (clamped rollout and line search on the TRUE cost)
  backward  = active-set LQR on -[dl_dx;dl_du] + co-state sweeps + outer products

All of it runs in hand-written HIP kernels when the true cost is a `QuadCost` and the true dynamics
a `LinDx`.  A callable true cost / dynamics (e.g. the pendulum) keeps backward_rec on the GPU kernel
and rolls the line search out with torch ops around the callable.

Differences from the reference, by design:
  * by default PNQP terminates per trajectory (the reference's batch-global tests make a trajectory's result
    depend on its batch-mates; per trajectory is what shards across GPUs).  `batch_coupled=True` gives the
    reference's batch semantics (pnqp.py:139-144,172,187) for a batch that fits one cooperative launch.  The line
    search's batch-global loop condition (mpc_step.py:196) never changes a trajectory's result, so it stays per
    trajectory in both modes;
  * `LqrBackOut.n_total_qp_iter` is the largest per-trajectory total (per-trajectory values in
    `self.n_qp_iter`; with `batch_coupled` they are all equal to the reference's number);
  * `full_du_norm` / `alpha_du_norm` reproduce the reference's reshape of a [T,nu,B] array to
    [B, T*nu] (mpc_step.py:261-263) unless `strict_math=True` (then true per-trajectory norms).
"""
from collections import namedtuple

import numpy as np
import torch

from . import _lib
from .lqr_recursion import _as_tensor, _device_of, _workspace, raise_info
from .util import LinDx, QuadCost, bdot, bmv, bquad, clamp, get_cost

LqrBackOut = namedtuple("lqrBackOut", "n_total_qp_iter")
LqrForOut = namedtuple("lqrForOut", "objs full_du_norm alpha_du_norm mean_alphas costs")


class _LazyForOut:
    """`LqrForOut` of the one-launch `MPCstep.forward` (mpc_step.py:175-286) with the two norms formed when first asked for:
    full_du_norm / alpha_du_norm are a handful of small launches each that most callers (BoxDDP's device loop has its own)
    never read.  Same fields, order and tuple behaviour as the namedtuple."""
    _fields = LqrForOut._fields
    __slots__ = ("objs", "mean_alphas", "costs", "_make_full", "_make_alpha", "_full", "_alpha")

    def __init__(self, objs, make_full, make_alpha, mean_alphas, costs):
        self.objs, self.mean_alphas, self.costs = objs, mean_alphas, costs
        self._make_full, self._make_alpha = make_full, make_alpha
        self._full = self._alpha = None

    @property
    def full_du_norm(self):
        if self._full is None:
            self._full = self._make_full()
        return self._full

    @property
    def alpha_du_norm(self):
        if self._alpha is None:
            self._alpha = self._make_alpha()
        return self._alpha

    def _astuple(self):
        return LqrForOut(self.objs, self.full_du_norm, self.alpha_du_norm, self.mean_alphas, self.costs)

    def __iter__(self):
        return iter(self._astuple())

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return self._astuple()[i]

    def _asdict(self):
        return self._astuple()._asdict()

    def __repr__(self):
        return repr(self._astuple())


def _is_simple_pendulum(dyn):
    """PendulumDx with the 3-parameter model (the one the imitation experiments use, env_dx/pendulum.py:40-63)"""
    from .pendulum import PendulumDx
    return isinstance(dyn, PendulumDx) and dyn.simple


def du_norm(u_old, u_new, scrambled=True):
    du = u_old - u_new
    T, B, nu = du.shape
    if scrambled:   # mpc_step.py:261-263: transpose to [T,nu,B] then reshape to [B, T*nu]
        du = du.permute(0, 2, 1).reshape(B, T * nu)
    else:
        du = du.permute(1, 0, 2).reshape(B, T * nu)
    return torch.sqrt((du ** 2).sum(dim=1))


class _MPCstepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_init, C, c, F, f, node):
        x, u = node._forward_impl(x_init, C, c, F, f)
        ctx.node = node
        ctx.retained = node._retained       # per-call state (a node object may be applied more than once)
        ctx.in_meta = [(t.dtype, t.device, tuple(t.shape)) if t is not None else None for t in (x_init, C, c, F, f)]
        return x, u

    @staticmethod
    def backward(ctx, dl_dx, dl_du):
        grads = ctx.node.backward((0, 1, 2, 3, 4), (dl_dx, dl_du), retained=ctx.retained)
        out = []
        for g, meta in zip(grads, ctx.in_meta):
            if meta is None or g is None:
                out.append(None)
                continue
            dtype, device, shape = meta
            g = g.to(device=device, dtype=dtype)
            if tuple(g.shape) != shape:
                pad = torch.zeros(shape, dtype=dtype, device=device)
                pad[: g.shape[0]] = g
                g = pad
            out.append(g)
        return tuple(out) + (None,)


def tiled_cost_gradient(T, B, nx, nu, dev, r, lo, hi, gx, gu, detach=None):
    """`MPCstep.backward` (mpc_step.py:330-460) for a cost that is one (Q, p) tiled over time and batch: -> (dx_init [B,nx],
    sum_{t,b} dC [ns,ns], sum_{t,b} dc [ns]), the sums formed inside the co-state kernel; dF, df are not formed.
    r: retained float32 device tensors C, c, F, x, u; detach = (du_norm_last [B], flag int32[1], eps) gates the incoming
    gradient per trajectory on the device (BoxDDP's detach mask).  None where the C entry point does not serve the size."""
    lib = _lib.load()
    ns = nx + nu
    out = torch.empty((B * nx + ns * ns + ns,), dtype=torch.float32, device=dev)      # one allocation
    dx0, dQ, dp = out[:B * nx].view(B, nx), out[B * nx:B * nx + ns * ns].view(ns, ns), out[B * nx + ns * ns:]
    need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
    ws = _workspace(need, dev)
    dn, dfl, de = (None, None, 0.0) if detach is None else (detach[0], detach[1], float(detach[2]))
    with _lib.guard(dev):
        rc = lib.dmpc_mpc_step_backward(T, B, nx, nu, _lib.ptr(r["C"]), _lib.ptr(r["c"]), _lib.ptr(r["F"]), _lib.ptr(r["x"]),
                                        _lib.ptr(r["u"]), _lib.ptr(lo), _lib.ptr(hi), _lib.ptr(gx), _lib.ptr(gu),
                                        _lib.ptr(dx0), None, None, None, None, _lib.ptr(dQ), _lib.ptr(dp), _lib.ptr(dn),
                                        _lib.ptr(dfl), de, _lib.ptr(ws), need, None, _lib.stream_ptr(dev))
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "dmpc_mpc_step_backward")
    return dx0, dQ, dp


class _MPCstepTiledFn(torch.autograd.Function):
    """The gradient node of `BoxDDP` for a `TiledQuadCost` (no-op forward, mpc/box_ddp.py:247-259): inputs are the
    un-tiled (Q [ns,ns], p [ns]); backward returns the gradient summed over time and batch, as the backward of the tiling
    would (`tiled_cost_gradient`); the linear model is a constant of this node (update_dynamics=False, :252-258).
    `spec` = (T, B, nx, nu, device, lower, upper, detach)."""

    @staticmethod
    def forward(ctx, x_init, Q, p, spec, retained, x_best, u_best):
        ctx.spec, ctx.retained = spec, retained
        ctx.meta = (x_init.dtype, x_init.device, Q.dtype, Q.device)
        ctx.need_x0 = x_init.requires_grad
        return x_best.detach(), u_best.detach()

    @staticmethod
    def backward(ctx, dl_dx, dl_du):
        T, B, nx, nu, dev, lo, hi, detach = ctx.spec
        gx = None if dl_dx is None else _lib.f32c(dl_dx, dev)
        gu = None if dl_du is None else _lib.f32c(dl_du, dev)
        got = tiled_cost_gradient(T, B, nx, nu, dev, ctx.retained, lo, hi, gx, gu, detach)
        if got is None:       # a ragged batch / a wide shape: the dense gradient, reduced here
            node = MPCstep(controls=ctx.retained["u"], T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=nx, n_ctrl=nu,
                           current_states=ctx.retained["x"], true_cost=None, true_dynamics=None, ls_decay=0.2, max_ls_iter=1,
                           no_op_forward=True)
            if detach is not None:
                keep = ((detach[1] == 0) | (detach[0] < detach[2])).to(torch.float32)[None, :, None]
                gx = None if gx is None else gx * keep
                gu = None if gu is None else gu * keep
            full = node.backward((0, 1, 2), (gx, gu), retained=dict(ctx.retained, f=None))
            got = (full[0], full[1].sum(dim=(0, 1)), full[2].sum(dim=(0, 1)))
        dx0, dQ, dp = got
        xd, xdev, qd, qdev = ctx.meta
        return (dx0.to(device=xdev, dtype=xd) if ctx.need_x0 else None, dQ.to(device=qdev, dtype=qd),
                dp.to(device=qdev, dtype=qd), None, None, None, None)


class MPCstep:
    """MPC forward backward calculation (mpc/mpc_step.py:33)."""

    def __init__(self, controls, T, u_upper, u_lower, n_batch, n_state, n_ctrl, current_states,
                 true_cost, true_dynamics, ls_decay, max_ls_iter, verbose=False, need_expand=False,
                 no_op_forward=False, strict_math=False, n_qp_iter=20, batch_coupled=False):
        self.controls = _as_tensor(controls)
        self.u_upper = _as_tensor(u_upper)
        self.u_lower = _as_tensor(u_lower)
        self.n_state, self.n_ctrl = int(n_state), int(n_ctrl)
        self.n_sc = self.n_state + self.n_ctrl
        self.n_batch, self.T = int(n_batch), int(T)
        self.verbose = verbose
        self.back_out = None
        self.for_out = None
        self.current_states = _as_tensor(current_states)
        self.true_cost = true_cost
        self.true_dynamics = true_dynamics
        self.need_expand = need_expand
        self.ls_decay = float(ls_decay)
        self.max_ls_iter = int(max_ls_iter)
        self.no_op_forward = no_op_forward
        self.strict_math = strict_math
        self.n_qp_iter_max = int(n_qp_iter)
        self.batch_coupled = bool(batch_coupled)
        self.n_qp_iter = None     # per-trajectory sum_t (1 + i_t)
        self.n_ls_iter = None     # per-trajectory line-search passes
        self.alphas = None
        self.info = None
        self._retained = None
        self._true_model = None   # (the true cost / dynamics tensors as the kernels take them, cached by identity)
        self._ws_need = None
        self._dev = _device_of(self.controls, self.current_states)
        self._out_dtype = self.controls.dtype if self.controls.dtype.is_floating_point else torch.float32
        self._out_device = self.controls.device
        d = self._dev
        self._u = _lib.f32c(self.controls.detach(), d)
        self._xs = _lib.f32c(self.current_states.detach(), d)
        self._lo = _lib.f32c(self.u_lower, d)
        self._hi = _lib.f32c(self.u_upper, d)
        T_, B_, nu_, nx_ = self.T, self.n_batch, self.n_ctrl, self.n_state
        assert list(self._u.shape) == [T_, B_, nu_] and list(self._xs.shape) == [T_, B_, nx_]
        assert list(self._lo.shape) == [T_, B_, nu_] and list(self._hi.shape) == [T_, B_, nu_]

    def _out(self, t):
        return t.to(device=self._out_device, dtype=self._out_dtype)

    def _fused_ok(self):
        return isinstance(self.true_cost, QuadCost) and isinstance(self.true_dynamics, LinDx)

    # ------------------------------------------------------------------ E2
    def backward_rec(self, C_hat, c_hat, F_hat, f_hat):
        """-> (Ks [T,B,nu,nx], ks [T,B,nu], LqrBackOut)   mpc_step.py:70-173"""
        lib = _lib.load()
        _lib.require_gpu()
        T, B, nx, nu, ns = self.T, self.n_batch, self.n_state, self.n_ctrl, self.n_sc
        d = self._dev
        C, c, F, f = (_lib.f32c(_as_tensor(t), d) for t in (C_hat, c_hat, F_hat, f_hat))
        assert list(C.shape) == [T, B, ns, ns], "C hat dim mismatch"
        assert list(c.shape) == [T, B, ns], "c hat dim mismatch"
        assert F.shape[0] in (T - 1, T) and list(F.shape[1:]) == [B, nx, ns], "F_hat dim mismatch"
        assert not bool(torch.isnan(self._u).any())
        assert bool((self._lo <= self._hi).all()), " lower is larger than upper"
        Ks = torch.empty((T, B, nu, nx), dtype=torch.float32, device=d)
        ks = torch.empty((T, B, nu), dtype=torch.float32, device=d)
        nqp = torch.empty((B,), dtype=torch.int32, device=d)
        info = torch.zeros(B, dtype=torch.int32, device=d)
        ws = None
        need = lib.dmpc_mpc_backward_rec_workspace_bytes(T, B, nx, nu, self.n_qp_iter_max, 1 if self.batch_coupled else 0)
        if need:       # batch-coupled decision slots, or the matrices of the tiled kernel (more than 8 controls / 64 columns)
            ws = _workspace(need, d)
        with _lib.guard(d):
            rc = lib.dmpc_mpc_backward_rec(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(f),
                                           _lib.ptr(self._u), _lib.ptr(self._lo), _lib.ptr(self._hi),
                                           self.n_qp_iter_max, 1 if self.batch_coupled else 0, _lib.ptr(Ks),
                                           _lib.ptr(ks), _lib.ptr(nqp), _lib.ptr(ws), need, _lib.ptr(info),
                                           _lib.stream_ptr(d))
        _lib.check(rc, "dmpc_mpc_backward_rec")
        self.n_qp_iter = nqp
        self.info = info
        assert not bool(torch.isnan(ks).any()) and not bool(torch.isnan(Ks).any())     # mpc_step.py:161-162
        return self._out(Ks), self._out(ks), LqrBackOut(n_total_qp_iter=int(nqp.max().item()))

    # ------------------------------------------------------------------ E3
    def forward_rec(self, Ks, ks, true_cost, true_dynamics, ls_decay, max_ls_iter):
        """-> (new_x, new_u, LqrForOut)   mpc_step.py:175-286"""
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        d = self._dev
        Kd = _lib.f32c(_as_tensor(Ks), d)
        kd = _lib.f32c(_as_tensor(ks), d)
        assert len(Kd) == T, "Ks length error"
        if isinstance(true_cost, QuadCost) and isinstance(true_dynamics, LinDx):
            lib = _lib.load()
            Ct, ct = _lib.f32c(_as_tensor(true_cost.C), d), _lib.f32c(_as_tensor(true_cost.c), d)
            Ft, ft = _lib.f32c(_as_tensor(true_dynamics.F), d), _lib.f32c(_as_tensor(true_dynamics.f), d)
            f32 = dict(dtype=torch.float32, device=d)
            x = torch.empty((T, B, nx), **f32)
            u = torch.empty((T, B, nu), **f32)
            u1 = torch.empty((T, B, nu), **f32)
            costs = torch.empty((B,), **f32)
            old = torch.empty((B,), **f32)
            alphas = torch.empty((B,), **f32)
            objs = torch.empty((T, B), **f32)
            nls = torch.empty((B,), dtype=torch.int32, device=d)
            info = torch.zeros(B, dtype=torch.int32, device=d)
            with _lib.guard(d):
                rc = lib.dmpc_mpc_forward_rec(T, B, nx, nu, _lib.ptr(Kd), _lib.ptr(kd), _lib.ptr(self._u),
                                              _lib.ptr(self._xs), _lib.ptr(self._lo), _lib.ptr(self._hi),
                                              _lib.ptr(Ct), _lib.ptr(ct), _lib.ptr(Ft), _lib.ptr(ft),
                                              float(ls_decay), int(max_ls_iter), _lib.ptr(x), _lib.ptr(u),
                                              _lib.ptr(costs), _lib.ptr(old), _lib.ptr(alphas), _lib.ptr(objs),
                                              _lib.ptr(u1), _lib.ptr(nls), _lib.ptr(info), _lib.stream_ptr(d))
            _lib.check(rc, "dmpc_mpc_forward_rec")
            raise_info(info, "MPCstep.forward_rec")                                   # mpc_step.py:284-285
        elif isinstance(true_cost, QuadCost) and _is_simple_pendulum(true_dynamics) and (nx, nu) == (3, 1):
            # the pendulum of env_dx/pendulum.py evaluated inside the kernel's line search (mpc_step.py:237-240)
            lib = _lib.load()
            Ct, ct = _lib.f32c(_as_tensor(true_cost.C), d), _lib.f32c(_as_tensor(true_cost.c), d)
            g_, m_, l_ = (float(v) for v in true_dynamics.params.detach().cpu().tolist())
            f32 = dict(dtype=torch.float32, device=d)
            x, u, u1 = torch.empty((T, B, nx), **f32), torch.empty((T, B, nu), **f32), torch.empty((T, B, nu), **f32)
            costs, old, alphas = torch.empty((B,), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
            objs = torch.empty((T, B), **f32)
            nls = torch.empty((B,), dtype=torch.int32, device=d)
            info = torch.zeros(B, dtype=torch.int32, device=d)
            with _lib.guard(d):
                rc = lib.dmpc_mpc_forward_rec_pendulum(
                    T, B, _lib.ptr(Kd), _lib.ptr(kd), _lib.ptr(self._u), _lib.ptr(self._xs), _lib.ptr(self._lo),
                    _lib.ptr(self._hi), _lib.ptr(Ct), _lib.ptr(ct), g_, m_, l_, float(true_dynamics.dt),
                    float(true_dynamics.max_torque), float(ls_decay), int(max_ls_iter), _lib.ptr(x), _lib.ptr(u),
                    _lib.ptr(costs), _lib.ptr(old), _lib.ptr(alphas), _lib.ptr(objs), _lib.ptr(u1), _lib.ptr(nls),
                    _lib.ptr(info), _lib.stream_ptr(d))
            _lib.check(rc, "dmpc_mpc_forward_rec_pendulum")
            raise_info(info, "MPCstep.forward_rec")
        else:
            x, u, u1, costs, alphas, objs, nls = self._forward_rec_callable(Kd, kd, true_cost, true_dynamics,
                                                                            ls_decay, max_ls_iter)
        self.alphas, self.n_ls_iter = alphas, nls
        scr = not self.strict_math
        res = LqrForOut(self._out(objs), self._out(du_norm(self._u, u1, scr)), self._out(du_norm(self._u, u, scr)),
                        float(alphas.mean().item()), self._out(costs))
        return self._out(x), self._out(u), res

    def _forward_rec_callable(self, Ks, ks, true_cost, true_dynamics, ls_decay, max_ls_iter, cap=64):
        """Line search around a callable cost / dynamics (mpc_step.py:237-240, 252-253): gains come from the
        HIP kernel, the rollout is torch ops on the device, termination is per trajectory."""
        T, B = self.T, self.n_batch
        u0, xs, lo, hi = self._u, self._xs, self._lo, self._hi

        def cost_of(x, u):
            if isinstance(true_cost, QuadCost):
                return get_cost(T, u, QuadCost(_lib.f32c(_as_tensor(true_cost.C), self._dev),
                                               _lib.f32c(_as_tensor(true_cost.c), self._dev)), x=x), None
            tau = torch.cat((x, u), dim=2)
            per = torch.stack([true_cost(tau[t]) for t in range(T)], dim=0)
            return per.sum(dim=0), per

        old, _ = cost_of(xs, u0)
        alphas = torch.ones(B, dtype=torch.float32, device=self._dev)
        active = torch.ones(B, dtype=torch.bool, device=self._dev)
        nls = torch.zeros(B, dtype=torch.int32, device=self._dev)
        best = None
        u_first = None
        for it in range(cap):
            new_x = [xs[0]]
            new_u = []
            for t in range(T):
                dxt = new_x[t] - xs[t]
                ut = bmv(Ks[t], dxt) + u0[t] + alphas[:, None] * ks[t]
                ut = torch.minimum(torch.maximum(ut, lo[t]), hi[t])
                # float32: snap controls within a few ulps of a bound onto it (see bound_tol in mpc_kernels.hpp)
                tol_lo = 1e-8 + 4 * 1.1920929e-07 * torch.clamp(lo[t].abs(), min=1.0)
                tol_hi = 1e-8 + 4 * 1.1920929e-07 * torch.clamp(hi[t].abs(), min=1.0)
                ut = torch.where(ut - lo[t] <= tol_lo, lo[t], ut)
                ut = torch.where(hi[t] - ut <= tol_hi, hi[t], ut)
                new_u.append(ut)
                if t < T - 1:
                    if isinstance(true_dynamics, LinDx):
                        Fd = _lib.f32c(_as_tensor(true_dynamics.F), self._dev)
                        nxt = bmv(Fd[t], torch.cat((new_x[t], ut), dim=1))
                        if true_dynamics.f is not None:
                            nxt = nxt + _lib.f32c(_as_tensor(true_dynamics.f), self._dev)[t]
                    else:
                        nxt = true_dynamics(new_x[t], ut)
                    assert not bool(torch.isnan(nxt).any())
                    new_x.append(nxt)
            X, U = torch.stack(new_x), torch.stack(new_u)
            cost, per = cost_of(X, U)
            if isinstance(true_cost, QuadCost):
                C_ = _lib.f32c(_as_tensor(true_cost.C), self._dev)
                c_ = _lib.f32c(_as_tensor(true_cost.c), self._dev)
            if per is None:
                tau = torch.cat((X, U), dim=2)
                per = 0.5 * torch.einsum("tbi,tbij,tbj->tb", tau, C_, tau) + (tau * c_).sum(dim=2)
            if best is None:
                best = [X.clone(), U.clone(), cost.clone(), per.clone()]
                u_first = U.clone()
            else:   # only trajectories still searching take the new pass
                m = active
                best[0][:, m], best[1][:, m], best[2][m], best[3][:, m] = X[:, m], U[:, m], cost[m], per[:, m]
            nls += active.to(torch.int32)
            if isinstance(true_cost, QuadCost):
                # current_cost > OLD_COST on the difference, formed per timestep without cancellation (as the kernels do:
                # obj(tau') - obj(tau) = 1/2 d'(C tau') + 1/2 tau'(C d) + c'd): the totals agree to float32 rounding
                # near a fixed point, where the reference decides in float64
                tau1, tau0 = torch.cat((X, U), dim=2), torch.cat((xs, u0), dim=2)
                d = tau1 - tau0
                delta = (d * (0.5 * torch.einsum("tbij,tbj->tbi", C_, tau1) + c_)).sum(dim=(0, 2)) + \
                    0.5 * (tau0 * torch.einsum("tbij,tbj->tbi", C_, d)).sum(dim=(0, 2))
                worse = (delta > 0) & active
            else:
                worse = (cost > old) & active
            alphas = torch.where(worse, alphas * ls_decay, alphas)
            active = worse
            if not bool(active.any()):
                break
        return best[0], best[1], u_first, best[2], alphas, best[3], nls

    # ------------------------------------------------------------------ E4
    def _forward_impl(self, x_init, C_hat, c_hat, F_hat, f_hat):
        d = self._dev
        C, c, F, f = (_lib.f32c(None if t is None else _as_tensor(t).detach(), d) for t in (C_hat, c_hat, F_hat, f_hat))
        x0 = _lib.f32c(_as_tensor(x_init).detach(), d)
        self._retained = dict(x_init=x0, C=C, c=c, F=F, f=f)
        if self.no_op_forward:                                                  # mpc_step.py:297-299
            self._retained.update(x=self._xs, u=self._u)
            return self._out(self._xs), self._out(self._u)
        T, B, nx, nu, ns = self.T, self.n_batch, self.n_state, self.n_ctrl, self.n_sc
        if self._fused_ok():
            lib = _lib.load()
            # the true model's arrays: converted once per (object, tensors) - the step is called in a loop with the same ones
            tm = self._true_model
            key = (self.true_cost.C, self.true_cost.c, self.true_dynamics.F, self.true_dynamics.f)
            if tm is None or any(a_ is not b_ for a_, b_ in zip(tm[0], key)):
                tm = (key, tuple(_lib.f32c(_as_tensor(v), d) for v in key))
                self._true_model = tm
            Ct, ct, Ft, ft = tm[1]
            # ONE allocation for the step's float outputs, one (zeroed) for its integer words; ONE read-back for everything the
            # reference turns into host scalars or asserts (NaN flags, n_total_qp_iter, mean_alphas): dmpc_mpc_step_status.
            # The views of the two blocks are cut while the kernel runs: the launch needs their addresses only.
            sizes = (T * B * nx, T * B * nu, T * B * nu, T * B * nu * nx, T * B * nu, B, B, B, T * B)
            blk = torch.empty((sum(sizes),), dtype=torch.float32, device=d)
            iblk = torch.zeros((3 * B + 8,), dtype=torch.int32, device=d)
            fp, ip = blk.data_ptr(), iblk.data_ptr()
            offs = [fp]
            for n_ in sizes[:-1]:
                offs.append(offs[-1] + 4 * n_)
            p_x, p_u, p_u1, p_Ks, p_ks, p_costs, p_old, p_alphas, p_objs = offs
            p_info, p_nqp, p_nls, p_status = ip, ip + 4 * B, ip + 8 * B, ip + 12 * B
            need = self._ws_need
            if need is None:
                need = self._ws_need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
            ws = _workspace(need, d)
            stream = _lib.stream_ptr(d)
            with _lib.guard(d):
                rc = lib.dmpc_mpc_step_forward(
                    T, B, nx, nu, C.data_ptr(), c.data_ptr(), F.data_ptr(), _lib.ptr(f), self._u.data_ptr(),
                    self._xs.data_ptr(), self._lo.data_ptr(), self._hi.data_ptr(), Ct.data_ptr(), ct.data_ptr(),
                    Ft.data_ptr(), _lib.ptr(ft), 1 if self.need_expand else 0, self.ls_decay, self.max_ls_iter,
                    self.n_qp_iter_max, 1 if self.batch_coupled else 0, p_x, p_u, p_Ks, p_ks, p_costs, p_old, p_alphas, p_objs,
                    p_u1, p_nqp, p_nls, ws.data_ptr(), need, p_info, stream)
                _lib.check(rc, "dmpc_mpc_step_forward")
                _lib.check(lib.dmpc_mpc_step_status(B, p_info, p_nqp, p_alphas, p_status, stream), "dmpc_mpc_step_status")
            x, u, u1, Ks, ks, costs, old, alphas, objs = (v.view(sh) for v, sh in zip(
                blk.split(sizes), ((T, B, nx), (T, B, nu), (T, B, nu), (T, B, nu, nx), (T, B, nu), (B,), (B,), (B,), (T, B))))
            info, nqp, nls, status = iblk[:B], iblk[B:2 * B], iblk[2 * B:3 * B], iblk[3 * B:]
            st = status.cpu().numpy()                                           # the step's one synchronisation
            if int(st[0]) & _lib.INFO_NONFINITE:                                # the reference asserts on NaN (:133-135, 211-223);
                raise AssertionError("MPCstep.forward: NaN/Inf in the solution of %d trajectories" % int(st[2]))   # the kernels flag every non-finite x, u
            self.info, self.n_qp_iter, self.n_ls_iter, self.alphas = info, nqp, nls, alphas
            self.Ks, self.ks = Ks, ks
            scr = not self.strict_math
            self.back_out = LqrBackOut(n_total_qp_iter=int(st[1]))
            u_nom = self._u
            self.for_out = _LazyForOut(self._out(objs), lambda: self._out(du_norm(u_nom, u1, scr)),
                                       lambda: self._out(du_norm(u_nom, u, scr)),
                                       float(st[4:6].view(np.float64)[0]) / B, self._out(costs))
            self._retained.update(x=x, u=u)
            return self._out(x), self._out(u)
        else:
            if self.need_expand:                                                 # mpc_step.py:305-317
                tau = torch.cat((self._xs, self._u), dim=2)
                c = (torch.einsum("tbij,tbj->tbi", C, tau) + c).contiguous()
                f = None
            Ks, ks, self.back_out = self.backward_rec(C, c, F, f)
            x, u, self.for_out = self.forward_rec(Ks, ks, self.true_cost, self.true_dynamics, self.ls_decay,
                                                  self.max_ls_iter)
            x, u = _lib.f32c(x, d), _lib.f32c(u, d)
        assert not bool(torch.isnan(u).any())
        self._retained.update(x=x, u=u)
        return self._out(x), self._out(u)

    def forward(self, inputs):
        """inputs = (x_init, C_hat, c_hat, F_hat, f_hat) -> (x, u)   mpc_step.py:288-328"""
        return self._forward_impl(*inputs)

    def apply(self, inputs):
        x_init, C, c, F, f = (_as_tensor(t) for t in inputs)
        return _MPCstepFn.apply(x_init, C, c, F, f, self)

    __call__ = apply

    # ------------------------------------------------------------------ E5
    def backward(self, target_input_indexes, grad_outputs, retained=None):
        """-> (dx_init, dC, dc, dF, df|None)   mpc_step.py:330-460"""
        r = self._retained if retained is None else retained
        assert r is not None, "backward() before forward()"
        lib = _lib.load()
        T, B, nx, nu, ns = self.T, self.n_batch, self.n_state, self.n_ctrl, self.n_sc
        d = self._dev
        dl_dx, dl_du = grad_outputs
        gx = None if dl_dx is None else _lib.f32c(_as_tensor(dl_dx), d)
        gu = None if dl_du is None else _lib.f32c(_as_tensor(dl_du), d)
        if gx is not None:
            assert list(gx.shape) == [T, B, nx]
        if gu is not None:
            assert list(gu.shape) == [T, B, nu]
        f32 = dict(dtype=torch.float32, device=d)
        dx0 = torch.empty((B, nx), **f32)
        dC = torch.empty((T, B, ns, ns), **f32)
        dc = torch.empty((T, B, ns), **f32)
        dF = torch.zeros(tuple(r["F"].shape), **f32)                              # zeros_like(F_hat), :428
        df = torch.empty((T - 1, B, nx), **f32) if r["f"] is not None else None   # :437-444
        info = torch.zeros(B, dtype=torch.int32, device=d)
        need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
        ws = _workspace(need, d)
        with _lib.guard(d):
            rc = lib.dmpc_mpc_step_backward(T, B, nx, nu, _lib.ptr(r["C"]), _lib.ptr(r["c"]), _lib.ptr(r["F"]),
                                            _lib.ptr(r["x"]), _lib.ptr(r["u"]), _lib.ptr(self._lo), _lib.ptr(self._hi),
                                            _lib.ptr(gx), _lib.ptr(gu), _lib.ptr(dx0), _lib.ptr(dC), _lib.ptr(dc),
                                            _lib.ptr(dF), _lib.ptr(df), None, None, None, None, 0.0, _lib.ptr(ws), need,
                                            _lib.ptr(info), _lib.stream_ptr(d))
        _lib.check(rc, "dmpc_mpc_step_backward")
        return tuple(None if g is None else self._out(g) for g in (dx0, dC, dc, dF, df))
