"""`DiffLqr`, `LqrNet`, `LqrNet_cost_dx` - same names and call signatures as
lqr/differentiable_lqr.py of the reference; chainer.FunctionNode / chainer.Link become a
torch.autograd.Function wrapper / torch.nn.Module.

Forward = fused LQR solve kernel (`dmpc_lqr_solve`); backward = analytic KKT gradient
(`dmpc_lqr_kkt_grad`: second LQR solve + co-state sweeps + outer products), both hand-written HIP.
Where the generated stream serves the size the pair runs in its training form: the forward solve also leaves its
gains, the control blocks of its Q-functions and its value functions [V_t | v_t] in HBM (`dmpc_lqr_solve_saving`), and
the gradient is ONE launch that reads neither C nor c (`dmpc_lqr_kkt_grad_saved`): the second solve - same C, same F -
only redoes the affine recursion with the saved gains, and the co-states are value gradients, lambda_t = V_t x_t + v_t,
d_lambda_t = V_t dx_t + v'_t, formed while d_tau is rolled out (`DMPC_NO_ADJOINT=1`: re-solve + co-state sweep).

Reference quirks kept by default, switch off with `strict_math=True`:
  dC_t = 0.5*(d_tau (x) tau) + (tau (x) d_tau)      differentiable_lqr.py:128
  df   = d_lambda[0:T-1]                            differentiable_lqr.py:133
"""
import os

import numpy as np
import torch

from . import _lib
from .lqr_recursion import (_as_tensor, _device_of, _workspace, saving_solve_available, solve_device, solve_device_f64,
                            solve_saving_device)
from .util import expand_time_batch


def kkt_grad_device(C, c, F, x, u, grad_x, grad_u, T, n_state, n_ctrl, strict_math=False, info=None,
                    need_dC=True, need_dF=True, need_df=True, saved=None):
    """Raw KKT gradient on float32 device tensors -> (d_x_init, dC, dc, dF, df).
    saved = (Ks, Quu, Qxu, Vv) of `solve_saving_device`: the whole gradient is one launch that reads neither C nor c
    (co-states as value gradients, `dmpc_lqr_kkt_grad_saved`); saved = (Ks, Quu, Qxu): the second solve reuses the
    forward solve's gains instead of repeating the Riccati sweep, the co-state sweep reads C."""
    lib = _lib.load()
    _lib.require_gpu()
    dev = C.device
    B = C.shape[1]
    nx, nu = n_state, n_ctrl
    ns = nx + nu
    f32 = dict(dtype=torch.float32, device=dev)
    dx0 = torch.empty((B, nx), **f32)
    dC = torch.empty((T, B, ns, ns), **f32) if need_dC else None
    dc = torch.empty((T, B, ns), **f32)
    dF = torch.empty((T - 1, B, nx, ns), **f32) if need_dF else None
    df = torch.empty((T - 1, B, nx), **f32) if need_df else None
    need = lib.dmpc_lqr_kkt_workspace_bytes(T, B, nx, nu)
    ws = _workspace(need, dev)
    with _lib.guard(dev):
        rc = _lib.E_UNSUPPORTED
        if saved is not None:
            Ks, Quu, Qxu = saved[:3]
            Vv = saved[3] if len(saved) > 3 else None
            rc = lib.dmpc_lqr_kkt_grad_saved(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(x), _lib.ptr(u),
                                             _lib.ptr(Ks), _lib.ptr(Quu), _lib.ptr(Qxu), _lib.ptr(Vv), _lib.ptr(grad_x),
                                             _lib.ptr(grad_u),
                                             1 if strict_math else 0, _lib.ptr(dx0), _lib.ptr(dC), _lib.ptr(dc),
                                             _lib.ptr(dF), _lib.ptr(df), _lib.ptr(ws), need, _lib.ptr(info),
                                             _lib.stream_ptr(dev))
        if rc == _lib.E_UNSUPPORTED:      # (nothing was launched)
            rc = lib.dmpc_lqr_kkt_grad(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(x), _lib.ptr(u),
                                       _lib.ptr(grad_x), _lib.ptr(grad_u), 1 if strict_math else 0, _lib.ptr(dx0),
                                       _lib.ptr(dC), _lib.ptr(dc), _lib.ptr(dF), _lib.ptr(df), _lib.ptr(ws), need,
                                       _lib.ptr(info), _lib.stream_ptr(dev))
    _lib.check(rc, "dmpc_lqr_kkt_grad")
    return dx0, dC, dc, dF, df


class _DiffLqrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_init, C, c, F, f, node):
        x, u = node._forward_impl(x_init, C, c, F, f)
        ctx.node = node
        # per-call state lives on the autograd ctx: one DiffLqr (e.g. LqrNet.lqr_layer) may run forward several times
        # before any backward (summed minibatches, a validation pass in between); node._retained only serves the
        # reference-style explicit forward()/backward() pair
        ctx.retained = node._retained
        ctx.had_f = f is not None
        ctx.in_meta = [(t.dtype, t.device, tuple(t.shape)) if t is not None else None for t in (x_init, C, c, F, f)]
        return x, u

    @staticmethod
    def backward(ctx, grad_x, grad_u):
        node = ctx.node
        grads = node.backward((0, 1, 2, 3, 4), (grad_x, grad_u), retained=ctx.retained)
        out = []
        for g, meta in zip(grads, ctx.in_meta):
            if meta is None or g is None:
                out.append(None)
                continue
            dtype, device, shape = meta
            g = g.to(device=device, dtype=dtype)
            if tuple(g.shape) != shape:           # F / f given with T slices: the last one gets no gradient
                pad = torch.zeros(shape, dtype=dtype, device=device)
                pad[: g.shape[0]] = g
                g = pad
            out.append(g)
        return tuple(out) + (None,)


def kkt_grad_device_f64(C, c, F, x, u, grad_x, grad_u, T, n_state, n_ctrl, strict_math=False, info=None):
    """The KKT gradient in float64 on float64 device tensors (`dmpc_lqr_kkt_grad_f64`) -> (d_x_init, dC, dc, dF, df)."""
    lib = _lib.load()
    _lib.require_gpu()
    dev = C.device
    B = C.shape[1]
    nx, nu = n_state, n_ctrl
    ns = nx + nu
    f64 = dict(dtype=torch.float64, device=dev)
    dx0, dC, dc = torch.empty((B, nx), **f64), torch.empty((T, B, ns, ns), **f64), torch.empty((T, B, ns), **f64)
    dF, df = torch.empty((T - 1, B, nx, ns), **f64), torch.empty((T - 1, B, nx), **f64)
    need = lib.dmpc_lqr_f64_workspace_bytes(T, B, nx, nu)
    ws = _workspace(need, dev)
    with _lib.guard(dev):
        rc = lib.dmpc_lqr_kkt_grad_f64(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(x), _lib.ptr(u),
                                       _lib.ptr(grad_x), _lib.ptr(grad_u), 1 if strict_math else 0, _lib.ptr(dx0), _lib.ptr(dC),
                                       _lib.ptr(dc), _lib.ptr(dF), _lib.ptr(df), _lib.ptr(ws), need, _lib.ptr(info),
                                       _lib.stream_ptr(dev))
    _lib.check(rc, "dmpc_lqr_kkt_grad_f64")
    return dx0, dC, dc, dF, df


class DiffLqr:
    """Differentiable LQR (module 1 of Amos et al. 2018).  lqr/differentiable_lqr.py:21-142."""

    def __init__(self, T, n_batch, n_state, n_ctrl, strict_math=False, save_gains=True, precision="float32"):
        # precision="float64" (not in the reference's signature): forward and backward on the float64 kernels - the
        # reference's own precision, an order of magnitude slower
        assert precision in ("float32", "float64")
        self.precision = precision
        self.T = int(T)
        self.n_batch = int(n_batch)
        self.n_state = int(n_state)
        self.n_ctrl = int(n_ctrl)
        self.n_sc = self.n_state + self.n_ctrl
        self.strict_math = bool(strict_math)
        self.save_gains = bool(save_gains) and os.environ.get("DMPC_NO_SAVED_GAINS") != "1"
        self._retained = None
        self.info = None

    def check_type_forward(self, inputs):
        """float inputs only, exactly five of them; f may be None (differentiable_lqr.py:41-63)"""
        assert len(inputs) == 5
        for t in inputs[:4]:
            assert _as_tensor(t).dtype.is_floating_point
        if inputs[4] is not None:
            assert _as_tensor(inputs[4]).dtype.is_floating_point

    # -- computation on the device -------------------------------------------------------------
    def _forward_impl(self, x_init, C, c, F, f):
        x_init, C, c, F, f = (_as_tensor(t) for t in (x_init, C, c, F, f))
        T, B, nx, nu, ns = self.T, self.n_batch, self.n_state, self.n_ctrl, self.n_sc
        assert list(x_init.shape) == [B, nx]
        assert list(C.shape) == [T, B, ns, ns], "C dim mismatch"
        assert list(c.shape) == [T, B, ns], "c dim mismatch"
        assert F.shape[0] in (T - 1, T) and list(F.shape[1:]) == [B, nx, ns], "F dim mismatch"
        if f is not None:
            assert list(f.shape) == [T - 1, B, nx], " f dim mismatch"
        dev = _device_of(C, c, F, x_init)
        if self.precision == "float64":
            d = [None if t is None else t.detach().to(device=dev, dtype=torch.float64).contiguous() for t in (x_init, C, c, F, f)]
            self.info = torch.zeros(B, dtype=torch.int32, device=dev)
            x, u, _, _ = solve_device_f64(d[1], d[2], d[3], d[4], d[0], None, T, nx, nu, info=self.info)
            self._retained = dict(x_init=d[0], C=d[1], c=d[2], F=d[3], x=x, u=u, saved=None, out_dtype=C.dtype,
                                  out_device=C.device, versions=tuple(t._version for t in d[:4]))
            return x.to(device=C.device, dtype=C.dtype), u.to(device=C.device, dtype=C.dtype)
        d = [_lib.f32c(t.detach() if t is not None else None, dev) for t in (x_init, C, c, F, f)]
        saved = None
        got = None
        use_saving = self.save_gains and saving_solve_available(T, B, nx, nu)
        # (the saving solve writes every trajectory's flags itself; the plain one ors into a cleared array)
        self.info = (torch.empty if use_saving else torch.zeros)(B, dtype=torch.int32, device=dev)
        if use_saving:
            # training form: K_t, Quu_t, Qxu_t stay in HBM (36 % more bytes written by the forward solve) and the
            # backward pass's second solve only redoes the affine recursion with them
            got = solve_saving_device(d[1], d[2], d[3], d[4], d[0], T, nx, nu, info=self.info)
        if got is not None:
            x, u, Ks, _, Quu, Qxu, Vv = got
            saved = (Ks, Quu, Qxu, Vv)
        else:
            if use_saving:
                self.info.zero_()
            x, u, _, _ = solve_device(d[1], d[2], d[3], d[4], d[0], None, T, nx, nu, info=self.info)
        # d[i] may share storage with the caller's tensors (detach() keeps the version counter): an in-place update of C
        # or F between forward and backward would be mixed with gains computed from their old values - checked in backward
        self._retained = dict(x_init=d[0], C=d[1], c=d[2], F=d[3], x=x, u=u, saved=saved, out_dtype=C.dtype,
                              out_device=C.device, versions=tuple(t._version for t in d[:4]))
        return x.to(device=C.device, dtype=C.dtype), u.to(device=C.device, dtype=C.dtype)

    # -- reference API ---------------------------------------------------------------------------
    def forward(self, inputs):
        """inputs = (x_init, C, c, F, f) -> (x, u); retains inputs 0-3 and both outputs (:65-76)"""
        self.check_type_forward(inputs)
        return self._forward_impl(*inputs)

    def apply(self, inputs):
        """differentiable call: gradients flow to whichever inputs require grad"""
        self.check_type_forward(inputs)
        x_init, C, c, F, f = (_as_tensor(t) for t in inputs)
        return _DiffLqrFn.apply(x_init, C, c, F, f, self)

    __call__ = apply

    def backward(self, target_input_indexes, grad_outputs, retained=None):
        """-> (d_x_init, dC, dc, dF, df) for upstream (grad_x, grad_u)  (differentiable_lqr.py:78-142)"""
        r = self._retained if retained is None else retained
        assert r is not None, "backward() before forward()"
        now = tuple(r[k]._version for k in ("x_init", "C", "c", "F"))
        if now != r.get("versions", now):
            raise RuntimeError("DiffLqr.backward: x_init, C, c or F was modified in place after forward() (tensor versions "
                               "%r -> %r); the retained solution and gains belong to the old values" % (r["versions"], now))
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        grad_x, grad_u = grad_outputs
        dev = r["C"].device
        if self.precision == "float64":
            g64 = lambda g, n: (_as_tensor(g).to(device=dev, dtype=torch.float64).contiguous() if g is not None
                                else torch.zeros((T, B, n), dtype=torch.float64, device=dev))
            out = kkt_grad_device_f64(r["C"], r["c"], r["F"], r["x"], r["u"], g64(grad_x, nx), g64(grad_u, nu), T, nx, nu,
                                      strict_math=self.strict_math)
            return tuple(g.to(device=r["out_device"], dtype=r["out_dtype"]) for g in out)
        gx = _lib.f32c(_as_tensor(grad_x), dev) if grad_x is not None else torch.zeros((T, B, nx), device=dev)
        gu = _lib.f32c(_as_tensor(grad_u), dev) if grad_u is not None else torch.zeros((T, B, nu), device=dev)
        assert list(gx.shape) == [T, B, nx] and list(gu.shape) == [T, B, nu]
        out = kkt_grad_device(r["C"], r["c"], r["F"], r["x"], r["u"], gx, gu, T, nx, nu,
                              strict_math=self.strict_math, saved=r.get("saved"))
        return tuple(g.to(device=r["out_device"], dtype=r["out_dtype"]) for g in out)


class LqrNet(torch.nn.Module):
    """LQR layer whose dynamics [A|B] are learnable (differentiable_lqr.py:145-198)."""

    def __init__(self, T, n_batch, n_state, n_ctrl, seed, dtype=torch.float64):
        super().__init__()
        self.T, self.n_batch, self.n_state, self.n_ctrl = T, n_batch, n_state, n_ctrl
        self.n_sc = n_ctrl + n_state
        np.random.seed(seed)                      # same draws as the reference (:167-172)
        alpha = 0.2
        A = np.eye(n_state) + alpha * np.random.randn(n_state, n_state)
        B = np.random.randn(n_state, n_ctrl)
        self.A = torch.nn.Parameter(torch.as_tensor(A, dtype=dtype))
        self.B = torch.nn.Parameter(torch.as_tensor(B, dtype=dtype))
        self.lqr_layer = DiffLqr(T, n_batch, n_state, n_ctrl)

    def forward(self, inputs):
        x_init, C, c, f = inputs
        ab_cat = torch.cat((self.A, self.B), dim=1)
        large_f_learner = expand_time_batch(ab_cat, self.T - 1, self.n_batch)
        assert list(large_f_learner.shape) == [self.T - 1, self.n_batch, self.n_state, self.n_sc]
        return self.lqr_layer.apply((x_init, C, c, large_f_learner, f))


class LqrNet_cost_dx(torch.nn.Module):
    """LQR layer with learnable cost (C, c) and dynamics (differentiable_lqr.py:201-248)."""

    def __init__(self, T, n_batch, n_state, n_ctrl, seed, dtype=torch.float64):
        super().__init__()
        self.T, self.n_batch, self.n_state, self.n_ctrl = T, n_batch, n_state, n_ctrl
        self.n_sc = n_ctrl + n_state
        np.random.seed(seed)                      # (:222-231)
        alpha = 0.2
        A = np.eye(n_state) + alpha * np.random.randn(n_state, n_state)
        B = np.random.randn(n_state, n_ctrl)
        C = np.eye(self.n_sc) + alpha * np.random.randn(self.n_sc, self.n_sc)
        c = np.random.randn(self.n_sc)
        self.A = torch.nn.Parameter(torch.as_tensor(A, dtype=dtype))
        self.B = torch.nn.Parameter(torch.as_tensor(B, dtype=dtype))
        self.C = torch.nn.Parameter(torch.as_tensor(C, dtype=dtype))
        self.c = torch.nn.Parameter(torch.as_tensor(c, dtype=dtype))
        self.lqr_layer = DiffLqr(T, n_batch, n_state, n_ctrl)

    def forward(self, inputs):
        x_init, f = inputs
        ab_cat = torch.cat((self.A, self.B), dim=1)
        large_f_learner = expand_time_batch(ab_cat, self.T - 1, self.n_batch)
        C = expand_time_batch(self.C, self.T, self.n_batch)
        c = expand_time_batch(self.c, self.T, self.n_batch)
        return self.lqr_layer.apply((x_init, C, c, large_f_learner, f))
