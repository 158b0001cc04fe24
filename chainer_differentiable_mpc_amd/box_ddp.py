"""`BoxDDP` - the outer box-constrained iLQR loop with the constructor and call signature of
mpc/box_ddp.py:24-291 of the reference.  Host-side control flow over device tensors; every heavy step
(`get_traj` rollouts aside, which are tiny torch ops) is an `MPCstep` running on the HIP kernels.

    solver = BoxDDP(T, u_lower, u_upper, n_batch, n_state, n_ctrl, u_init, ...)
    x, u, costs = solver((x_init, cost, dynamics))      # cost: QuadCost | callable, dynamics: LinDx | callable

Per-sample "best so far" bookkeeping is a masked `torch.where` instead of the reference's Python loop over
the batch (:200-209); stop tests, the final no-op MPCstep node that carries the gradient (:247-259) and the
detach mask for unconverged samples (:263-289) follow the reference.
"""
import ctypes
import os
import warnings

import torch

from . import _lib
from .approximate import approximate_cost, linearize_dynamics
from .lqr_recursion import _as_tensor, _device_of, _workspace
from .mpc_step import MPCstep, _MPCstepTiledFn
from .util import LinDx, QuadCost, TiledQuadCost, get_cost, get_traj


_UNRESOLVED = []     # solvers whose device loop has not been read back yet (lazy_status)


def table_log(tag, d, _seen=[]):
    """markdown-ish progress rows (util.py:67-91)"""
    if tag not in _seen:
        print('| ' + ' | '.join(str(di[0]) for di in d) + ' |')
        _seen.append(tag)
    row = []
    for di in d:
        row.append(di[2].format(float(di[1])) if len(di) == 3 else str(di[1]))
    print('| ' + ' | '.join(row) + ' |')


class BoxDDP(torch.nn.Module):
    def __init__(self, T, u_lower, u_upper, n_batch, n_state, n_ctrl, u_init, eps=1e-5, not_improved_lim=5,
                 line_search_decay=0.2, max_line_search_iter=10, best_cost_eps=1e-4, max_iter=10,
                 detach_unconverged=True, exit_unconverged=True, verbose=False, ilqr_verbose=False,
                 update_dynamics=True, quiet=False, device_loop=True, batch_coupled=False, lazy_status=False, graph=True):
        super().__init__()
        self.T, self.n_batch, self.n_state, self.n_ctrl = T, n_batch, n_state, n_ctrl
        self.n_sc = n_state + n_ctrl
        self.eps = eps
        self.not_improved_lim = not_improved_lim
        self.ls_decay = line_search_decay
        self.max_ls_iter = max_line_search_iter
        self.best_cost_eps = best_cost_eps
        self.max_iter = max_iter
        self.verbose = verbose
        self.ilqr_verbose = ilqr_verbose
        self.u_init = u_init
        self.detach_unconverged = detach_unconverged
        self.exit_unconverged = exit_unconverged
        self.update_dynamics = update_dynamics
        self.quiet = quiet          # suppress the reference's "Converged" / "Not improved lim" prints
        # QuadCost with a LinDx or the built-in pendulum: the whole loop is one chain of launches with the
        # stop tests on the device (`dmpc_box_ddp`); False keeps the host loop over MPCstep objects
        self.device_loop = device_loop
        # PNQP termination inside every MPC step: per trajectory (default; shard-invariant) or the reference's
        # batch-global tests (pnqp.py:139-144,172,187; the batch must fit one cooperative launch).  The outer
        # stop tests (box_ddp.py:223-230) are batch-global in both modes, as in the reference.
        self.batch_coupled = batch_coupled
        # The device-driven loop ends with ONE read-back (loop state, input asserts, NaN flags).  lazy_status=True does
        # not wait for it inside forward(): the solution and - for a `TiledQuadCost` - the gradient node with its detach
        # mask need nothing from the host, so a training loop keeps launching while the GPU works.  `status`, `n_iter`,
        # `info`, the asserts and the reference's non-convergence warning then happen on first access, and at the latest
        # when the next solve of any BoxDDP starts (errors surface one call late instead of never).
        self.lazy_status = lazy_status
        # device loop only: a solve called again on the same buffers replays its chain of launches from a hipGraph (_device_loop)
        self.graph = graph
        self._graphs = {}
        self._fast = None           # the last recorded call, as forward() recognises it (_replay)
        self._host_state = None     # lazy_status, chain launched: (pinned int32[8], event) the loop state is copied to
        self._pending = None        # (state [8] int32 on the device, info [B]) of a solve not read back yet
        self._warn_unconverged = False
        self._status = None
        self._info = None           # MPC step flags of the device loop, per trajectory
        self._n_iter = 0
        self._bounds_on = None
        self._u_zero = None
        self._best_norm_max = None
        if isinstance(u_lower, (int, float)):       # scalar bounds are broadcast to [T,B,nu] (:68-90)
            u_lower = torch.full((T, n_batch, n_ctrl), float(u_lower))
            u_upper = torch.full((T, n_batch, n_ctrl), float(u_upper))
        self.u_lower = _as_tensor(u_lower)
        self.u_upper = _as_tensor(u_upper)
        assert list(self.u_lower.shape) == [T, n_batch, n_ctrl], 'actual' + str(tuple(self.u_lower.shape))
        assert list(self.u_upper.shape) == [T, n_batch, n_ctrl]

    # (BoxDDP is a torch.nn.Module: a plain `self.x = ...` goes through Module.__setattr__ - parameter / buffer / submodule checks,
    # ~1.5 us each, fourteen of them per solve; the per-solve bookkeeping below writes the instance dictionary directly)
    def _say(self, msg):
        self.__dict__["_status"] = msg.strip()
        if not self.quiet:
            print(msg)

    # -- results of the loop's read-back (resolved on first access when lazy_status deferred it)
    @property
    def status(self):
        self._resolve()
        return self._status

    @status.setter
    def status(self, v):
        self.__dict__["_status"] = v

    @property
    def n_iter(self):
        self._resolve()
        return self._n_iter

    @n_iter.setter
    def n_iter(self, v):
        self.__dict__["_n_iter"] = v

    @property
    def info(self):
        self._resolve()
        return self._info

    @info.setter
    def info(self, v):
        self.__dict__["_info"] = v

    def _resolve(self):
        """the one synchronisation of the device loop: loop state, the input asserts of MPCstep (mpc_step.py:133-138), NaN
        flags, the reference's status line and its non-convergence warning (box_ddp.py:223-230,263-273)"""
        if self._pending is None:
            return
        state, info = self._pending
        self.__dict__["_pending"] = None
        if self in _UNRESOLVED:
            _UNRESOLVED.remove(self)
        if isinstance(state, tuple):       # (pinned words, event): copied out behind the chain, waited for on ITS event only -
            state[1].synchronize()         # a `.cpu()` here would wait for everything enqueued since (the update's backward
            st = state[0].tolist()         # pass, the optimiser step) and stall a loop that is meant to run ahead of the device
        else:
            st = state if isinstance(state, list) else state.cpu().tolist()      # (a replayed chain brings its state along)
        self.__dict__["_best_norm_max"] = bool(st[7])         # full_du_norm of the best iterate above eps somewhere (:263)
        assert not st[4]
        assert not st[5], " lower is larger than upper"
        if st[6]:                                 # the reference asserts on NaN inside every MPC step
            raise AssertionError("BoxDDP: NaN/Inf in the solution of %d trajectories" % st[6])
        self.__dict__["_info"] = info
        self.__dict__["_n_iter"] = st[1]
        if st[2] in self._STATUS:
            self._say(self._STATUS[st[2]])
        if self._warn_unconverged and self._best_norm_max:
            self.__dict__["_warn_unconverged"] = False
            self._warn()

    def _warn(self):
        if self.verbose:
            print("LQR Warning: All examples did not converge to a fixed point.")
            print("Detaching and *not* backpropping through the bad examples.")
        warnings.warn("LQR Warning: All examples did not converge to a fixed point.")

    _STATUS = {1: "Converged", 2: "Not improved lim", 3: "Not Converged "}

    @staticmethod
    def _nominal(T, u, x_init, dyn):
        """get_traj (util.py:239-277).  Under a LinDx on the GPU the rollout kernel is used: it sums in the order of
        the MPC step's own line-search rollout, so unchanged controls reproduce the nominal trajectory exactly and
        the stop test `full_du_norm < eps` sees zero at a fixed point (torch's bmv rounds differently)."""
        if isinstance(dyn, LinDx) and isinstance(x_init, torch.Tensor) and x_init.is_cuda and u.is_cuda:
            lib = _lib.load()
            d = x_init.device
            B, nx, nu = x_init.shape[0], x_init.shape[1], u.shape[2]
            x0, ud, F, f = _lib.f32c(x_init, d), _lib.f32c(u, d), _lib.f32c(dyn.F, d), _lib.f32c(dyn.f, d)
            x = torch.empty((T, B, nx), dtype=torch.float32, device=d)
            with _lib.guard(d):
                rc = lib.dmpc_lin_rollout(T, B, nx, nu, _lib.ptr(x0), _lib.ptr(ud), _lib.ptr(F), _lib.ptr(f),
                                          _lib.ptr(x), _lib.stream_ptr(d))
            _lib.check(rc, "dmpc_lin_rollout")
            return x.to(x_init.dtype)
        return get_traj(T, u, x_init, dyn)

    def _device_loop(self, x_init, cost, dynamics, u, lo, hi):
        """box_ddp.py:123-230 behind one C call; returns (best, last full_du_norm) or None when the problem is not
        of the supported kind (then the host loop runs)."""
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        ns = nx + nu
        if isinstance(dynamics, LinDx):
            kind, params, Fd, fd = 0, None, dynamics.F, dynamics.f
        elif hasattr(dynamics, "fused_ok") and dynamics.fused_ok(x_init, u) and (nx, nu) == (3, 1):
            g_, m_, l_ = dynamics.host_params()
            params = (ctypes.c_float * 6)(g_, m_, l_, float(dynamics.dt), float(dynamics.max_torque),
                                          1.0 if dynamics.clamp_grad_closed else 0.0)
            kind, Fd, fd = 1, None, None
        else:
            return None
        lib = _lib.load()
        _lib.require_gpu()
        if not torch.cuda.is_current_stream_capturing():
            for other in list(_UNRESOLVED):       # earlier solves whose read-back was deferred: their asserts come now
                other._resolve()
        d = _device_of(x_init, cost.C, u)
        x0 = _lib.f32c(x_init.detach(), d)
        C, c = _lib.f32c(_as_tensor(cost.C).detach(), d), _lib.f32c(_as_tensor(cost.c).detach(), d)
        F = None if Fd is None else _lib.f32c(_as_tensor(Fd).detach(), d)
        f = None if fd is None else _lib.f32c(_as_tensor(fd).detach(), d)
        if list(C.shape) != [T, B, ns, ns] or list(c.shape) != [T, B, ns]:
            return None
        if F is not None and (F.shape[0] not in (T - 1, T) or list(F.shape[1:]) != [B, nx, ns]):
            return None
        if f is not None and list(f.shape) != [T - 1, B, nx]:
            return None
        u0, lo_, hi_ = _lib.f32c(u, d), _lib.f32c(lo, d), _lib.f32c(hi, d)
        # one float and one int32 allocation per solve; the input asserts of MPCstep (mpc_step.py:133-138) and the
        # reductions over info / full_du_norm come back in state[4:8] with the loop state (box_ddp_summary_kernel)
        n_x, n_u = T * B * nx, T * B * nu
        need = lib.dmpc_box_ddp_workspace_bytes(T, B, nx, nu)
        scalars = (float(self.eps), int(self.not_improved_lim), float(self.ls_decay), int(self.max_ls_iter),
                   float(self.best_cost_eps), int(self.max_iter), 1 if self.batch_coupled else 0)

        n_f = n_x + n_u + 3 * B

        def split(both):      # ONE allocation: the float outputs, then (as int32) the flags and the loop state
            return both[:n_f], both[n_f:].view(torch.int32)

        def buffers():
            return split(torch.empty((n_f + B + 8,), dtype=torch.float32, device=d))   # (the ints: cleared by the chain's first launch)

        def launch(out, ints, ws):
            bx, bu = out[:n_x], out[n_x:n_x + n_u]
            tail = out[n_x + n_u:]
            with _lib.guard(d):
                return lib.dmpc_box_ddp(T, B, nx, nu, _lib.ptr(x0), _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(f), kind,
                                        None if params is None else ctypes.cast(params, ctypes.c_void_p), _lib.ptr(u0),
                                        _lib.ptr(lo_), _lib.ptr(hi_), scalars[0], scalars[1], scalars[2], scalars[3], scalars[4],
                                        scalars[5], 20, 1, scalars[6], _lib.ptr(bx), _lib.ptr(bu),
                                        _lib.ptr(tail[:B]), _lib.ptr(tail[B:2 * B]), _lib.ptr(tail[2 * B:]), _lib.ptr(ints[B:]),
                                        _lib.ptr(ws), need, _lib.ptr(ints[:B]), _lib.stream_ptr(d))

        # The chain of 22-23 launches as ONE hipGraph (round 5; verdict r04 item 2): a solve called again on the SAME buffers
        # (an MPC loop that updates its state in place, a benchmark) replays the chain recorded at its second call - the ~40 us
        # of host time in front of the chain's first launch go (0.28 -> 0.24 ms at B = 128,
        # profiles/r05/box_ddp_graph_ab.txt).  The key is every pointer, shape and scalar the chain was recorded with; a
        # call with other buffers runs the chain directly, as before.  The graph owns its outputs and workspace: what the
        # caller gets is a copy (one more small launch).  `graph=False` / DMPC_NO_DDP_GRAPH=1: never.
        capturing = torch.cuda.is_current_stream_capturing()
        key = None
        if self.graph and not capturing and not self.batch_coupled and os.environ.get("DMPC_NO_DDP_GRAPH") != "1":
            # (batch_coupled: its grid barriers need cooperative launches, which a stream capture does not take)
            key = (d, _lib.stream_ptr(d), T, B, nx, nu, kind, None if params is None else tuple(params), scalars) + tuple(
                None if t is None else (t.data_ptr(), tuple(t.shape)) for t in (x0, C, c, F, f, u0, lo_, hi_))
        entry = self._graphs.get(key) if key is not None else None
        rc = 0
        host_state = None
        if entry is not None and entry[0] is None:     # second call on these buffers: record the chain
            g_both = torch.empty((n_f + B + 8,), dtype=torch.float32, device=d)
            g_out, g_ints = split(g_both)
            g_ws = torch.empty(max(need, 1), dtype=torch.uint8, device=d)
            g_host = torch.empty((8,), dtype=torch.int32, pin_memory=True)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    rc = launch(g_out, g_ints, g_ws)
                    g_host.copy_(g_ints[B:], non_blocking=True)        # the loop state travels to the host inside the graph
            except RuntimeError:      # a launch the capture does not take: this solver launches its chain from now on
                rc = _lib.E_UNSUPPORTED
                self.graph = False
                self._graphs.clear()
            if rc == 0:
                entry = (g, g_both, n_f, g_ws, g_host, torch.cuda.Event())
                self._graphs[key] = entry
            else:
                self._graphs.pop(key, None)
                entry = None
                key = None
                rc = 0
        if entry is not None and entry[0] is not None:
            g, g_both, _nf, _ws, g_host, ev = entry
            g.replay()
            ev.record()
            out, ints = split(g_both.clone())                     # the caller's own copy (the graph owns its buffers)
            if not self.lazy_status:
                while not ev.query():                              # the one synchronisation of the loop: the chain, not the
                    pass                                           # copies (polled: a blocking wait wakes up ~20 us late)
                host_state = g_host.tolist()
            else:
                host_state = (g_host, ev)                          # read when the status is asked for (before the next replay)
        else:
            if key is not None and rc == 0:
                if len(self._graphs) >= 4:             # (a handful of buffer sets per solver; the oldest goes)
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[key] = (None,)            # seen once: the next call on these buffers records
            out, ints = buffers()
            rc = launch(out, ints, _workspace(need, d))
            if rc == 0 and self.lazy_status and not capturing:    # the loop state to pinned memory behind the chain, with its event
                if self._host_state is None:
                    self.__dict__["_host_state"] = (torch.empty((8,), dtype=torch.int32, pin_memory=True), torch.cuda.Event())
                self._host_state[0].copy_(ints[B:], non_blocking=True)
                self._host_state[1].record()
                host_state = self._host_state
        bx, bu = out[:n_x].view(T, B, nx), out[n_x:n_x + n_u].view(T, B, nu)
        bc, bn, ln = out[n_x + n_u:].view(3, B).unbind(0)
        info, state = ints[:B], ints[B:]
        if rc == _lib.E_UNSUPPORTED:
            return None
        _lib.check(rc, "dmpc_box_ddp")
        self.__dict__["_loop_flag"] = state[7:8]              # device int: some trajectory's best full_du_norm is above eps (:263)
        self.__dict__["_warn_unconverged"] = False
        if capturing:
            # recorded into a hipGraph, not executed: `state` holds nothing until a replay, and a replay is not this call -
            # there is no read-back to defer (a captured solve reports through its outputs and device flags only)
            self.__dict__["_pending"] = None
            if self in _UNRESOLVED:
                _UNRESOLVED.remove(self)
        else:
            self.__dict__["_pending"] = (state if host_state is None else host_state, info)
            if self not in _UNRESOLVED:
                _UNRESOLVED.append(self)
            if not self.lazy_status:
                self._resolve()                   # the one synchronisation of the loop
        dev, dt = x_init.device, x_init.dtype
        self.__dict__["_fast"] = None
        if entry is not None and entry[0] is not None and dt == torch.float32 and dev == d and x0 is not None and \
                x0.data_ptr() == x_init.data_ptr() and C.data_ptr() == cost.C.data_ptr() and c.data_ptr() == cost.c.data_ptr():
            # what forward() checks before it replays this recording without going through any of the above
            self.__dict__["_fast"] = (entry, x_init, cost, cost.C, cost.c, dynamics, x0.data_ptr(), C.data_ptr(), c.data_ptr(),
                          _lib.stream_ptr(d), scalars, None if params is None else tuple(params), d)
        best = {'x': bx.to(device=dev, dtype=dt), 'u': bu.to(device=dev, dtype=dt),
                'costs': bc.to(device=dev, dtype=dt), 'full_du_norm': bn.to(device=dev, dtype=dt)}
        return best, ln.to(device=dev, dtype=dt)

    def _replay(self, inputs):
        """forward() for the call an MPC loop makes over and over: the SAME tensors (updated in place), nothing to
        differentiate, the chain already recorded (`_device_loop`) - replay, copy out, read the loop state.  None: not that call."""
        entry, x_init, cost, Cc, cc, dynamics, p_x, p_C, p_c, stream, scalars, params, d = self._fast
        xi, co, dy = inputs
        if xi is not x_init or type(co) is not type(cost) or dy is not dynamics or co.C is not Cc or co.c is not cc or \
                self.u_init is not None:      # (the cost OBJECT may be a new one around the same tensors: QuadCost(C, c) per call)
            return None
        if xi.data_ptr() != p_x or Cc.data_ptr() != p_C or cc.data_ptr() != p_c or _lib.stream_ptr(d) != stream:
            return None
        if scalars != (float(self.eps), int(self.not_improved_lim), float(self.ls_decay), int(self.max_ls_iter),
                       float(self.best_cost_eps), int(self.max_iter), 1 if self.batch_coupled else 0):
            return None
        if self.verbose or self.ilqr_verbose or not self.graph or not self.device_loop or torch.cuda.is_current_stream_capturing():
            return None
        if torch.is_grad_enabled() and (xi.requires_grad or Cc.requires_grad or cc.requires_grad or (
                isinstance(dy, LinDx) and (dy.F.requires_grad or (dy.f is not None and dy.f.requires_grad)))):
            return None
        if params is not None:
            g_, m_, l_ = dy.host_params()
            if params != tuple((ctypes.c_float * 6)(g_, m_, l_, float(dy.dt), float(dy.max_torque), 1.0 if dy.clamp_grad_closed else 0.0)):
                return None
        for other in list(_UNRESOLVED):       # earlier solves whose read-back was deferred: their asserts come now
            other._resolve()
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        g, g_both, n_f, _ws, g_host, ev = entry
        g.replay()
        lazy = self.lazy_status
        ev.record()
        both = g_both.clone()
        out, ints = both[:n_f], both[n_f:].view(torch.int32)
        n_x, n_u = T * B * nx, T * B * nu
        self.__dict__["_loop_flag"] = ints[B + 7:B + 8]
        self.__dict__["_warn_unconverged"] = False
        self.__dict__["_best_norm_max"] = None
        if lazy:
            self.__dict__["_pending"] = ((g_host, ev), ints[:B])
        else:
            while not ev.query():
                pass
            self.__dict__["_pending"] = (g_host.tolist(), ints[:B])
        _UNRESOLVED.append(self)
        if not lazy:
            self._resolve()
        if self.detach_unconverged:              # (nothing to detach without a graph; the reference's warning stays, :263-273)
            if lazy:
                self.__dict__["_warn_unconverged"] = True
            elif self._best_norm_max:
                self._warn()
        return out[:n_x].view(T, B, nx), out[n_x:n_x + n_u].view(T, B, nu), out[n_x + n_u:n_x + n_u + B]

    def forward(self, inputs):
        if self._fast is not None:
            again = self._replay(inputs)
            if again is not None:
                return again
        x_init, cost, dynamics = inputs
        x_init = _as_tensor(x_init)
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        assert list(x_init.shape) == [B, nx], " x_init dim mismatch"
        dev, dt = x_init.device, x_init.dtype
        key = (dev, dt)
        if self._bounds_on is None or self._bounds_on[0] != key:   # the bounds travel to the device once, not per call
            self._bounds_on = (key, self.u_lower.to(device=dev, dtype=dt), self.u_upper.to(device=dev, dtype=dt))
        lo, hi = self._bounds_on[1], self._bounds_on[2]
        if self.u_init is None:
            if self._u_zero is None or self._u_zero[0] != key:   # read-only for every consumer below: made once
                self._u_zero = (key, torch.zeros((T, B, nu), dtype=dt, device=dev))
            u = self._u_zero[1]
        else:
            u = _as_tensor(self.u_init).to(device=dev, dtype=dt)
            if list(u.shape) == [T, nu]:
                u = u.unsqueeze(1).repeat(1, B, 1)
        assert list(u.shape) == [T, B, nu], "u dim mismatch, actual" + str(tuple(u.shape))
        u = u.detach()

        def models(x, u):
            if isinstance(dynamics, LinDx):
                Fm, fm = dynamics.F, dynamics.f
            else:
                Fm, fm = linearize_dynamics(x, u, dynamics)
            if isinstance(cost, QuadCost):
                Cm, cm = cost.C, cost.c
            else:
                Cm, cm, _ = approximate_cost(x, u, cost)
            return Cm, cm, Fm, fm

        def detached_dyn():
            if isinstance(dynamics, LinDx):
                return LinDx(dynamics.F.detach(), None if dynamics.f is None else dynamics.f.detach())
            return dynamics

        def detached_cost():
            if isinstance(cost, QuadCost):
                return QuadCost(cost.C.detach(), cost.c.detach())
            return cost

        if self.verbose:
            with torch.no_grad():
                c0 = get_cost(T, u, detached_cost(), detached_dyn(), x_init=x_init.detach())
            print('Initial mean(cost): {:.4e}'.format(float(c0.mean())))
        best = None
        n_not_improved = 0
        for_out = None
        last_norm = None
        self.__dict__["_best_norm_max"] = None
        if self.device_loop and not self.verbose and not self.ilqr_verbose and isinstance(cost, QuadCost):
            looped = self._device_loop(x_init, cost, dynamics, u, lo, hi)
            if looped is not None:
                best, last_norm = looped
        for i in range(self.max_iter if best is None else 0):
            with torch.no_grad():
                if hasattr(dynamics, "fused_ok") and dynamics.fused_ok(x_init, u) and isinstance(cost, QuadCost):
                    # pendulum: rollout and linearisation in one launch (get_traj + linearize_dynamics, :123-136)
                    x, Fm, fm = dynamics.rollout_linearize(x_init.detach(), u)
                    Cm, cm = cost.C, cost.c
                else:
                    x = self._nominal(T, u, x_init.detach(), detached_dyn())
                    Cm, cm, Fm, fm = models(x, u)
                step = MPCstep(controls=u, T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=nx, n_ctrl=nu,
                               current_states=x, true_cost=detached_cost(), true_dynamics=detached_dyn(),
                               ls_decay=self.ls_decay, max_ls_iter=self.max_ls_iter, verbose=self.ilqr_verbose,
                               need_expand=True, batch_coupled=self.batch_coupled)
                x, u = step.forward((x[0], Cm, cm, Fm, fm))
            back_out, for_out = step.back_out, step.for_out
            n_not_improved += 1
            if best is None:
                best = {'x': x.clone(), 'u': u.clone(), 'costs': for_out.costs.clone(),
                        'full_du_norm': for_out.full_du_norm.clone()}
            else:   # per-sample best (:200-209)
                better = for_out.costs <= best['costs'] + self.best_cost_eps
                if bool(better.any()):
                    n_not_improved = 0
                best['x'] = torch.where(better[None, :, None], x, best['x'])
                best['u'] = torch.where(better[None, :, None], u, best['u'])
                best['costs'] = torch.where(better, for_out.costs, best['costs'])
                best['full_du_norm'] = torch.where(better, for_out.full_du_norm, best['full_du_norm'])
            if self.verbose:
                table_log('lqr', (('iter', i), ('mean(cost)', best['costs'].mean(), '{:.4e}'),
                                  ('||full_du||_max', for_out.full_du_norm.max(), '{:.2e}'),
                                  ('mean(alphas)', for_out.mean_alphas, '{:.2e}'),
                                  ('total_qp_iters', back_out.n_total_qp_iter)))
            self.n_iter = i + 1
            if float(for_out.full_du_norm.max()) < self.eps:       # :223-230
                self._say("Converged")
                break
            if n_not_improved > self.not_improved_lim:
                self._say("Not improved lim")
                break
            if i == self.max_iter - 1:
                self._say("Not Converged ")
        x, u = best['x'], best['u']
        costs = best['costs']
        # lazy_status: the loop has not been read back (device loop only).  While a hipGraph is being captured nothing has run
        # and nothing may be read back: host decisions (the non-convergence warning) are skipped, the device flags decide
        deferred = self._pending is not None or (x.is_cuda and torch.cuda.is_current_stream_capturing())

        def unconverged():
            """some trajectory's best full_du_norm is above eps (:263) - a host decision: resolves a deferred read-back"""
            self._resolve()
            if last_norm is not None and self._best_norm_max is not None:
                return self._best_norm_max
            return float(best['full_du_norm'].max()) > self.eps

        fused = hasattr(dynamics, "fused_ok") and dynamics.fused_ok(x[0], u) and isinstance(cost, QuadCost)
        # Can anything upstream receive a gradient?  Decidable before the Taylor models are built when both models are
        # plain tensors: then, with nothing to differentiate, the no-op node and the detach blend below (which returns
        # its input unchanged when there is no graph) are skipped - no launches after the loop's one read-back.
        leaves = None
        tiled = isinstance(cost, TiledQuadCost) and not self.update_dynamics    # the gradient goes to (Q, p) alone
        if isinstance(cost, QuadCost) and (fused or isinstance(dynamics, LinDx)):
            leaves = [x_init] + ([cost.Q, cost.p] if isinstance(cost, TiledQuadCost) else [cost.C, cost.c]) + \
                ([dynamics.F, dynamics.f] if isinstance(dynamics, LinDx) else [])
        if not torch.is_grad_enabled() or (leaves is not None and not any(
                isinstance(t, torch.Tensor) and t.requires_grad for t in leaves)):
            if self.detach_unconverged:
                if deferred:
                    self.__dict__["_warn_unconverged"] = True          # (the warning comes with the read-back)
                elif unconverged():
                    self._warn()
            return x, u, costs
        # Taylor models at the best point and a no-op MPCstep node that carries the gradient (:234-259)
        if fused:
            _, Fm, fm = dynamics.rollout_linearize(x[0], u)   # constants of the graph: the pendulum is not learnt
            Cm, cm = cost.C, cost.c
        else:
            Cm, cm, Fm, fm = models(x, u)
        if self.update_dynamics:
            Cm, cm = Cm.detach(), cm.detach()
        else:
            Fm = Fm.detach()
            fm = None if fm is None else fm.detach()
        if tiled and last_norm is not None and x.is_cuda and (cost.Q.requires_grad or cost.p.requires_grad or
                                                              x_init.requires_grad):
            # One (Q, p) tiled over time and batch, solved by the device loop: the node's inputs are the un-tiled tensors,
            # its backward returns the gradient summed inside the co-state kernel (no [T,B,ns,ns] gradient, no reduction
            # launches after it), and the detach mask of :263-289 gates the incoming gradient there from the loop's own
            # device flags - the same gradients as blending x, u with their detached copies, without a host decision
            d = x.device
            retained = dict(C=_lib.f32c(cost.C, d), c=_lib.f32c(cost.c, d), F=_lib.f32c(Fm, d), x=_lib.f32c(x, d),
                            u=_lib.f32c(u, d))
            detach = (_lib.f32c(last_norm, d), self._loop_flag, self.eps) if self.detach_unconverged else None
            spec = (T, B, nx, nu, d, _lib.f32c(lo, d), _lib.f32c(hi, d), detach)
            x, u = _MPCstepTiledFn.apply(x[0].detach(), cost.Q, cost.p, spec, retained, x, u)
            if self.detach_unconverged:
                if deferred:
                    self.__dict__["_warn_unconverged"] = True
                elif unconverged():
                    self._warn()
            return x, u, costs
        node = MPCstep(controls=u, T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=nx, n_ctrl=nu, current_states=x,
                       true_cost=detached_cost(), true_dynamics=detached_dyn(), ls_decay=self.ls_decay,
                       max_ls_iter=self.max_ls_iter, verbose=self.ilqr_verbose, need_expand=True, no_op_forward=True)
        if isinstance(cost, TiledQuadCost) and not self.update_dynamics:
            # host loop / CPU tensors / a refused device loop: tile (Q, p) ON the graph before asking whether anything can
            # receive a gradient - TiledQuadCost.C / .c are detached copies, the learnable tensors are Q and p
            Cm = cost.Q[None, None].expand(T, B, -1, -1)
            cm = cost.p[None, None].expand(T, B, -1)
        needs_graph = any(isinstance(t, torch.Tensor) and t.requires_grad for t in (Cm, cm, Fm, fm, x_init))
        if needs_graph:
            x, u = node.apply((x[0].detach(), Cm, cm, Fm, fm))
        if self.detach_unconverged and deferred and x.is_cuda and torch.cuda.is_current_stream_capturing():
            # a hipGraph is being captured on this (generic) path: the read-back below would end the capture with an error deep
            # inside the runtime - say what is and is not capturable instead
            raise RuntimeError("BoxDDP: this solve cannot be captured in a hipGraph with detach_unconverged=True (the gradient "
                               "path of a cost that is not a TiledQuadCost, or the host loop, decides the detach mask on the "
                               "host); capture with detach_unconverged=False, or use a TiledQuadCost on the device loop")
        if self.detach_unconverged and unconverged():                                      # :263-289
            self._warn()
            if last_norm is None:
                last_norm = for_out.full_du_norm
            keep = (last_norm < self.eps).to(x.dtype)[None, :, None]
            x = x * keep + x.detach() * (1. - keep)
            u = u * keep + u.detach() * (1. - keep)
        return x, u, costs
