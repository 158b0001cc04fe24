"""`LqrRecursion` - same constructor and methods as lqr/lqr_recursion.py:18-209 of the reference,
computed by the fused gfx950 kernel behind `dmpc_lqr_solve` (include/dmpc.h).

Inputs may be torch tensors (any device / float dtype) or numpy arrays; arithmetic is float32 on
the GPU; outputs are torch tensors with the dtype and device of `C`.
"""
import numpy as np
import collections

import torch

from . import _lib


def _as_tensor(v):
    if v is None or isinstance(v, torch.Tensor):
        return v
    return torch.as_tensor(np.asarray(v))


def _device_of(*ts):
    for t in ts:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def raise_info(info, what):
    """Turn the kernels' per-trajectory flags into the reference's behaviour where it has one:
    MPCstep asserts on NaN/Inf (mpc_step.py:133-135,161-162,211-223).  Synchronises (reads `info`)."""
    if info is None or info.numel() == 0:
        return 0
    flags = int(np.bitwise_or.reduce(info.cpu().numpy()))
    if flags & _lib.INFO_NONFINITE:
        n_bad = int(((info & _lib.INFO_NONFINITE) != 0).sum().item())
        raise AssertionError("%s: NaN/Inf in the solution of %d trajectories" % (what, n_bad))
    return flags


class LqrRecursion:
    """LQR Recursion solver (time-varying, batched).  lqr/lqr_recursion.py:18."""

    def __init__(self, x_init, C, c, large_f, f, T, n_state, n_ctrl, u_zero_Index=None, precision="float32"):
        # precision="float64" (not in the reference's signature): solve_recursion() on the float64 kernels - the reference's
        # own precision, an order of magnitude slower (include/dmpc.h, `_f64`)
        assert precision in ("float32", "float64")
        self.precision = precision
        self.x_init = _as_tensor(x_init)
        self.C = _as_tensor(C)
        self.c = _as_tensor(c)
        self.F = _as_tensor(large_f)
        self.f = _as_tensor(f)
        self.T = int(T)
        self.n_batch = self.C.shape[1]
        self.n_state = int(n_state)
        self.n_ctrl = int(n_ctrl)
        self.n_sc = self.n_state + self.n_ctrl
        self.u_zero_Index = _as_tensor(u_zero_Index)
        # the reference's contracts (lqr_recursion.py:51-66); F is deliberately unchecked there
        # beyond what indexing needs: both [T-1,...] and [T,...] are accepted, F[t], t < T-1 is read
        assert list(self.x_init.shape) == [self.n_batch, self.n_state]
        assert list(self.C.shape) == [self.T, self.n_batch, self.n_sc, self.n_sc], "C dim mismatch"
        assert list(self.c.shape) == [self.T, self.n_batch, self.n_sc], \
            str(tuple(self.c.shape)) + " c dim mismatch: expected " + str([self.T, self.n_batch, self.n_sc])
        if self.T > 1:
            assert self.F.shape[0] in (self.T - 1, self.T) and \
                list(self.F.shape[1:]) == [self.n_batch, self.n_state, self.n_sc], "F dim mismatch"
        if self.f is not None:
            assert list(self.f.shape) == [self.T - 1, self.n_batch, self.n_state], " f dim mismatch"
        if self.u_zero_Index is not None:
            assert list(self.u_zero_Index.shape) == [self.T, self.n_batch, self.n_ctrl]
        self._out_dtype = self.C.dtype if self.C.dtype.is_floating_point else torch.float32
        self._out_device = self.C.device
        self._dev = _device_of(self.C, self.c, self.F, self.x_init)
        self.info = None

    # -- marshalling -------------------------------------------------------------------------
    def _dev_inputs(self):
        d = self._dev
        mask = None
        if self.u_zero_Index is not None:
            mask = self.u_zero_Index.to(device=d).to(torch.uint8).contiguous()
        return (_lib.f32c(self.C, d), _lib.f32c(self.c, d), _lib.f32c(self.F, d), _lib.f32c(self.f, d),
                _lib.f32c(self.x_init, d), mask)

    def _out(self, t):
        return t.to(device=self._out_device, dtype=self._out_dtype)

    def _new_info(self):
        self.info = torch.zeros(self.n_batch, dtype=torch.int32, device=self._dev)
        return self.info

    # -- reference API -----------------------------------------------------------------------
    def backward(self):
        """Riccati backward recursion -> (Ks, ks): lists of T per-step gains [B,nu,nx], [B,nu]
        in forward time order (lqr_recursion.py:69-158)."""
        lib = _lib.load()
        _lib.require_gpu()
        C, c, F, f, _, mask = self._dev_inputs()
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        Ks = torch.empty((T, B, nu, nx), dtype=torch.float32, device=self._dev)
        ks = torch.empty((T, B, nu), dtype=torch.float32, device=self._dev)
        info = self._new_info()
        ws, need = None, 0
        if lib.dmpc_lqr_kernel_family(nx, nu) == 5:     # beyond a wavefront's 64 columns: the sweep's matrices live in a workspace
            need = lib.dmpc_lqr_workspace_bytes(T, B, nx, nu)
            ws = _workspace(need, self._dev)
        with _lib.guard(self._dev):
            _lib.check(lib.dmpc_lqr_backward_sweep_ws(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F),
                                                      _lib.ptr(f), _lib.ptr(mask), _lib.ptr(Ks), _lib.ptr(ks),
                                                      _lib.ptr(ws), need, _lib.ptr(info), _lib.stream_ptr(self._dev)),
                       "dmpc_lqr_backward_sweep")
        Ks, ks = self._out(Ks), self._out(ks)
        return [Ks[t] for t in range(T)], [ks[t] for t in range(T)]

    def forward(self, Ks, ks):
        """closed-loop rollout with the given gains -> (x [T,B,nx], u [T,B,nu]) (lqr_recursion.py:160-200)"""
        assert len(Ks) == self.T, "Ks length error"
        lib = _lib.load()
        _lib.require_gpu()
        _, _, F, f, x0, mask = self._dev_inputs()
        T, B, nx, nu = self.T, self.n_batch, self.n_state, self.n_ctrl
        Kd = _lib.f32c(torch.stack(list(Ks), dim=0) if not isinstance(Ks, torch.Tensor) else Ks, self._dev)
        kd = _lib.f32c(torch.stack(list(ks), dim=0) if not isinstance(ks, torch.Tensor) else ks, self._dev)
        assert list(Kd.shape) == [T, B, nu, nx], "Kt dim mismatch"
        assert list(kd.shape) == [T, B, nu], "kt dim mismatch"
        x = torch.empty((T, B, nx), dtype=torch.float32, device=self._dev)
        u = torch.empty((T, B, nu), dtype=torch.float32, device=self._dev)
        info = self._new_info()
        with _lib.guard(self._dev):
            _lib.check(lib.dmpc_lqr_forward_sweep(T, B, nx, nu, _lib.ptr(Kd), _lib.ptr(kd), _lib.ptr(F),
                                                  _lib.ptr(f), _lib.ptr(x0), _lib.ptr(mask), _lib.ptr(x),
                                                  _lib.ptr(u), _lib.ptr(info), _lib.stream_ptr(self._dev)),
                       "dmpc_lqr_forward_sweep")
        return self._out(x), self._out(u)

    def solve_recursion(self):
        """backward + forward in ONE fused launch -> (x, u) (lqr_recursion.py:202-209)"""
        if self.precision == "float64":
            d = self._dev
            f64 = lambda t: None if t is None else t.to(device=d, dtype=torch.float64).contiguous()
            mask = None if self.u_zero_Index is None else self.u_zero_Index.to(device=d).to(torch.uint8).contiguous()
            x, u, _, _ = solve_device_f64(f64(self.C), f64(self.c), f64(self.F), f64(self.f), f64(self.x_init), mask,
                                          self.T, self.n_state, self.n_ctrl, info=self._new_info())
            return self._out(x), self._out(u)
        x, u, _, _ = solve_device(*self._dev_inputs(), self.T, self.n_state, self.n_ctrl,
                                  info=self._new_info())
        return self._out(x), self._out(u)


_ws_cache = collections.OrderedDict()
_WS_CACHE_PER_DEVICE = 4   # scratch buffers kept per device (least recently used first out): programs with short-lived streams


def _workspace(nbytes, device):
    """grow-only scratch per (device, stream) - the C library allocates nothing itself.  Calls enqueued on one stream
    are ordered, so they may share a buffer; calls on different streams get different buffers.  A buffer that is
    replaced by a larger one, or dropped as the least recently used OF ITS DEVICE (a process driving eight GPUs keeps
    eight working sets, not four buffers in all), goes back to torch's caching allocator, which keeps it tied to the stream
    it was allocated on (the same one), so kernels still in flight on it are safe.  The key is the raw stream handle: a
    handle that torch hands out again after the stream object died names a stream on which the earlier work has been
    submitted in order, which is all the sharing rule needs."""
    stream = _lib.stream_ptr(device) if device.type == "cuda" else 0
    key = (device.type, device.index, stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
        mine = [k for k in _ws_cache if k[:2] == key[:2]]
        for k in mine[:max(0, len(mine) - _WS_CACHE_PER_DEVICE)]:
            del _ws_cache[k]
    _ws_cache.move_to_end(key)
    return ws


def saving_solve_available(T, B, n_state, n_ctrl):
    """does `solve_saving_device` (and with it the saved-gains gradient) serve this size?"""
    lib = _lib.load()
    return bool(lib.dmpc_lqr_saving_available(T, B, n_state, n_ctrl))


def solve_saving_device(C, c, F, f, x_init, T, n_state, n_ctrl, info=None):
    """The training form of the fused solve (`dmpc_lqr_solve_saving`): (x, u, Ks, ks, Quu, Qxu, Vv) - Vv [T,B,nx,nx+1] the
    value functions [V_t | v_t] - or None where the generated stream does not serve the size (the caller then uses
    `solve_device`)."""
    lib = _lib.load()
    _lib.require_gpu()
    dev = C.device
    B = C.shape[1]
    nx, nu = n_state, n_ctrl
    f32 = dict(dtype=torch.float32, device=dev)
    x = torch.empty((T, B, nx), **f32)
    u = torch.empty((T, B, nu), **f32)
    Ks = torch.empty((T, B, nu, nx), **f32)
    ks = torch.empty((T, B, nu), **f32)
    Quu = torch.empty((T, B, nu, nu), **f32)
    Qxu = torch.empty((T, B, nx, nu), **f32)
    Vv = torch.empty((T, B, nx, nx + 1), **f32)
    with _lib.guard(dev):
        rc = lib.dmpc_lqr_solve_saving(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(f), _lib.ptr(x_init),
                                       _lib.ptr(Ks), _lib.ptr(ks), _lib.ptr(Quu), _lib.ptr(Qxu), _lib.ptr(Vv), _lib.ptr(x),
                                       _lib.ptr(u), _lib.ptr(info), _lib.stream_ptr(dev))
    if rc == _lib.E_UNSUPPORTED:
        return None
    _lib.check(rc, "dmpc_lqr_solve_saving")
    return x, u, Ks, ks, Quu, Qxu, Vv


def saved_solve_device(c, F, Ks, Quu, Qxu, x_init, T, n_state, n_ctrl, info=None):
    """`dmpc_lqr_saved_solve`: the problem of an earlier saving solve with another c (f = 0) and x_init -> (x, u)."""
    lib = _lib.load()
    _lib.require_gpu()
    dev = c.device
    B = c.shape[1]
    x = torch.empty((T, B, n_state), dtype=torch.float32, device=dev)
    u = torch.empty((T, B, n_ctrl), dtype=torch.float32, device=dev)
    with _lib.guard(dev):
        rc = lib.dmpc_lqr_saved_solve(T, B, n_state, n_ctrl, _lib.ptr(c), _lib.ptr(F), _lib.ptr(Ks), _lib.ptr(Quu),
                                      _lib.ptr(Qxu), _lib.ptr(x_init), _lib.ptr(x), _lib.ptr(u), _lib.ptr(info),
                                      _lib.stream_ptr(dev))
    _lib.check(rc, "dmpc_lqr_saved_solve")
    return x, u


_SOLVE_GEOMETRY = {}     # (T, B, nx, nu) -> (workspace bytes, kernel family, wide row kernel?)


def solve_device(C, c, F, f, x_init, mask, T, n_state, n_ctrl, want_gains=False, info=None, out=None):
    """Raw fused solve on float32 device tensors (no copies): the unit the benchmark times.
    Returns (x, u, Ks|None, ks|None)."""
    lib = _lib.load()
    _lib.require_gpu()
    dev = C.device
    B = C.shape[1]
    nx, nu = n_state, n_ctrl
    if out is None:
        x = torch.empty((T, B, nx), dtype=torch.float32, device=dev)
        u = torch.empty((T, B, nu), dtype=torch.float32, device=dev)
    else:
        x, u = out
    Ks = ks = None
    if want_gains:
        Ks = torch.empty((T, B, nu, nx), dtype=torch.float32, device=dev)
        ks = torch.empty((T, B, nu), dtype=torch.float32, device=dev)
    ws = None
    ws_bytes = 0
    # gains stay in LDS unless the horizon is long or the shape runs on the generic kernel
    geo = _SOLVE_GEOMETRY.get((T, B, nx, nu))
    if geo is None:       # (three library queries per problem size, not per call)
        geo = _SOLVE_GEOMETRY[(T, B, nx, nu)] = (lib.dmpc_lqr_workspace_bytes(T, B, nx, nu), lib.dmpc_lqr_kernel_family(nx, nu),
                                                 lib.dmpc_lqr_solve_path(T, B, nx, nu) == 9)
        if len(_SOLVE_GEOMETRY) > 256:
            _SOLVE_GEOMETRY.pop(next(iter(_SOLVE_GEOMETRY)))
    need, family, wide = geo
    per_traj_lds = T * nu * (nx + 1) * 4
    # (family 5 - beyond 64 columns - keeps every trajectory's matrices in the workspace, whoever receives the gains)
    # (... and the wide row kernel - solve path 9 - sends its gain rows through it on their way to the rollout)
    if family == 5 or wide or (not want_gains and (per_traj_lds * 16 > 60 * 1024 or family != 1)):
        ws = _workspace(need, dev)
        ws_bytes = need
    with _lib.guard(dev):
        rc = lib.dmpc_lqr_solve(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(f),
                                _lib.ptr(x_init), _lib.ptr(mask), _lib.ptr(Ks), _lib.ptr(ks), _lib.ptr(x),
                                _lib.ptr(u), _lib.ptr(ws), ws_bytes, _lib.ptr(info), _lib.stream_ptr(dev))
    _lib.check(rc, "dmpc_lqr_solve")
    return x, u, Ks, ks


def solve_device_f64(C, c, F, f, x_init, mask, T, n_state, n_ctrl, want_gains=False, info=None):
    """The fused solve in float64 on float64 device tensors (`dmpc_lqr_solve_f64`: the reference's precision).  Any shape: the
    register-resident column-per-lane kernels for the instantiated ones ((1,1) ... (12,3), (16,4), (16,8), (32,8);
    f64_row_kernels.hpp), one lane per trajectory for every other.  Returns (x, u, Ks|None, ks|None)."""
    lib = _lib.load()
    _lib.require_gpu()
    dev = C.device
    B = C.shape[1]
    nx, nu = n_state, n_ctrl
    for t in (C, c, F, f, x_init):
        assert t is None or (t.dtype == torch.float64 and t.is_contiguous())
    f64 = dict(dtype=torch.float64, device=dev)
    x, u = torch.empty((T, B, nx), **f64), torch.empty((T, B, nu), **f64)
    Ks = ks = None
    if want_gains:
        Ks, ks = torch.empty((T, B, nu, nx), **f64), torch.empty((T, B, nu), **f64)
    need = lib.dmpc_lqr_f64_workspace_bytes(T, B, nx, nu)
    ws = _workspace(need, dev)
    with _lib.guard(dev):
        rc = lib.dmpc_lqr_solve_f64(T, B, nx, nu, _lib.ptr(C), _lib.ptr(c), _lib.ptr(F), _lib.ptr(f), _lib.ptr(x_init),
                                    _lib.ptr(mask), _lib.ptr(Ks), _lib.ptr(ks), _lib.ptr(x), _lib.ptr(u), _lib.ptr(ws), need,
                                    _lib.ptr(info), _lib.stream_ptr(dev))
    _lib.check(rc, "dmpc_lqr_solve_f64")
    return x, u, Ks, ks
