"""Host-side mirror of the reference's util.py on torch tensors.

Names and argument meaning follow /root/reference/util.py: QuadCost/LinDx (:25-32), clamp
(:101-123), get_cost (:126-198), get_traj (:201-277), bmv (:280-298), bger (:301-329), bquad
(:332-358), expand_time_batch / expand_batch (:361-408), bdot (:411-434), batch LU factor/solve
(:462-528).  The tiny batched helpers are plain torch ops (inside the solver kernels they are
device functions); the LU pair calls the HIP library.
"""
from collections import namedtuple

import torch

from . import _lib

QuadCost = namedtuple("QuadCost", "C c", defaults=(None, None))
LinDx = namedtuple("LinDx", "F f", defaults=(None, None))


class TiledQuadCost(QuadCost):
    """A `QuadCost` that is ONE (Q [ns,ns], p [ns]) tiled over time and batch - the imitation loop's learnable cost
    (env_dx/il_env.py:119-129 repeats Q and p to [T,B,ns,ns], [T,B,ns]).  `.C` / `.c` are those dense tensors (what the
    kernels read; constants of the graph), `.Q` / `.p` the tensors a gradient flows to: `BoxDDP` then asks
    `MPCstep.backward` for the gradient already summed over time and batch (formed in the co-state kernel) instead of
    letting autograd reduce dC [T,B,ns,ns] and dc [T,B,ns] afterwards.  Everything that takes a `QuadCost` takes it."""

    def __new__(cls, Q, p, T, n_batch):
        C = Q.detach()[None, None].expand(T, n_batch, -1, -1).contiguous()
        c = p.detach()[None, None].expand(T, n_batch, -1).contiguous()
        self = super().__new__(cls, C, c)
        self.Q, self.p = Q, p
        return self


def bmv(a, x):
    assert a.shape[0] == x.shape[0], "batch mismatch"
    assert a.shape[2] == x.shape[1], "mat mul dim mismatch"
    assert x.dim() == 2, " x is not batch vector"
    return torch.matmul(a, x.unsqueeze(2)).squeeze(2)


def bger(x, y):
    return x.unsqueeze(2) @ y.unsqueeze(1)


def bquad(x, Q):
    assert x.shape[0] == Q.shape[0], "batch mismatch"
    assert x.shape[1] == Q.shape[1], "mat mul dim mismatch"
    assert Q.shape[2] == Q.shape[1], "Q is not square matrix"
    return (x.unsqueeze(1) @ Q @ x.unsqueeze(2)).squeeze(1).squeeze(1)


def bdot(x, y):
    assert x.shape == y.shape
    return (x.unsqueeze(1) @ y.unsqueeze(2)).squeeze(1).squeeze(1)


def clamp(x, lower, upper):
    assert x.shape == lower.shape
    assert x.shape == upper.shape
    assert bool((lower <= upper).all()), " lower is larger than upper"
    return torch.minimum(torch.maximum(x, lower), upper)


def expand_time_batch(m, time, n_batch):
    return m.unsqueeze(0).unsqueeze(0).expand((time, n_batch) + tuple(m.shape))


def expand_batch(m, n_batch):
    return m.unsqueeze(0).expand((n_batch,) + tuple(m.shape))


def get_traj(T, u, x_init, dynamics):
    """roll a control sequence out; dynamics is a LinDx or a callable (x, u) -> x'"""
    xs = [x_init]
    for t in range(T - 1):
        if isinstance(dynamics, LinDx):
            xu = torch.cat((xs[t], u[t]), dim=1)
            nx = bmv(dynamics.F[t], xu)
            if dynamics.f is not None:
                nx = nx + dynamics.f[t]
        else:
            nx = dynamics(xs[t], u[t])
        xs.append(nx)
    return torch.stack(xs, dim=0)


def get_cost(T, u, cost, dynamics=None, x_init=None, x=None):
    """sum_t 1/2 tau' C_t tau + c_t' tau (QuadCost) or sum_t cost(tau) -> [B]"""
    assert x_init is not None or x is not None
    if x is None:
        x = get_traj(T, u, x_init, dynamics)
    tau = torch.cat((x, u), dim=2)
    if isinstance(cost, QuadCost):
        quad = torch.einsum("tbi,tbij,tbj->b", tau, cost.C, tau)
        return 0.5 * quad + (tau * cost.c).sum(dim=(0, 2))
    return torch.stack([cost(tau[t]) for t in range(T)], dim=0).sum(dim=0)


def batch_lu_factor(A):
    """[B,n,n] -> (LU [B,n,n], pivots [B,n] int32, 1-based) - LAPACK getrf layout (util.py:462-482)"""
    assert A.dim() == 3 and A.shape[1] == A.shape[2], "Actual" + str(tuple(A.shape))
    _lib.require_gpu()
    lib = _lib.load()
    dev = A.device if A.is_cuda else torch.device("cuda", torch.cuda.current_device())
    a = _lib.f32c(A, dev)
    B, n, _ = a.shape
    LU = torch.empty_like(a)
    piv = torch.empty((B, n), dtype=torch.int32, device=dev)
    with _lib.guard(dev):
        _lib.check(lib.dmpc_batch_lu_factor(B, n, _lib.ptr(a), _lib.ptr(LU), _lib.ptr(piv), None,
                                            _lib.stream_ptr(dev)), "dmpc_batch_lu_factor")
    return LU.to(A.device), piv.to(A.device)


def batch_lu_solve(lu_and_piv, b):
    """float32 solve; b is [B,n] or [B,n,k] (util.py:505-528)"""
    LU, piv = lu_and_piv
    _lib.require_gpu()
    lib = _lib.load()
    dev = LU.device if LU.is_cuda else torch.device("cuda", torch.cuda.current_device())
    lu = _lib.f32c(LU, dev)
    pv = piv.to(device=dev, dtype=torch.int32).contiguous()
    vec = b.dim() == 2
    rhs = _lib.f32c(b.unsqueeze(2) if vec else b, dev)
    B, n, k = rhs.shape
    x = torch.empty_like(rhs)
    with _lib.guard(dev):
        _lib.check(lib.dmpc_batch_lu_solve(B, n, k, _lib.ptr(lu), _lib.ptr(pv), _lib.ptr(rhs), _lib.ptr(x),
                                           _lib.stream_ptr(dev)), "dmpc_batch_lu_solve")
    x = x.squeeze(2) if vec else x
    return x.to(b.device)
