"""`PNQP` - projected-Newton box QP with the call signature of mpc/pnqp.py:37 of the reference,
computed by the HIP kernel behind `dmpc_pnqp`.

    min 1/2 x'Hx + q'x   s.t. lower <= x <= upper        (Tassa et al. 2014, Algorithm 1)

Returns the reference's 4-tuple `(x, H_f | (LU, pivots), Index_f, i)`.

Termination: the reference reduces its |dx| < 1e-4 test and its Armijo loop over the whole batch
(pnqp.py:139-144, 172, 187), so there a row's result depends on which other rows share its batch.
  * `batch_coupled=False` (default, what shards across GPUs): each row stops on its own tests = the reference
    called with a batch of one per row; `i` is the largest per-row iteration index, `PNQP.last_info` keeps all;
  * `batch_coupled=True`: the reference's batch semantics (one grid-wide reduction per decision, cooperative launch;
    the batch must fit one launch).
"""
import warnings

import torch

from . import _lib
from .lqr_recursion import _as_tensor, _device_of, _workspace

GAMMA = 0.1  # pnqp.py:23


def calc_obj(H, q, x):
    """1/2 x'Hx + q'x per batch row (pnqp.py:26-33)"""
    return 0.5 * torch.einsum("bi,bij,bj->b", x, H, x) + (q * x).sum(dim=1)


def pnqp_device(H, q, lower, upper, x_init, n_iter, info=None, batch_coupled=False):
    """raw call on float32 device tensors -> (x, fac, piv, index_f, iters)"""
    lib = _lib.load()
    _lib.require_gpu()
    dev = H.device
    B, n = q.shape
    x = torch.empty((B, n), dtype=torch.float32, device=dev)
    fac = torch.empty((B, n, n), dtype=torch.float32, device=dev)
    piv = torch.empty((B, n), dtype=torch.int32, device=dev)
    idx_f = torch.empty((B, n), dtype=torch.float32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    ws, need = None, 0
    if batch_coupled:
        need = lib.dmpc_pnqp_workspace_bytes(B, n, int(n_iter), 1)     # (decision slots + the fixed-grid form's rows: any n, any B)
        ws = _workspace(need, dev)
    with _lib.guard(dev):
        rc = lib.dmpc_pnqp(B, n, _lib.ptr(H), _lib.ptr(q), _lib.ptr(lower), _lib.ptr(upper), _lib.ptr(x_init),
                           int(n_iter), 1 if batch_coupled else 0, _lib.ptr(x), _lib.ptr(fac), _lib.ptr(piv),
                           _lib.ptr(idx_f), _lib.ptr(iters), _lib.ptr(ws), need, _lib.ptr(info), _lib.stream_ptr(dev))
    _lib.check(rc, "dmpc_pnqp")
    return x, fac, piv, idx_f, iters


def PNQP(H, q, lower, upper, x_init=None, n_iter=20, batch_coupled=False):
    H, q, lower, upper, x_init = (_as_tensor(t) for t in (H, q, lower, upper, x_init))
    n_batch, n_dim = H.shape[0], H.shape[1]
    assert bool((lower <= upper).all()), " lower is larger than upper"
    assert list(H.shape) == [n_batch, n_dim, n_dim], "H dim mismatch"
    assert list(q.shape) == [n_batch, n_dim], "q dim mismatch expected" + str([n_batch, n_dim])
    assert list(lower.shape) == [n_batch, n_dim], "lower dim mismatch actual" + str(tuple(lower.shape))
    assert list(upper.shape) == [n_batch, n_dim], "upper dim mismatch"
    dev = _device_of(H, q)
    out_dev, out_dtype = H.device, (H.dtype if H.dtype.is_floating_point else torch.float32)
    d = [_lib.f32c(t, dev) for t in (H, q, lower, upper, x_init)]
    info = torch.zeros(n_batch, dtype=torch.int32, device=dev)
    x, fac, piv, idx_f, iters = pnqp_device(d[0], d[1], d[2], d[3], d[4], n_iter, info, batch_coupled)
    PNQP.last_info = dict(iters=iters, info=info)
    if bool(((info & _lib.INFO_QP_ITERCAP) != 0).any()):
        warnings.warn("Projected Newton Quadratic Programming warning: Did not converge")   # pnqp.py:192
    i = int(iters.max().item())
    conv = lambda t: t.to(device=out_dev, dtype=out_dtype)  # noqa: E731
    if n_dim == 1:
        return conv(x), conv(fac), conv(idx_f), i
    return conv(x), (conv(fac), piv.to(out_dev)), conv(idx_f), i


PNQP.last_info = None
