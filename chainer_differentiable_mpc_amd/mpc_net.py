"""`MpcNet_dx`, `MpcNet_cost` - learnable-dynamics MPC layers with the constructors and `forward` of
mpc/mpc_net.py:20-227 of the reference (where `MpcNet_dx` is defined twice and `MpcNet_cost` has the same
body).  Thin host wrappers over `BoxDDP`."""
import numpy as np
import torch

from .box_ddp import BoxDDP
from .lqr_recursion import _as_tensor
from .util import LinDx, expand_time_batch


class MpcNet_dx(torch.nn.Module):
    """MPC network whose linear dynamics [A|B] are learnable (mpc_net.py:20-87)."""

    def __init__(self, T, u_lower, u_upper, n_batch, n_state, n_ctrl, seed, u_init, eps=1e-5, not_improved_lim=5,
                 line_search_decay=0.2, max_line_search_iter=10, best_cost_eps=1e-4, max_iter=10,
                 verbose=False, ilqr_verbose=False, dtype=torch.float64, quiet=False):
        super().__init__()
        self.u_lower, self.u_upper = _as_tensor(u_lower), _as_tensor(u_upper)
        assert bool((self.u_lower <= self.u_upper).all()), " lower is larger than upper"
        self.T, self.n_batch, self.n_state, self.n_ctrl = T, n_batch, n_state, n_ctrl
        self.n_sc = n_ctrl + n_state
        assert list(self.u_lower.shape) == [T, n_batch, n_ctrl], 'actual' + str(tuple(self.u_lower.shape))
        assert list(self.u_upper.shape) == [T, n_batch, n_ctrl]
        np.random.seed(seed)                     # same draws as the reference (:59-64)
        alpha = 0.2
        A = np.eye(n_state) + alpha * np.random.randn(n_state, n_state)
        B = np.random.randn(n_state, n_ctrl)
        self.A = torch.nn.Parameter(torch.as_tensor(A, dtype=dtype))
        self.B = torch.nn.Parameter(torch.as_tensor(B, dtype=dtype))
        self.mpc_layer = BoxDDP(T=T, u_lower=self.u_lower, u_upper=self.u_upper, n_batch=n_batch, n_state=n_state,
                                n_ctrl=n_ctrl, u_init=u_init, eps=eps, not_improved_lim=not_improved_lim,
                                line_search_decay=line_search_decay, max_line_search_iter=max_line_search_iter,
                                best_cost_eps=best_cost_eps, max_iter=max_iter, verbose=verbose,
                                ilqr_verbose=ilqr_verbose, quiet=quiet)

    def forward(self, inputs):
        x_init, cost = inputs
        ab_cat = torch.cat((self.A, self.B), dim=1)
        large_f_learner = expand_time_batch(ab_cat, self.T - 1, self.n_batch)
        f = torch.zeros((self.T - 1, self.n_batch, self.n_state), dtype=ab_cat.dtype, device=ab_cat.device)
        return self.mpc_layer((x_init, cost, LinDx(large_f_learner, f)))


class MpcNet_cost(MpcNet_dx):
    """mpc_net.py:160-227 - identical body in the reference."""
