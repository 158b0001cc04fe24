"""chainer_differentiable_mpc_amd - MI355X-native differentiable-MPC inner solver.

Drop-in for the hot path of pfnet-research/chainer-differentiable-mpc: the batched LQR Riccati
backward/forward sweep, its analytic KKT gradient and the projected-Newton box QP run as
hand-written gfx950 HIP kernels behind the reference's own call signatures
(`LqrRecursion`, `DiffLqr`, `PNQP`, `MPCstep`, `LQR_active`, ...).  "Variable" in the reference
becomes `torch.Tensor` here.  Kernels are bound through a ctypes C-ABI (include/dmpc.h);
there is no CPU fallback.
"""
from . import synthetic  # noqa: F401
from ._lib import DmpcError, load as load_library  # noqa: F401
from .util import (LinDx, QuadCost, TiledQuadCost, batch_lu_factor, batch_lu_solve, bdot, bger, bmv, bquad,  # noqa: F401
                   clamp, expand_batch, expand_time_batch, get_cost, get_traj)
from .lqr_recursion import LqrRecursion  # noqa: F401
from .differentiable_lqr import DiffLqr, LqrNet, LqrNet_cost_dx  # noqa: F401
from .pnqp import PNQP  # noqa: F401
from .active_constrained_lqr import LQR_active  # noqa: F401
from .mpc_step import MPCstep, LqrBackOut, LqrForOut  # noqa: F401
from .approximate import approximate_cost, linearize_dynamics  # noqa: F401
from .box_ddp import BoxDDP  # noqa: F401
from .mpc_net import MpcNet_cost, MpcNet_dx  # noqa: F401
from .pendulum import PendulumDx  # noqa: F401
from .il_env import IL_Env, Pendulum_Net_cost_logit  # noqa: F401
from . import make_dataset  # noqa: F401

__version__ = "0.1.0"
