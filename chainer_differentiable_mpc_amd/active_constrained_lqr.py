"""`LQR_active` - LQR with clamped controls masked out of the gains; constructor and methods of
mpc/active_constrained_lqr.py:16-202 of the reference.  Runs the masked variant of the fused HIP
solve kernel (`dmpc_lqr_solve` with `u_zero_mask`)."""
from .lqr_recursion import LqrRecursion


class LQR_active(LqrRecursion):
    """Used by MPCstep.backward (mpc/mpc_step.py:374-376): q_u, Q_ux rows and Q_uu rows/columns of the
    clamped controls are zeroed (1e-8 on that diagonal), V/v are updated from the unmasked blocks and
    the rollout zeroes the clamped controls."""

    def __init__(self, x_init, C, c, large_f, f, T, n_state, n_ctrl, u_zero_Index=None):
        assert u_zero_Index is not None, "LQR_active needs the active-control index"
        super().__init__(x_init, C, c, large_f, f, T, n_state, n_ctrl, u_zero_Index=u_zero_Index)
