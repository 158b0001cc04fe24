"""`IL_Env` and `Pendulum_Net_cost_logit` - host-side callers of the hot path for the imitation-learning experiment
(SURVEY.md 8f rank 3), with the constructors and methods of env_dx/il_env.py:22-213 and env_dx/pendulum_net.py:12-39
of the reference.  Plain torch over `BoxDDP` / `PendulumDx` / `QuadCost`: data set generation with the true cost,
`mpc` / `mpc_Q` with a tiled diagonal / full cost, and the learnable cost q = sigmoid(logit), p = sqrt(q) * learn_p.
The interactive training driver (il_exp.py), its CSV logs and pickles are out of scope."""
import numpy as np
import torch

from .box_ddp import BoxDDP
from .pendulum import PendulumDx
from .util import TiledQuadCost


class IL_Env:
    """Imitation-learning environment (il_env.py:22).  `device` is where the trajectories are computed."""

    def __init__(self, env, lqr_iter=500, mpc_T=20, device="cuda", dtype=torch.float32, quiet=True, lazy_status=False):
        # lazy_status=True (opt-in): `BoxDDP(lazy_status=True)` - a solve returns before the device loop's read-back, so a
        # training loop keeps launching; asserts / NaN checks of a solve then arrive with `flush()`, the next solve, or the
        # first access of `last_solver.status`.  Default: every solve is checked before `mpc` returns, as in the reference.
        assert env == 'pendulum'                                     # il_env.py:35-38
        self.lazy_status = bool(lazy_status)
        self.env = env
        self.true_dx = PendulumDx()
        self.lqr_iter = lqr_iter
        self.mpc_T = mpc_T
        self.device, self.dtype, self.quiet = torch.device(device), dtype, quiet
        self.train_data = self.val_data = self.test_data = None

    # the pickle of env_dx/make_dataset.py holds numpy arrays; keep that format (and no device handles) on disk
    def flush(self):
        """resolve every deferred read-back (asserts, NaN flags, status, the non-convergence warning) - called at the data
        set and pickle boundaries; a training loop with lazy_status=True calls it once per epoch"""
        for solver in getattr(self, "_solvers", {}).values():
            solver._resolve()

    def __getstate__(self):
        self.flush()
        st = dict(self.__dict__)
        st.pop("_solvers", None)
        st.pop("last_solver", None)
        for k in ("train_data", "val_data", "test_data"):
            if isinstance(st[k], torch.Tensor):
                st[k] = st[k].detach().cpu().numpy().astype(np.float64)
        st["device"] = str(st["device"])
        st["dtype"] = str(st["dtype"]).replace("torch.", "")
        return st

    def __setstate__(self, st):
        st = dict(st)
        st["dtype"] = getattr(torch, st["dtype"])
        st["device"] = torch.device("cpu")          # data stays on the host until .to(device)
        for k in ("train_data", "val_data", "test_data"):
            if st[k] is not None:
                st[k] = torch.as_tensor(st[k], dtype=st["dtype"])
        self.__dict__.update(st)

    def to(self, device):
        self.device = torch.device(device)
        for k in ("train_data", "val_data", "test_data"):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(self.device))
        return self

    @staticmethod
    def sample_xinit(n_batch=1):
        """(cos th, sin th, dth), th ~ U(-pi/2, pi/2), dth ~ U(-1, 1): two successive numpy draws (il_env.py:55-69)"""
        th = np.random.rand(n_batch) * np.pi - 0.5 * np.pi
        thdot = np.random.rand(n_batch) * 2.0 - 1.0
        return np.stack((np.cos(th), np.sin(th), thdot), axis=1)

    def populate_data(self, n_train, n_val, n_test, seed=0):
        """expert trajectories [n, T, n_sc] under the true cost, split into train / val / test (il_env.py:71-102)"""
        np.random.seed(seed)
        xinit = self.sample_xinit(n_batch=n_train + n_val + n_test)
        true_q, true_p = self.true_dx.get_true_obj()
        with torch.no_grad():
            x_mpc, u_mpc = self.mpc(self.true_dx, xinit, true_q, true_p, update_dynamics=True)
        self.flush()
        tau = torch.cat((x_mpc, u_mpc), dim=2).transpose(0, 1).contiguous()
        self.train_data = tau[:n_train]
        self.val_data = tau[n_train:n_train + n_val]
        self.test_data = tau[tau.shape[0] - n_test:]

    def _solve(self, dx, xinit, Q, p, u_init, eps_override, lqr_iter_override, update_dynamics):
        xinit = torch.as_tensor(xinit, dtype=self.dtype, device=self.device)
        n_batch = xinit.shape[0]
        Q = Q.to(device=self.device, dtype=self.dtype)
        p = p.to(device=self.device, dtype=self.dtype)
        cost = TiledQuadCost(Q, p, self.mpc_T, n_batch)     # Q, p repeated to [T,B,ns,ns], [T,B,ns] (il_env.py:119-129)
        assert cost.C.dim() == 4 and cost.c.dim() == 3
        if u_init is not None:
            u_init = torch.as_tensor(u_init, dtype=self.dtype, device=self.device)
        # the reference builds a BoxDDP per call (il_env.py:131-156); the solver object holds nothing of a solve but its
        # status, so one per configuration is kept (a torch Module costs ~0.1 ms to construct - a seventh of a step)
        key = (n_batch, eps_override if eps_override else self.true_dx.mpc_eps,
               lqr_iter_override if lqr_iter_override else self.lqr_iter, bool(update_dynamics), self.quiet, str(self.device),
               getattr(self, "lazy_status", False))
        solver = self._solvers.get(key) if hasattr(self, "_solvers") else None
        if solver is None:
            solver = BoxDDP(T=self.mpc_T, u_lower=self.true_dx.lower, u_upper=self.true_dx.upper, n_batch=n_batch,
                            n_state=self.true_dx.n_state, n_ctrl=self.true_dx.n_ctrl, u_init=None, eps=key[1],
                            max_iter=key[2], verbose=False, exit_unconverged=False, detach_unconverged=True,
                            line_search_decay=self.true_dx.linesearch_decay,
                            max_line_search_iter=self.true_dx.max_linesearch_iter, update_dynamics=update_dynamics,
                            quiet=self.quiet, lazy_status=key[6])
            if not hasattr(self, "_solvers"):
                self._solvers = {}
            self._solvers[key] = solver
        solver.u_init = u_init
        self.last_solver = solver
        x_mpc, u_mpc, _ = solver((xinit, cost, dx))
        return x_mpc, u_mpc

    def mpc(self, dx, xinit, q, p, u_init=None, eps_override=None, lqr_iter_override=None, update_dynamics=False):
        """box-DDP under the diagonal cost diag(q), p tiled over time and batch -> (x [T,B,3], u [T,B,1])  (:104-158)"""
        return self._solve(dx, xinit, torch.diag(torch.as_tensor(q)), torch.as_tensor(p), u_init, eps_override,
                           lqr_iter_override, update_dynamics)

    def mpc_Q(self, dx, xinit, Q, p, u_init=None, eps_override=None, lqr_iter_override=None, update_dynamics=False):
        """the same with a full cost matrix Q [n_sc, n_sc]  (:160-213)"""
        return self._solve(dx, xinit, torch.as_tensor(Q), torch.as_tensor(p), u_init, eps_override,
                           lqr_iter_override, update_dynamics)


class Pendulum_Net_cost_logit(torch.nn.Module):
    """learnable pendulum cost: q = sigmoid(learn_q_logit), p = sqrt(q) * learn_p  (pendulum_net.py:12-39)"""

    def __init__(self, n_sc, device="cuda", dtype=torch.float32):
        super().__init__()
        self.n_sc = n_sc
        self.learn_q_logit = torch.nn.Parameter(torch.zeros(n_sc, device=device, dtype=dtype))
        self.learn_p = torch.nn.Parameter(torch.zeros(n_sc, device=device, dtype=dtype))

    def forward(self, xinit, env, train_warm_start_idxs=None):
        q = torch.sigmoid(self.learn_q_logit)
        p = torch.sqrt(q) * self.learn_p
        u_init = None
        if train_warm_start_idxs is not None:     # [B,T,nu] warm-start controls -> time-major (pendulum_net.py:36-37)
            u_init = torch.as_tensor(train_warm_start_idxs).transpose(0, 1)
        return env.mpc(env.true_dx, xinit, q, p, u_init=u_init)
