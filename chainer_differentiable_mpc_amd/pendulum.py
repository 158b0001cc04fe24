"""`PendulumDx` - the pendulum dynamics of env_dx/pendulum.py:31-145 (forward model, constants, true
objective) as a torch module, plus the analytic linearisation the reference obtains from chainer.grad
(mpc/approximate.py:77-119).  Plot/video helpers of the reference are out of scope."""
import numpy as np
import torch


class PendulumDx(torch.nn.Module):
    def __init__(self, params=None, simple=True):
        super().__init__()
        self.simple = simple
        self.max_torque = 2.0
        # Derivative of the torque clamp AT u = +-max_torque (forward(): torch.clamp; the reference: F.clip, env_dx/
        # pendulum.py:86, differentiated by chainer.grad in linearize_dynamics).  True = 1 there (closed interval: what
        # Chainer's ClipGrad and torch.clamp's autograd are taken to compute), False = 0.  Box-DDP's bounds equal the
        # torque limit, so saturated controls sit exactly on it and this decides their column of F_t.  THE one place
        # where the convention is set: `linearize`, the kernels (`rollout_linearize`, `dmpc_box_ddp`) and the tests'
        # oracle all read it.  Chainer is not installable here, so the closed interval is an assumption.
        self.clamp_grad_closed = True
        self.dt = 0.05
        self.n_state = 3
        self.n_ctrl = 1
        if params is None:
            params = torch.tensor([10., 1., 1.] if simple else [10., 1., 1., 0., 0.])   # g, m, l (, damping, bias)
        self.params = params
        assert len(self.params) == (3 if simple else 5)
        self.goal_state = torch.tensor([1., 0., 0.])
        self.goal_weights = torch.tensor([1., 1., 0.1])
        self.ctrl_penalty = 0.001
        self.lower = -2.
        self.upper = 2.
        self.mpc_eps = 1e-3
        self.linesearch_decay = 0.2
        self.max_linesearch_iter = 5

    def forward(self, x, u):
        """(cos th, sin th, dth), torque -> next state   (pendulum.py:65-102)"""
        squeeze = x.dim() == 1
        if squeeze:
            x, u = x.unsqueeze(0), u.unsqueeze(0)
        assert x.dim() == 2 and x.shape[0] == u.shape[0] and x.shape[1] == self.n_state and u.shape[1] == self.n_ctrl
        p = self.params.to(x)
        g, m, l = p[0], p[1], p[2]
        u = torch.clamp(u, -self.max_torque, self.max_torque)[:, 0]
        cos_th, sin_th, dth = x[:, 0], x[:, 1], x[:, 2]
        th = torch.atan2(sin_th, cos_th)
        if self.simple:
            newdth = dth + self.dt * (-3. * g / (2. * l) * (-sin_th) + 3. * u / (m * l ** 2))
        else:
            d, b = p[3], p[4]
            newdth = dth + self.dt * (-3. * g / (2. * l) * (-torch.sin(th + b)) + 3. * u / (m * l ** 2) - d * th)
        newth = th + newdth * self.dt
        state = torch.stack((torch.cos(newth), torch.sin(newth), newdth), dim=1)
        return state.squeeze(0) if squeeze else state

    def linearize(self, x, u):
        """analytic F_t = d next / d [x;u], f_t = next - F_t [x;u] along the re-rolled trajectory (simple model)"""
        if not self.simple:
            from .approximate import linearize_dynamics
            fn = lambda a, b: self.forward(a, b)  # noqa: E731
            return linearize_dynamics(x, u, fn)
        T = x.shape[0]
        p = self.params.to(x)
        g, m, l = p[0], p[1], p[2]
        dt = self.dt
        xs = [x[0]]
        Fs, fs = [], []
        for t in range(T - 1):
            xt, ut = xs[t], u[t]
            c, s, w = xt[:, 0], xt[:, 1], xt[:, 2]
            uc = torch.clamp(ut[:, 0], -self.max_torque, self.max_torque)
            au = ut[:, 0].abs()
            inside = (au <= self.max_torque if self.clamp_grad_closed else au < self.max_torque).to(xt.dtype)
            r2 = c * c + s * s
            th = torch.atan2(s, c)
            nw = w + dt * (3. * g / (2. * l) * s + 3. * uc / (m * l ** 2))
            nth = th + nw * dt
            # d th / d(c, s) = (-s, c) / (c^2 + s^2)
            dth_dc, dth_ds = -s / r2, c / r2
            dnw = torch.stack((torch.zeros_like(c), dt * 3. * g / (2. * l) * torch.ones_like(c), torch.ones_like(c),
                               dt * 3. / (m * l ** 2) * inside), dim=1)                       # d nw / d(c,s,w,u)
            dnth = torch.stack((dth_dc, dth_ds, torch.zeros_like(c), torch.zeros_like(c)), dim=1) + dt * dnw
            Ft = torch.stack((-torch.sin(nth)[:, None] * dnth, torch.cos(nth)[:, None] * dnth, dnw), dim=1)
            new_x = torch.stack((torch.cos(nth), torch.sin(nth), nw), dim=1)
            Fs.append(Ft)
            fs.append(new_x - torch.einsum("bij,bj->bi", Ft, torch.cat((xt, ut), dim=1)))
            xs.append(new_x)
        return torch.stack(Fs, 0), torch.stack(fs, 0)

    def fused_ok(self, x_init, u):
        """the one-launch rollout + linearisation applies: simple model, float tensors on the GPU, no gradient
        wanted through the parameters"""
        return (self.simple and isinstance(x_init, torch.Tensor) and x_init.is_cuda and u.is_cuda and
                not self.params.requires_grad and not x_init.requires_grad and not u.requires_grad)

    def host_params(self):
        """(g, m, l) as Python floats for the kernels' scalar arguments.  Host-resident parameters (the default) are read
        on every call - three floats, cheaper than any staleness check, and writes through `.data` are seen.  Parameters
        on the GPU cost a synchronising read-back, so that one is cached on (storage, version counter); a write through
        `.data` changes neither: call `invalidate_host_params()` after one."""
        p = self.params
        if not p.is_cuda:
            return tuple(float(v) for v in p.detach().tolist()[:3])
        key = (p.data_ptr(), p._version)
        if getattr(self, "_host_params", None) is None or self._host_params[0] != key:
            self._host_params = (key, tuple(float(v) for v in p.detach().cpu().tolist()[:3]))
        return self._host_params[1]

    def invalidate_host_params(self):
        self._host_params = None

    def rollout_linearize(self, x_init, u, want_model=True):
        """(x [T,B,3], F [T-1,B,3,4], f [T-1,B,3]) from x_init [B,3], u [T,B,1] in one kernel launch
        (`dmpc_pendulum_rollout_linearize`): get_traj + linearize_dynamics of the reference's BoxDDP loop"""
        from . import _lib
        lib = _lib.load()
        T, B = u.shape[0], u.shape[1]
        d = x_init.device
        x0, ud = _lib.f32c(x_init.detach(), d), _lib.f32c(u.detach(), d)
        x = torch.empty((T, B, 3), dtype=torch.float32, device=d)
        F = torch.empty((max(T - 1, 0), B, 3, 4), dtype=torch.float32, device=d) if want_model else None
        f = torch.empty((max(T - 1, 0), B, 3), dtype=torch.float32, device=d) if want_model else None
        g_, m_, l_ = self.host_params()
        with _lib.guard(d):
            rc = lib.dmpc_pendulum_rollout_linearize(T, B, _lib.ptr(x0), _lib.ptr(ud), g_, m_, l_, float(self.dt),
                                                     float(self.max_torque), int(self.clamp_grad_closed), _lib.ptr(x),
                                                     _lib.ptr(F), _lib.ptr(f),
                                                     _lib.stream_ptr(d))
        _lib.check(rc, "dmpc_pendulum_rollout_linearize")
        return x.to(x_init.dtype), (None if F is None else F.to(x_init.dtype)), (None if f is None else f.to(x_init.dtype))

    def get_true_obj(self):
        """(q, p): diagonal of Q and linear term of the true quadratic cost   (pendulum.py:122-145)"""
        q = torch.cat((self.goal_weights, self.ctrl_penalty * torch.ones(self.n_ctrl)))
        px = -torch.sqrt(self.goal_weights) * self.goal_state
        p = torch.cat((px, torch.zeros(self.n_ctrl)))
        return q, p


def sample_xinit(n_batch, seed=0):
    """x_init = (cos th, sin th, dth), th ~ U(-pi/2, pi/2), dth ~ U(-1, 1) - the draws of env_dx/il_env.py:55-69"""
    rng = np.random.RandomState(seed)
    th = (rng.rand(n_batch) - 0.5) * np.pi
    dth = (rng.rand(n_batch) - 0.5) * 2.0
    return np.stack((np.cos(th), np.sin(th), dth), axis=1)
