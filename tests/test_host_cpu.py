"""CPU: host-side logic of the package that needs no kernel - the small batched helpers of util.py against
oracle/linalg.py, `approximate_cost` / `linearize_dynamics` against vectors recorded from the
reference's mpc/approximate.py (8f-4), the pendulum's host functions against env_dx/pendulum.py / il_env.py
fixtures (8f-2), the data-set pickle round trip (8f-4)."""
import os
import pickle

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import util
from chainer_differentiable_mpc_amd.approximate import approximate_cost, linearize_dynamics
from chainer_differentiable_mpc_amd.il_env import IL_Env
from chainer_differentiable_mpc_amd.pendulum import PendulumDx, sample_xinit
from oracle import linalg as ola
from tests.helpers import GOLDEN


def _t(a):
    return torch.as_tensor(a)


def helpers_against_oracle(device):
    rng = np.random.RandomState(4)
    B, n, m = 7, 5, 3
    A, x, y, Q = rng.randn(B, m, n), rng.randn(B, n), rng.randn(B, n), rng.randn(B, n, n)
    z = rng.randn(B, m)
    lo = -np.abs(rng.randn(B, n))
    hi = np.abs(rng.randn(B, n))
    d = lambda a: torch.as_tensor(a, device=device)   # noqa: E731
    cases = [("bmv", util.bmv(d(A), d(x)), ola.bmv(A, x)), ("bger", util.bger(d(z), d(y)), ola.bger(z, y)),
             ("bquad", util.bquad(d(x), d(Q)), ola.bquad(x, Q)), ("bdot", util.bdot(d(x), d(y)), ola.bdot(x, y)),
             ("clamp", util.clamp(d(3 * x), d(lo), d(hi)), ola.clamp(3 * x, lo, hi)),
             ("expand_time_batch", util.expand_time_batch(d(Q[0]), 4, 3), ola.expand_time_batch(Q[0], 4, 3)),
             ("expand_batch", util.expand_batch(d(Q[0]), 3), ola.expand_batch(Q[0], 3))]
    for name, got, ref in cases:
        got = got.cpu().numpy()
        assert got.shape == ref.shape, name
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12, err_msg=name)
    with pytest.raises(AssertionError):
        util.clamp(d(x), d(hi + 1.0), d(hi))              # lower > upper (util.py:116)
    c = util.QuadCost()
    assert c.C is None and c.c is None and util.LinDx(F=1).f is None    # defaults (util.py:25-32)


def test_util_helpers_match_the_oracle_cpu():
    helpers_against_oracle("cpu")


def test_get_traj_and_get_cost_match_the_oracle():
    from chainer_differentiable_mpc_amd import synthetic
    from oracle import box_ddp as obox
    from oracle import mpc as ompc
    T, B, nx, nu = 6, 4, 3, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=9)
    u = np.random.RandomState(1).randn(T, B, nu)
    x = util.get_traj(T, _t(u), _t(p["x_init"]), util.LinDx(_t(p["F"]), _t(p["f"])))
    xr = obox.get_traj(T, u, p["x_init"], ompc.LinDx(p["F"], p["f"]))
    np.testing.assert_allclose(x.numpy(), xr, atol=1e-12)
    c = util.get_cost(T, _t(u), util.QuadCost(_t(p["C"]), _t(p["c"])), x=x)
    np.testing.assert_allclose(c.numpy(), ompc.get_cost(T, u, ompc.QuadCost(p["C"], p["c"]), xr), atol=1e-10)


def test_approximate_cost_matches_the_reference():
    """mpc/approximate.py:18-54 on a quadratic and a non-quadratic cost (vectors from the reference's own function)"""
    g = np.load(os.path.join(GOLDEN, "approx_cost.npz"))
    x, u = _t(g["x"]), _t(g["u"])
    Cq, cq = _t(g["Cq"]), _t(g["cq"])

    def quad(tau):
        return 0.5 * ((tau @ Cq) * tau).sum(1) + (tau * cq).sum(1)

    def nonquad(tau):
        return torch.sqrt(1.0 + (tau ** 2).sum(1)) + (torch.sin(tau[:, :-1]) * tau[:, 1:]).sum(1)

    for name, fn in (("quad", quad), ("nonquad", nonquad)):
        H, gr, c = approximate_cost(x, u, fn)
        np.testing.assert_allclose(H.numpy(), g[name + "_H"], atol=1e-12)
        np.testing.assert_allclose(gr.numpy(), g[name + "_g"], atol=1e-12)
        np.testing.assert_allclose(c.numpy(), g[name + "_cost"], atol=1e-12)
    # the quadratic model of a quadratic is the quadratic: H = sym(C), grads - H tau = c
    H, gr, _ = approximate_cost(x, u, quad)
    np.testing.assert_allclose(H[0, 0].numpy(), 0.5 * (g["Cq"] + g["Cq"].T), atol=1e-12)
    np.testing.assert_allclose(gr[0, 0].numpy(), g["cq"], atol=1e-12)
    # numpy central differences of the non-quadratic cost, an oracle independent of any autograd
    tau = np.concatenate((g["x"], g["u"]), axis=2)[1]
    f = lambda t: nonquad(_t(t)).numpy()      # noqa: E731
    eps = 1e-5
    n = tau.shape[1]
    grad = np.stack([(f(tau + eps * np.eye(n)[i]) - f(tau - eps * np.eye(n)[i])) / (2 * eps) for i in range(n)], axis=1)
    Hn, gn, _ = approximate_cost(x, u, nonquad)
    np.testing.assert_allclose(gn[1].numpy() + np.einsum("bij,bj->bi", Hn[1].numpy(), tau), grad, atol=1e-8)


def test_approximations_keep_the_graph_to_learnable_parameters():
    """the reference's f_t = new_x - R x - S u and `grad - H tau` stay on the graph (approximate.py:47,106):
    parameters of a learnable dynamics / cost receive gradients through them; the Jacobians / Hessians are constants"""
    g = np.load(os.path.join(GOLDEN, "approx_cost.npz"))
    x, u = _t(g["x"]), _t(g["u"])
    w = torch.tensor([1.0, 2.0, 0.5, 1.5, 0.25], dtype=torch.float64, requires_grad=True)
    H, gr, c = approximate_cost(x, u, lambda tau: 0.5 * (tau * tau * w).sum(1) + torch.sin(tau * w).sum(1))
    assert not H.requires_grad and gr.requires_grad and c.requires_grad
    gr.sum().backward()
    assert float(w.grad.abs().min()) > 0
    A = (torch.eye(3, dtype=torch.float64) * 0.9).requires_grad_(True)
    Bm = torch.ones((3, 2), dtype=torch.float64, requires_grad=True)
    F, f = linearize_dynamics(x, u, lambda a, b: torch.tanh(a @ A.T + b @ Bm.T))
    assert not F.requires_grad and f.requires_grad
    f.sum().backward()
    assert float(A.grad.abs().max()) > 0 and float(Bm.grad.abs().max()) > 0
    with torch.no_grad():
        F2, f2 = linearize_dynamics(x, u, lambda a, b: torch.tanh(a @ A.T + b @ Bm.T))
    assert not f2.requires_grad and torch.equal(F2, F)


def test_clamp_derivative_convention_is_one_flag():
    """The derivative of the torque clamp AT u = +-max_torque is set in ONE place, PendulumDx.clamp_grad_closed (an
    assumption about Chainer's F.clip backward): `linearize` follows it, the oracle takes it as an
    argument, and the two agree in both settings; inside and outside the interval nothing depends on it."""
    from oracle import box_ddp as obox
    T, B = 6, 5
    x0 = sample_xinit(B, seed=2)
    u = np.random.RandomState(5).uniform(-3.0, 3.0, size=(T, B, 1))
    u[1, :, 0] = 2.0          # exactly on the upper limit
    u[2, :, 0] = -2.0         # ... and on the lower one
    u[3, :, 0] = [2.5, -2.5, 1.0, -1.0, 0.0]
    col = {}
    for closed in (True, False):
        dx = PendulumDx()
        assert dx.clamp_grad_closed is True       # the default the fixtures were recorded with
        dx.clamp_grad_closed = closed
        x = util.get_traj(T, _t(u), _t(x0), dx)
        F, f = dx.linearize(x, _t(u))
        Fo, fo = obox.pendulum_linearize(x.numpy(), u, clamp_grad_closed=closed)
        np.testing.assert_allclose(F.numpy(), Fo, atol=1e-13)
        np.testing.assert_allclose(f.numpy(), fo, atol=1e-13)
        col[closed] = F.numpy()[:, :, 2, 3]         # d (new dtheta) / d u
    np.testing.assert_allclose(col[True][1:3], 0.15, rtol=1e-12)      # dt * 3 / (m l^2) at the limits ...
    assert np.all(col[False][1:3] == 0.0)                             # ... or nothing
    np.testing.assert_allclose(col[True][3], [0, 0, 0.15, 0.15, 0.15], rtol=1e-12)
    np.testing.assert_array_equal(col[True][[0, 3, 4]], col[False][[0, 3, 4]])


def test_pendulum_host_functions_match_the_reference():
    """PendulumDx.forward / get_true_obj / constants (env_dx/pendulum.py:31-145), the analytic and the autograd
    linearisation, IL_Env.sample_xinit (il_env.py:55-69)"""
    g = np.load(os.path.join(GOLDEN, "pendulum.npz"))
    dx = PendulumDx()
    np.testing.assert_allclose(dx(_t(g["x"]), _t(g["u"])).numpy(), g["next"], atol=1e-15)
    q, p = dx.get_true_obj()
    np.testing.assert_allclose(q.numpy(), g["q"], atol=1e-7)
    np.testing.assert_allclose(p.numpy(), g["p"], atol=1e-7)
    for name in ("dt", "max_torque", "lower", "upper", "mpc_eps", "linesearch_decay", "max_linesearch_iter"):
        assert float(getattr(dx, name)) == float(g[name]), name
    np.testing.assert_allclose(dx.params.numpy(), g["params"])
    Fa, fa = dx.linearize(_t(g["lin_x"]), _t(g["lin_u"]))
    Fg, fg = linearize_dynamics(_t(g["lin_x"]), _t(g["lin_u"]), lambda a, b: dx(a, b))
    for F, f in ((Fa, fa), (Fg, fg)):
        np.testing.assert_allclose(F.numpy(), g["lin_F"], atol=1e-13)
        np.testing.assert_allclose(f.numpy(), g["lin_f"], atol=1e-13)
    np.random.seed(0)
    np.testing.assert_allclose(IL_Env.sample_xinit(128), g["xinit128"], atol=1e-15)
    np.testing.assert_allclose(sample_xinit(128, seed=0), g["xinit128"], atol=1e-15)
    np.random.seed(0)
    xi = IL_Env.sample_xinit(1024)
    np.testing.assert_allclose(xi[:64], g["xinit1024_head"], atol=1e-15)
    np.testing.assert_allclose(xi.sum(axis=0), g["xinit1024_sum"], atol=1e-10)


def test_dataset_pickle_round_trip(tmp_path):
    """env_dx/make_dataset.py:28-34 / il_exp.py:44-45: the pickled IL_Env holds numpy arrays [n, T, n_sc]"""
    env = IL_Env('pendulum', lqr_iter=3, device="cpu")
    rng = np.random.RandomState(0)
    env.train_data = _t(rng.randn(5, 20, 4)).float()
    env.val_data = _t(rng.randn(2, 20, 4)).float()
    env.test_data = _t(rng.randn(1, 20, 4)).float()
    path = os.path.join(str(tmp_path), "pendulum.pkl")
    with open(path, "wb") as f:
        pickle.dump(env, f)
    with open(path, "rb") as f:
        raw = pickle.load(f)
    assert raw.lqr_iter == 3 and raw.mpc_T == 20 and isinstance(raw.true_dx, PendulumDx)
    assert torch.equal(raw.train_data, env.train_data) and list(raw.test_data.shape) == [1, 20, 4]
    st = env.__getstate__()
    assert isinstance(st["train_data"], np.ndarray) and st["train_data"].dtype == np.float64
