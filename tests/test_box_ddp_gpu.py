"""GPU: the callers of the hot path (SURVEY.md 8f "next" rows) - BoxDDP outer loop, pendulum linearisation,
MpcNet - against the reference's recorded trace and the numpy oracle."""
import os
import warnings

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import BoxDDP, LinDx, MPCstep, MpcNet_dx, PendulumDx, QuadCost, synthetic
from chainer_differentiable_mpc_amd.approximate import linearize_dynamics
from chainer_differentiable_mpc_amd.util import get_traj
from chainer_differentiable_mpc_amd.pendulum import sample_xinit
from oracle import box_ddp as obox
from oracle import mpc as ompc
from tests.helpers import GOLDEN, TOL_COSTATE, TOL_PRIMAL, TOL_STEP_PENDULUM, assert_close, assert_step_close, npy

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(a, dtype=dtype, device="cuda")


def test_box_ddp_reference_trace():
    """BoxDDP + LinDx + QuadCost: the reference's own run (tests/golden/boxddp_trace.npz)"""
    g = np.load(os.path.join(GOLDEN, "boxddp_trace.npz"))
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=True)
    solver = BoxDDP(T, -float(g["bound"]), float(g["bound"]), B, nx, nu, None, max_iter=10, quiet=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u, costs = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
    assert solver.status in str(g["stdout"])
    assert_close(npy(u), g["u"], TOL_PRIMAL, "u")
    assert_close(npy(x), g["x"], TOL_PRIMAL, "x")
    assert_close(npy(costs), g["costs"], TOL_PRIMAL, "costs")


def pendulum_problem(B, T, seed=0):
    dx = PendulumDx()
    q, pp = dx.get_true_obj()
    x0 = sample_xinit(B, seed=seed).astype(np.float32).astype(np.float64)
    Q = np.tile(np.diag(q.numpy().astype(np.float64)), (T, B, 1, 1))
    pv = np.tile(pp.numpy().astype(np.float64), (T, B, 1))
    return dx, x0, Q, pv


def _zero_control_cost(x0, Q, pv, T):
    B = x0.shape[0]
    xs = [x0]
    for t in range(T - 1):
        xs.append(obox.pendulum_step(xs[t], np.zeros((B, 1))))
    tau0 = np.concatenate((np.stack(xs), np.zeros((T, B, 1))), axis=2)
    return 0.5 * np.einsum("tbi,tbij,tbj->b", tau0, Q, tau0) + (tau0 * pv).sum(axis=(0, 2))


def test_pendulum_box_ddp_config2_against_oracle():
    """BASELINE.json configs[1]: pendulum box-DDP, batch=128, T=20 (env_dx/il_env.py:104-151, pendulum.py:40-63),
    against the oracle (itself pinned to the reference's BoxDDP + PendulumDx run, tests/golden/pendulum_boxddp.npz).
    The swing-up iteration amplifies rounding, so the loop is pinned where it is well-conditioned: the first
    iterations, and EVERY single iLQR step taken from common iterates along the run at the plain one-step tolerance
    (rows whose line search is a float32 tie: one of the search's candidates - tests/helpers.py)."""
    B, T = 128, 20
    dx, x0, Q, pv = pendulum_problem(B, T)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
    okw = dict(linearize=obox.pendulum_linearize, batch_coupled=False, **kw)
    cost_d, cost_o = QuadCost(dev(Q), dev(pv)), ompc.QuadCost(Q, pv)
    lo, hi = np.full((T, B, 1), dx.lower), np.full((T, B, 1), dx.upper)

    def product(max_iter):
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=max_iter, exit_unconverged=False, quiet=True, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return solver((dev(x0), cost_d, dx))

    # (1) the first two outer iterations against the oracle
    x, u, costs = product(2)
    xr, ur, cr, *_ = obox.box_ddp(x0, cost_o, obox.pendulum_step, T, dx.lower, dx.upper, 3, 1, max_iter=2, **okw)
    assert_close(npy(costs), cr, TOL_PRIMAL, "costs after 2 iterations")      # (measured 3.1e-5, profiles/r04/parity_margins.txt)
    assert np.mean(np.abs(npy(u) - ur) < 1e-3) > 0.99

    # (2) one iLQR step (linearise, PNQP backward pass, clamped line search on the true pendulum) from common
    # iterates taken along the oracle's run, incl. late ones with saturated torques
    n_tie = n_fork = 0
    for k in (1, 4, 8):
        _, uk, *_ = obox.box_ddp(x0, cost_o, obox.pendulum_step, T, dx.lower, dx.upper, 3, 1, max_iter=k, **okw)
        uk = uk.astype(np.float32).astype(np.float64)
        xk = obox.get_traj(T, uk, x0, obox.pendulum_step)
        Fm, fm = obox.pendulum_linearize(xk, uk)
        xo, uo, _, fo, Ko, ko = ompc.mpc_forward(Q, pv, Fm, fm, uk, xk, lo, hi, cost_o, obox.pendulum_step,
                                                 dx.linesearch_decay, dx.max_linesearch_iter, T, 3, 1,
                                                 need_expand=True, batch_coupled=False)
        with torch.no_grad():
            ud = dev(uk)
            xd, Fd, fd = dx.rollout_linearize(dev(x0), ud)
            step = MPCstep(controls=ud, T=T, u_upper=dev(hi), u_lower=dev(lo), n_batch=B, n_state=3, n_ctrl=1,
                           current_states=xd, true_cost=cost_d, true_dynamics=dx, ls_decay=dx.linesearch_decay,
                           max_ls_iter=dx.max_linesearch_iter, need_expand=True)
            xn, un = step.forward((xd[0], dev(Q), dev(pv), Fd, fd))
        old = ompc.get_cost(T, uk, cost_o, xk)

        def candidates(rows, alpha):
            xc, uc, _ = ompc.ls_rollout(Ko[:, rows], ko[:, rows], uk[:, rows], xk[:, rows], lo[:, rows], hi[:, rows],
                                        ompc.QuadCost(Q[:, rows], pv[:, rows]), obox.pendulum_step,
                                        np.full(len(rows), alpha), T)
            return xc, uc

        nt, nf = assert_step_close(npy(un), npy(xn), uo, xo, old, fo.costs, candidates, TOL_STEP_PENDULUM, "step from iterate %d" % k)
        n_tie, n_fork = n_tie + nt, n_fork + nf
        assert (npy(step.for_out.costs) <= old + 4e-6 * np.abs(old) + 1e-5).all(), k      # descent on every sample
    assert n_tie < 3 * B // 2                          # ties are a minority: most rows were held to the plain tolerance

    # (3) the full run: feasible, torque limit active, never worse than the zero-control rollout, the returned x is
    # the rollout of the returned u, and the best-so-far cost did not get worse after iteration 2
    c2 = npy(costs)
    x, u, costs = product(12)
    assert bool(((u >= dx.lower) & (u <= dx.upper)).all())
    assert float((u.abs() == 2.0).float().mean()) > 0.05
    assert (npy(costs) <= _zero_control_cost(x0, Q, pv, T) + 1e-4).all()
    assert (npy(costs) <= c2 + 1e-3).all()
    xs = obox.get_traj(T, npy(u).astype(np.float64), x0, obox.pendulum_step)
    assert np.abs(xs - npy(x)).max() < 5e-3


def test_box_ddp_lindx_b128_against_oracle():
    """the same loop on a well-conditioned problem of config 2's size (B=128, T=20, LinDx + QuadCost with active
    bounds): the full 10-iteration run against the oracle at the plain tolerance"""
    B, T, nx, nu = 128, 20, 3, 1
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=31, with_f=True)
    solver = BoxDDP(T, -0.3, 0.3, B, nx, nu, None, max_iter=10, quiet=True, eps=1e-3)   # the pendulum experiments use 1e-3 too (mpc_eps)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u, costs = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
    xr, ur, cr, status, n_iter, *_ = obox.box_ddp(p["x_init"], ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]),
                                                  T, -0.3, 0.3, nx, nu, batch_coupled=False, eps=1e-3)
    assert solver.status.strip() == status.strip() and solver.n_iter == n_iter, (solver.status, solver.n_iter, status, n_iter)
    assert float((u.abs() == 0.3).float().mean()) > 0.1
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(costs), cr, TOL_PRIMAL, "costs")


@pytest.mark.parametrize("dims", [(12, 4), (16, 4)], ids=lambda d: "%dx%d" % d)
def test_box_ddp_quadrotor_sized_against_oracle(dims):
    """`BoxDDP` with a LinDx at 12 / 16 states and four controls: the device-driven loop over the wide row kernels (the sweep
    with the box QP inside lqr_wide_kernel<..., MPC>, the line search of mpc_wide_forward_kernel.hpp) - the iterates and the
    status of the oracle's loop, and of the host loop over `MPCstep` objects"""
    nx, nu = dims
    B, T = 12, 8
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=35, with_f=True)
    xr, ur, cr, status, n_iter, *_ = obox.box_ddp(p["x_init"], ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]),
                                                  T, -0.25, 0.25, nx, nu, batch_coupled=False, eps=1e-3, max_iter=10)
    for device_loop in (True, False):
        solver = BoxDDP(T, -0.25, 0.25, B, nx, nu, None, max_iter=10, quiet=True, eps=1e-3, device_loop=device_loop)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, costs = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
        assert solver.status.strip() == status.strip() and solver.n_iter == n_iter, (solver.status, solver.n_iter, status, n_iter)
        assert float((u.abs() == 0.25).float().mean()) > 0.02
        # The loop stops once a step is shorter than eps = 1e-3 (the pendulum experiments' mpc_eps), i.e. within a fraction of
        # eps of its fixed point.  Where two runs of it END is therefore resolved to eps, not to the float32 contract - that is
        # the loop's own stopping resolution and is checked as such (measured: 2.6e-4 in x); PARITY is the next block.
        assert np.abs(npy(u) - ur).max() <= 1e-3 and np.abs(npy(x) - xr).max() <= 1e-3 * max(1.0, np.abs(xr).max())
        assert_close(npy(costs), cr, TOL_PRIMAL, "costs")
        # ... parity at the contract's tolerance: the loop's FIRST iteration (from the common initial controls), both sides.  The
        # loop converges in a handful of iterations here; from its second iterate on most rows lower the cost by less than
        # float32 resolves (line-search ties, tests/helpers.py: 4 of 12 rows after two iterations, all 12 one iteration before the
        # stop), so a later common iterate compares tie-breaking, not arithmetic.  (One MPC step at these shapes from synthetic
        # iterates is held to 1e-4 by tests/test_mpc_step_gpu.py.)
        cost_o, lin_o = ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"])
        x1r, u1r, c1r, *_ = obox.box_ddp(p["x_init"], cost_o, lin_o, T, -0.25, 0.25, nx, nu, batch_coupled=False, eps=1e-3, max_iter=1)
        one = BoxDDP(T, -0.25, 0.25, B, nx, nu, None, max_iter=1, quiet=True, eps=1e-3, device_loop=device_loop, exit_unconverged=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x1, u1, c1 = one((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
        assert_close(npy(u1), u1r, TOL_PRIMAL, "first iteration: u")
        assert_close(npy(x1), x1r, TOL_PRIMAL, "first iteration: x")
        assert_close(npy(c1), c1r, TOL_PRIMAL, "first iteration: costs")


def test_box_ddp_beyond_8_controls_runs_its_host_loop_against_oracle():
    """`BoxDDP` (mpc/box_ddp.py:93-291) on a problem with 12 controls: `dmpc_box_ddp`'s device-driven loop declines (its
    workspace has no room for the any-size kernels' matrices), `BoxDDP` runs the reference's loop over `MPCstep` objects on
    the tiled kernels (mpc_tiled.hpp) - same iterates, same status as the oracle"""
    B, T, nx, nu = 6, 8, 10, 12
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=33, with_f=True)
    solver = BoxDDP(T, -0.25, 0.25, B, nx, nu, None, max_iter=6, quiet=True, eps=1e-3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u, costs = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
    xr, ur, cr, status, n_iter, *_ = obox.box_ddp(p["x_init"], ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]),
                                                  T, -0.25, 0.25, nx, nu, batch_coupled=False, eps=1e-3, max_iter=6)
    assert solver.status.strip() == status.strip() and solver.n_iter == n_iter, (solver.status, solver.n_iter, status, n_iter)
    assert float((u.abs() == 0.25).float().mean()) > 0.02
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(costs), cr, TOL_PRIMAL, "costs")


def test_pendulum_analytic_linearisation_matches_autograd():
    from chainer_differentiable_mpc_amd.approximate import linearize_dynamics
    dx = PendulumDx()
    T, B = 6, 9
    x0 = dev(sample_xinit(B, seed=3), torch.float64)
    u = (torch.rand((T, B, 1), dtype=torch.float64, device="cuda") - 0.5) * 3.0
    xs = [x0]
    for t in range(T - 1):
        xs.append(dx(xs[t], u[t]))
    x = torch.stack(xs)
    Fa, fa = dx.linearize(x, u)
    Fg, fg = linearize_dynamics(x, u, lambda a, b: dx(a, b))
    assert float((Fa - Fg).abs().max()) < 1e-10 and float((fa - fg).abs().max()) < 1e-10


def test_mpcnet_gradient_flows_to_dynamics_parameters():
    """MpcNet_dx (mpc/mpc_net.py:20-87): d loss / d(A, B) through BoxDDP's final no-op MPCstep node, against the
    oracle's MPCstep.backward on the same converged iterate"""
    T, B, nx, nu = 5, 6, 3, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=17)
    lo = torch.full((T, B, nu), -0.25)
    hi = torch.full((T, B, nu), 0.25)
    net = MpcNet_dx(T, lo, hi, B, nx, nu, seed=1, u_init=None, max_iter=12, quiet=True).cuda()
    C, c = dev(p["C"], torch.float64), dev(p["c"], torch.float64)
    x0 = dev(p["x_init"], torch.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u, costs = net((x0, QuadCost(C, c)))
    w_x = torch.linspace(-1, 1, x.numel(), device="cuda", dtype=x.dtype).reshape(x.shape)
    (w_x * x).sum().add(u.sum()).backward()
    assert net.A.grad is not None and net.B.grad is not None
    # oracle: gradient of the same scalar through MPCstep.backward at (x, u), summed over time and batch
    AB = np.concatenate((net.A.detach().cpu().numpy(), net.B.detach().cpu().numpy()), axis=1)
    Fm = np.tile(AB, (T - 1, B, 1, 1))
    fm = np.zeros((T - 1, B, nx))
    xd, ud = npy(x), npy(u)
    tau = np.concatenate((xd, ud), axis=2)
    c_hat = np.einsum("tbij,tbj->tbi", p["C"], tau) + p["c"]      # need_expand re-centring is NOT applied to the retained c
    out = ompc.mpc_backward(xd[0], p["C"], p["c"], Fm, fm, xd, ud, lo.numpy().astype(np.float64),
                            hi.numpy().astype(np.float64), npy(w_x), np.ones((T, B, nu)), T, nx, nu)
    keep = (npy(net.mpc_layer.forward.__self__.mpc_layer_last_full_du) < net.mpc_layer.eps) if False else None
    dF = out[3].sum(axis=(0, 1))
    got = np.concatenate((net.A.grad.cpu().numpy(), net.B.grad.cpu().numpy()), axis=1)
    if net.mpc_layer.status == "Converged":
        assert_close(got, dF, TOL_COSTATE, "d(A|B)")
    else:   # unconverged samples are detached (box_ddp.py:263-289): only check the gradient is finite and non-zero
        assert np.isfinite(got).all() and np.abs(got).max() > 0


@pytest.mark.parametrize("tag", ["wide", "tight"])
def test_the_reference_mpcnet_experiment_first_training_iteration(tag):
    """experiment_mpc/MpcNet.py:24-120 at its own sizes - T=5, three states, three controls, B=128, expert_seed 42,
    train_seed 1 - against what the unmodified reference returned (tests/golden/mpcnet_experiment.npz): the expert's
    BoxDDP solve under the true (A, B), the learner's MpcNet_dx solve, the imitation loss (:80-90) and d loss / d(A, B).
    `wide`: the experiment's own +-10 box (never reached); `tight`: +-0.6, a quarter of the controls on the box."""
    g = np.load(os.path.join(GOLDEN, "mpcnet_experiment.npz"))
    T, B, nx, nu = int(g["T"]), int(g["B"]), int(g["nx"]), int(g["nu"])
    ns, bound = nx + nu, float(g[tag + "_bound"])
    lo, hi = torch.full((T, B, nu), -bound, dtype=torch.float64), torch.full((T, B, nu), bound, dtype=torch.float64)
    net = MpcNet_dx(T, lo, hi, B, nx, nu, 1, u_init=None, max_iter=10, quiet=True).cuda()
    np.testing.assert_array_equal(npy(net.A), g[tag + "_A0"])       # the same draws as the reference (mpc_net.py:59-64)
    np.testing.assert_array_equal(npy(net.B), g[tag + "_B0"])
    C = dev(np.tile(np.eye(ns), (T, B, 1, 1)), torch.float64)
    c = dev(np.tile(g[tag + "_p"], (T, B, 1)), torch.float64)
    F_exp = dev(np.tile(np.concatenate((g[tag + "_A_exp"], g[tag + "_B_exp"]), axis=1), (T - 1, B, 1, 1)), torch.float64)
    f_exp = torch.zeros((T - 1, B, nx), dtype=torch.float64, device="cuda")
    x_init = dev(g[tag + "_x_init"], torch.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            expert = BoxDDP(T, lo, hi, B, nx, nu, None, quiet=True)
            x_true, u_true, _ = expert((x_init, QuadCost(C, c), LinDx(F_exp, f_exp)))
        x_pred, u_pred, _ = net((x_init, QuadCost(C, c)))
    assert expert.status == "Converged" and net.mpc_layer.status == "Converged"      # as the reference printed
    assert_close(npy(x_true), g[tag + "_x_true"], TOL_PRIMAL, "expert x")
    assert_close(npy(u_true), g[tag + "_u_true"], TOL_PRIMAL, "expert u")
    assert_close(npy(x_pred), g[tag + "_x_pred"], TOL_PRIMAL, "learner x")
    assert_close(npy(u_pred), g[tag + "_u_pred"], TOL_PRIMAL, "learner u")
    if tag == "tight":
        on_box = np.abs(np.abs(g[tag + "_u_pred"]) - bound) <= 1e-8
        assert 0.1 < on_box.mean() < 0.5
        np.testing.assert_array_equal(np.abs(np.abs(npy(u_pred)) - bound) <= 1e-6, on_box)     # the same clamped set
    loss = ((u_true - u_pred) ** 2).mean() + ((x_true - x_pred) ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g[tag + "_loss"])) <= TOL_PRIMAL * max(1.0, abs(float(g[tag + "_loss"])))
    assert_close(npy(net.A.grad), g[tag + "_gA"], TOL_COSTATE, "d loss / dA")
    assert_close(npy(net.B.grad), g[tag + "_gB"], TOL_COSTATE, "d loss / dB")


def test_fused_pendulum_rollout_and_linearisation():
    """dmpc_pendulum_rollout_linearize against the torch restatement of env_dx/pendulum.py:84-98 and its analytic
    Jacobian (PendulumDx.forward / .linearize, themselves pinned to the oracle by the BoxDDP tests above), incl.
    saturated torques (derivative 0 outside the clamp)"""
    B, T = 37, 20
    dx = PendulumDx()
    x0 = dev(sample_xinit(B, seed=3))
    rng = np.random.RandomState(4)
    u = dev(rng.uniform(-3.0, 3.0, size=(T, B, 1)))     # a third of the torques beyond +-2
    x, F, f = dx.rollout_linearize(x0, u)
    xr = get_traj(T, u, x0, dx)
    Fr, fr = dx.linearize(xr, u)
    assert_close(npy(x), npy(xr), 2e-5, "x")
    assert_close(npy(F), npy(Fr), 2e-5, "F")
    assert_close(npy(f), npy(fr), 5e-5, "f")
    # the model reproduces the step it was taken around
    tau = torch.cat((x[:-1], u[:-1]), dim=2)
    nxt = torch.einsum("tbij,tbj->tbi", F, tau) + f
    assert_close(npy(nxt), npy(x[1:]), 2e-5, "F tau + f")


def test_imitation_step_config4_gradients_reach_the_cost_parameters():
    """BASELINE.json configs[3] (env_dx/il_env.py:104-158, il_exp.py:213-302), small batch: learnable cost
    q = sigmoid(logit), p = sqrt(q) * learn_p, true pendulum, update_dynamics=False - the imitation loss must send a
    finite, non-zero gradient through MPCstep.backward (dC, dc) to both parameter vectors"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "imitation_step", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "imitation_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.imitation_step(B=64, T=20, max_iter=6, seed=1)
    assert np.isfinite(r["loss"]) and r["loss"] > 0
    for g in (r["g_logit"], r["g_p"]):
        assert np.isfinite(g).all() and np.abs(g).max() > 0


def test_il_env_and_cost_net_one_rmsprop_step():
    """IL_Env.populate_data / .mpc (env_dx/il_env.py:71-158) and Pendulum_Net_cost_logit (pendulum_net.py:12-39):
    expert data under the true cost, one imitation update with the experiment's optimiser (RMSprop, lr 1e-2,
    alpha 0.5 - il_exp.py:230-240) moves both parameter vectors with finite values"""
    from chainer_differentiable_mpc_amd import IL_Env, Pendulum_Net_cost_logit
    env = IL_Env('pendulum', lqr_iter=6, mpc_T=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env.populate_data(n_train=24, n_val=4, n_test=4, seed=0)
        assert list(env.train_data.shape) == [24, 20, 4] and list(env.test_data.shape) == [4, 20, 4]
        # the data set starts where sample_xinit said and respects the torque limit
        np.random.seed(0)
        x0 = IL_Env.sample_xinit(32)
        assert_close(npy(env.train_data[:, 0, :3]), x0[:24], 1e-6, "x_init of the data set")
        assert float(env.train_data[:, :, 3].abs().max()) <= 2.0 + 1e-6
        net = Pendulum_Net_cost_logit(4)
        opt = torch.optim.RMSprop(net.parameters(), lr=1e-2, alpha=0.5)
        xinit = env.train_data[:, 0, :3]
        x_mpc, u_mpc = net(xinit, env)
        loss = ((torch.cat((x_mpc, u_mpc), dim=2).transpose(0, 1) - env.train_data) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    for prm in (net.learn_q_logit, net.learn_p):
        assert torch.isfinite(prm).all() and torch.isfinite(prm.grad).all()
    assert float(net.learn_q_logit.grad.abs().max()) > 0


def _run_both(make_solver, args):
    out = []
    for device_loop in (True, False):
        solver = make_solver(device_loop)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, costs = solver(args())
        if device_loop:      # the chain ran (a refused dmpc_box_ddp falls back to the host loop without a word)
            from chainer_differentiable_mpc_amd import _lib
            assert "box_ddp_summary_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
        out.append((npy(x), npy(u), npy(costs), solver.status, solver.n_iter))
    return out


@pytest.mark.parametrize("max_iter", [1, 4, 10])
def test_device_loop_matches_host_loop_pendulum(max_iter):
    """`dmpc_box_ddp` (stop tests and best-so-far selection on the device) against the host loop over MPCstep
    objects: same kernels in the same order, so iterates, status and iteration count agree"""
    B, T = 128, 20
    dx, x0, Q, pv = pendulum_problem(B, T, seed=3)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
    dev_, host = _run_both(
        lambda dl: BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=max_iter, exit_unconverged=False, quiet=True,
                          device_loop=dl, **kw),
        lambda: (dev(x0), QuadCost(dev(Q), dev(pv)), dx))
    assert dev_[3] == host[3] and dev_[4] == host[4], (dev_[3:], host[3:])
    assert_close(dev_[1], host[1], 1e-5, "u")
    assert_close(dev_[0], host[0], 1e-5, "x")
    assert_close(dev_[2], host[2], 1e-5, "costs")


@pytest.mark.parametrize("B,max_iter", [(128, 10), (1024, 10), (260, 4), (128, 1), (8, 3)])
def test_one_launch_iterations_are_bit_identical_to_sweep_and_search_as_two_launches(B, max_iter, monkeypatch):
    """round 5: an iteration of the pendulum chain is ONE launch (box_ddp_pendulum_iter_kernel: a workgroup sweeps its four
    trajectories, then searches them, beside the previous iteration's bookkeeping) - the same kernels' bodies in the same order,
    so every output, the status, the iteration count and the flags must be those of the two-launch chain
    (DMPC_NO_DDP_ITER_FUSED=1), bit for bit; incl. a run that stops early (the search of the iteration after the stop runs ahead
    of the flag and must leave no trace)"""
    from chainer_differentiable_mpc_amd import _lib
    T = 20
    dx, x0, Q, pv = pendulum_problem(B, T, seed=11)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter, max_iter=max_iter,
              exit_unconverged=False, quiet=True, graph=False)
    outs = {}
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("DMPC_NO_DDP_ITER_FUSED", "1")
        else:
            monkeypatch.delenv("DMPC_NO_DDP_ITER_FUSED")
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, c = solver((dev(x0), QuadCost(dev(Q), dev(pv)), dx))
        outs[mode] = (x, u, c, solver.status, solver.n_iter, solver.info.clone())
    for a, b in zip(outs["two"][:3], outs["one"][:3]):
        assert torch.equal(a, b)
    assert outs["two"][3:5] == outs["one"][3:5], (outs["two"][3:5], outs["one"][3:5])
    assert torch.equal(outs["two"][5], outs["one"][5])
    # an early stop: a loose eps ends the loop after a few iterations in both forms, with the same iterate
    kw2 = dict(kw, eps=0.5, max_iter=10)
    res = {}
    for mode in ("two", "one"):
        if mode == "two":
            monkeypatch.setenv("DMPC_NO_DDP_ITER_FUSED", "1")
        else:
            monkeypatch.delenv("DMPC_NO_DDP_ITER_FUSED")
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, **kw2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, c = solver((dev(x0), QuadCost(dev(Q), dev(pv)), dx))
        res[mode] = (x, u, c, solver.status, solver.n_iter, solver.info.clone())
    for a, b in zip(res["two"][:3], res["one"][:3]):
        assert torch.equal(a, b)
    assert res["two"][3:5] == res["one"][3:5]
    assert torch.equal(res["two"][5], res["one"][5])
    print("B=%d: %s after %d iterations; early stop: %s after %d" % (B, outs["one"][3], outs["one"][4], res["one"][3], res["one"][4]))


@pytest.mark.parametrize("B", [128, 1024])
def test_a_solve_called_again_on_the_same_buffers_replays_its_chain_from_a_graph(B):
    """verdict r04 item 2, the minimum asked: the chain of 22-23 launches captured as a hipGraph INSIDE `BoxDDP` by default.
    First call on a set of buffers: the chain, launched; second: recorded and replayed; from the third: replayed.  Every call
    must return what a solver with `graph=False` returns (bit for bit: the same kernels), in buffers of its own, and must follow
    the buffers' CONTENTS (an MPC loop updates its state in place)"""
    T = 20
    dx, x0, Q, pv = pendulum_problem(B, T, seed=3)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter, max_iter=10,
              exit_unconverged=False, quiet=True)
    x0d, cost = dev(x0), QuadCost(dev(Q), dev(pv))
    plain = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, graph=False, **kw)
    graph = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, **kw)
    assert graph.graph and not plain._graphs
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = plain((x0d, cost, dx))
        ref_status, ref_iter = plain.status, plain.n_iter
        kept = []
        for call in range(4):
            out = graph((x0d, cost, dx))
            assert graph.status == ref_status and graph.n_iter == ref_iter
            for a, b in zip(out, ref):
                assert torch.equal(a, b), "call %d" % call
            kept.append(out)
        (entry,) = graph._graphs.values()
        assert entry[0] is not None                                    # recorded at the second call
        assert len({o[0].data_ptr() for o in kept}) == len(kept)       # every call's results live in buffers of their own
        # other CONTENTS in the same buffers: the replay reads them
        x0d.copy_(dev(pendulum_problem(B, T, seed=9)[1]))
        ref2 = plain((x0d, cost, dx))
        out2 = graph((x0d, cost, dx))
        assert len(graph._graphs) == 1
        for a, b in zip(out2, ref2):
            assert torch.equal(a, b)
        assert not torch.equal(out2[0], kept[0][0])
        for a, b in zip(kept[0], ref):
            assert torch.equal(a, b)                                   # (the earlier results did not move)
        # other BUFFERS: launched directly, a second entry appears
        out3 = graph((x0d.clone(), cost, dx))
        assert len(graph._graphs) == 2
        for a, b in zip(out3, ref2):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B", [6, 260, 2304], ids=["ragged-plain-chain", "two-bookkeeping-workgroups", "keep-kernel"])
def test_device_loop_matches_host_loop_pendulum_other_batches(B):
    """the chain's other shapes: a batch that is not a multiple of four (register-bank kernels, one bookkeeping launch
    per iteration), one whose bookkeeping is shared by two workgroups with a partly filled second one, and one too
    large for the bookkeeping workgroups to move the kept trajectories themselves (box_ddp_keep_kernel)"""
    T = 20
    dx, x0, Q, pv = pendulum_problem(B, T, seed=4)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
    dev_, host = _run_both(
        lambda dl: BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=4, exit_unconverged=False, quiet=True,
                          device_loop=dl, **kw),
        lambda: (dev(x0), QuadCost(dev(Q), dev(pv)), dx))
    assert dev_[3] == host[3] and dev_[4] == host[4], (dev_[3:], host[3:])
    assert_close(dev_[1], host[1], 1e-5, "u")
    assert_close(dev_[0], host[0], 1e-5, "x")
    assert_close(dev_[2], host[2], 1e-5, "costs")


def test_device_loop_matches_host_loop_pendulum_long_horizon():
    """T = 40: past the horizon the wavefront-per-trajectory line search keeps in LDS - the lane-per-candidate kernel
    with its LDS-DMA ring runs (both loops)"""
    B, T = 32, 40
    dx, x0, Q, pv = pendulum_problem(B, T, seed=6)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter)
    dev_, host = _run_both(
        lambda dl: BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, max_iter=3, exit_unconverged=False, quiet=True,
                          device_loop=dl, **kw),
        lambda: (dev(x0), QuadCost(dev(Q), dev(pv)), dx))
    assert dev_[3] == host[3] and dev_[4] == host[4], (dev_[3:], host[3:])
    assert_close(dev_[1], host[1], 1e-5, "u")
    assert_close(dev_[0], host[0], 1e-5, "x")
    assert_close(dev_[2], host[2], 1e-5, "costs")
    # one iLQR step of that kernel pair against the oracle (from the zero-control iterate)
    lo, hi = np.full((T, B, 1), dx.lower), np.full((T, B, 1), dx.upper)
    uk = np.zeros((T, B, 1))
    xk = obox.get_traj(T, uk, x0, obox.pendulum_step)
    Fm, fm = obox.pendulum_linearize(xk, uk)
    cost_o = ompc.QuadCost(Q, pv)
    xo, uo, _, fo, Ko, ko = ompc.mpc_forward(Q, pv, Fm, fm, uk, xk, lo, hi, cost_o, obox.pendulum_step, dx.linesearch_decay,
                                             dx.max_linesearch_iter, T, 3, 1, need_expand=True, batch_coupled=False)
    with torch.no_grad():
        ud = dev(uk)
        xd, Fd, fd = dx.rollout_linearize(dev(x0), ud)
        step = MPCstep(controls=ud, T=T, u_upper=dev(hi), u_lower=dev(lo), n_batch=B, n_state=3, n_ctrl=1,
                       current_states=xd, true_cost=QuadCost(dev(Q), dev(pv)), true_dynamics=dx,
                       ls_decay=dx.linesearch_decay, max_ls_iter=dx.max_linesearch_iter, need_expand=True)
        xn, un = step.forward((xd[0], dev(Q), dev(pv), Fd, fd))
    old = ompc.get_cost(T, uk, cost_o, xk)

    def candidates(rows, alpha):
        xc, uc, _ = ompc.ls_rollout(Ko[:, rows], ko[:, rows], uk[:, rows], xk[:, rows], lo[:, rows], hi[:, rows],
                                    ompc.QuadCost(Q[:, rows], pv[:, rows]), obox.pendulum_step, np.full(len(rows), alpha), T)
        return xc, uc

    assert_step_close(npy(un), npy(xn), uo, xo, old, fo.costs, candidates, TOL_STEP_PENDULUM, "step at T = 40")


@pytest.mark.parametrize("shape", [(16, 8, 3, 2, 0.3), (64, 12, 8, 2, 0.5), (5, 6, 4, 2, 10.0), (12, 6, 5, 3, 0.5),
                                   (6, 5, 5, 9, 0.3), (4, 5, 66, 3, 0.4)])
def test_device_loop_matches_host_loop_lindx(shape):
    # the nominal rollout is a kernel here and torch ops there: rounding differs, the iteration amplifies it
    # (the last two shapes - 9 controls, 70 columns - run on the tiled kernels: the device loop took them in round 5)
    B, T, nx, nu, bound = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=11, with_f=True)
    dev_, host = _run_both(
        lambda dl: BoxDDP(T, -bound, bound, B, nx, nu, None, max_iter=10, quiet=True, device_loop=dl),
        lambda: (dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
    assert dev_[3] == host[3] and dev_[4] == host[4], (dev_[3:], host[3:])
    assert_close(dev_[1], host[1], TOL_PRIMAL, "u")
    assert_close(dev_[0], host[0], TOL_PRIMAL, "x")
    assert_close(dev_[2], host[2], TOL_PRIMAL, "costs")


def test_device_loop_stops_early_and_freezes_the_result():
    """an unconstrained LQ problem converges at the second step: later launches of the chain must be no-ops"""
    B, T, nx, nu = 32, 10, 4, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=5, with_f=True)
    solver = BoxDDP(T, -1e3, 1e3, B, nx, nu, None, max_iter=10, quiet=True)
    x, u, costs = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(dev(p["F"]), dev(p["f"]))))
    assert solver.status == "Converged" and solver.n_iter < 10
    from oracle import lqr as olqr
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    assert_close(npy(u), ur, 1e-4, "u")
    assert_close(npy(x), xr, 1e-4, "x")


def test_device_loop_gradient_through_the_final_node():
    """MpcNet-style use: the gradient is carried by the no-op MPCstep node after the device loop"""
    B, T, nx, nu = 8, 6, 3, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=2, with_f=True)
    grads = []
    for dl in (True, False):
        F = dev(p["F"]).clone().requires_grad_(True)
        solver = BoxDDP(T, -0.5, 0.5, B, nx, nu, None, max_iter=10, quiet=True, device_loop=dl,
                        detach_unconverged=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, _ = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), LinDx(F, dev(p["f"]))))
        (x.sum() + u.sum()).backward()
        grads.append(npy(F.grad))
    assert np.abs(grads[0]).max() > 0
    assert_close(grads[0], grads[1], TOL_COSTATE, "dF")
