#!/usr/bin/env python
"""Generate golden vectors by running the UNMODIFIED reference
(/root/reference, pfnet-research/chainer-differentiable-mpc) in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference imports `chainer` at module scope; `oracle/refshim` supplies a
forward-only numpy stand-in (our code) and restores torch<=1.2 `lu_solve` semantics
(see oracle/refshim/load_reference.py).  Inputs are NOT stored where they can be
regenerated from a seed by `chainer_differentiable_mpc_amd.synthetic` (legacy
RandomState, stable across numpy versions); an `in_checksum` guards against drift.
Everything stored is data (inputs / expected outputs); no reference source text.

Files written:
  lqr_<B>_<T>_<nx>_<nu>_<f|nof>.npz   rows A (x,u,Ks,ks) and B (five KKT gradients)
  pnqp_n<n>.npz                        row C: 4-tuple, batched and per-row runs
  lu_n<n>.npz                          row D: LU/pivots/solve (2-D and 3-D rhs)
  mpc_<B>_<T>_<nx>_<nu>_<exp|noexp>.npz rows E/F: MPCstep forward+backward, LQR_active
  boxddp_trace.npz                     BoxDDP + LinDx/QuadCost per-iteration trace
  anchors.npz                          reference outputs on the notebook problems
  pendulum.npz                         env_dx/pendulum.py: PendulumDx.forward, get_true_obj; il_env.sample_xinit;
                                       mpc/approximate.py linearize_dynamics around the pendulum (torques on/over the clamp)
  approx_cost.npz                      mpc/approximate.py approximate_cost on a quadratic and a non-quadratic cost
  pendulum_boxddp.npz                  BoxDDP around the non-linear PendulumDx (config 2 family): iterates after 1..4 steps
  imitation_16.npz                     config 4 chain, small: Pendulum_Net_cost_logit -> IL_Env.mpc -> loss -> d logit, d p
  imitation_step_1024.npz              config 4 at B=1024, T=20: one MPCstep from a common iterate + the no-op gradient node
  imitation_step_1024_it1.npz          the same step from the iterate after ONE box-DDP iteration (most rows have a resolvable margin)
  imitation_loop_16.npz                config 4's loop: three RMSprop updates with the evaluation pass's warm-start carry-over
  pnqp_n8_b256.npz                     PNQP n=8, B=256: the batch whose rows fork under the batch-global termination
  mpcnet_experiment.npz                the reference's experiment_mpc/MpcNet.py, first training iteration at its own sizes ((3,3), B=128, T=5)
"""
import io
import os
import sys
import warnings
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.refshim import load_reference  # noqa: E402

# import the generator module by path: keeps torch / the HIP loader out of this script
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location(
    "dmpc_synthetic", os.path.join(ROOT, "chainer_differentiable_mpc_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synthetic)

LQR_CASES = [(1, 5, 3, 1), (3, 6, 4, 2), (8, 10, 8, 2), (2, 6, 32, 8), (5, 7, 2, 1), (4, 5, 3, 2)]
MPC_CASES = [(1, 5, 3, 1, 0.625), (3, 6, 4, 2, 0.5), (8, 10, 8, 2, 0.375), (6, 8, 3, 1, 0.75), (4, 5, 3, 2, 0.3125)]


def checksum(d):
    return float(sum(np.abs(v).sum() for v in d.values() if v is not None))


def arr(v):
    return v.array if hasattr(v, "array") else np.asarray(v)


def gen_lqr(ref):
    V = ref.chainer.Variable
    for (B, T, nx, nu) in LQR_CASES:
        for with_f in (True, False):
            seed = 1000 + 17 * B + T + nx + nu
            p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed, with_f=with_f)
            f = None if p["f"] is None else V(p["f"])
            rec = ref.lqr_recursion.LqrRecursion(V(p["x_init"]), V(p["C"]), V(p["c"]), V(p["F"]), f, T, nx, nu)
            Ks, ks = rec.backward()
            x, u = rec.forward(Ks, ks)
            rng = np.random.RandomState(seed + 1)
            gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
            gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
            node = ref.differentiable_lqr.DiffLqr(T, B, nx, nu)
            x2, u2 = node.apply((p["x_init"], p["C"], p["c"], p["F"], p["f"]))
            assert np.array_equal(arr(x2), arr(x)) and np.array_equal(arr(u2), arr(u))
            dx0, dC, dc, dF, df = node.backward((0, 1, 2, 3, 4), (V(gx), V(gu)))
            name = "lqr_%d_%d_%d_%d_%s.npz" % (B, T, nx, nu, "f" if with_f else "nof")
            np.savez_compressed(
                os.path.join(HERE, name), B=B, T=T, nx=nx, nu=nu, seed=seed, with_f=with_f,
                in_checksum=checksum(p), grad_x=gx, grad_u=gu,
                x=arr(x), u=arr(u), Ks=np.stack([arr(k) for k in Ks]), ks=np.stack([arr(k) for k in ks]),
                d_x_init=arr(dx0), dC=arr(dC), dc=arr(dc), dF=arr(dF), df=arr(df))
            print("wrote", name)


def gen_lu(ref):
    for n in (2, 3, 4, 8):
        rng = np.random.RandomState(50 + n)
        B = 6
        A = rng.randn(B, n, n).astype(np.float32).astype(np.float64)
        b2 = rng.randn(B, n).astype(np.float32).astype(np.float64)
        b3 = rng.randn(B, n, 3).astype(np.float32).astype(np.float64)
        LU, piv = ref.util.xpbatch_lu_factor(A)
        x2 = ref.util.xpbatch_lu_solve((LU, piv), b2)
        x3 = ref.util.xpbatch_lu_solve((LU, piv), b3)
        np.savez_compressed(os.path.join(HERE, "lu_n%d.npz" % n), A=A, b2=b2, b3=b3, LU=LU, piv=piv,
                            x2=x2, x3=x3)
        print("wrote lu_n%d.npz" % n)


def gen_pnqp(ref):
    for n in (1, 2, 4, 8):
        B = 16
        p = synthetic.make_box_qp(B, n, seed=200 + n, bound=0.5)
        rng = np.random.RandomState(300 + n)
        warm = (0.3 * rng.randn(B, n)).astype(np.float32).astype(np.float64)
        out = {}
        for tag, x0 in (("cold", None), ("warm", warm)):
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                x, fac, idx_f, it = ref.pnqp.PNQP(p["H"], p["q"], p["lower"], p["upper"], x_init=x0, n_iter=20)
                out[tag + "_warned"] = len(w) > 0
            out[tag + "_x"] = x
            out[tag + "_idx_f"] = idx_f
            out[tag + "_it"] = it
            if n == 1:
                out[tag + "_Hf"] = fac
            else:
                out[tag + "_LU"], out[tag + "_piv"] = fac
            # per-row runs (the reference called with a batch of one)
            xs, its, idxs, warned = [], [], [], []
            for b in range(B):
                x0b = None if x0 is None else x0[b:b + 1]
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    xb, facb, idxb, itb = ref.pnqp.PNQP(p["H"][b:b + 1], p["q"][b:b + 1], p["lower"][b:b + 1],
                                                        p["upper"][b:b + 1], x_init=x0b, n_iter=20)
                    warned.append(len(w) > 0)
                xs.append(xb)
                its.append(itb)
                idxs.append(idxb)
            out[tag + "_row_x"] = np.concatenate(xs)
            out[tag + "_row_it"] = np.array(its)
            out[tag + "_row_idx_f"] = np.concatenate(idxs)
            out[tag + "_row_warned"] = np.array(warned)
        np.savez_compressed(os.path.join(HERE, "pnqp_n%d.npz" % n), n=n, B=B, seed=200 + n, warm=warm,
                            in_checksum=checksum(p), **out)
        print("wrote pnqp_n%d.npz" % n)
    # the notebook problem (experiment_mpc/Projected_Newton_Quadratic_Programming.py:20-47)
    H = np.array([[[7.9325, 4.9520, 1.0314, 0.2282], [4.9520, 8.7746, 1.7916, 3.3622],
                   [1.0314, 1.7916, 4.2824, -2.5979], [0.2282, 3.3622, -2.5979, 6.7064]],
                  [[3.4423, -1.9137, -0.9978, -4.4905], [-1.9137, 6.7254, 3.3720, 1.7444],
                   [-0.9978, 3.3720, 3.5695, -0.9766], [-4.4905, 1.7444, -0.9766, 13.0806]]])
    q = np.array([[-0.8277, 8.5116, -12.1597, 17.9497], [-3.5764, -5.3455, -3.2465, 4.3960]])
    lower = np.array([[-0.2843, -0.0063, -0.1808, -0.6669], [-0.1359, -0.3629, -0.2125, -0.0121]])
    upper = np.array([[0.1345, 0.0307, 0.0277, 0.9418], [0.6205, 0.2703, 0.4023, 0.2560]])
    x, (LU, piv), idx_f, it = ref.pnqp.PNQP(H, q, lower, upper)
    return dict(pnqp_H=H, pnqp_q=q, pnqp_lower=lower, pnqp_upper=upper, pnqp_x=x, pnqp_it=it,
                pnqp_idx_f=idx_f)


def nominal(p, T, nx, nu, B, seed, bound):
    """A feasible nominal control sequence and its rollout under (F, f)."""
    rng = np.random.RandomState(seed)
    u = np.clip(0.5 * rng.randn(T, B, nu), -bound, bound).astype(np.float32).astype(np.float64)
    xs = [p["x_init"]]
    for t in range(T - 1):
        xu = np.concatenate((xs[t], u[t]), axis=1)
        xn = np.einsum("bij,bj->bi", p["F"][t], xu)
        if p["f"] is not None:
            xn = xn + p["f"][t]
        xs.append(xn)
    # states stay the exact float64 rollout: the reference's line search compares the
    # re-rolled cost with the cost of `states` and never terminates if they differ at alpha->0
    x = np.stack(xs)
    return x, u


def gen_mpc(ref):
    U = ref.util
    for (B, T, nx, nu, bound) in MPC_CASES:
        for need_expand in (True, False):
            seed = 2000 + 13 * B + T + nx + nu
            p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed, with_f=True)
            x_nom, u_nom = nominal(p, T, nx, nu, B, seed + 1, bound)
            lo = -bound * np.ones((T, B, nu))
            hi = bound * np.ones((T, B, nu))
            step = ref.mpc_step.MPCstep(u_nom, T, hi, lo, B, nx, nu, x_nom, U.QuadCost(p["C"], p["c"]),
                                        U.LinDx(p["F"], p["f"]), ls_decay=0.2, max_ls_iter=5,
                                        need_expand=need_expand)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                x, u = step.apply((x_nom[0], p["C"], p["c"], p["F"], p["f"]))
            x, u = arr(x), arr(u)
            rng = np.random.RandomState(seed + 2)
            gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
            gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
            dx0, dC, dc, dF, df = step.backward((0, 1, 2, 3, 4), (gx, gu))
            active = (np.abs(u - lo) <= 1e-8) | (np.abs(u - hi) <= 1e-8)
            la = ref.active_constrained_lqr.LQR_active(np.zeros_like(x_nom[0]), p["C"],
                                                       -np.concatenate((gx, gu), axis=2), p["F"], None,
                                                       T, nx, nu, u_zero_Index=active)
            adx, adu = la.solve_recursion()
            fo = step.for_out
            # the same step with every trajectory as a batch of one: removes the reference's batch-global
            # PNQP termination (pnqp.py:139-144,172,187) - the semantics of the fused GPU kernels
            rows = {k: [] for k in ("x", "u", "costs", "d_x_init", "dC", "dc", "dF", "df", "n_qp")}
            for b in range(B):
                sl = slice(b, b + 1)
                st = ref.mpc_step.MPCstep(u_nom[:, sl], T, hi[:, sl], lo[:, sl], 1, nx, nu, x_nom[:, sl],
                                          U.QuadCost(p["C"][:, sl], p["c"][:, sl]), U.LinDx(p["F"][:, sl], p["f"][:, sl]),
                                          ls_decay=0.2, max_ls_iter=5, need_expand=need_expand)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    xb, ub = st.apply((x_nom[0, sl], p["C"][:, sl], p["c"][:, sl], p["F"][:, sl], p["f"][:, sl]))
                gb = st.backward((0, 1, 2, 3, 4), (gx[:, sl], gu[:, sl]))
                rows["x"].append(arr(xb)); rows["u"].append(arr(ub)); rows["costs"].append(st.for_out.costs)
                rows["n_qp"].append(st.back_out.n_total_qp_iter)
                for key, val in zip(("d_x_init", "dC", "dc", "dF", "df"), gb):
                    rows[key].append(arr(val))
            row = dict(row_x=np.concatenate(rows["x"], axis=1), row_u=np.concatenate(rows["u"], axis=1),
                       row_costs=np.concatenate(rows["costs"]), row_n_qp=np.array(rows["n_qp"]),
                       row_d_x_init=np.concatenate(rows["d_x_init"], axis=0), row_dC=np.concatenate(rows["dC"], axis=1),
                       row_dc=np.concatenate(rows["dc"], axis=1), row_dF=np.concatenate(rows["dF"], axis=1),
                       row_df=np.concatenate(rows["df"], axis=1))
            name = "mpc_%d_%d_%d_%d_%s.npz" % (B, T, nx, nu, "exp" if need_expand else "noexp")
            np.savez_compressed(
                os.path.join(HERE, name), B=B, T=T, nx=nx, nu=nu, seed=seed, bound=bound,
                need_expand=need_expand, in_checksum=checksum(p), x_nom=x_nom, u_nom=u_nom,
                grad_x=gx, grad_u=gu, x=x, u=u, n_total_qp_iter=step.back_out.n_total_qp_iter,
                objs=fo.objs, full_du_norm=fo.full_du_norm, alpha_du_norm=fo.alpha_du_norm,
                mean_alphas=fo.mean_alphas, costs=fo.costs, active=active,
                d_x_init=arr(dx0), dC=arr(dC), dc=arr(dc), dF=arr(dF), df=arr(df),
                active_dx=adx, active_du=adu, **row)
            print("wrote", name, "sat=%.2f" % active.mean(), "qp_it", step.back_out.n_total_qp_iter,
                  "mean_alpha", fo.mean_alphas)


def gen_boxddp(ref):
    if ref.box_ddp is None:
        print("box_ddp not importable:", ref.box_ddp_error)
        return
    U = ref.util
    B, T, nx, nu = 4, 5, 3, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=77, with_f=True)
    V = ref.chainer.Variable
    ddp = ref.box_ddp.BoxDDP(T, -0.25, 0.25, B, nx, nu, None, max_iter=10, verbose=False)
    buf = io.StringIO()
    with warnings.catch_warnings(record=True) as w, redirect_stdout(buf):
        warnings.simplefilter("always")
        x, u, costs = ddp((V(p["x_init"]), U.QuadCost(V(p["C"]), V(p["c"])), U.LinDx(V(p["F"]), V(p["f"]))))
    np.savez_compressed(os.path.join(HERE, "boxddp_trace.npz"), B=B, T=T, nx=nx, nu=nu, seed=77,
                        bound=0.25, in_checksum=checksum(p), x=arr(x), u=arr(u), costs=arr(costs),
                        stdout=buf.getvalue(), n_warn=len(w))
    print("wrote boxddp_trace.npz:", buf.getvalue().strip().replace("\n", " | "))


def gen_anchors(ref, extra):
    V = ref.chainer.Variable
    out = dict(extra)
    # examples/LQR_recursion_solver_one_variable.py:24-33
    T, nx, nu = 20, 2, 1
    F = np.tile(np.array([[1.0, 1.0, 0], [0, 1.0, 1.0]]), (T, 1, 1, 1))
    c = np.zeros((T, 1, 3))
    C = np.tile(np.array([[1.0, 0, 0], [0, 0, 0], [0, 0, 10]]), (T, 1, 1, 1))
    x0 = np.array([[1.0, 0.0]])
    rec = ref.lqr_recursion.LqrRecursion(V(x0), V(C), V(c), V(F), None, T, nx, nu)
    Ks, ks = rec.backward()
    x, u = rec.solve_recursion()
    out.update(onevar_Ks=np.stack([arr(k) for k in Ks]), onevar_x=arr(x), onevar_u=arr(u))
    # examples/Boyd_lqr.py:24-41
    T, nx, nu = 51, 3, 1
    F = np.tile(np.array([[1.0, 0, 0, 1], [1, 1.0, 0, 0], [0, 1, 1, 0]]), (T, 1, 1, 1))
    c = np.zeros((T, 1, 4))
    C = np.tile(np.diag([0, 0, 1.0, 1.0]), (T, 1, 1, 1))
    C[T - 1, 0, 3, 3] = 0.00000000000001
    x0 = np.array([[0.5428, 0.7633, 0.3504]])
    rec = ref.lqr_recursion.LqrRecursion(V(x0), V(C), V(c), V(F), None, T, nx, nu)
    Ks, ks = rec.backward()
    x, u = rec.solve_recursion()
    out.update(boyd_Ks=np.stack([arr(k) for k in Ks]), boyd_x=arr(x), boyd_u=arr(u))
    np.savez_compressed(os.path.join(HERE, "anchors.npz"), **out)
    print("wrote anchors.npz")


def _f32(a):
    return np.asarray(a).astype(np.float32).astype(np.float64)


def gen_pendulum(ref):
    V = ref.chainer.Variable
    dx = ref.pendulum.PendulumDx()
    rng = np.random.RandomState(11)
    B = 96
    th = rng.uniform(-np.pi, np.pi, B)
    th[:4] = [np.pi, -np.pi, 0.0, np.pi / 2]
    x = _f32(np.stack((np.cos(th), np.sin(th), rng.uniform(-3, 3, B)), axis=1))
    u = _f32(rng.uniform(-3.0, 3.0, (B, 1)))
    u[:8, 0] = [2.0, -2.0, 2.0, -2.0, 0.0, 2.5, -2.5, 1.9999999]
    nxt = arr(dx(V(x), V(u)))
    q, p = dx.get_true_obj()
    np.random.seed(0)
    xi128 = ref.il_env.IL_Env.sample_xinit(128)
    np.random.seed(0)
    xi1024 = ref.il_env.IL_Env.sample_xinit(1024)
    # linearize_dynamics (mpc/approximate.py:77-119) along a trajectory with torques inside, ON and beyond the clamp
    T, Bl = 20, 12
    ul = _f32(rng.uniform(-2.6, 2.6, (T, Bl, 1)))
    ul[:, 0, 0] = 2.0
    ul[:, 1, 0] = -2.0
    ul[::2, 2, 0] = 2.0
    np.random.seed(3)
    x0 = _f32(ref.il_env.IL_Env.sample_xinit(Bl))
    xs = [x0]
    for t in range(T - 1):
        xs.append(arr(dx(V(xs[t]), V(ul[t]))))
    xl = np.stack(xs)
    Fl, fl = ref.approximate.linearize_dynamics(V(xl), V(ul), dx)
    np.savez_compressed(os.path.join(HERE, "pendulum.npz"), x=x, u=u, next=nxt, q=q, p=p, params=arr(dx.params),
                        dt=dx.dt, max_torque=dx.max_torque, lower=dx.lower, upper=dx.upper, mpc_eps=dx.mpc_eps,
                        linesearch_decay=dx.linesearch_decay, max_linesearch_iter=dx.max_linesearch_iter,
                        xinit128=xi128, xinit1024_head=xi1024[:64], xinit1024_sum=xi1024.sum(axis=0),
                        lin_x=xl, lin_u=ul, lin_F=arr(Fl), lin_f=arr(fl))
    print("wrote pendulum.npz; d next/d u at u=+-2:", arr(Fl)[0, :2, 2, 3])


def gen_approx_cost(ref):
    V = ref.chainer.Variable
    F = ref.chainer.functions
    rng = np.random.RandomState(21)
    T, B, nx, nu = 4, 5, 3, 2
    ns = nx + nu
    x = _f32(rng.randn(T, B, nx))
    u = _f32(rng.randn(T, B, nu))
    L = rng.randn(ns, ns)
    Cq = _f32(L @ L.T + ns * np.eye(ns))
    cq = _f32(rng.randn(ns))
    out = dict(x=x, u=u, Cq=Cq, cq=cq)

    def quad(tau):          # 1/2 tau' C tau + c' tau, per batch row
        return 0.5 * F.sum(F.matmul(tau, Cq) * tau, axis=1) + F.sum(tau * cq, axis=1)

    def nonquad(tau):       # smooth, non-quadratic: sqrt(1 + |tau|^2) + sum sin(tau_i) tau_{i+1}
        r = F.sqrt(1.0 + F.sum(tau ** 2, axis=1))
        return r + F.sum(F.sin(tau[:, :-1]) * tau[:, 1:], axis=1)

    for name, fn in (("quad", quad), ("nonquad", nonquad)):
        H, g, cst = ref.approximate.approximate_cost(V(x), V(u), fn)
        out[name + "_H"], out[name + "_g"], out[name + "_cost"] = arr(H), arr(g), arr(cst)
    np.savez_compressed(os.path.join(HERE, "approx_cost.npz"), **out)
    print("wrote approx_cost.npz")


def _tiled_cost(ref, q, p, T, B):
    Q = np.tile(np.diag(q), (T, B, 1, 1))
    pv = np.tile(p, (T, B, 1))
    return ref.util.QuadCost(ref.chainer.Variable(Q), ref.chainer.Variable(pv))


def gen_pendulum_boxddp_b128(ref):
    """the same at config 2's own batch (BASELINE.json configs[1]: B=128, T=20) -> pendulum_boxddp_b128.npz"""
    gen_pendulum_boxddp(ref, B=128, name="pendulum_boxddp_b128.npz")


def gen_pendulum_boxddp(ref, B=16, name="pendulum_boxddp.npz"):
    """BoxDDP.forward (mpc/box_ddp.py:93-291) with the non-linear PendulumDx: linearize_dynamics by chainer.grad,
    MPCstep with the pendulum as the true dynamics callable.  The returned iterate after k = 1..4 outer iterations."""
    V = ref.chainer.Variable
    dx = ref.pendulum.PendulumDx()
    q, p = dx.get_true_obj()
    T = 20
    np.random.seed(5)
    x0 = _f32(ref.il_env.IL_Env.sample_xinit(B))
    out = dict(B=B, T=T, x_init=x0, q=q, p=p)
    for k in (1, 2, 3, 4):
        ddp = ref.box_ddp.BoxDDP(T=T, u_lower=dx.lower, u_upper=dx.upper, n_batch=B, n_state=3, n_ctrl=1, u_init=None,
                                 eps=dx.mpc_eps, max_iter=k, verbose=False, exit_unconverged=False,
                                 detach_unconverged=True, line_search_decay=dx.linesearch_decay,
                                 max_line_search_iter=dx.max_linesearch_iter, update_dynamics=True)
        buf = io.StringIO()
        with warnings.catch_warnings(), redirect_stdout(buf):
            warnings.simplefilter("ignore")
            x, u, costs = ddp((V(x0), _tiled_cost(ref, q, p, T, B), dx))
        out["x_%d" % k], out["u_%d" % k], out["costs_%d" % k] = arr(x), arr(u), arr(costs)
        out["stdout_%d" % k] = buf.getvalue()
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote %s; mean cost after 1..4:" % name, [float(out["costs_%d" % k].mean()) for k in (1, 2, 3, 4)])


def gen_imitation(ref):
    """config 4 (env_dx/il_exp.py:213-302 around env_dx/il_env.py:104-158 and pendulum_net.py:12-39)"""
    import importlib
    ch = ref.chainer
    V, F = ch.Variable, ch.functions
    pnet = importlib.import_module("pendulum_net")
    # ---- (a) the whole chain at B=16: expert under the true cost, learner with a perturbed cost, 5 iLQR iterations
    B, T = 16, 20
    env = ref.il_env.IL_Env('pendulum', lqr_iter=5, mpc_T=T)
    np.random.seed(7)
    xinit = _f32(env.sample_xinit(B))
    tq, tp = env.true_dx.get_true_obj()
    buf = io.StringIO()
    with warnings.catch_warnings(), redirect_stdout(buf):
        warnings.simplefilter("ignore")
        ex, eu = env.mpc(env.true_dx, xinit, tq, tp, update_dynamics=True)
        net = pnet.Pendulum_Net_cost_logit(4)
        net.learn_q_logit.array[:] = np.array([0.5, -0.25, -1.0, -3.0])
        net.learn_p.array[:] = np.array([-0.75, 0.125, 0.0625, 0.0])
        warm = np.zeros((B, T, 1))
        nom_x, nom_u = net(xinit, env, warm)
    us = np.transpose(arr(eu), (1, 0, 2))
    nu_ = F.transpose(nom_u, axes=(1, 0, 2))
    loss = F.mean((us - nu_) * (us - nu_))                          # il_exp.py:254-255
    g_logit, g_p = ch.grad([loss], [net.learn_q_logit, net.learn_p])
    np.savez_compressed(os.path.join(HERE, "imitation_16.npz"), B=B, T=T, lqr_iter=5, xinit=xinit, expert_x=arr(ex),
                        expert_u=arr(eu), q_logit=net.learn_q_logit.array, learn_p=net.learn_p.array,
                        nom_x=arr(nom_x), nom_u=arr(nom_u), loss=arr(loss), g_logit=arr(g_logit), g_p=arr(g_p))
    print("wrote imitation_16.npz loss %.6f g_logit %s g_p %s" % (float(arr(loss)), arr(g_logit), arr(g_p)))

    # ---- (b) B=1024, T=20: ONE MPCstep from a common iterate (forward + the no-op node's backward, the unit of work
    # of a config-4 iteration), learnable cost, true pendulum, update_dynamics=False (box_ddp.py:165-171,252-258)
    B = 1024
    dx = env.true_dx
    np.random.seed(0)
    xinit = _f32(env.sample_xinit(B))
    rng = np.random.RandomState(9)
    logit0, learn_p0 = np.array([0.5, -0.25, -1.0, -3.0]), np.array([-0.75, 0.125, 0.0625, 0.0])
    # the common iterate: three box-DDP iterations of the reference under the learner's cost (float32-rounded so that
    # the GPU path starts from identical numbers)
    q0 = 1.0 / (1.0 + np.exp(-logit0))
    env3 = ref.il_env.IL_Env('pendulum', lqr_iter=3, mpc_T=T)
    with warnings.catch_warnings(), redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        _, u3 = env3.mpc(dx, xinit, V(q0), V(np.sqrt(q0) * learn_p0), update_dynamics=True)
    u_k = _f32(arr(u3))
    expert_u = _f32(np.clip(u_k + 0.3 * rng.randn(T, B, 1), -2.0, 2.0))
    logit, learn_p = V(logit0), V(learn_p0)

    def cost_of(B):
        q = F.sigmoid(logit)
        p = F.sqrt(q) * learn_p
        Q = ref.util.chainer_diag(q)
        Q = F.repeat(F.repeat(F.expand_dims(F.expand_dims(Q, 0), 0), T, axis=0), B, axis=1)   # il_env.py:119-123
        pp = F.repeat(F.repeat(F.expand_dims(F.expand_dims(p, 0), 0), T, axis=0), B, axis=1)
        return Q, pp

    lo = np.full((T, B, 1), dx.lower)
    hi = np.full((T, B, 1), dx.upper)
    Q, pp = cost_of(B)
    x_k = ref.util.get_traj(T, V(u_k), x_init=V(xinit), dynamics=dx)
    Fk, fk = ref.approximate.linearize_dynamics(x_k, V(u_k), dx)
    step = ref.mpc_step.MPCstep(controls=V(u_k), T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=3, n_ctrl=1,
                                current_states=x_k, true_cost=ref.util.QuadCost(Q, pp), true_dynamics=dx,
                                ls_decay=dx.linesearch_decay, max_ls_iter=dx.max_linesearch_iter, need_expand=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x1, u1 = step.apply((arr(x_k)[0], Q, pp, arr(Fk), arr(fk)))
    fo = step.for_out
    # the gradient node (box_ddp.py:234-259): Taylor models at the new point, no-op forward, loss on the controls
    F1, f1 = ref.approximate.linearize_dynamics(x1, u1, dx)
    node = ref.mpc_step.MPCstep(controls=u1, T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=3, n_ctrl=1,
                                current_states=x1, true_cost=ref.util.QuadCost(Q, pp), true_dynamics=dx,
                                ls_decay=dx.linesearch_decay, max_ls_iter=dx.max_linesearch_iter, need_expand=True,
                                no_op_forward=True)
    x2, u2 = node.apply((arr(x1)[0], Q, pp, arr(F1), arr(f1)))
    loss = F.mean((expert_u - u2) * (expert_u - u2))
    g_logit, g_p, gQ, gp = ch.grad([loss], [logit, learn_p, Q, pp])
    S = np.arange(0, B, 8)                                               # 128 sampled trajectories for the big tensors
    np.savez_compressed(
        os.path.join(HERE, "imitation_step_1024.npz"), B=B, T=T, sample=S, logit=arr(logit), learn_p=arr(learn_p),
        u_k=u_k.astype(np.float32), expert_u=expert_u.astype(np.float32),
        x_k_s=arr(x_k)[:, S], F_k_s=arr(Fk)[:, S], f_k_s=arr(fk)[:, S],
        x1_s=arr(x1)[:, S], u1=arr(u1).astype(np.float32), costs=fo.costs, full_du_norm=fo.full_du_norm,
        mean_alphas=fo.mean_alphas, n_total_qp_iter=step.back_out.n_total_qp_iter,
        loss=arr(loss), g_logit=arr(g_logit), g_p=arr(g_p), dC_s=arr(gQ)[:, S], dc_s=arr(gp)[:, S])
    print("wrote imitation_step_1024.npz loss %.6f sat %.2f g_logit %s g_p %s mean_alpha %.3f" % (
        float(arr(loss)), float((np.abs(arr(u1)) == 2.0).mean()), arr(g_logit), arr(g_p), fo.mean_alphas))


def gen_imitation_early(ref):
    """config 4 at B=1024, T=20 once more, from an EARLIER common iterate: ONE box-DDP iteration of the reference under the
    learner's cost (imitation_step_1024.npz starts after three, where most rows are already near a fixed point and two
    thirds of them decide their line search by a margin float32 cannot resolve).  After one iteration the step is long and
    the cost drops by a resolvable margin on most rows: the pin of `MPCstep.forward` (mpc/mpc_step.py:288-328) on the
    rows that are NOT ties.  Forward only (the gradient node is pinned by imitation_step_1024.npz)."""
    ch = ref.chainer
    V, F = ch.Variable, ch.functions
    B, T = 1024, 20
    env1 = ref.il_env.IL_Env('pendulum', lqr_iter=1, mpc_T=T)
    dx = env1.true_dx
    np.random.seed(0)
    xinit = _f32(env1.sample_xinit(B))
    logit0, learn_p0 = np.array([0.5, -0.25, -1.0, -3.0]), np.array([-0.75, 0.125, 0.0625, 0.0])
    q0 = 1.0 / (1.0 + np.exp(-logit0))
    with warnings.catch_warnings(), redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        _, u1it = env1.mpc(dx, xinit, V(q0), V(np.sqrt(q0) * learn_p0), update_dynamics=True)
    u_k = _f32(arr(u1it))
    logit, learn_p = V(logit0), V(learn_p0)
    q = F.sigmoid(logit)
    p = F.sqrt(q) * learn_p
    Q = ref.util.chainer_diag(q)
    Q = F.repeat(F.repeat(F.expand_dims(F.expand_dims(Q, 0), 0), T, axis=0), B, axis=1)   # il_env.py:119-123
    pp = F.repeat(F.repeat(F.expand_dims(F.expand_dims(p, 0), 0), T, axis=0), B, axis=1)
    lo = np.full((T, B, 1), dx.lower)
    hi = np.full((T, B, 1), dx.upper)
    x_k = ref.util.get_traj(T, V(u_k), x_init=V(xinit), dynamics=dx)
    Fk, fk = ref.approximate.linearize_dynamics(x_k, V(u_k), dx)
    step = ref.mpc_step.MPCstep(controls=V(u_k), T=T, u_upper=hi, u_lower=lo, n_batch=B, n_state=3, n_ctrl=1,
                                current_states=x_k, true_cost=ref.util.QuadCost(Q, pp), true_dynamics=dx,
                                ls_decay=dx.linesearch_decay, max_ls_iter=dx.max_linesearch_iter, need_expand=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x1, u1 = step.apply((arr(x_k)[0], Q, pp, arr(Fk), arr(fk)))
    fo = step.for_out
    old = arr(ref.util.get_cost(T, V(u_k), ref.util.QuadCost(Q, pp), x=x_k)) if hasattr(ref.util, "get_cost") else None
    S = np.arange(0, B, 8)
    np.savez_compressed(
        os.path.join(HERE, "imitation_step_1024_it1.npz"), B=B, T=T, sample=S, logit=arr(logit), learn_p=arr(learn_p),
        u_k=u_k.astype(np.float32), x_k_s=arr(x_k)[:, S], x1=arr(x1).astype(np.float32), u1=arr(u1).astype(np.float32),
        costs=fo.costs, full_du_norm=fo.full_du_norm, mean_alphas=fo.mean_alphas,
        n_total_qp_iter=step.back_out.n_total_qp_iter)
    print("wrote imitation_step_1024_it1.npz sat %.2f mean_alpha %.3f mean cost %.4f" % (
        float((np.abs(arr(u1)) == 2.0).mean()), fo.mean_alphas, float(np.mean(fo.costs))))


def gen_imitation_loop(ref):
    """config 4's LOOP (env_dx/il_exp.py:213-302 and the evaluation pass :97-181), B=16, T=20, 10 iLQR iterations: K = 3
    consecutive updates with the reference's own pieces - Pendulum_Net_cost_logit.forward -> IL_Env.mpc, the loss of
    :254-255, chainer.grad - and what the loop does around them:
      * the training call passes `train_warm_start[idxs]` (:248; zeros - the buffer the loop fills is the differently
        spelled `train_warmstart`, :257), so every training solve starts cold;
      * `cost_update_q` starts False (:227,268-281): only learn_p moves during the first ten epochs;
      * opt = RMSprop(lr=1e-2, alpha=0.5) (:213): ms <- alpha ms + (1 - alpha) g^2, p <- p - lr g / (sqrt(ms) + 1e-8)
        (chainer.optimizers.RMSprop's rule with its default eps; the optimiser class itself is not part of the stand-in);
      * after each update an evaluation pass as dataset_loss (:97-181): the solve is warm-started from the controls the
        PREVIOUS pass predicted (`warmstart[idxs] = pred_u`, :122-124), no gradient.
    Stored per update k: loss, both gradients, learn_p after the update, the nominal controls; per evaluation pass: loss
    and the predicted controls (the next pass's warm start)."""
    import importlib
    ch = ref.chainer
    V, F = ch.Variable, ch.functions
    pnet = importlib.import_module("pendulum_net")
    B, T, K = 16, 20, 3
    env = ref.il_env.IL_Env('pendulum', lqr_iter=10, mpc_T=T)
    np.random.seed(11)
    xinit = _f32(env.sample_xinit(B))
    tq, tp = env.true_dx.get_true_obj()
    out = dict(B=B, T=T, K=K, lqr_iter=10, xinit=xinit, lr=1e-2, alpha=0.5, eps=1e-8)
    buf = io.StringIO()
    with warnings.catch_warnings(), redirect_stdout(buf):
        warnings.simplefilter("ignore")
        ex, eu = env.mpc(env.true_dx, xinit, tq, tp, update_dynamics=True)
        us = np.transpose(arr(eu), (1, 0, 2))                       # expert controls, [B,T,1] as the data set holds them
        net = pnet.Pendulum_Net_cost_logit(4)
        net.learn_q_logit.array[:] = np.array([0.5, -0.25, -1.0, -3.0])
        net.learn_p.array[:] = np.array([-0.75, 0.125, 0.0625, 0.0])
        out["q_logit0"], out["learn_p0"] = net.learn_q_logit.array.copy(), net.learn_p.array.copy()
        ms = np.zeros(4)
        train_warm_start = np.zeros((B, T, 1))                      # (:215; never written - see above)
        eval_warmstart = np.zeros((B, T, 1))                        # (:224)
        for k in range(K):
            nom_x, nom_u = net(xinit, env, train_warm_start)
            nu_ = F.transpose(nom_u, axes=(1, 0, 2))
            loss = F.mean((us - nu_) * (us - nu_))                  # :254-255
            g_logit, g_p = ch.grad([loss], [net.learn_q_logit, net.learn_p])
            g = arr(g_p)
            ms = 0.5 * ms + 0.5 * g * g
            net.learn_p.array[:] = net.learn_p.array - 1e-2 * g / (np.sqrt(ms) + 1e-8)
            out["loss_%d" % k], out["g_logit_%d" % k], out["g_p_%d" % k] = arr(loss), arr(g_logit), g
            out["nom_u_%d" % k], out["learn_p_%d" % k] = arr(nom_u), net.learn_p.array.copy()
            # evaluation pass (dataset_loss): warm start = the previous pass's prediction
            q = F.sigmoid(net.learn_q_logit)
            pp = F.sqrt(q) * net.learn_p
            _, pred_u = env.mpc(env.true_dx, xinit, q, pp, u_init=np.transpose(eval_warmstart, (1, 0, 2)))
            pred = np.transpose(arr(pred_u), (1, 0, 2))
            eval_warmstart[:] = pred
            out["eval_u_%d" % k] = arr(pred_u)
            out["eval_loss_%d" % k] = np.mean((us - pred) * (us - pred))
    out["expert_u"] = arr(eu)
    np.savez_compressed(os.path.join(HERE, "imitation_loop_16.npz"), **out)
    print("wrote imitation_loop_16.npz losses %s eval %s learn_p %s" % (
        [float(out["loss_%d" % k]) for k in range(K)], [float(out["eval_loss_%d" % k]) for k in range(K)],
        out["learn_p_%d" % (K - 1)]))


def gen_mpcnet_experiment(ref):
    """The first training iteration of the reference's own experiment experiment_mpc/MpcNet.py:24-120 at its own sizes -
    T=5, (3,3), B=128, bounds +-10, expert_seed 42, train_seed 1: expert BoxDDP under the true (A, B), the learner's
    MpcNet_dx, the imitation loss (:80-90) and d loss / d(A, B) - plus a tighter box (+-0.6: the clamped sets matter)."""
    import importlib
    ch = ref.chainer
    V, F, U = ch.Variable, ch.functions, ref.util
    mpc_net = importlib.import_module("mpc_net")          # (mpc/ is on the path once the reference is loaded)
    T, nx, nu, B, alpha = 5, 3, 3, 128, 0.2
    ns = nx + nu
    out = {}
    for tag, bound in (("wide", 10.0), ("tight", 0.6)):
        np.random.seed(42)                                               # expert_seed (:49-50)
        pvec = np.random.randn(ns)
        A_exp = np.eye(nx) + alpha * np.random.randn(nx, nx)
        B_exp = np.random.randn(nx, nu)
        F_exp = U.expand_time_batch(F.concat((V(A_exp), V(B_exp)), axis=1), T - 1, B)
        f_exp = V(np.zeros((T - 1, B, nx)))
        C = U.expand_time_batch(V(np.eye(ns)), T, B)
        c = U.expand_time_batch(V(pvec), T, B)
        lo = U.expand_time_batch(-bound * np.ones(nu), T, B)
        hi = U.expand_time_batch(bound * np.ones(nu), T, B)
        true_cost, true_dx = U.QuadCost(C, c), U.LinDx(F_exp, f_exp)
        buf = io.StringIO()
        with warnings.catch_warnings(), redirect_stdout(buf):
            warnings.simplefilter("ignore")
            net = mpc_net.MpcNet_dx(T, lo, hi, B, nx, nu, 1, u_init=None, max_iter=10, verbose=False)   # train_seed = 1
            x_init = V(_f32(np.random.randn(B, nx)))                      # (:108; float32-representable for the GPU path)
            expert = ref.box_ddp.BoxDDP(T, lo, hi, B, nx, nu, None)
            x_true, u_true, _ = expert.forward((x_init.array, true_cost, true_dx))
            x_pred, u_pred, _ = net((x_init, true_cost))
        x_true, u_true = arr(x_true), arr(u_true)
        loss = F.mean((u_true - u_pred) ** 2) + F.mean((x_true - x_pred) ** 2)   # (:86-89)
        gA, gB = ch.grad([loss], [net.A, net.B])
        sat = float(((np.abs(arr(u_pred)) >= bound - 1e-8)).mean())
        out.update({tag + "_bound": bound, tag + "_x_init": x_init.array, tag + "_A0": arr(net.A), tag + "_B0": arr(net.B),
                    tag + "_A_exp": A_exp, tag + "_B_exp": B_exp, tag + "_p": pvec,
                    tag + "_x_true": x_true, tag + "_u_true": u_true, tag + "_x_pred": arr(x_pred), tag + "_u_pred": arr(u_pred),
                    tag + "_loss": arr(loss), tag + "_gA": arr(gA), tag + "_gB": arr(gB), tag + "_stdout": buf.getvalue()})
        print("mpcnet experiment (%s): loss %.6f sat %.2f |gA| %.3e |gB| %.3e %s" % (
            tag, float(arr(loss)), sat, np.abs(arr(gA)).max(), np.abs(arr(gB)).max(), buf.getvalue().strip().replace("\n", " | ")[-160:]))
    np.savez_compressed(os.path.join(HERE, "mpcnet_experiment.npz"), T=T, nx=nx, nu=nu, B=B, **out)
    print("wrote mpcnet_experiment.npz")


def gen_pnqp_fork(ref):
    """PNQP n=8, B=256: batch-global convergence / Armijo tests make rows fork from their batch-of-one
    answers.  Batched run recorded in full; per-row answers for comparison."""
    n, B = 8, 256
    p = synthetic.make_box_qp(B, n, seed=1, bound=2.0, reg=0.1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x, (LU, piv), idx_f, it = ref.pnqp.PNQP(p["H"], p["q"], p["lower"], p["upper"], n_iter=20)
        warned = len(w) > 0
    xs, its = [], []
    for b in range(B):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xb, _, _, itb = ref.pnqp.PNQP(p["H"][b:b + 1], p["q"][b:b + 1], p["lower"][b:b + 1], p["upper"][b:b + 1],
                                          n_iter=20)
        xs.append(xb)
        its.append(itb)
    xr = np.concatenate(xs)
    np.savez_compressed(os.path.join(HERE, "pnqp_n8_b256.npz"), n=n, B=B, seed=1, bound=2.0, reg=0.1, in_checksum=checksum(p), x=x,
                        LU=LU, piv=piv, idx_f=idx_f, it=it, warned=warned, row_x=xr, row_it=np.array(its))
    d = np.abs(x - xr).max(axis=1)
    print("wrote pnqp_n8_b256.npz: it", it, "warned", warned, "rows differing > 1e-6:", int((d > 1e-6).sum()),
          "max", d.max())


def main():
    ref = load_reference.load()
    gen_lqr(ref)
    gen_lu(ref)
    extra = gen_pnqp(ref)
    gen_mpc(ref)
    gen_boxddp(ref)
    gen_anchors(ref, extra)
    gen_pendulum(ref)
    gen_approx_cost(ref)
    gen_pendulum_boxddp(ref)
    gen_pendulum_boxddp_b128(ref)
    gen_imitation(ref)
    gen_imitation_early(ref)
    gen_imitation_loop(ref)
    gen_pnqp_fork(ref)
    gen_mpcnet_experiment(ref)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        _ref = load_reference.load()
        for _name in sys.argv[1:]:
            globals()["gen_" + _name](_ref)
    else:
        main()
