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


def main():
    ref = load_reference.load()
    gen_lqr(ref)
    gen_lu(ref)
    extra = gen_pnqp(ref)
    gen_mpc(ref)
    gen_boxddp(ref)
    gen_anchors(ref, extra)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mpc":
        gen_mpc(load_reference.load())
    else:
        main()
