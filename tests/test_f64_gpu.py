"""GPU parity of the float64 entry points (`dmpc_lqr_solve_f64`, `dmpc_lqr_kkt_grad_f64`; SURVEY.md 8b `_f64`): outputs at
the reference's own precision - the oracle (float64 numpy restatement of lqr/lqr_recursion.py:69-209 and
lqr/differentiable_lqr.py:78-142) to 1e-9 relative, and the golden vectors recorded from the reference."""
import glob
import os

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import DiffLqr, LqrRecursion, synthetic
from oracle import kkt as okkt
from oracle import lqr as olqr
from oracle import mpc as ompc
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
TOL64 = 1e-9     # relative to max(1, |reference|): float64 rounding through T Riccati steps, far below any float32 bound


def close64(got, want, what, tol=TOL64):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err <= tol, "%s: %.3e" % (what, err)


def dev64(a):
    return None if a is None else torch.as_tensor(np.asarray(a, dtype=np.float64)).cuda()


@pytest.mark.parametrize("shape", [(5, 6, 3, 1), (9, 7, 8, 2), (3, 5, 6, 3), (2, 4, 20, 6), (70, 5, 4, 2), (2, 3, 32, 8)])
@pytest.mark.parametrize("with_f", [True, False])
def test_solve_and_gradient_at_reference_precision(shape, with_f):
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=5, with_f=with_f)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    x, u = LqrRecursion(dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"]), T, nx, nu,
                        precision="float64").solve_recursion()
    assert x.dtype == torch.float64
    close64(x.cpu().numpy(), xr, "x")
    close64(u.cpu().numpy(), ur, "u")
    rng = np.random.RandomState(3)
    gx, gu = rng.randn(T, B, nx), rng.randn(T, B, nu)
    for strict in (False, True):
        ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
        node = DiffLqr(T, B, nx, nu, strict_math=strict, precision="float64")
        node.forward((dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"])))
        out = node.backward((0, 1, 2, 3, 4), (dev64(gx), dev64(gu)))
        for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
            close64(got.cpu().numpy(), want, key)


def test_clamped_controls_at_reference_precision():
    B, T, nx, nu = 6, 7, 5, 3
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=21, with_f=False)
    act = np.random.RandomState(5).rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(np.zeros((B, nx)), p["C"], p["c"], p["F"], None, act, T, nx, nu)
    x, u = LqrRecursion(dev64(np.zeros((B, nx))), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), None, T, nx, nu,
                        u_zero_Index=torch.as_tensor(act).cuda(), precision="float64").solve_recursion()
    # the clamped system carries 1e-8 on its diagonal (active_constrained_lqr.py:126): the reference multiplies by its
    # explicit inverse (F.batch_inv), which rounds at cond * eps ~ 1e-8; the kernel solves by LU - they agree to that
    close64(x.cpu().numpy(), xr, "x", 1e-6)
    close64(u.cpu().numpy(), ur, "u", 1e-6)
    assert np.all(u.cpu().numpy()[act] == 0)


ROW_SHAPES = [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2), (4, 2), (6, 2), (8, 2), (4, 4), (8, 4), (12, 3), (16, 4), (16, 8), (32, 8)]


@pytest.mark.parametrize("dims", ROW_SHAPES, ids=lambda d: "%dx%d" % d)
def test_register_resident_float64_kernels_against_the_oracle(dims):
    """f64_row_kernels.hpp (round 4: the float64 FAST path - a trajectory per 16 lanes with `v_fmac_f64_dpp` blocks, or per
    wavefront): every instantiated shape, a ragged batch, with and without f, plain and clamped (LQR_active,
    mpc/active_constrained_lqr.py:110-145), gains out, and a horizon beyond what the LDS gain rows hold (gains through the
    workspace) - solution and gains to 1e-9 of the float64 oracle, the gradient's five outputs too (lqr/lqr_recursion.py:69-209,
    lqr/differentiable_lqr.py:78-142)."""
    from chainer_differentiable_mpc_amd import _lib
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device_f64
    nx, nu = dims
    assert _lib.load().dmpc_lqr_f64_path(nx, nu) == (1 if nx + nu + 1 <= 16 else 2)
    assert _lib.load().dmpc_lqr_f64_path(6, 3) == 0 and _lib.load().dmpc_lqr_f64_path(20, 6) == 0     # the one-lane family
    long_T = 1 + (160 * 1024) // ((256 // (16 if nx + nu + 1 <= 16 else 64)) * nu * (nx + 1) * 8)
    for (B, T, with_f, masked) in ((7, 9, True, False), (21, 6, False, True), (5, min(long_T, 400), True, False), (3, 1, True, False),
                                   (6, 2, False, False)):
        if nx >= 16 and T > 60:
            T = 60 if (256 // 64) * 60 * nu * (nx + 1) * 8 > 160 * 1024 else T      # (keep the wide shapes' share small)
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=7 * nx + nu, with_f=with_f)
        mask = np.random.RandomState(nx + T).rand(T, B, nu) < 0.4 if masked else None
        if masked:
            xr, ur = ompc.lqr_active_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], mask, T, nx, nu)
            Ksr = ksr = None
        else:
            Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
            xr, ur = olqr.lqr_forward(Ksr, ksr, p["x_init"], p["F"], p["f"], T, nx, nu)
        for want_gains in (True, False):
            x, u, Ks, ks = solve_device_f64(dev64(p["C"]), dev64(p["c"]), dev64(p["F"]) if T > 1 else None,
                                            dev64(p["f"]) if T > 1 else None, dev64(p["x_init"]),
                                            None if mask is None else torch.as_tensor(mask).cuda().to(torch.uint8).contiguous(),
                                            T, nx, nu, want_gains=want_gains)
            tol = 1e-6 if masked else TOL64         # (the 1e-8 regulariser against the reference's explicit inverse, see above)
            close64(x.cpu().numpy(), xr, "x B=%d T=%d" % (B, T), tol)
            close64(u.cpu().numpy(), ur, "u B=%d T=%d" % (B, T), tol)
            if want_gains and Ksr is not None:
                close64(Ks.cpu().numpy(), np.stack(Ksr), "Ks")
                close64(ks.cpu().numpy(), np.stack(ksr), "ks")
            if masked:
                assert np.all(u.cpu().numpy()[mask] == 0)
            elif dims in ((16, 8), (32, 8)):      # round 5: the plain solve of these runs on v_mfma_f64_16x16x4_f64 tiles
                assert "lqr_tile16_f64_kernel" in _lib.last_kernel_name(), _lib.last_kernel_name()
    B, T = 6, 8
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=nx)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    rng = np.random.RandomState(3)
    gx, gu = rng.randn(T, B, nx), rng.randn(T, B, nu)
    for strict in (False, True):
        ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
        node = DiffLqr(T, B, nx, nu, strict_math=strict, precision="float64")
        node.forward((dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"])))
        out = node.backward((0, 1, 2, 3, 4), (dev64(gx), dev64(gu)))
        for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
            close64(got.cpu().numpy(), want, key)


LQR_FILES = sorted(glob.glob(os.path.join(GOLDEN, "lqr_*.npz")))


@pytest.mark.parametrize("path", LQR_FILES, ids=[os.path.basename(p) for p in LQR_FILES])
def test_reference_golden_vectors_at_reference_precision(path):
    """the vectors recorded from the unmodified reference (float64): solution and gradient to 1e-9"""
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=bool(g["with_f"]))   # the recorded run's inputs
    node = DiffLqr(T, B, nx, nu, precision="float64")
    x, u = node.forward((dev64(p["x_init"]), dev64(p["C"]), dev64(p["c"]), dev64(p["F"]), dev64(p["f"])))
    close64(x.cpu().numpy(), g["x"], "x")
    close64(u.cpu().numpy(), g["u"], "u")
    out = node.backward((0, 1, 2, 3, 4), (dev64(g["grad_x"]), dev64(g["grad_u"])))
    for got, key in zip(out, ("d_x_init", "dC", "dc", "dF", "df")):
        close64(got.cpu().numpy(), g[key], key)


def test_headline_batch_every_trajectory_against_the_float64_kernels():
    """BASELINE.json configs[2] at FULL size (B=4096, T=50, (8,2)): the float32 fast path against the float64 kernels
    (themselves held to the oracle above) on EVERY trajectory - solution at the north star's 1e-4, gradient at 5e-4 - not
    on a sample as the oracle tests have to"""
    from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device, kkt_grad_device_f64
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device, solve_device_f64
    from tests.helpers import TOL_COSTATE, TOL_PRIMAL, assert_close
    B, T, nx, nu = 4096, 50, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d32 = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in p.items() if isinstance(v, np.ndarray)}
    d64 = {k: v.double() for k, v in d32.items()}      # identical inputs: the float32 values up-cast
    x, u, _, _ = solve_device(d32["C"], d32["c"], d32["F"], d32["f"], d32["x_init"], None, T, nx, nu)
    x64, u64, _, _ = solve_device_f64(d64["C"], d64["c"], d64["F"], d64["f"], d64["x_init"], None, T, nx, nu)
    assert_close(x.cpu().numpy(), x64.cpu().numpy(), TOL_PRIMAL, "x")
    assert_close(u.cpu().numpy(), u64.cpu().numpy(), TOL_PRIMAL, "u")
    gx, gu = torch.ones((T, B, nx), device="cuda"), torch.ones((T, B, nu), device="cuda")
    out = kkt_grad_device(d32["C"], d32["c"], d32["F"], x, u, gx, gu, T, nx, nu)
    ref = kkt_grad_device_f64(d64["C"], d64["c"], d64["F"], x64, u64, gx.double(), gu.double(), T, nx, nu)
    for got, want, key in zip(out, ref, ("d_x_init", "dC", "dc", "dF", "df")):
        assert_close(got.cpu().numpy(), want.cpu().numpy(), TOL_COSTATE if key in ("d_x_init", "dF", "df") else TOL_PRIMAL, key)


def test_config5_shard_every_trajectory_against_the_float64_kernels():
    """BASELINE.json configs[4], one GPU's shard at FULL size (B=8192, T=50, (32,8)): the float32 matrix-core path - the
    fused launch the benchmark times AND the gains-out form - against the float64 kernels on EVERY trajectory at the
    contract's 1e-4 (x, u, Ks, ks), as test_headline_batch_every_trajectory... does for (8,2).  Measured worst case over the
    8,192: x 2.9e-6, u 1.5e-6, Ks 5.1e-7, ks 5.4e-7 (profiles/r04/parity_margins.txt) - round 3 held 512 of them to 5e-4.
    Inputs are drawn on the device (the normalised generator's distributions) in chunks of 2,048 trajectories."""
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device, solve_device_f64
    from tests.helpers import TOL_PRIMAL, assert_close
    B, T, nx, nu, chunk = 8192, 50, 32, 8, 2048
    ns = nx + nu
    worst = dict(x=0.0, u=0.0, Ks=0.0, ks=0.0)
    for b0 in range(0, B, chunk):
        g = torch.Generator(device="cuda")
        g.manual_seed(50 + b0)
        L = torch.randn((T, chunk, ns, ns), generator=g, device="cuda")
        C = (L @ L.transpose(2, 3) + ns * torch.eye(ns, device="cuda")) / ns
        del L
        c = torch.randn((T, chunk, ns), generator=g, device="cuda")
        F = torch.cat((torch.eye(nx, device="cuda") + (0.2 / nx ** 0.5) * torch.randn((T - 1, chunk, nx, nx), generator=g, device="cuda"),
                       torch.randn((T - 1, chunk, nx, nu), generator=g, device="cuda")), dim=3).contiguous()
        f = 0.1 * torch.randn((T - 1, chunk, nx), generator=g, device="cuda")
        x0 = torch.randn((chunk, nx), generator=g, device="cuda")
        x, u, Ks, ks = solve_device(C, c, F, f, x0, None, T, nx, nu, want_gains=True)
        xf, uf, _, _ = solve_device(C, c, F, f, x0, None, T, nx, nu)            # the fused launch bench.py times
        x64, u64, Ks64, ks64 = solve_device_f64(C.double(), c.double(), F.double(), f.double(), x0.double(), None, T, nx, nu,
                                                want_gains=True)
        for key, got, want in (("x", x, x64), ("u", u, u64), ("Ks", Ks, Ks64), ("ks", ks, ks64), ("x", xf, x64), ("u", uf, u64)):
            err = float(((got.double() - want).abs() / want.abs().clamp(min=1.0)).max())
            assert bool(torch.isfinite(got).all())
            worst[key] = max(worst[key], err)
        del C, c, F, f, x, u, Ks, ks, xf, uf, x64, u64, Ks64, ks64
    from tests.helpers import log_margin
    for key, err in worst.items():
        log_margin(key + " (all 8192 trajectories)", err, TOL_PRIMAL)
        assert err <= TOL_PRIMAL, "%s: %.3e" % (key, err)


def test_wide_mpc_sweeps_against_the_oracle_at_the_contract():
    """`MPCstep.backward_rec` + `forward_rec` at (16,8) and (32,8) (the matrix-core sweep with the box QP inside,
    mpc/mpc_step.py:70-286) against the oracle at the contract's 1e-4 - gains, controls, states, costs"""
    import warnings
    from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost
    from oracle import box_ddp as obox
    from oracle import mpc as ompc
    from tests.helpers import TOL_STEP, assert_close
    # (32,4), (24,2), (21,3): padded inside the (32,8) instance since round 5 (before: the runtime-dimension kernel, 2-3x slower)
    for (B, T, nx, nu) in ((24, 12, 16, 8), (12, 10, 32, 8), (8, 8, 32, 4), (8, 8, 24, 2), (8, 8, 21, 3)):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=nx + 3)
        rng = np.random.RandomState(nx)
        u_nom = np.clip(0.3 * rng.randn(T, B, nu), -0.15, 0.15).astype(np.float32).astype(np.float64)
        lo, hi = np.full((T, B, nu), -0.15), np.full((T, B, nu), 0.15)
        cost_o, dyn_o = ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"])
        x_nom = obox.get_traj(T, u_nom, p["x_init"], dyn_o).astype(np.float32).astype(np.float64)
        xr, ur, _, fo, Ksr, ksr = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u_nom, x_nom, lo, hi, cost_o, dyn_o, 0.2, 5,
                                                   T, nx, nu, need_expand=True, batch_coupled=False)
        d = lambda a: torch.as_tensor(a, dtype=torch.float32).cuda()   # noqa: E731
        step = MPCstep(d(u_nom), T, d(hi), d(lo), B, nx, nu, d(x_nom), QuadCost(d(p["C"]), d(p["c"])), LinDx(d(p["F"]), d(p["f"])),
                       0.2, 5, need_expand=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, u = step.forward((d(x_nom[0]), d(p["C"]), d(p["c"]), d(p["F"]), d(p["f"])))
        what = " (%d,%d)" % (nx, nu)
        assert_close(step.Ks.cpu().numpy(), Ksr, TOL_STEP, "Ks" + what)
        assert_close(step.ks.cpu().numpy(), ksr, TOL_STEP, "ks" + what)
        assert_close(u.cpu().numpy(), ur, TOL_STEP, "u" + what)
        assert_close(x.cpu().numpy(), xr, TOL_STEP, "x" + what)
        assert_close(step.for_out.costs.cpu().numpy(), fo.costs, TOL_STEP, "costs" + what)
        un = u.cpu().numpy()
        on = (un == np.float32(-0.15)) | (un == np.float32(0.15))                    # the box is active somewhere, and where the
        assert on.any() and np.array_equal(on, (np.abs(ur - lo) <= 1e-8) | (np.abs(ur - hi) <= 1e-8))     # reference has it
        if nx >= 21:
            from chainer_differentiable_mpc_amd import _lib
            tau = np.concatenate((x_nom, u_nom), axis=2)
            c_hat = np.einsum("tbij,tbj->tbi", p["C"], tau) + p["c"]
            Ks2, ks2, _ = step.backward_rec(d(p["C"]), d(c_hat), d(p["F"]), None)
            assert "lqr_wave_mfma_backward<32, 8" in _lib.last_kernel_name(), _lib.last_kernel_name()
            assert_close(Ks2.cpu().numpy(), Ksr, TOL_STEP, "Ks (backward_rec)" + what)


def _fuzz_cases():
    rng = np.random.RandomState(2026)
    cases = []
    for _ in range(40):
        nx = int(rng.randint(1, 33))
        nu = int(rng.randint(1, min(8, 40 - nx) + 1))
        T = int(rng.choice([1, 2, 3, 7, 20, 51, 52, 75, 90]))
        B = int(rng.choice([1, 3, 4, 17, 64, 130]))
        if nx * T > 900:          # (keep the float64 path's share of the suite small)
            T = max(1, 900 // nx)
        cases.append((B, T, nx, nu, bool(rng.randint(2)), bool(rng.randint(2))))
    # ... and twenty around the wide row kernel's borders (lqr_wide_kernel.hpp: 8..16 states, nx + nu on either side of 16,
    # whole and ragged wavefronts - the padded form wants B % 4 == 0 and hands anything else to the wavefront containers)
    rng = np.random.RandomState(4)
    for _ in range(20):
        nx = int(rng.randint(8, 17))
        nu = int(rng.randint(max(1, 14 - nx), 9))
        T = int(rng.choice([2, 3, 9, 26, 51]))
        B = int(rng.choice([4, 8, 12, 36, 37, 64]))
        cases.append((B, T, nx, nu, bool(rng.randint(2)), bool(rng.randint(2))))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "B%d_T%d_%dx%d_%s%s" % (c[0], c[1], c[2], c[3], "f" if c[4] else "nof", "_masked" if c[5] else ""))
def test_float32_paths_against_the_float64_kernels_over_the_shape_space(case):
    """forty random (B, T, nx, nu) up to 32 states / 8 controls - specialised shapes, containers, wide containers, ragged
    batches, horizons on either side of the stash / ring / workspace limits, with and without f, plain and clamped: whichever
    float32 kernel the dispatch picks against the float64 kernel on identical inputs"""
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device, solve_device_f64
    from tests.helpers import TOL_PRIMAL, assert_close
    B, T, nx, nu, with_f, masked = case
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=B * 1000 + T * 10 + nx, with_f=with_f)
    d32 = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in p.items() if isinstance(v, np.ndarray)}
    d64 = {k: v.double() for k, v in d32.items()}
    mask = None
    if masked:
        mask = torch.as_tensor(np.random.RandomState(B + T).rand(T, B, nu) < 0.35).cuda().to(torch.uint8).contiguous()
    f32, f64 = d32.get("f") if with_f else None, d64.get("f") if with_f else None
    x, u, Ks, ks = solve_device(d32["C"], d32["c"], d32["F"] if T > 1 else None, f32, d32["x_init"], mask, T, nx, nu, want_gains=True)
    x64, u64, Ks64, ks64 = solve_device_f64(d64["C"], d64["c"], d64["F"] if T > 1 else None, f64, d64["x_init"], mask, T, nx, nu,
                                            want_gains=True)
    tol = TOL_PRIMAL    # 1e-4 for every shape, plain and clamped (round 3: 3e-4 / 2e-4; measured worst: 2.4e-6, profiles/r04/parity_margins.txt)
    assert_close(x.cpu().numpy(), x64.cpu().numpy(), tol, "x")
    assert_close(u.cpu().numpy(), u64.cpu().numpy(), tol, "u")
    assert_close(Ks.cpu().numpy(), Ks64.cpu().numpy(), tol, "Ks")
    assert_close(ks.cpu().numpy(), ks64.cpu().numpy(), tol, "ks")
    if masked:
        assert bool((u[mask.bool()] == 0).all())


@pytest.mark.parametrize("case", [(3, 4, 65, 1, True, False), (2, 5, 48, 20, False, False), (3, 3, 90, 7, True, True), (2, 6, 33, 31, True, False),
                                  (4, 3, 120, 2, False, True)],
                         ids=lambda c: "B%d_T%d_%dx%d_%s%s" % (c[0], c[1], c[2], c[3], "f" if c[4] else "nof", "_masked" if c[5] else ""))
def test_any_size_float32_kernels_against_the_float64_kernels(case):
    """beyond a wavefront's 64 columns (kernel family 5, lqr_tiled.hpp: a workgroup per trajectory) against the float64
    one-lane kernels (runtime dimensions, any size) on identical inputs - solution and gains at the contract's 1e-4"""
    from chainer_differentiable_mpc_amd import _lib
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device, solve_device_f64
    from tests.helpers import TOL_PRIMAL, assert_close
    B, T, nx, nu, with_f, masked = case
    assert _lib.load().dmpc_lqr_kernel_family(nx, nu) == 5
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=B * 1000 + T * 10 + nx, with_f=with_f)
    d32 = {k: torch.as_tensor(v, dtype=torch.float32).cuda() for k, v in p.items() if isinstance(v, np.ndarray)}
    d64 = {k: v.double() for k, v in d32.items()}
    mask = None
    if masked:
        mask = torch.as_tensor(np.random.RandomState(B + T).rand(T, B, nu) < 0.35).cuda().to(torch.uint8).contiguous()
    x, u, Ks, ks = solve_device(d32["C"], d32["c"], d32["F"], d32.get("f"), d32["x_init"], mask, T, nx, nu, want_gains=True)
    x64, u64, Ks64, ks64 = solve_device_f64(d64["C"], d64["c"], d64["F"], d64.get("f"), d64["x_init"], mask, T, nx, nu, want_gains=True)
    for got, want, key in ((x, x64, "x"), (u, u64, "u"), (Ks, Ks64, "Ks"), (ks, ks64, "ks")):
        assert_close(got.cpu().numpy(), want.cpu().numpy(), TOL_PRIMAL, key)
    if masked:
        assert bool((u[mask.bool()] == 0).all())


def _grad_fuzz_cases():
    rng = np.random.RandomState(77)
    cases = []
    for _ in range(24):
        nx = int(rng.randint(1, 25))
        nu = int(rng.randint(1, min(8, 32 - nx) + 1))
        T = int(rng.choice([2, 3, 7, 20, 50, 60]))
        B = int(rng.choice([1, 4, 5, 16, 68]))
        if nx * T > 600:
            T = max(2, 600 // nx)
        cases.append((B, T, nx, nu, bool(rng.randint(2))))
    rng = np.random.RandomState(5)      # ... and twelve around the wide kernels' borders (as in _fuzz_cases)
    for _ in range(12):
        nx = int(rng.randint(8, 17))
        nu = int(rng.randint(max(1, 14 - nx), 9))
        cases.append((int(rng.choice([4, 8, 36, 37])), int(rng.choice([2, 3, 9, 26])), nx, nu, bool(rng.randint(2))))
    return cases


@pytest.mark.parametrize("case", _grad_fuzz_cases(), ids=lambda c: "B%d_T%d_%dx%d_%s" % (c[0], c[1], c[2], c[3], "strict" if c[4] else "faithful"))
def test_float32_gradient_against_the_float64_kernels_over_the_shape_space(case):
    """DiffLqr forward + backward (whichever kernels the dispatch picks: saved gains / one launch, re-solve + co-state sweep,
    containers) against the float64 kernels on identical inputs"""
    from tests.helpers import TOL_COSTATE, TOL_PRIMAL, assert_close
    B, T, nx, nu, strict = case
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=B * 100 + T + nx)
    rng = np.random.RandomState(B + nx)
    gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
    gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
    outs = {}
    for prec, dt in (("float32", torch.float32), ("float64", torch.float64)):
        args = tuple(None if p[k] is None else torch.as_tensor(p[k], dtype=torch.float32).to(dt).cuda() for k in ("x_init", "C", "c", "F", "f"))
        node = DiffLqr(T, B, nx, nu, strict_math=strict, precision=prec)
        node.forward(args)
        outs[prec] = node.backward((0, 1, 2, 3, 4), (torch.as_tensor(gx, dtype=dt).cuda(), torch.as_tensor(gu, dtype=dt).cuda()))
    for got, want, key in zip(outs["float32"], outs["float64"], ("d_x_init", "dC", "dc", "dF", "df")):
        # the contract for every shape (round 3: 1e-3 beyond 12 states; measured worst: dF 3.2e-5, profiles/r04/parity_margins.txt)
        assert_close(got.cpu().numpy(), want.cpu().numpy(), TOL_COSTATE if key in ("d_x_init", "dF", "df") else TOL_PRIMAL, key)
