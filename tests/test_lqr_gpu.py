"""GPU parity: the fused HIP LQR solve (through the C-ABI) against the numpy oracle and the
golden vectors recorded from the reference.  Row A and F1 of SURVEY.md section 8."""
import glob
import os

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import LqrRecursion, _lib, synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import lqr as olqr
from oracle import mpc as ompc
from tests.helpers import GOLDEN, TOL_PRIMAL, assert_close, npy, to_dev

pytestmark = pytest.mark.gpu

LQR_FILES = sorted(glob.glob(os.path.join(GOLDEN, "lqr_*.npz")))


@pytest.mark.parametrize("path", LQR_FILES, ids=[os.path.basename(p) for p in LQR_FILES])
def test_solve_matches_reference_golden(path):
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=bool(g["with_f"]))
    d = to_dev(p)
    rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
    x, u = rec.solve_recursion()
    assert x.is_cuda and x.dtype == torch.float32 and list(x.shape) == [T, B, nx]
    assert_close(npy(x), g["x"], TOL_PRIMAL, "x")
    assert_close(npy(u), g["u"], TOL_PRIMAL, "u")
    assert int(rec.info.max().item()) == 0
    Ks, ks = rec.backward()
    assert len(Ks) == T and list(Ks[0].shape) == [B, nu, nx] and list(ks[0].shape) == [B, nu]
    assert_close(npy(torch.stack(Ks)), g["Ks"], TOL_PRIMAL, "Ks")
    assert_close(npy(torch.stack(ks)), g["ks"], TOL_PRIMAL, "ks")
    x2, u2 = rec.forward(Ks, ks)
    assert_close(npy(x2), g["x"], TOL_PRIMAL, "x (forward)")
    assert_close(npy(u2), g["u"], TOL_PRIMAL, "u (forward)")


def test_anchor_one_variable_notebook():
    """examples/LQR_recursion_solver_one_variable.ipynb:226-245,346-382 on the GPU."""
    T, nx, nu = 20, 2, 1
    F = np.tile(np.array([[1.0, 1.0, 0], [0, 1.0, 1.0]]), (T, 1, 1, 1))
    c = np.zeros((T, 1, 3))
    C = np.tile(np.array([[1.0, 0, 0], [0, 0, 0], [0, 0, 10]]), (T, 1, 1, 1))
    x0 = np.array([[1.0, 0.0]])
    rec = LqrRecursion(torch.tensor(x0).cuda(), torch.tensor(C).cuda(), torch.tensor(c).cuda(),
                       torch.tensor(F).cuda(), None, T, nx, nu)
    Ks, ks = rec.backward()
    x, u = rec.solve_recursion()
    assert x.dtype == torch.float64            # outputs follow the dtype of C
    np.testing.assert_allclose(npy(Ks[0])[0, 0], [-0.21140641, -0.7644787], atol=2e-6)
    np.testing.assert_allclose(npy(Ks[17])[0, 0], [-0.09090909, -0.18181818], atol=2e-6)
    assert np.all(npy(torch.stack(Ks[18:])) == 0)
    np.testing.assert_allclose(npy(x)[19, 0], [1.00328598e-03, -3.91198810e-04], atol=2e-6)
    a = np.load(os.path.join(GOLDEN, "anchors.npz"))
    assert_close(npy(x), a["onevar_x"], TOL_PRIMAL, "x")
    assert_close(npy(u), a["onevar_u"], TOL_PRIMAL, "u")


def test_anchor_boyd_notebook():
    """examples/Boyd_lqr.ipynb:508-558: steady-state gain; Quu = 1e-14 at the last step."""
    T, nx, nu = 51, 3, 1
    F = np.tile(np.array([[1.0, 0, 0, 1], [1, 1.0, 0, 0], [0, 1, 1, 0]]), (T, 1, 1, 1))
    c = np.zeros((T, 1, 4))
    C = np.tile(np.diag([0, 0, 1.0, 1.0]), (T, 1, 1, 1))
    C[T - 1, 0, 3, 3] = 0.00000000000001
    x0 = np.array([[0.5428, 0.7633, 0.3504]])
    rec = LqrRecursion(torch.tensor(x0).cuda(), torch.tensor(C).cuda(), torch.tensor(c).cuda(),
                       torch.tensor(F).cuda(), None, T, nx, nu)
    Ks, ks = rec.backward()
    np.testing.assert_allclose(npy(Ks[0])[0, 0], [-1.86152282, -1.34921019, -0.35888729], atol=2e-5)
    assert np.all(npy(torch.stack(Ks[-3:])) == 0)
    x, u = rec.solve_recursion()
    a = np.load(os.path.join(GOLDEN, "anchors.npz"))
    assert_close(npy(x), a["boyd_x"], TOL_PRIMAL, "x")
    assert_close(npy(u), a["boyd_u"], TOL_PRIMAL, "u")


@pytest.mark.parametrize("shape", [(7, 9, 5, 3), (3, 6, 7, 1), (2, 5, 12, 3), (5, 4, 1, 1), (2, 4, 20, 6), (2, 4, 40, 4), (3, 5, 10, 9),
                                   (2, 3, 50, 12)])
@pytest.mark.parametrize("with_f", [True, False])
def test_other_shapes_against_oracle(shape, with_f):
    """shapes that dispatch to other specialisations, to a container, or - beyond 32 states / 8 controls - to the
    runtime-dimension kernel"""
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=3, with_f=with_f)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
    x, u = rec.solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    Ks, ks = rec.backward()
    assert_close(npy(torch.stack(Ks)), Ksr, TOL_PRIMAL, "Ks")
    assert_close(npy(torch.stack(ks)), ksr, TOL_PRIMAL, "ks")
    x2, u2 = rec.forward(Ks, ks)
    assert_close(npy(x2), xr, TOL_PRIMAL, "x fwd")


CONTAINER_SHAPES = [(5, 1), (13, 1), (14, 1), (1, 2), (5, 2), (7, 2), (9, 2), (13, 2), (2, 3), (6, 3), (10, 3), (1, 4), (3, 4),
                    (6, 4), (9, 4), (11, 4)]


@pytest.mark.parametrize("dims", CONTAINER_SHAPES, ids=["%dx%d" % d for d in CONTAINER_SHAPES])
def test_container_shapes_against_oracle(dims):
    """shapes without a specialisation of their own run padded inside a larger register-resident kernel (solve path 7):
    solve, gains, rollout from given gains and the clamped variant against the oracle, ragged batch"""
    nx, nu = dims
    B, T = 19, 7
    lib = _lib.load()
    assert lib.dmpc_lqr_kernel_family(nx, nu) == 4 and lib.dmpc_lqr_solve_path(T, B, nx, nu) == 7
    for with_f in (True, False):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=100 + nx, with_f=with_f)
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        d = to_dev(p)
        rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
        x, u = rec.solve_recursion()
        assert_close(npy(x), xr, TOL_PRIMAL, "x")
        assert_close(npy(u), ur, TOL_PRIMAL, "u")
        Ks, ks = rec.backward()
        if nx > 16:   # the sweep runs in the smallest wavefront-per-trajectory instance that holds the problem
            inst = next(c for c in ((24, 4), (24, 8), (32, 4), (32, 8)) if nx <= c[0] and nu <= c[1])
            if dims == inst:     # a size that IS an instance runs the exact kernel: the 16x16x4 tile sweep (lqr_tile16.hpp)
                assert _lib.last_kernel_name().startswith("void dmpc::lqr_tile16_kernel<%d, %d, false>" % inst)
            else:
                assert _lib.last_kernel_name().startswith("void dmpc::lqr_wave_mfma_backward<%d, %d, false, false, true" % inst)
        assert_close(npy(torch.stack(Ks)), Ksr, TOL_PRIMAL, "Ks")
        assert_close(npy(torch.stack(ks)), ksr, TOL_PRIMAL, "ks")
        x2, u2 = rec.forward(Ks, ks)
        assert_close(npy(x2), xr, TOL_PRIMAL, "x fwd")
        assert_close(npy(u2), ur, TOL_PRIMAL, "u fwd")
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=7, with_f=False)
    act = np.random.RandomState(nx * 16 + nu).rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(np.zeros((B, nx)), p["C"], p["c"], p["F"], None, act, T, nx, nu)
    d = to_dev(p)
    x, u = LqrRecursion(torch.zeros_like(d["x_init"]), d["C"], d["c"], d["F"], None, T, nx, nu,
                        u_zero_Index=torch.as_tensor(act).cuda()).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x active")
    assert_close(npy(u), ur, TOL_PRIMAL, "u active")
    assert np.all(npy(u)[act] == 0)


WIDE_ROW_SHAPES = [(16, 4), (16, 8), (12, 4), (12, 8)]     # lqr_wide_kernel.hpp: two registers per matrix row
WAVE_CONTAINER_SHAPES = [(5, 5), (3, 8), (12, 4), (16, 4), (10, 6), (16, 8), (15, 7), (20, 6), (24, 8), (31, 7), (17, 1), (32, 3),
                         (24, 4), (21, 5), (25, 4), (32, 7)]


@pytest.mark.parametrize("dims", WAVE_CONTAINER_SHAPES, ids=["%dx%d" % d for d in WAVE_CONTAINER_SHAPES])
def test_wide_container_shapes_against_oracle(dims):
    """up to 32 states and 8 controls: padded inside the wavefront-per-trajectory kernels ((16,8) or (32,8) instance; sweep
    on the matrix cores, then the forward-only container kernel) - solve, gains, rollout, the clamped variant"""
    nx, nu = dims
    B, T = 7, 6
    lib = _lib.load()
    assert lib.dmpc_lqr_kernel_family(nx, nu) == 4
    assert lib.dmpc_lqr_solve_path(T, B, nx, nu) == (9 if dims in WIDE_ROW_SHAPES else 7)   # (9: the fused solve only)
    for with_f in (True, False):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=200 + nx, with_f=with_f)
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        d = to_dev(p)
        rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
        x, u = rec.solve_recursion()
        assert_close(npy(x), xr, TOL_PRIMAL, "x")
        assert_close(npy(u), ur, TOL_PRIMAL, "u")
        Ks, ks = rec.backward()
        assert_close(npy(torch.stack(Ks)), Ksr, TOL_PRIMAL, "Ks")
        assert_close(npy(torch.stack(ks)), ksr, TOL_PRIMAL, "ks")
        x2, u2 = rec.forward(Ks, ks)
        assert_close(npy(x2), xr, TOL_PRIMAL, "x fwd")
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=7, with_f=False)
    act = np.random.RandomState(nx * 16 + nu).rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(np.zeros((B, nx)), p["C"], p["c"], p["F"], None, act, T, nx, nu)
    d = to_dev(p)
    x, u = LqrRecursion(torch.zeros_like(d["x_init"]), d["C"], d["c"], d["F"], None, T, nx, nu,
                        u_zero_Index=torch.as_tensor(act).cuda()).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x active")
    assert_close(npy(u), ur, TOL_PRIMAL, "u active")
    assert np.all(npy(u)[act] == 0)


# ... and what runs padded inside them (at most 16 states, 8 controls, nx + nu >= 16: no 16-lane container)
WIDE_ROW_PADDED = [(13, 3), (15, 1), (14, 2), (10, 6), (8, 8), (9, 8), (15, 7), (16, 7), (16, 1), (12, 5), (11, 5)]


@pytest.mark.parametrize("dims", WIDE_ROW_SHAPES + WIDE_ROW_PADDED, ids=["%dx%d" % d for d in WIDE_ROW_SHAPES + WIDE_ROW_PADDED])
def test_wide_row_kernel_against_oracle(dims):
    """17 to 32 augmented columns, at most 16 states: the fused solve on the wide 16-lane row layout (lqr_wide_kernel.hpp; four
    trajectories per wavefront, gain rows through the workspace) - whole and ragged batches, two steps to a horizon that
    wraps its rings many times, with and without f, with and without the gains handed out; and the same solve on the
    path these shapes took before (DMPC_NO_WIDE is read once per process, so: the separate sweeps, which never take it)."""
    nx, nu = dims
    lib = _lib.load()
    padded = dims in WIDE_ROW_PADDED          # (the padded form wants whole wavefronts: B % 4 == 0)
    inst = dims if not padded else next(c for c in ((12, 4), (16, 4), (12, 8), (16, 8)) if nx <= c[0] and nu <= c[1])
    for (B, T, with_f, want_gains, seed) in ((64, 13, True, False, 1), (36 if padded else 37, 7, False, True, 2),
                                            (4, 2, True, True, 3), (8 if padded else 5, 41, True, False, 4)):
        assert lib.dmpc_lqr_solve_path(T, B, nx, nu) == 9
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=300 + seed + nx, with_f=with_f)
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        d = to_dev(p)
        x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=want_gains)
        assert _lib.last_kernel_name().startswith("void dmpc::lqr_wide_kernel<%d, %d" % inst)
        assert_close(npy(x), xr, TOL_PRIMAL, "x")
        assert_close(npy(u), ur, TOL_PRIMAL, "u")
        if want_gains:
            Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
            assert_close(npy(Ks), Ksr, TOL_PRIMAL, "Ks")
            assert_close(npy(ks), ksr, TOL_PRIMAL, "ks")
        rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
        x2, u2 = rec.forward(*rec.backward())          # the container kernels (the path of rounds 2-4)
        assert_close(npy(x), npy(x2), 2e-5, "x against the separate sweeps")
        assert_close(npy(u), npy(u2), 2e-5, "u against the separate sweeps")
    # the clamped variant (LQR_active, mpc/active_constrained_lqr.py:110-137) on the same kernel
    for (B, T, seed) in ((8, 6, 7), (36, 11, 8)):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed, with_f=False)
        act = np.random.RandomState(nx * 16 + nu + seed).rand(T, B, nu) < 0.4
        xr, ur = ompc.lqr_active_solve(np.zeros((B, nx)), p["C"], p["c"], p["F"], None, act, T, nx, nu)
        d = to_dev(p)
        x, u = LqrRecursion(torch.zeros_like(d["x_init"]), d["C"], d["c"], d["F"], None, T, nx, nu,
                            u_zero_Index=torch.as_tensor(act).cuda()).solve_recursion()
        assert _lib.last_kernel_name().startswith("void dmpc::lqr_wide_kernel<%d, %d" % inst)
        assert ", true, false>(" in _lib.last_kernel_name()     # <..., PAD, MASKED = true, MPC = false>
        assert_close(npy(x), xr, TOL_PRIMAL, "x active")
        assert_close(npy(u), ur, TOL_PRIMAL, "u active")
        assert np.all(npy(u)[act] == 0)
    # a batch below one wavefront of four trajectories does not take it (padded: nor a ragged one)
    assert lib.dmpc_lqr_solve_path(6, 3, nx, nu) == 7
    if padded:
        assert lib.dmpc_lqr_solve_path(6, 7, nx, nu) == 7


@pytest.mark.parametrize("dims", [(6, 3), (13, 2)])
def test_container_long_horizon_and_full_batch(dims):
    """gains through HBM (the horizon does not fit in LDS) and a batch that fills the chip, sampled against the oracle"""
    nx, nu = dims
    B, T = 2, 320
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=8)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    B, T = 4096, 20
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=9)
    d = to_dev(p)
    x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
    idx = np.arange(0, B, 311)
    xr, ur = olqr.lqr_solve(p["x_init"][idx], p["C"][:, idx], p["c"][:, idx], p["F"][:, idx], p["f"][:, idx], T, nx, nu)
    assert_close(npy(x)[:, idx], xr, TOL_PRIMAL, "x")
    assert_close(npy(u)[:, idx], ur, TOL_PRIMAL, "u")


def test_ragged_batch_and_T_slices_of_F():
    """B not a multiple of the 16 trajectories per workgroup; F given with T slices (Boyd_lqr.py:29-32)"""
    B, T, nx, nu = 37, 6, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=11)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    F_T = torch.cat((d["F"], torch.full_like(d["F"][:1], float("nan"))), dim=0)   # slice T-1 must not be read
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], F_T, d["f"], T, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


def test_batch_of_one_T_two():
    p = synthetic.make_lqr_problem(1, 2, 3, 1, seed=4)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], 2, 3, 1)
    d = to_dev(p)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], 2, 3, 1).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL)
    assert_close(npy(u), ur, TOL_PRIMAL)


@pytest.mark.parametrize("shape", [(3, 300, 3, 1), (2, 260, 8, 2)])
def test_long_horizon_spills_gains_to_hbm(shape):
    """gains no longer fit in LDS -> the kernel hands them to the forward sweep through HBM"""
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=8)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


@pytest.mark.parametrize("shape", [(6, 8, 3, 1), (8, 10, 8, 2), (4, 5, 3, 2), (3, 6, 5, 3), (2, 5, 32, 8)])
def test_active_set_lqr_against_oracle(shape):
    """LQR_active (mpc/active_constrained_lqr.py): clamped controls masked out of the gains"""
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=21, with_f=False)
    rng = np.random.RandomState(5)
    act = rng.rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(np.zeros((B, nx)), p["C"], p["c"], p["F"], None, act, T, nx, nu)
    d = to_dev(p)
    rec = LqrRecursion(torch.zeros_like(d["x_init"]), d["C"], d["c"], d["F"], None, T, nx, nu,
                       u_zero_Index=torch.as_tensor(act).cuda())
    x, u = rec.solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    assert np.all(npy(u)[act] == 0)


def test_headline_shape_properties_and_sampled_oracle():
    """BASELINE.json configs[2]: B=4096, T=50, nx=8, nu=2.  Size-independent properties on the full
    batch + the oracle on a sample of trajectories."""
    B, T, nx, nu = 4096, 50, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d = to_dev(p)
    x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=True)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all() and torch.isfinite(u).all()
    # (1) the rollout obeys the dynamics it was given: x_{t+1} = F_t [x_t;u_t] + f_t
    tau = torch.cat((x, u), dim=2)
    nxt = torch.einsum("tbij,tbj->tbi", d["F"], tau[:-1]) + d["f"]
    assert float((nxt - x[1:]).abs().max()) <= 1e-4 * max(1.0, float(x.abs().max()))
    assert torch.equal(x[0], d["x_init"])
    # (2) u_t = K_t x_t + k_t with the returned gains
    ufb = torch.einsum("tbij,tbj->tbi", Ks, x) + ks
    assert float((ufb - u).abs().max()) <= 1e-4 * max(1.0, float(u.abs().max()))
    # (3) shard invariance: solving a batch slice gives that slice (what multi-GPU sharding relies on)
    sl = slice(1024, 1024 + 512)
    xs, us, _, _ = solve_device(d["C"][:, sl].contiguous(), d["c"][:, sl].contiguous(),
                                d["F"][:, sl].contiguous(), d["f"][:, sl].contiguous(),
                                d["x_init"][sl].contiguous(), None, T, nx, nu)
    assert torch.equal(xs, x[:, sl]) and torch.equal(us, u[:, sl])
    # (4) stationarity of the LQR Lagrangian in u at the last step: C_uu u + C_ux x + c_u = 0
    CT = d["C"][T - 1]
    res = torch.einsum("bij,bj->bi", CT[:, nx:, :], tau[T - 1]) + d["c"][T - 1][:, nx:]
    assert float(res.abs().max()) <= 1e-3
    # (5) the oracle on a sample
    idx = np.random.RandomState(1).choice(B, 48, replace=False)
    xr, ur = olqr.lqr_solve(p["x_init"][idx], p["C"][:, idx], p["c"][:, idx], p["F"][:, idx], p["f"][:, idx],
                            T, nx, nu)
    assert_close(npy(x)[:, idx], xr, TOL_PRIMAL, "x")
    assert_close(npy(u)[:, idx], ur, TOL_PRIMAL, "u")


def test_config5_shard_properties_and_sampled_oracle():
    """BASELINE.json configs[4]: nx=32, nu=8, T=50, batch 65536 over 8 GPUs = 8192 trajectories per GPU (one
    shard, drawn on the device as bench.py does).  Size-independent properties on the whole shard + the oracle
    on a sample of its trajectories."""
    import bench
    B, T, nx, nu = 8192, 50, 32, 8
    _, d = bench.make_inputs(B, T, nx, nu, 3, torch.device("cuda"))
    x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=True)
    torch.cuda.synchronize()
    assert torch.isfinite(x).all() and torch.isfinite(u).all()
    tau = torch.cat((x, u), dim=2)
    nxt = torch.einsum("tbij,tbj->tbi", d["F"], tau[:-1]) + d["f"]                  # x_{t+1} = F_t [x_t;u_t] + f_t
    assert float((nxt - x[1:]).abs().max()) <= 1e-4 * max(1.0, float(x.abs().max()))
    assert torch.equal(x[0], d["x_init"])
    ufb = torch.einsum("tbij,tbj->tbi", Ks, x) + ks                                 # u_t = K_t x_t + k_t
    assert float((ufb - u).abs().max()) <= 1e-4 * max(1.0, float(u.abs().max()))
    sl = slice(4096, 4096 + 256)                                                    # shard invariance
    xs, us, _, _ = solve_device(d["C"][:, sl].contiguous(), d["c"][:, sl].contiguous(), d["F"][:, sl].contiguous(),
                                d["f"][:, sl].contiguous(), d["x_init"][sl].contiguous(), None, T, nx, nu)
    assert torch.equal(xs, x[:, sl]) and torch.equal(us, u[:, sl])
    CT = d["C"][T - 1]                                                              # stationarity at the last step
    res = torch.einsum("bij,bj->bi", CT[:, nx:, :], tau[T - 1]) + d["c"][T - 1][:, nx:]
    assert float(res.abs().max()) <= 1e-3
    idx = torch.as_tensor(np.random.RandomState(5).choice(B, 12, replace=False), device="cuda")
    p = {k: npy(d[k].index_select(0 if k == "x_init" else 1, idx)).astype(np.float64) for k in d}
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    assert_close(npy(x.index_select(1, idx)), xr, TOL_PRIMAL, "x")
    assert_close(npy(u.index_select(1, idx)), ur, TOL_PRIMAL, "u")


def test_reference_style_dynamics_still_within_primal_tolerance():
    """A = I + 0.2*randn (the reference's initialiser, rho(A) > 1): x,u stay within 1e-4"""
    B, T, nx, nu = 32, 50, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=2, reference_style_A=True)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


def test_singular_quu_sets_info_flag():
    B, T, nx, nu = 4, 3, 2, 1
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1, with_f=False)
    p["C"][T - 1, 2, nx:, nx:] = 0.0
    d = to_dev(p)
    rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], None, T, nx, nu)
    rec.solve_recursion()
    info = rec.info.cpu().numpy()
    assert info[2] & _lib.INFO_SINGULAR and info[2] & _lib.INFO_NONFINITE
    assert (info[[0, 1, 3]] == 0).all()


def test_cpu_and_numpy_inputs_round_trip():
    p = synthetic.make_lqr_problem(3, 4, 3, 1, seed=9)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], 4, 3, 1)
    x, u = LqrRecursion(p["x_init"], p["C"], p["c"], p["F"], p["f"], 4, 3, 1).solve_recursion()
    assert (not x.is_cuda) and x.dtype == torch.float64
    assert_close(x.numpy(), xr, TOL_PRIMAL)
    assert_close(u.numpy(), ur, TOL_PRIMAL)


@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_batched_lu_golden(n):
    """util.py:462-528 drop-ins against the reference's torch.lu / lu_solve outputs"""
    from chainer_differentiable_mpc_amd import batch_lu_factor, batch_lu_solve
    g = np.load(os.path.join(GOLDEN, "lu_n%d.npz" % n))
    A = torch.as_tensor(g["A"], dtype=torch.float32).cuda()
    LU, piv = batch_lu_factor(A)
    assert piv.dtype == torch.int32
    np.testing.assert_array_equal(piv.cpu().numpy(), g["piv"])
    assert_close(npy(LU), g["LU"], 1e-5, "LU")
    x2 = batch_lu_solve((LU, piv), torch.as_tensor(g["b2"], dtype=torch.float32).cuda())
    x3 = batch_lu_solve((LU, piv), torch.as_tensor(g["b3"], dtype=torch.float32).cuda())
    assert list(x2.shape) == list(g["x2"].shape) and list(x3.shape) == list(g["x3"].shape)
    assert_close(npy(x2), g["x2"], 1e-4, "x2")
    assert_close(npy(x3), g["x3"], 1e-4, "x3")


def test_batched_lu_large_n_generic_path():
    from chainer_differentiable_mpc_amd import batch_lu_factor, batch_lu_solve
    from oracle import linalg
    rng = np.random.RandomState(3)
    A = rng.randn(5, 12, 12).astype(np.float32).astype(np.float64)
    b = rng.randn(5, 12, 2).astype(np.float32).astype(np.float64)
    LUr, pivr = linalg.batch_lu_factor(A)
    xr = linalg.batch_lu_solve((LUr, pivr), b)
    LU, piv = batch_lu_factor(torch.as_tensor(A, dtype=torch.float32).cuda())
    np.testing.assert_array_equal(piv.cpu().numpy(), pivr)
    x = batch_lu_solve((LU, piv), torch.as_tensor(b, dtype=torch.float32).cuda())
    assert_close(npy(x), xr, TOL_PRIMAL, "x")


@pytest.mark.parametrize("shape", [(5, 1, 8, 2), (1, 1, 3, 1), (2, 1, 32, 8)])
def test_single_timestep_horizon(shape):
    """T = 1: no dynamics at all (F has zero slices), x_0 = x_init, u_0 = K_0 x_0 + k_0 from the last-step cost"""
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, 2, nx, nu, seed=6)
    C, c = p["C"][:1], p["c"][:1]
    xr, ur = olqr.lqr_solve(p["x_init"], C, c, p["F"][:0], None, 1, nx, nu)
    d = to_dev(dict(C=C, c=c, x_init=p["x_init"]))
    F0 = torch.empty((0, B, nx, nx + nu), dtype=torch.float32, device="cuda")
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], F0, None, 1, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


def test_two_streams_do_not_share_scratch():
    """the per-device scratch of round 1 was shared by every call whatever stream it ran on; it is keyed on the stream
    now: two (32,8) solves (gains through the workspace) and two KKT gradients enqueued on two streams at once give
    the answers of the same calls made one after the other"""
    from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device
    from chainer_differentiable_mpc_amd.lqr_recursion import _ws_cache, solve_device
    probs = []
    for seed, (B, T, nx, nu) in ((1, (96, 12, 32, 8)), (2, (64, 12, 32, 8))):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed)
        probs.append((to_dev(p), T, nx, nu))
    ref = []
    for d, T, nx, nu in probs:
        x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
        g = kkt_grad_device(d["C"], d["c"], d["F"], x, u, torch.ones_like(x), torch.ones_like(u), T, nx, nu)
        ref.append((x.clone(), u.clone(), [t.clone() for t in g]))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [None, None]
    for rep in range(3):
        for i, ((d, T, nx, nu), st) in enumerate(zip(probs, streams)):
            with torch.cuda.stream(st):
                x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
                g = kkt_grad_device(d["C"], d["c"], d["F"], x, u, torch.ones_like(x), torch.ones_like(u), T, nx, nu)
                out[i] = (x, u, g)
    torch.cuda.synchronize()
    assert len({k[2] for k in _ws_cache}) >= 3          # default stream + the two side streams own separate buffers
    for (x, u, g), (xr, ur, gr) in zip(out, ref):
        assert torch.equal(x, xr) and torch.equal(u, ur)
        for a, b in zip(g, gr):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dims", [(64, 16), (40, 30), (70, 3), (100, 1)], ids=lambda d: "%dx%d" % d)
def test_shapes_beyond_64_columns_against_oracle(dims):
    """VERDICT r03 "any shape": the reference has no size limit (lqr/lqr_recursion.py:69-209); problems with
    nx + nu + 1 > 64 used to be refused (DMPC_E_UNSUPPORTED).  Kernel family 5 (lqr_tiled.hpp: a workgroup per trajectory,
    runtime dimensions, the matrices in the caller's workspace) takes them: solve, gains, separate sweeps and the clamped
    variant against the oracle at the contract's 1e-4."""
    from chainer_differentiable_mpc_amd import _lib
    nx, nu = dims
    assert _lib.load().dmpc_lqr_kernel_family(nx, nu) == 5
    B, T = 5, 6
    for with_f in (True, False):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=nx + nu, with_f=with_f)
        d = to_dev(p)
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
        x, u = rec.solve_recursion()
        assert "lqr_tiled_kernel" in _lib.last_kernel_name()
        assert_close(npy(x), xr, TOL_PRIMAL, "x")
        assert_close(npy(u), ur, TOL_PRIMAL, "u")
        Ks, ks = rec.backward()
        assert_close(npy(torch.stack(Ks)), Ksr, TOL_PRIMAL, "Ks")
        assert_close(npy(torch.stack(ks)), ksr, TOL_PRIMAL, "ks")
        x2, u2 = rec.forward(Ks, ks)
        assert_close(npy(x2), xr, TOL_PRIMAL, "x fwd")
        assert_close(npy(u2), ur, TOL_PRIMAL, "u fwd")
    from oracle import mpc as ompc
    act = np.random.RandomState(nx).rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(p["x_init"], p["C"], p["c"], p["F"], None, act, T, nx, nu)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], None, T, nx, nu, u_zero_Index=torch.as_tensor(act).cuda()).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x active")
    assert_close(npy(u), ur, TOL_PRIMAL, "u active")
    assert np.all(npy(u)[act] == 0)
