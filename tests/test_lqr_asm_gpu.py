"""GPU parity of the generated instruction-stream solve (csrc/gen_lqr_asm.py -> lqr_asm_kernel) through the C-ABI:
both forward variants (F stash in accumulation registers / second LDS-DMA read), every optional input and
output, the horizon limits of each variant, ragged batches - against the numpy oracle (lqr/lqr_recursion.py)."""
import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import LqrRecursion, _lib, synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import lqr as olqr
from tests.helpers import TOL_PRIMAL, assert_close, npy, to_dev

pytestmark = pytest.mark.gpu

STASH, RING, DMA, GHBM = 4, 3, 2, 6


def run_case(B, T, nx, nu, with_f, want_gains, seed=5):
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed, with_f=with_f)
    d = to_dev(p)
    info = torch.zeros(B, dtype=torch.int32, device="cuda")
    x, u, Ks, ks = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu,
                                want_gains=want_gains, info=info)
    torch.cuda.synchronize()
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    assert int(info.abs().max().item()) == 0
    if want_gains:
        Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        assert_close(npy(Ks), Ksr, TOL_PRIMAL, "Ks")
        assert_close(npy(ks), ksr, TOL_PRIMAL, "ks")


@pytest.mark.parametrize("shape,path", [
    ((64, 10, 8, 2), STASH), ((4, 2, 8, 2), STASH), ((5, 3, 8, 2), STASH), ((9, 5, 8, 2), STASH),
    ((131, 50, 8, 2), STASH), ((6, 51, 8, 2), STASH),      # 51 = the last horizon that fits the stash registers
    ((6, 52, 8, 2), RING), ((5, 74, 8, 2), RING),          # 74 = the last horizon whose gains fit in LDS
    ((5, 75, 8, 2), GHBM), ((37, 100, 8, 2), GHBM), ((8, 200, 8, 2), GHBM), ((12, 300, 3, 1), GHBM),   # gain rows through HBM
    ((16, 20, 3, 1), RING), ((7, 9, 4, 2), STASH), ((11, 13, 6, 2), RING), ((8, 6, 2, 2), STASH),
    ((8, 6, 1, 1), STASH), ((8, 7, 2, 1), STASH), ((8, 6, 3, 2), RING)])
@pytest.mark.parametrize("with_f", [True, False])
@pytest.mark.parametrize("want_gains", [False, True])
def test_generated_stream_against_oracle(shape, path, with_f, want_gains):
    B, T, nx, nu = shape
    assert _lib.load().dmpc_lqr_solve_path(T, B, nx, nu) == path
    run_case(B, T, nx, nu, with_f, want_gains)
    if path == GHBM and not want_gains:      # <nx, nu, has_f, write_k, stash, masked, GHBM, save, affine, adj>
        assert _lib.last_kernel_name().endswith("false, false, false, true, false, false, false>(dmpc::LqrArgs)"), \
            _lib.last_kernel_name()


def test_batch_below_one_wave_takes_the_hip_kernel():
    assert _lib.load().dmpc_lqr_solve_path(10, 3, 8, 2) == 1
    run_case(3, 10, 8, 2, True, False)


def test_F_with_T_slices_last_slice_never_read():
    """F given with T slices (examples/Boyd_lqr.py:29-32): slice T-1 is poisoned"""
    B, T, nx, nu = 37, 6, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=11)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    F_T = torch.cat((d["F"], torch.full_like(d["F"][:1], float("nan"))), dim=0)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], F_T, d["f"], T, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


def test_non_symmetric_cost_matrices_are_followed_like_the_reference():
    """the reference never symmetrises C, V or Q (lqr_recursion.py:85-152); neither does the stream"""
    B, T, nx, nu = 12, 9, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=21)
    rng = np.random.RandomState(3)
    p["C"] = p["C"] + 0.05 * rng.standard_normal(p["C"].shape)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    x, u = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu).solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


@pytest.mark.parametrize("shape", [(8, 6, 8, 2), (8, 5, 3, 1)])
def test_singular_quu_and_nan_inputs_set_the_info_flags(shape):
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1, with_f=False)
    p["C"][T - 1, 2, nx:, nx:] = 0.0            # exact zero pivot at the first step of trajectory 2
    p["c"][1, 5, 0] = np.nan                    # a NaN cost term in trajectory 5
    d = to_dev(p)
    rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], None, T, nx, nu)
    rec.solve_recursion()
    info = rec.info.cpu().numpy()
    assert info[2] & _lib.INFO_SINGULAR and info[2] & _lib.INFO_NONFINITE
    assert info[5] & _lib.INFO_NONFINITE and not (info[5] & _lib.INFO_SINGULAR)
    assert (info[[0, 1, 3, 4, 6, 7]] == 0).all()


def test_repeated_launches_are_bitwise_reproducible():
    B, T, nx, nu = 256, 50, 8, 2
    d = to_dev(synthetic.make_lqr_problem(B, T, nx, nu, seed=2))
    outs = []
    for _ in range(3):
        x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
        outs.append((x.clone(), u.clone()))
    torch.cuda.synchronize()
    for x, u in outs[1:]:
        assert torch.equal(x, outs[0][0]) and torch.equal(u, outs[0][1])


@pytest.mark.parametrize("shape", [(6, 7, 32, 8), (9, 6, 8, 2), (5, 5, 12, 4)])
def test_quu_that_needs_row_interchanges(shape):
    """an SPD Quu whose leading entries are small, so that partial pivoting really moves rows (LAPACK getf2
    semantics of torch.lu / F.batch_inv, util.py:481, lqr_recursion.py:118) - at (32,8) this is the uniform-branch
    interchange of lqr_wave_mfma.hpp, at (8,2) the spelled-out 2x2 of the generated stream"""
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=13)
    rng = np.random.RandomState(7)
    scale = np.diag(np.linspace(0.3, 3.0, nu))          # later controls weigh more: |S[i][0]| > |S[0][0]| for some i
    for t in range(T):
        for b in range(B):
            G = rng.standard_normal((nu, nu))
            S = scale @ (G @ G.T + 0.5 * np.eye(nu)) @ scale
            p["C"][t, b, nx:, nx:] = S
            # keep the whole cost matrix positive definite
            p["C"][t, b, :nx, nx:] *= 0.1
            p["C"][t, b, nx:, :nx] *= 0.1
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    d = to_dev(p)
    rec = LqrRecursion(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu)
    x, u = rec.solve_recursion()
    assert int(rec.info.abs().max().item()) == 0
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")


@pytest.mark.parametrize("shape", [(64, 50, 8, 2), (8, 10, 8, 2), (12, 52, 8, 2), (16, 20, 3, 1), (8, 9, 4, 2),
                                   (8, 6, 2, 2), (8, 6, 1, 1), (8, 7, 2, 1), (6, 9, 8, 2), (7, 9, 8, 2)])
@pytest.mark.parametrize("with_f", [False, True])
def test_active_set_lqr_on_the_generated_stream(shape, with_f):
    """LQR_active (mpc/active_constrained_lqr.py:67-202): clamped controls get exactly-zero gain rows and controls.
    B * nu a multiple of 4 runs the masked generated stream, the last two shapes fall back to the HIP kernel."""
    from chainer_differentiable_mpc_amd import LQR_active
    from oracle import mpc as ompc
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=31, with_f=with_f)
    rng = np.random.RandomState(9)
    act = rng.rand(T, B, nu) < 0.4
    xr, ur = ompc.lqr_active_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], act, T, nx, nu)
    d = to_dev(p)
    rec = LQR_active(d["x_init"], d["C"], d["c"], d["F"], d["f"], T, nx, nu, u_zero_Index=torch.as_tensor(act).cuda())
    x, u = rec.solve_recursion()
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    assert np.all(npy(u)[act] == 0)
