"""GPU parity: analytic KKT gradient (DiffLqr.backward, lqr/differentiable_lqr.py:78-142) through the
C-ABI against the golden vectors recorded from the reference and against the numpy oracle.  Row B."""
import glob
import os

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import DiffLqr, LqrNet, synthetic
from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
from oracle import kkt as okkt
from oracle import lqr as olqr
from tests.helpers import GOLDEN, TOL_COSTATE, TOL_PRIMAL, assert_close, npy, to_dev

pytestmark = pytest.mark.gpu

LQR_FILES = sorted(glob.glob(os.path.join(GOLDEN, "lqr_*.npz")))
KEYS = ("d_x_init", "dC", "dc", "dF", "df")
TOLS = dict(d_x_init=TOL_COSTATE, dC=TOL_PRIMAL, dc=TOL_PRIMAL, dF=TOL_COSTATE, df=TOL_COSTATE)


@pytest.mark.parametrize("path", LQR_FILES, ids=[os.path.basename(p) for p in LQR_FILES])
def test_kkt_gradient_matches_reference_golden(path):
    g = np.load(path)
    B, T, nx, nu = int(g["B"]), int(g["T"]), int(g["nx"]), int(g["nu"])
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=int(g["seed"]), with_f=bool(g["with_f"]))
    d = to_dev(p)
    node = DiffLqr(T, B, nx, nu)
    x, u = node.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    assert_close(npy(x), g["x"], TOL_PRIMAL, "x")
    gx = torch.as_tensor(g["grad_x"], dtype=torch.float32).cuda()
    gu = torch.as_tensor(g["grad_u"], dtype=torch.float32).cuda()
    out = node.backward((0, 1, 2, 3, 4), (gx, gu))
    for got, key in zip(out, KEYS):
        assert_close(npy(got), g[key], TOLS[key], key)


@pytest.mark.parametrize("shape", [(4, 6, 8, 2), (3, 5, 5, 3), (2, 4, 32, 8)])
def test_strict_math_variant_against_oracle(shape):
    B, T, nx, nu = shape
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=12)
    rng = np.random.RandomState(13)
    gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
    gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=True)
    d = to_dev(p)
    node = DiffLqr(T, B, nx, nu, strict_math=True)
    node.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    out = node.backward((0, 1, 2, 3, 4), (torch.as_tensor(gx).cuda(), torch.as_tensor(gu).cuda()))
    for got, want, key in zip(out, ref, KEYS):
        assert_close(npy(got), want, TOLS[key], key)
    # the symmetric dC really is symmetric, the faithful one is not
    dC = npy(out[1])
    assert np.abs(dC - np.swapaxes(dC, 2, 3)).max() <= 1e-6 * max(1.0, np.abs(dC).max())


@pytest.mark.parametrize("dims", [(3, 3), (6, 3), (5, 1), (13, 2), (9, 4), (14, 1), (2, 4), (10, 3), (5, 5), (16, 4), (20, 6), (31, 7),
                                  (17, 1), (24, 8), (32, 7), (25, 4)], ids=lambda d: "%dx%d" % d)
@pytest.mark.parametrize("strict", [False, True], ids=["faithful", "strict"])
def test_container_shapes_against_oracle(dims, strict):
    """shapes without a specialisation: the second solve and the co-state sweep run padded inside a container kernel"""
    nx, nu = dims
    B, T = 21, 6
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=40 + nx)
    rng = np.random.RandomState(nx * 17 + nu)
    gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
    gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
    d = to_dev(p)
    node = DiffLqr(T, B, nx, nu, strict_math=strict)
    node.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    if nx > 16:   # (rollout and co-state sweep of these: the kernels that stage a step's blocks through LDS)
        from chainer_differentiable_mpc_amd import _lib
        if dims in ((24, 8), (24, 4), (32, 4)):   # a size that IS one of the sweep's instances: the exact kernel, rollout included
            assert _lib.last_kernel_name().startswith("void dmpc::lqr_tile16_kernel<%d, %d, true>" % dims)
        else:
            assert _lib.last_kernel_name().startswith("dmpc::lqr_staged_forward_kernel")
    out = node.backward((0, 1, 2, 3, 4), (torch.as_tensor(gx).cuda(), torch.as_tensor(gu).cuda()))
    if nx > 16:
        assert _lib.last_kernel_name().startswith("dmpc::costate_staged_kernel")
    for got, want, key in zip(out, ref, KEYS):
        assert_close(npy(got), want, TOLS[key], key)


@pytest.mark.parametrize("case", [(5, 2, 20, 6, True), (3, 3, 17, 2, False), (2, 41, 20, 6, True), (9, 4, 31, 8, False)],
                         ids=lambda c: "B%d_T%d_%dx%d" % c[:4])
def test_staged_kernels_at_short_and_ring_wrapping_horizons(case):
    """lqr_staged_forward_kernel / costate_staged_kernel (17+ states): two steps (the whole horizon inside the prologue's
    fetches), three (the ring exactly), a horizon that wraps it many times; with and without f"""
    B, T, nx, nu, with_f = case
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=500 + T, with_f=with_f)
    rng = np.random.RandomState(T * 31 + nx)
    gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
    gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu)
    d = to_dev(p)
    node = DiffLqr(T, B, nx, nu)
    x, u = node.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    assert_close(npy(x), xr, TOL_PRIMAL, "x")
    assert_close(npy(u), ur, TOL_PRIMAL, "u")
    out = node.backward((0, 1, 2, 3, 4), (torch.as_tensor(gx).cuda(), torch.as_tensor(gu).cuda()))
    for got, want, key in zip(out, ref, KEYS):
        if want is not None and got is not None:
            assert_close(npy(got), want, TOLS[key], key)


@pytest.mark.parametrize("dims", [(16, 4), (16, 8), (12, 8), (13, 3), (15, 1), (14, 2), (10, 6), (9, 8), (8, 8), (12, 5), (12, 4), (16, 2),
                                  (15, 7), (13, 5), (16, 6), (13, 2), (9, 4), (14, 1), (11, 4), (12, 1), (6, 7)],
                         ids=lambda d: "%dx%d" % d)
@pytest.mark.parametrize("strict", [False, True], ids=["faithful", "strict"])
def test_wide_costate_kernel_against_oracle(dims, strict):
    """13 to 31 elements of tau, at most 16 states: the gradient's second solve on lqr_wide_kernel (16 and more; below, the
    16-lane containers) and its co-state sweep on costate_wide_kernel (four trajectories per wavefront, tau in two registers) - whole batches (a ragged one takes the
    wavefront-per-trajectory container: compared with it too), short and ring-wrapping horizons."""
    from chainer_differentiable_mpc_amd import _lib
    nx, nu = dims
    for (B, T, seed) in ((8, 6, 1), (36, 2, 2), (4, 23, 3)):
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=60 + nx + seed)
        rng = np.random.RandomState(nx * 19 + nu + seed)
        gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
        gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
        ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
        d = to_dev(p)
        node = DiffLqr(T, B, nx, nu, strict_math=strict)
        node.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
        out = node.backward((0, 1, 2, 3, 4), (torch.as_tensor(gx).cuda(), torch.as_tensor(gu).cuda()))
        inst = dims if dims in ((16, 4), (16, 8), (12, 8)) else ((16, 4) if nu <= 4 else ((12, 8) if nx <= 12 else (16, 8)))   # (padded inside these)
        assert _lib.last_kernel_name().startswith("void dmpc::costate_wide_kernel<%d, %d" % inst)
        for got, want, key in zip(out, ref, KEYS):
            assert_close(npy(got), want, TOLS[key], key)


@pytest.mark.parametrize("dims", [(8, 2), (16, 4), (13, 3), (16, 8)], ids=lambda d: "%dx%d" % d)
def test_misaligned_solution_views_give_the_aligned_result(dims):
    """x, u and the upstream gradients four bytes off a 16-byte boundary (contiguous views the C-ABI accepts): the LDS-DMA
    co-state kernels need 16-byte chunks, so these calls take the other kernels - same numbers within the parity bar."""
    nx, nu = dims
    B, T = 8, 7
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=300 + nx)
    d = to_dev(p)
    x, u = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)[:2]
    g = torch.Generator(device="cpu").manual_seed(nx * 7 + nu)
    gx, gu = torch.randn(T, B, nx, generator=g).cuda(), torch.randn(T, B, nu, generator=g).cuda()

    def off(t):
        buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4 and v.is_contiguous()
        return v
    want = kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu)
    got = kkt_grad_device(d["C"], d["c"], d["F"], off(x), off(u), off(gx), off(gu), T, nx, nu)
    torch.cuda.synchronize()
    for a, b, key in zip(got, want, KEYS):
        assert_close(npy(a), npy(b), TOLS[key], key)


def test_autograd_through_lqrnet_reproduces_the_notebook_anchor():
    """examples/LQRnet.ipynb:184 - loss 0.661925 at iteration 0, dynamics mse 4.774785 after the first
    RMSprop step - with forward AND backward on the HIP path, driven by torch.autograd."""
    from tests.lqrnet_anchor import problem
    q = problem()
    T, nx, nu, B = q["T"], q["nx"], q["nu"], q["B"]
    dt = torch.float64
    F_e = torch.as_tensor(np.concatenate((q["A_e"], q["B_e"]), axis=1), dtype=dt).cuda()
    C = torch.as_tensor(q["C"], dtype=dt).cuda()
    c = torch.as_tensor(q["c"], dtype=dt).cuda()
    x0 = torch.as_tensor(q["x_init"], dtype=dt).cuda()
    net = LqrNet(T, B, nx, nu, seed=2).cuda()
    np.testing.assert_allclose(net.A.detach().cpu().numpy(), q["A"])      # same draws as the reference
    expert = DiffLqr(T, B, nx, nu)
    x_true, u_true = expert.forward((x0, C, c, F_e.expand(T - 1, B, nx, nx + nu), None))
    x_pred, u_pred = net((x0, C, c, None))
    loss = ((u_true - u_pred) ** 2).mean() + ((x_true - x_pred) ** 2).mean()
    assert abs(float(loss) - 0.661925) < 5e-6
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-2, alpha=0.99, eps=1e-8)   # chainer.optimizers.RMSprop defaults
    opt.zero_grad()
    loss.backward()
    opt.step()
    A_e = torch.as_tensor(q["A_e"], dtype=dt).cuda()
    B_e = torch.as_tensor(q["B_e"], dtype=dt).cuda()
    mse = ((net.A - A_e) ** 2).mean() + ((net.B - B_e) ** 2).mean()
    assert abs(float(mse) - 4.774785) < 5e-5


def test_gradients_flow_to_every_input_and_none_f():
    B, T, nx, nu = 5, 6, 4, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=31, with_f=True)
    d = {k: v.clone().requires_grad_(True) for k, v in to_dev(p).items()}
    node = DiffLqr(T, B, nx, nu)
    x, u = node.apply((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    w = torch.linspace(-1, 1, x.numel(), device="cuda").reshape(x.shape)
    ((w * x).sum() + u.sum()).backward()
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, npy(w), np.ones((T, B, nu)), T, nx, nu)
    for name, want, key in zip(("x_init", "C", "c", "F", "f"), ref, KEYS):
        assert_close(npy(d[name].grad), want, TOLS[key], name)


def test_headline_shape_sampled_oracle_and_linearity():
    """BASELINE.json configs[2] at full size: oracle on a sample + linearity in the upstream gradient"""
    B, T, nx, nu = 4096, 50, 8, 2
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d = to_dev(p)
    x, u, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
    gx = torch.ones((T, B, nx), device="cuda")
    gu = torch.ones((T, B, nu), device="cuda")
    out1 = kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu)
    out1 = [o.clone() for o in out1]
    out3 = kkt_grad_device(d["C"], d["c"], d["F"], x, u, 3 * gx, 3 * gu, T, nx, nu)
    for a, b3, key in zip(out1, out3, KEYS):
        assert torch.isfinite(a).all(), key
        scale = max(1.0, float(a.abs().max()))
        assert float((3 * a - b3).abs().max()) <= 2e-4 * 3 * scale, key
    idx = np.random.RandomState(2).choice(B, 24, replace=False)
    xr, ur = olqr.lqr_solve(p["x_init"][idx], p["C"][:, idx], p["c"][:, idx], p["F"][:, idx], p["f"][:, idx], T, nx, nu)
    ref = okkt.difflqr_backward(p["x_init"][idx], p["C"][:, idx], p["c"][:, idx], p["F"][:, idx], xr, ur,
                                np.ones((T, len(idx), nx)), np.ones((T, len(idx), nu)), T, nx, nu)
    sel = [npy(out1[0])[idx], npy(out1[1])[:, idx], npy(out1[2])[:, idx], npy(out1[3])[:, idx], npy(out1[4])[:, idx]]
    for got, want, key in zip(sel, ref, KEYS):
        assert_close(got, want, TOLS[key], key)


def test_two_forwards_before_backward_keep_their_own_state():
    """one DiffLqr object (as in LqrNet) applied twice before any backward - summed minibatches, a validation pass
    in between: each graph differentiates its own retained tensors"""
    import torch
    from chainer_differentiable_mpc_amd import DiffLqr
    T, B, nx, nu = 6, 5, 4, 2
    layer = DiffLqr(T, B, nx, nu)
    probs = [synthetic.make_lqr_problem(B, T, nx, nu, seed=s) for s in (21, 22)]
    grads = []
    for order in ("separate", "interleaved"):
        Fs = [torch.as_tensor(p["F"], dtype=torch.float32, device="cuda").requires_grad_(True) for p in probs]
        outs = []
        for p, F in zip(probs, Fs):
            d = to_dev(p)
            x, u = layer.apply((d["x_init"], d["C"], d["c"], F, d["f"]))
            outs.append(x.sum() + (u ** 2).sum())
            if order == "separate":
                outs[-1].backward()
        if order == "interleaved":
            (outs[0] + outs[1]).backward()
        grads.append([npy(F.grad) for F in Fs])
    for a, b in zip(grads[0], grads[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dims", [(64, 16), (40, 30)], ids=lambda d: "%dx%d" % d)
@pytest.mark.parametrize("strict", [False, True], ids=["faithful", "strict"])
def test_shapes_beyond_64_columns_against_oracle(dims, strict):
    """the KKT gradient at shapes beyond a wavefront's 64 columns (refused until round 4): second solve on the tiled kernel
    (lqr_tiled.hpp), co-state sweep on the runtime-dimension kernel - lqr/differentiable_lqr.py:78-142 has no size limit"""
    nx, nu = dims
    B, T = 4, 5
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=40 + nx)
    rng = np.random.RandomState(nx * 17 + nu)
    gx = rng.randn(T, B, nx).astype(np.float32).astype(np.float64)
    gu = rng.randn(T, B, nu).astype(np.float32).astype(np.float64)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    ref = okkt.difflqr_backward(p["x_init"], p["C"], p["c"], p["F"], xr, ur, gx, gu, T, nx, nu, strict_math=strict)
    d = to_dev(p)
    node = DiffLqr(T, B, nx, nu, strict_math=strict)
    node.forward((d["x_init"], d["C"], d["c"], d["F"], d["f"]))
    out = node.backward((0, 1, 2, 3, 4), (torch.as_tensor(gx).cuda(), torch.as_tensor(gu).cuda()))
    for got, want, key in zip(out, ref, KEYS):
        assert_close(npy(got), want, TOLS[key], key)
