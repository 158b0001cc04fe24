"""GPU parity: every kernel family behind `MPCstep.forward` (mpc/mpc_step.py:70-286) against the numpy oracle, and the
families against each other.  The launcher picks, in this order (csrc/mpc_api.hip):
  * the generated instruction stream with the box QP inside (mpc_asm_kernel.hpp): shapes of the generator, B % 4 == 0;
  * the HIP kernels fed by an LDS-DMA ring (mpc_dma_kernels.hpp, mpc_kernels.hpp `DMA`): other register-resident shapes;
  * the register-bank HIP kernels: ragged batches, unaligned views;
  * the runtime-dimension kernels (mpc_generic.hpp) - covered by test_mpc_step_gpu.py.
DMPC_NO_MPC_ASM / DMPC_NO_MPC_DMA switch families off (read once per process), which is how the same problem is sent
down each of them here."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost, synthetic
from oracle import mpc as ompc
from tests.helpers import TOL_STEP, assert_close, npy

pytestmark = pytest.mark.gpu
TOL = TOL_STEP      # 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev(a):
    return None if a is None else torch.as_tensor(a, dtype=torch.float32, device="cuda")


def problem(B, T, nx, nu, bound, seed):
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed, with_f=True)
    lo, hi = -bound * np.ones((T, B, nu)), bound * np.ones((T, B, nu))
    u0 = np.zeros((T, B, nu))
    xs = [p["x_init"]]
    for t in range(T - 1):
        xs.append(np.einsum("bij,bj->bi", p["F"][t], np.concatenate((xs[t], u0[t]), axis=1)) + p["f"][t])
    x0 = np.stack(xs).astype(np.float32).astype(np.float64)
    return p, lo, hi, u0, x0


# (B, T, nx, nu, bound, need_expand): the kernel family each case reaches is in the id
CASES = [
    pytest.param(8, 9, 8, 2, 0.25, True, id="stream-recentring-8-2"),
    pytest.param(8, 9, 8, 2, 0.25, False, id="stream-with-f-8-2"),
    pytest.param(12, 8, 3, 1, 0.5, True, id="stream-recentring-3-1"),
    pytest.param(4, 6, 3, 2, 0.375, False, id="stream-with-f-3-2"),
    pytest.param(4, 2, 3, 1, 0.5, True, id="stream-shortest-horizon-3-1"),
    pytest.param(4, 3, 8, 2, 0.25, False, id="stream-three-steps-8-2"),
    pytest.param(4, 4, 8, 2, 0.25, True, id="stream-four-steps-8-2"),
    pytest.param(4, 7, 4, 2, 0.25, True, id="stream-recentring-4-2"),
    pytest.param(8, 6, 4, 2, 0.25, False, id="stream-with-f-4-2"),
    pytest.param(4, 6, 2, 2, 0.375, True, id="stream-recentring-2-2"),
    pytest.param(4, 5, 2, 2, 0.375, False, id="stream-with-f-2-2"),
    pytest.param(8, 6, 1, 1, 0.5, True, id="stream-recentring-1-1"),
    pytest.param(4, 6, 1, 1, 0.5, False, id="stream-with-f-1-1"),
    pytest.param(4, 7, 2, 1, 0.5, True, id="stream-recentring-2-1"),
    pytest.param(4, 7, 2, 1, 0.5, False, id="stream-with-f-2-1"),
    pytest.param(8, 5, 3, 2, 0.375, True, id="stream-recentring-3-2"),
    pytest.param(8, 7, 6, 2, 0.25, True, id="dma-ring-sweep-stream-search-6-2"),
    pytest.param(4, 6, 6, 2, 0.25, False, id="dma-ring-sweep-stream-search-with-f-6-2"),
    pytest.param(4, 6, 4, 4, 0.375, False, id="dma-ring-4-4"),
    pytest.param(8, 6, 8, 4, 0.25, True, id="dma-ring-8-4"),
    pytest.param(4, 5, 12, 3, 0.25, True, id="dma-ring-12-3"),
    pytest.param(6, 8, 8, 2, 0.25, True, id="register-banks-ragged-8-2"),
    pytest.param(5, 7, 3, 1, 0.5, False, id="register-banks-ragged-3-1"),
]


@pytest.mark.parametrize("B,T,nx,nu,bound,need_expand", CASES)
def test_forward_and_backward_against_the_oracle(B, T, nx, nu, bound, need_expand):
    p, lo, hi, u0, x0 = problem(B, T, nx, nu, bound, seed=11)
    xr, ur, bo, fo, Ksr, ksr = ompc.mpc_forward(p["C"], p["c"], p["F"], p["f"], u0, x0, lo, hi,
                                                ompc.QuadCost(p["C"], p["c"]), ompc.LinDx(p["F"], p["f"]), 0.2, 5,
                                                T, nx, nu, need_expand=need_expand, batch_coupled=False)
    step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
                   LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=need_expand)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, u = step.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
    assert_close(npy(step.ks), ksr, TOL, "ks")
    assert_close(npy(step.Ks), Ksr, TOL, "Ks")
    assert_close(npy(u), ur, TOL, "u")
    assert_close(npy(x), xr, TOL, "x")
    assert_close(npy(step.for_out.costs), fo.costs, TOL, "costs")
    active = (npy(u) == lo) | (npy(u) == hi)
    np.testing.assert_array_equal(active, (np.abs(ur - lo) <= 1e-8) | (np.abs(ur - hi) <= 1e-8))
    assert active.any() and not active.all()
    clamped = (np.abs(ur - lo) <= 1e-8) | (np.abs(ur - hi) <= 1e-8)     # gain rows of clamped controls: exactly zero
    assert np.all(npy(step.Ks)[np.broadcast_to(clamped[..., None], Ksr.shape) & (Ksr == 0)] == 0)


_RUNNER = r"""
import sys, warnings
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost
from tests.test_mpc_paths_gpu import problem, dev
B, T, nx, nu, bound, need_expand = (int(sys.argv[4]) if len(sys.argv) > 4 else 64), 12, 8, 2, 0.25, sys.argv[3] == "1"
p, lo, hi, u0, x0 = problem(B, T, nx, nu, bound, seed=5)
step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
               LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=need_expand)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    x, u = step.forward((dev(x0[0]), dev(p["C"]), dev(p["c"]), dev(p["F"]), dev(p["f"])))
np.savez(sys.argv[2], x=x.cpu().numpy(), u=u.cpu().numpy(), Ks=step.Ks.cpu().numpy(), ks=step.ks.cpu().numpy(),
         costs=step.for_out.costs.cpu().numpy(), nqp=np.int64(step.back_out.n_total_qp_iter))
"""


@pytest.mark.parametrize("need_expand", [True, False], ids=["recentring", "with_f"])
def test_the_kernel_families_agree_with_each_other(tmp_path, need_expand):
    """(8,2), B=64: the generated stream, the DMA-ring HIP kernels and the register-bank HIP kernels on the same problem
    (one process each - the switches are read once): same active sets, same QP pass totals, values to 2e-5 (the stream's
    gain solve uses v_rcp_f32 without the Newton step the HIP kernels add)."""
    outs = {}
    for name, env in (("stream", {}), ("dma", {"DMPC_NO_MPC_ASM": "1"}),
                      ("banks", {"DMPC_NO_MPC_ASM": "1", "DMPC_NO_MPC_DMA": "1"})):
        out = str(tmp_path / (name + ".npz"))
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", _RUNNER, ROOT, out, "1" if need_expand else "0"], env=e, cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    ref = outs["banks"]
    for name in ("dma", "stream"):
        o = outs[name]
        for key in ("x", "u", "Ks", "ks", "costs"):
            assert_close(o[key], ref[key], 2e-5, "%s: %s" % (name, key))
        np.testing.assert_array_equal(o["Ks"] == 0, ref["Ks"] == 0)     # the same clamped sets
        assert int(o["nqp"]) == int(ref["nqp"]), (name, int(o["nqp"]), int(ref["nqp"]))
    np.testing.assert_array_equal(outs["dma"]["Ks"], ref["Ks"])          # the two HIP families: the same arithmetic
    np.testing.assert_array_equal(outs["dma"]["u"], ref["u"])


@pytest.mark.parametrize("need_expand,B", [(True, 64), (False, 64), (True, 36)], ids=["recentring", "with_f", "ragged_workgroup"])
def test_the_step_as_one_launch_is_the_two_launches_bit_for_bit(tmp_path, need_expand, B):
    """`dmpc_mpc_step_forward` runs both generated streams in one launch (mpc_step_fused_kernel.hpp); with
    DMPC_NO_MPC_FUSED=1 it is the two launches of before.  Same instruction streams, so everything returned is identical to
    the bit - also when the last workgroup holds fewer than four wavefronts (B = 36: they leave before the barrier)."""
    outs = {}
    for name, env in (("one", {}), ("two", {"DMPC_NO_MPC_FUSED": "1"})):
        out = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", _RUNNER, ROOT, out, "1" if need_expand else "0", str(B)],
                           env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for key in ("x", "u", "Ks", "ks", "costs", "nqp"):
        np.testing.assert_array_equal(outs["one"][key], outs["two"][key], err_msg=key)
    assert np.isfinite(outs["one"]["x"]).all()


_RUNNER_LS = r"""
import sys, warnings
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost
from tests.test_mpc_paths_gpu import problem, dev
B, T, nx, nu, bound = 32, 9, 8, 2, 0.5
p, lo, hi, u0, x0 = problem(B, T, nx, nu, bound, seed=9)
rng = np.random.RandomState(3)
# gains that are NOT descent directions for most trajectories: the search walks many step sizes (some to the cap)
Ks = 0.6 * rng.standard_normal((T, B, nu, nx))
ks = 0.8 * rng.standard_normal((T, B, nu))
ks[:, ::4] = 0.0
Ks[:, ::4] *= 0.02
step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(p["C"]), dev(p["c"])),
               LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    x, u, fo = step.forward_rec(dev(Ks), dev(ks), step.true_cost, step.true_dynamics, 0.2, 5)
np.savez(sys.argv[2], x=x.cpu().numpy(), u=u.cpu().numpy(), costs=fo.costs.cpu().numpy(), objs=fo.objs.cpu().numpy(),
         alphas=step.alphas.cpu().numpy(), nls=step.n_ls_iter.cpu().numpy(), du=fo.full_du_norm.cpu().numpy())
"""


def test_long_line_searches_agree_between_the_families(tmp_path):
    """forward_rec with gains that are not descent directions: most trajectories walk many step sizes, the wavefronts
    of the stream run their pass loop with some rows finished and others still searching.  Stream against the
    register-bank HIP kernel: the same pass counts and step sizes, trajectories to 2e-5, and the oracle on top."""
    outs = {}
    for name, env in (("stream", {}), ("banks", {"DMPC_NO_MPC_ASM": "1", "DMPC_NO_MPC_DMA": "1"})):
        out = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", _RUNNER_LS, ROOT, out], env=dict(os.environ, **env), cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    a, b = outs["stream"], outs["banks"]
    assert int(b["nls"].max()) >= 8 and len(np.unique(b["nls"])) >= 3      # long and ragged searches
    np.testing.assert_array_equal(a["nls"], b["nls"])
    np.testing.assert_array_equal(a["alphas"], b["alphas"])
    for key in ("x", "u", "costs", "objs", "du"):
        assert_close(a[key], b[key], 2e-5, key)


_RUNNER_QP = r"""
import sys, warnings
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost
from tests.test_mpc_paths_gpu import problem, dev
B, T, nx, nu, bound = 64, 12, 8, 2, 0.05
p, lo, hi, u0, x0 = problem(B, T, nx, nu, bound, seed=21)
rng = np.random.RandomState(4)
C = p["C"].copy()
# strongly coupled controls and large linear terms: clamped sets change from step to step, the QP takes several
# passes and its Armijo search more than one trial
C[..., nx:, nx:] = np.array([[1.0, 0.97], [0.97, 1.0]]) * rng.uniform(0.3, 3.0, (T, B, 1, 1))
c = p["c"] + 2.0 * rng.standard_normal(p["c"].shape)
step = MPCstep(dev(u0), T, dev(hi), dev(lo), B, nx, nu, dev(x0), QuadCost(dev(C), dev(c)),
               LinDx(dev(p["F"]), dev(p["f"])), ls_decay=0.2, max_ls_iter=5, need_expand=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    Ks, ks, bo = step.backward_rec(dev(C), dev(c), dev(p["F"]), dev(p["f"]))
np.savez(sys.argv[2], Ks=Ks.cpu().numpy(), ks=ks.cpu().numpy(), nqp=step.n_qp_iter.cpu().numpy(), C=C, c=c)
"""


def test_hard_box_qps_agree_between_the_families_and_with_the_oracle(tmp_path):
    """backward_rec on a problem whose box QPs take several passes (coupled controls, tight bounds, large linear terms):
    the stream's in-line projected Newton against the HIP kernels' (same pass totals per trajectory, same clamped
    sets) and against the numpy oracle"""
    outs = {}
    for name, env in (("stream", {}), ("banks", {"DMPC_NO_MPC_ASM": "1", "DMPC_NO_MPC_DMA": "1"})):
        out = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", _RUNNER_QP, ROOT, out], env=dict(os.environ, **env), cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    a, b = outs["stream"], outs["banks"]
    T, B = a["ks"].shape[:2]
    assert b["nqp"].mean() > 2.0 * T and b["nqp"].max() >= 2.4 * T      # more than two passes per QP on average
    assert np.mean(a["nqp"] == b["nqp"]) > 0.95           # a pass more or less only where a test sits on its threshold
    assert np.abs(a["nqp"].astype(int) - b["nqp"].astype(int)).max() <= 2
    np.testing.assert_array_equal(a["Ks"] == 0, b["Ks"] == 0)
    assert_close(a["ks"], b["ks"], 5e-5, "ks")
    assert_close(a["Ks"], b["Ks"], 5e-5, "Ks")
    p, lo, hi, u0, x0 = problem(B, T, 8, 2, 0.05, seed=21)
    Ksr, ksr, bo, Ifree = ompc.mpc_backward_rec(a["C"], a["c"], p["F"], p["f"], u0, lo, hi, T, 8, 2, batch_coupled=False)
    assert_close(a["ks"], ksr, TOL, "ks vs oracle")
    assert_close(a["Ks"], Ksr, TOL, "Ks vs oracle")
    assert np.all(a["Ks"][Ifree == 0] == 0) and (Ifree == 0).mean() > 0.2
