"""What "the slowest of a wavefront's four trajectories sets the pace" costs at config-3 size (MPCstep.forward, B = 4096, T = 50,
(8,2), bounds +-0.5): the same launch on a batch whose wavefronts each hold FOUR COPIES of one trajectory - every trajectory then
runs exactly its own number of QP passes and line-search passes, which is what per-trajectory pass counts inside a wavefront
(verdict r03, r04) could reach at best - against the ordinary batch of 4,096 different trajectories.
    python scripts/mpc_step_quad_copies.py     (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from chainer_differentiable_mpc_amd import LinDx, _lib  # noqa: E402
from chainer_differentiable_mpc_amd.util import get_traj  # noqa: E402

device = torch.device("cuda")
B, T, nx, nu = 4096, 50, 8, 2
lib = _lib.load()
P = _lib.ptr


def run(label, quad):
    Bd = B // 4 if quad else B
    p, d = bench.make_inputs(Bd, T, nx, nu, 0, device)
    torch.manual_seed(0)
    un = (0.5 * torch.randn((T, Bd, nu), device=device)).clamp(-0.5, 0.5)
    xn = get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
    if quad:   # trajectory b of the batch = trajectory b // 4 of the distinct ones: a wavefront (4 consecutive) holds one problem
        rep = lambda a, dim: a.repeat_interleave(4, dim=dim).contiguous()  # noqa: E731
        d = {k: rep(v, 0 if k == "x_init" else 1) for k, v in d.items()}
        un, xn = rep(un, 1), rep(xn, 1)
    lo, hi = torch.full((T, B, nu), -0.5, device=device), torch.full((T, B, nu), 0.5, device=device)
    f32 = dict(dtype=torch.float32, device=device)
    Ks, ks = torch.empty((T, B, nu, nx), **f32), torch.empty((T, B, nu), **f32)
    xo, uo, u1 = torch.empty((T, B, nx), **f32), torch.empty((T, B, nu), **f32), torch.empty((T, B, nu), **f32)
    costs, old, al = torch.empty((B,), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
    objs = torch.empty((T, B), **f32)
    nqp, nls = torch.empty((B,), dtype=torch.int32, device=device), torch.empty((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
    ws = torch.empty(need, dtype=torch.uint8, device=device)

    def mpc_fwd():
        rc = lib.dmpc_mpc_step_forward(T, B, nx, nu, P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), P(un), P(xn), P(lo), P(hi),
                                       P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), 1, 0.2, 5, 20, 0, P(xo), P(uo), P(Ks),
                                       P(ks), P(costs), P(old), P(al), P(objs), P(u1), P(nqp), P(nls), P(ws), need,
                                       P(info), _lib.stream_ptr(device))
        assert rc == 0, rc

    ts = sorted(bench.event_time(mpc_fwd, 30) * 1e6 for _ in range(5))
    torch.cuda.synchronize()
    # passes a wavefront runs: the maximum over its four trajectories of the per-trajectory totals is a lower bound of the sum over
    # timesteps of the per-timestep maxima (the counts are per trajectory, summed over t)
    q = nqp.view(-1, 4).float()
    l = nls.view(-1, 4).float()
    print("%-44s %.1f us (median of 5 x 30 launches; min %.1f)  QP passes per timestep: mean %.3f, mean over wavefronts of the "
          "largest of four totals %.3f;  line-search passes: mean %.3f, largest of four %.3f" % (
              label, ts[2], ts[0], float(q.mean()) / T, float(q.max(dim=1).values.mean()) / T, float(l.mean()),
              float(l.max(dim=1).values.mean())), flush=True)


run("4,096 different trajectories", False)
run("1,024 trajectories, four copies per wavefront", True)
run("4,096 different trajectories", False)
run("1,024 trajectories, four copies per wavefront", True)
