"""What the in-stream box QP costs in MPCstep.forward at config-3 size: the step timed with the QP's iteration cap at
1, 2, 3 and 20 (results with a cap below convergence are not the reference's - timing only), and the mean number of
QP passes per timestep at the full cap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import LinDx, _lib, synthetic, util
dev = torch.device("cuda")
B, T, nx, nu = 4096, 50, 8, 2
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])
torch.manual_seed(0)
un = (0.5 * torch.randn((T, B, nu), device=dev)).clamp(-0.5, 0.5)
xn = util.get_traj(T, un, x0, LinDx(F, f))
lo, hi = torch.full((T, B, nu), -0.5, device=dev), torch.full((T, B, nu), 0.5, device=dev)
lib = _lib.load(); P = _lib.ptr
f32 = dict(dtype=torch.float32, device=dev)
Ks, ks = torch.empty((T, B, nu, nx), **f32), torch.empty((T, B, nu), **f32)
xo, uo, u1 = torch.empty((T, B, nx), **f32), torch.empty((T, B, nu), **f32), torch.empty((T, B, nu), **f32)
costs, old, al = torch.empty((B,), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
objs = torch.empty((T, B), **f32)
nqp, nls = torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)
info = torch.zeros((B,), dtype=torch.int32, device=dev)
need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
for cap in ([int(v) for v in sys.argv[1:]] or (1, 2, 3, 20)):
    def run():
        rc = lib.dmpc_mpc_step_forward(T, B, nx, nu, P(C), P(c), P(F), P(f), P(un), P(xn), P(lo), P(hi), P(C), P(c), P(F), P(f),
                                       1, 0.2, 5, cap, 0, P(xo), P(uo), P(Ks), P(ks), P(costs), P(old), P(al), P(objs), P(u1),
                                       P(nqp), P(nls), P(ws), need, P(info), _lib.stream_ptr(dev))
        assert rc == 0
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record(); torch.cuda.synchronize()
    print("QP cap %2d: step %.1f us, QP passes per timestep %.2f, line-search passes %.2f" % (
        cap, e0.elapsed_time(e1) / 30 * 1e3, float(nqp.float().mean()) / T, float(nls.float().mean())))
