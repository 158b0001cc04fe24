"""GEN_TIMING builds only: per-wave s_memtime stamps of the generated LQR stream (prologue end, backward end, end)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
B, T, nx, nu = 4096, 50, 8, 2
if len(sys.argv) > 4:
    B, T, nx, nu = [int(v) for v in sys.argv[1:5]]
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
info = torch.zeros(B, dtype=torch.int32, device="cuda")
for it in range(5):
    info.zero_()
    solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, info=info)
    torch.cuda.synchronize()
a = info.cpu().numpy().astype(np.int64).reshape(-1, 4)[: (B // 4)]
names = ["prologue", "backward", "forward"]
prev = np.zeros(len(a))
print("waves %d; set-up   cycles: mean %8.0f  min %8d  max %8d   (kernel entry -> first instruction of the stream)" % (
    len(a), a[:, 0].mean(), a[:, 0].min(), a[:, 0].max()))
for i, n in enumerate(names):
    seg = a[:, i + 1] - prev
    print("%-9s cycles: mean %8.0f  min %8d  max %8d   (per step %.1f)" % (n, seg.mean(), seg.min(), seg.max(), seg.mean() / T))
    prev = a[:, i + 1]
print("total     cycles: mean %8.0f  max %d  (+ set-up: mean %.0f)" % (a[:, 3].mean(), a[:, 3].max(), (a[:, 3] + a[:, 0]).mean()))
