"""GEN_TIMING builds only: per-wave s_memtime stamps of the AFFINE stream (dmpc_lqr_saved_solve) - set-up, prologue,
backward sweep, rollout.  (scripts/asm_variants.sh "timing:GEN_TIMING=1" with SCRIPT=scripts/affine_phase_times.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import saved_solve_device, solve_saving_device
B, T, nx, nu = 4096, 50, 8, 2
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
got = solve_saving_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], T, nx, nu)
Ks, Quu, Qxu = got[2], got[4], got[5]
info = torch.zeros(B, dtype=torch.int32, device="cuda")
for it in range(5):
    info.zero_()
    saved_solve_device(d["c"], d["F"], Ks, Quu, Qxu, d["x_init"], T, nx, nu, info=info)
    torch.cuda.synchronize()
a = info.cpu().numpy().astype(np.int64).reshape(-1, 4)[: (B // 4)]
prev = np.zeros(len(a))
print("waves %d; set-up   cycles: mean %8.0f  min %8d  max %8d" % (len(a), a[:, 0].mean(), a[:, 0].min(), a[:, 0].max()))
for i, n in enumerate(["prologue", "backward", "forward"]):
    seg = a[:, i + 1] - prev
    print("%-9s cycles: mean %8.0f  min %8d  max %8d   (per step %.1f)" % (n, seg.mean(), seg.min(), seg.max(), seg.mean() / T))
    prev = a[:, i + 1]
print("total     cycles: mean %8.0f  max %d  (+ set-up: mean %.0f)" % (a[:, 3].mean(), a[:, 3].max(), (a[:, 3] + a[:, 0]).mean()))
