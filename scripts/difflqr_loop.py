"""DiffLqr forward + backward at config-3 size ((8,2), B=4096, T=50) in a loop - for `rocprofv3 --kernel-trace --stats`.
argv[1] = 0: the full second solve (DiffLqr(save_gains=False)); default: the saving solve + the re-solve from saved gains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import DiffLqr, synthetic
dev = torch.device("cuda")
save = not (len(sys.argv) > 1 and sys.argv[1] == "0")
B, T, nx, nu = 4096, 50, 8, 2
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])
gx, gu = torch.ones((T, B, nx), device=dev), torch.ones((T, B, nu), device=dev)
node = DiffLqr(T, B, nx, nu, save_gains=save)
for rep in range(60):
    node.forward((x0, C, c, F, f))
    out = node.backward((0, 1, 2, 3, 4), (gx, gu))
torch.cuda.synchronize()
print("done, saved gains:", node._retained["saved"] is not None)
