"""MPCstep.forward at config-3 size ((8,2), B=4096, T=50, need_expand) a few times - for `rocprofv3 --kernel-trace --stats`
and for the generator's timing knobs (GEN_FWD_NO_* builds of mpc_fwd_asm_gen.hpp give wrong results on purpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import LinDx, MPCstep, QuadCost, synthetic, util
dev = torch.device("cuda")
B, T, nx, nu, bound = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (4096, 50, 8, 2) + (0,)
bound = 0.5 if (nx, nu) == (8, 2) else 2.0
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])
u_nom = torch.zeros((T, B, nu), device=dev)
x_nom = util.get_traj(T, u_nom, x0, LinDx(F, f))
hi = bound * torch.ones((T, B, nu), device=dev); lo = -hi
for rep in range(12):
    step = MPCstep(u_nom, T, hi, lo, B, nx, nu, x_nom, QuadCost(C, c), LinDx(F, f), ls_decay=0.2, max_ls_iter=5, need_expand=True)
    x, u = step.forward((x0, C, c, F, f))
torch.cuda.synchronize()
print("done; mean passes %.2f" % float(step.n_ls_iter.float().mean()))
