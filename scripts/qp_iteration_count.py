import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from chainer_differentiable_mpc_amd import MPCstep, QuadCost, LinDx, synthetic, util
dev="cuda"
for (B,T,nx,nu,bound) in ((4096,50,8,2,0.5),(128,20,3,1,2.0)):
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])
    u_nom = torch.zeros((T, B, nu), device=dev)
    x_nom = util.get_traj(T, u_nom, x0, LinDx(F, f))
    hi = bound * torch.ones((T, B, nu), device=dev); lo = -hi
    step = MPCstep(u_nom, T, hi, lo, B, nx, nu, x_nom, QuadCost(C, c), LinDx(F, f), ls_decay=0.2, max_ls_iter=5, need_expand=True)
    x, u = step.forward((x0, C, c, F, f))
    n = step.n_qp_iter.cpu().numpy() / T
    print((B,T,nx,nu), "mean (1+it) per step %.2f, max %.2f; ls passes mean %.2f max %d; frac clamped %.3f" % (n.mean(), n.max(), step.n_ls_iter.float().mean().item(), step.n_ls_iter.max().item(), float(((u==hi)|(u==lo)).float().mean())))
