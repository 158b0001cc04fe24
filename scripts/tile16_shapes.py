"""The exact instances of the wavefront-per-trajectory solve - (32,8), (24,8), (16,8), (32,4), (24,4) - on the 16x16x4 tile kernel
(lqr_tile16.hpp): parity against the numpy oracle (small batch) and time against the 4x4x1 kernel (DMPC_NO_TILE16=1, a child
process).  Usage: tile16_shapes.py [B]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import chainer_differentiable_mpc_amd as dm
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 50
SHAPES = [(32, 8), (24, 8), (16, 8), (32, 4), (24, 4)]
child = os.environ.get("TILE16_CHILD") == "1"
if not child:
    from oracle import lqr as olqr
for nx, nu in SHAPES:
    if not child:
        p = dm.synthetic.make_lqr_problem(6, 9, nx, nu, seed=3)
        d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
        x, u = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, 9, nx, nu)[:2]
        xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], 9, nx, nu)
        ex = np.max(np.abs(x.cpu().numpy() - xr) / np.maximum(1, np.abs(xr)))
        eu = np.max(np.abs(u.cpu().numpy() - ur) / np.maximum(1, np.abs(ur)))
        assert ex <= 1e-4 and eu <= 1e-4, (nx, nu, ex, eu)
    p = dm.synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in p.items()}
    xo = torch.empty((T, B, nx), device="cuda"); uo = torch.empty((T, B, nu), device="cuda")
    for _ in range(3):
        solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(xo, uo))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(xo, uo))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    ns = nx + nu
    bts = 4 * (ns * ns + ns + nx * ns + 2 * nx + nu)
    print("%s (%d,%d) B=%d T=%d fused solve %.1f us (%.2f of 8 TB/s)%s" % ("4x4x1  " if child else "16x16x4", nx, nu, B, T, us, bts * B * T / (us * 1e-6) / 8e12,
          "" if child else "   parity vs oracle x %.1e u %.1e" % (ex, eu)), flush=True)
if not child:
    subprocess.run([sys.executable, os.path.abspath(__file__), str(B)], env=dict(os.environ, TILE16_CHILD="1", DMPC_NO_TILE16="1"), check=True)
