"""debug: the float64 tile kernel's first value function against numpy (library built with -DDMPC_T16F64_DEBUG)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device_f64
B, T, nx, nu = 4, 40, 32, 8
ns = nx + nu
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=5, with_f=True)
d = {k: torch.as_tensor(np.asarray(v, dtype=np.float64)).cuda() for k, v in p.items()}
x, u, Ks, ks = solve_device_f64(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, want_gains=True)
dbg = Ks.cpu().numpy().reshape(T, -1)[30:].reshape(-1)[:nx * nx + nx]     # trajectory 0 (b = 0: t = 30)
Vd, vd = dbg[:nx * nx].reshape(nx, nx), dbg[nx * nx:]
C, c = p["C"][T - 1, 0], p["c"][T - 1, 0]
Qxx, Qxu, Qux, Quu = C[:nx, :nx], C[:nx, nx:], C[nx:, :nx], C[nx:, nx:]
K = -np.linalg.solve(Quu, Qux); k = -np.linalg.solve(Quu, c[nx:])
V = Qxx + Qxu @ K + K.T @ Qux + K.T @ Quu @ K
v = c[:nx] + Qxu @ k + K.T @ c[nx:] + K.T @ Quu @ k
eV = np.abs(Vd - V)
print("V err max", eV.max(), "v err max", np.abs(vd - v).max())
bad = np.argwhere(eV > 1e-9)
print("bad entries", len(bad), "rows", sorted(set(bad[:, 0]))[:40], "cols", sorted(set(bad[:, 1]))[:40])

V2 = Qxx + Qxu @ K
print("err against Qxx + Qxu K alone", np.abs(Vd - V2).max())
D = Vd - Qxx
P = Qxu @ K
print("|Vd - Qxx| max", np.abs(D).max(), "|Qxu K| max", np.abs(P).max())
for name, cand in (("Qxu K", P), ("(Qxu K)^T", P.T), ("zero", 0 * P)):
    print("  D vs", name, np.abs(D - cand).max())
# is D = Qxu' K' for permuted contraction index?  try m -> perm(m)
import itertools
best = None
for perm_name, perm in (("4r+g <-> 4g+r on m (nu=8: m=4a+b -> ?)", None),):
    pass
# least squares: D = Qxu @ M @ K for an 8 x 8 matrix M ?
M = np.linalg.pinv(Qxu) @ D @ np.linalg.pinv(K)
print("D = Qxu M K with M =\n", np.round(M, 3))
