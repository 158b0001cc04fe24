"""CPU emulation of the wide-layout backward step from the generator's own FMA lists (index logic check)."""
import os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "chainer_differentiable_mpc_amd", "csrc"))
import numpy as np
import gen_dpp_blocks_wide as gw
from chainer_differentiable_mpc_amd import synthetic
from oracle import lqr as olqr

def run(nx, nu, T=5, seed=1):
    ns = nx + nu
    p = synthetic.make_lqr_problem(1, T, nx, nu, seed=seed, with_f=True)
    Ksr, ksr = olqr.lqr_backward(p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    vf, ftw, rk, vupd = gw.blocks(nx, nu)
    def col(M, c):   # augmented matrix [M | aff] column -> (block, lane)
        return c // 16, c % 16
    def to_regs(M, aff):     # M: rows x ns, aff: rows -> regs [rows][2][16]
        R = np.zeros((M.shape[0], 2, 16))
        for i in range(M.shape[0]):
            for c in range(ns):
                R[i][c // 16][c % 16] = M[i][c]
            R[i][ns // 16][ns % 16] = aff[i]
        return R
    def apply(fmas, env):
        for (a, x, y, ln) in fmas:
            an, ai, ah = re.match(r"(\w+)\[(\d+)\]\[(\d+)\]", a).groups()
            xn, xi, xh = re.match(r"(\w+)\[(\d+)\]\[(\d+)\]", x).groups()
            yn, yi, yh = re.match(r"(\w+)\[(\d+)\]\[(\d+)\]", y).groups()
            env[an][int(ai)][int(ah)] += env[xn][int(xi)][int(xh)][ln] * env[yn][int(yi)][int(yh)]
    V = np.zeros((nx, 2, 16))
    for t in range(T - 1, -1, -1):
        Q = to_regs(p["C"][t, 0], p["c"][t, 0])
        if t < T - 1:
            Fc = to_regs(p["F"][t, 0], p["f"][t, 0])
            W = np.zeros((nx, 2, 16))
            ab, al = ns // 16, ns % 16
            for i in range(nx):
                W[i][ab][al] = V[i][ab][al]
            env = dict(W=W, V=V, Fc=Fc, Q=Q)
            apply(vf, env); apply(ftw, env)
        Qu = Q[nx:].copy(); Kr = Q[nx:].copy()
        for k in range(nu):
            kb, kl = (nx + k) // 16, (nx + k) % 16
            pv = Kr[k][kb][kl]
            Kr[k] = Kr[k] / pv
            for i in range(nu):
                if i != k:
                    Kr[i] = Kr[i] - Kr[i][kb][kl] * Kr[k]
        Kt = -Kr
        K = np.array([[Kt[m][j // 16][j % 16] for j in range(nx)] for m in range(nu)])
        kk = np.array([Kt[m][ns // 16][ns % 16] for m in range(nu)])
        print(nx, nu, "t", t, "K err %.2e k err %.2e" % (np.abs(K - Ksr[t][0]).max(), np.abs(kk - ksr[t][0]).max()))
        if t > 0:
            R = Qu.copy()
            env = dict(R=R, Qu=Qu, Kt=Kt)
            apply(rk, env)
            V = Q[:nx].copy()
            env = dict(V=V, Q=Q, Kt=Kt, R=R)
            apply(vupd, env)
for s in ((12, 4), (16, 4), (16, 8)):
    run(*s)
