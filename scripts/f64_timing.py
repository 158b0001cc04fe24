#!/usr/bin/env python
"""Time the float64 solve / gradient at the headline size (and (32,8)) on whichever family the library picks."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chainer_differentiable_mpc_amd import synthetic, _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device_f64, solve_device
from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device_f64

def ev(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

CASES = ((4096, 50, 8, 2), (4096, 50, 3, 1), (4096, 50, 4, 4), (4096, 50, 12, 3), (2048, 50, 16, 8), (2048, 50, 32, 8))
if len(sys.argv) > 1 and sys.argv[1] == "headline":       # (the PMC passes: the headline size alone)
    CASES = CASES[:1]
for (B, T, nx, nu) in CASES:
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=0)
    d = {k: torch.as_tensor(v, dtype=torch.float64).cuda() for k, v in p.items()}
    for _ in range(200): solve_device_f64(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)   # clocks up
    t = ev(lambda: solve_device_f64(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu))
    name = _lib.last_kernel_name()
    x, u, _, _ = solve_device_f64(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
    gx, gu = torch.ones_like(x), torch.ones_like(u)
    tg = ev(lambda: kkt_grad_device_f64(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu), reps=10, warm=3)
    bts = 2 * synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
    print("f64 solve B=%d T=%d (%d,%d): %.1f us = %.3f of the HBM roof (%d B per timestep-solve); gradient %.1f us; path %d; %s" % (
        B, T, nx, nu, t, bts * B * T / (t * 1e-6) / 8e12, bts, tg, _lib.load().dmpc_lqr_f64_path(nx, nu), name), flush=True)
