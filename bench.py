#!/usr/bin/env python
"""Headline benchmark: LQR timestep-solves/sec @ batch=4096, T=50, n_x=8, n_u=2 (BASELINE.json).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one fused `solve_recursion` (backward Riccati sweep + forward rollout, x and u
materialised in HBM) over one batch of synthetic trajectories resident in HBM (not in the Infinity
Cache: consecutive steps solve different input sets).  With N
GPUs every rank solves its own shard of B trajectories (independent units, no data-path
collective; `--gather` adds the RCCL all-gather of (x*, u*) to the timed region), so scaling is
weak and `value` = N*B*T*K / max-over-ranks time.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline     dominant kernel vs the HBM roof: algorithmic bytes per launch (832 B per
               timestep-solve) / average launch duration measured with HIP events on
               the launch stream over the timed region; peak 8 TB/s (MI355X_MICROARCH.md).  The timed
               loop rotates over input sets (more than twice the 256 MiB Infinity Cache in total), so
               every step streams its inputs from HBM; `frac_same_inputs` is the launch re-solving one set
               (what rounds 1-2 reported), `box_copy_gbs` a plain copy on the same box, `kernel` the name
               the library reports for the last launch (dmpc_last_kernel_name).
  cpu_baseline the numpy float64 oracle (a port of the reference's algorithm; the reference's
               Python cannot travel) timed on this host's cores on a bounded sample: one thread, and
               (`cpu_baseline_mp`) the batch sharded over the host cores this process may use.
  secondary    (N = 1, headline workload) the other numbers of SURVEY.md 8d, each timed with HIP events in this
               run: DiffLqr forward + KKT backward at config 3, one shard of config 5 ((32,8), B=8192),
               the MPC step at config 3, config 2's box-DDP solve, config 4's imitation step.

Other invocations (the same JSON contract):
    python bench.py --workload cfg5-shard                    # one 8192-trajectory shard of config 5 on one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --workload cfg5-shard --gather     # config 5 (65536 trajectories) + the all-gather of (x*, u*)
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from chainer_differentiable_mpc_amd import synthetic  # noqa: E402
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md)
PARITY_TOL = 1e-4
INFINITY_CACHE_BYTES = 256 * 2 ** 20

WORKLOADS = {
    # name: (B per GPU, T, nx, nu)
    "headline": (4096, 50, 8, 2),        # BASELINE.json configs[2] - the configuration the metric is quoted on
    "cfg5-shard": (8192, 50, 32, 8),     # BASELINE.json configs[4]: 65536 trajectories sharded over 8 GPUs (see STRONG below)
    "pendulum": (1024, 20, 3, 1),        # shapes of configs[1]/[3] (pure LQR part)
}
# workloads whose TOTAL batch is fixed and split over the ranks (strong scaling): config 5 is 65,536 trajectories whatever N is -
# `--workload cfg5-shard --gpus N` solves 65536 / N per GPU (8 GPUs: the 8,192-trajectory shard the name says; 1 GPU: all of it)
STRONG = {"cfg5-shard": 65536}


def init_distributed(world, rank, local_rank, environ=None):
    """one process per GPU: (dist module or None, device, backend, device of the timing reductions).  Backend "nccl" is RCCL
    on ROCm, bound to this rank's device at init (device_id).  Rehearsal knobs (tests/test_dist_gpu.py runs this file with two
    ranks on a ONE-GPU box): DMPC_BENCH_BACKEND=gloo takes the collectives through the host (RCCL refuses two ranks on one
    device), DMPC_BENCH_DEVICE pins every rank's device."""
    environ = os.environ if environ is None else environ
    backend = environ.get("DMPC_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", int(environ.get("DMPC_BENCH_DEVICE", local_rank)))
    torch.cuda.set_device(device)
    red_dev = device if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, device, backend, red_dev


def make_inputs(B, T, nx, nu, seed, device):
    if B * T * (nx + nu) ** 2 <= 64 * 1024 * 1024:
        p = synthetic.make_lqr_problem(B, T, nx, nu, seed=seed)
        dev = {k: torch.as_tensor(v, dtype=torch.float32, device=device) for k, v in p.items()}
        return p, dev
    # large shards are drawn on the device (same distributions, torch generator) - SURVEY.md 8(d)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ns = nx + nu
    L = torch.randn((T, B, ns, ns), generator=g, device=device)
    C = (L @ L.transpose(2, 3) + ns * torch.eye(ns, device=device)) / ns
    del L
    c = torch.randn((T, B, ns), generator=g, device=device)
    A = torch.eye(nx, device=device) + (0.2 / nx ** 0.5) * torch.randn((T - 1, B, nx, nx), generator=g, device=device)
    Bm = torch.randn((T - 1, B, nx, nu), generator=g, device=device)
    F = torch.cat((A, Bm), dim=3).contiguous()
    del A, Bm
    f = 0.1 * torch.randn((T - 1, B, nx), generator=g, device=device)
    x_init = torch.randn((B, nx), generator=g, device=device)
    return None, dict(C=C, c=c, F=F, f=f, x_init=x_init)


def kernel_name():
    """the kernel the last library call of this thread launched, as rocprofv3 --kernel-trace lists it: asked of the
    library (dmpc_last_kernel_name -> hipKernelNameRefByPtr + demangling), not kept by hand"""
    from chainer_differentiable_mpc_amd import _lib
    return _lib.last_kernel_name()


def cpu_baseline(p, T, nx, nu, budget_s=12.0):
    """the oracle (kind "port") on this host, float64, one thread; bounded sample"""
    from oracle import lqr as olqr
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    B = p["C"].shape[1]
    times = []
    with ctx:
        t_all = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
            times.append(time.perf_counter() - t0)
            if len(times) >= 3 and time.perf_counter() - t_all > budget_s:
                break
            if len(times) >= 25:
                break
    med = statistics.median(times)
    return dict(value=B * T / med, unit="timestep-solves/s", cores=1, kind="port",
                sample="numpy float64 oracle (oracle/lqr.py, restates lqr/lqr_recursion.py), full workload "
                       "B=%d T=%d, median of %d runs, %d host cores present, BLAS limited to 1 thread"
                       % (B, T, len(times), os.cpu_count() or 0)), xr, ur


_MP_PROBLEM = None


def _mp_solve_shard(job):
    """worker: the oracle on one contiguous batch shard of the fork-inherited problem"""
    from oracle import lqr as olqr
    b0, b1, T, nx, nu = job
    p = _MP_PROBLEM
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        x, u = olqr.lqr_solve(p["x_init"][b0:b1], p["C"][:, b0:b1], p["c"][:, b0:b1], p["F"][:, b0:b1],
                              p["f"][:, b0:b1], T, nx, nu)
    return float(x.sum() + u.sum())


def cpu_baseline_multiprocess(p, T, nx, nu, procs, reps=3):
    """the same oracle with the batch sharded over `procs` worker processes (fork; started BEFORE this process touches
    the GPU).  Wall time of one whole-batch solve = the slowest shard; median of `reps`."""
    import multiprocessing as mp
    from chainer_differentiable_mpc_amd.dist import shard_bounds
    global _MP_PROBLEM
    _MP_PROBLEM = p
    B = p["C"].shape[1]
    jobs = [shard_bounds(B, r, procs) + (T, nx, nu) for r in range(procs)]
    times = []
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_mp_solve_shard, jobs)          # warm-up: page in numpy in every worker
        for _ in range(reps):
            t0 = time.perf_counter()
            pool.map(_mp_solve_shard, jobs, chunksize=1)
            times.append(time.perf_counter() - t0)
    _MP_PROBLEM = None
    med = statistics.median(times)
    return dict(value=B * T / med, unit="timestep-solves/s", cores=procs, kind="port",
                sample="numpy float64 oracle, full workload B=%d T=%d sharded over %d worker processes (one BLAS "
                       "thread each), median of %d runs; %d host cores present, %d usable by this process"
                       % (B, T, procs, reps, os.cpu_count() or 0, usable_cores()))


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:  # pragma: no cover
        return os.cpu_count() or 1


SETTLE_MS = 60.0       # how long the device is kept busy before a measurement (scripts/clock_ramp.py: from a cold start a
                       # launch takes 36 us in the first millisecond, 40-45 us from the 4th to the 10th, and its
                       # steady 33 us from ~30 ms on, for as many seconds as the load lasts)


class ClockSampler:
    """sclk / mclk / board power of one GPU read from sysfs (hwmon freq1_input, freq2_input, power1_input, power1_cap) by a
    thread every ~2 ms while the timed blocks run: the power state the number was measured in travels with it (boxes of the
    pool differ by 10 % on the same binary, MI355X_MICROARCH.md DVFS items 5-6).  Everything is optional: a box that does not show
    the files reports null."""

    def __init__(self, index=0):
        import glob
        cards = sorted(d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        self.dir = None
        if cards:
            hw = glob.glob(os.path.join(cards[min(index, len(cards) - 1)], "hwmon", "hwmon*"))
            self.dir = hw[0] if hw else None
        self.samples = []
        self._stop = False
        self._thread = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            self.samples.append((self._read("freq1_input"), self._read("freq2_input"), self._read("power1_input")))
            time.sleep(0.002)

    def __enter__(self):
        if self.dir is not None:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        return False

    def summary(self):
        if self.dir is None or not self.samples:
            return None

        def stat(i, scale):
            v = sorted(x[i] * scale for x in self.samples if x[i] is not None)
            return None if not v else {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        cap = self._read("power1_cap")
        return {"samples": len(self.samples), "sclk_mhz": stat(0, 1e-6), "mclk_mhz": stat(1, 1e-6), "power_w": stat(2, 1e-6),
                "power_cap_w": None if cap is None else cap * 1e-6,
                "what": "hwmon freq1_input / freq2_input / power1_input of this rank's GPU, sampled every ~2 ms by a host thread "
                        "(`clocks`: during the timed blocks - ~1 ms of load each between synchronisations, which the sensors' "
                        "refresh of a few milliseconds mostly misses; `clocks_run_up`: during the >= 60 ms run-up of the same "
                        "launch back to back that ends right before them).  sysfs reads up to ~10 % above the in-kernel clock "
                        "(MI355X_MICROARCH.md, DVFS give-back item 6)"}


def settle(fn, min_ms=SETTLE_MS, max_ms=600.0, block=25, fixed_blocks=None):
    """Bring the GPU to the power state of a running job before timing anything: fn() back to back in blocks of `block`
    for at least `min_ms`, until two consecutive blocks take the same time within 3 % (at most `max_ms`).  A freshly
    woken MI355X ramps its clocks over tens of milliseconds (and dips on the way); a measurement window that starts one
    millisecond after the first launch times that transient, not the kernel.  `fixed_blocks`: that many blocks, no
    stopping rule - for an fn() with a collective inside, where every rank must make the same number of calls.
    Returns (calls made, milliseconds spent, [first block's, last block's] seconds per call)."""
    fn()                               # (first-call costs - lazy initialisation, workspace allocation - are not the ramp)
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    calls, prev, first, last = 1, None, None, None
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(block):
            fn()
        e1.record()
        torch.cuda.synchronize()
        calls += block
        last = e0.elapsed_time(e1) * 1e-3 / block
        first = last if first is None else first
        spent = (time.perf_counter() - t_begin) * 1e3
        if fixed_blocks is not None:
            if calls >= 1 + fixed_blocks * block:
                return calls, spent, [first, last]
        elif spent >= max_ms or (spent >= min_ms and prev is not None and abs(last - prev) <= 0.03 * prev):
            return calls, spent, [first, last]
        prev = last


def event_time(fn, reps, warm=3, settled=True):
    """average duration of fn() in seconds: HIP events on the current stream around `reps` back-to-back calls, the
    device in its steady power state (`settle`)"""
    if settled:
        settle(fn, min_ms=SETTLE_MS / 2)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def hbm_copy_calibration(device, gib=1.0, reps=5):
    """what THIS box's memory system sustains on the plainest streaming pattern: a device-to-device copy of `gib` GiB
    (read + write, far beyond the Infinity Cache), GB/s of traffic, median of `reps` - MI355X_MICROARCH.md quotes
    6.29 TB/s for it; boxes of this pool differ (clocks under load), and the HBM-streamed numbers move with it"""
    n = int(gib * 2 ** 30) // 4
    a = torch.empty(n, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    ts = [event_time(lambda: b.copy_(a), 4, warm=1) for _ in range(reps)]
    del a, b
    torch.cuda.empty_cache()
    return 2 * n * 4 / statistics.median(ts) / 1e9


def secondary_shapes(device):
    """The fused solve and the KKT gradient of shapes beyond the headline's kernel at B=4096, T=50 - one line per kernel
    family: 16-lane HIP kernel, wide row kernel (exact / padded), wavefront-per-trajectory container with the staged rollout.
    Fractions are of the HBM roof on the path's algorithmic bytes (solve: 4(ns^2 + ns + nx ns + nx + ns) per timestep-solve)."""
    from chainer_differentiable_mpc_amd import _lib
    from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device
    out = {"what": "fused solve / DiffLqr gradient (second solve + co-state sweep) per shape, B=4096 T=50, HIP events over "
                   "20 back-to-back calls at the device's steady clocks (settle)"}
    B, T = 4096, 50
    for nx, nu in ((8, 4), (12, 4), (16, 4), (13, 3), (16, 8), (20, 6)):
        try:
            p, d = make_inputs(B, T, nx, nu, 0, device)
            x = torch.empty((T, B, nx), dtype=torch.float32, device=device)
            u = torch.empty((T, B, nu), dtype=torch.float32, device=device)
            gx, gu = torch.ones_like(x), torch.ones_like(u)
            t_solve = event_time(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)),
                                 20, warm=3)
            k_solve = _lib.last_kernel_name()
            t_grad = event_time(lambda: kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu), 20, warm=3)
            k_grad = _lib.last_kernel_name()
            bts = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
            out["%dx%d" % (nx, nu)] = {"us_solve": t_solve * 1e6, "frac_hbm_solve": bts * B * T / t_solve / 1e9 / HBM_PEAK_GBS,
                                      "us_gradient": t_grad * 1e6, "finite": bool(torch.isfinite(x).all()),
                                      "kernel": k_solve, "kernel_gradient_last": k_grad}
            del p, d, x, u, gx, gu
            torch.cuda.empty_cache()
        except Exception as e:  # pragma: no cover
            out["%dx%d" % (nx, nu)] = {"error": repr(e)}
    return out


def secondary_metrics(device, d_headline):
    """SURVEY.md 8d's secondary numbers, measured live (about 10 s in total)"""
    import warnings
    from chainer_differentiable_mpc_amd import BoxDDP, LinDx, MPCstep, PendulumDx, QuadCost
    from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device
    from chainer_differentiable_mpc_amd.pendulum import sample_xinit
    out = {}
    B, T, nx, nu = WORKLOADS["headline"]
    bts = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
    kts = synthetic.kkt_algorithmic_bytes_per_timestep(nx, nu)
    x = torch.empty((T, B, nx), dtype=torch.float32, device=device)
    u = torch.empty((T, B, nu), dtype=torch.float32, device=device)
    # (ii) DiffLqr forward + backward at config 3: fused solve, then the KKT gradient (second solve + co-state sweeps)
    d = d_headline
    gx, gu = torch.ones_like(x), torch.ones_like(u)

    def fwd_bwd_full():     # the second solve repeats the Riccati sweep (sizes without a saving stream; DiffLqr(save_gains=False))
        solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
        kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu)

    from chainer_differentiable_mpc_amd.lqr_recursion import solve_saving_device
    sv = [None]

    def fwd_bwd():          # what DiffLqr runs: the forward solve leaves K, Quu, Qxu, the second solve reuses them
        xs, us, Ks, _, Quu, Qxu, Vv = solve_saving_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], T, nx, nu)
        sv[0] = (xs, us, (Ks, Quu, Qxu, Vv))
        kkt_grad_device(d["C"], d["c"], d["F"], xs, us, gx, gu, T, nx, nu, saved=(Ks, Quu, Qxu, Vv))

    t_full = event_time(fwd_bwd_full, 50)
    tb_full = event_time(lambda: kkt_grad_device(d["C"], d["c"], d["F"], x, u, gx, gu, T, nx, nu), 50)
    t = event_time(fwd_bwd, 50)
    xs, us, saved = sv[0]
    tb = event_time(lambda: kkt_grad_device(d["C"], d["c"], d["F"], xs, us, gx, gu, T, nx, nu, saved=saved), 50)
    out["difflqr_fwd_bwd_cfg3"] = {
        "what": "DiffLqr forward + analytic KKT backward, B=4096 T=50 (8,2); outputs dC, dc, dF, df, dx_init materialised; "
                "the forward solve saves K, Quu, Qxu (+ 152 B per timestep written) and the backward's second solve reuses them",
        "us_fwd_bwd": t * 1e6, "us_bwd": tb * 1e6, "algorithmic_bytes": (bts + kts) * B * T,
        "frac_hbm": (bts + kts) * B * T / t / 1e9 / HBM_PEAK_GBS, "frac_hbm_bwd_only": kts * B * T / tb / 1e9 / HBM_PEAK_GBS,
        "us_fwd_bwd_full_second_solve": t_full * 1e6, "us_bwd_full_second_solve": tb_full * 1e6}
    del sv, xs, us, saved
    # (iii) the MPC step at config 3 (PNQP per timestep + line search), LinDx / QuadCost
    torch.manual_seed(0)
    un = (0.5 * torch.randn((T, B, nu), device=device)).clamp(-0.5, 0.5)
    from chainer_differentiable_mpc_amd.util import get_traj
    xn = get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
    lo, hi = torch.full((T, B, nu), -0.5, device=device), torch.full((T, B, nu), 0.5, device=device)
    step = MPCstep(un, T, hi, lo, B, nx, nu, xn, QuadCost(d["C"], d["c"]), LinDx(d["F"], d["f"]), 0.2, 5, need_expand=True)
    from chainer_differentiable_mpc_amd import _lib
    lib = _lib.load()
    f32 = dict(dtype=torch.float32, device=device)
    Ks, ks = torch.empty((T, B, nu, nx), **f32), torch.empty((T, B, nu), **f32)
    xo, uo, u1 = torch.empty((T, B, nx), **f32), torch.empty((T, B, nu), **f32), torch.empty((T, B, nu), **f32)
    costs, old, al = torch.empty((B,), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
    objs = torch.empty((T, B), **f32)
    nqp, nls = torch.empty((B,), dtype=torch.int32, device=device), torch.empty((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    P = _lib.ptr

    def mpc_fwd():
        rc = lib.dmpc_mpc_step_forward(T, B, nx, nu, P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), P(un), P(xn), P(lo), P(hi),
                                       P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), 1, 0.2, 5, 20, 0, P(xo), P(uo), P(Ks),
                                       P(ks), P(costs), P(old), P(al), P(objs), P(u1), P(nqp), P(nls), P(ws), need,
                                       P(info), _lib.stream_ptr(device))
        assert rc == 0, rc

    t = event_time(mpc_fwd, 30)
    # algorithmic bytes per timestep: the sweep reads C, c, F, u, lower, upper, x (re-centring) and writes K, k; every
    # line-search pass reads C, c, F, f, K, k, u, lower, upper, x and writes x, u (+ objs, u_first on the first pass)
    ns = nx + nu
    passes = float(nls.float().mean())
    sweep_b = 4 * (ns * ns + ns + nx * ns + 3 * nu + nx) + 4 * (nu * nx + nu)
    pass_b = 4 * (ns * ns + ns + nx * ns + nx + nu * nx + 4 * nu + nx) + 4 * ns
    mpc_bytes = B * T * (sweep_b + passes * pass_b + 4 * (1 + nu))
    # read-once: every input array once (C, c, F, f, u, lower, upper, x) + the outputs (K, k, x, u, objs, u_first) - what a
    # step that kept everything on chip between its sweep and its line-search passes would move
    once_b = 4 * (ns * ns + ns + nx * ns + nx + 3 * nu + nx) + 4 * (nu * nx + nu) + 4 * ns + 4 * (1 + nu)
    out["mpc_step_forward_cfg3"] = {"what": "MPCstep.forward (Taylor re-centring, backward_rec with one PNQP per timestep, "
                                            "line search) B=4096 T=50 (8,2), bounds +-0.5", "us": t * 1e6,
                                    "timestep_solves_per_s": B * T / t, "line_search_passes_mean": passes,
                                    "qp_passes_per_timestep_mean": float(nqp.float().mean()) / T,
                                    "algorithmic_bytes": int(mpc_bytes), "frac_hbm": mpc_bytes / t / 1e9 / HBM_PEAK_GBS,
                                    "read_once_bytes": int(B * T * once_b),
                                    "frac_hbm_read_once": B * T * once_b / t / 1e9 / HBM_PEAK_GBS}
    del ws, Ks, ks, xo, uo, u1, objs
    # (iv) config 2: pendulum box-DDP, B=128, T=20, 10 iLQR iterations; (v) config 4: imitation step at B=1024
    dx = PendulumDx()
    for name, Bp in (("config2_box_ddp_b128", 128), ("config4_box_ddp_b1024", 1024)):
        q, pp = dx.get_true_obj()
        Q = torch.diag(q).to(device)[None, None].expand(20, Bp, -1, -1).contiguous()
        pv = pp.to(device)[None, None].expand(20, Bp, -1).contiguous()
        x0 = torch.as_tensor(sample_xinit(Bp, seed=0), dtype=torch.float32, device=device)
        kw_ddp = dict(eps=dx.mpc_eps, max_iter=10, exit_unconverged=False, line_search_decay=dx.linesearch_decay,
                      max_line_search_iter=dx.max_linesearch_iter, quiet=True)
        res = {}
        for key, graph in (("ms_per_solve", True), ("ms_per_solve_launched", False)):
            # graph=True (the default): from its second call on the same buffers on, BoxDDP replays the chain of launches it
            # recorded (a hipGraph of its own); graph=False launches the chain every time, as rounds 1-4 did
            solver = BoxDDP(20, dx.lower, dx.upper, Bp, 3, 1, None, graph=graph, **kw_ddp)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                with torch.no_grad():
                    for _ in range(4):
                        solver((x0, QuadCost(Q, pv), dx))
                    torch.cuda.synchronize()
                    times = []          # every solve ends with its own host read-back (no other synchronisation here):
                    for _ in range(7):       # blocks of 5 solves, the median block reported - a mean over a few
                        t0 = time.perf_counter()     # milliseconds is at the mercy of one host hiccup
                        for _ in range(5):
                            solver((x0, QuadCost(Q, pv), dx))
                        times.append((time.perf_counter() - t0) / 5)
                    res[key] = float(np.median(times))
        t = res["ms_per_solve"]
        out[name] = {"what": "BoxDDP (pendulum, true cost, T=20, 10 iLQR iterations incl. the host synchronisation), "
                             "B=%d; median of 7 blocks of 5 solves on the same buffers: BoxDDP replays its chain of 11-12 launches "
                             "from the hipGraph it recorded itself (ms_per_solve_launched: graph=False, the chain launched every "
                             "time)" % Bp, "ms_per_solve": t * 1e3, "ms_per_solve_launched": res["ms_per_solve_launched"] * 1e3,
                     "ilqr_timestep_solves_per_s": Bp * 20 * solver.n_iter / t}
    # (vi) config 4 as the TRAINING update it names (env_dx/il_env.py:104-158, il_exp.py:213-302), timed by
    # scripts/imitation_update_timing.py in a process of its own: one of its three figures replays the update from a
    # hipGraph, and a capture that goes wrong inside the HIP runtime must not take the benchmark line with it
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "imitation_update_timing.py")], capture_output=True,
                           text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if line:
            out["config4_imitation_step"] = json.loads(line[-1])
        else:
            out["config4_imitation_step"] = {"error": "imitation_update_timing.py rc=%d: %s" % (r.returncode, r.stderr[-400:])}
    except subprocess.TimeoutExpired:       # only this entry carries the error: the other secondary numbers stand
        out["config4_imitation_step"] = {"error": "imitation_update_timing.py did not finish within 600 s"}
    return out


def secondary_cpu_baselines(p_headline, budget_s=4.0):
    """the CPU path timed beside every GPU number.  The numpy oracle (kind "port", one thread, float64) on a
    BOUNDED sample of each secondary leg's workload - a slice of the batch that finishes within ~`budget_s` seconds each; every
    entry says what the sample was and reports the leg's own unit (timestep-solves per second), so it scales to the full batch."""
    import warnings
    from oracle import box_ddp as obox
    from oracle import kkt as okkt
    from oracle import lqr as olqr
    from oracle import mpc as ompc
    out = {}
    B, T, nx, nu = WORKLOADS["headline"]

    def timed(fn, units, sample):
        t0 = time.perf_counter()
        n = 0
        while True:
            fn()
            n += 1
            if time.perf_counter() - t0 >= budget_s or n >= 5:
                break
        dt = (time.perf_counter() - t0) / n
        return {"value": units / dt, "unit": "timestep-solves/s", "cores": 1, "kind": "port", "seconds_per_sample": dt, "sample": sample}

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # DiffLqr forward + backward: the first 256 trajectories of the headline batch
        b = 256
        q = {k: (v[:b] if k == "x_init" else v[:, :b]) for k, v in p_headline.items()}
        gx, gu = np.ones((T, b, nx)), np.ones((T, b, nu))

        def difflqr():
            xr, ur = olqr.lqr_solve(q["x_init"], q["C"], q["c"], q["F"], q["f"], T, nx, nu)
            okkt.difflqr_backward(q["x_init"], q["C"], q["c"], q["F"], xr, ur, gx, gu, T, nx, nu)
        out["difflqr_fwd_bwd_cfg3"] = timed(difflqr, b * T, "oracle/lqr.py + oracle/kkt.py, forward solve + KKT gradient, %d of the %d trajectories, T=%d" % (b, B, T))
        # MPCstep.forward: the first 64 trajectories (need_expand, bounds +-0.5, the nominal controls of the GPU leg's recipe)
        b = 64
        q = {k: (v[:b] if k == "x_init" else v[:, :b]) for k, v in p_headline.items()}
        rng = np.random.RandomState(0)
        un = np.clip(0.5 * rng.randn(T, b, nu), -0.5, 0.5)
        lin = ompc.LinDx(q["F"], q["f"])
        xn = obox.get_traj(T, un, q["x_init"], lin)
        lo, hi = np.full((T, b, nu), -0.5), np.full((T, b, nu), 0.5)
        cost = ompc.QuadCost(q["C"], q["c"])
        out["mpc_step_forward_cfg3"] = timed(
            lambda: ompc.mpc_forward(q["C"], q["c"], q["F"], q["f"], un, xn, lo, hi, cost, lin, 0.2, 5, T, nx, nu, need_expand=True,
                                     batch_coupled=False),
            b * T, "oracle/mpc.py mpc_forward (re-centring, one PNQP per timestep, line search), %d of the %d trajectories, T=%d" % (b, B, T))
        # box-DDP on the pendulum: config 2 at its own batch, config 4 on a quarter of its batch; 10 iLQR iterations, T = 20
        from chainer_differentiable_mpc_amd import PendulumDx
        from chainer_differentiable_mpc_amd.pendulum import sample_xinit
        dx = PendulumDx()
        qv, pv = (t.numpy().astype(np.float64) for t in dx.get_true_obj())
        for name, b, full in (("config2_box_ddp_b128", 128, 128), ("config4_box_ddp_b1024", 256, 1024)):
            Q = np.broadcast_to(np.diag(qv), (20, b, 4, 4)).copy()
            pp = np.broadcast_to(pv, (20, b, 4)).copy()
            x0 = np.asarray(sample_xinit(b, seed=0), dtype=np.float64)
            res = {}

            def ddp():
                r = obox.box_ddp(x0, ompc.QuadCost(Q, pp), obox.pendulum_step, 20, dx.lower, dx.upper, 3, 1, eps=dx.mpc_eps,
                                 max_iter=10, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter,
                                 linearize=obox.pendulum_linearize, batch_coupled=False)
                res["n_iter"] = int(r[4])
            e = timed(ddp, 1, "oracle/box_ddp.py, pendulum, true cost, T=20, up to 10 iLQR iterations, %d of the %d trajectories" % (b, full))
            e["value"] = b * 20 * res["n_iter"] / e["seconds_per_sample"]
            e["unit"] = "iLQR timestep-solves/s"
            e["ms_per_solve_of_the_sample"] = e["seconds_per_sample"] * 1e3
            out[name] = e
    return out


def secondary_cfg5(device):
    """one shard of config 5: B=8192, T=50, (32,8) - time per solve, fraction of the HBM roof and of the fp32 MFMA peak"""
    B, T, nx, nu = WORKLOADS["cfg5-shard"]
    _, d = make_inputs(B, T, nx, nu, 5, device)
    x = torch.empty((T, B, nx), dtype=torch.float32, device=device)
    u = torch.empty((T, B, nu), dtype=torch.float32, device=device)
    t = event_time(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)), 5, warm=2)
    bts = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
    flops = 251861        # per timestep-solve at (32,8), SURVEY.md 8d
    return {"what": "one 8192-trajectory shard of config 5 (B=65536 over 8 GPUs), T=50, (32,8), fused solve",
            "ms_per_solve": t * 1e3, "timestep_solves_per_s": B * T / t, "algorithmic_bytes": bts * B * T,
            "frac_hbm": bts * B * T / t / 1e9 / HBM_PEAK_GBS,
            "frac_mfma_f32": flops * B * T / t / 1e12 / MFMA_F32_PEAK_TFLOPS, "kernel": kernel_name()}


def secondary_cfg5_full(device):
    """config 5 at its FULL size on ONE GPU: B=65536, T=50, (32,8) - 39 GB of inputs in this GPU's 288 GB.  The N = 1 point of
    a strong-scaling curve for the configuration BASELINE.json shards over 8 GPUs (8 x `cfg5_shard` on one device)"""
    _, T, nx, nu = WORKLOADS["cfg5-shard"]
    B = 65536
    _, d = make_inputs(B, T, nx, nu, 55, device)
    x = torch.empty((T, B, nx), dtype=torch.float32, device=device)
    u = torch.empty((T, B, nu), dtype=torch.float32, device=device)
    t = event_time(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u)), 3, warm=1)
    bts = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
    ok = bool(torch.isfinite(x).all()) and bool(torch.isfinite(u).all())
    # size-independent property at full size: the returned trajectory satisfies the dynamics it was rolled out with
    res = 0.0
    for b0 in range(0, B, 8192):       # (in slices: one einsum over 39 GB of F is not what this check is about)
        sl = slice(b0, b0 + 8192)
        tau = torch.cat((x[:-1, sl], u[:-1, sl]), dim=2)
        res = max(res, float((torch.einsum("tbij,tbj->tbi", d["F"][:, sl], tau) + d["f"][:, sl] - x[1:, sl]).abs().max()))
    return {"what": "config 5 whole (B=65536, T=50, (32,8)) on one GPU, fused solve: the N=1 point of its strong-scaling curve",
            "ms_per_solve": t * 1e3, "timestep_solves_per_s": B * T / t, "algorithmic_bytes": bts * B * T,
            "input_gb": sum(v.numel() * 4 for v in d.values()) / 1e9,
            "frac_hbm": bts * B * T / t / 1e9 / HBM_PEAK_GBS, "finite": ok, "dynamics_residual_max": res, "kernel": kernel_name()}


def secondary_f64(device, d):
    """the headline workload at the REFERENCE's precision (float64: lqr/differentiable_lqr.py:169-172): dmpc_lqr_solve_f64 on
    the register-resident float64 kernels (f64_row_kernels.hpp), with its own roofline - twice the float32 bytes"""
    from chainer_differentiable_mpc_amd import _lib
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device_f64
    B, T, nx, nu = WORKLOADS["headline"]
    d64 = {k: v.double() for k, v in d.items()}
    fn = lambda: solve_device_f64(d64["C"], d64["c"], d64["F"], d64["f"], d64["x_init"], None, T, nx, nu)   # noqa: E731
    t = event_time(fn, 30)
    name = kernel_name()
    x64, u64, _, _ = fn()
    x32, u32, _, _ = solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu)
    ex = float(((x32.double() - x64).abs() / x64.abs().clamp(min=1.0)).max())
    eu = float(((u32.double() - u64).abs() / u64.abs().clamp(min=1.0)).max())
    bts = 2 * synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
    return {"what": "the headline workload in float64 (B=4096, T=50, (8,2), fused solve), dtype f64", "us": t * 1e6,
            "timestep_solves_per_s": B * T / t,
            "roofline": {"bound": "hbm", "achieved": bts * B * T / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bts * B * T / t / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bts * B * T,
                         "kernel": name},
            "path": int(_lib.load().dmpc_lqr_f64_path(nx, nu)),
            "float32_stream_vs_this": {"max_rel_err_x": ex, "max_rel_err_u": eu, "tolerance": PARITY_TOL}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--gather", action="store_true", help="all-gather (x*, u*) over RCCL inside the timed region, overlapped "
                    "with the next solve (dist.GatherPipeline); the line then carries solve / gather / serial / overlapped times")
    ap.add_argument("--gather-chunks", type=int, default=0, help="pieces per gathered tensor (0: by size, <= 64 MB each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="worker processes of the sharded CPU baseline "
                    "(default: the cores this process may use, at most 16 - a one-GPU box's CPU share)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-settle", action="store_true", help="time from a cold start: no run-up to the steady power state "
                    "before the warm-up steps (the first milliseconds of a job instead of its steady rate)")
    ap.add_argument("--allow-secondary-failure", action="store_true",
                    help="exit 0 even when a secondary measurement raised (its error string is in the JSON line either way)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    B, T, nx, nu = WORKLOADS[args.workload]
    strong_total = STRONG.get(args.workload)
    if strong_total is not None:
        if strong_total % world:
            raise SystemExit("--workload %s splits %d trajectories over the ranks: --gpus %d does not divide it" % (args.workload, strong_total, world))
        B = strong_total // world
    # the CPU legs run first, on rank 0 at N = 1 only: the worker pool is forked before this process touches the GPU
    cb = cb_mp = xr = ur = p_host = cb_sec = None
    want_cpu = not args.no_cpu_baseline and world == 1 and B * T * (nx + nu) ** 2 <= 64 * 1024 * 1024
    if want_cpu:
        p_host = synthetic.make_lqr_problem(B, T, nx, nu, seed=rank)
        procs = args.cpu_procs or max(1, min(16, usable_cores()))
        if procs > 1:
            try:
                cb_mp = cpu_baseline_multiprocess(p_host, T, nx, nu, procs)
            except Exception as e:  # pragma: no cover - e.g. a box that does not allow that many processes
                cb_mp = dict(value=None, unit="timestep-solves/s", cores=procs, kind="port", sample="failed: %r" % (e,))
        cb, xr, ur = cpu_baseline(p_host, T, nx, nu, args.cpu_seconds)
        if args.workload == "headline" and not args.no_secondary:
            try:
                cb_sec = secondary_cpu_baselines(p_host)
            except Exception as e:  # pragma: no cover - the GPU legs stand without them
                cb_sec = {"error": repr(e)}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dist, device, backend, red_dev = init_distributed(world, rank, local_rank)

    p, d = make_inputs(B, T, nx, nu, seed=rank, device=device)
    # "Inputs resident in HBM when the timed region starts" - in HBM, not in the 256 MiB Infinity Cache: where one input
    # set would fit the cache, the timed loop rotates over enough sets (more than twice the cache, at most 8) that every
    # step streams its inputs from HBM.  Set 0 (seed = rank) is the one the CPU oracle solves.
    set_bytes = sum(v.numel() * v.element_size() for v in d.values())
    n_sets = 1 if set_bytes >= 2 * INFINITY_CACHE_BYTES else min(8, -(-2 * INFINITY_CACHE_BYTES // set_bytes))
    sets = [d] + [make_inputs(B, T, nx, nu, seed=1000 + 16 * rank + k, device=device)[1] for k in range(1, n_sets)]
    x = torch.empty((T, B, nx), dtype=torch.float32, device=device)
    u = torch.empty((T, B, nu), dtype=torch.float32, device=device)
    # --gather: the all-gather of (x*, u*) travels on a side stream, in chunks, while the NEXT solve runs
    # (dist.GatherPipeline; SURVEY.md section 5: at config 5 the gather costs about what the solve costs)
    gx = pipe = None
    if args.gather and world > 1:
        from chainer_differentiable_mpc_amd.dist import GatherPipeline
        pipe = GatherPipeline([tuple(x.shape), tuple(u.shape)], device, chunks=args.gather_chunks or None)
        gx = pipe

    k_step = [0]

    def step():
        k = k_step[0]
        e = sets[k % n_sets]
        k_step[0] += 1
        if pipe is None:
            solve_device(e["C"], e["c"], e["F"], e["f"], e["x_init"], None, T, nx, nu, out=(x, u))
        else:
            xs, us = pipe.local_buffers(k)
            solve_device(e["C"], e["c"], e["F"], e["f"], e["x_init"], None, T, nx, nu, out=(xs, us))
            pipe.gather(k)

    # For the record, the contract's protocol from a cold device first (W warm-up steps, K timed ones - a window of a
    # millisecond at the driver's K = 20): reported as `cold_start`, never as `value`
    cold = None
    if not args.no_settle:
        step()
        torch.cuda.synchronize()       # (first-call costs are not part of either figure)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        cold = (time.perf_counter() - c0) / args.steps
    # the device in the power state of a running job (every rank its own GPU), then the contract's W warm-up steps
    # (with --gather the step holds a collective: every rank runs the same, fixed number of blocks)
    # (the run-up is >= 60 ms of the same launch back to back: the sensors, which the SMU refreshes every few milliseconds, have
    # time to show the state the timed blocks then run in - the blocks themselves are ~1 ms each)
    with ClockSampler(device.index or 0) as run_up_clocks:
        pre_calls, pre_ms, pre_times = (0, 0.0, [None, None]) if args.no_settle else settle(
            step, fixed_blocks=(80 if B * T * (nx + nu) ** 2 < 1e8 else 4) if gx is not None else None)
    run_up_clock_info = None if args.no_settle else run_up_clocks.summary()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region, kBlocks times: EXACTLY K steps bracketed by a barrier + torch.cuda.synchronize() on both sides, the MAX
    # over ranks of each block's time; `value` is the MEDIAN block (one block of K = 20 is 0.7 ms of a 20 s run: one number from
    # it says little, the spread of eleven says whether a box is steady), min and max travel in `roofline.blocks`.  Kernel
    # duration: ONE pair of HIP events on the launch stream (torch's current stream) around the K back-to-back launches of each
    # block; span / K = average launch duration incl. boundaries.
    kBlocks = 11
    block_wall, block_ev = [], []
    with ClockSampler(device.index or 0) as clocks:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(kBlocks)]
        for blk in range(kBlocks):
            # every block is the contract's whole protocol: W untimed warm-up steps, barrier + synchronize, K timed steps,
            # synchronize + barrier (the device goes from its warm-up straight into the timed steps, as in a running job)
            for _ in range(args.warmup if blk > 0 else 0):       # (block 0's warm-up ran above)
                step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = evs[blk]
            t0 = time.perf_counter()
            ev0.record()
            for _ in range(args.steps):
                step()
            ev1.record()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if dist is not None:
                tmax = torch.tensor([el], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            block_wall.append(el)
            block_ev.append(ev0.elapsed_time(ev1) * 1e-3 / args.steps)
    order = sorted(range(kBlocks), key=lambda i: block_wall[i])
    mid = order[kBlocks // 2]
    elapsed = block_wall[mid]
    kern_s = block_ev[mid]
    clock_info = clocks.summary()
    kern_name = kernel_name() if gx is None else None      # (with --gather the last launch is RCCL's, not ours)
    gather_info = None
    if pipe is not None:
        # the three times the overlapped figure is made of, every rank making the same number of calls (collectives inside):
        # the solve alone, the gather alone (side stream, waited for), and the two one after the other on one stream
        e0 = sets[0]
        pipe.reset()      # (the timed region left its own numbering in the two buffer sets)

        def timed(fn, reps=10):
            pipe.reset()
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            dist.barrier()
            t_0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t_0) / reps

        def solve_only():
            solve_device(e0["C"], e0["c"], e0["F"], e0["f"], e0["x_init"], None, T, nx, nu, out=(x, u))

        def gather_only():
            pipe.gather(0)
            pipe.result(0)

        def serial():
            xs, us = pipe.local_buffers(0)
            solve_device(e0["C"], e0["c"], e0["F"], e0["f"], e0["x_init"], None, T, nx, nu, out=(xs, us))
            pipe.gather(0)
            pipe.result(0)
            torch.cuda.current_stream().synchronize()

        ts = torch.tensor([timed(solve_only), timed(gather_only), timed(serial)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        gather_info = {"solve_ms": float(ts[0]) * 1e3, "gather_ms": float(ts[1]) * 1e3, "serial_ms": float(ts[2]) * 1e3,
                       "overlapped_ms": elapsed / args.steps * 1e3, "chunks": pipe.chunks,
                       "gathered_bytes_per_rank": world * (x.numel() + u.numel()) * 4,
                       "what": "max over ranks; overlapped = the timed region of this line (solve k+1 on the launch stream "
                               "while the chunked all-gather of solve k's (x, u) runs on a side stream)"}
    # for the parity check: the solution of set 0, whatever set the last timed step solved
    solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
    torch.cuda.synchronize()
    x_host, u_host = x.cpu().numpy(), u.cpu().numpy()
    # context for the number above (N = 1): the same launch re-solving ONE input set (what rounds 1-2 timed: a set that
    # fits the Infinity Cache), and what this box's memory system sustains on a plain copy
    t_same = copy_gbs = None
    if rank == 0 and world == 1:
        del sets[1:]
        torch.cuda.empty_cache()
        if n_sets > 1:
            t_same = event_time(lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu,
                                                     out=(x, u)), 100, warm=30)
        copy_gbs = hbm_copy_calibration(device)

    if rank == 0:
        bytes_per_ts = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
        alg_bytes = bytes_per_ts * B * T
        achieved = alg_bytes / kern_s / 1e9
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "lqr_solve_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                rec = tj.get(args.workload, {})
                traffic = rec.get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_src = "recorded, not measured in this run: " + str(tj.get("_source", tpath))
                    if rec.get("batch") and rec["batch"] != B:      # counters of another batch size: traffic is linear in B
                        traffic = int(traffic * B / rec["batch"])
                        traffic_src += " (collected at B = %d, scaled to B = %d)" % (rec["batch"], B)
            except Exception:
                traffic = None
        out = {
            "metric": "LQR timestep-solves/sec @ batch=%d T=%d n_x=%d n_u=%d" % (B, T, nx, nu),
            "value": world * B * T * args.steps / elapsed,
            "unit": "timestep-solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if strong_total is None else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic random LQR, B=%d per GPU%s, T=%d, n_x=%d, n_u=%d, fused "
                                   "solve_recursion (Riccati backward + rollout), inputs resident in HBM (streamed: "
                                   "input sets in rotation)"
                                   % (args.workload, B, "" if strong_total is None else " (%d in all, split over the ranks: BASELINE.json configs[4])" % strong_total,
                                      T, nx, nu),
                       "global_batch": world * B, "parallelism": "batch-shard x%d%s" % (
                           world, " + overlapped all-gather(x,u)" if gx is not None else ", no collective")
                       + ("" if backend == "nccl" else " [REHEARSAL: backend %s, every rank on %s - not a scaling number]" % (backend, device))},
            # `frac` is the SMALLER of the two clocks' figures: HIP events on the launch stream around the K timed launches
            # (`frac_events`) and the host's wall clock around the same region incl. the final synchronisation
            # (`frac_wall`, from `ms_per_step`).  The protocol of the measurement travels inside this object (the driver's
            # parsed record keeps `roofline` whole): the untimed run-up and what the same W + K steps take from a cold start.
            "roofline": {"bound": "hbm", "achieved": min(achieved, alg_bytes / (elapsed / args.steps) / 1e9),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": min(achieved, alg_bytes / (elapsed / args.steps) / 1e9) / HBM_PEAK_GBS,
                         "frac_events": achieved / HBM_PEAK_GBS,
                         "frac_wall": alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kern_name,
                         "blocks": {"n": kBlocks, "steps_each": args.steps,
                                    "ms_per_step_min": min(block_wall) / args.steps * 1e3,
                                    "ms_per_step_median": elapsed / args.steps * 1e3,
                                    "ms_per_step_max": max(block_wall) / args.steps * 1e3,
                                    "kernel_ms_min": min(block_ev) * 1e3, "kernel_ms_max": max(block_ev) * 1e3,
                                    "ms_per_step_in_order": [round(w / args.steps * 1e3, 5) for w in block_wall],
                                    "kernel_ms_in_order": [round(e * 1e3, 5) for e in block_ev],
                                    "what": "value / ms_per_step / frac are the MEDIAN of n timed regions of exactly K steps each "
                                            "(barrier + synchronize on both sides, max over ranks)"},
                         "clocks": clock_info,
                         "clocks_run_up": run_up_clock_info,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_s * 1e3,
                         "kernel_ms_wall": elapsed / args.steps * 1e3,
                         "protocol": "value/frac are taken AFTER an untimed run-up of the same step to the device's steady "
                                     "clocks (device_run_up_steps launches, device_run_up_ms), then the contract's W warm-up and "
                                     "K timed steps; cold_start_ms_per_step is the same W + K protocol run before the run-up",
                         "device_run_up_steps": pre_calls, "device_run_up_ms": pre_ms,
                         "cold_start_ms_per_step": None if cold is None else cold * 1e3,
                         "cold_start_frac": None if cold is None else alg_bytes / cold / 1e9 / HBM_PEAK_GBS},
            # what ran on the device before the W warm-up steps and the K timed ones: the same step, untimed, until its
            # time had settled (the steady power state of a running job; scripts/clock_ramp.py has the ramp)
            "cold_start": None if cold is None else {
                "ms_per_step": cold * 1e3, "value": B * T / cold,
                "what": "rank 0's own W warm-up + K timed steps BEFORE the run-up, i.e. in the first milliseconds after the "
                        "device wakes (its clocks still ramping): what `value` would be without `device_run_up`"},
            "device_run_up": {"untimed_steps": pre_calls, "ms": pre_ms,
                              "ms_per_step_first_block_cold": None if pre_times[0] is None else pre_times[0] * 1e3,
                              "ms_per_step_last_block": None if pre_times[1] is None else pre_times[1] * 1e3,
                              "what": "none (--no-settle)" if args.no_settle else
                                      "the timed step itself, back to back in blocks of 25 for >= %.0f ms until two blocks agree "
                                      "within 3 %%: a freshly woken MI355X takes 36 us for this launch in its first millisecond, "
                                      "40-45 us from the 4th to the 10th, 33 us from ~30 ms on (profiles/r03/clock_ramp.txt)" % SETTLE_MS},
        }
        out["roofline"].update({
            "input_sets_in_rotation": n_sets,
            "inputs": "%d input sets of %.0f MB in rotation (%.0f MB against the 256 MiB Infinity Cache): every step "
                      "streams its inputs from HBM" % (n_sets, set_bytes / 1e6, n_sets * set_bytes / 1e6) if n_sets > 1
                      else "one input set of %.0f MB (beyond the 256 MiB Infinity Cache by itself)" % (set_bytes / 1e6)})
        if t_same is not None:
            out["roofline"].update({"frac_same_inputs": alg_bytes / t_same / 1e9 / HBM_PEAK_GBS, "kernel_ms_same_inputs": t_same * 1e3,
                                    "same_inputs_what": "the launch re-solving ONE input set 100 times (rounds 1-2 timed "
                                                        "this; the set fits the Infinity Cache, but the solve's nt loads "
                                                        "do not allocate there)"})
        if copy_gbs is not None:
            out["roofline"].update({"box_copy_gbs": copy_gbs,
                                    "box_copy_what": "device-to-device copy of 1 GiB on this box (read + write traffic, GB/s): "
                                                     "what its memory system sustains on the plainest streaming pattern "
                                                     "(MI355X_MICROARCH.md: 6290); boxes of the pool differ"})
        if gather_info is not None:
            out["gather"] = gather_info
        parity_ok, secondary_failed = True, False
        if cb is not None:
            out["cpu_baseline"] = cb
            if cb_mp is not None:
                out["cpu_baseline_mp"] = cb_mp
            xe = float(np.max(np.abs(x_host - xr) / np.maximum(1.0, np.abs(xr))))
            ue = float(np.max(np.abs(u_host - ur) / np.maximum(1.0, np.abs(ur))))
            parity_ok = xe <= PARITY_TOL and ue <= PARITY_TOL
            out["parity"] = {"max_rel_err_x": xe, "max_rel_err_u": ue, "tolerance": PARITY_TOL, "ok": parity_ok,
                             "against": "oracle/lqr.py on identical inputs"}
        if world == 1 and args.workload == "headline" and not args.no_secondary:
            # the headline line must come out whatever happens to the secondary measurements
            sec = {}
            try:
                sec = secondary_metrics(device, d)
            except Exception as e:  # pragma: no cover
                sec["error"] = "secondary_metrics: %r" % (e,)
            try:
                sec["shape_families"] = secondary_shapes(device)
            except Exception as e:  # pragma: no cover
                sec["shape_families"] = {"error": repr(e)}
            try:
                sec["headline_f64"] = secondary_f64(device, d)
            except Exception as e:  # pragma: no cover
                sec["headline_f64"] = {"error": repr(e)}
            try:
                del d, x, u
                torch.cuda.empty_cache()
                sec["cfg5_shard"] = secondary_cfg5(device)
            except Exception as e:  # pragma: no cover
                sec["cfg5_shard"] = {"error": repr(e)}
            try:
                torch.cuda.empty_cache()
                sec["cfg5_full_1gpu"] = secondary_cfg5_full(device)
            except Exception as e:  # pragma: no cover
                sec["cfg5_full_1gpu"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            if cb_sec is not None:        # the oracle timed beside every secondary GPU number
                if "error" in cb_sec:
                    sec["cpu_baselines_error"] = cb_sec["error"]
                for leg, e in cb_sec.items():
                    if leg != "error" and isinstance(sec.get(leg), dict):
                        sec[leg]["cpu_baseline"] = e
            out["secondary"] = sec
            secondary_failed = "error" in sec or any(isinstance(v, dict) and "error" in v for v in sec.values())
        print(json.dumps(out))
        if not parity_ok:        # a fast kernel whose results differ from the reference's is not done
            print("PARITY FAILURE: %r" % (out["parity"],), file=sys.stderr)
            sys.exit(3)
        if secondary_failed and not args.allow_secondary_failure:
            print("SECONDARY MEASUREMENT FAILED (see the \"error\" entries of the JSON line)", file=sys.stderr)
            sys.exit(4)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
