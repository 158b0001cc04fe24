#!/usr/bin/env python
"""Headline benchmark: LQR timestep-solves/sec @ batch=4096, T=50, n_x=8, n_u=2 (BASELINE.json).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one fused `solve_recursion` (backward Riccati sweep + forward rollout, x and u
materialised in HBM) over one batch of synthetic trajectories already resident in HBM.  With N
GPUs every rank solves its own shard of B trajectories (independent units, no data-path
collective; `--gather` adds the RCCL all-gather of (x*, u*) to the timed region), so scaling is
weak and `value` = N*B*T*K / max-over-ranks time.  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline     dominant kernel vs the HBM roof: algorithmic bytes per launch (832 B per
               timestep-solve) / average launch duration measured with HIP events on
               the launch stream over the timed region; peak 8 TB/s (MI355X_MICROARCH.md).
  cpu_baseline the numpy float64 oracle (a port of the reference's algorithm; the reference's
               Python cannot travel) timed on this host's cores on a bounded sample.
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

WORKLOADS = {
    # name: (B per GPU, T, nx, nu)
    "headline": (4096, 50, 8, 2),        # BASELINE.json configs[2] - the configuration the metric is quoted on
    "cfg5-shard": (8192, 50, 32, 8),     # BASELINE.json configs[4]: 65536 trajectories sharded over 8 GPUs
    "pendulum": (1024, 20, 3, 1),        # shapes of configs[1]/[3] (pure LQR part)
}


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


def kernel_name(T, B, nx, nu):
    """the kernel dmpc_lqr_solve dispatches to at this size (what rocprofv3 --kernel-trace lists)"""
    from chainer_differentiable_mpc_amd import _lib
    path = _lib.load().dmpc_lqr_solve_path(T, B, nx, nu)
    # template arguments as rocprofv3 prints them: <nx, nu, has_f, write_k, stash, masked>; the bench passes f, no gains
    return {0: "dmpc::lqr_generic_kernel", 1: "void dmpc::lqr_kernel<%d, %d, ...>(dmpc::LqrArgs)" % (nx, nu),
            2: "void dmpc::lqr_dma_kernel<%d, %d, ...>(dmpc::LqrArgs)" % (nx, nu),
            3: "void dmpc::lqr_asm_kernel<%d, %d, true, false, false, false>(dmpc::LqrArgs)" % (nx, nu),
            4: "void dmpc::lqr_asm_kernel<%d, %d, true, false, true, false>(dmpc::LqrArgs)" % (nx, nu),
            5: "void dmpc::lqr_wave_mfma_backward<%d, %d>(dmpc::LqrArgs) + forward-only dmpc::lqr_kernel" % (nx, nu)
            }.get(path, "?")


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--gather", action="store_true", help="all-gather (x*, u*) over RCCL inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    B, T, nx, nu = WORKLOADS[args.workload]
    p, d = make_inputs(B, T, nx, nu, seed=rank, device=device)
    x = torch.empty((T, B, nx), dtype=torch.float32, device=device)
    u = torch.empty((T, B, nu), dtype=torch.float32, device=device)
    gx = gu = None
    if args.gather and world > 1:
        gx = torch.empty((world,) + tuple(x.shape), dtype=torch.float32, device=device)
        gu = torch.empty((world,) + tuple(u.shape), dtype=torch.float32, device=device)

    def step():
        solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
        if gx is not None:
            dist.all_gather_into_tensor(gx, x)
            dist.all_gather_into_tensor(gu, u)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # kernel duration: ONE pair of HIP events on the launch stream (torch's current stream) around the K
    # back-to-back launches of the timed region; span / K = average launch duration incl. boundaries
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps

    if rank == 0:
        bytes_per_ts = synthetic.lqr_algorithmic_bytes_per_timestep(nx, nu)
        alg_bytes = bytes_per_ts * B * T
        achieved = alg_bytes / kern_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "lqr_solve_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "LQR timestep-solves/sec @ batch=%d T=%d n_x=%d n_u=%d" % (B, T, nx, nu),
            "value": world * B * T * args.steps / elapsed,
            "unit": "timestep-solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic random LQR, B=%d per GPU, T=%d, n_x=%d, n_u=%d, fused "
                                   "solve_recursion (Riccati backward + rollout), inputs resident in HBM"
                                   % (args.workload, B, T, nx, nu),
                       "global_batch": world * B, "parallelism": "batch-shard x%d%s" % (
                           world, " + all-gather(x,u)" if gx is not None else ", no collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name(T, B, nx, nu),
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_s * 1e3},
        }
        if not args.no_cpu_baseline and p is not None and world == 1:   # the CPU leg runs on rank 0 at N = 1 only
            cb, xr, ur = cpu_baseline(p, T, nx, nu, args.cpu_seconds)
            out["cpu_baseline"] = cb
            xe = float(np.max(np.abs(x.cpu().numpy() - xr) / np.maximum(1.0, np.abs(xr))))
            ue = float(np.max(np.abs(u.cpu().numpy() - ur) / np.maximum(1.0, np.abs(ur))))
            out["parity"] = {"max_rel_err_x": xe, "max_rel_err_u": ue, "tolerance": 1e-4,
                             "against": "oracle/lqr.py on identical inputs"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
