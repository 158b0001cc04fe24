"""CPU, world_size 2, gloo: the batch-sharding path (shard -> independent local solves -> all-gather).
The local solver here is the numpy oracle (test infrastructure) - on the GPU box the same helpers wrap
the HIP kernels (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chainer_differentiable_mpc_amd import synthetic
from chainer_differentiable_mpc_amd.dist import all_gather_batch, all_reduce_param_grad, shard_bounds, shard_problem
from oracle import lqr as olqr


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, B, T, nx, nu, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=3)
    full = [torch.as_tensor(p[k]) for k in ("x_init", "C", "c", "F", "f")]
    x0, C, c, F, f = shard_problem(*full)
    b0, b1 = shard_bounds(B, rank, world)
    assert C.shape[1] == b1 - b0 and x0.shape[0] == b1 - b0
    x, u = olqr.lqr_solve(x0.numpy(), C.numpy(), c.numpy(), F.numpy(), f.numpy(), T, nx, nu)
    gx = all_gather_batch(torch.as_tensor(x), B)
    gu = all_gather_batch(torch.as_tensor(u), B)
    gu2 = all_gather_batch(torch.as_tensor(u))          # sizes not given: exchanged first, ragged shards stay correct
    assert torch.equal(gu, gu2)
    # a parameter-shaped reduction: sum_{t,b} of something per-sample
    g = torch.as_tensor(x).sum(dim=(0, 1))
    all_reduce_param_grad(g)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), x=gx.numpy(), u=gu.numpy(), g=g.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])          # even and ragged split
def test_shard_solve_gather_world2(tmp_path, B):
    T, nx, nu = 5, 4, 2
    port = free_port()
    mp.spawn(worker, args=(2, port, B, T, nx, nu, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=3)
    xr, ur = olqr.lqr_solve(p["x_init"], p["C"], p["c"], p["F"], p["f"], T, nx, nu)
    np.testing.assert_allclose(got["x"], xr, rtol=0, atol=1e-12)    # sharding changes nothing
    np.testing.assert_allclose(got["u"], ur, rtol=0, atol=1e-12)
    np.testing.assert_allclose(got["g"], xr.sum(axis=(0, 1)), rtol=1e-12, atol=1e-12)


def test_shard_bounds_partition_the_batch():
    for B in (1, 5, 8, 4096, 65536):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def pipeline_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chainer_differentiable_mpc_amd.dist import GatherPipeline
    T, b, nx, nu = 7, 3, 4, 2
    for chunks in (1, 3, None):
        pipe = GatherPipeline([(T, b, nx), (T, b, nu)], "cpu", chunks=chunks)
        got = []
        for k in range(4):               # four "solves" through the two buffer sets
            x, u = pipe.local_buffers(k)
            x.copy_(torch.arange(T * b * nx, dtype=torch.float32).view(T, b, nx) + 1000 * rank + 10000 * k)
            u.copy_(-(torch.arange(T * b * nu, dtype=torch.float32).view(T, b, nu) + 1000 * rank + 10000 * k))
            pipe.gather(k)
            gx, gu = pipe.result(k)
            got.append((GatherPipeline.as_time_major(gx).clone(), GatherPipeline.as_time_major(gu).clone()))
        for k, (gx, gu) in enumerate(got):
            for r in range(world):
                want = torch.arange(T * b * nx, dtype=torch.float32).view(T, b, nx) + 1000 * r + 10000 * k
                assert torch.equal(gx[:, r * b:(r + 1) * b], want), (chunks, k, r)
                want_u = -(torch.arange(T * b * nu, dtype=torch.float32).view(T, b, nu) + 1000 * r + 10000 * k)
                assert torch.equal(gu[:, r * b:(r + 1) * b], want_u), (chunks, k, r)
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pipeline_world2_chunked(tmp_path):
    """`GatherPipeline` (the all-gather of (x*, u*) taken off the solver's stream, chunked along time): every piece of
    every rank lands at its [T, B, ...] position, through both buffer sets, for 1, 3 and the default number of chunks"""
    port = free_port()
    mp.spawn(pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))


def pipeline_guard_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chainer_differentiable_mpc_amd.dist import GatherPipeline
    T, b, nx = 4, 3, 2
    pipe = GatherPipeline([(T, b, nx)], "cpu", chunks=2)
    for k in range(3):
        pipe.local_buffers(k)[0].fill_(float(10 * k + rank))
        pipe.gather(k)
    assert float(GatherPipeline.as_time_major(pipe.result(2)[0])[0, 0, 0]) == 20.0
    assert float(GatherPipeline.as_time_major(pipe.result(1)[0])[0, b, 0]) == 11.0      # the one before the newest is still there
    for bad in (lambda: pipe.result(0), lambda: pipe.local_buffers(0), lambda: pipe.gather(0)):   # set 0 holds solve 2 by now
        try:
            bad()
        except ValueError as e:
            assert "solve" in str(e)
        else:
            raise AssertionError("a stale step index went through")
    pipe.reset()
    pipe.local_buffers(0)
    pipe.gather(0)
    pipe.result(0)
    ragged = None
    try:       # shards of different sizes: refused at construction (every rank raises: the size exchange is a collective)
        GatherPipeline([(T, b + rank, nx)], "cpu")
    except ValueError as e:
        ragged = str(e)
    assert ragged is not None and "equal shards" in ragged
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pipeline_refuses_stale_steps_and_ragged_shards(tmp_path):
    """the two buffer sets carry the number of the solve they hold - `result(k - 2)` after `gather(k)` used to read
    a buffer the side stream was overwriting; shards of different sizes used to reach the equal-size collective"""
    port = free_port()
    mp.spawn(pipeline_guard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))
