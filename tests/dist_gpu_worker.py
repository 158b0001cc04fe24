"""Worker of tests/test_dist_gpu.py: ONE rank of the product path under torch.distributed.  Started as a fresh process
(nothing has touched the GPU before this file runs), so it is what `bench.py --gpus N` / a user's launcher does per rank:

    shard_problem -> solve_device / kkt_grad_device (the HIP library) -> all_gather_batch / all_reduce_param_grad
    (+ the overlapped GatherPipeline over three consecutive solves)

Default: both ranks use GPU 0 (a one-GPU box) and the backend is gloo (RCCL refuses two ranks on one device), which dist.py
serves by staging the collectives through the host.  DIST_BACKEND=nccl: RCCL, rank r on device LOCAL_RANK - two devices for two
ranks, or a process group of ONE rank on device 0 (the collectives then run through RCCL's own code path on the side stream,
which is what a one-GPU box can prove of it).  Rank 0 also solves the UNSHARDED problem with the same library and writes
both to `out` for the parent to compare bit for bit.  Environment: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, B, T, nx, nu = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("DIST_BACKEND", "gloo")
    dev_index = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from chainer_differentiable_mpc_amd import _lib, synthetic
    from chainer_differentiable_mpc_amd.differentiable_lqr import kkt_grad_device
    from chainer_differentiable_mpc_amd.dist import (GatherPipeline, all_gather_batch, all_reduce_param_grad, shard_bounds,
                                                     shard_problem)
    from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    p = synthetic.make_lqr_problem(B, T, nx, nu, seed=11)
    full = [torch.as_tensor(p[k], dtype=torch.float32, device=dev) for k in ("x_init", "C", "c", "F", "f")]
    x0, C, c, F, f = shard_problem(*full)
    b0, b1 = shard_bounds(B, rank, world)
    assert C.shape[1] == b1 - b0
    x, u, _, _ = solve_device(C, c, F, f, x0, None, T, nx, nu)
    kernel = _lib.last_kernel_name()
    rng = np.random.RandomState(5)
    gx_full = torch.as_tensor(rng.randn(T, B, nx), dtype=torch.float32, device=dev)
    gu_full = torch.as_tensor(rng.randn(T, B, nu), dtype=torch.float32, device=dev)
    gx, gu = gx_full[:, b0:b1].contiguous(), gu_full[:, b0:b1].contiguous()
    dx0, dC, dc, dF, df = kkt_grad_device(C, c, F, x, u, gx, gu, T, nx, nu)
    # the exchanges of SURVEY.md 8e: the final (x*, u*) and a gradient reduced to parameter shape (dA, dB of LqrNet = the sum
    # of dF over time and batch: expand_time_batch's backward)
    x_all, u_all = all_gather_batch(x, B), all_gather_batch(u, B)
    dF_sum = all_reduce_param_grad(dF.double().sum(dim=(0, 1)))
    # the overlapped pipeline: three solves of three problems (x_init scaled), gathers on the side stream
    pipe = GatherPipeline([(T, b1 - b0, nx), (T, b1 - b0, nu)], dev, chunks=3)
    piped = []
    for k in range(3):
        xs, us = pipe.local_buffers(k)
        solve_device(C, c, F, f, x0 * (1.0 + 0.5 * k), None, T, nx, nu, out=(xs, us))
        pipe.gather(k)
        if k >= 1:      # consume the previous gather while this one is in flight
            piped.append(GatherPipeline.as_time_major(pipe.result(k - 1)[0]).clone())
    piped.append(GatherPipeline.as_time_major(pipe.result(2)[0]).clone())
    if backend == "nccl":     # RCCL's all-reduce and barrier too, whatever the group's size
        ones = torch.ones(4, device=dev)
        dist.all_reduce(ones)
        assert float(ones.sum()) == 4.0 * world
    torch.cuda.synchronize()
    if rank == 0:
        xf, uf, _, _ = solve_device(full[1], full[2], full[3], full[4], full[0], None, T, nx, nu)
        g_full = kkt_grad_device(full[1], full[2], full[3], xf, uf, gx_full, gu_full, T, nx, nu)
        piped_ref = []
        for k in range(3):
            xk, _, _, _ = solve_device(full[1], full[2], full[3], full[4], full[0] * (1.0 + 0.5 * k), None, T, nx, nu)
            piped_ref.append(xk)
        np.savez(out_path, x_all=x_all.cpu().numpy(), u_all=u_all.cpu().numpy(), x_full=xf.cpu().numpy(), u_full=uf.cpu().numpy(),
                 dF_sum=dF_sum.cpu().numpy(), dF_sum_full=g_full[3].double().sum(dim=(0, 1)).cpu().numpy(),
                 dx0_local=dx0.cpu().numpy(), dx0_full=g_full[0].cpu().numpy(), b0=b0, b1=b1,
                 piped=np.stack([t.cpu().numpy() for t in piped]), piped_ref=np.stack([t.cpu().numpy() for t in piped_ref]),
                 kernel=np.array(kernel), lib=np.array(_lib.LIB_PATH), backend=np.array(dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
