"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed layer.  Every trajectory of the LQR / KKT-gradient / MPC-step path is
independent, so the batch axis shards with NO data-path collective: each rank owns a contiguous slice
`[b0, b1)` of every time-major `[T, B, ...]` tensor.  The only exchanges are optional and at the end:
  * `all_gather_batch`  - the final x*, u* (SURVEY.md 8e); a direct all-gather, outputs stay sharded unless
    the caller asks for them;
  * `all_reduce_param_grad` - gradients AFTER reduction to parameter shape (e.g. dA, dB = sum_{t,b} dF of
    LqrNet: `expand_time_batch`'s backward is a sum), never the per-sample dC/dF.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_batch, rank, world):
    """contiguous, balanced: the first (n_batch % world) ranks get one extra trajectory"""
    base, extra = divmod(n_batch, world)
    b0 = rank * base + min(rank, extra)
    return b0, b0 + base + (1 if rank < extra else 0)


def shard_batch(t, rank, world, batch_dim=1):
    """this rank's slice of a tensor whose `batch_dim` is the batch axis (None passes through)"""
    if t is None:
        return None
    b0, b1 = shard_bounds(t.shape[batch_dim], rank, world)
    return t.narrow(batch_dim, b0, b1 - b0).contiguous()


def shard_problem(x_init, C, c, F, f, rank=None, world=None):
    """(x_init [B,nx], C [T,B,..], c, F, f|None) -> this rank's shard"""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return (shard_batch(x_init, rank, world, 0), shard_batch(C, rank, world), shard_batch(c, rank, world),
            shard_batch(F, rank, world), shard_batch(f, rank, world))


def all_gather_batch(local, n_batch_total=None, batch_dim=1, group=None):
    """all-gather along the batch axis; shards may be ragged.  With `n_batch_total` the shard sizes are those of
    `shard_bounds` (no extra exchange); without it the local sizes are all-gathered first, so ragged shards can
    never reach the equal-size collective by accident."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if n_batch_total is not None:
        sizes = [shard_bounds(n_batch_total, r, world)[1] - shard_bounds(n_batch_total, r, world)[0]
                 for r in range(world)]
        assert local.shape[batch_dim] == sizes[dist.get_rank(group)], "local shard does not match shard_bounds"
    else:
        mine = torch.tensor([local.shape[batch_dim]], dtype=torch.int64, device=local.device)
        every = torch.empty((world,), dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, mine, group=group)
        sizes = [int(v) for v in every.cpu().tolist()]
    moved = local.movedim(batch_dim, 0).contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((world * moved.shape[0],) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        dist.all_gather_into_tensor(out, moved, group=group)
    else:   # ragged split: pad every shard to the largest, gather, drop the padding
        mx = max(sizes)
        padded = torch.zeros((mx,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        padded[: moved.shape[0]] = moved
        buf = torch.empty((world * mx,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        out = torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
    return out.movedim(0, batch_dim).contiguous()


def all_reduce_param_grad(g, group=None):
    """sum a parameter-shaped gradient over ranks (in place)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g
