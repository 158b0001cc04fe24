"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed layer.  Every trajectory of the LQR / KKT-gradient / MPC-step path is
independent, so the batch axis shards with NO data-path collective: each rank owns a contiguous slice
`[b0, b1)` of every time-major `[T, B, ...]` tensor.  The only exchanges are optional and at the end:
  * `all_gather_batch`  - the final x*, u* (SURVEY.md 8e); a direct all-gather, outputs stay sharded unless
    the caller asks for them;
  * `all_reduce_param_grad` - gradients AFTER reduction to parameter shape (e.g. dA, dB = sum_{t,b} dF of
    LqrNet: `expand_time_batch`'s backward is a sum), never the per-sample dC/dF;
  * `GatherPipeline` - the same all-gather taken OFF the solver's stream (at config 5 the gather of
    (x*, u*) costs about what the solve costs): solve k+1 runs on the caller's stream while the (x, u) of solve k travel,
    in chunks, on a side stream; two output buffers in rotation.

A backend that cannot move device memory (gloo: the CPU tests, and the two-ranks-on-one-GPU test of the HIP path) is served
by staging the collective through host memory - same calls, same results.
"""
import torch
import torch.distributed as dist


def _needs_host_staging(t, group=None):
    """gloo has no device all-gather: device tensors go through the host for the collective"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_into(out, inp, group=None):
    """dist.all_gather_into_tensor for any backend / device combination; `out` is [world * n, ...] or [world, n, ...] for an
    input [n, ...] (gloo accepts only the concatenated form: the stacked one is passed as its flat view)"""
    if out.dim() == inp.dim() + 1:
        out = out.view((out.shape[0] * out.shape[1],) + tuple(out.shape[2:]))
    if _needs_host_staging(inp, group):
        h_out = torch.empty(out.shape, dtype=out.dtype, device="cpu")
        dist.all_gather_into_tensor(h_out, inp.cpu(), group=group)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


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
        _all_gather_into(every, mine, group)
        sizes = [int(v) for v in every.cpu().tolist()]
    moved = local.movedim(batch_dim, 0).contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((world * moved.shape[0],) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        _all_gather_into(out, moved, group)
    else:   # ragged split: pad every shard to the largest, gather, drop the padding
        mx = max(sizes)
        padded = torch.zeros((mx,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        padded[: moved.shape[0]] = moved
        buf = torch.empty((world * mx,) + tuple(moved.shape[1:]), dtype=moved.dtype, device=moved.device)
        _all_gather_into(buf, padded, group)
        out = torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
    return out.movedim(0, batch_dim).contiguous()


def all_reduce_param_grad(g, group=None):
    """sum a parameter-shaped gradient over ranks (in place)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if _needs_host_staging(g, group):
            h = g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            g.copy_(h)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


class GatherPipeline:
    """All-gather of a solve's outputs overlapped with the NEXT solve (one process per GPU, equal shards).

        pipe = GatherPipeline([(T, b, nx), (T, b, nu)], device)           # shapes of the local x, u
        for k in range(steps):
            x, u = pipe.local_buffers(k)            # where solve k writes (two sets in rotation)
            solve_device(..., out=(x, u))           # caller's stream
            pipe.gather(k)                          # enqueued on the side stream; returns at once
        gx, gu = pipe.result(k)                     # the gathered x, u of solve k (pieces), the caller's stream waits
        x_all = GatherPipeline.as_time_major(gx)    # [T, world * b, nx], the reference's layout (one copy)

    Each tensor travels in `chunks` pieces along the time axis, one collective per piece, written straight into its final
    place: a gathered tensor is a list of pieces `[world, dt, b, ...]` (rank-major inside a piece - what an all-gather writes
    without a transposing copy), all pieces of a tensor carved out of one allocation.  Pieces of a few tens of MB keep
    RCCL's ring busy without holding the side stream for the whole 270 MB of a config-5 shard in one call.  Ordering: the
    side stream waits for solve k (an event on the caller's stream); the caller's stream waits for gather k-2 before solve
    k overwrites that buffer set (`local_buffers`), and for gather k in `result(k)`."""

    def __init__(self, shapes, device, dtype=torch.float32, chunks=None, group=None, max_chunk_bytes=64 << 20):
        self.group = group
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.shapes = [tuple(int(v) for v in s) for s in shapes]
        T = self.shapes[0][0]
        assert all(s[0] == T for s in self.shapes), "every tensor is time-major with the same horizon"
        if chunks is None:
            esz = torch.empty(0, dtype=dtype).element_size()
            biggest = max(esz * self.world * _prod(s) for s in self.shapes)
            chunks = max(1, min(T, -(-biggest // max_chunk_bytes)))
        self.chunks = max(1, min(int(chunks), T))
        step = -(-T // self.chunks)
        self.slices = [(t0, min(T, t0 + step)) for t0 in range(0, T, step)]
        self.local = [[torch.empty(s, dtype=dtype, device=self.device) for s in self.shapes] for _ in range(2)]
        self.out = [[self._pieces(s, dtype) for s in self.shapes] for _ in range(2)]
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.gathered = [None, None]    # event: gather k has finished (side stream)
        self.step_of = [None, None]     # which solve each buffer set holds (or is receiving)
        self.newest = None              # the latest solve handed to gather()
        if self.world > 1:              # equal shards are assumed by every collective below: checked once, not per gather
            mine = torch.tensor([self.shapes[0][1]], dtype=torch.int64, device=self.device)
            every = torch.empty((self.world,), dtype=torch.int64, device=self.device)
            _all_gather_into(every, mine, group)
            sizes = [int(v) for v in every.cpu().tolist()]
            if len(set(sizes)) != 1:
                raise ValueError("GatherPipeline needs equal shards on every rank, got batch sizes %r "
                                 "(all_gather_batch handles ragged shards)" % (sizes,))

    def _pieces(self, shape, dtype):
        flat = torch.empty((self.world * _prod(shape),), dtype=dtype, device=self.device)
        pieces, off = [], 0
        for (t0, t1) in self.slices:
            n = self.world * (t1 - t0) * _prod(shape[1:])
            pieces.append(flat[off:off + n].view((self.world, t1 - t0) + tuple(shape[1:])))
            off += n
        return pieces

    def reset(self):
        """forget which solves the buffer sets hold (both streams are waited for first): the next solve may be numbered anew"""
        if self.cuda:
            torch.cuda.current_stream(self.device).synchronize()
            self.side.synchronize()
        self.gathered, self.step_of, self.newest = [None, None], [None, None], None

    def _check_current(self, k, what):
        """buffer set k % 2 must hold solve k: an older k names a set that a later gather is overwriting on the side stream"""
        if self.step_of[k % 2] != k:
            raise ValueError("GatherPipeline.%s(%d): that buffer set holds solve %r (two sets in rotation: only the newest "
                             "solve and the one before it are kept, newest = %r)" % (what, k, self.step_of[k % 2], self.newest))

    def local_buffers(self, k):
        """the buffer set solve k writes; the caller's stream first waits for the gather that last read it (k - 2)"""
        if self.newest is not None and k < self.newest - 1:
            raise ValueError("GatherPipeline.local_buffers(%d): solve %d has already been gathered" % (k, self.newest))
        if self.step_of[k % 2] is not None and self.step_of[k % 2] > k:
            raise ValueError("GatherPipeline.local_buffers(%d): that buffer set already holds solve %d" % (k, self.step_of[k % 2]))
        ev = self.gathered[k % 2]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return self.local[k % 2]

    def gather(self, k):
        """enqueue the all-gather of buffer set k % 2 behind the work now on the caller's stream, on the side stream"""
        if self.step_of[k % 2] is not None and self.step_of[k % 2] > k:
            raise ValueError("GatherPipeline.gather(%d): that buffer set already holds solve %d" % (k, self.step_of[k % 2]))
        self.step_of[k % 2] = k
        self.newest = k if self.newest is None else max(self.newest, k)
        loc, out = self.local[k % 2], self.out[k % 2]
        if not self.cuda:
            for pieces, t in zip(out, loc):
                for piece, (t0, t1) in zip(pieces, self.slices):
                    _all_gather_into(piece, t[t0:t1], self.group)
            return
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(done)
            for pieces, t in zip(out, loc):
                for piece, (t0, t1) in zip(pieces, self.slices):
                    _all_gather_into(piece, t[t0:t1], self.group)     # t[t0:t1] of a time-major tensor is contiguous
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.gathered[k % 2] = ev

    def result(self, k):
        """the gathered tensors of solve k (one list of pieces per tensor); the caller's stream waits for their arrival"""
        self._check_current(k, "result")
        ev = self.gathered[k % 2]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return self.out[k % 2]

    @staticmethod
    def as_time_major(pieces):
        """pieces [world, dt, b, ...] -> [T, world * b, ...] (a contiguous copy): the reference's layout"""
        rows = []
        for g in pieces:
            w, dt, b = g.shape[:3]
            rows.append(g.transpose(0, 1).reshape((dt, w * b) + tuple(g.shape[3:])))
        return torch.cat(rows, dim=0)


def _prod(shape):
    n = 1
    for v in shape:
        n *= int(v)
    return n
