"""Multi-GPU row sharding: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The reference merges per-tile results into one HashMap under a Mutex (src/sampler.rs:60-70).
Here every rank owns the supersampled rows of a block-cyclic set of 8-row blocks
(block b -> rank b % world), renders them into its own accumulator, and ONE gather per
batch of samples moves the shard accumulators to rank 0 over xGMI.  Nothing is reduced:
shards are disjoint, so a ring all-reduce (per-link bound on the xGMI mesh) would only add
traffic; a gather uses the 7 inbound links of rank 0 concurrently.

Everything here is plumbing over torch tensors; it runs unchanged on CPU tensors with the
gloo backend (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import numpy as np

DEFAULT_SHARD_ROWS = 8


def shard_row_index(nh: int, rank: int, world: int, shard_rows: int = DEFAULT_SHARD_ROWS) -> np.ndarray:
    """Frame rows owned by `rank`, in local order (same rule as mrt_create, csrc/mrt_api.cpp)."""
    y = np.arange(nh)
    return y[(y // shard_rows) % world == rank].astype(np.int64)


def padded_rows(nh: int, world: int, shard_rows: int = DEFAULT_SHARD_ROWS) -> int:
    """Rows every shard buffer is allocated for (the largest shard), so gathers are equal-sized."""
    if world == 1:
        return nh
    n_blocks = (nh + shard_rows - 1) // shard_rows
    return ((n_blocks + world - 1) // world) * shard_rows


def probe_gather(device=None, group=None) -> bool:
    """Decide ONCE, on every rank alike, whether this backend build offers `gather`; returns True when all_gather has
    to stand in for it.  A build without the op raises before anything is sent, on every rank (NotImplementedError /
    "not supported" RuntimeError); the ranks then agree on the answer with one all_reduce.  Called once per
    ShardedSampler: later failures of a collective are real failures and propagate."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    staged = device is not None and dist.get_backend(group) != "nccl"
    t = torch.zeros(4, dtype=torch.float32, device="cpu" if (staged or device is None) else device)
    missing = 0
    try:
        dist.gather(t, gather_list=[torch.empty_like(t) for _ in range(world)] if rank == 0 else None, dst=0, group=group)
    except NotImplementedError:
        missing = 1
    except RuntimeError as e:
        if "not supported" not in str(e).lower() and "not implemented" not in str(e).lower():
            raise
        missing = 1
    flag = torch.tensor([missing], dtype=torch.int32, device=t.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(flag.item())


def gather_frame(local, nh: int, nw: int, shard_rows: int = DEFAULT_SHARD_ROWS, dst: int = 0, group=None, out=None,
                 use_all_gather: bool = False, parts=None):
    """Gather shard accumulators ([padded_rows, nw, 3] f32 tensors, one per rank) to rank `dst` and
    place their rows into the full frame [nh, nw, 3].  Returns the frame on `dst`, None elsewhere.
    `use_all_gather` (from probe_gather, the same on every rank) selects all_gather for backends without gather.
    `parts`: receive buffers (world tensors shaped like the sent one) to reuse instead of allocating them per call."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    pr = padded_rows(nh, world, shard_rows)
    assert tuple(local.shape) == (pr, nw, 3), (tuple(local.shape), (pr, nw, 3))
    if world == 1:
        return local[:nh]
    # RCCL moves device tensors directly; a CPU backend (gloo, used by the tests and the one-GPU rehearsal) is fed
    # through host staging copies.
    staged = local.is_cuda and dist.get_backend(group) != "nccl"
    send = local.cpu() if staged else local
    if parts is not None:
        assert len(parts) == world and all(p.shape == send.shape and p.device == send.device for p in parts)
    if use_all_gather:
        parts = parts if parts is not None else [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
    elif rank == dst:
        parts = parts if parts is not None else [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=parts, dst=dst, group=group)
    else:
        dist.gather(send, gather_list=None, dst=dst, group=group)
    if rank != dst:
        return None
    frame = out if out is not None else torch.empty((nh, nw, 3), dtype=local.dtype, device=local.device)
    for r, part in enumerate(parts):
        rows = _rows_on(local.device, nh, r, world, shard_rows)
        frame.index_copy_(0, rows, part[: rows.numel()].to(local.device))
    return frame


_ROWS_CACHE = {}


def _rows_on(device, nh, rank, world, shard_rows):
    """shard_row_index as a tensor on `device`, cached (the gather runs once per batch of samples)."""
    import torch
    key = (str(device), nh, rank, world, shard_rows)
    if key not in _ROWS_CACHE:
        _ROWS_CACHE[key] = torch.from_numpy(shard_row_index(nh, rank, world, shard_rows)).to(device)
    return _ROWS_CACHE[key]


class ShardedSampler:
    """Sampler for rank `rank` of `world` GPUs: execute() renders this rank's rows and gathers the frame
    on rank 0; img() is served by rank 0 (Sampler::img needs every row).

    The render description is fixed at construction: a shard's rows live in a tensor bound to the library and its sample
    count cannot be restored into a rebuilt context, so editing `self.render` in place once samples are on board makes the
    next execute() raise MrtError (MRT_ERR_STATE) instead of silently rendering into fresh memory; build a new
    ShardedSampler for another scene."""

    def __init__(self, render, rank: int, world: int, device: int, seed: int = 1, shard_rows: int = DEFAULT_SHARD_ROWS, flags: int = 0):
        import torch
        from .sampler import Sampler

        self.rank, self.world, self.shard_rows = rank, world, shard_rows
        self.render = render
        self.s = Sampler(seed=seed, device=device, shard_index=rank, shard_count=world, shard_rows=shard_rows, flags=flags).create(render)
        self.nw, self.nh = self.s.nw, self.s.nh
        pr = self.s.padded_rows()
        assert pr == padded_rows(self.nh, world, shard_rows)
        self.dev = torch.device("cuda", device)
        if world > 1:
            # the accumulator lives in a torch tensor so that RCCL can send it without a copy
            self.local = torch.zeros((pr, self.nw, 3), dtype=torch.float32, device=self.dev)
            self.s.bind_accum(self.local.data_ptr(), self.local.numel() * 4)
            self.frame = torch.zeros((self.nh, self.nw, 3), dtype=torch.float32, device=self.dev) if rank == 0 else None
        else:
            self.local = self.frame = None       # one rank: the frame stays inside the library (mrt_accum / mrt_img serve it)
        self.count = 0
        self.use_all_gather = probe_gather(self.dev) if world > 1 else False
        # receive buffers of the gather, allocated once (rank 0, or every rank when all_gather stands in): 7 x 12.4 MB at 4K
        # per exchange otherwise.  A CPU backend (gloo rehearsals) receives into host memory.
        self._parts = None
        if world > 1 and (rank == 0 or self.use_all_gather):
            import torch.distributed as dist
            staged = dist.get_backend() != "nccl"
            self._parts = [torch.empty((pr, self.nw, 3), dtype=torch.float32, device="cpu" if staged else self.dev) for _ in range(world)]
        self.last_gather_ms = 0.0       # wall time of the last exchange on this rank (gather + row placement, synchronised)

    def execute(self, n_samples: int = 1, gather: bool = True):
        secs = self.s.execute(self.render, n_samples=n_samples)
        self.count += n_samples
        self.last_gather_ms = 0.0
        if gather and self.world > 1:
            import time
            t0 = time.perf_counter()
            gather_frame(self.local, self.nh, self.nw, self.shard_rows, dst=0, out=self.frame, use_all_gather=self.use_all_gather,
                         parts=self._parts)
            if self.world > 1:
                # like mrt_execute, return only when the exchange is done: the next launch (on the library's own stream)
                # accumulates into the buffer the collective is still reading
                import torch
                torch.cuda.current_stream(self.dev).synchronize()
                self.last_gather_ms = (time.perf_counter() - t0) * 1e3
        return secs

    def img(self):
        if self.rank != 0:
            return None
        if self.world == 1:
            return self.s.img()              # the context already holds every row
        import torch
        torch.cuda.synchronize(self.dev)
        self.s.set_accum_device(self.frame.data_ptr(), self.count)
        return self.s.img()

    def close(self):
        self.s.close()
