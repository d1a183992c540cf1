"""Host-side multi-GPU plumbing: one process per GPU, scan sharded by contiguous point blocks.

The data path never leaves the HIP module: per iteration each rank's kernel reduces its shard to one
28-double row, RCCL all-reduces the rows over xGMI on the module's own stream, and every rank runs the
identical 6x6 solve (reference thread merge: src/Registration.cpp:71-75).  What lives here is only
what the host language has to do around it: deciding who owns which points, and carrying the
128-byte RCCL unique id from rank 0 to everyone (any transport works; torch.distributed is used
because the launcher already set it up).
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of an n-point scan owned by `rank`: sizes differ by at most one and
    the blocks tile [0, n) in rank order."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(int(n), world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def share_bytes(payload_on_rank0: Callable[[], bytes], rank: int, nbytes: int, group=None) -> bytes:
    """Broadcast `nbytes` produced on rank 0 to every rank of the torch.distributed group."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    if rank == 0:
        raw = payload_on_rank0()
        if len(raw) != nbytes:
            raise ValueError(f"expected {nbytes} bytes, got {len(raw)}")
        buf = torch.tensor(list(raw), dtype=torch.uint8, device=device)
    else:
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=0, group=group)
    return bytes(buf.cpu().tolist())


def gather_bytes(payload: bytes, world_size: int, group=None) -> bytes:
    """All-gather one equally sized byte string per rank; returns them concatenated in rank order (what
    vgicp_peer_connect takes when the mailboxes of the device-initiated exchange are wired by hand;
    vgicp_comm_init does the same through RCCL by itself)."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor(list(payload), dtype=torch.uint8, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world_size)]
    dist.all_gather(parts, mine, group=group)
    return b"".join(bytes(p.cpu().tolist()) for p in parts)


def share_unique_id(ctx, rank: int, group=None) -> bytes:
    """Rank 0 asks the HIP module for an RCCL unique id; everyone receives the same 128 bytes."""
    from . import capi
    return share_bytes(ctx.comm_unique_id, rank, capi.UNIQUE_ID_BYTES, group)
