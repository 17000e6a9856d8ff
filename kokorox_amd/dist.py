"""Multi-GPU plumbing for the Kokoro forward: utterances shard, weights broadcast once.

The reference has no multi-device path at all (one `Mutex<Session>`,
/root/reference/kokorox/src/onn/ort_koko.rs:14,78).  Utterances (and the <=500-token
chunks of one long text, kokorox/src/tts/koko.rs:947) are independent, so the batch is
split contiguously across ranks with NO steady-state collective; the only exchange is a
one-time broadcast of the weight blob from rank 0 (RCCL over xGMI on GPUs, gloo in the CPU
tests).  torch.distributed is used as plumbing only.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n_items for this rank; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_list(items: List, rank: int, world: int) -> List:
    lo, hi = shard_range(len(items), rank, world)
    return items[lo:hi]


def broadcast_blob(path_on_rank0: str, device, rank: int, world: int):
    """Return a uint8 torch tensor on `device` holding the weight blob on every rank.

    Rank 0 reads the file; the other ranks receive it through one broadcast (two tiny
    broadcasts of the size, then the payload)."""
    import torch
    import torch.distributed as dist

    if rank == 0:
        host = torch.from_numpy(np.fromfile(path_on_rank0, dtype=np.uint8))
        n = torch.tensor([host.numel()], dtype=torch.int64, device=device)
    else:
        host = None
        n = torch.zeros(1, dtype=torch.int64, device=device)
    gloo = world > 1 and dist.get_backend() == "gloo"
    if world > 1:
        if gloo:  # CPU rehearsal backend: collectives on host tensors
            n_h = n.cpu()
            dist.broadcast(n_h, src=0)
            n = n_h
        else:
            dist.broadcast(n, src=0)
    if gloo:
        buf_h = host if rank == 0 else torch.empty(int(n.item()), dtype=torch.uint8)
        dist.broadcast(buf_h, src=0)
        return buf_h.to(device)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == 0:
        buf.copy_(host)
    if world > 1:
        dist.broadcast(buf, src=0)
    return buf


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))
