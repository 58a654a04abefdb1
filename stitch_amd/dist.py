"""Multi-GPU plumbing of the hot path: one process per GPU, reads sharded by rank, the reference index broadcast once
(RCCL on GPUs, gloo in the CPU tests).  There is no collective on the data path itself."""
import numpy as np

from .api import Index


def broadcast_index(index, dist, device, src=0):
    """Rank `src` passes its Index; every rank gets an equal Index back (blob broadcast as bytes)."""
    import torch
    rank = dist.get_rank()
    blob = index.serialize() if rank == src else b""
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == src:
        t.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(t, src=src)
    return Index.deserialize(bytes(t.cpu().numpy().tobytes()))


def shard_range(reads, world, rank):
    """Contiguous block [lo, hi) of the read stream for `rank`, cut only where consecutive reads differ so that runs of
    identical reads (FastxGroupingIterator, align/io.rs:118-146) stay on one rank; blocks concatenate in rank order."""
    n = len(reads)
    cuts = [0]
    for r in range(1, world):
        k = (n * r) // world
        while 0 < k < n and reads[k] == reads[k - 1]:
            k += 1
        cuts.append(max(k, cuts[-1]))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]
