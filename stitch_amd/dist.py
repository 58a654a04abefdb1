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
    identical reads (FastxGroupingIterator, align/io.rs:118-146) stay on one rank; blocks concatenate in rank order.
    The rule lives in the library (stitch_shard_range, include/stitch_gpu.h): the command-line front end uses the same."""
    import ctypes as C
    from .api import lib, _check
    seqs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    cat = b"".join(seqs)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs])
    buf = (C.c_uint8 * max(1, len(cat))).from_buffer_copy(cat or b"\0")
    lo, hi = C.c_uint32(0), C.c_uint32(0)
    _check(lib().stitch_shard_range(buf, offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(seqs)), C.c_uint32(world), C.c_uint32(rank), C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)
