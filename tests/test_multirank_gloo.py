"""N > 1 path on the CPU: two gloo ranks broadcast the reference index and shard the read stream exactly as bench.py
does over RCCL.  (The DP itself needs a GPU; here the blob, the sharding and the result order are checked.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stitch_amd
    from stitch_amd import dist as sdist
    from stitch_amd import synth
    db = synth.make_db(5, 300, 1001)
    index = stitch_amd.Index.from_targets([stitch_amd.TargetSeq(n, s) for n, s in db]) if rank == 0 else None
    got = sdist.broadcast_index(index, dist, torch.device("cpu"), src=0)
    ref = stitch_amd.Index.from_targets([stitch_amd.TargetSeq(n, s) for n, s in db]).serialize()
    reads = synth.make_reads(db, 101, 80, 7, dup_every=10)
    lo, hi = sdist.shard_range(reads, world, rank)
    spans = [None] * world
    dist.all_gather_object(spans, (lo, hi))
    q.put((rank, got.serialize() == ref, got.n_contigs(), spans, reads[lo - 1] != reads[lo] if 0 < lo < len(reads) else True))
    dist.destroy_process_group()


def test_two_ranks_broadcast_and_shard():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, nc, spans, clean_cut in out:
        assert same and nc == 5 and clean_cut
        assert spans[0][0] == 0 and spans[-1][1] == 101 and spans[0][1] == spans[1][0]      # blocks tile the stream in rank order


def test_shard_range_keeps_duplicate_runs_together():
    from stitch_amd.dist import shard_range
    reads = [b"A", b"B", b"B", b"B", b"C", b"D"]
    assert [shard_range(reads, 2, r) for r in range(2)] == [(0, 4), (4, 6)]
    assert [shard_range(reads, 3, r) for r in range(3)] == [(0, 4), (4, 4), (4, 6)]
    assert [shard_range([b"x"] * 5, 2, r) for r in range(2)] == [(0, 5), (5, 5)]


def test_shard_range_blocks_tile_the_stream_for_any_world():
    """the library's rule (stitch_shard_range): blocks are contiguous, in rank order, cover the stream, and no run of identical
    reads is cut — for every world size, including more ranks than reads"""
    import random
    from stitch_amd.dist import shard_range
    rng = random.Random(3)
    for trial in range(30):
        reads = []
        while len(reads) < rng.randint(0, 40):
            r = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 6)))
            reads += [r] * rng.randint(1, 4)
        for world in (1, 2, 3, 5, 8, 50):
            spans = [shard_range(reads, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == len(reads)
            assert all(spans[k][1] == spans[k + 1][0] for k in range(world - 1)) and all(lo <= hi for lo, hi in spans)
            for lo, hi in spans:
                assert lo == 0 or lo == len(reads) or reads[lo] != reads[lo - 1]
