#!/usr/bin/env python3
"""Benchmark of the `stitch align` hot path on MI355X (BASELINE.json: reads/sec + DP Gcells/sec on 10 kb synthetic
ONT reads vs a 50 x 5 kb construct DB; configs[1]).

  python bench.py --gpus N --steps K --warmup W [--reads-per-step R]

A step = one pass of the hot path (stitch_align_batch: DP fill, per-column jump reduce, fix-ups, traceback, chain
assembly) over one batch of R synthetic reads per GPU.  Reads are sharded by rank (weak scaling: R per GPU is
fixed); the only collective on the data path is the one-time broadcast of the serialized reference index from
rank 0 (RCCL), outside the timed region.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is the DP fill (stitch::fill_kernel).  `achieved` = algorithmic bytes (1 byte of
traceback per DP cell) / the kernel's launch time measured inside the library with HIP events on
the stream it runs on (stitch_last_timing).  cpu_baseline: the oracle (C++ restatement of the reference, "port")
timed on this host on a bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=64, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--contigs", type=int, default=50)
    ap.add_argument("--contig-len", type=int, default=5000)
    ap.add_argument("--cpu-reads", type=int, default=1, help="reads in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--cpu-prefix", type=int, default=1000, help="bases of each sample read the CPU aligns (0 = whole read: ~40 GB, minutes)")
    ap.add_argument("--cpu-threads", type=int, default=1)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on one device, gloo instead of RCCL
        if os.environ.get("STITCH_BENCH_DEVICE") is not None:
            local_rank = int(os.environ["STITCH_BENCH_DEVICE"])
        backend = os.environ.get("STITCH_BENCH_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")

    import stitch_amd
    from stitch_amd import synth

    # ---- reference index: built on rank 0, broadcast once over RCCL as a byte blob -----------------------------
    dev = torch.device("cuda", local_rank)
    from stitch_amd import dist as sdist
    db = synth.make_db(args.contigs, args.contig_len, 1001)
    index = stitch_amd.Index.from_targets([stitch_amd.TargetSeq(n, s) for n, s in db]) if rank == 0 else None
    if world > 1:
        index = sdist.broadcast_index(index, dist, dev, src=0)          # the one collective: RCCL broadcast of the index blob
    aligners = stitch_amd.Aligners(stitch_amd.Builder().build_options(), index, device=local_rank)

    # ---- this rank's shard of the synthetic reads (seed 42 + config id 2; rank-specific stream) ----------------
    R = args.reads_per_step
    total_steps = args.warmup + args.steps
    reads = synth.make_reads(db, R * total_steps, args.read_len, 44 + 1000 * rank)
    batches = []
    for s in range(total_steps):
        chunk = reads[s * R:(s + 1) * R]
        offs = np.zeros(len(chunk) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(r) for r in chunk])
        batches.append((np.frombuffer(b"".join(chunk), dtype=np.uint8), offs))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        aligners.align_packed_raw(*batches[s])
    sync()
    t0 = time.perf_counter()
    fill_ms = walk_ms = 0.0
    cells = 0
    launches = 0
    mapped = 0
    for s in range(args.warmup, total_steps):
        rr, ch, _ops = aligners.align_packed_raw(*batches[s])       # result arena views: what a compiled front end would read
        tm = aligners.timing()
        fill_ms += tm["fill_ms"]; walk_ms += tm["walk_ms"]; cells += tm["cells"]; launches += tm["launches"]
        mapped += int((ch["score"][rr["chains_begin"][rr["n_chains"] > 0]] >= 100).sum())
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        agg = torch.tensor([float(cells), fill_ms, float(launches)], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        cells_all = float(agg[0].item())
    else:
        cells_all = float(cells)

    if rank == 0:
        n_reads_all = R * args.steps * world
        value = n_reads_all / dt
        # roofline of the dominant kernel (this rank's launches; every rank runs the same kernel on the same shape)
        fill_s = fill_ms / 1e3
        achieved = (cells * 1.0 / fill_s) / 1e9 if fill_s > 0 else 0.0           # GB/s at 1 algorithmic byte per cell
        out = {
            "metric": "reads_per_sec", "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 scores and lengths packed in int32 words", "data": "synthetic",
            "config": {"workload": f"{args.read_len} bp chimeric ONT-like reads vs {args.contigs}x{args.contig_len} bp construct DB, "
                                   "local mode, single strand (BASELINE configs[1])",
                       "reads_per_step_per_gpu": R, "cells_per_read": args.read_len * args.contigs * args.contig_len,
                       "scoring": "A=1 B=-4 O=-6 E=-2 J=-10", "sharding": "reads by rank, index broadcast once"},
            "gcells_per_sec": cells_all / dt / 1e9,
            "device": (lambda p: {"name": p.name, "cus": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2**30)})(torch.cuda.get_device_properties(local_rank)),
            "mapped_fraction": mapped / float(R * args.steps),
            "roofline": {"bound": "hbm", "kernel": "stitch::fill_local16_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_cell": 1.0, "cells_per_launch": cells / max(1, launches),
                         "avg_launch_ms": fill_ms / max(1, launches), "walk_kernel_ms_per_step": walk_ms / args.steps,
                         "fill_gcells_per_sec": cells / fill_s / 1e9 if fill_s > 0 else 0.0},
        }
        # HBM-side traffic of the fill kernel: PMC counters cannot be read from inside this process, so the figure is
        # the committed rocprofv3 --pmc measurement (profiles/, separate FETCH_SIZE and WRITE_SIZE passes of this same
        # command, KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) scaled per cell.
        try:
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_local16.json")))[-1]        # the latest committed pass
            pmc = json.load(open(pmc_path))
            k = pmc["stitch::fill_local16_kernel"]
            bpc = (2.0 * k["FETCH_SIZE"]["avg_per_launch_raw"] + k["WRITE_SIZE"]["avg_per_launch_raw"]) * 1024.0 / pmc["cells_per_launch"]
            out["roofline"]["traffic"] = bpc * cells / max(1, launches)
            # second view, as SURVEY.md 8(d) asks: integer VALU issue.  A wave64 integer instruction occupies its SIMD for 4 clocks
            # (16 lanes/clock), so the ceiling is 256 CU x 4 SIMD x 2.4 GHz / 4 wave-instructions/s.
            vpc = k["SQ_INSTS_VALU"]["avg_per_launch_raw"] * 64.0 / pmc["cells_per_launch"]
            out["roofline"]["valu"] = {"wave_insts_per_64_cells": vpc, "achieved_wave_insts_per_s": vpc * (cells / 64.0) / fill_s,
                                       "peak_wave_insts_per_s": 256 * 4 * 2.4e9 / 4, "frac": vpc * (cells / 64.0) / fill_s / (256 * 4 * 2.4e9 / 4)}
            out["roofline"]["traffic_source"] = (f"profiles/{os.path.basename(pmc_path)}: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch of "
                                                 f"{pmc['cells_per_launch']:.3g} cells = {bpc:.2f} B/cell, scaled to this run's cells per launch")
        except (OSError, KeyError, ValueError, IndexError, TypeError):
            pass
        if world == 1 and args.cpu_reads > 0:
            from oracle import oracle as orc
            # One 10 kb read is 2.5e9 cells and 40 GB of 16-byte traceback cells for the reference layout (minutes per read
            # per core), so the bounded sample is a PREFIX of the first read(s) against the full DB; reads/s is scaled by
            # cells (the DP cost is linear in read length).
            pre = args.cpu_prefix if args.cpu_prefix > 0 else args.read_len
            sample = [r[:pre] for r in reads[:args.cpu_reads]]
            secs, ccells, cscores = orc.cpu_bench([(n, s) for n, s in db], sample, threads=args.cpu_threads)
            # the same sample through the HIP path: the two sides must agree (parity proper is tests/, this is a run-time cross-check)
            gres = aligners.align(sample)
            same = all(len(g[0]) > 0 and g[0][0].score == int(cscores[k]) for k, g in enumerate(gres))
            cells_per_read = args.read_len * args.contigs * args.contig_len
            out["cpu_baseline"] = {"value": ccells / secs / cells_per_read, "unit": "reads/s", "cores": args.cpu_threads, "kind": "port",
                                   "gcells_per_sec": ccells / secs / 1e9, "gpu_scores_equal_on_sample": bool(same),
                                   "sample": f"first {pre} bp of the first {len(sample)} read(s) vs the full DB: {ccells} cells in {secs:.1f} s on "
                                             f"{args.cpu_threads} thread(s); reads/s = cells/s / {cells_per_read} cells per {args.read_len} bp read; C++ "
                                             f"restatement of fulcrumgenomics/stitch with its 16-byte row-major traceback cells "
                                             f"({os.cpu_count()} host cores visible)"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
