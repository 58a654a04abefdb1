#!/usr/bin/env python3
"""Benchmark of the `stitch align` hot path on MI355X (BASELINE.json: reads/sec + DP Gcells/sec on 10 kb synthetic
ONT reads vs a 50 x 5 kb construct DB; configs[1]).

  python bench.py --gpus N --steps K --warmup W [--reads-per-step R]

A step = one pass of the hot path (stitch_align_batch: DP fill, per-column jump reduce, fix-ups, traceback, chain
assembly) over one batch of R synthetic reads per GPU.  ONE synthetic read stream is generated (the same on every
rank) and each step's N*R reads are sharded by rank at read-group boundaries (stitch_amd.dist.shard_range == the
reference's FastxGroupingIterator rule), weak scaling: R per GPU is fixed.  The only collective on the data path is the
one-time broadcast of the serialized reference index from rank 0 (RCCL), outside the timed region.  Rank 0 prints ONE
JSON line.

roofline: the dominant kernel is the DP fill.  `achieved` = algorithmic bytes (1 byte of traceback per DP cell,
SURVEY.md 8d) / the kernel's launch time measured inside the library with HIP events on the stream it runs on
(stitch_last_timing).  `traffic` and the wave-time split come from the committed rocprofv3 --pmc passes of the SAME
kernel sources (profiles/collect.sh records a hash of them; when it differs from the sources in this tree the figures are
left out as stale).  cpu_baseline: the oracle (C++ restatement of the reference, "port") timed on this host with the
reference's worker model on a bounded sample of the same workload, rank 0 at N=1 only, SAM text diffed against the HIP path.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHADER_CLOCK_HZ = 2.4e9    # MI355X_MICROARCH.md "Chip-level parameters": max clock 2400 MHz
# A conventional yardstick, NOT a ceiling: PMC instruction counts against one wave64 instruction per SIMD and
# four cycles.  profiles/ubench/valu_issue_mi355x.txt (with control rows): v_fma_f32 / v_add_f32 / v_add_u32 / v_and_b32 / v_mov_b32
# issue at 2.4-2.7 cycles per SIMD with two or more waves resident (the guide's SIMD-32 figure), v_max / v_cndmask / three-operand and
# packed instructions at 4.2-4.5, scalar instructions per wave rather than per SIMD, and a lone wave at one instruction per 5 cycles
# whatever the class.  The fill is bound by the latency between a wave's dependent instructions (a wave issues one per ~9 cycles).
VALU_CYCLES_PER_WAVE_INST = 4.0
FILL_KERNELS = {0: "stitch::fill_kernel", 1: "stitch::fill_local16_kernel", 2: "stitch::fill_regs_kernel", 3: "stitch::fill_regs32_kernel"}      # stitch_timing.fill_kind
KERNEL_SOURCES = ("fill_local16.hip", "fill_regs.hip", "dp_core.h", "walk_core.h", "stitch_api.cpp")


def kernel_src_hash():
    """Hash of the sources that decide what the fill kernel does and how it is launched (profiles/collect.sh stores it)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        p = os.path.join(ROOT, "stitch_amd", "csrc", f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def host_memory_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=640, help="reads per GPU per step.  The register-resident fill keeps 40 teams (40 reads x 50 contigs = 2000 of the chip's 2048 wave slots) resident for the whole step; a team that ends a read pulls the next one off a queue, the host walks finished reads and recycles their arena blocks meanwhile: 640 reads = 16 reads per team")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--contigs", type=int, default=50)
    ap.add_argument("--contig-len", type=int, default=5000)
    ap.add_argument("--cpu-reads", type=int, default=-1, help="0 = skip the cpu_baseline leg")
    ap.add_argument("--cpu-reads-per-worker", type=int, default=2, help="timed reads per worker thread of the CPU leg (>= 2; one more is aligned as warm-up)")
    ap.add_argument("--cpu-prefix", type=int, default=1000, help="bases of each sample read the CPU aligns (0 = WHOLE reads: 40 GB of 16-byte cells per worker and ~100 s per read and thread; the default keeps the CPU leg at about a minute; profiles/r04_cpu_whole_reads.json holds a whole-read run)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="worker threads of the CPU leg (0 = min(cores, free RAM / RAM per worker, CPU share of the box))")
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

    # ---- ONE synthetic read stream (seed 42 + config id 2), every step's world * R reads sharded by rank -------
    R = args.reads_per_step
    total_steps = args.warmup + args.steps
    stream = synth.make_reads(db, R * world * total_steps, args.read_len, 44)
    batches, my_reads = [], []
    for s in range(total_steps):
        step_reads = stream[s * R * world:(s + 1) * R * world]
        lo, hi = sdist.shard_range(step_reads, world, rank)
        chunk = step_reads[lo:hi]
        offs = np.zeros(len(chunk) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(r) for r in chunk])
        batches.append((np.frombuffer(b"".join(chunk), dtype=np.uint8), offs))
        my_reads.append(len(chunk))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        aligners.align_packed_raw(*batches[s])
    sync()
    t0 = time.perf_counter()
    fill_ms = walk_ms = fill_kernel_ms = 0.0
    clk_cycles = clk_ticks = 0
    cells = 0
    launches = 0
    fallbacks = 0
    retired = 0
    stream_runs = 0
    mapped = 0
    n_mine = 0
    kernel_name = None
    for s in range(args.warmup, total_steps):
        rr, ch, _ops = aligners.align_packed_raw(*batches[s])       # result arena views: what a compiled front end would read
        tm = aligners.timing()
        fill_ms += tm["fill_ms"]; walk_ms += tm["walk_ms"]; cells += tm["cells"]; launches += tm["launches"]; fill_kernel_ms += tm.get("fill_kernel_ms", tm["fill_ms"])
        clk_cycles += tm.get("clk_shader_cycles", 0); clk_ticks += tm.get("clk_ref_ticks", 0); fallbacks += tm.get("fallbacks", 0); retired += tm.get("teams_retired", 0); stream_runs += tm.get("stream_runs", 0)
        kernel_name = FILL_KERNELS.get(tm.get("fill_kind", 1), kernel_name)
        n_mine += my_reads[s]
        if len(ch):
            mapped += int((ch["score"][rr["chains_begin"][rr["n_chains"] > 0]] >= 100).sum())
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        agg = torch.tensor([float(cells), float(n_mine)], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        cells_all, n_reads_all = float(agg[0].item()), int(round(float(agg[1].item())))
    else:
        cells_all, n_reads_all = float(cells), n_mine

    if rank == 0:
        value = n_reads_all / dt
        kernel_name = kernel_name or "stitch::fill_local16_kernel"
        prop = torch.cuda.get_device_properties(local_rank)
        # roofline of the dominant kernel (this rank's launches; every rank runs the same kernel on the same shape)
        fill_s = fill_ms / 1e3
        # `achieved` is anchored on the WALL time of the timed steps: this rank's algorithmic bytes (1 per cell) / dt.
        # It needs no builder-side timer, includes walks, copies and host work between launches, and is what the driver's clock
        # around the run can check.  The busy-union figure (bytes / time during which a fill kernel was running) is kept as
        # `achieved_fill_busy`, the per-dispatch one as `achieved_per_dispatch` (x launches_in_flight = the busy figure).
        achieved = (cells * 1.0 / dt) / 1e9 if dt > 0 else 0.0                     # GB/s at 1 algorithmic byte per cell
        achieved_busy = (cells * 1.0 / fill_s) / 1e9 if fill_s > 0 else 0.0
        achieved_disp = (cells * 1.0 / (fill_kernel_ms / 1e3)) / 1e9 if fill_kernel_ms > 0 else 0.0
        cells_per_read = args.read_len * args.contigs * args.contig_len
        out = {
            "metric": "reads_per_sec", "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 scores and lengths packed in int32 words", "data": "synthetic",
            "config": {"workload": f"{args.read_len} bp chimeric ONT-like reads vs {args.contigs}x{args.contig_len} bp construct DB, "
                                   "local mode, single strand (BASELINE configs[1])",
                       "reads_per_step_per_gpu": R, "cells_per_read": args.read_len * args.contigs * args.contig_len,
                       "scoring": "A=1 B=-4 O=-6 E=-2 J=-10", "sharding": "one read stream cut by rank at read-group boundaries, index broadcast once"},
            "gcells_per_sec": cells_all / dt / 1e9,
            # `value` counts every input read, also the consecutive duplicates the reference aligns once (FastxGroupingIterator: every 50th
            # read of the synthetic set costs nothing); the rate by DP cells actually filled, in reads of the named shape:
            "reads_per_sec_by_cells": cells_all / dt / cells_per_read,
            "device": {"name": prop.name, "cus": prop.multi_processor_count, "hbm_gib": round(prop.total_memory / 2**30)},
            "mapped_fraction": mapped / float(max(1, n_mine)),
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "achieved_is": "this rank's algorithmic bytes / wall time of the timed steps",
                         "achieved_fill_busy": achieved_busy, "frac_fill_busy": achieved_busy / HBM_PEAK_GBS,
                         "achieved_per_dispatch": achieved_disp, "frac_per_dispatch": achieved_disp / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_cell": 1.0, "cells_per_launch": cells / max(1, launches),
                         # two launches are in flight (the next one takes the workgroup slots that finished reads of the current one free): a
                         # kernel's own duration (what rocprofv3 lists per dispatch) is longer than the time the launch adds to the job.
                         # `achieved` = algorithmic bytes / time during which the fill kernel was running (no double counting).
                         "avg_launch_ms": fill_kernel_ms / max(1, launches), "fill_busy_ms_per_launch": fill_ms / max(1, launches),
                         "launches_in_flight": fill_kernel_ms / fill_ms if fill_ms > 0 else 1.0, "walk_kernel_ms_per_step": walk_ms / args.steps,
                         "fill_gcells_per_sec": cells / fill_s / 1e9 if fill_s > 0 else 0.0,
                         # launches repeated on a slower kernel after a partner timeout (co-residency lost): must be 0 in a healthy run
                         "fill_fallbacks": fallbacks, "teams_retired": retired,
                         # how the fill was launched: one dispatch per step whose teams pull reads off a queue ("persistent teams"), or launch by
                         # launch with two fills in flight (STITCH_NO_STREAM=1, and always under rocprofv3, which reports a launch beside
                         # resident teams complete only when they are: DESIGN.md 4)
                         "launch_mode": "persistent teams" if stream_runs else "launch by launch", "persistent_runs": stream_runs},
        }
        # the shader clock the fill ran at, measured inside the kernel over the timed launches (s_memtime against the 100 MHz
        # s_memrealtime; fill_regs.hip).  The issue roofline below is priced at the nominal 2.4 GHz AND at this clock.
        clock_mhz = clk_cycles / clk_ticks * 100.0 if clk_ticks else None
        if clock_mhz:
            out["roofline"]["clock"] = {"in_kernel_mhz": clock_mhz, "nominal_mhz": SHADER_CLOCK_HZ / 1e6,
                                        "how": "s_memtime / s_memrealtime over the column loop of every timed fill launch; profiles/*_clock.txt holds the smi and PMC views"}
        # HBM-side traffic and the wave-time split of the fill kernel: PMC counters cannot be read from inside this process, so
        # the figures are the committed rocprofv3 --pmc measurement (profiles/, separate FETCH_SIZE and WRITE_SIZE passes of this
        # same command, KB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled per cell — and only
        # when that measurement was taken on the kernel sources of this tree.
        try:
            src = kernel_src_hash()
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")))
            pmc_path, pmc = None, None
            for p in reversed(cands):
                d = json.load(open(p))
                if d.get("kernel_src_sha") == src and kernel_name in d:
                    pmc_path, pmc = p, d
                    break
            if pmc is None:
                out["roofline"]["traffic_source"] = (f"no committed PMC pass matches the kernel sources of this tree (hash {src}): "
                                                     "traffic and the wave-time split are left out rather than reported stale")
            else:
                k = pmc[kernel_name]
                cpl = pmc["cells_per_launch"]
                bpc = (2.0 * k["FETCH_SIZE"]["avg_per_launch_raw"] + k["WRITE_SIZE"]["avg_per_launch_raw"]) * 1024.0 / cpl
                out["roofline"]["traffic"] = bpc * cells / max(1, launches)
                out["roofline"]["traffic_bytes_per_cell"] = bpc
                # the bound that binds ("What bounds the fill"): integer VALU issue, against CUs x 4 SIMDs x 2.4 GHz / 4 cycles
                # per wave64 instruction.  The peak is priced at the nominal clock; under this kernel the chip runs slower.
                vpc = k["SQ_INSTS_VALU"]["avg_per_launch_raw"] * 64.0 / cpl
                peak = prop.multi_processor_count * 4 * SHADER_CLOCK_HZ / VALU_CYCLES_PER_WAVE_INST
                out["roofline"]["valu"] = {"wave_insts_per_64_cells": vpc, "achieved_wave_insts_per_s": vpc * (cells / 64.0) / fill_s,
                                           "peak_wave_insts_per_s": peak, "frac": vpc * (cells / 64.0) / fill_s / peak,
                                           "peak_source": "yardstick: CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (profiles/ubench/valu_issue_mi355x.txt: 2.4-2.7 cycles for add / and / mov / fma, 4.2-4.5 for max / select / three-operand)"}
                # ... and every other instruction takes an issue turn of its SIMD as well (two waves per SIMD rarely issue side by side:
                # both mostly want the vector pipe): all instructions of the kernel against the same one-per-four-cycles peak
                names = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
                if all(n_ in k for n_ in names):
                    ipc = sum(k[n_]["avg_per_launch_raw"] for n_ in names) * 64.0 / cpl
                    out["roofline"]["issue"] = {"wave_insts_per_64_cells": ipc, "achieved_wave_insts_per_s": ipc * (cells / 64.0) / fill_s,
                                                "peak_wave_insts_per_s": peak, "frac": ipc * (cells / 64.0) / fill_s / peak,
                                                "what": "vector + scalar + LDS + memory instructions (PMC) against one issue turn per SIMD and four cycles at 2.4 GHz"}
                    if clock_mhz:
                        peak_m = prop.multi_processor_count * 4 * clock_mhz * 1e6 / VALU_CYCLES_PER_WAVE_INST
                        out["roofline"]["issue"]["frac_at_measured_clock"] = ipc * (cells / 64.0) / fill_s / peak_m
                        out["roofline"]["valu"]["frac_at_measured_clock"] = vpc * (cells / 64.0) / fill_s / peak_m
                out["roofline"]["binding"] = ("per-wave instruction latency: neither HBM (1.09 B/cell at the fabric) nor the vector pipe (about 55 % busy) is "
                                              "saturated; valu / issue below are PMC counts against a conventional one-instruction-per-4-cycles yardstick, not a ceiling")
                if "SQ_WAVE_CYCLES" in k and "SQ_WAIT_ANY" in k:
                    wc = k["SQ_WAVE_CYCLES"]["avg_per_launch_raw"]
                    out["roofline"]["wave_time_split"] = {n_: k[c_]["avg_per_launch_raw"] / wc for n_, c_ in
                                                          (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY")) if c_ in k}
                out["roofline"]["traffic_source"] = (f"profiles/{os.path.basename(pmc_path)} (kernel sources {src}): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch of "
                                                     f"{cpl:.3g} cells = {bpc:.2f} B/cell, scaled to this run's cells per launch")
        except (OSError, KeyError, ValueError, IndexError, TypeError, ZeroDivisionError):
            pass
        if world == 1 and args.cpu_reads != 0:
            out["cpu_baseline"] = cpu_leg(args, db, stream, aligners)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_leg(args, db, stream, aligners):
    """the C++ restatement of the reference with the reference's worker model — T threads, one aligner set
    (16-byte row-major traceback matrices of every contig) per thread — on a bounded sample of the same read stream, WARM: every
    worker builds its aligner set and aligns one read before the clock starts (the reference allocates a thread's matrices once,
    align/traceback/mod.rs:93-126; first-touch page faults are a property of a fresh process), the clock stops before anything is
    freed, and each worker aligns >= 2 timed reads.  T = min(cores, free RAM / RAM per worker, the box's CPU share); the scaling
    from one thread up to T is measured and printed.  SAM text from both sides is diffed on the sample."""
    from oracle import oracle as orc
    native = orc.native_lib() is not None          # g++ -O3 -march=native on THIS host; the portable build otherwise
    pre = args.cpu_prefix if 0 < args.cpu_prefix < args.read_len else args.read_len
    rows = args.contigs * (args.contig_len + 1)
    ram_per_worker = rows * (pre + 1) * 16                      # traceback/mod.rs:122-126
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    free = host_memory_available()
    # T = min(cores, free RAM / per-worker RAM); a one-GPU box lends this job a CPU share of 16 (gpurun)
    share = int(os.environ.get("STITCH_CPU_SHARE", "16"))
    T = args.cpu_threads if args.cpu_threads > 0 else max(1, min(cores, int(free * 0.6 // max(1, ram_per_worker)), share))
    per_worker = max(2, args.cpu_reads_per_worker)
    # distinct reads (a duplicated neighbour is the same job for both sides), each cut to the stated prefix
    cut = []
    for i, r in enumerate(stream):
        if i == 0 or r != stream[i - 1]:
            cut.append(r[:pre])
        if len(cut) == per_worker * T:
            break
    targets = [(n, s) for n, s in db]
    cells_per_read = args.read_len * args.contigs * args.contig_len
    # scaling: one thread, a few, all — each level times `per_worker` reads per worker after one warm-up read per worker
    levels = sorted({1, T})
    scaling = []
    secs = ccells = 0
    csam = busy = None
    for t in levels:
        sample = cut[:per_worker * t]
        secs, ccells, _scores, csam, busy = orc.cpu_bench_sam(targets, sample, threads=t, name_base=0, chunk=1, warm=1, native=True)
        scaling.append({"threads": t, "reads": len(sample), "seconds": round(secs, 2), "mcells_per_sec": round(ccells / secs / 1e6, 1),
                        "mcells_per_sec_per_thread": round(ccells / secs / 1e6 / t, 1)})
    sample = cut[:per_worker * T]
    # the same sample through the HIP path: SAM text must be identical (parity proper is tests/, this is the run-time diff)
    aligners.align(sample)
    gsam = ["".join(l + "\n" for l in aligners.format_sam(k, f"read_{k:07d}", sample[k], b"I" * len(sample[k]))) for k in range(len(sample))]
    same = [g == c for g, c in zip(gsam, csam)]
    per_thread = ccells / secs / 1e6 / T
    eff = scaling[-1]["mcells_per_sec"] / max(1e-9, scaling[0]["mcells_per_sec"] * T)
    why = "" if eff >= 0.7 else (f"; {T} threads reach {eff:.0%} of {T} x the one-thread rate: every cell is a 16-byte store into a row-major "
                                 f"(m+1) x (n+1) matrix (a new cache line and page per row), so the workers contend for memory bandwidth and TLB reach, not for cores")
    return {"value": ccells / secs / cells_per_read, "unit": "reads/s", "cores": T, "kind": "port", "warm": True,
            "gcells_per_sec": ccells / secs / 1e9, "mcells_per_sec_per_thread": per_thread, "scaling": scaling,
            "sam_identical_on_sample": bool(all(same)), "sam_reads_compared": len(same),
            "sample": f"{'WHOLE reads: the' if pre == args.read_len else 'first ' + str(pre) + ' bp of the'} first {len(sample)} distinct reads vs the full DB, {per_worker} timed reads per worker after one warm-up read per "
                      f"worker (matrices allocated and touched before the clock; the clock stops before they are freed): {ccells} cells in {secs:.1f} s on {T} "
                      f"worker threads = {per_thread:.1f} Mcells/s per thread (one aligner set per thread, one record per pull; {cores} cores visible, CPU share "
                      f"of the box {share}, {free / 2**30:.0f} GiB host RAM available, {ram_per_worker / 2**30:.1f} GiB of 16-byte traceback cells per worker; "
                      f"T = min(cores, RAM / per-worker RAM, share)); scaling {', '.join(str(x['threads']) + ' thr: ' + str(x['mcells_per_sec']) + ' Mcells/s' for x in scaling)}"
                      f"{why}; reads/s = cells/s / {cells_per_read} cells per {args.read_len} bp read; C++ restatement of fulcrumgenomics/stitch "
                      f"({'g++ -O3 -march=native, built on this host' if native else 'g++ -O3, portable flags'}), not the Rust binary",
            "build": "g++ -O3 -march=native (this host)" if native else "g++ -O3 (portable)", "whole_reads": pre == args.read_len}


if __name__ == "__main__":
    main()
