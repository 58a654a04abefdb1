"""Measurement tool (not collected by pytest): the other BASELINE.json configurations, at their shapes, through the C ABI.

    python tests/config_runs.py --config cfg1|cfg2|cfg3|cfg5 [--mode M] [--reads N] [--out gpurun_out/x.json]

cfg1  1000 x 150 bp vs one 5 kb contig, local               (every read also checked against the oracle)
cfg2  10 kb reads vs 50 x 5 kb, single strand, in the clipping mode --mode names (local: what bench.py measures; query-local,
      target-local, global: the 32-bit register-resident kernel, fill_regs32.hip)
cfg3  10 kb reads vs 50 x 5 kb, --double-strand --pre-align (k=12, w=50, s=100, subset)
cfg5  20 kb PacBio-like reads vs 200 x 5 kb circular, --circular --suboptimal

Besides the rate, every chain of cfg1 and cfg3 is checked for the size-independent property the full-size parity test
uses: the score recomputed from the chain's operation list (A=1 B=-4 O=-6 E=-2 J=-10) equals the reported score (not
cfg5: suboptimal chains and chains re-aligned across the origin are pieces of longer paths and carry the reference's
own sub-scores; that configuration's parity case is tests/test_gpu_parity.py::test_cfg5_shape_*).  One JSON line per run."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rescore(ops):
    """Score of an op array (uint64 records: kind in the low byte) under the CLI's default scoring, vectorised."""
    kind = (ops & 0xFF).astype(np.int64)
    sc = int((kind == 0).sum()) - 4 * int((kind == 1).sum()) - 10 * int((kind == 6).sum())
    for g in (2, 3):
        is_g = kind == g
        starts = is_g & ~np.concatenate(([False], is_g[:-1]))
        sc += -2 * int(is_g.sum()) - 6 * int(starts.sum())
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=["cfg1", "cfg2", "cfg3", "cfg5"])
    ap.add_argument("--mode", default="local", choices=["local", "query-local", "target-local", "global"])
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import stitch_amd
    from stitch_amd import synth

    if args.config == "cfg1":
        db = synth.make_db(50, 5000, 1001)[:1]
        n_reads = args.reads or 1000
        reads = synth.make_reads(db, n_reads, 150, 43, max_segments=1)
        opts = {}
        batch = args.batch or 1000
    elif args.config == "cfg2":
        db = synth.make_db(50, 5000, 1001)
        n_reads = args.reads or (320 if args.mode == "local" else 160)
        reads = synth.make_reads(db, n_reads, 10000, 44)
        opts = dict(mode=args.mode)
        batch = args.batch or (160 if args.mode == "local" else 80)       # four launches per call: 40 reads x 50 contigs fill fill_regs' 2048 wave slots, 20 reads fill_regs32's 1024
    elif args.config == "cfg3":
        db = synth.make_db(50, 5000, 1001)
        n_reads = args.reads or 2048
        reads = synth.make_reads(db, n_reads, 10000, 45, both_strands=True)
        opts = dict(double_strand=True, pre_align=True, kmer_size=12, band_width=50, pre_align_min_score=100, pre_align_subset_contigs=True)
        batch = args.batch or 1024                 # what stitch-align hands over per call
    else:
        db = synth.make_db(200, 5000, 1002)
        n_reads = args.reads or 36
        reads = synth.make_reads(db, n_reads, 20000, 47, sub=0.01, ins=0.005, dele=0.005, circular=True)
        opts = dict(circular=True, suboptimal=True)
        batch = args.batch or 36                   # 200 contigs = 50 workgroups per read: six reads per launch (120 GB of traceback), two launches in flight
    targets = [stitch_amd.TargetSeq(n, s) for n, s in db]
    al = stitch_amd.Builder(**opts).build_aligners(targets)

    def pack(chunk):
        offs = np.zeros(len(chunk) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(r) for r in chunk])
        return np.frombuffer(b"".join(chunk), dtype=np.uint8), offs

    al.align_packed_raw(*pack(reads[:min(batch, len(reads))]))                   # warm-up: first launch, arena allocation at its final size
    tot = dict(fill_ms=0.0, walk_ms=0.0, prealign_ms=0.0, prealign_host_ms=0.0, h2d_ms=0.0, d2h_ms=0.0, cells=0, launches=0)
    n_chains = n_bad = mapped = 0
    import hashlib
    digest = hashlib.sha256()                     # of every result record and operation: two runs of one configuration (persistent teams or launch by launch) must agree
    t0 = time.perf_counter()
    t_check = 0.0
    for s in range(0, len(reads), batch):
        rr, ch, op = al.align_packed_raw(*pack(reads[s:s + batch]))
        tm = al.timing()
        for k in tot:
            tot[k] += tm.get(k, 0)
        tc = time.perf_counter()
        digest.update(np.ascontiguousarray(rr).tobytes()); digest.update(np.ascontiguousarray(ch).tobytes()); digest.update(np.ascontiguousarray(op).tobytes())
        opw = op.view(np.uint64) if len(op) else np.zeros(0, dtype=np.uint64)
        for c in ch:
            n_chains += 1
            if args.config != "cfg5" and int(c["ops_len"]) and rescore(opw[int(c["ops_begin"]):int(c["ops_begin"]) + int(c["ops_len"])]) != int(c["score"]):
                n_bad += 1
        mapped += int((rr["n_chains"] > 0).sum())
        t_check += time.perf_counter() - tc
    dt = time.perf_counter() - t0 - t_check
    out = {"config": args.config, "mode": args.mode, "fill_kind": al.timing().get("fill_kind"), "reads": len(reads), "batch": batch, "seconds": dt, "reads_per_sec": len(reads) / dt,
           "gcells_per_sec": tot["cells"] / dt / 1e9, "chains": n_chains, "chains_whose_ops_do_not_rescore": None if args.config == "cfg5" else n_bad,
           "reads_with_chains": mapped, "results_sha256": digest.hexdigest()[:16], "stream_runs": al.timing().get("stream_runs"), "fallbacks": al.timing().get("fallbacks"), "teams_retired": al.timing().get("teams_retired"), **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in tot.items()}}

    if args.config == "cfg1":                                                  # small enough for the oracle: full comparison
        from oracle import oracle as orc
        o = orc.Aligners([(n, s.decode()) for n, s in db])
        res = al.align(reads)
        t1 = time.perf_counter()
        diff = 0
        for r, (got, _) in zip(reads, res):
            want = o.align(r.decode())
            diff += [c.key() for c in got] != [c.key() for c in want]
        out["oracle_seconds_1_core"] = time.perf_counter() - t1
        out["reads_differing_from_oracle"] = diff
    line = json.dumps(out)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return 1 if (n_bad or out.get("reads_differing_from_oracle", 0)) else 0


if __name__ == "__main__":
    sys.exit(main())
