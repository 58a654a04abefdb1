#!/usr/bin/env python3
"""End-to-end rate of the stand-alone front end at the cfg2 shape: FASTQ + FASTA files in, BAM on stdout out (what a user of `stitch
align` runs), next to the rate of the library call alone that bench.py reports.  One JSON line.

    python tools/e2e_rate.py [--reads 2560] [--batch 640] [--out gpurun_out/e2e.json]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2560)
    ap.add_argument("--batch", type=int, default=640)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from stitch_amd import build, synth
    cli = build.build_cli()
    db = synth.make_db(50, 5000, 1001)
    reads = synth.make_reads(db, args.reads, 10000, 44)
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, "ref.fa"); fq = os.path.join(d, "reads.fq")
        with open(ref, "w") as f:
            for n, s in db:
                f.write(f">{n}\n{s.decode()}\n")
        with open(fq, "w") as f:
            for k, r in enumerate(reads):
                f.write(f"@read_{k:07d}\n{r.decode()}\n+\n{'I' * len(r)}\n")
        res = {}
        for fmt in ("bam", "sam"):
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-f", fq, "-r", ref, "--batch", str(args.batch), "--output-format", fmt], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if args.out:
                open(args.out + f'.{fmt}.stderr', 'wb').write(r.stderr)
            if r.returncode != 0:
                raise SystemExit(r.stderr.decode()[-2000:])
            line = [l for l in r.stderr.decode().splitlines() if l.startswith("stitch-align:") and "reads/s" in l][-1]
            m = re.search(r"(\d+) reads, ([\d.]+) Gcells in ([\d.]+) s = ([\d.]+) reads/s end to end \(reader ([\d.]+) s, device calls ([\d.]+) s, formatter \+ writer ([\d.]+) s, side by side; first call ([\d.]+) s for (\d+) reads, the others ([\d.]+) reads/s", line)
            res[fmt] = {"process_seconds": dt, "process_reads_per_sec": args.reads / dt, "output_bytes": len(r.stdout),
                        "loop_seconds": float(m.group(3)), "loop_reads_per_sec": float(m.group(4)), "reader_s": float(m.group(5)),
                        "device_calls_s": float(m.group(6)), "formatter_writer_s": float(m.group(7)), "device_calls_reads_per_sec": args.reads / float(m.group(6)),
                        "first_call_s": float(m.group(8)), "device_calls_after_the_first_reads_per_sec": float(m.group(10)),
                        "loop_after_first_call_reads_per_sec": (args.reads - int(m.group(9))) / max(1e-9, float(m.group(3)) - float(m.group(8)))}
        out = {"what": "stitch-align (stitch_amd/cli) on files, cfg2 shape: 10 kb reads vs 50 x 5 kb, local, single strand", "reads": args.reads, "batch": args.batch,
               "note": "process = fork to exit incl. FASTA load, index, context and the first call's arena allocation (~5 s); loop = first read parsed to last record written; "
                       "device calls = time inside stitch_align_batch (what bench.py times), with reader and writer threads running beside it", **res}
        line = json.dumps(out)
        print(line)
        if args.out:
            with open(args.out, "w") as f:
                f.write(line + "\n")


if __name__ == "__main__":
    main()
