#!/bin/bash
# VALU instructions of one fill launch (deterministic: the objective measure for instruction-count changes).
#   bash profiles/valu_count.sh  -> prints SQ_INSTS_VALU of stitch::fill_local16_kernel and per cell
export TMPDIR=/tmp
d=gpurun_out/valu_$$
rocprofv3 --pmc SQ_INSTS_VALU -d "$d" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-reads 0 > "$d.log" 2>&1 || { echo "pmc run failed"; tail -5 "$d.log"; exit 1; }
python3 - "$d" <<'PY'
import csv, glob, sys
tot = 0.0
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "fill_local16" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            tot += float(r["Counter_Value"])
print("SQ_INSTS_VALU", tot, "per cell x64:", tot * 64 / 1.575e11)
PY
