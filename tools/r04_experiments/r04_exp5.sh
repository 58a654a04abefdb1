#!/bin/bash
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
run() { tag=$1; shift; echo "== $tag" | tee -a $O/log.txt; env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-reads 0 $EXTRA > $O/$tag.json 2> $O/$tag.err; python - <<PY | tee -a $O/log.txt
import json
try:
    d=json.load(open("$O/$tag.json")); r=d["roofline"]
    print("$tag", round(d["value"],1), "reads/s", "ms/step", round(d["ms_per_step"],1), "avg launch", round(r["avg_launch_ms"],1), "frac", round(r["frac"],4), "fallbacks", r.get("fill_fallbacks"))
except Exception as e: print("$tag failed", e)
PY
}
run base STITCH_X=1 && run xcdmap STITCH_REGS_MAP=2 && run xcdmap_plain STITCH_REGS_MAP=2 STITCH_EXP_PLAIN_GRANULES=1
