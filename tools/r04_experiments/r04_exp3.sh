#!/bin/bash
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
run() { tag=$1; shift; echo "== $tag" | tee -a $O/log.txt; env "$@" timeout -k 10 200 python bench.py --steps 1 --warmup 0 --cpu-reads 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY | tee -a $O/log.txt
import json
try:
    d=json.load(open("$O/$tag.json")); r=d["roofline"]
    print("$tag", round(d["value"],1), "reads/s", "avg launch", round(r["avg_launch_ms"],1), "fallbacks", r.get("fill_fallbacks"))
except Exception as e: print("$tag failed", e)
PY
grep -h "\[trace\]" $O/$tag.err | head -40 > $O/trace_$tag.txt
}
run t39 STITCH_TRACE=1 STITCH_STREAM_TEAMS=39
run range4 STITCH_TRACE=1 STITCH_STREAM_RANGE=4
run t36 STITCH_TRACE=1 STITCH_STREAM_TEAMS=36
