#!/bin/bash
# round 4, experiment 2: persistent teams — parity tests, then the cfg2 bench with and without them
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_stream.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/log.txt; tail -5 $O/tests.log | tee -a $O/log.txt
run() { tag=$1; shift; echo "== $tag" | tee -a $O/log.txt; env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-reads 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY | tee -a $O/log.txt
import json
try:
    d=json.load(open("$O/$tag.json")); r=d["roofline"]
    print("$tag", round(d["value"],1), "reads/s", "busy/launch", round(r["fill_busy_ms_per_launch"],1), "avg launch", round(r["avg_launch_ms"],1), "frac", round(r["frac"],4), "fallbacks", r.get("fill_fallbacks"))
except Exception as e: print("$tag failed", e)
PY
}
run stream STITCH_TRACE=1 && run nostream STITCH_NO_STREAM=1 && run stream640 STITCH_X=1
grep -h "\[trace\]" $O/stream.err | tail -60 > $O/trace_stream.txt
