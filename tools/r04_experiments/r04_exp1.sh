#!/bin/bash
# round 4, experiment 1: launch timeline of the cfg2 bench and the pairing of waves on a SIMD (8-wave workgroups)
set -o pipefail
O=gpurun_out/r4a; mkdir -p $O
run() { tag=$1; shift; echo "== $tag" | tee -a $O/log.txt; env "$@" python bench.py --steps 3 --warmup 1 --cpu-reads 0 > $O/$tag.json 2> $O/$tag.err; python - <<PY | tee -a $O/log.txt
import json
try:
    d=json.load(open("$O/$tag.json")); r=d["roofline"]
    print("$tag", round(d["value"],1), "reads/s", "busy/launch", round(r["fill_busy_ms_per_launch"],1), "avg launch", round(r["avg_launch_ms"],1), "in flight", round(r["launches_in_flight"],2))
except Exception as e: print("$tag failed", e)
PY
}
run base STITCH_TRACE=1
run w8 STITCH_REGS_WAVES=8 STITCH_TRACE=1
run w8map1 STITCH_REGS_WAVES=8 STITCH_REGS_MAP=1
run base_nooverlap STITCH_NO_FILL_OVERLAP=1
run w8_nooverlap STITCH_REGS_WAVES=8 STITCH_NO_FILL_OVERLAP=1
echo "== 640 per step" | tee -a $O/log.txt
python bench.py --steps 2 --warmup 1 --cpu-reads 0 --reads-per-step 640 > $O/r640.json 2> $O/r640.err; python -c "
import json; d=json.load(open('$O/r640.json')); print('r640', d['value'], d['roofline']['fill_busy_ms_per_launch'])" | tee -a $O/log.txt
grep -h "\[trace\]" $O/base.err | tail -24 > $O/trace_base.txt
grep -h "\[trace\]" $O/w8.err | tail -24 > $O/trace_w8.txt
