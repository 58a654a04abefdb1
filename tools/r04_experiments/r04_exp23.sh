#!/bin/bash
# do back-to-back processes stall launches beside the teams?  four bench processes in a row, traced
O=gpurun_out/r4retire; mkdir -p $O
for k in 1 2 3 4 5 6 7 8; do
  STITCH_TRACE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2> $O/b$k.err > $O/b$k.json || exit 1
  python - $O/b$k.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("run", sys.argv[1][-7:-5], round(d["value"],1), "fallbacks", d["roofline"].get("fill_fallbacks"), "retired", d["roofline"].get("teams_retired"), "ms/step", round(d["ms_per_step"]))
PY
  grep "asked to leave\|called off" $O/b$k.err | cut -c1-150; true
done
