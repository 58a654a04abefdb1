#!/bin/bash
set -o pipefail
O=gpurun_out/r4l; mkdir -p $O
for v in stream_w8 ; do
  export STITCH_REGS_WAVES=8
  timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 72 --batch 72 --out $O/cfg5_$v.json > /dev/null 2> $O/cfg5_$v.err; echo "cfg5 $v rc $?" | tee -a $O/log.txt
  python -c "
import json; d=json.load(open('$O/cfg5_$v.json')); print('cfg5 $v', round(d['reads_per_sec'],2), 'reads/s', d['results_sha256'], 'stream_runs', d['stream_runs'], 'fallbacks', d['fallbacks'], 'fill', d['fill_ms'], 'walk', d['walk_ms'], 'd2h', d['d2h_ms'], 'launches', d['launches'])" | tee -a $O/log.txt
done
