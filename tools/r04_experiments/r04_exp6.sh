#!/bin/bash
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_stream.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/log.txt; tail -3 $O/tests.log | tee -a $O/log.txt
for v in stream nostream; do
  if [ $v = nostream ]; then export STITCH_NO_STREAM=1; else unset STITCH_NO_STREAM; fi
  STITCH_TRACE=1 timeout -k 10 400 python tests/config_runs.py --config cfg5 --out $O/cfg5_$v.json > /dev/null 2> $O/cfg5_$v.err; echo "cfg5 $v rc $?" | tee -a $O/log.txt
  python -c "
import json; d=json.load(open('$O/cfg5_$v.json')); print('cfg5 $v', round(d['reads_per_sec'],2), 'reads/s', d['results_sha256'], 'stream_runs', d['stream_runs'], 'fallbacks', d['fallbacks'], 'fill', d['fill_ms'], 'walk', d['walk_ms'], 'launches', d['launches'])" | tee -a $O/log.txt
done
grep "\[trace\]" $O/cfg5_stream.err | head -60 > $O/trace_cfg5.txt
