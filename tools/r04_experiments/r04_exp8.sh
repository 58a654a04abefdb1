#!/bin/bash
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_regs.py tests/test_gpu_parity.py tests/test_gpu_regs32.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/log.txt; tail -3 $O/tests.log | tee -a $O/log.txt
for v in nostream stream nojoin_nostream; do
  unset STITCH_NO_STREAM STITCH_NO_JOIN
  if [ $v = nostream ]; then export STITCH_NO_STREAM=1; fi
  if [ $v = nojoin_nostream ]; then export STITCH_NO_STREAM=1 STITCH_NO_JOIN=1; fi
  STITCH_TRACE=1 timeout -k 10 300 python tests/config_runs.py --config cfg5 --out $O/cfg5_$v.json > /dev/null 2> $O/cfg5_$v.err; echo "cfg5 $v rc $?" | tee -a $O/log.txt
  python -c "
import json; d=json.load(open('$O/cfg5_$v.json')); print('cfg5 $v', round(d['reads_per_sec'],2), 'reads/s', d['results_sha256'], 'stream_runs', d['stream_runs'], 'fallbacks', d['fallbacks'], 'fill', d['fill_ms'], 'walk', d['walk_ms'], 'd2h', d['d2h_ms'], 'launches', d['launches'])" | tee -a $O/log.txt
done
grep "\[trace\]" $O/cfg5_stream.err | head -60 > $O/trace_cfg5.txt
