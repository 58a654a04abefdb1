#!/bin/bash
# one ticket, then the run is called off: the stream tests, eight bench processes in a row, the headline's profiles again
set -o pipefail
O=gpurun_out/collect_r04_b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_stream.py -x -q > gpurun_out/r04_final4_tests.txt 2>&1; rc=$?; tail -1 gpurun_out/r04_final4_tests.txt; [ $rc -eq 0 ] || exit 1
for k in 1 2 3 4 5 6 7 8; do STITCH_TRACE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2> gpurun_out/r04_final4_b$k.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'fallbacks', d['roofline'].get('fill_fallbacks'), 'retired', d['roofline'].get('teams_retired'))"; grep "asked to leave\|called off" gpurun_out/r04_final4_b$k.err | cut -c1-160; done | tee $O/r04_b_eight_benches_in_a_row.txt
bash profiles/collect.sh r04_b > gpurun_out/r04b_collect.log 2>&1; echo "collect rc $?"; tail -1 gpurun_out/r04b_collect.log | cut -c1-200
