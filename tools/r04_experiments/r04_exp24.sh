#!/bin/bash
O=gpurun_out/r4tk; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_stream.py -x -q > $O/t.txt 2>&1; tail -2 $O/t.txt
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 100 --batch 100 2>/dev/null | tail -1 | cut -c1-330 | tee $O/c5.json
for k in 1 2 3 4 5 6 7 8; do STITCH_TRACE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2> $O/b$k.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'fallbacks', d['roofline'].get('fill_fallbacks'), 'retired', d['roofline'].get('teams_retired'))"; grep "asked to leave\|called off" $O/b$k.err | cut -c1-160; done | tee $O/benches.txt
