#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_stream.py tests/test_cli.py -x -q 2>&1 | tail -3 | tee -a $O/log.txt
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-reads 0 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value'],1), 'by cells', round(d['reads_per_sec_by_cells'],1), 'fallbacks', d['roofline']['fill_fallbacks'], 'frac', round(d['roofline']['frac'],4))" | tee -a $O/log.txt
timeout -k 10 300 python tools/e2e_rate.py --out $O/e2e.json 2> $O/e2e.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: {kk: round(vv,2) if isinstance(vv,float) else vv for kk,vv in d[k].items()} for k in ('bam','sam')})" | tee -a $O/log.txt
