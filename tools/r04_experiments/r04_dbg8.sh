#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
timeout -k 10 250 python tools/r04_dbg7.py stream_first 2>/dev/null | tee -a $O/log.txt
timeout -k 10 400 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -3 | tee -a $O/log.txt
timeout -k 10 300 python tools/e2e_rate.py --out $O/e2e.json 2> $O/e2e.err | cut -c1-1800 | tee -a $O/log.txt
