#!/bin/bash
# y-suffix records as one 64-bit store each (build with STITCH_DEFINES=STITCH_YREC_B64): cfg5, and the suboptimal tests
O=gpurun_out/r4yb; mkdir -p $O
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2> $O/c5.err | tee $O/c5.json | cut -c1-700 && \
timeout -k 10 300 python -m pytest tests/test_gpu_stream.py tests/test_gpu_fullsize_configs.py -x -q 2>&1 | tail -3 | tee $O/t.txt
