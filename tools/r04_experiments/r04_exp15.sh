#!/bin/bash
# the early granule look: headline bench twice, the stream tests, cfg5
O=gpurun_out/r4x; mkdir -p $O
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2> $O/b1.err | tee $O/b1.json | cut -c1-900 && \
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2> $O/b2.err | tee $O/b2.json | cut -c1-300 && \
timeout -k 10 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_fill_regs.py -x -q 2>&1 | tail -3 | tee $O/t.txt && \
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2> $O/c5.err | tee $O/c5.json | cut -c1-600
