#!/bin/bash
# y-suffix records as one bit per cell (YB instances): parity first, then cfg5 with and without
O=gpurun_out/r4yc; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_stream.py tests/test_gpu_regs.py -x -q 2>&1 | tail -3 | tee $O/t1.txt && \
FUZZ_REGS=1 FUZZ_SUBOPT=1 FUZZ_SECONDS=120 FUZZ_SEED=31000 timeout -k 10 200 python tests/gpu_fuzz.py 2>&1 | tail -2 | tee $O/fz1.txt && \
FUZZ_STREAM=1 FUZZ_SUBOPT=1 FUZZ_SECONDS=120 FUZZ_SEED=32000 timeout -k 10 200 python tests/gpu_fuzz.py 2>&1 | tail -2 | tee $O/fz2.txt && \
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2> $O/c5.err | tee $O/c5.json | cut -c1-420 && \
STITCH_NO_YBITS=1 timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2> $O/c5n.err | tee $O/c5n.json | cut -c1-420 && \
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize_configs.py -x -q 2>&1 | tail -3 | tee $O/t2.txt
