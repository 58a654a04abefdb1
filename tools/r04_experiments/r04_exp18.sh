#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
FUZZ_STREAM=1 timeout -k 10 200 python tools/fuzz_case.py 7087 2>&1 | cut -c1-600 | tee $O/case2.txt && \
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "elsewhere or suboptimal" 2>&1 | tail -3 | tee $O/t2.txt && \
FUZZ_STREAM=1 FUZZ_SUBOPT=1 FUZZ_SECONDS=240 FUZZ_SEED=9000 timeout -k 10 330 python tests/gpu_fuzz.py 2>&1 | tail -4 | tee $O/fuzz_subopt_stream.txt && \
FUZZ_SUBOPT=1 FUZZ_SECONDS=200 FUZZ_SEED=12000 timeout -k 10 300 python tests/gpu_fuzz.py 2>&1 | tail -4 | tee $O/fuzz_subopt.txt
