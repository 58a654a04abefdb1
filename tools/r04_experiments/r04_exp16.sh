#!/bin/bash
# the whole GPU suite on the tree, then the fuzz through the persistent teams
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/suite.txt && \
FUZZ_STREAM=1 FUZZ_SECONDS=150 FUZZ_SEED=7000 timeout -k 10 260 python tests/gpu_fuzz.py 2>&1 | tail -4 | tee $O/fuzz_stream.txt
