#!/bin/bash
# fuzz sweep: the other clipping modes, the filter, persistent teams with the other modes' options drawn too
O=gpurun_out/r4fz; mkdir -p $O
FUZZ_MODES=1 FUZZ_SECONDS=240 FUZZ_SEED=21000 timeout -k 10 330 python tests/gpu_fuzz.py 2>&1 | tail -3 | tee $O/modes.txt && \
FUZZ_PREALIGN=1 FUZZ_SECONDS=240 FUZZ_SEED=22000 timeout -k 10 330 python tests/gpu_fuzz.py 2>&1 | tail -3 | tee $O/prealign.txt && \
FUZZ_STREAM=1 FUZZ_SECONDS=240 FUZZ_SEED=23000 timeout -k 10 330 python tests/gpu_fuzz.py 2>&1 | tail -3 | tee $O/stream.txt && \
FUZZ_MODES=1 FUZZ_SUBOPT=1 FUZZ_SECONDS=180 FUZZ_SEED=24000 timeout -k 10 270 python tests/gpu_fuzz.py 2>&1 | tail -3 | tee $O/modes_subopt.txt
