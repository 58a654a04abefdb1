#!/bin/bash
# the final tree under the fuzz tool: every eligible read on fill_regs (one-bit records under --suboptimal / --circular), through the teams, the other modes, the filter
O=gpurun_out/r4finalfuzz; mkdir -p $O
FUZZ_REGS=1 FUZZ_SECONDS=200 FUZZ_SEED=51000 timeout -k 10 300 python tests/gpu_fuzz.py > $O/regs.txt 2>&1; tail -1 $O/regs.txt
FUZZ_REGS=1 FUZZ_SUBOPT=1 FUZZ_SECONDS=150 FUZZ_SEED=52000 timeout -k 10 250 python tests/gpu_fuzz.py > $O/regs_subopt.txt 2>&1; tail -1 $O/regs_subopt.txt
FUZZ_STREAM=1 FUZZ_SECONDS=200 FUZZ_SEED=53000 timeout -k 10 300 python tests/gpu_fuzz.py > $O/stream.txt 2>&1; tail -1 $O/stream.txt
FUZZ_MODES=1 FUZZ_SECONDS=120 FUZZ_SEED=54000 timeout -k 10 220 python tests/gpu_fuzz.py > $O/modes.txt 2>&1; tail -1 $O/modes.txt
FUZZ_PREALIGN=1 FUZZ_SECONDS=120 FUZZ_SEED=55000 timeout -k 10 220 python tests/gpu_fuzz.py > $O/prealign.txt 2>&1; tail -1 $O/prealign.txt
