#!/bin/bash
# after the JoinRole refactor (walk_core.h is one of the hashed kernel sources): the --suboptimal tests, then the headline's profiles again
set -o pipefail
O=gpurun_out/collect_r04_b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_gpu_regs.py -x -q > gpurun_out/r04_final3_tests.txt 2>&1; rc=$?; tail -2 gpurun_out/r04_final3_tests.txt; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2>/dev/null | tail -1 | cut -c1-330
bash profiles/collect.sh r04_b > gpurun_out/r04b_collect.log 2>&1; echo "collect rc $?"; tail -1 gpurun_out/r04b_collect.log | cut -c1-200
