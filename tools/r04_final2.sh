#!/bin/bash
# after the last kernel change: fuzz through the teams, the profiles of tag r04_b again, the configurations' lines
set -o pipefail
O=gpurun_out/collect_r04_b; mkdir -p $O
FUZZ_STREAM=1 FUZZ_SECONDS=100 FUZZ_SEED=41000 timeout -k 10 200 python tests/gpu_fuzz.py > gpurun_out/r04_final_fuzz.txt 2>&1; tail -1 gpurun_out/r04_final_fuzz.txt
bash profiles/collect.sh r04_b > gpurun_out/r04b_collect.log 2>&1; echo "collect rc $?"; tail -1 gpurun_out/r04b_collect.log | cut -c1-200
bash profiles/collect_cmd.sh r04_b cfg5 python3 tests/config_runs.py --config cfg5 --reads 36 --batch 36 > gpurun_out/r04b_cfg5.log 2>&1; echo "cfg5 rc $?"
{
  timeout -k 10 200 python tests/config_runs.py --config cfg1 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg2 --mode global --reads 160 --batch 80 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg2 --mode query-local --reads 160 --batch 80 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg3 --reads 2048 --batch 1024 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg3 --reads 4096 --batch 2048 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 100 --batch 100 2>/dev/null | tail -1
  STITCH_NO_YBITS=1 timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2>/dev/null | tail -1
} > $O/r04_b_configs.json
wc -l $O/r04_b_configs.json
