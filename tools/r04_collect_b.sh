#!/bin/bash
# round 4, final tree: every profile the repository commits (tag r04_b), the configurations' own lines, the end-to-end rate
set -o pipefail
O=gpurun_out/collect_r04_b; mkdir -p $O
bash profiles/collect.sh r04_b > gpurun_out/r04b_collect.log 2>&1; echo "collect rc $?"; tail -2 gpurun_out/r04b_collect.log | cut -c1-300
bash profiles/collect_cmd.sh r04_b cfg5 python3 tests/config_runs.py --config cfg5 --reads 36 --batch 36 > gpurun_out/r04b_cfg5.log 2>&1; echo "cfg5 rc $?"
bash profiles/collect_cmd.sh r04_b cfg2_global python3 tests/config_runs.py --config cfg2 --mode global --reads 80 --batch 80 > gpurun_out/r04b_g.log 2>&1; echo "cfg2 global rc $?"
bash profiles/collect_cmd.sh r04_b cfg3 python3 tests/config_runs.py --config cfg3 --reads 1024 --batch 1024 > gpurun_out/r04b_cfg3.log 2>&1; echo "cfg3 rc $?"
{
  timeout -k 10 200 python tests/config_runs.py --config cfg1 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg2 --mode global --reads 160 --batch 80 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg2 --mode query-local --reads 160 --batch 80 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg3 --reads 2048 --batch 1024 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg3 --reads 4096 --batch 2048 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 71 --batch 71 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 100 --batch 100 2>/dev/null | tail -1
  timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 200 --batch 100 2>/dev/null | tail -1
} > $O/r04_b_configs.json
wc -l $O/r04_b_configs.json
timeout -k 10 300 python tools/e2e_rate.py --out $O/r04_b_e2e.json 2> $O/e2e.err | cut -c1-600
