#!/bin/bash
# the final tree: the whole GPU suite, then the profiles of tag r04_b again (kernel sources changed since), the configurations' lines, four bench processes in a row
set -o pipefail
O=gpurun_out/collect_r04_b; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final_suite.txt 2>&1; rc=$?; tail -3 gpurun_out/r04_final_suite.txt; [ $rc -eq 0 ] || exit 1
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
for k in 1 2 3 4; do timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-reads 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'fallbacks', d['roofline'].get('fill_fallbacks'), 'retired', d['roofline'].get('teams_retired'))"; done | tee $O/r04_b_four_benches.txt
