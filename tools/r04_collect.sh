#!/bin/bash
# round 4: every profile the repository commits, one box, one go
set -o pipefail
bash profiles/collect.sh r04_a > gpurun_out/r04_collect_a.log 2>&1; echo "collect a rc $?"; tail -2 gpurun_out/r04_collect_a.log
bash profiles/collect_cmd.sh r04_a cfg5 python3 tests/config_runs.py --config cfg5 --reads 36 --batch 36 > gpurun_out/r04_collect_cfg5.log 2>&1; echo "cfg5 rc $?"; tail -1 gpurun_out/r04_collect_cfg5.log
bash profiles/collect_cmd.sh r04_a cfg2_global python3 tests/config_runs.py --config cfg2 --mode global --reads 80 --batch 80 > gpurun_out/r04_collect_g.log 2>&1; echo "cfg2 global rc $?"; tail -1 gpurun_out/r04_collect_g.log
bash profiles/collect_cmd.sh r04_a cfg3 python3 tests/config_runs.py --config cfg3 --reads 1024 --batch 1024 > gpurun_out/r04_collect_cfg3.log 2>&1; echo "cfg3 rc $?"; tail -1 gpurun_out/r04_collect_cfg3.log
