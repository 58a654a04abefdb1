#!/bin/bash
set -o pipefail
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/log.txt; tail -4 $O/tests.log | tee -a $O/log.txt
for cfg in cfg3 cfg1; do
  timeout -k 10 300 python tests/config_runs.py --config $cfg --out $O/$cfg.json > /dev/null 2> $O/$cfg.err; echo "$cfg rc $?" | tee -a $O/log.txt
  python -c "
import json; d=json.load(open('$O/$cfg.json')); print('$cfg', round(d['reads_per_sec'],1), 'reads/s', 'fill', d['fill_ms'], 'walk', d['walk_ms'], 'prealign', d['prealign_ms'], 'h2d', d['h2d_ms'], 'd2h', d['d2h_ms'], 'launches', d['launches'], 'bad', d['chains_whose_ops_do_not_rescore'], d.get('reads_differing_from_oracle'))" | tee -a $O/log.txt
done
