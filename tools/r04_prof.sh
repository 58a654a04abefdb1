#!/bin/bash
# in-kernel section stamps (diagnostic build, never shipped): builds a second library into /tmp and runs cfg5 / cfg2 shapes with it
set -o pipefail
O=gpurun_out/r4m; mkdir -p $O
export STITCH_PROFILE_BUILD=1
python - <<PY > $O/build.log 2>&1
from stitch_amd import build as b
b.build(force=True)
PY
echo "build rc $?" | tee -a $O/log.txt
STITCH_PROFILE_DUMP=1 timeout -k 10 300 python tests/config_runs.py --config cfg5 --reads 12 --batch 12 > $O/cfg5.json 2> $O/cfg5.err; grep "\[prof\]" $O/cfg5.err | tail -8 | tee -a $O/log.txt
STITCH_PROFILE_DUMP=1 timeout -k 10 300 python tests/config_runs.py --config cfg2 --reads 80 --batch 80 > $O/cfg2.json 2> $O/cfg2.err; grep "\[prof\]" $O/cfg2.err | tail -4 | tee -a $O/log.txt
