#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
python - <<PY
import sys; sys.path.insert(0,'.')
from stitch_amd import synth
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 1280, 10000, 44)
open('/tmp/ref.fa','w').write(''.join(f'>{n}\n{s.decode()}\n' for n,s in db))
open('/tmp/r.fq','w').write(''.join(f'@read_{k:07d}\n{r.decode()}\n+\n{"I"*len(r)}\n' for k,r in enumerate(reads)))
PY
STITCH_TRACE=1 timeout -k 10 200 stitch_amd/bin/stitch-align -f /tmp/r.fq -r /tmp/ref.fa --batch 640 > /dev/null 2> $O/cli2.err
grep -c "\[trace\] jobs" $O/cli2.err; grep "persistent\|stitch-align\|launch jobs" $O/cli2.err | cut -c1-250 | head -20
grep "\[trace\] jobs" $O/cli2.err | awk 'NR%40==1' | cut -c1-200 | head -30
