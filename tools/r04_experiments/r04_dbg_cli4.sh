#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
python - <<PY
import sys; sys.path.insert(0,'.')
from stitch_amd import synth
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 320, 10000, 44)
open('/tmp/ref.fa','w').write(''.join(f'>{n}\n{s.decode()}\n' for n,s in db))
open('/tmp/r.fq','w').write(''.join(f'@read_{k:07d}\n{r.decode()}\n+\n{"I"*len(r)}\n' for k,r in enumerate(reads)))
PY
run() { tag=$1; shift; ( export "$@"; timeout -k 10 100 stitch_amd/bin/stitch-align -f /tmp/r.fq -r /tmp/ref.fa --batch 320 > /dev/null 2> $O/$tag.err ); echo "$tag: $(grep 'stitch-align:' $O/$tag.err | cut -c1-200)" | tee -a $O/log.txt; }
run serial STITCH_ALIGN_SERIAL=1
run threads STITCH_X=1
