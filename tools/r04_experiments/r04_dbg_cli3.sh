#!/bin/bash
O=gpurun_out/r4r; mkdir -p $O
python - <<PY
import sys; sys.path.insert(0,'.')
from stitch_amd import synth
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 640, 10000, 44)
open('/tmp/ref.fa','w').write(''.join(f'>{n}\n{s.decode()}\n' for n,s in db))
open('/tmp/r.fq','w').write(''.join(f'@read_{k:07d}\n{r.decode()}\n+\n{"I"*len(r)}\n' for k,r in enumerate(reads)))
PY
run() { tag=$1; shift; ( export "$@"; timeout -k 10 150 stitch_amd/bin/stitch-align -f /tmp/r.fq -r /tmp/ref.fa --batch $B > /dev/null 2> $O/$tag.err ); echo "$tag: $(grep 'stitch-align:' $O/$tag.err | cut -c1-200)" | tee -a $O/log.txt; }
B=640 run nostream STITCH_NO_STREAM=1
B=320 run stream320 STITCH_X=1
B=640 run stream640 STITCH_TRACE=1
B=640 run stream640_blocks60 STITCH_STREAM_BLOCKS=60
timeout -k 10 150 python tests/config_runs.py --config cfg2 --reads 1280 --batch 640 2>/dev/null | cut -c1-300 | tee -a $O/log.txt
