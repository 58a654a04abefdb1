#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
python - <<PY
import sys; sys.path.insert(0,'.')
from stitch_amd import synth
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 160, 10000, 44)
open('/tmp/ref.fa','w').write(''.join(f'>{n}\n{s.decode()}\n' for n,s in db))
open('/tmp/r.fq','w').write(''.join(f'@read_{k:07d}\n{r.decode()}\n+\n{"I"*len(r)}\n' for k,r in enumerate(reads)))
PY
run() { tag=$1; shift; ( export "$@"; timeout -k 10 100 stitch_amd/bin/stitch-align -f /tmp/r.fq -r /tmp/ref.fa --batch 160 > /dev/null 2> $O/$tag.err ); echo "$tag: $(grep 'stitch-align:' $O/$tag.err | cut -c1-120)" | tee -a $O/log.txt; }
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run plain STITCH_X=1
ldd stitch_amd/bin/stitch-align | head -20 > $O/ldd.txt
python - <<PY > $O/pyrun.txt 2>&1
import sys, time; sys.path.insert(0,'.')
import numpy as np, stitch_amd
from stitch_amd import synth
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 160, 10000, 44)
al = stitch_amd.Builder().build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
t0=time.time(); al.align(reads); print('python first call', time.time()-t0, al.timing()['fallbacks'], al.timing()['stream_runs'])
t0=time.time(); al.align(reads); print('python second call', time.time()-t0, al.timing()['fallbacks'])
PY
cat $O/pyrun.txt | tail -3 | tee -a $O/log.txt
