#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
python - <<PY
import sys; sys.path.insert(0,'.')
from stitch_amd import synth
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 200, 10000, 44)
open('/tmp/ref.fa','w').write(''.join(f'>{n}\n{s.decode()}\n' for n,s in db))
open('/tmp/r.fq','w').write(''.join(f'@read_{k:07d}\n{r.decode()}\n+\n{"I"*len(r)}\n' for k,r in enumerate(reads)))
PY
STITCH_TRACE=1 stitch_amd/bin/stitch-align -f /tmp/r.fq -r /tmp/ref.fa --batch 100 --output-format sam > /tmp/out.sam 2> $O/cli.err
tail -30 $O/cli.err | cut -c1-300
grep -vc "^@" /tmp/out.sam; grep -v "^@" /tmp/out.sam | cut -f1 | sort | uniq -c | sort -k1,1nr | head -5
grep -v "^@" /tmp/out.sam | head -3 | cut -c1-400
