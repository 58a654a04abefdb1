import sys, time, os
sys.path.insert(0, '.')
import stitch_amd
from stitch_amd import synth
mode = sys.argv[1]
db = synth.make_db(50, 5000, 1001)
reads = synth.make_reads(db, 200, 10000, 44)
al = stitch_amd.Builder().build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
def call(rs, tag):
    t0 = time.time(); al.align(rs); tm = al.timing()
    print(mode, tag, round(time.time() - t0, 2), 's fallbacks', tm['fallbacks'], 'stream_runs', tm['stream_runs'], flush=True)
if mode == 'classic_first':
    call(reads[:30], 'small classic call')
call(reads, 'call A')
call(reads, 'call B')
