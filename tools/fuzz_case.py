"""Debugging aid: one seed of tests/gpu_fuzz.py (FUZZ_STREAM / FUZZ_MODES / FUZZ_PREALIGN as exported) through the product as drawn, launch
by launch, and with the default kernel choice; every read compared with the oracle, the first difference printed."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stitch_amd
from oracle import oracle as orc
import gpu_fuzz as G

seed = int(sys.argv[1])
targets, reads, opts, n_check, env, lens = G.draw(seed)
print("seed", seed, "opts", opts, "lens", lens, "reads", len(reads), [len(r) for r in reads], "env", env, flush=True)
o = orc.Aligners(targets, **G.oracle_opts(opts))
want = [[c.key() for c in o.align(r)] for r in reads]
variants = [("as drawn", env), ("launch by launch", dict(env, STITCH_NO_STREAM="1")), ("default kernels", {})]
if len(sys.argv) > 2: variants += [(sys.argv[2], dict(env, **dict(kv.split("=") for kv in sys.argv[2].split(","))))]
for name, e in variants:
    for k in ("STITCH_REGS_MIN_ROWS", "STITCH_STREAM_TEAMS", "STITCH_STREAM_BLOCKS", "STITCH_NO_STREAM", "STITCH_NO_JOIN"): os.environ.pop(k, None)
    os.environ.update(e)
    al = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets])
    res = al.align(reads)
    tm = al.timing()
    got = [[c.key() for c in r[0]] for r in res]
    bad = [k for k in range(len(reads)) if got[k] != want[k]]
    print(f"{name}: kind {tm['fill_kind']} stream_runs {tm['stream_runs']} launches {tm['launches']} fallbacks {tm['fallbacks']}: reads that differ {bad}", flush=True)
    for k in bad[:2]:
        print("  read", k, "len", len(reads[k]), "chains got", len(got[k]), "want", len(want[k]))
        for i in range(max(len(got[k]), len(want[k]))):
            g = got[k][i] if i < len(got[k]) else None; w = want[k][i] if i < len(want[k]) else None
            if g != w: print("   chain", i, "\n     got ", g, "\n     want", w); break
