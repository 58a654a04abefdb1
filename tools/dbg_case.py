"""Debugging aid: one failing parity case through the generic kernel and through fill_regs32, column-n arrays diffed."""
import os, sys, random, glob, shutil, struct
import numpy as np
sys.path.insert(0, '.')
import stitch_amd
from tests import test_gpu_parity as P

def load(path):
    b = open(path, 'rb').read()
    n, nact, R, C = struct.unpack_from('4I', b, 0)
    o = 16
    arrs = {}
    for name in ("S", "Slen", "Ival", "Ilen", "Sn", "SnLen", "Ly"):
        arrs[name] = np.frombuffer(b, dtype=np.int32, count=R, offset=o).copy(); o += 4 * R
    for name in ("Lx", "jti", "jtf"):
        arrs[name] = np.frombuffer(b, dtype=np.uint32, count=C * (n + 1), offset=o).reshape(C, n + 1).copy(); o += 4 * C * (n + 1)
    cd = np.frombuffer(b, dtype=np.uint32, count=2 * C, offset=o).reshape(C, 2)
    return n, nact, R, C, arrs, cd

def run(kind_env, targets, reads, opts, tag):
    d = f"/tmp/dump_{tag}"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    os.environ["STITCH_DUMP_DIR"] = d
    for k in ("STITCH_NO_REGS32", "STITCH_FORCE_REGS32"): os.environ.pop(k, None)
    os.environ.update(kind_env)
    al = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets])
    res = al.align(reads)
    return d, res, al.timing()

seed, mode = int(sys.argv[1]), sys.argv[2]
rng = random.Random(900 + seed)
T = rng.randint(1, 6)
targets = [(f"t{k}", P.rand_seq(rng, rng.randint(20, 500))) for k in range(T)]
double = rng.random() < 0.5
opts = dict(double_strand=double, circular=rng.random() < 0.4, circular_slop=rng.choice([0, 5, 20]), suboptimal=rng.random() < 0.4,
            use_eq_and_x=rng.random() < 0.5, soft_clip=rng.random() < 0.5)
if rng.random() < 0.6:
    opts.update(match_score=rng.choice([1, 2]), mismatch_score=rng.choice([-1, -4]), gap_open=rng.choice([-6, -3, 0]), gap_extend=rng.choice([-2, -1]),
                default_jump_score=rng.choice([-10, -5, -1]))
if rng.random() < 0.3:
    opts.update(jump_score_same_contig_and_strand=rng.choice([-10, -3]), jump_score_inter_contig=rng.choice([-12, -4]))
reads = [P.chimera(rng, targets, rng.randint(5, 400), both=double) for _ in range(6)]
opts["mode"] = mode
which = int(sys.argv[3]) if len(sys.argv) > 3 else 0
os.environ["STITCH_REGS_MIN_ROWS"] = "0"
opts_nc = dict(opts); opts_nc["circular"] = False       # pass 1 only
da, ra, ta = run({"STITCH_NO_REGS32": "1"}, targets, [reads[which]], opts_nc, "gen")
db, rb, tb = run({"STITCH_FORCE_REGS32": "1"}, targets, [reads[which]], opts_nc, "r32")
print("kinds", ta["fill_kind"], tb["fill_kind"], "lens", [len(t[1]) for t in targets], "n", len(reads[which]))
fa, fb = sorted(glob.glob(da + "/*.bin"))[0], sorted(glob.glob(db + "/*.bin"))[0]
n, nact, R, C, A, cd = load(fa); _, _, _, _, B, _ = load(fb)
for c in range(C):
    m, roff = int(cd[c][0]), int(cd[c][1])
    for name in ("S", "Slen", "Ival", "Ilen", "Sn", "SnLen", "Ly"):
        a, b = A[name][roff:roff + m], B[name][roff:roff + m]
        bad = np.nonzero(a != b)[0]
        if len(bad): print(f"contig {c} (m={m}) {name}: {len(bad)} rows differ, first rows {bad[:8] + 1}: generic {a[bad[:8]]} regs32 {b[bad[:8]]}")
    for name in ("Lx", "jti", "jtf"):
        bad = np.nonzero(A[name][c] != B[name][c])[0]
        if len(bad): print(f"contig {c} {name}: {len(bad)} columns differ, first {bad[:8]}: generic {A[name][c][bad[:8]]} regs32 {B[name][c][bad[:8]]}")
print("chains equal:", [c.key() for c in ra[0][0]] == [c.key() for c in rb[0][0]])
cfocus = int(os.environ.get("DBG_CONTIG", "-1"))
if cfocus >= 0:
    m, roff = int(cd[cfocus][0]), int(cd[cfocus][1])
    print("rows where generic Sn > S:", [(i + 1, int(A["S"][roff + i]), int(A["Sn"][roff + i]), int(A["SnLen"][roff + i]), int(A["Ly"][roff + i]), "mine", int(B["Sn"][roff + i]), int(B["SnLen"][roff+i]), int(B["Ly"][roff+i])) for i in range(m) if A["Sn"][roff + i] > A["S"][roff + i]][:60])
    print("last rows S:", [int(v) for v in A["S"][roff + m - 16: roff + m]])
    print("max S row:", int(np.argmax(A["S"][roff:roff + m])) + 1, int(A["S"][roff:roff + m].max()))
    for r in (ra, rb):
        for ch in r[0][0]:
            print(ch)
if cfocus >= 0:
    lo = int(os.environ.get("DBG_ROW0", "100"))
    for i in range(lo - 1, m):
        print("row", i + 1, "S", int(A["S"][roff + i]), "gen Sn/len/Ly", int(A["Sn"][roff + i]), int(A["SnLen"][roff + i]), int(A["Ly"][roff + i]), "mine", int(B["Sn"][roff + i]), int(B["SnLen"][roff + i]), int(B["Ly"][roff + i]))
