"""Checks the column-parallel formulation used by the HIP kernels (phase A / insertion scan / phase C, row-m
seeding, suffix-clip reductions, compact traceback, fix-ups, walk) against the oracle, on the CPU.

tests/emu runs the SAME per-row functions and the same fix-up / walk code as the kernels (stitch_amd/csrc/dp_core.h,
walk_core.h) with the wave64 cross-lane steps emulated by loops, so this suite catches errors in the algorithm
before a GPU is involved.  The `-m gpu` suite (test_gpu_parity.py) then checks the kernels themselves."""
import json
import os
import random

import pytest

from oracle import oracle as orc
from tests.emu.emu import Emu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SINGLE = json.load(open(os.path.join(G, "single_contig.json")))
MULTI = json.load(open(os.path.join(G, "multi_contig.json")))
MIN = orc.MIN_SCORE
CLIPS = {"local": (0, 0, 0, 0), "querylocal": (MIN, MIN, 0, 0), "targetlocal": (0, 0, MIN, MIN), "global": (MIN, MIN, MIN, MIN),
         "query-local": (MIN, MIN, 0, 0), "target-local": (0, 0, MIN, MIN)}


def rc(seq):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    return "".join(comp[c] for c in reversed(seq))


def resolve(seq):
    return rc(seq[3:]) if seq.startswith("rc:") else seq


def params(match, mismatch, go, ge, js, jo, ji, mode, circular=False):
    xp, xs, yp, ys = CLIPS[mode]
    return [match, mismatch, go, ge, js, jo, ji, xp, xs, yp, ys, int(circular)]


def oracle_multi(p, contigs):
    al = orc.MultiContigAligner()
    sc = orc.scoring_array(match=p[0], mismatch=p[1], gap_open=p[2], gap_extend=p[3], jump_same=p[4], jump_opp=p[5], jump_inter=p[6],
                           xclip_prefix=p[7], xclip_suffix=p[8], yclip_prefix=p[9], yclip_suffix=p[10])
    for name, fwd, seq in contigs:
        al.add_contig(name, fwd, seq, bool(p[11]), sc)
    return al


@pytest.mark.parametrize("t", SINGLE, ids=[t["name"] for t in SINGLE])
def test_golden_single(t):
    s = t["scoring"]
    p = params(s["match"], s["mismatch"], s["gap_open"], s["gap_extend"], s["jump"], s["jump"], s["jump"], t["mode"], t["circular"])
    a = Emu(p, [("x", True, t["x"])]).job(t["y"])[0]
    e = t["expect"]
    # the reference's local()/querylocal()/targetlocal() strip clips from the op list (single_contig_aligner.rs:782-863)
    drop = {"local": (4, 5), "querylocal": (5,), "targetlocal": (4,), "global": ()}[t["mode"]]
    a.ops = [o for o in a.ops if o[0] not in drop]
    assert (a.xstart, a.xend, a.ystart, a.yend, a.score, a.start_contig_idx, a.cigar(), a.length) == \
        (e["xstart"], e["xend"], e["ystart"], e["yend"], e["score"], 0, e["cigar"], e["length"]), a


@pytest.mark.parametrize("t", MULTI, ids=[t["name"] for t in MULTI])
def test_golden_multi(t):
    for case in t["cases"]:
        c0 = t["contigs"][0]
        mismatch, go, ge, jump = c0["scoring"]
        js, jo, ji = case["jump_scores"] or (jump, jump, jump)
        p = params(1, mismatch, go, ge, js, jo, ji, c0["kind"])
        contigs = [(c["name"], c["is_forward"], resolve(c["seq"])) for c in t["contigs"]]
        a = Emu(p, contigs).job(resolve(t["y"]))[0]
        e = case["expect"]
        assert (a.xstart, a.xend, a.ystart, a.yend, a.score, a.start_contig_idx, a.cigar(), a.length) == \
            (e["xstart"], e["xend"], e["ystart"], e["yend"], e["score"], e["start_contig_idx"], e["cigar"], e["length"]), a


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, rate):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate:
            out.append(rng.choice("ACGT"))
        elif r < 1.5 * rate:
            continue
        elif r < 2 * rate:
            out.append(ch + rng.choice("ACGT"))
        else:
            out.append(ch)
    return "".join(out)


def random_case(rng, big=False):
    mode = rng.choice(["local", "local", "querylocal", "targetlocal", "global"])
    tie_heavy = rng.random() < 0.5
    alphabet = rng.choice(["ACGT", "AC", "A", "ACGTN"]) if tie_heavy else "ACGT"
    if tie_heavy:
        match, mismatch = rng.choice([(1, -1), (1, 0), (2, -1), (1, -3)])
        go, ge = rng.choice([(0, -1), (-1, -1), (-2, 0), (0, 0), (-3, -1)])
        jumps = [rng.choice([0, -1, -2, -3]) for _ in range(3)]
    else:
        match, mismatch = 1, rng.choice([-1, -4, -2])
        go, ge = rng.choice([(-6, -2), (-5, -1), (-3, -1)])
        jumps = [rng.choice([-10, -5, -1, -8])] * 3 if rng.random() < 0.5 else [rng.choice([-10, -4, -1]) for _ in range(3)]
    T = rng.randint(1, 4)
    double = rng.random() < 0.5
    circular = rng.random() < 0.3
    top = 600 if big else 40
    targets = [rand_seq(rng, rng.randint(1, top), alphabet) for _ in range(T)]
    contigs = [(f"t{k}", True, s) for k, s in enumerate(targets)]
    if double:
        contigs += [(f"t{k}", False, rc(s)) for k, s in enumerate(targets)]
    # read: chimera of pieces of contigs (with errors) and junk
    pieces = []
    for _ in range(rng.randint(1, 4)):
        if rng.random() < 0.2:
            pieces.append(rand_seq(rng, rng.randint(1, 12), alphabet))
        else:
            src = rng.choice(contigs)[2]
            a = rng.randrange(len(src)); b = rng.randint(a + 1, len(src))
            pieces.append(mutate(rng, src[a:b], rng.choice([0.0, 0.05, 0.15])))
    y = "".join(pieces) or "A"
    y = y[:(900 if big else 60)]
    p = params(match, mismatch, go, ge, jumps[0], jumps[1], jumps[2], mode, circular)
    return p, contigs, y


def compare_case(p, contigs, y, rng):
    emu = Emu(p, contigs)
    al = oracle_multi(p, contigs)
    C = len(contigs)
    subset = None
    if C > 1 and rng.random() < 0.4:
        subset = sorted(rng.sample(range(C), rng.randint(1, C - 1)))
    try:
        want = al.custom(y, subset)
    except RuntimeError as e:          # the reference itself is undefined here (oracle: "traceback index out of range")
        assert "out of range" in str(e)
        want = None
    got = emu.job(y, subset, 0)[0]
    if want is not None:
        assert got is not None and got.key() == want.key(), f"primary\nwant {want}\ngot  {got}\nparams {p}\ncontigs {contigs}\ny {y} subset {subset}"
    else:
        assert got is None
    act = subset if subset is not None else list(range(C))
    # traceback_from every active contig == the per-end-contig candidates of traceback_all
    cands = emu.job(y, subset, 1)
    for k, c in enumerate(act):
        try:
            w = al.traceback_from(len(y), c)
        except RuntimeError as e:
            assert "out of range" in str(e)
            assert cands[k] is None
            continue
        g = cands[k]
        assert (g is None) == (w is None)
        if w is not None:
            assert g.key() == w.key(), f"from {c}\nwant {w}\ngot  {g}\nparams {p}\ncontigs {contigs}\ny {y} subset {subset}"


    # ... and the same candidates when the walks JOIN the reference chain the way the device's do (walk_core.h JoinRole: emu mode 3)
    joined = emu.job(y, subset, 3)
    assert [None if c is None else c.key() for c in joined] == [None if c is None else c.key() for c in cands], \
        f"joined walks\nparams {p}\ncontigs {contigs}\ny {y} subset {subset}"


def test_joined_walks_keep_their_own_start_when_they_meet_the_reference_chain_in_column_0():
    """two chains that start in the same contig at different cells both arrive at (contig, row 0, column 0, start) once their prefix clips
    are done: the second must not take the first one's start coordinates (found on the device by tests/gpu_fuzz.py FUZZ_STREAM=1, seed
    7087; here the same read and contigs through the CPU emulation)"""
    import os
    os.environ["FUZZ_STREAM"] = "1"
    try:
        from tests import gpu_fuzz as G
        targets, reads, opts, _, _, lens = G.draw(7087)
    finally:
        del os.environ["FUZZ_STREAM"]
    assert lens == [34, 1453, 5, 29] and opts == dict(double_strand=True, circular=True, suboptimal=True)
    contigs = [(n, True, s) for n, s in targets] + [(n, False, rc(s)) for n, s in targets]
    p = params(1, -4, -6, -2, -10, -10, -10, "local", True)
    emu = Emu(p, contigs)
    y = reads[11]
    plain, joined = emu.job(y, None, 1), emu.job(y, None, 3)
    assert [c.key() for c in joined] == [c.key() for c in plain]
    ref = max(plain, key=lambda c: c.score)
    assert any(c.start_contig_idx == ref.start_contig_idx and (c.xstart, c.ystart) != (ref.xstart, ref.ystart) for c in plain)      # (the situation is there)


@pytest.mark.parametrize("seed", range(400))
def test_random_small(seed):
    rng = random.Random(seed)
    p, contigs, y = random_case(rng)
    compare_case(p, contigs, y, rng)


@pytest.mark.parametrize("seed", range(40))
def test_random_multi_tile(seed):
    """contigs longer than one 256-row tile: exercises the scan / reduction carries between tiles"""
    rng = random.Random(10_000 + seed)
    p, contigs, y = random_case(rng, big=True)
    compare_case(p, contigs, y, rng)


def local16_eligible(p):
    """mirror of local16_ok() in stitch_amd/csrc/stitch_api.cpp"""
    return p[7:11] == [0, 0, 0, 0] and p[2] + p[3] < 0


def compare_local(p, contigs, y, rng):
    emu = Emu(p, contigs)
    al = oracle_multi(p, contigs)
    C = len(contigs)
    subset = None
    if C > 1 and rng.random() < 0.4:
        subset = sorted(rng.sample(range(C), rng.randint(1, C - 1)))
    want = al.custom(y, subset)
    got = emu.job(y, subset, 0, local16=True)[0]
    assert got.key() == want.key(), f"primary\nwant {want}\ngot  {got}\nparams {p}\ncontigs {contigs}\ny {y} subset {subset}"
    act = subset if subset is not None else list(range(C))
    cands = emu.job(y, subset, 1, local16=True)
    for k, c in enumerate(act):
        w = al.traceback_from(len(y), c)
        assert cands[k].key() == w.key(), f"from {c}\nwant {w}\ngot  {cands[k]}\nparams {p}\ncontigs {contigs}\ny {y} subset {subset}"


@pytest.mark.parametrize("seed", range(600))
def test_random_local16(seed):
    """the Local-mode kernel's formulation: ordered-key selection, 16-bit state, filtered y-suffix trackers"""
    rng = random.Random(50_000 + seed)
    while True:
        p, contigs, y = random_case(rng, big=(seed % 12 == 0))
        if local16_eligible(p):
            break
    compare_local(p, contigs, y, rng)
