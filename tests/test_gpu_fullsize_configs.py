"""BASELINE configs[2] and configs[4] at their FULL shapes, as `-m gpu` tests (tests/config_runs.py measures them; this file is the
parity side):

  cfg3  10 kb reads vs 50 x 5 kb, --double-strand --pre-align (k = 12, w = 50, s = 100, subset): pre-alignment filter, then the
        jump DP on the passing contig-strands only (Aligners::align, aligners/mod.rs:246-340)
  cfg5  20 kb PacBio-like reads vs 200 circular 5 kb contigs, --circular --suboptimal: four granule registers per lane, 80 rows
        per lane, circular contigs, traceback_all + realign_origin (traceback/mod.rs:152-217, aligners/mod.rs:442-553)

Checked: the register-resident kernel is the one that runs (no fallback), cell counts, the size-independent properties the
domain offers (a chain's score recomputed from its operations; sortedness and the suboptimal threshold; constructed reads
with known answers), cfg3 reads compared with the oracle at full size on their passing contigs (a few GB of oracle cells), and
one oracle comparison that puts more than 64 contigs (NQ = 4), 5 kb contigs (80 rows per lane) and circular contigs together."""
import pytest

import stitch_amd
from oracle import oracle as orc
from stitch_amd import synth

pytestmark = pytest.mark.gpu


def recompute_score(ops):            # A=1 B=-4 O=-6 E=-2 J=-10 (CLI defaults)
    sc, run = 0, None
    for k, _, _ in ops:
        if k == 0: sc += 1
        elif k == 1: sc += -4
        elif k in (2, 3): sc += -2 + (-6 if run != k else 0)
        elif k == 6: sc += -10
        run = k
    return sc


def mem_available():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) * 1024
    return 0


# ---- cfg3 ---------------------------------------------------------------------------------------------------------------------------
CFG3 = dict(double_strand=True, pre_align=True, kmer_size=12, band_width=50, pre_align_min_score=100, pre_align_subset_contigs=True)


@pytest.fixture(scope="module")
def cfg3():
    db = synth.make_db(50, 5000, 1001)
    reads = synth.make_reads(db, 24, 10000, 45, both_strands=True)
    al = stitch_amd.Builder(**CFG3).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
    res = al.align(reads)
    return db, reads, al, res, al.timing()


def test_cfg3_full_size_kernel_choice_and_properties(cfg3):
    db, reads, al, res, tm = cfg3
    assert tm["fill_kind"] == 2 and tm["fallbacks"] == 0, tm          # the pre-filtered subsets still go to the register kernel
    mapped = [k for k, (ch, pre) in enumerate(res) if ch]
    assert len(mapped) >= 18                                           # ~10 % of the stream is random sequence: filtered out, unmapped
    cells = 0
    for k, (ch, pre) in enumerate(res):
        if not ch:
            assert pre is None                                         # nothing passed the filter: no xs score either (mod.rs:280-287)
            continue
        assert pre is not None and pre >= 100
        c = ch[0]
        assert recompute_score(c.operations) == c.score, k
        assert (c.ylen, c.xlen) == (10000, 5000) and c.yend <= 10000
        used = {c.start_contig_idx} | {o[1] for o in c.operations if o[0] == 6}
        assert all(0 <= u < 100 for u in used)
    # every DP ran on a SUBSET of the 100 contig-strands: far fewer cells than 24 x 10 000 x 500 000
    assert 0 < al.cells_filled < 0.25 * 24 * 10000 * 500000


def test_cfg3_reads_equal_the_oracle_at_full_size_on_their_passing_contigs(cfg3):
    db, reads, al, res, tm = cfg3
    avail = mem_available()
    if avail < 12 << 30:
        pytest.skip(f"the oracle needs a few GiB per read ({avail / 2**30:.0f} GiB available)")
    targets = [(n, s.decode()) for n, s in db]
    o = orc.Aligners(targets, **CFG3)
    picks = [k for k, (ch, _) in enumerate(res) if ch and any(op[0] == 6 for op in ch[0].operations)][:2] + [k for k, (ch, _) in enumerate(res) if not ch][:1]
    assert picks
    for k in picks:
        want = o.align(reads[k].decode())
        got, pre = res[k]
        assert [c.key() for c in got] == [c.key() for c in want], f"cfg3 read {k} differs from the oracle at full size"
        assert pre == o.prealign_score()
        assert al.format_sam(k, f"read_{k:07d}", reads[k], b"I" * len(reads[k])) == o.format_sam(f"read_{k:07d}", reads[k].decode(), "I" * len(reads[k]), prealign=o.prealign_score())


# ---- cfg5 ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg5():
    db = synth.make_db(200, 5000, 1002)
    reads = synth.make_reads(db, 6, 20000, 47, sub=0.01, ins=0.005, dele=0.005, circular=True)
    s3, s77, s150, s199 = (db[k][1] for k in (3, 77, 150, 199))
    # four whole plasmids, each rotated: every one aligns end to end with ONE zero-cost jump across its origin (single_contig_aligner.rs:
    # 258-289); the read starts and ends in mid-contig, so realign_origin has nothing to do (mod.rs:365-410)
    rot = lambda s, k: s[k:] + s[:k]
    reads = reads + [rot(s3, 3000) + rot(s77, 1234) + rot(s150, 4000) + rot(s199, 2500)]
    al = stitch_amd.Builder(circular=True, suboptimal=True).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
    res = al.align(reads)
    return db, reads, al, res, al.timing()


def test_cfg5_full_size_kernel_choice_and_properties(cfg5):
    db, reads, al, res, tm = cfg5
    assert tm["fill_kind"] == 2 and tm["fallbacks"] == 0, tm
    assert tm["wg_per_read"] >= 1                                      # (the last launch may be an origin re-alignment on a few contigs)
    assert al.cells_filled >= 7 * 20000 * 200 * 5000                   # every read against all 200 contigs (+ re-alignments)
    for k, (ch, _) in enumerate(res):
        assert ch, k
        scores = [c.score for c in ch]
        assert scores == sorted(scores, reverse=True)                  # mod.rs:318-329: sorted by score, the suboptimal threshold applied
        assert all(float(s) >= float(scores[0]) * 20.0 / 100.0 for s in scores)
        for c in ch:
            assert c.ylen == 20000 and c.xlen == 5000 and c.ystart <= c.yend <= 20000 and len(c.operations) > 0
            y = sum(1 for o in c.operations if o[0] in (0, 1, 2))
            assert y == c.yend - c.ystart                              # the operations consume exactly the chain's read span


def test_cfg5_constructed_read_known_answer(cfg5):
    db, reads, al, res, tm = cfg5
    best = res[-1][0][0]
    # 20 000 matches, three inter-contig jumps at -10, four free jumps across an origin
    assert (best.score, best.ystart, best.yend, best.start_contig_idx, best.end_contig_idx) == (20000 - 30, 0, 20000, 3, 199)
    jumps = [(o[1], o[2]) for o in best.operations if o[0] == 6]
    want = [(3, 0), (77, 1234), (77, 0), (150, 4000), (150, 0), (199, 2500), (199, 0)]
    # (where the base behind a junction happens to equal the next segment's, the jump may sit a base or two later: same score)
    assert [c for c, _ in jumps] == [c for c, _ in want] and all(abs(a - b) <= 3 for (_, a), (_, b) in zip(jumps, want)), jumps
    if jumps == want:
        assert best.cigar() == "2000=5000j3000=74C1766j3766=5000j1234=73C2766J1000=5000j4000=49C1500j2500=5000j2500="
    assert sum(1 for o in best.operations if o[0] == 0) == 20000 and len(best.operations) == 20000 + 7


def test_more_than_64_circular_5kb_contigs_equal_the_oracle():
    """NQ = 4 (70 active contigs), 80 rows per lane (5 kb contigs) and circular contigs TOGETHER, with --suboptimal, against the
    oracle: 800-base reads keep its matrices at 4.5 GB."""
    avail = mem_available()
    if avail < 10 << 30:
        pytest.skip(f"the oracle needs 4.5 GiB ({avail / 2**30:.0f} GiB available)")
    db = synth.make_db(70, 5000, 1002)
    reads = synth.make_reads(db, 3, 800, 48, sub=0.01, ins=0.005, dele=0.005, circular=True, random_frac=0.0)
    wrap = db[9][1][4700:] + db[9][1][:300] + db[41][1][2000:2200]     # across an origin, then another contig
    reads = reads + [wrap]
    targets = [(n, s.decode()) for n, s in db]
    al = stitch_amd.Builder(circular=True, suboptimal=True).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
    res = al.align(reads)
    tm = al.timing()
    assert tm["fallbacks"] == 0
    o = orc.Aligners(targets, circular=True, suboptimal=True)
    for k, r in enumerate(reads):
        want = o.align(r.decode())
        assert [c.key() for c in res[k][0]] == [c.key() for c in want], f"read {k}"
        assert al.format_sam(k, f"r{k}", r, b"I" * len(r)) == o.format_sam(f"r{k}", r.decode(), "I" * len(r))
