"""The pre-alignment's second-generation score kernels against the oracle's banded scorer, pair by pair:

  banded_score_window_kernel (prealign_window.hip) pairs with a band whose columns fit a window of 256 rows held in registers
  full_score_skew16_kernel (prealign_skew16.hip)  pairs without a k-mer match, scored over the full matrix as an anti-diagonal sweep in
                                                  packed 16-bit words, the target along the rows (128 strips of up to 40 rows)

With ONE target, one strand and pre_align_min_score = 1 the score `Aligners::align` reports (`xs`) IS the pair's score, so every
read of a batch checks one (read, target) pair; the reads are built to have no 12-mer (k = 14 is used) in common with the target and still align
well (a substitution every 8-11 bases, a few gaps), so that the scores are in the hundreds, not the handful a random pair gives.
STITCH_PREALIGN_V1=1 runs the first-generation kernels on the same inputs."""
import random

import pytest

import stitch_amd
from tests.test_prealign import banded, rnd

pytestmark = pytest.mark.gpu


def seedless_relative(rng, s, k, gaps=0.004):
    """`s` with a substitution at least every k - 1 bases (no k-mer survives), and a few insertions / deletions"""
    out, run = [], 0
    for ch in s:
        r = rng.random()
        if r < gaps:
            continue
        if run >= k - 2 - rng.randrange(0, 3):
            ch = rng.choice([c for c in "ACGT" if c != ch]); run = 0
        else:
            run += 1
        out.append(ch)
        if rng.random() < gaps:
            out.append(rng.choice("ACGT")); run = 0
    return "".join(out)


def scores_of(target, reads, **kw):
    opts = dict(pre_align=True, pre_align_min_score=1, kmer_size=14, band_width=50, pre_align_subset_contigs=True)
    opts.update(kw)
    al = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq("t", target)], device=0)
    return [pre if pre is not None else 0 for _, pre in al.align(reads)]


def want_of(target, reads, k=14, w=50, **sc):
    return [banded(r, target, k=k, w=w, **sc) for r in reads]


@pytest.mark.parametrize("n", [1, 5, 39, 40, 41, 127, 128, 129, 255, 1024, 1025, 2047, 3073, 4097, 5119, 5120])
def test_full_matrix_pairs_by_target_length(n, monkeypatch):
    """target lengths around the strips' edges (128 strips of RP rows, RP = 8, 16, 24, 32, 40), reads of ragged lengths"""
    rng = random.Random(1000 + n)
    target = rnd(rng, n)
    reads = []
    for m in (1, 2, 11, 63, 64, 65, 127, 128, 129, 200, 777, 1500):
        a = rng.randrange(0, max(1, n - m)) if n > m else 0
        reads.append(seedless_relative(rng, (target[a:a + m] + rnd(rng, m))[:m], 12))
    reads.append(seedless_relative(rng, target, 12))                          # end to end
    reads.append(rnd(rng, 900))                                                # unrelated
    want = want_of(target, reads)
    assert max(want) >= min(n, 40) // 3                                        # (the related reads do score)
    assert scores_of(target, reads) == want
    monkeypatch.setenv("STITCH_PREALIGN_V1", "1")
    assert scores_of(target, reads) == want


@pytest.mark.parametrize("sc", [dict(match=2, mismatch=-3, go=-5, ge=-1), dict(match=1, mismatch=-1, go=0, ge=-1), dict(match=3, mismatch=-7, go=-11, ge=-3),
                                dict(match=1, mismatch=-4, go=-6, ge=0), dict(match=6, mismatch=-1, go=-2, ge=-2)])
def test_full_matrix_pairs_other_scorings(sc):
    """(match = 6 on a 5 kb target: 6 * 5000 < 32 000 still fits the 16-bit words)"""
    rng = random.Random(77)
    target = rnd(rng, 5000)
    reads = [seedless_relative(rng, target[a:a + m], 12) for a, m in ((0, 5000), (1000, 2500), (4000, 1000), (10, 300))] + [rnd(rng, 1200)]
    kw = dict(match_score=sc["match"], mismatch_score=sc["mismatch"], gap_open=sc["go"], gap_extend=sc["ge"])
    assert scores_of(target, reads, **kw) == want_of(target, reads, **sc)


def test_full_matrix_scores_beyond_16_bits_take_the_32_bit_kernel():
    """match = 7: 7 * 5000 > 32 000, the packed kernel must decline (the 32-bit register kernel scores the pairs)"""
    rng = random.Random(78)
    target = rnd(rng, 5000)
    reads = [seedless_relative(rng, target, 12), seedless_relative(rng, target[500:4000], 12)]
    sc = dict(match=7, mismatch=-9, go=-6, ge=-2)
    kw = dict(match_score=7, mismatch_score=-9, gap_open=-6, gap_extend=-2)
    want = want_of(target, reads, **sc)
    assert max(want) > 7000
    assert scores_of(target, reads, **kw) == want


def test_full_matrix_long_reads_and_many_pairs():
    """10 kb reads against 5 kb targets (the cfg3 shape) on both strands of several targets: pair indexing across chunks and batches"""
    rng = random.Random(79)
    targets = [(f"t{k}", rnd(rng, rng.choice([5000, 4990, 3000]))) for k in range(4)]
    reads = []
    for k in range(70):                                                        # more than one chunk of 64
        t = targets[k % 4][1]
        reads.append((rnd(rng, 3000) + seedless_relative(rng, t[500:4500], 12) + rnd(rng, 3000))[:10000])
    opts = dict(pre_align=True, pre_align_min_score=150, kmer_size=14, band_width=50, double_strand=True)
    al = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0)
    got = al.align(reads)
    comp = str.maketrans("ACGT", "TGCA")
    for k in (0, 1, 2, 3, 65, 69):
        per = []
        for n, s in targets:
            per += [banded(reads[k], s, k=14), banded(reads[k], s.translate(comp)[::-1], k=14)]
        kept = [v for v in per if v >= 150]
        assert kept and got[k][1] == max(kept), (k, per, got[k][1])


# ---- banded_score_window_kernel (prealign_window.hip): the band's rows in a register window of 256 rows ---------------------------
def noisy(rng, s, sub=0.04, ins=0.02, dele=0.02):
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        out.append(rng.choice("ACGT") if r < dele + sub else ch)
        if rng.random() < ins:
            out.append(rng.choice("ACGT"))
    return "".join(out)


def banded_cases(rng, target):
    n = len(target)
    reads = [target,                                                               # the diagonal from (0, 0): row 0 in the band of every column
             noisy(rng, target[n // 3:]),                                          # starts in mid-target: empty columns on the left
             rnd(rng, 700) + noisy(rng, target[100:n - 200]) + rnd(rng, 500),      # the band far from row 0: the window slides 700 rows before its first column
             noisy(rng, target[:n // 2], 0.08, 0.04, 0.04),
             noisy(rng, target[50:400]), target[n - 60:], target[:40],
             noisy(rng, target[200:900]) + noisy(rng, target[1500:2400]),          # a deletion of 600: the backbone's gap is interpolated, columns taller than the window
             noisy(rng, target[200:900]) + rnd(rng, 900) + noisy(rng, target[900:1800]),      # an insertion of 900: the window must slide 900 rows at once
             rnd(rng, 400)]
    return reads


@pytest.mark.parametrize("w", [3, 20, 50, 62, 63, 100])
def test_banded_pairs_by_band_width(w, monkeypatch):
    """w <= 62: columns of 4 w + 1 <= 249 rows fit the window; w = 63 and 100 do not, their pairs take the LDS-ring kernel"""
    rng = random.Random(2000 + w)
    target = rnd(rng, 3000)
    reads = banded_cases(rng, target)
    want = want_of(target, reads, k=10, w=w)
    assert sum(v > 100 for v in want) >= 6
    assert scores_of(target, reads, kmer_size=10, band_width=w) == want
    monkeypatch.setenv("STITCH_PREALIGN_V1", "1")
    assert scores_of(target, reads, kmer_size=10, band_width=w) == want


@pytest.mark.parametrize("sc", [dict(match=2, mismatch=-3, go=-5, ge=-1), dict(match=1, mismatch=-1, go=0, ge=-1), dict(match=5, mismatch=-4, go=-10, ge=-1),
                                dict(match=1, mismatch=-4, go=-6, ge=0)])
def test_banded_pairs_other_scorings(sc):
    rng = random.Random(91)
    target = rnd(rng, 2500)
    reads = banded_cases(rng, target)
    kw = dict(match_score=sc["match"], mismatch_score=sc["mismatch"], gap_open=sc["go"], gap_extend=sc["ge"], kmer_size=10, band_width=40)
    assert scores_of(target, reads, **kw) == want_of(target, reads, k=10, w=40, **sc)


@pytest.mark.parametrize("seed", range(8))
def test_banded_pairs_random_shapes(seed):
    """ragged target and read lengths, several targets on both strands, random k and w: the scores decide which contigs are kept"""
    rng = random.Random(3000 + seed)
    targets = [(f"t{k}", rnd(rng, rng.choice([37, 300, 1023, 1024, 2600]))) for k in range(rng.randint(1, 4))]
    k, w = rng.choice([8, 10, 12]), rng.choice([5, 25, 50, 60])
    reads = []
    for _ in range(12):
        t = rng.choice(targets)[1]
        a = rng.randrange(0, max(1, len(t) - 30)); b = rng.randrange(a + 1, len(t) + 1)
        seg = noisy(rng, t[a:b], rng.choice([0.0, 0.03, 0.1]), 0.02, 0.02)
        reads.append(rnd(rng, rng.choice([0, 0, 5, 300])) + seg + rnd(rng, rng.choice([0, 0, 7, 200])))
    reads = [r for r in reads if r]
    opts = dict(pre_align=True, pre_align_min_score=20, kmer_size=k, band_width=w, double_strand=True)
    al = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0)
    got = al.align(reads)
    comp = str.maketrans("ACGT", "TGCA")
    for i, r in enumerate(reads):
        per = []
        for n, s in targets:
            per += [banded(r, s, k=k, w=w), banded(r, s.translate(comp)[::-1], k=k, w=w)]
        kept = [v for v in per if v >= 20]
        assert got[i][1] == (max(kept) if kept else None), (i, per, got[i][1])
