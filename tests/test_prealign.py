"""A15: the banded pre-alignment filter (`--pre-align`).  The arithmetic belongs to crate bio 1.1.0, which is not in the
reference tree and which no reference test pins: PARITY UNPINNED.  The oracle (oracle/prealign_oracle.cpp) restates the
crate's published algorithm; these tests pin the oracle's definition with known answers and compare the product (host
chaining + banded kernel) with it."""
import ctypes as C
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402


def banded(x, y, k=12, w=50, match=1, mismatch=-4, go=-6, ge=-2):
    L = orc.lib()
    L.orc_banded_local_score.restype = C.c_long
    xb = (C.c_uint8 * max(1, len(x))).from_buffer_copy((x or "\0").encode())
    yb = (C.c_uint8 * max(1, len(y))).from_buffer_copy((y or "\0").encode())
    return L.orc_banded_local_score(xb, C.c_size_t(len(x)), yb, C.c_size_t(len(y)), C.c_size_t(k), C.c_size_t(w), match, mismatch, go, ge)


def sw(x, y, match=1, mismatch=-4, go=-6, ge=-2):
    """plain affine-gap Smith-Waterman score (Gotoh), the band-free answer"""
    NEG = -10**9
    H = [0] * (len(y) + 1); D = [NEG] * (len(y) + 1)
    best = 0
    for i in range(1, len(x) + 1):
        diag, H[0], I = H[0], 0, NEG
        for j in range(1, len(y) + 1):
            D[j] = max(D[j] + ge, H[j] + go + ge)
            I = max(I + ge, H[j - 1] + go + ge)
            h = max(0, diag + (match if x[i - 1] == y[j - 1] else mismatch), D[j], I)
            diag, H[j] = H[j], h
            best = max(best, h)
    return best


def rnd(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, s, sub=0.03, ins=0.02, dele=0.02):
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        out.append(rng.choice("ACGT") if r < dele + sub else ch)
        if rng.random() < ins:
            out.append(rng.choice("ACGT"))
    return "".join(out)


def test_known_answers():
    rng = random.Random(3)
    y = rnd(rng, 400)
    assert banded(y[100:300], y) == 200                       # exact substring: every base matches
    assert banded(rnd(rng, 50) + y[100:300] + rnd(rng, 70), y) >= 200
    assert banded("ACGT", y, k=12) == sw("ACGT", y)           # no 12-mer fits: full matrix
    assert banded("", y) == 0 and banded(y, "") == 0
    x = y[50:150] + y[160:260]                                # one 10-base deletion: 200 matches - (6 + 2*10)
    assert banded(x, y) == sw(x, y) == 200 - 26


@pytest.mark.parametrize("seed", range(12))
def test_band_covers_the_optimum_on_noisy_reads(seed):
    """With a reasonable seed density and w above the path's excursion the banded score equals Smith-Waterman (the
    property the crate documents); a band can never beat it."""
    rng = random.Random(100 + seed)
    y = rnd(rng, rng.randint(300, 700))
    a = rng.randint(0, len(y) // 3); b = rng.randint(2 * len(y) // 3, len(y))
    x = rnd(rng, rng.randint(0, 60)) + mutate(rng, y[a:b]) + rnd(rng, rng.randint(0, 60))
    full = sw(x, y)
    assert banded(x, y, k=8, w=30) == full
    assert banded(x, y, k=8, w=0) <= full
    other = rnd(rng, 500)
    assert banded(x, other, k=8, w=30) <= sw(x, other)


def test_oracle_aligners_prealign_logic():
    rng = random.Random(9)
    targets = [(f"t{k}", rnd(rng, 500)) for k in range(4)]
    read = rnd(rng, 30) + mutate(rng, targets[2][1][100:400]) + rnd(rng, 30)
    o = orc.Aligners(targets, pre_align=True, pre_align_min_score=60, kmer_size=10, band_width=20)
    chains = o.align(read)
    assert len(chains) == 1 and chains[0].start_contig_idx == 2 and o.prealign_score() == banded(read, targets[2][1], k=10, w=20)
    assert o.align(rnd(rng, 200)) == [] and o.prealign_score() is None          # nothing passes: unmapped, no score
    # without sub-setting every contig takes part again, and the loop stops at the first passing target
    o2 = orc.Aligners(targets, pre_align=True, pre_align_min_score=60, pre_align_subset_contigs=False, kmer_size=10, band_width=20)
    o3 = orc.Aligners(targets)
    assert [c.key() for c in o2.align(read)] == [c.key() for c in o3.align(read)]
    # (round 4: every clipping mode is restated; in global mode the read must cover a whole target: this one does not reach the threshold)
    og = orc.Aligners(targets, pre_align=True, mode="global", pre_align_min_score=60, kmer_size=10, band_width=20)
    assert og.align(read) == [] and og.prealign_score() is None


def _bands(x, y, k, w, match=1, go=-6, ge=-2):
    import numpy as np
    from stitch_amd import api
    xb = (C.c_uint8 * max(1, len(x))).from_buffer_copy((x or "\0").encode()); yb = (C.c_uint8 * max(1, len(y))).from_buffer_copy((y or "\0").encode())
    lo = np.zeros(len(y) + 1, dtype=np.uint16); hi = np.zeros(len(y) + 1, dtype=np.uint16)
    L = api.lib()
    full_p = L.stitch_prealign_band(xb, len(x), yb, len(y), k, w, match, go, ge, lo.ctypes.data_as(C.POINTER(C.c_uint16)), hi.ctypes.data_as(C.POINTER(C.c_uint16)))
    assert full_p >= 0
    olo = np.zeros(len(y) + 1, dtype=np.uint32); ohi = np.zeros(len(y) + 1, dtype=np.uint32)
    full_o = orc.lib().orc_banded_band(xb, C.c_size_t(len(x)), yb, C.c_size_t(len(y)), C.c_size_t(k), C.c_size_t(w), match, go, ge,
                                      olo.ctypes.data_as(C.POINTER(C.c_uint32)), ohi.ctypes.data_as(C.POINTER(C.c_uint32)))
    return (full_p, lo.astype(np.uint32), hi.astype(np.uint32)), (full_o, olo, ohi)


def test_the_band_entry_point_rejects_a_positive_gap_score():
    """the backbone's pruned search assumes a penalty that grows with the gap (the context constructor rejects positive gap scores too)"""
    import numpy as np
    from stitch_amd import api
    x = "ACGTACGTACGTTTGACCA"; y = "ACGTACGTACGTTTGACCAGGT"
    xb = (C.c_uint8 * len(x)).from_buffer_copy(x.encode()); yb = (C.c_uint8 * len(y)).from_buffer_copy(y.encode())
    lo = np.zeros(len(y) + 1, dtype=np.uint16); hi = np.zeros(len(y) + 1, dtype=np.uint16)
    P16 = C.POINTER(C.c_uint16)
    for go, ge in [(1, -2), (-6, 1)]:
        rc = api.lib().stitch_prealign_band(xb, len(x), yb, len(y), 5, 3, 1, go, ge, lo.ctypes.data_as(P16), hi.ctypes.data_as(P16))
        assert rc == -1 and b"positive" in api.lib().stitch_last_error()
    assert api.lib().stitch_prealign_band(xb, len(x), yb, len(y), 5, 3, 1, 0, 0, lo.ctypes.data_as(P16), hi.ctypes.data_as(P16)) >= 0


@pytest.mark.parametrize("seed", range(240))
def test_product_band_equals_oracle_band(seed):
    """the library's host code (sorted k-mer index, analytic diagonal rasterisation) and the oracle's (maps, point by point)
    must produce the same band, column by column: seeds, backbone chain with its tie-breaks, gaps, extensions"""
    import numpy as np
    rng = random.Random(300 + seed)
    y = rnd(rng, rng.randint(50, 900))
    kind = seed % 4
    if kind == 0:
        x = rnd(rng, rng.randint(20, 600))                                   # unrelated: a few random seeds or none
    elif kind == 1:
        a = rng.randint(0, len(y) // 2); x = rnd(rng, rng.randint(0, 80)) + mutate(rng, y[a:a + rng.randint(40, 400)], 0.05, 0.03, 0.03) + rnd(rng, rng.randint(0, 80))
    elif kind == 2:                                                          # two segments with a big gap, repeats
        a = rng.randint(0, len(y) // 3); b = rng.randint(len(y) // 2, len(y) - 10)
        x = y[a:a + 60] + rnd(rng, rng.randint(0, 200)) + y[b:b + 60] + y[a:a + 40]
    else:
        unit = rnd(rng, rng.randint(3, 9)); y = (unit * 80)[:len(y)] if rng.random() < 0.5 else y   # low complexity: many seeds, many ties
        x = (unit * 40)[:rng.randint(30, 250)]
    k = rng.choice([4, 6, 8, 12]); w = rng.choice([0, 3, 20, 50])
    go, ge = rng.choice([(-6, -2), (0, -1), (-3, -3)])
    p, o = _bands(x, y, k, w, match=rng.choice([1, 2]), go=go, ge=ge)
    # (the library's flag says "no seed: full matrix", the oracle's "the band covers the matrix": a wide band around a real backbone
    # may cover it too)
    assert (not p[0]) or o[0]
    assert np.array_equal(p[1], o[1]) and np.array_equal(p[2], o[2])


@pytest.mark.parametrize("seed", range(40))
def test_product_band_with_bytes_other_than_acgt(seed):
    """k-mers holding N (or any other byte) match by their bytes: the 2-bit table of the four-letter k-mers and the byte-hash index
    beside it must together give the oracle's seeds; k = 33 has no table at all"""
    import numpy as np
    rng = random.Random(7000 + seed)
    alpha = "ACGTN" if seed % 2 else "ACGTNRY"
    y = "".join(rng.choice(alpha if rng.random() < 0.08 else "ACGT") for _ in range(rng.randint(80, 700)))
    a = rng.randint(0, len(y) // 2)
    x = rnd(rng, rng.randint(0, 60)) + mutate(rng, y[a:a + rng.randint(40, 300)], 0.03, 0.01, 0.01) + "".join(rng.choice(alpha) for _ in range(rng.randint(0, 50)))
    k = rng.choice([3, 5, 8, 12, 33]) if seed % 5 else 33
    p, o = _bands(x, y, k, rng.choice([2, 20]))
    assert (not p[0]) or o[0]
    assert np.array_equal(p[1], o[1]) and np.array_equal(p[2], o[2])


def _device_band(x, y, k, w, match=1, go=-6, ge=-2):
    import numpy as np
    from stitch_amd import api
    xb = (C.c_uint8 * max(1, len(x))).from_buffer_copy((x or "\0").encode()); yb = (C.c_uint8 * max(1, len(y))).from_buffer_copy((y or "\0").encode())
    lo = np.zeros(len(y) + 1, dtype=np.uint16); hi = np.zeros(len(y) + 1, dtype=np.uint16)
    cls = C.c_uint32(99)
    full = api.lib().stitch_prealign_band_device(0, xb, len(x), yb, len(y), k, w, match, go, ge, lo.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                 hi.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(cls))
    assert full >= 0, api.lib().stitch_last_error()
    return full, lo.astype(np.int64), hi.astype(np.int64), cls.value


def _kernel_class(lo, hi, m):
    """which score kernel a band goes to (prealign_window.hip band_fits_window, prealign_kernel.hip BAND_RING), restated"""
    prev, window = 0, True
    for c in range(1, len(lo)):
        r0, r1 = max(int(lo[c]), 1), min(int(hi[c]), m + 1)
        if r0 >= r1:
            continue
        if r0 < prev or r1 - ((r0 - 1) & ~3) > 256:
            window = False
        prev = max(prev, r0)
    if window:
        return 3
    return 2 if any(h > l and h - l > 512 for l, h in zip(lo, hi)) else 0


@pytest.mark.gpu
def test_device_band_equals_oracle_band():
    """the band as the DEVICE draws it from the backbone's pieces (prealign_band.hip: what production uses) against the oracle's,
    column by column, on the inputs of the two host tests above and on long gaps; and the score kernel it names"""
    import numpy as np
    cases = []
    for seed in range(240):
        rng = random.Random(300 + seed)
        y = rnd(rng, rng.randint(50, 900))
        kind = seed % 4
        if kind == 0:
            x = rnd(rng, rng.randint(20, 600))
        elif kind == 1:
            a = rng.randint(0, len(y) // 2); x = rnd(rng, rng.randint(0, 80)) + mutate(rng, y[a:a + rng.randint(40, 400)], 0.05, 0.03, 0.03) + rnd(rng, rng.randint(0, 80))
        elif kind == 2:
            a = rng.randint(0, len(y) // 3); b = rng.randint(len(y) // 2, len(y) - 10)
            x = y[a:a + 60] + rnd(rng, rng.randint(0, 200)) + y[b:b + 60] + y[a:a + 40]
        else:
            unit = rnd(rng, rng.randint(3, 9)); y = (unit * 80)[:len(y)] if rng.random() < 0.5 else y
            x = (unit * 40)[:rng.randint(30, 250)]
        k = rng.choice([4, 6, 8, 12]); w = rng.choice([0, 3, 20, 50])
        go, ge = rng.choice([(-6, -2), (0, -1), (-3, -3)])
        cases.append((x, y, k, w, rng.choice([1, 2]), go, ge))
    rng = random.Random(99)
    t = rnd(rng, 4000)
    for gap_x, gap_y in ((0, 600), (600, 0), (400, 150), (3, 500), (500, 3), (150, 0), (0, 150), (1, 1), (2, 0), (0, 2), (64, 65)):      # long and lopsided gaps between two exact stretches (the backbone keeps both: 1500 matches pay for the gap)
        x = t[100:1600] + rnd(rng, gap_x) + t[1600 + gap_y:3100 + gap_y]
        for w in (0, 7, 50):
            cases.append((x, t, 12, w, 1, -6, -2))
    cases.append((rnd(rng, 3000) + t[:300], t, 12, 50, 1, -6, -2))              # the band starts 3000 rows down
    cases.append((t[3700:], t, 12, 50, 1, -6, -2))                               # ... and 3700 columns in
    n_window = n_other = 0
    for x, y, k, w, match, go, ge in cases:
        full, lo, hi, cls = _device_band(x, y, k, w, match, go, ge)
        _, o = _bands(x, y, k, w, match=match, go=go, ge=ge)
        assert (not full) or o[0]
        assert np.array_equal(lo, o[1]) and np.array_equal(hi, o[2]), (len(x), len(y), k, w)
        if not full:
            assert cls == _kernel_class(lo, hi, len(x)), (len(x), len(y), k, w, cls)
            n_window += cls == 3; n_other += cls != 3
    assert n_window > 100 and n_other > 5



def banded_mode(x, y, mode, k=12, w=50, match=1, mismatch=-4, go=-6, ge=-2):
    """the oracle's banded score under the clip penalties of a clipping mode (x = the query: Options::banded_scoring, mod.rs:133-141)"""
    L = orc.lib()
    L.orc_banded_score.restype = C.c_long
    MIN = orc.MIN_SCORE
    xp = xs = MIN if mode in ("query-local", "global") else 0
    yp = ys = MIN if mode in ("target-local", "global") else 0
    xb = (C.c_uint8 * max(1, len(x))).from_buffer_copy((x or "\0").encode())
    yb = (C.c_uint8 * max(1, len(y))).from_buffer_copy((y or "\0").encode())
    return L.orc_banded_score(xb, C.c_size_t(len(x)), yb, C.c_size_t(len(y)), C.c_size_t(k), C.c_size_t(w), match, mismatch, go, ge, xp, xs, yp, ys)


def gotoh_mode(x, y, mode, match=1, mismatch=-4, go=-6, ge=-2):
    """band-free affine-gap score with end gaps free on the sequence a mode clips freely: written independently of the oracle's
    formulation (explicit start / end states instead of clip candidates)"""
    NEG = -10**12
    free_x = mode in ("local", "target-local")      # the query may be clipped
    free_y = mode in ("local", "query-local")       # the target may be clipped
    m, n = len(x), len(y)
    gap = lambda k: 0 if k == 0 else go + ge * k
    H = [[NEG] * (n + 1) for _ in range(m + 1)]; E = [[NEG] * (n + 1) for _ in range(m + 1)]; F = [[NEG] * (n + 1) for _ in range(m + 1)]
    for i in range(m + 1):
        for j in range(n + 1):
            if i == 0 and j == 0:
                H[i][j] = 0
                continue
            start = (0 if (free_x or i == 0) else gap(i)) + (0 if (free_y or j == 0) else gap(j))      # everything before (i, j) skipped
            best = start if ((free_x or i == 0) or (free_y or j == 0) or True) else NEG
            if i > 0:
                F[i][j] = max(F[i - 1][j] + ge, H[i - 1][j] + go + ge); best = max(best, F[i][j])
            if j > 0:
                E[i][j] = max(E[i][j - 1] + ge, H[i][j - 1] + go + ge); best = max(best, E[i][j])
            if i > 0 and j > 0:
                best = max(best, H[i - 1][j - 1] + (match if x[i - 1] == y[j - 1] else mismatch))
            H[i][j] = best
    out = NEG
    for i in range(m + 1):
        for j in range(n + 1):
            tail = (0 if (free_x or i == m) else gap(m - i)) + (0 if (free_y or j == n) else gap(n - j))
            out = max(out, H[i][j] + tail)
    return out


def test_clipping_modes_known_answers():
    """the general banded score (every clipping mode): with the band covering the whole matrix (no k-mer fits) it is the band-free
    score of the mode, written twice; known small cases by hand"""
    rng = random.Random(11)
    # by hand: query ACGT inside target TTACGTTT.  local / query-local: 4 matches = 4.  target-local: the target's other four bases
    # must be gapped: 4 + (go + 2 ge) * 2 = 4 - 20 = -16 ... or align nothing of the query and delete the whole target: go + 8 ge = -22.
    assert banded_mode("ACGT", "TTACGTTT", "local", k=12) == 4
    assert banded_mode("ACGT", "TTACGTTT", "query-local", k=12) == 4
    assert banded_mode("ACGT", "TTACGTTT", "target-local", k=12) == gotoh_mode("ACGT", "TTACGTTT", "target-local") == -16
    assert banded_mode("ACGT", "TTACGTTT", "global", k=12) == gotoh_mode("ACGT", "TTACGTTT", "global") == -16
    assert banded_mode("ACGTACGT", "ACGAACGT", "global", k=12) == 7 - 4                     # one substitution
    assert banded_mode("", "ACG", "global") == -12 and banded_mode("ACG", "", "query-local") == -12 and banded_mode("ACG", "", "target-local") == 0
    for trial in range(30):
        x = rnd(rng, rng.randint(1, 40)); y = rnd(rng, rng.randint(1, 40))
        if trial % 3 == 0:
            y = y[:10] + x + y[10:]
        for mode in ("local", "query-local", "target-local", "global"):
            assert banded_mode(x, y, mode, k=60) == gotoh_mode(x, y, mode), (x, y, mode)
    # local is the old definition
    for trial in range(10):
        y = rnd(rng, 300); x = mutate(rng, y[40:260])
        assert banded_mode(x, y, "local", k=8, w=20) == banded(x, y, k=8, w=20)
    # a band that covers the path: the banded score is the band-free one; a band that does not: never above it
    y = rnd(rng, 500); x = mutate(rng, y[100:420], 0.02, 0.01, 0.01)
    for mode in ("query-local", "target-local", "global"):
        assert banded_mode(x, y, mode, k=8, w=40) == gotoh_mode(x, y, mode)
        assert banded_mode(x, y, mode, k=8, w=0) <= gotoh_mode(x, y, mode)

# ---- product vs oracle ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(double_strand=True), dict(pre_align_subset_contigs=False, double_strand=True),
                                  dict(suboptimal=True, double_strand=True), dict(kmer_size=8, band_width=5, pre_align_min_score=30),
                                  dict(circular=True)])
def test_prealign_matches_oracle(opts):
    import stitch_amd
    from stitch_amd import synth
    rng = random.Random(21)
    db = synth.make_db(6, 700, 5)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 14, 400, 8, both_strands=opts.get("double_strand", False), random_frac=0.3)]
    reads += [rnd(rng, 40), targets[1][1][10:150], "ACGT" * 30]
    base = dict(pre_align=True, pre_align_min_score=50, kmer_size=10, band_width=25)
    base.update(opts)
    al = stitch_amd.Builder(**base).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0)
    got = al.align(reads)
    o = orc.Aligners(targets, **base)
    n_unmapped = 0
    for k, read in enumerate(reads):
        want = o.align(read)
        ps = o.prealign_score()
        assert got[k][1] == ps, (k, got[k][1], ps)
        assert [c.key() for c in got[k][0]] == [c.key() for c in want], k
        n_unmapped += ps is None
        if want and all(len(c.ops) for c in want):
            assert al.format_sam(k, f"r{k}", read, "I" * len(read)) == o.format_sam(f"r{k}", read, "I" * len(read), prealign=ps)
    assert 0 < n_unmapped < len(reads)


@pytest.mark.gpu
def test_banded_kernel_wide_bands_and_long_targets():
    """bands wider than one 64-row chunk, reads without seeds (full matrix), chunked scratch"""
    import stitch_amd
    rng = random.Random(5)
    targets = [("a", rnd(rng, 3000)), ("b", rnd(rng, 1500))]
    reads = [mutate(rng, targets[0][1][200:2600], 0.05, 0.04, 0.04), rnd(rng, 300), targets[1][1][100:1400], mutate(rng, targets[1][1], 0.1, 0.05, 0.05)]
    os.environ["STITCH_PREALIGN_BYTES"] = str(3 << 20)
    try:
        for w in (3, 40, 200):
            kw = dict(pre_align=True, pre_align_min_score=1, kmer_size=9, band_width=w)
            al = stitch_amd.Builder(**kw).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0)
            got = al.align(reads)
            o = orc.Aligners(targets, **kw)
            for k, read in enumerate(reads):
                o.align(read)
                assert got[k][1] == o.prealign_score(), (w, k)
    finally:
        del os.environ["STITCH_PREALIGN_BYTES"]


@pytest.mark.gpu
def test_banded_kernels_agree_ring_tall_and_global(monkeypatch):
    """the LDS-ring kernel, the global-state kernel it replaces (STITCH_BANDED_GLOBAL) and the oracle give the same scores:
    w = 20 gives columns of one or two 64-row blocks, w = 200 of several, and w = 300 makes columns taller than the ring
    (1201 > 512 rows for the first read), which must take the global-state kernel"""
    import stitch_amd
    rng = random.Random(17)
    t0, t1 = rnd(rng, 2600), rnd(rng, 900)
    targets = [("a", t0), ("b", t1)]
    reads = [t0[100:1100] + rnd(rng, 1500) + t0[1100:2300],                  # query gap of 1500: a tall column
             mutate(rng, t0[50:2500], 0.04, 0.03, 0.03),
             t1[20:880], mutate(rng, t1, 0.08, 0.04, 0.04) + rnd(rng, 200), rnd(rng, 500)]
    for w in (20, 200, 300):
        kw = dict(pre_align=True, pre_align_min_score=1, kmer_size=10, band_width=w, double_strand=True)
        o = orc.Aligners(targets, **kw)
        want = []
        for read in reads:
            o.align(read)
            want.append(o.prealign_score())
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("STITCH_BANDED_GLOBAL", env)
            else:
                monkeypatch.delenv("STITCH_BANDED_GLOBAL", raising=False)
            al = stitch_amd.Builder(**kw).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0)
            assert [g[1] for g in al.align(reads)] == want, (w, env)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["query-local", "target-local", "global"])
def test_prealign_in_the_other_clipping_modes_matches_oracle(mode):
    """`--pre-align` with `--mode`: the reference hands the mode's clip penalties to the banded scorer (Options::banded_scoring,
    mod.rs:133-141).  Scores of the general kernel against the oracle's, then kept contigs and chains (parity UNPINNED: both sides
    restate bio's documented penalties)."""
    import stitch_amd
    rng = random.Random(31)
    targets = [("a", rnd(rng, 260)), ("b", rnd(rng, 180)), ("c", rnd(rng, 220))]
    reads = [mutate(rng, targets[0][1][20:240]), mutate(rng, targets[1][1]), targets[2][1][5:215], rnd(rng, 120),
             mutate(rng, targets[0][1]) , targets[1][1][:170]]
    kw = dict(mode=mode, pre_align=True, pre_align_min_score=-400, kmer_size=8, band_width=15, double_strand=True)
    o = orc.Aligners(targets, **kw)
    al = stitch_amd.Builder(**kw).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0)
    try:
        got = al.align(reads)
    except stitch_amd.StitchError as e:
        assert "shorter contig" in str(e)          # (the reference-undefined end-of-read jump of these modes: DESIGN.md 2)
        pytest.skip("reference-undefined XJUMP case")
    for k, read in enumerate(reads):
        try:
            want = o.align(read)
        except RuntimeError as e:
            assert "out of range" in str(e)
            pytest.skip("reference-undefined XJUMP case (oracle)")
        assert got[k][1] == o.prealign_score(), (mode, k, got[k][1], o.prealign_score())
        assert [c.key() for c in got[k][0]] == [c.key() for c in want], (mode, k)
    # the scores alone, pair by pair (one target, threshold below everything: the reported score is the pair's)
    for tname, tseq in targets:
        for read in reads:
            kw1 = dict(mode=mode, pre_align=True, pre_align_min_score=-10**6, kmer_size=8, band_width=15)
            al1 = stitch_amd.Builder(**kw1).build_aligners([stitch_amd.TargetSeq(tname, tseq)], device=0)
            try:
                sc = al1.align([read])[0][1]
            except stitch_amd.StitchError:
                continue
            assert sc == banded_mode(read, tseq, mode, k=8, w=15), (mode, tname)


@pytest.mark.gpu
def test_prealign_general_path_equals_the_fast_path_in_local_mode(monkeypatch):
    """STITCH_PREALIGN_GENERAL=1 sends a Local run through the general kernel (32-bit band ranges, clip penalties all zero): same
    scores, kept contigs and chains as the fast kernels and as the oracle"""
    import stitch_amd
    from stitch_amd import synth
    db = synth.make_db(5, 600, 5)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 10, 350, 8, both_strands=True, random_frac=0.3)]
    kw = dict(pre_align=True, pre_align_min_score=50, kmer_size=10, band_width=25, double_strand=True)
    fast = stitch_amd.Builder(**kw).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0).align(reads)
    monkeypatch.setenv("STITCH_PREALIGN_GENERAL", "1")
    gen = stitch_amd.Builder(**kw).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets], device=0).align(reads)
    assert [(g[1], [c.key() for c in g[0]]) for g in gen] == [(f[1], [c.key() for c in f[0]]) for f in fast]


@pytest.mark.gpu
def test_prealign_of_a_read_beyond_65534_bases():
    """the fast path's band ranges are 16-bit; a longer read takes the general path (32-bit ranges) instead of being refused"""
    import stitch_amd
    rng = random.Random(41)
    t = rnd(rng, 900)
    read = rnd(rng, 33000) + mutate(rng, t[100:800]) + rnd(rng, 33000)
    kw = dict(pre_align=True, pre_align_min_score=100, kmer_size=12, band_width=30)
    al = stitch_amd.Builder(**kw).build_aligners([stitch_amd.TargetSeq("t", t)], device=0)
    got = al.align([read])
    assert len(read) > 65534 and got[0][1] == banded(read, t, k=12, w=30) and got[0][1] >= 300
    assert got[0][0] and got[0][0][0].score >= 300
