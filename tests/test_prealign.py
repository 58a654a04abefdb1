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
    with pytest.raises(RuntimeError):
        orc.Aligners(targets, pre_align=True, mode="global").align(read)


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
