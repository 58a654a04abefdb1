"""Parity tests proper: the HIP path (through the C ABI, stitch_amd.api) against the reference's golden vectors and
against the oracle on the same seeded inputs.  Bit-exact: scores, coordinates, contig indexes, op lists, SAM text.
Run on the GPU box with `pytest -m gpu`."""
import json
import os
import random

import pytest

import stitch_amd
from oracle import oracle as orc
from stitch_amd import synth

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SINGLE = json.load(open(os.path.join(G, "single_contig.json")))
MULTI = json.load(open(os.path.join(G, "multi_contig.json")))
MODE = {"local": "local", "querylocal": "query-local", "targetlocal": "target-local", "global": "global"}


def rc(seq):
    comp = dict(zip("AGCTYRWSKMDVHBN", "TCGARYWSMKHBDVN"))               # util/dna.rs:5-8
    comp.update({a.lower(): b.lower() for a, b in list(comp.items())})
    return "".join(comp[c] for c in reversed(seq))


def resolve(seq):
    return rc(seq[3:]) if seq.startswith("rc:") else seq


def summary(a, drop=()):
    b = stitch_amd.Alignment()
    for f in stitch_amd.Alignment.__slots__[:-1]:
        setattr(b, f, getattr(a, f))
    b.operations = [o for o in a.operations if o[0] not in drop]
    return (b.xstart, b.xend, b.ystart, b.yend, b.score, b.start_contig_idx, b.cigar(), b.length)


def expected(e):
    return (e["xstart"], e["xend"], e["ystart"], e["yend"], e["score"], e["start_contig_idx"], e["cigar"], e["length"])


def test_golden_single_contig():  # single_contig_aligner.rs:915-1773, all 63, one context each
    for t in SINGLE:
        s = t["scoring"]
        al = stitch_amd.Builder(mode=MODE[t["mode"]], match_score=s["match"], mismatch_score=s["mismatch"], gap_open=s["gap_open"],
                                gap_extend=s["gap_extend"], default_jump_score=s["jump"], circular=t["circular"],
                                keep_clipping=True).build_aligners([stitch_amd.TargetSeq("x", t["x"])])
        chains, _ = al.align_one(t["y"])
        drop = {"local": (4, 5), "querylocal": (5,), "targetlocal": (4,), "global": ()}[t["mode"]]
        assert summary(chains[0], drop) == expected(t["expect"]), (t["name"], chains[0])


def test_golden_multi_contig():  # multi_contig_aligner.rs:465-666 (the tests whose contigs are plain targets)
    for t in MULTI:
        if t["name"] == "test_jump_scores":
            continue   # its aligner order (fwd, rev, fwd) cannot be built through Builder::build_aligners; see below
        c0 = t["contigs"][0]
        mismatch, go, ge, jump = c0["scoring"]
        al = stitch_amd.Builder(mode="global" if c0["kind"] == "global" else "local", match_score=1, mismatch_score=mismatch,
                                gap_open=go, gap_extend=ge, default_jump_score=jump, keep_clipping=True) \
            .build_aligners([stitch_amd.TargetSeq(c["name"], resolve(c["seq"])) for c in t["contigs"]])
        chains, _ = al.align_one(resolve(t["y"]))
        assert summary(chains[0]) == expected(t["cases"][0]["expect"]), (t["name"], chains[0])


def test_golden_jump_score_priorities():  # multi_contig_aligner.rs:668-737
    """The reference adds its aligners as (chr1 fwd, chr1 rev, chr2 fwd); Builder::build_aligners orders a double-strand
    database as (chr1 fwd, chr2 fwd, chr1 rev, chr2 rev).  chr2's reverse strand (TTTTT) shares nothing with the read, and the
    five cases are decided by the jump scores' priority (same > opposite > inter), not by aligner order — so the GOLDEN
    expectation must come out once the product's contig indexes are renamed to the reference's (0 -> 0, 2 -> 1, 1 -> 2); the
    oracle, built like the product, is compared as well."""
    t = [x for x in MULTI if x["name"] == "test_jump_scores"][0]
    targets = [("chr1", resolve(t["contigs"][0]["seq"])), ("chr2", resolve(t["contigs"][2]["seq"]))]
    to_ref = {0: 0, 2: 1, 1: 2, 3: 3}
    for case in t["cases"]:
        js, jo, ji = case["jump_scores"]
        kw = dict(mode="local", match_score=1, mismatch_score=-1, gap_open=-100000, gap_extend=-100000)
        al = stitch_amd.Builder(double_strand=True, jump_score_same_contig_and_strand=js, jump_score_same_contig_opposite_strand=jo,
                                jump_score_inter_contig=ji, keep_clipping=True, **kw).build_aligners([stitch_amd.TargetSeq(*x) for x in targets])
        o = orc.Aligners(targets, double_strand=True, jump_same=js, jump_opp=jo, jump_inter=ji, match=1, mismatch=-1,
                         gap_open=-100000, gap_extend=-100000, mode="local")
        got, _ = al.align_one(t["y"])
        want = o.align(t["y"])
        assert [c.score for c in got] == [c.score for c in want]
        assert summary(got[0], (4, 5)) == (want[0].xstart, want[0].xend, want[0].ystart, want[0].yend, want[0].score,
                                           want[0].start_contig_idx, want[0].cigar(), want[0].length)
        # ... and the reference's own expectation, in the reference's contig numbering
        g = got[0]
        r = stitch_amd.Alignment()
        for f in stitch_amd.Alignment.__slots__[:-1]:
            setattr(r, f, getattr(g, f))
        r.start_contig_idx, r.end_contig_idx = to_ref[g.start_contig_idx], to_ref[g.end_contig_idx]
        r.operations = [(k, to_ref[a], b) if k == 6 else (k, a, b) for k, a, b in g.operations]
        e = case["expect"]
        assert (r.xstart, r.xend, r.ystart, r.yend, r.score, r.start_contig_idx, r.cigar(), r.length) == \
            (e["xstart"], e["xend"], e["ystart"], e["yend"], e["score"], e["start_contig_idx"], e["cigar"], e["length"]), (case, r)


def oracle_key(a):
    return a.key()


def product_key(a):
    return a.key()


def run_pair(targets, reads, check_sam=True, **opts):
    """Aligns `reads` with the product and the oracle under identical options; asserts identical chains (+ SAM text)."""
    po = dict(opts)
    oo = dict(opts)
    # option names differ slightly between Builder and the oracle helper
    ren = {"match_score": "match", "mismatch_score": "mismatch", "default_jump_score": "jump_score",
           "jump_score_same_contig_and_strand": "jump_same", "jump_score_same_contig_opposite_strand": "jump_opp",
           "jump_score_inter_contig": "jump_inter"}
    oo = {ren.get(k, k): v for k, v in oo.items()}
    if "pick_primary" in oo:
        oo["pick_primary"] = {"query-length": 0, "score": 1}[oo["pick_primary"]]
    al = stitch_amd.Builder(**po).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets])
    o = orc.Aligners(targets, **oo)
    res = al.align(reads)
    for k, read in enumerate(reads):
        want = o.align(read)
        got = res[k][0]
        assert [product_key(c) for c in got] == [oracle_key(c) for c in want], \
            f"read {k}: want {want}\n got {got}\n opts {opts}\n targets {targets}\n read {read}"
        if check_sam and all(len(c.operations) for c in got):
            q = "I" * len(read)
            assert al.format_sam(k, f"read_{k} extra", read, q) == o.format_sam(f"read_{k} extra", read, q), (k, opts)
    return al


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def chimera(rng, targets, n, err=0.05, both=False):
    parts = []
    while sum(map(len, parts)) < n:
        if rng.random() < 0.15:
            parts.append(rand_seq(rng, rng.randint(3, 25)))
            continue
        s = rng.choice(targets)[1]
        a = rng.randrange(len(s)); b = min(len(s), a + rng.randint(10, max(11, n // 2)))
        p = s[a:b]
        if both and rng.random() < 0.5:
            p = rc(p)
        p = "".join(c if rng.random() > err else rng.choice(["", rng.choice("ACGT"), c + rng.choice("ACGT")]) for c in p)
        parts.append(p)
    return "".join(parts)[:n] or "A"


@pytest.mark.parametrize("seed", range(24))
def test_random_options_vs_oracle(seed):
    rng = random.Random(seed)
    T = rng.randint(1, 5)
    targets = [(f"t{k}", rand_seq(rng, rng.randint(20, 400))) for k in range(T)]
    mode = rng.choice(["local", "local", "local", "query-local", "target-local", "global"])
    double = rng.random() < 0.5
    opts = dict(mode=mode, double_strand=double, circular=rng.random() < 0.4, circular_slop=rng.choice([0, 5, 20]),
                suboptimal=rng.random() < 0.4, suboptimal_pct=rng.choice([20.0, 50.0, 1.0]),
                use_eq_and_x=rng.random() < 0.5, soft_clip=rng.random() < 0.5, pick_primary=rng.choice(["query-length", "score"]),
                filter_secondary=rng.random() < 0.3)
    if rng.random() < 0.5:
        opts.update(match_score=1, mismatch_score=rng.choice([-1, -4]), gap_open=rng.choice([-6, -3, 0]), gap_extend=rng.choice([-2, -1]),
                    default_jump_score=rng.choice([-10, -5, -1]))
    if rng.random() < 0.3:
        opts.update(jump_score_same_contig_and_strand=rng.choice([-10, -3]), jump_score_inter_contig=rng.choice([-12, -4]))
    reads = [chimera(rng, targets, rng.randint(5, 300), both=double) for _ in range(6)]
    reads.append(reads[-1])                   # identical consecutive reads are aligned once (align/io.rs:118-146)
    reads.append(reads[0].lower())            # case-insensitive (seq_upper_case)
    try:
        run_pair(targets, reads, **opts)
    except stitch_amd.StitchError as e:
        # global / query-local + several contigs can hit the reference's out-of-range XJUMP: the oracle
        # must agree that the reference is undefined there
        assert "shorter contig" in str(e)
        o = orc.Aligners(targets, **{})
        pytest.skip("reference-undefined XJUMP case")
    except RuntimeError as e:
        assert "out of range" in str(e)
        pytest.skip("reference-undefined XJUMP case (oracle)")


def test_cfg1_shape_150bp_vs_5kb_plasmid():
    """BASELINE config 1 shape: 150 bp reads vs one 5 kb contig, local, single strand (a subset of the 1k reads)."""
    db = synth.make_db(1, 5000, 1001)
    reads = synth.make_reads(db, 64, 150, 43, max_segments=1)
    targets = [(n, s.decode()) for n, s in db]
    run_pair(targets, [r.decode() for r in reads])


def test_multi_tile_contigs_and_long_reads():
    """contigs of several 256-row tiles, reads of a few hundred columns, both strands, chimeric"""
    db = synth.make_db(6, 1100, 5)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 12, 700, 77, both_strands=True)]
    run_pair(targets, reads, double_strand=True)
    run_pair(targets, reads[:4], double_strand=True, suboptimal=True)


@pytest.mark.parametrize("batch", [1, 3, 40])
def test_ragged_contig_lengths_and_wave_ranges(batch):
    """contigs of 1 row to several tiles in one database: tile ranges of waves and workgroups with empty, single-tile
    and split contigs, for one read alone (many workgroups per read), a few, and a launch full of reads"""
    rng = random.Random(11 + batch)
    lens = [1, 2, 3, 255, 256, 257, 511, 513, 40, 1300, 7, 64]
    targets = [(f"c{k}", rand_seq(rng, n)) for k, n in enumerate(lens)]
    reads = [chimera(rng, [t for t in targets if len(t[1]) > 30], rng.randint(30, 500), both=True) for _ in range(batch)]
    run_pair(targets, reads, double_strand=True, check_sam=False)
    run_pair(targets[:4], reads[:2], check_sam=False)                      # a workgroup with fewer tiles than waves
    run_pair(targets, reads[:3], circular=True, suboptimal=True, check_sam=False)


@pytest.mark.parametrize("seed", range(10))
def test_scoring_at_the_limits_of_the_16_bit_kernel(seed):
    """penalties down to -16000 and match scores up to 100 with reads as long as match * n <= 32767 allows (local16_ok)"""
    rng = random.Random(5000 + seed)
    targets = [(f"t{k}", rand_seq(rng, rng.choice([30, 300, 900]))) for k in range(rng.randint(1, 5))]
    match = rng.choice([1, 7, 100])
    opts = dict(match_score=match, mismatch_score=rng.choice([-1, -16000, -300]), gap_open=rng.choice([0, -8000, -1]),
                gap_extend=rng.choice([-8000, -1, -16000]), default_jump_score=rng.choice([0, -16000, -5]),
                double_strand=rng.random() < 0.5, circular=rng.random() < 0.3)
    if opts["gap_open"] + opts["gap_extend"] < -16000:
        opts["gap_open"] = 0
    maxn = max(5, min(600, 32767 // match))
    reads = [chimera(rng, targets, rng.randint(5, maxn), both=opts["double_strand"]) for _ in range(4)]
    run_pair(targets, reads, check_sam=False, **opts)


def test_circular_realignment():
    """reads that wrap around the origin of circular contigs (realign_origin, aligners/mod.rs:442-553)"""
    rng = random.Random(3)
    targets = [(f"p{k}", rand_seq(rng, 300 + 50 * k)) for k in range(3)]
    reads = []
    for k in range(8):
        s = targets[k % 3][1]
        cut = rng.randrange(20, len(s) - 20)
        w = s[cut:] + s[:cut]                      # whole plasmid, rotated
        a = rng.randrange(0, 40); b = rng.randrange(len(w) - 40, len(w))
        reads.append(w[a:b])
    reads.append(targets[0][1][250:] + targets[0][1][:60] + targets[1][1][100:200])
    run_pair(targets, reads, circular=True)
    run_pair(targets, reads, circular=True, suboptimal=True, double_strand=True)


def test_cfg5_shape_many_circular_contigs_suboptimal():
    """BASELINE config 5 in miniature: 200 circular contigs (more contigs than lanes in a wavefront, many workgroups per
    read), PacBio-like reads that wrap around origins, --circular --suboptimal; chains and SAM text against the oracle"""
    db = synth.make_db(200, 150, 1002)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 5, 600, 47, sub=0.01, ins=0.005, dele=0.005, circular=True)]
    run_pair(targets, reads, circular=True, suboptimal=True)
    run_pair(targets[:120], reads[:2], circular=True, suboptimal=True, double_strand=True, check_sam=False)   # 240 contig-strands (255 is the reference's limit)


def test_iupac_codes_n_and_lower_case():
    """bases are compared as bytes after upper-casing (N matches N, R matches R: `MatchParams`, util/dna.rs:5-23), the
    reverse strand complements IUPAC codes (dna.rs:31-41); SAM text included (SEQ of reverse-strand records)"""
    rng = random.Random(77)
    targets = [(f"t{k}", rand_seq(rng, n, "ACGTNRYKMSWBDHV")) for k, n in enumerate([300, 520, 45])]
    targets[1] = (targets[1][0], targets[1][1].lower())
    reads = [chimera(rng, targets, rng.randint(40, 400), both=True) for _ in range(6)]
    reads += [rc(targets[0][1][20:250]), "N" * 30 + targets[2][1] + "n" * 12, targets[1][1][100:400].upper()]
    run_pair(targets, reads, double_strand=True)
    run_pair(targets, reads[:4], double_strand=True, suboptimal=True, soft_clip=True, use_eq_and_x=True)


def test_error_behaviour_matches_the_reference():
    """what the reference refuses, the product refuses (an error code and a message, never a silent fallback): contig
    indexes above 255 (packed_length_cell.rs:112-114, 139), and the oracle agrees; 256 contig-strands still work"""
    rng = random.Random(9)
    many = [(f"c{k}", rand_seq(rng, 20)) for k in range(129)]
    with pytest.raises(stitch_amd.StitchError, match="256 contig-strands"):
        stitch_amd.Builder(double_strand=True).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in many]).align(["ACGTACGTAC"])
    with pytest.raises(RuntimeError):
        orc.Aligners(many, double_strand=True).align("ACGTACGTAC")
    run_pair(many[:128], ["ACGTACGTAC" * 3, many[127][1] + rc(many[3][1])], double_strand=True)      # indexes 0..255


def test_batch_split_invariance_and_determinism(monkeypatch):
    """results do not depend on how reads are packed into launches; two runs are identical"""
    db = synth.make_db(4, 600, 9)
    targets = [stitch_amd.TargetSeq(n, s) for n, s in db]
    reads = synth.make_reads(db, 40, 400, 11)
    al = stitch_amd.Builder().build_aligners(targets)
    a = [[c.key() for c in ch] for ch, _ in al.align(reads)]
    b = [[c.key() for c in ch] for ch, _ in al.align(reads)]
    assert a == b
    monkeypatch.setenv("STITCH_ARENA_BYTES", str(24 << 20))      # forces many small launches
    al2 = stitch_amd.Builder().build_aligners(targets)
    c = [[c.key() for c in ch] for ch, _ in al2.align(reads)]
    assert a == c
    assert al2.timing()["launches"] > 1


def test_more_tiles_than_one_workgroup_slot_table_holds():
    """A database of 128 x 10 kb contigs on both strands = 256 aligners x 40 tiles = 10 240 tiles of the streaming Local
    kernel (contigs beyond 5120 rows: the register kernel does not take them).  ceil(tiles / 2048) = 5 workgroups per read
    would leave one workgroup with 52 contigs = 2080 tiles, more than its slot table holds (2048): the host has to give the
    read more workgroups (or fewer reads per launch), never cut tiles off.  Short reads keep the oracle's matrix small."""
    rng = random.Random(77)
    targets = [(f"big{k}", rand_seq(rng, 10000)) for k in range(128)]
    reads = [chimera(rng, targets[:40], 60, err=0.03, both=True) for _ in range(3)]
    al = run_pair(targets, reads, double_strand=True, check_sam=False)
    tm = al.timing()
    assert tm["fill_kind"] == 1 and tm["wg_per_read"] >= 6 and tm["fallbacks"] == 0, tm


def test_partner_timeout_on_a_read_beyond_one_slot_table_falls_back_to_the_generic_kernel(monkeypatch):
    """The same database with the first attempt declared timed out (test hook): one workgroup per read cannot hold 10 240 tiles in
    its slot table, so the repeat runs the generic kernel — with a wave count worked out for THAT kernel (the streaming kernel's
    12 waves exceed its launch bounds: the relaunch used to be rejected)."""
    monkeypatch.setenv("STITCH_TEST_FAIL_FIRST_ATTEMPT", "1")
    rng = random.Random(78)
    targets = [(f"big{k}", rand_seq(rng, 10000)) for k in range(128)]
    reads = [chimera(rng, targets[:40], 50, err=0.03, both=True) for _ in range(2)]
    al = run_pair(targets, reads, double_strand=True, check_sam=False)
    tm = al.timing()
    assert tm["fill_kind"] == 0 and tm["fallbacks"] >= 1 and tm["wg_per_read"] == 1, tm



def test_a_chain_that_starts_elsewhere_in_the_reference_chains_contig(monkeypatch):
    """--suboptimal, joined walks (walk_core.h VisitRec): a secondary chain that starts in the contig the reference chain starts in, at
    another cell, arrives at (contig, row 0, column 0) in the reference walk's state once its own prefix clips are done — it must keep its
    own start coordinates (found by tests/gpu_fuzz.py FUZZ_STREAM=1, seed 7087, read 11: circular, both strands, targets of 34, 1453, 5
    and 29 bases)"""
    monkeypatch.setenv("FUZZ_STREAM", "1")
    from tests import gpu_fuzz as G
    targets, reads, opts, _, _, lens = G.draw(7087)
    assert lens == [34, 1453, 5, 29] and len(reads[11]) == 124 and opts["suboptimal"]
    run_pair(targets, [reads[11], reads[3], reads[11]], **opts)
