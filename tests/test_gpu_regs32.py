"""The 32-bit register-resident kernel (stitch_amd/csrc/fill_regs32.hip) against the golden vectors and the oracle: every
clipping mode (query-local, target-local, global, and local when forced or when the read is beyond the 16-bit kernels' range).

By default only reads with at least 2048 active contig rows go to a register-resident kernel, so the parity suite's small cases
would never reach it: here STITCH_REGS_MIN_ROWS=0 sends EVERY eligible read to it and STITCH_FORCE_REGS32=1 makes it take the
Local-mode reads too (which otherwise run the 16-bit kernels).  The scenarios of tests/test_gpu_parity.py and
tests/test_gpu_regs.py are replayed, and the modes get scenarios of their own (ragged contig lengths, more than 64 contigs,
contigs of 5120 rows, circular contigs, the forced fallback, a 40 000-base Local read)."""
import random

import pytest

import stitch_amd
from oracle import oracle as orc
from stitch_amd import synth
from tests import test_gpu_parity as P

pytestmark = pytest.mark.gpu
MODES = ["query-local", "target-local", "global"]


@pytest.fixture(autouse=True)
def every_eligible_read_to_the_32_bit_register_kernel(monkeypatch):
    monkeypatch.setenv("STITCH_REGS_MIN_ROWS", "0")
    monkeypatch.setenv("STITCH_FORCE_REGS32", "1")


def run_mode(targets, reads, mode, **kw):
    """run_pair, tolerating the one reference-undefined case (an end-of-read jump into a shorter contig: DESIGN.md 2)"""
    try:
        return P.run_pair(targets, reads, mode=mode, **kw)
    except stitch_amd.StitchError as e:
        assert "shorter contig" in str(e)
    except RuntimeError as e:
        assert "out of range" in str(e)
    return None


def test_the_32_bit_register_kernel_is_the_one_that_runs(monkeypatch):
    db = synth.make_db(5, 700, 3)
    t = [stitch_amd.TargetSeq(n, s) for n, s in db]
    reads = synth.make_reads(db, 4, 300, 5)
    for mode in ["local"] + MODES:
        al = stitch_amd.Builder(mode=mode).build_aligners(t)
        al.align(reads)
        tm = al.timing()
        assert tm["fill_kind"] == 3 and tm["wg_per_read"] == 2 and tm["fallbacks"] == 0, (mode, tm)      # 5 contigs: two workgroups of four waves
    # without the test knobs: the other modes still get it once a read has 2048 rows, Local mode keeps its 16-bit kernels
    monkeypatch.delenv("STITCH_FORCE_REGS32"); monkeypatch.delenv("STITCH_REGS_MIN_ROWS")
    al = stitch_amd.Builder(mode="global").build_aligners(t)
    al.align(reads)
    assert al.timing()["fill_kind"] == 3
    al = stitch_amd.Builder(mode="local").build_aligners(t)
    al.align(reads)
    assert al.timing()["fill_kind"] == 2
    small = stitch_amd.Builder(mode="global").build_aligners(t[:2])
    small.align(reads)
    assert small.timing()["fill_kind"] == 0                              # 1400 rows: below the threshold, the generic kernel


def test_golden_single_contig():          # all 63 vectors of single_contig_aligner.rs:915-1773, in their own modes
    P.test_golden_single_contig()


def test_golden_multi_contig():
    P.test_golden_multi_contig()


def test_golden_jump_score_priorities():
    P.test_golden_jump_score_priorities()


@pytest.mark.parametrize("seed", range(24))
def test_random_options_vs_oracle(seed):
    P.test_random_options_vs_oracle(seed)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("seed", range(6))
def test_random_options_in_every_mode(mode, seed):
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
    reads.append(reads[-1]); reads.append(reads[0].lower())
    run_mode(targets, reads, mode, **opts)


def test_cfg1_shape_150bp_vs_5kb_plasmid():
    P.test_cfg1_shape_150bp_vs_5kb_plasmid()
    db = synth.make_db(1, 5000, 1001)
    reads = [r.decode() for r in synth.make_reads(db, 12, 150, 43, max_segments=1)]
    for mode in MODES:                                                    # one contig: no end-of-read jump into another one
        P.run_pair([(n, s.decode()) for n, s in db], reads, mode=mode)


def test_multi_tile_contigs_and_long_reads():
    P.test_multi_tile_contigs_and_long_reads()


@pytest.mark.parametrize("mode", ["local"] + MODES)
@pytest.mark.parametrize("batch", [1, 3, 40])
def test_ragged_contig_lengths(mode, batch):
    """contig lengths around the lane mapping's edges (1 row, 3, 4, 5, 255..257, 1300 rows: groups of four rows dealt to 64 lanes)"""
    rng = random.Random(11 + batch)
    lens = [1, 2, 3, 4, 5, 255, 256, 257, 511, 513, 40, 1300, 7, 64, 65, 63]
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate(lens)]
    reads = [P.chimera(rng, [t for t in targets if len(t[1]) > 30], rng.randint(30, 500), both=True) for _ in range(batch)]
    run_mode(targets, reads, mode, double_strand=True, check_sam=False)
    run_mode(targets[:4], reads[:2], mode, check_sam=False)
    run_mode(targets, reads[:3], mode, circular=True, suboptimal=True, check_sam=False)


@pytest.mark.parametrize("mode", ["local"] + MODES)
def test_more_than_64_active_contigs(mode):
    """four granule registers per lane (NQ = 4): 150 contigs, both strands"""
    rng = random.Random(21)
    targets = [(f"c{k}", P.rand_seq(rng, rng.choice([150, 151, 90]))) for k in range(100)]
    reads = [P.chimera(rng, targets[:30], rng.randint(100, 500), both=True) for _ in range(3)]
    run_mode(targets, reads, mode, double_strand=True, check_sam=False)
    run_mode(targets, reads[:2], mode, circular=True, suboptimal=True, check_sam=False)


@pytest.mark.parametrize("mode", ["local"] + MODES)
def test_long_contigs_80_rows_per_lane(mode):
    """contigs at the kernel's limit of 5120 rows (80 rows per lane) and just below, short reads to keep the oracle's matrix small"""
    rng = random.Random(31)
    targets = [("a", P.rand_seq(rng, 5120)), ("b", P.rand_seq(rng, 5119)), ("c", P.rand_seq(rng, 4097)), ("d", P.rand_seq(rng, 2561)), ("e", P.rand_seq(rng, 300))]
    reads = [P.chimera(rng, targets, rng.randint(60, 220), err=0.04) for _ in range(4)]
    reads.append(targets[0][1][5000:] + targets[1][1][:80])               # the last rows of one contig, the first of another
    run_mode(targets, reads, mode, check_sam=False)
    run_mode(targets[:3], reads[:2], mode, circular=True, check_sam=False)


def test_circular_realignment():
    P.test_circular_realignment()
    rng = random.Random(3)
    targets = [(f"p{k}", P.rand_seq(rng, 300 + 50 * k)) for k in range(3)]
    reads = []
    for k in range(6):
        s = targets[k % 3][1]
        cut = rng.randrange(20, len(s) - 20)
        w = s[cut:] + s[:cut]
        reads.append(w[rng.randrange(0, 40):rng.randrange(len(w) - 40, len(w))])
    for mode in MODES:
        run_mode(targets, reads, mode, circular=True)
        run_mode(targets[:1], reads[:3], mode, circular=True, suboptimal=True, double_strand=True)


def test_iupac_codes_n_and_lower_case():
    P.test_iupac_codes_n_and_lower_case()


def test_a_local_read_beyond_the_16_bit_kernels(monkeypatch):
    """40 000 bases with match = 1: match * n > 32767, so the 16-bit Local-mode kernels refuse it; the 32-bit register kernel takes it
    WITHOUT being forced.  Two short contigs keep the oracle's matrix at 24 M cells."""
    monkeypatch.delenv("STITCH_FORCE_REGS32")
    rng = random.Random(41)
    targets = [("u", P.rand_seq(rng, 300)), ("v", P.rand_seq(rng, 280))]
    unit = targets[0][1][20:260] + P.rand_seq(rng, 15) + targets[1][1][10:250]
    read = ""
    while len(read) < 40000:
        read += "".join(c if rng.random() > 0.03 else rng.choice("ACGT") for c in unit)
    read = read[:40000]
    al = P.run_pair(targets, [read], check_sam=True)
    tm = al.timing()
    assert tm["fill_kind"] == 3 and tm["fallbacks"] == 0, tm
    # a read that long with a perfect repeat structure scores far above 32767 / match only if the kernel really is 32-bit
    assert al.align([read])[0][0][0].score > 20000


def test_the_forced_fallback_relaunches_on_the_generic_kernel(monkeypatch):
    monkeypatch.setenv("STITCH_TEST_FAIL_FIRST_ATTEMPT", "1")
    db = synth.make_db(6, 900, 4)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 6, 400, 6, both_strands=True)]
    al = run_mode(targets[:1], reads, "global")
    if al is not None:
        tm = al.timing()
        assert tm["fallbacks"] >= 1 and tm["fill_kind"] == 0 and tm["wg_per_read"] == 1, tm


def test_batch_split_invariance(monkeypatch):
    db = synth.make_db(4, 600, 9)
    targets = [stitch_amd.TargetSeq(n, s) for n, s in db]
    reads = synth.make_reads(db, 30, 400, 11)
    for mode in ("target-local", "local"):
        al = stitch_amd.Builder(mode=mode).build_aligners(targets)
        a = [[c.key() for c in ch] for ch, _ in al.align(reads)]
        monkeypatch.setenv("STITCH_ARENA_BYTES", str(24 << 20))
        al2 = stitch_amd.Builder(mode=mode).build_aligners(targets)
        b = [[c.key() for c in ch] for ch, _ in al2.align(reads)]
        monkeypatch.delenv("STITCH_ARENA_BYTES")
        assert a == b and al2.timing()["launches"] > 1 and al2.timing()["fill_kind"] == 3
