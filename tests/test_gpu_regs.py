"""The register-resident Local-mode kernel (stitch_amd/csrc/fill_regs.hip) against the golden vectors and the oracle.

By default only reads with at least 2048 active contig rows go to it (smaller ones run fill_local16.hip), so the parity suite's
small cases would never reach it: here STITCH_REGS_MIN_ROWS=0 sends EVERY eligible read to it (Local mode, contigs of at most
5120 rows) and the scenarios of tests/test_gpu_parity.py are replayed.  Contig lengths around the lane
mapping's edges (1 row, 3, 4, 5, 255..257, 1300 rows: groups of four rows dealt to 64 lanes) get a test of their own."""
import random

import pytest

import stitch_amd
from stitch_amd import synth
from tests import test_gpu_parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def every_eligible_read_to_the_register_kernel(monkeypatch):
    monkeypatch.setenv("STITCH_REGS_MIN_ROWS", "0")


def test_the_register_kernel_is_the_one_that_runs(monkeypatch):
    """1400 contig rows: BELOW the default threshold of 2048, so fill_kind == 2 here proves that the fixture above is what sends
    this module's small cases to the register kernel — and the same database without it runs the streaming kernel."""
    db = synth.make_db(2, 700, 3)
    al = stitch_amd.Builder().build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
    al.align(synth.make_reads(db, 4, 300, 5))
    tm = al.timing()
    assert tm["fill_kind"] == 2 and tm["wg_per_read"] == 1          # 2 contigs: one workgroup of four waves
    db5 = synth.make_db(5, 700, 3)
    al5 = stitch_amd.Builder().build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db5])
    al5.align(synth.make_reads(db5, 4, 300, 5))
    tm = al5.timing()
    assert tm["fill_kind"] == 2 and tm["wg_per_read"] == 2          # 5 contigs: two workgroups of four waves
    monkeypatch.delenv("STITCH_REGS_MIN_ROWS")
    al_default = stitch_amd.Builder().build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
    al_default.align(synth.make_reads(db, 4, 300, 5))
    assert al_default.timing()["fill_kind"] == 1                    # default threshold: small reads stream


def test_golden_single_contig():
    P.test_golden_single_contig()


def test_golden_multi_contig():
    P.test_golden_multi_contig()


def test_golden_jump_score_priorities():
    P.test_golden_jump_score_priorities()


@pytest.mark.parametrize("seed", range(24))
def test_random_options_vs_oracle(seed):
    P.test_random_options_vs_oracle(seed)


def test_cfg1_shape_150bp_vs_5kb_plasmid():
    P.test_cfg1_shape_150bp_vs_5kb_plasmid()


def test_multi_tile_contigs_and_long_reads():
    P.test_multi_tile_contigs_and_long_reads()


@pytest.mark.parametrize("batch", [1, 3, 40])
def test_ragged_contig_lengths(batch):
    rng = random.Random(11 + batch)
    lens = [1, 2, 3, 4, 5, 255, 256, 257, 511, 513, 40, 1300, 7, 64, 65, 63]
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate(lens)]
    reads = [P.chimera(rng, [t for t in targets if len(t[1]) > 30], rng.randint(30, 500), both=True) for _ in range(batch)]
    P.run_pair(targets, reads, double_strand=True, check_sam=False)
    P.run_pair(targets[:4], reads[:2], check_sam=False)
    P.run_pair(targets, reads[:3], suboptimal=True, check_sam=False)


@pytest.mark.parametrize("seed", range(10))
def test_scoring_at_the_limits_of_the_16_bit_kernel(seed):
    P.test_scoring_at_the_limits_of_the_16_bit_kernel(seed)


def test_more_than_64_active_contigs():
    """granule records in more than one register per lane (fill_regs_kernel<4>), many workgroups per read"""
    db = synth.make_db(120, 150, 1002)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 5, 400, 47, sub=0.01, ins=0.005, dele=0.005)]
    P.run_pair(targets, reads, suboptimal=True)
    P.run_pair(targets, reads[:2], double_strand=True, check_sam=False)


def test_iupac_codes_n_and_lower_case():
    P.test_iupac_codes_n_and_lower_case()


def test_circular_realignment():
    """row 1 of a circular contig may continue from row m of the previous column at no cost; origin re-alignment on top"""
    P.test_circular_realignment()


def test_cfg5_shape_many_circular_contigs_suboptimal():
    P.test_cfg5_shape_many_circular_contigs_suboptimal()


def test_long_contigs_near_the_register_capacity():
    """contigs of 4990..5120 rows (79-80 rows per lane), both strands, chimeric reads of a few hundred columns"""
    rng = random.Random(21)
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate([5120, 5000, 4991, 5119, 3000])]
    reads = [P.chimera(rng, targets, rng.randint(150, 400), both=True) for _ in range(6)]
    P.run_pair(targets, reads, double_strand=True)
    P.run_pair(targets, reads[:3], suboptimal=True, check_sam=False)


def test_partner_timeout_falls_back_to_one_workgroup_per_read(monkeypatch):
    """a launch whose workgroups cannot all be resident ends with a bounded wait; the host repeats it once with one workgroup
    per read instead of failing the batch (test hook: the first attempt of every cooperative launch counts as timed out)"""
    monkeypatch.setenv("STITCH_TEST_FAIL_FIRST_ATTEMPT", "1")
    db = synth.make_db(6, 900, 4)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 8, 400, 6, both_strands=True)]
    al = P.run_pair(targets, reads, double_strand=True)
    tm = al.timing()
    assert tm["fallbacks"] >= 1 and tm["fill_kind"] == 1 and tm["wg_per_read"] == 1


def test_a_repeated_launch_beside_the_other_windows_fill(monkeypatch):
    """two fills in flight (two arena windows), more than two launches, and the first attempt of every launch counts as timed out: each
    repeat (one workgroup per read) starts while the other window's fill is running, and every read still equals the oracle"""
    monkeypatch.setenv("STITCH_TEST_FAIL_FIRST_ATTEMPT", "pipelined")
    monkeypatch.setenv("STITCH_REGS_MIN_ROWS", "0")
    db = synth.make_db(4, 2000, 14)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 12, 1500, 16)]
    monkeypatch.setenv("STITCH_ARENA_BYTES", str(64 << 20))      # (a read's traceback is 12 MB: two or three reads per window)
    al = P.run_pair(targets, reads)
    tm = al.timing()
    assert tm["launches"] > 2 and tm["fallbacks"] >= 2, tm


def test_one_bit_y_suffix_records_equal_the_eight_byte_ones(monkeypatch):
    """--suboptimal / --circular launches run the YB instances of fill_regs_kernel (a cell that holds the column's common word sets bit 7 of
    its traceback byte instead of storing an 8-byte y-suffix record; the scan behind the column loop writes each row's last one): the same
    chains as with every record stored (STITCH_NO_YBITS), and as the oracle — reads on their contig, chimeras, and reads that match nothing
    (every cell the jump candidate: all bits, no records)"""
    monkeypatch.setenv("STITCH_REGS_MIN_ROWS", "0")
    rng = random.Random(4242)
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate([900, 260, 1400, 333, 70])]
    reads = [P.chimera(rng, targets[:4], rng.randint(80, 700), both=False) for _ in range(10)] + [P.rand_seq(rng, 300), targets[2][1][100:900]]
    for opts in (dict(suboptimal=True), dict(suboptimal=True, circular=True), dict(circular=True)):
        al = P.run_pair(targets, reads, **opts)
        assert al.timing()["fill_kind"] == 2
        got = [[c.key() for c in r[0]] for r in al.align(reads)]
        monkeypatch.setenv("STITCH_NO_YBITS", "1")
        al2 = stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in targets])
        want = [[c.key() for c in r[0]] for r in al2.align(reads)]
        monkeypatch.delenv("STITCH_NO_YBITS")
        assert got == want, opts
