"""Persistent teams of the register-resident fill (stitch_api.cpp run_jobs_streaming, fill_regs.hip): one launch per call whose teams
pull the next read off a queue when theirs ends, the host walking finished reads and recycling their arena blocks meanwhile.

At production sizes the path needs more reads than the chip holds teams (41+ reads of 50 contigs).  Here the teams and the arena
blocks of a run are capped (STITCH_STREAM_TEAMS, STITCH_STREAM_BLOCKS) so that small batches queue up behind two or three teams and
every block is recycled several times; every eligible read goes to the register kernel (STITCH_REGS_MIN_ROWS=0).  Results are compared
with the oracle chain by chain (tests/test_gpu_parity.py run_pair) and with the launch-by-launch path (STITCH_NO_STREAM)."""
import random

import pytest

import stitch_amd
from stitch_amd import synth
from tests import test_gpu_parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def few_teams_few_blocks(monkeypatch):
    monkeypatch.setenv("STITCH_REGS_MIN_ROWS", "0")
    monkeypatch.setenv("STITCH_STREAM_TEAMS", "2")
    monkeypatch.setenv("STITCH_STREAM_BLOCKS", "3")


def _aligners(db, **opts):
    return stitch_amd.Builder(**opts).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])


def test_the_persistent_teams_are_what_runs(monkeypatch):
    db = synth.make_db(5, 700, 3)
    reads = synth.make_reads(db, 12, 300, 5)
    al = _aligners(db)
    got = al.align(reads)
    tm = al.timing()
    # (a context's first jobs go as one classic launch — what the runtime sets up on first use must not happen beside resident teams —
    # and the rest of the call through the queue)
    assert tm["fill_kind"] == 2 and tm["stream_runs"] == 1 and tm["launches"] == 2 and tm["fallbacks"] == 0, tm
    monkeypatch.setenv("STITCH_NO_STREAM", "1")
    al2 = _aligners(db)
    want = al2.align(reads)
    tm2 = al2.timing()
    assert tm2["stream_runs"] == 0 and tm2["fill_kind"] == 2
    assert [[c.key() for c in r[0]] for r in got] == [[c.key() for c in r[0]] for r in want]
    assert tm["cells"] == tm2["cells"]


@pytest.mark.parametrize("teams,blocks", [(1, 2), (2, 2), (2, 3), (3, 7), (4, 40)])
def test_queue_shapes_against_the_oracle(monkeypatch, teams, blocks):
    """one team (strictly serial), as many blocks as teams (a team waits for the host's walk before every read), spare blocks"""
    monkeypatch.setenv("STITCH_STREAM_TEAMS", str(teams))
    monkeypatch.setenv("STITCH_STREAM_BLOCKS", str(blocks))
    rng = random.Random(100 * teams + blocks)
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate([900, 650, 1300, 257, 40])]
    reads = [P.chimera(rng, [t for t in targets if len(t[1]) > 100], rng.randint(60, 500), both=False) for _ in range(17)]
    P.run_pair(targets, reads)


def test_reads_of_very_different_lengths_and_duplicates():
    """a team's reads differ 20-fold in length (jobs are handed out longest first), consecutive duplicates share a job"""
    rng = random.Random(5)
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate([700, 800, 600])]
    reads = []
    for k in range(14):
        r = P.chimera(rng, targets, rng.choice([40, 90, 400, 1100]), both=False)
        reads.append(r)
        if k % 4 == 1:
            reads.append(r)
    P.run_pair(targets, reads)


@pytest.mark.parametrize("seed", range(8))
def test_random_options_vs_oracle(seed):
    P.test_random_options_vs_oracle(seed)


def test_suboptimal_and_circular_through_the_queue():
    """--suboptimal (one walk per contig: walk_all_kernel on the finished jobs of a range) and --circular (the origin re-alignments are
    a second run of jobs on contig subsets of different sizes: those go launch by launch)"""
    db = synth.make_db(6, 500, 1002)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 9, 350, 47, sub=0.01, ins=0.005, dele=0.005, circular=True)]
    P.run_pair(targets, reads, suboptimal=True, circular=True)
    P.run_pair(targets, reads, circular=True, check_sam=False)


def test_more_than_64_contigs_and_double_strand():
    db = synth.make_db(70, 150, 1002)
    targets = [(n, s.decode()) for n, s in db]
    reads = [r.decode() for r in synth.make_reads(db, 7, 300, 47)]
    P.run_pair(targets, reads, suboptimal=True, check_sam=False)
    P.run_pair(targets[:9], reads, double_strand=True, check_sam=False)


def test_a_chain_beyond_its_buffer_ends_the_run_and_the_classic_path_finishes(monkeypatch):
    """free gaps make a chain longer than the default operations buffer: the exact-size re-walk allocates device memory, which would
    wait for the running teams — the run is called off and the rest goes launch by launch (stitch_timing.fallbacks says so)"""
    rng = random.Random(9)
    targets = [(f"c{k}", P.rand_seq(rng, 300)) for k in range(3)]
    reads = [P.chimera(rng, targets, 200, both=False) for _ in range(6)]
    P.run_pair(targets, reads, gap_open=0, gap_extend=-1, check_sam=False)


def test_a_slow_launch_beside_the_teams_retires_one_team(monkeypatch):
    """a launch beside the teams that waits beyond its (short) bound: ONE team is asked to leave — the first to reach the end of its read
    takes the ticket and frees its wave slots — and the others finish the queue; nothing is repeated (test hook: the first wait of the run
    counts as slow)"""
    monkeypatch.setenv("STITCH_TEST_STREAM_RETIRE", "1")
    monkeypatch.setenv("STITCH_STREAM_TEAMS", "3")
    rng = random.Random(78)
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate([800, 500, 900])]
    reads = [P.chimera(rng, targets, rng.randint(80, 400), both=False) for _ in range(24)]
    al = P.run_pair(targets, reads)
    tm = al.timing()
    assert tm["stream_runs"] == 1 and tm["fallbacks"] == 0 and tm["teams_retired"] == 1, tm
    # one team only: nobody may be asked to leave (the queue would never empty), the bound then calls the run off as before
    monkeypatch.setenv("STITCH_STREAM_TEAMS", "1")
    al = P.run_pair(targets, reads[:9])
    tm = al.timing()
    assert tm["teams_retired"] == 0 and tm["fallbacks"] == 0, tm


def test_a_stalled_launch_beside_the_teams_calls_the_run_off(monkeypatch):
    """a fix-up / walk launch (or a copy) beside resident teams that does not end within its bound: the host calls the run off, the teams
    leave after the read they are on, and what is left goes launch by launch (test hook: the first wait counts as a stall)"""
    monkeypatch.setenv("STITCH_TEST_STREAM_STALL", "1")
    rng = random.Random(77)
    targets = [(f"c{k}", P.rand_seq(rng, n)) for k, n in enumerate([800, 500, 900])]
    reads = [P.chimera(rng, targets, rng.randint(80, 400), both=False) for _ in range(16)]
    al = P.run_pair(targets, reads)
    tm = al.timing()
    assert tm["stream_runs"] == 1 and tm["fallbacks"] == 1, tm
