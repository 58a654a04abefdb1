"""Parity at the size the headline metric is quoted on: BASELINE configs[1], 10 000 bp reads against 50 x 5 000 bp contigs.

One read is 2.5e9 DP cells here: traceback offsets pass 2^31, job blocks are GiB-aligned and a launch holds as many reads
as the CUs allow, several workgroups each — the path bench.py times.  The oracle needs 40 GB of 16-byte cells and about
two minutes per read on one core, so a few reads of the launch are compared in full (chains, operation lists, SAM text:
Aligners::align, aligners/mod.rs:237-340; traceback/mod.rs:219-373) and the others through the size-independent properties
of tests/test_gpu_parity.py::test_full_size_properties, now at n = 10 000."""
import os
from concurrent.futures import ThreadPoolExecutor

import pytest

import stitch_amd
from oracle import oracle as orc
from stitch_amd import synth

pytestmark = pytest.mark.gpu

N, CONTIGS, M = 10000, 50, 5000
ORACLE_BYTES_PER_READ = CONTIGS * (M + 1) * (N + 1) * 16          # traceback/mod.rs:122-126


def mem_available():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) * 1024
    return 0


def recompute_score(ops):            # A=1 B=-4 O=-6 E=-2 J=-10 (CLI defaults)
    sc, run = 0, None
    for k, _, _ in ops:
        if k == 0: sc += 1
        elif k == 1: sc += -4
        elif k in (2, 3): sc += -2 + (-6 if run != k else 0)
        elif k == 6: sc += -10
        run = k
    return sc


@pytest.fixture(scope="module")
def launch():
    """ONE stitch_align_batch call with 64 reads of 10 kb: 60 reads of bench.py's stream (seed 44: chimeras with 3/2/2 %
    errors, 10 % random reads, a duplicated neighbour) and four constructed reads with known answers."""
    db = synth.make_db(CONTIGS, M, 1001)
    targets = [stitch_amd.TargetSeq(n, s) for n, s in db]
    reads = synth.make_reads(db, 60, N, 44)
    s7, s12, s31 = db[7][1], db[12][1], db[31][1]
    two = s7 + s31                                                   # two whole contigs back to back
    four = s7[1000:3500] + s31[2500:5000] + s12[0:2500] + s7[2000:4500]
    noise = synth.make_reads(db, 2, N, 99, random_frac=1.0)
    reads = reads + [two, four] + noise
    al = stitch_amd.Builder().build_aligners(targets)
    res = al.align(reads)
    tm = al.timing()
    return db, reads, al, res, tm


def test_cfg2_launch_shape(launch):
    db, reads, al, res, tm = launch
    assert len(reads) == 64 and all(len(r) == N for r in reads)
    distinct = 1 + sum(1 for k in range(1, len(reads)) if reads[k] != reads[k - 1])
    assert tm["jobs"] == distinct
    assert tm["launches"] <= 2                       # the 63 distinct reads share a launch (two if the arena had to be cut)
    assert tm["wg_per_read"] >= 2                    # several workgroups per read: the exchange path is the one under test
    assert al.cells_filled == distinct * N * CONTIGS * M
    assert tm["fill_kind"] in (1, 2)                 # a Local-mode 16-bit kernel, not the generic fallback


def test_cfg2_constructed_reads_known_answers(launch):
    db, reads, al, res, tm = launch
    a = res[60][0][0]                                # contig 7 whole, then contig 31 whole
    assert (a.score, a.start_contig_idx, a.end_contig_idx, a.xstart, a.xend, a.ystart, a.yend) == (N - 10, 7, 31, 0, M, 0, N)
    assert a.cigar() == "5000=24C5000j5000="
    b = res[61][0][0]                                # four segments: three jumps, all bases matched
    assert b.score == N - 30 and b.start_contig_idx == 7 and b.end_contig_idx == 7
    assert [o for o in b.operations if o[0] == 6] == [(6, 31, 2500), (6, 12, 0), (6, 7, 2000)]
    assert b.cigar() == "2500=24C1000j2500=19c5000j2500=5c500j2500="
    for ch, _ in res[62:]:                           # unrelated reads: a valid local alignment whose score its op list reproduces
        c = ch[0]
        assert recompute_score(c.operations) == c.score and c.score > 0
    for ch, _ in res[:60]:
        assert recompute_score(ch[0].operations) == ch[0].score


MODES = ["global", "query-local", "target-local"]


@pytest.fixture(scope="module")
def oracle_wave(launch):
    """The oracle's answers for everything this module compares at full size, computed SIDE BY SIDE (a read is 40 GB of 16-byte cells and
    about two minutes on one core; ctypes releases the GIL: one aligner set per thread): two chimeric reads of the Local-mode launch, and
    one read per non-Local mode.  Five at once are 187 GiB of cells — the box's share of host memory allows that, not seven."""
    db, reads, al, res, tm = launch
    avail = mem_available()
    workers = int(min(5, (avail * 0.8) // (ORACLE_BYTES_PER_READ + (2 << 30))))
    if workers < 1:
        pytest.skip(f"the oracle needs {ORACLE_BYTES_PER_READ / 2**30:.0f} GiB per read, {avail / 2**30:.0f} GiB available")
    targets = [(n, s.decode()) for n, s in db]
    # first reads of the stream that are chimeric (more than one chain segment)
    picks = [k for k in range(60) if any(o[0] == 6 for o in res[k][0][0].operations)][:2]
    if len(picks) < 2:
        picks = [0, 1]
    mode_reads = [r for r in reads[:8] if r != reads[0]][:len(MODES)]
    jobs = [("local", k, reads[k], f"read_{k:07d}") for k in picks] + [(m, None, r, "read_0000000") for m, r in zip(MODES, mode_reads)]

    def oracle_read(job):
        mode, k, read, name = job
        o = orc.Aligners(targets) if mode == "local" else orc.Aligners(targets, mode=mode)
        want = o.align(read)
        return mode, k, [c.key() for c in want], o.format_sam(name, read.decode(), "I" * N)

    with ThreadPoolExecutor(max_workers=workers) as ex:
        answers = list(ex.map(oracle_read, jobs))
    return answers, dict(zip(MODES, mode_reads))


def test_cfg2_reads_of_the_launch_equal_the_oracle_at_full_size(launch, oracle_wave):
    db, reads, al, res, tm = launch
    answers, _ = oracle_wave
    local = [a for a in answers if a[0] == "local"]
    for _, k, want_keys, want_sam in local:
        got = res[k][0]
        assert [c.key() for c in got] == want_keys, f"read {k} differs from the oracle at n = {N}"
        assert al.format_sam(k, f"read_{k:07d}", reads[k], b"I" * N) == want_sam, f"SAM text of read {k}"
    assert len(local) >= 2


def test_cfg2_size_reads_in_the_other_clipping_modes_equal_the_oracle(launch, oracle_wave):
    """One 10 kb read per non-Local mode against the 50 x 5 kb contigs, through the 32-bit register-resident kernel
    (fill_regs32.hip: scores of these modes leave the 16-bit range), chains, operation lists and SAM text compared with the oracle at
    full size (aligners/constants.rs:96-136 for the modes' clip penalties, single_contig_aligner.rs:453-470 for the end-of-read jump)."""
    db = launch[0]
    answers, mode_reads = oracle_wave
    for mode, _, want_keys, want_sam in [a for a in answers if a[0] != "local"]:
        read = mode_reads[mode]
        al = stitch_amd.Builder(mode=mode).build_aligners([stitch_amd.TargetSeq(n, s) for n, s in db])
        res = al.align([read])
        tm = al.timing()
        assert tm["fill_kind"] == 3 and al.cells_filled == N * CONTIGS * M, (mode, tm)
        assert [c.key() for c in res[0][0]] == want_keys, f"mode {mode}: chains differ from the oracle at n = {N}"
        assert al.format_sam(0, "read_0000000", read, b"I" * N) == want_sam, f"mode {mode}: SAM text"
        del al
    assert len([a for a in answers if a[0] != "local"]) == len(MODES)
