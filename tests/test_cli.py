"""The `stitch-align` front end (stitch_amd/cli/stitch_align.cpp): argument handling, FASTA/FASTQ(.gz) parsing, the SAM
header and the BAM/BGZF encoder run without a GPU; the end-to-end comparison with the oracle's SAM text is a GPU test."""
import gzip
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def cli():
    from stitch_amd import build
    return build.build_cli()


def run(cli, *args, check=True):
    # (BAM is the default output, as the reference's; the tests that read records as text ask for SAM)
    if "-r" in args and "--output-format" not in args:
        args = (*args, "--output-format", "sam")
    r = subprocess.run([cli, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if check:
        assert r.returncode == 0, r.stderr.decode()
    return r


def write_inputs(tmp_path, gz=True):
    ref = tmp_path / "ref.fa"
    ref.write_text(">construct_00 some description\nACGTACGTAC\nggttaacc\n>construct_01\nTTTTGGGGCCCCAAAA\n")
    fq = tmp_path / ("reads.fq.gz" if gz else "reads.fq")
    text = "@read_1 extra words\nACGTACGT\n+\nIIIIIIII\n@read_2\nggttaacc\n+read_2\nABCDEFGH\n"
    if gz:
        with gzip.open(fq, "wt") as f:
            f.write(text)
    else:
        fq.write_text(text)
    return str(ref), str(fq)


def test_help_and_bad_arguments(cli):
    r = run(cli, "--help")
    assert b"--reads-fastq" in r.stdout and b"--ref-fasta" in r.stdout
    assert run(cli, "--no-such-flag", check=False).returncode == 2
    assert run(cli, "-f", "a.fq", check=False).returncode == 2                       # no reference
    r = run(cli, "-f", "a.fq", "-a", "b.fa", "-r", "c.fa", check=False)
    assert r.returncode == 2 and b"exactly one of" in r.stderr


@pytest.mark.parametrize("gz", [True, False])
def test_dry_run_parses_inputs_and_writes_the_header(cli, tmp_path, gz):
    ref, fq = write_inputs(tmp_path, gz)
    r = run(cli, "-f", fq, "-r", ref, "--dry-run", "-d", "-x", "true", "-m", "Local", "-P", "score", "-J", "-12")
    lines = r.stdout.decode().splitlines()
    assert lines[0] == "@HD\tVN:1.6"
    assert lines[1] == "@SQ\tSN:construct_00\tLN:18" and lines[2] == "@SQ\tSN:construct_01\tLN:16"
    assert lines[3].startswith("@PG\tID:stitch\tPN:stitch\tVN:") and "--dry-run" in lines[3]
    assert len(lines) == 4
    assert b"2 targets, 2 reads, 16 bases" in r.stderr


def test_fasta_reads(cli, tmp_path):
    ref, _ = write_inputs(tmp_path)
    fa = tmp_path / "reads.fa"
    fa.write_text(">r1\nACGT\nACGT\n>r2\nGG\n")
    r = run(cli, "-a", str(fa), "-r", ref, "--dry-run", "-p", "-k", "10", "-w", "20", "-s", "30", "-x", "false")
    assert b"2 reads, 10 bases" in r.stderr


def read_bam(data):
    raw = gzip.decompress(data)             # BGZF = concatenated gzip members
    assert raw[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", raw, 4)
    text = raw[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p); p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p); p += 4
        name = raw[p:p + l_name - 1].decode(); p += l_name
        l_ref, = struct.unpack_from("<i", raw, p); p += 4
        refs.append((name, l_ref))
    recs = []
    while p < len(raw):
        bs, = struct.unpack_from("<i", raw, p); p += 4
        b = raw[p:p + bs]; p += bs
        ref, pos, l_name, mapq, bin_, n_cig, flag, l_seq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
        q = 32
        name = b[q:q + l_name - 1].decode(); q += l_name
        cig = "".join(f"{v >> 4}{'MIDNSHP=X'[v & 15]}" for v in struct.unpack_from(f"<{n_cig}I", b, q)); q += 4 * n_cig
        sq = "".join("=ACMGRSVTWYHKDBN"[b[q + k // 2] >> (4 if k % 2 == 0 else 0) & 15] for k in range(l_seq)); q += (l_seq + 1) // 2
        qual = b[q:q + l_seq]; q += l_seq
        tags = []
        while q < len(b):
            tag, ty = b[q:q + 2].decode(), chr(b[q + 2]); q += 3
            if ty == "i":
                v, = struct.unpack_from("<i", b, q); q += 4
            elif ty == "A":
                v = chr(b[q]); q += 1
            elif ty == "f":
                v, = struct.unpack_from("<f", b, q); q += 4
            elif ty == "Z":
                e = b.index(b"\0", q); v = b[q:e].decode(); q = e + 1
            else:
                raise AssertionError(ty)
            tags.append((tag, ty, v))
        recs.append(dict(ref=ref, pos=pos, mapq=mapq, bin=bin_, flag=flag, name=name, cigar=cig, seq=sq, qual=qual, nref=nref,
                         npos=npos, tlen=tlen, tags=tags))
    return text, refs, recs


SAM = ("@HD\tVN:1.6\n@SQ\tSN:c0\tLN:5000\n@SQ\tSN:c1\tLN:70000\n@PG\tID:stitch\tPN:stitch\n"
       "r1\t0\tc0\t101\t60\t3S5M1I2D4=1X\t*\t0\t0\tACGTACGTACGTAC\tIIIIIIIIIIIIII\tAS:i:42\tNM:i:3\tSA:Z:c1,5,+,4M,60,0;\tXx:A:q\txf:f:1.5\n"
       "r1\t2064\tc1\t65537\t0\t5H9M\t*\t0\t0\tACGTACGTA\t*\tqs:i:5\n"
       "r2\t4\t*\t0\t0\t*\t*\t0\t0\tNNACGT\t!!~~II\n")


@pytest.mark.parametrize("level", ["0", "6"])
def test_bam_encoder(cli, tmp_path, level):
    sam = tmp_path / "x.sam"
    sam.write_text(SAM)
    r = run(cli, "--convert-sam", str(sam), "-c", level)
    assert r.stdout.endswith(bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))   # BGZF EOF block
    text, refs, recs = read_bam(r.stdout)
    assert text == "".join(l + "\n" for l in SAM.splitlines() if l.startswith("@"))
    assert refs == [("c0", 5000), ("c1", 70000)]
    a, b, c = recs
    assert (a["name"], a["flag"], a["ref"], a["pos"], a["mapq"], a["cigar"]) == ("r1", 0, 0, 100, 60, "3S5M1I2D4=1X")
    assert a["seq"] == "ACGTACGTACGTAC" and a["qual"] == bytes([40] * 14) and a["bin"] == 4681 + (100 >> 14)
    assert a["tags"] == [("AS", "i", 42), ("NM", "i", 3), ("SA", "Z", "c1,5,+,4M,60,0;"), ("Xx", "A", "q"), ("xf", "f", 1.5)]
    assert (b["flag"], b["ref"], b["pos"], b["cigar"], b["seq"]) == (2064, 1, 65536, "5H9M", "ACGTACGTA") and b["qual"] == b"\xff" * 9
    assert b["bin"] == 4681 + (65536 >> 14) and b["tags"] == [("qs", "i", 5)]
    assert (c["flag"], c["ref"], c["pos"], c["cigar"], c["seq"]) == (4, -1, -1, "", "NNACGT") and c["qual"] == bytes([0, 0, 93, 93, 40, 40])


# ---- end to end on the GPU: SAM text of the CLI == the oracle's SamRecordFormatter, read by read --------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-d", "--suboptimal", "-X"], ["-C", "-S", "-P", "score"], ["-p", "-k", "10", "-w", "25", "-s", "40", "-d"]])
def test_cli_matches_oracle_sam(cli, tmp_path, extra):
    import random
    from oracle import oracle as orc
    from stitch_amd import synth
    rng = random.Random(7)
    db = synth.make_db(4, 600, 77)
    reads = [r.decode() for r in synth.make_reads(db, 12, 300, 5)]
    reads[3] = reads[2]                                            # a duplicate run
    ref = tmp_path / "ref.fa"
    ref.write_text("".join(f">{n} desc\n{s.decode().lower() if k % 2 else s.decode()}\n" for k, (n, s) in enumerate(db)))
    fq = tmp_path / "reads.fq"
    quals = ["".join(rng.choice("!5?I") for _ in r) for r in reads]
    fq.write_text("".join(f"@read_{k} x y\n{r}\n+\n{q}\n" for k, (r, q) in enumerate(zip(reads, quals))))
    out = run(cli, "-f", str(fq), "-r", str(ref), "--batch", "5", *extra).stdout.decode().splitlines()
    head = [l for l in out if l.startswith("@")]
    assert head[0] == "@HD\tVN:1.6" and [l.split("\t")[1] for l in head[1:-1]] == [f"SN:{n}" for n, _ in db]
    got = [l for l in out if not l.startswith("@")]
    opts = dict(double_strand="-d" in extra, suboptimal="--suboptimal" in extra, use_eq_and_x="-X" in extra, circular="-C" in extra,
                soft_clip="-S" in extra, pick_primary=1 if "score" in extra else 0)
    if "-p" in extra:
        opts.update(pre_align=True, kmer_size=10, band_width=25, pre_align_min_score=40)
    o = orc.Aligners([(n, s.decode()) for n, s in db], **opts)
    want = []
    for k, (r, q) in enumerate(zip(reads, quals)):
        o.align(r)
        want += o.format_sam(f"read_{k} x y", r, q, prealign=o.prealign_score())
    assert got == want
    # the BAM route carries the same records
    text, refs, recs = read_bam(run(cli, "-f", str(fq), "-r", str(ref), "--output-format", "bam", "-c", "1", *extra).stdout)
    assert refs == [(n, len(s)) for n, s in db] and len(recs) == len(want)
    for rec, line in zip(recs, want):
        f = line.split("\t")
        assert (rec["name"], rec["flag"], rec["pos"] + 1, rec["cigar"] or "*") == (f[0], int(f[1]), int(f[3]), f[5])


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-d", "--suboptimal"]])
def test_two_worker_processes_write_what_one_device_writes(cli, tmp_path, extra):
    """`--devices A,B`: one worker process per GPU (here both on the one GPU of the test box), the index handed over as the
    serialized blob, the read stream cut at read-group boundaries (stitch_shard_range), records concatenated in rank order:
    byte-identical to the single-device output apart from the @PG line (which quotes the command line) — SAM and BAM."""
    from stitch_amd import synth
    db = synth.make_db(5, 700, 31)
    reads = [r.decode() for r in synth.make_reads(db, 31, 250, 9, dup_every=5)]
    reads[15] = reads[14] = reads[16]                              # a run of identical reads across the middle cut: it must stay on one rank
    ref = tmp_path / "ref.fa"
    ref.write_text("".join(f">{n}\n{s.decode()}\n" for n, s in db))
    fq = tmp_path / "reads.fq"
    fq.write_text("".join(f"@read_{k}\n{r}\n+\n{'I' * len(r)}\n" for k, r in enumerate(reads)))
    one = run(cli, "-f", str(fq), "-r", str(ref), "--device", "0", "--batch", "7", *extra).stdout.decode().splitlines()
    two = run(cli, "-f", str(fq), "-r", str(ref), "--devices", "0,0", "--batch", "7", *extra).stdout.decode().splitlines()
    strip = lambda ls: [l for l in ls if not l.startswith("@PG")]
    assert strip(one) == strip(two) and len(strip(one)) > len(reads)
    three = run(cli, "-f", str(fq), "-r", str(ref), "--devices", "0,0,0", "--batch", "4", *extra).stdout.decode().splitlines()
    assert strip(one) == strip(three)
    t1, r1, c1 = read_bam(run(cli, "-f", str(fq), "-r", str(ref), "--output-format", "bam", *extra).stdout)
    t2, r2, c2 = read_bam(run(cli, "-f", str(fq), "-r", str(ref), "--output-format", "bam", "--devices", "0,0", *extra).stdout)
    assert r1 == r2 and c1 == c2


@pytest.mark.gpu
def test_workers_seek_to_their_block_in_gzip_fastq_and_multi_line_fasta(cli, tmp_path):
    """The parent scans the read file once (offset, length and hash per record) and every worker SEEKS to its block instead of
    parsing the blocks before it: gzip FASTQ (zlib inflates up to the offset) and multi-line FASTA with blank lines (a record's
    offset is that of its '>' line, read ahead by the record before it) give what one device writes."""
    import gzip
    from stitch_amd import synth
    db = synth.make_db(3, 500, 41)
    reads = [r.decode() for r in synth.make_reads(db, 23, 180, 19, dup_every=4)]
    ref = tmp_path / "ref.fa"
    ref.write_text("".join(f">{n}\n{s.decode()}\n" for n, s in db))
    fq = tmp_path / "reads.fq.gz"
    with gzip.open(fq, "wt") as f:
        f.write("".join(f"@r{k} c\n{r}\n+\n{'I' * len(r)}\n" for k, r in enumerate(reads)))
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(f">r{k}\n" + "\n".join(r[p:p + 50] for p in range(0, len(r), 50)) + ("\n\n" if k % 3 == 0 else "\n") for k, r in enumerate(reads)))
    strip = lambda ls: [l for l in ls if not l.startswith("@PG")]
    for flag, path in (("-f", fq), ("-a", fa)):
        one = run(cli, flag, str(path), "-r", str(ref), "--batch", "6").stdout.decode().splitlines()
        many = run(cli, flag, str(path), "-r", str(ref), "--devices", "0,0,0,0", "--batch", "3").stdout.decode().splitlines()
        assert strip(one) == strip(many) and len(strip(one)) > len(reads)
    assert not [p for p in os.listdir("/tmp") if p.startswith("stitch-align-")]          # the temporary directory is gone


@pytest.mark.gpu
def test_a_failing_worker_leaves_nothing_behind(cli, tmp_path):
    """a worker that cannot start (GPU ordinal that does not exist): the parent reports it, ends the other workers and removes its files"""
    from stitch_amd import synth
    db = synth.make_db(2, 300, 5)
    ref = tmp_path / "ref.fa"; ref.write_text("".join(f">{n}\n{s.decode()}\n" for n, s in db))
    fq = tmp_path / "r.fq"; fq.write_text("".join(f"@r{k}\n{r.decode()}\n+\n{'I' * len(r)}\n" for k, r in enumerate(synth.make_reads(db, 6, 120, 3))))
    r = subprocess.run([cli, "-f", str(fq), "-r", str(ref), "--devices", "0,97"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"worker" in r.stderr
    assert not [p for p in os.listdir("/tmp") if p.startswith("stitch-align-")]


def test_a_worker_that_dies_ends_a_stalled_one_promptly(cli, tmp_path):
    """the parent reaped its workers in rank order, so a rank stuck for good (inside an RCCL bootstrap whose partner
    died) held it forever.  Here the worker of device 5 stalls (test hook) and the other cannot start (no such device / no GPU): the
    parent must report the failure, end the stalled worker and remove its files within seconds.  Runs without a GPU."""
    import time
    from stitch_amd import synth
    db = synth.make_db(2, 300, 5)
    ref = tmp_path / "ref.fa"; ref.write_text("".join(f">{n}\n{s.decode()}\n" for n, s in db))
    fq = tmp_path / "r.fq"; fq.write_text("".join(f"@r{k}\n{r.decode()}\n+\n{'I' * len(r)}\n" for k, r in enumerate(synth.make_reads(db, 6, 120, 3))))
    before = set(p for p in os.listdir("/tmp") if p.startswith("stitch-align-"))
    t0 = time.time()
    r = subprocess.run([cli, "-f", str(fq), "-r", str(ref), "--devices", "5,97"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, STITCH_ALIGN_TEST_STALL_DEVICE="5"), timeout=120)
    assert r.returncode != 0 and b"worker 1 failed" in r.stderr, r.stderr.decode()[-800:]
    assert time.time() - t0 < 60
    assert set(p for p in os.listdir("/tmp") if p.startswith("stitch-align-")) <= before


@pytest.mark.gpu
def test_index_broadcast_over_rccl(cli, tmp_path):
    """`--index-via rccl`: the workers get the reference index through a native ncclBroadcast from rank 0 instead of reading the blob
    file.  RCCL refuses two ranks on one device, so the one GPU of the test box runs a world of ONE worker here (communicator
    created from the published unique id, both broadcasts issued); the output must be what the plain run writes."""
    from stitch_amd import synth
    db = synth.make_db(3, 400, 8)
    ref = tmp_path / "ref.fa"; ref.write_text("".join(f">{n}\n{s.decode()}\n" for n, s in db))
    fq = tmp_path / "r.fq"; fq.write_text("".join(f"@r{k}\n{r.decode()}\n+\n{'I' * len(r)}\n" for k, r in enumerate(synth.make_reads(db, 9, 150, 4))))
    strip = lambda ls: [l for l in ls if not l.startswith("@PG")]
    one = run(cli, "-f", str(fq), "-r", str(ref)).stdout.decode().splitlines()
    r = subprocess.run([cli, "-f", str(fq), "-r", str(ref), "--devices", "0", "--index-via", "rccl", "--output-format", "sam"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert b"over RCCL" in r.stderr
    assert strip(r.stdout.decode().splitlines()) == strip(one)
    # two ranks on one device: refused up front, with a message (not a hang inside RCCL)
    r2 = subprocess.run([cli, "-f", str(fq), "-r", str(ref), "--devices", "0,0", "--index-via", "rccl"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r2.returncode != 0 and b"one GPU per worker" in r2.stderr
