"""Pins the oracle (oracle/stitch_oracle.cpp) against every known-answer test the reference holds for the hot
path.  Fixtures: tests/golden/*.json (made by tests/golden/extract_golden.py)."""
import json
import os

import pytest

from oracle import oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SINGLE = json.load(open(os.path.join(G, "single_contig.json")))
MULTI = json.load(open(os.path.join(G, "multi_contig.json")))
ALN = json.load(open(os.path.join(G, "alignment.json")))


def rc(seq):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    return "".join(comp[c] for c in reversed(seq))


def resolve(seq):
    return rc(seq[3:]) if seq.startswith("rc:") else seq


def check(a, e):
    assert a.xstart == e["xstart"], a
    assert a.xend == e["xend"], a
    assert a.ystart == e["ystart"], a
    assert a.yend == e["yend"], a
    assert a.score == e["score"], a
    assert a.start_contig_idx == e["start_contig_idx"], a
    assert a.cigar() == e["cigar"], a
    assert a.length == e["length"], a


@pytest.mark.parametrize("t", SINGLE, ids=[t["name"] for t in SINGLE])
def test_single_contig_aligner(t):  # single_contig_aligner.rs:915-1773
    s = t["scoring"]
    sc = orc.scoring_array(match=s["match"], mismatch=s["mismatch"], gap_open=s["gap_open"], gap_extend=s["gap_extend"],
                           jump=s["jump"])
    a = orc.single_align(t["mode"], t["x"], t["y"], sc, t["circular"])
    check(a, t["expect"])


def multi_scoring(kind, sc, jumps=None):
    clip = orc.MIN_SCORE if kind == "global" else 0
    mismatch, go, ge, jump = sc
    js, jo, ji = jumps if jumps else (jump, jump, jump)
    return orc.scoring_array(match=1, mismatch=mismatch, gap_open=go, gap_extend=ge, jump_same=js, jump_opp=jo,
                             jump_inter=ji, xclip_prefix=clip, xclip_suffix=clip, yclip_prefix=clip, yclip_suffix=clip)


@pytest.mark.parametrize("t", MULTI, ids=[t["name"] for t in MULTI])
def test_multi_contig_aligner(t):  # multi_contig_aligner.rs:465-737
    al = orc.MultiContigAligner()
    for c in t["contigs"]:
        al.add_contig(c["name"], c["is_forward"], resolve(c["seq"]), False, multi_scoring(c["kind"], c["scoring"]))
    for case in t["cases"]:
        if case["jump_scores"]:
            c0 = t["contigs"][0]
            al.set_scoring(multi_scoring(c0["kind"], c0["scoring"], case["jump_scores"]))
        check(al.custom(resolve(t["y"])), case["expect"])


def mk(name):
    d = dict(ALN["alignments"][name])
    return orc.Alignment(**d)


@pytest.mark.parametrize("name", ALN["valid"])
def test_valid_alignments(name):  # alignment.rs:530-542
    assert mk(name).validate(), orc._err()


@pytest.mark.parametrize("name,contig,x", ALN["earliest_x_base"])
def test_earliest_x_base(name, contig, x):  # alignment.rs:544-563
    assert mk(name).earliest_x_base_for(contig) == x


@pytest.mark.parametrize("name,contig,x", ALN["latest_x_base"])
def test_latest_x_base_for(name, contig, x):  # alignment.rs:565-584
    assert mk(name).latest_x_base_for(contig) == x


@pytest.mark.parametrize("case", ALN["split_at_y"], ids=[c[0] for c in ALN["split_at_y"]])
def test_split_at_y(case):  # alignment.rs:679-707
    name, pivot, xstart, xend, ystart, yend, score, cigar, length = case
    a = mk(name).split_at_y(pivot)
    assert (a.xstart, a.xend, a.ystart, a.yend, a.score, a.start_contig_idx, a.cigar(), a.length) == \
        (xstart, xend, ystart, yend, score, 0, cigar, length), a


def test_packed_length_cell():  # traceback/packed_length_cell.rs:193-259
    import ctypes as C
    spec = json.load(open(os.path.join(G, "packed_cell.json")))
    L = orc.lib()

    def apply(cell, op, tb, ln, idx=0, frm=0):
        out = (C.c_uint32 * 8)()
        assert L.orc_cell_apply(cell, op, tb, ln, idx, frm, out) == 0
        return list(out)

    for op, lo in ((0, 0), (1, 2)):  # set_i / set_d
        cell = (C.c_uint32 * 4)()
        for tb in range(spec["tb_max"] + 1):
            assert apply(cell, op, 0, 0)[lo:lo + 2] == [0, 0]
            assert apply(cell, op, tb, spec["len_a"])[lo:lo + 2] == [tb, spec["len_a"]]
            apply(cell, op, 0, 0)
    cell = (C.c_uint32 * 4)()
    ln, idx, frm = spec["s_all"]
    for tb in range(spec["tb_max"] + 1):
        assert apply(cell, 3, 0, 0, 0, 0)[4:] == [0, 0, 0, 0]
        assert apply(cell, 2, tb, spec["len_a"])[4:] == [tb, spec["len_a"], 0, 0]
        assert apply(cell, 3, tb, ln, idx, frm)[4:] == [tb, ln, idx, frm]


def test_aligners_case_insensitive():  # aligners/mod.rs:984-1003
    spec = json.load(open(os.path.join(G, "aligners.json")))
    al = orc.Aligners([(spec["target"]["name"], spec["target"]["seq"])])
    chains = al.align(spec["read"]["seq"])
    assert len(chains) == spec["expect"]["n_chains"]
    assert chains[0].length == spec["expect"]["length"]
    assert chains[0].cigar() == spec["expect"]["cigar"]
    # lower-case input must give the same answer (seq_upper_case, align/io.rs:64-66)
    chains = al.align(spec["read"]["seq"].lower())
    assert chains[0].cigar() == spec["expect"]["cigar"]
