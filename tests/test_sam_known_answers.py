"""Hand-traced known answers for the host sections the reference holds no test for: SubAlignmentBuilder::build
(sub_alignment.rs:36-241) and SamRecordFormatter::format (aligners/mod.rs:622-973).

Product (stitch_amd/csrc/host_align.cpp) and oracle (oracle/stitch_oracle.cpp) were written from the same reading of the
reference, so comparing them with each other cannot see a shared misreading.  The vectors of tests/golden/sam_known_answers.json
were derived BY HAND from the Rust text (no program produced the expected lines) and are checked against BOTH.  How each was
traced, for the first vector (default scoring 1 / -4 / -6 / -2, M cigars):

  chain: xstart 10, ystart 2, contig 0, ops  M M X M  Xjump(1, 30)  X M M I M
  build (the builder's "query" is the contig x, its "target" the read y, until the final swap):
    i=2  X: num_edits 0 -> 1; cmp_op(M, X) is true without =/X, the run grows to MMXM
    i=4  Xjump: run of 4 flushed with ITS LAST OP (Match): score += 4 x 1; offsets x 10 -> 14, y 2 -> 6; cigar 4M
    i=5  X: num_edits 1 -> 2 FIRST (the counter is bumped before the flush), then the pending Xjump is flushed: sub 0 =
         {contig 0, x 10..14, y 2..6, 4M, score 4, NM 2}; reset: contig 1, x 30, y 6, NM 0
    i=8  I: num_edits 1; run XMM flushed with its last op (Match): score 3, x 33, y 9
    i=9  M: Ins flushed: score 3 - 6 - 2 = -5, x 34, cigar 3M 1I
    end  M flushed: score -4, x 35, y 10; sub 1 = {contig 1, x 30..35, y 6..10, 3M1I1M, score -4, NM 1}
    swap: query = read span, target = contig span, I <-> D: sub 0 q 2..6 t 10..14 4M; sub 1 q 6..10 t 30..35 3M1D1M
  format: primary = longest query span, ties by score: (4, 4) beats (4, -4) -> sub 0; soft clips 2S...14S and 6S...10S;
    POS = target_start + 1; SA entries in sub order rotated right by the primary's index (0).

The other vectors are traced the same way; their JSON entries say which quirk each pins (`why`)."""
import json
import os

import pytest

import stitch_amd
from oracle import oracle as orc
from stitch_amd import api

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sam_known_answers.json")
VECTORS = json.load(open(G))["vectors"]
REN = {"match_score": "match", "mismatch_score": "mismatch", "default_jump_score": "jump_score"}


def chains_of(v, cls):
    out = []
    for c in v["chains"]:
        if cls is orc.Alignment:
            out.append(orc.Alignment(mode=4, **{k: c[k] for k in c}))
        else:
            a = stitch_amd.Alignment()
            for f in stitch_amd.Alignment.__slots__[:-1]:
                setattr(a, f, c[f])
            a.operations = [tuple(o) for o in c["ops"]]
            out.append(a)
    return out


@pytest.mark.parametrize("v", VECTORS, ids=[v["name"] for v in VECTORS])
def test_oracle_reproduces_the_hand_traced_sam_lines(v):
    opts = {REN.get(k, k): (({"query-length": 0, "score": 1}[x]) if k == "pick_primary" else x) for k, x in v["options"].items()}
    o = orc.Aligners([(n, "A" * l) for n, l in v["targets"]], **opts)
    o.set_chains(chains_of(v, orc.Alignment))
    got = o.format_sam(v["head"], v["read"], v["quals"], prealign=v["prealign"])
    assert got == v["expect"]


@pytest.mark.parametrize("v", VECTORS, ids=[v["name"] for v in VECTORS])
def test_product_host_code_reproduces_the_hand_traced_sam_lines(v):
    """stitch_format_sam_chains: the library's SubAlignmentBuilder / SamRecordFormatter (host code, no device needed)"""
    got = api.format_sam_chains(stitch_amd.Builder(**v["options"]), v["targets"], v["head"], v["read"], v["quals"], chains_of(v, stitch_amd.Alignment),
                                prealign=v["prealign"])
    assert got == v["expect"]
