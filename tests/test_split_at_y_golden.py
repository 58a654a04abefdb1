"""The PRODUCT's Alignment::split_at_y (stitch_amd/csrc/host_align.cpp, used by realign_origin) on the reference's own
test vectors (align/alignment.rs:679-707, transcribed in tests/golden/alignment.json) — through the C ABI test hook
stitch_split_at_y; host code only, no device.  (tests/test_oracle_golden.py runs the same vectors through the oracle.)"""
import json
import os

import pytest

import stitch_amd
from stitch_amd import api

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALN = json.load(open(os.path.join(G, "alignment.json")))


def mk(name):
    d = ALN["alignments"][name]
    a = stitch_amd.Alignment()
    for f in stitch_amd.Alignment.__slots__[:-1]:
        setattr(a, f, d[f])
    a.operations = [tuple(o) for o in d["ops"]]
    return a, d["mode"]


@pytest.mark.parametrize("case", ALN["split_at_y"], ids=[c[0] for c in ALN["split_at_y"]])
def test_product_split_at_y(case):
    name, pivot, xstart, xend, ystart, yend, score, cigar, length = case
    a, mode = mk(name)
    r = api.split_at_y(a, mode, pivot)
    assert (r.xstart, r.xend, r.ystart, r.yend, r.score, r.start_contig_idx, r.cigar(), r.length) == \
        (xstart, xend, ystart, yend, score, 0, cigar, length), r
