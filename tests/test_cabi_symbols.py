"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/stitch_gpu.h
declares, round-trips the index blob and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import stitch_amd
from stitch_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "stitch_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(stitch_[a-z_0-9]+)\s*\(", hdr)))


def test_exports_match_header():
    syms = declared_symbols()
    assert set(syms) == set(api.EXPORTS), (syms, api.EXPORTS)
    L = api.lib()
    for s in syms:
        assert hasattr(L, s), s


def test_struct_sizes_match_header():
    assert C.sizeof(api._Op) == 8 and C.sizeof(api._Chain) == 56 and C.sizeof(api._ReadResult) == 24
    assert C.sizeof(api._Opts) == 24 * 4


def test_index_roundtrip():
    idx = stitch_amd.Index.from_targets([stitch_amd.TargetSeq("a", "acgtNN"), stitch_amd.TargetSeq("b b", "GGGCCC")])
    blob = idx.serialize()
    idx2 = stitch_amd.Index.deserialize(blob)
    assert idx2.n_contigs() == 2
    assert idx2.serialize() == blob
    assert b"ACGTNN" in blob          # upper-cased on build (util/target_seq.rs:111-115)
    with pytest.raises(stitch_amd.StitchError):
        stitch_amd.Index.deserialize(b"nope" + blob[4:])
    with pytest.raises(stitch_amd.StitchError):
        stitch_amd.Index.deserialize(blob[:-3])


def test_argument_errors():
    with pytest.raises(stitch_amd.StitchError):
        stitch_amd.Index.from_targets([])                                  # "Found no sequences in the FASTA"
    t = [stitch_amd.TargetSeq("a", "ACGT")]
    for bad in (dict(gap_open=1), dict(gap_extend=2), dict(default_jump_score=5)):
        with pytest.raises(stitch_amd.StitchError):                        # the reference's constructor asserts
            stitch_amd.Builder(**bad).build_aligners(t)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(stitch_amd.StitchError, match="no CPU path|HIP"):
        stitch_amd.Builder().build_aligners([stitch_amd.TargetSeq("a", "ACGT")])
