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
    assert C.sizeof(api._Timing) == 120


_CT = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "uint16_t": C.c_uint16, "uint8_t": C.c_uint8,
       "float": C.c_float, "double": C.c_double}
_RUST = {"i32": "int32_t", "u32": "uint32_t", "u64": "uint64_t", "u16": "uint16_t", "u8": "uint8_t", "f32": "float", "f64": "double"}


def header_structs():
    """{struct name: [(field, C type, array length or 0)]} of every `typedef struct NAME { ... } NAME;` in the header."""
    hdr = open(os.path.join(ROOT, "include", "stitch_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for name, body in re.findall(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", hdr, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty, rest = decl.split(None, 1)
            for f in rest.split(","):
                f = f.strip()
                m = re.fullmatch(r"(\w+)\[(\d+)\]", f)
                fields.append((m.group(1), ty, int(m.group(2))) if m else (f, ty, 0))
        out[name] = fields
    return out


def rust_block():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.search(r"```rust(.*?)```", doc, flags=re.S).group(1)


def rust_structs():
    """The same, read off the `#[repr(C)] pub struct` items INTEGRATION.md tells a maintainer to paste."""
    src = re.sub(r"/\*.*?\*/", "", rust_block(), flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for name, body in re.findall(r"pub\s+struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for f in re.findall(r"pub\s+(\w+)\s*:\s*(\[[^\]]+\]|\w+)", body):
            m = re.fullmatch(r"\[\s*(\w+)\s*;\s*(\d+)\s*\]", f[1])
            fields.append((f[0], _RUST[m.group(1)], int(m.group(2))) if m else (f[0], _RUST[f[1]], 0))
        out[name] = fields
    return out


def c_size(fields):
    class S(C.Structure):
        _fields_ = [(n, _CT[t] * k if k else _CT[t]) for n, t, k in fields]
    return C.sizeof(S)


def test_the_printed_rust_binding_matches_the_header():
    """The binding INTEGRATION.md prints is the one a maintainer pastes: every struct field for field (name, type, order, size) and
    every entry point of the header (StitchTiming was 24 bytes short of what stitch_last_timing writes)."""
    hs, rs = header_structs(), rust_structs()
    pairs = {"stitch_opts": "StitchOpts", "stitch_chain": "StitchChain", "stitch_op": "StitchOp", "stitch_read_result": "StitchReadResult",
             "stitch_timing": "StitchTiming"}
    assert set(hs) == set(pairs), sorted(hs)
    py = {"stitch_opts": api._Opts, "stitch_chain": api._Chain, "stitch_op": api._Op, "stitch_read_result": api._ReadResult, "stitch_timing": api._Timing}
    for cname, rname in pairs.items():
        assert rname in rs, rname
        assert [(t, k) for _, t, k in hs[cname]] == [(t, k) for _, t, k in rs[rname]], (cname, hs[cname], rs[rname])
        assert [n for n, _, _ in hs[cname]] == [n for n, _, _ in rs[rname]], (cname, hs[cname], rs[rname])
        assert c_size(hs[cname]) == c_size(rs[rname]) == C.sizeof(py[cname]), cname
        assert [n for n, _, _ in hs[cname]] == [n for n, _ in py[cname]._fields_], cname
    fns = set(re.findall(r"pub\s+fn\s+(stitch_\w+)", rust_block()))
    assert fns == set(declared_symbols()), (sorted(fns ^ set(declared_symbols())))


def test_last_timing_never_writes_beyond_the_callers_size():
    """stitch_last_timing copies min(out_size, sizeof(stitch_timing)) bytes (a binding built against an older, shorter struct)."""
    L = api.lib()
    buf = (C.c_uint8 * 256)(*([0xAB] * 256))
    # no context: the argument check comes first and nothing is written
    assert L.stitch_last_timing(None, C.cast(buf, C.c_void_p), C.c_size_t(88)) < 0
    assert bytes(buf) == bytes([0xAB] * 256)


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
