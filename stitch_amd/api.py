"""Host-side mirror of the reference's interface for the `stitch align` hot path, over the C ABI in
include/stitch_gpu.h (ctypes only: no torch types cross the boundary).

Reference (fg-stitch-lib/src/align)            here
  aligners/mod.rs:65-225   Builder / Options    Builder (same setter names, same defaults)
  util/target_seq.rs:15-36 TargetSeq            TargetSeq
  aligners/mod.rs:227-340  Aligners::align      Aligners.align (batch) / Aligners.align_one
  alignment.rs:16-149      Alignment, cigar()   Alignment
  aligners/mod.rs:606-973  SamRecordFormatter   Aligners.format_sam (SAM text)

There is no CPU fallback: importing works anywhere (so that the symbol table can be checked), but creating
Aligners without a HIP device raises StitchError.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MODES = {"local": 0, "query-local": 1, "target-local": 2, "global": 3}
OP_NAMES = ("Match", "Subst", "Del", "Ins", "Xclip", "Yclip", "Xjump", "Yjump")


class StitchError(RuntimeError):
    pass


class _Opts(C.Structure):  # == stitch_opts
    _fields_ = [("mode", C.c_int32), ("match_score", C.c_int32), ("mismatch_score", C.c_int32), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32), ("jump_same", C.c_int32), ("jump_opposite", C.c_int32), ("jump_inter", C.c_int32),
                ("double_strand", C.c_int32), ("circular", C.c_int32), ("circular_slop", C.c_int32), ("pre_align", C.c_int32),
                ("pre_align_min_score", C.c_int32), ("pre_align_subset_contigs", C.c_int32), ("kmer_size", C.c_int32),
                ("band_width", C.c_int32), ("suboptimal", C.c_int32), ("suboptimal_pct", C.c_float), ("soft_clip", C.c_int32),
                ("use_eq_and_x", C.c_int32), ("pick_primary", C.c_int32), ("filter_secondary", C.c_int32),
                ("filter_secondary_pct", C.c_float), ("keep_clipping", C.c_int32)]


class _Chain(C.Structure):  # == stitch_chain
    _fields_ = [("score", C.c_int32), ("xstart", C.c_uint32), ("xend", C.c_uint32), ("ystart", C.c_uint32), ("yend", C.c_uint32),
                ("xlen", C.c_uint32), ("ylen", C.c_uint32), ("start_contig_idx", C.c_uint32), ("end_contig_idx", C.c_uint32),
                ("length", C.c_uint32), ("ops_begin", C.c_uint64), ("ops_len", C.c_uint32), ("pad", C.c_uint32)]


class _Op(C.Structure):  # == stitch_op
    _fields_ = [("kind", C.c_uint8), ("pad", C.c_uint8), ("contig", C.c_uint16), ("arg", C.c_uint32)]


class _ReadResult(C.Structure):  # == stitch_read_result
    _fields_ = [("chains_begin", C.c_uint64), ("n_chains", C.c_uint32), ("prealign_score", C.c_int32), ("has_prealign", C.c_uint8),
                ("pad", C.c_uint8 * 3)]


class _Timing(C.Structure):  # == stitch_timing
    _fields_ = [("fill_ms", C.c_double), ("walk_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("host_ms", C.c_double), ("cells", C.c_uint64), ("launches", C.c_uint32), ("jobs", C.c_uint32),
                ("prealign_ms", C.c_double), ("prealign_host_ms", C.c_double), ("fill_kind", C.c_uint32), ("wg_per_read", C.c_uint32), ("fallbacks", C.c_uint32), ("stream_runs", C.c_uint32),
                ("clk_shader_cycles", C.c_uint64), ("clk_ref_ticks", C.c_uint64), ("fill_kernel_ms", C.c_double), ("teams_retired", C.c_uint32), ("reserved_", C.c_uint32)]


EXPORTS = ("stitch_opts_default", "stitch_index_build", "stitch_index_serialize", "stitch_index_deserialize",
           "stitch_index_n_contigs", "stitch_index_destroy", "stitch_ctx_create", "stitch_ctx_destroy", "stitch_align_batch",
           "stitch_format_sam", "stitch_format_sam_chains", "stitch_last_timing", "stitch_prealign_band", "stitch_prealign_band_device", "stitch_split_at_y", "stitch_shard_range", "stitch_last_error", "stitch_version")

_lib = None


def lib():
    """Loads libstitch_amd.so (building it with hipcc if the sources are newer).  Fails loudly if it cannot."""
    global _lib
    if _lib is None:
        path = _build.LIB_PATH
        if _build.needs_build():
            path = _build.build()
        L = C.CDLL(path)
        L.stitch_last_error.restype = C.c_char_p
        L.stitch_version.restype = C.c_char_p
        L.stitch_format_sam.restype = C.c_long
        L.stitch_format_sam_chains.restype = C.c_long
        L.stitch_split_at_y.restype = C.c_long
        L.stitch_index_n_contigs.restype = C.c_uint32
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise StitchError(f"stitch error {rc}: {lib().stitch_last_error().decode()}")


class TargetSeq:
    """util/target_seq.rs:15-36 (the reverse complement is derived inside the library)."""

    def __init__(self, name, seq, circular=False):
        self.name = name
        self.fwd = seq.encode() if isinstance(seq, str) else bytes(seq)
        self.circular = circular

    def __len__(self):
        return len(self.fwd)


class Alignment:
    """align/alignment.rs:16-51.  operations: tuples (kind, a, b) with kind as aligners/constants.rs:20-29:
    0 Match 1 Subst 2 Del 3 Ins 4 Xclip(a) 5 Yclip(a) 6 Xjump(a=contig, b=x) 7 Yjump(a)."""

    __slots__ = ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen", "start_contig_idx", "end_contig_idx", "length",
                 "operations")

    def key(self):
        return (self.score, self.xstart, self.xend, self.ystart, self.yend, self.xlen, self.ylen, self.start_contig_idx,
                self.end_contig_idx, self.length, tuple(self.operations))

    def cigar(self):
        """Debug cigar of alignment.rs:105-149 / constants.rs:37-59."""
        out, contig, x = [], self.start_contig_idx, self.xstart
        last, run = None, 0

        def flush():
            if run > 0:
                out.append(f"{run}{'=XDI'[last[0]]}")

        for op in self.operations:
            kind, a, b = op
            special = kind in (4, 5, 6)
            if (special or op != last) and run > 0:
                flush()
            if special:
                if kind == 4:
                    out.append(f"{a}A"); x += a
                elif kind == 5:
                    out.append(f"{a}B")
                else:
                    s = f"{a - contig}C" if a > contig else f"{contig - a}c" if a < contig else ""
                    out.append(s + (f"{b - x}J" if b >= x else f"{x - b}j"))
                    x = b; contig = a
                last, run = op, 0
            else:
                if kind in (0, 1, 3):
                    x += 1
                if op == last:
                    run += 1
                else:
                    last, run = op, 1
        flush()
        return "".join(out)

    def __repr__(self):
        return (f"contig-idx: {self.start_contig_idx}-{self.end_contig_idx} x-span: {self.xstart}-{self.xend}/{self.xlen} "
                f"y-span: {self.ystart}-{self.yend}/{self.ylen} score: {self.score} cigar: {self.cigar()} aln-len: {self.length}")


def split_at_y(aln, mode, y_pivot):
    """Alignment::split_at_y (alignment.rs:207-360) as the library implements it (host code; no device needed)."""
    ops = (_Op * max(1, len(aln.operations)))()
    for k, (kind, a, b) in enumerate(aln.operations):
        ops[k].kind = kind
        ops[k].contig = a if kind == 6 else 0
        ops[k].arg = b if kind == 6 else a
    c = _Chain()
    for f in Alignment.__slots__[:-1]:
        setattr(c, f, getattr(aln, f))
    c.ops_len = len(aln.operations)
    out, cap = _Chain(), 2 * len(aln.operations) + 16
    out_ops = (_Op * cap)()
    n = lib().stitch_split_at_y(C.byref(c), ops, int(MODES[mode]) if isinstance(mode, str) else int(mode), C.c_uint32(y_pivot), C.byref(out), out_ops, C.c_uint32(cap))
    if n < 0 or n > cap:
        raise StitchError(lib().stitch_last_error().decode())
    r = Alignment()
    for f in Alignment.__slots__[:-1]:
        setattr(r, f, int(getattr(out, f)))
    r.operations = [(int(o.kind), int(o.contig), int(o.arg)) if o.kind == 6 else (int(o.kind), int(o.arg) if o.kind in (4, 5, 7) else 0, 0) for o in out_ops[:n]]
    return r


def format_sam_chains(builder, targets, head, bases, quals, chains, prealign=None):
    """SamRecordFormatter::format (mod.rs:622-973) on caller-supplied chains, as the library's host code implements it (no device
    needed).  `targets` = [(name, length)], `chains` = [Alignment]; returns the SAM lines."""
    names = (C.c_char_p * len(targets))(*[n.encode() for n, _ in targets])
    lens = (C.c_uint32 * len(targets))(*[int(l) for _, l in targets])
    n_ops = sum(len(a.operations) for a in chains)
    ops = (_Op * max(1, n_ops))()
    chs = (_Chain * max(1, len(chains)))()
    at = 0
    for k, aln in enumerate(chains):
        for f in Alignment.__slots__[:-1]:
            setattr(chs[k], f, getattr(aln, f))
        chs[k].ops_begin, chs[k].ops_len = at, len(aln.operations)
        for kind, a, b in aln.operations:
            ops[at].kind = kind
            ops[at].contig = a if kind == 6 else 0
            ops[at].arg = b if kind == 6 else a
            at += 1
    b = bases.encode() if isinstance(bases, str) else bytes(bases)
    q = None if quals is None else (quals.encode() if isinstance(quals, str) else bytes(quals))
    opts = builder.build_options()
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        n = lib().stitch_format_sam_chains(C.byref(opts), names, lens, C.c_uint32(len(targets)), head.encode(), b, q, C.c_size_t(len(b)), chs,
                                           C.c_uint32(len(chains)), ops, int(prealign is not None), C.c_int32(prealign or 0), buf, C.c_size_t(cap))
        if n < 0:
            raise StitchError(lib().stitch_last_error().decode())
        if n < cap:
            return buf.value.decode().split("\n")
        cap = n + 1


class Builder:
    """aligners/mod.rs:65-116 — derive_builder style: every option is a chainable setter with the reference's name."""

    _DEFAULTS = dict(mode="local", match_score=1, mismatch_score=-4, gap_open=-6, gap_extend=-2, default_jump_score=-10,
                     jump_score_same_contig_and_strand=None, jump_score_same_contig_opposite_strand=None,
                     jump_score_inter_contig=None, kmer_size=12, band_width=50, double_strand=False, circular=False,
                     circular_slop=20, pre_align=False, pre_align_min_score=100, pre_align_subset_contigs=True,
                     suboptimal=False, suboptimal_pct=20.0, soft_clip=False, use_eq_and_x=False, pick_primary="query-length",
                     filter_secondary=False, filter_secondary_pct=10.0, keep_clipping=False)

    def __init__(self, **kw):
        self.o = dict(self._DEFAULTS)
        for k, v in kw.items():
            if k not in self.o:
                raise TypeError(f"unknown option {k}")
            self.o[k] = v

    def __getattr__(self, name):
        if name in Builder._DEFAULTS:
            def setter(value):
                self.o[name] = value
                return self
            return setter
        raise AttributeError(name)

    def build_options(self):
        o, s = self.o, _Opts()
        lib().stitch_opts_default(C.byref(s))
        s.mode = MODES[o["mode"]] if isinstance(o["mode"], str) else int(o["mode"])
        s.match_score, s.mismatch_score, s.gap_open, s.gap_extend = o["match_score"], o["mismatch_score"], o["gap_open"], o["gap_extend"]
        dj = o["default_jump_score"]                                  # Options::contig_scoring, mod.rs:143-152
        s.jump_same = dj if o["jump_score_same_contig_and_strand"] is None else o["jump_score_same_contig_and_strand"]
        s.jump_opposite = dj if o["jump_score_same_contig_opposite_strand"] is None else o["jump_score_same_contig_opposite_strand"]
        s.jump_inter = dj if o["jump_score_inter_contig"] is None else o["jump_score_inter_contig"]
        s.double_strand, s.circular, s.circular_slop = int(o["double_strand"]), int(o["circular"]), o["circular_slop"]
        s.pre_align, s.pre_align_min_score = int(o["pre_align"]), o["pre_align_min_score"]
        s.pre_align_subset_contigs, s.kmer_size, s.band_width = int(o["pre_align_subset_contigs"]), o["kmer_size"], o["band_width"]
        s.suboptimal, s.suboptimal_pct = int(o["suboptimal"]), o["suboptimal_pct"]
        s.soft_clip, s.use_eq_and_x = int(o["soft_clip"]), int(o["use_eq_and_x"])
        s.pick_primary = {"query-length": 0, "score": 1}.get(o["pick_primary"], o["pick_primary"])
        s.filter_secondary, s.filter_secondary_pct = int(o["filter_secondary"]), o["filter_secondary_pct"]
        s.keep_clipping = int(o["keep_clipping"])
        return s

    def build_aligners(self, target_seqs, device=0):
        """Builder::build_aligners (mod.rs:171-211) + build_sam_record_formatter (:213-225)."""
        return Aligners(self.build_options(), target_seqs, device)


class Index:
    """The reference index (&[TargetSeq]) as the library holds it; serialisable for the one-time broadcast."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_targets(cls, target_seqs):
        n = len(target_seqs)
        names = (C.c_char_p * n)(*[t.name.encode() for t in target_seqs])
        bufs = [(C.c_uint8 * len(t.fwd)).from_buffer_copy(t.fwd) for t in target_seqs]
        seqs = (C.POINTER(C.c_uint8) * n)(*[C.cast(b, C.POINTER(C.c_uint8)) for b in bufs])
        lens = (C.c_uint32 * n)(*[len(t.fwd) for t in target_seqs])
        h = C.c_void_p()
        _check(lib().stitch_index_build(names, seqs, lens, n, C.byref(h)))
        return cls(h)

    def serialize(self):
        n = C.c_size_t(0)
        _check(lib().stitch_index_serialize(self.h, None, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        _check(lib().stitch_index_serialize(self.h, buf, C.byref(n)))
        return bytes(buf)

    @classmethod
    def deserialize(cls, blob):
        h = C.c_void_p()
        b = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(lib().stitch_index_deserialize(b, C.c_size_t(len(blob)), C.byref(h)))
        return cls(h)

    def n_contigs(self):
        return int(lib().stitch_index_n_contigs(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().stitch_index_destroy(self.h)
            self.h = None


class Aligners:
    """aligners/mod.rs:227-340 for one device.  Not thread-safe (like `&mut Aligners`)."""

    def __init__(self, opts, target_seqs, device=0):
        self.opts = opts
        self.index = target_seqs if isinstance(target_seqs, Index) else Index.from_targets(target_seqs)
        self.h = C.c_void_p()
        _check(lib().stitch_ctx_create(int(device), self.index.h, C.byref(opts), C.byref(self.h)))
        self.cells_filled = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().stitch_ctx_destroy(self.h)
            self.h = None

    def align(self, reads):
        """Aligners::align for a batch of reads (bytes/str) -> list of (chains, pre_align_score or None), in input order."""
        seqs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        cat = b"".join(seqs)
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        return self.align_packed(np.frombuffer(cat, dtype=np.uint8), offs)

    def align_packed(self, bases, offsets):
        """Same, from a concatenated uint8 array and uint64 offsets (len n_reads+1)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        rr, ch, op = C.POINTER(_ReadResult)(), C.POINTER(_Chain)(), C.POINTER(_Op)()
        cells = C.c_uint64(0)
        _check(lib().stitch_align_batch(self.h, bases.ctypes.data_as(C.POINTER(C.c_uint8)), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        C.c_uint32(n), C.byref(rr), C.byref(ch), C.byref(op), C.byref(cells)))
        self.cells_filled = int(cells.value)
        out = []
        for r in range(n):
            chains = []
            for k in range(rr[r].n_chains):
                c = ch[rr[r].chains_begin + k]
                a = Alignment()
                for f in Alignment.__slots__[:-1]:
                    setattr(a, f, int(getattr(c, f)))
                if c.ops_len:
                    addr = C.addressof(op.contents) + 8 * int(c.ops_begin)
                    ops = np.frombuffer((C.c_uint64 * int(c.ops_len)).from_address(addr), dtype=np.uint64)
                else:
                    ops = np.zeros(0, dtype=np.uint64)
                kind = (ops & 0xFF).astype(np.int64); contig = ((ops >> 16) & 0xFFFF).astype(np.int64); arg = (ops >> 32).astype(np.int64)
                a.operations = [(int(k_), int(c_), int(a_)) if k_ == 6 else (int(k_), int(a_) if k_ in (4, 5, 7) else 0, 0)
                                for k_, c_, a_ in zip(kind, contig, arg)]
                chains.append(a)
            out.append((chains, int(rr[r].prealign_score) if rr[r].has_prealign else None))
        return out

    def align_packed_raw(self, bases, offsets):
        """Same call without building Python objects: returns (read_results, chains, ops) as numpy structured-array VIEWS
        of the library's result arena (valid until the next call).  This is what a compiled front end would consume."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        rr, ch, op = C.POINTER(_ReadResult)(), C.POINTER(_Chain)(), C.POINTER(_Op)()
        cells = C.c_uint64(0)
        _check(lib().stitch_align_batch(self.h, bases.ctypes.data_as(C.POINTER(C.c_uint8)), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        C.c_uint32(n), C.byref(rr), C.byref(ch), C.byref(op), C.byref(cells)))
        self.cells_filled = int(cells.value)
        if n == 0:
            return np.zeros(0, dtype=_ReadResult), np.zeros(0, dtype=_Chain), np.zeros(0, dtype=_Op)
        rr_a = np.ctypeslib.as_array(rr, shape=(n,))
        n_ch = int(rr_a["chains_begin"][-1] + rr_a["n_chains"][-1])
        ch_a = np.ctypeslib.as_array(ch, shape=(n_ch,)) if n_ch else np.zeros(0, dtype=_Chain)
        n_op = int(ch_a["ops_begin"][-1] + ch_a["ops_len"][-1]) if n_ch else 0
        op_a = np.ctypeslib.as_array(op, shape=(n_op,)) if n_op else np.zeros(0, dtype=_Op)
        return rr_a, ch_a, op_a

    def align_one(self, read):
        return self.align([read])[0]

    def format_sam(self, read_idx, head, bases, quals=None):
        """SamRecordFormatter::format (mod.rs:622-973) for read `read_idx` of the last batch -> list of SAM lines."""
        b = bases.encode() if isinstance(bases, str) else bytes(bases)
        q = None if quals is None else (quals.encode() if isinstance(quals, str) else bytes(quals))
        cap = 4096 + 64 * len(b)
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().stitch_format_sam(self.h, C.c_uint32(read_idx), head.encode(), b, q, C.c_size_t(len(b)), buf, C.c_size_t(cap))
            if n < 0:
                raise StitchError(lib().stitch_last_error().decode())
            if n < cap:
                return buf.value.decode().split("\n")
            cap = n + 1

    def timing(self):
        t = _Timing()
        _check(lib().stitch_last_timing(self.h, C.byref(t), C.c_size_t(C.sizeof(t))))
        return {f: getattr(t, f) for f, _ in _Timing._fields_}
