"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/liboracle.so (the line-by-line CPU restatement of the reference aligner, see
oracle/stitch_oracle.hpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package `stitch_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
MIN_SCORE = -858_993_459
MODES = {"local": 0, "query-local": 1, "querylocal": 1, "target-local": 2, "targetlocal": 2, "global": 3, "custom": 4}
OP_NAMES = ["Match", "Subst", "Del", "Ins", "Xclip", "Yclip", "Xjump", "Yjump"]


def build(force=False):
    """Compiles the oracle with g++ (oracle/Makefile)."""
    srcs = [os.path.join(HERE, f) for f in ("stitch_oracle.cpp", "oracle_capi.cpp", "stitch_oracle.hpp")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _declare(L):
    L.orc_last_error.restype = C.c_char_p
    for name in ("orc_single", "orc_mc_custom", "orc_mc_traceback_all", "orc_mc_traceback_from", "orc_chain",
                 "orc_aln_cigar", "orc_aln_split_at_y", "orc_aln_earliest_x", "orc_aln_latest_x", "orc_al_align",
                 "orc_al_chain", "orc_al_format_sam"):
        getattr(L, name).restype = C.c_long
    L.orc_mc_new.restype = C.c_void_p
    L.orc_al_new.restype = C.c_void_p
    L.orc_al_cells.restype = C.c_uint64
    L.orc_bench.restype = C.c_double
    L.orc_bench_warm.restype = C.c_double
    return L


_native = None
NATIVE_FLAGS = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-fwrapv", "-shared", "-pthread"]


def native_lib():
    """The same sources built `-O3 -march=native` ON THE MACHINE THAT RUNS THE BENCH, into a temporary directory:
    the portable liboracle.so travels with the repo to a host whose CPU this container does not know, a native build must not.
    bench.py's cpu_baseline leg times this one; returns None when the host has no g++ (the portable build is timed then)."""
    global _native
    if _native is None:
        import tempfile
        d = tempfile.mkdtemp(prefix="stitch_oracle_native_")
        out = os.path.join(d, "liboracle_native.so")
        srcs = [os.path.join(HERE, f) for f in ("stitch_oracle.cpp", "oracle_capi.cpp", "prealign_oracle.cpp")]
        try:
            subprocess.check_call([os.environ.get("CXX", "g++")] + NATIVE_FLAGS + ["-o", out] + srcs, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _native = _declare(C.CDLL(out))
        except (OSError, subprocess.CalledProcessError):
            _native = False
    return _native or None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_last_error.restype = C.c_char_p
        for name in ("orc_single", "orc_mc_custom", "orc_mc_traceback_all", "orc_mc_traceback_from", "orc_chain",
                     "orc_aln_cigar", "orc_aln_split_at_y", "orc_aln_earliest_x", "orc_aln_latest_x", "orc_al_align",
                     "orc_al_chain", "orc_al_format_sam"):
            getattr(_lib, name).restype = C.c_long
        _lib.orc_mc_new.restype = C.c_void_p
        _lib.orc_al_new.restype = C.c_void_p
        _lib.orc_al_cells.restype = C.c_uint64
        _lib.orc_bench.restype = C.c_double
        _lib.orc_bench_warm.restype = C.c_double
    return _lib


def _err():
    return lib().orc_last_error().decode()


def scoring_array(match=1, mismatch=-1, gap_open=-5, gap_extend=-1, jump=-10, jump_same=None, jump_opp=None,
                  jump_inter=None, xclip_prefix=MIN_SCORE, xclip_suffix=MIN_SCORE, yclip_prefix=MIN_SCORE,
                  yclip_suffix=MIN_SCORE):
    js = jump if jump_same is None else jump_same
    jo = jump if jump_opp is None else jump_opp
    ji = jump if jump_inter is None else jump_inter
    return (C.c_int32 * 11)(match, mismatch, gap_open, gap_extend, js, jo, ji, xclip_prefix, xclip_suffix,
                            yclip_prefix, yclip_suffix)


class Alignment:
    """Mirror of align/alignment.rs:16-51."""

    FIELDS = ("score", "xstart", "xend", "ystart", "yend", "xlen", "ylen", "start_contig_idx", "end_contig_idx",
              "length", "mode")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw.get(f, 0))
        self.ops = [tuple(o) for o in kw.get("ops", [])]   # (kind, a, b)

    @classmethod
    def from_wire(cls, buf):
        a = cls(**{f: int(buf[k]) for k, f in enumerate(cls.FIELDS)})
        n = int(buf[11])
        a.ops = [(int(buf[12 + 3 * k]), int(buf[13 + 3 * k]), int(buf[14 + 3 * k])) for k in range(n)]
        return a

    def to_wire(self):
        mode = MODES[self.mode] if isinstance(self.mode, str) else self.mode
        vals = [getattr(self, f) for f in self.FIELDS[:-1]] + [mode, len(self.ops)]
        for o in self.ops:
            vals.extend(o)
        return (C.c_int64 * len(vals))(*vals)

    def cigar(self):
        w = self.to_wire()
        buf = C.create_string_buffer(16 * (len(self.ops) + 4))
        n = lib().orc_aln_cigar(w, buf, len(buf))
        assert n < len(buf)
        return buf.value.decode()

    def split_at_y(self, y_pivot):
        return _call_aln(lambda out, cap: lib().orc_aln_split_at_y(self.to_wire(), C.c_size_t(y_pivot), out, cap))

    def validate(self):
        return bool(lib().orc_aln_validate(self.to_wire()))

    def earliest_x_base_for(self, contig):
        r = lib().orc_aln_earliest_x(self.to_wire(), C.c_size_t(contig))
        return None if r < 0 else r

    def latest_x_base_for(self, contig):
        r = lib().orc_aln_latest_x(self.to_wire(), C.c_size_t(contig))
        return None if r < 0 else r

    def key(self):
        """Everything that must be bit-identical between oracle and product."""
        return (self.score, self.xstart, self.xend, self.ystart, self.yend, self.xlen, self.ylen,
                self.start_contig_idx, self.end_contig_idx, self.length, tuple(self.ops))

    def __repr__(self):
        return (f"contig-idx: {self.start_contig_idx}-{self.end_contig_idx} x-span: {self.xstart}-{self.xend}/{self.xlen} "
                f"y-span: {self.ystart}-{self.yend}/{self.ylen} score: {self.score} cigar: {self.cigar()} aln-len: {self.length}")


def _call_aln(fn, cap=1 << 16):
    while True:
        out = (C.c_int64 * cap)()
        need = fn(out, C.c_size_t(cap))
        if need < 0:
            raise RuntimeError(_err())
        if need == 0:
            return None
        if need <= cap:
            return Alignment.from_wire(out)
        cap = need


def _u8(b):
    if isinstance(b, str):
        b = b.encode()
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0"), len(b)


def single_align(mode, x, y, scoring=None, circular=False):
    """SingleContigAligner::{local,querylocal,targetlocal,global} (single_contig_aligner.rs:733-872)."""
    sc = scoring if scoring is not None else scoring_array()
    xb, m = _u8(x)
    yb, n = _u8(y)
    md = MODES[mode] if isinstance(mode, str) else mode
    return _call_aln(lambda out, cap: lib().orc_single(md, sc, int(circular), xb, C.c_size_t(m), yb, C.c_size_t(n), out, cap),
                     cap=64 + 3 * (2 * (m + n) + 16))


class MultiContigAligner:
    """multi_contig_aligner.rs:54-388."""

    def __init__(self):
        self.h = C.c_void_p(lib().orc_mc_new())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mc_free(self.h)
            self.h = None

    def add_contig(self, name, is_forward, seq, circular, scoring):
        sb, m = _u8(seq)
        if lib().orc_mc_add(self.h, name.encode(), int(is_forward), sb, C.c_size_t(m), int(circular), scoring) != 0:
            raise RuntimeError(_err())

    def set_scoring(self, scoring):
        lib().orc_mc_set_scoring(self.h, scoring)

    def custom(self, y, subset=None):
        yb, n = _u8(y)
        if subset is None:
            sub, ns = None, 0
        else:
            sub, ns = (C.c_uint32 * len(subset))(*subset), len(subset)
        return _call_aln(lambda out, cap: lib().orc_mc_custom(self.h, yb, C.c_size_t(n), sub, C.c_size_t(ns), out, cap),
                         cap=64 + 3 * (4 * n + 4096))

    def traceback_all(self, n, subset=None):
        if subset is None:
            sub, ns = None, 0
        else:
            sub, ns = (C.c_uint32 * len(subset))(*subset), len(subset)
        k = lib().orc_mc_traceback_all(self.h, C.c_size_t(n), sub, C.c_size_t(ns))
        if k < 0:
            raise RuntimeError(_err())
        return [_call_aln(lambda out, cap, q=q: lib().orc_chain(C.c_size_t(q), out, cap)) for q in range(k)]

    def traceback_from(self, n, contig_index):
        return _call_aln(lambda out, cap: lib().orc_mc_traceback_from(self.h, C.c_size_t(n), C.c_size_t(contig_index), out, cap))


def options_arrays(mode="local", match=1, mismatch=-4, gap_open=-6, gap_extend=-2, jump_score=-10, jump_same=None,
                   jump_opp=None, jump_inter=None, double_strand=False, circular=False, circular_slop=20,
                   suboptimal=False, suboptimal_pct=20.0, soft_clip=False, use_eq_and_x=False, pick_primary=0,
                   filter_secondary=False, filter_secondary_pct=10.0, pre_align=False, pre_align_min_score=100,
                   pre_align_subset_contigs=True, kmer_size=12, band_width=50):
    """aligners/mod.rs:65-116 (defaults are the CLI's)."""
    o = (C.c_int32 * 24)()
    vals = [MODES[mode] if isinstance(mode, str) else mode, match, mismatch, gap_open, gap_extend, jump_score,
            int(jump_same is not None), jump_same or 0, int(jump_opp is not None), jump_opp or 0,
            int(jump_inter is not None), jump_inter or 0, int(double_strand), int(circular), circular_slop,
            int(suboptimal), int(soft_clip), int(use_eq_and_x), pick_primary, int(filter_secondary)]
    vals += [int(pre_align), pre_align_min_score, int(pre_align_subset_contigs), kmer_size]
    for k, v in enumerate(vals):
        o[k] = v
    f = (C.c_float * 3)(suboptimal_pct, filter_secondary_pct, float(band_width))
    return o, f


def _targets(targets):
    names = (C.c_char_p * len(targets))(*[t[0].encode() for t in targets])
    bufs = [(C.c_uint8 * len(t[1])).from_buffer_copy(t[1].encode() if isinstance(t[1], str) else bytes(t[1])) for t in targets]
    seqs = (C.POINTER(C.c_uint8) * len(targets))(*[C.cast(b, C.POINTER(C.c_uint8)) for b in bufs])
    lens = (C.c_size_t * len(targets))(*[len(t[1]) for t in targets])
    return names, seqs, lens, bufs


class Aligners:
    """aligners/mod.rs:227-553 — `targets` is a list of (name, sequence)."""

    def __init__(self, targets, **opts):
        self.o, self.f = options_arrays(**opts)
        names, seqs, lens, self._keep = _targets(targets)
        self.h = C.c_void_p(lib().orc_al_new(self.o, self.f, C.c_size_t(len(targets)), names, seqs, lens))
        if not self.h:
            raise RuntimeError(_err())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_al_free(self.h)
            self.h = None

    def align(self, read):
        rb, n = _u8(read)
        k = lib().orc_al_align(self.h, rb, C.c_size_t(n))
        if k < 0:
            raise RuntimeError(_err())
        return [_call_aln(lambda out, cap, q=q: lib().orc_al_chain(self.h, C.c_size_t(q), out, cap), cap=64 + 3 * (4 * n + 4096))
                for q in range(k)]

    def cells(self):
        return int(lib().orc_al_cells(self.h))

    def set_chains(self, chains):
        """Test hook: replace the chains format_sam works on with caller-supplied Alignments."""
        for k, a in enumerate(chains):
            lib().orc_al_set_chain(self.h, C.c_size_t(k), a.to_wire())

    def prealign_score(self):
        """The pre-alignment score of the last align() (None without --pre-align or when nothing passed)."""
        v = C.c_int32(0)
        return int(v.value) if lib().orc_al_prealign(self.h, C.byref(v)) else None

    def format_sam(self, head, bases, quals=None, prealign=None):
        """SamRecordFormatter::format on the chains of the last align() call -> list of SAM text lines."""
        bb, n = _u8(bases)
        qb = _u8(quals)[0] if quals is not None else None
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            r = lib().orc_al_format_sam(self.h, head.encode(), bb, C.c_size_t(n), qb, int(prealign is not None),
                                        C.c_int32(prealign or 0), buf, C.c_size_t(cap))
            if r < 0:
                raise RuntimeError(_err())
            if r < cap:
                return buf.value.decode().split("\n")
            cap = r + 1


def cpu_bench(targets, reads, threads=1, **opts):
    """bench.py cpu_baseline leg: (seconds, cells, scores) for `reads` (list of bytes) on `threads` threads."""
    o, f = options_arrays(**opts)
    names, seqs, lens, keep = _targets(targets)
    cat = b"".join(reads)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    rb = (C.c_uint8 * max(1, len(cat))).from_buffer_copy(cat or b"\0")
    cells = C.c_uint64(0)
    scores = np.zeros(len(reads), dtype=np.int64)
    secs = lib().orc_bench(o, f, C.c_size_t(len(targets)), names, seqs, lens, rb,
                           offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(len(reads)), int(threads),
                           C.byref(cells), scores.ctypes.data_as(C.POINTER(C.c_int64)))
    return secs, int(cells.value), scores


def cpu_bench_sam(targets, reads, threads=1, name_base=0, chunk=1, warm=1, native=False, **opts):
    """bench.py cpu_baseline leg with the reference's worker model (`threads` workers, one aligner set each, `chunk` records
    per pull), WARM: every worker builds its aligner set and aligns `warm` read(s) before the clock starts, and the clock
    stops before anything is freed.  Returns (seconds, cells, scores, [SAM text per read], [busy seconds per worker]) — records
    of read r are named read_%07d % (name_base + r)."""
    o, f = options_arrays(**opts)
    names, seqs, lens, keep = _targets(targets)
    cat = b"".join(reads)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    rb = (C.c_uint8 * max(1, len(cat))).from_buffer_copy(cat or b"\0")
    cells = C.c_uint64(0)
    scores = np.zeros(len(reads), dtype=np.int64)
    busy = np.zeros(max(1, int(threads)), dtype=np.float64)
    cap = (4 << 20) * max(1, len(reads)) + 64 * len(cat)          # (a chimeric read yields a record per segment, each with SEQ, QUAL and SA)
    buf = C.create_string_buffer(cap)
    soffs = np.zeros(len(reads) + 1, dtype=np.uint64)
    fn = ((native and native_lib()) or lib()).orc_bench_warm      # (native: the -march=native build of the same sources)
    fn.restype = C.c_double
    secs = fn(o, f, C.c_size_t(len(targets)), names, seqs, lens, rb,
              offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(len(reads)), int(threads), int(chunk), int(warm),
              C.byref(cells), scores.ctypes.data_as(C.POINTER(C.c_int64)), busy.ctypes.data_as(C.POINTER(C.c_double)),
              C.c_size_t(name_base), buf, C.c_size_t(cap), soffs.ctypes.data_as(C.POINTER(C.c_uint64)))
    if int(soffs[-1]) > cap:
        raise RuntimeError("SAM buffer too small")
    raw = buf.raw
    sam = [raw[int(soffs[k]):int(soffs[k + 1])].decode() for k in range(len(reads))]
    return secs, int(cells.value), scores, sam, [float(b) for b in busy]
