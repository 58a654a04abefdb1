"""TEST INFRASTRUCTURE: ctypes front end of tests/emu/libemu.so (CPU emulation of the HIP fill kernel)."""
import ctypes as C
import os
import subprocess

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libemu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", HERE, "libemu.so"], stdout=subprocess.DEVNULL)
        _lib = C.CDLL(LIB)
        _lib.emu_ctx_new.restype = C.c_void_p
        _lib.emu_job.restype = C.c_long
    return _lib


class Emu:
    """contigs: list of (name, is_forward, seq); params: 12 ints (see emu_ctx_new)."""

    def __init__(self, params, contigs):
        self.C = len(contigs)
        names = (C.c_char_p * self.C)(*[c[0].encode() for c in contigs])
        fwd = (C.c_int32 * self.C)(*[int(c[1]) for c in contigs])
        self._bufs = [(C.c_uint8 * len(c[2])).from_buffer_copy(c[2].encode() if isinstance(c[2], str) else bytes(c[2])) for c in contigs]
        seqs = (C.POINTER(C.c_uint8) * self.C)(*[C.cast(b, C.POINTER(C.c_uint8)) for b in self._bufs])
        lens = (C.c_uint32 * self.C)(*[len(c[2]) for c in contigs])
        self.h = C.c_void_p(lib().emu_ctx_new((C.c_int32 * 12)(*params), self.C, names, fwd, seqs, lens))

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_ctx_free(self.h)
            self.h = None

    def job(self, y, act=None, mode=0, frm=0, local16=False):
        y = y.encode() if isinstance(y, str) else bytes(y)
        act = list(range(self.C)) if act is None else sorted(act)
        cap = 64 + (len(act) if mode in (1, 3) else 1) * (12 + 3 * ((len(y) + 1) * (max(len(b) for b in self._bufs) + 2) + 64))
        out = (C.c_int64 * cap)()
        used = C.c_size_t(0)
        n = lib().emu_job(self.h, (C.c_uint8 * len(y)).from_buffer_copy(y), len(y), (C.c_uint32 * len(act))(*act), len(act), mode, frm,
                          out, C.c_size_t(cap), C.byref(used), int(local16))
        if n < 0:
            raise RuntimeError(f"emu_job failed: {n}")
        chains, o = [], 0
        for _ in range(n):
            nops = int(out[o + 11])
            if out[o + 10] == -1:
                chains.append(None)
            else:
                chains.append(orc.Alignment.from_wire(out[o:o + 12 + 3 * nops]))
            o += 12 + 3 * nops
        return chains
