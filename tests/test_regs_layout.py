"""The lane-interleaved traceback layout of the register-resident kernel (fill_regs.hip), checked on the CPU.

walk_core.h's tb_row_offset is what the traceback walk uses to find a cell's byte; it divides by multiplication with precomputed
reciprocals.  Here it is compared, for every row of contigs of many lengths, with the mapping as fill_regs.hip documents it,
written out independently: the contig's ceil(m / 4) groups of four rows go to the 64 lanes in order, the first `grem` lanes hold
one group more, every lane's first row sits in the fullest lane's top register (a lane with a group less ends in register 4), and
register idx of lane l is byte ((idx >> 2) * 64 + l) * 4 + (idx & 3)."""
import ctypes

import pytest

from tests.emu import emu


def documented_offset(m, i):
    ngr = (m + 3) // 4
    gq, grem = divmod(ngr, 64)
    row = i - 1
    big, small = 4 * (gq + 1), 4 * gq
    if row < grem * big:
        lane, uu, nrows = row // big, row % big, big
    else:
        rr = row - grem * big
        lane, uu, nrows = grem + rr // small, rr % small, small
    gtop = gq + (1 if grem else 0)
    idx = 4 * gtop - 1 - uu                     # the lane's first row is in register 4 * gtop - 1, whatever the lane
    assert idx == nrows - 1 - uu + (4 if grem and lane >= grem else 0)
    return ((idx >> 2) * 64 + lane) * 4 + (idx & 3)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 63, 64, 255, 256, 257, 259, 260, 511, 512, 513, 1000, 1300, 4095, 4096, 4097, 5000, 5117, 5118, 5119, 5120])
def test_row_to_byte_mapping_is_the_documented_bijection(m):
    lib = emu.lib()
    lib.emu_tb_row_offset.restype = ctypes.c_uint32
    lib.emu_tb_row_offset.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    seen = set()
    blocks = (m + 255) // 256                   # a contig's part of a column: whole 256-row blocks
    for i in range(1, m + 1):
        off = lib.emu_tb_row_offset(m, i)
        assert off == documented_offset(m, i), (m, i)
        assert off < 256 * blocks and off not in seen, (m, i, off)
        seen.add(off)
