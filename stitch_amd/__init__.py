"""stitch_amd — MI355X-native (gfx950) implementation of the `stitch align` hot path of fulcrumgenomics/stitch:
the jump-aware affine-gap DP, its per-column jump reduce and the traceback, as hand-written HIP kernels behind the
C ABI in include/stitch_gpu.h.  See DESIGN.md."""
from .api import Aligners, Alignment, Builder, Index, StitchError, TargetSeq, lib  # noqa: F401
