"""gomokuai_amd -- MI355X-native implementation of the GomokuAI self-play hot path.

The compute path is libgomoku_hip.so (hand-written HIP kernels for gfx950 behind the C-ABI declared in
include/gomoku_hip.h).  This package holds the host-side mirror of the reference's interface for that
path; there is no CPU fallback: without the HIP library or a GPU the compute entries raise.
"""
from . import lib  # noqa: F401
