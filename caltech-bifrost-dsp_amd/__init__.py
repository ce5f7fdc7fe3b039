"""MI355X-native X-engine for the LWA-352 bifrost pipeline (hot path only).

Corr / CorrAcc / Beamform / BeamformSumBeams blocks with the reference's Block + ring
interface (blocks/), a thin ctypes binding (ffi.py) and hand-written HIP for gfx950
(csrc/).  No PyTorch, no CPU fallback.
"""
from . import ffi  # noqa: F401

__all__ = ["ffi"]
