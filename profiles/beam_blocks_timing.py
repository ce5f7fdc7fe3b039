#!/usr/bin/env python3
"""Diagnostic: where a Beamform / BeamformSumBeams block thread spends its wall time per gulp on the GPU box
(library calls and ring operations timed by wrappers).  usage: beam_blocks_timing.py [nint]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import blocks_probe as bp  # noqa: E402
from caltech_bifrost_dsp_amd import backend, ring  # noqa: E402
import numpy as np  # noqa: E402

acc = collections.defaultdict(lambda: [0.0, 0])


def timed(cls, name, label=None):
    f = getattr(cls, name)
    lab = label or (cls.__name__ + "." + name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[lab]
            e[0] += time.perf_counter() - t
            e[1] += 1
    setattr(cls, name, w)


for n in ("bfBeamformRun", "beam_mark", "beam_wait", "bfBeamformIntegrate", "beam_sync"):
    timed(backend.HipBackend, n)
timed(ring.Ring, "_wait_for_room")
timed(ring.Ring, "_alloc_span")
timed(ring.Ring, "_commit")
timed(ring.WriteSpan, "__init__", "WriteSpan()")


def main():
    nint = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    bp.ffi.call("xengSetDevice", 0)
    gulp_bytes = bp.NTIME_GULP * bp.NCHAN * bp.NINPUT
    bp.run.ring = bp.ffi.DeviceBuffer(10 * gulp_bytes)
    rs = np.random.RandomState(0xdeadbeef)
    for g in range(10):
        bp.run.ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
    for which in (["bf"], ["bf"], ["bf", "sb"], ["bf", "sb"]):
        acc.clear()
        bp.run(which, nint)
        for k, (t, n) in sorted(acc.items()):
            print("    %-32s %7d calls  %8.1f us per call" % (k, n, t / max(n, 1) * 1e6))


if __name__ == "__main__":
    main()
