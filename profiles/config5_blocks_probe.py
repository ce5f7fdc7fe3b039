#!/usr/bin/env python3
"""Diagnostic: bench.py's config5_blocks leg (config 5 through the Python blocks on one GPU) under several interpreter
switch intervals, several repeats each, in one process.  usage: config5_blocks_probe.py [repeats] [interval ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
intervals = [float(v) for v in sys.argv[2:]] or [5e-5, 5e-3, 5e-4]
ffi.call("xengSetDevice", 0)
gulp_bytes = bench.NTIME_GULP * bench.NCHAN * bench.NINPUT
ring_gulps = 10
ring = ffi.DeviceBuffer(ring_gulps * gulp_bytes)
rs = np.random.RandomState(0xdeadbeef)
for g in range(ring_gulps):
    ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
# allocation churn: device / pinned allocations and frees made while the legs run (the rings are meant to recycle)
counts = {"alloc": 0, "free": 0, "alloc_bytes": 0}
_init, _free = ffi.DeviceBuffer.__init__, ffi.DeviceBuffer.free


def _count_init(self, nbytes, space=ffi.SPACE_CUDA):
    counts["alloc"] += 1
    counts["alloc_bytes"] += int(nbytes)
    _init(self, nbytes, space)


def _count_free(self):
    if self.ptr:
        counts["free"] += 1
    _free(self)


ffi.DeviceBuffer.__init__, ffi.DeviceBuffer.free = _count_init, _count_free
real_set = sys.setswitchinterval
for rep in range(reps):
    for iv in intervals:
        real_set(iv)
        sys.setswitchinterval = lambda v: None          # the leg's own setting is ignored: this probe chooses
        try:
            r = bench.config5_blocks_leg(ffi, ring, gulp_bytes, ring_gulps, 0, long_len=int(os.environ.get("XENG_PROBE_LONG_LEN", "50")))
        finally:
            sys.setswitchinterval = real_set
        print("switch interval %.0e s: %.4f ms per integration (%s Gb/s), fused %s; allocations %d (%.0f MB), frees %d in this leg" % (
            iv, r["ms_per_integration"] or -1, r["value"], r["corracc_fused_into_dumps"], counts["alloc"], counts["alloc_bytes"] / 1e6, counts["free"]), flush=True)
        counts.update(alloc=0, free=0, alloc_bytes=0)
