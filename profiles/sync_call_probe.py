#!/usr/bin/env python3
"""Diagnostic: the synchronous drop-in call (xengXgpuKernel per gulp, reference semantics) at config 2: ms per integration
and the host time of a non-dump call (copy + wait) and of the dump call.  XENG_LIB selects the build.
usage: sync_call_probe.py [nrep]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NPOL, NCHAN, NT, ACC = 352, 2, 96, 480, 2400
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 300
gps = ACC // NT
gulp = NT * NCHAN * NSTAND * NPOL
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, NPOL, NCHAN, NT, gps)
ffi.call("xengXgpuInitialize", 0)
ring = ffi.DeviceBuffer(10 * gulp)
rs = np.random.RandomState(0xdeadbeef)
for g in range(10):
    ring.upload(rs.randint(0, 255, size=gulp, dtype=np.uint8), offset=g * gulp)
outs = [ffi.DeviceBuffer(2 * NCHAN * 249216 * 4) for _ in range(2)]
L = ffi.lib()
gi = 0
tn, td = [], []
for k in range(60 + nrep):
    if k == 60:
        t0 = time.perf_counter()
        tn, td = [], []
    for g in range(gps):
        t = time.perf_counter()
        ffi.check("xengXgpuKernel", L.xengXgpuKernel(ring.ptr + (gi % 10) * gulp, outs[k & 1].ptr, int(g == gps - 1)))
        (td if g == gps - 1 else tn).append(time.perf_counter() - t)
        gi += 1
el = time.perf_counter() - t0
print("%s: %.4f ms per integration; non-dump call %.1f us (median), dump call %.1f us (median)" % (
    os.path.basename(ffi.LIB_PATH), el / nrep * 1e3, np.median(tn) * 1e6, np.median(td) * 1e6), flush=True)
