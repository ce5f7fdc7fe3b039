#!/usr/bin/env python3
"""Round 5: does the number of visibility buffers a streaming caller rotates through matter?  bench.py's C-ABI loop alternates two
(382 MB: partly resident in the 256 MB Infinity Cache); the Corr block's output ring rotates through 13-21 spans of 191 MB.
usage: out_buffers_probe.py [rounds] [integrations]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NCHAN, NT, G = 352, 96, 480, 5
NINPUT = NSTAND * 2
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
L = ffi.lib()
gulp = NT * NCHAN * NINPUT
ring = ffi.DeviceBuffer(10 * gulp)
ring.upload(np.random.RandomState(1).randint(0, 255, size=10 * gulp, dtype=np.uint8))
matlen = NCHAN * 249216
outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(16)]


def leg(nout):
    gi = 0
    for it in range(100 + nint):
        if it == 100:
            ffi.call("xengXgpuSync")
            t0 = time.perf_counter()
        o = outs[it % nout]
        for g in range(G):
            ffi.check("k", L.xengXgpuKernelAsync(ring.ptr + (gi % 10) * gulp, o.ptr, int(g == G - 1)))
            gi += 1
        ffi.call("xengXgpuSyncLag", 1)
    ffi.call("xengXgpuSync")
    return (time.perf_counter() - t0) / nint * 1e3


res = {}
for r in range(rounds):
    for nout in (2, 3, 4, 8, 16):
        res.setdefault(nout, []).append(leg(nout))
for nout, v in res.items():
    print("%2d output buffers: %s ms per integration" % (nout, " ".join("%.4f" % x for x in v)))
