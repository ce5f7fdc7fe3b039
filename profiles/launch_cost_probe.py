#!/usr/bin/env python3
"""Diagnostic: host time of the library's enqueue-only calls from one thread (no other Python threads): what a block
thread pays per gulp before any interpreter-lock effect.  usage: launch_cost_probe.py [ntime]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NT = int(sys.argv[1]) if len(sys.argv) > 1 else 480
NC, NI, NB, NS = 96, 704, 32, 24
ffi.call("xengSetDevice", 0)
ffi.call("xengBeamformInitialize", 0, NI, NC, NT, NB, 0)
rng = np.random.default_rng(0)
din = ffi.DeviceBuffer(NT * NC * NI).upload(rng.integers(0, 256, NT * NC * NI, dtype=np.uint8))
w = (rng.uniform(-17, 17, NC * NB * NI) + 1j * rng.uniform(-17, 17, NC * NB * NI)).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dout = ffi.DeviceBuffer(NC * NB * NT * 8)
dpow = ffi.DeviceBuffer((NB // 2) * (NT // NS) * NC * 16)
L = ffi.lib()
tk = ctypes.c_ulonglong()


def timeit(name, fn, n=300, depth=4, sync=lambda: ffi.call("xengBeamformSync")):
    for _ in range(20):
        fn()
    sync()
    per = []
    t_all = time.perf_counter()
    for k in range(n):
        t = time.perf_counter()
        fn()
        per.append(time.perf_counter() - t)
        if depth and k % depth == depth - 1:
            sync()
    sync()
    el = time.perf_counter() - t_all
    per = np.array(per) * 1e6
    print("%-44s host us per call: median %6.1f  mean %6.1f  p90 %6.1f   (wall per call incl. a sync every %d: %.1f us)" % (
        name, np.median(per), per.mean(), np.percentile(per, 90), depth, el / n * 1e6), flush=True)


timeit("xengBeamformRunVersioned (%d samples)" % NT, lambda: L.xengBeamformRunVersioned(din.ptr, dout.ptr, dw.ptr, 1))
timeit("xengBeamformIntegrate", lambda: L.xengBeamformIntegrate(dout.ptr, dpow.ptr, NS))
timeit("xengBeamformMark", lambda: L.xengBeamformMark(ctypes.byref(tk)))
timeit("Run + Mark", lambda: (L.xengBeamformRunVersioned(din.ptr, dout.ptr, dw.ptr, 1), L.xengBeamformMark(ctypes.byref(tk))))
timeit("Run, sync after every call", lambda: L.xengBeamformRunVersioned(din.ptr, dout.ptr, dw.ptr, 1), depth=1)
timeit("Run, 16 in flight", lambda: L.xengBeamformRunVersioned(din.ptr, dout.ptr, dw.ptr, 1), depth=16)
# CorrAcc's map on its own stream
n32 = 96 * 249216 * 2
a, b = ffi.DeviceBuffer(n32 * 4), ffi.DeviceBuffer(n32 * 4)
timeit("xengMapAddI32 (574 MB)", lambda: L.xengMapAddI32(a.ptr, b.ptr, n32), n=60, sync=lambda: ffi.call("xengMapSync"))
