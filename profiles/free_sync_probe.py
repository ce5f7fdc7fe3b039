#!/usr/bin/env python3
"""Diagnostic (round 4): does hipFree / hipHostFree wait for work queued on the library's streams (all created with
hipStreamNonBlocking)?  The ring used to rely on that ("hipFree synchronises the whole device") whenever a span buffer was
really freed.  Measured without touching freed memory: ~N ms of contractions are queued, then an UNRELATED small buffer is
freed and the call is timed; if the free returns long before the queue has drained, it does not wait for those streams.
usage: free_sync_probe.py [integrations]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NINT = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NSTAND, NCHAN, NT, G = 352, 96, 480, 5
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
gulp = NT * NCHAN * NSTAND * 2
vin = ffi.DeviceBuffer(G * gulp)
vin.upload(np.random.RandomState(1).randint(0, 255, size=G * gulp, dtype=np.uint8))
matlen = NCHAN * (NSTAND // 2 + 1) * (NSTAND // 4) * 16
outs = [ffi.DeviceBuffer(matlen * 8) for _ in range(2)]
L = ffi.lib()


def queue(n):
    for k in range(n):
        for g in range(G):
            ffi.call("xengXgpuKernelAsync", vin.ptr + g * gulp, outs[k & 1].ptr, int(g == G - 1))


def timed(fn):
    t0 = time.perf_counter()
    fn()
    return (time.perf_counter() - t0) * 1e3


queue(20)
ffi.call("xengXgpuSync")
t_q = timed(lambda: (queue(NINT), ffi.call("xengXgpuSync")))
print("%d integrations queued and drained: %.2f ms (%.3f ms each)" % (NINT, t_q, t_q / NINT))
for label, space in (("hipFree (device, 1 MB)", ffi.SPACE_CUDA), ("hipHostFree (pinned, 1 MB)", ffi.SPACE_CUDA_HOST),
                     ("hipFree (device, 64 MB)", ffi.SPACE_CUDA)):
    nb = (64 << 20) if "64" in label else (1 << 20)
    for rep in range(3):
        b = ffi.DeviceBuffer(nb, space)
        t_enq = timed(lambda: queue(NINT))
        t_free = timed(b.free)
        t_rest = timed(lambda: ffi.call("xengXgpuSync"))
        print("%-28s enqueue %.2f ms | free %.3f ms | rest of the queue %.2f ms -> the free %s" % (
            label, t_enq, t_free, t_rest, "WAITED for the streams" if t_rest < 0.2 * t_q else "did NOT wait for the non-blocking streams"))
# the same for an allocation (hipMalloc / hipHostMalloc): expected not to wait
for label, space in (("hipMalloc 1 MB", ffi.SPACE_CUDA), ("hipHostMalloc 1 MB", ffi.SPACE_CUDA_HOST)):
    queue(NINT)
    t0 = time.perf_counter()
    b = ffi.DeviceBuffer(1 << 20, space)
    t_a = (time.perf_counter() - t0) * 1e3
    t_rest = timed(lambda: ffi.call("xengXgpuSync"))
    print("%-28s alloc %.3f ms | rest of the queue %.2f ms" % (label, t_a, t_rest))
    b.free()
v = ctypes.c_char_p(L.xengVersion())
print("library:", v.value.decode())
ffi.call("xengXgpuDestroy")
