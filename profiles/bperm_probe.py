#!/usr/bin/env python3
"""Diagnostic: do cross-lane reads in a small kernel stay correct while the X-engine runs on the same CUs?
Needs a scratch library that contains csrc/diag_probe.hip (see profiles/bperm_probe.sh); XENG_LIB points to it."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa: F401,E402
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

L = ffi.lib()
probe = L.xengDiagBpermProbe
probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.restype = ctypes.c_int
NSTAND, NCHAN = 352, 96
gb = 480 * NCHAN * 704
matlen = NCHAN * 249216
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, 480, 5)
ffi.call("xengXgpuInitialize", 0)
ring = ffi.DeviceBuffer(5 * gb)
ring.upload(np.random.RandomState(1).randint(0, 255, size=5 * gb, dtype=np.uint8))
outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]


def run_probe(mode, label, nrep):
    tot = np.zeros(8, dtype=np.uint64)
    first = 0
    for _ in range(nrep):
        h = np.zeros(8, dtype=np.uint64)
        ffi.check("probe", probe(mode, 3000 if mode < 2 else 600, 1024, h.ctypes.data))
        tot[:6] += h[:6]
        first = first or int(h[6])
    checked = nrep * (3000 if mode < 2 else 600) * 1024 * 256 * 4
    print("%-46s %d wrong of %.2e reads; by position in the group of four %s; lanes 48-63: %d; first: got %08x want %08x" % (
        label, int(tot[0]), checked, [int(v) for v in tot[1:5]], int(tot[5]), first >> 32, first & 0xFFFFFFFF), flush=True)


import threading
import time

stop = threading.Event()
count = [0]


def feeder():                                      # keeps the contraction running (lag-1 streaming) until told to stop
    n = 0
    while not stop.is_set():
        for g in range(5):
            ffi.check("k", L.xengXgpuKernelAsync(ring.ptr + g * gb, outs[n & 1].ptr, int(g == 4)))
        ffi.call("xengXgpuSyncLag", 1)
        n += 1
    ffi.call("xengXgpuSync")
    count[0] = n


for mode, name in ((0, "ds_bpermute_b32"), (1, "DPP row_shl/row_shr"), (2, "packed-fp32 sums -> ds_bpermute_b32"), (3, "packed-fp32 sums -> DPP")):
    run_probe(mode, name + ", GPU otherwise idle:", 2)
    stop.clear()
    th = threading.Thread(target=feeder)
    th.start()
    time.sleep(0.05)
    t0 = time.perf_counter()
    run_probe(mode, name + ", beside the X-engine:", 4)
    el = time.perf_counter() - t0
    stop.set()
    th.join()
    print("   (%d contractions ran during %.2f s of probing)" % (count[0], el), flush=True)
