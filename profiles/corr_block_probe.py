#!/usr/bin/env python3
"""Round 5: bench.py's corr_block leg (Corr.main on device rings) and the C-ABI streaming loop in one process, for a kernel trace
(profiles/trace_gaps.py): where does the block's ~10 % over the loop go?
usage: corr_block_probe.py block|loop [integrations]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "block"
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", bench.NSTAND, bench.NPOL, bench.NCHAN, bench.NTIME_GULP, bench.ACC_LEN // bench.NTIME_GULP)
ffi.call("xengXgpuInitialize", 0)
gulp_bytes = bench.NTIME_GULP * bench.NCHAN * bench.NINPUT
ring = ffi.DeviceBuffer(10 * gulp_bytes)
ring.upload(np.random.RandomState(1).randint(0, 255, size=10 * gulp_bytes, dtype=np.uint8))
if mode == "block":
    r = bench.corr_block_leg(ffi, ring, gulp_bytes, 10, 0, nint=nint, nwarm=100)
    print("block: %.4f ms per integration" % r["ms_per_integration"])
else:
    L = ffi.lib()
    G = bench.ACC_LEN // bench.NTIME_GULP
    matlen = bench.NCHAN * 249216
    outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]
    gi = 0
    for it in range(100 + nint):
        if it == 100:
            ffi.call("xengXgpuSync")
            t0 = time.perf_counter()
        for g in range(G):
            ffi.check("k", L.xengXgpuKernelAsync(ring.ptr + (gi % 10) * gulp_bytes, outs[it & 1].ptr, int(g == G - 1)))
            gi += 1
        ffi.call("xengXgpuSyncLag", 1)
    ffi.call("xengXgpuSync")
    print("loop: %.4f ms per integration" % ((time.perf_counter() - t0) / nint * 1e3))
