#!/usr/bin/env python3
"""Diagnostic: device time of the HBM-bound side kernels at config-2 / config-4 sizes (HIP events around back-to-back
launches through the C ABI).  usage: side_kernels.py map|integrate|unpack|packetize   (env switches as in DESIGN.md)"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa
from caltech_bifrost_dsp_amd import ffi

L = ffi.lib()
what = sys.argv[1] if len(sys.argv) > 1 else "map"
NCHAN, NINPUT = 96, 704
matlen = NCHAN * 249216


def timed(fn, sync, nrep=50, nwarm=5):
    for _ in range(nwarm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(nrep):
        fn()
    sync()
    return (time.perf_counter() - t0) / nrep


if what == "map":
    a, b = ffi.DeviceBuffer(2 * matlen * 4), ffi.DeviceBuffer(2 * matlen * 4)
    ffi.call("xengMemset", a.ptr, 1, a.nbytes)
    ffi.call("xengMemset", b.ptr, 2, b.nbytes)
    for name, fn, nb in (("add", L.xengMapAddI32, 3), ("assign", L.xengMapAssignI32, 2)):
        t = timed(lambda: fn(a.ptr, b.ptr, 2 * matlen), lambda: ffi.call("xengMapSync"))
        print("map %-6s XENG_MAP_VAR=%s XENG_MAP_BLOCKS=%s: %.1f us  %.2f TB/s" % (
            name, os.environ.get("XENG_MAP_VAR", "default"), os.environ.get("XENG_MAP_BLOCKS", "default"), t * 1e6, nb * a.nbytes / t / 1e12))
elif what == "integrate":
    NT, NB, NS = 960, 32, 24
    ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, NT, NB, 0)
    beams = ffi.DeviceBuffer(NCHAN * NB * NT * 8)
    ffi.call("xengMemset", beams.ptr, 0, beams.nbytes)
    pw = ffi.DeviceBuffer((NB // 2) * (NT // NS) * NCHAN * 16)
    t = timed(lambda: L.xengBeamformIntegrate(beams.ptr, pw.ptr, NS), lambda: ffi.call("xengBeamformSync"), nrep=200)
    print("integrate: %.1f us  %.2f TB/s of %d MB" % (t * 1e6, beams.nbytes / t / 1e12, beams.nbytes >> 20))
elif what == "packetize":
    nst = NINPUT // 2
    ffi.call("xengXgpuConfigure", nst, 2, NCHAN, 480, 5)
    ffi.call("xengXgpuInitialize", 0)
    vis = ffi.DeviceBuffer(2 * matlen * 4)
    ffi.call("xengMemset", vis.ptr, 3, vis.nbytes)
    a2i = np.arange(NINPUT, dtype=np.int32)
    blm = np.zeros(nst * nst * 4, dtype=np.int32)
    cjm = np.zeros_like(blm)
    ffi.call("xengXgpuGetOrder", a2i.ctypes.data, blm.ctypes.data, cjm.ctypes.data)
    dbl, dcj = ffi.DeviceBuffer(blm.nbytes).upload(blm), ffi.DeviceBuffer(cjm.nbytes).upload(cjm)
    nbl = nst * (nst + 1) // 2
    dpay = ffi.DeviceBuffer(nbl * 4 * NCHAN * 8)
    for fmt in (1, 0):
        t = timed(lambda: L.xengXgpuPacketize(vis.ptr, dpay.ptr, dbl.ptr, dcj.ptr, fmt), lambda: None, nrep=30)
        print("packetize fmt %d: %.1f us per synchronous call  %.2f TB/s (read + write %d MB)" % (fmt, t * 1e6, 2 * dpay.nbytes / t / 1e12, 2 * dpay.nbytes >> 20))
elif what == "unpack":
    # the synchronous xengSnap2Unpack call at config-2 size: 5280 packets (480 samples x 11 packets of 64 inputs x 96 channels)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import xeng_oracle as orc          # (test infrastructure: builds the packet slab)
    NT = 480
    rng = np.random.default_rng(3)
    vin = rng.integers(0, 256, (NT, NCHAN, NINPUT // 2, 2), dtype=np.uint8)
    pk = orc.snap2_packets(vin, seq0=10 ** 12, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=0)
    slab = np.frombuffer(b"".join(pk), dtype=np.uint8)
    dslab = ffi.DeviceBuffer(slab.nbytes).upload(slab)
    dg = ffi.DeviceBuffer(vin.nbytes)
    placed, dropped = ctypes.c_int(), ctypes.c_int()
    fn = lambda: L.xengSnap2Unpack(dslab.ptr, len(pk), len(pk[0]), dg.ptr, 10 ** 12, NT, 0, NCHAN, NINPUT, 1, ctypes.byref(placed), ctypes.byref(dropped))
    t = timed(fn, lambda: None, nrep=300, nwarm=20)
    assert placed.value == len(pk) and dropped.value == 0 and np.array_equal(dg.download(np.uint8), vin.reshape(-1))
    print("unpack: %.1f us per synchronous call, %d packets of %d B -> %.1f MB gulp (%.2f TB/s read + write)" % (
        t * 1e6, len(pk), len(pk[0]), vin.nbytes / 1e6, (slab.nbytes + vin.nbytes) / t / 1e12))
    # a lossy slab takes the second pass (zero-fill + scatter)
    fn2 = lambda: L.xengSnap2Unpack(dslab.ptr, len(pk) - 7, len(pk[0]), dg.ptr, 10 ** 12, NT, 0, NCHAN, NINPUT, 1, ctypes.byref(placed), ctypes.byref(dropped))
    t2 = timed(fn2, lambda: None, nrep=100, nwarm=5)
    print("unpack, 7 packets missing: %.1f us per synchronous call (coverage check on the device, then zero-fill + second scatter)" % (t2 * 1e6))
