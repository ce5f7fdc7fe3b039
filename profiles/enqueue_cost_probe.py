#!/usr/bin/env python3
"""Diagnostic (round 4): host time of the enqueue-only calls of one config-5 integration, streaming (lag 1), per call kind.
usage: enqueue_cost_probe.py [integrations]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NCHAN, NT, G, NB, NS = 352, 96, 480, 5, 32, 24
NINPUT = NSTAND * 2
nint = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, 2 * NT, NB, 0)
L = ffi.lib()
gulp = NT * NCHAN * NINPUT
ring = ffi.DeviceBuffer(10 * gulp)
ring.upload(np.random.RandomState(1).randint(0, 256, size=10 * gulp, dtype=np.uint8))
rs = np.random.RandomState(2)
w = (rs.uniform(-17, 17, (NCHAN, NB, NINPUT)) + 1j * rs.uniform(-17, 17, (NCHAN, NB, NINPUT))).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dbeam = ffi.DeviceBuffer(NCHAN * NB * 2 * NT * 8)
dpow = ffi.DeviceBuffer((NB // 2) * (2 * NT // NS) * NCHAN * 16)
matlen = NCHAN * 249216
outs3 = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(3)]
acc_pair = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]
T = {}
pc = time.perf_counter


def timed(name, fn, *a):
    t0 = pc()
    rc = fn(*a)
    T.setdefault(name, []).append(pc() - t0)
    return rc


for mode in ("fused acc", "plain"):
    T.clear()
    gi = bi = 0
    for it in range(nint + 20):
        if it == 20:
            T.clear()
            t_all = pc()
        o = outs3[it % 3]
        for g in range(G):
            s = gi % 10
            last = g == G - 1
            if mode == "fused acc":
                ffi.check("k", timed("KernelAsyncAcc, dump" if last else "KernelAsyncAcc, gulp", L.xengXgpuKernelAsyncAcc, ring.ptr + s * gulp, o.ptr, int(last), acc_pair[it & 1].ptr, 1 if it < 2 else 2))
            else:
                ffi.check("k", timed("KernelAsync, dump" if last else "KernelAsync, gulp", L.xengXgpuKernelAsync, ring.ptr + s * gulp, o.ptr, int(last)))
            gi += 1
        for _ in range(2 + (it & 1)):
            k0 = (2 * bi) % 10
            ffi.check("r", timed("BeamformRunVersioned", L.xengBeamformRunVersioned, ring.ptr + k0 * gulp, dbeam.ptr, dw.ptr, 1))
            ffi.check("i", timed("BeamformIntegrate", L.xengBeamformIntegrate, dbeam.ptr, dpow.ptr, NS))
            bi += 1
        timed("XgpuSyncLag(1)", L.xengXgpuSyncLag, 1)
    ffi.call("xengDeviceSynchronize")
    el = (pc() - t_all) / nint
    print("%s: %.4f ms per integration" % (mode, el * 1e3))
    for k, v in sorted(T.items()):
        v = np.array(v) * 1e6
        print("   %-26s %6d calls  median %7.1f us  mean %7.1f  p95 %7.1f  max %8.1f" % (k, v.size, np.median(v), v.mean(), np.percentile(v, 95), v.max()))
