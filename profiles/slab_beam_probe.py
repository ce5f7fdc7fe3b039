#!/usr/bin/env python3
"""Diagnostic (round 4): the beamformer pair (Beamform 960 samples + power sums, config 4 shape) on plain gulps, on packed packet
slabs (stride 6176), on slabs whose payloads start on 128-byte lines (stride 6272) and on slabs with two packets out of order (the
slow path: both parts scattered by one work-group each), and (round 5) on slabs in arrival order with 1 % of the packets lost
(everything behind a loss one slot early, the slab shorter), alone on the GPU.  XENG_SLAB_TABLES=1 / 0: through the packet indices /
by strides + scatter.
usage: slab_beam_probe.py [rounds] [gulps per round]"""
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NCHAN, NT, NB, NS = 352, 96, 480, 32, 24
NINPUT = NSTAND * 2
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ngulp = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ffi.call("xengSetDevice", 0)
ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, 2 * NT, NB, 0)
L = ffi.lib()
npb = NSTAND // 32
stride = 32 + NCHAN * 64
npk = NT * npb
rs = np.random.RandomState(5)
STRIDE_A, LEAD_A = 49 * 128, 96
slabs, slabs_a, slabs_irr = [], [], []
for k in range(10):
    slab = np.zeros((npk, stride), dtype=np.uint8)
    i = 0
    for t in range(NT):
        for pb in range(npb):
            slab[i, :32] = np.frombuffer(struct.pack(">QLHHHHLLL", k * NT + t, 0, 64, NINPUT, NCHAN, NCHAN, 0, 0, pb * 64), dtype=np.uint8)
            i += 1
    slab[:, 32:] = rs.randint(0, 256, size=(npk, stride - 32), dtype=np.uint8)
    slabs.append(ffi.DeviceBuffer(slab.nbytes).upload(slab))
    swapped = slab.copy()
    swapped[[0, 1]] = swapped[[1, 0]]                      # two packets out of order: the gulp takes the scatter
    slabs_irr.append(ffi.DeviceBuffer(slab.nbytes).upload(swapped))
    a = np.zeros(LEAD_A + npk * STRIDE_A, dtype=np.uint8)
    a[LEAD_A:].reshape(npk, STRIDE_A)[:, :stride] = slab
    slabs_a.append(ffi.DeviceBuffer(a.nbytes).upload(a))
slabs_lossy = []
for k, b in enumerate(slabs):
    c = ffi.DeviceBuffer(npk * stride)
    lost = sorted(int(p) for p in rs.choice(npk - 1, size=npk // 100, replace=False))
    dst = src = 0
    for p in lost + [npk]:
        if p > src:
            ffi.call("xengMemcpy", c.ptr + dst * stride, b.ptr + src * stride, (p - src) * stride)
            dst += p - src
        src = p + 1
    slabs_lossy.append((c, dst))
gulp = NT * NCHAN * NINPUT
ring = ffi.DeviceBuffer(10 * gulp)
for k in range(10):
    ffi.check("u", L.xengSnap2UnpackAsync(slabs[k].ptr, npk, stride, ring.ptr + k * gulp, k * NT, NT, 0, NCHAN, NINPUT, 1))
ffi.call("xengDeviceSynchronize")
w = (rs.uniform(-17, 17, (NCHAN, NB, NINPUT)) + 1j * rs.uniform(-17, 17, (NCHAN, NB, NINPUT))).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dbeam = ffi.DeviceBuffer(NCHAN * NB * 2 * NT * 8)
dpow = ffi.DeviceBuffer((NB // 2) * (2 * NT // NS) * NCHAN * 16)


def run(mode, n):
    for it in range(n + 20):
        if it == 20:
            ffi.call("xengBeamformSync")
            t0 = time.perf_counter()
        k0 = (2 * it) % 10
        if mode == "plain":
            ffi.check("r", L.xengBeamformRunVersioned(ring.ptr + k0 * gulp, dbeam.ptr, dw.ptr, 1))
        elif mode == "irregular":
            ffi.check("r", L.xengBeamformRunSlabs(slabs_irr[k0].ptr, npk, NT, slabs_irr[k0 + 1].ptr, npk, stride, k0 * NT, 0, dbeam.ptr, dw.ptr, 1))
        elif mode == "lossy":
            (b0, n0), (b1, n1) = slabs_lossy[k0], slabs_lossy[k0 + 1]
            ffi.check("r", L.xengBeamformRunSlabs(b0.ptr, n0, NT, b1.ptr, n1, stride, k0 * NT, 0, dbeam.ptr, dw.ptr, 1))
        elif mode == "packed":
            ffi.check("r", L.xengBeamformRunSlabs(slabs[k0].ptr, npk, NT, slabs[k0 + 1].ptr, npk, stride, k0 * NT, 0, dbeam.ptr, dw.ptr, 1))
        else:
            ffi.check("r", L.xengBeamformRunSlabs(slabs_a[k0].ptr + LEAD_A, npk, NT, slabs_a[k0 + 1].ptr + LEAD_A, npk, STRIDE_A, k0 * NT, 0, dbeam.ptr, dw.ptr, 1))
        ffi.check("i", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
    ffi.call("xengBeamformSync")
    return (time.perf_counter() - t0) / n * 1e6


res = {}
for r in range(rounds):
    for mode in ("plain", "packed", "aligned", "lossy", "irregular"):
        res.setdefault(mode, []).append(run(mode, ngulp if mode != "irregular" else max(10, ngulp // 10)))
for mode, v in res.items():
    v = sorted(v)
    print("%-9s median %.1f us per 960-sample gulp (Run + Integrate), min %.1f max %.1f" % (mode, v[len(v) // 2], v[0], v[-1]))
import ctypes
nfb = ctypes.c_int()
ffi.call("xengBeamformGetSlabFallbacks", ctypes.byref(nfb))
print("parts scattered after all:", nfb.value)
