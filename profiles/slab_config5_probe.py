#!/usr/bin/env python3
"""Diagnostic (round 4): config 5 (Corr with fused CorrAcc + Beamform + power sums, concurrent) with each consumer fed either
from plain gulps or from packet slabs read in place -- which of the two pays what when both run together.
usage: slab_config5_probe.py [rounds] [integrations] [aligned]"""
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NCHAN, NT, G, NB, NS = 352, 96, 480, 5, 32, 24
NINPUT = NSTAND * 2
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ALIGNED = len(sys.argv) > 3 and sys.argv[3] == "aligned"
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, 2 * NT, NB, 0)
L = ffi.lib()
npb = NSTAND // 32
pstride = 32 + NCHAN * 64
npk = NT * npb
rs = np.random.RandomState(5)
stride, lead = (49 * 128, 96) if ALIGNED else (pstride, 0)
slabs = []
POOL, pool = bool(os.environ.get("PROBE_POOL")), None      # all slabs in ONE allocation (as bench.py's blocks-from-slabs leg holds them)
NSL = int(os.environ.get("PROBE_NSLABS", "10"))      # distinct slabs (and plain gulps) in the replay: 10 = 326 MB, more than fits in the caches beyond ~8
for k in range(NSL):
    slab = np.zeros((npk, pstride), dtype=np.uint8)
    i = 0
    for t in range(NT):
        for pb in range(npb):
            slab[i, :32] = np.frombuffer(struct.pack(">QLHHHHLLL", k * NT + t, 0, 64, NINPUT, NCHAN, NCHAN, 0, 0, pb * 64), dtype=np.uint8)
            i += 1
    slab[:, 32:] = rs.randint(0, 256, size=(npk, pstride - 32), dtype=np.uint8)
    a = np.zeros(lead + npk * stride, dtype=np.uint8)
    a[lead:].reshape(npk, stride)[:, :pstride] = slab
    if POOL:
        if pool is None:
            pool = ffi.DeviceBuffer(NSL * a.nbytes)

        class _W:                                  # (a window on the pool with the attributes the loop uses)
            pass
        w_ = _W()
        w_.ptr = pool.ptr + k * a.nbytes
        pool.upload(a, offset=k * a.nbytes)
        slabs.append(w_)
    else:
        slabs.append(ffi.DeviceBuffer(a.nbytes).upload(a))
gulp = NT * NCHAN * NINPUT
ring = ffi.DeviceBuffer(NSL * gulp)
for k in range(NSL):
    ffi.check("u", L.xengSnap2UnpackAsync(slabs[k].ptr + lead, npk, stride, ring.ptr + k * gulp, k * NT, NT, 0, NCHAN, NINPUT, 1))
ffi.call("xengDeviceSynchronize")
w = (rs.uniform(-17, 17, (NCHAN, NB, NINPUT)) + 1j * rs.uniform(-17, 17, (NCHAN, NB, NINPUT))).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dbeam = ffi.DeviceBuffer(NCHAN * NB * 2 * NT * 8)
dpow = ffi.DeviceBuffer((NB // 2) * (2 * NT // NS) * NCHAN * 16)
matlen = NCHAN * 249216
outs3 = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(3)]
acc_pair = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]


def run(corr, beam, n):
    gi = bi = 0
    for it in range(n + 12):
        if it == 12:
            ffi.call("xengDeviceSynchronize")
            t0 = time.perf_counter()
        o = outs3[it % 3]
        if corr != "none":
            for g in range(G):
                s = gi % NSL
                if corr == "slab":
                    ffi.check("s", L.xengXgpuKernelAsyncSlab(slabs[s].ptr + lead, npk, stride, s * NT, 0, o.ptr, int(g == G - 1), acc_pair[it & 1].ptr, 1 if it < 2 else 2))
                else:
                    ffi.check("k", L.xengXgpuKernelAsyncAcc(ring.ptr + s * gulp, o.ptr, int(g == G - 1), acc_pair[it & 1].ptr, 1 if it < 2 else 2))
                gi += 1
        if beam != "none":
            for _ in range(2 + (it & 1)):
                k0 = (2 * bi) % NSL
                if beam == "slab":
                    ffi.check("r", L.xengBeamformRunSlabs(slabs[k0].ptr + lead, npk, NT, slabs[k0 + 1].ptr + lead, npk, stride, k0 * NT, 0, dbeam.ptr, dw.ptr, 1))
                else:
                    ffi.check("r", L.xengBeamformRunVersioned(ring.ptr + k0 * gulp, dbeam.ptr, dw.ptr, 1))
                ffi.check("i", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
                bi += 1
        if corr != "none":
            ffi.call("xengXgpuSyncLag", 1)
        else:
            ffi.call("xengBeamformSync")
    ffi.call("xengDeviceSynchronize")
    return (time.perf_counter() - t0) / n * 1e3


res = {}
COMBOS = os.environ.get("PROBE_COMBOS")
combos = [("plain", "plain"), ("slab", "plain"), ("plain", "slab"), ("slab", "slab"), ("plain", "none"), ("slab", "none"), ("none", "plain"), ("none", "slab")]
if COMBOS:
    combos = [tuple(c.split("/")) for c in COMBOS.split(",")]
for r in range(rounds):
    for c in combos:
        res.setdefault(c, []).append(run(c[0], c[1], nint))
print("%d distinct slabs / gulps in the replay;" % NSL, "packet stride %d%s" % (stride, " (payloads on 128-byte lines)" if ALIGNED else " (packed)"))
for c, v in res.items():
    v = sorted(v)
    print("corr %-5s beam %-5s  median %.4f ms per integration (min %.4f max %.4f)" % (c[0], c[1], v[len(v) // 2], v[0], v[-1]))
