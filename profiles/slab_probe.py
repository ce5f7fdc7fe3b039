#!/usr/bin/env python3
"""Diagnostic (round 4): packets -> visibilities with the slabs read in place (xengXgpuKernelAsyncSlab), config 2, streaming;
with a -DXENG_DIAGNOSTICS build (XENG_LIB=profiles/_ab/libxeng_diag.so) also the A/B of the channel order of the slab launches:
neighbouring channels per XCD and round (default) against the replay order (XENG_SLAB_PLAIN_ORDER=1), interleaved in one process.
usage: slab_probe.py [rounds] [integrations] [sync|stream] [variant numbers, e.g. 1,3]
(sync: wait after every integration -- kernel durations without overlap;  variant 3 hands over slabs with one packet too many,
which take the scatter into the scratch gulps: the descriptor kernel on the plain gulp layout;  variant 4 reads slabs whose
payloads start on 128-byte lines: stride 6272, packets handed over at base + 96)"""
import ctypes
import os
import struct
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
SYNC = len(sys.argv) > 3 and sys.argv[3] == "sync"
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
L = ffi.lib()
nspp = 32
npb = NSTAND // nspp
stride = 32 + NCHAN * nspp * 2
npk = NT * npb
slab = np.zeros((npk + 1, stride), dtype=np.uint8)
k = 0
for t in range(NT):
    for pb in range(npb):
        slab[k, :32] = np.frombuffer(struct.pack(">QLHHHHLLL", t, 0, nspp * 2, NINPUT, NCHAN, NCHAN, 0, 0, pb * nspp * 2), dtype=np.uint8)
        k += 1
slab[npk] = slab[npk - 1]
slab[:, 32:] = np.random.RandomState(5).randint(0, 255, size=(npk + 1, stride - 32), dtype=np.uint8)
slabs = [ffi.DeviceBuffer(slab.nbytes).upload(slab) for _ in range(2 * G)]
# the same packets as a receiver could place them: payloads on 128-byte lines (49 lines per packet, header in the last 32 bytes of
# the line before)
STRIDE_A, LEAD_A = 49 * 128, 96
slab_a = np.zeros(LEAD_A + (npk + 1) * STRIDE_A, dtype=np.uint8)
slab_a[LEAD_A:].reshape(npk + 1, STRIDE_A)[:, :stride] = slab
slabs_a = [ffi.DeviceBuffer(slab_a.nbytes).upload(slab_a) for _ in range(2 * G)]
gulp = NT * NCHAN * NINPUT
ring = ffi.DeviceBuffer(2 * G * gulp)
matlen = NCHAN * 249216
outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]


def run(mode, n):
    kk = 0
    for it in range(n + 100):
        if it == 100:
            ffi.call("xengXgpuSync")
            t0 = time.perf_counter()
        for g in range(G):
            s = kk % (2 * G)
            if mode == "scatter":
                dst = ring.ptr + s * gulp
                ffi.check("u", L.xengSnap2UnpackAsync(slabs[s].ptr, npk, stride, dst, 0, NT, 0, NCHAN, NINPUT, 1))
                ffi.check("k", L.xengXgpuKernelAsync(dst, outs[it & 1].ptr, int(g == G - 1)))
            elif mode == "aligned":
                ffi.check("s", L.xengXgpuKernelAsyncSlab(slabs_a[s].ptr + LEAD_A, npk, STRIDE_A, 0, 0, outs[it & 1].ptr, int(g == G - 1), None, 0))
            else:
                ffi.check("s", L.xengXgpuKernelAsyncSlab(slabs[s].ptr, npk + (mode == "irregular"), stride, 0, 0, outs[it & 1].ptr, int(g == G - 1), None, 0))
            kk += 1
        ffi.call("xengXgpuSync" if SYNC else "xengXgpuSyncLag", *(() if SYNC else (1,)))
    ffi.call("xengXgpuSync")
    return (time.perf_counter() - t0) / n * 1e3


res = {}
variants = [("scatter", None), ("in place, neighbouring channels per XCD", None), ("in place, replay channel order", "1"),
            ("irregular", None), ("aligned", None), ("aligned, no clear/scatter launches", "scatter"), ("aligned, no helper launches", "all")]
if len(sys.argv) > 4:
    variants = [variants[int(i)] for i in sys.argv[4].split(",")]
for r in range(rounds):
    for name, env in variants:
        if env:
            os.environ["XENG_SLAB_PLAIN_ORDER" if env == "1" else "XENG_SLAB_SKIP"] = env
        ms = run(name if name in ("scatter", "irregular") else "aligned" if name.startswith("aligned") else "slab", nint)
        os.environ.pop("XENG_SLAB_PLAIN_ORDER", None)
        os.environ.pop("XENG_SLAB_SKIP", None)
        res.setdefault(name, []).append(ms)
        print("round %d %-44s %.4f ms per integration" % (r, name, ms), flush=True)
for name, v in res.items():
    v = sorted(v)
    print("%-44s median %.4f min %.4f max %.4f ms per integration = %.0f Gb/s" % (name, v[len(v) // 2], v[0], v[-1], 8 * NINPUT * 2400 * NCHAN / (v[len(v) // 2] * 1e-3) / 1e9))
nfb = ctypes.c_int()
ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(nfb))
print("gulps that took the scatter after all:", nfb.value, "(the replay-order rows only differ on a -DXENG_DIAGNOSTICS build)")
