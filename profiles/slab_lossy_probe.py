#!/usr/bin/env python3
"""Round 5: packets -> visibilities on regular and lossy slabs, config 2, streaming (enqueue integration n, wait for n-1), the
library chosen by XENG_LIB -- run once per build on the same box for an A/B of the offset-table path against the round-4 path
(regular: read in place by strides; any hole: zero-fill + scatter).
  plain      the same voltages as plain gulps (xengXgpuKernelAsync): the floor
  regular    every slab complete and in order
  one_lost   one packet lost per integration (its slot holds a copy of the next packet)
  slot_1pct  1 % of the packets of every gulp lost, slot model
  shift_1pct 1 % lost, arrival order: everything behind a loss one slot early, the slab shorter
usage: slab_lossy_probe.py [rounds] [integrations]"""
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
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
L = ffi.lib()
npb = NINPUT // 64
stride = 32 + NCHAN * 64
npk = NT * npb
rs = np.random.RandomState(0xdeadbeef)
gulp_bytes = NT * NCHAN * NINPUT
nslab = 2 * G
slabs, gulps = [], []
for k in range(nslab):
    v = rs.randint(0, 255, size=(NT, NCHAN, npb, 64), dtype=np.uint8)
    slab = np.zeros((npk, stride), dtype=np.uint8)
    slab[:, 32:] = v.transpose(0, 2, 1, 3).reshape(npk, NCHAN * 64)
    for t in range(NT):
        for b in range(npb):
            slab[t * npb + b, :32] = np.frombuffer(struct.pack(">QLHHHHLLL", k * NT + t, 0, 64, NINPUT, NCHAN, NCHAN, 0, 0, b * 64), dtype=np.uint8)
    slabs.append(ffi.DeviceBuffer(slab.nbytes).upload(slab))
    gulps.append(ffi.DeviceBuffer(gulp_bytes).upload(v.reshape(-1)))
matlen = NCHAN * 249216
outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]


def lossy(nlost, shift, seed=11):
    r = np.random.RandomState(seed)
    out = []
    for k, b in enumerate(slabs):
        c = ffi.DeviceBuffer(b.nbytes)
        lost = sorted(int(p) for p in r.choice(npk - 1, size=nlost(k), replace=False)) if nlost(k) else []
        if shift:
            dst = src = 0
            for p in lost + [npk]:
                if p > src:
                    ffi.call("xengMemcpy", c.ptr + dst * stride, b.ptr + src * stride, (p - src) * stride)
                    dst += p - src
                src = p + 1
            out.append((c, dst))
        else:
            ffi.call("xengMemcpy", c.ptr, b.ptr, b.nbytes)
            for p in lost:
                ffi.call("xengMemcpy", c.ptr + p * stride, b.ptr + (p + 1) * stride, stride)
            out.append((c, npk))
    return out


sets = {
    "regular": [(b, npk) for b in slabs],
    "one_lost": lossy(lambda k: 1 if k % G == 2 else 0, False),
    "slot_1pct": lossy(lambda k: npk // 100, False),
    "shift_1pct": lossy(lambda k: npk // 100, True),
}


def leg(name):
    kk = 0
    nwarm = 60
    for it in range(nwarm + nint):
        if it == nwarm:
            ffi.call("xengXgpuSync")
            t1 = time.perf_counter()
        for g in range(G):
            slot = kk % nslab
            if name == "plain":
                ffi.check("k", L.xengXgpuKernelAsync(gulps[slot].ptr, outs[it & 1].ptr, int(g == G - 1)))
            else:
                b, n = sets[name][slot]
                ffi.check("slab", L.xengXgpuKernelAsyncSlab(b.ptr, n, stride, slot * NT, 0, outs[it & 1].ptr, int(g == G - 1), None, 0))
            kk += 1
        ffi.call("xengXgpuSyncLag", 1)
    ffi.call("xengXgpuSync")
    return (time.perf_counter() - t1) / nint * 1e3


names = os.environ.get("PROBE_LEGS", "plain,regular,one_lost,slot_1pct,shift_1pct").split(",")
res = {n: [] for n in names}
for r in range(rounds):
    for n in names:
        res[n].append(leg(n))
nf = ctypes.c_int(-1)
ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(nf))
print("library %s%s: ms per integration, %d rounds x %d integrations; gulps scattered in all: %d" % (os.environ.get("XENG_LIB", "(shipped)"), " XENG_SLAB_TABLES=" + os.environ["XENG_SLAB_TABLES"] if "XENG_SLAB_TABLES" in os.environ else "", rounds, nint, nf.value))
for n in names:
    print("  %-11s %s   median %.4f  (%+.1f %% vs regular)" % (n, " ".join("%.4f" % x for x in res[n]), float(np.median(res[n])),
                                                            100 * (np.median(res[n]) / np.median(res["regular" if "regular" in res else names[0]]) - 1)))
