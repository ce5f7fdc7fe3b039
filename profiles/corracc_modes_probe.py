#!/usr/bin/env python3
"""Round 5: three ways of doing CorrAcc's long accumulation inside the concurrent config-5 pattern (C ABI, one GPU), interleaved
on one box:
   map     one 574 MB "a += b" map kernel per dump (the reference's order, corr_acc_block.py:298-306)
   fused   the dump's own epilogue read-modify-writes the accumulator (xengXgpuKernelAsyncAcc, round 3)
   group   the dumps of a group of K stay in their spans and are summed in ONE pass (xengMapSumI32, round 5)
   none    no long accumulation at all (the floor: contraction + beamformer chain)
Per 2400-sample integration: 5 gulps registered in place + 1 contraction, 2.5 beamformer gulps of 960 samples + power sums.
usage: corracc_modes_probe.py [rounds] [integrations per leg] [K] [modes, comma-separated: none,map,fused,group]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 120
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
NSTAND, NCHAN, NT, G, NB, NS = 352, 96, 480, 5, 32, 24
NINPUT = 2 * NSTAND
ffi.call("xengSetDevice", 0)
L = ffi.lib()
gulp = NT * NCHAN * NINPUT
RING = 10
rs = np.random.RandomState(0xdeadbeef)
ring = ffi.DeviceBuffer(RING * gulp).upload(rs.randint(0, 255, size=RING * gulp, dtype=np.uint8))
matbytes = NCHAN * ((NSTAND // 2 + 1) * (NSTAND // 4) * 16) * 2 * 4
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
P = K + 3
outs = [ffi.DeviceBuffer(matbytes) for _ in range(P)]
accs = [ffi.DeviceBuffer(matbytes) for _ in range(2)]
ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, 2 * NT, NB, 0)
w = (rs.uniform(-1, 1, (NCHAN, NB, NINPUT)) + 1j * rs.uniform(-1, 1, (NCHAN, NB, NINPUT))).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dbeam = ffi.DeviceBuffer(NCHAN * NB * 2 * NT * 8)
dpow = ffi.DeviceBuffer((NB // 2) * (2 * NT // NS) * NCHAN * 16)
gi = [0]
bi = [0]
SrcArray = ctypes.c_void_p * K


def bstep():
    src = ring.ptr + ((2 * bi[0]) % (RING - 1)) * gulp
    ffi.check("run", L.xengBeamformRunVersioned(src, dbeam.ptr, dw.ptr, 1))
    ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
    bi[0] += 1


def step(mode, n):
    o = outs[n % P]
    for g in range(G):
        dump = int(g == G - 1)
        src = ring.ptr + (gi[0] % RING) * gulp
        if mode == "fused":
            ffi.check("k", L.xengXgpuKernelAsyncAcc(src, o.ptr, dump, accs[n & 1].ptr, 1 if n < 2 else 2))
        else:
            ffi.check("k", L.xengXgpuKernelAsync(src, o.ptr, dump))
        gi[0] += 1
    for _ in range(2 + (n & 1)):
        bstep()
    ffi.call("xengXgpuSyncLag", 1)          # dump n-1 is complete
    if mode == "map" and n >= 1:
        ffi.call("xengMapSync")
        ffi.check("map", (L.xengMapAssignI32 if n == 1 else L.xengMapAddI32)(accs[0].ptr, outs[(n - 1) % P].ptr, matbytes // 4))
    if mode == "group":
        if n >= K and n % K == 0:           # dumps n-K .. n-1 are complete: one pass over their spans
            srcs = SrcArray(*[outs[(n - K + j) % P].ptr for j in range(K)])
            ffi.check("sum", L.xengMapSumI32(accs[0].ptr, srcs, K, matbytes // 4, int(n > K)))
        elif n % K == 1:
            ffi.call("xengMapSync")         # ... and have been read before dump n + 2 writes the first of them again


def leg(mode, n0, nwarm=12):
    for n in range(n0, n0 + nwarm):
        step(mode, n)
    ffi.call("xengDeviceSynchronize")
    t0 = time.perf_counter()
    for n in range(n0 + nwarm, n0 + nwarm + nint):
        step(mode, n)
    ffi.call("xengDeviceSynchronize")
    return (time.perf_counter() - t0) / nint * 1e3


modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["none", "map", "fused", "group"]
res = {m: [] for m in modes}
n0 = 0
for r in range(rounds):
    for m in modes:
        n0 = ((n0 + K - 1) // K) * K
        ms = leg(m, n0)
        n0 += 12 + nint
        res[m].append(ms)
    print("round %d  " % r + "  ".join("%s=%.4f" % (m, res[m][-1]) for m in modes), flush=True)
print("# ms per integration, config-5 pattern through the C ABI, K = %d dumps per group; median of %d interleaved rounds x %d integrations" % (K, rounds, nint))
for m in modes:
    v = sorted(res[m])
    print("%-6s median %.4f  min %.4f  max %.4f" % (m, v[len(v) // 2], v[0], v[-1]))
