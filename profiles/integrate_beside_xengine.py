#!/usr/bin/env python3
"""Diagnostic for DESIGN.md 4.10: xengBeamformIntegrate repeated beside running contractions, every result compared
with the stand-alone one.  With a `-DINTEG_DIAG=<n>` build of beamform.hip (XENG_LIB) it exercises the ds_bpermute
forms of the reduction: 10 as shipped until round 2, 11 + s_nop before the shuffles, 12 single-FMA accumulation instead of
packed fp32, 13 32-bit offsets instead of 64-bit pointer increments, 14 full waits + s_nop after every shuffle group.
usage: integrate_beside_xengine.py [rounds]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa: F401,E402
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NSTAND, NINPUT, NCHAN, NT_B, NB, NS = 352, 704, 96, 960, 32, 24
L = ffi.lib()
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, 480, 5)
ffi.call("xengXgpuInitialize", 0)
matlen = NCHAN * 249216
gb = 480 * NCHAN * NINPUT
ring = ffi.DeviceBuffer(5 * gb)
ring.upload(np.random.RandomState(3).randint(0, 255, size=5 * gb, dtype=np.uint8))
outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]
ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, NT_B, NB, 0)
rng = np.random.default_rng(9)
beams = (rng.normal(0, 3e3, NCHAN * NB * NT_B) + 1j * rng.normal(0, 3e3, NCHAN * NB * NT_B)).astype(np.complex64)
dbeam = ffi.DeviceBuffer(beams.nbytes).upload(beams)
npw = (NB // 2) * (NT_B // NS) * NCHAN * 4
dpow = ffi.DeviceBuffer(npw * 4)
snaps = [ffi.DeviceBuffer(npw * 4) for _ in range(K)]
shape = (NB // 2, NT_B // NS, NCHAN, 4)
ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
ffi.call("xengBeamformSync")
ref = dpow.download(np.uint32).reshape(shape)

stop = threading.Event()


def feeder():
    n = 0
    while not stop.is_set():
        for g in range(5):
            ffi.check("k", L.xengXgpuKernelAsync(ring.ptr + g * gb, outs[n & 1].ptr, int(g == 4)))
        ffi.call("xengXgpuSyncLag", 1)
        n += 1
    ffi.call("xengXgpuSync")


for beside in (False, True):
    if beside:
        th = threading.Thread(target=feeder)
        th.start()
    for n in range(K):
        ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
        ffi.call("xengBeamformSync")
        ffi.check("map", L.xengMapAssignI32(snaps[n].ptr, dpow.ptr, npw))
        ffi.call("xengMapSync")
    if beside:
        stop.set()
        th.join()
    bad, words, comps, lanes = 0, 0, set(), set()
    for n in range(K):
        d = snaps[n].download(np.uint32).reshape(shape) != ref
        if d.any():
            bad += 1
            words += int(d.sum())
            comps |= set(np.unique(np.nonzero(d)[3]).tolist())
            lanes |= set((np.unique(np.nonzero(d)[1]) % 8).tolist())
    print("%-28s %d of %d results differ from the stand-alone one (%d words; components %s, time blocks mod 8 %s)" % (
        "beside the contraction:" if beside else "GPU otherwise idle:", bad, K, words, sorted(comps), sorted(lanes)), flush=True)
