#!/usr/bin/env python3
"""Diagnostic timing of the beamformer path at BASELINE config 4 (704 inputs, 96 chan, 32 beams, 960 samples)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa
from caltech_bifrost_dsp_amd import ffi
NT, NC, NI, NB, NS = 960, 96, 704, 32, 24
ffi.call("xengBeamformInitialize", 0, NI, NC, NT, NB, 0)
rng = np.random.default_rng(0)
din = ffi.DeviceBuffer(NT * NC * NI).upload(rng.integers(0, 256, NT * NC * NI, dtype=np.uint8))
w = (rng.uniform(-17, 17, NC * NB * NI) + 1j * rng.uniform(-17, 17, NC * NB * NI)).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dout = ffi.DeviceBuffer(NC * NB * NT * 8)
dpow = ffi.DeviceBuffer((NB // 2) * (NT // NS) * NC * 16)
L = ffi.lib()
for _ in range(5):
    L.xengBeamformRun(din.ptr, dout.ptr, dw.ptr); L.xengBeamformIntegrate(dout.ptr, dpow.ptr, NS)
ffi.call("xengBeamformSync")
ffi.call("xengBeamformSetProfiling", 1)
tm = (ctypes.c_double * 2)(); cn = (ctypes.c_int * 2)()
ffi.call("xengBeamformGetTimes", tm, cn)
n = 30
t0 = time.perf_counter()
for _ in range(n):
    L.xengBeamformRun(din.ptr, dout.ptr, dw.ptr); L.xengBeamformIntegrate(dout.ptr, dpow.ptr, NS)
ffi.call("xengBeamformSync")
el = time.perf_counter() - t0
ffi.call("xengBeamformGetTimes", tm, cn)
run_us, int_us = tm[0] / cn[0] * 1e3, tm[1] / cn[1] * 1e3
flop = NT * NC * NB * NI * 8
print("Run %.1f us (%.1f TFLOP/s fp32-equivalent, %.1f%% of 157.3), Integrate %.1f us (%.0f GB/s), wall/iter %.1f us, ingest %.0f Gb/s" % (
    run_us, flop / run_us / 1e6, 100 * flop / run_us / 1e6 / 157.3, int_us, NC * NB * NT * 8 / int_us / 1e3, el / n * 1e6, 8 * NT * NC * NI / (el / n) / 1e9))
