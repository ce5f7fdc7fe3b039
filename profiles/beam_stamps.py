#!/usr/bin/env python3
"""Diagnostic: per-wave timeline of beamform_i8x3_kernel (s_memrealtime stamps; XENG_BEAM_STAMPS=1)."""
import ctypes, os, sys
import numpy as np
os.environ["XENG_BEAM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa
from caltech_bifrost_dsp_amd import ffi
NT, NC, NI, NB = 960, 96, 704, 32
ffi.call("xengBeamformInitialize", 0, NI, NC, NT, NB, 0)
rng = np.random.default_rng(0)
din = ffi.DeviceBuffer(NT * NC * NI).upload(rng.integers(0, 256, NT * NC * NI, dtype=np.uint8))
w = (rng.uniform(-17, 17, NC * NB * NI) + 1j * rng.uniform(-17, 17, NC * NB * NI)).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dout = ffi.DeviceBuffer(NC * NB * NT * 8)
L = ffi.lib()
for _ in range(5):
    L.xengBeamformRunVersioned(din.ptr, dout.ptr, dw.ptr, 1)
ffi.call("xengBeamformSync")
nw = 8 * NC * 4
st = np.zeros(nw * 4, dtype=np.uint64)
L.xengBeamformDebugReadStamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.xengBeamformDebugReadStamps(st.ctypes.data, st.size) == 0
st = st.reshape(-1, 4).astype(np.float64) / 100.0      # us
t0 = st[:, 0].min()
print("waves %d; kernel span (first entry -> last exit) %.1f us" % (len(st), st[:, 3].max() - t0))
print("entry times us: p10 %.1f median %.1f p90 %.1f max %.1f" % tuple(np.percentile(st[:, 0] - t0, [10, 50, 90, 100])))
print("per wave us: entry->first chunk %.2f | chunk loop %.2f (%.3f per chunk) | epilogue %.2f | total %.2f" % (
    np.median(st[:, 1] - st[:, 0]), np.median(st[:, 2] - st[:, 1]), np.median(st[:, 2] - st[:, 1]) / 21, np.median(st[:, 3] - st[:, 2]),
    np.median(st[:, 3] - st[:, 0])))

tot = st[:, 3] - st[:, 0]
for name, v in (("entry->first chunk", st[:, 1] - st[:, 0]), ("chunk loop", st[:, 2] - st[:, 1]), ("epilogue", st[:, 3] - st[:, 2]), ("total", tot),
                ("exit time", st[:, 3] - t0)):
    print("%-20s p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % ((name,) + tuple(np.percentile(v, [1, 10, 50, 90, 99, 100]))))
wg = st.reshape(-1, 4, 4)                      # [work-group][wave][stamp]
xcd = np.arange(len(wg)) % 8
for x in range(8):
    sel = wg[xcd == x]
    print("XCD %d: median total %.2f us, last exit %.2f us" % (x, np.median(sel[:, :, 3] - sel[:, :, 0]), (sel[:, :, 3] - t0).max()))
