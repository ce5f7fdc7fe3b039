import ctypes, os, sys
import numpy as np
os.environ["XENG_BEAM_STAMPS"] = "1"
sys.path.insert(0, "/root/repo")
import caltech_bifrost_dsp_amd
from caltech_bifrost_dsp_amd import ffi
NT, NC, NI, NB = 960, 96, 704, 32
ffi.call("xengBeamformInitialize", 0, NI, NC, NT, NB, 0)
rng = np.random.default_rng(0)
din = ffi.DeviceBuffer(NT * NC * NI).upload(rng.integers(0, 256, NT * NC * NI, dtype=np.uint8))
w = (rng.uniform(-17, 17, NC * NB * NI) + 1j * rng.uniform(-17, 17, NC * NB * NI)).astype(np.complex64)
dw = ffi.DeviceBuffer(w.nbytes).upload(w)
dout = ffi.DeviceBuffer(NC * NB * NT * 8)
L = ffi.lib()
L.xengBeamformDebugReadStamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for nlaunch in (5, 300, 3000, 20000):
    for k in range(nlaunch):
        L.xengBeamformRunVersioned(din.ptr, dout.ptr, dw.ptr, 1)
        if k % 64 == 63: ffi.call("xengBeamformSync")
    ffi.call("xengBeamformSync")
    st = np.zeros(8 * NC * 4 * 4, dtype=np.uint64)
    assert L.xengBeamformDebugReadStamps(st.ctypes.data, st.size) == 0
    st = st.reshape(-1, 4).astype(np.float64)
    loop_us = (st[:, 2] - st[:, 1]) / 100.0
    clk = st[:, 3] / loop_us / 1e3      # cycles per us -> GHz
    print("after %5d back-to-back launches: chunk loop %.2f us (median), shader clock in the loop %.3f GHz (median; p10 %.3f p90 %.3f)" % (
        nlaunch, np.median(loop_us), np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)), flush=True)
