#!/usr/bin/env python3
"""Diagnostic: in-kernel shader clock of xcorr_mfma_kernel (s_memtime / s_memrealtime stamps per wave).
Run with XENG_DBG_STAMPS=1 (set below).  Not a benchmark: the stamped build path is for shares/clocks only."""
import ctypes, os, sys, time
import numpy as np
os.environ["XENG_DBG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa
from caltech_bifrost_dsp_amd import ffi
NS, NC, NT, G = 352, 96, 480, 5
ffi.call("xengXgpuConfigure", NS, 2, NC, NT, G)
ffi.call("xengXgpuInitialize", 0)
gb = NT * NC * NS * 2
ring = ffi.DeviceBuffer(G * gb)
rs = np.random.RandomState(1)
for g in range(G):
    ring.upload(rs.randint(0, 255, size=gb, dtype=np.uint8), offset=g * gb)
outs = [ffi.DeviceBuffer(2 * NC * 249216 * 4) for _ in range(2)]
L = ffi.lib()
t0 = time.time()
n = 0
while time.time() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 2.0:     # >= 2 s of back-to-back launches
    for g in range(G):
        L.xengXgpuKernelAsync(ring.ptr + g * gb, outs[n & 1].ptr, int(g == G - 1))
    n += 1
    L.xengXgpuSyncLag(1)
L.xengXgpuSync()
L.xengXgpuDebugReadStamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
nw = ctypes.c_int()
L.xengXgpuDebugReadStamps(None, 0, ctypes.byref(nw))
st = np.zeros(nw.value * 8, dtype=np.uint64)
L.xengXgpuDebugReadStamps(st.ctypes.data, st.size, None)
st = st.reshape(-1, 8).astype(np.float64)
ok = st[:, 1] > 0
clk = st[ok, 0] / st[ok, 1] * 100e6
print("integrations run: %d" % n)
print("waves stamped: %d; in-kernel clock GHz: median %.3f  p10 %.3f  p90 %.3f" % (ok.sum(), np.median(clk) / 1e9, np.percentile(clk, 10) / 1e9, np.percentile(clk, 90) / 1e9))
print("loop cycles per wave: median %.0f (pure MFMA = 75*16*32 = 38400); loop time us: median %.2f" % (np.median(st[ok, 0]), np.median(st[ok, 1]) / 100.0))
span = (st[ok, 3].max() - st[ok, 2].min()) / 100.0
print("first loop start -> last loop end: %.1f us; sum of loop time / (1024 SIMDs * span) = %.3f" % (span, st[ok, 1].sum() / 100.0 / (1024 * span)))
pro = (st[ok, 2] - st[ok, 4]) / 100.0
epi = (st[ok, 5] - st[ok, 3]) / 100.0
tot = (st[ok, 5] - st[ok, 4]) / 100.0
print("per wave us: prologue median %.2f p90 %.2f | loop %.2f | epilogue median %.2f p90 %.2f | total %.2f" % (
    np.median(pro), np.percentile(pro, 90), np.median(st[ok, 1]) / 100.0, np.median(epi), np.percentile(epi, 90), np.median(tot)))
# per work item (channel, tile group) = 4 consecutive stamp rows: first wave entry -> last wave exit
it = st.reshape(-1, 4, 8)
it_start = np.where(it[:, :, 4] > 0, it[:, :, 4], np.inf).min(axis=1)      # (waves without a tile leave no stamps)
it_end = it[:, :, 5].max(axis=1)
dur = (it_end - it_start) / 100.0
print("per item us (entry of first wave -> exit of last): median %.2f mean %.2f p10 %.2f p90 %.2f max %.2f; sum/256 CUs = %.1f us" % (
    np.median(dur), dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90), dur.max(), dur.sum() / 256))
skew = (it[:, :, 4].max(axis=1) - it_start) / 100.0
print("entry skew between the waves of an item us: median %.2f p90 %.2f" % (np.median(skew), np.percentile(skew, 90)))
kspan = (st[ok, 5].max() - st[ok, 4].min()) / 100.0
print("kernel entry->exit span %.1f us; wave-resident fraction %.3f" % (kspan, tot.sum() / (1024 * kspan)))
