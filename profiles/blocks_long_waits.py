#!/usr/bin/env python3
"""Diagnostic (round 4): which thread sits where when the GPU runs dry?  bench.py's config5_blocks leg with every call that can
wait wrapped; calls longer than a threshold are listed in time order (thread, call, start, duration), with the commits of the
visibility spans as a clock.  usage: blocks_long_waits.py [nint] [nwarm] [threshold us]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import _xfast, backend, ffi  # noqa: E402
from caltech_bifrost_dsp_amd.blocks import corr_acc_block  # noqa: E402

nint = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nwarm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
thr = (float(sys.argv[3]) if len(sys.argv) > 3 else 800.0) * 1e-6
log = []
pc = time.perf_counter


def timed(owner, name, label):
    f = getattr(owner, name)

    def w(*a, **k):
        t = pc()
        try:
            return f(*a, **k)
        finally:
            d = pc() - t
            if d > thr:
                log.append((t, d, threading.current_thread().name, label))
    setattr(owner, name, w)


for n in ("ring_acquire", "ring_acquire_parts", "ring_reserve", "ring_next_sequence", "ring_commit_external", "ring_commit",
          "beam_run", "beam_run_parts", "beam_integrate", "beam_mark", "xgpu_kernel_async", "xgpu_kernel_async_acc"):
    timed(_xfast, n, n)
for n in ("beam_wait", "xgpu_sync_lag", "map_sync", "beam_sync", "xgpu_sync", "copy_async", "copy_wait", "map_add_i32"):
    timed(backend.HipBackend, n, n)
timed(corr_acc_block.CorrAcc, "plan_dump", "CorrAcc.plan_dump (Corr's thread)")
timed(corr_acc_block.CorrAcc, "_next_plan", "CorrAcc._next_plan")
timed(threading.Thread, "join", "Thread.join")
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", bench.NSTAND, bench.NPOL, bench.NCHAN, bench.NTIME_GULP, bench.ACC_LEN // bench.NTIME_GULP)
ffi.call("xengXgpuInitialize", 0)
gulp_bytes = bench.NTIME_GULP * bench.NCHAN * bench.NINPUT
ring = ffi.DeviceBuffer(10 * gulp_bytes)
ring.upload(np.random.RandomState(1).randint(0, 256, size=10 * gulp_bytes, dtype=np.uint8))
t_begin = pc()
SLABS = bool(os.environ.get("REPEAT_SLABS"))          # the leg on a ring of packet slabs
for n in ("xgpu_kernel_slab", "beam_run_slabs"):
    timed(_xfast, n, n)
res = bench.config5_blocks_leg(ffi, ring, gulp_bytes, 10, 0, nint=nint, nwarm=nwarm, in_ring_integrations=int(os.environ.get("IN_RING", "4")),
                               **(dict(long_len=40, from_slabs=True) if SLABS else {}))
print("%.4f ms per integration, windows %s" % (res["ms_per_integration"], res["window_ms"]))
t_warm = t_begin + 0.0
log.sort()
# only the measured part: after the warm-up (the leg's own clock is not exposed: the last nint integrations by time)
t_end = max(t + d for t, d, _, _ in log) if log else pc()
t_lo = t_end - nint * res["ms_per_integration"] * 1e-3
print("calls longer than %.0f us during the measured part (time in ms from its start):" % (thr * 1e6))
for t, d, th, label in log:
    if t + d >= t_lo:
        print("  t %8.2f ms  %7.2f ms  %-22s %s" % ((t - t_lo) * 1e3, d * 1e3, th, label))
