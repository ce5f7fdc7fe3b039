#!/usr/bin/env python3
"""Diagnostic (round 4): bench.py's config5_blocks leg repeated in one process -- how far do the runs differ on one box?
usage: config5_blocks_repeat.py [runs] [nint] [nwarm] [gc: default|freeze|off] [input ring in integrations, e.g. 4,8,16: one run each per round]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 400
nwarm = int(sys.argv[3]) if len(sys.argv) > 3 else 400
import gc
gcmode = sys.argv[4] if len(sys.argv) > 4 else "default"
PIN = os.environ.get("REPEAT_PIN", "none")      # none | node (the GPU's NUMA node, as bench.py pins) | N (the first N CPUs of that node)
if PIN != "none":
    from caltech_bifrost_dsp_amd import sharding
    import ctypes as _ct
    bus = _ct.create_string_buffer(64)
    ffi.call("xengSetDevice", 0)
    ffi.call("xengGetDevicePciBusId", 0, bus, 64)
    info = sharding.pin_rank(0, 1, bus.value.decode())
    if PIN != "node":
        cpus = sorted(os.sched_getaffinity(0))[:int(PIN)]
        os.sched_setaffinity(0, cpus)
    print("pinned (%s): %d CPUs, %s" % (PIN, len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4]), flush=True)
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", bench.NSTAND, bench.NPOL, bench.NCHAN, bench.NTIME_GULP, bench.ACC_LEN // bench.NTIME_GULP)
ffi.call("xengXgpuInitialize", 0)
gulp_bytes = bench.NTIME_GULP * bench.NCHAN * bench.NINPUT
ring = ffi.DeviceBuffer(10 * gulp_bytes)
ring.upload(np.random.RandomState(1).randint(0, 256, size=10 * gulp_bytes, dtype=np.uint8))
depths = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [4]
for r in range(runs * len(depths)):
    gc.collect()
    if gcmode == "freeze":
        gc.freeze()
    elif gcmode == "off":
        gc.disable()
    n0 = [g["collections"] for g in gc.get_stats()]
    if os.environ.get("REPEAT_SLABS"):
        res = bench.config5_blocks_leg(ffi, ring, gulp_bytes, 10, 0, nint=nint, nwarm=nwarm, long_len=50, from_slabs=True, in_ring_integrations=depths[r % len(depths)])
        print("from slabs, input ring %2d integrations, run %d: %.4f ms per integration, windows %s, scattered %s" % (depths[r % len(depths)], r, res["ms_per_integration"], res["window_ms"], res["slabs_scattered_after_all"]),
              {k: (v["alloc"], v["stamp_wait"]) for k, v in res["ring_allocations"].items()}, flush=True)
        continue
    res = bench.config5_blocks_leg(ffi, ring, gulp_bytes, 10, 0, nint=nint, nwarm=nwarm, in_ring_integrations=depths[r % len(depths)])
    print("input ring %2d integrations, run %d: %.4f ms per integration, windows %s, corracc %s" % (depths[r % len(depths)], r, res["ms_per_integration"], res["window_ms"], res.get("corracc_mode")),
          "gc collections per generation during the run:", [g["collections"] - a for g, a in zip(gc.get_stats(), n0)],
          "ring allocations:", {k: (v["alloc"], v["stamp_wait"]) for k, v in res["ring_allocations"].items()}, flush=True)
