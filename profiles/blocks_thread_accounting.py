#!/usr/bin/env python3
"""Diagnostic: per block thread of the config-5 topology, how the wall time of a run divides into waiting on a ring
(condition variable), waiting in a blocking library call, and everything else (bytecode + waiting for the interpreter lock).
usage: blocks_thread_accounting.py [nint]"""
import collections
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import blocks_probe as bp  # noqa: E402
import numpy as np  # noqa: E402
from caltech_bifrost_dsp_amd import backend  # noqa: E402

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
born = {}


def timed(owner, name, label):
    f = getattr(owner, name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[threading.current_thread().name][label]
            e[0] += time.perf_counter() - t
            e[1] += 1
    setattr(owner, name, w)


timed(threading.Condition, "wait", "ring wait")
# the native ring (round 4): its calls wait inside the extension
from caltech_bifrost_dsp_amd import _xfast  # noqa: E402
for n in ("ring_acquire", "ring_acquire_parts", "ring_reserve", "ring_next_sequence", "ring_commit_external", "ring_commit"):
    timed(_xfast, n, "ring call (work + wait): " + n)
for n in ("beam_run", "beam_run_parts", "beam_integrate", "beam_mark", "beam_ticket_done", "xgpu_kernel_async", "xgpu_kernel_async_acc", "xgpu_dump_done"):
    timed(_xfast, n, "enqueue / query: " + n)
for n in ("beam_wait", "xgpu_sync_lag", "map_sync", "stream_synchronize", "beam_sync", "xgpu_sync"):
    timed(backend.HipBackend, n, "library wait: " + n)
for n in ("bfBeamformRun", "bfBeamformIntegrate", "beam_mark", "bfXgpuKernelAsync", "bfXgpuKernelAsyncAcc", "map_add_i32", "map_assign_i32"):
    timed(backend.HipBackend, n, "enqueue: " + n)
orig_run = threading.Thread.run


def run(self):
    t = time.perf_counter()
    try:
        orig_run(self)
    finally:
        born[self.name] = time.perf_counter() - t


threading.Thread.run = run


def main():
    nint = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    bp.ffi.call("xengSetDevice", 0)
    gulp_bytes = bp.NTIME_GULP * bp.NCHAN * bp.NINPUT
    bp.run.ring = bp.ffi.DeviceBuffer(10 * gulp_bytes)
    rs = np.random.RandomState(0xdeadbeef)
    for g in range(10):
        bp.run.ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
    bg = int(os.environ.get("XENG_PROBE_BEAM_GULP", "960"))
    isp = int(os.environ.get("XENG_PROBE_IN_SPAN", "4"))
    bp.run(["corr", "cacc", "bf", "sb"], 400, beam_gulp=bg, in_span=isp)          # warm-up: the pinned spans of the slow ring exist afterwards
    acc.clear(); born.clear()
    bp.run(["corr", "cacc", "bf", "sb"], nint, beam_gulp=bg, in_span=isp)
    for th in sorted(born):
        tot = born[th]
        waits = sum(v[0] for v in acc[th].values())
        print("%-12s lived %.3f s = %.1f us per integration; everything else (bytecode + interpreter lock) %.1f us per integration" % (
            th, tot, tot / nint * 1e6, (tot - waits) / nint * 1e6))
        for k, (t, n) in sorted(acc[th].items()):
            print("      %-34s %6d calls %8.1f us per integration (%.1f us per call)" % (k, n, t / nint * 1e6, t / max(n, 1) * 1e6))


if __name__ == "__main__":
    main()
