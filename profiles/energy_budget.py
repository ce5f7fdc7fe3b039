#!/usr/bin/env python3
"""Energy budget of the fused X-engine kernel at the board's power cap.

The kernel runs at the package power limit, so a step costs (energy per integration) / (power): what a component costs is
measured by removing it and reading time AND power.  Needs the -DXENG_DIAGNOSTICS build (timing-only ablations; results
are wrong by construction): every variant streams config-2 integrations for ~3 s while a thread samples the package
power (hwmon power1_average, else rocm-smi), all in one process on one device, variants interleaved over the rounds.

  XENG_LIB=profiles/_ab/libxeng_diag.so python3 profiles/energy_budget.py [rounds] [seconds per variant]

Ablation bits (csrc/xcorr_kernels.h): 1 no LDS-DMA in the K loop (L2 -> LDS staging), 2 no nibble unpack (raw bytes into
the MFMAs: different operand statistics), 4 no LDS reads (operands constant), 8 no stage barrier, 16 no epilogue (nothing
stored), 32 every channel reads a 1 MB window of gulp 0 (input served by L2, no HBM fetch)."""
import ctypes
import glob
import os
import statistics
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NPOL, NCHAN, NTIME_GULP, ACC_LEN = 352, 2, 96, 480, 2400
NINPUT = NSTAND * NPOL
VARIANTS = [(0, "shipped kernel"), (32, "input from an L2-resident window (no HBM fetch of the voltages)"),
            (16, "no epilogue (nothing stored)"), (48, "both of the above"), (1, "no LDS-DMA in the K loop (no L2 -> LDS staging)"),
            (17, "no LDS-DMA, no epilogue"), (5, "no LDS-DMA, no LDS reads (operands constant, unpack still runs)"),
            (31, "MFMAs only: no DMA, unpack, LDS reads, barriers, epilogue (operands constant)")]


def power_reader():
    """-> callable returning package watts (None when the platform offers nothing)."""
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        files = [f for f in glob.glob(pat)]
        if files:
            def rd(fs=files):
                best = 0.0
                for f in fs:
                    try:
                        best = max(best, int(open(f).read().strip()) / 1e6)
                    except (OSError, ValueError):
                        pass
                return best or None
            if rd():
                return rd, files[0]
    def smi():
        try:
            out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            vals = [float(l.split(":")[-1]) for l in out.splitlines() if "Power (W)" in l]
            return max(vals) if vals else None
        except (OSError, ValueError, subprocess.TimeoutExpired):
            return None
    return smi, "rocm-smi --showpower"


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    ffi.call("xengSetDevice", 0)
    gulp_bytes = NTIME_GULP * NCHAN * NINPUT
    matlen = NCHAN * 249216
    ring = ffi.DeviceBuffer(10 * gulp_bytes)
    rs = np.random.RandomState(0xdeadbeef)
    for g in range(10):
        ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
    outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]
    L = ffi.lib()
    gps = ACC_LEN // NTIME_GULP
    ffi.call("xengXgpuConfigure", NSTAND, NPOL, NCHAN, NTIME_GULP, gps)
    ffi.call("xengXgpuInitialize", 0)
    rd, src = power_reader()
    print("power source: %s" % src, flush=True)
    gi = [0]

    def step():
        out = outs[(gi[0] // gps) & 1]
        for g in range(gps):
            rc = L.xengXgpuKernelAsync(ctypes.c_void_p(ring.ptr + (gi[0] % 10) * gulp_bytes), ctypes.c_void_p(out.ptr), int(g == gps - 1))
            if rc:
                ffi.check("kernel", rc)
            gi[0] += 1
        L.xengXgpuSyncLag(1)

    res = {a: [] for a, _ in VARIANTS}
    for rnd in range(rounds):
        for abl, _ in VARIANTS:
            os.environ["XENG_ABLATE"] = str(abl)
            for _ in range(1500):                 # settle clocks / power on this variant
                step()
            ffi.call("xengDeviceSynchronize")
            watts, stop = [], threading.Event()

            def sampler():
                while not stop.is_set():
                    w = rd()
                    if w:
                        watts.append(w)
                    time.sleep(0.05)
            th = threading.Thread(target=sampler)
            th.start()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < secs:
                for _ in range(200):
                    step()
                n += 200
            ffi.call("xengDeviceSynchronize")
            el = time.perf_counter() - t0
            stop.set()
            th.join()
            w = statistics.mean(watts[2:]) if len(watts) > 4 else float("nan")
            res[abl].append((el / n * 1e3, w))
            print("round %d ablate %2d: %.4f ms/step  %7.1f W  %6.1f mJ per integration" % (rnd, abl, el / n * 1e3, w, el / n * w * 1e3), flush=True)
    del os.environ["XENG_ABLATE"]
    time.sleep(1.0)
    idle = [rd() for _ in range(10) if not time.sleep(0.1)]
    idle = statistics.mean([x for x in idle if x]) if any(idle) else float("nan")
    print("\nidle package power: %.1f W" % idle)
    print("%-3s %-86s %9s %8s %9s %9s" % ("abl", "variant", "ms/step", "W", "mJ/int", "mJ above idle"))
    base = None
    for abl, name in VARIANTS:
        ms = statistics.median(x[0] for x in res[abl])
        w = statistics.median(x[1] for x in res[abl])
        mj, mjd = ms * w, ms * (w - idle)
        if abl == 0:
            base = (ms, w, mj, mjd)
        print("%-3d %-86s %9.4f %8.1f %9.1f %9.1f" % (abl, name, ms, w, mj, mjd))
    ffi.call("xengXgpuDestroy")


if __name__ == "__main__":
    main()
