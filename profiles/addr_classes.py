#!/usr/bin/env python3
"""Diagnostic (round 4): which virtual-address ranges this box hands out for hipMalloc, hipHostMalloc, pageable host memory,
thread stacks and the loaded library -- to classify the address of the round-3 GPU memory fault
(profiles/r03/fault_reference_cycle.txt: write to 0x792790f86000).  Allocates and frees only; launches nothing."""
import ctypes
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

FAULT = 0x792790f86000
ffi.call("xengSetDevice", 0)


def maps():
    out = []
    with open("/proc/self/maps") as fh:
        for line in fh:
            f = line.split()
            lo, hi = (int(v, 16) for v in f[0].split("-"))
            out.append((lo, hi, f[1], f[5] if len(f) > 5 else ""))
    return out


def where(p, m):
    for lo, hi, perm, name in m:
        if lo <= p < hi:
            return "[%012x-%012x %s %s] (%.1f MB, offset %#x)" % (lo, hi, perm, name or "anon", (hi - lo) / 1e6, p - lo)
    return "not in /proc/self/maps (no CPU mapping)"


sizes = {"gulp 32.4 MB": 480 * 96 * 704, "visibility span 191 MB": 191397888, "beam span 11.8 MB": 96 * 32 * 480 * 8,
         "power span 480 KB": 16 * 20 * 96 * 16, "slow span 383 MB": 2 * 191397888 // 2 * 2}
dev = {k: ffi.DeviceBuffer(v, ffi.SPACE_CUDA) for k, v in sizes.items()}
pin = {k: ffi.DeviceBuffer(v, ffi.SPACE_CUDA_HOST) for k, v in sizes.items() if "191" not in k}
big = np.empty(64 << 20, np.uint8)
small = np.empty(4096, np.uint8)
stack = {}


def th():
    x = ctypes.c_int(0)
    stack["thread"] = ctypes.addressof(x)


t = threading.Thread(target=th)
t.start()
t.join()
x0 = ctypes.c_int(0)
m = maps()
print("fault address of round 3: %#x (2 MB offset %#x, 4 KB aligned %s)" % (FAULT, FAULT & 0x1fffff, FAULT % 4096 == 0))
print("--- hipMalloc (device)")
for k, b in dev.items():
    print("  %-24s %#014x  2MB-aligned %-5s  %s" % (k, b.ptr, b.ptr % (2 << 20) == 0, where(b.ptr, m)))
print("--- hipHostMalloc (pinned host)")
for k, b in pin.items():
    print("  %-24s %#014x  2MB-aligned %-5s  %s" % (k, b.ptr, b.ptr % (2 << 20) == 0, where(b.ptr, m)))
print("--- pageable host")
print("  numpy 64 MB              %#014x  %s" % (big.ctypes.data, where(big.ctypes.data, m)))
print("  numpy 4 KB               %#014x  %s" % (small.ctypes.data, where(small.ctypes.data, m)))
print("  main stack               %#014x  %s" % (ctypes.addressof(x0), where(ctypes.addressof(x0), m)))
print("  thread stack             %#014x  %s" % (stack["thread"], where(stack["thread"], m)))
print("--- libraries / devices in /proc/self/maps")
seen = set()
for lo, hi, perm, name in m:
    base = os.path.basename(name)
    if ("libxeng" in base or "libamdhip" in base or "libhsa-runtime" in base or name.startswith("/dev/")) and (base, perm) not in seen:
        seen.add((base, perm))
        print("  %012x-%012x %s %s" % (lo, hi, perm, name))
print("--- large reservations (>= 1 GB) and the ten mappings nearest below/above each class")
for lo, hi, perm, name in m:
    if hi - lo >= (1 << 30):
        print("  %012x-%012x %s %-30s %.1f GB" % (lo, hi, perm, name or "anon", (hi - lo) / 2**30))
print("--- free + reallocate: is a freed address handed out again?")
for k in ("power span 480 KB", "beam span 11.8 MB"):
    for space, table in ((ffi.SPACE_CUDA, dev), (ffi.SPACE_CUDA_HOST, pin)):
        old = table[k].ptr
        table[k].free()
        nb = ffi.DeviceBuffer(sizes[k], space)
        print("  %-10s %-22s freed %#014x -> next allocation %#014x (%s)" % ("device" if space == ffi.SPACE_CUDA else "pinned", k, old, nb.ptr,
                                                                            "same address" if nb.ptr == old else "different"))
        table[k] = nb
print("--- address span of each class in this process")
for label, table in (("device", dev), ("pinned", pin)):
    ps = [b.ptr for b in table.values()]
    print("  %-7s %#014x .. %#014x" % (label, min(ps), max(ps)))
