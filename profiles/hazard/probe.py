#!/usr/bin/env python3
"""Diagnostic: do cross-lane reads in a small kernel stay correct while the X-engine runs on the same CUs?
Modes 0-3: ds_bpermute / DPP on tags and on packed-fp32 sums (round 2: all clean).  Modes 4-7: the packed multiply the
hazard lab singled out (diag_probe.hip, pkmul_probe_kernel).  Needs a scratch library that contains diag_probe.hip (see
profiles/hazard/probe.sh); XENG_LIB points to it.   usage: probe.py [first mode [last mode]]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import caltech_bifrost_dsp_amd  # noqa: F401,E402
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

L = ffi.lib()
probe = L.xengDiagBpermProbe
probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.restype = ctypes.c_int
NSTAND, NCHAN = 352, 96
gb = 480 * NCHAN * 704
matlen = NCHAN * 249216
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, 480, 5)
ffi.call("xengXgpuInitialize", 0)
ring = ffi.DeviceBuffer(5 * gb)
ring.upload(np.random.RandomState(1).randint(0, 255, size=5 * gb, dtype=np.uint8))
outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]


CAL = {}


def run_probe(mode, label, nrep):
    tot = np.zeros(8, dtype=np.uint64)
    first = 0
    for _ in range(nrep):
        h = np.zeros(8, dtype=np.uint64)
        if 200 <= mode < 216:                      # v_pk_mov_b32: expected source registers, calibrated on the idle GPU
            h[0] = CAL[mode]
        ffi.check("probe", probe(mode, 3000 if mode < 2 else 600, 1024, h.ctypes.data))
        tot[:6] += h[:6]
        first = first or int(h[6])
    checked = nrep * (3000 if mode < 2 else 600) * 1024 * 256 * 4
    if mode >= 4:
        print("%-64s %d wrong of %.2e products; by lane quarter %s; wrong LOW halves: %d; first: got %08x want %08x" % (
            label, int(tot[0]), checked / 4, [int(v) for v in tot[1:5]], int(tot[5]), first >> 32, first & 0xFFFFFFFF), flush=True)
    else:
        print("%-64s %d wrong of %.2e reads; by position in the group of four %s; lanes 48-63: %d; first: got %08x want %08x" % (
            label, int(tot[0]), checked, [int(v) for v in tot[1:5]], int(tot[5]), first >> 32, first & 0xFFFFFFFF), flush=True)


import threading
import time

stop = threading.Event()
count = [0]


def feeder():                                      # keeps the contraction running (lag-1 streaming) until told to stop
    n = 0
    while not stop.is_set():
        for g in range(5):
            ffi.check("k", L.xengXgpuKernelAsync(ring.ptr + g * gb, outs[n & 1].ptr, int(g == 4)))
        ffi.call("xengXgpuSyncLag", 1)
        n += 1
    ffi.call("xengXgpuSync")
    count[0] = n


MODES = ((0, "ds_bpermute_b32"), (1, "DPP row_shl/row_shr"), (2, "packed-fp32 sums -> ds_bpermute_b32"), (3, "packed-fp32 sums -> DPP"),
         (4, "v_pk_mul_f32 D, D, S op_sel:[0,1]"), (5, "v_pk_mul_f32 D, S, D op_sel:[1,0]"), (6, "v_pk_mul_f32 E, D, S op_sel:[0,1]"),
         (7, "two v_mul_f32 (control)"))
OPS = ("v_pk_mul_f32 D, A, B", "v_pk_add_f32 D, A, B", "v_pk_fma_f32 D, A, B, C (selects on A, B)", "v_pk_fma_f32 D, A, B, C (selects on B, C)")
SWEEP = tuple((100 + 16 * op + m, "%s op_sel:[%d,%d] op_sel_hi:[%d,%d]" % (OPS[op], m & 1, (m >> 1) & 1, (m >> 2) & 1, (m >> 3) & 1))
              for op in range(4) for m in range(16))
CLS = ("v_pk_mov_b32 D, A, B", "v_pk_add_u16 D, A, B", "v_pk_fma_f16 D, A, B, C (selects on A, B)")
CLASSES = tuple((200 + 16 * c + m, "%s op_sel:[%d,%d] op_sel_hi:[%d,%d]" % (CLS[c], m & 1, (m >> 1) & 1, (m >> 2) & 1, (m >> 3) & 1))
                for c in range(3) for m in range(16))
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
for mode, name in [mn for mn in MODES + SWEEP + CLASSES if lo <= mn[0] <= hi]:
    if 200 <= mode < 216:                          # which source register lands where: read off an idle GPU, one lane
        h = np.zeros(8, dtype=np.uint64)
        h[0] = 1 << 16
        ffi.check("probe", probe(mode, 1, 1, h.ctypes.data))
        CAL[mode] = int(h[7]) & 0xFFFF
        name += " (idle: low = reg %d, high = reg %d of a.x a.y b.x b.y)" % (CAL[mode] & 0xFF, CAL[mode] >> 8)
    if mode < 100 or mode >= 200:
        run_probe(mode, name + ", GPU otherwise idle:", 1 if mode >= 200 else 2)
    stop.clear()
    th = threading.Thread(target=feeder)
    th.start()
    time.sleep(0.05)
    t0 = time.perf_counter()
    run_probe(mode, name + ", beside the X-engine:", 4)
    el = time.perf_counter() - t0
    stop.set()
    th.join()
    if mode < 100:
        print("   (%d contractions ran during %.2f s of probing)" % (count[0], el), flush=True)
