#!/usr/bin/env python3
"""Round 5 (review item 2): HBM-side traffic of the contraction whose epilogue also feeds CorrAcc's long accumulator, one mode per
process so that the counters of `xcorr_fused_kernel<0, true, false>` can be told apart by mode (the mode is a launch parameter,
not a template argument): 0 = plain dumps, 1 = every dump assigns (a = b), 2 = every dump adds (a += b).
Run under `rocprofv3 --pmc ...` by profiles/lacc_pmc_run.sh; config-2 shapes (704 inputs, 96 channels, 5 gulps of 480).
usage: lacc_pmc_probe.py <mode 0|1|2> [integrations]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nint = int(sys.argv[2]) if len(sys.argv) > 2 else 12
NSTAND, NCHAN, NT, G = 352, 96, 480, 5
ffi.call("xengSetDevice", 0)
L = ffi.lib()
gulp = NT * NCHAN * NSTAND * 2
rs = np.random.RandomState(0xdeadbeef)
gulps = [ffi.DeviceBuffer(gulp).upload(rs.randint(0, 255, size=gulp, dtype=np.uint8)) for _ in range(G)]
matbytes = NCHAN * ((NSTAND // 2 + 1) * (NSTAND // 4) * 16) * 2 * 4
ffi.call("xengXgpuConfigure", NSTAND, 2, NCHAN, NT, G)
ffi.call("xengXgpuInitialize", 0)
outs = [ffi.DeviceBuffer(matbytes) for _ in range(3)]
acc = ffi.DeviceBuffer(matbytes)
for it in range(nint):
    for g in range(G):
        dump = int(g == G - 1)
        if mode and dump:
            ffi.check("acc", L.xengXgpuKernelAsyncAcc(gulps[g].ptr, outs[it % 3].ptr, 1, acc.ptr, mode))
        else:
            ffi.check("k", L.xengXgpuKernelAsync(gulps[g].ptr, outs[it % 3].ptr, dump))
    ffi.call("xengXgpuSyncLag", 1)
ffi.call("xengXgpuSync")
print("mode %d: %d integrations done" % (mode, nint), file=sys.stderr)
