#!/usr/bin/env python3
"""Round 5: the block topology with and without the native per-gulp loops (XENG_PUMP=0 / 1), interleaved in one process on one
box: bench.py's corr_block leg (Corr alone), config5_blocks (Corr -> CorrAcc, Beamform -> BeamformSumBeams) and the same from packet slabs.
usage: pump_ab_probe.py [rounds] [legs: corr,blocks,slabs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
legs = (sys.argv[2] if len(sys.argv) > 2 else "corr,blocks,slabs").split(",")
ffi.call("xengSetDevice", 0)
ffi.call("xengXgpuConfigure", bench.NSTAND, bench.NPOL, bench.NCHAN, bench.NTIME_GULP, bench.ACC_LEN // bench.NTIME_GULP)
ffi.call("xengXgpuInitialize", 0)
gulp_bytes = bench.NTIME_GULP * bench.NCHAN * bench.NINPUT
ring = ffi.DeviceBuffer(10 * gulp_bytes)
ring.upload(np.random.RandomState(1).randint(0, 256, size=10 * gulp_bytes, dtype=np.uint8))
res = {}
for r in range(rounds):
    for pump in ("1", "0"):
        os.environ["XENG_PUMP"] = pump
        if "corr" in legs:
            v = bench.corr_block_leg(ffi, ring, gulp_bytes, 10, 0)["ms_per_integration"]
            res.setdefault(("corr_block", pump), []).append(v)
        if "blocks" in legs:
            d = bench.config5_blocks_leg(ffi, ring, gulp_bytes, 10, 0)
            res.setdefault(("config5_blocks", pump), []).append(d["ms_per_integration"])
            print("   windows", d["window_ms"], flush=True)
        if "slabs" in legs:
            d = bench.config5_blocks_leg(ffi, ring, gulp_bytes, 10, 0, nint=300, nwarm=300, long_len=50, from_slabs=True)
            res.setdefault(("config5_blocks_from_slabs", pump), []).append(d["ms_per_integration"])
            print("   windows", d["window_ms"], flush=True)
        print("round %d XENG_PUMP=%s " % (r, pump) + "  ".join("%s=%.4f" % (k[0], v[-1]) for k, v in res.items() if k[1] == pump), flush=True)
print("# ms per integration; median (min .. max) over %d interleaved rounds" % rounds)
for k in sorted(res):
    v = sorted(res[k])
    print("%-28s XENG_PUMP=%s  %.4f (%.4f .. %.4f)" % (k[0], k[1], v[len(v) // 2], v[0], v[-1]))
