import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "profiles"))
import numpy as np
import blocks_probe as bp
bp.ffi.call("xengSetDevice", 0)
gulp_bytes = bp.NTIME_GULP * bp.NCHAN * bp.NINPUT
bp.run.ring = bp.ffi.DeviceBuffer(10 * gulp_bytes)
rs = np.random.RandomState(0xdeadbeef)
for g in range(10):
    bp.run.ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
W = ["corr", "cacc", "bf", "sb"]
bp.run(W, 600, beam_gulp=960, in_span=4)     # warm
for rep in range(2):
    for depth, isp in ((4, 4), (8, 4), (4, 8), (8, 8), (2, 8), (6, 6)):
        bp.run(W, 1000, beam_gulp=960, depth=depth, in_span=isp)
