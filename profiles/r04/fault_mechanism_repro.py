#!/usr/bin/env python3
"""Round-3 fault, the mechanism reproduced on the CPU (no GPU, nothing freed for real): the ring / array / DeviceBuffer
classes exactly as they were before the fix commit (29405b0^, extracted from this repository's history into a temporary
directory) plus the cached-reference variant of the faulting run (array -> reference -> array).  Question: when the cycle
collector finds such a span, which finalisers run, and does the allocation go back to the ring's free list?
Answer (fault_mechanism_repro.txt): the collector clears the owner's weak reference to its ring first, so the owner takes
its `buf.free()` branch, and `DeviceBuffer.__del__` runs as well: the allocation is really freed (hipFree / hipHostFree in
the product) at collector time, on whichever thread the collector runs -- never recycled."""
import gc
import itertools
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "pkg"))
for f in ("__init__.py", "ring.py", "ndarray.py", "ffi.py", "proclog.py"):
    src = subprocess.run(["git", "-C", ROOT, "show", "29405b0^:caltech-bifrost-dsp_amd/" + f], capture_output=True, text=True, check=True).stdout
    open(os.path.join(tmp, "pkg", f), "w").write(src)
sys.path.insert(0, tmp)
import pkg.ffi as ffi  # noqa: E402
import pkg.ring as ringmod  # noqa: E402

events = []
addr = itertools.count(0x1000000, 0x100000)


def fake_init(self, nbytes, space=ffi.SPACE_CUDA):
    self.nbytes, self.space, self.ptr = int(nbytes), space, next(addr)
    events.append(("alloc", hex(self.ptr)))


ffi.DeviceBuffer.__init__ = fake_init
ffi.call = lambda name, *a: None
gc.disable()
ring = ringmod.Ring("bf-output", space="cuda")
ring.resize(1024, 8 * 1024)
for survive in (False, True):
    order = []
    orig_put = ringmod.Ring._pool_put

    def put(self, buf, _o=orig_put):
        order.append(("owner.__del__ -> ring._pool_put", hex(buf.ptr) if buf.ptr else None))
        return _o(self, buf)
    ringmod.Ring._pool_put = put

    def fake_free(self):
        order.append(("DeviceBuffer.free -> hipFree", hex(self.ptr) if self.ptr else None))
        self.ptr = None
    ffi.DeviceBuffer.free = fake_free
    x = ring._alloc_span(1024)
    x._cached_ref = [x]                  # the variant of the faulting run: array -> cached as_BFarray reference -> array
    if survive:
        gc.collect()                     # (the span lives through one collection while referenced, as a span in `pending` does)
    del x
    gc.collect()
    print("span survived a collection first: %-5s finalisers run by the collector: %s | ring free list: %s" % (
        survive, order, [hex(b.ptr) if b.ptr else None for lst in ring._pool.values() for b in lst]))
    ringmod.Ring._pool_put = orig_put
print("-> the owner never reaches ring._pool_put (its weak reference to the ring is cleared before the finalisers run): the allocation is "
      "freed for real at collector time")
