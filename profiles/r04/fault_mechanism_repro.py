"""Round-3 fault, reproduced on the CPU with the ring / array / DeviceBuffer classes as they were at 29405b0^ plus the
cached-reference variant (array -> reference -> array): which finalisers does the cycle collector run, and in which order?"""
import gc, sys, itertools
sys.path.insert(0, "/tmp/oldpkg")
import pkg.ffi as ffi
import pkg.ndarray as nd
import pkg.ring as ringmod

events = []
addr = itertools.count(0x1000000, 0x100000)
def fake_init(self, nbytes, space=ffi.SPACE_CUDA):
    self.nbytes, self.space, self.ptr = int(nbytes), space, next(addr)
    events.append(("alloc", self.ptr))
def fake_free(self):
    if self.ptr:
        events.append(("hipFree", self.ptr))
        self.ptr = None
ffi.DeviceBuffer.__init__ = fake_init
ffi.DeviceBuffer.free = fake_free
ffi.call = lambda name, *a: None
gc.disable()
ring = ringmod.Ring("bf-output", space="cuda")
ring.resize(1024, 8 * 1024)
live = []
def make_span(cyclic):
    x = ring._alloc_span(1024)
    if cyclic:
        x._cached_ref = [x]          # the variant of the faulting run: array -> cached as_BFarray reference -> array
    return x
# 1. a span is allocated, lives through one collection while referenced (as a span in `pending` does), then dies
x = make_span(True)
p0 = x.ptr
gc.collect()                         # survivors are re-linked: the owner now precedes its DeviceBuffer in the collector's list
del x
gc.collect()
in_pool = [b for lst in ring._pool.values() for b in lst]
print("events:", [(e, hex(p)) for e, p in events])
print("pool after the collection:", [(hex(b.ptr) if b.ptr else None, b.nbytes) for b in in_pool])
freed = {p for e, p in events if e == "hipFree"}
print("allocation %#x: hipFree'd = %s, and a DeviceBuffer object for it sits in the ring's free list = %s" % (p0, p0 in freed, len(in_pool) == 1))

# 2. instrumented: order of the two finalisers, with and without a survived collection in between
for survive in (False, True):
    events.clear()
    order = []
    orig_put = ringmod.Ring._pool_put
    def put(self, buf, _o=orig_put):
        order.append(("owner.__del__ -> pool_put", hex(buf.ptr) if buf.ptr else None))
        return _o(self, buf)
    ringmod.Ring._pool_put = put
    def fake_free2(self):
        order.append(("DeviceBuffer.free", hex(self.ptr) if self.ptr else None))
        if self.ptr:
            events.append(("hipFree", self.ptr)); self.ptr = None
    ffi.DeviceBuffer.free = fake_free2
    x = make_span(True)
    if survive:
        gc.collect()
    del x
    gc.collect()
    print("survived a collection first:", survive, "->", order, "| pool:", [(hex(b.ptr) if b.ptr else None) for l in ring._pool.values() for b in l])
    ringmod.Ring._pool_put = orig_put
