"""Span memory lifetime of the rings (round 4; runs on both implementations, no GPU).

The round-3 review's W2: GPU memory lifetime was decided by Python reference counts plus a convention ("whoever dropped the
last reference had waited").  Now a released allocation carries a stamp -- here the tickets of a fake backend, in the
product libxeng's stream clocks -- and is reissued or freed only once that stamp is complete; and a reader that registers
late skips ahead instead of killing its block thread (the span discipline the reference relies on: corr_block.py:433-452)."""
import gc
import json
import threading
import time

import numpy as np

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd import ring as ringmod
from caltech_bifrost_dsp_amd.ring import Ring, WriteSpan


class FakeTickets:
    """The completion record of a fake backend: `issue()` enqueues a kernel (returns its ticket), `complete(t)` lets the
    GPU reach it.  A stamp is the latest ticket issued when the buffer was released."""

    def __init__(self):
        self.issued = 0
        self.completed = 0
        self.cv = threading.Condition()
        self.waits = []

    def issue(self):
        with self.cv:
            self.issued += 1
            return self.issued

    def complete(self, t):
        with self.cv:
            self.completed = max(self.completed, t)
            self.cv.notify_all()

    # the stamp-source protocol of the rings
    def now(self):
        return self.issued

    def done(self, stamp):
        return stamp <= self.completed, True

    def wait(self, stamp):
        with self.cv:
            self.waits.append(stamp)
            while self.completed < stamp:
                self.cv.wait(0.05)


def _one_span_through(ring, nbytes, fill):
    """writer reserves + commits one span, a reader reads it and lets go of everything; returns the span's address"""
    gen = ring.read(guarantee=True)
    with ring.begin_writing() as w:
        with w.begin_sequence(time_tag=0, header="{}") as oseq:
            with oseq.reserve(nbytes) as sp:
                sp.data.numpy()[...] = fill
                addr = sp.data.ptr
            del sp
    for iseq in gen:
        for ispan in iseq.read(nbytes):
            assert int(ispan.data.numpy()[0]) == fill
        del ispan
    del iseq, gen
    return addr


def test_a_buffer_released_under_an_unfinished_ticket_is_not_reissued_until_it_completes(ring_impl):
    tk = FakeTickets()
    r = Ring(name="lifetime", space="system")
    r.resize(64, 256)
    r.set_stamp_source(tk)
    t1 = tk.issue()                          # a kernel that writes the span is "in flight" ...
    addr = _one_span_through(r, 64, 7)       # ... while its last user lets go of it: stamped with ticket t1
    gc.collect()
    got = []

    def writer():
        with r.begin_writing() as w:
            with w.begin_sequence(time_tag=1, header="{}") as oseq:
                sp = oseq.reserve(64)        # the free list holds the one allocation, not complete: this waits
                got.append(sp.data.ptr)
                sp.close()

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    time.sleep(0.3)
    assert not got, "the allocation was reissued while ticket %d was unfinished" % t1
    assert tk.waits == [t1]                  # it is waiting for exactly that ticket
    tk.complete(t1)
    th.join(10)
    assert not th.is_alive() and got == [addr]        # ... and gets the same allocation once the ticket has completed
    c = r.counters
    assert c["reuse"] == 1 and c["stamp_wait"] == 1 and c["alloc"] == 1 and c["free"] == 0


def test_a_complete_stamp_is_reissued_without_waiting_and_oldest_first(ring_impl):
    tk = FakeTickets()
    r = Ring(name="fifo", space="system")
    r.resize(32, 1024)
    r.set_stamp_source(tk)
    gen = r.read(guarantee=True)
    addrs = []
    with r.begin_writing() as w:
        with w.begin_sequence(time_tag=0, header="{}") as oseq:
            for k in range(3):
                tk.issue()
                with oseq.reserve(32) as sp:
                    addrs.append(sp.data.ptr)
                del sp
    for iseq in gen:
        for ispan in iseq.read(32):
            pass
        del ispan
    del iseq, gen
    tk.complete(tk.issued)
    again = []
    with r.begin_writing() as w:
        with w.begin_sequence(time_tag=1, header="{}") as oseq:
            for k in range(3):
                sp = oseq.reserve(32)
                again.append(sp.data.ptr)
                sp.close()
                del sp
    assert again[0] == addrs[0]              # first released, first reissued (the oldest stamp is the likeliest to be complete)
    assert tk.waits == []
    assert r.counters["alloc"] == 3


def test_a_late_reader_skips_to_the_oldest_live_span(ring_impl):
    """A reader that registers after the writer has started (and after spans were overwritten: nobody applied back-pressure)
    starts at the oldest span still there, on a gulp boundary, and is told how much it missed -- round 3 raised 'data at 0
    was overwritten before it was read' here, which killed the block thread (BeamformSumBeams in gpurun_out/s2p/bp.txt)."""
    r = Ring(name="late", space="system")
    r.resize(8, 32)                          # room for four spans
    w = r.begin_writing()
    oseq = w.begin_sequence(time_tag=3, header=json.dumps({"seq0": 0}))
    for k in range(10):                      # no reader: the oldest spans are overwritten
        with oseq.reserve(8) as sp:
            sp.data.numpy()[...] = k
    seen = []
    gen = r.read(guarantee=True)             # registers late

    def reader():
        for iseq in gen:
            for ispan in iseq.read(8):
                seen.append((int(ispan.data.numpy()[0]), ispan.offset, ispan.skipped))

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    for k in range(10, 14):
        with oseq.reserve(8) as sp:
            sp.data.numpy()[...] = k
    oseq.end()
    w.__exit__(None, None, None)
    th.join(10)
    assert not th.is_alive()
    vals = [v for v, _, _ in seen]
    assert vals == list(range(vals[0], 14)) and 0 < vals[0] <= 9      # in order, nothing after the first span missed
    assert seen[0][2] == 8 * vals[0] and seen[0][1] == 8 * vals[0]     # told what it missed; offsets stay on gulp boundaries
    assert all(s == 0 for _, _, s in seen[1:])


def test_a_late_reader_does_not_see_sequences_that_are_gone(ring_impl):
    r = Ring(name="late-seq", space="system")
    r.resize(8, 16)                          # room for two spans
    with r.begin_writing() as w:
        with w.begin_sequence(time_tag=1, header=json.dumps({"n": 1})) as oseq:
            for k in range(5):
                with oseq.reserve(8) as sp:
                    sp.data.numpy()[...] = k
        with w.begin_sequence(time_tag=2, header=json.dumps({"n": 2})) as oseq:
            with oseq.reserve(8) as sp:
                sp.data.numpy()[...] = 50
            gen = r.read(guarantee=True)     # sequence 1 still holds its last span: it is the earliest sequence in the ring
    got = [(json.loads(iseq.header.tostring())["n"], [(int(s.data.numpy()[0]), s.skipped) for s in iseq.read(8)]) for iseq in gen]
    assert got == [(1, [(4, 32)]), (2, [(50, 0)])]
    # ... and one that registers when everything before the open sequence is gone starts at that sequence
    r2 = Ring(name="late-seq2", space="system")
    r2.resize(8, 8)
    with r2.begin_writing() as w:
        with w.begin_sequence(time_tag=1, header=json.dumps({"n": 1})) as oseq:
            with oseq.reserve(8) as sp:
                sp.data.numpy()[...] = 1
        with w.begin_sequence(time_tag=2, header=json.dumps({"n": 2})) as oseq:
            with oseq.reserve(8) as sp:
                sp.data.numpy()[...] = 2
            gen = r2.read(guarantee=True)
    assert [json.loads(iseq.header.tostring())["n"] for iseq in gen] == [2]


def test_collector_time_release_goes_back_to_the_free_list(ring_impl):
    """Span arrays caught in a reference cycle are released by the cycle collector, whenever and on whichever thread it
    runs.  Round 3: the collector cleared the owner's weak reference to its ring first and then ran BOTH finalisers, so the
    allocation was really freed at collector time (profiles/r04/fault_mechanism_repro.txt).  Now: back to the free list,
    stamped; nothing is freed."""
    tk = FakeTickets()
    r = Ring(name="cyc", space="system")
    r.resize(16, 64)
    r.set_stamp_source(tk)
    was = gc.isenabled()
    gc.disable()
    try:
        gen = r.read(guarantee=True)
        with r.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header="{}") as oseq:
                sp = oseq.reserve(16)
                addr = sp.data.ptr
                sp.data.cycle = [sp.data]            # array -> list -> array: only the collector can release it
                sp.close()
                del sp
        for iseq in gen:
            for ispan in iseq.read(16):
                pass
            del ispan
        del iseq, gen
        assert r.counters["free"] == 0 and r.counters["reuse"] == 0
        tk.issue()
        gc.collect()
        tk.complete(tk.issued)
        sp2 = WriteSpanOnNewSequence(r, 16)
        assert sp2.data.ptr == addr and r.counters["free"] == 0 and r.counters["reuse"] == 1
    finally:
        if was:
            gc.enable()


def WriteSpanOnNewSequence(r, nbytes):
    w = r.begin_writing()
    w.begin_sequence(time_tag=9, header="{}")
    return WriteSpan(r, nbytes)


def test_spans_outlive_their_ring_handle(ring_impl):
    """A reader may keep a span after the ring object is gone (Corr holds gulps until their dump has run): the memory stays
    valid until the last array lets go."""
    r = Ring(name="short-lived", space="system")
    r.resize(16, 64)
    gen = r.read(guarantee=True)
    with r.begin_writing() as w:
        with w.begin_sequence(time_tag=0, header="{}") as oseq:
            with oseq.reserve(16) as sp:
                sp.data.numpy()[...] = np.arange(16, dtype=np.uint8)
            del sp
    kept = None
    for iseq in gen:
        for ispan in iseq.read(16):
            kept = ispan.data
    del iseq, ispan, gen, r, w, oseq
    gc.collect()
    assert np.array_equal(kept.numpy(), np.arange(16, dtype=np.uint8))


def test_library_stamps_without_a_device():
    """xengStampNow / Done / Wait on a machine without a GPU: nothing can be in flight, so a stamp is complete at once (the
    device rings' stamp source: ring.LibraryStamps)."""
    from caltech_bifrost_dsp_amd import ffi
    import ctypes
    n = ctypes.c_int(-1)
    if ffi.lib().xengGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        import pytest
        pytest.skip("a GPU is present")
    src = ringmod.LibraryStamps()
    s = src.now()
    assert src.done(s) == (True, True)
    src.wait(s)


# ---------------------------------------------------------------- round 5 (review W2/W3)
def test_a_reader_that_was_registered_but_never_iterated_is_given_back(ring_impl):
    """`ring.read()` registers the reader at the call.  A block that registers at construction (CorrAcc) and is torn down before
    main() ever runs must not leave a guaranteed reader behind: the writer would wait for room for ever."""
    r = Ring(name="never-started", space="system")
    r.resize(8, 16)                          # room for two spans
    gen = r.read(guarantee=True)
    assert len(r._readers) == 1
    del gen                                  # never iterated: no generator `finally` runs -- the registration object closes the reader
    gc.collect()
    assert len(r._readers) == 0
    gen2 = r.read(guarantee=True)
    gen2.close()                             # ... and so does an explicit close() (CorrAcc.shutdown)
    assert len(r._readers) == 0
    done = []

    def writer():
        with r.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header="{}") as oseq:
                for k in range(6):           # three times the ring: would block behind a guaranteed reader that never reads
                    with oseq.reserve(8) as sp:
                        sp.data.numpy()[...] = k
        done.append(True)

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    th.join(10)
    assert done == [True]


def test_a_span_released_from_a_foreign_thread_keeps_its_stamp(ring_impl):
    """The last reference of a span may be dropped by any thread -- a publish helper, the collector, teardown.  The stamp is taken
    at that moment, by that thread, and the allocation still is not reissued before the stamp completes."""
    tk = FakeTickets()
    r = Ring(name="foreign", space="system")
    r.resize(64, 256)
    r.set_stamp_source(tk)
    gen = r.read(guarantee=True)
    with r.begin_writing() as w:
        with w.begin_sequence(time_tag=0, header="{}") as oseq:
            with oseq.reserve(64) as sp:
                addr = sp.data.ptr
            del sp
    held = []
    for iseq in gen:
        for ispan in iseq.read(64):
            held.append(ispan.data)
        del ispan
    del iseq, gen
    t1 = tk.issue()                          # "a kernel that reads the span" is enqueued ...

    def foreign():
        held.clear()                         # ... and a thread that never touched the ring drops the last reference
        gc.collect()

    th = threading.Thread(target=foreign)
    th.start()
    th.join()
    got = []

    def writer():
        with r.begin_writing() as w:
            with w.begin_sequence(time_tag=1, header="{}") as oseq:
                sp = oseq.reserve(64)
                got.append(sp.data.ptr)
                sp.close()

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    time.sleep(0.3)
    assert not got and tk.waits == [t1]
    tk.complete(t1)
    th.join(10)
    assert got == [addr]


def test_declared_streams_narrow_a_stamp_only_while_every_user_has_declared(ring_impl):
    """xengRingDeclareStreams: one user of the ring that never declared (a duck-typed block, a test reader with kernels of its
    own) and every stamp waits for all streams again -- a declaration by the others must not uncover its spans."""
    import pytest
    from caltech_bifrost_dsp_amd import ffi
    if ring_impl == "python":
        n = __import__("ctypes").c_int(-1)
        if not (ffi.lib().xengGetDeviceCount(__import__("ctypes").byref(n)) == 0 and n.value > 0):
            pytest.skip("the Python ring keeps library stamps only in the device spaces: needs a GPU")
    ALL, BEAM = 31, ffi.STREAMS["beam"]
    r = Ring(name="decl", space="system" if ring_impl == "native" else "cuda")
    assert r.stamp_classes() == (ALL, 0, 0)
    r.declare_streams("beam")                # the writer's block
    w = r.begin_writing()
    oseq = w.begin_sequence(time_tag=0, header="{}")
    assert r.stamp_classes() == (BEAM, 1, 1)
    g1 = r.read(guarantee=True)              # a reader whose block has not declared
    assert r.stamp_classes() == (ALL, 1, 2)
    r.declare_streams("beam")                # ... now it has
    assert r.stamp_classes() == (BEAM, 2, 2)
    g2 = r.read(guarantee=False)             # a second, undeclared reader (a test sink, a third-party block)
    assert r.stamp_classes() == (ALL, 2, 3)
    r.declare_streams()                      # a host-only user declares "no streams"
    assert r.stamp_classes() == (BEAM, 3, 3)
    g1.close(); g2.close()
    oseq.end()
    w.__exit__(None, None, None)
