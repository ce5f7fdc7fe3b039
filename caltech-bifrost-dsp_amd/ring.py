"""A minimal ring buffer implementing exactly the protocol the hot-path blocks use.

The reference blocks sit on `bifrost.ring.Ring`.  They only need this method set (enumerated
from corr_block.py, corr_acc_block.py, beamform_block.py, beamform_sum_beams_block.py;
SURVEY.md section 8b):

    Ring(name=, space=) .name .resize(contig_bytes, total_span=)
    ring.begin_writing() ctx -> .begin_sequence(time_tag=, header=<json str>, nringlet=) (ctx or .end())
    oseq.reserve(nbytes) ctx -> ospan.data / ospan.data_view(dtype[, shape]);  oseq.ring
    WriteSpan(oseq.ring, nbytes, nonblocking=False) + .close()
    ring.read(guarantee=) -> iseq.header(.tostring()/.tobytes()) .time_tag .nringlet
    iseq.read(gulp_nbytes) -> ispan.size / ispan.data / ispan.data_view(dtype)

The blocks are duck-typed against this protocol, so they accept a real bifrost Ring or this one.
This is a single-writer / multi-reader byte stream per sequence, with the data held as committed spans in the ring's space
('system' = host memory, so config 1 runs with no GPU; 'cuda' / 'cuda_host' = HIP allocations through libxeng).  Readers
that ask for `guarantee=True` apply back-pressure once `total_span` bytes are outstanding; a reader that registers late, or
that is not guaranteed and falls behind, skips ahead to the oldest span still in the ring (as a bifrost reader does).

Two implementations of the one protocol (round 4):

  * `NativeRing` (default): the bookkeeping -- committed spans, reader cursors, back-pressure, the free list of span
    allocations -- lives in libxeng (csrc/ring.hip, include/xeng.h "span rings"), as bifrost's ring is native
    (lwa352-pipeline.py:147-155); the classes here are thin handles, one foreign call per gulp and side.
  * `PyRing` (`XENG_RING=python`): the same in Python; kept as the reference implementation of the protocol, and the test
    suites run on both.

Span memory lifetime, either way: an allocation whose last user has let go is STAMPED with everything the library has enqueued
so far (include/xeng.h "stamps") and is handed out again -- or really freed -- only when that work has completed.  It is the
library's completion record that decides, not the moment at which some Python reference happened to be dropped.
"""
import collections
import ctypes
import os
import threading
import weakref

import numpy as np

from . import ffi
from .ndarray import XArray, copy_array, to_dtype, _SPACE_ID

# which implementation `Ring(...)` builds: 'native' | 'python' (tests switch it per test)
IMPLEMENTATION = os.environ.get("XENG_RING", "native")


def _stream_mask(classes):
    m = 0
    for c in classes:
        m |= ffi.STREAMS[c]
    return m


class LibraryStamps:
    """Stamps from libxeng's stream clocks (xengStampNow / Done / Wait): the source every device / pinned ring uses.
    `mask`: the union of the stream classes the ring's users have declared; it narrows a stamp only while EVERY user has
    declared (`declared` >= `users`: readers ever opened + the writer) -- one user that never did, and a stamp waits for all
    streams again (include/xeng.h xengRingDeclareStreams)."""
    mask = 0
    declared = 0
    users = 0

    def effective_mask(self):
        return self.mask if (self.mask and self.declared >= self.users) else 0

    def now(self, ptr=None, dev=None):
        """`dev`: the device the memory belongs to -- its clocks are the ones that count, whatever device is current on the thread
        that happens to drop the last reference (a helper thread, a finaliser)."""
        s = ffi.XengStamp()
        L = ffi.enqueue_lib()
        cur = ctypes.c_int(-1)
        switch = dev is not None and dev >= 0 and L.xengGetDevice(ctypes.byref(cur)) == 0 and cur.value != dev
        if switch:
            L.xengSetDevice(dev)
        try:
            ffi.check("xengStampNowFor", L.xengStampNowFor(ctypes.byref(s), ptr, self.effective_mask()))
        finally:
            if switch:
                L.xengSetDevice(cur.value)
        return s

    def done(self, s):
        """(done, waitable)"""
        d, w = ctypes.c_int(), ctypes.c_int()
        ffi.check("xengStampDone", ffi.enqueue_lib().xengStampDone(ctypes.byref(s), ctypes.byref(d), ctypes.byref(w)))
        return bool(d.value), bool(w.value)

    def wait(self, s):
        ffi.call("xengStampWait", ctypes.byref(s))


class _Allocation:
    """One span allocation of a PyRing in a device / pinned space: raw address + size, no finaliser of its own (the ring's
    free list holds these; only `_SpanOwner.__del__` and the ring's teardown ever free one, both behind the stamp)."""
    __slots__ = ("ptr", "nbytes", "space", "stamp", "keep", "dev")

    def __init__(self, ptr, nbytes, space, keep=None, dev=None):
        self.ptr, self.nbytes, self.space, self.stamp, self.keep = ptr, nbytes, space, None, keep      # keep: numpy memory behind a system-space allocation
        self.dev = dev             # device the allocation was made on (device / pinned spaces)


def _free_allocation(a, stamps):
    """Really free a span allocation -- behind its stamp.  An allocation whose stamp can never complete is leaked rather
    than freed under a kernel that may still use it."""
    if a.ptr is None:
        return
    if stamps is not None and a.stamp is not None:
        done, waitable = stamps.done(a.stamp)
        if not done:
            if not waitable:
                a.ptr = None
                return
            stamps.wait(a.stamp)
    if a.space == "system":
        a.ptr = a.keep = None      # (numpy memory)
        return
    ffi.call("xengFree", a.ptr, _SPACE_ID[a.space])
    a.ptr = None


class _SpanOwner:
    """`base` of a PyRing span array: when the last array that references it goes away the allocation is stamped and returns
    to its ring's free list.  Round 3 kept a `DeviceBuffer` (which frees in its own `__del__`) behind a weak reference to the
    ring here: when the cycle collector found such a span in a reference cycle it cleared the weak reference first and ran
    both finalisers, so the allocation was really freed (hipFree / hipHostFree) at collector time instead of being recycled
    -- the behaviour behind the round-3 churn.  Now the owner is the only finaliser, holds the ring strongly, and frees
    nothing that its stamp does not allow."""
    __slots__ = ("ring", "alloc")

    def __init__(self, ring, alloc):
        self.ring, self.alloc = ring, alloc

    def __del__(self):
        a, self.alloc = self.alloc, None
        if a is None:
            return
        try:
            self.ring._pool_put(a)
        except Exception:          # interpreter shutdown: the process is going away with its allocations
            pass


class _SequenceReader:
    """What `ring.read()` returns: the iterator over sequences plus the reader registration it stands for.  The registration is
    given back when the iteration ends, on close(), or when the object dies -- also when it was NEVER started (a generator that
    never ran has no `finally` to run: a block that registers its reader at construction and is torn down before main() would
    leave a guaranteed reader behind that blocks the ring's writer for ever)."""

    def __init__(self, gen, close_reader):
        self._gen, self._close_reader = gen, close_reader

    def __iter__(self):
        return self

    def __next__(self):
        try:
            return next(self._gen)
        except BaseException:
            self.close()
            raise

    def close(self):
        gen, self._gen = self._gen, None
        cr, self._close_reader = self._close_reader, None
        if gen is not None:
            try:
                gen.close()
            except Exception:
                pass
        if cr is not None:
            cr()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Header(bytes):
    """Sequence header bytes; the blocks call .tostring() (bifrost arrays) on it."""

    def tostring(self):
        return bytes(self)

    def tobytes(self):
        return bytes(self)


class _Chunk:
    __slots__ = ("offset", "nbytes", "data")

    def __init__(self, offset, nbytes, data):
        self.offset, self.nbytes, self.data = offset, nbytes, data


class _Sequence:
    def __init__(self, ring, index, time_tag, header, nringlet):
        self.ring, self.index = ring, index
        self.time_tag, self.nringlet = time_tag, nringlet
        self.header = _Header(header.encode() if isinstance(header, str) else bytes(header))
        self.chunks = collections.deque()     # committed spans still held, in order
        self.committed = 0        # bytes committed so far
        self.ended = False


class WriteSequence:
    """Writer-side handle of a sequence (what begin_sequence returns)."""

    def __init__(self, seq):
        self._seq = seq
        self.ring = seq.ring      # the blocks pass `oseq.ring` to WriteSpan

    def reserve(self, nbytes, nonblocking=False):
        return PyWriteSpan(self.ring, nbytes, nonblocking=nonblocking, _seq=self._seq)

    def commit_external(self, data):
        """Publish an existing array of the ring's space as the next span without copying it (a replay source:
        DummySource's test-file mode, dummy_source_block.py:207-222, re-sends the same gulps over and over).  The
        caller keeps the array unchanged while readers may still hold it."""
        assert data.space == self.ring.space, (data.space, self.ring.space)
        self.ring._wait_for_room(data.nbytes, False)
        self.ring._commit(self._seq, data.view(np.uint8), data.nbytes)

    def end(self):
        self.ring._end_sequence(self._seq)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.end()
        return False


class _SpanViews:
    def data_view(self, dtype=np.uint8, shape=None):
        v = self.data.view(dtype)
        return v.reshape(shape) if shape is not None else v


class PyWriteSpan(_SpanViews):
    """`WriteSpan(oseq.ring, nbytes, nonblocking=False)` (corr_block.py:435) -- reserved output
    memory in the ring's space; `.close()` (or leaving the `with`) commits all of it."""

    def __init__(self, ring, nbytes, nonblocking=False, _seq=None):
        self.ring = ring
        self._seq = _seq if _seq is not None else ring._open_seq
        if self._seq is None or self._seq.ended:
            raise RuntimeError("WriteSpan: no open sequence on ring %r" % ring.name)
        self.size = int(nbytes)
        ring._wait_for_room(self.size, nonblocking)
        self.data = ring._alloc_span(self.size)
        self._closed = False

    def commit(self, nbytes=None):
        if self._closed:
            return
        self._closed = True
        n = self.size if nbytes is None else int(nbytes)
        if n > 0:
            self.ring._commit(self._seq, self.data if n == self.size else self.data.byte_slice(0, n), n)

    def close(self):
        self.commit()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def WriteSpan(ring, nbytes, nonblocking=False):
    """`WriteSpan(oseq.ring, nbytes, nonblocking=False)` as the blocks spell it (corr_block.py:435, corr_acc_block.py:313)."""
    return ring._write_span(nbytes, nonblocking)


class ReadSpan(_SpanViews):
    """`skipped`: bytes of the sequence this reader never saw immediately before this span (overwritten before it got there:
    whole gulps; bifrost's `nframe_skipped`).  `offset`: byte offset of the span in its sequence.  `parts` (read_parts only):
    the gulp as one or two windows on the committed spans it lies in, in order -- `data` then gathers a copy only if asked."""
    __slots__ = ("_data", "size", "offset", "skipped", "parts")

    def __init__(self, data, size, offset=0, skipped=0, parts=None):
        self._data, self.size, self.offset, self.skipped, self.parts = data, size, offset, skipped, parts

    @property
    def data(self):
        if self._data is None and self.parts:
            if len(self.parts) == 1:
                self._data = self.parts[0]
            else:
                out = XArray(shape=(self.size,), dtype=np.uint8, space=self.parts[0].space)
                pos = 0
                for p in self.parts:
                    copy_array(out.byte_slice(pos, p.nbytes), p)
                    pos += p.nbytes
                self._data = out
        return self._data


class ReadSequence:
    def __init__(self, seq, reader):
        self._seq, self._reader = seq, reader
        self.header = seq.header
        self.time_tag, self.nringlet = seq.time_tag, seq.nringlet
        self.ring = seq.ring

    def read_parts(self, gulp_nbytes):
        """read(), but a gulp that lies in two committed spans is handed over as two windows (`ispan.parts`) instead of a
        gathered copy -- for a consumer that takes its gulp in two parts (Beamform: xengBeamformRunParts)."""
        return self.read(gulp_nbytes, parts=True)

    def read(self, gulp_nbytes, parts=False):
        """Yield full gulps as they become available; a short final gulp (size < gulp_nbytes) is
        yielded once when the sequence ends, as bifrost does (the blocks skip it:
        corr_block.py:389-391)."""
        ring, seq, rd = self.ring, self._seq, self._reader
        gulp_nbytes = int(gulp_nbytes)
        while True:
            with ring._cond:
                skipped = 0
                while True:
                    # data that was overwritten before this reader got to it (it registered late, or nobody held the data
                    # for it): skip ahead by whole gulps to the oldest span still there, as a bifrost reader does
                    lo = seq.chunks[0].offset if seq.chunks else seq.committed
                    if rd.offset < lo:
                        sk = -(-(lo - rd.offset) // gulp_nbytes) * gulp_nbytes
                        rd.offset += sk
                        skipped += sk
                    if seq.committed - rd.offset >= gulp_nbytes or seq.ended:
                        break
                    ring._cond.wait(0.5)
                avail = seq.committed - rd.offset
                n = min(avail, gulp_nbytes)
                if n <= 0:
                    return
                if parts:
                    pieces = ring._pieces(seq, rd.offset, n)
                    data = None if len(pieces) <= 2 else ring._assemble(seq, rd.offset, n)
                    pieces = pieces if data is None else [data]
                else:
                    data, pieces = ring._assemble(seq, rd.offset, n), None
                offset = rd.offset
            yield ReadSpan(data, n, offset, skipped, pieces)
            with ring._cond:
                rd.offset += n
                ring._gc()
                ring._cond.notify_all()
            if n < gulp_nbytes:
                return


class _Reader:
    def __init__(self, guarantee):
        self.guarantee = guarantee
        self.seq_index = 0
        self.offset = 0


class _Writer:
    def __init__(self, ring):
        self.ring = ring

    def begin_sequence(self, time_tag=0, header="", nringlet=1, name=None):
        return self.ring._begin_sequence(time_tag, header, nringlet)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.ring._end_writing()
        return False


class PyRing:
    # Every committed span is its own allocation, kept alive by reference counting: a reader that keeps
    # `ispan.data` may go on reading it after the ring has recycled the span (a bifrost ring is one circular
    # buffer and cannot promise that).  Corr uses this to let the X-engine read gulps in place.
    span_memory_outlives_release = True

    def __init__(self, name="", space="system", core=None):
        self.name, self.space = name, space
        self._cond = threading.Condition()
        self._seqs = []
        self._open_seq = None
        self._writing_ended = False
        self._readers = []
        self._capacity = 0
        self._live_bytes = 0
        self._gc_seq = 0           # sequences before this index hold no data any more
        self._pool = {}            # nbytes -> deque of released allocations, oldest first (device / pinned spaces)
        self._pool_bytes = 0
        self._pool_lock = threading.RLock()     # (re-entrant: a span released by the garbage collector inside _alloc_span puts itself back)
        self._stamps = LibraryStamps() if space != "system" else None
        self._dead = False
        self._owned_bytes = 0
        self.counters = {"alloc": 0, "free": 0, "reuse": 0, "stamp_wait": 0}

    def set_stamp_source(self, src):
        """Tests: completion tickets of a fake backend instead of the library's stream clocks -- an object with now() ->
        stamp, done(stamp) -> (done, waitable), wait(stamp).  Gives a system-space ring a free list too."""
        self._stamps = src

    def declare_streams(self, *classes):
        """The blocks on this ring name the library streams that touch its spans ('xgpu', 'map', 'beam', 'copy', 'consumer';
        calls accumulate): a released span then waits for those only (include/xeng.h xengRingDeclareStreams)."""
        if isinstance(self._stamps, LibraryStamps):
            self._stamps.mask |= _stream_mask(classes)
            self._stamps.declared += 1

    def _count_user(self):
        if isinstance(self._stamps, LibraryStamps):
            self._stamps.users += 1

    def stamp_classes(self):
        """(classes a span released now would wait for -- 31 = all --, declarations made, users seen)"""
        st = self._stamps
        if not isinstance(st, LibraryStamps):
            return 31, 0, 0
        return (st.effective_mask() or 31), st.declared, st.users

    def __del__(self):
        try:
            with self._pool_lock:
                self._dead = True
                pool, self._pool = self._pool, {}
            for dq in pool.values():
                for a in dq:
                    _free_allocation(a, self._stamps)
        except Exception:
            pass

    # ------------------------------------------------------------------ span memory
    def _write_span(self, nbytes, nonblocking):
        return PyWriteSpan(self, nbytes, nonblocking)

    def _alloc_span(self, nbytes):
        """Span memory in the ring's space.  'system' (without a stamp source): fresh zeroed numpy memory.  Otherwise the
        OLDEST released allocation of that size whose stamp is complete (contents: whatever its last user left, as in a
        circular bifrost ring); if every released one is still busy, wait for the oldest; with none released, a fresh
        zero-filled allocation."""
        if self._stamps is None:
            return XArray(shape=(nbytes,), dtype=np.uint8, space="system")
        a = None
        with self._pool_lock:
            dq = self._pool.get(nbytes)
            cand = dq.popleft() if dq else None        # the oldest release first: its stamp is the most likely to be complete
            if cand is not None:
                self._pool_bytes -= nbytes
        if cand is not None:
            done, waitable = self._stamps.done(cand.stamp) if cand.stamp is not None else (True, True)
            # still busy: a fresh allocation while the ring owns little (a deeper free list costs memory once, a wait costs
            # every gulp), else wait for it (outside the lock: kernels of other blocks, enqueued before the release)
            grow = (not done and isinstance(self._stamps, LibraryStamps)
                    and self._owned_bytes + nbytes <= 8 * max(self._capacity, 2 * nbytes))
            if not done and waitable and not grow:
                self.counters["stamp_wait"] += 1
                self._stamps.wait(cand.stamp)
                done = True
            if done:
                a = cand
                self.counters["reuse"] += 1
            else:
                with self._pool_lock:
                    if waitable:
                        self._pool[nbytes].appendleft(cand)        # still the next one to be reissued
                    else:
                        self._pool[nbytes].append(cand)            # waits for a launch nobody has enqueued: try the others first
                    self._pool_bytes += nbytes
        if a is None:
            self.counters["alloc"] += 1
            self._owned_bytes += nbytes
            if self.space == "system":
                keep = np.zeros(max(nbytes, 1), dtype=np.uint8)
                a = _Allocation(keep.ctypes.data, nbytes, "system", keep)
            else:
                buf = ffi.DeviceBuffer(max(nbytes, 1), _SPACE_ID[self.space])
                ffi.call("xengMemset", buf.ptr, 0, max(nbytes, 1))
                cur = ctypes.c_int(-1)
                a = _Allocation(buf.ptr, nbytes, self.space, dev=cur.value if ffi.enqueue_lib().xengGetDevice(ctypes.byref(cur)) == 0 else None)
                buf.ptr = None                    # (the allocation is the ring's now: DeviceBuffer.__del__ must never free it)
        return XArray(shape=(nbytes,), dtype=np.uint8, space=self.space, _ptr=a.ptr, _base=_SpanOwner(self, a))

    def _pool_put(self, a):
        """The last user of a span allocation has let go: stamp it and keep it for reuse (up to the ring's capacity in bytes;
        beyond that it is really freed, behind its stamp)."""
        if self._stamps is not None:
            a.stamp = self._stamps.now(a.ptr, a.dev) if isinstance(self._stamps, LibraryStamps) else self._stamps.now()
        with self._pool_lock:
            # (one bound for what the ring may own: really freed only past it -- steady state neither allocates nor frees)
            if not self._dead and (self._owned_bytes <= 8 * max(self._capacity, 2 * a.nbytes) or self._stamps is None or not isinstance(self._stamps, LibraryStamps)):
                self._pool.setdefault(a.nbytes, collections.deque()).append(a)
                self._pool_bytes += a.nbytes
                return
        self.counters["free"] += 1
        self._owned_bytes -= a.nbytes
        _free_allocation(a, self._stamps)

    # ------------------------------------------------------------------ writer side
    def resize(self, contig_bytes, total_span=None, nringlet=1):
        want = int(total_span) if total_span else 4 * int(contig_bytes)
        with self._cond:
            self._capacity = max(self._capacity, want, int(contig_bytes))

    def begin_writing(self):
        return _Writer(self)

    def _begin_sequence(self, time_tag, header, nringlet):
        with self._cond:
            if self._open_seq is not None and not self._open_seq.ended:
                self._open_seq.ended = True
            if not self._seqs:
                self._count_user()          # (the writer)
            seq = _Sequence(self, len(self._seqs), time_tag, header, nringlet)
            self._seqs.append(seq)
            self._open_seq = seq
            self._cond.notify_all()
        return WriteSequence(seq)

    def _end_sequence(self, seq):
        with self._cond:
            seq.ended = True
            if self._open_seq is seq:
                self._open_seq = None
            self._cond.notify_all()

    def _end_writing(self):
        with self._cond:
            if self._open_seq is not None:
                self._open_seq.ended = True
                self._open_seq = None
            self._writing_ended = True
            self._cond.notify_all()

    def _wait_for_room(self, nbytes, nonblocking):
        with self._cond:
            if self._capacity == 0:
                self._capacity = 4 * nbytes
            while self._live_bytes + nbytes > max(self._capacity, nbytes):
                self._gc()
                if self._live_bytes + nbytes <= max(self._capacity, nbytes):
                    break
                if not any(r.guarantee for r in self._readers):
                    self._drop_oldest()           # nobody applies back-pressure: overwrite, like bifrost
                    continue
                if nonblocking:
                    raise BlockingIOError("ring %r full" % self.name)
                self._cond.wait(0.5)

    def _commit(self, seq, data, nbytes):
        with self._cond:
            seq.chunks.append(_Chunk(seq.committed, nbytes, data))
            seq.committed += nbytes
            self._live_bytes += nbytes
            self._cond.notify_all()

    # ------------------------------------------------------------------ bookkeeping (hold _cond)
    def _gc(self):
        """Free committed spans every registered reader has moved past.  Released chunks leave the head of their
        sequence's list, so the cost per call is the number of chunks released, not the length of the sequence.
        (Runs once per gulp and reader: kept short.)"""
        readers = self._readers
        if not readers:
            return
        if len(readers) == 1:
            lo_seq, lo_off = readers[0].seq_index, readers[0].offset
        else:
            lo_seq, lo_off = min([(r.seq_index, r.offset) for r in readers])       # the slowest reader
        seqs, g, freed = self._seqs, self._gc_seq, 0
        while g < lo_seq:                              # sequences every reader has left are empty for good
            ch = seqs[g].chunks
            while ch:
                c = ch.popleft()
                if c.data is not None:
                    c.data = None
                    freed += c.nbytes
            g += 1
        self._gc_seq = g
        if lo_seq < len(seqs):
            ch = seqs[lo_seq].chunks
            while ch and ch[0].offset + ch[0].nbytes <= lo_off:
                c = ch.popleft()
                if c.data is not None:
                    c.data = None
                    freed += c.nbytes
        self._live_bytes -= freed

    def _drop_oldest(self):
        for seq in self._seqs[self._gc_seq:]:
            if seq.chunks:
                ch = seq.chunks.popleft()
                if ch.data is not None:
                    ch.data = None
                    self._live_bytes -= ch.nbytes
                return
        self._live_bytes = 0

    def _pieces(self, seq, offset, nbytes):
        """Bytes [offset, offset+nbytes) of a sequence as windows on the committed spans they lie in, in order."""
        pieces = []
        for ch in seq.chunks:
            if ch.offset >= offset + nbytes:
                break
            lo, hi = max(offset, ch.offset), min(offset + nbytes, ch.offset + ch.nbytes)
            if lo < hi:
                pieces.append(ch.data if (lo == ch.offset and hi == ch.offset + ch.nbytes) else ch.data.byte_slice(lo - ch.offset, hi - lo))
        return pieces

    def _assemble(self, seq, offset, nbytes):
        """Bytes [offset, offset+nbytes) of a sequence as one array: a zero-copy window when they
        lie inside one committed span, else a gathered copy in the ring's space."""
        c0 = seq.chunks[0]
        if c0.offset == offset and c0.nbytes == nbytes:       # the usual case: the gulp is the oldest span, whole
            return c0.data
        pieces = self._pieces(seq, offset, nbytes)
        if len(pieces) == 1:
            return pieces[0]
        out = XArray(shape=(nbytes,), dtype=np.uint8, space=self.space)
        pos = 0
        for p in pieces:
            copy_array(out.byte_slice(pos, p.nbytes), p)
            pos += p.nbytes
        return out

    # ------------------------------------------------------------------ reader side
    def read(self, guarantee=True):
        """Iterate over sequences.  The reader is registered when read() is called (not at the first
        next()), so data written between the call and the first iteration is kept for it.  A reader that registers late
        starts at the oldest sequence that still holds data or is still being written -- never at data that is gone."""
        rd = _Reader(guarantee)
        self._count_user()
        with self._cond:
            rd.seq_index = len(self._seqs)
            for seq in self._seqs[self._gc_seq:]:
                if seq.chunks or not seq.ended:
                    rd.seq_index = seq.index
                    break
            self._readers.append(rd)
        return _SequenceReader(self._read_sequences(rd), lambda: self._close_reader(rd))

    def _close_reader(self, rd):
        with self._cond:
            if rd in self._readers:
                self._readers.remove(rd)
                self._gc()
            self._cond.notify_all()

    def _read_sequences(self, rd):
        while True:
            with self._cond:
                while len(self._seqs) <= rd.seq_index and not self._writing_ended:
                    self._cond.wait(0.5)
                if len(self._seqs) <= rd.seq_index:
                    return
                seq = self._seqs[rd.seq_index]
                rd.offset = 0
            yield ReadSequence(seq, rd)
            with self._cond:
                rd.seq_index += 1
                rd.offset = 0
                self._gc()
                self._cond.notify_all()


# ====================================================================== the native ring: thin handles on csrc/ring.hip
_WOULD_BLOCK, _END = ffi.STATUS_WOULD_BLOCK, ffi.STATUS_END_OF_DATA
_xf = None


def _xfast():
    """The CPython extension that binds the per-gulp ring calls directly (csrc/pyext/xfast.cpp): it asks without waiting first
    and gives the interpreter lock up only for a call that has to sleep."""
    global _xf
    if _xf is None:
        try:
            from . import _xfast as m
        except ImportError as e:
            raise ImportError("the _xfast extension is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C caltech-bifrost-dsp_amd/csrc): %s" % e)
        _xf = m
    return _xf


class _NWriteSpan(_SpanViews):
    __slots__ = ("ring", "_seq_id", "size", "_span", "data", "_closed")

    def __init__(self, ring, seq_id, nbytes, nonblocking):
        self.ring, self._seq_id, self.size = ring, seq_id, int(nbytes)
        ptr, ref, self._span = ring._x.ring_reserve(ring, ring._h, seq_id, self.size, bool(nonblocking))
        self.data = XArray.window(ptr, self.size, ring.space, ref)
        self._closed = False

    def commit(self, nbytes=None):
        if self._closed:
            return
        self._closed = True
        n = self.size if nbytes is None else int(nbytes)
        if n > 0:
            self.ring._x.ring_commit(self.ring._h, self._seq_id, self._span, n)

    def close(self):
        self.commit()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class _NWriteSequence:
    def __init__(self, ring, seq_id):
        self.ring, self._seq_id = ring, seq_id

    def reserve(self, nbytes, nonblocking=False):
        return _NWriteSpan(self.ring, self._seq_id, nbytes, nonblocking)

    def commit_external(self, data):
        ring = self.ring
        assert data.space == ring.space, (data.space, ring.space)
        ring._external[data.ptr] = data          # (the ring hands out windows on the caller's memory: kept alive with the ring)
        ring._x.ring_commit_external(ring._h, self._seq_id, data.ptr, data.nbytes)

    def end(self):
        self.ring._enq.xengRingEndSequence(self.ring._h, self._seq_id)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.end()
        return False


class _NWriter:
    def __init__(self, ring):
        self.ring = ring

    def begin_sequence(self, time_tag=0, header="", nringlet=1, name=None):
        ring = self.ring
        hdr = header.encode() if isinstance(header, str) else bytes(header)
        seq = ctypes.c_longlong()
        ffi.check("xengRingBeginSequence", ring._enq.xengRingBeginSequence(ring._h, int(time_tag), hdr, len(hdr), int(nringlet), ctypes.byref(seq)))
        return _NWriteSequence(ring, seq.value)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.ring._enq.xengRingEndWriting(self.ring._h)
        return False


class _NReadSequence:
    def __init__(self, ring, rid, header, time_tag, nringlet):
        self.ring, self._rid = ring, rid
        self.header, self.time_tag, self.nringlet = header, time_tag, nringlet

    def read(self, gulp_nbytes):
        """Full gulps as they become available; a short final gulp once when the sequence ends (corr_block.py:389-391)."""
        ring, rid = self.ring, self._rid
        h, acquire, space, window = ring._h, ring._x.ring_acquire, ring.space, XArray.window
        gulp_nbytes = int(gulp_nbytes)
        advance = offset = 0
        while True:
            got = acquire(ring, h, rid, advance, gulp_nbytes)       # (one native call per gulp: moves on, collects, waits)
            if got is None:
                return
            ptr, size, ref, skipped = got
            offset += advance + skipped
            yield ReadSpan(window(ptr, size, space, ref), size, offset, skipped)
            del ref, got
            advance = size
            if size < gulp_nbytes:
                return


    def read_parts(self, gulp_nbytes):
        """read(), but a gulp that lies in two committed spans is handed over as two windows (`ispan.parts`) instead of a
        gathered copy (xengRingAcquireParts)."""
        ring, rid = self.ring, self._rid
        h, acquire, space, window = ring._h, ring._x.ring_acquire_parts, ring.space, XArray.window
        gulp_nbytes = int(gulp_nbytes)
        advance = offset = 0
        while True:
            got = acquire(ring, h, rid, advance, gulp_nbytes)
            if got is None:
                return
            skipped = got[0]
            parts = [window(ptr, n, space, ref) for ptr, n, ref in got[1:]]
            size = sum(p.nbytes for p in parts)
            offset += advance + skipped
            yield ReadSpan(None, size, offset, skipped, parts)
            del parts, got
            advance = size
            if size < gulp_nbytes:
                return


class _ReaderCount:
    def __init__(self, ring):
        self._ring = ring

    def __len__(self):
        return self._ring.info()["nreaders"]


class NativeRing:
    span_memory_outlives_release = True

    def __init__(self, name="", space="system", core=None):
        self.name, self.space = name, space
        self._lib, self._enq, self._x = ffi.lib(), ffi.enqueue_lib(), _xfast()
        h = ctypes.c_void_p()
        ffi.call("xengRingCreate", ctypes.byref(h), name.encode(), _SPACE_ID[space])
        self._h = h.value
        self._external = {}
        self._hooks = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.xengRingDestroy(h)
            except Exception:
                pass

    @property
    def _readers(self):
        return _ReaderCount(self)

    def info(self):
        cap, live, pool, nrd, nseq = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_int(), ctypes.c_longlong()
        cnt = (ctypes.c_ulonglong * 5)()
        ffi.check("xengRingGetInfo", self._enq.xengRingGetInfo(self._h, ctypes.byref(cap), ctypes.byref(live), ctypes.byref(pool), ctypes.byref(nrd),
                                                               ctypes.byref(nseq), cnt))
        return {"capacity": cap.value, "live_bytes": live.value, "pool_bytes": pool.value, "nreaders": nrd.value, "nseq": nseq.value,
                "alloc": cnt[0], "free": cnt[1], "reuse": cnt[2], "stamp_wait": cnt[3], "skipped_bytes": cnt[4]}

    @property
    def counters(self):
        return self.info()

    def set_stamp_source(self, src):
        """Tests: completion tickets of a fake backend instead of the library's stream clocks (see PyRing.set_stamp_source).
        A stamp of the source must be an integer."""
        def now(user, out):
            out[0] = int(src.now())

        def done(user, st):
            return int(bool(src.done(st[0])[0]))

        def wait(user, st):
            src.wait(st[0])
        self._hooks = (ffi.STAMP_NOW_FN(now), ffi.STAMP_DONE_FN(done), ffi.STAMP_WAIT_FN(wait))       # (kept alive with the ring)
        ffi.call("xengRingSetStampHooks", self._h, self._hooks[0], self._hooks[1], self._hooks[2], None)

    def declare_streams(self, *classes):
        """The blocks on this ring name the library streams that touch its spans ('xgpu', 'map', 'beam', 'copy', 'consumer';
        calls accumulate): a released span then waits for those only (include/xeng.h xengRingDeclareStreams)."""
        ffi.call("xengRingDeclareStreams", self._h, _stream_mask(classes))

    def stamp_classes(self):
        """(classes a span released now would wait for -- 31 = all --, declarations made, users seen)"""
        c, d, u = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
        ffi.call("xengRingGetStampClasses", self._h, ctypes.byref(c), ctypes.byref(d), ctypes.byref(u))
        return c.value, d.value, u.value

    def set_recycle(self, on=True):
        """System-space ring: recycle released span memory (no zero fill per span), as the device / pinned rings always do."""
        ffi.call("xengRingSetRecycle", self._h, int(bool(on)))

    def resize(self, contig_bytes, total_span=None, nringlet=1):
        ffi.check("xengRingResize", self._enq.xengRingResize(self._h, int(contig_bytes), int(total_span) if total_span else 0))

    def begin_writing(self):
        return _NWriter(self)

    def _write_span(self, nbytes, nonblocking):
        return _NWriteSpan(self, -1, nbytes, nonblocking)       # (-1: the open sequence)

    def read(self, guarantee=True):
        """Iterate over sequences.  The reader is registered when read() is called, so data written between the call and
        the first iteration is kept for it."""
        rid = ctypes.c_int()
        # (registered without giving up the interpreter lock: a reader thread that has just been started is registered before
        # the thread that started it runs on, as with the Python ring)
        ffi.check("xengRingOpenReader", self._enq.xengRingOpenReader(self._h, int(bool(guarantee)), ctypes.byref(rid)))
        rid = rid.value
        return _SequenceReader(self._read_sequences(rid), lambda: self._close_reader(rid))

    def _close_reader(self, rid):
        if self._h:
            self._enq.xengRingCloseReader(self._h, rid)

    def _read_sequences(self, rid):
        nxt = self._x.ring_next_sequence
        while True:
            got = nxt(self._h, rid)
            if got is None:
                return
            yield _NReadSequence(self, rid, _Header(got[0]), got[1], got[2])


def Ring(name="", space="system", core=None):
    """`Ring(name=, space=)` as lwa352-pipeline.py:147-155 creates them: the native ring unless XENG_RING=python."""
    if IMPLEMENTATION == "python":
        return PyRing(name=name, space=space, core=core)
    return NativeRing(name=name, space=space, core=core)


__all__ = ["Ring", "NativeRing", "PyRing", "WriteSpan", "to_dtype"]
