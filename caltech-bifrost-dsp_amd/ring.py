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
This implementation is a single-writer / multi-reader byte stream per sequence, with the data
held as committed spans in the ring's space ('system' = numpy memory, so config 1 runs with no
GPU; 'cuda' / 'cuda_host' = HIP allocations through libxeng).  Readers that ask for
`guarantee=True` apply back-pressure once `total_span` bytes are outstanding.
"""
import collections
import threading
import weakref

import numpy as np

from . import ffi
from .ndarray import XArray, copy_array, to_dtype, _SPACE_ID


class _PooledBuffer:
    """Owner of one span allocation in a device / pinned space.  When the last array that references it goes away
    the allocation returns to its ring's free list instead of hipFree (which synchronises the whole device and
    would stall the other blocks' streams: Corr alone turns over a 191 MB span per integration)."""

    def __init__(self, ring, buf):
        self._ring = weakref.ref(ring)
        self.buf = buf
        self.ptr = buf.ptr
        self.nbytes = buf.nbytes

    def __del__(self):
        ring = self._ring()
        buf, self.buf = self.buf, None
        if buf is None:
            return
        try:
            if ring is not None and ring._pool_put(buf):
                return
            # really freed (the ring is gone or its free list is full): rare, and never under a kernel -- whoever dropped
            # the last reference had waited for its own GPU work, but the device is drained first all the same
            ffi.call("xengDeviceSynchronize")
            buf.free()
        except Exception:          # interpreter shutdown: the process is going away with its allocations
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
        return WriteSpan(self.ring, nbytes, nonblocking=nonblocking, _seq=self._seq)

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


class WriteSpan:
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

    def data_view(self, dtype=np.uint8, shape=None):
        v = self.data.view(dtype)
        return v.reshape(shape) if shape is not None else v

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


class ReadSpan:
    def __init__(self, data, size):
        self.data, self.size = data, size

    def data_view(self, dtype=np.uint8, shape=None):
        v = self.data.view(dtype)
        return v.reshape(shape) if shape is not None else v


class ReadSequence:
    def __init__(self, seq, reader):
        self._seq, self._reader = seq, reader
        self.header = seq.header
        self.time_tag, self.nringlet = seq.time_tag, seq.nringlet
        self.ring = seq.ring

    def read(self, gulp_nbytes):
        """Yield full gulps as they become available; a short final gulp (size < gulp_nbytes) is
        yielded once when the sequence ends, as bifrost does (the blocks skip it:
        corr_block.py:389-391)."""
        ring, seq, rd = self.ring, self._seq, self._reader
        gulp_nbytes = int(gulp_nbytes)
        while True:
            with ring._cond:
                while seq.committed - rd.offset < gulp_nbytes and not seq.ended:
                    ring._cond.wait(0.5)
                avail = seq.committed - rd.offset
                n = min(avail, gulp_nbytes)
                if n <= 0:
                    return
                data = ring._assemble(seq, rd.offset, n)
            yield ReadSpan(data, n)
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


class Ring:
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
        self._pool = {}            # nbytes -> [free allocations] (device / pinned spaces)
        self._pool_bytes = 0
        self._pool_lock = threading.RLock()     # (re-entrant: a span released by the garbage collector inside _alloc_span puts itself back)

    # ------------------------------------------------------------------ span memory
    def _alloc_span(self, nbytes):
        """Span memory in the ring's space.  'system': fresh zeroed numpy memory.  Device / pinned spaces: recycled
        from the ring's free list when a span of that size has been released (contents then are whatever the last
        user left, as in a circular bifrost ring); a first-time allocation is zero-filled."""
        if self.space == "system":
            return XArray(shape=(nbytes,), dtype=np.uint8, space="system")
        with self._pool_lock:
            lst = self._pool.get(nbytes)
            buf = lst.pop() if lst else None
            if buf is not None:
                self._pool_bytes -= nbytes
        if buf is None:
            buf = ffi.DeviceBuffer(max(nbytes, 1), _SPACE_ID[self.space])
            ffi.call("xengMemset", buf.ptr, 0, max(nbytes, 1))
        owner = _PooledBuffer(self, buf)
        return XArray(shape=(nbytes,), dtype=np.uint8, space=self.space, _ptr=buf.ptr, _base=owner)

    def _pool_put(self, buf):
        """Keep a released allocation for reuse (up to the ring's capacity in bytes); False = caller frees it."""
        with self._pool_lock:
            if self._pool_bytes + buf.nbytes > max(self._capacity, 2 * buf.nbytes):
                return False
            self._pool.setdefault(buf.nbytes, []).append(buf)
            self._pool_bytes += buf.nbytes
        return True

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

    def _assemble(self, seq, offset, nbytes):
        """Bytes [offset, offset+nbytes) of a sequence as one array: a zero-copy window when they
        lie inside one committed span, else a gathered copy in the ring's space."""
        if not seq.chunks or seq.chunks[0].offset > offset:
            raise RuntimeError("ring %r: data at %d was overwritten before it was read" % (self.name, offset))
        c0 = seq.chunks[0]
        if c0.offset == offset and c0.nbytes == nbytes:       # the usual case: the gulp is the oldest span, whole
            return c0.data
        pieces = []
        for ch in seq.chunks:
            if ch.offset >= offset + nbytes:
                break
            lo, hi = max(offset, ch.offset), min(offset + nbytes, ch.offset + ch.nbytes)
            if lo < hi:
                pieces.append(ch.data if (lo == ch.offset and hi == ch.offset + ch.nbytes) else ch.data.byte_slice(lo - ch.offset, hi - lo))
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
        next()), so data written between the call and the first iteration is kept for it."""
        rd = _Reader(guarantee)
        with self._cond:
            self._readers.append(rd)
        return self._read_sequences(rd)

    def _read_sequences(self, rd):
        try:
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
        finally:
            with self._cond:
                if rd in self._readers:
                    self._readers.remove(rd)
                self._cond.notify_all()


__all__ = ["Ring", "WriteSpan", "to_dtype"]
