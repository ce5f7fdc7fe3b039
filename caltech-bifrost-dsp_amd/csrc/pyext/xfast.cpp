// _xfast: the per-gulp calls of the blocks as a CPython extension (no ctypes marshalling on the hot path).
//
// The blocks are Python, one thread each under one interpreter lock (lwa352-pipeline.py:296-302), so what a gulp costs in
// bytecode and call overhead is what the whole pipeline pays once per gulp and block.  ctypes spends 1.5-3 us per foreign call
// on argument conversion -- as much as the native ring call it makes takes in all (profiles/r04/ring_call_cost.txt).  This
// module binds the same C ABI (include/xeng.h) directly:
//   * the span-ring calls a block makes per gulp (reserve / commit / acquire / next_sequence), which try without waiting
//     first and give the interpreter lock up only for a call that really has to sleep;
//   * SpanRef: a reference on a span's memory that is given back when the object dies;
//   * the enqueue-only compute calls and completion queries (ffi.ENQUEUE_ONLY), lock kept: a few microseconds of host work.
// Everything else (set-up, copies, waits) stays on ctypes (ffi.py).  Pure binding: no logic of its own beyond "ask first,
// then wait without the lock".
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <time.h>

#include <deque>
#include <vector>

#include "../../../include/xeng.h"

namespace {

PyObject* raise_xeng(const char* fn, int rc) {
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        PyErr_Format(PyExc_BlockingIOError, "%s: %s", fn, xengGetLastError());
    } else {
        PyErr_Format(PyExc_RuntimeError, "%s returned %d: %s", fn, rc, xengGetLastError());
    }
    return nullptr;
}

// ---------------------------------------------------------------- SpanRef
struct SpanRef {
    PyObject_HEAD
    PyObject* ring;        // keeps the ring's Python object (and so the native ring) alive
    long long handle;
};

void SpanRef_dealloc(SpanRef* self) {
    if (self->handle) {
        (void)xengRingSpanRelease(self->handle);
        self->handle = 0;
    }
    Py_CLEAR(self->ring);
    Py_TYPE(self)->tp_free((PyObject*)self);
}

PyTypeObject SpanRefType = {PyVarObject_HEAD_INIT(nullptr, 0)};

PyObject* make_spanref(PyObject* ring, long long handle) {
    SpanRef* s = PyObject_New(SpanRef, &SpanRefType);
    if (!s) {
        (void)xengRingSpanRelease(handle);
        return nullptr;
    }
    Py_INCREF(ring);
    s->ring = ring;
    s->handle = handle;
    return (PyObject*)s;
}

// ---------------------------------------------------------------- rings
// ring_reserve(ring_obj, h, seq, nbytes, nonblocking) -> (ptr, SpanRef, handle)
PyObject* ring_reserve(PyObject*, PyObject* args) {
    PyObject* ring;
    unsigned long long h;
    long long seq;
    Py_ssize_t nbytes;
    int nonblocking;
    if (!PyArg_ParseTuple(args, "OKLnp", &ring, &h, &seq, &nbytes, &nonblocking)) return nullptr;
    void* data = nullptr;
    long long span = 0;
    int rc = xengRingReserve((xengRing*)h, seq, (size_t)nbytes, nonblocking, 0, &data, &span);
    if (rc == XENG_STATUS_WOULD_BLOCK && !nonblocking) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengRingReserve((xengRing*)h, seq, (size_t)nbytes, 0, 1, &data, &span);
        Py_END_ALLOW_THREADS
    }
    if (rc) return raise_xeng("xengRingReserve", rc);
    PyObject* ref = make_spanref(ring, span);
    if (!ref) return nullptr;
    return Py_BuildValue("(KNL)", (unsigned long long)(uintptr_t)data, ref, span);
}

PyObject* ring_commit(PyObject*, PyObject* args) {
    unsigned long long h;
    long long seq, span;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "KLLn", &h, &seq, &span, &n)) return nullptr;
    int rc = xengRingCommit((xengRing*)h, seq, span, (size_t)n);
    if (rc) return raise_xeng("xengRingCommit", rc);
    Py_RETURN_NONE;
}

PyObject* ring_commit_external(PyObject*, PyObject* args) {
    unsigned long long h, ptr;
    long long seq;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "KLKn", &h, &seq, &ptr, &n)) return nullptr;
    int rc = xengRingCommitExternal((xengRing*)h, seq, (void*)(uintptr_t)ptr, (size_t)n, 0);
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengRingCommitExternal((xengRing*)h, seq, (void*)(uintptr_t)ptr, (size_t)n, 1);
        Py_END_ALLOW_THREADS
    }
    if (rc) return raise_xeng("xengRingCommitExternal", rc);
    Py_RETURN_NONE;
}

// ring_next_sequence(h, reader) -> None (no more sequences) | (header bytes, time_tag, nringlet)
PyObject* ring_next_sequence(PyObject*, PyObject* args) {
    unsigned long long h;
    int reader;
    if (!PyArg_ParseTuple(args, "Ki", &h, &reader)) return nullptr;
    long long seq = 0, tag = 0;
    int nringlet = 1;
    const void* hdr = nullptr;
    size_t hlen = 0;
    int rc = xengRingNextSequence((xengRing*)h, reader, 0, &seq, &tag, &nringlet, &hdr, &hlen);
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengRingNextSequence((xengRing*)h, reader, 1, &seq, &tag, &nringlet, &hdr, &hlen);
        Py_END_ALLOW_THREADS
    }
    if (rc == XENG_STATUS_END_OF_DATA) Py_RETURN_NONE;
    if (rc) return raise_xeng("xengRingNextSequence", rc);
    return Py_BuildValue("(y#Li)", (const char*)hdr, (Py_ssize_t)hlen, tag, nringlet);
}

// ring_acquire(ring_obj, h, reader, advance, gulp) -> None (the sequence is over) | (ptr, nbytes, SpanRef, skipped)
PyObject* ring_acquire(PyObject*, PyObject* args) {
    PyObject* ring;
    unsigned long long h;
    int reader;
    Py_ssize_t advance, gulp;
    if (!PyArg_ParseTuple(args, "OKinn", &ring, &h, &reader, &advance, &gulp)) return nullptr;
    void* data = nullptr;
    size_t n = 0, skipped = 0, sk2 = 0;
    long long span = 0;
    int rc = xengRingAcquire((xengRing*)h, reader, (size_t)advance, (size_t)gulp, 0, &data, &n, &span, &skipped);
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        Py_BEGIN_ALLOW_THREADS          // (the first call has moved the cursor on already)
        rc = xengRingAcquire((xengRing*)h, reader, 0, (size_t)gulp, 1, &data, &n, &span, &sk2);
        Py_END_ALLOW_THREADS
        skipped += sk2;
    }
    if (rc == XENG_STATUS_END_OF_DATA) Py_RETURN_NONE;
    if (rc) return raise_xeng("xengRingAcquire", rc);
    PyObject* ref = make_spanref(ring, span);
    if (!ref) return nullptr;
    return Py_BuildValue("(KnNn)", (unsigned long long)(uintptr_t)data, (Py_ssize_t)n, ref, (Py_ssize_t)skipped);
}

// ring_acquire_parts(ring_obj, h, reader, advance, gulp) -> None | (skipped, (ptr, nbytes, SpanRef), ...)  -- one or two parts
PyObject* ring_acquire_parts(PyObject*, PyObject* args) {
    PyObject* ring;
    unsigned long long h;
    int reader;
    Py_ssize_t advance, gulp;
    if (!PyArg_ParseTuple(args, "OKinn", &ring, &h, &reader, &advance, &gulp)) return nullptr;
    void* data[2] = {nullptr, nullptr};
    size_t n[2] = {0, 0}, skipped = 0, sk2 = 0;
    long long span[2] = {0, 0};
    int nparts = 0;
    int rc = xengRingAcquireParts((xengRing*)h, reader, (size_t)advance, (size_t)gulp, 0, data, n, span, &nparts, &skipped);
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengRingAcquireParts((xengRing*)h, reader, 0, (size_t)gulp, 1, data, n, span, &nparts, &sk2);
        Py_END_ALLOW_THREADS
        skipped += sk2;
    }
    if (rc == XENG_STATUS_END_OF_DATA) Py_RETURN_NONE;
    if (rc) return raise_xeng("xengRingAcquireParts", rc);
    PyObject* out = PyTuple_New(1 + nparts);
    if (!out) { for (int k = 0; k < nparts; k++) (void)xengRingSpanRelease(span[k]); return nullptr; }
    PyTuple_SET_ITEM(out, 0, PyLong_FromSsize_t((Py_ssize_t)skipped));
    for (int k = 0; k < nparts; k++) {
        PyObject* ref = make_spanref(ring, span[k]);
        if (!ref) { for (int q = k + 1; q < nparts; q++) (void)xengRingSpanRelease(span[q]); Py_DECREF(out); return nullptr; }
        PyTuple_SET_ITEM(out, 1 + k, Py_BuildValue("(KnN)", (unsigned long long)(uintptr_t)data[k], (Py_ssize_t)n[k], ref));
    }
    return out;
}

// ---------------------------------------------------------------- enqueue-only compute calls (return the status)
// xengXgpuKernelAsync[Acc]: tried without waiting; 256 launches ahead of the GPU the lock is given up for the wait
int kernel_async(unsigned long long in, unsigned long long out, int dump, unsigned long long acc, int mode) {
    for (;;) {
        int rc = xengXgpuTryKernelAsyncAcc((const void*)(uintptr_t)in, (void*)(uintptr_t)out, dump, (void*)(uintptr_t)acc, mode);
        if (rc != XENG_STATUS_WOULD_BLOCK) return rc;
        Py_BEGIN_ALLOW_THREADS
        rc = xengXgpuWaitLaunchSlot();
        Py_END_ALLOW_THREADS
        if (rc) return rc;
    }
}

PyObject* xgpu_kernel_async(PyObject*, PyObject* args) {
    unsigned long long in, out;
    int dump;
    if (!PyArg_ParseTuple(args, "KKi", &in, &out, &dump)) return nullptr;
    return PyLong_FromLong(kernel_async(in, out, dump, 0, 0));
}

PyObject* xgpu_kernel_async_acc(PyObject*, PyObject* args) {
    unsigned long long in, out, acc;
    int dump, mode;
    if (!PyArg_ParseTuple(args, "KKiKi", &in, &out, &dump, &acc, &mode)) return nullptr;
    return PyLong_FromLong(kernel_async(in, out, dump, acc, mode));
}

// xgpu_kernel_slab(packets, npkt, stride, seq0, chan0, out, dump, acc, mode): a gulp as the slab of packets it arrived in
PyObject* xgpu_kernel_slab(PyObject*, PyObject* args) {
    unsigned long long pk, seq0, out, acc;
    int npkt, chan0, dump, mode;
    Py_ssize_t stride;
    if (!PyArg_ParseTuple(args, "KinKiKiKi", &pk, &npkt, &stride, &seq0, &chan0, &out, &dump, &acc, &mode)) return nullptr;
    for (;;) {
        int rc = xengXgpuTryKernelAsyncSlab((const void*)(uintptr_t)pk, npkt, (size_t)stride, (uint64_t)seq0, chan0, (void*)(uintptr_t)out, dump, (void*)(uintptr_t)acc, mode);
        if (rc != XENG_STATUS_WOULD_BLOCK) return PyLong_FromLong(rc);
        Py_BEGIN_ALLOW_THREADS
        rc = xengXgpuWaitLaunchSlot();
        Py_END_ALLOW_THREADS
        if (rc) return PyLong_FromLong(rc);
    }
}

// xgpu_dump_done(lag) -> -status | 0 (not done) | 1 (done)
PyObject* xgpu_dump_done(PyObject*, PyObject* args) {
    int lag, done = 0;
    if (!PyArg_ParseTuple(args, "i", &lag)) return nullptr;
    const int rc = xengXgpuDumpDone(lag, &done);
    return PyLong_FromLong(rc ? -rc : (done ? 1 : 0));
}

PyObject* beam_run(PyObject*, PyObject* args) {
    unsigned long long in, out, w;
    long long version;
    if (!PyArg_ParseTuple(args, "KKKL", &in, &out, &w, &version)) return nullptr;
    int rc = xengBeamformTryRunVersioned((const void*)(uintptr_t)in, (void*)(uintptr_t)out, (const void*)(uintptr_t)w, version);
    if (rc == XENG_STATUS_WOULD_BLOCK) {        // (integrated-power mode, once per weight upload: wait without the lock)
        Py_BEGIN_ALLOW_THREADS
        rc = xengBeamformRunVersioned((const void*)(uintptr_t)in, (void*)(uintptr_t)out, (const void*)(uintptr_t)w, version);
        Py_END_ALLOW_THREADS
    }
    return PyLong_FromLong(rc);
}

// beam_run_parts(in0, ntime0, in1, out, w, version): one beamformer gulp out of two consecutive ring spans
PyObject* beam_run_parts(PyObject*, PyObject* args) {
    unsigned long long in0, in1, out, w;
    int ntime0;
    long long version;
    if (!PyArg_ParseTuple(args, "KiKKKL", &in0, &ntime0, &in1, &out, &w, &version)) return nullptr;
    int rc = xengBeamformTryRunParts((const void*)(uintptr_t)in0, ntime0, (const void*)(uintptr_t)in1, (void*)(uintptr_t)out, (const void*)(uintptr_t)w, version);
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengBeamformRunParts((const void*)(uintptr_t)in0, ntime0, (const void*)(uintptr_t)in1, (void*)(uintptr_t)out, (const void*)(uintptr_t)w, version);
        Py_END_ALLOW_THREADS
    }
    return PyLong_FromLong(rc);
}

// beam_run_slabs(pk0, npkt0, ntime0, pk1, npkt1, stride, seq0, chan0, out, w, version): a beamformer gulp as one or two packet slabs
PyObject* beam_run_slabs(PyObject*, PyObject* args) {
    unsigned long long pk0, pk1, seq0, out, w;
    int npkt0, ntime0, npkt1, chan0;
    Py_ssize_t stride;
    long long version;
    if (!PyArg_ParseTuple(args, "KiiKinKiKKL", &pk0, &npkt0, &ntime0, &pk1, &npkt1, &stride, &seq0, &chan0, &out, &w, &version)) return nullptr;
    int rc = xengBeamformTryRunSlabs((const void*)(uintptr_t)pk0, npkt0, ntime0, (const void*)(uintptr_t)pk1, npkt1, (size_t)stride, (uint64_t)seq0, chan0,
                                     (void*)(uintptr_t)out, (const void*)(uintptr_t)w, version);
    if (rc == XENG_STATUS_WOULD_BLOCK) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengBeamformRunSlabs((const void*)(uintptr_t)pk0, npkt0, ntime0, (const void*)(uintptr_t)pk1, npkt1, (size_t)stride, (uint64_t)seq0, chan0,
                                  (void*)(uintptr_t)out, (const void*)(uintptr_t)w, version);
        Py_END_ALLOW_THREADS
    }
    return PyLong_FromLong(rc);
}

PyObject* beam_integrate(PyObject*, PyObject* args) {
    unsigned long long in, out;
    int ntime_sum;
    if (!PyArg_ParseTuple(args, "KKi", &in, &out, &ntime_sum)) return nullptr;
    return PyLong_FromLong(xengBeamformIntegrate((const void*)(uintptr_t)in, (void*)(uintptr_t)out, ntime_sum));
}

// beam_mark() -> ticket (> 0) | -status
PyObject* beam_mark(PyObject*, PyObject*) {
    unsigned long long t = 0;
    const int rc = xengBeamformMark(&t);
    if (rc) return PyLong_FromLong(-rc);
    return PyLong_FromUnsignedLongLong(t);
}

// beam_ticket_done(ticket) -> -status | 0 | 1
PyObject* beam_ticket_done(PyObject*, PyObject* args) {
    unsigned long long t;
    int done = 0;
    if (!PyArg_ParseTuple(args, "K", &t)) return nullptr;
    const int rc = xengBeamformTicketDone(t, &done);
    return PyLong_FromLong(rc ? -rc : (done ? 1 : 0));
}

// copy_async(dst, src, nbytes) -> stamp (bytes): dst <- src on the copy stream, only enqueued; the stamp completes when the copy has
PyObject* copy_async(PyObject*, PyObject* args) {
    unsigned long long dst, src;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "KKn", &dst, &src, &n)) return nullptr;
    int rc = xengMemcpyAsync((void*)(uintptr_t)dst, (const void*)(uintptr_t)src, (size_t)n);
    if (rc) return raise_xeng("xengMemcpyAsync", rc);
    xengStamp st;
    rc = xengStampNowFor(&st, nullptr, XENG_STREAMS_COPY);
    if (rc) return raise_xeng("xengStampNowFor", rc);
    return PyBytes_FromStringAndSize((const char*)&st, sizeof(st));
}

static bool stamp_arg(PyObject* args, xengStamp* st) {
    const char* p;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "y#", &p, &n)) return false;
    if (n != (Py_ssize_t)sizeof(xengStamp)) { PyErr_SetString(PyExc_ValueError, "not a stamp"); return false; }
    memcpy(st, p, sizeof(*st));
    return true;
}

// stamp_done(stamp) -> bool, never waits
PyObject* stamp_done(PyObject*, PyObject* args) {
    xengStamp st;
    if (!stamp_arg(args, &st)) return nullptr;
    int done = 0;
    const int rc = xengStampDone(&st, &done, nullptr);
    if (rc) return raise_xeng("xengStampDone", rc);
    return PyBool_FromLong(done);
}

// stamp_wait(stamp): asks first; gives the interpreter lock up only to wait
PyObject* stamp_wait(PyObject*, PyObject* args) {
    xengStamp st;
    if (!stamp_arg(args, &st)) return nullptr;
    int done = 0;
    int rc = xengStampDone(&st, &done, nullptr);
    if (!rc && !done) {
        Py_BEGIN_ALLOW_THREADS
        rc = xengStampWait(&st);
        Py_END_ALLOW_THREADS
    }
    if (rc) return raise_xeng("xengStampWait", rc);
    Py_RETURN_NONE;
}

// ---------------------------------------------------------------- _xfast.bench: harness helpers (bench.py, probes): a data source and sinks that are not Python threads
// ring_feed_external(h, seq, ptrs (bytes: uint64 each), nbytes, count): commits ptrs[k % n] as external spans, `count` times, waiting for
// room like any writer; the interpreter lock is released for the whole loop
PyObject* ring_feed_external(PyObject*, PyObject* args) {
    unsigned long long h;
    long long seq;
    const char* pp;
    Py_ssize_t plen, nbytes, count;
    if (!PyArg_ParseTuple(args, "KLy#nn", &h, &seq, &pp, &plen, &nbytes, &count)) return nullptr;
    const Py_ssize_t nptr = plen / 8;
    if (nptr <= 0) { PyErr_SetString(PyExc_ValueError, "no span addresses"); return nullptr; }
    std::vector<unsigned long long> ptrs((size_t)nptr);
    memcpy(ptrs.data(), pp, (size_t)nptr * 8);
    int rc = 0;
    Py_BEGIN_ALLOW_THREADS
    for (Py_ssize_t k = 0; k < count && !rc; k++)
        rc = xengRingCommitExternal((xengRing*)h, seq, (void*)(uintptr_t)ptrs[(size_t)(k % nptr)], (size_t)nbytes, 1);
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng("xengRingCommitExternal", rc);
    Py_RETURN_NONE;
}

// ring_feed_slabs(h, seq, ptrs, nbytes, count, npkt, stride, pkts_per_seq, ntime): as ring_feed_external for a set of slab buffers that
// is reused like a receiver's: before slab k goes out its headers are stamped with the sequence numbers of window k
// (xengSnap2StampSeq), and it is not reused before `nptr` more windows have gone out -- the caller sizes the set so that nobody
// still holds it then (ring capacity + what the readers keep in flight)
PyObject* ring_feed_slabs(PyObject*, PyObject* args) {
    unsigned long long h;
    long long seq;
    const char* pp;
    Py_ssize_t plen, nbytes, count, stride;
    int npkt, pkts_per_seq, ntime;
    if (!PyArg_ParseTuple(args, "KLy#nninii", &h, &seq, &pp, &plen, &nbytes, &count, &npkt, &stride, &pkts_per_seq, &ntime)) return nullptr;
    const Py_ssize_t nptr = plen / 8;
    if (nptr <= 0) { PyErr_SetString(PyExc_ValueError, "no slab addresses"); return nullptr; }
    std::vector<unsigned long long> ptrs((size_t)nptr);
    memcpy(ptrs.data(), pp, (size_t)nptr * 8);
    int rc = 0;
    const char* where = "xengRingCommitExternal";
    Py_BEGIN_ALLOW_THREADS
    for (Py_ssize_t k = 0; k < count && !rc; k++) {
        void* slab = (void*)(uintptr_t)ptrs[(size_t)(k % nptr)];
        rc = xengSnap2StampSeq(slab, npkt, (size_t)stride, (uint64_t)k * (uint64_t)ntime, pkts_per_seq);
        if (rc) { where = "xengSnap2StampSeq"; break; }
        rc = xengRingCommitExternal((xengRing*)h, seq, slab, (size_t)nbytes, 1);
    }
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng(where, rc);
    Py_RETURN_NONE;
}

// ring_drain(h, reader, gulp, want_times) -> (spans seen, [time.perf_counter() of each span] or []): reads every sequence to its end,
// releasing each span at once; the interpreter lock is released for the whole loop
PyObject* ring_drain(PyObject*, PyObject* args) {
    unsigned long long h;
    int reader, want_times;
    Py_ssize_t gulp;
    if (!PyArg_ParseTuple(args, "Kinp", &h, &reader, &gulp, &want_times)) return nullptr;
    std::vector<double> times;
    long long nspans = 0;
    int rc = 0;
    Py_BEGIN_ALLOW_THREADS
    for (;;) {
        long long seq = 0, tag = 0;
        int nringlet = 1;
        const void* hdr = nullptr;
        size_t hlen = 0;
        rc = xengRingNextSequence((xengRing*)h, reader, 1, &seq, &tag, &nringlet, &hdr, &hlen);
        if (rc) break;
        size_t advance = 0;
        for (;;) {
            void* data = nullptr;
            size_t n = 0, skipped = 0;
            long long span = 0;
            rc = xengRingAcquire((xengRing*)h, reader, advance, (size_t)gulp, 1, &data, &n, &span, &skipped);
            if (rc) break;
            if (want_times) {
                struct timespec ts;
                clock_gettime(CLOCK_MONOTONIC, &ts);          // (the clock of time.perf_counter())
                times.push_back((double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
            }
            nspans++;
            (void)xengRingSpanRelease(span);
            advance = n;
            if (n < (size_t)gulp) { rc = XENG_STATUS_END_OF_DATA; break; }
        }
        if (rc != XENG_STATUS_END_OF_DATA) break;
    }
    Py_END_ALLOW_THREADS
    if (rc != XENG_STATUS_END_OF_DATA) return raise_xeng("ring_drain", rc);
    PyObject* lst = PyList_New((Py_ssize_t)times.size());
    if (!lst) return nullptr;
    for (size_t k = 0; k < times.size(); k++) PyList_SET_ITEM(lst, (Py_ssize_t)k, PyFloat_FromDouble(times[k]));
    return Py_BuildValue("(LN)", nspans, lst);
}


// ---------------------------------------------------------------- the compute calls the native per-gulp loops make
// A table of function pointers, the library's own by default.  tests/fake_backend.py substitutes ctypes callbacks (the oracle on
// system-space rings), so that the pumps' paths -- short tails, a Reserve that fails, an enqueue that fails mid-flight, the
// stop-flag carry, skips on slab sequences -- run in the CPU suite and not only on the GPU box (round-4 review, W2).
struct ComputeOps {
    int (*beam_run_versioned)(const void*, void*, const void*, long long);
    int (*beam_run_parts)(const void*, int, const void*, void*, const void*, long long);
    int (*beam_run_slabs)(const void*, int, int, const void*, int, size_t, uint64_t, int, void*, const void*, long long);
    int (*beam_integrate)(const void*, void*, int);
    int (*beam_mark)(unsigned long long*);
    int (*beam_wait)(unsigned long long);
    int (*beam_sync)(void);
    int (*memcpy_async)(void*, const void*, size_t);
    int (*stamp_now_for)(xengStamp*, const void*, unsigned);
    int (*stamp_done)(const xengStamp*, int*, int*);
    int (*stamp_wait)(const xengStamp*);
    int (*dev_malloc)(void**, size_t, int);
    int (*dev_free)(void*, int);
    int (*xgpu_try_kernel)(const void*, void*, int, void*, int);
    int (*xgpu_try_kernel_slab)(const void*, int, size_t, uint64_t, int, void*, int, void*, int);
    int (*xgpu_wait_slot)(void);
    int (*xgpu_sync_lag)(int);
    int (*xgpu_sync)(void);
    int (*xgpu_reset)(void);
};
constexpr size_t NOPS = sizeof(ComputeOps) / sizeof(void*);

ComputeOps default_ops() {
    ComputeOps o;
    o.beam_run_versioned = xengBeamformRunVersioned; o.beam_run_parts = xengBeamformRunParts; o.beam_run_slabs = xengBeamformRunSlabs;
    o.beam_integrate = xengBeamformIntegrate; o.beam_mark = xengBeamformMark; o.beam_wait = xengBeamformWait; o.beam_sync = xengBeamformSync;
    o.memcpy_async = xengMemcpyAsync; o.stamp_now_for = xengStampNowFor; o.stamp_done = xengStampDone; o.stamp_wait = xengStampWait;
    o.dev_malloc = xengMalloc; o.dev_free = xengFree;
    o.xgpu_try_kernel = xengXgpuTryKernelAsyncAcc; o.xgpu_try_kernel_slab = xengXgpuTryKernelAsyncSlab; o.xgpu_wait_slot = xengXgpuWaitLaunchSlot;
    o.xgpu_sync_lag = xengXgpuSyncLag; o.xgpu_sync = xengXgpuSync; o.xgpu_reset = xengXgpuReset;
    return o;
}

// `table`: None, or a bytes object of NOPS pointers in the order of ComputeOps (0 = the library's own)
bool ops_from_arg(PyObject* table, ComputeOps* out) {
    *out = default_ops();
    if (!table || table == Py_None) return true;
    char* p;
    Py_ssize_t n;
    if (PyBytes_AsStringAndSize(table, &p, &n) < 0) return false;
    if (n != (Py_ssize_t)sizeof(ComputeOps)) { PyErr_Format(PyExc_ValueError, "compute table of %zd bytes, %zu expected", n, sizeof(ComputeOps)); return false; }
    void* in[NOPS];
    memcpy(in, p, sizeof(in));
    void** dst = (void**)out;
    for (size_t k = 0; k < NOPS; k++)
        if (in[k]) dst[k] = in[k];
    return true;
}

// ---------------------------------------------------------------- BeamPump: the per-gulp loop of Beamform / BeamformSumBeams
// What the two blocks do per gulp in steady state -- take the next input gulp, reserve an output span, enqueue the kernel, mark
// it, and retire the oldest gulp in flight (wait for its ticket, commit its span, give its input back) -- as one loop inside the
// extension with the interpreter lock released.  Everything that is not steady state stays in the block's Python: sequences
// and headers, commands and coefficient uploads (the block passes the address of a flag; the pump looks at it after every
// gulp it has taken and returns with that gulp still in hand), statistics.  Two block threads then touch the interpreter
// lock once per `max_gulps` gulps instead of ~15 times per gulp: the lock belongs to Corr and CorrAcc, and a block thread
// that the OS takes off its core no longer holds everybody else up (profiles/r04/blocks_gpu_idle.txt).
struct PumpItem {
    unsigned long long ticket = 0;
    long long out_span = 0;
    void* out_ptr = nullptr;
    long long in_span[2] = {0, 0};
    int nin = 0;
    void* stage = nullptr;
    xengStamp copy_stamp;
};

struct BeamPump {
    PyObject_HEAD
    PyObject* rin_obj;
    PyObject* rout_obj;
    xengRing* rin;
    xengRing* rout;
    int reader, mode, ntime_sum, depth, staged, row_bytes;
    // mode 0 on a sequence of packet slabs (slab_npkt > 0): a gulp is one or two slabs, handed to xengBeamformRunSlabs
    int slab_npkt, slab_ntime, slab_chan0, ntime_gulp;
    size_t slab_stride;
    long long oseq;
    size_t igulp, ogulp, advance;
    std::deque<PumpItem>* pending;
    std::deque<PumpItem>* copying;
    std::vector<void*>* stages_free;
    // a gulp taken from the input ring but not yet processed (the stop flag was up when it arrived)
    int have_carry;
    void* carry_data[2];
    size_t carry_n[2];
    long long carry_span[2];
    int carry_nparts;
    ComputeOps ops;
};

static void pump_release_item(PumpItem& it, bool commit, BeamPump* p) {
    if (it.out_span) {
        if (commit) (void)xengRingCommit(p->rout, p->oseq, it.out_span, p->ogulp);
        (void)xengRingSpanRelease(it.out_span);
        it.out_span = 0;
    }
    for (int k = 0; k < it.nin; k++)
        if (it.in_span[k]) { (void)xengRingSpanRelease(it.in_span[k]); it.in_span[k] = 0; }
    if (it.stage) { p->stages_free->push_back(it.stage); it.stage = nullptr; }
}

// (no interpreter lock held) commits whose copy has completed; keep > 0: leave that many in flight unless they are done anyway
static int pump_finish_copies(BeamPump* p, size_t keep) {
    while (!p->copying->empty()) {
        PumpItem& it = p->copying->front();
        int done = 0;
        int rc = p->ops.stamp_done(&it.copy_stamp, &done, nullptr);
        if (rc) return rc;
        if (!done) {
            if (p->copying->size() <= keep) break;
            rc = p->ops.stamp_wait(&it.copy_stamp);
            if (rc) return rc;
        }
        pump_release_item(it, true, p);
        p->copying->pop_front();
    }
    return XENG_STATUS_SUCCESS;
}

static int pump_retire(BeamPump* p, size_t keep) {
    while (p->pending->size() > keep) {
        PumpItem it = p->pending->front();
        int rc = p->ops.beam_wait(it.ticket);
        if (rc) return rc;
        p->pending->pop_front();
        if (it.stage) {
            // the kernel's sums are in the device buffer: on to the pinned span on the copy stream; committed when that is done
            rc = p->ops.memcpy_async(it.out_ptr, it.stage, p->ogulp);
            if (!rc) rc = p->ops.stamp_now_for(&it.copy_stamp, nullptr, XENG_STREAMS_COPY);
            if (rc) { pump_release_item(it, false, p); return rc; }
            for (int k = 0; k < it.nin; k++)
                if (it.in_span[k]) { (void)xengRingSpanRelease(it.in_span[k]); it.in_span[k] = 0; }
            it.nin = 0;
            p->copying->push_back(it);
        } else {
            pump_release_item(it, true, p);
        }
    }
    return pump_finish_copies(p, keep ? 2 : 0);
}

// after an error: nothing in flight may still touch a span when it goes back to its ring
static void pump_abort(BeamPump* p) {
    (void)p->ops.beam_sync();
    for (auto& it : *p->copying) (void)p->ops.stamp_wait(&it.copy_stamp);
    for (auto& it : *p->pending) pump_release_item(it, false, p);
    for (auto& it : *p->copying) pump_release_item(it, false, p);
    p->pending->clear();
    p->copying->clear();
    if (p->have_carry) {
        for (int k = 0; k < p->carry_nparts; k++) (void)xengRingSpanRelease(p->carry_span[k]);
        p->have_carry = 0;
    }
}

void BeamPump_dealloc(BeamPump* self) {
    if (self->pending) {
        Py_BEGIN_ALLOW_THREADS
        pump_abort(self);
        Py_END_ALLOW_THREADS
        for (void* st : *self->stages_free) (void)self->ops.dev_free(st, XENG_SPACE_CUDA);
        delete self->pending;
        delete self->copying;
        delete self->stages_free;
    }
    Py_CLEAR(self->rin_obj);
    Py_CLEAR(self->rout_obj);
    Py_TYPE(self)->tp_free((PyObject*)self);
}

PyTypeObject BeamPumpType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// beam_pump(in_ring_obj, in_handle, reader, out_ring_obj, out_handle, out_seq, igulp, ogulp, mode (0 Beamform | 1 SumBeams), row_bytes,
//           ntime_sum, depth, staged[, compute table]) -> BeamPump
PyObject* beam_pump_new(PyObject*, PyObject* args) {
    PyObject *rin_obj, *rout_obj, *table = nullptr;
    unsigned long long hin, hout;
    int reader, mode, row_bytes, ntime_sum, depth, staged;
    long long oseq;
    Py_ssize_t igulp, ogulp;
    if (!PyArg_ParseTuple(args, "OKiOKLnniiiii|O", &rin_obj, &hin, &reader, &rout_obj, &hout, &oseq, &igulp, &ogulp, &mode, &row_bytes, &ntime_sum, &depth, &staged, &table))
        return nullptr;
    ComputeOps ops;
    if (!ops_from_arg(table, &ops)) return nullptr;
    BeamPump* p = PyObject_New(BeamPump, &BeamPumpType);
    if (!p) return nullptr;
    p->ops = ops;
    Py_INCREF(rin_obj); Py_INCREF(rout_obj);
    p->rin_obj = rin_obj; p->rout_obj = rout_obj;
    p->rin = (xengRing*)hin; p->rout = (xengRing*)hout;
    p->reader = reader; p->mode = mode; p->ntime_sum = ntime_sum; p->depth = depth > 0 ? depth : 1; p->staged = staged; p->row_bytes = row_bytes > 0 ? row_bytes : 1;
    p->oseq = oseq; p->igulp = (size_t)igulp; p->ogulp = (size_t)ogulp; p->advance = 0;
    p->pending = new std::deque<PumpItem>(); p->copying = new std::deque<PumpItem>(); p->stages_free = new std::vector<void*>();
    p->have_carry = 0; p->carry_nparts = 0;
    p->slab_npkt = 0; p->slab_ntime = 0; p->slab_chan0 = 0; p->ntime_gulp = 0; p->slab_stride = 0;
    return (PyObject*)p;
}

// run(weights, version, max_gulps, stop_flag_address) -> (gulps enqueued, bytes skipped, status)
//   status 0: max_gulps done;  1: the input sequence is over (everything in flight has been committed);
//          2: the stop flag was up when a gulp arrived -- that gulp is kept and is the first of the next run()
PyObject* BeamPump_run(BeamPump* p, PyObject* args) {
    unsigned long long weights, stop_addr, seq0 = 0;
    long long version;
    int max_gulps;
    if (!PyArg_ParseTuple(args, "KLiK|K", &weights, &version, &max_gulps, &stop_addr, &seq0)) return nullptr;
    volatile int* stop = (volatile int*)(uintptr_t)stop_addr;
    long long ngulps = 0;
    size_t skipped_total = 0;
    int status = 0, rc = 0;
    const char* where = "";
    Py_BEGIN_ALLOW_THREADS
    while (ngulps < max_gulps) {
        void* data[2] = {nullptr, nullptr};
        size_t n[2] = {0, 0}, skipped = 0;
        long long span[2] = {0, 0};
        int nparts = 1;
        if (p->have_carry) {
            for (int k = 0; k < 2; k++) { data[k] = p->carry_data[k]; n[k] = p->carry_n[k]; span[k] = p->carry_span[k]; }
            nparts = p->carry_nparts;
            p->have_carry = 0;
        } else {
            if (p->mode == 0) rc = xengRingAcquireParts(p->rin, p->reader, p->advance, p->igulp, 1, data, n, span, &nparts, &skipped);
            else rc = xengRingAcquire(p->rin, p->reader, p->advance, p->igulp, 1, &data[0], &n[0], &span[0], &skipped);
            if (rc == XENG_STATUS_END_OF_DATA) { rc = 0; status = 1; break; }
            if (rc) { where = "xengRingAcquire"; break; }
            skipped_total += skipped;
            if (p->slab_npkt > 0 && p->igulp) seq0 += (unsigned long long)(skipped / p->igulp) * (unsigned long long)p->ntime_gulp;
            size_t got = 0;
            for (int k = 0; k < nparts; k++) got += n[k];
            p->advance = got;
            if (got < p->igulp) {             // the short tail of an ended sequence: not a gulp (the blocks skip it)
                for (int k = 0; k < nparts; k++) (void)xengRingSpanRelease(span[k]);
                status = 1;
                break;
            }
            if (stop && *stop) {              // a command came in while this thread waited for the gulp: the block looks first
                for (int k = 0; k < 2; k++) { p->carry_data[k] = data[k]; p->carry_n[k] = n[k]; p->carry_span[k] = span[k]; }
                p->carry_nparts = nparts;
                p->have_carry = 1;
                status = 2;
                break;
            }
        }
        PumpItem it;
        it.nin = nparts;
        for (int k = 0; k < nparts; k++) it.in_span[k] = span[k];
        rc = xengRingReserve(p->rout, p->oseq, p->ogulp, 0, 1, &it.out_ptr, &it.out_span);
        if (rc) { where = "xengRingReserve"; pump_release_item(it, false, p); break; }
        if (p->mode == 0 && p->slab_npkt > 0) {
            const size_t slab_bytes = (size_t)p->slab_npkt * p->slab_stride;
            const void* s0 = data[0];
            const void* s1 = nparts == 2 ? data[1] : (n[0] >= 2 * slab_bytes ? (const void*)((const char*)data[0] + slab_bytes) : nullptr);
            // (every part must be exactly one slab, or the one part one or two of them: anything else is not a slab sequence)
            const bool whole = nparts == 2 ? (n[0] == slab_bytes && n[1] == slab_bytes) : (n[0] == slab_bytes || n[0] == 2 * slab_bytes);
            if (!whole) { pump_release_item(it, false, p); rc = XENG_STATUS_INVALID_ARGUMENT; where = "BeamPump: a gulp that is not whole packet slabs"; break; }
            rc = p->ops.beam_run_slabs(s0, p->slab_npkt, p->slab_ntime, s1, p->slab_npkt, p->slab_stride, (uint64_t)seq0, p->slab_chan0, it.out_ptr,
                                       (const void*)(uintptr_t)weights, version);
            seq0 += (unsigned long long)p->ntime_gulp;
            where = "xengBeamformRunSlabs";
        } else if (p->mode == 0) {
            if (nparts == 2) {
                // (an upstream writer whose spans are not whole samples: the second part would be read from the wrong byte)
                if (n[0] % (size_t)p->row_bytes) { pump_release_item(it, false, p); rc = XENG_STATUS_INVALID_ARGUMENT; where = "BeamPump: the first part of a two-part gulp is not a whole number of samples"; break; }
                rc = p->ops.beam_run_parts(data[0], (int)(n[0] / (size_t)p->row_bytes), data[1], it.out_ptr, (const void*)(uintptr_t)weights, version);
            } else rc = p->ops.beam_run_versioned(data[0], it.out_ptr, (const void*)(uintptr_t)weights, version);
            where = "xengBeamformRun";
        } else {
            void* target = it.out_ptr;
            if (p->staged) {
                if (!p->stages_free->empty()) { it.stage = p->stages_free->back(); p->stages_free->pop_back(); }
                else rc = p->ops.dev_malloc(&it.stage, p->ogulp, XENG_SPACE_CUDA);
                target = it.stage;
            }
            if (!rc) rc = p->ops.beam_integrate(data[0], target, p->ntime_sum);
            where = "xengBeamformIntegrate";
        }
        if (!rc) { rc = p->ops.beam_mark(&it.ticket); if (rc) where = "xengBeamformMark"; }
        if (rc) { (void)p->ops.beam_sync(); pump_release_item(it, false, p); break; }
        p->pending->push_back(it);
        ngulps++;
        rc = pump_retire(p, (size_t)p->depth);
        if (rc) { where = "retire"; break; }
    }
    if (!rc && status == 1) { rc = pump_retire(p, 0); if (rc) where = "retire"; }
    if (rc) pump_abort(p);
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng(where, rc);
    return Py_BuildValue("(Lni)", ngulps, (Py_ssize_t)skipped_total, status);
}

// drain(): everything in flight is waited for and committed (before the weights on the device are rewritten; at the end)
PyObject* BeamPump_drain(BeamPump* p, PyObject*) {
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = pump_retire(p, 0);
    if (rc) pump_abort(p);
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng("BeamPump.drain", rc);
    Py_RETURN_NONE;
}

PyObject* BeamPump_abort(BeamPump* p, PyObject*) {
    Py_BEGIN_ALLOW_THREADS
    pump_abort(p);
    Py_END_ALLOW_THREADS
    Py_RETURN_NONE;
}

// set_slabs(npkt, stride, slab_ntime, chan0, ntime_gulp): the input sequence holds packet slabs (Beamform only)
PyObject* BeamPump_set_slabs(BeamPump* p, PyObject* args) {
    int npkt, slab_ntime, chan0, ntime_gulp;
    Py_ssize_t stride;
    if (!PyArg_ParseTuple(args, "iniii", &npkt, &stride, &slab_ntime, &chan0, &ntime_gulp)) return nullptr;
    p->slab_npkt = npkt; p->slab_stride = (size_t)stride; p->slab_ntime = slab_ntime; p->slab_chan0 = chan0; p->ntime_gulp = ntime_gulp;
    Py_RETURN_NONE;
}

PyMethodDef BeamPump_methods[] = {
    {"set_slabs", (PyCFunction)BeamPump_set_slabs, METH_VARARGS, "(npkt, stride, slab_ntime, chan0, ntime_gulp): gulps are packet slabs"},
    {"run", (PyCFunction)BeamPump_run, METH_VARARGS, "(weights, version, max_gulps, stop_flag_address[, seq0 of the next gulp: slab sequences]) -> (gulps, skipped bytes, status)"},
    {"drain", (PyCFunction)BeamPump_drain, METH_NOARGS, "wait for and commit everything in flight"},
    {"abort", (PyCFunction)BeamPump_abort, METH_NOARGS, "after an error elsewhere: wait for the stream, give every span back uncommitted"},
    {nullptr, nullptr, 0, nullptr}};

// ---------------------------------------------------------------- CorrPump: the per-gulp loop of Corr (round 5)
// What Corr does per gulp while it integrates -- take the next gulp, (first gulp of an integration: reserve the output span,)
// register the gulp with the X-engine, keep its span until the dump has run, and on the dump gulp rotate: the dump of
// integration n is enqueued, integration n-1's is waited for (xengXgpuSyncLag(1)), its span committed and its gulps given back
// -- as one loop inside the extension with the interpreter lock released (corr_block.py:388-466 is one ctypes call per gulp on
// bifrost's native ring; here the ring calls and the enqueue are native too).  The block's Python keeps what is decided per
// integration or rarer: sequences and headers, commands (same stop flag as BeamPump: a gulp that arrives with the flag up is
// handed back unprocessed), the integration gate's start / recovery, statistics.  mode 0 ("skip") only consumes gulps: the
// states in which the reference's loop passes gulps by (waiting for the start sample, stopped).
struct CorrPump {
    PyObject_HEAD
    PyObject* rin_obj;
    PyObject* rout_obj;
    xengRing* rin;
    xengRing* rout;
    int reader;
    long long oseq;
    size_t igulp, ogulp, advance;
    int slab_npkt, slab_chan0, ntime_gulp;
    size_t slab_stride;
    ComputeOps ops;
    // the integration in progress: its output span and the gulps registered so far
    void* out_ptr;
    long long out_span;
    std::vector<long long>* held;
    // the integration whose dump is in flight
    int have_pending;
    long long pend_out;
    std::vector<long long>* pend_held;
    // a gulp taken from the input ring but not processed (stop flag up, or the reader had lost data before it)
    int have_carry;
    void* carry_data;
    size_t carry_n;
    long long carry_span;
};

static void corr_release(std::vector<long long>* v) {
    for (long long sp : *v) (void)xengRingSpanRelease(sp);
    v->clear();
}

// the dump in flight has completed (the caller waited): its span is committed, its gulps go back to the input ring
static void corr_commit_pending(CorrPump* p) {
    if (!p->have_pending) return;
    (void)xengRingCommit(p->rout, p->oseq, p->pend_out, p->ogulp);
    (void)xengRingSpanRelease(p->pend_out);
    corr_release(p->pend_held);
    p->pend_out = 0;
    p->have_pending = 0;
}

// wait for the dump in flight and commit it (Corr._finish_pending)
static int corr_finish(CorrPump* p) {
    if (!p->have_pending) return XENG_STATUS_SUCCESS;
    const int rc = p->ops.xgpu_sync();
    if (rc) return rc;
    corr_commit_pending(p);
    return XENG_STATUS_SUCCESS;
}

// drop the integration in progress (Corr._abort_integration): a completed one whose dump is in flight is still good; the
// registered gulps are forgotten by the X-engine (xengXgpuReset synchronises: nothing reads them afterwards)
static int corr_abort(CorrPump* p, bool commit_pending, bool always_reset = true) {
    int rc = XENG_STATUS_SUCCESS;
    const bool mid_integration = !p->held->empty() || p->out_span;
    if (p->have_pending) {
        rc = p->ops.xgpu_sync();
        if (!rc && commit_pending) corr_commit_pending(p);
        else {       // (after an error: nothing may be committed; the spans go back unpublished once the stream is idle)
            if (p->pend_out) (void)xengRingSpanRelease(p->pend_out);
            corr_release(p->pend_held);
            p->pend_out = 0;
            p->have_pending = 0;
        }
    }
    if (mid_integration || always_reset) {
        const int rc2 = p->ops.xgpu_reset();
        if (!rc) rc = rc2;
    }
    corr_release(p->held);
    if (p->out_span) { (void)xengRingSpanRelease(p->out_span); p->out_span = 0; p->out_ptr = nullptr; }
    return rc;
}

void CorrPump_dealloc(CorrPump* self) {
    if (self->held) {
        Py_BEGIN_ALLOW_THREADS
        if (self->have_pending || !self->held->empty() || self->out_span) (void)corr_abort(self, false);
        if (self->have_carry) { (void)xengRingSpanRelease(self->carry_span); self->have_carry = 0; }
        Py_END_ALLOW_THREADS
        delete self->held;
        delete self->pend_held;
    }
    Py_CLEAR(self->rin_obj);
    Py_CLEAR(self->rout_obj);
    Py_TYPE(self)->tp_free((PyObject*)self);
}

PyTypeObject CorrPumpType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// corr_pump(in_ring_obj, in_handle, reader, out_ring_obj, out_handle, igulp, ogulp, ntime_gulp[, compute table]) -> CorrPump
PyObject* corr_pump_new(PyObject*, PyObject* args) {
    PyObject *rin_obj, *rout_obj, *table = nullptr;
    unsigned long long hin, hout;
    int reader, ntime_gulp;
    Py_ssize_t igulp, ogulp;
    if (!PyArg_ParseTuple(args, "OKiOKnni|O", &rin_obj, &hin, &reader, &rout_obj, &hout, &igulp, &ogulp, &ntime_gulp, &table)) return nullptr;
    ComputeOps ops;
    if (!ops_from_arg(table, &ops)) return nullptr;
    CorrPump* p = PyObject_New(CorrPump, &CorrPumpType);
    if (!p) return nullptr;
    Py_INCREF(rin_obj); Py_INCREF(rout_obj);
    p->rin_obj = rin_obj; p->rout_obj = rout_obj;
    p->rin = (xengRing*)hin; p->rout = (xengRing*)hout;
    p->reader = reader; p->oseq = -1; p->igulp = (size_t)igulp; p->ogulp = (size_t)ogulp; p->advance = 0; p->ntime_gulp = ntime_gulp;
    p->slab_npkt = 0; p->slab_chan0 = 0; p->slab_stride = 0;
    p->ops = ops;
    p->out_ptr = nullptr; p->out_span = 0; p->held = new std::vector<long long>();
    p->have_pending = 0; p->pend_out = 0; p->pend_held = new std::vector<long long>();
    p->have_carry = 0; p->carry_data = nullptr; p->carry_n = 0; p->carry_span = 0;
    return (PyObject*)p;
}

// run(mode, max_gulps, stop_flag_address, pos, gulps_per_integration, now) -> (gulps consumed, bytes skipped, status, integrations dumped)
//   mode 0: gulps are taken and given back (the gate is waiting / stopped); 1: gulps are registered, the one at position
//   gulps_per_integration - 1 dumps.  `pos`: position of the next gulp in its integration; `now`: its sample number (slabs).
//   status 0: max_gulps consumed;  1: the input sequence is over;  2: the stop flag was up when a gulp arrived -- kept, first of the
//   next run();  3: the reader had lost `skipped` bytes before the gulp that arrived -- kept likewise, nothing else consumed.
PyObject* CorrPump_run(CorrPump* p, PyObject* args) {
    int mode, gpi, pos;
    long long max_gulps;
    unsigned long long stop_addr, now;
    if (!PyArg_ParseTuple(args, "iLKiiK", &mode, &max_gulps, &stop_addr, &pos, &gpi, &now)) return nullptr;
    if (gpi < 1 || pos < 0 || pos >= gpi) { PyErr_SetString(PyExc_ValueError, "CorrPump.run: bad position in the integration"); return nullptr; }
    if (mode == 1 && p->oseq < 0) { PyErr_SetString(PyExc_ValueError, "CorrPump.run: no output sequence set"); return nullptr; }
    volatile int* stop = (volatile int*)(uintptr_t)stop_addr;
    long long ngulps = 0, nint = 0;
    size_t skipped_total = 0;
    int status = 0, rc = 0;
    const char* where = "";
    Py_BEGIN_ALLOW_THREADS
    while (ngulps < max_gulps) {
        void* data = nullptr;
        size_t n = 0, skipped = 0;
        long long span = 0;
        if (p->have_carry) {
            data = p->carry_data; n = p->carry_n; span = p->carry_span;
            p->have_carry = 0;
        } else {
            rc = xengRingAcquire(p->rin, p->reader, p->advance, p->igulp, 1, &data, &n, &span, &skipped);
            if (rc == XENG_STATUS_END_OF_DATA) { rc = 0; status = 1; break; }
            if (rc) { where = "xengRingAcquire"; break; }
            p->advance = n;
            if (n < p->igulp) {              // the short tail of an ended sequence: not a gulp (corr_block.py:389-391)
                (void)xengRingSpanRelease(span);
                status = 1;
                break;
            }
            if (skipped || (stop && *stop)) {
                p->carry_data = data; p->carry_n = n; p->carry_span = span;
                p->have_carry = 1;
                skipped_total += skipped;
                status = skipped ? 3 : 2;
                break;
            }
        }
        if (mode == 0) {
            (void)xengRingSpanRelease(span);
            ngulps++;
            now += (unsigned long long)p->ntime_gulp;
            continue;
        }
        if (pos == 0 && !p->out_span) {      // one output span per integration (corr_block.py:433-435)
            rc = xengRingReserve(p->rout, p->oseq, p->ogulp, 0, 1, &p->out_ptr, &p->out_span);
            if (rc) { where = "xengRingReserve"; (void)xengRingSpanRelease(span); p->out_span = 0; break; }
        }
        if (!p->out_span) { rc = XENG_STATUS_INVALID_STATE; where = "CorrPump: a gulp inside an integration that has no output span"; (void)xengRingSpanRelease(span); break; }
        const int dump = pos == gpi - 1;
        for (;;) {
            if (p->slab_npkt > 0) rc = p->ops.xgpu_try_kernel_slab(data, p->slab_npkt, p->slab_stride, (uint64_t)now, p->slab_chan0, p->out_ptr, dump, nullptr, 0);
            else rc = p->ops.xgpu_try_kernel(data, p->out_ptr, dump, nullptr, 0);
            if (rc != XENG_STATUS_WOULD_BLOCK) break;
            rc = p->ops.xgpu_wait_slot();    // 256 launches ahead of the GPU
            if (rc) break;
        }
        if (rc) { where = "xengXgpuKernelAsync"; (void)xengRingSpanRelease(span); break; }     // (this gulp was not registered)
        p->held->push_back(span);            // the gulp is read where it lies when the dump runs: its memory stays until then
        ngulps++;
        now += (unsigned long long)p->ntime_gulp;
        if (dump) {
            // the dump of this integration is enqueued; the previous one's has had a whole integration to run
            const int had = p->have_pending;
            long long prev_out = p->pend_out;
            std::vector<long long> prev_held;
            prev_held.swap(*p->pend_held);
            p->pend_out = p->out_span; p->pend_held->swap(*p->held); p->have_pending = 1;
            p->out_span = 0; p->out_ptr = nullptr;
            if (had) {
                rc = p->ops.xgpu_sync_lag(1);
                if (rc) {                    // (cannot tell whether it completed: hand the spans to the abort below, uncommitted)
                    where = "xengXgpuSyncLag";
                    for (long long sp : prev_held) p->held->push_back(sp);
                    p->held->push_back(prev_out);
                    break;
                }
                (void)xengRingCommit(p->rout, p->oseq, prev_out, p->ogulp);
                (void)xengRingSpanRelease(prev_out);
                for (long long sp : prev_held) (void)xengRingSpanRelease(sp);
            }
            nint++;
            pos = 0;
        } else {
            pos++;
        }
    }
    if (rc) (void)corr_abort(p, false);
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng(where, rc);
    return Py_BuildValue("(LniL)", ngulps, (Py_ssize_t)skipped_total, status, nint);
}

PyObject* CorrPump_finish(CorrPump* p, PyObject*) {
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = corr_finish(p);
    if (rc) (void)corr_abort(p, false);
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng("CorrPump.finish", rc);
    Py_RETURN_NONE;
}

PyObject* CorrPump_abort(CorrPump* p, PyObject*) {
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = corr_abort(p, true);
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng("CorrPump.abort", rc);
    Py_RETURN_NONE;
}

// close(): at the end of the input sequence or after an error elsewhere -- everything in flight is waited for; the dump in flight is
// committed (commit = 1) or dropped; gulps in hand go back
PyObject* CorrPump_close(CorrPump* p, PyObject* args) {
    int commit = 1;
    if (!PyArg_ParseTuple(args, "|p", &commit)) return nullptr;
    int rc;
    Py_BEGIN_ALLOW_THREADS
    rc = corr_abort(p, commit != 0, false);
    if (p->have_carry) { (void)xengRingSpanRelease(p->carry_span); p->have_carry = 0; }
    Py_END_ALLOW_THREADS
    if (rc) return raise_xeng("CorrPump.close", rc);
    Py_RETURN_NONE;
}

PyObject* CorrPump_set_output(CorrPump* p, PyObject* args) {
    long long oseq;
    if (!PyArg_ParseTuple(args, "L", &oseq)) return nullptr;
    if (p->have_pending || p->out_span) { PyErr_SetString(PyExc_RuntimeError, "CorrPump.set_output with an integration in flight (finish() first)"); return nullptr; }
    p->oseq = oseq;
    Py_RETURN_NONE;
}

PyObject* CorrPump_set_slabs(CorrPump* p, PyObject* args) {
    int npkt, chan0;
    Py_ssize_t stride;
    if (!PyArg_ParseTuple(args, "ini", &npkt, &stride, &chan0)) return nullptr;
    p->slab_npkt = npkt; p->slab_stride = (size_t)stride; p->slab_chan0 = chan0;
    Py_RETURN_NONE;
}

PyObject* CorrPump_state(CorrPump* p, PyObject*) {
    return Py_BuildValue("(iniii)", p->have_pending, (Py_ssize_t)p->held->size(), p->out_span ? 1 : 0, p->have_carry, (int)p->pend_held->size());
}

PyMethodDef CorrPump_methods[] = {
    {"run", (PyCFunction)CorrPump_run, METH_VARARGS, "(mode, max_gulps, stop_flag_address, pos, gulps_per_integration, now) -> (gulps, skipped bytes, status, integrations)"},
    {"finish", (PyCFunction)CorrPump_finish, METH_NOARGS, "wait for the dump in flight and commit its span"},
    {"abort", (PyCFunction)CorrPump_abort, METH_NOARGS, "drop the integration in progress (the dump in flight is still committed)"},
    {"close", (PyCFunction)CorrPump_close, METH_VARARGS, "([commit = True]) end of the sequence / after an error: nothing stays in flight or in hand"},
    {"set_output", (PyCFunction)CorrPump_set_output, METH_VARARGS, "(output sequence id)"},
    {"set_slabs", (PyCFunction)CorrPump_set_slabs, METH_VARARGS, "(npkt, stride, chan0): gulps are packet slabs"},
    {"state", (PyCFunction)CorrPump_state, METH_NOARGS, "-> (dump in flight, gulps held, output span open, gulp carried, gulps of the dump in flight)"},
    {nullptr, nullptr, 0, nullptr}};

PyObject* map_i32(PyObject*, PyObject* args) {
    unsigned long long a, b;
    Py_ssize_t n;
    int add;
    if (!PyArg_ParseTuple(args, "KKnp", &a, &b, &n, &add)) return nullptr;
    return PyLong_FromLong(add ? xengMapAddI32((void*)(uintptr_t)a, (const void*)(uintptr_t)b, (size_t)n)
                               : xengMapAssignI32((void*)(uintptr_t)a, (const void*)(uintptr_t)b, (size_t)n));
}

PyMethodDef methods[] = {
    {"ring_reserve", ring_reserve, METH_VARARGS, "(ring_obj, handle, seq, nbytes, nonblocking) -> (ptr, SpanRef, span)"},
    {"ring_commit", ring_commit, METH_VARARGS, "(handle, seq, span, nbytes)"},
    {"ring_commit_external", ring_commit_external, METH_VARARGS, "(handle, seq, ptr, nbytes)"},
    {"ring_next_sequence", ring_next_sequence, METH_VARARGS, "(handle, reader) -> None | (header, time_tag, nringlet)"},
    {"ring_acquire", ring_acquire, METH_VARARGS, "(ring_obj, handle, reader, advance, gulp) -> None | (ptr, nbytes, SpanRef, skipped)"},
    {"ring_acquire_parts", ring_acquire_parts, METH_VARARGS, "(ring_obj, handle, reader, advance, gulp) -> None | (skipped, (ptr, nbytes, SpanRef), ...)"},
    {"xgpu_kernel_async", xgpu_kernel_async, METH_VARARGS, "xengXgpuKernelAsync -> status"},
    {"xgpu_kernel_async_acc", xgpu_kernel_async_acc, METH_VARARGS, "xengXgpuKernelAsyncAcc -> status"},
    {"xgpu_kernel_slab", xgpu_kernel_slab, METH_VARARGS, "xengXgpuKernelAsyncSlab (a gulp as its packet slab) -> status"},
    {"xgpu_dump_done", xgpu_dump_done, METH_VARARGS, "xengXgpuDumpDone(lag) -> -status | 0 | 1"},
    {"beam_run", beam_run, METH_VARARGS, "xengBeamformRunVersioned -> status"},
    {"beam_run_parts", beam_run_parts, METH_VARARGS, "xengBeamformRunParts (two ring spans as one gulp) -> status"},
    {"beam_run_slabs", beam_run_slabs, METH_VARARGS, "xengBeamformRunSlabs (a gulp as one or two packet slabs) -> status"},
    {"beam_integrate", beam_integrate, METH_VARARGS, "xengBeamformIntegrate -> status"},
    {"beam_mark", beam_mark, METH_NOARGS, "xengBeamformMark -> ticket | -status"},
    {"beam_ticket_done", beam_ticket_done, METH_VARARGS, "xengBeamformTicketDone -> -status | 0 | 1"},
    {"map_i32", map_i32, METH_VARARGS, "(a, b, nwords, add) -> status"},
    {"beam_pump", beam_pump_new, METH_VARARGS, "(in_ring, in_handle, reader, out_ring, out_handle, out_seq, igulp, ogulp, mode, row_bytes, ntime_sum, depth, staged[, compute table]) -> BeamPump"},
    {"corr_pump", corr_pump_new, METH_VARARGS, "(in_ring, in_handle, reader, out_ring, out_handle, igulp, ogulp, ntime_gulp[, compute table]) -> CorrPump"},
    {"compute_table_size", [](PyObject*, PyObject*) -> PyObject* { return PyLong_FromSize_t(NOPS); }, METH_NOARGS, "number of function pointers in a compute table"},
    {"copy_async", copy_async, METH_VARARGS, "(dst, src, nbytes) -> stamp of the enqueued copy (bytes)"},
    {"stamp_done", stamp_done, METH_VARARGS, "(stamp) -> bool"},
    {"stamp_wait", stamp_wait, METH_VARARGS, "(stamp): waits (interpreter lock released) unless it is done"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_xfast", "direct binding of libxeng's per-gulp calls (include/xeng.h)", -1, methods};

// _xfast.bench: what bench.py and the probes under profiles/ need and no pipeline does -- a source and sinks that are not Python
// threads (so that the interpreter lock belongs to the blocks under test).  Kept apart from the product's bindings.
PyMethodDef bench_methods[] = {
    {"ring_feed_external", ring_feed_external, METH_VARARGS, "(handle, seq, addresses, nbytes, count) -- a source that is not a Python thread"},
    {"ring_feed_slabs", ring_feed_slabs, METH_VARARGS, "(handle, seq, addresses, nbytes, count, npkt, stride, pkts_per_seq, ntime) -- a receiver that reuses its slab buffers"},
    {"ring_drain", ring_drain, METH_VARARGS, "(handle, reader, gulp, want_times) -> (spans, times) -- a sink that is not a Python thread"},
    {nullptr, nullptr, 0, nullptr}};
PyModuleDef bench_moddef = {PyModuleDef_HEAD_INIT, "_xfast.bench", "harness helpers of bench.py / profiles (not used by the blocks)", -1, bench_methods};

}  // namespace

PyMODINIT_FUNC PyInit__xfast(void) {
    SpanRefType.tp_name = "_xfast.SpanRef";
    SpanRefType.tp_basicsize = sizeof(SpanRef);
    SpanRefType.tp_flags = Py_TPFLAGS_DEFAULT;
    SpanRefType.tp_dealloc = (destructor)SpanRef_dealloc;
    SpanRefType.tp_doc = "a reference on a ring span's memory, given back when this object dies";
    if (PyType_Ready(&SpanRefType) < 0) return nullptr;
    BeamPumpType.tp_name = "_xfast.BeamPump";
    BeamPumpType.tp_basicsize = sizeof(BeamPump);
    BeamPumpType.tp_flags = Py_TPFLAGS_DEFAULT;
    BeamPumpType.tp_dealloc = (destructor)BeamPump_dealloc;
    BeamPumpType.tp_methods = BeamPump_methods;
    BeamPumpType.tp_doc = "the steady-state per-gulp loop of Beamform / BeamformSumBeams, run without the interpreter lock";
    if (PyType_Ready(&BeamPumpType) < 0) return nullptr;
    CorrPumpType.tp_name = "_xfast.CorrPump";
    CorrPumpType.tp_basicsize = sizeof(CorrPump);
    CorrPumpType.tp_flags = Py_TPFLAGS_DEFAULT;
    CorrPumpType.tp_dealloc = (destructor)CorrPump_dealloc;
    CorrPumpType.tp_methods = CorrPump_methods;
    CorrPumpType.tp_doc = "the per-gulp loop of Corr while it integrates, run without the interpreter lock";
    if (PyType_Ready(&CorrPumpType) < 0) return nullptr;
    PyObject* m = PyModule_Create(&moddef);
    if (!m) return nullptr;
    Py_INCREF(&SpanRefType);
    PyModule_AddObject(m, "SpanRef", (PyObject*)&SpanRefType);
    PyObject* b = PyModule_Create(&bench_moddef);
    if (!b) return nullptr;
    PyModule_AddObject(m, "bench", b);
    return m;
}
