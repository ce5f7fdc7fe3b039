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
    {"copy_async", copy_async, METH_VARARGS, "(dst, src, nbytes) -> stamp of the enqueued copy (bytes)"},
    {"stamp_done", stamp_done, METH_VARARGS, "(stamp) -> bool"},
    {"stamp_wait", stamp_wait, METH_VARARGS, "(stamp): waits (interpreter lock released) unless it is done"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_xfast", "direct binding of libxeng's per-gulp calls (include/xeng.h)", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__xfast(void) {
    SpanRefType.tp_name = "_xfast.SpanRef";
    SpanRefType.tp_basicsize = sizeof(SpanRef);
    SpanRefType.tp_flags = Py_TPFLAGS_DEFAULT;
    SpanRefType.tp_dealloc = (destructor)SpanRef_dealloc;
    SpanRefType.tp_doc = "a reference on a ring span's memory, given back when this object dies";
    if (PyType_Ready(&SpanRefType) < 0) return nullptr;
    PyObject* m = PyModule_Create(&moddef);
    if (!m) return nullptr;
    Py_INCREF(&SpanRefType);
    PyModule_AddObject(m, "SpanRef", (PyObject*)&SpanRefType);
    return m;
}
