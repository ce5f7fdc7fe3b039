"""The minimal ring protocol (system space, no GPU)."""
import json
import threading

import numpy as np

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd.ring import Ring, WriteSpan
from caltech_bifrost_dsp_amd.ndarray import XArray, copy_array


def test_xarray_views_and_copy():
    a = XArray(np.arange(24, dtype=np.int32).reshape(2, 3, 4), space="system")
    assert a.shape == (2, 3, 4) and a.nbytes == 96
    assert np.array_equal(a.view(np.uint8).reshape(-1).numpy()[:4], np.array([0, 0, 0, 0], np.uint8))
    b = a.copy()
    b.numpy()[0, 0, 0] = 7
    assert a.numpy()[0, 0, 0] == 0
    c = XArray(shape=(24,), dtype="i32", space="system")
    copy_array(c, a)
    assert np.array_equal(c.numpy(), np.arange(24))
    c[...] = np.zeros(24, np.int32)
    assert not c.numpy().any()
    bf = a.as_BFarray().contents
    assert bf.ndim == 3 and bf.shape[2] == 4 and bf.strides[0] == 48 and bf.data == a.ptr


def test_write_read_gulps_regrouped():
    """Writer commits 10-byte spans; reader asks for 25-byte gulps: data is regrouped in order and
    the short tail is delivered once with size < gulp (the blocks skip it)."""
    ring = Ring(name="t", space="system")
    ring.resize(10, 1000)
    got = []

    def reader():
        for iseq in ring.read(guarantee=True):
            hdr = json.loads(iseq.header.tostring())
            for ispan in iseq.read(25):
                got.append((hdr["seq0"], ispan.size, bytes(ispan.data.numpy())))

    th = threading.Thread(target=reader)
    th.start()
    data = np.arange(70, dtype=np.uint8)
    with ring.begin_writing() as oring:
        with oring.begin_sequence(time_tag=5, header=json.dumps({"seq0": 100}), nringlet=1) as oseq:
            for k in range(7):
                with oseq.reserve(10) as ospan:
                    ospan.data_view(np.uint8).numpy()[...] = data[10 * k:10 * k + 10]
        oseq = oring.begin_sequence(time_tag=6, header=json.dumps({"seq0": 200}))
        sp = WriteSpan(oseq.ring, 25)
        sp.data.numpy()[...] = 9
        sp.close()
        oseq.end()
    th.join(10)
    assert not th.is_alive()
    assert [(s, n) for s, n, _ in got] == [(100, 25), (100, 25), (100, 20), (200, 25)]
    assert b"".join(b for s, n, b in got if s == 100) == data.tobytes()


def test_backpressure_with_guaranteed_reader():
    ring = Ring(name="bp", space="system")
    ring.resize(8, 16)            # room for two spans
    seen = []
    started = threading.Event()

    def reader():
        for iseq in ring.read(guarantee=True):
            started.set()
            for ispan in iseq.read(8):
                seen.append(int(ispan.data.numpy()[0]))

    th = threading.Thread(target=reader)
    th.start()
    with ring.begin_writing() as oring:
        with oring.begin_sequence(time_tag=0, header="{}") as oseq:
            for k in range(20):
                with oseq.reserve(8) as sp:
                    sp.data.numpy()[...] = k
    th.join(10)
    assert not th.is_alive()
    assert seen == list(range(20))      # nothing lost although the ring only holds 2 spans


def test_two_readers_see_everything():
    ring = Ring(name="two", space="system")
    ring.resize(4, 64)
    outs = [[], []]
    ready = threading.Barrier(3)

    def reader(k):
        gen = ring.read(guarantee=True)
        ready.wait()
        for iseq in gen:
            for ispan in iseq.read(4):
                outs[k].append(int(ispan.data_view(np.int32).numpy()[0]))

    ths = [threading.Thread(target=reader, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    ready.wait()
    import time
    time.sleep(0.05)
    with ring.begin_writing() as oring:
        with oring.begin_sequence(time_tag=0, header="{}") as oseq:
            for k in range(50):
                with oseq.reserve(4) as sp:
                    sp.data_view(np.int32).numpy()[0] = k
    for t in ths:
        t.join(10)
        assert not t.is_alive()
    assert outs[0][-1] == 49 and outs[1][-1] == 49
    assert outs[0] == sorted(outs[0]) and outs[1] == sorted(outs[1])


def test_span_memory_is_released_by_reference_counting_alone():
    """A span's memory goes back to its ring the moment the last user lets go of it: nothing on the per-gulp path (views,
    as_BFarray references) may form a reference cycle, or the release would wait for the cycle collector -- on the device
    rings that meant fresh allocations per gulp and frees at arbitrary moments."""
    import gc
    import weakref
    from caltech_bifrost_dsp_amd.ndarray import XArray
    was = gc.isenabled()
    gc.disable()
    try:
        a = XArray(shape=(64,), dtype=np.uint8, space="system")
        v = a.view('i8')
        r = v.as_BFarray()
        assert r.contents.data == a.ptr and r.data == a.ptr and r._as_parameter_ is not None
        w = v.reshape(8, 8).byte_slice(0, 16)
        refs = [weakref.ref(o) for o in (a, v, w)]
        del a, v, r, w
        assert all(x() is None for x in refs)
    finally:
        if was:
            gc.enable()
