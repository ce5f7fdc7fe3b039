"""The C-ABI library loads on a machine without a GPU and exports every symbol include/xeng.h
declares; entry points fail loudly (status + message), never silently, when there is no device."""
import ctypes
import os
import re

import pytest

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "xeng.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:xeng|bf)[A-Z]\w*)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 40
    L = ctypes.CDLL(ffi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libxeng.so does not export %s" % n
    bound = set(ffi.SYMBOLS) | set(ffi.STRING_SYMBOLS)
    assert set(names) == bound, (set(names) ^ bound)


def test_version_and_error_strings():
    L = ffi.lib()
    assert b"gfx950" in L.xengVersion()
    assert isinstance(L.xengGetLastError(), bytes)


def test_no_cpu_fallback_without_device():
    """Without a GPU the compute entry points return an error status with a message."""
    n = ctypes.c_int(-1)
    rc = ffi.lib().xengGetDeviceCount(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ffi.XengError) as ei:
        ffi.call("xengXgpuInitialize", 0)
    assert ei.value.status != 0 and str(ei.value)
    with pytest.raises(ffi.XengError):
        ffi.call("xengXgpuKernel", 16, 16, 1)          # not initialised -> INVALID_STATE, no compute
    with pytest.raises(ffi.XengError):
        ffi.call("xengBeamformRun", 16, 16, 16)
    done = ctypes.c_int(-1)
    for name, arg in (("xengBeamformTicketDone", 1), ("xengXgpuDumpDone", 0)):      # the queries do not pretend either
        with pytest.raises(ffi.XengError):
            ffi.call(name, arg, ctypes.byref(done))
    assert done.value == -1


def test_calls_made_with_the_interpreter_lock_kept_cannot_wait():
    """ffi.enqueue_lib() keeps the interpreter lock across the call, so nothing on it may wait for the GPU or for another
    thread.  Round 3 held the list to names; three of those names could wait: xengXgpuKernelAsync[Acc] at 256
    launches in flight, xengBeamformRun* once per weight upload in the integrated-power mode, xengSnap2UnpackAsync behind the
    synchronous call's mutex.  Now the library has forms that return XENG_STATUS_WOULD_BLOCK instead of waiting, and only those
    are bound with the lock kept (behaviour on the GPU: tests/test_enqueue_gpu.py).  Here, without a device: the waiting forms
    are not on the handle, the non-waiting forms are the library's own symbols, and each returns its error at once."""
    import time
    E = ffi.enqueue_lib()
    may_wait = ("xengXgpuKernelAsync", "xengXgpuKernelAsyncAcc", "xengXgpuKernel", "bfXgpuKernel", "xengBeamformRun", "xengBeamformRunVersioned",
                "bfBeamformRun", "xengSnap2UnpackAsync", "xengSnap2Unpack", "xengXgpuWaitLaunchSlot", "xengBeamformWait", "xengStampWait",
                "xengXgpuSync", "xengXgpuSyncLag", "xengBeamformSync", "xengMapSync", "xengMemcpy", "xengMemset", "xengMalloc", "xengFree")
    for name in may_wait:
        assert name in ffi.SYMBOLS and name not in ffi.ENQUEUE_ONLY, name
    for name in ("xengXgpuTryKernelAsyncAcc", "xengBeamformTryRunVersioned", "xengXgpuDumpDone", "xengBeamformTicketDone", "xengStampDone"):
        assert name in ffi.ENQUEUE_ONLY, name
    for name in ffi.ENQUEUE_ONLY:
        assert ctypes.cast(getattr(E, name), ctypes.c_void_p).value == ctypes.cast(getattr(ffi.lib(), name), ctypes.c_void_p).value
    n = ctypes.c_int(-1)
    if ffi.lib().xengGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        return
    t0 = time.perf_counter()
    assert E.xengXgpuTryKernelAsyncAcc(16, 16, 1, None, 0) != 0          # (not initialised: an error status, at once)
    assert E.xengBeamformTryRunVersioned(16, 16, 16, 1) != 0
    assert time.perf_counter() - t0 < 0.5


def test_argument_validation_needs_no_gpu():
    with pytest.raises(ffi.XengError):
        ffi.call("xengXgpuConfigure", 351, 2, 96, 480, 0)     # nstand not a multiple of 4
    with pytest.raises(ffi.XengError):
        ffi.call("xengXgpuConfigure", 352, 1, 96, 480, 0)     # npol != 2
    ffi.call("xengXgpuConfigure", 352, 2, 96, 480, 0)
    with pytest.raises(ffi.XengError):
        ffi.call("xengBeamformInitialize", 0, 703, 96, 960, 32, 0)


def test_get_order_and_reorder_are_host_functions():
    """GetOrder / Reorder are host code (corr_block.py:317-333, corr_output_full_block.py:669):
    they run without a GPU and agree with the oracle."""
    import numpy as np
    from oracle import xeng_oracle as orc
    ns, nchan = 16, 4
    ffi.call("xengXgpuConfigure", ns, 2, nchan, 8, 0)
    rng = np.random.default_rng(0)
    a2i = rng.permutation(ns * 2).astype(np.int32).reshape(ns, 2)
    bl = np.zeros((ns, ns, 2, 2), np.int32)
    cj = np.zeros_like(bl)
    ffi.call("xengXgpuGetOrder", a2i.ctypes.data, bl.ctypes.data, cj.ctypes.data)
    obl, ocj = orc.xgpu_get_order(a2i)
    assert np.array_equal(bl, obl) and np.array_equal(cj, ocj)
    planar = rng.integers(-1000, 1000, 2 * orc.per_chan(ns) * nchan).astype(np.int32)
    out = np.zeros((ns, ns, 2, 2, nchan, 2), np.int32)
    ffi.call("xengXgpuReorder", planar.ctypes.data, out.ctypes.data, bl.ctypes.data, cj.ctypes.data)
    assert np.array_equal(out, orc.xgpu_reorder(planar, bl, cj, nchan))
    ffi.call("xengXgpuConfigure", 352, 2, 96, 480, 0)


def test_as_bfarray_reference_passes_where_a_struct_pointer_is_expected():
    """XArray.as_BFarray() returns a light reference: ctypes turns it into the XENGarray pointer when a bf* entry point
    takes it (here the host-side bfXgpuGetOrder, no GPU needed), and `.contents` / `.data` answer like the pointer would."""
    import numpy as np
    from caltech_bifrost_dsp_amd.ndarray import XArray
    ns = 16
    ffi.call("xengXgpuConfigure", ns, 2, 4, 8, 0)
    a2i = np.random.default_rng(1).permutation(ns * 2).astype(np.int32).reshape(ns, 2)
    xa = XArray(a2i, space="system")
    bl, cj = XArray(shape=(ns, ns, 2, 2), dtype=np.int32, space="system"), XArray(shape=(ns, ns, 2, 2), dtype=np.int32, space="system")
    assert ffi.lib().bfXgpuGetOrder(xa.as_BFarray(), bl.as_BFarray(), cj.as_BFarray()) == 0
    rbl, rcj = np.zeros((ns, ns, 2, 2), np.int32), np.zeros((ns, ns, 2, 2), np.int32)
    ffi.call("xengXgpuGetOrder", a2i.ctypes.data, rbl.ctypes.data, rcj.ctypes.data)
    assert np.array_equal(bl.numpy(), rbl) and np.array_equal(cj.numpy(), rcj)
    r = xa.as_BFarray()
    assert r.data == xa.ptr and r.contents.data == xa.ptr and r.contents.ndim == 2 and r.contents.shape[0] == ns
    assert list(r.contents.strides[:2]) == [8, 4]
    ffi.call("xengXgpuConfigure", 352, 2, 96, 480, 0)
