"""ctypes binding of libxeng.so (include/xeng.h) -- the product's only route to the GPU.

This plays the role of `from bifrost.libbifrost import _bf` in the reference blocks
(corr_block.py:5, beamform_block.py:5).  There is no CPU fallback: if the HIP library is
missing or fails to load, importing `lib()` raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# XENG_LIB: load another build of the same library (the -DXENG_DIAGNOSTICS build used by profiles/: timing-only ablations)
LIB_PATH = os.environ.get("XENG_LIB") or os.path.join(_HERE, "libxeng.so")

STATUS_SUCCESS = 0
SPACE_SYSTEM, SPACE_CUDA, SPACE_CUDA_HOST = 1, 2, 3

_lib = None


class XengError(RuntimeError):
    def __init__(self, fn, status, msg):
        super().__init__("%s returned %d: %s" % (fn, status, msg))
        self.status = status


class XENGarray(ctypes.Structure):
    """Mirror of include/xeng.h XENGarray (= bifrost BFarray)."""
    _fields_ = [("data", ctypes.c_void_p), ("space", ctypes.c_int), ("dtype", ctypes.c_int),
                ("ndim", ctypes.c_int), ("shape", ctypes.c_long * 8), ("strides", ctypes.c_long * 8),
                ("immutable", ctypes.c_int), ("big_endian", ctypes.c_int), ("conjugated", ctypes.c_int)]


class XengStamp(ctypes.Structure):
    """include/xeng.h xengStamp: everything the library has enqueued so far (opaque)."""
    _fields_ = [("w", ctypes.c_ulonglong * 16)]


STATUS_WOULD_BLOCK, STATUS_END_OF_DATA = 6, 7
STREAMS = {"xgpu": 1, "map": 2, "beam": 4, "copy": 8, "consumer": 16, "xgpu_out": 32}       # include/xeng.h XENG_STREAMS_*
# test hooks of the span rings (xengRingSetStampHooks)
STAMP_NOW_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong))
STAMP_DONE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong))
STAMP_WAIT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong))

# every symbol include/xeng.h declares: name -> argtypes (all return int unless noted)
_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_pi = ctypes.POINTER(ctypes.c_int)
_pa = ctypes.POINTER(XENGarray)
_ll, _pll, _psz, _pvp = ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p)
_pst = ctypes.POINTER(XengStamp)
SYMBOLS = {
    "xengStampNow": [_pst], "xengStampNowFor": [_pst, _vp, ctypes.c_uint], "xengStampDone": [_pst, _pi, _pi], "xengStampWait": [_pst],
    "xengRingCreate": [_pvp, ctypes.c_char_p, _i], "xengRingDestroy": [_vp], "xengRingResize": [_vp, _sz, _sz], "xengRingSetRecycle": [_vp, _i], "xengRingDeclareStreams": [_vp, ctypes.c_uint], "xengRingGetStampClasses": [_vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)],
    "xengRingGetInfo": [_vp, _psz, _psz, _psz, _pi, _pll, ctypes.POINTER(ctypes.c_ulonglong)],
    "xengRingBeginSequence": [_vp, _ll, ctypes.c_char_p, _sz, _i, _pll], "xengRingEndSequence": [_vp, _ll], "xengRingEndWriting": [_vp],
    "xengRingReserve": [_vp, _ll, _sz, _i, _i, _pvp, _pll], "xengRingCommit": [_vp, _ll, _ll, _sz],
    "xengRingCommitExternal": [_vp, _ll, _vp, _sz, _i],
    "xengRingOpenReader": [_vp, _i, _pi], "xengRingCloseReader": [_vp, _i],
    "xengRingNextSequence": [_vp, _i, _i, _pll, _pll, _pi, _pvp, _psz],
    "xengRingAcquire": [_vp, _i, _sz, _sz, _i, _pvp, _psz, _pll, _psz], "xengRingSpanRelease": [_ll],
    "xengRingAcquireParts": [_vp, _i, _sz, _sz, _i, _pvp, _psz, _pll, _pi, _psz],
    "xengRingSetStampHooks": [_vp, STAMP_NOW_FN, STAMP_DONE_FN, STAMP_WAIT_FN, _vp],
    "xengGetDeviceCount": [_pi], "xengSetDevice": [_i], "xengGetDevice": [_pi], "xengDeviceSynchronize": [],
    "xengGetDeviceInfo": [_i, _pi, _pi, ctypes.POINTER(_sz), ctypes.c_char_p, _i], "xengGetDevicePciBusId": [_i, ctypes.c_char_p, _i],
    "xengMalloc": [ctypes.POINTER(_vp), _sz, _i], "xengFree": [_vp, _i], "xengMemcpy": [_vp, _vp, _sz],
    "xengMemcpyAsync": [_vp, _vp, _sz], "xengMemset": [_vp, _i, _sz], "xengStreamSynchronize": [],
    "xengXgpuConfigure": [_i, _i, _i, _i, _i], "xengXgpuInitialize": [_i], "xengXgpuDestroy": [],
    "xengXgpuKernel": [_vp, _vp, _i], "xengXgpuKernelAsync": [_vp, _vp, _i], "xengXgpuKernelAsyncAcc": [_vp, _vp, _i, _vp, _i], "xengXgpuTryKernelAsyncAcc": [_vp, _vp, _i, _vp, _i], "xengXgpuWaitLaunchSlot": [],
    "xengXgpuKernelAsyncSlab": [_vp, _i, _sz, ctypes.c_uint64, _i, _vp, _i, _vp, _i], "xengXgpuTryKernelAsyncSlab": [_vp, _i, _sz, ctypes.c_uint64, _i, _vp, _i, _vp, _i], "xengXgpuGetSlabFallbacks": [_pi], "xengXgpuGetSlabStats": [_pi, _pi], "xengXgpuSync": [], "xengXgpuSyncLag": [_i], "xengXgpuDumpDone": [_i, _pi], "xengXgpuReset": [],
    "xengXgpuCorrelate": [_vp, _vp, _i], "xengXgpuGetOrder": [_vp, _vp, _vp],
    "xengXgpuSubSelect": [_vp, _vp, _vp, _vp, _i, _i], "xengXgpuReorder": [_vp, _vp, _vp, _vp],
    "xengXgpuGetInfo": [_pi, _pi, _pi, _pi, ctypes.POINTER(ctypes.c_int64), _pi],
    "xengXgpuGetPath": [_pi, _pi], "xengXgpuGetKernel": [_pi, _pi],
    "xengXgpuPacketize": [_vp, _vp, _vp, _vp, _i],
    "xengSnap2Unpack": [_vp, _i, ctypes.c_size_t, _vp, ctypes.c_uint64, _i, _i, _i, _i, _i, _pi, _pi],
    "xengSnap2UnpackAsync": [_vp, _i, ctypes.c_size_t, _vp, ctypes.c_uint64, _i, _i, _i, _i, _i],
    "xengSnap2GetAsyncDrops": [_pi], "xengSnap2StampSeq": [_vp, _i, _sz, ctypes.c_uint64, _i],
    "xengXgpuSetProfiling": [_i], "xengXgpuGetTimes": [ctypes.POINTER(ctypes.c_double), _pi],
    "xengMapAssignI32": [_vp, _vp, _sz], "xengMapAddI32": [_vp, _vp, _sz], "xengMapSync": [],
    "xengMapSumI32": [_vp, _pvp, _i, _sz, _i],
    "xengBeamformInitialize": [_i, _i, _i, _i, _i, _i], "xengBeamformDestroy": [],
    "xengBeamformRun": [_vp, _vp, _vp], "xengBeamformRunVersioned": [_vp, _vp, _vp, ctypes.c_longlong], "xengBeamformTryRunVersioned": [_vp, _vp, _vp, ctypes.c_longlong],
    "xengBeamformRunParts": [_vp, _i, _vp, _vp, _vp, ctypes.c_longlong], "xengBeamformTryRunParts": [_vp, _i, _vp, _vp, _vp, ctypes.c_longlong],
    "xengBeamformRunSlabs": [_vp, _i, _i, _vp, _i, _sz, ctypes.c_uint64, _i, _vp, _vp, ctypes.c_longlong], "xengBeamformTryRunSlabs": [_vp, _i, _i, _vp, _i, _sz, ctypes.c_uint64, _i, _vp, _vp, ctypes.c_longlong], "xengBeamformGetSlabFallbacks": [_pi], "xengBeamformGetSlabStats": [_pi, _pi],
    "xengBeamformIntegrate": [_vp, _vp, _i],
    "xengBeamformIntegrateSingleBeam": [_vp, _vp, _i, _i], "xengBeamformMark": [ctypes.POINTER(ctypes.c_ulonglong)], "xengBeamformWait": [ctypes.c_ulonglong], "xengBeamformTicketDone": [ctypes.c_ulonglong, _pi], "xengBeamformSync": [],
    "xengBeamformSetProfiling": [_i], "xengBeamformGetTimes": [ctypes.POINTER(ctypes.c_double), _pi],
    "xengBeamformGetRouteInfo": [_pi, _pi, _pi],
    "bfXgpuInitialize": [_pa, _pa, _i], "bfXgpuKernel": [_pa, _pa, _i], "bfXgpuCorrelate": [_pa, _pa, _i],
    "bfXgpuGetOrder": [_pa, _pa, _pa], "bfXgpuSubSelect": [_pa, _pa, _pa, _pa, _i, _i],
    "bfXgpuReorder": [_pa, _pa, _pa, _pa], "bfBeamformInitialize": [_i, _i, _i, _i, _i, _i],
    "bfBeamformRun": [_pa, _pa, _pa], "bfBeamformIntegrate": [_pa, _pa, _i],
    "bfBeamformIntegrateSingleBeam": [_pa, _pa, _i, _i],
}
STRING_SYMBOLS = ["xengGetLastError", "xengVersion"]


def lib():
    """Load libxeng.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C caltech-bifrost-dsp_amd/csrc). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SYMBOLS.items():
            if os.environ.get("XENG_LIB") and not hasattr(L, name):
                continue            # (an older build loaded for an A/B by profiles/: calling what it lacks raises AttributeError there)
            f = getattr(L, name)
            f.argtypes = args
            f.restype = ctypes.c_int
        for name in STRING_SYMBOLS:
            getattr(L, name).restype = ctypes.c_char_p
            getattr(L, name).argtypes = []
        _lib = L
    return _lib


# Enqueue-only entry points (and the two completion queries): a few microseconds of host work, nothing to wait for.  The block threads call them through a
# second handle on the same library that does NOT release the interpreter lock around the call (ctypes.PyDLL): a thread that
# gives the lock up for a 3 us call has to win it back afterwards, and with several block threads in one interpreter that
# costs a sleep and a wake-up -- measured 50 us per call on the GPU box (profiles/r03/blocks_lock_handoff.txt), more than the
# GPU needs for the gulp.  Calls that wait (Sync, Wait, the synchronous X-engine call, copies) stay on the releasing handle.
# (Round 4: the list holds only calls that cannot wait by construction.  xengXgpuKernelAsync[Acc] wait at 256 launches in
# flight and xengBeamformRun* waits once after a weight upload in the integrated-power mode: their Try* forms return
# XENG_STATUS_WOULD_BLOCK instead, and the caller gives the lock up to wait; xengSnap2UnpackAsync shares a mutex with the
# synchronous call, which polls: it is made on the releasing handle.)
ENQUEUE_ONLY = ["xengXgpuTryKernelAsyncAcc", "xengXgpuTryKernelAsyncSlab", "xengBeamformTryRunVersioned", "xengBeamformTryRunParts", "xengBeamformTryRunSlabs",
                "xengBeamformIntegrate", "xengBeamformIntegrateSingleBeam", "xengBeamformMark", "xengMapAssignI32",
                "xengMapAddI32", "xengMapSumI32", "xengXgpuDumpDone", "xengBeamformTicketDone", "bfBeamformIntegrate", "bfBeamformIntegrateSingleBeam",
                # the span rings: bookkeeping calls, and the calls that can wait asked with may_block = 0 first
                "xengRingBeginSequence", "xengRingEndSequence", "xengRingEndWriting", "xengRingReserve", "xengRingCommit",
                "xengRingCommitExternal", "xengRingNextSequence", "xengRingAcquire", "xengRingAcquireParts", "xengRingSpanRelease", "xengRingGetInfo",
                "xengRingOpenReader", "xengRingCloseReader", "xengRingResize",
                "xengStampNow", "xengStampNowFor", "xengStampDone", "xengGetDevice", "xengSetDevice"]
_enq = None


def enqueue_lib():
    """The same libxeng.so, bound without releasing the interpreter lock; only the ENQUEUE_ONLY symbols are typed."""
    global _enq
    if _enq is None:
        lib()                                   # (raises when the extension is missing)
        E = ctypes.PyDLL(LIB_PATH)
        for name in ENQUEUE_ONLY:
            f = getattr(E, name)
            f.argtypes = SYMBOLS[name]
            f.restype = ctypes.c_int
        _enq = E
    return _enq


def check(name, status):
    if status != STATUS_SUCCESS:
        raise XengError(name, status, lib().xengGetLastError().decode())


def call(name, *args):
    check(name, getattr(lib(), name)(*args))


class DeviceBuffer:
    """A HIP allocation (space 'cuda') or pinned host allocation ('cuda_host')."""

    def __init__(self, nbytes, space=SPACE_CUDA):
        self.nbytes = int(nbytes)
        self.space = space
        p = ctypes.c_void_p()
        call("xengMalloc", ctypes.byref(p), self.nbytes, space)
        self.ptr = p.value

    def free(self):
        if self.ptr:
            call("xengFree", self.ptr, self.space)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        call("xengMemcpy", self.ptr + offset, arr.ctypes.data, arr.nbytes)
        return self

    def download(self, dtype, count=None, offset=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dtype.itemsize
        out = np.empty(count, dtype=dtype)
        call("xengMemcpy", out.ctypes.data, self.ptr + offset, out.nbytes)
        return out

    def as_host_array(self, dtype):
        assert self.space == SPACE_CUDA_HOST
        dtype = np.dtype(dtype)
        buf = (ctypes.c_char * self.nbytes).from_address(self.ptr)
        buf._owner = self
        return np.frombuffer(buf, dtype=dtype)


def device_count():
    n = ctypes.c_int(0)
    call("xengGetDeviceCount", ctypes.byref(n))
    return n.value


def device_pci_bus_id(gpu=0):
    buf = ctypes.create_string_buffer(32)
    call("xengGetDevicePciBusId", gpu, buf, 32)
    return buf.value.decode().lower()


def device_info(gpu=0):
    cu, clk, mem = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    name = ctypes.create_string_buffer(128)
    call("xengGetDeviceInfo", gpu, ctypes.byref(cu), ctypes.byref(clk), ctypes.byref(mem), name, 128)
    return {"num_cu": cu.value, "clock_khz": clk.value, "total_mem": mem.value, "name": name.value.decode()}
