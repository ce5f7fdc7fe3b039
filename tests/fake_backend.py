"""TEST INFRASTRUCTURE: a backend object with the same methods as
caltech_bifrost_dsp_amd.backend.HipBackend, implemented with the CPU oracle on 'system'-space
arrays.  It lets the not-gpu tests drive the block state machines (BASELINE config 1: "CPU ring
vs numpy, no GPU").  It is injected explicitly by tests; the product never falls back to it."""
import ctypes

import numpy as np

from oracle import xeng_oracle as orc


def _np(pa, dtype, count):
    a = pa.contents if hasattr(pa, "contents") else pa
    buf = (ctypes.c_char * (count * np.dtype(dtype).itemsize)).from_address(a.data)
    return np.frombuffer(buf, dtype=dtype, count=count)


class OracleBackend:
    BF_STATUS_SUCCESS = 0
    space_in = "system"

    def __init__(self):
        self.cfg = None
        self.acc = None
        self.kernel_calls = []
        self.resets = 0
        self.beam = None
        self.device = 0

    def set_device(self, gpu):
        self.device = gpu

    def get_device(self):
        return self.device

    def stream_synchronize(self):
        pass

    def map_sync(self):
        pass

    def beam_sync(self):
        pass

    def beam_mark(self):
        self.marks = getattr(self, "marks", 0) + 1
        return self.marks

    def beam_wait(self, ticket):
        self.waits = getattr(self, "waits", []) + [ticket]

    def last_error(self):
        return ""

    # X-engine
    def xgpu_configure(self, nstand, npol, nchan, ntime_gulp, max_gulps=0):
        self.cfg = dict(nstand=nstand, npol=npol, nchan=nchan, ntime=ntime_gulp)
        return 0

    def bfXgpuInitialize(self, i, o, gpu):
        self.acc = None
        return 0

    def bfXgpuKernel(self, in_arr, out_arr, do_dump):
        c = self.cfg
        n = c["ntime"] * c["nchan"] * c["nstand"] * c["npol"]
        vin = _np(in_arr, np.uint8, n)
        self.acc = orc.xgpu_correlate(vin, c["nstand"], c["nchan"], self.acc)
        self.kernel_calls.append(int(do_dump))
        if do_dump:
            out = _np(out_arr, np.int32, self.acc.size)
            out[...] = self.acc
            self.acc = None
        return 0

    def bfXgpuKernelAsync(self, in_arr, out_arr, do_dump):
        self.async_calls = getattr(self, "async_calls", 0) + 1
        return self.bfXgpuKernel(in_arr, out_arr, do_dump)

    def bfXgpuKernelAsyncAcc(self, in_arr, out_arr, do_dump, acc, acc_mode):
        rv = self.bfXgpuKernelAsync(in_arr, out_arr, do_dump)
        if do_dump:
            self.acc_calls = getattr(self, "acc_calls", []) + [int(acc_mode)]
            dumped = _np(out_arr, np.int32, acc.numpy().size)
            orc.map_i32(acc.numpy().reshape(-1), dumped, add=(acc_mode == 2))
        return rv

    def _unpack_slab(self, slab, npkt, pkt_stride, seq0, ntime, chan0, nchan, ninput):
        raw = slab.numpy().reshape(-1).view(np.uint8)
        pkts = [raw[i * pkt_stride:(i + 1) * pkt_stride].tobytes() for i in range(npkt)]
        gulp, _, _ = orc.snap2_unpack(pkts, seq0, ntime, chan0, nchan, ninput)
        return gulp

    def bfXgpuKernelSlab(self, slab, npkt, pkt_stride, seq0, chan0, out_arr, do_dump, acc=None, acc_mode=0):
        """the oracle's unpack, then the plain call on the unpacked gulp"""
        from caltech_bifrost_dsp_amd.ndarray import XArray
        c = self.cfg
        self.slab_calls = getattr(self, "slab_calls", 0) + 1
        gulp = self._unpack_slab(slab, npkt, pkt_stride, seq0, c["ntime"], chan0, c["nchan"], c["nstand"] * c["npol"])
        g = XArray(shape=[gulp.size], dtype='u8', space='system')
        g.numpy().reshape(-1).view(np.uint8)[...] = gulp.ravel()
        if acc is not None:
            return self.bfXgpuKernelAsyncAcc(g.as_BFarray(), out_arr, do_dump, acc, acc_mode)
        return self.bfXgpuKernelAsync(g.as_BFarray(), out_arr, do_dump)

    def xgpu_fused_acc_supported(self):
        return True

    def xgpu_sync(self):
        return 0

    def xgpu_sync_lag(self, lag):
        self.lag_syncs = getattr(self, "lag_syncs", 0) + 1
        return 0

    def xgpu_reset(self):
        self.acc = None
        self.resets += 1
        return 0

    def bfXgpuGetOrder(self, a2i, bl, cj):
        c = self.cfg
        ns, npol = c["nstand"], c["npol"]
        a = _np(a2i, np.int32, ns * npol).reshape(ns, npol)
        obl, ocj = orc.xgpu_get_order(a)
        _np(bl, np.int32, obl.size)[...] = obl.ravel()
        _np(cj, np.int32, ocj.size)[...] = ocj.ravel()
        return 0

    def xgpu_packetize(self, in_arr, out_arr, antpol_to_bl, is_conj, fmt):
        c = self.cfg
        bl = antpol_to_bl.numpy().reshape(c["nstand"], c["nstand"], c["npol"], c["npol"])
        cj = is_conj.numpy().reshape(bl.shape)
        reordered = orc.xgpu_reorder(in_arr.numpy().reshape(-1), bl, cj, c["nchan"])
        out_arr.numpy().reshape(-1)[...] = orc.corr_packet_payloads(reordered, bool(fmt)).ravel()
        return 0

    def bfXgpuSubSelect(self, in_arr, out_arr, vismap, conj, nchan_sum, unused=0):
        c = self.cfg
        matlen = orc.per_chan(c["nstand"]) * c["nchan"]
        nvis = int(vismap.contents.shape[0])
        planar = _np(in_arr, np.int32, 2 * matlen)
        out = orc.xgpu_subselect(planar, _np(vismap, np.int32, nvis), _np(conj, np.int32, nvis),
                                 c["nchan"], nchan_sum, c["nstand"])
        _np(out_arr, np.int32, out.size)[...] = out.ravel()
        return 0

    def snap2_unpack(self, packets, npkt, pkt_stride, out, seq0, ntime, chan0, nchan_tot, npol_tot, clear=True):
        raw = packets.numpy().reshape(-1).view(np.uint8)
        pkts = [raw[i * pkt_stride:(i + 1) * pkt_stride].tobytes() for i in range(npkt)]
        gulp, placed, dropped = orc.snap2_unpack(pkts, seq0, ntime, chan0, nchan_tot, npol_tot)
        o = out.numpy().reshape(-1).view(np.uint8)
        if clear:
            o[...] = gulp.ravel()
        else:
            o[...] = np.where(gulp.ravel() != 0, gulp.ravel(), o)
        return 0, placed, dropped

    # CorrAcc
    def map_assign_i32(self, a, b):
        orc.map_i32(a.numpy().reshape(-1), b.numpy().reshape(-1), add=False)
        return 0

    def map_add_i32(self, a, b):
        orc.map_i32(a.numpy().reshape(-1), b.numpy().reshape(-1), add=True)
        return 0

    def map_sum_i32(self, a, srcs, add):
        self.sum_calls = getattr(self, "sum_calls", []) + [(len(srcs), bool(add))]
        acc = a.numpy().reshape(-1)
        for k, b in enumerate(srcs):
            orc.map_i32(acc, b.numpy().reshape(-1).view(np.int32), add=(add or k > 0))
        return 0

    # beamformer
    def bfBeamformInitialize(self, gpu, ninput, nchan, ntime, nbeam, ntime_blocks):
        self.beam = dict(ninput=ninput, nchan=nchan, ntime=ntime, nbeam=nbeam, ntime_blocks=ntime_blocks)
        OracleBackend.shared_beam = self.beam       # process-global context, as in the reference
        return 0

    def bfBeamformRun(self, in_arr, out_arr, weights, version=0):
        b = self.beam
        vin = _np(in_arr, np.uint8, b["ntime"] * b["nchan"] * b["ninput"])
        w = _np(weights, np.complex64, b["nchan"] * b["nbeam"] * b["ninput"])
        out = orc.beamform(vin, w, b["ntime"], b["nchan"], b["ninput"], b["nbeam"])
        _np(out_arr, np.complex64, out.size)[...] = out.ravel()
        return 0

    def bfBeamformRunParts(self, part0, part1, out_arr, weights, version=0):
        b = self.beam
        self.parts_calls = getattr(self, "parts_calls", 0) + 1
        vin = np.concatenate([part0.numpy().reshape(-1), part1.numpy().reshape(-1)]).view(np.uint8)
        assert vin.size == b["ntime"] * b["nchan"] * b["ninput"]
        w = _np(weights, np.complex64, b["nchan"] * b["nbeam"] * b["ninput"])
        out = orc.beamform(vin, w, b["ntime"], b["nchan"], b["ninput"], b["nbeam"])
        _np(out_arr, np.complex64, out.size)[...] = out.ravel()
        return 0

    def bfBeamformRunSlabs(self, slab0, npkt0, ntime0, slab1, npkt1, pkt_stride, seq0, chan0, out_arr, weights, version=0):
        b = self.beam
        self.beam_slab_calls = getattr(self, "beam_slab_calls", 0) + 1
        if slab1 is None:
            ntime0 = b["ntime"]
        parts = [self._unpack_slab(slab0, npkt0, pkt_stride, seq0, ntime0, chan0, b["nchan"], b["ninput"])]
        if slab1 is not None:
            parts.append(self._unpack_slab(slab1, npkt1, pkt_stride, seq0 + ntime0, b["ntime"] - ntime0, chan0, b["nchan"], b["ninput"]))
        vin = np.concatenate([p.reshape(-1) for p in parts])
        w = _np(weights, np.complex64, b["nchan"] * b["nbeam"] * b["ninput"])
        out = orc.beamform(vin, w, b["ntime"], b["nchan"], b["ninput"], b["nbeam"])
        _np(out_arr, np.complex64, out.size)[...] = out.ravel()
        return 0

    def bfBeamformIntegrate(self, in_arr, out_arr, ntime_sum):
        b = getattr(OracleBackend, "shared_beam", None)
        if b is None:
            return 2
        beams = _np(in_arr, np.complex64, b["nchan"] * b["nbeam"] * b["ntime"]).reshape(b["nchan"], b["nbeam"], b["ntime"])
        out = orc.beamform_integrate(beams, ntime_sum)
        _np(out_arr, np.float32, out.size)[...] = out.ravel()
        return 0


# ---------------------------------------------------------------------------------------------------------------------
# The native per-gulp loops (csrc/pyext/xfast.cpp BeamPump / CorrPump) on CPU rings: a table of ctypes callbacks in the order
# of xfast.cpp's ComputeOps, so that the pumps call THIS backend (the oracle, on system-space span memory) where they would
# call libxeng.  Round-4 review W2: their error paths were reached on the GPU box only.
class _Ptr:
    """what `_np` needs of an `as_BFarray()` reference: the address"""
    def __init__(self, ptr):
        self.data = ptr


class _Bytes:
    """what the slab / parts calls need of an XArray: numpy() and ptr"""
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes
        self._a = np.frombuffer((ctypes.c_char * nbytes).from_address(ptr), dtype=np.uint8) if ptr else None

    def numpy(self):
        return self._a


class _Stamp(ctypes.Structure):
    _fields_ = [("w", ctypes.c_ulonglong * 16)]


_vp, _i, _ll, _sz, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t, ctypes.c_uint64
_COMPUTE_SIGNATURES = [
    ("beam_run_versioned", [_vp, _vp, _vp, _ll]), ("beam_run_parts", [_vp, _i, _vp, _vp, _vp, _ll]),
    ("beam_run_slabs", [_vp, _i, _i, _vp, _i, _sz, _u64, _i, _vp, _vp, _ll]), ("beam_integrate", [_vp, _vp, _i]),
    ("beam_mark", [ctypes.POINTER(ctypes.c_ulonglong)]), ("beam_wait", [ctypes.c_ulonglong]), ("beam_sync", []),
    ("memcpy_async", [_vp, _vp, _sz]), ("stamp_now_for", [ctypes.POINTER(_Stamp), _vp, ctypes.c_uint]),
    ("stamp_done", [ctypes.POINTER(_Stamp), ctypes.POINTER(_i), ctypes.POINTER(_i)]), ("stamp_wait", [ctypes.POINTER(_Stamp)]),
    ("dev_malloc", [ctypes.POINTER(_vp), _sz, _i]), ("dev_free", [_vp, _i]),
    ("xgpu_try_kernel", [_vp, _vp, _i, _vp, _i]), ("xgpu_try_kernel_slab", [_vp, _i, _sz, _u64, _i, _vp, _i, _vp, _i]),
    ("xgpu_wait_slot", []), ("xgpu_sync_lag", [_i]), ("xgpu_sync", []), ("xgpu_reset", []),
]


class PumpOracleBackend(OracleBackend):
    """OracleBackend whose beam_pump / corr_pump build the NATIVE pumps with a compute table that calls back into it.
    `fail`: {entry name: call number (1-based) at which that entry returns an error status} -- failure injection."""
    ERR = 5

    def __init__(self, fail=None, slab=None):
        super().__init__()
        self.fail = dict(fail or {})
        self.calls = {}
        self.callback_errors = []
        self.slab = slab                     # (npkt, stride, slab_ntime) of a slab sequence: sizes of the parts handed to run_slabs
        self._staged = {}
        self._tickets = 0
        self._copies = 0
        self._keep = []
        self._table = None

    # ---- the table
    def compute_table(self):
        if self._table is None:
            ptrs = (ctypes.c_void_p * len(_COMPUTE_SIGNATURES))()
            for k, (name, args) in enumerate(_COMPUTE_SIGNATURES):
                cb = ctypes.CFUNCTYPE(ctypes.c_int, *args)(self._guard(name, getattr(self, "_op_" + name)))
                self._keep.append(cb)
                ptrs[k] = ctypes.cast(cb, ctypes.c_void_p).value
            self._table = bytes(ptrs)
        return self._table

    def _guard(self, name, fn):
        def call(*a):
            n = self.calls[name] = self.calls.get(name, 0) + 1
            if self.fail.get(name) == n:
                return self.ERR
            try:
                return int(fn(*a) or 0)
            except Exception as e:              # (an exception cannot cross the C frames of the pump)
                self.callback_errors.append((name, repr(e)))
                return 3
        return call

    def beam_pump(self, iring, reader, oring, oseq_id, igulp, ogulp, mode, row_bytes=0, ntime_sum=0, depth=8, staged=False):
        if not (hasattr(iring, "_h") and hasattr(oring, "_h")):
            return None
        from caltech_bifrost_dsp_amd.ring import _xfast
        self.pump_args = dict(igulp=igulp, ogulp=ogulp, mode=mode, row_bytes=row_bytes)
        return _xfast().beam_pump(iring, iring._h, int(reader), oring, oring._h, int(oseq_id), int(igulp), int(ogulp), int(mode), int(row_bytes),
                                  int(ntime_sum), int(depth), int(bool(staged)), self.compute_table())

    def corr_pump(self, iring, reader, oring, igulp, ogulp, ntime_gulp):
        if not (hasattr(iring, "_h") and hasattr(oring, "_h")):
            return None
        from caltech_bifrost_dsp_amd.ring import _xfast
        self.corr_pumps = getattr(self, "corr_pumps", 0) + 1
        return _xfast().corr_pump(iring, iring._h, int(reader), oring, oring._h, int(igulp), int(ogulp), int(ntime_gulp), self.compute_table())

    # ---- beamformer entries
    def _op_beam_run_versioned(self, vin, out, w, version):
        return self.bfBeamformRun(_Ptr(vin), _Ptr(out), _Ptr(w), version=version)

    def _op_beam_run_parts(self, in0, ntime0, in1, out, w, version):
        b = self.beam
        row = b["nchan"] * b["ninput"]
        return self.bfBeamformRunParts(_Bytes(in0, ntime0 * row), _Bytes(in1, (b["ntime"] - ntime0) * row), _Ptr(out), _Ptr(w), version=version)

    def _op_beam_run_slabs(self, pk0, npkt0, ntime0, pk1, npkt1, stride, seq0, chan0, out, w, version):
        self.slab_seq0 = getattr(self, "slab_seq0", []) + [int(seq0)]
        s0 = _Bytes(pk0, npkt0 * stride)
        s1 = _Bytes(pk1, npkt1 * stride) if pk1 else None
        return self.bfBeamformRunSlabs(s0, npkt0, ntime0, s1, npkt1, stride, seq0, chan0, _Ptr(out), _Ptr(w), version=version)

    def _op_beam_integrate(self, vin, out, ntime_sum):
        return self.bfBeamformIntegrate(_Ptr(vin), _Ptr(out), ntime_sum)

    def _op_beam_mark(self, t):
        self._tickets += 1
        t[0] = self._tickets
        return 0

    def _op_beam_wait(self, t):
        self.waits = getattr(self, "waits", []) + [int(t)]
        return 0

    def _op_beam_sync(self):
        self.syncs = getattr(self, "syncs", 0) + 1
        return 0

    def _op_memcpy_async(self, dst, src, n):
        ctypes.memmove(dst, src, n)
        return 0

    def _op_stamp_now_for(self, st, buf, classes):
        self._copies += 1
        st[0].w[0] = self._copies
        return 0

    def _op_stamp_done(self, st, done, waitable):
        done[0] = 1
        if waitable:
            waitable[0] = 1
        return 0

    def _op_stamp_wait(self, st):
        return 0

    def _op_dev_malloc(self, out, n, space):
        buf = (ctypes.c_char * max(int(n), 1))()
        addr = ctypes.addressof(buf)
        self._staged[addr] = buf
        out[0] = addr
        return 0

    def _op_dev_free(self, p, space):
        self._staged.pop(p, None)
        return 0

    # ---- X-engine entries
    def _op_xgpu_try_kernel(self, vin, out, dump, acc, mode):
        return self.bfXgpuKernelAsync(_Ptr(vin), _Ptr(out), dump)

    def _op_xgpu_try_kernel_slab(self, pk, npkt, stride, seq0, chan0, out, dump, acc, mode):
        self.slab_seq0 = getattr(self, "slab_seq0", []) + [int(seq0)]
        return self.bfXgpuKernelSlab(_Bytes(pk, npkt * stride), npkt, stride, seq0, chan0, _Ptr(out), dump)

    def _op_xgpu_wait_slot(self):
        return 0

    def _op_xgpu_sync_lag(self, lag):
        self.lag_syncs = getattr(self, "lag_syncs", 0) + 1
        return 0

    def _op_xgpu_sync(self):
        self.xsyncs = getattr(self, "xsyncs", 0) + 1
        return 0

    def _op_xgpu_reset(self):
        return self.xgpu_reset()
