"""The object the blocks call where the reference calls `_bf` / `bifrost.device` / `bifrost.map`.

`HipBackend` forwards to libxeng.so (include/xeng.h) and is the only backend the product ships:
constructing it without the HIP library or without a GPU raises.  Tests may inject a different
object with the same methods (tests/fake_backend.py wraps the CPU oracle) to exercise the block
state machines without a GPU -- that is test infrastructure, never a fallback.
"""
import ctypes

from . import ffi


def _dev(a):
    """Device address behind an `as_BFarray()` reference (or a plain ctypes pointer to an XENGarray)."""
    d = getattr(a, "data", None)
    return d if d is not None else a.contents.data


class HipBackend:
    BF_STATUS_SUCCESS = ffi.STATUS_SUCCESS
    space_in = "cuda"          # memory space the compute entry points expect

    def __init__(self):
        self._lib = ffi.lib()   # raises ImportError if libxeng.so has not been built
        self._enq = ffi.enqueue_lib()   # enqueue-only calls, made without giving up the interpreter lock (ffi.ENQUEUE_ONLY)
        # ... and the ones a block makes per gulp bound directly (csrc/pyext/xfast.cpp): ctypes spends more on converting
        # their arguments than the library spends on the call
        from .ring import _xfast
        self._x = _xfast()

    # ---- device plumbing (bifrost.device.set_device / get_device / stream_synchronize)
    def set_device(self, gpu):
        ffi.call("xengSetDevice", int(gpu))

    def get_device(self):
        import ctypes
        g = ctypes.c_int()
        ffi.call("xengGetDevice", ctypes.byref(g))
        return g.value

    def stream_synchronize(self):
        """Every library stream of the device (bifrost.device.stream_synchronize has one stream to wait for; here each
        block works on its own, so the blocks use the per-stream waits below and leave the others running)."""
        ffi.call("xengStreamSynchronize")

    def map_sync(self):
        """CorrAcc's stream (corr_acc_block.py:317)."""
        ffi.call("xengMapSync")

    def beam_sync(self):
        """The beamformer's stream: Beamform / BeamformSumBeams (beamform_block.py:450, beamform_sum_beams_block.py:247)."""
        ffi.call("xengBeamformSync")

    def beam_mark(self):
        """Ticket for everything enqueued on the beamformer's stream so far (beam_wait waits for it)."""
        t = self._x.beam_mark()
        if t < 0:
            ffi.check("xengBeamformMark", -t)
        return t

    def beam_wait(self, ticket):
        # ask first, without giving up the interpreter lock; only a ticket the GPU has not reached yet is worth a blocking call
        d = self._x.beam_ticket_done(ticket)
        if d < 0:
            ffi.check("xengBeamformTicketDone", -d)
        if not d:
            ffi.call("xengBeamformWait", ticket)

    # ---- X-engine (corr_block.py:253,331,445)
    def xgpu_configure(self, nstand, npol, nchan, ntime_gulp, max_gulps=0):
        """xGPU's compile-time NSTATION/NFREQUENCY/NTIME (install_xgpu.sh:5) are runtime here."""
        return self._lib.xengXgpuConfigure(nstand, npol, nchan, ntime_gulp, max_gulps)

    def bfXgpuInitialize(self, in_arr, out_arr, gpu):
        return self._lib.bfXgpuInitialize(in_arr, out_arr, int(gpu))

    def bfXgpuKernel(self, in_arr, out_arr, do_dump):
        return self._lib.bfXgpuKernel(in_arr, out_arr, int(do_dump))

    def bfXgpuKernelAsync(self, in_arr, out_arr, do_dump):
        """Enqueue only: the gulp is read in place at dump time, so the caller keeps it alive and unchanged until
        xgpu_sync() (include/xeng.h: xengXgpuKernelAsync).  No reference counterpart."""
        return self._x.xgpu_kernel_async(_dev(in_arr), _dev(out_arr), int(do_dump))

    def bfXgpuKernelAsyncAcc(self, in_arr, out_arr, do_dump, acc, acc_mode):
        """bfXgpuKernelAsync whose dump also assigns (acc_mode 1) / adds (2) every stored word to the long accumulator
        `acc` -- CorrAcc's "a = b" / "a += b" (corr_acc_block.py:304-306) done by the contraction's epilogue."""
        return self._x.xgpu_kernel_async_acc(_dev(in_arr), _dev(out_arr), int(do_dump), acc.ptr, int(acc_mode))

    def bfXgpuKernelSlab(self, slab, npkt, pkt_stride, seq0, chan0, out_arr, do_dump, acc=None, acc_mode=0):
        """bfXgpuKernelAsync[Acc] on a gulp handed over as the slab of SNAP2 packets it arrived in (include/xeng.h
        xengXgpuKernelAsyncSlab): read in place when the slab is complete and in order, scattered on the device otherwise.
        No reference counterpart: bifrost's capture scatters on the CPU (capture_block.py:221-305)."""
        return self._x.xgpu_kernel_slab(slab.ptr, int(npkt), int(pkt_stride), int(seq0), int(chan0), _dev(out_arr), int(do_dump),
                                        acc.ptr if acc is not None else 0, int(acc_mode))

    def xgpu_fused_acc_supported(self):
        """True when the live X-engine context runs the default (fused corner turn) contraction kernel, the one whose
        epilogue can feed a long accumulator; other gulp shapes take the two-pass path and CorrAcc keeps its map."""
        import ctypes
        fused, fp6 = ctypes.c_int(), ctypes.c_int()
        rc = self._lib.xengXgpuGetPath(ctypes.byref(fused), ctypes.byref(fp6))
        return rc == ffi.STATUS_SUCCESS and fused.value == 1 and fp6.value == 0

    def xgpu_sync(self):
        return self._lib.xengXgpuSync()

    def xgpu_sync_lag(self, lag):
        """Wait until the dump issued `lag` dumps before the latest one is complete (lag 0 = the latest)."""
        d = self._x.xgpu_dump_done(int(lag))       # (asked without giving up the interpreter lock)
        if d:
            return ffi.STATUS_SUCCESS if d > 0 else -d
        return self._lib.xengXgpuSyncLag(int(lag))

    def bfXgpuGetOrder(self, antpol_to_input, antpol_to_bl, is_conj):
        return self._lib.bfXgpuGetOrder(antpol_to_input, antpol_to_bl, is_conj)

    def bfXgpuSubSelect(self, in_arr, out_arr, vismap, conj, nchan_sum, unused=0):
        return self._lib.bfXgpuSubSelect(in_arr, out_arr, vismap, conj, int(nchan_sum), int(unused))

    def xgpu_packetize(self, in_arr, out_arr, antpol_to_bl, is_conj, fmt):
        """Device reorder + per-baseline payloads of CorrOutputFull (corr_output_full_block.py:669, 461-467, 512-519)."""
        return self._lib.xengXgpuPacketize(in_arr.ptr, out_arr.ptr, antpol_to_bl.ptr, is_conj.ptr, int(fmt))

    def xgpu_reset(self):
        """Drop staged gulps / partial sums of an aborted integration (no reference counterpart)."""
        return self._lib.xengXgpuReset()

    # ---- ingest: SNAP2 packets -> gulp (the scatter bifrost's UDP capture does on the CPU; capture_block.py:296-305)
    def snap2_unpack(self, packets, npkt, pkt_stride, out, seq0, ntime, chan0, nchan_tot, npol_tot, clear=True):
        """Returns (status, packets placed, packets dropped)."""
        import ctypes
        placed, dropped = ctypes.c_int(), ctypes.c_int()
        rc = self._lib.xengSnap2Unpack(packets.ptr, int(npkt), int(pkt_stride), out.ptr, int(seq0), int(ntime), int(chan0),
                                       int(nchan_tot), int(npol_tot), int(bool(clear)), ctypes.byref(placed), ctypes.byref(dropped))
        return rc, placed.value, dropped.value

    # ---- CorrAcc (corr_acc_block.py:304,306: BFMap "a = b" / "a += b")
    def map_assign_i32(self, a, b):
        return self._x.map_i32(a.ptr, b.ptr, a.nbytes // 4, False)

    def map_add_i32(self, a, b):
        return self._x.map_i32(a.ptr, b.ptr, a.nbytes // 4, True)

    def map_sum_i32(self, a, srcs, add):
        """a = (add ? a : 0) + srcs[0] + ... + srcs[-1] in one pass (include/xeng.h xengMapSumI32): the long accumulation of a
        GROUP of dumps, their spans read once each.  Enqueued on the map stream like the two calls above; the caller may let go of
        the source spans at once on in-repo rings (released memory is reissued only behind its stamp)."""
        n = len(srcs)
        arr = (ctypes.c_void_p * n)(*[x.ptr for x in srcs])
        return self._enq.xengMapSumI32(a.ptr, arr, n, a.nbytes // 4, int(bool(add)))

    # ---- a copy that is only enqueued (CorrAcc's publish of a long integration: 383 MB over PCIe, 7 ms)
    def copy_async(self, dst, src):
        """Enqueue dst <- src on the library's copy stream; returns the stamp that completes when the copy has (copy_done /
        copy_wait).  The caller keeps both arrays alive and unchanged until then."""
        assert dst.nbytes == src.nbytes
        return self._x.copy_async(dst.ptr, src.ptr, dst.nbytes)

    def copy_done(self, stamp):
        return self._x.stamp_done(stamp)

    def copy_wait(self, stamp):
        self._x.stamp_wait(stamp)

    # ---- beamformer (beamform_block.py:251,449; beamform_sum_beams_block.py:245)
    _beam_row_bytes = 0

    def bfBeamformInitialize(self, gpu, ninput, nchan, ntime, nbeam, ntime_blocks):
        self._beam_row_bytes = int(ninput) * int(nchan)         # bytes per sample of a gulp (4+4 bit per input)
        return self._lib.bfBeamformInitialize(int(gpu), ninput, nchan, ntime, nbeam, ntime_blocks)

    def bfBeamformRun(self, in_arr, out_arr, weights, version=0):
        """`version` != 0 lets the library reuse its bf16-split copy of the weights while the caller has
        not changed them (the reference call shape has no such argument: version 0 = always re-split)."""
        if version:
            return self._x.beam_run(_dev(in_arr), _dev(out_arr), _dev(weights), int(version))
        return self._lib.bfBeamformRun(in_arr, out_arr, weights)

    def bfBeamformRunParts(self, part0, part1, out_arr, weights, version=0):
        """One beamformer gulp out of two consecutive spans of the input ring (arrays `part0`, `part1`: whole samples each),
        one launch, no gathered copy (include/xeng.h xengBeamformRunParts).  No reference counterpart: bifrost's circular ring
        hands the reference's 2-gulp read (lwa352-pipeline.py:172,279-282) out contiguously."""
        ntime0 = part0.nbytes // (self._beam_row_bytes or 1)
        return self._x.beam_run_parts(part0.ptr, ntime0, part1.ptr, _dev(out_arr), _dev(weights), int(version))

    def bfBeamformRunSlabs(self, slab0, npkt0, ntime0, slab1, npkt1, pkt_stride, seq0, chan0, out_arr, weights, version=0):
        """bfBeamformRun on a gulp handed over as one (slab1 None) or two consecutive slabs of SNAP2 packets (include/xeng.h
        xengBeamformRunSlabs)."""
        return self._x.beam_run_slabs(slab0.ptr, int(npkt0), int(ntime0), slab1.ptr if slab1 is not None else 0, int(npkt1), int(pkt_stride),
                                      int(seq0), int(chan0), _dev(out_arr), _dev(weights), int(version))

    def beam_pump(self, iring, reader, oring, oseq_id, igulp, ogulp, mode, row_bytes=0, ntime_sum=0, depth=8, staged=False):
        """The steady-state per-gulp loop of Beamform (mode 0) / BeamformSumBeams (mode 1) between two NATIVE rings as an object
        whose run() works without the interpreter lock (csrc/pyext/xfast.cpp BeamPump); None when the rings are not native
        or XENG_PUMP=0."""
        import os
        if os.environ.get("XENG_PUMP") == "0" or not (hasattr(iring, "_h") and hasattr(oring, "_h")):
            return None
        return self._x.beam_pump(iring, iring._h, int(reader), oring, oring._h, int(oseq_id), int(igulp), int(ogulp), int(mode), int(row_bytes),
                                 int(ntime_sum), int(depth), int(bool(staged)))

    def corr_pump(self, iring, reader, oring, igulp, ogulp, ntime_gulp):
        """The per-gulp loop of Corr while it integrates, between two NATIVE rings, as an object whose run() works without the
        interpreter lock (csrc/pyext/xfast.cpp CorrPump); None when the rings are not native or XENG_PUMP=0."""
        import os
        if os.environ.get("XENG_PUMP") == "0" or not (hasattr(iring, "_h") and hasattr(oring, "_h")):
            return None
        return self._x.corr_pump(iring, iring._h, int(reader), oring, oring._h, int(igulp), int(ogulp), int(ntime_gulp))

    def bfBeamformIntegrate(self, in_arr, out_arr, ntime_sum):
        # (bfBeamformIntegrate reads only the two data pointers from its structs: the raw entry point, no structs built per gulp)
        return self._x.beam_integrate(_dev(in_arr), _dev(out_arr), int(ntime_sum))

    def last_error(self):
        return self._lib.xengGetLastError().decode()


_default = None


def default_backend():
    global _default
    if _default is None:
        _default = HipBackend()
    return _default
