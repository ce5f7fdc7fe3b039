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
