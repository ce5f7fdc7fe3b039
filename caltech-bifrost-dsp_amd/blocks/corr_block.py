"""Corr: 4+4-bit voltages from a GPU ring -> xGPU-order int32 visibilities on a GPU ring.

Drop-in counterpart of pipeline/lwa352_pipeline/blocks/corr_block.py (class Corr :60-472):
same constructor signature (:221-222), command keys `start_time` / `acc_len` with the same
conditions (:240-243), header handling (:355-384), one output span per integration (:433-452),
perf/stats keys (:453-458), `regtile_index` (:27-58) and `update_baseline_indices` (:317-333).
Underneath, `_bf.bfXgpuKernel` is the hand-written HIP X-engine of libxeng (csrc/xcorr*.h*).

Divergences, all documented in DESIGN.md: `acc_len == 0` is a clean stop (the reference falls
through to `WriteSpan(oseq.ring ...)` with `oseq = None`, corr_block.py:424-435, and divides by
zero for `start_time = -1`, :398); an integration interrupted by a new command is dropped with
xengXgpuReset instead of leaking into the next one.
"""
import ctypes
import json
import time

import numpy as np

from ..backend import default_backend
from ..ndarray import XArray
from ..proclog import cpu_affinity
from ..ring import WriteSpan
from .block_base import Block, declare_streams
from .integration import IntegrationGate


def tri_index(i, j):
    """Triangular index of (i, j), valid for i >= j (corr_block.py:27-28)."""
    return (i * (i + 1)) // 2 + j


def regtile_index(in0, in1, nstand):
    """Word index of the real part of inputs (in0, in1), in1 >= in0, in the xGPU register-tile
    ordered buffer; the imaginary part is matlen words later (corr_block.py:37-58)."""
    a0, a1 = in0 >> 1, in1 >> 1
    p0, p1 = in0 & 1, in1 & 1
    quadrant_size = (nstand // 2 + 1) * nstand // 4
    cell = (2 * (a0 & 1) + (a1 & 1)) * quadrant_size + tri_index(a1 // 2, a0 // 2)
    return cell * 4 + 2 * p1 + p0


class Corr(Block):
    def __init__(self, log, iring, oring, ntime_gulp=2500,
                 guarantee=True, core=-1, nchan=192, npol=2, nstand=352, acc_len=2400, gpu=-1, test=False,
                 etcd_client=None, autostartat=0, ant_to_input=None, backend=None):
        assert (acc_len % ntime_gulp == 0), "Acculmulation length must be a multiple of gulp size"
        super(Corr, self).__init__(log, iring, oring, guarantee, core, etcd_client=etcd_client)
        self._bf = backend if backend is not None else default_backend()
        self.ntime_gulp = ntime_gulp
        self.nchan, self.npol, self.nstand = nchan, npol, nstand
        declare_streams(iring, 'xgpu')          # (gulps: read by whichever contraction they were registered for)
        declare_streams(oring, 'xgpu_out')      # (visibility spans: written by their own dump and by nothing else on those streams)
        self.matlen = nchan * (nstand // 2 + 1) * (nstand // 4) * npol * npol * 4
        self.gpu = gpu
        self.test = test
        self._pump_stop = ctypes.c_int(0)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)

        self.size_proclog.update({'nseq_per_gulp': self.ntime_gulp})
        self.igulp_size = self.ntime_gulp * nchan * nstand * npol * 1   # complex8
        self.ogulp_size = self.matlen * 8                               # complex64

        self.define_command_key('start_time', type=int, initial_val=autostartat,
                                condition=lambda x: (x == -1) or (x % self.ntime_gulp == 0))
        self.define_command_key('acc_len', type=int, initial_val=acc_len,
                                condition=lambda x: x % self.ntime_gulp == 0)
        self.update_stats({'xgpu_acc_len': self.ntime_gulp})

        # xGPU's compile-time sizes are runtime arguments here; the dummy arrays keep the call shape
        rv = self._bf.xgpu_configure(nstand, npol, nchan, ntime_gulp, min(10, max(1, acc_len // ntime_gulp)) if acc_len else 0)
        if rv == self._bf.BF_STATUS_SUCCESS:
            ibuf = XArray([0], dtype='i8', space='system')
            obuf = XArray([0], dtype='i64', space='system')
            rv = self._bf.bfXgpuInitialize(ibuf.as_BFarray(), obuf.as_BFarray(), self.gpu)
        if rv != self._bf.BF_STATUS_SUCCESS:
            self.log.error("xgpuIntialize returned %d" % rv)
            raise RuntimeError("xgpuInitialize returned %d: %s" % (rv, self._bf.last_error()))

        self.antpol_to_input = XArray(np.zeros([nstand, npol], dtype=np.int32), space='system')
        self.antpol_to_bl = XArray(np.zeros([nstand, nstand, npol, npol], dtype=np.int32), space='system')
        self.bl_is_conj = XArray(np.zeros([nstand, nstand, npol, npol], dtype=np.int32), space='system')
        if ant_to_input is not None:
            self.update_baseline_indices(ant_to_input)

    # `update_pending` is set by the command thread; the native per-gulp loop (backend.corr_pump) watches the word behind
    # `_pump_stop` and hands control back to this block's Python as soon as a gulp arrives with it up
    @property
    def update_pending(self):
        return self.__dict__.get('_update_pending', False)

    @update_pending.setter
    def update_pending(self, value):
        self.__dict__['_update_pending'] = value
        stop = self.__dict__.get('_pump_stop')
        if stop is not None and value:
            stop.value = 1

    # --- the reference's numpy self-test (corr_block.py:265-315), vectorised -----------------------
    def _test(self, din, nchan, nstand, npol):
        """CPU correlation of one gulp in the block's convention: out[c, s0, s1, 2*p0+p1] =
        sum_t conj(x[t,c,s0,p0]) * x[t,c,s1,p1]."""
        d = din.copy(space='system').numpy().view(np.uint8).reshape([self.ntime_gulp, nchan, nstand, npol])
        dr = (d >> 4).astype(np.int8)
        dr[dr > 7] -= 16
        di = (d & 0xf).astype(np.int8)
        di[di > 7] -= 16
        dc = dr.astype(np.float64) + 1j * di
        out = np.einsum('tcap,tcbq->cabpq', np.conj(dc), dc)
        return out.reshape(nchan, nstand, nstand, npol * npol)

    def _compare(self, din, dout, nchan, nstand, npol):
        """True iff the xGPU-order buffer `dout` equals the CPU result `din` for every s1 >= s0."""
        planar = dout.copy(space='system').numpy().view(np.int32).reshape(2, nchan, self.matlen // nchan)
        dout_c = planar[0] + 1j * planar[1]
        s0, s1, p0, p1 = np.meshgrid(np.arange(nstand), np.arange(nstand), [0, 1], [0, 1], indexing='ij')
        idx = regtile_index(2 * s0 + p0, 2 * s1 + p1, nstand)
        valid = s1 >= s0
        got = dout_c[:, np.where(valid, idx, 0)].reshape(nchan, nstand, nstand, npol * npol)
        ok = bool(np.all(got[:, valid.reshape(nstand, nstand, npol * npol)] ==
                         din[:, valid.reshape(nstand, nstand, npol * npol)]))
        self.log.info("CORR >> self-test MATCH? %s" % ok)
        self.update_stats({'test_match': ok})
        return ok

    def update_baseline_indices(self, ant_to_input):
        """[nstand x npol] input ids -> antpol_to_bl / bl_is_conj [nstand, nstand, npol, npol]
        (corr_block.py:317-333)."""
        cpu_affinity.set_core(self.core)
        self.antpol_to_input[...] = np.asarray(ant_to_input, dtype=np.int32)
        rv = self._bf.bfXgpuGetOrder(self.antpol_to_input.as_BFarray(), self.antpol_to_bl.as_BFarray(),
                                     self.bl_is_conj.as_BFarray())
        if rv != self._bf.BF_STATUS_SUCCESS:
            raise RuntimeError("xgpuGetOrder returned %d" % rv)

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})

        self.oring.resize(self.ogulp_size)
        time_tag = 1
        self._time_tag = 1
        gate = IntegrationGate(recovery_skip=10, round_start_to_acc_len=True)
        self.update_stats({'state': 'starting'})
        self._held = []
        self._pending = None
        try:
            self._main_loop(gate, time_tag)
        finally:
            # An exception must not release gulps or an output span that a dump in flight still reads or writes (their
            # memory would go back to the rings under the running kernel): wait for the X-engine first.
            if self._pending is not None or self._held:
                try:
                    self._bf.xgpu_sync()
                except Exception:
                    pass
                self._pending = None
                self._held = []

    def _main_loop(self, gate, time_tag):
        with self.oring.begin_writing() as oring:
            prev_time = time.time()
            self.update_pending = True
            oseq = ospan = None
            acquire_time = reserve_time = 0
            # In-place mode: when the input ring keeps a span's memory alive for as long as the span is referenced
            # (the in-repo Ring does; a bifrost ring is circular and does not), the gulps of an integration are only
            # registered (`bfXgpuKernelAsync`) and the contraction kernel reads them where they lie at dump time --
            # no copy of the voltages at all.  Otherwise the reference's synchronous call is used and libxeng keeps
            # a raw copy of every gulp until the dump.
            in_place = (getattr(self.iring, 'span_memory_outlives_release', False)
                        and hasattr(self._bf, 'bfXgpuKernelAsync') and hasattr(self._bf, 'xgpu_sync'))
            # ... and the block streams: the dump of integration n is only enqueued; its output span is committed
            # (and its gulps released) when the dump of integration n+1 has been enqueued and n's has completed
            # (xengXgpuSyncLag(1)), so the GPU never idles between integrations.  `_pending` = (span, held gulps) of
            # the integration whose dump is in flight.
            streaming = in_place and hasattr(self._bf, 'xgpu_sync_lag')
            # ... and feeds an attached CorrAcc from the dump's own epilogue (blocks/corr_acc_block.py, fused mode): the
            # CorrAcc that reads this block's output ring registered itself there; it decides, dump by dump, which
            # accumulator takes the dump and how (assign / add), and publishes when its long integration is complete
            long_acc = getattr(self.oring, 'long_accumulator', None) if (streaming and hasattr(self._bf, 'bfXgpuKernelAsyncAcc')
                                                                         and self._bf.xgpu_fused_acc_supported()) else None
            self.update_stats({'fused_corracc': long_acc is not None})
            self._held = []
            self._pending = None
            for iseq in self.iring.read(guarantee=self.guarantee):
                self.log.info('CORR >> new input sequence!')
                process_time = 0
                oseq = ospan = None
                ihdr = json.loads(iseq.header.tostring())
                now = ihdr['seq0']
                ohdr = ihdr.copy()
                if gate.recover(now):
                    self._abort_integration()
                    self.log.info("CORR >> Recovering start time set to %d. Accumulating %d samples" % (gate.start_time, gate.acc_len))
                    ohdr['acc_len'] = gate.acc_len
                    ohdr['seq0'] = gate.start_time
                # xGPU-order output is not an [nstand, npol]^2 array: drop the input maps (:381-384)
                ohdr.pop('ant_to_input', None)
                ohdr.pop('input_to_ant', None)
                self.sequence_proclog.update(ohdr)
                # A sequence of PACKET SLABS (Snap2Ingest(unpack=False), or a receiver that writes into a device ring): every
                # gulp is the slab of SNAP2 packets it arrived in, handed to the library as it is -- read in place by the
                # contraction when it is complete and in order, scattered on the device otherwise (xengXgpuKernelAsyncSlab).
                # The downstream header describes the visibilities, not the input layout.
                slab = ihdr.get('layout') == 'snap2_slab'
                igulp_size = self.igulp_size
                if slab:
                    if not in_place or self.test:
                        raise RuntimeError("CORR: packet-slab input needs the in-place (streaming) mode of the in-repo rings, and no test mode")
                    if ihdr.get('slab_ntime', self.ntime_gulp) != self.ntime_gulp:
                        raise RuntimeError("CORR: slabs of %d samples, ntime_gulp %d" % (ihdr['slab_ntime'], self.ntime_gulp))
                    slab_npkt, slab_stride, slab_chan0 = ihdr['npkt_per_gulp'], ihdr['pkt_stride'], ihdr['chan0']
                    igulp_size = slab_npkt * slab_stride
                    for k in ('layout', 'slab_ntime', 'npkt_per_gulp', 'pkt_stride'):
                        ohdr.pop(k, None)
                # The native per-gulp loop (csrc/pyext/xfast.cpp CorrPump; round 5): between native rings, streaming, with no CorrAcc
                # fused into the dumps (its decisions are taken per dump in Python) and outside the self-test mode
                pump = None
                if (streaming and long_acc is None and not self.test and hasattr(self._bf, 'corr_pump') and hasattr(iseq, '_rid')
                        and hasattr(self.oring, '_h')):
                    pump = self._bf.corr_pump(self.iring, iseq._rid, self.oring, igulp_size, self.ogulp_size, self.ntime_gulp)
                if pump is not None:
                    if slab:
                        pump.set_slabs(slab_npkt, slab_stride, slab_chan0)
                    self.update_stats({'pump': True})
                    time_tag = self._time_tag = max(time_tag, self._time_tag)
                    self._pump_sequence(pump, iseq, ihdr, ohdr, gate, now, oring, igulp_size)
                    time_tag = self._time_tag
                    continue
                for ispan in iseq.read(igulp_size):
                    if ispan.size < igulp_size:
                        self.log.info("CORR >>> Ignoring final gulp (expected %d bytes but got %d)" % (igulp_size, ispan.size))
                        continue
                    if getattr(ispan, 'skipped', 0):
                        # gulps this reader never saw (overwritten before it got to them; whole gulps, ring.py): the sample
                        # count moves on with them, and an integration they belonged to is lost -- realigned like a new
                        # upstream sequence (:360-371)
                        now += (ispan.skipped // igulp_size) * self.ntime_gulp
                        self.log.warning("CORR >> %d bytes of input were overwritten before they were read" % ispan.skipped)
                        if gate.recover(now):
                            self._abort_integration()
                            ospan = None
                            ohdr['acc_len'] = gate.acc_len
                            ohdr['seq0'] = gate.start_time
                    if self.update_pending:
                        self.update_command_vals()
                        if gate.running:
                            self._abort_integration()
                            ospan = None
                        gate.configure(now, self.command_vals['acc_len'], self.command_vals['start_time'])
                        self.log.info("CORR >> New start time set to %d. Accumulating %d samples" % (gate.start_time, gate.acc_len))
                        ohdr['acc_len'] = gate.acc_len
                        ohdr['seq0'] = gate.start_time
                    self.update_stats({'curr_sample': now})
                    if gate.acc_len == 0:
                        # acc_len = 0 is the stop command (:423-428); made a clean stop here
                        self.update_stats({'state': 'stopped'})
                        self._finish_pending()
                        if oseq:
                            oseq.end()
                        oseq = None
                        gate.running = False
                        now += self.ntime_gulp
                        continue
                    if gate.try_start(now, self.ntime_gulp):
                        self.log.info("CORR >> Start time %d reached." % gate.start_time)
                        self._finish_pending()
                        if oseq:
                            oseq.end()
                        self.sequence_proclog.update(ohdr)
                        if long_acc is not None:
                            ohdr_out = dict(ohdr, fused_corracc=1)
                            long_acc.plan_sequence(ohdr_out)
                        else:
                            ohdr_out = ohdr
                        oseq = oring.begin_sequence(time_tag=time_tag, header=json.dumps(ohdr_out), nringlet=iseq.nringlet)
                        time_tag += 1
                    if not gate.running:
                        self.update_stats({'state': 'waiting'})
                        now += self.ntime_gulp
                        continue
                    self.update_stats({'state': 'running'})
                    curr_time = time.time()
                    acquire_time = curr_time - prev_time
                    prev_time = curr_time
                    if now == gate.first:
                        ospan = WriteSpan(oseq.ring, self.ogulp_size, nonblocking=False)   # one span per integration
                        if self.test:
                            test_out = np.zeros([ihdr['nchan'], ihdr['nstand'], ihdr['nstand'], ihdr['npol'] ** 2], dtype=complex)
                        curr_time = time.time()
                        reserve_time = curr_time - prev_time
                        prev_time = curr_time
                    if not ospan:
                        self.log.error("CORR: trying to write to not-yet-opened ospan")
                        now += self.ntime_gulp
                        continue
                    if self.test:
                        test_out += self._test(ispan.data, ihdr['nchan'], ihdr['nstand'], ihdr['npol'])
                    if in_place:
                        self._held.append(ispan.data)      # keeps the gulp's memory alive until the dump has run
                        acc = None
                        if now == gate.last and long_acc is not None:
                            acc, acc_mode = long_acc.plan_dump()
                        if slab:
                            rv = self._bf.bfXgpuKernelSlab(ispan.data, slab_npkt, slab_stride, now, slab_chan0, ospan.data.as_BFarray(),
                                                           int(now == gate.last), acc, acc_mode if acc is not None else 0)
                        elif acc is not None:
                            rv = self._bf.bfXgpuKernelAsyncAcc(ispan.data.as_BFarray(), ospan.data.as_BFarray(), 1, acc, acc_mode)
                        else:
                            rv = self._bf.bfXgpuKernelAsync(ispan.data.as_BFarray(), ospan.data.as_BFarray(), int(now == gate.last))
                        if rv == self._bf.BF_STATUS_SUCCESS and now == gate.last:
                            if streaming:
                                prev, self._pending = self._pending, (ospan, self._held)
                                self._held = []
                                if prev is not None:
                                    rv = self._bf.xgpu_sync_lag(1)     # the previous dump is complete: commit its span
                                    prev[0].close()
                            else:
                                rv = self._bf.xgpu_sync()      # the output span is complete before it is committed
                                self._held = []
                    else:
                        rv = self._bf.bfXgpuKernel(ispan.data.as_BFarray(), ospan.data.as_BFarray(), int(now == gate.last))
                    if rv != self._bf.BF_STATUS_SUCCESS:
                        raise RuntimeError("xgpuKernel returned %d: %s" % (rv, self._bf.last_error()))
                    curr_time = time.time()
                    process_time += curr_time - prev_time
                    prev_time = curr_time
                    if now == gate.last:
                        if self.test:
                            self._finish_pending()
                            self._compare(test_out, ospan.data, ihdr['nchan'], ihdr['nstand'], ihdr['npol'])
                        if self._pending is None or self._pending[0] is not ospan:
                            ospan.close()
                        ospan = None
                        gbps = 8 * gate.acc_len * ihdr['nchan'] * ihdr['nstand'] * ihdr['npol'] / max(process_time, 1e-9) / 1e9
                        self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                                  'process_time': process_time, 'gbps': gbps})
                        self.update_stats({'last_end_sample': now, 'throughput': gbps})
                        process_time = 0
                        gate.advance(self.ntime_gulp)
                    now += self.ntime_gulp
                if ospan is not None:
                    self._abort_integration()       # upstream sequence ended mid-integration
                    ospan = None
                self._finish_pending()
                if oseq:
                    oseq.end()
                oseq = None

    PUMP_INTEGRATIONS = 4       # integrations per call of the native loop while no command is pending

    def _pump_sequence(self, pump, iseq, ihdr, ohdr, gate, now, oring, igulp_size):
        """One input sequence through the native per-gulp loop.  What stays here is what the loop above decides per integration or
        rarer -- commands, the gate's start / stop / recovery, output sequences and their headers, statistics -- in the same order
        per gulp: lost input, pending command, stop, start, waiting, running (corr_block.py:388-466)."""
        g = self.ntime_gulp
        stop_addr = ctypes.addressof(self._pump_stop)
        FOREVER = 1 << 40
        oseq = None
        prev_time = time.time()
        try:
            while True:
                self._pump_stop.value = 0       # (lowered BEFORE the flag is read: a command that comes in from here on raises it again)
                if self.update_pending:
                    self.update_command_vals()
                    if gate.running:
                        pump.abort()
                    gate.configure(now, self.command_vals['acc_len'], self.command_vals['start_time'])
                    self.log.info("CORR >> New start time set to %d. Accumulating %d samples" % (gate.start_time, gate.acc_len))
                    ohdr['acc_len'] = gate.acc_len
                    ohdr['seq0'] = gate.start_time
                self.update_stats({'curr_sample': now})
                nint = 0
                if gate.acc_len == 0:
                    # acc_len = 0 is the stop command (:423-428); a clean stop: gulps pass by until the next command
                    self.update_stats({'state': 'stopped'})
                    pump.finish()
                    if oseq:
                        oseq.end()
                    oseq = None
                    gate.running = False
                    n, skipped, status, _ = pump.run(0, FOREVER, stop_addr, 0, 1, now)
                elif not gate.running:
                    if gate.try_start(now, g):
                        self.log.info("CORR >> Start time %d reached." % gate.start_time)
                        pump.finish()
                        if oseq:
                            oseq.end()
                        self.sequence_proclog.update(ohdr)
                        oseq = oring.begin_sequence(time_tag=self._time_tag, header=json.dumps(ohdr), nringlet=iseq.nringlet)
                        self._time_tag += 1
                        pump.set_output(oseq._seq_id)
                        continue
                    self.update_stats({'state': 'waiting'})
                    n, skipped, status, _ = pump.run(0, (gate.start_time - now) // g if gate.start_time > now else FOREVER, stop_addr, 0, 1, now)
                else:
                    self.update_stats({'state': 'running'})
                    gpi = gate.acc_len // g
                    pos = (now - gate.first) // g
                    n, skipped, status, nint = pump.run(1, self.PUMP_INTEGRATIONS * gpi - pos, stop_addr, pos, gpi, now)
                now += n * g
                if nint:
                    for _ in range(nint):
                        gate.advance(g)
                    curr_time = time.time()
                    process_time = (curr_time - prev_time) / nint
                    gbps = 8 * gate.acc_len * ihdr['nchan'] * ihdr['nstand'] * ihdr['npol'] / max(process_time, 1e-9) / 1e9
                    self.perf_proclog.update({'acquire_time': 0.0, 'reserve_time': 0.0, 'process_time': process_time, 'gbps': gbps})
                    self.update_stats({'last_end_sample': gate.first - g, 'throughput': gbps})
                prev_time = time.time()
                if status == 3:
                    # gulps this reader never saw (overwritten before it got to them; whole gulps): the sample count moves on with
                    # them, and an integration they belonged to is lost -- realigned like a new upstream sequence (:360-371)
                    now += (skipped // igulp_size) * g
                    self.log.warning("CORR >> %d bytes of input were overwritten before they were read" % skipped)
                    if gate.recover(now):
                        pump.abort()
                        ohdr['acc_len'] = gate.acc_len
                        ohdr['seq0'] = gate.start_time
                if status == 1:
                    break
            held = pump.state()
            if held[1] or held[2]:
                pump.abort()                    # upstream sequence ended mid-integration
            pump.finish()
            if oseq:
                oseq.end()
        except BaseException:
            # (nothing in flight may still read a gulp or write a span when they go back to their rings)
            try:
                pump.close(False)
            except Exception:
                pass
            raise
        pump.close(True)

    def _finish_pending(self):
        """Streaming mode: wait for the dump in flight and commit its span."""
        pend, self._pending = getattr(self, '_pending', None), None
        if pend is not None:
            rv = self._bf.xgpu_sync()
            pend[0].close()
            if rv != self._bf.BF_STATUS_SUCCESS:
                raise RuntimeError("xgpuSync returned %d: %s" % (rv, self._bf.last_error()))

    def _abort_integration(self):
        self._finish_pending()                      # a completed integration whose dump is in flight is still good
        reset = getattr(self._bf, 'xgpu_reset', None)
        if reset is not None:
            reset()                                 # synchronises: nothing reads the held gulps afterwards
        self._held = []
