"""Beamform: voltage beams from 4+4-bit F-engine data on the GPU.

Drop-in counterpart of pipeline/lwa352_pipeline/blocks/beamform_block.py (class Beamform,
constructor :208, command handling :269-362, main :364-461): same constructor signature, the
`coeffs` command key with its two payload types (`calgains`, `beamcoeffs`), the three-stage
gain buffers (new -> cpu -> gpu) with per-beam timed activation (`load_sample`, :416-434), the
output header rewrite (:403-409) and one `_bf.bfBeamformRun` per gulp (:449) -- which here is
libxeng's fused nibble-decode + fp32-MFMA beamformer (csrc/beamform_kernels.h).

out[c, b, t] = sum_i gains[c, b, i] * x[t, c, i], cf32 [nchan, nbeam, ntime_gulp]
(beamformer_test.py:76-84).
"""
import collections
import json
import time

import numpy as np

from ..backend import default_backend
from ..ndarray import XArray
from ..proclog import cpu_affinity
from .block_base import Block, COMMAND_INVALID, COMMAND_OK, declare_streams


class Beamform(Block):
    STREAM_DEPTH = 8        # gulps whose kernels may be in flight behind the one being enqueued (streaming mode; 4 -> 8: -2.4 %, profiles/r04/blocks_depth_sweep.txt)

    def __init__(self, log, iring, oring, nchan=256, nbeam=1, ninput=352 * 2, ntime_gulp=2500, ntime_sum=None,
                 guarantee=True, core=-1, gpu=-1, etcd_client=None, backend=None):
        super(Beamform, self).__init__(log, iring, oring, guarantee, core, etcd_client=etcd_client)
        self._bf = backend if backend is not None else default_backend()
        self.ntime_gulp = ntime_gulp
        self.gpu = gpu
        self.ntime_sum = ntime_sum
        declare_streams(iring, 'beam')          # (both rings' spans are touched by the beamformer's stream only)
        declare_streams(oring, 'beam')
        if ntime_sum is not None:
            assert ntime_gulp % ntime_sum == 0
            self.ntime_blocks = ntime_gulp // ntime_sum
        else:
            self.ntime_blocks = ntime_gulp
        self.nchan, self.nbeam, self.ninput = nchan, nbeam, ninput
        self.freqs = np.zeros(self.nchan, dtype=np.float32)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        # gains: latest commanded (new) -> waiting for its load time (cpu) -> on the device (gpu)
        self.cal_gains = np.ones((nchan, nbeam, ninput), dtype=np.complex64)
        self.gains_cpu_new = np.zeros((nchan, nbeam, ninput), dtype=np.complex64)
        self.gains_cpu = np.zeros((nchan, nbeam, ninput), dtype=np.complex64)
        self.gains_gpu = XArray(shape=(nchan, nbeam, ninput), dtype=np.complex64, space=self._bf.space_in)
        self.gains_load_sample = np.zeros(nbeam)
        self._gains_version = 0       # bumped whenever gains_gpu is rewritten
        import ctypes
        self._pump_stop = ctypes.c_int(0)
        self.define_command_key('coeffs', type=dict, initial_val={})
        for b in range(self.nbeam):
            self.update_stats({'cal_gains%d' % b: [False, ] * ninput})
        if ntime_sum is not None:
            self.log.warning("Running Beamform block with ntime_sum != None is experimental!")
            rv = self._bf.bfBeamformInitialize(self.gpu, self.ninput, self.nchan, self.ntime_gulp, self.nbeam, self.ntime_blocks)
        else:
            rv = self._bf.bfBeamformInitialize(self.gpu, self.ninput, self.nchan, self.ntime_gulp, self.nbeam, 0)
        if rv != self._bf.BF_STATUS_SUCCESS:
            raise RuntimeError("bfBeamformInitialize returned %d: %s" % (rv, self._bf.last_error()))

    # `update_pending` is set by the command thread; the native per-gulp loop (backend.beam_pump) watches the word behind
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

    def _etcd_callback(self, watchresponse):
        """Every command is enacted immediately (all coefficient commands share one key, so a later
        one must not overwrite an earlier one before it is processed; beamform_block.py:269-318)."""
        cpu_affinity.set_core(self.core)
        self.acquire_control_lock()
        try:
            for event in watchresponse.events:
                try:
                    seq_id, kwargs = self._parse_event(event)
                except ValueError:
                    self.log.exception("BEAMFORM >> Failed to JSON-decode event %s" % str(event.value))
                    self._send_command_response("0", False, "JSON-decode failed!")
                    continue
                if kwargs is None:
                    continue
                try:
                    proc_ok = self._process_commands(kwargs, set_pending_flag=False)
                except Exception:
                    proc_ok = COMMAND_INVALID
                self.update_stats({'last_cmd_response': proc_ok})
                self.update_command_vals()
                self._send_command_response(seq_id, proc_ok == COMMAND_OK, str(proc_ok))
        finally:
            self.release_control_lock()

    def update_command_vals(self):
        """Apply pending `coeffs` commands (called with the control lock held; :320-362)."""
        cpu_affinity.set_core(self.core)
        self.command_vals.update(self._pending_command_vals)
        update_beam_cal_state = False
        for k, v in self._pending_command_vals.items():
            try:
                if not v:
                    continue
                if v['type'] == 'calgains':
                    i, b = v['input_id'], v['beam_id']
                    data = np.array(v['data'])
                    self.cal_gains[:, b, i] = data[0::2] + 1j * data[1::2]      # freq x beam x input
                    self.stats['cal_gains%d' % b][i] = True
                    update_beam_cal_state = True
                if v['type'] == 'beamcoeffs':
                    b = v['beam_id']
                    delays_ns = np.array(v['data']['delays'])
                    amps = np.array(v['data']['amps'])
                    phases = np.exp(1j * 2 * np.pi * self.freqs[:, None] * delays_ns * 1e-9)   # freq x input
                    self.gains_cpu_new[:, b, :] = amps * phases * self.cal_gains[:, b, :]
                    self.gains_load_sample[b] = v.get('load_sample', -1)        # default: load immediately
                    self.update_pending = True      # only beam coefficients trigger a device update
            except KeyError:
                self.log.error("BEAMFORM >> Failed to parse command")
        self.update_stats(self.command_vals)
        if update_beam_cal_state:
            self.update_stats({'cal_gains%d' % b: self.stats['cal_gains%d' % b] for b in range(self.nbeam)})

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core(), 'ngpu': 1,
                                  'gpu0': self._bf.get_device()})
        igulp_size = self.ntime_gulp * self.nchan * self.ninput           # 4+4 bit
        ogulp_size = self.ntime_blocks * self.nchan * self.nbeam * 8      # complex64
        self.oring.resize(ogulp_size)
        # Streaming (in-repo rings, which keep a span's memory alive while it is referenced): the kernels of up to
        # STREAM_DEPTH gulps are in flight; a gulp's output span is committed, and its input released, when ITS kernels
        # have completed (beam_mark / beam_wait tickets) -- the reference, and the path taken on a circular bifrost ring,
        # waits for the stream after every gulp (beamform_block.py:450), which leaves the GPU idle between gulps and, beside
        # the X-engine's persistent kernel, costs one launch boundary of that kernel per gulp.
        streaming = (getattr(self.iring, 'span_memory_outlives_release', False) and getattr(self.oring, 'span_memory_outlives_release', False)
                     and hasattr(self._bf, 'beam_mark'))
        pending = collections.deque()           # (ticket, output span, input data kept alive)

        def retire(keep):
            while len(pending) > keep:
                ticket, osp, _ = pending.popleft()
                self._bf.beam_wait(ticket)
                osp.close()
        try:
            self._main_loop(igulp_size, ogulp_size, streaming, pending, retire)
        finally:
            # An exception must not release spans that kernels in flight still read or write (their memory would go back
            # to the ring, be handed out again or freed, under a running kernel): wait for the stream first.
            if pending:
                try:
                    self._bf.beam_sync()
                except Exception:
                    pass
                pending.clear()

    PUMP_GULPS = 8          # gulps per call of the native loop while no command is pending

    def _pump_sequence(self, pump, this_gulp_time, igulp_size, copy_pending, slab=False):
        """One input sequence through the native per-gulp loop (native rings, in-place gulps): this method keeps what is not
        steady state -- coefficient loads at their load times (one gulp per call while any is pending, as the per-gulp loop
        checks before every gulp), the upload of new weights (everything in flight retired first), statistics."""
        import ctypes
        stop_addr = ctypes.addressof(self._pump_stop)
        prev_time = time.time()
        try:
            while True:
                self._pump_stop.value = 0       # (lowered BEFORE the flag is read: a command that comes in from here on raises it again)
                if self.update_pending:
                    copy_pending = self._load_pending_gains(this_gulp_time) or copy_pending
                if copy_pending:
                    pump.drain()                # (kernels in flight may still read the device copy of the weights)
                    self.gains_gpu[...] = self.gains_cpu
                    self._gains_version += 1
                    copy_pending = False
                n, skipped, status = pump.run(self.gains_gpu.ptr, self._gains_version, 1 if self.update_pending else self.PUMP_GULPS, stop_addr,
                                              int(this_gulp_time) if slab else 0)
                this_gulp_time += ((skipped // igulp_size) + n) * self.ntime_gulp
                curr_time = time.time()
                if n:
                    self.update_stats({'curr_sample': this_gulp_time - self.ntime_gulp})
                    self.perf_proclog.update({'acquire_time': 0.0, 'reserve_time': 0.0, 'process_time': (curr_time - prev_time) / n,
                                              'gbps': 8 * igulp_size * n / max(curr_time - prev_time, 1e-9) / 1e9})
                prev_time = curr_time
                if status == 1:
                    return this_gulp_time
        except BaseException:
            pump.abort()
            raise

    def _main_loop(self, igulp_size, ogulp_size, streaming, pending, retire):
        """Per input sequence: header work (_begin_sequence), then one of two per-gulp loops -- the native one (_pump_sequence) between
        native rings, the Python one (_python_sequence) otherwise (bifrost / Python rings, XENG_PUMP=0)."""
        with self.oring.begin_writing() as oring:
            for iseq in self.iring.read(guarantee=self.guarantee):
                seq = self._begin_sequence(iseq)
                with oring.begin_sequence(time_tag=iseq.time_tag, header=json.dumps(seq['ohdr'])) as oseq:
                    pump = None
                    if (streaming and seq['read_parts'] is not None and hasattr(self._bf, 'beam_pump') and hasattr(iseq, '_rid')
                            and hasattr(oseq, '_seq_id')):
                        pump = self._bf.beam_pump(self.iring, iseq._rid, self.oring, oseq._seq_id, seq['igulp_size'], ogulp_size, 0,
                                                  row_bytes=self.nchan * self.ninput, depth=self.STREAM_DEPTH)
                    if pump is not None:
                        if seq['slab']:
                            pump.set_slabs(seq['slab_npkt'], seq['slab_stride'], seq['slab_ntime'], seq['ihdr']['chan0'], self.ntime_gulp)
                        self._pump_sequence(pump, seq['this_gulp_time'], seq['igulp_size'], True, seq['slab'])
                    else:
                        self._python_sequence(iseq, oseq, seq, ogulp_size, streaming, pending, retire)

    def _begin_sequence(self, iseq):
        """A new input sequence: frequencies (coefficients are rebuilt and re-uploaded on every sequence), the output header
        (beamform_block.py:403-409), and how the gulps of this sequence are laid out."""
        self.update_pending = True
        ihdr = json.loads(iseq.header.tostring())
        self.sequence_proclog.update(ihdr)
        nchan, nstand, npol = ihdr['nchan'], ihdr['nstand'], ihdr['npol']
        chan_bw = ihdr['bw_hz'] / nchan
        assert nchan == self.nchan
        assert self.ninput == nstand * npol
        self.freqs = ihdr['sfreq'] + chan_bw * np.arange(nchan)
        ohdr = ihdr.copy()
        ohdr['nstand'] = self.nbeam
        ohdr['nbit'] = 32
        ohdr['npol'] = 1            # single-polarisation beams
        ohdr['complex'] = True
        ohdr['nbeam'] = self.nbeam
        seq = {'ihdr': ihdr, 'ohdr': ohdr, 'this_gulp_time': ihdr['seq0'], 'igulp_size': self.ntime_gulp * self.nchan * self.ninput,
               # A gulp that lies in two spans of the input ring (ntime_gulp = 2 x the writer's gulp, as the reference runs it:
               # lwa352-pipeline.py:172,279-282) is taken as two windows and beamformed in ONE launch, without a gathered copy
               'read_parts': getattr(iseq, 'read_parts', None) if hasattr(self._bf, 'bfBeamformRunParts') else None,
               # A sequence of PACKET SLABS (Snap2Ingest(unpack=False), or a receiver that writes into a device ring): a gulp is one or
               # two slabs of SNAP2 packets, handed to the library as they are (xengBeamformRunSlabs: read in place when complete and
               # in order, scattered on the device otherwise)
               'slab': ihdr.get('layout') == 'snap2_slab'}
        if seq['slab']:
            slab_ntime, slab_npkt, slab_stride = ihdr['slab_ntime'], ihdr['npkt_per_gulp'], ihdr['pkt_stride']
            if self.ntime_gulp not in (slab_ntime, 2 * slab_ntime) or not hasattr(self._bf, 'bfBeamformRunSlabs'):
                raise RuntimeError("BEAMFORM: slabs of %d samples cannot make gulps of %d" % (slab_ntime, self.ntime_gulp))
            seq.update(slab_ntime=slab_ntime, slab_npkt=slab_npkt, slab_stride=slab_stride, slab_bytes=slab_npkt * slab_stride,
                       igulp_size=(self.ntime_gulp // slab_ntime) * slab_npkt * slab_stride)
            for k in ('layout', 'slab_ntime', 'npkt_per_gulp', 'pkt_stride'):
                ohdr.pop(k, None)
        return seq

    # ---- one gulp on each input layout: returns (status, what has to stay alive until the kernel has run)
    def _run_slabs(self, ispan, ospan, seq, this_gulp_time):
        parts = getattr(ispan, 'parts', None)
        slab_bytes = seq['slab_bytes']
        if parts is not None and len(parts) == 2:
            held = parts
            s0, s1 = parts
        elif seq['igulp_size'] == slab_bytes:
            held = s0 = ispan.data
            s1 = None
        else:       # two slabs side by side in one span
            held = ispan.data
            s0 = XArray.window(held.ptr, slab_bytes, held.space, held)
            s1 = XArray.window(held.ptr + slab_bytes, slab_bytes, held.space, held)
        rv = self._bf.bfBeamformRunSlabs(s0, seq['slab_npkt'], seq['slab_ntime'], s1, seq['slab_npkt'], seq['slab_stride'], this_gulp_time,
                                         seq['ihdr']['chan0'], ospan.data.as_BFarray(), self.gains_gpu.as_BFarray(), version=self._gains_version)
        return rv, held

    def _run_parts(self, parts, ospan):
        return self._bf.bfBeamformRunParts(parts[0], parts[1], ospan.data.as_BFarray(), self.gains_gpu.as_BFarray(), version=self._gains_version), parts

    def _run_plain(self, ispan, ospan):
        # (the reference takes typed views, ispan.data_view('i8') / ospan.data_view(np.float32), :441-444; the call only needs the
        # spans' addresses, and two fewer objects per gulp is time under the interpreter lock)
        held = ispan.data
        return self._bf.bfBeamformRun(held.as_BFarray(), ospan.data.as_BFarray(), self.gains_gpu.as_BFarray(), version=self._gains_version), held

    def _load_pending_gains(self, this_gulp_time):
        """beamform_block.py:416-429: coefficients whose load sample has come move from `new` to `cpu`; True when the device copy
        has to be rewritten."""
        copy_pending = False
        self.acquire_control_lock()
        for b in range(self.nbeam):
            if self.gains_load_sample[b] == 0:      # 0 = nothing pending for this beam
                continue
            if this_gulp_time >= self.gains_load_sample[b]:
                self.gains_cpu[:, b, :] = self.gains_cpu_new[:, b, :]
                self.gains_load_sample[b] = 0
                copy_pending = True
        if self.gains_load_sample.sum() == 0:
            self.update_pending = False
        self.stats['update_pending'] = self.update_pending
        self.stats['last_cmd_proc_time'] = time.time()
        self.release_control_lock()
        return copy_pending

    def _python_sequence(self, iseq, oseq, seq, ogulp_size, streaming, pending, retire):
        """The per-gulp loop in Python (beamform_block.py:411-461)."""
        igulp_size, this_gulp_time, read_parts = seq['igulp_size'], seq['this_gulp_time'], seq['read_parts']
        copy_pending = True
        prev_time = time.time()
        for ispan in (read_parts(igulp_size) if read_parts is not None else iseq.read(igulp_size)):
            self.update_stats({'curr_sample': this_gulp_time})
            if ispan.size < igulp_size:
                continue
            if getattr(ispan, 'skipped', 0):     # gulps overwritten before this reader got to them (ring.py)
                this_gulp_time += (ispan.skipped // igulp_size) * self.ntime_gulp
            if self.update_pending:
                copy_pending = self._load_pending_gains(this_gulp_time) or copy_pending
            if copy_pending:
                retire(0)           # (kernels in flight may still read the device copy of the weights)
                self.gains_gpu[...] = self.gains_cpu
                self._gains_version += 1
                copy_pending = False
            curr_time = time.time()
            acquire_time = curr_time - prev_time
            prev_time = curr_time
            ospan = oseq.reserve(ogulp_size)
            try:
                curr_time = time.time()
                reserve_time = curr_time - prev_time
                prev_time = curr_time
                parts = getattr(ispan, 'parts', None)
                if seq['slab']:
                    rv, held = self._run_slabs(ispan, ospan, seq, this_gulp_time)
                elif parts is not None and len(parts) == 2:
                    rv, held = self._run_parts(parts, ospan)
                else:
                    rv, held = self._run_plain(ispan, ospan)
                if rv != self._bf.BF_STATUS_SUCCESS:
                    raise RuntimeError("bfBeamformRun returned %d: %s" % (rv, self._bf.last_error()))
                if streaming:
                    pending.append((self._bf.beam_mark(), ospan, held))
                    ospan = None
                    retire(self.STREAM_DEPTH)
                else:
                    self._bf.beam_sync()          # BFSync() of beamform_block.py:450, this block's stream only
            finally:
                if ospan is not None:
                    ospan.close()
            this_gulp_time += self.ntime_gulp
            curr_time = time.time()
            process_time = curr_time - prev_time
            prev_time = curr_time
            self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time, 'process_time': process_time,
                                      'gbps': 8 * igulp_size / max(process_time, 1e-9) / 1e9})
        retire(0)                   # the sequence ends: every gulp in flight is committed first
