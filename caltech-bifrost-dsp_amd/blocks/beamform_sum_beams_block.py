"""BeamformSumBeams: dual-polarisation power beams from pairs of voltage beams.

Drop-in counterpart of pipeline/lwa352_pipeline/blocks/beamform_sum_beams_block.py
(constructor :161-163, main :189-260).  Beams 2b / 2b+1 are treated as X / Y; every `ntime_sum`
samples it emits [XX, YY, Re(XY*), Im(XY*)] per channel, f32 [nbeam/2, ntime/ntime_sum, nchan, 4]
(:220-222; math beamformer_sum_test.py:64-77).  The beamformer context is the process-global
one the Beamform block created (:186-187): this block does not initialise it.
"""
import collections
import json
import time

import numpy as np

from ..backend import default_backend
from ..ndarray import XArray
from ..proclog import cpu_affinity
from .block_base import Block, declare_streams


class BeamformSumBeams(Block):
    STREAM_DEPTH = 8        # gulps whose kernel may be in flight behind the one being enqueued (streaming mode)

    def __init__(self, log, iring, oring, nchan=256,
                 ntime_gulp=2500, ntime_sum=24, guarantee=True, core=-1, gpu=-1,
                 etcd_client=None, backend=None):
        super(BeamformSumBeams, self).__init__(log, iring, oring, guarantee, core, etcd_client=etcd_client)
        self._bf = backend if backend is not None else default_backend()
        self.ntime_gulp = ntime_gulp
        self.gpu = gpu
        self.ntime_sum = ntime_sum
        declare_streams(iring, 'beam')
        declare_streams(oring, 'beam', 'copy')  # (the kernel writes the span itself, or a copy does on the non-streaming path)
        assert ntime_gulp % ntime_sum == 0
        self.ntime_blocks = ntime_gulp // ntime_sum
        self.nchan = nchan
        if self.gpu != -1:
            self._bf.set_device(self.gpu)

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core(), 'ngpu': 1,
                                  'gpu0': self._bf.get_device()})
        # Streaming (in-repo rings): up to STREAM_DEPTH gulps in flight; a span is committed when its own kernel (and copy) has
        # completed.  On a bifrost ring: the reference's Integrate -> wait -> copy (:243-250).
        # Where the sums go: the pipeline's output ring is pinned host memory (lwa352-pipeline.py:155).  Round 3 let the kernel
        # write the span itself; that saves a copy but makes the kernel wait for 1 MB of PCIe stores -- 31 us instead of 12, on
        # the stream the beamformer kernels queue on, 2.5 times per integration (profiles/r04/blocks_kernel_time.txt).  So
        # (round 4) the kernel writes a device buffer, and once ITS ticket is done the buffer goes to the pinned span on the
        # copy stream; the span is committed when that copy has completed.  A device output ring still takes the sums directly.
        streaming = (getattr(self.iring, 'span_memory_outlives_release', False) and getattr(self.oring, 'span_memory_outlives_release', False)
                     and hasattr(self._bf, 'beam_mark')
                     and (self.oring.space in ('cuda', 'cuda_host') or self.oring.space == self._bf.space_in))      # (the kernel writes the span)
        staged = streaming and self.oring.space == 'cuda_host' and hasattr(self._bf, 'copy_async')
        pending = collections.deque()           # (ticket, output span, input kept alive, device buffer or None)
        copying = collections.deque()           # (stamp of the copy, output span, device buffer)
        self._stages_free = []

        def finish_copies(keep):
            while copying and (len(copying) > keep or self._bf.copy_done(copying[0][0])):
                stamp, osp, stage = copying.popleft()
                self._bf.copy_wait(stamp)       # (returns at once when it is done)
                osp.close()
                self._stages_free.append(stage)

        def retire(keep):
            while len(pending) > keep:
                ticket, osp, _, stage = pending.popleft()
                self._bf.beam_wait(ticket)
                if stage is None:
                    osp.close()
                else:
                    copying.append((self._bf.copy_async(osp.data, stage), osp, stage))
            finish_copies(2 if keep else 0)
        self._staged = staged
        try:
            self._main_loop(streaming, pending, retire)
        finally:
            # (as in Beamform: spans of kernels in flight are not released by an exception before the stream is idle)
            if pending or copying:
                try:
                    self._bf.beam_sync()
                    for stamp, _, _ in copying:
                        self._bf.copy_wait(stamp)
                except Exception:
                    pass
                pending.clear()
                copying.clear()

    def _main_loop(self, streaming, pending, retire):
        with self.oring.begin_writing() as oring:
            for iseq in self.iring.read(guarantee=self.guarantee):
                ihdr = json.loads(iseq.header.tostring())
                self.sequence_proclog.update(ihdr)
                nchan, nbeam = ihdr['nchan'], ihdr['nbeam']
                assert nchan == self.nchan
                ohdr = ihdr.copy()
                ohdr['nbeam'] = nbeam // 2      # single-pol beams -> dual-pol
                ohdr['nbit'] = 32
                ohdr['complex'] = True
                ohdr['acc_len'] = self.ntime_sum
                ohdr['npol'] = 2
                self.bf_output = XArray(shape=(ohdr['nbeam'], self.ntime_blocks, nchan, 4), dtype=np.float32,
                                        space=self._bf.space_in)
                igulp_size = self.ntime_gulp * nchan * nbeam * 2 * 32 // 8
                ogulp_size = self.ntime_blocks * nchan * ohdr['nbeam'] * 4 * 4
                # output gulps are small: size the ring in units of input gulps (:225-228)
                self.oring.resize(ogulp_size * self.ntime_sum * 4)
                prev_time = time.time()
                with oring.begin_sequence(time_tag=iseq.time_tag, header=json.dumps(ohdr)) as oseq:
                    pump = None
                    if streaming and hasattr(self._bf, 'beam_pump') and hasattr(iseq, '_rid') and hasattr(oseq, '_seq_id'):
                        pump = self._bf.beam_pump(self.iring, iseq._rid, self.oring, oseq._seq_id, igulp_size, ogulp_size, 1,
                                                  ntime_sum=self.ntime_sum, depth=self.STREAM_DEPTH, staged=self._staged)
                    if pump is not None:
                        # the native per-gulp loop (csrc/pyext/xfast.cpp BeamPump): this block has no commands; back here every few
                        # gulps for the statistics
                        try:
                            while True:
                                n, _, status = pump.run(0, 0, 8, 0)
                                curr_time = time.time()
                                if n:
                                    self.perf_proclog.update({'acquire_time': 0.0, 'reserve_time': 0.0, 'process_time': (curr_time - prev_time) / n,
                                                              'gbps': 8 * igulp_size * n / max(curr_time - prev_time, 1e-9) / 1e9})
                                prev_time = curr_time
                                if status == 1:
                                    break
                        except BaseException:
                            pump.abort()
                            raise
                        continue
                    for ispan in iseq.read(igulp_size):
                        if ispan.size < igulp_size:
                            continue
                        curr_time = time.time()
                        acquire_time = curr_time - prev_time
                        prev_time = curr_time
                        ospan = oseq.reserve(ogulp_size)
                        try:
                            curr_time = time.time()
                            reserve_time = curr_time - prev_time
                            prev_time = curr_time
                            # (streaming: the kernel writes into the span itself; the call only needs addresses, so no typed views)
                            stage = None
                            if self._staged:
                                stage = self._stages_free.pop() if self._stages_free else None
                                if stage is None or stage.nbytes != ogulp_size:
                                    stage = XArray(shape=(ogulp_size,), dtype=np.uint8, space=self._bf.space_in)
                            target = stage if stage is not None else (ospan.data if streaming else self.bf_output)
                            rv = self._bf.bfBeamformIntegrate(ispan.data.as_BFarray(), target.as_BFarray(), self.ntime_sum)
                            if rv != self._bf.BF_STATUS_SUCCESS:
                                raise RuntimeError("bfBeamformIntegrate returned %d: %s" % (rv, self._bf.last_error()))
                            if streaming:
                                pending.append((self._bf.beam_mark(), ospan, ispan.data, stage))
                                ospan = None
                                retire(self.STREAM_DEPTH)
                            else:
                                self._bf.beam_sync()
                                ospan.data_view(np.float32).reshape(self.bf_output.shape)[...] = self.bf_output       # (synchronous copy)
                        finally:
                            if ospan is not None:
                                ospan.close()
                        curr_time = time.time()
                        process_time = curr_time - prev_time
                        prev_time = curr_time
                        self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                                  'process_time': process_time,
                                                  'gbps': 8 * igulp_size / max(process_time, 1e-9) / 1e9})
                    retire(0)
