"""Snap2Ingest: received F-engine packets -> gulps on the device.

Stands where `Capture` + `Copy` stand in the reference (capture_block.py:165-305, copy_block.py:92-160).
There, bifrost's C++ UDP capture (an absent submodule) scatters packet payloads into a pinned host ring on
the CPU and `Copy` moves the assembled gulps across PCIe.  Here the receiver only has to drop packets into
slabs -- one slab per window of `ntime_gulp` sequence numbers, packets in arrival order, `pkt_stride` bytes
apart, unused slots left zero -- and this block moves each slab across PCIe as it is and scatters it on the
device (`xengSnap2Unpack`): the host never touches a payload byte, and the output is the `gpu-input` ring
the X-engine and beamformer read.  With `unpack=False` the slabs are only moved: the output ring holds packet slabs, which
Corr and Beamform read in place (see __init__).  (Sockets / verbs receive are out of scope: the block reads slabs from a
ring; tests and the emulator of test_tx_vectors.py fill it.)

Packet format: test_tx_vectors.py:38-48,103-108 / test_tx_mt.c:39-49, header `>QLHHHHLLL`.

Input sequence header (written by the receiver): `seq0`, `sync_time`, `chan0`, and optionally `pkt_stride`,
`npkt_per_gulp` (defaults: constructor).  Span k of a sequence holds the window
[seq0 + k*ntime_gulp, seq0 + (k+1)*ntime_gulp).  Output sequence header: the dict capture_block.py:264-282
builds (time_tag, sync_time, seq0, chan0, nchan, system_nchan, fs_hz, sfreq, bw_hz, nstand, pipeline_id, npol,
complex, nbit).  Samples of packets that never arrived are zero; `stats['missing_frac']` reports them.
"""
import json
import time

from ..backend import default_backend
from ..ndarray import XArray, copy_array
from ..proclog import cpu_affinity
from .block_base import Block


class Snap2Ingest(Block):
    def __init__(self, log, iring, oring, ntime_gulp=480, nchan=96, nstand=352, npol=2,
                 nchan_per_pkt=96, nstand_per_pkt=32, npkt_per_gulp=None,
                 fs_hz=196000000, chan_bw_hz=23925.78125, system_nchan=184 * 16,
                 guarantee=True, core=-1, gpu=-1, buffer_multiplier=4, backend=None, unpack=True):
        super(Snap2Ingest, self).__init__(log, iring, oring, guarantee, core, etcd_client=None)
        from .block_base import declare_streams
        declare_streams(iring, 'copy', 'xgpu')  # (the scatter runs on the copy stream, or enqueue-only on the X-engine's staging stream)
        declare_streams(oring, 'copy', 'xgpu')
        self._bf = backend if backend is not None else default_backend()
        self.ntime_gulp, self.nchan, self.nstand, self.npol = ntime_gulp, nchan, nstand, npol
        self.fs_hz, self.chan_bw_hz, self.system_nchan = fs_hz, chan_bw_hz, system_nchan
        self.gpu = gpu
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.pkt_stride = 32 + nchan_per_pkt * nstand_per_pkt * npol
        # packets of one complete window: every (sequence number, channel block, stand block)
        self.npkt_nominal = ntime_gulp * (nchan // nchan_per_pkt) * (nstand // nstand_per_pkt)
        self.npkt_per_gulp = npkt_per_gulp if npkt_per_gulp is not None else self.npkt_nominal
        self.igulp_size = self.npkt_per_gulp * self.pkt_stride
        self.ogulp_size = ntime_gulp * nchan * nstand * npol
        # unpack=False (round 4): the slabs go to the device as they are and STAY slabs -- the output ring holds packet slabs, its
        # sequence header says so ('layout': 'snap2_slab', 'slab_ntime', 'npkt_per_gulp', 'pkt_stride'), and Corr / Beamform hand
        # them to the library's slab calls, which read complete, in-order slabs in place and scatter only the others
        # (xengXgpuKernelAsyncSlab, xengBeamformRunSlabs): no scatter pass and no second copy of the voltages on the device.
        self.unpack = unpack
        if not unpack:
            self.ogulp_size = self.igulp_size
        self.oring.resize(self.ogulp_size, total_span=buffer_multiplier * self.ogulp_size)
        self._slab_dev = None
        self.time_tag = 0
        self.size_proclog.update({'nseq_per_gulp': ntime_gulp})

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})
        with self.oring.begin_writing() as oring:
            for iseq in self.iring.read(guarantee=self.guarantee):
                ihdr = json.loads(iseq.header.tostring())
                seq0, chan0 = ihdr['seq0'], ihdr['chan0']
                stride = ihdr.get('pkt_stride', self.pkt_stride)
                npkt = ihdr.get('npkt_per_gulp', self.npkt_per_gulp)
                igulp = npkt * stride
                self.time_tag += 1
                ohdr = {'time_tag': self.time_tag, 'sync_time': ihdr.get('sync_time', 0), 'seq0': seq0,
                        'chan0': chan0, 'nchan': self.nchan, 'system_nchan': self.system_nchan, 'fs_hz': self.fs_hz,
                        'sfreq': chan0 * self.chan_bw_hz, 'bw_hz': self.nchan * self.chan_bw_hz, 'nstand': self.nstand,
                        'pipeline_id': self.pipeline_id, 'npol': self.npol, 'complex': True, 'nbit': 4}
                if not self.unpack:
                    ohdr.update({'layout': 'snap2_slab', 'slab_ntime': self.ntime_gulp, 'npkt_per_gulp': npkt, 'pkt_stride': stride})
                prev_time = time.time()
                placed_tot = dropped_tot = nwin = 0
                with oring.begin_sequence(time_tag=self.time_tag, header=json.dumps(ohdr), nringlet=iseq.nringlet) as oseq:
                    for ispan in iseq.read(igulp):
                        if ispan.size < igulp:
                            continue
                        curr_time = time.time()
                        acquire_time = curr_time - prev_time
                        prev_time = curr_time
                        slab = ispan.data
                        if not self.unpack:
                            with oseq.reserve(igulp) as ospan:
                                curr_time = time.time()
                                reserve_time = curr_time - prev_time
                                prev_time = curr_time
                                copy_array(ospan.data, slab)             # one H2D (or D2D) of the raw packets; nothing else
                            curr_time = time.time()
                            process_time = curr_time - prev_time
                            prev_time = curr_time
                            nwin += 1
                            self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time, 'process_time': process_time,
                                                      'gbps': 8 * igulp / max(process_time, 1e-9) / 1e9})
                            self.update_stats({'curr_sample': seq0 + nwin * self.ntime_gulp})
                            continue
                        if slab.space != self._bf.space_in:          # pinned host slab: one H2D of the raw packets
                            if self._slab_dev is None or self._slab_dev.nbytes != igulp:
                                self._slab_dev = XArray(shape=[igulp], dtype='u8', space=self._bf.space_in)
                            copy_array(self._slab_dev, slab)
                            slab = self._slab_dev
                        with oseq.reserve(self.ogulp_size) as ospan:
                            curr_time = time.time()
                            reserve_time = curr_time - prev_time
                            prev_time = curr_time
                            rv, placed, dropped = self._bf.snap2_unpack(slab, npkt, stride, ospan.data,
                                                                        seq0 + nwin * self.ntime_gulp, self.ntime_gulp,
                                                                        chan0, self.nchan, self.nstand * self.npol, True)
                            if rv != self._bf.BF_STATUS_SUCCESS:
                                self.log.error("snap2 unpack returned %d" % rv)
                                raise RuntimeError("snap2 unpack returned %d: %s" % (rv, self._bf.last_error()))
                        curr_time = time.time()
                        process_time = curr_time - prev_time
                        prev_time = curr_time
                        nwin += 1
                        placed_tot += placed
                        dropped_tot += dropped
                        self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                                  'process_time': process_time,
                                                  'gbps': 8 * self.ogulp_size / max(process_time, 1e-9) / 1e9})
                        self.update_stats({'curr_sample': seq0 + nwin * self.ntime_gulp, 'packets_placed': placed_tot,
                                           'packets_dropped': dropped_tot,
                                           'missing_frac': 1.0 - placed_tot / float(max(nwin * self.npkt_nominal, 1))})
