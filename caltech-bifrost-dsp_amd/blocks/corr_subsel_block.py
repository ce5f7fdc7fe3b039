"""CorrSubsel: sub-select and channel-sum visibilities from the xGPU-order buffer (fast-visibility path).

Counterpart of pipeline/lwa352_pipeline/blocks/corr_subsel_block.py (constructor :185-232,
update_subsel :234-250, main :252-332): picks `nvis_out` = 4704 baselines (48 stands, dual pol)
named as [[stand0, pol0], [stand1, pol1]], sums `nchan_sum` channels, applies the conjugation flag
and writes interleaved ci32 [nchan/nchan_sum, nvis].  The gather/sum/conjugate kernel is libxeng's
`subselect_kernel` behind `bfXgpuSubSelect` (:298).  Selection lists arrive through the `baselines`
command key and take effect at the next sequence / after the current block (:316-329).
"""
import json
import time

import numpy as np

from ..backend import default_backend
from ..ndarray import XArray, copy_array
from ..proclog import cpu_affinity
from .block_base import Block


class CorrSubsel(Block):
    nvis_out = 48 * 49 * 4 // 2     # 48-stand, dual-pol

    def __init__(self, log, iring, oring, guarantee=True, core=-1, etcd_client=None,
                 nchan=192, npol=2, nstand=352, nchan_sum=4, gpu=-1,
                 antpol_to_bl=None, bl_is_conj=None, backend=None, nvis_out=None):
        super(CorrSubsel, self).__init__(log, iring, oring, guarantee, core, etcd_client=etcd_client)
        from .block_base import declare_streams
        declare_streams(iring, 'consumer')      # (the gather kernel runs on the span consumers' stream)
        declare_streams(oring, 'consumer', 'copy')
        self._bf = backend if backend is not None else default_backend()
        if nvis_out is not None:
            self.nvis_out = nvis_out            # (the reference fixes 4704; smaller values ease testing)
        self.nchan_in = nchan
        self.nchan_out = nchan // nchan_sum
        self.nchan_sum = nchan_sum
        self.npol, self.nstand = npol, nstand
        self.gpu = gpu
        self.matlen = self.nchan_in * (nstand // 2 + 1) * (nstand // 4) * npol * npol * 4
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.igulp_size = self.matlen * 8
        dev, host = self._bf.space_in, ('cuda_host' if self._bf.space_in == 'cuda' else 'system')
        self._subsel = XArray(shape=[self.nvis_out], dtype='i32', space=dev)
        self._subsel_next = XArray(shape=[self.nvis_out], dtype='i32', space=host)
        self._conj = XArray(shape=[self.nvis_out], dtype='i32', space=dev)
        self._conj_next = XArray(shape=[self.nvis_out], dtype='i32', space=host)
        self.obuf_gpu = XArray(shape=[self.nchan_out, self.nvis_out, 2], dtype='i32', space=dev)
        self.ogulp_size = self.nchan_out * self.nvis_out * 8
        self.update_stats()
        self._antpol_to_bl = np.asarray(antpol_to_bl) if antpol_to_bl is not None else np.zeros([nstand, nstand, npol, npol], np.int32)
        self._bl_is_conj = np.asarray(bl_is_conj) if bl_is_conj is not None else np.zeros([nstand, nstand, npol, npol], np.int32)
        # default selection: pol-0 autos (:230)
        subsel = [[[i % nstand, 0], [i % nstand, 0]] for i in range(self.nvis_out)]
        self.define_command_key('baselines', type=list, initial_val=subsel,
                                condition=lambda x: len(x) == self.nvis_out)
        self.update_subsel(subsel)

    def update_subsel(self, baselines):
        """Translate [[s0,p0],[s1,p1]] pairs to buffer indices / conjugation flags (indexed [s0,s1,p0,p1], :244-250)."""
        cpu_affinity.set_core(self.core)
        sel, cj = self._subsel_next.numpy(), self._conj_next.numpy()
        for v in range(self.nvis_out):
            (s0, p0), (s1, p1) = baselines[v]
            sel[v] = self._antpol_to_bl[s0, s1, p0, p1]
            cj[v] = self._bl_is_conj[s0, s1, p0, p1]

    def _load_selection(self):
        self.update_command_vals()
        self.update_subsel(self.command_vals['baselines'])
        copy_array(self._subsel, self._subsel_next)
        copy_array(self._conj, self._conj_next)

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})
        self.oring.resize(self.ogulp_size)
        time_tag = 1
        with self.oring.begin_writing() as oring:
            prev_time = time.time()
            for iseq in self.iring.read(guarantee=self.guarantee):
                ihdr = json.loads(iseq.header.tostring())
                this_gulp_time = ihdr['seq0']
                acc_len = ihdr['acc_len']
                ohdr = ihdr.copy()
                ohdr['nchan'] = ihdr['nchan'] // self.nchan_sum
                ohdr['nvis'] = self.nvis_out
                chan_width = ihdr['bw_hz'] / ihdr['nchan']
                ohdr['sfreq'] = (ihdr['sfreq'] + ((self.nchan_sum - 1) * chan_width)) / self.nchan_sum
                self._load_selection()                      # always take the newest list at a sequence start
                ohdr['baselines'] = self.command_vals['baselines']
                ohdr['nchan_sum'] = self.nchan_sum
                oseq = oring.begin_sequence(time_tag=time_tag, header=json.dumps(ohdr), nringlet=iseq.nringlet)
                time_tag += 1
                for ispan in iseq.read(self.igulp_size):
                    if ispan.size < self.igulp_size:
                        continue
                    curr_time = time.time()
                    acquire_time = curr_time - prev_time
                    prev_time = curr_time
                    idata = ispan.data_view('i32')
                    with oseq.reserve(self.ogulp_size) as ospan:
                        curr_time = time.time()
                        reserve_time = curr_time - prev_time
                        prev_time = curr_time
                        rv = self._bf.bfXgpuSubSelect(idata.as_BFarray(), self.obuf_gpu.as_BFarray(),
                                                      self._subsel.as_BFarray(), self._conj.as_BFarray(),
                                                      self.nchan_sum, 0)
                        if rv != self._bf.BF_STATUS_SUCCESS:
                            self.log.error("xgpuSubSelect returned %d" % rv)
                            raise RuntimeError("xgpuSubSelect returned %d: %s" % (rv, self._bf.last_error()))
                        odata = ospan.data_view('i32').reshape(self.obuf_gpu.shape)
                        copy_array(odata, self.obuf_gpu)      # (SubSelect and the copy are complete on return)
                        curr_time = time.time()
                        process_time = curr_time - prev_time
                        prev_time = curr_time
                    self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                              'process_time': process_time, 'this_sample': this_gulp_time})
                    this_gulp_time += acc_len
                    if self.update_pending:                 # new selection: new sequence with an updated header
                        oseq.end()
                        self._load_selection()
                        ohdr['baselines'] = self.command_vals['baselines']
                        ohdr['seq0'] = this_gulp_time
                        oseq = oring.begin_sequence(time_tag=time_tag, header=json.dumps(ohdr), nringlet=iseq.nringlet)
                        time_tag += 1
                oseq.end()
