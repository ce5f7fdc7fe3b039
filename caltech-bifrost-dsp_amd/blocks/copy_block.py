"""Copy: ring -> ring, used for the host -> device ingest step (and device -> host).

Counterpart of pipeline/lwa352_pipeline/blocks/copy_block.py (constructor :92-94, main :119-160):
same signature, output ring sized to `4 x buffer_multiplier` gulps (or `buf_size_gbytes` rounded
down to whole gulps, :105-113), headers and time tags passed through, one `copy_array` per gulp
(:146).  With a 'cuda_host' (pinned) input ring and a 'cuda' output ring the copy is a
hipMemcpy H2D on libxeng's copy stream: the PCIe-bound leg of the ingest metric.
"""
import time

from ..ndarray import copy_array
from ..proclog import cpu_affinity
from .block_base import Block


class Copy(Block):
    def __init__(self, log, iring, oring, ntime_gulp=2500, buffer_multiplier=1,
                 guarantee=True, core=-1, nbyte_per_time=184 * 352 * 2, gpu=-1,
                 buf_size_gbytes=None, backend=None):
        super(Copy, self).__init__(log, iring, oring, guarantee, core, etcd_client=None)
        from .block_base import declare_streams
        declare_streams(iring, 'copy')
        declare_streams(oring, 'copy')
        cpu_affinity.set_core(self.core)
        self.ntime_gulp = ntime_gulp
        self.gpu = gpu
        self._bf = backend
        if self.gpu != -1 and self._bf is None and (iring.space != 'system' or oring.space != 'system'):
            from ..backend import default_backend
            self._bf = default_backend()
        if self.gpu != -1 and self._bf is not None:
            self._bf.set_device(self.gpu)
        self.buffer_multiplier = buffer_multiplier
        self.size_proclog.update({'nseq_per_gulp': self.ntime_gulp})
        self.igulp_size = self.ntime_gulp * nbyte_per_time
        if buf_size_gbytes is None:
            self.buf_size = 4 * self.igulp_size * self.buffer_multiplier
        else:
            unit = self.igulp_size * self.buffer_multiplier
            self.buf_size = int(1e9 * buf_size_gbytes) // unit * unit
        self.oring.resize(self.buffer_multiplier * self.igulp_size, total_span=self.buf_size)

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1 and self._bf is not None:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})
        with self.oring.begin_writing() as oring:
            for iseq in self.iring.read(guarantee=self.guarantee):
                prev_time = time.time()
                bytes_copied = 0
                acquire_time = reserve_time = process_time = 0
                with oring.begin_sequence(time_tag=iseq.time_tag, header=iseq.header, nringlet=iseq.nringlet) as oseq:
                    for ispan in iseq.read(self.igulp_size):
                        if ispan.size < self.igulp_size:
                            continue
                        curr_time = time.time()
                        acquire_time += curr_time - prev_time
                        prev_time = curr_time
                        with oseq.reserve(ispan.size) as ospan:
                            curr_time = time.time()
                            reserve_time += curr_time - prev_time
                            prev_time = curr_time
                            copy_array(ospan.data, ispan.data)       # synchronous on return
                        curr_time = time.time()
                        process_time += curr_time - prev_time
                        prev_time = curr_time
                        bytes_copied += ispan.size
                        if bytes_copied > 10e9:
                            self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                                      'process_time': process_time,
                                                      'gbps': 8 * bytes_copied / max(process_time, 1e-9) / 1e9})
                            bytes_copied = 0
                            acquire_time = reserve_time = process_time = 0
                if process_time > 0:
                    self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                              'process_time': process_time,
                                              'gbps': 8 * bytes_copied / max(process_time, 1e-9) / 1e9})
