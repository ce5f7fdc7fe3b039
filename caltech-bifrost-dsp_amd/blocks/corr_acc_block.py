"""CorrAcc: long accumulation of xGPU-order visibilities on the GPU.

Drop-in counterpart of pipeline/lwa352_pipeline/blocks/corr_acc_block.py (class CorrAcc,
constructor :170-171, main :193-336): accumulates upstream short integrations
(`BFMap("a = b")` on the first, `"a += b"` afterwards, :304-306) into an internal int32 buffer
and publishes it on the last (:310-319, `copy_array` + `stream_synchronize`; the output ring is
`cuda_host` in the pipeline, lwa352-pipeline.py:154).  The two map expressions are libxeng's
vectorised int32 kernels (csrc/corracc.hip).

Input header needs `seq0`, `acc_len` (:214-215); output adds `upstream_acc_len` (:216) and
rewrites `acc_len` / `seq0`.  `start_time == -1` starts on the current block (:244-245);
recovery after a new upstream sequence skips 2 integrations (:227).
"""
import json
import time

from ..backend import default_backend
from ..ndarray import XArray, copy_array
from ..proclog import cpu_affinity
from ..ring import WriteSpan
from .block_base import Block
from .integration import IntegrationGate


class CorrAcc(Block):
    def __init__(self, log, iring, oring,
                 guarantee=True, core=-1, nchan=192, npol=2, nstand=352, acc_len=24000, gpu=-1, etcd_client=None,
                 autostartat=0, backend=None):
        super(CorrAcc, self).__init__(log, iring, oring, guarantee, core, etcd_client=etcd_client)
        self._bf = backend if backend is not None else default_backend()
        self.nchan, self.npol, self.nstand = nchan, npol, nstand
        self.matlen = nchan * (nstand // 2 + 1) * (nstand // 4) * npol * npol * 4
        self.gpu = gpu
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.igulp_size = self.matlen * 8     # complex64
        self.ogulp_size = self.igulp_size
        # integration buffer lives where the input ring lives (device memory in the pipeline)
        self.accdata = XArray(shape=(self.igulp_size // 4,), dtype='i32', space=self._bf.space_in)
        self.define_command_key('start_time', type=int, initial_val=autostartat)
        self.define_command_key('acc_len', type=int, initial_val=acc_len)

    def _check_compat(self, gate, upstream_acc_len, upstream_start_time):
        if upstream_acc_len and gate.acc_len % upstream_acc_len != 0:
            self.log.error("CORRACC >> Requested acc_len %d incompatible with upstream integration %d" % (gate.acc_len, upstream_acc_len))
        if upstream_acc_len and gate.acc_len != 0 and ((gate.start_time - upstream_start_time) % upstream_acc_len != 0):
            self.log.error("CORRACC >> Requested start_time %d incompatible with upstream integration %d" % (gate.start_time, upstream_acc_len))

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})

        self.oring.resize(self.ogulp_size)
        oseq = ospan = None
        gate = IntegrationGate(recovery_skip=2, round_start_to_acc_len=False)
        process_time = 0
        acquire_time = reserve_time = 0
        time_tag = 1
        self.update_stats({'state': 'starting'})
        with self.oring.begin_writing() as oring:
            prev_time = time.time()
            self.update_pending = True
            for iseq in self.iring.read(guarantee=self.guarantee):
                ihdr = json.loads(iseq.header.tostring())
                ohdr = ihdr.copy()
                now = ihdr['seq0']
                step = ihdr['acc_len']                      # upstream integration length
                ohdr['upstream_acc_len'] = step
                upstream_start_time = now
                self.sequence_proclog.update(ohdr)
                if gate.recover(now):
                    self.log.info("CORRACC >> Recovering start time set to %d. Accumulating %d samples" % (gate.start_time, gate.acc_len))
                    self._check_compat(gate, step, upstream_start_time)
                    ohdr['acc_len'] = gate.acc_len
                    ohdr['seq0'] = gate.start_time
                for ispan in iseq.read(self.igulp_size):
                    if ispan.size < self.igulp_size:
                        continue
                    if self.update_pending:
                        self.update_command_vals()
                        gate.configure(now, self.command_vals['acc_len'], self.command_vals['start_time'])
                        self.log.info("CORRACC >> New start time at %d. Accumulation: %d samples" % (gate.start_time, gate.acc_len))
                        self._check_compat(gate, step, upstream_start_time)
                        ohdr['acc_len'] = gate.acc_len
                        ohdr['seq0'] = gate.start_time
                    self.stats.update({'curr_sample': now})
                    self.update_stats()
                    if gate.acc_len == 0:                   # stop command (:257-263)
                        self.update_stats({'state': 'stopped'})
                        if oseq:
                            oseq.end()
                        oseq = None
                        gate.running = False
                        now += step
                        continue
                    if gate.try_start(now, step):
                        if oseq:
                            oseq.end()
                        self.sequence_proclog.update(ohdr)
                        oseq = oring.begin_sequence(time_tag=time_tag, header=json.dumps(ohdr), nringlet=iseq.nringlet)
                        time_tag += 1
                        self.log.info("CORRACC >> Start time %d reached. Accumulating to %d (upstream accumulation: %d)" % (gate.start_time, gate.last, step))
                    if not gate.running:
                        self.update_stats({'state': 'waiting_start_missed' if now > gate.start_time else 'waiting'})
                        now += step
                        continue
                    self.update_stats({'state': 'running'})
                    curr_time = time.time()
                    acquire_time = curr_time - prev_time
                    prev_time = curr_time
                    idata = ispan.data_view('i32')
                    if now == gate.first:
                        curr_time = time.time()
                        reserve_time = curr_time - prev_time
                        prev_time = curr_time
                        rv = self._bf.map_assign_i32(self.accdata, idata)      # "a = b"
                    else:
                        rv = self._bf.map_add_i32(self.accdata, idata)         # "a += b"
                    if rv != self._bf.BF_STATUS_SUCCESS:
                        raise RuntimeError("CorrAcc map returned %d: %s" % (rv, self._bf.last_error()))
                    # the input span is recycled when the loop advances: the map must have read it (this block's
                    # stream only: the X-engine and beamformer streams keep running)
                    self._bf.map_sync()
                    curr_time = time.time()
                    process_time += curr_time - prev_time
                    prev_time = curr_time
                    if now == gate.last:
                        ospan = WriteSpan(oseq.ring, self.ogulp_size, nonblocking=False)
                        odata = ospan.data_view('i32').reshape(self.accdata.shape)
                        copy_array(odata, self.accdata)       # (synchronous: complete before the span is committed)
                        ospan.close()
                        ospan = None
                        curr_time = time.time()
                        process_time += curr_time - prev_time
                        prev_time = curr_time
                        self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                                  'process_time': process_time})
                        self.update_stats({'last_end_sample': now})
                        process_time = 0
                        gate.advance(step)
                    now += step
            if ospan:
                ospan.close()
            if oseq:
                oseq.end()
