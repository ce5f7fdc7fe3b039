"""CorrOutputPart: fast (sub-selected) visibilities -> UDP packets.

Counterpart of pipeline/lwa352_pipeline/blocks/corr_output_part_block.py (constructor :318-344,
send_packets_py :346-364, main :403-470).  Pure host code behind `CorrSubsel`: the input span is
int32[nchan][nvis][2] (corr_subsel_block.py), transposed to [nvis][nchan][2] and sent `nvis_per_packet`
visibilities at a time, everything big-endian (the reference stages the data in a `'>i'` array, :429, :459):

    header  `>QQ2d4I` = sync_time, spectra_id, bw_hz, sfreq_hz, acc_len, nvis_per_packet, nchan, chan0   (:351-359)
            + int32 baselines[nvis_per_packet][2][2]  ([[stand0, pol0], [stand1, pol1]] per visibility)  (:360)
    payload int32 data[nvis_per_packet][nchan][2]                                                     (:363)

(docs/source/outputs.rst:120-141.)  Only this python format is built: with `use_cor_fmt=True` the reference
hands the data to bifrost's `cor` packet writer, an absent submodule.  `dest_ip == "0.0.0.0"` skips sending
(:447); a `sink(packet_bytes)` callable receives every packet in tests.
"""
import json
import socket
import struct
import time

import numpy as np

from ..proclog import cpu_affinity
from .block_base import Block


class CorrOutputPart(Block):
    def __init__(self, log, iring, use_cor_fmt=False,
                 guarantee=True, core=-1, etcd_client=None, dest_port=10001, nvis_per_packet=16,
                 nchan_sum=1, pipeline_idx=1, npipeline=1, sink=None):
        super(CorrOutputPart, self).__init__(log, iring, None, guarantee, core, etcd_client=etcd_client)
        if use_cor_fmt:
            raise NotImplementedError("CorrOutputPart: the COR format is produced by bifrost's packet writer in the "
                                      "reference (not available here); use use_cor_fmt=False")
        self.nvis_per_packet = nvis_per_packet
        self.nchan_sum = nchan_sum
        self.pipeline_idx = pipeline_idx
        self.npipeline = npipeline
        wrapped_idx = ((self.pipeline_idx - 1) % self.npipeline) + 1
        self.tuning = ((self.nchan_sum << 16) | (self.npipeline << 8) | wrapped_idx) & 0x00FFFFFF
        self.use_cor_fmt = use_cor_fmt
        self.sink = sink
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.sock.settimeout(0.01)
        self.define_command_key('dest_ip', type=str, initial_val='0.0.0.0')
        self.define_command_key('dest_port', type=int, initial_val=dest_port)
        self.update_command_vals()

    def send_packets_py(self, dout, baselines, sync_time, this_gulp_time, bw_hz, sfreq, upstream_acc_len, nchan, chan0):
        cpu_affinity.set_core(self.core)
        baselines_flat = baselines.flatten()
        n = self.nvis_per_packet
        for vn in range(len(baselines) // n):
            header = struct.pack(">QQ2d4I", sync_time, this_gulp_time, bw_hz, sfreq, upstream_acc_len, n, nchan, chan0) \
                + baselines_flat[vn * 4 * n:(vn + 1) * 4 * n].tobytes()
            pkt = header + dout[vn * n:(vn + 1) * n].tobytes()
            if self.sink is not None:
                self.sink(pkt)
            if self.command_vals['dest_ip'] != "0.0.0.0":
                self.sock.sendto(pkt, (self.command_vals['dest_ip'], self.command_vals['dest_port']))

    def main(self):
        cpu_affinity.set_core(self.core)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})
        prev_time = time.time()
        for iseq in self.iring.read(guarantee=self.guarantee):
            self.update_pending = True                        # reprocess commands on each new sequence (:410)
            ihdr = json.loads(iseq.header.tostring())
            this_gulp_time = ihdr['seq0']
            upstream_acc_len = ihdr['acc_len']
            baselines = np.array(ihdr['baselines'], dtype='>i')
            nchan = ihdr['nchan']
            chan0 = ihdr['chan0']
            bw_hz = ihdr['bw_hz']
            nvis = ihdr['nvis']
            sfreq = ihdr['sfreq']
            igulp_size = nvis * nchan * 8
            dout = np.zeros(shape=[nvis, nchan, 2], dtype='>i')
            for ispan in iseq.read(igulp_size):
                if ispan.size < igulp_size:
                    continue                                  # skip last gulp
                if self.update_pending:
                    self.update_command_vals()
                    self.log.info("CORR PART OUTPUT >> Updating destination to %s:%s"
                                  % (self.command_vals['dest_ip'], self.command_vals['dest_port']))
                self.update_stats({'curr_sample': this_gulp_time})
                curr_time = time.time()
                acquire_time = curr_time - prev_time
                prev_time = curr_time
                if self.command_vals['dest_ip'] != "0.0.0.0" or self.sink is not None:
                    data = ispan.data
                    idata = (data.numpy() if hasattr(data, 'numpy') else np.asarray(data)).view(np.int32) \
                        .reshape(nchan, nvis, 2).transpose(1, 0, 2)
                    dout[...] = idata                          # formats the data big-endian for sending (:459)
                    self.send_packets_py(dout, baselines, ihdr['sync_time'], this_gulp_time, bw_hz, sfreq,
                                         upstream_acc_len, nchan, chan0)
                curr_time = time.time()
                process_time = curr_time - prev_time
                prev_time = curr_time
                self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': 0, 'process_time': process_time})
                self.update_stats()
                this_gulp_time += upstream_acc_len
