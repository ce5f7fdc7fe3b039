"""BeamformOutput: integrated power beams -> one packet per (beam, time sample).

Counterpart of pipeline/lwa352_pipeline/blocks/beamform_output_block.py (constructor :226-248, destination
handling :250-301, send_packets_python :303-309, main :311-378).  Pure host code: the span written by
BeamformSumBeams is f32[nbeam][ntime_gulp][nchan][4] (beamform_sum_beams_block.py:220-222), so the payload of
beam b at time t -- f32[nchan][4] = [XX, YY, re(XY), im(XY)] per channel, native endianness -- is already
contiguous and is sent as it lies.

Header: the 18-byte "PBEAM" struct of the reference docstring (:167-180), built exactly as
`send_packets_python` does (:304-306): six uint8 (server, beam, gbe/tuning, nchan, nbeam, nserver), two
big-endian uint16 (navg, chan0), one big-endian uint64 seq that advances by ntime_gulp*navg per packet (:309).
The reference's default transmit path hands the same fields to bifrost's `pbeam1` UDPTransmit (an absent
submodule; the reference marks its own python path "not tested", :240): this block always uses the python
format.  Destinations: `dest_ip` / `dest_port` lists indexed by beam modulo their length (:131-146), "0.0.0.0"
skips a beam; a `sink(beam, packet_bytes)` callable receives every packet in tests.
"""
import json
import socket
import time

import numpy as np

from ..proclog import cpu_affinity
from .block_base import Block


class BeamformOutput(Block):
    def __init__(self, log, iring,
                 guarantee=True, core=-1, etcd_client=None, dest_port=10000,
                 ntime_gulp=480, pipeline_idx=1, nchan=96, nbeam=16, sink=None):
        super(BeamformOutput, self).__init__(log, iring, None, guarantee, core, etcd_client=etcd_client)
        self.ntime_gulp = ntime_gulp
        self.pipeline_idx = pipeline_idx
        self.nchan = nchan
        self.nbeam = nbeam
        self.sink = sink
        self.define_command_key('dest_ip', type=list, initial_val=['0.0.0.0'])
        self.define_command_key('dest_port', type=list, initial_val=[dest_port])
        self.update_command_vals()
        self.socks = [None] * nbeam
        self.beam_ips = [None] * nbeam
        self.beam_ports = [None] * nbeam
        self._update_destinations()

    def _update_destinations(self):
        """Beam i goes to dest_ip[i % len], dest_port[i % len] (:131-146, 268-292)."""
        self.update_command_vals()
        for beam in range(self.nbeam):
            ip = self.command_vals['dest_ip'][beam % len(self.command_vals['dest_ip'])]
            port = self.command_vals['dest_port'][beam % len(self.command_vals['dest_port'])]
            if self.beam_ips[beam] == ip and self.beam_ports[beam] == port:
                continue
            self.beam_ips[beam], self.beam_ports[beam] = ip, port
            if self.socks[beam] is not None:
                self.socks[beam].close()
                self.socks[beam] = None
            if ip != '0.0.0.0':
                self.log.info("Sending beam %d to %s:%d" % (beam, ip, port))
                self.socks[beam] = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
                self.socks[beam].connect((ip, port))
        self.update_stats({'dest_ip': self.beam_ips, 'dest_port': self.beam_ports, 'last_update_time': time.time()})

    def send_packets_python(self, src, tuning, nsrc, navg, chan0, seq, b, d):
        """d: f32[ntime_gulp][nchan*4] of beam b (:303-309)."""
        header0 = np.array([src, b, tuning, self.nchan, self.nbeam, nsrc], dtype='>u1').tobytes() \
            + np.array([navg, chan0], dtype='>u2').tobytes()
        header1 = np.array([seq], dtype='>u8')
        for t in range(self.ntime_gulp):
            pkt = header0 + header1.tobytes() + d[t].tobytes()
            if self.sink is not None:
                self.sink(b, pkt)
            if self.socks[b] is not None:
                self.socks[b].send(pkt)
            header1[0] += self.ntime_gulp * navg

    def main(self):
        cpu_affinity.set_core(self.core)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})
        prev_time = time.time()
        for iseq in self.iring.read(guarantee=self.guarantee):
            ihdr = json.loads(iseq.header.tostring())
            this_gulp_time = ihdr['seq0']
            upstream_acc_len = ihdr['acc_len']
            assert self.nchan == ihdr['nchan']
            assert self.nbeam == ihdr['nbeam']
            nbit = ihdr['nbit']
            npipeline = ihdr['system_nchan'] // self.nchan
            chan0 = ihdr['chan0']
            npol = ihdr['npol']
            igulp_size = self.ntime_gulp * self.nchan * self.nbeam * npol ** 2 * nbit // 8
            for ispan in iseq.read(igulp_size):
                if ispan.size < igulp_size:
                    continue                                  # ignore final gulp
                if self.update_pending:
                    self._update_destinations()
                self.update_stats({'curr_sample': this_gulp_time})
                curr_time = time.time()
                acquire_time = curr_time - prev_time
                prev_time = curr_time
                data = ispan.data
                idata = (data.numpy() if hasattr(data, 'numpy') else np.asarray(data)).view(np.float32) \
                    .reshape(self.nbeam, self.ntime_gulp, self.nchan * npol ** 2)
                for beam in range(self.nbeam):
                    if self.beam_ips[beam] != '0.0.0.0' or self.sink is not None:
                        try:
                            self.send_packets_python(self.pipeline_idx - 1, 1, npipeline, upstream_acc_len, chan0,
                                                     this_gulp_time, beam, idata[beam])
                        except OSError as e:
                            self.log.error("BEAM OUTPUT >> Sending error (beam %d): %s" % (beam, str(e)))
                curr_time = time.time()
                process_time = curr_time - prev_time
                prev_time = curr_time
                self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': 0, 'process_time': process_time})
                self.update_stats({'last_end_sample': this_gulp_time})
                this_gulp_time += upstream_acc_len * self.ntime_gulp
