"""CorrOutputFull: turn long-integration visibilities (xGPU order) into per-baseline packets.

Counterpart of pipeline/lwa352_pipeline/blocks/corr_output_full_block.py (constructor :363-417,
send_packets_py :436-482, send_packets_bf :498-548, check_against_file :550-603, main :605-707).

The reference reorders the whole matrix on the host (`bfXgpuReorder`, 380 MB for 352 stands) and then slices
one payload per dual-pol baseline (s0 <= s1).  Here the reorder, the conjugation and the per-baseline
transposition are one device kernel (`xengXgpuPacketize`): the block receives the payloads already in
sending order, int32[nbaseline][npol][npol][nchan][2] (python format) or [nbaseline][nchan][npol][npol][2]
(COR format), copies them to the host once and only prepends headers.

Packet formats:
* ``use_cor_fmt=False``: 56-byte big-endian header `>QQ2d4I` + `>2I` followed by the payload in native
  byte order, exactly as :443-463 builds it (docs/source/outputs.rst:29-46 describes the fields).
* ``use_cor_fmt=True``: the reference hands payloads to bifrost's `cor` packet writer (an absent
  submodule).  The 32-byte header follows the `struct cor` of the reference docstring (:213-226); how
  bifrost fills it (sync word 0x5CDEC0DE, id 0x02 with the 24-bit tuning word, 1-based stands, time_tag,
  navg, first channel, gain) is restated from the LWA COR convention and is NOT pinned by anything in the
  reference tree.

Destinations: UDP (`dest_ip`/`dest_port`), a file (`dest_file`), and/or a `sink(packet_bytes)` callable
(tests).  `dest_ip == "0.0.0.0"` with no file and no sink skips sending, as in the reference (:689-699).
"""
import json
import os
import socket
import struct
import time

import numpy as np

from ..backend import default_backend
from ..ndarray import XArray, copy_array
from ..proclog import cpu_affinity
from .block_base import Block

COR_SYNC_WORD = 0x5CDEC0DE
COR_ID = 0x02


class CorrOutputFull(Block):
    def __init__(self, log, iring,
                 guarantee=True, core=-1, nchan=192, npol=2, nstand=352, etcd_client=None, dest_port=10000,
                 checkfile=None, checkfile_acc_len=1, antpol_to_bl=None, bl_is_conj=None, use_cor_fmt=True,
                 nchan_sum=1, pipeline_idx=1, npipeline=1, gpu=-1, backend=None, sink=None):
        super(CorrOutputFull, self).__init__(log, iring, None, guarantee, core, etcd_client=etcd_client)
        from .block_base import declare_streams
        declare_streams(iring, 'consumer', 'copy')
        self._bf = backend if backend is not None else default_backend()
        self.nchan_sum = nchan_sum
        self.pipeline_idx = pipeline_idx
        self.npipeline = npipeline
        self.nchan, self.npol, self.nstand = nchan, npol, nstand
        self.gpu = gpu
        self.sink = sink
        self.matlen = nchan * (nstand // 2 + 1) * (nstand // 4) * npol * npol * 4
        # (:377-381) tuning word of the COR header
        wrapped_idx = ((self.pipeline_idx - 1) % self.npipeline) + 1
        self.tuning = ((self.nchan_sum << 16) | (self.npipeline << 8) | wrapped_idx) & 0x00FFFFFF
        self.igulp_size = self.matlen * 8
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        dev = self._bf.space_in
        host = 'cuda_host' if dev == 'cuda' else 'system'
        self.nbl = nstand * (nstand + 1) // 2
        self.payload_words = npol * npol * nchan * 2
        self._map_shape = [nstand, nstand, npol, npol]
        self.antpol_to_bl = np.zeros(self._map_shape, dtype=np.int32)
        self.bl_is_conj = np.zeros(self._map_shape, dtype=np.int32)
        if antpol_to_bl is not None:
            self.antpol_to_bl[...] = antpol_to_bl
        if bl_is_conj is not None:
            self.bl_is_conj[...] = bl_is_conj
        self._bl_dev = XArray(shape=self._map_shape, dtype='i32', space=dev)
        self._cj_dev = XArray(shape=self._map_shape, dtype='i32', space=dev)
        self._maps_dirty = True
        self._in_dev = None                                   # staging for host-space input rings
        self._payload_dev = XArray(shape=[self.nbl, self.payload_words], dtype='i32', space=dev)
        self._payload_host = XArray(shape=[self.nbl, self.payload_words], dtype='i32', space=host)
        self.payloads = None                                  # numpy view of the last integration's payloads
        self.dump_size = nstand * (nstand + 1) * npol * npol * nchan * 2 * 4 / 2.

        self.checkfile_acc_len = checkfile_acc_len
        self.check_results = []                               # (good, bad) per checked integration
        if checkfile is None:
            self.checkfile = None
        else:
            self.checkfile = open(checkfile, 'rb')
            self.checkfile_nbytes = os.path.getsize(checkfile)
            self.log.info("CORR OUTPUT >> Checkfile %s (%d bytes, accumulation length %d)"
                          % (self.checkfile.name, self.checkfile_nbytes, self.checkfile_acc_len))
        self.use_cor_fmt = use_cor_fmt
        self.output_file = None
        self.sock = None
        self.frame_number = 0

        self.define_command_key('dest_ip', type=str, initial_val='0.0.0.0')
        self.define_command_key('dest_file', type=str, initial_val='')
        self.define_command_key('dest_port', type=int, initial_val=dest_port)
        self.define_command_key('max_mbps', type=int, initial_val=-1)
        self.update_command_vals()

    # ---- golden-file check (:419-434, 550-603) -------------------------------------------------
    def get_checkfile_corr(self, t):
        """One integration [nchan, nstand, nstand, npol, npol] complex128 of the test file, wrapping at the end."""
        dim = np.array([self.nchan, self.nstand, self.nstand, self.npol, self.npol])
        nbytes = int(dim.prod()) * 2 * 8
        seekloc = (nbytes * t) % self.checkfile_nbytes
        self.checkfile.seek(seekloc)
        raw = self.checkfile.read(nbytes)
        if len(raw) != nbytes:
            self.log.error("CORR OUTPUT >> Failed to get correlation matrix from checkfile")
            return np.zeros(dim, dtype=complex)
        return np.frombuffer(raw, dtype=complex).reshape(dim)

    def baseline_view(self, k):
        """Payload k as [p0][p1][chan][2] whatever the packet format."""
        p = self.payloads[k]
        if self.use_cor_fmt:
            return p.reshape(self.nchan, self.npol, self.npol, 2).transpose(1, 2, 0, 3)
        return p.reshape(self.npol, self.npol, self.nchan, 2)

    def check_against_file(self, upstream_acc_len, upstream_start_time):
        assert upstream_acc_len % self.checkfile_acc_len == 0, "CORR OUTPUT >> Testfile acc len not compatible with pipeline acc len"
        assert upstream_start_time % self.checkfile_acc_len == 0, "CORR OUTPUT >> Testfile acc len not compatible with pipeline start time"
        nblocks = upstream_acc_len // self.checkfile_acc_len
        dtest = np.zeros([self.nchan, self.nstand, self.nstand, self.npol, self.npol], dtype=complex)
        for i in range(nblocks):
            dtest += self.get_checkfile_corr(upstream_start_time // self.checkfile_acc_len + i)
        good = bad = 0
        k = 0
        for s0 in range(self.nstand):
            for s1 in range(s0, self.nstand):
                got = self.baseline_view(k)                                  # [p0][p1][chan][2]
                exp = dtest[:, s0, s1].transpose(1, 2, 0)                    # [p0][p1][chan]
                ok_r = np.all(got[..., 0] == exp.real, axis=2)
                ok_i = np.all(got[..., 1] == exp.imag, axis=2)
                nbad = int((~ok_r).sum() + (~ok_i).sum())
                if nbad:
                    self.log.error("CORR OUTPUT >> test vector mismatch! stands [%d, %d]: %d of %d components"
                                   % (s0, s1, nbad, 2 * self.npol * self.npol))
                bad += nbad
                good += 2 * self.npol * self.npol - nbad
                k += 1
        (self.log.error if bad else self.log.info)("CORR OUTPUT >> test vector check complete. Good: %d, Bad: %d" % (good, bad))
        self.check_results.append((good, bad))
        return bad == 0

    # ---- packets ----------------------------------------------------------------------------------
    def _emit(self, pkt):
        if self.sink is not None:
            self.sink(pkt)
        if self.output_file is not None:
            self.output_file.write(pkt)
        elif self.sock is not None and self.command_vals['dest_ip'] != "0.0.0.0":
            self.sock.sendto(pkt, (self.command_vals['dest_ip'], self.command_vals['dest_port']))

    def _throttle(self, state, pkt_payload_bits):
        """max_mbps pacing in blocks of ~1 MB (:468-478)."""
        if self.command_vals['max_mbps'] <= 0:
            return
        state[0] += pkt_payload_bits
        if state[0] > 8000000:
            min_time = state[0] / (1.e6 * self.command_vals['max_mbps'])
            delay = min_time - (time.time() - state[1])
            if delay > 0:
                time.sleep(delay)
            state[0], state[1] = 0, time.time()

    def send_packets_py(self, sync_time, this_gulp_time, bw_hz, sfreq, upstream_acc_len, chan0, verbose=False):
        cpu_affinity.set_core(self.core)
        start_time = time.time()
        header_static = struct.pack(">QQ2d4I", sync_time, this_gulp_time, bw_hz, sfreq,
                                    upstream_acc_len, self.nchan, chan0, self.npol)
        pkt_payload_bits = self.nchan * self.npol * self.npol * 8 * 8
        state = [0, time.time()]
        k = 0
        for s0 in range(self.nstand):
            for s1 in range(s0, self.nstand):
                self._emit(header_static + struct.pack(">2I", s0, s1) + self.payloads[k].tobytes())
                k += 1
                self._throttle(state, pkt_payload_bits)
        elapsed = max(time.time() - start_time, 1e-9)
        gbps = 8 * self.dump_size / elapsed / 1e9
        if verbose:
            self.log.info("CORR OUTPUT >> Sending complete for time %d in %.2f seconds (%f Gb/s)" % (this_gulp_time, elapsed, gbps))
        self.update_stats({'output_gbps': gbps})

    def cor_header(self, time_tag, chan0, gain, navg, s0, s1):
        """32-byte `struct cor` (:213-226), big-endian; field fill restated from the LWA COR convention."""
        word1 = (COR_ID << 24) | (self.tuning & 0x00FFFFFF)
        return struct.pack(">IIIhhqihh", COR_SYNC_WORD, word1, 0, chan0, gain, time_tag, navg, s0 + 1, s1 + 1)

    def send_packets_bf(self, time_tag, chan0, gain, navg, verbose=False):
        cpu_affinity.set_core(self.core)
        start_time = time.time()
        pkt_payload_bits = self.nchan * self.npol * self.npol * 8 * 8
        state = [0, time.time()]
        k = 0
        for s0 in range(self.nstand):
            for s1 in range(s0, self.nstand):
                self._emit(self.cor_header(time_tag, chan0, gain, navg, s0, s1) + self.payloads[k].tobytes())
                k += 1
                self._throttle(state, pkt_payload_bits)
        elapsed = max(time.time() - start_time, 1e-9)
        gbps = 8 * self.dump_size / elapsed / 1e9
        if verbose:
            self.log.info("CORR OUTPUT >> Sending complete for time_tag %d in %.2f seconds (%d Bytes; %.2f Gb/s)"
                          % (time_tag, elapsed, self.dump_size, gbps))
        self.update_stats({'output_gbps': gbps})

    # ---- device work --------------------------------------------------------------------------------
    def _packetize(self, ispan):
        """xGPU-order span -> self.payloads (host), via the device kernel."""
        idata = ispan.data_view('i32')
        if idata.space != self._bf.space_in:                  # host ring (the pipeline's cuda_host CorrAcc output)
            if self._in_dev is None:
                self._in_dev = XArray(shape=[2 * self.matlen], dtype='i32', space=self._bf.space_in)
            copy_array(self._in_dev, idata)                   # (copies are complete on return)
            idata = self._in_dev
        if self._maps_dirty:
            copy_array(self._bl_dev, XArray(np.ascontiguousarray(self.antpol_to_bl)))
            copy_array(self._cj_dev, XArray(np.ascontiguousarray(self.bl_is_conj)))
            self._maps_dirty = False
        rv = self._bf.xgpu_packetize(idata, self._payload_dev, self._bl_dev, self._cj_dev, 1 if self.use_cor_fmt else 0)
        if rv != self._bf.BF_STATUS_SUCCESS:
            self.log.error("xgpuPacketize returned %d" % rv)
            raise RuntimeError("xgpuPacketize returned %d: %s" % (rv, self._bf.last_error()))
        copy_array(self._payload_host, self._payload_dev)
        self.payloads = self._payload_host.numpy().reshape(self.nbl, self.payload_words)

    def _update_destination(self):
        self.update_command_vals()
        if self.output_file is not None:
            self.output_file.close()
            self.output_file = None
        if self.command_vals['dest_file'] != "":
            try:
                self.output_file = open(self.command_vals['dest_file'], "wb")
                self.log.info("CORR OUTPUT >> Updating destination to file %s" % self.command_vals['dest_file'])
            except OSError:
                self.log.error("CORR OUTPUT >> Tried to open file %s for output but failed" % self.command_vals['dest_file'])
        elif self.command_vals['dest_ip'] != "0.0.0.0":
            if self.sock is not None:
                self.sock.close()
            self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
            self.log.info("CORR OUTPUT >> Updating destination to %s:%s (max rate %.1f Mbits/s)"
                          % (self.command_vals['dest_ip'], self.command_vals['dest_port'], self.command_vals['max_mbps']))

    def main(self):
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})
        prev_time = time.time()
        for iseq in self.iring.read(guarantee=self.guarantee):
            self.update_pending = True                        # reprocess commands on each new sequence (:612)
            ihdr = json.loads(iseq.header.tostring())
            this_gulp_time = ihdr['seq0']
            upstream_acc_len = ihdr['acc_len']
            nchan = ihdr['nchan']
            chan0 = ihdr['chan0']
            bw_hz = ihdr['bw_hz']
            print_on_send = True
            samples_per_spectra = int(nchan * ihdr['fs_hz'] / bw_hz) if self.use_cor_fmt else 0
            sfreq = ihdr['sfreq'] if not self.use_cor_fmt else 0.0
            if 'ant_to_bl_id' in ihdr:
                self.antpol_to_bl[...] = ihdr['ant_to_bl_id']
                self._maps_dirty = True
            if 'bl_is_conj' in ihdr:
                self.bl_is_conj[...] = ihdr['bl_is_conj']
                self._maps_dirty = True
            for ispan in iseq.read(self.igulp_size):
                if ispan.size < self.igulp_size:
                    continue                                  # skip last gulp
                if self.update_pending:
                    print_on_send = True
                    self._update_destination()
                self.update_stats({'curr_sample': this_gulp_time})
                curr_time = time.time()
                acquire_time = curr_time - prev_time
                prev_time = curr_time
                self._packetize(ispan)
                if self.checkfile:
                    self.check_against_file(upstream_acc_len, this_gulp_time)
                sending = (self.command_vals['dest_ip'] != "0.0.0.0" or self.command_vals['dest_file'] != ""
                           or self.sink is not None)
                if sending:
                    if self.use_cor_fmt:
                        time_tag = this_gulp_time * samples_per_spectra
                        self.send_packets_bf(time_tag, chan0, 0, upstream_acc_len * samples_per_spectra,
                                             verbose=print_on_send)
                    else:
                        self.send_packets_py(ihdr['sync_time'], this_gulp_time, bw_hz, sfreq,
                                             upstream_acc_len, chan0, verbose=print_on_send)
                elif print_on_send:
                    self.log.info("CORR OUTPUT >> Skipping sending for time %d" % this_gulp_time)
                print_on_send = False
                curr_time = time.time()
                process_time = curr_time - prev_time
                prev_time = curr_time
                self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': 0,
                                          'process_time': process_time})
                self.update_stats()
                this_gulp_time += upstream_acc_len
        if self.checkfile:
            self.checkfile.close()
        if self.output_file is not None:
            self.output_file.close()
