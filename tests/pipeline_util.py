"""Helpers to run hot-path blocks as threads on in-repo rings (the way lwa352-pipeline.py:296-302
runs one thread per block)."""
import json
import logging
import threading
import time

import numpy as np

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd.ring import Ring

LOG = logging.getLogger("xeng-test")


def source_header(nchan, nstand, npol, seq0=0, chan0=0, sfreq=0.0, chan_bw=23925.78125, **extra):
    """The sequence header contract of the reference's sources (dummy_source_block.py:231-245)."""
    input_to_ant = [[s, p] for s in range(nstand) for p in range(npol)]
    ant_to_input = np.arange(nstand * npol).reshape(nstand, npol).tolist()
    hdr = {'nchan': nchan, 'system_nchan': 32 * nchan, 'chan0': chan0, 'bw_hz': chan_bw * nchan,
           'fs_hz': 196000000, 'sfreq': sfreq, 'nstand': nstand, 'npol': npol, 'seq0': seq0,
           'input_to_ant': input_to_ant, 'ant_to_input': ant_to_input, 'sync_time': 0, 'pipeline_id': 0}
    hdr.update(extra)
    return hdr


class Source(threading.Thread):
    """Writes sequences [(header dict, bytes-like data, span_nbytes), ...] into a ring."""

    def __init__(self, ring, sequences, wait_readers=1, gap=0.0):
        super().__init__(daemon=True)
        self.ring, self.sequences, self.wait_readers, self.gap = ring, sequences, wait_readers, gap

    def run(self):
        t0 = time.time()
        while len(self.ring._readers) < self.wait_readers and time.time() - t0 < 10:
            time.sleep(0.005)
        with self.ring.begin_writing() as oring:
            for k, (hdr, data, span) in enumerate(self.sequences):
                data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
                with oring.begin_sequence(time_tag=k, header=json.dumps(hdr), nringlet=1) as oseq:
                    for off in range(0, data.size, span):
                        n = min(span, data.size - off)
                        with oseq.reserve(n) as sp:
                            sp.data[...] = data[off:off + n]          # host -> ring space (H2D for 'cuda')
                        if self.gap:
                            time.sleep(self.gap)


class GatedSource(threading.Thread):
    """Writes one sequence gulp by gulp; before gulp k it waits for gates[k] (a threading.Event) when there is one -- so a test
    can make something happen between two named gulps by construction, not by timing."""

    def __init__(self, ring, hdr, data, span, gates):
        super().__init__(daemon=True)
        self.ring, self.hdr, self.data, self.span, self.gates = ring, hdr, data.reshape(-1).view(np.uint8), span, gates
        self.written = 0

    def run(self):
        t0 = time.time()
        while len(self.ring._readers) < 1 and time.time() - t0 < 10:
            time.sleep(0.002)
        with self.ring.begin_writing() as oring:
            with oring.begin_sequence(time_tag=0, header=json.dumps(self.hdr), nringlet=1) as oseq:
                for k, off in enumerate(range(0, self.data.size, self.span)):
                    ev = self.gates.get(k)
                    if ev is not None:
                        assert ev.wait(20), "gate %d never opened" % k
                    n = min(self.span, self.data.size - off)
                    with oseq.reserve(n) as sp:
                        sp.data[...] = self.data[off:off + n]
                    self.written = k + 1


def wait_for(cond, what, timeout=20):
    t0 = time.time()
    while not cond():
        assert time.time() - t0 < timeout, "timed out waiting for " + what
        time.sleep(0.002)


class Sink(threading.Thread):
    """Collects (header, time_tag, [span bytes as np.uint8 copies]) per sequence."""

    def __init__(self, ring, gulp_nbytes, on_span=None):
        super().__init__(daemon=True)
        self.ring, self.gulp = ring, gulp_nbytes
        self.sequences = []
        self.on_span = on_span
        self._gen = ring.read(guarantee=True)       # register before anything is written

    def run(self):
        for iseq in self._gen:
            hdr = json.loads(iseq.header.tostring())
            spans = []
            self.sequences.append((hdr, iseq.time_tag, spans))
            for ispan in iseq.read(self.gulp):
                if ispan.size < self.gulp:
                    continue
                spans.append(ispan.data.numpy().copy())
                if self.on_span:
                    self.on_span(hdr, len(spans))


def run_blocks(blocks, source, sinks, timeout=60):
    ths = [threading.Thread(target=b.main, daemon=True) for b in blocks]
    for s in sinks:
        s.start()
    for t in ths:
        t.start()
    source.start()
    for t in [source] + ths + list(sinks):
        t.join(timeout)
        assert not t.is_alive(), "pipeline thread did not finish: %r" % (t,)
