#!/usr/bin/env python3
"""Diagnostic (round 4, review item 7): Beamform -> BeamformSumBeams alone on device rings, fed by the zero-copy replay source,
with 480-sample beamformer gulps (one launch per input span) and with 960-sample gulps taken as two spans per launch
(read_parts + xengBeamformRunParts).  ms per 2400-sample integration of input.  usage: beam_pair_probe.py [integrations]"""
import json
import logging
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402
from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams  # noqa: E402
from caltech_bifrost_dsp_amd.ndarray import XArray  # noqa: E402
from caltech_bifrost_dsp_amd.ring import Ring  # noqa: E402

NINT = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ffi.call("xengSetDevice", 0)
gulp_bytes = bench.NTIME_GULP * bench.NCHAN * bench.NINPUT
ring_gulps = 10
ring = ffi.DeviceBuffer(ring_gulps * gulp_bytes)
rs = np.random.RandomState(0xdeadbeef)
for g in range(ring_gulps):
    ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
gps = bench.ACC_LEN // bench.NTIME_GULP
log = logging.getLogger("pair")
for ngulp in (1, 2, 1, 2):
    nt = ngulp * bench.NTIME_GULP
    r_in, r_bf, r_pow = Ring("gpu-input", space="cuda"), Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    r_in.resize(gulp_bytes, total_span=2 * gps * gulp_bytes)
    bf = Beamform(log, r_in, r_bf, nchan=bench.NCHAN, nbeam=32, ninput=bench.NINPUT, ntime_gulp=nt, gpu=0)
    sb = BeamformSumBeams(log, r_bf, r_pow, nchan=bench.NCHAN, ntime_gulp=nt, ntime_sum=24, gpu=0)
    rng = np.random.default_rng(7)
    bf.gains_cpu[...] = (rng.uniform(-17, 17, bf.gains_cpu.shape) + 1j * rng.uniform(-17, 17, bf.gains_cpu.shape)).astype(np.complex64)
    spans = [XArray(shape=(gulp_bytes,), dtype=np.uint8, space="cuda", _ptr=ring.ptr + g * gulp_bytes, _base=ring) for g in range(ring_gulps)]
    hdr = {'nchan': bench.NCHAN, 'chan0': 0, 'bw_hz': bench.NCHAN * 23925.78125, 'fs_hz': 196000000, 'sfreq': 0.0, 'nstand': bench.NSTAND,
           'npol': 2, 'seq0': 0, 'sync_time': 0, 'pipeline_id': 0, 'system_nchan': 32 * bench.NCHAN}
    stamps = []
    nwarm = 200

    def source():
        t0 = time.time()
        while len(r_in._readers) < 1 and time.time() - t0 < 10:
            time.sleep(0.002)
        with r_in.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header=json.dumps(hdr), nringlet=1) as oseq:
                for k in range((nwarm + NINT) * gps):
                    oseq.commit_external(spans[k % ring_gulps])

    gen = r_pow.read(guarantee=True)

    def sink():
        for iseq in gen:
            for ispan in iseq.read(16 * (nt // 24) * bench.NCHAN * 16):
                stamps.append(time.perf_counter())

    ths = [threading.Thread(target=f, daemon=True) for f in (sink, bf.main, sb.main, source)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    per_int = gps / ngulp                       # beam gulps per integration of input
    k0 = int(nwarm * per_int)
    el = stamps[-1] - stamps[k0]
    nint = (len(stamps) - 1 - k0) / per_int
    print("Beamform -> BeamformSumBeams, %4d-sample gulps (%d input span%s per launch): %.4f ms per integration (%d power spans)"
          % (nt, ngulp, "s" if ngulp > 1 else "", el / nint * 1e3, len(stamps)), flush=True)
    del bf, sb, r_in, r_bf, r_pow, gen
