#!/usr/bin/env python3
"""Diagnostic: which block limits config 5 through the Python blocks?  Runs subsets of {Corr, CorrAcc, Beamform, BeamformSumBeams} on
in-repo rings at config-2 size from a zero-copy replay source and prints the wall time per 2400-sample integration.
usage: blocks_probe.py [nint]"""
import json
import logging
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402
from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, Corr, CorrAcc  # noqa: E402
from caltech_bifrost_dsp_amd.ndarray import XArray  # noqa: E402
from caltech_bifrost_dsp_amd.ring import Ring  # noqa: E402

NSTAND, NPOL, NCHAN, NTIME_GULP, ACC_LEN = 352, 2, 96, 480, 2400
NINPUT = NSTAND * NPOL


def run(which, nint, beam_gulp=480, depth=4, ring_gulps=10, in_span=2):
    gps = ACC_LEN // NTIME_GULP
    gulp_bytes = NTIME_GULP * NCHAN * NINPUT
    ring = run.ring
    nbeam, ns = 32, 24
    r_in = Ring("gpu-input", space="cuda")
    r_vis, r_slow = Ring("corr-output", space="cuda"), Ring("corr-slow-output", space="cuda_host")
    r_bf, r_pow = Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    r_in.resize(gulp_bytes, total_span=in_span * gps * gulp_bytes)
    log = logging.getLogger("probe")
    blocks, sinks = [], []
    stamps = []

    def drain(rg, gulp, on_span=None):
        gen = rg.read(guarantee=True)

        def go():
            for iseq in gen:
                for ispan in iseq.read(gulp):
                    if on_span:
                        on_span()
        return threading.Thread(target=go, daemon=True)
    nreaders = 0
    if "corr" in which:
        corr = Corr(log, r_in, r_vis, ntime_gulp=NTIME_GULP, nchan=NCHAN, npol=NPOL, nstand=NSTAND, acc_len=ACC_LEN, autostartat=0, gpu=0)
        blocks.append(corr)
        nreaders += 1
        if "cacc" in which:
            cacc = CorrAcc(log, r_vis, r_slow, nchan=NCHAN, npol=NPOL, nstand=NSTAND, acc_len=50 * ACC_LEN, autostartat=0, gpu=0)
            blocks.append(cacc)
            if "classic" in which:
                r_vis.long_accumulator = None          # CorrAcc maps every span itself
            sinks.append(drain(r_slow, cacc.ogulp_size))
        sinks.append(drain(r_vis, corr.ogulp_size, lambda: stamps.append(time.perf_counter())))
    if "bf" in which:
        Beamform.STREAM_DEPTH = depth
        BeamformSumBeams.STREAM_DEPTH = depth
        bf = Beamform(log, r_in, r_bf, nchan=NCHAN, nbeam=nbeam, ninput=NINPUT, ntime_gulp=beam_gulp, gpu=0)
        rng = np.random.default_rng(7)
        bf.gains_cpu[...] = (rng.uniform(-17, 17, bf.gains_cpu.shape) + 1j * rng.uniform(-17, 17, bf.gains_cpu.shape)).astype(np.complex64)
        blocks.append(bf)
        nreaders += 1
        bstamps = []
        if "sb" in which:
            sb = BeamformSumBeams(log, r_bf, r_pow, nchan=NCHAN, ntime_gulp=beam_gulp, ntime_sum=ns, gpu=0)
            blocks.append(sb)
            sinks.append(drain(r_pow, (nbeam // 2) * (beam_gulp // ns) * NCHAN * 16, lambda: bstamps.append(time.perf_counter())))
        else:
            sinks.append(drain(r_bf, beam_gulp * NCHAN * nbeam * 8, lambda: bstamps.append(time.perf_counter())))
    spans = [XArray(shape=(gulp_bytes,), dtype=np.uint8, space="cuda", _ptr=ring.ptr + g * gulp_bytes, _base=ring) for g in range(ring_gulps)]
    hdr = {'nchan': NCHAN, 'chan0': 0, 'bw_hz': NCHAN * 23925.78125, 'fs_hz': 196000000, 'sfreq': 0.0, 'nstand': NSTAND, 'npol': NPOL,
           'seq0': 0, 'sync_time': 0, 'pipeline_id': 0, 'system_nchan': 32 * NCHAN}

    def source():
        t0 = time.time()
        while len(r_in._readers) < nreaders and time.time() - t0 < 10:
            time.sleep(0.002)
        with r_in.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header=json.dumps(hdr), nringlet=1) as oseq:
                for k in range(nint * gps):
                    oseq.commit_external(spans[k % ring_gulps])
    ths = sinks + [threading.Thread(target=b.main, daemon=True) for b in blocks] + [threading.Thread(target=source, daemon=True)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    el = time.perf_counter() - t0
    nw = nint // 4
    if stamps and len(stamps) > nw + 1:
        per = (stamps[-1] - stamps[nw]) / (len(stamps) - 1 - nw) * 1e3
    elif "bf" in which and len(bstamps) > 10:
        k = len(bstamps) // 4
        per = (bstamps[-1] - bstamps[k]) / (len(bstamps) - 1 - k) * (ACC_LEN / beam_gulp) * 1e3
    else:
        per = el / nint * 1e3
    print("%-22s beam gulp %4d depth %d input ring %d integrations: %.4f ms per integration" % ("+".join(which), beam_gulp, depth, in_span, per), flush=True)


def main():
    nint = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ffi.call("xengSetDevice", 0)
    gulp_bytes = NTIME_GULP * NCHAN * NINPUT
    run.ring = ffi.DeviceBuffer(10 * gulp_bytes)
    rs = np.random.RandomState(0xdeadbeef)
    for g in range(10):
        run.ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
    for which in (["corr"], ["corr", "cacc"], ["corr", "cacc", "classic"], ["corr"], ["corr", "cacc"], ["bf"], ["bf", "sb"], ["corr", "bf"], ["corr", "bf", "sb"], ["corr", "cacc", "bf", "sb"]):
        run(which, nint)
    for which in (["bf", "sb"], ["corr", "bf", "sb"], ["corr", "cacc", "bf", "sb"], ["corr", "cacc", "bf", "sb"]):
        run(which, nint, beam_gulp=960)
    run(["corr", "cacc", "bf", "sb"], nint, beam_gulp=960, depth=8)
    run(["corr", "cacc", "bf", "sb"], nint, beam_gulp=960, depth=2)
    for k in (4, 8, 4):
        run(["corr", "cacc", "bf", "sb"], 2 * nint, beam_gulp=960, in_span=k)
    run(["corr", "cacc", "bf", "sb"], 2 * nint, beam_gulp=960, depth=2, in_span=4)


if __name__ == "__main__":
    main()
