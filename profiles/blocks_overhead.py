#!/usr/bin/env python3
"""Diagnostic (CPU only): interpreter time per 2400-sample integration of the config-5 block topology
(Corr -> CorrAcc fused, Beamform -> BeamformSumBeams, source + three sinks) with a backend whose compute calls do nothing,
on system-space in-repo rings.  What remains is what the Python side costs per gulp; with --profile, a cProfile listing
per block thread.  usage: blocks_overhead.py [nint] [--profile] [--which corr,cacc,bf,sb]"""
import argparse
import cProfile
import io
import json
import logging
import os
import pstats
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, Corr, CorrAcc  # noqa: E402
from caltech_bifrost_dsp_amd.ndarray import XArray  # noqa: E402
from caltech_bifrost_dsp_amd import ring as ringmod  # noqa: E402
from fake_backend import OracleBackend  # noqa: E402  (test infrastructure: only its no-compute shell is used here)


class _PyRingNoFill(ringmod.PyRing):
    """System-space spans without the zero fill (the device rings recycle pooled allocations: no fill there either)."""
    _free = {}

    def _alloc_span(self, nbytes):
        buf = self._free.get(nbytes)
        if buf is None:
            buf = self._free[nbytes] = np.empty(nbytes, dtype=np.uint8)
        return XArray(shape=(nbytes,), dtype=np.uint8, space=self.space, _ptr=buf.ctypes.data, _base=buf)


def Ring(name, space="system"):
    if ringmod.IMPLEMENTATION == "python":
        return _PyRingNoFill(name, space=space)
    r = ringmod.NativeRing(name, space=space)
    r.set_recycle(True)
    return r


class NullBackend(OracleBackend):
    def bfXgpuKernel(self, i, o, d): return 0
    def bfXgpuKernelAsync(self, i, o, d): return 0
    def bfXgpuKernelAsyncAcc(self, i, o, d, acc, m): return 0
    def map_assign_i32(self, a, b): return 0
    def map_add_i32(self, a, b): return 0
    def bfBeamformRun(self, i, o, w, version=0): return 0
    def bfBeamformIntegrate(self, i, o, n): return 0
    def beam_mark(self): return 1
    def beam_wait(self, t): pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("nint", nargs="?", type=int, default=2000)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--which", default="corr,cacc,bf,sb")
    ap.add_argument("--switch", type=float, default=0.0, help="sys.setswitchinterval (0 = leave the default)")
    ap.add_argument("--ring", default=ringmod.IMPLEMENTATION, help="native | python: which implementation of the ring protocol")
    a = ap.parse_args()
    ringmod.IMPLEMENTATION = a.ring
    if a.switch:
        sys.setswitchinterval(a.switch)
    which = a.which.split(",")
    NSTAND, NPOL, NCHAN, NT, ACC = 16, 2, 4, 480, 2400
    NIN = NSTAND * NPOL
    gps = ACC // NT
    gulp = NT * NCHAN * NIN
    be = NullBackend()
    log = logging.getLogger("ovh")
    r_in = Ring("gpu-input", space="system")
    r_vis, r_slow = Ring("corr-output", space="system"), Ring("corr-slow-output", space="system")
    r_bf, r_pow = Ring("bf-output", space="system"), Ring("bf-pow-output", space="system")
    r_in.resize(gulp, total_span=2 * gps * gulp)
    r_pow.space = "cuda_host"      # (a label only -- the memory stays host memory: BeamformSumBeams then takes its streaming path, the one the pipeline runs)
    blocks, sinks, stamps = [], [], []

    def drain(rg, g, on=None):
        gen = rg.read(guarantee=True)

        def go():
            for iseq in gen:
                for _ in iseq.read(g):
                    if on:
                        on()
        return threading.Thread(target=go, daemon=True)
    nreaders = 0
    nbeam, ns = 32, 24
    if "corr" in which:
        corr = Corr(log, r_in, r_vis, ntime_gulp=NT, nchan=NCHAN, npol=NPOL, nstand=NSTAND, acc_len=ACC, autostartat=0, gpu=-1, backend=be)
        blocks.append(corr)
        nreaders += 1
        if "cacc" in which:
            cacc = CorrAcc(log, r_vis, r_slow, nchan=NCHAN, npol=NPOL, nstand=NSTAND, acc_len=50 * ACC, autostartat=0, gpu=-1, backend=be)
            blocks.append(cacc)
            sinks.append(drain(r_slow, cacc.ogulp_size))
        sinks.append(drain(r_vis, corr.ogulp_size, lambda: stamps.append(time.perf_counter())))
    if "bf" in which:
        bf = Beamform(log, r_in, r_bf, nchan=NCHAN, nbeam=nbeam, ninput=NIN, ntime_gulp=NT, gpu=-1, backend=be)
        blocks.append(bf)
        nreaders += 1
        bst = []
        if "sb" in which:
            sb = BeamformSumBeams(log, r_bf, r_pow, nchan=NCHAN, ntime_gulp=NT, ntime_sum=ns, gpu=-1, backend=be)
            blocks.append(sb)
            sinks.append(drain(r_pow, (nbeam // 2) * (NT // ns) * NCHAN * 16, lambda: bst.append(time.perf_counter())))
        else:
            sinks.append(drain(r_bf, NT * NCHAN * nbeam * 8, lambda: bst.append(time.perf_counter())))
    spans = [XArray(shape=(gulp,), dtype=np.uint8, space="system") for _ in range(10)]
    hdr = {'nchan': NCHAN, 'chan0': 0, 'bw_hz': NCHAN * 23925.78125, 'fs_hz': 196000000, 'sfreq': 0.0, 'nstand': NSTAND, 'npol': NPOL,
           'seq0': 0, 'sync_time': 0, 'pipeline_id': 0, 'system_nchan': 32 * NCHAN}

    def source():
        while len(r_in._readers) < nreaders:
            time.sleep(0.001)
        with r_in.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header=json.dumps(hdr)) as oseq:
                for k in range(a.nint * gps):
                    oseq.commit_external(spans[k % len(spans)])
    profs = {}

    cpu = {}

    def wrap(name, fn):
        if not a.profile:
            def run_cpu():
                t = time.thread_time()
                fn()
                cpu[name] = time.thread_time() - t
            return run_cpu

        def run():
            pr = cProfile.Profile()
            profs[name] = pr
            pr.runcall(fn)
        return run
    threads = [threading.Thread(target=wrap(type(b).__name__, b.main), daemon=True) for b in blocks] + sinks
    for t in threads:
        t.start()
    t0 = time.perf_counter()
    src = threading.Thread(target=wrap("source", source), daemon=True)
    src.start()
    src.join()
    for t in threads:
        t.join(60)
    el = time.perf_counter() - t0
    print("ring %s, %s: %d integrations in %.3f s = %.1f us of interpreter per integration (%d gulps each)" % (a.ring, a.which, a.nint, el, el / a.nint * 1e6, gps))
    if cpu:
        print("    CPU time per integration (thread_time): " + ", ".join("%s %.0f us" % (k, v / a.nint * 1e6) for k, v in cpu.items()) +
              " | sum %.0f us" % (sum(cpu.values()) / a.nint * 1e6))
    for name, pr in profs.items():
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
        print("=====", name)
        print("\n".join(s.getvalue().splitlines()[:32]))


if __name__ == "__main__":
    main()
