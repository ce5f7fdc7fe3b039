#!/usr/bin/env python3
"""A/B of the config-2 streaming step between environment variants of the X-engine, interleaved in ONE process on one
device (guide rule 24): every round re-initialises the library under each variant's environment (the switches are read
at xengXgpuInitialize), runs `--steps` streaming integrations after a warm-up and prints ms per step; at the end the
median and minimum per variant.  With --alone also the stand-alone launch time (sync per integration, HIP events).

  python3 profiles/ab_step.py --rounds 5 "" XENG_TILING=64
  python3 profiles/ab_step.py --lib A=path/libA.so --lib B=path/libB.so A: B:XENG_TILING=64     (several builds)
"""
import argparse
import ctypes
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NSTAND, NPOL, NCHAN, NTIME_GULP, ACC_LEN = 352, 2, 96, 480, 2400
NINPUT = NSTAND * NPOL


def worker(args):
    """One library build: rounds x variants inside one process; prints 'R <variant index> <ms step> <us alone>' lines."""
    import caltech_bifrost_dsp_amd  # noqa: F401
    from caltech_bifrost_dsp_amd import ffi
    ffi.call("xengSetDevice", 0)
    gulp_bytes = NTIME_GULP * NCHAN * NINPUT
    matlen = NCHAN * 249216
    ring = ffi.DeviceBuffer(args.ring_gulps * gulp_bytes)
    rs = np.random.RandomState(0xdeadbeef)
    for g in range(args.ring_gulps):
        ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
    outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]
    L = ffi.lib()
    gps = ACC_LEN // NTIME_GULP
    variants = [v for v in args.variants]
    for rnd in range(args.rounds):
        for vi, var in enumerate(variants):
            env = dict(kv.split("=", 1) for kv in var.split(",") if kv)
            for k, v in env.items():
                os.environ[k] = v
            try:
                ffi.call("xengXgpuConfigure", NSTAND, NPOL, NCHAN, NTIME_GULP, gps)
                ffi.call("xengXgpuInitialize", 0)
            finally:
                if not args.keep_env:
                    for k in env:
                        del os.environ[k]
            gi = 0

            def step(sync_all=False):
                nonlocal gi
                out = outs[(gi // gps) & 1]
                for g in range(gps):
                    rc = L.xengXgpuKernelAsync(ctypes.c_void_p(ring.ptr + (gi % args.ring_gulps) * gulp_bytes), ctypes.c_void_p(out.ptr), int(g == gps - 1))
                    ffi.check("xengXgpuKernelAsync", rc)
                    gi += 1
                rc = L.xengXgpuSync() if sync_all else L.xengXgpuSyncLag(1)
                assert rc == 0

            for _ in range(args.warm):
                step()
            ffi.call("xengDeviceSynchronize")
            if args.profiling:
                ffi.call("xengXgpuSetProfiling", 1)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            ffi.call("xengDeviceSynchronize")
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            if args.profiling:
                ffi.call("xengXgpuSetProfiling", 0)
            alone = 0.0
            if args.alone:
                tm = (ctypes.c_double * 2)()
                cn = (ctypes.c_int * 2)()
                ffi.call("xengXgpuSetProfiling", 1)
                ffi.call("xengXgpuGetTimes", tm, cn)
                for _ in range(args.alone):
                    step(sync_all=True)
                ffi.call("xengXgpuGetTimes", tm, cn)
                ffi.call("xengXgpuSetProfiling", 0)
                alone = tm[1] / max(cn[1], 1) * 1e3
            ffi.call("xengXgpuDestroy")
            if args.keep_env:
                for k in env:
                    del os.environ[k]
            print("R %d %.4f %.1f" % (vi, ms, alone), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+", help="comma-separated KEY=VALUE lists ('' = default); with --lib: NAME:KEY=VALUE,...")
    ap.add_argument("--lib", action="append", default=[], help="NAME=path of another libxeng build (XENG_LIB); each build runs in its own process, rounds alternate between processes")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warm", type=int, default=600)
    ap.add_argument("--alone", type=int, default=0, help="also time N stand-alone launches per round")
    ap.add_argument("--ring-gulps", type=int, default=10)
    ap.add_argument("--profiling", action="store_true", help="HIP-event profiling of every launch on during the streaming loop (as in the timed region of bench.py)")
    ap.add_argument("--keep-env", action="store_true", help="keep a variant's environment set while it runs (switches read per launch: XENG_ABLATE of the diagnostic build)")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    libs = dict(kv.split("=", 1) for kv in args.lib)
    if not libs:          # one build: its variants alternate inside one child process per round
        libs = {"": ""}
        args.variants = [":" + v for v in args.variants]
    res = {v: [] for v in args.variants}
    alone = {v: [] for v in args.variants}
    by_lib = {}
    for v in args.variants:
        name, _, env = v.partition(":")
        by_lib.setdefault(name, []).append((v, env))
    # one child per build and round-robin over the builds per round: variants of one build alternate inside its process
    for rnd in range(args.rounds):
        for name, lst in by_lib.items():
            env = dict(os.environ)
            if libs.get(name):
                env["XENG_LIB"] = libs[name]
            cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--rounds", "1", "--steps", str(args.steps), "--warm", str(args.warm),
                   "--alone", str(args.alone), "--ring-gulps", str(args.ring_gulps)] + (["--profiling"] if args.profiling else []) + (["--keep-env"] if args.keep_env else []) + [e for _, e in lst]
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            if out.returncode:
                sys.stderr.write(out.stderr[-2000:])
                raise SystemExit("worker failed")
            for line in out.stdout.splitlines():
                if line.startswith("R "):
                    _, vi, ms, al = line.split()
                    key = lst[int(vi)][0]
                    res[key].append(float(ms))
                    alone[key].append(float(al))
            print("round %d %-10s " % (rnd, name) + "  ".join("%s=%.4f" % (k or "default", res[k][-1]) for k, _ in lst), flush=True)
    for v in args.variants:
        r = res[v]
        line = "%-40s ms/step median %.4f min %.4f max %.4f (n=%d)" % (v or "default", statistics.median(r), min(r), max(r), len(r))
        if args.alone:
            line += "   alone us median %.1f min %.1f" % (statistics.median(alone[v]), min(alone[v]))
        print(line)


if __name__ == "__main__":
    main()
