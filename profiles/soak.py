#!/usr/bin/env python3
"""Soak run: the concurrent X-engine pattern of BASELINE config 5 repeated N times on FIXED inputs, every result folded
into a running sum on the device, so that one wrong word in one of the N results shows up in the final comparison.

Each integration contracts the same five gulps, each beamformer gulp reads the same voltages with the same weights, so every
dump / beam gulp / power block / sub-selection / payload set / unpacked gulp must be bit-identical to the one computed
alone on an idle GPU.  Results are added word-wise as int32 bit patterns (xengMapAddI32: wrap-around arithmetic, so floats
are summed as their bit patterns too) and the sum is compared with N x the stand-alone pattern modulo 2^32.  Exercised
together: contraction streams (lag-1 streaming, the dumps of phase 1 also feed two alternating long accumulators), beam stream (weight re-split every third round, Run + Integrate, then the
fused integrated-power mode), map stream, consumer stream (SubSelect, Packetize), staging stream (SNAP2 unpack); phase 3: both
consumers fed from packet slabs read in place (and, now and then, through the scatter of an irregular slab).

This is what found the wrong power sums of round 2: a kernel that is right alone on the GPU and in every
parity test can still be wrong beside the contraction.

usage: soak.py [N]      (default 1500 rounds per phase; prints one line per result and exits non-zero on a mismatch)
tests/test_soak_gpu.py runs a shorter version (with SNAP2 packets from the oracle's emulator) in the GPU suite.
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import caltech_bifrost_dsp_amd  # noqa: F401,E402
from caltech_bifrost_dsp_amd import ffi  # noqa: E402

NSTAND, NPOL, NCHAN, NTIME_GULP, ACC_LEN = 352, 2, 96, 480, 2400
NINPUT = NSTAND * NPOL
NT_B, NB, NS = 960, 32, 24


def times_mod32(ref_u32, n):
    return ((ref_u32.astype(np.uint64) * np.uint64(n)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def soak(N=1500, packets=None, log=print):
    """Returns [(name, results checked, words that differ)].  packets: (slab bytes, npkt, stride, seq0, expected gulp
    uint8[NTIME_GULP*NCHAN*NINPUT]) to include the SNAP2 unpack, or None."""
    L = ffi.lib()
    G = ACC_LEN // NTIME_GULP
    gulp_bytes = NTIME_GULP * NCHAN * NINPUT
    matlen = NCHAN * 249216
    results = []

    def report(name, acc, ref_u32, n):
        got = acc.download(np.uint32, count=ref_u32.size)
        bad = int(np.count_nonzero(got != times_mod32(ref_u32, n)))
        log("  %-34s %d results x %d words: %s" % (name, n, ref_u32.size, "identical" if bad == 0 else "%d WORDS DIFFER" % bad))
        results.append((name, n, bad))

    ffi.call("xengSetDevice", 0)
    ffi.call("xengXgpuConfigure", NSTAND, NPOL, NCHAN, NTIME_GULP, G)
    ffi.call("xengXgpuInitialize", 0)
    rs = np.random.RandomState(0x50a4)
    ring = ffi.DeviceBuffer(G * gulp_bytes)
    for g in range(G):
        ring.upload(rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
    outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(3)]
    rng = np.random.default_rng(0x50a5)
    wts = (rng.uniform(-17, 17, NCHAN * NB * NINPUT) + 1j * rng.uniform(-17, 17, NCHAN * NB * NINPUT)).astype(np.complex64)
    dw = ffi.DeviceBuffer(wts.nbytes).upload(wts)
    nbeam_words = NCHAN * NB * NT_B * 2
    npow_words = (NB // 2) * (NT_B // NS) * NCHAN * 4
    dbeam = [ffi.DeviceBuffer(nbeam_words * 4) for _ in range(2)]
    dpow = [ffi.DeviceBuffer(npow_words * 4) for _ in range(2)]
    nvis, nsum = 4656, 4
    vismap = rng.integers(0, 249216, nvis).astype(np.int32)
    conj = rng.integers(0, 2, nvis).astype(np.int32)
    dmap, dconj = ffi.DeviceBuffer(vismap.nbytes).upload(vismap), ffi.DeviceBuffer(conj.nbytes).upload(conj)
    nsub_words = (NCHAN // nsum) * nvis * 2
    dsub = ffi.DeviceBuffer(nsub_words * 4)
    a2i = np.arange(NINPUT, dtype=np.int32)
    blm = np.zeros(NSTAND * NSTAND * 4, dtype=np.int32)
    cjm = np.zeros_like(blm)
    ffi.call("xengXgpuGetOrder", a2i.ctypes.data, blm.ctypes.data, cjm.ctypes.data)
    dbl, dcj = ffi.DeviceBuffer(blm.nbytes).upload(blm), ffi.DeviceBuffer(cjm.nbytes).upload(cjm)
    npay_words = (NSTAND * (NSTAND + 1) // 2) * 4 * NCHAN * 2
    dpay = ffi.DeviceBuffer(npay_words * 4)
    acc_vis, acc_beam = ffi.DeviceBuffer(2 * matlen * 4), ffi.DeviceBuffer(nbeam_words * 4)
    acc_pow, acc_sub = ffi.DeviceBuffer(npow_words * 4), ffi.DeviceBuffer(nsub_words * 4)
    acc_pay = ffi.DeviceBuffer(npay_words * 4)
    acc_fused = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(2)]      # long accumulators of the fused dumps (alternating)
    if packets is not None:
        slab, npkt, stride, seq0, expect = packets
        dslab = ffi.DeviceBuffer(len(slab)).upload(np.frombuffer(slab, dtype=np.uint8))
        dgulp = [ffi.DeviceBuffer(gulp_bytes) for _ in range(2)]
        acc_gulp = ffi.DeviceBuffer(gulp_bytes)
        ref_gulp = np.ascontiguousarray(expect, dtype=np.uint8).view(np.uint32)

    import ctypes
    _fused, _fp6 = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengXgpuGetPath", ctypes.byref(_fused), ctypes.byref(_fp6))
    can_fuse = _fused.value == 1 and _fp6.value == 0     # (the two-pass X-engine, XENG_RAW=0, has no fused long accumulation)

    def integration(out, acc=None, mode=0):
        for g in range(G):
            if acc is None or not can_fuse:
                ffi.check("kernel", L.xengXgpuKernelAsync(ring.ptr + g * gulp_bytes, out.ptr, int(g == G - 1)))
            else:       # CorrAcc's add fused into the dump
                ffi.check("kernel", L.xengXgpuKernelAsyncAcc(ring.ptr + g * gulp_bytes, out.ptr, int(g == G - 1), acc.ptr, mode))

    # ---- stand-alone results: every call once, on an otherwise idle GPU
    ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, NT_B, NB, 0)
    integration(outs[0])
    ffi.call("xengXgpuSync")
    ref_vis = outs[0].download(np.uint32)
    ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dbeam[0].ptr, dw.ptr, 1))
    ffi.check("int", L.xengBeamformIntegrate(dbeam[0].ptr, dpow[0].ptr, NS))
    ffi.call("xengBeamformSync")
    ref_beam, ref_pow = dbeam[0].download(np.uint32), dpow[0].download(np.uint32)
    ffi.check("sub", L.xengXgpuSubSelect(outs[0].ptr, dsub.ptr, dmap.ptr, dconj.ptr, nvis, nsum))
    ref_sub = dsub.download(np.uint32)
    ffi.check("pack", L.xengXgpuPacketize(outs[0].ptr, dpay.ptr, dbl.ptr, dcj.ptr, 1))
    ref_pay = dpay.download(np.uint32)

    # ---- phase 1: contraction (lag-1 streaming) + beamformer Run/Integrate + CorrAcc-style adds + consumers + ingest
    accs = [acc_vis, acc_beam, acc_pow, acc_sub, acc_pay] + ([acc_gulp] if packets is not None else [])
    for a in accs:
        ffi.call("xengMemset", a.ptr, 0, a.nbytes)
    t0 = time.perf_counter()
    nsubsel = npack = 0
    for n in range(N):
        ffi.call("xengMapSync")                       # the adds of the previous round have released their sources
        if packets is not None:
            ffi.check("unpack", L.xengSnap2UnpackAsync(dslab.ptr, npkt, stride, dgulp[n & 1].ptr, seq0, NTIME_GULP, 0, NCHAN, NINPUT, 1))
        integration(outs[n % 3], acc_fused[n & 1], 1 if n < 2 else 2)
        # the weights "change" every third round (same values, new version): the split / routing kernels run beside the contraction
        ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dbeam[n & 1].ptr, dw.ptr, 1 + n // 3))
        ffi.check("int", L.xengBeamformIntegrate(dbeam[n & 1].ptr, dpow[n & 1].ptr, NS))
        ffi.call("xengXgpuSyncLag", 1)                # dump n-1 is complete (and so is this round's unpack: same stream order)
        if n >= 1:
            prev = outs[(n - 1) % 3]
            ffi.check("map", L.xengMapAddI32(acc_vis.ptr, prev.ptr, 2 * matlen))
            if n % 4 == 0:
                ffi.check("sub", L.xengXgpuSubSelect(prev.ptr, dsub.ptr, dmap.ptr, dconj.ptr, nvis, nsum))
                ffi.check("map", L.xengMapAddI32(acc_sub.ptr, dsub.ptr, nsub_words))
                nsubsel += 1
            if n % 16 == 8:
                ffi.check("pack", L.xengXgpuPacketize(prev.ptr, dpay.ptr, dbl.ptr, dcj.ptr, 1))
                ffi.check("map", L.xengMapAddI32(acc_pay.ptr, dpay.ptr, npay_words))
                npack += 1
        ffi.call("xengBeamformSync")
        ffi.check("map", L.xengMapAddI32(acc_beam.ptr, dbeam[n & 1].ptr, nbeam_words))
        ffi.check("map", L.xengMapAddI32(acc_pow.ptr, dpow[n & 1].ptr, npow_words))
        if packets is not None and n >= 1:
            ffi.check("map", L.xengMapAddI32(acc_gulp.ptr, dgulp[(n - 1) & 1].ptr, gulp_bytes // 4))
    ffi.call("xengXgpuSync")
    ffi.call("xengMapSync")
    ffi.check("map", L.xengMapAddI32(acc_vis.ptr, outs[(N - 1) % 3].ptr, 2 * matlen))
    if packets is not None:
        ffi.check("map", L.xengMapAddI32(acc_gulp.ptr, dgulp[(N - 1) & 1].ptr, gulp_bytes // 4))
    ffi.call("xengMapSync")
    el = time.perf_counter() - t0
    log("phase 1: %d concurrent rounds in %.2f s (%.3f ms each)" % (N, el, el / N * 1e3))
    report("visibility dumps", acc_vis, ref_vis, N)
    if can_fuse:
        ffi.check("map", L.xengMapAddI32(acc_fused[0].ptr, acc_fused[1].ptr, 2 * matlen))
        ffi.call("xengMapSync")
        report("long accumulation in the dumps", acc_fused[0], ref_vis, N)
    report("voltage beams", acc_beam, ref_beam, N)
    report("power sums (Integrate)", acc_pow, ref_pow, N)
    report("sub-selections", acc_sub, ref_sub, nsubsel)
    report("COR payload sets", acc_pay, ref_pay, npack)
    if packets is not None:
        report("unpacked SNAP2 gulps", acc_gulp, ref_gulp, N)
        drops = ctypes.c_int(-1)
        ffi.call("xengSnap2GetAsyncDrops", ctypes.byref(drops))
        log("  packets dropped by the %d unpacks: %d" % (N, drops.value))
        results.append(("SNAP2 drops", N, drops.value))

    # ---- phase 2: the fused integrated-power mode of the beamformer beside the contraction
    ffi.call("xengXgpuSync")
    ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, NT_B, NB, NT_B // NS)
    for _ in range(3):                                # (the first calls after a weight upload take the composed path)
        ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dpow[0].ptr, dw.ptr, 7))
        ffi.call("xengBeamformSync")
    ref_pow2 = dpow[0].download(np.uint32)
    for a in (acc_vis, acc_pow):
        ffi.call("xengMemset", a.ptr, 0, a.nbytes)
    t0 = time.perf_counter()
    for n in range(N):
        ffi.call("xengMapSync")
        integration(outs[n % 3])
        ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dpow[n & 1].ptr, dw.ptr, 7))
        ffi.call("xengXgpuSyncLag", 1)
        if n >= 1:
            ffi.check("map", L.xengMapAddI32(acc_vis.ptr, outs[(n - 1) % 3].ptr, 2 * matlen))
        ffi.call("xengBeamformSync")
        ffi.check("map", L.xengMapAddI32(acc_pow.ptr, dpow[n & 1].ptr, npow_words))
    ffi.call("xengXgpuSync")
    ffi.call("xengMapSync")
    ffi.check("map", L.xengMapAddI32(acc_vis.ptr, outs[(N - 1) % 3].ptr, 2 * matlen))
    ffi.call("xengMapSync")
    el = time.perf_counter() - t0
    log("phase 2: %d concurrent rounds in %.2f s (%.3f ms each)" % (N, el, el / N * 1e3))
    report("visibility dumps", acc_vis, ref_vis, N)
    report("power sums (fused epilogue)", acc_pow, ref_pow2, N)
    p1, p2 = ref_pow.view(np.float32).reshape(-1, 4), ref_pow2.view(np.float32).reshape(-1, 4)
    scale = np.sqrt(p1[:, 0] * p1[:, 1])[:, None] + 1e-30
    log("  fused vs composed power sums: max |diff| / sqrt(XX YY) = %.2e" % float(np.max(np.abs(p2 - p1) / scale)))
    # ---- phase 3 (round 4): the same five gulps handed over as PACKET SLABS, read in place by both consumers beside each
    # other; every other round one of the correlator's slabs, every third round the beamformer's second slab comes with its
    # packets in a shuffled order (the scatter path) -- the results must still be the stand-alone ones
    if can_fuse:
        import struct
        ffi.call("xengXgpuSync")
        ffi.call("xengBeamformInitialize", 0, NINPUT, NCHAN, NT_B, NB, 0)
        nblk, stride = NINPUT // 64, 32 + NCHAN * 64
        npk = NTIME_GULP * nblk
        seq_base = 10 ** 12 + 5
        slabs, slabs_shuffled = [], []
        for g in range(G):
            gulp = ring.download(np.uint8, count=gulp_bytes, offset=g * gulp_bytes)
            pay = gulp.reshape(NTIME_GULP, NCHAN, nblk, 64).transpose(0, 2, 1, 3).reshape(npk, NCHAN * 64)
            slab = np.zeros((npk, stride), dtype=np.uint8)
            slab[:, 32:] = pay
            k = 0
            for t in range(NTIME_GULP):
                for b in range(nblk):
                    slab[k, :32] = np.frombuffer(struct.pack(">QLHHHHLLL", seq_base + g * NTIME_GULP + t, 3, 64, NINPUT, NCHAN, NCHAN, 0, 0, b * 64), dtype=np.uint8)
                    k += 1
            slabs.append(ffi.DeviceBuffer(slab.nbytes).upload(slab))
            slabs_shuffled.append(ffi.DeviceBuffer(slab.nbytes).upload(slab[rng.permutation(npk)]))
        for a in (acc_vis, acc_beam, acc_pow, acc_fused[0], acc_fused[1]):
            ffi.call("xengMemset", a.ptr, 0, a.nbytes)
        N3 = max(6, N // 2)
        t0 = time.perf_counter()
        for n in range(N3):
            ffi.call("xengMapSync")
            for g in range(G):
                src = slabs_shuffled[g] if (n & 1) and g == n % G else slabs[g]
                ffi.check("slab", L.xengXgpuKernelAsyncSlab(src.ptr, npk, stride, seq_base + g * NTIME_GULP, 0, outs[n % 3].ptr, int(g == G - 1),
                                                            acc_fused[n & 1].ptr, 1 if n < 2 else 2))
            second = slabs_shuffled[1] if n % 3 == 2 else slabs[1]
            ffi.check("run", L.xengBeamformRunSlabs(slabs[0].ptr, npk, NTIME_GULP, second.ptr, npk, stride, seq_base, 0, dbeam[n & 1].ptr, dw.ptr, 1 + n // 3))
            ffi.check("int", L.xengBeamformIntegrate(dbeam[n & 1].ptr, dpow[n & 1].ptr, NS))
            ffi.call("xengXgpuSyncLag", 1)
            if n >= 1:
                ffi.check("map", L.xengMapAddI32(acc_vis.ptr, outs[(n - 1) % 3].ptr, 2 * matlen))
            ffi.call("xengBeamformSync")
            ffi.check("map", L.xengMapAddI32(acc_beam.ptr, dbeam[n & 1].ptr, nbeam_words))
            ffi.check("map", L.xengMapAddI32(acc_pow.ptr, dpow[n & 1].ptr, npow_words))
        ffi.call("xengXgpuSync")
        ffi.call("xengMapSync")
        ffi.check("map", L.xengMapAddI32(acc_vis.ptr, outs[(N3 - 1) % 3].ptr, 2 * matlen))
        ffi.check("map", L.xengMapAddI32(acc_fused[0].ptr, acc_fused[1].ptr, 2 * matlen))
        ffi.call("xengMapSync")
        el = time.perf_counter() - t0
        log("phase 3: %d concurrent rounds on packet slabs in %.2f s (%.3f ms each)" % (N3, el, el / N3 * 1e3))
        report("visibility dumps from slabs", acc_vis, ref_vis, N3)
        report("long accumulation from slabs", acc_fused[0], ref_vis, N3)
        report("voltage beams from slabs", acc_beam, ref_beam, N3)
        report("power sums from slabs", acc_pow, ref_pow, N3)
        nfx, nix, nfb, nib = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        ffi.call("xengXgpuGetSlabStats", ctypes.byref(nfx), ctypes.byref(nix))
        ffi.call("xengBeamformGetSlabStats", ctypes.byref(nfb), ctypes.byref(nib))
        # (round 5: both consumers scatter the first shuffled slabs and read the later ones where they lie -- the correlator through
        # offset tables, the beamformer through packet indices; XENG_SLAB_TABLES=1: from the first; =0: never)
        want = (N3 // 2, N3 // 3)
        forced = os.environ.get("XENG_SLAB_TABLES")
        log("  slabs not regular: correlator %d scattered + %d through a table (expected %d in all); beamformer %d scattered + %d through an index (expected %d in all)"
            % (nfx.value, nix.value, want[0], nfb.value, nib.value, want[1]))
        bad = int(nfx.value + nix.value != want[0]) + int(nfb.value + nib.value != want[1])
        if forced == "1":          # (the fp32 beamformer kernel has no index path: its irregular parts are scattered whatever the switch says)
            bad += int(nfx.value != 0) + int(nfb.value != 0 and os.environ.get("XENG_BEAM") != "f32" and not os.environ.get("XENG_BEAM_F32"))
        elif forced == "0":
            bad += int(nix.value != 0) + int(nib.value != 0)
        else:
            bad += int(nfx.value < 1) + int(nfb.value < 1)
        results.append(("slab scatter counts", N3, bad))
        for b in slabs + slabs_shuffled:
            b.free()
    ffi.call("xengBeamformDestroy")
    ffi.call("xengXgpuDestroy")
    return results


if __name__ == "__main__":
    res = soak(int(sys.argv[1]) if len(sys.argv) > 1 else 1500, log=lambda s: print(s, flush=True))
    bad = [r for r in res if r[2]]
    print("SOAK %s" % ("FAILED: %s" % bad if bad else "OK"))
    sys.exit(1 if bad else 0)
