#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X X-engine (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] -- 704 inputs (352 stands x 2 pol),
96 channels, 4+4-bit -> int32 correlator on one MI355X.  One *step* = one short integration
as lwa352-pipeline.py runs it: acc_len 2400 = 5 gulps of 480 samples fed through the C ABI
(xengXgpuKernel*, dump on the 5th), on synthetic F-engine voltages already resident in HBM
(a replay ring of pre-generated gulps; make_golden_inputs.py's generator, seed 0xdeadbeef).

N > 1: frequency channels shard embarrassingly, 96 channels per GPU (configs[2]); there is no data-path
collective -- the process group (gloo) is only used for the barriers and the max-over-ranks of the wall time.
Either torch.distributed.run starts one rank per GPU (RANK / WORLD_SIZE in the environment), or
`python bench.py --gpus N` on its own starts N fresh child processes itself, one per GPU, before anything has
touched the GPU (sharding.spawn_ranks), and relays rank 0's line.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NSTAND, NPOL, NCHAN, NTIME_GULP, ACC_LEN = 352, 2, 96, 480, 2400
NINPUT = NSTAND * NPOL
CMAC_PER_UNIT = NINPUT * (NINPUT + 1) // 2           # 248160
OPS_PER_UNIT = 8 * CMAC_PER_UNIT                      # 1 985 280 int8 ops
PEAK_CUS, PEAK_CLK_HZ = 256, 2.4e9                    # MI355X_MICROARCH.md chip-level parameters
PEAK_INT8_OPS = PEAK_CUS * 8192 * PEAK_CLK_HZ         # 5.03e15 dense int8 MFMA ops/s
HBM_PEAK_GBS = 8000.0


def cpu_baseline(budget_s=12.0, verify=None):
    """The CPU oracle (oracle/xeng_oracle.c, OpenMP over channels) on the host cores: a bounded
    sample of the same workload (whole 480-sample x 96-channel x 704-input gulps).  The oracle is the checker, never the
    thing shipped: its first five gulps are the replay ring's gulps 0..4, so the integration it forms on the way also
    verifies what the GPU produced in the config-5 leg (`verify`: arrays downloaded after that leg)."""
    from oracle import xeng_oracle as orc
    orc.build()
    rs = np.random.RandomState(0xdeadbeef)
    gulps = [rs.randint(0, 255, size=NTIME_GULP * NCHAN * NINPUT, dtype=np.uint8).reshape(NTIME_GULP, NCHAN, NSTAND, NPOL)
             for _ in range(ACC_LEN // NTIME_GULP)]
    # the checker first (untimed): the scalar oracle's integration of gulps 0..4
    first_int = None
    for g in gulps:
        first_int = orc.xgpu_correlate(g, NSTAND, NCHAN, first_int)
    # ... then the baseline: the same contraction as the host cores can do it (oracle/xeng_cpu_fast.c: -march=native, AVX-512 VNNI where
    # the host has it, built on THIS machine), held to the scalar oracle word for word on its first integration
    fl = orc.fast_lib()
    # how many threads: all the cores this process may run on -- unless fewer are faster, which is the case in a container whose CPU
    # quota is far below the cores it sees (the pool's one-GPU boxes: 256 visible, a share of about 16).  One gulp per candidate.
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else int(fl.fast_num_threads())
    cands = sorted({n for n in (ncpu, ncpu // 2, ncpu // 4, 64, 32, 16, 8) if 1 <= n <= ncpu}, reverse=True)
    trial = {}
    for n in cands:
        fl.fast_set_threads(n)
        tt = time.time()
        orc.xgpu_correlate_fast(gulps[0], NSTAND, NCHAN, None)
        trial[n] = time.time() - tt
    nthreads = min(trial, key=trial.get)
    fl.fast_set_threads(nthreads)
    acc = None
    fast_first = None
    t0 = time.time()
    ngulp = 0
    while True:
        acc = orc.xgpu_correlate_fast(gulps[ngulp % len(gulps)], NSTAND, NCHAN, acc)
        ngulp += 1
        if ngulp == len(gulps):
            fast_first = acc.copy()
        el = time.time() - t0
        if (el > budget_s and ngulp >= len(gulps)) or ngulp >= 4000:
            break
    fast_ok = bool(np.array_equal(fast_first, first_int))
    units = ngulp * NTIME_GULP * NCHAN
    vin = gulps[0]
    # for honesty: the reference's own golden loop, restated in numpy (make_golden_inputs.py:156-158:
    # per time sample an outer product x conj(x)^T over all 704 inputs, full square), one core, 2 spectra x 24 channels
    nsp, ncg = 2, 24
    re, im = orc.decode(vin[:nsp, :ncg])
    x = (re + 1j * im).reshape(nsp, ncg, NINPUT)
    t1 = time.time()
    g = np.zeros((ncg, NINPUT, NINPUT), dtype=complex)
    for t in range(nsp):
        g += x[t, :, :, None] * np.conj(x[t, :, None, :])
    el_np = time.time() - t1
    out = {"value": round(8 * NINPUT * units / el / 1e9, 4), "unit": "Gb/s",
           "cores": int(nthreads), "cores_visible": int(ncpu), "seconds_per_gulp_by_threads": {str(k): round(v, 3) for k, v in trial.items()}, "kind": "port",
           "kind_detail": "oracle/xeng_cpu_fast.c: the oracle's contraction written for this host (-march=native, %s, 4 x 32 register tiles, OpenMP over "
                          "channels), built on this machine and held to the scalar oracle (oracle/xeng_oracle.c) word for word; bifrost's own CPU "
                          "correlator is not in /root/reference (empty submodule), so this is a stated baseline, not the reference's"
                          % ("AVX-512 VNNI vpdpwssd", "AVX-512 vpmaddwd", "no AVX-512: plain C")[2 - int(fl.fast_isa())],
           "equals_scalar_oracle": fast_ok,
           "cmac_per_s": units * CMAC_PER_UNIT / el,
           "sample": "%d gulps of %d samples x %d chan x %d inputs (%.1f s)" % (ngulp, NTIME_GULP, NCHAN, NINPUT, el),
           "reference_numpy_golden_loop": {"cmac_per_s": round(nsp * ncg * NINPUT * NINPUT / el_np, 1), "cores": 1,
                                           "sample": "%d spectra x %d chan, full-square outer products (%.1f s)" % (nsp, ncg, el_np)}}
    if verify is not None:
        out["config5_check"] = config5_check(verify, first_int, gulps)
    return out


def config5_check(verify, first_int, gulps):
    """What the config-5 pattern left on the GPU, held to the oracle: `first_int` = the oracle's integration of `gulps`
    (replay-ring gulps 0..4 of this rank's generator); `verify` = arrays downloaded after three such integrations."""
    from oracle import xeng_oracle as orc
    chk = {"visibilities_bit_exact": bool(np.array_equal(verify["vis"], first_int)),
           "corracc_sum_bit_exact": bool(verify.get("corracc") is None or np.array_equal(verify["corracc"], 3 * first_int.astype(np.int64))),
           "corracc_fused_in_dump_bit_exact": bool(verify.get("corracc_fused") is None or np.array_equal(verify["corracc_fused"], 3 * first_int.astype(np.int64))),
           "corracc_grouped_bit_exact": bool(verify.get("corracc_grouped") is None or np.array_equal(verify["corracc_grouped"], 3 * first_int.astype(np.int64)))}
    nt_b, nb = verify["beams"].shape[2], verify["beams"].shape[1]
    v2 = np.concatenate([gulps[0], gulps[1]]).reshape(nt_b, NCHAN, NINPUT)
    exp = orc.beamform(v2, verify["weights"].reshape(NCHAN, nb, NINPUT), nt_b, NCHAN, NINPUT, nb)
    err = float(np.max(np.abs(verify["beams"].astype(np.complex128) - exp)) / np.sqrt(np.mean(np.abs(exp) ** 2)))
    chk["beams_max_err_over_rms"] = err
    chk["beams_within_1e-5"] = bool(err <= 1e-5)
    pexp = orc.beamform_integrate(verify["beams"], verify["ntime_sum"])
    chk["power_beams_max_err_over_max"] = float(np.max(np.abs(verify["power"] - pexp)) / np.abs(pexp).max())
    chk["power_beams_ok"] = bool(chk["power_beams_max_err_over_max"] <= 1e-5)
    chk["ok"] = bool(chk["visibilities_bit_exact"] and chk["corracc_sum_bit_exact"] and chk["corracc_fused_in_dump_bit_exact"] and
                     chk["corracc_grouped_bit_exact"] and chk["beams_within_1e-5"] and chk["power_beams_ok"])
    return chk


def corr_block_leg(ffi, ring, gulp_bytes, ring_gulps, gpu, nint=400, nwarm=600):
    # (nwarm: the output ring grows its pool of 191 MB span buffers during the first few hundred integrations -- every allocation is a
    # 3 ms hole in the kernel trace, profiles/r05/trace_corr_block.txt; with 100 warm-up integrations they fell into the timed window and
    # the leg read 10 % above the C-ABI loop, which it equals in steady state)
    """Corr.main on in-repo 'cuda' rings at config-2 size.  The source publishes the replay ring's gulps as spans without
    copying them (WriteSequence.commit_external); the sink discards the visibility spans."""
    import json as _json
    import logging
    import threading
    from caltech_bifrost_dsp_amd.blocks import Corr
    from caltech_bifrost_dsp_amd.ndarray import XArray
    from caltech_bifrost_dsp_amd.ring import Ring
    gulps_per_step = ACC_LEN // NTIME_GULP
    r0, r1 = Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda")
    r0.resize(gulp_bytes, total_span=2 * gulps_per_step * gulp_bytes)
    # (every user of a ring declares the library streams it puts work on -- the harness's source and sink enqueue nothing -- or
    # the ring's stamps wait for all streams: include/xeng.h xengRingDeclareStreams)
    r0.declare_streams()
    r1.declare_streams()
    blk = Corr(logging.getLogger("bench-corr"), r0, r1, ntime_gulp=NTIME_GULP, nchan=NCHAN, npol=NPOL, nstand=NSTAND,
               acc_len=ACC_LEN, autostartat=0, gpu=gpu)
    hdr = {'nchan': NCHAN, 'chan0': 0, 'bw_hz': NCHAN * 23925.78125, 'fs_hz': 196000000, 'sfreq': 0.0, 'nstand': NSTAND, 'npol': NPOL,
           'seq0': 0, 'sync_time': 0, 'pipeline_id': 0, 'system_nchan': 32 * NCHAN}
    spans = [XArray(shape=(gulp_bytes,), dtype=np.uint8, space="cuda", _ptr=ring.ptr + g * gulp_bytes, _base=ring) for g in range(ring_gulps)]
    stamps = []

    def source():
        import time as _t
        t0 = _t.time()
        while len(r0._readers) < 1 and _t.time() - t0 < 10:
            _t.sleep(0.002)
        with r0.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header=_json.dumps(hdr), nringlet=1) as oseq:
                for k in range((nwarm + nint) * gulps_per_step):
                    oseq.commit_external(spans[k % ring_gulps])

    gen = r1.read(guarantee=True)

    def sink():
        for iseq in gen:
            for ispan in iseq.read(blk.ogulp_size):
                stamps.append(time.perf_counter())

    ths = [threading.Thread(target=f, daemon=True) for f in (sink, blk.main, source)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    n = len(stamps)
    rate = 8 * NINPUT * ACC_LEN * NCHAN * (n - 1 - nwarm) / (stamps[-1] - stamps[nwarm]) / 1e9 if n > nwarm + 1 else 0.0
    return {"value": round(rate, 1), "unit": "Gb/s", "integrations": n,
            "ms_per_integration": round((stamps[-1] - stamps[nwarm]) / max(n - 1 - nwarm, 1) * 1e3, 4) if n > nwarm + 1 else None,
            "block_gbps_stat": round(float(blk.stats.get('throughput', 0.0)), 1),
            "note": "Corr.main (one Python thread, streaming commits: span n closes after dump n+1 is enqueued) on in-repo device "
                    "rings, zero-copy replay source; wall rate between the spans arriving at a sink, and the block's own "
                    "`throughput` stat of its last integration (corr_block.py:453 formula)"}


def config5_blocks_leg(ffi, ring, gulp_bytes, ring_gulps, gpu, nint=400, nwarm=400, long_len=50, from_slabs=False, in_ring_integrations=4, chan0=0, seed=7,
                       keep_span=None):
    """BASELINE config 5 through the BLOCKS on one GPU: Corr -> CorrAcc and Beamform -> BeamformSumBeams as four Python
    threads on in-repo rings (gpu-input read in place by Corr and Beamform), fed by a zero-copy replay source.  CorrAcc's
    long accumulation (`long_len` dumps) is done by the dumps' own epilogue (fused mode, blocks/corr_acc_block.py).  The warm-up
    covers the start of the pipeline: the first long integrations allocate the 383 MB pinned-host spans of the slow ring
    (hipHostMalloc: ~0.1 s each, device-wide synchronisations), which the ring then recycles.
    from_slabs: the input ring holds PACKET SLABS (5280 SNAP2 packets per 480-sample gulp, header layout 'snap2_slab') as a
    receiver -- or Snap2Ingest(unpack=False) -- leaves them; Corr and Beamform hand them to the library's slab calls, which
    read them in place.  The source is a receiver's set of 64 slab buffers reused round robin, each re-stamped with its window's
    sequence numbers before it goes out."""
    import json as _json
    import logging
    import threading
    from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, Corr, CorrAcc
    from caltech_bifrost_dsp_amd.ndarray import XArray
    from caltech_bifrost_dsp_amd.ring import Ring
    gulps_per_step = ACC_LEN // NTIME_GULP
    nbeam, ns = 32, 24
    if from_slabs:
        import struct
        nblk, stride = NINPUT // 64, 32 + NCHAN * 64
        npk = NTIME_GULP * nblk
        slab = np.zeros((npk, stride), dtype=np.uint8)
        for t in range(NTIME_GULP):
            for b in range(nblk):
                slab[t * nblk + b, 8:32] = np.frombuffer(struct.pack(">LHHHHLLL", 0, 64, NINPUT, NCHAN, NCHAN, 0, 0, b * 64), dtype=np.uint8)
        slab[:, 32:] = np.random.RandomState(5).randint(0, 256, size=(npk, stride - 32), dtype=np.uint8)
        # a receiver's set of slab buffers, reused round robin: window k lands in buffer k mod nslabs, its headers stamped with the
        # window's sequence numbers just before it is handed on (xengSnap2StampSeq).  nslabs = the input ring (20 slabs) + what Corr
        # (10) and Beamform (18) keep in flight + 16: nobody still reads a buffer when its turn comes again.
        nslabs = in_ring_integrations * gulps_per_step + 44
        slab_pool = ffi.DeviceBuffer(nslabs * slab.nbytes)
        for k in range(nslabs):
            slab_pool.upload(slab, offset=k * slab.nbytes)
        spans = [XArray(shape=(slab.nbytes,), dtype=np.uint8, space="cuda", _ptr=slab_pool.ptr + k * slab.nbytes, _base=slab_pool) for k in range(nslabs)]
        gulp_bytes, ring_gulps = slab.nbytes, nslabs
    r_in = Ring("gpu-input", space="cuda")
    r_vis, r_slow = Ring("corr-output", space="cuda"), Ring("corr-slow-output", space="cuda_host")
    r_bf, r_pow = Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    # (input ring of four integrations: Beamform alone keeps up to five 960-sample gulps = ten input spans referenced while their
    # kernels are in flight, which was the whole of round 3's two-integration ring -- Corr then waited for gulps, not for the GPU;
    # the spans are windows on the replay buffer, the depth costs no memory)
    r_in.resize(gulp_bytes, total_span=in_ring_integrations * gulps_per_step * gulp_bytes)
    r_in.declare_streams()          # (the harness's source: commits windows on the replay buffer, enqueues nothing; the sinks declare in drain())
    log = logging.getLogger("bench-config5")
    corr = Corr(log, r_in, r_vis, ntime_gulp=NTIME_GULP, nchan=NCHAN, npol=NPOL, nstand=NSTAND, acc_len=ACC_LEN, autostartat=0, gpu=gpu)
    cacc = CorrAcc(log, r_vis, r_slow, nchan=NCHAN, npol=NPOL, nstand=NSTAND, acc_len=long_len * ACC_LEN, autostartat=0, gpu=gpu)
    # Beamform on GPU_NGULP = 2 input gulps per call, as the reference runs it (lwa352-pipeline.py:172,279-282): the two spans
    # are taken as the two windows of one 960-sample gulp (ring read_parts + xengBeamformRunParts), one launch, no copy
    nt_b = 2 * NTIME_GULP
    bf = Beamform(log, r_in, r_bf, nchan=NCHAN, nbeam=nbeam, ninput=NINPUT, ntime_gulp=nt_b, gpu=gpu)
    sb = BeamformSumBeams(log, r_bf, r_pow, nchan=NCHAN, ntime_gulp=nt_b, ntime_sum=ns, gpu=gpu)
    if os.environ.get("XENG_BENCH_VIS_SPANS"):       # (diagnosis: a deeper corr-output ring)
        r_vis.resize(corr.ogulp_size, total_span=int(os.environ["XENG_BENCH_VIS_SPANS"]) * corr.ogulp_size)
    rng = np.random.default_rng(seed)
    bf.gains_cpu[...] = (rng.uniform(-17, 17, bf.gains_cpu.shape) + 1j * rng.uniform(-17, 17, bf.gains_cpu.shape)).astype(np.complex64)
    hdr = {'nchan': NCHAN, 'chan0': chan0, 'bw_hz': NCHAN * 23925.78125, 'fs_hz': 196000000, 'sfreq': chan0 * 23925.78125, 'nstand': NSTAND, 'npol': NPOL,
           'seq0': 0, 'sync_time': 0, 'pipeline_id': 0, 'system_nchan': 32 * NCHAN}
    if from_slabs:
        hdr.update({'layout': 'snap2_slab', 'slab_ntime': NTIME_GULP, 'npkt_per_gulp': npk, 'pkt_stride': stride})
    else:
        spans = [XArray(shape=(gulp_bytes,), dtype=np.uint8, space="cuda", _ptr=ring.ptr + g * gulp_bytes, _base=ring) for g in range(ring_gulps)]
    stamps, nslow = [], [0]

    # The source and the three sinks are the harness, not the product: on the native ring they run as loops inside the
    # extension with the interpreter lock released (`_xfast.ring_feed_external` / `ring_drain`), so that the lock is shared by
    # the four block threads (+ CorrAcc's publish helper) only, as in a pipeline whose neighbours are not Python loops.
    native_harness = hasattr(r_in, "_h") and hasattr(getattr(r_in, "_x", None), "bench") and not os.environ.get("XENG_BENCH_PY_HARNESS")

    def source():
        import time as _t
        t0 = _t.time()
        while len(r_in._readers) < 2 and _t.time() - t0 < 10:
            _t.sleep(0.002)
        with r_in.begin_writing() as w:
            with w.begin_sequence(time_tag=0, header=_json.dumps(hdr), nringlet=1) as oseq:
                total = (nwarm + nint) * gulps_per_step
                if native_harness:
                    ptrs = np.array([sp.ptr for sp in spans], dtype=np.uint64).tobytes()
                    if from_slabs:
                        r_in._x.bench.ring_feed_slabs(r_in._h, oseq._seq_id, ptrs, gulp_bytes, total, npk, stride, nblk, NTIME_GULP)
                    else:
                        r_in._x.bench.ring_feed_external(r_in._h, oseq._seq_id, ptrs, gulp_bytes, total)
                else:
                    for k in range(total):
                        if from_slabs:
                            ffi.call("xengSnap2StampSeq", spans[k % ring_gulps].ptr, npk, stride, k * NTIME_GULP, nblk)
                        oseq.commit_external(spans[k % ring_gulps])

    kept = {}

    def drain(rg, gulp, on_span=None, times=None, keep=None):
        rg.declare_streams()        # (a sink that only counts spans)
        if keep is not None:        # (the N-rank leg: span number `keep` is copied to the host for the check against the oracle)
            gen_k = rg.read(guarantee=True)

            def run_keep():
                n = 0
                for iseq in gen_k:
                    for ispan in iseq.read(gulp):
                        if times is not None:
                            times.append(time.perf_counter())
                        if n == keep:
                            kept['vis'] = ispan.data.numpy().view(np.int32).copy()
                        n += 1
            return threading.Thread(target=run_keep, daemon=True)
        if native_harness:
            import ctypes as _ct
            rid = _ct.c_int()
            ffi.check("xengRingOpenReader", rg._enq.xengRingOpenReader(rg._h, 1, _ct.byref(rid)))       # (registered now, before any thread starts)

            def run_native():
                nsp, tt = rg._x.bench.ring_drain(rg._h, rid.value, gulp, times is not None)
                if times is not None:
                    times.extend(tt)
                if on_span:
                    for _ in range(nsp):
                        on_span()
            return threading.Thread(target=run_native, daemon=True)
        gen = rg.read(guarantee=True)

        def run():
            for iseq in gen:
                for ispan in iseq.read(gulp):
                    if times is not None:
                        times.append(time.perf_counter())
                    if on_span:
                        on_span()
        return threading.Thread(target=run, daemon=True)

    def slow_span():
        nslow[0] += 1
    ths = [drain(r_vis, corr.ogulp_size, times=stamps, keep=keep_span), drain(r_slow, cacc.ogulp_size, slow_span),
           drain(r_pow, (nbeam // 2) * (nt_b // ns) * NCHAN * 16)]
    ths += [threading.Thread(target=f, daemon=True) for f in (corr.main, cacc.main, bf.main, sb.main, source)]
    # Four block threads (+ the harness threads on the Python ring) under one interpreter lock.  The blocks keep the lock across their enqueue-only library calls and ask
    # before they wait (ffi.enqueue_lib, backend.beam_wait / xgpu_sync_lag), so a thread gives the lock up only when it really
    # has to sleep; the interpreter's switch interval stays at its default (a short one, 5e-5 s, measured 0.51-1.25 ms per
    # integration over six runs against 0.51-0.62 for the default: profiles/r03/blocks_lock_handoff.txt).
    import gc
    gc_mode = os.environ.get("XENG_BENCH_GC", "")           # (diagnosis: "freeze" / "off" for the duration of the leg)
    if gc_mode == "freeze":
        gc.collect()
        gc.freeze()
    elif gc_mode == "off":
        gc.disable()
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    if gc_mode == "freeze":
        gc.unfreeze()
    elif gc_mode == "off":
        gc.enable()
    n = len(stamps)
    ok = n > nwarm + 1
    el = (stamps[-1] - stamps[nwarm]) if ok else 0.0
    if from_slabs:
        import ctypes as _ct
        nfx, nfb = _ct.c_int(-1), _ct.c_int(-1)
        ffi.call("xengXgpuGetSlabFallbacks", _ct.byref(nfx))
        ffi.call("xengBeamformGetSlabFallbacks", _ct.byref(nfb))
        del spans
        slab_pool.free()
        return {"value": round(8 * NINPUT * ACC_LEN * NCHAN * (n - 1 - nwarm) / el / 1e9, 1) if ok else 0.0, "unit": "Gb/s",
                "ms_per_integration": round(el / max(n - 1 - nwarm, 1) * 1e3, 4) if ok else None, "integrations": n,
                "window_ms": [round((stamps[nwarm + (k + 1) * ((n - 1 - nwarm) // 4)] - stamps[nwarm + k * ((n - 1 - nwarm) // 4)]) / ((n - 1 - nwarm) // 4) * 1e3, 4)
                              for k in range(4)] if ok and (n - 1 - nwarm) >= 4 else [],
                "slabs_scattered_after_all": {"corr": int(nfx.value), "beamform": int(nfb.value)},
                "corracc_mode": "fused" if cacc.stats.get('fused') else ("grouped, %d dumps per pass" % cacc.group_dumps) if cacc.stats.get('grouped') else "map",
                "ring_allocations": {r.name: {k: int(v) for k, v in dict(r.counters).items() if k in ("alloc", "free", "reuse", "stamp_wait")}
                                     for r in (r_vis, r_slow, r_bf, r_pow)},
                "note": "config5_blocks with an input ring of packet slabs (a set of %d slab buffers of 5280 SNAP2 packets, %.1f GB, reused round robin and "
                        "re-stamped per window): the blocks hand the slabs to xengXgpuKernelAsyncSlab / xengBeamformRunSlabs, which read them in place" % (nslabs, nslabs * slab.nbytes / 1e9)}
    # (the leg in four windows: the Python threads settle into an interleaving, and not always into the same one)
    wins = []
    if ok:
        q = (n - 1 - nwarm) // 4
        wins = [round((stamps[nwarm + (k + 1) * q] - stamps[nwarm + k * q]) / q * 1e3, 4) for k in range(4)] if q > 0 else []
    return {"value": round(8 * NINPUT * ACC_LEN * NCHAN * (n - 1 - nwarm) / el / 1e9, 1) if ok else 0.0, "unit": "Gb/s",
            **({"kept": kept} if keep_span is not None else {}),        # (the N-rank leg's span for the oracle: popped before anything is printed)
            "seconds": round(el, 6), "timed_integrations": max(n - 1 - nwarm, 0),
            "ms_per_integration": round(el / max(n - 1 - nwarm, 1) * 1e3, 4) if ok else None, "integrations": n, "window_ms": wins,
            "long_integrations_published": nslow[0], "corracc_mode": "fused" if cacc.stats.get('fused') else ("grouped, %d dumps per pass" % cacc.group_dumps) if cacc.stats.get('grouped') else "map",
            # span allocations per ring over the whole leg: made, really freed, reissued from the free list, waits for a stamp at reissue
            "ring_allocations": {r.name: {k: int(v) for k, v in dict(r.counters).items() if k in ("alloc", "free", "reuse", "stamp_wait")}
                                 for r in (r_vis, r_slow, r_bf, r_pow)},
            "note": "config 5 through the blocks on one GPU: Corr -> CorrAcc (%d dumps per long integration, the spans of every ten summed in one "
                    "pass; published to a pinned-host ring) and Beamform (960-sample gulps = two input spans per call) -> BeamformSumBeams, four "
                    "Python threads on in-repo rings, zero-copy replay source; wall rate between visibility spans at a sink" % long_len}


def beam_weights(chan0, seed):
    """Weights as the Beamform block builds them (beamform_block.py:343-350) from random delays in [0,12) ns, amplitudes in
    [10,17) and calibration gains as beamformer_test.py:131-139, for the channels of one shard."""
    NB = 32
    rng = np.random.default_rng(seed)
    freqs = 50e6 + (chan0 + np.arange(NCHAN)) * 23925.78125
    wts = np.zeros((NCHAN, NB, NINPUT), np.complex64)
    for b_ in range(NB):
        delays_ns, amps = rng.uniform(0, 12, NINPUT), rng.uniform(10, 17, NINPUT)
        cal = (rng.uniform(-1, 1, (NCHAN, NINPUT)) + 1j * rng.uniform(-1, 1, (NCHAN, NINPUT))).astype(np.complex64)
        wts[:, b_, :] = amps * np.exp(1j * 2 * np.pi * freqs[:, None] * delays_ns * 1e-9) * cal
    return np.ascontiguousarray(wts.reshape(-1))


def config5_blocks_workload(args, ffi, dist, rank, world, gpu, ring, gulp_bytes, pin, info, sh):
    """`--workload config5_blocks`: config 5 through the BLOCKS on every rank -- the part of the pipeline with host threads, pinned-host
    publishes and an interpreter, i.e. the part that could fail to scale (lwa352-start-pipeline.sh:1-8 runs four such pipeline
    processes per server on pinned cores).  Every rank runs Corr -> CorrAcc and Beamform -> BeamformSumBeams on its own GPU, its own
    96 channels (header chan0 / sfreq of channel block `rank`, input seed 0xdeadbeef + rank, gain seed 7 + rank); one step = one
    integration of every rank, timed between visibility spans at a sink after `warmup` integrations; value = ingest of all ranks /
    max-over-ranks time.  Outside the timed region every rank holds one visibility span to the oracle."""
    if dist is not None:
        dist.barrier()
    leg = config5_blocks_leg(ffi, ring, gulp_bytes, args.ring_gulps, gpu, nint=args.steps, nwarm=max(args.warmup, 20), chan0=NCHAN * rank, seed=7 + rank,
                             keep_span=2 * (args.ring_gulps // (ACC_LEN // NTIME_GULP)))
    ffi.call("xengDeviceSynchronize")
    el = leg["seconds"] * args.steps / max(leg["timed_integrations"], 1)          # (time of args.steps integrations at the measured rate)
    ok = None
    if not args.no_cpu_baseline:
        from oracle import xeng_oracle as orc
        orc.build()
        gps = ACC_LEN // NTIME_GULP
        rs5 = np.random.RandomState(0xdeadbeef + rank)                              # (the replay ring's gulps 0 .. gps-1 of this rank: main())
        acc = None
        for g in range(gps):
            blk = rs5.randint(0, 255, size=gulp_bytes, dtype=np.uint8) if args.data == "random" else None
            if blk is None:
                break
            acc = orc.xgpu_correlate(blk.reshape(NTIME_GULP, NCHAN, NSTAND, NPOL), NSTAND, NCHAN, acc)
        vis = leg.pop("kept").get("vis")
        ok = {"visibilities_bit_exact": bool(acc is not None and vis is not None and np.array_equal(vis.reshape(-1), acc.reshape(-1))),
              "span": "integration %d of the stream = replay gulps 0..%d" % (2 * (args.ring_gulps // gps), gps - 1)}
        ok["ok"] = ok["visibilities_bit_exact"]
    leg.pop("kept", None)
    per_rank_ms = [round(v / args.steps * 1e3, 4) for v in sh.gather_over_ranks(dist, el)]
    oks, wins = [ok], [leg["window_ms"]]
    if dist is not None:
        import torch
        oks, wins = [None] * world, [None] * world
        dist.all_gather_object(oks, ok)
        dist.all_gather_object(wins, leg["window_ms"])
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        dist.barrier()
    units = ACC_LEN * NCHAN * args.steps * world
    gbps = 8 * NINPUT * units / el / 1e9
    return {
        "metric": "xengine_ingest_gbps_704in_96ch", "value": round(gbps, 2), "unit": "Gb/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 (4+4-bit samples) -> int32; beams fp32", "data": "synthetic",
        "config": {"workload": "config 5 through the BLOCKS on every GPU: Corr -> CorrAcc (%s) and Beamform (960-sample gulps) -> BeamformSumBeams, four block threads per "
                               "rank on in-repo rings, zero-copy replay source; 704 inputs, %d chan/GPU" % (leg.get("corracc_mode"), NCHAN),
                   "nchan_total": NCHAN * world, "chan0_per_rank": [NCHAN * r for r in range(world)], "sharding": "channels, %d per GPU, no collective" % NCHAN},
        "per_rank_ms": per_rank_ms, "per_rank_ms_min": min(per_rank_ms), "per_rank_ms_max": max(per_rank_ms), "per_rank_window_ms": wins,
        "cmac_per_s": CMAC_PER_UNIT * units / el, "design_rate_x": round(gbps / world / 12.94, 1),
        "roofline": {"kernel": "xcorr_fused_kernel", "bound": "mfma", "unit": "TFLOP/s",
                     "achieved": round(OPS_PER_UNIT * ACC_LEN * NCHAN / (el / args.steps) / 1e12, 1), "peak": round(PEAK_INT8_OPS / 1e12, 1),
                     "frac": round(OPS_PER_UNIT * ACC_LEN * NCHAN / (el / args.steps) / PEAK_INT8_OPS, 4), "traffic": None,
                     "note": "the contraction's algorithmic int8 ops per integration / time per integration of the slowest rank, while the beamformer chain, CorrAcc "
                             "and the host side of four blocks share the GPU and the interpreter (config 5 through the blocks: not the contraction alone)"},
        "verified": {"ok": all(bool(o and o["ok"]) for o in oks) if oks[0] is not None else None, "per_rank": oks},
        "rank_placement": {"numa_node": pin["numa_node"], "ncpus": len(pin["cpus"]), "source": pin["source"]},
        "device": info,
    }


def config5_workload(args, ffi, dist, rank, world, gpu, ring, gulp_bytes, pin, info, sh):
    """`--workload config5`: BASELINE config 5 as N ranks run it -- "Full X-engine: corner-turn + Corr + CorrAcc long-accum +
    Beamform concurrent on HIP streams, 704 inputs, 768 chan across 8 GPUs" (lwa352-start-pipeline.sh:1-8: one pipeline
    process per channel block).  Every rank runs, on its own GPU and its own 96 channels (chan0 = 96 * rank, input seed
    0xdeadbeef + rank): per 2400-sample integration five gulps registered in place + one contraction (X-engine streams), 2.5
    beamformer gulps of 960 samples + their power sums (beam stream), and CorrAcc's long accumulation as one pass over the
    spans of every ten dumps (map stream).  No collective: the process group carries the barriers and the max over ranks.  One step = one
    integration of every rank; value = ingest of all ranks / max-over-ranks time.  Outside the timed region every rank
    holds one dumped span, the accumulator sum, one beam gulp and its power sums to the oracle."""
    L = ffi.lib()
    gulps_per_step = ACC_LEN // NTIME_GULP
    matlen = NCHAN * 249216
    NT_B, NB, NS = 960, 32, 24
    chan0 = NCHAN * rank
    ffi.call("xengBeamformInitialize", gpu, NINPUT, NCHAN, NT_B, NB, 0)
    wts = beam_weights(chan0, 0xaabbccdd + rank)
    dw = ffi.DeviceBuffer(wts.nbytes).upload(wts)
    dbeam = ffi.DeviceBuffer(NCHAN * NB * NT_B * 8)
    dpow = ffi.DeviceBuffer((NB // 2) * (NT_B // NS) * NCHAN * 16)
    KG = 10                               # dumps per group of CorrAcc's long accumulation (one xengMapSumI32 pass per group)
    outs_g = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(KG + 3)]
    acc_long = ffi.DeviceBuffer(2 * matlen * 4)
    kfn = L.xengXgpuKernelAsync
    SrcArr = ctypes.c_void_p * KG
    gi, bi = [0], [0]

    def bstep(i):
        src = ring.ptr + ((2 * i) % (args.ring_gulps - 1)) * gulp_bytes          # two consecutive 480-sample gulps = one 960-sample beam gulp
        ffi.check("run", L.xengBeamformRunVersioned(src, dbeam.ptr, dw.ptr, 1))
        ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))

    def step(n):
        o = outs_g[n % len(outs_g)]
        for g in range(gulps_per_step):
            ffi.check("kernel", kfn(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, o.ptr, int(g == gulps_per_step - 1)))
            gi[0] += 1
        for _ in range(2 + (n & 1)):
            bstep(bi[0])
            bi[0] += 1
        ffi.call("xengXgpuSyncLag", 1)              # dump n-1 is complete
        if n >= KG and n % KG == 0:                 # ... and with it the group n-K .. n-1: one pass over its spans (map stream)
            srcs = SrcArr(*[outs_g[(n - KG + j) % len(outs_g)].ptr for j in range(KG)])
            ffi.check("sum", L.xengMapSumI32(acc_long.ptr, srcs, KG, 2 * matlen, int(n > KG)))
        elif n % KG == 1:
            ffi.call("xengMapSync")                 # (read before dump n + 2 writes the first of those spans again)

    def barrier():
        ffi.call("xengDeviceSynchronize")
        if dist is not None:
            dist.barrier()

    n = 0
    for _ in range(min(args.prewarm, 300) + args.warmup):
        step(n)
        n += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(n)
        n += 1
    ffi.call("xengDeviceSynchronize")
    el = time.perf_counter() - t0
    # ---- outside the timed region: the same pattern on known inputs, held to the oracle on every rank
    ok = None
    if not args.no_cpu_baseline:
        from oracle import xeng_oracle as orc
        orc.build()
        rs5 = np.random.RandomState(0xdeadbeef + rank)
        gulps = [rs5.randint(0, 255, size=gulp_bytes, dtype=np.uint8) for _ in range(gulps_per_step)]
        for g in range(gulps_per_step):
            ring.upload(gulps[g], offset=g * gulp_bytes)
        ffi.call("xengXgpuSync")
        ffi.call("xengMapSync")
        for k in range(3):
            for g in range(gulps_per_step):
                ffi.check("kernel", kfn(ring.ptr + g * gulp_bytes, outs_g[k].ptr, int(g == gulps_per_step - 1)))
            ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dbeam.ptr, dw.ptr, 1))
            ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
            ffi.call("xengXgpuSyncLag", 1)
        ffi.call("xengXgpuSync")
        g3 = (ctypes.c_void_p * 3)(*[outs_g[k].ptr for k in range(3)])
        ffi.check("sum", L.xengMapSumI32(acc_long.ptr, g3, 3, 2 * matlen, 0))
        ffi.call("xengMapSync")
        ffi.call("xengBeamformSync")
        verify = {"vis": outs_g[2].download(np.int32), "corracc": None, "corracc_fused": None,
                  "corracc_grouped": acc_long.download(np.int32).astype(np.int64),
                  "beams": dbeam.download(np.complex64).reshape(NCHAN, NB, NT_B), "weights": wts, "ntime_sum": NS,
                  "power": dpow.download(np.float32).reshape(NB // 2, NT_B // NS, NCHAN, 4)}
        g4 = [g.reshape(NTIME_GULP, NCHAN, NSTAND, NPOL) for g in gulps]
        acc = None
        for g in g4:
            acc = orc.xgpu_correlate(g, NSTAND, NCHAN, acc)
        ok = config5_check(verify, acc, g4)
    per_rank_ms = [round(v / args.steps * 1e3, 4) for v in sh.gather_over_ranks(dist, el)]
    oks = [ok]
    if dist is not None:
        import torch
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        dist.barrier()
    ffi.call("xengBeamformDestroy")
    units = ACC_LEN * NCHAN * args.steps * world
    gbps = 8 * NINPUT * units / el / 1e9
    return {
        "metric": "xengine_ingest_gbps_704in_96ch", "value": round(gbps, 2), "unit": "Gb/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 (4+4-bit samples) -> int32; beams fp32", "data": "synthetic",
        "config": {"workload": "config 5, full X-engine per GPU: Corr (5 x 480-sample gulps, fused corner turn) + CorrAcc (groups of 10 dumps summed in one pass) "
                               "+ Beamform (32 beams, 960-sample gulps) + power beams, concurrent on HIP streams; 704 inputs, %d chan/GPU" % NCHAN,
                   "nchan_total": NCHAN * world, "chan0_per_rank": [NCHAN * r for r in range(world)],
                   "sharding": "channels, %d per GPU, no collective" % NCHAN,
                   "input": "device-resident replay ring, %d gulps, seed 0xdeadbeef + rank" % args.ring_gulps},
        "per_rank_ms": per_rank_ms, "per_rank_ms_min": min(per_rank_ms), "per_rank_ms_max": max(per_rank_ms),
        "cmac_per_s": CMAC_PER_UNIT * units / el,
        "mfma_peak_frac_end_to_end": round(8 * CMAC_PER_UNIT * units / el / (PEAK_INT8_OPS * world), 4),
        "design_rate_x": round(gbps / world / 12.94, 1),
        "roofline": {"kernel": "xcorr_fused_kernel", "bound": "mfma", "unit": "TFLOP/s",
                     "achieved": round(OPS_PER_UNIT * ACC_LEN * NCHAN / (el / args.steps) / 1e12, 1), "peak": round(PEAK_INT8_OPS / 1e12, 1),
                     "frac": round(OPS_PER_UNIT * ACC_LEN * NCHAN / (el / args.steps) / PEAK_INT8_OPS, 4), "traffic": None,
                     "note": "the contraction's algorithmic int8 ops per integration / time per integration of the slowest rank, while the "
                             "beamformer, the power sums and CorrAcc's group sums share the GPU (config 5: not the contraction alone)"},
        "verified": {"ok": all(bool(o and o["ok"]) for o in oks) if oks[0] is not None else None, "per_rank": oks},
        "rank_placement": {"numa_node": pin["numa_node"], "ncpus": len(pin["cpus"]), "source": pin["source"]},
        "device": info,
    }


def _leg(name):
    """XENG_BENCH_TRACE=1: names each leg on stderr as it starts (a fault under a profiler then says where it was)"""
    if os.environ.get("XENG_BENCH_TRACE"):
        print("[bench] %s" % name, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="correlator", choices=["correlator", "config5", "config5_blocks"],
                    help="correlator: BASELINE config 2 / 3 (the headline: one contraction per step); config5: the full X-engine per "
                         "GPU (Corr + fused CorrAcc + Beamform + power beams concurrently), sharded like config 3")
    # steady state of the streaming pipeline is reached after a few hundred integrations (clock / power ramp of a
    # cold GPU, launch overlap pattern): see --prewarm.  The defaults time 2000 integrations (0.43 s)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm", type=int, default=1500,
                    help="integrations run before the --warmup steps (outside the timed region) to bring a cold GPU's "
                         "clocks / power state and the streaming launch pattern to their steady state, whatever "
                         "--steps/--warmup the caller picks (0.3 s)")
    ap.add_argument("--ring-gulps", type=int, default=10, help="device-resident replay ring depth (gulps)")
    ap.add_argument("--lag", type=int, default=1, choices=[1, 2, 3],
                    help="streaming depth: after enqueueing integration n wait for dump n-lag (lag+1 output spans)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pin", action="store_true", help="diagnostic: leave the process's CPU affinity alone (default: the cores next to the GPU)")
    ap.add_argument("--sustained", type=int, default=10000,
                    help="integrations of the untimed `sustained` leg (two runs of this many; 0 = skip)")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false", help="skip the PCIe-inclusive measurement")
    ap.add_argument("--no-beamform", dest="beamform", action="store_false", help="skip the config-4 beamformer leg")
    ap.add_argument("--no-blocks", dest="blocks", action="store_false", help="skip the legs that run the Python blocks (profiler passes: thousands of launches per leg)")
    ap.add_argument("--data", default="random", choices=["random", "zeros", "0x88", "gaussian"],
                    help="diagnostic only: constant inputs show the DVFS give-back (the reported value uses random)")
    ap.add_argument("--sync-per-call", action="store_true",
                    help="time the drop-in synchronous xengXgpuKernel (the reference's call semantics)")
    ap.add_argument("--rehearse-on-gpu0", action="store_true",
                    help="testing only: every rank uses GPU 0 (to rehearse the N>1 code path on a 1-GPU box)")
    ap.add_argument("--sync-per-integration", action="store_true",
                    help="enqueue the gulps of one integration, then wait for its dump before the next")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="testing only (no GPU): ranks skip the X-engine work and report a fixed fake time, so that the "
                         "N-rank launch, the process group, the max-over-ranks and the JSON line can be checked on CPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: be the launcher.  Nothing in this process has touched the GPU yet.
        import caltech_bifrost_dsp_amd  # noqa: F401
        from caltech_bifrost_dsp_amd import sharding
        rc, out0, errs = sharding.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(out0)
        if rc:
            for r, e in enumerate(errs):
                if e.strip():
                    sys.stderr.write("---- rank %d stderr ----\n%s\n" % (r, e))
        sys.exit(rc)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # plumbing only: barrier + max-reduce of the wall time
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on stdout: keep stdout for the one JSON line
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from caltech_bifrost_dsp_amd import sharding as _sh
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if args.selftest_spawn:
        import torch
        pin = _sh.pin_rank(local_rank, local_world)          # (no GPU here: every rank takes its share of the allowed CPUs)
        own = 1e-3 * (rank + 1)
        t = torch.tensor([own], dtype=torch.float64)
        if dist is not None:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
        el = float(t.item())
        per_rank = [round(v / args.steps * 1e3, 4) for v in _sh.gather_over_ranks(dist, own)]
        masks = [None] * world
        if dist is not None:
            dist.all_gather_object(masks, pin["cpus"])
        else:
            masks = [pin["cpus"]]
        if rank == 0:
            units = ACC_LEN * NCHAN * args.steps * world
            print(json.dumps({"metric": "xengine_ingest_gbps_704in_96ch", "value": round(8 * NINPUT * units / el / 1e9, 2),
                              "unit": "Gb/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                              "per_rank_ms": per_rank, "per_rank_ms_min": min(per_rank), "per_rank_ms_max": max(per_rank),
                              "rank_cpus": masks, "placement": pin["source"],
                              "vs_baseline": None, "data": "selftest (no GPU work; not a result)",
                              "config": {"workload": "selftest:%s" % args.workload, "nchan_total": NCHAN * world,
                                         "chan0_per_rank": [NCHAN * r for r in range(world)]}}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import caltech_bifrost_dsp_amd  # noqa: F401
    from caltech_bifrost_dsp_amd import ffi

    gpu = 0 if args.rehearse_on_gpu0 else local_rank
    ffi.call("xengSetDevice", gpu)
    info = ffi.device_info(gpu)
    # host placement: this rank (and every thread it starts) on the cores next to its GPU (sharding.pin_rank)
    host_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pin = (_sh.pin_rank(local_rank, local_world, ffi.device_pci_bus_id(gpu)) if not args.no_pin else
           {"source": "none (--no-pin)", "cpus": sorted(host_cpus or []), "numa_node": None})
    gulps_per_step = ACC_LEN // NTIME_GULP
    ffi.call("xengXgpuConfigure", NSTAND, NPOL, NCHAN, NTIME_GULP, gulps_per_step)
    ffi.call("xengXgpuInitialize", gpu)
    gulp_bytes = NTIME_GULP * NCHAN * NINPUT
    matlen = NCHAN * 249216

    # device-resident replay ring of synthetic F-engine voltages (chan0 = 96*rank; seed + rank)
    ring = ffi.DeviceBuffer(args.ring_gulps * gulp_bytes)
    rs = np.random.RandomState(0xdeadbeef + rank)
    for g in range(args.ring_gulps):
        if args.data == "random":
            blk = rs.randint(0, 255, size=gulp_bytes, dtype=np.uint8)
        elif args.data == "gaussian":
            # what an F-engine's 4-bit requantiser emits: rounded Gaussian, sigma 2.5 levels, clipped to -7..7
            q = np.clip(np.rint(rs.normal(0.0, 2.5, size=(2, gulp_bytes))), -7, 7).astype(np.int8)
            blk = (((q[0] & 0xF) << 4) | (q[1] & 0xF)).astype(np.uint8)
        else:
            blk = np.full(gulp_bytes, 0 if args.data == "zeros" else 0x88, dtype=np.uint8)
        ring.upload(blk, offset=g * gulp_bytes)
    if args.workload in ("config5", "config5_blocks"):
        res = (config5_workload if args.workload == "config5" else config5_blocks_workload)(args, ffi, dist, rank, world, gpu, ring, gulp_bytes, pin, info, _sh)
        if rank == 0:
            print(json.dumps(res))
        ffi.call("xengXgpuDestroy")
        if dist is not None:
            dist.destroy_process_group()
        return
    nout = args.lag + 1
    outs = [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(max(2, nout))]     # output spans rotate, as ring spans do
    kern = "xengXgpuKernel" if args.sync_per_call else "xengXgpuKernelAsync"
    L = ffi.lib()
    kfn = getattr(L, kern)
    if args.sync_per_call:
        call_mode = "sync-per-call (reference call semantics)"
    elif args.sync_per_integration:
        call_mode = "enqueue gulps, sync per integration"
    else:
        call_mode = "streaming: enqueue integration n, then wait for dump n-%d (xengXgpuSyncLag(%d)), %d output spans" % (args.lag, args.lag, nout)

    gi = [0]
    si = [0]
    units_per_step_c = ACC_LEN * NCHAN

    def step():
        out = outs[si[0] % nout]
        si[0] += 1
        for g in range(gulps_per_step):
            rc = kfn(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, out.ptr, int(g == gulps_per_step - 1))
            if rc:
                ffi.check(kern, rc)
            gi[0] += 1
        if args.sync_per_call:
            return
        # the span of dump n-1 (or n) is complete before it would be committed downstream
        rc = L.xengXgpuSync() if args.sync_per_integration else L.xengXgpuSyncLag(args.lag)
        if rc:
            ffi.check("xengXgpuSync", rc)

    def barrier():
        ffi.call("xengDeviceSynchronize")
        if dist is not None:
            dist.barrier()

    for _ in range(args.prewarm):
        step()
    for _ in range(args.warmup):
        step()
    ffi.call("xengXgpuSetProfiling", 1)
    tm = (ctypes.c_double * 2)()
    cn = (ctypes.c_int * 2)()
    ffi.call("xengXgpuGetTimes", tm, cn)       # clear
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ffi.call("xengDeviceSynchronize")
    el = time.perf_counter() - t0
    ffi.call("xengXgpuGetTimes", tm, cn)
    _leg('one-launch-at-a-time')
    # outside the timed region: the same kernels launched one integration at a time (no overlap between
    # launches), to document the stand-alone duration of each kernel next to the streaming one
    iso_tm = (ctypes.c_double * 2)()
    iso_cn = (ctypes.c_int * 2)()
    if not args.sync_per_call and not args.sync_per_integration:
        ffi.call("xengXgpuSync")
        ffi.call("xengXgpuGetTimes", iso_tm, iso_cn)    # clear
        for _ in range(20):
            for g in range(gulps_per_step):
                ffi.check(kern, kfn(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, outs[0].ptr, int(g == gulps_per_step - 1)))
                gi[0] += 1
            ffi.call("xengXgpuSync")
        ffi.call("xengXgpuGetTimes", iso_tm, iso_cn)
    ffi.call("xengXgpuSetProfiling", 0)
    _leg('sustained')
    # outside the timed region: the same streaming pattern SUSTAINED -- 10 000 integrations (> 2 s) in windows of 1000, on the
    # replay ring of the timed region and on one of 40 gulps (1.3 GB: past the 256 MB Infinity Cache, so every gulp comes
    # from HBM whatever the last-level cache holds)
    sustained = None
    if rank == 0 and world == 1 and not args.sync_per_call and not args.sync_per_integration and args.sustained > 0:
        def sustained_run(ring_ptr, ngulps, nsteps, window):
            k, marks = 0, []
            ffi.call("xengXgpuSync")
            for n in range(nsteps + 1):
                if n % window == 0:
                    ffi.call("xengXgpuSync") if n == 0 else None
                    marks.append(time.perf_counter())
                if n == nsteps:
                    break
                out = outs[n % nout]
                for g in range(gulps_per_step):
                    rc = kfn(ring_ptr + (k % ngulps) * gulp_bytes, out.ptr, int(g == gulps_per_step - 1))
                    if rc:
                        ffi.check(kern, rc)
                    k += 1
                rc = L.xengXgpuSyncLag(args.lag)
                if rc:
                    ffi.check("xengXgpuSyncLag", rc)
            ffi.call("xengXgpuSync")
            total = time.perf_counter() - marks[0]
            w = sorted((b - a) / window * 1e3 for a, b in zip(marks[:-1], marks[1:]))
            return {"integrations": nsteps, "seconds": round(total, 3), "ms_per_step": round(total / nsteps * 1e3, 4),
                    "window": window, "window_ms_min": round(w[0], 4), "window_ms_median": round(w[len(w) // 2], 4),
                    "window_ms_max": round(w[-1], 4), "gbps": round(8 * NINPUT * units_per_step_c * nsteps / total / 1e9, 1),
                    "ring_gulps": ngulps, "ring_mb": round(ngulps * gulp_bytes / 1e6, 1)}
        sustained = {"replay_ring": sustained_run(ring.ptr, args.ring_gulps, args.sustained, 1000)}
        big = ffi.DeviceBuffer(40 * gulp_bytes)
        for g in range(40):                       # distinct addresses are what matters here: device copies of the ring's gulps
            ffi.call("xengMemcpy", big.ptr + g * gulp_bytes, ring.ptr + (g % args.ring_gulps) * gulp_bytes, gulp_bytes)
        sustained["ring_40_gulps_past_infinity_cache"] = sustained_run(big.ptr, 40, args.sustained, 1000)
        big.free()
        sustained["note"] = ("outside the driver's timed region: the timed pattern (enqueue integration n, wait for dump n-%d) held for %d "
                             "integrations; per-1000-integration windows; the second run reads a 1.3 GB ring, larger than the "
                             "256 MB Infinity Cache" % (args.lag, args.sustained))
    _leg('pcie')
    # outside the timed region: PCIe-inclusive regime (SURVEY 8d "two reporting regimes", ii): gulps start in
    # pinned host memory, are copied H2D (xengMemcpy, the Copy block's copy_array) and then correlated
    verify = None
    pcie = None
    if args.h2d and rank == 0 and world == 1:
        nh = 2 * gulps_per_step
        hostbuf = ffi.DeviceBuffer(nh * gulp_bytes, ffi.SPACE_CUDA_HOST)
        hostbuf.as_host_array(np.uint8)[:] = np.random.RandomState(1).randint(0, 255, size=nh * gulp_bytes, dtype=np.uint8)
        ffi.call("xengXgpuSync")
        t1 = time.perf_counter()
        nint = 6
        for it in range(nint):
            for g in range(gulps_per_step):
                slot = (it * gulps_per_step + g) % nh
                ffi.call("xengMemcpy", ring.ptr + slot * gulp_bytes, hostbuf.ptr + slot * gulp_bytes, gulp_bytes)
                ffi.check(kern, kfn(ring.ptr + slot * gulp_bytes, outs[it & 1].ptr, int(g == gulps_per_step - 1)))
            ffi.call("xengXgpuSyncLag", 1)
        ffi.call("xengXgpuSync")
        el2 = time.perf_counter() - t1
        pcie = {"value": round(8 * NINPUT * units_per_step_c * nint / el2 / 1e9, 1), "unit": "Gb/s",
                "h2d_GBs": round(gulp_bytes * gulps_per_step * nint / el2 / 1e9, 1),
                "note": "pinned host -> H2D -> X-engine, %d integrations; link-bound (PCIe Gen5 x16)" % nint}
        hostbuf.free()
    _leg('packetize')
    # outside the timed region: the slow-visibility output step (SURVEY 8f rows 2+4): device reorder of one
    # integration into per-baseline packet payloads (CorrOutputFull), HBM-bound
    pktz = None
    if args.beamform and rank == 0 and world == 1:
        ffi.call("xengXgpuSync")
        a2i = np.arange(NINPUT, dtype=np.int32)
        nst = NINPUT // 2
        blm = np.zeros(nst * nst * 4, dtype=np.int32)
        cjm = np.zeros_like(blm)
        ffi.call("xengXgpuGetOrder", a2i.ctypes.data, blm.ctypes.data, cjm.ctypes.data)
        dbl, dcj = ffi.DeviceBuffer(blm.nbytes).upload(blm), ffi.DeviceBuffer(cjm.nbytes).upload(cjm)
        nbl = nst * (nst + 1) // 2
        dpay = ffi.DeviceBuffer(nbl * 4 * NCHAN * 8)
        for _ in range(3):
            ffi.call("xengXgpuPacketize", outs[0].ptr, dpay.ptr, dbl.ptr, dcj.ptr, 1)
        t1 = time.perf_counter()
        nrep = 20
        for _ in range(nrep):
            ffi.call("xengXgpuPacketize", outs[0].ptr, dpay.ptr, dbl.ptr, dcj.ptr, 1)
        pk_ms = (time.perf_counter() - t1) / nrep * 1e3
        pk_bytes = nbl * 4 * NCHAN * 8 * 2          # read every addressed word once, write every payload word once
        pktz = {"kernel": "packetize_kernel", "avg_us": round(pk_ms * 1e3, 1), "payload_bytes": nbl * 4 * NCHAN * 8,
                "roofline": {"bound": "hbm", "achieved": round(pk_bytes / (pk_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(pk_bytes / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "note": "xGPU-order int32 planes -> %d COR payloads [chan][pol][pol][2] in sending order; wall time of the "
                        "synchronous call (includes launch + stream sync)" % nbl}
        for b in (dbl, dcj, dpay):
            b.free()
    _leg('ingest')
    # outside the timed region: the ingest step: one config-2 gulp worth of SNAP2 packets
    # (5280 packets of 32 + 6144 bytes, device-resident) scattered into the gulp layout
    ingest = None
    if args.beamform and rank == 0 and world == 1:
        import struct
        nspp, ncb = 32, 1                                     # stands per packet, channel blocks
        npb = (NINPUT // 2) // nspp
        stride = 32 + NCHAN * nspp * 2
        npk = NTIME_GULP * ncb * npb
        slab = np.zeros((npk, stride), dtype=np.uint8)
        k = 0
        for t in range(NTIME_GULP):
            for pb in range(npb):
                slab[k, :32] = np.frombuffer(struct.pack(">QLHHHHLLL", t, 0, nspp * 2, NINPUT, NCHAN, NCHAN, 0, 0, pb * nspp * 2), dtype=np.uint8)
                k += 1
        slab[:, 32:] = np.random.RandomState(5).randint(0, 255, size=(npk, stride - 32), dtype=np.uint8)
        dslab = ffi.DeviceBuffer(slab.nbytes).upload(slab)
        dgulp = ffi.DeviceBuffer(gulp_bytes)
        placed = ctypes.c_int()
        for _ in range(3):
            ffi.call("xengSnap2Unpack", dslab.ptr, npk, stride, dgulp.ptr, 0, NTIME_GULP, 0, NCHAN, NINPUT, 1, ctypes.byref(placed), None)
        assert placed.value == npk
        t1 = time.perf_counter()
        nrep = 20
        for _ in range(nrep):
            ffi.call("xengSnap2Unpack", dslab.ptr, npk, stride, dgulp.ptr, 0, NTIME_GULP, 0, NCHAN, NINPUT, 1, None, None)
        in_ms = (time.perf_counter() - t1) / nrep * 1e3
        in_bytes = slab.nbytes + gulp_bytes                  # packets read, gulp written once (a complete slab needs no zero-fill)
        ingest = {"kernel": "snap2_unpack_kernel", "avg_us": round(in_ms * 1e3, 1), "packets": npk, "packet_bytes": stride,
                  "ingest_gbps": round(8 * gulp_bytes / (in_ms * 1e-3) / 1e9, 1),
                  "roofline": {"bound": "hbm", "achieved": round(in_bytes / (in_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(in_bytes / (in_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                  "note": "wall time of the synchronous call: one launch -- scatter kernel (device time: profiles/r03 kernel stats) whose last work-group checks the coverage and reports to pinned memory + "
                          "host poll of that word; no memset, copy or stream wait, no zero-fill pass for a complete slab"}
        # packets -> visibilities, device resident (BASELINE: "throughput on synthetic F-engine packets"): every gulp of
        # every integration is first scattered out of its packet slab (enqueue-only, on the X-engine's staging stream),
        # then registered with the X-engine; same streaming pattern as the timed region
        # (slab k carries the sequence numbers of gulp k: 480 k .. 480 k + 479)
        slabs = [dslab]
        hdr_seq = slab[:, :8].copy()
        for k in range(1, 2 * gulps_per_step):
            for t in range(NTIME_GULP):
                slab[t * npb:(t + 1) * npb, :8] = np.frombuffer(struct.pack(">Q", k * NTIME_GULP + t), dtype=np.uint8)
            slabs.append(ffi.DeviceBuffer(slab.nbytes).upload(slab))
        slab[:, :8] = hdr_seq
        # the same packets as a receiver that chooses where each packet lands can place them: payloads on 128-byte lines
        # (49 lines per packet; the header in the last 32 bytes of the line before)
        stride_a, lead_a = 49 * 128, 96
        slab_a = np.zeros(lead_a + npk * stride_a, dtype=np.uint8)
        slab_a[lead_a:].reshape(npk, stride_a)[:, :stride] = slab
        slabs_a = [ffi.DeviceBuffer(slab_a.nbytes).upload(slab_a)]
        for k in range(1, len(slabs)):
            slabs_a.append(ffi.DeviceBuffer(slab_a.nbytes))
            ffi.call("xengMemcpy", slabs_a[k].ptr, slabs_a[0].ptr, slab_a.nbytes)

        def packets_leg(direct, slabs=slabs, nrep=300, nwarm=100):
            kk = 0
            for it in range(nwarm + nrep):
                if it == nwarm:
                    ffi.call("xengXgpuSync")
                    t1 = time.perf_counter()
                for g in range(gulps_per_step):
                    slot = kk % (2 * gulps_per_step)
                    if direct == 2:
                        ffi.check("slab", L.xengXgpuKernelAsyncSlab(slabs_a[slot].ptr + lead_a, npk, stride_a, 0, 0, outs[it & 1].ptr, int(g == gulps_per_step - 1), None, 0))
                    elif direct:    # the slab IS the gulp: verified on the device, read in place by the contraction (no scatter pass)
                        ffi.check("slab", L.xengXgpuKernelAsyncSlab(slabs[slot].ptr, npk, stride, slot * NTIME_GULP, 0, outs[it & 1].ptr, int(g == gulps_per_step - 1), None, 0))
                    else:
                        dst = ring.ptr + slot * gulp_bytes
                        ffi.check("unpack", L.xengSnap2UnpackAsync(slabs[slot].ptr, npk, stride, dst, slot * NTIME_GULP, NTIME_GULP, 0, NCHAN, NINPUT, 1))
                        ffi.check(kern, L.xengXgpuKernelAsync(dst, outs[it & 1].ptr, int(g == gulps_per_step - 1)))
                    kk += 1
                ffi.call("xengXgpuSyncLag", 1)
            ffi.call("xengXgpuSync")
            return time.perf_counter() - t1, nrep
        _leg('packets via scatter')
        el4, nrep = packets_leg(False)
        ingest["packets_to_visibilities_scatter"] = {
            "value": round(8 * NINPUT * units_per_step_c * nrep / el4 / 1e9, 1), "unit": "Gb/s",
            "ms_per_step": round(el4 / nrep * 1e3, 4),
            "note": "device-resident packet slabs (5280 packets per gulp) -> xengSnap2UnpackAsync -> xengXgpuKernelAsync, "
                    "%d integrations (the path of rounds 2-3)" % nrep}
        _leg('packets in place, aligned')
        el4, nrep = packets_leg(2)
        nfb = ctypes.c_int(-1)
        ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(nfb))
        slab_a_vis = outs[(300 + 100 - 1) & 1].download(np.int32)
        ingest["packets_to_visibilities_payloads_on_cache_lines"] = {
            "value": round(8 * NINPUT * units_per_step_c * nrep / el4 / 1e9, 1), "unit": "Gb/s",
            "ms_per_step": round(el4 / nrep * 1e3, 4), "gulps_scattered_after_all": int(nfb.value), "packet_stride": stride_a,
            "note": "as packets_to_visibilities with every packet placed so that its payload starts on a 128-byte line (stride 6272, "
                    "the slab handed over at +96): in the packed slab half of the 64-byte rows straddle a 64-byte boundary"}
        _leg('packets in place, packed')
        el4, nrep = packets_leg(1)
        nfb = ctypes.c_int(-1)
        ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(nfb))
        slab_vis = outs[(300 + 100 - 1) & 1].download(np.int32)
        ingest["packets_to_visibilities"] = {
            "value": round(8 * NINPUT * units_per_step_c * nrep / el4 / 1e9, 1), "unit": "Gb/s",
            "ms_per_step": round(el4 / nrep * 1e3, 4), "gulps_scattered_after_all": int(nfb.value),
            "note": "device-resident packet slabs -> xengXgpuKernelAsyncSlab: every slab is verified on the device and, being "
                    "regular, read by the contraction where it lies (no scatter pass); %d integrations" % nrep}
        # (same slabs both ways: the two paths must agree word for word)
        for g in range(gulps_per_step):
            dst = ring.ptr + g * gulp_bytes
            slot = (4 * gulps_per_step + g) % (2 * gulps_per_step)
            ffi.check("unpack", L.xengSnap2UnpackAsync(slabs[slot].ptr, npk, stride, dst, slot * NTIME_GULP, NTIME_GULP, 0, NCHAN, NINPUT, 1))
            ffi.check(kern, L.xengXgpuKernelAsync(dst, outs[0].ptr, int(g == gulps_per_step - 1)))
        ffi.call("xengXgpuSync")
        want_vis = outs[0].download(np.int32)
        ingest["packets_to_visibilities"]["equals_scatter_path"] = bool(np.array_equal(slab_vis, want_vis))
        ingest["packets_to_visibilities_payloads_on_cache_lines"]["equals_scatter_path"] = bool(np.array_equal(slab_a_vis, want_vis))
        for b in slabs_a:
            b.free()
        # ... and on a LOSSY link (the reference's transmitter has a deliberate-loss switch: test_tx_mt.c:22,108-118).  Two receiver
        # models: "slot" -- the slot of a lost packet holds a duplicate of its neighbour (what the round-4 bench measured); "shift" -- the
        # receiver stores packets in arrival order, so everything behind a loss sits one slot early and the slab is shorter.  Round 5:
        # either way the contraction reads the packets where they lie, through the offset table built on the device (slab.hip);
        # round 4 sent every such gulp through zero-fill + scatter (+8 % / +27 % on the slot model)
        def lossy_copy(nlost, seed, shift):
            rs = np.random.RandomState(seed)
            out = []
            for k, b in enumerate(slabs):
                c = ffi.DeviceBuffer(b.nbytes)
                lost = sorted(int(p) for p in rs.choice(npk - 1, size=nlost(k), replace=False)) if nlost(k) else []
                if shift:
                    dst, src = 0, 0
                    for p in lost + [npk]:
                        if p > src:
                            ffi.call("xengMemcpy", c.ptr + dst * stride, b.ptr + src * stride, (p - src) * stride)
                            dst += p - src
                        src = p + 1
                    out.append((c, dst))
                else:
                    ffi.call("xengMemcpy", c.ptr, b.ptr, b.nbytes)
                    for p in lost:
                        ffi.call("xengMemcpy", c.ptr + p * stride, b.ptr + (p + 1) * stride, stride)
                    out.append((c, npk))
            return out

        def lossy_leg(sl, nrep=200, nwarm=60):
            kk = 0
            for it in range(nwarm + nrep):
                if it == nwarm:
                    ffi.call("xengXgpuSync")
                    t1 = time.perf_counter()
                for g in range(gulps_per_step):
                    slot = kk % (2 * gulps_per_step)
                    ffi.check("slab", L.xengXgpuKernelAsyncSlab(sl[slot][0].ptr, sl[slot][1], stride, slot * NTIME_GULP, 0, outs[it & 1].ptr, int(g == gulps_per_step - 1), None, 0))
                    kk += 1
                ffi.call("xengXgpuSyncLag", 1)
            ffi.call("xengXgpuSync")
            return time.perf_counter() - t1, nrep
        lossy = {}
        for name, nlost, shift in (("one_packet_lost_per_integration", lambda k: 1 if k % gulps_per_step == 2 else 0, False),
                                   ("one_percent_lost", lambda k: npk // 100, False),
                                   ("one_percent_lost_arrival_order", lambda k: npk // 100, True)):
            _leg('packets in place, ' + name)
            sl = lossy_copy(nlost, 11, shift)
            nfb, nirr = ctypes.c_int(-1), ctypes.c_int(-1)
            ffi.call("xengXgpuGetSlabStats", ctypes.byref(nfb), ctypes.byref(nirr))
            el5, nrep5 = lossy_leg(sl)
            ffi.call("xengXgpuGetSlabStats", ctypes.byref(nfb), ctypes.byref(nirr))
            got = outs[(200 + 60 - 1) & 1].download(np.int32)
            for g in range(gulps_per_step):
                dst = ring.ptr + g * gulp_bytes
                slot = ((200 + 60 - 1) * gulps_per_step + g) % (2 * gulps_per_step)
                ffi.check("unpack", L.xengSnap2UnpackAsync(sl[slot][0].ptr, sl[slot][1], stride, dst, slot * NTIME_GULP, NTIME_GULP, 0, NCHAN, NINPUT, 1))
                ffi.check(kern, L.xengXgpuKernelAsync(dst, outs[0].ptr, int(g == gulps_per_step - 1)))
            ffi.call("xengXgpuSync")
            lossy[name] = {"value": round(8 * NINPUT * units_per_step_c * nrep5 / el5 / 1e9, 1), "unit": "Gb/s", "ms_per_step": round(el5 / nrep5 * 1e3, 4),
                           "gulps_scattered_after_all": int(nfb.value), "gulps_read_through_an_irregular_table": int(nirr.value), "gulps": (200 + 60) * gulps_per_step,
                           "equals_scatter_path": bool(np.array_equal(got, outs[0].download(np.int32)))}
            for b, _ in sl:
                b.free()
        lossy["note"] = ("packets_to_visibilities with packets lost.  Slot model: the lost packet's slot holds a duplicate of the next one; arrival "
                         "order: the packets behind a loss sit one slot early, the slab is shorter.  Every gulp is read where it lies through its "
                         "offset table (round 4: zero-fill + scatter of every gulp with a hole); equals_scatter_path compares with "
                         "xengSnap2UnpackAsync + xengXgpuKernelAsync on the same slabs")
        ingest["packets_to_visibilities"]["lossy"] = lossy
        dgulp.free()
    _leg('beamform')
    # outside the timed region: BASELINE config 4 -- Beamform (32 beams, 96 chan, 960 samples, fp32 weights)
    # + BeamformSumBeams (16 dual-pol power beams, ntime_sum 24) on the same GPU
    beam = None
    if args.beamform and rank == 0 and world == 1:
        ffi.call("xengXgpuSync")
        NT_B, NB, NS = 960, 32, 24
        ffi.call("xengBeamformInitialize", gpu, NINPUT, NCHAN, NT_B, NB, 0)
        # weights as the Beamform block builds them (beamform_block.py:343-350) from random delays in [0,12) ns, amplitudes in
        # [10,17) and calibration gains as beamformer_test.py:131-139
        wts = beam_weights(0, 0xaabbccdd)
        dw = ffi.DeviceBuffer(wts.nbytes).upload(wts)
        dbeam = ffi.DeviceBuffer(NCHAN * NB * NT_B * 8)
        dpow = ffi.DeviceBuffer((NB // 2) * (NT_B // NS) * NCHAN * 16)
        btm = (ctypes.c_double * 2)()
        bcn = (ctypes.c_int * 2)()

        def bstep(i):
            # two consecutive 480-sample ring gulps form one 960-sample beamformer gulp (GPU_NGULP = 2)
            src = ring.ptr + ((2 * i) % (args.ring_gulps - 1)) * gulp_bytes
            ffi.check("run", L.xengBeamformRunVersioned(src, dbeam.ptr, dw.ptr, 1))
            ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
        for i in range(5):
            bstep(i)
        ffi.call("xengBeamformSync")
        ffi.call("xengBeamformSetProfiling", 1)
        ffi.call("xengBeamformGetTimes", btm, bcn)
        tb = time.perf_counter()
        nb_it = 40
        for i in range(nb_it):
            bstep(i)
        ffi.call("xengBeamformSync")
        elb = time.perf_counter() - tb
        ffi.call("xengBeamformGetTimes", btm, bcn)
        run_us = btm[0] / max(bcn[0], 1) * 1e3
        flop = NT_B * NCHAN * NB * NINPUT * 8
        beam = {"workload": "704 inputs, 96 chan, 32 beams, 960 samples/gulp, then 16 dual-pol power beams (ntime_sum 24)",
                "ingest_gbps": round(8 * NT_B * NCHAN * NINPUT / (elb / nb_it) / 1e9, 1),
                "run_us": round(run_us, 1), "integrate_us": round(btm[1] / max(bcn[1], 1) * 1e3, 1),
                "algorithmic_tflops": round(flop / run_us / 1e6, 1),
                "roofline": ({"kernel": "beamform_bf16x3_kernel", "bound": "mfma", "unit": "TFLOP/s",
                              "achieved": round(3 * flop / run_us / 1e6, 1), "peak": 2500.0,
                              "frac": round(3 * flop / run_us / 1e6 / 2500.0, 4),
                              "note": "bf16 MFMA flops issued = 3 x 8 x nbeam x ninput per (sample, chan) (three bf16 terms "
                                      "per fp32 weight); algorithmic fp32 flops are 1/3 of that"}
                             if os.environ.get("XENG_BEAM") == "bf16x3" else
                             {"kernel": "beamform_i8x3_kernel", "bound": "mfma", "unit": "TFLOP/s",
                              "achieved": round(3 * flop / run_us / 1e6, 1), "peak": round(PEAK_INT8_OPS / 1e12, 1),
                              "frac": round(3 * flop / run_us / 1e6 / (PEAK_INT8_OPS / 1e12), 4),
                              "note": "int8 MFMA ops issued = 3 x 8 x nbeam x ninput per (sample, chan) (three base-255 "
                                      "digits per fp32 weight); algorithmic fp32 flops are 1/3 of that (algorithmic_tflops, "
                                      "against the 157 TFLOP/s fp32 MFMA peak the survey names)"})}
        _leg('config 5')
        # ---- BASELINE config 5 (one GPU's share): Corr + CorrAcc + Beamform + SumBeams concurrently, each on
        # its own HIP stream, all fed from the same device-resident gulps
        ffi.call("xengXgpuSync")
        acc_long = ffi.DeviceBuffer(2 * matlen * 4)
        outs3 = outs + [ffi.DeviceBuffer(2 * matlen * 4)]
        ffi.call("xengBeamformSetProfiling", 0)
        nfull = 60
        bi = [0]

        def full_step(n):
            o = outs3[n % 3]
            for g in range(gulps_per_step):
                ffi.check(kern, kfn(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, o.ptr, int(g == gulps_per_step - 1)))
                gi[0] += 1
            for _ in range(2 + (n & 1)):            # 2.5 beamformer gulps of 960 samples per 2400-sample integration
                bstep(bi[0])
                bi[0] += 1
            ffi.call("xengXgpuSyncLag", 1)          # dump n-1 is complete: add it to the long accumulation
            if n >= 1:
                ffi.call("xengMapSync")             # the previous add has released its source span
                fn = L.xengMapAssignI32 if n == 1 else L.xengMapAddI32
                ffi.check("map", fn(acc_long.ptr, outs3[(n - 1) % 3].ptr, 2 * matlen))
        for n in range(6):
            full_step(n)
        ffi.call("xengDeviceSynchronize")
        tf = time.perf_counter()
        for n in range(6, 6 + nfull):
            full_step(n)
        ffi.call("xengDeviceSynchronize")
        elf = time.perf_counter() - tf
        beam["full_xengine_concurrent"] = {
            "ingest_gbps": round(8 * NINPUT * units_per_step_c * nfull / elf / 1e9, 1),
            "ms_per_integration": round(elf / nfull * 1e3, 4),
            "note": "config 5 on one GPU: per 2400-sample integration 5 gulps registered in place + 1 fused MFMA contraction (X-engine streams), "
                    "2.5 beamformer gulps + power sums (beam stream), 1 CorrAcc int32 map over 191 MB (map stream)"}
        # ... and with CorrAcc's long accumulation fused into the dump's epilogue (xengXgpuKernelAsyncAcc): no map kernel, the
        # two accumulators alternate so that consecutive dumps still overlap (they are added once per long integration)
        acc_pair = [acc_long, ffi.DeviceBuffer(2 * matlen * 4)]
        afn = L.xengXgpuKernelAsyncAcc

        def full_step_fused(n, first):
            o = outs3[n % 3]
            for g in range(gulps_per_step):
                ffi.check("kernel", afn(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, o.ptr, int(g == gulps_per_step - 1),
                                        acc_pair[n & 1].ptr, 1 if first else 2))
                gi[0] += 1
            for _ in range(2 + (n & 1)):
                bstep(bi[0])
                bi[0] += 1
            ffi.call("xengXgpuSyncLag", 1)
        if not args.sync_per_call:
            for n in range(6):
                full_step_fused(n, n < 2)
            ffi.call("xengDeviceSynchronize")
            tf = time.perf_counter()
            for n in range(6, 6 + nfull):
                full_step_fused(n, False)
            ffi.call("xengDeviceSynchronize")
            elf2 = time.perf_counter() - tf
            beam["full_xengine_concurrent"]["fused_corracc"] = {
                "ingest_gbps": round(8 * NINPUT * units_per_step_c * nfull / elf2 / 1e9, 1),
                "ms_per_integration": round(elf2 / nfull * 1e3, 4),
                "note": "the same with the CorrAcc add done in the contraction's epilogue (xengXgpuKernelAsyncAcc, two alternating "
                        "accumulators): one pass over the 191 MB accumulator per dump instead of a 574 MB map kernel"}
        # ... and with CorrAcc's long accumulation done GROUP by group (round 5, the CorrAcc block's default): the dumps of a group of
        # K integrations stay in their spans (K + 3 of them: 2.5 GB of 288) and are summed in one pass (xengMapSumI32) -- 191 + 382 / K
        # MB of traffic per dump instead of 574 (map) or 382 read-modify-written inside the contraction's epilogue (fused)
        KG = 10
        outs_g = outs3 + [ffi.DeviceBuffer(2 * matlen * 4) for _ in range(KG + 3 - len(outs3))]
        SrcArr = ctypes.c_void_p * KG

        def full_step_grouped(n):
            o = outs_g[n % len(outs_g)]
            for g in range(gulps_per_step):
                ffi.check(kern, kfn(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, o.ptr, int(g == gulps_per_step - 1)))
                gi[0] += 1
            for _ in range(2 + (n & 1)):
                bstep(bi[0])
                bi[0] += 1
            ffi.call("xengXgpuSyncLag", 1)          # dump n-1 is complete
            if n >= KG and n % KG == 0:             # ... and with it the whole group n-K .. n-1: one pass over its spans
                srcs = SrcArr(*[outs_g[(n - KG + j) % len(outs_g)].ptr for j in range(KG)])
                ffi.check("sum", L.xengMapSumI32(acc_long.ptr, srcs, KG, 2 * matlen, int(n > KG)))
            elif n % KG == 1:
                ffi.call("xengMapSync")             # (the sum has read its spans before dump n + 2 writes the first of them again)
        if not args.sync_per_call:
            for n in range(KG, KG + 6):
                full_step_grouped(n)
            ffi.call("xengDeviceSynchronize")
            tf = time.perf_counter()
            for n in range(2 * KG, 2 * KG + nfull):
                full_step_grouped(n)
            ffi.call("xengDeviceSynchronize")
            elf3 = time.perf_counter() - tf
            fx = beam["full_xengine_concurrent"]
            fx["map_per_dump"] = {"ingest_gbps": fx["ingest_gbps"], "ms_per_integration": fx["ms_per_integration"], "note": fx["note"]}
            fx.update({"ingest_gbps": round(8 * NINPUT * units_per_step_c * nfull / elf3 / 1e9, 1), "ms_per_integration": round(elf3 / nfull * 1e3, 4),
                       "corracc": "grouped: %d dumps per xengMapSumI32 pass" % KG,
                       "note": "config 5 on one GPU: per 2400-sample integration 5 gulps registered in place + 1 fused-corner-turn MFMA contraction (X-engine "
                               "streams), 2.5 beamformer gulps + power sums (beam stream), and CorrAcc's long accumulation as one pass over the spans of every "
                               "%d dumps (map stream: 191 + 382 / %d MB per dump); map_per_dump / fused_corracc: the same with the reference's per-dump "
                               "map and with the add in the contraction's epilogue" % (KG, KG)})
        _leg('config 5 from packets')
        # ... and fed from PACKETS (north star: "throughput on synthetic F-engine packets"): the ten device-resident packet slabs of
        # the ingest leg above instead of replay gulps.  Both consumers read the slabs where they lie (xengXgpuKernelAsyncSlab,
        # xengBeamformRunSlabs: two slabs = one 960-sample beam gulp); the comparison leg scatters every slab into the replay
        # ring first (xengSnap2UnpackAsync) and runs the plain calls on the copies.
        if not args.sync_per_call:
            sfn, ufn, bsl = L.xengXgpuKernelAsyncSlab, L.xengSnap2UnpackAsync, L.xengBeamformRunSlabs
            nslab = 2 * gulps_per_step

            def full_step_packets(n, first, in_place):
                # (CorrAcc as in the leg above and in the CorrAcc block: the spans of a group of KG dumps summed in one pass)
                o = outs_g[n % len(outs_g)]
                for g in range(gulps_per_step):
                    slot = gi[0] % nslab
                    if in_place:
                        ffi.check("slab", sfn(slabs[slot].ptr, npk, stride, slot * NTIME_GULP, 0, o.ptr, int(g == gulps_per_step - 1), None, 0))
                    else:
                        dst = ring.ptr + slot * gulp_bytes
                        ffi.check("unpack", ufn(slabs[slot].ptr, npk, stride, dst, slot * NTIME_GULP, NTIME_GULP, 0, NCHAN, NINPUT, 1))
                        ffi.check(kern, kfn(dst, o.ptr, int(g == gulps_per_step - 1)))
                    gi[0] += 1
                for _ in range(2 + (n & 1)):
                    k0 = (2 * bi[0]) % nslab
                    if in_place:
                        ffi.check("run", bsl(slabs[k0].ptr, npk, NTIME_GULP, slabs[k0 + 1].ptr, npk, stride, k0 * NTIME_GULP, 0, dbeam.ptr, dw.ptr, 1))
                    else:       # (the copies of slabs k0, k0 + 1 lie side by side in the replay ring: scattered by this or an earlier integration)
                        ffi.check("run", L.xengBeamformRunVersioned(ring.ptr + k0 * gulp_bytes, dbeam.ptr, dw.ptr, 1))
                    ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
                    bi[0] += 1
                ffi.call("xengXgpuSyncLag", 1)
                if n >= KG and n % KG == 0:
                    srcs = SrcArr(*[outs_g[(n - KG + j) % len(outs_g)].ptr for j in range(KG)])
                    ffi.check("sum", L.xengMapSumI32(acc_long.ptr, srcs, KG, 2 * matlen, int(n > KG)))
                elif n % KG == 1:
                    ffi.call("xengMapSync")
            pk = {}
            snap = {}
            for in_place in (False, True):
                gi[0] = bi[0] = 0
                for n in range(KG, KG + 6):
                    full_step_packets(n, False, in_place)
                ffi.call("xengDeviceSynchronize")
                tf = time.perf_counter()
                for n in range(2 * KG, 2 * KG + nfull):
                    full_step_packets(n, False, in_place)
                ffi.call("xengDeviceSynchronize")
                elp = time.perf_counter() - tf
                pk["in_place" if in_place else "through_scatter"] = {
                    "ingest_gbps": round(8 * NINPUT * units_per_step_c * nfull / elp / 1e9, 1), "ms_per_integration": round(elp / nfull * 1e3, 4)}
                snap[in_place] = (outs_g[(2 * KG + nfull - 1) % len(outs_g)].download(np.int32), dbeam.download(np.uint32), dpow.download(np.uint32))
            nfx, nfb = ctypes.c_int(-1), ctypes.c_int(-1)
            ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(nfx))
            ffi.call("xengBeamformGetSlabFallbacks", ctypes.byref(nfb))
            pk["slabs_scattered_after_all"] = {"corr": int(nfx.value), "beamform": int(nfb.value)}
            # ... and on a lossy link (round 5): the same ten slabs in arrival order with 1 % of their packets lost (everything behind a
            # loss one slot early, the slabs shorter).  Both consumers read them where they lie -- the contraction through offset tables,
            # the beamformer through packet indices -- once the device has told the host that the link is lossy (XENG_SLAB_TABLES=0:
            # round 4's zero-fill + scatter of every gulp, for both)
            rsl = np.random.RandomState(23)
            lossy_slabs = []
            for b in slabs:
                cbuf = ffi.DeviceBuffer(b.nbytes)
                lost = sorted(int(q) for q in rsl.choice(npk - 1, size=npk // 100, replace=False))
                dstp = srcp = 0
                for q in lost + [npk]:
                    if q > srcp:
                        ffi.call("xengMemcpy", cbuf.ptr + dstp * stride, b.ptr + srcp * stride, (q - srcp) * stride)
                        dstp += q - srcp
                    srcp = q + 1
                lossy_slabs.append((cbuf, dstp))

            def full_step_lossy(n):
                o = outs_g[n % len(outs_g)]
                for g in range(gulps_per_step):
                    slot = gi[0] % nslab
                    ffi.check("slab", sfn(lossy_slabs[slot][0].ptr, lossy_slabs[slot][1], stride, slot * NTIME_GULP, 0, o.ptr, int(g == gulps_per_step - 1), None, 0))
                    gi[0] += 1
                for _ in range(2 + (n & 1)):
                    k0 = (2 * bi[0]) % nslab
                    ffi.check("run", bsl(lossy_slabs[k0][0].ptr, lossy_slabs[k0][1], NTIME_GULP, lossy_slabs[k0 + 1][0].ptr, lossy_slabs[k0 + 1][1], stride,
                                         k0 * NTIME_GULP, 0, dbeam.ptr, dw.ptr, 1))
                    ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
                    bi[0] += 1
                ffi.call("xengXgpuSyncLag", 1)
                if n >= KG and n % KG == 0:
                    srcs = SrcArr(*[outs_g[(n - KG + j) % len(outs_g)].ptr for j in range(KG)])
                    ffi.check("sum", L.xengMapSumI32(acc_long.ptr, srcs, KG, 2 * matlen, int(n > KG)))
                elif n % KG == 1:
                    ffi.call("xengMapSync")
            gi[0] = bi[0] = 0
            for n in range(KG, 2 * KG):
                full_step_lossy(n)
            ffi.call("xengDeviceSynchronize")
            st = [ctypes.c_int(-1) for _ in range(4)]
            ffi.call("xengXgpuGetSlabStats", ctypes.byref(st[0]), ctypes.byref(st[1]))
            ffi.call("xengBeamformGetSlabStats", ctypes.byref(st[2]), ctypes.byref(st[3]))
            tf = time.perf_counter()
            for n in range(2 * KG, 2 * KG + nfull):
                full_step_lossy(n)
            ffi.call("xengDeviceSynchronize")
            elp = time.perf_counter() - tf
            ffi.call("xengXgpuGetSlabStats", ctypes.byref(st[0]), ctypes.byref(st[1]))
            ffi.call("xengBeamformGetSlabStats", ctypes.byref(st[2]), ctypes.byref(st[3]))
            pk["one_percent_lost_arrival_order"] = {
                "ingest_gbps": round(8 * NINPUT * units_per_step_c * nfull / elp / 1e9, 1), "ms_per_integration": round(elp / nfull * 1e3, 4),
                "corr": {"gulps_scattered": int(st[0].value), "gulps_through_an_irregular_table": int(st[1].value)},
                "beamform": {"parts_scattered": int(st[2].value), "parts_through_an_irregular_index": int(st[3].value)},
                "slab_tables_env": os.environ.get("XENG_SLAB_TABLES", "unset: by the device's hint")}
            for b, _ in lossy_slabs:
                b.free()
            pk["in_place_equals_through_scatter"] = {"visibilities": bool(np.array_equal(snap[0][0], snap[1][0])), "beams": bool(np.array_equal(snap[0][1], snap[1][1])),
                                                     "power_sums": bool(np.array_equal(snap[0][2], snap[1][2]))}
            pk["note"] = ("config 5 (grouped CorrAcc, as above), fed from ten device-resident slabs of 5280 SNAP2 packets each: per integration 5 slabs "
                          "to the correlator and 2.5 slab pairs to the beamformer; in_place = both read the packets where they lie")
            beam["full_xengine_concurrent"]["from_packet_slabs"] = pk
            del snap
        for b in slabs:
            b.free()
        # the same concurrent pattern once more on KNOWN inputs (three integrations of ring gulps 0..4, beams of gulps 0+1),
        # kept for the oracle to check in the cpu_baseline leg: one dumped span, the CorrAcc sum (= 3 x that span) and one
        # beam gulp with its power sums
        if not args.no_cpu_baseline and not args.sync_per_call:
            # (the PCIe and ingest legs above have rewritten the replay ring: put gulps 0..4 of the generator back)
            rs5 = np.random.RandomState(0xdeadbeef)
            for g in range(gulps_per_step):
                ring.upload(rs5.randint(0, 255, size=gulp_bytes, dtype=np.uint8), offset=g * gulp_bytes)
            for n in range(3):
                o = outs3[n]
                for g in range(gulps_per_step):
                    ffi.check(kern, kfn(ring.ptr + g * gulp_bytes, o.ptr, int(g == gulps_per_step - 1)))
                ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dbeam.ptr, dw.ptr, 1))
                ffi.check("int", L.xengBeamformIntegrate(dbeam.ptr, dpow.ptr, NS))
                ffi.call("xengXgpuSyncLag", 1)
                if n >= 1:
                    ffi.call("xengMapSync")
                    ffi.check("map", (L.xengMapAssignI32 if n == 1 else L.xengMapAddI32)(acc_long.ptr, outs3[n - 1].ptr, 2 * matlen))
            ffi.call("xengXgpuSync")
            ffi.call("xengMapSync")
            ffi.check("map", L.xengMapAddI32(acc_long.ptr, outs3[2].ptr, 2 * matlen))
            ffi.call("xengMapSync")
            ffi.call("xengBeamformSync")
            corracc_sum = acc_long.download(np.int32).astype(np.int64)
            # the fused flavour on the same three integrations (accumulators alternate: A gets n = 0 and 2, B gets n = 1)
            for n in range(3):
                for g in range(gulps_per_step):
                    ffi.check("kernel", afn(ring.ptr + g * gulp_bytes, outs3[n].ptr, int(g == gulps_per_step - 1),
                                            acc_pair[n & 1].ptr, 1 if n < 2 else 2))
                ffi.call("xengXgpuSyncLag", 1)
            ffi.call("xengXgpuSync")
            corracc_fused = acc_pair[0].download(np.int32).astype(np.int64) + acc_pair[1].download(np.int32).astype(np.int64)
            # the grouped flavour: the three dumps above, still in their spans, summed in one pass (acc_long is acc_pair[0]: read out first)
            g3 = (ctypes.c_void_p * 3)(*[outs3[n].ptr for n in range(3)])
            ffi.check("sum", L.xengMapSumI32(acc_long.ptr, g3, 3, 2 * matlen, 0))
            ffi.call("xengMapSync")
            verify = {"vis": outs3[2].download(np.int32), "corracc": corracc_sum,
                      "corracc_fused": corracc_fused,
                      "corracc_grouped": acc_long.download(np.int32).astype(np.int64),
                      "beams": dbeam.download(np.complex64).reshape(NCHAN, NB, NT_B), "weights": wts, "ntime_sum": NS,
                      "power": dpow.download(np.float32).reshape(NB // 2, NT_B // NS, NCHAN, 4)}
        # the reference's "integrated" mode (bfBeamformInitialize ntime_blocks > 0, beamform_block.py:108-110): power sums
        # formed in the beamformer kernel's epilogue, no voltage beams in memory
        ffi.call("xengBeamformInitialize", gpu, NINPUT, NCHAN, NT_B, NB, NT_B // NS)
        for i in range(6):
            ffi.check("run", L.xengBeamformRunVersioned(ring.ptr, dpow.ptr, dw.ptr, 1))
            ffi.call("xengBeamformSync")
        ffi.call("xengBeamformSetProfiling", 1)
        ffi.call("xengBeamformGetTimes", btm, bcn)
        for i in range(40):
            ffi.check("run", L.xengBeamformRunVersioned(ring.ptr + ((2 * i) % (args.ring_gulps - 1)) * gulp_bytes, dpow.ptr, dw.ptr, 1))
        ffi.call("xengBeamformSync")
        ffi.call("xengBeamformGetTimes", btm, bcn)
        ffi.call("xengBeamformSetProfiling", 0)
        beam["integrated_mode"] = {"run_us": round(btm[0] / max(bcn[0], 1) * 1e3, 1), "integrate_launches": int(bcn[1]),
                                   "note": "ntime_blocks = %d: one launch per gulp, power sums in the epilogue (the composed path is "
                                           "run_us + integrate_us above)" % (NT_B // NS)}
        ffi.call("xengBeamformDestroy")
    per_rank_ms = [round(v / args.steps * 1e3, 4) for v in _sh.gather_over_ranks(dist, el)]
    placements = [None] * world
    if dist is not None:
        import torch
        dist.all_gather_object(placements, {"numa_node": pin["numa_node"], "ncpus": len(pin["cpus"]), "source": pin["source"]})
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        dist.barrier()
    else:
        placements = [{"numa_node": pin["numa_node"], "ncpus": len(pin["cpus"]), "source": pin["source"]}]

    units_per_step = ACC_LEN * NCHAN
    total_units = units_per_step * args.steps * world
    gbps = 8 * NINPUT * total_units / el / 1e9
    cmacs = CMAC_PER_UNIT * total_units / el
    ct_ms = tm[0] / max(cn[0], 1)
    mm_ms = tm[1] / max(cn[1], 1)
    ops_per_launch = OPS_PER_UNIT * units_per_step
    achieved = ops_per_launch / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0
    ct_bytes = 2 * gulp_bytes
    # HBM traffic of the dominant kernel from the committed PMC passes (profiles/pmc_run.sh:
    # FETCH_SIZE x2 for the gfx950 wide-load under-count + WRITE_SIZE, per launch); null if absent
    fused, fp6 = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengXgpuGetPath", ctypes.byref(fused), ctypes.byref(fp6))
    kname = "xcorr_fused_kernel" if fused.value else ("xcorr_fp6_kernel" if fp6.value else "xcorr_mfma_kernel")
    # (PMC counters cannot be collected inside this run; the committed figure counts only while the X-engine sources it was
    # taken from are the ones this library was built from, else traffic is null)
    traffic = traffic_slabs = None
    import hashlib
    for rnd in ("r05", "r04", "r03"):              # (the latest PMC pass whose X-engine sources are the ones this library was built from)
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            hh = hashlib.sha256()
            for f in pmc.get("xcorr_sources", []):
                with open(os.path.join(ROOT, "caltech-bifrost-dsp_amd", "csrc", f), "rb") as fh:
                    hh.update(fh.read())
            if hh.hexdigest() == pmc.get("xcorr_sources_sha256"):
                traffic = pmc.get(kname + "_bytes_per_launch")
                traffic_slabs = pmc.get(kname + "_slabs_bytes_per_launch")
                break
        except (OSError, ValueError):
            pass
    res = {
        "metric": "xengine_ingest_gbps_704in_96ch", "value": round(gbps, 2), "unit": "Gb/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": args.prewarm,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int8 (4+4-bit samples) -> int32",
        "data": "synthetic" if args.data == "random" else "synthetic-%s (diagnostic: the reported result uses uniform random nibbles)" % args.data,
        "config": {"workload": "704-input (352 ant x 2 pol), %d chan/GPU, 4+4b->int32 correlator, acc_len %d = %d gulps x %d"
                               % (NCHAN, ACC_LEN, gulps_per_step, NTIME_GULP),
                   "nchan_total": NCHAN * world, "sharding": "channels, %d per GPU, no collective" % NCHAN,
                   "call_mode": call_mode,
                   "input": "device-resident replay ring, %d gulps" % args.ring_gulps},
        "sustained": sustained,
        "per_rank_ms": per_rank_ms, "per_rank_ms_min": min(per_rank_ms), "per_rank_ms_max": max(per_rank_ms),
        "rank_placement": placements,
        "cmac_per_s": cmacs,
        "mfma_peak_frac_end_to_end": round(8 * cmacs / (PEAK_INT8_OPS * world), 4),
        "design_rate_x": round(gbps / world / 12.94, 1),
        # one launch of the dominant kernel per step: achieved = algorithmic ops per launch / time per step on this GPU
        # (in the streaming mode two consecutive launches share the GPU, so the HIP-event duration of ONE launch is
        # longer than a step -- it is kept as a side key together with the duration of a launch that runs alone)
        "roofline": {"kernel": kname, "bound": "mfma",
                     "achieved": round(ops_per_launch / (el / args.steps) / 1e12, 1),
                     "peak": round(PEAK_INT8_OPS / 1e12, 1), "unit": "TFLOP/s",
                     "frac": round(ops_per_launch / (el / args.steps) / PEAK_INT8_OPS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": units_per_step * NINPUT + 2 * matlen * 4,
                     "avg_launch_us_overlapped": round(mm_ms * 1e3, 1),
                     "launch_concurrency": round(mm_ms / (el / args.steps * 1e3), 2) if el > 0 else None,
                     "frac_of_overlapped_launch": round(achieved / (PEAK_INT8_OPS / 1e12), 4),
                     "note": "int8 TOP/s; algorithmic ops = 8*704*705/2 per (sample,chan) x %d units per launch; frac = ops "
                             "per launch / (time per step = one launch) / peak.  HIP events on the X-engine streams: avg "
                             "launch %.1f us over %d launches while %.2f launches share the GPU (avg_launch_us_overlapped); "
                             "roofline_isolated_launches times the same kernel alone"
                             % (units_per_step, mm_ms * 1e3, cn[1], mm_ms / (el / args.steps * 1e3) if el > 0 else 0.0)},
        "device": info,
    }
    if cn[0] > 0:
        res["corner_turn"] = {"bound": "hbm", "avg_us": round(ct_ms * 1e3, 2), "launches": int(cn[0]),
                              "achieved_GBs": round(ct_bytes / (ct_ms * 1e-3) / 1e9, 1) if ct_ms > 0 else 0.0,
                              "peak_GBs": HBM_PEAK_GBS, "bytes_per_launch": ct_bytes,
                              "note": "two-pass path: corner turn, or raw gulp copy of the synchronous calls"}
    else:
        res["corner_turn"] = {"fused": True, "note": "no separate pass: gulps are read in place and transposed in "
                                                     "the contraction kernel's LDS staging (ds_read_b64_tr_b8)"}
    if pcie is not None:
        res["pcie_inclusive"] = pcie
    if beam is not None:
        res["beamform"] = beam
    if pktz is not None:
        res["corr_output_packetize"] = pktz
    if ingest is not None:
        # the contraction of the packet legs against the same roofline as the headline (one launch per integration; the scatter /
        # verify passes run beside it).  traffic: PMC, the descriptor instantiation on packed slabs (stride 6176), else null
        for key, tr, alg in (("packets_to_visibilities", traffic_slabs, units_per_step_c * NINPUT + 2 * matlen * 4 + gulps_per_step * NTIME_GULP * (NINPUT // 64) * 32),
                             ("packets_to_visibilities_payloads_on_cache_lines", None, units_per_step_c * NINPUT + 2 * matlen * 4 + gulps_per_step * NTIME_GULP * (NINPUT // 64) * 32),
                             ("packets_to_visibilities_scatter", None, None)):
            leg = ingest.get(key)
            if isinstance(leg, dict) and leg.get("ms_per_step"):
                ach = OPS_PER_UNIT * units_per_step_c / (leg["ms_per_step"] * 1e-3)
                leg["roofline"] = {"kernel": "xcorr_fused_kernel" + ("" if key.endswith("scatter") else " (gulps by descriptor)"), "bound": "mfma",
                                   "achieved": round(ach / 1e12, 1), "peak": round(PEAK_INT8_OPS / 1e12, 1), "unit": "TFLOP/s",
                                   "frac": round(ach / PEAK_INT8_OPS, 4), "traffic": tr, "algorithmic_bytes_per_launch": alg}
        res["ingest_unpack"] = ingest
    if iso_cn[1] > 0:
        iso_mm = iso_tm[1] / iso_cn[1]
        iso_ach = ops_per_launch / (iso_mm * 1e-3) / 1e12
        res["roofline_isolated_launches"] = {
            "kernel": kname, "avg_us": round(iso_mm * 1e3, 1), "achieved": round(iso_ach, 1),
            "frac": round(iso_ach / (PEAK_INT8_OPS / 1e12), 4), "launches": int(iso_cn[1]),
            "corner_turn_avg_us": round(iso_tm[0] / max(iso_cn[0], 1) * 1e3, 2),
            "note": "same kernel, one integration at a time (outside the timed region): in the timed streaming "
                    "region consecutive launches overlap (the next one takes over CUs as work-groups of the "
                    "previous one run out of items), which lengthens each launch but shortens the step"}
    _leg('sync per call')
    # outside the timed region: the reference's own call semantics (corr_block.py:445: synchronous bfXgpuKernel per gulp:
    # the input may be recycled on return, the dump is complete on return) -- what an unmodified Corr on a circular
    # bifrost ring would see
    if rank == 0 and world == 1 and not args.sync_per_call and not args.sync_per_integration:
        ffi.call("xengXgpuSync")
        nrep, nwarm = 200, 50
        for k in range(nwarm + nrep):
            if k == nwarm:
                t1 = time.perf_counter()
            for g in range(gulps_per_step):
                ffi.check("xengXgpuKernel", L.xengXgpuKernel(ring.ptr + (gi[0] % args.ring_gulps) * gulp_bytes, outs[k & 1].ptr,
                                                             int(g == gulps_per_step - 1)))
                gi[0] += 1
        el5 = time.perf_counter() - t1
        res["sync_per_call"] = {"value": round(8 * NINPUT * units_per_step_c * nrep / el5 / 1e9, 1), "unit": "Gb/s",
                                "ms_per_step": round(el5 / nrep * 1e3, 4),
                                "note": "the drop-in synchronous call per gulp (raw device copy of every non-dump gulp + wait; the "
                                        "dump gulp is read in place and the call returns when the visibilities are complete), %d integrations" % nrep}
    _leg('blocks')
    # outside the timed region: the Corr BLOCK itself (blocks/corr_block.py: ring protocol, header handling, state machine,
    # one Python thread) on in-repo device rings at config-2 size, fed by a zero-copy replay source: the rate a pipeline
    # user of the block sees, next to the C-ABI rate above
    if rank == 0 and world == 1 and args.beamform and args.blocks and not args.sync_per_call and not args.sync_per_integration:
        res["corr_block"] = corr_block_leg(ffi, ring, gulp_bytes, args.ring_gulps, gpu)
        res["config5_blocks"] = config5_blocks_leg(ffi, ring, gulp_bytes, args.ring_gulps, gpu)
        res["config5_blocks"]["from_packet_slabs"] = config5_blocks_leg(ffi, ring, gulp_bytes, args.ring_gulps, gpu, nint=300, nwarm=300, long_len=50, from_slabs=True)
    _leg('one call per integration')
    # outside the timed region: SURVEY 8d's other device-resident case, one 2400-sample call per integration
    # (xGPU's NTIME = acc_len; needs its own context, so it runs last)
    if args.beamform and rank == 0 and world == 1 and args.ring_gulps >= 2 * gulps_per_step:
        ffi.call("xengXgpuSync")
        ffi.call("xengXgpuConfigure", NINPUT // 2, 2, NCHAN, ACC_LEN, 1)
        ffi.call("xengXgpuInitialize", gpu)
        nrep, nwarm = 300, 100
        for k in range(nwarm + nrep):
            if k == nwarm:
                ffi.call("xengXgpuSync")
                t1 = time.perf_counter()
            ffi.check(kern, L.xengXgpuKernelAsync(ring.ptr + (k & 1) * gulps_per_step * gulp_bytes, outs[k & 1].ptr, 1))
            ffi.call("xengXgpuSyncLag", 1)
        ffi.call("xengXgpuSync")
        el3 = time.perf_counter() - t1
        res["single_call_t2400"] = {"value": round(8 * NINPUT * units_per_step_c * nrep / el3 / 1e9, 1), "unit": "Gb/s",
                                    "ms_per_step": round(el3 / nrep * 1e3, 4),
                                    "note": "ntime_gulp = acc_len = 2400: one enqueue-only call per integration, same streaming pattern"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if host_cpus:
                os.sched_setaffinity(0, host_cpus)      # the CPU baseline runs on all host cores, not on the GPU's share
            res["cpu_baseline"] = cpu_baseline(verify=verify)
            if verify is not None and beam is not None:
                beam["full_xengine_concurrent"]["verified"] = res["cpu_baseline"].pop("config5_check")
        print(json.dumps(res))
    ffi.call("xengXgpuDestroy")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
