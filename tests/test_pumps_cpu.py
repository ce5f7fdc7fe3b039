"""The native per-gulp loops (csrc/pyext/xfast.cpp BeamPump, CorrPump) on CPU rings (round 5).

The pumps run the steady state of Beamform / BeamformSumBeams / Corr inside the extension with the interpreter lock released.
In the product they call libxeng; here their compute table points at tests/fake_backend.PumpOracleBackend (the oracle on
system-space span memory), so everything they do around the compute calls -- ring traffic, retiring, the stop-flag hand-back, the
error paths -- runs without a GPU.  Reference behaviour the results are held to: the blocks' own Python loops
(corr_block.py:388-466, beamform_block.py:411-461), which every scenario also runs."""
import json
import threading
import time

import numpy as np
import pytest

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd import ring as ringmod
from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, Corr
from caltech_bifrost_dsp_amd.ring import Ring
from oracle import xeng_oracle as orc
from tests.fake_backend import OracleBackend, PumpOracleBackend
from tests.pipeline_util import LOG, GatedSource, Sink, Source, run_blocks, source_header, wait_for


@pytest.fixture(autouse=True)
def native_rings():
    was = ringmod.IMPLEMENTATION
    ringmod.IMPLEMENTATION = "native"
    yield
    ringmod.IMPLEMENTATION = was


def cmd(idx, **kwargs):
    return json.dumps({"cmd": "update", "val": {"kwargs": kwargs}, "id": str(idx)})


def ring_balance(*rings):
    """every span allocation of these (system-space, non-recycling) rings has been given back"""
    for r in rings:
        i = r.info()
        assert i["alloc"] == i["free"], "ring %s: %d allocations, %d freed" % (r.name, i["alloc"], i["free"])


# ====================================================================================================== CorrPump
def corr_run(backend_cls, nseq_gulps, g=2, acc_len=4, autostartat=0, seq0s=None, commands=None, C=2, S=4, fail=None, tail=0):
    rng = np.random.default_rng(1)
    iring, oring = Ring("in"), Ring("out")
    be = backend_cls(fail=fail) if backend_cls is PumpOracleBackend else backend_cls()
    blk = Corr(LOG, iring, oring, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc_len, autostartat=autostartat, backend=be)
    for c in (commands or []):
        blk.process_command_strings(c)
    seqs, data = [], []
    for k, ng in enumerate(nseq_gulps):
        d = rng.integers(0, 256, (ng * g + (tail if k == len(nseq_gulps) - 1 else 0), C, S, 2), dtype=np.uint8)
        data.append(d)
        seqs.append((source_header(C, S, 2, seq0=(seq0s[k] if seq0s else 0)), d, g * C * S * 2))
    sink = Sink(oring, blk.ogulp_size)
    run_blocks([blk], Source(iring, seqs), [sink])
    return sink.sequences, be, data, blk, (iring, oring)


SCENARIOS = {
    "plain": dict(nseq_gulps=[16]),
    "start in the future, short tail": dict(nseq_gulps=[11], autostartat=6, tail=1),
    "start = -1 rounds up": dict(nseq_gulps=[8], autostartat=-1, seq0s=[2]),
    "new upstream sequence: recovery": dict(nseq_gulps=[4, 30], seq0s=[0, 8]),
    "sequence ends mid-integration": dict(nseq_gulps=[7, 8], seq0s=[0, 1000]),
    "start never reached": dict(nseq_gulps=[6], autostartat=1000),
    "stopped from the start": dict(nseq_gulps=[6], commands=[cmd(1, acc_len=0)]),
    "one gulp per integration": dict(nseq_gulps=[9], g=4, acc_len=4),
    "long integrations": dict(nseq_gulps=[24], g=2, acc_len=12),
}


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_corr_through_the_pump_equals_the_python_loop(name):
    """Every scenario through Corr's Python per-gulp loop (OracleBackend has no corr_pump) and through the native loop: the same
    output sequences, headers, time tags, spans (= the oracle's integrations), the same X-engine calls in the same order."""
    kw = SCENARIOS[name]
    ref, be0, data, blk0, _ = corr_run(OracleBackend, **kw)
    got, be1, _, blk1, rings = corr_run(PumpOracleBackend, **kw)
    assert blk0.stats.get('pump') is None and (blk1.stats.get('pump') is True or not got and not ref)
    assert be1.callback_errors == []
    assert len(got) == len(ref)
    for (h0, t0, s0), (h1, t1, s1) in zip(ref, got):
        assert h0 == h1 and t0 == t1 and len(s0) == len(s1)
        assert all(np.array_equal(a, b) for a, b in zip(s0, s1))
    assert be1.kernel_calls == be0.kernel_calls and be1.resets == be0.resets
    assert blk1.stats['state'] == blk0.stats['state']
    if 'last_end_sample' in blk0.stats:
        assert blk1.stats['last_end_sample'] == blk0.stats['last_end_sample']
    del got, ref
    ring_balance(*rings)


def test_corr_pump_hands_a_gulp_back_when_a_command_arrives():
    """A command that arrives while the pump waits for gulp 6 takes effect on exactly gulp 6 -- the gulp is handed back to the
    block's Python unprocessed (status 2), the running integration is dropped, and the new start time is honoured -- as the
    per-gulp loop would have done (corr_block.py:392-404).  The source holds gulp 6 back until the command is in: no timing."""
    C, S, g, acc = 2, 4, 2, 4
    vin = np.random.default_rng(3).integers(0, 256, (40, C, S, 2), dtype=np.uint8)
    res = {}
    for cls in (OracleBackend, PumpOracleBackend):
        iring, oring = Ring("in"), Ring("out")
        be = cls()
        blk = Corr(LOG, iring, oring, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
        gate6 = threading.Event()
        src = GatedSource(iring, source_header(C, S, 2), vin, g * C * S * 2, {6: gate6})
        sink = Sink(oring, blk.ogulp_size)
        th = threading.Thread(target=blk.main, daemon=True)
        sink.start(); th.start(); src.start()
        wait_for(lambda: len(be.kernel_calls) == 6, "six gulps registered")       # gulps 0..5 are in; the block waits for gulp 6
        blk.process_command_strings(cmd(1, start_time=20, acc_len=8))
        gate6.set()
        for t in (src, th, sink):
            t.join(20)
            assert not t.is_alive()
        res[cls] = [(h, tag, [sp.view(np.int32).copy() for sp in spans]) for h, tag, spans in sink.sequences]
        assert be.resets == 2              # the integration [12, 16) was dropped at gulp 6; [36, 44) was cut off by the end of the stream
        if cls is PumpOracleBackend:
            assert blk.stats.get('pump') is True and be.callback_errors == []
    (h0, _, s0), (h1, _, s1) = res[PumpOracleBackend]
    assert h0['seq0'] == 0 and h0['acc_len'] == 4 and len(s0) == 3                 # [0,4) [4,8) [8,12)
    assert h1['seq0'] == 20 and h1['acc_len'] == 8 and len(s1) == 2                # [20,28) [28,36)
    assert np.array_equal(s1[0], orc.xgpu_correlate(vin[20:28], S, C))
    a, b = res[OracleBackend], res[PumpOracleBackend]
    assert len(a) == len(b) and all(x[0] == y[0] and x[1] == y[1] and len(x[2]) == len(y[2]) and all(np.array_equal(p, q) for p, q in zip(x[2], y[2]))
                                    for x, y in zip(a, b))


def test_corr_pump_stop_and_restart():
    """acc_len = 0 (the stop command) while the pump runs, then a restart: gulps pass by while stopped, and the integrations after
    the restart are the oracle's."""
    C, S, g, acc = 2, 4, 2, 4
    vin = np.random.default_rng(4).integers(0, 256, (48, C, S, 2), dtype=np.uint8)
    iring, oring = Ring("in"), Ring("out")
    be = PumpOracleBackend()
    blk = Corr(LOG, iring, oring, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    g4, g10 = threading.Event(), threading.Event()
    src = GatedSource(iring, source_header(C, S, 2), vin, g * C * S * 2, {4: g4, 10: g10})
    sink = Sink(oring, blk.ogulp_size)
    th = threading.Thread(target=blk.main, daemon=True)
    sink.start(); th.start(); src.start()
    wait_for(lambda: len(be.kernel_calls) == 4, "four gulps registered")
    blk.process_command_strings(cmd(1, acc_len=0))
    g4.set()
    wait_for(lambda: src.written == 10 and blk.stats.get('state') == 'stopped', "gulps 4..9 written while stopped")
    blk.process_command_strings(cmd(2, acc_len=4, start_time=24))
    g10.set()
    for t in (src, th, sink):
        t.join(20)
        assert not t.is_alive()
    assert be.callback_errors == []
    (h0, t0, s0), (h1, t1, s1) = sink.sequences
    assert (h0['seq0'], t0, len(s0)) == (0, 1, 2) and (h1['seq0'], t1, len(s1)) == (24, 2, 6)
    assert np.array_equal(s1[0].view(np.int32), orc.xgpu_correlate(vin[24:28], S, C))
    assert len(be.kernel_calls) == 4 + 12                                          # nothing was registered while stopped / waiting


@pytest.mark.parametrize("entry,at", [("xgpu_try_kernel", 7), ("xgpu_sync_lag", 2), ("xgpu_try_kernel", 1)])
def test_corr_pump_error_mid_flight_gives_every_span_back(entry, at):
    """An enqueue (or a wait) that fails in the middle of the stream: the block raises, nothing more is committed, and every span
    -- the gulps of the integration in progress, the dump in flight and its gulps, the open output span -- goes back to its ring."""
    C, S, g, acc = 2, 4, 2, 4
    vin = np.random.default_rng(5).integers(0, 256, (40, C, S, 2), dtype=np.uint8)
    iring, oring = Ring("in"), Ring("out")
    be = PumpOracleBackend(fail={entry: at})
    blk = Corr(LOG, iring, oring, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    sink = Sink(oring, blk.ogulp_size)
    err = []

    def main():
        try:
            blk.main()
        except Exception as e:
            err.append(e)
        finally:
            oring.begin_writing().__exit__(None, None, None)       # (what a pipeline's teardown does: downstream readers wake up)

    th = threading.Thread(target=main, daemon=True)
    src = Source(iring, [(source_header(C, S, 2), vin, g * C * S * 2)])
    sink.start(); th.start(); src.start()
    th.join(20)
    assert not th.is_alive() and len(err) == 1 and "returned %d" % be.ERR in str(err[0])
    # the source may be blocked on a full ring whose only reader died: let it go
    for gen in list(getattr(iring, "_open", [])):
        gen.close()
    committed = sum(len(sp) for _, _, sp in sink.sequences)
    assert committed <= (at - 1) // 2 if entry == "xgpu_try_kernel" else committed <= at
    assert getattr(be, "xsyncs", 0) >= 1 or at == 1                                               # waited for the X-engine before the spans went back
    st = None
    del blk
    import gc
    gc.collect()


# ====================================================================================================== BeamPump
def beam_chain(backend, vin, w, C, ninput, nbeam, g, ns, src_gulp=None, fail_ok=False):
    """gpu-input -> Beamform -> bf-output -> BeamformSumBeams -> bf-pow-output on system rings; returns (beam spans, power spans, blocks, rings)"""
    S = ninput // 2
    r0, r1, r2 = Ring("gpu-input"), Ring("bf-output"), Ring("bf-pow-output")
    bf = Beamform(LOG, r0, r1, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=backend)
    sb = BeamformSumBeams(LOG, r1, r2, nchan=C, ntime_gulp=g, ntime_sum=ns, backend=backend)
    bf.gains_cpu[...] = w
    beams, power = Sink(r1, g * C * nbeam * 8), Sink(r2, (nbeam // 2) * (g // ns) * C * 16)
    run_blocks([bf, sb], Source(r0, [(source_header(C, S, 2), vin, (src_gulp or g) * C * ninput)], wait_readers=1), [beams, power])
    return beams, power, (bf, sb), (r0, r1, r2)


@pytest.mark.parametrize("src_gulp", [None, 4])
def test_beam_chain_through_the_pumps_equals_the_oracle(src_gulp):
    """Beamform -> BeamformSumBeams through the native loops (gulps in one span, and gulps that lie in two spans of the input ring:
    one launch, no gathered copy), a short tail that is not a gulp at the end: every voltage gulp and every power gulp is the
    oracle's, the tail is passed by, and every span goes back."""
    C, ninput, nbeam, g, ns = 2, 16, 4, 8, 4
    rng = np.random.default_rng(8)
    ngulp = 7
    vin = rng.integers(0, 256, (ngulp * g + 3, C, ninput), dtype=np.uint8)         # (+ 3 samples: the short tail)
    w = (rng.uniform(-1, 1, (C, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (C, nbeam, ninput))).astype(np.complex64)
    be = PumpOracleBackend()
    beams, power, (bf, sb), rings = beam_chain(be, vin, w, C, ninput, nbeam, g, ns, src_gulp)
    assert be.callback_errors == []
    (_, _, bsp), = beams.sequences
    (_, _, psp), = power.sequences
    assert len(bsp) == ngulp and len(psp) == ngulp
    for k in range(ngulp):
        exp = orc.beamform(vin[k * g:(k + 1) * g], w, g, C, ninput, nbeam)
        assert np.array_equal(bsp[k].view(np.complex64).reshape(exp.shape), exp), k
        assert np.array_equal(psp[k].view(np.float32).reshape(-1), orc.beamform_integrate(exp, ns).ravel()), k
    assert be.calls.get("beam_run_parts", 0) == (ngulp if src_gulp else 0) and be.calls.get("beam_run_versioned", 0) == (0 if src_gulp else ngulp)
    assert be.calls["beam_integrate"] == ngulp and be.calls["beam_mark"] == 2 * ngulp
    del beams, power, bsp, psp
    import gc
    gc.collect()
    ring_balance(*rings)


def test_beam_pump_applies_a_coefficient_load_on_the_exact_gulp():
    """A `beamcoeffs` command that arrives while the pump waits for gulp 4 is applied to exactly gulp 4 (beamform_block.py:416-434:
    the flag is looked at before every gulp) -- the pump hands that gulp back with the flag up.  The source holds gulp 4 back
    until the command is in."""
    C, ninput, nbeam, g = 2, 16, 2, 4
    S = ninput // 2
    rng = np.random.default_rng(9)
    vin = rng.integers(0, 256, (8 * g, C, ninput), dtype=np.uint8)
    be = PumpOracleBackend()
    r0, r1 = Ring("gpu-input"), Ring("bf-output")
    bf = Beamform(LOG, r0, r1, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=be)
    w0 = (rng.uniform(-1, 1, (C, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (C, nbeam, ninput))).astype(np.complex64)
    bf.gains_cpu[...] = w0
    gate4 = threading.Event()
    hdr = source_header(C, S, 2, sfreq=1e6)
    src = GatedSource(r0, hdr, vin, g * C * ninput, {4: gate4})
    sink = Sink(r1, g * C * nbeam * 8)
    th = threading.Thread(target=bf.main, daemon=True)
    sink.start(); th.start(); src.start()
    wait_for(lambda: be.calls.get("beam_run_versioned", 0) == 4, "four gulps enqueued")
    delays, amps = rng.uniform(0, 12, ninput), rng.uniform(10, 17, ninput)
    bf.process_command_strings(cmd(1, coeffs={"type": "beamcoeffs", "beam_id": 1, "data": {"delays": delays.tolist(), "amps": amps.tolist()}}))
    gate4.set()
    for t in (src, th, sink):
        t.join(20)
        assert not t.is_alive()
    assert be.callback_errors == []
    (_, _, spans), = sink.sequences
    assert len(spans) == 8
    freqs = hdr['sfreq'] + (hdr['bw_hz'] / C) * np.arange(C)
    w1 = w0.copy()
    w1[:, 1, :] = (amps * np.exp(1j * 2 * np.pi * freqs[:, None] * delays * 1e-9) * bf.cal_gains[:, 1, :]).astype(np.complex64)
    for k in range(8):
        exp = orc.beamform(vin[k * g:(k + 1) * g], w0 if k < 4 else w1, g, C, ninput, nbeam)
        assert np.array_equal(spans[k].view(np.complex64).reshape(exp.shape), exp), k


@pytest.mark.parametrize("entry,at", [("beam_run_versioned", 4), ("beam_mark", 3), ("beam_wait", 2), ("beam_integrate", 3)])
def test_beam_pump_error_mid_flight_gives_every_span_back(entry, at):
    """A compute call that fails with gulps in flight: the block raises after waiting for its stream, the gulps in flight are NOT
    committed, and every span goes back to its ring (round-4 review: these paths ran on the GPU box only)."""
    C, ninput, nbeam, g, ns = 2, 16, 4, 8, 4
    S = ninput // 2
    rng = np.random.default_rng(10)
    vin = rng.integers(0, 256, (12 * g, C, ninput), dtype=np.uint8)
    w = (rng.uniform(-1, 1, (C, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (C, nbeam, ninput))).astype(np.complex64)
    be = PumpOracleBackend(fail={entry: at})
    r0, r1, r2 = Ring("gpu-input"), Ring("bf-output"), Ring("bf-pow-output")
    bf = Beamform(LOG, r0, r1, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=be)
    sb = BeamformSumBeams(LOG, r1, r2, nchan=C, ntime_gulp=g, ntime_sum=ns, backend=be)
    bf.gains_cpu[...] = w
    power = Sink(r2, (nbeam // 2) * (g // ns) * C * 16)
    errs = []

    def guarded(blk, oring):
        def run():
            try:
                blk.main()
            except Exception as e:
                errs.append((type(blk).__name__, e))
            finally:
                try:
                    oring.begin_writing().__exit__(None, None, None)
                except Exception:
                    pass
        return threading.Thread(target=run, daemon=True)

    ths = [guarded(bf, r1), guarded(sb, r2)]
    src = Source(r0, [(source_header(C, S, 2), vin, g * C * ninput)], wait_readers=1)
    power.start()
    for t in ths:
        t.start()
    src.start()
    for t in ths:
        t.join(20)
        assert not t.is_alive()
    failing = "BeamformSumBeams" if entry == "beam_integrate" else "Beamform"
    assert [n for n, _ in errs] == [failing] or (failing == "BeamformSumBeams" and [n for n, _ in errs][:1] == [failing])
    assert "returned %d" % be.ERR in str(errs[0][1])
    assert be.syncs >= 1                                   # the stream was waited for before any span went back
    npow = sum(len(sp) for _, _, sp in power.sequences)
    assert npow < 12


def test_beam_pump_reserve_failure_and_odd_part_sizes():
    """Driving the pump object directly: (1) the output sequence has ended when the pump reserves -- the call fails, the input gulp
    goes back, nothing is in flight afterwards; (2) a two-part gulp whose first part is not a whole number of samples is refused
    instead of being read from the wrong byte (an upstream writer that commits odd span sizes)."""
    C, ninput, nbeam, g = 2, 16, 2, 4
    row = C * ninput
    be = PumpOracleBackend()
    be.bfBeamformInitialize(0, ninput, C, g, nbeam, 0)
    w = np.ones((C, nbeam, ninput), dtype=np.complex64)
    # (1)
    r0, r1 = Ring("in"), Ring("out")
    gen = r0.read(guarantee=True)
    wr = r0.begin_writing()
    iseq_w = wr.begin_sequence(time_tag=0, header="{}")
    for k in range(3):
        with iseq_w.reserve(g * row) as sp:
            sp.data.numpy()[...] = k
    ow = r1.begin_writing()
    oseq = ow.begin_sequence(time_tag=0, header="{}")
    iseq = next(gen)
    pump = be.beam_pump(r0, iseq._rid, r1, oseq._seq_id, g * row, g * C * nbeam * 8, 0, row_bytes=row, depth=2)
    oseq.end()
    with pytest.raises(RuntimeError, match="xengRingReserve"):
        pump.run(w.ctypes.data, 1, 4, 0)
    assert be.calls.get("beam_run_versioned", 0) == 0 and be.calls.get("beam_sync", 0) >= 1
    del pump
    # (2)
    r2, r3 = Ring("in2"), Ring("out2")
    gen2 = r2.read(guarantee=True)
    w2 = r2.begin_writing()
    s2 = w2.begin_sequence(time_tag=0, header="{}")
    odd = g * row // 2 + 5                                   # not a multiple of the sample size
    for n in (odd, g * row - odd):
        with s2.reserve(n) as sp:
            sp.data.numpy()[...] = 1
    ow3 = r3.begin_writing()
    oseq3 = ow3.begin_sequence(time_tag=0, header="{}")
    iseq2 = next(gen2)
    pump2 = be.beam_pump(r2, iseq2._rid, r3, oseq3._seq_id, g * row, g * C * nbeam * 8, 0, row_bytes=row, depth=2)
    with pytest.raises(RuntimeError, match="whole number of samples"):
        pump2.run(w.ctypes.data, 1, 4, 0)
    assert be.calls.get("beam_run_parts", 0) == 0 and be.callback_errors == []
    del pump2


def test_beam_pump_on_a_slab_sequence_counts_time_across_skipped_slabs():
    """Beamform on a ring of PACKET SLABS whose reader registers late: the slabs it never saw move the sample count on, and the
    first gulp it does process is handed to the slab call with the sequence number of ITS first sample (BeamPump keeps `seq0` in
    step with what the ring skipped); the beams are the oracle's for the gulps that were processed."""
    from caltech_bifrost_dsp_amd.blocks import Snap2Ingest  # noqa: F401  (slab headers as the ingest block writes them)
    C, S, P, g, nbeam = 4, 16, 2, 8, 2
    ninput = S * P
    nslab = 12
    T = nslab * g
    rng = np.random.default_rng(21)
    vin = rng.integers(0, 256, (T, C, S, P), dtype=np.uint8)
    seq0, chan0 = 4000, 64
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=1, nchan_blocks=1, nstand_per_pkt=8, chan0_pipeline=chan0)
    per = len(pk) // nslab
    stride = len(pk[0])
    slab_bytes = per * stride
    be = PumpOracleBackend()
    r0, r1 = Ring("gpu-input-slabs"), Ring("bf-output")
    r0.resize(slab_bytes, 4 * slab_bytes)                      # room for four slabs: the early ones are overwritten before the block reads
    hdr = source_header(C, S, P, seq0=seq0, chan0=chan0, layout='snap2_slab', slab_ntime=g, npkt_per_gulp=per, pkt_stride=stride)
    w = (rng.uniform(-1, 1, (C, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (C, nbeam, ninput))).astype(np.complex64)
    wr = r0.begin_writing()
    oseq = wr.begin_sequence(time_tag=0, header=json.dumps(hdr))
    for k in range(8):                                         # eight slabs before anybody reads: four survive
        with oseq.reserve(slab_bytes) as sp:
            sp.data.numpy()[...] = np.frombuffer(b"".join(pk[k * per:(k + 1) * per]), dtype=np.uint8)
    bf = Beamform(LOG, r0, r1, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=2 * g, backend=be, guarantee=True)
    bf.gains_cpu[...] = w
    sink = Sink(r1, 2 * g * C * nbeam * 8)
    th = threading.Thread(target=bf.main, daemon=True)
    sink.start(); th.start()
    for k in range(8, nslab):
        with oseq.reserve(slab_bytes) as sp:
            sp.data.numpy()[...] = np.frombuffer(b"".join(pk[k * per:(k + 1) * per]), dtype=np.uint8)
    oseq.end()
    wr.__exit__(None, None, None)
    for t in (th, sink):
        t.join(20)
        assert not t.is_alive()
    assert be.callback_errors == []
    (_, _, spans), = sink.sequences
    # slabs 0..3 were overwritten (4 slabs = 2 beamformer gulps skipped): gulps of slabs (4,5) (6,7) (8,9) (10,11)
    assert be.slab_seq0 == [seq0 + 4 * g + 2 * g * k for k in range(4)]
    assert len(spans) == 4
    for k, sp in enumerate(spans):
        t0 = 4 * g + 2 * g * k
        exp = orc.beamform(vin[t0:t0 + 2 * g].reshape(2 * g, C, ninput), w, 2 * g, C, ninput, nbeam)
        assert np.array_equal(sp.view(np.complex64).reshape(exp.shape), exp), k


def test_sum_beams_pump_staged_copies_commit_in_order():
    """BeamformSumBeams' pump with a staged output (the pinned-host ring of the pipeline: the kernel writes a device buffer, a copy
    carries it to the span, the span is committed when the copy is done): every power gulp arrives, in order, equal to the
    oracle's; the staging buffers are reused, not leaked."""
    C, nbeam, g, ns = 2, 4, 8, 4
    rng = np.random.default_rng(31)
    ngulp = 9
    beams = (rng.normal(size=(ngulp, C, nbeam, g)) + 1j * rng.normal(size=(ngulp, C, nbeam, g))).astype(np.complex64)
    be = PumpOracleBackend()
    be.bfBeamformInitialize(0, 16, C, g, nbeam, 0)
    igulp, ogulp = C * nbeam * g * 8, (nbeam // 2) * (g // ns) * C * 16
    r0, r1 = Ring("bf-output"), Ring("bf-pow-output")
    r0.resize(igulp, (ngulp + 1) * igulp)          # (everything is written before the pump reads: room for all of it)
    r1.resize(ogulp, (ngulp + 1) * ogulp)
    gen = r0.read(guarantee=True)
    sink = Sink(r1, ogulp)
    sink.start()
    wr = r0.begin_writing()
    iseq_w = wr.begin_sequence(time_tag=0, header=json.dumps({}))
    for k in range(ngulp):
        with iseq_w.reserve(igulp) as sp:
            sp.data.numpy()[...] = beams[k].reshape(-1).view(np.uint8)
    iseq_w.end()
    wr.__exit__(None, None, None)
    ow = r1.begin_writing()
    oseq = ow.begin_sequence(time_tag=0, header=json.dumps({}))
    iseq = next(gen)
    pump = be.beam_pump(r0, iseq._rid, r1, oseq._seq_id, igulp, ogulp, 1, ntime_sum=ns, depth=3, staged=True)
    total = 0
    while True:
        n, skipped, status = pump.run(0, 0, 4, 0)
        total += n
        if status == 1:
            break
    assert total == ngulp and be.callback_errors == []
    oseq.end()
    ow.__exit__(None, None, None)
    sink.join(10)
    (_, _, spans), = sink.sequences
    assert len(spans) == ngulp
    for k in range(ngulp):
        assert np.array_equal(spans[k].view(np.float32), orc.beamform_integrate(beams[k], ns).ravel()), k
    assert be.calls["memcpy_async"] == ngulp and be.calls["dev_malloc"] <= 3 + 3 + 1        # (depth in flight + copies in flight, reused)
    del pump
    assert be.calls.get("dev_free", 0) == be.calls["dev_malloc"] and not be._staged
