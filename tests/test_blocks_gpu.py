"""The blocks on device-space rings with the real HIP backend (libxeng): Corr -> CorrAcc and
Beamform -> BeamformSumBeams as threads, one HIP stream each, against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, Corr, CorrAcc  # noqa: E402
from caltech_bifrost_dsp_amd.ring import Ring  # noqa: E402
from oracle import xeng_oracle as orc  # noqa: E402
from tests.pipeline_util import LOG, Sink, Source, run_blocks, source_header  # noqa: E402
from tests.test_blocks_cpu import _beam_cmds  # noqa: E402


def fused_xengine_expected():
    """The shipped switch XENG_RAW=0 selects the two-pass X-engine (identical results, no fused long accumulation): the path
    assertions below follow the switch the suite was started with."""
    import os
    return os.environ.get("XENG_RAW") != "0"


@pytest.mark.parametrize("mode", ["group", "fused", "map"])
@pytest.mark.parametrize("g", [32, 96])
def test_corr_corracc_on_device_rings(g, mode):
    """gpu-input (cuda) -> Corr -> corr-output (cuda) -> CorrAcc -> corr-slow-output (cuda_host),
    the ring spaces of lwa352-pipeline.py:147-155; bit-exact vs the oracle, in the three ways CorrAcc can accumulate: the
    spans of a group of dumps summed in one pass (the default), the add fused into the dumps' epilogue (gulps of 96 samples
    run the default contraction kernel, which can do that; gulps of 32 take the two-pass X-engine and CorrAcc's own map), and
    the reference's map per span."""
    C, S, acc, lacc = 8, 48, 2 * g, 4 * g
    rng = np.random.default_rng(5)
    vin = rng.integers(0, 256, (4 * lacc // 2, C, S, 2), dtype=np.uint8)      # 256 samples = 2 long integrations
    r0, r1, r2 = Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda"), Ring("corr-slow-output", space="cuda_host")
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, gpu=0, test=True)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, gpu=0, accumulate=mode)
    fast = Sink(r1, corr.ogulp_size)
    slow = Sink(r2, cacc.ogulp_size)
    run_blocks([corr, cacc], Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)], wait_readers=1), [fast, slow])
    (h1, _, sp1), = fast.sequences
    assert len(sp1) == 4 and h1['acc_len'] == acc
    for k, sp in enumerate(sp1):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[k * acc:(k + 1) * acc], S, C))
    assert corr.stats['test_match'] is True
    (h2, _, sp2), = slow.sequences
    assert h2['upstream_acc_len'] == acc and h2['acc_len'] == lacc and len(sp2) == 2
    for k, sp in enumerate(sp2):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[k * lacc:(k + 1) * lacc], S, C))
    fz = mode == "fused" and g == 96 and fused_xengine_expected()
    assert cacc.stats['fused'] is fz and cacc.fused_dumps == (4 if fz else 0)   # accumulated by the dumps' own epilogue
    assert cacc.stats['grouped'] is (mode != "map" and not fz)                   # (a fused CorrAcc whose Corr cannot fuse sums groups)


@pytest.mark.parametrize("fused", [True, False, "group"])
def test_corracc_full_size_misaligned_start(fused):
    """BASELINE config 2 size (704 inputs, 96 channels, acc_len 2400 = 5 x 480): CorrAcc starts in the middle of the
    stream (start_time = one upstream integration, not 0) with three dumps per long integration -- two in one partial
    accumulator, one in the other.  Every published slow span equals 3 x the stand-alone integration, bit for bit (the
    stream repeats one integration, whose visibilities tests/test_xcorr_gpu.py::test_config2_full_size holds to the
    oracle), with the add fused into the dumps and with the classic map."""
    import threading
    from tests.gpu_util import Xgpu
    C, S, g, acc, nrep = 96, 352, 480, 2400, 8
    G, gulp_bytes = acc // g, g * C * S * 2
    rng = np.random.default_rng(2024)
    vin = rng.integers(0, 256, (acc, C, S, 2), dtype=np.uint8)
    x = Xgpu(S, C, g, max_gulps=G)
    ref = x.run(vin, use_async=True).astype(np.int64)
    x.close()
    r0, r1, r2 = Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda"), Ring("corr-slow-output", space="cuda_host")
    r0.resize(gulp_bytes, total_span=2 * G * gulp_bytes)
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, gpu=0)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=3 * acc, autostartat=acc, gpu=0,
                   accumulate="fused" if fused is True else "map" if fused is False else fused)
    if fused == "group":
        cacc.group_dumps = 2              # (three dumps per long integration: one group of two, one of one)
    verdicts, hdrs = [], []
    want = (3 * ref).astype(np.int32).reshape(-1)

    def slow_sink(gen=r2.read(guarantee=True)):
        import json
        for iseq in gen:
            hdrs.append(json.loads(iseq.header.tostring()))
            for ispan in iseq.read(cacc.ogulp_size):
                if ispan.size == cacc.ogulp_size:
                    verdicts.append(bool(np.array_equal(ispan.data.numpy().view(np.int32).reshape(-1), want)))

    sink = threading.Thread(target=slow_sink, daemon=True)
    sink.start()
    data = np.tile(vin.reshape(-1), nrep)
    run_blocks([corr, cacc], Source(r0, [(source_header(C, S, 2), data, gulp_bytes)], wait_readers=1), [], timeout=240)
    sink.join(60)
    assert not sink.is_alive()
    assert verdicts == [True, True], verdicts                  # [2400, 9600) and [9600, 16800); the third is cut off
    assert hdrs[0]['seq0'] == acc and hdrs[0]['acc_len'] == 3 * acc and hdrs[0]['upstream_acc_len'] == acc
    fz = fused is True and fused_xengine_expected()
    assert cacc.stats['fused'] is fz and (cacc.fused_dumps == 7) is fz
    assert cacc.stats['grouped'] is (fused == "group" or (fused is True and not fz))


def test_beamform_sumbeams_on_device_rings():
    nchan, nstand, nbeam, g, ns = 4, 32, 8, 96, 24
    ninput = nstand * 2
    rng = np.random.default_rng(0xaabbccdd)
    vin = rng.integers(0, 256, (2 * g, nchan, ninput), dtype=np.uint8)
    r0, r1, r2 = Ring("gpu-input", space="cuda"), Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, gpu=0)
    sb = BeamformSumBeams(LOG, r1, r2, nchan=nchan, ntime_gulp=g, ntime_sum=ns, gpu=0)
    sfreq, bw = 40e6, 23925.78125
    bf.freqs = sfreq + bw * np.arange(nchan)
    cmds, cal, delays, amps = _beam_cmds(nchan, nbeam, ninput, rng)
    bf.process_command_strings(cmds)
    s1, s2 = Sink(r1, g * nchan * nbeam * 8), Sink(r2, (nbeam // 2) * (g // ns) * nchan * 16)
    run_blocks([bf, sb], Source(r0, [(source_header(nchan, nstand, 2, sfreq=sfreq, chan_bw=bw), vin, g * nchan * ninput)]), [s1, s2])
    (h1, _, sp1), = s1.sequences
    (h2, _, sp2), = s2.sequences
    assert len(sp1) == 2 and len(sp2) == 2 and h2['nbeam'] == nbeam // 2
    for k in range(2):
        exp = orc.beamform(vin[k * g:(k + 1) * g], bf.gains_cpu, g, nchan, ninput, nbeam)
        got = sp1[k].view(np.complex64).reshape(exp.shape)
        assert np.max(np.abs(got - exp)) / np.sqrt(np.mean(np.abs(exp) ** 2)) <= 1e-5
        pexp = orc.beamform_integrate(got, ns)
        pgot = sp2[k].view(np.float32).reshape(pexp.shape)
        assert np.all(np.isclose(pgot, pexp, rtol=1e-5, atol=1e-5 * np.abs(pexp).max()))


@pytest.mark.parametrize("pump", ["1", "0"])
def test_beamform_timed_coefficient_load_on_device_rings(pump, monkeypatch):
    """Forty gulps through Beamform -> BeamformSumBeams with coefficients that load at gulp 11 (beamform_block.py:416-429), on
    the native per-gulp loop (csrc/pyext/xfast.cpp BeamPump, which hands control back to the block when a load is pending) and
    on the Python loop (XENG_PUMP=0): zero beams before the load sample, the commanded ones from it on, every gulp against the
    oracle, power sums included; a command that arrives WHILE the pipeline runs takes effect at its load sample too.  No timing:
    the source holds gulp 20 back until that command is in, so it lands between gulps 12 and 33 by construction."""
    import threading
    import time
    monkeypatch.setenv("XENG_PUMP", pump)
    nchan, nstand, nbeam, g, ns, ngulp = 4, 32, 8, 96, 24, 40
    ninput = nstand * 2
    rng = np.random.default_rng(0xbeef)
    vin = rng.integers(0, 256, (ngulp * g, nchan, ninput), dtype=np.uint8)
    r0, r1, r2 = Ring("gpu-input", space="cuda"), Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, gpu=0)
    sb = BeamformSumBeams(LOG, r1, r2, nchan=nchan, ntime_gulp=g, ntime_sum=ns, gpu=0)
    sfreq, bw = 40e6, 23925.78125
    bf.freqs = sfreq + bw * np.arange(nchan)
    cmds, cal, delays, amps = _beam_cmds(nchan, nbeam, ninput, rng, load_sample=11 * g)
    bf.process_command_strings(cmds)
    first = bf.gains_cpu_new.copy()
    # a second set of coefficients, commanded from another thread once the pipeline is running, to load at gulp 33 (far enough ahead for 500 command strings to be parsed on a busy host)
    cmds2, _, _, _ = _beam_cmds(nchan, nbeam, ninput, np.random.default_rng(99), load_sample=33 * g)
    second = {}

    from tests.pipeline_util import GatedSource, wait_for
    gate20 = threading.Event()
    src = GatedSource(r0, source_header(nchan, nstand, 2, sfreq=sfreq, chan_bw=bw), vin, g * nchan * ninput, {20: gate20})

    def late_command():
        wait_for(lambda: src.written == 20, "the source to have written gulps 0..19")       # (gulp 20 waits for the gate below)
        bf.process_command_strings(cmds2)
        second['gains'] = bf.gains_cpu_new.copy()
        second['at'] = bf.stats.get('curr_sample', -1)
        gate20.set()
    s1, s2 = Sink(r1, g * nchan * nbeam * 8), Sink(r2, (nbeam // 2) * (g // ns) * nchan * 16)
    th = threading.Thread(target=late_command, daemon=True)
    th.start()
    run_blocks([bf, sb], src, [s1, s2])
    th.join(20)
    (h1, _, sp1), = s1.sequences
    (h2, _, sp2), = s2.sequences
    assert len(sp1) == ngulp and len(sp2) == ngulp
    assert 'gains' in second and second['at'] < 20 * g, second.get('at')        # (commanded before gulp 20 was even written; loads at gulp 33)
    zero = np.zeros_like(first)
    for k in range(ngulp):
        w = zero if k < 11 else first if k < 33 else second['gains']
        exp = orc.beamform(vin[k * g:(k + 1) * g], w, g, nchan, ninput, nbeam)
        got = sp1[k].view(np.complex64).reshape(exp.shape)
        assert np.max(np.abs(got - exp)) <= 1e-5 * max(np.sqrt(np.mean(np.abs(exp) ** 2)), 1e-30), k
        pexp = orc.beamform_integrate(got, ns)
        pgot = sp2[k].view(np.float32).reshape(pexp.shape)
        assert np.all(np.isclose(pgot, pexp, rtol=1e-5, atol=1e-5 * max(np.abs(pexp).max(), 1e-30))), k


def test_ingest_copy_corr_subsel_chain():
    """The ingest side and the fast-visibility side of Corr: pinned host ring
    -> Copy (H2D) -> gpu-input (cuda) -> Corr -> corr-output (cuda) -> CorrSubsel -> cuda_host ring."""
    from caltech_bifrost_dsp_amd.blocks import Copy, CorrSubsel
    C, S, g, acc, nvis, nsum = 8, 48, 32, 64, 40, 4
    rng = np.random.default_rng(21)
    vin = rng.integers(0, 256, (2 * acc, C, S, 2), dtype=np.uint8)
    r_host = Ring("capture", space="cuda_host")
    r_in, r_vis, r_fast = Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda"), Ring("corr-fast-output", space="cuda_host")
    hdr = source_header(C, S, 2, seq0=0)
    cp = Copy(LOG, r_host, r_in, ntime_gulp=g, nbyte_per_time=C * S * 2, gpu=0)
    corr = Corr(LOG, r_in, r_vis, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, gpu=0,
                ant_to_input=hdr['ant_to_input'])
    sub = CorrSubsel(LOG, r_vis, r_fast, nchan=C, npol=2, nstand=S, nchan_sum=nsum, gpu=0, nvis_out=nvis,
                     antpol_to_bl=corr.antpol_to_bl.numpy(), bl_is_conj=corr.bl_is_conj.numpy())
    sel = [[[int(a), int(b)], [int(c), int(d)]] for a, b, c, d in
           zip(rng.integers(0, S, nvis), rng.integers(0, 2, nvis), rng.integers(0, S, nvis), rng.integers(0, 2, nvis))]
    sub.process_command_strings('{"cmd": "update", "val": {"kwargs": {"baselines": %s}}, "id": "1"}' % str(sel).replace("'", '"'))
    sink = Sink(r_fast, sub.ogulp_size)
    run_blocks([cp, corr, sub], Source(r_host, [(hdr, vin, g * C * S * 2)]), [sink])
    (ohdr, _, spans), = sink.sequences
    assert len(spans) == 2 and ohdr['nvis'] == nvis
    bl, cj = orc.xgpu_get_order(np.arange(S * 2, dtype=np.int32).reshape(S, 2))
    vismap = np.array([bl[s0, s1, p0, p1] for (s0, p0), (s1, p1) in sel], np.int32)
    conj = np.array([cj[s0, s1, p0, p1] for (s0, p0), (s1, p1) in sel], np.int32)
    for k, sp in enumerate(spans):
        planar = orc.xgpu_correlate(vin[k * acc:(k + 1) * acc], S, C)
        exp = orc.xgpu_subselect(planar, vismap, conj, C, nsum, S)
        assert np.array_equal(sp.view(np.int32).reshape(exp.shape), exp)


@pytest.mark.parametrize("host_ring", [False, True])
def test_corr_corracc_output_full_chain(tmp_path, host_ring):
    """The slow-visibility side: Corr -> CorrAcc -> CorrOutputFull with the device
    packetiser (xengXgpuPacketize).  CorrAcc publishes into a cuda_host ring as in lwa352-pipeline.py:154
    (the block then stages the span on the device) or into a cuda ring (read in place).  The block's golden
    check (corr_output_full_block.py:550-603) runs against visibilities computed independently in numpy."""
    from caltech_bifrost_dsp_amd.blocks import CorrOutputFull
    C, S, g, acc, lacc = 8, 20, 32, 64, 128
    rng = np.random.default_rng(9)
    vin = rng.integers(0, 256, (2 * lacc, C, S, 2), dtype=np.uint8)
    # golden [t][chan][s0][s1][p0][p1] = sum x[s0,p0] conj(x[s1,p1])  (make_golden_inputs.py:156-158)
    re, im = orc.decode(vin)
    x = (re + 1j * im).reshape(2, lacc, C, S * 2)
    gold = np.einsum('ktci,ktcj->kcij', x, np.conj(x)).reshape(2, C, S, 2, S, 2).transpose(0, 1, 2, 4, 3, 5)
    check = tmp_path / "golden_raw.bin"
    check.write_bytes(np.ascontiguousarray(gold.astype(np.complex128)).tobytes())
    r0, r1 = Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda")
    r2 = Ring("corr-slow-output", space="cuda_host" if host_ring else "cuda")
    hdr = source_header(C, S, 2, seq0=0, sync_time=7)
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, gpu=0,
                ant_to_input=hdr['ant_to_input'])
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, gpu=0)
    pkts = []
    out = CorrOutputFull(LOG, r2, nchan=C, npol=2, nstand=S, checkfile=str(check), checkfile_acc_len=lacc,
                         antpol_to_bl=corr.antpol_to_bl.numpy(), bl_is_conj=corr.bl_is_conj.numpy(),
                         use_cor_fmt=False, gpu=0, sink=pkts.append)
    run_blocks([corr, cacc, out], Source(r0, [(hdr, vin, g * C * S * 2)], wait_readers=1), [])
    nbl = S * (S + 1) // 2
    assert out.check_results == [(nbl * 8, 0)] * 2
    assert len(pkts) == 2 * nbl and all(len(p) == 56 + 4 * C * 8 for p in pkts)
    # spot-check one packet's payload against the golden matrix
    k = nbl + (3 * S - (3 * 2) // 2 + (11 - 3))              # second integration, baseline (3, 11): s0*S - s0(s0-1)/2 + s1 - s0
    pay = np.frombuffer(pkts[k][56:], dtype=np.int32).reshape(2, 2, C, 2)
    assert np.array_equal(pay[..., 0], np.moveaxis(gold[1, :, 3, 11].real, 0, -1))
    assert np.array_equal(pay[..., 1], np.moveaxis(gold[1, :, 3, 11].imag, 0, -1))


def test_corr_reads_gulps_in_place_from_the_in_repo_ring():
    """Corr on the in-repo ring (span memory stays alive while referenced): gulps are handed to the X-engine
    with the enqueue-only call and read in place by the fused kernel -- no raw copy, no corner turn -- and the
    visibilities are the oracle's.  A sequence that ends mid-integration drops the partial sums."""
    import ctypes
    from caltech_bifrost_dsp_amd import ffi
    C, S, g, acc = 8, 32, 96, 288
    rng = np.random.default_rng(17)
    vin = rng.integers(0, 256, (2 * acc + g, C, S, 2), dtype=np.uint8)   # two integrations + one stray gulp
    r0, r1 = Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda")
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, gpu=0)
    fused, fp6 = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengXgpuGetPath", ctypes.byref(fused), ctypes.byref(fp6))
    if not fused_xengine_expected():
        pytest.skip("XENG_RAW=0: the two-pass X-engine copies its gulps (this test is about the fused kernel reading in place)")
    assert fused.value == 1
    ffi.call("xengXgpuSetProfiling", 1)
    tm, cn = (ctypes.c_double * 2)(), (ctypes.c_int * 2)()
    ffi.call("xengXgpuGetTimes", tm, cn)                                  # clear
    sink = Sink(r1, corr.ogulp_size)
    run_blocks([corr], Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)], wait_readers=1), [sink])
    ffi.call("xengXgpuGetTimes", tm, cn)
    ffi.call("xengXgpuSetProfiling", 0)
    assert cn[0] == 0 and cn[1] == 2                                      # no copies / corner turns; two contractions
    (h, _, spans), = sink.sequences
    assert len(spans) == 2
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[k * acc:(k + 1) * acc], S, C))


def test_full_topology_packets_in_packets_out():
    """The hot path and its neighbours wired as scripts/lwa352-pipeline.py:147-155,232-285 wires them, all blocks as
    concurrent threads on one GPU, each on its own HIP stream:

        F-engine packets -> Snap2Ingest -> gpu-input (cuda, two readers)
            |- Corr -> corr-output -> CorrAcc -> corr-slow-output (cuda_host) -> CorrOutputFull -> packets
            |- Beamform -> bf-output -> BeamformSumBeams -> bf-pow-output (cuda_host) -> BeamformOutput -> packets

    Every visibility packet and every power-beam packet is checked against the oracle."""
    import struct
    import threading
    import time
    from caltech_bifrost_dsp_amd.blocks import BeamformOutput, CorrOutputFull, Snap2Ingest
    C, S, g, acc, lacc, nbeam, ns = 8, 32, 96, 192, 384, 4, 24
    ninput = S * 2
    T = 2 * lacc
    rng = np.random.default_rng(2024)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    seq0, chan0 = 7680, 96
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=11, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=chan0)
    per_win = len(pk) // (T // g)
    slabs = b"".join(b"".join(pk[w * per_win + i] for i in rng.permutation(per_win)) for w in range(T // g))
    r_pk = Ring("packets", space="cuda_host")
    r_in = Ring("gpu-input", space="cuda")
    r_vis, r_slow = Ring("corr-output", space="cuda"), Ring("corr-slow-output", space="cuda_host")
    r_bf, r_pow = Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    ing = Snap2Ingest(LOG, r_pk, r_in, ntime_gulp=g, nchan=C, nstand=S, npol=2, nchan_per_pkt=C, nstand_per_pkt=32, gpu=0,
                      buffer_multiplier=T // g + 1, system_nchan=C * 4)
    hdr = source_header(C, S, 2)
    corr = Corr(LOG, r_in, r_vis, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=seq0, gpu=0,
                ant_to_input=hdr['ant_to_input'])
    cacc = CorrAcc(LOG, r_vis, r_slow, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=seq0, gpu=0)
    vis_pk, beam_pk = [], []
    cout = CorrOutputFull(LOG, r_slow, nchan=C, npol=2, nstand=S, antpol_to_bl=corr.antpol_to_bl.numpy(),
                          bl_is_conj=corr.bl_is_conj.numpy(), use_cor_fmt=False, gpu=0, sink=vis_pk.append)
    bf = Beamform(LOG, r_in, r_bf, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=g, gpu=0)
    sb = BeamformSumBeams(LOG, r_bf, r_pow, nchan=C, ntime_gulp=g, ntime_sum=ns, gpu=0)
    bout = BeamformOutput(LOG, r_pow, ntime_gulp=g // ns, pipeline_idx=1, nchan=C, nbeam=nbeam // 2,
                          sink=lambda b, p: beam_pk.append((b, p)))
    bf.freqs = chan0 * 23925.78125 + 23925.78125 * np.arange(C)
    cmds, cal, delays, amps = _beam_cmds(C, nbeam, ninput, rng)
    bf.process_command_strings(cmds)
    blocks = [ing, corr, cacc, cout, bf, sb, bout]
    ths = [threading.Thread(target=b.main, daemon=True) for b in blocks]
    for t in ths:
        t.start()
    t0 = time.time()
    while len(r_in._readers) < 2 and time.time() - t0 < 20:      # both consumers attached before data flows
        time.sleep(0.01)
    src = Source(r_pk, [({'seq0': seq0, 'chan0': chan0, 'sync_time': 11}, slabs, ing.igulp_size)], wait_readers=1)
    src.start()
    for t in [src] + ths:
        t.join(120)
        assert not t.is_alive(), "pipeline thread did not finish: %r" % (t,)
    # --- slow visibilities: two long integrations, one packet per baseline
    nbl = S * (S + 1) // 2
    assert len(vis_pk) == 2 * nbl
    bl, cj = corr.antpol_to_bl.numpy(), corr.bl_is_conj.numpy()
    for it in range(2):
        planar = orc.xgpu_correlate(vin[it * lacc:(it + 1) * lacc], S, C)
        pay = orc.corr_packet_payloads(orc.xgpu_reorder(planar, bl, cj, C), False)
        for k in range(nbl):
            p = vis_pk[it * nbl + k]
            assert np.array_equal(np.frombuffer(p[56:], dtype=np.int32), pay[k]), (it, k)
        f = struct.unpack(">QQ2d4I2I", vis_pk[it * nbl][:56])
        assert f[0] == 11 and f[1] == seq0 + it * lacc and f[4:8] == (lacc, C, chan0, 2)
    # --- power beams: one packet per (dual-pol beam, integrated sample)
    nblk = g // ns
    assert len(beam_pk) == (T // g) * (nbeam // 2) * nblk
    k = 0
    for sp in range(T // g):
        beams = orc.beamform(vin[sp * g:(sp + 1) * g].reshape(g, C, ninput), bf.gains_cpu, g, C, ninput, nbeam)
        power = orc.beamform_integrate(beams, ns)
        for b in range(nbeam // 2):
            for t in range(nblk):
                bb, p = beam_pk[k]
                k += 1
                got = np.frombuffer(p[18:], dtype=np.float32).reshape(C, 4)
                assert bb == b and np.allclose(got, power[b, t], rtol=2e-5, atol=2e-5 * np.abs(power).max())
    assert ing.stats['packets_placed'] == len(pk)
