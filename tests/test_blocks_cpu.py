"""Host logic of the hot-path blocks on CPU rings (no GPU): BASELINE config 1 -- "16-ant x 2pol,
4 chan, correlate+accumulate on a CPU ring vs numpy (pipeline/verification)".  The compute calls
go to tests/fake_backend.OracleBackend (the CPU oracle behind the backend interface); what is
under test here is the Block/ring/state-machine code the product ships."""
import json
import logging
import os

import numpy as np
import pytest

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd.blocks import (Beamform, BeamformSumBeams, Block, Corr, CorrAcc, regtile_index,
                                            COMMAND_INVALID, COMMAND_NOT_RECOGNIZED, COMMAND_OK, COMMAND_WRONG_TYPE)
from caltech_bifrost_dsp_amd.ring import Ring
from oracle import xeng_oracle as orc
from tests.fake_backend import OracleBackend
from tests.pipeline_util import LOG, Sink, Source, run_blocks, source_header


def load_dat(path):
    with open(path, "rb") as fh:
        meta = json.loads(fh.readline().decode())
        raw = fh.read()
    dt = np.uint8 if "uint8" in meta["dtype"] else np.complex128
    return meta, np.frombuffer(raw, dtype=dt).reshape(meta["shape"])


def cmd(idx, **kwargs):
    return json.dumps({"cmd": "update", "val": {"kwargs": kwargs}, "id": str(idx)})


# ----------------------------------------------------------------------------------------------
def test_regtile_index_matches_oracle():
    for ns in (16, 352):
        rng = np.random.default_rng(ns)
        for _ in range(500):
            i0, i1 = sorted(int(v) for v in rng.integers(0, 2 * ns, 2))
            assert regtile_index(i0, i1, ns) == orc.regtile_index(i0, i1, ns)


@pytest.mark.parametrize("tag", ["deadbeef", "chanramp"])
@pytest.mark.parametrize("ntime_gulp", [4, 2])
def test_config1_corr_on_cpu_ring_vs_golden(golden_dir, tag, ntime_gulp):
    """Golden input replayed through Corr on a system-space ring; the xGPU-order output spans,
    reordered with the block's own antpol_to_bl / bl_is_conj maps, equal the reference's golden
    visibilities exactly (the check of corr_output_full_block.py:550-603)."""
    _, vin = load_dat(os.path.join(golden_dir, "in_8t_4c_16s_2p_%s.dat" % tag))
    meta, corr = load_dat(os.path.join(golden_dir, "corr_8t_4a_4c_16s_2p_%s.dat" % tag))
    T, C, S, P = vin.shape
    acc_len = meta["acc_len"]
    iring, oring = Ring("gpu-input"), Ring("corr-output")
    be = OracleBackend()
    hdr = source_header(C, S, P, seq0=0)
    blk = Corr(LOG, iring, oring, ntime_gulp=ntime_gulp, nchan=C, npol=P, nstand=S, acc_len=acc_len,
               autostartat=0, ant_to_input=hdr['ant_to_input'], backend=be, test=True)
    sink = Sink(oring, blk.ogulp_size)
    run_blocks([blk], Source(iring, [(hdr, vin, ntime_gulp * C * S * P)]), [sink])
    assert len(sink.sequences) == 1
    ohdr, time_tag, spans = sink.sequences[0]
    assert time_tag == 1 and ohdr['seq0'] == 0 and ohdr['acc_len'] == acc_len
    assert 'ant_to_input' not in ohdr and 'input_to_ant' not in ohdr and ohdr['nchan'] == C
    assert len(spans) == T // acc_len
    assert be.kernel_calls == ([0] * (acc_len // ntime_gulp - 1) + [1]) * (T // acc_len)
    bl, cj = blk.antpol_to_bl.numpy(), blk.bl_is_conj.numpy()
    for k, sp in enumerate(spans):
        ro = orc.xgpu_reorder(sp.view(np.int32), bl, cj, C)
        for s0 in range(S):
            for s1 in range(s0, S):
                g = corr[k, :, s0, s1]
                assert np.array_equal(ro[s0, s1, :, :, :, 0], np.moveaxis(g.real, 0, -1).astype(np.int32))
                assert np.array_equal(ro[s0, s1, :, :, :, 1], np.moveaxis(g.imag, 0, -1).astype(np.int32))
    assert blk.stats['test_match'] is True           # the block's own --testcorr self-check (corr_block.py:265-315)
    assert blk.stats['state'] == 'running' and blk.stats['last_end_sample'] == T - ntime_gulp


def _corr_scenario(nseq_gulps, ntime_gulp=2, acc_len=4, autostartat=0, seq0s=None, commands=None, C=2, S=4):
    """Run Corr over sequences of random data; returns (sink sequences, backend, per-sequence inputs)."""
    rng = np.random.default_rng(1)
    iring, oring = Ring("in"), Ring("out")
    be = OracleBackend()
    blk = Corr(LOG, iring, oring, ntime_gulp=ntime_gulp, nchan=C, npol=2, nstand=S, acc_len=acc_len,
               autostartat=autostartat, backend=be)
    for c in (commands or []):
        blk.process_command_strings(c)
    seqs, data = [], []
    for k, ng in enumerate(nseq_gulps):
        seq0 = (seq0s[k] if seq0s else 0)
        d = rng.integers(0, 256, (ng * ntime_gulp, C, S, 2), dtype=np.uint8)
        data.append(d)
        seqs.append((source_header(C, S, 2, seq0=seq0), d, ntime_gulp * C * S * 2))
    sink = Sink(oring, blk.ogulp_size)
    run_blocks([blk], Source(iring, seqs), [sink])
    return sink.sequences, be, data, blk


def test_corr_start_time_in_future_and_short_tail():
    # start at sample 6; 11 gulps of 2 (samples 0..21) plus a short tail -> integrations [6,10) [10,14) [14,18) [18,22)
    seqs, be, data, blk = _corr_scenario([11], autostartat=6)
    (ohdr, tag, spans), = seqs
    assert ohdr['seq0'] == 6 and ohdr['acc_len'] == 4 and len(spans) == 4
    for k, sp in enumerate(spans):
        exp = orc.xgpu_correlate(data[0][6 + 4 * k:10 + 4 * k], 4, 2)
        assert np.array_equal(sp.view(np.int32), exp)


def test_corr_start_minus_one_rounds_up_to_acc_len():
    # corr_block.py:397-398: -1 -> next multiple of acc_len after the current gulp; seq0 = 2 -> start at 4
    seqs, be, data, blk = _corr_scenario([8], autostartat=-1, seq0s=[2])
    (ohdr, tag, spans), = seqs
    assert ohdr['seq0'] == 4
    assert np.array_equal(spans[0].view(np.int32), orc.xgpu_correlate(data[0][2:6], 4, 2))   # samples 4..7 = rows 2..5


def test_corr_recovery_after_new_sequence_skips_ten_integrations():
    # corr_block.py:360-371: new upstream sequence while running -> start = last_start + (missed+10)*acc_len
    seqs, be, data, blk = _corr_scenario([4, 30], seq0s=[0, 8])
    assert [s[0]['seq0'] for s in seqs] == [0, 0 + (8 // 4 + 10) * 4]
    assert [s[1] for s in seqs] == [1, 2]                    # time_tag counts output sequences
    assert len(seqs[0][2]) == 2
    first_new = (48 - 8)                                       # row offset of sample 48 in sequence 2
    assert np.array_equal(seqs[1][2][0].view(np.int32), orc.xgpu_correlate(data[1][first_new:first_new + 4], 4, 2))


def test_corr_command_validation_and_update():
    iring, oring = Ring("i"), Ring("o")
    blk = Corr(LOG, iring, oring, ntime_gulp=2, nchan=2, npol=2, nstand=4, acc_len=4, backend=OracleBackend())
    blk.process_command_strings(cmd(1, acc_len=3))            # not a multiple of the gulp
    assert blk.stats['last_cmd_response'] == COMMAND_INVALID and blk.last_response['val']['status'] == 'error'
    blk.process_command_strings(cmd(2, acc_len="8"))
    assert blk.stats['last_cmd_response'] == COMMAND_WRONG_TYPE
    blk.process_command_strings(cmd(3, bogus=1))
    assert blk.stats['last_cmd_response'] == COMMAND_NOT_RECOGNIZED
    blk.process_command_strings(json.dumps({"cmd": "nope", "val": {}, "id": "4"}))
    assert blk.last_response['val']['response'] == "Invalid command"
    blk.process_command_strings(cmd(5, acc_len=8, start_time=-1))
    assert blk.stats['last_cmd_response'] == COMMAND_OK and blk.update_pending
    assert blk.command_vals['acc_len'] == 4                   # active values only change in main()
    blk.update_command_vals()
    assert blk.command_vals['acc_len'] == 8 and blk.command_vals['start_time'] == -1 and not blk.update_pending
    assert blk.command_key.endswith('/Corr/%d' % blk.instance_id)


def test_corr_streams_with_lag_one_commits():
    """On the in-repo ring Corr only enqueues its gulps and dumps; the span of integration n is committed after the dump
    of integration n+1 has been enqueued (one xgpu_sync_lag per integration but the first), the last one at sequence end.
    Spans still come out complete, in order, one per integration."""
    seqs, be, data, blk = _corr_scenario([12])
    (ohdr, tag, spans), = seqs
    assert len(spans) == 6 and getattr(be, "async_calls", 0) == 12 and be.lag_syncs == 5
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(data[0][4 * k:4 * k + 4], 4, 2))


def test_corr_acc_len_zero_is_a_clean_stop():
    seqs, be, data, blk = _corr_scenario([6], commands=[cmd(1, acc_len=0)])
    assert seqs == [] and be.kernel_calls == [] and blk.stats['state'] == 'stopped'


# ----------------------------------------------------------------------------------------------
def test_corr_then_corracc_chain():
    """Corr (acc 4) -> CorrAcc (acc 8, start -1 = "now"): each long integration is the sum of two
    short ones; header carries upstream_acc_len (corr_acc_block.py:216)."""
    C, S, g = 2, 8, 2
    rng = np.random.default_rng(3)
    vin = rng.integers(0, 256, (32, C, S, 2), dtype=np.uint8)
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-slow-output")
    be = OracleBackend()
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=4, autostartat=0, backend=be)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=8, autostartat=-1, backend=be)
    sink = Sink(r2, cacc.ogulp_size)
    run_blocks([corr, cacc], Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)]), [sink])
    (ohdr, tag, spans), = sink.sequences
    assert ohdr['upstream_acc_len'] == 4 and ohdr['acc_len'] == 8 and ohdr['seq0'] == 0 and tag == 1
    assert len(spans) == 4
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[8 * k:8 * k + 8], S, C))
    assert cacc.stats['state'] == 'running' and cacc.stats['last_end_sample'] == 28


def test_corracc_start_time_and_stop():
    C, S = 2, 4
    matlen = C * orc.per_chan(S)
    rng = np.random.default_rng(4)
    blocks = rng.integers(-1000, 1000, (10, 2 * matlen)).astype(np.int32)
    r1, r2 = Ring("a"), Ring("b")
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=12, autostartat=8, backend=OracleBackend())
    hdr = dict(source_header(C, S, 2, seq0=0), acc_len=4)
    sink = Sink(r2, cacc.ogulp_size)
    run_blocks([cacc], Source(r1, [(hdr, blocks, 2 * matlen * 4)]), [sink])
    (ohdr, tag, spans), = sink.sequences
    assert ohdr['seq0'] == 8 and len(spans) == 2              # [8,20) and [20,32): upstream blocks 2-4, 5-7
    assert np.array_equal(spans[0].view(np.int32), blocks[2:5].sum(0, dtype=np.int32))
    assert np.array_equal(spans[1].view(np.int32), blocks[5:8].sum(0, dtype=np.int32))


# ----------------------------------------------------------------------------------------------
def _beam_cmds(nchan, nbeam, ninput, rng, load_sample=None):
    """Command stream as beamformer_test.py:152-183 sends it: cal gains per (beam,input), then beam coeffs."""
    cmds, cal, delays, amps = [], {}, {}, {}
    k = 0
    for b in range(nbeam):
        for i in range(ninput):
            g = rng.uniform(-1, 1, 2 * nchan)
            cal[b, i] = g[0::2] + 1j * g[1::2]
            cmds.append(cmd(k, coeffs={'type': 'calgains', 'input_id': i, 'beam_id': b, 'data': g.tolist()}))
            k += 1
    for b in range(nbeam):
        delays[b], amps[b] = rng.uniform(0, 12, ninput), rng.uniform(10, 17, ninput)
        c = {'type': 'beamcoeffs', 'beam_id': b, 'data': {'delays': delays[b].tolist(), 'amps': amps[b].tolist()}}
        if load_sample is not None:
            c['load_sample'] = load_sample
        cmds.append(cmd(k, coeffs=c))
        k += 1
    return cmds, cal, delays, amps


def test_beamform_and_sum_beams_chain():
    """DummySource-style input -> Beamform (weights commanded as the control library does) ->
    BeamformSumBeams; voltage beams and power beams equal the oracle with the weights of
    beamform_block.py:343-350; headers rewritten as beamform_block.py:403-409 /
    beamform_sum_beams_block.py:211-216."""
    nchan, nstand, nbeam, g, ntime_sum = 3, 6, 4, 8, 4
    ninput = nstand * 2
    rng = np.random.default_rng(0xaabbccdd)
    vin = rng.integers(0, 256, (3 * g, nchan, ninput), dtype=np.uint8)
    sfreq, chan_bw = 50e6, 23925.78125
    r0, r1, r2 = Ring("gpu-input"), Ring("bf-output"), Ring("bf-pow-output")
    be = OracleBackend()
    bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=be)
    sb = BeamformSumBeams(LOG, r1, r2, nchan=nchan, ntime_gulp=g, ntime_sum=ntime_sum, backend=be)
    cmds, cal, delays, amps = _beam_cmds(nchan, nbeam, ninput, rng)
    hdr = source_header(nchan, nstand, 2, seq0=0, sfreq=sfreq, chan_bw=chan_bw)
    # commands are processed against the sequence's frequencies: inject them once main() has the header
    bf.freqs = sfreq + chan_bw * np.arange(nchan)
    bf.process_command_strings(cmds)
    s1, s2 = Sink(r1, g * nchan * nbeam * 8), Sink(r2, (nbeam // 2) * (g // ntime_sum) * nchan * 16)
    run_blocks([bf, sb], Source(r0, [(hdr, vin, g * nchan * ninput)], wait_readers=1), [s1, s2])
    freqs = sfreq + chan_bw * np.arange(nchan)
    w = np.zeros((nchan, nbeam, ninput), np.complex64)
    for b in range(nbeam):
        calb = np.stack([cal[b, i] for i in range(ninput)], axis=1)            # [chan, input]
        w[:, b, :] = amps[b] * np.exp(1j * 2 * np.pi * freqs[:, None] * delays[b] * 1e-9) * calb
    assert np.allclose(bf.gains_cpu, w, rtol=1e-6, atol=1e-6)
    (h1, _, spans1), = s1.sequences
    assert h1['nstand'] == nbeam and h1['npol'] == 1 and h1['nbit'] == 32 and h1['complex'] and h1['nbeam'] == nbeam
    assert len(spans1) == 3
    for k, sp in enumerate(spans1):
        exp = orc.beamform(vin[k * g:(k + 1) * g], bf.gains_cpu, g, nchan, ninput, nbeam)
        assert np.array_equal(sp.view(np.complex64).reshape(exp.shape), exp)
    (h2, _, spans2), = s2.sequences
    assert h2['nbeam'] == nbeam // 2 and h2['npol'] == 2 and h2['acc_len'] == ntime_sum
    for k, sp in enumerate(spans2):
        beams = spans1[k].view(np.complex64).reshape(nchan, nbeam, g)
        exp = orc.beamform_integrate(beams, ntime_sum)
        assert np.array_equal(sp.view(np.float32).reshape(exp.shape), exp)


def test_beamform_timed_coefficient_load():
    """load_sample: new coefficients take effect on the first gulp whose start time >= load_sample
    (beamform_block.py:416-429); before that the beam has zero gains."""
    nchan, nstand, nbeam, g = 2, 4, 2, 4
    ninput = nstand * 2
    rng = np.random.default_rng(7)
    vin = rng.integers(0, 256, (4 * g, nchan, ninput), dtype=np.uint8)
    r0, r1 = Ring("i"), Ring("o")
    bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=OracleBackend())
    bf.freqs = 1e6 + 1e3 * np.arange(nchan)
    cmds, cal, delays, amps = _beam_cmds(nchan, nbeam, ninput, rng, load_sample=2 * g)
    bf.process_command_strings(cmds)
    hdr = source_header(nchan, nstand, 2, seq0=0, sfreq=1e6, chan_bw=1e3)
    s1 = Sink(r1, g * nchan * nbeam * 8)
    run_blocks([bf], Source(r0, [(hdr, vin, g * nchan * ninput)]), [s1])
    (h1, _, spans), = s1.sequences
    assert not spans[0].any() and not spans[1].any()
    for k in (2, 3):
        exp = orc.beamform(vin[k * g:(k + 1) * g], bf.gains_cpu_new, g, nchan, ninput, nbeam)
        assert np.array_equal(spans[k].view(np.complex64).reshape(exp.shape), exp)


def test_block_instance_ids_and_keys():
    class A(Block):
        pass

    class B(Block):
        pass
    Block.set_id(3)
    a0, a1, b0 = (k(LOG, Ring("x"), None, True, -1) for k in (A, A, B))
    assert (a0.instance_id, a1.instance_id, b0.instance_id) == (0, 1, 0)
    assert '/pipeline/3/A/1' in a1.command_key and a1.monitor_key.startswith('/mon/corr/x/')
    Block.set_id(0)


# ----------------------------------------------------------------------------------------------
def test_copy_block_passes_headers_and_data():
    from caltech_bifrost_dsp_amd.blocks import Copy
    r0, r1 = Ring("capture"), Ring("gpu-input")
    cp = Copy(LOG, r0, r1, ntime_gulp=4, nbyte_per_time=10, buffer_multiplier=2)
    assert cp.igulp_size == 40 and cp.buf_size == 4 * 40 * 2
    data = np.arange(100, dtype=np.uint8)                   # 2.5 gulps: the half gulp is dropped
    hdr = source_header(1, 5, 2, seq0=7)
    sink = Sink(r1, 40)
    run_blocks([cp], Source(r0, [(hdr, data, 20)]), [sink])
    (ohdr, tag, spans), = sink.sequences
    assert ohdr == hdr and len(spans) == 2
    assert np.array_equal(np.concatenate(spans), data[:80])


def test_corr_then_subsel_chain(golden_dir):
    """Corr -> CorrSubsel on CPU rings: the sub-selected, channel-summed visibilities equal the golden
    x[s0,p0]*conj(x[s1,p1]) (verification/test_corr_part_rx.py:49-85), including after a `baselines`
    command, which starts a new output sequence (corr_subsel_block.py:316-329)."""
    from caltech_bifrost_dsp_amd.blocks import CorrSubsel
    z = np.load(os.path.join(golden_dir, "golden_64t_32a_8c_32s_2p_deadbeef.npz"))
    vin, gre, gim = z["vin"], z["corr_re"], z["corr_im"]
    T, C, S, P = vin.shape
    nvis, nsum = 12, 4
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-fast-output")
    be = OracleBackend()
    hdr = source_header(C, S, P, seq0=0)
    corr = Corr(LOG, r0, r1, ntime_gulp=16, nchan=C, npol=P, nstand=S, acc_len=32, autostartat=0,
                ant_to_input=hdr['ant_to_input'], backend=be)
    sub = CorrSubsel(LOG, r1, r2, nchan=C, npol=P, nstand=S, nchan_sum=nsum, backend=be, nvis_out=nvis,
                     antpol_to_bl=corr.antpol_to_bl.numpy(), bl_is_conj=corr.bl_is_conj.numpy())
    rng = np.random.default_rng(6)
    sel = [[[int(a), int(b)], [int(c), int(d)]] for a, b, c, d in
           zip(rng.integers(0, S, nvis), rng.integers(0, 2, nvis), rng.integers(0, S, nvis), rng.integers(0, 2, nvis))]
    sub.process_command_strings(cmd(1, baselines=sel))
    sink = Sink(r2, sub.ogulp_size)
    run_blocks([corr, sub], Source(r0, [(hdr, vin, 16 * C * S * P)]), [sink])
    (ohdr, tag, spans), = sink.sequences
    assert ohdr['nchan'] == C // nsum and ohdr['nvis'] == nvis and ohdr['nchan_sum'] == nsum and ohdr['baselines'] == sel
    assert len(spans) == 2
    for k, sp in enumerate(spans):
        got = sp.view(np.int32).reshape(C // nsum, nvis, 2)
        for v, ((s0, p0), (s1, p1)) in enumerate(sel):
            assert np.array_equal(got[:, v, 0], gre[k][:, s0, s1, p0, p1].reshape(C // nsum, nsum).sum(1))
            assert np.array_equal(got[:, v, 1], gim[k][:, s0, s1, p0, p1].reshape(C // nsum, nsum).sum(1))
    # wrong-length selection lists are rejected (condition len == nvis_out)
    sub.process_command_strings(cmd(2, baselines=sel[:-1]))
    assert sub.stats['last_cmd_response'] == COMMAND_INVALID


@pytest.mark.parametrize("use_cor_fmt", [False, True])
def test_corr_then_output_full_packets(golden_dir, tmp_path, use_cor_fmt):
    """Corr -> CorrOutputFull on CPU rings with the reference's golden visibilities as `checkfile`
    (corr_output_full_block.py:550-603: every component of every baseline must match) and a packet sink:
    one packet per dual-pol baseline s0 <= s1 in sending order, 56-byte `>QQ2d4I2I` header + [p0][p1][chan][2]
    payload (:443-463), or the 32-byte COR header + [chan][p0][p1][2] payload (:213-226, 512-519)."""
    import struct
    from caltech_bifrost_dsp_amd.blocks import CorrOutputFull
    _, vin = load_dat(os.path.join(golden_dir, "in_8t_4c_16s_2p_deadbeef.dat"))
    meta, corr = load_dat(os.path.join(golden_dir, "corr_8t_4a_4c_16s_2p_deadbeef.dat"))
    T, C, S, P = vin.shape
    acc_len = meta["acc_len"]
    check = tmp_path / "golden_raw.bin"                      # raw complex128 [t][chan][s0][s1][p0][p1], as :419-434 reads it
    check.write_bytes(np.ascontiguousarray(corr).tobytes())
    r0, r1 = Ring("gpu-input"), Ring("corr-output")
    be = OracleBackend()
    hdr = source_header(C, S, P, seq0=0, sync_time=1600000000, sfreq=4.0e7)
    cblk = Corr(LOG, r0, r1, ntime_gulp=2, nchan=C, npol=P, nstand=S, acc_len=acc_len, autostartat=0,
                ant_to_input=hdr['ant_to_input'], backend=be)
    pkts = []
    oblk = CorrOutputFull(LOG, r1, nchan=C, npol=P, nstand=S, checkfile=str(check), checkfile_acc_len=acc_len,
                          antpol_to_bl=cblk.antpol_to_bl.numpy(), bl_is_conj=cblk.bl_is_conj.numpy(),
                          use_cor_fmt=use_cor_fmt, nchan_sum=1, pipeline_idx=3, npipeline=2, backend=be,
                          sink=pkts.append)
    run_blocks([cblk, oblk], Source(r0, [(hdr, vin, 2 * C * S * P)]), [])
    nint, nbl = T // acc_len, S * (S + 1) // 2
    assert oblk.check_results == [(nbl * 8, 0)] * nint           # the reference's own golden check, all good
    assert len(pkts) == nint * nbl
    hlen = 32 if use_cor_fmt else 56
    k = 0
    for it in range(nint):
        for s0 in range(S):
            for s1 in range(s0, S):
                pkt = pkts[k]
                k += 1
                assert len(pkt) == hlen + 4 * C * 2 * 4
                pay = np.frombuffer(pkt[hlen:], dtype=np.int32)
                g = corr[it, :, s0, s1]                              # [chan][p0][p1] complex
                if use_cor_fmt:
                    sync, w1, secs, f0, gain, tt, navg, si, sj = struct.unpack(">IIIhhqihh", pkt[:32])
                    spp = int(C * hdr['fs_hz'] / hdr['bw_hz'])
                    assert sync == 0x5CDEC0DE and w1 >> 24 == 2 and (w1 & 0xFFFFFF) == (1 << 16) | (2 << 8) | 1
                    assert (f0, gain, tt, navg, si, sj) == (0, 0, it * acc_len * spp, acc_len * spp, s0 + 1, s1 + 1)
                    pay = pay.reshape(C, P, P, 2)
                    assert np.array_equal(pay[..., 0], g.real) and np.array_equal(pay[..., 1], g.imag)
                else:
                    f = struct.unpack(">QQ2d4I2I", pkt[:56])
                    assert f == (1600000000, it * acc_len, hdr['bw_hz'], 4.0e7, acc_len, C, 0, P, s0, s1)
                    pay = pay.reshape(P, P, C, 2)
                    assert np.array_equal(pay[..., 0], np.moveaxis(g.real, 0, -1))
                    assert np.array_equal(pay[..., 1], np.moveaxis(g.imag, 0, -1))
    assert oblk.stats['curr_sample'] == (nint - 1) * acc_len and 'output_gbps' in oblk.stats


def test_snap2_ingest_then_corr(golden_dir):
    """Packets of the reference's SNAP2 emulator (test_tx_vectors.py:79-112) -> Snap2Ingest -> Corr on CPU rings:
    the gulps are restored exactly, the sequence header is the one capture_block.py:264-282 builds, and the
    visibilities equal the golden file; a lost packet blanks its samples and is reported."""
    from caltech_bifrost_dsp_amd.blocks import Snap2Ingest
    _, vin = load_dat(os.path.join(golden_dir, "in_8t_4c_16s_2p_deadbeef.dat"))
    meta, corr = load_dat(os.path.join(golden_dir, "corr_8t_4a_4c_16s_2p_deadbeef.dat"))
    T, C, S, P = vin.shape
    g = 2                                                              # sequence numbers per gulp
    be = OracleBackend()
    seq0, chan0 = 4000, 96
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=1234, nchan_blocks=2, nstand_per_pkt=8, chan0_pipeline=chan0)
    per_win = len(pk) // (T // g)
    rng = np.random.default_rng(8)
    slabs = []
    for w in range(T // g):                                           # arrival order inside a window is arbitrary
        win = pk[w * per_win:(w + 1) * per_win]
        slabs.append(b"".join(win[i] for i in rng.permutation(per_win)))
    r_pk, r_in, r_vis = Ring("packets"), Ring("gpu-input"), Ring("corr-output")
    ing = Snap2Ingest(LOG, r_pk, r_in, ntime_gulp=g, nchan=C, nstand=S, npol=P, nchan_per_pkt=C // 2, nstand_per_pkt=8,
                      backend=be)
    assert ing.npkt_per_gulp == per_win and ing.pkt_stride == len(pk[0])
    hdr_src = source_header(C, S, P, seq0=seq0, chan0=chan0)
    cblk = Corr(LOG, r_in, r_vis, ntime_gulp=g, nchan=C, npol=P, nstand=S, acc_len=meta["acc_len"], autostartat=seq0,
                ant_to_input=hdr_src['ant_to_input'], backend=be)
    raw_sink, vis_sink = Sink(r_in, ing.ogulp_size), Sink(r_vis, cblk.ogulp_size)
    run_blocks([ing, cblk], Source(r_pk, [({'seq0': seq0, 'chan0': chan0, 'sync_time': 1234}, b"".join(slabs), ing.igulp_size)]),
               [raw_sink, vis_sink])
    (h, tag, spans), = raw_sink.sequences
    assert (h['seq0'], h['chan0'], h['nchan'], h['nstand'], h['npol'], h['sync_time'], h['nbit'], h['complex']) == \
           (seq0, chan0, C, S, P, 1234, 4, True)
    assert h['sfreq'] == chan0 * 23925.78125 and h['bw_hz'] == C * 23925.78125 and tag == 1
    assert np.array_equal(np.concatenate(spans).reshape(vin.shape), vin)
    assert ing.stats['packets_placed'] == len(pk) and ing.stats['packets_dropped'] == 0 and ing.stats['missing_frac'] == 0.0
    # Corr's seq0-relative start: the visibilities equal the golden ones
    (vh, _, vspans), = vis_sink.sequences
    bl, cj = cblk.antpol_to_bl.numpy(), cblk.bl_is_conj.numpy()
    ro = orc.xgpu_reorder(vspans[0].view(np.int32), bl, cj, C)
    gold = corr[0]
    assert np.array_equal(ro[1, 5, :, :, :, 0], np.moveaxis(gold[:, 1, 5].real, 0, -1))
    assert np.array_equal(ro[1, 5, :, :, :, 1], np.moveaxis(gold[:, 1, 5].imag, 0, -1))

    # a window with one packet lost (slot left zero) and a stray packet of another window in its place
    r_pk2, r_in2 = Ring("packets"), Ring("gpu-input")
    ing2 = Snap2Ingest(LOG, r_pk2, r_in2, ntime_gulp=g, nchan=C, nstand=S, npol=P, nchan_per_pkt=C // 2, nstand_per_pkt=8,
                       backend=be)
    win = list(pk[:per_win])
    win[3] = pk[per_win]                                               # belongs to the next window -> dropped
    sink2 = Sink(r_in2, ing2.ogulp_size)
    run_blocks([ing2], Source(r_pk2, [({'seq0': seq0, 'chan0': chan0}, b"".join(win), ing2.igulp_size)]), [sink2])
    got = sink2.sequences[0][2][0].reshape(g, C, S * P)
    exp = vin[:g].reshape(g, C, S * P).copy()
    import struct
    seq, _, npol, _, nchan, _, _, c0, p0 = struct.unpack(orc.SNAP2_HDR, pk[3][:32])
    exp[seq - seq0, c0 - chan0:c0 - chan0 + nchan, p0:p0 + npol] = 0
    assert np.array_equal(got, exp)
    assert ing2.stats['packets_placed'] == per_win - 1 and ing2.stats['packets_dropped'] == 1
    assert abs(ing2.stats['missing_frac'] - 1.0 / per_win) < 1e-12


def test_beamform_sum_beams_output_packets():
    """Beamform -> BeamformSumBeams -> BeamformOutput: one 18-byte-header packet per (dual-pol beam, integrated
    time sample), header fields and the seq step of beamform_output_block.py:303-309, payload f32[nchan][4]
    = [XX, YY, re XY, im XY] (the oracle's power sums), destinations per beam modulo the list length."""
    import struct
    from caltech_bifrost_dsp_amd.blocks import BeamformOutput
    nchan, nstand, nbeam, g, ntime_sum = 3, 6, 4, 8, 4
    ninput = nstand * 2
    rng = np.random.default_rng(5)
    vin = rng.integers(0, 256, (2 * g, nchan, ninput), dtype=np.uint8)
    sfreq, chan_bw = 50e6, 23925.78125
    r0, r1, r2 = Ring("gpu-input"), Ring("bf-output"), Ring("bf-pow-output")
    be = OracleBackend()
    bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=be)
    sb = BeamformSumBeams(LOG, r1, r2, nchan=nchan, ntime_gulp=g, ntime_sum=ntime_sum, backend=be)
    pk = []
    out = BeamformOutput(LOG, r2, ntime_gulp=g // ntime_sum, pipeline_idx=3, nchan=nchan, nbeam=nbeam // 2,
                         sink=lambda b, p: pk.append((b, p)))
    cmds, cal, delays, amps = _beam_cmds(nchan, nbeam, ninput, rng)
    hdr = source_header(nchan, nstand, 2, seq0=960, chan0=nchan * 2, sfreq=sfreq, chan_bw=chan_bw)
    hdr['system_nchan'] = nchan * 8
    bf.freqs = sfreq + chan_bw * np.arange(nchan)
    bf.process_command_strings(cmds)
    run_blocks([bf, sb, out], Source(r0, [(hdr, vin, g * nchan * ninput)], wait_readers=1), [])
    nblk = g // ntime_sum
    assert len(pk) == 2 * (nbeam // 2) * nblk
    k = 0
    for span in range(2):
        beams = orc.beamform(vin[span * g:(span + 1) * g], bf.gains_cpu, g, nchan, ninput, nbeam)
        power = orc.beamform_integrate(beams, ntime_sum)                  # [nbeam/2][nblk][nchan][4]
        for b in range(nbeam // 2):
            for t in range(nblk):
                beam, p = pk[k]
                k += 1
                assert beam == b and len(p) == 18 + nchan * 16
                server, bb, tuning, nc, nb, nserver = struct.unpack(">6B", p[:6])
                navg, chan0 = struct.unpack(">2H", p[6:10])
                seq, = struct.unpack(">Q", p[10:18])
                assert (server, bb, tuning, nc, nb, nserver, navg, chan0) == (2, b, 1, nchan, nbeam // 2, 8, ntime_sum, nchan * 2)
                # spans advance by acc_len * ntime_gulp samples (:378); inside a span seq steps by ntime_gulp * navg (:309)
                assert seq == 960 + span * ntime_sum * nblk + t * nblk * ntime_sum
                assert np.array_equal(np.frombuffer(p[18:], dtype=np.float32).reshape(nchan, 4), power[b, t])
    assert out.stats['last_end_sample'] == 960 + ntime_sum * nblk
    # destinations: beam i -> dest_ip[i % len]
    out.process_command_strings(cmd(1, dest_ip=['0.0.0.0', '127.0.0.1'], dest_port=[10000, 10001]))
    out._update_destinations()
    assert out.beam_ips == ['0.0.0.0', '127.0.0.1'] and out.beam_ports == [10000, 10001]
    assert out.socks[0] is None and out.socks[1] is not None
    out.socks[1].close()


def test_corr_subsel_then_output_part_packets(golden_dir):
    """Corr -> CorrSubsel -> CorrOutputPart: packets of `nvis_per_packet` visibilities, big-endian header
    `>QQ2d4I` + baselines + data[nvis][nchan][2] (corr_output_part_block.py:346-364), values = the golden
    x[s0,p0]*conj(x[s1,p1]) summed over nchan_sum channels."""
    import struct
    from caltech_bifrost_dsp_amd.blocks import CorrOutputPart, CorrSubsel
    z = np.load(os.path.join(golden_dir, "golden_64t_32a_8c_32s_2p_deadbeef.npz"))
    vin, gre, gim = z["vin"], z["corr_re"], z["corr_im"]
    T, C, S, P = vin.shape
    nvis, nsum, npp = 12, 4, 4
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-fast-output")
    be = OracleBackend()
    hdr = source_header(C, S, P, seq0=0, sync_time=99, sfreq=3.0e7)
    corr = Corr(LOG, r0, r1, ntime_gulp=16, nchan=C, npol=P, nstand=S, acc_len=32, autostartat=0,
                ant_to_input=hdr['ant_to_input'], backend=be)
    sub = CorrSubsel(LOG, r1, r2, nchan=C, npol=P, nstand=S, nchan_sum=nsum, backend=be, nvis_out=nvis,
                     antpol_to_bl=corr.antpol_to_bl.numpy(), bl_is_conj=corr.bl_is_conj.numpy())
    rng = np.random.default_rng(2)
    sel = [[[int(a), int(b)], [int(c), int(d)]] for a, b, c, d in
           zip(rng.integers(0, S, nvis), rng.integers(0, 2, nvis), rng.integers(0, S, nvis), rng.integers(0, 2, nvis))]
    sub.process_command_strings(cmd(1, baselines=sel))
    pk = []
    out = CorrOutputPart(LOG, r2, nvis_per_packet=npp, nchan_sum=nsum, sink=pk.append)
    run_blocks([corr, sub, out], Source(r0, [(hdr, vin, 16 * C * S * P)]), [])
    nco = C // nsum
    assert len(pk) == 2 * (nvis // npp)
    k = 0
    for it in range(2):
        for vn in range(nvis // npp):
            p = pk[k]
            k += 1
            hlen = 56 - 8 + 8 + 16 * npp                       # `>QQ2d4I` is 48 bytes; then 4 int32 per visibility
            f = struct.unpack(">QQ2d4I", p[:48])
            chan_width = hdr['bw_hz'] / C
            assert f[:2] == (99, it * 32) and f[2] == hdr['bw_hz'] and f[4:] == (32, npp, nco, 0)
            assert abs(f[3] - (3.0e7 + (nsum - 1) * chan_width) / nsum) < 1e-6            # corr_subsel_block.py header
            bl = np.frombuffer(p[48:48 + 16 * npp], dtype='>i4').reshape(npp, 2, 2)
            assert bl.tolist() == sel[vn * npp:(vn + 1) * npp]
            data = np.frombuffer(p[48 + 16 * npp:], dtype='>i4').reshape(npp, nco, 2)
            assert len(p) == hlen - 8 + npp * nco * 8
            for v in range(npp):
                (s0, p0), (s1, p1) = sel[vn * npp + v]
                assert np.array_equal(data[v, :, 0], gre[it][:, s0, s1, p0, p1].reshape(nco, nsum).sum(1))
                assert np.array_equal(data[v, :, 1], gim[it][:, s0, s1, p0, p1].reshape(nco, nsum).sum(1))
    with pytest.raises(NotImplementedError):
        CorrOutputPart(LOG, r2, use_cor_fmt=True)


# ---------------------------------------------------------------------------------------------- fused CorrAcc
def _corr_corracc(seqs, C, S, g, acc, lacc, cacc_start, fused, cacc_cmds=(), group_dumps=None):
    """Corr -> CorrAcc on in-repo rings with the oracle backend.  fused = True: the dumps feed CorrAcc's accumulators themselves
    (bfXgpuKernelAsyncAcc); False: the reference's map per span (corr_acc_block.py:298-306); 'group' (round 5, the default of
    the block): the spans of a group of dumps are summed in one pass (map_sum_i32)."""
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-slow-output")
    be = OracleBackend()
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=cacc_start, backend=be,
                   accumulate='fused' if fused is True else 'map' if fused is False else fused)
    if group_dumps is not None:
        cacc.group_dumps = group_dumps
    if fused is True:
        assert r1.long_accumulator is cacc
    else:
        assert getattr(r1, 'long_accumulator', None) is None
    for c in cacc_cmds:
        cacc.process_command_strings(c)
    fast, slow = Sink(r1, corr.ogulp_size), Sink(r2, cacc.ogulp_size)
    run_blocks([corr, cacc], Source(r0, seqs), [fast, slow])
    return corr, cacc, be, fast, slow


@pytest.mark.parametrize("cacc_start,lacc", [(0, 12), (8, 12), (-1, 8), (4, 4), (16, 20)])
def test_fused_corracc_equals_classic_and_oracle(cacc_start, lacc):
    """The long integration published with the add fused into Corr's dumps equals the classic map path and N x the oracle
    integration, for aligned starts, a start in the middle of the stream, "now" (-1), one dump per long integration and an
    odd number (5) of dumps per long integration (the two partial accumulators hold 3 + 2 dumps)."""
    C, S, g, acc = 2, 8, 2, 4
    rng = np.random.default_rng(11)
    vin = rng.integers(0, 256, (64, C, S, 2), dtype=np.uint8)
    seqs = [(source_header(C, S, 2), vin, g * C * S * 2)]
    out = {}
    for fused in (True, False, 'group'):
        corr, cacc, be, fast, slow = _corr_corracc(seqs, C, S, g, acc, lacc, cacc_start, fused, group_dumps=2 if fused == 'group' else None)
        assert corr.stats['fused_corracc'] is (fused is True) and cacc.stats['fused'] is (fused is True)
        assert cacc.stats['grouped'] is (fused == 'group')
        (fh, _, fspans), = fast.sequences
        assert len(fspans) == 16                                           # the fast stream is untouched
        for k, sp in enumerate(fspans):
            assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[acc * k:acc * (k + 1)], S, C))
        out[fused] = [(h, [sp.view(np.int32).copy() for sp in spans]) for h, _, spans in slow.sequences]
        ndumps = getattr(be, "acc_calls", [])
        if fused == 'group':
            # groups of (at most) two dumps, one pass each: the first of a long integration assigns, the others add
            per = lacc // acc
            start = 0 if cacc_start == -1 else cacc_start
            nlong = (64 - start) // lacc
            one = [(min(2, per - k), k > 0) for k in range(0, per, 2)]
            assert be.sum_calls[:len(one) * nlong] == one * nlong and ndumps == []
        elif fused:
            start = 0 if cacc_start == -1 else cacc_start
            nlong = (64 - start) // lacc
            assert len(out[True][0][1]) == nlong
            # every dump of a started long integration went to an accumulator: assign for the first two of each, add after
            per = lacc // acc
            want = ([1] * min(2, per) + [2] * max(0, per - 2)) * nlong
            assert ndumps[:len(want)] == want and cacc.fused_dumps >= len(want)
            (h, spans), = out[True]
            assert h['seq0'] == start and h['acc_len'] == lacc and h['upstream_acc_len'] == acc and 'fused_corracc' not in h
            for k, sp in enumerate(spans):
                assert np.array_equal(sp, orc.xgpu_correlate(vin[start + lacc * k:start + lacc * (k + 1)], S, C))
        else:
            assert ndumps == []
    assert len(out[True]) == len(out[False]) == len(out['group'])
    for other in (False, 'group'):
        for (h1, s1), (h2, s2) in zip(out[True], out[other]):
            assert h1 == h2 and len(s1) == len(s2) and all(np.array_equal(a, b) for a, b in zip(s1, s2))


def test_fused_corracc_follows_commands_and_new_upstream_sequences():
    """A start-time command that arrives before the stream, a second upstream sequence (the gate recovers two long
    integrations later, corr_acc_block.py:221-227) and the long integration cut off by the end of the first sequence:
    fused and classic publish the same spans under the same headers."""
    C, S, g, acc, lacc = 2, 4, 2, 4, 8
    rng = np.random.default_rng(12)
    d0 = rng.integers(0, 256, (28, C, S, 2), dtype=np.uint8)      # 7 dumps: starts at 8 -> one long integration [8,24), one cut off
    d1 = rng.integers(0, 256, (80, C, S, 2), dtype=np.uint8)
    seqs = [(source_header(C, S, 2, seq0=0), d0, g * C * S * 2), (source_header(C, S, 2, seq0=1000), d1, g * C * S * 2)]
    res = {}
    for fused in (True, False, 'group'):
        corr, cacc, be, fast, slow = _corr_corracc(seqs, C, S, g, acc, lacc, 0, fused, cacc_cmds=[cmd(1, start_time=8)])
        res[fused] = [(h, tag, [sp.view(np.int32).copy() for sp in spans]) for h, tag, spans in slow.sequences]
        assert sum(len(sp) for _, _, sp in res[fused]) >= 2
    assert len(res[True]) == len(res[False]) == len(res['group'])
    for other in (False, 'group'):
        for (h1, t1, s1), (h2, t2, s2) in zip(res[True], res[other]):
            assert h1 == h2 and t1 == t2 and len(s1) == len(s2) and all(np.array_equal(a, b) for a, b in zip(s1, s2))
    (h, _, spans) = res[True][0]
    assert h['seq0'] == 8 and np.array_equal(spans[0], orc.xgpu_correlate(d0[8:16], S, C))


# ---------------------------------------------------------------------------------------------- failures mid-stream
def test_a_failing_block_waits_for_its_kernels_before_it_lets_go_of_their_spans():
    """A block thread that dies with gulps in flight (a library call fails mid-stream) must wait for the GPU before its
    spans are released: their memory goes back to the ring -- to be handed out again or freed -- and a kernel still writing
    there would be a device memory fault.  Beamform / BeamformSumBeams: beam_sync; Corr: xgpu_sync."""
    class Failing(OracleBackend):
        def __init__(self, fail_at):
            super().__init__()
            self.fail_at, self.calls, self.order = fail_at, 0, []

        def bfBeamformRun(self, i, o, w, version=0):
            self.calls += 1
            if self.calls == self.fail_at:
                return 3
            return super().bfBeamformRun(i, o, w, version=version)

        def bfXgpuKernelAsync(self, i, o, d):
            self.calls += 1
            if self.calls == self.fail_at:
                return 3
            return super().bfXgpuKernelAsync(i, o, d)

        def beam_sync(self):
            self.order.append("beam_sync")

        def xgpu_sync(self):
            self.order.append("xgpu_sync")
            return 0

    nchan, nstand, nbeam, g = 2, 4, 2, 4
    ninput = nstand * 2
    rng = np.random.default_rng(5)
    vin = rng.integers(0, 256, (6 * g, nchan, ninput), dtype=np.uint8)
    # Beamform: the third Run fails while two gulps are in flight
    r0, r1 = Ring("gpu-input"), Ring("bf-output")
    be = Failing(3)
    bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=be)
    sink = Sink(r1, g * nchan * nbeam * 8)
    src = Source(r0, [(source_header(nchan, nstand, 2), vin, g * nchan * ninput)], wait_readers=1)
    sink.start()
    src.start()
    with pytest.raises(RuntimeError, match="bfBeamformRun returned 3"):
        bf.main()
    assert be.order == ["beam_sync"]
    # Corr: the second gulp of the second integration fails while the first integration's dump is in flight
    r2, r3 = Ring("in"), Ring("out")
    be = Failing(4)
    blk = Corr(LOG, r2, r3, ntime_gulp=g, nchan=nchan, npol=2, nstand=nstand, acc_len=2 * g, autostartat=0, backend=be)
    sink2 = Sink(r3, blk.ogulp_size)
    src2 = Source(r2, [(source_header(nchan, nstand, 2), vin.reshape(6 * g, nchan, nstand, 2), g * nchan * ninput)], wait_readers=1)
    sink2.start()
    src2.start()
    with pytest.raises(RuntimeError, match="xgpuKernel returned 3"):
        blk.main()
    assert be.order and be.order[-1] == "xgpu_sync"
