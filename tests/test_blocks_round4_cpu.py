"""Round-4 host-logic cases of the blocks (CPU rings, oracle backend; both ring implementations)."""
import threading
import time

import numpy as np

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd.blocks import Corr, CorrAcc
from caltech_bifrost_dsp_amd.ring import Ring
from oracle import xeng_oracle as orc
from tests.fake_backend import OracleBackend
from tests.pipeline_util import LOG, Sink, Source, run_blocks, source_header


def test_fused_registration_needs_a_guaranteed_first_reader():
    """Only one CorrAcc per ring, and only a guaranteed one, may have the upstream Corr's dumps feed its accumulators: a
    second CorrAcc (it would overwrite the ring's single slot) and a reader without the guarantee (it may skip spans, and
    the decision queue would fall out of step) keep the map path."""
    C, S = 2, 8
    be = OracleBackend()
    r1, r2, r3 = Ring("corr-output"), Ring("slow-a"), Ring("slow-b")
    a = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=8, backend=be, accumulate='fused')
    b = CorrAcc(LOG, r1, r3, nchan=C, npol=2, nstand=S, acc_len=8, backend=be, accumulate='fused')
    assert r1.long_accumulator is a and a._iseqs is not None and b._iseqs is None
    r4 = Ring("corr-output-2")
    c = CorrAcc(LOG, r4, Ring("slow-c"), nchan=C, npol=2, nstand=S, acc_len=8, backend=be, guarantee=False, accumulate='fused')
    assert getattr(r4, 'long_accumulator', None) is None and c._iseqs is None
    del a, b, c


def test_two_corraccs_on_one_ring_one_fused_one_classic():
    C, S, g, acc, lacc = 2, 8, 2, 4, 8
    vin = np.random.default_rng(5).integers(0, 256, (32, C, S, 2), dtype=np.uint8)
    be = OracleBackend()
    r0, r1, r2, r3 = Ring("gpu-input"), Ring("corr-output"), Ring("slow-a"), Ring("slow-b")
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    a = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, backend=be, accumulate='fused')
    b = CorrAcc(LOG, r1, r3, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, backend=be, accumulate='fused')
    sa, sb = Sink(r2, a.ogulp_size), Sink(r3, b.ogulp_size)
    run_blocks([corr, a, b], Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)]), [sa, sb])
    assert a.stats['fused'] is True and b.stats['fused'] is False         # (the second one maps every span itself)
    want = [orc.xgpu_correlate(vin[lacc * k:lacc * (k + 1)], S, C) for k in range(4)]
    for sink in (sa, sb):
        (_, _, spans), = sink.sequences
        assert len(spans) == 4
        for k, sp in enumerate(spans):
            assert np.array_equal(sp.view(np.int32), want[k])


def test_corr_counts_time_across_skipped_gulps():
    """A Corr that registers after its input ring has overwritten gulps starts on a gulp boundary with the right sample
    count (round 3: the late reader raised and the block thread died): the integrations it emits are those of the oracle
    at the sample counts its output headers name."""
    C, S, g, acc = 2, 8, 2, 4
    vin = np.random.default_rng(9).integers(0, 256, (48, C, S, 2), dtype=np.uint8)
    gulp = g * C * S * 2
    r0, r1 = Ring("gpu-input"), Ring("corr-output")
    r0.resize(gulp, 4 * gulp)
    be = OracleBackend()
    w = r0.begin_writing()
    import json
    oseq = w.begin_sequence(time_tag=0, header=json.dumps(source_header(C, S, 2)))
    raw = vin.reshape(-1)
    for k in range(8):                        # 8 gulps into a ring of 4, nobody reading: gulps 0..3 are overwritten
        with oseq.reserve(gulp) as sp:
            sp.data[...] = raw[k * gulp:(k + 1) * gulp]
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=-1, backend=be)
    sink = Sink(r1, corr.ogulp_size)
    sink.start()
    th = threading.Thread(target=corr.main, daemon=True)
    th.start()
    t0 = time.time()
    while len(r0._readers) < 1 and time.time() - t0 < 10:
        time.sleep(0.005)
    for k in range(8, 24):
        with oseq.reserve(gulp) as sp:
            sp.data[...] = raw[k * gulp:(k + 1) * gulp]
    oseq.end()
    w.__exit__(None, None, None)
    th.join(20)
    sink.join(20)
    assert not th.is_alive() and not sink.is_alive()
    assert sink.sequences
    hdr, _, spans = sink.sequences[0]
    start = hdr['seq0']
    assert start % acc == 0 and start >= 4 * g and spans          # first sample seen is >= gulp 4; starts on an acc_len boundary
    for k, sp in enumerate(spans):
        t = start + k * acc
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[t:t + acc], S, C)), (k, t)


def test_fused_plan_waits_for_a_stalled_consumer_instead_of_failing():
    """Downstream back-pressure in fused mode: while corr-slow-output is full (its consumer stalls), CorrAcc cannot publish, the
    accumulator pair stays busy and the upstream Corr waits in plan_dump -- without a deadline, as the classic path and the
    reference wait on the guaranteed ring.  When the consumer resumes everything is published, nothing lost."""
    C, S, g, acc, lacc = 2, 8, 2, 2, 4
    nlong = 8
    vin = np.random.default_rng(3).integers(0, 256, (nlong * lacc, C, S, 2), dtype=np.uint8)
    be = OracleBackend()
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-slow-output")
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, backend=be, accumulate="fused")
    r2.resize(cacc.ogulp_size, cacc.ogulp_size)              # room for ONE long integration
    gate = threading.Event()
    got = []
    gen = r2.read(guarantee=True)

    def stalled_sink():
        for iseq in gen:
            for ispan in iseq.read(cacc.ogulp_size):
                if not got:
                    gate.wait(20)                            # the consumer stalls on the first span
                got.append(ispan.data.numpy().view(np.int32).copy())

    fast = Sink(r1, corr.ogulp_size)
    ths = [threading.Thread(target=f, daemon=True) for f in (stalled_sink, corr.main, cacc.main)]
    fast.start()
    for t in ths:
        t.start()
    src = Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)])
    src.start()
    time.sleep(1.0)                                          # everything backs up behind the stalled consumer
    assert all(t.is_alive() for t in ths[1:]) and len(got) == 0
    gate.set()
    for t in [src] + ths + [fast]:
        t.join(30)
        assert not t.is_alive()
    assert len(got) == nlong
    for k, sp in enumerate(got):
        assert np.array_equal(sp, orc.xgpu_correlate(vin[lacc * k:lacc * (k + 1)], S, C))


def test_beamform_takes_a_gulp_that_lies_in_two_ring_spans_without_a_copy():
    """The reference runs Beamform on 2 x the capture gulp (lwa352-pipeline.py:172,279-282: ntime_gulp = GPU_NGULP * GSIZE) out
    of bifrost's circular buffer.  Here the writer's 8-sample spans are taken two at a time as the two windows of one
    16-sample gulp (`read_parts` -> bfBeamformRunParts): every gulp one call, no gathered copy, results those of the oracle
    on the 16-sample gulp; a writer whose spans already hold whole gulps still takes the one-part call."""
    from caltech_bifrost_dsp_amd.blocks import Beamform
    nchan, nstand, nbeam, g = 3, 6, 4, 16
    ninput = nstand * 2
    rng = np.random.default_rng(21)
    vin = rng.integers(0, 256, (4 * g, nchan, ninput), dtype=np.uint8)
    w = (rng.uniform(-1, 1, (nchan, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (nchan, nbeam, ninput))).astype(np.complex64)
    for span_samples, want_parts_calls in ((g // 2, 4), (g, 0)):
        r0, r1 = Ring("gpu-input"), Ring("bf-output")
        be = OracleBackend()
        bf = Beamform(LOG, r0, r1, nchan=nchan, nbeam=nbeam, ninput=ninput, ntime_gulp=g, backend=be)
        bf.gains_cpu[...] = w
        s1 = Sink(r1, g * nchan * nbeam * 8)
        hdr = source_header(nchan, nstand, 2, seq0=0, sfreq=50e6)
        run_blocks([bf], Source(r0, [(hdr, vin, span_samples * nchan * ninput)], wait_readers=1), [s1])
        (_, _, spans), = s1.sequences
        assert len(spans) == 4 and getattr(be, "parts_calls", 0) == want_parts_calls
        for k, sp in enumerate(spans):
            exp = orc.beamform(vin[k * g:(k + 1) * g], w, g, nchan, ninput, nbeam)
            assert np.array_equal(sp.view(np.complex64).reshape(exp.shape), exp), (span_samples, k)


class _AsyncCopyBackend(OracleBackend):
    """the oracle backend with an enqueue-only copy that completes a few polls later (the HIP backend's copy_async /
    copy_done / copy_wait: CorrAcc's publish of a long integration)"""

    def __init__(self):
        super().__init__()
        self.copies = []

    def copy_async(self, dst, src):
        stamp = {"dst": dst, "src": src, "done": False}
        self.copies.append(stamp)
        return stamp

    def _complete(self, stamp):
        if not stamp["done"]:
            stamp["dst"].numpy().reshape(-1)[...] = stamp["src"].numpy().reshape(-1)      # (the copy lands only now)
            stamp["done"] = True

    def copy_done(self, stamp):
        return stamp["done"]

    def copy_wait(self, stamp):
        time.sleep(0.02)                   # (the copy takes a while: the block's thread must not sit here)
        self._complete(stamp)


def test_fused_corracc_publishes_without_stopping():
    """Fused mode: the copy of a finished long integration is only enqueued; a helper waits for it (20 ms here), commits the
    span and hands the accumulator pair back, while CorrAcc goes on following the upstream spans -- also when Corr needs the
    pair again before anything else arrives (three dumps per long integration: the pair of long integration j is wanted
    again two long integrations later).  Every long integration arrives complete and in order, equal to N x the oracle's."""
    C, S, g, acc, lacc = 2, 8, 2, 2, 6
    nlong = 7
    vin = np.random.default_rng(13).integers(0, 256, (nlong * lacc + 2, C, S, 2), dtype=np.uint8)
    be = _AsyncCopyBackend()
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-slow-output")
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, backend=be, accumulate="fused")
    fast, slow = Sink(r1, corr.ogulp_size), Sink(r2, cacc.ogulp_size)
    run_blocks([corr, cacc], Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)]), [fast, slow])
    assert cacc.stats['fused'] is True and len(be.copies) == nlong and all(c["done"] for c in be.copies)
    (_, _, spans), = slow.sequences
    assert len(spans) == nlong
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[lacc * k:lacc * (k + 1)], S, C)), k


def test_grouped_corracc_publishes_without_stopping_and_lets_its_spans_go():
    """Grouped mode (round 5, the block's default on in-repo rings): the dumps of a group are summed in one pass, long integrations
    alternate between two accumulators, and the copy of a finished one is only enqueued (a helper waits for it and commits)
    while the block goes on collecting the next group.  Every long integration arrives complete and in order, equal to N x the
    oracle's; groups are cut at the long-integration boundary (5 dumps, groups of 2: 2 + 2 + 1); no span stays referenced."""
    import gc
    C, S, g, acc, lacc = 2, 8, 2, 2, 10
    nlong = 5
    vin = np.random.default_rng(14).integers(0, 256, (nlong * lacc + 2, C, S, 2), dtype=np.uint8)
    be = _AsyncCopyBackend()
    r0, r1, r2 = Ring("gpu-input"), Ring("corr-output"), Ring("corr-slow-output")
    corr = Corr(LOG, r0, r1, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, backend=be)
    cacc = CorrAcc(LOG, r1, r2, nchan=C, npol=2, nstand=S, acc_len=lacc, autostartat=0, backend=be)
    assert cacc.acc_mode == 'group' and getattr(r1, 'long_accumulator', None) is None
    cacc.group_dumps = 2
    fast, slow = Sink(r1, corr.ogulp_size), Sink(r2, cacc.ogulp_size)
    run_blocks([corr, cacc], Source(r0, [(source_header(C, S, 2), vin, g * C * S * 2)]), [fast, slow])
    assert cacc.stats['grouped'] is True and cacc.stats['fused'] is False and corr.stats['fused_corracc'] is False
    assert len(be.copies) == nlong and all(c["done"] for c in be.copies)
    assert be.sum_calls[:3 * nlong] == [(2, False), (2, True), (1, True)] * nlong
    (_, _, spans), = slow.sequences
    assert len(spans) == nlong
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[lacc * k:lacc * (k + 1)], S, C)), k
    del fast, slow, spans
    gc.collect()
    if hasattr(r1, "info"):
        assert r1.info()["live_bytes"] == 0


def test_packet_slabs_stay_slabs_from_ingest_to_corr_and_beamform():
    """Snap2Ingest(unpack=False) -> a ring of PACKET SLABS -> Corr and Beamform (two slabs per beamformer gulp): the blocks hand
    every slab to the library's slab calls (here: the oracle's unpack + the plain call) with the right sequence number and
    channel offset, the downstream headers describe the products (no layout keys), and the results are those of the unpacked
    pipeline -- also for a window whose packets arrived in another order and one that lost a packet."""
    from caltech_bifrost_dsp_amd.blocks import Beamform, Snap2Ingest
    C, S, P, g, nbeam = 4, 16, 2, 8, 4
    T = 4 * g
    ninput = S * P
    rng = np.random.default_rng(12)
    vin = rng.integers(0, 256, (T, C, S, P), dtype=np.uint8)
    seq0, chan0 = 6000, 192
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=77, nchan_blocks=1, nstand_per_pkt=8, chan0_pipeline=chan0)
    per_win = len(pk) // (T // g)
    wins = [list(pk[w * per_win:(w + 1) * per_win]) for w in range(T // g)]
    wins[1] = [wins[1][i] for i in rng.permutation(per_win)]                  # arrival order is arbitrary
    lost = wins[2][5]
    wins[2][5] = wins[2][6]                                                     # one packet lost, its slot holds a duplicate
    received = vin.copy().reshape(T, C, ninput)
    import struct
    lseq, _, lnpol, _, lnchan, _, _, lc0, lp0 = struct.unpack(orc.SNAP2_HDR, lost[:32])
    received[lseq - seq0, lc0 - chan0:lc0 - chan0 + lnchan, lp0:lp0 + lnpol] = 0
    be = OracleBackend()
    r_pk, r_slab, r_vis, r_beam = Ring("packets"), Ring("gpu-input-slabs"), Ring("corr-output"), Ring("bf-output")
    ing = Snap2Ingest(LOG, r_pk, r_slab, ntime_gulp=g, nchan=C, nstand=S, npol=P, nchan_per_pkt=C, nstand_per_pkt=8, backend=be, unpack=False)
    assert ing.ogulp_size == ing.igulp_size == per_win * len(pk[0])
    hdr_src = source_header(C, S, P, seq0=seq0, chan0=chan0)
    cblk = Corr(LOG, r_slab, r_vis, ntime_gulp=g, nchan=C, npol=P, nstand=S, acc_len=2 * g, autostartat=seq0,
                ant_to_input=hdr_src['ant_to_input'], backend=be)
    bf = Beamform(LOG, r_slab, r_beam, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=2 * g, backend=be)
    w = (rng.uniform(-1, 1, (C, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (C, nbeam, ninput))).astype(np.complex64)
    bf.gains_cpu[...] = w
    vis_sink, beam_sink = Sink(r_vis, cblk.ogulp_size), Sink(r_beam, 2 * g * C * nbeam * 8)
    slabs = b"".join(b"".join(wn) for wn in wins)
    run_blocks([ing, cblk, bf], Source(r_pk, [({'seq0': seq0, 'chan0': chan0, 'sync_time': 77}, slabs, ing.igulp_size)]), [vis_sink, beam_sink])
    assert be.slab_calls == T // g and be.beam_slab_calls == T // (2 * g)
    (vh, _, vspans), = vis_sink.sequences
    (bh, _, bspans), = beam_sink.sequences
    for h in (vh, bh):
        assert not any(k in h for k in ('layout', 'slab_ntime', 'npkt_per_gulp', 'pkt_stride')) and h['chan0'] == chan0
    assert len(vspans) == 2 and len(bspans) == 2
    for k in range(2):
        exp = orc.xgpu_correlate(received[2 * k * g:(2 * k + 2) * g].reshape(2 * g, C, S, P), S, C)
        assert np.array_equal(vspans[k].view(np.int32), exp.ravel()), k
        expb = orc.beamform(received[2 * k * g:(2 * k + 2) * g], w, 2 * g, C, ninput, nbeam)
        assert np.array_equal(bspans[k].view(np.complex64).reshape(expb.shape), expb), k
