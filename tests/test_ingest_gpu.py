"""Ingest on the GPU: xengSnap2Unpack against the oracle's restatement of the reference SNAP2 emulator
(test_tx_vectors.py:79-112), and Snap2Ingest -> Corr on device rings."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import ffi  # noqa: E402
from caltech_bifrost_dsp_amd.blocks import Corr, Snap2Ingest  # noqa: E402
from caltech_bifrost_dsp_amd.ring import Ring  # noqa: E402
from oracle import xeng_oracle as orc  # noqa: E402
from tests.pipeline_util import LOG, Sink, Source, run_blocks, source_header  # noqa: E402


def _unpack(slab, npkt, stride, seq0, ntime, chan0, nchan, ninput, clear=1, out=None):
    dpk = ffi.DeviceBuffer(max(len(slab), 1)).upload(np.frombuffer(slab, dtype=np.uint8))
    dout = out if out is not None else ffi.DeviceBuffer(ntime * nchan * ninput)
    placed, dropped = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengSnap2Unpack", dpk.ptr, npkt, stride, dout.ptr, seq0, ntime, chan0, nchan, ninput, clear,
             ctypes.byref(placed), ctypes.byref(dropped))
    got = dout.download(np.uint8).reshape(ntime, nchan, ninput)
    dpk.free()
    if out is None:
        dout.free()
    return got, placed.value, dropped.value


@pytest.mark.parametrize("T,C,S,nchan_blocks,nstand_per_pkt", [
    (6, 8, 64, 2, 32),        # the deployed packet shape: 32 stands x 2 pol = 64 bytes per channel row (16-byte path)
    (5, 6, 12, 3, 3),         # 6-byte rows: byte path
    (480, 96, 352, 1, 32),    # config 2: one gulp, 5280 packets of 6176 bytes
])
def test_unpack_matches_oracle(T, C, S, nchan_blocks, nstand_per_pkt):
    rng = np.random.default_rng(T + S)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    seq0, chan0 = 10 ** 12 + 5, 1000                                   # sequence numbers beyond 32 bits
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=9, nchan_blocks=nchan_blocks, nstand_per_pkt=nstand_per_pkt,
                           chan0_pipeline=chan0)
    stride = len(pk[0])
    order = rng.permutation(len(pk))
    got, placed, dropped = _unpack(b"".join(pk[i] for i in order), len(pk), stride, seq0, T, chan0, C, S * 2)
    assert placed == len(pk) and dropped == 0
    assert np.array_equal(got.reshape(vin.shape), vin)


def test_unpack_reference_transmitter_fixture(golden_dir):
    """Bytes the reference's own F-engine emulator sends (tests/golden/snap2_*.bin, recorded from
    test_transmitters/test_tx_vectors.py by oracle/make_golden_snap2.py) -> xengSnap2Unpack == the reference-generated
    input file those packets were cut from (tests/golden/in_8t_4c_64s_2p_deadbeef.dat)."""
    import json
    import os
    with open(os.path.join(golden_dir, "snap2_8t_4c_64s_2p_deadbeef.bin"), "rb") as fh:
        meta = json.loads(fh.readline().decode())
        blob = fh.read()
    with open(os.path.join(golden_dir, "in_8t_4c_64s_2p_deadbeef.dat"), "rb") as fh:
        hdr = json.loads(fh.readline().decode())
        vin = np.frombuffer(fh.read(), dtype=np.uint8).reshape(hdr["shape"])
    T, C, S, P = vin.shape
    got, placed, dropped = _unpack(blob, meta["npkt"], meta["pkt_bytes"], 0, T, 0, C, S * P)
    assert placed == meta["npkt"] and dropped == 0
    assert np.array_equal(got.reshape(vin.shape), vin)
    # the same packets in reverse order, into a window that starts two spectra later: the first two spectra drop out
    n, b = meta["npkt"], meta["pkt_bytes"]
    rev = b"".join(blob[k * b:(k + 1) * b] for k in reversed(range(n)))
    got2, placed2, dropped2 = _unpack(rev, n, b, 2, T, 0, C, S * P)
    per_seq = n // T
    assert placed2 == n - 2 * per_seq and dropped2 == 2 * per_seq
    assert np.array_equal(got2.reshape(vin.shape)[:T - 2], vin[2:]) and not got2.reshape(vin.shape)[T - 2:].any()


def test_unpack_zero_fill_only_when_packets_are_missing():
    """clear=1 on a gulp that holds stale bytes: a complete slab overwrites every byte (no zero-fill pass is needed, and
    none of the stale bytes may survive); with packets lost, duplicated or off the packet grid the missing samples read
    as zero, never as stale data."""
    T, C, S = 16, 8, 64
    rng = np.random.default_rng(11)
    vin = rng.integers(1, 256, (T, C, S, 2), dtype=np.uint8)
    pk = orc.snap2_packets(vin, seq0=5, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=0)
    stride = len(pk[0])
    out = ffi.DeviceBuffer(T * C * S * 2)

    def stale():
        ffi.call("xengMemset", out.ptr, 0xAA, out.nbytes)

    stale()
    got, placed, dropped = _unpack(b"".join(pk), len(pk), stride, 5, T, 0, C, S * 2, out=out)
    assert placed == len(pk) and dropped == 0 and np.array_equal(got.reshape(vin.shape), vin)
    for lost, extra in (([3, 40], []), ([], [pk[7]]), ([9], [pk[9 + 1], pk[2]])):        # loss; duplicate; loss + duplicates
        stale()
        sel = [p for i, p in enumerate(pk) if i not in lost] + extra
        exp, ep, ed = orc.snap2_unpack(sel, 5, T, 0, C, S * 2)
        got, placed, dropped = _unpack(b"".join(sel), len(sel), stride, 5, T, 0, C, S * 2, out=out)
        assert (placed, dropped) == (ep, ed) and np.array_equal(got, exp)
    # a second geometry in the same rows (whole-band packets on top of half-band ones): irregular -> blank + rescatter
    stale()
    wide = orc.snap2_packets(vin[:2], seq0=5, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=0)
    pad = [p + bytes(len(wide[0]) - len(p)) for p in pk[8:]]                              # same stride for both kinds
    sel = wide + pad
    exp, ep, ed = orc.snap2_unpack(sel, 5, T, 0, C, S * 2)
    got, placed, dropped = _unpack(b"".join(sel), len(sel), len(wide[0]), 5, T, 0, C, S * 2, out=out)
    assert (placed, dropped) == (ep, ed) and np.array_equal(got, exp)
    out.free()


def test_async_unpack_counts_its_own_drops():
    """xengSnap2UnpackAsync keeps its drops in a counter of its own: xengSnap2GetAsyncDrops reads and clears it, and a
    synchronous call in between neither sees nor disturbs it."""
    T, C, S = 8, 8, 64
    rng = np.random.default_rng(12)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    pk = orc.snap2_packets(vin, seq0=100, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=0)
    stride = len(pk[0])
    stray = [orc.snap2_packets(vin[:1], seq0=100 + T, nchan_blocks=2)[0], orc.snap2_packets(vin[:1], seq0=99, nchan_blocks=2)[0], bytes(stride)]
    slab = b"".join(pk + stray)
    dpk = ffi.DeviceBuffer(len(slab)).upload(np.frombuffer(slab, dtype=np.uint8))
    dout = ffi.DeviceBuffer(T * C * S * 2)
    n = ctypes.c_int(-1)
    ffi.call("xengSnap2GetAsyncDrops", ctypes.byref(n))
    ffi.call("xengSnap2UnpackAsync", dpk.ptr, len(pk) + 3, stride, dout.ptr, 100, T, 0, C, S * 2, 1)
    ffi.call("xengSnap2UnpackAsync", dpk.ptr, len(pk) + 3, stride, dout.ptr, 100, T, 0, C, S * 2, 1)
    got, placed, dropped = _unpack(slab, len(pk) + 3, stride, 100, T, 0, C, S * 2)       # synchronous call in between
    assert (placed, dropped) == (len(pk), 3)
    ffi.call("xengSnap2GetAsyncDrops", ctypes.byref(n))
    assert n.value == 6
    ffi.call("xengSnap2GetAsyncDrops", ctypes.byref(n))
    assert n.value == 0
    assert np.array_equal(dout.download(np.uint8).reshape(vin.shape), vin)
    dpk.free(); dout.free()


def test_unpack_drops_and_blanks():
    T, C, S = 8, 8, 64
    rng = np.random.default_rng(1)
    vin = rng.integers(1, 256, (T, C, S, 2), dtype=np.uint8)
    seq0, chan0 = 777, 96
    pk = orc.snap2_packets(vin, seq0=seq0, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=chan0)
    stride = len(pk[0])
    stray = [orc.snap2_packets(vin[:1], seq0=seq0 + T, nchan_blocks=2, chan0_pipeline=chan0)[0],      # next window
             orc.snap2_packets(vin[:1], seq0=seq0 - 1, nchan_blocks=2, chan0_pipeline=chan0)[0],      # previous window
             orc.snap2_packets(vin[:1], seq0=seq0, nchan_blocks=2, chan0_pipeline=chan0 + C)[0],      # other channels
             bytes(stride)]                                                                             # empty slot
    lost = {2, 17}
    mixed = [p for i, p in enumerate(pk) if i not in lost] + stray + [pk[4]]                           # + a duplicate
    mixed = [mixed[i] for i in rng.permutation(len(mixed))]
    exp, eplaced, edropped = orc.snap2_unpack(mixed, seq0, T, chan0, C, S * 2)
    got, placed, dropped = _unpack(b"".join(mixed), len(mixed), stride, seq0, T, chan0, C, S * 2)
    assert (placed, dropped) == (eplaced, edropped) == (len(pk) - 2 + 1, 4)
    assert np.array_equal(got, exp)
    assert (got == 0).sum() == 2 * (C // 2) * 64                      # exactly the two lost packets' samples
    # clear=0 leaves what is already in the gulp (a second slab of the same window completes it)
    dout = ffi.DeviceBuffer(T * C * S * 2)
    _unpack(b"".join(mixed), len(mixed), stride, seq0, T, chan0, C, S * 2, clear=1, out=dout)
    late = [pk[i] for i in sorted(lost)]
    got2, placed2, _ = _unpack(b"".join(late), 2, stride, seq0, T, chan0, C, S * 2, clear=0, out=dout)
    assert placed2 == 2 and np.array_equal(got2.reshape(vin.shape), vin)
    dout.free()
    with pytest.raises(ffi.XengError):
        ffi.call("xengSnap2Unpack", None, 1, stride, None, 0, T, 0, C, S * 2, 1, None, None)


def test_ingest_corr_on_device_rings():
    """pinned packet slabs -> Snap2Ingest (H2D of raw packets + device scatter) -> gpu-input (cuda) -> Corr."""
    T, C, S, g, acc = 128, 8, 64, 32, 64
    rng = np.random.default_rng(12)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    seq0, chan0 = 6400, 192
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=5, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=chan0)
    per_win = len(pk) // (T // g)
    slabs = b"".join(b"".join(pk[w * per_win + i] for i in rng.permutation(per_win)) for w in range(T // g))
    r_pk, r_in, r_vis = Ring("packets", space="cuda_host"), Ring("gpu-input", space="cuda"), Ring("corr-output", space="cuda")
    ing = Snap2Ingest(LOG, r_pk, r_in, ntime_gulp=g, nchan=C, nstand=S, npol=2, nchan_per_pkt=C // 2, nstand_per_pkt=32, gpu=0)
    hdr = source_header(C, S, 2)
    corr = Corr(LOG, r_in, r_vis, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=seq0, gpu=0,
                ant_to_input=hdr['ant_to_input'])
    sink = Sink(r_vis, corr.ogulp_size)
    run_blocks([ing, corr], Source(r_pk, [({'seq0': seq0, 'chan0': chan0, 'sync_time': 5}, slabs, ing.igulp_size)]), [sink])
    (h, _, spans), = sink.sequences
    assert h['seq0'] == seq0 and h['chan0'] == chan0 and len(spans) == T // acc
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), orc.xgpu_correlate(vin[k * acc:(k + 1) * acc], S, C))
    assert ing.stats['packets_placed'] == len(pk) and ing.stats['missing_frac'] == 0.0


@pytest.mark.parametrize("tables", [None, "1"])
def test_ingest_leaves_slabs_and_corr_and_beamform_read_them_in_place(tables, monkeypatch):
    """(tables "1": XENG_SLAB_TABLES=1 -- both consumers follow their offset tables / packet indices from the first launch on.)
    pinned packet slabs -> Snap2Ingest(unpack=False): one H2D per slab, nothing else -> a device ring of SLABS -> Corr
    (xengXgpuKernelAsyncSlab, with a CorrAcc fed from its dumps) and Beamform (xengBeamformRunSlabs, two slabs per gulp) ->
    BeamformSumBeams.  Regular windows are read where they lie; one window arrives in another order and one has lost a packet
    (scattered on the device).  Every product equals the oracle on what was received."""
    import ctypes
    import os
    import struct
    from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, CorrAcc
    if os.environ.get("XENG_RAW") == "0":
        pytest.skip("packet slabs need the fused contraction kernel (the two-pass X-engine refuses them)")
    if tables is not None:
        monkeypatch.setenv("XENG_SLAB_TABLES", tables)
    else:
        monkeypatch.delenv("XENG_SLAB_TABLES", raising=False)
    T, C, S, g, acc, nbeam, ntime_sum = 768, 8, 64, 96, 192, 32, 16           # (gulps of 96 samples: the fused contraction kernel)
    ninput = S * 2
    rng = np.random.default_rng(14)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    seq0, chan0 = 9600, 192                                   # (a start time on a gulp boundary)
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=5, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=chan0)
    per_win = len(pk) // (T // g)
    wins = [list(pk[w * per_win:(w + 1) * per_win]) for w in range(T // g)]
    wins[2] = [wins[2][i] for i in rng.permutation(per_win)]
    lost = wins[5][9]
    wins[5][9] = wins[5][10]
    received = vin.copy().reshape(T, C, ninput)
    lseq, _, lnpol, _, lnchan, _, _, lc0, lp0 = struct.unpack(orc.SNAP2_HDR, lost[:32])
    received[lseq - seq0, lc0 - chan0:lc0 - chan0 + lnchan, lp0:lp0 + lnpol] = 0
    slabs = b"".join(b"".join(wn) for wn in wins)
    r_pk, r_slab = Ring("packets", space="cuda_host"), Ring("gpu-input-slabs", space="cuda")
    r_vis, r_slow, r_beam, r_pow = Ring("corr-output", space="cuda"), Ring("corr-slow", space="cuda_host"), Ring("bf-output", space="cuda"), Ring("bf-pow", space="cuda_host")
    ing = Snap2Ingest(LOG, r_pk, r_slab, ntime_gulp=g, nchan=C, nstand=S, npol=2, nchan_per_pkt=C, nstand_per_pkt=32, gpu=0, unpack=False,
                      buffer_multiplier=8)
    hdr = source_header(C, S, 2)
    corr = Corr(LOG, r_slab, r_vis, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=seq0, gpu=0, ant_to_input=hdr['ant_to_input'])
    cacc = CorrAcc(LOG, r_vis, r_slow, nchan=C, npol=2, nstand=S, acc_len=2 * acc, autostartat=seq0, gpu=0)
    bf = Beamform(LOG, r_slab, r_beam, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=2 * g, gpu=0)
    w = (rng.uniform(-1, 1, (C, nbeam, ninput)) + 1j * rng.uniform(-1, 1, (C, nbeam, ninput))).astype(np.complex64)
    bf.gains_cpu[...] = w
    sb = BeamformSumBeams(LOG, r_beam, r_pow, nchan=C, ntime_gulp=2 * g, ntime_sum=ntime_sum, gpu=0)
    vis_sink, slow_sink, pow_sink = Sink(r_vis, corr.ogulp_size), Sink(r_slow, corr.ogulp_size), Sink(r_pow, (nbeam // 2) * (2 * g // ntime_sum) * C * 16)
    run_blocks([ing, corr, cacc, bf, sb], Source(r_pk, [({'seq0': seq0, 'chan0': chan0, 'sync_time': 5}, slabs, ing.igulp_size)]),
               [vis_sink, slow_sink, pow_sink])
    (h, _, spans), = vis_sink.sequences
    assert h['seq0'] == seq0 and h['chan0'] == chan0 and 'layout' not in h and len(spans) == T // acc
    exp_vis = [orc.xgpu_correlate(received[k * acc:(k + 1) * acc].reshape(acc, C, S, 2), S, C) for k in range(T // acc)]
    for k, sp in enumerate(spans):
        assert np.array_equal(sp.view(np.int32), exp_vis[k].ravel()), k
    (_, _, slow), = slow_sink.sequences
    assert len(slow) == T // (2 * acc)
    for k, sp in enumerate(slow):
        assert np.array_equal(sp.view(np.int32), (exp_vis[2 * k] + exp_vis[2 * k + 1]).ravel()), k
    (_, _, pows), = pow_sink.sequences
    assert len(pows) == T // (2 * g)
    for k, sp in enumerate(pows):
        beams = orc.beamform(received[2 * k * g:(2 * k + 2) * g], w, 2 * g, C, ninput, nbeam)
        exp = orc.beamform_integrate(beams, ntime_sum)
        got = sp.view(np.float32).reshape(exp.shape)
        assert np.max(np.abs(got - exp)) <= 2e-5 * np.sqrt(np.mean(exp[..., :2] ** 2)), k
    nfx, nix, nfb, nib = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
    ffi.call("xengXgpuGetSlabStats", ctypes.byref(nfx), ctypes.byref(nix))
    ffi.call("xengBeamformGetSlabStats", ctypes.byref(nfb), ctypes.byref(nib))
    # windows 2 and 5 are irregular: each consumer scatters the first and, if its host side has seen the device's hint by then, reads
    # the next where it lies -- through an offset table (X-engine) / a packet index (beamformer), round 5; the other six are read in
    # place by both
    if tables == "1":
        assert (nfx.value, nix.value, nfb.value, nib.value) == (0, 2, 0, 2)
    else:
        assert nfx.value + nix.value == 2 and nfx.value >= 1 and nfb.value + nib.value == 2 and nfb.value >= 1


def test_stamp_seq_rewrites_the_sequence_numbers_only():
    """xengSnap2StampSeq (the emulator's side of a receiver that reuses its slab buffers): packet p gets seq0 + p / pkts_per_seq in
    its first eight header bytes, big-endian; every other byte of the slab is what it was; the stamped slab unpacks as the window
    that starts at seq0."""
    import struct
    T, C, S = 96, 8, 64
    rng = np.random.default_rng(4)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    pk = orc.snap2_packets(vin, seq0=5, sync_time=9, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=0)
    stride, nblk = len(pk[0]) + 32, S // 32              # (a stride with slack between the packets: the slack stays untouched too)
    raw = np.full((len(pk), stride), 0xA5, dtype=np.uint8)
    for i, p in enumerate(pk):
        raw[i, :len(p)] = np.frombuffer(p, dtype=np.uint8)
    d = ffi.DeviceBuffer(raw.nbytes).upload(raw)
    seq0 = 10 ** 13 + 77
    ffi.call("xengSnap2StampSeq", d.ptr, len(pk), stride, seq0, nblk)
    got = d.download(np.uint8).reshape(len(pk), stride)
    want = raw.copy()
    for i in range(len(pk)):
        want[i, :8] = np.frombuffer(struct.pack(">Q", seq0 + i // nblk), dtype=np.uint8)
    assert np.array_equal(got, want)
    out = ffi.DeviceBuffer(vin.size)
    placed = ctypes.c_int()
    ffi.call("xengSnap2Unpack", d.ptr, len(pk), stride, out.ptr, seq0, T, 0, C, S * 2, 1, ctypes.byref(placed), None)
    assert placed.value == len(pk) and np.array_equal(out.download(np.uint8), vin.reshape(-1))
    with pytest.raises(ffi.XengError):
        ffi.call("xengSnap2StampSeq", d.ptr, len(pk), stride, seq0, 0)
    d.free()
    out.free()


def test_unpack_async_is_ordered_before_the_contraction():
    """xengSnap2UnpackAsync + xengXgpuKernelAsync: the scatter runs on the X-engine's staging stream, so the dump's
    contraction reads complete gulps; visibilities equal the oracle on the original voltages."""
    T, C, S, g = 288, 8, 32, 96
    rng = np.random.default_rng(3)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    pk = orc.snap2_packets(vin, seq0=0, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=0)
    stride, per_win = len(pk[0]), len(pk) // (T // g)
    ffi.call("xengXgpuConfigure", S, 2, C, g, 0)
    ffi.call("xengXgpuInitialize", 0)
    matlen = orc.per_chan(S) * C
    out = ffi.DeviceBuffer(matlen * 8)
    slabs, gulps = [], []
    for w in range(T // g):
        slab = b"".join(pk[w * per_win + i] for i in rng.permutation(per_win))
        slabs.append(ffi.DeviceBuffer(len(slab)).upload(np.frombuffer(slab, dtype=np.uint8)))
        gulps.append(ffi.DeviceBuffer(g * C * S * 2))
        ffi.call("xengMemset", gulps[-1].ptr, 0xEE, gulps[-1].nbytes)
    for w in range(T // g):
        ffi.call("xengSnap2UnpackAsync", slabs[w].ptr, per_win, stride, gulps[w].ptr, w * g, g, 0, C, S * 2, 1)
        ffi.call("xengXgpuKernelAsync", gulps[w].ptr, out.ptr, int(w == T // g - 1))
    ffi.call("xengXgpuSync")
    assert np.array_equal(out.download(np.int32), orc.xgpu_correlate(vin, S, C))
    ffi.call("xengXgpuDestroy")


def test_unpack_call_state_is_clean_for_the_next_call():
    """The synchronous call's kernel reports for itself (its last work-group checks the coverage, writes {drops, complete}
    to pinned memory and clears the device-side state with atomics): sixty calls in a row with changing loss, foreign and
    window patterns -- including slabs large enough for a multi-level completion ticket -- each give the oracle's gulp and
    counts, i.e. nothing of one call's coverage, drop count or ticket survives into the next."""
    T, C, S = 24, 16, 128
    rng = np.random.default_rng(2025)
    vin = rng.integers(1, 256, (T, C, S, 2), dtype=np.uint8)
    pk = orc.snap2_packets(vin, seq0=77, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=0)      # 24 x 2 x 4 = 192 packets
    stride = len(pk[0])
    foreign = orc.snap2_packets(vin[:2], seq0=10 ** 6, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=0)    # outside the window
    out = ffi.DeviceBuffer(vin.nbytes)
    for k in range(60):
        kind = k % 4
        if kind == 0:
            sel = list(pk)                                               # complete
        elif kind == 1:
            lost = set(rng.choice(len(pk), size=int(rng.integers(1, 9)), replace=False).tolist())
            sel = [p for i, p in enumerate(pk) if i not in lost]         # loss -> second pass
        elif kind == 2:
            sel = list(pk) + foreign[:int(rng.integers(1, 6))]           # complete, plus dropped strangers
        else:
            sel = [pk[i] for i in rng.permutation(len(pk))]              # complete, shuffled
        ffi.call("xengMemset", out.ptr, 0xEE, out.nbytes)
        exp, ep, ed = orc.snap2_unpack(sel, 77, T, 0, C, S * 2)
        got, placed, dropped = _unpack(b"".join(sel), len(sel), stride, 77, T, 0, C, S * 2, out=out)
        assert (placed, dropped) == (ep, ed), (k, kind, placed, dropped, ep, ed)
        assert np.array_equal(got, exp), (k, kind)
    out.free()
