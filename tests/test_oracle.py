"""Pin the CPU oracle (oracle/xeng_oracle.c) against the reference's golden vectors.

The golden files in tests/golden/ were produced by the reference's own
pipeline/verification/make_golden_inputs.py (see oracle/make_golden.py) and by
the reference's SoftwareBf functions (oracle/make_golden_beamform.py).
"""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import xeng_oracle as orc


def load_dat(path):
    with open(path, "rb") as fh:
        meta = json.loads(fh.readline().decode())
        raw = fh.read()
    dt = np.uint8 if "uint8" in meta["dtype"] else np.complex128
    return meta, np.frombuffer(raw, dtype=dt).reshape(meta["shape"])


def golden_sets(golden_dir):
    out = []
    for tag in ("deadbeef", "chanramp"):
        mi, vin = load_dat(os.path.join(golden_dir, "in_8t_4c_16s_2p_%s.dat" % tag))
        mc, corr = load_dat(os.path.join(golden_dir, "corr_8t_4a_4c_16s_2p_%s.dat" % tag))
        out.append((tag, vin, np.round(corr.real).astype(np.int64), np.round(corr.imag).astype(np.int64), mc["acc_len"]))
    z = np.load(os.path.join(golden_dir, "golden_64t_32a_8c_32s_2p_deadbeef.npz"))
    out.append(("64in", z["vin"], z["corr_re"].astype(np.int64), z["corr_im"].astype(np.int64), 32))
    return out


def test_decode_all_bytes():
    b = np.arange(256, dtype=np.uint8)
    re = np.empty(256, np.int8)
    im = np.empty(256, np.int8)
    orc.lib().orc_decode(b.ctypes.data_as(ctypes.c_void_p), re.ctypes.data_as(ctypes.c_void_p),
                         im.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(256))
    nre, nim = orc.decode(b)
    assert np.array_equal(re, nre) and np.array_equal(im, nim)
    assert re.min() == -8 and re.max() == 7 and im.min() == -8 and im.max() == 7
    assert re[0x88] == -8 and im[0x88] == -8 and re[0x7f] == 7 and im[0x7f] == -1


def test_regtile_index_c_vs_numpy():
    for nstand in (4, 16, 32, 352):
        rng = np.random.default_rng(nstand)
        for _ in range(2000):
            i0, i1 = sorted(rng.integers(0, 2 * nstand, 2))
            assert orc.lib().orc_regtile_index(int(i0), int(i1), nstand) == orc.regtile_index(int(i0), int(i1), nstand)
        assert orc.lib().orc_xgpu_per_chan(nstand, 2) == orc.per_chan(nstand)
    assert orc.per_chan(352) == 249216 and orc.per_chan(16) == 576


def test_regtile_addresses_are_unique_and_count():
    # measured property: 4*(n/2) words per plane per channel are never addressed
    nstand = 32
    s0, s1, p0, p1 = np.meshgrid(np.arange(nstand), np.arange(nstand), [0, 1], [0, 1], indexing="ij")
    m = s1 >= s0
    idx = orc.regtile_index(2 * s0 + p0, 2 * s1 + p1, nstand)[m]
    assert len(np.unique(idx)) == len(idx)
    assert orc.per_chan(nstand) - len(idx) == 4 * (nstand // 2)
    assert idx.max() < orc.per_chan(nstand)


def test_oracle_matches_reference_golden(golden_dir):
    """C oracle -> GetOrder -> Reorder == make_golden_inputs.py output, exactly
    (the check corr_output_full_block.py:550-603 does, for s1 >= s0)."""
    for tag, vin, gre, gim, acc_len in golden_sets(golden_dir):
        T, C, S, P = vin.shape
        a2i = np.arange(S * P, dtype=np.int32).reshape(S, P)
        bl, cj = orc.xgpu_get_order(a2i)
        for blk in range(T // acc_len):
            planar = orc.xgpu_correlate(vin[blk * acc_len:(blk + 1) * acc_len], S, C)
            ro = orc.xgpu_reorder(planar, bl, cj, C)          # [s0,s1,p0,p1,c,2]
            for s0 in range(S):
                for s1 in range(s0, S):
                    assert np.array_equal(ro[s0, s1, :, :, :, 0], np.moveaxis(gre[blk, :, s0, s1], 0, -1)), (tag, s0, s1)
                    assert np.array_equal(ro[s0, s1, :, :, :, 1], np.moveaxis(gim[blk, :, s0, s1], 0, -1)), (tag, s0, s1)


def test_oracle_matches_xgpu_test_convention(golden_dir):
    """xgpu_test.py:99-133: buffer[regtile_index(2*s0+p0, 2*s1+p1)] == sum conj(x[s0,p0]) * x[s1,p1]."""
    for tag, vin, gre, gim, acc_len in golden_sets(golden_dir):
        T, C, S, P = vin.shape
        planar = orc.xgpu_correlate(vin[:acc_len], S, C)
        re, im = orc.xgpu_lookup_numpy(planar, S, C)
        m = (np.arange(S)[:, None] <= np.arange(S)[None, :])[None, :, :, None, None]
        # golden = x0 conj(x1); stored = conj(x0) x1 = conj(golden)
        assert np.array_equal(re * m, gre[0] * m), tag
        assert np.array_equal(im * m, -gim[0] * m), tag


def test_accumulate_then_dump_semantics(golden_dir):
    """xgpu_test.py:76-83: G calls, dump on the last == one call over G*ntime samples."""
    tag, vin, gre, gim, acc_len = golden_sets(golden_dir)[2]
    T, C, S, P = vin.shape
    whole = orc.xgpu_correlate(vin, S, C)
    acc = None
    for g in range(4):
        acc = orc.xgpu_correlate(vin[g * (T // 4):(g + 1) * (T // 4)], S, C, acc)
    assert np.array_equal(whole, acc)


def test_numpy_golden_restatement(golden_dir):
    for tag, vin, gre, gim, acc_len in golden_sets(golden_dir):
        rr, ii = orc.golden_corr(vin[:acc_len])
        assert np.array_equal(rr, gre[0]) and np.array_equal(ii, gim[0])


def test_input_permutation_get_order(golden_dir):
    """GetOrder with a shuffled antpol_to_input still reorders to golden of the permuted inputs."""
    tag, vin, gre, gim, acc_len = golden_sets(golden_dir)[0]
    T, C, S, P = vin.shape
    rng = np.random.default_rng(5)
    perm = rng.permutation(S * P).astype(np.int32).reshape(S, P)   # [s,p] -> correlator input
    planar = orc.xgpu_correlate(vin[:acc_len], S, C)
    bl, cj = orc.xgpu_get_order(perm)
    ro = orc.xgpu_reorder(planar, bl, cj, C)
    flat_re = gre[0].transpose(0, 1, 3, 2, 4).reshape(C, S * P, S * P)   # [c, in0, in1]
    flat_im = gim[0].transpose(0, 1, 3, 2, 4).reshape(C, S * P, S * P)
    for s0 in range(S):
        for s1 in range(S):
            for p0 in range(P):
                for p1 in range(P):
                    i0, i1 = perm[s0, p0], perm[s1, p1]
                    assert np.array_equal(ro[s0, s1, p0, p1, :, 0], flat_re[:, i0, i1])
                    assert np.array_equal(ro[s0, s1, p0, p1, :, 1], flat_im[:, i0, i1])


def test_subselect_matches_golden(golden_dir):
    """test_corr_part_rx.py:49-85: subselected, channel-summed vis == golden[s0,s1,p0,p1] summed."""
    tag, vin, gre, gim, acc_len = golden_sets(golden_dir)[2]
    T, C, S, P = vin.shape
    planar = orc.xgpu_correlate(vin[:acc_len], S, C)
    bl, cj = orc.xgpu_get_order(np.arange(S * P, dtype=np.int32).reshape(S, P))
    rng = np.random.default_rng(9)
    sel = [((int(a), int(b)), (int(c), int(d))) for a, b, c, d in
           zip(rng.integers(0, S, 50), rng.integers(0, 2, 50), rng.integers(0, S, 50), rng.integers(0, 2, 50))]
    vismap = np.array([bl[s0, s1, p0, p1] for (s0, p0), (s1, p1) in sel], dtype=np.int32)
    conj = np.array([cj[s0, s1, p0, p1] for (s0, p0), (s1, p1) in sel], dtype=np.int32)
    out = orc.xgpu_subselect(planar, vismap, conj, C, 4, S)
    for v, ((s0, p0), (s1, p1)) in enumerate(sel):
        assert np.array_equal(out[:, v, 0], gre[0][:, s0, s1, p0, p1].reshape(C // 4, 4).sum(1))
        assert np.array_equal(out[:, v, 1], gim[0][:, s0, s1, p0, p1].reshape(C // 4, 4).sum(1))


def test_map_i32():
    rng = np.random.default_rng(1)
    a = rng.integers(-2**31, 2**31 - 1, 1000, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31 - 1, 1000, dtype=np.int64).astype(np.int32)
    exp = (a.astype(np.int64) + b.astype(np.int64)).astype(np.int32)   # wraps
    a2 = a.copy()
    orc.map_i32(a2, b, add=True)
    assert np.array_equal(a2, exp)
    orc.map_i32(a2, b, add=False)
    assert np.array_equal(a2, b)


@pytest.mark.parametrize("tag", ["small", "tile"])
def test_beamform_matches_reference_functions(golden_dir, tag):
    """oracle beamformer / power sums vs outputs of the reference's own SoftwareBf functions
    (which accumulate in complex64, so the tolerance is the reference's own,
    beamformer_test.py:109 / beamformer_sum_test.py:104, scaled to the data)."""
    z = np.load(os.path.join(golden_dir, "beamform_%s.npz" % tag))
    vin, w, beams, power = z["vin"], z["weights"], z["beams"], z["power"]
    ntime, nchan, ninput = vin.shape
    nbeam = w.shape[1]
    re, im = orc.decode(vin)
    assert np.array_equal((re + 1j * im).astype(np.complex64), z["decoded"])
    ob = orc.beamform(vin, w, ntime, nchan, ninput, nbeam)
    scale = np.sqrt(np.mean(np.abs(beams) ** 2))
    assert np.max(np.abs(ob - beams)) / scale < 1e-5
    assert np.all(np.isclose(ob, beams, rtol=1e-4, atol=1e-4 * scale))
    op = orc.beamform_integrate(beams, int(z["ntime_sum"]))
    assert op.shape == power.shape
    assert np.all(np.isclose(op, power, rtol=1e-5, atol=1e-4))
    for b in range(nbeam // 2):
        assert np.array_equal(orc.beamform_integrate_single(beams, int(z["ntime_sum"]), b), op[b])


def test_packet_payloads_follow_the_sending_order():
    """corr_output_full_block.py:461-467 / :512-519 on a reordered matrix whose entries encode their own index."""
    nstand, nchan = 6, 5
    r = np.zeros((nstand, nstand, 2, 2, nchan, 2), dtype=np.int32)
    idx = np.indices(r.shape)
    r[...] = idx[0] * 100000 + idx[1] * 10000 + idx[2] * 1000 + idx[3] * 100 + idx[4] * 10 + idx[5]
    py = orc.corr_packet_payloads(r, False)
    cor = orc.corr_packet_payloads(r, True)
    nbl = nstand * (nstand + 1) // 2
    assert py.shape == cor.shape == (nbl, 4 * nchan * 2)
    k = 0
    for s0 in range(nstand):
        for s1 in range(s0, nstand):
            assert py[k].tobytes() == r[s0, s1].tobytes()                       # send_packets_py payload
            assert np.array_equal(cor[k].reshape(nchan, 2, 2, 2), r[s0, s1].transpose(2, 0, 1, 3))
            k += 1
    h = orc.corr_packet_header_py(1600000000, 2400, 2.3e6, 5.0e7, 2400, nchan, 96, 2, 3, 5)
    assert len(h) == 56                                                           # docs/source/outputs.rst:29
    assert h[:8] == (1600000000).to_bytes(8, "big") and h[-8:] == (3).to_bytes(4, "big") + (5).to_bytes(4, "big")


def test_snap2_packets_follow_the_reference_emulator_and_round_trip(golden_dir):
    """test_tx_vectors.py:79-112 on the reference's own golden input file: packet count, header fields, payload
    slices; unpack(packets) restores the array; lost / late / foreign packets are dropped and leave zeros."""
    import struct
    with open(os.path.join(golden_dir, "in_8t_4c_16s_2p_deadbeef.dat"), "rb") as fh:
        meta = json.loads(fh.readline().decode())
        vin = np.frombuffer(fh.read(), dtype=np.uint8).reshape(meta["shape"])
    T, C, S, P = vin.shape
    pk = orc.snap2_packets(vin, seq0=1000, sync_time=77, nchan_blocks=2, nstand_per_pkt=8, chan0_pipeline=192)
    assert len(pk) == T * 2 * (S // 8) and all(len(p) == 32 + (C // 2) * 8 * P for p in pk)
    # third packet of sequence 1001: channel block 0, stands 16.. (third block of 8) -- there are only 2: so block (1, 0)
    h = struct.unpack(orc.SNAP2_HDR, pk[2 * (S // 8) + 2][:32])
    assert h == (1001, 77, 8 * P, S * P, C // 2, C, 1, 192 + C // 2, 0)
    assert pk[2 * (S // 8) + 2][32:] == vin[1, C // 2:, 0:8, :].tobytes()
    out, placed, dropped = orc.snap2_unpack(pk, 1000, T, 192, C, S * P)
    assert placed == len(pk) and dropped == 0 and np.array_equal(out.reshape(vin.shape), vin)
    # shuffled, one packet lost, one duplicated, one from the next window, one for another pipeline's channels
    rng = np.random.default_rng(3)
    lost = 5
    extra = [pk[7], orc.snap2_packets(vin[:1], seq0=1000 + T, nstand_per_pkt=8, chan0_pipeline=192)[0],
             orc.snap2_packets(vin[:1], seq0=1000, nstand_per_pkt=8, chan0_pipeline=192 + C)[0]]
    mixed = [p for i, p in enumerate(pk) if i != lost] + extra
    mixed = [mixed[i] for i in rng.permutation(len(mixed))]
    out2, placed2, dropped2 = orc.snap2_unpack(mixed, 1000, T, 192, C, S * P)
    assert placed2 == len(pk) and dropped2 == 2
    exp = vin.copy().reshape(T, C, S * P)
    seq, _, npol, _, nchan, _, _, chan0, pol0 = struct.unpack(orc.SNAP2_HDR, pk[lost][:32])
    exp[seq - 1000, chan0 - 192:chan0 - 192 + nchan, pol0:pol0 + npol] = 0
    assert np.array_equal(out2, exp)


def _load_snap2_fixture(golden_dir):
    with open(os.path.join(golden_dir, "snap2_8t_4c_64s_2p_deadbeef.bin"), "rb") as fh:
        meta = json.loads(fh.readline().decode())
        blob = fh.read()
    n, b = meta["npkt"], meta["pkt_bytes"]
    assert len(blob) == n * b
    with open(os.path.join(golden_dir, "in_8t_4c_64s_2p_deadbeef.dat"), "rb") as fh:
        hdr = json.loads(fh.readline().decode())
        vin = np.frombuffer(fh.read(), dtype=np.uint8).reshape(hdr["shape"])
    return meta, [blob[k * b:(k + 1) * b] for k in range(n)], vin


def test_snap2_oracle_is_pinned_by_the_reference_transmitter(golden_dir):
    """The fixture holds the datagrams the reference's own F-engine emulator (test_transmitters/test_tx_vectors.py,
    run under a recording socket by oracle/make_golden_snap2.py) sends for the reference-generated input file:
    oracle.snap2_packets must reproduce them byte for byte, and oracle.snap2_unpack must turn them back into that file."""
    meta, pkts, vin = _load_snap2_fixture(golden_dir)
    assert vin.shape == (meta["ntime"], meta["nchan"], meta["nstand"], meta["npol"])
    mine = orc.snap2_packets(vin, seq0=0, sync_time=0, nchan_blocks=meta["nchan_blocks"], nstand_per_pkt=meta["nstand_per_pkt"])
    assert len(mine) == len(pkts)
    for k, (a, b) in enumerate(zip(mine, pkts)):
        assert a == b, "packet %d differs from the reference transmitter's" % k
    out, placed, dropped = orc.snap2_unpack(pkts, 0, meta["ntime"], 0, meta["nchan"], meta["nstand"] * meta["npol"])
    assert placed == len(pkts) and dropped == 0
    assert np.array_equal(out.reshape(vin.shape), vin)


def test_vectorised_cpu_baseline_equals_the_scalar_oracle():
    """bench.py's cpu_baseline runs oracle/xeng_cpu_fast.c (the same contraction written for the host it runs on: -march=native, AVX-512
    VNNI where there is one, built on this machine).  Word for word the scalar oracle -- every word of the planes, the diagonal
    cells' unaddressed ones included -- for first gulps (assign) and following ones (accumulate)."""
    for (T, C, S) in [(8, 4, 16), (96, 3, 48), (40, 2, 36), (32, 2, 352)]:
        v = np.random.default_rng(T + S).integers(0, 256, (2 * T, C, S, 2), dtype=np.uint8)
        a = orc.xgpu_correlate(v[:T], S, C)
        b = orc.xgpu_correlate_fast(v[:T], S, C)
        assert np.array_equal(a, b), (T, C, S)
        assert np.array_equal(orc.xgpu_correlate(v[T:], S, C, a), orc.xgpu_correlate_fast(v[T:], S, C, b)), (T, C, S)
    assert orc.fast_lib().fast_isa() in (0, 1, 2) and orc.fast_lib().fast_num_threads() >= 1
