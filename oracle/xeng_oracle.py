"""Python face of the CPU oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.

Two layers:
  * an independent pure-numpy restatement of the reference's index function and
    golden convention (used to pin the C oracle), and
  * ctypes bindings to oracle/libxeng_oracle.so (xeng_oracle.c), which is what
    the GPU parity tests compare the HIP path against at full size.
Citations are relative to /root/reference/pipeline.

Pinning (tests/test_oracle.py): visibilities / maps / reorder against the reference's own golden generator
(oracle/make_golden.py), the beamformer against the reference's SoftwareBf functions (oracle/make_golden_beamform.py),
the SNAP2 packet stream (snap2_packets / snap2_unpack) against datagrams recorded from the reference's own transmitter
test_transmitters/test_tx_vectors.py (oracle/make_golden_snap2.py -> tests/golden/snap2_*.bin).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """gcc-compile xeng_oracle.c -> libxeng_oracle.so (no-op when up to date)."""
    so = os.path.join(HERE, "libxeng_oracle.so")
    src = os.path.join(HERE, "xeng_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libxeng_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libxeng_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, i, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.orc_num_threads.restype = i
        L.orc_xgpu_per_chan.restype = i64
        L.orc_xgpu_per_chan.argtypes = [i, i]
        L.orc_regtile_index.restype = i64
        L.orc_regtile_index.argtypes = [i, i, i]
        L.orc_xgpu_correlate.argtypes = [vp, vp, i, i, i, i]
        L.orc_xgpu_get_order.argtypes = [vp, vp, vp, i, i]
        L.orc_xgpu_reorder.argtypes = [vp, vp, vp, vp, i, i, i]
        L.orc_xgpu_subselect.argtypes = [vp, vp, vp, vp, i, i, i, i, i]
        L.orc_map_i32.argtypes = [vp, vp, ctypes.c_size_t, i]
        L.orc_map_i32.restype = None
        L.orc_beamform.argtypes = [vp, vp, vp, i, i, i, i]
        L.orc_beamform_integrate.argtypes = [vp, vp, i, i, i, i]
        L.orc_beamform_integrate_single.argtypes = [vp, vp, i, i, i, i, i]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------
# numpy restatement (independent of the C code)
# ----------------------------------------------------------------------------
def decode(u8):
    """corr_block.py:270-275 / make_golden_inputs.py:145-149."""
    u8 = np.asarray(u8, dtype=np.uint8)
    re = (u8 >> 4).astype(np.int16)
    re[re > 7] -= 16
    im = (u8 & 0xF).astype(np.int16)
    im[im > 7] -= 16
    return re, im


def per_chan(nstand, npol=2):
    """corr_block.py:231."""
    return (nstand // 2 + 1) * (nstand // 4) * npol * npol * 4


def tri_index(i, j):
    """corr_block.py:27-28."""
    return (i * (i + 1)) // 2 + j


def regtile_index(in0, in1, nstand):
    """corr_block.py:37-58 (valid for in1 >= in0)."""
    a0, a1, p0, p1 = in0 >> 1, in1 >> 1, in0 & 1, in1 & 1
    quadrant_index = tri_index(a1 // 2, a0 // 2)
    quadrant = 2 * (a0 & 1) + (a1 & 1)
    quadrant_size = (nstand // 2 + 1) * nstand // 4
    return (quadrant * quadrant_size + quadrant_index) * 4 + 2 * p1 + p0


def golden_corr(vin):
    """make_golden_inputs.py:150-158: out[c,s0,s1,p0,p1] = sum_t x[t,c,s0,p0]*conj(x[t,c,s1,p1]).
    vin uint8[T,C,S,P] -> (re, im) int64 arrays."""
    re, im = decode(vin)
    re = re.astype(np.int64)
    im = im.astype(np.int64)
    rr = np.einsum("tcap,tcbq->cabpq", re, re) + np.einsum("tcap,tcbq->cabpq", im, im)
    ii = np.einsum("tcap,tcbq->cabpq", im, re) - np.einsum("tcap,tcbq->cabpq", re, im)
    return rr, ii


def xgpu_lookup_numpy(planar, nstand, nchan):
    """xgpu_test.py:99-133 in reverse: read the planar xGPU buffer through
    regtile_index for every s1 >= s0 and return (re, im)[c,s0,s1,p0,p1] of
    conj(x[s0,p0]) * x[s1,p1] (zero for s1 < s0)."""
    pc = per_chan(nstand)
    planar = np.asarray(planar, dtype=np.int32).reshape(2, nchan, pc)
    s0, s1, p0, p1 = np.meshgrid(np.arange(nstand), np.arange(nstand), [0, 1], [0, 1], indexing="ij")
    idx = regtile_index(2 * s0 + p0, 2 * s1 + p1, nstand)
    valid = s1 >= s0
    idx = np.where(valid, idx, 0)
    re = np.where(valid, planar[0][:, idx], 0)
    im = np.where(valid, planar[1][:, idx], 0)
    return re, im


# ----------------------------------------------------------------------------
# C oracle wrappers
# ----------------------------------------------------------------------------
def xgpu_correlate(vin, nstand, nchan, acc=None):
    """vin uint8[T,C,nstand,2] (any shape with that many bytes).  Returns/updates
    the planar int32[2*matlen] accumulator (xgpu_test.py:76-83 semantics: call
    with acc=None for the first gulp, pass acc back for the following ones)."""
    vin = np.ascontiguousarray(vin, dtype=np.uint8)
    ntime = vin.size // (nchan * nstand * 2)
    assert ntime * nchan * nstand * 2 == vin.size
    accumulate = acc is not None
    if acc is None:
        acc = np.empty(2 * per_chan(nstand) * nchan, dtype=np.int32)
    rc = lib().orc_xgpu_correlate(_p(vin), _p(acc), ntime, nchan, nstand, int(accumulate))
    assert rc == 0, rc
    return acc


_FAST = None


def fast_lib():
    """oracle/xeng_cpu_fast.c built with -march=native ON THIS MACHINE: the library's file name carries a hash of the machine's CPU
    flags, so a binary that travelled from another box (the build container, a different GPU host) is never loaded."""
    global _FAST
    if _FAST is None:
        import hashlib
        flags = ""
        try:
            with open("/proc/cpuinfo") as fh:
                for line in fh:
                    if line.startswith("flags"):
                        flags = line
                        break
        except OSError:
            pass
        so = "libxeng_cpu_fast_%s.so" % hashlib.sha256(flags.encode()).hexdigest()[:12]
        path = os.path.join(HERE, so)
        src = os.path.join(HERE, "xeng_cpu_fast.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", HERE, "-B", so, "FAST_SO=" + so], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(path)
        L.fast_xgpu_correlate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.fast_xgpu_correlate.restype = ctypes.c_int
        L.fast_num_threads.restype = ctypes.c_int
        L.fast_set_threads.argtypes = [ctypes.c_int]
        L.fast_set_threads.restype = None
        L.fast_isa.restype = ctypes.c_int
        _FAST = L
    return _FAST


def xgpu_correlate_fast(vin, nstand, nchan, acc=None):
    """xgpu_correlate through the vectorised translation unit (bench.py's cpu_baseline); same arguments, same words."""
    vin = np.ascontiguousarray(vin, dtype=np.uint8)
    ntime = vin.size // (nchan * nstand * 2)
    assert ntime * nchan * nstand * 2 == vin.size
    accumulate = acc is not None
    if acc is None:
        acc = np.empty(2 * per_chan(nstand) * nchan, dtype=np.int32)
    rc = fast_lib().fast_xgpu_correlate(_p(vin), _p(acc), ntime, nchan, nstand, int(accumulate))
    assert rc == 0, rc
    return acc


def xgpu_get_order(antpol_to_input):
    a = np.ascontiguousarray(antpol_to_input, dtype=np.int32)
    nstand, npol = a.shape
    bl = np.zeros((nstand, nstand, npol, npol), dtype=np.int32)
    cj = np.zeros_like(bl)
    lib().orc_xgpu_get_order(_p(a), _p(bl), _p(cj), nstand, npol)
    return bl, cj


def xgpu_reorder(planar, bl, cj, nchan):
    nstand, _, npol, _ = bl.shape
    planar = np.ascontiguousarray(planar, dtype=np.int32)
    out = np.zeros((nstand, nstand, npol, npol, nchan, 2), dtype=np.int32)
    lib().orc_xgpu_reorder(_p(planar), _p(out), _p(np.ascontiguousarray(bl)), _p(np.ascontiguousarray(cj)),
                           nstand, npol, nchan)
    return out


def corr_packet_payloads(reordered, use_cor_fmt):
    """Payloads of the full-correlation packets in sending order, int32[nbl][4*nchan*2].
    use_cor_fmt=False: corr_output_full_block.py:461-467 (`reordered_data[s0, s1].tobytes()` for s0, s1 >= s0,
    i.e. [p0][p1][chan][2]); True: :512-519 (`reordered_data[i, i:]` transposed [0,3,1,2,4] -> [chan][p0][p1][2])."""
    nstand = reordered.shape[0]
    out = []
    for s0 in range(nstand):
        sdata = reordered[s0, s0:]                      # [s1 - s0][p0][p1][chan][2]
        if use_cor_fmt:
            sdata = sdata.transpose(0, 3, 1, 2, 4)
        out.append(np.ascontiguousarray(sdata).reshape(sdata.shape[0], -1))
    return np.concatenate(out)


def corr_packet_header_py(sync_time, spectra_id, bw_hz, sfreq, acc_len, nchan, chan0, npol, s0, s1):
    """corr_output_full_block.py:443-452, 463 (`>QQ2d4I` + `>2I`): 56 bytes, big-endian
    (docs/source/outputs.rst:33-46)."""
    import struct
    return struct.pack(">QQ2d4I", sync_time, spectra_id, bw_hz, sfreq, acc_len, nchan, chan0, npol) + struct.pack(">2I", s0, s1)


SNAP2_HDR = ">QLHHHHLLL"     # test_tx_vectors.py:103-108, test_tx_mt.c:39-49 (32 bytes)


def snap2_packets(data, seq0=0, sync_time=0, nchan_blocks=2, nstand_per_pkt=32, chan0_pipeline=0):
    """The packet stream the reference's SNAP2 emulator sends for data u8[ntime][nchan][nstand][npol]
    (test_tx_vectors.py:79-112): per sequence number, per channel block, per block of `nstand_per_pkt` stands,
    one packet = header `>QLHHHHLLL` + payload [nchan_per_pkt][nstand_per_pkt][npol].  Returns a list of bytes
    in sending order.  (The emulator numbers channels from 0; chan0_pipeline offsets them as a pipeline that
    does not start at channel 0 sees them.)"""
    import struct
    ntime, nchan, nstand, npol = data.shape
    assert nchan % nchan_blocks == 0 and nstand % nstand_per_pkt == 0
    nchan_per_pkt = nchan // nchan_blocks
    npol_blocks = nstand // nstand_per_pkt
    pkts = []
    for t in range(ntime):
        for cb in range(nchan_blocks):
            for pb in range(npol_blocks):
                hdr = struct.pack(SNAP2_HDR, seq0 + t, sync_time, npol * nstand_per_pkt, nstand * npol,
                                  nchan_per_pkt, nchan, cb, chan0_pipeline + cb * nchan_per_pkt, pb * nstand_per_pkt * npol)
                pay = data[t, cb * nchan_per_pkt:(cb + 1) * nchan_per_pkt, pb * nstand_per_pkt:(pb + 1) * nstand_per_pkt, :]
                pkts.append(hdr + np.ascontiguousarray(pay).tobytes())
    return pkts


def snap2_unpack(pkts, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot):
    """Inverse: scatter packets (any order) into u8[ntime][nchan_tot][npol_tot]; returns (gulp, nplaced, ndropped).
    Row c of a packet goes to [seq - seq0][chan0 - chan0_pipeline + c][pol0 : pol0 + npol]; packets outside the
    window / geometry are dropped; samples no packet covers stay 0."""
    import struct
    out = np.zeros((ntime, nchan_tot, npol_tot), dtype=np.uint8)
    placed = dropped = 0
    for p in pkts:
        seq, _, npol, _, nchan, _, _, chan0, pol0 = struct.unpack(SNAP2_HDR, p[:32])
        t, c0 = seq - seq0, chan0 - chan0_pipeline
        if not (0 <= t < ntime and npol > 0 and nchan > 0 and c0 >= 0 and c0 + nchan <= nchan_tot
                and pol0 + npol <= npol_tot and nchan * npol <= len(p) - 32):
            dropped += 1
            continue
        out[t, c0:c0 + nchan, pol0:pol0 + npol] = np.frombuffer(p[32:32 + nchan * npol], dtype=np.uint8).reshape(nchan, npol)
        placed += 1
    return out, placed, dropped


def xgpu_subselect(planar, vismap, conj, nchan, nchan_sum, nstand, npol=2):
    planar = np.ascontiguousarray(planar, dtype=np.int32)
    vismap = np.ascontiguousarray(vismap, dtype=np.int32)
    conj = np.ascontiguousarray(conj, dtype=np.int32)
    out = np.zeros((nchan // nchan_sum, vismap.size, 2), dtype=np.int32)
    rc = lib().orc_xgpu_subselect(_p(planar), _p(out), _p(vismap), _p(conj), vismap.size, nchan, nchan_sum, nstand, npol)
    assert rc == 0
    return out


def map_i32(a, b, add):
    assert a.dtype == np.int32 and b.dtype == np.int32 and a.size == b.size
    lib().orc_map_i32(_p(a), _p(b), a.size, int(add))
    return a


def beamform(vin, weights, ntime, nchan, ninput, nbeam):
    vin = np.ascontiguousarray(vin, dtype=np.uint8)
    w = np.ascontiguousarray(weights, dtype=np.complex64)
    assert vin.size == ntime * nchan * ninput and w.size == nchan * nbeam * ninput
    out = np.zeros((nchan, nbeam, ntime), dtype=np.complex64)
    lib().orc_beamform(_p(vin), _p(w), _p(out), ntime, nchan, ninput, nbeam)
    return out


def beamform_integrate(beams, ntime_sum):
    b = np.ascontiguousarray(beams, dtype=np.complex64)
    nchan, nbeam, ntime = b.shape
    out = np.zeros((nbeam // 2, ntime // ntime_sum, nchan, 4), dtype=np.float32)
    rc = lib().orc_beamform_integrate(_p(b), _p(out), nchan, nbeam, ntime, ntime_sum)
    assert rc == 0
    return out


def beamform_integrate_single(beams, ntime_sum, beam):
    b = np.ascontiguousarray(beams, dtype=np.complex64)
    nchan, nbeam, ntime = b.shape
    out = np.zeros((ntime // ntime_sum, nchan, 4), dtype=np.float32)
    rc = lib().orc_beamform_integrate_single(_p(b), _p(out), nchan, nbeam, ntime, ntime_sum, beam)
    assert rc == 0
    return out
