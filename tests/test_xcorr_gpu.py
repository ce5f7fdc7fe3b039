"""GPU parity tests of the X-engine (Corr) path, through the C ABI, against the CPU oracle.
Integer work: the bar is bit-exact on every word of both planes."""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import xeng_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util
    assert gpu_util.ffi.device_count() >= 1
    return gpu_util


def oracle_run(vin, nstand, nchan, ntime_gulp):
    acc = None
    v = vin.reshape(-1, ntime_gulp, nchan, nstand, 2)
    for g in range(v.shape[0]):
        acc = orc.xgpu_correlate(v[g], nstand, nchan, acc)
    return acc


def test_corner_turn_layout(gpu):
    """Stage 1 in isolation: the staging area holds, for (c, ib, kt, sub, lane=(h,r)), the 16
    samples kt*32+16h+[0,16) of input ib*64+sub*32+r (as a set: the order inside a fragment is free)."""
    nstand, nchan, ntime = 48, 3, 80      # 96 inputs -> 2 blocks (second half padded); 80 = 2.5 K tiles
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=2)
    assert x.path() == (0, 0)             # 80 samples per gulp: not whole 96-sample stages -> two-pass path
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=3)
    x.inbuf = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    gpu.ffi.call("xengXgpuKernelAsync", x.inbuf.ptr, x.out.ptr, 0)
    gpu.ffi.call("xengXgpuSync")
    L = gpu.ffi.lib()
    L.xengXgpuDebugReadStash.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    cap_kt, nblk = ctypes.c_int(), ctypes.c_int()
    assert L.xengXgpuDebugReadStash(None, 0, ctypes.byref(cap_kt), ctypes.byref(nblk)) == 0
    cap_kt, nblk = cap_kt.value, nblk.value
    assert nblk == 2
    stash = np.zeros(nchan * nblk * cap_kt * 2048, dtype=np.uint8)
    assert L.xengXgpuDebugReadStash(stash.ctypes.data, stash.nbytes, None, None) == 0
    stash = stash.reshape(nchan, nblk, cap_kt, 2, 2, 32, 16)   # c, ib, kt, sub, h, r, 16
    flat = vin.reshape(ntime, nchan, nstand * 2)
    gkt = (ntime + 31) // 32
    for c in range(nchan):
        for ib in range(nblk):
            for kt in range(gkt):
                for sub in range(2):
                    for h in range(2):
                        for r in range(32):
                            i = ib * 64 + sub * 32 + r
                            t0 = kt * 32 + 16 * h
                            exp = np.zeros(16, np.uint8)
                            if i < nstand * 2:
                                n = max(0, min(16, ntime - t0))
                                exp[:n] = flat[t0:t0 + n, c, i]
                            got = stash[c, ib, kt, sub, h, r]
                            assert np.array_equal(np.sort(got), np.sort(exp)), (c, ib, kt, sub, h, r)
                            assert np.array_equal(got, exp)   # this implementation also keeps time order
    x.close()


def _load_dat(path):
    with open(path, "rb") as fh:
        meta = json.loads(fh.readline().decode())
        raw = fh.read()
    dt = np.uint8 if "uint8" in meta["dtype"] else np.complex128
    return meta, np.frombuffer(raw, dtype=dt).reshape(meta["shape"])


@pytest.mark.parametrize("tag", ["deadbeef", "chanramp"])
def test_golden_cfg1(gpu, golden_dir, tag):
    """BASELINE config 1 shapes (16 stands x 2 pol, 4 chan) against the reference's golden file:
    HIP X-engine -> GetOrder -> Reorder (both through the C ABI) == make_golden_inputs.py output."""
    _, vin = _load_dat(os.path.join(golden_dir, "in_8t_4c_16s_2p_%s.dat" % tag))
    meta, corr = _load_dat(os.path.join(golden_dir, "corr_8t_4a_4c_16s_2p_%s.dat" % tag))
    T, C, S, P = vin.shape
    acc_len = meta["acc_len"]
    x = gpu.Xgpu(S, C, acc_len)
    a2i = np.arange(S * P, dtype=np.int32).reshape(S, P)
    bl = np.zeros((S, S, P, P), np.int32)
    cj = np.zeros_like(bl)
    gpu.ffi.call("xengXgpuGetOrder", a2i.ctypes.data, bl.ctypes.data, cj.ctypes.data)
    obl, ocj = orc.xgpu_get_order(a2i)
    assert np.array_equal(bl, obl) and np.array_equal(cj, ocj)
    for blk in range(T // acc_len):
        planar = x.run(vin[blk * acc_len:(blk + 1) * acc_len])
        assert np.array_equal(planar, orc.xgpu_correlate(vin[blk * acc_len:(blk + 1) * acc_len], S, C))
        ro = np.zeros((S, S, P, P, C, 2), np.int32)
        gpu.ffi.call("xengXgpuReorder", planar.ctypes.data, ro.ctypes.data, bl.ctypes.data, cj.ctypes.data)
        for s0 in range(S):
            for s1 in range(s0, S):
                g = corr[blk, :, s0, s1]                      # [c, p0, p1]
                assert np.array_equal(ro[s0, s1, :, :, :, 0], np.moveaxis(g.real, 0, -1).astype(np.int32))
                assert np.array_equal(ro[s0, s1, :, :, :, 1], np.moveaxis(g.imag, 0, -1).astype(np.int32))
    x.close()


def test_golden_64input(gpu, golden_dir):
    z = np.load(os.path.join(golden_dir, "golden_64t_32a_8c_32s_2p_deadbeef.npz"))
    vin = z["vin"]
    T, C, S, P = vin.shape
    x = gpu.Xgpu(S, C, 32)
    for blk in range(2):
        planar = x.run(vin[blk * 32:(blk + 1) * 32])
        re, im = orc.xgpu_lookup_numpy(planar, S, C)
        m = (np.arange(S)[:, None] <= np.arange(S)[None, :])[None, :, :, None, None]
        assert np.array_equal(re * m, z["corr_re"][blk] * m)
        assert np.array_equal(im * m, -z["corr_im"][blk] * m)     # stored = conj(golden), xgpu_test.py:111
    x.close()


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp,kind", [
    (16, 4, 8, 1, "full"),          # one half-empty 64-input block, K tile mostly padding
    (32, 8, 32, 3, "full"),         # exactly one block
    (36, 2, 64, 2, "full"),         # 72 inputs: not a multiple of 16 -> register-only corner turn
    (48, 5, 96, 2, "random"),       # 2 blocks (one diagonal work-group), nchan not a multiple of 8
    (80, 8, 480, 2, "full"),        # 3 blocks: odd count -> leftover row packing; the reference gulp length
    (96, 3, 160, 2, "88"),          # every nibble -8: largest magnitudes, exercises the x16 scaling / P-Q split
    (176, 2, 64, 1, "full"),        # 6 blocks: off-diagonal squares
    (208, 1, 100, 3, "full"),       # 7 blocks (odd), ntime not a multiple of 32 (padded K tile per gulp)
    (8, 2, 96, 1, "full"),          # whole 96-sample stages but rows of 32 bytes: below the fused kernel's 128-byte minimum -> two-pass path
])
def test_parity_vs_oracle(gpu, nstand, nchan, ntime, ngulp, kind):
    x = gpu.Xgpu(nstand, nchan, ntime)
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, kind, seed=nstand + ntime)
    got = x.run(vin)
    exp = oracle_run(vin, nstand, nchan, ntime)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)
    # async enqueue + one sync gives the same answer
    assert np.array_equal(x.run(vin, use_async=True), exp)
    x.close()


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp,kind", [
    (32, 8, 96, 3, "full"),          # one full 64-input block
    (40, 3, 96, 2, "full"),          # 80 inputs: last block has one live 16-byte chunk (clamped source columns)
    (64, 5, 192, 2, "random"),       # 2 blocks, 2 stages per gulp, nchan not a multiple of 8
    (96, 3, 96, 4, "88"),            # 3 blocks, every nibble -8
    (160, 2, 288, 2, "full"),        # 5 blocks (odd)
    (224, 1, 96, 1, "full"),         # 7 blocks, a single stage: the pipeline prologue covers all of K
    (80, 8, 480, 2, "full"),         # the reference gulp length; 2.5 blocks
    (512, 8, 96, 1, "random"),       # 1024 inputs = 16 blocks: 136 wave tiles in 34 tile groups, more work-groups than CUs
    (72, 4, 96, 2, "full"),          # 144 inputs: 3 blocks whose last 32-input fragment is all padding (Z wave on padded fragments)
    (8, 8, 96, 1, "full"),           # 16 inputs x 8 channels: one block, one live 16-byte chunk, rows of exactly 128 bytes
    (200, 3, 96, 1, "random"),       # 400 inputs = 6.25 blocks -> 7 blocks, the last one a quarter full
])
def test_parity_fused_corner_turn(gpu, nstand, nchan, ntime, ngulp, kind):
    """Default path when gulps are whole 96-sample stages: the contraction kernel reads the time-major
    gulps in place (async) or a raw copy (sync) and transposes in LDS.  Same words as the oracle, and as
    the two-pass path (XENG_RAW=0) on the same input."""
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, kind, seed=nstand + ntime)
    exp = oracle_run(vin, nstand, nchan, ntime)
    x = gpu.Xgpu(nstand, nchan, ntime)
    assert x.path() == (1, 0)
    assert np.array_equal(x.run(vin), exp)                     # synchronous drop-in calls
    assert np.array_equal(x.run(vin, use_async=True), exp)     # in place
    x.close()
    os.environ["XENG_RAW"] = "0"
    try:
        x = gpu.Xgpu(nstand, nchan, ntime)
        assert x.path() == (0, 0)
        assert np.array_equal(x.run(vin, use_async=True), exp)
        x.close()
    finally:
        del os.environ["XENG_RAW"]


def test_fused_mixed_sync_async_and_early_flush(gpu):
    """One integration fed by both call flavours, longer than the staging depth: the first flush (3 gulps)
    and the dump (2 gulps) accumulate into the same span; sync gulps are copied, async ones read in place."""
    nstand, nchan, ntime, ngulp = 64, 8, 96, 5
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=3)
    assert x.path() == (1, 0)
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "full", seed=11)
    x.inbuf = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    scratch = gpu.ffi.DeviceBuffer(x.gulp_bytes)
    for g in range(ngulp):
        src = x.inbuf.ptr + g * x.gulp_bytes
        if g % 2 == 0:
            gpu.ffi.call("xengXgpuKernelAsync", src, x.out.ptr, int(g == ngulp - 1))
        else:
            # sync: the caller's buffer is recycled right after the call returns
            gpu.ffi.call("xengMemcpy", scratch.ptr, src, x.gulp_bytes)
            gpu.ffi.call("xengXgpuKernel", scratch.ptr, x.out.ptr, 0)
            gpu.ffi.call("xengMemset", scratch.ptr, 0xEE, x.gulp_bytes)
    gpu.ffi.call("xengXgpuSync")
    assert np.array_equal(x.out.download(np.int32), oracle_run(vin, nstand, nchan, ntime))
    with pytest.raises(gpu.ffi.XengError):                      # in-place reads are 16-byte loads
        gpu.ffi.call("xengXgpuKernelAsync", x.inbuf.ptr + 4, x.out.ptr, 0)
    x.close()
    scratch.free()


def test_mid_integration_flush_accumulates(gpu):
    """Staging depth 2 with 5 gulps: the library flushes early (overwrite), later flushes
    read-modify-write the caller's buffer; the dumped result is still the 5-gulp sum, and the
    next integration starts from zero again (xgpu_test.py:76-83 semantics)."""
    nstand, nchan, ntime = 48, 8, 64
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=2)
    v1 = gpu.synth_voltages(ntime * 5, nchan, nstand, "full", seed=11)
    v2 = gpu.synth_voltages(ntime * 3, nchan, nstand, "full", seed=12)
    assert np.array_equal(x.run(v1), oracle_run(v1, nstand, nchan, ntime))
    assert np.array_equal(x.run(v2, poison=False), oracle_run(v2, nstand, nchan, ntime))
    x.close()


def test_output_buffer_change_inside_integration_is_an_error(gpu):
    nstand, nchan, ntime = 16, 8, 32
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=1)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "random")
    x.inbuf = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    other = gpu.ffi.DeviceBuffer(x.out.nbytes)
    gpu.ffi.call("xengXgpuKernel", x.inbuf.ptr, x.out.ptr, 0)       # flushed into x.out (depth 1)
    with pytest.raises(gpu.ffi.XengError):
        gpu.ffi.call("xengXgpuKernel", x.inbuf.ptr, other.ptr, 1)
    x.close()
    other.free()


def test_host_buffer_correlate(gpu):
    """bfXgpuCorrelate shape (xgpu_test.py:86-89): host in, host out on dump."""
    nstand, nchan, ntime = 32, 8, 64
    gpu.ffi.call("xengXgpuConfigure", nstand, 2, nchan, ntime, 0)
    gpu.ffi.call("xengXgpuInitialize", 0)
    vin = gpu.synth_voltages(ntime * 2, nchan, nstand, "full", seed=4)
    out = np.zeros(2 * orc.per_chan(nstand) * nchan, np.int32)
    flat = vin.reshape(2, -1)
    gpu.ffi.call("xengXgpuCorrelate", flat[0].ctypes.data, out.ctypes.data, 0)
    gpu.ffi.call("xengXgpuCorrelate", flat[1].ctypes.data, out.ctypes.data, 1)
    assert np.array_equal(out, oracle_run(vin, nstand, nchan, ntime))
    gpu.ffi.call("xengXgpuDestroy")


def test_subselect(gpu):
    nstand, nchan, ntime = 32, 8, 64
    x = gpu.Xgpu(nstand, nchan, ntime)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=8)
    planar = x.run(vin)
    bl, cj = orc.xgpu_get_order(np.arange(nstand * 2, dtype=np.int32).reshape(nstand, 2))
    rng = np.random.default_rng(2)
    nvis = 300
    s0, s1 = rng.integers(0, nstand, nvis), rng.integers(0, nstand, nvis)
    p0, p1 = rng.integers(0, 2, nvis), rng.integers(0, 2, nvis)
    vismap = bl[s0, s1, p0, p1].astype(np.int32)
    conj = cj[s0, s1, p0, p1].astype(np.int32)
    dv = gpu.ffi.DeviceBuffer(vismap.nbytes).upload(vismap)
    dc = gpu.ffi.DeviceBuffer(conj.nbytes).upload(conj)
    do = gpu.ffi.DeviceBuffer((nchan // 4) * nvis * 8)
    gpu.ffi.call("xengXgpuSubSelect", x.out.ptr, do.ptr, dv.ptr, dc.ptr, nvis, 4)
    got = do.download(np.int32).reshape(nchan // 4, nvis, 2)
    assert np.array_equal(got, orc.xgpu_subselect(planar, vismap, conj, nchan, 4, nstand))
    x.close()


def test_consumers_do_not_wait_for_the_contraction(gpu):
    """The reference runs Corr, CorrSubsel and CorrOutputFull on their own threads (lwa352-pipeline.py:232-262, 296-302).
    While the Corr thread sits in a synchronous dump (xengXgpuKernel(..., doDump=1)) behind a queue of contractions,
    a second thread's xengXgpuSubSelect on the PREVIOUS, completed span must return without waiting for that dump: the
    library holds its context lock only while enqueueing, and SubSelect is ordered behind the producer of its own input
    span alone.  Also: SubSelect on a span whose contraction is still queued waits for exactly that contraction."""
    import threading
    import time
    nstand, nchan, ntime, ngulp = 352, 96, 480, 5
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=ngulp)
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "random", seed=5)
    prev = gpu.ffi.DeviceBuffer(x.out.nbytes)
    x.out, keep = prev, x.out
    planar = x.run(vin)                               # completed span in `prev`
    x.out = keep
    bl, cj = orc.xgpu_get_order(np.arange(nstand * 2, dtype=np.int32).reshape(nstand, 2))
    nvis = 4704
    rng = np.random.default_rng(4)
    s0, s1 = rng.integers(0, nstand, nvis), rng.integers(0, nstand, nvis)
    vismap = bl[s0, s1, 0, 0].astype(np.int32)
    conj = cj[s0, s1, 0, 0].astype(np.int32)
    dv = gpu.ffi.DeviceBuffer(vismap.nbytes).upload(vismap)
    dc = gpu.ffi.DeviceBuffer(conj.nbytes).upload(conj)
    do = gpu.ffi.DeviceBuffer((nchan // 4) * nvis * 8)
    exp = orc.xgpu_subselect(planar, vismap, conj, nchan, 4, nstand)
    gpu.ffi.call("xengXgpuSubSelect", prev.ptr, do.ptr, dv.ptr, dc.ptr, nvis, 4)     # warm (first launch, event creation)
    L = gpu.ffi.lib()
    gb = x.gulp_bytes
    nint = 150                                        # ~35 ms of queued contractions into x.out
    t = {}
    entered = threading.Event()

    def corr_thread():
        for n in range(nint):
            for g in range(ngulp):
                last = n == nint - 1 and g == ngulp - 1
                if last:
                    entered.set()
                    t["dump_call"] = time.perf_counter()
                    rc = L.xengXgpuKernel(x.inbuf.ptr + g * gb, x.out.ptr, 1)        # synchronous dump: waits for the whole queue
                else:
                    rc = L.xengXgpuKernelAsync(x.inbuf.ptr + g * gb, x.out.ptr, int(g == ngulp - 1))
                assert rc == 0
        t["dump_done"] = time.perf_counter()

    def subsel_thread():
        entered.wait()
        time.sleep(0.002)                             # the Corr thread is inside its dump call now
        t["sub_call"] = time.perf_counter()
        assert L.xengXgpuSubSelect(prev.ptr, do.ptr, dv.ptr, dc.ptr, nvis, 4) == 0
        t["sub_done"] = time.perf_counter()

    a, b = threading.Thread(target=corr_thread), threading.Thread(target=subsel_thread)
    a.start(); b.start(); a.join(); b.join()
    assert np.array_equal(do.download(np.int32).reshape(nchan // 4, nvis, 2), exp)
    dump_wait = t["dump_done"] - t["dump_call"]
    assert dump_wait > 0.010, dump_wait               # the dump really was waiting behind the queue
    assert t["sub_done"] < t["dump_done"] - 0.005, (t["sub_done"] - t["sub_call"], dump_wait)
    # ordering with its own producer: enqueue one integration into `prev` asynchronously and sub-select it at once
    vin2 = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "random", seed=6)
    x.inbuf.upload(vin2.reshape(-1))
    for g in range(ngulp):
        gpu.ffi.call("xengXgpuKernelAsync", x.inbuf.ptr + g * gb, prev.ptr, int(g == ngulp - 1))
    gpu.ffi.call("xengXgpuSubSelect", prev.ptr, do.ptr, dv.ptr, dc.ptr, nvis, 4)
    gpu.ffi.call("xengXgpuSync")
    exp2 = orc.xgpu_subselect(oracle_run(vin2, nstand, nchan, ntime), vismap, conj, nchan, 4, nstand)
    assert np.array_equal(do.download(np.int32).reshape(nchan // 4, nvis, 2), exp2)
    prev.free()
    x.close()


def test_config2_full_size(gpu):
    """BASELINE config 2: 704 inputs, 96 channels, 5 gulps of 480 (acc_len 2400), bit-exact vs the
    C oracle on every word of both planes -- random bytes (the golden generator), all-0x88, and the generator's
    --chanramp set."""
    nstand, nchan, ntime, ngulp = 352, 96, 480, 5
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "random")
    exp = oracle_run(vin, nstand, nchan, ntime)
    os.environ["XENG_RAW"] = "0"                # the two-pass path (corner turn + fragment-major staging)
    try:
        x = gpu.Xgpu(nstand, nchan, ntime)
        assert x.path() == (0, 0)
        assert np.array_equal(x.run(vin), exp)
        x.close()
    finally:
        del os.environ["XENG_RAW"]
    x = gpu.Xgpu(nstand, nchan, ntime)          # default: gulps read in place, corner turn fused
    assert x.path() == (1, 0)
    assert np.array_equal(x.run(vin, use_async=True), exp)
    got = x.run(vin)
    assert np.array_equal(got, exp)
    # size-independent properties at full size: autos are real and positive, planes have the xGPU length
    matlen = 96 * 249216                       # per_chan = 249216 words per plane per channel
    assert got.size == 2 * matlen
    auto = orc.regtile_index(2 * 7, 2 * 7, nstand)
    assert got[auto] > 0 and got[matlen + auto] == 0
    v88 = gpu.synth_voltages(ntime, nchan, nstand, "88")
    g88 = x.run(v88)
    # every product is (-8-8j)*conj(-8-8j) = 128: all words 128*ntime (re) / 0 (im), incl. the unaddressed ones
    assert np.all(g88[:matlen] == 128 * ntime) and np.all(g88[matlen:] == 0)
    # the reference generator's third input set (make_golden_inputs.py:112-116, --chanramp: every sample of channel c
    # is the byte c) at full size, one whole integration, against the oracle on every word; and in closed form: byte c
    # = (re, im) nibbles, every visibility of channel c is acc_len * (re^2 + im^2) + 0j
    vramp = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "chanramp")
    gramp = x.run(vramp, use_async=True)
    assert np.array_equal(gramp, oracle_run(vramp, nstand, nchan, ntime))
    cb = np.arange(nchan)
    re, im = (cb >> 4) - 16 * ((cb >> 4) > 7), (cb & 15) - 16 * ((cb & 15) > 7)
    planes = gramp.reshape(2, nchan, -1)
    assert np.all(planes[0] == (ntime * ngulp * (re * re + im * im))[:, None]) and np.all(planes[1] == 0)
    x.close()


@pytest.mark.parametrize("ntime,fused", [(64, 0), (96, 1)])
def test_streaming_lagged_sync_two_outputs(gpu, ntime, fused):
    """The streaming call pattern of bench.py: integration n+1 is enqueued (into the other output
    span) before the caller waits for dump n with xengXgpuSyncLag(1).  Every dumped span must equal
    the oracle, i.e. the double-buffered staging areas / three streams keep their ordering."""
    nstand, nchan, ngulp, nint = 80, 8, 3, 7
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=ngulp)
    assert x.path() == (fused, 0)
    outs = [x.out, gpu.ffi.DeviceBuffer(x.out.nbytes)]
    vin = gpu.synth_voltages(ntime * ngulp * nint, nchan, nstand, "full", seed=77).reshape(nint, ngulp, -1)
    din = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    results = {}
    for n in range(nint):
        for g in range(ngulp):
            gpu.ffi.call("xengXgpuKernelAsync", din.ptr + (n * ngulp + g) * x.gulp_bytes, outs[n & 1].ptr, int(g == ngulp - 1))
        gpu.ffi.call("xengXgpuSyncLag", 1)
        if n >= 1:
            results[n - 1] = outs[(n - 1) & 1].download(np.int32)     # complete although dump n is still running
    gpu.ffi.call("xengXgpuSyncLag", 0)
    results[nint - 1] = outs[(nint - 1) & 1].download(np.int32)
    for n in range(nint):
        assert np.array_equal(results[n], oracle_run(vin[n], nstand, nchan, ntime)), n
    with pytest.raises(gpu.ffi.XengError):
        gpu.ffi.call("xengXgpuSyncLag", 4)
    # the non-blocking form: everything has been waited for, so every lag reads "done"; bad arguments are errors
    import ctypes
    done = ctypes.c_int(-1)
    for lag in range(4):
        gpu.ffi.call("xengXgpuDumpDone", lag, ctypes.byref(done))
        assert done.value == 1
    for g in range(ngulp):           # one more integration in flight: the query answers at once, SyncLag(0) then makes it 1
        gpu.ffi.call("xengXgpuKernelAsync", din.ptr + g * x.gulp_bytes, outs[0].ptr, int(g == ngulp - 1))
    gpu.ffi.call("xengXgpuDumpDone", 0, ctypes.byref(done))
    assert done.value in (0, 1)
    gpu.ffi.call("xengXgpuDumpDone", 1, ctypes.byref(done))
    assert done.value == 1           # the dump before it completed long ago
    gpu.ffi.call("xengXgpuSyncLag", 0)
    gpu.ffi.call("xengXgpuDumpDone", 0, ctypes.byref(done))
    assert done.value == 1
    assert np.array_equal(outs[0].download(np.int32), oracle_run(vin[0], nstand, nchan, ntime))
    with pytest.raises(gpu.ffi.XengError):
        gpu.ffi.call("xengXgpuDumpDone", 4, ctypes.byref(done))
    with pytest.raises(gpu.ffi.XengError):
        gpu.ffi.call("xengXgpuDumpDone", 0, None)
    x.close()
    outs[1].free()


def test_reset_drops_partial_integration(gpu):
    nstand, nchan, ntime = 32, 8, 32
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=1)        # depth 1: the first gulp is flushed into `out`
    v = gpu.synth_voltages(ntime * 3, nchan, nstand, "full", seed=5)
    x.inbuf = gpu.ffi.DeviceBuffer(v.size).upload(v)
    gpu.ffi.call("xengXgpuKernel", x.inbuf.ptr, x.out.ptr, 0)
    gpu.ffi.call("xengXgpuReset")                           # abandon it
    gpu.ffi.call("xengXgpuKernel", x.inbuf.ptr + x.gulp_bytes, x.out.ptr, 0)
    gpu.ffi.call("xengXgpuKernel", x.inbuf.ptr + 2 * x.gulp_bytes, x.out.ptr, 1)
    assert np.array_equal(x.out.download(np.int32), oracle_run(v[ntime:], nstand, nchan, ntime))
    x.close()


@pytest.mark.parametrize("nstand,nchan,permute", [(16, 4, False), (48, 7, True), (36, 96, False)])
def test_packetize_matches_reorder_then_slice(gpu, nstand, nchan, permute):
    """xengXgpuPacketize (device) == bfXgpuReorder followed by the per-baseline slicing of
    CorrOutputFull.send_packets_py / send_packets_bf, for the identity and a scrambled antpol_to_input map."""
    ntime = 32
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=nstand)
    planar = orc.xgpu_correlate(vin, nstand, nchan)
    a2i = np.arange(nstand * 2, dtype=np.int32).reshape(nstand, 2)
    if permute:
        a2i = np.random.RandomState(4).permutation(nstand * 2).astype(np.int32).reshape(nstand, 2)
    bl, cj = orc.xgpu_get_order(a2i)
    reordered = orc.xgpu_reorder(planar, bl, cj, nchan)
    x = gpu.Xgpu(nstand, nchan, ntime)
    din = gpu.ffi.DeviceBuffer(planar.nbytes).upload(planar)
    dbl = gpu.ffi.DeviceBuffer(bl.nbytes).upload(np.ascontiguousarray(bl))
    dcj = gpu.ffi.DeviceBuffer(cj.nbytes).upload(np.ascontiguousarray(cj))
    nbl = nstand * (nstand + 1) // 2
    dout = gpu.ffi.DeviceBuffer(nbl * 4 * nchan * 8)
    for fmt in (0, 1):
        gpu.ffi.call("xengMemset", dout.ptr, 0x5A, dout.nbytes)
        gpu.ffi.call("xengXgpuPacketize", din.ptr, dout.ptr, dbl.ptr, dcj.ptr, fmt)
        got = dout.download(np.int32).reshape(nbl, -1)
        assert np.array_equal(got, orc.corr_packet_payloads(reordered, bool(fmt))), fmt
    with pytest.raises(gpu.ffi.XengError):
        gpu.ffi.call("xengXgpuPacketize", din.ptr, dout.ptr, dbl.ptr, dcj.ptr, 2)
    x.close()
    for b in (din, dbl, dcj, dout):
        b.free()


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp", [
    (352, 16, 96, 2),      # 11 blocks: 16 tile groups, 11 Z waves, one wave with a single live cell
    (288, 8, 96, 1),       # 9 blocks: even number of block pairs, Z(L) takes a cell of a whole bottom tile
    (416, 8, 96, 1),       # 13 blocks
    (96, 8, 192, 2),       # 3 blocks: no squares at all
    (344, 8, 96, 1),       # 688 inputs: 11 blocks, the last one three quarters full (padded fragments, clamped columns)
])
def test_fragment_tiling_matches_tile_tiling(gpu, nstand, nchan, ntime, ngulp):
    """The fragment-level tiling (2x2 waves + Z waves: a diagonal 64x64 tile plus a free cell of an off-diagonal tile)
    stores the same words as the oracle and as the 64x64-tile tiling it replaces (XENG_TILING=64, the A/B switch)."""
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "full", seed=77 + nstand)
    exp = oracle_run(vin, nstand, nchan, ntime)
    x = gpu.Xgpu(nstand, nchan, ntime)
    assert x.path() == (1, 0)
    assert np.array_equal(x.run(vin, use_async=True), exp)
    assert np.array_equal(x.run(vin), exp)
    x.close()
    os.environ["XENG_TILING"] = "64"
    try:
        x = gpu.Xgpu(nstand, nchan, ntime)
        assert np.array_equal(x.run(vin, use_async=True), exp)
        x.close()
    finally:
        del os.environ["XENG_TILING"]


def _kernel_in_use(gpu):
    import ctypes
    w, k = ctypes.c_int(), ctypes.c_int()
    gpu.ffi.call("xengXgpuGetKernel", ctypes.byref(w), ctypes.byref(k))
    return w.value, k.value


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp", [
    (352, 8, 96, 5),       # 704 inputs: 16 tile groups, all three operand patterns (2x2 waves, both halves of a Z wave); 5 stages: two pairs + an odd last stage
    (352, 8, 192, 2),      # an even number of stages: no half K-tile at the end
    (352, 8, 96, 1),       # a single stage: one K-tile and a half
    (96, 8, 96, 3),        # 3 blocks: no squares at all
    (344, 8, 96, 2),       # 688 inputs: padded fragments, clamped columns
    (80, 8, 480, 2),       # 2.5 blocks, config-2 gulp length (5 stages per gulp)
])
def test_eight_wave_16x16x64_kernel_matches_the_four_wave_kernel_and_the_oracle(gpu, nstand, nchan, ntime, ngulp):
    """Round 5: with XENG_KLOOP=16 plain and slab launches take xcorr_fused16_kernel (8 waves per work-group, v_mfma_i32_16x16x64_i8,
    K-tiles of 64 samples that straddle the 96-sample stages, the half K-tile of an odd last stage zeroed in registers,
    accumulators brought into the 32x32 layout by lane swaps for the shared epilogue); the default is the four-wave 32x32x32
    kernel.  Same words either way, and the oracle's -- through the enqueue-only calls and the synchronous ones (raw copies +
    accumulate-into-stored flushes)."""
    vin = gpu.synth_voltages(ntime * ngulp, nchan, nstand, "full", seed=5 + nstand + ntime)
    exp = oracle_run(vin, nstand, nchan, ntime)
    x = gpu.Xgpu(nstand, nchan, ntime)
    if os.environ.get("XENG_RAW") != "0" and os.environ.get("XENG_KLOOP") == "16":
        assert x.path() == (1, 0) and _kernel_in_use(gpu) == (8, 64)
    else:
        assert _kernel_in_use(gpu) == (4, 32)
    assert np.array_equal(x.run(vin, use_async=True), exp)
    assert np.array_equal(x.run(vin), exp)
    x.close()
    was = os.environ.get("XENG_KLOOP")
    os.environ["XENG_KLOOP"] = "16"
    try:
        x = gpu.Xgpu(nstand, nchan, ntime)
        assert _kernel_in_use(gpu) == ((8, 64) if os.environ.get("XENG_RAW") != "0" else (4, 32))
        assert np.array_equal(x.run(vin, use_async=True), exp)
        assert np.array_equal(x.run(vin), exp)
        x.close()
    finally:
        if was is None:
            del os.environ["XENG_KLOOP"]
        else:
            os.environ["XENG_KLOOP"] = was


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp,max_gulps,two_accs", [
    (80, 8, 96, 3, 3, True),      # 2.5 blocks: interior, diagonal and padded tiles; two accumulators (dumps may overlap)
    (80, 8, 96, 3, 3, False),     # one accumulator: the dumps are ordered by the library
    (48, 5, 96, 5, 2, True),      # staging depth 2 with 5 gulps: the dump is a read-modify-write flush
    (352, 8, 96, 1, 1, True),     # 704 inputs: the config-2 tiling (16 tile groups per channel, Z waves)
])
def test_long_accumulation_fused_into_the_dump(gpu, nstand, nchan, ntime, ngulp, max_gulps, two_accs):
    """xengXgpuKernelAsyncAcc = xengXgpuKernelAsync + CorrAcc's "a = b" / "a += b" (corr_acc_block.py:298-306) on the
    dumped span, done in the contraction's epilogue: every dump still equals the oracle, and the accumulator(s) hold
    exactly the int32 sum of the dumps."""
    nint = 6
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=max_gulps)
    assert x.path() == (1, 0)
    outs = [x.out, gpu.ffi.DeviceBuffer(x.out.nbytes)]
    accs = [gpu.ffi.DeviceBuffer(x.out.nbytes) for _ in range(2 if two_accs else 1)]
    for a in accs:
        gpu.ffi.call("xengMemset", a.ptr, 0x77, a.nbytes)       # "assign" must not depend on what was there
    vin = gpu.synth_voltages(ntime * ngulp * nint, nchan, nstand, "full", seed=5 + nstand).reshape(nint, ngulp, -1)
    din = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    results = {}
    for n in range(nint):
        acc = accs[n % len(accs)]
        first_use = n < len(accs)
        for g in range(ngulp):
            gpu.ffi.call("xengXgpuKernelAsyncAcc", din.ptr + (n * ngulp + g) * x.gulp_bytes, outs[n & 1].ptr, int(g == ngulp - 1),
                         acc.ptr, 1 if first_use else 2)
        gpu.ffi.call("xengXgpuSyncLag", 1)
        if n >= 1:
            results[n - 1] = outs[(n - 1) & 1].download(np.int32)
    gpu.ffi.call("xengXgpuSync")
    results[nint - 1] = outs[(nint - 1) & 1].download(np.int32)
    total = np.zeros_like(results[0], dtype=np.int64)
    for n in range(nint):
        exp = oracle_run(vin[n], nstand, nchan, ntime)
        assert np.array_equal(results[n], exp), n
        total += exp
    got = sum(a.download(np.int32).astype(np.int64) for a in accs)
    assert np.array_equal(got, total)
    # the separate CorrAcc map gives the same accumulator
    ref = gpu.ffi.DeviceBuffer(x.out.nbytes)
    for n in range(nint):
        outs[0].upload(results[n])
        gpu.ffi.call("xengMapAssignI32" if n == 0 else "xengMapAddI32", ref.ptr, outs[0].ptr, results[n].size)
    gpu.ffi.call("xengMapSync")
    assert np.array_equal(ref.download(np.int32).astype(np.int64), total)
    with pytest.raises(gpu.ffi.XengError):                       # accumulator == out
        gpu.ffi.call("xengXgpuKernelAsyncAcc", din.ptr, outs[0].ptr, 1, outs[0].ptr, 1)
    gpu.ffi.call("xengXgpuReset")
    x.close()
    for b in [outs[1], ref, din] + accs:
        b.free()


def test_deep_asynchronous_queue(gpu):
    """A caller that never waits: 700 dumps enqueued back to back (more than the library's ring of 256 launch events, so the
    enqueuer meets the back-pressure of the event ring -- outside the context lock -- many times), rotating over three output
    spans and two long accumulators.  Every span and both accumulators hold the right words at the end."""
    nstand, nchan, ntime, nint = 32, 8, 96, 700
    x = gpu.Xgpu(nstand, nchan, ntime)
    assert x.path() == (1, 0)
    vin = gpu.synth_voltages(ntime * 3, nchan, nstand, "full", seed=99).reshape(3, -1)
    din = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    outs = [x.out] + [gpu.ffi.DeviceBuffer(x.out.nbytes) for _ in range(2)]
    accs = [gpu.ffi.DeviceBuffer(x.out.nbytes) for _ in range(2)]
    L = gpu.ffi.lib()
    for n in range(nint):
        rc = L.xengXgpuKernelAsyncAcc(din.ptr + (n % 3) * x.gulp_bytes, outs[n % 3].ptr, 1, accs[n & 1].ptr, 1 if n < 2 else 2)
        assert rc == 0, gpu.ffi.lib().xengGetLastError()
    gpu.ffi.call("xengXgpuSync")
    exp = [oracle_run(vin[k].reshape(ntime, nchan, nstand, 2), nstand, nchan, ntime) for k in range(3)]
    for k in range(3):
        assert np.array_equal(outs[k].download(np.int32), exp[k]), k
    # accumulator a got the dumps n = a, a+2, ...: n % 3 cycles through the three inputs
    for a in range(2):
        tot = np.zeros_like(exp[0], dtype=np.int64)
        for n in range(a, nint, 2):
            tot += exp[n % 3]
        assert np.array_equal(accs[a].download(np.int32), tot.astype(np.int32)), a       # (int32 wrap-around is the kernel's arithmetic too)
    x.close()
    for b in outs[1:] + accs + [din]:
        b.free()
