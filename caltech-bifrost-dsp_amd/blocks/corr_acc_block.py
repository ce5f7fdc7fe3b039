"""CorrAcc: long accumulation of xGPU-order visibilities on the GPU.

Drop-in counterpart of pipeline/lwa352_pipeline/blocks/corr_acc_block.py (class CorrAcc,
constructor :170-171, main :193-336): accumulates upstream short integrations
(`BFMap("a = b")` on the first, `"a += b"` afterwards, :304-306) into an internal int32 buffer
and publishes it on the last (:310-319, `copy_array` + `stream_synchronize`; the output ring is
`cuda_host` in the pipeline, lwa352-pipeline.py:154).  The two map expressions are libxeng's
vectorised int32 kernels (csrc/corracc.hip).

Input header needs `seq0`, `acc_len` (:214-215); output adds `upstream_acc_len` (:216) and
rewrites `acc_len` / `seq0`.  `start_time == -1` starts on the current block (:244-245);
recovery after a new upstream sequence skips 2 integrations (:227).

Grouped mode (round 5; the default on in-repo rings; no reference counterpart).  The reference adds every dump into the
accumulator as it arrives -- per dump 191 MB read + 191 MB read-modify-written.  That order saves memory, it is not part of the
result (int32 addition wraps: any grouping of the sum gives the same words), and an MI355X has 288 GB: this block keeps the
spans of up to `group_dumps` dumps referenced (the in-repo ring keeps a span's memory alive while it is referenced, and reissues
it only behind its stamp) and sums them in ONE pass (`xengMapSumI32`): 191 + 382 / K MB per dump instead of 574.  Config 5 through
the C ABI, one box, interleaved: 0.316 ms per integration grouped (K = 10) against 0.359 with the per-dump map and 0.375 with the
fused epilogue below (profiles/r05/corracc_modes_k10.txt).  The gate is stepped per span exactly as before; long integrations
alternate between two accumulators so that one is published (helper thread, 8 MB pieces) while the next accumulates.

Fused mode (round 3, now opt-in: `XENG_CORRACC=fused`; no reference counterpart).  When the input ring is the in-repo one and the upstream `Corr` streams
(blocks/corr_block.py), the "a = b" / "a += b" of every dump is done by the contraction kernel's own epilogue
(`xengXgpuKernelAsyncAcc`, csrc/xcorr_kernels.h `LACC`): one pass over the accumulator per dump instead of a 574 MB map
kernel.  The gate below is the same state machine either way; what changes is WHO steps it and WHEN:

  * classic: this block's thread steps it for every span it reads, then runs the map;
  * fused:   `Corr` steps it (`plan_sequence` / `plan_dump`) right before it enqueues a dump -- the decision (skip, assign,
             add, first, last, new output sequence) depends on headers, sample counts and commands only, never on data --
             and this block's thread follows the queue of decisions when the spans arrive: it opens output sequences and,
             on the last dump of a long integration, adds the two partial accumulators and publishes.
    Dumps alternate between two accumulators (dumps that name the same accumulator are serialised by the library, two
    keep consecutive dumps independent), and long integrations alternate between two such pairs, so integration j+1
    accumulates while j is being published: 4 x 191 MB at config-2 size.
"""
import collections
import json
import os
import threading
import time

from ..backend import default_backend
from ..ndarray import XArray, copy_array
from ..proclog import cpu_affinity
from ..ring import WriteSpan
from .block_base import Block, declare_streams
from .integration import IntegrationGate


class _Decision:
    """What to do with one upstream span."""
    __slots__ = ("kind", "now", "begin_hdr", "first", "last", "state", "acc_set", "acc_half", "mode", "nhalves")

    def __init__(self, kind, now):
        self.kind, self.now = kind, now           # kind: 'stop' | 'wait' | 'acc'
        self.begin_hdr = None                     # header (dict) of an output sequence to open before this span
        self.first = self.last = False
        self.state = None
        self.acc_set = self.acc_half = 0          # fused: which accumulator the dump feeds ...
        self.mode = 0                             # ... 1 assign, 2 add
        self.nhalves = 1                          # fused, on `last`: accumulators of the pair that hold data


class CorrAcc(Block):
    PUBLISH_PIECE_BYTES = 8 << 20        # a long integration goes to the pinned output span in pieces of this size (see main)

    def __init__(self, log, iring, oring,
                 guarantee=True, core=-1, nchan=192, npol=2, nstand=352, acc_len=24000, gpu=-1, etcd_client=None,
                 autostartat=0, backend=None, accumulate=None):
        super(CorrAcc, self).__init__(log, iring, oring, guarantee, core, etcd_client=etcd_client)
        self._bf = backend if backend is not None else default_backend()
        self.nchan, self.npol, self.nstand = nchan, npol, nstand
        declare_streams(iring, 'map')           # (the map kernels read the input spans; the publish is a copy into the output span)
        declare_streams(oring, 'copy')
        self.matlen = nchan * (nstand // 2 + 1) * (nstand // 4) * npol * npol * 4
        self.gpu = gpu
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.igulp_size = self.matlen * 8     # complex64
        self.ogulp_size = self.igulp_size
        # integration buffer lives where the input ring lives (device memory in the pipeline)
        self.accdata = XArray(shape=(self.igulp_size // 4,), dtype='i32', space=self._bf.space_in)
        self.define_command_key('start_time', type=int, initial_val=autostartat)
        self.define_command_key('acc_len', type=int, initial_val=acc_len)
        # gate + header state, stepped by _begin_upstream_sequence / _decide (either thread, never both for one sequence)
        self._gate = IntegrationGate(recovery_skip=2, round_start_to_acc_len=False)
        self.update_pending = True                # the first gate step loads the initial command values
        self._ohdr = None
        self._upstream_start = 0
        # fused mode
        self._plan = collections.deque()          # ('seq', seq0) | _Decision, in upstream order
        self._plan_cv = threading.Condition()
        self._plan_now = 0
        self._plan_step = 0
        self._fused_accs = None                   # [[XArray, XArray], [XArray, XArray]]
        self._set_busy = [False, False]           # pair holds a long integration that is not published yet
        self._long_index = 0                      # long integrations started (fused)
        self._dump_index = 0                      # dumps inside the current long integration (fused)
        self._open_set = None                     # pair of the long integration in progress (first seen, last not yet)
        self._stopping = False
        self.fused_dumps = 0                      # dumps accumulated by the contraction's epilogue (stat)
        # the upstream Corr finds its long accumulator through the ring it writes (no change to the pipeline script).  Only a
        # guaranteed reader that sees every span from the first can follow Corr's decisions one by one: the reader is
        # registered here, before anything is written; a second CorrAcc on the same ring, or a reader without the guarantee,
        # keeps the map path.
        self._iseqs = None
        # which way the long accumulation is done (module docstring): 'group' where the ring and the backend allow it, 'fused' only
        # when asked for, the reference's per-dump map otherwise
        want = accumulate if accumulate is not None else os.environ.get("XENG_CORRACC", "group")
        if want not in ("group", "fused", "map"):
            raise ValueError("CorrAcc: accumulate = %r (group | fused | map)" % (want,))
        in_repo_ring = getattr(iring, 'span_memory_outlives_release', False)
        self.group_dumps = max(1, min(16, int(os.environ.get("XENG_CORRACC_GROUP", "10"))))
        self.acc_mode = 'group' if (want in ('group', 'fused') and in_repo_ring and hasattr(self._bf, 'map_sum_i32')) else 'map'
        self._accdata2 = None                     # grouped mode: the second accumulator (long integrations alternate)
        if (want == 'fused' and in_repo_ring and hasattr(self._bf, 'bfXgpuKernelAsyncAcc') and guarantee
                and getattr(iring, 'long_accumulator', None) is None):
            iring.long_accumulator = self
            self._iseqs = iring.read(guarantee=True)

    def shutdown(self):
        """Give back the reader registered at construction when main() never ran (or raised before its first iteration): an
        unstarted generator's `finally` never runs, the reader would stay open and the upstream Corr would wait for room on a
        full ring for ever.  Idempotent; main() consumes the same generator, whose own `finally` closes the reader normally."""
        gen, self._iseqs = self._iseqs, None
        if gen is not None:
            if getattr(self.iring, 'long_accumulator', None) is self:
                self.iring.long_accumulator = None
            close = getattr(gen, 'close', None)     # (ring.py _SequenceReader: closes the registration even if never iterated)
            if close is not None:
                try:
                    close()
                except Exception:
                    pass
        with self._plan_cv:
            self._stopping = True
            self._plan_cv.notify_all()

    def __del__(self):
        try:
            if getattr(self, '_iseqs', None) is not None and not getattr(self, '_main_entered', False):
                self.shutdown()
        except Exception:
            pass

    # ------------------------------------------------------------------ the gate, one step per upstream sequence / span
    def _check_compat(self, gate, upstream_acc_len, upstream_start_time):
        if upstream_acc_len and gate.acc_len % upstream_acc_len != 0:
            self.log.error("CORRACC >> Requested acc_len %d incompatible with upstream integration %d" % (gate.acc_len, upstream_acc_len))
        if upstream_acc_len and gate.acc_len != 0 and ((gate.start_time - upstream_start_time) % upstream_acc_len != 0):
            self.log.error("CORRACC >> Requested start_time %d incompatible with upstream integration %d" % (gate.start_time, upstream_acc_len))

    def _begin_upstream_sequence(self, ihdr):
        """corr_acc_block.py:212-233: header of the output sequences, recovery of a running gate."""
        gate = self._gate
        ohdr = dict(ihdr)
        now, step = ihdr['seq0'], ihdr['acc_len']
        ohdr['upstream_acc_len'] = step
        ohdr.pop('fused_corracc', None)
        self._upstream_start = now
        self.sequence_proclog.update(ohdr)
        if gate.recover(now):
            self.log.info("CORRACC >> Recovering start time set to %d. Accumulating %d samples" % (gate.start_time, gate.acc_len))
            self._check_compat(gate, step, now)
            ohdr['acc_len'] = gate.acc_len
            ohdr['seq0'] = gate.start_time
        self._ohdr = ohdr

    def _decide(self, now, step):
        """corr_acc_block.py:237-296 for one upstream span starting at sample `now`."""
        gate, ohdr = self._gate, self._ohdr
        if self.update_pending:
            self.update_command_vals()
            gate.configure(now, self.command_vals['acc_len'], self.command_vals['start_time'])
            self.log.info("CORRACC >> New start time at %d. Accumulation: %d samples" % (gate.start_time, gate.acc_len))
            self._check_compat(gate, step, self._upstream_start)
            ohdr['acc_len'] = gate.acc_len
            ohdr['seq0'] = gate.start_time
        self.stats.update({'curr_sample': now})
        self.update_stats()
        if gate.acc_len == 0:                   # stop command (:257-263)
            gate.running = False
            d = _Decision('stop', now)
            d.state = 'stopped'
            return d
        begin = None
        if gate.try_start(now, step):
            begin = dict(ohdr)
            self.log.info("CORRACC >> Start time %d reached. Accumulating to %d (upstream accumulation: %d)" % (gate.start_time, gate.last, step))
        if not gate.running:
            d = _Decision('wait', now)
            d.state = 'waiting_start_missed' if now > gate.start_time else 'waiting'
            return d
        d = _Decision('acc', now)
        d.begin_hdr = begin
        d.state = 'running'
        d.first, d.last = now == gate.first, now == gate.last
        if d.last:
            gate.advance(step)
        return d

    # ------------------------------------------------------------------ fused mode: called by the upstream Corr's thread
    def plan_sequence(self, ohdr_upstream):
        """Corr is about to open an output sequence with this header (and will mark it 'fused_corracc')."""
        if self._fused_accs is None:
            n = self.igulp_size // 4
            self._fused_accs = [[XArray(shape=(n,), dtype='i32', space=self._bf.space_in) for _ in range(2)] for _ in range(2)]
        with self._plan_cv:
            self._begin_upstream_sequence(ohdr_upstream)
            self._abandon_open_set()              # (a long integration cut off by the end of the upstream sequence)
            self._plan_now = ohdr_upstream['seq0']
            self._plan_step = ohdr_upstream['acc_len']
            self._plan.append(('seq', self._plan_now))
            self._plan_cv.notify_all()

    def _abandon_open_set(self):
        """under _plan_cv: a long integration that will never see its last dump (new command, stop, upstream sequence
        ended) is never published: its accumulator pair is free again"""
        if self._open_set is not None:
            self._set_busy[self._open_set] = False
            self._open_set = None
            self._plan_cv.notify_all()

    def plan_dump(self):
        """Corr is about to enqueue the dump of its next integration: returns (accumulator, mode) for
        bfXgpuKernelAsyncAcc, or (None, 0) when this dump is not part of a long integration.  Waits -- as the classic path
        and the reference wait on the guaranteed ring -- while the accumulator pair it needs is still being published
        (downstream back-pressure), for as long as this block is alive."""
        with self._plan_cv:
            d = self._decide(self._plan_now, self._plan_step)
            self._plan_now += self._plan_step
            acc = None
            if d.kind != 'acc' or d.first:
                self._abandon_open_set()
            if d.kind == 'acc':
                if d.first:
                    self._long_index += 1
                    self._dump_index = 0
                    s = self._long_index & 1
                    t0 = time.time()
                    while self._set_busy[s] and not self._stopping:     # (never in normal operation: publishing one long
                        if not self._plan_cv.wait(0.05) and time.time() - t0 > 10.0:      # integration takes far less than accumulating one)
                            self.log.warning("CORRACC >> accumulator pair still unpublished (its output ring is full?): the upstream Corr waits")
                            t0 = time.time()
                    self._set_busy[s] = True
                    self._open_set = s
                d.acc_set = self._long_index & 1
                d.acc_half = self._dump_index & 1
                d.mode = 1 if self._dump_index < 2 else 2
                self._dump_index += 1
                d.nhalves = min(2, self._dump_index)
                if d.last:
                    self._open_set = None         # (stays busy until this block's thread has published it)
                acc = self._fused_accs[d.acc_set][d.acc_half]
                self.fused_dumps += 1
            self._plan.append(d)
            self._plan_cv.notify_all()
            return acc, d.mode

    def _next_plan(self, want_seq):
        t0 = time.time()
        with self._plan_cv:
            while not self._plan:          # (Corr queues the decision before it enqueues the dump, long before the span arrives)
                if not self._plan_cv.wait(0.05) and time.time() - t0 > 10.0:
                    self.log.warning("CORRACC >> no decision from the upstream Corr yet for a span that has arrived")
                    t0 = time.time()
            e = self._plan.popleft()
        is_seq = isinstance(e, tuple)
        if is_seq != want_seq:
            raise RuntimeError("CORRACC >> decision queue out of step with the input ring (%r)" % (e,))
        return e

    # ------------------------------------------------------------------ this block's thread
    def main(self):
        self._main_entered = True
        cpu_affinity.set_core(self.core)
        if self.gpu != -1:
            self._bf.set_device(self.gpu)
        self.bind_proclog.update({'ncore': 1, 'core0': cpu_affinity.get_core()})

        self.oring.resize(self.ogulp_size)
        oseq = ospan = None
        process_time = 0
        acquire_time = reserve_time = 0
        time_tag = 1
        self.update_stats({'state': 'starting'})
        # Fused mode publishes without stopping: the copy of a finished long integration into the (pinned-host) output span --
        # 383 MB over PCIe, 7 ms, twenty short integrations' worth -- is only enqueued; a short-lived helper thread waits for
        # it (interpreter lock released), commits the span and hands the accumulator pair back to Corr, while this thread
        # goes on following the upstream spans: corr-output never backs up into Corr and the X-engine keeps running.  (The
        # other pair accumulates the next long integration; one publish in flight at a time.)
        self._publishing = None                   # (helper thread, copy stamp, [exception])
        async_publish = hasattr(self._bf, 'copy_async')

        def start_publish(osp, acc_set, dst, src, wait_map=False):
            err = []

            def complete():
                try:
                    # (the library takes the device from the calling thread: without this the copies of a pipeline on GPU 1 would
                    # go onto GPU 0's copy stream and clock, and create a context on another pipeline's GPU)
                    if self.gpu != -1:
                        self._bf.set_device(self.gpu)
                    if wait_map:                  # grouped mode: the last group's sum, on the map stream, has written `src`
                        self._bf.map_sync()
                    # In pieces, each waited for before the next is enqueued: one 191 MB copy keeps the copy stream and the
                    # PCIe link to itself for 3.2 ms, and BeamformSumBeams' power sums (1 MB per gulp, same stream, same link)
                    # queue up behind it -- its thread stops, bf-output fills, Beamform stops, the input ring fills, Corr
                    # starves: the GPU sat idle for ~3 ms at every long integration (profiles/r04/blocks_gpu_idle.txt).
                    # Between two pieces the other copies get their turn.
                    nbytes, piece = src.nbytes, self.PUBLISH_PIECE_BYTES
                    for off in range(0, nbytes, piece):
                        m = min(piece, nbytes - off)
                        self._bf.copy_wait(self._bf.copy_async(XArray.window(dst.ptr + off, m, dst.space, dst), XArray.window(src.ptr + off, m, src.space, src)))
                    osp.close()
                except Exception as e:            # (re-raised by the block's thread at the next join)
                    err.append(e)
                finally:
                    if acc_set is not None:
                        with self._plan_cv:
                            self._set_busy[acc_set] = False
                            self._plan_cv.notify_all()
            th = threading.Thread(target=complete, name="corracc-publish", daemon=True)
            self._publishing = (th, None, err)
            th.start()

        def finish_publish():
            pub, self._publishing = self._publishing, None
            if pub is not None:
                pub[0].join()
                if pub[2]:
                    raise pub[2][0]
        try:
            with self.oring.begin_writing() as oring:
                prev_time = time.time()
                # (the command values are loaded by the first gate step, whichever thread makes it: set in __init__, not
                # here -- in fused mode the upstream Corr may already have planned dumps when this thread starts)
                for iseq in (self._iseqs if self._iseqs is not None else self.iring.read(guarantee=self.guarantee)):
                    ihdr = json.loads(iseq.header.tostring())
                    # (fused: Corr planned this sequence with THIS block; any other reader of the ring maps the spans itself)
                    fused = bool(ihdr.get('fused_corracc')) and getattr(self.iring, 'long_accumulator', None) is self
                    now, step = ihdr['seq0'], ihdr['acc_len']
                    if fused:
                        self._next_plan(want_seq=True)
                    else:
                        self._begin_upstream_sequence(ihdr)
                    grouped = not fused and self.acc_mode == 'group'
                    self.update_stats({'fused': fused, 'grouped': grouped, 'group_dumps': self.group_dumps if grouped else 0})
                    group, first_sum, acc_cur, nlong = [], True, self.accdata, 0
                    for ispan in iseq.read(self.igulp_size):
                        if ispan.size < self.igulp_size:
                            continue
                        if not fused and getattr(ispan, 'skipped', 0):
                            # spans overwritten before this reader got to them (ring.py): the sample count moves on, and a long
                            # integration they belonged to is lost -- realigned like a new upstream sequence (:221-236)
                            now += (ispan.skipped // self.igulp_size) * step
                            if self._gate.recover(now):
                                self._ohdr['acc_len'], self._ohdr['seq0'] = self._gate.acc_len, self._gate.start_time
                        d = self._next_plan(want_seq=False) if fused else self._decide(now, step)
                        if fused and d.now != now:
                            raise RuntimeError("CORRACC >> decision for sample %d, span of sample %d" % (d.now, now))
                        now += step
                        self.update_stats({'state': d.state})
                        if d.kind == 'stop':
                            finish_publish()
                            if oseq:
                                oseq.end()
                            oseq = None
                            continue
                        if d.kind == 'wait':
                            continue
                        if d.begin_hdr is not None:
                            finish_publish()            # (its span belongs to the sequence that ends here)
                            if oseq:
                                oseq.end()
                            self.sequence_proclog.update(d.begin_hdr)
                            oseq = oring.begin_sequence(time_tag=time_tag, header=json.dumps(d.begin_hdr), nringlet=iseq.nringlet)
                            time_tag += 1
                        curr_time = time.time()
                        acquire_time = curr_time - prev_time
                        prev_time = curr_time
                        if d.first:
                            curr_time = time.time()
                            reserve_time = curr_time - prev_time
                            prev_time = curr_time
                        if grouped:
                            if d.first:
                                # a new long integration: drop what an interrupted one had collected; the other accumulator (the
                                # previous one may still be on its way to the output ring)
                                group, first_sum = [], True
                                nlong += 1
                                if async_publish:
                                    if self._accdata2 is None:
                                        self._accdata2 = XArray(shape=self.accdata.shape, dtype='i32', space=self._bf.space_in)
                                    acc_cur = self._accdata2 if (nlong & 1) else self.accdata
                            group.append(ispan.data)          # (the reference keeps the span's memory alive: nothing is copied)
                            if len(group) >= self.group_dumps or d.last:
                                rv = self._bf.map_sum_i32(acc_cur, group, add=not first_sum)      # a (+)= b0 + b1 + ... in one pass
                                if rv != self._bf.BF_STATUS_SUCCESS:
                                    raise RuntimeError("CorrAcc map returned %d: %s" % (rv, self._bf.last_error()))
                                first_sum = False
                                # (the spans go back to the ring here, behind the stamp of the map stream: not reissued before the sum
                                # has read them -- DESIGN.md 4.8; nothing waits)
                                group = []
                        elif not fused:
                            idata = ispan.data_view('i32')
                            if d.first:
                                rv = self._bf.map_assign_i32(self.accdata, idata)      # "a = b"
                            else:
                                rv = self._bf.map_add_i32(self.accdata, idata)         # "a += b"
                            if rv != self._bf.BF_STATUS_SUCCESS:
                                raise RuntimeError("CorrAcc map returned %d: %s" % (rv, self._bf.last_error()))
                            # the input span is recycled when the loop advances: the map must have read it (this block's
                            # stream only: the X-engine and beamformer streams keep running)
                            self._bf.map_sync()
                        # (fused: the dump that produced this span has also accumulated it -- Corr commits a span only
                        # after its dump has completed)
                        curr_time = time.time()
                        process_time += curr_time - prev_time
                        prev_time = curr_time
                        if d.last:
                            result = acc_cur if grouped else self.accdata
                            if fused:
                                pair = self._fused_accs[d.acc_set]
                                result = pair[0]
                                if d.nhalves == 2:
                                    rv = self._bf.map_add_i32(pair[0], pair[1])
                                    if rv != self._bf.BF_STATUS_SUCCESS:
                                        raise RuntimeError("CorrAcc map returned %d: %s" % (rv, self._bf.last_error()))
                                    self._bf.map_sync()
                            finish_publish()            # (one copy in flight at a time: the previous one is long done)
                            ospan = WriteSpan(oseq.ring, self.ogulp_size, nonblocking=False)
                            odata = ospan.data_view('i32').reshape(result.shape)
                            if fused and async_publish:
                                start_publish(ospan, d.acc_set, odata, result)
                                ospan = None
                            elif grouped and async_publish:
                                start_publish(ospan, None, odata, result, wait_map=True)
                                ospan = None
                            else:
                                if grouped:
                                    self._bf.map_sync()
                                copy_array(odata, result)     # (synchronous: complete before the span is committed)
                                ospan.close()
                                ospan = None
                                if fused:
                                    with self._plan_cv:
                                        self._set_busy[d.acc_set] = False
                                        self._plan_cv.notify_all()
                            curr_time = time.time()
                            process_time += curr_time - prev_time
                            prev_time = curr_time
                            self.perf_proclog.update({'acquire_time': acquire_time, 'reserve_time': reserve_time,
                                                      'process_time': process_time})
                            self.update_stats({'last_end_sample': d.now})
                            process_time = 0
                finish_publish()
                if ospan:
                    ospan.close()
                if oseq:
                    oseq.end()
        finally:
            if self._publishing is not None:      # (an exception: the copy in flight still writes its span -- wait before it is let go)
                self._publishing[0].join()
                self._publishing = None
            with self._plan_cv:                   # a Corr thread waiting for an accumulator pair must not hang
                self._stopping = True
                self._plan_cv.notify_all()
