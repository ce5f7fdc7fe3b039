/*
 * xeng.h -- C ABI of the MI355X-native LWA-352 X-engine library (libxeng.so).
 *
 * This is the drop-in boundary for the hot path of realtimeradio/caltech-bifrost-dsp:
 * the functions below are what the reference's Python blocks reach through
 * `from bifrost.libbifrost import _bf` (ctypes).  Every entry point cites the
 * reference call site it replaces (paths relative to
 * /root/reference/pipeline/lwa352_pipeline/blocks unless stated).
 *
 * Conventions (same as the reference's BFstatus convention, corr_block.py:254):
 *   - every function returns int, 0 (XENG_STATUS_SUCCESS) on success, non-zero
 *     on error; nothing throws across the ABI; xengGetLastError() returns a
 *     thread-local message for the last failure.
 *   - the caller owns every data buffer; the library owns contexts and scratch.
 *   - contexts are process-global singletons like the reference's (one xGPU
 *     context: corr_block.py:249-256; one beamformer context shared by Beamform
 *     and BeamformSumBeams: beamform_sum_beams_block.py:186-187).
 *   - plain pointers and sizes only; "dev" pointers are HIP device pointers,
 *     "host" pointers are ordinary (ideally pinned) host memory.
 *
 * Two layers are exported:
 *   xeng*   raw-pointer functions (sizes are runtime arguments of Configure /
 *           Initialize; xGPU's were compile-time, install_xgpu.sh:5), and
 *   bf*     adapters with bifrost's names and BFarray* argument shapes, so a
 *           bifrost build could bind them 1:1.
 */
#ifndef XENG_H_
#define XENG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XENG_STATUS_SUCCESS            0
#define XENG_STATUS_INVALID_ARGUMENT   1
#define XENG_STATUS_INVALID_STATE      2
#define XENG_STATUS_DEVICE_ERROR       3
#define XENG_STATUS_UNSUPPORTED        4
#define XENG_STATUS_MEM_ALLOC_FAILED   5
#define XENG_STATUS_WOULD_BLOCK        6   /* a call that was asked not to wait would have had to */
#define XENG_STATUS_END_OF_DATA        7   /* ring readers: no more sequences / no more data in this sequence */

/* memory spaces, numbered as bifrost's BFspace [from memory of bifrost/src/bifrost/memory.h] */
#define XENG_SPACE_AUTO       0
#define XENG_SPACE_SYSTEM     1
#define XENG_SPACE_CUDA       2   /* device memory (HIP) -- keeps bifrost's name for the space */
#define XENG_SPACE_CUDA_HOST  3   /* pinned host memory */

/* mirrors bifrost's BFarray (bifrost/src/bifrost/array.h) [struct layout from memory, unverifiable here] */
#define XENG_MAX_DIMS 8
typedef struct XENGarray_ {
    void *data;
    int   space;
    int   dtype;
    int   ndim;
    long  shape[XENG_MAX_DIMS];
    long  strides[XENG_MAX_DIMS];
    int   immutable;
    int   big_endian;
    int   conjugated;
} XENGarray;

const char *xengGetLastError(void);
const char *xengVersion(void);

/* ---------------------------------------------------------------- device / memory plumbing
 * replaces bifrost.device.set_device / stream_synchronize and BFArray(space='cuda'|'cuda_host')
 * + copy_array (corr_acc_block.py:315-317, beamform_block.py:433, copy_block.py:146). */
int xengGetDeviceCount(int *count);
int xengSetDevice(int gpu);
int xengGetDevice(int *gpu);
int xengDeviceSynchronize(void);
int xengGetDeviceInfo(int gpu, int *num_cu, int *clock_khz, size_t *total_mem, char *name, int name_len);
/* "dddd:bb:dd.f" of a device: /sys/bus/pci/devices/<id>/{numa_node,local_cpulist} name the host cores next to it (the
 * reference pins every block thread, corr_block.py:336; sharding.pin_rank pins a rank to its GPU's NUMA node) */
int xengGetDevicePciBusId(int gpu, char *bus_id, int len);
int xengMalloc(void **ptr, size_t nbytes, int space);          /* XENG_SPACE_CUDA or XENG_SPACE_CUDA_HOST */
int xengFree(void *ptr, int space);
int xengMemcpy(void *dst, const void *src, size_t nbytes);     /* any direction, synchronous on return */
int xengMemcpyAsync(void *dst, const void *src, size_t nbytes);/* on the library's copy stream */
int xengMemset(void *dst, int value, size_t nbytes);
int xengStreamSynchronize(void);                               /* all library streams of the current device */

/* ---------------------------------------------------------------- stamps: what has been enqueued so far
 * A stamp names everything the library has enqueued on its streams of the current device up to now, by any thread -- including
 * gulps handed to xengXgpuKernelAsync whose contraction has not been enqueued yet.  It is complete when all of that has run.
 * Taking one enqueues nothing and costs a few loads; asking about one records an event only on a stream that is still
 * busy.  The span rings below stamp every allocation at the moment its last user lets go and reissue or free it only once the
 * stamp is complete: GPU memory lifetime does not rest on who dropped which Python reference when.
 * No reference counterpart: a bifrost ring is one circular buffer that is never freed while the pipeline runs. */
typedef struct xengStamp_ { unsigned long long w[16]; } xengStamp;
/* stream classes a stamp waits for (bits 32..36 of w[0]; xengStampNow sets all): the X-engine (staging stream, contractions,
 * registered gulps), the CorrAcc map stream, the beamformer stream, the copy stream, the span consumers' stream */
#define XENG_STREAMS_XGPU      1u
#define XENG_STREAMS_MAP       2u
#define XENG_STREAMS_BEAM      4u
#define XENG_STREAMS_COPY      8u
#define XENG_STREAMS_CONSUMER 16u
#define XENG_STREAMS_ALL      31u
/* a buffer that contractions only WRITE (a visibility span, a long accumulator; never handed over as a gulp): its stamp names
 * the last launch enqueued into that very buffer instead of every launch enqueued so far */
#define XENG_STREAMS_XGPU_OUT 32u
int xengStampNow(xengStamp *stamp);
int xengStampNowFor(xengStamp *stamp, const void *buf, unsigned classes);   /* the stamp of one buffer whose users are `classes` (0: all) */
/* non-blocking: *done = 1 when Wait would not wait; *waitable (may be NULL) = 0 when the stamp waits for an X-engine launch
 * that nobody has enqueued yet (only the owner of those gulps can end that: a dump, or xengXgpuReset) */
int xengStampDone(const xengStamp *stamp, int *done, int *waitable);
int xengStampWait(const xengStamp *stamp);

/* ---------------------------------------------------------------- span rings
 * The bookkeeping of the ring the blocks sit on -- bifrost.ring.Ring in the reference (lwa352-pipeline.py:147-155; protocol
 * used by the hot-path blocks: corr_block.py:342-350,388,433-452; corr_acc_block.py:313-318; beamform_block.py:440-450) --
 * for pipelines that run without bifrost: committed spans, reader cursors, back-pressure and the free list of span
 * allocations, native.  caltech-bifrost-dsp_amd/ring.py wraps these calls in the reference's Python protocol.
 * Every span is its own reference-counted allocation in `space` (system / device / pinned): a reader that keeps its span
 * handle may go on reading the bytes after the ring has moved on.  Calls that can wait take `may_block`: 0 returns
 * XENG_STATUS_WOULD_BLOCK instead of waiting (a caller that holds an interpreter lock asks first and gives the lock up only
 * for a call that has to sleep). */
typedef struct xengRing_ xengRing;
int xengRingCreate(xengRing **ring, const char *name, int space);
int xengRingDestroy(xengRing *ring);     /* wakes every waiter; spans still referenced stay valid until released */
int xengRingResize(xengRing *ring, size_t contig_bytes, size_t total_span);        /* ring.resize(): capacity in bytes (0: 4 x contig) */
/* Which of the library's streams touch this ring's spans (XENG_STREAMS_*; calls accumulate).  Every USER of a ring -- its
 * writer and each of its readers -- calls this exactly once (classes may be 0: a user that enqueues nothing on the spans, e.g.
 * a host reader whose copies are complete when it lets go).  A released span waits for the declared union -- the beamformer's
 * rings XENG_STREAMS_BEAM, Corr's XENG_STREAMS_XGPU ... -- so that a beam span is not held back by the 200 us contraction that
 * happened to be enqueued before its release; but ONLY while every user has declared: the ring counts the readers it has ever
 * opened plus its writer, and as long as there are more of those than declarations (a block that does not know the call, a
 * test reader with kernels of its own) every stamp waits for everything the library had enqueued, as on an undeclared ring. */
int xengRingDeclareStreams(xengRing *ring, unsigned classes);
/* the classes a span released now would wait for (XENG_STREAMS_* union, or all of them), declarations made, users seen */
int xengRingGetStampClasses(xengRing *ring, unsigned *classes, unsigned *declared, unsigned *users);
/* system-space rings hand out fresh zero-filled memory per span by default; on != 0 recycles released spans as the device /
 * pinned rings always do (contents: whatever the last user left, as in a circular bifrost ring) */
int xengRingSetRecycle(xengRing *ring, int on);
/* counters: allocations made, really freed, reissued from the free list, waits for a stamp at reissue, bytes skipped by readers */
int xengRingGetInfo(xengRing *ring, size_t *capacity, size_t *live_bytes, size_t *pool_bytes, int *nreaders, long long *nseq,
                    unsigned long long counters[5]);
/* writer: one at a time.  BeginSequence ends the sequence that was open. */
int xengRingBeginSequence(xengRing *ring, long long time_tag, const void *header, size_t header_len, int nringlet, long long *seq);
int xengRingEndSequence(xengRing *ring, long long seq);
int xengRingEndWriting(xengRing *ring);
/* WriteSpan(ring, nbytes, nonblocking) (corr_block.py:435): waits until the ring has room (guaranteed readers apply
 * back-pressure; without one the oldest span is overwritten, as in bifrost), then hands out span memory -- a released
 * allocation whose stamp is complete, or a fresh zero-filled one.  nonblocking: XENG_STATUS_WOULD_BLOCK + message when full. */
int xengRingReserve(xengRing *ring, long long seq, size_t nbytes, int nonblocking, int may_block, void **data, long long *span);
int xengRingCommit(xengRing *ring, long long seq, long long span, size_t nbytes);       /* the first nbytes of the span become readable */
/* publish the caller's own memory as the next span without a copy (a replay source: dummy_source_block.py:207-222 re-sends
 * the same gulps); the caller keeps it valid and unchanged while readers may hold it */
int xengRingCommitExternal(xengRing *ring, long long seq, void *data, size_t nbytes, int may_block);
/* readers.  A reader registered late starts at the oldest sequence still in the ring; data overwritten before a reader got
 * to it is skipped by whole gulps (*skipped bytes), as a bifrost reader skips frames. */
int xengRingOpenReader(xengRing *ring, int guarantee, int *reader);
int xengRingCloseReader(xengRing *ring, int reader);
/* XENG_STATUS_END_OF_DATA once writing has ended and every sequence has been seen; *header stays valid until the next call */
int xengRingNextSequence(xengRing *ring, int reader, int may_block, long long *seq, long long *time_tag, int *nringlet,
                         const void **header, size_t *header_len);
/* iseq.read(gulp_nbytes): first moves the reader `advance` bytes on (the previous gulp), then waits for the next gulp.
 * *nbytes < gulp_nbytes only for the short tail of an ended sequence; XENG_STATUS_END_OF_DATA when the sequence is over.
 * *span holds a reference on the memory: xengRingSpanRelease when done with it. */
int xengRingAcquire(xengRing *ring, int reader, size_t advance, size_t gulp_nbytes, int may_block, void **data, size_t *nbytes,
                    long long *span, size_t *skipped);
/* The same, but a gulp that lies in TWO committed spans comes back as two windows (*nparts = 2: data / nbytes / span of each,
 * in order) instead of a gathered copy -- for a consumer that can take its gulp in two parts (xengBeamformRunParts). */
int xengRingAcquireParts(xengRing *ring, int reader, size_t advance, size_t gulp_nbytes, int may_block, void *data[2], size_t nbytes[2],
                         long long span[2], int *nparts, size_t *skipped);
int xengRingSpanRelease(long long span);      /* a span handle from Reserve or Acquire: the last release stamps the allocation */
/* tests: tickets of a fake backend instead of the library's stream clocks (and a free list for system-space rings) */
typedef void (*xengRingStampNowFn)(void *user, unsigned long long stamp[2]);
typedef int (*xengRingStampDoneFn)(void *user, const unsigned long long stamp[2]);
typedef void (*xengRingStampWaitFn)(void *user, const unsigned long long stamp[2]);
int xengRingSetStampHooks(xengRing *ring, xengRingStampNowFn now, xengRingStampDoneFn done, xengRingStampWaitFn wait, void *user);

/* ---------------------------------------------------------------- X-engine (Corr)
 * replaces _bf.bfXgpuInitialize / bfXgpuKernel / bfXgpuCorrelate / bfXgpuGetOrder /
 * bfXgpuSubSelect / bfXgpuReorder. */

/* Sizes the next xengXgpuInitialize will use (xGPU compile-time NSTATION/NFREQUENCY/NTIME,
 * install_xgpu.sh:5; block args corr_block.py:221-231).  max_gulps_per_flush bounds how many
 * gulps are held back (kept in HBM and contracted in one launch at dump time; more gulps than
 * that are flushed early and accumulated in out_dev); 0 picks a default.  Defaults before any call: 352, 2, 96, 480. */
int xengXgpuConfigure(int nstand, int npol, int nchan, int ntime_gulp, int max_gulps_per_flush);

/* corr_block.py:251-256, xgpu_test.py:76.  Creates (or re-creates) the process-global context on `gpu`. */
int xengXgpuInitialize(int gpu);
int xengXgpuDestroy(void);

/* corr_block.py:445, xgpu_test.py:81-83.  in_dev: uint8[ntime_gulp][nchan][nstand][npol] 4+4 bit
 * (hi nibble real, lo nibble imag).  out_dev: int32[2][nchan][per_chan] planar re|im in xGPU
 * register-tile order (corr_block.py:27-58).  Gulps accumulate until a call with doDump=1, after
 * which out_dev holds the sum over all gulps since the previous dump and the accumulation restarts.
 * The same out_dev must be passed for every gulp of one integration (as Corr.main does: one
 * WriteSpan per integration, corr_block.py:433-435).  Synchronous: on return the input has been
 * consumed and, if doDump, the output is complete. */
int xengXgpuKernel(const void *in_dev, void *out_dev, int doDump);

/* Same, but enqueue only: the caller must keep in_dev valid AND UNCHANGED until the dump that consumes
 * it has completed (xengXgpuSync, or xengXgpuSyncLag covering that dump) before reading out_dev or
 * recycling in_dev.  On the default path the contraction kernel reads the gulps in place at dump time
 * (corner turn fused into its LDS staging): no copy of the gulp is made, in_dev must be 16-byte
 * aligned.  (No reference counterpart; this is how a ring-resident pipeline streams gulps.) */
int xengXgpuKernelAsync(const void *in_dev, void *out_dev, int doDump);
/* The same with CorrAcc's long accumulation (corr_acc_block.py:298-306, "a = b" / "a += b") fused into the dump: when
 * doDump is set and acc_dev is not NULL, every visibility the dump stores to out_dev is also assigned (acc_mode 1) or added
 * (acc_mode 2) to acc_dev, a planar int32 buffer of the same size and layout -- one pass over the accumulator in the
 * contraction's epilogue instead of a separate xengMapAssignI32 / xengMapAddI32 over both buffers (382 MB less HBM
 * traffic per config-2 dump).  Dumps that name the same accumulator are ordered; consecutive dumps overlap only when the
 * caller alternates between two accumulators (and adds them at the end of the long integration).  Readers of acc_dev:
 * xengXgpuSync / xengXgpuSyncLag as for out_dev.  XENG_STATUS_UNSUPPORTED on the non-default contraction paths. */
int xengXgpuKernelAsyncAcc(const void *in_dev, void *out_dev, int doDump, void *acc_dev, int acc_mode);
/* The two enqueue-only calls above wait when the caller is 256 launches ahead of the GPU (every launch owns one of 256
 * completion events).  This form never waits: XENG_STATUS_WOULD_BLOCK then, nothing enqueued; xengXgpuWaitLaunchSlot blocks
 * until a launch may be enqueued again.  (A caller that holds an interpreter lock tries, and gives the lock up to wait.) */
int xengXgpuTryKernelAsyncAcc(const void *in_dev, void *out_dev, int doDump, void *acc_dev, int acc_mode);
int xengXgpuWaitLaunchSlot(void);
/* A gulp handed over as the SLAB OF PACKETS it arrived in (round 4; layout: "Ingest" below; the gulp's window starts at seq0,
 * channel 0 of the pipeline is chan0_pipeline).  Enqueue-only like xengXgpuKernelAsync[Acc], same rules for out_dev / acc_dev /
 * doDump, and the slab must stay valid and unchanged until the dump that consumes it has completed.  On the device, without a host
 * round trip (round 5): every packet of the deployed geometry (one packet per sample and 64-input block, all channels) is entered
 * into an index -- the last packet that carries a (sample, block) wins, as in xengSnap2Unpack -- and the contraction kernel reads the
 * voltages out of the packets WHERE THEY LIE, through a table of their offsets: in order, shifted by lost packets, reordered,
 * duplicated, mixed with foreign or out-of-window packets, any packet count -- no scatter pass, no copy of the gulp at all;
 * samples nobody carries read as zero.  Only a slab that holds valid packets of ANOTHER geometry (several channel blocks per sample,
 * fewer inputs per packet), or one the table cannot describe (stride not a multiple of 16, unaligned, >= 2 GiB), goes through
 * zero-fill + scatter into the library's staging area, with the rules of xengSnap2Unpack.  The results are those of unpack +
 * correlate either way.  Slabs and plain gulps cannot be mixed inside one integration.  xengXgpuGetSlabFallbacks: gulps that took
 * the scatter since it was last called; xengXgpuGetSlabStats: those, and the gulps read in place whose packets were not all in
 * place (both wait for the staging stream).  No reference counterpart: bifrost's capture scatters on the CPU. */
int xengXgpuKernelAsyncSlab(const void *packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int chan0_pipeline, void *out_dev,
                            int doDump, void *acc_dev, int acc_mode);
int xengXgpuTryKernelAsyncSlab(const void *packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int chan0_pipeline, void *out_dev,
                               int doDump, void *acc_dev, int acc_mode);      /* never waits: see xengXgpuTryKernelAsyncAcc */
int xengXgpuGetSlabFallbacks(int *nfallback);
int xengXgpuGetSlabStats(int *nscattered, int *nirregular);
int xengXgpuSync(void);
/* Wait until all but the last `lag` (0..3) dumps are complete -- lag 1 lets a streaming caller enqueue
 * integration n+1 (into a different out_dev) before it waits for integration n, so the contraction of
 * n+1 fills the CUs that the contraction of n vacates.  lag 0 waits for the latest dump. */
int xengXgpuSyncLag(int lag);
/* The non-blocking form: *done = 1 when xengXgpuSyncLag(lag) would return without waiting.  (A caller that shares an
 * interpreter lock with other threads asks first and only gives the lock up for a call that really has to wait.) */
int xengXgpuDumpDone(int lag, int *done);

/* Drop the gulps staged and the partial sums accumulated since the last dump (an integration that
 * is abandoned, e.g. when a new start_time command interrupts it: corr_block.py:392-404 resets
 * `start` mid-integration).  No reference counterpart: xGPU would silently carry the partial sums
 * into the next integration. */
int xengXgpuReset(void);

/* xgpu_test.py:86-89: host-buffer variant (H2D, kernel, D2H on dump). */
int xengXgpuCorrelate(const void *in_host, void *out_host, int doDump);

/* corr_block.py:317-333.  Host arrays: antpol_to_input int32[nstand][npol];
 * antpol_to_bl, is_conj int32[nstand][nstand][npol][npol] indexed [s0][s1][p0][p1]
 * (corr_subsel_block.py:248-250).  is_conj=1: negate the stored imaginary part to obtain
 * x[s0,p0]*conj(x[s1,p1]) (corr_output_full_block.py:582-591). */
int xengXgpuGetOrder(const int32_t *antpol_to_input, int32_t *antpol_to_bl, int32_t *is_conj);

/* corr_subsel_block.py:298.  in_dev: planar xGPU buffer; out_dev: int32[nchan/nchan_sum][nvis][2];
 * vismap_dev/conj_dev: int32[nvis] device arrays. */
int xengXgpuSubSelect(const void *in_dev, void *out_dev, const int32_t *vismap_dev,
                      const int32_t *conj_dev, int nvis, int nchan_sum);

/* corr_output_full_block.py:669 + :461-467 / :512-519 done on the device: planar xGPU buffer -> the packet
 * payloads CorrOutputFull sends, one per dual-pol baseline s0 <= s1 in sending order
 * (k = s0*nstand - s0(s0-1)/2 + s1 - s0): out_dev int32[nstand(nstand+1)/2][npol][npol][nchan][2] (fmt 0,
 * send_packets_py) or [..][nchan][npol][npol][2] (fmt 1, COR).  Maps: device copies of the GetOrder
 * arrays.  The caller must have synchronised the contraction that produced in_dev.  Synchronous. */
int xengXgpuPacketize(const void *in_dev, void *out_dev, const int32_t *antpol_to_bl_dev,
                      const int32_t *is_conj_dev, int fmt);

/* corr_output_full_block.py:669.  Host: planar xGPU buffer -> int32[nstand][nstand][npol][npol][nchan][2]. */
int xengXgpuReorder(const void *in_host, void *out_host, const int32_t *antpol_to_bl, const int32_t *is_conj);

/* sizes of the current context */
int xengXgpuGetInfo(int *nstand, int *npol, int *nchan, int *ntime_gulp, int64_t *matlen, int *max_gulps);

/* which contraction path the current context runs: fused_corner_turn = 1 when gulps are read in place
 * (ninput % 16 == 0, ntime_gulp % 96 == 0, not disabled with XENG_RAW=0), else the two-pass path
 * (corner turn into a fragment-major staging area); fp6 = 1 for the opt-in XENG_MFMA=fp6 experiment. */
int xengXgpuGetPath(int *fused_corner_turn, int *fp6);
/* the contraction kernel plain and slab launches of the current context take: 4 waves per work-group on v_mfma_i32_32x32x32_i8
 * (mfma_k 32: the default, the two-pass path, and always for dumps that feed a long accumulator) or, with XENG_KLOOP=16, 8 waves
 * on v_mfma_i32_16x16x64_i8 (xcorr_fused16.h; mfma_k 64) */
int xengXgpuGetKernel(int *waves_per_group, int *mfma_k);

/* profiling: HIP events around each kernel on the context's stream.  GetTimes returns and clears
 * the totals (ms) and launch counts since the last call: [0]=corner turn (two-pass path) or raw
 * gulp copy (synchronous calls on the fused path), [1]=MFMA contraction. */
int xengXgpuSetProfiling(int enable);
int xengXgpuGetTimes(double ms[2], int count[2]);

/* ---------------------------------------------------------------- Ingest (SNAP2 F-engine packets)
 * Scatter received packets into a gulp on the device.  In the reference this scatter happens on the CPU inside
 * bifrost's UDP capture, which capture_block.py:296-305 only configures; the packet format is pinned by the
 * reference's transmitters (test_tx_vectors.py:38-48,103-108; test_tx_mt.c:39-49): 32-byte big-endian header
 * `>QLHHHHLLL` = seq, sync_time, npol, npol_tot, nchan, nchan_tot, chan_block_id, chan0, pol0, then
 * u8[nchan][npol] 4+4-bit samples.  packets_dev: npkt packets, pkt_stride bytes apart (any order, duplicates
 * allowed).  out_dev: u8[ntime][nchan_tot][npol_tot]; row c of a packet lands at
 * [seq - seq0][chan0 - chan0_pipeline + c][pol0 ..].  Packets outside the window [seq0, seq0+ntime) or outside
 * the gulp geometry are dropped and counted.  clear != 0: samples that no packet covers read as 0 (blanked) -- a slab
 * that covers the whole gulp is scattered without any zero-fill, otherwise the gulp is zero-filled and scattered again.
 * Synchronous; the counters may be NULL. */
int xengSnap2Unpack(const void *packets_dev, int npkt, size_t pkt_stride, void *out_dev, uint64_t seq0, int ntime,
                    int chan0_pipeline, int nchan_tot, int npol_tot, int clear, int *nplaced, int *ndropped);
/* Same, enqueue only, on the X-engine's staging stream: a gulp unpacked this way and then passed to
 * xengXgpuKernelAsync is complete before the contraction of its dump reads it.  No counters are returned. */
int xengSnap2UnpackAsync(const void *packets_dev, int npkt, size_t pkt_stride, void *out_dev, uint64_t seq0, int ntime,
                         int chan0_pipeline, int nchan_tot, int npol_tot, int clear);
/* Packets the enqueue-only calls have dropped (out of window / foreign / malformed) since this was last called; waits for
 * the staging stream and clears the count.  The synchronous call reports its own drops in *ndropped. */
int xengSnap2GetAsyncDrops(int *ndropped);

/* ---------------------------------------------------------------- test / bench harness (no pipeline calls these)
 * Emulator side of the F-engine link: a receiver reuses its slab buffers; this re-stamps the sequence numbers of a device-resident
 * slab for its next window -- packet p gets seq0 + p / pkts_per_seq (big-endian, header bytes 0..7), nothing else changes.  Complete
 * on return.  (The Python side of the harness lives in the extension's `_xfast.bench` sub-module: a source and sinks that are not
 * Python threads.) */
int xengSnap2StampSeq(void *packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int pkts_per_seq);

/* ---------------------------------------------------------------- CorrAcc
 * replaces bifrost.map "a = b" / "a += b" on int32 (corr_acc_block.py:304,306).  Device pointers;
 * enqueued on the library's map stream; xengStreamSynchronize() (corr_acc_block.py:317) completes it. */
int xengMapAssignI32(void *a_dev, const void *b_dev, size_t nwords);
int xengMapAddI32(void *a_dev, const void *b_dev, size_t nwords);
/* a = (add ? a : 0) + srcs[0] + ... + srcs[nsrc - 1] in one pass (srcs: HOST array of nsrc device pointers, 1 <= nsrc <=
 * XENG_MAP_SUM_MAX).  The reference adds every dump as it arrives (corr_acc_block.py:298-306: 574 MB of traffic per config-2
 * dump); int32 addition wraps, so any grouping gives the same words, and with 288 GB of HBM CorrAcc keeps the spans of a
 * group of dumps and sums them together: 191 + 382 / nsrc MB per dump.  Map stream, like the two calls above. */
#define XENG_MAP_SUM_MAX 16
int xengMapSumI32(void *a_dev, const void *const *srcs_dev, int nsrc, size_t nwords, int add);
int xengMapSync(void);   /* wait for the map stream only */

/* ---------------------------------------------------------------- Beamformer
 * replaces _bf.bfBeamformInitialize / Run / Integrate / IntegrateSingleBeam. */

/* beamform_block.py:251-253.  ntime_blocks==0: voltage mode.  >0: "integrated power" mode the reference
 * marks experimental (beamform_block.py:108-110) -- implemented as Run then Integrate (parity unpinned). */
int xengBeamformInitialize(int gpu, int ninput, int nchan, int ntime, int nbeam, int ntime_blocks);
int xengBeamformDestroy(void);

/* beamform_block.py:446-449.  in_dev uint8[ntime][nchan][ninput] 4+4 bit; weights_dev
 * cf32[nchan][nbeam][ninput] interleaved; out_dev cf32[nchan][nbeam][ntime]:
 * out[c,b,t] = sum_i w[c,b,i]*x[t,c,i] (beamformer_test.py:76-84).  Asynchronous on the beamformer
 * stream; xengBeamformSync()/xengStreamSynchronize() is the BFSync() of beamform_block.py:450. */
int xengBeamformRun(const void *in_dev, void *out_dev, const void *weights_dev);
/* Same, for callers that know when the weights change: the library re-splits the fp32 weights into the
 * bf16 terms its MFMA kernel uses only when (weights_dev, weights_version) differs from the last call
 * (version 0 = always re-split, which is what xengBeamformRun / bfBeamformRun do). */
int xengBeamformRunVersioned(const void *in_dev, void *out_dev, const void *weights_dev, long long weights_version);
/* Run* is enqueue-only except in the integrated-power mode right after a weight upload, where it waits once for the routing
 * answer of the new weights.  This form never waits: XENG_STATUS_WOULD_BLOCK then (the weights are prepared and remembered;
 * call xengBeamformRunVersioned with the same arguments to wait and run). */
int xengBeamformTryRunVersioned(const void *in_dev, void *out_dev, const void *weights_dev, long long weights_version);
/* The gulp in two parts: samples [0, ntime0) at in0_dev, samples [ntime0, ntime) at in1_dev -- two consecutive spans of the
 * input ring taken as ONE beamformer gulp, one launch, no gathered copy.  The reference's Beamform reads GPU_NGULP = 2 capture
 * gulps per call (lwa352-pipeline.py:172,279-282: ntime_gulp = 2 x 480) out of bifrost's circular buffer, where two gulps
 * are contiguous; on a ring of separate spans this call gives the same.  RunVersioned semantics otherwise; the Try form never
 * waits (see xengBeamformTryRunVersioned). */
int xengBeamformRunParts(const void *in0_dev, int ntime0, const void *in1_dev, void *out_dev, const void *weights_dev, long long weights_version);
int xengBeamformTryRunParts(const void *in0_dev, int ntime0, const void *in1_dev, void *out_dev, const void *weights_dev, long long weights_version);
/* The gulp as the SLABS OF PACKETS it arrived in (round 4; cf. xengXgpuKernelAsyncSlab, layout: "Ingest" below): one slab
 * (packets1_dev NULL) or two consecutive ones -- samples [0, ntime0) from seq0 on, samples [ntime0, ntime) from seq0 + ntime0 on;
 * ntime0 a multiple of 16, inputs a multiple of 16.  Each slab is verified on the beam stream; a regular one is read by the
 * beamformer kernels where it lies, any other is scattered into the context's scratch gulp first (xengSnap2Unpack's rules), so
 * the beams are those of unpack + Run either way -- bit for bit, the same kernels do the arithmetic.  RunVersioned semantics
 * otherwise; the slabs must stay valid and unchanged until the call's kernels have completed (xengBeamformMark).
 * xengBeamformGetSlabFallbacks: parts that took the scatter since it was last called (waits for the beam stream). */
int xengBeamformRunSlabs(const void *packets0_dev, int npkt0, int ntime0, const void *packets1_dev, int npkt1, size_t pkt_stride,
                         uint64_t seq0, int chan0_pipeline, void *out_dev, const void *weights_dev, long long weights_version);
int xengBeamformTryRunSlabs(const void *packets0_dev, int npkt0, int ntime0, const void *packets1_dev, int npkt1, size_t pkt_stride,
                            uint64_t seq0, int chan0_pipeline, void *out_dev, const void *weights_dev, long long weights_version);   /* never waits: see xengBeamformTryRunVersioned */
int xengBeamformGetSlabFallbacks(int *nfallback);
/* (round 5) On a lossy link -- more than a quarter of the parts of the last eight calls not regular -- the beamformer reads the parts
 * where they lie as well, through a packet index built by the verify launch (lost, shifted, reordered, duplicated packets; samples
 * nobody carries read as zero; the int8 and bf16 kernels, not the fp32 one); XENG_SLAB_TABLES=1 / 0 pins that on / off.
 * xengBeamformGetSlabStats: parts scattered, and parts read through an index that was not regular, since the last call. */
int xengBeamformGetSlabStats(int *nscattered, int *nirregular);

/* beamform_sum_beams_block.py:243-246.  in_dev cf32[nchan][nbeam][ntime];
 * out_dev f32[nbeam/2][ntime/ntime_sum][nchan][4] = [XX, YY, Re XY*, Im XY*]. */
int xengBeamformIntegrate(const void *in_dev, void *out_dev, int ntime_sum);

/* beamform_sum_single_beam_block.py:114: one dual-pol beam -> f32[ntime/ntime_sum][nchan][4]. */
int xengBeamformIntegrateSingleBeam(const void *in_dev, void *out_dev, int ntime_sum, int beam_id);
int xengBeamformSync(void);
/* Completion tickets on the beamformer's stream: Mark returns a ticket for everything enqueued so far (Run, Integrate,
 * by any thread), Wait blocks until that point has been reached.  They let the Beamform / BeamformSumBeams blocks keep
 * several gulps in flight and commit each output span when its own kernels are done (no reference counterpart: the
 * reference waits for the whole stream after every gulp, beamform_block.py:450). */
int xengBeamformMark(unsigned long long *ticket);
int xengBeamformWait(unsigned long long ticket);
int xengBeamformTicketDone(unsigned long long ticket, int *done);   /* non-blocking: *done = 1 when Wait would not wait */
int xengBeamformSetProfiling(int enable);
int xengBeamformGetTimes(double ms[2], int count[2]);   /* [0]=Run, [1]=Integrate */
/* How the last weight upload was routed (waits for the beam stream): (channel, beam tile) pairs in all, pairs that run on
 * the bf16x3 kernel because their fixed-point image would not hold the 1e-5 bar, and outlier inputs that the int8x3
 * kernel adds in fp32 (summed over tiles).  Zeros for the bf16x3 / f32 modes.  No reference counterpart: the
 * reference's cuBLAS CF32 GEMM (bf_src/cublas_beamform.cu:248-276) has one route. */
int xengBeamformGetRouteInfo(int *tiles_total, int *tiles_bf16, int *outlier_inputs);

/* ---------------------------------------------------------------- bifrost-named adapters
 * Exact argument shapes of the reference's call sites; data pointers are taken from the
 * BFarray-like structs, sizes from the configured context. */
int bfXgpuInitialize(XENGarray *in, XENGarray *out, int gpu_dev);                      /* corr_block.py:253 */
int bfXgpuKernel(XENGarray *in, XENGarray *out, int doDump);                           /* corr_block.py:445 */
int bfXgpuCorrelate(XENGarray *in, XENGarray *out, int doDump);                        /* xgpu_test.py:86-89 */
int bfXgpuGetOrder(XENGarray *antpol_to_input, XENGarray *antpol_to_bl, XENGarray *is_conj); /* corr_block.py:331-333 */
int bfXgpuSubSelect(XENGarray *in, XENGarray *out, XENGarray *vismap, XENGarray *conj,
                    int nchan_sum, int unused);                                         /* corr_subsel_block.py:298 */
int bfXgpuReorder(XENGarray *in, XENGarray *out, XENGarray *baselines, XENGarray *is_conj); /* corr_output_full_block.py:669 */
int bfBeamformInitialize(int gpu, int ninput, int nchan, int ntime, int nbeam, int ntime_blocks); /* beamform_block.py:251 */
int bfBeamformRun(XENGarray *in, XENGarray *out, XENGarray *weights);                  /* beamform_block.py:449 */
int bfBeamformIntegrate(XENGarray *in, XENGarray *out, int ntime_sum);                 /* beamform_sum_beams_block.py:245 */
int bfBeamformIntegrateSingleBeam(XENGarray *in, XENGarray *out, int ntime_sum, int beam_id); /* beamform_sum_single_beam_block.py:114 */

#ifdef __cplusplus
}
#endif
#endif /* XENG_H_ */
