// Host side of the X-engine: process-global context, work-group tiling, C ABI.
// Replaces _bf.bfXgpuInitialize/Kernel/Correlate/GetOrder/SubSelect/Reorder
// (call sites: corr_block.py:251-256,445,317-333; corr_subsel_block.py:298;
//  corr_output_full_block.py:669; verification/xgpu_test.py:76-89).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "xcorr_kernels.h"
#include "slab.h"
#include "xeng_common.h"

namespace xeng {

struct XgpuConfig {
    int nstand = 352, npol = 2, nchan = 96, ntime_gulp = 480, max_gulps = 0;
};

struct XgpuContext {
    bool live = false;
    int gpu = 0;
    XgpuConfig cfg;
    int ncu = 256;
    int ninput = 0, nblk64 = 0, gkt = 0, cap_kt = 0, cap_gulps = 0, kt_stage = 1;
    int ct_pitch = 0;          // LDS row pitch of the transposing corner turn (0: register-only fallback)
    // raw: no corner-turn pass.  The contraction kernel (xcorr_fused_kernel) reads time-major gulps in place and
    // transposes in LDS: asynchronous calls hand over the caller's buffer itself, synchronous calls a raw copy
    // of it in the staging area (except the dump call, which waits for the contraction anyway).
    bool raw = false;
    bool kloop16 = false;      // raw: XENG_KLOOP=16 -- plain and slab launches take the eight-wave 16x16x64 kernel (xcorr_fused16.h)
    FragGroup* fgroups_dev = nullptr;          // fused kernel: fragment-level tile groups (xcorr_tiling.h) ...
    int nfg = 0;
    WorkList work;                             // ... and the persistent work-groups' item lists
    WorkList work_pairs;                       // the same items with neighbouring channels per XCD and round (packet-slab launches: xcorr_tiling.h build_work)
    size_t gulp_bytes = 0;
    const uint8_t* gulp_ptr[XC_MAX_GULPS] = {};
    bool fp6 = false;          // -DXENG_EXPERIMENTS builds, XENG_MFMA=fp6: E3M2 codes + block-scaled FP6 MFMA (exact)
    int ghk = 0;               // fp6: 32-sample half-tiles per gulp; cap_kt then counts 64-sample K steps
    int64_t per_chan = 0, matlen = 0;
    // Two staging areas (raw gulp copies of the synchronous calls, or corner-turned fragments on the two-pass
    // path): filled on `stream` while the contraction of the previous flush reads the other one.
    uint8_t* stash[2] = {nullptr, nullptr};
    size_t stash_bytes = 0;
    int cur = 0;                               // staging area being filled
    WgDesc* descs_dev = nullptr;
    int nwg = 0;
    hipStream_t stream = nullptr;              // raw copies / corner turns (+ H2D of the host-buffer variant)
    // MFMA contractions rotate over nmm streams: consecutive launches are independent (unless they touch the same output
    // span).  Two.  Round 3 re-measured 1 / 2 / 3 / 4 / 6 / 8 (diagnostic builds, XENG_MM_STREAMS): within +-2 %, and the
    // sign depends on the harness -- profiles/ab_step.py reads 3+ streams 2 % faster than two, bench.py's own loop in fresh
    // processes reads two 2 % faster than four (profiles/r03/ab_streams.txt) -- so the smaller number of queues stays.
    static constexpr int NMM = 8;
    hipStream_t stream_mm2[NMM] = {};
    int nmm = 2;
    unsigned long long nlaunch = 0;
    hipStream_t stream_mm = nullptr;           // = stream_mm2[0] (sub-selection, D2H)
    // Last enqueued contraction that writes a buffer (output span or long accumulator): launch number and stream.
    // A later launch or a consumer that names the buffer waits for that launch's own event (a ring of the last NEV
    // launches; older than that: the stream's latest event, which is later in stream order) -- whichever stream it ran
    // on and however many launches went to the streams since.
    struct Writer { unsigned long long seq; int stream; };
    std::map<const void*, Writer> writers;
    static constexpr int NEV = 256;
    // ONE event per launch (more records behind a kernel delay the next dispatch on that hardware queue): everything that
    // has to wait for a launch -- a later launch into the same buffer, a consumer, the refill of a staging area, a dump's
    // caller -- names the launch by its number.  A slot is re-recorded by launch n + NEV only after launch n has completed
    // (the enqueuer waits for it if it has not: back-pressure at NEV launches in flight), so a number whose slot has been
    // taken over needs no wait at all.  (The wait itself happens outside the context lock: wait_for_event_slot.)
    hipEvent_t ev_ring[NEV] = {};              // completion of launch number n: ev_ring[n % NEV]
    const void* ring_buf[NEV][2] = {};         // ... and the buffers it wrote (forgotten as writers when the slot is reused)
    unsigned long long last_seq[NMM] = {};     // number of the latest contraction on each stream (0: none)
    hipEvent_t ev_ct = nullptr;                // staged gulps of the area about to be contracted are complete
    unsigned long long staging_seen = ~0ull;   // staging_stream_ops() at the last such record
    unsigned long long area_seq[2] = {0, 0};   // number of the contraction that reads staging area b (0: none)
    unsigned long long dump_seq[4] = {0, 0, 0, 0};   // numbers of the last dumps
    unsigned long long ndump = 0;
    // integration state
    int nfilled = 0;           // gulps staged since the last flush
    bool acc_started = false;  // out already holds a partial sum of the current integration
    void* acc_out = nullptr;   // where that partial sum lives
    // host staging for xengXgpuCorrelate
    uint8_t* in_dev = nullptr;
    int32_t* out_dev = nullptr;
    unsigned long long* stamps = nullptr;   // diagnostic (XENG_DBG_STAMPS=1)
    // packet slabs as gulps (xengXgpuKernelAsyncSlab): per staging area one descriptor per gulp, written on the staging stream
    GulpDesc* gdesc_dev[2] = {nullptr, nullptr};
    SlabArgs* gargs_dev[2] = {nullptr, nullptr};   // ... and what the scatter of a gulp that turns out irregular needs (slab.h)
    SlabSite slab_site;                     // input counts that are not whole 64-input blocks: every slab is scattered (slab_prepare_kernel, forced)
    SlabIndexSite slab_index;               // (round 5) otherwise: every slab is read in place through its offset table (slab.h)
    SlabIndexJob slab_job;                  // the slabs staged since the last flush
    int slab_forced = 0;                    // gulps of the legacy path since they were last read (counted on the host: all of them are scattered)
    int slab_hint_seen = 0;                 // slab_index.hint_host at the last look ...
    int slab_recent_irr[8] = {}, slab_recent_n[8] = {}, slab_recent_pos = 0;   // ... what it moved by at each of the last eight launches, and their gulps
    unsigned long long slab_tables_until = 0;
    int slab_force_tables = -1;             // XENG_SLAB_TABLES=1 / 0: always / never (tests, A/B); unset: by the hint
    bool slab_mode = false;                 // the gulps staged since the last flush are slabs (no mixing inside one flush)
    EventTimer timer;
};

static std::mutex g_mu;
static XgpuConfig g_cfg;
static XgpuContext g_ctx;
// Bumped whenever registered-but-uncontracted gulps are dropped (Reset, Destroy / re-Initialize): a stamp that waits for the
// launch that would have read them (xeng_common.h, Stamp::xgpu_seq) is void afterwards.
static unsigned long long g_epoch = 1;
static unsigned long long g_ctx_gen = 1;         // bumped when the context is destroyed (its streams have been waited for)

// frees whatever the context holds -- also the partial state of an Initialize that failed half way (x.live false)
static int destroy_locked() {
    XgpuContext& x = g_ctx;
    (void)hipSetDevice(x.gpu);
    if (x.stream) (void)hipStreamSynchronize(x.stream);
    for (int b = 0; b < XgpuContext::NMM; b++) {
        if (x.stream_mm2[b]) (void)hipStreamSynchronize(x.stream_mm2[b]);
    }
    for (int k = 0; k < XgpuContext::NEV; k++)
        if (x.ev_ring[k]) (void)hipEventDestroy(x.ev_ring[k]);
    for (int b = 0; b < 2; b++)
        if (x.stash[b]) (void)hipFree(x.stash[b]);
    if (x.ev_ct) (void)hipEventDestroy(x.ev_ct);
    if (x.descs_dev) (void)hipFree(x.descs_dev);
    if (x.fgroups_dev) (void)hipFree(x.fgroups_dev);
    if (x.work.dev) (void)hipFree(x.work.dev);
    if (x.work_pairs.dev) (void)hipFree(x.work_pairs.dev);
    if (x.in_dev) (void)hipFree(x.in_dev);
    if (x.stamps) (void)hipFree(x.stamps);
    if (x.out_dev) (void)hipFree(x.out_dev);
    for (int b = 0; b < 2; b++)
        if (x.gdesc_dev[b]) (void)hipFree(x.gdesc_dev[b]);
    for (int b = 0; b < 2; b++)
        if (x.gargs_dev[b]) (void)hipFree(x.gargs_dev[b]);
    slab_site_destroy(&x.slab_site);
    slab_index_site_destroy(&x.slab_index);
    x.timer.destroy();
    x = XgpuContext();
    g_epoch++;
    g_ctx_gen++;
    return XENG_STATUS_SUCCESS;
}

template <int ABL>
static void launch_abl(const XcorrParams& p, hipStream_t s) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_mfma_kernel<ABL>), dim3(p.nchan * p.nwg), dim3(256), 0, s, p);
}
static void launch_xcorr(const XcorrParams& p, hipStream_t s, bool raw, int grid_size, bool kloop16) {
    if (raw) {
        const dim3 grid(grid_size);
#ifdef XENG_DIAGNOSTICS
        const int fabl = getenv("XENG_ABLATE") ? atoi(getenv("XENG_ABLATE")) : 0;     // (diagnostic build: read per launch, so one process can alternate)
        if (kloop16 && fabl == 16 && !p.gdesc && !p.acc2_mode) { hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused16_kernel<16>), grid, dim3(512), 0, s, p); return; }
        switch (fabl) {
            case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<1>), grid, dim3(256), 0, s, p); return;
            case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<2>), grid, dim3(256), 0, s, p); return;
            case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<4>), grid, dim3(256), 0, s, p); return;
            case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<8>), grid, dim3(256), 0, s, p); return;
            case 9: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<9>), grid, dim3(256), 0, s, p); return;
            case 15: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<15>), grid, dim3(256), 0, s, p); return;
            case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<16>), grid, dim3(256), 0, s, p); return;
            case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<32>), grid, dim3(256), 0, s, p); return;
            case 48: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<48>), grid, dim3(256), 0, s, p); return;
            case 5: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<5>), grid, dim3(256), 0, s, p); return;
            case 17: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<17>), grid, dim3(256), 0, s, p); return;
            case 31: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<31>), grid, dim3(256), 0, s, p); return;
            case 64: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<64>), grid, dim3(256), 0, s, p); return;
            case 128: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<128>), grid, dim3(256), 0, s, p); return;   // diagonal cells at 3/4 of their MFMAs
            case 256: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<256>), grid, dim3(256), 0, s, p); return;   // offset-binary operands
            case 384: hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<384>), grid, dim3(256), 0, s, p); return;
            default: break;
        }
#endif
        // the eight-wave 16x16x64 kernel (xcorr_fused16.h) for gulps by pointer without a long accumulator; the four-wave kernel otherwise
        // (round 5: and for gulps through their offset tables -- those are laid out for the four-wave kernel's pieces)
        if (kloop16 && !p.acc2_mode && !(p.gdesc && p.by_table)) {
            if (p.gdesc) hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused16_kernel<0, true>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused16_kernel<0>), grid, dim3(512), 0, s, p);
            return;
        }
        if (p.gdesc && p.by_table) {          // gulps by descriptor, every one through its offset table (packet slabs on a lossy link)
            if (p.acc2_mode) hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<0, true, true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<0, false, true, true>), grid, dim3(256), 0, s, p);
            return;
        }
        if (p.gdesc) {          // gulps by descriptor, by strides (packet slabs)
            if (p.acc2_mode) hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<0, true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<0, false, true>), grid, dim3(256), 0, s, p);
            return;
        }
        if (p.acc2_mode) hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<0, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(xcorr_fused_kernel<0>), grid, dim3(256), 0, s, p);
        return;
    }
#ifdef XENG_DIAGNOSTICS
    // timing-only ablations of the K loop (results are wrong): build with -DXENG_DIAGNOSTICS, select with
    // XENG_ABLATE = 1 (no LDS-DMA) | 2 (no unpack) | 4 (no LDS reads) | 8 (no barrier); see profiles/r01/README.md
    const int abl = getenv("XENG_ABLATE") ? atoi(getenv("XENG_ABLATE")) : 0;
    switch (abl) {
        case 1: launch_abl<1>(p, s); return;
        case 2: launch_abl<2>(p, s); return;
        case 4: launch_abl<4>(p, s); return;
        case 8: launch_abl<8>(p, s); return;
        case 9: launch_abl<9>(p, s); return;
        case 15: launch_abl<15>(p, s); return;
        default: break;
    }
#endif
    launch_abl<0>(p, s);
}

// completion event of launch number `seq`, or null when that launch is known to have completed (its slot was taken over)
static hipEvent_t launch_event(unsigned long long seq) {
    XgpuContext& x = g_ctx;
    if (seq == 0 || x.nlaunch - seq >= (unsigned long long)XgpuContext::NEV) return nullptr;
    return x.ev_ring[seq % XgpuContext::NEV];
}

// under g_mu: make stream `s` (contraction stream `self`, or -1 for a foreign stream) wait for the last enqueued writer of `buf`
static int order_after_writer(hipStream_t s, int self, const void* buf) {
    XgpuContext& x = g_ctx;
    auto it = x.writers.find(buf);
    if (it == x.writers.end() || it->second.stream == self) return XENG_STATUS_SUCCESS;   // (same stream: stream order)
    if (hipEvent_t ev = launch_event(it->second.seq)) XENG_HIP(hipStreamWaitEvent(s, ev, 0));
    return XENG_STATUS_SUCCESS;
}

// contracts the staged gulps into `out`; caller holds g_mu
// acc / acc_mode: long accumulator that the dump also assigns (1) or adds (2) its values to (fused path only)
static int flush_locked(void* out, bool dump, void* acc = nullptr, int acc_mode = 0) {
    XgpuContext& x = g_ctx;
    if (x.nfilled == 0) return XENG_STATUS_SUCCESS;
    if (!dump) { acc = nullptr; acc_mode = 0; }       // partial sums of an integration never reach the long accumulator
    if (x.acc_started && x.acc_out != out)
        XENG_FAIL(XENG_STATUS_INVALID_STATE,
                  "xgpu: output buffer changed inside one integration (partial sums live in %p, got %p)",
                  x.acc_out, out);
    int nkt = x.nfilled * x.gkt;
#ifdef XENG_EXPERIMENTS
    if (x.fp6) {
        const int nhk = x.nfilled * x.ghk;
        if (nhk & 1)   // the last 64-sample K step is half filled: its second half must hold the value 0
            hipLaunchKernelGGL(fp6_zero_half_kernel, dim3(x.cfg.nchan * x.nblk64), dim3(64), 0, x.stream,
                               x.stash[x.cur], x.nblk64, x.cap_kt, nhk >> 1);
        stream_tick(STREAM_XGPU);
        nkt = (nhk + 1) >> 1;
    }
#endif
    const int rem = (x.fp6 || x.raw) ? 0 : nkt % x.kt_stage;
    if (rem) {  // zero-fill the K padding of every (channel, block) row of the stash
        const int padk = x.kt_stage - rem;
        XENG_HIP(hipMemset2DAsync(x.stash[x.cur] + (size_t)nkt * KT_BYTES, (size_t)x.cap_kt * KT_BYTES, 0,
                                  (size_t)padk * KT_BYTES, (size_t)x.cfg.nchan * x.nblk64, x.stream));
        stream_tick(STREAM_XGPU);
        nkt += padk;
    }
    const bool slab_tables = x.slab_mode && x.slab_index.tab32 != nullptr;
    bool slab_by_table = false;
    if (slab_tables) {
        // packet slabs: index + offset table of every gulp, descriptors (slab.h); zero-fill + scatter only of gulps that carry packets of
        // another geometry -- five short launches per integration on the staging stream, beside the previous contraction
        x.slab_job.ngulp = x.nfilled;
        // By strides while the link is clean or nearly so (round 4's kernel: nothing extra in its K loop; an irregular gulp costs a
        // zero-fill + scatter, about 15 % of a gulp), through the tables (3-4 % on every gulp, irregular or not) once more than a quarter
        // of the gulps of the last eight launches were not regular.  The device counts them in pinned memory, read here without a
        // wait: a hint that lags by a launch or two -- either kernel is exact on any slab.
        {
            const int seen = *(volatile int*)x.slab_index.hint_host;
            x.slab_recent_irr[x.slab_recent_pos] = seen - x.slab_hint_seen;
            x.slab_recent_n[x.slab_recent_pos] = x.nfilled;
            x.slab_recent_pos = (x.slab_recent_pos + 1) & 7;
            x.slab_hint_seen = seen;
            int irr = 0, n = 0;
            for (int k = 0; k < 8; k++) { irr += x.slab_recent_irr[k]; n += x.slab_recent_n[k]; }
            if (4 * irr > n) x.slab_tables_until = x.nlaunch + 16;      // (a little hysteresis: the counts lag by a launch or two)
            slab_by_table = x.slab_force_tables > 0 || (x.slab_force_tables < 0 && x.nlaunch < x.slab_tables_until);
        }
        if (int rcs = slab_index_enqueue(x.stream, x.slab_index, x.cur, x.slab_job, slab_by_table, x.gdesc_dev[x.cur], x.gargs_dev[x.cur])) return rcs;
        staging_stream_touched();
    } else if (x.slab_mode) {
        // (inputs that are not whole 64-input blocks: every gulp was forced to its scratch copy by slab_prepare_kernel; plain launch from there)
        if (int rcs = slab_fallback_enqueue(x.stream, x.gdesc_dev[x.cur], x.gargs_dev[x.cur], x.nfilled)) return rcs;
        staging_stream_touched();
    }
    XcorrParams p;
    p.stash = x.stash[x.cur]; p.out = (int32_t*)out; p.descs = x.descs_dev;
    p.nwg = x.raw ? x.nfg : x.nwg; p.nchan = x.cfg.nchan; p.nblk64 = x.nblk64; p.cap_kt = x.cap_kt; p.nkt = nkt;
    p.nstand = x.cfg.nstand; p.per_chan = x.per_chan; p.matlen = x.matlen;
    p.accumulate = x.acc_started ? 1 : 0;
    p.stamps = x.stamps;
    p.spg = x.raw ? x.cfg.ntime_gulp / (XC_KT * 32) : 0;
    p.ninput = x.ninput;
    p.fgroups = x.fgroups_dev; p.work = x.work.dev; p.maxi = x.work.maxi; p.nstage = nkt / XC_KT;
    p.acc2 = (int32_t*)acc; p.acc2_mode = acc ? acc_mode : 0;
    p.gdesc = slab_tables ? x.gdesc_dev[x.cur] : nullptr;
    p.by_table = slab_by_table ? 1 : 0;
    if (slab_tables && !diag_env("XENG_SLAB_PLAIN_ORDER")) { p.work = x.work_pairs.dev; p.maxi = x.work_pairs.maxi; }    // (diagnostic builds: A/B switch)
    for (int g = 0; g < XC_MAX_GULPS; g++) p.gulps[g] = g < x.nfilled ? x.gulp_ptr[g] : nullptr;
    // the contraction starts when this area's corner turns are done and runs beside the next area's
    const int si = (int)(x.nlaunch++ % x.nmm);
    hipStream_t smm = x.stream_mm2[si];
    if (const unsigned long long ops = staging_stream_ops(); ops != x.staging_seen) {
        XENG_HIP(hipEventRecord(x.ev_ct, x.stream));       // (only when something was put on the staging stream since the
        XENG_HIP(hipStreamWaitEvent(smm, x.ev_ct, 0));     // last contraction: gulps read in place need no barrier packet)
        x.staging_seen = ops;
    }
    // contractions that touch the same output (partial sums of one integration, or a caller that
    // reuses one buffer for consecutive integrations) stay ordered; independent ones may overlap
    const unsigned long long seq = x.nlaunch;          // this launch's number (>= 1)
    {
        // this launch takes over the event slot of launch seq - NEV, which must have completed (see XgpuContext): normally
        // long ago; NEV launches deep in an unsynchronised queue the enqueuer waits here
        const int k = (int)(seq % XgpuContext::NEV);
        if (seq > (unsigned long long)XgpuContext::NEV) {
            if (hipEventQuery(x.ev_ring[k]) != hipSuccess) {
                (void)hipGetLastError();          // (hipErrorNotReady is not an error)
                XENG_HIP(hipEventSynchronize(x.ev_ring[k]));     // (not reached through the C ABI: its callers have waited, below)
            }
            for (const void* buf : x.ring_buf[k]) {
                auto it = buf ? x.writers.find(buf) : x.writers.end();
                if (it != x.writers.end() && it->second.seq == seq - XgpuContext::NEV) x.writers.erase(it);
            }
        }
        x.ring_buf[k][0] = out; x.ring_buf[k][1] = acc;
    }
    for (const void* buf : {(const void*)out, (const void*)acc}) {
        if (!buf) continue;
        int rc = order_after_writer(smm, si, buf);
        if (rc) return rc;
        x.writers[buf] = XgpuContext::Writer{seq, si};
    }
    int slot = x.timer.begin(smm, 1);
#ifdef XENG_EXPERIMENTS
    if (x.fp6) hipLaunchKernelGGL(xcorr_fp6_kernel, dim3(p.nchan * p.nwg), dim3(256), 0, smm, p);
    else
#endif
    launch_xcorr(p, smm, x.raw, fused_grid(x.cfg.nchan, x.nfg, x.ncu), x.kloop16);
    x.timer.end(smm, slot);
    XENG_HIP(hipGetLastError());
    XENG_HIP(hipEventRecord(x.ev_ring[seq % XgpuContext::NEV], smm));
    x.area_seq[x.cur] = seq;
    x.last_seq[si] = seq;
    if (dump) {
        x.dump_seq[x.ndump & 3] = seq;
        x.ndump++;
    }
    x.cur ^= 1;
    x.nfilled = 0;
    x.slab_mode = false;
    x.acc_started = !dump;
    x.acc_out = dump ? nullptr : out;
    return XENG_STATUS_SUCCESS;
}

// What a call still has to wait for after it has released the context lock: the staging stream (the caller's
// input has been copied / corner-turned) and, for a dump, the completion event of that dump.
struct PendingWait {
    int gpu = 0;
    hipStream_t staging = nullptr;
    hipEvent_t dump = nullptr;
};

// Waits happen OUTSIDE g_mu (the other blocks' calls -- SubSelect, Packetize, GetOrder -- must not stall for a whole
// contraction: lwa352-pipeline.py:232-262 runs them on their own threads); the lock is retaken only to fold finished
// timing events into the profile.
static int wait_unlocked(const PendingWait& w) {
    XENG_HIP(hipSetDevice(w.gpu));
    if (w.staging) XENG_HIP(hipStreamSynchronize(w.staging));
    if (w.dump) XENG_HIP(hipEventSynchronize(w.dump));
    return XENG_STATUS_SUCCESS;
}

static int kernel_locked(const void* in_dev, void* out_dev, int doDump, bool sync, PendingWait* pw, void* acc = nullptr,
                         int acc_mode = 0) {
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized (call xengXgpuInitialize)");
    if (!in_dev || !out_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: null buffer");
    if (acc) {
        if (!x.raw)
            XENG_FAIL(XENG_STATUS_UNSUPPORTED, "xgpu: the fused long accumulation needs the default contraction kernel (use xengMapAddI32)");
        if (acc_mode != 1 && acc_mode != 2) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: acc_mode must be 1 (assign) or 2 (add)");
        if (((uintptr_t)acc & 15) || acc == out_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: accumulator must be 16-byte aligned and distinct from out");
    }
    if (((uintptr_t)out_dev & 15) || ((uintptr_t)in_dev & (x.raw ? 15 : 3)))
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: out must be 16-byte and in %d-byte aligned", x.raw ? 16 : 4);
    XENG_HIP(hipSetDevice(x.gpu));
    if (x.nfilled > 0 && x.slab_mode)
        XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: packet slabs and plain gulps cannot be mixed between two flushes");
    if (x.nfilled == 0)                          // this staging area may still be read by an earlier contraction
        if (hipEvent_t ev = launch_event(x.area_seq[x.cur])) XENG_HIP(hipStreamWaitEvent(x.stream, ev, 0));
    uint8_t* const stash = x.stash[x.cur];
    int slot = -1;
    if (x.raw) {
        if (sync && !doDump) {   // the caller may recycle in_dev on return: keep a raw copy (a synchronous dump
                                 // waits for the contraction before it returns, so its own gulp is read in place)
            uint8_t* copy = stash + (size_t)x.nfilled * x.gulp_bytes;
            slot = x.timer.begin(x.stream, 0);
            if (x.gulp_bytes % 16 == 0) {     // (in_dev and the staging area are 16-byte aligned on this path)
                const size_t n16 = x.gulp_bytes / 16;
                const unsigned blocks = (unsigned)std::min<size_t>(2048, (n16 + 1023) / 1024);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(gulp_copy_kernel<4>), dim3(blocks), dim3(256), 0, x.stream, (v4i*)copy, (const v4i*)in_dev, n16);
                XENG_HIP(hipGetLastError());
            } else {
                XENG_HIP(hipMemcpyAsync(copy, in_dev, x.gulp_bytes, hipMemcpyDeviceToDevice, x.stream));
            }
            x.gulp_ptr[x.nfilled] = copy;
        } else {
            x.gulp_ptr[x.nfilled] = (const uint8_t*)in_dev;
        }
#ifdef XENG_EXPERIMENTS
    } else if (x.fp6) {
        slot = x.timer.begin(x.stream, 0);
        const size_t l6 = (((size_t)32 * x.ninput + 1023) & ~(size_t)1023) + (size_t)x.nblk64 * 2 * F6_FRAG;
        hipLaunchKernelGGL(corner_turn_fp6_kernel, dim3(x.cfg.nchan, x.ghk), dim3(192), l6, x.stream,
                           (const uint8_t*)in_dev, stash, x.cfg.ntime_gulp, x.cfg.nchan, x.ninput, x.nblk64,
                           x.cap_kt, x.nfilled * x.ghk);
#endif
    } else if (x.ct_pitch > 0) {
        slot = x.timer.begin(x.stream, 0);
        const size_t l8 = (((size_t)16 * x.ct_pitch + 1023) & ~(size_t)1023);
        hipLaunchKernelGGL(corner_turn_tr8_kernel, dim3(x.cfg.nchan, x.gkt, 2), dim3(256), l8, x.stream,
                           (const uint8_t*)in_dev, stash, x.cfg.ntime_gulp, x.cfg.nchan, x.ninput, x.nblk64,
                           x.cap_kt, x.nfilled * x.gkt, x.ct_pitch);
    } else {
        slot = x.timer.begin(x.stream, 0);
        const int ct_items = 2 * x.nblk64 * 16;                   // (input quad, k-half) work items per (K-tile, channel)
        const int ct_threads = std::min(1024, ((ct_items + 63) / 64) * 64);
        hipLaunchKernelGGL(corner_turn_kernel, dim3(x.gkt, x.cfg.nchan), dim3(ct_threads), 0, x.stream,
                           (const uint8_t*)in_dev, stash, x.cfg.ntime_gulp, x.cfg.nchan, x.ninput, x.nblk64,
                           x.cap_kt, x.nfilled * x.gkt);
    }
    x.timer.end(x.stream, slot);
    XENG_HIP(hipGetLastError());
    if (!x.raw || (sync && !doDump)) staging_stream_touched();     // a copy or a corner turn went to the staging stream
    x.nfilled++;
    if (doDump || x.nfilled == x.cap_gulps) {
        int rc = flush_locked(out_dev, doDump != 0, acc, acc_mode);
        if (rc) return rc;
    } else if (x.acc_started && x.acc_out != out_dev) {
        XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: output buffer changed inside one integration");
    }
    if (sync && pw) {
        // input consumed = its copy / corner turn is done; on a dump the output must be complete too
        pw->gpu = x.gpu;
        pw->staging = x.stream;
        if (doDump) pw->dump = launch_event(x.dump_seq[(x.ndump - 1) & 3]);      // contractions that touch one span are ordered
    }
    return XENG_STATUS_SUCCESS;
}

// ---- the X-engine's part of a stamp (xeng_common.h): gulps handed over with xengXgpuKernelAsync are read by a launch that
// does not exist yet; whoever stamps a buffer meanwhile must also wait for that launch
void xgpu_pending_launch(unsigned long long* seq, unsigned long long* epoch, unsigned long long* nlaunch, unsigned long long* ctx) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    *seq = (x.live && x.nfilled > 0) ? x.nlaunch + 1 : 0;
    *epoch = g_epoch;
    *nlaunch = x.live ? x.nlaunch : 0;
    *ctx = g_ctx_gen;
}

// have all contractions up to launch number `upto` of context generation `ctx` completed?  Every launch owns a completion
// event (ev_ring) and the streams take the launches in rotation, so the last nmm launches cover every stream: nothing is recorded.
unsigned long long xgpu_last_writer(const void* buf) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live || !buf) return 0;
    auto it = x.writers.find(buf);
    return it == x.writers.end() ? 0 : it->second.seq;
}

int xgpu_launches_poll(unsigned long long upto, unsigned long long ctx, bool* done, hipEvent_t* ev, bool exact) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    *done = true;
    if (ev) *ev = nullptr;
    if (upto == 0 || !x.live || ctx != g_ctx_gen) return XENG_STATUS_SUCCESS;      // (a destroyed context has waited for its streams)
    for (int k = 0; k < (exact ? 1 : x.nmm) && (unsigned long long)k < upto; k++) {
        if (hipEvent_t e = launch_event(upto - k)) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                *done = false;
                if (ev && !*ev) *ev = e;
            } else if (q != hipSuccess) {
                XENG_HIP(q);
            }
        }
    }
    return XENG_STATUS_SUCCESS;
}

int xgpu_pending_poll(unsigned long long seq, unsigned long long epoch, bool* done, bool* launched, hipEvent_t* ev, int* gpu) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    *done = true;
    *launched = true;
    if (ev) *ev = nullptr;
    if (seq == 0 || !x.live || epoch != g_epoch) return XENG_STATUS_SUCCESS;     // (dropped gulps: Reset / Destroy have waited for the streams)
    if (x.nlaunch < seq) {
        *done = false;
        *launched = false;
        return XENG_STATUS_SUCCESS;
    }
    if (hipEvent_t e = launch_event(seq)) {
        const hipError_t q = hipEventQuery(e);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            *done = false;
            if (ev) *ev = e;
            if (gpu) *gpu = x.gpu;
        } else if (q != hipSuccess) {
            XENG_HIP(q);
        }
    }
    return XENG_STATUS_SUCCESS;
}

}  // namespace xeng

using namespace xeng;

extern "C" {

int xengXgpuConfigure(int nstand, int npol, int nchan, int ntime_gulp, int max_gulps_per_flush) {
    if (npol != 2) XENG_FAIL(XENG_STATUS_UNSUPPORTED, "xgpu: npol must be 2 (got %d)", npol);
    if (nstand <= 0 || nstand % 4) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: nstand must be a positive multiple of 4 (got %d)", nstand);
    if (nstand * npol > 255 * 64) XENG_FAIL(XENG_STATUS_UNSUPPORTED, "xgpu: too many inputs (%d)", nstand * npol);
    if (nchan <= 0 || ntime_gulp <= 0 || max_gulps_per_flush < 0)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: bad sizes nchan=%d ntime_gulp=%d max_gulps=%d", nchan, ntime_gulp, max_gulps_per_flush);
    std::lock_guard<std::mutex> lk(g_mu);
    g_cfg.nstand = nstand; g_cfg.npol = npol; g_cfg.nchan = nchan; g_cfg.ntime_gulp = ntime_gulp;
    g_cfg.max_gulps = max_gulps_per_flush;
    return XENG_STATUS_SUCCESS;
}

static int initialize_locked(int gpu);

int xengXgpuInitialize(int gpu) {
    std::lock_guard<std::mutex> lk(g_mu);
    destroy_locked();
    const int rc = initialize_locked(gpu);
    if (rc) destroy_locked();          // (keeps the error message: destroy does not touch it)
    return rc;
}

static int initialize_locked(int gpu) {
    XgpuContext& x = g_ctx;
    x.cfg = g_cfg;
    x.gpu = gpu < 0 ? 0 : gpu;
    XENG_HIP(hipSetDevice(x.gpu));
    {
        int ncu = 0;
        XENG_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, x.gpu));
        const char* e = diag_env("XENG_GRID");          // experiment: persistent grid size
        x.ncu = e ? atoi(e) : ncu;
        if (x.ncu < 1) x.ncu = 1;
    }
    x.ninput = x.cfg.nstand * x.cfg.npol;
    x.nblk64 = (x.ninput + 63) / 64;
    x.gkt = (x.cfg.ntime_gulp + 31) / 32;
    x.kt_stage = XC_KT;   // K is zero-padded to a whole number of stages at flush time
    // staging depth: default ~4800 samples (two 2400-sample integrations' worth never needed at once;
    // the X-engine flushes early when full).  K per launch must stay <= 65535 samples (int32 bound).
    int cap = x.cfg.max_gulps > 0 ? x.cfg.max_gulps : std::max(1, 4800 / (x.gkt * 32));
    cap = std::min(cap, std::max(1, 65535 / (x.gkt * 32)));
    {
        // default: fused corner turn whenever the shape allows it (whole 16-byte input chunks, gulps made
        // of whole 96-sample stages); XENG_RAW=0 keeps the two-pass path
        const char* r = getenv("XENG_RAW");
#ifdef XENG_EXPERIMENTS
        const char* m = getenv("XENG_MFMA");
#else
        const char* m = nullptr;
#endif
        x.gulp_bytes = (size_t)x.cfg.ntime_gulp * x.cfg.nchan * x.ninput;
        x.raw = !(m && !strcmp(m, "fp6")) && !(r && !strcmp(r, "0")) && x.ninput % 16 == 0 &&
                x.cfg.ntime_gulp % (XC_KT * 32) == 0 && x.gulp_bytes + (size_t)16 * x.cfg.nchan * x.ninput < (1ull << 32) &&
                // (the kernel's per-lane LDS-DMA offsets subtract the instruction's immediate, up to 3072 bytes, from
                // n * 8 rows: they stay non-negative -- the VGPR offset is unsigned -- only for rows of at least 128 bytes)
                (size_t)x.cfg.nchan * x.ninput >= 128;
        if (x.raw) cap = std::min(cap, XC_MAX_GULPS);   // gulp pointers travel in the kernel arguments
        // XENG_KLOOP=16: plain and slab launches take the eight-wave 16x16x64 kernel (xcorr_fused16.h).  Exact and fully tested, but not the
        // default: against the four-wave kernel its streaming step measured -3.0 % and -3.6 % on two boxes, +1.7 .. +7.8 % on three others
        // (one launch alone: equal or 1-3 % faster everywhere; what differs is how consecutive persistent launches share the chip:
        // profiles/r05/ab_kloop16_*.txt)
        const char* kl = getenv("XENG_KLOOP");
        x.kloop16 = x.raw && kl && !strcmp(kl, "16");
        // XENG_SLAB_TABLES=1 / 0: packet slabs always / never through their offset tables (default: by strides until a gulp was not regular)
        const char* st = getenv("XENG_SLAB_TABLES");
        x.slab_force_tables = st ? (strcmp(st, "0") ? 1 : 0) : -1;
        x.slab_hint_seen = 0; x.slab_recent_pos = 0; x.slab_tables_until = 0;
        for (int k = 0; k < 8; k++) x.slab_recent_irr[k] = x.slab_recent_n[k] = 0;
    }
    x.cap_gulps = cap;
    x.cap_kt = ((cap * x.gkt + x.kt_stage - 1) / x.kt_stage) * x.kt_stage;
    x.per_chan = (int64_t)(x.cfg.nstand / 2 + 1) * (x.cfg.nstand / 4) * x.cfg.npol * x.cfg.npol * 4;
    x.matlen = x.per_chan * x.cfg.nchan;
    x.stash_bytes = (size_t)x.cfg.nchan * x.nblk64 * x.cap_kt * KT_BYTES;
    if (x.ninput % 16 == 0) {
        // LDS row pitch: >= ninput, a multiple of 16 bytes, and 8 mod 64 dwords (conflict-free transposing reads)
        int pitch = x.ninput;
        while (((pitch / 4) & 63) != 8) pitch += 16;
        if ((size_t)16 * pitch + 1024 <= 64 * 1024) x.ct_pitch = pitch;
    }
#ifdef XENG_EXPERIMENTS
    {
        const char* m = getenv("XENG_MFMA");
        const size_t l6 = (((size_t)32 * x.ninput + 1023) & ~(size_t)1023) + (size_t)x.nblk64 * 2 * F6_FRAG;
        x.fp6 = m && !strcmp(m, "fp6") && x.ninput % 16 == 0 && l6 <= 64 * 1024;
        if (x.fp6) {
            x.ghk = (x.cfg.ntime_gulp + 31) / 32;
            x.cap_kt = (cap * x.ghk + 1) / 2;                       // 64-sample K steps
            x.stash_bytes = (size_t)x.cfg.nchan * x.nblk64 * x.cap_kt * F6_KT_BYTES;
        }
    }
#endif
    if (x.raw) x.stash_bytes = (size_t)cap * x.gulp_bytes;   // raw copies of synchronously handed gulps

    for (int b = 0; b < 2; b++) {
        XENG_HIP(hipMalloc((void**)&x.stash[b], x.stash_bytes));
        XENG_HIP(hip_memset_now(x.stash[b], 0, x.stash_bytes));
    }
    XENG_HIP(hipEventCreateWithFlags(&x.ev_ct, hipEventDisableTiming));
    if (x.raw) {
        for (int b = 0; b < 2; b++) {
            XENG_HIP(hipMalloc((void**)&x.gdesc_dev[b], XC_MAX_GULPS * sizeof(GulpDesc)));
            XENG_HIP(hip_memset_now(x.gdesc_dev[b], 0, XC_MAX_GULPS * sizeof(GulpDesc)));
            XENG_HIP(hipMalloc((void**)&x.gargs_dev[b], XC_MAX_GULPS * sizeof(SlabArgs)));
            XENG_HIP(hip_memset_now(x.gargs_dev[b], 0, XC_MAX_GULPS * sizeof(SlabArgs)));
        }
        if (int rcs = slab_site_create(&x.slab_site)) return rcs;
    }
    std::vector<WgDesc> descs = build_wg_descs(x.nblk64);
    x.nwg = (int)descs.size();
    if (x.raw) {
        // fragment-level tile groups and the persistent work-groups' item lists (the same for every K length)
        const bool tiles64 = getenv("XENG_TILING") && !strcmp(getenv("XENG_TILING"), "64");        // A/B switch: the 64x64 tiling
        const bool plain_order = diag_env("XENG_ITEM_ORDER") && !strcmp(diag_env("XENG_ITEM_ORDER"), "plain");   // A/B switch
        const std::vector<FragGroup> groups = tiles64 ? frag_groups_from_tiles(x.nblk64) : build_frag_groups(x.nblk64);
        if (check_frag_groups(groups, x.nblk64) != -1) XENG_FAIL(XENG_STATUS_DEVICE_ERROR, "xgpu: the tiling of %d blocks is not an exact cover", x.nblk64);
        x.nfg = (int)groups.size();
        XENG_HIP(hipMalloc((void**)&x.fgroups_dev, groups.size() * sizeof(FragGroup)));
        XENG_HIP(hipMemcpy(x.fgroups_dev, groups.data(), groups.size() * sizeof(FragGroup), hipMemcpyHostToDevice));
        const std::vector<uint64_t> masks = group_block_masks(groups);
        x.work = build_work(fused_grid(x.cfg.nchan, x.nfg, x.ncu), x.cfg.nchan, x.nfg, plain_order ? nullptr : &masks);
        XENG_HIP(hipMalloc((void**)&x.work.dev, x.work.entries.size() * sizeof(WorkEntry)));
        XENG_HIP(hipMemcpy(x.work.dev, x.work.entries.data(), x.work.entries.size() * sizeof(WorkEntry), hipMemcpyHostToDevice));
        x.work_pairs = build_work(fused_grid(x.cfg.nchan, x.nfg, x.ncu), x.cfg.nchan, x.nfg, plain_order ? nullptr : &masks, true);
        XENG_HIP(hipMalloc((void**)&x.work_pairs.dev, x.work_pairs.entries.size() * sizeof(WorkEntry)));
        XENG_HIP(hipMemcpy(x.work_pairs.dev, x.work_pairs.entries.data(), x.work_pairs.entries.size() * sizeof(WorkEntry), hipMemcpyHostToDevice));
    }
    XENG_HIP(hipMalloc((void**)&x.descs_dev, descs.size() * sizeof(WgDesc)));
    XENG_HIP(hipMemcpy(x.descs_dev, descs.data(), descs.size() * sizeof(WgDesc), hipMemcpyHostToDevice));
    if (const char* o = diag_env("XENG_STREAM_ORDER")) {           // experiment: which library streams exist before the X-engine's
        hipStream_t tmp;
        for (const char* c = o; *c; c++) {
            const StreamId id = *c == 'c' ? STREAM_COPY : *c == 'm' ? STREAM_MAP : *c == 'b' ? STREAM_BEAM : *c == 'k' ? STREAM_CONSUMER : STREAM_XGPU;
            int rc0 = get_stream(id, &tmp);
            if (rc0) return rc0;
        }
    }
    int rc = get_stream(STREAM_XGPU, &x.stream);
    if (rc) return rc;
    if (const char* e = diag_env("XENG_MM_STREAMS")) x.nmm = std::max(1, std::min(XgpuContext::NMM, atoi(e)));   // experiment
    for (int t = 0; t < x.nmm; t++) {      // (only the streams in use: every stream takes a share of a hardware queue)
        rc = get_stream(mm_stream_id(t), &x.stream_mm2[t]);
        if (rc) return rc;
    }
    for (int k = 0; k < XgpuContext::NEV; k++) XENG_HIP(hipEventCreateWithFlags(&x.ev_ring[k], hipEventDisableTiming));
    x.stream_mm = x.stream_mm2[0];
    if (diag_env("XENG_DBG_STAMPS")) {
        const int ng = std::max(x.nwg, x.nfg);
        XENG_HIP(hipMalloc((void**)&x.stamps, (size_t)x.cfg.nchan * ng * 4 * 8 * sizeof(unsigned long long)));
        XENG_HIP(hip_memset_now(x.stamps, 0, (size_t)x.cfg.nchan * ng * 4 * 8 * sizeof(unsigned long long)));
    }
    // (the uploads above went through the null stream; the context's streams do not wait for it: everything is in place
    // before the first launch can be enqueued -- see hip_memset_now, xeng_common.h)
    XENG_HIP(hipStreamSynchronize(nullptr));
    x.live = true;
    return XENG_STATUS_SUCCESS;
}

int xengXgpuDestroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return destroy_locked();
}

static void drain_timer() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx.live) g_ctx.timer.drain();
}

// Back-pressure of the event ring, outside the context lock: the next launch re-records the slot of launch n - NEV, which
// must have completed first.  Normally it has, long ago; a caller that is NEV launches ahead of the GPU waits here, and
// the other blocks' calls (SubSelect, Packetize, ...) go on meanwhile.
static int wait_for_event_slot(bool may_block = true) {
    for (;;) {
        hipEvent_t ev = nullptr;
        int gpu = 0;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            XgpuContext& x = g_ctx;
            if (!x.live) return XENG_STATUS_SUCCESS;             // (the call itself reports it)
            const unsigned long long next = x.nlaunch + 1;
            if (next <= (unsigned long long)XgpuContext::NEV) return XENG_STATUS_SUCCESS;
            ev = x.ev_ring[next % XgpuContext::NEV];
            gpu = x.gpu;
            if (hipEventQuery(ev) == hipSuccess) return XENG_STATUS_SUCCESS;
            (void)hipGetLastError();
        }
        if (!may_block) XENG_FAIL(XENG_STATUS_WOULD_BLOCK, "xgpu: %d launches in flight (xengXgpuWaitLaunchSlot, then retry)", XgpuContext::NEV);
        XENG_HIP(hipSetDevice(gpu));
        XENG_HIP(hipEventSynchronize(ev));
    }
}

int xengXgpuKernel(const void* in_dev, void* out_dev, int doDump) {
    PendingWait pw;
    int rc0 = wait_for_event_slot();
    if (rc0) return rc0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        int rc = kernel_locked(in_dev, out_dev, doDump, true, &pw);
        if (rc) return rc;
    }
    int rc = wait_unlocked(pw);
    if (rc) return rc;
    if (doDump) drain_timer();
    return XENG_STATUS_SUCCESS;
}

int xengXgpuKernelAsync(const void* in_dev, void* out_dev, int doDump) {
    int rc0 = wait_for_event_slot();
    if (rc0) return rc0;
    std::lock_guard<std::mutex> lk(g_mu);
    return kernel_locked(in_dev, out_dev, doDump, false, nullptr);
}

int xengXgpuKernelAsyncAcc(const void* in_dev, void* out_dev, int doDump, void* acc_dev, int acc_mode) {
    int rc0 = wait_for_event_slot();
    if (rc0) return rc0;
    std::lock_guard<std::mutex> lk(g_mu);
    return kernel_locked(in_dev, out_dev, doDump, false, nullptr, acc_dev, acc_dev ? acc_mode : 0);
}

int xengXgpuTryKernelAsyncAcc(const void* in_dev, void* out_dev, int doDump, void* acc_dev, int acc_mode) {
    int rc0 = wait_for_event_slot(false);
    if (rc0) return rc0;
    std::lock_guard<std::mutex> lk(g_mu);
    return kernel_locked(in_dev, out_dev, doDump, false, nullptr, acc_dev, acc_dev ? acc_mode : 0);
}

int xengXgpuWaitLaunchSlot(void) { return wait_for_event_slot(true); }

// A gulp handed over as the slab of packets it arrived in (slab.h).  Nothing is launched here: the slabs of an integration are indexed
// on the device when the integration is flushed (slab_index_enqueue) and read where they lie -- by strides, or through their offset
// tables -- or, holding packets of another geometry, scattered into the library's staging area and read from there.
static int kernel_slab(const void* packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int chan0_pipeline, void* out_dev,
                       int doDump, void* acc_dev, int acc_mode, bool may_block) {
    int rc0 = wait_for_event_slot(may_block);
    if (rc0) return rc0;
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized (call xengXgpuInitialize)");
    if (!x.raw) XENG_FAIL(XENG_STATUS_UNSUPPORTED, "xgpu: packet slabs need the default contraction kernel (unpack with xengSnap2UnpackAsync instead)");
    if (!packets_dev || !out_dev || npkt < 0 || pkt_stride < 32 || pkt_stride > 0x7FFFFFFFu)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: bad slab (npkt %d, stride %zu)", npkt, pkt_stride);
    if (acc_dev) {
        if (acc_mode != 1 && acc_mode != 2) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: acc_mode must be 1 (assign) or 2 (add)");
        if (((uintptr_t)acc_dev & 15) || acc_dev == out_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: accumulator must be 16-byte aligned and distinct from out");
    }
    if ((uintptr_t)out_dev & 15) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: out must be 16-byte aligned");
    if (x.nfilled > 0 && !x.slab_mode)
        XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: packet slabs and plain gulps cannot be mixed between two flushes");
    if (x.acc_started && x.acc_out != out_dev) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: output buffer changed inside one integration");
    XENG_HIP(hipSetDevice(x.gpu));
    if (x.nfilled == 0)                          // this staging area (scratch gulps, descriptors) may still be read by an earlier contraction
        if (hipEvent_t ev = launch_event(x.area_seq[x.cur])) XENG_HIP(hipStreamWaitEvent(x.stream, ev, 0));
    const int k = x.nfilled;
    SlabArgs a;
    a.pkts = (const uint8_t*)packets_dev; a.npkt = npkt; a.stride = (uint32_t)pkt_stride; a.seq0 = seq0;
    a.ntime = x.cfg.ntime_gulp; a.chan0 = chan0_pipeline; a.nchan = x.cfg.nchan; a.ninput = x.ninput; a.nblk = x.ninput / 64;
    uint8_t* scratch = x.stash[x.cur] + (size_t)k * x.gulp_bytes;
    if (x.ninput % 64 == 0) {
        // read in place through an offset table, whatever order the packets are in: nothing is launched per call -- the index and
        // table passes of all gulps of the integration go onto the staging stream with the flush (slab_index_enqueue)
        if (!x.slab_index.tab32)
            if (int rcs = slab_index_site_create(&x.slab_index, x.cfg.ntime_gulp, x.cfg.nchan, x.ninput)) return rcs;
        x.slab_job.a[k] = a; x.slab_job.scratch[k] = scratch; x.slab_job.force[k] = slab_indexable(a) ? 0 : 1;
        x.gulp_ptr[k] = nullptr;
    } else {
        // (the contraction reads whole 64-input blocks through a table; other input counts: scattered, then a plain launch)
        const bool maybe = false;
        if (int rcs = slab_prepare_enqueue(x.stream, x.slab_site, &a, &maybe, 1, x.gdesc_dev[x.cur] + k, x.gargs_dev[x.cur] + k, &scratch, false)) return rcs;
        XENG_HIP(hipGetLastError());
        staging_stream_touched();
        x.slab_forced++;
        x.gulp_ptr[k] = scratch;
    }
    x.slab_mode = true;
    x.nfilled++;
    if (doDump || x.nfilled == x.cap_gulps) return flush_locked(out_dev, doDump != 0, acc_dev, acc_dev ? acc_mode : 0);
    return XENG_STATUS_SUCCESS;
}

int xengXgpuKernelAsyncSlab(const void* packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int chan0_pipeline, void* out_dev,
                            int doDump, void* acc_dev, int acc_mode) {
    return kernel_slab(packets_dev, npkt, pkt_stride, seq0, chan0_pipeline, out_dev, doDump, acc_dev, acc_mode, true);
}

// (never waits: XENG_STATUS_WOULD_BLOCK 256 launches ahead of the GPU -- xengXgpuWaitLaunchSlot, then retry)
int xengXgpuTryKernelAsyncSlab(const void* packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int chan0_pipeline, void* out_dev,
                               int doDump, void* acc_dev, int acc_mode) {
    return kernel_slab(packets_dev, npkt, pkt_stride, seq0, chan0_pipeline, out_dev, doDump, acc_dev, acc_mode, false);
}

static int slab_stats_locked(int* nscattered, int* nirregular) {
    XgpuContext& x = g_ctx;
    int ns = x.slab_forced, ni = 0;
    x.slab_forced = 0;
    if (x.slab_index.tab32) {
        XENG_HIP(hipSetDevice(x.gpu));
        stream_tick(STREAM_XGPU);
        int a = 0, b = 0;
        if (int rcs = slab_index_site_read(x.stream, x.slab_index, &a, &b)) return rcs;
        ns += a; ni += b;
    }
    if (nscattered) *nscattered = ns;
    if (nirregular) *nirregular = ni;
    return XENG_STATUS_SUCCESS;
}

// gulps handed over as slabs that went through zero-fill + scatter (packets of another geometry; round 4: any irregular slab) since
// the last call; waits for the staging stream
int xengXgpuGetSlabFallbacks(int* nfallback) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    if (!nfallback) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "GetSlabFallbacks: null pointer");
    *nfallback = 0;
    return slab_stats_locked(nfallback, nullptr);
}

// ... and (round 5) the gulps that were read in place although their packets were not all in place (lost, shifted, reordered,
// duplicated, foreign): through their offset table, no scatter
int xengXgpuGetSlabStats(int* nscattered, int* nirregular) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    if (!nscattered || !nirregular) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "GetSlabStats: null pointer");
    *nscattered = *nirregular = 0;
    return slab_stats_locked(nscattered, nirregular);
}

int xengXgpuSync(void) {
    int gpu, nmm;
    hipStream_t st, mm[XgpuContext::NMM];
    {
        std::lock_guard<std::mutex> lk(g_mu);
        XgpuContext& x = g_ctx;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
        gpu = x.gpu; nmm = x.nmm; st = x.stream;
        for (int t = 0; t < nmm; t++) mm[t] = x.stream_mm2[t];
    }
    XENG_HIP(hipSetDevice(gpu));
    XENG_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < nmm; t++) XENG_HIP(hipStreamSynchronize(mm[t]));
    drain_timer();
    return XENG_STATUS_SUCCESS;
}

int xengXgpuSyncLag(int lag) {
    PendingWait pw;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        XgpuContext& x = g_ctx;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
        if (lag < 0 || lag > 3) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "SyncLag: lag must be 0..3 (got %d)", lag);
        pw.gpu = x.gpu;
        if (x.ndump > (unsigned long long)lag) pw.dump = launch_event(x.dump_seq[(x.ndump - 1 - lag) & 3]);
    }
    int rc = wait_unlocked(pw);
    if (rc) return rc;
    drain_timer();
    return XENG_STATUS_SUCCESS;
}

int xengXgpuDumpDone(int lag, int* done) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    if (!done || lag < 0 || lag > 3) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "DumpDone: lag must be 0..3 (got %d)", lag);
    *done = 1;
    if (x.ndump > (unsigned long long)lag)
        if (hipEvent_t ev = launch_event(x.dump_seq[(x.ndump - 1 - lag) & 3])) {
            XENG_HIP(hipSetDevice(x.gpu));
            const hipError_t e = hipEventQuery(ev);
            if (e != hipSuccess && e != hipErrorNotReady) XENG_HIP(e);
            if (e == hipErrorNotReady) (void)hipGetLastError();
            *done = e == hipSuccess;
        }
    if (*done) x.timer.drain();          // (as xengXgpuSyncLag does: finished timing pairs are folded into the profile)
    return XENG_STATUS_SUCCESS;
}

// Drops the partial integration.  The state is reset under the lock; the wait for what was already enqueued happens
// outside it (other blocks' calls must not stall behind a whole contraction).
int xengXgpuReset(void) {
    int gpu, nmm;
    hipStream_t st, mm[XgpuContext::NMM];
    {
        std::lock_guard<std::mutex> lk(g_mu);
        XgpuContext& x = g_ctx;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
        x.nfilled = 0;
        x.slab_mode = false;
        x.acc_started = false;
        x.acc_out = nullptr;
        g_epoch++;
        gpu = x.gpu; nmm = x.nmm; st = x.stream;
        for (int t = 0; t < nmm; t++) mm[t] = x.stream_mm2[t];
    }
    XENG_HIP(hipSetDevice(gpu));
    XENG_HIP(hipStreamSynchronize(st));
    for (int t = 0; t < nmm; t++) XENG_HIP(hipStreamSynchronize(mm[t]));
    drain_timer();
    return XENG_STATUS_SUCCESS;
}

// Host-buffer convenience path of xgpu_test.py:86-89.  One caller at a time (g_correlate_mu keeps the private device
// buffers to one call); the context lock is held to enqueue only.
static std::mutex g_correlate_mu;
int xengXgpuCorrelate(const void* in_host, void* out_host, int doDump) {
    std::lock_guard<std::mutex> clk(g_correlate_mu);
    PendingWait pw;
    hipStream_t st;
    int32_t* out_dev;
    size_t out_bytes;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        XgpuContext& x = g_ctx;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
        if (!in_host || !out_host) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xgpu: null buffer");
        XENG_HIP(hipSetDevice(x.gpu));
        const size_t in_bytes = (size_t)x.cfg.ntime_gulp * x.cfg.nchan * x.ninput;
        out_bytes = (size_t)x.matlen * 2 * sizeof(int32_t);
        if (!x.in_dev) XENG_HIP(hipMalloc((void**)&x.in_dev, in_bytes));
        if (!x.out_dev) XENG_HIP(hipMalloc((void**)&x.out_dev, out_bytes));
        XENG_HIP(hipMemcpyAsync(x.in_dev, in_host, in_bytes, hipMemcpyHostToDevice, x.stream));
        staging_stream_touched();
        int rc = kernel_locked(x.in_dev, x.out_dev, doDump, true, &pw);
        if (rc) return rc;
        st = x.stream; out_dev = x.out_dev;
    }
    int rc = wait_unlocked(pw);
    if (rc) return rc;
    if (doDump) {
        XENG_HIP(hipMemcpyAsync(out_host, out_dev, out_bytes, hipMemcpyDeviceToHost, st));
        stream_tick(STREAM_XGPU);
        XENG_HIP(hipStreamSynchronize(st));
        drain_timer();
    }
    return XENG_STATUS_SUCCESS;
}

int xengXgpuGetOrder(const int32_t* antpol_to_input, int32_t* antpol_to_bl, int32_t* is_conj) {
    XgpuConfig cfg;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        cfg = g_ctx.live ? g_ctx.cfg : g_cfg;
    }
    if (!antpol_to_input || !antpol_to_bl || !is_conj) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "GetOrder: null array");
    const int bad = get_order_host(antpol_to_input, antpol_to_bl, is_conj, cfg.nstand, cfg.npol);
    if (bad >= 0) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "GetOrder: input id %d out of range at [%d]", antpol_to_input[bad], bad);
    return XENG_STATUS_SUCCESS;
}

// The consumers of a visibility span (CorrSubsel, CorrOutputFull: their own threads in the reference,
// lwa352-pipeline.py:232-262) run on a stream of their own, ordered behind the contraction that produced the span
// by its completion event -- not by the X-engine's context lock, and they wait for their own kernel only.
// Every call waits on an event of its own (a small pool), so CorrSubsel's call never waits behind CorrOutputFull's.
static std::mutex g_evpool_mu;
static std::vector<hipEvent_t> g_evpool;
static int take_event(hipEvent_t* ev) {
    {
        std::lock_guard<std::mutex> lk(g_evpool_mu);
        if (!g_evpool.empty()) { *ev = g_evpool.back(); g_evpool.pop_back(); return XENG_STATUS_SUCCESS; }
    }
    XENG_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return XENG_STATUS_SUCCESS;
}
static void give_event(hipEvent_t ev) {
    std::lock_guard<std::mutex> lk(g_evpool_mu);
    g_evpool.push_back(ev);
}
// waits for the caller's own kernel, outside every lock
static int finish_consumer(hipEvent_t ev) {
    const hipError_t e = hipEventSynchronize(ev);
    give_event(ev);
    XENG_HIP(e);
    return XENG_STATUS_SUCCESS;
}

int xengXgpuSubSelect(const void* in_dev, void* out_dev, const int32_t* vismap_dev, const int32_t* conj_dev,
                      int nvis, int nchan_sum) {
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        XgpuContext& x = g_ctx;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
        if (!in_dev || !out_dev || !vismap_dev || !conj_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "SubSelect: null buffer");
        if (nvis <= 0 || nchan_sum <= 0 || x.cfg.nchan % nchan_sum)
            XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "SubSelect: nvis=%d nchan_sum=%d nchan=%d", nvis, nchan_sum, x.cfg.nchan);
        XENG_HIP(hipSetDevice(x.gpu));
        hipStream_t s;
        int rc = get_stream(STREAM_CONSUMER, &s);
        if (rc) return rc;
        rc = order_after_writer(s, -1, in_dev);
        if (rc) return rc;
        hipLaunchKernelGGL(subselect_kernel, dim3((nvis + 255) / 256, x.cfg.nchan / nchan_sum), dim3(256), 0, s,
                           (const int32_t*)in_dev, (int32_t*)out_dev, vismap_dev, conj_dev, nvis, nchan_sum,
                           x.per_chan, x.matlen);
        XENG_HIP(hipGetLastError());
        stream_tick(STREAM_CONSUMER);
        rc = take_event(&ev);
        if (rc) return rc;
        const hipError_t re = hipEventRecord(ev, s);
        if (re != hipSuccess) { give_event(ev); XENG_HIP(re); }
    }
    return finish_consumer(ev);
}

int xengXgpuPacketize(const void* in_dev, void* out_dev, const int32_t* antpol_to_bl_dev, const int32_t* is_conj_dev,
                      int fmt) {
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        XgpuContext& x = g_ctx;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
        if (!in_dev || !out_dev || !antpol_to_bl_dev || !is_conj_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Packetize: null buffer");
        if (fmt != 0 && fmt != 1) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Packetize: fmt must be 0 ([pol][pol][chan][2]) or 1 ([chan][pol][pol][2])");
        if ((uintptr_t)out_dev & 7) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Packetize: out must be 8-byte aligned");
        // LDS row pitch (int2 units): >= nchan and 1 mod 32, i.e. 2 mod 64 dwords: the 64 rows of phase A fall on distinct banks
        int pitch = x.cfg.nchan;
        while ((pitch & 31) != 1) pitch++;
        const size_t lds = (size_t)64 * pitch * sizeof(int2);
        if (lds > 160 * 1024) XENG_FAIL(XENG_STATUS_UNSUPPORTED, "Packetize: nchan=%d needs %zu B of LDS", x.cfg.nchan, lds);
        XENG_HIP(hipSetDevice(x.gpu));
        if (lds > 64 * 1024)
            XENG_HIP(hipFuncSetAttribute((const void*)packetize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipStream_t s;
        int rc = get_stream(STREAM_CONSUMER, &s);
        if (rc) return rc;
        rc = order_after_writer(s, -1, in_dev);
        if (rc) return rc;
        hipLaunchKernelGGL(packetize_kernel, dim3(x.cfg.nstand, (x.cfg.nstand + 15) / 16), dim3(256), lds, s,
                           (const int32_t*)in_dev, (int2*)out_dev, antpol_to_bl_dev, is_conj_dev, x.cfg.nstand, x.cfg.nchan,
                           x.per_chan, x.matlen, pitch, fmt);
        XENG_HIP(hipGetLastError());
        stream_tick(STREAM_CONSUMER);
        rc = take_event(&ev);
        if (rc) return rc;
        const hipError_t re = hipEventRecord(ev, s);
        if (re != hipSuccess) { give_event(ev); XENG_HIP(re); }
    }
    return finish_consumer(ev);
}

int xengXgpuReorder(const void* in_host, void* out_host, const int32_t* bl, const int32_t* conj) {
    XgpuConfig cfg;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        cfg = g_ctx.live ? g_ctx.cfg : g_cfg;
    }
    if (!in_host || !out_host || !bl || !conj) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Reorder: null buffer");
    const int64_t per_chan = (int64_t)(cfg.nstand / 2 + 1) * (cfg.nstand / 4) * cfg.npol * cfg.npol * 4;
    const int64_t matlen = per_chan * cfg.nchan;
    const size_t nbl = (size_t)cfg.nstand * cfg.nstand * cfg.npol * cfg.npol;
    const long bad = reorder_host((const int32_t*)in_host, (int32_t*)out_host, bl, conj, nbl, cfg.nchan, per_chan, matlen);
    if (bad >= 0) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Reorder: baseline index %d out of range", bl[bad]);
    return XENG_STATUS_SUCCESS;
}

int xengXgpuGetInfo(int* nstand, int* npol, int* nchan, int* ntime_gulp, int64_t* matlen, int* max_gulps) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    if (nstand) *nstand = x.cfg.nstand;
    if (npol) *npol = x.cfg.npol;
    if (nchan) *nchan = x.cfg.nchan;
    if (ntime_gulp) *ntime_gulp = x.cfg.ntime_gulp;
    if (matlen) *matlen = x.matlen;
    if (max_gulps) *max_gulps = x.cap_gulps;
    return XENG_STATUS_SUCCESS;
}

int xengXgpuGetPath(int* fused_corner_turn, int* fp6) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    if (fused_corner_turn) *fused_corner_turn = x.raw ? 1 : 0;
    if (fp6) *fp6 = x.fp6 ? 1 : 0;
    return XENG_STATUS_SUCCESS;
}

int xengXgpuGetKernel(int* waves_per_group, int* mfma_k) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    const bool k16 = x.raw && x.kloop16;
    if (waves_per_group) *waves_per_group = k16 ? 8 : 4;
    if (mfma_k) *mfma_k = k16 ? 64 : 32;
    return XENG_STATUS_SUCCESS;
}

int xengXgpuSetProfiling(int enable) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_ctx.timer.enabled = enable != 0;
    return XENG_STATUS_SUCCESS;
}

int xengXgpuGetTimes(double ms[2], int count[2]) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (x.live && x.stream) {
        XENG_HIP(hipStreamSynchronize(x.stream));
        for (int t = 0; t < x.nmm; t++) XENG_HIP(hipStreamSynchronize(x.stream_mm2[t]));
        x.timer.drain();
    }
    for (int k = 0; k < 2; k++) {
        if (ms) ms[k] = x.timer.total_ms[k];
        if (count) count[k] = x.timer.count[k];
        x.timer.total_ms[k] = 0;
        x.timer.count[k] = 0;
    }
    return XENG_STATUS_SUCCESS;
}

// diagnostic hook: per-wave clock stamps of the last MFMA launch (needs XENG_DBG_STAMPS=1 at Initialize)
int xengXgpuDebugReadStamps(unsigned long long* host, size_t nwords, int* nwaves) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live || !x.stamps) XENG_FAIL(XENG_STATUS_INVALID_STATE, "stamps not enabled");
    const int ng = x.raw ? x.nfg : x.nwg;
    const size_t n = (size_t)x.cfg.nchan * ng * 4 * 8;
    if (nwaves) *nwaves = x.cfg.nchan * ng * 4;
    for (int t = 0; t < x.nmm; t++) XENG_HIP(hipStreamSynchronize(x.stream_mm2[t]));
    if (host) XENG_HIP(hipMemcpy(host, x.stamps, std::min(nwords, n) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return XENG_STATUS_SUCCESS;
}

// debug/test hook (not part of the drop-in surface): copy the staging area to the host
int xengXgpuDebugReadStash(void* host, size_t nbytes, int* cap_kt, int* nblk64) {
    std::lock_guard<std::mutex> lk(g_mu);
    XgpuContext& x = g_ctx;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "xgpu: not initialized");
    if (cap_kt) *cap_kt = x.cap_kt;
    if (nblk64) *nblk64 = x.nblk64;
    if (host) {
        XENG_HIP(hipStreamSynchronize(x.stream));
        XENG_HIP(hipMemcpy(host, x.stash[x.cur], std::min(nbytes, x.stash_bytes), hipMemcpyDeviceToHost));
    }
    return XENG_STATUS_SUCCESS;
}

}  // extern "C"
