// Host side of the beamformer: process-global context + C ABI.
// Replaces _bf.bfBeamformInitialize / Run / Integrate / IntegrateSingleBeam
// (beamform_block.py:251-253,449; beamform_sum_beams_block.py:245;
//  beamform_sum_single_beam_block.py:114).
#include <cstdlib>
#include <mutex>

#include "beamform_kernels.h"
#include <algorithm>
#include <vector>

#include "xeng_common.h"

namespace xeng {

struct BeamContext {
    bool live = false;
    int gpu = 0, ninput = 0, nchan = 0, ntime = 0, nbeam = 0, ntime_blocks = 0;
    float* scratch = nullptr;  // voltage beams for the ntime_blocks>0 ("integrated") mode
    uint8_t* wprep = nullptr;  // bf16x3-split weights (beam_weights_prep_kernel)
    size_t wprep_bytes = 0;
    int nchunk = 0, nbtile = 0;
    const void* w_cached = nullptr;   // weights the prepared copy was made from ...
    long long w_version = 0;          // ... and their caller-supplied version (0 = never reuse)
    bool use_f32 = false;             // XENG_BEAM_F32=1 / XENG_BEAM=f32: the fp32-MFMA kernel
    bool use_i8 = false;              // default: fixed-point digits on the int8 MFMA (XENG_BEAM=bf16x3: the bf16 split)
    uint8_t* wq = nullptr;            // int8x3 digit planes
    float* wscale = nullptr;          // ... and their per-(channel, beam) scale
    float* wmax = nullptr;            // inlier row maxima (between the prep passes)
    int* wsum = nullptr;              // [nchan][nbtile][3][32] sums of the Wi digits (int8x3 kernel's accumulator start)
    // precision control of the int8x3 route (beamform_kernels.h, "precision of the fixed-point weights")
    int* row_out = nullptr;           // [nchan][nbtile*32][BI_ROW_OUT] outlier inputs per row
    int* route = nullptr;             // [nchan*nbtile] 1 = the tile runs on the bf16x3 kernel; [nchan*nbtile] = any
    int* out_n = nullptr;             // [nchan*nbtile] outlier inputs per tile
    int* out_idx = nullptr;           // [nchan*nbtile][BI_TILE_OUT]
    float2* out_R = nullptr;          // [nchan*nbtile][BI_TILE_OUT][32] their fp32 weights
    int* any_host = nullptr;          // pinned copy of route[nchan*nbtile] ...
    hipEvent_t ev_route = nullptr;    // ... valid once this event has completed
    bool route_known = false, need_bf16 = true;
    unsigned long long* stamps = nullptr;   // diagnostic (XENG_BEAM_STAMPS=1): per wave {entry, first chunk, loop end, exit}
    int nchunk_i8 = 0;
    hipStream_t stream = nullptr;
    // xengBeamformMark / Wait: completion tickets on the beam stream (a ring of events; ticket n -> marks[n % NMARK])
    static constexpr int NMARK = 64;
    hipEvent_t marks[NMARK] = {};
    unsigned long long nmarks = 0;
    // gulps handed over as packet slabs (xengBeamformRunSlabs; slab.h): descriptors + the scratch gulp that an irregular slab is
    // scattered into -- written and read in stream order on the beam stream, so one set.  ONE helper launch per call: measured
    // (profiles/r04/slab_paths.txt) four short launches in front of a 35 us kernel pair cost 18 us, the same passes on a
    // stream of their own with an event each way 9-12 us, one launch 5 us.
    SlabSite slab_site;
    // (round 5) on a lossy link the parts are read through packet indices instead (slab.h, SlabIndexPrep; the TAB instantiations of the
    // int8x3 / bf16x3 kernels): chosen per call by a counter in pinned memory, read without a wait -- any part of the last eight calls
    // not regular, then 64 calls of hysteresis; XENG_SLAB_TABLES=1 / 0 pins it
    SlabIndexPrep slab_ix;
    uint8_t* zero_page = nullptr;       // 64 bytes of zeros: what the rows of a lost packet read
    int slab_irr_seen = 0, slab_recent_irr[8] = {}, slab_recent_n[8] = {}, slab_recent_pos = 0, slab_force_tables = -1;
    unsigned long long slab_tab_until = 0;
    int slab_seen = 0;                  // slab_site.fallbacks_host at the last look ...
    unsigned long long nslab_calls = 0, slab_lossy_until = 0;   // ... and until which call the scatter passes are launched as grids
    GulpDesc* gdesc = nullptr;          // [part]
    SlabArgs* gargs = nullptr;
    uint8_t* slab_scratch = nullptr;    // [ntime][nchan][ninput]
    EventTimer timer;
};
static std::mutex g_bmu;
static BeamContext g_b;

static int beam_destroy_locked() {
    if (!g_b.live) return XENG_STATUS_SUCCESS;
    (void)hipSetDevice(g_b.gpu);
    if (g_b.stream) (void)hipStreamSynchronize(g_b.stream);
    stream_clocks_forget(g_b.gpu, STREAM_BEAM);          // (everything on the stream has completed; the mark events go away below)
    if (g_b.scratch) (void)hipFree(g_b.scratch);
    if (g_b.wprep) (void)hipFree(g_b.wprep);
    if (g_b.wq) (void)hipFree(g_b.wq);
    if (g_b.wscale) (void)hipFree(g_b.wscale);
    if (g_b.wmax) (void)hipFree(g_b.wmax);
    if (g_b.wsum) (void)hipFree(g_b.wsum);
    if (g_b.row_out) (void)hipFree(g_b.row_out);
    if (g_b.route) (void)hipFree(g_b.route);
    if (g_b.out_n) (void)hipFree(g_b.out_n);
    if (g_b.out_idx) (void)hipFree(g_b.out_idx);
    if (g_b.out_R) (void)hipFree(g_b.out_R);
    if (g_b.any_host) (void)hipHostFree(g_b.any_host);
    if (g_b.ev_route) (void)hipEventDestroy(g_b.ev_route);
    for (int k = 0; k < BeamContext::NMARK; k++)
        if (g_b.marks[k]) (void)hipEventDestroy(g_b.marks[k]);
    if (g_b.stamps) (void)hipFree(g_b.stamps);
    slab_site_destroy(&g_b.slab_site);
    slab_index_prep_destroy(&g_b.slab_ix);
    if (g_b.zero_page) (void)hipFree(g_b.zero_page);
    g_b.zero_page = nullptr;
    if (g_b.gdesc) (void)hipFree(g_b.gdesc);
    if (g_b.gargs) (void)hipFree(g_b.gargs);
    if (g_b.slab_scratch) (void)hipFree(g_b.slab_scratch);
    g_b.timer.destroy();
    g_b = BeamContext();
    return XENG_STATUS_SUCCESS;
}

// pow_out != nullptr: integrated-power mode, fused into the int8x3 kernel when that is possible (see the kernel); returns
// *fused = false when the caller has to run the voltage mode into its scratch and integrate separately
// in1 / split: the gulp in two parts (samples [split, ntime) at in1; beamform_kernels.h GulpAddr); one part: in1 null
// gd: the parts are described on the device (packet slabs; in / in1 unused), split as above
// tab: (with gd) parts may be named by packet indices: the TAB instantiations
static int run_locked(const void* in, float* out, const void* w, long long version, float* pow_out = nullptr, int ntime_sum = 0,
                      bool* fused = nullptr, bool may_wait = true, const void* in1v = nullptr, int split = 0, const GulpDesc* gd = nullptr, bool tab = false) {
    BeamContext& x = g_b;
    if (fused) *fused = false;
    const uint8_t* in1 = (const uint8_t*)in1v;
    if (!in1 && !gd) { in1 = (const uint8_t*)in; split = x.ntime; }
    if (x.use_f32) {
        dim3 grid((x.ntime + BF_NT - 1) / BF_NT, x.nchan, (x.nbeam + 31) / 32);
        int slot = x.timer.begin(x.stream, 0);
        hipLaunchKernelGGL(gd ? beamform_f32_kernel<true> : beamform_f32_kernel<false>, grid, dim3(256), 0, x.stream, (const uint8_t*)in, (const float*)w, out,
                           x.ntime, x.nchan, x.ninput, x.nbeam, in1, split, gd);
        x.timer.end(x.stream, slot);
    stream_tick(STREAM_BEAM);
        XENG_HIP(hipGetLastError());
        return XENG_STATUS_SUCCESS;
    }
    if (x.use_i8) {
        const int ntile = x.nchan * x.nbtile;
        if (!(version != 0 && version == x.w_version && w == x.w_cached)) {
            // row statistics -> outliers / routing -> digits (and, for routed tiles only, the bf16 split)
            XENG_HIP(hipMemsetAsync(x.route, 0, (size_t)(ntile + 1) * sizeof(int), x.stream));
            hipLaunchKernelGGL(beam_weights_rowstat_kernel, dim3(8 * x.nbtile, x.nchan), dim3(256), 0, x.stream,
                               (const float*)w, x.wscale, x.wmax, x.row_out, x.route, x.wsum, x.nchan, x.nbeam, x.ninput, x.nbtile);
            const size_t ol = (size_t)((x.ninput + 63) & ~63) + (BI_TILE_OUT + 1) * sizeof(int);
            hipLaunchKernelGGL(beam_weights_outlier_kernel, dim3(x.nbtile, x.nchan), dim3(256), ol, x.stream,
                               (const float*)w, x.row_out, x.out_n, x.out_idx, x.out_R, x.route, x.nchan, x.nbeam, x.ninput, x.nbtile);
            hipLaunchKernelGGL(beam_weights_prep_i8_kernel, dim3(x.nchunk_i8 * BI_KS, x.nbtile, x.nchan), dim3(256), 0, x.stream,
                               (const float*)w, x.wq, x.wmax, x.nchan, x.nbeam, x.ninput, x.nchunk_i8 * BI_KS, x.nbtile, x.route, x.wsum);
            hipLaunchKernelGGL(beam_weights_prep_kernel, dim3((x.ninput + 63) / 64, x.nbtile, x.nchan), dim3(256), 0, x.stream,
                               (const float*)w, x.wprep, x.nchan, x.nbeam, x.ninput, x.nchunk, x.nbtile, x.route);
            XENG_HIP(hipGetLastError());
            // whether any tile was routed travels to the host without a wait: a caller that keeps its weights
            // (versioned) skips the bf16x3 launch once the answer has arrived and is "none"
            XENG_HIP(hipMemcpyAsync(x.any_host, x.route + ntile, sizeof(int), hipMemcpyDeviceToHost, x.stream));
            XENG_HIP(hipEventRecord(x.ev_route, x.stream));
            x.route_known = false;
            x.need_bf16 = true;
            x.w_cached = w;
            x.w_version = version;
        } else if (!x.route_known && hipEventQuery(x.ev_route) == hipSuccess) {
            x.route_known = true;
            x.need_bf16 = *x.any_host != 0;
        }
        if (pow_out && !x.route_known) {
            // Integrated-power mode: which path forms the power sums (the fused epilogue, or Run -> Integrate) must not
            // depend on how far the GPU has got -- the two sum in different orders, so the last bits would differ from
            // call to call.  One wait per weight upload, in this mode only: the routing answer decides, not the clock.
            if (!may_wait && hipEventQuery(x.ev_route) != hipSuccess) {
                (void)hipGetLastError();
                // (the weights are prepared and remembered: the blocking call that follows finds them and only waits)
                XENG_FAIL(XENG_STATUS_WOULD_BLOCK, "Beamform: the routing answer of a weight upload is not back yet");
            }
            XENG_HIP(hipEventSynchronize(x.ev_route));
            x.route_known = true;
            x.need_bf16 = *x.any_host != 0;
        }
        dim3 grid(((x.ntime + BI_NT - 1) / BI_NT) * x.nchan * x.nbtile);
        int slot = x.timer.begin(x.stream, 0);
        // the fused power sums need every tile on this kernel (known once the routing answer of these weights has come
        // back), time blocks that straddle at most two work-groups, and whole beam pairs
        const bool fuse = pow_out && x.route_known && !x.need_bf16 && ntime_sum <= BI_NT && x.nbeam % 2 == 0;
        if (fuse) {
            XENG_HIP(hipMemsetAsync(pow_out, 0, (size_t)(x.nbeam / 2) * (x.ntime / ntime_sum) * x.nchan * 4 * sizeof(float), x.stream));
            *fused = true;
        }
        hipLaunchKernelGGL(gd ? (tab ? beamform_i8x3_kernel<true, true> : beamform_i8x3_kernel<true>) : beamform_i8x3_kernel<false>, grid, dim3(256), 0, x.stream,
                           (const uint8_t*)in, x.wq, x.wscale, x.wsum, out,
                           x.ntime, x.nchan, x.ninput, x.nbeam, x.nchunk_i8, x.nbtile, x.route, x.out_n, x.out_idx, x.out_R, x.stamps,
                           fuse ? pow_out : (float*)nullptr, ntime_sum, in1, split, gd, (const uint8_t*)x.zero_page);
        if (x.need_bf16) {
            dim3 grid3(((x.ntime + BF3_NT - 1) / BF3_NT) * x.nchan * x.nbtile);
            hipLaunchKernelGGL(gd ? (tab ? beamform_bf16x3_kernel<true, true> : beamform_bf16x3_kernel<true>) : beamform_bf16x3_kernel<false>, grid3, dim3(64 * BF3_NW), 0,
                               x.stream, (const uint8_t*)in, x.wprep, out,
                               x.ntime, x.nchan, x.ninput, x.nbeam, x.nchunk, x.nbtile, x.route, in1, split, gd, (const uint8_t*)x.zero_page);
        }
        x.timer.end(x.stream, slot);
    stream_tick(STREAM_BEAM);
        XENG_HIP(hipGetLastError());
        return XENG_STATUS_SUCCESS;
    }
    // split the fp32 weights into three bf16 terms unless this exact (pointer, version) is already prepared
    if (!(version != 0 && version == x.w_version && w == x.w_cached)) {
        hipLaunchKernelGGL(beam_weights_prep_kernel, dim3((x.ninput + 63) / 64, x.nbtile, x.nchan), dim3(256), 0, x.stream,
                           (const float*)w, x.wprep, x.nchan, x.nbeam, x.ninput, x.nchunk, x.nbtile, (const int*)nullptr);
        XENG_HIP(hipGetLastError());
        x.w_cached = w;
        x.w_version = version;
    }
    dim3 grid(((x.ntime + BF3_NT - 1) / BF3_NT) * x.nchan * x.nbtile);
    int slot = x.timer.begin(x.stream, 0);
    hipLaunchKernelGGL(gd ? (tab ? beamform_bf16x3_kernel<true, true> : beamform_bf16x3_kernel<true>) : beamform_bf16x3_kernel<false>, grid, dim3(64 * BF3_NW), 0, x.stream,
                       (const uint8_t*)in, x.wprep, out,
                       x.ntime, x.nchan, x.ninput, x.nbeam, x.nchunk, x.nbtile, (const int*)nullptr, in1, split, gd, (const uint8_t*)x.zero_page);
    x.timer.end(x.stream, slot);
    stream_tick(STREAM_BEAM);
    XENG_HIP(hipGetLastError());
    return XENG_STATUS_SUCCESS;
}

static int integrate_locked(const void* in, void* out, int ntime_sum, int pair0, int npair) {
    BeamContext& x = g_b;
    if (ntime_sum <= 0 || x.ntime % ntime_sum) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Integrate: ntime %d not a multiple of ntime_sum %d", x.ntime, ntime_sum);
    if (x.nbeam % 2) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Integrate: nbeam %d must be even (X/Y pairs)", x.nbeam);
    const int nblk = x.ntime / ntime_sum;
    dim3 grid((x.nchan + 3) / 4, npair, std::max(1, std::min(8, (nblk + 7) / 8)));
    int slot = x.timer.begin(x.stream, 1);
    hipLaunchKernelGGL(beam_integrate_kernel, grid, dim3(256), 0, x.stream, (const float2*)in, (float4*)out, x.nchan,
                       x.nbeam, x.ntime, ntime_sum, pair0, npair);
    x.timer.end(x.stream, slot);
    stream_tick(STREAM_BEAM);
    XENG_HIP(hipGetLastError());
    return XENG_STATUS_SUCCESS;
}

}  // namespace xeng

using namespace xeng;

extern "C" {

int xengBeamformInitialize(int gpu, int ninput, int nchan, int ntime, int nbeam, int ntime_blocks) {
    if (ninput <= 0 || ninput % 4 || nchan <= 0 || ntime <= 0 || nbeam <= 0 || ntime_blocks < 0)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: bad sizes ninput=%d nchan=%d ntime=%d nbeam=%d ntime_blocks=%d",
                  ninput, nchan, ntime, nbeam, ntime_blocks);
    if (ntime_blocks > 0 && ntime % ntime_blocks)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: ntime %d not a multiple of ntime_blocks %d", ntime, ntime_blocks);
    std::lock_guard<std::mutex> lk(g_bmu);
    beam_destroy_locked();
    BeamContext& x = g_b;
    x.gpu = gpu < 0 ? 0 : gpu;
    XENG_HIP(hipSetDevice(x.gpu));
    x.ninput = ninput; x.nchan = nchan; x.ntime = ntime; x.nbeam = nbeam; x.ntime_blocks = ntime_blocks;
    if (ntime_blocks > 0) XENG_HIP(hipMalloc((void**)&x.scratch, (size_t)nchan * nbeam * ntime * 8));
    x.nchunk = (ninput + BF3_KC - 1) / BF3_KC;
    x.nbtile = (nbeam + 31) / 32;
    x.wprep_bytes = (size_t)nchan * x.nbtile * x.nchunk * BF3_WCHUNK;
    XENG_HIP(hipMalloc((void**)&x.wprep, x.wprep_bytes));
    XENG_HIP(hip_memset_now(x.wprep, 0, x.wprep_bytes));
    // the bf16x3 kernel moves the packed voltages by 16-byte LDS-DMA columns: inputs must be a multiple of 16
    const char* mode = getenv("XENG_BEAM");
    x.use_f32 = getenv("XENG_BEAM_F32") != nullptr || (mode && !strcmp(mode, "f32")) || (ninput % 16) != 0;
    x.use_i8 = !x.use_f32 && !(mode && !strcmp(mode, "bf16x3"));     // default; XENG_BEAM=bf16x3 | f32 select the others
    // XENG_SLAB_TABLES=1 / 0: packet slabs always / never through their packet indices (default: by strides until the link loses packets)
    const char* st = getenv("XENG_SLAB_TABLES");
    x.slab_force_tables = st ? (strcmp(st, "0") ? 1 : 0) : -1;
    if (x.use_i8) {
        x.nchunk_i8 = (ninput + BI_KC - 1) / BI_KC;
        const size_t qb = (size_t)nchan * x.nbtile * x.nchunk_i8 * BI_WCHUNK;
        XENG_HIP(hipMalloc((void**)&x.wq, qb));
        XENG_HIP(hip_memset_now(x.wq, 0, qb));
        XENG_HIP(hipMalloc((void**)&x.wscale, (size_t)nchan * x.nbtile * 32 * sizeof(float)));
        XENG_HIP(hipMalloc((void**)&x.wmax, (size_t)nchan * x.nbtile * 32 * sizeof(float)));
        XENG_HIP(hipMalloc((void**)&x.wsum, (size_t)nchan * x.nbtile * 3 * 32 * sizeof(int)));
        const size_t ntile = (size_t)nchan * x.nbtile;
        XENG_HIP(hipMalloc((void**)&x.row_out, ntile * 32 * BI_ROW_OUT * sizeof(int)));
        XENG_HIP(hipMalloc((void**)&x.route, (ntile + 1) * sizeof(int)));
        XENG_HIP(hipMalloc((void**)&x.out_n, ntile * sizeof(int)));
        XENG_HIP(hipMalloc((void**)&x.out_idx, ntile * BI_TILE_OUT * sizeof(int)));
        XENG_HIP(hipMalloc((void**)&x.out_R, ntile * BI_TILE_OUT * 32 * sizeof(float2)));
        XENG_HIP(hipHostMalloc((void**)&x.any_host, sizeof(int)));
        XENG_HIP(hipEventCreateWithFlags(&x.ev_route, hipEventDisableTiming));
        if (diag_env("XENG_BEAM_STAMPS")) {
            const size_t nw = (size_t)((ntime + BI_NT - 1) / BI_NT) * nchan * x.nbtile * 4 * 4;
            XENG_HIP(hipMalloc((void**)&x.stamps, nw * sizeof(unsigned long long)));
            XENG_HIP(hip_memset_now(x.stamps, 0, nw * sizeof(unsigned long long)));
        }
    }
    int rc = get_stream(STREAM_BEAM, &x.stream);
    if (rc) return rc;
    XENG_HIP(hipStreamSynchronize(nullptr));       // (null-stream fills above: complete before the beam stream's first kernel; hip_memset_now)
    x.live = true;
    return XENG_STATUS_SUCCESS;
}

// diagnostic hook: per-wave clock stamps of the last int8x3 launch (needs XENG_BEAM_STAMPS=1 at Initialize)
int xengBeamformDebugReadStamps(unsigned long long* host, size_t nwords) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live || !x.stamps) XENG_FAIL(XENG_STATUS_INVALID_STATE, "beam stamps not enabled");
    XENG_HIP(hipStreamSynchronize(x.stream));
    XENG_HIP(hipMemcpy(host, x.stamps, nwords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return XENG_STATUS_SUCCESS;
}

int xengBeamformDestroy(void) {
    std::lock_guard<std::mutex> lk(g_bmu);
    return beam_destroy_locked();
}

int xengBeamformRun(const void* in_dev, void* out_dev, const void* weights_dev) {
    return xengBeamformRunVersioned(in_dev, out_dev, weights_dev, 0);
}

// ... and (round 5) the parts that were read where they lay although their packets were not all in place, through their index
int xengBeamformGetSlabStats(int* nscattered, int* nirregular) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    if (!nscattered || !nirregular) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "GetSlabStats: null pointer");
    *nscattered = *nirregular = 0;
    if (!x.slab_site.tally) return XENG_STATUS_SUCCESS;
    XENG_HIP(hipSetDevice(x.gpu));
    stream_tick(STREAM_BEAM);
    if (x.slab_ix.irregular) {
        XENG_HIP(hipMemcpyAsync(nirregular, x.slab_ix.irregular, sizeof(int), hipMemcpyDeviceToHost, x.stream));
        XENG_HIP(hipMemsetAsync(x.slab_ix.irregular, 0, sizeof(int), x.stream));
    }
    return slab_site_read_fallbacks(x.stream, x.slab_site, nscattered);
}

static int run_versioned(const void* in_dev, void* out_dev, const void* weights_dev, long long weights_version, bool may_wait,
                         const void* in1_dev = nullptr, int ntime0 = 0);

int xengBeamformRunVersioned(const void* in_dev, void* out_dev, const void* weights_dev, long long weights_version) {
    return run_versioned(in_dev, out_dev, weights_dev, weights_version, true);
}

int xengBeamformTryRunVersioned(const void* in_dev, void* out_dev, const void* weights_dev, long long weights_version) {
    return run_versioned(in_dev, out_dev, weights_dev, weights_version, false);
}

int xengBeamformRunParts(const void* in0_dev, int ntime0, const void* in1_dev, void* out_dev, const void* weights_dev, long long weights_version) {
    return run_versioned(in0_dev, out_dev, weights_dev, weights_version, true, in1_dev, ntime0);
}

int xengBeamformTryRunParts(const void* in0_dev, int ntime0, const void* in1_dev, void* out_dev, const void* weights_dev, long long weights_version) {
    return run_versioned(in0_dev, out_dev, weights_dev, weights_version, false, in1_dev, ntime0);
}

// The gulp as the slabs of F-engine packets it arrived in (slab.h): one slab (packets1 null, ntime0 = ntime) or two consecutive
// ones (samples [0, ntime0) and [ntime0, ntime): two capture gulps per beamformer gulp, lwa352-pipeline.py:172,279-282).  Each is
// verified on the beam stream (one launch); a regular slab is read where it lies, anything else is scattered into the context's scratch gulp
// first (the rules of xengSnap2UnpackAsync: missing samples read as zero, foreign and out-of-window packets dropped).
static int run_slabs(const void* packets0_dev, int npkt0, int ntime0, const void* packets1_dev, int npkt1, size_t pkt_stride, uint64_t seq0,
                     int chan0_pipeline, void* out_dev, const void* weights_dev, long long weights_version, bool may_wait) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized (call xengBeamformInitialize)");
    if (!packets0_dev || !out_dev || !weights_dev || npkt0 < 0 || npkt1 < 0 || pkt_stride < 32 || pkt_stride > 0x7FFFFFFFu)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: bad slab (npkt %d / %d, stride %zu)", npkt0, npkt1, pkt_stride);
    if (((uintptr_t)weights_dev & 15) || ((uintptr_t)out_dev & 15)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: weights/out must be 16-byte aligned");
    if (!packets1_dev) ntime0 = x.ntime;
    // (parts on 16-sample boundaries: a wave's 16 rows of one load never straddle two descriptors; and whole 16-byte pieces)
    if (ntime0 <= 0 || ntime0 > x.ntime || (packets1_dev && (ntime0 == x.ntime || ntime0 % 16)) || x.ninput % 16)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: slab parts of %d + %d samples (inputs %d): parts must be multiples of 16 samples, inputs of 16",
                  ntime0, x.ntime - ntime0, x.ninput);
    XENG_HIP(hipSetDevice(x.gpu));
    const size_t row = (size_t)x.nchan * x.ninput;
    if (!x.gdesc) {
        if (int rc = slab_site_create(&x.slab_site)) return rc;
        XENG_HIP(hipMalloc((void**)&x.gdesc, 2 * sizeof(GulpDesc)));
        XENG_HIP(hip_memset_now(x.gdesc, 0, 2 * sizeof(GulpDesc)));
        XENG_HIP(hipMalloc((void**)&x.gargs, 2 * sizeof(SlabArgs)));
        XENG_HIP(hip_memset_now(x.gargs, 0, 2 * sizeof(SlabArgs)));
        XENG_HIP(hipMalloc((void**)&x.slab_scratch, (size_t)x.ntime * row));
        if (x.ninput % 64 == 0 && !x.use_f32) {
            if (int rc = slab_index_prep_create(&x.slab_ix, x.ntime, x.ninput)) return rc;
            XENG_HIP(hipMalloc((void**)&x.zero_page, 64));
            XENG_HIP(hip_memset_now(x.zero_page, 0, 64));
        }
        x.slab_irr_seen = 0; x.slab_recent_pos = 0; x.slab_tab_until = 0;
        for (int k = 0; k < 8; k++) x.slab_recent_irr[k] = x.slab_recent_n[k] = 0;
    }
    const int nparts = packets1_dev ? 2 : 1;
    SlabArgs a[2];
    bool maybe[2] = {false, false};
    uint8_t* scratch[2] = {x.slab_scratch, x.slab_scratch + (size_t)ntime0 * row};
    for (int k = 0; k < nparts; k++) {
        a[k].pkts = (const uint8_t*)(k ? packets1_dev : packets0_dev); a[k].npkt = k ? npkt1 : npkt0; a[k].stride = (uint32_t)pkt_stride;
        a[k].seq0 = seq0 + (k ? (uint64_t)ntime0 : 0); a[k].ntime = k ? x.ntime - ntime0 : ntime0; a[k].chan0 = chan0_pipeline; a[k].nchan = x.nchan;
        a[k].ninput = x.ninput; a[k].nblk = x.ninput / 64;
        maybe[k] = slab_maybe_regular(a[k], 1);          // (the kernels form 64-bit row addresses)
    }
    // a clean link: one launch, and an irregular gulp is scattered by that launch's last work-group (0.9 ms per 32 MB part).
    // Once a gulp has taken that path -- the counter in pinned memory moves, a moment later -- the next 64 calls launch the
    // scatter passes as grids behind the verify pass instead (two more short launches, 9 us per call; 20 x faster when needed).
    if (const int seen = *(volatile int*)x.slab_site.fallbacks_host; seen != x.slab_seen) { x.slab_seen = seen; x.slab_lossy_until = x.nslab_calls + 64; }
    const bool lossy = x.nslab_calls++ < x.slab_lossy_until;
    // (round 5) through the packet indices once a part of the last eight calls was not regular (the verify pass and the index pass both
    // count them in pinned memory); by strides -- round 4's kernels, an irregular part scattered -- on a clean link
    bool tab = false;
    if (x.slab_ix.tab[0]) {
        const int seen = *(volatile int*)x.slab_ix.irregular_host + x.slab_seen;     // (by index: irregular or scattered; by strides: scattered)
        x.slab_recent_irr[x.slab_recent_pos] = seen - x.slab_irr_seen;
        x.slab_recent_n[x.slab_recent_pos] = nparts;
        x.slab_recent_pos = (x.slab_recent_pos + 1) & 7;
        x.slab_irr_seen = seen;
        int irr = 0, n = 0;
        for (int k = 0; k < 8; k++) { irr += x.slab_recent_irr[k]; n += x.slab_recent_n[k]; }
        // (the beam stream may run many calls behind its enqueuer, so the counts arrive late and in bursts: once over the threshold the
        // next 64 calls stay on the indices)
        // (by index a regular part costs 3 % more than by strides, a scattered part 40 %: any irregular part among the last eight calls
        // is enough)
        if (irr > 0 && n > 0) x.slab_tab_until = x.nslab_calls + 64;
        tab = x.slab_force_tables > 0 || (x.slab_force_tables < 0 && x.nslab_calls < x.slab_tab_until);
    }
    if (tab) {
        bool ok[2] = {false, false};
        for (int k = 0; k < nparts; k++) ok[k] = slab_index_prep_ok(a[k]);
        if (int rc = slab_index_prepare_enqueue(x.stream, x.slab_site, x.slab_ix, a, ok, nparts, x.gdesc, x.gargs, scratch, !lossy)) return rc;
    } else if (int rc = slab_prepare_enqueue(x.stream, x.slab_site, a, maybe, nparts, x.gdesc, x.gargs, scratch, !lossy)) return rc;
    if (lossy)
        if (int rc = slab_fallback_enqueue(x.stream, x.gdesc, x.gargs, nparts)) return rc;
    stream_tick(STREAM_BEAM);
    if (x.ntime_blocks == 0) return run_locked(nullptr, (float*)out_dev, weights_dev, weights_version, nullptr, 0, nullptr, true, nullptr, ntime0, x.gdesc, tab);
    bool fused = false;
    int rc = run_locked(nullptr, x.scratch, weights_dev, weights_version, (float*)out_dev, x.ntime / x.ntime_blocks, &fused, may_wait, nullptr, ntime0, x.gdesc, tab);
    if (rc || fused) return rc;
    return integrate_locked(x.scratch, out_dev, x.ntime / x.ntime_blocks, 0, x.nbeam / 2);
}

int xengBeamformRunSlabs(const void* packets0_dev, int npkt0, int ntime0, const void* packets1_dev, int npkt1, size_t pkt_stride, uint64_t seq0,
                         int chan0_pipeline, void* out_dev, const void* weights_dev, long long weights_version) {
    return run_slabs(packets0_dev, npkt0, ntime0, packets1_dev, npkt1, pkt_stride, seq0, chan0_pipeline, out_dev, weights_dev, weights_version, true);
}

// (never waits: see xengBeamformTryRunVersioned.  A WOULD_BLOCK call has enqueued its verify pass; the blocking call that follows
// enqueues it again -- harmless, same stream, same result)
int xengBeamformTryRunSlabs(const void* packets0_dev, int npkt0, int ntime0, const void* packets1_dev, int npkt1, size_t pkt_stride, uint64_t seq0,
                            int chan0_pipeline, void* out_dev, const void* weights_dev, long long weights_version) {
    return run_slabs(packets0_dev, npkt0, ntime0, packets1_dev, npkt1, pkt_stride, seq0, chan0_pipeline, out_dev, weights_dev, weights_version, false);
}

// beamformer gulp parts handed over as slabs that took the scratch path since the last call; waits for the beam stream
int xengBeamformGetSlabFallbacks(int* nfallback) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    if (!nfallback) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "GetSlabFallbacks: null pointer");
    *nfallback = 0;
    if (!x.slab_site.tally) return XENG_STATUS_SUCCESS;
    XENG_HIP(hipSetDevice(x.gpu));
    stream_tick(STREAM_BEAM);
    return slab_site_read_fallbacks(x.stream, x.slab_site, nfallback);
}

static int run_versioned(const void* in_dev, void* out_dev, const void* weights_dev, long long weights_version, bool may_wait,
                         const void* in1_dev, int ntime0) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized (call xengBeamformInitialize)");
    if (!in_dev || !out_dev || !weights_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: null buffer");
    if (((uintptr_t)weights_dev & 15) || ((uintptr_t)out_dev & 15) || ((uintptr_t)in_dev & 3))
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: weights/out must be 16-byte, in 4-byte aligned");
    if (in1_dev) {
        if (ntime0 <= 0 || ntime0 >= x.ntime) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: first part of %d samples in a gulp of %d", ntime0, x.ntime);
        if ((uintptr_t)in1_dev & 3) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Beamform: in must be 4-byte aligned");
    }
    XENG_HIP(hipSetDevice(x.gpu));
    if (x.ntime_blocks == 0) return run_locked(in_dev, (float*)out_dev, weights_dev, weights_version, nullptr, 0, nullptr, true, in1_dev, ntime0);
    bool fused = false;
    int rc = run_locked(in_dev, x.scratch, weights_dev, weights_version, (float*)out_dev, x.ntime / x.ntime_blocks, &fused, may_wait, in1_dev, ntime0);
    if (rc || fused) return rc;
    return integrate_locked(x.scratch, out_dev, x.ntime / x.ntime_blocks, 0, x.nbeam / 2);
}

int xengBeamformIntegrate(const void* in_dev, void* out_dev, int ntime_sum) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized (the Beamform block initializes the shared context)");
    if (!in_dev || !out_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Integrate: null buffer");
    XENG_HIP(hipSetDevice(x.gpu));
    return integrate_locked(in_dev, out_dev, ntime_sum, 0, x.nbeam / 2);
}

int xengBeamformIntegrateSingleBeam(const void* in_dev, void* out_dev, int ntime_sum, int beam_id) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    if (!in_dev || !out_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Integrate: null buffer");
    if (beam_id < 0 || beam_id >= x.nbeam / 2) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "IntegrateSingleBeam: beam %d out of range", beam_id);
    XENG_HIP(hipSetDevice(x.gpu));
    return integrate_locked(in_dev, out_dev, ntime_sum, beam_id, 1);
}

int xengBeamformSync(void) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    XENG_HIP(hipSetDevice(x.gpu));
    XENG_HIP(hipStreamSynchronize(x.stream));
    x.timer.drain();
    return XENG_STATUS_SUCCESS;
}

// Completion tickets, so that a block can keep several gulps in flight and commit each output span when ITS kernels are
// done (the Beamform and BeamformSumBeams blocks share this stream; xengBeamformSync waits for everything on it).
int xengBeamformMark(unsigned long long* ticket) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    if (!ticket) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Mark: null ticket");
    XENG_HIP(hipSetDevice(x.gpu));
    hipEvent_t& ev = x.marks[x.nmarks % BeamContext::NMARK];
    if (!ev) XENG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const unsigned long long upto = stream_clock_now(STREAM_BEAM);      // (read BEFORE the record: everything counted precedes it)
    XENG_HIP(hipEventRecord(ev, x.stream));
    stream_clock_external_mark(STREAM_BEAM, ev, upto);                  // stamps of released spans find this event: none of their own on this stream
    *ticket = ++x.nmarks;
    return XENG_STATUS_SUCCESS;
}

int xengBeamformWait(unsigned long long ticket) {
    hipEvent_t ev = nullptr;
    int gpu = 0;
    {
        std::lock_guard<std::mutex> lk(g_bmu);
        BeamContext& x = g_b;
        if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
        if (ticket == 0 || ticket > x.nmarks) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Wait: unknown ticket %llu", ticket);
        gpu = x.gpu;
        // a ticket whose event slot has been re-recorded is NMARK marks old: wait for the newer record of that slot, which
        // is later on the same stream
        ev = x.marks[(ticket - 1) % BeamContext::NMARK];
    }
    XENG_HIP(hipSetDevice(gpu));
    XENG_HIP(hipEventSynchronize(ev));          // (outside the lock: the other block keeps enqueueing)
    return XENG_STATUS_SUCCESS;
}

int xengBeamformTicketDone(unsigned long long ticket, int* done) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    if (!done || ticket == 0 || ticket > x.nmarks) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "TicketDone: unknown ticket %llu", ticket);
    XENG_HIP(hipSetDevice(x.gpu));
    const hipError_t e = hipEventQuery(x.marks[(ticket - 1) % BeamContext::NMARK]);   // (a re-recorded slot: see Wait)
    if (e != hipSuccess && e != hipErrorNotReady) XENG_HIP(e);
    if (e == hipErrorNotReady) (void)hipGetLastError();
    *done = e == hipSuccess;
    return XENG_STATUS_SUCCESS;
}

int xengBeamformGetRouteInfo(int* tiles_total, int* tiles_bf16, int* outlier_inputs) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (!x.live) XENG_FAIL(XENG_STATUS_INVALID_STATE, "Beamform: not initialized");
    const int ntile = x.nchan * x.nbtile;
    int nbf = 0, nout = 0;
    if (x.use_i8) {
        XENG_HIP(hipSetDevice(x.gpu));
        XENG_HIP(hipStreamSynchronize(x.stream));
        std::vector<int> r(ntile), n(ntile);
        XENG_HIP(hipMemcpy(r.data(), x.route, ntile * sizeof(int), hipMemcpyDeviceToHost));
        XENG_HIP(hipMemcpy(n.data(), x.out_n, ntile * sizeof(int), hipMemcpyDeviceToHost));
        for (int k = 0; k < ntile; k++) { nbf += r[k] != 0; if (!r[k]) nout += n[k]; }
    }
    if (tiles_total) *tiles_total = ntile;
    if (tiles_bf16) *tiles_bf16 = nbf;
    if (outlier_inputs) *outlier_inputs = nout;
    return XENG_STATUS_SUCCESS;
}

int xengBeamformSetProfiling(int enable) {
    std::lock_guard<std::mutex> lk(g_bmu);
    g_b.timer.enabled = enable != 0;
    return XENG_STATUS_SUCCESS;
}

int xengBeamformGetTimes(double ms[2], int count[2]) {
    std::lock_guard<std::mutex> lk(g_bmu);
    BeamContext& x = g_b;
    if (x.live && x.stream) {
        XENG_HIP(hipStreamSynchronize(x.stream));
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

}  // extern "C"
