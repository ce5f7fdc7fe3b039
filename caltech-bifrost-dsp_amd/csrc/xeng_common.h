// Shared host-side helpers for libxeng (error convention, per-device streams).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/xeng.h"

namespace xeng {

// thread-local message behind xengGetLastError()
void set_error(const char* fmt, ...);

#define XENG_FAIL(code, ...)            \
    do {                                \
        ::xeng::set_error(__VA_ARGS__); \
        return (code);                  \
    } while (0)

#define XENG_HIP(call)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            XENG_FAIL(XENG_STATUS_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call,             \
                      hipGetErrorString(e_), __FILE__, __LINE__);                           \
    } while (0)

// Streams owned by the library on the current device: one per block thread of the
// reference pipeline (Corr, CorrAcc map, Beamform, copies) so they overlap
// (lwa352-pipeline.py:296-302 runs one thread per block on the same GPU).
enum StreamId { STREAM_XGPU = 0, STREAM_MAP = 1, STREAM_BEAM = 2, STREAM_COPY = 3, STREAM_XGPU_MM = 4, STREAM_XGPU_MM2 = 5, STREAM_XGPU_MM3 = 6, STREAM_XGPU_MM4 = 7, STREAM_CONSUMER = 8,
                STREAM_XGPU_MM5 = 9, STREAM_XGPU_MM6 = 10, STREAM_XGPU_MM7 = 11, STREAM_XGPU_MM8 = 12, STREAM_COUNT = 13 };
inline StreamId mm_stream_id(int t) { return (StreamId)(t < 4 ? STREAM_XGPU_MM + t : STREAM_XGPU_MM5 + (t - 4)); }
int get_stream(StreamId which, hipStream_t* out);   // lazily created per device
// Work counter of the X-engine's staging stream (STREAM_XGPU: gulp copies, corner turns, enqueue-only ingest scatters):
// whoever enqueues there bumps it, so that a contraction only waits for that stream when something was put on it since
// the last contraction (a barrier packet in front of every launch otherwise).
void staging_stream_touched();
unsigned long long staging_stream_ops();
int sync_all_streams();

// Stream clocks (round 4): the memory-lifetime mechanism of the span rings (ring.hip).  Every enqueue on a library stream
// ticks that stream's clock AFTER the enqueue; a stamp is the vector of clock values at one moment ("everything enqueued so
// far, on every stream").  A stamp is complete when each stream has passed its value -- found out with an event that is
// recorded on the stream only when somebody asks (no event per launch): an event recorded later on the same stream
// completes after all the work the stamp counts.  A ring stamps a span buffer when its last user lets go of it and hands
// it out again only once the stamp is complete, so a buffer released while kernels that were enqueued before the release
// still run is never reissued (or freed) under them.
void stream_tick(StreamId which);                                   // call after every enqueue on stream `which` (current device)
// Which streams can have touched a buffer (a ring's blocks declare it: xengRingDeclareStreams; default: all of them).  A stamp
// waits only for those: a span of the beamformer's output ring does not wait for the 200 us contraction that happened to be
// enqueued before its release.
// STAMP_XGPU_OUT: spans that contractions only WRITE (a visibility span, a long accumulator): the stamp then names the last
// launch enqueued into that very buffer (the X-engine keeps that per buffer anyway) instead of every launch enqueued so far.
enum StampClass { STAMP_XGPU = 1, STAMP_MAP = 2, STAMP_BEAM = 4, STAMP_COPY = 8, STAMP_CONSUMER = 16, STAMP_ALL = 31, STAMP_XGPU_OUT = 32 };
struct Stamp {
    int dev = -1;                                                   // -1: nothing to wait for
    unsigned mask = STAMP_ALL;
    unsigned long long clk[STREAM_COUNT] = {};                      // (the contraction streams' entries are unused: xgpu_launch)
    unsigned long long xgpu_seq = 0, xgpu_epoch = 0;                // gulps registered with the X-engine but not contracted yet: the launch that will read them
    unsigned long long xgpu_launch = 0, xgpu_ctx = 0;               // contractions enqueued so far (each owns a completion event: none is recorded for a stamp)
    bool xgpu_exact = false;                                        // xgpu_launch is the one launch that writes the buffer (STAMP_XGPU_OUT)
};
int stamp_now(Stamp* s, const void* buf = nullptr, unsigned mask = STAMP_ALL, int of_dev = -1);    // of_dev < 0: the calling thread's current device
// *done: every clock has passed; *waitable false: it waits for a launch nobody has enqueued yet (only its enqueuer can end that wait)
int stamp_poll(const Stamp& s, bool* done, bool* waitable);
int stamp_wait(const Stamp& s);                                     // blocks (event waits happen outside every library lock)
void stream_clocks_forget(int dev, StreamId which);                 // the stream has been synchronised: everything ticked so far is complete
// an event somebody records on the stream anyway (xengBeamformMark) serves as a clock mark: upto = stream_clock_now() read BEFORE the record
unsigned long long stream_clock_now(StreamId which);
void stream_clock_external_mark(StreamId which, hipEvent_t ev, unsigned long long upto);
// X-engine side of a stamp (xcorr.hip)
void xgpu_pending_launch(unsigned long long* seq, unsigned long long* epoch, unsigned long long* nlaunch, unsigned long long* ctx);
int xgpu_pending_poll(unsigned long long seq, unsigned long long epoch, bool* done, bool* launched, hipEvent_t* ev, int* gpu);
int xgpu_launches_poll(unsigned long long upto, unsigned long long ctx, bool* done, hipEvent_t* ev, bool exact = false);
unsigned long long xgpu_last_writer(const void* buf);               // number of the last enqueued launch that writes buf (0: none on record = long done)

// Experiment / diagnostic switches (grid sizes, map variants, clock stamps, item order ...) exist only in
// -DXENG_DIAGNOSTICS builds (profiles/); the shipped library reads XENG_RAW, XENG_BEAM[_F32] and XENG_TILING only.
// hipMemset on the null stream, COMPLETE on return.  hipMemset itself may return before the fill has run (it is a kernel on the
// null stream, which the library's non-blocking streams do not wait for): a kernel enqueued right afterwards on one of those
// streams can then write the buffer first and have its words zeroed under it.  Seen once the fill was delayed: under
// `rocprofv3 --pmc`, which runs one kernel at a time in submission order, the first xengBeamformRunSlabs call beside a
// running contraction lost its descriptors to the late fill and the beamformer kernel read a null base.
inline hipError_t hip_memset_now(void* p, int value, size_t n) {
    hipError_t e = hipMemset(p, value, n);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(nullptr);
}

inline const char* diag_env(const char* name) {
#ifdef XENG_DIAGNOSTICS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// profiling helper: pairs of events accumulated per kind
struct EventTimer {
    static constexpr int MAXPEND = 64;
    bool enabled = false;
    hipEvent_t start[MAXPEND], stop[MAXPEND];
    int kind[MAXPEND];
    int npend = 0, ncreated = 0;
    double total_ms[2] = {0, 0};
    int count[2] = {0, 0};
    int begin(hipStream_t s, int k);   // returns slot or -1
    void end(hipStream_t s, int slot);
    int drain();                       // requires the stream to be idle (synchronized)
    void destroy();
};

}  // namespace xeng
