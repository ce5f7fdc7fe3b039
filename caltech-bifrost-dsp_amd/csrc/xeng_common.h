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

// Experiment / diagnostic switches (grid sizes, map variants, clock stamps, item order ...) exist only in
// -DXENG_DIAGNOSTICS builds (profiles/); the shipped library reads XENG_RAW, XENG_BEAM[_F32] and XENG_TILING only.
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
