// Device / memory plumbing behind the C ABI (include/xeng.h, "device / memory plumbing").
// Replaces bifrost.device.set_device / stream_synchronize, BFArray(space='cuda'|'cuda_host')
// allocation and copy_array for the hot-path blocks.
#include <atomic>
#include <mutex>
#include <utility>

#include "xeng_common.h"

namespace xeng {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static constexpr int MAXDEV = 16;
static std::mutex g_stream_mu;
static hipStream_t g_streams[MAXDEV][STREAM_COUNT];
static bool g_stream_ok[MAXDEV][STREAM_COUNT];

static std::atomic<unsigned long long> g_staging_ops{0};
void staging_stream_touched() { g_staging_ops.fetch_add(1, std::memory_order_relaxed); }
unsigned long long staging_stream_ops() { return g_staging_ops.load(std::memory_order_relaxed); }

int get_stream(StreamId which, hipStream_t* out) {
    int dev = 0;
    XENG_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAXDEV) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "device %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (!g_stream_ok[dev][which]) {
        // HIP multiplexes its streams onto a few hardware queues per priority class (4 by default); streams that share
        // a hardware queue run in order, so a short kernel of one block would wait behind every queued contraction of
        // another.  Two classes keep them apart: the X-engine's own work (staging stream + four contraction streams) at
        // normal priority, and everything the other blocks put on the GPU -- CorrAcc map, beamformer, span consumers and the
        // bulk copies (CorrAcc's 191 MB publish, the Copy block: a multi-millisecond copy at the head of a queue that a
        // contraction stream shares would stall the X-engine behind it) -- at high priority, one hardware queue each.
        int lo = 0, hi = 0;
        XENG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));       // numerically lower = higher priority
        const bool high = which == STREAM_MAP || which == STREAM_BEAM || which == STREAM_CONSUMER || which == STREAM_COPY;
        XENG_HIP(hipStreamCreateWithPriority(&g_streams[dev][which], hipStreamNonBlocking, high ? hi : 0 < lo ? 0 : lo));
        g_stream_ok[dev][which] = true;
    }
    *out = g_streams[dev][which];
    return XENG_STATUS_SUCCESS;
}

int sync_all_streams() {
    int dev = 0;
    XENG_HIP(hipGetDevice(&dev));
    for (int s = 0; s < STREAM_COUNT; s++) {
        hipStream_t st;
        bool ok;
        {
            std::lock_guard<std::mutex> lk(g_stream_mu);
            ok = g_stream_ok[dev][s];
            st = g_streams[dev][s];
        }
        if (ok) XENG_HIP(hipStreamSynchronize(st));
    }
    return XENG_STATUS_SUCCESS;
}

int EventTimer::begin(hipStream_t s, int k) {
    if (!enabled || npend >= MAXPEND) return -1;
    if (npend >= ncreated) {
        if (hipEventCreate(&start[ncreated]) != hipSuccess) return -1;
        if (hipEventCreate(&stop[ncreated]) != hipSuccess) return -1;
        ncreated++;
    }
    int slot = npend++;
    kind[slot] = k;
    (void)hipEventRecord(start[slot], s);
    return slot;
}
void EventTimer::end(hipStream_t s, int slot) {
    if (slot >= 0) (void)hipEventRecord(stop[slot], s);
}
int EventTimer::drain() {
    // consume the pairs that have completed; keep the rest pending (streams may still be running)
    int keep = 0;
    for (int i = 0; i < npend; i++) {
        float ms = 0;
        if (hipEventQuery(stop[i]) == hipSuccess && hipEventElapsedTime(&ms, start[i], stop[i]) == hipSuccess) {
            total_ms[kind[i]] += ms;
            count[kind[i]]++;
        } else {
            if (keep != i) {
                std::swap(start[keep], start[i]);
                std::swap(stop[keep], stop[i]);
                std::swap(kind[keep], kind[i]);
            }
            keep++;
        }
    }
    npend = keep;
    (void)hipGetLastError();   // hipEventQuery(not ready) is not an error
    return 0;
}
void EventTimer::destroy() {
    for (int i = 0; i < ncreated; i++) {
        (void)hipEventDestroy(start[i]);
        (void)hipEventDestroy(stop[i]);
    }
    ncreated = npend = 0;
}

}  // namespace xeng

using namespace xeng;

extern "C" {

const char* xengGetLastError(void) { return g_err; }
const char* xengVersion(void) { return "xeng-mi355x 0.1 (gfx950)"; }

int xengGetDeviceCount(int* count) {
    if (!count) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "null count");
    XENG_HIP(hipGetDeviceCount(count));
    return XENG_STATUS_SUCCESS;
}
int xengSetDevice(int gpu) {
    XENG_HIP(hipSetDevice(gpu));
    return XENG_STATUS_SUCCESS;
}
int xengGetDevice(int* gpu) {
    if (!gpu) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "null gpu");
    XENG_HIP(hipGetDevice(gpu));
    return XENG_STATUS_SUCCESS;
}
int xengDeviceSynchronize(void) {
    XENG_HIP(hipDeviceSynchronize());
    return XENG_STATUS_SUCCESS;
}
int xengGetDeviceInfo(int gpu, int* num_cu, int* clock_khz, size_t* total_mem, char* name, int name_len) {
    hipDeviceProp_t p;
    XENG_HIP(hipGetDeviceProperties(&p, gpu));
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (total_mem) *total_mem = p.totalGlobalMem;
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    return XENG_STATUS_SUCCESS;
}
int xengGetDevicePciBusId(int gpu, char* bus_id, int len) {
    if (!bus_id || len < 16) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "bus id buffer of at least 16 bytes needed");
    XENG_HIP(hipDeviceGetPCIBusId(bus_id, len, gpu));
    return XENG_STATUS_SUCCESS;
}
int xengMalloc(void** ptr, size_t nbytes, int space) {
    if (!ptr) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "null ptr");
    if (space == XENG_SPACE_CUDA) {
        XENG_HIP(hipMalloc(ptr, nbytes ? nbytes : 1));
    } else if (space == XENG_SPACE_CUDA_HOST) {
        XENG_HIP(hipHostMalloc(ptr, nbytes ? nbytes : 1, hipHostMallocDefault));
    } else {
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xengMalloc: space %d is not device or pinned host", space);
    }
    return XENG_STATUS_SUCCESS;
}
int xengFree(void* ptr, int space) {
    if (!ptr) return XENG_STATUS_SUCCESS;
    if (space == XENG_SPACE_CUDA) {
        XENG_HIP(hipFree(ptr));
    } else if (space == XENG_SPACE_CUDA_HOST) {
        XENG_HIP(hipHostFree(ptr));
    } else {
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "xengFree: bad space %d", space);
    }
    return XENG_STATUS_SUCCESS;
}
int xengMemcpy(void* dst, const void* src, size_t nbytes) {
    hipStream_t s;
    int rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    XENG_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDefault, s));
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
int xengMemcpyAsync(void* dst, const void* src, size_t nbytes) {
    hipStream_t s;
    int rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    XENG_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDefault, s));
    return XENG_STATUS_SUCCESS;
}
int xengMemset(void* dst, int value, size_t nbytes) {
    hipStream_t s;
    int rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    XENG_HIP(hipMemsetAsync(dst, value, nbytes, s));
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
int xengStreamSynchronize(void) { return sync_all_streams(); }

}  // extern "C"
