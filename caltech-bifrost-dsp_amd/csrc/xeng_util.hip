// Device / memory plumbing behind the C ABI (include/xeng.h, "device / memory plumbing").
// Replaces bifrost.device.set_device / stream_synchronize, BFArray(space='cuda'|'cuda_host')
// allocation and copy_array for the hot-path blocks.
#include <time.h>

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
void staging_stream_touched() {
    g_staging_ops.fetch_add(1, std::memory_order_relaxed);
    stream_tick(STREAM_XGPU);
}
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

// ---------------------------------------------------------------- stream clocks (xeng_common.h)
struct StreamClock {
    std::atomic<unsigned long long> enq{0};     // enqueues so far (ticked after each)
    std::mutex mu;
    unsigned long long done = 0;                // everything up to this tick is known to have completed
    // marks: events recorded on this stream with the tick they cover.  Sixty-four: two blocks that keep eight gulps in flight
    // each on the beam stream have sixteen of their own Mark events pending at any time; with eight slots the older half was
    // forgotten, a released span's stamp was then only covered by one of the NEWEST marks -- complete a whole queue (1 ms)
    // later -- and Beamform's reserve waited for span memory 157 times per 2000 gulps while the beam queue ran dry
    // (profiles/r04/blocks_gpu_idle.txt).
    static constexpr int NMARK = 64;
    struct Mark { unsigned long long upto = 0; hipEvent_t ev = nullptr; bool pending = false; hipEvent_t own = nullptr; } marks[NMARK];   // own: the slot's own event (ev may be a borrowed one)
};
static StreamClock g_clock[MAXDEV][STREAM_COUNT];

struct DeviceGuard {            // events are created and recorded with their device current; the caller's device is put back
    int prev = -1, want;
    explicit DeviceGuard(int d) : want(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};

void stream_tick(StreamId which) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return;
    g_clock[dev][which].enq.fetch_add(1, std::memory_order_release);
}

void stream_clocks_forget(int dev, StreamId which) {
    if (dev < 0 || dev >= MAXDEV) return;
    StreamClock& c = g_clock[dev][which];
    std::lock_guard<std::mutex> lk(c.mu);
    c.done = c.enq.load(std::memory_order_acquire);
    for (auto& m : c.marks) m.pending = false;          // (borrowed events may be destroyed by their owner after this)
}

unsigned long long stream_clock_now(StreamId which) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 0;
    return g_clock[dev][which].enq.load(std::memory_order_acquire);
}

void stream_clock_external_mark(StreamId which, hipEvent_t ev, unsigned long long upto) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return;
    StreamClock& c = g_clock[dev][which];
    std::lock_guard<std::mutex> lk(c.mu);
    if (upto <= c.done) return;
    StreamClock::Mark* slot = nullptr;
    for (auto& m : c.marks) {
        if (m.pending && m.ev == ev) { slot = &m; break; }      // the owner re-recorded this event: it now covers more
        if (!m.pending && !slot) slot = &m;
    }
    if (!slot) {                                         // all slots pending: the oldest borrowed one gives way (a later mark covers it)
        for (auto& m : c.marks)
            if (m.ev != m.own && (!slot || m.upto < slot->upto)) slot = &m;
        if (!slot) return;
    }
    slot->ev = ev;
    slot->upto = upto;
    slot->pending = true;
}

// Has stream (dev, which) passed tick t?  If not, *wait_ev is an event to wait for before asking again (recorded now when no
// pending mark covers t).  Never blocks.
static int clock_poll(int dev, StreamId which, unsigned long long t, bool* done, hipEvent_t* wait_ev) {
    *done = true;
    if (wait_ev) *wait_ev = nullptr;
    if (t == 0) return XENG_STATUS_SUCCESS;
    StreamClock& c = g_clock[dev][which];
    std::lock_guard<std::mutex> lk(c.mu);
    if (t <= c.done) return XENG_STATUS_SUCCESS;
    StreamClock::Mark* cover = nullptr;
    StreamClock::Mark* free_slot = nullptr;
    StreamClock::Mark* oldest = nullptr;
    // events of one stream complete in the order of their records: ask the oldest pending mark, go on while the answer is yes
    for (;;) {
        oldest = nullptr;
        for (auto& m : c.marks)
            if (m.pending && (!oldest || m.upto < oldest->upto)) oldest = &m;
        if (!oldest) break;
        const hipError_t e = hipEventQuery(oldest->ev);
        if (e == hipSuccess) {
            oldest->pending = false;
            if (oldest->upto > c.done) c.done = oldest->upto;
            continue;
        }
        if (e != hipErrorNotReady) XENG_HIP(e);
        (void)hipGetLastError();
        break;
    }
    for (auto& m : c.marks) {
        if (m.pending) {
            if (m.upto >= t && (!cover || m.upto < cover->upto)) cover = &m;
        } else if (!free_slot) {
            free_slot = &m;
        }
    }
    if (t <= c.done) return XENG_STATUS_SUCCESS;
    *done = false;
    if (!cover && free_slot) {
        hipStream_t s;
        {
            std::lock_guard<std::mutex> slk(g_stream_mu);
            if (!g_stream_ok[dev][which]) {        // (ticks without a stream cannot happen; nothing to wait for then)
                *done = true;
                return XENG_STATUS_SUCCESS;
            }
            s = g_streams[dev][which];
        }
        DeviceGuard g(dev);
        const unsigned long long upto = c.enq.load(std::memory_order_acquire);     // (read BEFORE the record: every enqueue counted here precedes it)
        if (!free_slot->own) XENG_HIP(hipEventCreateWithFlags(&free_slot->own, hipEventDisableTiming));
        free_slot->ev = free_slot->own;
        XENG_HIP(hipEventRecord(free_slot->ev, s));
        free_slot->upto = upto;
        free_slot->pending = true;
        cover = free_slot;
        if (hipEventQuery(cover->ev) == hipSuccess) {       // an idle stream: complete at once
            cover->pending = false;
            if (upto > c.done) c.done = upto;
            *done = true;
            return XENG_STATUS_SUCCESS;
        }
        (void)hipGetLastError();
    }
    if (wait_ev) *wait_ev = cover ? cover->ev : (oldest ? oldest->ev : nullptr);   // (all slots pending below t: wait for the oldest, ask again)
    return XENG_STATUS_SUCCESS;
}

static unsigned class_of(int stream) {
    switch (stream) {
        case STREAM_XGPU: return STAMP_XGPU;
        case STREAM_MAP: return STAMP_MAP;
        case STREAM_BEAM: return STAMP_BEAM;
        case STREAM_COPY: return STAMP_COPY;
        case STREAM_CONSUMER: return STAMP_CONSUMER;
        default: return 0;            // the contraction streams: by launch number (Stamp::xgpu_launch), never by a recorded event
    }
}

int stamp_now(Stamp* s, const void* buf, unsigned mask, int of_dev) {
    *s = Stamp();
    s->mask = mask;
    // of_dev: the device whose clocks are read (a ring buffer's own device: the thread that drops the last reference -- a helper
    // thread, a finaliser, teardown -- may have another device current, and that device's clocks say nothing about this memory)
    int dev = of_dev;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return XENG_STATUS_SUCCESS;                // no device: nothing can be in flight
    }
    if (dev < 0 || dev >= MAXDEV) return XENG_STATUS_SUCCESS;
    s->dev = dev;
    for (int k = 0; k < STREAM_COUNT; k++) s->clk[k] = g_clock[dev][k].enq.load(std::memory_order_acquire);
    xgpu_pending_launch(&s->xgpu_seq, &s->xgpu_epoch, &s->xgpu_launch, &s->xgpu_ctx);
    if ((mask & STAMP_XGPU_OUT) && !(mask & STAMP_XGPU) && buf) {       // a span contractions only write: its own last launch
        s->xgpu_launch = xgpu_last_writer(buf);
        s->xgpu_exact = true;
        s->xgpu_seq = 0;
    }
    return XENG_STATUS_SUCCESS;
}

// one pass over what the stamp waits for; *wait_ev: something to wait for before asking again (may be null: ask again later)
static int stamp_pass(const Stamp& s, bool* done, bool* waitable, hipEvent_t* wait_ev) {
    *done = true;
    if (waitable) *waitable = true;
    if (wait_ev) *wait_ev = nullptr;
    if (s.dev < 0) return XENG_STATUS_SUCCESS;
    for (int k = 0; k < STREAM_COUNT; k++) {
        if (!(class_of(k) & s.mask)) continue;
        bool d = true;
        hipEvent_t ev = nullptr;
        int rc = clock_poll(s.dev, (StreamId)k, s.clk[k], &d, wait_ev ? &ev : nullptr);
        if (rc) return rc;
        if (!d) {
            *done = false;
            if (wait_ev && !*wait_ev) *wait_ev = ev;
        }
    }
    if (s.mask & (STAMP_XGPU | STAMP_XGPU_OUT)) {
        bool d = true;
        hipEvent_t ev = nullptr;
        int rc = xgpu_launches_poll(s.xgpu_launch, s.xgpu_ctx, &d, &ev, s.xgpu_exact);
        if (rc) return rc;
        if (!d) {
            *done = false;
            if (wait_ev && !*wait_ev) *wait_ev = ev;
        }
        if (s.xgpu_seq) {
            bool launched = true;
            ev = nullptr;
            rc = xgpu_pending_poll(s.xgpu_seq, s.xgpu_epoch, &d, &launched, &ev, nullptr);
            if (rc) return rc;
            if (!d) {
                *done = false;
                if (wait_ev && !*wait_ev) *wait_ev = ev;
            }
            if (!launched && waitable) *waitable = false;
        }
    }
    return XENG_STATUS_SUCCESS;
}

int stamp_poll(const Stamp& s, bool* done, bool* waitable) { return stamp_pass(s, done, waitable, nullptr); }

int stamp_wait(const Stamp& s) {
    for (int spins = 0;; spins++) {
        bool done = true, waitable = true;
        hipEvent_t ev = nullptr;
        int rc = stamp_pass(s, &done, &waitable, &ev);
        if (rc) return rc;
        if (done) return XENG_STATUS_SUCCESS;
        if (ev) {
            XENG_HIP(hipEventSynchronize(ev));          // (outside every lock)
            continue;
        }
        // registered gulps whose contraction nobody has enqueued yet: only their owner can end this (a dump, or xengXgpuReset).
        // Not reached through the rings (they never wait for an unwaitable stamp).
        if (!waitable && spins > 40000) XENG_FAIL(XENG_STATUS_INVALID_STATE, "stamp: waiting for an X-engine launch that was never enqueued");
        struct timespec ts = {0, 50000};
        nanosleep(&ts, nullptr);
    }
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
    stream_tick(STREAM_COPY);
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
int xengMemcpyAsync(void* dst, const void* src, size_t nbytes) {
    hipStream_t s;
    int rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    XENG_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDefault, s));
    stream_tick(STREAM_COPY);
    return XENG_STATUS_SUCCESS;
}
int xengMemset(void* dst, int value, size_t nbytes) {
    hipStream_t s;
    int rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    XENG_HIP(hipMemsetAsync(dst, value, nbytes, s));
    stream_tick(STREAM_COPY);
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
int xengStreamSynchronize(void) { return sync_all_streams(); }

// words: 0 = device + 1 | class mask << 32; 1..5 the five non-contraction stream clocks; 6/7 pending launch + epoch; 8/9 launches + context
static const int PACKED_STREAMS[5] = {STREAM_XGPU, STREAM_MAP, STREAM_BEAM, STREAM_COPY, STREAM_CONSUMER};
static void stamp_pack(const Stamp& s, xengStamp* o) {
    memset(o, 0, sizeof(*o));
    o->w[0] = (unsigned long long)(s.dev + 1) | ((unsigned long long)s.mask << 32);
    for (int k = 0; k < 5; k++) o->w[1 + k] = s.clk[PACKED_STREAMS[k]];
    o->w[6] = s.xgpu_seq; o->w[7] = s.xgpu_epoch; o->w[8] = s.xgpu_launch; o->w[9] = s.xgpu_ctx; o->w[10] = s.xgpu_exact ? 1 : 0;
}
static void stamp_unpack(const xengStamp* o, Stamp* s) {
    *s = Stamp();
    s->dev = (int)(o->w[0] & 0xFFFFFFFFull) - 1;
    s->mask = (unsigned)(o->w[0] >> 32) & (STAMP_ALL | STAMP_XGPU_OUT);
    s->xgpu_exact = o->w[10] != 0;
    for (int k = 0; k < 5; k++) s->clk[PACKED_STREAMS[k]] = o->w[1 + k];
    s->xgpu_seq = o->w[6]; s->xgpu_epoch = o->w[7]; s->xgpu_launch = o->w[8]; s->xgpu_ctx = o->w[9];
}

int xengStampNow(xengStamp* stamp) {
    if (!stamp) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "StampNow: null stamp");
    Stamp s;
    int rc = stamp_now(&s);
    if (rc) return rc;
    stamp_pack(s, stamp);
    return XENG_STATUS_SUCCESS;
}
int xengStampNowFor(xengStamp* stamp, const void* buf, unsigned classes) {
    if (!stamp) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "StampNowFor: null stamp");
    Stamp s;
    int rc = stamp_now(&s, buf, classes ? (classes & (STAMP_ALL | STAMP_XGPU_OUT)) : (unsigned)STAMP_ALL);
    if (rc) return rc;
    stamp_pack(s, stamp);
    return XENG_STATUS_SUCCESS;
}
int xengStampDone(const xengStamp* stamp, int* done, int* waitable) {
    if (!stamp || !done) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "StampDone: null argument");
    Stamp s;
    stamp_unpack(stamp, &s);
    if (s.dev >= MAXDEV) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "StampDone: not a stamp");
    bool d = true, w = true;
    int rc = stamp_poll(s, &d, &w);
    if (rc) return rc;
    *done = d;
    if (waitable) *waitable = w;
    return XENG_STATUS_SUCCESS;
}
int xengStampWait(const xengStamp* stamp) {
    if (!stamp) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "StampWait: null stamp");
    Stamp s;
    stamp_unpack(stamp, &s);
    if (s.dev >= MAXDEV) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "StampWait: not a stamp");
    return stamp_wait(s);
}

}  // extern "C"
