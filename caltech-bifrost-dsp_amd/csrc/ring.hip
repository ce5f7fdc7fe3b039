// Span rings: the bookkeeping of the in-repo ring behind caltech-bifrost-dsp_amd/ring.py, native.
//
// The reference blocks sit on bifrost's C++ ring (lwa352-pipeline.py:147-155 creates them; the blocks use the protocol
// enumerated in SURVEY.md section 8b).  Round 3 kept that bookkeeping -- committed spans, reader cursors, back-pressure, the
// free list of span allocations -- in Python, where it cost every block 32-39 us of bytecode per gulp under the one interpreter
// lock.  Here it is plain structs behind a mutex and a condition variable; the Python classes are thin handles
// (one foreign call per gulp and side).
//
// Memory model (unlike bifrost's one circular buffer): every committed span is its own allocation with a reference count.
// References: the ring (while the span is committed and some reader still needs it), and every handle a reader or writer
// holds.  A reader that keeps its handle goes on reading the bytes after the ring has moved on -- Corr uses this to let
// the X-engine read gulps in place -- and the allocation returns to the ring's free list when the last reference is gone.
//
// Lifetime by construction (round 4; the round-3 fault is described in DESIGN.md 4.8): a released allocation is STAMPED with
// the library's stream clocks (xeng_common.h) and is handed out again -- or really freed -- only once every kernel that was
// enqueued before the release has completed.  Nothing in Python decides when device memory is freed.
#include <time.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "xeng_common.h"

namespace xeng {

struct RingCore;

struct Buf {
    std::atomic<long> refs{1};
    void* ptr = nullptr;
    size_t nbytes = 0;
    int space = XENG_SPACE_SYSTEM;
    int dev = -1;                    // device the allocation belongs to (device / pinned spaces): its stamps are taken and polled THERE,
                                     // whatever device is current on the thread that happens to drop the last reference
    bool owned = true;               // false: the caller's memory (xengRingCommitExternal), never freed or pooled here
    RingCore* ring = nullptr;        // (holds a reference on the ring)
    Stamp stamp;                     // library clocks at the moment the last user let go
    unsigned long long hook_stamp[2] = {0, 0};
};

struct Chunk {
    uint64_t offset;                 // byte offset in the sequence
    size_t nbytes;
    Buf* buf;
    size_t boff;                     // offset of the chunk's first byte in buf
};

struct Seq {
    long long index = 0, time_tag = 0;
    int nringlet = 1;
    std::string header;
    std::deque<Chunk> chunks;        // committed spans still held, in order
    uint64_t committed = 0;
    bool ended = false;
};

struct Reader {
    bool open = false, guarantee = true, in_seq = false;
    long long seq_index = 0;
    uint64_t offset = 0;
};

struct RingCore {
    std::atomic<long> refs{1};
    std::mutex mu;
    std::condition_variable cv;
    std::string name;
    int space = XENG_SPACE_SYSTEM;
    std::deque<std::unique_ptr<Seq>> seqs;
    long long seq_base = 0;          // index of seqs.front()
    long long nseq = 0;              // sequences begun so far
    Seq* open_seq = nullptr;
    bool writing_ended = false, destroyed = false;
    std::vector<Reader> readers;
    size_t capacity = 0, live_bytes = 0;
    // free list of span allocations (device / pinned spaces, and system space while stamp hooks are installed)
    std::mutex pool_mu;
    std::map<size_t, std::deque<Buf*>> pool;       // oldest release first
    std::vector<Buf*> graveyard;     // allocations over the pool limit: really freed by the next call that may block
    size_t pool_bytes = 0;
    // test hooks: another source of stamps (tests/: a fake backend's tickets) instead of the library's stream clocks
    xengRingStampNowFn hook_now = nullptr;
    xengRingStampDoneFn hook_done = nullptr;
    xengRingStampWaitFn hook_wait = nullptr;
    void* hook_user = nullptr;
    bool recycle_system = false;     // system space: recycle span memory like the device spaces do (default: fresh zeroed memory per span)
    unsigned stream_mask = 0;        // union of the stream classes the ring's users have declared (xengRingDeclareStreams)
    // A stamp is narrowed to stream_mask only while EVERY user of the ring has declared: `declared` counts declarations, `users`
    // counts the readers ever opened plus the writer (its first sequence).  One user that never declared -- a duck-typed block, a
    // test reader with kernels of its own -- and every stamp of the ring waits for all streams again.
    std::atomic<unsigned> declared{0}, users{0};
    bool writer_seen = false;        // (under mu)
    std::atomic<size_t> owned_bytes{0};      // span allocations the ring owns, in the free list or out
    // statistics
    std::atomic<unsigned long long> n_alloc{0}, n_free{0}, n_reuse{0}, n_stamp_wait{0}, n_skipped{0};
};

static void ring_unref(RingCore* r) {
    if (r->refs.fetch_sub(1) == 1) delete r;
}

// Span memory a ring may own (in its free list + handed out): eight times its capacity.  Released spans carry the stamp of
// everything their streams had queued at that moment, so the free list has to be deep enough for the oldest entry to have
// completed when it is needed again (bf-output at config 5: a new span every 77 us, stamps ~150 us deep, 10-14 spans out).
static size_t own_limit(const RingCore* r, size_t nbytes) { return 8 * std::max(r->capacity, 2 * nbytes); }

static bool pooled_space(const RingCore* r) { return r->space != XENG_SPACE_SYSTEM || r->hook_now || r->recycle_system; }

static int raw_alloc(int space, size_t nbytes, void** out) {
    const size_t n = nbytes ? nbytes : 1;
    if (space == XENG_SPACE_SYSTEM) {
        void* p = nullptr;
        if (posix_memalign(&p, 64, (n + 63) & ~(size_t)63) != 0 || !p) XENG_FAIL(XENG_STATUS_MEM_ALLOC_FAILED, "ring: out of host memory (%zu bytes)", n);
        memset(p, 0, n);
        *out = p;
        return XENG_STATUS_SUCCESS;
    }
    int rc = xengMalloc(out, n, space);
    if (rc) return rc;
    rc = xengMemset(*out, 0, n);                  // a first-time allocation reads as zeros (a recycled one holds what its last user left)
    if (rc) { (void)xengFree(*out, space); *out = nullptr; }
    return rc;
}

static void raw_free(int space, void* p) {
    if (!p) return;
    if (space == XENG_SPACE_SYSTEM) free(p);
    else (void)xengFree(p, space);
}

// the classes a stamp of this ring waits for: the declared union while every user has declared, else everything
static unsigned effective_mask(const RingCore* r) {
    const unsigned m = r->stream_mask;
    if (!m || r->declared.load() < r->users.load()) return (unsigned)STAMP_ALL;
    return m;
}

static void buf_stamp(Buf* b) {
    RingCore* r = b->ring;
    if (r->hook_now) r->hook_now(r->hook_user, b->hook_stamp);
    else if (b->space != XENG_SPACE_SYSTEM) {
        (void)stamp_now(&b->stamp, b->ptr, effective_mask(r), b->dev);
    }
}

// done / waitable of a released allocation's stamp; never blocks
static void buf_poll(Buf* b, bool* done, bool* waitable) {
    RingCore* r = b->ring;
    *done = true;
    *waitable = true;
    if (r->hook_done) {
        *done = r->hook_done(r->hook_user, b->hook_stamp) != 0;
        return;
    }
    if (b->space == XENG_SPACE_SYSTEM) return;
    if (stamp_poll(b->stamp, done, waitable) != XENG_STATUS_SUCCESS) { *done = false; *waitable = false; }   // (cannot tell: treat as busy)
}

static int buf_wait(Buf* b) {
    RingCore* r = b->ring;
    r->n_stamp_wait++;
    if (r->hook_wait) { r->hook_wait(r->hook_user, b->hook_stamp); return XENG_STATUS_SUCCESS; }
    if (b->space == XENG_SPACE_SYSTEM) return XENG_STATUS_SUCCESS;
    return stamp_wait(b->stamp);
}

// really free an allocation the ring owns: only behind its stamp (the kernels enqueued before its release have completed)
static void buf_destroy(Buf* b, bool may_wait) {
    RingCore* r = b->ring;
    if (b->owned && b->ptr) {
        bool done = true, waitable = true;
        buf_poll(b, &done, &waitable);
        if (!done && may_wait && waitable) done = buf_wait(b) == XENG_STATUS_SUCCESS;
        if (done) {
            raw_free(b->space, b->ptr);
            r->n_free++;
        }
        // (not done and not waitable: the allocation is leaked rather than freed under a kernel that may still use it)
        r->owned_bytes -= b->nbytes;
    }
    delete b;
    ring_unref(r);
}

static void buf_release(Buf* b) {
    if (b->refs.fetch_sub(1) != 1) return;
    RingCore* r = b->ring;
    if (!b->owned) { delete b; ring_unref(r); return; }
    if (!pooled_space(r)) { raw_free(b->space, b->ptr); r->n_free++; r->owned_bytes -= b->nbytes; delete b; ring_unref(r); return; }
    buf_stamp(b);
    {
        std::lock_guard<std::mutex> lk(r->pool_mu);
        if (!r->destroyed) {
            // one bound for what the ring may own (own_limit): an allocation is really freed only past it, so a ring in
            // steady state neither allocates nor frees (hipFree synchronises the device)
            if (r->owned_bytes.load() > own_limit(r, b->nbytes)) {
                r->graveyard.push_back(b);      // freed by a later call that may block (hipFree synchronises the device)
            } else {
                r->pool[b->nbytes].push_back(b);
                r->pool_bytes += b->nbytes;
            }
            return;
        }
    }
    buf_destroy(b, true);                       // the ring is gone: nobody will take the allocation over
}

static void bury(RingCore* r) {
    std::vector<Buf*> dead;
    {
        std::lock_guard<std::mutex> lk(r->pool_mu);
        dead.swap(r->graveyard);
    }
    for (Buf* b : dead) buf_destroy(b, true);
}

// An allocation of `nbytes` for a new span: the oldest released one whose stamp is complete, else wait for the oldest (when
// the caller may block), else a fresh one.  XENG_STATUS_WOULD_BLOCK when may_block is 0 and the call would have to wait or
// to allocate device memory.
static int buf_obtain(RingCore* r, size_t nbytes, int may_block, Buf** out) {
    *out = nullptr;
    if (pooled_space(r)) {
        // the oldest release first: its stamp is the most likely to be complete (the stamps are polled outside the list's lock)
        Buf* cand = nullptr;
        {
            std::lock_guard<std::mutex> lk(r->pool_mu);
            auto it = r->pool.find(nbytes);
            if (it != r->pool.end() && !it->second.empty()) {
                cand = it->second.front();
                it->second.pop_front();
                r->pool_bytes -= nbytes;
            }
        }
        if (cand) {
            bool done, waitable;
            buf_poll(cand, &done, &waitable);
            // still busy: a fresh allocation while the ring owns little (a deeper free list costs memory once; a wait costs
            // every gulp), else wait for it (kernels of other blocks, enqueued before the release)
            const bool grow = !done && r->owned_bytes.load() + nbytes <= own_limit(r, nbytes) && !r->hook_now;
            if (!done && waitable && may_block && !grow) done = buf_wait(cand) == XENG_STATUS_SUCCESS;
            if (done) {
                *out = cand;
            } else {
                std::lock_guard<std::mutex> lk(r->pool_mu);
                if (waitable) r->pool[nbytes].push_front(cand);       // still the next one to be reissued
                else r->pool[nbytes].push_back(cand);                 // waits for a launch nobody has enqueued: try the others first
                r->pool_bytes += nbytes;
                if (waitable && !may_block && !grow) return XENG_STATUS_WOULD_BLOCK;
            }
        }
        if (*out) {
            (*out)->refs.store(1);
            r->n_reuse++;
            return XENG_STATUS_SUCCESS;
        }
    }
    if (!may_block && (r->space != XENG_SPACE_SYSTEM || nbytes > (1u << 20))) return XENG_STATUS_WOULD_BLOCK;
    void* p = nullptr;
    int rc = raw_alloc(r->space, nbytes, &p);
    if (rc) return rc;
    Buf* b = new Buf();
    b->ptr = p; b->nbytes = nbytes; b->space = r->space; b->ring = r;
    if (r->space != XENG_SPACE_SYSTEM && hipGetDevice(&b->dev) != hipSuccess) { (void)hipGetLastError(); b->dev = -1; }
    r->refs.fetch_add(1);
    r->n_alloc++;
    r->owned_bytes += nbytes;
    *out = b;
    return XENG_STATUS_SUCCESS;
}

// ---- bookkeeping (hold r->mu)
static Seq* seq_at(RingCore* r, long long index) {
    if (index < r->seq_base || index >= r->seq_base + (long long)r->seqs.size()) return nullptr;
    return r->seqs[(size_t)(index - r->seq_base)].get();
}

static void drop_chunk(RingCore* r, Seq* s, std::vector<Buf*>* released) {
    Chunk c = s->chunks.front();
    s->chunks.pop_front();
    r->live_bytes -= c.nbytes;
    released->push_back(c.buf);
}

// free committed spans every open reader has moved past; sequences all readers have left are forgotten
static void gc(RingCore* r, std::vector<Buf*>* released) {
    bool any = false;
    long long lo_seq = 0;
    uint64_t lo_off = 0;
    for (const Reader& rd : r->readers) {
        if (!rd.open) continue;
        if (!any || rd.seq_index < lo_seq || (rd.seq_index == lo_seq && rd.offset < lo_off)) { lo_seq = rd.seq_index; lo_off = rd.offset; }
        any = true;
    }
    if (!any) return;
    while (!r->seqs.empty() && r->seq_base < lo_seq) {
        Seq* s = r->seqs.front().get();
        while (!s->chunks.empty()) drop_chunk(r, s, released);
        if (s == r->open_seq) break;           // (cannot happen: a reader never passes an open sequence)
        r->seqs.pop_front();
        r->seq_base++;
    }
    if (Seq* s = seq_at(r, lo_seq))
        while (!s->chunks.empty() && s->chunks.front().offset + s->chunks.front().nbytes <= lo_off) drop_chunk(r, s, released);
}

static bool drop_oldest(RingCore* r, std::vector<Buf*>* released) {
    for (auto& sp : r->seqs)
        if (!sp->chunks.empty()) { drop_chunk(r, sp.get(), released); return true; }
    r->live_bytes = 0;
    return false;
}

static void release_all(std::vector<Buf*>& v) {
    for (Buf* b : v) buf_release(b);
    v.clear();
}

// room for nbytes more committed bytes; returns WOULD_BLOCK instead of sleeping when may_block is 0
static int wait_for_room(RingCore* r, std::unique_lock<std::mutex>& lk, size_t nbytes, int nonblocking, int may_block) {
    if (r->capacity == 0) r->capacity = 4 * nbytes;
    std::vector<Buf*> released;
    for (;;) {
        const size_t cap = std::max(r->capacity, nbytes);
        if (r->live_bytes + nbytes <= cap) break;
        gc(r, &released);
        if (r->live_bytes + nbytes <= cap) break;
        bool guaranteed = false;
        for (const Reader& rd : r->readers) guaranteed |= rd.open && rd.guarantee;
        if (!guaranteed) {                      // nobody applies back-pressure: overwrite the oldest span, like bifrost
            if (!drop_oldest(r, &released)) break;
            continue;
        }
        if (nonblocking || !may_block || r->destroyed) {
            lk.unlock(); release_all(released); lk.lock();
            if (r->destroyed) XENG_FAIL(XENG_STATUS_INVALID_STATE, "ring '%s' was destroyed", r->name.c_str());
            if (nonblocking) XENG_FAIL(XENG_STATUS_WOULD_BLOCK, "ring '%s' full", r->name.c_str());
            return XENG_STATUS_WOULD_BLOCK;
        }
        if (!released.empty()) { lk.unlock(); release_all(released); lk.lock(); continue; }
        r->cv.wait_for(lk, std::chrono::milliseconds(100));
    }
    if (!released.empty()) { lk.unlock(); release_all(released); lk.lock(); }
    return XENG_STATUS_SUCCESS;
}

static int copy_bytes(int space, void* dst, const void* src, size_t n) {
    if (space == XENG_SPACE_SYSTEM) { memcpy(dst, src, n); return XENG_STATUS_SUCCESS; }
    return xengMemcpy(dst, src, n);
}

}  // namespace xeng

using namespace xeng;

#define RING_ARG(r_) \
    RingCore* r = (RingCore*)(r_); \
    if (!r) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "ring: null handle")

extern "C" {

int xengRingCreate(xengRing** ring, const char* name, int space) {
    if (!ring) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "RingCreate: null handle pointer");
    if (space != XENG_SPACE_SYSTEM && space != XENG_SPACE_CUDA && space != XENG_SPACE_CUDA_HOST)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "RingCreate: unknown space %d", space);
    RingCore* r = new RingCore();
    r->name = name ? name : "";
    r->space = space;
    *ring = (xengRing*)r;
    return XENG_STATUS_SUCCESS;
}

int xengRingDestroy(xengRing* ring) {
    RING_ARG(ring);
    std::vector<Buf*> released;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (r->destroyed) return XENG_STATUS_SUCCESS;
        r->destroyed = true;
        r->writing_ended = true;
        for (auto& sp : r->seqs) {
            sp->ended = true;
            while (!sp->chunks.empty()) drop_chunk(r, sp.get(), &released);
        }
        r->open_seq = nullptr;
        r->cv.notify_all();
    }
    release_all(released);
    std::vector<Buf*> dead;
    {
        std::lock_guard<std::mutex> lk(r->pool_mu);
        for (auto& kv : r->pool)
            for (Buf* b : kv.second) dead.push_back(b);
        r->pool.clear();
        r->pool_bytes = 0;
        for (Buf* b : r->graveyard) dead.push_back(b);
        r->graveyard.clear();
    }
    for (Buf* b : dead) buf_destroy(b, true);
    ring_unref(r);                     // (spans still referenced by handles keep the core alive until they are released)
    return XENG_STATUS_SUCCESS;
}

int xengRingSetStampHooks(xengRing* ring, xengRingStampNowFn now, xengRingStampDoneFn done, xengRingStampWaitFn wait, void* user) {
    RING_ARG(ring);
    std::lock_guard<std::mutex> lk(r->pool_mu);
    r->hook_now = now; r->hook_done = done; r->hook_wait = wait; r->hook_user = user;
    return XENG_STATUS_SUCCESS;
}

int xengRingDeclareStreams(xengRing* ring, unsigned classes) {
    RING_ARG(ring);
    std::lock_guard<std::mutex> lk(r->pool_mu);
    r->stream_mask |= classes & (STAMP_ALL | STAMP_XGPU_OUT);
    r->declared.fetch_add(1);          // one declaration per user (classes may be 0: a user that enqueues nothing on the spans)
    return XENG_STATUS_SUCCESS;
}

int xengRingGetStampClasses(xengRing* ring, unsigned* classes, unsigned* declared, unsigned* users) {
    RING_ARG(ring);
    if (classes) *classes = effective_mask(r);
    if (declared) *declared = r->declared.load();
    if (users) *users = r->users.load();
    return XENG_STATUS_SUCCESS;
}

int xengRingSetRecycle(xengRing* ring, int on) {
    RING_ARG(ring);
    std::lock_guard<std::mutex> lk(r->pool_mu);
    r->recycle_system = on != 0;
    return XENG_STATUS_SUCCESS;
}

int xengRingResize(xengRing* ring, size_t contig_bytes, size_t total_span) {
    RING_ARG(ring);
    const size_t want = total_span ? total_span : 4 * contig_bytes;
    std::lock_guard<std::mutex> lk(r->mu);
    r->capacity = std::max(r->capacity, std::max(want, contig_bytes));
    return XENG_STATUS_SUCCESS;
}

int xengRingGetInfo(xengRing* ring, size_t* capacity, size_t* live_bytes, size_t* pool_bytes, int* nreaders, long long* nseq,
                    unsigned long long counters[5]) {
    RING_ARG(ring);
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (capacity) *capacity = r->capacity;
        if (live_bytes) *live_bytes = r->live_bytes;
        if (nreaders) {
            int n = 0;
            for (const Reader& rd : r->readers) n += rd.open;
            *nreaders = n;
        }
        if (nseq) *nseq = r->nseq;
    }
    if (pool_bytes) {
        std::lock_guard<std::mutex> lk(r->pool_mu);
        *pool_bytes = r->pool_bytes;
    }
    if (counters) {
        counters[0] = r->n_alloc; counters[1] = r->n_free; counters[2] = r->n_reuse; counters[3] = r->n_stamp_wait; counters[4] = r->n_skipped;
    }
    return XENG_STATUS_SUCCESS;
}

// ---------------------------------------------------------------- writer
int xengRingBeginSequence(xengRing* ring, long long time_tag, const void* header, size_t header_len, int nringlet, long long* seq) {
    RING_ARG(ring);
    if (!seq) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "BeginSequence: null seq");
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->destroyed) XENG_FAIL(XENG_STATUS_INVALID_STATE, "ring '%s' was destroyed", r->name.c_str());
    if (r->open_seq) r->open_seq->ended = true;
    if (!r->writer_seen) { r->writer_seen = true; r->users.fetch_add(1); }
    std::unique_ptr<Seq> s(new Seq());
    s->index = r->nseq++;
    s->time_tag = time_tag;
    s->nringlet = nringlet;
    if (header && header_len) s->header.assign((const char*)header, header_len);
    r->open_seq = s.get();
    *seq = s->index;
    if (r->seqs.empty()) r->seq_base = s->index;
    r->seqs.push_back(std::move(s));
    r->cv.notify_all();
    return XENG_STATUS_SUCCESS;
}

int xengRingEndSequence(xengRing* ring, long long seq) {
    RING_ARG(ring);
    std::lock_guard<std::mutex> lk(r->mu);
    if (Seq* s = seq_at(r, seq)) {
        s->ended = true;
        if (r->open_seq == s) r->open_seq = nullptr;
    }
    r->cv.notify_all();
    return XENG_STATUS_SUCCESS;
}

int xengRingEndWriting(xengRing* ring) {
    RING_ARG(ring);
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->open_seq) { r->open_seq->ended = true; r->open_seq = nullptr; }
    r->writing_ended = true;
    r->cv.notify_all();
    return XENG_STATUS_SUCCESS;
}

int xengRingReserve(xengRing* ring, long long seq, size_t nbytes, int nonblocking, int may_block, void** data, long long* span) {
    RING_ARG(ring);
    if (!data || !span) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Reserve: null output");
    if (may_block) bury(r);
    {
        std::unique_lock<std::mutex> lk(r->mu);
        Seq* s = seq < 0 ? r->open_seq : seq_at(r, seq);
        if (!s || s->ended) XENG_FAIL(XENG_STATUS_INVALID_STATE, "WriteSpan: no open sequence on ring '%s'", r->name.c_str());
        int rc = wait_for_room(r, lk, nbytes, nonblocking, may_block);
        if (rc) return rc;
    }
    Buf* b = nullptr;
    int rc = buf_obtain(r, nbytes, may_block, &b);
    if (rc) return rc;
    *data = b->ptr;
    *span = (long long)(intptr_t)b;
    return XENG_STATUS_SUCCESS;
}

int xengRingCommit(xengRing* ring, long long seq, long long span, size_t nbytes) {
    RING_ARG(ring);
    Buf* b = (Buf*)(intptr_t)span;
    if (!b || b->ring != r) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Commit: not a span of ring '%s'", r->name.c_str());
    if (nbytes > b->nbytes) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Commit: %zu bytes of a %zu-byte span", nbytes, b->nbytes);
    if (nbytes == 0) return XENG_STATUS_SUCCESS;
    std::lock_guard<std::mutex> lk(r->mu);
    Seq* s = seq < 0 ? r->open_seq : seq_at(r, seq);
    if (!s) return XENG_STATUS_SUCCESS;         // every reader has left that sequence: nobody can see the span
    b->refs.fetch_add(1);
    s->chunks.push_back(Chunk{s->committed, nbytes, b, 0});
    s->committed += nbytes;
    r->live_bytes += nbytes;
    r->cv.notify_all();
    return XENG_STATUS_SUCCESS;
}

int xengRingCommitExternal(xengRing* ring, long long seq, void* data, size_t nbytes, int may_block) {
    RING_ARG(ring);
    if (!data || nbytes == 0) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "CommitExternal: empty span");
    std::unique_lock<std::mutex> lk(r->mu);
    Seq* s = seq < 0 ? r->open_seq : seq_at(r, seq);
    if (!s || s->ended) XENG_FAIL(XENG_STATUS_INVALID_STATE, "commit_external: no open sequence on ring '%s'", r->name.c_str());
    int rc = wait_for_room(r, lk, nbytes, 0, may_block);
    if (rc) return rc;
    s = seq < 0 ? r->open_seq : seq_at(r, seq);
    if (!s) XENG_FAIL(XENG_STATUS_INVALID_STATE, "commit_external: the sequence is gone (ring '%s')", r->name.c_str());
    Buf* b = new Buf();
    b->ptr = data; b->nbytes = nbytes; b->space = r->space; b->owned = false; b->ring = r;
    r->refs.fetch_add(1);
    s->chunks.push_back(Chunk{s->committed, nbytes, b, 0});       // (the ring's reference is the only one)
    s->committed += nbytes;
    r->live_bytes += nbytes;
    r->cv.notify_all();
    return XENG_STATUS_SUCCESS;
}

// ---------------------------------------------------------------- reader
int xengRingOpenReader(xengRing* ring, int guarantee, int* reader) {
    RING_ARG(ring);
    if (!reader) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "OpenReader: null reader");
    std::lock_guard<std::mutex> lk(r->mu);
    size_t k = 0;
    while (k < r->readers.size() && r->readers[k].open) k++;
    if (k == r->readers.size()) r->readers.emplace_back();
    Reader& rd = r->readers[k];
    rd = Reader();
    rd.open = true;
    r->users.fetch_add(1);
    rd.guarantee = guarantee != 0;
    // A reader that registers late starts at the oldest sequence that still holds data (or is still being written), as
    // a bifrost reader opens the earliest sequence in the ring -- never at data that is gone.
    rd.seq_index = r->nseq;
    for (auto& sp : r->seqs)
        if (!sp->chunks.empty() || !sp->ended) { rd.seq_index = sp->index; break; }
    *reader = (int)k;
    return XENG_STATUS_SUCCESS;
}

int xengRingCloseReader(xengRing* ring, int reader) {
    RING_ARG(ring);
    std::vector<Buf*> released;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (reader < 0 || reader >= (int)r->readers.size()) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "CloseReader: bad reader %d", reader);
        r->readers[reader].open = false;
        gc(r, &released);
        r->cv.notify_all();
    }
    release_all(released);
    return XENG_STATUS_SUCCESS;
}

int xengRingNextSequence(xengRing* ring, int reader, int may_block, long long* seq, long long* time_tag, int* nringlet,
                         const void** header, size_t* header_len) {
    RING_ARG(ring);
    std::vector<Buf*> released;
    int rc = XENG_STATUS_SUCCESS;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (reader < 0 || reader >= (int)r->readers.size() || !r->readers[reader].open)
            XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "NextSequence: bad reader %d", reader);
        Reader* rd = &r->readers[reader];
        if (rd->in_seq) {
            rd->in_seq = false;
            rd->seq_index++;
            rd->offset = 0;
            gc(r, &released);
            r->cv.notify_all();
        }
        for (;;) {
            rd = &r->readers[reader];
            if (rd->seq_index < r->seq_base) rd->seq_index = r->seq_base;       // (sequences that are gone)
            if (rd->seq_index < r->nseq) break;
            if (r->writing_ended || r->destroyed) { rc = XENG_STATUS_END_OF_DATA; break; }
            if (!may_block) { rc = XENG_STATUS_WOULD_BLOCK; break; }
            r->cv.wait_for(lk, std::chrono::milliseconds(100));
        }
        if (rc == XENG_STATUS_SUCCESS) {
            Seq* s = seq_at(r, rd->seq_index);
            rd->in_seq = true;
            rd->offset = 0;
            if (seq) *seq = s->index;
            if (time_tag) *time_tag = s->time_tag;
            if (nringlet) *nringlet = s->nringlet;
            if (header) *header = s->header.data();          // (valid until this reader asks for its next sequence)
            if (header_len) *header_len = s->header.size();
        }
    }
    release_all(released);
    return rc;
}

// max_parts: a gulp that lies in up to that many committed spans is returned as windows on them (data / nbytes / span arrays
// of that size, *nparts filled in); one that needs more pieces is gathered into one copy as before
static int acquire_common(xengRing* ring, int reader, size_t advance, size_t gulp_nbytes, int may_block, int max_parts, void** data,
                          size_t* nbytes, long long* span, int* nparts, size_t* skipped) {
    RING_ARG(ring);
    if (!data || !nbytes || !span || gulp_nbytes == 0) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Acquire: bad argument");
    std::vector<Buf*> released;
    std::vector<Chunk> pieces;
    int rc = XENG_STATUS_SUCCESS;
    size_t n = 0, skip_total = 0;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (reader < 0 || reader >= (int)r->readers.size() || !r->readers[reader].open || !r->readers[reader].in_seq)
            XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Acquire: reader %d has no open sequence", reader);
        if (advance) {
            r->readers[reader].offset += advance;
            gc(r, &released);
            r->cv.notify_all();
        }
        for (;;) {
            Reader& rd = r->readers[reader];
            Seq* s = seq_at(r, rd.seq_index);
            if (!s) { rc = XENG_STATUS_END_OF_DATA; break; }
            // data that was overwritten before this reader got to it (no guaranteed reader held it, or this reader
            // registered late): skip ahead by whole gulps to the oldest span still there, as a bifrost reader does
            const uint64_t lo = s->chunks.empty() ? s->committed : s->chunks.front().offset;
            if (rd.offset < lo) {
                const uint64_t sk = ((lo - rd.offset + gulp_nbytes - 1) / gulp_nbytes) * gulp_nbytes;
                rd.offset += sk;
                skip_total += sk;
                r->n_skipped += sk;
            }
            const uint64_t avail = s->committed > rd.offset ? s->committed - rd.offset : 0;
            if (avail >= gulp_nbytes || s->ended || r->destroyed) {
                n = (size_t)std::min<uint64_t>(avail, gulp_nbytes);
                if (n == 0) { rc = XENG_STATUS_END_OF_DATA; break; }
                for (const Chunk& c : s->chunks) {
                    if (c.offset >= rd.offset + n) break;
                    const uint64_t a = std::max<uint64_t>(rd.offset, c.offset), b = std::min<uint64_t>(rd.offset + n, c.offset + c.nbytes);
                    if (a < b) {
                        c.buf->refs.fetch_add(1);
                        pieces.push_back(Chunk{a, (size_t)(b - a), c.buf, c.boff + (size_t)(a - c.offset)});
                    }
                }
                break;
            }
            if (!may_block) { rc = XENG_STATUS_WOULD_BLOCK; break; }
            r->cv.wait_for(lk, std::chrono::milliseconds(100));
        }
    }
    release_all(released);
    if (skipped) *skipped = skip_total;
    if (rc) return rc;
    if ((int)pieces.size() <= max_parts) {          // the usual case: the gulp lies inside one committed span (or max_parts of them) -- windows, no copy
        for (size_t k = 0; k < pieces.size(); k++) {
            data[k] = (uint8_t*)pieces[k].buf->ptr + pieces[k].boff;
            nbytes[k] = pieces[k].nbytes;
            span[k] = (long long)(intptr_t)pieces[k].buf;
        }
        if (nparts) *nparts = (int)pieces.size();
        return XENG_STATUS_SUCCESS;
    }
    if (!may_block) {
        // the gathered copy may wait for span memory and copies synchronously: not with the caller's interpreter lock held.  The
        // cursor has moved on already; the caller asks again with advance = 0 and may_block = 1 (xfast.cpp: ask first, then wait).
        for (Chunk& c : pieces) buf_release(c.buf);
        return XENG_STATUS_WOULD_BLOCK;
    }
    *nbytes = n;
    if (nparts) *nparts = 1;
    // gathered copy in the ring's space
    Buf* g = nullptr;
    rc = buf_obtain(r, n, 1, &g);
    size_t pos = 0;
    for (Chunk& c : pieces) {
        if (!rc) rc = copy_bytes(r->space, (uint8_t*)g->ptr + pos, (const uint8_t*)c.buf->ptr + c.boff, c.nbytes);
        pos += c.nbytes;
        buf_release(c.buf);
    }
    if (rc) { if (g) buf_release(g); return rc; }
    *data = g->ptr;
    *span = (long long)(intptr_t)g;
    return XENG_STATUS_SUCCESS;
}

int xengRingAcquire(xengRing* ring, int reader, size_t advance, size_t gulp_nbytes, int may_block, void** data, size_t* nbytes,
                    long long* span, size_t* skipped) {
    return acquire_common(ring, reader, advance, gulp_nbytes, may_block, 1, data, nbytes, span, nullptr, skipped);
}

int xengRingAcquireParts(xengRing* ring, int reader, size_t advance, size_t gulp_nbytes, int may_block, void* data[2], size_t nbytes[2],
                         long long span[2], int* nparts, size_t* skipped) {
    if (!nparts) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "AcquireParts: null nparts");
    return acquire_common(ring, reader, advance, gulp_nbytes, may_block, 2, data, nbytes, span, nparts, skipped);
}

int xengRingSpanRelease(long long span) {
    Buf* b = (Buf*)(intptr_t)span;
    if (!b) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "SpanRelease: null span");
    buf_release(b);
    return XENG_STATUS_SUCCESS;
}

}  // extern "C"
