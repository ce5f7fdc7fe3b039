// Stress harness for the span-ring C ABI (include/xeng.h "span rings"; csrc/ring.hip + csrc/xeng_util.hip host code), built
// by tests/test_host_sanitizers.py with -fsanitize=thread and again with -fsanitize=address,undefined.  No GPU: system-space
// rings, with stamp hooks standing in for the library's stream clocks (a "GPU" thread that completes tickets with a lag).
//
// What the reference leans on bifrost's mature native ring for (block_base.py:149,229-255 is the only locking its blocks
// need), this repo implements itself -- refcounts, two mutexes, a condition variable, a stamped free list -- so it is run
// under the tools that find lifetime and ordering bugs:
//   scenario 1  writer + guaranteed reader that HOLDS spans + lossy reader + a late reader, several sequences, stamp hooks with a
//               lagging completer: no gulp lost or reordered for the guaranteed reader, every byte as written, nothing reissued
//               under an incomplete stamp, everything returned at the end;
//   scenario 2  the same on a recycling system ring without hooks (xengRingSetRecycle) and with two-part acquires;
//   scenario 3  xengRingDestroy while a reader sleeps in Acquire and while another still holds spans;
//   scenario 4  a span released from a foreign thread; declarations vs users (xengRingGetStampClasses);
//   scenario 5  gathered reads asked with may_block = 0 return WOULD_BLOCK and lose nothing.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <thread>
#include <vector>

#include "../../include/xeng.h"

// ---- the X-engine side of a stamp (csrc/xcorr.hip) is not linked: no launches exist here
#include <hip/hip_runtime.h>
namespace xeng {
void xgpu_pending_launch(unsigned long long* seq, unsigned long long* epoch, unsigned long long* nlaunch, unsigned long long* ctx) {
    *seq = *epoch = *nlaunch = *ctx = 0;
}
int xgpu_pending_poll(unsigned long long, unsigned long long, bool* done, bool* launched, hipEvent_t* ev, int*) {
    *done = true; *launched = true; if (ev) *ev = nullptr; return 0;
}
int xgpu_launches_poll(unsigned long long, unsigned long long, bool* done, hipEvent_t* ev, bool) { *done = true; if (ev) *ev = nullptr; return 0; }
unsigned long long xgpu_last_writer(const void*) { return 0; }
}  // namespace xeng

#define CHECK(cond, ...)                                                          \
    do {                                                                          \
        if (!(cond)) {                                                            \
            fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond);   \
            fprintf(stderr, __VA_ARGS__);                                         \
            fprintf(stderr, " (last error: %s)\n", xengGetLastError());           \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

static void nap_us(int us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); }

// ---- a fake GPU: tickets issued at "enqueue", completed by a thread that lags behind
struct FakeGpu {
    std::atomic<unsigned long long> issued{0}, completed{0};
    std::atomic<bool> stop{false};
    std::atomic<unsigned long long> reissued_under_incomplete{0}, waits{0};
    void run() {
        while (!stop.load()) {
            const unsigned long long i = issued.load(), c = completed.load();
            if (c < i) completed.store(c + 1);           // one ticket per tick: behind whenever the threads enqueue faster, never stuck
            nap_us(20);
        }
        completed.store(issued.load());
    }
};
static void hook_now(void* u, unsigned long long st[2]) { st[0] = ((FakeGpu*)u)->issued.load(); st[1] = 0; }
static int hook_done(void* u, const unsigned long long st[2]) { return st[0] <= ((FakeGpu*)u)->completed.load(); }
static void hook_wait(void* u, const unsigned long long st[2]) {
    FakeGpu* g = (FakeGpu*)u;
    g->waits++;
    while (g->completed.load() < st[0]) {
        if (g->stop.load()) g->completed.store(g->issued.load());
        nap_us(10);
    }
}

static const size_t GULP = 4096;

static void fill(uint8_t* p, size_t n, uint32_t seq, uint32_t k) {
    uint32_t* w = (uint32_t*)p;
    for (size_t i = 0; i < n / 4; i++) w[i] = (seq << 24) ^ (k << 8) ^ (uint32_t)i;
}
static bool verify(const uint8_t* p, size_t n, uint32_t seq, uint32_t k, size_t first_word = 0) {
    const uint32_t* w = (const uint32_t*)p;
    for (size_t i = 0; i < n / 4; i++)
        if (w[i] != ((seq << 24) ^ (k << 8) ^ (uint32_t)(first_word + i))) return false;
    return true;
}

struct Result { long long gulps = 0, skipped = 0, seqs = 0; };

// reads every sequence to its end; `hold` spans are kept referenced behind the cursor; `lossy` naps now and then
static Result reader_thread(xengRing* r, int rid, int hold, bool lossy, FakeGpu* gpu, bool strict, bool two_parts) {
    Result res;
    std::deque<long long> held;
    for (;;) {
        long long seq = 0, tag = 0;
        int nringlet = 0;
        const void* hdr = nullptr;
        size_t hlen = 0;
        int rc = xengRingNextSequence(r, rid, 1, &seq, &tag, &nringlet, &hdr, &hlen);
        if (rc == XENG_STATUS_END_OF_DATA) break;
        CHECK(rc == 0, "NextSequence rc %d", rc);
        CHECK(hlen == 8 && memcmp(hdr, "hdr", 3) == 0, "header");
        res.seqs++;
        size_t advance = 0;
        uint64_t offset = 0;
        const size_t want = two_parts ? 2 * GULP : GULP;
        for (;;) {
            void* data[2] = {nullptr, nullptr};
            size_t n[2] = {0, 0}, skipped = 0;
            long long span[2] = {0, 0};
            int nparts = 1;
            if (two_parts) rc = xengRingAcquireParts(r, rid, advance, want, 1, data, n, span, &nparts, &skipped);
            else rc = xengRingAcquire(r, rid, advance, want, 1, &data[0], &n[0], &span[0], &skipped);
            if (rc == XENG_STATUS_END_OF_DATA) break;
            CHECK(rc == 0, "Acquire rc %d", rc);
            if (strict) CHECK(skipped == 0, "the guaranteed reader lost %zu bytes", skipped);
            CHECK(skipped % want == 0, "skipped %zu is not whole gulps", skipped);
            offset += skipped;
            res.skipped += (long long)skipped;
            size_t got = 0;
            for (int k = 0; k < nparts; k++) {
                const uint64_t o = offset + got;
                CHECK(n[k] % 4 == 0, "part size");
                // the bytes are what the writer put at this offset of this sequence (gulp index = offset / GULP)
                size_t done = 0;
                while (done < n[k]) {
                    const uint64_t oo = o + done;
                    const size_t in_gulp = (size_t)(oo % GULP), m = std::min(n[k] - done, GULP - in_gulp);
                    CHECK(verify((const uint8_t*)data[k] + done, m, (uint32_t)tag, (uint32_t)(oo / GULP), in_gulp / 4), "contents of sequence %lld at byte %llu", tag,
                          (unsigned long long)oo);
                    done += m;
                }
                got += n[k];
            }
            if (gpu) gpu->issued++;                         // "a kernel that reads this gulp was enqueued"
            for (int k = 0; k < nparts; k++) held.push_back(span[k]);
            while ((int)held.size() > hold) { CHECK(xengRingSpanRelease(held.front()) == 0, "release"); held.pop_front(); }
            offset += got;
            advance = got;
            if (got == want) res.gulps++;
            if (lossy && (res.gulps % 37) == 0) nap_us(strict ? 300 : 1500);
            if (got < want) {                               // the short tail of an ended sequence
                while (!held.empty()) { xengRingSpanRelease(held.front()); held.pop_front(); }
                // (keep asking: the next call says END_OF_DATA)
            }
        }
        while (!held.empty()) { CHECK(xengRingSpanRelease(held.front()) == 0, "release"); held.pop_front(); }
    }
    return res;
}

static void writer_thread(xengRing* r, int nseq, int ngulps, FakeGpu* gpu) {
    for (int s = 0; s < nseq; s++) {
        long long seq = -1;
        CHECK(xengRingBeginSequence(r, 100 + s, "hdr\0\0\0\0", 8, 1, &seq) == 0, "BeginSequence");
        for (int k = 0; k < ngulps; k++) {
            void* data = nullptr;
            long long span = 0;
            int rc = xengRingReserve(r, seq, GULP, 0, 0, &data, &span);        // ask first, as a caller under an interpreter lock does
            if (rc == XENG_STATUS_WOULD_BLOCK) rc = xengRingReserve(r, seq, GULP, 0, 1, &data, &span);
            CHECK(rc == 0, "Reserve rc %d", rc);
            fill((uint8_t*)data, GULP, (uint32_t)(100 + s), (uint32_t)k);
            if (gpu) gpu->issued++;                          // "the kernel that wrote the span"
            CHECK(xengRingCommit(r, seq, span, GULP) == 0, "Commit");
            CHECK(xengRingSpanRelease(span) == 0, "release");
        }
        CHECK(xengRingEndSequence(r, seq) == 0, "EndSequence");
    }
    CHECK(xengRingEndWriting(r) == 0, "EndWriting");
}

static void scenario_streams(bool hooks, bool two_parts, int rounds, bool with_guaranteed = true) {
    for (int round = 0; round < rounds; round++) {
        FakeGpu gpu;
        std::thread gth;
        xengRing* r = nullptr;
        CHECK(xengRingCreate(&r, "stress", XENG_SPACE_SYSTEM) == 0, "create");
        CHECK(xengRingResize(r, GULP, 16 * GULP) == 0, "resize");
        if (hooks) {
            CHECK(xengRingSetStampHooks(r, hook_now, hook_done, hook_wait, &gpu) == 0, "hooks");
            gth = std::thread([&] { gpu.run(); });
        } else {
            CHECK(xengRingSetRecycle(r, 1) == 0, "recycle");
        }
        int rid_g = -1, rid_l = -1;
        // (without a guaranteed reader nobody applies back-pressure: the writer overwrites the oldest spans and the readers skip)
        CHECK(xengRingOpenReader(r, with_guaranteed ? 1 : 0, &rid_g) == 0 && xengRingOpenReader(r, 0, &rid_l) == 0, "open readers");
        const int NSEQ = 3, NG = 1200;
        Result rg, rl, rlate;
        std::thread tg([&] { rg = reader_thread(r, rid_g, 6, !with_guaranteed, hooks ? &gpu : nullptr, with_guaranteed, two_parts); });
        std::thread tl([&] { rl = reader_thread(r, rid_l, 2, true, hooks ? &gpu : nullptr, false, false); });
        std::thread tw([&] { writer_thread(r, NSEQ, NG, hooks ? &gpu : nullptr); });
        nap_us(3000);
        int rid_late = -1;
        CHECK(xengRingOpenReader(r, 0, &rid_late) == 0, "late reader");
        std::thread tlate([&] { rlate = reader_thread(r, rid_late, 1, false, nullptr, false, false); });
        tw.join(); tg.join(); tl.join(); tlate.join();
        if (with_guaranteed)
            CHECK(rg.seqs == NSEQ && rg.gulps == (long long)NSEQ * NG / (two_parts ? 2 : 1), "guaranteed reader saw %lld gulps in %lld sequences", rg.gulps, rg.seqs);
        CHECK(rl.gulps * (long long)GULP + rl.skipped <= (long long)NSEQ * NG * (long long)GULP, "lossy reader accounting");
        CHECK(xengRingCloseReader(r, rid_g) == 0 && xengRingCloseReader(r, rid_l) == 0 && xengRingCloseReader(r, rid_late) == 0, "close");
        size_t cap = 0, live = 0, pool = 0;
        int nrd = 0;
        long long nseq = 0;
        unsigned long long cnt[5];
        CHECK(xengRingGetInfo(r, &cap, &live, &pool, &nrd, &nseq, cnt) == 0, "info");
        CHECK(live == 0 && nrd == 0 && nseq == NSEQ, "after the run: live %zu readers %d sequences %lld", live, nrd, nseq);
        CHECK(cnt[0] >= 1 && cnt[2] > 0, "the free list was never used (alloc %llu reuse %llu)", cnt[0], cnt[2]);
        CHECK(cnt[0] * GULP <= 8 * 16 * GULP + 64 * GULP, "the ring allocated %llu spans: beyond its bound", cnt[0]);
        if (round == 0)
            printf("   %s%s%s: alloc %llu free %llu reuse %llu stamp waits %llu, lossy reader skipped %lld bytes, late reader saw %lld gulps\n", hooks ? "hooks" : "recycle",
                   two_parts ? " two-part" : "", with_guaranteed ? "" : " (no guaranteed reader)", cnt[0], cnt[1], cnt[2], cnt[3], rl.skipped, rlate.gulps);
        CHECK(xengRingDestroy(r) == 0, "destroy");
        if (hooks) { gpu.stop.store(true); gth.join(); }
    }
}

static void scenario_destroy_under_readers() {
    for (int round = 0; round < 20; round++) {
        xengRing* r = nullptr;
        CHECK(xengRingCreate(&r, "doomed", XENG_SPACE_SYSTEM) == 0, "create");
        CHECK(xengRingSetRecycle(r, 1) == 0, "recycle");
        CHECK(xengRingResize(r, GULP, 8 * GULP) == 0, "resize");
        int rid_s = -1, rid_h = -1;
        CHECK(xengRingOpenReader(r, 1, &rid_s) == 0 && xengRingOpenReader(r, 0, &rid_h) == 0, "open");
        long long seq = -1;
        CHECK(xengRingBeginSequence(r, 1, "hdr\0\0\0\0", 8, 1, &seq) == 0, "begin");
        for (int k = 0; k < 4; k++) {
            void* d = nullptr; long long sp = 0;
            CHECK(xengRingReserve(r, seq, GULP, 0, 1, &d, &sp) == 0, "reserve");
            fill((uint8_t*)d, GULP, 1, (uint32_t)k);
            CHECK(xengRingCommit(r, seq, sp, GULP) == 0 && xengRingSpanRelease(sp) == 0, "commit");
        }
        // the holder takes two spans and keeps them across the destroy
        std::vector<long long> kept;
        std::vector<void*> kept_ptr;
        {
            long long s2 = 0, tag = 0; int nr = 0; const void* h = nullptr; size_t hl = 0;
            CHECK(xengRingNextSequence(r, rid_h, 1, &s2, &tag, &nr, &h, &hl) == 0, "next");
            size_t adv = 0;
            for (int k = 0; k < 2; k++) {
                void* d = nullptr; size_t n = 0, sk = 0; long long sp = 0;
                CHECK(xengRingAcquire(r, rid_h, adv, GULP, 1, &d, &n, &sp, &sk) == 0 && n == GULP, "acquire");
                kept.push_back(sp); kept_ptr.push_back(d); adv = n;
            }
        }
        // the sleeper reads everything there is and then sleeps in Acquire (the sequence is still open)
        std::atomic<int> sleeper_rc{-1};
        std::thread ts([&] {
            long long s2 = 0, tag = 0; int nr = 0; const void* h = nullptr; size_t hl = 0;
            if (xengRingNextSequence(r, rid_s, 1, &s2, &tag, &nr, &h, &hl) != 0) { sleeper_rc = 99; return; }
            size_t adv = 0;
            for (;;) {
                void* d = nullptr; size_t n = 0, sk = 0; long long sp = 0;
                int rc = xengRingAcquire(r, rid_s, adv, GULP, 1, &d, &n, &sp, &sk);
                if (rc) { sleeper_rc = rc; return; }
                xengRingSpanRelease(sp);
                adv = n;
            }
        });
        nap_us(2000 + 300 * round);
        CHECK(xengRingDestroy(r) == 0, "destroy under readers");
        ts.join();
        CHECK(sleeper_rc.load() == XENG_STATUS_END_OF_DATA, "the sleeping reader woke with rc %d", sleeper_rc.load());
        // spans outlive their ring: the bytes are still there, and giving them back afterwards is fine
        for (size_t k = 0; k < kept.size(); k++) {
            CHECK(verify((const uint8_t*)kept_ptr[k], GULP, 1, (uint32_t)k), "a held span changed under its holder after destroy");
            CHECK(xengRingSpanRelease(kept[k]) == 0, "late release");
        }
    }
}

static void scenario_foreign_release_and_declarations() {
    FakeGpu gpu;
    xengRing* r = nullptr;
    CHECK(xengRingCreate(&r, "decl", XENG_SPACE_SYSTEM) == 0, "create");
    CHECK(xengRingSetStampHooks(r, hook_now, hook_done, hook_wait, &gpu) == 0, "hooks");
    unsigned classes = 0, declared = 0, users = 0;
    CHECK(xengRingGetStampClasses(r, &classes, &declared, &users) == 0 && classes == 31u && declared == 0 && users == 0, "undeclared ring: %u %u %u", classes, declared, users);
    CHECK(xengRingDeclareStreams(r, XENG_STREAMS_BEAM) == 0, "declare (writer)");
    long long seq = -1;
    CHECK(xengRingBeginSequence(r, 7, "hdr\0\0\0\0", 8, 1, &seq) == 0, "begin");
    CHECK(xengRingGetStampClasses(r, &classes, &declared, &users) == 0 && classes == (unsigned)XENG_STREAMS_BEAM && declared == 1 && users == 1, "writer declared: %u %u %u", classes,
          declared, users);
    int rid = -1;
    CHECK(xengRingOpenReader(r, 1, &rid) == 0, "reader");
    CHECK(xengRingGetStampClasses(r, &classes, &declared, &users) == 0 && classes == 31u && users == 2, "an undeclared reader widens every stamp: %u %u %u", classes, declared, users);
    CHECK(xengRingDeclareStreams(r, 0) == 0, "declare (reader, no classes)");
    CHECK(xengRingGetStampClasses(r, &classes, &declared, &users) == 0 && classes == (unsigned)XENG_STREAMS_BEAM && declared == 2, "all declared: %u %u %u", classes, declared, users);
    // a span whose last reference is dropped by a thread that has nothing to do with the ring: stamped (ticket 5 incomplete) and
    // not reissued until the ticket completes
    void* d = nullptr; long long sp = 0;
    CHECK(xengRingReserve(r, seq, GULP, 0, 1, &d, &sp) == 0, "reserve");
    gpu.issued = 5;
    std::thread foreign([&] { CHECK(xengRingSpanRelease(sp) == 0, "foreign release"); });
    foreign.join();
    void* d2 = nullptr; long long sp2 = 0;
    CHECK(xengRingReserve(r, seq, GULP, 0, 0, &d2, &sp2) == XENG_STATUS_WOULD_BLOCK, "a span under an incomplete ticket was handed out without waiting");
    gpu.completed = 5;
    CHECK(xengRingReserve(r, seq, GULP, 0, 0, &d2, &sp2) == 0 && d2 == d, "after the ticket: the same memory again");
    CHECK(xengRingSpanRelease(sp2) == 0, "release");
    CHECK(xengRingCloseReader(r, rid) == 0 && xengRingDestroy(r) == 0, "teardown");
}

static void scenario_gather_asks_first() {
    xengRing* r = nullptr;
    CHECK(xengRingCreate(&r, "gather", XENG_SPACE_SYSTEM) == 0, "create");
    CHECK(xengRingResize(r, GULP, 64 * GULP) == 0, "resize");
    int rid = -1;
    CHECK(xengRingOpenReader(r, 1, &rid) == 0, "reader");
    long long seq = -1;
    CHECK(xengRingBeginSequence(r, 9, "hdr\0\0\0\0", 8, 1, &seq) == 0, "begin");
    for (int k = 0; k < 6; k++) {
        void* d = nullptr; long long sp = 0;
        CHECK(xengRingReserve(r, seq, GULP, 0, 1, &d, &sp) == 0, "reserve");
        fill((uint8_t*)d, GULP, 9, (uint32_t)k);
        CHECK(xengRingCommit(r, seq, sp, GULP) == 0 && xengRingSpanRelease(sp) == 0, "commit");
    }
    long long s2 = 0, tag = 0; int nr = 0; const void* h = nullptr; size_t hl = 0;
    CHECK(xengRingNextSequence(r, rid, 1, &s2, &tag, &nr, &h, &hl) == 0, "next");
    // a gulp of three spans cannot be handed out as one window (or two): it needs a gathered copy, which may wait -- so a caller
    // that must not wait is told so, and the retry (advance 0) gets the gulp
    void* d = nullptr; size_t n = 0, sk = 0; long long sp = 0;
    CHECK(xengRingAcquire(r, rid, 0, 3 * GULP, 0, &d, &n, &sp, &sk) == XENG_STATUS_WOULD_BLOCK, "gather with may_block = 0");
    CHECK(xengRingAcquire(r, rid, 0, 3 * GULP, 1, &d, &n, &sp, &sk) == 0 && n == 3 * GULP, "the retry");
    for (int k = 0; k < 3; k++) CHECK(verify((const uint8_t*)d + k * GULP, GULP, 9, (uint32_t)k), "gathered contents");
    CHECK(xengRingSpanRelease(sp) == 0, "release");
    // ... and the cursor moves on correctly: the next gulp is spans 3..5
    CHECK(xengRingAcquire(r, rid, 3 * GULP, 3 * GULP, 0, &d, &n, &sp, &sk) == XENG_STATUS_WOULD_BLOCK, "second gather asked first");
    CHECK(xengRingAcquire(r, rid, 0, 3 * GULP, 1, &d, &n, &sp, &sk) == 0 && n == 3 * GULP, "second retry");
    for (int k = 0; k < 3; k++) CHECK(verify((const uint8_t*)d + k * GULP, GULP, 9, (uint32_t)(3 + k)), "second gathered contents");
    CHECK(xengRingSpanRelease(sp) == 0, "release");
    CHECK(xengRingCloseReader(r, rid) == 0 && xengRingDestroy(r) == 0, "teardown");
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    fflush(stdout); printf("scenario 1: stamp hooks, lagging completer\n");
    scenario_streams(true, false, rounds);
    fflush(stdout); printf("scenario 2: recycling system ring, two-part acquires\n");
    scenario_streams(false, true, rounds);
    scenario_streams(false, false, 1);
    scenario_streams(true, false, 2, false);
    scenario_streams(false, false, 2, false);      // (no stamp waits to slow the writer down: the readers really fall behind and skip)
    fflush(stdout); printf("scenario 3: destroy under readers\n");
    scenario_destroy_under_readers();
    fflush(stdout); printf("scenario 4: foreign-thread release, declarations vs users\n");
    scenario_foreign_release_and_declarations();
    fflush(stdout); printf("scenario 5: gathered reads ask first\n");
    scenario_gather_asks_first();
    printf("ring stress: all scenarios passed\n");
    return 0;
}
