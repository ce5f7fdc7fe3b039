// Ingest: scatter SNAP2 F-engine packets into the time-major gulp layout the X-engine and the beamformer
// read (uint8[ntime][nchan][nstand*npol], pol fastest; corr_block.py:115-116).
//
// In the reference the scatter happens on the CPU inside bifrost's UDP capture (an absent submodule;
// capture_block.py:296-305 only configures it) and the assembled gulp then crosses PCIe in the Copy block.
// Here the receive slab (packets as they arrived, headers included: 32 B per ~6 KB) crosses PCIe and the
// scatter is a device kernel, so the host touches no payload byte.
//
// Packet (test_tx_vectors.py:38-48,103-108; test_tx_mt.c:39-49), header big-endian `>QLHHHHLLL`:
//   u64 seq | u32 sync_time (magic) | u16 npol | u16 npol_tot | u16 nchan | u16 nchan_tot |
//   u32 chan_block_id | u32 chan0 | u32 pol0 ; payload u8[nchan][npol] (4+4 bit, npol = stands*2 in the packet)
// Destination of payload row c: gulp[seq - seq0][chan0 - chan0_pipeline + c][pol0 .. pol0 + npol).
#include <chrono>
#include <cstring>
#include <mutex>

#include "xeng_common.h"

namespace xeng {

__device__ __forceinline__ uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
__device__ __forceinline__ uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

// The gulp is stored THROUGH the caches (system scope: sc0 sc1).  The synchronous call reports completion from inside the
// kernel (the last work-group raises a word in pinned memory, below) and its caller may hand the gulp to another stream --
// or read it from the host -- the moment it returns, before the kernel has formally ended: a plain store could still sit
// dirty in its XCD's L2 then (the eight L2s are only written back by the end-of-kernel release).  With write-through stores a
// work-group's `s_waitcnt vmcnt(0)` before its ticket means its bytes are in memory, so the report implies the whole gulp is.
// The scatter streams every byte once: nothing is lost by not keeping it in L2.
typedef unsigned int snap2_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_through16(uint8_t* p, snap2_v4u v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_through1(uint8_t* p, uint8_t v) {
    asm volatile("global_store_byte %0, %1, off sc0 sc1" ::"v"(p), "v"((uint32_t)v) : "memory");
}

// One wave per packet (four per work-group, grid-stride): the header is fetched with two 16-byte loads (all lanes,
// same address) and byte-swapped in registers; payload rows move as 16-byte pieces, eight loads in flight per lane
// before the first store, when the geometry is 16-byte aligned (the deployed 64-byte rows are), else byte by byte.
__global__ __launch_bounds__(256) void snap2_unpack_kernel(const uint8_t* __restrict__ pkts, int npkt, size_t stride,
                                                           uint8_t* __restrict__ out, unsigned long long seq0, int ntime,
                                                           int chan0_pipe, int nchan_tot, int npol_tot, int payload_max,
                                                           int* __restrict__ counter,
                                                           unsigned long long* __restrict__ row_cover,
                                                           unsigned int* __restrict__ row_geom,
                                                           unsigned int* __restrict__ done_count,
                                                           unsigned long long* __restrict__ host_state, unsigned long long call_id) {
    const bool aligned = (((uintptr_t)pkts | (uintptr_t)out | stride) & 15) == 0;
    const bool fast16 = aligned && (npol_tot & 15) == 0;         // 16-byte pieces are possible (wave-uniform)
    const int n_spec = payload_max >> 4;                         // whole 16-byte pieces of a packet's payload area
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    int ndropped = 0;
    for (int p = blockIdx.x * 4 + wave; p < npkt; p += gridDim.x * 4) {
        const uint8_t* h = pkts + (size_t)p * stride;
        const uint8_t* src = h + 32;
        unsigned long long seq;
        int npol, nchan;
        long long chan0, pol0;
        v4u v[8];
        if (aligned) {
            const uint4 h0 = *reinterpret_cast<const uint4*>(h), h1 = *reinterpret_cast<const uint4*>(h + 16);
            // The first 8 x 64 payload pieces are fetched together with the header, before it has been looked at (they lie
            // inside the slab whatever the header says): header, validation and payload are then one memory round trip,
            // not two.  (The asm pins the loads here; the compiler otherwise sinks them behind the validation.)
            if (fast16) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = lane + u * 64;
                    v[u] = *reinterpret_cast<const v4u*>(src + (size_t)(i < n_spec ? i : 0) * 16);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("" : "+v"(v[u]));
            }
            seq = ((unsigned long long)__builtin_bswap32(h0.x) << 32) | __builtin_bswap32(h0.y);
            npol = (int)(__builtin_bswap32(h0.w) >> 16);                  // bytes 12-13
            nchan = (int)(__builtin_bswap32(h1.x) >> 16);                 // bytes 16-17
            chan0 = (long long)__builtin_bswap32(h1.z) - chan0_pipe;      // bytes 24-27
            pol0 = __builtin_bswap32(h1.w);                               // bytes 28-31
        } else {
            seq = ((unsigned long long)be32(h) << 32) | be32(h + 4);
            npol = (int)be16(h + 12);
            nchan = (int)be16(h + 16);
            chan0 = (long long)be32(h + 24) - chan0_pipe;
            pol0 = be32(h + 28);
        }
        // wave-uniform validation: window, geometry, payload size
        const bool ok = seq >= seq0 && seq - seq0 < (unsigned long long)ntime && npol > 0 && nchan > 0 && chan0 >= 0 &&
                        chan0 + nchan <= nchan_tot && pol0 + npol <= npol_tot && (long long)nchan * npol <= payload_max;
        if (!ok) {
            ndropped++;
            continue;
        }
        uint8_t* dst = out + (((size_t)(seq - seq0) * nchan_tot + (size_t)chan0) * npol_tot + (size_t)pol0);
        if (fast16 && ((npol | (int)pol0) & 15) == 0) {
            const int per_row = npol >> 4;                       // 16-byte pieces per channel row
            const int n = nchan * per_row;                       // (<= n_spec: checked above)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = lane + u * 64;
                if (i < n) {
                    const int c = i / per_row, j = i - c * per_row;
                    store_through16(dst + (size_t)c * npol_tot + j * 16, v[u]);
                }
            }
            for (int i0 = lane + 8 * 64; i0 < n; i0 += 8 * 64) {     // payloads beyond 8 KiB
                v4u w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * 64;
                    if (i < n) w[u] = *reinterpret_cast<const v4u*>(src + (size_t)i * 16);   // rows are contiguous in the packet
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * 64;
                    if (i < n) {
                        const int c = i / per_row, j = i - c * per_row;
                        store_through16(dst + (size_t)c * npol_tot + j * 16, w[u]);
                    }
                }
            }
        } else {
            for (int i = lane; i < nchan * npol; i += 64) {
                const int c = i / npol, j = i - c * npol;
                store_through1(dst + (size_t)c * npol_tot + j, src[i]);
            }
        }
        if (row_cover && lane == 0) {
            // coverage of the gulp, so that the caller can skip the zero-fill when nothing is missing: per time row the
            // set of packet cells (channel block, input block) that arrived.  Exact when the row's packets share one
            // geometry, sit on its grid and the row has at most 63 cells; anything else sets bit 63 = "row irregular".
            // (After the stores have been issued: the compare-and-swap's round trip overlaps with them.)
            const int t = (int)(seq - seq0);
            const unsigned int g = ((unsigned int)nchan << 16) | (unsigned int)npol;
            const unsigned int g0 = atomicCAS(&row_geom[t], 0u, g);
            unsigned long long bit = 1ull << 63;
            if ((g0 == 0u || g0 == g) && nchan_tot % nchan == 0 && npol_tot % npol == 0 && chan0 % nchan == 0 && pol0 % npol == 0) {
                const long long cell = (chan0 / nchan) * (npol_tot / npol) + pol0 / npol;
                if ((long long)(nchan_tot / nchan) * (npol_tot / npol) <= 63) bit = 1ull << cell;
            }
            atomicOr(&row_cover[t], bit);
        }
    }
    // only drops are counted on the device (rare): thousands of waves adding to one counter serialise in L2
    // (4096 same-address atomics cost ~40 us); placed = npkt - dropped on the host
    if (lane == 0 && ndropped) atomicAdd(counter, ndropped);
    if (!host_state) return;
    // Synchronous call: the work-group that finishes LAST reports to the host itself -- it checks the rows' coverage (what
    // snap2_complete() does on the host), clears the block for the next call, writes {drops, complete} into the pinned
    // mirror and then raises the call's id in the mirror's first word.  The host polls that word: one launch per call, no
    // memset before it, no copy and no stream synchronisation behind it.  Every wave first waits for its own atomics to be
    // acknowledged (vmcnt counts them), so the ticket of a work-group is taken after all its updates.
    __shared__ int last_wg;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // two-level ticket (thousands of returning atomics on ONE word serialise at ~10 ns each): work-group b counts in
        // sub-counter b % 16; the last arrival of a sub-counter counts in the top word; the last of those reports
        const unsigned int k = blockIdx.x & 15u, members = (gridDim.x - k + 15u) >> 4;
        int last = 0;
        if (atomicAdd(&done_count[1 + k], 1u) == members - 1) {
            atomicAnd(&done_count[1 + k], 0u);
            const unsigned int groups = gridDim.x < 16u ? gridDim.x : 16u;
            last = atomicAdd(&done_count[0], 1u) == groups - 1;
        }
        last_wg = last;
    }
    __syncthreads();
    if (!last_wg) return;
    // (only atomics and coherent loads ever touch the block: a plain store would leave a dirty line in this XCD's L2 beside
    // words that the other XCDs update at the memory side)
    bool ok = true;
    if (row_cover) {
        // the check of snap2_complete(), one time row per thread: every cell of the row's packet grid arrived
        for (int t = threadIdx.x; t < ntime; t += blockDim.x) {
            const unsigned int g = __hip_atomic_load(&row_geom[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long cv = __hip_atomic_load(&row_cover[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int nchan = (int)(g >> 16), npol = (int)(g & 0xFFFF);
            bool row_ok = g != 0 && nchan > 0 && npol > 0 && nchan_tot % nchan == 0 && npol_tot % npol == 0;
            if (row_ok) {
                const long long cells = (long long)(nchan_tot / nchan) * (npol_tot / npol);
                row_ok = cells <= 63 && cv == (1ull << cells) - 1;            // (bit 63 = irregular row)
            }
            ok = ok && row_ok;
            atomicAnd(&row_cover[t], 0ull);                                   // cleared for the next call (no-return atomics)
            atomicAnd(&row_geom[t], 0u);
        }
    }
    const int complete = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) {
        const unsigned long long drops = (unsigned long long)(unsigned int)atomicExch(counter, 0);
        atomicAnd(&done_count[0], 0u);                                        // the next call's kernel is behind this one in stream order
        __hip_atomic_store(&host_state[1], drops | ((unsigned long long)(complete ? 1 : 0) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&host_state[0], call_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// per device: the state block of the running synchronous call (drops + coverage of up to SNAP2_MAX_ROWS time rows) and the
// drops of the enqueue-only calls since they were last read
constexpr int SNAP2_MAX_ROWS = 8192;
struct IngestState {
    int* counters = nullptr;                 // state block of the synchronous call: [8 B: drops][row_cover][row_geom]
    unsigned int* tickets = nullptr;         // completion tickets of the synchronous call's kernel: [top][16 sub-counters]
    int* async_drops = nullptr;              // ... and, behind it, the drop counter of the enqueue-only calls
    int* scratch_drops = nullptr;            // drop counter of the second scatter of a lossy slab (already counted)
    void* host = nullptr;                    // pinned mirror: [8 B: id of the last finished call][the block]
    int* host_async = nullptr;
    unsigned long long ncalls = 0;
};
static IngestState g_ingest[16];
static std::mutex g_ingest_mu;     // the state of a device is shared by all callers

}  // namespace xeng

using namespace xeng;

static int snap2_launch(hipStream_t s, const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, uint64_t seq0,
                        int ntime, int chan0_pipeline, int nchan_tot, int npol_tot, int clear, int* counter,
                        unsigned long long* row_cover, unsigned int* row_geom, unsigned int* done_count = nullptr,
                        unsigned long long* host_state = nullptr, unsigned long long call_id = 0) {
    if (clear) XENG_HIP(hipMemsetAsync(out_dev, 0, (size_t)ntime * nchan_tot * npol_tot, s));   // missing packets = blanked samples
    if (npkt > 0) {
        hipLaunchKernelGGL(snap2_unpack_kernel, dim3((npkt + 3) / 4 < 2048 ? (npkt + 3) / 4 : 2048), dim3(256), 0, s, (const uint8_t*)packets_dev, npkt,
                           pkt_stride, (uint8_t*)out_dev, (unsigned long long)seq0, ntime, chan0_pipeline, nchan_tot, npol_tot,
                           (int)(pkt_stride - 32), counter, row_cover, row_geom, done_count, host_state, call_id);
        XENG_HIP(hipGetLastError());
    }
    return XENG_STATUS_SUCCESS;
}

static int snap2_check(const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, int ntime, int nchan_tot, int npol_tot,
                       int* dev) {
    if (!packets_dev || !out_dev) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Snap2Unpack: null buffer");
    if (npkt < 0 || ntime <= 0 || nchan_tot <= 0 || npol_tot <= 0 || pkt_stride < 33)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Snap2Unpack: npkt=%d ntime=%d nchan_tot=%d npol_tot=%d stride=%zu", npkt, ntime,
                  nchan_tot, npol_tot, pkt_stride);
    XENG_HIP(hipGetDevice(dev));
    if (*dev < 0 || *dev >= 16) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "device %d out of range", *dev);
    IngestState& st = g_ingest[*dev];
    if (!st.counters) {
        const size_t nb = 8 + SNAP2_MAX_ROWS * (sizeof(unsigned long long) + sizeof(unsigned int)) + 8 + 16 + 128;
        uint8_t* base = nullptr;
        XENG_HIP(hipMalloc((void**)&base, nb));
        XENG_HIP(hip_memset_now(base, 0, nb));
        XENG_HIP(hipHostMalloc(&st.host, nb + 8, hipHostMallocDefault));     // (coherent pinned memory: the kernel writes it)
        memset(st.host, 0, nb + 8);
        st.async_drops = (int*)(base + nb - 128 - 16);
        st.scratch_drops = (int*)(base + nb - 128 - 8);
        st.tickets = (unsigned int*)(base + nb - 128);
        st.host_async = (int*)((uint8_t*)st.host + 8 + nb - 128 - 16);
        st.counters = (int*)base;
    }
    return XENG_STATUS_SUCCESS;
}

extern "C" int xengSnap2Unpack(const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, uint64_t seq0, int ntime,
                               int chan0_pipeline, int nchan_tot, int npol_tot, int clear, int* nplaced, int* ndropped) {
    std::lock_guard<std::mutex> lk(g_ingest_mu);
    int dev = 0;
    int rc = snap2_check(packets_dev, npkt, pkt_stride, out_dev, ntime, nchan_tot, npol_tot, &dev);
    if (rc) return rc;
    IngestState& st = g_ingest[dev];
    hipStream_t s;
    rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    // With `clear`, a complete slab needs no zero-fill at all (the packets overwrite every byte): scatter first while
    // recording which packet cells arrived, and fall back to zero-fill + scatter only when something is missing.  That
    // saves one full write of the gulp (32 MB at config 2) in the normal, loss-free case.
    const bool track = clear && npkt > 0 && ntime <= SNAP2_MAX_ROWS;
    int dropped = 0;
    if (npkt > 0 && ntime <= SNAP2_MAX_ROWS) {
        // one launch, and the kernel's last work-group reports to the pinned mirror (see the kernel): state block =
        // [drops | ticket][row_cover: ntime x 8 B][row_geom: ntime x 4 B], self-clearing
        unsigned long long* cover = (unsigned long long*)((uint8_t*)st.counters + 8);
        unsigned int* geom = (unsigned int*)((uint8_t*)st.counters + 8 + (size_t)ntime * sizeof(unsigned long long));
        unsigned long long* hs = (unsigned long long*)st.host;
        const unsigned long long id = ++st.ncalls;
        rc = snap2_launch(s, packets_dev, npkt, pkt_stride, out_dev, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot, clear && !track,
                          st.counters, track ? cover : nullptr, track ? geom : nullptr, st.tickets, hs, id);
        stream_tick(STREAM_COPY);
        if (rc) return rc;
        // poll the mirror's first word (sub-microsecond once the kernel has finished); a kernel that does not report within
        // 2 s is a fault: fall back to the stream, whose error the HIP call returns
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(&hs[0], __ATOMIC_ACQUIRE) != id) {
            if ((++spins & 0x3FFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                XENG_HIP(hipStreamSynchronize(s));
                if (__atomic_load_n(&hs[0], __ATOMIC_ACQUIRE) != id) XENG_FAIL(XENG_STATUS_DEVICE_ERROR, "Snap2Unpack: the scatter kernel did not report");
                break;
            }
        }
        dropped = (int)(hs[1] & 0xFFFFFFFFull);
        const bool complete = (hs[1] >> 32) != 0;
        if (track && !complete) {
            // something is missing (or the stream is irregular): blank the gulp and scatter again
            rc = snap2_launch(s, packets_dev, npkt, pkt_stride, out_dev, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot, 1,
                              st.scratch_drops, nullptr, nullptr);
            stream_tick(STREAM_COPY);
            if (rc) return rc;
            XENG_HIP(hipStreamSynchronize(s));
        }
    } else {
        // no packets, or a window too long for the coverage block: zero-fill (if asked) + scatter, drops by a copy
        XENG_HIP(hipMemsetAsync(st.scratch_drops, 0, sizeof(int), s));
        rc = snap2_launch(s, packets_dev, npkt, pkt_stride, out_dev, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot, clear,
                          st.scratch_drops, nullptr, nullptr);
        stream_tick(STREAM_COPY);
        if (rc) return rc;
        XENG_HIP(hipMemcpyAsync(st.host_async + 2, st.scratch_drops, sizeof(int), hipMemcpyDeviceToHost, s));
        XENG_HIP(hipStreamSynchronize(s));
        dropped = st.host_async[2];
    }
    if (nplaced) *nplaced = npkt - dropped;
    if (ndropped) *ndropped = dropped;
    return XENG_STATUS_SUCCESS;
}

// Enqueue-only flavour on the X-engine's staging stream: a gulp unpacked this way and then handed to
// xengXgpuKernelAsync is ordered before the contraction that reads it (the dump waits for that stream).  Nothing is
// waited for; the packets these calls drop are counted on the device in a counter of their own, read (and cleared)
// with xengSnap2GetAsyncDrops.
extern "C" int xengSnap2UnpackAsync(const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, uint64_t seq0, int ntime,
                                    int chan0_pipeline, int nchan_tot, int npol_tot, int clear) {
    std::lock_guard<std::mutex> lk(g_ingest_mu);
    int dev = 0;
    int rc = snap2_check(packets_dev, npkt, pkt_stride, out_dev, ntime, nchan_tot, npol_tot, &dev);
    if (rc) return rc;
    hipStream_t s;
    rc = get_stream(STREAM_XGPU, &s);
    if (rc) return rc;
    rc = snap2_launch(s, packets_dev, npkt, pkt_stride, out_dev, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot, clear,
                      g_ingest[dev].async_drops, nullptr, nullptr);
    staging_stream_touched();              // (the next contraction waits for this stream; ticked after the enqueue: stream clocks)
    return rc;
}

// Packets dropped by the enqueue-only calls since the last call of this function; waits for the staging stream.
extern "C" int xengSnap2GetAsyncDrops(int* ndropped) {
    std::lock_guard<std::mutex> lk(g_ingest_mu);
    if (!ndropped) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Snap2GetAsyncDrops: null pointer");
    int dev = 0;
    XENG_HIP(hipGetDevice(&dev));
    *ndropped = 0;
    if (dev < 0 || dev >= 16 || !g_ingest[dev].counters) return XENG_STATUS_SUCCESS;    // nothing was ever unpacked
    hipStream_t s;
    int rc = get_stream(STREAM_XGPU, &s);
    if (rc) return rc;
    XENG_HIP(hipMemcpyAsync(g_ingest[dev].host_async, g_ingest[dev].async_drops, sizeof(int), hipMemcpyDeviceToHost, s));
    XENG_HIP(hipMemsetAsync(g_ingest[dev].async_drops, 0, sizeof(int), s));
    XENG_HIP(hipStreamSynchronize(s));
    *ndropped = *g_ingest[dev].host_async;
    return XENG_STATUS_SUCCESS;
}

// ---------------------------------------------------------------- emulator side (tests, bench): re-stamp a slab's sequence numbers
// A receiver reuses its slab buffers: the same memory holds the packets of window k, then of window k + N.  The benchmark's
// replay source does the same with a fixed set of slabs and needs their headers to say the new window: packet p of the slab
// gets sequence number seq0 + p / pkts_per_seq (big-endian, first 8 bytes of the header: test_tx_vectors.py:38-48), nothing
// else is touched.  On the copy stream, complete on return.
namespace xeng {
__global__ void snap2_stamp_seq_kernel(uint8_t* pkts, int npkt, size_t stride, unsigned long long seq0, int pkts_per_seq) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npkt) return;
    const unsigned long long seq = seq0 + (unsigned long long)(p / pkts_per_seq);
    uint8_t* h = pkts + (size_t)p * stride;
    uint32_t hi = (uint32_t)(seq >> 32), lo = (uint32_t)seq;
    *reinterpret_cast<uint32_t*>(h) = __builtin_bswap32(hi);
    *reinterpret_cast<uint32_t*>(h + 4) = __builtin_bswap32(lo);
}
}  // namespace xeng

extern "C" int xengSnap2StampSeq(void* packets_dev, int npkt, size_t pkt_stride, uint64_t seq0, int pkts_per_seq) {
    if (!packets_dev || npkt < 0 || pkt_stride < 32 || (pkt_stride & 3) || ((uintptr_t)packets_dev & 3) || pkts_per_seq <= 0)
        XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "Snap2StampSeq: bad arguments (npkt %d, stride %zu, packets per sequence number %d)", npkt, pkt_stride, pkts_per_seq);
    if (npkt == 0) return XENG_STATUS_SUCCESS;
    hipStream_t s;
    int rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    hipLaunchKernelGGL(xeng::snap2_stamp_seq_kernel, dim3((npkt + 255) / 256), dim3(256), 0, s, (uint8_t*)packets_dev, npkt, pkt_stride,
                       (unsigned long long)seq0, pkts_per_seq);
    XENG_HIP(hipGetLastError());
    stream_tick(STREAM_COPY);
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}

