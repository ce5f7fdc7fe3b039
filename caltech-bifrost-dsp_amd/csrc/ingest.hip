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
#include <mutex>

#include "xeng_common.h"

namespace xeng {

__device__ __forceinline__ uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
__device__ __forceinline__ uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }

// One wave per packet (four per work-group, grid-stride): the header is fetched with two 16-byte loads (all lanes,
// same address) and byte-swapped in registers; payload rows move as 16-byte pieces, eight loads in flight per lane
// before the first store, when the geometry is 16-byte aligned (the deployed 64-byte rows are), else byte by byte.
__global__ __launch_bounds__(256) void snap2_unpack_kernel(const uint8_t* __restrict__ pkts, int npkt, size_t stride,
                                                           uint8_t* __restrict__ out, unsigned long long seq0, int ntime,
                                                           int chan0_pipe, int nchan_tot, int npol_tot, int payload_max,
                                                           int* __restrict__ counters) {
    const bool aligned = (((uintptr_t)pkts | (uintptr_t)out | stride) & 15) == 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ndropped = 0;
    for (int p = blockIdx.x * 4 + wave; p < npkt; p += gridDim.x * 4) {
        const uint8_t* h = pkts + (size_t)p * stride;
        unsigned long long seq;
        int npol, nchan;
        long long chan0, pol0;
        if (aligned) {
            const uint4 h0 = *reinterpret_cast<const uint4*>(h), h1 = *reinterpret_cast<const uint4*>(h + 16);
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
        const uint8_t* src = h + 32;
        uint8_t* dst = out + (((size_t)(seq - seq0) * nchan_tot + (size_t)chan0) * npol_tot + (size_t)pol0);
        if (aligned && ((npol | npol_tot | (int)pol0) & 15) == 0) {
            const int per_row = npol >> 4;                       // 16-byte pieces per channel row
            const int n = nchan * per_row;
            for (int i0 = lane; i0 < n; i0 += 8 * 64) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * 64;
                    if (i < n) v[u] = *reinterpret_cast<const uint4*>(src + (size_t)i * 16);   // rows are contiguous in the packet
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * 64;
                    if (i < n) {
                        const int c = i / per_row, j = i - c * per_row;
                        *reinterpret_cast<uint4*>(dst + (size_t)c * npol_tot + j * 16) = v[u];
                    }
                }
            }
        } else {
            for (int i = lane; i < nchan * npol; i += 64) {
                const int c = i / npol, j = i - c * npol;
                dst[(size_t)c * npol_tot + j] = src[i];
            }
        }
    }
    // only drops are counted on the device (rare): thousands of waves adding to one counter serialise in L2
    // (4096 same-address atomics cost ~40 us); placed = npkt - dropped on the host
    if (lane == 0 && ndropped) atomicAdd(&counters[1], ndropped);
}

static int* g_counters[16] = {};
static std::mutex g_ingest_mu;     // the drop counter of a device is shared by all callers

}  // namespace xeng

using namespace xeng;

static int snap2_launch(hipStream_t s, const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, uint64_t seq0,
                        int ntime, int chan0_pipeline, int nchan_tot, int npol_tot, int clear, int* counters) {
    if (clear) XENG_HIP(hipMemsetAsync(out_dev, 0, (size_t)ntime * nchan_tot * npol_tot, s));   // missing packets = blanked samples
    if (npkt > 0) {
        hipLaunchKernelGGL(snap2_unpack_kernel, dim3((npkt + 3) / 4 < 2048 ? (npkt + 3) / 4 : 2048), dim3(256), 0, s, (const uint8_t*)packets_dev, npkt,
                           pkt_stride, (uint8_t*)out_dev, (unsigned long long)seq0, ntime, chan0_pipeline, nchan_tot, npol_tot,
                           (int)(pkt_stride - 32), counters);
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
    if (!g_counters[*dev]) {
        XENG_HIP(hipMalloc((void**)&g_counters[*dev], 2 * sizeof(int)));
        XENG_HIP(hipMemset(g_counters[*dev], 0, 2 * sizeof(int)));
    }
    return XENG_STATUS_SUCCESS;
}

extern "C" int xengSnap2Unpack(const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, uint64_t seq0, int ntime,
                               int chan0_pipeline, int nchan_tot, int npol_tot, int clear, int* nplaced, int* ndropped) {
    std::lock_guard<std::mutex> lk(g_ingest_mu);
    int dev = 0;
    int rc = snap2_check(packets_dev, npkt, pkt_stride, out_dev, ntime, nchan_tot, npol_tot, &dev);
    if (rc) return rc;
    hipStream_t s;
    rc = get_stream(STREAM_COPY, &s);
    if (rc) return rc;
    XENG_HIP(hipMemsetAsync(g_counters[dev], 0, 2 * sizeof(int), s));
    rc = snap2_launch(s, packets_dev, npkt, pkt_stride, out_dev, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot, clear,
                      g_counters[dev]);
    if (rc) return rc;
    int host[2] = {0, 0};
    XENG_HIP(hipMemcpyAsync(host, g_counters[dev], sizeof(host), hipMemcpyDeviceToHost, s));
    XENG_HIP(hipStreamSynchronize(s));
    if (nplaced) *nplaced = npkt - host[1];
    if (ndropped) *ndropped = host[1];
    return XENG_STATUS_SUCCESS;
}

// Enqueue-only flavour on the X-engine's staging stream: a gulp unpacked this way and then handed to
// xengXgpuKernelAsync is ordered before the contraction that reads it (the dump waits for that stream).  Drop
// counts accumulate on the device until the next synchronous call; nothing is waited for.
extern "C" int xengSnap2UnpackAsync(const void* packets_dev, int npkt, size_t pkt_stride, void* out_dev, uint64_t seq0, int ntime,
                                    int chan0_pipeline, int nchan_tot, int npol_tot, int clear) {
    std::lock_guard<std::mutex> lk(g_ingest_mu);
    int dev = 0;
    int rc = snap2_check(packets_dev, npkt, pkt_stride, out_dev, ntime, nchan_tot, npol_tot, &dev);
    if (rc) return rc;
    hipStream_t s;
    rc = get_stream(STREAM_XGPU, &s);
    if (rc) return rc;
    return snap2_launch(s, packets_dev, npkt, pkt_stride, out_dev, seq0, ntime, chan0_pipeline, nchan_tot, npol_tot, clear,
                        g_counters[dev]);
}
