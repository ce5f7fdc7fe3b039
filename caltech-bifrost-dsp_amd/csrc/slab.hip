// Packet slabs as gulps: the device passes (slab.h) shared by the X-engine (xcorr.hip) and the beamformer (beamform.hip).
#include "slab.h"

#include <algorithm>

#include "xeng_common.h"

namespace xeng {

__device__ __forceinline__ uint32_t slab_be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}
struct SlabHeader { unsigned long long seq; int npol, nchan; long long chan0, pol0; };
__device__ __forceinline__ SlabHeader slab_header(const uint8_t* h, int chan0_pipe) {
    SlabHeader r;
    r.seq = ((unsigned long long)slab_be32(h) << 32) | slab_be32(h + 4);
    r.npol = (int)(slab_be32(h + 12) >> 16);
    r.nchan = (int)(slab_be32(h + 16) >> 16);
    r.chan0 = (long long)slab_be32(h + 24) - chan0_pipe;
    r.pol0 = slab_be32(h + 28);
    return r;
}

// zero-fill: `nthreads` threads (this one: `tid`) share the 16-byte pieces of the scratch gulp
__device__ __forceinline__ void slab_clear_part(const SlabArgs& a, uint8_t* scratch, size_t tid, size_t nthreads) {
    const size_t n16 = (size_t)a.ntime * a.nchan * a.ninput / 16;     // (scratch gulps are whole 16-byte pieces: checked on the host)
    uint4* dst = reinterpret_cast<uint4*>(scratch);
    for (size_t k = tid; k < n16; k += nthreads) dst[k] = make_uint4(0, 0, 0, 0);
}

// scatter: one wave per packet (any order, duplicates allowed), `nwaves` waves (this one: `wave`) share the packets; the
// validation of snap2_unpack_kernel (ingest.hip), rows of npol bytes
__device__ __forceinline__ void slab_scatter_part(const SlabArgs& a, uint8_t* scratch, int wave, int nwaves, int lane) {
    const int payload_max = (int)a.stride - 32;
    for (int p = wave; p < a.npkt; p += nwaves) {
        const uint8_t* hp = a.pkts + (size_t)p * a.stride;
        const SlabHeader h = slab_header(hp, a.chan0);
        const bool ok = h.seq >= a.seq0 && h.seq - a.seq0 < (unsigned long long)a.ntime && h.npol > 0 && h.nchan > 0 && h.chan0 >= 0 &&
                        h.chan0 + h.nchan <= a.nchan && h.pol0 + h.npol <= a.ninput && (long long)h.nchan * h.npol <= payload_max;
        if (!ok) continue;
        uint8_t* dst = scratch + (((size_t)(h.seq - a.seq0) * a.nchan + (size_t)h.chan0) * a.ninput + (size_t)h.pol0);
        const uint8_t* src = hp + 32;
        const int n = h.nchan * h.npol;
        if (((h.npol | (int)h.pol0 | a.ninput) & 15) == 0 && (a.stride & 15) == 0 && (((uintptr_t)a.pkts | (uintptr_t)scratch) & 15) == 0) {
            const int per_row = h.npol >> 4;             // 16-byte pieces (the deployed 64-byte rows: four per channel)
            for (int i = lane; i < (n >> 4); i += 64)
                *reinterpret_cast<uint4*>(dst + (size_t)(i / per_row) * a.ninput + (size_t)(i % per_row) * 16) = *reinterpret_cast<const uint4*>(src + (size_t)i * 16);
        } else {
            for (int i = lane; i < n; i += 64) dst[(size_t)(i / h.npol) * a.ninput + (i % h.npol)] = src[i];
        }
    }
}

// verify + describe in one launch.  Every wave checks its 64 packets and adds ONE word to the tally: (packets out of place) << 32
// | 1.  The wave that reads back "all other waves have added theirs" has, in the same word, the gulp's total: it writes the
// descriptor and re-arms the tally.  One returning atomic per wave carries both the count and the ticket, so nothing has to be
// ordered and the kernel needs NO fence: an agent-scope fence on this part is an L2 write-back + invalidate on every XCD, and
// beside a running contraction (which lives on L2 hits of the gulps) that cost 7 % of the streaming rate (measured:
// profiles/r04/slab_paths.txt).  No LDS either.
// One launch serves up to two gulps (grid.y; the beamformer's two parts): gulp k has its own tally word, descriptor, scratch.
// fallbacks: gulps that took the scratch path (read by xeng*GetSlabFallbacks).
// force[k]: the host already knows the slab cannot be regular (packet count, stride, alignment); then one work-group, no check.
struct SlabJob {
    SlabArgs a[2];
    uint8_t* scratch[2];
    int force[2];
};
// INLINE (the beamformer's calls while the link is clean: one launch per call on the stream of its kernels, nothing else): a
// gulp that turns out irregular is zero-filled and scattered HERE, by the one work-group whose wave took the last ticket --
// 256 threads instead of a grid, a few milliseconds for a 32 MB gulp; the regular case costs one short launch and no launch that only
// finds out that it has nothing to do.  Every such gulp also bumps a counter in pinned host memory; the host, seeing it move,
// switches its next calls to the other form.  Otherwise (the X-engine: per integration, off its critical path; the beamformer
// after a recent loss) slab_clear_kernel + slab_scatter_kernel follow.
template <bool INLINE>
__global__ __launch_bounds__(256) void slab_prepare_kernel(SlabJob job, unsigned long long* __restrict__ tallies, int* __restrict__ fallbacks,
                                                                          int* __restrict__ fallbacks_host, GulpDesc* __restrict__ descs,
                                                                          SlabArgs* __restrict__ args_out) {
    __shared__ int s_fallback_here;                         // (INLINE only; 4 bytes of LDS fit beside any resident kernel)
    const int k = blockIdx.y;
    const SlabArgs& a = job.a[k];
    const int force_scratch = job.force[k];
    const unsigned int nblocks = force_scratch ? 1u : (unsigned int)((a.npkt + (int)blockDim.x - 1) / (int)blockDim.x);
    if (blockIdx.x >= nblocks) return;                      // (the grid is sized for the larger gulp; whole work-groups leave)
    if (INLINE) {
        if (threadIdx.x == 0) s_fallback_here = 0;
        __syncthreads();
    }
    bool ok = true;
    if (!force_scratch) {
        const int p = blockIdx.x * blockDim.x + threadIdx.x;
        if (p < a.npkt) {
            const SlabHeader h = slab_header(a.pkts + (size_t)p * a.stride, a.chan0);
            const int t = p / a.nblk, b = p % a.nblk;
            ok = h.seq == a.seq0 + (unsigned long long)t && h.pol0 == (long long)b * 64 && h.npol == 64 && h.nchan == a.nchan && h.chan0 == 0;
        }
    }
    const unsigned long long nbad = (unsigned long long)__popcll(__ballot(!ok));
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long nwaves = (unsigned long long)nblocks * (blockDim.x >> 6);
        const unsigned long long before = atomicAdd(&tallies[k], (nbad << 32) | 1ull);
        if ((before & 0xFFFFFFFFull) == nwaves - 1) {
            const bool fb = force_scratch || ((before >> 32) + nbad) != 0;
            GulpDesc d;
            if (fb) {
                d.base = job.scratch[k]; d.t_stride = (uint32_t)a.nchan * (uint32_t)a.ninput; d.c_stride = (uint32_t)a.ninput; d.b_stride = 64;
                atomicAdd(fallbacks, 1);
                // (the host's hint that the link is losing packets: pinned memory, read there without a wait)
                __hip_atomic_fetch_add(fallbacks_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                d.base = a.pkts + 32; d.t_stride = (uint32_t)a.nblk * a.stride; d.c_stride = 64; d.b_stride = a.stride;
            }
            d.pad = fb ? 1u : 0u;
            d.pad2 = 0;
            descs[k] = d;
            args_out[k] = a;          // (for the scatter kernels, should this gulp need them)
            tallies[k] = 0;           // (re-armed for the next gulp: launches on one stream, in order)
            if (INLINE && fb) s_fallback_here = 1;
        }
    }
    if (!INLINE) return;
    __syncthreads();
    if (!s_fallback_here) return;
    slab_clear_part(a, job.scratch[k], threadIdx.x, blockDim.x);
    __syncthreads();                  // (the zero-fill of this work-group is ordered before its scatter)
    slab_scatter_part(a, job.scratch[k], threadIdx.x >> 6, blockDim.x >> 6, threadIdx.x & 63);
}

// Once per integration, behind the prepare kernels of its gulps (grid.y = gulp): zero-fill and scatter of the gulps whose
// descriptor says "scratch"; the groups of every other gulp return at once.
__global__ __launch_bounds__(256) void slab_clear_kernel(const GulpDesc* __restrict__ desc, const SlabArgs* __restrict__ args) {
    const GulpDesc& d = desc[blockIdx.y];
    if (!d.pad) return;
    slab_clear_part(args[blockIdx.y], const_cast<uint8_t*>(d.base), (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ __launch_bounds__(256) void slab_scatter_kernel(const GulpDesc* __restrict__ desc, const SlabArgs* __restrict__ args) {
    const GulpDesc& d = desc[blockIdx.y];
    if (!d.pad) return;
    const SlabArgs a = args[blockIdx.y];
    slab_scatter_part(a, const_cast<uint8_t*>(d.base), blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, threadIdx.x & 63);
}

int slab_site_create(SlabSite* s) {
    void* p = nullptr;
    XENG_HIP(hipMalloc(&p, 32));
    XENG_HIP(hip_memset_now(p, 0, 32));
    s->tally = (unsigned long long*)p;       // two words: one per gulp of a launch
    s->fallbacks = (int*)p + 4;
    XENG_HIP(hipHostMalloc((void**)&s->fallbacks_host, sizeof(int)));
    *s->fallbacks_host = 0;
    return XENG_STATUS_SUCCESS;
}

void slab_site_destroy(SlabSite* s) {
    if (s->tally) (void)hipFree(s->tally);
    if (s->fallbacks_host) (void)hipHostFree(s->fallbacks_host);
    *s = SlabSite();
}

bool slab_maybe_regular(const SlabArgs& a, int rows) {
    return a.ninput > 0 && a.ninput % 64 == 0 && a.npkt > 0 && a.npkt == a.ntime * a.nblk && a.stride >= 32 + (size_t)a.nchan * 64 &&
           a.stride % 16 == 0 && ((uintptr_t)a.pkts & 15) == 0 && (uint64_t)a.nblk * a.stride * rows < (1ull << 31);
}

int slab_prepare_enqueue(hipStream_t stream, const SlabSite& site, const SlabArgs* a, const bool* maybe, int ngulp, GulpDesc* descs, SlabArgs* args_out,
                         uint8_t* const* scratch, bool inline_fallback) {
    SlabJob job;
    // (256 threads either way: four waves of <= 40 registers fit on a CU beside a contraction work-group -- 444 of 512 registers per
    // lane taken --; with 1024-thread groups the verify pass of the beamformer waited for a whole CU, i.e. for the running
    // contraction to end: 63 us on average instead of 8, profiles/r04/slab_paths.txt)
    const int bs = 256;
    unsigned int nblocks = 1;
    for (int k = 0; k < 2; k++) {
        const int kk = k < ngulp ? k : 0;
        job.a[k] = a[kk]; job.scratch[k] = scratch[kk]; job.force[k] = maybe[kk] ? 0 : 1;
        if (maybe[kk]) nblocks = std::max(nblocks, (unsigned int)((a[kk].npkt + bs - 1) / bs));
    }
    hipLaunchKernelGGL(inline_fallback ? slab_prepare_kernel<true> : slab_prepare_kernel<false>, dim3(nblocks, ngulp), dim3(bs), 0, stream, job, site.tally,
                       site.fallbacks, site.fallbacks_host, descs, args_out);
    XENG_HIP(hipGetLastError());
    return XENG_STATUS_SUCCESS;
}

int slab_fallback_enqueue(hipStream_t stream, const GulpDesc* descs, const SlabArgs* args, int ngulp) {
    // (both return at once for every gulp whose descriptor does not say "scratch": small grids, grid-stride loops)
    hipLaunchKernelGGL(slab_clear_kernel, dim3(512, ngulp), dim3(256), 0, stream, descs, args);
    hipLaunchKernelGGL(slab_scatter_kernel, dim3(512, ngulp), dim3(256), 0, stream, descs, args);
    XENG_HIP(hipGetLastError());
    return XENG_STATUS_SUCCESS;
}

int slab_site_read_fallbacks(hipStream_t stream, const SlabSite& site, int* n) {
    XENG_HIP(hipMemcpyAsync(n, site.fallbacks, sizeof(int), hipMemcpyDeviceToHost, stream));
    XENG_HIP(hipMemsetAsync(site.fallbacks, 0, sizeof(int), stream));
    XENG_HIP(hipStreamSynchronize(stream));
    return XENG_STATUS_SUCCESS;
}

}  // namespace xeng
