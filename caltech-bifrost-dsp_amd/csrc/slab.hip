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

// verify + describe in one launch.  Every wave checks its 64 packets and adds ONE word to the tally: (packets out of place) << 32
// | 1.  The wave that reads back "all other waves have added theirs" has, in the same word, the gulp's total: it writes the
// descriptor and re-arms the tally.  One returning atomic per wave carries both the count and the ticket, so nothing has to be
// ordered and the kernel needs NO fence: an agent-scope fence on this part is an L2 write-back + invalidate on every XCD, and
// beside a running contraction (which lives on L2 hits of the gulps) that cost 7 % of the streaming rate (measured:
// profiles/r04/slab_paths.txt).  No LDS either.
// tally: the 64-bit word; fallbacks: gulps that took the scratch path (read by xengXgpuGetSlabFallbacks).
// force_scratch: the host already knows the slab cannot be regular (packet count, stride, alignment); then one wave, no check.
__global__ __launch_bounds__(256) void slab_prepare_kernel(SlabArgs a, unsigned long long* __restrict__ tally, int* __restrict__ fallbacks,
                                                           GulpDesc* __restrict__ desc, SlabArgs* __restrict__ args_out, uint8_t* scratch,
                                                           int force_scratch) {
    bool ok = true;
    if (!force_scratch) {
        const int p = blockIdx.x * 256 + threadIdx.x;
        if (p < a.npkt) {
            const SlabHeader h = slab_header(a.pkts + (size_t)p * a.stride, a.chan0);
            const int t = p / a.nblk, b = p % a.nblk;
            ok = h.seq == a.seq0 + (unsigned long long)t && h.pol0 == (long long)b * 64 && h.npol == 64 && h.nchan == a.nchan && h.chan0 == 0;
        }
    }
    const unsigned long long nbad = (unsigned long long)__popcll(__ballot(!ok));
    if ((threadIdx.x & 63) != 0) return;
    const unsigned long long nwaves = (unsigned long long)gridDim.x * (blockDim.x >> 6);
    const unsigned long long before = atomicAdd(tally, (nbad << 32) | 1ull);
    if ((before & 0xFFFFFFFFull) != nwaves - 1) return;
    const bool fb = force_scratch || ((before >> 32) + nbad) != 0;
    GulpDesc d;
    if (fb) {
        d.base = scratch; d.t_stride = (uint32_t)a.nchan * (uint32_t)a.ninput; d.c_stride = (uint32_t)a.ninput; d.b_stride = 64;
        atomicAdd(fallbacks, 1);
    } else {
        d.base = a.pkts + 32; d.t_stride = (uint32_t)a.nblk * a.stride; d.c_stride = 64; d.b_stride = a.stride;
    }
    d.pad = fb ? 1u : 0u;
    d.pad2 = 0;
    *desc = d;
    *args_out = a;            // (for the scatter at flush time, should this gulp need it)
    *tally = 0;               // (re-armed for the next gulp: launches on one stream, in order)
}

// Once per integration, behind the prepare kernels of its gulps (grid.y = gulp): zero-fill and scatter of the gulps whose
// descriptor says "scratch"; the groups of every other gulp return at once.
__global__ __launch_bounds__(256) void slab_clear_kernel(const GulpDesc* __restrict__ desc, const SlabArgs* __restrict__ args) {
    const GulpDesc& d = desc[blockIdx.y];
    if (!d.pad) return;
    const size_t n16 = (size_t)args[blockIdx.y].ntime * args[blockIdx.y].nchan * args[blockIdx.y].ninput / 16;     // (scratch gulps are whole 16-byte pieces: checked on the host)
    uint4* scratch = reinterpret_cast<uint4*>(const_cast<uint8_t*>(d.base));
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) scratch[k] = make_uint4(0, 0, 0, 0);
}

// one wave per packet (any order, duplicates allowed): the validation of snap2_unpack_kernel, rows of npol bytes
__global__ __launch_bounds__(256) void slab_scatter_kernel(const GulpDesc* __restrict__ desc, const SlabArgs* __restrict__ args) {
    const GulpDesc& d = desc[blockIdx.y];
    if (!d.pad) return;
    const SlabArgs a = args[blockIdx.y];
    uint8_t* scratch = const_cast<uint8_t*>(d.base);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int payload_max = (int)a.stride - 32;
    for (int p = blockIdx.x * 4 + wave; p < a.npkt; p += gridDim.x * 4) {
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


int slab_site_create(SlabSite* s) {
    void* p = nullptr;
    XENG_HIP(hipMalloc(&p, 16));
    XENG_HIP(hipMemset(p, 0, 16));
    s->tally = (unsigned long long*)p;
    s->fallbacks = (int*)p + 2;
    return XENG_STATUS_SUCCESS;
}

void slab_site_destroy(SlabSite* s) {
    if (s->tally) (void)hipFree(s->tally);
    *s = SlabSite();
}

bool slab_maybe_regular(const SlabArgs& a, int rows) {
    return a.ninput > 0 && a.ninput % 64 == 0 && a.npkt > 0 && a.npkt == a.ntime * a.nblk && a.stride >= 32 + (size_t)a.nchan * 64 &&
           a.stride % 16 == 0 && ((uintptr_t)a.pkts & 15) == 0 && (uint64_t)a.nblk * a.stride * rows < (1ull << 31);
}

int slab_prepare_enqueue(hipStream_t stream, const SlabSite& site, const SlabArgs& a, bool maybe, GulpDesc* desc, SlabArgs* args_out, uint8_t* scratch) {
    hipLaunchKernelGGL(slab_prepare_kernel, dim3(maybe ? (a.npkt + 255) / 256 : 1), dim3(256), 0, stream, a, site.tally, site.fallbacks, desc, args_out,
                       scratch, maybe ? 0 : 1);
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
