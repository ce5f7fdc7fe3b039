// Packet slabs as gulps: the device passes (slab.h) shared by the X-engine (xcorr.hip) and the beamformer (beamform.hip).
#include "slab.h"

#include <algorithm>
#include <stdlib.h>
#include <vector>

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
            d.table = nullptr;
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

// (round 5) the same launch for the beamformer's TAB kernels (slab.h, SlabIndexPrep): every packet of the deployed geometry is entered
// into the part's index; the wave that takes the last ticket knows whether any valid packet had another geometry (then: the scratch
// gulp, as above) and whether every (sample, block) sits in its regular slot (statistics and the host's hint only: the kernels follow
// the index either way).  Tally word: packets in place << 32 | waves that saw another geometry << 16 | ticket.
struct SlabIndexPrepJob {
    SlabArgs a[2];
    uint8_t* scratch[2];
    int force[2];
    uint32_t* tab[2];
    uint32_t gen;
};
template <bool INLINE>
__global__ __launch_bounds__(256) void slab_index_prepare_kernel(SlabIndexPrepJob job, unsigned long long* __restrict__ tallies, int* __restrict__ fallbacks,
                                                                 int* __restrict__ fallbacks_host, int* __restrict__ irregular, int* __restrict__ irregular_host,
                                                                 GulpDesc* __restrict__ descs, SlabArgs* __restrict__ args_out) {
    __shared__ int s_fallback_here;
    const int k = blockIdx.y;
    const SlabArgs& a = job.a[k];
    const int force_scratch = job.force[k];
    const unsigned int nblocks = force_scratch ? 1u : (unsigned int)((a.npkt + (int)blockDim.x - 1) / (int)blockDim.x);
    if (blockIdx.x >= nblocks) return;
    if (INLINE) {
        if (threadIdx.x == 0) s_fallback_here = 0;
        __syncthreads();
    }
    bool in_place = false, other = false;
    if (!force_scratch) {
        const int p = blockIdx.x * blockDim.x + threadIdx.x;
        if (p < a.npkt) {
            const SlabHeader h = slab_header(a.pkts + (size_t)p * a.stride, a.chan0);
            const int payload_max = (int)a.stride - 32;
            const bool ok = h.seq >= a.seq0 && h.seq - a.seq0 < (unsigned long long)a.ntime && h.npol > 0 && h.nchan > 0 && h.chan0 >= 0 &&
                            h.chan0 + h.nchan <= a.nchan && h.pol0 + h.npol <= a.ninput && (long long)h.nchan * h.npol <= payload_max;
            if (ok) {
                if (h.npol != 64 || h.nchan != a.nchan || h.chan0 != 0 || (h.pol0 & 63) != 0) other = true;
                else {
                    const unsigned int home = (unsigned int)(h.seq - a.seq0) * (unsigned int)a.nblk + (unsigned int)(h.pol0 >> 6);
                    atomicMax(&job.tab[k][home], (job.gen << SLAB_GEN_SHIFT) | ((uint32_t)p + 1u));
                    in_place = home == (unsigned int)p;
                }
            }
        }
    }
    const unsigned long long nin = (unsigned long long)__popcll(__ballot(in_place)), noth = __ballot(other) != 0ull ? 1ull : 0ull;
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long nwaves = (unsigned long long)nblocks * (blockDim.x >> 6);
        const unsigned long long before = atomicAdd(&tallies[k], (nin << 32) | (noth << 16) | 1ull);
        if ((before & 0xFFFFull) == nwaves - 1) {
            const bool fb = force_scratch || (((before >> 16) & 0xFFFFull) + noth) != 0;
            const bool regular = (before >> 32) + nin == (unsigned long long)a.ntime * a.nblk;
            GulpDesc d;
            if (fb) {
                d.base = job.scratch[k]; d.t_stride = (uint32_t)a.nchan * (uint32_t)a.ninput; d.c_stride = (uint32_t)a.ninput; d.b_stride = 64; d.pad = 1u;
                d.table = nullptr;
                atomicAdd(fallbacks, 1);
                __hip_atomic_fetch_add(fallbacks_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                d.base = a.pkts; d.t_stride = job.gen; d.c_stride = 64; d.b_stride = a.stride; d.pad = 2u;
                d.table = job.tab[k];
                if (!regular) atomicAdd(irregular, 1);
            }
            if (fb || !regular) __hip_atomic_fetch_add(irregular_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            descs[k] = d;
            args_out[k] = a;
            tallies[k] = 0;
            if (INLINE && fb) s_fallback_here = 1;
        }
    }
    if (!INLINE) return;
    __syncthreads();
    if (!s_fallback_here) return;
    slab_clear_part(a, job.scratch[k], threadIdx.x, blockDim.x);
    __syncthreads();
    slab_scatter_part(a, job.scratch[k], threadIdx.x >> 6, blockDim.x >> 6, threadIdx.x & 63);
}

// Once per integration, behind the prepare kernels of its gulps (grid.y = gulp): zero-fill and scatter of the gulps whose
// descriptor says "scratch"; the groups of every other gulp return at once.
__global__ __launch_bounds__(256) void slab_clear_kernel(const GulpDesc* __restrict__ desc, const SlabArgs* __restrict__ args) {
    const GulpDesc& d = desc[blockIdx.y];
    if (d.pad != 1u) return;          // (0: read in place, 2: through its table)
    slab_clear_part(args[blockIdx.y], const_cast<uint8_t*>(d.base), (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ __launch_bounds__(256) void slab_scatter_kernel(const GulpDesc* __restrict__ desc, const SlabArgs* __restrict__ args) {
    const GulpDesc& d = desc[blockIdx.y];
    if (d.pad != 1u) return;          // (0: read in place, 2: through its table)
    const SlabArgs a = args[blockIdx.y];
    slab_scatter_part(a, const_cast<uint8_t*>(d.base), blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, threadIdx.x & 63);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (round 5) The X-engine's passes.  Instead of deciding "regular or scatter", every slab gets an offset TABLE (slab.h): where the
// payload row of every (sample, 64-input block) lies, whatever the order the packets arrived in, and which samples nobody carries.
// While the link is clean the contraction addresses the slabs by their strides as before (and the first slab that is not regular goes
// through the round-4 scatter); from then on it follows the tables, so lost, shifted, reordered and duplicated packets cost nothing but
// the table; only packets of another geometry (two channel blocks per sample, 32 inputs per packet ...) still send a gulp through
// zero-fill + scatter.
//   slab_index_clear_kernel   tab32 = 0
//   slab_index_kernel         one thread per packet: a valid packet of the deployed geometry takes its (sample, block) entry with
//                             atomicMax(index + 1) -- the LAST packet that carries a sample wins, as in the oracle's scatter (later
//                             packets overwrite earlier ones); a valid packet of another geometry flags the gulp
//   slab_table_kernel         one thread per table row; notes whether any sample is not where a regular slab has it
//   slab_index_finish_kernel  the gulp's descriptor -- by strides, by table, or the scratch gulp: see there --, counters; zero-fill of
//                             the gulps that are scattered (slab_scatter_kernel follows)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_index_clear_kernel(SlabIndexJob job, uint32_t* __restrict__ tab32, uint32_t* __restrict__ meta, int per_gulp) {
    const int g = blockIdx.y;
    const int n = job.a[g].ntime * job.a[g].nblk;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) tab32[(size_t)g * per_gulp + k] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 4) meta[g * 4 + threadIdx.x] = (threadIdx.x == 0 && job.force[g]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void slab_index_kernel(SlabIndexJob job, uint32_t* __restrict__ tab32, uint32_t* __restrict__ meta, int per_gulp) {
    const int g = blockIdx.y;
    if (job.force[g]) return;
    const SlabArgs& a = job.a[g];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.npkt) return;
    const SlabHeader h = slab_header(a.pkts + (size_t)p * a.stride, a.chan0);
    const int payload_max = (int)a.stride - 32;
    const bool ok = h.seq >= a.seq0 && h.seq - a.seq0 < (unsigned long long)a.ntime && h.npol > 0 && h.nchan > 0 && h.chan0 >= 0 &&
                    h.chan0 + h.nchan <= a.nchan && h.pol0 + h.npol <= a.ninput && (long long)h.nchan * h.npol <= payload_max;
    if (!ok) return;                                            // (dropped, as the scatter drops it)
    if (h.npol != 64 || h.nchan != a.nchan || h.chan0 != 0 || (h.pol0 & 63) != 0) { atomicOr(&meta[g * 4], 1u); return; }
    const unsigned int home = (unsigned int)(h.seq - a.seq0) * (unsigned int)a.nblk + (unsigned int)(h.pol0 >> 6);
    atomicMax(&tab32[(size_t)g * per_gulp + home], (uint32_t)p + 1u);
}

__global__ __launch_bounds__(256) void slab_table_kernel(SlabIndexJob job, const uint32_t* __restrict__ tab32, uint32_t* __restrict__ meta, int per_gulp,
                                                         uint32_t* __restrict__ tables, size_t table_u32) {
    const int g = blockIdx.y;
    const SlabArgs& a = job.a[g];
    if (meta[g * 4] != 0) return;                             // (packets of another geometry: this gulp is scattered)
    const int spg = a.ntime / 96, nrow = a.nblk * spg * 16;
    const int row = blockIdx.x * 256 + threadIdx.x;
    bool irregular = false;
    if (row < nrow) {
        const int r8 = row & 7, h = (row >> 3) & 1, sl = (row >> 4) % spg, b = (row >> 4) / spg;
        uint32_t w[SLAB_ROW_U32];
#pragma unroll
        for (int k = 0; k < SLAB_ROW_U32; k++) w[k] = 0u;
#pragma unroll
        for (int n = 0; n < 6; n++) {
            const int t = sl * 96 + 48 * h + 8 * n + r8;
            const uint32_t e = tab32[(size_t)g * per_gulp + (size_t)t * a.nblk + b];
            const uint32_t slot = e ? e - 1u : 0u;               // (nobody carries it: any valid bytes -- the kernel zeroes their copy)
            if (!e) w[6] |= 1u << n;
            irregular = irregular || !e || slot != (uint32_t)(t * a.nblk + b);
            w[n] = slot * a.stride + 32u + SLAB_OFF_BIAS - 1024u * (uint32_t)(n & 3);
        }
        uint4* dst = reinterpret_cast<uint4*>(tables + (size_t)g * table_u32 + (size_t)row * SLAB_ROW_U32);
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    }
    if (__ballot(irregular) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&meta[g * 4 + 1], 1u);
}

// the verdict (one thread per gulp).  by_table = 0 (the contraction kernel that follows addresses gulps by strides: the link has been
// clean): every packet in place -> the slab by its strides, as in round 4; anything else -> the scratch gulp, zero-filled here and
// scattered by the next launch, and the host is told (a counter in pinned memory, read there without a wait: its next launches read
// every gulp through its table).  by_table = 1: a slab -> through its table, whatever the order of its packets; only packets of another
// geometry -> the scratch gulp, through the static table of a time-major gulp.
__global__ __launch_bounds__(256) void slab_index_finish_kernel(SlabIndexJob job, int by_table, const uint32_t* __restrict__ meta, uint32_t* __restrict__ tables,
                                                                size_t table_u32, const uint32_t* __restrict__ plain_table, GulpDesc* __restrict__ descs,
                                                                SlabArgs* __restrict__ args_out, int* __restrict__ counters, int* __restrict__ hint_host) {
    const int g = blockIdx.y;
    const SlabArgs& a = job.a[g];
    const bool partial = meta[g * 4] != 0, irregular = meta[g * 4 + 1] != 0;
    const bool scatter = partial || (irregular && !by_table);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        GulpDesc d;
        if (scatter) {
            d.base = job.scratch[g]; d.t_stride = (uint32_t)a.nchan * (uint32_t)a.ninput; d.c_stride = (uint32_t)a.ninput; d.b_stride = 64; d.pad = 1u;
            d.table = plain_table;
            atomicAdd(&counters[0], 1);
        } else if (by_table) {
            d.base = a.pkts; d.t_stride = (uint32_t)a.nblk * a.stride; d.c_stride = 64; d.b_stride = a.stride; d.pad = 2u;
            d.table = tables + (size_t)g * table_u32;
            if (irregular) atomicAdd(&counters[1], 1);
        } else {
            d.base = a.pkts + 32; d.t_stride = (uint32_t)a.nblk * a.stride; d.c_stride = 64; d.b_stride = a.stride; d.pad = 0u;
            d.table = nullptr;
        }
        descs[g] = d;
        args_out[g] = a;
        if (partial || irregular) __hip_atomic_fetch_add(hint_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!scatter) return;
    slab_clear_part(a, job.scratch[g], (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

int slab_index_site_create(SlabIndexSite* s, int ntime, int nchan, int ninput) {
    s->ntime = ntime; s->nblk = ninput / 64;
    const size_t per_gulp = (size_t)ntime * s->nblk, tu = slab_table_u32(ntime, s->nblk);
    XENG_HIP(hipMalloc((void**)&s->tab32, SLAB_MAX_GULPS * per_gulp * 4));
    XENG_HIP(hipMalloc((void**)&s->meta, SLAB_MAX_GULPS * 4 * 4 + 16));
    XENG_HIP(hip_memset_now(s->meta, 0, SLAB_MAX_GULPS * 4 * 4 + 16));
    s->counters = (int*)(s->meta + SLAB_MAX_GULPS * 4);
    for (int b = 0; b < 2; b++) {
        XENG_HIP(hipMalloc((void**)&s->tables[b], SLAB_MAX_GULPS * tu * 4));
        XENG_HIP(hip_memset_now(s->tables[b], 0, SLAB_MAX_GULPS * tu * 4));
    }
    // the table of a time-major gulp (scratch copies): sample t, block b at t * nchan * ninput + 64 b
    std::vector<uint32_t> pt(tu, 0u);
    const int spg = ntime / 96;
    for (int b = 0; b < s->nblk; b++)
        for (int sl = 0; sl < spg; sl++)
            for (int h = 0; h < 2; h++)
                for (int r8 = 0; r8 < 8; r8++) {
                    uint32_t* w = pt.data() + ((((size_t)b * spg + sl) * 2 + h) * 8 + r8) * SLAB_ROW_U32;
                    for (int n = 0; n < 6; n++) {
                        const size_t t = (size_t)sl * 96 + 48 * h + 8 * n + r8;
                        w[n] = (uint32_t)(t * nchan * ninput + 64u * b) + SLAB_OFF_BIAS - 1024u * (uint32_t)(n & 3);
                    }
                }
    XENG_HIP(hipMalloc((void**)&s->plain_table, tu * 4));
    XENG_HIP(hipMemcpy(s->plain_table, pt.data(), tu * 4, hipMemcpyHostToDevice));
    XENG_HIP(hipHostMalloc((void**)&s->hint_host, sizeof(int)));
    *s->hint_host = 0;
    return XENG_STATUS_SUCCESS;
}

void slab_index_site_destroy(SlabIndexSite* s) {
    if (s->plain_table) (void)hipFree(s->plain_table);
    if (s->hint_host) (void)hipHostFree(s->hint_host);
    if (s->tab32) (void)hipFree(s->tab32);
    if (s->meta) (void)hipFree(s->meta);
    for (int b = 0; b < 2; b++) if (s->tables[b]) (void)hipFree(s->tables[b]);
    *s = SlabIndexSite();
}

bool slab_indexable(const SlabArgs& a) {
    return a.ninput > 0 && a.ninput % 64 == 0 && a.ntime % 96 == 0 && a.npkt > 0 && a.npkt < (1 << 24) && a.stride >= 32 + (size_t)a.nchan * 64 &&
           a.stride % 16 == 0 && ((uintptr_t)a.pkts & 15) == 0 && (uint64_t)a.npkt * a.stride + SLAB_OFF_BIAS + 64 < (1ull << 31) &&
           (uint64_t)a.nblk * a.stride * 96 < (1ull << 31);       // (a regular slab by its strides: 32-bit per-lane offsets over the rows of a stage)
}

int slab_index_enqueue(hipStream_t stream, const SlabIndexSite& site, int area, const SlabIndexJob& job, bool by_table, GulpDesc* descs, SlabArgs* args_out) {
    const int per_gulp = site.ntime * site.nblk;
    const size_t tu = slab_table_u32(site.ntime, site.nblk);
    int npkt_max = 1;
    for (int g = 0; g < job.ngulp; g++) npkt_max = std::max(npkt_max, job.a[g].npkt);
    const int nrow = site.nblk * (site.ntime / 96) * 16;
    hipLaunchKernelGGL(slab_index_clear_kernel, dim3((per_gulp + 1023) / 1024, job.ngulp), dim3(256), 0, stream, job, site.tab32, site.meta, per_gulp);
    hipLaunchKernelGGL(slab_index_kernel, dim3((npkt_max + 255) / 256, job.ngulp), dim3(256), 0, stream, job, site.tab32, site.meta, per_gulp);
    hipLaunchKernelGGL(slab_table_kernel, dim3((nrow + 255) / 256, job.ngulp), dim3(256), 0, stream, job, (const uint32_t*)site.tab32, site.meta, per_gulp,
                       site.tables[area], tu);
    hipLaunchKernelGGL(slab_index_finish_kernel, dim3(512, job.ngulp), dim3(256), 0, stream, job, by_table ? 1 : 0, (const uint32_t*)site.meta, site.tables[area], tu,
                       (const uint32_t*)site.plain_table, descs, args_out, site.counters, site.hint_host);
    hipLaunchKernelGGL(slab_scatter_kernel, dim3(512, job.ngulp), dim3(256), 0, stream, (const GulpDesc*)descs, (const SlabArgs*)args_out);
    XENG_HIP(hipGetLastError());
    return XENG_STATUS_SUCCESS;
}

int slab_index_site_read(hipStream_t stream, const SlabIndexSite& site, int* nscattered, int* nirregular) {
    int v[2] = {0, 0};
    XENG_HIP(hipMemcpyAsync(v, site.counters, sizeof(v), hipMemcpyDeviceToHost, stream));
    XENG_HIP(hipMemsetAsync(site.counters, 0, sizeof(v), stream));
    XENG_HIP(hipStreamSynchronize(stream));
    if (nscattered) *nscattered = v[0];
    if (nirregular) *nirregular = v[1];
    return XENG_STATUS_SUCCESS;
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

int slab_index_prep_create(SlabIndexPrep* s, int ntime, int ninput) {
    s->tab_u32 = (size_t)ntime * (size_t)(ninput / 64);
    for (int k = 0; k < 2; k++) {
        XENG_HIP(hipMalloc((void**)&s->tab[k], s->tab_u32 * 4));
        XENG_HIP(hip_memset_now(s->tab[k], 0, s->tab_u32 * 4));
    }
    XENG_HIP(hipMalloc((void**)&s->irregular, 16));
    XENG_HIP(hip_memset_now(s->irregular, 0, 16));
    XENG_HIP(hipHostMalloc((void**)&s->irregular_host, sizeof(int)));
    *s->irregular_host = 0;
    s->gen = 0;
    // (test hook: start near the wrap of the 12-bit generation, 4095 calls away otherwise -- tests/test_slab_gpu.py)
    if (const char* g0 = getenv("XENG_SLAB_GEN0")) s->gen = (uint32_t)std::min<long>(std::max<long>(atol(g0), 0), (long)SLAB_GEN_MAX);
    return XENG_STATUS_SUCCESS;
}

void slab_index_prep_destroy(SlabIndexPrep* s) {
    for (int k = 0; k < 2; k++) if (s->tab[k]) (void)hipFree(s->tab[k]);
    if (s->irregular) (void)hipFree(s->irregular);
    if (s->irregular_host) (void)hipHostFree(s->irregular_host);
    *s = SlabIndexPrep();
}

bool slab_index_prep_ok(const SlabArgs& a) {
    return a.ninput > 0 && a.ninput % 64 == 0 && a.npkt > 0 && (uint32_t)a.npkt < SLAB_SLOT_MASK && a.stride >= 32 + (size_t)a.nchan * 64 && a.stride % 16 == 0 &&
           ((uintptr_t)a.pkts & 15) == 0;
}

int slab_index_prepare_enqueue(hipStream_t stream, const SlabSite& site, SlabIndexPrep& ix, const SlabArgs* a, const bool* ok, int ngulp, GulpDesc* descs,
                               SlabArgs* args_out, uint8_t* const* scratch, bool inline_fallback) {
    if (ix.gen >= SLAB_GEN_MAX) {            // the generation wraps: entries of 4095 calls ago must not come back to life
        for (int k = 0; k < 2; k++) XENG_HIP(hipMemsetAsync(ix.tab[k], 0, ix.tab_u32 * 4, stream));
        ix.gen = 0;
    }
    ix.gen++;
    SlabIndexPrepJob job;
    const int bs = 256;
    unsigned int nblocks = 1;
    for (int k = 0; k < 2; k++) {
        const int kk = k < ngulp ? k : 0;
        job.a[k] = a[kk]; job.scratch[k] = scratch[kk]; job.force[k] = ok[kk] ? 0 : 1; job.tab[k] = ix.tab[k];
        if (ok[kk]) nblocks = std::max(nblocks, (unsigned int)((a[kk].npkt + bs - 1) / bs));
    }
    job.gen = ix.gen;
    hipLaunchKernelGGL(inline_fallback ? slab_index_prepare_kernel<true> : slab_index_prepare_kernel<false>, dim3(nblocks, ngulp), dim3(bs), 0, stream, job, site.tally,
                       site.fallbacks, site.fallbacks_host, ix.irregular, ix.irregular_host, descs, args_out);
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
