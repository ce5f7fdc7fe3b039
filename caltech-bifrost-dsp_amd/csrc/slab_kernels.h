// Packet slabs as gulps (round 4): the helper kernels behind xengXgpuKernelAsyncSlab.
//
// The F-engines send, per time sample, one packet per group of 64 inputs: 32-byte big-endian header `>QLHHHHLLL` (seq, sync_time,
// npol, npol_tot, nchan, nchan_tot, chan_block_id, chan0, pol0) + payload u8[nchan][npol] (test_transmitters/test_tx_vectors.py:
// 38-48,103-108; test_tx_mt.c:39-49).  A receiver that stores them in arrival order produces, when nothing is lost or
// reordered, a REGULAR slab: packet (t, b) at index t * nblk + b.  Such a slab already is the gulp, in another order of the same
// bytes: sample t, channel c, input block b at  slab + 32 + (t * nblk + b) * stride + c * 64  -- and the contraction kernel can
// read it there (GulpDesc, xcorr_kernels.h) instead of reading a copy that a scatter pass made.  These kernels decide that on
// the device, without a host round trip:
//   slab_prepare_kernel   one thread per packet: is packet p the packet (p / nblk, p % nblk) of this gulp?  The last group to
//                         finish writes the gulp's descriptor -- the slab itself, or (any packet out of place, lost, foreign,
//                         duplicated) the scratch gulp below
//   slab_clear_kernel, slab_scatter_kernel   once per integration, for the gulps whose descriptor says "scratch": zero-fill +
//                         scatter with the validation rules of snap2_unpack_kernel (ingest.hip); otherwise they return at once
#pragma once
#include <stdint.h>

#include "xcorr_kernels.h"

namespace xeng {

struct SlabArgs {
    const uint8_t* pkts;
    int npkt;
    uint32_t stride;
    unsigned long long seq0;
    int ntime, chan0, nchan, ninput, nblk;
};

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
__global__ __launch_bounds__(256) void slab_clear_kernel(const GulpDesc* __restrict__ desc, size_t n16) {
    const GulpDesc& d = desc[blockIdx.y];
    if (!d.pad) return;
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

}  // namespace xeng
