// Packet slabs as gulps (round 4): descriptors and the helper passes behind xengXgpuKernelAsyncSlab and xengBeamformRunSlabs.
//
// The F-engines send, per time sample, one packet per group of 64 inputs: 32-byte big-endian header `>QLHHHHLLL` (seq, sync_time,
// npol, npol_tot, nchan, nchan_tot, chan_block_id, chan0, pol0) + payload u8[nchan][npol] (test_transmitters/test_tx_vectors.py:
// 38-48,103-108; test_tx_mt.c:39-49).  A receiver that stores them in arrival order produces, when nothing is lost or
// reordered, a REGULAR slab: packet (t, b) at index t * nblk + b.  Such a slab already is the gulp, in another order of the same
// bytes: sample t, channel c, input block b at  slab + 32 + (t * nblk + b) * stride + c * 64  -- and the contraction kernel can
// read it there (GulpDesc below) instead of reading a copy that a scatter pass made.  These kernels decide that on
// the device, without a host round trip:
//   slab_prepare_kernel   one thread per packet: is packet p the packet (p / nblk, p % nblk) of this gulp?  The last group to
//                         finish writes the gulp's descriptor -- the slab itself, or (any packet out of place, lost, foreign,
//                         duplicated) the scratch gulp below
//   slab_clear_kernel, slab_scatter_kernel   once per integration, for the gulps whose descriptor says "scratch": zero-fill +
//                         scatter with the validation rules of snap2_unpack_kernel (ingest.hip); otherwise they return at once
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace xeng {

// Where the bytes of one gulp lie: sample t, channel c, 64-input block b, byte j of the block at
//   base + t * t_stride + c * c_stride + b * b_stride + j.
// A time-major gulp u8[t][c][input]: (nchan * ninput, ninput, 64).  A regular slab of SNAP2 packets -- packet (t, b) at index
// t * nblocks + b, 32-byte header + payload [nchan][64 inputs] (test_tx_vectors.py:38-48,103-108) --: base = slab + 32,
// (nblocks * pkt_stride, 64, pkt_stride): the contraction reads the packets where they lie, no scatter pass.
struct GulpDesc {
    const uint8_t* base;
    uint32_t t_stride, c_stride, b_stride, pad;
    uint64_t pad2;
};
static_assert(sizeof(GulpDesc) == 32, "GulpDesc is read as eight aligned dwords");

struct SlabArgs {
    const uint8_t* pkts;
    int npkt;
    uint32_t stride;
    unsigned long long seq0;
    int ntime, chan0, nchan, ninput, nblk;      // ntime: samples of THIS gulp (its scratch copy holds ntime * nchan * ninput bytes)
};

// One consumer's device-side state (slab.hip): every consumer verifies on its own stream, with its own tally.
struct SlabSite {
    unsigned long long* tally = nullptr;     // the 64-bit words of slab_prepare_kernel (one per gulp of a launch)
    int* fallbacks = nullptr;                // gulps that took the scratch path since they were last read
    int* fallbacks_host = nullptr;           // ... and ever, in pinned host memory: read by the host without a wait (a hint: is the link losing packets?)
};
int slab_site_create(SlabSite* s);
void slab_site_destroy(SlabSite* s);
// could this slab be regular at all?  (whole 64-input blocks, one packet per (sample, block), payload rows of 64 bytes, 16-byte
// pieces, and 32-bit per-lane offsets that hold `rows` sample rows)
bool slab_maybe_regular(const SlabArgs& a, int rows);
// enqueue on `stream`, ONE launch: verify ngulp (1 or 2) slabs and write descs[k] (the slab itself, or scratch[k]) and args_out[k].
// inline_fallback: an irregular gulp is also zero-filled and scattered by this launch (one work-group: slow, rare); otherwise
// slab_fallback_enqueue has to follow
int slab_prepare_enqueue(hipStream_t stream, const SlabSite& site, const SlabArgs* a, const bool* maybe, int ngulp, GulpDesc* descs, SlabArgs* args_out,
                         uint8_t* const* scratch, bool inline_fallback);
// enqueue on `stream`, behind the prepare passes of these gulps: zero-fill + scatter of those whose descriptor says "scratch"
int slab_fallback_enqueue(hipStream_t stream, const GulpDesc* descs, const SlabArgs* args, int ngulp);
// gulps that took the scratch path since the last call (waits for the stream)
int slab_site_read_fallbacks(hipStream_t stream, const SlabSite& site, int* n);

}  // namespace xeng
