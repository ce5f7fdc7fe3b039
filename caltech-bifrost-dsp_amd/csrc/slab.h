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
    uint32_t t_stride, c_stride, b_stride, pad;      // pad: 0 read in place, 1 scratch gulp (clear + scatter precede), 2 the slab through its table
    const uint32_t* table;                           // (round 5, the contraction's table kernel only) the offset table of this gulp, below
};
static_assert(sizeof(GulpDesc) == 32, "GulpDesc is read as eight aligned dwords");

// (round 5) The contraction does not form addresses from the strides any more: it follows a TABLE, so that a slab with lost,
// shifted, reordered or duplicated packets is read where it lies as well -- no scatter pass.  One row of 16 dwords per (64-input
// block b, 96-sample stage sl, half h of the stage, row r8 of a piece): [n] for n = 0..5 the byte offset, from the gulp's base, of the
// payload row of sample t = 96 sl + 48 h + 8 n + r8 (plus a bias that undoes the instruction's immediate offset: SLAB_OFF_BIAS -
// 1024 (n & 3)); [6] bit n set: no packet carries that sample -- the offset then names valid bytes whose copy in LDS the kernel
// overwrites with zeros.  A wave fetches the 2 x 512 bytes of its two blocks per stage with ONE LDS-DMA instruction, three stages
// before it issues the pieces they describe.  Plain time-major gulps (scratch copies) use a static table of the same shape.
constexpr int SLAB_ROW_U32 = 16;
constexpr uint32_t SLAB_OFF_BIAS = 4096;
inline size_t slab_table_u32(int ntime, int nblk) { return (size_t)nblk * (size_t)(ntime / 96) * 2 * 8 * SLAB_ROW_U32; }

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
// (round 5) the beamformer on a lossy link: ONE launch per call as before, but instead of "regular or scatter" it enters every packet
// of the deployed geometry into the part's index -- u32[ntime][nblk], entry = generation << 20 | 1 + the LAST slab index that carries
// (sample, block); entries of earlier calls carry older generations and read as "nobody carries it", so nothing is cleared per call --
// and the part's descriptor names that index (pad 2: base = the slab, t_stride = the generation, b_stride = the packet stride): the
// beamformer kernels' TAB instantiations look every row up there and read it where it lies, or zeros.  Only packets of another geometry
// still send a part through zero-fill + scatter.
constexpr uint32_t SLAB_GEN_SHIFT = 20, SLAB_SLOT_MASK = (1u << SLAB_GEN_SHIFT) - 1u, SLAB_GEN_MAX = (1u << (32 - SLAB_GEN_SHIFT)) - 1u;
struct SlabIndexPrep {
    uint32_t* tab[2] = {nullptr, nullptr};   // [part][ntime_max * nblk]
    size_t tab_u32 = 0;
    uint32_t gen = 0;                        // generation of the last call (1 .. SLAB_GEN_MAX; at the wrap the indices are cleared)
    int* irregular = nullptr;                // parts read through an index that was not regular, since they were last read
    int* irregular_host = nullptr;           // ... and ever, in pinned host memory (the host's hint: is the link losing packets?)
};
int slab_index_prep_create(SlabIndexPrep* s, int ntime, int ninput);
void slab_index_prep_destroy(SlabIndexPrep* s);
// could this part be read through an index?  (whole 64-input blocks, 16-byte pieces, slot numbers below 2^20)
bool slab_index_prep_ok(const SlabArgs& a);
int slab_index_prepare_enqueue(hipStream_t stream, const SlabSite& site, SlabIndexPrep& ix, const SlabArgs* a, const bool* ok, int ngulp, GulpDesc* descs,
                               SlabArgs* args_out, uint8_t* const* scratch, bool inline_fallback);
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

// ---- (round 5) the X-engine's passes: index, table, (rarely) scatter -- once per integration, grid.y = gulp --------------------------
constexpr int SLAB_MAX_GULPS = 16;
struct SlabIndexSite {
    uint32_t* tab32 = nullptr;               // [gulp][ntime * nblk]: 1 + the LAST slab index that carries (sample, block); 0: nobody does
    uint32_t* meta = nullptr;                // [gulp][4]: {packets with another geometry, rows not in place, -, -}
    uint32_t* tables[2] = {nullptr, nullptr};// [area][gulp][slab_table_u32()]
    uint32_t* plain_table = nullptr;         // the table of a time-major gulp u8[ntime][nchan][ninput] (scratch copies under the table kernel)
    int* counters = nullptr;                 // {gulps scattered, gulps read through an irregular table} since they were last read
    int* hint_host = nullptr;                // pinned: gulps that were not regular, ever -- read by the host without a wait (is the link losing packets?)
    int ntime = 0, nblk = 0;
};
int slab_index_site_create(SlabIndexSite* s, int ntime, int nchan, int ninput);
void slab_index_site_destroy(SlabIndexSite* s);
struct SlabIndexJob {
    SlabArgs a[SLAB_MAX_GULPS];
    uint8_t* scratch[SLAB_MAX_GULPS];
    int force[SLAB_MAX_GULPS];               // the host already knows that this slab cannot be read in place
    int ngulp;
};
// enqueue on `stream`: descs[g] / args_out[g] for every gulp of the job (tables of staging area `area`); by_table: the contraction that
// follows reads every gulp through its table (xcorr_fused_kernel<.., TAB>), else by strides (irregular slabs are scattered)
int slab_index_enqueue(hipStream_t stream, const SlabIndexSite& site, int area, const SlabIndexJob& job, bool by_table, GulpDesc* descs, SlabArgs* args_out);
int slab_index_site_read(hipStream_t stream, const SlabIndexSite& site, int* nscattered, int* nirregular);
// could this slab be read through a table?  (whole 64-input blocks, payload rows of 64 bytes in 16-byte pieces, 31-bit offsets)
bool slab_indexable(const SlabArgs& a);

}  // namespace xeng
