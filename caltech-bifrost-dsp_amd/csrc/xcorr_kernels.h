// Device kernels of the MI355X X-engine (gfx950 only).
//
// What they replace: the xGPU CUDA X-engine + its DEVSWIZZLE input swizzle that the
// reference reaches through _bf.bfXgpuKernel (corr_block.py:445; xGPU itself is an
// empty submodule in the reference tree).  Nothing here is derived from xGPU code.
//
// Gulps stay in HBM and all gulps of an integration (K = n_gulps * ntime) are contracted in ONE launch, so
// the 191 MB int32 accumulator is written once per integration instead of being read-modify-written every
// 480 samples.  Two paths:
//
//  default   xcorr_fused_kernel     (int8 MFMA; persistent work-groups, corner turn fused into the LDS staging)
//     in   the gulps themselves, uint8[t][c][i] 4+4 bit (corr_block.py:115-116), read where they lie:
//          LDS-DMA of 96-sample x 128-byte tiles, operand fragments through ds_read_b64_tr_b8.
//
//  two-pass  corner_turn_tr8_kernel / corner_turn_kernel   (HBM-bound, one launch per gulp)
//     out  stash[c][ib][kt][sub][lane][16 B]   "MFMA-fragment-major":
//          ib  = 64-input block, kt = 32-sample K tile, sub = 32-input half,
//          lane = h*32 + r holds the 16 samples t = kt*32 + 16h + (0..15) of input
//          i = ib*64 + sub*32 + r, still packed 4+4 bit.  One (kt, sub) fragment is
//          1 KiB and is byte-for-byte the A (or B) register image of
//          v_mfma_i32_32x32x32_i8, so the contraction kernel moves it HBM -> LDS with
//          linear global_load_lds_dwordx4 and LDS -> VGPR with conflict-free ds_read_b128.
//            xcorr_mfma_kernel      (int8 MFMA, one work-group per (channel, tile group))
//          (also: xcorr_fp6_kernel + corner_turn_fp6_kernel, the opt-in FP6 experiment)
//
// Common to the contraction kernels:
//     Nibbles are sign-extended "for free": (x & 0xF0F0F0F0) is 16*re as int8,
//     ((x<<4) & 0xF0F0F0F0) is 16*im; all products are exact multiples of 256 and
//     the epilogue shifts them back (>> 8).  No negated operand is needed: the
//     imaginary part is kept as two accumulators P = sum ai*br, Q = sum ar*bi and
//     subtracted in the epilogue (-(-8) does not fit int8 after the x16 scaling).
//     Bound: |acc| <= 2*128*128*K < 2^31  =>  K <= 65535 samples per launch.
//     The epilogue (xcorr_store_tile) writes the xGPU register-tile order (corr_block.py:27-58) directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "slab.h"

#include "xcorr_tiling.h"

namespace xeng {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int FRAG_BYTES = 1024;          // one 32-input x 32-sample fragment
constexpr int KT_BYTES = 2 * FRAG_BYTES;  // one 64-input block x 32 samples
constexpr int XC_NBUF = 3;                // LDS ring depth

// ---------------------------------------------------------------------------------------
// stage 1: corner turn (time-major 4+4 bit -> fragment-major stash)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void transpose4x4_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d,
                                                   uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
    // rows a..d = 4 consecutive samples, bytes = 4 consecutive inputs; o_j = input j, 4 samples
    uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
    uint32_t t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
    uint32_t u0 = __builtin_amdgcn_perm(d, c, 0x05010400u);  // c0 d0 c1 d1
    uint32_t u1 = __builtin_amdgcn_perm(d, c, 0x07030602u);  // c2 d2 c3 d3
    o0 = __builtin_amdgcn_perm(u0, t0, 0x05040100u);         // a0 b0 c0 d0
    o1 = __builtin_amdgcn_perm(u0, t0, 0x07060302u);         // a1 b1 c1 d1
    o2 = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
    o3 = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
}

// Register-only fallback (input counts that are not a multiple of 16).
// grid (gkt, nchan), one thread per work item (threads = 32*nblk64 rounded up to a wave, set by the
// launcher) so every load of the tile is in flight at once.  Each work item is (input quad q, k-half h): it reads
// 16 samples x 4 inputs as 16 coalesced dwords (a wave covers 256 contiguous bytes of one
// [t][c] row per load) and writes the four inputs' 16-byte fragment entries (64 contiguous B).
__global__ __launch_bounds__(1024) void corner_turn_kernel(const uint8_t* __restrict__ in,
                                                          uint8_t* __restrict__ stash, int ntime,
                                                          int nchan, int ninput, int nblk64,
                                                          int cap_kt, int kt_off) {
    const int kt = blockIdx.x, c = blockIdx.y;
    const int nq = nblk64 * 16;
    const size_t row_stride = (size_t)nchan * ninput;
    const uint8_t* src_c = in + (size_t)c * ninput;
    for (int item = threadIdx.x; item < 2 * nq; item += blockDim.x) {
        const int h = item / nq, q = item - h * nq;
        const int i0 = q * 4;
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int t = kt * 32 + 16 * h + j;
            v[j] = (i0 < ninput && t < ntime)
                       ? *reinterpret_cast<const uint32_t*>(src_c + (size_t)t * row_stride + i0)
                       : 0u;
        }
        uint32_t o[4][4];
#pragma unroll
        for (int g = 0; g < 4; g++)
            transpose4x4_bytes(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], o[0][g], o[1][g],
                               o[2][g], o[3][g]);
        const int ib = i0 >> 6, sub = (i0 >> 5) & 1, r = i0 & 31;
        uint8_t* dst = stash + ((((size_t)c * nblk64 + ib) * cap_kt + (kt_off + kt)) * 2 + sub) * FRAG_BYTES +
                       (h * 32 + r) * 16;
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<uint4*>(dst + 16 * j) = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
    }
}

// Corner turn on the hardware byte-transposing LDS read (default for input counts that are a multiple
// of 16).  ds_read_b64_tr_b8 (profiles/microbench/tr8_probe.hip): in every 16-lane group, lane 2q+p supplies
// the address of row q, columns 8p..8p+7 of an 8x16 byte block and lane i receives column i of the 8 rows.
// With rows = samples and columns = inputs, two such reads give a lane the 16 samples of one input:
// exactly one 16-byte fragment entry, which it stores straight to HBM (lanes of a wave cover two contiguous
// 512-byte runs).  No permutes, no output image in LDS.
//   grid (channel, K tile, k-half), 256 threads.  LDS: 16 rows at a pitch that is 8 mod 64 dwords, so the
//   transposing reads are bank-conflict-free; the rows arrive by LDS-DMA (the LDS image is linear per 1 KiB
//   piece; columns past ninput inside the pitch are padding).
__global__ __launch_bounds__(256) void corner_turn_tr8_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ stash, int ntime,
                                                              int nchan, int ninput, int nblk64,
                                                              int cap_kt, int kt_off, int pitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ct_lds[];
    const int c = blockIdx.x, kt = blockIdx.y, hz = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = blockDim.x >> 6;
    const size_t row_stride = (size_t)nchan * ninput;
    const uint8_t* src_c = in + (size_t)c * ninput;
    const int t_base = kt * 32 + 16 * hz;
    const int t_valid = max(0, min(16, ntime - t_base));

    // phase 1: LDS byte o = row * pitch + col
    const int npiece = (16 * pitch + 1023) >> 10;
    for (int n = wave; n < npiece; n += nwave) {
        const int off = n * 1024 + lane * 16;
        int t = off / pitch, i = off - t * pitch;
        if (t >= t_valid || i + 16 > ninput) { t = 0; i = 0; }     // padding / missing rows: any valid address
        const uint8_t* g = src_c + (size_t)(t_valid > 0 ? t_base + t : 0) * row_stride + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(ct_lds + n * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // byte masks for rows that do not exist (t >= t_valid): wave-uniform
    const int v0 = min(t_valid, 8), v1 = max(t_valid - 8, 0);
    const unsigned long long m0 = v0 >= 8 ? ~0ull : ((1ull << (8 * v0)) - 1), m1 = v1 >= 8 ? ~0ull : ((1ull << (8 * v1)) - 1);
    const int grp = lane >> 4, w = lane & 15;
    const int rd_off = (w >> 1) * pitch + (w & 1) * 8 + grp * 16;   // row q = w/2, columns 8*(w&1).. of this group's 16
    const int nstep = nblk64;                                        // 64 inputs per wave step
    for (int st = wave; st < nstep; st += nwave) {
        const int col0 = st * 64;
        // all 64 lanes execute the transposing reads (the instruction needs EXEC = all ones); columns past
        // ninput read padding, zeroed below
        v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(ct_lds + rd_off + col0));
        v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(ct_lds + rd_off + col0 + 8 * pitch));
        const int I = col0 + lane;                                   // this lane's input: column 16*grp + (lane&15)
        const bool live = I < ninput;
        uint4 val;
        val.x = live ? ((uint32_t)lo.x & (uint32_t)m0) : 0u;
        val.y = live ? ((uint32_t)lo.y & (uint32_t)(m0 >> 32)) : 0u;
        val.z = live ? ((uint32_t)hi.x & (uint32_t)m1) : 0u;
        val.w = live ? ((uint32_t)hi.y & (uint32_t)(m1 >> 32)) : 0u;
        // fragment entry (ib = st, sub = lane>>5, k-half hz, r = lane&31)
        uint8_t* dst = stash + ((((size_t)c * nblk64 + st) * cap_kt + (kt_off + kt)) * 2 + (lane >> 5)) * FRAG_BYTES +
                       (hz * 32 + (lane & 31)) * 16;
        *reinterpret_cast<uint4*>(dst) = val;
    }
}

// ---------------------------------------------------------------------------------------
// stage 2: int8-MFMA contraction + xGPU-order epilogue
// ---------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int dpp_xor1(int x) {
    return __builtin_amdgcn_mov_dpp(x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

__device__ __forceinline__ int64_t tri64(int64_t i, int64_t j) { return (i * (i + 1)) / 2 + j; }

constexpr int XC_MAX_GULPS = 16;
struct XcorrParams {
    const uint8_t* stash;
    int32_t* out;
    const WgDesc* descs;
    int nwg, nchan, nblk64, cap_kt, nkt, nstand;
    int64_t per_chan, matlen;
    int accumulate;
    unsigned long long* stamps;   // diagnostic only (null in production): per wave {d_memtime, d_memrealtime, loop cycles}
    // RAW variant: the gulps themselves (time-major, uint8[ntime][nchan][ninput]), read in place
    const uint8_t* gulps[XC_MAX_GULPS];
    int spg;                      // 96-sample stages per gulp
    int ninput;
    // fused kernel: fragment-level tile groups, per work-group list of items (WorkEntry[gridDim.x][maxi]), stages per item
    const FragGroup* fgroups;
    const uint32_t* work;
    int maxi;
    int nstage;
    // long accumulation fused into the dump (CorrAcc's "a = b" / "a += b", corr_acc_block.py:298-306, applied to the
    // values this launch stores): acc2 = planar int32 buffer like `out`; acc2_mode 0 none, 1 assign, 2 add
    int32_t* acc2;
    int acc2_mode;
    // DESC instantiation (round 4): the gulps of this launch are described in DEVICE memory, one GulpDesc each, written on the
    // staging stream before the launch -- a gulp may then be a slab of F-engine packets read where it lies (xengXgpuKernelAsyncSlab)
    const GulpDesc* gdesc;
    int by_table;                 // (round 5, host side only: launch the TAB instantiation -- every gulp through its offset table)
};


struct Frags {   // the 8 unpacked int8 operand fragments of one 64x64 wave tile and one K-tile
    v4i ar[2], ai[2], br[2], bi[2];
};
struct RawFrags {  // the same, still packed 4+4 bit (as read from LDS)
    v4i a[2], b[2];
};

__device__ __forceinline__ Frags unpack_frags(const RawFrags& r) {
    const v4i M = (v4i)(0xF0F0F0F0);
    Frags u;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        u.ar[m] = r.a[m] & M;          // 16 * re  (hi nibble in place, two's complement)
        u.ai[m] = (r.a[m] << 4) & M;   // 16 * im
        u.br[m] = r.b[m] & M;
        u.bi[m] = (r.b[m] << 4) & M;
    }
    return u;
}

// A/B (diagnostic builds, ABL & 256; results are wrong): operands as an offset-binary decode would supply them -- x ^ 0x88 read as
// unsigned nibbles 0..15, same two VALU ops per operand dword -- to measure what smaller, sign-free operands are worth at the
// power cap before anybody builds the row-sum corrections they would need (round-3 review, item 3)
__device__ __forceinline__ Frags unpack_frags_offset_binary(const RawFrags& r) {
    const v4i M = (v4i)(0x0F0F0F0F), X = (v4i)(0x88888888u);
    Frags u;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const v4i a = r.a[m] ^ X, b = r.b[m] ^ X;     // (one more VALU op than the real thing would need: the XOR could be folded into the ingest)
        u.ar[m] = (a >> 4) & M;
        u.ai[m] = a & M;
        u.br[m] = (b >> 4) & M;
        u.bi[m] = b & M;
    }
    return u;
}

constexpr int XC_KT = 3;      // K-tiles (32 samples each) per LDS stage
constexpr int XC_RING = 4;    // LDS ring depth (stages), two-pass kernel
#ifndef XF_NO_RELAX
#define XF_NO_RELAX 0         // 1: A/B build without the relaxed first-stage wait behind an epilogue
#endif
#ifndef XF_DEPTH
#define XF_DEPTH 3            // fused kernel: stages of LDS-DMA in flight ahead of the MFMAs (ring = XF_DEPTH+1 stages of 24 KiB; 3, 4, 5 measure the same)
#endif

// ---- epilogue shared by both contraction kernels: D[i][j] = sum x_i conj(x_j), lane = column j,
// register = row i.  MFMA C/D map (32x32): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Registers 4u..4u+3 of a lane are (station 2Rh, pol 0/1), (station 2Rh+1, pol 0/1) of one column
// (station C, pol = lane&1).  The even lane of a pair assembles the cell of station 2Rh, the odd lane
// the cell of station 2Rh+1; the two words each is missing (the other polC) come from its partner
// lane by DPP (quad_perm [1,0,3,2]):
//   even: {v0, odd.v0, v1, odd.v1}     odd: {even.v2, v2, even.v3, v3}
//
// A lane quad 4k..4k+3 then holds the four quadrants of column station pair k, i.e. four cells that lie
// a quarter of the matrix apart, while the cells of one quadrant and eight consecutive column pairs are
// contiguous (8 x 16 B = one 128-byte run).  Each 16-byte store of a lane would be its own request to
// L2; so the cells are first moved across lanes (ds_bpermute: lane 8q+k takes the cell of lane 4k+q in
// its 32-lane half) and every 8 adjacent lanes write one contiguous run.
// NT = 32-column MFMA tiles per wave (2: a 64x64 wave tile; 1: a 64x32 wave tile whose column half is n0).
// LACC: every cell that is stored is also assigned / added to the long accumulator p.acc2 (XcorrParams).
template <int NT = 2, bool LACC = false>
__device__ __forceinline__ void xcorr_store_tile(const XcorrParams& p, int c, int blk_a, int blk_b, bool skip01, int lane,
                                                 const v16i (&accR)[2][NT], const v16i (&accP)[2][NT],
                                                 const v16i (&accQ)[2][NT], bool add_to_stored = false, int n0 = 0) {
    const int qs = (int)(((int64_t)(p.nstand / 2 + 1) * p.nstand) / 4);
    int32_t* out_r = p.out + (int64_t)c * p.per_chan;
    int32_t* out_i = out_r + p.matlen;
    int32_t* acc_r = LACC ? p.acc2 + (int64_t)c * p.per_chan : nullptr;
    int32_t* acc_i = LACC ? acc_r + p.matlen : nullptr;
    const bool acc_add = LACC && p.acc2_mode == 2;
    auto long_acc = [&](int w, int4 cr, int4 ci) {           // the stored cell -> the long accumulator
        int4* ar = reinterpret_cast<int4*>(acc_r + w);
        int4* ai = reinterpret_cast<int4*>(acc_i + w);
        if (acc_add) {
            const int4 o_r = *ar, o_i = *ai;
            cr.x += o_r.x; cr.y += o_r.y; cr.z += o_r.z; cr.w += o_r.w;
            ci.x += o_i.x; ci.y += o_i.y; ci.z += o_i.z; ci.w += o_i.w;
        }
        *ar = cr;
        *ai = ci;
    };
    const int odd = lane & 1;
    auto cell = [&](int v0, int v1, int v2, int v3) {
        const int g0 = dpp_xor1(odd ? v0 : v2), g1 = dpp_xor1(odd ? v1 : v3);
        return odd ? make_int4(g0, v2, g1, v3) : make_int4(v0, g0, v1, g1);
    };
    const int pull = ((lane & 32) | (4 * (lane & 7) + ((lane >> 3) & 3))) * 4;   // byte address of the source lane
    auto regroup = [&](int4 v) {
        return make_int4(__builtin_amdgcn_ds_bpermute(pull, v.x), __builtin_amdgcn_ds_bpermute(pull, v.y),
                         __builtin_amdgcn_ds_bpermute(pull, v.z), __builtin_amdgcn_ds_bpermute(pull, v.w));
    };
    // identity of the cell this lane stores after the regrouping
    const int quad = (lane >> 3) & 3;          // 2*(C&1) + (R&1)
    const int cpar = quad >> 1, rpar = quad & 1;
    // interior tiles (strictly below the block diagonal, no padded inputs) need no per-cell mask
    const bool interior = __builtin_amdgcn_readfirstlane((int)(blk_a > blk_b && blk_a * 64 + 64 <= 2 * p.nstand)) != 0;
    const bool accumulate = p.accumulate != 0 || add_to_stored;
    if (interior && !accumulate) {
        // the common case (55 of 66 tiles, first flush of an integration) as straight-line code: no per-cell
        // branches, so the lane regrouping of one cell overlaps the arithmetic of the next.  (Regrouping through a
        // per-wave LDS scratch -- one ds_write_b128 + one ds_read_b128 per cell instead of four ds_bpermute -- measured
        // the same: the epilogue's cost is the write traffic, 14 % of the step by ablation, not its instruction stream.)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const int ibase = blk_a * 64 + m * 32, jbase = blk_b * 64 + (n0 + n) * 32;
                const int Ch = (jbase >> 2) + (lane & 7);
                const int wcol = (quad * qs + Ch) * 4;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int Rh = (ibase >> 2) + 2 * u + (lane >> 5);
                    int vr[4], vi[4];
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        vr[v] = accR[m][n][4 * u + v] >> 8;
                        vi[v] = (accP[m][n][4 * u + v] - accQ[m][n][4 * u + v]) >> 8;
                    }
                    const int4 cr = regroup(cell(vr[0], vr[1], vr[2], vr[3]));
                    const int4 ci = regroup(cell(vi[0], vi[1], vi[2], vi[3]));
                    const int w = wcol + ((Rh * (Rh + 1)) >> 1) * 4;
                    *reinterpret_cast<int4*>(out_r + w) = cr;
                    *reinterpret_cast<int4*>(out_i + w) = ci;
                    if (LACC) long_acc(w, cr, ci);
                }
            }
        return;
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < NT; n++) {
            if (m == 0 && n0 + n == 1 && skip01) continue;      // never stored (and not computed) on diagonal tiles
            const int ibase = blk_a * 64 + m * 32, jbase = blk_b * 64 + (n0 + n) * 32;
            const int Ch = (jbase >> 2) + (lane & 7);
            const int C = 2 * Ch + cpar;
            const int wcol = (quad * qs + Ch) * 4;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int Rh = (ibase >> 2) + 2 * u + (lane >> 5);
                const int R = 2 * Rh + rpar;
                int vr[4], vi[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    vr[v] = accR[m][n][4 * u + v] >> 8;
                    vi[v] = (accP[m][n][4 * u + v] - accQ[m][n][4 * u + v]) >> 8;
                }
                int4 cr = regroup(cell(vr[0], vr[1], vr[2], vr[3]));
                int4 ci = regroup(cell(vi[0], vi[1], vi[2], vi[3]));
                const int w = wcol + ((Rh * (Rh + 1)) >> 1) * 4;
                int4* pr = reinterpret_cast<int4*>(out_r + w);
                int4* pi = reinterpret_cast<int4*>(out_i + w);
                if (interior || (Rh >= Ch && R < p.nstand && C < p.nstand)) {
                    if (accumulate) {
                        const int4 o_r = *pr, o_i = *pi;
                        cr.x += o_r.x; cr.y += o_r.y; cr.z += o_r.z; cr.w += o_r.w;
                        ci.x += o_i.x; ci.y += o_i.y; ci.z += o_i.z; ci.w += o_i.w;
                    }
                    *pr = cr;
                    *pi = ci;
                    if (LACC) long_acc(w, cr, ci);
                }
            }
        }
}

// the 16 (12 on diagonal tiles) int8 MFMAs of one K-tile: R += ar*br + ai*bi, P += ai*br, Q += ar*bi
__device__ __forceinline__ void xcorr_mfma_tile(const Frags& u, bool skip01, v16i (&accR)[2][2], v16i (&accP)[2][2],
                                                v16i (&accQ)[2][2]) {
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            if (m == 0 && n == 1 && skip01) continue;
            accR[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ar[m], u.br[n], accR[m][n], 0, 0, 0);
            accP[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ai[m], u.br[n], accP[m][n], 0, 0, 0);
            accQ[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ar[m], u.bi[n], accQ[m][n], 0, 0, 0);
            accR[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ai[m], u.bi[n], accR[m][n], 0, 0, 0);
        }
}

// =======================================================================================
// Two-pass path, pass 2: operands come from the fragment-major staging area written by the corner turn.
// One work-group per (channel, tile group).
// ABL: timing-only ablation bits (results are wrong unless ABL == 0): 1 no LDS-DMA in the loop,
// 2 no nibble unpack, 4 no LDS reads in the loop, 8 no barrier/vmcnt wait in the loop.
// =======================================================================================
template <int ABL>
__global__ __launch_bounds__(256, 1) void xcorr_mfma_kernel(XcorrParams p) {
    constexpr int KT_STAGE = XC_KT;
    constexpr int SLOT_BYTES = KT_STAGE * KT_BYTES;
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES;
    constexpr int NLOAD = 2 * KT_STAGE;  // 1 KiB LDS-DMA pieces per wave per stage
    __shared__ __attribute__((aligned(16))) uint8_t lds[XC_RING * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long r_entry = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;

    // block -> (channel, work-group).  Blocks b and b+8 share an XCD (round-robin dispatch),
    // so give each XCD whole channels: all tiles of a channel then stream the same stash
    // rows through one L2.  Placement only affects speed.
    int c, wg;
    {
        const int b = blockIdx.x;
        if ((p.nchan & 7) == 0) {
            const int xcd = b & 7, slot = b >> 3;
            c = xcd + 8 * (slot / p.nwg);
            wg = slot % p.nwg;
        } else {
            c = b / p.nwg;
            wg = b % p.nwg;
        }
    }
    const WgDesc* dp = p.descs + wg;   // indexed in memory: no runtime-indexed register arrays
    const int a_slot = dp->wave_a[wave], b_slot = dp->wave_b[wave];
    const bool active = a_slot != 0xFF;
    const int blk_a = active ? dp->slot_blk[a_slot] : 0, blk_b = active ? dp->slot_blk[b_slot] : 0;

    // this wave stages 64-input block slot_blk[wave]: KT_STAGE*2 KiB contiguous per stage
    const uint8_t* gsrc =
        p.stash + ((size_t)c * p.nblk64 + dp->slot_blk[wave]) * (size_t)p.cap_kt * KT_BYTES + lane * 16;
    const int nstage = p.nkt / KT_STAGE;

    // one 1 KiB LDS-DMA piece n (0..NLOAD-1) of stage s into ring buffer s % XC_RING.  Stages past
    // the end re-read the last real stage (never consumed): the issue stays unconditional, so every
    // stage costs exactly NLOAD pieces on the vmcnt counter and the loop body is one scheduling region.
    auto issue_piece = [&](int s, int n) {
        const int ssrc = s < nstage ? s : nstage - 1;
        const uint8_t* g = gsrc + (size_t)ssrc * SLOT_BYTES + n * FRAG_BYTES;
        uint8_t* l = lds + (s & (XC_RING - 1)) * STAGE_BYTES + wave * SLOT_BYTES + n * FRAG_BYTES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    };

    v16i accR[2][2], accP[2][2], accQ[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            accR[m][n] = (v16i)(0);
            accP[m][n] = (v16i)(0);
            accQ[m][n] = (v16i)(0);
        }

    // Idle waves (a_slot == 0xFF) run the same loop on slot 0 and skip the epilogue: keeping the
    // MFMA chain unconditional keeps the 192 accumulator registers in place (a wave-uniform
    // branch around it makes hipcc shuttle them AGPR<->VGPR every stage).
    const int a_off = (active ? a_slot : 0) * SLOT_BYTES + lane * 16;
    const int b_off = (active ? b_slot : 0) * SLOT_BYTES + lane * 16;

    // fragments of K-tile j (0..KT_STAGE-1) of stage s
    auto load_raw = [&](int s, int j) {
        const uint8_t* base = lds + (s & (XC_RING - 1)) * STAGE_BYTES + j * KT_BYTES;
        RawFrags r;
        r.a[0] = *reinterpret_cast<const v4i*>(base + a_off);
        r.a[1] = *reinterpret_cast<const v4i*>(base + a_off + FRAG_BYTES);
        r.b[0] = *reinterpret_cast<const v4i*>(base + b_off);
        r.b[1] = *reinterpret_cast<const v4i*>(base + b_off + FRAG_BYTES);
        return r;
    };

    // Diagonal wave tiles (row block == column block) never store the upper-right 32x32 MFMA tile
    // (rows 0-31 x columns 32-63: Rh < Ch): skip its 4 MFMAs (wave-uniform branch; idle waves too).
    const bool skip01 = __builtin_amdgcn_readfirstlane((int)(!active || blk_a == blk_b)) != 0;

    // ---- software pipeline over K-tiles g = s*KT_STAGE + j ------------------------------------
    //   MFMA(g)  ||  unpack(g+1)  ||  LDS read(g+2)  ||  LDS-DMA of stage s+3
    // 4-deep LDS ring.  At the end of stage s every wave waits until its own pieces of stage s+2
    // have landed (counted vmcnt: stage s+3 stays in flight), then one barrier: stages s+1 and s+2
    // are now visible to all waves (the LDS reads two K-tiles ahead cross into the next stage), and
    // everybody is done reading stage s, whose buffer the DMA of stage s+4 overwrites.
#pragma unroll
    for (int st = 0; st < 3; st++)
#pragma unroll
        for (int n = 0; n < NLOAD; n++) issue_piece(st, n);
    wait_vmcnt<NLOAD>();
    __builtin_amdgcn_s_barrier();

    Frags cur = unpack_frags(load_raw(0, 0));
    RawFrags raw = load_raw(0, 1);
    unsigned long long t_start = 0, r_start = 0;
    if (p.stamps) {   // diagnostic build path: shader clock vs 100 MHz reference (MI355X_MICROARCH, DVFS item 6)
        t_start = __builtin_amdgcn_s_memtime();
        r_start = __builtin_amdgcn_s_memrealtime();
    }
    for (int s = 0; s < nstage; s++) {
#pragma unroll
        for (int j = 0; j < KT_STAGE; j++) {
            if (!(ABL & 1)) {
                issue_piece(s + 3, 2 * j);
                issue_piece(s + 3, 2 * j + 1);
            }
            xcorr_mfma_tile(cur, skip01, accR, accP, accQ);
            if (ABL & 2) {
#pragma unroll
                for (int m = 0; m < 2; m++) { cur.ar[m] = raw.a[m]; cur.ai[m] = raw.a[m]; cur.br[m] = raw.b[m]; cur.bi[m] = raw.b[m]; }
            } else {
                cur = unpack_frags(raw);
            }
            // K-tile g+2: same stage for j < KT_STAGE-2, else the next stage (already visible; past the
            // end of K the read lands in a valid ring buffer and is never used)
            if (!(ABL & 4)) raw = (j + 2 < KT_STAGE) ? load_raw(s, j + 2) : load_raw(s + 1, j + 2 - KT_STAGE);
            else { asm volatile("" : "+v"(raw.a[0]), "+v"(raw.a[1]), "+v"(raw.b[0]), "+v"(raw.b[1])); }
            // pin the interleave: 1 MFMA : 3 VALU (the 48 mask/shift ops of the next K-tile hide under
            // the 16 MFMAs of this one), the two DMA pieces early, the four LDS reads in the second half
#pragma unroll
            for (int i = 0; i < 16; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                   // VALU
                if (i == 1 || i == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
                if (i >= 8 && i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
        }
        if (!(ABL & 8)) {
            if (!(ABL & 1)) wait_vmcnt<NLOAD>();
            __builtin_amdgcn_s_barrier();
        }
    }
    wait_vmcnt<0>();   // no LDS-DMA may still be in flight when the wave ends
    if (p.stamps) {
        // wait for the last MFMA to retire before stamping (read one accumulator element)
        asm volatile("" :: "v"(accR[1][1][15]), "v"(accQ[1][1][15]), "v"(accP[1][1][15]));
        const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* o = p.stamps + ((size_t)(c * p.nwg + wg) * 4 + wave) * 8;
            o[0] = t_end - t_start; o[1] = r_end - r_start; o[2] = r_start; o[3] = r_end; o[4] = r_entry;
        }
    }
    if (!active) return;
    xcorr_store_tile(p, c, blk_a, blk_b, skip01, lane, accR, accP, accQ);
    if (p.stamps && lane == 0) p.stamps[((size_t)(c * p.nwg + wg) * 4 + wave) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
}

// =======================================================================================
// Default path: corner turn fused into the LDS staging, persistent work-groups, fragment-level tiling.
//
// The kernel reads the gulps where they lie (time-major rows of ninput bytes per channel).  Per stage
// (96 samples) and pair of 64-input blocks the LDS-DMA brings 96 rows x 128 B; the operand fragments come
// out of LDS through ds_read_b64_tr_b8, the byte-transposing read (profiles/microbench/tr8_probe.hip):
// 16-lane group (h, rg) of a wave reads the 8x16 byte block rows 16h+8hh.., inputs 32 sub + 16 rg.., and
// lane r receives 8 consecutive samples of input r -- two reads make the 16 bytes of an MFMA operand
// register quad.  (Which samples sit in which byte does not matter: A and B use the same map.)
//
// LDS image: slots are staged in pairs (0,1) and (2,3); the image of a pair and stage is
// [96 rows][128 B] (slot parity = 64-byte half) with chunk position = chunk ^ 2*((row>>1)&3): the swizzle
// is applied to the per-lane DMA source address (the LDS side of the DMA is linear) and to the read
// address, so that the 32 lanes of a read phase fall on 32 distinct even banks.  Wave w brings rows
// 48*(w&1).. of pair w>>1; a piece is 8 rows x 128 B, lane = 8*row + chunk position.  The tiling makes
// most pairs adjacent blocks (2k, 2k+1), whose halves then form whole 128-byte lines (half as many L2
// requests as 64-byte row segments).
//
// Tiling (xcorr_tiling.h, FragGroup): a wave contracts four 32x32 cells from four operand fragments, each of which may
// be either 32-input half of any staged block -- as a 2x2 outer product (an off-diagonal 64x64 tile), or in the Z
// pattern (the three stored cells of a diagonal 64x64 tile plus one free cell of an off-diagonal tile): the K loop
// exists in both wirings, selected per item by a wave-uniform branch outside the loop.  704 inputs: 16 tile groups.
//
// Persistence: the grid is one work-group per CU; each walks a list of (channel, tile group) items
// (whole channels per XCD, as above).  The LDS-DMA stream runs ahead of the MFMA stream by three stages
// ACROSS items: while an item's last K-tiles are contracted and its tiles are stored, the first stages of
// the next item are already landing, so the fill of the pipeline (and the dispatch of a new work-group)
// is paid once per CU instead of once per item.
// =======================================================================================

// the 16 (12 when cell 1 is dead) int8 MFMAs of one K-tile: for every cell R += xr*yr + xi*yi, P += xi*yr, Q += xr*yi.
// Operands: u.a = fragments 0, 1; u.b = fragments 2, 3 of the wave (FragGroup).  Unpacked fragments are the same thing
// whether they enter as A or B operand, so the Z wiring multiplies a diagonal fragment with itself.
// skipdiag (diagnostic builds, ABL & 128; results are wrong): the two diagonal cells of a Z wave sit this K-tile out.  Done on every
// fourth K-tile it takes away a quarter of their MFMA work -- what 16x16x64 MFMAs on the three needed sub-cells would save --
// without building that data path: an upper bound of the lever (round-3 review, item 3).
template <bool Z>
__device__ __forceinline__ void xcorr_mfma_cells(const Frags& u, bool skip1, v16i (&accR)[2][2], v16i (&accP)[2][2],
                                                 v16i (&accQ)[2][2], bool skipdiag = false) {
    auto cell = [&](int m, int n, const v4i& xr, const v4i& xi, const v4i& yr, const v4i& yi) {
        accR[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xr, yr, accR[m][n], 0, 0, 0);
        accP[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xi, yr, accP[m][n], 0, 0, 0);
        accQ[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xr, yi, accQ[m][n], 0, 0, 0);
        accR[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xi, yi, accR[m][n], 0, 0, 0);
    };
    if (Z) {
        if (!skipdiag) cell(0, 0, u.ar[0], u.ai[0], u.ar[0], u.ai[0]);  // (d0, d0)
        if (!skip1) cell(0, 1, u.br[0], u.bi[0], u.br[1], u.bi[1]);     // the free cell (r, c)
        cell(1, 0, u.ar[1], u.ai[1], u.ar[0], u.ai[0]);                 // (d1, d0)
        if (!skipdiag) cell(1, 1, u.ar[1], u.ai[1], u.ar[1], u.ai[1]);  // (d1, d1)
    } else {
        cell(0, 0, u.ar[0], u.ai[0], u.br[0], u.bi[0]);
        if (!skip1) cell(0, 1, u.ar[0], u.ai[0], u.br[1], u.bi[1]);
        cell(1, 0, u.ar[1], u.ai[1], u.br[0], u.bi[0]);
        cell(1, 1, u.ar[1], u.ai[1], u.br[1], u.bi[1]);
    }
}

// Epilogue of the fragment-tiled kernel: accumulator (m, n) is cell p = 2m + n with rows = inputs 32*row[p].., columns =
// inputs 32*col[p].. (wave-uniform).  Same register-tile order, lane regrouping and masks as xcorr_store_tile.
// fast: all four cells live, strictly below the diagonal, no padded inputs, nothing to add to -- exactly 32 stores.
// MR: accumulator rows m in use (2: all four cells; 1: cells 0 and 1 only -- the two-cell wave tiles of experiments/xcorr_fused16.h)
template <bool LACC, int MR = 2>
__device__ __forceinline__ void xcorr_store_cells(const XcorrParams& p, int c, const int (&row)[4], const int (&col)[4],
                                                  int live, bool fast, bool accumulate, int lane, const v16i (&accR)[2][2],
                                                  const v16i (&accP)[2][2], const v16i (&accQ)[2][2]) {
    const int qs = (int)(((int64_t)(p.nstand / 2 + 1) * p.nstand) / 4);
    int32_t* out_r = p.out + (int64_t)c * p.per_chan;
    int32_t* out_i = out_r + p.matlen;
    int32_t* acc_r = LACC ? p.acc2 + (int64_t)c * p.per_chan : nullptr;
    int32_t* acc_i = LACC ? acc_r + p.matlen : nullptr;
    const bool acc_add = LACC && p.acc2_mode == 2;
    auto long_acc = [&](int w, int4 cr, int4 ci) {           // the stored cell -> the long accumulator
        int4* ar = reinterpret_cast<int4*>(acc_r + w);
        int4* ai = reinterpret_cast<int4*>(acc_i + w);
        if (acc_add) {
            const int4 o_r = *ar, o_i = *ai;
            cr.x += o_r.x; cr.y += o_r.y; cr.z += o_r.z; cr.w += o_r.w;
            ci.x += o_i.x; ci.y += o_i.y; ci.z += o_i.z; ci.w += o_i.w;
        }
        *ar = cr;
        *ai = ci;
    };
    const int odd = lane & 1;
    auto cell = [&](int v0, int v1, int v2, int v3) {
        const int g0 = dpp_xor1(odd ? v0 : v2), g1 = dpp_xor1(odd ? v1 : v3);
        return odd ? make_int4(g0, v2, g1, v3) : make_int4(v0, g0, v1, g1);
    };
    const int pull = ((lane & 32) | (4 * (lane & 7) + ((lane >> 3) & 3))) * 4;   // byte address of the source lane
    auto regroup = [&](int4 v) {
        return make_int4(__builtin_amdgcn_ds_bpermute(pull, v.x), __builtin_amdgcn_ds_bpermute(pull, v.y),
                         __builtin_amdgcn_ds_bpermute(pull, v.z), __builtin_amdgcn_ds_bpermute(pull, v.w));
    };
    const int quad = (lane >> 3) & 3;          // 2*(C&1) + (R&1) of the cell this lane stores after the regrouping
    const int cpar = quad >> 1, rpar = quad & 1;
    if (fast) {
        // straight-line code: no per-cell branches, so the lane regrouping of one cell overlaps the arithmetic of the next
#pragma unroll
        for (int m = 0; m < MR; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const int ibase = row[2 * m + n] * 32, jbase = col[2 * m + n] * 32;
                const int Ch = (jbase >> 2) + (lane & 7);
                const int wcol = (quad * qs + Ch) * 4;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int Rh = (ibase >> 2) + 2 * u + (lane >> 5);
                    int vr[4], vi[4];
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        vr[v] = accR[m][n][4 * u + v] >> 8;
                        vi[v] = (accP[m][n][4 * u + v] - accQ[m][n][4 * u + v]) >> 8;
                    }
                    const int4 cr = regroup(cell(vr[0], vr[1], vr[2], vr[3]));
                    const int4 ci = regroup(cell(vi[0], vi[1], vi[2], vi[3]));
                    const int w = wcol + ((Rh * (Rh + 1)) >> 1) * 4;
                    *reinterpret_cast<int4*>(out_r + w) = cr;
                    *reinterpret_cast<int4*>(out_i + w) = ci;
                    if (LACC) long_acc(w, cr, ci);
                }
            }
        return;
    }
#pragma unroll
    for (int m = 0; m < MR; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            if (!((live >> (2 * m + n)) & 1)) continue;          // wave-uniform
            const int ibase = row[2 * m + n] * 32, jbase = col[2 * m + n] * 32;
            const bool interior = row[2 * m + n] > col[2 * m + n] && ibase + 32 <= 2 * p.nstand;
            const int Ch = (jbase >> 2) + (lane & 7);
            const int C = 2 * Ch + cpar;
            const int wcol = (quad * qs + Ch) * 4;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int Rh = (ibase >> 2) + 2 * u + (lane >> 5);
                const int R = 2 * Rh + rpar;
                int vr[4], vi[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    vr[v] = accR[m][n][4 * u + v] >> 8;
                    vi[v] = (accP[m][n][4 * u + v] - accQ[m][n][4 * u + v]) >> 8;
                }
                int4 cr = regroup(cell(vr[0], vr[1], vr[2], vr[3]));
                int4 ci = regroup(cell(vi[0], vi[1], vi[2], vi[3]));
                const int w = wcol + ((Rh * (Rh + 1)) >> 1) * 4;
                int4* pr = reinterpret_cast<int4*>(out_r + w);
                int4* pi = reinterpret_cast<int4*>(out_i + w);
                if (interior || (Rh >= Ch && R < p.nstand && C < p.nstand)) {
                    if (accumulate) {
                        const int4 o_r = *pr, o_i = *pi;
                        cr.x += o_r.x; cr.y += o_r.y; cr.z += o_r.z; cr.w += o_r.w;
                        ci.x += o_i.x; ci.y += o_i.y; ci.z += o_i.z; ci.w += o_i.w;
                    }
                    *pr = cr;
                    *pi = ci;
                    if (LACC) long_acc(w, cr, ci);
                }
            }
        }
}

// ABL: timing-only ablation bits as for xcorr_mfma_kernel (diagnostic builds; results are wrong unless 0);
// 16: no epilogue, 32: every channel reads a 1 MB window of gulp 0 that stays in L2, 64: half-item skew between the two
// channels of a round (an experiment: results stay right).
// DESC: gulps by descriptor (GulpDesc above) instead of by pointer: strides are per gulp and read with scalar loads; the default
// instantiation is untouched.
// TAB (round 5; a third instantiation, the DESC one is what it was): every gulp of the launch is read through its offset TABLE
// (slab.h) -- a slab of packets with lost, shifted, reordered or duplicated packets is read where it lies, like a regular one:
//   * per stage a wave brings the 1 KiB of table rows of its two 64-input blocks into a small LDS ring with ONE more LDS-DMA
//     instruction (the stage walker runs SIX stages ahead of the MFMAs for it; the pieces still run three stages ahead),
//   * one iteration before it issues the pieces of a stage, every lane reads its six offsets from there (ds_read_b128 + b64, under
//     the MFMAs of the stage's last K-tile: the rows landed two stages ago, nothing waits),
//   * when the pieces of a stage have landed, lanes whose sample nobody carries overwrite their 16 bytes with zeros (a flag word in
//     the same row, read when the stage begins; a wave-uniform branch per stage when nothing is missing).
// Every stage costs NVM = 7 operations on the vmcnt counter instead of 6.
template <int ABL, bool LACC = false, bool DESC = false, bool TAB = false>
__global__ __launch_bounds__(256, 1) void xcorr_fused_kernel(XcorrParams p) {
    constexpr int KT_STAGE = XC_KT;
    constexpr int SLOT_BYTES = KT_STAGE * KT_BYTES;
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES;
    constexpr int NLOAD = 2 * KT_STAGE;  // 1 KiB LDS-DMA pieces per wave per stage
#ifndef XT_EXPERIMENT
#define XT_EXPERIMENT 0       // timing-only A/B builds of the TAB path (results wrong): 1 no hole flags / fix, 2 no table fetch in the loop, 3 both
                              // (profiles/r05/slab_tables_ablation.txt: the fetch costs ~3 %, the flags ~1-2 %, everything else nothing)
#endif
    constexpr int NVM = NLOAD + ((TAB && !(XT_EXPERIMENT & 2)) ? 1 : 0);    // vector-memory operations per wave and stage (TAB: + the table rows)
    constexpr int DEPTH = XF_DEPTH;          // the LDS-DMA of stage S+DEPTH is issued while stage S is contracted
    constexpr int RING = DEPTH + 1;          // LDS ring (stages)
    static_assert(!TAB || (DESC && DEPTH == 3), "the table ring (flags of S+2, offsets of S+3 and S+4, S+5 landed, S+6 in flight) assumes DEPTH 3");
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t idx_lds[TAB ? 8 * 4096 : 16];   // [stage & 7][wave][block of the pair][row r8][64 B]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t row_stride = (uint32_t)p.nchan * (uint32_t)p.ninput;
    // tile-group table through the constant address space: wave-uniform reads become scalar loads (a vector
    // load would make the compiler wait for vmcnt(0), i.e. for the whole LDS-DMA stream, at every item switch)
    // (32-byte groups read as aligned dwords: slot_blk | wave[0..3])
    typedef const __attribute__((address_space(4))) uint32_t* DescPtr;
    const DescPtr groups = (DescPtr)(uintptr_t)p.fgroups;
    static_assert(sizeof(FragGroup) == 32, "group layout");

    // entry k of this work-group's list; false past the end
    const DescPtr work = (DescPtr)(uintptr_t)p.work + (size_t)blockIdx.x * p.maxi;
    struct Item { int c, wg; };
    auto item = [&](int k, Item& it) {
        if (k >= p.maxi) return false;
        const uint32_t w = work[k];
        if (!(w & WORK_VALID)) return false;
        it.c = (int)(w & 0xFFFF); it.wg = (int)((w >> 16) & 0x7FFF);
        return true;
    };

    // ---- issue side: the stage stream (item, gulp, stage in gulp) DEPTH stages ahead of the MFMAs ----
    int is_k = 0, is_c = 0, is_g = 0, is_sl = 0, is_issued = 0;
    uint32_t is_voff[NLOAD] = {};
    const uint8_t* is_stage = nullptr;
    // DESC: this lane's 64-input block and byte position inside it (per item), the current gulp's layout (per gulp)
    uint32_t is_blk = 0, is_sub = 0, d_t = row_stride, d_c = (uint32_t)p.ninput;
    const uint8_t* d_base = nullptr;
    const DescPtr gdesc = (DescPtr)(uintptr_t)p.gdesc;
    // TAB: what next_stage() made of the walker's stage -- base of its pieces, source of its table rows --, the bases of the three stages in
    // between (the pieces run three stages behind the walker), and per lane: its 16 bytes of the wave's table rows (per item), its row
    // in the LDS copy and its 16-byte piece of a 64-byte payload row (constants), the flags of the stage that lands
    const int lchunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);            // source chunk 0..7 of the 128-byte pair row
    const uint32_t t_sub = (uint32_t)(lchunk & 3) * 16u, ix_row = (uint32_t)(wave * 1024 + (lchunk >> 2) * 512 + (lane >> 3) * 64);
    uint32_t ix_voff = 0, hole_flags = 0;
    const uint8_t *d_tab = nullptr, *nx_sb = nullptr, *nx_ix = nullptr, *sb_q0 = nullptr, *sb_q1 = nullptr, *sb_q2 = nullptr, *sb_now = nullptr;
    auto load_desc = [&](int g) {          // gulp g's layout (scalar loads) -> this lane's six piece offsets
        const uint32_t lo = gdesc[g * 8], hi = gdesc[g * 8 + 1];
        d_base = (const uint8_t*)(((uint64_t)hi << 32) | lo);
        if (TAB) {
            d_c = gdesc[g * 8 + 3];
            d_tab = (const uint8_t*)(((uint64_t)gdesc[g * 8 + 7] << 32) | gdesc[g * 8 + 6]);
            return;
        }
        d_t = gdesc[g * 8 + 2];
        d_c = gdesc[g * 8 + 3];
        const uint32_t d_b = gdesc[g * 8 + 4];
        const uint32_t lane_off = (uint32_t)(lane >> 3) * d_t + is_blk * d_b + is_sub;
#pragma unroll
        for (int n = 0; n < NLOAD; n++) is_voff[n] = lane_off + (uint32_t)n * 8u * d_t - (uint32_t)((n & 3) * 1024);
    };
    auto is_setup = [&](const Item& it) {
        is_c = it.c;
        // (columns past ninput in the last block: any valid bytes of the row; their products are never stored)
        const uint32_t slots = groups[it.wg * 8];                         // slot_blk[0..3]
        const int chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);          // source chunk 0..7 of the 128-byte pair row
        const int blk0 = (slots >> (8 * (wave & 2))) & 0xFF, blk1 = (slots >> (8 * (wave & 2) + 8)) & 0xFF;
        if (TAB) {
            // (whole 64-input blocks only: the host hands this kernel nothing else)
            ix_voff = (uint32_t)((lane & 32) ? blk1 : blk0) * (uint32_t)p.spg * 1024u + (uint32_t)(lane & 31) * 16u;
        } else if (DESC) {
            is_blk = (uint32_t)((chunk >> 2) ? blk1 : blk0);
            is_sub = (uint32_t)(chunk & 3) * 16u;
            if (is_blk * 64u + is_sub + 16u > (uint32_t)p.ninput) { is_blk = 0; is_sub = 0; }
            load_desc(0);
        } else {
        const uint32_t col = (uint32_t)((chunk >> 2) ? blk1 : blk0) * 64u + (uint32_t)(chunk & 3) * 16u;
        const uint32_t lane_off = (uint32_t)(lane >> 3) * row_stride + (col + 16u <= (uint32_t)p.ninput ? col : 0u);
#pragma unroll
        // (never negative: the host takes this kernel only when a row has at least 128 bytes, xengXgpuInitialize)
        for (int n = 0; n < NLOAD; n++) is_voff[n] = lane_off + (uint32_t)n * 8u * row_stride - (uint32_t)((n & 3) * 1024);
        }
        is_g = 0;
        is_sl = 0;
        is_issued = 0;
    };
    auto next_stage = [&]() {
        if (is_issued == p.nstage) {         // this item is fully issued: go on with the next one, if any
            Item nx;
            if (!item(is_k + 1, nx)) return;   // past the end: keep re-reading the last stage (never consumed)
            is_k++;
            is_setup(nx);
        }
        if (TAB) {
            if (is_sl == 0) load_desc(is_g);
            nx_sb = d_base + (size_t)is_c * d_c - SLAB_OFF_BIAS;      // (never dereferenced as it stands: every table offset carries the bias)
            nx_ix = d_tab + (size_t)(is_sl * 2 + (wave & 1)) * 512;
        } else if (DESC) {
            // (a new gulp may be laid out differently: its descriptor is loaded here, when its first stage is set up -- not when
            // the previous gulp's last stage was, whose pieces are still to be issued with the old offsets)
            if (is_sl == 0 && is_g > 0) load_desc(is_g);
            is_stage = d_base + (size_t)(is_sl * (KT_STAGE * 32)) * d_t + (size_t)is_c * d_c;
        }
        else if (ABL & 32)   // timing only: every channel reads channel (c & 7)'s first stage of gulp 0 (a 1 MB window that stays in L2)
            is_stage = p.gulps[0] + (size_t)(is_c & 7) * (size_t)p.ninput;
        else
        is_stage = p.gulps[is_g] + ((size_t)(is_sl * (KT_STAGE * 32)) * p.nchan + is_c) * (size_t)p.ninput;
        is_issued++;
        if (++is_sl == p.spg) { is_sl = 0; is_g++; }
    };
    // one 1 KiB piece (8 rows x 128 B) of the stage last returned by next_stage() into ring buffer S % XC_RING.
    // Issued from asm: hipcc cannot prove that the transposing reads do not alias a pending builtin LDS-DMA
    // and would put `s_waitcnt vmcnt(0)` in front of every one of them; the counted vmcnt + barrier at the
    // end of each stage is the real ordering.
    // M0 (the LDS base of the transfer) is written without saving it: nothing else in this kernel uses M0
    // (no builtin LDS-DMA, no s_movrel/sendmsg; hipcc rejects "m0" as a clobber, so the ISA was checked).
    // The six pieces of a stage are issued in order 0..5.  The instruction's immediate offset is added to the global
    // AND to the LDS address, so with per-lane offsets that subtract it again on the global side
    // (raw_voff[n] = lane offset + n * 8 rows - imm) the pieces share one scalar base per stage and M0 is written
    // twice per stage (pieces 0-3: imm 0..3072, pieces 4-5: M0 + 4096, imm 0, 1024) instead of once per piece.
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds);
    const uint32_t idx_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)idx_lds);
    // TAB: the table rows of the stage last returned by next_stage(): 2 x 512 B (the wave's two blocks), one instruction
    auto issue_table = [&](int slot) {
        const uint32_t la = idx_base + slot * 4096 + wave * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(ix_voff), "s"(nx_ix), "s"(la) : "memory");
    };
    // ... this lane's six offsets out of a landed table block
    auto load_offsets = [&](int slot) {
        const uint8_t* row = idx_lds + slot * 4096 + ix_row;
        const uint4 e = *reinterpret_cast<const uint4*>(row);
        const uint2 f = *reinterpret_cast<const uint2*>(row + 16);
        is_voff[0] = e.x + t_sub; is_voff[1] = e.y + t_sub; is_voff[2] = e.z + t_sub; is_voff[3] = e.w + t_sub;
        is_voff[4] = f.x + t_sub; is_voff[5] = f.y + t_sub;
    };
    // ... and, once the pieces of that stage have landed in ring slot `ring_slot`: zeros where no packet carries the sample
    auto load_flags = [&](int slot) { hole_flags = *reinterpret_cast<const uint32_t*>(idx_lds + slot * 4096 + ix_row + 24); };
    auto fix_holes = [&](int ring_slot) {
        if (__builtin_amdgcn_ballot_w64(hole_flags != 0u) != 0ull) {
            // (a loop over the set bits of the lanes that have any -- usually one lost packet: four lanes, one piece, one pass --
            // instead of six masked stores: a third of the instructions on a link that loses 1 % of its packets)
            uint8_t* own = lds + ring_slot * STAGE_BYTES + wave * SLOT_BYTES + lane * 16;
            uint32_t fl = hole_flags;
            while (fl != 0u) {
                const int n = __builtin_ctz(fl);
                *reinterpret_cast<uint4*>(own + n * 1024) = make_uint4(0u, 0u, 0u, 0u);
                fl &= fl - 1u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the barrier that publishes the stage does not wait for LDS writes)
        }
    };
    auto issue_piece = [&](int ring_slot, int n) {
        const uint8_t* sb = TAB ? sb_now : is_stage + (size_t)(48 * (wave & 1)) * (DESC ? d_t : row_stride);   // scalar base of this wave's rows
        const uint32_t la = lds_base + ring_slot * STAGE_BYTES + wave * SLOT_BYTES + (n < 4 ? 0 : 4096);
        const uint32_t vo = is_voff[n];
        if (n == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo), "s"(sb), "s"(la) : "memory");
        if (n == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(vo), "s"(sb) : "memory");
        if (n == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(vo), "s"(sb) : "memory");
        if (n == 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(vo), "s"(sb) : "memory");
        if (n == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo), "s"(sb), "s"(la) : "memory");
        if (n == 5) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(vo), "s"(sb) : "memory");
    };

    // read side: lane 16*grp + 2q + pp addresses row 16*(grp>>1) + q (+8 for the second read), chunk position
    // (4*slot parity + 2*sub + (grp&1)) ^ 2*((q>>1)&3), bytes 8pp..8pp+7: slot parity flips address bit 6,
    // sub = 1 (the upper 32 inputs of a block) flips bit 5
    const int tr_off = ((lane >> 5) * 16 + ((lane & 15) >> 1)) * 128 +
                       (((lane >> 4) & 1) ^ (((lane >> 2) & 3) << 1)) * 16 + (lane & 1) * 8;

    Item it;
    if (!item(0, it)) return;                // (the host never launches work-groups without work)
    if (ABL & 64) {   // timing only: the second channel of every round (work-groups 16-31 of an XCD) starts half an item late,
                      // so that the two channels' epilogue bursts do not coincide
        if ((blockIdx.x >> 3) & 16) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < 1500ull) __builtin_amdgcn_s_sleep(32);     // 15 us at 100 MHz
        }
    }
    is_setup(it);
    if (TAB) {
        // table rows of stages 0-5, then the pieces of stages 0-2 from them
        const uint8_t* sbs[6];
#pragma unroll
        for (int st = 0; st < 6; st++) {
            next_stage();
            sbs[st] = nx_sb;
            issue_table(st);
        }
        wait_vmcnt<0>();
#pragma unroll
        for (int st = 0; st < DEPTH; st++) {
            load_offsets(st);
            sb_now = sbs[st];
#pragma unroll
            for (int n = 0; n < NLOAD; n++) issue_piece(st, n);
        }
        wait_vmcnt<NLOAD>();                 // stages 0 and 1 have landed
        load_flags(0); fix_holes(0);
        load_flags(1); fix_holes(1);
        load_offsets(3);                     // for the pieces of stage 3, issued while stage 0 is contracted
        sb_q0 = sbs[3]; sb_q1 = sbs[4]; sb_q2 = sbs[5];
    } else {
#pragma unroll
    for (int st = 0; st < DEPTH; st++) {
        next_stage();
#pragma unroll
        for (int n = 0; n < NLOAD; n++) issue_piece(st, n);
    }
    wait_vmcnt<(DEPTH - 2) * NLOAD>();       // stages 0 and 1 have landed
    }
    __builtin_amdgcn_s_barrier();

    // ring slots of the stage being contracted (rs), the next one (rs1) and the one being filled (rf):
    // the stage counter is continuous across items
    int rs = 0, rs1 = 1, rf = DEPTH;
    auto bump = [&](int& r) { r = (r + 1 == RING) ? 0 : r + 1; };
    // TAB: q = (number of the stage being contracted) & 7, the table ring's phase
    int q = 0;
    // what a stage begins with: the walker moves on (TAB: six stages ahead; the base of the pieces issued now was made three stages
    // ago), the flags of the stage that lands meanwhile are fetched
    auto stage_begin = [&]() {
        if (TAB) { sb_now = sb_q0; sb_q0 = sb_q1; sb_q1 = sb_q2; }
        next_stage();
        if (TAB) {
            sb_q2 = nx_sb;
            if (!(ABL & 1) && !(XT_EXPERIMENT & 2)) issue_table((q + 6) & 7);
            if (!(XT_EXPERIMENT & 1)) load_flags((q + 2) & 7);
        }
    };
    // TAB, once the last piece of the stage has been issued: the offsets of the pieces that the NEXT stage issues (their rows landed
    // two stages ago; the reads run under the MFMAs of this K-tile)
    auto stage_offsets = [&]() {
        if (TAB) load_offsets((q + 4) & 7);
    };
    // ... and ends with: this wave's pieces of stage S+2 (and everything older) have landed; TAB: zeros for its missing samples; one barrier
    auto stage_end = [&](bool relaxed) {
        if (!(ABL & 8)) {
            // (first stage behind a straight-line epilogue: the 32 tile stores of that epilogue sit between the DMA of
            // stage S+2, which this wait is for, and the pieces just issued; vmcnt retires in issue order, so
            // allowing them to stay in flight does not let stage S+2 slip -- and the wave does not stall for the
            // write acknowledgements of the previous item)
            if (!(ABL & 1)) { if (relaxed && !XF_NO_RELAX) wait_vmcnt<(DEPTH - 2) * NVM + 32>(); else wait_vmcnt<(DEPTH - 2) * NVM>(); }
            if (TAB) {
                if (!(XT_EXPERIMENT & 1)) fix_holes(rs1 + 1 == RING ? 0 : rs1 + 1);
                q = (q + 1) & 7;
            }
            __builtin_amdgcn_s_barrier();
        }
        bump(rs); bump(rs1); bump(rf);
    };
    bool stores_in_flight = false;            // the previous item of this wave ended with the 32-store epilogue
    for (int k = 0; item(k, it); k++) {
        const int c = it.c, wg = it.wg;
        const unsigned long long r_entry = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const uint32_t slots = groups[wg * 8], ww = groups[wg * 8 + 1 + wave];
        if (!(ww & FRAG_BUSY)) {
            // A wave without a cell only keeps the stage stream and the barriers going (no MFMAs on dummy data: the
            // kernel is power-limited).  Nothing is live across this branch.
            for (int s = 0; s < p.nstage; s++) {
                stage_begin();
                if (!(ABL & 1)) {
#pragma unroll
                    for (int n = 0; n < NLOAD; n++) issue_piece(rf, n);
                }
                stage_offsets();
                stage_end(false);
            }
            stores_in_flight = false;
            continue;
        }
        // operand fragments: LDS offsets and 32-input block numbers (wave-uniform positions)
        const int live = (int)((ww >> 16) & 15);
        const bool zpat = (ww & FRAG_Z) != 0;
        const bool skip1 = !(live & 2);         // cell 1 is dead: its 4 MFMAs per K-tile are skipped
        int off[4], b32[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int pos = (int)((ww >> (3 * q)) & 7);
            off[q] = (pos >> 2) * (2 * SLOT_BYTES) + (tr_off ^ (((pos >> 1) & 1) * 64) ^ ((pos & 1) * 32));
            b32[q] = (int)((slots >> (8 * (pos >> 1))) & 0xFF) * 2 + (pos & 1);
        }
        int row[4], col[4];
        if (zpat) {
            row[0] = b32[0]; col[0] = b32[0]; row[1] = b32[2]; col[1] = b32[3];
            row[2] = b32[1]; col[2] = b32[0]; row[3] = b32[1]; col[3] = b32[1];
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) { row[q] = b32[q >> 1]; col[q] = b32[2 + (q & 1)]; }
        }

        auto load_raw = [&](int ring_slot, int j) {     // fragments of K-tile j (0..KT_STAGE-1) of a landed stage
            const uint8_t* base = lds + ring_slot * STAGE_BYTES + j * (2 * KT_BYTES);
            auto tr = [&](int o) {
                return __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(base + o));
            };
            RawFrags r;
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const v2i a0 = tr(off[f]), a1 = tr(off[f] + 1024);
                const v2i b0 = tr(off[2 + f]), b1 = tr(off[2 + f] + 1024);
                r.a[f] = (v4i){a0.x, a0.y, a1.x, a1.y};
                r.b[f] = (v4i){b0.x, b0.y, b1.x, b1.y};
            }
            return r;
        };

        v16i accR[2][2], accP[2][2], accQ[2][2];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 2; n++) {
                accR[m][n] = (v16i)(0);
                accP[m][n] = (v16i)(0);
                accQ[m][n] = (v16i)(0);
            }

        unsigned long long t_start = 0, r_start = 0;
        if (p.stamps) {   // diagnostic: shader clock vs 100 MHz reference (MI355X_MICROARCH, DVFS item 6)
            t_start = __builtin_amdgcn_s_memtime();
            r_start = __builtin_amdgcn_s_memrealtime();
        }
        // ---- software pipeline over K-tiles: MFMA(g) || unpack(g+1) || LDS read(g+2) || LDS-DMA of stage S+DEPTH.
        // At the end of stage S every wave waits until its own pieces of stage S+2 have landed (counted
        // vmcnt: the younger stages stay in flight), then one barrier: stages S+1 and S+2 are visible to all
        // waves and everybody is done reading stage S, whose buffer the DMA of stage S+DEPTH+1 overwrites.
        auto kloop = [&](auto zc) {
            constexpr bool Z = decltype(zc)::value;
            Frags cur = unpack_frags(load_raw(rs, 0));
            RawFrags raw = load_raw(rs, 1);
            for (int s = 0; s < p.nstage; s++) {
                stage_begin();
#pragma unroll
                for (int j = 0; j < KT_STAGE; j++) {
                    if (!(ABL & 1)) {
                        issue_piece(rf, 2 * j);
                        issue_piece(rf, 2 * j + 1);
                    }
                    if (j == KT_STAGE - 1) stage_offsets();
                    xcorr_mfma_cells<Z>(cur, skip1, accR, accP, accQ, (ABL & 128) && Z && ((s * KT_STAGE + j) & 3) == 3);
                    if (ABL & 256) {
                        cur = unpack_frags_offset_binary(raw);
                    } else if (ABL & 2) {
#pragma unroll
                        for (int m = 0; m < 2; m++) { cur.ar[m] = raw.a[m]; cur.ai[m] = raw.a[m]; cur.br[m] = raw.b[m]; cur.bi[m] = raw.b[m]; }
                    } else {
                        cur = unpack_frags(raw);
                    }
                    // K-tile g+2: same stage for j < KT_STAGE-2, else the next stage (already visible; at the end
                    // of an item the read lands in a valid ring buffer and is not used)
                    if (!(ABL & 4)) raw = (j + 2 < KT_STAGE) ? load_raw(rs, j + 2) : load_raw(rs1, j + 2 - KT_STAGE);
                    else { asm volatile("" : "+v"(raw.a[0]), "+v"(raw.a[1]), "+v"(raw.b[0]), "+v"(raw.b[1])); }
                    // pin the interleave: 1 MFMA : 3 VALU (the 48 mask/shift ops of the next K-tile hide under the
                    // 16 MFMAs of this one), the eight transposing LDS reads in the second half
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                   // VALU
                        if (i >= 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
                    }
                }
                stage_end(s == 0 && stores_in_flight);
            }
        };
        if (zpat) kloop(std::true_type{});
        else kloop(std::false_type{});
        if (p.stamps) {
            asm volatile("" :: "v"(accR[1][1][15]), "v"(accQ[1][1][15]), "v"(accP[1][1][15]));
            const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                unsigned long long* o = p.stamps + ((size_t)(c * p.nwg + wg) * 4 + wave) * 8;
                o[0] = t_end - t_start; o[1] = r_end - r_start; o[2] = r_start; o[3] = r_end; o[4] = r_entry;
            }
        }
        // fast epilogue = exactly 32 store instructions and nothing else
        bool fast = live == 15 && p.accumulate == 0;
#pragma unroll
        for (int q = 0; q < 4; q++) fast = fast && row[q] > col[q] && row[q] * 32 + 32 <= 2 * p.nstand;
        fast = __builtin_amdgcn_readfirstlane((int)fast) != 0;
        stores_in_flight = !(ABL & 16) && !LACC && fast;
        if (ABL & 16) {   // timing only: no epilogue (keep the accumulators live)
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) asm volatile("" :: "v"(accR[m][n][0]), "v"(accP[m][n][5]), "v"(accQ[m][n][9]));
        } else
        xcorr_store_cells<LACC>(p, c, row, col, live, fast, p.accumulate != 0, lane, accR, accP, accQ);
        if (p.stamps && lane == 0) p.stamps[((size_t)(c * p.nwg + wg) * 4 + wave) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
    wait_vmcnt<0>();   // no LDS-DMA may still be in flight when the wave ends
}

// ---------------------------------------------------------------------------------------
// Raw copy of a synchronously handed gulp into the staging area (xengXgpuKernel, reference call semantics: the caller may
// recycle its buffer on return, corr_block.py:445-452).  HBM-bound: 16 B per lane, U pieces in flight, grid-stride over
// 2048 work-groups like the CorrAcc map, non-temporal both ways (the source is read once; the copy is read by the contraction
// tens of microseconds later, through L2 misses either way: 32 MB against 4 MB of L2 per XCD).
// ---------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void gulp_copy_kernel(v4i* __restrict__ dst, const v4i* __restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (size_t k0 = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; k0 < n16; k0 += stride) {
        v4i y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t k = k0 + (size_t)u * blockDim.x;
            if (k < n16) y[u] = __builtin_nontemporal_load(src + k);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t k = k0 + (size_t)u * blockDim.x;
            if (k < n16) __builtin_nontemporal_store(y[u], dst + k);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bfXgpuSubSelect replacement (corr_subsel_block.py:298): gather + channel sum + conjugate
// ---------------------------------------------------------------------------------------
__global__ void subselect_kernel(const int32_t* __restrict__ xg, int32_t* __restrict__ out,
                                 const int32_t* __restrict__ vismap, const int32_t* __restrict__ conj,
                                 int nvis, int nchan_sum, int64_t per_chan, int64_t matlen) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x, co = blockIdx.y;
    if (v >= nvis) return;
    const int64_t w = vismap[v];
    int32_t r = 0, i = 0;
    for (int k = 0; k < nchan_sum; k++) {
        const int64_t o = (int64_t)(co * nchan_sum + k) * per_chan + w;
        r += xg[o];
        i += xg[matlen + o];
    }
    int2 res = make_int2(r, conj[v] ? -i : i);
    reinterpret_cast<int2*>(out)[(int64_t)co * nvis + v] = res;
}

// ---------------------------------------------------------------------------------------
// Full-correlation packet payloads on the device (CorrOutputFull: bfXgpuReorder on the host,
// corr_output_full_block.py:669, then one payload per dual-pol baseline s0 <= s1,
// :461-467 `reordered_data[s0, s1].tobytes()` = int32[npol][npol][nchan][2] (fmt 0) or, for the COR
// format, :512-519 int32[nchan][npol][npol][2] (fmt 1)).  Payload k = baseline (s0, s1) in sending order
// (s0 ascending, s1 from s0): k = s0*nstand - s0(s0-1)/2 + (s1 - s0).
//
// HBM-bound gather + transpose through LDS, driven by the GetOrder maps (any antpol_to_input).
// grid (s1, tile of 16 stands s0), 256 threads.  A tile is 64 rows = (s0 parity, 8 stands, p0, p1):
// with the identity input map the 32 rows of one parity are one 128-byte run of xGPU cells (16-byte loads of
// whole cells: a wave reads two full lines per plane and channel group).  Phase A: rows x channels -> LDS (int2 re|im, conjugated per map);
// phase B: every baseline's payload (4*nchan int2, contiguous) is written out in order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void packetize_kernel(const int32_t* __restrict__ xg, int2* __restrict__ out,
                                                        const int32_t* __restrict__ antpol_to_bl,
                                                        const int32_t* __restrict__ is_conj, int nstand, int nchan,
                                                        int64_t per_chan, int64_t matlen, int pitch, int fmt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pk_lds[];
    int2* tile = reinterpret_cast<int2*>(pk_lds);       // [64 rows][pitch] int2
    const int s1 = blockIdx.x, s0base = blockIdx.y * 16;
    if (s0base > s1) return;
    const int t = threadIdx.x;
    {
        // thread = (baseline slot bs of the tile's 16, channel group cg of 16).  bs = parity*8 + k: the 8 baselines
        // of one parity are neighbouring xGPU cells.  When the four map entries of a baseline lie in one aligned
        // 4-word cell (any input map that keeps a stand's two polarisations together) the cell is fetched with one
        // 16-byte load per plane and channel; otherwise word by word.
        const int bs = t & 15, cg = t >> 4;
        const int s0 = s0base + 2 * (bs & 7) + (bs >> 3);
        if (s0 <= s1 && s0 < nstand) {
            const size_t m = ((size_t)s0 * nstand + s1) * 4;
            const int4 w4 = *reinterpret_cast<const int4*>(antpol_to_bl + m);
            const int4 c4 = *reinterpret_cast<const int4*>(is_conj + m);
            const int wv[4] = {w4.x, w4.y, w4.z, w4.w};
            const bool cj[4] = {c4.x != 0, c4.y != 0, c4.z != 0, c4.w != 0};
            const int cell = wv[0] & ~3;
            const bool one_cell = (wv[1] & ~3) == cell && (wv[2] & ~3) == cell && (wv[3] & ~3) == cell;
            const int rbase = ((bs >> 3) << 5) | ((bs & 7) << 2);
#pragma unroll 6
            for (int c = cg; c < nchan; c += 16) {
                const int64_t o = (int64_t)c * per_chan;
                int re[4], im[4];
                if (one_cell) {
                    const int4 r4 = *reinterpret_cast<const int4*>(xg + o + cell);
                    const int4 i4 = *reinterpret_cast<const int4*>(xg + matlen + o + cell);
                    const int rr[4] = {r4.x, r4.y, r4.z, r4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
                    for (int pp = 0; pp < 4; pp++) {
                        const int k = wv[pp] & 3;
                        re[pp] = k == 0 ? rr[0] : k == 1 ? rr[1] : k == 2 ? rr[2] : rr[3];
                        im[pp] = k == 0 ? ii[0] : k == 1 ? ii[1] : k == 2 ? ii[2] : ii[3];
                    }
                } else {
#pragma unroll
                    for (int pp = 0; pp < 4; pp++) {
                        re[pp] = xg[o + wv[pp]];
                        im[pp] = xg[matlen + o + wv[pp]];
                    }
                }
#pragma unroll
                for (int pp = 0; pp < 4; pp++) tile[(rbase | pp) * pitch + c] = make_int2(re[pp], cj[pp] ? -im[pp] : im[pp]);
            }
        }
    }
    __syncthreads();
    // phase B: wave w writes the payloads of baselines w, w+4, ...; a lane moves two consecutive int2 (16 bytes) per step
    const int per_bl = 4 * nchan;                        // int2 elements per payload
    const int lane = t & 63, wave = t >> 6;
    for (int b = wave; b < 16; b += 4) {                 // b = 2*k + parity: s0 = s0base + b
        const int s0 = s0base + b;
        if (s0 > s1 || s0 >= nstand) continue;
        const int rowb = ((b & 1) << 5) | ((b >> 1) << 2);
        const int64_t k = (int64_t)s0 * nstand - ((int64_t)s0 * (s0 - 1)) / 2 + (s1 - s0);
        int2* dst = out + k * per_bl;
        if (fmt) {                                       // [chan][pol][pol][2]: e = 4c + pp
            for (int e = 2 * lane; e < per_bl; e += 128) {
                const int c = e >> 2, pp = e & 3;        // (pp is 0 or 2: the pair pp, pp+1 shares the channel)
                const int2 v0 = tile[(rowb | pp) * pitch + c], v1 = tile[(rowb | (pp + 1)) * pitch + c];
                *reinterpret_cast<int4*>(dst + e) = make_int4(v0.x, v0.y, v1.x, v1.y);
            }
        } else {                                         // [pol][pol][chan][2]: e = pp * nchan + c
            for (int pp = 0; pp < 4; pp++)
                for (int c = lane; c < nchan; c += 64) dst[pp * nchan + c] = tile[(rowb | pp) * pitch + c];
        }
    }
}

#include "xcorr_fused16.h"

#ifdef XENG_EXPERIMENTS
#include "experiments/xcorr_fp6.h"
#endif

}  // namespace xeng
