// Device kernels of the MI355X X-engine (gfx950 only).
//
// What they replace: the xGPU CUDA X-engine + its DEVSWIZZLE input swizzle that the
// reference reaches through _bf.bfXgpuKernel (corr_block.py:445; xGPU itself is an
// empty submodule in the reference tree).  Nothing here is derived from xGPU code:
// the design is a two-stage HBM-resident pipeline built for CDNA4.
//
//  stage 1  corner_turn_kernel      (HBM-bound, one launch per gulp)
//     in   uint8[t][c][i]  4+4 bit   (corr_block.py:115-116)
//     out  stash[c][ib][kt][sub][lane][16 B]   "MFMA-fragment-major":
//          ib  = 64-input block, kt = 32-sample K tile, sub = 32-input half,
//          lane = h*32 + r holds the 16 samples t = kt*32 + 16h + (0..15) of input
//          i = ib*64 + sub*32 + r, still packed 4+4 bit.  One (kt, sub) fragment is
//          1 KiB and is byte-for-byte the A (or B) register image of
//          v_mfma_i32_32x32x32_i8, so the contraction kernel moves it HBM -> LDS with
//          linear global_load_lds_dwordx4 and LDS -> VGPR with conflict-free ds_read_b128.
//
//  stage 2  xcorr_mfma_kernel       (int8 MFMA-bound, one launch per dump/flush)
//     contracts all staged gulps (K = n_gulps * ntime) in one pass, so the 191 MB
//     int32 accumulator is written once per integration instead of being
//     read-modify-written every 480 samples.
//     Nibbles are sign-extended "for free": (x & 0xF0F0F0F0) is 16*re as int8,
//     ((x<<4) & 0xF0F0F0F0) is 16*im; all products are exact multiples of 256 and
//     the epilogue shifts them back (>> 8).  No negated operand is needed: the
//     imaginary part is kept as two accumulators P = sum ai*br, Q = sum ar*bi and
//     subtracted in the epilogue (-(-8) does not fit int8 after the x16 scaling).
//     Bound: |acc| <= 2*128*128*K < 2^31  =>  K <= 65535 samples per launch.
//     Epilogue writes the xGPU register-tile order (corr_block.py:27-58) directly:
//     a lane pair exchanges two registers by DPP so every lane stores whole
//     16-byte cells [polR][polC].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xeng {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int FRAG_BYTES = 1024;          // one 32-input x 32-sample fragment
constexpr int KT_BYTES = 2 * FRAG_BYTES;  // one 64-input block x 32 samples
constexpr int XC_NBUF = 3;                // LDS ring depth
constexpr int XC_NSLOT = 4;               // 64-input blocks resident per stage

// Work-group descriptor: which 64-input blocks a work-group stages (one per wave),
// and which (row block, col block) tile each of its 4 waves contracts.
struct WgDesc {
    uint8_t slot_blk[XC_NSLOT];  // 64-input block loaded by wave w into LDS slot w
    uint8_t wave_a[4];           // LDS slot of the wave's row block (0xFF: wave idle)
    uint8_t wave_b[4];           // LDS slot of the wave's column block
    uint8_t nwave;
    uint8_t pad[3];
};

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

// grid (gkt, nchan), 256 threads.  Each work item is (input quad q, k-half h): it reads
// 16 samples x 4 inputs as 16 coalesced dwords (a wave covers 256 contiguous bytes of one
// [t][c] row per load) and writes the four inputs' 16-byte fragment entries (64 contiguous B).
__global__ __launch_bounds__(256) void corner_turn_kernel(const uint8_t* __restrict__ in,
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

struct XcorrParams {
    const uint8_t* stash;
    int32_t* out;
    const WgDesc* descs;
    int nwg, nchan, nblk64, cap_kt, nkt, nstand;
    int64_t per_chan, matlen;
    int accumulate;
};

template <int KT_STAGE>
__global__ __launch_bounds__(256, 1) void xcorr_mfma_kernel(XcorrParams p) {
    constexpr int SLOT_BYTES = KT_STAGE * KT_BYTES;
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES;
    constexpr int NLOAD = 2 * KT_STAGE;  // 1 KiB LDS-DMA pieces per wave per stage
    __shared__ __attribute__((aligned(16))) uint8_t lds[XC_NBUF * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

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

    auto issue = [&](int s, int buf) {
        const uint8_t* g = gsrc + (size_t)s * SLOT_BYTES;
        uint8_t* l = lds + buf * STAGE_BYTES + wave * SLOT_BYTES;
#pragma unroll
        for (int n = 0; n < NLOAD; n++)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g + n * FRAG_BYTES),
                (__attribute__((address_space(3))) void*)(l + n * FRAG_BYTES), 16, 0, 0);
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

    auto compute = [&](int buf) {
        const uint8_t* base = lds + buf * STAGE_BYTES;
#pragma unroll
        for (int k = 0; k < KT_STAGE; k++) {
            const v4i a0 = *reinterpret_cast<const v4i*>(base + a_off + k * KT_BYTES);
            const v4i a1 = *reinterpret_cast<const v4i*>(base + a_off + k * KT_BYTES + FRAG_BYTES);
            const v4i b0 = *reinterpret_cast<const v4i*>(base + b_off + k * KT_BYTES);
            const v4i b1 = *reinterpret_cast<const v4i*>(base + b_off + k * KT_BYTES + FRAG_BYTES);
            const v4i M = (v4i)(0xF0F0F0F0);
            v4i ar[2], ai[2], br[2], bi[2];
            ar[0] = a0 & M; ai[0] = (a0 << 4) & M;
            ar[1] = a1 & M; ai[1] = (a1 << 4) & M;
            br[0] = b0 & M; bi[0] = (b0 << 4) & M;
            br[1] = b1 & M; bi[1] = (b1 << 4) & M;
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    accR[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[m], br[n], accR[m][n], 0, 0, 0);
                    accP[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[m], br[n], accP[m][n], 0, 0, 0);
                    accQ[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[m], bi[n], accQ[m][n], 0, 0, 0);
                    accR[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[m], bi[n], accR[m][n], 0, 0, 0);
                }
        }
    };

    // 3-deep LDS ring, one barrier per stage.  At the top of stage s the wave's own pieces of
    // stage s have landed (counted vmcnt leaves stage s+1 in flight); the barrier then proves
    // everybody's have, and that every wave is done reading buffer (s-1)%3, which the
    // LDS-DMA of stage s+2 overwrites next.
    issue(0, 0);
    if (nstage > 1) issue(1, 1);
    int buf = 0;
    int s = 0;
    for (; s + 2 < nstage; s++) {
        wait_vmcnt<NLOAD>();
        __builtin_amdgcn_s_barrier();
        issue(s + 2, buf >= 1 ? buf - 1 : XC_NBUF - 1);
        compute(buf);
        buf = (buf + 1 == XC_NBUF) ? 0 : buf + 1;
    }
    if (nstage > 1) {
        wait_vmcnt<NLOAD>();
        __builtin_amdgcn_s_barrier();
        compute(buf);
        buf = (buf + 1 == XC_NBUF) ? 0 : buf + 1;
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute(buf);
    if (!active) return;

    // ---- epilogue: D[i][j] = sum x_i conj(x_j), lane = column j, register = row i.
    // MFMA C/D map (32x32): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    const int64_t qs = ((int64_t)(p.nstand / 2 + 1) * p.nstand) / 4;
    int32_t* out_r = p.out + (int64_t)c * p.per_chan;
    int32_t* out_i = out_r + p.matlen;
    const int odd = lane & 1;
    const int cpar = (lane >> 1) & 1;          // C & 1 of this lane's column station
    const int64_t quad = 2 * cpar + odd;       // quadrant of the cell this lane stores
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int ibase = blk_a * 64 + m * 32, jbase = blk_b * 64 + n * 32;
            const int Ch = (jbase >> 2) + ((lane & 31) >> 2);
            const int C = 2 * Ch + cpar;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int Rh = (ibase >> 2) + 2 * u + (lane >> 5);
                const int R = 2 * Rh + odd;
                int vr[4], vi[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    vr[v] = accR[m][n][4 * u + v] >> 8;
                    vi[v] = (accP[m][n][4 * u + v] - accQ[m][n][4 * u + v]) >> 8;
                }
                // even lane keeps station R&1=0 (regs 0,1), odd lane station R&1=1 (regs 2,3);
                // the partner lane (other polC) supplies the missing two words.
                const int gr0 = dpp_xor1(odd ? vr[0] : vr[2]), gr1 = dpp_xor1(odd ? vr[1] : vr[3]);
                const int gi0 = dpp_xor1(odd ? vi[0] : vi[2]), gi1 = dpp_xor1(odd ? vi[1] : vi[3]);
                int4 cr = odd ? make_int4(gr0, vr[2], gr1, vr[3]) : make_int4(vr[0], gr0, vr[1], gr1);
                int4 ci = odd ? make_int4(gi0, vi[2], gi1, vi[3]) : make_int4(vi[0], gi0, vi[1], gi1);
                if (Rh >= Ch && R < p.nstand && C < p.nstand) {
                    const int64_t w = (quad * qs + tri64(Rh, Ch)) * 4;
                    int4* pr = reinterpret_cast<int4*>(out_r + w);
                    int4* pi = reinterpret_cast<int4*>(out_i + w);
                    if (p.accumulate) {
                        const int4 o_r = *pr, o_i = *pi;
                        cr.x += o_r.x; cr.y += o_r.y; cr.z += o_r.z; cr.w += o_r.w;
                        ci.x += o_i.x; ci.y += o_i.y; ci.z += o_i.z; ci.w += o_i.w;
                    }
                    *pr = cr;
                    *pi = ci;
                }
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

}  // namespace xeng
