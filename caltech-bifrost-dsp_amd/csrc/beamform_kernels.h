// Beamformer device kernels (gfx950).
//
// Replace the bifrost beamform library behind _bf.bfBeamformRun / bfBeamformIntegrate
// (beamform_block.py:449, beamform_sum_beams_block.py:245).  The reference prototype
// (bf_src/cublas_beamform.cu) expands the 4+4-bit voltages to a cf32 copy 8x their size
// (trans_4bit_to_float :21-41) and runs a cuBLAS CF32 batched GEMM (:248-276) over it.
// Here the nibble -> float conversion is fused into the GEMM's operand fetch: the packed
// voltages go HBM -> LDS once and are converted in registers right before the MFMA, so the
// 519 MB fp32 round trip of the prototype does not exist.
//
// beamform_f32_kernel: per channel  out[b][t] = sum_i w[b][i] * x[t][i]  (no conjugation,
// beamformer_test.py:76-84) as 4 real fp32 MFMAs (v_mfma_f32_32x32x2_f32, exact fp32 fma
// chain) per pair of inputs:  re += wr*xr + (-wi)*xi ;  im += wr*xi + wi*xr.
// Work-group = 4 waves = 32 beams x 128 samples of one channel; the K (input) loop streams
// 64-input chunks of W (cf32) and X (packed) through LDS.  MFMA lane-half h contracts inputs
// 32h..32h+31 of a chunk (the contraction order is free), so each lane reads contiguous K.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slab.h"

namespace xeng {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BF_KC = 64;                 // inputs per LDS chunk
constexpr int BF_WS = BF_KC * 8 + 16;     // W row stride (bytes): conflict-free ds_read_b128
constexpr int BF_XS = BF_KC + 4;          // X row stride (bytes): conflict-free ds_read_b32
constexpr int BF_NT = 128;                // samples per work-group

// A gulp in up to two parts (round 4): samples [0, split) start at `in`, samples [split, ntime) at `in1` -- two consecutive spans
// of the input ring taken as one 960-sample gulp without a gathered copy (the reference reads GPU_NGULP = 2 capture gulps per
// beamformer gulp, lwa352-pipeline.py:172,279-282; bifrost's one circular buffer gives that for free).  One part: split = ntime.
//
// DESC instantiations (round 4, xengBeamformRunSlabs): each part is described in DEVICE memory by a GulpDesc (slab.h) -- a slab
// of F-engine packets read where it lies, or the scratch copy that an irregular slab was scattered into.  Then the strides
// are per part: sample, channel and 64-input block steps come from the descriptor (scalar loads), and a row of inputs is no
// longer contiguous: input i sits (i >> 6) * b_stride + (i & 63) bytes into its row (16-byte pieces never straddle a block).
// The default instantiations compile to the code they were before.
// TAB instantiations (round 5, a lossy link): a part may be a slab read through its packet INDEX (descriptor pad 2, slab.h: SlabIndexPrep) --
// the row of (sample t, block b) lies in the packet whose slot the index names, or nowhere (then: a page of zeros).  For such a part
// row() returns the address of the sample's index row instead of its data, and piece() turns an entry into an address.
template <bool DESC, bool TAB = false>
struct GulpAddr {
    const uint8_t* p[2];
    uint32_t ts[2], cs[2], bs[2];
    int split;
    uint32_t mode[2];                 // TAB: the descriptors' pad (2: by index)
    const uint32_t* tab[2];
    int nblk;
    __device__ __forceinline__ GulpAddr(const uint8_t* in, const uint8_t* in1, int split_, const GulpDesc* __restrict__ gd, int nchan, int ninput) {
        split = split_;
        nblk = ninput >> 6;
        mode[0] = mode[1] = 0; tab[0] = tab[1] = nullptr;
        if (DESC) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                p[k] = gd[k].base; ts[k] = gd[k].t_stride; cs[k] = gd[k].c_stride; bs[k] = gd[k].b_stride;
                if (TAB) { mode[k] = gd[k].pad; tab[k] = gd[k].table; }
            }
        } else {
            p[0] = in; p[1] = in1;
            ts[0] = ts[1] = (uint32_t)nchan * (uint32_t)ninput; cs[0] = cs[1] = (uint32_t)ninput; bs[0] = bs[1] = 64;
        }
    }
    // first byte of the inputs of (sample t, channel c); *bstride: the step between that row's 64-input blocks
    __device__ __forceinline__ const uint8_t* row(int t, int c, uint32_t* bstride) const {
        const bool second = t >= split;
        const uint8_t* base = second ? p[1] : p[0];
        const uint32_t tstr = DESC ? (second ? ts[1] : ts[0]) : ts[0], cstr = DESC ? (second ? cs[1] : cs[0]) : cs[0];
        *bstride = DESC ? (second ? bs[1] : bs[0]) : 64u;
        if (TAB && (second ? mode[1] : mode[0]) == 2u)      // (the sample's index row: see piece())
            return reinterpret_cast<const uint8_t*>((second ? tab[1] : tab[0]) + (size_t)(second ? t - split : t) * nblk);
        return base + (size_t)(second ? t - split : t) * tstr + (size_t)c * cstr;
    }
    // TAB: is the part that holds sample t read through its index?  (uniform over the 16 rows of an LDS-DMA piece: parts begin on
    // 16-sample boundaries)
    __device__ __forceinline__ bool indexed(int t) const { return TAB && (t >= split ? mode[1] : mode[0]) == 2u; }
    // ... then: the 16 bytes at input i of (sample t, channel c), given the sample's index row (what row() returned) -- in the packet
    // the entry names if it is of this call's generation, else in the page of zeros
    __device__ __forceinline__ const uint8_t* piece(const uint8_t* row_, int t, int c, int i, const uint8_t* zeros) const {
        return piece_from(reinterpret_cast<const uint32_t*>(row_)[i >> 6], t, c, i, zeros);
    }
    __device__ __forceinline__ const uint8_t* piece_from(uint32_t e, int t, int c, int i, const uint8_t* zeros) const {
        const bool second = t >= split;
        const uint32_t gen = second ? ts[1] : ts[0], stride = second ? bs[1] : bs[0];
        const uint8_t* base = second ? p[1] : p[0];
        const bool have = (e >> SLAB_GEN_SHIFT) == gen && (e & SLAB_SLOT_MASK) != 0u;
        return have ? base + (size_t)((e & SLAB_SLOT_MASK) - 1u) * stride + 32u + (size_t)c * 64u + (size_t)(i & 63) : zeros + (i & 63);
    }
};
template <bool DESC>
__device__ __forceinline__ size_t gulp_col(int i, uint32_t bstride) {
    return DESC ? (size_t)(i >> 6) * bstride + (size_t)(i & 63) : (size_t)i;
}

template <bool DESC = false>
__global__ __launch_bounds__(256) void beamform_f32_kernel(const uint8_t* __restrict__ in,
                                                           const float* __restrict__ w,
                                                           float* __restrict__ out, int ntime, int nchan,
                                                           int ninput, int nbeam, const uint8_t* __restrict__ in1, int split,
                                                           const GulpDesc* __restrict__ gd) {
    const GulpAddr<DESC> ga(in, in1, split, gd, nchan, ninput);
    __shared__ __attribute__((aligned(16))) uint8_t lds[32 * BF_WS + BF_NT * BF_XS];
    uint8_t* ldsW = lds;
    uint8_t* ldsX = lds + 32 * BF_WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.y, b0 = blockIdx.z * 32, t0 = blockIdx.x * BF_NT;
    const int h = lane >> 5, j = lane & 31;

    v16f acc_r = (v16f)(0.f), acc_i = (v16f)(0.f);

    const int nchunk = (ninput + BF_KC - 1) / BF_KC;
    for (int ch = 0; ch < nchunk; ch++) {
        const int k0 = ch * BF_KC;
        __syncthreads();  // previous chunk fully consumed
        // W chunk: 32 beams x 64 inputs x 8 B, 16 B per thread, 4 passes
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int e = pass * 256 + tid;  // 16-byte element: row = e / 32, col16 = e % 32
            const int row = e >> 5, col = e & 31;
            const int b = b0 + row, i = k0 + col * 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < nbeam && i < ninput) {
                const float* src = w + (((size_t)c * nbeam + b) * ninput + i) * 2;
                if (i + 1 < ninput) v = *reinterpret_cast<const float4*>(src);
                else { v.x = src[0]; v.y = src[1]; }
            }
            *reinterpret_cast<float4*>(ldsW + row * BF_WS + col * 16) = v;
        }
        // X chunk: 128 samples x 64 B, 16 B per thread, 2 passes
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int e = pass * 256 + tid;
            const int row = e >> 2, col = e & 3;
            const int t = t0 + row, i = k0 + col * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t < ntime && i < ninput) {
                uint32_t bstr;
                const uint8_t* src = ga.row(t, c, &bstr);
                src += gulp_col<DESC>(i, bstr);
                if (i + 16 <= ninput) v = *reinterpret_cast<const uint4*>(src);
                else {
                    uint32_t tmp[4] = {0, 0, 0, 0};
                    for (int q = 0; q < ninput - i; q++) tmp[q >> 2] |= (uint32_t)src[q] << (8 * (q & 3));
                    v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            uint32_t* dst = reinterpret_cast<uint32_t*>(ldsX + row * BF_XS + col * 16);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        __syncthreads();
        const uint8_t* wrow = ldsW + j * BF_WS + h * 256;            // beam j, inputs 32h..
        const uint8_t* xrow = ldsX + (wave * 32 + j) * BF_XS + h * 32;  // sample j of this wave
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t xw = *reinterpret_cast<const uint32_t*>(xrow + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const float4 wv = *reinterpret_cast<const float4*>(wrow + (4 * q + e) * 8);
#pragma unroll
                for (int z = 0; z < 2; z++) {
                    const int sh = 8 * (e + z);
                    // hi nibble = real, lo nibble = imag, two's complement (beamformer_test.py:69-73)
                    const float xr = (float)(int)__builtin_amdgcn_sbfe((int)xw, sh + 4, 4);
                    const float xi = (float)(int)__builtin_amdgcn_sbfe((int)xw, sh, 4);
                    const float wr = z ? wv.z : wv.x, wi = z ? wv.w : wv.y;
                    acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(wr, xr, acc_r, 0, 0, 0);
                    acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(wr, xi, acc_i, 0, 0, 0);
                    acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(-wi, xi, acc_r, 0, 0, 0);
                    acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(wi, xr, acc_i, 0, 0, 0);
                }
            }
        }
    }
    // C/D map: col (sample) = lane&31, row (beam) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int t = t0 + wave * 32 + j;
    if (t < ntime) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int b = b0 + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (b < nbeam)
                *reinterpret_cast<float2*>(out + (((size_t)c * nbeam + b) * ntime + t) * 2) =
                    make_float2(acc_r[g], acc_i[g]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bf16x3 beamformer (default).  The voltages (-8..7) are exact in bf16; every fp32 weight is split
// exactly into three bf16 terms  w = w1 + w2 + w3 + O(2^-25 |w|)  by beam_weights_prep_kernel, and
//   out = sum_i (w1 + w2 + w3) * x
// runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: products are exact (8-bit x 4-bit
// significands), so the only roundings are the fp32 adds of the accumulator, as in the fp32 path.
// 12 bf16 MFMAs per 16 inputs replace 32 fp32 MFMAs (v_mfma_f32_32x32x2_f32) at 1/2 the cycles each.
//
// Prepared weights: Wp[c][beam tile][chunk][term 3][re|im][32 beams][64 inputs] bf16, rows padded to
// 144 B so the ds_read_b128 of the A operand is bank-conflict-free; zero for beams/inputs past the end.
// ---------------------------------------------------------------------------------------
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// LDS-DMA issued from inline asm: hipcc tracks the builtin form as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read of the array, which serialises the prefetch of
// chunk n+1 with the compute of chunk n.  From asm the transfer is invisible to that bookkeeping; the
// kernel's own `s_waitcnt vmcnt(0)` + barrier orders it (cdna_hip_programming.md 5.7, M0 recipe).
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)p;
}
constexpr int BF3_KC = 32;                           // inputs per LDS chunk
constexpr int BF3_ROW = BF3_KC * 2;                  // 64 B of bf16 per (term, re|im, beam) row, no padding:
                                                     //   16-byte piece p of row j sits at position p ^ ((j>>2)&3)
constexpr int BF3_WCHUNK = 3 * 2 * 32 * BF3_ROW;      // 12288 B of weights per chunk
#ifndef BF3_NW
#define BF3_NW 4                                      // waves per work-group (measured: 4 -> 52.5 us, 8 -> 58.4 (384 work-groups
#endif                                                //  leave half the CUs with one), 6 -> 64.6 (waves land unevenly on the SIMDs))
constexpr int BF3_NT = 32 * BF3_NW;                   // samples per work-group: one 32-sample MFMA column tile per wave
constexpr int BF3_XCHUNK = BF3_NT * BF3_KC;           // packed voltages per chunk (1 KiB per wave)
constexpr int BF3_STAGE = BF3_WCHUNK + BF3_XCHUNK;    // 16 KiB for 4 waves
constexpr int BF3_WSLOTS = (12 + BF3_NW - 1) / BF3_NW;  // weight pieces issued per wave and chunk (12 pieces in all)
constexpr int BF3_RING = 3;                           // stages: the LDS-DMA runs two chunks ahead of the MFMAs

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;   // weights are finite
}

// grid (ceil(ninput/64), nbtile, nchan), 256 threads: thread = (beam tid/8, 8 inputs (tid%8)*8) of a 64-input
// span = two chunks.  Layout Wp[c][beam tile][chunk][term][re|im][32 beams][32 inputs] bf16.
// route (may be null): per (channel, beam tile) 0 = the tile runs on the int8x3 kernel (nothing to prepare here), 1 = here
__global__ __launch_bounds__(256) void beam_weights_prep_kernel(const float* __restrict__ w, uint8_t* __restrict__ wp,
                                                                int nchan, int nbeam, int ninput, int nchunk, int nbtile,
                                                                const int* __restrict__ route) {
    const int sp = blockIdx.x, bt = blockIdx.y, c = blockIdx.z;
    if (route && !route[c * nbtile + bt]) return;
    const int beam = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * 8;
    const int b = bt * 32 + beam;
    const int ch = 2 * sp + (k0 >> 5);                 // chunk of this thread's 8 inputs
    if (ch >= nchunk) return;
    uint32_t t[3][2][4];   // [term][re|im][4 dwords = 8 bf16]
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = sp * 64 + k0 + j;
        float re = 0.f, im = 0.f;
        if (b < nbeam && i < ninput) {
            const float2 v = *reinterpret_cast<const float2*>(w + (((size_t)c * nbeam + b) * ninput + i) * 2);
            re = v.x; im = v.y;
        }
        float r[2] = {re, im};
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            float rem = r[comp];
#pragma unroll
            for (int term = 0; term < 3; term++) {
                const uint32_t hb = f32_to_bf16_rne(rem);
                rem -= __uint_as_float(hb << 16);          // exact: the bf16 term cancels the leading bits
                if (j & 1) t[term][comp][j >> 1] |= hb << 16; else t[term][comp][j >> 1] = hb;
            }
        }
    }
    uint8_t* base = wp + (((size_t)c * nbtile + bt) * nchunk + ch) * BF3_WCHUNK;
    const int pos = ((k0 & 31) >> 3) ^ ((beam >> 2) & 3);  // swizzled 16-byte piece of the row
#pragma unroll
    for (int term = 0; term < 3; term++)
#pragma unroll
        for (int comp = 0; comp < 2; comp++)
            *reinterpret_cast<uint4*>(base + ((term * 2 + comp) * 32 + beam) * BF3_ROW + pos * 16) =
                make_uint4(t[term][comp][0], t[term][comp][1], t[term][comp][2], t[term][comp][3]);
}

__device__ __forceinline__ v8bf as_v8bf(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = {a, b, c, d};
    return __builtin_bit_cast(v8bf, v);
}

// grid nchan * nbtile * ceil(ntime/BF3_NT) (1-D), 64*BF3_NW threads; wave w owns samples t0 + 32w .. +31.
// Config 4 is a single round of 768 work-groups (3 per CU), each a chain of 22 chunk steps: the kernel is
// bound by how well the staging latency hides, not by MFMA throughput.  So the LDS-DMA runs two chunks
// ahead on a ring of three 16 KiB stages (counted vmcnt, one barrier per chunk), and three work-groups per
// CU (48 KiB each) interleave on every SIMD.
template <bool DESC = false, bool TAB = false>
__global__ __launch_bounds__(64 * BF3_NW, BF3_NW == 8 ? 2 : 3) void beamform_bf16x3_kernel(const uint8_t* __restrict__ in,
                                                                 const uint8_t* __restrict__ wp,
                                                                 float* __restrict__ out, int ntime, int nchan,
                                                                 int ninput, int nbeam, int nchunk, int nbtile,
                                                                 const int* __restrict__ route, const uint8_t* __restrict__ in1, int split,
                                                                 const GulpDesc* __restrict__ gd, const uint8_t* __restrict__ zeros = nullptr) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[BF3_RING * BF3_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid of nchan*nbtile*nttile blocks.  Blocks b and b+8 share an XCD: give each XCD whole channels
    // so the work-groups of a channel read its split weights through one L2 (speed only).
    const int nttile = (ntime + BF3_NT - 1) / BF3_NT, per_c = nbtile * nttile;
    int c, rem;
    if (DESC && (nchan & 15) == 0) {      // packet slabs: channels 2k, 2k+1 share every cache line of a packet -- both on one XCD, back to back
        const int b = blockIdx.x, slot = b >> 3, q = slot / per_c;
        c = 16 * (q >> 1) + 2 * (b & 7) + (q & 1); rem = slot % per_c;
    } else if ((nchan & 7) == 0) { const int b = blockIdx.x, slot = b >> 3; c = (b & 7) + 8 * (slot / per_c); rem = slot % per_c; }
    else { c = blockIdx.x / per_c; rem = blockIdx.x % per_c; }
    const int bt = rem / nttile, t0 = (rem % nttile) * BF3_NT;
    if (route && !route[c * nbtile + bt]) return;      // this (channel, beam tile) runs on the int8x3 kernel
    const int h = lane >> 5, j = lane & 31;
    const uint8_t* wsrc = wp + (((size_t)c * nbtile + bt) * nchunk) * BF3_WCHUNK + lane * 16;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    // X piece `wave` of a chunk = samples 32*wave .. +31, 32 B each; this lane: row 32*wave + lane/2, 16-byte half
    // (lane&1) ^ ((row>>3)&1): the swizzle (applied on the source side, the LDS side of the DMA is lane-linear)
    // makes the 8-byte fragment reads below bank-conflict-free.  Rows past ntime: any valid row (never stored).
    int xt = t0 + wave * 32 + (lane >> 1);
    if (xt >= ntime) xt = ntime - 1;
    const GulpAddr<DESC, TAB> ga(in, in1, split, gd, nchan, ninput);
    uint32_t xbstr;
    const uint8_t* xsrc = ga.row(xt, c, &xbstr);
    // TAB: the part of this wave's rows is read through its index (xsrc: the sample's index row).  The entry is looked up when the
    // piece is issued -- this kernel only runs the few routed tiles, its staging pipeline is not worth a prefetch register
    const bool xidx = TAB && __builtin_amdgcn_readfirstlane((int)ga.indexed(xt)) != 0;
    const int xhalf = ((lane & 1) ^ ((lane >> 4) & 1)) * 16;
    // every stage costs exactly BF3_WSLOTS + 1 pieces per wave on the vmcnt counter (chunks past the end re-read
    // the last one; weight slots past the 12th piece re-copy an earlier piece onto itself)
    auto issue = [&](int ch, int buf) {
        const int cs = ch < nchunk ? ch : nchunk - 1;
        const uint32_t l = lds0 + buf * BF3_STAGE;
#pragma unroll
        for (int n = 0; n < BF3_WSLOTS; n++) {
            const int pc = (wave + BF3_NW * n) % 12;
            lds_dma16(wsrc + (size_t)cs * BF3_WCHUNK + pc * 1024, l + pc * 1024);
        }
        int i = cs * BF3_KC + xhalf;
        if (i + 16 > ninput) i = 0;                    // columns past the end meet zero weights
        lds_dma16(xidx ? ga.piece(xsrc, xt, c, i, zeros) : xsrc + gulp_col<DESC>(i, xbstr), l + BF3_WCHUNK + wave * 1024);
    };
    v16f acc_r = (v16f)(0.f), acc_i = (v16f)(0.f);
    issue(0, 0);
    issue(1, 1);
    int buf = 0, nbuf = 2;
    for (int ch = 0; ch < nchunk; ch++) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BF3_WSLOTS + 1) : "memory");   // this wave's pieces of chunk ch have landed (ch+1 in flight)
        __builtin_amdgcn_s_barrier();                  // ... for all waves; and everybody is done reading chunk ch-1,
        issue(ch + 2, nbuf);                           // whose buffer the DMA of chunk ch+2 now overwrites
        const uint8_t* lw = lds + buf * BF3_STAGE + j * BF3_ROW;
        const uint8_t* lx = lds + buf * BF3_STAGE + BF3_WCHUNK + (wave * 32 + j) * BF3_KC + h * 8;
#pragma unroll
        for (int s = 0; s < 2; s++) {                  // 16 inputs per step: this lane-half takes 8 of them
            const uint2 xb = *reinterpret_cast<const uint2*>(lx + ((s ^ ((j >> 3) & 1)) * 16));
            uint32_t xr[4], xi[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {              // bytes 2q, 2q+1 -> one dword of two bf16
                const uint32_t wd = q < 2 ? xb.x : xb.y;
                const int sh = 16 * (q & 1);
                // hi nibble = real, lo nibble = imag, two's complement (beamformer_test.py:69-73);
                // small integers convert exactly; bf16 = upper half of the fp32
                const float r0 = (float)(int)__builtin_amdgcn_sbfe((int)wd, sh + 4, 4);
                const float r1 = (float)(int)__builtin_amdgcn_sbfe((int)wd, sh + 12, 4);
                const float i0 = (float)(int)__builtin_amdgcn_sbfe((int)wd, sh, 4);
                const float i1 = (float)(int)__builtin_amdgcn_sbfe((int)wd, sh + 8, 4);
                xr[q] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                xi[q] = __builtin_amdgcn_perm(__float_as_uint(i1), __float_as_uint(i0), 0x07060302u);
            }
            const v8bf Xr = as_v8bf(xr[0], xr[1], xr[2], xr[3]);
            const v8bf Xi = as_v8bf(xi[0], xi[1], xi[2], xi[3]);
            const v8bf nXi = as_v8bf(xi[0] ^ 0x80008000u, xi[1] ^ 0x80008000u, xi[2] ^ 0x80008000u, xi[3] ^ 0x80008000u);
            const int wpos = ((2 * s + h) ^ ((j >> 2) & 3)) * 16;
#pragma unroll
            for (int term = 2; term >= 0; term--) {    // smallest term first
                const uint4 a = *reinterpret_cast<const uint4*>(lw + ((term * 2 + 0) * 32) * BF3_ROW + wpos);
                const uint4 b = *reinterpret_cast<const uint4*>(lw + ((term * 2 + 1) * 32) * BF3_ROW + wpos);
                const v8bf Wr = as_v8bf(a.x, a.y, a.z, a.w), Wi = as_v8bf(b.x, b.y, b.z, b.w);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr, Xr, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr, Xi, acc_i, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wi, nXi, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wi, Xr, acc_i, 0, 0, 0);
            }
        }
        buf = buf + 1 == BF3_RING ? 0 : buf + 1;
        nbuf = nbuf + 1 == BF3_RING ? 0 : nbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
    // C/D map: col (sample) = lane&31, row (beam) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int t = t0 + wave * 32 + j;
    if (t < ntime) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int b = bt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (b < nbeam)
                *reinterpret_cast<float2*>(out + (((size_t)c * nbeam + b) * ntime + t) * 2) =
                    make_float2(acc_r[g], acc_i[g]);
        }
    }
}

// =======================================================================================
// int8 x 3 beamformer: fixed-point weights on the int8 MFMA (twice the bf16 rate, half the weight bytes).
//
// Every fp32 weight component is quantised per (channel, beam) row to q = round(w / wmax * QMAX),
// QMAX = 127*(255^2 + 255 + 1), and written as three balanced base-255 digits d1, d2, d3 in [-127, 127]
// (q = d1*255^2 + d2*255 + d3; the representation is symmetric, so -q is the negated digits).  The voltages
// enter as 16*re / 16*im (the mask trick of the X-engine: no conversion at all).  All products and the int32
// sums over the inputs are exact; per digit plane
//     Tre = sum (dr * 16xr - di * 16xi),   Tim = sum (dr * 16xi + di * 16xr)          (|T| <= 2*127*128*ninput)
// and the epilogue combines   out = (wmax / QMAX / 16) * (255^2 * T1 + 255 * T2 + T3)   in fp32.
// Quantisation step = wmax * 1.2e-7 (an fp32 ulp of the row's largest weight): the output error is
// <= 4e-6 * wmax/wrms of the output RMS in the worst (fully coherent) case and ~1e-7 for ordinary data, inside
// the 1e-5 bar; measured in tests/test_beamform_gpu.py.
//
// Layout Wq[c][beam tile][32-input K step][digit][Wr | Wi][lane = 32h + beam][16 B]: the A-operand image of
// v_mfma_i32_32x32x32_i8 (lane holds inputs 16h..16h+15 of its beam), 6 KiB per K step; scale[c][beam];
// wsum[c][beam tile][digit][beam] = sum over the inputs of the Wi digit (the "- di * 16xi" term runs as
// di * 16*(~xi) = -di*16xi - 16*di, so no negated plane is stored, staged or read; the accumulator starts at 16*wsum).
// Kernel structure as beamform_bf16x3_kernel: ring of BI_RING stages (two K steps: 12 KiB of digits + 8 KiB of
// voltages), LDS-DMA one chunk ahead, one barrier per chunk, three work-groups per CU.
// =======================================================================================
#ifndef BI_KSTEPS
#define BI_KSTEPS 2
#endif
constexpr int BI_KS = BI_KSTEPS;                            // int8 MFMA K steps (32 inputs each) per chunk
constexpr int BI_KC = 32 * BI_KS;                   // inputs per chunk
constexpr int BI_WSTEP = 3 * 2 * 1024;              // per K step: digits x {Wr, Wi} x 1 KiB operand image
constexpr int BI_WCHUNK = BI_KS * BI_WSTEP;         // 12 KiB
constexpr int BI_NT = 128;                          // samples per work-group (4 waves x 32)
constexpr int BI_XCHUNK = BI_NT * BI_KC;            // 8 KiB of packed voltages per chunk
constexpr int BI_STAGE = BI_WCHUNK + BI_XCHUNK;     // 20 KiB
constexpr int BI_WSLOTS = (BI_KS * 6 + 3) / 4;      // weight pieces issued per wave and chunk
constexpr int BI_XSLOTS = BI_XCHUNK / 1024 / 4;     // voltage pieces issued per wave and chunk
#ifndef BI_RING_STAGES
#define BI_RING_STAGES 2
#endif
constexpr int BI_RING = BI_RING_STAGES;
constexpr int BI_QMAX = 127 * (255 * 255 + 255 + 1);

// ---- precision of the fixed-point weights (what keeps the int8 route inside the 1e-5 bar for ANY weights) --------
// A row scale taken from the row maximum makes the quantisation step of every weight 1.2e-7 of the LARGEST one.  A few
// dominant weights (a huge calibration gain on a dead or quiet input) would then cost the ordinary weights their
// significant bits while contributing nothing to the output.  So per (channel, beam) row:
//   * up to BI_ROW_OUT entries that stand out by whole binades are *outliers*: their digits are zero and the
//     product w * x is added in fp32 in the kernel's epilogue (exact to fp32 rounding, like the reference's CF32 GEMM,
//     bf_src/cublas_beamform.cu:248-276).  Rule: E = the smallest fp32 exponent such that at most BI_ROW_OUT entries
//     have a larger exponent of max(|re|, |im|) and none of those lies within BI_GAP_BINADES binades (they stand
//     out: the top of a smooth distribution is not an outlier); the row scale is the exact maximum of the rest.
//   * if the largest remaining entry is still more than BI_GUARD_BINADES binades above the row's MEDIAN non-zero entry
//     (a heavy tail rather than a few outliers: the output may be made by weights that are small against the row
//     scale), or a beam tile collects more than BI_TILE_OUT distinct outlier inputs, the (channel, beam tile) is
//     routed to the bf16x3 kernel (every weight exact to 24 bits on its own scale); decided on the device:
//     route[c][tile] = 1.  Inside the guard the step is < 2^-23 * 2^(BI_GUARD_BINADES+1) of the median weight.
constexpr int BI_ROW_OUT = 8;
constexpr int BI_TILE_OUT = 32;
constexpr int BI_GUARD_BINADES = 4;
constexpr int BI_GAP_BINADES = 3;

// pass 1a, grid (8 * nbtile, nchan), 256 threads = 4 waves = 4 rows.  Outputs per row: scale (with the 1/16 of the
// voltage scaling folded in), wmax (inlier maximum, for pass 2), row_out[BI_ROW_OUT] (outlier inputs, -1 = none);
// route[nchan*nbtile] (+ route[nchan*nbtile] = "any tile routed"), zeroed by the caller.
__global__ __launch_bounds__(256) void beam_weights_rowstat_kernel(const float* __restrict__ w, float* __restrict__ scale,
                                                                   float* __restrict__ wmax, int* __restrict__ row_out,
                                                                   int* __restrict__ route, int* __restrict__ wsum,
                                                                   int nchan, int nbeam, int ninput, int nbtile) {
    // (reductions go through LDS memory and LDS atomics, not through ds_bpermute: see beam_integrate_kernel)
    __shared__ int hist[4][256];
    __shared__ int lanetot[4][64];
    __shared__ int red[4][4];                                // per wave: E (min), Emed (max), inlier maximum (max, float bits), bucket 0
    const int c = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;                  // beam index within the padded tiles
    const bool live = row < nbeam;
    for (int k = lane; k < 256; k += 64) hist[wave][k] = 0;
    if (lane == 0) { red[wave][0] = 1 << 30; red[wave][1] = 0; red[wave][2] = 0; }
    __syncthreads();
    const float* wrow = w + ((size_t)c * nbeam + (live ? row : 0)) * ninput * 2;
    if (live)
        for (int i = lane; i < ninput; i += 64) {
            const float2 v = *reinterpret_cast<const float2*>(wrow + 2 * i);
            atomicAdd(&hist[wave][(__float_as_uint(fmaxf(fabsf(v.x), fabsf(v.y))) >> 23) & 0xFF], 1);
        }
    __syncthreads();
    // E = the smallest exponent bucket with at most BI_ROW_OUT entries above it AND none of them within BI_GAP_BINADES
    // binades (outliers stand out; the top of a smooth distribution is not an outlier).  Lane l owns buckets 4l..4l+3.
    int cnt[4], tot = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { cnt[q] = hist[wave][4 * lane + q]; tot += cnt[q]; }
    lanetot[wave][lane] = tot;
    if (lane == 0) red[wave][3] = cnt[0];                    // entries with a zero exponent field (zeros, denormals)
    __syncthreads();
    int suf = 0;                                             // inclusive suffix sum over lanes >= lane
    for (int l = lane; l < 64; l++) suf += lanetot[wave][l];
    int all = suf;
    for (int l = 0; l < lane; l++) all += lanetot[wave][l];
    // ... and the bucket of the median non-zero entry: the largest bucket b >= 1 with at least half of them at or above it
    const int n_nz = all - red[wave][3];
    int above = suf - tot, Emed = 0;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        hist[wave][4 * lane + q] = above;                    // entries strictly above bucket 4*lane+q (own buckets: no race)
        above += cnt[q];                                     // now: entries at or above it
        if (2 * above >= n_nz && 4 * lane + q >= 1) Emed = max(Emed, 4 * lane + q);
    }
    __syncthreads();
    int E = 1 << 30;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        const int b = 4 * lane + q, ab = hist[wave][b];
        if (ab <= BI_ROW_OUT && ab == hist[wave][min(b + BI_GAP_BINADES, 255)]) E = b;
    }
    if (E < (1 << 30)) atomicMin(&red[wave][0], E);
    if (Emed > 0) atomicMax(&red[wave][1], Emed);
    __syncthreads();
    E = red[wave][0];
    Emed = red[wave][1];
    // second sweep: inlier maximum / sum of squares, outlier list (wave-level compaction)
    float m = 0.f;
    int nout = 0;
    int* ro = row_out + ((size_t)c * nbtile * 32 + row) * BI_ROW_OUT;
    for (int i0 = 0; i0 < ninput; i0 += 64) {
        const int i = i0 + lane;
        bool is_out = false;
        if (live && i < ninput) {
            const float2 v = *reinterpret_cast<const float2*>(wrow + 2 * i);
            const float a = fmaxf(fabsf(v.x), fabsf(v.y));
            if ((int)((__float_as_uint(a) >> 23) & 0xFF) <= E) m = fmaxf(m, a);
            else is_out = true;
        }
        const unsigned long long mask = __ballot(is_out);
        if (is_out) ro[nout + __popcll(mask & ((1ull << lane) - 1))] = i;     // (at most BI_ROW_OUT by the choice of E)
        nout += __popcll(mask);
    }
    if (m > 0.f) atomicMax(&red[wave][2], __float_as_int(m));   // (non-negative floats order like their bit patterns)
    __syncthreads();
    m = __int_as_float(red[wave][2]);
    if (lane >= nout && lane < BI_ROW_OUT) ro[lane] = -1;
    if (lane < 3) wsum[((size_t)(c * nbtile + (row >> 5)) * 3 + lane) * 32 + (row & 31)] = 0;   // pass 2 adds the digits up
    if (lane == 0) {
        scale[(size_t)c * nbtile * 32 + row] = m > 0.f ? m / (float)BI_QMAX / 16.f : 0.f;
        wmax[(size_t)c * nbtile * 32 + row] = m;
        if (n_nz > 0 && (int)((__float_as_uint(m) >> 23) & 0xFF) - Emed > BI_GUARD_BINADES) {
            route[c * nbtile + (row >> 5)] = 1;
            route[nchan * nbtile] = 1;
        }
    }
}

// pass 1b, grid (nbtile, nchan), 256 threads = (row tid/8, outlier slot tid%8): union of the tile's outlier inputs
// -> out_n[c][tile], out_idx[c][tile][BI_TILE_OUT], out_R[c][tile][BI_TILE_OUT][32 rows] (the fp32 weight of
// (row, input) where the row lists that input, else 0)
__global__ __launch_bounds__(256) void beam_weights_outlier_kernel(const float* __restrict__ w, const int* __restrict__ row_out,
                                                                   int* __restrict__ out_n, int* __restrict__ out_idx,
                                                                   float2* __restrict__ out_R, int* __restrict__ route,
                                                                   int nchan, int nbeam, int ninput, int nbtile) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ol_lds[];     // ninput bytes of marks (padded to 4) + list
    uint8_t* mark = ol_lds;
    int* list = reinterpret_cast<int*>(ol_lds + ((ninput + 63) & ~63));  // [BI_TILE_OUT] + count
    const int bt = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const int tile = c * nbtile + bt;
    for (int i = tid; i < ninput; i += 256) mark[i] = 0;
    float2* R = out_R + (size_t)tile * BI_TILE_OUT * 32;
    for (int k = tid; k < BI_TILE_OUT * 32; k += 256) R[k] = make_float2(0.f, 0.f);
    __syncthreads();
    const int r = tid >> 3, sl = tid & 7;
    const int idx = row_out[((size_t)tile * 32 + r) * BI_ROW_OUT + sl];
    if (idx >= 0) mark[idx] = 1;
    __syncthreads();
    if (tid < 64) {                                          // one wave compacts the marks into the sorted union
        int n = 0;
        for (int i0 = 0; i0 < ninput; i0 += 64) {
            const int i = i0 + tid;
            const bool on = i < ninput && mark[i];
            const unsigned long long mask = __ballot(on);
            const int pos = n + __popcll(mask & ((1ull << tid) - 1));
            if (on && pos < BI_TILE_OUT) list[pos] = i;
            n += __popcll(mask);
        }
        if (tid == 0) {
            list[BI_TILE_OUT] = n;
            if (n > BI_TILE_OUT) { route[tile] = 1; route[nchan * nbtile] = 1; n = 0; }   // too many: bf16x3 takes the tile
            out_n[tile] = n;
        }
    }
    __syncthreads();
    const int n = min(list[BI_TILE_OUT], BI_TILE_OUT);
    if (tid < BI_TILE_OUT) out_idx[(size_t)tile * BI_TILE_OUT + tid] = tid < n ? list[tid] : 0;
    if (idx >= 0 && list[BI_TILE_OUT] <= BI_TILE_OUT) {
        int k = 0;
        while (k < n && list[k] != idx) k++;
        const int b = bt * 32 + r;
        if (k < n && b < nbeam) R[k * 32 + r] = *reinterpret_cast<const float2*>(w + (((size_t)c * nbeam + b) * ninput + idx) * 2);
    }
}

// pass 2, grid (nchunk, nbtile, nchan), 256 threads = (beam tid/8, 4 inputs (tid%8)*4 of the chunk)
__global__ __launch_bounds__(256) void beam_weights_prep_i8_kernel(const float* __restrict__ w, uint8_t* __restrict__ wq,
                                                                   const float* __restrict__ wmax, int nchan, int nbeam,
                                                                   int ninput, int nchunk, int nbtile,
                                                                   const int* __restrict__ route, int* __restrict__ wsum) {
    const int ch = blockIdx.x, bt = blockIdx.y, c = blockIdx.z;      // ch: 32-input K step (nchunk = number of steps, padded to whole chunks)
    if (route[c * nbtile + bt]) return;                              // the bf16x3 kernel takes this (channel, beam tile)
    const int beam = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const int b = bt * 32 + beam;
    const float* wrow = w + ((size_t)c * nbeam + (b < nbeam ? b : 0)) * ninput * 2;
    const float m = wmax[(size_t)(c * nbtile + bt) * 32 + beam];
    const float inv = m > 0.f ? (float)BI_QMAX / m : 0.f;
    auto digits = [](int q, int (&d)[3]) {      // balanced base 255, most significant first
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            int r = q % 255;                     // C remainder: sign of q
            if (r > 127) r -= 255;
            if (r < -127) r += 255;
            d[k] = r;
            q = (q - r) / 255;
        }
    };
    const int h = l8 >> 2, byte0 = (l8 & 3) * 4, lane = 32 * h + beam;
    uint32_t pk[3][2] = {};                      // [digit][Wr | Wi] 4 packed int8
    int dsum[3] = {0, 0, 0};                     // this thread's share of sum_i Wi digit (see beamform_i8x3_kernel)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = ch * 32 + l8 * 4 + j;
        int qr = 0, qi = 0;
        if (b < nbeam && i < ninput) {
            const float2 v = *reinterpret_cast<const float2*>(wrow + 2 * i);
            if (fmaxf(fabsf(v.x), fabsf(v.y)) <= m) {     // (an outlier keeps zero digits: it is added in fp32 in the epilogue)
                // (the row maximum itself may round to QMAX + 1, whose leading digit would be 128: clamp)
                qr = max(-BI_QMAX, min(BI_QMAX, (int)rintf(v.x * inv)));
                qi = max(-BI_QMAX, min(BI_QMAX, (int)rintf(v.y * inv)));
            }
        }
        int dr[3], di[3];
        digits(qr, dr);
        digits(qi, di);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            pk[t][0] |= (uint32_t)(dr[t] & 0xFF) << (8 * j);
            pk[t][1] |= (uint32_t)(di[t] & 0xFF) << (8 * j);
            dsum[t] += di[t];
        }
    }
    uint8_t* base = wq + (((size_t)c * nbtile + bt) * nchunk + ch) * BI_WSTEP + lane * 16 + byte0;
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int k = 0; k < 2; k++) *reinterpret_cast<uint32_t*>(base + (t * 2 + k) * 1024) = pk[t][k];
    // per (row, digit): sum of the Wi digits over all inputs (integer atomics: order-independent)
#pragma unroll
    for (int t = 0; t < 3; t++) {
        int v = dsum[t];                         // 8-lane sum by DPP row shifts: the total lands in lane l8 == 0
        v += __builtin_amdgcn_update_dpp(0, v, 0x104 /* row_shl:4 */, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x102, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);
        if (l8 == 0 && v != 0) atomicAdd(&wsum[((size_t)(c * nbtile + bt) * 3 + t) * 32 + beam], v);
    }
}

template <bool DESC = false, bool TAB = false>
__global__ __launch_bounds__(256, 3) void beamform_i8x3_kernel(const uint8_t* __restrict__ in,
                                                               const uint8_t* __restrict__ wq,
                                                               const float* __restrict__ scale, const int* __restrict__ wsum,
                                                               float* __restrict__ out, int ntime, int nchan,
                                                               int ninput, int nbeam, int nchunk, int nbtile,
                                                               const int* __restrict__ route, const int* __restrict__ out_n,
                                                               const int* __restrict__ out_idx, const float2* __restrict__ out_R,
                                                               unsigned long long* __restrict__ stamps,
                                                               float* __restrict__ pow_out, int ntime_sum,
                                                               const uint8_t* __restrict__ in1, int split,
                                                               const GulpDesc* __restrict__ gd, const uint8_t* __restrict__ zeros = nullptr) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[BI_RING * BI_STAGE];
    const unsigned long long r0 = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;   // diagnostic (XENG_BEAM_STAMPS=1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nttile = (ntime + BI_NT - 1) / BI_NT, per_c = nbtile * nttile;
    int c, rem;
    if (DESC && (nchan & 15) == 0) {      // packet slabs: channels 2k, 2k+1 share every cache line of a packet -- both on one XCD, back to back
        const int b = blockIdx.x, slot = b >> 3, q = slot / per_c;
        c = 16 * (q >> 1) + 2 * (b & 7) + (q & 1); rem = slot % per_c;
    } else if ((nchan & 7) == 0) { const int b = blockIdx.x, slot = b >> 3; c = (b & 7) + 8 * (slot / per_c); rem = slot % per_c; }
    else { c = blockIdx.x / per_c; rem = blockIdx.x % per_c; }
    const int bt = rem / nttile, t0 = (rem % nttile) * BI_NT;
    const int routed = route[c * nbtile + bt];         // this (channel, beam tile) runs on the bf16x3 kernel: checked below,
    const int n_outl = out_n[c * nbtile + bt];         // behind the first DMA; outlier inputs of this tile (epilogue)
    const int h = lane >> 5, j = lane & 31;
    const uint8_t* wsrc = wq + (((size_t)c * nbtile + bt) * nchunk) * BI_WCHUNK + lane * 16;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    // X pieces 2*wave, 2*wave+1 of a chunk = samples 32*wave .. +31, 64 B each (16 rows per 1 KiB piece); lane: row
    // lane/4 of the piece, 16-byte position lane&3 holding source piece (lane&3) ^ ((row>>2)&1): the swizzle (on the
    // source side; the LDS side of the DMA is lane-linear) makes the 16-byte operand reads conflict-free at 64-byte pitch
    const GulpAddr<DESC, TAB> ga(in, in1, split, gd, nchan, ninput);
    const uint8_t* xsrc[BI_XSLOTS];
    uint32_t xbstr[BI_XSLOTS];
    // TAB: piece n's part is read through its index (xsrc[n]: the sample's index row; a chunk is one 64-input block, i.e. one entry);
    // the entry of the chunk that the NEXT issue() brings is fetched right after this one's -- it arrives under the MFMAs
    bool xidx[BI_XSLOTS];
    int xtt[BI_XSLOTS];
    uint32_t e_nx[BI_XSLOTS] = {};
#pragma unroll
    for (int n = 0; n < BI_XSLOTS; n++) {
        int xt = t0 + wave * 32 + n * 16 + (lane >> 2);
        if (xt >= ntime) xt = ntime - 1;
        xsrc[n] = ga.row(xt, c, &xbstr[n]);
        xtt[n] = xt;
        xidx[n] = TAB && __builtin_amdgcn_readfirstlane((int)ga.indexed(xt)) != 0;
    }
    // (from inline asm, like the LDS-DMA: a load the compiler knows about gets `s_waitcnt vmcnt(0)` in front of the next one -- it
    // cannot count the asm LDS-DMA pieces in between -- and that wait would sit between a chunk's pieces and its MFMAs.  The register
    // is tied in and out, so the value lives in one register from the load to the kernel's own vmcnt(0) at the top of the next chunk,
    // which is what makes it valid; nothing reads it before.)
    auto load_entries = [&](int ch) {
        if (!TAB) return;
        const int cs = ch < nchunk ? ch : nchunk - 1;
#pragma unroll
        for (int n = 0; n < BI_XSLOTS; n++)
            if (xidx[n]) {
                const uint32_t* ep = reinterpret_cast<const uint32_t*>(xsrc[n]) + ((cs * BI_KC) >> 6);
                asm volatile("global_load_dword %0, %1, off" : "+v"(e_nx[n]) : "v"(ep) : "memory");
            }
    };
    const int xpiece = ((lane & 3) ^ ((lane >> 4) & 1)) * 16;
    // every stage costs exactly BI_WSLOTS + BI_XSLOTS pieces per wave on the vmcnt counter (chunks past the end
    // re-read the last one; digit slots past the last piece re-copy an earlier piece onto itself)
    auto issue = [&](int ch, int buf) {
        const int cs = ch < nchunk ? ch : nchunk - 1;
        const uint32_t l = lds0 + buf * BI_STAGE;
#pragma unroll
        for (int n = 0; n < BI_WSLOTS; n++) {
            const int pc = (wave + 4 * n) % (BI_KS * 6);
            lds_dma16(wsrc + (size_t)cs * BI_WCHUNK + pc * 1024, l + pc * 1024);
        }
#pragma unroll
        for (int n = 0; n < BI_XSLOTS; n++) {
            int i = cs * BI_KC + xpiece;
            if (i + 16 > ninput) i = 0;                // columns past the end meet zero digits
            lds_dma16(xidx[n] ? ga.piece_from(e_nx[n], xtt[n], c, i, zeros) : xsrc[n] + gulp_col<DESC>(i, xbstr[n]), l + BI_WCHUNK + (wave * BI_XSLOTS + n) * 1024);
        }
    };
    static_assert(!TAB || (BI_KC == 64 && BI_RING == 2), "TAB: one chunk = one 64-input block = one index entry; the entry fetched behind a chunk's pieces is "
                                                          "covered by the vmcnt(0) at the top of the next chunk (ring of two stages)");
#pragma unroll
    for (int k = 0; k < BI_RING - 1; k++) {          // first: the DMA of the first chunk(s) is the critical path
        if (TAB) { load_entries(k); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        issue(k, k);
    }
    load_entries(BI_RING - 1);
    if (routed) {                                                // (the route flag's load has travelled beside the DMA)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may be in flight when the wave ends
        return;
    }
    // the 16 row scales this lane needs in the epilogue: fetched now, off the critical path
    float sc[16];
#pragma unroll
    for (int g = 0; g < 16; g++) sc[g] = scale[(size_t)(c * nbtile + bt) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h];
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    typedef int v16i_ __attribute__((ext_vector_type(16)));
    v16i_ acc_re[3], acc_im[3];
    // Tre = sum(dr*xr - di*xi) without a negated digit plane: the MFMA gets ~xi = -xi - 1 (one XOR on the masked
    // operand; -xi itself would not fit for xi = -8), sum(di * 16*(~xi)) = -sum(di * 16*xi) - 16 * sum(di), and the
    // accumulator starts at +16 * sum(di) (exact integers; wsum from the prep pass)
#pragma unroll
    for (int t = 0; t < 3; t++) {
        acc_im[t] = (v16i_)(0);
#pragma unroll
        for (int g = 0; g < 16; g++)
            acc_re[t][g] = 16 * wsum[((size_t)(c * nbtile + bt) * 3 + t) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h];
    }
    int buf = 0, nbuf = BI_RING - 1;
    unsigned long long r1 = 0;
    for (int ch = 0; ch < nchunk; ch++) {
        // this wave's pieces of chunk ch have landed (the BI_RING-2 younger chunks stay in flight)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((BI_WSLOTS + BI_XSLOTS) * (BI_RING - 2)) : "memory");
        __builtin_amdgcn_s_barrier();                  // ... for all waves; and everybody is done reading chunk ch-1,
        issue(ch + BI_RING - 1, nbuf);                 // whose buffer the DMA of chunk ch+BI_RING-1 now overwrites
        load_entries(ch + BI_RING);
        if (stamps && ch == 0) r1 = __builtin_amdgcn_s_memrealtime();
        const v4i_ M = (v4i_)(0xF0F0F0F0);
#pragma unroll
        for (int ks = 0; ks < BI_KS; ks++) {
            const uint8_t* lw = lds + buf * BI_STAGE + ks * BI_WSTEP + lane * 16;
            const v4i_ xraw = *reinterpret_cast<const v4i_*>(lds + buf * BI_STAGE + BI_WCHUNK + (wave * 32 + j) * BI_KC +
                                                              (((2 * ks + h) ^ ((j >> 2) & 1)) * 16));
            const v4i_ Xr = xraw & M, Xi = (xraw << 4) & M;    // 16*re, 16*im (hi nibble real, lo nibble imag; beamformer_test.py:69-73)
            const v4i_ nXi = Xi ^ M;                           // 16 * ~im = -16*im - 16
#pragma unroll
            for (int t = 2; t >= 0; t--) {
                const v4i_ Wr = *reinterpret_cast<const v4i_*>(lw + (t * 2 + 0) * 1024);
                const v4i_ Wi = *reinterpret_cast<const v4i_*>(lw + (t * 2 + 1) * 1024);
                acc_re[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Wr, Xr, acc_re[t], 0, 0, 0);
                acc_im[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Wr, Xi, acc_im[t], 0, 0, 0);
                acc_re[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Wi, nXi, acc_re[t], 0, 0, 0);
                acc_im[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Wi, Xr, acc_im[t], 0, 0, 0);
            }
        }
        buf = buf + 1 == BI_RING ? 0 : buf + 1;
        nbuf = nbuf + 1 == BI_RING ? 0 : nbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may be in flight when the wave ends
    unsigned long long r2 = 0;
    if (stamps) {
        asm volatile("" :: "v"(acc_re[0][15]), "v"(acc_im[2][15]));
        r2 = __builtin_amdgcn_s_memrealtime();
    }
    // C/D map: col (sample) = lane&31, row (beam) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int t = t0 + wave * 32 + j;
    float re[16], im[16];
#pragma unroll
    for (int g = 0; g < 16; g++) {
        const float s = sc[g];
        re[g] = s * (((float)acc_re[0][g] * 65025.f + (float)acc_re[1][g] * 255.f) + (float)acc_re[2][g]);
        im[g] = s * (((float)acc_im[0][g] * 65025.f + (float)acc_im[1][g] * 255.f) + (float)acc_im[2][g]);
    }
    if (n_outl > 0) {
        // the tile's outlier weights (zero digits above) times the voltages, in fp32
        uint32_t cbstr;
        const uint8_t* xcol = ga.row(t < ntime ? t : ntime - 1, c, &cbstr);
        const int tile = c * nbtile + bt;
        for (int k = 0; k < n_outl; k++) {
            const int oi = out_idx[(size_t)tile * BI_TILE_OUT + k];
            const int xb = (TAB && ga.indexed(t < ntime ? t : ntime - 1)) ? ga.piece(xcol, t < ntime ? t : ntime - 1, c, oi, zeros)[0] : xcol[gulp_col<DESC>(oi, cbstr)];
            const float xr = (float)(int)__builtin_amdgcn_sbfe(xb, 4, 4), xi = (float)(int)__builtin_amdgcn_sbfe(xb, 0, 4);
            const float2* Rk = out_R + ((size_t)tile * BI_TILE_OUT + k) * 32;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const float2 R = Rk[(g & 3) + 8 * (g >> 2) + 4 * h];
                re[g] += R.x * xr - R.y * xi;
                im[g] += R.x * xi + R.y * xr;
            }
        }
    }
    if (pow_out) {
        // Integrated-power mode (bfBeamformInitialize ntime_blocks > 0, "experimental" in the reference:
        // beamform_block.py:108-110): the power sums of bfBeamformIntegrate (beamform_sum_beams_block.py:243-246,
        // cublas_beamform.cu:46-79) are formed here and the voltage beams never go to memory.  Per sample and beam pair
        // (X = beam 2p, Y = 2p+1, both rows in one lane) the four products go to LDS (the staging ring is free now);
        // then every time block of ntime_sum samples is summed in sample order.  A block that lies inside this
        // work-group's 128 samples is stored; one that straddles two work-groups is added atomically to the
        // zero-initialised output -- two addends only (the caller requires ntime_sum <= 128), so the sum does not
        // depend on their order.
        __builtin_amdgcn_s_barrier();                  // all waves are past their last LDS read of the ring
        float4* pl = reinterpret_cast<float4*>(lds);   // [128 samples][16 pairs]
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            const int pr = ((g & 3) >> 1) + 4 * (g >> 2) + 2 * h;
            const float xr = re[g], xi = im[g], yr = re[g + 1], yi = im[g + 1];
            pl[(wave * 32 + j) * 16 + pr] = make_float4(xr * xr + xi * xi, yr * yr + yi * yi, xr * yr + xi * yi, xi * yr - xr * yi);
        }
        __syncthreads();
        const int nblk = ntime / ntime_sum, tend = min(t0 + BI_NT, ntime);
        const int pc = tid & 63, pr = pc >> 2, comp = pc & 3;
        const int pair = bt * 16 + pr;
        const float* pf = reinterpret_cast<const float*>(lds);
        if (2 * pair + 1 < nbeam)
            for (int b = t0 / ntime_sum + (tid >> 6); b < nblk && b * ntime_sum < tend; b += 4) {
                const int s0 = max(b * ntime_sum, t0), s1 = min((b + 1) * ntime_sum, tend);
                float acc = 0.f;
                for (int sm = s0; sm < s1; sm++) acc += pf[((sm - t0) * 16 + pr) * 4 + comp];
                float* dst = pow_out + (((size_t)pair * nblk + b) * nchan + c) * 4 + comp;
                if (s0 == b * ntime_sum && s1 == (b + 1) * ntime_sum) *dst = acc;
                else atomicAdd(dst, acc);
            }
    } else if (t < ntime) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int b = bt * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (b < nbeam) *reinterpret_cast<float2*>(out + (((size_t)c * nbeam + b) * ntime + t) * 2) = make_float2(re[g], im[g]);
        }
    }
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// beam power sums (beamformer_sum_test.py:64-77, cublas_beamform.cu:46-79).
// in cf32[nchan][nbeam][ntime] -> out f32[npair][ntime/ntime_sum][nchan][4]; one wave per
// (channel, beam pair): lanes stride the time blocks, 8-lane groups sweep one block's samples
// so global reads are contiguous, then reduce across the group with DPP row shifts.
// (XENG_EXTERNAL_BEAM_INTEGRATE: the hazard lab, profiles/hazard/, compiles this translation unit with the round-2 form
// of the kernel in its place; defining the macro without supplying a kernel does not build.)
#ifndef XENG_EXTERNAL_BEAM_INTEGRATE
__global__ __launch_bounds__(256) void beam_integrate_kernel(const float2* __restrict__ in, float4* __restrict__ out,
                                                             int nchan, int nbeam, int ntime, int ntime_sum,
                                                             int pair0, int npair_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave, bp = blockIdx.y;  // bp: output pair index
    if (c >= nchan) return;
    const int nblk = ntime / ntime_sum;
    const float2* x = in + ((size_t)c * nbeam + 2 * (pair0 + bp)) * ntime;
    const float2* y = x + ntime;
    const int sub = lane & 7, grp = lane >> 3;  // 8 groups of 8 lanes
    // gridDim.z slices the time blocks (8 per slice when the grid is large enough): the kernel is a short
    // latency chain, so more resident waves beat longer loops
    for (int tb = blockIdx.z * 8 + grp; tb < nblk; tb += 8 * gridDim.z) {
        float xx = 0.f, yy = 0.f, xyr = 0.f, xyi = 0.f;
        for (int t = sub; t < ntime_sum; t += 8) {
            const float2 a = x[(size_t)tb * ntime_sum + t], b = y[(size_t)tb * ntime_sum + t];
            xx += a.x * a.x + a.y * a.y;
            yy += b.x * b.x + b.y * b.y;
            xyr += a.x * b.x + a.y * b.y;
            xyi += a.y * b.x - a.x * b.y;
        }
        // 8-lane sums by DPP row shifts (lane i += lane i+4, i+2, i+1: the total lands in lane sub == 0; same association
        // as an xor butterfly).  NOT ds_bpermute (__shfl_xor): see DESIGN.md 4.10 and profiles/hazard/ -- no kernel of
        // this object may contain ds_bpermute / ds_swizzle (tests/test_abi.py checks the ISA).
#define XENG_DPP_ADD(v, ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
        XENG_DPP_ADD(xx, 0x104); XENG_DPP_ADD(yy, 0x104); XENG_DPP_ADD(xyr, 0x104); XENG_DPP_ADD(xyi, 0x104);
        XENG_DPP_ADD(xx, 0x102); XENG_DPP_ADD(yy, 0x102); XENG_DPP_ADD(xyr, 0x102); XENG_DPP_ADD(xyi, 0x102);
        XENG_DPP_ADD(xx, 0x101); XENG_DPP_ADD(yy, 0x101); XENG_DPP_ADD(xyr, 0x101); XENG_DPP_ADD(xyi, 0x101);
#undef XENG_DPP_ADD
        if (sub == 0) out[((size_t)bp * nblk + tb) * nchan + c] = make_float4(xx, yy, xyr, xyi);
    }
    (void)npair_out;
}
#endif

}  // namespace xeng
