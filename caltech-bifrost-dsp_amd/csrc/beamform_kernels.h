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

namespace xeng {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BF_KC = 64;                 // inputs per LDS chunk
constexpr int BF_WS = BF_KC * 8 + 16;     // W row stride (bytes): conflict-free ds_read_b128
constexpr int BF_XS = BF_KC + 4;          // X row stride (bytes): conflict-free ds_read_b32
constexpr int BF_NT = 128;                // samples per work-group

__global__ __launch_bounds__(256) void beamform_f32_kernel(const uint8_t* __restrict__ in,
                                                           const float* __restrict__ w,
                                                           float* __restrict__ out, int ntime, int nchan,
                                                           int ninput, int nbeam) {
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
                const uint8_t* src = in + ((size_t)t * nchan + c) * ninput + i;
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
constexpr int BF3_ROW = 144;                         // 64 bf16 + 16 B pad
constexpr int BF3_WCHUNK = 3 * 2 * 32 * BF3_ROW;      // 27648 B of weights per 64-input chunk
constexpr int BF3_NT = 128;                           // samples per work-group: 4 waves x 32, one per SIMD
constexpr int BF3_XCHUNK = BF3_NT * BF_KC;            // 12288 B of packed voltages per chunk
constexpr int BF3_STAGE = BF3_WCHUNK + BF3_XCHUNK;

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;   // weights are finite
}

// grid (nchunk, nbtile, nchan), 256 threads: thread = (beam tid/8, 8 inputs (tid%8)*8)
__global__ __launch_bounds__(256) void beam_weights_prep_kernel(const float* __restrict__ w, uint8_t* __restrict__ wp,
                                                                int nchan, int nbeam, int ninput, int nchunk, int nbtile) {
    const int ch = blockIdx.x, bt = blockIdx.y, c = blockIdx.z;
    const int beam = threadIdx.x >> 3, k0 = (threadIdx.x & 7) * 8;
    const int b = bt * 32 + beam;
    uint32_t t[3][2][4];   // [term][re|im][4 dwords = 8 bf16]
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = ch * BF_KC + k0 + j;
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
#pragma unroll
    for (int term = 0; term < 3; term++)
#pragma unroll
        for (int comp = 0; comp < 2; comp++)
            *reinterpret_cast<uint4*>(base + ((term * 2 + comp) * 32 + beam) * BF3_ROW + k0 * 2) =
                make_uint4(t[term][comp][0], t[term][comp][1], t[term][comp][2], t[term][comp][3]);
    // the 16-byte pad of each row is never read
}

__device__ __forceinline__ v8bf as_v8bf(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = {a, b, c, d};
    return __builtin_bit_cast(v8bf, v);
}

// grid nchan * nbtile * ceil(ntime/128) (1-D), 256 threads; wave w owns samples t0 + 32w .. +31.
// One LDS stage (36 KB) per work-group: four work-groups per CU interleave, so one group's LDS-DMA
// latency and barriers hide under the MFMAs of the other three (4 waves per SIMD, all SIMDs equal).
__global__ __launch_bounds__(256, 4) void beamform_bf16x3_kernel(const uint8_t* __restrict__ in,
                                                                 const uint8_t* __restrict__ wp,
                                                                 float* __restrict__ out, int ntime, int nchan,
                                                                 int ninput, int nbeam, int nchunk, int nbtile) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[BF3_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid of nchan*nbtile*nttile blocks.  Blocks b and b+8 share an XCD: give each XCD whole channels
    // so the work-groups of a channel read its 304 KB of split weights through one L2 (speed only).
    const int nttile = (ntime + BF3_NT - 1) / BF3_NT, per_c = nbtile * nttile;
    int c, rem;
    if ((nchan & 7) == 0) { const int b = blockIdx.x, slot = b >> 3; c = (b & 7) + 8 * (slot / per_c); rem = slot % per_c; }
    else { c = blockIdx.x / per_c; rem = blockIdx.x % per_c; }
    const int bt = rem / nttile, t0 = (rem % nttile) * BF3_NT;
    const int h = lane >> 5, j = lane & 31;
    const uint8_t* wsrc = wp + (((size_t)c * nbtile + bt) * nchunk) * BF3_WCHUNK + lane * 16;
    // X piece n (1 KiB) of a chunk = samples 16n..16n+15, 64 B each; this lane: row 16n + lane/4, bytes (lane%4)*16
    const size_t row_stride = (size_t)nchan * ninput;
    constexpr int NWP = BF3_WCHUNK / 1024;            // 27 weight pieces per chunk
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    auto issue = [&](int ch, int buf) {
        const uint32_t l = lds0 + buf * BF3_STAGE;
        for (int n = wave; n < NWP; n += BF3_NT / 32)
            lds_dma16(wsrc + (size_t)ch * BF3_WCHUNK + n * 1024, l + n * 1024);
        for (int n = wave; n < BF3_NT / 16; n += BF3_NT / 32) {
            int t = t0 + n * 16 + (lane >> 2);
            int i = ch * BF_KC + (lane & 3) * 16;
            if (t >= ntime) t = ntime - 1;             // rows past the end: any valid row (never stored)
            if (i + 16 > ninput) i = 0;                // columns past the end meet zero weights
            const uint8_t* g = in + (size_t)t * row_stride + (size_t)c * ninput + i;
            lds_dma16(g, l + BF3_WCHUNK + n * 1024);
        }
    };
    v16f acc_r = (v16f)(0.f), acc_i = (v16f)(0.f);
    for (int ch = 0; ch < nchunk; ch++) {
        const int buf = 0;
        if (ch > 0) __builtin_amdgcn_s_barrier();      // everybody is done reading chunk ch-1
        issue(ch, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // chunk ch landed for all waves
        const uint8_t* lw = lds + buf * BF3_STAGE + j * BF3_ROW + h * 16;
        const uint8_t* lx = lds + buf * BF3_STAGE + BF3_WCHUNK + (wave * 32 + j) * BF_KC + h * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) {                  // 16 inputs per step: this lane-half takes 8 of them
            const uint2 xb = *reinterpret_cast<const uint2*>(lx + s * 16);
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
#pragma unroll
            for (int term = 2; term >= 0; term--) {    // smallest term first
                const uint4 a = *reinterpret_cast<const uint4*>(lw + ((term * 2 + 0) * 32) * BF3_ROW + s * 32);
                const uint4 b = *reinterpret_cast<const uint4*>(lw + ((term * 2 + 1) * 32) * BF3_ROW + s * 32);
                const v8bf Wr = as_v8bf(a.x, a.y, a.z, a.w), Wi = as_v8bf(b.x, b.y, b.z, b.w);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr, Xr, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr, Xi, acc_i, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wi, nXi, acc_r, 0, 0, 0);
                acc_i = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wi, Xr, acc_i, 0, 0, 0);
            }
        }
    }
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

// beam power sums (beamformer_sum_test.py:64-77, cublas_beamform.cu:46-79).
// in cf32[nchan][nbeam][ntime] -> out f32[npair][ntime/ntime_sum][nchan][4]; one wave per
// (channel, beam pair): lanes stride the time blocks, 8-lane groups sweep one block's samples
// so global reads are contiguous, then reduce across the group with DPP/shuffles.
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
    for (int tb = grp; tb < nblk; tb += 8) {
        float xx = 0.f, yy = 0.f, xyr = 0.f, xyi = 0.f;
        for (int t = sub; t < ntime_sum; t += 8) {
            const float2 a = x[(size_t)tb * ntime_sum + t], b = y[(size_t)tb * ntime_sum + t];
            xx += a.x * a.x + a.y * a.y;
            yy += b.x * b.x + b.y * b.y;
            xyr += a.x * b.x + a.y * b.y;
            xyi += a.y * b.x - a.x * b.y;
        }
#pragma unroll
        for (int o = 4; o >= 1; o >>= 1) {
            xx += __shfl_xor(xx, o);
            yy += __shfl_xor(yy, o);
            xyr += __shfl_xor(xyr, o);
            xyi += __shfl_xor(xyi, o);
        }
        if (sub == 0) out[((size_t)bp * nblk + tb) * nchan + c] = make_float4(xx, yy, xyr, xyi);
    }
    (void)npair_out;
}

}  // namespace xeng
