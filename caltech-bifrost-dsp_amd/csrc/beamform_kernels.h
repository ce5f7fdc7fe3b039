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
