// Prototype (diagnostic, not part of the product; results are NOT checked) of the three-product beamformer sketched in
// what would the config-4 gulp cost with 9 instead of 12 int8 MFMAs per 32-input K step?
//   * persistent grid, one 8-wave work-group per CU, three (channel, 128-sample) tiles each, XCD-aware (a channel's tiles on one XCD)
//   * v_mfma_i32_16x16x64_i8; wave = 16 beams (one half) x 32 samples; 9 planes (3 forms x 3 digits) x 2 sample blocks = 72 accumulator registers
//   * per 64-input chunk: 18 KiB of digit planes + 8 KiB of packed voltages by LDS-DMA into a 3-stage ring, two chunks ahead,
//     one barrier per chunk, continuous across tiles; voltage forms 16xr, 16xi, 8(xr+xi) by mask / shift / v_lerp_u8
//   * epilogue per tile: fp32 recombination (shape only) and float2 stores of the 32 x 128 outputs
// Build: hipcc -O3 --offload-arch=gfx950 beam3_proto.hip -o beam3_proto ;  run: ./beam3_proto
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NT = 960, NC = 96, NI = 704, NB = 32;
constexpr int KC = 64, NCHUNK = NI / KC, NPLANE = 9, WCH = NPLANE * 2 * 1024, NTT = 128, XCH = NTT * KC, STAGE = WCH + XCH, RING = 3;
constexpr int TILES_PER_C = (NT + NTT - 1) / NTT, ITEMS = 3, NSTEP = ITEMS * NCHUNK, NSLOT = 4, NSTORE = 8;

__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <int NPROD>      // 9: three-product form; 12: four products per digit (reference point in the same structure)
__global__ __launch_bounds__(512, 1) void beam3_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ wq,
                                                       float* __restrict__ out, const float* __restrict__ scale) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sg = wave & 3, bh = wave >> 2;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;          // 32 work-groups per XCD, 96 tiles per XCD
    const size_t row_stride = (size_t)NC * NI;
    auto tile_of = [&](int k, int& c, int& t0) {
        const int u = slot * ITEMS + k;
        c = xcd + 8 * (u / TILES_PER_C);
        t0 = (u % TILES_PER_C) * NTT;
    };
    // one chunk step: every wave issues exactly NSLOT pieces (pieces 0..17 digits, 18..25 voltages, the rest duplicates)
    auto issue = [&](int g) {
        const int gg = g < NSTEP ? g : NSTEP - 1;
        int c, t0;
        tile_of(gg / NCHUNK, c, t0);
        const int ch = gg % NCHUNK;
        const uint32_t l = lds0 + (uint32_t)(g % RING) * STAGE;
#pragma unroll
        for (int n = 0; n < NSLOT; n++) {
            int p = wave + 8 * n;
            if (p >= 26) p -= 8;
            const int q = p - 18;
            int t = t0 + (q < 0 ? 0 : q) * 16 + (lane >> 2);
            if (t >= NT) t = NT - 1;
            const uint8_t* src = p < 18 ? wq + (((size_t)c * NCHUNK + ch) * 18 + p) * 1024 + lane * 16
                                        : in + (size_t)t * row_stride + (size_t)c * NI + ch * KC + (lane & 3) * 16;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(p < 18 ? l + (uint32_t)p * 1024u : l + (uint32_t)WCH + (uint32_t)q * 1024u);
            lds_dma16(src, dst);
        }
    };
    v4i acc[NPROD == 9 ? 9 : 12][2];
#pragma unroll
    for (int p = 0; p < (NPROD == 9 ? 9 : 12); p++) { acc[p][0] = (v4i)(0); acc[p][1] = (v4i)(0); }
    issue(0);
    issue(1);
    bool stored = false;
    for (int g = 0; g < NSTEP; g++) {
        if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT + NSTORE) : "memory");     // (the tile stores of the last step are younger than the DMA waited for)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory");
        stored = false;
        __builtin_amdgcn_s_barrier();
        issue(g + 2);
        const uint8_t* st = lds + (g % RING) * STAGE;
        const v4i M = (v4i)(0xF0F0F0F0), M8 = (v4i)(0x80808080);
        v4i xr[2], xi[2], xs[2];
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            const v4i x = *reinterpret_cast<const v4i*>(st + WCH + (sg * 32 + nb * 16 + (lane & 15)) * KC + (lane >> 4) * 16);
            const v4i t = x << 4;
            xr[nb] = x & M;
            xi[nb] = t & M;
            if (NPROD == 9) {
                const v4i s = (x ^ t) & M8;
#pragma unroll
                for (int k = 0; k < 4; k++) xs[nb][k] = (int)__builtin_amdgcn_lerp((uint32_t)xr[nb][k], (uint32_t)xi[nb][k], 0u) ^ s[k];
            } else {
                xs[nb] = xi[nb] ^ M;       // the ~xi operand of the shipped kernel
            }
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (NPROD == 9) {
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const v4i w = *reinterpret_cast<const v4i*>(st + ((f * 3 + d) * 2 + bh) * 1024 + lane * 16);
                    const int p = f * 3 + d;
#pragma unroll
                    for (int nb = 0; nb < 2; nb++) {
                        const v4i& b = f == 0 ? xs[nb] : f == 1 ? xr[nb] : xi[nb];
                        acc[p][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, b, acc[p][nb], 0, 0, 0);
                    }
                }
            } else {
                // four products per digit with two digit planes (wr, wi) -- the planes 6..8 of a stage are simply not read
                const v4i wr = *reinterpret_cast<const v4i*>(st + ((0 * 3 + d) * 2 + bh) * 1024 + lane * 16);
                const v4i wi = *reinterpret_cast<const v4i*>(st + ((1 * 3 + d) * 2 + bh) * 1024 + lane * 16);
#pragma unroll
                for (int nb = 0; nb < 2; nb++) {
                    acc[d][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wr, xr[nb], acc[d][nb], 0, 0, 0);
                    acc[3 + d][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wr, xi[nb], acc[3 + d][nb], 0, 0, 0);
                    acc[d][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wi, xs[nb], acc[d][nb], 0, 0, 0);
                    acc[3 + d][nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wi, xr[nb], acc[3 + d][nb], 0, 0, 0);
                }
            }
        }
        if (g % NCHUNK == NCHUNK - 1) {
            int c, t0;
            tile_of(g / NCHUNK, c, t0);
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                int t = t0 + sg * 32 + nb * 16 + (lane & 15);
                if (t >= NT) t = NT - 1;            // (prototype: every lane stores, so that the counted wait below holds for every wave)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int beam = bh * 16 + 4 * (lane >> 4) + r;
                    const float s = scale[c * NB + beam];
                    float k[3];
#pragma unroll
                    for (int f = 0; f < 3; f++) {
                        const int p0 = NPROD == 9 ? f * 3 : (f < 2 ? f * 3 : 0);
                        k[f] = ((float)acc[p0][nb][r] * 65025.f + (float)acc[p0 + 1][nb][r] * 255.f) + (float)acc[p0 + 2][nb][r];
                    }
                    float2 v = NPROD == 9 ? make_float2(s * (k[0] - k[2]), s * (k[0] + k[1])) : make_float2(s * k[0], s * k[1]);
                    *reinterpret_cast<float2*>(out + (((size_t)c * NB + beam) * NT + t) * 2) = v;
                }
            }
            stored = true;
#pragma unroll
            for (int p = 0; p < (NPROD == 9 ? 9 : 12); p++) { acc[p][0] = (v4i)(0); acc[p][1] = (v4i)(0); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NPROD>
static float run(const uint8_t* din, const uint8_t* dw, float* dout, const float* dsc, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(beam3_kernel<NPROD>), dim3(256), dim3(512), 0, 0, din, dw, dout, dsc);
    hipDeviceSynchronize();
    float best = 1e9f, tot = 0;
    for (int i = 0; i < reps; i++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(beam3_kernel<NPROD>), dim3(256), dim3(512), 0, 0, din, dw, dout, dsc);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best; tot += ms;
    }
    printf("%2d products per digit-triple: %.1f us per gulp (HIP events, mean of %d; best %.1f)\n", NPROD, tot / reps * 1e3, reps, best * 1e3);
    return tot / reps;
}

int main() {
    if (hipSetDevice(0) != hipSuccess) { printf("no device\n"); return 1; }
    const size_t nin = (size_t)NT * NC * NI, nw = (size_t)NC * NCHUNK * 18 * 1024, nout = (size_t)NC * NB * NT * 2;
    std::vector<uint8_t> hin(nin), hw(nw);
    srand(1);
    for (auto& v : hin) v = (uint8_t)rand();
    for (auto& v : hw) v = (uint8_t)rand();
    std::vector<float> hs(NC * NB, 1e-6f);
    uint8_t *din, *dw; float *dout, *dsc;
    hipMalloc(&din, nin); hipMalloc(&dw, nw); hipMalloc(&dout, nout * 4); hipMalloc(&dsc, hs.size() * 4);
    hipMemcpy(din, hin.data(), nin, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), nw, hipMemcpyHostToDevice);
    hipMemcpy(dsc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 3; round++) { run<12>(din, dw, dout, dsc, 200); run<9>(din, dw, dout, dsc, 200); }
    printf("(the shipped beamform_i8x3_kernel: 32 us by the same measure, profiles/beam_probe.py)\n");
    return 0;
}
