// Micro-benchmark (diagnostic): how many independent VALU ops fit beside int8 MFMAs on one SIMD, with one or two waves per
// SIMD.  The VALU ops work on registers no MFMA touches (no hazards), the MFMAs on independent accumulators.
// Pattern per iteration (a 64x32 wave tile over 64 samples): 16x16x64: 32 MFMAs + V VALU;  32x32x32: 16 MFMAs + V VALU.
// Build: hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// SHAPE 16: 8 tiles of 16x16, 3 planes -> 24 accumulators of 4 regs; SHAPE 32: 2 tiles of 32x32, 3 planes -> 6 x 16 regs
template <int SHAPE, int NW, int VPM4>   // VPM4 = VALU per MFMA x 4 (for 16x16x64) or per MFMA x 1 (32x32x32: VALU per MFMA)
__global__ __launch_bounds__(64 * NW, NW / 4) void k(const int* __restrict__ seed, int* __restrict__ out, unsigned long long* st, int iters) {
    const int t = blockIdx.x * 64 * NW + threadIdx.x;
    v4i a[4], b[2], x[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = seed[(t * 40 + i * 4 + j) & 0xFFFFF] & 0xF0F0F0F0; x[i][j] = seed[(t * 40 + 24 + i * 4 + j) & 0xFFFFF]; }
    for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) b[i][j] = seed[(t * 40 + 16 + i * 4 + j) & 0xFFFFF] & 0xF0F0F0F0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int sum = 0;
    if (SHAPE == 16) {
        v4i c[24] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 32; q++) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[(q * 7) % 24]) : "v"(a[q & 3]), "v"(b[(q >> 2) & 1]));
                // VPM4/4 VALU per MFMA on average
                constexpr int nv = VPM4 / 4, extra = VPM4 % 4;
#pragma unroll
                for (int v = 0; v < nv + ((q & 3) < extra ? 1 : 0); v++)
                    asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0" : "+v"(x[(q + v) & 3][v & 3]));
            }
        }
        for (int i = 0; i < 24; i++) sum += c[i][0] + c[i][3];
    } else {
        v16i c[6] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c[(q * 5) % 6]) : "v"(a[q & 3]), "v"(b[(q >> 2) & 1]));
#pragma unroll
                for (int v = 0; v < VPM4; v++)
                    asm volatile("v_and_b32 %0, 0xf0f0f0f0, %0" : "+v"(x[(q + v) & 3][v & 3]));
            }
        }
        for (int i = 0; i < 6; i++) sum += c[i][0] + c[i][9];
    }
    for (int i = 0; i < 4; i++) sum += x[i][0] + x[i][1] + x[i][2] + x[i][3];
    asm volatile("" :: "v"(sum));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = sum;
    if ((threadIdx.x & 63) == 0) { st[2 * (t >> 6)] = t1 - t0; st[2 * (t >> 6) + 1] = r1 - r0; }
}

template <int SHAPE, int NW, int VPM4>
void run(const int* seed, int* out, unsigned long long* st) {
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(HIP_KERNEL_NAME(k<SHAPE, NW, VPM4>), dim3(blocks), dim3(64 * NW), 0, 0, seed, out, st, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    const int reps = 40;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL(HIP_KERNEL_NAME(k<SHAPE, NW, VPM4>), dim3(blocks), dim3(64 * NW), 0, 0, seed, out, st, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks * NW);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0, cyc = 0; for (int w = 0; w < blocks * NW; w++) { clk += (double)h[2 * w] / (double)h[2 * w + 1] * 100e6; cyc += (double)h[2 * w]; }
    clk /= blocks * NW; cyc /= blocks * NW;
    const double nm = SHAPE == 16 ? 32 : 16;
    const double ops = (double)reps * blocks * NW * iters * nm * (SHAPE == 16 ? 2.0 * 16 * 16 * 64 : 2.0 * 32 * 32 * 32);
    const double vpm = SHAPE == 16 ? VPM4 / 4.0 : VPM4;
    printf("%dx%d  %d wave(s)/SIMD  %.2f VALU/MFMA: %8.1f TOP/s  clock %.3f GHz  %.1f cycles/MFMA per wave  (pipe-bound: %d per SIMD)\n",
           SHAPE, SHAPE, NW / 4, vpm, ops / (ms * 1e-3) / 1e12, clk / 1e9, cyc / (iters * nm), SHAPE == 16 ? 16 : 32);
}

int main() {
    int* seed; int* out; unsigned long long* st;
    std::vector<int> h(1 << 20);
    srand(1); for (auto& v : h) v = (int)((unsigned)rand() * 2654435761u ^ (unsigned)rand());
    (void)hipMalloc(&seed, h.size() * 4); (void)hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 16);
    run<16, 4, 0>(seed, out, st); run<16, 4, 4>(seed, out, st); run<16, 4, 6>(seed, out, st); run<16, 4, 8>(seed, out, st); run<16, 4, 9>(seed, out, st);
    run<16, 8, 0>(seed, out, st); run<16, 8, 4>(seed, out, st); run<16, 8, 6>(seed, out, st); run<16, 8, 8>(seed, out, st); run<16, 8, 9>(seed, out, st); run<16, 8, 12>(seed, out, st);
    run<32, 4, 0>(seed, out, st); run<32, 4, 3>(seed, out, st); run<32, 4, 5>(seed, out, st); run<32, 4, 6>(seed, out, st);
    run<32, 8, 0>(seed, out, st); run<32, 8, 3>(seed, out, st); run<32, 8, 5>(seed, out, st); run<32, 8, 6>(seed, out, st);
    return 0;
}
