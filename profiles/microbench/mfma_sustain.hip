// Micro-benchmark (diagnostic, not part of the product): sustained rate and in-kernel clock of
// MFMA flavours on random operands held in registers: one wave per SIMD, 4 independent
// accumulators, ~2-3 ms per launch.  Build: hipcc -O3 --offload-arch=gfx950 mfma_sustain.hip -o mfma_sustain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const int* __restrict__ seed, float* __restrict__ out, unsigned long long* st, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    v8i a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) { a[i][j] = seed[(t * 64 + i * 8 + j) & 0xFFFFF]; b[i][j] = seed[(t * 64 + 32 + i * 8 + j) & 0xFFFFF]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0;
    if (MODE == 0) {   // i8 32x32x32
        v16i c[4] = {};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v4i x = {a[i][0], a[i][1], a[i][2], a[i][3]}, y = {b[i][0], b[i][1], b[i][2], b[i][3]};
                c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c[i], 0, 0, 0);
            }
        for (int i = 0; i < 4; i++) sum += c[i][0] + c[i][7];
    } else if (MODE == 1) {   // i8 16x16x64
        v4i32 c[4] = {};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                v4i x = {a[i][0], a[i][1], a[i][2], a[i][3]}, y = {b[i][0], b[i][1], b[i][2], b[i][3]};
                c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c[i], 0, 0, 0);
            }
        for (int i = 0; i < 4; i++) sum += c[i][0] + c[i][3];
    } else if (MODE == 2) {   // bf6 (e3m2) 32x32x64, scale 1.0
        v16f c[4] = {};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[i], c[i], 3, 3, 0, 127, 0, 127);
        for (int i = 0; i < 4; i++) sum += c[i][0] + c[i][7];
    } else if (MODE == 3) {   // bf6 16x16x128
        v4f c[4] = {};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[i], c[i], 3, 3, 0, 127, 0, 127);
        for (int i = 0; i < 4; i++) sum += c[i][0] + c[i][3];
    } else if (MODE == 4) {   // fp8 e4m3 via f8f6f4 32x32x64
        v16f c[4] = {};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[i], c[i], 0, 0, 0, 127, 0, 127);
        for (int i = 0; i < 4; i++) sum += c[i][0] + c[i][7];
    }
    asm volatile("" :: "v"(sum));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = sum;
    if ((threadIdx.x & 63) == 0) { st[2 * (t >> 6)] = t1 - t0; st[2 * (t >> 6) + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, double ops_per_mfma, const int* seed, float* out, unsigned long long* st, int nbits_random) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 40;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0; for (int w = 0; w < blocks * 4; w++) clk += (double)h[2 * w] / (double)h[2 * w + 1] * 100e6;
    clk /= blocks * 4;
    double mfmas = (double)reps * blocks * 4 * 4.0 * iters;
    double cyc_per = (double)h[0] / (4.0 * iters);
    printf("%-28s %8.1f TOP/s   clock %.3f GHz   %.1f cycles/MFMA   (%.2f ms/launch)\n", name, mfmas * ops_per_mfma / (ms * 1e-3) / 1e12, clk / 1e9, cyc_per, ms / reps);
}

int main(int argc, char** argv) {
    const int zero = argc > 1 && atoi(argv[1]) == 1;
    int* seed; float* out; unsigned long long* st;
    std::vector<int> h(1 << 20);
    srand(1);
    for (auto& v : h) v = zero ? 0 : (int)((unsigned)rand() * 2654435761u ^ (unsigned)rand());
    if (argc > 1 && atoi(argv[1]) == 2) for (auto& v : h) v &= 0xF0F0F0F0;   // like the x16-scaled nibbles
    if (argc > 1 && atoi(argv[1]) == 3)                                        // sign-extended 4-bit values per byte
        for (auto& v : h) { unsigned o = 0; for (int k = 0; k < 4; k++) { int b = (signed char)((v >> (8 * k)) & 0xF0) >> 4; o |= (unsigned)(b & 0xFF) << (8 * k); } v = (int)o; }
    if (argc > 1 && atoi(argv[1]) == 4) for (auto& v : h) v &= 0x0F0F0F0F;   // unsigned 4-bit values per byte
    if (argc > 1 && atoi(argv[1]) == 5) for (auto& v : h) v &= 0x07070707;   // unsigned 3-bit values per byte
    hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&st, 256 * 4 * 16);
    printf("operands: class %s (0 random bits, 1 zeros, 2 x16-scaled 4-bit, 3 sign-extended 4-bit, 4 unsigned 4-bit, 5 unsigned 3-bit)\n", argc > 1 ? argv[1] : "0");
    run<0>("i8  32x32x32", 2.0 * 32 * 32 * 32, seed, out, st, 0);
    run<1>("i8  16x16x64", 2.0 * 16 * 16 * 64, seed, out, st, 0);
    run<2>("bf6(e3m2) 32x32x64 scaled", 2.0 * 32 * 32 * 64, seed, out, st, 0);
    run<3>("bf6(e3m2) 16x16x128 scaled", 2.0 * 16 * 16 * 128, seed, out, st, 0);
    run<4>("fp8(e4m3) 32x32x64 scaled", 2.0 * 32 * 32 * 64, seed, out, st, 0);
    return 0;
}
