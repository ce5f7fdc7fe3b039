// Micro-benchmark (diagnostic, not part of the product): int8 MFMA shape vs sustained rate / in-kernel clock.
// Same output tile per wave (64x64 int32, 3 planes as the X-engine's R/P/Q would need 192 registers; here 64x64 x 1
// plane = 64 registers per shape), operands held in registers, MFMAs issued from inline asm so that the loop body is
// exactly the MFMAs.  One wave per SIMD, 256 work-groups.
//   mode 0: v_mfma_i32_32x32x32_i8, 4 accumulators (2x2 tiles), 8 operand quads per 32-sample step
//   mode 1: v_mfma_i32_16x16x64_i8, 16 accumulators (4x4 tiles), 8 operand quads per 64-sample step
//   mode 2/3: as 0/1 plus 3 (resp. 1.5) VALU mask/shift ops per MFMA on live registers (the X-engine's unpack load)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o mfma_shape ; run: ./mfma_shape [class]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MFMA32(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA16(acc, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const int* __restrict__ seed, int* __restrict__ out, unsigned long long* st, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    v4i a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) { a[i][j] = seed[(t * 32 + i * 4 + j) & 0xFFFFF]; b[i][j] = seed[(t * 32 + 16 + i * 4 + j) & 0xFFFFF]; }
    const v4i M = (v4i)(0xF0F0F0F0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int sum = 0;
    if (MODE == 0 || MODE == 2) {
        v16i c[2][2] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int kk = 0; kk < 2; kk++)          // two 32-sample steps = 64 samples
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < 2; n++) {
                        MFMA32(c[m][n], a[2 * kk + m], b[2 * kk + n]);
                        if (MODE == 2) {   // 3 VALU per MFMA: keep the operand class (x16-scaled nibbles) invariant
                            v4i& x = (n ? b : a)[2 * kk + m];
                            asm volatile("v_and_b32 %0, %0, %3\n\tv_lshlrev_b32 %1, 0, %1\n\tv_and_b32 %2, %2, %3"
                                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "v"(M[0]));
                        }
                    }
        }
        for (int m = 0; m < 2; m++) for (int n = 0; n < 2; n++) sum += c[m][n][0] + c[m][n][7];
    } else {
        v4i c[4][4] = {};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    MFMA16(c[m][n], a[m], b[n]);
                    if (MODE == 3 && (n & 1)) {
                        v4i& x = (m & 1) ? b[n] : a[m];
                        asm volatile("v_and_b32 %0, %0, %3\n\tv_lshlrev_b32 %1, 0, %1\n\tv_and_b32 %2, %2, %3"
                                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "v"(M[0]));
                    }
                }
        }
        for (int m = 0; m < 4; m++) for (int n = 0; n < 4; n++) sum += c[m][n][0] + c[m][n][3];
    }
    asm volatile("" :: "v"(sum));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = sum;
    if ((threadIdx.x & 63) == 0) { st[2 * (t >> 6)] = t1 - t0; st[2 * (t >> 6) + 1] = r1 - r0; }
}

static int g_reps = 60;
template <int MODE>
void run(const char* name, double ops_per_iter, int mfma_per_iter, const int* seed, int* out, unsigned long long* st) {
    const int iters = 8000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = g_reps;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, seed, out, st, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks * 4);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0; for (int w = 0; w < blocks * 4; w++) clk += (double)h[2 * w] / (double)h[2 * w + 1] * 100e6;
    clk /= blocks * 4;
    const double total_ops = (double)reps * blocks * 4 * iters * ops_per_iter;
    printf("%-34s %8.1f TOP/s   clock %.3f GHz   %.1f cycles/MFMA   (%.2f ms/launch)\n", name,
           total_ops / (ms * 1e-3) / 1e12, clk / 1e9, (double)h[0] / ((double)iters * mfma_per_iter), ms / reps);
}

int main(int argc, char** argv) {
    const int cls = argc > 1 ? atoi(argv[1]) : 2;
    int* seed; int* out; unsigned long long* st;
    std::vector<int> h(1 << 20);
    srand(1);
    for (auto& v : h) v = cls == 1 ? 0 : (int)((unsigned)rand() * 2654435761u ^ (unsigned)rand());
    if (cls == 2) for (auto& v : h) v &= 0xF0F0F0F0;
    if (cls == 4) for (auto& v : h) v &= 0x0F0F0F0F;
    (void)hipMalloc(&seed, h.size() * 4); (void)hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&st, 256 * 4 * 16);
    printf("operand class %d (0 random bits, 1 zeros, 2 x16-scaled 4-bit, 4 unsigned 4-bit)\n", cls);
    const double ops = 2.0 * 64 * 64 * 64;     // per iteration: a 64x64 tile over 64 samples
    const int only = argc > 2 ? atoi(argv[2]) : -1;      // run one mode for a long time (power sampling): mode, reps
    if (argc > 3) g_reps = atoi(argv[3]);
    for (int rep = 0; rep < (only >= 0 ? 1 : 2); rep++) {
        if (only < 0 || only == 0) run<0>("i8 32x32x32 (8 per 64 samples)", ops, 8, seed, out, st);
        if (only < 0 || only == 1) run<1>("i8 16x16x64 (16 per 64 samples)", ops, 16, seed, out, st);
        if (only < 0 || only == 2) run<2>("i8 32x32x32 + 3 VALU/MFMA", ops, 8, seed, out, st);
        if (only < 0 || only == 3) run<3>("i8 16x16x64 + 1.5 VALU/MFMA", ops, 16, seed, out, st);
    }
    return 0;
}
