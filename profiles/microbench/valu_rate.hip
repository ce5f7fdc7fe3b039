// Issue rate of the VALU instructions the X-engine's operand unpack could use (gfx950): cycles per wave64 instruction,
// one wave per SIMD, 16 independent chains.   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(unsigned* out, unsigned long long* cyc, int iters, unsigned seed) {
    unsigned v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = seed * (threadIdx.x + 1) + k * 0x01234567u;
    const unsigned a = seed ^ 0x0F0F0F0Fu, b = seed | 0x80808080u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define ONE(k)                                                                                                        \
    if (KIND == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[k]) : "v"(a));                                        \
    if (KIND == 1) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));                            \
    if (KIND == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(v[k]) : "v"(a), "v"(b));             \
    if (KIND == 3) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(v[k]) : "v"(a));                                \
    if (KIND == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));                           \
    if (KIND == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));                           \
    if (KIND == 6) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(v[k]));                                              \
    if (KIND == 7) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));                             \
    if (KIND == 8) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[k]) : "v"(a));                                     \
    if (KIND == 9) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[k]) : "v"(a));                                        \
    if (KIND == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));                        \
    if (KIND == 11) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[k]) : "v"(a), "v"(b));                           \
    if (KIND == 12) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[k]) : "v"(a));                               \
    if (KIND == 13) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(v[k]));                                          \
    if (KIND == 14) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(v[k]));                                          \
    if (KIND == 15) asm volatile("v_dot4_i32_i8 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
        OPS(ONE)
#undef ONE
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) r ^= v[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
static void run(const char* name, unsigned* out, unsigned long long* cyc) {
    const int iters = 4096, nb = 256;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(nb), dim3(256), 0, 0, out, cyc, iters, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += (double)x;
    // s_memtime = shader clock
    printf("%-18s %8.3f shader cycles per 1000 instr (s_memtime)\n", name, s / nb / (iters * 16.0) * 1000.0);
}
int main() {
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("v_and_b32", out, cyc); run<1>("v_lerp_u8", out, cyc); run<2>("v_bitop3_b32", out, cyc);
    run<3>("v_add_lshl_u32", out, cyc); run<4>("v_add3_u32", out, cyc); run<5>("v_perm_b32", out, cyc);
    run<6>("v_lshlrev_b32", out, cyc); run<7>("v_sad_u8", out, cyc); run<8>("v_pk_add_u16", out, cyc);
    run<9>("v_xor_b32", out, cyc); run<10>("v_and_or_b32", out, cyc); run<11>("v_bfi_b32", out, cyc);
    run<12>("v_lshl_add_u32", out, cyc); run<13>("v_pk_lshrrev_b16", out, cyc); run<14>("v_pk_ashrrev_i16", out, cyc);
    run<15>("v_dot4_i32_i8", out, cyc);
    return 0;
}
