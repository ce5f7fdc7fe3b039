// Stand-alone reproducer for the upstream report (profiles/hazard/REPORT.md): a packed-fp32 multiply with op_sel:[0,1] returns a
// wrong LOW result in lanes 48-63 while an MFMA kernel shares the CUs.  No libxeng: one MFMA spin kernel, one probe kernel.
//   hipcc -O2 --offload-arch=gfx950 repro_standalone.hip -o repro && ./repro
// Prints the wrong / checked counts of the probe on an idle GPU and beside the MFMA kernel, for op_sel:[0,1] and, as the
// control, the same products with the sources swapped (op_sel:[1,0]).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// 64 KB of LDS per work-group: two groups (8 waves) per CU, so the probe's waves find room beside them
__global__ __launch_bounds__(256) void mfma_spin(int iters, int* sink) {
    __shared__ int pad[16384];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    v16i c = {};
    pad[threadIdx.x] = (int)blockIdx.x;
    for (int i = 0; i < iters; i++) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    if (c[0] == 0x7fffffff) *sink = c[1] + pad[threadIdx.x ^ 1];
}

template <int SWAPPED>
__global__ __launch_bounds__(256) void pk_probe(int iters, const float* __restrict__ tab, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned nerr = 0, nq[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const int base = ((gw * 131 + it * 17) & 1023) * 64;
        const v2f a = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane) & 65535));
        const v2f b = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane + 7) & 65535));
        v2f d = {a.y, a.x};
        float wlo, whi;                                                  // a.y * b.y | a.x * b.y by scalar multiplies
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(wlo), "=&v"(whi) : "v"(a.y), "v"(a.x), "v"(b.y));
        if (SWAPPED) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(d) : "v"(b));
        else asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(d) : "v"(b));
        if (d.x != wlo || d.y != whi) { nerr++; nq[lane >> 4]++; }
    }
    if (nerr) {
        atomicAdd(&out[0], (unsigned long long)nerr);
        for (int k = 0; k < 4; k++) atomicAdd(&out[1 + k], (unsigned long long)nq[k]);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    std::vector<float> h(2 * 65536);
    unsigned r = 12345u;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (float)((int)((r >> 20) & 31) - 16); }
    float* tab; unsigned long long* out; int* sink;
    CK(hipMalloc((void**)&tab, h.size() * sizeof(float)));
    CK(hipMalloc((void**)&out, 8 * sizeof(unsigned long long)));
    CK(hipMalloc((void**)&sink, sizeof(int)));
    CK(hipMemcpy(tab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    const int iters = 600, nblocks = 1024, reps = 8;
    for (int beside = 0; beside < 2; beside++)
        for (int swapped = 0; swapped < 2; swapped++) {
            unsigned long long tot[5] = {0, 0, 0, 0, 0};
            for (int rep = 0; rep < reps; rep++) {
                CK(hipMemsetAsync(out, 0, 8 * sizeof(unsigned long long), sb));
                if (beside) for (int k = 0; k < 4; k++) hipLaunchKernelGGL(mfma_spin, dim3(512), dim3(256), 0, sa, 40000, sink);
                if (swapped) hipLaunchKernelGGL(pk_probe<1>, dim3(nblocks), dim3(256), 0, sb, iters, tab, out);
                else hipLaunchKernelGGL(pk_probe<0>, dim3(nblocks), dim3(256), 0, sb, iters, tab, out);
                unsigned long long host[5];
                CK(hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, sb));
                CK(hipStreamSynchronize(sb));
                CK(hipStreamSynchronize(sa));
                for (int k = 0; k < 5; k++) tot[k] += host[k];
            }
            printf("%-22s %-34s %llu wrong of %.2e; by lane quarter [%llu, %llu, %llu, %llu]\n", beside ? "beside an MFMA kernel:" : "idle GPU:",
                   swapped ? "v_pk_mul_f32 D, S, D op_sel:[1,0]" : "v_pk_mul_f32 D, D, S op_sel:[0,1]", tot[0],
                   (double)reps * iters * nblocks * 256, tot[1], tot[2], tot[3], tot[4]);
        }
    return 0;
}
