// Diagnostic only (not part of libxeng.so: profiles/bperm_probe.sh links it into a scratch library).
// Does a cross-lane read keep returning the right data while other kernels share the CU?  Every wave reads known
// tags from lane ^ 4 in groups of four back-to-back operations, as beam_integrate_kernel's 8-lane reduction did,
// and counts what comes back wrong.  mode 0: ds_bpermute_b32; mode 1: DPP row shifts (row_shl:4 / row_shr:4).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xeng_common.h"

namespace xeng {

template <int MODE>
__global__ __launch_bounds__(256) void bperm_probe_kernel(int iters, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    const int addr = (lane ^ 4) << 2;
    unsigned nerr = 0, nk[4] = {0, 0, 0, 0}, nhi = 0;
    unsigned long long first = 0;
    for (int it = 0; it < iters; it++) {
        int v[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (gw << 18) ^ ((it & 0x3FF) << 8) ^ (k << 6) ^ lane;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (MODE == 0) r[k] = __builtin_amdgcn_ds_bpermute(addr, v[k]);
            else {
                const int up = __builtin_amdgcn_update_dpp(0, v[k], 0x104 /* row_shl:4 */, 0xF, 0xF, true);
                const int dn = __builtin_amdgcn_update_dpp(0, v[k], 0x114 /* row_shr:4 */, 0xF, 0xF, true);
                r[k] = (lane & 4) ? dn : up;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int want = (gw << 18) ^ ((it & 0x3FF) << 8) ^ (k << 6) ^ (lane ^ 4);
            if (r[k] != want) {
                nerr++; nk[k]++; nhi += lane >= 48;
                if (!first) first = ((unsigned long long)(unsigned)r[k] << 32) | (unsigned)want;
            }
        }
    }
    if (nerr) {
        atomicAdd(&out[0], (unsigned long long)nerr);
        for (int k = 0; k < 4; k++) atomicAdd(&out[1 + k], (unsigned long long)nk[k]);
        atomicAdd(&out[5], (unsigned long long)nhi);
        atomicCAS(&out[6], 0ull, first);
    }
}

// mode 2/3: the values that travel are fp32 sums formed by packed-fp32 instructions right before the cross-lane read
// (as in the beam_integrate_kernel that failed): 2 = ds_bpermute_b32, 3 = DPP.  The terms come from memory so that the
// compiler cannot fold them; all values are small integers, so every sum is exact and the expected result can be
// recomputed for lane ^ 4 from the same table.
template <int MODE>
__global__ __launch_bounds__(256) void bperm_probe_pk_kernel(int iters, const float* __restrict__ tab,
                                                             unsigned long long* __restrict__ out) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    const int addr = (lane ^ 4) << 2;
    unsigned nerr = 0, nk[4] = {0, 0, 0, 0}, nhi = 0;
    unsigned long long first = 0;
    for (int it = 0; it < iters; it++) {
        const int base = ((gw * 131 + it * 17) & 1023) * 64;
        v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, b0 = {0.f, 0.f}, b1 = {0.f, 0.f};      // own sums, and those of lane ^ 4
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const v2f x = *reinterpret_cast<const v2f*>(tab + 2 * (((base + lane) + 64 * t) & 65535));
            const v2f y = *reinterpret_cast<const v2f*>(tab + 2 * (((base + lane) + 64 * t + 7) & 65535));
            a0 += x * x + y * y;
            a1 += x * y + (v2f){y.y, y.x} * (v2f){x.y, -x.x};
            const v2f xs = *reinterpret_cast<const v2f*>(tab + 2 * (((base + (lane ^ 4)) + 64 * t) & 65535));
            const v2f ys = *reinterpret_cast<const v2f*>(tab + 2 * (((base + (lane ^ 4)) + 64 * t + 7) & 65535));
            b0 += xs * xs + ys * ys;
            b1 += xs * ys + (v2f){ys.y, ys.x} * (v2f){xs.y, -xs.x};
        }
        float v[4] = {a0.x, a0.y, a1.x, a1.y}, want[4] = {b0.x, b0.y, b1.x, b1.y}, r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (MODE == 2) r[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v[k])));
            else {
                const int up = __builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x104, 0xF, 0xF, true);
                const int dn = __builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x114, 0xF, 0xF, true);
                r[k] = __int_as_float((lane & 4) ? dn : up);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (r[k] != want[k]) {
                nerr++; nk[k]++; nhi += lane >= 48;
                if (!first) first = ((unsigned long long)__float_as_uint(r[k]) << 32) | __float_as_uint(want[k]);
            }
    }
    if (nerr) {
        atomicAdd(&out[0], (unsigned long long)nerr);
        for (int k = 0; k < 4; k++) atomicAdd(&out[1 + k], (unsigned long long)nk[k]);
        atomicAdd(&out[5], (unsigned long long)nhi);
        atomicCAS(&out[6], 0ull, first);
    }
}

}  // namespace xeng

extern "C" int xengDiagBpermProbe(int mode, int iters, int nblocks, unsigned long long* host8) {
    static hipStream_t s = nullptr;
    static unsigned long long* dev = nullptr;
    static float* tab = nullptr;
    if (!s) {
        int lo = 0, hi = 0;
        XENG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        XENG_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
        XENG_HIP(hipMalloc((void**)&dev, 8 * sizeof(unsigned long long)));
        XENG_HIP(hipMalloc((void**)&tab, 2 * 65536 * sizeof(float)));
        float* h = new float[2 * 65536];
        unsigned r = 12345u;
        for (int i = 0; i < 2 * 65536; i++) { r = r * 1664525u + 1013904223u; h[i] = (float)((int)((r >> 20) & 31) - 16); }
        XENG_HIP(hipMemcpy(tab, h, 2 * 65536 * sizeof(float), hipMemcpyHostToDevice));
        delete[] h;
    }
    XENG_HIP(hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), s));
    if (mode == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_kernel<0>), dim3(nblocks), dim3(256), 0, s, iters, dev);
    else if (mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_kernel<1>), dim3(nblocks), dim3(256), 0, s, iters, dev);
    else if (mode == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_pk_kernel<2>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_pk_kernel<3>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    XENG_HIP(hipGetLastError());
    XENG_HIP(hipMemcpyAsync(host8, dev, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
