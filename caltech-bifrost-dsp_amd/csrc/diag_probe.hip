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

}  // namespace xeng

extern "C" int xengDiagBpermProbe(int mode, int iters, int nblocks, unsigned long long* host8) {
    static hipStream_t s = nullptr;
    static unsigned long long* dev = nullptr;
    if (!s) {
        int lo = 0, hi = 0;
        XENG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        XENG_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
        XENG_HIP(hipMalloc((void**)&dev, 8 * sizeof(unsigned long long)));
    }
    XENG_HIP(hipMemsetAsync(dev, 0, 8 * sizeof(unsigned long long), s));
    if (mode == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_kernel<0>), dim3(nblocks), dim3(256), 0, s, iters, dev);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_kernel<1>), dim3(nblocks), dim3(256), 0, s, iters, dev);
    XENG_HIP(hipGetLastError());
    XENG_HIP(hipMemcpyAsync(host8, dev, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
