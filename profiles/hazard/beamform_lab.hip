// Hazard lab: csrc/beamform.hip with the ROUND-2 form of beam_integrate_kernel in place of the shipped
// one -- packed-fp32 accumulation (the compiler's choice) reduced over 8-lane groups with __shfl_xor = ds_bpermute_b32.
// That form returned wrong cross-power sums in lanes 48-63 whenever an MFMA kernel shared the CU.  It is compiled to
// assembly here; profiles/hazard/build.py then applies ONE instruction-level edit per variant to that assembly and
// links each into a scratch library (profiles/hazard/lib/, never libxeng.so).  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define XENG_EXTERNAL_BEAM_INTEGRATE 1
namespace xeng {
__global__ __launch_bounds__(256) void beam_integrate_kernel(const float2* __restrict__ in, float4* __restrict__ out,
                                                             int nchan, int nbeam, int ntime, int ntime_sum,
                                                             int pair0, int npair_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave, bp = blockIdx.y;
    if (c >= nchan) return;
    const int nblk = ntime / ntime_sum;
    const float2* x = in + ((size_t)c * nbeam + 2 * (pair0 + bp)) * ntime;
    const float2* y = x + ntime;
    const int sub = lane & 7, grp = lane >> 3;
    for (int tb = blockIdx.z * 8 + grp; tb < nblk; tb += 8 * gridDim.z) {
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
            const float a0 = __shfl_xor(xx, o), a1 = __shfl_xor(yy, o), a2 = __shfl_xor(xyr, o), a3 = __shfl_xor(xyi, o);
            xx += a0; yy += a1; xyr += a2; xyi += a3;
        }
        if (sub == 0) out[((size_t)bp * nblk + tb) * nchan + c] = make_float4(xx, yy, xyr, xyi);
    }
    (void)npair_out;
}
}  // namespace xeng
#include "../../caltech-bifrost-dsp_amd/csrc/beamform.hip"
