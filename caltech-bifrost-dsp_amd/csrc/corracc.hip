// CorrAcc long accumulation: replaces bifrost.map("a = b") / ("a += b") on the planar
// int32 xGPU buffer (corr_acc_block.py:304,306).  HBM-bound: 16 B per lane, grid-stride,
// 2048 blocks (8 per CU) so every XCD streams.
#include "xeng_common.h"

namespace xeng {

template <bool ADD>
__global__ __launch_bounds__(256) void map_i32_kernel(int4* __restrict__ a, const int4* __restrict__ b,
                                                      size_t n16, int32_t* __restrict__ a_tail,
                                                      const int32_t* __restrict__ b_tail, int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += stride) {
        int4 y = b[k];
        if (ADD) {
            const int4 x = a[k];
            y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w;
        }
        a[k] = y;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) {
        if (ADD) a_tail[threadIdx.x] += b_tail[threadIdx.x];
        else a_tail[threadIdx.x] = b_tail[threadIdx.x];
    }
}

static int map_i32(void* a, const void* b, size_t nwords, bool add) {
    if (!a || !b) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "map: null buffer");
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "map: buffers must be 16-byte aligned");
    if (nwords == 0) return XENG_STATUS_SUCCESS;
    hipStream_t s;
    int rc = get_stream(STREAM_MAP, &s);
    if (rc) return rc;
    const size_t n16 = nwords / 4;
    const int ntail = (int)(nwords % 4);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    int32_t* at = (int32_t*)a + n16 * 4;
    const int32_t* bt = (const int32_t*)b + n16 * 4;
    if (add)
        hipLaunchKernelGGL(map_i32_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (int4*)a, (const int4*)b, n16, at, bt, ntail);
    else
        hipLaunchKernelGGL(map_i32_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (int4*)a, (const int4*)b, n16, at, bt, ntail);
    XENG_HIP(hipGetLastError());
    return XENG_STATUS_SUCCESS;
}

}  // namespace xeng

extern "C" {
int xengMapAssignI32(void* a_dev, const void* b_dev, size_t nwords) { return xeng::map_i32(a_dev, b_dev, nwords, false); }
int xengMapAddI32(void* a_dev, const void* b_dev, size_t nwords) { return xeng::map_i32(a_dev, b_dev, nwords, true); }
int xengMapSync(void) {
    hipStream_t s;
    int rc = xeng::get_stream(xeng::STREAM_MAP, &s);
    if (rc) return rc;
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
}
