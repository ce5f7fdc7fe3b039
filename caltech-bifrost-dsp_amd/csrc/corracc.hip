// CorrAcc long accumulation: replaces bifrost.map("a = b") / ("a += b") on the planar
// int32 xGPU buffer (corr_acc_block.py:304,306).  HBM-bound: 16 B per lane, grid-stride,
// 2048 blocks (8 per CU) so every XCD streams.
#include <cstdlib>

#include "xeng_common.h"

namespace xeng {

// U = 16-byte pieces in flight per lane: every wave instruction covers one contiguous KiB, a work-group 4 KiB per
// piece; all loads of an iteration are issued before the first add.  NT: non-temporal loads and stores (the 574 MB of
// an "a += b" over config-2 planes pass through the caches once).
template <bool ADD, int U, bool NT>
__global__ __launch_bounds__(256) void map_i32_kernel(int4* __restrict__ a, const int4* __restrict__ b,
                                                      size_t n16, int32_t* __restrict__ a_tail,
                                                      const int32_t* __restrict__ b_tail, int ntail) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (size_t k0 = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; k0 < n16; k0 += stride) {
        v4i_ x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t k = k0 + (size_t)u * blockDim.x;
            if (k < n16) {
                y[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(b) + k) : reinterpret_cast<const v4i_*>(b)[k];
                if (ADD) x[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(a) + k) : reinterpret_cast<const v4i_*>(a)[k];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t k = k0 + (size_t)u * blockDim.x;
            if (k < n16) {
                if (ADD) y[u] += x[u];
                if (NT) __builtin_nontemporal_store(y[u], reinterpret_cast<v4i_*>(a) + k);
                else reinterpret_cast<v4i_*>(a)[k] = y[u];
            }
        }
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
    int32_t* at = (int32_t*)a + n16 * 4;
    const int32_t* bt = (const int32_t*)b + n16 * 4;
    // XENG_MAP_VAR = U*10 + NT overrides the defaults measured on config-2 planes (profiles/side_kernels.py map):
    // "a += b" one piece per lane, non-temporal (88.8 us = 6.46 TB/s); "a = b" four pieces, non-temporal (62.6 us = 6.11 TB/s)
    static const int var_env = diag_env("XENG_MAP_VAR") ? atoi(diag_env("XENG_MAP_VAR")) : 0;
    const int var = var_env ? var_env : (add ? 11 : 41);
    static const int maxb = diag_env("XENG_MAP_BLOCKS") ? atoi(diag_env("XENG_MAP_BLOCKS")) : 2048;
    const int U = var / 10 == 1 ? 1 : var / 10 == 2 ? 2 : var / 10 == 8 ? 8 : 4;
    size_t blocks = (n16 + 256 * (size_t)U - 1) / (256 * (size_t)U);
    if (blocks > (size_t)maxb) blocks = (size_t)maxb;
    if (blocks == 0) blocks = 1;
    const dim3 g((unsigned)blocks), t(256);
#define XENG_MAP_LAUNCH(ADD_, U_, NT_) hipLaunchKernelGGL(HIP_KERNEL_NAME(map_i32_kernel<ADD_, U_, NT_>), g, t, 0, s, (int4*)a, (const int4*)b, n16, at, bt, ntail)
#define XENG_MAP_PICK(ADD_)                                                      \
    switch (var) {                                                               \
        case 10: XENG_MAP_LAUNCH(ADD_, 1, false); break;                         \
        case 11: XENG_MAP_LAUNCH(ADD_, 1, true); break;                          \
        case 20: XENG_MAP_LAUNCH(ADD_, 2, false); break;                         \
        case 21: XENG_MAP_LAUNCH(ADD_, 2, true); break;                          \
        case 40: XENG_MAP_LAUNCH(ADD_, 4, false); break;                         \
        case 80: XENG_MAP_LAUNCH(ADD_, 8, false); break;                         \
        case 81: XENG_MAP_LAUNCH(ADD_, 8, true); break;                          \
        default: XENG_MAP_LAUNCH(ADD_, 4, true); break;                          \
    }
    if (add) { XENG_MAP_PICK(true) } else { XENG_MAP_PICK(false) }
#undef XENG_MAP_PICK
#undef XENG_MAP_LAUNCH
    XENG_HIP(hipGetLastError());
    stream_tick(STREAM_MAP);
    return XENG_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------------------
// Deferred long accumulation (round 5): a (+)= b_0 + b_1 + ... + b_{N-1} in ONE pass.
//
// The reference adds every upstream dump into its accumulator as it arrives (corr_acc_block.py:298-306): per dump 191 MB read
// + 191 MB read-modify-written = 574 MB of HBM traffic.  That order is a memory-saving choice, not part of the result -- int32
// addition wraps, so any grouping of the sum gives the same words -- and an MI355X has 288 GB: CorrAcc keeps the visibility
// spans of up to N dumps (they sit in the corr-output ring anyway) and sums them together, reading every span once and
// touching the accumulator once per GROUP: 191 + 382 / N MB per dump (N = 10: 229 MB, 0.4 x the per-dump map; the
// contraction-epilogue variant of round 3 moves 382 MB per dump and stalls the MFMA pipe for its read-modify-write).
// All N + 1 streams of a piece are in flight before the first add; non-temporal both ways (every byte passes once).
struct SumSources { const int4* b[XENG_MAP_SUM_MAX]; };

template <int N, bool ADD>
__global__ __launch_bounds__(256) void map_sum_i32_kernel(int4* __restrict__ a, SumSources src, size_t n16, int ntail) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += stride) {
        v4i_ y[N], x = (v4i_)(0);
#pragma unroll
        for (int j = 0; j < N; j++) y[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(src.b[j]) + k);
        if (ADD) x = __builtin_nontemporal_load(reinterpret_cast<const v4i_*>(a) + k);
#pragma unroll
        for (int j = 0; j < N; j++) x += y[j];
        __builtin_nontemporal_store(x, reinterpret_cast<v4i_*>(a) + k);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) {
        int32_t* at = reinterpret_cast<int32_t*>(a) + n16 * 4 + threadIdx.x;
        int32_t v = ADD ? *at : 0;
#pragma unroll
        for (int j = 0; j < N; j++) v += (reinterpret_cast<const int32_t*>(src.b[j]) + n16 * 4)[threadIdx.x];
        *at = v;
    }
}

template <int N>
static void launch_sum(dim3 g, hipStream_t s, int4* a, const SumSources& src, size_t n16, int ntail, bool add) {
    if (add) hipLaunchKernelGGL(HIP_KERNEL_NAME(map_sum_i32_kernel<N, true>), g, dim3(256), 0, s, a, src, n16, ntail);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(map_sum_i32_kernel<N, false>), g, dim3(256), 0, s, a, src, n16, ntail);
}

static int map_sum_i32(void* a, const void* const* srcs, int nsrc, size_t nwords, bool add) {
    if (!a || !srcs || nsrc < 1) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "map sum: null buffer / no sources");
    if (nsrc > XENG_MAP_SUM_MAX) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "map sum: %d sources, at most %d per call", nsrc, XENG_MAP_SUM_MAX);
    if ((uintptr_t)a & 15) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "map sum: buffers must be 16-byte aligned");
    SumSources src;
    for (int j = 0; j < nsrc; j++) {
        if (!srcs[j] || ((uintptr_t)srcs[j] & 15)) XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "map sum: source %d is null or not 16-byte aligned", j);
        src.b[j] = (const int4*)srcs[j];
    }
    if (nwords == 0) return XENG_STATUS_SUCCESS;
    hipStream_t s;
    int rc = get_stream(STREAM_MAP, &s);
    if (rc) return rc;
    const size_t n16 = nwords / 4;
    const int ntail = (int)(nwords % 4);
    static const int maxb = diag_env("XENG_MAP_BLOCKS") ? atoi(diag_env("XENG_MAP_BLOCKS")) : 2048;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > (size_t)maxb) blocks = (size_t)maxb;
    if (blocks == 0) blocks = 1;
    const dim3 g((unsigned)blocks);
    switch (nsrc) {
#define XENG_SUM_CASE(N_) case N_: launch_sum<N_>(g, s, (int4*)a, src, n16, ntail, add); break;
        XENG_SUM_CASE(1) XENG_SUM_CASE(2) XENG_SUM_CASE(3) XENG_SUM_CASE(4) XENG_SUM_CASE(5) XENG_SUM_CASE(6) XENG_SUM_CASE(7) XENG_SUM_CASE(8)
        XENG_SUM_CASE(9) XENG_SUM_CASE(10) XENG_SUM_CASE(11) XENG_SUM_CASE(12) XENG_SUM_CASE(13) XENG_SUM_CASE(14) XENG_SUM_CASE(15) XENG_SUM_CASE(16)
#undef XENG_SUM_CASE
    }
    XENG_HIP(hipGetLastError());
    stream_tick(STREAM_MAP);
    return XENG_STATUS_SUCCESS;
}

}  // namespace xeng

extern "C" {
int xengMapSumI32(void* a_dev, const void* const* srcs_dev, int nsrc, size_t nwords, int add) { return xeng::map_sum_i32(a_dev, srcs_dev, nsrc, nwords, add != 0); }
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
