// Diagnostic only (not part of libxeng.so: profiles/hazard/probe.sh links it into a scratch library).
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

// mode 4-7: the instruction the hazard lab (profiles/hazard/build.py) singled out in the round-2 beam_integrate_kernel:
//     v_pk_mul_f32 v[30:31], v[30:31], v[24:25] op_sel:[0,1]      (D = S0; the LOW result takes the HIGH register of S1)
// returned a wrong low product in lanes 48-63 whenever an MFMA kernel shared the CU.  Here it runs on its own, on known
// small integers fetched from memory as in that kernel (two global_load_dwordx2, then the packed multiply), and both
// halves are compared with v_mul_f32 products.   4: that form   5: sources swapped (D = S1, op_sel:[1,0])
// 6: destination in a third register pair   7: the same products by two v_mul_f32 (control)
template <int MODE>
__global__ __launch_bounds__(256) void pkmul_probe_kernel(int iters, const float* __restrict__ tab,
                                                          unsigned long long* __restrict__ out) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    unsigned nerr = 0, nlo = 0, nhalf[4] = {0, 0, 0, 0};
    unsigned long long first = 0;
    for (int it = 0; it < iters; it++) {
        const int base = ((gw * 131 + it * 17) & 1023) * 64;
        const v2f a = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane) & 65535));          // (a.x, a.y)
        const v2f b = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane + 7) & 65535));      // (b.x, b.y)
        v2f d = {a.y, a.x};                                                                      // as the kernel arranged v[30:31]
        float wlo, whi;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(wlo), "=&v"(whi) : "v"(a.y), "v"(a.x), "v"(b.y));
        if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(d) : "v"(b));
        else if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(d) : "v"(b));
        else if (MODE == 6) { v2f e; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(e) : "v"(d), "v"(b)); d = e; }
        else asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(d.x), "=&v"(d.y) : "v"(a.y), "v"(a.x), "v"(b.y));
        const bool blo = d.x != wlo, bhi = d.y != whi;
        if (blo || bhi) {
            nerr++; nlo += blo; nhalf[lane >> 4]++;
            if (!first) first = ((unsigned long long)__float_as_uint(blo ? d.x : d.y) << 32) | __float_as_uint(blo ? wlo : whi);
        }
    }
    if (nerr) {
        atomicAdd(&out[0], (unsigned long long)nerr);
        for (int k = 0; k < 4; k++) atomicAdd(&out[1 + k], (unsigned long long)nhalf[k]);     // by lane quarter
        atomicAdd(&out[5], (unsigned long long)nlo);                                           // wrong LOW results
        atomicCAS(&out[6], 0ull, first);
    }
}

// mode 100 + 16*op + m: sweep of the packed-fp32 operand selects.  op 0: v_pk_mul_f32, 1: v_pk_add_f32, 2: v_pk_fma_f32 (the
// selects of src0/src1 as given, src2 taken straight), 3: v_pk_fma_f32 with the selects applied to src1/src2 (src0 straight).
// m = op_sel[0] | op_sel[1] << 1 | op_sel_hi[0] << 2 | op_sel_hi[1] << 3 for the two swept sources: the low result takes the
// high register of a source when its op_sel bit is 1, the high result takes the LOW register when its op_sel_hi bit is 0.
#define PK_SWEEP_CASE(OPNAME, M, TAIL)                                                                                        \
    asm volatile(OPNAME " %0, %1, %2" TAIL : "=&v"(d) : "v"(a), "v"(b), "v"(c))
template <int OP, int M>
__global__ __launch_bounds__(256) void pk_sweep_kernel(int iters, const float* __restrict__ tab, unsigned long long* __restrict__ out) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    constexpr int s0 = M & 1, s1 = (M >> 1) & 1, h0 = (M >> 2) & 1, h1 = (M >> 3) & 1;
    unsigned nerr = 0, nlo = 0, nq[4] = {0, 0, 0, 0};
    unsigned long long first = 0;
    for (int it = 0; it < iters; it++) {
        const int base = ((gw * 131 + it * 17) & 1023) * 64;
        const v2f a = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane) & 65535));
        const v2f b = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane + 7) & 65535));
        const v2f c = *reinterpret_cast<const v2f*>(tab + 2 * ((base + lane + 19) & 65535));
        v2f d;
        // the two swept sources: (a, b) for op 0-2, (b, c) for op 3
        const v2f p = OP == 3 ? b : a, q = OP == 3 ? c : b;
        const float pl = s0 ? p.y : p.x, ql = s1 ? q.y : q.x, ph = h0 ? p.y : p.x, qh = h1 ? q.y : q.x;
        float wlo, whi;
        if (OP == 0) { wlo = pl * ql; whi = ph * qh; }
        else if (OP == 1) { wlo = pl + ql; whi = ph + qh; }
        else if (OP == 2) { wlo = pl * ql + c.x; whi = ph * qh + c.y; }
        else { wlo = a.x * pl + ql; whi = a.y * ph + qh; }
        asm volatile("" : "+v"(wlo), "+v"(whi));          // (small integers: every result is exact however it is rounded or fused)
#define SEL2 " op_sel:[%c4,%c5] op_sel_hi:[%c6,%c7]"
        if (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" SEL2 : "=&v"(d) : "v"(a), "v"(b), "v"(c), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
        else if (OP == 1) asm volatile("v_pk_add_f32 %0, %1, %2" SEL2 : "=&v"(d) : "v"(a), "v"(b), "v"(c), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
        else if (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[%c4,%c5,0] op_sel_hi:[%c6,%c7,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,%c4,%c5] op_sel_hi:[1,%c6,%c7]" : "=&v"(d) : "v"(a), "v"(b), "v"(c), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
#undef SEL2
        const bool blo = d.x != wlo, bhi = d.y != whi;
        if (blo || bhi) {
            nerr++; nlo += blo; nq[lane >> 4]++;
            if (!first) first = ((unsigned long long)__float_as_uint(blo ? d.x : d.y) << 32) | __float_as_uint(blo ? wlo : whi);
        }
    }
    if (nerr) {
        atomicAdd(&out[0], (unsigned long long)nerr);
        for (int k = 0; k < 4; k++) atomicAdd(&out[1 + k], (unsigned long long)nq[k]);
        atomicAdd(&out[5], (unsigned long long)nlo);
        atomicCAS(&out[6], 0ull, first);
    }
}
template <int OP, int M>
static void launch_sweep(int m, int nblocks, hipStream_t s, int iters, const float* tab, unsigned long long* dev) {
    if (m == M) hipLaunchKernelGGL(HIP_KERNEL_NAME(pk_sweep_kernel<OP, M>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    if constexpr (M > 0) launch_sweep<OP, M - 1>(m, nblocks, s, iters, tab, dev);
}

// mode 200 + 16*cls + m (round 4): is the fault a property of the OPERAND SELECT or of the packed-fp32 opcodes?  The same sweep of
// the two swept sources' selects (m as above) for the other VOP3P classes gfx950 has:
//   cls 0: v_pk_mov_b32    (64-bit register pairs, like packed fp32; which source register lands in which half of the result is
//                           calibrated on an idle GPU first -- no semantics assumed: the inputs are four distinct tags per lane)
//   cls 1: v_pk_add_u16    (32-bit registers; op_sel picks the 16-bit half that feeds the LOW result)
//   cls 2: v_pk_fma_f16    (the same with a third source taken straight; small integers: exact in fp16)
template <int CLS, int M>
__global__ __launch_bounds__(256) void pk_class_kernel(int iters, const float* __restrict__ tab, unsigned long long* __restrict__ out,
                                                       int map_lo, int map_hi, int calibrate) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    constexpr int s0 = M & 1, s1 = (M >> 1) & 1, h0 = (M >> 2) & 1, h1 = (M >> 3) & 1;
    unsigned nerr = 0, nlo = 0, nq[4] = {0, 0, 0, 0};
    unsigned long long first = 0;
    for (int it = 0; it < iters; it++) {
        const int base = ((gw * 131 + it * 17) & 1023) * 64;
        // small integers from memory, as in the kernel the fault was found in
        const int ia = (int)tab[2 * ((base + lane) & 65535)], ib = (int)tab[2 * ((base + lane + 7) & 65535) + 1];
        const int ic = (int)tab[2 * ((base + lane + 19) & 65535)];
        unsigned wlo = 0, whi = 0, glo = 0, ghi = 0;
        if (CLS == 0) {
            const unsigned t = (unsigned)(base + lane) * 8u;
            const v2u a = {t + 1u, t + 2u}, b = {t + 3u, t + 4u};
            v2u d;
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[%c3,%c4] op_sel_hi:[%c5,%c6]" : "=&v"(d) : "v"(a), "v"(b), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
            glo = d.x - t; ghi = d.y - t;                       // 1..4: which source register it is
            if (calibrate) { wlo = glo; whi = ghi; if (gw == 0 && lane == 0 && it == 0) out[7] = (unsigned long long)glo | ((unsigned long long)ghi << 8); }
            else { wlo = (unsigned)map_lo; whi = (unsigned)map_hi; }
        } else if (CLS == 1) {
            const unsigned a = ((unsigned)(ia + 16) & 0xFFFFu) | ((unsigned)(ib + 300) << 16), b = ((unsigned)(ic + 16) & 0xFFFFu) | ((unsigned)(ia + 700) << 16);
            unsigned d;
            asm volatile("v_pk_add_u16 %0, %1, %2 op_sel:[%c3,%c4] op_sel_hi:[%c5,%c6]" : "=&v"(d) : "v"(a), "v"(b), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
            const unsigned ah[2] = {a & 0xFFFFu, a >> 16}, bh[2] = {b & 0xFFFFu, b >> 16};
            wlo = (ah[s0] + bh[s1]) & 0xFFFFu; whi = (ah[h0] + bh[h1]) & 0xFFFFu;
            glo = d & 0xFFFFu; ghi = d >> 16;
        } else {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const float af[2] = {(float)(ia & 3) - 2.f, (float)(ib & 3) + 1.f}, bf[2] = {(float)(ic & 3) - 1.f, (float)(ia & 7) - 3.f};
            const float cf[2] = {(float)(ib & 7), (float)(ic & 7) - 4.f};
            const h2 a = {(_Float16)af[0], (_Float16)af[1]}, b = {(_Float16)bf[0], (_Float16)bf[1]}, c = {(_Float16)cf[0], (_Float16)cf[1]};
            h2 d;
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[%c4,%c5,0] op_sel_hi:[%c6,%c7,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c), "n"(s0), "n"(s1), "n"(h0), "n"(h1));
            const h2 w = {(_Float16)(af[s0] * bf[s1] + cf[0]), (_Float16)(af[h0] * bf[h1] + cf[1])};
            wlo = __builtin_bit_cast(unsigned short, w.x); whi = __builtin_bit_cast(unsigned short, w.y);
            glo = __builtin_bit_cast(unsigned short, d.x); ghi = __builtin_bit_cast(unsigned short, d.y);
        }
        const bool blo = glo != wlo, bhi = ghi != whi;
        if (blo || bhi) {
            nerr++; nlo += blo; nq[lane >> 4]++;
            if (!first) first = ((unsigned long long)(blo ? glo : ghi) << 32) | (blo ? wlo : whi);
        }
    }
    if (nerr) {
        atomicAdd(&out[0], (unsigned long long)nerr);
        for (int k = 0; k < 4; k++) atomicAdd(&out[1 + k], (unsigned long long)nq[k]);
        atomicAdd(&out[5], (unsigned long long)nlo);
        atomicCAS(&out[6], 0ull, first);
    }
}
template <int CLS, int M>
static void launch_class(int m, int nblocks, hipStream_t s, int iters, const float* tab, unsigned long long* dev, int ml, int mh, int cal) {
    if (m == M) hipLaunchKernelGGL(HIP_KERNEL_NAME(pk_class_kernel<CLS, M>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev, ml, mh, cal);
    if constexpr (M > 0) launch_class<CLS, M - 1>(m, nblocks, s, iters, tab, dev, ml, mh, cal);
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
    else if (mode == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::bperm_probe_pk_kernel<3>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    else if (mode == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::pkmul_probe_kernel<4>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    else if (mode == 5) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::pkmul_probe_kernel<5>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    else if (mode == 6) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::pkmul_probe_kernel<6>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    else if (mode == 7) hipLaunchKernelGGL(HIP_KERNEL_NAME(xeng::pkmul_probe_kernel<7>), dim3(nblocks), dim3(256), 0, s, iters, tab, dev);
    else if (mode >= 100 && mode < 164) {
        const int op = (mode - 100) >> 4, m = (mode - 100) & 15;
        if (op == 0) xeng::launch_sweep<0, 15>(m, nblocks, s, iters, tab, dev);
        else if (op == 1) xeng::launch_sweep<1, 15>(m, nblocks, s, iters, tab, dev);
        else if (op == 2) xeng::launch_sweep<2, 15>(m, nblocks, s, iters, tab, dev);
        else xeng::launch_sweep<3, 15>(m, nblocks, s, iters, tab, dev);
    } else if (mode >= 200 && mode < 248) {
        // host8[0] on entry: (map_lo | map_hi << 8 | calibrate << 16) for the v_pk_mov_b32 class
        const int cls = (mode - 200) >> 4, m = (mode - 200) & 15;
        const int ml = (int)(host8[0] & 0xFF), mh = (int)((host8[0] >> 8) & 0xFF), cal = (int)((host8[0] >> 16) & 1);
        if (cls == 0) xeng::launch_class<0, 15>(m, nblocks, s, iters, tab, dev, ml, mh, cal);
        else if (cls == 1) xeng::launch_class<1, 15>(m, nblocks, s, iters, tab, dev, ml, mh, cal);
        else xeng::launch_class<2, 15>(m, nblocks, s, iters, tab, dev, ml, mh, cal);
    } else XENG_FAIL(XENG_STATUS_INVALID_ARGUMENT, "probe: unknown mode %d", mode);
    XENG_HIP(hipGetLastError());
    XENG_HIP(hipMemcpyAsync(host8, dev, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    XENG_HIP(hipStreamSynchronize(s));
    return XENG_STATUS_SUCCESS;
}
