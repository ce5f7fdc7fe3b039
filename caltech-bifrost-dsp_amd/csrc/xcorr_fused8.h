// xcorr_fused8_kernel: the fused X-engine contraction with EIGHT waves per work-group (two per SIMD).
//
// Same data path as xcorr_fused_kernel (xcorr_kernels.h: gulps read in place by LDS-DMA, byte-transposing LDS reads,
// persistent work-groups walking host-built item lists, xGPU-order epilogue) and the same work-group tile (up to four
// 64-input blocks staged per 96-sample stage, tile groups of build_wg_descs), but every 64x64 tile of a group is
// split into two 64-row x 32-column wave tiles: wave w contracts column half w>>2 of tile w&3.  Waves w and w+4
// share a SIMD (a work-group's waves are dealt to the SIMDs cyclically), so the two halves of one 64x64 tile sit on
// one SIMD and every SIMD carries the same MFMA work as in the four-wave kernel.
//
// Why: with one wave per SIMD the in-order issue of that wave is the bottleneck, not the matrix pipe -- a 1 KiB
// LDS-DMA piece costs the issuing wave ~76 cycles, the 48 unpack VALU per K-tile ~100, and the next MFMA waits behind
// them (measured 785 cycles per K-tile for 512 cycles of MFMA).  A second wave on the SIMD issues its MFMAs into those
// holes.  The price is 1.5x the LDS reads and unpack VALU per MFMA (A is unpacked by both halves); the accumulators
// (3 x 2 tiles x 16 registers = 96 per wave) leave room for two waves in the 512-register file.
#pragma once
#include "xcorr_kernels.h"

namespace xeng {

struct Frags8 {      // unpacked operands of one 64x32 wave tile and one K-tile
    v4i ar[2], ai[2], br, bi;
};
struct RawFrags8 {   // still packed 4+4 bit
    v4i a[2], b;
};

__device__ __forceinline__ Frags8 unpack_frags8(const RawFrags8& r) {
    const v4i M = (v4i)(0xF0F0F0F0);
    Frags8 u;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        u.ar[m] = r.a[m] & M;
        u.ai[m] = (r.a[m] << 4) & M;
    }
    u.br = r.b & M;
    u.bi = (r.b << 4) & M;
    return u;
}

// the 8 (4 for the upper half of a diagonal tile) MFMAs of one K-tile
__device__ __forceinline__ void xcorr_mfma_tile8(const Frags8& u, bool skip_m0, v16i (&accR)[2][1], v16i (&accP)[2][1],
                                                 v16i (&accQ)[2][1]) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
        if (m == 0 && skip_m0) continue;
        accR[m][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ar[m], u.br, accR[m][0], 0, 0, 0);
        accP[m][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ai[m], u.br, accP[m][0], 0, 0, 0);
        accQ[m][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ar[m], u.bi, accQ[m][0], 0, 0, 0);
        accR[m][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.ai[m], u.bi, accR[m][0], 0, 0, 0);
    }
}

__global__ __launch_bounds__(512, 2) void xcorr_fused8_kernel(XcorrParams p) {
    constexpr int KT_STAGE = XC_KT;
    constexpr int SLOT_BYTES = KT_STAGE * KT_BYTES;          // 6 KiB: one 64-input block x 96 samples
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES;       // 24 KiB
    constexpr int NLOAD = 3;                                 // 1 KiB LDS-DMA pieces per wave per stage (8 waves x 3)
    constexpr int DEPTH = XF_DEPTH;
    constexpr int RING = DEPTH + 1;
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wq = wave & 3, nh = wave >> 2;                 // tile of the group, column half
    const uint32_t row_stride = (uint32_t)p.nchan * (uint32_t)p.ninput;
    typedef const __attribute__((address_space(4))) uint32_t* DescPtr;
    const DescPtr descs = (DescPtr)(uintptr_t)p.descs;
    static_assert(sizeof(WgDesc) == 16, "descriptor layout");

    const DescPtr work = (DescPtr)(uintptr_t)p.work + (size_t)blockIdx.x * p.maxi * 4;
    struct Item { int c, wg, stage0, nst, slice, nslices, chain; };
    auto item = [&](int k, Item& it) {
        if (k >= p.maxi) return false;
        const uint32_t w0 = work[k * 4], w1 = work[k * 4 + 1], w2 = work[k * 4 + 2];
        if (!(w2 >> 16)) return false;
        it.c = (int)(w0 & 0xFFFF); it.wg = (int)(w0 >> 16);
        it.stage0 = (int)(w1 & 0xFFFF); it.nst = (int)(w1 >> 16);
        it.slice = (int)(w2 & 0xFF); it.nslices = (int)((w2 >> 8) & 0xFF);
        it.chain = (int)work[k * 4 + 3];
        return true;
    };

    // ---- issue side (see xcorr_fused_kernel): wave w brings rows 24*(w&3).. (three 8-row pieces) of slot pair w>>2
    int is_k = 0, is_c = 0, is_g = 0, is_sl = 0, is_issued = 0, is_nst = 0;
    uint32_t is_voff[NLOAD] = {};
    const uint8_t* is_stage = nullptr;
    auto is_setup = [&](const Item& it) {
        is_c = it.c; is_nst = it.nst;
        const uint32_t slots = descs[it.wg * 4];
        const int chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);
        const int blk0 = (slots >> (16 * (wave >> 2))) & 0xFF, blk1 = (slots >> (16 * (wave >> 2) + 8)) & 0xFF;
        const uint32_t col = (uint32_t)((chunk >> 2) ? blk1 : blk0) * 64u + (uint32_t)(chunk & 3) * 16u;
        const uint32_t lane_off = (uint32_t)(lane >> 3) * row_stride + (col + 16u <= (uint32_t)p.ninput ? col : 0u);
#pragma unroll
        for (int n = 0; n < NLOAD; n++) is_voff[n] = lane_off + (uint32_t)n * 8u * row_stride - (uint32_t)(n * 1024);
        is_g = it.stage0 / p.spg;
        is_sl = it.stage0 - is_g * p.spg;
        is_issued = 0;
    };
    auto next_stage = [&]() {
        if (is_issued == is_nst) {
            Item nx;
            if (!item(is_k + 1, nx)) return;   // past the end: keep re-reading the last stage (never consumed)
            is_k++;
            is_setup(nx);
        }
        is_stage = p.gulps[is_g] + ((size_t)(is_sl * (KT_STAGE * 32)) * p.nchan + is_c) * (size_t)p.ninput;
        is_issued++;
        if (++is_sl == p.spg) { is_sl = 0; is_g++; }
    };
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds);
    // the three pieces of a stage share one scalar base and one M0 write (immediate offsets advance both sides)
    auto issue_stage = [&](int ring_slot) {
        const uint8_t* sb = is_stage + (size_t)(24 * (wave & 3)) * row_stride;
        const uint32_t la = lds_base + ring_slot * STAGE_BYTES + (wave >> 2) * (2 * SLOT_BYTES) + (wave & 3) * 3072;
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, %3\n\t"
                     "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:2048"
                     :: "v"(is_voff[0]), "v"(is_voff[1]), "v"(is_voff[2]), "s"(sb), "s"(la) : "memory");
    };

    const int tr_off = ((lane >> 5) * 16 + ((lane & 15) >> 1)) * 128 +
                       (((lane >> 4) & 1) ^ (((lane >> 2) & 3) << 1)) * 16 + (lane & 1) * 8;

    Item it;
    if (!item(0, it)) return;
    is_setup(it);
#pragma unroll
    for (int st = 0; st < DEPTH; st++) {
        next_stage();
        issue_stage(st);
    }
    wait_vmcnt<(DEPTH - 2) * NLOAD>();
    __builtin_amdgcn_s_barrier();

    int rs = 0, rs1 = 1, rf = DEPTH;
    auto bump = [&](int& r) { r = (r + 1 == RING) ? 0 : r + 1; };
    for (int k = 0; item(k, it); k++) {
        const int c = it.c, wg = it.wg;
        const uint32_t slots = descs[wg * 4], wa4 = descs[wg * 4 + 1], wb4 = descs[wg * 4 + 2];
        const int a_slot = (wa4 >> (8 * wq)) & 0xFF, b_slot = (wb4 >> (8 * wq)) & 0xFF;
        const bool active = a_slot != 0xFF;
        const int sa = active ? a_slot : 0, sb = active ? b_slot : 0;
        const int blk_a = (slots >> (8 * sa)) & 0xFF, blk_b = (slots >> (8 * sb)) & 0xFF;
        const int a_off = (sa >> 1) * (2 * SLOT_BYTES) + (tr_off ^ ((sa & 1) * 64));
        const int b_off = (sb >> 1) * (2 * SLOT_BYTES) + ((tr_off ^ ((sb & 1) * 64)) ^ (nh * 32));
        const bool diag = __builtin_amdgcn_readfirstlane((int)(blk_a == blk_b)) != 0;
        const bool skip_m0 = diag && nh == 1;      // rows 0-31 x columns 32-63 of a diagonal tile are never stored

        auto load_raw = [&](int ring_slot, int j) {
            const uint8_t* base = lds + ring_slot * STAGE_BYTES + j * (2 * KT_BYTES);
            auto tr = [&](int off) {
                return __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(base + off));
            };
            RawFrags8 r;
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const v2i a0 = tr(a_off ^ (sub * 32)), a1 = tr((a_off ^ (sub * 32)) + 1024);
                r.a[sub] = (v4i){a0.x, a0.y, a1.x, a1.y};
            }
            const v2i b0 = tr(b_off), b1 = tr(b_off + 1024);
            r.b = (v4i){b0.x, b0.y, b1.x, b1.y};
            return r;
        };

        if (!active) {
            // a wave without a tile only keeps the stage stream and the barriers going
            for (int s = 0; s < it.nst; s++) {
                next_stage();
                issue_stage(rf);
                wait_vmcnt<(DEPTH - 2) * NLOAD>();
                __builtin_amdgcn_s_barrier();
                bump(rs); bump(rs1); bump(rf);
            }
            if (it.slice + 1 < it.nslices) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            continue;
        }
        v16i accR[2][1], accP[2][1], accQ[2][1];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            accR[m][0] = (v16i)(0);
            accP[m][0] = (v16i)(0);
            accQ[m][0] = (v16i)(0);
        }

        // software pipeline over K-tiles: MFMA(g) || unpack(g+1) || LDS read(g+2) || LDS-DMA of stage S+DEPTH
        Frags8 cur = unpack_frags8(load_raw(rs, 0));
        RawFrags8 raw = load_raw(rs, 1);
        for (int s = 0; s < it.nst; s++) {
            next_stage();
#pragma unroll
            for (int j = 0; j < KT_STAGE; j++) {
                if (j == 0) issue_stage(rf);
                xcorr_mfma_tile8(cur, skip_m0, accR, accP, accQ);
                cur = unpack_frags8(raw);
                raw = (j + 2 < KT_STAGE) ? load_raw(rs, j + 2) : load_raw(rs1, j + 2 - KT_STAGE);
                // 1 MFMA : 4-5 VALU (the 36 mask/shift ops of the next K-tile under the 8 MFMAs of this one), the six
                // transposing LDS reads behind the first MFMAs
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // MFMA
                    if (i < 4) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);        // VALU
                    else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if (i >= 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
                }
            }
            wait_vmcnt<(DEPTH - 2) * NLOAD>();
            __builtin_amdgcn_s_barrier();
            bump(rs); bump(rs1); bump(rf);
        }
        if (it.slice > 0) {
            const uint32_t target = p.epoch * 16u + (uint32_t)it.slice;
            while (__hip_atomic_load(p.flags + it.chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
                __builtin_amdgcn_s_sleep(4);
            asm volatile("" ::: "memory");
        }
        xcorr_store_tile<1>(p, c, blk_a, blk_b, diag, lane, accR, accP, accQ, it.slice > 0, nh);
        if (it.slice + 1 < it.nslices) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (threadIdx.x == 0)
                __hip_atomic_store(p.flags + it.chain, p.epoch * 16u + (uint32_t)it.slice + 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    wait_vmcnt<0>();
}

}  // namespace xeng
