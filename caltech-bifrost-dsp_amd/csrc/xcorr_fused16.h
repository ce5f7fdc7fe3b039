// xcorr_fused16_kernel (round 5): the fused contraction with EIGHT waves per work-group on v_mfma_i32_16x16x64_i8.
//
// Same kernel as xcorr_fused_kernel (xcorr_kernels.h) in everything a caller sees -- persistent XCD-aware grid, item lists, tile
// groups, LDS image and swizzle, LDS-DMA stream three stages ahead across items, xGPU-order epilogue, bit-exact results -- with
// another K loop.  Why: the chip grants a matrix-dense stream of 16x16x64 MFMAs 13 % more issue rate than of 32x32x32
// (profiles/r02/r2_mfma_shape.txt: 4530 vs 4011 TOP/s bare), but a 16x16 MFMA holds the issue port half of its 16 cycles, and
// one wave per SIMD cannot feed 64 of them per K-tile beside its 96 mask / shift VALU, its LDS reads and its LDS-DMA
// (EXPERIMENTS 4.2c).  Two waves per SIMD can: while one masks and shifts, the other multiplies.  Measured inside the real
// kernel before it was made exact (timing-only build, one box, four interleaved rounds: profiles/r05/ab_kloop16_timing_only.txt):
// 0.2145 -> 0.2067 ms per streaming step (-3.6 %).
//
//   * wave tile = 32 x 64 inputs: TWO 32x32 cells = eight 16x16 sub-cells x 3 accumulators x 4 registers = 96 registers (a
//     wave has 256 at two per SIMD).  Wave 2w + m of the work-group takes cells 2m, 2m+1 of wave w of the tile group
//     (xcorr_tiling.h FragGroup): from a 2x2 wiring (a_m, b0) (a_m, b1); from a Z wiring (d0, d0) (r, c) | (d1, d0) (d1, d1).
//     Three operand patterns over at most three 32-input fragments X, Y, Z:  P0 (X,Y) (X,Z) . P1 (X,X) (Y,Z) . P2 (X,Y) (X,X);
//     the K loop exists once per pattern, chosen per item by a wave-uniform branch outside the loop.
//   * a K-tile is 64 samples: lane group q = lane >> 4 of a transposing read takes the 16-sample band q of the K-tile, so the four
//     bands may lie in different stages of the ring.  Stages stay 96 samples (a gulp is 5 x 96; 64 does not divide it): per PAIR
//     of stages three K-tiles -- A = rows 0-63 of stage S; B = rows 64-95 of S | rows 0-31 of S+1; C = rows 32-95 of S+1; an odd
//     last stage gives A and a half tile whose bands 2, 3 are zeroed in registers.  Which sample sits in which byte of an operand
//     does not matter (both operands use the same map; the contraction is a sum over k).
//   * per K-tile and wave: 12 transposing reads, 72 mask / shift VALU (pinned 2-2-2-3 behind the MFMAs), 32 MFMAs; the reads of
//     K-tile g+1 go out early in K-tile g, its unpack follows in the same period (one tile = 64 samples of look-ahead).
//   * LDS-DMA: three 1 KiB pieces per wave and stage (24 rows of one block pair), one M0 write per stage; two barriers per pair
//     of stages -- one per 96 samples, as before.
//   * epilogue: the accumulators of a cell's four sub-cells are brought into the register layout of a 32x32 MFMA with two lane
//     swaps per register pair (v_permlane16_swap + v_permlane32_swap) and stored by the SAME code as the four-wave kernel
//     (xcorr_store_cells): order, masks, accumulate-into-stored are shared, not restated.
// Not covered here (the four-wave kernel stays for it): the long accumulator in the epilogue (CorrAcc's opt-in fused mode).
#pragma once

#ifndef XF16_SCHED
#define XF16_SCHED 1          // 1: 2-2-2-3 VALU pinned behind the MFMAs; 2: the compiler's order; 3: reads, MFMA burst, then the unpack (A/B builds)
#endif
#ifndef XF16_PRIO
#define XF16_PRIO 0           // 1: the wave raises its issue priority for the MFMAs of a K-tile (A/B builds)
#endif

// sub-cell accumulators (i, j) -> idx 2 i + j of one 32x32 cell, 16x16 MFMA layout (column = lane & 15, row = 4 (lane >> 4) + reg)
// into the 32x32 MFMA layout (column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): register 4 k + r of lane
// (C, h) is row 8 k + 4 h + r -- sub-cell (k >> 1, C >> 4), lane group 2 (k & 1) + h, register r.  With A, B = register r of
// sub-cells (i, 0), (i, 1) as rows of 16 lanes [A0 A1 A2 A3], [B0 B1 B2 B3]:  k = 2 i wants [A0 B0 A1 B1], k = 2 i + 1 wants
// [A2 B2 A3 B3]: permlane16_swap (A's odd rows <-> B's even rows) then permlane32_swap (A's rows 2, 3 <-> B's rows 0, 1).
__device__ __forceinline__ v16i xcorr_sub16_to_cell32(const v4i (&sub)[4]) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v16i out;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const v2u s1 = __builtin_amdgcn_permlane16_swap((unsigned)sub[2 * i][r], (unsigned)sub[2 * i + 1][r], false, false);
            const v2u s2 = __builtin_amdgcn_permlane32_swap(s1.x, s1.y, false, false);
            out[4 * (2 * i) + r] = (int)s2.x;
            out[4 * (2 * i + 1) + r] = (int)s2.y;
        }
    return out;
}

// DESC: gulps by descriptor (GulpDesc in device memory, slab.h: a gulp may be a slab of F-engine packets read where it lies) -- base
// and strides per gulp, read with scalar loads when the gulp's first stage is set up.
template <int ABL, bool DESC = false>
__global__ __launch_bounds__(512, 1) void xcorr_fused16_kernel(XcorrParams p) {
    constexpr int SLOT_BYTES = XC_KT * KT_BYTES;       // 96 rows x 64 B: one 64-input block of a stage
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES; // 24 KiB
    constexpr int NLOAD = 3;                           // 1 KiB LDS-DMA pieces per wave per stage
    constexpr int RING = 4;
    constexpr int NST = 16;                            // stores of one straight-line epilogue (2 cells x 4 row groups x 2 planes)
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // 0..7
    const uint32_t row_stride = (uint32_t)p.nchan * (uint32_t)p.ninput;
    typedef const __attribute__((address_space(4))) uint32_t* DescPtr;
    const DescPtr groups = (DescPtr)(uintptr_t)p.fgroups;
    const DescPtr work = (DescPtr)(uintptr_t)p.work + (size_t)blockIdx.x * p.maxi;
    struct Item { int c, wg; };
    auto item = [&](int k, Item& it) {
        if (k >= p.maxi) return false;
        const uint32_t w = work[k];
        if (!(w & WORK_VALID)) return false;
        it.c = (int)(w & 0xFFFF); it.wg = (int)((w >> 16) & 0x7FFF);
        return true;
    };

    // ---- issue side: wave W brings rows 24 (W & 3) .. + 23 of block pair W >> 2; the stage stream runs ahead across items
    int is_k = 0, is_c = 0, is_g = 0, is_sl = 0, is_issued = 0;
    uint32_t is_voff = 0;                    // piece n of a stage: is_voff + n * (8 rows - 1 KiB)
    const uint8_t* is_stage = nullptr;
    uint32_t piece_step = 8u * row_stride - 1024u;            // (rows of at least 128 bytes: xengXgpuInitialize)
    // DESC: this lane's 64-input block and byte position inside it (per item), the current gulp's layout (per gulp)
    uint32_t is_blksub = 0, d_t = row_stride, d_c = (uint32_t)p.ninput;       // (block << 8 | byte position: one register)
    const uint8_t* d_base = nullptr;
    const DescPtr gdesc = (DescPtr)(uintptr_t)p.gdesc;
    auto load_desc = [&](int g) {
        const uint32_t lo = gdesc[g * 8], hi = gdesc[g * 8 + 1];
        d_base = (const uint8_t*)(((uint64_t)hi << 32) | lo);
        d_t = gdesc[g * 8 + 2];
        d_c = gdesc[g * 8 + 3];
        const uint32_t d_b = gdesc[g * 8 + 4];
        is_voff = (uint32_t)(lane >> 3) * d_t + (is_blksub >> 8) * d_b + (is_blksub & 0xFFu);
        piece_step = 8u * d_t - 1024u;
    };
    auto is_setup = [&](const Item& it) {
        is_c = it.c;
        const uint32_t slots = groups[it.wg * 8];
        const int chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);          // source chunk 0..7 of the 128-byte pair row
        const int pr = wave >> 2;
        const int blk0 = (slots >> (16 * pr)) & 0xFF, blk1 = (slots >> (16 * pr + 8)) & 0xFF;
        // (columns past ninput in the last block: any valid bytes of the row; their products are never stored)
        if (DESC) {
            const uint32_t blk = (uint32_t)((chunk >> 2) ? blk1 : blk0), sub = (uint32_t)(chunk & 3) * 16u;
            is_blksub = blk * 64u + sub + 16u > (uint32_t)p.ninput ? 0u : (blk << 8) | sub;
            load_desc(0);
        } else {
            const uint32_t col = (uint32_t)((chunk >> 2) ? blk1 : blk0) * 64u + (uint32_t)(chunk & 3) * 16u;
            is_voff = (uint32_t)(lane >> 3) * row_stride + (col + 16u <= (uint32_t)p.ninput ? col : 0u);
        }
        is_g = 0; is_sl = 0; is_issued = 0;
    };
    auto next_stage = [&]() {
        if (is_issued == p.nstage) {         // this item is fully issued: go on with the next one, if any
            Item nx;
            if (!item(is_k + 1, nx)) return;   // past the end: keep re-reading the last stage (never consumed)
            is_k++;
            is_setup(nx);
        }
        if (DESC) {
            // (a new gulp may be laid out differently: its descriptor is loaded when its first stage is set up; the stage before it has
            // been issued in full -- a stage's three pieces go out together)
            if (is_sl == 0 && is_g > 0) load_desc(is_g);
            is_stage = d_base + (size_t)(is_sl * (XC_KT * 32)) * d_t + (size_t)is_c * d_c;
        } else
        is_stage = p.gulps[is_g] + ((size_t)(is_sl * (XC_KT * 32)) * p.nchan + is_c) * (size_t)p.ninput;
        is_issued++;
        if (++is_sl == p.spg) { is_sl = 0; is_g++; }
    };
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds);
    // (issued from asm, M0 written without saving it, as in xcorr_fused_kernel: nothing else in this kernel uses M0)
    auto issue_stage = [&](int ring_slot) {
        const uint8_t* sb = is_stage + (size_t)(24 * (wave & 3)) * (DESC ? d_t : row_stride);
        const uint32_t la = __builtin_amdgcn_readfirstlane(lds_base + ring_slot * STAGE_BYTES + (wave >> 2) * (2 * SLOT_BYTES) + (wave & 3) * 3072);
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:2048"
                     :: "v"(is_voff), "v"(is_voff + piece_step), "v"(is_voff + 2u * piece_step), "s"(sb), "s"(la) : "memory");
    };

    // read side: lane 16 q + 2 r + pp addresses row 16 q + r (+ 8 for the second read) of the K-tile's band q, chunk position
    // (4 * slot parity + 2 * 32-input half + 16-input half) ^ 2 * ((r >> 1) & 3), bytes 8 pp ..
    const int band = lane >> 4;
    const int tr16 = (16 * band + ((lane & 15) >> 1)) * 128 + ((((lane >> 2) & 3) << 1)) * 16 + (lane & 1) * 8;

    Item it;
    if (!item(0, it)) return;                // (the host never launches work-groups without work)
    is_setup(it);
#pragma unroll
    for (int st = 0; st < 4; st++) {
        next_stage();
        issue_stage(st);
    }
    wait_vmcnt<2 * NLOAD>();                 // stages 0 and 1 have landed
    __builtin_amdgcn_s_barrier();

    int rs = 0;                              // ring slot of the item's current stage (the stage counter is continuous across items)
    auto slot_of = [&](int d) { return (rs + d) & (RING - 1); };
    int stores_in_flight = 0;                // (wave-uniform) the previous item of this wave ended with the 16-store epilogue
    const v4i M = (v4i)(0xF0F0F0F0);

    for (int k = 0; item(k, it); k++) {
        const int c = it.c, wg = it.wg;
        const uint32_t slots = groups[wg * 8], ww = groups[wg * 8 + 1 + (wave >> 1)];
        const int npair = p.nstage >> 1;
        const bool tail = (p.nstage & 1) != 0;
        auto stage_wait = [&]() {
            // (behind a straight-line epilogue its 16 stores sit between the LDS-DMA this wait is for and the pieces issued since;
            // vmcnt retires in issue order, so letting them stay in flight does not let that stage slip: xcorr_fused_kernel)
            if (__builtin_amdgcn_readfirstlane(stores_in_flight)) wait_vmcnt<NLOAD + NST>(); else wait_vmcnt<NLOAD>();
            __builtin_amdgcn_s_barrier();
        };
        if (!(ww & FRAG_BUSY)) {
            // a wave pair without a cell keeps the stage stream and the barriers going
            stores_in_flight = 0;
            for (int pr = 0; pr < npair; pr++) {
                stage_wait();
                next_stage(); issue_stage(slot_of(0));
                stage_wait();
                next_stage(); issue_stage(slot_of(1));
                rs = slot_of(2);
            }
            if (tail) {
                stage_wait();
                next_stage(); issue_stage(slot_of(0));
                rs = slot_of(1);
            }
            continue;
        }
        // this wave's two cells and the pattern of its operand fragments (header)
        const bool zpat = (ww & FRAG_Z) != 0;
        const int m = wave & 1;
        const int live = (int)((ww >> (16 + 2 * m)) & 3);
        const int pat = __builtin_amdgcn_readfirstlane(!zpat ? 0 : (m == 0 ? 1 : 2));
        int off[3], b32[3];                  // (a fragment's second 16 inputs: off ^ 16)
        {
            const int pos4[4] = {(int)(ww & 7), (int)((ww >> 3) & 7), (int)((ww >> 6) & 7), (int)((ww >> 9) & 7)};
            //                    2x2: X = a_m, Y = b0, Z = b1      Z, m = 0: X = d0, Y = r, Z = c      Z, m = 1: X = d1, Y = d0, (Z = d0)
            const int posq[3] = {pos4[m], zpat && m ? pos4[0] : pos4[2], zpat && m ? pos4[0] : pos4[3]};
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const int pos = posq[f];
                const int cp = 4 * ((pos >> 1) & 1) + 2 * (pos & 1);        // chunk position of the fragment's first 16 inputs
                off[f] = (pos >> 2) * (2 * SLOT_BYTES) + (tr16 ^ (cp * 16));
                b32[f] = (int)((slots >> (8 * (pos >> 1))) & 0xFF) * 2 + (pos & 1);
            }
        }

        v4i accR[8], accP[8], accQ[8];       // sub-cell q = 4 cell + 2 i + j
#pragma unroll
        for (int q = 0; q < 8; q++) { accR[q] = (v4i)(0); accP[q] = (v4i)(0); accQ[q] = (v4i)(0); }

        struct Ops { v4i r[6], i[6]; };      // unpacked operands of one K-tile: X0 X1 Y0 Y1 Z0 Z1 (16 x re, 16 x im)
        // the 12 transposing reads of a K-tile whose band q starts at LDS byte `base` (per lane: bands may lie in different stages)
        auto read_tile = [&](uint32_t base, v4i (&raw)[6]) {
#pragma unroll
            for (int f = 0; f < 3; f++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t o = base + (uint32_t)(off[f] ^ (h * 16));
                    const v2i a0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + o));
                    const v2i a1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + o + 1024));
                    raw[2 * f + h] = (v4i){a0.x, a0.y, a1.x, a1.y};
                }
        };
        auto unpack = [&](const v4i (&raw)[6], Ops& o) {
#pragma unroll
            for (int q = 0; q < 6; q++) { o.r[q] = raw[q] & M; o.i[q] = (raw[q] << 4) & M; }
        };
        // the 32 MFMAs of one K-tile: per sub-cell R += xr*yr + xi*yi, P += xi*yr, Q += xr*yi (no negated operand: -(-8) x 16 overflows)
        auto mfma_tile = [&](const Ops& o, auto patc) {
            constexpr int PAT = decltype(patc)::value;
#if XF16_PRIO
            __builtin_amdgcn_s_setprio(2);
#endif
            constexpr int RA[3][2] = {{0, 0}, {0, 1}, {0, 0}}, CB[3][2] = {{1, 2}, {0, 2}, {1, 0}};      // fragment of the cell's rows / columns
#pragma unroll
            for (int cc = 0; cc < 2; cc++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int q = cc * 4 + i * 2 + j, x = 2 * RA[PAT][cc] + i, y = 2 * CB[PAT][cc] + j;
                        accR[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.r[x], o.r[y], accR[q], 0, 0, 0);
                        accP[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.i[x], o.r[y], accP[q], 0, 0, 0);
                        accQ[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.r[x], o.i[y], accQ[q], 0, 0, 0);
                        accR[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.i[x], o.i[y], accR[q], 0, 0, 0);
                    }
#if XF16_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        auto pin = [&]() {
#if XF16_SCHED == 1
#pragma unroll
            for (int i = 0; i < 32; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // MFMA
                if (i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
                if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); // VALU 2-2-2-3
                else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
#elif XF16_SCHED == 3
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                      // the reads of the next K-tile
            __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);                      // this K-tile's MFMAs back to back (the SIMD's other wave unpacks meanwhile)
            __builtin_amdgcn_sched_group_barrier(0x002, 80, 0);                      // then the unpack
#endif
        };

        auto kloop = [&](auto patc) {
            Ops cur;
            v4i raw[6];
            read_tile((uint32_t)(slot_of(0) * STAGE_BYTES), raw);          // K-tile A of the item's first stage
            unpack(raw, cur);
            for (int pr = 0; pr < npair; pr++) {
                const uint32_t b0 = (uint32_t)(slot_of(0) * STAGE_BYTES), b1 = (uint32_t)(slot_of(1) * STAGE_BYTES), b2 = (uint32_t)(slot_of(2) * STAGE_BYTES);
                // period A: MFMA(A) || read + unpack B (bands 0, 1: rows 64.. of stage S; bands 2, 3: rows 0.. of stage S+1)
                read_tile(band < 2 ? b0 + 64 * 128 : b1 - 32 * 128, raw);
                mfma_tile(cur, patc);
                unpack(raw, cur);
                pin();
                stage_wait();                  // stage S+2 has landed; everybody is done reading stage S
                // period B: MFMA(B) || read + unpack C (rows 32..95 of stage S+1) || LDS-DMA of stage S+4 into S's slot
                next_stage();
                issue_stage(slot_of(0));
                read_tile(b1 + 32 * 128, raw);
                mfma_tile(cur, patc);
                unpack(raw, cur);
                pin();
                stage_wait();                  // stage S+3 has landed; everybody is done reading stage S+1
                stores_in_flight = 0;
                // period C: MFMA(C) || read + unpack the next K-tile A (rows 0..63 of stage S+2) || LDS-DMA of stage S+5 into (S+1)'s slot
                next_stage();
                issue_stage(slot_of(1));
                read_tile(b2, raw);
                mfma_tile(cur, patc);
                unpack(raw, cur);
                pin();
                rs = slot_of(2);
            }
            if (tail) {
                // the odd last stage: K-tile A (in `cur`), then half a K-tile: bands 0, 1 = rows 64..95, bands 2, 3 = nothing
                const uint32_t b0 = (uint32_t)(slot_of(0) * STAGE_BYTES);
                read_tile(band < 2 ? b0 + 64 * 128 : b0, raw);
                mfma_tile(cur, patc);
#pragma unroll
                for (int q = 0; q < 6; q++) raw[q] = band < 2 ? raw[q] : (v4i)(0);
                unpack(raw, cur);
                pin();
                stage_wait();                  // the next item's second stage has landed; everybody is done reading this one
                stores_in_flight = 0;
                next_stage();
                issue_stage(slot_of(0));
                mfma_tile(cur, patc);
                rs = slot_of(1);
            }
        };
        // diagnostic (XENG_DBG_STAMPS=1; null in production): shader clock vs the 100 MHz reference, per item and wave PAIR (the even wave
        // of a pair reports: the buffer has four slots per item, as for the four-wave kernel)
        unsigned long long t_start = 0, r_start = 0;
        const unsigned long long r_entry = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (p.stamps) { t_start = __builtin_amdgcn_s_memtime(); r_start = __builtin_amdgcn_s_memrealtime(); }
        if (pat == 0) kloop(std::integral_constant<int, 0>{});
        else if (pat == 1) kloop(std::integral_constant<int, 1>{});
        else kloop(std::integral_constant<int, 2>{});
        if (p.stamps) {
            asm volatile("" :: "v"(accR[7][3]), "v"(accQ[7][3]), "v"(accP[7][3]));       // (the last MFMAs have retired)
            const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && m == 0) {
                unsigned long long* o = p.stamps + ((size_t)(c * p.nwg + wg) * 4 + (wave >> 1)) * 8;
                o[0] = t_end - t_start; o[1] = r_end - r_start; o[2] = r_start; o[3] = r_end; o[4] = r_entry;
            }
        }

        // ---- epilogue: the two cells in the 32x32 register layout, through the shared store path
        int row[4], col[4];
        {
            const int ra0 = 0, ra1 = pat == 1 ? 1 : 0, cb0 = pat == 1 ? 0 : 1, cb1 = pat == 2 ? 0 : 2;
            row[0] = b32[ra0]; col[0] = b32[cb0]; row[1] = b32[ra1]; col[1] = b32[cb1];
            row[2] = row[3] = col[2] = col[3] = 0;
        }
        bool fast = live == 3 && p.accumulate == 0;
#pragma unroll
        for (int q = 0; q < 2; q++) fast = fast && row[q] > col[q] && row[q] * 32 + 32 <= 2 * p.nstand;
        fast = __builtin_amdgcn_readfirstlane((int)fast) != 0;
        if (ABL & 16) {       // timing only: no epilogue (keep the accumulators live)
#pragma unroll
            for (int q = 0; q < 8; q++) asm volatile("" :: "v"(accR[q][0]), "v"(accP[q][1]), "v"(accQ[q][2]));
            stores_in_flight = 0;
        } else {
            v16i eR[2][2], eP[2][2], eQ[2][2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const v4i sr[4] = {accR[4 * n], accR[4 * n + 1], accR[4 * n + 2], accR[4 * n + 3]};
                const v4i sp[4] = {accP[4 * n], accP[4 * n + 1], accP[4 * n + 2], accP[4 * n + 3]};
                const v4i sq[4] = {accQ[4 * n], accQ[4 * n + 1], accQ[4 * n + 2], accQ[4 * n + 3]};
                eR[0][n] = xcorr_sub16_to_cell32(sr);
                eP[0][n] = xcorr_sub16_to_cell32(sp);
                eQ[0][n] = xcorr_sub16_to_cell32(sq);
                eR[1][n] = eP[1][n] = eQ[1][n] = (v16i)(0);
            }
            int lane_e = lane;                  // (laundered: the epilogue's per-lane constants are not to be hoisted into the K loop's registers)
            asm volatile("" : "+v"(lane_e));
            xcorr_store_cells<false, 1>(p, c, row, col, live, fast, p.accumulate != 0, lane_e, eR, eP, eQ);
            stores_in_flight = fast ? 1 : 0;
        }
        if (p.stamps && lane == 0 && m == 0) p.stamps[((size_t)(c * p.nwg + wg) * 4 + (wave >> 1)) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    }
    wait_vmcnt<0>();   // no LDS-DMA may still be in flight when the wave ends
}
