// EXPERIMENT (round 5; built only with -DXENG_EXPERIMENTS, selected with XENG_KLOOP=16 in a -DXENG_DIAGNOSTICS build):
// the K loop of xcorr_fused_kernel on v_mfma_i32_16x16x64_i8 with EIGHT waves per work-group -- two per SIMD, wave tiles of
// 32 x 64 inputs (two 32x32 cells = eight 16x16 sub-cells, 96 accumulator registers) -- inside the REAL kernel: the same
// persistent XCD-aware grid, item lists, tile groups, LDS image and swizzle, LDS-DMA stream running ahead across items, and the
// same epilogue traffic.  TIMING ONLY: the accumulators are stored through the shipped epilogue in the wrong register order, all
// waves use the 2x2 wiring (Z waves contract their off-diagonal neighbours), and the last stage of an item re-reads 32 of its
// rows instead of a zero band -- the instruction mix, the LDS / L2 / HBM traffic and the store pattern are those a correct
// kernel would have, the values are not.  The round-2 prototype of this loop alone (profiles/microbench/kloop_proto.hip)
// measured 3281-3294 TOP/s against 3055-3183 for the shipped shape; this file asks the same question of the whole kernel
// (round-4 review, item 6).
//
// What changes against xcorr_fused_kernel:
//   * a K-tile is 64 samples: lane group q = lane >> 4 of a transposing read takes the 16-sample band q, so the four bands of a
//     K-tile may lie in different stages of the ring.  Stages stay 96 samples (480 = 5 x 96; 64 does not divide a gulp): per
//     PAIR of stages three K-tiles -- A = rows 0-63 of stage S; B = rows 64-95 of S | rows 0-31 of S+1; C = rows 32-95 of S+1;
//   * per K-tile and wave 6 operand fragments of 16 inputs x 64 samples (X0 X1 | Y0 Y1 | Z0 Z1: cells X x Y and X x Z), 12
//     transposing reads, 72 mask / shift VALU, 32 MFMAs of 16 cycles: 2-2-2-3 VALU behind the MFMAs, the reads of K-tile g+1 in
//     the first half of K-tile g's MFMAs and its unpack in the second (one tile of look-ahead = 64 samples, as today's two of 32);
//   * the LDS-DMA of a stage is three 1 KiB pieces per wave (24 rows of one block pair), one M0 write per stage; two barriers per
//     pair of stages, i.e. one per 96 samples as today.
#pragma once
#ifndef XF16_SCHED
#define XF16_SCHED 0
#endif

template <int ABL>
__global__ __launch_bounds__(512, 1) void xcorr_fused16_kernel(XcorrParams p) {
    constexpr int SLOT_BYTES = XC_KT * KT_BYTES;       // 96 rows x 64 B: one 64-input block of a stage
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES; // 24 KiB
    constexpr int NLOAD = 3;                           // 1 KiB LDS-DMA pieces per wave per stage
    constexpr int RING = 4;
    constexpr int NST = 16;                            // stores of one epilogue (2 cells x 4 row groups x 2 planes)
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

    // ---- issue side: wave W brings rows 24 * (W & 3) .. + 23 of block pair W >> 2
    int is_k = 0, is_c = 0, is_g = 0, is_sl = 0, is_issued = 0;
    uint32_t is_voff = 0;                    // piece n of a stage: is_voff + n * (8 rows - 1 KiB): one register for the three
    const uint8_t* is_stage = nullptr;
    const uint32_t piece_step = 8u * row_stride - 1024u;
    auto is_setup = [&](const Item& it) {
        is_c = it.c;
        const uint32_t slots = groups[it.wg * 8];
        const int chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);
        const int pr = wave >> 2;
        const int blk0 = (slots >> (16 * pr)) & 0xFF, blk1 = (slots >> (16 * pr + 8)) & 0xFF;
        const uint32_t col = (uint32_t)((chunk >> 2) ? blk1 : blk0) * 64u + (uint32_t)(chunk & 3) * 16u;
        const uint32_t lane_off = (uint32_t)(lane >> 3) * row_stride + (col + 16u <= (uint32_t)p.ninput ? col : 0u);
        is_voff = lane_off;
        is_g = 0; is_sl = 0; is_issued = 0;
    };
    auto next_stage = [&]() {
        if (is_issued == p.nstage) {
            Item nx;
            if (!item(is_k + 1, nx)) return;
            is_k++;
            is_setup(nx);
        }
        is_stage = p.gulps[is_g] + ((size_t)(is_sl * (XC_KT * 32)) * p.nchan + is_c) * (size_t)p.ninput;
        is_issued++;
        if (++is_sl == p.spg) { is_sl = 0; is_g++; }
    };
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds);
    auto issue_stage = [&](int ring_slot) {
        const uint8_t* sb = is_stage + (size_t)(24 * (wave & 3)) * row_stride;
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
    if (!item(0, it)) return;
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
    int stores_in_flight = 0;               // (wave-uniform)
    const v4i M = (v4i)(0xF0F0F0F0);

    for (int k = 0; item(k, it); k++) {
        const int c = it.c, wg = it.wg;
        const uint32_t slots = groups[wg * 8], ww = groups[wg * 8 + 1 + (wave >> 1)];
        const int npair = p.nstage >> 1;
        const bool tail = (p.nstage & 1) != 0;
        if (!(ww & FRAG_BUSY)) {
            // a wave pair without a cell keeps the stage stream and the barriers going
            for (int pr = 0; pr < npair; pr++) {
                wait_vmcnt<NLOAD>(); __builtin_amdgcn_s_barrier();
                next_stage(); issue_stage(slot_of(0));
                wait_vmcnt<NLOAD>(); __builtin_amdgcn_s_barrier();
                next_stage(); issue_stage(slot_of(1));
                rs = slot_of(2);
            }
            if (tail) {
                wait_vmcnt<NLOAD>(); __builtin_amdgcn_s_barrier();
                next_stage(); issue_stage(slot_of(0));
                rs = slot_of(1);
            }
            stores_in_flight = 0;
            continue;
        }
        // operand fragments: X = the wave's row fragment (32 inputs), Y and Z its two column fragments; 16-input halves h
        int off[3], b32[3];                  // (the fragment's second 16 inputs: off ^ 16)
        {
            const int posq[3] = {(int)((ww >> (3 * (wave & 1))) & 7), (int)((ww >> 6) & 7), (int)((ww >> 9) & 7)};
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const int pos = posq[f];
                const int cp = 4 * ((pos >> 1) & 1) + 2 * (pos & 1);        // chunk position of the fragment's first 16 inputs
                off[f] = (pos >> 2) * (2 * SLOT_BYTES) + (tr16 ^ (cp * 16));
                b32[f] = (int)((slots >> (8 * (pos >> 1))) & 0xFF) * 2 + (pos & 1);
            }
        }
        const int row[4] = {b32[0], b32[0], b32[0], b32[0]}, col[4] = {b32[1], b32[2], b32[1], b32[2]};

        v4i accR[8], accP[8], accQ[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { accR[q] = (v4i)(0); accP[q] = (v4i)(0); accQ[q] = (v4i)(0); }

        struct Ops { v4i r[6], i[6]; };          // unpacked operands of one K-tile: X0 X1 Y0 Y1 Z0 Z1
        // the 12 transposing reads of a K-tile whose band q starts at LDS byte address `base` (per lane: bands may lie in different stages)
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
        auto mfma_tile = [&](const Ops& o) {
#pragma unroll
            for (int cc = 0; cc < 2; cc++)           // cell X x Y, then X x Z
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int q = cc * 4 + i * 2 + j, y = 2 + 2 * cc + j;
                        accR[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.r[i], o.r[y], accR[q], 0, 0, 0);
                        accP[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.i[i], o.r[y], accP[q], 0, 0, 0);
                        accQ[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.r[i], o.i[y], accQ[q], 0, 0, 0);
                        accR[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(o.i[i], o.i[y], accR[q], 0, 0, 0);
                    }
        };
        // One K-tile period of a wave: the 12 transposing reads of the NEXT K-tile first, the 32 MFMAs of this one, then the 72 mask /
        // shift VALU that turn the next tile into operands IN PLACE (the MFMAs have read the old values: one operand set, 48
        // registers; 96 accumulators + 48 + 24 raw stay far below the 256 a wave has at two waves per SIMD).  Nothing inside one wave
        // overlaps the unpack with the MFMAs: the OTHER wave of the SIMD does -- while this one masks and shifts, that one multiplies.
        // SCHED 1 (XF16_SCHED): instead pin 2-2-2-3 VALU behind the MFMAs (the compiler then needs a second operand set).
        auto stage_wait = [&]() {
            if (__builtin_amdgcn_readfirstlane(stores_in_flight)) wait_vmcnt<NLOAD + NST>(); else wait_vmcnt<NLOAD>();
            __builtin_amdgcn_s_barrier();
        };
        auto fence = [&]() {
#if XF16_SCHED == 0
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        auto pin = [&]() {
#if XF16_SCHED == 1
#pragma unroll
            for (int i = 0; i < 32; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
#endif
        };

        Ops cur;
        v4i raw[6];
        read_tile((uint32_t)(slot_of(0) * STAGE_BYTES), raw);          // K-tile A of the item's first stage
        unpack(raw, cur);
        for (int pr = 0; pr < npair; pr++) {
            const uint32_t b0 = (uint32_t)(slot_of(0) * STAGE_BYTES), b1 = (uint32_t)(slot_of(1) * STAGE_BYTES), b2 = (uint32_t)(slot_of(2) * STAGE_BYTES);
            // period A: MFMA(A) || read B (bands 0, 1: rows 64.. of stage S; bands 2, 3: rows 0.. of stage S+1)
            read_tile(band < 2 ? b0 + 64 * 128 : b1 - 32 * 128, raw);
            fence();
            mfma_tile(cur);
            fence();
            unpack(raw, cur);
            pin();
            stage_wait();                      // stage S+2 has landed; everybody is done reading stage S
            // period B: MFMA(B) || read C (rows 32..95 of stage S+1) || DMA of stage S+4 into S's slot
            next_stage();
            issue_stage(slot_of(0));
            read_tile(b1 + 32 * 128, raw);
            fence();
            mfma_tile(cur);
            fence();
            unpack(raw, cur);
            pin();
            stage_wait();                      // stage S+3 has landed; everybody is done reading stage S+1
            stores_in_flight = 0;
            // period C: MFMA(C) || read the next K-tile A (rows 0..63 of stage S+2) || DMA of stage S+5 into (S+1)'s slot
            next_stage();
            issue_stage(slot_of(1));
            read_tile(b2, raw);
            fence();
            mfma_tile(cur);
            fence();
            unpack(raw, cur);
            pin();
            rs = slot_of(2);
        }
        if (tail) {
            // the odd last stage: K-tile A, then half a K-tile (timing: rows 32..95 again instead of rows 64..95 + a zero band)
            const uint32_t b0 = (uint32_t)(slot_of(0) * STAGE_BYTES);
            read_tile(b0 + 32 * 128, raw);
            fence();
            mfma_tile(cur);
            fence();
            unpack(raw, cur);
            pin();
            stage_wait();                      // the next item's second stage has landed; everybody is done reading this one
            stores_in_flight = 0;
            next_stage();
            issue_stage(slot_of(0));
            mfma_tile(cur);
            rs = slot_of(1);
        }
        // epilogue: two cells through the shipped store path (values in the wrong register order: timing only)
        v16i eR[2][2], eP[2][2], eQ[2][2];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                eR[0][n][e] = accR[4 * n + (e >> 2)][e & 3];
                eP[0][n][e] = accP[4 * n + (e >> 2)][e & 3];
                eQ[0][n][e] = accQ[4 * n + (e >> 2)][e & 3];
                eR[1][n][e] = 0; eP[1][n][e] = 0; eQ[1][n][e] = 0;
            }
        bool fast = p.accumulate == 0;
#pragma unroll
        for (int q = 0; q < 2; q++) fast = fast && row[q] > col[q] && row[q] * 32 + 32 <= 2 * p.nstand;
        fast = __builtin_amdgcn_readfirstlane((int)fast) != 0;
        if (ABL & 16) {
#pragma unroll
            for (int q = 0; q < 8; q++) asm volatile("" :: "v"(accR[q][0]), "v"(accP[q][1]), "v"(accQ[q][2]));
            stores_in_flight = 0;
        } else {
            int lane_e = lane;                  // (laundered: the epilogue's per-lane constants are not to be hoisted into the K loop's registers)
            asm volatile("" : "+v"(lane_e));
            xcorr_store_cells<false, 1>(p, c, row, col, 3, fast, p.accumulate != 0, lane_e, eR, eP, eQ);
            stores_in_flight = fast ? 1 : 0;
        }
    }
    wait_vmcnt<0>();
}
