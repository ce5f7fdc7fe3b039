// FP6 (E3M2) X-engine route: an EXPERIMENT, compiled into libxeng only with -DXENG_EXPERIMENTS (then selected by
// XENG_MFMA=fp6).  Bit-exact, but slower end to end than the int8 path: kept for the tiling work that
// could cash in the 1.55x MFMA rate.  Included by xcorr_kernels.h inside namespace xeng.
#pragma once
// =======================================================================================
// FP6 (E3M2) route -- opt-in (XENG_MFMA=fp6).  Every integer -8..8 is exactly representable in the
// OCP FP6 E3M2 format, and gfx950's block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, scale 2^0)
// runs FP6 at twice the int8 rate with fp32 accumulation: the products (|.| <= 64) and their sums
// (< 2^24 for K <= 131071 samples) are exact, so the visibilities stay bit-exact
// (profiles/microbench/fp6_exact_probe.hip).  The corner turn emits the 6-bit codes, so the
// contraction kernel has no unpack work at all: fragments go LDS -> VGPR -> MFMA.
//
//   stash6[c][ib][kt64][sub][plane re|im][1536 B]      (6144 B per 64-input block and 64 samples)
//   fragment = MFMA operand image of 32 inputs x 64 samples: lane = 32h + r holds the 32 codes of
//   input r, samples 32h..32h+31, as a 192-bit string (field j at bit 6j): dwords 0-3 at
//   [lane*16], dwords 4-5 at [1024 + lane*8]  (ds_read_b128 + ds_read_b64, both conflict-free).
// =======================================================================================
typedef int v8i __attribute__((ext_vector_type(8)));
constexpr int F6_FRAG = 1536;
constexpr int F6_KT_BYTES = 4 * F6_FRAG;     // (sub, plane) x 1536 per 64-input block per 64 samples

// four two's-complement nibbles (one per byte, 0..15) -> four E3M2 codes (one per byte)
__device__ __forceinline__ uint32_t nib4_to_e3m2(uint32_t n) {
    const uint32_t idx = n & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(0x17161514u, 0x12100C00u, idx);   //  0..7  -> 0,12,16,18,20,21,22,23
    const uint32_t hi = __builtin_amdgcn_perm(0x2C303234u, 0x35363738u, idx);   // -8..-1 -> sign | code(|v|)
    const uint32_t sel = ((n & 0x08080808u) >> 1) | 0x03020100u;               // byte k: k (>=0) or 4+k (<0)
    return __builtin_amdgcn_perm(hi, lo, sel);
}
// four 6-bit codes (one per byte) -> 24 contiguous bits
__device__ __forceinline__ uint32_t pack4x6(uint32_t w) {
    const uint32_t t = (w & 0x003F003Fu) | ((w >> 2) & 0x0FC00FC0u);
    return (t & 0xFFFu) | ((t >> 4) & 0xFFF000u);
}
// eight 24-bit groups -> 192 bits
__device__ __forceinline__ void pack8x24(const uint32_t (&g)[8], uint32_t (&d)[6]) {
    d[0] = g[0] | (g[1] << 24);
    d[1] = (g[1] >> 8) | (g[2] << 16);
    d[2] = (g[2] >> 16) | (g[3] << 8);
    d[3] = g[4] | (g[5] << 24);
    d[4] = (g[5] >> 8) | (g[6] << 16);
    d[5] = (g[6] >> 16) | (g[7] << 8);
}

// grid (channel, 32-sample half-tile of the gulp), 192 threads.  hk_off = index of the gulp's first
// half-tile in the staging area (a 64-sample fragment may be completed by two gulps).
__global__ __launch_bounds__(192) void corner_turn_fp6_kernel(const uint8_t* __restrict__ in,
                                                              uint8_t* __restrict__ stash, int ntime,
                                                              int nchan, int ninput, int nblk64,
                                                              int cap_kt64, int hk_off) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ct_lds[];
    const int c = blockIdx.x, hkl = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = blockDim.x >> 6;
    const int in_bytes = 32 * ninput;
    uint8_t* lin = ct_lds;
    uint8_t* lout = ct_lds + ((in_bytes + 1023) & ~1023);
    const size_t row_stride = (size_t)nchan * ninput;
    const uint8_t* src_c = in + (size_t)c * ninput;
    const int t_base = hkl * 32;
    const int t_valid = max(0, min(32, ntime - t_base));

    const int npiece = (in_bytes + 1023) >> 10;
    for (int n = wave; n < npiece; n += nwave) {
        const int off = n * 1024 + lane * 16;
        int t = off / ninput, i = off - t * ninput;
        if (t >= t_valid) { t = 0; i = 0; }
        const uint8_t* g = src_c + (size_t)(t_valid > 0 ? t_base + t : 0) * row_stride + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lin + n * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // each thread: 4 inputs x 32 samples -> 4 x (re 24 B, im 24 B)
    const int nq = nblk64 * 16;
    for (int q = tid; q < nq; q += blockDim.x) {
        const int i0 = q * 4;
        uint32_t o[4][8];          // o[input][group of 4 samples], bytes = packed 4+4-bit samples
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint32_t v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = 4 * g + j;
                v[j] = (i0 < ninput && t < t_valid) ? *reinterpret_cast<const uint32_t*>(lin + t * ninput + i0) : 0u;
            }
            transpose4x4_bytes(v[0], v[1], v[2], v[3], o[0][g], o[1][g], o[2][g], o[3][g]);
        }
        // half-fragment image in LDS: [frag = i0>>5][plane][512 B: r*16 | 256 B: r*8]
        uint8_t* fb = lout + (i0 >> 5) * (2 * 768) + (i0 & 31) * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t gr[8], gi[8], dr[6], di[6];
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const uint32_t w = o[j][g];
                gr[g] = pack4x6(nib4_to_e3m2((w >> 4) & 0x0F0F0F0Fu));   // hi nibble = real
                gi[g] = pack4x6(nib4_to_e3m2(w & 0x0F0F0F0Fu));          // lo nibble = imag
            }
            pack8x24(gr, dr);
            pack8x24(gi, di);
            *reinterpret_cast<uint4*>(fb + 16 * j) = make_uint4(dr[0], dr[1], dr[2], dr[3]);
            *reinterpret_cast<uint2*>(fb + 512 - (i0 & 31) * 8 + 8 * j) = make_uint2(dr[4], dr[5]);
            *reinterpret_cast<uint4*>(fb + 768 + 16 * j) = make_uint4(di[0], di[1], di[2], di[3]);
            *reinterpret_cast<uint2*>(fb + 768 + 512 - (i0 & 31) * 8 + 8 * j) = make_uint2(di[4], di[5]);
        }
    }
    __syncthreads();

    // stream the image out: per (fragment, plane) a 512-B run (dwords 0-3) and a 256-B run (dwords 4-5)
    const int hk = hk_off + hkl, kt64 = hk >> 1, h = hk & 1;
    const int nchunk = nblk64 * 2 * 2 * 48;      // 16-byte chunks
    for (int ci = tid; ci < nchunk; ci += blockDim.x) {
        const int fp = ci / 48, w = ci - fp * 48;
        const int f = fp >> 1, plane = fp & 1;
        const uint4 val = *reinterpret_cast<const uint4*>(lout + ci * 16);
        uint8_t* frag = stash + (((size_t)c * nblk64 + (f >> 1)) * cap_kt64 + kt64) * F6_KT_BYTES + ((f & 1) * 2 + plane) * F6_FRAG;
        uint8_t* dst = w < 32 ? frag + h * 512 + w * 16 : frag + 1024 + h * 256 + (w - 32) * 16;
        *reinterpret_cast<uint4*>(dst) = val;
    }
}

// zero the second (h = 1) half of every fragment of K tile kt64 (an odd number of 32-sample half-tiles
// was staged: code 0 is the value 0)
__global__ void fp6_zero_half_kernel(uint8_t* __restrict__ stash, int nblk64, int cap_kt64, int kt64) {
    const int cb = blockIdx.x;   // channel * nblk64 + ib
    uint8_t* base = stash + ((size_t)cb * cap_kt64 + kt64) * F6_KT_BYTES;
    for (int ci = threadIdx.x; ci < 4 * 48; ci += blockDim.x) {
        const int fr = ci / 48, w = ci - fr * 48;
        uint8_t* dst = w < 32 ? base + fr * F6_FRAG + 512 + w * 16 : base + fr * F6_FRAG + 1024 + 256 + (w - 32) * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
    }
}

typedef int v6i __attribute__((ext_vector_type(6)));
// One block-scaled MFMA, E3M2 x E3M2 (cbsz/blgp 3), scales 2^0, from inline asm: the builtin takes 8-dword
// operands and hipcc assembles the 6 live dwords through AGPR copies (144 v_accvgpr_write per K step);
// here the operands are 192-bit VGPR tuples and the accumulator stays in AGPRs.  `s_nop 1` covers the
// VALU-write -> MFMA-read wait states hipcc does not insert inside asm (guide 5.7 item 2).
__device__ __forceinline__ void mfma_e3m2(v16f& acc, const v6i& a, const v6i& b, int scale127) {
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:3 blgp:3"
                 : "+a"(acc) : "v"(a), "v"(b), "v"(scale127) : "memory");   // "memory": keeps the hand-placed
                 // LDS reads / LDS-DMA issues between the MFMAs where the source puts them
}

__global__ __launch_bounds__(256, 1) void xcorr_fp6_kernel(XcorrParams p) {
    constexpr int SLOT_BYTES = F6_KT_BYTES;            // one K step (64 samples) per stage
    constexpr int STAGE_BYTES = XC_NSLOT * SLOT_BYTES;  // 24 KB
    constexpr int NLOAD = SLOT_BYTES / 1024;            // 6 LDS-DMA pieces per wave per stage
    __shared__ __attribute__((aligned(16))) uint8_t lds[XC_RING * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int c, wg;
    {
        const int b = blockIdx.x;
        if ((p.nchan & 7) == 0) { const int xcd = b & 7, slot = b >> 3; c = xcd + 8 * (slot / p.nwg); wg = slot % p.nwg; }
        else { c = b / p.nwg; wg = b % p.nwg; }
    }
    const WgDesc* dp = p.descs + wg;
    const int a_slot = dp->wave_a[wave], b_slot = dp->wave_b[wave];
    const bool active = a_slot != 0xFF;
    const int blk_a = active ? dp->slot_blk[a_slot] : 0, blk_b = active ? dp->slot_blk[b_slot] : 0;
    const uint8_t* gsrc = p.stash + ((size_t)c * p.nblk64 + dp->slot_blk[wave]) * (size_t)p.cap_kt * SLOT_BYTES + lane * 16;
    const int nstage = p.nkt;    // K steps of 64 samples

    auto issue_piece = [&](int s, int n) {
        const int ssrc = s < nstage ? s : nstage - 1;
        const uint8_t* g = gsrc + (size_t)ssrc * SLOT_BYTES + n * 1024;
        uint8_t* l = lds + (s & (XC_RING - 1)) * STAGE_BYTES + wave * SLOT_BYTES + n * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    };
    v16f accR[2][2], accP[2][2], accQ[2][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) { accR[m][n] = (v16f)(0.f); accP[m][n] = (v16f)(0.f); accQ[m][n] = (v16f)(0.f); }

    const int a_base = (active ? a_slot : 0) * SLOT_BYTES, b_base = (active ? b_slot : 0) * SLOT_BYTES;
    // Operand fragments, double-buffered in registers: set [s & 1] holds K step s.  Index f = sub*2 + plane
    // (re|im); dwords 0-3 (Lo) and 4-5 (Hi).  All indices are compile-time (the K loop is unrolled by two).
    v4i aLo[2][4], bLo[2][4];
    v2i aHi[2][4], bHi[2][4];
    const bool skip01 = __builtin_amdgcn_readfirstlane((int)(!active || blk_a == blk_b)) != 0;
    const int scale127 = 127;     // E8M0 exponent 127 = 2^0 in byte 0 of the scale operand

    // one of the 16 register loads of K step s into set SET: part 0..15 = (A|B, fragment f, Lo|Hi)
    auto load_part = [&](auto setc, int s, int part) {
        constexpr int SET = decltype(setc)::value;
        const uint8_t* base = lds + (s & (XC_RING - 1)) * STAGE_BYTES;
        const int f = (part >> 1) & 3;
        const uint8_t* fb = base + ((part & 8) ? b_base : a_base) + f * F6_FRAG;
        if (part & 8) {
            if (part & 1) bHi[SET][f] = *reinterpret_cast<const v2i*>(fb + 1024 + lane * 8);
            else bLo[SET][f] = *reinterpret_cast<const v4i*>(fb + lane * 16);
        } else {
            if (part & 1) aHi[SET][f] = *reinterpret_cast<const v2i*>(fb + 1024 + lane * 8);
            else aLo[SET][f] = *reinterpret_cast<const v4i*>(fb + lane * 16);
        }
    };

    // K step s from register set CUR, hand-interleaved in program order:
    //   MFMA t  |  LDS read part t of K step s+1 into the other set  |  after every odd MFMA one LDS-DMA piece
    // so the DMA issue (~37 cycles a piece) and the LDS latency run under the 35-cycle MFMAs.
    auto kstep = [&](auto curc, int s) {
        constexpr int CUR = decltype(curc)::value;
        using NXT = std::integral_constant<int, CUR ^ 1>;
        v6i A[4], B[4];
#pragma unroll
        for (int f = 0; f < 4; f++) {
            A[f] = (v6i){aLo[CUR][f].x, aLo[CUR][f].y, aLo[CUR][f].z, aLo[CUR][f].w, aHi[CUR][f].x, aHi[CUR][f].y};
            B[f] = (v6i){bLo[CUR][f].x, bLo[CUR][f].y, bLo[CUR][f].z, bLo[CUR][f].w, bHi[CUR][f].x, bHi[CUR][f].y};
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int m = t >> 3, n = (t >> 2) & 1, k = t & 3;
            // (no diagonal-tile skip here: a branch around asm MFMAs makes hipcc copy the accumulators)
            if (k == 0) mfma_e3m2(accR[m][n], A[2 * m], B[2 * n], scale127);
            else if (k == 1) mfma_e3m2(accP[m][n], A[2 * m + 1], B[2 * n], scale127);
            else if (k == 2) mfma_e3m2(accQ[m][n], A[2 * m], B[2 * n + 1], scale127);
            else mfma_e3m2(accR[m][n], A[2 * m + 1], B[2 * n + 1], scale127);
            load_part(NXT{}, s + 1, t);
            if ((t & 1) && (t >> 1) < NLOAD) issue_piece(s + 3, t >> 1);
        }
        wait_vmcnt<NLOAD>();
        __builtin_amdgcn_s_barrier();                   // stages s+1, s+2 visible; stage s free
    };

    // same ring protocol as xcorr_mfma_kernel with one K step per stage
#pragma unroll
    for (int st = 0; st < 3; st++)
#pragma unroll
        for (int n = 0; n < NLOAD; n++) issue_piece(st, n);
    wait_vmcnt<NLOAD>();
    __builtin_amdgcn_s_barrier();                       // stages 0 and 1 visible
#pragma unroll
    for (int part = 0; part < 16; part++) load_part(std::integral_constant<int, 0>{}, 0, part);
    for (int s = 0; s < nstage; s += 2) {
        kstep(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nstage) kstep(std::integral_constant<int, 1>{}, s + 1);
    }
    wait_vmcnt<0>();
    // the last asm MFMA's result must have retired before the epilogue reads the accumulators
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (!active) return;

    const int qs = (int)(((int64_t)(p.nstand / 2 + 1) * p.nstand) / 4);
    int32_t* out_r = p.out + (int64_t)c * p.per_chan;
    int32_t* out_i = out_r + p.matlen;
    const int odd = lane & 1, cpar = (lane >> 1) & 1, quad = 2 * cpar + odd;
    auto cell = [&](int v0, int v1, int v2, int v3) {
        const int g0 = dpp_xor1(odd ? v0 : v2), g1 = dpp_xor1(odd ? v1 : v3);
        return odd ? make_int4(g0, v2, g1, v3) : make_int4(v0, g0, v1, g1);
    };
    const bool interior = __builtin_amdgcn_readfirstlane((int)(blk_a > blk_b && blk_a * 64 + 64 <= 2 * p.nstand)) != 0;
    const bool accumulate = p.accumulate != 0;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
            if (m == 0 && n == 1 && skip01) continue;
            const int ibase = blk_a * 64 + m * 32, jbase = blk_b * 64 + n * 32;
            const int Ch = (jbase >> 2) + ((lane & 31) >> 2), C = 2 * Ch + cpar;
            const int wcol = (quad * qs + Ch) * 4;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int Rh = (ibase >> 2) + 2 * u + (lane >> 5), R = 2 * Rh + odd;
                int vr[4], vi[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {       // integers < 2^24: exact in fp32, exact conversion
                    vr[v] = (int)accR[m][n][4 * u + v];
                    vi[v] = (int)(accP[m][n][4 * u + v] - accQ[m][n][4 * u + v]);
                }
                int4 cr = cell(vr[0], vr[1], vr[2], vr[3]);
                int4 ci = cell(vi[0], vi[1], vi[2], vi[3]);
                const int w = wcol + ((Rh * (Rh + 1)) >> 1) * 4;
                int4* pr = reinterpret_cast<int4*>(out_r + w);
                int4* pi = reinterpret_cast<int4*>(out_i + w);
                if (interior || (Rh >= Ch && R < p.nstand && C < p.nstand)) {
                    if (accumulate) {
                        const int4 o_r = *pr, o_i = *pi;
                        cr.x += o_r.x; cr.y += o_r.y; cr.z += o_r.z; cr.w += o_r.w;
                        ci.x += o_i.x; ci.y += o_i.y; ci.z += o_i.z; ci.w += o_i.w;
                    }
                    *pr = cr;
                    *pi = ci;
                }
            }
        }
}

