// Prototype of the X-engine K loop (diagnostic, not part of the product; results are not checked): prices MFMA shape and
// waves per SIMD with the REAL instruction mix of xcorr_fused_kernel -- LDS-DMA staging of 4 x 64-input blocks per
// stage, byte-transposing LDS reads, nibble unpack VALU, int8 MFMAs on three accumulator planes, one barrier per stage --
// but without item lists, epilogue or the triangular tiling.  Every work-group streams the same L2-resident rows.
//   SHAPE 32: v_mfma_i32_32x32x32_i8   SHAPE 16: v_mfma_i32_16x16x64_i8
//   NW 4: wave tile 64x64, one wave per SIMD      NW 8: wave tile 64x32, two waves per SIMD
// Stage = 64 samples (16 KiB), ring of 6 stages, DMA 3 (4 for the 16x16x64 loop) stages ahead, one stage in flight at each barrier.
// Build: hipcc -O3 --offload-arch=gfx950 kloop_proto.hip -o kloop_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int STAGE = 4 * 64 * 64;       // 4 blocks x 64 inputs x 64 samples
constexpr int RING = 6;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int SHAPE, int NW, int VAR = 0>
__global__ __launch_bounds__(64 * NW, NW / 4) void k(const uint8_t* __restrict__ src, int* __restrict__ out,
                                                    unsigned long long* st, int nstage, uint32_t row_stride) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[RING * STAGE];
    constexpr int DEPTH = SHAPE == 16 ? 4 : 3;   // stages of DMA in flight ahead of the MFMAs (the 16x16x64 loop reads LDS two stages ahead)
    constexpr int NC = NW == 4 ? 2 : 1;          // 32-column halves per wave
    constexpr int NPIECE = 16 / NW;              // 1 KiB pieces per wave per stage
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wq = wave & 3, nh = wave >> 2;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds);
    // DMA: wave brings NPIECE pieces (8 rows x 128 B each) of one block pair; per-lane source offset as in the real kernel
    const int chunk = (lane & 7) ^ (((lane >> 4) & 3) << 1);
    const uint32_t lane_off = (uint32_t)(lane >> 3) * row_stride + (uint32_t)chunk * 16u + (uint32_t)(wave & 1) * 128u;
    uint32_t voff[NPIECE];
#pragma unroll
    for (int n = 0; n < NPIECE; n++) voff[n] = lane_off + (uint32_t)n * 8u * row_stride - (uint32_t)(n * 1024);
    const uint8_t* gbase = src + (size_t)(blockIdx.x & 7) * 4096 + (size_t)(wave * NPIECE * 8) * row_stride;
    auto issue = [&](int s, int slot) {
        const uint8_t* sb = gbase + (size_t)((s & 15) * 64) * row_stride;     // 16 stages of rows, re-read (L2 resident)
        const uint32_t la = lds_base + slot * STAGE + wave * NPIECE * 1024;
        if (NPIECE == 4)
            asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                         "global_load_lds_dwordx4 %2, %4 offset:2048\n\tglobal_load_lds_dwordx4 %3, %4 offset:3072"
                         :: "v"(voff[0]), "v"(voff[1]), "v"(voff[2 % NPIECE]), "v"(voff[3 % NPIECE]), "s"(sb), "s"(la) : "memory");
        else
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                         :: "v"(voff[0]), "v"(voff[1]), "s"(sb), "s"(la) : "memory");
    };
    // read side: image per stage = 2 pairs x [64 rows][128 B]; same swizzled address map as the real kernel
    const int tr_off = ((lane >> 5) * 16 + ((lane & 15) >> 1)) * 128 + (((lane >> 4) & 1) ^ (((lane >> 2) & 3) << 1)) * 16 + (lane & 1) * 8;
    const int a_off = (wq >> 1) * 8192 + (tr_off ^ ((wq & 1) * 64));
    const int b_off = ((wq & 1) ^ 1) * 8192 + (tr_off ^ (((wq >> 1) & 1) * 64)) ^ (nh * 32);
    auto tr = [&](const uint8_t* base, int off) {
        return __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(base + off));
    };
    const v4i M = (v4i)(0xF0F0F0F0);

#pragma unroll
    for (int s = 0; s < DEPTH; s++) issue(s, s);
    wait_vmcnt<NPIECE>();
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int sum = 0;
    int rs = 0, rs1 = 1, rf = DEPTH;
    auto bump = [&](int& r) { r = (r + 1 == RING) ? 0 : r + 1; };

    if (SHAPE == 32) {
        v16i R[2][NC], P[2][NC], Q[2][NC];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NC; n++) { R[m][n] = (v16i)(0); P[m][n] = (v16i)(0); Q[m][n] = (v16i)(0); }
        auto load_raw = [&](int slot, int j, v4i (&a)[2], v4i (&b)[NC]) {      // K-tile j (32 samples) of a stage
            const uint8_t* base = lds + slot * STAGE + j * 4096;
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const v2i a0 = tr(base, a_off ^ (sub * 32)), a1 = tr(base, (a_off ^ (sub * 32)) + 1024);
                a[sub] = (v4i){a0.x, a0.y, a1.x, a1.y};
            }
#pragma unroll
            for (int n = 0; n < NC; n++) {
                const v2i b0 = tr(base, b_off ^ (n * 32)), b1 = tr(base, (b_off ^ (n * 32)) + 1024);
                b[n] = (v4i){b0.x, b0.y, b1.x, b1.y};
            }
        };
        v4i ra[2], rb[NC], ar[2], ai[2], br[NC], bi[NC];
        load_raw(rs, 0, ra, rb);
#pragma unroll
        for (int m = 0; m < 2; m++) { ar[m] = ra[m] & M; ai[m] = (ra[m] << 4) & M; }
#pragma unroll
        for (int n = 0; n < NC; n++) { br[n] = rb[n] & M; bi[n] = (rb[n] << 4) & M; }
        load_raw(rs, 1, ra, rb);
        for (int s = 0; s < nstage; s++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (j == 0) issue(s + DEPTH, rf);
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < NC; n++) {
                        R[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[m], br[n], R[m][n], 0, 0, 0);
                        P[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[m], br[n], P[m][n], 0, 0, 0);
                        Q[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[m], bi[n], Q[m][n], 0, 0, 0);
                        R[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[m], bi[n], R[m][n], 0, 0, 0);
                    }
#pragma unroll
                for (int m = 0; m < 2; m++) { ar[m] = ra[m] & M; ai[m] = (ra[m] << 4) & M; }
#pragma unroll
                for (int n = 0; n < NC; n++) { br[n] = rb[n] & M; bi[n] = (rb[n] << 4) & M; }
                if (j == 0) load_raw(rs1, 0, ra, rb); else load_raw(rs1, 1, ra, rb);
#pragma unroll
                for (int i = 0; i < 8 * NC; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NC == 2) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    else if (i < 4) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    if (i >= (NC == 2 ? 8 : 2)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            wait_vmcnt<NPIECE>();
            __builtin_amdgcn_s_barrier();
            bump(rs); bump(rs1); bump(rf);
        }
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NC; n++) sum += R[m][n][0] + P[m][n][5] + Q[m][n][9];
    } else {
        constexpr int CT = 2 * NC;               // 16-column tiles per wave
        v4i R[4][CT], P[4][CT], Q[4][CT];
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < CT; n++) { R[m][n] = (v4i)(0); P[m][n] = (v4i)(0); Q[m][n] = (v4i)(0); }
        // one K step = 64 samples: lane group g = lane>>4 takes samples 16g..16g+15 of its 16 inputs
        const int tr16 = ((lane >> 4) * 16 + ((lane & 15) >> 1)) * 128 + ((((lane >> 4) & 1) ^ (((lane >> 2) & 3) << 1))) * 16 + (lane & 1) * 8;
        const int a16 = (wq >> 1) * 8192 + (tr16 ^ ((wq & 1) * 64));
        const int b16 = ((wq & 1) ^ 1) * 8192 + (tr16 ^ (((wq >> 1) & 1) * 64)) ^ (nh * 32);
        auto load_raw = [&](int slot, v4i (&a)[4], v4i (&b)[CT]) {
            const uint8_t* base = lds + slot * STAGE;
#pragma unroll
            for (int rt = 0; rt < 4; rt++) {
                const v2i a0 = tr(base, a16 ^ (rt * 16)), a1 = tr(base, (a16 ^ (rt * 16)) + 1024);
                a[rt] = (v4i){a0.x, a0.y, a1.x, a1.y};
            }
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const v2i b0 = tr(base, b16 ^ (ct * 16)), b1 = tr(base, (b16 ^ (ct * 16)) + 1024);
                b[ct] = (v4i){b0.x, b0.y, b1.x, b1.y};
            }
        };
        v4i ra[4], rb[CT], ar[4], ai[4], br[CT], bi[CT];
        load_raw(rs, ra, rb);
#pragma unroll
        for (int m = 0; m < 4; m++) { ar[m] = ra[m] & M; ai[m] = (ra[m] << 4) & M; }
#pragma unroll
        for (int n = 0; n < CT; n++) { br[n] = rb[n] & M; bi[n] = (rb[n] << 4) & M; }
        load_raw(rs1, ra, rb);
        int rs2 = 2;
        constexpr int SCH = VAR & 15;
        for (int s = 0; s < nstage; s++) {
            if (!(VAR & 16)) issue(s + DEPTH, rf);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < CT; n++) {
                    R[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ar[m], br[n], R[m][n], 0, 0, 0);
                    P[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai[m], br[n], P[m][n], 0, 0, 0);
                    Q[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ar[m], bi[n], Q[m][n], 0, 0, 0);
                    R[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai[m], bi[n], R[m][n], 0, 0, 0);
                }
            if (!(VAR & 128)) {
#pragma unroll
            for (int m = 0; m < 4; m++) { ar[m] = ra[m] & M; ai[m] = (ra[m] << 4) & M; }
#pragma unroll
            for (int n = 0; n < CT; n++) { br[n] = rb[n] & M; bi[n] = (rb[n] << 4) & M; }
            } else {
#pragma unroll
            for (int m = 0; m < 4; m++) { ar[m] = ra[m]; ai[m] = ra[m]; }
#pragma unroll
            for (int n = 0; n < CT; n++) { br[n] = rb[n]; bi[n] = rb[n]; }
            }
            if (!(VAR & 64)) load_raw(rs2, ra, rb);
            else {
#pragma unroll
                for (int m = 0; m < 4; m++) asm volatile("" : "+v"(ra[m]));
#pragma unroll
                for (int n = 0; n < CT; n++) asm volatile("" : "+v"(rb[n]));
            }
            // 16 CT MFMAs; (4 + CT) * 12 VALU; 2 * (4 + CT) LDS reads
            if (SCH == 0) {
#pragma unroll
            for (int i = 0; i < 16 * CT; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (CT == 4) { if ((i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }          // 96 VALU / 64 MFMA
                else { if ((i & 3) != 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }                   // 72 VALU / 32 MFMA
                if (CT == 4 ? (i >= 32 && (i & 1)) : (i >= 8 && (i & 1))) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            } else if (SCH == 1) {      // at most 2 (sometimes 3) VALU behind each MFMA: 2 VALU = the 8 issue cycles an MFMA leaves
#pragma unroll
            for (int i = 0; i < 16 * CT; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (CT == 4) { if (i & 1) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
                else { if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
                if (CT == 4 ? (i >= 32 && (i & 1)) : (i >= 8 && (i & 1))) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            }   // VAR == 2: no hints
            if (!(VAR & 16)) wait_vmcnt<NPIECE>();
            if (!(VAR & 32)) __builtin_amdgcn_s_barrier();
            bump(rs); bump(rs1); bump(rs2); bump(rf);
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int n = 0; n < CT; n++) sum += R[m][n][0] + P[m][n][1] + Q[m][n][3];
    }
    wait_vmcnt<0>();
    asm volatile("" :: "v"(sum));
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 * NW + threadIdx.x] = sum;
    if (lane == 0) { st[2 * (blockIdx.x * NW + wave)] = t1 - t0; st[2 * (blockIdx.x * NW + wave) + 1] = r1 - r0; }
}

static int g_reps = 100;
template <int SHAPE, int NW, int VAR = 0>
void run(const char* name, const uint8_t* src, int* out, unsigned long long* st, uint32_t row_stride) {
    const int nstage = 1500, blocks = 256;     // 1500 stages of 64 samples = 96000 samples
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(HIP_KERNEL_NAME(k<SHAPE, NW, VAR>), dim3(blocks), dim3(64 * NW), 0, 0, src, out, st, nstage, row_stride);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = g_reps;
    for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL(HIP_KERNEL_NAME(k<SHAPE, NW, VAR>), dim3(blocks), dim3(64 * NW), 0, 0, src, out, st, nstage, row_stride);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks * NW);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0, cyc = 0; for (int w = 0; w < blocks * NW; w++) { clk += (double)h[2 * w] / (double)h[2 * w + 1] * 100e6; cyc += (double)h[2 * w]; }
    clk /= blocks * NW; cyc /= blocks * NW;
    // complex MACs: per work-group a 128x128 tile (4 wave tiles of 64x64) x 64 samples per stage; 8 int8 ops per cMAC
    const double ops = (double)reps * blocks * nstage * 128.0 * 128.0 * 64.0 * 8.0;
    printf("%-28s %8.1f TOP/s   clock %.3f GHz   %.0f cycles/stage (MFMA pipe: 1024)   %.3f ms/launch\n", name,
           ops / (ms * 1e-3) / 1e12, clk / 1e9, cyc / nstage, ms / reps);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;      // run one variant for a long time (power sampling): index, reps
    if (argc > 2) g_reps = atoi(argv[2]);
    const uint32_t row_stride = 96 * 704;
    const size_t bytes = (size_t)row_stride * (16 * 64 + 256) + (1 << 20);
    uint8_t* src; int* out; unsigned long long* st;
    std::vector<uint8_t> h(bytes);
    srand(1); for (auto& v : h) v = (uint8_t)(rand() >> 7);
    (void)hipMalloc(&src, bytes); (void)hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 16);
    for (int rep = 0; rep < (only >= 0 ? 1 : 2); rep++) {
        if (only < 0 || only == 0) run<32, 4>("32x32x32, 4 waves (64x64)", src, out, st, row_stride);
        if (only < 0 || only == 1) run<32, 8>("32x32x32, 8 waves (64x32)", src, out, st, row_stride);
        if (only < 0 || only == 2) run<16, 4, 2>("16x16x64, 4 waves, no hints", src, out, st, row_stride);
        if (only < 0 || only == 3) run<16, 8, 1>("16x16x64, 8 waves, VALU 2-2-2-3", src, out, st, row_stride);
        if (only < 0 || only == 4) run<16, 8, 1 + 64>("   - no LDS reads", src, out, st, row_stride);
        if (only < 0 || only == 5) run<16, 8, 1 + 16 + 32 + 64>("   - no DMA/barrier/LDS reads", src, out, st, row_stride);
        if (only < 0 || only == 6) run<16, 8, 1 + 16 + 32 + 64 + 128>("   - MFMA only", src, out, st, row_stride);
    }
    return 0;
}
