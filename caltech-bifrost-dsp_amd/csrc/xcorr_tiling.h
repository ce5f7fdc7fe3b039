// Host-side tiling and index logic of the X-engine: no HIP in here, so the same code is compiled by hipcc into libxeng
// and by g++ with -fsanitize=address,undefined into the host test driver (tests/host/tiling_check.cpp).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace xeng {

constexpr int XC_NSLOT = 4;               // 64-input blocks resident per stage

// Work-group descriptor: which 64-input blocks a work-group stages (one per wave),
// and which (row block, col block) tile each of its 4 waves contracts.
struct WgDesc {
    uint8_t slot_blk[XC_NSLOT];  // 64-input block loaded by wave w into LDS slot w
    uint8_t wave_a[4];           // LDS slot of the wave's row block (0xFF: wave idle)
    uint8_t wave_b[4];           // LDS slot of the wave's column block
    uint8_t nwave;
    uint8_t pad[3];
};

// One entry of a work-group's list: a (channel, tile group) item, or one K slice of it.  Whole items cover all
// stages; the items left over after dealing whole items evenly are cut along K into slices, one per work-group,
// so that every work-group of a launch contracts (almost) the same number of stages.  Slice j > 0 adds to what
// slice j-1 stored: it waits for flags[chain] >= epoch*16 + j before its read-modify-write.
struct WorkEntry {
    uint32_t c_wg;        // channel | tile group << 16
    uint32_t stages;      // first stage | number of stages << 16
    uint32_t slice;       // slice index | slices of the item << 8 | valid << 16
    uint32_t chain;       // flag index of a split item
};

// --------------------------------------------------------------------------------------
// Triangular tiling of the nblk64 x nblk64 grid of 64x64-input wave tiles onto work-groups
// of 4 waves that share at most 4 staged 64-input blocks (hard part 2).
//   * off-diagonal 128x128 squares: 4 tiles, 4 blocks
//   * diagonal pairs: 3 tiles (+1 tile of the unpaired last block row when nblk64 is odd)
//   * what is left of the last block row is packed 3-4 tiles per work-group
// 704 inputs -> 11 blocks -> 66 tiles in 17 work-groups (97 % of wave slots busy).
// --------------------------------------------------------------------------------------
inline std::vector<WgDesc> build_wg_descs(int nblk64) {
    std::vector<WgDesc> out;
    auto blank = []() {
        WgDesc d;
        memset(&d, 0, sizeof(d));
        for (int w = 0; w < 4; w++) d.wave_a[w] = d.wave_b[w] = 0xFF;
        return d;
    };
    auto finish = [&](WgDesc d, int nslot) {
        for (int s = nslot; s < XC_NSLOT; s++) d.slot_blk[s] = d.slot_blk[0];  // harmless duplicate loads
        out.push_back(d);
    };
    const int np = nblk64 / 2;
    const int L = (nblk64 & 1) ? nblk64 - 1 : -1;
    std::vector<int> left;  // column blocks j of the remaining tiles (L, j)
    if (L >= 0)
        for (int j = 0; j <= L; j++) left.push_back(j);
    for (int k = 1; k < np; k++)
        for (int m = 0; m < k; m++) {
            WgDesc d = blank();
            d.slot_blk[0] = 2 * k; d.slot_blk[1] = 2 * k + 1; d.slot_blk[2] = 2 * m; d.slot_blk[3] = 2 * m + 1;
            const uint8_t wa[4] = {0, 0, 1, 1}, wb[4] = {2, 3, 2, 3};
            for (int w = 0; w < 4; w++) { d.wave_a[w] = wa[w]; d.wave_b[w] = wb[w]; }
            d.nwave = 4;
            finish(d, 4);
        }
    for (int k = 0; k < np; k++) {
        WgDesc d = blank();
        d.slot_blk[0] = 2 * k; d.slot_blk[1] = 2 * k + 1;
        d.wave_a[0] = 0; d.wave_b[0] = 0;
        d.wave_a[1] = 1; d.wave_b[1] = 0;
        d.wave_a[2] = 1; d.wave_b[2] = 1;
        d.nwave = 3;
        int nslot = 2;
        auto it = std::find(left.begin(), left.end(), 2 * k);
        if (it != left.end()) {
            left.erase(it);
            d.slot_blk[2] = (uint8_t)L; nslot = 3;
            d.wave_a[3] = 2; d.wave_b[3] = 0;
            d.nwave = 4;
        }
        finish(d, nslot);
    }
    while (!left.empty()) {
        WgDesc d = blank();
        d.slot_blk[0] = (uint8_t)L;
        int nslot = 1, nw = 0;
        for (size_t q = 0; q < left.size() && nw < 4;) {
            const int j = left[q];
            int slot = -1;
            if (j == L) slot = 0;
            else if (nslot < XC_NSLOT) { slot = nslot; d.slot_blk[nslot++] = (uint8_t)j; }
            if (slot < 0) { q++; continue; }
            d.wave_a[nw] = 0; d.wave_b[nw] = (uint8_t)slot; nw++;
            left.erase(left.begin() + q);
        }
        d.nwave = (uint8_t)nw;
        finish(d, nslot);
    }
    return out;
}

// --------------------------------------------------------------------------------------
// Work lists of the persistent fused kernel (WorkEntry[grid][maxi], xcorr_kernels.h).
// Work-groups b with the same b & 7 sit on one XCD and share that XCD's items (channels = xcd mod 8), dealt
// round-robin so that concurrent work-groups contract neighbouring tile groups of the same channels.  With
// n items for W work-groups every work-group gets n / W whole items and the first n % W one more
// (704 inputs x 96 channels on 256 CUs: 204 items per XCD for 32 work-groups = 7 items for 12 of them, 6 for
// 20; the next launch's work-groups take over the CUs of the latter).  Opt-in (XENG_SPLITK=1): the left-over
// items are cut along K into W slices in all, one per work-group, with an ordered read-modify-write hand-over
// between the slices of an item -- balanced, but not faster (see xengXgpuInitialize).
// --------------------------------------------------------------------------------------
struct WorkList {
    std::vector<WorkEntry> entries;
    int maxi = 0, nchains = 0;
    uint32_t* dev = nullptr;
};

// stagger: every second work-group of an XCD class contracts its FIRST item in two K halves (the second half adds to what
// the first stored), which shifts all its later item boundaries by half an item against its neighbours': the epilogues of
// the 256 work-groups (128 KB of stores each) then no longer fall into the same few microseconds.
// Order of a channel's tile groups in its XCD's item list.  The W work-groups of an XCD contract W consecutive items
// at a time ("a round"); a channel whose nwg items straddle a round boundary is contracted in two parts, about one
// item time apart, and whatever input blocks the second part needs have left the XCD's L2 by then.  So the groups of
// such a channel are ordered to make (distinct blocks of the first part) + (distinct blocks of the second part) minimal
// -- e.g. 704 inputs, 32 work-groups per XCD: 184 -> 171 block fetches per XCD and launch against 132 unavoidable.
// Exhaustive over the subsets of the smaller part (17 groups: at most 24310); identity when that would be too many,
// when the channel spans more than two rounds, or when it is not split.  start = index of the channel's first item.
inline std::vector<int> channel_group_order(const std::vector<WgDesc>& descs, int start, int W) {
    const int nwg = (int)descs.size();
    std::vector<int> order(nwg);
    for (int i = 0; i < nwg; i++) order[i] = i;
    const int B = (start / W + 1) * W;                      // first round boundary behind `start`
    const int head = B - start, tail = nwg - head;
    if (tail <= 0 || tail > W || nwg > 30) return order;
    std::vector<uint64_t> mask(nwg, 0);
    for (int g = 0; g < nwg; g++)
        for (int s = 0; s < XC_NSLOT; s++) mask[g] |= 1ull << (descs[g].slot_blk[s] & 63);
    const int small = std::min(head, tail);
    double combos = 1;
    for (int i = 0; i < small; i++) combos = combos * (nwg - i) / (i + 1);
    if (combos > 200000) return order;
    auto cost = [&](uint32_t sel) {                         // sel: the groups of the smaller part
        uint64_t a = 0, b = 0;
        for (int g = 0; g < nwg; g++) ((sel >> g) & 1 ? a : b) |= mask[g];
        return __builtin_popcountll(a) + __builtin_popcountll(b);
    };
    uint32_t best = 0;
    int best_cost = 1 << 30;
    // Gosper's hack over all `small`-subsets of nwg groups, in increasing order: ties keep the first subset found
    for (uint32_t sel = (1u << small) - 1; sel < (1u << nwg);) {
        const int c = cost(sel);
        if (c < best_cost) { best_cost = c; best = sel; }
        const uint32_t lo = sel & (0u - sel), r = sel + lo;
        if (r >= (1u << nwg) || r == 0) break;
        sel = (((r ^ sel) >> 2) / lo) | r;
    }
    // identity unless it really saves a block fetch
    uint32_t ident = 0;
    for (int g = 0; g < nwg; g++) if ((small == tail) == (g >= head)) ident |= 1u << g;   // the groups the identity puts in the smaller part
    if (cost(ident) <= best_cost) return order;
    std::vector<int> in_small, in_large;
    for (int g = 0; g < nwg; g++) ((best >> g) & 1 ? in_small : in_large).push_back(g);
    const std::vector<int>& first = small == head ? in_small : in_large;
    const std::vector<int>& second = small == head ? in_large : in_small;
    int k = 0;
    for (int g : first) order[k++] = g;
    for (int g : second) order[k++] = g;
    return order;
}

inline WorkList build_work(int grid, int nchan, int nwg, int nstage, bool splitk, bool stagger = false,
                           const std::vector<WgDesc>* descs = nullptr) {
    WorkList wl;
    const bool xcd_map = (nchan & 7) == 0 && (grid & 7) == 0;
    const int ngroup = xcd_map ? 8 : 1;
    const int W = grid / ngroup;
    const int n = xcd_map ? (nchan / 8) * nwg : nchan * nwg;
    const int f = n / W, r = n % W;
    stagger = stagger && !splitk && f >= 1 && nstage >= 2;
    wl.maxi = f + (r ? 1 : 0) + (stagger ? 1 : 0);
    wl.entries.assign((size_t)grid * wl.maxi, WorkEntry{0, 0, 0, 0});
    // (only per-XCD lists are split: the slices of an item exchange partial sums through one XCD's L2)
    const bool split = splitk && xcd_map && r > 0 && nstage >= (W + r - 1) / r;
    wl.nchains = split ? ngroup * r : (stagger ? grid : 0);
    // (whole items dealt per XCD: the groups of a channel that is contracted in two rounds are ordered for L2 reuse)
    std::vector<std::vector<int>> gorder;
    if (descs && xcd_map && !splitk && !stagger && (int)descs->size() == nwg)
        for (int q = 0; q < nchan / 8; q++) gorder.push_back(channel_group_order(*descs, q * nwg, W));
    auto put = [&](int b, int k, int x, int idx, int stage0, int nst, int slice, int nslices, int chain) {
        const int q = idx / nwg;
        const int wg = gorder.empty() ? idx - q * nwg : gorder[q][idx - q * nwg];
        const int c = xcd_map ? x + 8 * q : q;
        WorkEntry& e = wl.entries[(size_t)b * wl.maxi + k];
        e.c_wg = (uint32_t)c | ((uint32_t)wg << 16);
        e.stages = (uint32_t)stage0 | ((uint32_t)nst << 16);
        e.slice = (uint32_t)slice | ((uint32_t)nslices << 8) | (1u << 16);
        e.chain = (uint32_t)chain;
    };
    for (int x = 0; x < ngroup; x++) {
        auto block_of = [&](int j) { return xcd_map ? j * 8 + x : j; };
        for (int j = 0; j < W; j++) {
            const bool two = stagger && (j & 1);     // this work-group's first item goes in two halves
            int k = 0;
            if (two) {
                put(block_of(j), 0, x, j, 0, nstage / 2, 0, 2, block_of(j));
                put(block_of(j), 1, x, j, nstage / 2, nstage - nstage / 2, 1, 2, block_of(j));
                k = 1;
            }
            for (int q = two ? 1 : 0; q < f; q++) put(block_of(j), q + k, x, j + q * W, 0, nstage, 0, 1, 0);
        }
        if (!r) continue;
        if (!split) {
            for (int i = 0; i < r; i++) put(block_of(i), f + ((stagger && (i & 1)) ? 1 : 0), x, f * W + i, 0, nstage, 0, 1, 0);
            continue;
        }
        int j = 0;
        for (int i = 0; i < r; i++) {
            const int ns = W / r + (i < W % r ? 1 : 0);
            for (int sl = 0; sl < ns; sl++, j++) {
                const int s0 = (int)((int64_t)sl * nstage / ns), s1 = (int)((int64_t)(sl + 1) * nstage / ns);
                put(block_of(j), f, x, f * W + i, s0, s1 - s0, sl, ns, x * r + i);
            }
        }
    }
    return wl;
}

// persistent grid of the fused kernel: one work-group per CU, a multiple of 8 when channels are dealt per XCD
inline int fused_grid(int nchan, int nwg, int ncu) {
    const int nitems = nchan * nwg;
    if ((nchan & 7) == 0 && ncu >= 8) return 8 * std::min(ncu / 8, (nchan / 8) * nwg);
    return std::min(ncu, nitems);
}
inline int64_t regtile_index_host(int in0, int in1, int nstand) {
    // corr_block.py:37-58
    const int a0 = in0 >> 1, a1 = in1 >> 1, p0 = in0 & 1, p1 = in1 & 1;
    const int64_t qi = ((int64_t)(a1 / 2) * (a1 / 2 + 1)) / 2 + a0 / 2;
    const int64_t quadrant = 2 * (a0 & 1) + (a1 & 1);
    const int64_t qs = ((int64_t)(nstand / 2 + 1) * nstand) / 4;
    return (quadrant * qs + qi) * 4 + 2 * p1 + p0;
}


// bfXgpuGetOrder (corr_block.py:317-333): for every (s0, s1, p0, p1) the word of the xGPU-order plane that holds the
// pair and whether it has to be conjugated to read x[s0,p0] * conj(x[s1,p1]).  Returns the index of the first entry of
// antpol_to_input that is out of range, or -1.
inline int get_order_host(const int32_t* antpol_to_input, int32_t* antpol_to_bl, int32_t* is_conj, int ns, int np) {
    const int ninput = ns * np;
    for (int k = 0; k < ninput; k++)
        if (antpol_to_input[k] < 0 || antpol_to_input[k] >= ninput) return k;
    for (int s0 = 0; s0 < ns; s0++)
        for (int s1 = 0; s1 < ns; s1++)
            for (int p0 = 0; p0 < np; p0++)
                for (int p1 = 0; p1 < np; p1++) {
                    const int i0 = antpol_to_input[s0 * np + p0], i1 = antpol_to_input[s1 * np + p1];
                    const size_t k = (((size_t)s0 * ns + s1) * np + p0) * np + p1;
                    // stored word at regtile_index(lo,hi) is conj(x[lo])*x[hi] (xgpu_test.py:111-131);
                    // the consumer wants x[s0,p0]*conj(x[s1,p1]) (corr_output_full_block.py:582-591)
                    if (i1 >= i0) { antpol_to_bl[k] = (int32_t)regtile_index_host(i0, i1, ns); is_conj[k] = 1; }
                    else          { antpol_to_bl[k] = (int32_t)regtile_index_host(i1, i0, ns); is_conj[k] = 0; }
                }
    return -1;
}

// bfXgpuReorder (corr_output_full_block.py:669): xGPU-order planes -> int32[nbl][nchan][2] (re, im).  Returns the index
// of the first baseline whose word is out of range, or -1.
inline long reorder_host(const int32_t* xg, int32_t* out, const int32_t* bl, const int32_t* conj, size_t nbl, int nchan,
                         int64_t per_chan, int64_t matlen) {
    for (size_t k = 0; k < nbl; k++) {
        if (bl[k] < 0 || bl[k] >= per_chan) return (long)k;
        int32_t* o = out + k * nchan * 2;
        for (int c = 0; c < nchan; c++) {
            const int64_t w = (int64_t)c * per_chan + bl[k];
            o[2 * c] = xg[w];
            o[2 * c + 1] = conj[k] ? -xg[matlen + w] : xg[matlen + w];
        }
    }
    return -1;
}

}  // namespace xeng
