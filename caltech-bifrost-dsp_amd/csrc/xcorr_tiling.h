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

// One entry of a persistent work-group's list: a (channel, tile group) item, all K stages of it.
//   channel | tile group << 16 | valid << 31
typedef uint32_t WorkEntry;
constexpr uint32_t WORK_VALID = 1u << 31;

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
// Fragment-level tiling of the fused kernel (round 3).  The unit is the 32x32-input MFMA tile ("cell" (i, j) of the
// triangle of 32-input blocks, j <= i); a wave contracts four cells from four operand fragments, in one of two patterns:
//   2x2   operands a0 a1 | b0 b1:  cells (a0,b0) (a0,b1) (a1,b0) (a1,b1)           -- a 64x64 off-diagonal tile
//   Z     operands d0 d1 | r  c :  cells (d0,d0) (r,c) (d1,d0) (d1,d1)             -- a diagonal 64x64 tile, whose
//         upper-right cell is never stored, plus one FREE cell (r, c) taken from an off-diagonal tile
// (same cost either way: 4 fragments read and unpacked, 16 MFMAs per K-tile).  A tile group = 4 waves sharing <= 4 staged
// 64-input blocks.  With nblk64 = 2 np + 1 blocks (L = the last one):
//   * squares: blocks {2k, 2k+1} x {2m, 2m+1}, k > m: four whole off-diagonal tiles
//   * per k: blocks {2k, 2k+1, L (, 4j+1)}: Z(2k), Z(2k+1), tile (2k+1, 2k), tile (L, 2k); the two free cells come from a
//     DONOR tile: k = 2j and k = 2j+1 share the donor (L, 4j+1), and (L, 4j+3) stays whole; an unpaired last k takes two
//     cells of (L, 2k+1) and leaves one to Z(L) and one to a wave of its own
//   * collectors: block L + up to three others: the whole tiles (L, 4j+3), Z(L), the left-over cell
// 704 inputs: 10 + 5 + 1 = 16 tile groups, 253 cells in 256 wave slots (the 64x64 tiling below: 17 groups), so a round
// of an XCD's 32 work-groups is exactly two channels: no item tail, and no channel is contracted in two parts.
// Even block counts (and whatever the construction above does not improve) use the 64x64 tiling, one 2x2 wave per tile.
// --------------------------------------------------------------------------------------
struct FragGroup {               // read by the device as 8 aligned dwords (scalar loads)
    uint8_t slot_blk[XC_NSLOT];  // 64-input block staged in LDS slot s (slots are staged in pairs 0|1 and 2|3)
    uint32_t wave[4];            // see frag_wave()
    uint32_t pad[3];
};
constexpr uint32_t FRAG_Z = 1u << 12;       // Z pattern
constexpr uint32_t FRAG_BUSY = 1u << 31;    // the wave has at least one live cell
// operand position = slot * 2 + 32-input half; pos[] = a0 a1 b0 b1 (Z: d0 d1 r c); live bit p = 2m + n <-> accumulator
// (m, n): 2x2 cell (a_m, b_n); Z: p = 0 (d0,d0), 1 (r,c), 2 (d1,d0), 3 (d1,d1)
inline uint32_t frag_wave(const int pos[4], bool z, int live) {
    uint32_t w = 0;
    for (int q = 0; q < 4; q++) w |= (uint32_t)(pos[q] & 7) << (3 * q);
    return w | (z ? FRAG_Z : 0) | ((uint32_t)(live & 15) << 16) | (live ? FRAG_BUSY : 0);
}
// the four cells of a wave as (row 32-block, column 32-block), in accumulator order p = 2m + n
inline void frag_wave_cells(const FragGroup& g, int w, int (&row)[4], int (&col)[4]) {
    int b32[4];
    for (int q = 0; q < 4; q++) {
        const int pos = (g.wave[w] >> (3 * q)) & 7;
        b32[q] = g.slot_blk[pos >> 1] * 2 + (pos & 1);
    }
    if (g.wave[w] & FRAG_Z) {
        row[0] = b32[0]; col[0] = b32[0]; row[1] = b32[2]; col[1] = b32[3];
        row[2] = b32[1]; col[2] = b32[0]; row[3] = b32[1]; col[3] = b32[1];
    } else {
        for (int p = 0; p < 4; p++) { row[p] = b32[p >> 1]; col[p] = b32[2 + (p & 1)]; }
    }
}

namespace detail {
struct GroupBuilder {
    FragGroup g;
    int nslot = 0, nwave = 0;
    GroupBuilder() { memset(&g, 0, sizeof(g)); }
    int slot_of(int blk) {                       // -1: no slot left
        for (int s = 0; s < nslot; s++) if (g.slot_blk[s] == blk) return s;
        if (nslot == XC_NSLOT) return -1;
        g.slot_blk[nslot] = (uint8_t)blk;
        return nslot++;
    }
    bool fits(const std::vector<int>& blks) const {
        int need = 0;
        for (size_t i = 0; i < blks.size(); i++) {
            bool have = false;
            for (int s = 0; s < nslot; s++) have = have || g.slot_blk[s] == blks[i];
            for (size_t k = 0; k < i; k++) have = have || blks[k] == blks[i];
            need += have ? 0 : 1;
        }
        return nwave < 4 && nslot + need <= XC_NSLOT;
    }
    int pos32(int b32) { return slot_of(b32 >> 1) * 2 + (b32 & 1); }
    // whole (or partly live) off-diagonal 64x64 tile (A, B): cells (2A+m, 2B+n)
    void tile(int A, int B, int live = 15) {
        const int pos[4] = {pos32(2 * A), pos32(2 * A + 1), pos32(2 * B), pos32(2 * B + 1)};
        g.wave[nwave++] = frag_wave(pos, false, live);
    }
    // diagonal tile of block D plus the free cell (r32, c32) (r32 < 0: none)
    void ztile(int D, int r32, int c32) {
        const int d0 = pos32(2 * D), d1 = pos32(2 * D + 1);
        const int pos[4] = {d0, d1, r32 < 0 ? d0 : pos32(r32), r32 < 0 ? d1 : pos32(c32)};
        g.wave[nwave++] = frag_wave(pos, true, r32 < 0 ? 13 : 15);
    }
    FragGroup finish() {
        for (int s = nslot; s < XC_NSLOT; s++) g.slot_blk[s] = g.slot_blk[0];   // harmless duplicate loads
        return g;
    }
};
}  // namespace detail

// the 64x64 tiling above in fragment form
inline std::vector<FragGroup> frag_groups_from_tiles(int nblk64) {
    std::vector<FragGroup> out;
    for (const WgDesc& d : build_wg_descs(nblk64)) {
        FragGroup g;
        memset(&g, 0, sizeof(g));
        memcpy(g.slot_blk, d.slot_blk, XC_NSLOT);
        for (int w = 0; w < 4; w++) {
            if (d.wave_a[w] == 0xFF) continue;
            const int a = d.wave_a[w], b = d.wave_b[w];
            const int pos[4] = {2 * a, 2 * a + 1, 2 * b, 2 * b + 1};
            g.wave[w] = frag_wave(pos, false, d.slot_blk[a] == d.slot_blk[b] ? 13 : 15);
        }
        out.push_back(g);
    }
    return out;
}

inline std::vector<FragGroup> build_frag_groups(int nblk64) {
    using detail::GroupBuilder;
    std::vector<FragGroup> tiles = frag_groups_from_tiles(nblk64);
    if (!(nblk64 & 1) || nblk64 < 3) return tiles;
    std::vector<FragGroup> out;
    const int np = nblk64 / 2, L = nblk64 - 1;
    for (int k = 1; k < np; k++)
        for (int m = 0; m < k; m++) {
            GroupBuilder b;
            b.slot_of(2 * k); b.slot_of(2 * k + 1); b.slot_of(2 * m); b.slot_of(2 * m + 1);   // adjacent pairs: whole 128-byte lines
            b.tile(2 * k, 2 * m); b.tile(2 * k, 2 * m + 1); b.tile(2 * k + 1, 2 * m); b.tile(2 * k + 1, 2 * m + 1);
            out.push_back(b.finish());
        }
    // cell q = 2 (row half) + (column half) of the off-diagonal tile (L, t)
    auto cell_r = [&](int q) { return 2 * L + (q >> 1); };
    auto cell_c = [&](int t, int q) { return 2 * t + (q & 1); };
    std::vector<int> whole;                               // column blocks t of the tiles (L, t) left whole
    for (int k = 0; k < np; k++) {
        const int j = k >> 1;
        const bool paired = 2 * j + 1 < np;
        const int donor = paired ? 4 * j + 1 : 2 * k + 1;
        const int q0 = (paired && (k & 1)) ? 2 : 0;      // the odd member of a pair takes the donor's second row
        GroupBuilder b;
        b.slot_of(2 * k); b.slot_of(2 * k + 1);           // the adjacent pair first: its rows are whole 128-byte lines
        b.ztile(2 * k, cell_r(q0), cell_c(donor, q0));
        b.ztile(2 * k + 1, cell_r(q0 + 1), cell_c(donor, q0 + 1));
        b.tile(2 * k + 1, 2 * k);
        b.tile(L, 2 * k);
        out.push_back(b.finish());
        if (paired && (k & 1)) whole.push_back(2 * k + 1);
    }
    // collectors
    std::vector<GroupBuilder> coll;
    auto place = [&](const std::vector<int>& blks) -> GroupBuilder& {
        for (auto& c : coll) if (c.fits(blks)) return c;
        coll.emplace_back();
        if (blks.size() > 1 && blks[1] == L - 1) coll.back().slot_of(L - 1);        // (L-1, L) in memory order
        coll.back().slot_of(L);
        return coll.back();
    };
    if (np & 1) {                                         // the unpaired k left cells 2 and 3 of (L, L-1)
        GroupBuilder& c = place({L, L - 1});
        c.ztile(L, cell_r(2), cell_c(L - 1, 2));
        GroupBuilder& c2 = place({L, L - 1});
        c2.tile(L, L - 1, 1 << 3);
    } else if (!whole.empty()) {                          // Z(L) takes cell 0 of one of the whole tiles
        const int t = whole.back();
        whole.pop_back();
        GroupBuilder& c = place({L, t});
        c.ztile(L, cell_r(0), cell_c(t, 0));
        GroupBuilder& c2 = place({L, t});
        c2.tile(L, t, 14);
    } else {
        place({L}).ztile(L, -1, -1);
    }
    for (int t : whole) place({L, t}).tile(L, t);
    for (auto& c : coll) out.push_back(c.finish());
    return out.size() < tiles.size() ? out : tiles;
}

// every cell (i, j), j <= i < n32, is live in exactly one wave; operands sit in staged blocks; -1 or the first bad group
inline int check_frag_groups(const std::vector<FragGroup>& gs, int nblk64) {
    const int n32 = 2 * nblk64;
    std::vector<int> seen((size_t)n32 * n32, 0);
    for (size_t gi = 0; gi < gs.size(); gi++) {
        for (int s = 0; s < XC_NSLOT; s++) if (gs[gi].slot_blk[s] >= nblk64) return (int)gi;
        for (int w = 0; w < 4; w++) {
            const uint32_t ww = gs[gi].wave[w];
            const int live = (ww >> 16) & 15;
            if (!!(ww & FRAG_BUSY) != (live != 0)) return (int)gi;
            int row[4], col[4];
            frag_wave_cells(gs[gi], w, row, col);
            for (int p = 0; p < 4; p++) {
                if (!((live >> p) & 1)) continue;
                if (row[p] >= n32 || col[p] > row[p]) return (int)gi;
                if (seen[(size_t)row[p] * n32 + col[p]]++) return (int)gi;
            }
        }
    }
    for (int i = 0; i < n32; i++)
        for (int j = 0; j <= i; j++) if (seen[(size_t)i * n32 + j] != 1) return (int)gs.size();
    return -1;
}

// --------------------------------------------------------------------------------------
// Work lists of the persistent fused kernel (WorkEntry[grid][maxi], xcorr_kernels.h).
// Work-groups b with the same b & 7 sit on one XCD and share that XCD's items (channels = xcd mod 8), dealt
// round-robin so that concurrent work-groups contract neighbouring tile groups of the same channels.  With
// n items for W work-groups every work-group gets n / W whole items and the first n % W one more
// (704 inputs x 96 channels on 256 CUs: 12 x 16 = 192 items per XCD for 32 work-groups = 6 each, two whole channels
// per round).
// --------------------------------------------------------------------------------------
struct WorkList {
    std::vector<WorkEntry> entries;
    int maxi = 0;
    uint32_t* dev = nullptr;
};

// Order of a channel's tile groups in its XCD's item list.  The W work-groups of an XCD contract W consecutive items
// at a time ("a round"); a channel whose nwg items straddle a round boundary is contracted in two parts, about one
// item time apart, and whatever input blocks the second part needs have left the XCD's L2 by then.  So the groups of
// such a channel are ordered to make (distinct blocks of the first part) + (distinct blocks of the second part) minimal.
// Exhaustive over the subsets of the smaller part; identity when that would be too many, when the channel spans more
// than two rounds, or when it is not split (the 16 groups of 704 inputs never are).  start = index of the channel's
// first item.  blocks[g] = bit mask of the 64-input blocks group g stages.
inline std::vector<int> channel_group_order(const std::vector<uint64_t>& mask, int start, int W) {
    const int nwg = (int)mask.size();
    std::vector<int> order(nwg);
    for (int i = 0; i < nwg; i++) order[i] = i;
    const int B = (start / W + 1) * W;                      // first round boundary behind `start`
    const int head = B - start, tail = nwg - head;
    if (tail <= 0 || tail > W || nwg > 30) return order;
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
inline std::vector<uint64_t> group_block_masks(const std::vector<FragGroup>& gs) {
    std::vector<uint64_t> m(gs.size(), 0);
    for (size_t g = 0; g < gs.size(); g++)
        for (int s = 0; s < XC_NSLOT; s++) m[g] |= 1ull << (gs[g].slot_blk[s] & 63);
    return m;
}

// masks: per tile group the blocks it stages (null: groups in their natural order)
// pair_channels (packet slabs, round 4): the q-th channel of XCD x is 16 (q / 2) + 2 x + (q & 1) instead of x + 8 q, so that the two
// channels an XCD contracts in one round are NEIGHBOURS: in a packet the 64 bytes of channel c and of channel c + 1 share a
// 128-byte line, and a line should pass through one L2, once (needs nchan % 16 == 0)
inline WorkList build_work(int grid, int nchan, int nwg, const std::vector<uint64_t>* masks = nullptr, bool pair_channels = false) {
    WorkList wl;
    const bool xcd_map = (nchan & 7) == 0 && (grid & 7) == 0;
    const int ngroup = xcd_map ? 8 : 1;
    const int W = grid / ngroup;
    const int n = xcd_map ? (nchan / 8) * nwg : nchan * nwg;
    const int f = n / W, r = n % W;
    wl.maxi = f + (r ? 1 : 0);
    wl.entries.assign((size_t)grid * wl.maxi, 0u);
    // (whole items dealt per XCD: the groups of a channel that is contracted in two rounds are ordered for L2 reuse)
    std::vector<std::vector<int>> gorder;
    if (masks && xcd_map && (int)masks->size() == nwg)
        for (int q = 0; q < nchan / 8; q++) gorder.push_back(channel_group_order(*masks, q * nwg, W));
    auto put = [&](int b, int k, int x, int idx) {
        const int q = idx / nwg;
        const int wg = gorder.empty() ? idx - q * nwg : gorder[q][idx - q * nwg];
        const int c = !xcd_map ? q : (pair_channels && (nchan & 15) == 0) ? 16 * (q >> 1) + 2 * x + (q & 1) : x + 8 * q;
        wl.entries[(size_t)b * wl.maxi + k] = (uint32_t)c | ((uint32_t)wg << 16) | WORK_VALID;
    };
    for (int x = 0; x < ngroup; x++) {
        auto block_of = [&](int j) { return xcd_map ? j * 8 + x : j; };
        for (int j = 0; j < W; j++)
            for (int q = 0; q < f; q++) put(block_of(j), q, x, j + q * W);
        for (int i = 0; i < r; i++) put(block_of(i), f, x, f * W + i);
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
