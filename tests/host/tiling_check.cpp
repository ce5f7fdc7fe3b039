// Host-only property checks of the X-engine's tiling / work lists / index maps, built with
// g++ -fsanitize=address,undefined (tests/test_host_sanitizers.py).  Includes the very header libxeng compiles.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include "../../caltech-bifrost-dsp_amd/csrc/xcorr_tiling.h"

using namespace xeng;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fails++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

// every tile (a, b), a >= b, of the nblk x nblk block triangle is contracted by exactly one wave of one tile group
static void check_descs(int nblk) {
    const std::vector<WgDesc> d = build_wg_descs(nblk);
    std::map<std::pair<int, int>, int> seen;
    int slots_busy = 0;
    for (size_t g = 0; g < d.size(); g++) {
        int nw = 0;
        for (int w = 0; w < 4; w++) {
            if (d[g].wave_a[w] == 0xFF) { CHECK(d[g].wave_b[w] == 0xFF, "nblk %d group %zu wave %d half idle", nblk, g, w); continue; }
            CHECK(d[g].wave_a[w] < XC_NSLOT && d[g].wave_b[w] < XC_NSLOT, "nblk %d group %zu slot out of range", nblk, g);
            const int a = d[g].slot_blk[d[g].wave_a[w]], b = d[g].slot_blk[d[g].wave_b[w]];
            CHECK(a < nblk && b < nblk && a >= b, "nblk %d group %zu tile (%d,%d)", nblk, g, a, b);
            seen[{a, b}]++;
            nw++;
        }
        CHECK(nw == d[g].nwave && nw >= 1, "nblk %d group %zu nwave %d vs %d", nblk, g, nw, d[g].nwave);
        for (int s = 0; s < XC_NSLOT; s++) CHECK(d[g].slot_blk[s] < nblk, "nblk %d group %zu slot block", nblk, g);
        slots_busy += nw;
    }
    CHECK((int)seen.size() == nblk * (nblk + 1) / 2, "nblk %d: %zu tiles covered of %d", nblk, seen.size(), nblk * (nblk + 1) / 2);
    for (auto& kv : seen) CHECK(kv.second == 1, "nblk %d tile (%d,%d) covered %d times", nblk, kv.first.first, kv.first.second, kv.second);
    CHECK(slots_busy == nblk * (nblk + 1) / 2, "nblk %d slots", nblk);
}

// fragment-level tiling: every 32x32 cell of the triangle is live in exactly one wave, every operand sits in a staged
// block, Z waves hold a diagonal 64x64 tile; 704 inputs (11 blocks) come out as 16 groups with 253 cells in 256 slots
static void check_frag(int nblk) {
    const std::vector<FragGroup> gs = build_frag_groups(nblk);
    CHECK(check_frag_groups(gs, nblk) == -1, "nblk %d: fragment tiling is not an exact cover (group %d)", nblk, check_frag_groups(gs, nblk));
    CHECK(check_frag_groups(frag_groups_from_tiles(nblk), nblk) == -1, "nblk %d: 64x64 tiling in fragment form", nblk);
    CHECK(gs.size() <= build_wg_descs(nblk).size(), "nblk %d: %zu groups, 64x64 tiling %zu", nblk, gs.size(), build_wg_descs(nblk).size());
    std::map<std::pair<int, int>, int> seen;
    int live_cells = 0;
    for (size_t g = 0; g < gs.size(); g++) {
        for (int w = 0; w < 4; w++) {
            const uint32_t ww = gs[g].wave[w];
            int row[4], col[4];
            frag_wave_cells(gs[g], w, row, col);
            for (int p = 0; p < 4; p++)
                if ((ww >> (16 + p)) & 1) { seen[{row[p], col[p]}]++; live_cells++; }
            if (ww & FRAG_Z) {
                CHECK(row[0] == col[0] && row[3] == col[3] && row[2] == row[3] && col[2] == col[0] && row[3] == row[0] + 1 && !(row[0] & 1),
                      "nblk %d group %zu wave %d: Z wave is not a diagonal tile", nblk, g, w);
                CHECK(((ww >> 16) & 13) == 13, "nblk %d group %zu wave %d: Z wave with a dead diagonal cell", nblk, g, w);
            }
        }
    }
    const int n32 = 2 * nblk;
    CHECK(live_cells == n32 * (n32 + 1) / 2 && (int)seen.size() == live_cells, "nblk %d: %d live cells, %zu distinct, %d needed", nblk, live_cells, seen.size(), n32 * (n32 + 1) / 2);
    if (nblk == 11) CHECK(gs.size() == 16 && live_cells == 253, "config 2: %zu groups, %d cells", gs.size(), live_cells);
    // a corrupted tiling is caught: kill one live cell / duplicate one
    if (!gs.empty()) {
        std::vector<FragGroup> bad = gs;
        bad[0].wave[0] &= ~(1u << 16);
        CHECK(check_frag_groups(bad, nblk) != -1, "nblk %d: missing cell not reported", nblk);
        bad = gs;
        bad.push_back(gs[0]);
        CHECK(check_frag_groups(bad, nblk) != -1, "nblk %d: duplicated cells not reported", nblk);
    }
}

// every (channel, tile group) appears exactly once per launch, on the right XCD class, lists end once
static void check_work(int ncu, int nchan, int nblk) {
    const std::vector<FragGroup> gs = build_frag_groups(nblk);
    const int nwg = (int)gs.size();
    const int grid = fused_grid(nchan, nwg, ncu);
    CHECK(grid >= 1 && grid <= std::max(ncu, 1) && grid <= nchan * nwg, "grid %d (ncu %d, items %d)", grid, ncu, nchan * nwg);
    const std::vector<uint64_t> masks = group_block_masks(gs);
    // ... and the channel-pair order of the packet-slab launches: the same exact cover, neighbouring channels per XCD and round
    if ((nchan & 15) == 0 && (grid & 7) == 0) {
        const WorkList wp = build_work(grid, nchan, nwg, &masks, true);
        std::map<std::pair<int, int>, int> seen;
        for (int b = 0; b < grid; b++)
            for (int k = 0; k < wp.maxi; k++) {
                const WorkEntry e = wp.entries[(size_t)b * wp.maxi + k];
                if (!(e & WORK_VALID)) continue;
                const int c = e & 0xFFFF, wg = (e >> 16) & 0x7FFF;
                CHECK(c < nchan && wg < nwg, "pair order: entry (%d,%d)", c, wg);
                CHECK(((c >> 1) & 7) == (b & 7), "pair order: channel %d on work-group %d: wrong XCD class", c, b);
                seen[{c, wg}]++;
            }
        CHECK((int)seen.size() == nchan * nwg, "pair order: %zu items of %d", seen.size(), nchan * nwg);
        for (auto& kv : seen) CHECK(kv.second == 1, "pair order: item (%d,%d) %d times", kv.first.first, kv.first.second, kv.second);
        if (ncu == 256 && nchan == 96 && nblk == 11)
            for (int x = 0; x < 8; x++)
                for (int k = 0; k < 6; k++) {
                    std::set<int> chans;
                    for (int j = 0; j < 32; j++) chans.insert(wp.entries[(size_t)(j * 8 + x) * wp.maxi + k] & 0xFFFF);
                    CHECK(chans.size() == 2 && (*chans.begin() ^ 1) == *chans.rbegin(), "config 2, pair order: round %d of XCD %d does not hold two neighbouring channels", k, x);
                }
    }
    const WorkList wl = build_work(grid, nchan, nwg, &masks);
    CHECK((int)wl.entries.size() == grid * wl.maxi, "entries");
    std::map<std::pair<int, int>, int> items;
    int longest = 0, shortest = 1 << 30;
    for (int b = 0; b < grid; b++) {
        bool ended = false;
        int n = 0;
        for (int k = 0; k < wl.maxi; k++) {
            const WorkEntry e = wl.entries[(size_t)b * wl.maxi + k];
            if (!(e & WORK_VALID)) { ended = true; continue; }
            CHECK(!ended, "work-group %d: valid entry after the end of its list", b);
            const int c = e & 0xFFFF, wg = (e >> 16) & 0x7FFF;
            CHECK(c < nchan && wg < nwg, "entry (%d,%d)", c, wg);
            if ((nchan & 7) == 0 && (grid & 7) == 0) CHECK((c & 7) == (b & 7), "channel %d on work-group %d: wrong XCD class", c, b);
            items[{c, wg}]++;
            n++;
        }
        longest = std::max(longest, n); shortest = std::min(shortest, n);
    }
    CHECK((int)items.size() == nchan * nwg, "%zu items of %d", items.size(), nchan * nwg);
    for (auto& kv : items) CHECK(kv.second == 1, "item (%d,%d) %d times", kv.first.first, kv.first.second, kv.second);
    CHECK(longest - shortest <= 1, "lists of %d and %d items", shortest, longest);
    // config 2 on 256 CUs: six items for every work-group, and every round of an XCD's 32 work-groups is two whole channels
    if (ncu == 256 && nchan == 96 && nblk == 11) {
        CHECK(longest == 6 && shortest == 6, "config 2: %d..%d items per work-group", shortest, longest);
        for (int x = 0; x < 8; x++)
            for (int k = 0; k < 6; k++) {
                std::set<int> chans;
                for (int j = 0; j < 32; j++) chans.insert(wl.entries[(size_t)(j * 8 + x) * wl.maxi + k] & 0xFFFF);
                CHECK(chans.size() == 2, "config 2: round %d of XCD %d touches %zu channels", k, x, chans.size());
            }
    }
}

// channel_group_order: a permutation, never worse than the plain order
static void check_group_order() {
    for (int nblk : {2, 3, 5, 8, 11, 12, 16}) {
        for (int frag = 0; frag < 2; frag++) {
            const std::vector<FragGroup> d = frag ? build_frag_groups(nblk) : frag_groups_from_tiles(nblk);
            const std::vector<uint64_t> masks = group_block_masks(d);
            const int nwg = (int)d.size();
            auto blocks = [&](const std::vector<int>& o, int a, int b) {
                uint64_t s = 0;
                for (int i = a; i < b; i++) s |= masks[o[i]];
                return __builtin_popcountll(s);
            };
            for (int W : {4, 7, 13, 32}) {
                int plain = 0, tuned = 0;
                for (int q = 0; q < 12; q++) {
                    const std::vector<int> o = channel_group_order(masks, q * nwg, W);
                    std::vector<int> id(nwg), sorted = o;
                    for (int i = 0; i < nwg; i++) id[i] = i;
                    std::sort(sorted.begin(), sorted.end());
                    CHECK(sorted == id, "nblk %d W %d channel %d: not a permutation", nblk, W, q);
                    const int B = (q * nwg / W + 1) * W, head = std::min(nwg, B - q * nwg);
                    plain += blocks(id, 0, head) + (head < nwg ? blocks(id, head, nwg) : 0);
                    tuned += blocks(o, 0, head) + (head < nwg ? blocks(o, head, nwg) : 0);
                }
                CHECK(tuned <= plain, "nblk %d W %d: %d block fetches, plain order %d", nblk, W, tuned, plain);
                // the 17-group 64x64 tiling of config 2 needed 184 -> 171 block fetches per XCD and launch; the 16 groups
                // of the fragment tiling are never split over two rounds: 12 channels x 11 blocks, the unavoidable minimum
                if (nblk == 11 && W == 32) CHECK(frag ? (plain == 132 && tuned == 132) : (plain == 184 && tuned == 171), "config 2 (%d): %d -> %d block fetches per XCD", frag, plain, tuned);
            }
        }
    }
}

static void check_order(int ns) {
    const int np = 2, ninput = ns * np;
    std::vector<int32_t> a2i(ninput), bl((size_t)ns * ns * np * np), cj(bl.size());
    for (int k = 0; k < ninput; k++) a2i[k] = (k * 7 + 3) % ninput;               // a permutation when gcd(7, ninput) = 1
    if (ninput % 7 == 0) for (int k = 0; k < ninput; k++) a2i[k] = ninput - 1 - k;
    CHECK(get_order_host(a2i.data(), bl.data(), cj.data(), ns, np) == -1, "get_order failed");
    const int64_t per_chan = (int64_t)(ns / 2 + 1) * (ns / 4) * np * np * 4;
    std::set<int64_t> words;
    for (size_t k = 0; k < bl.size(); k++) { CHECK(bl[k] >= 0 && bl[k] < per_chan, "word %d out of the plane", bl[k]); words.insert(bl[k]); }
    // every unordered input pair has its own word (pairs (i,j) and (j,i) share it with opposite conjugation)
    CHECK((int64_t)words.size() == (int64_t)ninput * (ninput + 1) / 2, "%zu distinct words for %d inputs", words.size(), ninput);
    a2i[0] = ninput;
    CHECK(get_order_host(a2i.data(), bl.data(), cj.data(), ns, np) == 0, "out-of-range input id not reported");
    // reorder: a plane holding its own word index comes back as [bl, +-(bl + matlen)]
    const int nchan = 3;
    const int64_t matlen = per_chan * nchan;
    std::vector<int32_t> xg(2 * matlen), out(bl.size() * nchan * 2);
    for (int64_t w = 0; w < 2 * matlen; w++) xg[w] = (int32_t)w;
    a2i[0] = 3 % ninput;
    get_order_host(a2i.data(), bl.data(), cj.data(), ns, np);
    CHECK(reorder_host(xg.data(), out.data(), bl.data(), cj.data(), bl.size(), nchan, per_chan, matlen) == -1, "reorder failed");
    for (size_t k = 0; k < bl.size(); k += 97)
        for (int c = 0; c < nchan; c++) {
            const int64_t w = c * per_chan + bl[k];
            CHECK(out[(k * nchan + c) * 2] == w && out[(k * nchan + c) * 2 + 1] == (cj[k] ? -(matlen + w) : matlen + w), "reorder value");
        }
    bl[5] = (int32_t)per_chan;
    CHECK(reorder_host(xg.data(), out.data(), bl.data(), cj.data(), bl.size(), nchan, per_chan, matlen) == 5, "bad baseline not reported");
}

int main() {
    for (int nblk = 1; nblk <= 24; nblk++) check_descs(nblk);
    for (int nblk = 1; nblk <= 40; nblk++) check_frag(nblk);
    const int shapes[][3] = {{256, 96, 11}, {256, 8, 2}, {256, 3, 1}, {64, 96, 11}, {256, 5, 11}, {304, 96, 11}, {8, 16, 3}, {1, 1, 1},
                             {256, 192, 11}, {256, 96, 5}, {256, 24, 7}};
    for (auto& s : shapes) check_work(s[0], s[1], s[2]);
    check_group_order();
    for (int ns : {4, 8, 16, 32, 352}) check_order(ns);
    if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    printf("tiling_check: all properties hold\n");
    return 0;
}
