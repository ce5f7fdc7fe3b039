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

// every (channel, tile group) appears exactly once per launch; K slices of a split item tile [0, nstage) in order
static void check_work(int ncu, int nchan, int nblk, int nstage, bool splitk, bool stagger = false) {
    const int nwg = (int)build_wg_descs(nblk).size();
    const int grid = fused_grid(nchan, nwg, ncu);
    CHECK(grid >= 1 && grid <= std::max(ncu, 1) && grid <= nchan * nwg, "grid %d (ncu %d, items %d)", grid, ncu, nchan * nwg);
    const std::vector<WgDesc> descs = build_wg_descs(nblk);
    const WorkList wl = build_work(grid, nchan, nwg, nstage, splitk, stagger, &descs);
    CHECK((int)wl.entries.size() == grid * wl.maxi, "entries");
    std::map<std::pair<int, int>, std::vector<std::pair<int, int>>> items;     // (c, wg) -> [(stage0, nst)] by slice
    std::map<std::pair<int, int>, int> nslices;
    for (int b = 0; b < grid; b++) {
        bool ended = false;
        for (int k = 0; k < wl.maxi; k++) {
            const WorkEntry& e = wl.entries[(size_t)b * wl.maxi + k];
            if (!(e.slice >> 16)) { ended = true; continue; }
            CHECK(!ended, "work-group %d: valid entry after the end of its list", b);
            const int c = e.c_wg & 0xFFFF, wg = e.c_wg >> 16, s0 = e.stages & 0xFFFF, nst = e.stages >> 16;
            const int sl = e.slice & 0xFF, ns = (e.slice >> 8) & 0xFF;
            CHECK(c < nchan && wg < nwg && nst >= 1 && s0 + nst <= nstage && sl < ns, "entry (%d,%d) stages %d+%d slice %d/%d", c, wg, s0, nst, sl, ns);
            if ((nchan & 7) == 0 && (grid & 7) == 0) CHECK((c & 7) == (b & 7), "channel %d on work-group %d: wrong XCD class", c, b);
            auto& v = items[{c, wg}];
            if ((int)v.size() <= sl) v.resize(sl + 1, {-1, -1});
            CHECK(v[sl].first < 0, "item (%d,%d) slice %d twice", c, wg, sl);
            v[sl] = {s0, nst};
            nslices[{c, wg}] = ns;
            if (ns > 1) CHECK((int)e.chain < wl.nchains, "chain %u of %d", e.chain, wl.nchains);
        }
    }
    CHECK((int)items.size() == nchan * nwg, "%zu items of %d", items.size(), nchan * nwg);
    for (auto& kv : items) {
        int pos = 0;
        CHECK((int)kv.second.size() == nslices[kv.first], "item slices");
        for (auto& sl : kv.second) { CHECK(sl.first == pos, "item (%d,%d): slice starts at %d, expected %d", kv.first.first, kv.first.second, sl.first, pos); pos += sl.second; }
        CHECK(pos == nstage, "item (%d,%d) covers %d of %d stages", kv.first.first, kv.first.second, pos, nstage);
    }
}

// channel_group_order: a permutation, never worse than the plain order; config 2 (11 blocks, 32 work-groups per XCD):
// 184 -> 171 block fetches per XCD and launch
static void check_group_order() {
    for (int nblk : {2, 3, 5, 8, 11, 12, 16}) {
        const std::vector<WgDesc> d = build_wg_descs(nblk);
        const int nwg = (int)d.size();
        auto blocks = [&](const std::vector<int>& o, int a, int b) {
            std::set<int> s;
            for (int i = a; i < b; i++)
                for (int k = 0; k < XC_NSLOT; k++) s.insert(d[o[i]].slot_blk[k]);
            return (int)s.size();
        };
        for (int W : {4, 7, 13, 32}) {
            int plain = 0, tuned = 0;
            for (int q = 0; q < 12; q++) {
                const std::vector<int> o = channel_group_order(d, q * nwg, W);
                std::vector<int> id(nwg), sorted = o;
                for (int i = 0; i < nwg; i++) id[i] = i;
                std::sort(sorted.begin(), sorted.end());
                CHECK(sorted == id, "nblk %d W %d channel %d: not a permutation", nblk, W, q);
                const int B = (q * nwg / W + 1) * W, head = std::min(nwg, B - q * nwg);
                plain += blocks(id, 0, head) + (head < nwg ? blocks(id, head, nwg) : 0);
                tuned += blocks(o, 0, head) + (head < nwg ? blocks(o, head, nwg) : 0);
            }
            CHECK(tuned <= plain, "nblk %d W %d: %d block fetches, plain order %d", nblk, W, tuned, plain);
            if (nblk == 11 && W == 32) CHECK(plain == 184 && tuned == 171, "config 2: %d -> %d block fetches per XCD", plain, tuned);
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
    const int shapes[][4] = {{256, 96, 11, 25}, {256, 96, 11, 5}, {256, 8, 2, 3}, {256, 3, 1, 1}, {64, 96, 11, 25}, {256, 5, 11, 7},
                             {304, 96, 11, 25}, {8, 16, 3, 2}, {1, 1, 1, 1}, {256, 192, 11, 50}};
    for (auto& s : shapes)
        for (int sk = 0; sk < 3; sk++) check_work(s[0], s[1], s[2], s[3], sk == 1, sk == 2);
    check_group_order();
    for (int ns : {4, 8, 16, 32, 352}) check_order(ns);
    if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    printf("tiling_check: all properties hold\n");
    return 0;
}
