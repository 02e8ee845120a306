// CPU-only check of the host-side layout builder of the wide reduced system (ceres_slam_amd/csrc/ssba_wide_layout.cpp),
// built by tests/test_host_layout.py with -fsanitize=address,undefined (and once with -fsanitize=thread).  Random banded
// problems: every (landmark, free pose) observation must sit in exactly one slot of its item, every co-visible pose pair
// must find its block with that item's contribution, every pose its gradient rows.  Exit code 0 = all invariants hold.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>

#include "ssba_types.h"
#include "ssba_wide_layout.h"

using namespace ssba;

static int fail(const char *what, int seed) { printf("FAIL (seed %d): %s\n", seed, what); return 1; }

static int run(int seed, int P, int Lact, int max_track, int n_const) {
    std::mt19937 rng(seed);
    std::vector<int> pose_free(P, -1);
    int nfree = 0;
    for (int k = 0; k < P; ++k) pose_free[k] = k < n_const ? -1 : nfree++;
    struct Lm { std::vector<uint32_t> poses; int flo, fhi; };
    std::vector<Lm> lms(Lact);
    for (auto &l : lms) {
        const int len = 2 + (int)(rng() % (uint32_t)(max_track - 1));
        const int first = (int)(rng() % (uint32_t)std::max(1, P - len + 1));
        l.flo = 1 << 30; l.fhi = -1;
        for (int k = first; k < std::min(P, first + len); ++k) {
            if (rng() % 5 == 0 && k != first) continue;       // a dropped frame
            l.poses.push_back((uint32_t)k);
            if (pose_free[k] >= 0) { l.flo = std::min(l.flo, pose_free[k]); l.fhi = std::max(l.fhi, pose_free[k]); }
        }
        if (l.fhi < 0) l.flo = -1;
    }
    std::sort(lms.begin(), lms.end(), [](const Lm &a, const Lm &b) {
        const uint32_t fa = a.flo < 0 ? 0xFFFFFFFFu : (uint32_t)a.flo, fb = b.flo < 0 ? 0xFFFFFFFFu : (uint32_t)b.flo;
        if (fa != fb) return fa < fb;
        return a.fhi < b.fhi;
    });
    const uint32_t Lpad = std::max(256, (Lact + 255) / 256 * 256);
    std::vector<uint32_t> lm_start(Lpad + 1, 0), obs_pose;
    for (int l = 0; l < Lact; ++l) { obs_pose.insert(obs_pose.end(), lms[l].poses.begin(), lms[l].poses.end()); lm_start[l + 1] = (uint32_t)obs_pose.size(); }
    for (uint32_t l = Lact; l < Lpad; ++l) lm_start[l + 1] = lm_start[Lact];
    WideLayout w;
    if (!build_wide_layout(nfree, (uint32_t)Lact, Lpad, lm_start.data(), obs_pose.data(), pose_free.data(), 1 + rng() % 40, w))
        return max_track > WSP ? 0 : fail("builder rejected a problem whose tracks fit the window", seed);
    if (w.n != std::max(1, (nfree + WSP - 1) / WSP)) return fail("super-block count", seed);
    std::vector<uint32_t> item_of(Lact, 0xFFFFFFFFu);
    for (uint32_t it = 0; it < w.n_items; ++it) {
        if (w.item_begin[it] >= w.item_end[it]) return fail("empty item", seed);
        if (it && w.item_begin[it] < w.item_end[it - 1]) return fail("items overlap", seed);
        for (uint32_t l = w.item_begin[it]; l < w.item_end[it]; ++l) item_of[l] = it;
    }
    std::set<std::pair<uint32_t, uint32_t>> blocks;
    for (size_t b = 0; b < w.blk_a.size(); ++b) {
        if (w.blk_a[b] > w.blk_b[b] || w.blk_b[b] - w.blk_a[b] >= (uint32_t)WSP || w.blk_b[b] >= (uint32_t)nfree) return fail("block range", seed);
        if (!blocks.insert({w.blk_a[b], w.blk_b[b]}).second) return fail("duplicate block", seed);
        if (b && std::make_pair(w.blk_a[b - 1], w.blk_b[b - 1]) >= std::make_pair(w.blk_a[b], w.blk_b[b])) return fail("blocks unsorted", seed);
        for (uint32_t i = w.blk_start[b]; i < w.blk_start[b + 1]; ++i) {
            const uint32_t it = w.blk_contrib[i] / (WSP * WSP), sp = w.blk_contrib[i] % (WSP * WSP), sa = sp / WSP, sb = sp % WSP;
            if (it >= w.n_items || w.item_base[it] + sa != w.blk_a[b] || w.item_base[it] + sb != w.blk_b[b]) return fail("contribution does not belong to its block", seed);
            if (i > w.blk_start[b] && w.blk_contrib[i - 1] / (WSP * WSP) >= it) return fail("contributions of a block not in item order", seed);
        }
    }
    for (int f = 0; f < nfree; ++f) if (!blocks.count({(uint32_t)f, (uint32_t)f})) return fail("a free pose without its diagonal block", seed);
    for (int l = 0; l < Lact; ++l) {
        std::vector<int> fs;
        for (uint32_t e = lm_start[l]; e < lm_start[l + 1]; ++e) if (pose_free[obs_pose[e]] >= 0) fs.push_back(pose_free[obs_pose[e]]);
        if (fs.empty()) { if (item_of[l] != 0xFFFFFFFFu) return fail("a landmark without free poses inside an item", seed); continue; }
        const uint32_t it = item_of[l];
        if (it == 0xFFFFFFFFu) return fail("a landmark with free poses outside every item", seed);
        const int base = (int)w.item_base[it];
        int found = 0;
        for (int s = 0; s < WSP; ++s) {
            const uint32_t e = w.slot_obs[(size_t)l * WSP + s];
            if (e == 0xFFFFFFFFu) continue;
            if (e < lm_start[l] || e >= lm_start[l + 1] || pose_free[obs_pose[e]] != base + s) return fail("slot table entry", seed);
            ++found;
        }
        if (found != (int)fs.size()) return fail("observations missing from the slot table", seed);
        for (int a : fs) {
            bool row = false;
            for (uint32_t j = w.prow_start[a]; j < w.prow_start[a + 1]; ++j) row = row || w.prow_contrib[j] == it * WSP + (uint32_t)(a - base);
            if (!row) return fail("gradient row of a pose misses an item", seed);
            for (int b : fs) {
                if (b < a) continue;
                bool ok = false;
                for (size_t q = 0; q < w.blk_a.size() && !ok; ++q)
                    if ((int)w.blk_a[q] == a && (int)w.blk_b[q] == b)
                        for (uint32_t i = w.blk_start[q]; i < w.blk_start[q + 1]; ++i)
                            ok = ok || w.blk_contrib[i] == it * (uint32_t)(WSP * WSP) + (uint32_t)((a - base) * WSP + (b - base));
                if (!ok) return fail("a co-visible pose pair misses its item's contribution", seed);
            }
        }
    }
    return 0;
}

int main() {
    int bad = 0;
    for (int seed = 1; seed <= 40 && !bad; ++seed) bad |= run(seed, 5 + seed * 3, 20 + 7 * seed, 2 + seed % 23, seed % 3);
    bad |= run(100, 2, 30, 2, 0);
    bad |= run(101, 30, 40, 40, 1);        // tracks beyond the window: the builder must refuse
    bad |= run(102, 300, 4000, 24, 1);
    if (!bad) printf("wide layout: all invariants hold\n");
    return bad;
}
