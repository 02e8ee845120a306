// Host-side layout of the wide reduced system (see ssba_wide_layout.h, ssba_types.h: WideSys).
#include "ssba_wide_layout.h"

#include <algorithm>

#include "ssba_types.h"

namespace ssba {

bool build_wide_layout(int nfree, uint32_t Lact, uint32_t Lpad, const uint32_t *lm_start, const uint32_t *obs_pose, const int *pose_free,
                       uint32_t max_item_landmarks, WideLayout &out) {
    out = WideLayout{};
    out.n = std::max(1, (nfree + WSP - 1) / WSP);
    out.slot_obs.assign((size_t)Lpad * WSP, 0xFFFFFFFFu);
    std::vector<int> flo(Lact, -1), fhi(Lact, -1);
    for (uint32_t l = 0; l < Lact; ++l) {
        int lo = 1 << 30, hi = -1;
        for (uint32_t e = lm_start[l]; e < lm_start[l + 1]; ++e) {
            const int f = pose_free[obs_pose[e]];
            if (f >= 0) { lo = std::min(lo, f); hi = std::max(hi, f); }
        }
        if (hi >= 0) {
            if (hi - lo > WSP - 1) return false;
            flo[l] = lo; fhi[l] = hi;
            out.bandwidth = std::max<uint32_t>(out.bandwidth, (uint32_t)(hi - lo));
        }
    }
    // items: runs of consecutive landmarks whose free poses fit [base, base + WSP)
    std::vector<uint32_t> item_of(Lact, 0xFFFFFFFFu);
    for (uint32_t l = 0; l < Lact;) {
        if (flo[l] < 0) { ++l; continue; }
        const int base = flo[l];
        uint32_t e = l;
        while (e < Lact && e - l < max_item_landmarks && flo[e] >= base && fhi[e] - base <= WSP - 1) ++e;
        const uint32_t it = (uint32_t)out.item_begin.size();
        out.item_begin.push_back(l); out.item_end.push_back(e); out.item_base.push_back((uint32_t)base);
        for (uint32_t q = l; q < e; ++q) item_of[q] = it;
        l = e;
    }
    out.n_items = (uint32_t)out.item_begin.size();
    // slot table
    for (uint32_t l = 0; l < Lact; ++l) {
        if (item_of[l] == 0xFFFFFFFFu) continue;
        const int base = (int)out.item_base[item_of[l]];
        for (uint32_t e = lm_start[l]; e < lm_start[l + 1]; ++e) {
            const int f = pose_free[obs_pose[e]];
            if (f >= 0) out.slot_obs[(size_t)l * WSP + (size_t)(f - base)] = e;
        }
    }
    // gather lists.  Keys (a, b - a) are bounded by nfree * WSP: counting sort keeps item order inside a block
    struct Contrib { uint32_t key, c; };
    std::vector<Contrib> contribs;
    std::vector<std::pair<uint32_t, uint32_t>> prow;
    std::vector<uint32_t> masks;
    for (uint32_t it = 0; it < out.n_items; ++it) {
        const int base = (int)out.item_base[it];
        masks.clear();
        uint32_t any = 0;
        for (uint32_t l = out.item_begin[it]; l < out.item_end[it]; ++l) {
            uint32_t m = 0;
            for (int s = 0; s < WSP; ++s) if (out.slot_obs[(size_t)l * WSP + s] != 0xFFFFFFFFu) m |= 1u << s;
            any |= m;
            if (std::find(masks.begin(), masks.end(), m) == masks.end()) masks.push_back(m);
        }
        for (int sa = 0; sa < WSP; ++sa)
            for (int sb = sa; sb < WSP; ++sb) {
                bool on = false;
                for (uint32_t m : masks) on = on || (((m >> sa) & 1u) && ((m >> sb) & 1u));
                if (!on) continue;
                contribs.push_back({(uint32_t)(base + sa) * WSP + (uint32_t)(sb - sa), it * (uint32_t)(WSP * WSP) + (uint32_t)(sa * WSP + sb)});
            }
        for (int s = 0; s < WSP; ++s)
            if ((any >> s) & 1u) prow.push_back({(uint32_t)(base + s), it * (uint32_t)WSP + (uint32_t)s});
    }
    {
        const size_t nkeys = (size_t)std::max(nfree, 1) * WSP;
        std::vector<uint32_t> cnt(nkeys + 1, 0);
        for (auto &c : contribs) cnt[c.key + 1]++;
        for (size_t k = 0; k < nkeys; ++k) cnt[k + 1] += cnt[k];
        std::vector<uint32_t> sorted(contribs.size());
        std::vector<uint32_t> at(cnt.begin(), cnt.end() - 1);
        for (auto &c : contribs) sorted[at[c.key]++] = c.c;
        // every free pose has its diagonal block (H_pp), whether a landmark contributes or not
        for (int f = 0; f < nfree; ++f)
            for (int dlt = 0; dlt < WSP; ++dlt) {
                const size_t k = (size_t)f * WSP + dlt;
                if (cnt[k + 1] == cnt[k] && dlt != 0) continue;
                if (f + dlt >= nfree) continue;
                out.blk_a.push_back((uint32_t)f); out.blk_b.push_back((uint32_t)(f + dlt));
                out.blk_start.push_back((uint32_t)out.blk_contrib.size());
                out.blk_contrib.insert(out.blk_contrib.end(), sorted.begin() + cnt[k], sorted.begin() + cnt[k + 1]);
            }
        out.blk_start.push_back((uint32_t)out.blk_contrib.size());
    }
    std::stable_sort(prow.begin(), prow.end(), [](const std::pair<uint32_t, uint32_t> &x, const std::pair<uint32_t, uint32_t> &y) { return x.first < y.first; });
    out.prow_start.assign((size_t)nfree + 1, 0);
    for (auto &pr : prow) out.prow_start[pr.first + 1]++;
    for (int f = 0; f < nfree; ++f) out.prow_start[f + 1] += out.prow_start[f];
    out.prow_contrib.reserve(prow.size());
    for (auto &pr : prow) out.prow_contrib.push_back(pr.second);
    return true;
}

}  // namespace ssba
