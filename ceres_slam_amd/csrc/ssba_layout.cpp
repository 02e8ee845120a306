// Host side of ssba_finalize (see ssba_layout.h).
#include "ssba_layout.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <system_error>
#include <thread>

#include "../../include/ssba.h"
#include "ssba_types.h"

namespace ssba {

// pieces [0, n) of a loop on up to 16 host threads: fn(t) for t = 0 .. nt - 1 (piece 0 on the caller's thread; pieces whose thread
// cannot be started run here too)
template <class F> static void run_pieces(int nt, F &&fn) {
    std::vector<std::thread> th;
    int started = 1;
    try {
        for (; started < nt; ++started) th.emplace_back(fn, started);
    } catch (const std::system_error &) {
    }
    fn(0);
    for (int t = started; t < nt; ++t) fn(t);
    for (auto &x : th) x.join();
}
static int host_threads(uint64_t work, uint64_t threshold) {
    const unsigned hw = std::thread::hardware_concurrency();
    return (work >= threshold && hw > 1) ? (int)std::min<unsigned>(hw, 16u) : 1;
}
// stable counting sort (keys < nkeys)
template <class T, class K> static void stable_count_sort(std::vector<T> &v, std::vector<T> &tmp, size_t nkeys, K key) {
    std::vector<uint32_t> at(nkeys + 1, 0);
    for (auto &x : v) at[(size_t)key(x) + 1]++;
    for (size_t q = 0; q < nkeys; ++q) at[q + 1] += at[q];
    tmp.resize(v.size());
    for (auto &x : v) tmp[at[(size_t)key(x)]++] = x;
    v.swap(tmp);
}

int build_layout(const LayoutInput &in, Layout &out, std::string &err, const std::function<void(const char *)> &mark) {
    out = Layout{};
    const uint32_t P = in.P, L = in.L;
    const uint64_t N = in.obs_pose.size();
    const bool ph = in.lighting;
    auto set_error = [&](const char *m) { err = m; };
    struct Phase {
        const std::function<void(const char *)> &f;
        void mark(const char *s) { if (f) f(s); }
    } phase{mark};
    // landmark-major observation lists (stable: keeps the caller's order)
    std::vector<uint32_t> lm_start(L + 1, 0), pose_cnt(P, 0);
    for (uint64_t i = 0; i < N; ++i) { lm_start[in.obs_point[i] + 1]++; pose_cnt[in.obs_pose[i]]++; }
    for (uint32_t j = 0; j < L; ++j) lm_start[j + 1] += lm_start[j];
    std::vector<uint32_t> lm_obs(N), cur(lm_start.begin(), lm_start.end() - 1);
    for (uint64_t i = 0; i < N; ++i) lm_obs[cur[in.obs_point[i]]++] = (uint32_t)i;

    // free poses: in the problem (observed) and not constant
    std::vector<uint32_t> pf_cnt(P, 0);
    // pose-only residual blocks.  A relative-pose block becomes two half entries, one on each of its non-constant
    // poses (types 2 / 3: data[12] the other pose, data[13] = 1 on the half that counts the cost, data[14] the
    // position of the other half or -1); the cross terms of the two halves are added where both are free.
    std::vector<PoseFactor> pfs = in.pose_factors;
    for (auto &f : pfs)
        if (in.pose_const[f.pose]) { set_error("a unary residual block sits on a constant pose"); return SSBA_ERR_UNSUPPORTED; }
    for (auto &rf : in.rel_factors) {
        const bool c1 = in.pose_const[rf.pose1], c2 = in.pose_const[rf.pose2];
        if (c1 && c2) { set_error("a relative-pose block between two constant poses"); return SSBA_ERR_UNSUPPORTED; }
        const int ia = (int)pfs.size(), ib = ia + (c1 ? 0 : 1);
        for (int side = 0; side < 2; ++side) {
            if (side == 0 ? c1 : c2) continue;
            PoseFactor f{};
            f.pose = side == 0 ? rf.pose1 : rf.pose2;
            f.type = 2 + side;
            f.huber = rf.huber;
            memcpy(f.data, rf.T_ref, sizeof rf.T_ref);
            f.data[12] = (double)(side == 0 ? rf.pose2 : rf.pose1);
            f.data[13] = (side == 0 || c1) ? 1.0 : 0.0;
            f.data[14] = (c1 || c2) ? -1.0 : (double)(side == 0 ? ib : ia);      // host index for now
            memcpy(f.S, rf.S, sizeof rf.S);
            pfs.push_back(f);
        }
    }
    for (auto &f : pfs) pf_cnt[f.pose]++;
    if (!pfs.empty() && (ph || in.partitioned)) {
        set_error("pose priors / sun observations / relative-pose blocks are not available with lighting terms or with the partitioned "
                  "reduced solve (ssba_set_partition); landmark sharding with the all-reduce of the reduced system takes them");
        return SSBA_ERR_UNSUPPORTED;
    }
    out.pose_free.assign(P, -1);
    out.free_pose.clear();
    for (uint32_t k = 0; k < P; ++k)
        if ((pose_cnt[k] > 0 || pf_cnt[k] > 0 || in.world_size > 1) && !in.pose_const[k]) {
            out.pose_free[k] = (int)out.free_pose.size();
            out.free_pose.push_back((int)k);
        }
    const int nfree = (int)out.free_pose.size();

    phase.mark("finalize: 1 landmark lists");
    // per-landmark sorted pose sets; envelope checks.  The windowed layout needs tracks <= TW and a pose co-visibility
    // span <= SBP (block-tridiagonal reduced system); anything else takes the general path with a dense reduced system.
    bool dense = false, span_violation = false;
    bool dense_only = false;            // something only the blocked Cholesky of the general path covers (not the 144-row super-blocks)
    int max_span = 0;                   // largest free-pose distance inside one landmark's track
    std::vector<uint32_t> wide;         // landmarks whose free poses span more than SBP
    struct LmInfo { uint32_t j, kmin, kmax; int flo, fhi; };
    std::vector<LmInfo> order;
    std::vector<uint32_t> lm_pose_sorted(N);      // per landmark (lm_start range): its poses, ascending
    std::vector<uint32_t> lm_obs_by_pose;         // lm_obs in that order, where it differs from the caller's (else empty)
    {
        // the scan of the landmarks is cut into pieces (one per host thread at large sizes: 1 M landmarks took 38 ms on one);
        // every piece collects what the single pass collected, the pieces are joined in order
        struct Piece {
            std::vector<LmInfo> order;
            std::vector<uint32_t> wide;
            std::vector<std::pair<uint32_t, uint32_t>> fix;      // (position in lm_obs, observation) of landmarks listed out of pose order
            bool dense = false, dense_only = false, span = false;
            int max_span = 0;
        };
        const int nt = host_threads(N, 400000);
        std::vector<Piece> pc((size_t)nt);
        run_pieces(nt, [&](int t) {
            Piece &a = pc[(size_t)t];
            const uint32_t j0 = (uint32_t)((uint64_t)L * (uint64_t)t / (uint64_t)nt), j1 = (uint32_t)((uint64_t)L * (uint64_t)(t + 1) / (uint64_t)nt);
            a.order.reserve(j1 - j0);
            std::vector<uint32_t> ob;
            for (uint32_t j = j0; j < j1; ++j) {
                const uint32_t n = lm_start[j + 1] - lm_start[j];
                if (n == 0) continue;
                if (n > (uint32_t)TW) a.dense = true;     // longer tracks than the window layout holds: general (dense) path
                uint32_t *ks = &lm_pose_sorted[lm_start[j]];
                bool sorted = true;
                for (uint32_t e = 0; e < n; ++e) {
                    ks[e] = in.obs_pose[lm_obs[lm_start[j] + e]];
                    if (e && ks[e] < ks[e - 1]) sorted = false;
                }
                if (!sorted) {      // rare (datasets list a landmark's observations by state): order the observation indices with the poses
                    ob.assign(lm_obs.begin() + lm_start[j], lm_obs.begin() + lm_start[j + 1]);
                    std::stable_sort(ob.begin(), ob.end(), [&](uint32_t x, uint32_t y) { return in.obs_pose[x] < in.obs_pose[y]; });
                    for (uint32_t e = 0; e < n; ++e) { ks[e] = in.obs_pose[ob[e]]; a.fix.push_back({lm_start[j] + e, ob[e]}); }
                }
                if (std::adjacent_find(ks, ks + n) != ks + n) a.dense = a.dense_only = true;     // two residual blocks on one (pose, landmark): no window slot for the second
                int flo = 1 << 30, fhi = -1;
                for (uint32_t e = 0; e < n; ++e) { const int f = out.pose_free[ks[e]]; if (f >= 0) { flo = std::min(flo, f); fhi = std::max(fhi, f); } }
                a.order.push_back({j, ks[0], ks[n - 1], fhi >= 0 ? flo : -1, fhi});
                if (fhi - flo > SBP) { a.span = true; a.wide.push_back(j); }
                if (fhi >= 0) a.max_span = std::max(a.max_span, fhi - flo);
            }
        });
        size_t n_order = 0, n_fix = 0;
        for (auto &a : pc) { n_order += a.order.size(); n_fix += a.fix.size(); }
        order.reserve(n_order);
        if (n_fix) lm_obs_by_pose = lm_obs;
        for (auto &a : pc) {
            order.insert(order.end(), a.order.begin(), a.order.end());
            wide.insert(wide.end(), a.wide.begin(), a.wide.end());
            for (auto &f : a.fix) lm_obs_by_pose[f.first] = f.second;
            dense |= a.dense; dense_only |= a.dense_only; span_violation |= a.span;
            max_span = std::max(max_span, a.max_span);
        }
    }
    if (in.points_const && !ph) {
        set_error("constant position blocks are only available with lighting terms (stage 2 of --multistage)");
        return SSBA_ERR_UNSUPPORTED;
    }
    if (const char *e = getenv("SSBA_FORCE_DENSE")) if (e[0] == '1') dense = dense_only = true;
    if (in.per_obs_S) dense = dense_only = true;     // per-block stiffness lives in the general layout only
    if (!in.rel_factors.empty()) dense = dense_only = true;     // pose-pose couplings outside the landmark structure
    // Closure border: when the only thing outside the windowed envelope is the co-visibility span of a few landmarks
    // (a loop closure: the last states see landmarks of the first ones), the far poses of those landmarks -- at most
    // NBP / 6 = 5 -- leave the chain and become a dense border of the block-tridiagonal system, solved with the
    // machinery of the free shared blocks of config 3 (ssba_border.hip).  They keep their place among the free poses
    // (numbered last), their rows of the chain system are identity rows.  Everything else takes the general path.
    int nchain = nfree;
    if (span_violation && !dense) {
        const char *e = getenv("SSBA_NO_CLOSURE_BORDER");
        bool ok = !(e && e[0] == '1') && !in.no_closure_border && !ph && in.world_size == 1 && !in.per_obs_S && in.rel_factors.empty() && pfs.empty();
        std::vector<uint8_t> is_border(nfree, 0);
        int nborder = 0;
        for (size_t q = 0; q < wide.size() && ok; ++q) {
            const uint32_t j = wide[q];
            const uint32_t *ks = &lm_pose_sorted[lm_start[j]];
            const uint32_t n = lm_start[j + 1] - lm_start[j];
            int flo = 1 << 30;
            for (uint32_t e = 0; e < n; ++e) { const int f = out.pose_free[ks[e]]; if (f >= 0) flo = std::min(flo, f); }
            for (uint32_t e = 0; e < n; ++e) {
                const int f = out.pose_free[ks[e]];
                if (f >= 0 && f - flo > SBP && !is_border[f]) { is_border[f] = 1; if (++nborder > NBP / 6) ok = false; }
            }
        }
        if (ok) {       // what is left of every landmark's pose set must fit the envelope
            std::vector<int> chain_index(nfree, -1);
            int c = 0;
            for (int f = 0; f < nfree; ++f) if (!is_border[f]) chain_index[f] = c++;
            for (uint32_t j = 0; j < L && ok; ++j) {
                int flo = 1 << 30, fhi = -1;
                for (uint32_t e = lm_start[j]; e < lm_start[j + 1]; ++e) {
                    const int f = out.pose_free[lm_pose_sorted[e]];
                    if (f >= 0 && !is_border[f]) { flo = std::min(flo, chain_index[f]); fhi = std::max(fhi, chain_index[f]); }
                }
                if (fhi - flo > SBP) ok = false;
            }
            if (ok) {   // renumber: chain poses in their order, border poses last
                nchain = c;
                std::vector<int> nf(nfree);
                int b = nchain;
                for (int f = 0; f < nfree; ++f) nf[f] = is_border[f] ? b++ : chain_index[f];
                std::vector<int> fp(nfree);
                for (int f = 0; f < nfree; ++f) fp[nf[f]] = out.free_pose[f];
                out.free_pose = fp;
                for (int f = 0; f < nfree; ++f) out.pose_free[out.free_pose[f]] = f;
            }
        }
        if (!ok) dense = true;
    }
    const int nborder = nfree - nchain;
    // Long tracks on block-cyclic machinery: a problem that left the windowed layout only because its tracks are longer than
    // TW observations (free poses within a span of WSP) keeps the general observation layout, but its reduced system is block
    // tridiagonal over super-blocks of WSP = 24 poses: matrix-core Schur items + parallel cyclic reduction on 144-row blocks
    // (ssba_wide.hip) instead of the blocked Cholesky.  SSBA_NO_WIDE=1 keeps the blocked Cholesky (A/B, tests).
    bool wide_sys = false;
    if (dense && !dense_only && !ph && !in.no_wide && max_span <= WSP - 1 && nfree > 0) {
        const char *e = getenv("SSBA_NO_WIDE");
        wide_sys = !(e && e[0] == '1');
    }
    if (wide_sys && in.partitioned) {
        set_error("ssba_set_partition: the partitioned reduced solve is built for the windowed layout (tracks <= SSBA_MAX_TRACK); long tracks "
                  "shard with the all-reduce of the reduced system (no partition)");
        return SSBA_ERR_UNSUPPORTED;
    }
    if (dense && wide_sys) {
        // device order = (first free pose, last free pose, landmark): Schur items are runs of consecutive landmarks
        // (`order` was generated in landmark order: two stable counting sorts, last key first)
        std::vector<LmInfo> tmp;
        stable_count_sort(order, tmp, (size_t)nfree + 1, [](const LmInfo &a) { return (uint32_t)(a.fhi + 1); });
        stable_count_sort(order, tmp, (size_t)nfree + 1, [&](const LmInfo &a) { return a.flo < 0 ? (uint32_t)nfree : (uint32_t)a.flo; });
    } else if (dense) {
        // the reduced system of the general path is stored as a dense lower triangle + right-hand-side rows; only its
        // structurally non-zero tiles are ever touched (symbolic factorisation below), so its size is bounded by memory,
        // not by time: SSBA_DENSE_MAX_GB (default 160 of the 288 GB of an MI355X; 10 000 free poses = 28.8 GB)
        const double dn_gb = (double)(6.0 * nfree + 2 * DN_BS) * (6.0 * nfree + DN_BS) * 8.0 / 1e9;
        const char *mg = getenv("SSBA_DENSE_MAX_GB");
        const double dn_cap = mg ? atof(mg) : 160.0;
        if (in.world_size > 1 || dn_gb > dn_cap || (6 * (long)nfree + DN_BS) / DN_BS >= 65535) {
            set_error("problem structure (tracks > SSBA_MAX_TRACK or co-visibility span > 12 poses) needs the dense reduced system: "
                      "single GPU only, and its array must fit SSBA_DENSE_MAX_GB (default 160)");
            return SSBA_ERR_UNSUPPORTED;
        }
        // device order = landmark order (`order` was generated that way)
    } else {
        // device order = (first pose, last pose, landmark): two stable counting sorts, last key first (a comparison sort of a
        // million landmarks took 30 ms)
        std::vector<LmInfo> tmp;
        stable_count_sort(order, tmp, (size_t)P, [](const LmInfo &a) { return a.kmax; });
        stable_count_sort(order, tmp, (size_t)P, [](const LmInfo &a) { return a.kmin; });
    }
    const uint32_t Lact = (uint32_t)order.size();
    const uint32_t Lpad = std::max<uint32_t>(256, (Lact + 255) / 256 * 256);

    phase.mark("finalize: 2 pose sets + order");
    // greedy windows: consecutive landmarks whose pose sets fit one list of <= TW poses
    std::vector<uint32_t> win_pose, win_begin;   // win_begin has n_windows+1 entries
    std::vector<uint32_t> lm_win(Lpad, 0);
    {
        std::vector<uint32_t> cur_set, merged;
        uint32_t begin = 0;
        auto close = [&](uint32_t end) {
            const uint32_t w = (uint32_t)win_begin.size();
            win_begin.push_back(begin);
            for (int s = 0; s < TW; ++s) win_pose.push_back(s < (int)cur_set.size() ? cur_set[s] : 0xFFFFFFFFu);
            for (uint32_t l = begin; l < end; ++l) lm_win[l] = w;
            begin = end;
        };
        for (uint32_t l = 0; l < Lact && !dense; ++l) {
            const uint32_t *kb = &lm_pose_sorted[lm_start[order[l].j]], *ke = &lm_pose_sorted[lm_start[order[l].j + 1]];
            if (std::includes(cur_set.begin(), cur_set.end(), kb, ke)) continue;      // the common case: nothing new
            merged.clear();
            std::set_union(cur_set.begin(), cur_set.end(), kb, ke, std::back_inserter(merged));
            if (merged.size() > (size_t)TW) {
                close(l);
                cur_set.assign(kb, ke);
            } else {
                cur_set.swap(merged);
            }
        }
        if (Lact > 0 && !dense) close(Lact);
        win_begin.push_back(dense ? 0 : Lact);
        if (dense) win_begin.assign(1, 0);
    }
    const uint32_t n_windows = (uint32_t)win_begin.size() - 1;

    phase.mark("finalize: 3 windows");
    // ELL observation arrays, masks
    const uint32_t n_groups = Lpad / LMG;
    // (not zeroed here: the threads that fill the slots write the defaults of their groups first, u = v = 0, d = 1)
    const size_t n_ell = (size_t)n_groups * TW * LMG;
    raw_vector<double> ou(n_ell), ov(n_ell), od(n_ell);
    std::vector<uint32_t> lm_mask(Lpad, 0);
    raw_vector<double> oint, onx, ony, onz;
    std::vector<uint32_t> lm_mat;
    if (ph) {
        oint.resize(n_ell); onx.resize(n_ell); ony.resize(n_ell); onz.resize(n_ell);
        lm_mat.assign(Lpad, 0);
    }
    auto ell_defaults = [&](uint32_t g0, uint32_t g1) {      // groups [g0, g1) of 64 landmarks
        const size_t a = (size_t)g0 * TW * LMG, b = (size_t)g1 * TW * LMG;
        std::fill(ou.begin() + a, ou.begin() + b, 0.0); std::fill(ov.begin() + a, ov.begin() + b, 0.0); std::fill(od.begin() + a, od.begin() + b, 1.0);
        if (ph) {
            std::fill(oint.begin() + a, oint.begin() + b, 0.0); std::fill(onx.begin() + a, onx.begin() + b, 0.0);
            std::fill(ony.begin() + a, ony.begin() + b, 0.0); std::fill(onz.begin() + a, onz.begin() + b, 1.0);
        }
    };
    if (dense) ell_defaults(0, n_groups);       // the general layout does not use them (lighting terms: re-assigned below)
    if (dense && ph) {
        const size_t n1 = std::max<size_t>(N, 1);
        ou.assign(n1, 0.0); ov.assign(n1, 0.0); od.assign(n1, 1.0);
        oint.assign(n1, 0.0); onx.assign(n1, 0.0); ony.assign(n1, 0.0); onz.assign(n1, 1.0);
    }
    out.user_of_dev.assign(Lpad, 0xFFFFFFFFu);
    // general path: landmark-major observation arrays + the pose-major index list into them
    std::vector<uint32_t> dn_lm_start, dn_obs_pose, dn_obs_lm, dn_pose_start, dn_pose_obs, dn_zpos;
    std::vector<double> dn_u, dn_v, dn_d, dn_Sobs, dn_prec;
    std::vector<uint32_t> dn_blk_a, dn_blk_b, dn_blk_start, dn_pair_a, dn_pair_b, dn_pose_mat_start, dn_ztile;
    DensePlan dplan;
    WideLayout wlay;
    if (dense) {
        dn_lm_start.assign(Lpad + 1, 0);
        for (uint32_t l = 0; l < Lact; ++l) {
            const uint32_t j = order[l].j;
            out.user_of_dev[l] = j;
            lm_mask[l] = 1u;
            if (ph) lm_mat[l] = in.ph_mat_of_point[j];
            for (uint32_t e = lm_start[j]; e < lm_start[j + 1]; ++e) {
                const uint32_t i = lm_obs[e];
                if (ph) {      // the lighting kernels read the same arrays as in the windowed layout, filled landmark-major
                    const size_t o = dn_obs_pose.size();
                    ou[o] = in.obs_uvd[3 * (size_t)i]; ov[o] = in.obs_uvd[3 * (size_t)i + 1]; od[o] = in.obs_uvd[3 * (size_t)i + 2];
                    oint[o] = in.ph_intensity[i];
                    onx[o] = in.ph_nobs[3 * (size_t)i]; ony[o] = in.ph_nobs[3 * (size_t)i + 1]; onz[o] = in.ph_nobs[3 * (size_t)i + 2];
                }
                dn_obs_pose.push_back(in.obs_pose[i]);
                dn_obs_lm.push_back(l);
                dn_u.push_back(in.obs_uvd[3 * (size_t)i]); dn_v.push_back(in.obs_uvd[3 * (size_t)i + 1]); dn_d.push_back(in.obs_uvd[3 * (size_t)i + 2]);
                if (in.per_obs_S) dn_Sobs.insert(dn_Sobs.end(), &in.obs_S[9 * (size_t)i], &in.obs_S[9 * (size_t)i] + 9);
            }
            dn_lm_start[l + 1] = (uint32_t)dn_obs_pose.size();
        }
        for (uint32_t l = Lact; l < Lpad; ++l) dn_lm_start[l + 1] = dn_lm_start[Lact];
        dn_pose_start.assign(P + 1, 0);
        for (uint32_t q : dn_obs_pose) dn_pose_start[q + 1]++;
        for (uint32_t k = 0; k < P; ++k) dn_pose_start[k + 1] += dn_pose_start[k];
        dn_pose_obs.resize(dn_obs_pose.size());
        std::vector<uint32_t> cur2(dn_pose_start.begin(), dn_pose_start.end() - 1);
        for (uint32_t i = 0; i < dn_obs_pose.size(); ++i) dn_pose_obs[cur2[dn_obs_pose[i]]++] = i;
        if (ph) {       // the border kernel walks a pose's observations material by material
            for (uint32_t k = 0; k < P; ++k) {
                std::stable_sort(dn_pose_obs.begin() + dn_pose_start[k], dn_pose_obs.begin() + dn_pose_start[k + 1],
                                 [&](uint32_t x, uint32_t y) { return lm_mat[dn_obs_lm[x]] < lm_mat[dn_obs_lm[y]]; });
                uint32_t q = dn_pose_start[k];
                for (uint32_t m = 0; m <= in.M; ++m) {
                    while (q < dn_pose_start[k + 1] && lm_mat[dn_obs_lm[dn_pose_obs[q]]] < m) ++q;
                    dn_pose_mat_start.push_back(q);
                }
            }
        }
        // the per-observation factor Z (k_dn_wy) is stored POSE-major: the pairs of a block (a, b) then walk through pose a's
        // and pose b's records in ascending order instead of striding through a landmark-major array
        dn_zpos.resize(dn_obs_pose.size());
        for (uint32_t i = 0; i < dn_pose_obs.size(); ++i) dn_zpos[dn_pose_obs[i]] = i;
        if (!ph && !in.per_obs_S) {      // pose-major copy of the observation records for k_linearize_poses
            dn_prec.resize(4 * dn_pose_obs.size());
            for (size_t i = 0; i < dn_pose_obs.size(); ++i) {
                const uint32_t e = dn_pose_obs[i];
                const int64_t lm = (int64_t)dn_obs_lm[e];
                dn_prec[4 * i] = dn_u[e]; dn_prec[4 * i + 1] = dn_v[e]; dn_prec[4 * i + 2] = dn_d[e];
                memcpy(&dn_prec[4 * i + 3], &lm, 8);
            }
        }
        if (wide_sys) {     // 144-row super-blocks: Schur items, slot table, gather lists (no pair lists, no symbolic Cholesky)
            if (!build_wide_layout(nfree, Lact, Lpad, dn_lm_start.data(), dn_obs_pose.data(), out.pose_free.data(), 128u, wlay)) {
                set_error("internal: a landmark's free poses span more than the wide window");
                return SSBA_ERR_STATE;
            }
        }
    }
    if (dense && !wide_sys) {
        // blocks (a <= b) of S = H_pp - sum_l Y_l W_l^T and, per block, the observation pairs (ea, eb) of one
        // landmark that contribute Y_ea W_eb^T; sorted by block so that one wave owns one block (no atomics)
        struct Pr { uint32_t a, b, ea, eb; };
        std::vector<Pr> prs;
        uint64_t npairs = 0;
        for (uint32_t l = 0; l < Lact; ++l) { const uint64_t n = dn_lm_start[l + 1] - dn_lm_start[l]; npairs += n * (n + 1) / 2; }
        if (npairs > (1ull << 28)) { set_error("general-structure path: too many co-visibility pairs"); return SSBA_ERR_UNSUPPORTED; }
        prs.reserve((size_t)npairs + nfree);
        for (int f = 0; f < nfree; ++f) prs.push_back({(uint32_t)f, (uint32_t)f, 0xFFFFFFFFu, 0xFFFFFFFFu});   // every free pose has its diagonal block
        for (uint32_t l = 0; l < Lact; ++l)
            for (uint32_t ea = dn_lm_start[l]; ea < dn_lm_start[l + 1]; ++ea) {
                const int fa = out.pose_free[dn_obs_pose[ea]];
                if (fa < 0) continue;
                for (uint32_t eb = dn_lm_start[l]; eb < dn_lm_start[l + 1]; ++eb) {
                    const int fb = out.pose_free[dn_obs_pose[eb]];
                    if (fb < fa) continue;       // fb == fa keeps both orders of a repeated (pose, landmark) pair
                    prs.push_back({(uint32_t)fa, (uint32_t)fb, ea, eb});
                }
            }
        for (auto &rf : in.rel_factors) {          // off-diagonal block of a relative-pose block (no observation pairs)
            const int f1 = out.pose_free[rf.pose1], f2 = out.pose_free[rf.pose2];
            if (f1 >= 0 && f2 >= 0) prs.push_back({(uint32_t)std::min(f1, f2), (uint32_t)std::max(f1, f2), 0xFFFFFFFFu, 0xFFFFFFFFu});
        }
        std::sort(prs.begin(), prs.end(), [](const Pr &x, const Pr &y) {
            if (x.a != y.a) return x.a < y.a;
            if (x.b != y.b) return x.b < y.b;
            if (x.ea != y.ea) return x.ea < y.ea;
            return x.eb < y.eb;
        });
        for (size_t i = 0; i < prs.size(); ++i) {
            if (i == 0 || prs[i].a != prs[i - 1].a || prs[i].b != prs[i - 1].b) {
                dn_blk_a.push_back(prs[i].a); dn_blk_b.push_back(prs[i].b);
                dn_blk_start.push_back((uint32_t)dn_pair_a.size());
            }
            if (prs[i].ea != 0xFFFFFFFFu) { dn_pair_a.push_back(dn_zpos[prs[i].ea]); dn_pair_b.push_back(dn_zpos[prs[i].eb]); }
        }
        dn_blk_start.push_back((uint32_t)dn_pair_a.size());
        // symbolic Cholesky at DN_BS-block granularity (natural order): banded problems stay banded, a loop closure
        // fills the rows between its two ends; block row nbk (the right-hand side) is in every column
        const int nbk = (6 * nfree + DN_BS - 1) / DN_BS;
        std::vector<std::vector<uint8_t>> nz(nbk, std::vector<uint8_t>(nbk, 0));
        for (size_t i = 0; i < dn_blk_a.size(); ++i)
            for (int r = 0; r < 6; r += 5)
                for (int c = 0; c < 6; c += 5) nz[(dn_blk_b[i] * 6 + c) / DN_BS][(dn_blk_a[i] * 6 + r) / DN_BS] = 1;
        dplan.row_start.assign(1, 0); dplan.tile_start.assign(1, 0);
        for (int j = 0; j < nbk; ++j) {
            std::vector<uint32_t> R;
            for (int i = j + 1; i < nbk; ++i) if (nz[i][j]) R.push_back((uint32_t)i);
            for (size_t x = 0; x < R.size(); ++x)
                for (size_t y = 0; y <= x; ++y) nz[R[x]][R[y]] = 1;
            R.push_back((uint32_t)nbk);
            for (size_t x = 0; x < R.size(); ++x)
                for (size_t y = 0; y <= x && R[y] < (uint32_t)nbk; ++y) { dplan.ti.push_back(R[x]); dplan.tk.push_back(R[y]); }
            dplan.rows.insert(dplan.rows.end(), R.begin(), R.end());
            dplan.row_start.push_back((uint32_t)dplan.rows.size());
            dplan.tile_start.push_back((uint32_t)dplan.ti.size());
        }
        dplan.col_start.assign(1, 0); dplan.upd_last.assign(nbk + 1, 0);
        for (int i = 0; i <= nbk; ++i) {      // block row i of the factor: its columns j < i - 1 (j = i - 1 is handled by the solving work-group)
            if (i >= 1 && i < nbk) dplan.upd_last[i] = nz[i][i - 1];
            for (int j = 0; i < nbk && j + 1 < i; ++j) if (nz[i][j]) dplan.cols.push_back((uint32_t)j);
            dplan.col_start.push_back((uint32_t)dplan.cols.size());
        }
        dplan.nbk = nbk;
        // every tile the factorisation touches: the diagonal and the non-zero block rows of each block column (the
        // right-hand-side block row nbk among them)
        for (int j = 0; j < nbk; ++j) {
            dn_ztile.push_back(((uint32_t)j << 16) | (uint32_t)j);
            for (uint32_t x = dplan.row_start[j]; x < dplan.row_start[j + 1]; ++x) dn_ztile.push_back((dplan.rows[x] << 16) | (uint32_t)j);
        }
    }
    // ELL slots and the pose-major reference list (landmark*16 + slot) in one pass: a landmark's poses and its window's
    // pose list are both ascending (one merge per landmark), and landmarks are visited in device order, so counting
    // leaves every pose's references ascending
    std::vector<uint32_t> pose_obs_start(P + 1, 0);
    raw_vector<uint32_t> pose_obs_ref(dense ? 0 : N);
    std::vector<uint32_t> pose_mat_start;   // config 3: references of a pose sorted by material, P*(M+1) offsets
    const uint32_t Mm = ph ? in.M : 0;
    if (!dense) {
        for (uint32_t k = 0; k < P; ++k) pose_obs_start[k + 1] = pose_obs_start[k] + pose_cnt[k];
        const uint32_t *lmo = lm_obs_by_pose.empty() ? lm_obs.data() : lm_obs_by_pose.data();
        // landmarks [l0, l1) in device order; at[k] = where the next reference of pose k goes
        auto fill = [&](uint32_t l0, uint32_t l1, uint32_t *at) {
            for (uint32_t l = l0; l < l1; ++l) {
                const uint32_t j = order[l].j, w = lm_win[l];
                out.user_of_dev[l] = j;
                if (ph) lm_mat[l] = in.ph_mat_of_point[j];
                const uint32_t *wp = &win_pose[(size_t)w * TW];
                const size_t base = (size_t)(l / LMG) * (TW * LMG) + (l % LMG);
                int s = 0;
                uint32_t mask = 0;
                for (uint32_t e = lm_start[j]; e < lm_start[j + 1]; ++e) {
                    const uint32_t i = lmo[e], k = lm_pose_sorted[e];
                    while (wp[s] != k) ++s;
                    const size_t oi = base + (size_t)s * LMG;
                    ou[oi] = in.obs_uvd[3 * (size_t)i];
                    ov[oi] = in.obs_uvd[3 * (size_t)i + 1];
                    od[oi] = in.obs_uvd[3 * (size_t)i + 2];
                    if (ph) {
                        oint[oi] = in.ph_intensity[i];
                        onx[oi] = in.ph_nobs[3 * (size_t)i];
                        ony[oi] = in.ph_nobs[3 * (size_t)i + 1];
                        onz[oi] = in.ph_nobs[3 * (size_t)i + 2];
                    }
                    mask |= 1u << s;
                    pose_obs_ref[at[k]++] = l * 16u + (uint32_t)s;
                }
                lm_mask[l] = mask;
            }
        };
        // Large problems: the landmark range is cut into one piece per host thread.  A first pass counts every piece's
        // references per pose, a prefix over the pieces gives each its own write positions -- the lists come out exactly as
        // the single pass leaves them (12 M observations at C4: 121 ms of a 330 ms ssba_finalize on one thread).
        // Pieces end on group boundaries (64 landmarks): a piece first writes the defaults of its groups, then its slots.
        const int nt = host_threads(N, 400000);
        auto piece = [&](int t) { return t >= nt ? Lpad : (uint32_t)((uint64_t)Lact * (uint64_t)t / (uint64_t)nt) / LMG * LMG; };
        if (nt == 1) {
            std::vector<uint32_t> at(pose_obs_start.begin(), pose_obs_start.end() - 1);
            ell_defaults(0, n_groups);
            fill(0, Lact, at.data());
        } else {
            std::vector<std::vector<uint32_t>> at((size_t)nt, std::vector<uint32_t>(P, 0));
            run_pieces(nt, [&](int t) {
                uint32_t *c = at[(size_t)t].data();
                for (uint32_t l = piece(t); l < std::min(piece(t + 1), Lact); ++l) {
                    const uint32_t j = order[l].j;
                    for (uint32_t e = lm_start[j]; e < lm_start[j + 1]; ++e) ++c[lm_pose_sorted[e]];
                }
            });
            for (uint32_t k = 0; k < P; ++k) {
                uint32_t pos = pose_obs_start[k];
                for (int t = 0; t < nt; ++t) { const uint32_t c = at[(size_t)t][k]; at[(size_t)t][k] = pos; pos += c; }
            }
            run_pieces(nt, [&](int t) {
                ell_defaults(piece(t) / LMG, piece(t + 1) / LMG);
                fill(piece(t), std::min(piece(t + 1), Lact), at[(size_t)t].data());
            });
        }
    }
    for (uint32_t k = 0; k < P && ph && !dense; ++k) {
        uint32_t *rb = &pose_obs_ref[pose_obs_start[k]], *re = &pose_obs_ref[pose_obs_start[k + 1]];
        std::stable_sort(rb, re, [&](uint32_t x, uint32_t y) { return lm_mat[x >> 4] < lm_mat[y >> 4]; });
        size_t q = 0;
        for (uint32_t m = 0; m <= Mm; ++m) {
            while (rb + q < re && lm_mat[rb[q] >> 4] < m) ++q;
            pose_mat_start.push_back((uint32_t)(pose_obs_start[k] + q));
        }
    }

    phase.mark("finalize: 4 observation arrays");
    // Schur work items (slabs): windows, split when long
    const uint32_t kItemMax = 128;
    std::vector<uint32_t> slab_win, slab_b, slab_e;
    for (uint32_t w = 0; w < n_windows; ++w) {
        const uint32_t b = win_begin[w], e = win_begin[w + 1], len = e - b;
        const uint32_t parts = (len + kItemMax - 1) / kItemMax;
        for (uint32_t q = 0; q < parts; ++q) {
            slab_win.push_back(w);
            slab_b.push_back(b + (uint32_t)((uint64_t)len * q / parts));
            slab_e.push_back(b + (uint32_t)((uint64_t)len * (q + 1) / parts));
        }
    }
    const uint32_t n_slabs = (uint32_t)slab_win.size();

    phase.mark("finalize: 5 slabs");
    // reduced-system block structure
    struct Contrib { uint32_t a, b, c; };
    std::vector<Contrib> contribs;
    std::vector<std::pair<uint32_t, uint32_t>> prow;   // (free pose, slab*TW+slot)
    std::vector<Contrib> cb_contribs;                  // closure border: (row free pose, border pose, slab pair | bit 31: transposed)
    std::vector<std::pair<uint32_t, uint32_t>> cb_prow;
    uint32_t bandwidth = 0;
    for (uint32_t it = 0; it < n_slabs; ++it) {
        const uint32_t w = slab_win[it];
        uint32_t slot_any = 0;
        bool pair_any[NPAIR] = {false};
        uint32_t seen[8], n_seen = 0;       // neighbouring landmarks mostly share their slot mask: expand each distinct one once
        for (uint32_t l = slab_b[it]; l < slab_e[it]; ++l) {
            const uint32_t m = lm_mask[l];
            bool known = false;
            for (uint32_t q = 0; q < n_seen; ++q) known |= seen[q] == m;
            if (known) continue;
            seen[n_seen < 8 ? n_seen++ : (l & 7u)] = m;
            slot_any |= m;
            int n = 0;
            for (int a = 0; a < TW; ++a)
                for (int b = a; b < TW; ++b, ++n)
                    if (((m >> a) & 1u) && ((m >> b) & 1u)) pair_any[n] = true;
        }
        int n = 0;
        for (int a = 0; a < TW; ++a)
            for (int b = a; b < TW; ++b, ++n) {
                if (!pair_any[n]) continue;
                const uint32_t ka = win_pose[(size_t)w * TW + a], kb = win_pose[(size_t)w * TW + b];
                const int fa = out.pose_free[ka], fb = out.pose_free[kb];
                if (fa < 0 || fb < 0) continue;
                if (fa >= nchain || fb >= nchain) {     // a block of the closure border: row = the smaller free index
                    const bool swap = fa > fb;          // slab block is (slot a rows) x (slot b columns)
                    cb_contribs.push_back({(uint32_t)(swap ? fb : fa), (uint32_t)(swap ? fa : fb), (it * NPAIR + (uint32_t)n) | (swap ? 0x80000000u : 0u)});
                    continue;
                }
                contribs.push_back({(uint32_t)fa, (uint32_t)fb, it * NPAIR + (uint32_t)n});
                bandwidth = std::max<uint32_t>(bandwidth, (uint32_t)(fb - fa));
            }
        for (int s = 0; s < TW; ++s) {
            if (!((slot_any >> s) & 1u)) continue;
            const int f = out.pose_free[win_pose[(size_t)w * TW + s]];
            if (f >= nchain) cb_prow.push_back({(uint32_t)f, it * TW + (uint32_t)s});
            else if (f >= 0) prow.push_back({(uint32_t)f, it * TW + (uint32_t)s});
        }
    }
    if (bandwidth > (uint32_t)SBP) {      // cannot happen: such problems took the dense path above
        set_error("pose co-visibility bandwidth exceeds the block-tridiagonal envelope of this build");
        return SSBA_ERR_UNSUPPORTED;
    }
    std::vector<uint32_t> sblk_a, sblk_b, sblk_start, sblk_contrib;
    // blocks (a, b), b - a <= SBP, in (a, b) order with their contributions ascending -- a counting sort over the
    // (a, b - a) keys (the contributions were generated in ascending order); every free pose gets its diagonal block
    // even without landmark contributions
    {
        const size_t nkeys = (size_t)nchain * (SBP + 1);
        std::vector<uint32_t> kstart(nkeys + 1, 0);
        for (auto &c : contribs) kstart[(size_t)c.a * (SBP + 1) + (c.b - c.a) + 1]++;
        for (size_t q = 0; q < nkeys; ++q) kstart[q + 1] += kstart[q];
        sblk_contrib.resize(contribs.size());
        std::vector<uint32_t> at(kstart.begin(), kstart.end() - 1);
        for (auto &c : contribs) sblk_contrib[at[(size_t)c.a * (SBP + 1) + (c.b - c.a)]++] = c.c;
        for (size_t q = 0; q < nkeys; ++q) {
            const uint32_t a = (uint32_t)(q / (SBP + 1)), off = (uint32_t)(q % (SBP + 1));
            if (kstart[q + 1] == kstart[q] && off != 0) continue;
            sblk_a.push_back(a);
            sblk_b.push_back(a + off);
            sblk_start.push_back(kstart[q]);
        }
        sblk_start.push_back((uint32_t)contribs.size());
    }
    const uint32_t n_sblk = (uint32_t)sblk_a.size();
    std::sort(prow.begin(), prow.end());
    std::vector<uint32_t> prow_start(nfree + 1, 0), prow_contrib;
    for (auto &pr : prow) prow_start[pr.first + 1]++;
    for (int f = 0; f < nfree; ++f) prow_start[f + 1] += prow_start[f];
    for (auto &pr : prow) prow_contrib.push_back(pr.second);
    // closure border: blocks (row free pose, border pose) in order, every border pose with its diagonal block; then one
    // pseudo-block per border pose for its right-hand-side contributions (b = 0xFFFFFFFF)
    std::vector<uint32_t> cb_a, cb_b, cb_start, cb_contrib;
    if (nborder) {
        for (int f = nchain; f < nfree; ++f) cb_contribs.push_back({(uint32_t)f, (uint32_t)f, 0xFFFFFFFFu});   // marker: diagonal block exists
        std::stable_sort(cb_contribs.begin(), cb_contribs.end(), [](const Contrib &x, const Contrib &y) {
            if (x.a != y.a) return x.a < y.a;
            if (x.b != y.b) return x.b < y.b;
            return (x.c == 0xFFFFFFFFu) > (y.c == 0xFFFFFFFFu);
        });
        for (size_t i = 0; i < cb_contribs.size(); ++i) {
            if (i == 0 || cb_contribs[i].a != cb_contribs[i - 1].a || cb_contribs[i].b != cb_contribs[i - 1].b) {
                cb_a.push_back(cb_contribs[i].a); cb_b.push_back(cb_contribs[i].b);
                cb_start.push_back((uint32_t)cb_contrib.size());
            }
            if (cb_contribs[i].c != 0xFFFFFFFFu) cb_contrib.push_back(cb_contribs[i].c);
        }
        std::sort(cb_prow.begin(), cb_prow.end());
        for (int f = nchain; f < nfree; ++f) {
            cb_a.push_back((uint32_t)f); cb_b.push_back(0xFFFFFFFFu);
            cb_start.push_back((uint32_t)cb_contrib.size());
            for (auto &pr : cb_prow) if ((int)pr.first == f) cb_contrib.push_back(pr.second);
        }
        cb_start.push_back((uint32_t)cb_contrib.size());
    }

    phase.mark("finalize: 6 block structure");
    out.pfs = std::move(pfs);
    out.nfree = nfree; out.nchain = nchain; out.nborder = nborder;
    out.dense = dense; out.wide_sys = wide_sys;
    out.Lact = Lact; out.Lpad = Lpad; out.n_groups = n_groups; out.n_windows = n_windows; out.n_slabs = n_slabs; out.n_sblk = n_sblk;
    out.bandwidth = bandwidth;
#define MV(x) out.x = std::move(x)
    MV(win_pose); MV(lm_win); MV(lm_mask); MV(lm_mat); MV(pose_obs_start); MV(pose_obs_ref); MV(pose_mat_start);
    MV(ou); MV(ov); MV(od); MV(oint); MV(onx); MV(ony); MV(onz);
    MV(slab_win); MV(slab_b); MV(slab_e); MV(sblk_a); MV(sblk_b); MV(sblk_start); MV(sblk_contrib); MV(prow_start); MV(prow_contrib);
    MV(cb_a); MV(cb_b); MV(cb_start); MV(cb_contrib);
    MV(dn_lm_start); MV(dn_obs_pose); MV(dn_obs_lm); MV(dn_pose_start); MV(dn_pose_obs); MV(dn_zpos);
    MV(dn_u); MV(dn_v); MV(dn_d); MV(dn_Sobs); MV(dn_prec);
    MV(dn_blk_a); MV(dn_blk_b); MV(dn_blk_start); MV(dn_pair_a); MV(dn_pair_b); MV(dn_pose_mat_start); MV(dn_ztile);
    MV(dplan); MV(wlay);
#undef MV
    return SSBA_OK;
}

}  // namespace ssba
