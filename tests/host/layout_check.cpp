// CPU-only check of the host phase of ssba_finalize (ceres_slam_amd/csrc/ssba_layout.cpp + ssba_wide_layout.cpp), built by
// tests/test_host_layout.py with -fsanitize=address,undefined and once with -fsanitize=thread (the observation arrays of large
// problems are filled by several host threads).  Problems: a C1-shaped one (50 poses / 2 000 landmarks), a C2-shaped one
// (1 000 poses / 40 000 landmarks / 450 000 observations: above the 400 000 from which several host threads fill the arrays), long tracks (the wide layout), tracks beyond the
// wide window (blocked Cholesky: pair lists + symbolic factorisation), a loop closure (closure border), unsorted and repeated
// observations, lighting terms, per-block stiffness, constant poses.  Invariants: every observation sits in exactly one place
// of the layout that was chosen, with its own (u, v, d); windows hold at most TW poses; the pose-major lists hold every
// observation of a pose once; every co-visible pair of free poses has its block.  Exit code 0 = all invariants hold.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../../include/ssba.h"
#include "ssba_layout.h"
#include "ssba_types.h"

using namespace ssba;

struct Prob {
    uint32_t P = 0, L = 0;
    std::vector<uint32_t> obs_pose, obs_point;
    std::vector<double> obs_uvd, obs_S, ph_intensity, ph_nobs;
    std::vector<uint8_t> pose_const;
    std::vector<uint32_t> ph_mat;
    std::vector<PoseFactor> pfs;
    std::vector<RelFactor> rfs;
    bool per_obs_S = false, lighting = false;
    uint32_t M = 0;
};

// landmark j is seen by `track` consecutive poses starting where j sits along the trajectory (synth.py's shape)
static Prob make(uint32_t P, uint32_t L, uint32_t track, int seed, bool shuffle = false, int n_const = 1) {
    Prob q;
    q.P = P; q.L = L;
    q.pose_const.assign(P, 0);
    for (int k = 0; k < n_const && k < (int)P; ++k) q.pose_const[k] = 1;
    std::mt19937 rng(seed);
    for (uint32_t j = 0; j < L; ++j) {
        const uint32_t t = std::min(track, P);
        const uint32_t first = (uint32_t)((uint64_t)j * (P - t + 1) / L);
        for (uint32_t k = first; k < first + t; ++k) {
            if (t > 3 && rng() % 17 == 0) continue;       // a dropped frame
            q.obs_pose.push_back(k); q.obs_point.push_back(j);
            q.obs_uvd.push_back((double)q.obs_pose.size()); q.obs_uvd.push_back(0.5 * (double)k); q.obs_uvd.push_back(1.0 + (double)j);
        }
    }
    if (shuffle) {      // the caller's order is arbitrary (Ceres takes residual blocks in any order)
        std::vector<uint32_t> perm(q.obs_pose.size());
        for (uint32_t i = 0; i < perm.size(); ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        Prob r = q;
        for (uint32_t i = 0; i < perm.size(); ++i) {
            r.obs_pose[i] = q.obs_pose[perm[i]]; r.obs_point[i] = q.obs_point[perm[i]];
            for (int c = 0; c < 3; ++c) r.obs_uvd[3 * (size_t)i + c] = q.obs_uvd[3 * (size_t)perm[i] + c];
        }
        q = r;
    }
    return q;
}

static int fail(const char *name, const std::string &what) { printf("FAIL (%s): %s\n", name, what.c_str()); return 1; }

// expect: 0 windowed, 1 wide, 2 blocked Cholesky, 3 closure border, -1 must be rejected
static int check(const char *name, const Prob &q, int expect, int world = 1) {
    const std::vector<double> none;
    LayoutInput in{q.P, q.L, q.obs_pose, q.obs_point, q.obs_uvd, q.pose_const, q.per_obs_S, q.obs_S, q.lighting, q.M, q.ph_mat,
                   q.ph_intensity, q.ph_nobs, false, q.pfs, q.rfs, world, false, false, false};
    Layout lay;
    std::string err;
    int phases = 0;
    const int rc = build_layout(in, lay, err, [&](const char *) { ++phases; });
    if (expect < 0) return rc != SSBA_OK && !err.empty() ? 0 : fail(name, "a problem that must be rejected was accepted");
    if (rc != SSBA_OK) return fail(name, "rejected: " + err);
    if (phases != 6) return fail(name, "phase marks");
    const uint64_t N = q.obs_pose.size();
    const int got = lay.dense ? (lay.wide_sys ? 1 : 2) : (lay.nborder ? 3 : 0);
    if (got != expect) return fail(name, "layout kind " + std::to_string(got) + " instead of " + std::to_string(expect));
    // free poses
    if ((int)lay.free_pose.size() != lay.nfree || lay.nchain + lay.nborder != lay.nfree) return fail(name, "free pose counts");
    for (int f = 0; f < lay.nfree; ++f) if (lay.pose_free[lay.free_pose[f]] != f) return fail(name, "pose_free / free_pose are not inverse");
    for (uint32_t k = 0; k < q.P; ++k) if (q.pose_const[k] && lay.pose_free[k] >= 0) return fail(name, "a constant pose is free");
    // landmarks: a permutation of the observed ones
    std::vector<uint8_t> seen_lm(q.L, 0);
    uint32_t nact = 0;
    for (uint32_t l = 0; l < lay.Lpad; ++l) {
        const uint32_t j = lay.user_of_dev[l];
        if (j == 0xFFFFFFFFu) continue;
        if (j >= q.L || seen_lm[j]) return fail(name, "user_of_dev is not injective");
        seen_lm[j] = 1; ++nact;
    }
    if (nact != lay.Lact || lay.Lpad % 256) return fail(name, "active landmark count / padding");
    std::vector<uint32_t> dev_of(q.L, 0xFFFFFFFFu);
    for (uint32_t l = 0; l < lay.Lpad; ++l) if (lay.user_of_dev[l] != 0xFFFFFFFFu) dev_of[lay.user_of_dev[l]] = l;
    std::multiset<std::pair<uint32_t, uint32_t>> want;        // (device landmark, pose) of every observation
    for (uint64_t i = 0; i < N; ++i) want.insert({dev_of[q.obs_point[i]], q.obs_pose[i]});
    std::map<std::pair<uint32_t, uint32_t>, double> uval;     // u of an observation carries its 1-based index
    for (uint64_t i = 0; i < N; ++i) uval[{dev_of[q.obs_point[i]], q.obs_pose[i]}] += q.obs_uvd[3 * i];
    std::multiset<std::pair<uint32_t, uint32_t>> have;
    std::map<std::pair<uint32_t, uint32_t>, double> usum;
    if (!lay.dense) {
        if (lay.win_pose.size() != (size_t)lay.n_windows * TW) return fail(name, "window table size");
        for (uint32_t w = 0; w < lay.n_windows; ++w)
            for (int s = 1; s < TW; ++s) {
                const uint32_t a = lay.win_pose[(size_t)w * TW + s - 1], b = lay.win_pose[(size_t)w * TW + s];
                if (b != 0xFFFFFFFFu && !(a < b)) return fail(name, "window poses not ascending");
            }
        for (uint32_t l = 0; l < lay.Lact; ++l) {
            const uint32_t w = lay.lm_win[l];
            if (w >= lay.n_windows) return fail(name, "landmark without a window");
            for (int s = 0; s < TW; ++s) {
                if (!((lay.lm_mask[l] >> s) & 1u)) continue;
                const uint32_t k = lay.win_pose[(size_t)w * TW + s];
                if (k == 0xFFFFFFFFu) return fail(name, "mask bit on an empty slot");
                have.insert({l, k});
                usum[{l, k}] += lay.ou[(size_t)(l / LMG) * (TW * LMG) + (size_t)s * LMG + (l % LMG)];
            }
        }
        // pose-major references
        if (lay.pose_obs_start.size() != (size_t)q.P + 1 || lay.pose_obs_start[q.P] != N) return fail(name, "pose list offsets");
        std::multiset<std::pair<uint32_t, uint32_t>> refs;
        for (uint32_t k = 0; k < q.P; ++k)
            for (uint32_t i = lay.pose_obs_start[k]; i < lay.pose_obs_start[k + 1]; ++i) {
                const uint32_t l = lay.pose_obs_ref[i] >> 4, s = lay.pose_obs_ref[i] & 15u;
                if (l >= lay.Lact || lay.win_pose[(size_t)lay.lm_win[l] * TW + s] != k) return fail(name, "a pose reference points at another pose's slot");
                if (i > lay.pose_obs_start[k]) {       // ascending; with lighting terms by material first (the border kernel walks a pose's list material by material)
                    const uint32_t lp = lay.pose_obs_ref[i - 1] >> 4;
                    const uint32_t ma = q.lighting ? lay.lm_mat[lp] : 0, mb = q.lighting ? lay.lm_mat[l] : 0;
                    if (ma > mb || (ma == mb && lay.pose_obs_ref[i] <= lay.pose_obs_ref[i - 1])) return fail(name, "pose references not in order");
                }
                refs.insert({l, k});
            }
        if (refs != want) return fail(name, "pose-major references do not cover the observations");
        // Schur items cover the landmarks of their window
        std::vector<uint8_t> covered(lay.Lact, 0);
        for (uint32_t it = 0; it < lay.n_slabs; ++it)
            for (uint32_t l = lay.slab_b[it]; l < lay.slab_e[it]; ++l) {
                if (lay.lm_win[l] != lay.slab_win[it] || covered[l]) return fail(name, "Schur items");
                covered[l] = 1;
            }
        for (uint32_t l = 0; l < lay.Lact; ++l) if (!covered[l]) return fail(name, "a landmark without a Schur item");
        // every co-visible pair of chain poses has its block of the reduced system, inside the envelope
        std::set<std::pair<uint32_t, uint32_t>> blocks;
        for (uint32_t b = 0; b < lay.n_sblk; ++b) {
            if (lay.sblk_b[b] < lay.sblk_a[b] || lay.sblk_b[b] - lay.sblk_a[b] > (uint32_t)SBP) return fail(name, "block outside the envelope");
            blocks.insert({lay.sblk_a[b], lay.sblk_b[b]});
        }
        std::set<std::pair<uint32_t, uint32_t>> border_blocks;
        for (size_t b = 0; b < lay.cb_a.size(); ++b) border_blocks.insert({lay.cb_a[b], lay.cb_b[b]});
        std::map<uint32_t, std::vector<int>> frees;       // device landmark -> free indices
        for (auto &o : want) { const int f = lay.pose_free[o.second]; if (f >= 0) frees[o.first].push_back(f); }
        for (auto &kv : frees)
            for (int fa : kv.second)
                for (int fb : kv.second) {
                    if (fb < fa) continue;
                    const bool border = fa >= lay.nchain || fb >= lay.nchain;
                    if (!border && !blocks.count({(uint32_t)fa, (uint32_t)fb})) return fail(name, "a co-visible pose pair has no block");
                    if (border && !border_blocks.count({(uint32_t)fa, (uint32_t)fb})) return fail(name, "a pair with a border pose has no border block");
                }
    } else {
        if (lay.dn_lm_start.size() != (size_t)lay.Lpad + 1 || lay.dn_lm_start[lay.Lpad] != N) return fail(name, "landmark-major offsets");
        for (uint32_t l = 0; l < lay.Lact; ++l)
            for (uint32_t e = lay.dn_lm_start[l]; e < lay.dn_lm_start[l + 1]; ++e) {
                if (lay.dn_obs_lm[e] != l) return fail(name, "dn_obs_lm");
                have.insert({l, lay.dn_obs_pose[e]});
                usum[{l, lay.dn_obs_pose[e]}] += lay.dn_u[e];
            }
        std::vector<uint8_t> hit(N, 0);
        for (uint32_t k = 0; k < q.P; ++k)
            for (uint32_t i = lay.dn_pose_start[k]; i < lay.dn_pose_start[k + 1]; ++i) {
                const uint32_t e = lay.dn_pose_obs[i];
                if (e >= N || hit[e] || lay.dn_obs_pose[e] != k || lay.dn_zpos[e] != i) return fail(name, "pose-major index list");
                hit[e] = 1;
                if (!lay.dn_prec.empty()) {
                    int64_t lm;
                    memcpy(&lm, &lay.dn_prec[4 * (size_t)i + 3], 8);
                    if (lay.dn_prec[4 * (size_t)i] != lay.dn_u[e] || lm != (int64_t)lay.dn_obs_lm[e]) return fail(name, "pose-major records");
                }
            }
        if (std::count(hit.begin(), hit.end(), 1) != (long)N) return fail(name, "pose-major index list incomplete");
        if (lay.wide_sys) {
            if (lay.wlay.n != std::max(1, (lay.nfree + WSP - 1) / WSP) || lay.wlay.n_items == 0) return fail(name, "wide layout sizes");
        } else {
            std::set<std::pair<uint32_t, uint32_t>> blocks;
            for (size_t b = 0; b < lay.dn_blk_a.size(); ++b) blocks.insert({lay.dn_blk_a[b], lay.dn_blk_b[b]});
            for (uint32_t l = 0; l < lay.Lact; ++l)
                for (uint32_t ea = lay.dn_lm_start[l]; ea < lay.dn_lm_start[l + 1]; ++ea)
                    for (uint32_t eb = lay.dn_lm_start[l]; eb < lay.dn_lm_start[l + 1]; ++eb) {
                        const int fa = lay.pose_free[lay.dn_obs_pose[ea]], fb = lay.pose_free[lay.dn_obs_pose[eb]];
                        if (fa >= 0 && fb >= fa && !blocks.count({(uint32_t)fa, (uint32_t)fb})) return fail(name, "a co-visible pose pair has no block");
                    }
            // symbolic factorisation: every block column lists ascending rows and ends with the right-hand-side row
            const DensePlan &dp = lay.dplan;
            if (dp.nbk != (6 * lay.nfree + DN_BS - 1) / DN_BS || (int)dp.row_start.size() != dp.nbk + 1) return fail(name, "symbolic factorisation sizes");
            for (int j = 0; j < dp.nbk; ++j) {
                if (dp.row_start[j + 1] <= dp.row_start[j] || dp.rows[dp.row_start[j + 1] - 1] != (uint32_t)dp.nbk) return fail(name, "right-hand-side row missing from a block column");
                for (uint32_t x = dp.row_start[j] + 1; x < dp.row_start[j + 1]; ++x) if (dp.rows[x] <= dp.rows[x - 1] || dp.rows[x - 1] <= (uint32_t)j) return fail(name, "block rows of a column");
            }
        }
    }
    if (have != want) return fail(name, "the layout does not hold exactly the caller's observations");
    for (auto &kv : uval) if (usum[kv.first] != kv.second) return fail(name, "an observation's value sits in another one's place");
    printf("ok   %-34s kind %d  free %d (chain %d + border %d)  landmarks %u  windows %u  items %u  blocks %u\n", name, got, lay.nfree, lay.nchain, lay.nborder,
           lay.Lact, lay.n_windows, lay.n_slabs, lay.dense ? (uint32_t)(lay.wide_sys ? lay.wlay.blk_a.size() : lay.dn_blk_a.size()) : lay.n_sblk);
    return 0;
}

int main() {
    int bad = 0;
    bad += check("C1 shape (50 / 2 000 / T 12)", make(50, 2000, 12, 1), 0);
    bad += check("C1 shape, caller's order shuffled", make(50, 2000, 12, 2, true), 0);
    bad += check("short tracks, several constant poses", make(200, 5000, 4, 3, false, 7), 0);
    bad += check("C2 shape at 0.4 scale (threaded fill)", make(1000, 40000, 12, 4), 0);
    bad += check("the same, shuffled (threaded fill)", make(1000, 40000, 12, 5, true), 0);
    bad += check("long tracks T 24 (wide)", make(300, 9000, 24, 6), 1);
    bad += check("long tracks T 13, shuffled (wide)", make(80, 2000, 13, 7, true), 1);
    bad += check("tracks T 40 (blocked Cholesky)", make(120, 1500, 40, 8), 2);
    bad += check("long tracks on two ranks (wide)", make(300, 9000, 20, 9), 1, 2);
    bad += check("tracks T 40 on two ranks", make(120, 1500, 40, 10), -1, 2);
    {   // loop closure: the last three states see 60 landmarks of the first three
        Prob q = make(400, 12000, 8, 11);
        for (uint32_t j = 0; j < 60; ++j)
            for (uint32_t k = q.P - 3; k < q.P; ++k) {
                q.obs_pose.push_back(k); q.obs_point.push_back(j);
                q.obs_uvd.push_back((double)q.obs_pose.size()); q.obs_uvd.push_back(0.0); q.obs_uvd.push_back(1.0);
            }
        bad += check("loop closure (closure border)", q, 3);
        bad += check("loop closure on two ranks", q, -1, 2);
    }
    {   // too many closing states for a border: general layout
        Prob q = make(100, 2000, 6, 12);
        for (uint32_t j = 0; j < 40; ++j)
            for (uint32_t k = q.P - 9; k < q.P; ++k) {
                q.obs_pose.push_back(k); q.obs_point.push_back(j);
                q.obs_uvd.push_back((double)q.obs_pose.size()); q.obs_uvd.push_back(0.0); q.obs_uvd.push_back(1.0);
            }
        bad += check("nine closing states (blocked Cholesky)", q, 2);
    }
    {   // two residual blocks on one (pose, landmark) pair
        Prob q = make(30, 600, 6, 13);
        q.obs_pose.push_back(q.obs_pose[0]); q.obs_point.push_back(q.obs_point[0]);
        q.obs_uvd.push_back((double)q.obs_pose.size()); q.obs_uvd.push_back(0.0); q.obs_uvd.push_back(1.0);
        bad += check("repeated observation (blocked Cholesky)", q, 2);
    }
    {   // per-block stiffness
        Prob q = make(30, 600, 6, 14);
        q.per_obs_S = true;
        q.obs_S.assign(9 * q.obs_pose.size(), 1.0);
        bad += check("per-block stiffness (blocked Cholesky)", q, 2);
    }
    {   // lighting terms, windowed and with long tracks
        for (int t : {8, 20}) {
            Prob q = make(60, 1500, (uint32_t)t, 15 + t);
            q.lighting = true; q.M = 3;
            q.ph_mat.resize(q.L);
            for (uint32_t j = 0; j < q.L; ++j) q.ph_mat[j] = j % 3;
            q.ph_intensity.assign(q.obs_pose.size(), 0.5);
            q.ph_nobs.assign(3 * q.obs_pose.size(), 0.0);
            bad += check(t == 8 ? "lighting terms (windowed)" : "lighting terms, T 20 (blocked Cholesky)", q, t == 8 ? 0 : 2);
        }
    }
    {   // a pose graph without landmarks: relative-pose blocks only
        Prob q;
        q.P = 6; q.L = 0;
        q.pose_const.assign(6, 0); q.pose_const[0] = 1;
        for (uint32_t k = 0; k + 1 < 6; ++k) { RelFactor rf{}; rf.pose1 = k; rf.pose2 = k + 1; q.rfs.push_back(rf); }
        bad += check("pose graph (relative-pose blocks only)", q, 2);
    }
    if (bad) { printf("%d problem(s) failed\n", bad); return 1; }
    printf("all invariants hold\n");
    return 0;
}
