// Host side of ssba_finalize: everything between the caller's residual blocks and the arrays that go to the device -- landmark
// order, windows and slots of the windowed layout, the closure border, the general (landmark-major) layout with its pair lists and
// the symbolic factorisation of the blocked Cholesky, the wide layout (ssba_wide_layout.h), Schur work items and the gather lists of
// the reduced system.  Plain C++ (no HIP): the library links it, and tests/host/layout_check.cpp compiles it with
// -fsanitize=address,undefined and -fsanitize=thread and feeds it C1 / C2-shaped / long-track / loop-closure problems
// (tests/test_host_layout.py).  Replaces what problem construction does inside Ceres (Problem::AddResidualBlock ... the
// ordering and symbolic phases of Solve; tests/dataset_vo.cpp:41-66 is the reference's call site).
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ssba_wide_layout.h"

namespace ssba {

// std::vector that does not zero its storage (the ELL observation arrays are hundreds of MB at C4: every element is written
// by the threads that fill them, a single-threaded zero-fill first cost as much as the fill)
template <class T> struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using raw_vector = std::vector<T, default_init_allocator<T>>;

// unary pose residual blocks (pose prior, sun sensor; types 2 / 3: the two halves of a relative-pose block)
struct PoseFactor { uint32_t pose; int type; double data[18], S[36], huber; };
struct RelFactor { uint32_t pose1, pose2; double T_ref[12], S[36], huber; };

// Launch plan of the blocked Cholesky of the general-structure path (host copy of the offsets into the index arrays
// Dev::dn_rows / dn_ti / dn_tk / dn_cols; built by the symbolic phase below)
struct DensePlan {
    int nbk = 0;
    std::vector<uint32_t> rows, ti, tk, cols;                    // uploaded
    std::vector<uint32_t> row_start, tile_start, col_start;      // per block column / block row
    std::vector<uint8_t> upd_last;                               // block (i, i-1) non-zero
};

// what the host phase reads of a handle (references: nothing is copied)
struct LayoutInput {
    uint32_t P, L;
    const std::vector<uint32_t> &obs_pose, &obs_point;
    const std::vector<double> &obs_uvd;
    const std::vector<uint8_t> &pose_const;
    bool per_obs_S;
    const std::vector<double> &obs_S;
    bool lighting;
    uint32_t M;
    const std::vector<uint32_t> &ph_mat_of_point;
    const std::vector<double> &ph_intensity, &ph_nobs;
    bool points_const;
    const std::vector<PoseFactor> &pose_factors;
    const std::vector<RelFactor> &rel_factors;
    int world_size;
    bool partitioned;                   // ssba_set_partition with more than one rank
    bool no_closure_border, no_wide;
};

struct Layout {
    std::vector<int> pose_free, free_pose;      // pose -> free index or -1; free index -> pose
    std::vector<uint32_t> user_of_dev;          // device landmark -> caller's landmark or 0xFFFFFFFF
    std::vector<PoseFactor> pfs;                // unary blocks incl. the halves of the relative-pose blocks
    int nfree = 0, nchain = 0, nborder = 0;
    bool dense = false, wide_sys = false;
    uint32_t Lact = 0, Lpad = 0, n_groups = 0, n_windows = 0, n_slabs = 0, n_sblk = 0, bandwidth = 0;
    // windowed layout
    std::vector<uint32_t> win_pose, lm_win, lm_mask, lm_mat, pose_obs_start, pose_mat_start;
    raw_vector<uint32_t> pose_obs_ref;
    raw_vector<double> ou, ov, od, oint, onx, ony, onz;
    std::vector<uint32_t> slab_win, slab_b, slab_e, sblk_a, sblk_b, sblk_start, sblk_contrib, prow_start, prow_contrib;
    std::vector<uint32_t> cb_a, cb_b, cb_start, cb_contrib;      // closure border
    // general layout
    std::vector<uint32_t> dn_lm_start, dn_obs_pose, dn_obs_lm, dn_pose_start, dn_pose_obs, dn_zpos;
    std::vector<double> dn_u, dn_v, dn_d, dn_Sobs, dn_prec;
    std::vector<uint32_t> dn_blk_a, dn_blk_b, dn_blk_start, dn_pair_a, dn_pair_b, dn_pose_mat_start, dn_ztile;
    DensePlan dplan;
    WideLayout wlay;
};

// Returns an ssba_status code (include/ssba.h); on failure `err` says why and `out` is to be discarded.  `mark` (optional) is
// called at the end of every phase with its name (SSBA_API_TIMING).
int build_layout(const LayoutInput &in, Layout &out, std::string &err, const std::function<void(const char *)> &mark = nullptr);

}  // namespace ssba
