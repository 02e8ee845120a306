// Host-side layout of the wide (144-row super-block) reduced system: Schur items, slot table and gather lists.
// Plain C++ (no HIP): compiled into libssba.so and, on its own, by the sanitizer target of tests/ (CPU only).
//
// Reference: Ceres builds its block structure from whatever graph the driver hands it (tests/dataset_vo.cpp:41-56 adds
// one residual block per observation of the file); a landmark seen from 13 .. 24 consecutive states is ordinary there.
#pragma once
#include <stdint.h>

#include <vector>

namespace ssba {

struct WideLayout {
    int n = 0;                          // super-blocks of WSP poses
    uint32_t n_items = 0;
    std::vector<uint32_t> item_begin, item_end, item_base;      // landmarks [begin, end) in device order; first free pose of the window
    std::vector<uint32_t> slot_obs;                             // Lpad x WSP
    std::vector<uint32_t> blk_a, blk_b, blk_start, blk_contrib; // non-zero 6x6 blocks (a <= b) and what they sum
    std::vector<uint32_t> prow_start, prow_contrib;             // per free pose: (item * WSP + slot)
    uint32_t bandwidth = 0;                                     // max free-pose distance of co-observers
};

// lm_start: Lpad + 1 offsets of the landmark-major observation list (device order), obs_pose: pose of each observation,
// pose_free: pose -> free index or -1.  Landmarks must be ordered by (first free pose, last free pose); landmarks without
// a free pose come last.  Returns false when a landmark's free poses span more than WSP - 1 (not banded enough).
bool build_wide_layout(int nfree, uint32_t Lact, uint32_t Lpad, const uint32_t *lm_start, const uint32_t *obs_pose, const int *pose_free,
                       uint32_t max_item_landmarks, WideLayout &out);

}  // namespace ssba
