// Host-side launch interface between ssba_api.hip and ssba_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "ssba_layout.h"     // DensePlan
#include "ssba_types.h"

namespace ssba {

enum KernelClass {
    KC_LIN_LM = 0,
    KC_LIN_POSE,
    KC_SCHUR,
    KC_ASSEMBLE,
    KC_BCR_FACTOR,
    KC_BCR_REDUCE,
    KC_BCR_BACKSUB,
    KC_BACKSUB_EVAL,
    KC_DOGLEG,
    KC_BORDER,
    KC_COPY,
    KC_SMALL,
    KC_COUNT
};

static const char *const kKernelClassName[KC_COUNT] = {
    "k_linearize_landmarks", "k_linearize_poses", "k_schur_windows", "k_assemble_reduced",
    "k_bcr_factor", "k_bcr_reduce", "k_bcr_backsub", "k_backsub_eval", "k_dogleg_gn+k_dogleg_eval", "border(shared blocks)", "copy(k_best,k_commit)",
    "small(control,reductions)"};


// Stream + optional per-kernel-class HIP-event timing (events are recorded on the
// same stream the kernels run on and resolved lazily by collect()).
struct Launcher {
    hipStream_t stream = nullptr;
    int timing = 0;
    DensePlan dense;
    bool spb_in_place_ok = false;    // set by the iteration's front part on one GPU (no exchange sums Spb between the Schur launches and the reduced solve)
    bool spb_rides = false;      // launch_ph_schur -> launch_bcr: k_ph_spb_assemble has also written the border columns where they ride (no copy)
    WideSys wide{};              // host copy of the wide system's sizes and pointers (Dev::wide is the device copy), n = 0: none
    struct Pending { int cls; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[KC_COUNT] = {0};
    unsigned long long launches[KC_COUNT] = {0};

    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
    void begin(int cls) {
        if (!timing) return;
        Pending p{cls, get_event(), get_event()};
        hipEventRecord(p.a, stream);
        pending.push_back(p);
    }
    void end(int cls) {
        (void)cls;
        if (!timing) return;
        hipEventRecord(pending.back().b, stream);
    }
    void collect() {   // caller has synchronised the stream
        for (auto &p : pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
                total_ms[p.cls] += ms;
                launches[p.cls] += 1;
            }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    void destroy() {
        for (auto &p : pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
        pending.clear();
        for (auto e : pool) hipEventDestroy(e);
        pool.clear();
    }
};

#define LAUNCH(cls, kern, grid, block, shmem, ...)                               \
    do {                                                                         \
        const dim3 _g = (grid);                                                  \
        if (_g.x > 0 && _g.y > 0) {                                              \
            L.begin(cls);                                                        \
            hipLaunchKernelGGL(kern, _g, block, shmem, L.stream, __VA_ARGS__);   \
            L.end(cls);                                                          \
        }                                                                        \
    } while (0)

int upload_pair_table(hipStream_t s);
int upload_bcr_tables(hipStream_t s);
struct ZeroRange { void *ptr; uint64_t bytes; };
void launch_zero_ranges(hipStream_t stream, const ZeroRange *ranges, int n);       // one work-group per range (<= 1 MiB each)
int configure_kernels();
int configure_schur();
int configure_dense();
// matrix-core block factor / reduce of the reduced-camera solve (ssba_bcr_mfma.hip)
int configure_bcr_mf();
void launch_bcr_factor_mf(Launcher &L, const Dev &d, int nblocks, int lev, int top, int which, bool coupled, bool ride = false, int solve = 0);
void launch_bcr_reduce_mf(Launcher &L, const Dev &d, int nblocks, int ny, int lev, int which, bool ride = false);
void launch_pcr_fused_step(Launcher &L, const Dev &d, int n, int q, int which = 2);        // one launch per step of the fused plan (PcrFused); which = 3: the separator system
void launch_pcr_fused_top(Launcher &L, const Dev &d, int n, int steps, int solve, int which = 2);
void launch_reset(Launcher &L, const Dev &d, const Options &o);
bool launch_can_fuse_all(const Dev &d);
void launch_linearize(Launcher &L, const Dev &d, bool fuse_ctrl = false, bool fuse_all = false, bool skip_reduce = false);    // fuse_ctrl / fuse_best / fuse_all: see ssba_kernels.hip (k_check, launch_linearize)
void launch_schur(Launcher &L, const Dev &d, bool fuse_ctrl = false, bool check_in_schur = false);
bool launch_best_fusable(const Dev &d);
void launch_finish_check(Launcher &L, const Dev &d, bool fuse_ctrl = false, bool fuse_best = false, bool check_in_schur = false, bool best_in_commit = false);     // best_in_commit: launch_decide_commit(.., with_best) of the same iteration does k_best's copy
bool bcr_rhs_rides_in_bwd(const Dev &d);      // the decoupled last step's right-hand side is solved by k_bcrm_bwd as one more column (no k_bcr_backsub launch)
bool bcr_border_rides(const Dev &d);      // the border columns go through the forward part of the solve inside the factor / reduce launches
// fuse_update: the last step of the plan also updates the poses (bcr_updates_poses(d) must hold)
bool bcr_updates_poses(const Dev &d);
void launch_bcr(Launcher &L, const Dev &d, bool allow_pcr = true, bool fuse_update = false);   // allow_pcr = false keeps the factors of every level (multi-rhs sweeps)
// partitioned (multi-rank) solve: pack the chain ends into the separator exchange vector; after the exchange:
// damping + convergence checks, separator BCR, scatter, back-substitution of the chain interior
void launch_finish_local(Launcher &L, const Dev &d);
void launch_sep_pack(Launcher &L, const Dev &d);
void launch_sep_finish_check(Launcher &L, const Dev &d, bool fuse_best = false);
void launch_bcr_separators(Launcher &L, const Dev &d);
void launch_sep_scatter(Launcher &L, const Dev &d);
void launch_mask_unowned_poses(Launcher &L, const Dev &d, double *poses);
int eval_parts(const Dev &d);      // partial sums the evaluation kernel of this layout leaves
void launch_update_eval(Launcher &L, const Dev &d, bool fuse_reduce = false, bool fuse_best = false, bool pose_update_done = false);
void launch_dogleg_eval(Launcher &L, const Dev &d, int stage = 0, int own_poses = 1, bool reduce_later = false);     // reduce_later: launch_ph_ls_fast forms the evaluation sums;      // stage: see ssba_kernels.hip (landmark sharding: one more exchange point)
void launch_decide_commit(Launcher &L, const Dev &d, bool fuse_reduce = false, bool fuse_all = false, int n_pose_parts = -1, bool with_best = false);
// config 3 (ssba_phong_solver.hip)
int upload_phong_tables(hipStream_t s);
int configure_phong();
void launch_ph_linearize(Launcher &L, const Dev &d);
void launch_ph_schur(Launcher &L, const Dev &d, bool check_in_schur = false);
void launch_ph_backsub_eval(Launcher &L, const Dev &d, int fuse_best = 0, bool border_moved = false);   // border_moved: k_pose_update has run with its border work-group
void launch_ph_dogleg_gn(Launcher &L, const Dev &d);
void launch_ph_dogleg_eval(Launcher &L, const Dev &d);
void launch_pose_update(Launcher &L, const Dev &d, int ls_round = 0);
void launch_ph_ls_probe(Launcher &L, const Dev &d, double alpha, int moved, int stage = 0, int rank = 0, int world = 1);     // stage: landmark sharding (1: up to the packed sums, 2: from the exchanged vector on)
void launch_ph_ls_pack(Launcher &L, const Dev &d, int rank, int world);
void launch_ph_ls_accept(Launcher &L, const Dev &d);
void launch_ph_ls_fast(Launcher &L, const Dev &d, bool reduce_eval = false, int x_world = 0);          // bounds: the Armijo test of the full step on the device, then d.ls_rounds blindly enqueued rounds of the search (no-ops unless the test failed); what they cannot finish parks the solver for the host
void launch_ls_resume(Launcher &L, const Dev &d);
// border of free shared blocks (ssba_border.hip): multi-right-hand-side BCR solve + arrowhead system
int configure_border();
void launch_border_solve(Launcher &L, const Dev &d);
void launch_border_finish(Launcher &L, const Dev &d);
void launch_border_scale(Launcher &L, const Dev &d);
// lighting terms on the general layout (ssba_phong_solver.hip): C^-1 + per-observation 6x6 W / Y; the border kernels
void launch_ph_dense_wy(Launcher &L, const Dev &d);
void launch_ph_dense_border(Launcher &L, const Dev &d);
void launch_bcr_multi_rhs(Launcher &L, const Dev &d);
// general structure: dense reduced camera system (ssba_dense.hip)
void launch_dense_schur(Launcher &L, const Dev &d, bool fuse_finish = false);      // fuse_finish (wide system, single GPU): the assembly also finishes the system
void launch_dense_finish(Launcher &L, const Dev &d, bool fused = false);          // fused: launch_dense_schur(.., true) did it (wide system)
void launch_dense_solve(Launcher &L, const Dev &d, int n_rhs_rows = 1);   // rows of the rhs block row to back-substitute
// general layout, banded with tracks of <= WSP observations: 144-row super-blocks (ssba_wide.hip)
int configure_wide();
void launch_wide_schur(Launcher &L, const Dev &d, bool fuse_finish);
void launch_wide_finish(Launcher &L, const Dev &d, bool fused);
bool launch_ctrl_fusable(const Dev &d);       // ssba_kernels.hip: k_check forms the linearisation sums itself on this layout
void launch_wide_solve(Launcher &L, const Dev &d);

}  // namespace ssba
