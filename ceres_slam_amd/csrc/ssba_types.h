// Internal device-side data model of the stereo-BA back end (not part of the ABI).
//
// HBM layout (all fp64 unless noted; see DESIGN.md "Data layout"):
//   * landmarks are re-ordered on the host into *windows*: runs of consecutive
//     landmarks whose observing poses all lie in one list of <= TW poses.  Inside a
//     window an observation is addressed by (landmark, slot) where slot indexes the
//     window's pose list, so the pose index never has to be stored per observation.
//   * observations live in a transposed ELL layout in groups of 64 landmarks:
//         ou/ov/od[(l/64)*TW*64 + slot*64 + (l%64)]
//     so that a wavefront with one landmark per lane reads 512 contiguous bytes per
//     array per slot; a 12-bit mask per landmark marks the occupied slots.
//   * points and per-landmark results are structure-of-arrays over the padded
//     landmark count (component-major), poses stay in the reference's 12-double
//     [t | R row-major] block.
//   * the reduced camera system S is block tridiagonal over super-blocks of SBP
//     consecutive free poses (BD = 6*SBP rows): D[I] (BD x BD, full) and
//     L[I] = S[I, I-1] (BD x BD), followed by rhs / g_p / diag(H_pp) vectors; that
//     whole region is one contiguous "exchange vector" (the only thing ranks have
//     to all-reduce when landmarks are sharded).
#pragma once
#include <stdint.h>

namespace ssba {

constexpr int TW = 12;                  // window width = max track length (ssba.h SSBA_MAX_TRACK)
constexpr int NPAIR = TW * (TW + 1) / 2;  // 78 pose pairs (sa <= sb) per window
constexpr int SBP = 12;                 // poses per super-block of the reduced system
constexpr int BD = 6 * SBP;             // 72 rows per super-block
constexpr int LMG = 64;                 // landmarks per ELL group (= wavefront)
constexpr int SLAB_DOUBLES = NPAIR * 36 + TW * 6;   // per Schur work item
constexpr int MAX_LEVELS = 18;          // plain levels below a parallel top of <= 128 blocks: 2^17 x 128 super-blocks
constexpr int MAX_SLEVELS = 1;          // the separator system of a partitioned (multi-rank) solve: one level, parallel cyclic reduction
constexpr int MAX_SEP = 65;             // separators = ranks - 1
constexpr int NSCAL = 16;
constexpr int NBP = 32;                 // padded width of the border of free shared blocks (nb <= NBP)
constexpr int NBQ = 7;                  // border entries one intensity row touches: [phong 3 | kd | light 3]
constexpr int NPP = 4;                  // pose partials per block: |dx|^2, non-finite, unary-factor candidate cost, unary-factor model change
constexpr int NDL = 6;                  // dogleg partials: |gradient_|^2, |gn|^2, gradient_.gn, |J v|^2, |J gn|^2, Jv.Jgn
constexpr int NBV = 49;                 // per-landmark border sums: S_bb part 28 | rhs_b 7 | diag H_bb 7 | g_b 7

struct Options {   // device copy of ssba_options
    int max_num_iterations, max_nonmono, jacobi_scaling, max_invalid, ignore_convergence, strategy;
    int dogleg_type, pad_;    // 0 TRADITIONAL_DOGLEG, 1 SUBSPACE_DOGLEG
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diag,
        max_lm_diag, function_tolerance, gradient_tolerance, parameter_tolerance;
};

// Trust-region state, lives in device memory; the host only reads it.
struct State {
    Options opt;
    int iteration;            // number of the iteration being worked on
    int terminated, termination_type;
    int need_linearize;       // x changed since the last linearisation
    int just_linearized;      // set by the linearisation kernels, cleared by k_check
    int last_successful;      // previous iteration accepted its step
    int accepted;             // decision of the current iteration (for k_commit)
    int copy_best;            // value of check_count at which x improved on the best cost: k_best copies while they are equal
    int check_count;          // number of (non-terminated) k_check runs
    int step_failed;          // Cholesky breakdown / non-finite step this iteration
    int num_successful, num_unsuccessful, num_invalid;
    int log_count;
    double radius, decrease_factor;
    double x_cost, x_norm, gmax, minimum_cost;
    double candidate_cost, model_cost_change, step_norm, relative_decrease, cost_change;
    // TrustRegionStepEvaluator
    double se_minimum, se_current, se_reference, se_candidate, se_acc_ref, se_acc_cand;
    int se_num_nonmono;
    int dl_reuse;             // dogleg: Gauss-Newton step and gradient of this point are still valid
    double initial_cost;
    // DoglegStrategy [Ceres dogleg_strategy.cc]
    double mu, alpha, dl_step_norm, grad_norm, gn_norm, g_dot_gn, beta, gamma;
    double dl_jv2, dl_jg2, dl_jvg;        // |J v|^2, |J gn|^2, (J v).(J gn) of this linearisation point (kept while dl_reuse)
    double dl_mcc;                        // model cost change of delta = beta gn + gamma v, from the six sums (k_dogleg_interp)
    // SUBSPACE_DOGLEG model: u_i = sub_e[i][0] * gradient_ + sub_e[i][1] * gauss_newton_step_
    int sub_one_dim, sub_pad_;
    double sub_e[2][2], sub_g[2], sub_B[3];
    double ls_alpha;          // projected line search (bounds): the candidate is Plus(x, ls_alpha * delta)
    // bounds: the full step failed the Armijo test on the device (k_ph_ls_fast).  ls_active: the device-side search is under
    // way (1: its first evaluation, phi and phi' at the full step, is still to come; 2: running) -- the blindly enqueued
    // search rounds of the iteration are no-ops without it.  A search that needs more rounds than were enqueued (or whose
    // failure has to restore the full step) is handed to the host: terminated = 1 with termination_type = LS_PENDING parks
    // every kernel until ssba_api.hip has run the search (finish_pending_search).  ls_steps / ls_searches: evaluations /
    // searches completed on the device.
    int ls_pending, ls_active, ls_steps, ls_searches;
};
constexpr int LS_PENDING = 3;   // State::termination_type while a projected line search waits for the host (never reported)

struct IterLog {   // device arrays, capacity entries
    int capacity;
    double *cost, *cost_change, *gmax, *step_norm, *relative_decrease, *radius;
    int *successful;
};

struct BcrLevel {
    int n;            // blocks at this level
    double *D, *L, *r;  // n blocks each (L[0] unused)
    double *YU;       // n/2 blocks: G^-1 L[i+1]^T for odd i
    double *B;        // n x BD x NBP extra right-hand sides (border columns), or nullptr
    int pin;          // chain with pinned ends (partitioned solve) and n even: the last block (odd index) is
                      // NOT eliminated at this level but carried over as the last block of the next level
    const int *pos;   // n: level-0 position of each block (i << level for a plain plan)
};

// Parallel cyclic reduction of the top of the plan (ssba_bcr.hip): from level `level` on (n <= PCR_MAX_BLOCKS blocks)
// every block eliminates BOTH neighbours at distance 2^k in every step, so after log2(n) steps the blocks are
// decoupled and solved directly: no back-substitution sweep over those levels.  Works in place on the level's D
// and r; L ping-pongs through Lbuf, the factor products live in YL / YU / yr.
struct PcrPlan {
    int level, n, steps;            // level < 0: not used
    double *Lbuf, *LbufT, *YL, *YU, *yr;   // n blocks each (yr: n x BD); LbufT = the couplings transposed (the "U" operands)
    // free shared blocks: the border columns follow through the same steps afterwards, so every step keeps its
    // products (YL, YU hold steps x n blocks) and its factor Gs; Bb / yB are the columns and G^-1 of them (n x BD x NBP)
    int keep;
    double *Gs, *Bb, *yB;
    // partitioned solve: the first / last block of the level is a separator shared with the neighbouring rank.  A pinned
    // block is never eliminated: it folds its neighbours in like every other block, but the others KEEP their coupling
    // to it instead of folding it (Lbuf[i] = S[i, first] once i - 2^k < 0, Ubuf[i] = S[i, last] once i + 2^k > n - 1),
    // so after the last step every interior block is coupled to the pinned ones only and the pinned rows hold the
    // Schur complement of the chain interior.
    int pin0, pin1;
    double *Ubuf;
};
constexpr int PCR_MAX_BLOCKS = 128;

// One launch per step of the parallel cyclic reduction (ssba_bcr_mfma.hip, k_bcr_factor_mf<.., FUSED>): the products the
// next step needs are Gram products of ONE block's own factor outputs,
//     GUU(b) = YU(b)^T YU(b) -> D'(b + s),   GLL(b) = YL(b)^T YL(b) -> D'(b - s),   GUL(b) = YU(b)^T YL(b) -> the couplings
//     S'[b + s, b - s] = -GUL(b) and S'[b - s, b + s] = -GUL(b)^T,
// so the block that has just factored itself forms them from LDS (no reduce launch, no staging of three neighbours'
// operands from HBM) and the NEXT step's load phase assembles  D(e) - GUU(e - s) - GLL(e + s)  on the way into the
// registers.  Everything ping-pongs by step parity (a step reads what the previous one wrote).
struct PcrFused {
    int on;
    double *Dpp[2], *rpp[2];                    // assembled D (upper tiles) and r of a step: n x BD x BD, n x BD
    double *GLL[2], *GUU[2], *GUL[2], *GULT[2]; // n x BD x BD each (GLL / GUU: upper tiles)
    double *gL[2], *gU[2];                      // YL^T yr, YU^T yr: n x BD
    // chains with pinned ends (partitioned solve): a coupling that points at a pinned block is never folded again; the
    // block that holds it saves it once (canonical orientation: rows of the block) and reads it back in every later step
    double *Lkeep, *Ukeep;                      // n x BD x BD each, or null
};

// Long tracks on block-cyclic machinery (ssba_wide.hip): landmarks with 13 .. WSP observations (free poses within a span of
// WSP) give a reduced system that is block tridiagonal over super-blocks of WSP = 24 poses (WBD = 144 rows).  The problem
// keeps the general (landmark-major) observation layout; the middle of the iteration -- Schur product, assembly, reduced
// solve -- runs on 144-wide blocks: matrix-core Schur items over windows of 24 consecutive free poses, a gather into the
// block-tridiagonal system, parallel cyclic reduction with one factor + one reduce launch per step.  The struct lives in
// device memory (Dev has no room for it by value); the host keeps a copy for the launch shapes.
constexpr int WSP = 24;                 // poses per wide super-block = slots of a wide window
constexpr int WBD = 6 * WSP;            // 144 = 9 tiles of 16
constexpr int WNT = WBD / 16;           // 9
constexpr int WSLAB_TILES = WNT * (WNT + 1) / 2 + WNT;      // 45 upper tiles + 9 tiles of the gradient column
constexpr int WSLAB_DOUBLES = WSLAB_TILES * 256;
struct WideSys {
    int n, steps, n_items, n_blk;       // super-blocks, PCR steps = ceil(log2 n), Schur items, non-zero 6x6 blocks (a <= b)
    // exchange vector of the wide system: [D (n x WBD x WBD, row-major) | L (n: L[I] = S[I, I-1]) | rhs (n x WBD)]
    double *xw;
    uint64_t off_L, off_rhs, count;
    // Schur items: landmarks [begin, end) whose free poses lie in [base, base + WSP); slot s = free pose base + s
    const uint32_t *item_begin, *item_end, *item_base;
    const uint32_t *slot_obs;           // Lpad x WSP: observation (index into dn_u / dn_v / dn_d) of (landmark, slot) or 0xFFFFFFFF
    double *slab;                       // n_items x WSLAB_DOUBLES, tile-major
    // gather lists: per block the (item * WSP * WSP + slot_a * WSP + slot_b) it sums, per free pose the (item * WSP + slot)
    const uint32_t *blk_a, *blk_b, *blk_start, *blk_contrib;
    const uint32_t *prow_start, *prow_contrib;
    // parallel cyclic reduction
    double *U;                          // n blocks: U[e] = S[e, e + stride] from the second step on
    double *YL, *YU, *yr;               // G^-1 [L | U | r] of the current step
};

struct Dev {
    // camera, stiffness, loss
    double fu, fv, cu, cv, b;
    double S[9];
    double huber_a;
    // sizes
    int P, nfree, Nsb, nf_pad;       // nf_pad = Nsb*SBP
    int Lpad, n_groups, n_windows, n_slabs, n_sblk;
    int n_lm_blocks;                 // blocks of 256 landmarks (partials)
    uint32_t n_obs;
    // parameters
    double *poses, *cand_poses, *best_poses, *init_poses;   // P*12
    double *pts, *cand_pts, *best_pts, *init_pts;           // 3*Lpad, component-major
    const int *pose_free;            // P   -> free index or -1
    const int *free_pose;            // nfree -> pose id
    // observations
    const double *ou, *ov, *od;      // n_groups*TW*64
    const uint32_t *lm_mask;         // Lpad
    const uint32_t *lm_win;          // Lpad
    const uint32_t *win_pose;        // n_windows*TW (0xFFFFFFFF = empty slot)
    const uint32_t *pose_obs_start;  // P+1
    const uint32_t *pose_obs_ref;    // n_obs: landmark*16 + slot
    // linearisation
    double *hll, *gl, *sl;           // 6*Lpad, 3*Lpad, 3*Lpad (component-major); config 3: 21 / 6 / 6 rows
    double *hpp, *gp;                // P*21 (upper, row-major packed), P*6
    double *sp;                      // nf_pad*6 Jacobi scale of free poses
    // Schur
    const uint32_t *slab_win, *slab_lm_begin, *slab_lm_end;  // n_slabs
    double *slab;                    // n_slabs*SLAB_DOUBLES
    const uint32_t *sblk_a, *sblk_b, *sblk_start, *sblk_contrib;  // n_sblk(+1), contributions
    const uint32_t *prow_start, *prow_contrib;                   // nfree+1, (slab*TW+slot)
    // reduced system (exchange vector) and solution
    double *xv;                      // [D0 | L0 | rhs | gpx | hdiag | scal]
    uint64_t off_D, off_L, off_rhs, off_gp, off_hdiag, off_scal, xv_count;
    double *x0;                      // nf_pad*6 pose step (LM step / dogleg Gauss-Newton step)
    double *vp, *vl, *dl_gn;         // dogleg: s^2 g / D^2 of poses (P*6) and landmarks (3*Lpad), GN landmark step
    double *part_dl;                 // dogleg partial sums: (n_lm_blocks + n_pose_blocks + 1 [border]) * NDL
    int n_levels;
    BcrLevel lev[MAX_LEVELS];
    PcrPlan pcr;
    PcrFused pcrf;                   // the fused-step buffers of `pcr` (single GPU, no border columns)
    // partitioned solve (one rank per contiguous chain of super-blocks): this rank's chain is
    // super-blocks [chain0, chain1]; the end it shares with a neighbouring rank is pinned (pin0: rank > 0, pin1:
    // rank < world - 1), the world - 1 shared blocks of all ranks form the separator system
    int part, rank, world, n_sep, chain0, chain1, pin0, pin1;
    int sep_sb[MAX_SEP];             // separator s = super-block sep_sb[s], shared by the ranks s and s + 1
    BcrLevel slev[MAX_SLEVELS];      // slev[0]: the separator system (n_sep blocks) inside the exchange vector
    PcrPlan spcr;                    // its parallel cyclic reduction (replicated on every rank)
    PcrFused spcrf;                  // ... one launch per step (the separator plan has no pinned ends)
    double *sepv;                    // [Dsep | Lsep | rhs | gp | hdiag | scal]: the (small) exchange vector
    uint64_t soff_D, soff_L, soff_rhs, soff_gp, soff_hdiag, soff_scal, sepv_count;
    double *xsep;                    // n_sep * BD separator solution
    // reductions
    double *part_lin;                // n_lm_blocks*4: cost, |x_pts|^2, max|g_l|, -
    double *part_chk;                               // [nfree][2]: projected-gradient norm and |x|^2 of a pose (k_assemble_reduced -> k_check)
    double *part_eval;               // n_lm_blocks*4: cand cost, mcc, |dl|^2, nonfinite
    double *part_pose;               // ceil(P/256)*2: |dx_pose|^2, nonfinite
    double *scal2;                   // NSCAL: second exchange vector
    double *scal_dl;                 // NSCAL: the six dogleg sums of this rank (landmark sharding: summed over the ranks between k_dogleg_sum and k_dogleg_interp)
    double *gmax_l;                  // 1: landmark part of the gradient max norm
    int n_pose_blocks;
    State *st;
    IterLog log;
    unsigned long long *dbg;         // in-kernel stamps (diagnostic builds only, -DSSBA_STAMPS)
    // config 3 (stereo + Phong intensity + normal blocks; landmark block = [position | normal])
    int phong, light_type;
    double int_stiff, Sn[9];
    double *nrm, *cand_nrm, *best_nrm, *init_nrm;   // 3*Lpad, component-major
    const double *oi, *onx, *ony, *onz;             // ELL: observed intensity and normal
    const uint32_t *lm_mat;                         // Lpad -> material
    // shared blocks, packed ambient state [light 3 | phong 3M (ka,ks,alpha) | texture M (kd)]
    int M, nsh;
    double *sh, *cand_sh, *best_sh, *init_sh;
    double *cinv;                                   // 21*Lpad damped landmark block inverse C^-1 (packed upper)
    double *cfac;                                   // 21*Lpad its factor M = L^-1 (packed lower, row-major): C^-1 = M^T M
    double *dlm;                                    // 6*Lpad landmark step (local coordinates)
    // free shared blocks = dense border of the reduced system (nb columns; offsets, -1 = constant)
    int nb, b_light, b_phong, b_tex;
    // closure border (stereo problems with a loop closure): the last nfree - nchain free poses are border columns
    // (6 each) of the block-tridiagonal system instead of chain rows; cb_* = the slab contributions of their blocks
    int cb, nchain, n_cb;
    const uint32_t *cb_a, *cb_b, *cb_start, *cb_contrib;
    const uint32_t *pose_mat_start;                 // P*(M+1): pose_obs_ref is sorted by material inside a pose
    double *lmV, *lmH, *lmG;                        // per landmark: H_lb 42, H_bb 28, g_b 7 (component-major)
    // windowed layout: the pose rows of the border ride in the Schur product (k_ph_schur_windows<true>)
    double *lmMV;                                   // per landmark M V_j (42, component-major); null: k_ph_border_poses instead
    double *Hpb, *HpbL;                             // nf_pad*6 x NBP   H_pb (k_ph_hpb, on linearisation); P x M x 18 light-column partials
    double *slabB;                                  // n_slabs x 72 x NBP   border tiles of the Schur items
    double *part_b;                                 // n_lm_blocks * M * NBV
    double *Spb;                                    // nf_pad*6 x NBP   S_pb (rows of free poses)
    double *Zb;                                     // nf_pad*6 x NBP   S_pp^-1 S_pb
    double *part_g;                                 // gram partials: n_gram x (NBP*NBP + NBP)
    int n_gram;
    // border system, NBP-strided: Sbb (NBP*NBP) | rhsb | gb | hb | sb | db | vb
    double *bsys;
    // bounds on the shared blocks (SetParameterLower/UpperBound): [ka, ks, alpha, kd]; projected Plus +
    // Armijo line search when constrained
    int constrained, ls_rounds;                     // ls_rounds: search evaluations enqueued per iteration for the device-side Armijo search
    double blo[4], bhi[4];
    double *part_ls;                                // n_lm_blocks * NLS line-search partials
    double *ls_out;                                 // NLS_OUT scalars the host reads per probe, then NLS_MACH doubles holding the device-side search's state (ssba_linesearch.h: Armijo)
    // unary pose residual blocks (pose prior, sun sensor), sorted by pose
    int pf_owner;                                   // landmark sharding: this rank adds the unary blocks to the sums that are exchanged (H_pp, g_p, cost); every rank evaluates them at the candidate
    int n_pf, pos_const;                            // pos_const: every position block constant (--multistage stage 2; lighting problems)
    const uint32_t *pf_start;                       // P+1
    const int *pf_type;                             // F
    const double *pf_data, *pf_S, *pf_huber;        // F*18, F*36, F
    double *pf_cost;                                // P: cost of the factors of each pose at the linearisation point
    // general structure (tracks longer than TW, loop closures): dense reduced camera system (ssba_dense.hip)
    int dense, n_dn, dn_pad;                        // n_dn = 6 * nfree, dn_pad = n_dn rounded up to DN_BS (= row stride of dn_S)
    const uint32_t *dn_lm_start;                    // Lpad+1: observations of a landmark, landmark-major
    const uint32_t *dn_obs_pose;                    // N
    const double *dn_u, *dn_v, *dn_d;               // N
    const double *dn_Sobs;                          // N*9 or null: one stiffness per stereo residual block (dataset_vo_sun.cpp:56-65)
    const uint32_t *dn_pose_start, *dn_pose_obs;    // P+1, N: landmark-major observation indices of a pose
    const uint32_t *dn_obs_lm;                      // N: landmark of an observation
    const double *dn_prec;                          // N x 4 or null: (u, v, d, landmark as integer bits) in the order of dn_pose_obs
    double *dn_W, *dn_Y;                            // per observation: ONE buffer of N*18 (lighting: N*36), Z = W M^T with C^-1 = M^T M
    const uint32_t *dn_zpos;                        // observation -> its record in dn_Y (pose-major)
    double *dn_Mg;                                  // M g_l per landmark (3 or 6 x Lpad)
    // blocks (a <= b) of the reduced system with the observation pairs (of one landmark) that contribute to each
    int dn_nblk;
    const uint32_t *dn_blk_a, *dn_blk_b, *dn_blk_start, *dn_pair_a, *dn_pair_b;
    const uint32_t *dn_blk_rf_start, *dn_blk_rf;    // per block: relative-pose entries (first half | bit 31: block is J_2^T J_1), or null
    // block-level (DN_BS) symbolic factorisation: non-zero block rows below the diagonal of every block column
    // (the rhs row last), the tile pairs of every trailing update, and the non-zero block columns of every block row
    const uint32_t *dn_rows, *dn_ti, *dn_tk, *dn_cols, *dn_row_start;
    const uint32_t *dn_ztile;                       // the structurally non-zero DN_BS tiles of the factor (row << 16 | column), right-hand-side row included:
    int dn_nztile;                                  // what an iteration zero-fills instead of the whole dense array
    double *dn_S;                                   // (dn_pad + DN_BS) x dn_pad, row-major, lower triangle; row dn_pad holds the
                                                    // right-hand side, so the factorisation leaves L^-1 rhs there
    // general layout, banded with tracks of <= WSP observations: block-tridiagonal system of 144-wide blocks (ssba_wide.hip)
    const WideSys *wide;                            // device copy, or null
};
// Every kernel takes Dev by value: explicit kernel arguments are limited to 4 KB, and the hidden arguments of the code
// object (256 B) and the handful of scalars beside Dev have to fit next to it.
static_assert(sizeof(Dev) <= 4096 - 256 - 128, "Dev no longer fits the kernel-argument segment next to the scalars and hidden arguments");
constexpr int DN_BS = 64;                           // block size of the dense Cholesky
constexpr int NLS = 6;        // per block: cost, phi', |dx_l|^2, nonfinite, max|delta_l|, g_l . delta_l
constexpr int NLS_MACH = 24;
constexpr int NLS_OUT = 8;    // cost, phi', |dx_l|^2, nonfinite_l, max|delta|, g . delta, valid, x_cost
// landmark sharding with bounds: the exchange vector of a line-search evaluation, behind the search state in ls_out:
// [cost, phi', |dx_l|^2, nonfinite_l, g_l . delta_l, 0, 0, 0 | max|delta_l| of rank r in slot 8 + r (a maximum through the SUM exchange)]
constexpr int NLS_X = 8, NLS_X_RANKS = 64;
constexpr int BS_SBB = 0, BS_RHS = NBP * NBP, BS_G = BS_RHS + NBP, BS_H = BS_G + NBP, BS_S = BS_H + NBP,
              BS_DB = BS_S + NBP, BS_VB = BS_DB + NBP, BS_COUNT = BS_VB + NBP;   // VB: dogleg v_b = s^2 g / D^2

}  // namespace ssba
