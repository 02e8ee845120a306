/*
 * ssba.h -- C ABI of the MI355X-native stereo bundle-adjustment back end.
 *
 * This is the drop-in boundary for the one hot path of utiasSTARS/ceres-slam:
 * everything that happens inside `ceres::Solve` for the stereo BA problems its
 * drivers build.  The reference has no FFI or plugin table;
 * its "plugin point" is the Ceres C++ API as used by tests/dataset_vo.cpp:22-85 and
 * tests/dataset_ba_phong.cpp:26-255.  Each entry point below names the reference
 * call(s) it replaces (file:line into the reference repo).  A header-only C++ shim
 * with the reference's call shapes on top of this ABI is include/ceres_slam_amd/
 * ceres_shim.hpp; the binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no exceptions cross the ABI; every function returns an ssba_status
 *    (0 = ok, negative = error); ssba_status_string() names it.
 *  - parameter memory stays OWNED BY THE CALLER (the reference hands Ceres raw
 *    pointers into std::vector<SE3>/Point: dataset_vo.cpp:51-53); the library keeps
 *    device mirrors and writes the caller's blocks back in place when a solve ends
 *    with a usable solution -- exactly Ceres's contract.
 *  - a pose block is 12 doubles [t(3) | R row-major(9)] = T_c_g
 *    (include/ceres_slam/geometry/se3group.hpp:425-429); a point block is 3 doubles.
 *  - a handle is single-owner and not thread-safe (like ceres::Problem).
 *  - all arithmetic is IEEE fp64 on the GPU; there is NO CPU fallback: without a
 *    usable HIP device every compute entry point returns SSBA_ERR_NO_DEVICE.
 */
#ifndef SSBA_H_
#define SSBA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libssba.so is built with -fvisibility=hidden and a linker version script: the entry points declared in this header are
 * the ONLY dynamic symbols it defines (tests/test_capi_symbols.py compares `nm -D --defined-only` with this file). */
#if defined(__GNUC__) || defined(__clang__)
#define SSBA_API __attribute__((visibility("default")))
#else
#define SSBA_API
#endif

#define SSBA_VERSION 1
/* longest landmark track (observations of one landmark) of the windowed layout; problems with longer tracks, or
 * with co-observing free poses more than 12 apart, run the general-structure kernels (ssba_stats.general_structure = 1)
 * -- except a loop closure whose far side is at most 5 states and whose landmarks keep <= 12 observations: those
 * states become a border of the block-tridiagonal reduced system and the windowed kernels stay (general_structure = 2;
 * any number of poses).  The border covers Levenberg-Marquardt solves; a DOGLEG solve or ssba_pose_covariance on such a
 * handle runs the symbolic phase again and continues on the general path (general_structure becomes 1);
 * SSBA_NO_CLOSURE_BORDER=1 in the environment of ssba_finalize selects the general path from the start */
#define SSBA_MAX_TRACK 12

typedef struct ssba_problem ssba_problem;

typedef enum {
    SSBA_OK = 0,
    SSBA_ERR_INVALID_ARGUMENT = -1,
    SSBA_ERR_HIP = -2,              /* a HIP runtime call failed (see ssba_last_error) */
    SSBA_ERR_NUMERICAL_FAILURE = -3,
    SSBA_ERR_NOT_FINALIZED = -4,
    SSBA_ERR_NO_DEVICE = -5,        /* no usable gfx950 device: no CPU fallback exists  */
    SSBA_ERR_UNSUPPORTED = -6,      /* structure outside this build's envelope          */
    SSBA_ERR_STATE = -7,            /* call sequence error                              */
    SSBA_ERR_TIMEOUT = -8           /* an RCCL set-up call did not return in time       */
} ssba_status;

/* StereoCamera<double> intrinsics (include/ceres_slam/stereo_camera.hpp:159-163) */
typedef struct { double fu, fv, cu, cv, b; } ssba_camera;

/* The ceres::Solver::Options fields the reference drivers set
 * (tests/dataset_vo.cpp:65-74, tests/dataset_ba_phong.cpp:79-88) plus the Ceres 1.x
 * defaults that shape the trust-region loop.  Unlisted Ceres options keep their
 * defaults.  trust_region_strategy_type: LEVENBERG_MARQUARDT (what dataset_vo.cpp runs)
 * and DOGLEG with dogleg_type TRADITIONAL_DOGLEG / SUBSPACE_DOGLEG (dataset_ba_phong.cpp:85-86,
 * dataset_vo_sun.cpp:142-143), both on an exact Schur-complement solve.  linear_solver_type
 * has no field: every solve is the reduced-camera-system solve Ceres' SPARSE_SCHUR performs. */
typedef struct {
    int32_t max_num_iterations;                 /* 50; drivers: 1000                   */
    int32_t use_nonmonotonic_steps;             /* 0;  drivers: 1                      */
    int32_t max_consecutive_nonmonotonic_steps; /* 5                                   */
    int32_t jacobi_scaling;                     /* 1                                   */
    int32_t max_num_consecutive_invalid_steps;  /* 5                                   */
    int32_t minimizer_progress_to_stdout;       /* 0                                   */
    int32_t num_threads;                        /* accepted, ignored (GPU)             */
    int32_t num_linear_solver_threads;          /* accepted, ignored (GPU)             */
    double initial_trust_region_radius;         /* 1e4                                 */
    double max_trust_region_radius;             /* 1e16                                */
    double min_trust_region_radius;             /* 1e-32                               */
    double min_relative_decrease;               /* 1e-3                                */
    double min_lm_diagonal;                     /* 1e-6                                */
    double max_lm_diagonal;                     /* 1e32                                */
    double function_tolerance;                  /* 1e-6                                */
    double gradient_tolerance;                  /* 1e-10                               */
    double parameter_tolerance;                 /* 1e-8                                */
    int32_t trust_region_strategy_type;         /* 0 = LEVENBERG_MARQUARDT (Ceres default, what
                                                   tests/dataset_vo.cpp runs); 1 = DOGLEG
                                                   (tests/dataset_ba_phong.cpp:85)                */
    int32_t dogleg_type;                        /* 0 = TRADITIONAL_DOGLEG (Ceres default); 1 =
                                                   SUBSPACE_DOGLEG (tests/dataset_ba_phong.cpp:86,
                                                   tests/dataset_vo_sun.cpp:143)                  */
} ssba_options;

/* ceres::TerminationType values the path can produce */
enum { SSBA_CONVERGENCE = 0, SSBA_NO_CONVERGENCE = 1, SSBA_FAILURE = 2 };

/* ceres::Solver::Summary fields the drivers print (summary.BriefReport():
 * tests/dataset_vo.cpp:82) plus per-phase device timings. */
typedef struct {
    int32_t termination_type;
    int32_t num_iterations;          /* recorded iterations incl. iteration 0          */
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    double initial_cost;
    double final_cost;
    double total_time_s;             /* wall time of ssba_solve incl. write-back       */
    double device_time_s;            /* GPU time of the iteration loop (HIP events)    */
    /* bounds [Solver::Summary::num_line_search_steps]: evaluations of the projected line search, and how many searches
     * ran entirely on the device / were finished by the host (SSBA_LS_ROUNDS; unset: 2 evaluations enqueued per iteration, up to 4 once a search needed more) */
    int32_t num_line_search_steps;
    int32_t num_line_searches_on_device;
    int32_t num_line_searches_by_host;
    int32_t reserved_;
} ssba_summary;

/* ---- lifetime --------------------------------------------------------------------- */
/* replaces: ceres::Problem problem; + the shared StereoCamera captured by every
 * functor (tests/dataset_vo.cpp:26, stereo_reprojection_error.hpp:73).
 * device < 0 selects the current HIP device. */
SSBA_API int ssba_create(const ssba_camera *camera, int device, ssba_problem **out);
SSBA_API int ssba_destroy(ssba_problem *p);

/* ---- problem building ------------------------------------------------------------- */
/* replaces: the pose blocks passed to AddResidualBlock + SetParameterization(pose,
 * SE3Perturbation) (tests/dataset_vo.cpp:51-58; include/ceres_slam/perturbations.hpp:45-76).
 * `poses` is num*12 doubles, caller-owned, updated in place by a solve. */
SSBA_API int ssba_add_pose_blocks(ssba_problem *p, double *poses, uint32_t num);
/* replaces: the point blocks passed to AddResidualBlock (tests/dataset_vo.cpp:53);
 * `points` is num*3 doubles, caller-owned, updated in place. */
SSBA_API int ssba_add_point_blocks(ssba_problem *p, double *points, uint32_t num);
/* replaces: StereoReprojectionErrorAutomatic::Create(camera, obs, stiffness) +
 * problem.AddResidualBlock(cost, NULL, pose_k, point_j) for every observation
 * (include/ceres_slam/stereo_reprojection_error.hpp:59-69; tests/dataset_vo.cpp:39-56).
 * uvd is num*3 (u_l, v_l, d); stiffness is the shared 3x3 row-major Sigma^{-1/2}
 * (tests/dataset_vo.cpp:29-32).  May be called several times; order is kept. */
SSBA_API int ssba_add_stereo_observations(ssba_problem *p, const uint32_t *pose_index,
                                 const uint32_t *point_index, const double *uvd,
                                 uint64_t num, const double stiffness[9]);
/* replaces: problem.SetParameterBlockConstant / SetParameterBlockVariable(pose)
 * (tests/dataset_vo.cpp:62; tests/dataset_ba_phong.cpp:76,219-243) */
SSBA_API int ssba_set_pose_constant(ssba_problem *p, uint32_t pose, int is_constant);
/* replaces: passing `new ceres::HuberLoss(a)` instead of NULL to AddResidualBlock
 * (call-site shape tests/dataset_vo_sun.cpp:89-95); a <= 0 restores the NULL loss */
SSBA_API int ssba_set_huber_loss(ssba_problem *p, double a);
/* Uploads the problem graph and builds the device-side structure (once per graph). */
SSBA_API int ssba_finalize(ssba_problem *p);

/* ---- solving ---------------------------------------------------------------------- */
/* Ceres 1.x defaults */
SSBA_API void ssba_default_options(ssba_options *o);
/* replaces: ceres::Solve(options, &problem, &summary) (tests/dataset_vo.cpp:81).
 * Blocking.  Re-uploads the caller's current parameter values first. */
SSBA_API int ssba_solve(ssba_problem *p, const ssba_options *o, ssba_summary *s);
/* replaces: summary.BriefReport() (tests/dataset_vo.cpp:82) */
SSBA_API int ssba_brief_report(const ssba_summary *s, char *buf, size_t buf_len);

/* Stepwise form of ssba_solve for the bench harness and for multi-GPU drivers:
 * begin (upload + iteration 0), `step` enqueues n trust-region iterations on the
 * stream without host synchronisation, end synchronises and writes back.
 * With ignore_convergence != 0 the convergence tests are skipped so that exactly n
 * iterations of full work are executed (bench.py's timed region). */
SSBA_API int ssba_solve_begin(ssba_problem *p, const ssba_options *o, int ignore_convergence);
SSBA_API int ssba_solve_step(ssba_problem *p, int n);
SSBA_API int ssba_solve_end(ssba_problem *p, ssba_summary *s);
/* restores the parameter values uploaded by ssba_solve_begin and restarts the
 * trust-region state (device-to-device, asynchronous) */
SSBA_API int ssba_solve_restart(ssba_problem *p);
SSBA_API int ssba_synchronize(ssba_problem *p);

/* Per-iteration log of the last solve (the columns of Ceres's progress table).
 * Arrays of `capacity` entries or NULL; returns the number of recorded iterations. */
SSBA_API int ssba_iteration_log(ssba_problem *p, int32_t capacity, double *cost, double *cost_change,
                       double *gradient_max_norm, double *step_norm, double *relative_decrease,
                       double *trust_region_radius, int32_t *step_is_successful);

/* ---- streams, multi-GPU exchange, instrumentation ---------------------------------- */
/* Run on a caller-provided hipStream_t (e.g. torch's current stream). */
SSBA_API int ssba_set_stream(ssba_problem *p, void *hip_stream);
/* Landmark sharding: every rank adds ALL pose blocks and only ITS landmarks and
 * observations.  The library calls `fn` at the two exchange points of an iteration
 * with a device buffer of `count` doubles that must be reduced in place over all
 * ranks (op 0 = sum, 1 = max) on the stream given to ssba_set_stream -- e.g. a
 * torch.distributed all_reduce over RCCL.  fn == NULL (default) = single GPU. */
typedef int (*ssba_exchange_fn)(void *ctx, void *device_buffer, uint64_t count, int op);
SSBA_API int ssba_set_exchange(ssba_problem *p, ssba_exchange_fn fn, void *ctx);
/* Declares (before ssba_finalize) that the landmarks are sharded over `world_size` ranks:
 * every non-constant pose is then kept in the reduced system even if THIS rank holds no
 * observation of it, so that all ranks agree on the layout of the exchanged system. */
SSBA_API int ssba_set_distributed(ssba_problem *p, int world_size, int rank);
/* Partitioned reduced solve (after ssba_set_distributed, before ssba_finalize).  The free poses are
 * grouped into super-blocks of 12 (free-pose index / 12); separator_superblocks (world_size + 1 ascending
 * entries, first = 0, last = the last super-block) says that rank r's landmarks only observe poses of the
 * super-blocks [sep[r], sep[r+1]] -- i.e. the landmark ranges are cut where the co-visibility band crosses
 * a super-block boundary, so that consecutive ranks share exactly one super-block.  Each rank then
 * eliminates the interior of its own chain of the block-tridiagonal reduced camera system and only the
 * shared chain ends (the separator system: (world_size - 1) blocks of 72 x 72) are summed over the ranks and
 * solved everywhere -- < 1 MB per iteration instead of the whole reduced system, and no redundant solve of
 * the other ranks' chains.  At the end of the solve the poses are gathered (one more exchange), so every
 * rank returns the complete trajectory.  Without this call the ranks sum the whole reduced system and
 * each solves all of it (works for any sharding).  num = 0 clears the partition. */
SSBA_API int ssba_set_partition(ssba_problem *p, const uint32_t *separator_superblocks, uint32_t num);
/* Native exchange: the library itself enqueues ncclAllReduce (RCCL over xGMI) on the solver's stream at the exchange
 * points -- one process per GPU, no host language in the loop (this is what a sharded
 * tests/dataset_vo.cpp:22-85 links against).  Rank 0 calls ssba_rccl_unique_id (128 bytes) and hands the bytes to the
 * other ranks by whatever the application has (MPI, a file, torch.distributed); every rank then calls ssba_set_rccl
 * AFTER ssba_set_distributed(world_size, rank) -- a collective call: it returns once all ranks have joined.  It
 * replaces any callback set with ssba_set_exchange.  librccl.so is loaded on first use. */
#define SSBA_RCCL_UNIQUE_ID_BYTES 128
SSBA_API int ssba_rccl_unique_id(void *out, uint64_t size);
SSBA_API int ssba_set_rccl(ssba_problem *p, const void *unique_id, uint64_t size);
/* Both set-up calls above are time-limited (RCCL's own calls are not): after SSBA_RCCL_TIMEOUT_S seconds (environment,
 * default 180) they return SSBA_ERR_TIMEOUT and ssba_last_error() says which RCCL call did not return, which librccl.so
 * file the process had mapped (PyTorch ships its own copy) and its version, and the IPC / debug environment; the caller
 * can fall back to ssba_set_exchange.  ssba_rccl_describe writes that description of the loaded library at any time;
 * ssba_rccl_ranks returns ncclCommCount of the handle's communicator (0 without one). */
SSBA_API int ssba_rccl_describe(char *buf, uint64_t size);
SSBA_API int ssba_rccl_ranks(ssba_problem *p, int *count);
/* number of doubles in the per-iteration reduced-system exchange */
SSBA_API int ssba_exchange_size(ssba_problem *p, uint64_t *count);

/* Kernel timing with HIP events on the library's stream.  mode 0 = off, 1 = time every
 * kernel class (eager launches).  ssba_kernel_times returns up to `capacity` rows. */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
} ssba_kernel_time;
SSBA_API int ssba_set_kernel_timing(ssba_problem *p, int mode);
SSBA_API int ssba_kernel_times(ssba_problem *p, ssba_kernel_time *rows, int32_t capacity, int32_t *num);

/* Diagnostic builds only (-DSSBA_STAMPS: tools/stamps_bcr.py, tools/bcr_bench.hip): copies the first n (<= 8192) in-kernel
 * time stamps of the handle's debug buffer; the buffer stays zero in a normal build. */
SSBA_API int ssba_debug_stamps(ssba_problem *p, unsigned long long *out, int n);

/* Problem statistics after ssba_finalize. */
typedef struct {
    uint32_t num_poses, num_free_poses, num_points, num_active_points;
    uint64_t num_observations;
    uint32_t num_windows;          /* landmark groups sharing a <=12-pose window        */
    uint32_t num_superblocks;      /* 72x72 super-blocks of the block-tridiagonal S     */
    uint32_t num_reduced_blocks;   /* non-zero 6x6 blocks of S (upper incl. diagonal)   */
    uint32_t pose_bandwidth;       /* max free-pose index distance of co-observers      */
    uint64_t device_bytes;         /* device memory held by the handle                  */
    uint32_t general_structure;    /* 1: tracks > SSBA_MAX_TRACK or span > 12 poses -> dense reduced system; 2: windowed layout + closure border */
    uint32_t pcr_blocks;           /* blocks handed to the parallel cyclic reduction (<= 128), 0 = plain BCR */
    uint32_t pcr_fused;            /* 1: one launch per step of that reduction (no border columns, single GPU; SSBA_NO_PCR_FUSED=1 in the environment of a solve keeps factor + reduce launches) */
    uint32_t wide_superblocks;     /* > 0: general layout whose tracks have 13..24 observations (free poses within a span of 24): the reduced system is block tridiagonal over this many super-blocks of 24 poses (144 rows), solved by parallel cyclic reduction (general_structure stays 1; SSBA_NO_WIDE=1 at ssba_finalize keeps the blocked Cholesky) */
} ssba_stats;
SSBA_API int ssba_get_stats(ssba_problem *p, ssba_stats *st);

/* Destroyed handles leave their device buffers, pinned host buffers and streams in a process-wide cache for the next
 * handle (drivers that solve thousands of small windows are bound by hipMalloc / hipFree otherwise); SSBA_POOL_MB caps
 * the cached device bytes (default 2048, 0 = no caching).  This call frees what is cached now. */
SSBA_API int ssba_release_cached_memory(void);

/* ---- test hooks (parity tests call these through the C ABI) ------------------------ */
/* Normal-equation blocks at the caller's current parameters, in user index order:
 * cost, g_p (P*6), g_l (L*3), H_pp (P*36 row-major), H_ll (L*9).  Blocks of constant
 * poses are zero.  Any output may be NULL. */
SSBA_API int ssba_evaluate(ssba_problem *p, double *cost, double *g_p, double *g_l, double *H_pp,
                  double *H_ll);
/* Dense reduced camera system S (n*n row-major, n = 6*num_free_poses) and rhs (n) for
 * trust-region radius `radius` at the caller's current parameters (S * delta_p = rhs),
 * and the step the device solver computes from it: delta_p (P*6), delta_l (L*3) and the
 * model cost change.  Any output may be NULL. */
SSBA_API int ssba_lm_step(ssba_problem *p, const ssba_options *o, double radius, double *S,
                 double *rhs, double *delta_p, double *delta_l, double *model_cost_change);
/* The scalar state machine of the projected line search [Ceres 1.x line_search.cc ArmijoLineSearch::DoSearch, CUBIC
 * interpolation; bounds: tests/dataset_ba_phong.cpp:143-181] replayed on a given sequence of evaluations: phi(0),
 * phi'(0), max |direction|, then values[k] / gradients[k] = phi, phi' at the k-th step it asks for (steps_out[k], capacity
 * n).  on_device = 0 runs it on the host (no GPU needed), 1 in a one-lane kernel on `device` -- the solver uses both (the
 * search rounds enqueued with an iteration, and the searches handed back to the host) and they must agree bit for bit.
 * Returns the number of steps asked for (<= n), a negative status on error; *optimal_step = the accepted step or -1. */
SSBA_API int ssba_armijo_trace(const double *values, const double *gradients, int32_t n, double initial_cost,
                      double initial_gradient, double dir_max_norm, double *steps_out, double *optimal_step,
                      int32_t on_device, int32_t device);

/* ---- config 3: lighting terms on the same graph (tests/dataset_ba_phong.cpp:101-204) ---- */
/* The Phong driver adds, for every stereo observation (pose k, vertex j), an intensity residual
 * block over (pose, position, normal, material Phong parameters, texture, light) (:108-139) and a
 * normal residual block over (pose, normal) (:181-188), with UnitVectorPerturbation on the normal
 * (:191-193) and on a directional light (:201-204).  Landmark blocks become [position | normal]
 * (6-D local step); the shared blocks -- the light, and the Phong parameters [ka, ks, alpha] and the
 * texture kd of each material -- are a dense border of the reduced camera system, solved together
 * with the poses.
 *
 * ssba_add_normal_blocks    : replaces AddParameterBlock / SetParameterization of
 *                             map_vertices[j].normal(); normals (num*3) caller-owned, updated in
 *                             place by ssba_solve like the points; num must equal the point count.
 * ssba_add_material_blocks  : material()->phong_params().data() (M*3) and texture()->data() (M),
 *                             caller-owned and updated in place when free; material_of_point maps
 *                             each point to its material (copied).  M <= SSBA_MAX_MATERIALS.
 * ssba_add_light_block      : dataset.light_pos.data() (light_type 0) or light_dir.data() (1),
 *                             caller-owned, updated in place when free.
 * ssba_set_shared_block_constant : SetParameterBlockConstant on ALL blocks of one kind
 *                             (:166-172, :205-206 -- the driver's "DEBUG: hold ... constant" lines and
 *                             its --multistage stages): which = SSBA_BLOCK_LIGHT / _PHONG / _TEXTURE.
 *                             Shared blocks are free by default, as in the driver.
 * ssba_add_lighting_observations : intensity (num) and observed normal (num*3) of the i-th stereo
 *                             observation already added (same order), 1/sqrt(int_var) and the 3x3
 *                             normal stiffness; num must equal the stereo observation count at
 *                             ssba_finalize.
 * ssba_set_shared_block_bounds : SetParameterLowerBound / SetParameterUpperBound on entry `index`
 *                             of ALL Phong-parameter (index 0..2 = ka, ks, alpha) or texture (index
 *                             0 = kd) blocks, as the driver does for every material (:142-180: ka, ks,
 *                             kd in [0,1], alpha >= 1); +-INFINITY = no bound.  With a bound on a free
 *                             block the problem is constrained and the minimiser does what Ceres 1.x
 *                             does: Plus projects onto the box and every trust-region step goes through
 *                             a projected Armijo line search.  The search runs on the device inside the
 *                             iteration's hipGraph: the test of the full step, then up to SSBA_LS_ROUNDS
 *                             (environment; unset: two, and as many as a search took -- at most four -- once
 *                             one has run out of them) further evaluations; a search that needs more parks the
 *                             solver until the host -- inside ssba_solve / ssba_solve_step / ssba_solve_end --
 *                             has run it (same state machine, same bits: csrc/ssba_linesearch.h).
 * With lighting observations present ssba_evaluate / ssba_lm_step return 6-wide landmark blocks:
 * g_l (L*6), H_ll (L*36), delta_l (L*6, local coordinates).  ssba_set_huber_loss keeps its meaning
 * (the loss sits on the stereo residual blocks; lighting blocks take a NULL loss, :113,186).  Lighting
 * terms shard by landmarks like the stereo terms (ssba_set_distributed); with FREE shared blocks the
 * border sums are exchanged in the all-reduce mode only (DOGLEG and bounds included: the projected line
 * search adds its sums over the ranks at every evaluation and is driven by the host in lockstep) -- the
 * partitioned reduced solve with free shared blocks returns SSBA_ERR_UNSUPPORTED. */
#define SSBA_MAX_MATERIALS 15
enum { SSBA_BLOCK_LIGHT = 0, SSBA_BLOCK_PHONG = 1, SSBA_BLOCK_TEXTURE = 2 };
SSBA_API int ssba_add_normal_blocks(ssba_problem *p, double *normals, uint32_t num);
SSBA_API int ssba_add_material_blocks(ssba_problem *p, double *phong, double *texture, uint32_t num_materials,
                             const uint32_t *material_of_point, uint32_t num_points);
SSBA_API int ssba_add_light_block(ssba_problem *p, double *light, int light_type);
SSBA_API int ssba_set_shared_block_constant(ssba_problem *p, int which, int is_constant);
SSBA_API int ssba_set_shared_block_bounds(ssba_problem *p, int which, int index, double lower, double upper);
SSBA_API int ssba_add_lighting_observations(ssba_problem *p, const double *intensity,
                                   double intensity_stiffness, const double *normal_obs,
                                   const double normal_stiffness[9], uint64_t num);
/* test hook: the border of the last ssba_lm_step -- nb = 3 [light] + 3M [Phong] + M [texture] as
 * freed (that column order), S_pb (6*num_free_poses x nb), S_bb (nb x nb, damped), rhs_b (nb) of
 *   [S S_pb; S_pb^T S_bb] [delta_p; delta_b] = [rhs; rhs_b]
 * and the border step delta_b (nb).  Any output may be NULL. */
SSBA_API int ssba_border_system(ssba_problem *p, uint32_t *nb, double *S_pb, double *S_bb, double *rhs_b,
                       double *delta_b);

/* ---- Phong-lighting rows: batch evaluation on the device ------ */
/* replaces, for each of n residual-block instances, the evaluation Ceres performs on
 *   IntensityErrorPointLightAutomatic / IntensityErrorDirectionalLightAutomatic
 *     (include/ceres_slam/intensity_error_point_light.hpp:24-112,
 *      intensity_error_directional_light.hpp:24-113; blocks 12,3,3,3,1,3 -> 1 residual)
 *   NormalErrorAutomatic (include/ceres_slam/normal_error.hpp:22-54; blocks 12,3 -> 3)
 * with SE3Perturbation on the pose and UnitVectorPerturbation on the normal (and on the light
 * direction when light_type = 1) (perturbations.hpp:45-113; tests/dataset_ba_phong.cpp:102-204).
 * Per instance i: pose i (12), point i (3), normal i (3), phong i (ka,ks,alpha), texture i (kd),
 * observed colour i, observed normal i (3); shared: the light (position, light_type 0, or
 * direction, 1), the intensity stiffness 1/sqrt(int_var) and the 3x3 normal stiffness.
 * Outputs: r_int (n), J_int (n x 19 = [pose 6 | point 3 | normal 3 | phong 3 | texture 1 |
 * light 3], local coordinates), r_nrm (n x 3), J_nrm_pose (n x 18), J_nrm_n (n x 9).
 * This is the arithmetic of BASELINE config 3; its solver integration (6-D landmark blocks,
 * shared light/material border, bounds) is the next row. */
SSBA_API int ssba_phong_evaluate(int device, int light_type, uint64_t n, const double *poses,
                        const double *points, const double *normals, const double *phong,
                        const double *texture, const double light[3], const double *colour,
                        double stiffness, const double *normal_obs, const double normal_stiffness[9],
                        double *r_int, double *J_int, double *r_nrm, double *J_nrm_pose,
                        double *J_nrm_n);

/* replaces: problem.SetParameterBlockConstant(map_vertices[j].position().data()) on EVERY position block (stage 2 of
 * --multistage, tests/dataset_ba_phong.cpp:210-228; SetParameterBlockVariable afterwards = 0).  Lighting problems only:
 * the landmark block is then the normal alone.  Before ssba_finalize. */
SSBA_API int ssba_set_point_blocks_constant(ssba_problem *p, int is_constant);

/* ---- unary pose residual blocks (tests/dataset_vo_sun.cpp:80-124) ------- */
/* ssba_add_pose_prior replaces problem.AddResidualBlock(PoseErrorAutomatic::Create(T_k_0_ref, stiffness), loss,
 *   pose) (include/ceres_slam/pose_error.hpp:22-55; dataset_vo_sun.cpp:116-118): r = stiffness * log(T_ref T^-1) with
 *   the reference's log = [translation ; axis-angle] (se3group.hpp:337-342), 6 residuals; stiffness 6x6 row-major.
 * ssba_add_sun_observation replaces AddResidualBlock(SunSensorErrorAutomatic::Create(observed_dir_c, expected_dir_g,
 *   stiffness, az_err_thresh, zen_err_thresh), loss, pose) (sun_sensor_error.hpp:35-104; dataset_vo_sun.cpp:80-99):
 *   azimuth / zenith of R * expected_dir_g against the observation, wrap-around, outlier thresholds; 2 residuals;
 *   stiffness 2x2 row-major.  huber_a > 0 wraps the block in ceres::HuberLoss(huber_a) (:87-92), 0 = NULL loss.
 * Both before ssba_finalize; not on constant poses; not together with lighting terms or landmark sharding yet. */
SSBA_API int ssba_add_pose_prior(ssba_problem *p, uint32_t pose, const double T_ref[12], const double stiffness[36], double huber_a);
/* ssba_add_relative_pose replaces problem.AddResidualBlock(RelativePoseErrorAutomatic::Create(T_2_1_ref, stiffness), loss,
 *   pose1, pose2) (include/ceres_slam/relative_pose_error.hpp:22-57; tests/blowup_test.cpp:70-76):
 *   r = stiffness * log(T_2_1_ref * T_1 * T_2^-1), 6 residuals.  Problems with such blocks run the general-structure kernels. */
SSBA_API int ssba_add_relative_pose(ssba_problem *p, uint32_t pose1, uint32_t pose2, const double T_2_1_ref[12], const double stiffness[36],
                           double huber_a);
SSBA_API int ssba_add_sun_observation(ssba_problem *p, uint32_t pose, const double observed_dir_c[3],
                             const double expected_dir_g[3], const double stiffness[4], double az_err_thresh,
                             double zen_err_thresh, double huber_a);

/* ssba_pose_covariance replaces ceres::Covariance::Compute + GetCovarianceBlockInTangentSpace for one pose block
 * (tests/dataset_vo_sun.cpp:159-183): cov (6x6 row-major) = that pose's block of (J^T J)^-1 in local (tangent)
 * coordinates at the caller's current parameters, i.e. of the inverse of the undamped reduced camera system.
 * SSBA_ERR_NUMERICAL_FAILURE when the system is rank deficient (Ceres: "Covariance computation failed"). */
SSBA_API int ssba_pose_covariance(ssba_problem *p, uint32_t pose, double cov[36]);

/* ---- front end: the VO initial guess --------------------------------- */
/* ssba_frontend_ransac replaces, for `num_pairs` pairs of consecutive states at once,
 *   PointCloudAligner::compute_transformation_and_inliers (src/ceres_slam/point_cloud_aligner.cpp:64-136)
 * as compute_initial_guess calls it (src/ceres_slam/dataset_problem.cpp:246-249: 400 iterations, threshold 4;
 * dataset_problem_phong.cpp:346-348: threshold 9).  Pair p owns the matched points
 * [offset[p], offset[p+1]) of pts0 / pts1 (camera-frame points from StereoCamera::triangulate, 3 doubles each,
 * paired by position as the reference pairs them); samples = num_pairs * num_iters * 3 indices local to the
 * pair; thresh bounds the squared stereo reprojection error (:117-124).  Outputs: T (num_pairs * 12,
 * [t | R row-major] = T_1_0 of the hypothesis with the most inliers, the first one on ties, identity when no
 * hypothesis has an inlier), inlier flags (one byte per matched point; may be NULL), count (num_pairs; may be
 * NULL), device_time_s (kernel time; may be NULL).
 * ssba_ransac_samples (host) restates the reference's draw sequence (:69-91): std::mt19937 re-seeded with 42 in
 * every call and std::uniform_int_distribution<uint>(0, n-1), three distinct indices per iteration.  The
 * distribution is implementation-defined; libstdcxx_variant 1 = libstdc++ >= 11, 0 = libstdc++ <= 10. */
/* ssba_frontend_vo: the WHOLE DatasetProblem::compute_initial_guess(k1, k2) (src/ceres_slam/dataset_problem.cpp:179-270) on the
 * device for `num_states` consecutive states: reciprocal matches of every pair of consecutive states by landmark id
 * (:209-222; both lists in their state's order, paired by position), StereoCamera::triangulate of the matches
 * (:225-230), the num_iters-hypothesis 3-point RANSAC of ALL pairs (one lane per alignment, one workgroup per inlier count;
 * the draws are the reference's: std::mt19937(42) + uniform_int_distribution, libstdcxx_variant as for
 * ssba_ransac_samples), pose chaining poses[k] = T_k_km1 * poses[k-1] (:255) and the map initialisation from the inliers
 * of the first pair that sees a landmark (:259-269).  Observations of state k are the entries
 * [state_start[k], state_start[k+1]) of point_id / uvd (3 per observation).  poses: num_states x 12, poses[0] is the input;
 * map_points (num_points x 3) and initialized (num_points flags) are updated for newly initialised landmarks only.
 * match_count / inlier_count (num_states - 1; may be NULL).  SSBA_ERR_NUMERICAL_FAILURE: a pair has fewer than 3 matches. */
SSBA_API int ssba_frontend_vo(const ssba_camera *camera, int device, uint32_t num_states, const uint32_t *state_start,
                     const uint32_t *point_id, const double *uvd, uint32_t num_points, uint32_t num_iters, double thresh,
                     int libstdcxx_variant, double *poses, double *map_points, uint8_t *initialized, uint32_t *match_count,
                     uint32_t *inlier_count, double *device_time_s);
SSBA_API int ssba_ransac_samples(uint32_t n, uint32_t num_iters, int libstdcxx_variant, uint32_t *idx3);
SSBA_API int ssba_frontend_ransac(const ssba_camera *camera, int device, uint32_t num_pairs, const uint32_t *offset,
                         const double *pts0, const double *pts1, const uint32_t *samples, uint32_t num_iters,
                         double thresh, double *T, uint8_t *inlier, uint32_t *count, double *device_time_s);

SSBA_API const char *ssba_status_string(int status);
SSBA_API const char *ssba_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SSBA_H_ */
