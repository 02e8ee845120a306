// C ABI (include/ssba.h) of the MI355X stereo-BA back end: host-side problem graph,
// symbolic phase (landmark windows, reduced-system structure, BCR level plan), device
// mirrors and the enqueue-only trust-region loop.  No CPU fallback exists by design.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ssba.h"
#include "ssba_pool.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "ssba_launch.h"
#include "ssba_layout.h"
#include "ssba_wide_layout.h"
#include "ssba_linesearch.h"
#include "ssba_types.h"

#include <limits>

using namespace ssba;

static thread_local std::string g_last_error;
static void set_error(const std::string &s) { g_last_error = s; }

#define HIPCHECK(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
            return SSBA_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

constexpr int LS_ROUNDS_FIRST = 2, LS_ROUNDS_MOST = 4;      // device-side evaluations of the projected line search per iteration (ssba_solve_begin)
struct ssba_problem {
    ssba_camera cam{};
    int device = 0;
    // caller-owned parameter memory
    double *user_poses = nullptr, *user_points = nullptr;
    uint32_t P = 0, L = 0;
    // residual blocks, in the caller's order
    std::vector<uint32_t> obs_pose, obs_point;
    std::vector<double> obs_uvd;
    double S[9] = {0};
    bool have_S = false, per_obs_S = false, points_const = false;
    bool no_closure_border = false;     // a caller asked for something the closure border does not cover (DOGLEG, covariance): general path
    bool no_wide = false;               // ... or something the 144-row super-blocks of ssba_wide.hip do not cover (covariance): blocked Cholesky
    std::vector<double> obs_S;          // 9 per observation once two stereo blocks differ in stiffness
    std::vector<uint8_t> pose_const;
    double huber_a = 0.0;
    bool finalized = false;
    // config 3: lighting terms
    double *user_normals = nullptr;
    uint32_t num_normals = 0;
    double *user_phong = nullptr, *user_texture = nullptr, *user_light = nullptr;   // caller-owned shared blocks
    uint32_t M = 0;
    std::vector<uint32_t> ph_mat_of_point;
    int ph_light_type = 0;
    uint32_t shared_const = 0;             // bit 0 light, bit 1 Phong parameters, bit 2 textures
    std::vector<double> h_sh;              // packed [light 3 | phong 3M | texture M]
    double blo[4] = {-std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity(),
                     -std::numeric_limits<double>::infinity(), -std::numeric_limits<double>::infinity()};
    double bhi[4] = {std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(),
                     std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity()};
    // unary pose residual blocks (pose prior, sun sensor)
    std::vector<PoseFactor> pose_factors;      // (ssba_layout.h)
    std::vector<RelFactor> rel_factors;
    double *h_ls = nullptr;                // pinned: line-search scalars + state
    int num_line_search_steps = 0, num_line_searches_by_host = 0;      // evaluations / searches driven by the host (the device counts its own in the state)
    int ls_rounds_wanted = LS_ROUNDS_FIRST;    // search evaluations enqueued with every iteration: grows when a search ran out of them (finish_pending_search)
    std::vector<double> ph_intensity, ph_nobs;
    double ph_int_stiff = 0.0, ph_Sn[9] = {0};
    bool lighting() const { return !ph_intensity.empty(); }
    // host structure
    std::vector<int> pose_free, free_pose;
    std::vector<uint32_t> user_of_dev;   // Lpad -> user landmark or 0xFFFFFFFF
    ssba_stats stats{};
    // device
    Dev d{};
    Launcher launcher;
    hipStream_t own_stream = nullptr;
    std::vector<void *> allocs;
    bool defer_zero = false;                  // ssba_finalize: zero-fills are collected and done by ONE launch (flush_zero)
    std::vector<ZeroRange> zero_list;
    // uploads of ssba_finalize: copied into pinned chunks and sent asynchronously on the solver's stream, released after one
    // synchronisation at the end (a synchronous pageable hipMemcpy per array cost more than the layout work of small windows)
    std::vector<void *> stage_chunks;
    char *stage_cur = nullptr;
    size_t stage_left = 0;
    uint64_t dev_bytes = 0;
    State *h_state = nullptr;   // pinned
    double *h_stage = nullptr;  // pinned staging for parameter upload/download
    size_t h_stage_count = 0;
    ssba_exchange_fn xfn = nullptr;
    void *xctx = nullptr;
    int world_size = 1, rank = 0;
    State *ls_ring = nullptr;              // bounds: pinned copies of the state, one iteration behind (ssba_solve_step)
    hipEvent_t ls_ev[2] = {nullptr, nullptr};
    ncclComm_t rccl_comm = nullptr;        // native exchange (ssba_set_rccl): all-reduces enqueued on the solver's stream
    std::vector<uint32_t> sep_sb;          // partitioned solve: separator super-blocks (world_size + 1 entries)
    // one trust-region iteration captured as a hipGraph (single-GPU, un-instrumented path)
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    hipGraph_t graph2 = nullptr;            // GRAPH_ITERS iterations per replay (ssba_solve_step(n >= GRAPH_ITERS)): a tenth of the replay gaps
    hipGraphExec_t gexec2 = nullptr;
    // multi-rank: the kernel runs between the exchange points are captured as separate graphs
    hipGraph_t seg_graph[4] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t seg_exec[4] = {nullptr, nullptr, nullptr, nullptr};
    bool use_graph = true;
    int eager_iters = 0;                      // iterations enqueued kernel by kernel since the last graph was dropped (enqueue_iteration)
    // solve bookkeeping
    bool began = false;
    ssba_options opt{};
    int ignore_convergence = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::chrono::steady_clock::time_point t_begin;
    std::vector<double> log_cost, log_cost_change, log_gmax, log_step, log_rd, log_radius;
    std::vector<int32_t> log_ok;
    std::vector<double> log_stage;
    int log_capacity = 0;
};

template <class T>
static int dalloc(ssba_problem *p, T **out, size_t n) {
    void *ptr = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHECK(pool_malloc(&ptr, bytes));
    p->allocs.push_back(ptr);
    p->dev_bytes += bytes;
    *out = (T *)ptr;
    return SSBA_OK;
}
static int stage_alloc(ssba_problem *p, size_t bytes, char **out) {
    bytes = (bytes + 255) / 256 * 256;
    if (bytes > p->stage_left) {
        const size_t chunk = std::max<size_t>(bytes, p->stage_chunks.empty() ? (size_t)1 << 20 : (size_t)16 << 20);
        void *h = nullptr;
        HIPCHECK(pool_host_malloc(&h, chunk));
        p->stage_chunks.push_back(h);
        p->stage_cur = (char *)h;
        p->stage_left = chunk;
    }
    *out = p->stage_cur;
    p->stage_cur += bytes;
    p->stage_left -= bytes;
    return SSBA_OK;
}
// waits for the staged uploads and hands the chunks back to the pool
static void stage_release(ssba_problem *p) {
    if (p->stage_chunks.empty()) return;
    hipStreamSynchronize(p->launcher.stream);
    for (void *h : p->stage_chunks) pool_host_free(h);
    p->stage_chunks.clear();
    p->stage_cur = nullptr;
    p->stage_left = 0;
}
// host copy into the pinned staging chunk; large arrays (the ELL observation arrays of C4 are 96 MB each) on four threads
static void stage_copy(char *dst, const void *src, size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (bytes >= ((size_t)16 << 20) && hw > 1) ? (int)std::min<unsigned>(hw, 4u) : 1;
    if (nt == 1) { memcpy(dst, src, bytes); return; }
    const size_t piece = (bytes / (size_t)nt + 4095) / 4096 * 4096;
    std::vector<std::thread> th;
    size_t done = 0;
    try {
        for (int t = 1; t < nt; ++t) {
            const size_t o = (size_t)t * piece;
            if (o >= bytes) break;
            th.emplace_back([=] { memcpy(dst + o, (const char *)src + o, std::min(piece, bytes - o)); });
            done = std::min(bytes, o + piece);
        }
    } catch (const std::system_error &) {
    }
    memcpy(dst, src, std::min(piece, bytes));
    for (auto &x : th) x.join();
    const size_t covered = std::max(done, std::min(piece, bytes));
    if (covered < bytes) memcpy(dst + covered, (const char *)src + covered, bytes - covered);       // pieces whose thread could not be started
}
template <class T, class A>
static int dupload(ssba_problem *p, const T **out, const std::vector<T, A> &v) {
    T *ptr = nullptr;
    int rc = dalloc(p, &ptr, v.size());
    if (rc) return rc;
    if (!v.empty()) {
        const size_t bytes = v.size() * sizeof(T);
        char *h = nullptr;
        if ((rc = stage_alloc(p, bytes, &h))) return rc;
        stage_copy(h, v.data(), bytes);
        HIPCHECK(hipMemcpyAsync(ptr, h, bytes, hipMemcpyHostToDevice, p->launcher.stream));
    }
    *out = ptr;
    return SSBA_OK;
}
template <class T>
static int dzero(ssba_problem *p, T **out, size_t n) {
    int rc = dalloc(p, out, n);
    if (rc) return rc;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (p->defer_zero) {      // chunks of <= 1 MiB: one work-group each
        for (size_t o = 0; o < bytes; o += (size_t)1 << 20) p->zero_list.push_back({(char *)*out + o, (uint64_t)std::min<size_t>(bytes - o, (size_t)1 << 20)});
        return SSBA_OK;
    }
    HIPCHECK(hipMemsetAsync(*out, 0, bytes, p->launcher.stream));
    return SSBA_OK;
}
// The ~70 zero-fills of ssba_finalize in one launch (each hipMemsetAsync costs the host ~2 us: 0.15 ms per handle, which is
// what a two-state window notices); buffers come 256-byte aligned from the pool.
static int flush_zero(ssba_problem *p) {
    p->defer_zero = false;
    if (p->zero_list.empty()) return SSBA_OK;
    const size_t bytes = p->zero_list.size() * sizeof(ZeroRange);
    ZeroRange *dl = nullptr;
    int rc = dalloc(p, &dl, p->zero_list.size());
    if (rc) return rc;
    char *h = nullptr;
    if ((rc = stage_alloc(p, bytes, &h))) return rc;
    memcpy(h, p->zero_list.data(), bytes);
    HIPCHECK(hipMemcpyAsync(dl, h, bytes, hipMemcpyHostToDevice, p->launcher.stream));
    launch_zero_ranges(p->launcher.stream, dl, (int)p->zero_list.size());
    p->zero_list.clear();
    return SSBA_OK;
}

static void drop_graph(ssba_problem *p) {
    if (p->gexec) { hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    if (p->gexec2) { hipGraphExecDestroy(p->gexec2); p->gexec2 = nullptr; }
    if (p->graph2) { hipGraphDestroy(p->graph2); p->graph2 = nullptr; }
    if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
    for (int i = 0; i < 4; ++i) {
        if (p->seg_exec[i]) { hipGraphExecDestroy(p->seg_exec[i]); p->seg_exec[i] = nullptr; }
        if (p->seg_graph[i]) { hipGraphDestroy(p->seg_graph[i]); p->seg_graph[i] = nullptr; }
    }
    p->eager_iters = 0;
}

static void free_device(ssba_problem *p) {
    drop_graph(p);
    stage_release(p);
    if (!p->allocs.empty()) hipStreamSynchronize(p->launcher.stream);      // cached buffers go to the next handle
    for (void *a : p->allocs) pool_free(a);
    p->allocs.clear();
    p->dev_bytes = 0;
    if (p->h_state) { pool_host_free(p->h_state); p->h_state = nullptr; }
    if (p->h_stage) { pool_host_free(p->h_stage); p->h_stage = nullptr; }
    if (p->h_ls) { pool_host_free(p->h_ls); p->h_ls = nullptr; }
    p->finalized = false;
}

// A run of kernels between two exchange points: captured once as a hipGraph and replayed (the multi-rank
// path has ~60 launches per iteration; launched eagerly the host, not the GPU, would set the iteration time).
template <class F>
static int run_segment(ssba_problem *p, int idx, F body) {
    hipStream_t s = p->launcher.stream;
    if (!p->use_graph || p->launcher.timing || idx < 0) { body(); return SSBA_OK; }
    if (!p->seg_exec[idx]) {
        HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        body();
        hipError_t e = hipStreamEndCapture(s, &p->seg_graph[idx]);
        if (e != hipSuccess) { set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); return SSBA_ERR_HIP; }
        HIPCHECK(hipGraphInstantiate(&p->seg_exec[idx], p->seg_graph[idx], nullptr, nullptr, 0));
    }
    HIPCHECK(hipGraphLaunch(p->seg_exec[idx], s));
    return SSBA_OK;
}

// SSBA_API_TIMING=1: wall time per entry point, printed when the process exits (drivers that solve thousands of
// small windows are bound by set-up cost, not by the kernels)
struct ApiTimes {
    bool on = getenv("SSBA_API_TIMING") != nullptr;
    std::map<std::string, std::pair<double, long>> t;
    ~ApiTimes() {
        if (!on) return;
        for (auto &kv : t) fprintf(stderr, "[ssba] %-24s %8ld calls %10.3f ms total %8.3f ms each\n", kv.first.c_str(), kv.second.second,
                                   1e3 * kv.second.first, 1e3 * kv.second.first / std::max(1L, kv.second.second));
    }
};
static ApiTimes g_api_times;
struct ApiTimer {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit ApiTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~ApiTimer() {
        if (!g_api_times.on) return;
        auto &e = g_api_times.t[name];
        e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        e.second += 1;
    }
};

// phases of one entry point: PhaseTimer t; ... t.mark("finalize: windows"); -- rows of the SSBA_API_TIMING table
struct PhaseTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char *name) {
        if (!g_api_times.on) return;
        const auto t1 = std::chrono::steady_clock::now();
        auto &e = g_api_times.t[name];
        e.first += std::chrono::duration<double>(t1 - t0).count();
        e.second += 1;
        t0 = t1;
    }
};

extern "C" {

const char *ssba_status_string(int s) {
    switch (s) {
        case SSBA_OK: return "ok";
        case SSBA_ERR_INVALID_ARGUMENT: return "invalid argument";
        case SSBA_ERR_HIP: return "HIP runtime error";
        case SSBA_ERR_NUMERICAL_FAILURE: return "numerical failure";
        case SSBA_ERR_NOT_FINALIZED: return "problem not finalized";
        case SSBA_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case SSBA_ERR_UNSUPPORTED: return "problem structure not supported by this build";
        case SSBA_ERR_STATE: return "call sequence error";
        case SSBA_ERR_TIMEOUT: return "a collective-library set-up call did not return in time (see ssba_last_error)";
    }
    return "unknown status";
}
const char *ssba_last_error(void) { return g_last_error.c_str(); }

void ssba_default_options(ssba_options *o) {
    if (!o) return;
    o->max_num_iterations = 50;
    o->use_nonmonotonic_steps = 0;
    o->max_consecutive_nonmonotonic_steps = 5;
    o->jacobi_scaling = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->minimizer_progress_to_stdout = 0;
    o->num_threads = 1;
    o->num_linear_solver_threads = 1;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->trust_region_strategy_type = 0;
    o->dogleg_type = 0;
}

int ssba_create(const ssba_camera *camera, int device, ssba_problem **out) {
    ApiTimer api_timer("ssba_create");
    if (!camera || !out) return SSBA_ERR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("hipGetDeviceCount found no device");
        return SSBA_ERR_NO_DEVICE;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return SSBA_ERR_NO_DEVICE;
    }
    if (device >= n) return SSBA_ERR_INVALID_ARGUMENT;
    HIPCHECK(hipSetDevice(device));
    ssba_problem *p = new ssba_problem();
    p->cam = *camera;
    p->device = device;
    if (pool_stream_acquire(&p->own_stream) != hipSuccess) {
        delete p;
        return SSBA_ERR_HIP;
    }
    p->launcher.stream = p->own_stream;
    if (const char *e = getenv("SSBA_NO_GRAPH")) p->use_graph = !(e[0] == '1');
    hipEventCreate(&p->ev_begin);
    hipEventCreate(&p->ev_end);
    *out = p;
    return SSBA_OK;
}

static void rccl_release(ssba_problem *p);

int ssba_destroy(ssba_problem *p) {
    ApiTimer api_timer("ssba_destroy");
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    hipSetDevice(p->device);
    hipStreamSynchronize(p->launcher.stream);
    p->launcher.destroy();
    free_device(p);
    rccl_release(p);
    if (p->ev_begin) hipEventDestroy(p->ev_begin);
    if (p->ev_end) hipEventDestroy(p->ev_end);
    if (p->ls_ring) pool_host_free(p->ls_ring);
    for (hipEvent_t e : p->ls_ev) if (e) hipEventDestroy(e);
    if (p->own_stream) { hipStreamSynchronize(p->own_stream); pool_stream_release(p->own_stream); }
    delete p;
    return SSBA_OK;
}

int ssba_release_cached_memory(void) {
    pool_trim();
    return SSBA_OK;
}

int ssba_add_pose_blocks(ssba_problem *p, double *poses, uint32_t num) {
    if (!p || (!poses && num)) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    p->user_poses = poses;
    p->P = num;
    p->pose_const.assign(num, 0);
    return SSBA_OK;
}

int ssba_add_point_blocks(ssba_problem *p, double *points, uint32_t num) {
    if (!p || (!points && num)) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    p->user_points = points;
    p->L = num;
    return SSBA_OK;
}

int ssba_add_stereo_observations(ssba_problem *p, const uint32_t *pose_index, const uint32_t *point_index,
                                 const double *uvd, uint64_t num, const double stiffness[9]) {
    if (!p || !stiffness || (num && (!pose_index || !point_index || !uvd))) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    for (int c = 0; c < 9; ++c)
        if (!std::isfinite(stiffness[c])) return SSBA_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < num; ++i) {
        if (pose_index[i] >= p->P || point_index[i] >= p->L) {
            set_error("observation references a parameter block that was not added");
            return SSBA_ERR_INVALID_ARGUMENT;
        }
        for (int c = 0; c < 3; ++c)
            if (!std::isfinite(uvd[3 * i + c])) return SSBA_ERR_INVALID_ARGUMENT;
    }
    // one stiffness for all blocks (every reference driver but one) stays a kernel constant; a second, different
    // matrix (tests/dataset_vo_sun.cpp:56-65: one per map point) switches to one matrix per residual block
    if (p->have_S && !p->per_obs_S && memcmp(p->S, stiffness, sizeof p->S) != 0) {
        p->per_obs_S = true;
        p->obs_S.resize(9 * p->obs_pose.size());
        for (size_t i = 0; i < p->obs_pose.size(); ++i) memcpy(&p->obs_S[9 * i], p->S, sizeof p->S);
    }
    if (!p->have_S) memcpy(p->S, stiffness, sizeof p->S);
    p->have_S = true;
    if (p->per_obs_S)
        for (uint64_t i = 0; i < num; ++i) p->obs_S.insert(p->obs_S.end(), stiffness, stiffness + 9);
    p->obs_pose.insert(p->obs_pose.end(), pose_index, pose_index + num);
    p->obs_point.insert(p->obs_point.end(), point_index, point_index + num);
    p->obs_uvd.insert(p->obs_uvd.end(), uvd, uvd + 3 * num);
    return SSBA_OK;
}

int ssba_add_normal_blocks(ssba_problem *p, double *normals, uint32_t num) {
    if (!p || (!normals && num)) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    p->user_normals = normals;
    p->num_normals = num;
    return SSBA_OK;
}

int ssba_add_material_blocks(ssba_problem *p, double *phong, double *texture, uint32_t num_materials,
                             const uint32_t *material_of_point, uint32_t num_points) {
    if (!p || !phong || !texture || !material_of_point || num_materials == 0) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    if (num_materials > SSBA_MAX_MATERIALS) {
        set_error("more than SSBA_MAX_MATERIALS materials");
        return SSBA_ERR_UNSUPPORTED;
    }
    for (uint32_t j = 0; j < num_points; ++j)
        if (material_of_point[j] >= num_materials) return SSBA_ERR_INVALID_ARGUMENT;
    p->user_phong = phong;
    p->user_texture = texture;
    p->M = num_materials;
    p->ph_mat_of_point.assign(material_of_point, material_of_point + num_points);
    return SSBA_OK;
}

int ssba_add_light_block(ssba_problem *p, double *light, int light_type) {
    if (!p || !light || (light_type != 0 && light_type != 1)) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    p->user_light = light;
    p->ph_light_type = light_type;
    return SSBA_OK;
}

int ssba_set_shared_block_constant(ssba_problem *p, int which, int is_constant) {
    if (!p || which < 0 || which > 2) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;   // the border of the reduced system depends on it
    if (is_constant) p->shared_const |= (1u << which);
    else p->shared_const &= ~(1u << which);
    return SSBA_OK;
}

int ssba_add_pose_prior(ssba_problem *p, uint32_t pose, const double T_ref[12], const double stiffness[36], double huber_a) {
    if (!p || !T_ref || !stiffness || pose >= p->P) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    PoseFactor f{};
    f.pose = pose; f.type = 0; f.huber = huber_a > 0.0 ? huber_a : 0.0;
    memcpy(f.data, T_ref, 12 * sizeof(double));
    memcpy(f.S, stiffness, 36 * sizeof(double));
    p->pose_factors.push_back(f);
    return SSBA_OK;
}

int ssba_add_relative_pose(ssba_problem *p, uint32_t pose1, uint32_t pose2, const double T_2_1_ref[12], const double stiffness[36],
                           double huber_a) {
    if (!p || !T_2_1_ref || !stiffness || pose1 >= p->P || pose2 >= p->P || pose1 == pose2) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    RelFactor f{};
    f.pose1 = pose1; f.pose2 = pose2; f.huber = huber_a > 0.0 ? huber_a : 0.0;
    memcpy(f.T_ref, T_2_1_ref, sizeof f.T_ref);
    memcpy(f.S, stiffness, sizeof f.S);
    p->rel_factors.push_back(f);
    return SSBA_OK;
}

int ssba_add_sun_observation(ssba_problem *p, uint32_t pose, const double observed_dir_c[3], const double expected_dir_g[3],
                             const double stiffness[4], double az_err_thresh, double zen_err_thresh, double huber_a) {
    if (!p || !observed_dir_c || !expected_dir_g || !stiffness || pose >= p->P) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    PoseFactor f{};
    f.pose = pose; f.type = 1; f.huber = huber_a > 0.0 ? huber_a : 0.0;
    memcpy(f.data, observed_dir_c, 3 * sizeof(double));
    memcpy(f.data + 3, expected_dir_g, 3 * sizeof(double));
    f.data[6] = az_err_thresh; f.data[7] = zen_err_thresh;
    memcpy(f.S, stiffness, 4 * sizeof(double));
    p->pose_factors.push_back(f);
    return SSBA_OK;
}

int ssba_set_shared_block_bounds(ssba_problem *p, int which, int index, double lower, double upper) {
    if (!p || (which != SSBA_BLOCK_PHONG && which != SSBA_BLOCK_TEXTURE) || !(lower <= upper)) return SSBA_ERR_INVALID_ARGUMENT;
    if ((which == SSBA_BLOCK_PHONG && (index < 0 || index > 2)) || (which == SSBA_BLOCK_TEXTURE && index != 0))
        return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    const int i = which == SSBA_BLOCK_PHONG ? index : 3;
    p->blo[i] = lower;
    p->bhi[i] = upper;
    return SSBA_OK;
}

int ssba_add_lighting_observations(ssba_problem *p, const double *intensity, double intensity_stiffness,
                                   const double *normal_obs, const double normal_stiffness[9], uint64_t num) {
    if (!p || !normal_stiffness || (num && (!intensity || !normal_obs))) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    if (!p->ph_intensity.empty() &&
        (p->ph_int_stiff != intensity_stiffness || memcmp(p->ph_Sn, normal_stiffness, sizeof p->ph_Sn) != 0)) {
        set_error("all lighting residual blocks must share their stiffness (as the reference driver does)");
        return SSBA_ERR_UNSUPPORTED;
    }
    for (uint64_t i = 0; i < num; ++i) {
        if (!std::isfinite(intensity[i])) return SSBA_ERR_INVALID_ARGUMENT;
        for (int c = 0; c < 3; ++c)
            if (!std::isfinite(normal_obs[3 * i + c])) return SSBA_ERR_INVALID_ARGUMENT;
    }
    p->ph_int_stiff = intensity_stiffness;
    memcpy(p->ph_Sn, normal_stiffness, sizeof p->ph_Sn);
    p->ph_intensity.insert(p->ph_intensity.end(), intensity, intensity + num);
    p->ph_nobs.insert(p->ph_nobs.end(), normal_obs, normal_obs + 3 * num);
    return SSBA_OK;
}

int ssba_set_pose_constant(ssba_problem *p, uint32_t pose, int is_constant) {
    if (!p || pose >= p->P) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;   // the reduced-system structure depends on it
    p->pose_const[pose] = is_constant ? 1 : 0;
    return SSBA_OK;
}

int ssba_set_point_blocks_constant(ssba_problem *p, int is_constant) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    p->points_const = is_constant != 0;
    return SSBA_OK;
}

int ssba_set_huber_loss(ssba_problem *p, double a) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    p->huber_a = a > 0.0 ? a : 0.0;
    if (p->finalized) { p->d.huber_a = p->huber_a; drop_graph(p); }   // kernel arguments are baked into the graph
    return SSBA_OK;
}

int ssba_set_stream(ssba_problem *p, void *hip_stream) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->began) return SSBA_ERR_STATE;
    p->launcher.stream = hip_stream ? (hipStream_t)hip_stream : p->own_stream;
    return SSBA_OK;
}

// ---- native RCCL exchange ----------------------------------------------------------------------------------------
// The collectives of a sharded solve (the reduced pose system or, partitioned, the separator system,
// plus 16 scalars) are ncclAllReduce calls enqueued by the library itself on the solver's stream, between the captured
// kernel segments of an iteration -- no host language in the loop, so the C++ shim and the example drivers can shard
// too.  librccl.so is loaded on first use (dlopen): a single-GPU process never maps it.
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    bool ok = false;
    bool preloaded = false;      // the copy was mapped by the process already (PyTorch's), not loaded by this library
};
RcclApi *rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // a copy the process has mapped already (PyTorch ships and uses its own librccl.so) is taken before a second one
        // is loaded from the ROCm installation: two RCCL instances in one process share no state
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (a.lib) { a.preloaded = true; break; }
        }
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (a.lib) break;
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!a.lib) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
        a.GetVersion = (decltype(a.GetVersion))dlsym(a.lib, "ncclGetVersion");
        a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && a.GetErrorString;
        return a;
    }();
    return &api;
}
int rccl_exchange(void *ctx, void *buf, uint64_t count, int op) {
    ssba_problem *p = (ssba_problem *)ctx;
    RcclApi *a = rccl_api();
    const ncclResult_t r = a->AllReduce(buf, buf, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, p->rccl_comm, p->launcher.stream);
    if (r != ncclSuccess) { set_error(std::string("ncclAllReduce: ") + a->GetErrorString(r)); return 1; }
    return 0;
}
}  // namespace

static void rccl_release(ssba_problem *p) {
    if (!p->rccl_comm) return;
    hipStreamSynchronize(p->launcher.stream);
    rccl_api()->CommDestroy(p->rccl_comm);
    p->rccl_comm = nullptr;
    if (p->xfn == rccl_exchange) { p->xfn = nullptr; p->xctx = nullptr; }
}

// RCCL's set-up calls block without a time limit of their own, and a hang there (seen on this pool: DESIGN.md section 6) would
// hang the caller.  Both run on a helper thread and the caller waits at most SSBA_RCCL_TIMEOUT_S seconds (default 180); on
// expiry the thread is left behind (it cannot be cancelled inside the driver), the call returns SSBA_ERR_TIMEOUT and
// ssba_last_error() holds a record of where it sat: the phase, the librccl.so file that was mapped and its version, the
// world, HSA_ENABLE_IPC_MODE_LEGACY and the NCCL_DEBUG_FILE to read.  The caller can then fall back to another exchange.
extern "C++" {
namespace {
std::string rccl_describe() {
    RcclApi *a = rccl_api();
    std::string out = "librccl: ";
    Dl_info info;
    if (a->ok && dladdr((void *)a->CommInitRank, &info) && info.dli_fname) out += info.dli_fname; else out += "(not loaded)";
    int v = 0;
    if (a->ok && a->GetVersion && a->GetVersion(&v) == ncclSuccess) out += " version " + std::to_string(v);
    out += a->preloaded ? " (already mapped by the process)" : " (loaded by libssba.so)";
    const char *ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY"), *dbg = getenv("NCCL_DEBUG"), *dbgf = getenv("NCCL_DEBUG_FILE");
    out += std::string("; HSA_ENABLE_IPC_MODE_LEGACY=") + (ipc ? ipc : "(unset)") + "; NCCL_DEBUG=" + (dbg ? dbg : "(unset)") +
           "; NCCL_DEBUG_FILE=" + (dbgf ? dbgf : "(unset)");
    return out;
}
double rccl_timeout_s() {
    const char *e = getenv("SSBA_RCCL_TIMEOUT_S");
    const double t = e ? atof(e) : 180.0;
    return t > 0.0 ? t : 180.0;
}
struct RcclJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t result = ncclSuccess;
    ncclUniqueId id;
    ncclComm_t comm = nullptr;
    bool abandoned = false;                             // the caller gave up waiting: a communicator that completes late is the thread's to destroy
    ncclResult_t (*destroy)(ncclComm_t) = nullptr;
};
// runs fn(job) on a helper thread; false = not finished within the time limit.  The thread keeps the job alive; a call that
// completes after the caller gave up destroys its own communicator (nothing is leaked, nobody will ever use it).  A process
// whose RCCL set-up timed out still has a thread inside RCCL: it should finish its work and leave (bench.py falls back to
// torch.distributed for the run and exits normally; process teardown while that thread is mid-call is RCCL's to survive).
template <class F>
bool rccl_run_limited(std::shared_ptr<RcclJob> job, F fn, double *elapsed_s) {
    const auto t0 = std::chrono::steady_clock::now();
    std::thread([job, fn] {
        const ncclResult_t r = fn(job.get());
        ncclComm_t late = nullptr;
        {
            std::lock_guard<std::mutex> lock(job->mu);
            job->result = r;
            job->done = true;
            if (job->abandoned) { late = job->comm; job->comm = nullptr; }
            job->cv.notify_all();
        }
        if (late && job->destroy) job->destroy(late);
    }).detach();
    std::unique_lock<std::mutex> lock(job->mu);
    const bool ok = job->cv.wait_for(lock, std::chrono::duration<double>(rccl_timeout_s()), [&] { return job->done; });
    if (!ok) job->abandoned = true;
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return ok;
}
}  // namespace
}  // extern "C++"

int ssba_rccl_describe(char *buf, uint64_t size) {
    if (!buf || size == 0) return SSBA_ERR_INVALID_ARGUMENT;
    snprintf(buf, (size_t)size, "%s", rccl_describe().c_str());
    return SSBA_OK;
}

int ssba_rccl_unique_id(void *out, uint64_t size) {
    if (!out || size < sizeof(ncclUniqueId)) return SSBA_ERR_INVALID_ARGUMENT;
    RcclApi *a = rccl_api();
    if (!a->ok) { set_error("librccl.so could not be loaded"); return SSBA_ERR_UNSUPPORTED; }
    auto job = std::make_shared<RcclJob>();
    double el = 0.0;
    if (!rccl_run_limited(job, [a](RcclJob *j) { return a->GetUniqueId(&j->id); }, &el)) {
        char t[64]; snprintf(t, sizeof t, "%.0f", el);
        set_error(std::string("ncclGetUniqueId did not return within ") + t + " s (phase: bootstrap root set-up); " + rccl_describe());
        return SSBA_ERR_TIMEOUT;
    }
    if (job->result != ncclSuccess) { set_error(std::string("ncclGetUniqueId: ") + a->GetErrorString(job->result) + "; " + rccl_describe()); return SSBA_ERR_HIP; }
    memcpy(out, &job->id, sizeof job->id);
    return SSBA_OK;
}

int ssba_set_rccl(ssba_problem *p, const void *unique_id, uint64_t size) {
    if (!p || !unique_id || size < sizeof(ncclUniqueId)) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->began) return SSBA_ERR_STATE;
    RcclApi *a = rccl_api();
    if (!a->ok) { set_error("librccl.so could not be loaded"); return SSBA_ERR_UNSUPPORTED; }
    HIPCHECK(hipSetDevice(p->device));
    rccl_release(p);
    auto job = std::make_shared<RcclJob>();
    memcpy(&job->id, unique_id, sizeof job->id);
    job->destroy = a->CommDestroy;
    const int device = p->device, world = p->world_size, rank = p->rank;
    double el = 0.0;
    if (!rccl_run_limited(job, [a, device, world, rank](RcclJob *j) {
            if (hipSetDevice(device) != hipSuccess) return ncclUnhandledCudaError;       // the current device is per thread
            return a->CommInitRank(&j->comm, world, j->id, rank);
        }, &el)) {
        char t[160];
        snprintf(t, sizeof t, "ncclCommInitRank(world %d, rank %d, device %d) did not return within %.0f s; ", world, rank, device, el);
        set_error(std::string(t) + rccl_describe());
        return SSBA_ERR_TIMEOUT;
    }
    if (job->result != ncclSuccess) { set_error(std::string("ncclCommInitRank: ") + a->GetErrorString(job->result) + "; " + rccl_describe()); return SSBA_ERR_HIP; }
    p->rccl_comm = job->comm;
    p->xfn = rccl_exchange;
    p->xctx = p;
    return SSBA_OK;
}

int ssba_rccl_ranks(ssba_problem *p, int *count) {
    if (!p || !count) return SSBA_ERR_INVALID_ARGUMENT;
    *count = 0;
    if (!p->rccl_comm) return SSBA_OK;
    RcclApi *a = rccl_api();
    if (!a->CommCount || a->CommCount(p->rccl_comm, count) != ncclSuccess) { set_error("ncclCommCount failed"); return SSBA_ERR_HIP; }
    return SSBA_OK;
}

int ssba_set_exchange(ssba_problem *p, ssba_exchange_fn fn, void *ctx) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    p->xfn = fn;
    p->xctx = ctx;
    return SSBA_OK;
}

int ssba_set_distributed(ssba_problem *p, int world_size, int rank) {
    if (!p || world_size < 1 || rank < 0 || rank >= world_size) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    p->world_size = world_size;
    p->rank = rank;
    return SSBA_OK;
}

int ssba_set_partition(ssba_problem *p, const uint32_t *separator_superblocks, uint32_t num) {
    if (!p || (num && !separator_superblocks)) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_ERR_STATE;
    if (num == 0) { p->sep_sb.clear(); return SSBA_OK; }
    if ((int)num != p->world_size + 1 || num > (uint32_t)MAX_SEP) {
        set_error("ssba_set_partition: world_size + 1 separators expected (call ssba_set_distributed first)");
        return SSBA_ERR_INVALID_ARGUMENT;
    }
    for (uint32_t i = 0; i + 1 < num; ++i)
        if (separator_superblocks[i + 1] <= separator_superblocks[i]) return SSBA_ERR_INVALID_ARGUMENT;
    p->sep_sb.assign(separator_superblocks, separator_superblocks + num);
    return SSBA_OK;
}

int ssba_exchange_size(ssba_problem *p, uint64_t *count) {
    if (!p || !count) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->finalized) return SSBA_ERR_NOT_FINALIZED;
    *count = p->d.part ? p->d.sepv_count : p->d.xv_count + (p->d.wide ? p->launcher.wide.count : 0);
    return SSBA_OK;
}

// diagnostic: raw in-kernel stamp buffer (only filled by -DSSBA_STAMPS builds)
int ssba_debug_stamps(ssba_problem *p, unsigned long long *out, int n) {
    if (!p || !p->finalized || n > 8192) return SSBA_ERR_INVALID_ARGUMENT;
    HIPCHECK(hipStreamSynchronize(p->launcher.stream));
    HIPCHECK(hipMemcpy(out, p->d.dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return SSBA_OK;
}

int ssba_get_stats(ssba_problem *p, ssba_stats *st) {
    if (!p || !st) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->finalized) return SSBA_ERR_NOT_FINALIZED;
    *st = p->stats;
    st->device_bytes = p->dev_bytes;
    return SSBA_OK;
}

// ---------------------------------------------------------------------------------
// symbolic phase
// ---------------------------------------------------------------------------------
int ssba_finalize(ssba_problem *p) {
    ApiTimer api_timer("ssba_finalize");
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->finalized) return SSBA_OK;
    if (!p->have_S && !p->obs_pose.empty()) return SSBA_ERR_INVALID_ARGUMENT;
    HIPCHECK(hipSetDevice(p->device));
    const uint32_t P = p->P, L = p->L;
    const uint64_t N = p->obs_pose.size();
    const bool ph = p->lighting();
    if (ph) {
        if (p->ph_intensity.size() != N) {
            set_error("lighting observations must pair one-to-one with the stereo observations");
            return SSBA_ERR_INVALID_ARGUMENT;
        }
        if (p->num_normals != L || p->ph_mat_of_point.size() != L || !p->user_light) {
            set_error("lighting terms need a normal and a material for every point, and the light");
            return SSBA_ERR_INVALID_ARGUMENT;
        }
        // Landmark sharding: the lighting terms of a landmark live on its rank like the stereo terms.  With
        // FREE shared blocks the border sums over all landmarks (S_pb, S_bb, reduced border gradient) are exchanged next to
        // the reduced system (enqueue_front); that exists for the all-reduce mode, not for the partitioned reduced solve.
    }
    if (N >= (1ull << 28) || L >= (1u << 27)) {
        set_error("problem too large for the 32-bit observation references of this build");
        return SSBA_ERR_UNSUPPORTED;
    }

    PhaseTimer phase;
    // host phase (ssba_layout.cpp, plain C++): landmark order, windows / slots or the general layout, Schur items, gather lists
    Layout lay;
    {
        const LayoutInput in{P, L, p->obs_pose, p->obs_point, p->obs_uvd, p->pose_const, p->per_obs_S, p->obs_S, ph, p->M, p->ph_mat_of_point,
                             p->ph_intensity, p->ph_nobs, p->points_const, p->pose_factors, p->rel_factors, p->world_size, p->sep_sb.size() > 2,
                             p->no_closure_border, p->no_wide};
        std::string err;
        const int lrc = build_layout(in, lay, err, [&](const char *what) { phase.mark(what); });
        if (lrc) { set_error(err); return lrc; }
    }
    p->pose_free = std::move(lay.pose_free);
    p->free_pose = std::move(lay.free_pose);
    p->user_of_dev = std::move(lay.user_of_dev);
    const int nfree = lay.nfree, nchain = lay.nchain, nborder = lay.nborder;
    const bool dense = lay.dense, wide_sys = lay.wide_sys;
    const uint32_t Lact = lay.Lact, Lpad = lay.Lpad, n_groups = lay.n_groups, n_windows = lay.n_windows, n_slabs = lay.n_slabs, n_sblk = lay.n_sblk;
    const uint32_t bandwidth = lay.bandwidth;
    auto &pfs = lay.pfs;
    auto &win_pose = lay.win_pose; auto &lm_win = lay.lm_win; auto &lm_mask = lay.lm_mask; auto &lm_mat = lay.lm_mat;
    auto &pose_obs_start = lay.pose_obs_start; auto &pose_obs_ref = lay.pose_obs_ref; auto &pose_mat_start = lay.pose_mat_start;
    auto &ou = lay.ou; auto &ov = lay.ov; auto &od = lay.od; auto &oint = lay.oint; auto &onx = lay.onx; auto &ony = lay.ony; auto &onz = lay.onz;
    auto &slab_win = lay.slab_win; auto &slab_b = lay.slab_b; auto &slab_e = lay.slab_e;
    auto &sblk_a = lay.sblk_a; auto &sblk_b = lay.sblk_b; auto &sblk_start = lay.sblk_start; auto &sblk_contrib = lay.sblk_contrib;
    auto &prow_start = lay.prow_start; auto &prow_contrib = lay.prow_contrib;
    auto &cb_a = lay.cb_a; auto &cb_b = lay.cb_b; auto &cb_start = lay.cb_start; auto &cb_contrib = lay.cb_contrib;
    auto &dn_lm_start = lay.dn_lm_start; auto &dn_obs_pose = lay.dn_obs_pose; auto &dn_obs_lm = lay.dn_obs_lm;
    auto &dn_pose_start = lay.dn_pose_start; auto &dn_pose_obs = lay.dn_pose_obs; auto &dn_zpos = lay.dn_zpos;
    auto &dn_u = lay.dn_u; auto &dn_v = lay.dn_v; auto &dn_d = lay.dn_d; auto &dn_Sobs = lay.dn_Sobs; auto &dn_prec = lay.dn_prec;
    auto &dn_blk_a = lay.dn_blk_a; auto &dn_blk_b = lay.dn_blk_b; auto &dn_blk_start = lay.dn_blk_start;
    auto &dn_pair_a = lay.dn_pair_a; auto &dn_pair_b = lay.dn_pair_b; auto &dn_pose_mat_start = lay.dn_pose_mat_start; auto &dn_ztile = lay.dn_ztile;
    auto &dplan = lay.dplan; auto &wlay = lay.wlay;

    // ---- device mirrors ------------------------------------------------------------
    free_device(p);
    Dev &d = p->d;
    d = Dev{};
    d.fu = p->cam.fu; d.fv = p->cam.fv; d.cu = p->cam.cu; d.cv = p->cam.cv; d.b = p->cam.b;
    memcpy(d.S, p->S, sizeof d.S);
    d.huber_a = p->huber_a;
    d.P = (int)P; d.nfree = nfree; d.nchain = nchain;
    d.Nsb = std::max(1, (nfree + SBP - 1) / SBP);
    d.nf_pad = d.Nsb * SBP;
    d.Lpad = (int)Lpad; d.n_groups = (int)n_groups; d.n_windows = (int)n_windows;
    d.n_slabs = (int)n_slabs; d.n_sblk = (int)n_sblk;
    d.n_lm_blocks = (int)(Lpad / 256);
    d.n_pose_blocks = (int)((P + 255) / 256);
    if (d.n_pose_blocks < 1) d.n_pose_blocks = 1;
    d.n_obs = (uint32_t)N;
    int rc;
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    p->defer_zero = true;       // until flush_zero() below
    TRY(dzero(p, &d.poses, (size_t)P * 12)); TRY(dzero(p, &d.cand_poses, (size_t)P * 12));
    TRY(dzero(p, &d.best_poses, (size_t)P * 12)); TRY(dzero(p, &d.init_poses, (size_t)P * 12));
    TRY(dzero(p, &d.pts, (size_t)Lpad * 3)); TRY(dzero(p, &d.cand_pts, (size_t)Lpad * 3));
    TRY(dzero(p, &d.best_pts, (size_t)Lpad * 3)); TRY(dzero(p, &d.init_pts, (size_t)Lpad * 3));
    TRY(dupload(p, &d.pose_free, p->pose_free));
    TRY(dupload(p, &d.free_pose, p->free_pose));
    TRY(dupload(p, &d.ou, ou)); TRY(dupload(p, &d.ov, ov)); TRY(dupload(p, &d.od, od));
    TRY(dupload(p, &d.lm_mask, lm_mask)); TRY(dupload(p, &d.lm_win, lm_win));
    if (win_pose.empty()) win_pose.assign(TW, 0xFFFFFFFFu);
    TRY(dupload(p, &d.win_pose, win_pose));
    TRY(dupload(p, &d.pose_obs_start, pose_obs_start)); TRY(dupload(p, &d.pose_obs_ref, pose_obs_ref));
    TRY(dzero(p, &d.hll, (size_t)Lpad * (ph ? 21 : 6))); TRY(dzero(p, &d.gl, (size_t)Lpad * (ph ? 6 : 3)));
    TRY(dzero(p, &d.sl, (size_t)Lpad * (ph ? 6 : 3)));
    if (ph) {
        d.phong = 1;
        d.pos_const = p->points_const ? 1 : 0;
        d.light_type = p->ph_light_type;
        d.M = (int)p->M;
        static_assert(3 + 4 * SSBA_MAX_MATERIALS <= 64, "k_ph_border_update gives every entry of the shared blocks a lane of one wave");
        d.nsh = 3 + 4 * (int)p->M;
        d.b_light = d.b_phong = d.b_tex = -1;
        if (!(p->shared_const & 1u)) { d.b_light = d.nb; d.nb += 3; }
        if (!(p->shared_const & 2u)) { d.b_phong = d.nb; d.nb += 3 * d.M; }
        if (!(p->shared_const & 4u)) { d.b_tex = d.nb; d.nb += d.M; }
        if (d.nb > NBP) { set_error("border of free shared blocks wider than this build supports"); return SSBA_ERR_UNSUPPORTED; }
        // bounds only matter on free blocks; a constrained problem runs the projected line search
        for (int i = 0; i < 4; ++i) {
            d.blo[i] = p->blo[i]; d.bhi[i] = p->bhi[i];
            const bool is_free = i < 3 ? d.b_phong >= 0 : d.b_tex >= 0;
            if (is_free && (std::isfinite(p->blo[i]) || std::isfinite(p->bhi[i]))) d.constrained = 1;
        }
        d.int_stiff = p->ph_int_stiff;
        memcpy(d.Sn, p->ph_Sn, sizeof d.Sn);
        TRY(dzero(p, &d.nrm, (size_t)Lpad * 3)); TRY(dzero(p, &d.cand_nrm, (size_t)Lpad * 3));
        TRY(dzero(p, &d.best_nrm, (size_t)Lpad * 3)); TRY(dzero(p, &d.init_nrm, (size_t)Lpad * 3));
        TRY(dupload(p, &d.oi, oint)); TRY(dupload(p, &d.onx, onx)); TRY(dupload(p, &d.ony, ony)); TRY(dupload(p, &d.onz, onz));
        TRY(dupload(p, &d.lm_mat, lm_mat));
        TRY(dzero(p, &d.sh, (size_t)d.nsh)); TRY(dzero(p, &d.cand_sh, (size_t)d.nsh));
        TRY(dzero(p, &d.best_sh, (size_t)d.nsh)); TRY(dzero(p, &d.init_sh, (size_t)d.nsh));
        p->h_sh.assign((size_t)d.nsh, 0.0);
        TRY(dzero(p, &d.cinv, (size_t)Lpad * 21)); TRY(dzero(p, &d.cfac, (size_t)Lpad * 21)); TRY(dzero(p, &d.dlm, (size_t)Lpad * 6));
        TRY(dupload(p, &d.pose_mat_start, dense ? dn_pose_mat_start : pose_mat_start));
        if (d.nb) {
            TRY(dzero(p, &d.lmV, (size_t)Lpad * 42)); TRY(dzero(p, &d.lmH, (size_t)Lpad * 28)); TRY(dzero(p, &d.lmG, (size_t)Lpad * 7));
            const char *e = getenv("SSBA_BORDER_POSE_KERNEL");       // 1: the pose-by-pose kernel k_ph_border_poses (A/B, tests)
            if (!dense && !(e && e[0] == '1')) {
                TRY(dzero(p, &d.lmMV, (size_t)Lpad * 42));
                TRY(dzero(p, &d.Hpb, (size_t)std::max(1, (nfree + SBP - 1) / SBP) * SBP * 6 * NBP));
                TRY(dzero(p, &d.slabB, (size_t)n_slabs * 72 * NBP));
                TRY(dzero(p, &d.HpbL, (size_t)P * p->M * 18));
            }
            TRY(dzero(p, &d.part_b, (size_t)(Lpad / 256 + 1) * d.M * NBV));   // + one row of column sums
            TRY(dzero(p, &d.bsys, (size_t)BS_COUNT));
            d.n_gram = 64;
            TRY(dzero(p, &d.part_g, (size_t)d.n_gram * (NBP * NBP + NBP)));
        }
        TRY(dzero(p, &d.part_ls, (size_t)(Lpad / 256) * NLS));
        TRY(dzero(p, &d.ls_out, (size_t)NLS_OUT + NLS_MACH + NLS_X + NLS_X_RANKS));
    }
    TRY(dzero(p, &d.hpp, (size_t)P * 21)); TRY(dzero(p, &d.gp, (size_t)P * 6));
    TRY(dzero(p, &d.sp, (size_t)d.nf_pad * 6));
    TRY(dupload(p, &d.slab_win, slab_win)); TRY(dupload(p, &d.slab_lm_begin, slab_b));
    TRY(dupload(p, &d.slab_lm_end, slab_e));
    TRY(dzero(p, &d.slab, (size_t)n_slabs * SLAB_DOUBLES));
    TRY(dupload(p, &d.sblk_a, sblk_a)); TRY(dupload(p, &d.sblk_b, sblk_b));
    TRY(dupload(p, &d.sblk_start, sblk_start)); TRY(dupload(p, &d.sblk_contrib, sblk_contrib));
    TRY(dupload(p, &d.prow_start, prow_start)); TRY(dupload(p, &d.prow_contrib, prow_contrib));
    if (nborder) {      // closure border: 6 columns per border pose
        d.cb = 1; d.nb = 6 * nborder; d.n_cb = (int)cb_a.size();
        d.b_light = d.b_phong = d.b_tex = -1;
        TRY(dupload(p, &d.cb_a, cb_a)); TRY(dupload(p, &d.cb_b, cb_b));
        TRY(dupload(p, &d.cb_start, cb_start)); TRY(dupload(p, &d.cb_contrib, cb_contrib));
        TRY(dzero(p, &d.bsys, (size_t)BS_COUNT));
        d.n_gram = 64;
        TRY(dzero(p, &d.part_g, (size_t)d.n_gram * (NBP * NBP + NBP)));
    }
    phase.mark("finalize: 7 device mirrors");
    // exchange vector
    const uint64_t blk = (uint64_t)BD * BD;
    d.off_D = 0;
    d.off_L = d.off_D + (uint64_t)d.Nsb * blk;
    d.off_rhs = d.off_L + (uint64_t)d.Nsb * blk;
    d.off_gp = d.off_rhs + (uint64_t)d.Nsb * BD;
    d.off_hdiag = d.off_gp + (uint64_t)d.Nsb * BD;
    d.off_scal = d.off_hdiag + (uint64_t)d.Nsb * BD;
    d.xv_count = d.off_scal + NSCAL;
    TRY(dzero(p, &d.xv, d.xv_count));
    TRY(dzero(p, &d.x0, (size_t)d.nf_pad * 6));
    if (d.nb) {
        TRY(dzero(p, &d.Spb, (size_t)d.nf_pad * 6 * NBP));
        TRY(dzero(p, &d.Zb, (size_t)d.nf_pad * 6 * NBP));
    }
    TRY(dzero(p, &d.vp, (size_t)P * 6)); TRY(dzero(p, &d.vl, (size_t)Lpad * (ph ? 6 : 3))); TRY(dzero(p, &d.dl_gn, (size_t)Lpad * (ph ? 6 : 3)));
    // BCR level plan.  Plain: all super-blocks, odd blocks eliminated level by level down to one block.
    // Partitioned (ssba_set_partition): this rank's chain [sep[rank], sep[rank+1]]; an end shared with a neighbouring
    // rank is pinned.  Plain levels until <= PCR_MAX_BLOCKS blocks are left, parallel cyclic reduction with pinned ends
    // from there; the world_size - 1 shared blocks form the separator system (parallel cyclic reduction as well).
    auto upload_pos = [&](const std::vector<int> &pos, const int **out) -> int { return dupload(p, out, pos); };
    int pcr_max = PCR_MAX_BLOCKS;               // SSBA_PCR_MAX_BLOCKS lowers it (tests: plain levels below a parallel top on small problems)
    if (const char *e = getenv("SSBA_PCR_MAX_BLOCKS")) pcr_max = std::min(PCR_MAX_BLOCKS, std::max(2, atoi(e)));
    const bool part = p->sep_sb.size() > 2;     // one rank: nothing is shared, the plain plan applies
    if (part) {
        if (ph && d.nb) { set_error("landmark sharding with free shared lighting blocks is not available yet"); return SSBA_ERR_UNSUPPORTED; }
        if ((int)p->sep_sb.back() != d.Nsb - 1 || p->sep_sb.front() != 0) {
            set_error("ssba_set_partition: the separators must start at super-block 0 and end at the last one");
            return SSBA_ERR_INVALID_ARGUMENT;
        }
        d.part = 1; d.rank = p->rank; d.world = p->world_size; d.n_sep = p->world_size - 1;
        for (int i = 0; i < d.n_sep; ++i) d.sep_sb[i] = (int)p->sep_sb[i + 1];
        d.chain0 = (int)p->sep_sb[p->rank]; d.chain1 = (int)p->sep_sb[p->rank + 1];
        d.pin0 = p->rank > 0; d.pin1 = p->rank + 1 < p->world_size;
        // every observation of this rank must fall into its chain (the sharding has to be aligned to super-blocks)
        for (uint64_t i = 0; i < N; ++i) {
            const int f = p->pose_free[p->obs_pose[i]];
            if (f >= 0 && (f / SBP < d.chain0 || f / SBP > d.chain1)) {
                set_error("ssba_set_partition: an observation of this rank touches a pose outside its chain of super-blocks");
                return SSBA_ERR_INVALID_ARGUMENT;
            }
        }
    }
    {
        int n = part ? d.chain1 - d.chain0 + 1 : d.Nsb, lev = 0;
        const size_t c0 = part ? (size_t)d.chain0 : 0;
        d.lev[0].n = n;
        d.lev[0].D = d.xv + d.off_D + c0 * blk;
        d.lev[0].L = d.xv + d.off_L + c0 * blk;
        d.lev[0].r = d.xv + d.off_rhs + c0 * BD;
        std::vector<int> pos(n);
        for (int i = 0; i < n; ++i) pos[i] = i;
        for (;;) {
            if (d.nb) TRY(dzero(p, &d.lev[lev].B, (size_t)n * BD * NBP));
            TRY(dzero(p, &d.lev[lev].YU, (size_t)std::max(1, n / 2) * blk));
            TRY(upload_pos(pos, &d.lev[lev].pos));
            if (part ? n <= pcr_max : n == 1) break;
            d.lev[lev].pin = (part && d.pin1 && (n % 2 == 0)) ? 1 : 0;
            const int n2 = d.lev[lev].pin ? n / 2 + 1 : (n + 1) / 2;
            std::vector<int> pos2(n2);
            for (int m = 0; m < n2; ++m) pos2[m] = (d.lev[lev].pin && m == n2 - 1) ? pos[n - 1] : pos[2 * m];
            ++lev;
            if (lev >= MAX_LEVELS) return SSBA_ERR_UNSUPPORTED;
            d.lev[lev].n = n2;
            TRY(dzero(p, &d.lev[lev].D, (size_t)n2 * blk));
            TRY(dzero(p, &d.lev[lev].L, (size_t)n2 * blk));
            TRY(dzero(p, &d.lev[lev].r, (size_t)n2 * BD));
            n = n2;
            pos.swap(pos2);
        }
        d.n_levels = lev + 1;
    }
    // parallel cyclic reduction of the top of the plan: single-GPU solves without multi-right-hand-side sweeps, and the
    // chain of a partitioned solve
    d.pcr.level = -1;
    auto make_pcr = [&](PcrPlan &P, int level, int n, int keep, int pin0, int pin1) -> int {
        P.level = level; P.n = n; P.steps = 0; P.keep = keep; P.pin0 = pin0; P.pin1 = pin1;
        for (int s2 = 1; s2 < n; s2 <<= 1) ++P.steps;
        const size_t slots = keep ? (size_t)std::max(P.steps, 1) * n : (size_t)n;
        TRY(dzero(p, &P.Lbuf, (size_t)n * blk)); TRY(dzero(p, &P.LbufT, (size_t)n * blk));
        TRY(dzero(p, &P.YL, slots * blk)); TRY(dzero(p, &P.YU, slots * blk));
        TRY(dzero(p, &P.yr, (size_t)n * BD));
        if (pin1) TRY(dzero(p, &P.Ubuf, (size_t)n * blk));
        if (keep) {
            TRY(dzero(p, &P.Gs, slots * blk));
            TRY(dzero(p, &P.Bb, (size_t)n * BD * NBP)); TRY(dzero(p, &P.yB, (size_t)n * BD * NBP));
        }
        return SSBA_OK;
    };
    auto make_fused = [&](PcrFused &F, size_t n, bool pins) -> int {      // one launch per step (PcrFused): ping-pong buffers of the assembled blocks and the Gram products
        for (int q = 0; q < 2; ++q) {
            TRY(dzero(p, &F.Dpp[q], n * blk)); TRY(dzero(p, &F.rpp[q], n * BD));
            TRY(dzero(p, &F.GLL[q], n * blk)); TRY(dzero(p, &F.GUU[q], n * blk));
            TRY(dzero(p, &F.GUL[q], n * blk)); TRY(dzero(p, &F.GULT[q], n * blk));
            TRY(dzero(p, &F.gL[q], n * BD)); TRY(dzero(p, &F.gU[q], n * BD));
        }
        if (pins) { TRY(dzero(p, &F.Lkeep, n * blk)); TRY(dzero(p, &F.Ukeep, n * blk)); }
        F.on = 1;
        return SSBA_OK;
    };
    {
        const char *e = getenv("SSBA_NO_PCR");
        // with free shared blocks the border columns follow through the kept factors of every step (ssba_border.hip);
        // that variant covers plans that are parallel from level 0 on
        if (part) {
            TRY(make_pcr(d.pcr, d.n_levels - 1, d.lev[d.n_levels - 1].n, 0, d.pin0, d.pin1));
            if (!d.nb) TRY(make_fused(d.pcrf, (size_t)d.pcr.n, true));
        } else if ((!d.nb || d.lev[0].n <= pcr_max) && p->world_size == 1 && !dense && !(e && e[0] == '1')) {
            int k = 0;
            while (d.lev[k].n > pcr_max) ++k;
            TRY(make_pcr(d.pcr, k, d.lev[k].n, d.nb ? 1 : 0, 0, 0));
            if (!d.nb) TRY(make_fused(d.pcrf, (size_t)d.pcr.n, false));
        }
    }
    if (part) {
        const uint64_t ns = (uint64_t)d.n_sep;
        d.soff_D = 0;
        d.soff_L = d.soff_D + ns * blk;
        d.soff_rhs = d.soff_L + ns * blk;
        d.soff_gp = d.soff_rhs + ns * BD;
        d.soff_hdiag = d.soff_gp + ns * BD;
        d.soff_scal = d.soff_hdiag + ns * BD;
        d.sepv_count = d.soff_scal + NSCAL + (uint64_t)d.world;     // + one gradient-max slot per rank
        TRY(dzero(p, &d.sepv, d.sepv_count));
        TRY(dzero(p, &d.xsep, (size_t)ns * BD));
        d.slev[0].n = d.n_sep;
        d.slev[0].D = d.sepv + d.soff_D;
        d.slev[0].L = d.sepv + d.soff_L;
        d.slev[0].r = d.sepv + d.soff_rhs;
        std::vector<int> pos(d.n_sep);
        for (int i = 0; i < d.n_sep; ++i) pos[i] = i;
        TRY(upload_pos(pos, &d.slev[0].pos));
        TRY(make_pcr(d.spcr, 0, d.n_sep, 0, 0, 0));
        TRY(make_fused(d.spcrf, (size_t)ns, false));        // one launch per step of the separator solve
    }
    std::vector<uint32_t> dn_blk_rf_start, dn_blk_rf;
    if (!pfs.empty()) {
        std::vector<uint32_t> start(P + 1, 0);
        for (auto &f : pfs) start[f.pose + 1]++;
        for (uint32_t k = 0; k < P; ++k) start[k + 1] += start[k];
        std::vector<uint32_t> cur(start.begin(), start.end() - 1);
        const size_t F = pfs.size();
        std::vector<int> type(F);
        std::vector<double> data(F * 18), S(F * 36), hub(F);
        std::vector<uint32_t> pos(F);
        for (size_t i = 0; i < F; ++i) pos[i] = cur[pfs[i].pose]++;     // stable: keeps the caller's order inside a pose
        for (size_t i = 0; i < F; ++i) {
            auto &f = pfs[i];
            const uint32_t q = pos[i];
            type[q] = f.type; hub[q] = f.huber;
            memcpy(&data[18 * (size_t)q], f.data, sizeof f.data);
            if (f.type >= 2 && f.data[14] >= 0.0) data[18 * (size_t)q + 14] = (double)pos[(size_t)f.data[14]];
            memcpy(&S[36 * (size_t)q], f.S, sizeof f.S);
        }
        if (dense) {       // relative-pose blocks of every block of the reduced system: first-half entry, bit 31 = transposed
            std::map<std::pair<uint32_t, uint32_t>, std::vector<uint32_t>> of_block;
            for (size_t i = 0; i < F; ++i)
                if (pfs[i].type == 2 && pfs[i].data[14] >= 0.0) {
                    const int f1 = p->pose_free[pfs[i].pose], f2 = p->pose_free[(uint32_t)pfs[i].data[12]];
                    of_block[{(uint32_t)std::min(f1, f2), (uint32_t)std::max(f1, f2)}].push_back(pos[i] | (f1 > f2 ? 0x80000000u : 0u));
                }
            dn_blk_rf_start.assign(1, 0);
            for (size_t b = 0; b < dn_blk_a.size(); ++b) {
                auto it = of_block.find({dn_blk_a[b], dn_blk_b[b]});
                if (it != of_block.end()) dn_blk_rf.insert(dn_blk_rf.end(), it->second.begin(), it->second.end());
                dn_blk_rf_start.push_back((uint32_t)dn_blk_rf.size());
            }
        }
        d.n_pf = (int)F;
        d.pf_owner = (p->world_size <= 1 || p->rank == 0) ? 1 : 0;
        TRY(dupload(p, &d.pf_start, start)); TRY(dupload(p, &d.pf_type, type));
        TRY(dupload(p, &d.pf_data, data)); TRY(dupload(p, &d.pf_S, S)); TRY(dupload(p, &d.pf_huber, hub));
        TRY(dzero(p, &d.pf_cost, (size_t)P));
    }
    p->launcher.wide = WideSys{};
    if (dense && wide_sys) {
        d.dense = 1;
        d.n_dn = 6 * nfree;
        d.dn_pad = (d.n_dn + DN_BS - 1) / DN_BS * DN_BS;
        TRY(dupload(p, &d.dn_lm_start, dn_lm_start)); TRY(dupload(p, &d.dn_obs_pose, dn_obs_pose)); TRY(dupload(p, &d.dn_obs_lm, dn_obs_lm));
        TRY(dupload(p, &d.dn_u, dn_u)); TRY(dupload(p, &d.dn_v, dn_v)); TRY(dupload(p, &d.dn_d, dn_d));
        TRY(dupload(p, &d.dn_pose_start, dn_pose_start)); TRY(dupload(p, &d.dn_pose_obs, dn_pose_obs)); TRY(dupload(p, &d.dn_zpos, dn_zpos));
        if (!dn_prec.empty()) TRY(dupload(p, &d.dn_prec, dn_prec));
        WideSys &w = p->launcher.wide;
        const uint64_t wblk = (uint64_t)WBD * WBD;
        w.n = wlay.n; w.n_items = (int)wlay.n_items; w.n_blk = (int)wlay.blk_a.size();
        w.steps = 0;
        for (int s2 = 1; s2 < w.n; s2 <<= 1) ++w.steps;
        w.off_L = (uint64_t)w.n * wblk; w.off_rhs = 2 * (uint64_t)w.n * wblk; w.count = w.off_rhs + (uint64_t)w.n * WBD;
        TRY(dzero(p, &w.xw, w.count));
        TRY(dupload(p, &w.item_begin, wlay.item_begin)); TRY(dupload(p, &w.item_end, wlay.item_end)); TRY(dupload(p, &w.item_base, wlay.item_base));
        TRY(dupload(p, &w.slot_obs, wlay.slot_obs));
        TRY(dzero(p, &w.slab, (size_t)std::max<uint32_t>(wlay.n_items, 1) * WSLAB_DOUBLES));
        TRY(dupload(p, &w.blk_a, wlay.blk_a)); TRY(dupload(p, &w.blk_b, wlay.blk_b));
        TRY(dupload(p, &w.blk_start, wlay.blk_start)); TRY(dupload(p, &w.blk_contrib, wlay.blk_contrib));
        TRY(dupload(p, &w.prow_start, wlay.prow_start)); TRY(dupload(p, &w.prow_contrib, wlay.prow_contrib));
        TRY(dzero(p, &w.U, (size_t)w.n * wblk)); TRY(dzero(p, &w.YL, (size_t)w.n * wblk)); TRY(dzero(p, &w.YU, (size_t)w.n * wblk));
        TRY(dzero(p, &w.yr, (size_t)w.n * WBD));
        std::vector<WideSys> wv(1, w);
        TRY(dupload(p, &d.wide, wv));
    } else if (dense) {
        d.dense = 1;
        d.n_dn = 6 * nfree;
        d.dn_pad = (d.n_dn + DN_BS - 1) / DN_BS * DN_BS;
        d.dn_nblk = (int)dn_blk_a.size();
        TRY(dupload(p, &d.dn_blk_a, dn_blk_a)); TRY(dupload(p, &d.dn_blk_b, dn_blk_b)); TRY(dupload(p, &d.dn_blk_start, dn_blk_start));
        TRY(dupload(p, &d.dn_pair_a, dn_pair_a)); TRY(dupload(p, &d.dn_pair_b, dn_pair_b));
        if (!dn_blk_rf.empty()) { TRY(dupload(p, &d.dn_blk_rf_start, dn_blk_rf_start)); TRY(dupload(p, &d.dn_blk_rf, dn_blk_rf)); }
        TRY(dupload(p, &d.dn_rows, dplan.rows)); TRY(dupload(p, &d.dn_ti, dplan.ti)); TRY(dupload(p, &d.dn_tk, dplan.tk));
        TRY(dupload(p, &d.dn_cols, dplan.cols));
        TRY(dupload(p, &d.dn_row_start, dplan.row_start));
        TRY(dupload(p, &d.dn_ztile, dn_ztile));
        d.dn_nztile = (int)dn_ztile.size();
        p->launcher.dense = dplan;
        TRY(dupload(p, &d.dn_lm_start, dn_lm_start)); TRY(dupload(p, &d.dn_obs_pose, dn_obs_pose)); TRY(dupload(p, &d.dn_obs_lm, dn_obs_lm));
        TRY(dupload(p, &d.dn_u, dn_u)); TRY(dupload(p, &d.dn_v, dn_v)); TRY(dupload(p, &d.dn_d, dn_d));
        if (p->per_obs_S) TRY(dupload(p, &d.dn_Sobs, dn_Sobs));
        TRY(dupload(p, &d.dn_pose_start, dn_pose_start)); TRY(dupload(p, &d.dn_pose_obs, dn_pose_obs)); TRY(dupload(p, &d.dn_zpos, dn_zpos));
        if (!dn_prec.empty()) TRY(dupload(p, &d.dn_prec, dn_prec));
        TRY(dzero(p, &d.dn_Y, dn_obs_pose.size() * (ph ? 36 : 18)));
        d.dn_W = d.dn_Y;        // one factor Z = W M^T for both sides of a pair product (k_dn_wy, k_ph_dn_wy)
        TRY(dzero(p, &d.dn_Mg, (size_t)Lpad * (ph ? 6 : 3)));
        TRY(dzero(p, &d.dn_S, (size_t)(d.dn_pad + DN_BS) * std::max(d.dn_pad, DN_BS)));
    }
    TRY(dzero(p, &d.part_lin, (size_t)d.n_groups * 4));       // one entry per block of 256 landmarks, or per group of 64 (window layout)
    TRY(dzero(p, &d.part_eval, (size_t)d.n_groups * 4));
    TRY(dzero(p, &d.part_chk, (size_t)std::max<uint32_t>(d.nfree, 1) * 2));
    TRY(dzero(p, &d.part_pose, (size_t)(std::max(d.n_pose_blocks, d.Nsb) + 1) * NPP));   // + one entry for the border of shared blocks
    TRY(dzero(p, &d.part_dl, (size_t)(d.n_lm_blocks + d.n_pose_blocks + 1) * NDL));
    TRY(dzero(p, &d.scal2, (size_t)NSCAL));
    TRY(dzero(p, &d.scal_dl, (size_t)NSCAL));
    TRY(dzero(p, &d.gmax_l, (size_t)1));
    TRY(dzero(p, &d.st, (size_t)1));
    TRY(dzero(p, &d.dbg, (size_t)8192));
    HIPCHECK(pool_host_malloc((void **)&p->h_state, sizeof(State)));
    HIPCHECK(pool_host_malloc((void **)&p->h_ls, NLS_OUT * sizeof(double)));
    p->h_stage_count = std::max<size_t>((size_t)P * 12, (size_t)Lpad * 3);
    HIPCHECK(pool_host_malloc((void **)&p->h_stage, std::max<size_t>(p->h_stage_count, 1) * sizeof(double)));
    {
        // constant tables and kernel attributes are per device, not per handle: once per process and device
        static std::mutex mu;
        static std::map<int, int> done;       // bit 0 stereo / reduced solve, bit 1 lighting kernels
        std::lock_guard<std::mutex> lock(mu);
        int &flags = done[p->device];
        if (!(flags & 1)) {
            if (upload_pair_table(p->launcher.stream)) { set_error("pair table upload failed"); return SSBA_ERR_HIP; }
            if (upload_bcr_tables(p->launcher.stream)) { set_error("BCR tile table upload failed"); return SSBA_ERR_HIP; }
            if (configure_schur()) { set_error("hipFuncSetAttribute(k_schur_windows) failed"); return SSBA_ERR_HIP; }
            if (configure_kernels()) { set_error("hipFuncSetAttribute failed"); return SSBA_ERR_HIP; }
            if (configure_dense()) { set_error("hipFuncSetAttribute(k_dn_*_mf) failed"); return SSBA_ERR_HIP; }
            if (configure_wide()) { set_error("hipFuncSetAttribute(k_wd_*) failed"); return SSBA_ERR_HIP; }
            flags |= 1;
        }
        if (ph && !(flags & 2)) {
            if (upload_phong_tables(p->launcher.stream)) { set_error("pair table upload failed"); return SSBA_ERR_HIP; }
            if (configure_phong()) { set_error("hipFuncSetAttribute(k_ph_schur_windows) failed"); return SSBA_ERR_HIP; }
            if (configure_border()) { set_error("hipFuncSetAttribute(border kernels) failed"); return SSBA_ERR_HIP; }
            flags |= 2;
        }
    }
    TRY(flush_zero(p));
#undef TRY
    stage_release(p);
    phase.mark("finalize: 8 plans + tables");
    p->stats.num_poses = P; p->stats.num_free_poses = (uint32_t)nfree;
    p->stats.num_points = L; p->stats.num_active_points = Lact;
    p->stats.num_observations = N; p->stats.num_windows = n_windows;
    p->stats.num_superblocks = (uint32_t)d.Nsb; p->stats.num_reduced_blocks = n_sblk;
    p->stats.pose_bandwidth = bandwidth;
    p->stats.general_structure = dense ? 1u : nborder ? 2u : 0u;      // 2: windowed layout + closure border
    p->stats.pcr_blocks = d.pcr.level >= 0 ? (uint32_t)d.pcr.n : 0u;
    p->stats.pcr_fused = d.pcrf.on ? 1u : 0u;
    p->stats.wide_superblocks = 0;
    if (dense && wide_sys) {
        p->stats.num_reduced_blocks = (uint32_t)wlay.blk_a.size();
        p->stats.pose_bandwidth = wlay.bandwidth;
        p->stats.num_windows = wlay.n_items;
        p->stats.wide_superblocks = (uint32_t)wlay.n;
    } else if (dense) {      // the dense reduced system: its non-zero blocks and the real co-visibility span
        p->stats.num_reduced_blocks = (uint32_t)dn_blk_a.size();
        for (size_t i = 0; i < dn_blk_a.size(); ++i) p->stats.pose_bandwidth = std::max(p->stats.pose_bandwidth, dn_blk_b[i] - dn_blk_a[i]);
    }
    p->finalized = true;
    return SSBA_OK;
}

// ---------------------------------------------------------------------------------
// parameter transfer
// ---------------------------------------------------------------------------------
static int upload_params(ssba_problem *p) {
    Dev &d = p->d;
    hipStream_t s = p->launcher.stream;
    if (p->P) HIPCHECK(hipMemcpyAsync(d.poses, p->user_poses, (size_t)p->P * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    double *st = p->h_stage;
    const size_t Lp = (size_t)d.Lpad;
    for (size_t l = 0; l < Lp; ++l) {
        const uint32_t j = p->user_of_dev[l];
        for (int c = 0; c < 3; ++c) st[c * Lp + l] = (j == 0xFFFFFFFFu) ? (c == 2 ? 1.0 : 0.0) : p->user_points[3 * (size_t)j + c];
    }
    HIPCHECK(hipStreamSynchronize(s));   // poses copied from pageable memory
    HIPCHECK(hipMemcpyAsync(d.pts, st, Lp * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (d.phong) {
        for (size_t l = 0; l < Lp; ++l) {
            const uint32_t j = p->user_of_dev[l];
            for (int c = 0; c < 3; ++c) st[c * Lp + l] = (j == 0xFFFFFFFFu) ? (c == 2 ? 1.0 : 0.0) : p->user_normals[3 * (size_t)j + c];
        }
        HIPCHECK(hipMemcpyAsync(d.nrm, st, Lp * 3 * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHECK(hipStreamSynchronize(s));
        const size_t M = p->M;
        memcpy(p->h_sh.data(), p->user_light, 3 * sizeof(double));
        memcpy(p->h_sh.data() + 3, p->user_phong, 3 * M * sizeof(double));
        memcpy(p->h_sh.data() + 3 + 3 * M, p->user_texture, M * sizeof(double));
        if (d.constrained) {   // "x = Plus(x, 0)": the minimiser starts from the projection onto the feasible set
            if (d.b_phong >= 0)
                for (size_t c = 0; c < 3 * M; ++c) p->h_sh[3 + c] = std::min(std::max(p->h_sh[3 + c], d.blo[c % 3]), d.bhi[c % 3]);
            if (d.b_tex >= 0)
                for (size_t c = 0; c < M; ++c) p->h_sh[3 + 3 * M + c] = std::min(std::max(p->h_sh[3 + 3 * M + c], d.blo[3]), d.bhi[3]);
        }
        HIPCHECK(hipMemcpy(d.sh, p->h_sh.data(), (size_t)d.nsh * sizeof(double), hipMemcpyHostToDevice));
    }
    return SSBA_OK;
}

static int download_params(ssba_problem *p, const double *dev_poses, const double *dev_pts, const double *dev_nrm,
                           const double *dev_sh) {
    Dev &d = p->d;
    hipStream_t s = p->launcher.stream;
    const size_t Lp = (size_t)d.Lpad;
    HIPCHECK(hipMemcpyAsync(p->h_stage, dev_pts, Lp * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    for (size_t l = 0; l < Lp; ++l) {
        const uint32_t j = p->user_of_dev[l];
        if (j == 0xFFFFFFFFu) continue;
        for (int c = 0; c < 3; ++c) p->user_points[3 * (size_t)j + c] = p->h_stage[c * Lp + l];
    }
    if (d.phong) {
        HIPCHECK(hipMemcpyAsync(p->h_stage, dev_nrm, Lp * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        for (size_t l = 0; l < Lp; ++l) {
            const uint32_t j = p->user_of_dev[l];
            if (j == 0xFFFFFFFFu) continue;
            for (int c = 0; c < 3; ++c) p->user_normals[3 * (size_t)j + c] = p->h_stage[c * Lp + l];
        }
        if (d.nb) {   // free shared blocks are written back like every other parameter block
            const size_t M = p->M;
            HIPCHECK(hipMemcpy(p->h_sh.data(), dev_sh, (size_t)d.nsh * sizeof(double), hipMemcpyDeviceToHost));
            if (d.b_light >= 0) memcpy(p->user_light, p->h_sh.data(), 3 * sizeof(double));
            if (d.b_phong >= 0) memcpy(p->user_phong, p->h_sh.data() + 3, 3 * M * sizeof(double));
            if (d.b_tex >= 0) memcpy(p->user_texture, p->h_sh.data() + 3 + 3 * M, M * sizeof(double));
        }
    }
    if (p->P) {
        // constant / unobserved poses are never touched on the device, so a plain copy is exact
        HIPCHECK(hipMemcpyAsync(p->user_poses, dev_poses, (size_t)p->P * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
    }
    return SSBA_OK;
}

static Options to_device_options(const ssba_options *o, int ignore_convergence) {
    Options d{};
    d.max_num_iterations = o->max_num_iterations;
    d.max_nonmono = o->use_nonmonotonic_steps ? o->max_consecutive_nonmonotonic_steps : 0;
    d.jacobi_scaling = o->jacobi_scaling;
    d.max_invalid = o->max_num_consecutive_invalid_steps;
    d.ignore_convergence = ignore_convergence;
    d.strategy = o->trust_region_strategy_type == 1 ? 1 : 0;
    d.dogleg_type = o->dogleg_type == 1 ? 1 : 0;
    d.initial_radius = o->initial_trust_region_radius;
    d.max_radius = o->max_trust_region_radius;
    d.min_radius = o->min_trust_region_radius;
    d.min_relative_decrease = o->min_relative_decrease;
    d.min_lm_diag = o->min_lm_diagonal;
    d.max_lm_diag = o->max_lm_diagonal;
    d.function_tolerance = o->function_tolerance;
    d.gradient_tolerance = o->gradient_tolerance;
    d.parameter_tolerance = o->parameter_tolerance;
    return d;
}

static int ensure_log(ssba_problem *p, int capacity) {
    if (capacity <= p->log_capacity) return SSBA_OK;
    Dev &d = p->d;
    int rc;
    // kernels take Dev by value: a captured graph has the old log pointers and capacity baked in and would keep
    // writing there while ssba_solve_end reads the new (zero-filled) buffers
    drop_graph(p);
    // the seven columns of the iteration log in ONE buffer: one zero-fill here and one read-back in ssba_solve_end instead of
    // seven each (a blocking 50-byte hipMemcpy costs what a 50-kilobyte one does: ~15 us; two-state windows solve in 3 iterations)
    double *blk = nullptr;
    if ((rc = dzero(p, &blk, (size_t)7 * capacity))) return rc;
    d.log.cost = blk; d.log.cost_change = blk + capacity; d.log.gmax = blk + 2 * (size_t)capacity;
    d.log.step_norm = blk + 3 * (size_t)capacity; d.log.relative_decrease = blk + 4 * (size_t)capacity;
    d.log.radius = blk + 5 * (size_t)capacity;
    d.log.successful = reinterpret_cast<int32_t *>(blk + 6 * (size_t)capacity);
    d.log.capacity = capacity;
    p->log_capacity = capacity;
    return SSBA_OK;
}

// one trust-region iteration, enqueue only
static int enqueue_kernels(ssba_problem *p);
constexpr int LAZY_GRAPH_ITERS = 6;

static int finish_pending_search(ssba_problem *p);

static int enqueue_iteration(ssba_problem *p) {
    // The kernel sequence of an iteration is fixed (all control flow is on the device), so on
    // the plain single-GPU path it is captured once and replayed: ~45 launches become one.
    if (p->use_graph && !p->xfn && !p->launcher.timing) {
        hipStream_t s = p->launcher.stream;
        // A graph pays from the seventh iteration on: capture + instantiation (and its destruction with the handle, or at the
        // next call that changes a kernel argument) cost about what six iterations gain from the replay.  The thousands of
        // two-state windows the reference's scripts solve converge in fewer (examples/dataset_vo_sun_gpu: 3.6 -> 3.2 ms per
        // window); a C2 solve pays 6 x 0.2 ms once.
        if (!p->gexec && p->eager_iters < LAZY_GRAPH_ITERS) { ++p->eager_iters; return enqueue_kernels(p); }
        if (!p->gexec) {
            HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            int rc = enqueue_kernels(p);
            hipError_t e = hipStreamEndCapture(s, &p->graph);
            if (rc) return rc;
            if (e != hipSuccess) { set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); return SSBA_ERR_HIP; }
            HIPCHECK(hipGraphInstantiate(&p->gexec, p->graph, nullptr, nullptr, 0));
        }
        HIPCHECK(hipGraphLaunch(p->gexec, s));
        return SSBA_OK;
    }
    return enqueue_kernels(p);
}

static int enqueue_front(ssba_problem *p);
// bounds on one GPU: k_ph_ls_fast forms the sums of the evaluation kernel's partials (no exchange sits between them)
static bool ls_reduces_eval(const ssba_problem *p) { return p->d.constrained && !p->xfn; }
static double *ls_x(const Dev &d) { return d.ls_out + NLS_OUT + NLS_MACH; }
// single GPU, where neither the update / evaluation kernels nor the linearisation carry k_best's copy (bounds, free shared blocks,
// dogleg, lighting terms on the general layout): the commit launch of the same iteration does it
static bool best_in_commit(const ssba_problem *p) {
    const bool fuse_best = !p->d.constrained && !p->d.nb && p->opt.trust_region_strategy_type != 1;
    return !p->xfn && !p->d.part && !(fuse_best && launch_best_fusable(p->d));
}
// bounds with landmark sharding: the host looks at the state after EVERY iteration -- an iteration enqueued behind a parked one
// would run its exchanges over buffers whose kernels did nothing and sum the linearisation of a rejected step a second time
static bool lockstep(const ssba_problem *p) { return p->d.constrained && p->xfn; }

// GRAPH_ITERS iterations in one graph replay (the plain single-GPU path, ssba_solve_step): the sequence is fixed and
// every kernel turns into a no-op once the state says "terminated", so a batch is safe to enqueue blindly; it saves
// 7/8 of the ~9 us between replays.  (ssba_solve polls the state one replay behind and keeps single iterations: idle
// iterations after termination would cost what the batches save.)  Returns SSBA_ERR_STATE when this handle does not
// replay graphs (the caller falls back to single iterations).
constexpr int GRAPH_ITERS = 10;
static int enqueue_batch(ssba_problem *p) {
    if (!p->use_graph || p->xfn || p->launcher.timing) return SSBA_ERR_STATE;
    hipStream_t s = p->launcher.stream;
    if (!p->gexec2) {
        HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        int rc = 0;
        for (int i = 0; i < GRAPH_ITERS && !rc; ++i) rc = enqueue_kernels(p);
        hipError_t e = hipStreamEndCapture(s, &p->graph2);
        if (rc) return rc;
        if (e != hipSuccess) { set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); return SSBA_ERR_HIP; }
        HIPCHECK(hipGraphInstantiate(&p->gexec2, p->graph2, nullptr, nullptr, 0));
    }
    HIPCHECK(hipGraphLaunch(p->gexec2, s));
    return SSBA_OK;
}

// single GPU, LM, windowed stereo layout: the linearisation kernels commit the accepted step (ssba_kernels.hip:
// launch_linearize); SSBA_NO_FUSE_ALL=1 keeps the k_commit launch (A/B, tests)
static bool fuse_all_launches(const ssba_problem *p) {
    static const bool off = [] { const char *e = getenv("SSBA_NO_FUSE_ALL"); return e && e[0] == '1'; }();
    return !off && !p->xfn && !p->d.constrained && !p->d.nb && p->opt.trust_region_strategy_type != 1 && launch_can_fuse_all(p->d);
}

static int enqueue_kernels(ssba_problem *p) {
    int rc = enqueue_front(p);
    if (rc) return rc;
    // single GPU, LM: the decision kernel forms the evaluation sums itself (no exchange sits between them)
    const bool fuse = !p->xfn && !p->d.constrained && p->opt.trust_region_strategy_type != 1;
    const bool fuse_upd = fuse_all_launches(p) && !p->d.dense && bcr_updates_poses(p->d);       // the reduced solve updated the poses, one partial per block
    // bounds [trust_region_minimizer.cc DoLineSearch]: the Armijo test of the full step runs on the device; when it fails
    // (rare) the state is parked and the host drives the search at its next look at the state (finish_pending_search)
    if (p->d.constrained && p->xfn) {
        // landmark sharding: the landmark part of g . delta and max|delta| are sums / maxima over the ranks
        launch_ph_ls_pack(p->launcher, p->d, p->rank, p->world_size);
        if (p->xfn(p->xctx, ls_x(p->d), (uint64_t)(NLS_X + p->world_size), 0)) { set_error("exchange callback failed"); return SSBA_ERR_STATE; }
        launch_ph_ls_fast(p->launcher, p->d, false, p->world_size);
    } else if (p->d.constrained) launch_ph_ls_fast(p->launcher, p->d, ls_reduces_eval(p));
    if ((rc = run_segment(p, p->xfn ? 2 : -1, [&] { launch_decide_commit(p->launcher, p->d, fuse, fuse_all_launches(p), fuse_upd ? p->d.pcr.n : -1, best_in_commit(p)); }))) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return SSBA_ERR_HIP; }
    return SSBA_OK;
}

// Bounds-constrained problems [Ceres 1.x trust_region_minimizer.cc]: after the trust-region step the minimiser runs a
// projected Armijo line search along it.  Its first sample -- the full step -- is tested on the device (k_ph_ls_fast); only
// a rejected full step comes here: the stream is idle, the state parked (terminated, LS_PENDING), the iteration's front
// part done.  The device evaluates the line-search function, the host drives the search (one synchronisation per
// evaluation, ssba_linesearch.h), then the usual accept / reject logic judges the shortened step and the solver goes on.
static int finish_pending_search(ssba_problem *p) {
    ApiTimer api_timer("line search finished by the host");
    Dev &d = p->d;
    Launcher &L = p->launcher;
    int rc;
    auto fetch = [&]() -> int {
        HIPCHECK(hipMemcpyAsync(p->h_ls, d.ls_out, NLS_OUT * sizeof(double), hipMemcpyDeviceToHost, L.stream));
        HIPCHECK(hipMemcpyAsync(p->h_state, d.st, sizeof(State), hipMemcpyDeviceToHost, L.stream));
        HIPCHECK(hipStreamSynchronize(L.stream));
        return SSBA_OK;
    };
    // one evaluation of the line-search function; with landmark sharding the landmark sums of the ranks are added in the middle of it
    auto probe = [&](double alpha) -> int {
        if (!p->xfn) { launch_ph_ls_probe(L, d, alpha, 1); return SSBA_OK; }
        launch_ph_ls_probe(L, d, alpha, 1, 1, p->rank, p->world_size);
        if (p->xfn(p->xctx, ls_x(d), (uint64_t)(NLS_X + p->world_size), 0)) { set_error("exchange callback failed"); return SSBA_ERR_STATE; }
        launch_ph_ls_probe(L, d, -1.0, 0, 2, p->rank, p->world_size);
        return SSBA_OK;
    };
    launch_ls_resume(L, d);                 // the evaluation kernels test `terminated`
    if ((rc = probe(1.0))) return rc;       // the full step again, with phi'(1) this time (rounds of the device-side search may have moved the candidate)
    ++p->num_line_searches_by_host;
    if ((rc = fetch())) return rc;
    if (!p->h_state->terminated && p->h_ls[6] != 0.0) {
        Armijo a;
        a.begin(p->h_ls[7], p->h_ls[5], p->h_ls[4]);
        double at = 1.0;
        a.feed(p->h_ls[0], p->h_ls[1]);
        ++p->num_line_search_steps;
        while (!a.done) {
            at = a.current.x;
            if ((rc = probe(at))) return rc;
            if ((rc = fetch())) return rc;
            a.feed(p->h_ls[0], p->h_ls[1]);
            ++p->num_line_search_steps;
        }
        const double want = a.success ? a.optimal_step : 1.0;    // a failed search leaves delta alone
        // the device-side rounds ran out on a search that does finish: enqueue as many as it took from now on
        if (a.success && !p->xfn && !getenv("SSBA_LS_ROUNDS") && a.num_feeds > d.ls_rounds && d.ls_rounds < LS_ROUNDS_MOST) {
            p->ls_rounds_wanted = d.ls_rounds = std::min(a.num_feeds, LS_ROUNDS_MOST);
            drop_graph(p);
        }
        if (want != at && (rc = probe(want))) return rc;
        if (want != 1.0 || at != 1.0) launch_ph_ls_accept(L, d);
    }
    launch_decide_commit(L, d, false, false, -1, best_in_commit(p));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return SSBA_ERR_HIP; }
    return SSBA_OK;
}
// the state as the host last saw it says "a line search waits for the host"
static bool search_pending(const ssba_problem *p) { return p->h_state->terminated && p->h_state->termination_type == LS_PENDING; }

static int enqueue_front(ssba_problem *p) {
    Dev &d = p->d;
    Launcher &L = p->launcher;
    int rc;
    const bool multi = p->xfn != nullptr && !d.constrained;     // segments only matter (and are only valid) between exchanges
    auto X = [&](void *buf, uint64_t n, int op) -> int {
        if (p->xfn(p->xctx, buf, n, op)) { set_error("exchange callback failed"); return SSBA_ERR_STATE; }
        return SSBA_OK;
    };
    if (d.part) {
        // partitioned reduced solve: eliminate this rank's chain interior, sum the chain ends (the separator
        // system: ~1 MB instead of the whole reduced system) over the ranks, solve it everywhere, back-substitute
        if (!p->xfn) { set_error("a partitioned problem needs an exchange callback"); return SSBA_ERR_STATE; }
        const bool pbest = !d.constrained && !d.nb && p->opt.trust_region_strategy_type != 1;      // see fuse_best below
        if ((rc = run_segment(p, 0, [&] { launch_linearize(L, d, false, false, !d.phong); launch_schur(L, d); launch_finish_local(L, d); launch_bcr(L, d); launch_sep_pack(L, d); }))) return rc;
        if ((rc = X(d.sepv, d.sepv_count, 0))) return rc;      // sums; the landmark gradient maximum travels in per-rank slots
        if ((rc = run_segment(p, 1, [&] { launch_sep_finish_check(L, d, pbest); launch_bcr_separators(L, d); launch_update_eval(L, d, false, pbest); }))) return rc;
        if ((rc = X(d.scal2, NSCAL, 0))) return rc;
        return SSBA_OK;
    }
    // single GPU: the small launches between the big kernels are folded into their neighbours (ssba_kernels.hip, k_check)
    const bool fuse_ctrl = !p->xfn && !d.constrained && !d.nb;
    const bool fuse_best = !d.constrained && !d.nb && p->opt.trust_region_strategy_type != 1;      // with an exchange too: none sits between k_check and the update
    const bool fuse_all = fuse_all_launches(p);
    // k_check's work rides in the Schur launch (ssba_kernels.hip: k_schur_windows); SSBA_CHECK_LAUNCH=1 keeps the launch (A/B, tests)
    const char *cl_env = getenv("SSBA_CHECK_LAUNCH");       // (read at every capture: tests switch it between handles)
    const bool check_launch = cl_env && cl_env[0] == '1';
    const bool check_in_schur = fuse_ctrl && p->opt.trust_region_strategy_type != 1 && (fuse_all || d.phong) && !check_launch;
    if ((rc = run_segment(p, multi ? 0 : -1, [&] { launch_linearize(L, d, fuse_ctrl, fuse_all); if (d.dense) launch_dense_schur(L, d, fuse_ctrl && launch_ctrl_fusable(d)); else { L.spb_in_place_ok = !p->xfn; launch_schur(L, d, fuse_ctrl, check_in_schur); L.spb_in_place_ok = false; } }))) return rc;
    if (p->xfn) {
        if ((rc = X(d.xv, d.xv_count, 0))) return rc;
        if (d.wide && (rc = X(L.wide.xw, L.wide.count, 0))) return rc;      // long tracks: the 144-row super-blocks [D | L | rhs]
        if ((rc = X(d.gmax_l, 1, 1))) return rc;
        if (d.nb) {     // free shared blocks: every rank holds the border sums of ITS landmarks
            if ((rc = X(d.Spb, (uint64_t)d.nf_pad * 6 * NBP, 0))) return rc;
            if ((rc = X(d.bsys, (uint64_t)BS_S, 0))) return rc;       // S_bb | reduced gradient | g_b | diag H_bb
        }
    }
    if ((rc = run_segment(p, multi ? 1 : -1, [&] {
            if (p->xfn && d.nb) launch_border_scale(L, d);      // Jacobi scale of the border from the SUMMED diagonal
            launch_finish_check(L, d, fuse_ctrl, fuse_best, check_in_schur, best_in_commit(p));
            const bool fuse_upd = fuse_all && !d.dense && bcr_updates_poses(d);     // the last step of the reduced solve updates the poses
            if (d.dense) launch_dense_solve(L, d);      // incl. the rows of the free shared blocks
            else { launch_bcr(L, d, true, fuse_upd); if (d.nb) launch_border_solve(L, d); }
            // DOGLEG with landmark sharding: the six sums of the dogleg model are summed over the ranks between the two halves
            // (bounds on one GPU: the evaluation sums are formed by the Armijo test's launch -- launch_ph_ls_fast in enqueue_kernels)
            if (p->opt.trust_region_strategy_type == 1) launch_dogleg_eval(L, d, p->xfn ? 1 : 0, p->rank == 0 ? 1 : 0, ls_reduces_eval(p));
            else launch_update_eval(L, d, !p->xfn, fuse_best, fuse_upd);
        }))) return rc;
    if (p->xfn && p->opt.trust_region_strategy_type == 1) {
        if ((rc = X(d.scal_dl, NSCAL, 0))) return rc;
        if ((rc = run_segment(p, multi ? 3 : -1, [&] { launch_dogleg_eval(L, d, 2); }))) return rc;
    }
    if (p->xfn && (rc = X(d.scal2, NSCAL, 0))) return rc;
    return SSBA_OK;
}

static int reset_solver(ssba_problem *p) {
    Dev &d = p->d;
    hipStream_t s = p->launcher.stream;
    HIPCHECK(hipMemcpyAsync(d.poses, d.init_poses, (size_t)d.P * 12 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(d.pts, d.init_pts, (size_t)d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(d.best_poses, d.init_poses, (size_t)d.P * 12 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(d.best_pts, d.init_pts, (size_t)d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(d.cand_poses, d.init_poses, (size_t)d.P * 12 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(d.cand_pts, d.init_pts, (size_t)d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (d.phong) {
        const size_t nb = (size_t)d.Lpad * 3 * sizeof(double);
        HIPCHECK(hipMemcpyAsync(d.nrm, d.init_nrm, nb, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(d.best_nrm, d.init_nrm, nb, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(d.cand_nrm, d.init_nrm, nb, hipMemcpyDeviceToDevice, s));
        const size_t ns = (size_t)d.nsh * sizeof(double);
        HIPCHECK(hipMemcpyAsync(d.sh, d.init_sh, ns, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(d.best_sh, d.init_sh, ns, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(d.cand_sh, d.init_sh, ns, hipMemcpyDeviceToDevice, s));
    }
    launch_reset(p->launcher, d, to_device_options(&p->opt, p->ignore_convergence));
    return SSBA_OK;
}

// The closure border (loop closures as a dense border of the block-tridiagonal system, LM only) does not cover everything
// the general-structure path does.  A caller that asks for one of those things on a handle that was finalized with a
// border gets the other layout: ssba_finalize runs again from the host-side problem graph the handle still holds.
// The same hand-over for the 144-row super-blocks of long tracks (ssba_wide.hip: no covariance sweep): the blocked Cholesky has one.
static int refinalize_without_wide(ssba_problem *p, const char *what) {
    if (p->began) {
        set_error(std::string(what) + ": not available on this handle once a solve has begun (long tracks run on 144-row super-blocks; ask "
                  "before the first solve, or set SSBA_NO_WIDE=1)");
        return SSBA_ERR_STATE;
    }
    const double nf = (double)p->free_pose.size();
    const double dn_gb = (6.0 * nf + 2 * DN_BS) * (6.0 * nf + DN_BS) * 8.0 / 1e9;
    const char *mg = getenv("SSBA_DENSE_MAX_GB");
    const double dn_cap = mg ? atof(mg) : 160.0;
    if (p->world_size > 1 || dn_gb > dn_cap || (6 * (long)p->free_pose.size() + DN_BS) / DN_BS >= 65535) {
        set_error(std::string(what) + ": needs the blocked Cholesky of the general path (single GPU, reduced system within SSBA_DENSE_MAX_GB)");
        return SSBA_ERR_UNSUPPORTED;
    }
    p->no_wide = true;
    p->finalized = false;
    const int rc = ssba_finalize(p);
    if (rc) { p->no_wide = false; p->finalized = false; }
    return rc;
}
static int refinalize_without_closure_border(ssba_problem *p, const char *what) {
    if (p->began) {
        set_error(std::string(what) + ": not available on the closure border of this handle once a solve has begun (the loop closure was "
                  "folded into a border of the chain at ssba_finalize; ask before the first solve, or set SSBA_NO_CLOSURE_BORDER=1)");
        return SSBA_ERR_STATE;
    }
    // the general layout must fit BEFORE anything of the working layout is torn down (same test as ssba_finalize)
    const double nf = (double)p->free_pose.size();
    const double dn_gb = (6.0 * nf + 2 * DN_BS) * (6.0 * nf + DN_BS) * 8.0 / 1e9;
    const char *mg = getenv("SSBA_DENSE_MAX_GB");
    const double dn_cap = mg ? atof(mg) : 160.0;
    if (dn_gb > dn_cap || (6 * (long)p->free_pose.size() + DN_BS) / DN_BS >= 65535) {
        char buf[320];
        snprintf(buf, sizeof buf, "%s: not available on the closure border, and the general layout it would need holds a %.1f GB reduced "
                 "system (SSBA_DENSE_MAX_GB = %.0f): the handle keeps its closure-border layout (LM solves still work)", what, dn_gb, dn_cap);
        set_error(buf);
        return SSBA_ERR_UNSUPPORTED;
    }
    p->no_closure_border = true;
    p->finalized = false;
    const int rc = ssba_finalize(p);
    if (rc) {       // (an allocation failed after the old layout was freed): a later ssba_finalize builds the border layout again
        p->no_closure_border = false;
        p->finalized = false;
    }
    return rc;
}

int ssba_solve_begin(ssba_problem *p, const ssba_options *o, int ignore_convergence) {
    ApiTimer api_timer("ssba_solve_begin");
    if (!p || !o) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->finalized) return SSBA_ERR_NOT_FINALIZED;
    if (o->max_num_iterations < 0 || !(o->initial_trust_region_radius > 0.0)) return SSBA_ERR_INVALID_ARGUMENT;
    if (o->trust_region_strategy_type != 0 && o->trust_region_strategy_type != 1) return SSBA_ERR_INVALID_ARGUMENT;
    if (p->d.part && !p->xfn) { set_error("a partitioned problem needs an exchange callback"); return SSBA_ERR_STATE; }
    if (o->trust_region_strategy_type == 1 && p->xfn && p->d.part) {
        set_error("DOGLEG with landmark sharding runs on the all-reduce of the reduced system (not with ssba_set_partition: the pose "
                  "sums of its model need every rank's interior poses)");
        return SSBA_ERR_UNSUPPORTED;
    }
    if (o->dogleg_type != 0 && o->dogleg_type != 1) return SSBA_ERR_INVALID_ARGUMENT;
    if (o->trust_region_strategy_type == 1 && p->d.cb) {
        // DOGLEG is not implemented on the closure border: the symbolic phase runs again and puts the problem on the
        // general-structure path, which has it (once per handle; the caller's blocks and pointers are unchanged)
        int rc = refinalize_without_closure_border(p, "DOGLEG");
        if (rc) return rc;
    }
    if (p->d.phong && p->xfn && p->d.nb && (p->d.dense || (p->d.constrained && p->world_size > NLS_X_RANKS))) {
        set_error("lighting terms: landmark sharding with free shared blocks is not available on the general layout (nor with bounds on more than 64 ranks)");
        return SSBA_ERR_UNSUPPORTED;
    }
    if (p->opt.trust_region_strategy_type != o->trust_region_strategy_type)
        drop_graph(p);   // other kernel sequence: every captured graph (single iteration, batch, segments) goes
    if (p->d.constrained) {
        // bounds: evaluations of the projected line search enqueued with every iteration (device-side search); a search
        // that needs more is finished by the host.  SSBA_LS_ROUNDS in the environment of a solve (0: host only -- A/B, tests)
        // LS_ROUNDS_FIRST evaluations to begin with (a round that is not needed still costs its three launches: ~9 us per
        // iteration at C3); a handle whose search once needed more is given what that search took (up to LS_ROUNDS_MOST) from
        // then on -- one search finished by the host is the price of finding out
        int rounds = p->ls_rounds_wanted;
        if (const char *e = getenv("SSBA_LS_ROUNDS")) rounds = std::min(20, std::max(0, atoi(e)));
        if (p->xfn) rounds = 0;     // landmark sharding: every evaluation has an exchange in it, the host drives the search (finish_pending_search)
        if (rounds != p->d.ls_rounds) { p->d.ls_rounds = rounds; drop_graph(p); }
    }
    HIPCHECK(hipSetDevice(p->device));
    p->opt = *o;
    p->ignore_convergence = ignore_convergence;
    p->d.huber_a = p->huber_a;
    int cap = o->max_num_iterations + 4;
    if (cap > (1 << 20)) cap = 1 << 20;
    int rc = ensure_log(p, cap);
    if (rc) return rc;
    p->t_begin = std::chrono::steady_clock::now();
    p->num_line_search_steps = p->num_line_searches_by_host = 0;
    rc = upload_params(p);
    if (rc) return rc;
    hipStream_t s = p->launcher.stream;
    HIPCHECK(hipMemcpyAsync(p->d.init_poses, p->d.poses, (size_t)p->d.P * 12 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(p->d.init_pts, p->d.pts, (size_t)p->d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (p->d.phong) {
        HIPCHECK(hipMemcpyAsync(p->d.init_nrm, p->d.nrm, (size_t)p->d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(p->d.init_sh, p->d.sh, (size_t)p->d.nsh * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    rc = reset_solver(p);
    if (rc) return rc;
    HIPCHECK(hipEventRecord(p->ev_begin, s));
    p->began = true;
    return SSBA_OK;
}

int ssba_solve_restart(ssba_problem *p) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->began) return SSBA_ERR_STATE;
    return reset_solver(p);
}

static int fetch_state(ssba_problem *p);

static int enqueue_n(ssba_problem *p, int n) {
    int i = 0;
    for (; i + GRAPH_ITERS <= n; i += GRAPH_ITERS) {         // batches where the handle replays graphs
        const int rc = enqueue_batch(p);
        if (rc == SSBA_ERR_STATE) break;
        if (rc) return rc;
    }
    for (; i < n; ++i) {
        int rc = enqueue_iteration(p);
        if (rc) return rc;
    }
    return SSBA_OK;
}

int ssba_solve_step(ssba_problem *p, int n) {
    if (!p || n < 0) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->began) return SSBA_ERR_STATE;
    if (!p->d.constrained) return enqueue_n(p, n);
    // Bounds: an iteration whose full step fails the device-side Armijo test parks the solver, and everything enqueued
    // behind it does nothing.  So the host stays ONE iteration ahead and looks at the state of the iteration before (as
    // ssba_solve does): a parked search costs one idle iteration, is finished on the host, and the count goes on.
    hipStream_t st = p->launcher.stream;
    if (!p->ls_ring) {
        HIPCHECK(pool_host_malloc((void **)&p->ls_ring, 2 * sizeof(State)));
        HIPCHECK(hipEventCreateWithFlags(&p->ls_ev[0], hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&p->ls_ev[1], hipEventDisableTiming));
    }
    int rc = fetch_state(p);
    if (rc) return rc;
    if (search_pending(p) && (rc = finish_pending_search(p))) return rc;
    if (lockstep(p)) {
        for (int i = 0; i < n; ++i) {
            if ((rc = enqueue_iteration(p))) return rc;
            if ((rc = fetch_state(p))) return rc;
            if (search_pending(p) && (rc = finish_pending_search(p))) return rc;
        }
        return SSBA_OK;
    }
    int judged = 0;                 // iterations of this call that reached their accept / reject decision
    long it = 0;
    bool outstanding = false;       // an enqueued iteration whose state has not been looked at
    while (judged < n || outstanding) {
        const bool more = judged + (outstanding ? 1 : 0) < n;
        int slot = -1;
        if (more) {
            if ((rc = enqueue_iteration(p))) return rc;
            slot = (int)(it++ & 1);
            HIPCHECK(hipMemcpyAsync(&p->ls_ring[slot], p->d.st, sizeof(State), hipMemcpyDeviceToHost, st));
            HIPCHECK(hipEventRecord(p->ls_ev[slot], st));
        }
        if (outstanding) {          // the iteration before the one just enqueued
            const int prev = more ? slot ^ 1 : (int)((it - 1) & 1);
            HIPCHECK(hipEventSynchronize(p->ls_ev[prev]));
            if (p->ls_ring[prev].terminated && p->ls_ring[prev].termination_type == LS_PENDING) {
                // parked: the iteration enqueued behind it (if any) did nothing and does not count
                if ((rc = fetch_state(p))) return rc;
                if (search_pending(p) && (rc = finish_pending_search(p))) return rc;
                ++judged;
                outstanding = false;
                continue;
            }
            ++judged;
        }
        outstanding = more;
    }
    return SSBA_OK;
}

int ssba_synchronize(ssba_problem *p) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    HIPCHECK(hipStreamSynchronize(p->launcher.stream));
    p->launcher.collect();
    return SSBA_OK;
}

static int fetch_state(ssba_problem *p) {
    HIPCHECK(hipMemcpyAsync(p->h_state, p->d.st, sizeof(State), hipMemcpyDeviceToHost, p->launcher.stream));
    HIPCHECK(hipStreamSynchronize(p->launcher.stream));
    return SSBA_OK;
}

int ssba_solve_end(ssba_problem *p, ssba_summary *s) {
    ApiTimer api_timer("ssba_solve_end");
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->began) return SSBA_ERR_STATE;
    hipStream_t st = p->launcher.stream;
    int rc = fetch_state(p);
    if (rc) return rc;
    if (search_pending(p)) {        // (a stepwise caller stopped on a parked line search)
        if ((rc = finish_pending_search(p))) return rc;
        if ((rc = fetch_state(p))) return rc;
    }
    HIPCHECK(hipEventRecord(p->ev_end, st));
    HIPCHECK(hipStreamSynchronize(st));
    p->launcher.collect();
    const State &S = *p->h_state;
    const int n = std::min(S.log_count, p->d.log.capacity);
    p->log_cost.resize(n); p->log_cost_change.resize(n); p->log_gmax.resize(n); p->log_step.resize(n);
    p->log_rd.resize(n); p->log_radius.resize(n); p->log_ok.resize(n);
    if (n > 0 && (size_t)7 * p->d.log.capacity * sizeof(double) <= (size_t)256 << 10) {
        // the whole log block in one transfer (ensure_log): columns at stride `capacity`
        const size_t cap = (size_t)p->d.log.capacity;
        p->log_stage.resize(7 * cap);
        HIPCHECK(hipMemcpy(p->log_stage.data(), p->d.log.cost, (6 * cap + (cap + 1) / 2) * sizeof(double), hipMemcpyDeviceToHost));
        const double *b = p->log_stage.data();
        std::copy(b, b + n, p->log_cost.begin());
        std::copy(b + cap, b + cap + n, p->log_cost_change.begin());
        std::copy(b + 2 * cap, b + 2 * cap + n, p->log_gmax.begin());
        std::copy(b + 3 * cap, b + 3 * cap + n, p->log_step.begin());
        std::copy(b + 4 * cap, b + 4 * cap + n, p->log_rd.begin());
        std::copy(b + 5 * cap, b + 5 * cap + n, p->log_radius.begin());
        memcpy(p->log_ok.data(), b + 6 * cap, (size_t)n * sizeof(int32_t));
    } else if (n > 0) {
        HIPCHECK(hipMemcpy(p->log_cost.data(), p->d.log.cost, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(p->log_cost_change.data(), p->d.log.cost_change, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(p->log_gmax.data(), p->d.log.gmax, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(p->log_step.data(), p->d.log.step_norm, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(p->log_rd.data(), p->d.log.relative_decrease, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(p->log_radius.data(), p->d.log.radius, n * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(p->log_ok.data(), p->d.log.successful, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    const int term = S.terminated ? S.termination_type : SSBA_NO_CONVERGENCE;
    // the solution is usable unless the minimiser failed: write the lowest-cost iterate back
    if (term != SSBA_FAILURE) {
        if (p->d.part) {   // every rank holds the poses of its own chain: gather them (sum of owner-masked copies)
            launch_mask_unowned_poses(p->launcher, p->d, p->d.best_poses);
            if (p->xfn(p->xctx, p->d.best_poses, (uint64_t)p->d.P * 12, 0)) { set_error("exchange callback failed"); return SSBA_ERR_STATE; }
            HIPCHECK(hipStreamSynchronize(st));
        }
        rc = download_params(p, p->d.best_poses, p->d.best_pts, p->d.best_nrm, p->d.best_sh);
        if (rc) return rc;
    }
    if (s) {
        memset(s, 0, sizeof *s);
        s->termination_type = term;
        s->num_iterations = n;
        s->num_successful_steps = S.num_successful;
        s->num_unsuccessful_steps = S.num_unsuccessful;
        s->num_line_search_steps = S.ls_steps + p->num_line_search_steps;
        s->num_line_searches_on_device = S.ls_searches;
        s->num_line_searches_by_host = p->num_line_searches_by_host;
        s->initial_cost = S.initial_cost;
        // solver.cc SetSummaryFinalCost: minimum over the recorded iteration costs
        double fc = S.initial_cost;
        for (int i = 0; i < n; ++i) fc = std::min(fc, p->log_cost[i]);
        s->final_cost = fc;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev_begin, p->ev_end) == hipSuccess) s->device_time_s = 1e-3 * ms;
        s->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - p->t_begin).count();
    }
    p->began = false;
    return term == SSBA_FAILURE ? SSBA_ERR_NUMERICAL_FAILURE : SSBA_OK;
}

int ssba_solve(ssba_problem *p, const ssba_options *o, ssba_summary *s) {
    ApiTimer api_timer("ssba_solve");
    int rc = ssba_solve_begin(p, o, 0);
    if (rc) return rc;
    // Enqueue-ahead loop: the device decides accept/reject/convergence itself; the host
    // only polls the `terminated` word, one iteration behind the device.
    hipStream_t st = p->launcher.stream;
    hipEvent_t ev[2];
    hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
    hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
    State *ring = nullptr;
    if (pool_host_malloc((void **)&ring, 2 * sizeof(State)) != hipSuccess) {
        hipEventDestroy(ev[0]);
        hipEventDestroy(ev[1]);
        p->began = false;
        set_error("pinned host allocation failed");
        return SSBA_ERR_HIP;
    }
    long max_enqueue = (long)o->max_num_iterations + 3;
    bool done = false;
    for (long it = 0; it < max_enqueue && !done; ++it) {
        rc = enqueue_iteration(p);      // (one iteration per replay here: the host polls one round behind, and idle iterations
                                        // after termination would cost what pairs save)
        if (rc) break;
        if (lockstep(p)) {
            if ((rc = fetch_state(p))) break;
            if (search_pending(p)) {
                if ((rc = finish_pending_search(p))) break;
                if ((rc = fetch_state(p))) break;
            }
            if (p->h_state->terminated) done = true;
            continue;
        }
        const int slot = (int)(it & 1);
        hipMemcpyAsync(&ring[slot], p->d.st, sizeof(State), hipMemcpyDeviceToHost, st);
        hipEventRecord(ev[slot], st);
        if (it >= 1) {
            const int prev = (int)((it - 1) & 1);
            hipEventSynchronize(ev[prev]);
            if (ring[prev].terminated && ring[prev].termination_type == LS_PENDING) {
                // bounds: the full step of an iteration failed the device-side Armijo test; what was enqueued behind it has
                // done nothing.  Drain, let the host drive the search, go on.
                if ((rc = fetch_state(p))) break;
                if (search_pending(p) && (rc = finish_pending_search(p))) break;
                // (the ring slot of this round was filled before the search finished: refill both)
                hipMemcpyAsync(&ring[0], p->d.st, sizeof(State), hipMemcpyDeviceToHost, st);
                hipMemcpyAsync(&ring[1], p->d.st, sizeof(State), hipMemcpyDeviceToHost, st);
                hipEventRecord(ev[0], st);
                hipEventRecord(ev[1], st);
                max_enqueue += 1;       // the one iteration enqueued behind the parked one did nothing; the parked one is judged now and counts
                continue;
            }
            if (ring[prev].terminated) done = true;
            if (o->minimizer_progress_to_stdout)
                printf("iter %4d cost %.6e radius %.3e\n", ring[prev].iteration, ring[prev].x_cost, ring[prev].radius);
        }
    }
    hipStreamSynchronize(st);
    hipEventDestroy(ev[0]);
    hipEventDestroy(ev[1]);
    pool_host_free(ring);
    if (rc) { p->began = false; return rc; }
    return ssba_solve_end(p, s);
}

int ssba_brief_report(const ssba_summary *s, char *buf, size_t n) {
    if (!s || !buf || n == 0) return SSBA_ERR_INVALID_ARGUMENT;
    static const char *names[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    const char *t = (s->termination_type >= 0 && s->termination_type <= 2) ? names[s->termination_type] : "UNKNOWN";
    // format of ceres::Solver::Summary::BriefReport()
    snprintf(buf, n, "Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
             s->num_successful_steps + s->num_unsuccessful_steps, s->initial_cost, s->final_cost, t);
    return SSBA_OK;
}

int ssba_iteration_log(ssba_problem *p, int32_t capacity, double *cost, double *cost_change,
                       double *gradient_max_norm, double *step_norm, double *relative_decrease,
                       double *trust_region_radius, int32_t *step_is_successful) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    const int n = (int)p->log_cost.size();
    const int m = std::min(n, (int)capacity);
    for (int i = 0; i < m; ++i) {
        if (cost) cost[i] = p->log_cost[i];
        if (cost_change) cost_change[i] = p->log_cost_change[i];
        if (gradient_max_norm) gradient_max_norm[i] = p->log_gmax[i];
        if (step_norm) step_norm[i] = p->log_step[i];
        if (relative_decrease) relative_decrease[i] = p->log_rd[i];
        if (trust_region_radius) trust_region_radius[i] = p->log_radius[i];
        if (step_is_successful) step_is_successful[i] = p->log_ok[i];
    }
    return n;
}

int ssba_set_kernel_timing(ssba_problem *p, int mode) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    hipStreamSynchronize(p->launcher.stream);
    p->launcher.collect();
    p->launcher.timing = mode ? 1 : 0;
    for (int i = 0; i < KC_COUNT; ++i) { p->launcher.total_ms[i] = 0; p->launcher.launches[i] = 0; }
    return SSBA_OK;
}

int ssba_kernel_times(ssba_problem *p, ssba_kernel_time *rows, int32_t capacity, int32_t *num) {
    if (!p || !num) return SSBA_ERR_INVALID_ARGUMENT;
    hipStreamSynchronize(p->launcher.stream);
    p->launcher.collect();
    int n = 0;
    for (int i = 0; i < KC_COUNT && n < capacity; ++i) {
        if (rows) {
            memset(&rows[n], 0, sizeof rows[n]);
            snprintf(rows[n].name, sizeof rows[n].name, "%s", kKernelClassName[i]);
            rows[n].launches = p->launcher.launches[i];
            rows[n].total_ms = p->launcher.total_ms[i];
        }
        ++n;
    }
    *num = n;
    return SSBA_OK;
}

// ---------------------------------------------------------------------------------
// test hooks
// ---------------------------------------------------------------------------------
static int begin_hook(ssba_problem *p, const ssba_options *o, double radius) {
    if (!p->finalized) return SSBA_ERR_NOT_FINALIZED;
    if (p->d.part) { set_error("test hooks are not available on a partitioned problem"); return SSBA_ERR_UNSUPPORTED; }
    if (p->began) return SSBA_ERR_STATE;
    HIPCHECK(hipSetDevice(p->device));
    ssba_options opt;
    if (o) opt = *o; else ssba_default_options(&opt);
    if (radius > 0.0) opt.initial_trust_region_radius = radius;
    if (p->d.phong && p->xfn && p->d.nb && (p->d.dense || (p->d.constrained && p->world_size > NLS_X_RANKS))) {
        set_error("lighting terms: landmark sharding with free shared blocks is not available on the general layout (nor with bounds on more than 64 ranks)");
        return SSBA_ERR_UNSUPPORTED;
    }
    if (p->opt.trust_region_strategy_type != opt.trust_region_strategy_type)
        drop_graph(p);   // other kernel sequence (as in ssba_solve_begin)
    p->opt = opt;
    p->ignore_convergence = 1;
    p->d.huber_a = p->huber_a;
    int rc = ensure_log(p, 16);
    if (rc) return rc;
    rc = upload_params(p);
    if (rc) return rc;
    hipStream_t s = p->launcher.stream;
    HIPCHECK(hipMemcpyAsync(p->d.init_poses, p->d.poses, (size_t)p->d.P * 12 * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHECK(hipMemcpyAsync(p->d.init_pts, p->d.pts, (size_t)p->d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (p->d.phong) {
        HIPCHECK(hipMemcpyAsync(p->d.init_nrm, p->d.nrm, (size_t)p->d.Lpad * 3 * sizeof(double), hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(p->d.init_sh, p->d.sh, (size_t)p->d.nsh * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    return reset_solver(p);
}

int ssba_evaluate(ssba_problem *p, double *cost, double *g_p, double *g_l, double *H_pp, double *H_ll) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    int rc = begin_hook(p, nullptr, 0.0);
    if (rc) return rc;
    Dev &d = p->d;
    launch_linearize(p->launcher, d);
    HIPCHECK(hipStreamSynchronize(p->launcher.stream));
    HIPCHECK(hipGetLastError());
    const size_t Lp = (size_t)d.Lpad;
    if (cost) HIPCHECK(hipMemcpy(cost, d.xv + d.off_scal, sizeof(double), hipMemcpyDeviceToHost));
    if (g_p || H_pp) {
        std::vector<double> gp((size_t)p->P * 6), hp((size_t)p->P * 21);
        if (p->P) {
            HIPCHECK(hipMemcpy(gp.data(), d.gp, gp.size() * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(hp.data(), d.hpp, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        for (uint32_t k = 0; k < p->P; ++k) {
            const bool fr = p->pose_free[k] >= 0;
            if (g_p) for (int c = 0; c < 6; ++c) g_p[6 * (size_t)k + c] = fr ? gp[6 * (size_t)k + c] : 0.0;
            if (H_pp) {
                int n = 0;
                for (int a = 0; a < 6; ++a)
                    for (int c = a; c < 6; ++c, ++n) {
                        const double v = fr ? hp[21 * (size_t)k + n] : 0.0;
                        H_pp[36 * (size_t)k + 6 * a + c] = v;
                        H_pp[36 * (size_t)k + 6 * c + a] = v;
                    }
            }
        }
    }
    if (g_l || H_ll) {
        const int ld = d.phong ? 6 : 3, nh = ld * (ld + 1) / 2;
        std::vector<double> gl(Lp * ld), hl(Lp * nh);
        HIPCHECK(hipMemcpy(gl.data(), d.gl, gl.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(hl.data(), d.hll, hl.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (g_l) memset(g_l, 0, (size_t)p->L * ld * sizeof(double));
        if (H_ll) memset(H_ll, 0, (size_t)p->L * ld * ld * sizeof(double));
        for (size_t l = 0; l < Lp; ++l) {
            const uint32_t j = p->user_of_dev[l];
            if (j == 0xFFFFFFFFu) continue;
            if (g_l) for (int c = 0; c < ld; ++c) g_l[ld * (size_t)j + c] = gl[c * Lp + l];
            if (H_ll) {
                int c = 0;   // packed upper triangle, row-major
                for (int a = 0; a < ld; ++a)
                    for (int b = a; b < ld; ++b, ++c) {
                        H_ll[(size_t)ld * ld * j + ld * a + b] = hl[c * Lp + l];
                        H_ll[(size_t)ld * ld * j + ld * b + a] = hl[c * Lp + l];
                    }
            }
        }
    }
    return SSBA_OK;
}

// test hook: the line-search state machine on a recorded sequence of evaluations, on the host or in a one-lane kernel
namespace {
struct ArmijoTraceOut { int count; double optimal; };
__host__ __device__ inline void armijo_replay(const double *values, const double *gradients, int n, double c0, double g0, double dmax,
                                              double *steps_out, ArmijoTraceOut *out) {
    ssba::Armijo a;
    a.begin(c0, g0, dmax);
    int k = 0;
    while (!a.done && k < n) {
        steps_out[k] = a.current.x;
        a.feed(values[k], gradients[k]);
        ++k;
    }
    out->count = k;
    out->optimal = a.success ? a.optimal_step : -1.0;
}
__global__ void k_armijo_replay(const double *values, const double *gradients, int n, double c0, double g0, double dmax, double *steps_out,
                                ArmijoTraceOut *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) armijo_replay(values, gradients, n, c0, g0, dmax, steps_out, out);
}
}  // namespace
int ssba_armijo_trace(const double *values, const double *gradients, int32_t n, double initial_cost, double initial_gradient,
                      double dir_max_norm, double *steps_out, double *optimal_step, int32_t on_device, int32_t device) {
    if (!values || !gradients || !steps_out || !optimal_step || n < 0) return SSBA_ERR_INVALID_ARGUMENT;
    ArmijoTraceOut out{0, -1.0};
    if (!on_device) {
        armijo_replay(values, gradients, n, initial_cost, initial_gradient, dir_max_norm, steps_out, &out);
    } else {
        HIPCHECK(hipSetDevice(device < 0 ? 0 : device));
        double *dv = nullptr, *dg = nullptr, *ds = nullptr;
        ArmijoTraceOut *dout = nullptr;
        const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(double);
        hipError_t e = hipMalloc(&dv, bytes);
        if (e == hipSuccess) e = hipMalloc(&dg, bytes);
        if (e == hipSuccess) e = hipMalloc(&ds, bytes);
        if (e == hipSuccess) e = hipMalloc(&dout, sizeof out);
        if (e == hipSuccess) e = hipMemcpy(dv, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dg, gradients, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_armijo_replay, dim3(1), dim3(64), 0, 0, dv, dg, (int)n, initial_cost, initial_gradient, dir_max_norm, ds, dout);
            e = hipDeviceSynchronize();
        }
        if (e == hipSuccess) e = hipMemcpy(steps_out, ds, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&out, dout, sizeof out, hipMemcpyDeviceToHost);
        hipFree(dv); hipFree(dg); hipFree(ds); hipFree(dout);
        if (e != hipSuccess) { set_error(std::string("ssba_armijo_trace: ") + hipGetErrorString(e)); return SSBA_ERR_HIP; }
    }
    *optimal_step = out.optimal;
    return out.count;
}

int ssba_lm_step(ssba_problem *p, const ssba_options *o, double radius, double *S, double *rhs,
                 double *delta_p, double *delta_l, double *model_cost_change) {
    if (!p || !(radius > 0.0)) return SSBA_ERR_INVALID_ARGUMENT;
    int rc = begin_hook(p, o, radius);
    if (rc) return rc;
    Dev &d = p->d;
    Launcher &L = p->launcher;
    launch_linearize(L, d);
    if (d.dense) launch_dense_schur(L, d); else launch_schur(L, d);
    launch_finish_check(L, d);
    HIPCHECK(hipStreamSynchronize(L.stream));
    HIPCHECK(hipGetLastError());
    const int nf = d.nfree, n = 6 * nf;
    if ((S || rhs) && d.wide) {       // block tridiagonal over super-blocks of WSP poses: D row-major, L[I] = S(I, I - 1)
        const WideSys &w = L.wide;
        std::vector<double> M((size_t)w.count);
        HIPCHECK(hipMemcpy(M.data(), w.xw, M.size() * sizeof(double), hipMemcpyDeviceToHost));
        const size_t wb = (size_t)WBD * WBD;
        if (S) {
            memset(S, 0, (size_t)n * n * sizeof(double));
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const int I = i / WBD, J = j / WBD;
                    double v = 0.0;
                    if (I == J) v = M[(size_t)I * wb + (size_t)(i % WBD) * WBD + (j % WBD)];
                    else if (I == J + 1) v = M[w.off_L + (size_t)I * wb + (size_t)(i % WBD) * WBD + (j % WBD)];
                    else if (J == I + 1) v = M[w.off_L + (size_t)J * wb + (size_t)(j % WBD) * WBD + (i % WBD)];
                    S[(size_t)i * n + j] = v;
                }
        }
        if (rhs) for (int i = 0; i < n; ++i) rhs[i] = M[w.off_rhs + i];
    } else if ((S || rhs) && d.dense) {      // lower triangle + the right-hand-side row of the dense matrix
        std::vector<double> M((size_t)(d.dn_pad + 1) * d.dn_pad);
        if (!M.empty()) HIPCHECK(hipMemcpy(M.data(), d.dn_S, M.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (S)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) S[(size_t)i * n + j] = S[(size_t)j * n + i] = M[(size_t)i * d.dn_pad + j];
        if (rhs) for (int i = 0; i < n; ++i) rhs[i] = M[(size_t)d.dn_pad * d.dn_pad + i];
    } else if (S || rhs) {
        const size_t blk = (size_t)BD * BD;
        std::vector<double> D((size_t)d.Nsb * blk), Lb((size_t)d.Nsb * blk), r((size_t)d.Nsb * BD);
        HIPCHECK(hipMemcpy(D.data(), d.xv + d.off_D, D.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(Lb.data(), d.xv + d.off_L, Lb.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(r.data(), d.xv + d.off_rhs, r.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (S) {
            memset(S, 0, (size_t)n * n * sizeof(double));
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const int I = i / BD, J = j / BD;
                    double v = 0.0;
                    if (I == J) v = D[(size_t)I * blk + (size_t)(i % BD) * BD + (j % BD)];
                    // even-indexed coupling blocks are stored transposed
                    else if (I == J + 1) v = (I & 1) ? Lb[(size_t)I * blk + (size_t)(i % BD) * BD + (j % BD)]
                                                     : Lb[(size_t)I * blk + (size_t)(j % BD) * BD + (i % BD)];
                    else if (J == I + 1) v = (J & 1) ? Lb[(size_t)J * blk + (size_t)(j % BD) * BD + (i % BD)]
                                                     : Lb[(size_t)J * blk + (size_t)(i % BD) * BD + (j % BD)];
                    S[(size_t)i * n + j] = v;
                }
        }
        if (rhs) for (int i = 0; i < n; ++i) rhs[i] = r[i];
    }
    if (d.dense) launch_dense_solve(L, d);
    else { launch_bcr(L, d); if (d.nb) launch_border_solve(L, d); }
    launch_update_eval(L, d);
    HIPCHECK(hipStreamSynchronize(L.stream));
    HIPCHECK(hipGetLastError());
    rc = fetch_state(p);
    if (rc) return rc;
    double sc[NSCAL];
    HIPCHECK(hipMemcpy(sc, d.scal2, sizeof sc, hipMemcpyDeviceToHost));
    if (p->h_state->step_failed || sc[3] != 0.0) return SSBA_ERR_NUMERICAL_FAILURE;
    if (model_cost_change) {
        *model_cost_change = sc[1];
        if (d.n_pf) {     // rows of the unary pose residual blocks (summed by k_decide in a solve)
            std::vector<double> pp((size_t)d.n_pose_blocks * NPP);
            HIPCHECK(hipMemcpy(pp.data(), d.part_pose, pp.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int i = 0; i < d.n_pose_blocks; ++i) *model_cost_change += pp[(size_t)i * NPP + 3];
        }
    }
    if (delta_p) {
        std::vector<double> x((size_t)d.nf_pad * 6);
        HIPCHECK(hipMemcpy(x.data(), d.x0, x.size() * sizeof(double), hipMemcpyDeviceToHost));
        memset(delta_p, 0, (size_t)p->P * 6 * sizeof(double));
        for (int f = 0; f < nf; ++f)
            for (int c = 0; c < 6; ++c) delta_p[6 * (size_t)p->free_pose[f] + c] = x[6 * (size_t)f + c];
    }
    if (delta_l && d.phong) {
        const size_t Lp = (size_t)d.Lpad;
        std::vector<double> a(Lp * 6);
        HIPCHECK(hipMemcpy(a.data(), d.dlm, a.size() * sizeof(double), hipMemcpyDeviceToHost));
        memset(delta_l, 0, (size_t)p->L * 6 * sizeof(double));
        for (size_t l = 0; l < Lp; ++l) {
            const uint32_t j = p->user_of_dev[l];
            if (j == 0xFFFFFFFFu) continue;
            for (int c = 0; c < 6; ++c) delta_l[6 * (size_t)j + c] = a[c * Lp + l];
        }
    } else if (delta_l) {
        const size_t Lp = (size_t)d.Lpad;
        std::vector<double> a(Lp * 3), b(Lp * 3);
        HIPCHECK(hipMemcpy(a.data(), d.cand_pts, a.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(b.data(), d.pts, b.size() * sizeof(double), hipMemcpyDeviceToHost));
        memset(delta_l, 0, (size_t)p->L * 3 * sizeof(double));
        for (size_t l = 0; l < Lp; ++l) {
            const uint32_t j = p->user_of_dev[l];
            if (j == 0xFFFFFFFFu) continue;
            for (int c = 0; c < 3; ++c) delta_l[3 * (size_t)j + c] = a[c * Lp + l] - b[c * Lp + l];
        }
    }
    return SSBA_OK;
}

// ceres::Covariance::Compute + GetCovarianceBlockInTangentSpace for one pose block (tests/dataset_vo_sun.cpp:159-183):
// the pose's 6x6 block of (J^T J)^-1 in local coordinates = the same block of the inverse of the (undamped) reduced
// camera system.  Six unit right-hand sides go through the block-cyclic-reduction factors of S.
int ssba_pose_covariance(ssba_problem *p, uint32_t pose, double cov[36]) {
    ApiTimer api_timer("ssba_pose_covariance");
    if (!p || !cov || pose >= p->P) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->finalized) return SSBA_ERR_NOT_FINALIZED;
    if (p->d.cb) {      // the closure border has no covariance sweep: the general path has (symbolic phase again, once)
        int rc = refinalize_without_closure_border(p, "covariance");
        if (rc) return rc;
    }
    if (p->d.wide) {
        int rc = refinalize_without_wide(p, "covariance");
        if (rc) return rc;
    }
    // (lighting terms: the 3-wide local parameterisation of a unit normal has rank 2 -- UnitVectorPerturbation, perturbations.hpp:98-102 --
    // so the UNDAMPED landmark blocks this sweep needs are singular; the oracle's reduced system at radius 1e300 fails the same way)
    if (p->d.part || p->d.phong) {
        set_error("covariance: not available on partitioned problems or with lighting terms");
        return SSBA_ERR_UNSUPPORTED;
    }
    const int f = p->pose_free[pose];
    if (f < 0) { set_error("covariance of a constant pose"); return SSBA_ERR_INVALID_ARGUMENT; }
    Dev &d = p->d;
    int rc;
    if (d.dense) {
        // general layout: the six unit vectors go into rows 0..5 of the right-hand-side block row, the blocked
        // Cholesky forward-solves them with the factorisation, then one back-substitution sweep per row
        ssba_options o;
        ssba_default_options(&o);
        if ((rc = begin_hook(p, &o, 1e300))) return rc;
        Launcher &L = p->launcher;
        launch_linearize(L, d);
        launch_dense_schur(L, d);
        const size_t lda = (size_t)d.dn_pad;
        HIPCHECK(hipMemsetAsync(d.dn_S + lda * lda, 0, (size_t)DN_BS * lda * sizeof(double), L.stream));
        const double one = 1.0;
        for (int c = 0; c < 6; ++c)
            HIPCHECK(hipMemcpyAsync(d.dn_S + (lda + c) * lda + (size_t)f * 6 + c, &one, sizeof one, hipMemcpyHostToDevice, L.stream));
        launch_finish_check(L, d);
        launch_dense_solve(L, d, 6);
        HIPCHECK(hipStreamSynchronize(L.stream));
        HIPCHECK(hipGetLastError());
        if ((rc = fetch_state(p))) return rc;
        if (p->h_state->step_failed) { set_error("covariance: the reduced camera system is not positive definite (gauge freedom?)"); return SSBA_ERR_NUMERICAL_FAILURE; }
        for (int c = 0; c < 6; ++c) {
            double col[6];
            HIPCHECK(hipMemcpy(col, d.dn_S + (lda + c) * lda + (size_t)f * 6, sizeof col, hipMemcpyDeviceToHost));
            for (int r = 0; r < 6; ++r) cov[6 * r + c] = col[r];
        }
        return SSBA_OK;
    }
    if (!d.Spb) {      // multi-right-hand-side buffers are only allocated with a border: add them now
        drop_graph(p);
        if ((rc = dzero(p, &d.Spb, (size_t)d.nf_pad * 6 * NBP))) return rc;
        if ((rc = dzero(p, &d.Zb, (size_t)d.nf_pad * 6 * NBP))) return rc;
        for (int l = 0; l < d.n_levels; ++l)
            if ((rc = dzero(p, &d.lev[l].B, (size_t)d.lev[l].n * BD * NBP))) return rc;
        if (configure_border()) { set_error("hipFuncSetAttribute(border kernels) failed"); return SSBA_ERR_HIP; }
    }
    ssba_options o;
    ssba_default_options(&o);
    rc = begin_hook(p, &o, 1e300);        // radius -> infinity: no Levenberg-Marquardt damping in S
    if (rc) return rc;
    Launcher &L = p->launcher;
    launch_linearize(L, d);
    launch_schur(L, d);
    launch_finish_check(L, d);
    launch_bcr(L, d, false);       // the multi-right-hand-side sweeps need the factors of every level
    HIPCHECK(hipMemsetAsync(d.Spb, 0, (size_t)d.nf_pad * 6 * NBP * sizeof(double), L.stream));
    {
        std::vector<double> unit((size_t)6 * NBP, 0.0);
        for (int c = 0; c < 6; ++c) unit[(size_t)c * NBP + c] = 1.0;
        HIPCHECK(hipMemcpyAsync(d.Spb + (size_t)f * 6 * NBP, unit.data(), unit.size() * sizeof(double), hipMemcpyHostToDevice, L.stream));
        HIPCHECK(hipStreamSynchronize(L.stream));
    }
    launch_bcr_multi_rhs(L, d);
    HIPCHECK(hipStreamSynchronize(L.stream));
    HIPCHECK(hipGetLastError());
    if ((rc = fetch_state(p))) return rc;
    if (p->h_state->step_failed) { set_error("covariance: the reduced camera system is not positive definite (gauge freedom?)"); return SSBA_ERR_NUMERICAL_FAILURE; }
    std::vector<double> z((size_t)6 * NBP);
    HIPCHECK(hipMemcpy(z.data(), d.Zb + (size_t)f * 6 * NBP, z.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) cov[6 * r + c] = z[(size_t)r * NBP + c];
    HIPCHECK(hipMemsetAsync(d.Spb, 0, (size_t)d.nf_pad * 6 * NBP * sizeof(double), L.stream));
    return SSBA_OK;
}

int ssba_border_system(ssba_problem *p, uint32_t *nb_out, double *S_pb, double *S_bb, double *rhs_b, double *delta_b) {
    if (!p) return SSBA_ERR_INVALID_ARGUMENT;
    if (!p->finalized) return SSBA_ERR_NOT_FINALIZED;
    Dev &d = p->d;
    if (nb_out) *nb_out = (uint32_t)d.nb;
    if (!d.nb) return SSBA_OK;
    HIPCHECK(hipStreamSynchronize(p->launcher.stream));
    const int nb = d.nb, nf = d.nfree;
    std::vector<double> bs((size_t)BS_COUNT);
    HIPCHECK(hipMemcpy(bs.data(), d.bsys, bs.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(p->h_state, d.st, sizeof(State), hipMemcpyDeviceToHost));
    if (S_pb) {
        std::vector<double> sp((size_t)d.nf_pad * 6 * NBP);
        HIPCHECK(hipMemcpy(sp.data(), d.Spb, sp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < 6 * nf; ++i)
            for (int c = 0; c < nb; ++c) S_pb[(size_t)i * nb + c] = sp[(size_t)i * NBP + c];
    }
    const State &S = *p->h_state;
    const double radius = S.opt.strategy ? 1.0 / S.mu : S.radius;
    for (int a = 0; a < nb; ++a) {
        if (rhs_b) rhs_b[a] = -bs[BS_RHS + a];
        if (delta_b) delta_b[a] = bs[BS_DB + a];
        if (S_bb)
            for (int c = 0; c < nb; ++c) {
                double v = bs[BS_SBB + a * NBP + c];
                if (a == c) {
                    const double s2 = bs[BS_S + a] * bs[BS_S + a];
                    v += std::min(std::max(bs[BS_H + a] * s2, S.opt.min_lm_diag), S.opt.max_lm_diag) / (radius * s2);
                }
                S_bb[(size_t)a * nb + c] = v;
            }
    }
    return SSBA_OK;
}

}  // extern "C"
