// HIP kernels of the stereo-BA hot path for gfx950 (MI355X).  fp64 throughout.
//
// Kernel classes (one LM iteration, in launch order; DESIGN.md has the roofline
// of each):
//   k_linearize_landmarks  per-landmark H_ll, g_l, cost        (HBM stream, 1 lane/landmark)
//   k_linearize_poses      per-pose H_pp, g_p                   (gather, 1 block/pose)
//   k_schur_windows        sum_j (W C^-1)_a W_b^T per window    (fp64 FMA bound, output-stationary)
//   k_assemble_reduced     slabs -> block-tridiagonal S, rhs    (HBM)
//   k_finish_reduced       Jacobi scale + LM damping on diag(S)
//   k_check                Ceres FinalizeIterationAndCheck...   (1 block)
//   k_bcr_factor/reduce/backsub  block cyclic reduction of S    (latency bound, log2 levels)
//   k_pose_update          candidate poses = Plus(x, delta_p)
//   k_backsub_eval         delta_l, model cost change, candidate cost (HBM stream)
//   k_decide               step quality, accept/reject, radius  (1 block)
//   k_commit / k_best      x <- candidate, best <- x
//
// Arithmetic follows /root/reference include/ceres_slam/{stereo_reprojection_error,
// stereo_camera,perturbations}.hpp and geometry/{se3group,so3group}.hpp; the
// trust-region logic follows Ceres 1.13/1.14 (see oracle/ssba_oracle.h).
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "ssba_types.h"
#include "ssba_launch.h"
#include "ssba_device.h"
#include "ssba_posefactor_device.h"
#include "ssba_check.h"

namespace ssba {

__constant__ uint8_t c_pair_a[NPAIR];
__constant__ uint8_t c_pair_b[NPAIR];

int upload_pair_table(hipStream_t s) {
    uint8_t a[NPAIR], b[NPAIR];
    int n = 0;
    for (int i = 0; i < TW; ++i)
        for (int j = i; j < TW; ++j) { a[n] = (uint8_t)i; b[n] = (uint8_t)j; ++n; }
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_pair_a), a, NPAIR, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(c_pair_b), b, NPAIR, 0, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}

// Observations of one landmark.  Windowed layout (DN = false): TW slots of the landmark's window in the
// transposed ELL arrays, present where the mask bit is set.  General layout (DN = true, ssba_dense.hip):
// the landmark's contiguous range of the landmark-major observation arrays.
template <bool DN> struct LmObs;
template <> struct LmObs<false> {
    uint32_t mask, win;
    size_t obase;
    __device__ __forceinline__ LmObs(const Dev &d, int l, uint32_t m)
        : mask(m), win(d.lm_win[l]), obase((size_t)(l >> 6) * (TW * LMG) + (l & 63)) {}
    __device__ __forceinline__ int count() const { return TW; }
    __device__ __forceinline__ bool has(int s) const { return (mask >> s) & 1u; }
    __device__ __forceinline__ uint32_t pose(const Dev &d, int s) const { return d.win_pose[win * TW + s]; }
    __device__ __forceinline__ double u(const Dev &d, int s) const { return d.ou[obase + s * LMG]; }
    __device__ __forceinline__ double v(const Dev &d, int s) const { return d.ov[obase + s * LMG]; }
    __device__ __forceinline__ double dd(const Dev &d, int s) const { return d.od[obase + s * LMG]; }
    __device__ __forceinline__ void stiffness(const Dev &d, int, double S[9]) const {
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] = d.S[c];
    }
};
template <> struct LmObs<true> {
    uint32_t b, n;
    __device__ __forceinline__ LmObs(const Dev &d, int l, uint32_t) : b(d.dn_lm_start[l]), n(d.dn_lm_start[l + 1] - d.dn_lm_start[l]) {}
    __device__ __forceinline__ int count() const { return (int)n; }
    __device__ __forceinline__ bool has(int) const { return true; }
    __device__ __forceinline__ uint32_t pose(const Dev &d, int s) const { return d.dn_obs_pose[b + s]; }
    __device__ __forceinline__ double u(const Dev &d, int s) const { return d.dn_u[b + s]; }
    __device__ __forceinline__ double v(const Dev &d, int s) const { return d.dn_v[b + s]; }
    __device__ __forceinline__ double dd(const Dev &d, int s) const { return d.dn_d[b + s]; }
    __device__ __forceinline__ void stiffness(const Dev &d, int s, double S[9]) const {     // per residual block when given
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] = d.dn_Sobs ? d.dn_Sobs[(size_t)(b + s) * 9 + c] : d.S[c];
    }
};

// ------------------------------------------------------------------ kernels ---

// One lane per landmark.  Streams the ELL observation arrays (coalesced), gathers the
// (few, shared) pose blocks through L1/L2.  Writes H_ll (6), g_l (3) component-major and
// per-block partials {cost, |x_l|^2, max|g_l|}.  At iteration 0 also the Jacobi scale.
template <bool DN> __global__ __launch_bounds__(256) void k_linearize_landmarks(Dev d) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double cost = 0.0, xn = 0.0, gm = 0.0;
    if (mask) {
        const LmObs<DN> ob(d, l, mask);
        const double px = d.pts[l], py = d.pts[(size_t)d.Lpad + l], pz = d.pts[2 * (size_t)d.Lpad + l];
        double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        for (int s = 0; s < ob.count(); ++s) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            const double *T = d.poses + (size_t)k * 12;
            ObsLin o;
            double Sk[9];
            ob.stiffness(d, s, Sk);
            obs_linearize_S(d, Sk, T, px, py, pz, ob.u(d, s), ob.v(d, s), ob.dd(d, s), o);
            double Jl[9];
            jac_point(o, T, Jl);
            cost += o.half_rho;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                h[0] += Jl[3 * i] * Jl[3 * i];
                h[1] += Jl[3 * i] * Jl[3 * i + 1];
                h[2] += Jl[3 * i] * Jl[3 * i + 2];
                h[3] += Jl[3 * i + 1] * Jl[3 * i + 1];
                h[4] += Jl[3 * i + 1] * Jl[3 * i + 2];
                h[5] += Jl[3 * i + 2] * Jl[3 * i + 2];
                g[0] += Jl[3 * i] * o.r[i];
                g[1] += Jl[3 * i + 1] * o.r[i];
                g[2] += Jl[3 * i + 2] * o.r[i];
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) d.hll[(size_t)c * d.Lpad + l] = h[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) d.gl[(size_t)c * d.Lpad + l] = g[c];
        if (st.iteration == 0) {   // Jacobi scaling, computed once [trust_region_minimizer.cc]
            const double hd[3] = {h[0], h[3], h[5]};
#pragma unroll
            for (int c = 0; c < 3; ++c)
                d.sl[(size_t)c * d.Lpad + l] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(hd[c])) : 1.0;
        }
        xn = px * px + py * py + pz * pz;
        gm = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
    }
    const double c0 = block_sum(cost, sm);
    const double c1 = block_sum(xn, sm);
    const double c2 = block_max(gm, sm);
    if (threadIdx.x == 0) {
        d.part_lin[blockIdx.x * 4 + 0] = c0;
        d.part_lin[blockIdx.x * 4 + 1] = c1;
        d.part_lin[blockIdx.x * 4 + 2] = c2;
    }
}

// One block per pose: gathers that pose's observations through the pose-major
// reference list, accumulates the 21 unique entries of H_pp and g_p in registers and
// reduces them across the block in a fixed order (no float atomics).
// PS / PT: where the current iterate lives (the candidate buffers when the launch also commits the step the last
// iteration accepted; then `commit` copies this pose into d.poses).  Call with the whole workgroup, k a free pose.
constexpr int LP_THREADS = 128, LP_CHUNK = 5;
template <bool DN, int NT, int CH> __device__ __forceinline__ void lin_pose_body(const Dev &d, int k, const double *__restrict__ PS,
                                                                 const double *__restrict__ PT, bool commit) {
    __shared__ double sm[NT / 64][27];
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = PS[(size_t)k * 12 + i];
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    const uint32_t b = DN ? d.dn_pose_start[k] : d.pose_obs_start[k], e = DN ? d.dn_pose_start[k + 1] : d.pose_obs_start[k + 1];
    auto accumulate = [&](const double *Sk, double px, double py, double pz, double ou, double ov, double od) {
        ObsLin o;
        obs_linearize_S(d, Sk, T, px, py, pz, ou, ov, od, o);
        pose_normal_terms(o, acc);
    };
    double Sk[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) Sk[c] = d.S[c];
    if (DN && d.dn_prec) {
        // as below; the records (u, v, d, landmark) of a pose's observations lie in list order (a copy made by ssba_finalize:
        // through the landmark-major arrays every lane touched three cache lines of its own -- 87 -> 25 us at 600 poses x 2 400 observations)
        for (uint32_t i0 = b + threadIdx.x; i0 < e; i0 += NT * CH) {
            double4 rec[CH];
            double pt[CH][3];
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (i0 + NT * q < e) rec[q] = reinterpret_cast<const double4 *>(d.dn_prec)[i0 + NT * q];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                if (i0 + NT * q >= e) continue;
                const size_t l = (size_t)__double_as_longlong(rec[q].w);
                pt[q][0] = PT[l]; pt[q][1] = PT[(size_t)d.Lpad + l]; pt[q][2] = PT[2 * (size_t)d.Lpad + l];
            }
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (i0 + NT * q < e) accumulate(Sk, pt[q][0], pt[q][1], pt[q][2], rec[q].x, rec[q].y, rec[q].z);
        }
    } else if (DN) {
        for (uint32_t i = b + threadIdx.x; i < e; i += NT) {
            const uint32_t oi = d.dn_pose_obs[i];
            const int l = (int)d.dn_obs_lm[oi];
            double So[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) So[c] = d.dn_Sobs ? d.dn_Sobs[(size_t)oi * 9 + c] : Sk[c];
            accumulate(So, PT[l], PT[(size_t)d.Lpad + l], PT[2 * (size_t)d.Lpad + l], d.dn_u[oi], d.dn_v[oi], d.dn_d[oi]);
        }
    } else {
        // CH observations per round: their references, then their 6 CH operands are in flight together (a rolled loop
        // pays two dependent memory round trips per observation)
        for (uint32_t i0 = b + threadIdx.x; i0 < e; i0 += NT * CH) {
            uint32_t ref[CH];
            double in[CH][6];
#pragma unroll
            for (int q = 0; q < CH; ++q) ref[q] = i0 + NT * q < e ? d.pose_obs_ref[i0 + NT * q] : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                if (ref[q] == 0xFFFFFFFFu) continue;
                const int l = (int)(ref[q] >> 4), sl = (int)(ref[q] & 15u);
                const size_t oi = (size_t)(l >> 6) * (TW * LMG) + (size_t)sl * LMG + (l & 63);
                in[q][0] = PT[l]; in[q][1] = PT[(size_t)d.Lpad + l]; in[q][2] = PT[2 * (size_t)d.Lpad + l];
                in[q][3] = d.ou[oi]; in[q][4] = d.ov[oi]; in[q][5] = d.od[oi];
            }
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (ref[q] != 0xFFFFFFFFu) accumulate(Sk, in[q][0], in[q][1], in[q][2], in[q][3], in[q][4], in[q][5]);
        }
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        const double v = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        double v = 0.0;
        if (threadIdx.x < 27)
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) v += sm[w][threadIdx.x];
        if (d.n_pf && d.pf_owner) {
            // unary residual blocks of this pose (pose prior, sun sensor): every one of the 28 lanes evaluates them
            // and takes its own entry of J^T J (21), J^T r (6) or the cost (lane 27).  With landmark sharding the poses are
            // replicated and these sums are added over the ranks: ONE rank contributes the unary blocks (pf_owner)
            int a = 0, c = 0;
            if (threadIdx.x < 21) { int n = threadIdx.x; while (n >= 6 - a) { n -= 6 - a; ++a; } c = a + n; }
            for (uint32_t e = d.pf_start[k]; e < d.pf_start[k + 1]; ++e) {
                double r[6], J[36];
                int dim;
                const int other = pf_other_pose(d, (int)e);
                const double cost = pf_evaluate(d, (int)e, T, other >= 0 ? PS + (size_t)other * 12 : nullptr, r, J, &dim);
                if (threadIdx.x < 21) { for (int m = 0; m < dim; ++m) v += J[6 * m + a] * J[6 * m + c]; }
                else if (threadIdx.x < 27) { for (int m = 0; m < dim; ++m) v += J[6 * m + (threadIdx.x - 21)] * r[m]; }
                else if (pf_counts_cost(d, (int)e)) v += cost;
            }
        }
        if (threadIdx.x < 21) d.hpp[(size_t)k * 21 + threadIdx.x] = v;
        else if (threadIdx.x < 27) d.gp[(size_t)k * 6 + (threadIdx.x - 21)] = v;
        else if (d.n_pf) d.pf_cost[k] = v;
    }
    constexpr int CL = NT >= 128 ? 64 : 32;      // commit lanes: clear of the 28 lanes that store the sums
    if (commit && threadIdx.x >= CL && threadIdx.x < CL + 12) d.poses[(size_t)k * 12 + threadIdx.x - CL] = PS[(size_t)k * 12 + threadIdx.x - CL];
}
// fuse (single GPU, LM): the step the decision kernel accepted is committed on the way -- the iterate is read from the
// candidate buffers and this pose written to x (k_linearize_landmarks_w(.., fuse), which runs first, did the points and
// read the candidate poses too), so k_commit is not launched.
template <bool DN, int NT, int CH> __global__ __launch_bounds__(NT) void k_linearize_poses(Dev d, int fuse) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    const int k = xcd_contiguous_item((int)blockIdx.x, d.P);      // neighbouring poses (they see the same landmarks) on one XCD's L2
    if (k < 0 || d.pose_free[k] < 0) return;
    const bool commit = fuse && st.accepted;
    if (!fuse) {
        // a landmark shard sees a fraction of the poses (rank r of N: its own ~P/N and the neighbours' edges): the blocks
        // of a pose nothing refers to here stay at the zeros they were allocated with, and the workgroup leaves at once
        // (rank 4 of 8 at C2 x 8: 98 -> ~30 us per launch)
        const bool no_obs = DN ? d.dn_pose_start[k] == d.dn_pose_start[k + 1] : d.pose_obs_start[k] == d.pose_obs_start[k + 1];
        if (no_obs && (!d.n_pf || d.pf_start[k] == d.pf_start[k + 1])) return;
    }
    lin_pose_body<DN, NT, CH>(d, k, commit ? d.cand_poses : d.poses, commit ? d.cand_pts : d.pts, commit);
}

// Window layout, SP lanes per landmark: one block = one group of 64 landmarks (the ELL unit), wave w takes the slots
// w, w + SP, w + 2 SP, ... of every landmark, so the loads stay coalesced (lane = landmark) and the poses of a wave are
// broadcast reads, but SP times as many waves hide the dependent fp64 chains of the linearisation.  The SP partial
// sums per landmark are combined through LDS in a fixed order by wave 0.
// Waves per 64 landmarks (template parameter SP of the two kernels).  Sweep on C2 (profiles/r02_landmark_kernel_shape.txt):
// 1 / 2 / 3 / 4 / 6 waves -> 19.6 / 16.7 / 18.8 / 19.5 / 27.0 us (linearisation), 38.4 / 30.4 / 33.8 / 34.1 / 48.4 us (evaluation).
constexpr int LMW_SPLIT = 2;
// PS / PT / commit: see lin_pose_body
// DN (landmark-major lists of the general layout instead of window slots): the same split, wave w takes the observations
// w, w + SP, ... of every landmark's list (r04: 235 four-wave work-groups on 256 CUs at 60 000 landmarks left every SIMD
// with ONE wave of 24 dependent fp64 chains; 940 work-groups of SP = 4 put 3.7 on it).
template <bool DN, int SP> __device__ __forceinline__ void lin_landmarks_w_body(const Dev &d, const State &st, int grp, const double *__restrict__ PS,
                                                     const double *__restrict__ PT, bool commit) {
    __shared__ double sm[SP];
    __shared__ double red[SP > 1 ? SP - 1 : 1][10][LMG];
    const int w = threadIdx.x >> 6, li = threadIdx.x & 63;
    const int l = grp * LMG + li;
    const uint32_t mask = d.lm_mask[l];
    const double px = PT[l], py = PT[(size_t)d.Lpad + l], pz = PT[2 * (size_t)d.Lpad + l];
    if (commit && w == 0) { d.pts[l] = px; d.pts[(size_t)d.Lpad + l] = py; d.pts[2 * (size_t)d.Lpad + l] = pz; }
    double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, cost = 0.0;
    auto accumulate = [&](const ObsLin &o, const double *T) {
        double Jl[9];
        jac_point(o, T, Jl);
        cost += o.half_rho;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            h[0] += Jl[3 * i] * Jl[3 * i];
            h[1] += Jl[3 * i] * Jl[3 * i + 1];
            h[2] += Jl[3 * i] * Jl[3 * i + 2];
            h[3] += Jl[3 * i + 1] * Jl[3 * i + 1];
            h[4] += Jl[3 * i + 1] * Jl[3 * i + 2];
            h[5] += Jl[3 * i + 2] * Jl[3 * i + 2];
            g[0] += Jl[3 * i] * o.r[i];
            g[1] += Jl[3 * i + 1] * o.r[i];
            g[2] += Jl[3 * i + 2] * o.r[i];
        }
    };
    if (mask && !DN) {
        // window layout: the pose indices of this lane's slots up front, the operands of slot q + 1 requested before slot q is worked
        // on (the rolled loop made two dependent round trips per slot: see k_backsub_eval_w)
        const LmObs<DN> ob(d, l, mask);
        constexpr int NS = DN ? 1 : TW / SP;
        uint32_t kk[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) kk[q] = ob.has(w + q * SP) ? ob.pose(d, w + q * SP) : 0xFFFFFFFFu;
        struct In { double T[12], u, v, dd; };
        auto fetch = [&](int q, In &in) {
            if (kk[q] == 0xFFFFFFFFu) return;
            const int s = w + q * SP;
            const double *T = PS + (size_t)kk[q] * 12;
#pragma unroll
            for (int c = 0; c < 12; ++c) in.T[c] = T[c];
            in.u = ob.u(d, s); in.v = ob.v(d, s); in.dd = ob.dd(d, s);
        };
        In cur, nxt;
        fetch(0, cur);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (q + 1 < NS) fetch(q + 1, nxt);
            if (kk[q] != 0xFFFFFFFFu) {
                ObsLin o;
                obs_linearize(d, cur.T, px, py, pz, cur.u, cur.v, cur.dd, o);
                accumulate(o, cur.T);
            }
            if (q + 1 < NS) cur = nxt;
        }
    }
    if (mask && DN) {
        const LmObs<DN> ob(d, l, mask);
        for (int s = w; s < ob.count(); s += SP) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            const double *T = PS + (size_t)k * 12;
            ObsLin o;
            double Sk[9];
            ob.stiffness(d, s, Sk);
            obs_linearize_S(d, Sk, T, px, py, pz, ob.u(d, s), ob.v(d, s), ob.dd(d, s), o);
            accumulate(o, T);
        }
    }
    if (w > 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) red[w - 1][c][li] = h[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) red[w - 1][6 + c][li] = g[c];
        red[w - 1][9][li] = cost;
    }
    __syncthreads();
    double xn = 0.0, gm = 0.0;
    if (w == 0) {
#pragma unroll
        for (int q = 0; q < SP - 1; ++q) {
#pragma unroll
            for (int c = 0; c < 6; ++c) h[c] += red[q][c][li];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] += red[q][6 + c][li];
            cost += red[q][9][li];
        }
        if (mask) {
#pragma unroll
            for (int c = 0; c < 6; ++c) d.hll[(size_t)c * d.Lpad + l] = h[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) d.gl[(size_t)c * d.Lpad + l] = g[c];
            if (st.iteration == 0) {   // Jacobi scaling, computed once [trust_region_minimizer.cc]
                const double hd[3] = {h[0], h[3], h[5]};
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    d.sl[(size_t)c * d.Lpad + l] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(hd[c])) : 1.0;
            }
            xn = px * px + py * py + pz * pz;
            gm = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        }
    } else {
        cost = 0.0;
    }
    const double c0 = block_sum(cost, sm);
    const double c1 = block_sum(xn, sm);
    const double c2 = block_max(gm, sm);
    if (threadIdx.x == 0) {
        d.part_lin[grp * 4 + 0] = c0;
        d.part_lin[grp * 4 + 1] = c1;
        d.part_lin[grp * 4 + 2] = c2;
    }
}
template <bool DN, int SP> __global__ __launch_bounds__(64 * SP) void k_linearize_landmarks_w(Dev d, int fuse) {
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    const bool commit = fuse && st.accepted;
    lin_landmarks_w_body<DN, SP>(d, st, (int)blockIdx.x, commit ? d.cand_poses : d.poses, commit ? d.cand_pts : d.pts, commit);
}
// Both linearisation passes of the window layout in ONE launch: the first n_groups work-groups are k_linearize_landmarks_w<false,
// LMW_SPLIT>'s, the others k_linearize_poses<false, 128, 5>'s -- same block shape, neither reads what the other writes (with the
// commit folded in both read the candidate buffers and write x).  The second pass's ramp fills the first one's tail: 31.1 -> 28.4 us
// at C2 (0.3458 -> 0.3431 ms per iteration, same bits).  SSBA_LIN_TWO_LAUNCHES=1 keeps the two launches (A/B, tests).
__global__ __launch_bounds__(LP_THREADS) void k_linearize_both(Dev d, int fuse, int n_groups) {
    static_assert(LP_THREADS == 64 * LMW_SPLIT, "one block shape for both passes");
    const State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    const bool commit = fuse && st.accepted;
    const double *PS = commit ? d.cand_poses : d.poses, *PT = commit ? d.cand_pts : d.pts;
    // (the landmark groups first: with the pose work-groups -- the longer ones -- in front the launch measured 10 us SLOWER at C2)
    if ((int)blockIdx.x < n_groups) {
        lin_landmarks_w_body<false, LMW_SPLIT>(d, st, (int)blockIdx.x, PS, PT, commit);
        return;
    }
    const int k = xcd_contiguous_item((int)blockIdx.x - n_groups, d.P);
    if (k < 0 || d.pose_free[k] < 0) return;
    if (!fuse) {
        const bool no_obs = d.pose_obs_start[k] == d.pose_obs_start[k + 1];
        if (no_obs && (!d.n_pf || d.pf_start[k] == d.pf_start[k + 1])) return;
    }
    lin_pose_body<false, LP_THREADS, LP_CHUNK>(d, k, PS, PT, commit);
}


// Schur complement contributions, output-stationary on the fp64 matrix cores: one 256-thread block per work item (a
// window or a slice of it).  Batches of 21 landmarks are half-linearised by 252 (landmark, slot) producer lanes --
// W = J_p^T J_l is recomputed from the observation, never stored in HBM.  With C^-1 = M^T M (M = L^-1 of the damped
// landmark block) the contribution W_a C^-1 W_b^T is the SYMMETRIC product Z_a Z_b^T of one factor Z = W M^T, so one
// k-major matrix  Zm[k][col]  is staged in LDS: k = 3*landmark + c (63 per batch + one zero row), col = 6*slot + dof
// (72, padded to 80); column 72 carries u = M g_l, so the reduced gradient sum_j Z u falls out of the same product.
//     S[72 x 73] += Zm^T Zm
// is accumulated by the four waves with v_mfma_f64_16x16x4_f64: the 15 upper 16x16 tiles of the 5 x 5 tile grid, 4 / 4 /
// 4 / 3 per wave, accumulators resident in registers for the whole item.  (The fp64 MFMA pipe runs at its full rate from
// ONE wave per SIMD and dependent accumulation costs nothing extra, but it IS the fp64 VALU datapath -- an MFMA wave and
// an FMA wave on one SIMD take the sum of their times: tools/fp64_calib.hip.)  Every entry is one sum in landmark
// order: deterministic, no partials, no float atomics.
// Operand / result layout of the instruction (cdna_hip_programming.md): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 * reg][col = lane & 15].
constexpr int SCHUR_THREADS = 256;
constexpr int SCHUR_BATCH = 21;                  // landmarks per batch: 21 x 12 slots = 252 producer lanes
constexpr int SCHUR_KB = 64;                     // 63 factor rows + one zero row = 16 MFMA steps of k = 4
constexpr int SCHUR_RS = 80;                     // row stride of Zm: 5 tiles; 80 % 32 = 16 keeps the half-wave reads conflict-free
constexpr int SCHUR_LDS_DOUBLES = SCHUR_KB * SCHUR_RS;     // 40 960 B
constexpr int SCHUR_ITEM_MAX = 128;              // landmarks per work item at most (ssba_layout.cpp: kItemMax)
constexpr int SCHUR_LDS_BYTES = (SCHUR_LDS_DOUBLES + 9 * SCHUR_ITEM_MAX) * 8;       // + M and M g_l of the item's landmarks: 50 176 B
typedef double schur_d4 __attribute__((ext_vector_type(4)));

// n_zero > 0: the first n_zero workgroups clear the block-tridiagonal reduced system (D and L of every super-block) that
// k_assemble_reduced fills next -- a memset launch less per iteration, hidden beside the Schur items.
// check_parts > 0 (single GPU, LM: launch_schur(.., check_in_schur)): one more work-group at the head of the grid does
// k_check's work (the sums of the linearisation partials, the projected-gradient norm over the poses, the convergence tests
// and the iteration bookkeeping).  Everything it reads was written by the linearisation launches, nothing it writes is read by
// the Schur items (radius, options and the termination flag apart -- a stale 0 there costs one wasted pass), so the
// 10 us dependent launch disappears inside this 75 us one.  k_assemble_reduced, which now runs AFTER the iteration
// counter was advanced, is told so (its `it0`).
__global__ __launch_bounds__(SCHUR_THREADS, 2) void k_schur_windows(Dev d, int n_zero, int check_parts) {
    // (r04: the FIRST work-group of the grid, not the last -- at C4 its sums over 15 625 partial entries and 10 000 poses take
    // longer than a Schur item, and as the last work-group it was the launch's tail: +80 us)
    if (check_parts > 0 && blockIdx.x == 0) { check_body(d, check_parts, true); return; }
    const int bid = (int)blockIdx.x - (check_parts > 0 ? 1 : 0);
    const State &st = *d.st;
    // the solver state is written by the previous launch on another XCD: its read is a ~2 us round trip.  It is tested
    // after the item's first operand reads have been issued, not before (a launch that returns at once measures 4.5 us
    // against 2.75 us for an empty kernel: that difference sits at the head of every kernel that tests the state first)
    const int dead = st.terminated | st.dl_reuse;
    if (bid < n_zero) {
        if (dead) return;
        // D and L of every super-block; of this rank's chain only in a partitioned solve (the rest is never assembled)
        const size_t b0 = d.part ? (size_t)d.chain0 * BD * BD : 0;
        const size_t n2 = (d.part ? (size_t)(d.chain1 - d.chain0 + 1) : (size_t)d.Nsb) * (BD * BD / 2);       // double2 per range
        double2 *zD = reinterpret_cast<double2 *>(d.xv + d.off_D + b0), *zL = reinterpret_cast<double2 *>(d.xv + d.off_L + b0);
        for (size_t i = (size_t)bid * SCHUR_THREADS + threadIdx.x; i < n2; i += (size_t)n_zero * SCHUR_THREADS) {
            zD[i] = make_double2(0.0, 0.0);
            zL[i] = make_double2(0.0, 0.0);
        }
        return;
    }
    extern __shared__ __align__(16) double schur_lds[];
    double *sZ = schur_lds;                              // [k][col]
    double *sM = schur_lds + SCHUR_LDS_DOUBLES;          // [c][landmark of the item]: M (6), u = M g_l (3) -- once per landmark, not once per (landmark, slot)
    const int item = bid - n_zero;
    const uint32_t win = d.slab_win[item];
    const int lb = (int)d.slab_lm_begin[item], le = (int)d.slab_lm_end[item];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool producer = t < SCHUR_BATCH * TW;
    const int li = t / TW, s = t - li * TW;
    // tiles (ti <= tj) of this wave; wave 3 has three
    const int ti0 = wv == 0 ? 0 : wv == 1 ? 0 : wv == 2 ? 1 : 3, tj0 = wv == 0 ? 0 : wv == 1 ? 4 : wv == 2 ? 4 : 3;
    const int ti1 = wv == 0 ? 0 : wv == 1 ? 1 : wv == 2 ? 2 : 3, tj1 = wv == 0 ? 1 : wv == 1 ? 1 : wv == 2 ? 2 : 4;
    const int ti2 = wv == 0 ? 0 : wv == 1 ? 1 : wv == 2 ? 2 : 4, tj2 = wv == 0 ? 2 : wv == 1 ? 2 : wv == 2 ? 3 : 4;
    const int ti3 = wv == 0 ? 0 : wv == 1 ? 1 : wv == 2 ? 2 : 4, tj3 = wv == 0 ? 3 : wv == 1 ? 3 : wv == 2 ? 4 : 4;
    const bool has3 = wv != 3;              // (wave 3's fourth product repeats its third: not stored)
    const bool a1x = wv == 0 || wv == 3;    // A operands: tile 0 row ti0, tiles 2 and 3 row ti3 (= ti2), tile 1 row ti1 = ti0 or ti3
    schur_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;

    // padding columns 73..79 and the zero row k = 63 stay zero for the whole item (column 72 is rewritten every batch)
    for (int e = t; e < SCHUR_KB * 8; e += SCHUR_THREADS) sZ[(e >> 3) * SCHUR_RS + 72 + (e & 7)] = 0.0;
    if (t < SCHUR_RS) sZ[(SCHUR_KB - 1) * SCHUR_RS + t] = 0.0;

    bool pose_ok = false;
    double T[12];
    if (producer) {
        const uint32_t k = d.win_pose[win * TW + s];
        pose_ok = (k != 0xFFFFFFFFu) && d.pose_free[k] >= 0;
        if (pose_ok) {
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = d.poses[(size_t)k * 12 + i];
        }
    }

    // Raw inputs of this lane's (landmark, slot) for the NEXT batch are fetched from HBM while the
    // matrix phase works on the current one (the loads stay in flight across the barrier and are only
    // waited for at the top of the next producer phase).
    struct Raw { double u, v, dd, p[3]; uint32_t mask; bool in_range; } raw;
    auto prefetch = [&](int l0) {
        const int l = l0 + li;
        raw.in_range = producer && l < le;
        raw.mask = 0;
        if (raw.in_range) {
            raw.mask = d.lm_mask[l];
            const size_t oi = (size_t)(l >> 6) * (TW * LMG) + (size_t)s * LMG + (l & 63);
            raw.u = d.ou[oi]; raw.v = d.ov[oi]; raw.dd = d.od[oi];
#pragma unroll
            for (int c = 0; c < 3; ++c) raw.p[c] = d.pts[(size_t)c * d.Lpad + l];
        }
    };
    prefetch(lb);
    // the damped 3 x 3 block factor of every landmark of the item, by one lane each (r04: it was formed by all twelve slot lanes of
    // a landmark in every batch -- a quarter of the producer's fp64 instructions, on the pipe the matrix instructions need)
    double lh[6], lsc[3], lg[3];
    const bool lm_lane = t < le - lb;
    if (lm_lane) {
        const int l = lb + t;
#pragma unroll
        for (int c = 0; c < 6; ++c) lh[c] = d.hll[(size_t)c * d.Lpad + l];
#pragma unroll
        for (int c = 0; c < 3; ++c) { lsc[c] = d.sl[(size_t)c * d.Lpad + l]; lg[c] = d.gl[(size_t)c * d.Lpad + l]; }
    }
    if (dead) return;
    if (lm_lane) {
        double dmp[3], m[6];
        const double hd[3] = {lh[0], lh[3], lh[5]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // LM diagonal in unscaled coordinates (landmark_damping)
            const double s2 = lsc[c] * lsc[c];
            dmp[c] = fmin(fmax(hd[c] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) * fast_rcp(damp_radius(st) * s2);
        }
        if (!chol3_inv_fast(lh, dmp, m)) {
            d.st->step_failed = 1;
#pragma unroll
            for (int c = 0; c < 6; ++c) m[c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) sM[c * SCHUR_ITEM_MAX + t] = m[c];
        sM[6 * SCHUR_ITEM_MAX + t] = m[0] * lg[0];
        sM[7 * SCHUR_ITEM_MAX + t] = m[1] * lg[0] + m[2] * lg[1];
        sM[8 * SCHUR_ITEM_MAX + t] = m[3] * lg[0] + m[4] * lg[1] + m[5] * lg[2];
    }
    __syncthreads();

    for (int l0 = lb; l0 < le; l0 += SCHUR_BATCH) {
        if (producer) {
            double z[18];       // [c][a]: three runs of six contiguous doubles in the k-major matrix
            double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            bool live = false;
            const int il = l0 - lb + li;        // landmark of the item (its factor: sM)
            if (raw.in_range) {
#pragma unroll
                for (int c = 0; c < 6; ++c) m[c] = sM[c * SCHUR_ITEM_MAX + il];
            }
            if (s == 0) {       // u = M g_l
#pragma unroll
                for (int c = 0; c < 3; ++c) sZ[(li * 3 + c) * SCHUR_RS + 72] = raw.in_range ? sM[(6 + c) * SCHUR_ITEM_MAX + il] : 0.0;
            }
            if (raw.in_range && pose_ok && ((raw.mask >> s) & 1u)) {
                live = true;
                obs_schur_factor(d, d.S, T, raw.p[0], raw.p[1], raw.p[2], raw.u, raw.v, raw.dd, m, z);       // Z = W M^T
            }
            if (!live) {
#pragma unroll
                for (int i = 0; i < 18; ++i) z[i] = 0.0;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double2 *dz = reinterpret_cast<double2 *>(sZ + (li * 3 + c) * SCHUR_RS + s * 6);     // 48-byte runs, 16-byte aligned
#pragma unroll
                for (int q = 0; q < 3; ++q) dz[q] = make_double2(z[6 * c + 2 * q], z[6 * c + 2 * q + 1]);
            }
        }
        __syncthreads();
        prefetch(l0 + SCHUR_BATCH);
        {
            // Branch-free and software-pipelined (r04: the rolled form waited for its six LDS reads in front of every k-step's
            // matrix instructions and branched around wave 3's missing fourth tile): the operands of k-step ks + 1 are requested
            // before the four instructions of k-step ks are issued; wave 3 multiplies its last tile twice (the second result is
            // never stored), so all waves run the same straight-line code.  Same-box A/B at C2: -3.4 us.  (One instantiation per
            // wave with compile-time tile lists -- what pays in k_wd_schur -- measured +5 us here.)
            const int kq = lane >> 4, i = lane & 15;
            const double *z0 = sZ + kq * SCHUR_RS + i;
            double ax = z0[16 * ti0], ay = z0[16 * ti3], b0 = z0[16 * tj0], b1 = z0[16 * tj1], b2 = z0[16 * tj2], b3 = z0[16 * tj3];
#pragma unroll
            for (int ks = 0; ks < SCHUR_KB / 4; ++ks) {
                double nax = 0.0, nay = 0.0, nb0 = 0.0, nb1 = 0.0, nb2 = 0.0, nb3 = 0.0;
                if (ks + 1 < SCHUR_KB / 4) {
                    const double *zr = z0 + 4 * (ks + 1) * SCHUR_RS;
                    nax = zr[16 * ti0]; nay = zr[16 * ti3]; nb0 = zr[16 * tj0]; nb1 = zr[16 * tj1]; nb2 = zr[16 * tj2]; nb3 = zr[16 * tj3];
                }
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1x ? ax : ay, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b2, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, b3, acc3, 0, 0, 0);
                ax = nax; ay = nay; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
            }
        }
        __syncthreads();
    }
    // one slab (78 blocks + 12 rhs vectors) per item.  Diagonal pair blocks are symmetric: the upper entry (row <= col,
    // always inside a computed tile) is written to both positions.
    auto store_tile = [&](const schur_d4 &acc, int ti, int tj) {
        const int col = 16 * tj + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + (lane >> 4) + 4 * r;
            if (row >= 72) continue;
            const int a = row / 6, ra = row - 6 * a;
            if (col == 72) { d.slab[(size_t)item * SLAB_DOUBLES + NPAIR * 36 + row] = acc[r]; continue; }
            if (col > 72 || col < row) continue;
            const int b = col / 6, cb = col - 6 * b;
            double *out = d.slab + (size_t)item * SLAB_DOUBLES + (size_t)(a * TW - (a * (a - 1)) / 2 + (b - a)) * 36;
            out[ra * 6 + cb] = acc[r];
            if (a == b) out[cb * 6 + ra] = acc[r];
        }
    };
    store_tile(acc0, ti0, tj0);
    store_tile(acc1, ti1, tj1);
    store_tile(acc2, ti2, tj2);
    if (has3) store_tile(acc3, ti3, tj3);
}

__device__ __forceinline__ int tri21(int r, int c) {   // packed upper index, r <= c
    return r * 6 - (r * (r - 1)) / 2 + (c - r);
}

// Gathers the slabs into the block-tridiagonal reduced system (pure stores, fixed
// summation order): one thread per (6x6 block, element), plus one thread per rhs entry.
// S = H_pp - sum slabs (pose damping is added after the exchange, in k_finish_reduced).
// fuse_finish (single GPU: nothing is exchanged between this kernel and the solve): the work of k_finish_reduced is done
// here -- Jacobi scale of the poses at iteration 0, LM damping on the diagonal, rhs = -reduced gradient, identity rows
// for the padding of the last super-block.
__global__ __launch_bounds__(256) void k_assemble_reduced(Dev d, int fuse_finish, int it0) {      // it0: value of st.iteration in the first pass (1 when k_check's work ran before this launch)
    const State &st = *d.st;
    const int dead = st.terminated | st.dl_reuse;        // tested once the first index reads are in flight (a cold read)
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_el = (size_t)d.n_sblk * 36;
    double *D0 = d.xv + d.off_D, *L0 = d.xv + d.off_L;
    if (gid < n_el) {
        const uint32_t blk = (uint32_t)(gid / 36);
        const int e = (int)(gid - (size_t)blk * 36);
        int r = e / 6, c = e - r * 6;
        const uint32_t fa = d.sblk_a[blk], fb = d.sblk_b[blk];
        const uint32_t ib0 = d.sblk_start[blk], ie0 = d.sblk_start[blk + 1];
        uint32_t cw0[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cw0[q] = ib0 + q < ie0 ? d.sblk_contrib[ib0 + q] : 0xFFFFFFFFu;
        if (dead) return;
        int er = r, ec = c;
        if (fa == fb && c > r) { er = c; ec = r; }   // read the lower triangle: exact symmetry
        // contributions in chunks of eight: all indices, then all values in flight, summed in list order (a rolled loop pays
        // two dependent memory round trips per contribution; a block of the band collects ~10 of them)
        double v = 0.0;
        const uint32_t ib = ib0, ie = ie0;
        for (uint32_t i0 = ib; i0 < ie; i0 += 8) {
            uint32_t cw[8];
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cw[q] = i0 == ib ? cw0[q] : (i0 + q < ie ? d.sblk_contrib[i0 + q] : 0xFFFFFFFFu);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                x[q] = cw[q] != 0xFFFFFFFFu ? d.slab[(size_t)(cw[q] / NPAIR) * SLAB_DOUBLES + (size_t)(cw[q] % NPAIR) * 36 + er * 6 + ec] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) v += x[q];
        }
        v = -v;
        if (fa == fb) {
            const int k = d.free_pose[fa];
            const double h = d.hpp[(size_t)k * 21 + tri21(min(r, c), max(r, c))];
            v += h;
            if (fuse_finish && r == c) {
                const size_t i = (size_t)fa * 6 + r;
                double sc;
                if (st.iteration == it0) { sc = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0; d.sp[i] = sc; }
                else sc = d.sp[i];
                const double s2 = sc * sc;
                v += fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
            }
        }
        const uint32_t Ia = fa / SBP, Ib = fb / SBP;
        const int row = (int)(fa - Ia * SBP) * 6 + r, col = (int)(fb - Ib * SBP) * 6 + c;
        if (Ia == Ib) {
            D0[(size_t)Ia * BD * BD + (size_t)row * BD + col] = v;
            if (fa != fb) D0[(size_t)Ia * BD * BD + (size_t)col * BD + row] = v;
        } else {   // Ib == Ia + 1: entry (row in block Ia, col in block Ib) = L[Ib]^T;
                   // even-indexed coupling blocks are stored transposed (ssba_bcr.hip); in a partitioned
                   // solve the index is the position in this rank's chain
            if ((Ib - (uint32_t)d.chain0) & 1u) L0[(size_t)Ib * BD * BD + (size_t)col * BD + row] = v;
            else L0[(size_t)Ib * BD * BD + (size_t)row * BD + col] = v;
        }
    } else if (dead) {
        return;
    } else if (gid < n_el + (size_t)d.nfree * 6) {
        const size_t i = gid - n_el;
        const uint32_t f = (uint32_t)(i / 6);
        const int c = (int)(i - (size_t)f * 6);
        const int k = d.free_pose[f];
        const double g = d.gp[(size_t)k * 6 + c];
        double v = g;
        const uint32_t jb = d.prow_start[f], je = d.prow_start[f + 1];
        for (uint32_t j0 = jb; j0 < je; j0 += 8) {
            uint32_t cw[8];
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cw[q] = j0 + q < je ? d.prow_contrib[j0 + q] : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                x[q] = cw[q] != 0xFFFFFFFFu ? d.slab[(size_t)(cw[q] / TW) * SLAB_DOUBLES + NPAIR * 36 + (cw[q] % TW) * 6 + c] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) v -= x[q];
        }
        d.xv[d.off_rhs + i] = fuse_finish ? -v : v;             // reduced gradient; negated here or in k_finish_reduced
        d.xv[d.off_gp + i] = g;
        d.xv[d.off_hdiag + i] = d.hpp[(size_t)k * 21 + tri21(c, c)];
    } else if (fuse_finish && gid < n_el + (size_t)d.nf_pad * 6) {       // padding rows of the last super-block: identity
        const size_t i = gid - n_el;
        const int f = (int)(i / 6), c = (int)(i - (size_t)f * 6), I = f / SBP, row = (f - I * SBP) * 6 + c;
        d.xv[d.off_D + (size_t)I * BD * BD + (size_t)row * BD + row] = 1.0;
        d.xv[d.off_rhs + i] = 0.0;
    } else if (fuse_finish && gid < n_el + (size_t)d.nf_pad * 6 + (size_t)d.nfree) {
        // one lane per free pose: its share of k_check's projected-gradient norm |x - Plus(x, -g)|_inf and of |x|^2 (an SE(3)
        // exponential per pose: 1 000 of them side by side here instead of on the 1 024 lanes of the one k_check work-group)
        const int f = (int)(gid - n_el - (size_t)d.nf_pad * 6), k = d.free_pose[f];
        const double *T = d.poses + (size_t)k * 12;
        double ng[6], Tn[12], gm = 0.0, xn = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) ng[c] = -d.gp[(size_t)k * 6 + c];
        se3_plus(T, ng, Tn);
#pragma unroll
        for (int c = 0; c < 12; ++c) { gm = fmax(gm, fabs(T[c] - Tn[c])); xn += T[c] * T[c]; }
        d.part_chk[2 * f] = gm;
        d.part_chk[2 * f + 1] = xn;
    }
}

// After the (optional) all-reduce: Jacobi scale of the poses at iteration 0, LM damping
// on the diagonal, rhs = -reduced gradient, identity rows for the padding.
__global__ __launch_bounds__(256) void k_finish_reduced(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (d.part)     // the separator vector (last iteration's sums: consumed) is cleared for k_sep_pack -- no memset launch
        for (size_t q = (size_t)i; q < d.sepv_count; q += (size_t)gridDim.x * 256) d.sepv[q] = 0.0;
    if (i >= d.nf_pad * 6) return;
    const int f = i / 6, c = i - f * 6;
    const int I = f / SBP, row = (f - I * SBP) * 6 + c;
    double *Dd = d.xv + d.off_D + (size_t)I * BD * BD + (size_t)row * BD + row;
    if (d.part) {
        if (I < d.chain0 || I > d.chain1) return;
        if ((d.pin0 && I == d.chain0) || (d.pin1 && I == d.chain1)) {
            // a shared chain end is a separator: the neighbouring rank contributes to it too, so it is damped after the exchange
            // (k_sep_finish); only the sign convention rhs = -gradient is applied here
            d.xv[d.off_rhs + i] = f < d.nfree ? -d.xv[d.off_rhs + i] : 0.0;
            return;
        }
    }
    if (f < d.nchain) {        // (the poses of a closure border keep identity rows in the chain system)
        const double h = d.xv[d.off_hdiag + i];
        if (st.iteration == 0) d.sp[i] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
        const double s = d.sp[i], s2 = s * s;
        *Dd += fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
        d.xv[d.off_rhs + i] = -d.xv[d.off_rhs + i];
    } else {
        *Dd = 1.0;
        d.xv[d.off_rhs + i] = 0.0;
    }
}

// Closure border (ssba_finalize): the blocks of the reduced system that involve a border pose, gathered from the Schur
// slabs like k_assemble_reduced does for the chain: S_pb (chain pose rows x 6 columns per border pose) = - sum slabs,
// S_bb = H_pp(border) - sum slabs, and per border pose the reduced gradient, gradient, diag H_pp and (iteration 0) the
// Jacobi scale.  One thread per (block, element); contribution codes carry bit 31 when the slab block is the transpose.
__global__ __launch_bounds__(256) void k_cb_assemble(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int blk = (int)(gid / 36);
    if (blk >= d.n_cb) return;
    const int e = (int)(gid - (size_t)blk * 36), r = e / 6, c = e - 6 * r;
    const uint32_t fa = d.cb_a[blk], fb = d.cb_b[blk];
    const uint32_t ib = d.cb_start[blk], ie = d.cb_start[blk + 1];
    if (fb == 0xFFFFFFFFu) {        // right-hand side and diagonal terms of border pose fa
        if (e >= 6) return;
        const int k = d.free_pose[fa], col = ((int)fa - d.nchain) * 6 + e;
        const double g = d.gp[(size_t)k * 6 + e], h = d.hpp[(size_t)k * 21 + tri21(e, e)];
        double v = g;
        for (uint32_t i = ib; i < ie; ++i) {
            const uint32_t cw = d.cb_contrib[i];
            v -= d.slab[(size_t)(cw / TW) * SLAB_DOUBLES + NPAIR * 36 + (cw % TW) * 6 + e];
        }
        d.bsys[BS_RHS + col] = v;
        d.bsys[BS_G + col] = g;
        d.bsys[BS_H + col] = h;
        if (st.iteration == 0) d.bsys[BS_S + col] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
        return;
    }
    double v = 0.0;
    for (uint32_t i = ib; i < ie; ++i) {
        const uint32_t cw = d.cb_contrib[i], code = cw & 0x7FFFFFFFu;
        v += d.slab[(size_t)(code / NPAIR) * SLAB_DOUBLES + (size_t)(code % NPAIR) * 36 + ((cw >> 31) ? c * 6 + r : r * 6 + c)];
    }
    const int col = ((int)fb - d.nchain) * 6 + c;
    if ((int)fa < d.nchain) {
        d.Spb[((size_t)fa * 6 + r) * NBP + col] = -v;
    } else {
        const int row = ((int)fa - d.nchain) * 6 + r;
        double sv = -v;
        if (fa == fb) sv += d.hpp[(size_t)d.free_pose[fa] * 21 + tri21(r < c ? r : c, r < c ? c : r)];
        d.bsys[BS_SBB + row * NBP + col] = sv;
        if (fa != fb) d.bsys[BS_SBB + col * NBP + row] = sv;
    }
}

__global__ __launch_bounds__(1024) void k_check(Dev d, int fused_parts) {      // 1024 lanes: one pose each at C2 (the exponential map is a long dependent chain)
    check_body(d, fused_parts, false);
}


__global__ __launch_bounds__(256) void k_best(Dev d) {
    const State &st = *d.st;
    if (st.copy_best != st.check_count) return;       // set by the k_check just before; a later k_check moves the count on
    const size_t n = (size_t)d.P * 12 > (size_t)d.Lpad * 3 ? (size_t)d.P * 12 : (size_t)d.Lpad * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {   // few blocks: most launches only test the flag
        if (i < (size_t)d.P * 12) d.best_poses[i] = d.poses[i];
        if (i < (size_t)d.Lpad * 3) {
            d.best_pts[i] = d.pts[i];
            if (d.phong) d.best_nrm[i] = d.nrm[i];
        }
        if (d.phong && i < (size_t)d.nsh) d.best_sh[i] = d.sh[i];
    }
}

// candidate poses = Plus(x, delta_p)  [Evaluator::Plus with SE3Perturbation]
// fuse_best: also does k_best's share for the poses (x -> best when the k_check of this iteration saw the cost improve;
// like k_best, before the termination test -- the improving iterate may be the converged one); -1: the trial point of a
// device-side line-search round (bounds), a no-op unless a search is under way
// With free shared blocks (lighting terms) the launch has one more work-group, which moves those (ph_border_update_block).
__global__ __launch_bounds__(256) void k_pose_update(Dev d, int fuse_best) {
    if (blockIdx.x == (unsigned)d.n_pose_blocks) {
        __shared__ double sdf[64];
        ph_border_update_block(d, fuse_best < 0 ? 1 : 0, sdf);
        return;
    }
    const State &st = *d.st;
    __shared__ double sm[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    // operand reads first, the solver state (a cold read: see k_schur_windows) is tested with them in flight
    double Tk[12];
    int fk = -1;
    if (k < d.P) {
        fk = d.pose_free[k];
#pragma unroll
        for (int c = 0; c < 12; ++c) Tk[c] = d.poses[(size_t)k * 12 + c];
    }
    if (fuse_best > 0 && st.copy_best == st.check_count && k < d.P) {
#pragma unroll
        for (int c = 0; c < 12; ++c) d.best_poses[(size_t)k * 12 + c] = Tk[c];
    }
    if (st.terminated || (fuse_best < 0 && !st.ls_active)) return;       // fuse_best < 0: a round of the device-side line search
    double dn = 0.0, nonfinite = 0.0, pf_cc = 0.0, pf_mcc = 0.0;
    if (k < d.P) {
        const int f = fk;
        const double *T = Tk;
        double *C = d.cand_poses + (size_t)k * 12;
        if (f >= 0 && !st.step_failed) {
            double eps[6], Tn[12];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                eps[c] = st.opt.strategy ? st.beta * d.x0[(size_t)f * 6 + c] + st.gamma * d.vp[(size_t)k * 6 + c]
                                         : d.x0[(size_t)f * 6 + c];
                eps[c] *= st.ls_alpha;      // 1 except inside the projected line search (bounds)
                if (!isfinite(eps[c])) nonfinite = 1.0;
            }
            se3_plus(T, eps, Tn);
            if (d.n_pf)     // pose-only residual blocks: model cost change at x, cost at the candidate
                for (uint32_t e = d.pf_start[k]; e < d.pf_start[k + 1]; ++e) {
                    double r[6], J[36], rc[6];
                    int dim;
                    // relative-pose half entry: the other pose, its step and its candidate (computed here: the other
                    // thread's candidate may not be written yet)
                    const int other = pf_other_pose(d, (int)e);
                    const double *To = nullptr;
                    double eo[6] = {0, 0, 0, 0, 0, 0}, Tno[12];
                    if (other >= 0) {
                        To = d.poses + (size_t)other * 12;
                        const int fo = d.pose_free[other];
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            if (fo >= 0)
                                eo[c] = st.ls_alpha * (st.opt.strategy ? st.beta * d.x0[(size_t)fo * 6 + c] + st.gamma * d.vp[(size_t)other * 6 + c]
                                                                        : d.x0[(size_t)fo * 6 + c]);
                        }
                        if (fo >= 0) se3_plus(To, eo, Tno);
                        else for (int c = 0; c < 12; ++c) Tno[c] = To[c];
                    }
                    pf_evaluate(d, (int)e, T, To, r, J, &dim);
                    double jd[6];
                    for (int m = 0; m < dim; ++m) {
                        jd[m] = 0.0;
                        for (int c = 0; c < 6; ++c) jd[m] += J[6 * m + c] * eps[c];
                        pf_mcc -= jd[m] * (r[m] + 0.5 * jd[m]);
                    }
                    const int partner = pf_partner(d, (int)e);
                    if (d.pf_type[e] == 2 && partner >= 0) {      // (J_1 d_1 + J_2 d_2) is one row vector: the cross term, once
                        double r2[6], J2[36];
                        pf_evaluate(d, partner, To, T, r2, J2, &dim);
                        for (int m = 0; m < 6; ++m) {
                            double jo = 0.0;
                            for (int c = 0; c < 6; ++c) jo += J2[6 * m + c] * eo[c];
                            pf_mcc -= jd[m] * jo;
                        }
                    }
                    if (pf_counts_cost(d, (int)e)) pf_cc += pf_evaluate(d, (int)e, Tn, other >= 0 ? Tno : nullptr, rc, nullptr, &dim);
                }
            // partitioned solve: the separator poses are updated by both neighbouring ranks, counted once
            const bool owned = !d.part || (f >= (d.rank == 0 ? d.chain0 : d.chain0 + 1) * SBP && f < (d.chain1 + 1) * SBP);
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                C[c] = Tn[c];
                const double df = Tn[c] - T[c];
                if (owned) dn += df * df;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 12; ++c) C[c] = T[c];
        }
    }
    const double a = block_sum(dn, sm);
    const double b = block_sum(nonfinite, sm);
    const double c2 = block_sum(pf_cc, sm), c3 = block_sum(pf_mcc, sm);
    if (threadIdx.x == 0) {
        d.part_pose[blockIdx.x * NPP] = a;
        d.part_pose[blockIdx.x * NPP + 1] = b;
        d.part_pose[blockIdx.x * NPP + 2] = c2;
        d.part_pose[blockIdx.x * NPP + 3] = c3;
    }
}

// One lane per landmark: back-substitution delta_l = -C^-1 (g_l + sum_s W_s^T delta_p,s),
// candidate point, model cost change -(J d)^T (r + J d / 2) and candidate cost, all in
// one pass over the landmark's observations (second sweep hits L1/L2).
template <bool DN> __global__ __launch_bounds__(256) void k_backsub_eval(Dev d) {
    const State &st = *d.st;
    if (st.terminated) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double ccost = 0.0, mcc = 0.0, dn = 0.0, nonfinite = 0.0;
    const double px = d.pts[l], py = d.pts[(size_t)d.Lpad + l], pz = d.pts[2 * (size_t)d.Lpad + l];
    double nx = px, ny = py, nz = pz;
    if (mask && !st.step_failed) {
        const LmObs<DN> ob(d, l, mask);
        const double gl[3] = {d.gl[l], d.gl[(size_t)d.Lpad + l], d.gl[2 * (size_t)d.Lpad + l]};
        double tt[3] = {gl[0], gl[1], gl[2]};
        // With e = J_p delta_p of an observation the model cost change -(J d)^T (r + J d / 2) of this landmark's
        // observations is  -(sum e.r + dl.g_l) - (sum e.e + 2 dl.(tt - g_l) + dl^T H_ll dl) / 2 :
        // one linearisation pass instead of two (H_ll and g_l are those of this linearisation point)
        double er = 0.0, ee = 0.0;
        for (int s = 0; s < ob.count(); ++s) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            const int f = d.pose_free[k];
            if (f < 0) continue;
            const double *T = d.poses + (size_t)k * 12;
            ObsLin o;
            double Sk[9];
            ob.stiffness(d, s, Sk);
            obs_linearize_S(d, Sk, T, px, py, pz, ob.u(d, s), ob.v(d, s), ob.dd(d, s), o);
            const double *dp = d.x0 + (size_t)f * 6;      // (as in k_backsub_eval_w: no Jacobians formed)
            double jd[3], y[3];
            pose_step_rows(o, dp, jd);
#pragma unroll
            for (int i = 0; i < 3; ++i) { er += jd[i] * o.r[i]; ee += jd[i] * jd[i]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) y[c] = o.A[c] * jd[0] + o.A[3 + c] * jd[1] + o.A[6 + c] * jd[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) tt[c] += T[3 + c] * y[0] + T[6 + c] * y[1] + T[9 + c] * y[2];
        }
        double h[6], dmp[3], Ci[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
        landmark_damping(d, st, l, h, dmp);
        double dl[3] = {0, 0, 0};
        if (inv3_spd(h, dmp, Ci)) {
            dl[0] = -(Ci[0] * tt[0] + Ci[1] * tt[1] + Ci[2] * tt[2]);
            dl[1] = -(Ci[1] * tt[0] + Ci[3] * tt[1] + Ci[4] * tt[2]);
            dl[2] = -(Ci[2] * tt[0] + Ci[4] * tt[1] + Ci[5] * tt[2]);
        } else {
            nonfinite = 1.0;
        }
        if (!isfinite(dl[0]) || !isfinite(dl[1]) || !isfinite(dl[2])) nonfinite = 1.0;
        nx = px + dl[0]; ny = py + dl[1]; nz = pz + dl[2];
        dn = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2];
        {
            const double hd0 = h[0] * dl[0] + h[1] * dl[1] + h[2] * dl[2], hd1 = h[1] * dl[0] + h[3] * dl[1] + h[4] * dl[2],
                         hd2 = h[2] * dl[0] + h[4] * dl[1] + h[5] * dl[2];
            const double dg = dl[0] * gl[0] + dl[1] * gl[1] + dl[2] * gl[2];
            const double dt = dl[0] * (tt[0] - gl[0]) + dl[1] * (tt[1] - gl[1]) + dl[2] * (tt[2] - gl[2]);
            mcc = -(er + dg) - 0.5 * (ee + 2.0 * dt + (dl[0] * hd0 + dl[1] * hd1 + dl[2] * hd2));
        }
        for (int s = 0; s < ob.count(); ++s) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            double Sk[9];
            ob.stiffness(d, s, Sk);
            ccost += obs_cost_S(d, Sk, d.cand_poses + (size_t)k * 12, nx, ny, nz, ob.u(d, s), ob.v(d, s), ob.dd(d, s));
        }
    }
    d.cand_pts[l] = nx;
    d.cand_pts[(size_t)d.Lpad + l] = ny;
    d.cand_pts[2 * (size_t)d.Lpad + l] = nz;
    const double a = block_sum(ccost, sm);
    const double b = block_sum(mcc, sm);
    const double c = block_sum(dn, sm);
    const double e = block_sum(nonfinite, sm);
    if (threadIdx.x == 0) {
        d.part_eval[blockIdx.x * 4 + 0] = a;
        d.part_eval[blockIdx.x * 4 + 1] = b;
        d.part_eval[blockIdx.x * 4 + 2] = c;
        d.part_eval[blockIdx.x * 4 + 3] = e;
    }
}


// The body of the Ceres trust-region loop after the candidate evaluation: step validity,
// parameter / function tolerance, step quality, accept / reject, radius update.
// n_eval_parts > 0 (single GPU, no exchange between the evaluation and the decision): the sums of k_reduce_eval are
// formed here, one launch less per iteration.
__device__ __forceinline__ void decide_body(Dev &d, State &st, int n_eval_parts, int n_pose_parts) {
    // (r04: four partial entries in flight per lane -- the rolled loop made seven dependent round trips at C2 -- and the eight sums
    // of this kernel in ONE block reduction: it was 8.5 us of dependent launch per iteration)
    __shared__ double sm[8 * 4];
    double red[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // e0..e3 | a, b, pcc, pmc
    if (n_eval_parts > 0) {
        for (int i0 = threadIdx.x; i0 < n_eval_parts; i0 += 4 * 256) {
            double4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 256 * q;
                x[q] = i < n_eval_parts ? reinterpret_cast<const double4 *>(d.part_eval)[i] : make_double4(0.0, 0.0, 0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { red[0] += x[q].x; red[1] += x[q].y; red[2] += x[q].z; red[3] += x[q].w; }
        }
    }
    for (int i = threadIdx.x; i < n_pose_parts + (d.nb ? 1 : 0); i += 256) {   // last entry: border of shared blocks
        red[4] += d.part_pose[i * NPP];
        red[5] += d.part_pose[i * NPP + 1];
        if (d.n_pf && i < n_pose_parts) { red[6] += d.part_pose[i * NPP + 2]; red[7] += d.part_pose[i * NPP + 3]; }
    }
    block_sums(red, sm);
    if (n_eval_parts > 0 && threadIdx.x == 0) { d.scal2[0] = red[0]; d.scal2[1] = red[1]; d.scal2[2] = red[2]; d.scal2[3] = red[3]; }
    double a = red[4], b = red[5], pcc = red[6], pmc = red[7];
    // (the solver state is a cold read at the head of a launch: tested here, with the partial sums already formed)
    if (threadIdx.x != 0 || st.terminated) return;
    if (d.part) { a = 0.0; b = 0.0; }     // already in scal2 (k_reduce_eval(.., add_pose)), summed over ranks
    const Options &o = st.opt;
    const double candidate_cost_raw = d.scal2[0] + pcc;      // + unary pose residual blocks
    const double mcc = o.strategy ? st.dl_mcc : d.scal2[1] + pmc;      // dogleg: from the sums of its model, all residual rows included
    const double step_norm = sqrt(d.scal2[2] + a);
    const bool finite_step = (d.scal2[3] + b) == 0.0 && !st.step_failed;
    st.model_cost_change = mcc;
    // ComputeTrustRegionStep: LINEAR_SOLVER_FAILURE or model_cost_change <= 0 -> invalid
    const bool step_is_valid = finite_step && (mcc > 0.0);
    st.step_failed = 0;
    if (!step_is_valid) {
        // HandleInvalidStep
        if (++st.num_invalid >= o.max_invalid && !o.ignore_convergence) {
            st.terminated = 1; st.termination_type = 2;   // FAILURE
        }
        if (o.strategy) {            // DoglegStrategy::StepIsInvalid
            st.mu *= 10.0; st.dl_reuse = 0;
        } else {
            st.radius /= st.decrease_factor;
            st.decrease_factor *= 2.0;
        }
        ++st.num_unsuccessful;
        log_push(d, st, st.x_cost, 0.0, 0.0, 0.0, 0);
        return;
    }
    st.num_invalid = 0;
    const double candidate_cost = isfinite(candidate_cost_raw) ? candidate_cost_raw : DBL_MAX;
    st.candidate_cost = candidate_cost;
    st.step_norm = step_norm;
    st.cost_change = st.x_cost - candidate_cost;
    if (!o.ignore_convergence) {
        // ParameterToleranceReached
        if (step_norm <= o.parameter_tolerance * (st.x_norm + o.parameter_tolerance)) {
            st.terminated = 1; st.termination_type = 0; return;
        }
        // FunctionToleranceReached
        if (fabs(st.cost_change) <= o.function_tolerance * st.x_cost) {
            st.terminated = 1; st.termination_type = 0; return;
        }
    }
    // TrustRegionStepEvaluator::StepQuality
    const double rd0 = (st.se_current - candidate_cost) / mcc;
    const double rd1 = (st.se_reference - candidate_cost) / (st.se_acc_ref + mcc);
    const double rd = rd0 > rd1 ? rd0 : rd1;
    st.relative_decrease = rd;
    if (rd > o.min_relative_decrease) {
        // HandleSuccessfulStep (re-linearisation happens at the top of the next iteration)
        st.accepted = 1;
        st.last_successful = 1;
        st.need_linearize = 1;
        ++st.num_successful;
        if (o.strategy) {
            // DoglegStrategy::StepAccepted
            if (rd < 0.25) st.radius *= 0.5;
            if (rd > 0.75) st.radius = fmax(st.radius, 3.0 * st.dl_step_norm);
            st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
            st.dl_reuse = 0;
        } else {
            // LevenbergMarquardtStrategy::StepAccepted
            const double tq = 2.0 * rd - 1.0;
            st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - tq * tq * tq);
            st.radius = fmin(o.max_radius, st.radius);
            st.decrease_factor = 2.0;
        }
        // TrustRegionStepEvaluator::StepAccepted
        st.se_current = candidate_cost;
        st.se_acc_cand += mcc;
        st.se_acc_ref += mcc;
        if (st.se_current < st.se_minimum) {
            st.se_minimum = st.se_current;
            st.se_num_nonmono = 0;
            st.se_candidate = st.se_current;
            st.se_acc_cand = 0.0;
        } else {
            ++st.se_num_nonmono;
            if (st.se_current > st.se_candidate) {
                st.se_candidate = st.se_current;
                st.se_acc_cand = 0.0;
            }
        }
        if (st.se_num_nonmono == o.max_nonmono) {
            st.se_reference = st.se_candidate;
            st.se_acc_ref = st.se_acc_cand;
        }
    } else {
        // HandleUnsuccessfulStep: StepRejected
        if (o.strategy) {
            st.radius *= 0.5; st.dl_reuse = 1;
        } else {
            st.radius /= st.decrease_factor;
            st.decrease_factor *= 2.0;
        }
        ++st.num_unsuccessful;
        log_push(d, st, candidate_cost, st.cost_change, step_norm, rd, 0);
    }
}
__global__ __launch_bounds__(256) void k_decide(Dev d, int n_eval_parts, int n_pose_parts) {
    State &st = *d.st;
    decide_body(d, st, n_eval_parts, n_pose_parts);       // tests st.terminated itself, after its partial sums are read
}

// The same pass in the window layout with SP lanes per landmark (see k_linearize_landmarks_w): wave w takes the slots
// w, w + SP, ...; the partial sums of W^T delta_p and of the model-cost terms meet in LDS, every lane then forms
// delta_l itself (fixed order, so the SP lanes of a landmark hold identical candidates) and evaluates the candidate
// cost of its own slots.
// fuse_best: also does k_best's share for the points (see k_pose_update)
template <bool DN, int SP> __global__ __launch_bounds__(64 * SP) void k_backsub_eval_w(Dev d, int fuse_best) {
    const State &st = *d.st;
    __shared__ double sm4[4 * SP];
    __shared__ double red[SP][5][LMG];
    const int w = threadIdx.x >> 6, li = threadIdx.x & 63;
    const int l = blockIdx.x * LMG + li;
    // operand reads first, the solver state (a cold read: see k_schur_windows) is tested with them in flight
    const uint32_t mask = d.lm_mask[l];
    const double px = d.pts[l], py = d.pts[(size_t)d.Lpad + l], pz = d.pts[2 * (size_t)d.Lpad + l];
    if (fuse_best && st.copy_best == st.check_count) {
        if (w == 0) { d.best_pts[l] = px; d.best_pts[(size_t)d.Lpad + l] = py; d.best_pts[2 * (size_t)d.Lpad + l] = pz; }
        // fuse_best == 2: the poses too (the reduced solve updated them itself, and it does nothing once the solver has
        // terminated -- the improving iterate may be the converged one)
        if (fuse_best == 2)       // (grid-stride: few landmarks and many poses must still be covered)
            for (size_t gi = (size_t)blockIdx.x * (64 * SP) + threadIdx.x; gi < (size_t)d.P * 12; gi += (size_t)gridDim.x * (64 * SP))
                d.best_poses[gi] = d.poses[gi];
    }
    if (st.terminated) return;
    double ccost = 0.0, mcc = 0.0, dn = 0.0, nonfinite = 0.0;
    double nx = px, ny = py, nz = pz;
    const bool act = mask && !st.step_failed;
    const LmObs<DN> ob(d, l, mask);
    double tp[3] = {0.0, 0.0, 0.0}, er = 0.0, ee = 0.0;
    // Window layout: the pose index of every slot of this lane and its free index are requested up front, and the operands of slot
    // q + 1 before slot q is worked on (r04, from the ISA: the rolled loop made three dependent round trips per slot -- pose index,
    // free index, pose / step / observation -- 18 for the six slots of a lane, and the second sweep twelve more)
    constexpr int NS = DN ? 1 : TW / SP;
    uint32_t kk[NS];
    int ff[NS];
    if (!DN) {
#pragma unroll
        for (int q = 0; q < NS; ++q) kk[q] = (act && ob.has(w + q * SP)) ? ob.pose(d, w + q * SP) : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < NS; ++q) ff[q] = kk[q] != 0xFFFFFFFFu ? d.pose_free[kk[q]] : -1;
    }
    if (!DN && act) {
        struct In { double T[12], u, v, dd, dp[6]; };
        auto fetch = [&](int q, In &in) {
            if (ff[q] < 0) return;
            const int s = w + q * SP;
            const double *T = d.poses + (size_t)kk[q] * 12, *dp = d.x0 + (size_t)ff[q] * 6;
#pragma unroll
            for (int c = 0; c < 12; ++c) in.T[c] = T[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) in.dp[c] = dp[c];
            in.u = ob.u(d, s); in.v = ob.v(d, s); in.dd = ob.dd(d, s);
        };
        In cur, nxt;
        fetch(0, cur);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (q + 1 < NS) fetch(q + 1, nxt);
            if (ff[q] >= 0) {
                ObsLin o;
                obs_linearize(d, cur.T, px, py, pz, cur.u, cur.v, cur.dd, o);
                double jd[3], y[3];
                pose_step_rows(o, cur.dp, jd);
#pragma unroll
                for (int i = 0; i < 3; ++i) { er += jd[i] * o.r[i]; ee += jd[i] * jd[i]; }
#pragma unroll
                for (int c = 0; c < 3; ++c) y[c] = o.A[c] * jd[0] + o.A[3 + c] * jd[1] + o.A[6 + c] * jd[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) tp[c] += cur.T[3 + c] * y[0] + cur.T[6 + c] * y[1] + cur.T[9 + c] * y[2];
            }
            if (q + 1 < NS) cur = nxt;
        }
    }
    if (DN && act) {
        for (int s = w; s < ob.count(); s += SP) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            const int f = d.pose_free[k];
            if (f < 0) continue;
            const double *T = d.poses + (size_t)k * 12;
            ObsLin o;
            if (DN) {
                double Sk[9];
                ob.stiffness(d, s, Sk);
                obs_linearize_S(d, Sk, T, px, py, pz, ob.u(d, s), ob.v(d, s), ob.dd(d, s), o);
            } else {
                obs_linearize(d, T, px, py, pz, ob.u(d, s), ob.v(d, s), ob.dd(d, s), o);
            }
            // J_p dp = A (dp_t + dp_r x q)  and  J_l^T (J_p dp) = R^T (A^T (J_p dp)):  33 multiply-adds instead of the 72 through
            // J_p = A [I | -q^] and J_l = A R (r04: at C4 this pass is bound by its fp64 instruction count, not by HBM)
            const double *dp = d.x0 + (size_t)f * 6;
            double jd[3], y[3];
            pose_step_rows(o, dp, jd);
#pragma unroll
            for (int i = 0; i < 3; ++i) { er += jd[i] * o.r[i]; ee += jd[i] * jd[i]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) y[c] = o.A[c] * jd[0] + o.A[3 + c] * jd[1] + o.A[6 + c] * jd[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) tp[c] += T[3 + c] * y[0] + T[6 + c] * y[1] + T[9 + c] * y[2];
        }
    }
    red[w][0][li] = tp[0]; red[w][1][li] = tp[1]; red[w][2][li] = tp[2]; red[w][3][li] = er; red[w][4][li] = ee;
    __syncthreads();
    if (act) {
        const double gl[3] = {d.gl[l], d.gl[(size_t)d.Lpad + l], d.gl[2 * (size_t)d.Lpad + l]};
        double tt[3] = {gl[0], gl[1], gl[2]};
        er = 0.0; ee = 0.0;
#pragma unroll
        for (int q = 0; q < SP; ++q) {
            tt[0] += red[q][0][li]; tt[1] += red[q][1][li]; tt[2] += red[q][2][li];
            er += red[q][3][li]; ee += red[q][4][li];
        }
        double h[6], dmp[3], Ci[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
        landmark_damping(d, st, l, h, dmp);
        double dl[3] = {0, 0, 0};
        if (inv3_spd(h, dmp, Ci)) {
            dl[0] = -(Ci[0] * tt[0] + Ci[1] * tt[1] + Ci[2] * tt[2]);
            dl[1] = -(Ci[1] * tt[0] + Ci[3] * tt[1] + Ci[4] * tt[2]);
            dl[2] = -(Ci[2] * tt[0] + Ci[4] * tt[1] + Ci[5] * tt[2]);
        } else if (w == 0) {
            nonfinite = 1.0;
        }
        if (w == 0 && (!isfinite(dl[0]) || !isfinite(dl[1]) || !isfinite(dl[2]))) nonfinite = 1.0;
        nx = px + dl[0]; ny = py + dl[1]; nz = pz + dl[2];
        if (w == 0) {
            dn = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2];
            const double hd0 = h[0] * dl[0] + h[1] * dl[1] + h[2] * dl[2], hd1 = h[1] * dl[0] + h[3] * dl[1] + h[4] * dl[2],
                         hd2 = h[2] * dl[0] + h[4] * dl[1] + h[5] * dl[2];
            const double dg = dl[0] * gl[0] + dl[1] * gl[1] + dl[2] * gl[2];
            const double dt = dl[0] * (tt[0] - gl[0]) + dl[1] * (tt[1] - gl[1]) + dl[2] * (tt[2] - gl[2]);
            mcc = -(er + dg) - 0.5 * (ee + 2.0 * dt + (dl[0] * hd0 + dl[1] * hd1 + dl[2] * hd2));
        }
        if (!DN) {      // candidate cost of this lane's slots: pose indices from above, operands one slot ahead
            struct In2 { double T[12], u, v, dd; };
            auto fetch2 = [&](int q, In2 &in) {
                if (kk[q] == 0xFFFFFFFFu) return;
                const int s = w + q * SP;
                const double *T = d.cand_poses + (size_t)kk[q] * 12;
#pragma unroll
                for (int c = 0; c < 12; ++c) in.T[c] = T[c];
                in.u = ob.u(d, s); in.v = ob.v(d, s); in.dd = ob.dd(d, s);
            };
            In2 cur, nxt;
            fetch2(0, cur);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (q + 1 < NS) fetch2(q + 1, nxt);
                if (kk[q] != 0xFFFFFFFFu) ccost += obs_cost(d, cur.T, nx, ny, nz, cur.u, cur.v, cur.dd);
                if (q + 1 < NS) cur = nxt;
            }
        }
        for (int s = w; DN && s < ob.count(); s += SP) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            double Sk[9];
            ob.stiffness(d, s, Sk);
            ccost += obs_cost_S(d, Sk, d.cand_poses + (size_t)k * 12, nx, ny, nz, ob.u(d, s), ob.v(d, s), ob.dd(d, s));
        }
    }
    if (w == 0) {
        d.cand_pts[l] = nx;
        d.cand_pts[(size_t)d.Lpad + l] = ny;
        d.cand_pts[2 * (size_t)d.Lpad + l] = nz;
    }
    double r4[4] = {ccost, mcc, dn, nonfinite};       // one block reduction for the four sums
    block_sums(r4, sm4);
    if (threadIdx.x == 0) reinterpret_cast<double4 *>(d.part_eval)[blockIdx.x] = make_double4(r4[0], r4[1], r4[2], r4[3]);
}


// ------------------------------------------------------------------- dogleg ---
// TRADITIONAL_DOGLEG [Ceres 1.x dogleg_strategy.cc] in unscaled coordinates.  With the Jacobi
// scale s and D^2 = clamp(s^2 diag(J^T J)):  gradient_ = s g / D,  Gauss-Newton step (D-scaled) =
// D delta_gn / s,  Cauchy step length alpha = |gradient_|^2 / |J v|^2 with v = s^2 g / D^2, and the
// step returned to the minimiser is  delta = beta * delta_gn + gamma * v.

// per pose: v_p and the pose parts of |gradient_|^2, |gn|^2, gradient_.gn
// (free shared blocks: one more work-group, whose first lane does the border part -- ph_dogleg_border_lane)
__global__ __launch_bounds__(256) void k_dogleg_vec(Dev d) {
    if (blockIdx.x == (unsigned)d.n_pose_blocks) {
        if (threadIdx.x == 0) ph_dogleg_border_lane(d);
        return;
    }
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double sm[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    double gsq = 0.0, nsq = 0.0, dot = 0.0, pjv = 0.0, pjg = 0.0, pvg = 0.0;
    if (k < d.P) {
        const int f = d.pose_free[k];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v = 0.0;
            if (f >= 0) {
                const double s = d.sp[(size_t)f * 6 + c], g = d.xv[d.off_gp + (size_t)f * 6 + c];
                const double h = d.xv[d.off_hdiag + (size_t)f * 6 + c], gn = d.x0[(size_t)f * 6 + c];
                const double s2 = s * s, D2 = fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag);
                v = s2 * g / D2;
                gsq += s2 * g * g / D2;
                nsq += D2 * gn * gn / s2;
                dot += g * gn;
            }
            d.vp[(size_t)k * 6 + c] = v;
        }
        if (d.n_pf && f >= 0)      // rows of the pose-only residual blocks in |J v|^2, |J gn|^2, (J v).(J gn)
            for (uint32_t e = d.pf_start[k]; e < d.pf_start[k + 1]; ++e) {
                double r[6], J[36], jv[6], jg[6];
                int dim;
                const int other = pf_other_pose(d, (int)e), partner = pf_partner(d, (int)e);
                const double *To = other >= 0 ? d.poses + (size_t)other * 12 : nullptr;
                pf_evaluate(d, (int)e, d.poses + (size_t)k * 12, To, r, J, &dim);
                for (int m = 0; m < dim; ++m) {
                    jv[m] = 0.0; jg[m] = 0.0;
                    for (int c = 0; c < 6; ++c) { jv[m] += J[6 * m + c] * d.vp[(size_t)k * 6 + c]; jg[m] += J[6 * m + c] * d.x0[(size_t)f * 6 + c]; }
                    pjv += jv[m] * jv[m]; pjg += jg[m] * jg[m]; pvg += jv[m] * jg[m];
                }
                if (d.pf_type[e] == 2 && partner >= 0) {      // cross terms of the two halves of a relative-pose block
                    const int fo = d.pose_free[other];
                    double r2[6], J2[36], vo[6], go[6];
                    pf_evaluate(d, partner, To, d.poses + (size_t)k * 12, r2, J2, &dim);
                    for (int c = 0; c < 6; ++c) {      // v of the other pose, recomputed (its thread may not have stored it yet)
                        const double so = d.sp[(size_t)fo * 6 + c], go_ = d.xv[d.off_gp + (size_t)fo * 6 + c], ho = d.xv[d.off_hdiag + (size_t)fo * 6 + c];
                        const double D2 = fmin(fmax(ho * so * so, st.opt.min_lm_diag), st.opt.max_lm_diag);
                        vo[c] = so * so * go_ / D2;
                        go[c] = d.x0[(size_t)fo * 6 + c];
                    }
                    for (int m = 0; m < 6; ++m) {
                        double jvo = 0.0, jgo = 0.0;
                        for (int c = 0; c < 6; ++c) { jvo += J2[6 * m + c] * vo[c]; jgo += J2[6 * m + c] * go[c]; }
                        pjv += 2.0 * jv[m] * jvo; pjg += 2.0 * jg[m] * jgo; pvg += jv[m] * jgo + jvo * jg[m];
                    }
                }
            }
    }
    const double a = block_sum(gsq, sm), b = block_sum(nsq, sm), c = block_sum(dot, sm);
    const double e3 = block_sum(pjv, sm), e4 = block_sum(pjg, sm), e5 = block_sum(pvg, sm);
    if (threadIdx.x == 0) {
        double *o = d.part_dl + (size_t)(d.n_lm_blocks + blockIdx.x) * NDL;
        o[0] = a; o[1] = b; o[2] = c; o[3] = e3; o[4] = e4; o[5] = e5;
    }
}

// per landmark: Gauss-Newton back-substitution delta_l = -C^-1 (g_l + sum W^T delta_p), v_l, the
// landmark parts of the norms, and |J v|^2 over the landmark's observations
template <bool DN> __global__ __launch_bounds__(256) void k_dogleg_gn(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.dl_reuse) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double gsq = 0.0, nsq = 0.0, dot = 0.0, jv2 = 0.0, jg2 = 0.0, jvg = 0.0;
    double dl[3] = {0, 0, 0}, vl[3] = {0, 0, 0};
    if (mask && !st.step_failed) {
        const LmObs<DN> ob(d, l, mask);
        const double px = d.pts[l], py = d.pts[(size_t)d.Lpad + l], pz = d.pts[2 * (size_t)d.Lpad + l];
        const double gl[3] = {d.gl[l], d.gl[(size_t)d.Lpad + l], d.gl[2 * (size_t)d.Lpad + l]};
        double tt[3] = {gl[0], gl[1], gl[2]}, tv[3] = {0.0, 0.0, 0.0};
        // ONE pass over the landmark's rows (r04: a second one formed J v and J gn row by row): with e_g = J_p gn_p and e_v = J_p v_p
        // of an observation,  |J gn|^2 = dl^T H_ll dl + 2 dl . sum J_l^T e_g + sum |e_g|^2  (H_ll = sum J_l^T J_l, the rows of constant
        // poses included: their e is 0; sum J_l^T e_g = tt - g_l), |J v|^2 and Jv . Jgn alike -- see k_ph_dogleg_gn
        double see = 0.0, svv = 0.0, sev = 0.0;
        for (int s = 0; s < ob.count(); ++s) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            const int f = d.pose_free[k];
            if (f < 0) continue;
            const double *T = d.poses + (size_t)k * 12;
            ObsLin o;
            double Sk[9];
            ob.stiffness(d, s, Sk);
            obs_linearize_S(d, Sk, T, px, py, pz, ob.u(d, s), ob.v(d, s), ob.dd(d, s), o);
            double Jp[18], Jl[9], jd[3], jv[3];
            jac_pose(o, Jp);
            jac_point(o, T, Jl);
            const double *dp = d.x0 + (size_t)f * 6, *vp = d.vp + (size_t)k * 6;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                jd[i] = 0.0; jv[i] = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) { jd[i] += Jp[6 * i + c] * dp[c]; jv[i] += Jp[6 * i + c] * vp[c]; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                tt[c] += Jl[c] * jd[0] + Jl[3 + c] * jd[1] + Jl[6 + c] * jd[2];
                tv[c] += Jl[c] * jv[0] + Jl[3 + c] * jv[1] + Jl[6 + c] * jv[2];
            }
            see += jd[0] * jd[0] + jd[1] * jd[1] + jd[2] * jd[2];
            svv += jv[0] * jv[0] + jv[1] * jv[1] + jv[2] * jv[2];
            sev += jv[0] * jd[0] + jv[1] * jd[1] + jv[2] * jd[2];
        }
        double h[6], dmp[3], Ci[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) h[c] = d.hll[(size_t)c * d.Lpad + l];
        const double H[6] = {h[0], h[1], h[2], h[3], h[4], h[5]};       // (undamped: inv3_spd may work on h in place)
        landmark_damping(d, st, l, h, dmp);
        if (inv3_spd(h, dmp, Ci)) {
            dl[0] = -(Ci[0] * tt[0] + Ci[1] * tt[1] + Ci[2] * tt[2]);
            dl[1] = -(Ci[1] * tt[0] + Ci[3] * tt[1] + Ci[4] * tt[2]);
            dl[2] = -(Ci[2] * tt[0] + Ci[4] * tt[1] + Ci[5] * tt[2]);
        } else {
            d.st->step_failed = 1;
        }
        const double hd[3] = {H[0], H[3], H[5]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double s = d.sl[(size_t)c * d.Lpad + l], s2 = s * s;
            const double D2 = fmin(fmax(hd[c] * s2, st.opt.min_lm_diag), st.opt.max_lm_diag);
            vl[c] = s2 * gl[c] / D2;
            gsq += s2 * gl[c] * gl[c] / D2;
            nsq += D2 * dl[c] * dl[c] / s2;
            dot += gl[c] * dl[c];
        }
        // upper triangle H = [h0 h1 h2; . h3 h4; . . h5]
        const double Hd[3] = {H[0] * dl[0] + H[1] * dl[1] + H[2] * dl[2], H[1] * dl[0] + H[3] * dl[1] + H[4] * dl[2], H[2] * dl[0] + H[4] * dl[1] + H[5] * dl[2]};
        const double Hv[3] = {H[0] * vl[0] + H[1] * vl[1] + H[2] * vl[2], H[1] * vl[0] + H[3] * vl[1] + H[4] * vl[2], H[2] * vl[0] + H[4] * vl[1] + H[5] * vl[2]};
        double vHv = 0.0, dHd = 0.0, vHd = 0.0, vtv = 0.0, dtg = 0.0, vtg = 0.0, dtv = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double tg = tt[c] - gl[c];
            vHv += vl[c] * Hv[c]; dHd += dl[c] * Hd[c]; vHd += vl[c] * Hd[c];
            vtv += vl[c] * tv[c]; dtg += dl[c] * tg; vtg += vl[c] * tg; dtv += dl[c] * tv[c];
        }
        jv2 = vHv + 2.0 * vtv + svv;
        jg2 = dHd + 2.0 * dtg + see;
        jvg = vHd + vtg + dtv + sev;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d.dl_gn[(size_t)c * d.Lpad + l] = dl[c];
        d.vl[(size_t)c * d.Lpad + l] = vl[c];
    }
    const double a = block_sum(gsq, sm), b = block_sum(nsq, sm), c = block_sum(dot, sm), e = block_sum(jv2, sm);
    const double e2 = block_sum(jg2, sm), e3 = block_sum(jvg, sm);
    if (threadIdx.x == 0) {
        double *o = d.part_dl + (size_t)blockIdx.x * NDL;
        o[0] = a; o[1] = b; o[2] = c; o[3] = e; o[4] = e2; o[5] = e3;
    }
}

// DoglegStrategy::ComputeTraditionalDoglegStep
__device__ void traditional_dogleg(State &st) {
    const double r = st.radius;
    if (st.gn_norm <= r) {                              // Gauss-Newton step inside the region
        st.beta = 1.0; st.gamma = 0.0; st.dl_step_norm = st.gn_norm;
    } else if (st.grad_norm * st.alpha >= r) {          // Cauchy point outside: scaled steepest descent
        st.beta = 0.0; st.gamma = -r / st.grad_norm; st.dl_step_norm = r;
    } else {                                            // on the dogleg
        const double b_dot_a = -st.alpha * st.g_dot_gn;
        const double a_sq = (st.alpha * st.grad_norm) * (st.alpha * st.grad_norm);
        const double bma = a_sq - 2.0 * b_dot_a + st.gn_norm * st.gn_norm;
        const double cc = b_dot_a - a_sq;
        const double dd = sqrt(cc * cc + bma * (r * r - a_sq));
        const double bt = (cc <= 0.0) ? (dd - cc) / bma : (r * r - a_sq) / (dd + cc);
        st.beta = bt; st.gamma = -st.alpha * (1.0 - bt);
        st.dl_step_norm = sqrt(st.gamma * st.gamma * st.grad_norm * st.grad_norm + 2.0 * st.gamma * st.beta * st.g_dot_gn +
                               st.beta * st.beta * st.gn_norm * st.gn_norm);
    }
}

// DoglegStrategy::ComputeSubspaceModel from the Gram matrices of (gradient_, gauss_newton_step_) in the
// D-scaled space (pn2, qn2, pq) and of their Jacobian images (jj = |Jv|^2, |J gn|^2, Jv.Jgn).
// ColPivHouseholderQR pivots the longer column first; rank threshold epsilon * min(rows, cols).
__device__ bool subspace_model(State &st, double pn2, double qn2, double pq, const double jj[3]) {
    const bool a_is_p = pn2 >= qn2;
    const double an2 = a_is_p ? pn2 : qn2, bn2 = a_is_p ? qn2 : pn2;
    if (!(an2 > 0.0)) return false;
    const double an = sqrt(an2), proj = pq / an;
    double wn2 = bn2 - proj * proj;
    if (wn2 < 0.0) wn2 = 0.0;
    const double wn = sqrt(wn2);
    const int ia = a_is_p ? 0 : 1, ib = 1 - ia;
    st.sub_e[0][ia] = 1.0 / an; st.sub_e[0][ib] = 0.0;
    st.sub_one_dim = wn <= 2.0 * DBL_EPSILON * an;
    if (st.sub_one_dim) return true;
    st.sub_e[1][ia] = -proj / (an * wn); st.sub_e[1][ib] = 1.0 / wn;
    for (int i = 0; i < 2; ++i) st.sub_g[i] = st.sub_e[i][0] * pn2 + st.sub_e[i][1] * pq;
    int n = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = i; j < 2; ++j)
            st.sub_B[n++] = st.sub_e[i][0] * st.sub_e[j][0] * jj[0] +
                            (st.sub_e[i][0] * st.sub_e[j][1] + st.sub_e[i][1] * st.sub_e[j][0]) * jj[2] +
                            st.sub_e[i][1] * st.sub_e[j][1] * jj[1];
    return true;
}

// DoglegStrategy::FindMinimumOnTrustRegionBoundary
__device__ bool subspace_boundary_minimum(const State &st, double mo[2]) {
    const double B00 = st.sub_B[0], B01 = st.sub_B[1], B11 = st.sub_B[2], g0 = st.sub_g[0], g1 = st.sub_g[1];
    const double detB = B00 * B11 - B01 * B01, trB = B00 + B11, r2 = st.radius * st.radius;
    const double ag0 = B11 * g0 - B01 * g1, ag1 = -B01 * g0 + B00 * g1;     // B_adj g
    double poly[5];
    poly[0] = r2;
    poly[1] = 2.0 * r2 * trB;
    poly[2] = r2 * (trB * trB + 2.0 * detB) - (g0 * g0 + g1 * g1);
    poly[3] = -2.0 * ((g0 * ag0 + g1 * ag1) - r2 * detB * trB);
    poly[4] = r2 * detB * detB - (ag0 * ag0 + ag1 * ag1);
    double roots[4];
    const int nroots = poly_roots_real(poly, 5, roots);
    mo[0] = mo[1] = 0.0;
    if (nroots < 0) return false;
    double best = DBL_MAX;
    bool found = false;
    for (int i = 0; i < nroots; ++i) {
        // x = -(B + y I)^-1 g through a partial-pivot LU of the 2x2
        double a = B00 + roots[i], b = B01, c = B01, dd = B11 + roots[i], r0 = g0, r1 = g1;
        if (fabs(c) > fabs(a)) { double t; t = a; a = c; c = t; t = b; b = dd; dd = t; t = r0; r0 = r1; r1 = t; }
        const double l = c / a, u = dd - l * b;
        const double x1 = (r1 - l * r0) / u, x0 = (r0 - b * x1) / a;
        const double x[2] = {-x0, -x1};
        const double nx = sqrt(x[0] * x[0] + x[1] * x[1]);
        if (nx > 0) {
            const double sx[2] = {st.radius / nx * x[0], st.radius / nx * x[1]};
            const double f = 0.5 * (sx[0] * (B00 * sx[0] + B01 * sx[1]) + sx[1] * (B01 * sx[0] + B11 * sx[1])) + g0 * sx[0] + g1 * sx[1];
            found = true;
            if (f < best) { best = f; mo[0] = x[0]; mo[1] = x[1]; }
        }
    }
    return found;
}

// The scalar part of DoglegStrategy::ComputeStep: norms, Cauchy point, subspace model (when the point
// is new) and beta, gamma, |step| of delta = beta * delta_gn + gamma * v for the current radius (1 block)
// Landmark sharding: the six dogleg sums (|gradient_|^2, |gn|^2, gradient_.gn, |J v|^2, |J gn|^2, Jv.Jgn) are sums over all
// residual blocks and parameters.  Every rank forms the part of ITS landmarks (own_poses: the pose and shared-block terms --
// replicated data, identical on every rank -- are counted by one rank only) into scal_dl; the ranks sum that vector (one more
// exchange point); k_dogleg_interp(from_scal = 1) goes on from the summed values.
__global__ __launch_bounds__(256) void k_dogleg_sum(Dev d, int own_poses) {
    const State &st = *d.st;
    __shared__ double sm[4];
    double acc[NDL];
#pragma unroll
    for (int q = 0; q < NDL; ++q) acc[q] = 0.0;
    if (!st.terminated && !st.dl_reuse) {
        const int n = d.n_lm_blocks + (own_poses ? d.n_pose_blocks + (d.nb ? 1 : 0) : 0);
        for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
            for (int q = 0; q < NDL; ++q) acc[q] += d.part_dl[(size_t)i * NDL + q];
    }
#pragma unroll
    for (int q = 0; q < NDL; ++q) acc[q] = block_sum(acc[q], sm);
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int q = 0; q < NSCAL; ++q) d.scal_dl[q] = q < NDL ? acc[q] : 0.0;
}

// The solver state is worked on in LDS: the scalar part below is ONE lane's chain of some sixty reads and writes of State
// fields (and the subspace model's eigen-decomposition and root finder between them), each a round trip to L2 while the struct
// sits in global memory (21 us for this launch up to r04).  The work-group copies the struct in with the loads of the partial
// sums in flight, lane 0 works on the copy, the work-group writes it back.
__global__ __launch_bounds__(256) void k_dogleg_interp(Dev d, int from_scal) {
    __shared__ State st;
    __shared__ double sm[NDL * 4];
    static_assert(sizeof(State) % sizeof(unsigned long long) == 0, "State is copied in 8-byte words");
    constexpr int NWORDS = (int)(sizeof(State) / sizeof(unsigned long long));
    unsigned long long *gw = reinterpret_cast<unsigned long long *>(d.st), *lw = reinterpret_cast<unsigned long long *>(&st);
    for (int i = threadIdx.x; i < NWORDS; i += 256) lw[i] = gw[i];
    double acc[NDL];
#pragma unroll
    for (int q = 0; q < NDL; ++q) acc[q] = 0.0;
    if (!from_scal) {       // (not used while dl_reuse holds)
        const int n = d.n_lm_blocks + d.n_pose_blocks + (d.nb ? 1 : 0);   // last entry: border of shared blocks
        for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
            for (int q = 0; q < NDL; ++q) acc[q] += d.part_dl[(size_t)i * NDL + q];
    }
    __syncthreads();
    if (st.terminated) return;
    block_sums<NDL>(acc, sm);
    if (threadIdx.x == 0) {
        if (from_scal) {
#pragma unroll
            for (int q = 0; q < NDL; ++q) acc[q] = d.scal_dl[q];
        }
        if (!st.dl_reuse) {
            st.grad_norm = sqrt(acc[0]); st.gn_norm = sqrt(acc[1]); st.g_dot_gn = acc[2];
            st.alpha = acc[0] / acc[3];   // ComputeCauchyPoint
            st.dl_jv2 = acc[3]; st.dl_jg2 = acc[4]; st.dl_jvg = acc[5];
            st.dl_reuse = 1;              // reuse_ = true until the next accepted / invalid step
            if (st.opt.dogleg_type == 1) {
                const double jj[3] = {acc[3], acc[4], acc[5]};
                if (!subspace_model(st, acc[0], acc[1], acc[2], jj)) st.step_failed = 1;   // LINEAR_SOLVER_FAILURE
            }
        }
        if (st.opt.dogleg_type != 1) {
            traditional_dogleg(st);
        } else {
            // ComputeSubspaceDoglegStep
            double m2[2];
            if (st.gn_norm <= st.radius) {
                st.beta = 1.0; st.gamma = 0.0; st.dl_step_norm = st.gn_norm;
            } else if (st.sub_one_dim) {
                st.beta = 0.0; st.gamma = -st.radius / st.grad_norm; st.dl_step_norm = st.radius;
            } else if (!subspace_boundary_minimum(st, m2)) {
                traditional_dogleg(st);           // "Taking traditional dogleg step instead."
            } else {
                st.gamma = m2[0] * st.sub_e[0][0] + m2[1] * st.sub_e[1][0];    // coefficient of gradient_ -> v
                st.beta = m2[0] * st.sub_e[0][1] + m2[1] * st.sub_e[1][1];     // coefficient of gauss_newton_step_
                st.dl_step_norm = st.radius;
            }
        }
        // Model cost change of delta = beta gn + gamma v [trust_region_minimizer.cc: -model_residuals . (residuals + model_residuals / 2),
        // model_residuals = J delta]:  -(delta . g) - |J delta|^2 / 2  with  delta . g = beta (g . gn) + gamma (g . v),  g . v = |gradient_|^2
        // and |J delta|^2 from the three row-space sums -- the evaluation kernels need no Jacobian pass for it (r04).
        const double b = st.beta, g = st.gamma;
        st.dl_mcc = -(b * st.g_dot_gn + g * st.grad_norm * st.grad_norm) - 0.5 * (b * b * st.dl_jg2 + 2.0 * b * g * st.dl_jvg + g * g * st.dl_jv2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NWORDS; i += 256) gw[i] = lw[i];
}

// per landmark: delta_l = beta * gn + gamma * v, candidate point, model cost change, candidate cost
template <bool DN> __global__ __launch_bounds__(256) void k_dogleg_eval(Dev d) {
    const State &st = *d.st;
    if (st.terminated) return;
    __shared__ double sm[4];
    const int l = blockIdx.x * 256 + threadIdx.x;
    const uint32_t mask = d.lm_mask[l];
    double ccost = 0.0, mcc = 0.0, dn = 0.0, nonfinite = 0.0;
    const double px = d.pts[l], py = d.pts[(size_t)d.Lpad + l], pz = d.pts[2 * (size_t)d.Lpad + l];
    double nx = px, ny = py, nz = pz;
    if (mask && !st.step_failed) {
        const LmObs<DN> ob(d, l, mask);
        double dl[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dl[c] = st.beta * d.dl_gn[(size_t)c * d.Lpad + l] + st.gamma * d.vl[(size_t)c * d.Lpad + l];
        if (!isfinite(dl[0]) || !isfinite(dl[1]) || !isfinite(dl[2])) nonfinite = 1.0;
        nx = px + dl[0]; ny = py + dl[1]; nz = pz + dl[2];
        dn = dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2];
        for (int s = 0; s < ob.count(); ++s) {
            if (!ob.has(s)) continue;
            const uint32_t k = ob.pose(d, s);
            const double u = ob.u(d, s), v = ob.v(d, s), dd = ob.dd(d, s);
            double Sk[9];
            ob.stiffness(d, s, Sk);
            // (the model cost change comes from the six sums of the dogleg model: k_dogleg_interp, State::dl_mcc)
            ccost += obs_cost_S(d, Sk, d.cand_poses + (size_t)k * 12, nx, ny, nz, u, v, dd);
        }
    }
    d.cand_pts[l] = nx;
    d.cand_pts[(size_t)d.Lpad + l] = ny;
    d.cand_pts[2 * (size_t)d.Lpad + l] = nz;
    const double a = block_sum(ccost, sm), b = block_sum(mcc, sm), c = block_sum(dn, sm), e = block_sum(nonfinite, sm);
    if (threadIdx.x == 0) {
        d.part_eval[blockIdx.x * 4 + 0] = a;
        d.part_eval[blockIdx.x * 4 + 1] = b;
        d.part_eval[blockIdx.x * 4 + 2] = c;
        d.part_eval[blockIdx.x * 4 + 3] = e;
    }
}

// Sums the per-block partials of the linearisation into the exchange scalars
// scal[0] = cost, scal[1] = |x_points|^2 and gmax_l (fixed order, one block).
__global__ __launch_bounds__(256) void k_reduce_lin(Dev d, int n_parts) {
    State &st = *d.st;
    if (st.terminated || !st.need_linearize) return;
    __shared__ double sm[4];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n_parts; i += 256) {
        a += d.part_lin[i * 4];
        b += d.part_lin[i * 4 + 1];
        c = fmax(c, d.part_lin[i * 4 + 2]);
    }
    if (d.n_pf)
        for (int k = threadIdx.x; k < d.P; k += 256) a += d.pf_cost[k];     // unary pose residual blocks
    a = block_sum(a, sm);
    b = block_sum(b, sm);
    c = block_max(c, sm);
    if (threadIdx.x == 0) {
        d.xv[d.off_scal + 0] = a;
        d.xv[d.off_scal + 1] = b;
        *d.gmax_l = c;
        st.need_linearize = 0;
        st.just_linearized = 1;
    }
}

// add_pose (partitioned solve): the pose part of |dx|^2 and of the non-finite flag (owned poses only: k_pose_update's
// partials) joins the landmark sums before the exchange -- k_eval_add_pose's work, one launch less
__global__ __launch_bounds__(256) void k_reduce_eval(Dev d, int n_parts, int add_pose) {
    const State &st = *d.st;
    if (st.terminated) return;
    __shared__ double sm[4];
    double a = 0.0, b = 0.0, c = 0.0, e = 0.0, pa = 0.0, pb = 0.0;
    for (int i = threadIdx.x; i < n_parts; i += 256) {
        a += d.part_eval[i * 4];
        b += d.part_eval[i * 4 + 1];
        c += d.part_eval[i * 4 + 2];
        e += d.part_eval[i * 4 + 3];
    }
    if (add_pose)
        for (int i = threadIdx.x; i < d.n_pose_blocks; i += 256) { pa += d.part_pose[i * NPP]; pb += d.part_pose[i * NPP + 1]; }
    a = block_sum(a, sm);
    b = block_sum(b, sm);
    c = block_sum(c, sm);
    e = block_sum(e, sm);
    if (add_pose) { pa = block_sum(pa, sm); pb = block_sum(pb, sm); }
    if (threadIdx.x == 0) {
        d.scal2[0] = a; d.scal2[1] = b; d.scal2[2] = c + pa; d.scal2[3] = e + pb;
    }
}

// with_best: the launch also does k_best's copy (x -> best when the k_check of THIS iteration saw the cost improve) in front of
// its own -- nothing reads the best iterate or moves x between the two places, and like k_best that part does not test
// `terminated` (the improving iterate may be the converged one).  Where neither the update / evaluation kernels (fuse_best) nor
// the linearisation (fuse_all) carry the copies -- bounds, free shared blocks, dogleg -- this saves the k_best launch.
__global__ __launch_bounds__(256) void k_commit(Dev d, int with_best) {
    const State &st = *d.st;
    const bool best = with_best && st.copy_best == st.check_count;
    const bool commit = !st.terminated && st.accepted;
    if (!best && !commit) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (best) {
        if (i < (size_t)d.P * 12) d.best_poses[i] = d.poses[i];
        if (i < (size_t)d.Lpad * 3) {
            d.best_pts[i] = d.pts[i];
            if (d.phong) d.best_nrm[i] = d.nrm[i];
        }
        if (d.phong && i < (size_t)d.nsh) d.best_sh[i] = d.sh[i];
    }
    if (!commit) return;
    if (i < (size_t)d.P * 12) d.poses[i] = d.cand_poses[i];
    if (i < (size_t)d.Lpad * 3) {
        d.pts[i] = d.cand_pts[i];
        if (d.phong) d.nrm[i] = d.cand_nrm[i];
    }
    if (d.phong && i < (size_t)d.nsh) d.sh[i] = d.cand_sh[i];
}

// (re)start of a solve: reset the trust-region state on the device
__global__ void k_reset_state(Dev d, Options opt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    State &st = *d.st;
    st.opt = opt;
    st.iteration = 0; st.terminated = 0; st.termination_type = 1;
    st.need_linearize = 1; st.just_linearized = 0; st.last_successful = 0; st.accepted = 0;
    st.copy_best = 0; st.check_count = 0; st.step_failed = 0;
    st.num_successful = 0; st.num_unsuccessful = 0; st.num_invalid = 0; st.log_count = 0;
    st.radius = opt.initial_radius; st.decrease_factor = 2.0;
    st.x_cost = 0.0; st.x_norm = 0.0; st.gmax = 0.0; st.minimum_cost = 0.0;
    st.candidate_cost = 0.0; st.model_cost_change = 0.0; st.step_norm = 0.0;
    st.relative_decrease = 0.0; st.cost_change = 0.0; st.initial_cost = 0.0;
    st.dl_reuse = 0; st.mu = 1e-8; st.alpha = 0.0; st.dl_step_norm = 0.0; st.grad_norm = 0.0; st.gn_norm = 0.0;
    st.g_dot_gn = 0.0; st.beta = 1.0; st.gamma = 0.0;
    st.ls_alpha = 1.0;
    st.ls_pending = st.ls_active = st.ls_steps = st.ls_searches = 0;
    st.sub_one_dim = 0; st.sub_g[0] = st.sub_g[1] = 0.0; st.sub_B[0] = st.sub_B[1] = st.sub_B[2] = 0.0;
    st.sub_e[0][0] = st.sub_e[0][1] = st.sub_e[1][0] = st.sub_e[1][1] = 0.0;
}

// ------------------------------------------------------------------ partitioned solve ---
// One rank per contiguous chain of super-blocks (landmark sharding).  After the local
// elimination of the chain interior (ssba_bcr.hip, pinned ends) the two end blocks hold this rank's share of
// the separator system; k_sep_pack writes them, the ends' gradient / diag(H_pp) and the scalars of the
// linearisation into the (zeroed) separator vector, which the ranks then sum.
// n_lin_parts > 0: the last workgroup also forms the sums of the linearisation partials (k_reduce_lin's work: that launch
// is skipped; nothing between the linearisation and this kernel reads them).
__global__ __launch_bounds__(256) void k_sep_pack(Dev d, int n_lin_parts) {
    State &st = *d.st;
    if (st.terminated) return;
    const BcrLevel &E = d.lev[d.pcr.level];          // the pinned first / last block of this level: the shared chain ends
    const int t = threadIdx.x, r = d.rank, last = E.n - 1;
    const size_t blk = (size_t)BD * BD;
    // separator r - 1 is this chain's first block, separator r its last one
    if (blockIdx.x < 2) {
        if (!(blockIdx.x ? d.pin1 : d.pin0)) return;
        double *dst = d.sepv + d.soff_D + (size_t)(r - 1 + blockIdx.x) * blk;
        const double *src = E.D + (size_t)(blockIdx.x ? last : 0) * blk;
        for (int i = t; i < BD * BD; i += 256) dst[i] = src[i];
    } else if (blockIdx.x == 2) {
        // coupling of the two ends, S[last, first], left in the plan's buffer by the last fold of the pinned last block
        // -> coupling block of separator r (an even index is stored transposed)
        if (!(d.pin0 && d.pin1)) return;
        double *dst = d.sepv + d.soff_L + (size_t)r * blk;
        const double *src = d.pcr.Lbuf + (size_t)last * blk;
        const bool tr = (r & 1) == 0;
        for (int i = t; i < BD * BD; i += 256) {
            const int row = i / BD, col = i - row * BD;
            dst[tr ? col * BD + row : i] = src[i];
        }
    } else {
        __shared__ double sm[4];
        for (int e = 0; e < 2; ++e) {
            if (!(e ? d.pin1 : d.pin0)) continue;
            const int sb = e ? d.chain1 : d.chain0;
            if (t < BD) {
                d.sepv[d.soff_rhs + (size_t)(r - 1 + e) * BD + t] = E.r[(size_t)(e ? last : 0) * BD + t];
                d.sepv[d.soff_gp + (size_t)(r - 1 + e) * BD + t] = d.xv[d.off_gp + (size_t)sb * BD + t];
                d.sepv[d.soff_hdiag + (size_t)(r - 1 + e) * BD + t] = d.xv[d.off_hdiag + (size_t)sb * BD + t];
            }
        }
        const bool fused = n_lin_parts > 0, lin = fused ? st.need_linearize != 0 : st.just_linearized != 0;
        double la = 0.0, lb = 0.0, lc = 0.0;
        if (fused && lin) {
            for (int i = t; i < n_lin_parts; i += 256) {
                la += d.part_lin[i * 4];
                lb += d.part_lin[i * 4 + 1];
                lc = fmax(lc, d.part_lin[i * 4 + 2]);
            }
            if (d.n_pf)
                for (int k = t; k < d.P; k += 256) la += d.pf_cost[k];     // unary pose residual blocks
            la = block_sum(la, sm);
            lb = block_sum(lb, sm);
            lc = block_max(lc, sm);
        }
        // interior poses of this rank: projected gradient and |x|^2 join the landmark sums of the linearisation
        double gm = 0.0, xn = 0.0;
        if (lin) {
            for (int i = (d.chain0 + d.pin0) * SBP + t; i < (d.chain1 + 1 - d.pin1) * SBP && i < d.nfree; i += 256) {
                const int k = d.free_pose[i];
                const double *T = d.poses + (size_t)k * 12;
                double ng[6], Tn[12];
#pragma unroll
                for (int c = 0; c < 6; ++c) ng[c] = -d.xv[d.off_gp + (size_t)i * 6 + c];
                se3_plus(T, ng, Tn);
#pragma unroll
                for (int c = 0; c < 12; ++c) { gm = fmax(gm, fabs(T[c] - Tn[c])); xn += T[c] * T[c]; }
            }
        }
        const double gmp = block_max(gm, sm), xnp = block_sum(xn, sm);
        if (t == 0) {
            double *sc = d.xv + d.off_scal, *ss = d.sepv + d.soff_scal;
            if (lin) {
                if (fused) { sc[0] = la; sc[1] = lb; *d.gmax_l = lc; st.need_linearize = 0; st.just_linearized = 1; }
                ss[0] = sc[0]; ss[1] = sc[1] + xnp;
                // the maximum over ranks rides in the SUM exchange: every rank fills its own slot, the others add zeros
                ss[NSCAL + d.rank] = fmax(*d.gmax_l, gmp);
                sc[0] = 0.0; sc[1] = 0.0;
            }
        }
    }
}

// after the exchange: Jacobi scale (iteration 0) and LM damping of the separator diagonals, identity rows for
// the padding of the last super-block
__global__ __launch_bounds__(256) void k_sep_finish(Dev d) {
    const State &st = *d.st;
    if (st.terminated) return;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.n_sep * BD) return;
    const int s = q / BD, row = q - s * BD;
    const int i = d.sep_sb[s] * BD + row, f = i / 6;
    double *Dd = d.sepv + d.soff_D + (size_t)s * BD * BD + (size_t)row * BD + row;
    if (f < d.nfree) {
        const double h = d.sepv[d.soff_hdiag + q];
        if (st.iteration == 0) d.sp[i] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
        const double sc = d.sp[i], s2 = sc * sc;
        *Dd += fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
    } else {
        *Dd = 1.0;
        d.sepv[d.soff_rhs + q] = 0.0;
    }
}

// The separator solution is known on every rank: all separator poses are updated everywhere (they enter the
// convergence checks of every rank), the chain ends among them seed the back-substitution of the interior.
__global__ __launch_bounds__(256) void k_sep_scatter(Dev d) {
    const State &st = *d.st;
    if (st.terminated || st.step_failed) return;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.n_sep * BD) return;
    const int s = q / BD, row = q - s * BD;
    d.x0[(size_t)d.sep_sb[s] * BD + row] = d.xsep[q];
}

// zero the poses another rank owns, so that a sum over ranks gathers the solution
__global__ __launch_bounds__(256) void k_mask_unowned_poses(Dev d, double *poses) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.P) return;
    const int f = d.pose_free[k];
    const bool owned = f < 0 ? d.rank == 0 : (f >= (d.rank == 0 ? d.chain0 : d.chain0 + 1) * SBP && f < (d.chain1 + 1) * SBP);
    if (!owned)
        for (int c = 0; c < 12; ++c) poses[(size_t)k * 12 + c] = 0.0;
}

// zero-fill of a list of buffers in one launch (ssba_finalize): work-group r clears range r
__global__ __launch_bounds__(256) void k_zero_ranges(const ZeroRange *ranges) {
    const ZeroRange r = ranges[blockIdx.x];
    double2 *q = reinterpret_cast<double2 *>(r.ptr);
    const uint64_t n16 = r.bytes / 16;
    for (uint64_t i = threadIdx.x; i < n16; i += 256) q[i] = make_double2(0.0, 0.0);
    unsigned char *tail = reinterpret_cast<unsigned char *>(r.ptr) + n16 * 16;
    if (threadIdx.x < (r.bytes & 15)) tail[threadIdx.x] = 0;
}
void launch_zero_ranges(hipStream_t stream, const ZeroRange *ranges, int n) {
    if (n > 0) hipLaunchKernelGGL(k_zero_ranges, dim3(n), dim3(256), 0, stream, ranges);
}

// ----------------------------------------------------------------- launchers ---
int configure_schur() {
    return hipFuncSetAttribute((const void *)k_schur_windows, hipFuncAttributeMaxDynamicSharedMemorySize, SCHUR_LDS_BYTES) == hipSuccess ? 0 : -1;
}

void launch_reset(Launcher &L, const Dev &d, const Options &o) {
    hipLaunchKernelGGL(k_reset_state, dim3(1), dim3(64), 0, L.stream, d, o);
}

// Window layout: several lanes per landmark pay off while one lane per landmark leaves the SIMDs short of waves (C2: 1.5
// waves per SIMD); from ~4 waves per SIMD on the plain mapping wins (C4: 15 600 waves; measured 0.106 / 0.224 ms vs
// 0.133 / 0.272 ms for the split kernels).
// r04: above that size the window kernels run with ONE wave per 64 landmarks (SP = 1: the plain mapping's lane count) instead
// of handing the problem to the generic kernels -- those lack the folded control work (commit inside the linearisation, best
// copy inside the evaluation, k_check's sums per group), so C4 on one GPU paid five dependent single-block launches (60 us) per
// iteration that C2 does not have.
static bool lm_split(const Dev &d) {
    static const bool cliff = [] { const char *e = getenv("SSBA_LM_CLIFF"); return e && e[0] == '1'; }();      // r03's bound (A/B)
    return !d.dense && !d.phong && (!cliff || d.Lpad <= 262144);
}
static int lm_sp(const Dev &d) { return d.Lpad <= 262144 ? LMW_SPLIT : 1; }
// General layout (landmark-major lists): waves per 64 landmarks of the two landmark passes; 0 = one lane per landmark in
// work-groups of 256 (the r03 kernels; SSBA_DN_SPLIT=0/1/2/4 for A/B).
static int dn_sp(const Dev &d) {
    static const int forced = [] { const char *e = getenv("SSBA_DN_SPLIT"); return e ? atoi(e) : -1; }();
    if (!d.dense || d.phong) return 0;
    if (forced >= 0) return forced == 1 || forced == 2 || forced == 4 ? forced : 0;
    return d.Lpad <= 131072 ? 4 : (d.Lpad <= 262144 ? 2 : 1);
}
// entries of part_lin / part_eval: one per group of 64 landmarks, or per block of 256
static int lm_parts(const Dev &d) { return lm_split(d) || dn_sp(d) ? d.n_groups : d.n_lm_blocks; }

// fuse_ctrl (single GPU, windowed stereo layout; see k_check): k_reduce_lin's sums are formed by k_check
// r04: the general layout too (stereo blocks: wide super-blocks and blocked Cholesky) -- k_check forms the sums of k_reduce_lin and
// walks the poses itself (check_body)
static bool ctrl_fusable(const Dev &d) { return !d.part && (!d.dense || !d.phong); }       // (lighting terms included on the windowed layout: same partial sums, same reduced system)
// fuse_best (the copy of x to the best iterate rides in the update / evaluation kernels): no exchange sits between k_check's
// decision and those kernels in any mode, so the partitioned multi-GPU solve takes it too
// lighting terms with free shared blocks: k_pose_update / k_dogleg_vec get one more work-group for the border entries
static int border_block(const Dev &d) { return d.phong && d.nb ? 1 : 0; }
static bool best_fusable(const Dev &d) { return !d.nb && (d.phong ? !d.dense : (lm_split(d) || dn_sp(d) > 0)); }       // (free shared blocks have a best copy of their own: k_best)
bool launch_ctrl_fusable(const Dev &d) { return ctrl_fusable(d); }
bool launch_can_fuse_all(const Dev &d) { return ctrl_fusable(d) && !d.phong && (lm_split(d) || dn_sp(d) > 0); }
// fuse_all (single GPU, LM, windowed stereo layout, launch_can_fuse_all): the linearisation kernels commit the accepted
// step on the way (no k_commit launch)
// skip_reduce (partitioned solve): k_sep_pack(.., n_lin_parts) forms the sums of the partials
void launch_linearize(Launcher &L, const Dev &d, bool fuse_ctrl, bool fuse_all, bool skip_reduce) {
    fuse_ctrl = fuse_ctrl && ctrl_fusable(d);
    if (d.phong) {
        launch_ph_linearize(L, d);
    } else {
        static const bool one_launch = [] { const char *e = getenv("SSBA_LIN_TWO_LAUNCHES"); return !(e && e[0] == '1'); }();
        if (one_launch && lm_split(d) && lm_sp(d) == LMW_SPLIT && !d.dense) {
            LAUNCH(KC_LIN_LM, k_linearize_both, dim3(d.n_groups + xcd_contiguous_grid(d.P)), dim3(LP_THREADS), 0, d, fuse_all ? 1 : 0, d.n_groups);
            if (!fuse_ctrl && !skip_reduce) LAUNCH(KC_SMALL, k_reduce_lin, dim3(1), dim3(256), 0, d, lm_parts(d));
            return;
        }
        if (lm_split(d) && lm_sp(d) == LMW_SPLIT) LAUNCH(KC_LIN_LM, (k_linearize_landmarks_w<false, LMW_SPLIT>), dim3(d.n_groups), dim3(64 * LMW_SPLIT), 0, d, fuse_all ? 1 : 0);
        else if (lm_split(d)) LAUNCH(KC_LIN_LM, (k_linearize_landmarks_w<false, 1>), dim3(d.n_groups), dim3(64), 0, d, fuse_all ? 1 : 0);
        else if (dn_sp(d) == 4) LAUNCH(KC_LIN_LM, (k_linearize_landmarks_w<true, 4>), dim3(d.n_groups), dim3(256), 0, d, fuse_all ? 1 : 0);
        else if (dn_sp(d) == 2) LAUNCH(KC_LIN_LM, (k_linearize_landmarks_w<true, 2>), dim3(d.n_groups), dim3(128), 0, d, fuse_all ? 1 : 0);
        else if (dn_sp(d) == 1) LAUNCH(KC_LIN_LM, (k_linearize_landmarks_w<true, 1>), dim3(d.n_groups), dim3(64), 0, d, fuse_all ? 1 : 0);
        else LAUNCH(KC_LIN_LM, (d.dense ? k_linearize_landmarks<true> : k_linearize_landmarks<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
        // 128 lanes per pose, five observations in flight per lane (sweep on C2, profiles/r02_pose_kernel_shape.txt: 64 / 128 / 192 /
        // 256 / 512 lanes x 3-10 observations: 24.5 us here, 31 us for 256 x 3, 51 us for 512 x 3)
        LAUNCH(KC_LIN_POSE, (d.dense ? k_linearize_poses<true, LP_THREADS, LP_CHUNK> : k_linearize_poses<false, LP_THREADS, LP_CHUNK>), dim3(xcd_contiguous_grid(d.P)), dim3(LP_THREADS), 0, d, fuse_all ? 1 : 0);
    }
    if (!fuse_ctrl && !skip_reduce) LAUNCH(KC_SMALL, k_reduce_lin, dim3(1), dim3(256), 0, d, lm_parts(d));
}

void launch_schur(Launcher &L, const Dev &d, bool fuse_ctrl, bool check_in_schur) {
    fuse_ctrl = fuse_ctrl && ctrl_fusable(d);
    check_in_schur = check_in_schur && fuse_ctrl && (d.phong || lm_split(d));
    // the reduced system is cleared on the way: by 128 extra workgroups of the stereo Schur launch, by k_ph_invert with lighting terms
    const int n_zero = 128;
    if (d.phong) launch_ph_schur(L, d, check_in_schur);
    else LAUNCH(KC_SCHUR, k_schur_windows, dim3(d.n_slabs + n_zero + (check_in_schur ? 1 : 0)), dim3(SCHUR_THREADS), SCHUR_LDS_BYTES, d, n_zero, check_in_schur ? d.n_groups : 0);
    const size_t n = (size_t)d.n_sblk * 36 + (size_t)(fuse_ctrl ? d.nf_pad : d.nfree) * 6 + (fuse_ctrl ? (size_t)d.nfree : 0);
    LAUNCH(KC_ASSEMBLE, k_assemble_reduced, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d, fuse_ctrl ? 1 : 0, check_in_schur ? 1 : 0);
    if (d.cb) LAUNCH(KC_BORDER, k_cb_assemble, dim3((unsigned)(((size_t)d.n_cb * 36 + 255) / 256)), dim3(256), 0, d);
}

void launch_finish_local(Launcher &L, const Dev &d) {
    LAUNCH(KC_SMALL, k_finish_reduced, dim3((d.nf_pad * 6 + 255) / 256), dim3(256), 0, d);
}

// fuse_ctrl: one launch instead of k_finish_reduced + k_check (+ the k_reduce_lin skipped by launch_linearize);
// fuse_best: k_best's copy is done by the update / evaluation kernels of launch_update_eval(.., fuse_best)
bool launch_best_fusable(const Dev &d) { return best_fusable(d); }
void launch_finish_check(Launcher &L, const Dev &d, bool fuse_ctrl, bool fuse_best, bool check_in_schur, bool best_in_commit) {
    fuse_ctrl = fuse_ctrl && ctrl_fusable(d);
    check_in_schur = check_in_schur && fuse_ctrl && (d.phong || lm_split(d));
    if (d.dense) launch_dense_finish(L, d, fuse_ctrl);
    else if (!fuse_ctrl) LAUNCH(KC_SMALL, k_finish_reduced, dim3((d.nf_pad * 6 + 255) / 256), dim3(256), 0, d);
    if (!check_in_schur) LAUNCH(KC_SMALL, k_check, dim3(1), dim3(1024), 0, d, fuse_ctrl ? lm_parts(d) : 0);
    if ((fuse_best && best_fusable(d)) || best_in_commit) return;
    const size_t n = (size_t)d.P * 12 > (size_t)d.Lpad * 3 ? (size_t)d.P * 12 : (size_t)d.Lpad * 3;
    LAUNCH(KC_COPY, k_best, dim3((unsigned)std::min<size_t>((n + 255) / 256, 512)), dim3(256), 0, d);
}

// fuse_reduce: the caller's launch_decide_commit(.., true) forms the evaluation sums (no exchange in between)
void launch_update_eval(Launcher &L, const Dev &d, bool fuse_reduce, bool fuse_best, bool pose_update_done) {
    const int fb = fuse_best && best_fusable(d) ? 1 : 0;
    if (!pose_update_done) LAUNCH(KC_SMALL, k_pose_update, dim3(d.n_pose_blocks + border_block(d)), dim3(256), 0, d, fb);
    if (d.phong) launch_ph_backsub_eval(L, d, fb, !pose_update_done);
    else if (lm_split(d) && lm_sp(d) == LMW_SPLIT) LAUNCH(KC_BACKSUB_EVAL, (k_backsub_eval_w<false, LMW_SPLIT>), dim3(d.n_groups), dim3(64 * LMW_SPLIT), 0, d, pose_update_done ? 2 : fb);
    else if (lm_split(d)) LAUNCH(KC_BACKSUB_EVAL, (k_backsub_eval_w<false, 1>), dim3(d.n_groups), dim3(64), 0, d, pose_update_done ? 2 : fb);
    else if (dn_sp(d) == 4) LAUNCH(KC_BACKSUB_EVAL, (k_backsub_eval_w<true, 4>), dim3(d.n_groups), dim3(256), 0, d, pose_update_done ? 2 : fb);
    else if (dn_sp(d) == 2) LAUNCH(KC_BACKSUB_EVAL, (k_backsub_eval_w<true, 2>), dim3(d.n_groups), dim3(128), 0, d, pose_update_done ? 2 : fb);
    else if (dn_sp(d) == 1) LAUNCH(KC_BACKSUB_EVAL, (k_backsub_eval_w<true, 1>), dim3(d.n_groups), dim3(64), 0, d, pose_update_done ? 2 : fb);
    else LAUNCH(KC_BACKSUB_EVAL, (d.dense ? k_backsub_eval<true> : k_backsub_eval<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
    if (!fuse_reduce) LAUNCH(KC_SMALL, k_reduce_eval, dim3(1), dim3(256), 0, d, lm_parts(d), d.part ? 1 : 0);
}

void launch_sep_pack(Launcher &L, const Dev &d) {       // (the separator vector was cleared by k_finish_reduced)
    LAUNCH(KC_SMALL, k_sep_pack, dim3(4), dim3(256), 0, d, d.phong ? 0 : lm_parts(d));
}
void launch_sep_finish_check(Launcher &L, const Dev &d, bool fuse_best) {
    LAUNCH(KC_SMALL, k_sep_finish, dim3((d.n_sep * BD + 255) / 256), dim3(256), 0, d);
    LAUNCH(KC_SMALL, k_check, dim3(1), dim3(1024), 0, d, 0);
    if (fuse_best && best_fusable(d)) return;
    const size_t n = (size_t)d.P * 12 > (size_t)d.Lpad * 3 ? (size_t)d.P * 12 : (size_t)d.Lpad * 3;
    LAUNCH(KC_COPY, k_best, dim3((unsigned)std::min<size_t>((n + 255) / 256, 512)), dim3(256), 0, d);
}
void launch_sep_scatter(Launcher &L, const Dev &d) {
    LAUNCH(KC_SMALL, k_sep_scatter, dim3((d.n_sep * BD + 255) / 256), dim3(256), 0, d);
}
void launch_mask_unowned_poses(Launcher &L, const Dev &d, double *poses) {
    LAUNCH(KC_SMALL, k_mask_unowned_poses, dim3((d.P + 255) / 256), dim3(256), 0, d, poses);
}

void launch_pose_update(Launcher &L, const Dev &d, int ls_round) {
    LAUNCH(KC_SMALL, k_pose_update, dim3(d.n_pose_blocks + border_block(d)), dim3(256), 0, d, ls_round ? -1 : 0);
}

// stage 0: everything (single GPU).  Landmark sharding: stage 1 = up to this rank's six sums (scal_dl; own_poses on one rank),
// stage 2 = from the summed vector on
int eval_parts(const Dev &d) { return d.phong ? d.n_lm_blocks : lm_parts(d); }
void launch_dogleg_eval(Launcher &L, const Dev &d, int stage, int own_poses, bool reduce_later) {
    if (stage != 2) {
        LAUNCH(KC_SMALL, k_dogleg_vec, dim3(d.n_pose_blocks + border_block(d)), dim3(256), 0, d);
        if (d.phong) launch_ph_dogleg_gn(L, d);
        else LAUNCH(KC_DOGLEG, (d.dense ? k_dogleg_gn<true> : k_dogleg_gn<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
    }
    if (stage == 1) {
        LAUNCH(KC_SMALL, k_dogleg_sum, dim3(1), dim3(256), 0, d, own_poses);
        return;
    }
    LAUNCH(KC_SMALL, k_dogleg_interp, dim3(1), dim3(256), 0, d, stage == 2 ? 1 : 0);
    LAUNCH(KC_SMALL, k_pose_update, dim3(d.n_pose_blocks + border_block(d)), dim3(256), 0, d, 0);
    if (d.phong) launch_ph_dogleg_eval(L, d);
    else LAUNCH(KC_DOGLEG, (d.dense ? k_dogleg_eval<true> : k_dogleg_eval<false>), dim3(d.n_lm_blocks), dim3(256), 0, d);
    if (!reduce_later) LAUNCH(KC_SMALL, k_reduce_eval, dim3(1), dim3(256), 0, d, d.n_lm_blocks, 0);
}

void launch_decide_commit(Launcher &L, const Dev &d, bool fuse_reduce, bool fuse_all, int n_pose_parts, bool with_best) {
    LAUNCH(KC_SMALL, k_decide, dim3(1), dim3(256), 0, d, fuse_reduce ? lm_parts(d) : 0,
           n_pose_parts >= 0 ? n_pose_parts : d.n_pose_blocks);
    if (fuse_all) return;       // the next linearisation commits (launch_linearize)
    const size_t n = (size_t)d.P * 12 > (size_t)d.Lpad * 3 ? (size_t)d.P * 12 : (size_t)d.Lpad * 3;
    LAUNCH(KC_COPY, k_commit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d, with_best ? 1 : 0);
}

}  // namespace ssba
