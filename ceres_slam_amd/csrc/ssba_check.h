// Iteration bookkeeping shared by the control kernel (k_check, ssba_kernels.hip) and the Schur launches that carry its work
// as one extra work-group (k_schur_windows, k_ph_schur_windows).
#pragma once
#include "ssba_device.h"
#include "ssba_types.h"

namespace ssba {

static __device__ __forceinline__ void log_push(Dev &d, State &st, double cost, double cost_change, double step_norm, double rd,
                         int ok) {
    const int i = st.log_count++;
    if (i < d.log.capacity) {
        d.log.cost[i] = cost; d.log.cost_change[i] = cost_change; d.log.gmax[i] = st.gmax;
        d.log.step_norm[i] = step_norm; d.log.relative_decrease[i] = rd; d.log.radius[i] = st.radius;
        d.log.successful[i] = ok;
    }
}

// Ceres TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue (plus the
// reductions of EvaluateGradientAndJacobian): one block.
// fused_parts > 0 (single GPU, windowed stereo layout: nothing is exchanged between the linearisation and this kernel):
// the block also does the work of k_reduce_lin (sums of the linearisation partials); k_finish_reduced's work is done by
// k_assemble_reduced(.., fuse_finish) -- two launches less per iteration.  (A first version did k_finish_reduced's
// work here too: 6 000 scattered diagonal entries from ONE block cost 16 us against 5 us for the 24-block launch.)
// in_schur (check_body called from the extra work-group of k_schur_windows, i.e. BEFORE k_assemble_reduced): the per-pose
// partials of k_assemble_reduced do not exist yet -- the poses are walked here, from g_p, hidden inside the 75 us launch
static __device__ __forceinline__ void check_body(Dev &d, int fused_parts, bool in_schur) {
    State &st = *d.st;
    if (st.terminated) return;
    __shared__ double sm[16];
    double gm = 0.0, xn = 0.0, cost = 0.0;
    const bool lin = fused_parts > 0 ? st.need_linearize != 0 : st.just_linearized != 0;
    double f_cost = 0.0, f_xn = 0.0, f_gml = 0.0;
    if (fused_parts > 0) {
        if (lin) {          // k_reduce_lin
            double a = 0.0, b = 0.0, c = 0.0;
            for (int i = threadIdx.x; i < fused_parts; i += (int)blockDim.x) {
                a += d.part_lin[i * 4];
                b += d.part_lin[i * 4 + 1];
                c = fmax(c, d.part_lin[i * 4 + 2]);
            }
            if (d.n_pf)
                for (int k = threadIdx.x; k < d.P; k += (int)blockDim.x)
                    a += d.pf_cost[k];
            f_cost = block_sum(a, sm);
            f_xn = block_sum(b, sm);
            f_gml = block_max(c, sm);
        }
    }
    if (lin) {
        // landmark partials (already all-reduced in scal[] when sharded: see host).  Partitioned solve: the
        // interior poses of every rank went into those sums before the exchange (k_sep_pack); what is left
        // are the separator poses, whose gradient is the sum over ranks held in the separator vector
        const int npose = d.part ? d.n_sep * SBP : d.nfree;
        if (fused_parts > 0 && !in_schur && !d.dense) {       // per pose by k_assemble_reduced(.., fuse_finish); the general layout has no such pass: the poses are walked below
            for (int q = threadIdx.x; q < d.nfree; q += (int)blockDim.x) { gm = fmax(gm, d.part_chk[2 * q]); xn += d.part_chk[2 * q + 1]; }
        } else
        for (int q = threadIdx.x; q < npose; q += (int)blockDim.x) {
            int i = q;
            const double *gsrc = in_schur ? d.gp + (size_t)d.free_pose[q] * 6 : d.xv + d.off_gp + (size_t)q * 6;
            if (d.part) {
                const int s = q / SBP;
                i = d.sep_sb[s] * SBP + (q - s * SBP);
                if (i >= d.nfree) continue;
                gsrc = d.sepv + d.soff_gp + (size_t)q * 6;
            }
            const int k = d.free_pose[i];
            const double *T = d.poses + (size_t)k * 12;
            double ng[6], Tn[12];
#pragma unroll
            for (int c = 0; c < 6; ++c) ng[c] = -gsrc[c];
            se3_plus(T, ng, Tn);    // projected gradient: |x - Plus(x, -g)|_inf
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                gm = fmax(gm, fabs(T[c] - Tn[c]));
                xn += T[c] * T[c];
            }
        }
    }
    // free shared blocks: lane 0's walk below reads the blocks and their gradient from LDS (up to r04 from global memory, entry
    // by entry -- two dozen dependent round trips in front of the termination tests); the barriers of the sums publish the copy
    __shared__ double s_sh[64], s_gb[NBP];
    if (d.nb && lin) {
        if ((int)threadIdx.x < d.nsh && threadIdx.x < 64) s_sh[threadIdx.x] = d.sh[threadIdx.x];
        if ((int)threadIdx.x < d.nb) s_gb[threadIdx.x] = d.bsys[BS_G + threadIdx.x];
    }
    const double gmp = block_max(gm, sm);
    const double xnp = block_sum(xn, sm);
    (void)cost;
    if (threadIdx.x != 0) return;
    ++st.check_count;
    if (lin) {
        double *sc = d.part ? d.sepv + d.soff_scal : d.xv + d.off_scal;
        if (fused_parts > 0) {
            sc[0] = f_cost; sc[1] = f_xn; *d.gmax_l = f_gml;
            st.need_linearize = 0;
        }
        st.x_cost = sc[0];
        double xnb = 0.0, gmb = 0.0;
        if (d.nb) {   // free shared blocks: |x_b|^2 and |x_b - Plus(x_b, -g_b)|_inf
            const double *gb = s_gb, *sh = s_sh;
            if (d.b_light >= 0) {
                double ng[3] = {-gb[d.b_light], -gb[d.b_light + 1], -gb[d.b_light + 2]}, nl[3];
                if (d.light_type == 1) unit_plus(sh, ng, nl);
                else for (int c = 0; c < 3; ++c) nl[c] = sh[c] + ng[c];
                for (int c = 0; c < 3; ++c) { xnb += sh[c] * sh[c]; gmb = fmax(gmb, fabs(nl[c] - sh[c])); }
            }
            // Plus projects onto the bounds [Ceres ParameterBlock::Plus], so the projected gradient does too
            if (d.b_phong >= 0)
                for (int c = 0; c < 3 * d.M; ++c) {
                    const double v = sh[3 + c];
                    double nv = v - gb[d.b_phong + c];
                    if (d.constrained) nv = fmin(fmax(nv, d.blo[c % 3]), d.bhi[c % 3]);
                    xnb += v * v; gmb = fmax(gmb, fabs(nv - v));
                }
            if (d.b_tex >= 0)
                for (int c = 0; c < d.M; ++c) {
                    const double v = sh[3 + 3 * d.M + c];
                    double nv = v - gb[d.b_tex + c];
                    if (d.constrained) nv = fmin(fmax(nv, d.blo[3]), d.bhi[3]);
                    xnb += v * v; gmb = fmax(gmb, fabs(nv - v));
                }
        }
        st.x_norm = sqrt(sc[1] + xnp + xnb);
        double gml = *d.gmax_l;
        if (d.part) {      // one slot per rank behind the scalars (k_sep_pack)
            gml = 0.0;
            for (int r = 0; r < d.world; ++r) gml = fmax(gml, sc[NSCAL + r]);
        }
        st.gmax = fmax(fmax(gmp, gml), gmb);
        st.just_linearized = 0;
        sc[0] = 0.0;   // consumed: later all-reduces of the exchange vector add zeros
        sc[1] = 0.0;
    }
    if (st.iteration == 0) {
        // IterationZero
        st.initial_cost = st.x_cost;
        st.minimum_cost = st.x_cost;
        st.se_minimum = st.se_current = st.se_reference = st.se_candidate = st.x_cost;
        st.se_acc_ref = st.se_acc_cand = 0.0;
        st.se_num_nonmono = 0;
        log_push(d, st, st.x_cost, 0.0, 0.0, 0.0, 0);
        st.num_unsuccessful = 1;   // iteration 0 is recorded with step_is_successful = false
    } else if (st.last_successful) {
        // log row of the successful iteration uses the re-evaluated cost and gradient
        log_push(d, st, st.x_cost, st.cost_change, st.step_norm, st.relative_decrease, 1);
        if (st.x_cost < st.minimum_cost) {
            st.minimum_cost = st.x_cost;
            st.copy_best = st.check_count;
        }
    }
    if (!st.opt.ignore_convergence) {
        if (st.iteration >= st.opt.max_num_iterations) {
            st.terminated = 1; st.termination_type = 1; return;   // NO_CONVERGENCE
        }
        if (st.gmax <= st.opt.gradient_tolerance) {
            st.terminated = 1; st.termination_type = 0; return;
        }
        if (st.radius <= st.opt.min_radius) {
            st.terminated = 1; st.termination_type = 0; return;
        }
    }
    ++st.iteration;
    st.last_successful = 0;
    st.accepted = 0;
    st.ls_alpha = 1.0;
    // step_failed may already carry a landmark-block breakdown from k_schur_windows
}

}  // namespace ssba
