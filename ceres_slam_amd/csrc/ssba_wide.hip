// Long tracks at windowed-path speed: the middle of the iteration for banded problems whose landmarks have up to WSP = 24
// observations (ssba_types.h: WideSys).  Ceres takes whatever graph the dataset holds (tests/dataset_vo.cpp:41-56); stereo
// tracks of 13 .. 24 frames are routine, and the 12-slot windows / 72-row super-blocks of ssba_kernels.hip do not hold them.
// Such problems keep the general (landmark-major) layout for linearisation, back-substitution and trust-region control and
// run, instead of the dense blocked Cholesky of ssba_dense.hip (a chain of n / 64 dependent potrf -> trsm -> syrk launches):
//
//   k_wd_schur     one 512-lane workgroup per window of 24 consecutive free poses: batches of 21 landmarks are half-linearised
//                  by 504 (landmark, slot) producer lanes (W = J_p^T J_l recomputed, never stored; Z = W M^T with C^-1 = M^T M),
//                  staged k-major in LDS (64 x 176 doubles; column 144 carries M g_l) and  S[144 x 145] += Zm^T Zm  is
//                  accumulated output-stationary on v_mfma_f64_16x16x4_f64 by the eight waves: 45 upper tiles + 9 tiles of the
//                  gradient column, 7 / 7 / 7 / 7 / 7 / 7 / 7 / 5 per wave in row segments (the A operand is read once per
//                  segment and k-step).  One slab (54 tiles) per item, every entry one sum in landmark order: no atomics.
//   k_wd_assemble  gather of the slabs into the block-tridiagonal system over super-blocks of 24 poses (D, L row-major
//                  144 x 144, rhs) through host-built lists, fixed order; H_pp on the diagonal blocks
//   k_wd_finish    Jacobi scale at iteration 0, LM damping on the diagonal, identity on the padding rows
//   k_wd_factor    parallel cyclic reduction, step s: every block e factors D_e = U^T U in LDS (16 x 16 tiles: diagonal tile by
//                  one wave with the inverse of its factor built alongside, row panel and trailing update on the matrix cores)
//                  and solves  [YL | YU | yr] = U^-T [S(e, e-s) | S(e, e+s) | r]  with the right-hand-side tiles in accumulator
//                  registers (one 16-column tile per wave, three workgroups per block each repeating the factorisation)
//   k_wd_reduce    one wave per 16 x 16 output tile:  D' = D - YU(e-s)^T YU(e-s) - YL(e+s)^T YL(e+s),  r' likewise,
//                  S'(e, e-2s) = -YU(e-s)^T YL(e-s),  S'(e, e+2s) = -YL(e+s)^T YU(e+s)   (operands straight from L2)
//   k_wd_factor<1> after ceil(log2 n) steps the blocks are decoupled: factor, forward and backward solve, pose step into x0
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "ssba_device.h"
#include "ssba_launch.h"
#include "ssba_types.h"

namespace ssba {

typedef double wd4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ wd4 wmf(double a, double b, wd4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
static __device__ __forceinline__ double wd_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// slab tiles: row tile i <= column tile j <= WNT (column tile WNT = the gradient column); upper tiles of a 9 x 9 block
static __host__ __device__ __forceinline__ int wd_slab_tile(int i, int j) { return i * (WNT + 1) - (i * (i - 1)) / 2 + (j - i); }
static __host__ __device__ __forceinline__ int wd_utile(int i, int j) { return i * WNT - (i * (i - 1)) / 2 + (j - i); }

#ifdef WD_STAMPS      // -DWD_STAMPS (tools/wide_bench.hip, tools/stamps_wide.py): shader-clock stamps of one middle work-group into Dev::dbg
#define WD_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 2) d.dbg[64 * (threadIdx.x >> 6) + (i)] = clock64(); } while (0)
#define WS_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0) d.dbg[256 + 64 * (threadIdx.x >> 8) + (i)] = clock64(); } while (0)
#else
#define WD_STAMP(i) do { } while (0)
#define WS_STAMP(i) do { } while (0)
#endif

// ---- Schur items ---------------------------------------------------------------------------------------------
constexpr int WS_THREADS = 512;
constexpr int WS_BATCH = 21;                    // landmarks per batch: 21 x 24 slots = 504 producer lanes
constexpr int WS_KB = 64;                       // 63 factor rows + one zero row = 16 matrix steps of k = 4
constexpr int WS_RS = 176;                      // row stride of Zm: 352 words = 32 mod 64, the two k-groups of a half-wave read disjoint banks
constexpr int WS_LDS_DOUBLES = WS_KB * WS_RS;   // 90 112 B
constexpr int WS_ITEM_MAX = 128;                // landmarks per item at most (ssba_layout.cpp: build_wide_layout(.., 128, ..))
constexpr int WS_LDS_BYTES = (WS_LDS_DOUBLES + 9 * WS_ITEM_MAX) * 8;

// The products of one batch for wave WV: its tiles are row WV from column WV on (segment A) and, for the waves with short rows, a
// second segment that fills them up (WsTiles); 16 k-steps, the operands of k-step ks + 1 requested before the products of ks are issued.
template <int WV> struct WsTiles {
    static constexpr int rowA = WV, colA0 = WV, nA = WV < 4 ? 7 : WNT + 1 - WV;
    static constexpr int rowB = WV == 4 ? 2 : WV == 5 ? 1 : WV == 6 ? 0 : 8;
    static constexpr int colB0 = WV == 4 ? 9 : WV == 5 ? 8 : WV == 6 ? 7 : 8;
    static constexpr int nB = WV < 4 ? 0 : WV == 4 ? 1 : WV == 5 ? 2 : WV == 6 ? 3 : 2;
};
template <int WV> static __device__ __forceinline__ void wd_products(const double *sZ, int lane, wd4 (&acc)[7]) {
    using W = WsTiles<WV>;
    constexpr int NT = W::nA + W::nB;
    const double *zb = sZ + (lane >> 4) * WS_RS + (lane & 15);
    double aA = zb[16 * W::rowA], aB = W::nB ? zb[16 * W::rowB] : 0.0, b[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) b[q] = zb[16 * (q < W::nA ? W::colA0 + q : W::colB0 + (q - W::nA))];
#pragma unroll
    for (int ks = 0; ks < WS_KB / 4; ++ks) {
        double naA = 0.0, naB = 0.0, nb[NT];
        if (ks + 1 < WS_KB / 4) {
            const double *zr = zb + 4 * (ks + 1) * WS_RS;
            naA = zr[16 * W::rowA];
            if (W::nB) naB = zr[16 * W::rowB];
#pragma unroll
            for (int q = 0; q < NT; ++q) nb[q] = zr[16 * (q < W::nA ? W::colA0 + q : W::colB0 + (q - W::nA))];
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = wmf(q < W::nA ? aA : aB, b[q], acc[q]);
        if (ks + 1 < WS_KB / 4) {
            aA = naA; aB = naB;
#pragma unroll
            for (int q = 0; q < NT; ++q) b[q] = nb[q];
        }
    }
}

// n_zero: the first n_zero work-groups clear D | L of the block-tridiagonal system that k_wd_assemble fills next (a launch less per
// iteration, as in k_schur_windows): the in-place reduction of the previous iteration left products in structural zeros.
__global__ __launch_bounds__(WS_THREADS) void k_wd_schur(Dev d, int n_zero) {
    const State &st = *d.st;
    const WideSys &w = *d.wide;
    const int dead = st.terminated | st.dl_reuse;       // tested once the first operand reads are in flight
    if ((int)blockIdx.x < n_zero) {
        if (dead) return;
        double2 *z = reinterpret_cast<double2 *>(w.xw);
        const size_t n2 = w.off_rhs / 2;
        for (size_t i = (size_t)blockIdx.x * WS_THREADS + threadIdx.x; i < n2; i += (size_t)n_zero * WS_THREADS) z[i] = make_double2(0.0, 0.0);
        return;
    }
    extern __shared__ __align__(16) double wd_lds[];
    double *sZ = wd_lds;
    double *sM = wd_lds + WS_LDS_DOUBLES;       // [c][landmark of the item]: M (6), M g_l (3)
    WS_STAMP(0);
    const int item = (int)blockIdx.x - n_zero;
    const int lb = (int)w.item_begin[item], le = (int)w.item_end[item], base = (int)w.item_base[item];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const bool producer = t < WS_BATCH * WSP;
    const int li = t / WSP, s = t - li * WSP;
    // tiles of this wave: segment A = row wv from column colA0 (nA tiles), segment B fills the waves with short rows
    const int rowA = wv, colA0 = wv, nA = wv < 4 ? 7 : WNT + 1 - wv;
    const int rowB = wv == 4 ? 2 : wv == 5 ? 1 : wv == 6 ? 0 : 8;
    const int colB0 = wv == 4 ? 9 : wv == 5 ? 8 : wv == 6 ? 7 : 8;
    const int nB = wv < 4 ? 0 : wv == 4 ? 1 : wv == 5 ? 2 : wv == 6 ? 3 : 2;
    wd4 acc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[q] = wd4{0.0, 0.0, 0.0, 0.0};

    // columns 145..159 (read by the gradient-column tiles) and the zero row k = 63 stay zero for the whole item
    // (column 144 of the rows 0..62 is rewritten by the producers every batch: not touched here, so no barrier is needed)
    for (int e = t; e < WS_KB * 15; e += WS_THREADS) sZ[(e / 15) * WS_RS + WBD + 1 + (e % 15)] = 0.0;
    if (t <= WBD) sZ[(WS_KB - 1) * WS_RS + t] = 0.0;

    bool pose_ok = false;
    double T[12];
    if (producer) {
        const int f = base + s;
        pose_ok = f < d.nfree;
        if (pose_ok) {
            const int k = d.free_pose[f];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = d.poses[(size_t)k * 12 + i];
        }
    }
    // raw inputs of this lane's (landmark, slot): the observation index two batches ahead, its data one batch ahead -- both
    // stay in flight across the matrix phase of the batch before
    struct Raw { double u, v, dd, p[3]; bool in_range, have; } raw;
    uint32_t e_next = 0xFFFFFFFFu;
    auto fetch_slot = [&](int l0) {
        const int l = l0 + li;
        e_next = (producer && l < le) ? w.slot_obs[(size_t)l * WSP + s] : 0xFFFFFFFFu;
    };
    auto prefetch = [&](int l0) {
        const int l = l0 + li;
        raw.in_range = producer && l < le;
        raw.have = raw.in_range && pose_ok && e_next != 0xFFFFFFFFu;
        if (raw.in_range) {
            if (raw.have) { raw.u = d.dn_u[e_next]; raw.v = d.dn_v[e_next]; raw.dd = d.dn_d[e_next]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) raw.p[c] = d.pts[(size_t)c * d.Lpad + l];
        }
    };
    fetch_slot(lb);
    prefetch(lb);
    fetch_slot(lb + WS_BATCH);
    // the damped 3 x 3 block factor M and u = M g_l of every landmark of the item, by one lane each (it was formed by all 24 slot
    // lanes of a landmark in every batch, on the pipe the matrix instructions need)
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
        for (int c = 0; c < 6; ++c) sM[c * WS_ITEM_MAX + t] = m[c];
        sM[6 * WS_ITEM_MAX + t] = m[0] * lg[0];
        sM[7 * WS_ITEM_MAX + t] = m[1] * lg[0] + m[2] * lg[1];
        sM[8 * WS_ITEM_MAX + t] = m[3] * lg[0] + m[4] * lg[1] + m[5] * lg[2];
    }
    __syncthreads();
    WS_STAMP(1);
    int ws_b = 0;
    (void)ws_b;

    for (int l0 = lb; l0 < le; l0 += WS_BATCH) {
        if (producer) {
            double z[18];       // [c][a]: three runs of six contiguous doubles in the k-major matrix
            double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            const int il = l0 - lb + li;        // landmark of the item (its factor: sM)
            if (raw.in_range) {
#pragma unroll
                for (int c = 0; c < 6; ++c) m[c] = sM[c * WS_ITEM_MAX + il];
            }
            if (s == 0) {       // u = M g_l
#pragma unroll
                for (int c = 0; c < 3; ++c) sZ[(li * 3 + c) * WS_RS + WBD] = raw.in_range ? sM[(6 + c) * WS_ITEM_MAX + il] : 0.0;
            }
            if (raw.have) {
                obs_schur_factor(d, d.S, T, raw.p[0], raw.p[1], raw.p[2], raw.u, raw.v, raw.dd, m, z);       // Z = W M^T
            } else {
#pragma unroll
                for (int i = 0; i < 18; ++i) z[i] = 0.0;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double2 *dz = reinterpret_cast<double2 *>(sZ + (li * 3 + c) * WS_RS + s * 6);     // 48-byte runs, 16-byte aligned
#pragma unroll
                for (int q = 0; q < 3; ++q) dz[q] = make_double2(z[6 * c + 2 * q], z[6 * c + 2 * q + 1]);
            }
        }
        WS_STAMP(2 + 4 * ws_b);
        __syncthreads();
        WS_STAMP(3 + 4 * ws_b);
        prefetch(l0 + WS_BATCH);
        fetch_slot(l0 + 2 * WS_BATCH);
        // One straight-line instantiation per wave (stamps, tools/stamps_wide.py: the rolled form with run-time tile lists branched
        // around every product of a wave with fewer than seven tiles, selected its A operand with vector moves and waited for its
        // nine LDS reads in front of each k-step -- 20 k cycles per batch where the matrix instructions need 13.8 k)
        switch (wv) {
            case 0: wd_products<0>(sZ, lane, acc); break;
            case 1: wd_products<1>(sZ, lane, acc); break;
            case 2: wd_products<2>(sZ, lane, acc); break;
            case 3: wd_products<3>(sZ, lane, acc); break;
            case 4: wd_products<4>(sZ, lane, acc); break;
            case 5: wd_products<5>(sZ, lane, acc); break;
            case 6: wd_products<6>(sZ, lane, acc); break;
            default: wd_products<7>(sZ, lane, acc); break;
        }
        WS_STAMP(4 + 4 * ws_b);
        __syncthreads();
        WS_STAMP(5 + 4 * ws_b);
        if (ws_b < 8) ++ws_b;
    }
    // one slab per item, tile-major: a register of a tile is 512 contiguous bytes
    double *out = w.slab + (size_t)item * WSLAB_DOUBLES;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        if (q >= nA + nB) continue;
        const int ti = q < nA ? rowA : rowB, tj = q < nA ? colA0 + q : colB0 + (q - nA);
        double *pt = out + (size_t)wd_slab_tile(ti, tj) * 256 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) pt[64 * r] = acc[q][r];
    }
    WS_STAMP(60);
}

// The reduction works in place on D, L and r (its steps overwrite whole blocks, structural zeros included), and the gather
// below only stores the structurally non-zero entries: the blocks are cleared before every assembly.
static __device__ __forceinline__ int wd_tri21(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

// one thread per (non-zero 6 x 6 block, element) + one per entry of the reduced gradient
// fuse_finish (single GPU: nothing is exchanged between this kernel and the solve): k_wd_finish's work is done here -- Jacobi
// scale at iteration 0 and LM damping by the lane that writes a diagonal entry, identity rows of the padding by lanes of their own
__global__ __launch_bounds__(256) void k_wd_assemble(Dev d, int fuse_finish) {
    const State &st = *d.st;
    const WideSys &w = *d.wide;
    const int dead = st.terminated | st.dl_reuse;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_el = (size_t)w.n_blk * 36;
    const size_t blk = (size_t)WBD * WBD;
    if (gid < n_el) {
        const uint32_t b = (uint32_t)(gid / 36);
        const int e = (int)(gid - (size_t)b * 36);
        const int r = e / 6, c = e - r * 6;
        const uint32_t fa = w.blk_a[b], fb = w.blk_b[b];
        const uint32_t ib = w.blk_start[b], ie = w.blk_start[b + 1];
        if (dead) return;
        int er = r, ec = c;
        if (fa == fb && c < r) { er = c; ec = r; }      // the upper entry of a diagonal block: always inside a computed tile
        double v = 0.0;
        for (uint32_t i0 = ib; i0 < ie; i0 += 8) {
            uint32_t cw[8];
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) cw[q] = i0 + q < ie ? w.blk_contrib[i0 + q] : 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                x[q] = 0.0;
                if (cw[q] != 0xFFFFFFFFu) {
                    const uint32_t it = cw[q] / (WSP * WSP), sp = cw[q] - it * (WSP * WSP);
                    const int sa = (int)(sp / WSP), sb = (int)(sp - (uint32_t)sa * WSP);
                    const int row = 6 * sa + er, col = 6 * sb + ec;
                    x[q] = w.slab[(size_t)it * WSLAB_DOUBLES + (size_t)wd_slab_tile(row >> 4, col >> 4) * 256 + (row & 15) * 16 + (col & 15)];
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v += x[q];
        }
        v = -v;
        if (fa == fb) {
            const double h = d.hpp[(size_t)d.free_pose[fa] * 21 + wd_tri21(min(r, c), max(r, c))];
            v += h;
            if (fuse_finish && r == c) {
                const size_t i = (size_t)fa * 6 + r;
                double sc;
                if (st.iteration == 0) { sc = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0; d.sp[i] = sc; }
                else sc = d.sp[i];
                const double s2 = sc * sc;
                v += fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
            }
        }
        const uint32_t Ia = fa / WSP, Ib = fb / WSP;
        const int row = (int)(fa - Ia * WSP) * 6 + r, col = (int)(fb - Ib * WSP) * 6 + c;
        if (Ia == Ib) {
            w.xw[(size_t)Ia * blk + (size_t)row * WBD + col] = v;
            if (fa != fb) w.xw[(size_t)Ia * blk + (size_t)col * WBD + row] = v;
        } else {        // S(Ia rows, Ib columns), Ib = Ia + 1: stored in L[Ib] = S(Ib, Ia) at (col, row)
            w.xw[w.off_L + (size_t)Ib * blk + (size_t)col * WBD + row] = v;
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
        for (uint32_t j = w.prow_start[f]; j < w.prow_start[f + 1]; ++j) {
            const uint32_t cw = w.prow_contrib[j], it = cw / WSP;
            const int row = 6 * (int)(cw - it * WSP) + c;
            v -= w.slab[(size_t)it * WSLAB_DOUBLES + (size_t)wd_slab_tile(row >> 4, WNT) * 256 + (row & 15) * 16];
        }
        w.xw[w.off_rhs + i] = -v;              // right-hand side = -reduced gradient (linear in the ranks' partial sums)
        d.xv[d.off_gp + i] = g;
        d.xv[d.off_hdiag + i] = d.hpp[(size_t)k * 21 + wd_tri21(c, c)];
    } else if (fuse_finish && gid < n_el + (size_t)w.n * WBD) {        // padding of the last super-block
        const size_t i = gid - n_el;
        const size_t I = i / WBD, r = i - I * WBD;
        w.xw[I * blk + r * WBD + r] = 1.0;
        w.xw[w.off_rhs + i] = 0.0;
    }
}

__global__ __launch_bounds__(256) void k_wd_finish(Dev d) {
    const State &st = *d.st;
    const WideSys &w = *d.wide;
    if (st.terminated || st.dl_reuse) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= w.n * WBD) return;
    const int I = i / WBD, r = i - I * WBD;
    double *Dd = w.xw + (size_t)I * WBD * WBD + (size_t)r * WBD + r;
    if (i < 6 * d.nfree) {
        const double h = d.xv[d.off_hdiag + i];
        if (st.iteration == 0) d.sp[i] = st.opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
        const double s = d.sp[i], s2 = s * s;
        *Dd += fmin(fmax(h * s2, st.opt.min_lm_diag), st.opt.max_lm_diag) / (damp_radius(st) * s2);
    } else {        // padding of the last super-block (its off-diagonal entries were allocated as zeros and are never written)
        *Dd = 1.0;
        w.xw[w.off_rhs + i] = 0.0;
    }
}

// Lanes of ONE wave hand values to each other through LDS below (the LDS unit serves a wave's requests in issue order, so no
// hardware wait is needed) -- but to the compiler a lane that skips a predicated store has not changed memory, and it reuses
// what it loaded before the store (measured: the loads were sunk into the `if (lane < 16)` block, the other 48 lanes read
// stale registers).  A compiler-level memory barrier makes it store what is pending and load again.
#define WD_WAVE_LDS_SYNC() do { asm volatile("" ::: "memory"); } while (0)

// ---- parallel cyclic reduction over 144-row blocks ------------------------------------------------------------
constexpr int WF_THREADS = 512;
constexpr int WF_NU = WNT * (WNT + 1) / 2;      // 45 upper tiles of U
// U tiles | U_kk^-T tiles (decoupled step) | sub-step operands P, Q of two diagonal tiles | 1 / sqrt(pivot) of two | y | x | t
constexpr int WF_OFF_W = WF_NU * 256, WF_OFF_P = WF_OFF_W + WNT * 256, WF_OFF_Q = WF_OFF_P + 2 * 256, WF_OFF_RS = WF_OFF_Q + 2 * 256,
              WF_OFF_Y = WF_OFF_RS + 32, WF_OFF_X = WF_OFF_Y + WBD, WF_OFF_T = WF_OFF_X + WBD;
constexpr int WF_LDS_DOUBLES = WF_OFF_T + 16;

// Diagonal tile T (16 x 16, accumulator layout: register q of lane (g, j) = row 4 q + g, column j), four sub-steps of four
// pivots -- the scheme of ssba_bcr_mfma.hip's factor_tile: the 4 x 4 pivot block is factored LDL^T "uniformly" (every lane
// computes the same scalars from v_readlane copies; the serial chain is four reciprocals), the inverse M of its unit-lower
// factor becomes the A operand P of ONE instruction  c = M T[rows]  and the scaled pivot rows Q = -c / d the A operand of ONE
// instruction that updates the rows below.  Every other tile of the block row (panel tiles, right-hand sides) follows with the
// same two operands per sub-step (wd_apply).  Rows stay in the LDL^T scaling; rs = 1 / sqrt(d) normalises them afterwards.
// sP / sQ: [sub-step][lane]; srs: 16 reciprocal square roots.  Returns false on a non-positive / non-finite pivot.
static __device__ __forceinline__ bool wd_factor_tile(wd4 &T, double *__restrict__ sP, double *__restrict__ sQ, double *__restrict__ srs, int lane) {
    const int g = lane >> 4, j = lane & 15;
    const double kDiag = (j < 4 && g == j) ? 1.0 : 0.0, k10 = (j == 1 && g == 0) ? 1.0 : 0.0, k20 = (j == 2 && g == 0) ? 1.0 : 0.0,
                 k21 = (j == 2 && g == 1) ? 1.0 : 0.0, k30 = (j == 3 && g == 0) ? 1.0 : 0.0, k31 = (j == 3 && g == 1) ? 1.0 : 0.0,
                 k32 = (j == 3 && g == 2) ? 1.0 : 0.0;
    const double kG0 = g == 0 ? 1.0 : 0.0, kG1 = g == 1 ? 1.0 : 0.0, kG2 = g == 2 ? 1.0 : 0.0, kG3 = g == 3 ? 1.0 : 0.0;
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = 4 * r;      // S[a][c] = T[4r + a][4r + c] sits in register r of lane 16 a + 4r + c
        const double tr = T[r];
        const double s00 = wd_readlane(tr, b), s10 = wd_readlane(tr, 16 + b), s20 = wd_readlane(tr, 32 + b), s30 = wd_readlane(tr, 48 + b);
        const double s11 = wd_readlane(tr, 16 + b + 1);
        double s21 = wd_readlane(tr, 32 + b + 1), s31 = wd_readlane(tr, 48 + b + 1);
        const double s22 = wd_readlane(tr, 32 + b + 2);
        double s32 = wd_readlane(tr, 48 + b + 2);
        const double s33 = wd_readlane(tr, 48 + b + 3);
        const double x0 = __builtin_amdgcn_rcp(s00), e0 = fma(-s00, x0, 1.0);
        const double t10 = s10 * x0, t20 = s20 * x0, t30 = s30 * x0;
        const double l10 = fma(t10, e0, t10), l20 = fma(t20, e0, t20), l30 = fma(t30, e0, t30);
        const double d1 = fma(-s10, l10, s11);
        s21 = fma(-s20, l10, s21); s31 = fma(-s30, l10, s31);
        const double x1 = __builtin_amdgcn_rcp(d1), e1 = fma(-d1, x1, 1.0);
        const double t21 = s21 * x1, t31 = s31 * x1;
        const double l21 = fma(t21, e1, t21), l31 = fma(t31, e1, t31);
        const double d2 = fma(-s21, l21, fma(-s20, l20, s22));
        s32 = fma(-s31, l21, fma(-s30, l20, s32));
        const double x2 = __builtin_amdgcn_rcp(d2), e2 = fma(-d2, x2, 1.0);
        const double t32 = s32 * x2;
        const double l32 = fma(t32, e2, t32);
        const double d3 = fma(-s32, l32, fma(-s31, l31, fma(-s30, l30, s33)));
        const double x3 = __builtin_amdgcn_rcp(d3), e3 = fma(-d3, x3, 1.0);
        const double rc0 = fma(x0, e0, x0), rc1 = fma(x1, e1, x1), rc2 = fma(x2, e2, x2), rc3 = fma(x3, e3, x3);
        ok = ok && rc0 > 0.0 && rc1 > 0.0 && rc2 > 0.0 && rc3 > 0.0 && rc0 < INFINITY && rc1 < INFINITY && rc2 < INFINITY && rc3 < INFINITY;
        // M = l^-1 (unit lower); P: lane (g, i = j) holds M[i][g] for i < 4
        const double n20 = fma(l21, l10, -l20);
        const double n31 = fma(l32, l21, -l31);
        const double n30 = -fma(l32, n20, fma(l31, -l10, l30));
        double Pv = fma(-l10, k10, kDiag);
        Pv = fma(n20, k20, Pv);
        Pv = fma(-l21, k21, Pv);
        Pv = fma(n31, k31, Pv);
        Pv = fma(-l32, k32, Pv);
        Pv = fma(n30, k30, Pv);
        const wd4 y4 = wmf(Pv, tr, wd4{0.0, 0.0, 0.0, 0.0});
        const double rcg = fma(rc3, kG3, fma(rc2, kG2, fma(rc1, kG1, rc0 * kG0)));
        const double qs = (j > b + 3) ? -rcg : 0.0;
        const double y = y4[0];             // lane (g, j): unnormalised pivot row c_g[j]
        T[r] = y;
        const double Qv = y * qs;
        if (r < 3) T = wmf(Qv, y, T);
        sP[r * 64 + lane] = Pv;
        sQ[r * 64 + lane] = Qv;
        if (j == 0) srs[b + g] = sqrt(rcg);
    }
    return ok;
}
// the four sub-steps on another tile of the block row (accumulator layout), then the rows normalised by 1 / sqrt(pivot)
static __device__ __forceinline__ void wd_apply(wd4 &X, const double *__restrict__ sP, const double *__restrict__ sQ, const double *__restrict__ srs, int lane) {
    double P[4], Q[4], rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { P[r] = sP[r * 64 + lane]; Q[r] = sQ[r * 64 + lane]; rs[r] = srs[4 * r + (lane >> 4)]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const wd4 y4 = wmf(P[r], X[r], wd4{0.0, 0.0, 0.0, 0.0});
        X[r] = y4[0];
        if (r < 3) X = wmf(Q[r], y4[0], X);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] *= rs[r];
}
static __device__ __forceinline__ wd4 wd_tile_load(const double *__restrict__ T, int lane) {
    wd4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = T[(4 * q + (lane >> 4)) * 16 + (lane & 15)];
    return v;
}
static __device__ __forceinline__ void wd_tile_store(double *__restrict__ T, const wd4 &v, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) T[(4 * q + (lane >> 4)) * 16 + (lane & 15)] = v[q];
}
// T_ij -= U_ki^T U_kj  (normalised row tiles, row-major in LDS); returns the updated tile
static __device__ __forceinline__ wd4 wd_tile_update(const double *__restrict__ Uki, const double *__restrict__ Ukj, const double *__restrict__ Tij, int lane) {
    const int kq = lane >> 4, i = lane & 15;
    double a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { a[s] = -Uki[(4 * s + kq) * 16 + i]; b[s] = Ukj[(4 * s + kq) * 16 + i]; }
    wd4 acc = wd_tile_load(Tij, lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = wmf(a[s], b[s], acc);
    return acc;
}

// FINAL = 0: a step of the reduction (grid n x ng, the right-hand-side tiles of [L | U | r] dealt to waves 1..7 of ng
// workgroups);  FINAL = 1: the decoupled last step (grid n): factor, forward and backward solve, pose step into x0.
// Schedule of a block row k: (b) the panel tiles (k, j > k) and the right-hand-side tiles take the sub-steps of diagonal tile
// k; (c) wave 0 updates diagonal tile k + 1 and factors it at once (look-ahead) while the other waves update the rest of the
// trailing tiles and the right-hand sides -- two workgroup barriers per block row, the pivot chain never waits for an update.
template <int FINAL>
__global__ __launch_bounds__(WF_THREADS) void k_wd_factor(Dev d, int step, int ng) {
    const WideSys &w = *d.wide;
    const StateFlags sf = state_flags_vmem(d.st);
    extern __shared__ __align__(16) double wf_lds[];
    double *sU = wf_lds, *sW = wf_lds + WF_OFF_W, *sP = wf_lds + WF_OFF_P, *sQ = wf_lds + WF_OFF_Q, *sRS = wf_lds + WF_OFF_RS;
    double *sy = wf_lds + WF_OFF_Y, *sx = wf_lds + WF_OFF_X, *stv = wf_lds + WF_OFF_T;
    __shared__ int s_bad;
    const int e = (int)blockIdx.x / ng, part = (int)blockIdx.x - e * ng;
    const int n = w.n, s = 1 << step;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, g = lane >> 4, jj = lane & 15;
    const size_t blk = (size_t)WBD * WBD;
    const bool hasL = !FINAL && e - s >= 0, hasU = !FINAL && e + s < n;
    const double *Dg = w.xw + (size_t)e * blk;
    const double *Lg = w.xw + w.off_L + (size_t)e * blk;
    // S(e, e + s): the transpose of L[e + 1] in the first step, the reduce kernel's output afterwards
    const double *Ug = step == 0 ? w.xw + w.off_L + (size_t)(hasU ? e + 1 : e) * blk : w.U + (size_t)e * blk;
    const double *rg = w.xw + w.off_rhs + (size_t)e * WBD;
    // this wave's column tile of [L | U | r]: waves 1..7 (wave 0 factors the diagonal tiles: it carries none)
    const int c = wv == 0 ? -1 : FINAL ? (wv == 1 ? 2 * WNT : -1) : part + ng * (wv - 1);
    const bool act = c >= 0 && c <= 2 * WNT && (c < WNT ? hasL : c < 2 * WNT ? hasU : true);
    if (t == 0) s_bad = 0;
    WD_STAMP(0);
    wd4 rt[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) rt[j] = wd4{0.0, 0.0, 0.0, 0.0};
    if (act) {
        if (c == 2 * WNT) {
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) rt[j][q] = jj == 0 ? rg[16 * j + 4 * q + g] : 0.0;
        } else if (c < WNT) {
            const double *pp = Lg + (size_t)g * WBD + 16 * c + jj;
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) rt[j][q] = pp[(16 * j + 4 * q) * WBD];
        } else if (step == 0) {
            const double *pp = Ug + (size_t)(16 * (c - WNT) + jj) * WBD + g;
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) rt[j][q] = pp[16 * j + 4 * q];
        } else {
            const double *pp = Ug + (size_t)g * WBD + 16 * (c - WNT) + jj;
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) rt[j][q] = pp[(16 * j + 4 * q) * WBD];
        }
    }
    // upper tiles of D into LDS, tile by tile (row-major 16 x 16); wave 0 keeps the first diagonal tile in registers
    // (r04, stamps of tools/wide_bench.hip -DWD_STAMPS: a rolled loop waited for every tile before it asked for the next one --
    // six dependent round trips per wave, 13.5 k of the launch's 95 k cycles; all of a wave's tiles are in flight together now)
    wd4 dg = {0.0, 0.0, 0.0, 0.0};
    {
        constexpr int WF_TPW = (WF_NU + 7) / 8;       // tiles per wave: 6
        wd4 v[WF_TPW];
#pragma unroll
        for (int m = 0; m < WF_TPW; ++m) {
            const int tl = wv + 8 * m;
            if (tl < WF_NU) {
                int ti = 0, rem = tl;                   // tile tl = (ti, tj) of the upper triangle, row by row
                while (rem >= WNT - ti) { rem -= WNT - ti; ++ti; }
                const int tj = ti + rem;
                const double *pp = Dg + (size_t)(16 * ti + g) * WBD + 16 * tj + jj;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[m][q] = pp[(size_t)4 * q * WBD];
            }
        }
#pragma unroll
        for (int m = 0; m < WF_TPW; ++m) {
            const int tl = wv + 8 * m;
            if (tl >= WF_NU) continue;
            if (tl == 0) dg = v[m];
            else wd_tile_store(sU + tl * 256, v[m], lane);
        }
    }
    if (sf.dead()) return;
    WD_STAMP(1);
    if (wv == 0) {
        if (!wd_factor_tile(dg, sP, sQ, sRS, lane)) s_bad = 1;
    }
    WD_STAMP(2);
    __syncthreads();
    WD_STAMP(3);

#pragma unroll
    for (int k = 0; k < WNT; ++k) {
        const double *kP = sP + (k & 1) * 256, *kQ = sQ + (k & 1) * 256, *kRS = sRS + (k & 1) * 16;
        // (b) the panel tiles of block row k (one per wave) and block row k of the right-hand sides
        {
            const int j = k + 1 + wv;
            if (j < WNT) {
                wd4 x = wd_tile_load(sU + wd_utile(k, j) * 256, lane);
                wd_apply(x, kP, kQ, kRS, lane);
                wd_tile_store(sU + wd_utile(k, j) * 256, x, lane);
            }
            if (FINAL && wv == 2) {        // U_kk^-T (row-major) for the backward solve: the sub-steps applied to the identity
                wd4 x;
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = (4 * q + g == jj) ? 1.0 : 0.0;
                wd_apply(x, kP, kQ, kRS, lane);
                wd_tile_store(sW + k * 256, x, lane);
            }
        }
        WD_STAMP(4 + 4 * k);
        __syncthreads();
        WD_STAMP(5 + 4 * k);
        // (c) wave 0: diagonal tile k + 1, updated and factored at once; the others: the rest of the trailing tiles
        //     (i, j), k < i <= j, and the right-hand sides below block row k
        if (wv == 0) {
            if (k + 1 < WNT) {
                wd4 x = wd_tile_update(sU + wd_utile(k, k + 1) * 256, sU + wd_utile(k, k + 1) * 256, sU + wd_utile(k + 1, k + 1) * 256, lane);
                if (!wd_factor_tile(x, sP + ((k + 1) & 1) * 256, sQ + ((k + 1) & 1) * 256, sRS + ((k + 1) & 1) * 16, lane)) s_bad = 1;
            }
        } else {
            int cnt = -1;
#pragma unroll
            for (int i = k + 1; i < WNT; ++i)
#pragma unroll
                for (int j = i; j < WNT; ++j, ++cnt) {
                    if (cnt < 0) continue;          // (k + 1, k + 1) is wave 0's
                    // (wave 4 shares its SIMD -- one fp64 pipe -- with the pivot wave: it takes no trailing tiles; 91.0 k -> 87.9 k cycles)
                    if (wv != 4 && cnt % 6 == (wv < 4 ? wv - 1 : wv - 2)) {
                        const wd4 x = wd_tile_update(sU + wd_utile(k, i) * 256, sU + wd_utile(k, j) * 256, sU + wd_utile(i, j) * 256, lane);
                        wd_tile_store(sU + wd_utile(i, j) * 256, x, lane);
                    }
                }
            // block row k of the right-hand sides: here, under the pivot wave's factorisation, not in phase (b) where that wave waits
            // for it (stamps, tools/wide_bench.hip -DWD_STAMPS: 87.9 k -> 83.8 k cycles per launch)
            if (act) wd_apply(rt[k], kP, kQ, kRS, lane);
            if (act) {
#pragma unroll
                for (int j = k + 1; j < WNT; ++j) {
                    double a[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = -sU[wd_utile(k, j) * 256 + (4 * q + g) * 16 + jj];
#pragma unroll
                    for (int q = 0; q < 4; ++q) rt[j] = wmf(a[q], rt[k][q], rt[j]);
                }
            }
        }
        WD_STAMP(6 + 4 * k);
        __syncthreads();
        WD_STAMP(7 + 4 * k);
    }
    if (s_bad) {        // non-positive pivot: Cholesky breakdown, the step is rejected (Ceres: LM retries with a smaller radius)
        if (t == 0) d.st->step_failed = 1;
        return;
    }
    if (!FINAL) {
        if (!act) return;
        WD_STAMP(40);
        if (c == 2 * WNT) {
            double *py = w.yr + (size_t)e * WBD;
            if (jj == 0) {
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) py[16 * j + 4 * q + g] = rt[j][q];
            }
        } else {
            double *pp = (c < WNT ? w.YL : w.YU) + (size_t)e * blk + (size_t)g * WBD + 16 * (c < WNT ? c : c - WNT) + jj;
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) pp[(16 * j + 4 * q) * WBD] = rt[j][q];
        }
        return;
    }
    // ---- decoupled block: U x = y by block rows from the bottom (the wave that holds y) ------------------------
    if (wv != 1) return;
    if (jj == 0) {
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) sy[16 * j + 4 * q + g] = rt[j][q];
    }
    WD_WAVE_LDS_SYNC();
    const int i = lane & 15, pt = lane >> 4;
#pragma unroll
    for (int k = WNT - 1; k >= 0; --k) {
        double p = 0.0;
#pragma unroll
        for (int j = k + 1; j < WNT; ++j) {
            const double *Ukj = sU + wd_utile(k, j) * 256 + i * 16;
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) p += Ukj[pt + 4 * cq] * sx[16 * j + pt + 4 * cq];
        }
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        if (lane < 16) stv[i] = sy[16 * k + i] - p;
        WD_WAVE_LDS_SYNC();
        const double *Wk = sW + k * 256 + i;        // (U_kk^-1)[i][c] = (U_kk^-T)[c][i]
        double xk = 0.0;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) xk += Wk[(pt + 4 * cq) * 16] * stv[pt + 4 * cq];
        xk += __shfl_xor(xk, 16, 64);
        xk += __shfl_xor(xk, 32, 64);
        if (lane < 16) sx[16 * k + i] = xk;
        WD_WAVE_LDS_SYNC();
    }
    for (int r = lane; r < WBD; r += 64) {
        const size_t row = (size_t)e * WBD + r;
        if (row < (size_t)d.nfree * 6) d.x0[row] = sx[r];
    }
}

// one wave per 16 x 16 output tile of block e: jobs 0..44 tiles of D' (upper), 45..53 block rows of r', 54..134 tiles of
// S'(e, e - 2s), 135..215 tiles of S'(e, e + 2s)
constexpr int WR_JOBS = WF_NU + WNT + 2 * WNT * WNT;        // 216
constexpr int WR_WG_PER_BLOCK = WR_JOBS / 4;                // 54 workgroups of four waves

// acc -= A[:, 16 ti ..]^T B[:, 16 tj ..]  over the 144 rows (B = a vector in column 0 when bvec)
static __device__ __forceinline__ wd4 wd_product(const double *__restrict__ A, const double *__restrict__ B, int ti, int tj, bool bvec, wd4 acc, int lane) {
    const int kq = lane >> 4, i = lane & 15;
    const double *pa = A + (size_t)kq * WBD + 16 * ti + i;
    double a[WBD / 4], b[WBD / 4];
    if (bvec) {
#pragma unroll
        for (int ks = 0; ks < WBD / 4; ++ks) { a[ks] = pa[(size_t)4 * ks * WBD]; b[ks] = i == 0 ? B[4 * ks + kq] : 0.0; }
    } else {
        const double *pb = B + (size_t)kq * WBD + 16 * tj + i;
#pragma unroll
        for (int ks = 0; ks < WBD / 4; ++ks) { a[ks] = pa[(size_t)4 * ks * WBD]; b[ks] = pb[(size_t)4 * ks * WBD]; }
    }
#pragma unroll
    for (int ks = 0; ks < WBD / 4; ++ks) acc = wmf(-a[ks], b[ks], acc);
    return acc;
}

__global__ __launch_bounds__(256, 2) void k_wd_reduce(Dev d, int step) {
    const WideSys &w = *d.wide;
    const StateFlags sf = state_flags_vmem(d.st);
    const int e = (int)blockIdx.x / WR_WG_PER_BLOCK;
    const int lane = threadIdx.x & 63, g = lane >> 4, jj = lane & 15;
    const int job = ((int)blockIdx.x - e * WR_WG_PER_BLOCK) * 4 + (threadIdx.x >> 6);
    const int n = w.n, s = 1 << step, prev = e - s, next = e + s;
    const size_t blk = (size_t)WBD * WBD;
    const bool hasP = prev >= 0, hasN = next < n;
    if (sf.dead()) return;
    if (job < WF_NU + WNT) {
        if (!hasP && !hasN) return;
        const bool vec = job >= WF_NU;
        int ti = 0, tj = 0;
        if (vec) { ti = job - WF_NU; tj = 0; }
        else { int q = job; while (q >= WNT - ti) { q -= WNT - ti; ++ti; } tj = ti + q; }
        wd4 acc;
        double *out = vec ? w.xw + w.off_rhs + (size_t)e * WBD + 16 * ti + g : w.xw + (size_t)e * blk + (size_t)(16 * ti + g) * WBD + 16 * tj + jj;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = vec ? (jj == 0 ? out[4 * q] : 0.0) : out[(size_t)4 * q * WBD];
        if (hasP) acc = wd_product(w.YU + (size_t)prev * blk, vec ? w.yr + (size_t)prev * WBD : w.YU + (size_t)prev * blk, ti, tj, vec, acc, lane);
        if (hasN) acc = wd_product(w.YL + (size_t)next * blk, vec ? w.yr + (size_t)next * WBD : w.YL + (size_t)next * blk, ti, tj, vec, acc, lane);
        if (vec) {
            if (jj == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) out[4 * q] = acc[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) out[(size_t)4 * q * WBD] = acc[q];
        }
        return;
    }
    const bool lower = job < WF_NU + WNT + WNT * WNT;
    const int q2 = job - (WF_NU + WNT) - (lower ? 0 : WNT * WNT);
    const int ti = q2 / WNT, tj = q2 - ti * WNT;
    wd4 acc = {0.0, 0.0, 0.0, 0.0};
    double *out;
    if (lower) {        // S'(e, e - 2s) = -YU(prev)^T YL(prev): needs prev and its own lower neighbour
        if (!hasP || prev - s < 0) return;
        acc = wd_product(w.YU + (size_t)prev * blk, w.YL + (size_t)prev * blk, ti, tj, false, acc, lane);
        out = w.xw + w.off_L + (size_t)e * blk;
    } else {            // S'(e, e + 2s) = -YL(next)^T YU(next)
        if (!hasN || next + s >= n) return;
        acc = wd_product(w.YL + (size_t)next * blk, w.YU + (size_t)next * blk, ti, tj, false, acc, lane);
        out = w.U + (size_t)e * blk;
    }
    out += (size_t)(16 * ti + g) * WBD + 16 * tj + jj;
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(size_t)4 * q * WBD] = acc[q];
}

// ---- host side ---------------------------------------------------------------------------------------------------
int configure_wide() {
    if (hipFuncSetAttribute((const void *)k_wd_schur, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_wd_factor<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WF_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    return hipFuncSetAttribute((const void *)k_wd_factor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WF_LDS_DOUBLES * sizeof(double))) == hipSuccess ? 0 : -1;
}
// fuse_finish: see k_wd_assemble (then launch_wide_finish(.., true) launches nothing)
void launch_wide_schur(Launcher &L, const Dev &d, bool fuse_finish) {
    const WideSys &w = L.wide;
    const int n_zero = std::min(64, std::max(1, w.n * 4));
    LAUNCH(KC_SCHUR, k_wd_schur, dim3(w.n_items + n_zero), dim3(WS_THREADS), WS_LDS_BYTES, d, n_zero);
    LAUNCH(KC_ASSEMBLE, k_wd_assemble, dim3((unsigned)(((size_t)w.n_blk * 36 + (size_t)w.n * WBD + 255) / 256)), dim3(256), 0, d, fuse_finish ? 1 : 0);
}
void launch_wide_finish(Launcher &L, const Dev &d, bool fused) {
    if (fused) return;
    LAUNCH(KC_SMALL, k_wd_finish, dim3((L.wide.n * WBD + 255) / 256), dim3(256), 0, d);
}
void launch_wide_solve(Launcher &L, const Dev &d) {
    const WideSys &w = L.wide;
    const int ng = 3;
    for (int q = 0; q < w.steps; ++q) {
        LAUNCH(KC_BCR_FACTOR, k_wd_factor<0>, dim3(w.n * ng), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, q, ng);
        LAUNCH(KC_BCR_REDUCE, k_wd_reduce, dim3(w.n * WR_WG_PER_BLOCK), dim3(256), 0, d, q);
    }
    LAUNCH(KC_BCR_FACTOR, k_wd_factor<1>, dim3(w.n), dim3(WF_THREADS), WF_LDS_DOUBLES * sizeof(double), d, w.steps, 1);
}

}  // namespace ssba
