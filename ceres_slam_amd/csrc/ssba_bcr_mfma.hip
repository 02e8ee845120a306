// Block factor / reduce kernels of the reduced-camera solve on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Same contract as the first-generation kernels in ssba_bcr.hip (kept there as the cross-check, SSBA_BCR_LEGACY=1):
//   k_bcr_factor_mf:  D = G G^T ; YL = G^-1 L ; YU = G^-1 U^T ; yr = G^-1 r      (G stored lower, 1/G_kk on its diagonal)
//   k_bcr_reduce_mf:  D' = D - YU^T YU - YL^T YL ; L' = -YU^T YL ; r' = r - YU^T yr - YL^T yr
//
// Factor.  The 72 x 72 block is factored as U^T U on the UPPER triangle in 16-wide column tiles that live in the MFMA
// accumulator layout (lane (g, j) = (lane >> 4, lane & 15), register q: row 4q + g, column j of the tile) for the whole
// kernel.  That layout is closed under everything the factorisation needs:
//   * a finished row panel X (rows = the pivots k) is, register by register, already the A *and* the B operand of the
//     trailing update  T_ij -= X_i^T X_j  (k-step s of the instruction sums over the rows {4s + g} = register s);
//   * the panel solve runs in sub-steps of four pivots: the 4 x 4 pivot block is factored LDL^T "uniformly" (every lane
//     computes the same 30 scalars from v_readlane copies -- the only serial chain of the kernel: four reciprocals),
//     its unit-lower inverse M becomes the A operand P of ONE instruction  c = M T[rows]  and the scaled pivot rows
//     Q = -c / d the A operand of ONE instruction that updates the rows below; every other tile of the block row
//     ([D | r] to the right of the diagonal, [L | U^T]) follows with the same two operands.
// Only P, Q (512 B each) and the finished D panels (as A operands of other waves' updates) go through LDS; the
// right-hand sides never leave their registers.  Square roots are off the chain: rows stay in the LDL^T scaling and
// are multiplied by 1/sqrt(d) when they are stored.
// Work split: a workgroup = 4 waves (one per SIMD; fp64 MFMA reaches its rate from one wave).  Wave w owns column w of
// [D | r] (wave 0: columns 0 and 4) and factors its diagonal tile one step ahead of the others' trailing updates; the nine
// column tiles of [L | U^T] are dealt to the waves of gridDim.y workgroups (each of which repeats the cheap D part), so a
// chain of 84 blocks fills 252 CUs.
//
// Reduce.  Operands staged once in LDS (row stride 80: columns 72.. hold yr and zeros, so r' falls out of the
// same products), output tiles dealt to the waves, K = 72 = 18 instructions per tile and product.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <algorithm>

#include "ssba_device.h"
#include "ssba_launch.h"
#include "ssba_types.h"

namespace ssba {

typedef double mf_d4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ mf_d4 mf(double a, double b, mf_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
static __device__ __forceinline__ double mf_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double mf_rcp(double a) {      // hardware estimate (~2^-26) + one Newton step
    double r = __builtin_amdgcn_rcp(a);
    return fma(fma(-a, r, 1.0), r, r);
}
static __device__ __forceinline__ double mf_rsqrt(double a) {
    double r = __builtin_amdgcn_rsq(a);
    r = r * (1.5 - 0.5 * a * r * r);
    r = r * (1.5 - 0.5 * a * r * r);
    return r;
}

#ifdef SSBA_STAMPS
#define MF_STAMP(i) do { if (bx == 3 && by == 0 && (threadIdx.x & 63) == 0) d.dbg[(threadIdx.x >> 6) * 64 + (i)] = clock64(); } while (0)
// phases of EVERY workgroup (its thread 0) on the chip-wide 100 MHz clock: 0 entry, 1 loaded, 2 factored, 3 staged, 4 done
#define MF_WG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 500) d.dbg[4096 + 8 * (int)blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define MF_STAMP(i) do { } while (0)
#define MF_WG_STAMP(k) do { } while (0)
#endif

// XCD-aware placement: workgroups go to the eight XCDs round-robin by linear id (id % 8), and each XCD has its own L2.
// The workgroups that share a block's operands should sit on one XCD so that the operands cross the fabric once per
// XCD.  The n x ny (block, part) pairs are listed grouped by block % 8 and XCD c takes a contiguous run of that list
// (its ids c, c + 8, ...): exactly n x ny ids, no padding workgroups.  id -> (block e, part y).
static __device__ __forceinline__ void xcd_map(int id, int n, int ny, int &e, int &y) {
    // (no divisions: this runs in the prologue of every launch, in front of the first load)
    const int N = n * ny, c = id & 7;
    const int q = (N + 7) >> 3, r = (N + 7) & 7;                // ids of XCD k: (N - k + 7) >> 3 = q for k <= r, q - 1 beyond
    int idx = (id >> 3) + c * q - max(0, c - (r + 1));          // + the ids of the XCDs before c
    e = 0; y = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sk = ((n - k + 7) >> 3) * ny;                 // pairs of the blocks with e % 8 == k
        if (idx < sk) {
            const int bq = ny == 1 ? idx : ny == 2 ? (idx >> 1) : ny == 3 ? ((idx * 0xAAAB) >> 17) : ((idx * 0x4000) >> 16);     // idx / ny, idx < 2^15, ny <= 4
            y = idx - bq * ny;
            e = k + 8 * bq;
            return;
        }
        idx -= sk;
    }
}
static int xcd_grid(int n, int ny) { return n * ny; }

constexpr int MF_THREADS = 256;
constexpr int NDT = 5;          // tile rows / column tiles of [D | r | 0]
constexpr int NRT = 9;          // column tiles of [L | U^T]

struct FactorOps {
    const double *Dg, *Lg, *Ug, *rin, *Bg;
    double *oD, *oYL, *oYU, *orr, *saveU, *oYB, *xsol;
    int f0;                                 // first free pose of the block (solve + pose update)
    bool hasL, hasU, trL, trU;
    // fused steps (PcrFused): D = Dg - GA - GB, r = rin - ga - gb (null: nothing to subtract), couplings = sgn x [Lg | Ug];
    // nD / nr: where the assembled D (upper tiles) / r go for the next step; the Gram products of this block
    const double *GA, *GB, *ga, *gb;
    double sgnL, sgnU;
    double *nD, *nr, *oGLL, *oGUU, *oGUL, *oGULT, *ogL, *ogU;
    // chains with pinned ends: saveL (like saveU) keeps a coupling that will not be folded again; a pinned block is only
    // assembled (and, the last one, tracks its coupling to the first); which Gram products somebody will read
    double *saveL;
    bool pinned, gLL, gUU, gUL;
};

// operands and destinations of one block: mirrors k_bcr_factor (ssba_bcr.hip).  false: this block has nothing to do.
// ride: the border columns (d.nb > 0: free shared blocks, closure border) go through the factorisation as two more
// column tiles of the right-hand sides -- yB = G^-1 B, what k_bcrm_fwd (ssba_border.hip) did in a launch of its own
static __device__ __forceinline__ bool factor_ops(const Dev &d, int lev, int top, int which, int bx, bool copier, bool ride, FactorOps &o) {
    o.trL = o.trU = false;
    o.saveU = nullptr;
    o.Bg = nullptr; o.oYB = nullptr; o.xsol = nullptr;
    o.GA = o.GB = o.ga = o.gb = nullptr; o.sgnL = o.sgnU = 1.0; o.nD = o.nr = nullptr;
    o.saveL = nullptr; o.pinned = false; o.gLL = o.gUU = o.gUL = false;
    if (which >= 2) {
        const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
        const BcrLevel &B = which == 3 ? d.slev[0] : d.lev[d.pcr.level];
        const int blk = bx, s = 1 << lev, last = B.n - 1;
        if (P.pin0 && lev == 0 && blk == 1 && copier) {
            const double2 *s2 = reinterpret_cast<const double2 *>(B.L + (size_t)BD * BD);
            double2 *d2 = reinterpret_cast<double2 *>(P.Lbuf + (size_t)BD * BD);
            for (int e = threadIdx.x; e < BD * BD / 2; e += MF_THREADS) d2[e] = s2[e];
        }
        if ((P.pin0 && blk == 0) || (P.pin1 && blk == last)) return false;
        o.hasL = blk - s >= 0 || (P.pin0 && blk > 0);
        o.hasU = blk + s <= last || (P.pin1 && blk < last);
        o.Dg = B.D + (size_t)blk * BD * BD;
        o.rin = B.r + (size_t)blk * BD;
        if (lev == 0) {
            o.Lg = B.L + (size_t)blk * BD * BD;
            o.Ug = B.L + (size_t)(o.hasU ? blk + 1 : blk) * BD * BD;
            o.trL = o.trU = (blk & 1) == 0;
        } else {
            o.Lg = P.Lbuf + (size_t)blk * BD * BD;
            o.Ug = (P.pin1 && blk + s > last) ? P.Ubuf + (size_t)blk * BD * BD : P.LbufT + (size_t)(o.hasU ? blk + s : blk) * BD * BD;
        }
        if (P.pin1 && blk + s == last) o.saveU = P.Ubuf + (size_t)blk * BD * BD;
        const size_t so = P.keep ? (size_t)lev * B.n + blk : (size_t)blk;
        o.oD = top ? B.D + (size_t)blk * BD * BD : (P.keep ? P.Gs + so * BD * BD : nullptr);
        o.oYL = o.hasL ? P.YL + so * BD * BD : nullptr;
        o.oYU = o.hasU ? P.YU + so * BD * BD : nullptr;
        o.orr = top ? B.r + (size_t)blk * BD : P.yr + (size_t)blk * BD;
        if (ride && which == 2) { o.Bg = P.Bb + (size_t)blk * BD * NBP; o.oYB = P.yB + (size_t)blk * BD * NBP; }
        if (top && which == 2) {        // used by the decoupled launch with `solve`
            o.xsol = d.x0 + (size_t)d.chain0 * BD + (size_t)B.pos[blk] * BD;
            o.f0 = (d.chain0 + B.pos[blk]) * SBP;
        }
    } else {
        const BcrLevel &L = d.lev[lev];
        const int blk = top ? 0 : 2 * bx + 1;
        o.hasL = !top;
        o.hasU = !top && (blk + 1 < L.n);
        o.Dg = L.D + (size_t)blk * BD * BD;
        o.Lg = L.L + (size_t)blk * BD * BD;
        o.Ug = L.L + (size_t)(o.hasU ? blk + 1 : blk) * BD * BD;
        o.rin = L.r + (size_t)blk * BD;
        o.oD = L.D + (size_t)blk * BD * BD;
        o.oYL = o.hasL ? L.L + (size_t)blk * BD * BD : nullptr;
        o.oYU = top ? nullptr : L.YU + (size_t)bx * BD * BD;
        o.orr = L.r + (size_t)blk * BD;
        if (ride) { o.Bg = L.B + (size_t)blk * BD * NBP; o.oYB = L.B + (size_t)blk * BD * NBP; }
    }
    return true;
}

// Operands of block bx in step `lev` (stride 1 << lev) of the fused plan, or of its decoupled last step (top).  Step 0 reads
// the level's own D / L / r; step q >= 1 assembles them from what step q - 1 left (see PcrFused).
// which = 2: the plan of the chain (d.pcr), 3: the separator system of a partitioned solve (d.spcr, solution into d.xsep).
// Chains with pinned ends (P.pin0 / P.pin1, see PcrPlan): a pinned block is never factored -- it folds its neighbours like
// every block (it is assembled) -- and never folded: a block whose neighbour at the current stride is pinned, or beyond the
// chain, KEEPS its coupling to the pinned block from then on (PcrFused::Lkeep / Ukeep).
static __device__ __forceinline__ void fused_ops(const Dev &d, int which, int lev, int top, int bx, FactorOps &o) {
    const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
    const PcrFused &F = which == 3 ? d.spcrf : d.pcrf;
    const BcrLevel &B = which == 3 ? d.slev[0] : d.lev[P.level];
    const size_t blk = (size_t)BD * BD;
    const int s = 1 << lev, h = s >> 1, last = B.n - 1, e = bx;
    const int lo = P.pin0 ? 1 : 0, hi = P.pin1 ? last - 1 : last;          // the blocks that are factored (and folded by others)
    o.saveU = o.saveL = nullptr; o.Bg = nullptr; o.oYB = nullptr; o.xsol = nullptr;
    o.oYL = o.oYU = nullptr;
    o.GA = o.GB = o.ga = o.gb = nullptr;
    o.nD = o.nr = nullptr;
    o.pinned = e < lo || e > hi;
    const bool pinned_chain = P.pin0 || P.pin1;
    // couplings of this block at this stride: to e -/+ s, or the kept one to the pinned end; the decoupled last step of a
    // pinned chain still carries the kept ones
    // (in the decoupled last step s >= n: only kept couplings are left)
    o.hasL = e >= 1 && (e - s >= 0 || P.pin0);
    o.hasU = e <= last - 1 && (e + s <= last || P.pin1);
    if (o.pinned) {     // no right-hand sides, except that the pinned last block tracks its coupling to the pinned first one
        o.hasU = false;
        o.hasL = e == last && e >= 1 && P.pin0 && P.pin1;
    }
    o.trL = o.trU = false;
    o.sgnL = o.sgnU = 1.0;
    o.Lg = B.L + e * blk; o.Ug = B.L + e * blk;         // (never read without hasL / hasU)
    if (lev == 0) {
        o.Dg = B.D + e * blk; o.rin = B.r + (size_t)e * BD;
        o.Lg = B.L + e * blk; o.Ug = B.L + (size_t)(e < last ? e + 1 : e) * blk;
        o.trL = o.trU = (e & 1) == 0;       // even coupling blocks of a level are stored transposed (ssba_bcr.hip)
    } else {
        const int set = (lev - 1) & 1, prev = e - h, next = e + h;
        o.Dg = lev == 1 ? B.D + e * blk : F.Dpp[(lev - 1) & 1] + e * blk;
        o.rin = lev == 1 ? B.r + (size_t)e * BD : F.rpp[(lev - 1) & 1] + (size_t)e * BD;
        if (prev >= lo) { o.GA = F.GUU[set] + prev * blk; o.ga = F.gU[set] + (size_t)prev * BD; }
        if (next <= hi) { o.GB = F.GLL[set] + next * blk; o.gb = F.gL[set] + (size_t)next * BD; }
        // the coupling was renewed by the last step iff the neighbour at half the stride was folded; else it is a kept one
        if (o.hasL) { if (prev >= lo) { o.Lg = F.GUL[set] + prev * blk; o.sgnL = -1.0; } else o.Lg = F.Lkeep + e * blk; }
        if (o.hasU) { if (next <= hi) { o.Ug = F.GULT[set] + next * blk; o.sgnU = -1.0; } else o.Ug = F.Ukeep + e * blk; }
        if (!top) { o.nD = F.Dpp[lev & 1] + e * blk; o.nr = F.rpp[lev & 1] + (size_t)e * BD; }
    }
    if (pinned_chain && !top) {
        // save a coupling when the NEXT step will find it kept (its half-stride neighbour, e -/+ s, pinned or outside) and this
        // step did not read it from the keep buffer already
        if (o.hasL && e - s < lo && o.Lg != F.Lkeep + e * blk) o.saveL = F.Lkeep + e * blk;
        if (o.hasU && e + s > hi && o.Ug != F.Ukeep + e * blk) o.saveU = F.Ukeep + e * blk;
    }
    // the Gram products somebody reads: the next step's block e - s / e + s assembles from them (a kept coupling feeds nobody's D)
    o.gLL = o.hasL && e - s >= 0 && !top;
    o.gUU = o.hasU && e + s <= last && !top;
    o.gUL = o.hasL && o.hasU && !top;
    o.oD = top ? B.D + e * blk : nullptr;
    o.orr = top ? B.r + (size_t)e * BD : nullptr;
    if (top) {
        o.xsol = which == 3 ? d.xsep + (size_t)B.pos[e] * BD : d.x0 + (size_t)d.chain0 * BD + (size_t)B.pos[e] * BD;
        o.f0 = (d.chain0 + B.pos[e]) * SBP;
        if (pinned_chain) {         // the factor outputs k_bcr_backsub reads after the separator solve; the pinned rows in place
            o.xsol = nullptr;
            o.oYL = o.hasL ? P.YL + e * blk : nullptr;
            o.oYU = o.hasU ? P.YU + e * blk : nullptr;
            if (o.pinned) { o.nD = B.D + e * blk; o.nr = B.r + (size_t)e * BD; }
            // S[last, first], for k_sep_pack: the pinned last block's coupling to the pinned first one
            if (o.pinned && e == last && P.pin0 && P.pin1) o.saveL = P.Lbuf + e * blk;
        }
    }
    const int oset = lev & 1;
    o.oGLL = F.GLL[oset] + e * blk; o.oGUU = F.GUU[oset] + e * blk;
    o.oGUL = F.GUL[oset] + e * blk; o.oGULT = F.GULT[oset] + e * blk;
    o.ogL = F.gL[oset] + (size_t)e * BD; o.ogU = F.gU[oset] + (size_t)e * BD;
}

struct FactorLds {
    // every block row has its own slots: nothing is overwritten during a factorisation, so the hand-offs below need
    // no write-after-read protection
    double P[NDT][4][64], Q[NDT][4][64];    // sub-step operands of the diagonal tiles: [block row][sub-step][lane]
    double A[NDT - 1][4][4][64];            // finished D panels as A operands: [block row][column tile - 1][register][lane]
    double rc[80], rs[80];                  // 1 / pivot, 1 / sqrt(pivot) by row
    // hand-offs between the four waves (LDS words, monotonic): no workgroup barrier inside the factorisation
    int seqPQ;                              // sub-steps published so far: 4 k + r + 1
    int seqA[4];                            // per column tile - 1: block rows whose panel is published (k + 1)
    int seqY[8];                            // fused steps: per block row, the waves whose rows of [YL | YU | yr] are in LDS (5 = all)
    int bad;
};

// Hand-off words live in LDS and are read / written with workgroup-scope relaxed atomics (plain ds_read / ds_write):
// the LDS unit serves the requests of a wave in issue order, so data written before the flag is visible to whoever
// sees the flag; the compiler is kept from reordering around the flag access by an empty asm with a memory clobber.
// The spin is wave-uniform (v_readfirstlane) and BOUNDED: a lost hand-off must never hang the GPU -- it flags the
// step as failed instead (S.bad = 2).
#define MF_WAIT_GE(word, v)                                                                                              \
    do {                                                                                                                 \
        int it_ = 0;                                                                                                     \
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < (v)) { \
            if (++it_ > (1 << 20)) { S.bad = 2; break; }                                                                 \
            __builtin_amdgcn_s_sleep(1);                                                                                 \
        }                                                                                                                \
        asm volatile("" ::: "memory");                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)
// the store must not sink below the arithmetic that follows it (the instruction scheduler would happily post a
// sub-step's flag a whole sub-step late): full scheduling barriers on both sides
#define MF_POST(word, v)                                                                                                 \
    do {                                                                                                                 \
        asm volatile("" ::: "memory");                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __hip_atomic_store(&(word), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    } while (0)

// keeps the operand reads issued above it ahead of the matrix instructions below it (memory clobber for the compiler's
// middle end, a scheduling barrier for the machine scheduler, which would otherwise sink the reads again)
#define MF_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// NRW: column tiles of [L | U^T] per wave (1: three workgroups per block, 2: two, 3: one, 0: none -- blocks without couplings)
// rlo .. nrt: the column tiles of the right-hand sides [L | U^T | B] this launch carries (0 .. 9 without border columns,
// 0 .. 11 with them, 9 .. 11 for decoupled blocks with border columns)
// solve (decoupled blocks without border columns, NRW = 0): the workgroup also solves  G^T x = yr  -- the block's part of
// the solution -- from an LDS copy of G, which is what k_bcr_backsub did in a launch of its own.
// MODE 0: operands and destinations of the classic plans (factor_ops).
// MODE 1: the decoupled last step of the fused plan: D and r are assembled from the previous step's Gram products on load.
// MODE 2: a step of the fused plan (PcrFused): 8 waves -- waves 0..3 run the D stream as always, waves 4..7 carry ALL nine
//         column tiles of [L | U^T] (every workgroup of a block repeats the whole factorisation: the products that follow
//         need every column, and a cross-workgroup exchange would cost more than the repetition) -- then [YL | yr] and
//         [YU | yr] go to LDS block row by block row AS THEY BECOME FINAL, and every wave but the first adds the contribution of
//         a finished block row to its share of the 55 Gram tiles (16 rows = four instructions per tile) in the time it would
//         otherwise spend waiting for the pivot chain: when the last block row is done, so are the products.
static __device__ __forceinline__ void gram_tile(int u, int &kind, int &ti, int &tj);
static __device__ __forceinline__ void gram_store(const FactorOps &o, int kind, int ti, int tj, const double (&v)[4], double *scr, int lane);
constexpr int MF_THREADS2 = 512;
// ROLE 0: every wave runs both streams (MODE 0, 1).  MODE 2 instantiates the body twice -- ROLE 1: the D stream only (waves
// 0..3), ROLE 2: right-hand-side tiles only (waves 4..7) -- so that neither role carries the other's register arrays
// (one body with run-time roles needed 538 spilled registers at the 256 a wave of a 512-lane workgroup may use).
template <int NRW, int MODE, int ROLE>
static __device__ __forceinline__ void factor_body(const Dev &d, FactorLds &S, double *mf_solve_lds, int lev, int top, int which, int nblocks, int ns, int rlo, int nrt, int solve) {
    constexpr bool HAS_D = ROLE != 2, HAS_R = ROLE != 1;
    State &st = *d.st;
    const StateFlags sf = state_flags_vmem(d.st);       // tested once the operand reads are in flight (ssba_device.h)
    FactorOps o;
    int bx, by;
    xcd_map((int)blockIdx.x, nblocks, ns, bx, by);
    if (MODE) fused_ops(d, which, lev, top, bx, o);
    else if (!factor_ops(d, lev, top, which, bx, by == 0, nrt > NRT, o)) return;
    const int t = threadIdx.x, lane = t & 63, g = lane >> 4, j = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr bool rwave = HAS_R;       // this wave carries right-hand-side tiles
    const bool storeG = by == 0;
    // this workgroup's share of the nine right-hand-side column tiles, dealt to waves 1, 2, 3, 0, 1, ... (wave 0 owns the
    // longest column of [D | r] and two diagonal tiles: it comes last)
    const int rfirst = MODE == 2 ? 0 : rlo + (by * (nrt - rlo)) / ns;
    const int rcnt = MODE == 2 ? NRT : rlo + ((by + 1) * (nrt - rlo)) / ns - rfirst;
    constexpr int NRA = NRW > 0 ? NRW : 1;        // array extents (NRW = 0: the decoupled last step has no right-hand-side tiles)
    int rcol[NRA];
    bool ract[NRA];
#pragma unroll
    for (int q = 0; q < NRW; ++q) {
        const int idx = (MODE == 2 ? (w & 3) : ((w + 3) & 3)) + 4 * q;
        rcol[q] = rfirst + idx;
        ract[q] = rwave && idx < rcnt && ((rcol[q] <= 4 && o.hasL) || (rcol[q] >= 4 && rcol[q] < NRT && o.hasU) || (rcol[q] >= NRT && o.Bg));
    }
    const int dj = !HAS_D ? -1 : w == 0 ? 4 : w;          // column tile of [D | r] this wave owns (wave 0 also owns tile (0, 0)); -1: none
    if (t == 0) { S.bad = 0; S.seqPQ = 0; }
    if (t < 4) S.seqA[t] = 0;
    if (t < 8) { S.rc[72 + t] = 1.0; S.seqY[t] = 0; }
    // mf_solve_lds: solve: G (BD x BD) | yr (BD) | x;  MODE 2: [YL | yr] and [YU | yr], 72 x 80 each
    if (MODE == 2) {        // columns 72..79 of both staged operands: yr goes into column 72 later, the rest stays zero
        for (int i = t; i < 2 * BD * 8; i += MF_THREADS2) mf_solve_lds[(i >> 3) * 80 + BD + (i & 7)] = 0.0;
    }
    // the only workgroup barrier before the end of the factorisation: the hand-off words are initialised.  It sits BEFORE
    // the loads so that wave 0 can start on the first diagonal tile as soon as ITS data is there, without waiting for the
    // other waves' right-hand-side tiles.
    __syncthreads();
    MF_STAMP(0);

    // ---- load: straight into the accumulator layout (128-byte row segments): one per-lane base pointer per column
    //      tile, compile-time row offsets, every load issued before the first is waited for ----------------------
    mf_d4 dt[NDT], d00, rt[NRA][NDT];
    {
        const int colD = 16 * max(dj, 0) + j;
        const int offD = g * BD + min(colD, BD - 1);
        double rv[MODE ? 1 : NDT][4];
        if (HAS_D && !MODE) {
            const double *pD = o.Dg + offD, *pr = o.rin + g;
#pragma unroll
            for (int q = 0; q < 4; ++q) d00[q] = (w == 0) ? o.Dg[(4 * q + g) * BD + j] : 0.0;      // first: the first diagonal tile waits for nothing else
#pragma unroll
            for (int k = 0; k < NDT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = k < 4 || q < 2;           // rows 72..79 are padding
                    dt[k][q] = (in && k <= dj) ? pD[(16 * k + 4 * q) * BD] : 0.0;
                    rv[k][q] = (in && w == 0) ? pr[16 * k + 4 * q] : 0.0;
                }
        }
        if (HAS_D && MODE) {
            // fused plan: [D | r] = [Dg - GA - GB | rin - ga - gb], the three reads of an entry issued together.  A lane of
            // column 72 (wave 0's tile column 4) reads the right-hand-side vectors instead of the blocks -- same loop, its own
            // base pointers and row stride -- and the lanes beyond it read nothing: no separate arrays for r.
            const bool isD = colD < BD, isR = colD == BD;
            const long rstride = isD ? BD : 1, off0 = isD ? offD : g;
            const double *b0 = isD ? o.Dg : o.rin, *b1 = isD ? o.GA : o.ga, *b2 = isD ? o.GB : o.gb;
            const bool act = isD || isR;
            mf_d4 x1[NDT], x2[NDT], y1, y2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d00[q] = (w == 0) ? o.Dg[(4 * q + g) * BD + j] : 0.0;
                y1[q] = (w == 0 && o.GA) ? o.GA[(4 * q + g) * BD + j] : 0.0;
                y2[q] = (w == 0 && o.GB) ? o.GB[(4 * q + g) * BD + j] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NDT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = (k < 4 || q < 2) && k <= dj && act;
                    const long off = off0 + (16 * k + 4 * q) * rstride;
                    dt[k][q] = in ? b0[off] : 0.0;
                    x1[k][q] = (in && b1) ? b1[off] : 0.0;
                    x2[k][q] = (in && b2) ? b2[off] : 0.0;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) d00[q] = d00[q] - y1[q] - y2[q];
#pragma unroll
            for (int k = 0; k < NDT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) dt[k][q] = dt[k][q] - x1[k][q] - x2[k][q];
        }
        if (!HAS_D) {
            d00 = mf_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < NDT; ++k) dt[k] = mf_d4{0.0, 0.0, 0.0, 0.0};
        }
        bool rok[NRA];
#pragma unroll
        for (int q = 0; q < NRW; ++q) {
            const int col = 16 * rcol[q] + j;
            const bool isL = col < BD, isB = rcol[q] >= NRT;
            const int cc = isL ? col : col - BD;
            const bool tr = isL ? o.trL : o.trU;
            const double *base = isL ? o.Lg : o.Ug;
            rok[q] = ract[q] && (isB || (isL ? o.hasL : o.hasU));
            if (!ract[q]) {
#pragma unroll
                for (int k = 0; k < NDT; ++k) rt[q][k] = mf_d4{0.0, 0.0, 0.0, 0.0};
            } else if (isB) {                            // border columns: BD x NBP, row-major
                const double *pp = o.Bg + g * NBP + (col - 2 * BD);
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) rt[q][k][qq] = (k < 4 || qq < 2) ? pp[(16 * k + 4 * qq) * NBP] : 0.0;
            } else if (tr) {
                const double *pp = base + cc * BD + g;
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) rt[q][k][qq] = (k < 4 || qq < 2) ? pp[16 * k + 4 * qq] : 0.0;
            } else {
                const double *pp = base + g * BD + cc;
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) rt[q][k][qq] = (k < 4 || qq < 2) ? pp[(16 * k + 4 * qq) * BD] : 0.0;
            }
        }
        if (sf.dead()) return;
        // masks: column 72 of [D | r] is the right-hand side, the columns after it are zero; absent couplings are zero
        if (!MODE && dj == 4) {
#pragma unroll
            for (int k = 0; k < NDT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) dt[k][q] = colD < BD ? dt[k][q] : colD == BD ? rv[k][q] : 0.0;
        }
        if (HAS_D && MODE && o.nD && by == 0) {      // the assembled block (upper tiles) and right-hand side, for the next step
#pragma unroll
            for (int k = 0; k < NDT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!((k < 4 || q < 2) && k <= dj)) continue;
                    if (colD < BD) {
                        o.nD[offD + (16 * k + 4 * q) * BD] = dt[k][q];
                        // (a pinned block of a partitioned chain in its final form: k_sep_pack copies the whole block)
                        if (MODE == 1 && o.pinned && k < dj) o.nD[(size_t)colD * BD + 16 * k + 4 * q + g] = dt[k][q];
                    } else if (colD == BD) o.nr[g + 16 * k + 4 * q] = dt[k][q];
                }
            if (w == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o.nD[(4 * q + g) * BD + j] = d00[q];
            }
        }
        if (HAS_R && MODE) {        // couplings read from the Gram products carry a minus sign, kept ones and the level's own do not
#pragma unroll
            for (int q = 0; q < NRW; ++q) {
                const double sg = (16 * rcol[q] + j < BD) ? o.sgnL : o.sgnU;
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) rt[q][k][qq] *= sg;
            }
        }
#pragma unroll
        for (int q = 0; q < NRW; ++q) {
            if (!ract[q]) continue;
            if (!rok[q]) {
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) rt[q][k][qq] = 0.0;
            }
            const int col = 16 * rcol[q] + j;
            const bool saver = MODE != 2 || by == 0;       // (the workgroups of a fused step all hold every tile)
            if (o.saveU && rok[q] && col >= BD && rcol[q] < NRT && saver) {
                double *ps = o.saveU + g * BD + col - BD;
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        if (k < 4 || qq < 2) ps[(16 * k + 4 * qq) * BD] = rt[q][k][qq];
            }
            if (MODE && o.saveL && rok[q] && col < BD && saver) {
                double *ps = o.saveL + g * BD + col;
#pragma unroll
                for (int k = 0; k < NDT; ++k)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        if (k < 4 || qq < 2) ps[(16 * k + 4 * qq) * BD] = rt[q][k][qq];
            }
        }
        if (MODE && o.pinned) return;       // a pinned end of a partitioned chain: assembled (and its coupling tracked), never factored
    }

    // lane masks of the sub-step operands (all-ones / zero words): bit selects instead of branches
    auto mk = [](bool c) { return c ? 0xFFFFFFFFu : 0u; };
    const uint32_t mG0 = mk(g == 0), mG1 = mk(g == 1), mG2 = mk(g == 2), mG3 = mk(g == 3);
    const uint32_t mDiag = mk(j < 4 && g == j);
    const uint32_t m10 = mk(j == 1 && g == 0), m20 = mk(j == 2 && g == 0), m21 = mk(j == 2 && g == 1);
    const uint32_t m30 = mk(j == 3 && g == 0), m31 = mk(j == 3 && g == 1), m32 = mk(j == 3 && g == 2);
    // the same masks as 0 / 1 factors: a blend of wave-uniform values into the per-lane operand is a short chain of FMAs
    // (8 issue cycles each) instead of two v_cndmask + one v_or per term
    const double kDiag = mDiag ? 1.0 : 0.0, k10 = m10 ? 1.0 : 0.0, k20 = m20 ? 1.0 : 0.0, k21 = m21 ? 1.0 : 0.0;
    const double k30 = m30 ? 1.0 : 0.0, k31 = m31 ? 1.0 : 0.0, k32 = m32 ? 1.0 : 0.0;
    const double kG0 = mG0 ? 1.0 : 0.0, kG1 = mG1 ? 1.0 : 0.0, kG2 = mG2 ? 1.0 : 0.0, kG3 = mG3 ? 1.0 : 0.0;
    auto sel = [](double v, uint32_t m, double acc) {      // (v & m) | acc, bitwise
        const uint32_t lo = ((uint32_t)__double2loint(v) & m) | (uint32_t)__double2loint(acc);
        const uint32_t hi = ((uint32_t)__double2hiint(v) & m) | (uint32_t)__double2hiint(acc);
        return __hiloint2double((int)hi, (int)lo);
    };
    // ---- sub-steps of a diagonal tile T (block row k; nsub = 4, or 2 for the half tile of rows 64..71) -----------
    // A non-positive pivot is not handled here (that would sit on the chain): it ends up in S.piv and is found after
    // the factorisation; nothing below loops on data, so garbage just flows through.
    auto factor_tile = [&](mf_d4 &T, int k, int nsub) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nsub) break;
            const int b = 4 * r;      // S[a][c] = T[4r + a][4r + c] sits in register r of lane 16 a + 4r + c
            const double tr = T[r];
            const double s00 = mf_readlane(tr, b), s10 = mf_readlane(tr, 16 + b), s20 = mf_readlane(tr, 32 + b), s30 = mf_readlane(tr, 48 + b);
            const double s11 = mf_readlane(tr, 16 + b + 1);
            double s21 = mf_readlane(tr, 32 + b + 1), s31 = mf_readlane(tr, 48 + b + 1);
            const double s22 = mf_readlane(tr, 32 + b + 2);
            double s32 = mf_readlane(tr, 48 + b + 2);
            const double s33 = mf_readlane(tr, 48 + b + 3);
            // LDL^T of the pivot block: unit lower l, pivots d.  The serial chain is, per pivot: hardware reciprocal
            // estimate x0 -> { t = s x0 , e = 1 - d x0 } -> l = t + t e  (= s / d to ~2^-52) -> next pivot; the refined
            // reciprocal itself (x0 + x0 e, for the row scaling) is off the chain.
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
            // M = l^-1 (unit lower); P: lane (g, i = j) holds M[i][g] for i < 4
            const double n20 = fma(l21, l10, -l20);
            const double n31 = fma(l32, l21, -l31);
            const double n30 = -fma(l32, n20, fma(l31, -l10, l30));
            // P = sum of entry x lane mask; the terms known early first, the last pivot's (n31, l32, n30) at the end
            double Pv = fma(-l10, k10, kDiag);
            Pv = fma(n20, k20, Pv);
            Pv = fma(-l21, k21, Pv);
            Pv = fma(n31, k31, Pv);
            Pv = fma(-l32, k32, Pv);
            Pv = fma(n30, k30, Pv);
            const mf_d4 y4 = mf(Pv, tr, mf_d4{0.0, 0.0, 0.0, 0.0});
            const double rcg = fma(rc3, kG3, fma(rc2, kG2, fma(rc1, kG1, rc0 * kG0)));
            const double qs = (j > b + 3) ? -rcg : 0.0;
            const double y = y4[0];             // lane (g, j): unnormalised pivot row c_g[j] = (d l^T ...)[4r + g][j]
            T[r] = y;
            const double Qv = y * qs;
            if (r + 1 < nsub) T = mf(Qv, y, T);
            S.P[k][r][lane] = Pv;
            S.Q[k][r][lane] = Qv;
            if (j == 0) S.rc[16 * k + b + g] = rcg;         // four lanes, one reciprocal pivot each
            MF_POST(S.seqPQ, 4 * k + r + 1);
        }
    };
    // Two instruction streams per wave.  The D stream (this wave's column of [D | r]) is what the next factorisation
    // waits for, so it runs first in every block row: forward substitution of tile (k, dj) sub-step by sub-step as the
    // diagonal tile's operands arrive; a finished register of it is at once the A operand (-x / d(row)) and the B
    // operand of the update of the wave's OWN diagonal tile (dj, dj), and goes out to LDS as the A operand of
    // everybody's updates of block row dj.  The R stream (the wave's tiles of [L | U^T]) follows one block row behind
    // and fills the time the wave would otherwise spend waiting for the next diagonal tile.
    auto d_panel = [&](int k) {
        auto sub = [&](int r, double Pv, double Qv, double rcr) {
            const mf_d4 yd = mf(Pv, dt[k][r], mf_d4{0.0, 0.0, 0.0, 0.0});
            const double y = yd[0];
            dt[k][r] = y;
            const double a = -y * rcr;
            S.A[k][max(dj, 1) - 1][r][lane] = a;       // (dj >= 1 wherever this runs)
            if (r < 3) dt[k] = mf(Qv, y, dt[k]);      // the next sub-step waits for this one: issued before the update of (dj, dj)
#pragma unroll
            for (int i = 1; i < NDT; ++i)
                if (i == dj) dt[i] = mf(a, y, dt[i]);
        };
        // a wave that arrives late finds the whole diagonal tile published: one look at the hand-off word, all twelve
        // operand reads in one batch (otherwise every sub-step pays a flag read and an operand read round trip)
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&S.seqPQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 4 * k + 4) {
            asm volatile("" ::: "memory");
            double Pa[4], Qa[4], ra[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { Pa[r] = S.P[k][r][lane]; Qa[r] = S.Q[k][r][lane]; ra[r] = S.rc[16 * k + 4 * r + g]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) sub(r, Pa[r], Qa[r], ra[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // flag and operands in ONE batch of LDS reads (the LDS serves a wave's reads in order: operands read
                // after a flag value that says "published" are the published ones); retried until the flag is there
                double Pv, Qv, rcr;
                int it_ = 0;
                for (;;) {
                    const int f = __hip_atomic_load(&S.seqPQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                    Pv = S.P[k][r][lane]; Qv = S.Q[k][r][lane]; rcr = S.rc[16 * k + 4 * r + g];
                    asm volatile("" ::: "memory");
                    if (__builtin_amdgcn_readfirstlane(f) >= 4 * k + r + 1) break;
                    if (++it_ > (1 << 20)) { S.bad = 2; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_sched_barrier(0);
                sub(r, Pv, Qv, rcr);
            }
        }
        MF_POST(S.seqA[dj - 1], k + 1);
    };
    // tiles (i, dj), k < i < dj, of the D column: updated with the panels of the other columns
    auto d_update = [&](int k) {
#pragma unroll
        for (int i = 1; i < NDT - 1; ++i) {
            if (i <= k || i >= dj) continue;
            MF_WAIT_GE(S.seqA[i - 1], k + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) dt[i] = mf(S.A[k][i - 1][s][lane], dt[k][s], dt[i]);
        }
    };
    // ---- fused steps: [YL | yr] and [YU | yr] (72 x 80 each, rows scaled by 1 / sqrt(pivot)) are published block row by block
    //      row; S.seqY[k] counts the waves that have written their part of block row k (the four right-hand-side waves and
    //      wave 0 for yr).  Every wave but wave 0 owns up to NG of the 55 Gram tiles and adds the contribution of the block
    //      rows in order as they complete: without blocking wherever it still has substitution work, blocking at the end. ----
    double *sAL = mf_solve_lds, *sAU = mf_solve_lds + BD * 80;
    // Who takes which tile.  Workgroup `by` of the ns that share the block owns the tiles u = by, by + ns, ... (18 or 19 of
    // the 55 at ns = 3); its m-th tile goes to D wave 1 (m < 6), 2 (m < 11) or 3 (m < 15) -- they are done with the pivot
    // chain early and add block rows while the right-hand-side waves, which the matrix pipe of the CU is busy with, still
    // work on them -- and the few beyond that to the right-hand-side waves, which form them after their last block row.
    constexpr int NG = MODE == 2 ? (ROLE == 1 ? 6 : 4) : 1;
    int gti[NG], gtj[NG], gkind[NG];
    bool ghave[NG];
    mf_d4 gacc[NG];
    int grow = 0;           // next block row to add
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            int m;
            if (ROLE == 1) m = w == 1 ? i : w == 2 ? 6 + i : w == 3 ? 11 + i : 1000;
            else m = 15 + ((w + 3) & 3) + 4 * i;               // waves 5, 6, 7, 4 (wave 4 carries three column tiles: it comes last)
            const bool mine = ROLE == 1 ? (w == 1 ? i < 6 : w == 2 ? i < 5 : w == 3 ? i < 4 : false) : true;
            const int u = by + m * ns;
            gram_tile(min(max(u, 0), 54), gkind[i], gti[i], gtj[i]);
            ghave[i] = mine && u < 55 && (gkind[i] == 0 ? o.gLL : gkind[i] == 1 ? o.gUU : o.gUL);
            gacc[i] = mf_d4{0.0, 0.0, 0.0, 0.0};
        }
    }
    auto y_post = [&](int k) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) __hip_atomic_fetch_add(&S.seqY[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto y_rows = [&](int k) {          // right-hand-side waves: block row k of the wave's column tiles
        double rsk[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) rsk[qq] = (k < 4 || qq < 2) ? sqrt(S.rc[16 * k + 4 * qq + g]) : 0.0;
#pragma unroll
        for (int q = 0; q < NRW; ++q) {
            if (!ract[q]) continue;
            const int col = 16 * rcol[q] + j;
            double *pp = col < BD ? sAL + g * 80 + col : sAU + g * 80 + (col - BD);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                if (k < 4 || qq < 2) pp[(16 * k + 4 * qq) * 80] = rt[q][k][qq] * rsk[qq];
        }
        y_post(k);
    };
    auto yr_rows = [&](int k) {         // wave 0: block row k of yr = column 72 of [D | r] (lanes j == 8 of its column tile 4)
        if (j == 8) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                if (k < 4 || qq < 2) {
                    const int row = 16 * k + 4 * qq + g;
                    const double v = dt[k][qq] * sqrt(S.rc[row]);
                    sAL[row * 80 + BD] = v;
                    sAU[row * 80 + BD] = v;
                }
        }
        y_post(k);
    };
    auto gram_rows = [&](int k) {       // k at run time: the contribution of block row k to this wave's tiles
        const int off = (16 * k + g) * 80 + j;
        double a[NG][4], b[NG][4];
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (!ghave[i]) continue;
            const double *sa = (gkind[i] == 0 ? sAL : sAU) + off + 16 * gti[i], *sb = (gkind[i] == 1 ? sAU : sAL) + off + 16 * gtj[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {       // (the last block row has eight rows)
                a[i][q] = (q < 2 || k < 4) ? sa[4 * q * 80] : 0.0;
                b[i][q] = (q < 2 || k < 4) ? sb[4 * q * 80] : 0.0;
            }
        }
        MF_FENCE();
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (!ghave[i]) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < 2 || k < 4) gacc[i] = mf(a[i][q], b[i][q], gacc[i]);
        }
    };
    auto r_step = [&](int k) {
        const int nsub = k < 4 ? 4 : 2;
        MF_WAIT_GE(S.seqPQ, 4 * k + nsub);          // this stream lags: the whole diagonal tile is normally long done
        double Pa[4], Qa[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { Pa[r] = r < nsub ? S.P[k][r][lane] : 0.0; Qa[r] = r < nsub ? S.Q[k][r][lane] : 0.0; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nsub) break;
            const double Pv = Pa[r], Qv = Qa[r];
            mf_d4 yq[NRA];
#pragma unroll
            for (int q = 0; q < NRW; ++q)
                if (ract[q]) yq[q] = mf(Pv, rt[q][k][r], mf_d4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
            for (int q = 0; q < NRW; ++q)
                if (ract[q]) {
                    rt[q][k][r] = yq[q][0];
                    if (r + 1 < nsub) rt[q][k] = mf(Qv, yq[q][0], rt[q][k]);
                }
        }
        if (MODE == 2) y_rows(k);       // block row k of this wave's columns of [YL | YU] is final: into LDS for the Gram products
        // the panels of block row k (this stream lags, they are normally all there): one pass over the flags, then all
        // A operands in one batch of LDS reads
#pragma unroll
        for (int i = 1; i < NDT; ++i)
            if (i > k) MF_WAIT_GE(S.seqA[i - 1], k + 1);
        double a[NDT][4];
#pragma unroll
        for (int i = 1; i < NDT; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) a[i][s] = i > k ? S.A[k][i - 1][s][lane] : 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 1; i < NDT; ++i) {
                if (i <= k) continue;
#pragma unroll
                for (int q = 0; q < NRW; ++q)
                    if (ract[q]) rt[q][i] = mf(a[i][s], rt[q][k][s], rt[q][i]);
            }
    };

    MF_STAMP(1);
    MF_WG_STAMP(1);
    if (HAS_D && w == 0) factor_tile(d00, 0, 4);
#pragma unroll
    for (int k = 0; k < NDT; ++k) {
        MF_STAMP(2 + 4 * k);
        if (HAS_D && dj > k) d_panel(k);
        if (MODE == 2 && HAS_D && w == 0 && k < NDT - 1) yr_rows(k);        // (wave 0 owns column tile 4: its panel of block row k is final)
        MF_STAMP(3 + 4 * k);
        if (HAS_D && k + 1 < NDT && dj == k + 1) factor_tile(dt[k + 1], k + 1, k + 1 < 4 ? 4 : 2);   // the next diagonal tile is this wave's
        if (MODE == 2 && HAS_D && w == 0 && k + 1 == NDT - 1) yr_rows(NDT - 1);
        MF_STAMP(4 + 4 * k);
        if (HAS_D && dj > k + 1) d_update(k);
        if (k >= 1 && rwave) r_step(k - 1);
        MF_STAMP(5 + 4 * k);
    }
    if (rwave) r_step(NDT - 1);
    MF_STAMP(30);
    MF_WG_STAMP(2);
    if (MODE == 2 && (HAS_R || w > 0)) {       // what is left of the Gram products: block rows in order, as they complete
        for (; grow < NDT; ++grow) {
            MF_WAIT_GE(S.seqY[grow], 5);
            gram_rows(grow);
        }
    }
    __syncthreads();
    if (t < BD) {
        // 1 / pivot: a non-positive or non-finite pivot shows here (negative, infinite or NaN reciprocal): Cholesky breakdown,
        // the step is rejected (Ceres: LM retries with a smaller radius)
        const double rc = S.rc[t];
        if (!(rc > 0.0) || !(rc < INFINITY)) S.bad = 1;
        S.rs[t] = sqrt(rc);
    }
    __syncthreads();
    if (S.bad) {
        if (t == 0) st.step_failed = 1;
        return;
    }
    if (MODE == 2) {
        // ---- the Gram tiles of this wave (the hand-off slots of the factorisation are dead: S.A is the transposition scratch) ----
        MF_STAMP(32);
        MF_WG_STAMP(3);
        double *scr = &S.A[0][0][0][0] + w * (16 * 17);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (!ghave[i]) continue;
            const double v[4] = {gacc[i][0], gacc[i][1], gacc[i][2], gacc[i][3]};
            gram_store(o, gkind[i], gti[i], gtj[i], v, scr, lane);
        }
        MF_STAMP(33);
        MF_WG_STAMP(4);
        return;
    }
    // ---- store: rows scaled by 1/sqrt(d) ---------------------------------------------------------------------
    double rsv[NDT][4];
#pragma unroll
    for (int k = 0; k < NDT; ++k)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) rsv[k][qq] = (k < 4 || qq < 2) ? S.rs[16 * k + 4 * qq + g] : 0.0;
#pragma unroll
    for (int q = 0; q < NRW; ++q) {
        if (!ract[q]) continue;
        const int col = 16 * rcol[q] + j;
        if (rcol[q] >= NRT) {        // yB = G^-1 B
            double *pb = o.oYB + g * NBP + (col - 2 * BD);
#pragma unroll
            for (int k = 0; k < NDT; ++k)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    if (k < 4 || qq < 2) pb[(16 * k + 4 * qq) * NBP] = rt[q][k][qq] * rsv[k][qq];
            continue;
        }
        double *dst = col < BD ? o.oYL : o.oYU;
        if (!dst) continue;
        double *pp = dst + g * BD + (col < BD ? col : col - BD);
#pragma unroll
        for (int k = 0; k < NDT; ++k)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                if (k < 4 || qq < 2) pp[(16 * k + 4 * qq) * BD] = rt[q][k][qq] * rsv[k][qq];
    }
    const bool do_solve = NRW == 0 && solve && o.xsol;
    if (storeG) {
        // G = U^T: column c of U is row c of G; lane (g, j) of tile (k, c) holds U[16 k + 4 q + g][16 c + j]
        auto store_g = [&](const mf_d4 &T, int k, int c, bool diag) {
            const int col = 16 * c + j;
            if (col < BD && o.oD) {
                double *pp = o.oD + col * BD + g;
                double *pl = mf_solve_lds + col * BD + g;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    if (!(k < 4 || qq < 2)) continue;
                    const int row = 16 * k + 4 * qq + g;
                    if (!diag || col >= row) {
                        const double v = (diag && col == row) ? rsv[k][qq] : T[qq] * rsv[k][qq];
                        pp[16 * k + 4 * qq] = v;
                        if (do_solve) pl[16 * k + 4 * qq] = v;
                    }
                }
            }
            if (col == BD) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    if (k < 4 || qq < 2) {
                        const double v = T[qq] * rsv[k][qq];
                        o.orr[16 * k + 4 * qq + g] = v;
                        if (do_solve) mf_solve_lds[BD * BD + 16 * k + 4 * qq + g] = v;
                    }
            }
        };
#pragma unroll
        for (int k = 0; k < NDT; ++k)
            if (k <= dj) store_g(dt[k], k, dj, k == dj);
        if (w == 0) store_g(d00, 0, 0, true);
    }
    if (do_solve) {
        // wave 0: G^T x = yr by a column sweep from the bottom (k_bcr_backsub's); the diagonal of G holds 1 / G_kk
        __syncthreads();
        if (w == 0) {
            const double *sG = mf_solve_lds, *sv = mf_solve_lds + BD * BD;
            auto bc = [](double v, int ln) {
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), ln), __builtin_amdgcn_readlane(__double2loint(v), ln));
            };
            double lo = sv[lane];
            double hi = lane < BD - 64 ? sv[64 + lane] : 0.0;
#pragma unroll
            for (int k = BD - 1; k >= 64; --k) {
                const double xk = bc(hi, k - 64) * sG[k * BD + k];
                const double gl = sG[k * BD + lane];
                const double gh = (lane < k - 64) ? sG[k * BD + 64 + lane] : 0.0;
                lo -= gl * xk;
                hi = (lane == k - 64) ? xk : hi - gh * xk;
            }
#pragma unroll
            for (int k = 63; k >= 0; --k) {
                const double xk = bc(lo, k) * sG[k * BD + k];
                const double gl = (lane < k) ? sG[k * BD + lane] : 0.0;
                lo = (lane == k) ? xk : lo - gl * xk;
            }
            o.xsol[lane] = lo;
            if (lane < BD - 64) o.xsol[64 + lane] = hi;
            if (solve == 2) {
                // ... and updates its twelve poses: candidate = Plus(x, delta_p), |dx|^2 and the non-finite flag as this
                // block's partial sums (k_pose_update's job; k_decide reads one partial per block then); the copy of x to
                // the best iterate is k_backsub_eval_w's (it has to happen after termination too)
                double *sx = mf_solve_lds + BD * BD + BD;
                sx[lane] = lo;
                if (lane < BD - 64) sx[64 + lane] = hi;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double dn = 0.0, nonfinite = 0.0;
                const int f = o.f0 + lane;
                if (lane < SBP && f < d.nfree) {
                    const int k = d.free_pose[f];
                    double T[12], eps[6], Tn[12];
#pragma unroll
                    for (int c = 0; c < 12; ++c) T[c] = d.poses[(size_t)k * 12 + c];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        eps[c] = sx[6 * lane + c] * st.ls_alpha;
                        if (!isfinite(eps[c])) nonfinite = 1.0;
                    }
                    se3_plus(T, eps, Tn);
#pragma unroll
                    for (int c = 0; c < 12; ++c) {
                        d.cand_poses[(size_t)k * 12 + c] = Tn[c];
                        const double df = Tn[c] - T[c];
                        dn += df * df;
                    }
                }
                const double a = wave_sum(dn), b = wave_sum(nonfinite);
                if (lane == 0) {
                    double *pp = d.part_pose + (size_t)bx * NPP;
                    pp[0] = a; pp[1] = b; pp[2] = 0.0; pp[3] = 0.0;
                }
            }
        }
    }
    MF_STAMP(31);
}

template <int NRW, int MODE>
__global__ __launch_bounds__(MODE == 2 ? MF_THREADS2 : MF_THREADS) void k_bcr_factor_mf(Dev d, int lev, int top, int which, int nblocks, int ns, int rlo, int nrt, int solve) {
    __shared__ FactorLds S;
    extern __shared__ __align__(16) double mf_dyn_lds[];
#ifdef SSBA_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 500) {
        d.dbg[4096 + 8 * (int)blockIdx.x] = wall_clock64();       // entry of the workgroup on the chip-wide 100 MHz clock
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d.dbg[4096 + 8 * (int)blockIdx.x + 5] = 1000 + (xcc & 0xf);      // the XCD it runs on
    }
#endif
    if (MODE == 2) {
        if (threadIdx.x < MF_THREADS) factor_body<0, 2, 1>(d, S, mf_dyn_lds, lev, top, which, nblocks, ns, rlo, nrt, solve);
        else factor_body<NRW, 2, 2>(d, S, mf_dyn_lds, lev, top, which, nblocks, ns, rlo, nrt, solve);
    } else {
        factor_body<NRW, MODE, 0>(d, S, mf_dyn_lds, lev, top, which, nblocks, ns, rlo, nrt, solve);
    }
}

// ---- reduce ---------------------------------------------------------------------------------------------------
constexpr int RS = 80;                      // LDS row stride of a staged operand (columns 72..79: yr | zeros)
constexpr int RED_OPERAND_DOUBLES = 3 * BD * RS;
constexpr int RED_LDS_DOUBLES = RED_OPERAND_DOUBLES + 4 * 16 * 17;
constexpr int STAGE_NLD = (BD * BD / 2 + MF_THREADS - 1) / MF_THREADS;     // 11 double2 per lane and operand

struct StageRegs { double2 v[STAGE_NLD]; double y; };

// global reads of a BD x BD row-major block (or zeros) + its right-hand-side column: issued, not waited for
static __device__ __forceinline__ void stage_issue(StageRegs &R, const double *__restrict__ src, const double *__restrict__ yv) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
#pragma unroll
    for (int q = 0; q < STAGE_NLD; ++q) {
        const int e = threadIdx.x + q * MF_THREADS;
        R.v[q] = (src && e < BD * BD / 2) ? s2[e] : make_double2(0.0, 0.0);
    }
    R.y = (threadIdx.x < BD && yv) ? yv[threadIdx.x] : 0.0;
}
static __device__ __forceinline__ void stage_write(double *dst, const StageRegs &R) {
#pragma unroll
    for (int q = 0; q < STAGE_NLD; ++q) {
        const int e = threadIdx.x + q * MF_THREADS;
        if (e >= BD * BD / 2) continue;
        const int r = (2 * e) / BD, c = 2 * e - r * BD;
        *reinterpret_cast<double2 *>(dst + r * RS + c) = R.v[q];
    }
    if (threadIdx.x < BD) {
        double *p = dst + threadIdx.x * RS + BD;
        p[0] = R.y;
#pragma unroll
        for (int c = 1; c < 8; ++c) p[c] = 0.0;
    }
}

// operands of one product  A(:, 16 ti ..)^T B(:, 16 tj ..)  over the BD staged rows: 36 LDS reads, then 18 instructions.
// The reads of the next product are issued before the instructions of the current one (the compiler would otherwise
// sink every read to its use and pay an LDS round trip per instruction pair).
struct TnOps { double a[BD / 4], b[BD / 4]; };
static __device__ __forceinline__ void tn_load(TnOps &O, const double *sA, const double *sB, int ti, int tj, int g, int j) {
    const double *pa = sA + g * RS + 16 * ti + j, *pb = sB + g * RS + 16 * tj + j;
#pragma unroll
    for (int s = 0; s < BD / 4; ++s) { O.a[s] = pa[4 * s * RS]; O.b[s] = pb[4 * s * RS]; }
}
static __device__ __forceinline__ mf_d4 tn_mma(const TnOps &O, mf_d4 acc) {
#pragma unroll
    for (int s = 0; s < BD / 4; ++s) acc = mf(O.a[s], O.b[s], acc);
    return acc;
}
// keeps the operand reads issued above it ahead of the matrix instructions below it (memory clobber for the compiler's
// middle end, a scheduling barrier for the machine scheduler, which would otherwise sink the reads again)

// Gram products of one block's own factor outputs (fused plan, see PcrFused): units 0..14 the upper tiles of
// GLL = [YL | yr]^T [YL | yr] (column 72 = YL^T yr), 15..29 the same for YU, 30..54 the tiles of GUL = YU^T YL, which is
// stored in both orientations.  u -> (kind, tile row, tile column).
static __device__ __forceinline__ void gram_tile(int u, int &kind, int &ti, int &tj) {
    kind = u < 15 ? 0 : u < 30 ? 1 : 2;
    if (kind == 2) {
        const int c = u - 30;
        ti = c / 5;
        tj = c - 5 * ti;
    } else {
        const int n = u - 15 * kind;       // n -> (ti, tj): rows of the upper triangle start at 0, 5, 9, 12, 14
        const int a = n >= 14 ? 4 : n >= 12 ? 3 : n >= 9 ? 2 : n >= 5 ? 1 : 0;
        const int st0 = a == 4 ? 14 : a == 3 ? 12 : a == 2 ? 9 : a == 1 ? 5 : 0;
        ti = a;
        tj = min(a + (n - st0), NDT - 1);
    }
}
static __device__ __forceinline__ void gram_store(const FactorOps &o, int kind, int ti, int tj, const double (&v)[4], double *scr, int lane) {
    const int g = lane >> 4, j = lane & 15;
    const int r0 = 16 * ti + g, c0 = 16 * tj + j;
    if (kind < 2) {
        double *dst = kind == 0 ? o.oGLL : o.oGUU, *gv = kind == 0 ? o.ogL : o.ogU;
        if (c0 < BD) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + 4 * q < BD) dst[(r0 + 4 * q) * BD + c0] = v[q];
        } else if (c0 == BD) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + 4 * q < BD) gv[r0 + 4 * q] = v[q];
        }
        return;
    }
    if (c0 < BD) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r0 + 4 * q < BD) o.oGUL[(r0 + 4 * q) * BD + c0] = v[q];
    }
    // the transposed copy leaves through a per-wave LDS tile, as full row segments
#pragma unroll
    for (int q = 0; q < 4; ++q) scr[j * 17 + 4 * q + g] = v[q];
    MF_FENCE();
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = scr[(4 * q + g) * 17 + j];
    MF_FENCE();
    const int r1 = 16 * tj + g, c1 = 16 * ti + j;
    if (c1 < BD) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r1 + 4 * q < BD) o.oGULT[(r1 + 4 * q) * BD + c1] = r[q];
    }
}

// One block's reduction as a job of three staged operands -- A0 = YU(prev) | yr(prev), A1 = YL(next) | yr(next),
// A2 = YL(prev) -- and 55 units of 18 instructions: the 15 upper tiles of  D' = D - A0^T A0 - A1^T A1  (two units
// each; column 72 of the staged operands makes r' = r - A0^T yr - A1^T yr fall out of the same tiles) and the 25 tiles
// of the new coupling  -(A0^T A2).  Three workgroups (12 waves) share the units, at most five per wave.
struct ReduceJob {
    const double *a0, *ya0, *a1, *ya1, *a2;
    const double *dbase, *rbase;            // D and r of the block (inputs of the symmetric part)
    double *dout, *rout;                    // D', r'
    double *out, *outT;                     // -(A0^T A2) and its transpose (either may be null)
    bool sym, cpl;                          // which parts exist (chain ends, pinned blocks)
    // border columns riding along (null without them):  B' = B - A0(:, :72)^T yB(prev) - A1(:, :72)^T yB(next),
    // BD x NBP row-major each -- what k_bcrm_upd (ssba_border.hip) did in a launch of its own
    const double *yb0, *yb1, *bbase;
    double *bout;
};
constexpr int BRS = 40;                     // LDS row stride of a staged yB: two of them fit exactly where A2 sat

#ifdef SSBA_STAMPS
#define RJ_STAMP(i) do { if (stamp_here && (threadIdx.x & 63) == 0) dbg[2048 + (threadIdx.x >> 6) * 64 + 16 * wg + (i)] = clock64(); } while (0)
#else
#define RJ_STAMP(i) do { } while (0)
#endif
// wg / nwg: this workgroup's index among the workgroups of the job.  dead: the solver state says "nothing to do"
// (read by the caller; tested here, after the operand reads are in flight).
static __device__ __forceinline__ void reduce_job(const ReduceJob &J, double *lds, int wg, int nwg, const StateFlags &sf, unsigned long long *dbg, bool stamp_here) {
    double *sA0 = lds, *sA1 = lds + BD * RS, *sA2 = lds + 2 * BD * RS;
    const int t = threadIdx.x, lane = t & 63, g = lane >> 4, j = lane & 15, w = t >> 6;
    const int W = 4 * wg + w, NW = 4 * nwg;     // wave index among the job's waves
    RJ_STAMP(0);
    StageRegs R0, R1, R2;
    stage_issue(R0, (J.sym || J.cpl) ? J.a0 : nullptr, J.sym ? J.ya0 : nullptr);
    stage_issue(R1, J.sym ? J.a1 : nullptr, J.sym ? J.ya1 : nullptr);
    stage_issue(R2, J.cpl ? J.a2 : nullptr, nullptr);
    if (sf.dead()) return;
    // units of this wave: symmetric tiles n = W, W + NW (< 15); coupling tiles: the first ones go to the waves with
    // a single symmetric tile so that nobody gets more than five units
    constexpr int NSYM = 2, NCPL = 4;
    int sti[NSYM], stj[NSYM], cti[NCPL], ctj[NCPL];
    bool shave[NSYM], chave[NCPL];
    double base[NSYM][4], rb[NSYM][4];
    // waves 0 .. nextra-1 hold two symmetric tiles (none in a job without a symmetric part: coupling to a pinned block)
    const int nextra = (J.sym && 15 - NW > 0) ? 15 - NW : 0;
#pragma unroll
    for (int sl = 0; sl < NSYM; ++sl) {
        const int n = W + NW * sl;
        shave[sl] = J.sym && n < 15;
        // n -> (ti, tj): rows of the upper triangle start at 0, 5, 9, 12, 14
        const int a = n >= 14 ? 4 : n >= 12 ? 3 : n >= 9 ? 2 : n >= 5 ? 1 : 0;
        const int st0 = a == 4 ? 14 : a == 3 ? 12 : a == 2 ? 9 : a == 1 ? 5 : 0;
        sti[sl] = a;
        stj[sl] = min(a + (n - st0), NDT - 1);
        // the entries of D / r this lane finishes: fetched now, under the staging
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rowc = min(16 * sti[sl] + 4 * q + g, BD - 1), colc = min(16 * stj[sl] + j, BD - 1);
            base[sl][q] = shave[sl] ? J.dbase[rowc * BD + colc] : 0.0;
            rb[sl][q] = shave[sl] ? J.rbase[rowc] : 0.0;
        }
    }
    {
        // 25 coupling tiles over the waves: waves with two symmetric tiles take one, the others share the rest evenly
        const int light = NW - nextra;                  // waves with one symmetric tile
#pragma unroll
        for (int sl = 0; sl < NCPL; ++sl) {
            int c;
            if (W < nextra) c = sl == 0 ? W : 25;
            else c = nextra + (W - nextra) + light * sl;
            chave[sl] = J.cpl && c < 25;
            c = min(c, 24);
            cti[sl] = c / 5;
            ctj[sl] = c - 5 * cti[sl];
        }
    }
    const bool p0 = J.a0 != nullptr, p1 = J.a1 != nullptr;
    // products in order, operand reads one product ahead of the instructions (two register sets, alternating).
    // (Tried: grouping the products by operand and writing A1 / A2 to LDS between the instruction batches so that
    // their staging hides under the A0 products -- the two extra barriers cost more than the overlap gains, because
    // the waves with two symmetric tiles make every phase wait: 12.8 -> 14.1 us per launch.)
    TnOps O0, O1;
    mf_d4 sacc[NSYM], cacc[NCPL];
#pragma unroll
    for (int sl = 0; sl < NSYM; ++sl) sacc[sl] = mf_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sl = 0; sl < NCPL; ++sl) cacc[sl] = mf_d4{0.0, 0.0, 0.0, 0.0};
    const bool u0 = shave[0] && p0, u1 = shave[0] && p1, u2 = shave[1] && p0, u3 = shave[1] && p1;
    stage_write(sA0, R0);
    stage_write(sA1, R1);
    stage_write(sA2, R2);
    RJ_STAMP(1);
    __syncthreads();
    RJ_STAMP(2);
    if (u0) tn_load(O0, sA0, sA0, sti[0], stj[0], g, j);
    MF_FENCE();
    if (u1) tn_load(O1, sA1, sA1, sti[0], stj[0], g, j);
    MF_FENCE();
    if (u0) sacc[0] = tn_mma(O0, sacc[0]);
    RJ_STAMP(4);
    MF_FENCE();
    if (u2) tn_load(O0, sA0, sA0, sti[1], stj[1], g, j);
    MF_FENCE();
    if (u1) sacc[0] = tn_mma(O1, sacc[0]);
    RJ_STAMP(5);
    MF_FENCE();
    if (u3) tn_load(O1, sA1, sA1, sti[1], stj[1], g, j);
    MF_FENCE();
    if (u2) sacc[1] = tn_mma(O0, sacc[1]);
    RJ_STAMP(6);
    MF_FENCE();
    if (chave[0]) tn_load(O0, sA0, sA2, cti[0], ctj[0], g, j);
    MF_FENCE();
    if (u3) sacc[1] = tn_mma(O1, sacc[1]);
    RJ_STAMP(7);
    MF_FENCE();
    if (chave[1]) tn_load(O1, sA0, sA2, cti[1], ctj[1], g, j);
    MF_FENCE();
    if (chave[0]) cacc[0] = tn_mma(O0, cacc[0]);
    RJ_STAMP(8);
    MF_FENCE();
    if (chave[2]) tn_load(O0, sA0, sA2, cti[2], ctj[2], g, j);
    MF_FENCE();
    if (chave[1]) cacc[1] = tn_mma(O1, cacc[1]);
    RJ_STAMP(9);
    MF_FENCE();
    if (chave[3]) tn_load(O1, sA0, sA2, cti[3], ctj[3], g, j);
    MF_FENCE();
    if (chave[2]) cacc[2] = tn_mma(O0, cacc[2]);
    RJ_STAMP(10);
    MF_FENCE();
    if (chave[3]) cacc[3] = tn_mma(O1, cacc[3]);
    RJ_STAMP(11);
    // border columns: their operand reads are issued here and land while the tiles above are stored
    constexpr int YB_NLD = (BD * NBP / 2 + MF_THREADS - 1) / MF_THREADS;      // 5 double2 per lane and operand
    double2 yv0[YB_NLD], yv1[YB_NLD];
    const int bti = min(W >> 1, NDT - 1), btj = W & 1;       // this wave's 16 x 16 tile of the BD x NBP columns (10 tiles)
    const bool bhave = J.bout != nullptr && W < 2 * NDT;
    double bb[4];
    if (J.bout) {
#pragma unroll
        for (int q = 0; q < YB_NLD; ++q) {
            const int e = t + q * MF_THREADS;
            const bool in = e < BD * NBP / 2;
            yv0[q] = (J.yb0 && in) ? reinterpret_cast<const double2 *>(J.yb0)[e] : make_double2(0.0, 0.0);
            yv1[q] = (J.yb1 && in) ? reinterpret_cast<const double2 *>(J.yb1)[e] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rowc = min(16 * bti + 4 * q + g, BD - 1);
            bb[q] = bhave ? J.bbase[rowc * NBP + 16 * btj + j] : 0.0;
        }
    }
    // ---- stores.  The mirror images (lower triangle of D', transposed coupling) go through a per-wave LDS tile so
    //      that they leave as 128-byte row segments too instead of 64 scattered doubles per instruction ----------
    double *scr = lds + RED_OPERAND_DOUBLES + w * (16 * 17);      // a private tile per wave, behind the staged operands
    auto transposed = [&](const mf_d4 &v) {
#pragma unroll
        for (int q = 0; q < 4; ++q) scr[j * 17 + 4 * q + g] = v[q];
        MF_FENCE();
        mf_d4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = scr[(4 * q + g) * 17 + j];
        MF_FENCE();
        return r;
    };
#pragma unroll
    for (int sl = 0; sl < NSYM; ++sl) {
        if (!shave[sl]) continue;
        mf_d4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = base[sl][q] - sacc[sl][q];
        const int r0 = 16 * sti[sl] + g, c0 = 16 * stj[sl] + j;
        if (c0 < BD) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + 4 * q < BD) J.dout[(r0 + 4 * q) * BD + c0] = v[q];
        } else if (c0 == BD) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + 4 * q < BD) J.rout[r0 + 4 * q] = rb[sl][q] - sacc[sl][q];
        }
        if (sti[sl] != stj[sl]) {       // the tile below the diagonal: rows of tile column tj, columns of tile row ti
            const mf_d4 vt = transposed(v);
            const int r1 = 16 * stj[sl] + g, c1 = 16 * sti[sl] + j;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r1 + 4 * q < BD) J.dout[(r1 + 4 * q) * BD + c1] = vt[q];
        }
    }
#pragma unroll
    for (int sl = 0; sl < NCPL; ++sl) {
        if (!chave[sl]) continue;
        mf_d4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = -cacc[sl][q];
        const int r0 = 16 * cti[sl] + g, c0 = 16 * ctj[sl] + j;
        if (J.out && c0 < BD) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + 4 * q < BD) J.out[(r0 + 4 * q) * BD + c0] = v[q];
        }
        if (J.outT) {
            const mf_d4 vt = transposed(v);
            const int r1 = 16 * ctj[sl] + g, c1 = 16 * cti[sl] + j;
            if (c1 < BD) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r1 + 4 * q < BD) J.outT[(r1 + 4 * q) * BD + c1] = vt[q];
            }
        }
    }
    RJ_STAMP(12);
    if (J.bout) {
        double *sY0 = sA2, *sY1 = sA2 + BD * BRS;
        __syncthreads();            // every wave is through with A2
#pragma unroll
        for (int q = 0; q < YB_NLD; ++q) {
            const int e = t + q * MF_THREADS;
            if (e >= BD * NBP / 2) continue;
            const int r = (2 * e) / NBP, c = 2 * e - r * NBP;
            *reinterpret_cast<double2 *>(sY0 + r * BRS + c) = yv0[q];
            *reinterpret_cast<double2 *>(sY1 + r * BRS + c) = yv1[q];
        }
        __syncthreads();
        if (bhave) {
            mf_d4 acc = mf_d4{0.0, 0.0, 0.0, 0.0};
            const double *pa0 = sA0 + g * RS + 16 * bti + j, *pa1 = sA1 + g * RS + 16 * bti + j;
            const double *pb0 = sY0 + g * BRS + 16 * btj + j, *pb1 = sY1 + g * BRS + 16 * btj + j;
            double a[BD / 4], b[BD / 4];
            if (J.yb0) {
#pragma unroll
                for (int k = 0; k < BD / 4; ++k) { a[k] = pa0[4 * k * RS]; b[k] = pb0[4 * k * BRS]; }
                MF_FENCE();
#pragma unroll
                for (int k = 0; k < BD / 4; ++k) acc = mf(a[k], b[k], acc);
            }
            if (J.yb1) {
#pragma unroll
                for (int k = 0; k < BD / 4; ++k) { a[k] = pa1[4 * k * RS]; b[k] = pb1[4 * k * BRS]; }
                MF_FENCE();
#pragma unroll
                for (int k = 0; k < BD / 4; ++k) acc = mf(a[k], b[k], acc);
            }
            const int r0 = 16 * bti + g;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + 4 * q < BD) J.bout[(r0 + 4 * q) * NBP + 16 * btj + j] = bb[q] - acc[q];
        }
    }
}

// 1-D grid of xcd_grid(blocks, ny) workgroups, ny = 3 (+ 2 for the coupling to a pinned last block).  Parallel cyclic
// reduction (which >= 2): D', r' in place, L' (+ its transpose) into the plan's buffers.  Plain levels (which = 0):
// D', r', L' of the next level.  Same operand rules as k_bcr_reduce (ssba_bcr.hip).
__global__ __launch_bounds__(MF_THREADS) void k_bcr_reduce_mf(Dev d, int lev, int which, int nblocks, int ny, int ride, int x_lo) {
    const StateFlags sf = state_flags_vmem(d.st);       // tested after the operand reads have been issued (ssba_device.h)
    extern __shared__ __align__(16) double lds[];
    int bx, y;
    if ((int)blockIdx.x < nblocks * ny) xcd_map((int)blockIdx.x, nblocks, ny, bx, y);
    else {      // tail of the grid: the two extra workgroups of the blocks x_lo.. that couple to a pinned last block
        const int j = (int)blockIdx.x - nblocks * ny;
        bx = x_lo + (j >> 1);
        y = 3 + (j & 1);
    }
    ReduceJob J;
    J.a0 = J.a1 = J.a2 = J.ya0 = J.ya1 = J.dbase = J.rbase = nullptr;
    J.dout = J.rout = J.out = J.outT = nullptr;
    J.yb0 = J.yb1 = J.bbase = nullptr; J.bout = nullptr;
    J.sym = J.cpl = false;
    int wg = y, nwg = 3;
    if (which >= 2) {
        const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
        const BcrLevel &B = which == 3 ? d.slev[0] : d.lev[d.pcr.level];
        const int e = bx, s = 1 << lev, prev = e - s, next = e + s, last = B.n - 1;
        const int lo = P.pin0 ? 1 : 0, hi = P.pin1 ? last - 1 : last;
        const bool hasPrev = prev >= lo, hasNext = next <= hi;
        const size_t so = P.keep ? (size_t)lev * B.n : 0;
        if (y < 3) {
            J.sym = hasPrev || hasNext;
            J.cpl = hasPrev && (prev - s >= 0 || P.pin0);
            if (hasPrev) { J.a0 = P.YU + (so + prev) * BD * BD; J.ya0 = P.yr + (size_t)prev * BD; }
            if (hasNext) { J.a1 = P.YL + (so + next) * BD * BD; J.ya1 = P.yr + (size_t)next * BD; }
            if (J.cpl) J.a2 = P.YL + (so + prev) * BD * BD;
            J.dbase = J.dout = B.D + (size_t)e * BD * BD;
            J.rbase = J.rout = B.r + (size_t)e * BD;
            J.out = P.Lbuf + (size_t)e * BD * BD;
            J.outT = P.LbufT + (size_t)e * BD * BD;
            if (ride && which == 2 && J.sym) {
                if (hasPrev) J.yb0 = P.yB + (size_t)prev * BD * NBP;
                if (hasNext) J.yb1 = P.yB + (size_t)next * BD * NBP;
                J.bbase = J.bout = P.Bb + (size_t)e * BD * NBP;
            }
        } else {
            // e + s is folded and its far side is the pinned last block: that coupling has no transposed twin in the
            // pinned block's own row, so it is computed here (two workgroups)
            if (!P.pin1 || (P.pin0 && e == 0) || !hasNext || next + s <= last) return;
            J.cpl = true;
            J.a0 = P.YL + (so + next) * BD * BD;
            J.a2 = P.YU + (so + next) * BD * BD;
            J.out = P.Ubuf + (size_t)e * BD * BD;
            wg = y - 3; nwg = 2;
        }
    } else {
        const BcrLevel &L = d.lev[lev];
        const BcrLevel &N = d.lev[lev + 1];
        const int m = bx, e = 2 * m, t = threadIdx.x;
        if (L.pin && m == L.n / 2) {
            // pinned end of a partitioned chain: carried over unchanged (see k_bcr_reduce)
            if (sf.dead()) return;
            const int src = L.n - 1;
            if (y == 0) {
                const double2 *s2 = reinterpret_cast<const double2 *>(L.D + (size_t)src * BD * BD);
                double2 *d2 = reinterpret_cast<double2 *>(N.D + (size_t)m * BD * BD);
                for (int i = t; i < BD * BD / 2; i += MF_THREADS) d2[i] = s2[i];
                if (t < BD) N.r[(size_t)m * BD + t] = L.r[(size_t)src * BD + t];
            } else if (y == 1) {
                const double *sl = L.L + (size_t)src * BD * BD;
                double *dl = N.L + (size_t)m * BD * BD;
                const bool tr = (m & 1) == 0;
                for (int i = t; i < BD * BD; i += MF_THREADS) {
                    const int r = i / BD, c = i - r * BD;
                    dl[tr ? c * BD + r : i] = sl[i];
                }
            }
            return;
        }
        if (y >= 3) return;
        const bool hasPrev = e - 1 >= 0, hasNext = e + 1 < L.n && !(L.pin && e + 1 == L.n - 1);
        const int tp = (e - 2) / 2;
        J.sym = true;
        J.cpl = m > 0;
        if (hasPrev) { J.a0 = L.YU + (size_t)tp * BD * BD; J.ya0 = L.r + (size_t)(e - 1) * BD; }
        if (hasNext) { J.a1 = L.L + (size_t)(e + 1) * BD * BD; J.ya1 = L.r + (size_t)(e + 1) * BD; }
        if (J.cpl) J.a2 = L.L + (size_t)(e - 1) * BD * BD;
        J.dbase = L.D + (size_t)e * BD * BD;
        J.rbase = L.r + (size_t)e * BD;
        J.dout = N.D + (size_t)m * BD * BD;
        J.rout = N.r + (size_t)m * BD;
        // an even-indexed coupling block of the next level is stored transposed
        if ((m & 1) == 0) J.outT = N.L + (size_t)m * BD * BD;
        else J.out = N.L + (size_t)m * BD * BD;
        if (ride) {
            if (hasPrev) J.yb0 = L.B + (size_t)(e - 1) * BD * NBP;
            if (hasNext) J.yb1 = L.B + (size_t)(e + 1) * BD * NBP;
            J.bbase = L.B + (size_t)e * BD * NBP;
            J.bout = N.B + (size_t)m * BD * NBP;
        }
    }
    reduce_job(J, lds, wg, nwg, sf, d.dbg, bx == 5);
}

// ---- host side --------------------------------------------------------------------------------------------------
// solve: 0 no, 1 the decoupled blocks solve themselves, 2 ... and update their poses
void launch_bcr_factor_mf(Launcher &L, const Dev &d, int nblocks, int lev, int top, int which, bool coupled, bool ride, int solve) {
    // workgroups per block: fill the chip when the level is short; blocks without couplings (the decoupled last step)
    // have no right-hand-side tiles to share out
    // Several workgroups per block all read D and r while the first of them writes G and yr: only where those go to
    // buffers of their own (the steps of a parallel plan before its last one).  Plain levels and last steps work in
    // place (G over D, yr over r) and keep one workgroup per block.
    // ride: two more column tiles (the border columns); a fourth workgroup per block keeps wave 0 of every workgroup
    // -- the one that owns the pivots -- free of right-hand-side tiles (the factor kernel's LDS lets 3 workgroups share a CU)
    const bool in_place = top || which < 2;
    const int rlo = (!coupled && ride) ? NRT : 0, nrt = ride ? NRT + 2 : NRT;
    const int ns = (!coupled || in_place) ? 1 : nblocks <= 85 ? (ride ? 4 : 3) : nblocks <= 128 ? 2 : 1;
    const int per_wg = (nrt - rlo + ns - 1) / ns;
    const int grid = xcd_grid(nblocks, ns);
    const size_t sh_solve = (size_t)(BD * BD + 2 * BD) * sizeof(double);
    if (!coupled && !ride) LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<0, 0>), dim3(grid), dim3(MF_THREADS), solve ? sh_solve : 0, d, lev, top, which, nblocks, ns, rlo, nrt, solve);
    else if (per_wg <= 3) LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<1, 0>), dim3(grid), dim3(MF_THREADS), 0, d, lev, top, which, nblocks, ns, rlo, nrt, 0);
    else if (per_wg <= 7) LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<2, 0>), dim3(grid), dim3(MF_THREADS), 0, d, lev, top, which, nblocks, ns, rlo, nrt, 0);
    else LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<3, 0>), dim3(grid), dim3(MF_THREADS), 0, d, lev, top, which, nblocks, ns, rlo, nrt, 0);
}

// Fused plan (PcrFused): step q = factorisation of every block at stride 2^q + the Gram products the next step assembles its
// operands from, ONE launch; the workgroups of a block (3 up to 85 blocks, 2 up to 128) repeat the factorisation and share
// the Gram tiles.  The decoupled last step assembles its blocks the same way, solves them and (solve = 2) updates the poses.
constexpr size_t FUSED_LDS = (size_t)2 * BD * 80 * sizeof(double);
void launch_pcr_fused_step(Launcher &L, const Dev &d, int n, int q, int which) {
    const int ns = n <= 85 ? 3 : 2;         // (a plan has at most PCR_MAX_BLOCKS = 128 blocks; the kernel's tile share assumes ns >= 2; 2 at C2: 0.372 against 0.364 ms)
    LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<3, 2>), dim3(xcd_grid(n, ns)), dim3(MF_THREADS2), FUSED_LDS, d, q, 0, which, n, ns, 0, NRT, 0);
}
void launch_pcr_fused_top(Launcher &L, const Dev &d, int n, int steps, int solve, int which) {
    const size_t sh_solve = (size_t)(BD * BD + 2 * BD) * sizeof(double);
    const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
    if (P.pin0 || P.pin1) {     // a chain with pinned ends: its blocks keep their couplings to those (right-hand-side tiles), nothing is solved yet
        LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<3, 1>), dim3(xcd_grid(n, 1)), dim3(MF_THREADS), 0, d, steps, 1, which, n, 1, 0, NRT, 0);
        return;
    }
    LAUNCH(KC_BCR_FACTOR, (k_bcr_factor_mf<0, 1>), dim3(xcd_grid(n, 1)), dim3(MF_THREADS), solve ? sh_solve : 0, d, steps, 1, which, n, 1, 0, NRT, solve);
}

// ny_legacy: 2, or 3 with the coupling to a pinned last block (the grid.y of k_bcr_reduce)
void launch_bcr_reduce_mf(Launcher &L, const Dev &d, int nblocks, int ny_legacy, int lev, int which, bool ride) {
    // ny_legacy == 3 (a partitioned chain whose last block is pinned): the blocks e with e + s inside the chain and
    // e + 2 s beyond its last block also compute the coupling of e to that block -- two more workgroups each, listed
    // after the 3 n regular ones (a 5 n grid with 2 n - 2 s workgroups that return at once held a CU's LDS each while
    // they did: 17-21 us per launch against 11.6)
    int n_extra = 0, x_lo = 0;
    if (ny_legacy == 3 && which >= 2) {
        const PcrPlan &P = which == 3 ? d.spcr : d.pcr;
        const BcrLevel &B = which == 3 ? d.slev[0] : d.lev[d.pcr.level];
        const int s = 1 << lev, last = B.n - 1, hi = P.pin1 ? last - 1 : last;
        x_lo = std::max(P.pin0 ? 1 : 0, last - 2 * s + 1);
        const int x_hi = hi - s;
        if (P.pin1 && x_hi >= x_lo) n_extra = x_hi - x_lo + 1;
    }
    LAUNCH(KC_BCR_REDUCE, k_bcr_reduce_mf, dim3(xcd_grid(nblocks, 3) + 2 * n_extra), dim3(MF_THREADS), (size_t)RED_LDS_DOUBLES * sizeof(double), d, lev, which, nblocks, 3, ride ? 1 : 0, x_lo);
}

int configure_bcr_mf() {
    if (hipFuncSetAttribute((const void *)k_bcr_reduce_mf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RED_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcr_factor_mf<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((BD * BD + 2 * BD) * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcr_factor_mf<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((BD * BD + 2 * BD) * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_bcr_factor_mf<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS) != hipSuccess) return -1;
    return 0;
}

}  // namespace ssba
