// plp_cheby_r_impl.hpp -- kernels of plp_cheby_r.hip / plp_lp_r.hip (two translation units only to keep the build
// parallel): stand-alone LP batches with four rows per lane (gfx950): Chebyshev-ball LPs (form F1,
// polytope/polytope.py:1283-1288) and generic LPs whose origin is feasible:
//
//   lp_r_kernel<N,GS>       : lpsolve() batches (solvers.py:76-106), n <= 8, origin feasible (no phase 1)
//   lp_p1_r_kernel<N,GS>    : the LPs of such a batch that need phase 1 (some h_i < 0)
//   cheby_r_kernel<D,GS>    : a batch of polytopes (cheby_ball / is_fulldim, :1241-1300, :962-985)
//   adjacent_r_kernel<D,GS> : all pairs of n cells (is_adjacent(overlap=True), :1843-1866, under the pair
//                             loop of find_adjacent_regions, prop2partition.py:57-61)
//
// A polytope (or a stacked pair) of up to 16 / 32 / 64 rows takes a group of GS = 4 / 8 / 16 lanes, lane l
// holding rows 4l..4l+3 in VGPRs, so a wavefront carries 16 / 8 / 4 LPs (plp_simplex_r.hpp).  The LP runs
// on the fast pivot path (forced first pivot "r enters, row argmin b_i/||a_i|| leaves", then Dantzig);
// when it ends with ST_RETRY (a dictionary that needs Bland's rule) the wavefront rebuilds the LP and
// solves it with the general engine, behind a wave-uniform branch that is almost never taken.
#pragma once
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_simplex_r.hpp"

namespace plp {
// One wavefront per workgroup for the batch kernels of this file (they use no LDS and no barrier): the tail of a
// launch is balanced wave by wave.  Against 256 threads (rocprofv3 kernel durations): lp_r_kernel<6,8> 271 -> 172 us,
// <16,32> 103 -> 81, <8,16> 186 -> 142, cheby_r_kernel<16,32> 692 -> 590, adjacent_r_kernel<4,4> 188 -> 181,
// lp_p1_r_kernel<3,4> 250 -> 199; cheby_r_kernel<3,4> and lp_r_kernel<3,4> at 100 k LPs unchanged.
#ifndef PLP_R_BLOCK
#define PLP_R_BLOCK 64
#endif
constexpr int RBLK = PLP_R_BLOCK;  // threads per workgroup

namespace {

#ifndef PLP_ROWS_BIG_D
#define PLP_ROWS_BIG_D 2
#endif
// rows per lane: 4 for d <= 8; 2 for d = 9..16, where four rows of up to 17 columns would not fit the VGPR file
template <int D>
struct RowsPerLane { static constexpr int value = D <= 8 ? 4 : PLP_ROWS_BIG_D; };

// Solve the Chebyshev LP of the rows handed out by `rowA(rr, kk)` / `rowb(rr)` (rr = row index < m).
// Returns the LP status; x[0..D-1] = centre, x[D] = radius, replicated over the group (valid if status 0).
template <int D, int GS, int R, bool FAST, class FA, class FB>
__device__ __forceinline__ int cheby_r_lp(const Grp& g, bool valid, int m, int row0, FA rowA, FB rowb, double* x) {
    SimplexR<D + 1, R, !FAST, true> S;
    S.reset(D + 1, m, row0);
    double qi[R];
    unsigned actb = 0u;
    bool inf0 = false, finite = true;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool h = valid & (row0 + k < m) & (m <= GS * R);
        double nrm2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            const double v = h ? rowA(row0 + k, kk) : 0.0;
            S.T[k][kk] = v;
            nrm2 = nrm2 + v * v;
            finite = finite & isfinite(v);
        }
        const double bk = h ? rowb(row0 + k) : 0.0;
        finite = finite & isfinite(bk);
        const double nrm = sqrt(nrm2);
        const bool zero = !(nrm > 0.0);
        const bool on = h & !zero;
        S.T[k][D] = on ? nrm : 0.0;
        S.beta[k] = on ? bk : 0.0;
        qi[k] = bk / nrm;
        actb |= on ? (1u << k) : 0u;
        inf0 = inf0 | (h & zero & (bk < -TOL_FEAS));
    }
    S.ract = actb;
    const bool infeasible0 = grp_ballot(inf0, g) != 0;
    const bool bad = (grp_ballot(!finite, g) != 0) | (m > GS * R);
    S.cost[D] = -1.0;
    if constexpr (FAST) {
        S.mode = M_P2;
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) S.init_q[k] = qi[k];
        S.init_elig = actb;
        S.mode = M_INIT;
        S.init_col = D;
        S.mode_after_init = M_P2;
    }
    if (!valid | bad) { S.mode = M_DONE; S.status = ST_NUM; }
    else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
    if constexpr (FAST) S.template run_fast<GS, true>(g, qi, actb);
    else S.run(g);
#pragma unroll
    for (int j = 0; j <= D; ++j) {
        bool found;
        const double mine = S.x_of(j, found);
        const uint64_t ob = grp_ballot(found, g);
        const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
        x[j] = ob ? v : 0.0;
    }
    return S.status;
}

template <int D, int GS, int R, class FA, class FB>
__device__ __forceinline__ int cheby_r_solve(const Grp& g, bool valid, int m, int row0, FA rowA, FB rowb, double* x,
                                             int force_retry) {
    int st = cheby_r_lp<D, GS, R, true>(g, valid, m, row0, rowA, rowb, x);
    if (force_retry) st = ST_RETRY;  // test hook (PLP_CHEBY_RETRY_ALL=1): every LP takes the hand-over below
    if (__any(st == ST_RETRY)) {  // rare: redo with the general engine (Bland's rule available)
        double x2[D + 1];
        const int st2 = cheby_r_lp<D, GS, R, false>(g, valid & (st == ST_RETRY), m, row0, rowA, rowb, x2);
        if (st == ST_RETRY) {
            st = st2;
#pragma unroll
            for (int j = 0; j <= D; ++j) x[j] = x2[j];
        }
    }
    return st;
}

}  // namespace

template <int D, int GS>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void cheby_r_kernel(long long B, int m_max,
                                                                          const double* __restrict__ A,
                                                                          const double* __restrict__ b,
                                                                          const int* __restrict__ mrows,
                                                                          double* __restrict__ r,
                                                                          double* __restrict__ xc,
                                                                          int* __restrict__ status,
                                                                          int force_retry) {
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    constexpr int R = RowsPerLane<D>::value;
    const int row0 = g.gl * R;
    const long long p = (long long)blockIdx.x * gpb + gib;
    const bool valid = p < B;
    const int m = valid ? (mrows ? mrows[p] : m_max) : 0;
    // lane l loads its 4 consecutive rows straight from HBM (4*D contiguous doubles)
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, R>(
        g, valid, m, row0, [&](int rr, int kk) { return A[(p * m_max + rr) * D + kk]; },
        [&](int rr) { return b[p * m_max + rr]; }, x, force_retry);
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    if (valid & (g.gl == 0)) {
#pragma unroll
        for (int j = 0; j < D; ++j) xc[p * D + j] = (st == ST_OPT) ? x[j] : qnan;
        r[p] = (st == ST_OPT) ? x[D] : qnan;
        status[p] = st;
    }
}

// Bounding box (polytope/polytope.py:1314-1411): the reference solves 2d generic LPs per polytope (F3:
// min +-e_i.x over {Ax <= b}), each from scratch.  Here one lane group solves the Chebyshev LP of its polytope first
// and, when that yields a centre with r >= BBOX_MIN_R, starts all 2d LPs from the dictionary translated to that
// centre (primal feasible, no phase 1; the fused reduce does the same for its box), reading x_i off the optimal
// value.  status: 0 = lb/ub hold the box (+-inf where an LP is unbounded, :1376/:1398), 1 = not handled here
// (empty / flat / unbounded-ball polytopes and Bland cases): the caller solves the generic LPs for those.
constexpr double BBOX_MIN_R = 1e-6;

template <int D, int GS>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void bbox_r_kernel(long long B, int m_max,
                                                                         const double* __restrict__ A,
                                                                         const double* __restrict__ b,
                                                                         const int* __restrict__ mrows,
                                                                         double* __restrict__ lb,
                                                                         double* __restrict__ ub,
                                                                         int* __restrict__ status, int force_retry,
                                                                        signed char* __restrict__ basis8, double* __restrict__ centre) {
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    constexpr int R = RowsPerLane<D>::value;
    const int row0 = g.gl * R;
    const long long p = (long long)blockIdx.x * gpb + gib;
    const bool valid = p < B;
    const int m = valid ? (mrows ? mrows[p] : m_max) : 0;
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, R>(
        g, valid, m, row0, [&](int rr, int kk) { return A[(p * m_max + rr) * D + kk]; },
        [&](int rr) { return b[p * m_max + rr]; }, x, force_retry);
    const bool ok = valid & (st == ST_OPT) & (x[D] >= BBOX_MIN_R);
    bool handed = !ok;
    // my rows and their slacks at the centre
    double T0[R][D], be0[R];
    unsigned has = 0u;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool h = ok & (row0 + k < m);
        has |= h ? (1u << k) : 0u;
        double s = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            T0[k][kk] = h ? A[(p * m_max + row0 + k) * D + kk] : 0.0;
            s = fma(T0[k][kk], ok ? x[kk] : 0.0, s);
        }
        be0[k] = h ? fmax(b[p * m_max + row0 + k] - s, 0.0) : 0.0;
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    for (int it = 0; it < 2 * D; ++it) {  // lower_0, upper_0, lower_1, upper_1, ...
        const int kx = it >> 1;
        const bool up = it & 1;
        double xck = 0.0;
        SimplexR<D, R, false, false> S;
        S.reset(D, m, row0);
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            xck = (kk == kx) ? x[kk] : xck;
            S.cost[kk] = (kk == kx) ? (up ? -1.0 : 1.0) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
#pragma unroll
            for (int kk = 0; kk < D; ++kk) S.T[k][kk] = T0[k][kk];
            S.beta[k] = be0[k];
        }
        S.ract = has;
        S.mode = ok ? M_P2 : M_DONE;
        S.template run_fast<GS>(g);
        if (basis8 && (valid & (g.gl == 0))) {  // the final basis, for the verifier (plp_verify.hip)
#pragma unroll
            for (int kk = 0; kk < D; ++kk) basis8[((size_t)p * 2 * D + it) * D + kk] = (signed char)(S.cv[kk] < D ? -1 - S.cv[kk] : S.cv[kk] - D);
        }
        // zeta = c.x' = -negz ; x_k = xc_k + x'_k ; lower: c = +e_k, upper: c = -e_k
        double val;
        if (S.status == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
        else if (S.status == ST_UNBND) val = up ? pinf : -pinf;
        else { val = qnan; handed = true; }
        if (valid & (g.gl == 0)) (up ? ub : lb)[p * D + kx] = ok ? val : qnan;
    }
    if (valid & (g.gl == 0)) status[p] = handed ? 1 : 0;
    if (centre && (valid & (g.gl == 0))) {
#pragma unroll
        for (int kk = 0; kk < D; ++kk) centre[(size_t)p * D + kk] = ok ? x[kk] : qnan;
    }
}

// The same for small batches (latency form, cf. reduce_split_kernel): ONE polytope per wavefront, every lane group solves
// the Chebyshev LP on the same rows (identical results, nothing to exchange), then group g takes the g-th of the 2d LPs
// (round-robin).  Same engine and arithmetic per LP: outputs bitwise equal to bbox_r_kernel.
template <int D, int GS>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void bbox_split_kernel(long long B, int m_max,
                                                                             const double* __restrict__ A,
                                                                             const double* __restrict__ b,
                                                                             const int* __restrict__ mrows,
                                                                             double* __restrict__ lb,
                                                                             double* __restrict__ ub,
                                                                             int* __restrict__ status, int force_retry,
                                                                        signed char* __restrict__ basis8, double* __restrict__ centre) {
    static_assert(RBLK == 64, "one wavefront per workgroup: the polytope's verdict is a wave-wide vote");
    const Grp g(GS);
    constexpr int NGRP = RBLK / GS;
    const int grp = threadIdx.x / GS;
    constexpr int R = RowsPerLane<D>::value;
    const int row0 = g.gl * R;
    const long long p = blockIdx.x;
    const bool valid = p < B;
    const int m = valid ? (mrows ? mrows[p] : m_max) : 0;
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, R>(
        g, valid, m, row0, [&](int rr, int kk) { return A[(p * m_max + rr) * D + kk]; },
        [&](int rr) { return b[p * m_max + rr]; }, x, force_retry);
    const bool ok = valid & (st == ST_OPT) & (x[D] >= BBOX_MIN_R);
    bool handed = !ok;
    double T0[R][D], be0[R];
    unsigned has = 0u;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool h = ok & (row0 + k < m);
        has |= h ? (1u << k) : 0u;
        double s = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            T0[k][kk] = h ? A[(p * m_max + row0 + k) * D + kk] : 0.0;
            s = fma(T0[k][kk], ok ? x[kk] : 0.0, s);
        }
        be0[k] = h ? fmax(b[p * m_max + row0 + k] - s, 0.0) : 0.0;
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    for (int round = 0; round * NGRP < 2 * D; ++round) {  // group g: LP g, g + NGRP, ...
        const int itq = round * NGRP + grp;
        const bool mine = itq < 2 * D;
        const int it = mine ? itq : 0;
        const int kx = it >> 1;
        const bool up = it & 1;
        double xck = 0.0;
        SimplexR<D, R, false, false> S;
        S.reset(D, m, row0);
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            xck = (kk == kx) ? x[kk] : xck;
            S.cost[kk] = (kk == kx) ? (up ? -1.0 : 1.0) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
#pragma unroll
            for (int kk = 0; kk < D; ++kk) S.T[k][kk] = T0[k][kk];
            S.beta[k] = be0[k];
        }
        S.ract = has;
        S.mode = (ok & mine) ? M_P2 : M_DONE;
        S.template run_fast<GS>(g);
        if (basis8 && (valid & mine & (g.gl == 0))) {  // the final basis, for the verifier (plp_verify.hip)
#pragma unroll
            for (int kk = 0; kk < D; ++kk) basis8[((size_t)p * 2 * D + it) * D + kk] = (signed char)(S.cv[kk] < D ? -1 - S.cv[kk] : S.cv[kk] - D);
        }
        double val;
        if (S.status == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
        else if (S.status == ST_UNBND) val = up ? pinf : -pinf;
        else { val = qnan; handed = handed | mine; }
        if (valid & mine & (g.gl == 0)) (up ? ub : lb)[p * D + kx] = ok ? val : qnan;
    }
    const bool any_handed = __any(handed);  // (all lanes of the wavefront work on this one polytope)
    if (valid & (threadIdx.x == 0)) status[p] = any_handed ? 1 : 0;
    if (centre && (valid & (threadIdx.x == 0))) {
#pragma unroll
        for (int kk = 0; kk < D; ++kk) centre[(size_t)p * D + kk] = ok ? x[kk] : qnan;
    }
}

// One lane group per pair (i, j < i): the rows of both cells are stacked with b + inflate, and the pair
// counts iff the Chebyshev LP of the stack is optimal with r > thresh.  Adjacency: inflate = abs_tol,
// thresh = abs_tol / 10 (`is_fulldim(dummy, abs_tol / 10)`, polytope.py:1860-1866); overlap
// (Partition.are_disjoint, prop2partition.py:123-192: `is_fulldim(region.intersect(other))`): inflate = 0,
// thresh = abs_tol.  The stacked LP is built straight from the resident cells (n cells stay in L2), so
// nothing is staged by the host.  adj is n x n, symmetric, ones on the diagonal; with `compact` set the
// kernel instead solves the pairs p_lo <= p < p_hi (p = i (i - 1) / 2 + j) and writes compact[p - p_lo]
// (the shard of one rank when the pair space is split across GPUs).
template <int D, int GS>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void adjacent_r_kernel(
    int n, int m_max, const double* __restrict__ A, const double* __restrict__ b, const int* __restrict__ mrows,
    double inflate, double thresh, unsigned char* __restrict__ adj, long long p_lo, long long p_hi,
    unsigned char* __restrict__ compact, int force_retry, int cross_n1) {
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    constexpr int R = RowsPerLane<D>::value;
    const int row0 = g.gl * R;
    const long long p = p_lo + (long long)blockIdx.x * gpb + gib;
    const bool valid = p < p_hi;
    // p -> (i, j) with j < i, p = i (i - 1) / 2 + j;  cross pairs (cross_n1 > 0, compact only): the table holds two lists,
    // n1 cells then n - n1 cells, and p = a (n - n1) + c pairs cell a of the first with cell c of the second
    long long i = valid ? (long long)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5) : 1;
    while (i * (i - 1) / 2 > p) --i;
    while ((i + 1) * i / 2 <= p) ++i;
    long long j = valid ? p - i * (i - 1) / 2 : 0;
    if (cross_n1 > 0) {
        const long long n2 = n - cross_n1;
        i = valid ? p / n2 : 0;
        j = valid ? cross_n1 + (p - i * n2) : 0;
    }
    const int mi = valid ? (mrows ? mrows[i] : m_max) : 0;
    const int mj = valid ? (mrows ? mrows[j] : m_max) : 0;
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, R>(
        g, valid, mi + mj, row0,
        [&](int rr, int kk) { return A[(((rr < mi) ? i : j) * m_max + ((rr < mi) ? rr : rr - mi)) * D + kk]; },
        [&](int rr) { return b[((rr < mi) ? i : j) * m_max + ((rr < mi) ? rr : rr - mi)] + inflate; },  // b1 += abs_tol; b2 += abs_tol
        x, force_retry);
    const bool yes = (st == ST_OPT) & (x[D] > thresh);
    if (compact) {
        if (valid & (g.gl == 0)) compact[p - p_lo] = yes ? 1 : 0;
        return;
    }
    if (valid & (g.gl == 0)) {
        adj[i * n + j] = yes ? 1 : 0;
        adj[j * n + i] = yes ? 1 : 0;
    }
    // diagonal
    const long long t = (long long)blockIdx.x * RBLK + threadIdx.x;
    if (t < n) adj[t * n + t] = 1;
}

// lanes per LP for `rows` rows at R rows per lane: the smallest of 4 / 8 / 16 / 32 that holds them
#define PLP_DISPATCH_GS(R, rows, CALL)                      \
    do {                                                    \
        if ((rows) <= 4 * (R)) { constexpr int GSV = 4; return CALL; }   \
        if ((rows) <= 8 * (R)) { constexpr int GSV = 8; return CALL; }   \
        if ((rows) <= 16 * (R)) { constexpr int GSV = 16; return CALL; } \
        if constexpr ((R) < 4) {                            \
            if ((rows) <= 32 * (R)) { constexpr int GSV = 32; return CALL; } \
        }                                                   \
        if constexpr ((R) < 2) {                            \
            constexpr int GSV = 64;                         \
            return CALL;                                    \
        }                                                   \
        return 1;                                           \
    } while (0)

static int force_retry_env() {
    const char* fr = getenv("PLP_CHEBY_RETRY_ALL");
    return (fr && fr[0] == '1') ? 1 : 0;
}

// Where the two-phase run on the fast path pays: measured on MI355X against the one-row-per-lane two-phase
// kernel (100k LPs, m=16): n=3 1.35x, n=4 1.22x, n=5 1.0x, n=6 0.88x, n=8 0.7x (the extra column, the carried
// cost row and the sign bookkeeping push four rows per lane past 250 VGPRs: one wave per SIMD), n=2 0.84x.
// Two rows per lane for n = 5..8 (groups twice as wide, two waves per SIMD) was measured too: 0.84x / 0.82x / 0.66x
// of the one-row kernel at (16,5) / (16,6) / (32,8) -- bitwise the same results, so the cut-off stays at n = 4.
#ifndef PLP_P1_FAST_MAXN
#define PLP_P1_FAST_MAXN 4
#endif
template <int N>
struct P1_FAST { static constexpr bool value = (N >= 3 && N <= PLP_P1_FAST_MAXN); };
// rows per lane of the phase-1 kernel: 4 for n <= 4; 2 beyond (groups twice as wide), which keeps the extra column
// and the carried cost row within two waves per SIMD
template <int N>
struct P1Rows { static constexpr int value = N <= 4 ? 4 : 2; };

// Generic LP  min c'x  s.t.  G x <= h, x free  (solvers.py:76-106) when the origin is feasible (every
// h_i >= 0): phase 2 starts from the all-slack dictionary, which is what the two-phase kernel of
// plp_lp.hip does too in that case, so both walk the same path.  LPs that need phase 1 (or, later,
// Bland's rule) end with ST_RETRY and are redone by that kernel in a second launch.
template <int N, int GS>
__global__ __launch_bounds__(RBLK, (N <= 4 ? 3 : 1)) void lp_r_kernel(long long B, int m_max,
                                                                       const double* __restrict__ c,
                                                                       const double* __restrict__ G,
                                                                       const double* __restrict__ h,
                                                                       const int* __restrict__ mrows,
                                                                       double* __restrict__ x,
                                                                       double* __restrict__ fun,
                                                                       int* __restrict__ status,
                                                                       int* __restrict__ iters) {
    constexpr int R = RowsPerLane<N>::value;
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    const int row0 = g.gl * R;
    const long long lp = (long long)blockIdx.x * gpb + gib;
    const bool valid = lp < B;
    const int m = valid ? (mrows ? mrows[lp] : m_max) : 0;
    SimplexR<N, R, false, true> S;
    S.reset(N, m, row0);
    double cc[N];
    bool finite = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        cc[j] = valid ? c[lp * N + j] : 0.0;
        finite = finite & isfinite(cc[j]);
        S.cost[j] = cc[j];
    }
    unsigned actb = 0u;
    bool inf0 = false, neg = false;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool hr = valid & (row0 + k < m) & (m <= GS * R);
        bool zero = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double v = hr ? G[(lp * m_max + row0 + k) * N + j] : 0.0;
            S.T[k][j] = v;
            zero = zero & (v == 0.0);
            finite = finite & isfinite(v);
        }
        const double hk = hr ? h[lp * m_max + row0 + k] : 0.0;
        finite = finite & isfinite(hk);
        const bool on = hr & !zero;
        S.beta[k] = on ? hk : 0.0;
        actb |= on ? (1u << k) : 0u;
        inf0 = inf0 | (hr & zero & (hk < -TOL_FEAS));  // 0 <= h_i < 0
        neg = neg | (on & (hk < 0.0));
    }
    S.ract = actb;
    const bool infeasible0 = grp_ballot(inf0, g) != 0;
    const bool bad = (grp_ballot(!finite, g) != 0) | (m > GS * R);
    const bool need_p1 = grp_ballot(neg, g) != 0;
    S.mode = M_P2;
    if (!valid | bad) { S.mode = M_DONE; S.status = ST_NUM; }
    else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
    else if (need_p1) { S.mode = M_DONE; S.status = P1_FAST<N>::value ? ST_RETRY_P1 : ST_RETRY; }
    S.template run_fast<GS, false>(g);
    const bool ok = S.status == ST_OPT;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    double f = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        bool found;
        const double mine = S.x_of(j, found);
        const uint64_t ob = grp_ballot(found, g);
        const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
        const double xj = ob ? v : 0.0;
        f = fma(cc[j], xj, f);
        if (valid & (g.gl == 0)) x[lp * N + j] = ok ? xj : qnan;
    }
    if (valid & (g.gl == 0)) {
        fun[lp] = ok ? f : qnan;
        status[lp] = S.status;
        if (iters) iters[lp] = S.iters;
    }
}

// The LPs lp_r_kernel marked ST_RETRY_P1 (some h_i < 0): two-phase run on the fast path with the artificial
// variable in an extra column and the real objective carried along; Bland cases leave with ST_RETRY for the
// two-phase kernel of plp_lp.hip (third launch).
template <int N, int GS, int R>
__global__ __launch_bounds__(RBLK, (R == 2 ? 2 : (N <= 3 ? 2 : 1))) void lp_p1_r_kernel(long long B, int m_max,
                                                                          const double* __restrict__ c,
                                                                          const double* __restrict__ G,
                                                                          const double* __restrict__ h,
                                                                          const int* __restrict__ mrows,
                                                                          double* __restrict__ x,
                                                                          double* __restrict__ fun,
                                                                          int* __restrict__ status,
                                                                          int* __restrict__ iters) {
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    const int row0 = g.gl * R;
    const long long lp = (long long)blockIdx.x * gpb + gib;
    const bool valid = (lp < B) && status[lp] == ST_RETRY_P1;
    if (!__syncthreads_or(valid)) return;  // no LP of this workgroup needs phase 1
    const int m = valid ? (mrows ? mrows[lp] : m_max) : 0;
    SimplexR<N + 1, R, false, true, true> S;
    S.reset(N, m, row0);
    S.cv[N] = ID_TR;  // the artificial variable t sits in the last column (not free: reset leaves its cfree bit clear)
    double cc[N], qi[R];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        cc[j] = valid ? c[lp * N + j] : 0.0;
        S.cost2[j] = cc[j];
    }
    S.cost[N] = 1.0;  // phase 1: minimise t
    unsigned actb = 0u;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool hr = valid & (row0 + k < m);
        bool zero = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double v = hr ? G[(lp * m_max + row0 + k) * N + j] : 0.0;
            S.T[k][j] = v;
            zero = zero & (v == 0.0);
        }
        const double hk = hr ? h[lp * m_max + row0 + k] : 0.0;
        const bool on = hr & !zero;
        S.T[k][N] = on ? -1.0 : 0.0;
        S.beta[k] = on ? hk : 0.0;
        qi[k] = S.beta[k];
        actb |= on ? (1u << k) : 0u;
    }
    S.ract = actb;
    S.mode = valid ? M_P2 : M_DONE;
    S.template run_two_phase<GS>(g, qi, actb, valid);
    const bool ok = S.status == ST_OPT;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    double f = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        bool found;
        const double mine = S.x_of(j, found);
        const uint64_t ob = grp_ballot(found, g);
        const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
        const double xj = ob ? v : 0.0;
        f = fma(cc[j], xj, f);
        if (valid & (g.gl == 0)) x[lp * N + j] = ok ? xj : qnan;
    }
    if (valid & (g.gl == 0)) {
        fun[lp] = ok ? f : qnan;
        status[lp] = S.status;
        if (iters) iters[lp] = S.iters;
    }
}

}  // namespace plp
