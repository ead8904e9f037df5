// plp_reduce_lane.hip -- reduce_lane_kernel<D>: the fused reduce() (polytope/polytope.py:1053-1163) for polytopes of up
// to 16 rows in d <= 3 -- BASELINE configs[1], the bench shape -- with the box LPs (F3, :1118-1134) and the redundancy
// LPs the presolve leaves (F2, :1142-1160) solved ONE LP PER LANE (plp_lane_lp.hpp) instead of one LP per lane group.
//
// Why.  In reduce_r_mix_kernel<3> (plp_reduce_r_impl.hpp) the six box LPs of a polytope run one after the other on its
// lane group of four, in lock-step with the 15 other groups of the wavefront: 24.9 dictionary pivots of ~280 VALU
// instructions per tile, 45 % of the kernel's instruction stream, two thirds of it selects / cross-lane moves for a
// dynamic pivot position; the pooled redundancy LPs another 24 %.  An LP of this shape needs no dictionary: from the
// Chebyshev centre the optimum is reached by walking facet -> edge -> vertex (-> neighbouring vertices), every step one
// ratio test over the 16 rows.  With the rows in LDS a LANE can do that alone, so a wavefront advances 64 LPs per
// instruction instead of 16 and no value crosses lanes inside an LP:
//     F3: the 96 box LPs of a tile = two rounds (4 LPs per polytope, then 2),
//     F2: the ~34 LPs the presolve leaves in a tile = one round (lane t takes the t-th LP of the tile's list).
// F1 (the Chebyshev LP, d + 1 columns), the dedupe, the prefilter arithmetic and the presolve are those of the lane-group
// kernel, instruction for instruction: r, xc and every verdict that does not come out of an F3 / F2 LP are bit for bit
// the same; the LP optima agree to rounding (tests/test_lane_lp_host.py: 1e-12 against the oracle's simplex), so keep
// masks, flags and LP counts are the oracle's (tests/test_gpu_parity.py, bench.py's full-batch check).
//
// LDS layout: POLYTOPE-INTERLEAVED -- element (row i, column k) of the tile's polytope p at sA[(i * D + k) * 16 + p],
// b and the per-row scalar likewise at [i * 16 + p].  Sixteen lanes that read the same element of sixteen different
// polytopes (every LDS read of this kernel has that shape) touch 16 consecutive doubles = all 32 banks once; in the
// polytope-major layout of the lane-group kernels they are 384 B apart = one bank, a 16-way conflict.
//
// An LP the lane engine hands back (ST_RETRY: a run of degenerate steps, dependent active rows) sends its polytope to the
// general engine (plp_reduce_general.hpp, Bland's rule) at the end of the SAME tile: a step is one launch.
#include <stdlib.h>

#include <type_traits>

#include "plp_lane_lp.hpp"
#include "plp_reduce_general.hpp"
#include "plp_reduce_r_impl.hpp"

namespace plp {

#ifndef PLP_LANE_CH
#define PLP_LANE_CH 4   // rows per trip of the ratio loop (the fewest rows a lane tests: 16 / 4 in quad mode)
#endif
#ifndef PLP_LANE_DEDUPE_SCREEN
#define PLP_LANE_DEDUPE_SCREEN 1
#endif

constexpr int LN_ROWS = 16;   // row slots per polytope (ROWS = 32: polytopes of 17..32 rows, e.g. the stack of Polytope.intersect, ref :268-275)
// GS lanes per polytope (4 / 8 / 16), R = 16 / GS rows per lane in the lane-group stages, NG = 64 / GS polytopes per tile
// (= per wavefront).  GS = 4 is the throughput form; 8 and 16 put fewer polytopes on a wavefront and finish a tile in
// about 0.6 / 0.4 of the time: the latency forms for batches that cannot fill the chip, and the tail of a large launch.

static inline size_t reduce_lane_smem_bytes(int D, int GS, int ROWS = LN_ROWS) { return (size_t)(64 / GS) * ROWS * (D + 2) * 8; }

// The polytopes of a tile that the fast path handed back (bit GS p of `rb64`: polytope p), redone by the general engine
// (plp_reduce_general.hpp: one dictionary row per lane, 16 lanes per polytope, Bland's rule in the simplex).
template <int D, int GS, int ROWS>
__device__ __noinline__ void reduce_lane_redo(unsigned char* smem_raw, const long long tile, const int ntile, const uint64_t rb64,
                                              int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
                                              const int* __restrict__ mrows, double abs_tol,
                                              unsigned long long* __restrict__ keep_out, int* __restrict__ flags_out,
                                              double* __restrict__ r_out, double* __restrict__ xc_out,
                                              int* __restrict__ nlp_out) {
    constexpr int PER = 64 / ROWS;   // polytopes per pass of the general engine (ROWS lanes each)
    for (int sub = 0; sub < (64 / GS) / PER; ++sub) {
        unsigned some = 0u;   // bit q: polytope PER sub + q
#pragma unroll
        for (int q = 0; q < PER; ++q) some |= (unsigned)((rb64 >> (GS * (PER * sub + q))) & 1ull) << q;
        if (some == 0u) continue;   // wave-uniform
        const int q = (threadIdx.x & 63) / ROWS;
        const bool mine = ((some >> q) & 1u) != 0u;
        const int left = ntile - PER * sub;
        reduce_general_tile<D, RBLOCK>(smem_raw, tile + PER * sub, left < PER ? left : PER, mine, m_max, ROWS, Ag, bg, mrows, abs_tol,
                                       keep_out, flags_out, r_out, xc_out, nlp_out);
    }
}

// BBOX: the stand-alone bounding boxes of plp_bbox_batch (polytope.py:1314-1411) instead of reduce(): the same load, F1 and
// box LPs on lanes; no dedupe, no prefilter, no redundancy LPs.  r_out / xc_out then are lb / ub [B][D], flags_out the status
// (0: lb / ub hold the box, -inf / +inf where an LP is unbounded (:1376, :1398); 1: not settled here -- empty, flat or
// unbounded-ball polytopes, an LP handed back -- the caller solves the generic LPs), the contract of bbox_r_kernel.
template <int D, int GS, int ROWS = LN_ROWS, bool BBOX = false>
__device__ __forceinline__ void reduce_lane_tile(
    const long long tile, long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out, int* __restrict__ nlp_out,
    unsigned long long* __restrict__ ctr) {
    static_assert(D >= 1 && D <= 4, "the lane engine walks in R^3 (lower dimensions are embedded) or R^4");
    static_assert(RBLOCK == 64, "one wavefront per workgroup");
    static_assert(GS == 4 || GS == 8 || GS == 16, "lanes per polytope");
    static_assert((ROWS == 16 || ROWS == 32) && ROWS / GS >= 1 && ROWS / GS <= 4, "row slots per polytope, at most four per lane");
    constexpr int rows = ROWS, R = rows / GS, NG = 64 / GS;
    constexpr unsigned RMASK = (1u << R) - 1u;
    constexpr int LS = NG;   // stride between consecutive elements of one polytope
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const Grp g(GS);
    const int lane = g.lane;
    const int gib = lane / GS;   // my polytope inside the tile
    const int row0 = g.gl * R;   // my first row
    double* sA = reinterpret_cast<double*>(smem_raw);   // [rows * D][NG]
    double* sb = sA + (size_t)rows * D * NG;             // [rows][NG]
    double* san = sb + (size_t)rows * NG;                // [rows][NG]: 1 / ||a_i||, later beta_i = max(b_i - a_i.xc, 0)
    // (one base pointer per lane, everything else a compile-time offset from it: registers are what this kernel runs out of)
    double* myA = sA + gib;
    constexpr int OFF_B = rows * D * NG, OFF_N = OFF_B + rows * NG;   // sb - sA, san - sA in doubles
    double* const myb = myA + OFF_B;
    double* const myan = myA + OFF_N;
#define LA(i, kk) myA[((i) * D + (kk)) * LS]
#define LB(i) myA[OFF_B + (i) * LS]
#define LN(i) myA[OFF_N + (i) * LS]
    // my own rows row0 + k: two bases (the A rows have a stride of D elements, b / beta of one) and compile-time offsets
    double* const rA = myA + row0 * D * LS;
    double* const rS = myA + row0 * LS;
#define OA(k, kk) rA[((k) * D + (kk)) * LS]
#define OB(k) rS[OFF_B + (k) * LS]
#define ON(k) rS[OFF_N + (k) * LS]
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    const int ntile = (B - tile) < NG ? (int)(B - tile) : NG;
    __syncthreads();
    {
        // lane (q, p): polytope p, elements q, q + 4, ... of its record -- the wavefront reads 16 records x 32 B per
        // instruction (whole 32 B sectors; the rest of each 128 B line is used by the next three iterations) and writes
        // 16 consecutive doubles x 4 rows of the interleaved tile (no bank conflict)
        const int p = lane % NG, q = lane / NG;   // (q < GS)
        const int rowsz = m_max * D;
        const bool pv = p < ntile;
        const double* src = Ag + (tile + (pv ? p : 0)) * rowsz;
        // (every load unconditional, from a clamped index, and all of them issued before the first LDS store: a load under
        // a condition becomes a branch of its own, and sixteen of those in a row are sixteen HBM round trips)
        double va[rows * D / GS], vb[rows / GS];
        const double* srcb = bg + (tile + (pv ? p : 0)) * m_max;
        if (m_max == rows) {   // (wave-uniform; full records: one address, immediate offsets)
#pragma unroll
            for (int it = 0; it < rows * D / GS; ++it) va[it] = src[q + GS * it];
#pragma unroll
            for (int it = 0; it < rows / GS; ++it) vb[it] = srcb[q + GS * it];
        } else {
#pragma unroll
            for (int it = 0; it < rows * D / GS; ++it) {
                const int rem = q + GS * it;
                va[it] = src[rem < rowsz ? rem : 0];
            }
#pragma unroll
            for (int it = 0; it < rows / GS; ++it) {
                const int row = q + GS * it;
                vb[it] = srcb[row < m_max ? row : 0];
            }
        }
#pragma unroll
        for (int it = 0; it < rows * D / GS; ++it) {
            const int rem = q + GS * it;
            sA[rem * NG + p] = (pv & (rem < rowsz)) ? va[it] : 0.0;
        }
#pragma unroll
        for (int it = 0; it < rows / GS; ++it) {
            const int row = q + GS * it;
            sb[row * NG + p] = (pv & (row < m_max)) ? vb[it] : 0.0;
        }
    }
    __syncthreads();
    // (outputs are addressed as (array + tile)[gib]: the tile offset is wave-uniform and lives in scalar registers, the
    // lane keeps its 32-bit gib instead of a 64-bit polytope index)
    const bool valid = gib < ntile;
    const int m = valid ? (mrows ? (mrows + tile)[gib] : m_max) : 0;
    double xc[D];
    double rr = 0.0;
    bool ball, fulldim, f1open = false;
    uint64_t live = 0ull;
    unsigned has = 0u;
    bool retry = force_retry != 0;
    // ---------------------------------------------------------------- F1: Chebyshev ball (as reduce_r_tile<D,4,4>)
    {
        SimplexR<D + 1, R, false, true> S;
        double qi[R];
        S.reset(D + 1, m, row0);
        unsigned actb = 0u;
        bool inf0 = false, finite = true;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool h = valid & (row0 + k < m) & (m <= rows);
            has |= h ? (1u << k) : 0u;
            double nrm2 = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) {
                const double v = h ? OA(k, kk) : 0.0;
                S.T[k][kk] = v;
                nrm2 = nrm2 + v * v;
                finite = finite & isfinite(v);
            }
            const double bk = h ? OB(k) : 0.0;
            finite = finite & isfinite(bk);
            const double nrm = sqrt(nrm2);
            ON(k) = 1.0 / nrm;
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
        const bool bad = (grp_ballot(!finite, g) != 0) | (m > rows);
        S.cost[D] = -1.0;
        S.mode = M_P2;
        if (!valid | bad) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
#ifndef PLP_LANE_DBG_NOF1
        S.template run_fast<GS, true>(g, qi, actb);
#endif
        retry = retry | (valid & (S.status == ST_RETRY));
        const bool ok = S.status == ST_OPT;
        f1open = valid & !ok & (S.status != ST_INFEAS);   // (RF_F1OPEN, plp_common.hpp)
#pragma unroll
        for (int j = 0; j <= D; ++j) {
            bool found;
            const double mine = S.x_of(j, found);
            const uint64_t ob = grp_ballot(found, g);
            const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
            const double xj = ob ? v : 0.0;
            if (j < D) xc[j < D ? j : 0] = xj; else rr = xj;
        }
        ball = ok & (rr >= 0.0);  // cheby_ball: status 0 and r >= 0 (:1289-1293)
        fulldim = ball & (rr > abs_tol);
        if constexpr (BBOX) fulldim = ball & (rr >= 1e-6);   // (BBOX_MIN_R of bbox_r_kernel: a centre worth starting from)
    }
    if (!BBOX && (valid & (g.gl == 0))) {
        (r_out + tile)[gib] = ball ? rr : 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) (xc_out + tile * D)[gib * D + k] = ball ? xc[k] : qnan;
    }
    __syncthreads();  // 1/||a|| of every row is in LDS
#ifdef PLP_LANE_DBG_F1ONLY
    if (valid & (g.gl == 0)) { (keep_out + tile)[gib] = 0; (flags_out + tile)[gib] = 0; (nlp_out + tile)[gib] = 1; }
    return;
#endif
    // ---------------------------------------------------------------- dedupe (:1094-1110): every pair of rows once
    if constexpr (BBOX) {
#pragma unroll
        for (int k = 0; k < R; ++k) live |= spread_rows<R, GS>(grp_ballot(((has >> k) & 1u) != 0u, g)) << k;
    } else {
        unsigned remmask = 0u;
#if PLP_LANE_DEDUPE_SCREEN
        const double scr_thr = sqrt(2.0 * abs_tol) * 1.0001 + 1e-12;   // (wave-uniform)
#endif
        double ni[R][D], bin_[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double an_i = ON(k);
#pragma unroll
            for (int kk = 0; kk < D; ++kk) ni[k][kk] = OA(k, kk) * an_i;
            bin_[k] = OB(k) * an_i;
        }
#pragma unroll 2
        for (int t = 1; t <= rows / 2; ++t) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int i = row0 + k;
                const int j = (i + t) & (rows - 1);
                const double an_j = LN(j);
#if PLP_LANE_DEDUPE_SCREEN
                // unit rows with dot > 1 - tol differ by less than sqrt(2 tol) in every component: when no pair of the
                // wavefront passes that test on the first component (random rows: never) the pair is skipped
                if (!__any(fabs(ni[k][0] - LA(j, 0) * an_j) < scr_thr)) continue;
#endif
                double dot = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) dot = dot + ni[k][kk] * (LA(j, kk) * an_j);
                const double bjn = LB(j) * an_j;
                const bool par = valid & (m <= rows) & (i < m) & (j < m) & (dot > 1.0 - abs_tol);
                // the reference's rule for the pair (lo, hi), lo < hi (:1104-1109): b_lo < b_hi removes hi, else lo
                const bool i_lo = i < j;
                const double blo = i_lo ? bin_[k] : bjn, bhi = i_lo ? bjn : bin_[k];
                const int lo = i_lo ? i : j, hi = i_lo ? j : i;
                const int gone = (blo < bhi) ? hi : lo;
                remmask |= par ? (1u << gone) : 0u;
            }
        }
#pragma unroll
        for (int o = 1; o < GS; o <<= 1) remmask |= (unsigned)__shfl_xor((int)remmask, o, 64);
        const unsigned removed = (remmask >> row0) & RMASK;
#pragma unroll
        for (int k = 0; k < R; ++k)
            live |= spread_rows<R, GS>(grp_ballot((((has & ~removed) >> k) & 1u) != 0u, g)) << k;
    }
    // The LPs below live in centre-relative coordinates: a_i.x' <= beta_i, beta_i = max(b_i - a_i.xc, 0), which replaces
    // 1/||a_i|| in LDS -- every LP of the polytope reads it as it is (the lane-group kernels keep s_i = a_i.xc and form
    // b_i - s_i in every LP set-up: the same number)
    // A centre that violates a row (centre_off, plp_common.hpp) is no centre.  An interior centre leaves every b_i - a_i.xc
    // positive, so the test proper runs only in a wavefront that saw a negative one, and forms everything it needs again
    // from the rows (nothing is kept for it across the branch: four doubles held for it were an 8-byte spill store per lane in the hot
    // path -- WRITE_SIZE 6.2 -> 10.2 MB per launch).  The hot path pays one compare per row (+0.4 % on the bench step; the full
    // test on every row: +1.0 %, same-box A/B).
    bool neg = false;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        double sk = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) sk = fma(((has >> k) & 1u) ? OA(k, kk) : 0.0, ball ? xc[kk] : 0.0, sk);
        const double raw = (((has >> k) & 1u) ? OB(k) : 0.0) - sk;
        neg = neg | (raw < 0.0);
        ON(k) = fmax(raw, 0.0);
    }
    if (__any(neg & ball)) {
        bool off = false;
        const double xs = centre_scale<D>(xc);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool hk = ((has >> k) & 1u) != 0u;
            double sk = 0.0, nrm2 = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) {
                const double v = hk ? OA(k, kk) : 0.0;
                sk = fma(v, ball ? xc[kk] : 0.0, sk);
                nrm2 = nrm2 + v * v;
            }
            const double bk = hk ? OB(k) : 0.0;
            off = off | (hk & centre_off(bk - sk, 1.0 / sqrt(nrm2), bk, xs));   // (1 / |a|: the value F1's set-up stored)
        }
        if (ball & (grp_ballot(off, g) != 0)) {   // F1 "optimal" outside the polytope: nothing below may start from it
            ball = false; fulldim = false; f1open = true;
            if (!BBOX && (valid & (g.gl == 0))) {
                (r_out + tile)[gib] = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) (xc_out + tile * D)[gib * D + k] = qnan;
            }
        }
    }
    // rows that dropped out (never present, or removed by the dedupe / the prefilter) are zeroed -- A, b and s -- by their
    // owner lane: a zero row never stops a ray and passes every presolve test
    auto zero_dead = [&](unsigned alive) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (!((alive >> k) & 1u)) {
#pragma unroll
                for (int kk = 0; kk < D; ++kk) OA(k, kk) = 0.0;
                OB(k) = 0.0;
                ON(k) = 0.0;
            }
        }
    };
    zero_dead(((unsigned)(live >> row0) & RMASK));
    int flags = fulldim ? 0 : (RF_EMPTY | (f1open ? RF_F1OPEN : 0));
    int nlp = 1;
    uint64_t keep = 0ull;
    int stage = 0;  // 0 done, 1 needs the box, 2 needs the redundancy LPs
    if (fulldim) {
        const int neq = __popcll(live);
        if (neq <= D + 1) { flags = RF_EARLY; keep = live; }
        else stage = (neq > 3 * D) ? 1 : 2;
        if constexpr (BBOX) stage = 1;   // every polytope with a usable centre gets its 2 d LPs
    }
    __syncthreads();   // the lanes of OTHER groups read these rows from here on
    auto any_lane = [](bool p) { return __any(p) != 0; };
    // One LP per lane -- or per PAIR / QUAD of neighbouring lanes when a round has no more than 32 / 16 LPs: the lanes of an
    // LP carry the same walk (same point, same active rows, same direction: computed redundantly), each runs the ratio
    // test over its half / quarter of the row slots, and the candidates meet through one / two DPP exchanges (smaller
    // ratio, on ties the lower row: the row a single lane would have found first).  `nparts` is wave-uniform.
    // pA / pbeta: the polytope's rows / right-hand sides in the interleaved tile; RELAX: row krv's right-hand side + 0.1
    // (:1149) -- an add of 0.1 or 0, not a select between a constant and the LDS value (that becomes a branch around the load).
    using LpState = std::conditional_t<D == 4, lane::Lp4, lane::Lp3>;   // d <= 3: the walk in R^3 (lower dimensions embedded); d = 4: in R^4
    auto lane_solve = [&](LpState& S, const double (&cv)[4], const bool go_, const double* pA, const double* pbeta, auto relax_tag,
                          int krv, const int nparts, const int part) {
        constexpr bool RELAX = decltype(relax_tag)::value;
        const int cnt = rows / nparts;   // wave-uniform
        const int i0 = part * cnt;
        auto row_of = [&](const double* base, double& a0, double& a1, double& a2, double& a3) {
            a0 = base[0];
            a1 = D > 1 ? base[(D > 1 ? 1 : 0) * LS] : 0.0;
            a2 = D > 2 ? base[(D > 2 ? 2 : 0) * LS] : 0.0;
            a3 = D > 3 ? base[(D > 3 ? 3 : 0) * LS] : 0.0;
        };
        // the ratio test of one pass: my share of the rows, then the exchange with the other lanes of the LP
        auto ratio = [&](auto x0_tag, const double (&dv)[4], const double (&xv)[4], double tolp, double& bs, double& bd, int& bi) {
            constexpr bool X0 = decltype(x0_tag)::value;   // the walk's first pass: x' = 0, every slack is its beta
            if constexpr (RELAX) asm volatile("" : "+v"(krv));   // (keeps the per-row compares inside the walk)
            const double* ra = pA + i0 * D * LS;
            const double* rb_ = pbeta + i0 * LS;
            int ic = i0;
            for (int ch = 0; ch < cnt; ch += PLP_LANE_CH) {
#pragma unroll
                for (int r = 0; r < PLP_LANE_CH; ++r) {
                    double a0, a1, a2, a3;
                    row_of(ra + (r * D) * LS, a0, a1, a2, a3);
                    double beta = rb_[r * LS];
                    if constexpr (RELAX) beta = beta + ((ic + r == krv) ? 0.1 : 0.0);
                    if constexpr (D == 4)
                        lane::ratio_row4(a0, a1, a2, a3, beta, ic + r, dv[0], dv[1], dv[2], dv[3], xv[0], xv[1], xv[2], xv[3], tolp,
                                         bs, bd, bi);
                    else if constexpr (X0)
                        lane::ratio_row0(a0, a1, a2, beta, ic + r, dv[0], dv[1], dv[2], tolp, bs, bd, bi);
                    else
                        lane::ratio_row(a0, a1, a2, beta, ic + r, dv[0], dv[1], dv[2], xv[0], xv[1], xv[2], tolp, bs, bd, bi);
                }
                ra += PLP_LANE_CH * D * LS;
                rb_ += PLP_LANE_CH * LS;
                ic += PLP_LANE_CH;
            }
            auto meet = [&](auto ctrl) {
                constexpr int CTRL = decltype(ctrl)::value;
                const double ps = dpp_d<CTRL>(bs), pd = dpp_d<CTRL>(bd);
                const int pi = dpp_i<CTRL>(bi);
                const double l = ps * bd, r_ = bs * pd;
                const bool theirs = (pd > 0.0) & (!(bd > 0.0) | (l < r_) | ((l == r_) & (pi < bi)));
                bs = theirs ? ps : bs;
                bd = theirs ? pd : bd;
                bi = theirs ? pi : bi;
            };
            if (nparts > 1) meet(std::integral_constant<int, PLP_DPP_XOR1>{});
            if (nparts > 2) meet(std::integral_constant<int, PLP_DPP_XOR2>{});
        };
        if constexpr (D == 4) {
            lane::walk4(
                S, cv[0], cv[1], cv[2], cv[3], go_,
                [&](int i, double& a0, double& a1, double& a2, double& a3) { row_of(pA + (i * D) * LS, a0, a1, a2, a3); },
                [&](double d0, double d1, double d2, double d3, double x0, double x1, double x2, double x3, double tolp, double& bs,
                    double& bd, int& bi) {
                    const double dv[4] = {d0, d1, d2, d3}, xv[4] = {x0, x1, x2, x3};
                    ratio(std::false_type{}, dv, xv, tolp, bs, bd, bi);
                },
                any_lane);
        } else {
            lane::walk3(
                S, cv[0], cv[1], cv[2], go_,
                [&](int i, double& a0, double& a1, double& a2) {
                    double a3;
                    row_of(pA + (i * D) * LS, a0, a1, a2, a3);
                },
                [&](double d0, double d1, double d2, double x0, double x1, double x2, double tolp, double& bs, double& bd, int& bi) {
                    const double dv[4] = {d0, d1, d2, 0.0}, xv[4] = {x0, x1, x2, 0.0};
                    ratio(std::false_type{}, dv, xv, tolp, bs, bd, bi);
                },
                [&](double d0, double d1, double d2, double tolp, double& bs, double& bd, int& bi) {
                    const double dv[4] = {d0, d1, d2, 0.0}, xv[4] = {0.0, 0.0, 0.0, 0.0};
                    ratio(std::true_type{}, dv, xv, tolp, bs, bd, bi);
                },
                any_lane);
        }
    };
    auto x_of_lp = [](const LpState& S, int k) {
        double v = k == 0 ? S.x0 : (k == 1 ? S.x1 : S.x2);
        if constexpr (D == 4) v = k == 3 ? S.x3 : v;
        return v;
    };
    // ---------------------------------------------------------------- F3: bounding box (:1367-1409)
#ifdef PLP_LANE_DBG_NOF3
    if (false) {
#else
    if (__any(stage == 1)) {
#endif
        const bool go = stage == 1;
        const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
        bool lpfail = false;
        // LP `it`: lower_0, upper_0, lower_1, upper_1, ...  A round takes the next GS of the polytope's 2 D LPs, one per lane
        // -- or, when no more than GS / 2 are left for it, one per PAIR of lanes (GS = 16: always pairs):
        //   d = 3:  GS = 4: LPs 0..3, then 4, 5 in pairs;  GS = 8: one round;  GS = 16: one round in pairs
        //   d = 4:  GS = 4: LPs 0..3, then 4..7;           GS = 8: one round;  GS = 16: one round in pairs
        constexpr int NLP = 2 * D;
        constexpr int NROUND = GS == 16 ? 1 : (NLP + GS - 1) / GS;
        auto f3_pair = [](int rd) { return GS == 16 || 2 * (NLP - rd * GS < GS ? NLP - rd * GS : GS) <= GS; };
        auto f3_round_of = [](int it) { return GS == 16 ? 0 : it / GS; };
        auto f3_lane_of = [&](int it) { const int rd = f3_round_of(it); return f3_pair(rd) ? 2 * (it - rd * GS) : it - rd * GS; };
        double val[NROUND];
#pragma unroll
        for (int q = 0; q < NROUND; ++q) val[q] = 0.0;
#pragma unroll 1
        for (int rd = 0; rd < NROUND; ++rd) {
            const bool pair = f3_pair(rd);
            const int nparts = pair ? 2 : 1;
            const int it = rd * GS + (pair ? (g.gl >> 1) : g.gl);
            const int part = pair ? (g.gl & 1) : 0;
            const bool mine = go & (it < 2 * D);
            const int kx = it >> 1;
            const bool up = it & 1;
            const double cs = up ? -1.0 : 1.0;
            LpState S;
            const double cv[4] = {kx == 0 ? cs : 0.0, kx == 1 ? cs : 0.0, kx == 2 ? cs : 0.0, kx == 3 ? cs : 0.0};
            lane_solve(S, cv, mine, myA, myan, std::false_type{}, -1, nparts, part);
            double xck = 0.0;
#pragma unroll
            for (int kk = 0; kk < D; ++kk) xck = (kk == kx) ? xc[kk] : xck;
            const double xk = x_of_lp(S, kx);
            double v;
            if (S.status == ST_OPT) v = xck + xk;
            else if (S.status == ST_UNBND) v = up ? pinf : -pinf;
            else { v = qnan; lpfail = lpfail | (mine & (S.status != ST_RETRY)); }
            retry = retry | (mine & (S.status == ST_RETRY));
            if constexpr (BBOX) {
                // the point the walk ended on, for the verifier (plp_verify.hip: it reads a basis off it); BBOX: keep_out is that buffer
                double* xfin = reinterpret_cast<double*>(keep_out);
                if (xfin && (valid & mine & (part == 0))) {
#pragma unroll
                    for (int kk = 0; kk < D; ++kk) xfin[((size_t)(tile + gib) * 2 * D + it) * D + kk] = xc[kk] + x_of_lp(S, kk);
                }
            }
#pragma unroll
            for (int q = 0; q < NROUND; ++q) val[q] = (q == rd) ? v : val[q];
        }
        // an LP handed back or failed anywhere in my group concerns the polytope
        lpfail = grp_ballot(lpfail, g) != 0;
        retry = retry | (grp_ballot(retry, g) != 0);
        if constexpr (BBOX) {
            // lb / ub from the lanes that hold them; a polytope with an LP handed back or failed is left to the caller
#pragma unroll
            for (int it2 = 0; it2 < NLP; ++it2) {
                const double v = bcast(val[f3_round_of(it2)], g.gbase + f3_lane_of(it2));
                if (valid & (g.gl == 0)) ((it2 & 1) ? xc_out + tile * D : r_out + tile * D)[gib * D + (it2 >> 1)] = go ? v : qnan;
            }
            if (valid & (g.gl == 0)) (flags_out + tile)[gib] = (go & !retry & !lpfail) ? 0 : 1;
            return;
        }
        // prefilter sums, accumulated in k order (:1131-1134); LP `it` sits in lane f3_lane_of(it) of round f3_round_of(it)
        double s1[R], s2[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { s1[k] = 0.0; s2[k] = 0.0; }
#pragma unroll
        for (int kx = 0; kx < D; ++kx) {
            const int itl = 2 * kx, ith = 2 * kx + 1;
            const double lo = bcast(val[f3_round_of(itl)], g.gbase + f3_lane_of(itl));
            const double hi = bcast(val[f3_round_of(ith)], g.gbase + f3_lane_of(ith));
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double aik = OA(k, kx);
                const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                s1[k] = s1[k] + pa * (hi - lo);
                s2[k] = s2[k] + aik * lo;
            }
        }
        uint64_t outb = 0ull;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool out = go & (((lloc >> k) & 1u) != 0u) & ((s1[k] - (OB(k) - s2[k])) < -1e-4);
            outb |= spread_rows<R, GS>(grp_ballot(out, g)) << k;
        }
        __syncthreads();   // every lane's LPs have read the rows: the owners may zero the ones the prefilter removes
        if (go) {
            live = live & ~outb;
            zero_dead(((unsigned)(live >> row0) & RMASK));
            nlp += 2 * D;
            if (lpfail) flags |= RF_LPFAIL;
            if (__popcll(live) <= D + 1) { flags |= RF_EARLY; keep = live; stage = 0; }
            else stage = 2;
        }
        __syncthreads();
    }
    if constexpr (BBOX) {   // (no polytope of the tile had a usable centre)
        if (valid & (g.gl == 0)) {
            (flags_out + tile)[gib] = 1;
#pragma unroll
            for (int k = 0; k < D; ++k) { (r_out + tile * D)[gib * D + k] = qnan; (xc_out + tile * D)[gib * D + k] = qnan; }
        }
        return;
    }
    // ---------------------------------------------------------------- F2: redundancy LPs (:1142-1160)
#ifdef PLP_LANE_DBG_NOF2
    if (false) {
#else
    if (__any(stage == 2)) {
#endif
        const unsigned lloc = ((unsigned)(live >> row0) & RMASK);
        uint64_t todo = (stage == 2) ? live : 0ull;
        if (stage == 2) nlp += __popcll(live);
#ifndef PLP_LANE_DBG_NOPRE
        {   // rows the ray presolve settles as "keep" need no LP (their h[k] round trip is applied in LDS by the owner)
            const unsigned okb = f2_presolve<D, R, LS, true>(myA, myb, myan, row0, m_max, (stage == 2) ? lloc : 0u, abs_tol);
            uint64_t cert = 0ull;
#pragma unroll
            for (int k = 0; k < R; ++k) cert |= spread_rows<R, GS>(grp_ballot(((okb >> k) & 1u) != 0u, g)) << k;
            keep |= cert;
            todo &= ~cert;
            ctr_add(ctr, -__popcll(cert), g.gl == 0);
        }
#endif
        __syncthreads();
        // The LPs the presolve left, of all polytopes of the tile, form ONE list (polytope order, then row order); a round
        // takes the next 64 / 32 / 16 of them on one / two / four lanes each (whatever fills the wavefront).  Row k's own
        // right-hand side is relaxed by 0.1 (:1149).  (The reference's in-place round trip leaves rows that had their turn
        // before k at (b + 0.1) - 0.1, an ulp of b away; an LP optimum moves by no more, ten orders below the tolerance its
        // verdict is read with -- the rows are taken as they stand in LDS.)
        const unsigned todo32 = (unsigned)todo;                       // unsettled live rows of MY polytope
        const int n_g = __popc(todo32);
        int total = 0, nmax = 0;
        for (int p = 0; p < NG; ++p) {
            const int np = __builtin_amdgcn_readlane(n_g, p * GS);
            total += np;
            nmax = np > nmax ? np : nmax;
        }
        int off_g = 0;   // position of my polytope's first LP in the list
        for (int p = 0, run = 0; p < NG; ++p) {
            off_g = (gib == p) ? run : off_g;
            run += __builtin_amdgcn_readlane(n_g, p * GS);
        }
        unsigned retry_polys = 0u;   // bit p: an LP of polytope p was handed back (wave-uniform)
        int rb = 0;
        while (rb < total) {
            const int nrem = total - rb;
            const int nparts = nrem <= 16 ? 4 : (nrem <= 32 ? 2 : 1);
            const int sh = nparts == 4 ? 2 : (nparts == 2 ? 1 : 0);
            const int per = 64 >> sh;
            // list position t -> (polytope, row)
            const int t = rb + (lane >> sh);
            const int part = lane & (nparts - 1);
            int tp = 0, toff = 0, run = 0;
            unsigned ttd = 0u;
            for (int p = 0; p < NG; ++p) {
                const int np = __builtin_amdgcn_readlane(n_g, p * GS);
                const unsigned tdp = (unsigned)__builtin_amdgcn_readlane((int)todo32, p * GS);
                const bool at = t >= run;
                tp = at ? p : tp;
                ttd = at ? tdp : ttd;
                toff = at ? run : toff;
                run += np;
            }
            const bool mine = t < total;
            {
                const int rank = t - toff;
                for (int i = 0; i < nmax; ++i) ttd = (i < rank) ? (ttd & (ttd - 1u)) : ttd;
            }
            const int kr = mine ? (__ffs((int)ttd) - 1) : 0;
            const double* pA = sA + tp;
            const double* pan = pA + OFF_N;
            double c[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < D; ++kk) c[kk] = -pA[(kr * D + kk) * LS];   // f = -A[k,:]  (:1145)
            LpState S;
            lane_solve(S, c, mine, pA, pan, std::true_type{}, kr, nparts, part);
            // objective - h[k] (:1156):  -fun - hk = (a_k.xc + a_k.x') - hk = a_k.x' - (hk - a_k.xc),  hk = (b_k + 0.1) - 0.1
            // after its round trip (:1149-1151):  hk - a_k.xc = beta_k up to the rounding of that round trip (1e-17)
            double akx = -lane::dot3(c[0], c[1], c[2], S.x0, S.x1, S.x2);
            if constexpr (D == 4) akx = -lane::dot4(c[0], c[1], c[2], c[3], S.x0, S.x1, S.x2, S.x3);
            const double obj = akx - pan[kr * LS];
            const bool keepk = mine & (((S.status == ST_OPT) & (obj > abs_tol)) | (S.status == ST_UNBND));
            const uint64_t all = __ballot(keepk);   // bit (t - rb) * nparts: the LP at list position t says "keep"
            const uint64_t rt = __ballot(mine & (S.status == ST_RETRY));
            if (rt != 0ull) {   // rare
                for (int p = 0; p < NG; ++p)
                    if (__any(mine & (S.status == ST_RETRY) & (tp == p))) retry_polys |= 1u << p;
            }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int rw = row0 + k;
                const int tt = off_g + __popc(todo32 & ((1u << rw) - 1u)) - rb;
                const bool own = (((todo32 >> rw) & 1u) != 0u) & (tt >= 0) & (tt < per);
                const bool kept = own & (((all >> ((tt << sh) & 63)) & 1ull) != 0ull);
                keep |= spread_rows<R, GS>(grp_ballot(kept, g)) << k;
            }
            rb += per;
        }
        retry = retry | (((retry_polys >> gib) & 1u) != 0u);
        if (stage == 2) flags |= RF_MINREP;
    }
    // ---------------------------------------------------------------- results
    if (valid & (g.gl == 0) & !retry) {
        (keep_out + tile)[gib] = keep;
        (flags_out + tile)[gib] = flags;
        (nlp_out + tile)[gib] = nlp;
    }
    ctr_add(ctr, nlp, valid & (g.gl == 0));   // every LP the reference issues, less the presolved ones
    // Polytopes handed back (an LP that needs Bland's rule, dependent active rows; PLP_REDUCE_RETRY_ALL=1: all of them) are
    // redone HERE by the general engine, four at a time on the wavefront's 64 lanes -- the launch is complete, no second
    // pass follows it.  Rare: the call sits behind a wave-uniform branch and is not inlined.
    const uint64_t rb64 = __ballot(retry & valid & (g.gl == 0));   // bit GS p: polytope p of the tile
    if (rb64 != 0ull) {
        __threadfence_block();   // my r / xc stores of these polytopes are out before they are written again
        __syncthreads();
        reduce_lane_redo<D, GS, ROWS>(smem_raw, tile, ntile, rb64, m_max, Ag, bg, mrows, abs_tol, keep_out, flags_out, r_out, xc_out,
                            nlp_out);
    }
#undef LA
#undef OA
#undef OB
#undef ON
#undef LB
#undef LN
}

#ifndef PLP_REDUCE_LANE_WAVES
#define PLP_REDUCE_LANE_WAVES(D) ((D) <= 3 ? 4 : 3)   // (d <= 3, GS = 4: 16 one-wavefront workgroups of 10 240 B are the CU's 160 KB: four waves per SIMD; d = 4: F1's five-column dictionary wants the registers of three)
#endif

template <int D, int GS, int ROWS = LN_ROWS>
__global__ __launch_bounds__(RBLOCK, PLP_REDUCE_LANE_WAVES(D)) void reduce_lane_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg, const int* __restrict__ mrows,
    double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out, int* __restrict__ flags_out,
    double* __restrict__ r_out, double* __restrict__ xc_out, int* __restrict__ nlp_out, unsigned long long* __restrict__ ctr) {
    reduce_lane_tile<D, GS, ROWS>((long long)blockIdx.x * (64 / GS), B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out,
                                  flags_out, r_out, xc_out, nlp_out, ctr);
}

// The first `nbig` workgroups take tiles of 64 / GSA polytopes, the rest tiles of 64 / GSB (half as many, which finish in
// about 0.6 of the time): the launch drains over one tile lifetime, and with short tiles dispatched last that window shrinks.
template <int D, int ROWS, int GSA, int GSB>
__global__ __launch_bounds__(RBLOCK, PLP_REDUCE_LANE_WAVES(D)) void reduce_lane_mix_kernel(
    int nbig, long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg,
    const int* __restrict__ mrows, double abs_tol, int force_retry, unsigned long long* __restrict__ keep_out,
    int* __restrict__ flags_out, double* __restrict__ r_out, double* __restrict__ xc_out, int* __restrict__ nlp_out,
    unsigned long long* __restrict__ ctr) {
    constexpr int NA = 64 / GSA, NB_ = 64 / GSB;
    if ((int)blockIdx.x < nbig)
        reduce_lane_tile<D, GSA, ROWS>((long long)blockIdx.x * NA, B, m_max, Ag, bg, mrows, abs_tol, force_retry, keep_out,
                                       flags_out, r_out, xc_out, nlp_out, ctr);
    else
        reduce_lane_tile<D, GSB, ROWS>((long long)nbig * NA + (long long)((int)blockIdx.x - nbig) * NB_, B, m_max, Ag, bg, mrows,
                                       abs_tol, force_retry, keep_out, flags_out, r_out, xc_out, nlp_out, ctr);
}

template <int D, int GS, int ROWS>
__global__ __launch_bounds__(RBLOCK, PLP_REDUCE_LANE_WAVES(D)) void bbox_lane_kernel(
    long long B, int m_max, const double* __restrict__ Ag, const double* __restrict__ bg, const int* __restrict__ mrows,
    int force_retry, double* __restrict__ lb, double* __restrict__ ub, int* __restrict__ status, double* __restrict__ xfin) {
    // (BBOX: the tile's keep_out argument carries the verifier's buffer of final points, [B][2 D][D] doubles or nullptr)
    reduce_lane_tile<D, GS, ROWS, true>((long long)blockIdx.x * (64 / GS), B, m_max, Ag, bg, mrows, 0.0, force_retry,
                                        reinterpret_cast<unsigned long long*>(xfin), status, lb, ub, nullptr, nullptr);
}

// Tile shape by batch size, measured on (16,3) batches (scripts/debug/lane_sweep.py, us per launch GS 4 / 8 / 16):
//   B = 3 000: 51 / 35 / 27.5    8 000: 56 / 38 / 34    12 000: 58 / 46 / 41    16 000: 58 / 47 / 49    20 000: 66 / 56 / 62
//   30 000: 71 / 69 / 81    40 000: 89 / 88 / 101    (lane-group kernels: 40 / 48 / 60 / 62 / 75 / 91 / 106)
#ifndef PLP_REDUCE_LANE32_GS16_MAXB
#define PLP_REDUCE_LANE32_GS16_MAXB 16000  // 17..32 rows: batches up to this size on 4 polytopes per wavefront
#endif
#ifndef PLP_REDUCE_LANE_GS8_MAXB
#define PLP_REDUCE_LANE_GS8_MAXB 40000   // batches up to this size: 8 polytopes per wavefront
#endif
#ifndef PLP_REDUCE_LANE_GS16_MAXB
#define PLP_REDUCE_LANE_GS16_MAXB 14000  // ... up to this size: 4 polytopes per wavefront
#endif
// Larger batches: 16 polytopes per wavefront, the LAST eighth of the tiles (at most 1024) as 8-polytope tiles.  The
// workgroups of a launch are handed out over tens of microseconds and the launch ends when the ones that started last
// end: short tiles there cut 12 us off 50 000 .. 100 000 polytopes (100 000: 166 us without, 153-155 with 2/64 .. 8/64 of
// the tiles, 159-172 beyond 12/64; 50 000: 104 -> 91).  PLP_REDUCE_LANE_MIX=k: k / 64 of the tiles (0: none).

template <int D>
static int launch_reduce_lane_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double abs_tol,
                                unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    if (B > 2147483647ll) return 2;
    const char* fr = getenv("PLP_REDUCE_RETRY_ALL");
    const int force = (fr && fr[0] == '1') ? 1 : 0;
    const char* eg = getenv("PLP_REDUCE_LANE_GS");   // 4 / 8 / 16: that tile shape whatever the batch size (A/B)
    if (m_max > LN_ROWS) {
        // 17..32 rows: 32 row slots per polytope, 8 polytopes per wavefront (four rows per lane) or 4 (two rows per lane)
        int gs32 = B <= PLP_REDUCE_LANE32_GS16_MAXB ? 16 : 8;
        if (eg) gs32 = atoi(eg) == 16 ? 16 : 8;
        const long long ng32 = 64 / gs32;
        long long blocks32 = (B + ng32 - 1) / ng32;
        if (blocks32 < 1) blocks32 = 1;
        const char* mx32 = getenv("PLP_REDUCE_LANE_MIX");
        long long tail32 = blocks32 / 8 < 1024 ? blocks32 / 8 : 1024;
        if (mx32) tail32 = blocks32 * atoi(mx32) / 64;
        if (eg || D == 4) tail32 = 0;   // (d = 4: the short tiles do not pay at 32 row slots, (20,4) x 50 000: 315 us with, 300 without)
        if (gs32 == 16)
            hipLaunchKernelGGL((reduce_lane_kernel<D, 16, 32>), dim3((unsigned)blocks32), dim3(RBLOCK), reduce_lane_smem_bytes(D, 16, 32),
                               st, B, m_max, A, b, mrows, abs_tol, force, keep, flags, r, xc, nlp, t_reduce_ctr);
        else if (tail32 > 0 && blocks32 > 512) {
            const long long nbig = blocks32 - tail32;
            const long long nsmall = (B - nbig * 8 + 3) / 4;
            hipLaunchKernelGGL((reduce_lane_mix_kernel<D, 32, 8, 16>), dim3((unsigned)(nbig + nsmall)), dim3(RBLOCK),
                               reduce_lane_smem_bytes(D, 8, 32), st, (int)nbig, B, m_max, A, b, mrows, abs_tol, force, keep, flags, r,
                               xc, nlp, t_reduce_ctr);
        } else
            hipLaunchKernelGGL((reduce_lane_kernel<D, 8, 32>), dim3((unsigned)blocks32), dim3(RBLOCK), reduce_lane_smem_bytes(D, 8, 32),
                               st, B, m_max, A, b, mrows, abs_tol, force, keep, flags, r, xc, nlp, t_reduce_ctr);
        return 3;
    }
    int gs = B <= PLP_REDUCE_LANE_GS16_MAXB ? 16 : (B <= PLP_REDUCE_LANE_GS8_MAXB ? 8 : 4);
    if (eg) gs = atoi(eg) == 16 ? 16 : (atoi(eg) == 8 ? 8 : 4);
    const long long ng = 64 / gs;
    long long blocks = (B + ng - 1) / ng;
    if (blocks < 1) blocks = 1;
    if (gs == 16) {
        hipLaunchKernelGGL((reduce_lane_kernel<D, 16>), dim3((unsigned)blocks), dim3(RBLOCK), reduce_lane_smem_bytes(D, 16), st, B,
                           m_max, A, b, mrows, abs_tol, force, keep, flags, r, xc, nlp, t_reduce_ctr);
    } else if (gs == 8) {
        hipLaunchKernelGGL((reduce_lane_kernel<D, 8>), dim3((unsigned)blocks), dim3(RBLOCK), reduce_lane_smem_bytes(D, 8), st, B,
                           m_max, A, b, mrows, abs_tol, force, keep, flags, r, xc, nlp, t_reduce_ctr);
    } else {
        const char* mx = getenv("PLP_REDUCE_LANE_MIX");
        long long tail_tiles = blocks / 8 < 1024 ? blocks / 8 : 1024;
        if (mx) tail_tiles = blocks * atoi(mx) / 64;
        if (eg) tail_tiles = 0;   // (a forced shape is that shape only)
        if (tail_tiles > 0 && blocks > 512) {
            const long long nbig = blocks - tail_tiles;
            const long long rest = B - nbig * 16;
            const long long nsmall = (rest + 7) / 8;
            hipLaunchKernelGGL((reduce_lane_mix_kernel<D, LN_ROWS, 4, 8>), dim3((unsigned)(nbig + nsmall)), dim3(RBLOCK),
                               reduce_lane_smem_bytes(D, 4), st, (int)nbig, B, m_max, A, b, mrows, abs_tol, force, keep, flags, r,
                               xc, nlp, t_reduce_ctr);
        } else {
            hipLaunchKernelGGL((reduce_lane_kernel<D, 4>), dim3((unsigned)blocks), dim3(RBLOCK), reduce_lane_smem_bytes(D, 4), st,
                               B, m_max, A, b, mrows, abs_tol, force, keep, flags, r, xc, nlp, t_reduce_ctr);
        }
    }
    return 3;   // complete: what the fast path hands back is redone inside the kernel, no second pass
}

template <int D>
static int launch_bbox_lane_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double* lb, double* ub,
                              int* status, hipStream_t st, double* xfin) {
    if (B > 2147483647ll) return 1;
    const int force = 0;   // (nothing to force: what this kernel does not settle goes back to the caller as status 1)
    const bool wide = m_max > LN_ROWS;
    int gs = wide ? (B <= PLP_REDUCE_LANE32_GS16_MAXB ? 16 : 8) : (B <= PLP_REDUCE_LANE_GS16_MAXB ? 16 : (B <= PLP_REDUCE_LANE_GS8_MAXB ? 8 : 4));
    const long long ng = 64 / gs;
    long long blocks = (B + ng - 1) / ng;
    if (blocks < 1) blocks = 1;
#define PLP_BBL(GSV, RV)                                                                                                     \
    hipLaunchKernelGGL((bbox_lane_kernel<D, GSV, RV>), dim3((unsigned)blocks), dim3(RBLOCK), reduce_lane_smem_bytes(D, GSV, RV), st, B, \
                       m_max, A, b, mrows, force, lb, ub, status, xfin)
    if (wide) { if (gs == 16) PLP_BBL(16, 32); else PLP_BBL(8, 32); }
    else if (gs == 16) PLP_BBL(16, 16);
    else if (gs == 8) PLP_BBL(8, 16);
    else PLP_BBL(4, 16);
#undef PLP_BBL
    return 0;
}

// bounding boxes of polytopes with up to 32 rows in d <= 3 (the contract of launch_bbox); 0 when launched, 1 when not taken
int launch_bbox_lane(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb, double* ub,
                     int* status, hipStream_t st, double* xfin) {
    if (m_max < 1 || m_max > 2 * LN_ROWS) return 1;
    switch (d) {
        case 1: return launch_bbox_lane_d<1>(B, m_max, A, b, mrows, lb, ub, status, st, xfin);
        case 2: return launch_bbox_lane_d<2>(B, m_max, A, b, mrows, lb, ub, status, st, xfin);
        case 3: return launch_bbox_lane_d<3>(B, m_max, A, b, mrows, lb, ub, status, st, xfin);
        default: return 1;
    }
}

// returns 3 when launched (complete: launch_reduce adds no second pass), 1 when this kernel does not take the shape
int launch_reduce_lane(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                       unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    if (m_max < 1 || m_max > 2 * LN_ROWS) return 1;
    switch (d) {
        case 1: return launch_reduce_lane_d<1>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        case 2: return launch_reduce_lane_d<2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        case 3: return launch_reduce_lane_d<3>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        case 4: return launch_reduce_lane_d<4>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        default: return 1;
    }
}

}  // namespace plp
