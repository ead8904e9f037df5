// plp_lane_lp.hpp -- ONE LP PER LANE for d = 3: the box (F3) and redundancy (F2) LPs of the fused reduce
// (polytope/polytope.py:1118-1134, :1142-1160) when the polytope's Chebyshev centre is known.
//
// Every such LP is   min c.x'   s.t.  a_i.x' <= beta_i  (beta_i = b_i - a_i.xc > 0: the centre is strictly inside),
// x' free, started at x' = 0.  The dictionary engines (plp_simplex_r.hpp) spread ONE LP over a lane group and carry a
// dictionary whose pivot costs ~280 VALU instructions per 16 LPs of a wavefront, most of them cross-lane traffic and
// selects on a dynamic pivot column.  Here a lane owns a whole LP and carries NO dictionary: the rows stay in LDS (the 6
// box LPs and the redundancy LPs of a polytope read the same rows), the lane keeps the point x', the (at most three)
// active rows and walks
//     interior --(-c)--> a facet --(projected -c)--> an edge --(along it)--> a vertex --(edge by edge)--> the optimum,
// i.e. the primal active-set form of the simplex method in x-space: a step is a ratio test over the rows (the only loop),
// a direction is a projection written with cross products.  64 LPs advance per instruction instead of 16 and nothing
// crosses lanes.  The optimum VALUE of an LP is unique, so box values and redundancy objectives agree with any other
// simplex code to rounding (1e-13 here); the verdicts taken from them (prefilter < -1e-4, objective > abs_tol) are the
// oracle's except on exact ties, which no two LP codes share.
//
// What is NOT decided here is handed back (ST_RETRY -> the polytope is redone by the general engine with Bland's rule):
// a run of degenerate steps (cycling risk), active rows that are numerically dependent, the iteration cap.
//
// The same source compiles for the host (g++, tests/cabi/lane_lp_host.cpp: the engine against the oracle's simplex on
// millions of LPs without a GPU) and for the device.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include "plp_common.hpp"
#define PLP_LANE_FN __device__ __forceinline__
#else
#define PLP_LANE_FN static inline
namespace plp {
enum : int { ST_OPT = 0, ST_ITER = 1, ST_INFEAS = 2, ST_UNBND = 3, ST_NUM = 4 };
constexpr int ST_RETRY = 5;
constexpr double TOL_D = 1e-9, TOL_PIV = 1e-9, DEGEN_EPS = 1e-12;
constexpr int BLAND_AFTER = 6;
}  // namespace plp
#endif

namespace plp {
namespace lane {

constexpr int LANE_MAX_ITERS = 48;   // steps + drops of one LP before it is handed back

struct Lp3 {
    double x0, x1, x2;     // x' (relative to the Chebyshev centre)
    int w0, w1, w2;        // active rows (w0 the oldest); valid up to nact
    int nact;
    int status;            // -1 running
    int iters;
    int ndeg;              // consecutive degenerate steps
};

PLP_LANE_FN void cross3(double a0, double a1, double a2, double b0, double b1, double b2, double& e0, double& e1,
                        double& e2) {
    e0 = fma(a1, b2, -(a2 * b1));
    e1 = fma(a2, b0, -(a0 * b2));
    e2 = fma(a0, b1, -(a1 * b0));
}

PLP_LANE_FN double dot3(double a0, double a1, double a2, double b0, double b1, double b2) {
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}

// One LP, to the end.  ROWS(i, a0, a1, a2): row i of the polytope (zeroed rows allowed: they never block).
// RATIO(d0, d1, d2, x0, x1, x2, tolp, bs, bd, bi): the ratio test -- over the rows with a.d > tolp the one with the smallest
// (beta_i - a_i.x)+ / a_i.d, the FIRST such row on ties; bs / bd its slack and a.d, bi its index (-1: none).  The caller
// owns the loop: the host build and the plain device form walk all rows in one lane (ratio_rows below), the device
// may split the rows of one LP over two or four lanes and combine (plp_reduce_lane.hip).
// ANY(pred): true while any lane of the wavefront still runs (device: __any; host: the predicate itself).
template <class RowF, class RatioF, class AnyF>
PLP_LANE_FN void walk3(Lp3& S, const double c0, const double c1, const double c2, const bool go, RowF ROWS, RatioF RATIO,
                       AnyF ANY) {
    S.x0 = S.x1 = S.x2 = 0.0;
    S.w0 = S.w1 = S.w2 = -1;
    S.nact = 0;
    S.iters = 0;
    S.ndeg = 0;
    S.status = go ? -1 : ST_OPT;
    const double cn1 = fabs(c0) + fabs(c1) + fabs(c2);
    if (go && !(cn1 > 0.0)) S.status = ST_OPT;   // c = 0: every point is optimal
    while (ANY(S.status < 0)) {
        const bool run = S.status < 0;
        // ---------------- direction: -c projected onto the planes of the active rows (scaled by positive factors)
        double d0 = -c0, d1 = -c1, d2 = -c2;
        double n00 = 0, n01 = 0, n02 = 0, n10 = 0, n11 = 0, n12 = 0, n20 = 0, n21 = 0, n22 = 0;
        if (ANY(run && S.nact >= 1)) {
            if (S.nact >= 1) ROWS(S.w0, n00, n01, n02);
            if (S.nact >= 2) ROWS(S.w1, n10, n11, n12);
            if (S.nact >= 3) ROWS(S.w2, n20, n21, n22);
        }
        bool stalled = false;      // no descent left inside the active planes: look at the multipliers
        double dscale = cn1;       // |d|_1 of a direction that is "as long as c" (what TOL_D is relative to)
        if (S.nact == 1) {
            const double nn = dot3(n00, n01, n02, n00, n01, n02);
            const double cn = dot3(c0, c1, c2, n00, n01, n02);
            d0 = fma(cn, n00, -(nn * c0));
            d1 = fma(cn, n01, -(nn * c1));
            d2 = fma(cn, n02, -(nn * c2));
            dscale = nn * cn1;
        } else if (S.nact == 2) {
            double e0, e1, e2;
            cross3(n00, n01, n02, n10, n11, n12, e0, e1, e2);
            const double ce = dot3(c0, c1, c2, e0, e1, e2);
            d0 = -(ce * e0);
            d1 = -(ce * e1);
            d2 = -(ce * e2);
            dscale = dot3(e0, e1, e2, e0, e1, e2) * cn1;
            // the two active rows (numerically) parallel: their planes do not define an edge
            const double nn0 = dot3(n00, n01, n02, n00, n01, n02), nn1 = dot3(n10, n11, n12, n10, n11, n12);
            if (run && !(dot3(e0, e1, e2, e0, e1, e2) > 1e-16 * nn0 * nn1)) S.status = ST_RETRY;
        } else if (S.nact == 3) {
            d0 = d1 = d2 = 0.0;
        }
        {
            const double dn1 = fabs(d0) + fabs(d1) + fabs(d2);
            stalled = !(dn1 > TOL_D * dscale);
        }
        if (ANY(S.status < 0 && stalled)) {
            if (S.status < 0 && stalled) {
                // multipliers of  c + sum lambda_j n_j = 0  over the active rows (least squares when fewer than three)
                // all >= -tol: optimal.  Otherwise the row with the most negative multiplier is dropped and the
                // direction is the projection onto the remaining ones (a descent direction that leaves that row)
                if (S.nact == 0) {
                    S.status = ST_OPT;   // (c ~ 0 relative to itself cannot happen; kept for completeness)
                } else if (S.nact == 1) {
                    const double cn = dot3(c0, c1, c2, n00, n01, n02);
                    if (cn <= 0.0) S.status = ST_OPT;           // lambda = -c.n / n.n >= 0
                    else { S.nact = 0; d0 = -c0; d1 = -c1; d2 = -c2; }   // (leaves the row; cannot follow a step onto it)
                } else if (S.nact == 2) {
                    const double g00 = dot3(n00, n01, n02, n00, n01, n02), g11 = dot3(n10, n11, n12, n10, n11, n12);
                    const double g01 = dot3(n00, n01, n02, n10, n11, n12);
                    const double r0 = -dot3(c0, c1, c2, n00, n01, n02), r1 = -dot3(c0, c1, c2, n10, n11, n12);
                    // det > 0 (checked above); lambda_0 ~ r0 g11 - r1 g01, lambda_1 ~ r1 g00 - r0 g01
                    const double l0 = fma(r0, g11, -(r1 * g01)), l1 = fma(r1, g00, -(r0 * g01));
                    // lambda_j |n_j| >= -TOL_D |c|, with |n_j| <= w_j = (1 + n_j.n_j) / 2 in its place (no square root;
                    // equal for unit rows, a little stricter otherwise) and det <= g00 g11
                    const double w0 = 0.5 * (1.0 + g00), w1 = 0.5 * (1.0 + g11);
                    const double tol0 = TOL_D * g11 * w0 * cn1, tol1 = TOL_D * g00 * w1 * cn1;
                    if (l0 >= -tol0 && l1 >= -tol1) S.status = ST_OPT;
                    else {
                        // drop the more negative one (as weighted multipliers; a row with a negative one either way)
                        const bool drop0 = l0 * w0 < l1 * w1;
                        if (drop0) { S.w0 = S.w1; n00 = n10; n01 = n11; n02 = n12; }
                        S.nact = 1;
                        const double nn = dot3(n00, n01, n02, n00, n01, n02);
                        const double cn = dot3(c0, c1, c2, n00, n01, n02);
                        d0 = fma(cn, n00, -(nn * c0));
                        d1 = fma(cn, n01, -(nn * c1));
                        d2 = fma(cn, n02, -(nn * c2));
                    }
                } else {
                    double u00, u01, u02, u10, u11, u12, u20, u21, u22;
                    cross3(n10, n11, n12, n20, n21, n22, u00, u01, u02);   // u0 = n1 x n2
                    cross3(n20, n21, n22, n00, n01, n02, u10, u11, u12);   // u1 = n2 x n0
                    cross3(n00, n01, n02, n10, n11, n12, u20, u21, u22);   // u2 = n0 x n1
                    const double det = dot3(n00, n01, n02, u00, u01, u02);
                    const double g00 = dot3(n00, n01, n02, n00, n01, n02), g11 = dot3(n10, n11, n12, n10, n11, n12);
                    const double g22 = dot3(n20, n21, n22, n20, n21, n22);
                    if (!(det * det > 1e-18 * (g00 * g11 * g22))) {
                        S.status = ST_RETRY;   // three active planes that (nearly) share a line
                    } else {
                        // lambda_j = -(c.u_j) / det ; compare sign-corrected numerators lambda_j |det| = -(c.u_j) sgn(det)
                        const double sg = det > 0.0 ? 1.0 : -1.0;
                        const double l0 = -sg * dot3(c0, c1, c2, u00, u01, u02);
                        const double l1 = -sg * dot3(c0, c1, c2, u10, u11, u12);
                        const double l2 = -sg * dot3(c0, c1, c2, u20, u21, u22);
                        // lambda_j |n_j| / |c|  =  l_j |n_j| / (|det| |c|), with w_j = (1 + n_j.n_j) / 2 >= |n_j| in its place
                        const double a0 = l0 * (0.5 * (1.0 + g00)), a1 = l1 * (0.5 * (1.0 + g11)), a2 = l2 * (0.5 * (1.0 + g22));
                        const double tol = TOL_D * fabs(det) * cn1;
                        if (a0 >= -tol && a1 >= -tol && a2 >= -tol) S.status = ST_OPT;
                        else {
                            // most negative goes; the edge of the other two, oriented off the dropped row: -sgn(det) u_j
                            int j = 0;
                            double am = a0;
                            if (a1 < am) { am = a1; j = 1; }
                            if (a2 < am) { am = a2; j = 2; }
                            if (j == 0) { d0 = -sg * u00; d1 = -sg * u01; d2 = -sg * u02; S.w0 = S.w1; S.w1 = S.w2; }
                            else if (j == 1) { d0 = -sg * u10; d1 = -sg * u11; d2 = -sg * u12; S.w1 = S.w2; }
                            else { d0 = -sg * u20; d1 = -sg * u21; d2 = -sg * u22; }
                            S.nact = 2;
                        }
                    }
                }
            }
        }
        // ---------------- ratio test over the rows: the first row the ray x' + t d meets
        const bool step = S.status < 0;
        const double dn1 = fabs(d0) + fabs(d1) + fabs(d2);
        const double tolp = TOL_PIV * dn1;
        double bs = 1.0, bd = 0.0;   // best slack / best a.d  (ratio bs / bd; bd = 0: none yet)
        int bi = -1;
        if (ANY(step)) RATIO(d0, d1, d2, S.x0, S.x1, S.x2, tolp, bs, bd, bi);
        if (step) {
            ++S.iters;
            if (bi < 0) {
                S.status = ST_UNBND;     // a descent ray that no row stops
            } else {
                const double t = bs / bd;
                S.x0 = fma(t, d0, S.x0);
                S.x1 = fma(t, d1, S.x1);
                S.x2 = fma(t, d2, S.x2);
                if (S.nact == 0) S.w0 = bi;
                else if (S.nact == 1) S.w1 = bi;
                else S.w2 = bi;
                S.nact += 1;
                S.ndeg = (t * dn1 <= DEGEN_EPS) ? S.ndeg + 1 : 0;
                if (S.ndeg >= BLAND_AFTER || S.iters >= LANE_MAX_ITERS) S.status = ST_RETRY;
            }
        }
    }
}

// One step of the ratio test: row (a0, a1, a2) with right-hand side `beta` and index i against the best so far.
//     sl / ad < bs / bd   <=>   sl * bd < bs * ad      (ad, bd > 0; nothing yet: bd = 0 -> 0 < bs * ad)
PLP_LANE_FN void ratio_row(const double a0, const double a1, const double a2, const double beta, const int i, const double d0,
                           const double d1, const double d2, const double x0, const double x1, const double x2,
                           const double tolp, double& bs, double& bd, int& bi) {
    const double ad = dot3(a0, a1, a2, d0, d1, d2);
    const double ax = dot3(a0, a1, a2, x0, x1, x2);
    const double sl = fmax(beta - ax, 0.0);
    const bool better = (ad > tolp) & (sl * bd < bs * ad);
    bs = better ? sl : bs;
    bd = better ? ad : bd;
    bi = better ? i : bi;
}

// The whole LP in one lane, M row slots, BETA(i) the right-hand sides (>= 0).
// PASS(): called at the top of every ratio test (device: keeps loop-invariant per-row predicates of BETA from being
// hoisted out of the walk).
template <int M, class RowF, class BetaF, class AnyF, class PassF>
PLP_LANE_FN void solve3(Lp3& S, const double c0, const double c1, const double c2, const bool go, RowF ROWS, BetaF BETA,
                        AnyF ANY, PassF PASS) {
    walk3(S, c0, c1, c2, go, ROWS,
          [&](double d0, double d1, double d2, double x0, double x1, double x2, double tolp, double& bs, double& bd, int& bi) {
              PASS();
#pragma unroll
              for (int i = 0; i < M; ++i) {
                  double a0, a1, a2;
                  ROWS(i, a0, a1, a2);
                  ratio_row(a0, a1, a2, BETA(i), i, d0, d1, d2, x0, x1, x2, tolp, bs, bd, bi);
              }
          },
          ANY);
}

template <int M, class RowF, class BetaF, class AnyF>
PLP_LANE_FN void solve3(Lp3& S, const double c0, const double c1, const double c2, const bool go, RowF ROWS, BetaF BETA,
                        AnyF ANY) {
    solve3<M>(S, c0, c1, c2, go, ROWS, BETA, ANY, [] {});
}

}  // namespace lane
}  // namespace plp
