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
constexpr double TOL_D = 1e-9, TOL_PIV = 1e-7, DEGEN_EPS = 1e-12;
constexpr int BLAND_AFTER = 6;
}  // namespace plp
#endif

namespace plp {
namespace lane {

constexpr int LANE_MAX_ITERS = 48;   // steps + drops of one LP before it is handed back
// The walk's own tolerances, RELATIVE to the direction's / the cost's length (the dictionary simplex's TOL_D / TOL_PIV are
// absolute 1e-9 on dictionary entries).  A row whose a.d is below LANE_TOL_PIV |d| is not a blocking row: the walk may pass
// it by t a.d, a few 1e-11; a projected gradient below LANE_TOL_D |c| counts as zero: the optimum is off by no more than that
// times the distance left.  (1e-9 for both, the first version, showed as 2e-10 on 3 of 120 000 box LPs at d = 4.)
constexpr double LANE_TOL_D = 1e-11, LANE_TOL_PIV = 1e-11;
// what cancellation leaves of a multiplier's numerator, as a multiple of the sum of its terms' magnitudes (walk4: 64 ulps)
constexpr double LANE_M_NOISE = 64 * 2.220446049250313e-16;

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

// -c projected onto the plane of ONE active row n, scaled by n.n > 0:  p = (c.n) n - (n.n) c  -- and, where little of c is
// left, once more against n (then scaled by (n.n)^2; `scale` says which).  p is the difference of two vectors of length
// |c| n.n: when c is nearly along n (a box LP against a row that is nearly an axis) what is left of them is orthogonal to
// n only up to THEIR rounding, 1e-16 |c| n.n, and a step of length t along it leaves the plane by t (n.p) -- 5e-9 on box rows
// tilted by 1e-8 (scripts/soak_lane.py).  The second pass removes it; it runs per lane (the result of an LP does not depend
// on its neighbours in the wavefront) and only when some lane of the wavefront needs it (ANY).
template <class AnyF>
PLP_LANE_FN void proj1(const double c0, const double c1, const double c2, const double cn1, const double n0, const double n1,
                       const double n2, const double nn, double& d0, double& d1, double& d2, double& scale, AnyF ANY) {
    const double cn = dot3(c0, c1, c2, n0, n1, n2);
    d0 = fma(cn, n0, -(nn * c0));
    d1 = fma(cn, n1, -(nn * c1));
    d2 = fma(cn, n2, -(nn * c2));
    scale = nn;
    const double pn1 = fabs(d0) + fabs(d1) + fabs(d2);
    const bool small = (pn1 > 0.0) & (pn1 < 1e-4 * (nn * cn1));
    if (ANY(small)) {
        const double np_ = dot3(n0, n1, n2, d0, d1, d2);
        const double q0 = fma(nn, d0, -(np_ * n0)), q1 = fma(nn, d1, -(np_ * n1)), q2 = fma(nn, d2, -(np_ * n2));
        d0 = small ? q0 : d0;
        d1 = small ? q1 : d1;
        d2 = small ? q2 : d2;
        scale = small ? nn * nn : nn;
    }
}

// Two active rows closer than this (sin^2 of their angle) do not define an edge the walk can follow: their cross product
// carries a relative error of 1e-16 / sin, and the walk would drift off both planes by that times the step.  1e-10 = 1e-5 rad:
// drift below 1e-10 of the step.  (The fused reduce removes rows closer than 4.5e-4 rad before any LP runs, ref :1094-1110.)
constexpr double LANE_PAIR_SIN2 = 1e-10;

// One LP, to the end.  ROWS(i, a0, a1, a2): row i of the polytope (zeroed rows allowed: they never block).
// RATIO(d0, d1, d2, x0, x1, x2, tolp, bs, bd, bi): the ratio test -- over the rows with a.d > tolp the one with the smallest
// (beta_i - a_i.x)+ / a_i.d, the FIRST such row on ties; bs / bd its slack and a.d, bi its index (-1: none).  The caller
// owns the loop: the host build and the plain device form walk all rows in one lane (ratio_rows below), the device
// may split the rows of one LP over two or four lanes and combine (plp_reduce_lane.hip).
// ANY(pred): true while any lane of the wavefront still runs (device: __any; host: the predicate itself).
// RATIO0(d0, d1, d2, tolp, bs, bd, bi): the same test from x' = 0, where every slack is its beta_i (the first pass of every
// walk: no a_i.x to form; the values are the ones RATIO would find, bit for bit).
// `warm` (the same for every lane of the wavefront): S comes filled in by the caller -- a feasible point x' ON the planes
// of its nact active rows w0.. (a vertex, an edge or a facet point another LP of the same polytope ended on), iters =
// ndeg = 0 -- and the walk goes on from there instead of from the centre.
template <class RowF, class RatioF, class Ratio0F, class AnyF>
PLP_LANE_FN void walk3(Lp3& S, const double c0, const double c1, const double c2, const bool go, RowF ROWS, RatioF RATIO,
                       Ratio0F RATIO0, AnyF ANY, const bool warm = false) {
    if (!warm) {
        S.x0 = S.x1 = S.x2 = 0.0;
        S.w0 = S.w1 = S.w2 = -1;
        S.nact = 0;
        S.iters = 0;
        S.ndeg = 0;
    }
    S.status = go ? -1 : ST_OPT;
    const double cn1 = fabs(c0) + fabs(c1) + fabs(c2);
    if (go && !(cn1 > 0.0)) S.status = ST_OPT;   // c = 0: every point is optimal
    // ---------------- first pass, all lanes together: from the interior point along -c to the first facet (what the
    // general pass below does at nact = 0, without its case distinctions)
    if (!warm && ANY(S.status < 0)) {
        const bool step = S.status < 0;
        const double d0 = -c0, d1 = -c1, d2 = -c2;
        const double tolp = LANE_TOL_PIV * cn1;
        double bs = 1.0, bd = 0.0;
        int bi = -1;
        RATIO0(d0, d1, d2, tolp, bs, bd, bi);
        if (step) {
            S.iters = 1;
            if (bi < 0) {
                S.status = ST_UNBND;
            } else {
                const double t = bs / bd;
                S.x0 = fma(t, d0, 0.0);
                S.x1 = fma(t, d1, 0.0);
                S.x2 = fma(t, d2, 0.0);
                S.w0 = bi;
                S.nact = 1;
                S.ndeg = (t * cn1 <= DEGEN_EPS) ? 1 : 0;
            }
        }
    }
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
            double sc;
            proj1(c0, c1, c2, cn1, n00, n01, n02, nn, d0, d1, d2, sc, ANY);
            dscale = sc * cn1;
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
            if (run && !(dot3(e0, e1, e2, e0, e1, e2) > LANE_PAIR_SIN2 * nn0 * nn1)) S.status = ST_RETRY;
        } else if (S.nact == 3) {
            d0 = d1 = d2 = 0.0;
        }
        {
            const double dn1 = fabs(d0) + fabs(d1) + fabs(d2);
            stalled = !(dn1 > LANE_TOL_D * dscale);
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
                    const double tol0 = LANE_TOL_D * g11 * w0 * cn1, tol1 = LANE_TOL_D * g00 * w1 * cn1;
                    if (l0 >= -tol0 && l1 >= -tol1) S.status = ST_OPT;
                    else {
                        // drop the more negative one (as weighted multipliers; a row with a negative one either way)
                        const bool drop0 = l0 * w0 < l1 * w1;
                        if (drop0) { S.w0 = S.w1; n00 = n10; n01 = n11; n02 = n12; }
                        S.nact = 1;
                        double sc;
                        proj1(c0, c1, c2, cn1, n00, n01, n02, dot3(n00, n01, n02, n00, n01, n02), d0, d1, d2, sc, [](bool q) { return q; });
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
                        const double tol = LANE_TOL_D * fabs(det) * cn1;
                        if (a0 >= -tol && a1 >= -tol && a2 >= -tol) S.status = ST_OPT;
                        else {
                            // most negative goes; the edge of the other two, oriented off the dropped row: -sgn(det) u_j
                            int j = 0;
                            double am = a0;
                            if (a1 < am) { am = a1; j = 1; }
                            if (a2 < am) { am = a2; j = 2; }
                            double gg;   // n_a.n_a n_b.n_b of the two rows that stay: u_j is THEIR cross product
                            if (j == 0) { d0 = -sg * u00; d1 = -sg * u01; d2 = -sg * u02; S.w0 = S.w1; S.w1 = S.w2; gg = g11 * g22; }
                            else if (j == 1) { d0 = -sg * u10; d1 = -sg * u11; d2 = -sg * u12; S.w1 = S.w2; gg = g22 * g00; }
                            else { d0 = -sg * u20; d1 = -sg * u21; d2 = -sg * u22; gg = g00 * g11; }
                            S.nact = 2;
                            if (!(dot3(d0, d1, d2, d0, d1, d2) > LANE_PAIR_SIN2 * gg)) S.status = ST_RETRY;
                        }
                    }
                }
            }
        }
        // ---------------- ratio test over the rows: the first row the ray x' + t d meets
        const bool step = S.status < 0;
        const double dn1 = fabs(d0) + fabs(d1) + fabs(d2);
        const double tolp = LANE_TOL_PIV * dn1;
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

// ------------------------------------------------------------------------------------------------------------------
template <class RowF, class RatioF, class AnyF>
PLP_LANE_FN void walk3(Lp3& S, const double c0, const double c1, const double c2, const bool go, RowF ROWS, RatioF RATIO,
                       AnyF ANY, const bool warm = false) {
    walk3(S, c0, c1, c2, go, ROWS, RATIO,
          [&](double d0, double d1, double d2, double tolp, double& bs, double& bd, int& bi) {
              RATIO(d0, d1, d2, 0.0, 0.0, 0.0, tolp, bs, bd, bi);
          },
          ANY, warm);
}

// The same walk in R^4 (d = 4: the dimension of BASELINE config 4).  Directions are projections written with the Gram
// matrix of the active rows and its adjugate (no division: everything is scaled by the positive determinant), the
// vertex test with generalised cross products:  for rows p, q, r of R^4,  e = gcross(p, q, r)  is orthogonal to all
// three and  v.e = det[v; p; q; r].
struct Lp4 {
    double x0, x1, x2, x3;
    int w0, w1, w2, w3;
    int nact, status, iters, ndeg;
};

PLP_LANE_FN double dot4(double a0, double a1, double a2, double a3, double b0, double b1, double b2, double b3) {
    return fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 * b0)));
}

PLP_LANE_FN void gcross4(double p0, double p1, double p2, double p3, double q0, double q1, double q2, double q3, double r0,
                         double r1, double r2, double r3, double& e0, double& e1, double& e2, double& e3) {
    // 2 x 2 minors of (q, r), then the four 3 x 3 minors with alternating signs
    const double m01 = fma(q0, r1, -(q1 * r0)), m02 = fma(q0, r2, -(q2 * r0)), m03 = fma(q0, r3, -(q3 * r0));
    const double m12 = fma(q1, r2, -(q2 * r1)), m13 = fma(q1, r3, -(q3 * r1)), m23 = fma(q2, r3, -(q3 * r2));
    e0 = fma(p1, m23, fma(-p2, m13, p3 * m12));
    e1 = -fma(p0, m23, fma(-p2, m03, p3 * m02));
    e2 = fma(p0, m13, fma(-p1, m03, p3 * m01));
    e3 = -fma(p0, m12, fma(-p1, m02, p2 * m01));
}

// ROWS(i, a0, a1, a2, a3);  RATIO(d0, d1, d2, d3, x0, x1, x2, x3, tolp, bs, bd, bi);  ANY as in walk3.
template <class RowF, class RatioF, class AnyF>
PLP_LANE_FN void walk4(Lp4& S, const double c0, const double c1, const double c2, const double c3, const bool go, RowF ROWS,
                       RatioF RATIO, AnyF ANY) {
    S.x0 = S.x1 = S.x2 = S.x3 = 0.0;
    S.w0 = S.w1 = S.w2 = S.w3 = -1;
    S.nact = 0;
    S.iters = 0;
    S.ndeg = 0;
    S.status = go ? -1 : ST_OPT;
    const double cn1 = fabs(c0) + fabs(c1) + fabs(c2) + fabs(c3);
    if (go && !(cn1 > 0.0)) S.status = ST_OPT;
    while (ANY(S.status < 0)) {
        const bool run = S.status < 0;
        bool dropped = false;   // this pass ended by taking a row off the list (no step)
        double d0 = -c0, d1 = -c1, d2 = -c2, d3 = -c3;
        double n00 = 0, n01 = 0, n02 = 0, n03 = 0, n10 = 0, n11 = 0, n12 = 0, n13 = 0;
        double n20 = 0, n21 = 0, n22 = 0, n23 = 0, n30 = 0, n31 = 0, n32 = 0, n33 = 0;
        if (ANY(run && S.nact >= 1)) {
            if (S.nact >= 1) ROWS(S.w0, n00, n01, n02, n03);
            if (S.nact >= 2) ROWS(S.w1, n10, n11, n12, n13);
            if (S.nact >= 3) ROWS(S.w2, n20, n21, n22, n23);
            if (S.nact >= 4) ROWS(S.w3, n30, n31, n32, n33);
        }
        // Gram entries and right-hand sides r_j = c.n_j of the active rows (zero rows where there is none)
        const double g00 = dot4(n00, n01, n02, n03, n00, n01, n02, n03), g11 = dot4(n10, n11, n12, n13, n10, n11, n12, n13);
        const double g22 = dot4(n20, n21, n22, n23, n20, n21, n22, n23);
        const double g01 = dot4(n00, n01, n02, n03, n10, n11, n12, n13), g02 = dot4(n00, n01, n02, n03, n20, n21, n22, n23);
        const double g12 = dot4(n10, n11, n12, n13, n20, n21, n22, n23);
        const double r0 = dot4(c0, c1, c2, c3, n00, n01, n02, n03), r1 = dot4(c0, c1, c2, c3, n10, n11, n12, n13);
        const double r2 = dot4(c0, c1, c2, c3, n20, n21, n22, n23);
        // m_j: det * (coefficient of n_j in the projection of c onto the span of the active rows); lambda_j = -m_j / det
        double det = 1.0, m0 = 0.0, m1 = 0.0, m2 = 0.0;
        if (S.nact == 1) {
            det = g00;
            m0 = r0;
        } else if (S.nact == 2) {
            det = fma(g00, g11, -(g01 * g01));
            m0 = fma(r0, g11, -(r1 * g01));
            m1 = fma(r1, g00, -(r0 * g01));
            if (run && !(det > LANE_PAIR_SIN2 * g00 * g11)) S.status = ST_RETRY;   // (numerically) parallel active rows
        } else if (S.nact == 3) {
            const double A00 = fma(g11, g22, -(g12 * g12)), A01 = fma(g02, g12, -(g01 * g22)), A02 = fma(g01, g12, -(g02 * g11));
            const double A11 = fma(g00, g22, -(g02 * g02)), A12 = fma(g01, g02, -(g00 * g12)), A22 = fma(g00, g11, -(g01 * g01));
            det = fma(g00, A00, fma(g01, A01, g02 * A02));
            m0 = fma(A00, r0, fma(A01, r1, A02 * r2));
            m1 = fma(A01, r0, fma(A11, r1, A12 * r2));
            m2 = fma(A02, r0, fma(A12, r1, A22 * r2));
            if (run && !(det > 1e-16 * g00 * g11 * g22)) S.status = ST_RETRY;   // three active planes that nearly share a plane
        }
        // The direction  det * (-c projected off the active rows) = sum m_j n_j - det c  is the difference of vectors of length
        // det |c|: where the cost lies nearly IN the span of the active rows (a box LP against a row tilted by 1e-9 from its
        // axis) what is left of them is orthogonal to the rows only up to THEIR rounding, and the step along it -- as long as
        // the direction is short -- leaves the planes by that much (1.4e-4 on such a polytope: scripts/soak_lane.py, seed 23;
        // the box value then fails the prefilter's 1e-4 by a hair and a facet is dropped).  One row: projected a second
        // time, always (proj1's reason).  Two rows: projected a second time where less than 1e-4 of the cost is left.
        // Three rows: the complement is a line, e = gcross(n_0, n_1, n_2) (orthogonal to all three to rounding of ITS
        // length, |e|^2 = det), and the direction is -(c.e) e -- the same vector, det * (projection of -c), without a
        // difference.
        bool twice = false;   // (nact == 2: the direction carries another factor det)
        if (S.nact == 3) {
            double e0, e1, e2, e3;
            gcross4(n00, n01, n02, n03, n10, n11, n12, n13, n20, n21, n22, n23, e0, e1, e2, e3);
            const double ce = dot4(c0, c1, c2, c3, e0, e1, e2, e3);
            d0 = -(ce * e0);
            d1 = -(ce * e1);
            d2 = -(ce * e2);
            d3 = -(ce * e3);
        } else if (S.nact >= 1 && S.nact <= 2) {
            d0 = fma(m1, n10, fma(m0, n00, -(det * c0)));
            d1 = fma(m1, n11, fma(m0, n01, -(det * c1)));
            d2 = fma(m1, n12, fma(m0, n02, -(det * c2)));
            d3 = fma(m1, n13, fma(m0, n03, -(det * c3)));
            if (S.nact == 1) {   // once more against the row (proj1's reason)
                const double nd = dot4(n00, n01, n02, n03, d0, d1, d2, d3);
                d0 = fma(g00, d0, -(nd * n00));
                d1 = fma(g00, d1, -(nd * n01));
                d2 = fma(g00, d2, -(nd * n02));
                d3 = fma(g00, d3, -(nd * n03));
            } else {
                const double dn = fabs(d0) + fabs(d1) + fabs(d2) + fabs(d3);
                twice = (dn > 0.0) & (dn < 1e-4 * (det * cn1));
                if (ANY(run && twice)) {
                    const double s0 = dot4(d0, d1, d2, d3, n00, n01, n02, n03), s1 = dot4(d0, d1, d2, d3, n10, n11, n12, n13);
                    const double k0 = fma(s0, g11, -(s1 * g01)), k1 = fma(s1, g00, -(s0 * g01));
                    const double q0 = fma(det, d0, -fma(k1, n10, k0 * n00)), q1 = fma(det, d1, -fma(k1, n11, k0 * n01));
                    const double q2 = fma(det, d2, -fma(k1, n12, k0 * n02)), q3 = fma(det, d3, -fma(k1, n13, k0 * n03));
                    d0 = twice ? q0 : d0;
                    d1 = twice ? q1 : d1;
                    d2 = twice ? q2 : d2;
                    d3 = twice ? q3 : d3;
                }
            }
        } else if (S.nact == 4) {
            d0 = d1 = d2 = d3 = 0.0;
        }
        const double dscale = (S.nact >= 1 && S.nact <= 3) ? ((S.nact == 1 || twice) ? det * det : det) * cn1 : cn1;
        const bool stalled = !(fabs(d0) + fabs(d1) + fabs(d2) + fabs(d3) > LANE_TOL_D * dscale);
        if (ANY(S.status < 0 && stalled)) {
            if (S.status < 0 && stalled) {
                // multipliers of  c + sum lambda_j n_j = 0:  lambda_j |n_j| >= -TOL_D |c|  with  w_j = (1 + n_j.n_j) / 2 >= |n_j|
                // in its place; all fine: optimal.  Otherwise the row with the most negative one is dropped.
                const double w0 = 0.5 * (1.0 + g00), w1 = 0.5 * (1.0 + g11), w2 = 0.5 * (1.0 + g22);
                if (S.nact == 0) {
                    S.status = ST_OPT;
                } else if (S.nact <= 3) {
                    // The m_j are sums of products that cancel (m_j = 0 for a row the cost does not lean on): what is left of
                    // the cancellation -- a few ulps of the terms -- is NOT a negative multiplier, however small det makes the
                    // tolerance (three rows of norms 2.3 / 0.1 / 2.2, det 2.7e-6: m_1 = 1.6e-16 against a tolerance of 2.7e-17;
                    // scripts/soak_lane.py, seed 112, family `scaled`).  Hence the floor under each tolerance.
                    const double tol = LANE_TOL_D * det * cn1;
                    double f0 = 0.0, f1 = 0.0, f2 = 0.0;
                    if (S.nact == 2) {
                        f0 = fabs(r0 * g11) + fabs(r1 * g01);
                        f1 = fabs(r1 * g00) + fabs(r0 * g01);
                    } else if (S.nact == 3) {
                        const double B00 = fabs(g11 * g22) + g12 * g12, B01 = fabs(g02 * g12) + fabs(g01 * g22), B02 = fabs(g01 * g12) + fabs(g02 * g11);
                        const double B11 = fabs(g00 * g22) + g02 * g02, B12 = fabs(g01 * g02) + fabs(g00 * g12), B22 = fabs(g00 * g11) + g01 * g01;
                        f0 = B00 * fabs(r0) + B01 * fabs(r1) + B02 * fabs(r2);
                        f1 = B01 * fabs(r0) + B11 * fabs(r1) + B12 * fabs(r2);
                        f2 = B02 * fabs(r0) + B12 * fabs(r1) + B22 * fabs(r2);
                    }
                    const double a0 = -m0 * w0, a1 = S.nact >= 2 ? -m1 * w1 : 0.0, a2 = S.nact >= 3 ? -m2 * w2 : 0.0;
                    const double t0 = fmax(tol, LANE_M_NOISE * f0 * w0), t1 = fmax(tol, LANE_M_NOISE * f1 * w1), t2 = fmax(tol, LANE_M_NOISE * f2 * w2);
                    if (a0 >= -t0 && a1 >= -t1 && a2 >= -t2) {
                        S.status = ST_OPT;
                    } else {
                        // The row furthest below its tolerance goes.  The pass ends there: the next one forms the direction on
                        // the remaining rows with the care the top of the loop takes (second projection where the cost lies
                        // nearly in their span).  Written out here as  sum k_j n_j - dd c  it was, for a cost PARALLEL to a row
                        // that stays (a box LP against its own box row), what rounding leaves of a zero vector -- 1e-15 long,
                        // and the walk followed it for t = 1.8e15 to another facet (same polytope: box value 0.02 for -1.26).
                        int j = 0;
                        double am = a0 + t0;
                        if (a1 + t1 < am) { am = a1 + t1; j = 1; }
                        if (a2 + t2 < am) { am = a2 + t2; j = 2; }
                        if (j == 0) S.w0 = S.w1;
                        if (j <= 1) S.w1 = S.w2;
                        S.nact -= 1;
                        dropped = true;
                    }
                } else {
                    // a vertex: u_j = gcross(the other three rows) is orthogonal to them, N^-1[:, j] = u_j / (n_j.u_j),
                    // lambda_j = -(u_j.c) / (n_j.u_j); leaving row j: along -sgn(n_j.u_j) u_j
                    const double g33 = dot4(n30, n31, n32, n33, n30, n31, n32, n33);
                    const double w3 = 0.5 * (1.0 + g33);
                    double am = 0.0, bd0 = 0.0, bd1 = 0.0, bd2 = 0.0, bd3 = 0.0, dj_abs = 0.0;
                    int jbest = -1;
                    bool all_ok = true, singular = false;
#define PLP_W4_VERTEX(J, P, Q, R, NJ0, NJ1, NJ2, NJ3, WJ)                                                          \
                    {                                                                                                   \
                        double u0, u1, u2, u3;                                                                          \
                        gcross4(P##0, P##1, P##2, P##3, Q##0, Q##1, Q##2, Q##3, R##0, R##1, R##2, R##3, u0, u1, u2, u3); \
                        const double dj = dot4(NJ0, NJ1, NJ2, NJ3, u0, u1, u2, u3);                                     \
                        const double sg = dj > 0.0 ? 1.0 : -1.0;                                                        \
                        const double lj = -sg * dot4(c0, c1, c2, c3, u0, u1, u2, u3);                                   \
                        const double aj = lj * (WJ);                                                                    \
                        dj_abs = fabs(dj);                                                                              \
                        singular = singular | !(dj * dj > 1e-18 * (g00 * g11 * g22 * g33));                            \
                        all_ok = all_ok & (aj >= -(LANE_TOL_D * dj_abs * cn1));                                              \
                        const bool better = (jbest < 0) | (aj < am);                                                    \
                        am = better ? aj : am;                                                                          \
                        jbest = better ? (J) : jbest;                                                                   \
                        bd0 = better ? -sg * u0 : bd0;                                                                  \
                        bd1 = better ? -sg * u1 : bd1;                                                                  \
                        bd2 = better ? -sg * u2 : bd2;                                                                  \
                        bd3 = better ? -sg * u3 : bd3;                                                                  \
                    }
                    PLP_W4_VERTEX(0, n1, n2, n3, n00, n01, n02, n03, w0)
                    PLP_W4_VERTEX(1, n0, n2, n3, n10, n11, n12, n13, w1)
                    PLP_W4_VERTEX(2, n0, n1, n3, n20, n21, n22, n23, w2)
                    PLP_W4_VERTEX(3, n0, n1, n2, n30, n31, n32, n33, w3)
#undef PLP_W4_VERTEX
                    if (singular) S.status = ST_RETRY;   // four active planes that (nearly) share a line
                    else if (all_ok) S.status = ST_OPT;
                    else {
                        d0 = bd0; d1 = bd1; d2 = bd2; d3 = bd3;
                        if (jbest == 0) S.w0 = S.w1;
                        if (jbest <= 1) S.w1 = S.w2;
                        if (jbest <= 2) S.w2 = S.w3;
                        S.nact = 3;
                    }
                }
            }
        }
        if (dropped) {   // (counts as a pass: a chain of drops ends like a chain of steps)
            ++S.iters;
            if (S.iters >= LANE_MAX_ITERS) S.status = ST_RETRY;
        }
        const bool step = (S.status < 0) & !dropped;
        const double dn1 = fabs(d0) + fabs(d1) + fabs(d2) + fabs(d3);
        const double tolp = LANE_TOL_PIV * dn1;
        double bs = 1.0, bd = 0.0;
        int bi = -1;
        if (ANY(step)) RATIO(d0, d1, d2, d3, S.x0, S.x1, S.x2, S.x3, tolp, bs, bd, bi);
        if (step) {
            ++S.iters;
            if (bi < 0) {
                S.status = ST_UNBND;
            } else {
                const double t = bs / bd;
                S.x0 = fma(t, d0, S.x0);
                S.x1 = fma(t, d1, S.x1);
                S.x2 = fma(t, d2, S.x2);
                S.x3 = fma(t, d3, S.x3);
                if (S.nact == 0) S.w0 = bi;
                else if (S.nact == 1) S.w1 = bi;
                else if (S.nact == 2) S.w2 = bi;
                else S.w3 = bi;
                S.nact += 1;
                S.ndeg = (t * dn1 <= DEGEN_EPS) ? S.ndeg + 1 : 0;
                if (S.ndeg >= BLAND_AFTER || S.iters >= LANE_MAX_ITERS) S.status = ST_RETRY;
            }
        }
    }
}

PLP_LANE_FN void ratio_row4(const double a0, const double a1, const double a2, const double a3, const double beta, const int i,
                            const double d0, const double d1, const double d2, const double d3, const double x0,
                            const double x1, const double x2, const double x3, const double tolp, double& bs, double& bd,
                            int& bi) {
    const double ad = dot4(a0, a1, a2, a3, d0, d1, d2, d3);
    const double ax = dot4(a0, a1, a2, a3, x0, x1, x2, x3);
    const double sl = fmax(beta - ax, 0.0);
    const bool better = (ad > tolp) & (sl * bd < bs * ad);
    bs = better ? sl : bs;
    bd = better ? ad : bd;
    bi = better ? i : bi;
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

// ... from x' = 0: the slack is beta itself
PLP_LANE_FN void ratio_row0(const double a0, const double a1, const double a2, const double beta, const int i, const double d0,
                            const double d1, const double d2, const double tolp, double& bs, double& bd, int& bi) {
    const double ad = dot3(a0, a1, a2, d0, d1, d2);
    const double sl = fmax(beta, 0.0);
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
                        AnyF ANY, PassF PASS, const bool warm = false) {
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
          ANY, warm);
}

template <int M, class RowF, class BetaF, class AnyF>
PLP_LANE_FN void solve3(Lp3& S, const double c0, const double c1, const double c2, const bool go, RowF ROWS, BetaF BETA,
                        AnyF ANY, const bool warm = false) {
    solve3<M>(S, c0, c1, c2, go, ROWS, BETA, ANY, [] {}, warm);
}

}  // namespace lane
}  // namespace plp
