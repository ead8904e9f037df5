/*
 * plp_oracle_q.c -- TEST INFRASTRUCTURE ONLY (part of libplp_oracle.so, see plp_oracle.c).
 *
 * The same LP (polytope/solvers.py:76-106, :149-158: min c'x s.t. Gx <= h, x free) solved by the same textbook
 * method as plpo_lp_solve -- two-phase primal simplex in dictionary form, free variables enter and never leave,
 * Dantzig pricing, Bland's rule after BLAND_AFTER degenerate pivots -- but in IEEE binary128 (__float128, 113-bit
 * significand, libquadmath), on the equilibrated LP, with the tolerances of the certificate (Q_TOL_*): the dictionary
 * carries no rounding of its own at the level of those tolerances whatever the condition number of the bases it
 * passes through, so the vertex it stops at is a certified optimum of the LP as given.  It is the ARBITER of the round-6
 * parity work:
 *   - plpo_lp_solve hands an LP to it when the double engine's answer fails its a-posteriori certificate
 *     (plp_oracle.c: lp_certify), so that the oracle never returns an unverified answer;
 *   - tests/golden/make_golden_bbox_dup.py records it next to the reference's (HiGHS) answer, so that a fixture says
 *     per case how far the REFERENCE is from the exact optimum (HiGHS works to a 1e-7 feasibility tolerance);
 *   - the soaks (scripts/soak_*.py) ask it when kernel and oracle differ.
 * Status semantics are the double engine's: infeasible <=> the phase-1 optimum exceeds TOL_FEAS (1e-7, HiGHS's primal
 * feasibility tolerance -- the reference accepts such LPs as feasible); unbounded <=> no blocking row.
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <string.h>

typedef __float128 qreal;

#define QMAXM 256
#define QMAXN 18
/* Tolerances on the EQUILIBRATED dictionary (rows scaled to |G_i|_inf = 1, cost to |c|_inf = 1): the semantics of the
 * certificate in plp_oracle.c (lp_certify: dual 1e-9, primal 1e-13), not "exact".  An exact optimum is NOT what the
 * reference computes on data with rows an ulp apart: a copy of the row x_0 <= 2 tilted by 1e-16 towards x_1 lets the
 * exact LP reach x_0 = 2.84 at x_1 = 1e16 (scripts/soak_lane.py family `dup`, seed 5 trial 6) where HiGHS and every
 * double-precision code answer 2; a rate of change of the objective below 1e-9 per unit of distance is zero for all of them
 * (Q_TOL_D = the double engine's TOL_D; tests/golden/g22 case 162: at 1e-12 this engine gave a sliver tilted by 5e-11 the
 * Chebyshev radius 1.5, at a centre 6e10 away, where HiGHS says 2.5e-6).  What binary128 buys
 * is that the dictionary carries no rounding of its own at that level, whatever the basis' condition number. */
#define Q_TOL_D 1e-9Q
#define Q_TOL_NOISE 1e-13Q   /* reduced costs between this and Q_TOL_D: judged by what they buy (qrun) */
#define Q_TOL_GAIN 1e-10Q    /* ... an improvement above this (of max(1, |objective|)) within range */
#define Q_TOL_PIV 1e-12Q
#define Q_TOL_FEAS 1e-7Q
#define Q_DEGEN 1e-24Q
#define Q_BLAND_AFTER 6
enum { Q_OPT = 0, Q_ITER = 1, Q_INFEAS = 2, Q_UNBND = 3, Q_NUM = 4 };
#define Q_ID_T (-1)

typedef struct {
    int m, n, nc;
    qreal T[QMAXM][QMAXN], beta[QMAXM], cost[QMAXN], negz, cost2[QMAXN], negz2;
    int carry, rowvar[QMAXM], colvar[QMAXN], rowsgn[QMAXM], colsgn[QMAXN], rowact[QMAXM], coldead[QMAXN];
    int iters, maxit;
} qdict_t;
#define QFREE(D, id) ((unsigned)(id) < (unsigned)(D)->n)

static void qpivot(qdict_t *D, int r, int e)
{
    const int nc = D->nc, m = D->m;
    qreal rho[QMAXN], rhob;
    const qreal p = 1.0Q / D->T[r][e];
    for (int j = 0; j < nc; ++j) rho[j] = D->T[r][j] * p;
    rho[e] = p;
    rhob = D->beta[r] * p;
    for (int i = 0; i < m; ++i) {
        if (i == r) continue;
        const qreal f = D->T[i][e];
        if (f == 0.0Q) continue;
        D->T[i][e] = 0.0Q;
        for (int j = 0; j < nc; ++j) D->T[i][j] -= f * rho[j];
        D->beta[i] -= f * rhob;
    }
    {
        const qreal f = D->cost[e];
        D->cost[e] = 0.0Q;
        for (int j = 0; j < nc; ++j) D->cost[j] -= f * rho[j];
        D->negz -= f * rhob;
    }
    if (D->carry) {
        const qreal f = D->cost2[e];
        D->cost2[e] = 0.0Q;
        for (int j = 0; j < nc; ++j) D->cost2[j] -= f * rho[j];
        D->negz2 -= f * rhob;
    }
    for (int j = 0; j < nc; ++j) D->T[r][j] = rho[j];
    D->beta[r] = rhob;
    const int vin = D->colvar[e], vout = D->rowvar[r], sin_ = D->colsgn[e], sout = D->rowsgn[r];
    D->rowvar[r] = vin;  D->rowsgn[r] = sin_;
    D->colvar[e] = vout; D->colsgn[e] = sout;
    D->rowact[r] = !QFREE(D, vin);
    D->iters++;
}

/* ratio test of column e (entering upwards; negate: downwards): the leaving row (-1: none) and the step */
static int qratio(const qdict_t *D, int e, int negate, int bland, qreal *step)
{
    int r = -1;
    qreal rmin = 0.0Q;
    for (int i = 0; i < D->m; ++i) {
        if (!D->rowact[i]) continue;
        const qreal a = negate ? -D->T[i][e] : D->T[i][e];
        if (!(a > Q_TOL_PIV)) continue;
        const qreal bi = D->beta[i] > 0.0Q ? D->beta[i] : 0.0Q;
        const qreal q = bi / a;
        if (r < 0 || q < rmin || (bland && q == rmin && D->rowvar[i] < D->rowvar[r])) { rmin = q; r = i; }
    }
    *step = rmin;
    return r;
}

/* bigstep: PLPO_BIG times the scale of the data in the units of the equilibrated dictionary (how far a step may go and count) */
static int qrun(qdict_t *D, double bigstep)
{
    int ndeg = 0;
    for (;;) {
        const int bland = (ndeg >= Q_BLAND_AFTER);
        int e = -1, bestid = 0x7fffffff;
        qreal best = 0.0Q;
        for (int j = 0; j < D->nc; ++j) {
            if (D->coldead[j]) continue;
            const qreal dj = D->cost[j], aj = fabsq(dj);
            const int elig = QFREE(D, D->colvar[j]) ? (aj > Q_TOL_D) : (dj < -Q_TOL_D);
            if (!elig) continue;
            if (bland) { if (D->colvar[j] < bestid) { bestid = D->colvar[j]; e = j; } }
            else if (aj > best) { best = aj; e = j; }
        }
        if (e < 0) {
            /* no column above the engines' tolerance.  Those between the rounding level and it are judged by what they buy:
             * the step they allow times the rate, against Q_TOL_GAIN of max(1, |objective|), provided the step ends on a row
             * and moves x by no more than bigstep
             * (polytope_amd/csrc/plp_verify.hpp: careful_run, the same rule) */
            const qreal obj = fabsq(D->negz);
            const qreal thr = Q_TOL_GAIN * (obj > 1.0Q ? obj : 1.0Q);
            qreal gain = 0.0Q;
            for (int j = 0; j < D->nc; ++j) {
                if (D->coldead[j]) continue;
                const qreal dj = D->cost[j], aj = fabsq(dj);
                const int grey = QFREE(D, D->colvar[j]) ? (aj > Q_TOL_NOISE) : (dj < -Q_TOL_NOISE);
                if (!grey) continue;
                qreal step;
                const int r = qratio(D, j, dj > 0.0Q, 0, &step);
                if (r < 0) continue;   /* unbounded at a rate below the engines' tolerance: not a direction (HiGHS agrees) */
                /* how far x moves per unit of the entering variable: the rows that hold the free variables (and itself, if free) */
                qreal dx = QFREE(D, D->colvar[j]) ? 1.0Q : 0.0Q;
                for (int i = 0; i < D->m; ++i)
                    if (QFREE(D, D->rowvar[i]) && fabsq(D->T[i][j]) > dx) dx = fabsq(D->T[i][j]);
                if (!(step * dx <= (qreal)bigstep)) continue;   /* the step leaves the range of the data: not taken either */
                const qreal gj = aj * step;
                if (gj > thr && gj > gain) { gain = gj; e = j; }
            }
            if (e < 0) return Q_OPT;
        }
        if (D->iters >= D->maxit) return Q_ITER;
        if (D->cost[e] > 0.0Q) {
            for (int i = 0; i < D->m; ++i) D->T[i][e] = -D->T[i][e];
            D->cost[e] = -D->cost[e];
            if (D->carry) D->cost2[e] = -D->cost2[e];
            D->colsgn[e] = -D->colsgn[e];
        }
        qreal rmin;
        const int r = qratio(D, e, 0, bland, &rmin);
        if (r < 0) return Q_UNBND;
        ndeg = (rmin <= Q_DEGEN) ? ndeg + 1 : 0;
        qpivot(D, r, e);
    }
}

/* basis[n] (optional): the nonbasic variables of the final dictionary -- row index i >= 0 for the slack of row i
 * (the row is active), -1 - j for the free variable x_j left at zero.  Filled for status 0 only. */
int plpo_lp_solve_q(int m, int n, const double *c, const double *G, const double *h,
                    double *x, double *fun, int *iters, int *basis)
{
    static qdict_t D;   /* (not re-entrant across threads: the oracle's callers are processes) */
    qdict_t *d = &D;
    if (iters) *iters = 0;
    for (int j = 0; j < n; ++j) x[j] = NAN;
    *fun = NAN;
    if (m > QMAXM || n > QMAXN - 1 || n < 1 || m < 0) return Q_NUM;
    for (int j = 0; j < n; ++j) if (!isfinite(c[j])) return Q_NUM;
    for (int i = 0; i < m; ++i) {
        if (!isfinite(h[i])) return Q_NUM;
        for (int j = 0; j < n; ++j) if (!isfinite(G[i * n + j])) return Q_NUM;
    }
    memset(d, 0, sizeof(*d));
    d->m = m; d->n = n; d->nc = n;
    for (int i = 0; i < m; ++i) { d->rowvar[i] = n + i; d->rowsgn[i] = 1; d->rowact[i] = 1; }
    for (int j = 0; j < QMAXN; ++j) { d->colvar[j] = j; d->colsgn[j] = 1; }
    d->maxit = 200 * (m + n) + 1000;
    int need_p1 = 0;
    for (int i = 0; i < m; ++i) {
        double gmax = 0.0;
        for (int j = 0; j < n; ++j) if (fabs(G[i * n + j]) > gmax) gmax = fabs(G[i * n + j]);
        if (!(gmax > 0.0)) {   /* 0 <= h_i: vacuous or infeasible (as the double engine) */
            if (h[i] < -1e-7) return Q_INFEAS;
            d->rowact[i] = 0; d->beta[i] = 0.0Q;
            continue;
        }
        const qreal sc = 1.0Q / (qreal)gmax;   /* row equilibration: the slack of row i in units of |G_i|_inf */
        for (int j = 0; j < n; ++j) d->T[i][j] = (qreal)G[i * n + j] * sc;
        d->beta[i] = (qreal)h[i] * sc;
        if (h[i] < 0.0) need_p1 = 1;
    }
    double cmax = 0.0;
    for (int j = 0; j < n; ++j) if (fabs(c[j]) > cmax) cmax = fabs(c[j]);
    for (int j = 0; j < n; ++j) d->cost[j] = cmax > 0.0 ? (qreal)c[j] / (qreal)cmax : 0.0Q;
    double hs = 1.0;   /* max(1, max_i |h_i| / |G_i|_inf): plp_oracle.c's lp_scale without the cost */
    for (int i = 0; i < m; ++i) {
        double gmax = 0.0;
        for (int j = 0; j < n; ++j) if (fabs(G[i * n + j]) > gmax) gmax = fabs(G[i * n + j]);
        if (gmax > 0.0 && fabs(h[i]) > hs * gmax) hs = fabs(h[i]) / gmax;
    }
    const double bigstep = 1e9 * hs;
    int st;
    if (need_p1) {
        const int tc = n;
        d->nc = n + 1;
        d->colvar[tc] = Q_ID_T;
        for (int j = 0; j < n; ++j) { d->cost2[j] = d->cost[j]; d->cost[j] = 0.0Q; }
        d->cost2[tc] = 0.0Q; d->cost[tc] = 1.0Q; d->carry = 1;
        int r0 = -1;
        for (int i = 0; i < m; ++i) {
            if (!d->rowact[i]) continue;
            d->T[i][tc] = -1.0Q;
            if (r0 < 0 || d->beta[i] < d->beta[r0]) r0 = i;
        }
        qpivot(d, r0, tc);
        st = qrun(d, bigstep);
        if (st != Q_OPT) { if (iters) *iters = d->iters; return st == Q_ITER ? Q_ITER : Q_NUM; }
        int rt = -1, ct = -1;
        for (int i = 0; i < m; ++i) if (d->rowvar[i] == Q_ID_T) rt = i;
        for (int j = 0; j < d->nc; ++j) if (d->colvar[j] == Q_ID_T) ct = j;
        if (rt >= 0) {
            if (d->beta[rt] > Q_TOL_FEAS) { if (iters) *iters = d->iters; return Q_INFEAS; }
            int e = -1; qreal big = Q_TOL_PIV;
            for (int j = 0; j < d->nc; ++j)
                if (fabsq(d->T[rt][j]) > big) { big = fabsq(d->T[rt][j]); e = j; }
            if (e >= 0) {
                qpivot(d, rt, e);
                if (!QFREE(d, d->rowvar[rt]) && d->beta[rt] < 0.0Q) d->beta[rt] = 0.0Q;
                ct = e;
            } else d->rowact[rt] = 0;
        }
        if (ct >= 0) d->coldead[ct] = 1;
        for (int i = 0; i < m; ++i) if (d->rowact[i] && d->beta[i] < 0.0Q) d->beta[i] = 0.0Q;
        for (int j = 0; j < d->nc; ++j) d->cost[j] = d->cost2[j];
        d->negz = d->negz2; d->carry = 0;
    }
    st = qrun(d, bigstep);
    if (iters) *iters = d->iters;
    if (st != Q_OPT) return st;
    qreal xq[QMAXN];
    for (int j = 0; j < n; ++j) xq[j] = 0.0Q;
    for (int i = 0; i < m; ++i)
        if (QFREE(d, d->rowvar[i])) xq[d->rowvar[i]] = d->rowsgn[i] * d->beta[i];
    qreal f = 0.0Q;
    for (int j = 0; j < n; ++j) { f += (qreal)c[j] * xq[j]; x[j] = (double)xq[j]; }
    *fun = (double)f;
    if (basis) {
        int k = 0;
        for (int j = 0; j < d->nc && k < n; ++j) {
            if (d->coldead[j] || d->colvar[j] == Q_ID_T) continue;
            basis[k++] = QFREE(d, d->colvar[j]) ? -1 - d->colvar[j] : d->colvar[j] - n;
        }
        while (k < n) basis[k++] = -1000;
    }
    return Q_OPT;
}

/* bounding_box (polytope.py:1367-1409) through the binary128 engine: lb/ub, +-inf on status 3, the reference's
 * l = 0 / u = l on status 2 (:1378-1402).  Returns 0 or the offending status (1 / 4). */
int plpo_bounding_box_q(int m, int d, const double *A, const double *b, double *lb, double *ub)
{
    double c[QMAXN], x[QMAXN], fun;
    int bad = 0;
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < d; ++i) {
            for (int k = 0; k < d; ++k) c[k] = 0.0;
            c[i] = s ? -1.0 : 1.0;
            const int st = plpo_lp_solve_q(m, d, c, A, b, x, &fun, NULL, NULL);
            double *o = s ? ub : lb;
            if (st == Q_OPT) o[i] = x[i];
            else if (st == Q_UNBND) o[i] = s ? INFINITY : -INFINITY;
            else if (st == Q_INFEAS) o[i] = s ? lb[i] : 0.0;
            else { o[i] = NAN; bad = st; }
        }
    return bad;
}
