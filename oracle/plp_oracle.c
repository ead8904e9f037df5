/*
 * plp_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, scalar, FP64)
 * of the batched small-LP hot path of tulip-control/polytope.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (polytope_amd/) never imports, links or calls it.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *
 *   plpo_lp_solve      polytope/solvers.py:76-106 (lpsolve) and :149-158
 *                      (_solve_lp_using_scipy):  min c'x  s.t. Gx <= h, x free,
 *                      status codes 0/1/2/3/4 as scipy.optimize.linprog.
 *                      The arithmetic of the reference lives in a third-party
 *                      dependency that is NOT under /root/reference:
 *                      scipy.optimize.linprog -> HiGHS (requirements/default.txt pins
 *                      scipy==1.10.0; pyproject.toml:28-32 scipy>=1.10.0; this image has
 *                      scipy 1.15.3).  HiGHS is a (dual revised) simplex code; the
 *                      optimal value of an LP is unique, so what is restated here is the
 *                      published textbook algorithm -- the two-phase primal simplex
 *                      method in dictionary form (Chvatal, "Linear Programming", ch. 2-3,
 *                      8: auxiliary variable x0 for phase 1, free variables enter and
 *                      never leave) with Dantzig pricing and Bland's anti-cycling rule.
 *                      Parity is pinned on golden vectors generated from the imported
 *                      reference (tests/golden/make_golden.py).
 *   plpo_cheby         polytope/polytope.py:1241-1300 (cheby_ball)       LP form F1
 *   plpo_bounding_box  polytope/polytope.py:1314-1411 (bounding_box)     LP form F3
 *   plpo_reduce        polytope/polytope.py:1053-1163 (reduce)           LP form F2
 *   plpo_reduce_batch  the same, looped over a packed batch (full-batch parity checks)
 *   plpo_contains      polytope/polytope.py:206-218, :732-746 (contains)
 *   plpo_assign        polytope/quickhull.py:117-121 (distance), :224-245 / :311-336
 *                      (outside-set assignment), :87-102 (get_furthest)
 *   plpo_hull_reassign polytope/quickhull.py:273-283, :311-336, :87-102 (one iteration of the
 *                      main loop: pooling, re-assignment, furthest point) on index arrays
 *
 * Pinned: every function is checked against the fixtures tests/golden/g1..g10 (outputs of the
 * imported reference on seeded inputs, tests/test_oracle_golden.py, tests/test_quickhull.py); g5, g9 and
 * g11 pin the Python layer above it (tests/test_python_api.py).
 *
 * The pivot rules, tolerances and operation order of plpo_lp_solve are the ones the HIP
 * kernels use (polytope_amd/csrc/plp_simplex.hpp) so that CPU and GPU walk the same
 * vertex path; results are nevertheless compared with a tolerance (1e-9), never bitwise.
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PLPO_MAXM 256 /* rows of one LP (the HIP engine goes past 64 rows with its LDS-resident dictionary) */
#define PLPO_MAXM_RED PLPO_MAXM /* rows of a polytope handed to plpo_reduce: the keep mask is PLPO_KEEP_WORDS 64-bit words */
#define PLPO_KEEP_WORDS ((PLPO_MAXM + 63) / 64)
#define PLPO_MAXN 18 /* d+1 structural columns (<=17) + phase-1 artificial */

#define TOL_D 1e-9     /* reduced-cost (dual feasibility) tolerance            */
#ifndef TOL_PIV
#define TOL_PIV 1e-7   /* smallest admissible pivot element (1e-9 until round 5: see polytope_amd/csrc/plp_common.hpp) */
#endif
#define TOL_FEAS 1e-7  /* phase-1 infeasibility accepted (HiGHS primal tol)    */
#define DEGEN_EPS 1e-12 /* step length regarded as degenerate                  */
#define BLAND_AFTER 6  /* consecutive degenerate pivots before Bland's rule    */

enum { ST_OPT = 0, ST_ITER = 1, ST_INFEAS = 2, ST_UNBND = 3, ST_NUM = 4 };

/* Dictionary:  basic_i = beta[i] - sum_j T[i][j] * nb_j      (i < m, j < nc)
 *              (-zeta) = negz    - sum_j cost[j] * nb_j      (minimise zeta)
 * A second cost row (cost2/negz2) can be carried through the pivots (phase 1). */
typedef struct {
    int m, n, nc;
    double T[PLPO_MAXM][PLPO_MAXN];
    double beta[PLPO_MAXM];
    double cost[PLPO_MAXN], negz;
    double cost2[PLPO_MAXN], negz2;
    int carry;
    int rowvar[PLPO_MAXM]; /* id of the basic variable of row i            */
    int colvar[PLPO_MAXN]; /* id of the nonbasic variable of column j      */
    int rowsgn[PLPO_MAXM]; /* +-1 for a basic free variable                */
    int colsgn[PLPO_MAXN];
    int rowact[PLPO_MAXM]; /* row takes part in the ratio test             */
    int coldead[PLPO_MAXN];
    int iters, maxit;
    int unb_col;           /* run(): the entering column when it returned ST_UNBND */
} dict_t;

/* ids: 0..n-1 structural (free) x_j; n..n+m-1 slack of row i; -1 artificial t
 * (lowest id, so that ties in the ratio test let t leave first, Chvatal p.41) */
#define ID_T (-1)
#define ISFREE(D, id) ((unsigned)(id) < (unsigned)(D)->n)

static void pivot(dict_t *D, int r, int e)
{
    const int nc = D->nc, m = D->m;
    double rho[PLPO_MAXN], rhob;
    const double p = 1.0 / D->T[r][e];
    for (int j = 0; j < nc; ++j) rho[j] = D->T[r][j] * p;
    rho[e] = p;
    rhob = D->beta[r] * p;
    for (int i = 0; i < m; ++i) {
        if (i == r) continue;
        const double f = D->T[i][e];
        D->T[i][e] = 0.0;
        for (int j = 0; j < nc; ++j) D->T[i][j] = fma(-f, rho[j], D->T[i][j]);
        D->beta[i] = fma(-f, rhob, D->beta[i]);
    }
    {
        const double f = D->cost[e];
        D->cost[e] = 0.0;
        for (int j = 0; j < nc; ++j) D->cost[j] = fma(-f, rho[j], D->cost[j]);
        D->negz = fma(-f, rhob, D->negz);
    }
    if (D->carry) {
        const double f = D->cost2[e];
        D->cost2[e] = 0.0;
        for (int j = 0; j < nc; ++j) D->cost2[j] = fma(-f, rho[j], D->cost2[j]);
        D->negz2 = fma(-f, rhob, D->negz2);
    }
    for (int j = 0; j < nc; ++j) D->T[r][j] = rho[j];
    D->beta[r] = rhob;
    /* swap variable bookkeeping */
    const int vin = D->colvar[e], vout = D->rowvar[r];
    const int sin_ = D->colsgn[e], sout = D->rowsgn[r];
    D->rowvar[r] = vin;  D->rowsgn[r] = sin_;
    D->colvar[e] = vout; D->colsgn[e] = sout;
    D->rowact[r] = !ISFREE(D, vin); /* a free variable never leaves again */
    D->iters++;
}

/* Primal simplex from a primal-feasible dictionary.  Returns ST_OPT/ST_UNBND/ST_ITER. */
static int run(dict_t *D)
{
    int ndeg = 0;
    for (;;) {
        const int bland = (ndeg >= BLAND_AFTER);
        int e = -1;
        double best = 0.0;
        int bestid = 0x7fffffff;
        for (int j = 0; j < D->nc; ++j) {
            if (D->coldead[j]) continue;
            const double dj = D->cost[j];
            const int isfree = ISFREE(D, D->colvar[j]);
            const int elig = isfree ? (fabs(dj) > TOL_D) : (dj < -TOL_D);
            if (!elig) continue;
            if (bland) {
                if (D->colvar[j] < bestid) { bestid = D->colvar[j]; e = j; }
            } else {
                if (fabs(dj) > best) { best = fabs(dj); e = j; }
            }
        }
        if (e < 0) return ST_OPT;
        if (D->iters >= D->maxit) return ST_ITER;
        if (D->cost[e] > 0.0) { /* free variable entering downwards: x := -x */
            for (int i = 0; i < D->m; ++i) D->T[i][e] = -D->T[i][e];
            D->cost[e] = -D->cost[e];
            if (D->carry) D->cost2[e] = -D->cost2[e];
            D->colsgn[e] = -D->colsgn[e];
        }
        /* ratio test: min beta_i/T_ie over active rows with T_ie > TOL_PIV */
        int r = -1;
        double rmin = INFINITY;
        for (int i = 0; i < D->m; ++i) {
            if (!D->rowact[i]) continue;
            const double a = D->T[i][e];
            if (!(a > TOL_PIV)) continue;
            const double bi = D->beta[i] > 0.0 ? D->beta[i] : 0.0;
            const double q = bi * (1.0 / a);
            /* ties: Dantzig mode keeps the first (lowest) row, Bland mode the lowest variable id */
            if (r < 0 || q < rmin || (bland && q == rmin && D->rowvar[i] < D->rowvar[r])) { rmin = q; r = i; }
        }
        if (r < 0) { D->unb_col = e; return ST_UNBND; }
        ndeg = (rmin <= DEGEN_EPS) ? ndeg + 1 : 0;
        pivot(D, r, e);
    }
}

static void dict_init(dict_t *D, int m, int n)
{
    /* clear the scalars, the cost rows and the first m rows only (the struct is sized for PLPO_MAXM rows) */
    memset(D->T, 0, (size_t)m * sizeof(D->T[0]));
    memset(D->beta, 0, (size_t)m * sizeof(double));
    memset(D->cost, 0, sizeof(D->cost)); memset(D->cost2, 0, sizeof(D->cost2));
    memset(D->coldead, 0, sizeof(D->coldead));
    D->negz = D->negz2 = 0.0; D->carry = 0; D->iters = 0; D->unb_col = -1;
    D->m = m; D->n = n; D->nc = n;
    for (int i = 0; i < m; ++i) { D->rowvar[i] = n + i; D->rowsgn[i] = 1; D->rowact[i] = 1; }
    for (int j = 0; j < PLPO_MAXN; ++j) { D->colvar[j] = j; D->colsgn[j] = 1; }
    D->maxit = 50 * (m + n) + 100;
}

static void extract_x(const dict_t *D, double *x)
{
    for (int j = 0; j < D->n; ++j) x[j] = 0.0;
    for (int i = 0; i < D->m; ++i)
        if (ISFREE(D, D->rowvar[i])) x[D->rowvar[i]] = D->rowsgn[i] * D->beta[i];
}

/* min c'x s.t. Gx<=h, x free.  G is m x n row-major.  x/fun written only for status 0
 * (NaN otherwise).  solvers.py:149-158 semantics.  The double-precision dictionary simplex on its own: what the
 * fused reduce (plpo_reduce) calls, and the first stage of plpo_lp_solve below.
 * basis (optional, n + 2 ints; status 0 and 3): the n nonbasic variables of the final dictionary -- i >= 0: the slack of
 * row i (the row is active), -1 - j: the free variable x_j left at zero --, then for status 3 the position (in that
 * list) of the variable along which the LP is unbounded and the sign with which it moves. */
int plpo_lp_solve_raw(int m, int n, const double *c, const double *G, const double *h,
                      double *x, double *fun, int *iters, int *basis)
{
    static const double qnan = NAN;
    dict_t D;
    int st;
    if (iters) *iters = 0;
    for (int j = 0; j < n; ++j) x[j] = qnan;
    *fun = qnan;
    if (m > PLPO_MAXM || n > PLPO_MAXN - 1 || n < 1 || m < 0) return ST_NUM;
    for (int j = 0; j < n; ++j) if (!isfinite(c[j])) return ST_NUM;
    for (int i = 0; i < m; ++i) {
        if (!isfinite(h[i])) return ST_NUM;
        for (int j = 0; j < n; ++j) if (!isfinite(G[i * n + j])) return ST_NUM;
    }
    dict_init(&D, m, n);
    int need_p1 = 0;
    for (int i = 0; i < m; ++i) {
        int zero = 1;
        for (int j = 0; j < n; ++j) { D.T[i][j] = G[i * n + j]; if (G[i * n + j] != 0.0) zero = 0; }
        D.beta[i] = h[i];
        if (zero) { /* 0 <= h_i : vacuous or infeasible */
            if (h[i] < -TOL_FEAS) return ST_INFEAS;
            D.rowact[i] = 0; D.beta[i] = 0.0;
            continue;
        }
        if (h[i] < 0.0) need_p1 = 1;
    }
    for (int j = 0; j < n; ++j) D.cost[j] = c[j];
    if (need_p1) {
        /* auxiliary problem: min t  s.t. Gx - t <= h, t >= 0 (Chvatal's x0) */
        const int tc = n;
        D.nc = n + 1;
        D.colvar[tc] = ID_T;
        for (int j = 0; j < n; ++j) { D.cost2[j] = D.cost[j]; D.cost[j] = 0.0; }
        D.cost2[tc] = 0.0; D.cost[tc] = 1.0; D.carry = 1;
        int r0 = -1;
        for (int i = 0; i < m; ++i) {
            if (!D.rowact[i]) continue;
            D.T[i][tc] = -1.0;
            if (r0 < 0 || D.beta[i] < D.beta[r0]) r0 = i;
        }
        pivot(&D, r0, tc);
        st = run(&D);
        if (st != ST_OPT) { if (iters) *iters = D.iters; return st == ST_ITER ? ST_ITER : ST_NUM; }
        int rt = -1, ct = -1;
        for (int i = 0; i < m; ++i) if (D.rowvar[i] == ID_T) rt = i;
        for (int j = 0; j < D.nc; ++j) if (D.colvar[j] == ID_T) ct = j;
        if (rt >= 0) {
            if (D.beta[rt] > TOL_FEAS) { if (iters) *iters = D.iters; return ST_INFEAS; }
            /* drive t out of the basis (degenerate pivot on the largest element) */
            int e = -1; double big = TOL_PIV;
            for (int j = 0; j < D.nc; ++j)
                if (fabs(D.T[rt][j]) > big) { big = fabs(D.T[rt][j]); e = j; }
            if (e >= 0) {
                pivot(&D, rt, e);
                if (!ISFREE(&D, D.rowvar[rt]) && D.beta[rt] < 0.0) D.beta[rt] = 0.0;
                ct = e;
            } else {
                D.rowact[rt] = 0; /* 0 = t : redundant row */
            }
        }
        if (ct >= 0) D.coldead[ct] = 1;
        for (int i = 0; i < m; ++i) if (D.rowact[i] && D.beta[i] < 0.0) D.beta[i] = 0.0;
        for (int j = 0; j < D.nc; ++j) D.cost[j] = D.cost2[j];
        D.negz = D.negz2; D.carry = 0;
    }
    st = run(&D);
    if (iters) *iters = D.iters;
    if (basis && (st == ST_OPT || st == ST_UNBND)) {
        int k = 0;
        basis[n] = -1; basis[n + 1] = 0;
        for (int j = 0; j < D.nc && k < n; ++j) {
            if (D.coldead[j] || D.colvar[j] == ID_T) continue;
            if (st == ST_UNBND && j == D.unb_col) { basis[n] = k; basis[n + 1] = D.colsgn[j]; }
            basis[k++] = ISFREE(&D, D.colvar[j]) ? -1 - D.colvar[j] : D.colvar[j] - n;
        }
        while (k < n) basis[k++] = -1000;   /* (cannot happen: n live columns) */
    }
    if (st != ST_OPT) return st;
    extract_x(&D, x);
    double f = 0.0;
    for (int j = 0; j < n; ++j) f = fma(c[j], x[j], f);
    *fun = f;
    return ST_OPT;
}

/* ---------------------------------------------------------------------------------------------------------------
 * A-posteriori certificate of an LP answer (round 6).  The dictionary simplex above stops on ABSOLUTE tolerances
 * (reduced costs above -1e-9, pivots above 1e-7) applied to a dictionary that carries the rounding of every pivot
 * it went through.  On rows a hair apart (polytope.py's bounding_box, :1314-1411, has no dedupe in front of its LPs)
 * that is worth 1e-7 .. 1e-5 on a box of size 3, and sides that are finite come out infinite.  So the answer is
 * checked against the ORIGINAL rows, from the final basis alone:
 *   M = the n rows that define the vertex (active rows; e_j for a free variable left at zero),
 *   x = M^-1 rhs and y = -M^-T c by LU with partial pivoting + iterative refinement (residuals in binary128),
 *   primal:  h_i - G_i.x >= -2e-14 * max(|h_i|, |G_i|_inf * max(1, |x|_inf))   for every row,
 *   dual:    y_k |G_k|_inf >= -1e-13 |c|_inf on active rows, |y_k| <= 1e-13 |c|_inf on the free variables (a multiplier
 *            between that and the engine's own 1e-9 is judged by what it buys: plp_oracle_q.c, qrun)
 * -- an optimal basis of the LP as given, its vertex computed to the last bits whatever path led there.  An
 * unbounded answer is checked the same way (ray w = M^-1 (-/+ u_e): G_i.w <= 1e-12 |G_i| |w| for every row, c.w < 0).
 * What fails is solved again by plpo_lp_solve_q (binary128; plp_oracle_q.c): the oracle never returns an answer it
 * could not verify.  The HIP library does the same with its own verifier kernel and a double-double engine
 * (polytope_amd/csrc/plp_verify.hip). */
#include <quadmath.h>
int plpo_lp_solve_q(int m, int n, const double *c, const double *G, const double *h,
                    double *x, double *fun, int *iters, int *basis);

/* An LP whose optimum lies beyond PLPO_BIG times the scale of its data is UNBOUNDED for the reference: HiGHS takes no
 * pivot below ~1e-9 of a (scaled) column, so a row that stops a ray only that far out is no blocking row -- measured on
 * polytopes with two rows 1e-9 .. 1e-11 rad apart facing each other (scripts/soak_lane.py, family `dup`): box sides
 * whose exact value is 1e5 .. 1.5e9 come back from scipy.optimize.linprog as that value, beyond ~2e9 as +-inf, with
 * garbage in between (exact 2.4e9 -> 4.8e9).  The exact optimum decides, so the rule is independent of the path: both
 * this oracle and the HIP library apply it to the certified / re-solved value.  scale = |c|_inf * max(1, max_i
 * |h_i| / |G_i|_inf). */
#define PLPO_BIG 1e9
#define PLPO_FAR 1e4            /* a certificate counts at a vertex within this many times the data's scale: see plpo_lp_solve */
#define PLPO_SMALL_ENTRY 1e-9   /* HiGHS's small_matrix_value: see plpo_lp_solve */
#define PLPO_TOL_PRIMAL 2e-14   /* (1e-10 let through a vertex 1e-8 outside a twin row: 7e-7 on an optimum of 235) */
static double g_tol_dual = 1e-13;   /* a multiplier below this is rounding; between it and the engine's 1e-9: binary128 judges it by what it buys */
#define PLPO_TOL_DUAL g_tol_dual
void plpo_set_tol_dual(double t) { g_tol_dual = t; }
static double lp_scale(int m, int n, const double *c, const double *G, const double *h)
{
    double cmax = 0.0, hs = 1.0;
    for (int j = 0; j < n; ++j) if (fabs(c[j]) > cmax) cmax = fabs(c[j]);
    for (int i = 0; i < m; ++i) {
        double gmax = 0.0;
        for (int j = 0; j < n; ++j) if (fabs(G[i * n + j]) > gmax) gmax = fabs(G[i * n + j]);
        if (gmax > 0.0 && fabs(h[i]) > hs * gmax) hs = fabs(h[i]) / gmax;
    }
    return cmax * hs;
}

static int g_certify = 1;            /* plpo_set_certify(0): plpo_lp_solve = the raw double engine (experiments) */
static long g_cert_stat[4];          /* answers certified, handed to binary128, of those: status changed, - */
void plpo_set_certify(int on) { g_certify = on; }
void plpo_cert_stats(long *out) { for (int k = 0; k < 4; ++k) out[k] = g_cert_stat[k]; }

/* LU of the n x n matrix M (row-major, overwritten) with partial pivoting; perm[k] = row of M in position k.
 * Returns 0 when a pivot is below 1e-13 of the largest entry (singular to working precision). */
static int lu_factor(int n, double *M, int *perm)
{
    double big = 0.0;
    for (int k = 0; k < n * n; ++k) if (fabs(M[k]) > big) big = fabs(M[k]);
    if (!(big > 0.0)) return 0;
    for (int k = 0; k < n; ++k) perm[k] = k;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (fabs(M[i * n + k]) > fabs(M[p * n + k])) p = i;
        if (!(fabs(M[p * n + k]) > 1e-13 * big)) return 0;
        if (p != k) {
            for (int j = 0; j < n; ++j) { const double t = M[k * n + j]; M[k * n + j] = M[p * n + j]; M[p * n + j] = t; }
            const int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        const double inv = 1.0 / M[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = M[i * n + k] * inv;
            M[i * n + k] = f;
            for (int j = k + 1; j < n; ++j) M[i * n + j] = fma(-f, M[k * n + j], M[i * n + j]);
        }
    }
    return 1;
}
/* solve M z = r  (LU, perm from lu_factor) */
static void lu_solve(int n, const double *LU, const int *perm, const double *r, double *z)
{
    double t[PLPO_MAXN];
    for (int k = 0; k < n; ++k) {
        double s = r[perm[k]];
        for (int j = 0; j < k; ++j) s = fma(-LU[k * n + j], t[j], s);
        t[k] = s;
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = t[k];
        for (int j = k + 1; j < n; ++j) s = fma(-LU[k * n + j], z[j], s);
        z[k] = s / LU[k * n + k];
    }
}
/* solve M' z = r */
static void lu_solve_t(int n, const double *LU, const int *perm, const double *r, double *z)
{
    double t[PLPO_MAXN];
    for (int k = 0; k < n; ++k) {          /* U' t = r */
        double s = r[k];
        for (int j = 0; j < k; ++j) s = fma(-LU[j * n + k], t[j], s);
        t[k] = s / LU[k * n + k];
    }
    for (int k = n - 1; k >= 0; --k) {     /* L' w = t */
        double s = t[k];
        for (int j = k + 1; j < n; ++j) s = fma(-LU[j * n + k], t[j], s);
        t[k] = s;
    }
    for (int k = 0; k < n; ++k) z[perm[k]] = t[k];
}

/* z = M^-1 r with three rounds of refinement (residuals in binary128); TRANS: M' z = r */
static void lu_solve_refined(int n, const double *M0, const double *LU, const int *perm, const double *r, double *z, int trans)
{
    double rr[PLPO_MAXN], dz[PLPO_MAXN];
    if (trans) lu_solve_t(n, LU, perm, r, z); else lu_solve(n, LU, perm, r, z);
    for (int it = 0; it < 3; ++it) {
        for (int k = 0; k < n; ++k) {
            __float128 s = r[k];
            for (int j = 0; j < n; ++j) s -= (__float128)(trans ? M0[j * n + k] : M0[k * n + j]) * z[j];
            rr[k] = (double)s;
        }
        if (trans) lu_solve_t(n, LU, perm, rr, dz); else lu_solve(n, LU, perm, rr, dz);
        for (int j = 0; j < n; ++j) z[j] += dz[j];
    }
}

/* status 0 / 3 of the raw engine with its basis -> 1 and (status 0) the polished x, fun when the certificate holds */
#include <stdio.h>
static int g_cert_dbg = 0;
void plpo_set_cert_debug(int on) { g_cert_dbg = on; }
#define CERT_FAIL(code) do { if (g_cert_dbg) fprintf(stderr, "lp_certify: fail %d (m %d n %d status %d)\n", code, m, n, status); return 0; } while (0)
static int lp_certify(int m, int n, const double *c, const double *G, const double *h, const int *basis, int status,
                      double *x, double *fun)
{
    double M0[PLPO_MAXN * PLPO_MAXN], LU[PLPO_MAXN * PLPO_MAXN], rhs[PLPO_MAXN], z[PLPO_MAXN];
    int perm[PLPO_MAXN];
    if (n > PLPO_MAXN - 1) CERT_FAIL(1);
    for (int k = 0; k < n; ++k) {
        const int v = basis[k];
        if (v >= 0) {
            if (v >= m) CERT_FAIL(2);
            for (int j = 0; j < n; ++j) M0[k * n + j] = G[v * n + j];
            rhs[k] = h[v];
        } else {
            const int j0 = -1 - v;
            if (j0 < 0 || j0 >= n) CERT_FAIL(3);
            for (int j = 0; j < n; ++j) M0[k * n + j] = (j == j0) ? 1.0 : 0.0;
            rhs[k] = 0.0;
        }
    }
    memcpy(LU, M0, sizeof(double) * (size_t)n * n);
    if (!lu_factor(n, LU, perm)) CERT_FAIL(4);
    double cmax = 0.0;
    for (int j = 0; j < n; ++j) if (fabs(c[j]) > cmax) cmax = fabs(c[j]);
    /* the vertex */
    lu_solve_refined(n, M0, LU, perm, rhs, z, 0);
    double zmax = 0.0;
    for (int j = 0; j < n; ++j) { if (!isfinite(z[j])) CERT_FAIL(5); if (fabs(z[j]) > zmax) zmax = fabs(z[j]); }
    const double xs = zmax > 1.0 ? zmax : 1.0;
    if (status == ST_UNBND) {
        /* the vertex z the engine stood on and the ray w it left along: every row that w runs into must lie beyond
         * the point where the objective passes PLPO_BIG times the scale of the data (lp_scale) */
        const int e = basis[n];
        double w[PLPO_MAXN], ru[PLPO_MAXN];
        if (e < 0 || e >= n) CERT_FAIL(6);
        for (int k = 0; k < n; ++k) ru[k] = 0.0;
        ru[e] = basis[e] >= 0 ? -1.0 : (double)basis[n + 1];   /* the slack of an active row grows / the free variable moves by its sign */
        lu_solve_refined(n, M0, LU, perm, ru, w, 0);
        __float128 cw = 0, cz = 0;
        double wmax = 0.0;
        for (int j = 0; j < n; ++j) { if (!isfinite(w[j])) CERT_FAIL(7); if (fabs(w[j]) > wmax) wmax = fabs(w[j]); cw += (__float128)c[j] * w[j]; cz += (__float128)c[j] * z[j]; }
        if (!((double)cw < 0.0)) CERT_FAIL(8);
        const double big = PLPO_BIG * lp_scale(m, n, c, G, h);
        for (int i = 0; i < m; ++i) {
            __float128 gw = 0, sl = h[i];
            double gmax = 0.0;
            for (int j = 0; j < n; ++j) {
                gw += (__float128)G[i * n + j] * w[j]; sl -= (__float128)G[i * n + j] * z[j];
                if (fabs(G[i * n + j]) > gmax) gmax = fabs(G[i * n + j]);
            }
            double tol = gmax * xs;
            if (fabs(h[i]) > tol) tol = fabs(h[i]);
            if ((double)sl < -PLPO_TOL_PRIMAL * tol) CERT_FAIL(9);             /* the vertex itself must be feasible */
            int inb = 0;                                          /* rows of the basis: G_k.w = 0 (or -1) by construction */
            for (int k = 0; k < n; ++k) inb |= (basis[k] == i);
            if (inb || !((double)gw > 1e-14 * gmax * wmax)) continue;   /* (below the rounding of w: not a blocking row) */
            if (sl < 0) sl = 0;
            const __float128 t = sl / gw;                        /* the ray meets row i here ... */
            if (fabsq(cz + t * cw) <= (__float128)big) CERT_FAIL(10); /* ... before the objective is out of range */
        }
        return 1;
    }
    /* dual: M' y = -c */
    double y[PLPO_MAXN], nc_[PLPO_MAXN];
    for (int j = 0; j < n; ++j) nc_[j] = -c[j];
    lu_solve_refined(n, M0, LU, perm, nc_, y, 1);
    for (int k = 0; k < n; ++k) {
        if (!isfinite(y[k])) CERT_FAIL(11);
        if (basis[k] >= 0) {
            double gmax = 0.0;
            for (int j = 0; j < n; ++j) if (fabs(M0[k * n + j]) > gmax) gmax = fabs(M0[k * n + j]);
            if (y[k] * gmax < -PLPO_TOL_DUAL * cmax) CERT_FAIL(12);
        } else if (fabs(y[k]) > PLPO_TOL_DUAL * cmax) CERT_FAIL(13);
    }
    for (int i = 0; i < m; ++i) {
        __float128 s = h[i];
        double gmax = 0.0;
        for (int j = 0; j < n; ++j) { s -= (__float128)G[i * n + j] * z[j]; if (fabs(G[i * n + j]) > gmax) gmax = fabs(G[i * n + j]); }
        double tol = gmax * xs;
        if (fabs(h[i]) > tol) tol = fabs(h[i]);
        if ((double)s < -PLPO_TOL_PRIMAL * tol) CERT_FAIL(14);
    }
    __float128 f = 0;
    for (int j = 0; j < n; ++j) { f += (__float128)c[j] * z[j]; x[j] = z[j]; }
    *fun = (double)f;
    return 1;
}

/* lpsolve (solvers.py:76-106, :149-158): the double engine, its answer certified, binary128 where that fails; an
 * optimum out of range (PLPO_BIG) is reported unbounded. */
int plpo_lp_solve(int m, int n, const double *c, const double *G, const double *h,
                  double *x, double *fun, int *iters)
{
    static const double qnan = NAN;
    int basis[PLPO_MAXN + 2];
    /* HiGHS -- the reference's solver behind scipy.optimize.linprog (solvers.py:152-158) -- treats matrix entries of magnitude
     * <= 1e-9 as ZERO (its `small_matrix_value`; checked here on this image: min x0 s.t. -x0 + eps x1 <= 2, |x1| <= 1e6 gives
     * -2 - 1e6 eps down to eps = 1.0000001e-9 and -2 from 1e-9 on).  So does this function, as the HIP library's verifier and
     * careful engine do (LpView::g, csrc/plp_verify.hpp): a row tilted by 1e-16 from its twin is that twin for the reference,
     * not a plane that meets it 1e16 away. */
    double Gc[PLPO_MAXM * PLPO_MAXN];
    if (g_certify && m <= PLPO_MAXM && n <= PLPO_MAXN && m >= 0 && n >= 1) {
        for (int k = 0; k < m * n; ++k) Gc[k] = fabs(G[k]) <= PLPO_SMALL_ENTRY ? 0.0 : G[k];
        G = Gc;
    }
    int st = plpo_lp_solve_raw(m, n, c, G, h, x, fun, iters, basis);
    if (!g_certify || st == ST_INFEAS || m > PLPO_MAXM || n > PLPO_MAXN - 1 || n < 1) return st;
    /* (an "unbounded" of the double engine is not certified by its ray any more: the HIP library has no ray to look at and sends
     * every such LP to its careful engine -- the same flow here, or slivers end "unbounded" on one side and at a corner 1e8 away
     * on the other; lp_certify keeps the ray check for tests) */
    int certified = st == ST_OPT && lp_certify(m, n, c, G, h, basis, st, x, fun);
    if (certified) {
        /* a certificate counts at a vertex within PLPO_FAR times the data's scale: its row test allows 2e-14 of |G_i| |x|, which
         * 4e9 out is 1e-4 (LpView::far_vertex, csrc/plp_verify.hpp); far vertices go to binary128 */
        double xmax = 0.0, hs = 1.0;
        for (int j = 0; j < n; ++j) if (fabs(x[j]) > xmax) xmax = fabs(x[j]);
        if (xmax > PLPO_FAR) {
            for (int i = 0; i < m; ++i) {
                double gm = 0.0;
                for (int j = 0; j < n; ++j) if (fabs(G[i * n + j]) > gm) gm = fabs(G[i * n + j]);
                if (gm > 0.0 && fabs(h[i]) > hs * gm) hs = fabs(h[i]) / gm;
            }
            if (xmax > PLPO_FAR * hs) certified = 0;
        }
    }
    if (certified) {
        ++g_cert_stat[0];
    } else {
        if (st == ST_NUM) {   /* non-finite input (the raw engine's argument check): nothing to solve */
            int fin = 1;
            for (int j = 0; j < n; ++j) if (!isfinite(c[j])) fin = 0;
            for (int i = 0; i < m && fin; ++i) { if (!isfinite(h[i])) fin = 0; for (int j = 0; j < n; ++j) if (!isfinite(G[i * n + j])) fin = 0; }
            if (!fin) return st;
        }
        ++g_cert_stat[1];
        const int sq = plpo_lp_solve_q(m, n, c, G, h, x, fun, NULL, NULL);
        if (sq != st) ++g_cert_stat[2];
        st = sq;
    }
    if (st == ST_OPT) {
        /* Out of range: the VALUE.  (Until the entries below 1e-9 were dropped as HiGHS drops them, a far VERTEX counted too --
         * for the corners 1e16 away that rows an ulp apart define.  Those are gone with the entries; and where the optimal face
         * is long -- the ball of an unbounded polytope can slide along it -- which of its vertices an engine ends on is an accident
         * of its path, the value is not: tests/golden/found/wide104_t132_k295.npz, radius 3 with the centre 4e9 away.) */
        const double sc = lp_scale(m, n, c, G, h);
        if (fabs(*fun) > PLPO_BIG * sc) st = ST_UNBND;
    }
    if (st != ST_OPT) { for (int j = 0; j < n; ++j) x[j] = qnan; *fun = qnan; }
    return st;
}

/* F1 (polytope.py:1283-1288): c = -e_{d+1}, G = [A | sqrt(sum(A*A,1))], h = b.
 * Returns the raw LP status; r = x[-1], xc = x[:-1] (NaN unless status 0). */
static int cheby_impl(int m, int d, const double *A, const double *b, double *r, double *xc, int *iters, int cert)
{
    double G[PLPO_MAXM * PLPO_MAXN], c[PLPO_MAXN], x[PLPO_MAXN], fun;
    const int n = d + 1;
    if (m > PLPO_MAXM || d > 16) return ST_NUM;
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) { G[i * n + k] = A[i * d + k]; s += A[i * d + k] * A[i * d + k]; }
        G[i * n + d] = sqrt(s);
    }
    for (int k = 0; k < d; ++k) c[k] = 0.0;
    c[d] = -1.0;
    const int st = cert ? plpo_lp_solve(m, n, c, G, b, x, &fun, iters) : plpo_lp_solve_raw(m, n, c, G, b, x, &fun, iters, NULL);
    *r = x[d];
    for (int k = 0; k < d; ++k) xc[k] = x[k];
    return st;
}

int plpo_cheby(int m, int d, const double *A, const double *b, double *r, double *xc, int *iters)
{
    return cheby_impl(m, d, A, b, r, xc, iters, 1);
}

/* F3 (polytope.py:1367-1409).  lb/ub get +-inf on status 3, 0 / lb on status 2.
 * Returns 0, or the offending LP status (1/4) where the reference raises RuntimeError. */
static int bbox_impl(int m, int d, const double *A, const double *b, double *lb, double *ub, int *nlp, int cert)
{
    double c[PLPO_MAXN], x[PLPO_MAXN], fun;
    int bad = 0;
    for (int i = 0; i < d; ++i) {
        for (int k = 0; k < d; ++k) c[k] = 0.0;
        c[i] = 1.0;
        int st = cert ? plpo_lp_solve(m, d, c, A, b, x, &fun, NULL) : plpo_lp_solve_raw(m, d, c, A, b, x, &fun, NULL, NULL);
        if (nlp) ++*nlp;
        if (st == ST_OPT) lb[i] = x[i];
        else if (st == ST_UNBND) lb[i] = -INFINITY;
        else if (st == ST_INFEAS) lb[i] = 0.0;
        else { lb[i] = NAN; bad = st; }
    }
    for (int i = 0; i < d; ++i) {
        for (int k = 0; k < d; ++k) c[k] = 0.0;
        c[i] = -1.0;
        int st = cert ? plpo_lp_solve(m, d, c, A, b, x, &fun, NULL) : plpo_lp_solve_raw(m, d, c, A, b, x, &fun, NULL, NULL);
        if (nlp) ++*nlp;
        if (st == ST_OPT) ub[i] = x[i];
        else if (st == ST_UNBND) ub[i] = INFINITY;
        else if (st == ST_INFEAS) ub[i] = lb[i];
        else { ub[i] = NAN; bad = st; }
    }
    return bad;
}

int plpo_bounding_box(int m, int d, const double *A, const double *b, double *lb, double *ub, int *nlp)
{
    return bbox_impl(m, d, A, b, lb, ub, nlp, 1);
}

/* flags returned by plpo_reduce */
#define RF_EMPTY 1    /* not full-dimensional: reference returns Polytope()       */
#define RF_EARLY 2    /* returned at neq <= nx+1 (minrep stays False)             */
#define RF_MINREP 4   /* went through the redundancy LPs (minrep = True)          */
#define RF_LPFAIL 8   /* a bounding-box LP came back with status 1/4 (RuntimeError) */
#define PLPO_TOL_CENTRE 1e-6
#define RF_F1OPEN 32  /* RF_EMPTY because the Chebyshev LP did not end optimal (unbounded / a limit): polytope_amd/csrc/plp_common.hpp */

/* reduce (polytope.py:1053-1163) on ONE polytope that is not already minrep.
 * keep: bit i set <=> input row i survives.  bout[i] = b value of row i as the
 * reference would hand it to Polytope(...) (after the +0.1/-0.1 round trip).
 * r/xc: the Chebyshev ball computed by is_fulldim (polytope.py:1081). */
int plpo_reduce(int m, int d, const double *A, const double *b, double abs_tol,
                uint64_t *keep, double *bout, double *r, double *xc, int *nlp)
{
    int idx[PLPO_MAXM], neq = 0, flags = 0;
    double Aw[PLPO_MAXM * 16], bw[PLPO_MAXM];
    for (int w = 0; w < PLPO_KEEP_WORDS; ++w) keep[w] = 0;   /* keep[PLPO_KEEP_WORDS]: bit i of word i / 64 <=> row i kept */
    *nlp = 0;
    for (int i = 0; i < m; ++i) bout[i] = b[i];
    if (m > PLPO_MAXM_RED) { *r = 0.0; for (int k = 0; k < d; ++k) xc[k] = NAN; return RF_EMPTY; }
    /* :1081 is_fulldim -> cheby_ball -> F1 */
    int st = cheby_impl(m, d, A, b, r, xc, NULL, 0);   /* the fused kernels' engine: not certified (see lp_certify) */
    ++*nlp;
    if (st == ST_OPT && *r >= 0.0) {
        /* The fused kernels start every later LP from this centre (centre-relative coordinates, ray presolve): an "optimal"
         * centre that violates a row of the polytope -- the raw engine next to twin rows: tests/golden/found/lane93_t21_k20730.npz,
         * radius right, centre 5 outside -- is refused by them and by this function alike (centre_off, csrc/plp_common.hpp). */
        double xs = 1.0;
        for (int k = 0; k < d; ++k) xs = fmax(xs, fabs(xc[k]));
        for (int i = 0; i < m; ++i) {
            double s = 0.0, n2 = 0.0;
            for (int k = 0; k < d; ++k) { s = fma(A[i * d + k], xc[k], s); n2 = n2 + A[i * d + k] * A[i * d + k]; }
            const double inv = 1.0 / sqrt(n2);
            if (isfinite(inv) && (b[i] - s) * inv < -PLPO_TOL_CENTRE * fmax(fabs(b[i]) * inv, xs)) { st = ST_NUM; break; }
        }
    }
    if (!(st == ST_OPT && *r >= 0.0)) { *r = 0.0; for (int k = 0; k < d; ++k) xc[k] = NAN; }
    if (!(*r > abs_tol)) return RF_EMPTY | ((st != ST_OPT && st != ST_INFEAS) ? RF_F1OPEN : 0);
    /* :1087-1089 drop rows with b == inf */
    for (int i = 0; i < m; ++i) if (b[i] != INFINITY) idx[neq++] = i;
    /* :1094-1110 parallel-row dedupe */
    {
        double an[PLPO_MAXM], nrm[PLPO_MAXM * 16];
        int rem[PLPO_MAXM];
        for (int p = 0; p < neq; ++p) {
            const double *a = A + idx[p] * d;
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += a[k] * a[k];
            an[p] = 1.0 / sqrt(s);
            for (int k = 0; k < d; ++k) nrm[p * d + k] = a[k] * an[p];
            rem[p] = 0;
        }
        for (int p = 0; p < neq; ++p)
            for (int q = p + 1; q < neq; ++q) {
                double dot = 0.0;
                for (int k = 0; k < d; ++k) dot += nrm[p * d + k] * nrm[q * d + k];
                if (dot > 1.0 - abs_tol) {
                    const double bp = b[idx[p]] * an[p], bq = b[idx[q]] * an[q];
                    if (bp < bq) rem[q] = 1; else rem[p] = 1;
                }
            }
        int k2 = 0;
        for (int p = 0; p < neq; ++p) if (!rem[p]) idx[k2++] = idx[p];
        neq = k2;
    }
    /* :1114-1116 */
    if (neq <= d + 1) {
        for (int p = 0; p < neq; ++p) keep[idx[p] >> 6] |= (uint64_t)1 << (idx[p] & 63);
        return RF_EARLY;
    }
    for (int p = 0; p < neq; ++p) {
        for (int k = 0; k < d; ++k) Aw[p * d + k] = A[idx[p] * d + k];
        bw[p] = b[idx[p]];
    }
    /* :1118-1134 bounding-box prefilter */
    if (neq > 3 * d) {
        double lb[16], ub[16];
        if (bbox_impl(neq, d, Aw, bw, lb, ub, nlp, 0)) flags |= RF_LPFAIL;
        int k2 = 0;
        for (int p = 0; p < neq; ++p) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double a = Aw[p * d + k];
                const double pa = (a > 0.0 ? 1.0 : 0.0) * a;
                s1 += pa * (ub[k] - lb[k]);
                s2 += a * lb[k];
            }
            const int out = (s1 - (bw[p] - s2)) < -1e-4;
            if (!out) {
                idx[k2] = idx[p];
                for (int k = 0; k < d; ++k) Aw[k2 * d + k] = Aw[p * d + k];
                bw[k2] = bw[p];
                ++k2;
            }
        }
        neq = k2;
    }
    /* :1136-1138 */
    if (neq <= d + 1) {
        for (int p = 0; p < neq; ++p) keep[idx[p] >> 6] |= (uint64_t)1 << (idx[p] & 63);
        return flags | RF_EARLY;
    }
    /* :1142-1160 one redundancy LP per row; h[k] +0.1 / -0.1 round trip persists */
    for (int k = 0; k < neq; ++k) {
        double c[16], x[16], fun;
        for (int j = 0; j < d; ++j) c[j] = -Aw[k * d + j];
        bw[k] += 0.1;
        st = plpo_lp_solve_raw(neq, d, c, Aw, bw, x, &fun, NULL, NULL);
        ++*nlp;
        bw[k] -= 0.1;
        if (st == ST_OPT) {
            const double obj = -fun - bw[k];
            if (obj > abs_tol) keep[idx[k] >> 6] |= (uint64_t)1 << (idx[k] & 63);
        } else if (st == ST_UNBND) {
            keep[idx[k] >> 6] |= (uint64_t)1 << (idx[k] & 63);
        }
        bout[idx[k]] = bw[k];
    }
    return flags | RF_MINREP;
}

/* plpo_reduce over a packed batch A[B][m][d], b[B][m] (the layout of the C ABI's plp_reduce_batch): one call of
 * plpo_reduce per polytope, nothing shared between them.  keep0[B] = word 0 of the keep mask (m <= 64 here),
 * flags[B], r[B], nlp[B].  Used by the full-batch parity checks (bench.py, tests): the loop lives in C so that a
 * host core checks ~50 k (16,3) polytopes per second instead of paying a ctypes call for each. */
int plpo_reduce_batch(int64_t B, int m, int d, const double *A, const double *b, double abs_tol,
                      uint64_t *keep0, int32_t *flags, double *r, int32_t *nlp)
{
    if (m > 64 || d > 16) return -1;
    for (int64_t k = 0; k < B; ++k) {
        uint64_t kw[PLPO_KEEP_WORDS];
        double bout[64], xc[16];
        int n = 0;
        flags[k] = plpo_reduce(m, d, A + (size_t)k * m * d, b + (size_t)k * m, abs_tol, kw, bout, &r[k], xc, &n);
        keep0[k] = kw[0];
        nlp[k] = n;
    }
    return 0;
}

/* contains (polytope.py:217-218): out[p*N + q] = all_i( A_p[i,:].X[:,q] - b_p[i] < tol ).
 * X is [N][d] (point-major).  The dot product is a k-ordered fma chain, like the kernel. */
void plpo_contains(int P, int m_max, int d, const double *A, const double *b, const int32_t *mrows,
                   int64_t N, const double *X, double abs_tol, uint8_t *out)
{
    for (int p = 0; p < P; ++p) {
        const int m = mrows ? mrows[p] : m_max;
        const double *Ap = A + (size_t)p * m_max * d, *bp = b + (size_t)p * m_max;
        for (int64_t q = 0; q < N; ++q) {
            const double *x = X + q * d;
            int ok = 1;
            for (int i = 0; i < m; ++i) {
                double s = Ap[i * d] * x[0];
                for (int k = 1; k < d; ++k) s = fma(Ap[i * d + k], x[k], s);
                if (!(s - bp[i] < abs_tol)) ok = 0;
            }
            out[(size_t)p * N + q] = (uint8_t)ok;
        }
    }
}

/* Region.contains (polytope.py:732-746): OR over polytopes. */
void plpo_region_contains(int P, int m_max, int d, const double *A, const double *b, const int32_t *mrows,
                          int64_t N, const double *X, double abs_tol, uint8_t *out)
{
    for (int64_t q = 0; q < N; ++q) out[q] = 0;
    for (int p = 0; p < P; ++p) {
        const int m = mrows ? mrows[p] : m_max;
        const double *Ap = A + (size_t)p * m_max * d, *bp = b + (size_t)p * m_max;
        for (int64_t q = 0; q < N; ++q) {
            if (out[q]) continue;
            const double *x = X + q * d;
            int ok = 1;
            for (int i = 0; i < m && ok; ++i) {
                double s = Ap[i * d] * x[0];
                for (int k = 1; k < d; ++k) s = fma(Ap[i * d + k], x[k], s);
                if (!(s - bp[i] < abs_tol)) ok = 0;
            }
            if (ok) out[q] = 1;
        }
    }
}

/* np.sum(n * p) as numpy evaluates it on a contiguous vector of d doubles (quickhull.py:121): the products first,
 * then add.reduce's pairwise_sum (numpy/_core/src/umath/loops_utils.h.src) -- below 8 elements one running sum in index
 * order; from 8 on eight running sums r[j] over the blocks of eight, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
 * then the d % 8 left-over elements added one by one (d <= 128: no recursion).  Checked bit for bit against the
 * imported reference's distance() at d = 8, 9, 12, 16 (tests/golden g18). */
static double np_sum_prod(const double *n, const double *x, int d)
{
    if (d < 8) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += n[k] * x[k];
        return s;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = n[j] * x[j];
    int i = 8;
    for (; i < d - (d % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += n[i + j] * x[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < d; ++i) res += n[i] * x[i];
    return res;
}

/* quickhull outside-set assignment + furthest point.
 * distance (quickhull.py:117-121): sum(n*p) - d0 ; assignment (:224-245, :311-336): a point
 * goes to the FIRST facet (list order) with dist > abs_tol; get_furthest (:87-102): arg-max
 * with strict '<' (first maximum wins).  facet_of_point = -1 when inside all facets.
 * The dot product is numpy's sum(n*p): np_sum_prod above (index order below d = 8, eight partial sums from 8 on). */
void plpo_assign(int64_t N, int d, const double *X, int F, const double *normals, const double *offsets,
                 double abs_tol, int32_t *facet_of_point, double *dist, int64_t *argmax_per_facet,
                 double *max_per_facet)
{
    for (int f = 0; f < F; ++f) { argmax_per_facet[f] = -1; max_per_facet[f] = -INFINITY; }
    for (int64_t q = 0; q < N; ++q) {
        const double *x = X + q * d;
        facet_of_point[q] = -1; dist[q] = 0.0;
        for (int f = 0; f < F; ++f) {
            const double *nf = normals + (size_t)f * d;
            const double dd = np_sum_prod(nf, x, d) - offsets[f];
            if (dd > abs_tol) {
                facet_of_point[q] = f; dist[q] = dd;
                if (argmax_per_facet[f] < 0 || max_per_facet[f] < dd) { argmax_per_facet[f] = q; max_per_facet[f] = dd; }
                break;
            }
        }
    }
}

/* One iteration of quickhull's outside-set bookkeeping (polytope/quickhull.py:273-283 pooling of
 * the visible facets' points, :311-336 re-assignment to the new facets in creation order, :87-102
 * furthest point), on index arrays instead of Python lists: owner[q] is the facet id owning point
 * q (-1: none), dead[id] != 0 marks the visible facets.  New facets get ids new_id0 + j.
 * Ties of the furthest point go to the lowest point index. */
void plpo_hull_reassign(int64_t N, int d, const double *X, int32_t *owner, double *dist,
                        const uint8_t *dead, int new_id0, int n_new, const double *normals,
                        const double *offsets, double abs_tol, int64_t *argmax, double *maxd, int64_t *count)
{
    for (int f = 0; f < n_new; ++f) { argmax[f] = -1; maxd[f] = 0.0; count[f] = 0; }
    for (int64_t q = 0; q < N; ++q) {
        const int own = owner[q];
        if (own < 0 || own >= new_id0 || !dead[own]) continue;
        const double *x = X + q * d;
        owner[q] = -1; dist[q] = 0.0;
        for (int f = 0; f < n_new; ++f) {
            const double *nf = normals + (size_t)f * d;
            const double dd = np_sum_prod(nf, x, d) - offsets[f];
            if (dd > abs_tol) {
                owner[q] = new_id0 + f; dist[q] = dd; count[f] += 1;
                if (argmax[f] < 0 || maxd[f] < dd) { argmax[f] = q; maxd[f] = dd; }
                break;
            }
        }
    }
}

int plpo_version(void) { return 1; }
