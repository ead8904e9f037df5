// plp_verify.hpp -- a-posteriori verification of LP answers and the careful engine behind it (round 6).
//
// Why.  The dictionary engines (plp_simplex*.hpp, plp_wide.hpp, plp_lazy.hpp, plp_lds.hip) stop on ABSOLUTE tolerances
// (reduced costs above -1e-9, pivots above 1e-7) applied to a dictionary that carries the rounding of every pivot it went
// through.  On rows a hair apart -- bounding_box (polytope/polytope.py:1314-1411) has no dedupe in front of its 2d LPs,
// lpsolve (polytope/solvers.py:76-106) takes whatever it is given -- that is worth 1e-7 .. 1e-5 on a box of size 3, and
// sides that are finite came out infinite (profiles/r05/soak_wide_r05*.log).  So no answer of those engines leaves the
// library unverified any more:
//
//   certify()   From the final BASIS alone (the n rows / free variables that define the vertex -- handed over by the engine
//               as masks, or read off its x: basis_from_x) and the ORIGINAL rows: M = those rows, x = M^-1 rhs and
//               y = -M^-T c by LU with partial pivoting + iterative refinement with double-double residuals; then
//                   primal  h_i - G_i.x >= -1e-10 max(|h_i|, |G_i|_inf max(1, |x|_inf))          for every row,
//                   dual    y_k |G_k|_inf >= -1e-12 |c|_inf on active rows, |y_k| <= 1e-12 |c|_inf on free variables:
//               an optimal basis of the LP as given, its vertex recomputed to the last bits whatever path led there (the
//               polished x replaces the engine's).  An unbounded answer is checked the same way: the vertex the engine
//               stood on and the ray it left along, every row the ray runs into beyond the point where the objective
//               passes BIG times the scale of the data.
//   careful_solve()  What fails is solved again from scratch: the same textbook two-phase dictionary simplex, in
//               double-double arithmetic on the row-equilibrated LP, tolerances 1e-12 (the certificate's).  Scalar code,
//               one LP per thread, dictionary in global memory: slow (milliseconds) and rare (nothing on random, ragged,
//               rescaled, flat or lattice data; ~5 % of the LPs of polytopes with rows 1e-16 .. 1e-5 rad apart).
//   An optimum beyond BIG = 1e9 times the scale of the data is reported UNBOUNDED -- what HiGHS does with such LPs
//   (measured: box sides of exact value up to 1.5e9 come back as that value, beyond ~2e9 as +-inf).
//
// oracle/plp_oracle.c (lp_certify, binary128 residuals) and oracle/plp_oracle_q.c (binary128 dictionary) are the test-side
// restatement of the same two steps.  Everything here compiles for the host too: tests/cabi/verify_host.cpp runs both
// steps against the oracle's on the CPU (tests/test_verify_host.py).
#pragma once
#include <stdint.h>

#include "plp_dd.hpp"

namespace plp {
namespace verify {

constexpr int VNMAX = 17;         // columns of an LP (d + 1)
constexpr int VNC = VNMAX + 1;    // + the phase-1 artificial
constexpr int VW = 20;            // doubles per dictionary row in the careful engine's scratch: VNC columns, beta, spare
constexpr double V_BIG = 1e9;     // optimum beyond V_BIG x scale(data): unbounded
constexpr double V_TOL_DUAL = 1e-12;
constexpr double V_TOL_PRIMAL = 1e-10;
constexpr double C_TOL_D = 1e-12, C_TOL_PIV = 1e-12, C_TOL_FEAS = 1e-7, C_DEGEN = 1e-24;
constexpr int C_BLAND_AFTER = 6;
enum : int { V_OPT = 0, V_ITER = 1, V_INFEAS = 2, V_UNBND = 3, V_NUM = 4 };
enum : int { LP_GENERIC = 0, LP_CHEBY = 1, LP_BOXSIDE = 2 };

// One LP  min c.x  s.t.  G x <= h  as the three LP forms of the reference give it, without materialising it:
//   LP_GENERIC  c, G (m x n row-major), h as passed to lpsolve (solvers.py:152-154)
//   LP_CHEBY    F1 (polytope.py:1283-1288): G = [A | sqrt(sum(A*A, 1))], h = b, c = (0, .., 0, -1);  n = d + 1
//   LP_BOXSIDE  F3 (polytope.py:1367-1396): G = A, h = b, c = +e_k (side = 2k: lower) / -e_k (side = 2k + 1: upper); n = d
struct LpView {
    int m, n, kind, side;
    const double* G;  // generic: G; otherwise A (m x d, d = n - 1 for LP_CHEBY)
    const double* h;
    const double* c;  // generic only
    PLP_HD double g(int i, int j) const {
        if (kind == LP_CHEBY) {
            const int d = n - 1;
            if (j < d) return G[(long)i * d + j];
            double s = 0.0;
            for (int k = 0; k < d; ++k) s = s + G[(long)i * d + k] * G[(long)i * d + k];
            return sqrt(s);
        }
        return G[(long)i * n + j];
    }
    PLP_HD double hh(int i) const { return h[i]; }
    PLP_HD double cc(int j) const {
        if (kind == LP_GENERIC) return c[j];
        if (kind == LP_CHEBY) return j == n - 1 ? -1.0 : 0.0;
        return (j == (side >> 1)) ? ((side & 1) ? -1.0 : 1.0) : 0.0;
    }
    PLP_HD double row_inf(int i) const {
        double gm = 0.0;
        for (int j = 0; j < n; ++j) gm = fmax(gm, fabs(g(i, j)));
        return gm;
    }
    PLP_HD double c_inf() const {
        double cm = 0.0;
        for (int j = 0; j < n; ++j) cm = fmax(cm, fabs(cc(j)));
        return cm;
    }
    // |c|_inf * max(1, max_i |h_i| / |G_i|_inf): what "out of range" is measured against
    PLP_HD double scale() const {
        double hs = 1.0;
        for (int i = 0; i < m; ++i) {
            const double gm = row_inf(i);
            if (gm > 0.0 && fabs(hh(i)) > hs * gm) hs = fabs(hh(i)) / gm;
        }
        return c_inf() * hs;
    }
};

// ---------------------------------------------------------------------------------------------------- LU, refined solves
// Cert<VN>: the certificate for LPs of up to VN columns (VN sizes the per-thread arrays: 5 / 9 / 17 on the device, so that
// the small shapes do not pay the scratch memory of the large ones)
template <int VN>
struct Cert {
// LU of the n x n matrix (row-major, stride VN) with partial pivoting; false: singular to working precision
static PLP_HD bool lu_factor(int n, double* LU, int* perm) {
    double big = 0.0;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j) big = fmax(big, fabs(LU[k * VN + j]));
    if (!(big > 0.0)) return false;
    for (int k = 0; k < n; ++k) perm[k] = k;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i)
            if (fabs(LU[i * VN + k]) > fabs(LU[p * VN + k])) p = i;
        if (!(fabs(LU[p * VN + k]) > 1e-13 * big)) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                const double t = LU[k * VN + j];
                LU[k * VN + j] = LU[p * VN + j];
                LU[p * VN + j] = t;
            }
            const int t = perm[k];
            perm[k] = perm[p];
            perm[p] = t;
        }
        const double inv = 1.0 / LU[k * VN + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = LU[i * VN + k] * inv;
            LU[i * VN + k] = f;
            for (int j = k + 1; j < n; ++j) LU[i * VN + j] = fma(-f, LU[k * VN + j], LU[i * VN + j]);
        }
    }
    return true;
}
static PLP_HD void lu_solve(int n, const double* LU, const int* perm, const double* r, double* z) {
    double t[VN];
    for (int k = 0; k < n; ++k) {
        double s = r[perm[k]];
        for (int j = 0; j < k; ++j) s = fma(-LU[k * VN + j], t[j], s);
        t[k] = s;
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = t[k];
        for (int j = k + 1; j < n; ++j) s = fma(-LU[k * VN + j], z[j], s);
        z[k] = s / LU[k * VN + k];
    }
}
static PLP_HD void lu_solve_t(int n, const double* LU, const int* perm, const double* r, double* z) {
    double t[VN];
    for (int k = 0; k < n; ++k) {
        double s = r[k];
        for (int j = 0; j < k; ++j) s = fma(-LU[j * VN + k], t[j], s);
        t[k] = s / LU[k * VN + k];
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = t[k];
        for (int j = k + 1; j < n; ++j) s = fma(-LU[j * VN + k], t[j], s);
        t[k] = s;
    }
    for (int k = 0; k < n; ++k) z[perm[k]] = t[k];
}

// entry (k, j) of the basis matrix: row basis[k] of G, or e_j0 for the free variable j0 = -1 - basis[k]
static PLP_HD double basis_entry(const LpView& lp, const int* basis, int k, int j) {
    const int v = basis[k];
    return v >= 0 ? lp.g(v, j) : ((j == -1 - v) ? 1.0 : 0.0);
}
// z = M^-1 r (trans: M^-T r) with three rounds of refinement, residuals accumulated in double-double
static PLP_HD void solve_refined(const LpView& lp, const int* basis, int n, const double* LU, const int* perm, const double* r,
                          double* z, bool trans) {
    double rr[VN], dz[VN];
    if (trans) lu_solve_t(n, LU, perm, r, z);
    else lu_solve(n, LU, perm, r, z);
    for (int it = 0; it < 3; ++it) {
        for (int k = 0; k < n; ++k) {
            dd s = dd_make(r[k]);
            for (int j = 0; j < n; ++j) {
                const double mkj = trans ? basis_entry(lp, basis, j, k) : basis_entry(lp, basis, k, j);
                s = dd_sub(s, two_prod(mkj, z[j]));
            }
            rr[k] = dd_to_double(s);
        }
        if (trans) lu_solve_t(n, LU, perm, rr, dz);
        else lu_solve(n, LU, perm, rr, dz);
        for (int j = 0; j < n; ++j) z[j] += dz[j];
    }
}

// ---------------------------------------------------------------------------------------------------- the certificate
// status: V_OPT / V_UNBND as the engine reported it.  basis[0..n): >= 0 an active row, -1 - j the free variable x_j held at
// xref[j] (xref == nullptr: 0); basis[n], basis[n + 1] (V_UNBND): position in the list of the variable the ray moves and its
// sign.  true: certified; for V_OPT x[0..n) = the polished vertex, *fun = c.x.
static PLP_HD bool certify(const LpView& lp, int status, const int* basis, const double* xref, double* x, double* fun) {
    const int n = lp.n, m = lp.m;
    double LU[VN * VN], rhs[VN], z[VN];
    int perm[VN];
    if (n > VN || n < 1) return false;
    for (int k = 0; k < n; ++k) {
        const int v = basis[k];
        if (v >= 0) {
            if (v >= m) return false;
            rhs[k] = lp.hh(v);
        } else {
            const int j0 = -1 - v;
            if (j0 < 0 || j0 >= n) return false;
            rhs[k] = xref ? xref[j0] : 0.0;
        }
        for (int j = 0; j < n; ++j) LU[k * VN + j] = basis_entry(lp, basis, k, j);
    }
    if (!lu_factor(n, LU, perm)) return false;
    const double cmax = lp.c_inf();
    solve_refined(lp, basis, n, LU, perm, rhs, z, false);
    double zmax = 0.0;
    for (int j = 0; j < n; ++j) {
        if (!(fabs(z[j]) < 1e300)) return false;
        zmax = fmax(zmax, fabs(z[j]));
    }
    const double xs = zmax > 1.0 ? zmax : 1.0;
    if (status == V_UNBND) {
        const int e = basis[n];
        double w[VN], ru[VN];
        if (e < 0 || e >= n) return false;
        for (int k = 0; k < n; ++k) ru[k] = 0.0;
        ru[e] = basis[e] >= 0 ? -1.0 : (double)basis[n + 1];  // the slack of an active row grows / the free variable moves by its sign
        solve_refined(lp, basis, n, LU, perm, ru, w, false);
        dd cw = dd_make(0.0), cz = dd_make(0.0);
        double wmax = 0.0;
        for (int j = 0; j < n; ++j) {
            if (!(fabs(w[j]) < 1e300)) return false;
            wmax = fmax(wmax, fabs(w[j]));
            cw = dd_add(cw, two_prod(lp.cc(j), w[j]));
            cz = dd_add(cz, two_prod(lp.cc(j), z[j]));
        }
        if (!(cw.hi < 0.0)) return false;
        const double big = V_BIG * lp.scale();
        for (int i = 0; i < m; ++i) {
            dd gw = dd_make(0.0), sl = dd_make(lp.hh(i));
            double gmax = 0.0;
            for (int j = 0; j < n; ++j) {
                const double gij = lp.g(i, j);
                gw = dd_add(gw, two_prod(gij, w[j]));
                sl = dd_sub(sl, two_prod(gij, z[j]));
                gmax = fmax(gmax, fabs(gij));
            }
            const double tol = fmax(gmax * xs, fabs(lp.hh(i)));
            if (sl.hi < -V_TOL_PRIMAL * tol) return false;  // the vertex itself must be feasible
            bool inb = false;                                // rows of the basis: G_k.w = 0 (or -1) by construction
            for (int k = 0; k < n; ++k) inb = inb | (basis[k] == i);
            if (inb || !(gw.hi > 1e-14 * gmax * wmax)) continue;  // (below the rounding of w: not a blocking row)
            if (dd_lt_d(sl, 0.0)) sl = dd_make(0.0);
            const dd t = dd_div(sl, gw);                     // the ray meets row i here ...
            const dd obj = dd_add(cz, dd_mul(t, cw));
            if (!(fabs(obj.hi) > big)) return false;         // ... before the objective is out of range
        }
        return true;
    }
    // dual: M' y = -c
    double y[VN], nc_[VN];
    for (int j = 0; j < n; ++j) nc_[j] = -lp.cc(j);
    solve_refined(lp, basis, n, LU, perm, nc_, y, true);
    for (int k = 0; k < n; ++k) {
        if (!(fabs(y[k]) < 1e300)) return false;
        if (basis[k] >= 0) {
            const double gmax = lp.row_inf(basis[k]);
            if (y[k] * gmax < -V_TOL_DUAL * cmax) return false;
        } else if (fabs(y[k]) > V_TOL_DUAL * cmax) return false;
    }
    for (int i = 0; i < m; ++i) {
        dd s = dd_make(lp.hh(i));
        double gmax = 0.0;
        for (int j = 0; j < n; ++j) {
            const double gij = lp.g(i, j);
            s = dd_sub(s, two_prod(gij, z[j]));
            gmax = fmax(gmax, fabs(gij));
        }
        const double tol = fmax(gmax * xs, fabs(lp.hh(i)));
        if (s.hi < -V_TOL_PRIMAL * tol) return false;
    }
    dd f = dd_make(0.0);
    for (int j = 0; j < n; ++j) {
        f = dd_add(f, two_prod(lp.cc(j), z[j]));
        x[j] = z[j];
    }
    *fun = dd_to_double(f);
    return true;
}

// A basis read off an engine's x (engines that do not hand over theirs): the rows whose slack at x is below 1e-9 of their
// scale, in order of increasing slack, as long as they are linearly independent of the ones taken so far (modified
// Gram-Schmidt; a copy of a taken row adds nothing to the cone they span); free variables x_j held at x_j complete it
// where x lies on a face rather than at a vertex.  At a degenerate vertex the choice may not be the dual-feasible one: the
// certificate then fails and the LP goes to the careful engine.  false: no basis (more than VN candidates ...).
static PLP_HD bool basis_from_x(const LpView& lp, const double* x, int* basis) {
    const int n = lp.n, m = lp.m;
    double Q[VN * VN];  // orthonormal rows spanning the accepted rows
    int nb = 0;
    double xs = 1.0;
    for (int j = 0; j < n; ++j) {
        if (!(fabs(x[j]) < 1e300)) return false;
        xs = fmax(xs, fabs(x[j]));
    }
    // candidates by increasing slack: repeated selection of the smallest slack above the last one taken (m is small)
    double last = -1e300;
    int lasti = -1;
    for (int round = 0; round < m && nb < n; ++round) {
        int bi = -1;
        double bs = 1e300;
        for (int i = 0; i < m; ++i) {
            double s = lp.hh(i), gmax = 0.0;
            for (int j = 0; j < n; ++j) {
                const double gij = lp.g(i, j);
                s = fma(-gij, x[j], s);
                gmax = fmax(gmax, fabs(gij));
            }
            if (!(gmax > 0.0)) continue;
            const double tol = fmax(gmax * xs, fabs(lp.hh(i)));
            if (!(s <= 1e-9 * tol)) continue;
            if (s < last || (s == last && i <= lasti)) continue;  // taken or looked at already
            if (s < bs) { bs = s; bi = i; }
        }
        if (bi < 0) break;
        last = bs;
        lasti = bi;
        double v[VN], nrm0 = 0.0, nrm1 = 0.0;
        for (int j = 0; j < n; ++j) { v[j] = lp.g(bi, j); nrm0 = fma(v[j], v[j], nrm0); }
        for (int q = 0; q < nb; ++q) {
            double dq = 0.0;
            for (int j = 0; j < n; ++j) dq = fma(Q[q * VN + j], v[j], dq);
            for (int j = 0; j < n; ++j) v[j] = fma(-dq, Q[q * VN + j], v[j]);
        }
        for (int j = 0; j < n; ++j) nrm1 = fma(v[j], v[j], nrm1);
        if (!(nrm1 > 1e-12 * nrm0)) continue;  // (1e-6 of its length: dependent on the rows taken so far)
        const double inv = 1.0 / sqrt(nrm1);
        for (int j = 0; j < n; ++j) Q[nb * VN + j] = v[j] * inv;
        basis[nb++] = bi;
    }
    // complete with free variables (unit vectors), most independent first
    while (nb < n) {
        int bj = -1;
        double bn = 0.0;
        for (int j0 = 0; j0 < n; ++j0) {
            double r2 = 1.0;  // |e_j0 - Q'Q e_j0|^2 = 1 - sum_q Q[q][j0]^2
            for (int q = 0; q < nb; ++q) r2 = fma(-Q[q * VN + j0], Q[q * VN + j0], r2);
            if (r2 > bn) { bn = r2; bj = j0; }
        }
        if (bj < 0 || !(bn > 1e-12)) return false;
        double v[VN], nrm1 = 0.0;
        for (int j = 0; j < n; ++j) v[j] = (j == bj) ? 1.0 : 0.0;
        for (int q = 0; q < nb; ++q) {
            const double dq = Q[q * VN + bj];
            for (int j = 0; j < n; ++j) v[j] = fma(-dq, Q[q * VN + j], v[j]);
        }
        for (int j = 0; j < n; ++j) nrm1 = fma(v[j], v[j], nrm1);
        if (!(nrm1 > 1e-12)) return false;
        const double inv = 1.0 / sqrt(nrm1);
        for (int j = 0; j < n; ++j) Q[nb * VN + j] = v[j] * inv;
        basis[nb++] = -1 - bj;
    }
    return true;
}
};  // struct Cert

// ---------------------------------------------------------------------------------------------------- the careful engine
// Dictionary in double-double: element (i, j) of this LP at hi[(i * VW + j) * stride], lo likewise (stride = LPs solved
// side by side: consecutive threads touch consecutive addresses).  Rows 0 .. m - 1, row m = cost, row m + 1 = carried
// cost; column VNC = beta / negz.  rowinfo[i * stride] = rowvar << 2 | (rowsgn < 0) << 1 | rowact.
struct CarefulMem {
    double* hi;
    double* lo;
    int* rowinfo;
    long stride;
    PLP_HD dd get(int i, int j) const {
        const long e = ((long)i * VW + j) * stride;
        return dd{hi[e], lo[e]};
    }
    PLP_HD void set(int i, int j, dd v) const {
        const long e = ((long)i * VW + j) * stride;
        hi[e] = v.hi;
        lo[e] = v.lo;
    }
    PLP_HD int rv(int i) const { return rowinfo[(long)i * stride] >> 2; }
    PLP_HD int rsgn(int i) const { return (rowinfo[(long)i * stride] & 2) ? -1 : 1; }
    PLP_HD bool ract(int i) const { return rowinfo[(long)i * stride] & 1; }
    PLP_HD void rset(int i, int var, int sgn, bool act) const {
        rowinfo[(long)i * stride] = (var * 4) | (sgn < 0 ? 2 : 0) | (act ? 1 : 0);
    }
};
PLP_HD size_t careful_doubles_per_lp(int m_max) { return (size_t)(m_max + 2) * VW; }

struct CarefulState {
    int m, n, nc, carry, iters, maxit;
    int colvar[VNC], colsgn[VNC], coldead[VNC];
};
constexpr int C_ID_T = -1;

PLP_HD void careful_pivot(const CarefulMem& M, CarefulState& S, int r, int e) {
    const int nc = S.nc, m = S.m;
    dd rho[VNC];
    const dd one = dd_make(1.0);
    const dd p = dd_div(one, M.get(r, e));
    for (int j = 0; j < nc; ++j) rho[j] = dd_mul(M.get(r, j), p);
    rho[e] = p;
    const dd rhob = dd_mul(M.get(r, VNC), p);
    const int rows = m + 1 + (S.carry ? 1 : 0);  // constraint rows, the cost row, the carried cost row
    for (int i = 0; i < rows; ++i) {
        if (i == r) continue;
        const dd f = M.get(i, e);
        if (f.hi == 0.0 && f.lo == 0.0) continue;
        M.set(i, e, dd_make(0.0));
        for (int j = 0; j < nc; ++j) M.set(i, j, dd_fnma(f, rho[j], M.get(i, j)));
        M.set(i, VNC, dd_fnma(f, rhob, M.get(i, VNC)));
    }
    for (int j = 0; j < nc; ++j) M.set(r, j, rho[j]);
    M.set(r, VNC, rhob);
    const int vin = S.colvar[e], vout = M.rv(r), sin_ = S.colsgn[e], sout = M.rsgn(r);
    M.rset(r, vin, sin_, !((unsigned)vin < (unsigned)S.n));  // a free variable never leaves again
    S.colvar[e] = vout;
    S.colsgn[e] = sout;
    S.iters++;
}

PLP_HD int careful_run(const CarefulMem& M, CarefulState& S) {
    int ndeg = 0;
    const int m = S.m;
    for (;;) {
        const bool bland = ndeg >= C_BLAND_AFTER;
        int e = -1, bestid = 0x7fffffff;
        dd best = dd_make(0.0);
        for (int j = 0; j < S.nc; ++j) {
            if (S.coldead[j]) continue;
            const dd dj = M.get(m, j), aj = dd_abs(dj);
            const bool fr = (unsigned)S.colvar[j] < (unsigned)S.n;
            const bool elig = fr ? dd_gt_d(aj, C_TOL_D) : dd_lt_d(dj, -C_TOL_D);
            if (!elig) continue;
            if (bland) {
                if (S.colvar[j] < bestid) { bestid = S.colvar[j]; e = j; }
            } else if (dd_gt(aj, best)) {
                best = aj;
                e = j;
            }
        }
        if (e < 0) return V_OPT;
        if (S.iters >= S.maxit) return V_ITER;
        if (dd_gt_d(M.get(m, e), 0.0)) {  // free variable entering downwards: x := -x
            for (int i = 0; i < m; ++i) M.set(i, e, dd_neg(M.get(i, e)));
            M.set(m, e, dd_neg(M.get(m, e)));
            if (S.carry) M.set(m + 1, e, dd_neg(M.get(m + 1, e)));
            S.colsgn[e] = -S.colsgn[e];
        }
        int r = -1;
        dd rmin = dd_make(0.0);
        for (int i = 0; i < m; ++i) {
            if (!M.ract(i)) continue;
            const dd a = M.get(i, e);
            if (!dd_gt_d(a, C_TOL_PIV)) continue;
            dd bi = M.get(i, VNC);
            if (dd_lt_d(bi, 0.0)) bi = dd_make(0.0);
            const dd q = dd_div(bi, a);
            if (r < 0 || dd_lt(q, rmin) || (bland && dd_eq(q, rmin) && M.rv(i) < M.rv(r))) {
                rmin = q;
                r = i;
            }
        }
        if (r < 0) return V_UNBND;
        ndeg = !dd_gt_d(rmin, C_DEGEN) ? ndeg + 1 : 0;
        careful_pivot(M, S, r, e);
    }
}

// min c.x s.t. Gx <= h, x free, from scratch.  x[0..n), *fun for status V_OPT (untouched otherwise).  The textbook method of
// the other engines (two phases, free variables enter and never leave, Dantzig pricing, Bland's rule after C_BLAND_AFTER
// degenerate pivots) on the EQUILIBRATED LP -- row i divided by |G_i|_inf, the cost by |c|_inf -- with the certificate's
// tolerances (C_TOL_*).
PLP_HD int careful_solve(const LpView& lp, const CarefulMem& M, double* x, double* fun, int* iters_out) {
    const int m = lp.m, n = lp.n;
    CarefulState S;
    S.m = m; S.n = n; S.nc = n; S.carry = 0; S.iters = 0; S.maxit = 200 * (m + n) + 1000;
    if (iters_out) *iters_out = 0;
    if (n > VNMAX || n < 1 || m < 0) return V_NUM;
    for (int j = 0; j < VNC; ++j) { S.colvar[j] = j; S.colsgn[j] = 1; S.coldead[j] = 0; }
    for (int j = 0; j < n; ++j)
        if (!(fabs(lp.cc(j)) < 1e300)) return V_NUM;
    bool need_p1 = false;
    const dd zero = dd_make(0.0);
    for (int i = 0; i < m; ++i) {
        double gmax = 0.0;
        bool fin = fabs(lp.hh(i)) < 1e300;
        for (int j = 0; j < n; ++j) {
            const double gij = lp.g(i, j);
            fin = fin & (fabs(gij) < 1e300);
            gmax = fmax(gmax, fabs(gij));
        }
        if (!fin) return V_NUM;
        for (int j = 0; j <= VNC; ++j) M.set(i, j, zero);
        if (!(gmax > 0.0)) {  // 0 <= h_i: vacuous or infeasible
            if (lp.hh(i) < -C_TOL_FEAS) return V_INFEAS;
            M.rset(i, n + i, 1, false);
            continue;
        }
        const dd sc = dd_div(dd_make(1.0), dd_make(gmax));
        for (int j = 0; j < n; ++j) M.set(i, j, dd_mul_d(sc, lp.g(i, j)));
        M.set(i, VNC, dd_mul_d(sc, lp.hh(i)));
        M.rset(i, n + i, 1, true);
        if (lp.hh(i) < 0.0) need_p1 = true;
    }
    const double cmax = lp.c_inf();
    for (int j = 0; j <= VNC; ++j) { M.set(m, j, zero); M.set(m + 1, j, zero); }
    for (int j = 0; j < n; ++j) M.set(m, j, cmax > 0.0 ? dd_div(dd_make(lp.cc(j)), dd_make(cmax)) : zero);
    int st;
    if (need_p1) {
        const int tc = n;
        S.nc = n + 1;
        S.colvar[tc] = C_ID_T;
        for (int j = 0; j < n; ++j) { M.set(m + 1, j, M.get(m, j)); M.set(m, j, zero); }
        M.set(m, tc, dd_make(1.0));
        S.carry = 1;
        int r0 = -1;
        for (int i = 0; i < m; ++i) {
            if (!M.ract(i)) continue;
            M.set(i, tc, dd_make(-1.0));
            if (r0 < 0 || dd_lt(M.get(i, VNC), M.get(r0, VNC))) r0 = i;
        }
        careful_pivot(M, S, r0, tc);
        st = careful_run(M, S);
        if (st != V_OPT) { if (iters_out) *iters_out = S.iters; return st == V_ITER ? V_ITER : V_NUM; }
        int rt = -1, ct = -1;
        for (int i = 0; i < m; ++i) if (M.rv(i) == C_ID_T) rt = i;
        for (int j = 0; j < S.nc; ++j) if (S.colvar[j] == C_ID_T) ct = j;
        if (rt >= 0) {
            if (dd_gt_d(M.get(rt, VNC), C_TOL_FEAS)) { if (iters_out) *iters_out = S.iters; return V_INFEAS; }
            int e = -1;
            dd big = dd_make(C_TOL_PIV);
            for (int j = 0; j < S.nc; ++j) {
                const dd a = dd_abs(M.get(rt, j));
                if (dd_gt(a, big)) { big = a; e = j; }
            }
            if (e >= 0) {
                careful_pivot(M, S, rt, e);
                if (!((unsigned)M.rv(rt) < (unsigned)n) && dd_lt_d(M.get(rt, VNC), 0.0)) M.set(rt, VNC, zero);
                ct = e;
            } else {
                M.rset(rt, M.rv(rt), M.rsgn(rt), false);
            }
        }
        if (ct >= 0) S.coldead[ct] = 1;
        for (int i = 0; i < m; ++i)
            if (M.ract(i) && dd_lt_d(M.get(i, VNC), 0.0)) M.set(i, VNC, zero);
        for (int j = 0; j <= VNC; ++j) M.set(m, j, M.get(m + 1, j));
        S.carry = 0;
    }
    st = careful_run(M, S);
    if (iters_out) *iters_out = S.iters;
    if (st != V_OPT) return st;
    // x_j = sgn * beta of the row that holds it (in units of the row's own scale: the free variables were not scaled)
    dd xq[VNMAX];
    for (int j = 0; j < n; ++j) xq[j] = zero;
    for (int i = 0; i < m; ++i) {
        const int v = M.rv(i);
        if ((unsigned)v < (unsigned)n) xq[v] = M.rsgn(i) < 0 ? dd_neg(M.get(i, VNC)) : M.get(i, VNC);
    }
    dd f = zero;
    for (int j = 0; j < n; ++j) {
        f = dd_add(f, dd_mul_d(xq[j], lp.cc(j)));
        x[j] = dd_to_double(xq[j]);
    }
    *fun = dd_to_double(f);
    return V_OPT;
}

// what every caller does with an optimum: out of range -> unbounded
PLP_HD int range_rule(const LpView& lp, int status, double fun) {
    return (status == V_OPT && fabs(fun) > V_BIG * lp.scale()) ? V_UNBND : status;
}

}  // namespace verify
}  // namespace plp
