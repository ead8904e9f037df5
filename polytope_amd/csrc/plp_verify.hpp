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
//                   primal  h_i - G_i.x >= -2e-14 max(|h_i|, |G_i|_inf max(1, |x|_inf))          for every row,
//                   dual    y_k |G_k|_inf >= -1e-13 |c|_inf on active rows, |y_k| <= 1e-13 |c|_inf on free variables:
//               an optimal basis of the LP as given, its vertex recomputed to the last bits whatever path led there (the
//               polished x replaces the engine's).  An unbounded answer is checked the same way: the vertex the engine
//               stood on and the ray it left along, every row the ray runs into beyond the point where the objective
//               passes BIG times the scale of the data.
//   careful_solve()  What fails is solved again from scratch: the same textbook two-phase dictionary simplex, in
//               double-double arithmetic on the row-equilibrated LP, with the certificate's tolerances.  Scalar code,
//               one LP per thread, dictionary in global memory: slow (milliseconds) and rare (nothing on random, ragged,
//               rescaled, flat or lattice data; ~5 % of the LPs of polytopes with rows 1e-16 .. 1e-5 rad apart).
//   Tolerances.  The engines stop at reduced costs above -1e-9 (TOL_D): the rate at which the objective changes per unit of
//   distance.  What such a rate is worth depends on how far the direction goes -- 1e-9 of the polytope's extent, which on a
//   sliver that reaches 1e4 is 1e-5 of a box side of 3 (profiles/r05/soak_wide_r05l_*.log) -- so here a reduced cost between the
//   rounding of the factorisation (1e-13) and 1e-9 is judged by what it BUYS: the careful engine (and the oracle's binary128
//   twin) enter such a column when the step it allows improves the objective by more than 1e-10 of max(1, |objective|) and
//   moves x by no more than BIG times the scale of the data (a sliver's hair direction pays 5e-8 on a radius of 1.5 -- 2e9 away:
//   HiGHS does not go there, nor do we); the certificate accepts no multiplier below -1e-13, so an answer with one in that zone
//   goes to the careful engine.  Below 1e-13 a coefficient is zero: a copy of the row x_0 <= 2 tilted by 1e-16 towards x_1 lets
//   the exact LP reach x_0 = 2.84 at x_1 = 1e16 where HiGHS and every double-precision code answer 2.
//   An optimum beyond BIG = 1e9 times the scale of the data is reported UNBOUNDED -- what HiGHS does with such LPs
//   (measured: box sides of exact value up to 1.5e9 come back as that value, beyond ~2e9 as +-inf).
//
// oracle/plp_oracle.c (lp_certify, binary128 residuals) and its binary128 dictionary engine beside it are the test-side
// restatement of the same two steps.  Everything here compiles for the host too: tests/cabi/verify_host.cpp runs both
// steps against the oracle's on the CPU (tests/test_verify_host.py).
#pragma once
#include <stdint.h>

#include "plp_dd.hpp"

#if defined(__HIPCC__)
#define PLP_UNROLL _Pragma("unroll")
#else
#define PLP_UNROLL
#endif

namespace plp {
namespace verify {

constexpr int VNMAX = 17;         // columns of an LP (d + 1)
constexpr int VNC = VNMAX + 1;    // + the phase-1 artificial
constexpr int VW = 20;            // doubles per dictionary row in the careful engine's scratch: VNC columns, beta, spare
constexpr double V_BIG = 1e9;     // optimum beyond V_BIG x scale(data): unbounded
constexpr double V_SMALL_ENTRY = 1e-9;   // matrix entries HiGHS drops (LpView::g)
constexpr double V_FAR = 1e4;            // a certificate counts at a vertex within V_FAR x the data's scale (LpView::far_vertex)
constexpr double V_TOL_DUAL = 1e-13;  // a multiplier below this (of |c| / |G_k|) is rounding; see the note on tolerances below
constexpr double V_TOL_PRIMAL = 2e-14;  // (ten ulps of |G_i| |x|: the fma chain itself is good to n / 2 ulps; 1e-13 left 1.2e-9 of a sliver's 2.6e7; 1e-10 let through a vertex 1e-8 outside a twin row: 7e-7 on the optimum of 235, seed 4 of verify_smoke)
constexpr double C_TOL_D = 1e-9, C_TOL_PIV = 1e-12, C_TOL_FEAS = 1e-7, C_DEGEN = 1e-24;
constexpr double C_TOL_NOISE = 1e-13;  // reduced costs between this and C_TOL_D: judged by what they buy (careful_run)
constexpr double C_TOL_GAIN = 1e-10;   // ... an improvement above this (of max(1, |objective|)) within range
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
    // HiGHS -- the reference's solver (solvers.py:152-158) -- takes matrix entries of magnitude <= 1e-9 for ZERO (its
    // small_matrix_value; checked on this image, oracle/plp_oracle.c: plpo_lp_solve): the LP this view describes is the LP with
    // those entries dropped -- a row tilted by 1e-16 from its twin IS that twin, not a plane that meets it 1e16 away.  (The
    // norms of F1 are formed by numpy from the full rows before HiGHS sees them: from the undropped entries here too.)
    static PLP_HD double kept(double v) { return fabs(v) <= V_SMALL_ENTRY ? 0.0 : v; }
    PLP_HD double g(int i, int j) const {
        if (kind == LP_CHEBY) {
            const int d = n - 1;
            if (j < d) return kept(G[(long)i * d + j]);
            double s = 0.0;
            for (int k = 0; k < d; ++k) s = s + G[(long)i * d + k] * G[(long)i * d + k];
            return kept(sqrt(s));
        }
        return kept(G[(long)i * n + j]);
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
    // max(1, max_i |h_i| / |G_i|_inf): how far from the origin the data puts its planes
    PLP_HD double h_scale() const {
        double hs = 1.0;
        for (int i = 0; i < m; ++i) {
            const double gm = row_inf(i);
            if (gm > 0.0 && fabs(hh(i)) > hs * gm) hs = fabs(hh(i)) / gm;
        }
        return hs;
    }
    // |c|_inf * h_scale(): what "out of range" is measured against
    PLP_HD double scale() const { return c_inf() * h_scale(); }
    // A certificate is taken from a vertex within V_FAR x h_scale() only.  Its row test allows 2e-14 of |G_i| |x|: at a vertex
    // 4e9 out -- where the free variables of a basis were held at a Chebyshev centre that had slid along an unbounded face -- that
    // is 1e-4, and a box side of -3.0000001 passed against a row that puts it at -3.000000025 (tests/golden/found/
    // wide104_t132_k295.npz).  Far vertices are for the careful engine, whose arithmetic has the digits.
    PLP_HD bool far_vertex(double xmax) const { return xmax > V_FAR && xmax > V_FAR * h_scale(); }
    // the same with h_scale() handed in (the device's row passes gather it as they go: max(1, max_i |h_i| / |G_i|_inf))
    static PLP_HD bool far_vertex_hs(double xmax, double hs) { return xmax > V_FAR && xmax > V_FAR * hs; }
};

// ---------------------------------------------------------------------------------------------------- LU, refined solves
// Cert<VN, ST>: the certificate for LPs of up to VN columns.  Its work arrays (the n x n factorisation, a dozen vectors) are
// NOT private arrays -- indexed by run-time loop counters they would live in scratch memory, a dependent chain of global
// round trips per element (measured: 10x the LP engines' own time) -- but views into a workspace the caller provides, element e
// at ws[e * ST]: on the device one LP per lane with ST = 64 and the workspace in LDS (lane l at ws + l: consecutive lanes,
// consecutive banks), on the host ST = 1.  ws_doubles() doubles per LP.
template <int VN, int ST, int NF = 0>
struct Cert {
// NF > 0: every LP has exactly NF columns, known at compile time -- with ST = 1 and the workspace a local array the loops
// below unroll completely and the arrays become registers (the device's small instances, n <= 5: one LP per lane, no LDS)
static PLP_HD int ncols(const LpView& lp) { return NF ? NF : lp.n; }
struct Vec {
    double* p;
    PLP_HD double& operator[](int e) const { return p[e * ST]; }
};
static constexpr int KC = VN + 7;                       // candidate rows kept by basis_from_x
// workspace layout (in doubles): [LU / Q: VN * VN][M0: the basis matrix itself, VN * VN (before it: the candidate list, 2 KC)]
// [rhs VN][z VN][y VN][t VN][rr VN][dz VN][v VN][perm VN]
// [basis VN + 2: the basis list itself, as doubles -- >= 0 an active row, -1 - j a free variable; (unbounded) position, sign]
// [x VN: the engine's point (a basis is read off it), or the values the free variables of the basis are held at]
// [gm VN: |row k of the basis matrix|_inf]
// (v doubles as -c once the basis is chosen: vertex_and_dual_finish expects it there)
static constexpr int O_LU = 0, O_M0 = VN * VN, O_RHS = 2 * VN * VN, O_Z = O_RHS + VN, O_Y = O_Z + VN, O_T = O_Y + VN,
                     O_RR = O_T + VN, O_DZ = O_RR + VN, O_V = O_DZ + VN, O_PERM = O_V + VN, O_BAS = O_PERM + VN,
                     O_X = O_BAS + VN + 2, O_GM = O_X + VN, WS_DOUBLES = O_GM + VN,
                     O_CS = O_M0, O_CI = O_M0 + KC;   // (the candidate list is done with before the basis matrix is stored)
static_assert(2 * KC <= VN * VN || VN < 5, "the candidate list shares the basis matrix' space");
static PLP_HD constexpr int ws_doubles() { return WS_DOUBLES; }
static PLP_HD Vec at(double* ws, int off) { return Vec{ws + off * ST}; }

// v[idx], v[idx] = val at a run-time index (NF: a select chain over static indices)
static PLP_HD double pick(Vec v, int n, int idx) {
    if constexpr (NF > 0) {
        double r = 0.0;
        PLP_UNROLL
        for (int i = 0; i < n; ++i) r = (i == idx) ? v[i] : r;
        return r;
    } else {
        (void)n;
        return v[idx];
    }
}
static PLP_HD void put(Vec v, int n, int idx, double val) {
    if constexpr (NF > 0) {
        PLP_UNROLL
        for (int i = 0; i < n; ++i)
            if (i == idx) v[i] = val;
    } else {
        (void)n;
        v[idx] = val;
    }
}
// dst[doff + j] -= f * src[soff + j] for j0 <= j < j1, four at a time with the loads ahead of the stores: through the views
// the compiler cannot tell that a store to dst does not feed the next load of src (the workspace is one array), and an
// element-by-element loop pays an LDS round trip per element
static PLP_HD void row_axpy(Vec dst, int doff, Vec src, int soff, double f, int j0, int j1) {
    if constexpr (NF > 0) {   // (registers: a plain loop that unrolls)
        PLP_UNROLL
        for (int j = j0; j < j1; ++j) dst[doff + j] = fma(-f, src[soff + j], dst[doff + j]);
        return;
    }
    int j = j0;
    if constexpr (NF == 0) {
        for (; j + 4 <= j1; j += 4) {
            const double s0 = src[soff + j], s1 = src[soff + j + 1], s2 = src[soff + j + 2], s3 = src[soff + j + 3];
            const double d0 = dst[doff + j], d1 = dst[doff + j + 1], d2 = dst[doff + j + 2], d3 = dst[doff + j + 3];
            dst[doff + j] = fma(-f, s0, d0);
            dst[doff + j + 1] = fma(-f, s1, d1);
            dst[doff + j + 2] = fma(-f, s2, d2);
            dst[doff + j + 3] = fma(-f, s3, d3);
        }
    }
    PLP_UNROLL
    for (; j < j1; ++j) dst[doff + j] = fma(-f, src[soff + j], dst[doff + j]);
}
// LU of the n x n matrix (row-major, stride VN) with partial pivoting, in steps so that the lanes of an LP can share the row
// updates (plp_verify.hip: the leader runs lu_begin / lu_pivot, every lane lu_rows on its rows, a barrier between them):
//   lu_begin: the largest entry (0: the matrix is zero), perm = identity
//   lu_pivot(k): pivot search in column k, row swap; false: singular to working precision; *pv = |pivot|
//   lu_rows(k, first, step): rows i = k + 1 + first, + step, ...: the multiplier and the update of the row
static PLP_HD double lu_begin(int n, Vec gm, Vec perm) {   // (gm: the rows' largest entries, basis_row)
    double big = 0.0;
    PLP_UNROLL
    for (int k = 0; k < n; ++k) big = fmax(big, gm[k]);
    PLP_UNROLL
    for (int k = 0; k < n; ++k) perm[k] = k;
    return big;
}
static PLP_HD bool lu_pivot(int n, int k, Vec LU, Vec perm, double big, double* pv_out) {
    int p = k;
    double pv = fabs(LU[k * VN + k]);
    int i = k + 1;
    if constexpr (NF == 0) {   // (four LDS loads in flight; the comparisons in the same order)
        for (; i + 4 <= n; i += 4) {
            const double a0 = fabs(LU[i * VN + k]), a1 = fabs(LU[(i + 1) * VN + k]);
            const double a2 = fabs(LU[(i + 2) * VN + k]), a3 = fabs(LU[(i + 3) * VN + k]);
            if (a0 > pv) { pv = a0; p = i; }
            if (a1 > pv) { pv = a1; p = i + 1; }
            if (a2 > pv) { pv = a2; p = i + 2; }
            if (a3 > pv) { pv = a3; p = i + 3; }
        }
    }
    PLP_UNROLL
    for (; i < n; ++i) {
        const double a = fabs(LU[i * VN + k]);
        if (a > pv) { pv = a; p = i; }
    }
    *pv_out = pv;
    if (!(pv > 1e-13 * big)) return false;
    if constexpr (NF > 0) {   // (static indices only: the arrays are registers)
        PLP_UNROLL
        for (int i = k + 1; i < n; ++i) {
            if (i == p) {
                PLP_UNROLL
                for (int j = 0; j < n; ++j) {
                    const double t = LU[k * VN + j];
                    LU[k * VN + j] = LU[i * VN + j];
                    LU[i * VN + j] = t;
                }
                const double t = perm[k];
                perm[k] = perm[i];
                perm[i] = t;
            }
        }
    } else if (p != k) {
        int j = 0;
        for (; j + 4 <= n; j += 4) {
            const double a0 = LU[k * VN + j], a1 = LU[k * VN + j + 1], a2 = LU[k * VN + j + 2], a3 = LU[k * VN + j + 3];
            const double b0 = LU[p * VN + j], b1 = LU[p * VN + j + 1], b2 = LU[p * VN + j + 2], b3 = LU[p * VN + j + 3];
            LU[k * VN + j] = b0; LU[k * VN + j + 1] = b1; LU[k * VN + j + 2] = b2; LU[k * VN + j + 3] = b3;
            LU[p * VN + j] = a0; LU[p * VN + j + 1] = a1; LU[p * VN + j + 2] = a2; LU[p * VN + j + 3] = a3;
        }
        for (; j < n; ++j) {
            const double t = LU[k * VN + j];
            LU[k * VN + j] = LU[p * VN + j];
            LU[p * VN + j] = t;
        }
        const double t = perm[k];
        perm[k] = perm[p];
        perm[p] = t;
    }
    return true;
}
static PLP_HD void lu_rows(int n, int k, Vec LU, int first, int step) {
    const double inv = 1.0 / LU[k * VN + k];
    PLP_UNROLL
    for (int i = k + 1 + first; i < n; i += step) {
        const double f = LU[i * VN + k] * inv;
        LU[i * VN + k] = f;
        row_axpy(LU, i * VN, LU, k * VN, f, k + 1, n);
    }
}
// The whole factorisation on one lane.  Written out, not as a loop over the steps above: through their pointer / by-step
// interface the register instance (NF > 0) no longer has its workspace split into registers -- all of it goes to scratch
// memory (measured: the small verify kernel 60 -> 100 us per 100 000 LPs).
static PLP_HD bool lu_factor(int n, Vec LU, Vec perm, double* pivot_ratio) {
    double big = 0.0, pmin = 1e300;
    PLP_UNROLL
    for (int k = 0; k < n; ++k)
        PLP_UNROLL
        for (int j = 0; j < n; ++j) big = fmax(big, fabs(LU[k * VN + j]));
    if (!(big > 0.0)) return false;
    PLP_UNROLL
    for (int k = 0; k < n; ++k) perm[k] = k;
    PLP_UNROLL
    for (int k = 0; k < n; ++k) {
        int p = k;
        double pv = fabs(LU[k * VN + k]);
        PLP_UNROLL
        for (int i = k + 1; i < n; ++i) {
            const double a = fabs(LU[i * VN + k]);
            if (a > pv) { pv = a; p = i; }
        }
        if (!(pv > 1e-13 * big)) return false;
        pmin = fmin(pmin, pv);
        if constexpr (NF > 0) {   // (static indices only: the arrays are registers)
            PLP_UNROLL
            for (int i = k + 1; i < n; ++i) {
                if (i == p) {
                    PLP_UNROLL
                    for (int j = 0; j < n; ++j) {
                        const double t = LU[k * VN + j];
                        LU[k * VN + j] = LU[i * VN + j];
                        LU[i * VN + j] = t;
                    }
                    const double t = perm[k];
                    perm[k] = perm[i];
                    perm[i] = t;
                }
            }
        } else if (p != k) {
            PLP_UNROLL
            for (int j = 0; j < n; ++j) {
                const double t = LU[k * VN + j];
                LU[k * VN + j] = LU[p * VN + j];
                LU[p * VN + j] = t;
            }
            const double t = perm[k];
            perm[k] = perm[p];
            perm[p] = t;
        }
        const double inv = 1.0 / LU[k * VN + k];
        PLP_UNROLL
        for (int i = k + 1; i < n; ++i) {
            const double f = LU[i * VN + k] * inv;
            LU[i * VN + k] = f;
            row_axpy(LU, i * VN, LU, k * VN, f, k + 1, n);
        }
    }
    *pivot_ratio = pmin / big;
    return true;
}
// s - sum_{j0 <= j < j1} a[aoff + j * astep] b[j], one fma after the other in index order.  In the LDS instances (NF == 0: the
// loop does not unroll) the loads of four terms are issued together: element by element every term waits for an LDS round trip
// of its own, and the leader lane of an LP runs these chains alone (the triangular solves were a third of the kernel).
static PLP_HD double dot_sub(double s, Vec a, int aoff, int astep, Vec b, int j0, int j1) {
    int j = j0;
    if constexpr (NF == 0) {
        for (; j + 4 <= j1; j += 4) {
            const double a0 = a[aoff + j * astep], a1 = a[aoff + (j + 1) * astep], a2 = a[aoff + (j + 2) * astep], a3 = a[aoff + (j + 3) * astep];
            const double b0 = b[j], b1 = b[j + 1], b2 = b[j + 2], b3 = b[j + 3];
            s = fma(-a0, b0, s);
            s = fma(-a1, b1, s);
            s = fma(-a2, b2, s);
            s = fma(-a3, b3, s);
        }
    }
    PLP_UNROLL
    for (; j < j1; ++j) s = fma(-a[aoff + j * astep], b[j], s);
    return s;
}
// z = M^-1 r; t: n doubles of work space
static PLP_HD void lu_solve(int n, Vec LU, Vec perm, Vec r, Vec z, Vec t) {
    PLP_UNROLL
    for (int k = 0; k < n; ++k) t[k] = dot_sub(pick(r, n, (int)perm[k]), LU, k * VN, 1, t, 0, k);
    PLP_UNROLL
    for (int k = n - 1; k >= 0; --k) z[k] = dot_sub(t[k], LU, k * VN, 1, z, k + 1, n) / LU[k * VN + k];
}
static PLP_HD void lu_solve_t(int n, Vec LU, Vec perm, Vec r, Vec z, Vec t) {
    PLP_UNROLL
    for (int k = 0; k < n; ++k) t[k] = dot_sub(r[k], LU, k, VN, t, 0, k) / LU[k * VN + k];
    PLP_UNROLL
    for (int k = n - 1; k >= 0; --k) t[k] = dot_sub(t[k], LU, k, VN, t, k + 1, n);
    PLP_UNROLL
    for (int k = 0; k < n; ++k) put(z, n, (int)perm[k], t[k]);
}

// z = M^-1 r (trans: M^-T r) by ONE instruction stream whatever `trans` is: two lanes of a wavefront solve the vertex' system and
// the multipliers' side by side (plp_verify.hip).  The numbers are lu_solve's / lu_solve_t's: the same fma chains, and the
// divisions the other variant does not have are by 1.0.
static PLP_HD void lu_solve_any(int n, Vec LU, Vec perm, Vec r, Vec z, Vec t, bool trans) {
    const int sr = trans ? 1 : VN, sc = trans ? VN : 1;   // entry (k, j) of the triangular factors: LU[k * sr + j * sc]
    for (int k = 0; k < n; ++k) {
        const int ik = (int)perm[k];
        const double dk = LU[k * VN + k];
        t[k] = dot_sub(r[trans ? k : ik], LU, k * sr, sc, t, 0, k) / (trans ? dk : 1.0);
    }
    for (int k = n - 1; k >= 0; --k) {
        const double dk = LU[k * VN + k];
        t[k] = dot_sub(t[k], LU, k * sr, sc, t, k + 1, n) / (trans ? 1.0 : dk);
    }
    for (int k = 0; k < n; ++k) z[trans ? (int)perm[k] : k] = t[k];
}

// entry (k, j) of the basis matrix: row basis[k] of G, or e_j0 for the free variable j0 = -1 - basis[k]
static PLP_HD double basis_entry(const LpView& lp, Vec basis, int k, int j) {
    const int v = (int)basis[k];
    return v >= 0 ? lp.g(v, j) : ((j == -1 - v) ? 1.0 : 0.0);
}
// z = M^-1 r (trans: M^-T r) with up to `rounds` rounds of refinement, residuals accumulated in double-double against the
// basis matrix kept beside its factorisation (ws[O_M0 ..): no global memory inside these dependent chains); a round whose
// correction is below the last bits of z ends it.  rounds = 0 for a well-conditioned basis (smallest pivot above 1e-4 of the
// largest entry): LU with partial pivoting is backward stable, the residuals of both systems are a few ulps of |M| |z| as they
// stand -- which is all the certificate's conclusion (an optimal basis, its value) rests on; the refinement is for bases with
// rows a hair apart, where it keeps the multipliers' signs and the vertex' coordinates meaningful.
static PLP_HD void solve_refined(int n, double* ws, Vec r, Vec z, bool trans, int rounds, bool have0 = false) {
    const Vec LU = at(ws, O_LU), M0 = at(ws, O_M0), perm = at(ws, O_PERM), t = at(ws, O_T), rr = at(ws, O_RR), dz = at(ws, O_DZ);
    if (have0) { }   // (z already holds the plain solve: lu_solve_any)
    else if (trans) lu_solve_t(n, LU, perm, r, z, t);
    else lu_solve(n, LU, perm, r, z, t);
    PLP_UNROLL
    for (int it = 0; it < rounds; ++it) {
        PLP_UNROLL
        for (int k = 0; k < n; ++k) {
            dd s = dd_make(r[k]);
            PLP_UNROLL
            for (int j = 0; j < n; ++j) s = dd_sub(s, two_prod(trans ? M0[j * VN + k] : M0[k * VN + j], z[j]));
            rr[k] = dd_to_double(s);
        }
        if (trans) lu_solve_t(n, LU, perm, rr, dz, t);
        else lu_solve(n, LU, perm, rr, dz, t);
        double dmax = 0.0, zmax = 0.0;
        PLP_UNROLL
        for (int j = 0; j < n; ++j) {
            const double zj = z[j] + dz[j];
            z[j] = zj;
            dmax = fmax(dmax, fabs(dz[j]));
            zmax = fmax(zmax, fabs(zj));
        }
        if (!(dmax > 2e-16 * zmax)) break;
    }
}

// h_i - G_i.v (a plain fma chain in column order) and |G_i|_inf.  In the instances whose column count is a run-time value the
// loop does not unroll and every element would wait for the one before it -- a global-memory round trip each; four loads are
// issued ahead of their fmas instead (the chain itself is unchanged: same operations, same order).
static PLP_HD double row_slack(const LpView& lp, int i, Vec v, double* gmax_out) {
    const int n = ncols(lp);
    double s = lp.hh(i), gmax = 0.0;
    int j = 0;
    if constexpr (NF == 0) {
        for (; j + 4 <= n; j += 4) {
            const double g0 = lp.g(i, j), g1 = lp.g(i, j + 1), g2 = lp.g(i, j + 2), g3 = lp.g(i, j + 3);
            const double v0 = v[j], v1 = v[j + 1], v2 = v[j + 2], v3 = v[j + 3];
            s = fma(-g0, v0, s);
            s = fma(-g1, v1, s);
            s = fma(-g2, v2, s);
            s = fma(-g3, v3, s);
            gmax = fmax(fmax(gmax, fmax(fabs(g0), fabs(g1))), fmax(fabs(g2), fabs(g3)));
        }
    }
    PLP_UNROLL
    for (; j < n; ++j) {
        const double gij = lp.g(i, j);
        s = fma(-gij, v[j], s);
        gmax = fmax(gmax, fabs(gij));
    }
    *gmax_out = gmax;
    return s;
}

// ---------------------------------------------------------------------------------------------------- the certificate
// One row against a vertex z (xs = max(1, |z|_inf)): slack h_i - G_i.z >= -2e-14 max(|h_i|, |G_i|_inf xs).  A plain fma chain:
// its rounding (n / 2 ulps of |G_i| |z| at worst, typically one or two) stays below the tolerance.
static PLP_HD bool row_feasible(const LpView& lp, int i, Vec z, double xs) {
    double gmax;
    const double s = row_slack(lp, i, z, &gmax);
    return !(s < -V_TOL_PRIMAL * fmax(gmax * xs, fabs(lp.hh(i))));
}
// the same, and the row's share of LpView::h_scale() into *hs (the pass has the row in hand: far_vertex_hs needs no pass of its own --
// on data 1e5 from the origin every vertex is beyond V_FAR, and a leader lane walking all m x n entries for the scale doubled
// the kernel's time there)
static PLP_HD bool row_feasible_hs(const LpView& lp, int i, Vec z, double xs, double* hs) {
    double gmax;
    const double s = row_slack(lp, i, z, &gmax);
    const double ah = fabs(lp.hh(i));
    if (gmax > 0.0 && ah > *hs * gmax) *hs = ah / gmax;
    return !(s < -V_TOL_PRIMAL * fmax(gmax * xs, ah));
}

// row k of the basis matrix (both copies) and of its right-hand side; false: an index out of range
static PLP_HD bool basis_row(const LpView& lp, bool have_xref, double* ws, int k) {
    const Vec basis = at(ws, O_BAS), xref = at(ws, O_X), LU = at(ws, O_LU), M0 = at(ws, O_M0), rhs = at(ws, O_RHS);
    const int n = ncols(lp), m = lp.m;
    const int v = (int)basis[k];
    if (v >= 0) {
        if (v >= m) return false;
        rhs[k] = lp.hh(v);
    } else {
        const int j0 = -1 - v;
        if (j0 < 0 || j0 >= n) return false;
        rhs[k] = have_xref ? pick(xref, n, j0) : 0.0;
    }
    int j = 0;
    double gm = 0.0;
    if constexpr (NF == 0) {   // (four loads in flight: see row_slack)
        for (; j + 4 <= n; j += 4) {
            const double e0 = basis_entry(lp, basis, k, j), e1 = basis_entry(lp, basis, k, j + 1);
            const double e2 = basis_entry(lp, basis, k, j + 2), e3 = basis_entry(lp, basis, k, j + 3);
            LU[k * VN + j] = e0; LU[k * VN + j + 1] = e1; LU[k * VN + j + 2] = e2; LU[k * VN + j + 3] = e3;
            M0[k * VN + j] = e0; M0[k * VN + j + 1] = e1; M0[k * VN + j + 2] = e2; M0[k * VN + j + 3] = e3;
            gm = fmax(fmax(gm, fmax(fabs(e0), fabs(e1))), fmax(fabs(e2), fabs(e3)));
        }
    }
    PLP_UNROLL
    for (; j < n; ++j) {
        const double e = basis_entry(lp, basis, k, j);
        LU[k * VN + j] = e;
        M0[k * VN + j] = e;
        gm = fmax(gm, fabs(e));
    }
    at(ws, O_GM)[k] = gm;
    return true;
}
// -c into the workspace (O_V): entry j
static PLP_HD void cost_entry(const LpView& lp, double* ws, int j) { at(ws, O_V)[j] = -lp.cc(j); }
// what follows the factorisation (pr = smallest pivot / largest entry): vertex, value, multipliers
// (presolved: z and y hold the plain solves already)
static PLP_HD bool vertex_and_dual_finish(const LpView& lp, bool want_dual, double* ws, double pr, double* fun, double* xs_out,
                                          bool presolved = false) {
    const Vec basis = at(ws, O_BAS);
    const int n = ncols(lp);
    const Vec rhs = at(ws, O_RHS), z = at(ws, O_Z), y = at(ws, O_Y), nc_ = at(ws, O_V), gm = at(ws, O_GM);
    const int rounds = pr < 1e-4 ? 3 : 0;
    solve_refined(n, ws, rhs, z, false, rounds, presolved);
    double zmax = 0.0;
    PLP_UNROLL
    for (int j = 0; j < n; ++j) {
        if (!(fabs(z[j]) < 1e300)) return false;
        zmax = fmax(zmax, fabs(z[j]));
    }
    *xs_out = zmax > 1.0 ? zmax : 1.0;
    dd f = dd_make(0.0);
    double cmax = 0.0;   // (nc_ = -c: cost_entry, before this function)
    PLP_UNROLL
    for (int j = 0; j < n; ++j) {
        const double cj = -nc_[j];
        f = dd_add(f, two_prod(cj, z[j]));
        cmax = fmax(cmax, fabs(cj));
    }
    *fun = dd_to_double(f);
    if (!want_dual) return true;
    // dual: M' y = -c
    solve_refined(n, ws, nc_, y, true, rounds, presolved);
    if (rounds == 0) {   // a multiplier between -1e-9 and the tolerance: rounding of the plain solve, or real?  refine, then judge
        bool grey = false;
        PLP_UNROLL
        for (int k = 0; k < n; ++k) {
            const double yk = basis[k] >= 0.0 ? y[k] : -fabs(y[k]);
            grey = grey | ((yk < 0.0) & (yk > -1e-9 * cmax));
        }
        if (grey) solve_refined(n, ws, nc_, y, true, 3);
    }
    PLP_UNROLL
    for (int k = 0; k < n; ++k) {
        const double yk = y[k];
        if (!(fabs(yk) < 1e300)) return false;
        if (basis[k] >= 0.0) {
            if (yk * gm[k] < -V_TOL_DUAL * cmax) return false;
        } else if (fabs(yk) > V_TOL_DUAL * cmax) return false;
    }
    return true;
}

static PLP_HD bool vertex_and_dual(const LpView& lp, bool have_xref, bool want_dual, double* ws, double* fun, double* xs_out) {
    const int n = ncols(lp);
    if (n > VN || n < 1) return false;
    PLP_UNROLL
    for (int k = 0; k < n; ++k)
        if (!basis_row(lp, have_xref, ws, k)) return false;
    PLP_UNROLL
    for (int j = 0; j < n; ++j) cost_entry(lp, ws, j);
    double pr = 1.0;
    if (!lu_factor(n, at(ws, O_LU), at(ws, O_PERM), &pr)) return false;
    return vertex_and_dual_finish(lp, want_dual, ws, pr, fun, xs_out);
}

// status: V_OPT / V_UNBND as the engine reported it; basis list in the workspace, entries n, n + 1 (V_UNBND): position in the list of the variable
// the ray moves and its sign.  true: certified; for V_OPT x[0..n) = the polished vertex, *fun = c.x.
static PLP_HD bool certify(const LpView& lp, int status, bool have_xref, double* ws, double* x, double* fun) {
    const int n = ncols(lp), m = lp.m;
    const Vec basis = at(ws, O_BAS);
    double xs = 1.0, f = 0.0;
    if (!vertex_and_dual(lp, have_xref, status != V_UNBND, ws, &f, &xs)) return false;
    const Vec z = at(ws, O_Z);
    if (status == V_UNBND) {
        const int e = (int)basis[n];
        const Vec w = at(ws, O_Y), ru = at(ws, O_V);
        if (e < 0 || e >= n) return false;
        PLP_UNROLL
        for (int k = 0; k < n; ++k) ru[k] = 0.0;
        ru[e] = basis[e] >= 0.0 ? -1.0 : basis[n + 1];  // the slack of an active row grows / the free variable moves by its sign
        solve_refined(n, ws, ru, w, false, 3);
        dd cw = dd_make(0.0), cz = dd_make(0.0);
        double wmax = 0.0;
        PLP_UNROLL
        for (int j = 0; j < n; ++j) {
            if (!(fabs(w[j]) < 1e300)) return false;
            wmax = fmax(wmax, fabs(w[j]));
            cw = dd_add(cw, two_prod(lp.cc(j), w[j]));
            cz = dd_add(cz, two_prod(lp.cc(j), z[j]));
        }
        if (!(cw.hi < 0.0)) return false;
        const double big = V_BIG * lp.scale();
        PLP_UNROLL
        for (int i = 0; i < m; ++i) {
            dd gw = dd_make(0.0), sl = dd_make(lp.hh(i));
            double gmax = 0.0;
            PLP_UNROLL
            for (int j = 0; j < n; ++j) {
                const double gij = lp.g(i, j);
                gw = dd_add(gw, two_prod(gij, w[j]));
                sl = dd_sub(sl, two_prod(gij, z[j]));
                gmax = fmax(gmax, fabs(gij));
            }
            const double tol = fmax(gmax * xs, fabs(lp.hh(i)));
            if (sl.hi < -V_TOL_PRIMAL * tol) return false;  // the vertex itself must be feasible
            bool inb = false;                                // rows of the basis: G_k.w = 0 (or -1) by construction
            PLP_UNROLL
            for (int k = 0; k < n; ++k) inb = inb | ((int)basis[k] == i);
            if (inb || !(gw.hi > 1e-14 * gmax * wmax)) continue;  // (below the rounding of w: not a blocking row)
            if (dd_lt_d(sl, 0.0)) sl = dd_make(0.0);
            const dd t = dd_div(sl, gw);                     // the ray meets row i here ...
            const dd obj = dd_add(cz, dd_mul(t, cw));
            if (!(fabs(obj.hi) > big)) return false;         // ... before the objective is out of range
        }
        return true;
    }
    PLP_UNROLL
    for (int i = 0; i < m; ++i)
        if (!row_feasible(lp, i, z, xs)) return false;
    PLP_UNROLL
    for (int j = 0; j < n; ++j) x[j] = z[j];
    *fun = f;
    return true;
}

// ---- a basis read off an engine's x (engines that do not hand over theirs).  Candidates: the rows whose slack at x is below
// 1e-9 of their scale, kept sorted by (slack, row) in the workspace -- at most KC of them (more: no basis, the LP goes to
// the careful engine).  cn: their number so far (KC + 1: overflow).
static PLP_HD void cand_add(double* ws, int& cn, double s, int i) {
    const Vec cs = at(ws, O_CS), ci = at(ws, O_CI);
    if (cn >= KC) { cn = KC + 1; return; }
    if constexpr (NF > 0) {   // (static indices only)
        int pos = 0;
        PLP_UNROLL
        for (int k = 0; k < KC; ++k) pos += ((k < cn) && ((cs[k] < s) || ((cs[k] == s) && (ci[k] < (double)i)))) ? 1 : 0;
        PLP_UNROLL
        for (int k = KC - 1; k > 0; --k) {
            if ((k > pos) & (k <= cn)) { cs[k] = cs[k - 1]; ci[k] = ci[k - 1]; }
        }
        PLP_UNROLL
        for (int k = 0; k < KC; ++k)
            if (k == pos) { cs[k] = s; ci[k] = (double)i; }
        ++cn;
        return;
    }
    int k = cn++;
    while (k > 0 && (cs[k - 1] > s || (cs[k - 1] == s && ci[k - 1] > (double)i))) {
        cs[k] = cs[k - 1];
        ci[k] = ci[k - 1];
        --k;
    }
    cs[k] = s;
    ci[k] = (double)i;
}
// slack of row i at x (plain fma chain) and whether the row is a candidate (xs = max(1, |x|_inf))
static PLP_HD bool row_candidate(const LpView& lp, int i, Vec x, double xs, double* slack) {
    double gmax;
    const double s = row_slack(lp, i, x, &gmax);
    *slack = s;
    return (gmax > 0.0) & (s <= 1e-9 * fmax(gmax * xs, fabs(lp.hh(i))));
}
// The candidates in order, as long as they are linearly independent of the ones taken so far (a copy of a taken row adds
// nothing to the cone they span: elimination with column pivoting in the LDS instances, modified Gram-Schmidt in the register
// ones); free variables x_j held where they are complete the basis when x lies on a face rather than at a vertex.  At a degenerate vertex the choice may not be the dual-feasible one: the certificate
// then fails and the LP goes to the careful engine.  (Q shares the workspace of the factorisation, which comes after it.)
static PLP_HD bool select_basis(const LpView& lp, double* ws, int cn) {
    const int n = ncols(lp);
    const Vec basis = at(ws, O_BAS);
    const Vec Q = at(ws, O_LU), v = at(ws, O_V), ci = at(ws, O_CI);  // Q: orthonormal rows spanning the accepted rows
    int nb = 0;
    if (cn > KC) return false;
    if (cn == n) {
        // exactly n candidates -- a non-degenerate vertex, the generic case -- ARE the basis; should they be dependent, the
        // factorisation says so (vertex_and_dual: singular) and the LP goes to the careful engine.  Saves the O(n^3)
        // orthogonalisation, the largest term of the certificate at n = 17.
        PLP_UNROLL
        for (int k = 0; k < n; ++k) basis[k] = ci[k];
        return true;
    }
    if constexpr (NF > 0) {   // the same with static indices only: rows of Q beyond nb are zero, "row nb" is a select
        PLP_UNROLL
        for (int e = 0; e < n * VN; ++e) Q[e] = 0.0;
        PLP_UNROLL
        for (int q0 = 0; q0 < KC; ++q0) {
            if ((q0 < cn) & (nb < n)) {
                const int bi = (int)ci[q0];
                double nrm0 = 0.0, nrm1 = 0.0;
                PLP_UNROLL
                for (int j = 0; j < n; ++j) { const double g = lp.g(bi, j); v[j] = g; nrm0 = fma(g, g, nrm0); }
                PLP_UNROLL
                for (int q = 0; q < n; ++q) {   // (rows q >= nb of Q are zero: they change nothing)
                    double dq = 0.0;
                    PLP_UNROLL
                    for (int j = 0; j < n; ++j) dq = fma(Q[q * VN + j], v[j], dq);
                    PLP_UNROLL
                    for (int j = 0; j < n; ++j) v[j] = fma(-dq, Q[q * VN + j], v[j]);
                }
                PLP_UNROLL
                for (int j = 0; j < n; ++j) nrm1 = fma(v[j], v[j], nrm1);
                if (nrm1 > 1e-12 * nrm0) {
                    const double inv = 1.0 / sqrt(nrm1);
                    PLP_UNROLL
                    for (int q = 0; q < n; ++q) {
                        if (q == nb) {
                            PLP_UNROLL
                            for (int j = 0; j < n; ++j) Q[q * VN + j] = v[j] * inv;
                            basis[q] = (double)bi;
                        }
                    }
                    ++nb;
                }
            }
        }
        PLP_UNROLL
        for (int round = 0; round < n; ++round) {   // complete with free variables (unit vectors), most independent first
            if (nb < n) {
                int bj = -1;
                double bn = 0.0;
                PLP_UNROLL
                for (int j0 = 0; j0 < n; ++j0) {
                    double r2 = 1.0;
                    PLP_UNROLL
                    for (int q = 0; q < n; ++q) r2 = fma(-Q[q * VN + j0], Q[q * VN + j0], r2);
                    if (r2 > bn) { bn = r2; bj = j0; }
                }
                if (bj < 0 || !(bn > 1e-12)) return false;
                double nrm1 = 0.0;
                PLP_UNROLL
                for (int j = 0; j < n; ++j) v[j] = (j == bj) ? 1.0 : 0.0;
                PLP_UNROLL
                for (int q = 0; q < n; ++q) {
                    double dq = 0.0;
                    PLP_UNROLL
                    for (int j = 0; j < n; ++j) dq = (j == bj) ? Q[q * VN + j] : dq;
                    PLP_UNROLL
                    for (int j = 0; j < n; ++j) v[j] = fma(-dq, Q[q * VN + j], v[j]);
                }
                PLP_UNROLL
                for (int j = 0; j < n; ++j) nrm1 = fma(v[j], v[j], nrm1);
                if (!(nrm1 > 1e-12)) return false;
                const double inv = 1.0 / sqrt(nrm1);
                PLP_UNROLL
                for (int q = 0; q < n; ++q) {
                    if (q == nb) {
                        PLP_UNROLL
                        for (int j = 0; j < n; ++j) Q[q * VN + j] = v[j] * inv;
                        basis[q] = (double)(-1 - bj);
                    }
                }
                ++nb;
            }
        }
        return nb == n;
    }
    // One elimination with column pivoting over the candidates in their order: a row is reduced by the rows taken so far (it
    // keeps zeros in their pivot columns), what is left of it outside those columns decides -- below 1e-6 of the row: dependent
    // (a twin of a taken row adds nothing to the cone), otherwise taken, its largest entry there the next pivot column.  The
    // columns that are never pivoted on are the free variables.  O(nb^2 n / 2) on the leader lane, no dot products; it replaced
    // a modified Gram-Schmidt plus a separate completion (0.18 of the 0.25 ms a generic (64,16) LP batch spent in this kernel).
    const Vec taken = at(ws, O_T), pcol = at(ws, O_RR);
    (void)v;
    for (int j = 0; j < n; ++j) taken[j] = 0.0;
    for (int q0 = 0; q0 < cn && nb < n; ++q0) {
        const int bi = (int)ci[q0];
        double rmax = 0.0;
        {   // (the row from global memory, four loads in flight: see row_slack)
            int j = 0;
            for (; j + 4 <= n; j += 4) {
                const double g0 = lp.g(bi, j), g1 = lp.g(bi, j + 1), g2 = lp.g(bi, j + 2), g3 = lp.g(bi, j + 3);
                Q[nb * VN + j] = g0; Q[nb * VN + j + 1] = g1; Q[nb * VN + j + 2] = g2; Q[nb * VN + j + 3] = g3;
                rmax = fmax(fmax(rmax, fmax(fabs(g0), fabs(g1))), fmax(fabs(g2), fabs(g3)));
            }
            for (; j < n; ++j) { const double g = lp.g(bi, j); Q[nb * VN + j] = g; rmax = fmax(rmax, fabs(g)); }
        }
        for (int q = 0; q < nb; ++q) {
            const int cq = (int)pcol[q];
            const double f = Q[nb * VN + cq] / Q[q * VN + cq];
            row_axpy(Q, nb * VN, Q, q * VN, f, 0, n);
            Q[nb * VN + cq] = 0.0;
        }
        int pj = -1;
        double pa = 0.0;
        for (int j = 0; j < n; ++j) {
            const double a = fabs(Q[nb * VN + j]);
            if ((taken[j] == 0.0) & (a > pa)) { pa = a; pj = j; }
        }
        if (pj < 0 || !(pa > 1e-6 * rmax)) continue;
        taken[pj] = 1.0;
        pcol[nb] = (double)pj;
        basis[nb++] = (double)bi;
    }
    for (int j = 0; j < n; ++j)
        if (taken[j] == 0.0) basis[nb++] = (double)(-1 - j);
    if (nb != n) return false;
    return true;
}
static PLP_HD double x_scale(const LpView& lp, Vec x, bool* finite) {
    double xs = 1.0;
    *finite = true;
    PLP_UNROLL
    for (int j = 0; j < ncols(lp); ++j) {
        if (!(fabs(x[j]) < 1e300)) *finite = false;
        xs = fmax(xs, fabs(x[j]));
    }
    return xs;
}
// (the point: ws[O_X ..))
static PLP_HD bool basis_from_x(const LpView& lp, double* ws) {
    const Vec x = at(ws, O_X);
    bool fin;
    const double xs = x_scale(lp, x, &fin);
    if (!fin) return false;
    int cn = 0;
    PLP_UNROLL
    for (int i = 0; i < lp.m; ++i) {
        double s;
        if (row_candidate(lp, i, x, xs, &s)) cand_add(ws, cn, s, i);
    }
    return select_basis(lp, ws, cn);
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

// ratio test of column e (entering upwards): the leaving row (-1: none) and the step
PLP_HD int careful_ratio(const CarefulMem& M, const CarefulState& S, int e, bool negate, bool bland, dd* step) {
    int r = -1;
    dd rmin = dd_make(0.0);
    for (int i = 0; i < S.m; ++i) {
        if (!M.ract(i)) continue;
        dd a = M.get(i, e);
        if (negate) a = dd_neg(a);
        if (!dd_gt_d(a, C_TOL_PIV)) continue;
        dd bi = M.get(i, VNC);
        if (dd_lt_d(bi, 0.0)) bi = dd_make(0.0);
        const dd q = dd_div(bi, a);
        if (r < 0 || dd_lt(q, rmin) || (bland && dd_eq(q, rmin) && M.rv(i) < M.rv(r))) {
            rmin = q;
            r = i;
        }
    }
    *step = rmin;
    return r;
}

// `bigstep`: BIG times the scale of the data in the units of the (equilibrated) dictionary -- how far a step may go and count
PLP_HD int careful_run(const CarefulMem& M, CarefulState& S, double bigstep) {
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
        if (e < 0) {
            // no column above the engines' tolerance.  Those between the rounding level and it are judged by what they buy:
            // the step they allow times the rate, against C_TOL_GAIN of max(1, |objective|) -- provided the step ends on a row
            // and moves x by no more than `bigstep` (a direction that only pays beyond the range of the data is none)
            dd obj = dd_abs(M.get(m, VNC));
            const double thr = C_TOL_GAIN * fmax(1.0, obj.hi);
            double gain = 0.0;
            for (int j = 0; j < S.nc; ++j) {
                if (S.coldead[j]) continue;
                const dd dj = M.get(m, j), aj = dd_abs(dj);
                const bool fr = (unsigned)S.colvar[j] < (unsigned)S.n;
                const bool grey = fr ? dd_gt_d(aj, C_TOL_NOISE) : dd_lt_d(dj, -C_TOL_NOISE);
                if (!grey) continue;
                dd step;
                const int r = careful_ratio(M, S, j, dd_gt_d(dj, 0.0), false, &step);
                if (r < 0) continue;   // unbounded at a rate below the engines' tolerance: not a direction (HiGHS agrees)
                // how far x moves per unit of the entering variable: the rows that hold the free variables (and itself, if free)
                double dx = fr ? 1.0 : 0.0;
                for (int i = 0; i < m; ++i)
                    if ((unsigned)M.rv(i) < (unsigned)S.n) dx = fmax(dx, fabs(M.get(i, j).hi));
                if (!(step.hi * dx <= bigstep)) continue;   // the step leaves the range of the data: not taken either
                const double gj = aj.hi * step.hi;
                if (gj > thr && gj > gain) { gain = gj; e = j; }
            }
            if (e < 0) return V_OPT;
        }
        if (S.iters >= S.maxit) return V_ITER;
        if (dd_gt_d(M.get(m, e), 0.0)) {  // free variable entering downwards: x := -x
            for (int i = 0; i < m; ++i) M.set(i, e, dd_neg(M.get(i, e)));
            M.set(m, e, dd_neg(M.get(m, e)));
            if (S.carry) M.set(m + 1, e, dd_neg(M.get(m + 1, e)));
            S.colsgn[e] = -S.colsgn[e];
        }
        dd rmin;
        const int r = careful_ratio(M, S, e, false, bland, &rmin);
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
    const double bigstep = V_BIG * (cmax > 0.0 ? lp.scale() / cmax : 1.0);   // (rows and cost are equilibrated: distances in |G_i|_inf units)
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
        st = careful_run(M, S, bigstep);
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
    st = careful_run(M, S, bigstep);
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

// what every caller does with an optimum: out of range -> unbounded.  Out of range: the VALUE beyond BIG times the scale of
// the data.  (xmax = |x|_inf is no longer looked at: a far vertex counted while rows an ulp apart still met 1e16 away; with the
// entries below 1e-9 dropped as HiGHS drops them those corners are gone, and on a long optimal face -- the ball of an unbounded
// polytope slides along it -- the vertex an engine ends on is an accident of its path, the value is not.)
PLP_HD int range_rule_c(const LpView& lp, int status, double fun, double xmax, double cmax) {   // cmax = |c|_inf
    (void)xmax;
    if (status != V_OPT) return status;
    if (!(fabs(fun) > V_BIG * cmax)) return status;   // (the scale is at least |c|_inf: a pass over the rows only where it can matter)
    return fabs(fun) > V_BIG * lp.scale() ? V_UNBND : status;
}
PLP_HD int range_rule(const LpView& lp, int status, double fun, double xmax) {
    if (status != V_OPT) return status;
    return range_rule_c(lp, status, fun, xmax, lp.c_inf());
}

}  // namespace verify
}  // namespace plp
