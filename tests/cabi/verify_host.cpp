// verify_host.cpp -- TEST INFRASTRUCTURE: polytope_amd/csrc/plp_verify.hpp (the verifier and the careful double-double LP
// engine of libplp_hip.so) compiled for the HOST, behind plain C entry points for tests/test_verify_host.py, which runs
// them against the oracle's certificate / binary128 engine.  The header is the same source the device compiles
// (explicit fma, -ffp-contract=off on both sides).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../polytope_amd/csrc/plp_verify.hpp"

using namespace plp::verify;

static LpView view(int kind, int m, int n, int side, const double* c, const double* G, const double* h) {
    LpView lp;
    lp.kind = kind; lp.m = m; lp.n = n; lp.side = side; lp.c = c; lp.G = G; lp.h = h;
    return lp;
}

extern "C" {

// x-mode: the basis is read off x_in; -> 1 certified (x_out, fun_out polished), 0 not, -1 no basis
int plpv_certify_from_x(int kind, int m, int n, int side, const double* c, const double* G, const double* h,
                        const double* x_in, double* x_out, double* fun_out, int* basis_out) {
    const LpView lp = view(kind, m, n, side, c, G, h);
    // (n <= 5: the instance with the column count fixed at compile time and static indices only -- what the device's
    // one-LP-per-lane kernel instantiates, its arrays in registers; here it must give what the general instance gives)
#define PLPV_FROM_X(CT)                                                                                    \
    {                                                                                                      \
        double ws[CT::WS_DOUBLES];                                                                         \
        for (int j = 0; j < n; ++j) ws[CT::O_X + j] = x_in[j];                                             \
        if (!CT::basis_from_x(lp, ws)) return -1;                                                          \
        if (basis_out) for (int k = 0; k < n; ++k) basis_out[k] = (int)ws[CT::O_BAS + k];                  \
        return CT::certify(lp, V_OPT, true, ws, x_out, fun_out) ? 1 : 0;                                   \
    }
    using C2 = Cert<2, 1, 2>;
    using C3 = Cert<3, 1, 3>;
    using C4 = Cert<4, 1, 4>;
    using C5 = Cert<5, 1, 5>;
    using CTG = Cert<VNMAX, 1>;
    if (n == 2) PLPV_FROM_X(C2)
    if (n == 3) PLPV_FROM_X(C3)
    if (n == 4) PLPV_FROM_X(C4)
    if (n == 5) PLPV_FROM_X(C5)
    PLPV_FROM_X(CTG)
#undef PLPV_FROM_X
}

// basis mode (status 0 / 3); basis[n + 2] as the oracle's plpo_lp_solve_raw hands it over
int plpv_certify_basis(int kind, int m, int n, int side, const double* c, const double* G, const double* h, int status,
                       const int* basis, const double* xref, double* x_out, double* fun_out) {
    const LpView lp = view(kind, m, n, side, c, G, h);
    // (the small instance where it applies: what the device launches for these shapes)
    if (n <= 5) {
        double ws[Cert<5, 1>::WS_DOUBLES];
        for (int k = 0; k < n + 2; ++k) ws[Cert<5, 1>::O_BAS + k] = basis[k];
        for (int j = 0; j < n; ++j) ws[Cert<5, 1>::O_X + j] = xref ? xref[j] : 0.0;
        return Cert<5, 1>::certify(lp, status, xref != nullptr, ws, x_out, fun_out) ? 1 : 0;
    }
    double ws[Cert<VNMAX, 1>::WS_DOUBLES];
    for (int k = 0; k < n + 2; ++k) ws[Cert<VNMAX, 1>::O_BAS + k] = basis[k];
    for (int j = 0; j < n; ++j) ws[Cert<VNMAX, 1>::O_X + j] = xref ? xref[j] : 0.0;
    return Cert<VNMAX, 1>::certify(lp, status, xref != nullptr, ws, x_out, fun_out) ? 1 : 0;
}

int plpv_careful(int kind, int m, int n, int side, const double* c, const double* G, const double* h, double* x, double* fun,
                 int* iters) {
    const LpView lp = view(kind, m, n, side, c, G, h);
    const size_t nd = careful_doubles_per_lp(m);
    std::vector<double> hi(nd), lo(nd);
    std::vector<int> ri(m + 2);
    CarefulMem M{hi.data(), lo.data(), ri.data(), 1};
    double f = 0.0;
    int st = careful_solve(lp, M, x, &f, iters);
    double xm = 0.0;
    if (st == V_OPT) for (int j = 0; j < n; ++j) xm = fmax(xm, fabs(x[j]));
    st = range_rule(lp, st, f, xm);
    *fun = f;
    return st;
}

// lu_solve_any (the device's side-by-side solves) against lu_solve / lu_solve_t on the n x n matrix M (row-major): number of
// entries of the two solutions that differ in any bit (0 expected), -1 if M is singular
int plpv_solve_any_check(int n, const double* M, const double* r) {
    using CT = Cert<VNMAX, 1>;
    double ws[CT::WS_DOUBLES], za[VNMAX], zb[VNMAX];
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j) ws[CT::O_LU + k * VNMAX + j] = M[k * n + j];
    double pr;
    if (!CT::lu_factor(n, CT::at(ws, CT::O_LU), CT::at(ws, CT::O_PERM), &pr)) return -1;
    for (int j = 0; j < n; ++j) ws[CT::O_RHS + j] = r[j];
    int bad = 0;
    for (int trans = 0; trans < 2; ++trans) {
        if (trans) CT::lu_solve_t(n, CT::at(ws, CT::O_LU), CT::at(ws, CT::O_PERM), CT::at(ws, CT::O_RHS), CT::at(ws, CT::O_Z), CT::at(ws, CT::O_T));
        else CT::lu_solve(n, CT::at(ws, CT::O_LU), CT::at(ws, CT::O_PERM), CT::at(ws, CT::O_RHS), CT::at(ws, CT::O_Z), CT::at(ws, CT::O_T));
        for (int j = 0; j < n; ++j) za[j] = ws[CT::O_Z + j];
        CT::lu_solve_any(n, CT::at(ws, CT::O_LU), CT::at(ws, CT::O_PERM), CT::at(ws, CT::O_RHS), CT::at(ws, CT::O_Y), CT::at(ws, CT::O_DZ), trans != 0);
        for (int j = 0; j < n; ++j) zb[j] = ws[CT::O_Y + j];
        for (int j = 0; j < n; ++j) bad += memcmp(&za[j], &zb[j], sizeof(double)) != 0;
    }
    return bad;
}
}
