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
    int basis[VNMAX + 2];
    if (!Cert<VNMAX>::basis_from_x(lp, x_in, basis)) return -1;
    if (basis_out) memcpy(basis_out, basis, sizeof(int) * n);
    return Cert<VNMAX>::certify(lp, V_OPT, basis, x_in, x_out, fun_out) ? 1 : 0;
}

// basis mode (status 0 / 3); basis[n + 2] as the oracle's plpo_lp_solve_raw hands it over
int plpv_certify_basis(int kind, int m, int n, int side, const double* c, const double* G, const double* h, int status,
                       const int* basis, const double* xref, double* x_out, double* fun_out) {
    const LpView lp = view(kind, m, n, side, c, G, h);
    // (the small instance where it applies: what the device launches for these shapes)
    if (n <= 5) return Cert<5>::certify(lp, status, basis, xref, x_out, fun_out) ? 1 : 0;
    return Cert<VNMAX>::certify(lp, status, basis, xref, x_out, fun_out) ? 1 : 0;
}

int plpv_careful(int kind, int m, int n, int side, const double* c, const double* G, const double* h, double* x, double* fun,
                 int* iters) {
    const LpView lp = view(kind, m, n, side, c, G, h);
    const size_t nd = careful_doubles_per_lp(m);
    std::vector<double> hi(nd), lo(nd);
    std::vector<int> ri(m + 2);
    CarefulMem M{hi.data(), lo.data(), ri.data(), 1};
    double f = 0.0;
    const int st = range_rule(lp, careful_solve(lp, M, x, &f, iters), f);
    *fun = f;
    return st;
}
}
