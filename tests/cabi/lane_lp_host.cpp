// Host build of polytope_amd/csrc/plp_lane_lp.hpp (the one-LP-per-lane engine of the fused reduce, d = 3) against the
// oracle's dictionary simplex: TEST INFRASTRUCTURE (compiled with g++ by tests/test_lane_lp_host.py, links
// oracle/libplp_oracle.so).  For every polytope of a packed batch: Chebyshev centre by the oracle, then the 2d box LPs
// (F3, polytope.py:1367-1396) and the m redundancy LPs (F2, :1145-1156) in the centre-relative form the kernel solves,
// each by both engines.
#include <stdint.h>
#include <string.h>

#include "../../polytope_amd/csrc/plp_lane_lp.hpp"

extern "C" {
int plpo_cheby(int m, int d, const double* A, const double* b, double* r, double* xc, int* iters);
int plpo_lp_solve(int m, int n, const double* c, const double* G, const double* h, double* x, double* fun, int* iters);
}

namespace {
struct Stats {
    long long lps, retry, status_diff, opt_both, unb_both;
    double max_abs_diff;
    long long iters_sum, iters_max;
    long long hist[16];          // iterations per LP, clipped
    long long worst_poly, worst_lp;
    long long f3_rounds, f3_round_iters;   // lock-step model: 64 LPs (16 polytopes x 4) per round, max iterations
    long long f2_rounds, f2_round_iters;
};

template <int M>
int run_lane(const double* A, const double* beta, const double* c, double* x, int* iters) {
    plp::lane::Lp3 S;
    plp::lane::solve3<M>(
        S, c[0], c[1], c[2], true,
        [&](int i, double& a0, double& a1, double& a2) { a0 = A[i * 3]; a1 = A[i * 3 + 1]; a2 = A[i * 3 + 2]; },
        [&](int i) { return beta[i]; }, [](bool p) { return p; });
    x[0] = S.x0; x[1] = S.x1; x[2] = S.x2;
    *iters = S.iters;
    return S.status;
}
}  // namespace

// A[B][16][3], b[B][16] (rows beyond m[p] zero).  stats: see Stats.  Returns 0.
extern "C" int lane_check(long long B, const double* A, const double* b, const int* mrows, double* stats_out, int which) {
    Stats st;
    memset(&st, 0, sizeof st);
    long long f3_it[64], nf3 = 0, f2_it[64], nf2 = 0;
    for (long long p = 0; p < B; ++p) {
        const int m = mrows ? mrows[p] : 16;
        const double* Ap = A + p * 48;
        const double* bp = b + p * 16;
        double r, xc[3];
        const int s1 = plpo_cheby(m, 3, Ap, bp, &r, xc, nullptr);
        if (s1 != 0 || !(r > 1e-7)) continue;
        double Az[48], beta[16];
        memset(Az, 0, sizeof Az);
        for (int i = 0; i < 16; ++i) {
            beta[i] = 0.0;
            if (i < m) {
                for (int k = 0; k < 3; ++k) Az[i * 3 + k] = Ap[i * 3 + k];
                const double s = fma(Ap[i * 3 + 2], xc[2], fma(Ap[i * 3 + 1], xc[1], Ap[i * 3] * xc[0]));
                beta[i] = fmax(bp[i] - s, 0.0);
            }
        }
        auto one = [&](const double* c, const double* bt, bool f3) {
            double xl[3], xo[3], fo = 0.0;
            int itl = 0;
            const int sl = run_lane<16>(Az, bt, c, xl, &itl);
            const int so = plpo_lp_solve(m, 3, c, Az, bt, xo, &fo, nullptr);
            ++st.lps;
            st.iters_sum += itl;
            if (itl > st.iters_max) st.iters_max = itl;
            ++st.hist[itl < 15 ? itl : 15];
            if (f3) { f3_it[nf3++] = itl; if (nf3 == 64) { long long mx = 0; for (auto v : f3_it) mx = v > mx ? v : mx; st.f3_round_iters += mx; ++st.f3_rounds; nf3 = 0; } }
            else { f2_it[nf2++] = itl; if (nf2 == 64) { long long mx = 0; for (auto v : f2_it) mx = v > mx ? v : mx; st.f2_round_iters += mx; ++st.f2_rounds; nf2 = 0; } }
            if (sl == plp::ST_RETRY) { ++st.retry; return; }
            if (sl != so) { ++st.status_diff; st.worst_poly = p; return; }
            if (sl == 0) {
                ++st.opt_both;
                const double fl = fma(c[2], xl[2], fma(c[1], xl[1], c[0] * xl[0]));
                const double d = fabs(fl - fo) / fmax(1.0, fabs(fo));   // (nearly unbounded polytopes: optima of 1e4 and more)
                if (d > st.max_abs_diff) { st.max_abs_diff = d; st.worst_lp = p; }
            } else if (sl == 3) ++st.unb_both;
        };
        if (which & 1)
            for (int it = 0; it < 6; ++it) {
                double c[3] = {0, 0, 0};
                c[it >> 1] = (it & 1) ? -1.0 : 1.0;
                one(c, beta, true);
            }
        if (which & 2)
            for (int k = 0; k < m; ++k) {
                double c[3] = {-Az[k * 3], -Az[k * 3 + 1], -Az[k * 3 + 2]};
                double bt[16];
                memcpy(bt, beta, sizeof bt);
                bt[k] = fmax((bp[k] + 0.1) - fma(Ap[k * 3 + 2], xc[2], fma(Ap[k * 3 + 1], xc[1], Ap[k * 3] * xc[0])), 0.0);
                one(c, bt, false);
            }
    }
    double* o = stats_out;
    o[0] = (double)st.lps; o[1] = (double)st.retry; o[2] = (double)st.status_diff; o[3] = (double)st.opt_both;
    o[4] = (double)st.unb_both; o[5] = st.max_abs_diff; o[6] = (double)st.iters_sum; o[7] = (double)st.iters_max;
    for (int i = 0; i < 16; ++i) o[8 + i] = (double)st.hist[i];
    o[24] = (double)st.worst_poly; o[25] = (double)st.worst_lp;
    o[26] = (double)st.f3_rounds; o[27] = (double)st.f3_round_iters; o[28] = (double)st.f2_rounds; o[29] = (double)st.f2_round_iters;
    return 0;
}

// one LP  min c.x  s.t.  A x <= beta (16 row slots, zero rows beyond m), from x = 0: status, x, iterations
extern "C" int lane_solve_one(const double* A16, const double* beta16, const double* c, double* x, int* iters) {
    return run_lane<16>(A16, beta16, c, x, iters);
}

// ---- d = 4 (walk4): the same comparison, A[B][16][4]
namespace {
int run_lane4(const double* A, const double* beta, const double* c, double* x, int* iters) {
    plp::lane::Lp4 S;
    plp::lane::walk4(
        S, c[0], c[1], c[2], c[3], true,
        [&](int i, double& a0, double& a1, double& a2, double& a3) { a0 = A[i * 4]; a1 = A[i * 4 + 1]; a2 = A[i * 4 + 2]; a3 = A[i * 4 + 3]; },
        [&](double d0, double d1, double d2, double d3, double x0, double x1, double x2, double x3, double tolp, double& bs,
            double& bd, int& bi) {
            for (int i = 0; i < 16; ++i)
                plp::lane::ratio_row4(A[i * 4], A[i * 4 + 1], A[i * 4 + 2], A[i * 4 + 3], beta[i], i, d0, d1, d2, d3, x0, x1, x2, x3,
                                      tolp, bs, bd, bi);
        },
        [](bool p) { return p; });
    x[0] = S.x0; x[1] = S.x1; x[2] = S.x2; x[3] = S.x3;
    *iters = S.iters;
    return S.status;
}
}  // namespace

extern "C" int lane_check4(long long B, const double* A, const double* b, const int* mrows, double* stats_out, int which) {
    Stats st;
    memset(&st, 0, sizeof st);
    for (long long p = 0; p < B; ++p) {
        const int m = mrows ? mrows[p] : 16;
        const double* Ap = A + p * 64;
        const double* bp = b + p * 16;
        double r, xc[4];
        const int s1 = plpo_cheby(m, 4, Ap, bp, &r, xc, nullptr);
        if (s1 != 0 || !(r > 1e-7)) continue;
        double Az[64], beta[16], sxc[16];
        memset(Az, 0, sizeof Az);
        for (int i = 0; i < 16; ++i) {
            beta[i] = 0.0;
            sxc[i] = 0.0;
            if (i < m) {
                for (int k = 0; k < 4; ++k) Az[i * 4 + k] = Ap[i * 4 + k];
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s = fma(Ap[i * 4 + k], xc[k], s);
                sxc[i] = s;
                beta[i] = fmax(bp[i] - s, 0.0);
            }
        }
        auto one = [&](const double* c, const double* bt) {
            double xl[4], xo[4], fo = 0.0;
            int itl = 0;
            const int sl = run_lane4(Az, bt, c, xl, &itl);
            const int so = plpo_lp_solve(m, 4, c, Az, bt, xo, &fo, nullptr);
            ++st.lps;
            st.iters_sum += itl;
            if (itl > st.iters_max) st.iters_max = itl;
            ++st.hist[itl < 15 ? itl : 15];
            if (sl == plp::ST_RETRY) { ++st.retry; return; }
            if (sl != so) { ++st.status_diff; st.worst_poly = p; return; }
            if (sl == 0) {
                ++st.opt_both;
                double fl = 0.0;
                for (int k = 0; k < 4; ++k) fl = fma(c[k], xl[k], fl);
                const double d = fabs(fl - fo) / fmax(1.0, fabs(fo));
                if (d > st.max_abs_diff) { st.max_abs_diff = d; st.worst_lp = p; }
            } else if (sl == 3) ++st.unb_both;
        };
        if (which & 1)
            for (int it = 0; it < 8; ++it) {
                double c[4] = {0, 0, 0, 0};
                c[it >> 1] = (it & 1) ? -1.0 : 1.0;
                one(c, beta);
            }
        if (which & 2)
            for (int k = 0; k < m; ++k) {
                double c[4] = {-Az[k * 4], -Az[k * 4 + 1], -Az[k * 4 + 2], -Az[k * 4 + 3]};
                double bt[16];
                memcpy(bt, beta, sizeof bt);
                bt[k] = fmax((bp[k] + 0.1) - sxc[k], 0.0);
                one(c, bt);
            }
    }
    double* o = stats_out;
    o[0] = (double)st.lps; o[1] = (double)st.retry; o[2] = (double)st.status_diff; o[3] = (double)st.opt_both;
    o[4] = (double)st.unb_both; o[5] = st.max_abs_diff; o[6] = (double)st.iters_sum; o[7] = (double)st.iters_max;
    for (int i = 0; i < 16; ++i) o[8 + i] = (double)st.hist[i];
    o[24] = (double)st.worst_poly; o[25] = (double)st.worst_lp;
    return 0;
}

extern "C" int lane_solve_one4(const double* A16, const double* beta16, const double* c, double* x, int* iters) {
    return run_lane4(A16, beta16, c, x, iters);
}
