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

// ... with m row slots (the 17..32-row tiles of the device kernel)
extern "C" int lane_solve_one4m(int m, const double* A, const double* beta, const double* c, double* x, int* iters) {
    plp::lane::Lp4 S;
    plp::lane::walk4(
        S, c[0], c[1], c[2], c[3], true,
        [&](int i, double& a0, double& a1, double& a2, double& a3) { a0 = A[i * 4]; a1 = A[i * 4 + 1]; a2 = A[i * 4 + 2]; a3 = A[i * 4 + 3]; },
        [&](double d0, double d1, double d2, double d3, double x0, double x1, double x2, double x3, double tolp, double& bs,
            double& bd, int& bi) {
            for (int i = 0; i < m; ++i)
                plp::lane::ratio_row4(A[i * 4], A[i * 4 + 1], A[i * 4 + 2], A[i * 4 + 3], beta[i], i, d0, d1, d2, d3, x0, x1, x2, x3,
                                      tolp, bs, bd, bi);
        },
        [](bool p) { return p; });
    x[0] = S.x0; x[1] = S.x1; x[2] = S.x2; x[3] = S.x3;
    *iters = S.iters;
    return S.status;
}

// ---- warm starts: the redundancy LP of row k started from the best of the vertices the polytope's box LPs ended on
// (instead of from the centre).  stats: [0] LPs compared, [1] status differences, [2] max objective difference,
// [3..18] histogram of cold iterations, [19..34] of warm iterations (rows whose LP says "redundant" only),
// [35] lock-step model: groups of 34 such LPs, sum of max cold iterations, [36] of max warm iterations, [37] groups,
// [38] warm walks handed back, [39] cold walks handed back
extern "C" int lane_warm_stats(long long B, const double* A, const double* b, double* o) {
    for (int i = 0; i < 48; ++i) o[i] = 0.0;
    long long gc[34], gw[34], ng = 0;
    for (long long p = 0; p < B; ++p) {
        const double* Ap = A + p * 48;
        const double* bp = b + p * 16;
        double r, xc[3];
        if (plpo_cheby(16, 3, Ap, bp, &r, xc, nullptr) != 0 || !(r > 1e-7)) continue;
        double beta[16];
        for (int i = 0; i < 16; ++i)
            beta[i] = fmax(bp[i] - fma(Ap[i * 3 + 2], xc[2], fma(Ap[i * 3 + 1], xc[1], Ap[i * 3] * xc[0])), 0.0);
        auto rows = [&](int i, double& a0, double& a1, double& a2) { a0 = Ap[i * 3]; a1 = Ap[i * 3 + 1]; a2 = Ap[i * 3 + 2]; };
        plp::lane::Lp3 V[6];
        for (int it = 0; it < 6; ++it) {
            double c[3] = {0, 0, 0};
            c[it >> 1] = (it & 1) ? -1.0 : 1.0;
            plp::lane::solve3<16>(V[it], c[0], c[1], c[2], true, rows, [&](int i) { return beta[i]; }, [](bool q) { return q; });
        }
        // [40] box LPs 4, 5 cold iterations (sum), [41] warm from the best of the vertices of LPs 0..3, [42] count
        for (int it = 4; it < 6; ++it) {
            double c[3] = {0, 0, (it & 1) ? -1.0 : 1.0};
            int best = -1;
            double bv = 0.0;
            for (int q = 0; q < 4; ++q) {
                if (V[q].status != plp::ST_OPT) continue;
                const double v = -(c[2] * V[q].x2);
                if (best < 0 || v > bv) { best = q; bv = v; }
            }
            if (best < 0 || V[it].status != plp::ST_OPT) continue;
            plp::lane::Lp3 W = V[best];
            W.iters = 0; W.ndeg = 0;
            plp::lane::solve3<16>(W, c[0], c[1], c[2], true, rows, [&](int i) { return beta[i]; }, [](bool q) { return q; }, true);
            o[40] += V[it].iters; o[41] += W.iters; o[42] += 1;
            o[43] = fmax(o[43], fabs(W.x2 - V[it].x2));
        }
        for (int k = 0; k < 16; ++k) {
            const double c[3] = {-Ap[k * 3], -Ap[k * 3 + 1], -Ap[k * 3 + 2]};
            auto bt = [&](int i) { return beta[i] + (i == k ? 0.1 : 0.0); };
            plp::lane::Lp3 C, W;
            plp::lane::solve3<16>(C, c[0], c[1], c[2], true, rows, bt, [](bool q) { return q; });
            int best = -1;
            double bv = 0.0;
            for (int it = 0; it < 6; ++it) {
                if (V[it].status != plp::ST_OPT) continue;
                const double v = -(c[0] * V[it].x0 + c[1] * V[it].x1 + c[2] * V[it].x2);
                if (best < 0 || v > bv) { best = it; bv = v; }
            }
            if (best < 0) continue;
            W = V[best];
            // row k itself among the active rows: its plane moves away by 0.1, the point stays on the others
            if (W.nact >= 1 && W.w0 == k) { W.w0 = W.w1; W.w1 = W.w2; --W.nact; }
            else if (W.nact >= 2 && W.w1 == k) { W.w1 = W.w2; --W.nact; }
            else if (W.nact >= 3 && W.w2 == k) { --W.nact; }
            W.iters = 0;
            W.ndeg = 0;
            plp::lane::solve3<16>(W, c[0], c[1], c[2], true, rows, bt, [](bool q) { return q; }, true);
            o[38] += W.status == plp::ST_RETRY;
            o[39] += C.status == plp::ST_RETRY;
            if (W.status == plp::ST_RETRY || C.status == plp::ST_RETRY) continue;
            o[0] += 1;
            if (W.status != C.status) { o[1] += 1; continue; }
            if (C.status != plp::ST_OPT) continue;
            const double fc = c[0] * C.x0 + c[1] * C.x1 + c[2] * C.x2, fw = c[0] * W.x0 + c[1] * W.x1 + c[2] * W.x2;
            o[2] = fmax(o[2], fabs(fc - fw));
            const double obj = -fc - beta[k];
            if (obj > 1e-7) continue;   // "keep": the presolve settles nearly all of these before any LP
            o[3 + (C.iters < 15 ? C.iters : 15)] += 1;
            o[19 + (W.iters < 15 ? W.iters : 15)] += 1;
            gc[ng] = C.iters; gw[ng] = W.iters;
            if (++ng == 34) {
                long long mc = 0, mw = 0;
                for (int q = 0; q < 34; ++q) { mc = gc[q] > mc ? gc[q] : mc; mw = gw[q] > mw ? gw[q] : mw; }
                o[35] += mc; o[36] += mw; o[37] += 1;
                ng = 0;
            }
        }
    }
    return 0;
}
