// sanitize_main.cpp -- TEST INFRASTRUCTURE (tests/test_sanitizers.py): one program, built with -fsanitize=address,undefined,
// that drives the host-compilable native code of the repo over seeded inputs:
//   * oracle/plp_oracle.c + plp_oracle_q.c (the CPU oracle: LPs, certificate, binary128 fallback, reduce, boxes, hull pass),
//   * tests/cabi/lane_lp_host.cpp (polytope_amd/csrc/plp_lane_lp.hpp: the one-LP-per-lane walk, host build),
//   * tests/cabi/verify_host.cpp (polytope_amd/csrc/plp_verify.hpp: certificate + careful double-double engine, host build),
//   * polytope_amd/csrc/plp_quickhull_host.hip (quickhull's main loop: plain C++, included below) over a CPU stand-in of the
//     device session it talks to (plp_hull_*: here the oracle's hull pass on host arrays -- only so that the loop can run
//     under the sanitizers; the product's session is plp_hull.hip).
// Exit code 0 and nothing on stderr = no heap / stack / UB finding.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/plp.h"

extern "C" {
int plpo_lp_solve(int m, int n, const double* c, const double* G, const double* h, double* x, double* fun, int* iters);
int plpo_cheby(int m, int d, const double* A, const double* b, double* r, double* xc, int* iters);
int plpo_bounding_box(int m, int d, const double* A, const double* b, double* lb, double* ub, int* nlp);
int plpo_reduce(int m, int d, const double* A, const double* b, double abs_tol, uint64_t* keep, double* bout, double* r, double* xc, int* nlp);
void plpo_hull_reassign(int64_t N, int d, const double* X, int32_t* owner, double* dist, const uint8_t* dead, int new_id0, int n_new,
                        const double* normals, const double* offsets, double abs_tol, int64_t* argmax, double* maxd, int64_t* count);
int lane_check(long long B, const double* A, const double* b, const int* mrows, double* stats_out, int which);
int lane_check4(long long B, const double* A, const double* b, const int* mrows, double* stats_out, int which);
int plpv_certify_from_x(int kind, int m, int n, int side, const double* c, const double* G, const double* h, const double* x_in,
                        double* x_out, double* fun_out, int* basis_out);
int plpv_careful(int kind, int m, int n, int side, const double* c, const double* G, const double* h, double* x, double* fun, int* iters);
}

// ---- CPU stand-in of the device session (plp_hull.hip) for the quickhull loop
struct plp_hull {
    int64_t N;
    int d;
    std::vector<double> X, dist;
    std::vector<int32_t> owner;
    std::vector<uint8_t> dead;
    int next_id;
};
extern "C" {
int plp_hull_create(plp_ctx*, int64_t N, int d, const double* X, plp_hull** out) {
    plp_hull* h = new plp_hull();
    h->N = N; h->d = d;
    h->X.assign(X, X + (size_t)N * d);
    h->dist.assign((size_t)N, 0.0);
    h->owner.assign((size_t)N, 0);   // facet 0: the virtual facet that owns every point at creation
    h->dead.assign(1, 0);
    h->next_id = 1;
    *out = h;
    return 0;
}
int plp_hull_destroy(plp_hull* h) { delete h; return 0; }
int plp_hull_drop(plp_hull* h, int64_t n, const int64_t* idx) {
    for (int64_t i = 0; i < n; ++i) h->owner[(size_t)idx[i]] = -1;
    return 0;
}
int plp_hull_reassign(plp_hull* h, int n_dead, const int32_t* dead_ids, int n_new, const double* normals, const double* offsets,
                      double abs_tol, int32_t* new_id0, int64_t* argmax, double* maxd, int64_t* count) {
    for (int i = 0; i < n_dead; ++i) h->dead[(size_t)dead_ids[i]] = 1;
    *new_id0 = h->next_id;
    plpo_hull_reassign(h->N, h->d, h->X.data(), h->owner.data(), h->dist.data(), h->dead.data(), h->next_id, n_new, normals, offsets,
                       abs_tol, argmax, maxd, count);
    h->next_id += n_new;
    h->dead.resize((size_t)h->next_id, 0);
    return 0;
}
int plp_hull_read(plp_hull* h, int32_t* owner, double* dist) {
    if (owner) memcpy(owner, h->owner.data(), sizeof(int32_t) * (size_t)h->N);
    if (dist) memcpy(dist, h->dist.data(), sizeof(double) * (size_t)h->N);
    return 0;
}
}

#include "../../polytope_amd/csrc/plp_quickhull_host.hip"

static uint64_t g_s = 88172645463325252ull;
static double urand() {
    g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17;
    return (double)(g_s >> 11) * (1.0 / 9007199254740992.0);
}
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

static void make_poly(int m, int d, double* A, double* b, bool dup) {
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) { A[i * d + k] = nrand(); s += A[i * d + k] * A[i * d + k]; }
        s = 1.0 / sqrt(s);
        for (int k = 0; k < d; ++k) A[i * d + k] *= s;
        b[i] = 0.5 + urand();
    }
    if (m >= 2 * d)
        for (int k = 0; k < 2 * d; ++k) {
            for (int j = 0; j < d; ++j) A[k * d + j] = 0.0;
            A[k * d + (k % d)] = k < d ? 1.0 : -1.0;
            b[k] = 2.0;
        }
    if (dup && m >= 4)
        for (int t = 0; t < m / 4; ++t) {
            const int i = (int)(urand() * m) % m, j = (int)(urand() * m) % m;
            const double eps[5] = {0.0, 1e-16, 1e-9, 1e-7, 1e-5}, sh[5] = {0.0, 0.0, 1e-7, -1e-7, 0.1};
            const double e = eps[(int)(urand() * 5) % 5];
            for (int k = 0; k < d; ++k) A[j * d + k] = A[i * d + k] + e * nrand();
            b[j] = b[i] + sh[(int)(urand() * 5) % 5];
        }
}

int main() {
    int fails = 0;
    // ---- the oracle and the verifier's host build on random and hair-apart polytopes, d = 2..16
    const int shapes[][2] = {{2, 8}, {3, 16}, {4, 24}, {6, 32}, {8, 40}, {13, 30}, {16, 64}};
    for (const auto& sh : shapes) {
        const int d = sh[0], m = sh[1];
        std::vector<double> A((size_t)m * d), b(m), x(d + 1), lb(d), ub(d), bout(m), xc(d);
        for (int rep = 0; rep < 24; ++rep) {
            make_poly(m, d, A.data(), b.data(), rep % 2 == 1);
            double r, fun;
            int nlp = 0, it = 0;
            uint64_t keep[4];
            const int st = plpo_cheby(m, d, A.data(), b.data(), &r, xc.data(), &it);
            plpo_bounding_box(m, d, A.data(), b.data(), lb.data(), ub.data(), &nlp);
            plpo_reduce(m, d, A.data(), b.data(), 1e-7, keep, bout.data(), &r, xc.data(), &nlp);
            std::vector<double> c(d, 0.0), xo(d), xk(d);
            c[rep % d] = (rep & 1) ? -1.0 : 1.0;
            const int s2 = plpo_lp_solve(m, d, c.data(), A.data(), b.data(), x.data(), &fun, &it);
            double fk = 0.0;
            const int s3 = plpv_careful(0, m, d, 0, c.data(), A.data(), b.data(), xk.data(), &fk, &it);
            if (s2 != s3) { fprintf(stderr, "careful engine: status %d, oracle %d (d %d m %d rep %d)\n", s3, s2, d, m, rep); ++fails; }
            if (s2 == 0) {
                double fo = 0.0;
                plpv_certify_from_x(0, m, d, 0, c.data(), A.data(), b.data(), x.data(), xo.data(), &fo, nullptr);
            }
            (void)st;
        }
    }
    // ---- the lane walk's host build against the oracle ((16,3) and (16,4) records)
    {
        const long long B = 300;
        std::vector<double> A3((size_t)B * 48), A4((size_t)B * 64), b(B * 16), stats(64);
        for (long long p = 0; p < B; ++p) make_poly(16, 3, &A3[(size_t)p * 48], &b[(size_t)p * 16], p % 5 == 4);
        lane_check(B, A3.data(), b.data(), nullptr, stats.data(), 3);
        for (long long p = 0; p < B; ++p) make_poly(16, 4, &A4[(size_t)p * 64], &b[(size_t)p * 16], p % 5 == 4);
        lane_check4(B, A4.data(), b.data(), nullptr, stats.data(), 3);
    }
    // ---- quickhull's main loop (host) over the stand-in session: Gaussian clouds, d = 2..5
    for (int d = 2; d <= 5; ++d) {
        const int64_t N = d <= 3 ? 3000 : 600;
        std::vector<double> X((size_t)N * d);
        for (auto& v : X) v = nrand();
        // a start simplex: the first d + 1 points; translate so that its centroid is the origin (quickhull.py:188-192)
        std::vector<int64_t> simplex(d + 1);
        std::vector<double> cen(d, 0.0);
        for (int i = 0; i <= d; ++i) { simplex[i] = i; for (int k = 0; k < d; ++k) cen[k] += X[(size_t)i * d + k] / (d + 1); }
        for (int64_t i = 0; i < N; ++i) for (int k = 0; k < d; ++k) X[(size_t)i * d + k] -= cen[k];
        plp_qh_result* res = nullptr;
        static char fake_ctx[8];   // (the stand-in session never looks at it)
        const int rc = plp_quickhull_run(reinterpret_cast<plp_ctx*>(fake_ctx), N, d, X.data(), simplex.data(), 1e-7, nullptr, &res);
        if (rc != 0) { fprintf(stderr, "plp_quickhull_run rc %d at d = %d: %s\n", rc, d, plp_quickhull_last_error()); ++fails; continue; }
        int64_t nf = 0, iters = 0, made = 0;
        plp_qh_result_sizes(res, &nf, &iters, &made);
        std::vector<double> nrm((size_t)nf * d), off(nf);
        std::vector<int64_t> verts((size_t)nf * d);
        plp_qh_result_copy(res, nrm.data(), off.data(), verts.data());
        // every point inside every facet
        double worst = 0.0;
        for (int64_t i = 0; i < N; ++i)
            for (int64_t f = 0; f < nf; ++f) {
                double s = -off[f];
                for (int k = 0; k < d; ++k) s += nrm[(size_t)f * d + k] * X[(size_t)i * d + k];
                if (s > worst) worst = s;
            }
        if (nf < d + 1 || worst > 1e-7) { fprintf(stderr, "hull d = %d: %lld facets, a point %.3e outside\n", d, (long long)nf, worst); ++fails; }
        plp_qh_result_free(res);
    }
    printf("sanitize_main: %d failures\n", fails);
    return fails ? 1 : 0;
}
