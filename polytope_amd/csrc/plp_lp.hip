// plp_lp.hip -- batched stand-alone LP kernels (gfx950).
//
//   lp_kernel<N>    : B independent LPs  min c'x s.t. Gx<=h, x free   (solvers.py:76-106,149-158)
//   cheby_kernel<D> : B Chebyshev-ball LPs (form F1, polytope.py:1283-1288):
//                     c = -e_{d+1}, G = [A | sqrt(sum(A*A,1))], h = b
//
// One LP per lane group (GS lanes, GS >= rows), 64/GS LPs per wavefront, 256-thread
// workgroups, grid-stride over the batch.  Chebyshev batches with d <= 8 go to the four-rows-per-lane
// kernels of plp_cheby_r.hip; cheby_kernel here serves d > 8 (and PLP_CHEBY_1ROW=1).
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_simplex.hpp"

namespace plp {

// Chebyshev batches on the one-LP-per-wavefront engine (plp_wide.hip): d >= PLP_WIDE_MIN_D and either more than
// PLP_WIDE_MIN_M rows (the lane groups then hold one or two LPs per wavefront anyway and pay select chains for the
// entering column) or at most PLP_WIDE_SMALL_B LPs (every LP gets a wavefront slot at once: what counts is the latency
// of one LP, and a wave-uniform pivot is the shortest).  Measured with scripts/debug/wide_grid.py (round 3, after the
// pivot loop lost its register copies): (33..64 rows, d = 5..10) 15-40 % faster than the lane groups at B = 20 000,
// every shape with d >= 5 at B = 2 000; (<= 32 rows, B = 20 000) the lane groups stay 10-45 % ahead.
#ifndef PLP_WIDE_MIN_D
#define PLP_WIDE_MIN_D 5
#endif
#ifndef PLP_WIDE_MIN_M
#define PLP_WIDE_MIN_M 32
#endif
#ifndef PLP_WIDE_SMALL_B
#define PLP_WIDE_SMALL_B 4096
#endif

// Generic LP batches on the one-LP-per-wavefront engine (plp_lp_wide.hip): every batch with n = 5..16.  Measured against
// the lane-group kernels (scripts/debug/lp_wide_ab.py, 20 000 LPs, ms; x / status / iterations bitwise equal on every
// input): LPs that need phase 1 (64,16) 1.74 -> 0.34, (64,8) 0.59 -> 0.17, (32,12) 0.48 -> 0.17, (32,6) 0.21 -> 0.09,
// (16,5) 0.092 -> 0.063; origin-feasible ones (64,16) 0.43 -> 0.25, (32,6) a tie, (16,5) / (24,5) / (32,8) 10-30 % behind
// (four rows per lane pack several of those per wavefront) -- the mix of a batch is not known to the host, and the
// reference's LPs are posed in original coordinates, where the origin is rarely feasible.
// (A two-pass route for large batches of LPs with 32 rows and fewer -- lp_r_kernel for the origin-feasible ones, then this
// engine on what it hands over -- was measured: the feasible batches win their 15-27 % back, the two-phase ones lose
// 20-30 % to the extra pass ((32,6): 0.097 -> 0.116 ms); not kept.)
// PLP_LP_WIDE=0 / 1: never / always (A/B, tests)
static bool lp_wide_on(long long B, int m_max, int n) {
    if (n < 5 || n > MAX_D || m_max < 1 || m_max > MAX_M || B > 2147483647ll) return false;  // (what launch_lp_w takes)
    const char* w = getenv("PLP_LP_WIDE");
    if (w) return w[0] == '1';
    const char* one = getenv("PLP_LP_1ROW");
    if (one && one[0] == '1') return false;
    return true;
}

template <int N>
__global__ __launch_bounds__(BLOCK) void lp_kernel(long long B, int m_max, int gs,
                                                   const double* __restrict__ c,
                                                   const double* __restrict__ G,
                                                   const double* __restrict__ h,
                                                   const int* __restrict__ mrows,
                                                   double* __restrict__ x, double* __restrict__ fun,
                                                   int* __restrict__ status, int* __restrict__ iters,
                                                   int retry_only) {
    constexpr int NC = N + 1;  // + phase-1 artificial
    const Grp g(gs);
    const int gpb = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    if (retry_only) {  // normally nothing was handed over: one round trip of status loads, then leave
        bool any = false;
        for (long long base = (long long)blockIdx.x * gpb; base < B; base += (long long)gridDim.x * gpb)
            any = any | ((base + gib < B) && status[base + gib] == ST_RETRY);
        if (!__syncthreads_or(any)) return;
    }
    for (long long base = (long long)blockIdx.x * gpb; base < B; base += (long long)gridDim.x * gpb) {
        const long long lp = base + gib;
        bool valid = lp < B;
        // second pass after lp_r_kernel: only the LPs it handed over (status ST_RETRY: phase 1 or Bland needed)
        if (retry_only) {
            valid = valid && status[lp] == ST_RETRY;
            if (!__any(valid)) continue;
        }
        const int m = valid ? (mrows ? mrows[lp] : m_max) : 0;
        const int i = g.gl;
        const bool has_row = valid && i < m;
        Simplex<NC, true> S;
        S.reset(N, m, i);
        double cc[N];
        bool finite = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            cc[j] = valid ? c[lp * N + j] : 0.0;
            finite = finite && isfinite(cc[j]);
        }
        bool zero = true;
        double hi = 0.0;
        if (has_row) {
            const double* Gr = G + (lp * m_max + i) * N;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                S.T[j] = Gr[j];
                zero = zero && (S.T[j] == 0.0);
                finite = finite && isfinite(S.T[j]);
            }
            hi = h[lp * m_max + i];
            finite = finite && isfinite(hi);
        }
        S.beta = hi;
        S.rowact = has_row && !zero;
        const bool infeasible0 = grp_ballot(has_row && zero && hi < -TOL_FEAS, g) != 0;  // 0 <= h_i < 0
        if (has_row && zero) S.beta = 0.0;
        const bool bad = grp_ballot(!finite, g) != 0 || m > gs;
        const bool need_p1 = grp_ballot(S.rowact && hi < 0.0, g) != 0;
        S.set_col(N, ID_T);
        if (need_p1) {
            S.T[N] = S.rowact ? -1.0 : 0.0;
            S.cost[N] = 1.0;
#pragma unroll
            for (int j = 0; j < N; ++j) S.cost2[j] = cc[j];
            S.mode = M_INIT;
            S.init_col = N;
            S.init_q = S.beta;
            S.init_elig = S.rowact;
            S.mode_after_init = M_P1;
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) S.cost[j] = cc[j];
            S.dead = 1u << N;
            S.mode = M_P2;
        }
        if (!valid) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (bad) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }

        S.run(g);

        // gather x: variable j sits in the row whose basic id is j (0 when nonbasic)
        const bool ok = S.status == ST_OPT;
        const double mine = S.x_value();
        const bool holds = S.holds_x();
        double f = 0.0;
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const uint64_t ob = grp_ballot(holds && S.rowvar == j, g);
            const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
            const double xj = ob ? v : 0.0;
            f = fma(cc[j], xj, f);
            if (valid && g.gl == 0) x[lp * N + j] = ok ? xj : qnan;
        }
        if (valid && g.gl == 0) {
            fun[lp] = ok ? f : qnan;
            status[lp] = S.status;
            if (iters) iters[lp] = S.iters;
        }
    }
}

template <int D>
__global__ __launch_bounds__(BLOCK) void cheby_kernel(long long B, int m_max, int gs,
                                                      const double* __restrict__ A,
                                                      const double* __restrict__ b,
                                                      const int* __restrict__ mrows,
                                                      double* __restrict__ r, double* __restrict__ xc,
                                                      int* __restrict__ status) {
    constexpr int NC = D + 1;
    const Grp g(gs);
    const int gpb = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    for (long long base = (long long)blockIdx.x * gpb; base < B; base += (long long)gridDim.x * gpb) {
        const long long p = base + gib;
        const bool valid = p < B;
        const int m = valid ? (mrows ? mrows[p] : m_max) : 0;
        const int i = g.gl;
        const bool has_row = valid && i < m;
        Simplex<NC, false> S;
        S.reset(NC, m, i);
        bool finite = true;
        double bi = 0.0, nrm2 = 0.0;
        if (has_row) {
            const double* Ar = A + (p * m_max + i) * D;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                S.T[k] = Ar[k];
                nrm2 = nrm2 + S.T[k] * S.T[k];
                finite = finite && isfinite(S.T[k]);
            }
            bi = b[p * m_max + i];
            finite = finite && isfinite(bi);
        }
        const double nrm = sqrt(nrm2);
        S.T[D] = nrm;
        S.beta = bi;
        const bool zero = !(nrm > 0.0);
        S.rowact = has_row && !zero;
        if (!S.rowact) { S.beta = 0.0; S.T[D] = 0.0; }
        const bool infeasible0 = grp_ballot(has_row && zero && bi < -TOL_FEAS, g) != 0;
        const bool bad = grp_ballot(!finite, g) != 0 || m > gs;
        S.cost[D] = -1.0;
        S.mode = M_INIT;
        S.init_col = D;
        S.init_q = bi / nrm;
        S.init_elig = S.rowact;
        S.mode_after_init = M_P2;
        if (!valid || bad) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }

        S.run(g);

        const bool ok = S.status == ST_OPT;
        const double mine = S.x_value();
        const bool holds = S.holds_x();
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const uint64_t ob = grp_ballot(holds && S.rowvar == j, g);
            const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
            const double xj = ok ? (ob ? v : 0.0) : qnan;
            if (valid && g.gl == 0) {
                if (j < D) xc[p * D + j] = xj; else r[p] = xj;
            }
        }
        if (valid && g.gl == 0) status[p] = S.status;
    }
}

static inline int pick_grid(long long B, int gs) {
    const long long gpb = BLOCK / gs;
    long long blocks = (B + gpb - 1) / gpb;
    const long long cap = 1ll << 20;  // one LP tile per block: the dispatcher balances uneven pivot counts
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int N>
static void launch_lp_n(long long B, int m_max, int gs, const double* c, const double* G, const double* h,
                        const int* mrows, double* x, double* fun, int* status, int* iters, int retry, hipStream_t st) {
    int blocks = pick_grid(B, gs);
    if (retry && blocks > 256 * 8) blocks = 256 * 8;  // second pass: mostly status reads, a grid-stride sweep
    hipLaunchKernelGGL(lp_kernel<N>, dim3(blocks), dim3(BLOCK), 0, st, B, m_max, gs, c, G, h, mrows, x,
                       fun, status, iters, retry);
}

template <int D>
static void launch_cheby_d(long long B, int m_max, int gs, const double* A, const double* b, const int* mrows,
                           double* r, double* xc, int* status, hipStream_t st) {
    hipLaunchKernelGGL(cheby_kernel<D>, dim3(pick_grid(B, gs)), dim3(BLOCK), 0, st, B, m_max, gs, A, b, mrows, r,
                       xc, status);
}

#define PLP_CASE_N(K) case K: launch_lp_n<K>(B, m_max, gs, c, G, h, mrows, x, fun, status, iters, retry, st); break;
#define PLP_CASE_D(K) case K: launch_cheby_d<K>(B, m_max, gs, A, b, mrows, r, xc, status, st); break;

// phase 1: the fast kernels only -- returns 0 with *more = 1 when LPs they hand over (status ST_RETRY) may exist and the
// caller has to look; phase 2: the general kernel over exactly those.  (launch_lp below = both, back to back.)  The
// synchronous host entry point uses the phases for small batches: the normally idle general pass is a launch saved.
int launch_lp_phase(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                    double* x, double* fun, int* status, int* iters, hipStream_t st, int phase, int* more) {
    if (more) *more = 0;
    const char* lds = getenv("PLP_LDS");
    if (m_max > MAX_M || (lds && lds[0] == '1')) {
        if (phase == 2) return 0;
        return launch_lp_lds(B, m_max, n, c, G, h, mrows, x, fun, status, iters, st);
    }
    const int gs = group_size_for(m_max);
    if (gs < 0 || n < 1 || n > MAX_D + 1) return 2;
    if (lp_wide_on(B, m_max, n)) {  // both phases in one kernel: nothing is handed over
        if (phase == 2) return 0;
        if (launch_lp_w(B, m_max, n, c, G, h, mrows, x, fun, status, iters, st) == 0) return 0;
    }
    const char* one = getenv("PLP_LP_1ROW");
    int retry = 0;
    if (phase == 2) retry = 1;
    else if (!(one && one[0] == '1') && launch_lp_r(B, m_max, n, c, G, h, mrows, x, fun, status, iters, st) == 0) {
        if (more) *more = 1;
        return 0;
    }
    switch (n) {
        PLP_CASE_N(1) PLP_CASE_N(2) PLP_CASE_N(3) PLP_CASE_N(4) PLP_CASE_N(5) PLP_CASE_N(6)
        PLP_CASE_N(7) PLP_CASE_N(8) PLP_CASE_N(9) PLP_CASE_N(10) PLP_CASE_N(11) PLP_CASE_N(12)
        PLP_CASE_N(13) PLP_CASE_N(14) PLP_CASE_N(15) PLP_CASE_N(16) PLP_CASE_N(17)
        default: return 2;
    }
    return 0;
}

int launch_lp(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
              double* x, double* fun, int* status, int* iters, hipStream_t st) {
    // more than 64 rows (or PLP_LDS=1: A/B, tests): the LDS-resident engine, one LP per wavefront (plp_lds.hip)
    const char* lds = getenv("PLP_LDS");
    if (m_max > MAX_M || (lds && lds[0] == '1')) return launch_lp_lds(B, m_max, n, c, G, h, mrows, x, fun, status, iters, st);
    const int gs = group_size_for(m_max);
    if (gs < 0 || n < 1 || n > MAX_D + 1) return 2;
    if (lp_wide_on(B, m_max, n) && launch_lp_w(B, m_max, n, c, G, h, mrows, x, fun, status, iters, st) == 0) return 0;
    // n <= 8: LPs whose origin is feasible (no phase 1) are solved by the four-rows-per-lane fast path
    // (plp_cheby_r.hip); it marks the others ST_RETRY and the launch below redoes exactly those
    // (PLP_LP_1ROW=1 keeps everything on this file's kernel: A/B, tests).
    const char* one = getenv("PLP_LP_1ROW");
    int retry = 0;
    if (!(one && one[0] == '1') && launch_lp_r(B, m_max, n, c, G, h, mrows, x, fun, status, iters, st) == 0) retry = 1;
    switch (n) {
        PLP_CASE_N(1) PLP_CASE_N(2) PLP_CASE_N(3) PLP_CASE_N(4) PLP_CASE_N(5) PLP_CASE_N(6)
        PLP_CASE_N(7) PLP_CASE_N(8) PLP_CASE_N(9) PLP_CASE_N(10) PLP_CASE_N(11) PLP_CASE_N(12)
        PLP_CASE_N(13) PLP_CASE_N(14) PLP_CASE_N(15) PLP_CASE_N(16) PLP_CASE_N(17)
        default: return 2;
    }
    return 0;
}

int launch_cheby(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                 double* xc, int* status, hipStream_t st) {
    const char* lds = getenv("PLP_LDS");
    if (m_max > MAX_M || (lds && lds[0] == '1')) return launch_cheby_lds(B, m_max, d, A, b, mrows, r, xc, status, st);
    const int gs = group_size_for(m_max);
    if (gs < 0 || d < 1 || d > MAX_D) return 2;
    // large shapes: one LP per wavefront with a wave-uniform pivot column (plp_wide.hip); PLP_CHEBY_WIDE=0 keeps the
    // lane-group kernels, PLP_CHEBY_WIDE=1 sends every shape it supports (d >= 5) there: A/B, tests
    const char* wide = getenv("PLP_CHEBY_WIDE");
    const bool wide_on = wide ? wide[0] == '1' : (d >= PLP_WIDE_MIN_D && (m_max > PLP_WIDE_MIN_M || B <= PLP_WIDE_SMALL_B));
    if (wide_on && !(wide && wide[0] == '0') && launch_cheby_w(B, m_max, d, A, b, mrows, r, xc, status, st) == 0) return 0;
    // d <= 8: four rows per lane (PLP_CHEBY_1ROW=1 keeps the one-row-per-lane kernel: A/B, tests)
    const char* one = getenv("PLP_CHEBY_1ROW");
    if (!(one && one[0] == '1') && launch_cheby_r(B, m_max, d, A, b, mrows, r, xc, status, st) == 0) return 0;
    switch (d) {
        PLP_CASE_D(1) PLP_CASE_D(2) PLP_CASE_D(3) PLP_CASE_D(4) PLP_CASE_D(5) PLP_CASE_D(6)
        PLP_CASE_D(7) PLP_CASE_D(8) PLP_CASE_D(9) PLP_CASE_D(10) PLP_CASE_D(11) PLP_CASE_D(12)
        PLP_CASE_D(13) PLP_CASE_D(14) PLP_CASE_D(15) PLP_CASE_D(16)
        default: return 2;
    }
    return 0;
}

}  // namespace plp
