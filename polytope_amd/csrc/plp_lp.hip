// plp_lp.hip -- batched stand-alone LP kernels (gfx950).
//
//   lp_kernel<N>    : B independent LPs  min c'x s.t. Gx<=h, x free   (solvers.py:76-106,149-158)
//   cheby_kernel<D> : B Chebyshev-ball LPs (form F1, polytope.py:1283-1288):
//                     c = -e_{d+1}, G = [A | sqrt(sum(A*A,1))], h = b
//
// One LP per lane group (GS lanes, GS >= rows), 64/GS LPs per wavefront, 256-thread
// workgroups, grid-stride over the batch.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_simplex.hpp"
#include "plp_simplex_r.hpp"

namespace plp {

template <int N>
__global__ __launch_bounds__(BLOCK) void lp_kernel(long long B, int m_max, int gs,
                                                   const double* __restrict__ c,
                                                   const double* __restrict__ G,
                                                   const double* __restrict__ h,
                                                   const int* __restrict__ mrows,
                                                   double* __restrict__ x, double* __restrict__ fun,
                                                   int* __restrict__ status, int* __restrict__ iters) {
    constexpr int NC = N + 1;  // + phase-1 artificial
    const Grp g(gs);
    const int gpb = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    for (long long base = (long long)blockIdx.x * gpb; base < B; base += (long long)gridDim.x * gpb) {
        const long long lp = base + gib;
        const bool valid = lp < B;
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

// Chebyshev batch with four rows per lane (SimplexR): a polytope of up to 16 / 32 / 64 rows takes a
// group of 4 / 8 / 16 lanes, so a wavefront carries 16 / 8 / 4 LPs (see plp_simplex_r.hpp).  Lane l
// loads its 4 consecutive rows straight from HBM (4*D contiguous doubles).
template <int D>
__global__ __launch_bounds__(BLOCK, (D <= 4 ? 3 : 1)) void cheby_r_kernel(long long B, int m_max, int gs,
                                                                          const double* __restrict__ A,
                                                                          const double* __restrict__ b,
                                                                          const int* __restrict__ mrows,
                                                                          double* __restrict__ r,
                                                                          double* __restrict__ xc,
                                                                          int* __restrict__ status) {
    constexpr int R = 4;
    const Grp g(gs);
    const int gpb = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    const int row0 = g.gl * R;
    const long long p = (long long)blockIdx.x * gpb + gib;
    const bool valid = p < B;
    const int m = valid ? (mrows ? mrows[p] : m_max) : 0;
    SimplexR<D + 1, R, true> S;
    S.reset(D + 1, m, row0);
    unsigned actb = 0u;
    bool inf0 = false, finite = true;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const bool h = valid & (row0 + k < m) & (m <= gs * R);
        double nrm2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            const double v = h ? A[(p * m_max + row0 + k) * D + kk] : 0.0;
            S.T[k][kk] = v;
            nrm2 = nrm2 + v * v;
            finite = finite & isfinite(v);
        }
        const double bk = h ? b[p * m_max + row0 + k] : 0.0;
        finite = finite & isfinite(bk);
        const double nrm = sqrt(nrm2);
        const bool zero = !(nrm > 0.0);
        const bool on = h & !zero;
        S.T[k][D] = on ? nrm : 0.0;
        S.beta[k] = on ? bk : 0.0;
        S.init_q[k] = bk / nrm;
        actb |= on ? (1u << k) : 0u;
        inf0 = inf0 | (h & zero & (bk < -TOL_FEAS));
    }
    S.ract = actb;
    S.init_elig = actb;
    const bool infeasible0 = grp_ballot(inf0, g) != 0;
    const bool bad = (grp_ballot(!finite, g) != 0) | (m > gs * R);
    S.cost[D] = -1.0;
    S.mode = M_INIT;
    S.init_col = D;
    S.mode_after_init = M_P2;
    if (!valid | bad) { S.mode = M_DONE; S.status = ST_NUM; }
    else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
    S.run(g);
    const bool ok = S.status == ST_OPT;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
    for (int j = 0; j <= D; ++j) {
        bool found;
        const double mine = S.x_of(j, found);
        const uint64_t ob = grp_ballot(found, g);
        const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
        const double xj = ok ? (ob ? v : 0.0) : qnan;
        if (valid & (g.gl == 0)) {
            if (j < D) xc[p * D + j] = xj; else r[p] = xj;
        }
    }
    if (valid & (g.gl == 0)) status[p] = S.status;
}

template <int D>
static void launch_cheby_r_d(long long B, int m_max, const double* A, const double* b, const int* mrows,
                             double* r, double* xc, int* status, hipStream_t st) {
    const int gs = m_max <= 16 ? 4 : (m_max <= 32 ? 8 : 16);
    const long long gpb = BLOCK / gs;
    const long long blocks = (B + gpb - 1) / gpb;
    hipLaunchKernelGGL(cheby_r_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), 0, st, B, m_max, gs, A, b, mrows, r,
                       xc, status);
}

// Adjacency of all pairs of n single-polytope cells (polytope.py:1843-1866 `is_adjacent(overlap=True)`
// under prop2partition.py:57-61 `find_adjacent_regions`): for the pair (i, j < i) the rows of both
// cells are stacked with b + abs_tol, and the pair is adjacent iff the Chebyshev LP of the stack is
// optimal with r > abs_tol/10 (`is_fulldim(dummy, abs_tol / 10)`).  One lane group per pair builds
// the stacked LP straight from the resident cells (n cells stay in L2), so nothing is staged by
// the host.  adj is n x n, symmetric, ones on the diagonal.
template <int D>
__global__ __launch_bounds__(BLOCK, (D <= 4 ? 3 : 1)) void adjacent_r_kernel(
    int n, int m_max, int gs, const double* __restrict__ A, const double* __restrict__ b,
    const int* __restrict__ mrows, double abs_tol, unsigned char* __restrict__ adj) {
    constexpr int R = 4;
    const Grp g(gs);
    const int gpb = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    const int row0 = g.gl * R;
    const long long npairs = (long long)n * (n - 1) / 2;
    const long long p = (long long)blockIdx.x * gpb + gib;
    const bool valid = p < npairs;
    // p -> (i, j) with j < i, p = i (i - 1) / 2 + j
    long long i = valid ? (long long)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5) : 1;
    while (i * (i - 1) / 2 > p) --i;
    while ((i + 1) * i / 2 <= p) ++i;
    const long long j = valid ? p - i * (i - 1) / 2 : 0;
    const int mi = valid ? (mrows ? mrows[i] : m_max) : 0;
    const int mj = valid ? (mrows ? mrows[j] : m_max) : 0;
    const int m = mi + mj;
    SimplexR<D + 1, R, true> S;
    S.reset(D + 1, m, row0);
    unsigned actb = 0u;
    bool inf0 = false, finite = true;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int rr = row0 + k;
        const bool h = valid & (rr < m) & (m <= gs * R);
        const long long cell = (rr < mi) ? i : j;
        const int row = (rr < mi) ? rr : rr - mi;
        double nrm2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            const double v = h ? A[(cell * m_max + row) * D + kk] : 0.0;
            S.T[k][kk] = v;
            nrm2 = nrm2 + v * v;
            finite = finite & isfinite(v);
        }
        const double bk = h ? b[cell * m_max + row] + abs_tol : 0.0;  // b1 += abs_tol; b2 += abs_tol
        finite = finite & isfinite(bk);
        const double nrm = sqrt(nrm2);
        const bool zero = !(nrm > 0.0);
        const bool on = h & !zero;
        S.T[k][D] = on ? nrm : 0.0;
        S.beta[k] = on ? bk : 0.0;
        S.init_q[k] = bk / nrm;
        actb |= on ? (1u << k) : 0u;
        inf0 = inf0 | (h & zero & (bk < -TOL_FEAS));
    }
    S.ract = actb;
    S.init_elig = actb;
    const bool infeasible0 = grp_ballot(inf0, g) != 0;
    const bool bad = (grp_ballot(!finite, g) != 0) | (m > gs * R);
    S.cost[D] = -1.0;
    S.mode = M_INIT;
    S.init_col = D;
    S.mode_after_init = M_P2;
    if (!valid | bad) { S.mode = M_DONE; S.status = ST_NUM; }
    else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
    S.run(g);
    bool found;
    const double mine = S.x_of(D, found);
    const uint64_t ob = grp_ballot(found, g);
    const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
    const double rad = ob ? v : 0.0;
    const bool yes = (S.status == ST_OPT) & (rad > abs_tol / 10);
    if (valid & (g.gl == 0)) {
        adj[i * n + j] = yes ? 1 : 0;
        adj[j * n + i] = yes ? 1 : 0;
    }
    // diagonal
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (t < n) adj[t * n + t] = 1;
}

template <int D>
static int launch_adjacent_d(int n, int m_max, const double* A, const double* b, const int* mrows, double abs_tol,
                             unsigned char* adj, hipStream_t st) {
    const int rows = 2 * m_max;
    const int gs = rows <= 16 ? 4 : (rows <= 32 ? 8 : 16);
    const long long gpb = BLOCK / gs;
    const long long npairs = (long long)n * (n - 1) / 2;
    long long blocks = (npairs + gpb - 1) / gpb;
    const long long bdiag = ((long long)n + BLOCK - 1) / BLOCK;
    if (blocks < bdiag) blocks = bdiag;
    if (blocks < 1) blocks = 1;
    if (blocks > 2147483647ll) return 2;
    hipLaunchKernelGGL(adjacent_r_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), 0, st, n, m_max, gs, A, b, mrows,
                       abs_tol, adj);
    return 0;
}

int launch_adjacent(int n, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                    unsigned char* adj, hipStream_t st) {
    if (n < 0 || m_max < 1 || 2 * m_max > MAX_M || d < 1 || d > 8) return 2;
    if (n == 0) return 0;
    switch (d) {
        case 1: return launch_adjacent_d<1>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 2: return launch_adjacent_d<2>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 3: return launch_adjacent_d<3>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 4: return launch_adjacent_d<4>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 5: return launch_adjacent_d<5>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 6: return launch_adjacent_d<6>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 7: return launch_adjacent_d<7>(n, m_max, A, b, mrows, abs_tol, adj, st);
        case 8: return launch_adjacent_d<8>(n, m_max, A, b, mrows, abs_tol, adj, st);
        default: return 2;
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
                        const int* mrows, double* x, double* fun, int* status, int* iters, hipStream_t st) {
    hipLaunchKernelGGL(lp_kernel<N>, dim3(pick_grid(B, gs)), dim3(BLOCK), 0, st, B, m_max, gs, c, G, h, mrows, x,
                       fun, status, iters);
}

template <int D>
static void launch_cheby_d(long long B, int m_max, int gs, const double* A, const double* b, const int* mrows,
                           double* r, double* xc, int* status, hipStream_t st) {
    hipLaunchKernelGGL(cheby_kernel<D>, dim3(pick_grid(B, gs)), dim3(BLOCK), 0, st, B, m_max, gs, A, b, mrows, r,
                       xc, status);
}

#define PLP_CASE_N(K) case K: launch_lp_n<K>(B, m_max, gs, c, G, h, mrows, x, fun, status, iters, st); break;
#define PLP_CASE_D(K) case K: launch_cheby_d<K>(B, m_max, gs, A, b, mrows, r, xc, status, st); break;

int launch_lp(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
              double* x, double* fun, int* status, int* iters, hipStream_t st) {
    const int gs = group_size_for(m_max);
    if (gs < 0 || n < 1 || n > MAX_D + 1) return 2;
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
    const int gs = group_size_for(m_max);
    if (gs < 0 || d < 1 || d > MAX_D) return 2;
    // d <= 8: four rows per lane (PLP_CHEBY_1ROW=1 keeps the one-row-per-lane kernel: A/B, tests)
    const char* one = getenv("PLP_CHEBY_1ROW");
    if (d <= 8 && m_max >= 1 && !(one && one[0] == '1') && (B + 63) / 4 < 2147483647ll) {
        switch (d) {
            case 1: launch_cheby_r_d<1>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 2: launch_cheby_r_d<2>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 3: launch_cheby_r_d<3>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 4: launch_cheby_r_d<4>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 5: launch_cheby_r_d<5>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 6: launch_cheby_r_d<6>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 7: launch_cheby_r_d<7>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
            case 8: launch_cheby_r_d<8>(B, m_max, A, b, mrows, r, xc, status, st); return 0;
        }
    }
    switch (d) {
        PLP_CASE_D(1) PLP_CASE_D(2) PLP_CASE_D(3) PLP_CASE_D(4) PLP_CASE_D(5) PLP_CASE_D(6)
        PLP_CASE_D(7) PLP_CASE_D(8) PLP_CASE_D(9) PLP_CASE_D(10) PLP_CASE_D(11) PLP_CASE_D(12)
        PLP_CASE_D(13) PLP_CASE_D(14) PLP_CASE_D(15) PLP_CASE_D(16)
        default: return 2;
    }
    return 0;
}

}  // namespace plp
