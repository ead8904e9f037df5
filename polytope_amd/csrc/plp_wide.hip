// plp_wide.hip -- one LP per WAVEFRONT, one dictionary row per lane, for the large shapes (d = 9..16, m <= 64).
//
//   cheby_w_kernel<D> : Chebyshev-ball LPs (form F1, polytope/polytope.py:1283-1288) of a batch of polytopes
//
// The lane-group engines (plp_simplex.hpp, plp_simplex_r.hpp) let several LPs share a wavefront, so the entering
// column differs from lane to lane and every access T[e] is a chain of selects over the columns: at 17 columns that
// chain (and its twin for the column fix-up) is most of the ~500 VALU instructions a (64,17) pivot costs there.  Here a
// wavefront owns ONE LP: the entering column e and the pivot row r are wave-uniform (SGPRs), so
//   * T[e] is one indexed register move (the row is a register vector), not 17 selects;
//   * the reduced costs live ONCE, in LDS, lane j looking after column j: pricing is one load, one compare and a
//     DPP wave reduction instead of a 17-step scan of replicated values, and the cost row's update is one FMA;
//   * the pivot row is scaled in place by its own lane (exec = that lane), stored to LDS, and read back by all lanes
//     as broadcast ds_reads -- no ds_bpermute pairs, no VALU slots spent on moving it;
//   * the ratio test's exact f64 minimum is a DPP reduction over the 64 lanes (row_bcast15/31 for the upper levels).
// About 110 VALU instructions per pivot.  Pivot rules as everywhere (oracle/plp_oracle.c restates them): free
// variables enter in either direction and never leave, Dantzig pricing (largest |c_j|, lowest column on ties), ratio
// test ties to the lowest row, Bland's rule after BLAND_AFTER consecutive degenerate pivots, forced first pivot
// "r enters, row argmin b_i/||a_i|| leaves" for F1.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_wave.hpp"
#include "plp_wide.hpp"

namespace plp {

using namespace wide;

template <int D>
__global__ __launch_bounds__(64) void cheby_w_kernel(long long B, int m_max, const double* __restrict__ A,
                                                     const double* __restrict__ b, const int* __restrict__ mrows,
                                                     double* __restrict__ r_out, double* __restrict__ xc,
                                                     int* __restrict__ status) {
    constexpr int NC = D + 1;
    __shared__ WideShared<NC> sh;
    const int lane = threadIdx.x;
    const long long p = blockIdx.x;
    if (p >= B) return;
    const int m = mrows ? mrows[p] : m_max;
    const bool has = lane < m;
    typename RowVec<NC>::type Tv = (typename RowVec<NC>::type)(0.0);
    double T16 = 0.0;
    double nrm2 = 0.0;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = has ? A[(p * m_max + lane) * D + k] : 0.0;
        ROW_SET(k, v);
        nrm2 = nrm2 + v * v;
        finite = finite & isfinite(v);
    }
    const double bi = has ? b[p * m_max + lane] : 0.0;
    finite = finite & isfinite(bi);
    const double nrm = sqrt(nrm2);
    const bool zero = !(nrm > 0.0);
    bool rowact = has & !zero;
    ROW_SET(D, rowact ? nrm : 0.0);
    double beta = rowact ? bi : 0.0;
    int rowvar = NC + lane, rowneg = 0;
    if (lane <= NC) {
        sh.cost[lane] = lane == D ? -1.0 : 0.0;
        sh.cv[lane] = (lane + 1) << 1;
    }
    const bool infeasible0 = __ballot(has & zero & (bi < -TOL_FEAS)) != 0;
    const bool bad = (__ballot(!finite) != 0) | (m > 64);
    __syncthreads();
    int st, iters = 0;
    if (bad) st = ST_NUM;
    else if (infeasible0) st = ST_INFEAS;
    else st = wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bi / nrm, iters);
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double mine = rowneg ? -beta : beta;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const uint64_t ob = __ballot(rowvar == j);
        const double v = ob ? uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
        const double xj = (st == ST_OPT) ? v : qnan;
        if (lane == 0) {
            if (j < D) xc[p * D + j] = xj; else r_out[p] = xj;
        }
    }
    if (lane == 0) status[p] = st;
}

// The same LP on a ROW SUBSET of one resident table (region_diff's search, plp_rdiff.hip): LP q of the class is list
// sel[q] of the batch, its rows rows[off[p] .. off[p + 1]) (at most 64).  out[p] as cheby_gather_r_kernel writes it: the
// radius if the LP is optimal with r >= 0, 0 if optimal with r < 0, NaN for any other status.
template <int D>
__global__ __launch_bounds__(64) void cheby_gather_w_kernel(long long nlp, const int* __restrict__ off,
                                                            const int* __restrict__ rows, const int* __restrict__ sel,
                                                            const double* __restrict__ A, const double* __restrict__ b,
                                                            double* __restrict__ out) {
    constexpr int NC = D + 1;
    __shared__ WideShared<NC> sh;
    const int lane = threadIdx.x;
    const long long q = blockIdx.x;
    if (q >= nlp) return;
    const int p = sel[q];
    const int o = off[p];
    const int m = off[p + 1] - o;
    const bool has = lane < m;
    const long long row = has ? rows[o + lane] : 0;
    typename RowVec<NC>::type Tv = (typename RowVec<NC>::type)(0.0);
    double T16 = 0.0;
    double nrm2 = 0.0;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = has ? A[row * D + k] : 0.0;
        ROW_SET(k, v);
        nrm2 = nrm2 + v * v;
        finite = finite & isfinite(v);
    }
    const double bi = has ? b[row] : 0.0;
    finite = finite & isfinite(bi);
    const double nrm = sqrt(nrm2);
    const bool zero = !(nrm > 0.0);
    bool rowact = has & !zero;
    ROW_SET(D, rowact ? nrm : 0.0);
    double beta = rowact ? bi : 0.0;
    int rowvar = NC + lane, rowneg = 0;
    if (lane <= NC) {
        sh.cost[lane] = lane == D ? -1.0 : 0.0;
        sh.cv[lane] = (lane + 1) << 1;
    }
    const bool infeasible0 = __ballot(has & zero & (bi < -TOL_FEAS)) != 0;
    const bool bad = (__ballot(!finite) != 0) | (m > 64);
    __syncthreads();
    int st, iters = 0;
    if (bad) st = ST_NUM;
    else if (infeasible0) st = ST_INFEAS;
    else st = wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bi / nrm, iters);
    const double mine = rowneg ? -beta : beta;
    const uint64_t ob = __ballot(rowvar == D);
    const double r = ob ? uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
    if (lane == 0) out[p] = st != ST_OPT ? __builtin_nan("") : (r >= 0.0 ? r : 0.0);
}

template <int D>
static int launch_cheby_gather_w_d(long long nlp, const int* off, const int* rows, const int* sel, const double* A,
                                   const double* b, double* out, hipStream_t st) {
    if (nlp > 2147483647ll) return 2;
    if (nlp < 1) return 0;
    hipLaunchKernelGGL((cheby_gather_w_kernel<D>), dim3((unsigned)nlp), dim3(64), 0, st, nlp, off, rows, sel, A, b, out);
    return 0;
}

#define PLP_CASE_GW(K) case K: return launch_cheby_gather_w_d<K>(nlp, off, rows, sel, A, b, out, st);

// lists of at most 64 rows, d = 5..16; returns 1 when it does not apply
int launch_cheby_gather_w(int d, long long nlp, const int* off, const int* rows, const int* sel, const double* A,
                          const double* b, double* out, hipStream_t st) {
    switch (d) {
        PLP_CASE_GW(2) PLP_CASE_GW(3) PLP_CASE_GW(4)
        PLP_CASE_GW(5) PLP_CASE_GW(6) PLP_CASE_GW(7) PLP_CASE_GW(8) PLP_CASE_GW(9) PLP_CASE_GW(10)
        PLP_CASE_GW(11) PLP_CASE_GW(12) PLP_CASE_GW(13) PLP_CASE_GW(14) PLP_CASE_GW(15) PLP_CASE_GW(16)
        default: return 1;
    }
}

// Pair LPs of find_adjacent_regions / is_adjacent(overlap = True) (polytope/polytope.py:1843-1866 under the pair loop of
// prop2partition.py:57-61), one pair per wavefront: the rows of cell i and of cell j stacked (at most 64), every b
// inflated, adjacent iff the Chebyshev radius exceeds `thresh` -- the contract of adjacent_r_kernel (plp_cheby_r_impl.hpp),
// for d = 5..16 (the lane-group kernel stops at d = 8 and holds one or two such LPs per wavefront from 33 rows on).
template <int D>
__global__ __launch_bounds__(64) void adjacent_w_kernel(int n, int m_max, const double* __restrict__ A,
                                                        const double* __restrict__ b, const int* __restrict__ mrows,
                                                        double inflate, double thresh, unsigned char* __restrict__ adj,
                                                        long long p_lo, long long p_hi, unsigned char* __restrict__ compact,
                                                        int cross_n1) {
    constexpr int NC = D + 1;
    __shared__ WideShared<NC> sh;
    const int lane = threadIdx.x;
    if (!compact) {  // diagonal
        const long long t = (long long)blockIdx.x * 64 + lane;
        if (t < n) adj[t * n + t] = 1;
    }
    const long long p = p_lo + (long long)blockIdx.x;
    if (p >= p_hi) return;
    // p -> (i, j) with j < i, p = i (i - 1) / 2 + j
    long long i = (long long)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
    while (i * (i - 1) / 2 > p) --i;
    while ((i + 1) * i / 2 <= p) ++i;
    long long j = p - i * (i - 1) / 2;
    if (cross_n1 > 0) {   // two lists in one table (see adjacent_r_kernel): cell i of the first with cell j of the second
        const long long n2 = n - cross_n1;
        i = p / n2;
        j = cross_n1 + (p - i * n2);
    }
    const int mi = mrows ? mrows[i] : m_max;
    const int mj = mrows ? mrows[j] : m_max;
    const int m = mi + mj;
    const bool has = lane < m;
    const long long row = has ? ((lane < mi) ? i * m_max + lane : j * m_max + (lane - mi)) : 0;
    typename RowVec<NC>::type Tv = (typename RowVec<NC>::type)(0.0);
    double T16 = 0.0;
    double nrm2 = 0.0;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = has ? A[row * D + k] : 0.0;
        ROW_SET(k, v);
        nrm2 = nrm2 + v * v;
        finite = finite & isfinite(v);
    }
    const double bi = has ? b[row] + inflate : 0.0;  // b1 += abs_tol; b2 += abs_tol
    finite = finite & isfinite(bi);
    const double nrm = sqrt(nrm2);
    const bool zero = !(nrm > 0.0);
    bool rowact = has & !zero;
    ROW_SET(D, rowact ? nrm : 0.0);
    double beta = rowact ? bi : 0.0;
    int rowvar = NC + lane, rowneg = 0;
    if (lane <= NC) {
        sh.cost[lane] = lane == D ? -1.0 : 0.0;
        sh.cv[lane] = (lane + 1) << 1;
    }
    const bool infeasible0 = __ballot(has & zero & (bi < -TOL_FEAS)) != 0;
    const bool bad = (__ballot(!finite) != 0) | (m > 64);
    __syncthreads();
    int st, iters = 0;
    if (bad) st = ST_NUM;
    else if (infeasible0) st = ST_INFEAS;
    else st = wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bi / nrm, iters);
    const double mine = rowneg ? -beta : beta;
    const uint64_t ob = __ballot(rowvar == D);
    const double r = ob ? uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
    const bool yes = (st == ST_OPT) & (r > thresh);
    if (lane == 0) {
        if (compact) {
            compact[p - p_lo] = yes ? 1 : 0;
        } else {
            adj[i * n + j] = yes ? 1 : 0;
            adj[j * n + i] = yes ? 1 : 0;
        }
    }
}

template <int D>
static int launch_adjacent_w_d(int n, int m_max, const double* A, const double* b, const int* mrows, double inflate,
                               double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                               int cross_n1, hipStream_t st) {
    long long blocks = p_hi - p_lo;
    const long long bdiag = compact ? 0 : ((long long)n + 63) / 64;
    if (blocks < bdiag) blocks = bdiag;
    if (blocks < 1) blocks = 1;
    if (blocks > 2147483647ll) return 2;
    hipLaunchKernelGGL((adjacent_w_kernel<D>), dim3((unsigned)blocks), dim3(64), 0, st, n, m_max, A, b, mrows, inflate, thresh,
                       adj, p_lo, p_hi, compact, cross_n1);
    return 0;
}

#define PLP_CASE_AW(K) \
    case K: return launch_adjacent_w_d<K>(n, m_max, A, b, mrows, inflate, thresh, adj, p_lo, p_hi, compact, cross_n1, st);

// pairs [p_lo, p_hi) of n cells, 2 * m_max <= 64, d = 5..16; returns 1 when it does not apply
int launch_adjacent_w(int n, int m_max, int d, const double* A, const double* b, const int* mrows, double inflate,
                      double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                      hipStream_t st, int cross_n1) {
    switch (d) {
        PLP_CASE_AW(5) PLP_CASE_AW(6) PLP_CASE_AW(7) PLP_CASE_AW(8) PLP_CASE_AW(9) PLP_CASE_AW(10)
        PLP_CASE_AW(11) PLP_CASE_AW(12) PLP_CASE_AW(13) PLP_CASE_AW(14) PLP_CASE_AW(15) PLP_CASE_AW(16)
        default: return 1;
    }
}

template <int D>
static int launch_cheby_w_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double* r,
                            double* xc, int* status, hipStream_t st) {
    if (B > 2147483647ll) return 2;
    hipLaunchKernelGGL((cheby_w_kernel<D>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(64), 0, st, B, m_max, A, b, mrows, r, xc,
                       status);
    return 0;
}

// one LP per wavefront (m_max <= 64); returns 1 when it does not apply
int launch_cheby_w(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                   double* xc, int* status, hipStream_t st) {
    if (m_max < 1 || m_max > 64) return 1;
    switch (d) {
        case 5: return launch_cheby_w_d<5>(B, m_max, A, b, mrows, r, xc, status, st);
        case 6: return launch_cheby_w_d<6>(B, m_max, A, b, mrows, r, xc, status, st);
        case 7: return launch_cheby_w_d<7>(B, m_max, A, b, mrows, r, xc, status, st);
        case 8: return launch_cheby_w_d<8>(B, m_max, A, b, mrows, r, xc, status, st);
        case 9: return launch_cheby_w_d<9>(B, m_max, A, b, mrows, r, xc, status, st);
        case 10: return launch_cheby_w_d<10>(B, m_max, A, b, mrows, r, xc, status, st);
        case 11: return launch_cheby_w_d<11>(B, m_max, A, b, mrows, r, xc, status, st);
        case 12: return launch_cheby_w_d<12>(B, m_max, A, b, mrows, r, xc, status, st);
        case 13: return launch_cheby_w_d<13>(B, m_max, A, b, mrows, r, xc, status, st);
        case 14: return launch_cheby_w_d<14>(B, m_max, A, b, mrows, r, xc, status, st);
        case 15: return launch_cheby_w_d<15>(B, m_max, A, b, mrows, r, xc, status, st);
        case 16: return launch_cheby_w_d<16>(B, m_max, A, b, mrows, r, xc, status, st);
        default: return 1;
    }
}

}  // namespace plp
