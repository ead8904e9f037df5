// plp_cheby_r.hip -- launchers of the Chebyshev-ball and pair-adjacency batches on R rows per lane
// (kernels: plp_cheby_r_impl.hpp).
#include "plp_cheby_r_impl.hpp"

namespace plp {

template <int D, int GS>
static int launch_cheby_r_dg(long long B, int m_max, const double* A, const double* b, const int* mrows, double* r,
                             double* xc, int* status, hipStream_t st) {
    constexpr long long gpb = RBLK / GS;
    const long long blocks = (B + gpb - 1) / gpb;
    if (blocks > 2147483647ll) return 1;
    hipLaunchKernelGGL((cheby_r_kernel<D, GS>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(RBLK), 0, st, B, m_max,
                       A, b, mrows, r, xc, status, force_retry_env());
    return 0;
}

template <int D>
static int launch_cheby_r_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double* r,
                            double* xc, int* status, hipStream_t st) {
    PLP_DISPATCH_GS(RowsPerLane<D>::value, m_max, (launch_cheby_r_dg<D, GSV>(B, m_max, A, b, mrows, r, xc, status, st)));
}

// returns 0 when launched, 1 when this kernel does not apply
int launch_cheby_r(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                   double* xc, int* status, hipStream_t st) {
    if (m_max < 1 || m_max > MAX_M) return 1;
    switch (d) {
        case 1: return launch_cheby_r_d<1>(B, m_max, A, b, mrows, r, xc, status, st);
        case 2: return launch_cheby_r_d<2>(B, m_max, A, b, mrows, r, xc, status, st);
        case 3: return launch_cheby_r_d<3>(B, m_max, A, b, mrows, r, xc, status, st);
        case 4: return launch_cheby_r_d<4>(B, m_max, A, b, mrows, r, xc, status, st);
        case 5: return launch_cheby_r_d<5>(B, m_max, A, b, mrows, r, xc, status, st);
        case 6: return launch_cheby_r_d<6>(B, m_max, A, b, mrows, r, xc, status, st);
        case 7: return launch_cheby_r_d<7>(B, m_max, A, b, mrows, r, xc, status, st);
        case 8: return launch_cheby_r_d<8>(B, m_max, A, b, mrows, r, xc, status, st);
        case 9: return launch_cheby_r_d<9>(B, m_max, A, b, mrows, r, xc, status, st);
        case 10: return launch_cheby_r_d<10>(B, m_max, A, b, mrows, r, xc, status, st);
        case 11: return launch_cheby_r_d<11>(B, m_max, A, b, mrows, r, xc, status, st);
        case 12: return launch_cheby_r_d<12>(B, m_max, A, b, mrows, r, xc, status, st);
        case 13: return launch_cheby_r_d<13>(B, m_max, A, b, mrows, r, xc, status, st);
        case 14: return launch_cheby_r_d<14>(B, m_max, A, b, mrows, r, xc, status, st);
        case 15: return launch_cheby_r_d<15>(B, m_max, A, b, mrows, r, xc, status, st);
        case 16: return launch_cheby_r_d<16>(B, m_max, A, b, mrows, r, xc, status, st);
        default: return 1;
    }
}

template <int D, int GS>
static int launch_adjacent_dg(int n, int m_max, const double* A, const double* b, const int* mrows, double inflate,
                              double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                              int cross_n1, hipStream_t st) {
    constexpr long long gpb = RBLK / GS;
    long long blocks = (p_hi - p_lo + gpb - 1) / gpb;
    const long long bdiag = compact ? 0 : ((long long)n + RBLK - 1) / RBLK;
    if (blocks < bdiag) blocks = bdiag;
    if (blocks < 1) blocks = 1;
    if (blocks > 2147483647ll) return 2;
    hipLaunchKernelGGL((adjacent_r_kernel<D, GS>), dim3((unsigned)blocks), dim3(RBLK), 0, st, n, m_max, A, b, mrows,
                       inflate, thresh, adj, p_lo, p_hi, compact, force_retry_env(), cross_n1);
    return 0;
}

template <int D>
static int launch_adjacent_d(int n, int m_max, const double* A, const double* b, const int* mrows, double inflate,
                             double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                             int cross_n1, hipStream_t st) {
    const int rows = 2 * m_max;
    PLP_DISPATCH_GS(RowsPerLane<D>::value, rows,
                    (launch_adjacent_dg<D, GSV>(n, m_max, A, b, mrows, inflate, thresh, adj, p_lo, p_hi, compact, cross_n1, st)));
}

#define PLP_CASE_ADJ(K) \
    case K: return launch_adjacent_d<K>(n, m_max, A, b, mrows, inflate, thresh, adj, p_lo, p_hi, compact, cross_n1, st);

// compact == nullptr: all pairs into the n x n matrix adj; else pairs [p_lo, p_hi) into compact[p - p_lo]
int launch_adjacent(int n, int m_max, int d, const double* A, const double* b, const int* mrows, double inflate,
                    double thresh, unsigned char* adj, long long p_lo, long long p_hi, unsigned char* compact,
                    hipStream_t st, int cross_n1) {
    if (n < 0 || m_max < 1 || 2 * m_max > MAX_M || d < 1 || d > MAX_D) return 2;
    if (cross_n1 < 0 || cross_n1 > n || (cross_n1 > 0 && !compact)) return 2;
    const long long npairs = cross_n1 > 0 ? (long long)cross_n1 * (n - cross_n1) : (long long)n * (n - 1) / 2;
    if (!compact) { p_lo = 0; p_hi = npairs; }
    if (p_lo < 0 || p_hi > npairs || p_lo > p_hi) return 2;
    if (n == 0 || (compact && p_lo == p_hi)) return 0;
    // d >= 9 (no lane-group kernel), and d = 5..8 when the stacked pair has more than 32 rows or the pairs are few: one
    // pair per wavefront (plp_wide.hip), the rule of the Chebyshev batches (plp_lp.hip).  PLP_ADJ_WIDE=0 / 1: A/B, tests
    {
        const char* aw = getenv("PLP_ADJ_WIDE");
        const bool wide = d > 8 || (aw ? aw[0] == '1' : (d >= 5 && (2 * m_max > 32 || p_hi - p_lo <= 4096)));
        if (wide && !(aw && aw[0] == '0' && d <= 8) &&
            launch_adjacent_w(n, m_max, d, A, b, mrows, inflate, thresh, adj, p_lo, p_hi, compact, st, cross_n1) == 0)
            return 0;
    }
    switch (d) {
        PLP_CASE_ADJ(1) PLP_CASE_ADJ(2) PLP_CASE_ADJ(3) PLP_CASE_ADJ(4)
        PLP_CASE_ADJ(5) PLP_CASE_ADJ(6) PLP_CASE_ADJ(7) PLP_CASE_ADJ(8)
        default: return 2;
    }
}

}  // namespace plp
