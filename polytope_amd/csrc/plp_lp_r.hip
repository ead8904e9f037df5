// plp_lp_r.hip -- launcher of the generic LP batches on R rows per lane (kernels: plp_cheby_r_impl.hpp).
#include "plp_cheby_r_impl.hpp"

namespace plp {

template <int N, int GS>
static int launch_lp_r_ng(long long B, int m_max, const double* c, const double* G, const double* h, const int* mrows,
                          double* x, double* fun, int* status, int* iters, hipStream_t st) {
    constexpr long long gpb = RBLK / GS;
    const long long blocks = (B + gpb - 1) / gpb;
    if (blocks > 2147483647ll) return 1;
    hipLaunchKernelGGL((lp_r_kernel<N, GS>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(RBLK), 0, st, B, m_max, c,
                       G, h, mrows, x, fun, status, iters);
    if constexpr (P1_FAST<N>::value) {
        constexpr int RP = P1Rows<N>::value;
        constexpr int GSP = GS * RowsPerLane<N>::value / RP;  // same row slots, RP rows per lane
        constexpr long long gpbp = RBLK / GSP;
        const long long blocksp = (B + gpbp - 1) / gpbp;
        hipLaunchKernelGGL((lp_p1_r_kernel<N, GSP, RP>), dim3((unsigned)(blocksp < 1 ? 1 : blocksp)), dim3(RBLK), 0, st,
                           B, m_max, c, G, h, mrows, x, fun, status, iters);
    }
    return 0;
}

template <int N>
static int launch_lp_r_n(long long B, int m_max, const double* c, const double* G, const double* h, const int* mrows,
                         double* x, double* fun, int* status, int* iters, hipStream_t st) {
    PLP_DISPATCH_GS(RowsPerLane<N>::value, m_max,
                    (launch_lp_r_ng<N, GSV>(B, m_max, c, G, h, mrows, x, fun, status, iters, st)));
}

#define PLP_CASE_LPR(K) case K: return launch_lp_r_n<K>(B, m_max, c, G, h, mrows, x, fun, status, iters, st);

int launch_lp_r(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                double* x, double* fun, int* status, int* iters, hipStream_t st) {
    if (m_max < 1 || m_max > MAX_M || B < 1) return 1;
    switch (n) {
        PLP_CASE_LPR(1) PLP_CASE_LPR(2) PLP_CASE_LPR(3) PLP_CASE_LPR(4)
        PLP_CASE_LPR(5) PLP_CASE_LPR(6) PLP_CASE_LPR(7) PLP_CASE_LPR(8)
        PLP_CASE_LPR(9) PLP_CASE_LPR(10) PLP_CASE_LPR(11) PLP_CASE_LPR(12) PLP_CASE_LPR(13)  // two rows per lane
        PLP_CASE_LPR(14) PLP_CASE_LPR(15) PLP_CASE_LPR(16) PLP_CASE_LPR(17)
        default: return 1;
    }
}

}  // namespace plp
