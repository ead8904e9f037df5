// plp_reduce_r2c.hip -- fused reduce() for d = 5..8 on TWO rows per lane (groups of 8 / 16 / 32 lanes for up to 16 / 32 /
// 64 rows).  Four rows of 6..9 columns per lane cost 190..244 VGPRs (two wavefronts per SIMD, plp_reduce_r.hip); two rows
// fit three wavefronts, and the kernel at these shapes is bound by the latency of the pivot's dependency chain, not by
// instruction issue.  Measured (MI355X, ms per batch, four rows -> two rows per lane): (32,6) B = 20 000 0.561 -> 0.497,
// (32,8) 0.674 -> 0.566, (24,5) 0.355 -> 0.310, (64,8) B = 5 000 0.721 -> 0.616; a tie once the batch fills the chip either
// way ((32,6) B = 100 000: 1.99 -> 1.95, (64,8) B = 20 000: 1.735 -> 1.725).  Outputs bitwise equal in keep / flags / nlp.
// Same kernel template as everything else (plp_reduce_r_impl.hpp); its own translation unit for a parallel build.
#include "plp_reduce_r_impl.hpp"

namespace plp {

template <int D>
static int launch_r2c_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double abs_tol,
                        unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    if (m_max <= 16) return launch_reduce_r_dg<D, 8, 2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    if (m_max <= 32) return launch_reduce_r_dg<D, 16, 2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    return launch_reduce_r_dg<D, 32, 2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
}

#define PLP_CASE_R2C(K) case K: return launch_r2c_d<K>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);

int launch_reduce_r2c(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                      double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                      hipStream_t st) {
    switch (d) {
        PLP_CASE_R2C(5) PLP_CASE_R2C(6) PLP_CASE_R2C(7) PLP_CASE_R2C(8)
        default: return 1;
    }
}

}  // namespace plp
