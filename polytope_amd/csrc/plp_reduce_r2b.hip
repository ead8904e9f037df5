// plp_reduce_r2b.hip -- fused reduce() for d = 13..16: two rows per lane (four rows of up to 17 columns do not fit the
// VGPR file), groups of 16 / 32 lanes for up to 32 / 64 rows.  Same kernel template as d <= 8.
#include "plp_reduce_r_impl.hpp"

#ifndef PLP_REDUCE_WG_MAXB
#define PLP_REDUCE_WG_MAXB 1500
#endif

namespace plp {

template <int D>
static int launch_r2_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double abs_tol,
                       unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    // PLP_REDUCE_R1=1 (A/B): one row per lane, one polytope of up to 64 rows per wavefront -- twice the wavefronts
    const char* r1 = getenv("PLP_REDUCE_R1");
    if (r1 && r1[0] == '1' && m_max > 32)
        return launch_reduce_r_dg<D, 64, 1>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    // more than 32 rows: one polytope per wavefront, F3 / F2 without a stored dictionary (plp_lazy.hpp) -- measured 1.2x
    // (48 rows, d = 9) to 2.1x (36 rows, d = 14) faster than two rows per lane, outputs bitwise equal; with 32 rows and
    // fewer the four-polytopes-per-wavefront form below wins or ties.  PLP_REDUCE_LAZY=0 / 1: never / always (A/B).
    const char* lz = getenv("PLP_REDUCE_LAZY");
    // (round 4: and any row count up to PLP_REDUCE_WG_MAXB polytopes, see plp_reduce_r.hip)
    const bool small_batch = !lz && B <= PLP_REDUCE_WG_MAXB && !(r1 && r1[0]);
    if ((lz && lz[0] == '1') || (m_max > 32 && !(lz && lz[0] == '0')) || small_batch)
        return launch_reduce_lazy<D>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    if (m_max <= 32) return launch_reduce_r_dg<D, 16, 2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    return launch_reduce_r_dg<D, 32, 2>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
}

#define PLP_CASE_R2(K) case K: return launch_r2_d<K>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);

int launch_reduce_r2b(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                      double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                      hipStream_t st) {
    switch (d) {
        PLP_CASE_R2(13) PLP_CASE_R2(14) PLP_CASE_R2(15) PLP_CASE_R2(16)
        default: return 1;
    }
}

}  // namespace plp
