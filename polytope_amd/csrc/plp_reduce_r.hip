// plp_reduce_r.hip -- dispatch of the fused reduce() on R rows per lane (kernel: plp_reduce_r_impl.hpp).
//   d <= 8 : four rows per lane, groups of 4 / 8 / 16 lanes (this file); d = 5..8 beyond the latency form's batch sizes:
//            two rows per lane (plp_reduce_r2c.hip); d = 5..8 with more than 32 rows: one polytope per wavefront
//            (reduce_wdense_kernel, plp_reduce_r_impl.hpp)
//   d >= 9 : two rows per lane, groups of 16 / 32 lanes (plp_reduce_r2a.hip d = 9..12, plp_reduce_r2b.hip d = 13..16;
//            separate translation units only to keep the build parallel)
#include "plp_reduce_r_impl.hpp"

#ifndef PLP_REDUCE_WG_MAXB
#define PLP_REDUCE_WG_MAXB 1500   // d >= 5, any row count: one polytope per workgroup up to this many polytopes
#endif
#ifndef PLP_REDUCE_LAZY_MID
#define PLP_REDUCE_LAZY_MID 1  // d = 5..8 with more than 32 rows: one polytope per wavefront (reduce_wdense_kernel) by default
#endif

#ifndef PLP_REDUCE_LANE_MINB
#define PLP_REDUCE_LANE_MINB 0   // (16,3)-class batches larger than this: one LP per lane (plp_reduce_lane.hip); its 4-polytope tiles are ahead of the lane-group latency form down to a single polytope
#endif

#ifndef PLP_REDUCE_LANE4_MINB
#define PLP_REDUCE_LANE4_MINB 40000   // d = 4, fewer than 14 rows: batches larger than this
#endif
#ifndef PLP_REDUCE_LANE4_MINB_ROWS14
#define PLP_REDUCE_LANE4_MINB_ROWS14 3000   // d = 4, 14..32 rows: batches larger than this
#endif

namespace plp {

int launch_reduce_lane(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                       unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st);
int launch_reduce_r2a(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                      double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                      hipStream_t st);
int launch_reduce_r2b(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                      double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                      hipStream_t st);
int launch_reduce_r2c(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                      double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                      hipStream_t st);

template <int D>
static int launch_reduce_r_d(long long B, int m_max, int gs, const double* A, const double* b, const int* mrows,
                             double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                             hipStream_t st) {
    if constexpr (D <= PLP_REDUCE_R8_MAXD) {
        const char* e8 = getenv("PLP_REDUCE_R8");
        if (gs == 4 && !(e8 && e8[0] == '0'))
            return launch_reduce_r_dg<D, 2, 8>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    }
    if constexpr (D >= 5) {
        // more than 32 rows: one polytope per wavefront as for d >= 9 -- with the F3 / F2 LPs on the dense one-LP-per-wavefront
        // engine (reduce_wdense_kernel; round 3: (64,8) B = 5 000 0.657 -> 0.453 ms, (48,6) B = 20 000 1.08 -> 0.80 ms, ahead
        // of the two-rows-per-lane kernel AND of the latency form at every batch size, scripts/debug/wdense_ab.py);
        // PLP_REDUCE_LAZY=0 / 1: never / always (A/B)
        const char* lz = getenv("PLP_REDUCE_LAZY");
        // ... and, round 4, ANY row count while the batch is small: one polytope per workgroup, its LPs on 2 / 4 wavefronts
        // (reduce_wsplit_kernel), is 1.15x .. 2x ahead of the lane-group kernels up to ~1 000 polytopes at every (m <= 32, d)
        // measured, level at 2 000 .. 4 000, behind beyond (scripts/debug/mid_wsplit_table.py); the A/B switches of the
        // lane-group forms keep their meaning
        const bool small_batch = !lz && B <= PLP_REDUCE_WG_MAXB && !getenv("PLP_REDUCE_SPLIT") && !getenv("PLP_REDUCE_MIDR2") &&
                                 !getenv("PLP_REDUCE_HALF") && !getenv("PLP_REDUCE_MIX");
        if ((lz && lz[0] == '1') || (PLP_REDUCE_LAZY_MID && m_max > 32 && !(lz && lz[0] == '0')) || small_batch)
            return launch_reduce_lazy<D>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    }
    if constexpr (D >= 5) {
        // d = 5..8 beyond the latency form's batch sizes: two rows per lane (plp_reduce_r2c.hip: three wavefronts per SIMD
        // instead of two); PLP_REDUCE_MIDR2=0 / 1: never / always (A/B)
        const char* r2 = getenv("PLP_REDUCE_MIDR2");
        const char* sp = getenv("PLP_REDUCE_SPLIT");
        const bool latency_form = (sp && sp[0] == '1') || (!(sp && sp[0] == '0') && B <= PLP_REDUCE_SPLIT_MAXB(D, gs));
        if ((r2 && r2[0] == '1') || (!(r2 && r2[0] == '0') && !latency_form))
            return launch_reduce_r2c(B, m_max, D, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    }
    if (gs == 4) return launch_reduce_r_dg<D, 4>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    if (gs == 8) return launch_reduce_r_dg<D, 8>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    return launch_reduce_r_dg<D, 16>(B, m_max, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
}

#define PLP_CASE_RR(K) \
    case K: return launch_reduce_r_d<K>(B, m_max, gs, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);

// returns 0 when launched, 1 when this kernel does not apply (caller falls through to reduce_kernel)
int launch_reduce_r(long long B, int m_max, int d, const double* A, const double* b, const int* mrows,
                    double abs_tol, unsigned long long* keep, int* flags, double* r, double* xc, int* nlp,
                    hipStream_t st) {
    if (m_max < 1 || m_max > MAX_M || d < 1 || d > MAX_D) return 1;
    if (d <= 4 && m_max <= 32) {
        // up to 32 rows in d <= 4 (the bench shape; the stacks of Polytope.intersect): F3 / F2 one LP per lane (plp_reduce_lane.hip), at every batch size;
        // PLP_REDUCE_LANE=0 / 1: never / always (A/B).  Any switch of the lane-group forms keeps them.
        const char* ln = getenv("PLP_REDUCE_LANE");
        const bool other = getenv("PLP_REDUCE_SPLIT") || getenv("PLP_REDUCE_HALF") || getenv("PLP_REDUCE_MIX") || getenv("PLP_REDUCE_R8");
        // d = 4 (the walk in R^4, three waves per SIMD) -- measured after the walk's direction with three active rows became a
        // generalised cross product (scripts/debug/lane_d4_sweep.py, us lane-group / lane): (8,4) x 20 000 21 / 24, x 50 000 51 / 37;
        // (12,4) x 10 000 37 / 39, x 30 000 63 / 68, x 50 000 99 / 76; (16,4) x 2 000 43 / 44, x 5 000 68 / 51, x 10 000 87 / 70,
        // x 50 000 209 / 150; (20,4) x 2 000 61 / 58, x 5 000 102 / 75, x 50 000 359 / 276; (32,4) x 500 57 / 60, x 2 000 79 / 63,
        // x 5 000 144 / 89, x 10 000 183 / 123, x 50 000 483 / 325
        const long long minb = d == 4 ? (m_max >= 14 ? PLP_REDUCE_LANE4_MINB_ROWS14 : PLP_REDUCE_LANE4_MINB) : PLP_REDUCE_LANE_MINB;
        if ((ln && ln[0] == '1') || (!(ln && ln[0] == '0') && !other && B > minb))
            return launch_reduce_lane(B, m_max, d, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    }
    if (d > 8) {
        const char* e2 = getenv("PLP_REDUCE_R2");  // PLP_REDUCE_R2=0: d > 8 stays on the one-row-per-lane kernel (A/B)
        if (e2 && e2[0] == '0') return 1;
        return d <= 12 ? launch_reduce_r2a(B, m_max, d, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st)
                       : launch_reduce_r2b(B, m_max, d, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
    }
    const int gs = group_size_r(m_max);
    switch (d) {
        PLP_CASE_RR(1) PLP_CASE_RR(2) PLP_CASE_RR(3) PLP_CASE_RR(4)
        PLP_CASE_RR(5) PLP_CASE_RR(6) PLP_CASE_RR(7) PLP_CASE_RR(8)
        default: return 1;
    }
}

}  // namespace plp

#ifdef PLP_STAGE_STATS
extern "C" __attribute__((visibility("default"))) int plp_debug_stage_stats(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(plp::plp_stage_stats), 128) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(plp::plp_stage_stats), z, 128) != hipSuccess) return 1;
    }
    return 0;
}
#endif
