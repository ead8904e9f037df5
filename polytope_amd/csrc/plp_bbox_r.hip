// plp_bbox_r.hip -- launcher of the fused bounding-box batches (kernel: plp_cheby_r_impl.hpp, bbox_r_kernel).
#include <stdlib.h>

#include "plp_cheby_r_impl.hpp"

namespace plp {

int launch_bbox_lane(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb, double* ub,
                     int* status, hipStream_t st, double* xfin);

template <int D, int GS>
static int launch_bbox_r_dg(long long B, int m_max, const double* A, const double* b, const int* mrows, double* lb,
                            double* ub, int* status, hipStream_t st, BoxHandover* ho) {
    signed char* b8 = ho ? ho->basis8 : nullptr;
    double* ctr = ho ? ho->centre : nullptr;
    if (ho) ho->mode = (b8 && ctr) ? 1 : 0;
    // small batches: one polytope per wavefront, its 2d LPs over the lane groups (PLP_BBOX_SPLIT=0 / 1: never / always).
    // Measured device time per call, batch form: (16,3) B = 64 34 us, (32,6) 123 us, (64,8) 251 us.
    const char* sp = getenv("PLP_BBOX_SPLIT");
    // ((64,8) at B = 4096: 275 us batch form, 345 us latency form -- four groups per wavefront there)
    if ((sp && sp[0] == '1') || (!(sp && sp[0] == '0') && B <= (GS >= 16 ? 1024 : 4096))) {
        hipLaunchKernelGGL((bbox_split_kernel<D, GS>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(RBLK), 0, st, B, m_max, A, b, mrows,
                           lb, ub, status, force_retry_env(), b8, ctr);
        return 0;
    }
    constexpr long long gpb = RBLK / GS;
    const long long blocks = (B + gpb - 1) / gpb;
    if (blocks > 2147483647ll) return 1;
    hipLaunchKernelGGL((bbox_r_kernel<D, GS>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(RBLK), 0, st, B, m_max,
                       A, b, mrows, lb, ub, status, force_retry_env(), b8, ctr);
    return 0;
}

template <int D>
static int launch_bbox_r_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double* lb,
                           double* ub, int* status, hipStream_t st, BoxHandover* ho) {
    PLP_DISPATCH_GS(RowsPerLane<D>::value, m_max, (launch_bbox_r_dg<D, GSV>(B, m_max, A, b, mrows, lb, ub, status, st, ho)));
}

#define PLP_CASE_BB(K) case K: return launch_bbox_r_d<K>(B, m_max, A, b, mrows, lb, ub, status, st, ho);

// returns 0 when launched, 1 when no fused kernel applies (the caller uses the generic LPs); d = 9..16: plp_bbox_lazy.hip
int launch_bbox(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb,
                double* ub, int* status, hipStream_t st, BoxHandover* ho) {
    if (ho) ho->mode = 0;
    if (m_max < 1 || m_max > MAX_M || B < 1) return 1;
    // up to 32 rows in d <= 3: the 2 d box LPs one LP per lane (plp_reduce_lane.hip, bbox_lane_kernel); PLP_BBOX_LANE=0: never (A/B)
    {
        const char* bl = getenv("PLP_BBOX_LANE");
        // (d = 4 through walk4, measured and not enabled: 20-30 % ahead of the lane-group kernels from 2 000 polytopes on --
        // (16,4) x 100 000 209 -> 173 us, (32,4) 400 -> 290 -- but with no dedupe in front of it (bounding_box has none, ref
        // :1314-1411) 20 of 6 407 polytopes with rows duplicated 1e-16 .. 1e-5 rad apart came out with a bound that is off by 2
        // (scripts/soak_lane.py 150 101, trial 97); the walk in R^3 hands such pairs back)
        if (d <= 3 && m_max <= 32 && !(bl && bl[0] == '0') && !getenv("PLP_BBOX_SPLIT") &&
            launch_bbox_lane(B, m_max, d, A, b, mrows, lb, ub, status, st, ho ? ho->xfin : nullptr) == 0) {
            if (ho) ho->mode = ho->xfin ? 2 : 0;
            return 0;
        }
    }
    // d = 5..8 with more than 32 rows, beyond the latency form's batch sizes: one polytope per wavefront, wave-uniform
    // pivots (plp_bbox_lazy.hip).  Measured (scripts/debug/bbox_wide_ab.py, ms): (64,8) B = 5 000 0.523 -> 0.262, B = 20 000
    // 1.31 -> 0.82, (48,6) 0.52 -> 0.45, (33,5) 0.32 -> 0.30; the latency form keeps B <= 1024 ((64,8) B = 1000: 0.103
    // against 0.157: its 2d LPs run side by side), the lane groups 32 rows and fewer ((32,6) 0.313 against 0.339).
    // PLP_BBOX_WIDE=0 / 1: never / every shape with d >= 5 (A/B)
    const char* bw = getenv("PLP_BBOX_WIDE");
    // (round 4: and every batch of up to 2 000 polytopes, any row count: four wavefronts per polytope there, bbox_wsplit_kernel,
    // 1.2x .. 1.85x ahead of the latency form; PLP_BBOX_SPLIT set: the lane-group forms keep their A/B meaning)
    const bool small_batch = !bw && !getenv("PLP_BBOX_SPLIT") && !getenv("PLP_BBOX_WSPLIT") && B <= 2000;
    if (d >= 5 && d <= 8 && (bw ? bw[0] == '1' : ((m_max > 32 && B > 1024) || small_batch)) &&
        launch_bbox_lazy(B, m_max, d, A, b, mrows, lb, ub, status, st, ho) == 0)
        return 0;
    switch (d) {
        PLP_CASE_BB(1) PLP_CASE_BB(2) PLP_CASE_BB(3) PLP_CASE_BB(4)
        PLP_CASE_BB(5) PLP_CASE_BB(6) PLP_CASE_BB(7) PLP_CASE_BB(8)
        default: return launch_bbox_lazy(B, m_max, d, A, b, mrows, lb, ub, status, st, ho);
    }
}

}  // namespace plp
