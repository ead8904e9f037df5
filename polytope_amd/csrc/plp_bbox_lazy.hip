// plp_bbox_lazy.hip -- fused bounding boxes, one polytope of up to 64 rows per wavefront (polytope/polytope.py:1314-1411):
// the Chebyshev LP on the one-LP-per-wavefront engine (plp_wide.hpp), then the 2d LPs min / max x_k from its centre
//   WDENSE (d = 5..13): on the same dense engine (wide::solve_dense: wave-uniform pivots, Bland's rule inside),
//   otherwise (d = 14..16): WITHOUT a stored dictionary (plp_lazy.hpp).
// d = 9..16 always come here, d = 5..8 with more than 32 rows (the lane-group kernel bbox_r_kernel holds one or two such
// polytopes per wavefront and pays select chains for the pivot column).  Same contract as bbox_r_kernel
// (plp_cheby_r_impl.hpp, d <= 8): status 0 = lb / ub hold the box (+-inf where an LP is unbounded, :1376 / :1398),
// status 1 = not handled here (empty / flat / unbounded-ball polytopes, LPs that ask for Bland's rule or run past the
// step limit): the caller solves the generic LPs for those.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_lazy.hpp"
#include "plp_wide.hpp"

namespace plp {

namespace {
constexpr double BBOX_LAZY_MIN_R = 1e-6;  // (bbox_r_kernel's BBOX_MIN_R)
}

template <int D, bool WDENSE>
__global__ __launch_bounds__(64, 3) void bbox_lazy_kernel(long long B, int m_max, const double* __restrict__ A,
                                                          const double* __restrict__ b, const int* __restrict__ mrows,
                                                          double* __restrict__ lb, double* __restrict__ ub,
                                                          int* __restrict__ status, signed char* __restrict__ basis8,
                                                          double* __restrict__ centre) {
    constexpr int NC = D + 1;
    __shared__ __attribute__((aligned(16))) double sA[64 * D];
    __shared__ wide::WideShared<NC> sh;
    const int lane = threadIdx.x;
    const long long p = blockIdx.x;
    if (p >= B) return;
    const int m = mrows ? mrows[p] : m_max;
    const bool has = lane < m;
    // ---- F1 (set-up as cheby_w_kernel); my row also goes to LDS for the lazy LPs
    typename wide::RowVec<NC>::type Tv = (typename wide::RowVec<NC>::type)(0.0);
    double T16 = 0.0;
    double nrm2 = 0.0;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double v = has ? A[(p * m_max + lane) * D + k] : 0.0;
        ROW_SET(k, v);
        sA[lane * D + k] = v;
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
    else st = wide::wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bi / nrm, iters);
    const double mine = rowneg ? -beta : beta;
    double x[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const uint64_t ob = __ballot(rowvar == j);
        x[j] = ob ? wide::uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
    }
    const bool ok = (st == ST_OPT) & (x[D] >= BBOX_LAZY_MIN_R);
    bool handed = !ok;
    // ---- my slack at the centre (bbox_r_kernel: s by an fma chain, beta = max(b - s, 0))
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) s = fma(has ? sA[lane * D + k] : 0.0, ok ? x[k] : 0.0, s);
    const double be0 = (ok & has) ? fmax(bi - s, 0.0) : 0.0;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    for (int it = 0; it < 2 * D; ++it) {  // lower_0, upper_0, lower_1, upper_1, ...
        const int kx = it >> 1;
        const bool up = it & 1;
        double xck = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) xck = (k == kx) ? x[k] : xck;
        double val = qnan;
        if (__builtin_amdgcn_readfirstlane((int)ok)) {
            double negz = 0.0;
            int s2;
            signed char* bo = basis8 ? basis8 + ((size_t)p * 2 * D + it) * D : nullptr;  // the LP's final basis, for the verifier
            if constexpr (WDENSE)
                s2 = wide::solve_dense<D>(lane, m, sA, (lane == kx) ? (up ? -1.0 : 1.0) : 0.0, be0, has, negz,
                                          *reinterpret_cast<wide::WideShared<D>*>(&sh), bo);
            else
                s2 = lazy::solve<D>(lane, m, sA, (lane == kx) ? (up ? -1.0 : 1.0) : 0.0, be0, has, negz, bo);
            // zeta = c.x' = -negz ; x_k = xc_k + x'_k ; lower: c = +e_k, upper: c = -e_k
            if (s2 == ST_OPT) val = up ? (xck + negz) : (xck - negz);
            else if (s2 == ST_UNBND) val = up ? pinf : -pinf;
            else handed = true;
        }
        if (lane == 0) (up ? ub : lb)[p * D + kx] = ok ? val : qnan;
    }
    if (lane == 0) status[p] = handed ? 1 : 0;
    if (centre && lane == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) centre[(size_t)p * D + k] = ok ? x[k] : qnan;
    }
}

// ---- small batches: NW wavefronts per polytope (the pattern of reduce_wsplit_kernel, plp_reduce_r_impl.hpp).  One workgroup
// per polytope: wavefront 0 runs F1 and leaves the centre in LDS, then wavefront w solves every NW-th of the 2d box LPs on the
// rows they share -- each LP exactly as in bbox_lazy_kernel (same engine, same set-up): lb / ub / status bit for bit.
template <int D, int NW, bool WDENSE>
__global__ __launch_bounds__(64 * NW) void bbox_wsplit_kernel(long long B, int m_max, const double* __restrict__ A,
                                                              const double* __restrict__ b, const int* __restrict__ mrows,
                                                              double* __restrict__ lb, double* __restrict__ ub,
                                                              int* __restrict__ status, signed char* __restrict__ basis8,
                                                              double* __restrict__ centre) {
    constexpr int NC = D + 1;
    __shared__ __attribute__((aligned(16))) double sA[64 * D];
    __shared__ wide::WideShared<NC> shw[NW];
    __shared__ double sx[NC + 1];
    __shared__ unsigned sflag[2];   // [0]: F1 gave a usable ball, [1]: some LP was handed back
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long p = blockIdx.x;
    if (p >= B) return;
    const int m = mrows ? mrows[p] : m_max;
    const bool has = lane < m;
    double bi = 0.0;
    if (w == 0) {
        // ---- F1 (set-up as bbox_lazy_kernel); my row also goes to LDS for the box LPs
        wide::WideShared<NC>& sh = shw[0];
        typename wide::RowVec<NC>::type Tv = (typename wide::RowVec<NC>::type)(0.0);
        double T16 = 0.0;
        double nrm2 = 0.0;
        bool finite = true;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double v = has ? A[(p * m_max + lane) * D + k] : 0.0;
            ROW_SET(k, v);
            sA[lane * D + k] = v;
            nrm2 = nrm2 + v * v;
            finite = finite & isfinite(v);
        }
        bi = has ? b[p * m_max + lane] : 0.0;
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
        wide::wave_sync();
        int st, iters = 0;
        if (bad) st = ST_NUM;
        else if (infeasible0) st = ST_INFEAS;
        else st = wide::wide_run<NC>(lane, m, Tv, T16, beta, rowvar, rowneg, rowact, sh, NC, true, bi / nrm, iters);
        const double mine = rowneg ? -beta : beta;
        double xr = 0.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const uint64_t ob = __ballot(rowvar == j);
            const double xj = ob ? wide::uniform_lane(mine, __ffsll((long long)ob) - 1) : 0.0;
            if (lane == 0) sx[j] = xj;
            if (j == D) xr = xj;
        }
        if (lane == 0) {
            sflag[0] = ((st == ST_OPT) & (xr >= BBOX_LAZY_MIN_R)) ? 1u : 0u;
            sflag[1] = 0u;
        }
    }
    __syncthreads();
    if (w != 0) bi = has ? b[p * m_max + lane] : 0.0;
    const bool ok = sflag[0] != 0u;
    double x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = sx[k];
    // ---- my slack at the centre (bbox_r_kernel: s by an fma chain, beta = max(b - s, 0))
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) s = fma(has ? sA[lane * D + k] : 0.0, ok ? x[k] : 0.0, s);
    const double be0 = (ok & has) ? fmax(bi - s, 0.0) : 0.0;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    bool handed = false;
    for (int it = w; it < 2 * D; it += NW) {  // lower_0, upper_0, lower_1, upper_1, ...: wavefront w every NW-th
        const int kx = it >> 1;
        const bool up = it & 1;
        double xck = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) xck = (k == kx) ? x[k] : xck;
        double val = qnan;
        if (__builtin_amdgcn_readfirstlane((int)ok)) {   // (the same in every lane, and said so: the LP runs on full wavefronts)
            double negz = 0.0;
            int s2;
            signed char* bo = basis8 ? basis8 + ((size_t)p * 2 * D + it) * D : nullptr;  // the LP's final basis, for the verifier
            if constexpr (WDENSE)
                s2 = wide::solve_dense<D>(lane, m, sA, (lane == kx) ? (up ? -1.0 : 1.0) : 0.0, be0, has, negz,
                                          *reinterpret_cast<wide::WideShared<D>*>(&shw[w]), bo);
            else
                s2 = lazy::solve<D>(lane, m, sA, (lane == kx) ? (up ? -1.0 : 1.0) : 0.0, be0, has, negz, bo);
            if (s2 == ST_OPT) val = up ? (xck + negz) : (xck - negz);
            else if (s2 == ST_UNBND) val = up ? pinf : -pinf;
            else handed = true;
        }
        if (lane == 0) (up ? ub : lb)[p * D + kx] = ok ? val : qnan;
    }
    if (handed & (lane == 0)) atomicOr(&sflag[1], 1u);
    __syncthreads();
    if ((w == 0) & (lane == 0)) status[p] = (!ok | (sflag[1] != 0u)) ? 1 : 0;
    if (centre && (w == 0) && lane < D) centre[(size_t)p * D + lane] = ok ? sx[lane] : qnan;
}

#ifndef PLP_BBOX_WSPLIT_MAXB
// Measured (scripts/debug/bbox_wsplit_sweep.py, ms in use / four wavefronts per polytope): (64,8) B = 1 0.094 / 0.051, 500 0.116 /
// 0.071, 2 000 0.168 / 0.129; (32,6) 500 0.056 / 0.041; (64,12) 500 0.136 / 0.077, 2 000 0.168 / 0.151, 4 000 0.228 / 0.258;
// (64,16) 1 000 0.144 / 0.098
#define PLP_BBOX_WSPLIT_MAXB 2000
#endif
#ifndef PLP_BBOX_WDENSE_MAXD
#define PLP_BBOX_WDENSE_MAXD 13  // (as PLP_REDUCE_WDENSE_MAXD: beyond, the LPs are too short to pay for a dictionary reload)
#endif

template <int D>
static int launch_bbox_lazy_d(long long B, int m_max, const double* A, const double* b, const int* mrows, double* lb,
                              double* ub, int* status, hipStream_t st, BoxHandover* ho) {
    if (B > 2147483647ll) return 1;
    signed char* b8 = ho ? ho->basis8 : nullptr;
    double* ctr = ho ? ho->centre : nullptr;
    if (ho) ho->mode = (b8 && ctr) ? 1 : 0;
    const char* wd = getenv("PLP_BBOX_WDENSE");  // 0 / 1: never / always the dense engine for the 2d LPs (A/B)
    // small batches: four wavefronts per polytope (PLP_BBOX_WSPLIT=0 / 1: never / always; PLP_BBOX_WSPLIT_MAXB)
    const char* ws = getenv("PLP_BBOX_WSPLIT");
    const char* wb = getenv("PLP_BBOX_WSPLIT_MAXB");
    const long long maxb = wb ? atoll(wb) : PLP_BBOX_WSPLIT_MAXB;
    if ((ws && ws[0] == '1') || (!(ws && ws[0] == '0') && B <= maxb)) {
        if (wd ? wd[0] == '1' : (D <= PLP_BBOX_WDENSE_MAXD))
            hipLaunchKernelGGL((bbox_wsplit_kernel<D, 4, true>), dim3((unsigned)B), dim3(256), 0, st, B, m_max, A, b, mrows, lb, ub,
                               status, b8, ctr);
        else
            hipLaunchKernelGGL((bbox_wsplit_kernel<D, 4, false>), dim3((unsigned)B), dim3(256), 0, st, B, m_max, A, b, mrows, lb, ub,
                               status, b8, ctr);
        return 0;
    }
    if (wd ? wd[0] == '1' : (D <= PLP_BBOX_WDENSE_MAXD))
        hipLaunchKernelGGL((bbox_lazy_kernel<D, true>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(64), 0, st, B, m_max, A, b, mrows,
                           lb, ub, status, b8, ctr);
    else
        hipLaunchKernelGGL((bbox_lazy_kernel<D, false>), dim3((unsigned)(B < 1 ? 1 : B)), dim3(64), 0, st, B, m_max, A, b, mrows,
                           lb, ub, status, b8, ctr);
    return 0;
}

#define PLP_CASE_BL(K) case K: return launch_bbox_lazy_d<K>(B, m_max, A, b, mrows, lb, ub, status, st, ho);

// d = 5..16, m_max <= 64; returns 1 when it does not apply
int launch_bbox_lazy(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* lb,
                     double* ub, int* status, hipStream_t st, BoxHandover* ho) {
    if (ho) ho->mode = 0;
    if (m_max < 1 || m_max > 64 || B < 1) return 1;
    switch (d) {
        PLP_CASE_BL(5) PLP_CASE_BL(6) PLP_CASE_BL(7) PLP_CASE_BL(8)
        PLP_CASE_BL(9) PLP_CASE_BL(10) PLP_CASE_BL(11) PLP_CASE_BL(12)
        PLP_CASE_BL(13) PLP_CASE_BL(14) PLP_CASE_BL(15) PLP_CASE_BL(16)
        default: return 1;
    }
}

}  // namespace plp
