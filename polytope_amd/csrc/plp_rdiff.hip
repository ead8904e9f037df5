// plp_rdiff.hip -- Chebyshev LPs on ROW SUBSETS of one resident constraint table (gfx950): the LPs of region_diff's
// search (polytope/polytope.py:2148-2152, :2212-2224, :2272-2274).
//
// Every LP of that search is the Chebyshev ball of {A[rows] x <= B[rows]} for an index list `rows` into ONE table of
// m + 2M rows (the minuend's rows, the subtrahends' new rows, and their negations).  The reference builds a Polytope
// per index list; here the table stays in HBM/L2 and a batch is described by index lists only (CSR: off[nlp + 1],
// rows[]), which the lane groups gather while loading -- nothing but 4-byte indices crosses PCIe per search step.
//
//   cheby_gather_r_kernel<D> : lists of up to 16 / 32 / 64 rows (three size classes, one launch) on the four-rows-per-lane engine (d <= 8)
//   (lists beyond 64 rows, and d > 8: cheby_gather_lds_kernel in plp_lds.hip, dictionary in LDS)
//
// out[p] = the radius as cheby_ball reads it (:1289-1297): x[-1] if the LP is optimal with r >= 0, 0 if optimal with
// r < 0, NaN if the LP ended with any other status (the reference reads 0 there; the search must know the difference:
// a cell without a verdict is solved again below, a cell SOLVED as empty is not).
#include <stdlib.h>

#include "plp_cheby_r_impl.hpp"

namespace plp {

// one lane group = one LP of the size class GS (lists of up to 4 * GS rows); q = position inside the class
template <int D, int GS, int R = RowsPerLane<D>::value>
__device__ __forceinline__ void gather_body(long long q0, long long nlp, const int* __restrict__ off,
                                            const int* __restrict__ rows, const int* __restrict__ sel,
                                            const double* __restrict__ A, const double* __restrict__ b,
                                            double* __restrict__ out, int force_retry) {
    const Grp g(GS);
    constexpr int gpb = RBLK / GS;
    const int gib = threadIdx.x / GS;
    const int row0 = g.gl * R;
    const long long q = q0 * gpb + gib;
    const bool valid = q < nlp;
    const int p = valid ? sel[q] : 0;          // position of this LP in the batch (lists are grouped by size class)
    const int o = valid ? off[p] : 0;
    const int m = valid ? off[p + 1] - o : 0;
    double x[D + 1];
    const int st = cheby_r_solve<D, GS, R>(
        g, valid, m, row0, [&](int rr, int kk) { return A[(long long)rows[o + rr] * D + kk]; },
        [&](int rr) { return b[rows[o + rr]]; }, x, force_retry);
    if (valid & (g.gl == 0)) out[p] = st != ST_OPT ? __builtin_nan("") : (x[D] >= 0.0 ? x[D] : 0.0);
}

// All three size classes in ONE launch (the search pays per launch, not per LP): workgroups [0, nb0) take the n0 lists
// of up to 16 rows, [nb0, nb0 + nb1) the n1 lists of up to 32, the rest the n2 lists of up to 64.
template <int D>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void cheby_gather_r_kernel(int nb0, int nb1, long long n0, long long n1,
                                                                                 long long n2, const int* __restrict__ off,
                                                                                 const int* __restrict__ rows,
                                                                                 const int* __restrict__ sel,
                                                                                 const double* __restrict__ A,
                                                                                 const double* __restrict__ b,
                                                                                 double* __restrict__ out, int force_retry) {
    const int bid = blockIdx.x;
    if (bid < nb0) gather_body<D, 4>(bid, n0, off, rows, sel, A, b, out, force_retry);
    else if (bid < nb0 + nb1) gather_body<D, 8>(bid - nb0, n1, off, rows, sel + n0, A, b, out, force_retry);
    else gather_body<D, 16>(bid - nb0 - nb1, n2, off, rows, sel + n0 + n1, A, b, out, force_retry);
}

// The same with ONE row per lane (groups of 16 / 32 / 64 lanes): a pivot costs the wavefront about half the instructions,
// so a lone wavefront finishes its LP sooner.  A search batch is a few dozen to a few hundred LPs -- far fewer
// wavefronts than the chip holds either way -- and the search waits for the slowest of them: latency, not throughput.
template <int D>
__global__ __launch_bounds__(RBLK, (D <= 4 ? 3 : 1)) void cheby_gather_r1_kernel(int nb0, int nb1, long long n0, long long n1,
                                                                                  long long n2, const int* __restrict__ off,
                                                                                  const int* __restrict__ rows,
                                                                                  const int* __restrict__ sel,
                                                                                  const double* __restrict__ A,
                                                                                  const double* __restrict__ b,
                                                                                  double* __restrict__ out, int force_retry) {
    const int bid = blockIdx.x;
    if (bid < nb0) gather_body<D, 16, 1>(bid, n0, off, rows, sel, A, b, out, force_retry);
    else if (bid < nb0 + nb1) gather_body<D, 32, 1>(bid - nb0, n1, off, rows, sel + n0, A, b, out, force_retry);
    else gather_body<D, 64, 1>(bid - nb0 - nb1, n2, off, rows, sel + n0 + n1, A, b, out, force_retry);
}

// After the LP kernels of a batch (same stream): copy the n radii to host-mapped memory and raise the batch's sequence
// number there; the host spins on that word instead of paying a stream synchronisation per batch.
__global__ __launch_bounds__(256) void rdiff_publish_kernel(long long n, const double* __restrict__ src,
                                                            double* __restrict__ host_out,
                                                            unsigned long long* __restrict__ host_flag,
                                                            unsigned long long seq) {
    for (long long k = threadIdx.x; k < n; k += 256) host_out[k] = src[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int D>
static int launch_gather_d(long long n0, long long n1, long long n2, const int* off, const int* rows, const int* sel,
                           const double* A, const double* b, double* out, hipStream_t st) {
    // small batches (the search's): one row per lane; PLP_RDIFF_R1=0 / 1: never / always (A/B)
    const char* r1 = getenv("PLP_RDIFF_R1");
    const bool lowlat = r1 ? r1[0] == '1' : (n0 + n1 + n2 <= 2048);
    if (lowlat) {
        const long long b0 = (n0 + RBLK / 16 - 1) / (RBLK / 16), b1 = (n1 + RBLK / 32 - 1) / (RBLK / 32),
                        b2 = (n2 + RBLK / 64 - 1) / (RBLK / 64);
        if (b0 + b1 + b2 < 1) return 0;
        hipLaunchKernelGGL((cheby_gather_r1_kernel<D>), dim3((unsigned)(b0 + b1 + b2)), dim3(RBLK), 0, st, (int)b0, (int)b1, n0,
                           n1, n2, off, rows, sel, A, b, out, force_retry_env());
        return 0;
    }
    const long long nb0 = (n0 + RBLK / 4 - 1) / (RBLK / 4), nb1 = (n1 + RBLK / 8 - 1) / (RBLK / 8),
                    nb2 = (n2 + RBLK / 16 - 1) / (RBLK / 16);
    const long long blocks = nb0 + nb1 + nb2;
    if (blocks < 1) return 0;
    if (blocks > 2147483647ll) return 2;
    hipLaunchKernelGGL((cheby_gather_r_kernel<D>), dim3((unsigned)blocks), dim3(RBLK), 0, st, (int)nb0, (int)nb1, n0, n1, n2,
                       off, rows, sel, A, b, out, force_retry_env());
    return 0;
}

// LPs sel[0 .. n0 + n1 + n2) of the batch: n0 with at most 16 rows, then n1 with at most 32, then n2 with at most 64;
// d <= 8; returns 1 when it does not apply
int launch_cheby_gather_r(int d, long long n0, long long n1, long long n2, const int* off, const int* rows, const int* sel,
                          const double* A, const double* b, double* out, hipStream_t st) {
    switch (d) {
        case 1: return launch_gather_d<1>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 2: return launch_gather_d<2>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 3: return launch_gather_d<3>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 4: return launch_gather_d<4>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 5: return launch_gather_d<5>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 6: return launch_gather_d<6>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 7: return launch_gather_d<7>(n0, n1, n2, off, rows, sel, A, b, out, st);
        case 8: return launch_gather_d<8>(n0, n1, n2, off, rows, sel, A, b, out, st);
        default: return 1;
    }
}

void launch_rdiff_publish(long long n, const double* src, double* host_out, unsigned long long* host_flag,
                          unsigned long long seq, hipStream_t st) {
    hipLaunchKernelGGL(rdiff_publish_kernel, dim3(1), dim3(256), 0, st, n, src, host_out, host_flag, seq);
}

}  // namespace plp
