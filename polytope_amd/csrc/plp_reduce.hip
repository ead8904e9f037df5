// plp_reduce.hip -- fused redundancy removal for a packed batch of H-polytopes (gfx950):
// dispatch (launch_reduce) + the one-row-per-lane kernel that serves d > 8; d <= 8 goes to the
// four-rows-per-lane kernel of plp_reduce_r.hip, which is ~1.3x faster on the bench workload.
//
// Reference behaviour restated (polytope/polytope.py:1053-1163, `reduce`), per polytope:
//   1. is_fulldim -> cheby_ball: LP F1, r > abs_tol                      (:1081, :962-985, :1241-1300)
//   2. pairwise parallel-row dedupe on unit rows                          (:1094-1112)
//   3. early return when neq <= nx+1                                      (:1114-1116)
//   4. if neq > 3 nx: bounding box = 2 nx LPs F3, prefilter rows          (:1118-1134, :1367-1409)
//   5. early return when neq <= nx+1                                      (:1136-1138)
//   6. one redundancy LP F2 per remaining row k (h[k] += 0.1 ... -= 0.1)  (:1142-1160)
// Output: 64-bit keep mask over the INPUT rows, flags, Chebyshev ball, number of LPs solved.
//
// Mapping: a 256-thread workgroup takes a tile of NG = 256/GS polytopes (GS lanes per group,
// GS >= rows).  The tile's rows are read from HBM once, coalesced, into LDS.  Group p then runs
// the whole pipeline of polytope p: lane i keeps row i (a_i, b_i) in VGPRs for the entire
// sequence F1 -> dedupe -> 2d F3 -> prefilter -> one F2 per surviving row; the 64/GS groups of a
// wavefront advance in lockstep, LP by LP.  The LPs of one polytope share everything but the
// objective and one right-hand side entry, so setting one up is a handful of register moves:
//   * F2/F3 start from the dictionary translated to the Chebyshev centre (b - A xc > 0), which is
//     primal feasible: no phase 1;
//   * the optimal value is read off the dictionary (zeta = -negz), not recomputed from x.
// LDS is only read after the staging barrier (row k's coefficients = objective of F2(k), other
// rows for the dedupe), so the groups never synchronise with each other.
// HBM traffic = 8 m (d+1) bytes in + 24 + 8 d bytes out per polytope.
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_reduce_general.hpp"
#include "plp_simplex.hpp"

namespace plp {

thread_local unsigned long long* t_reduce_ctr = nullptr;
thread_local unsigned long long* t_reduce_retry = nullptr;
thread_local unsigned long long t_reduce_epoch = 0ull;

static inline size_t reduce_smem_bytes(int gs, int D) {
    const int NG = BLOCK / gs;
    return ((size_t)NG * gs * (D + 1) * 8 + 15) & ~(size_t)15;
}

// Waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument).
// Measured on MI355X at d=3 (100k polytopes, m=16): 3 waves 0.820 ms, 4 waves 0.719 ms,
// 5 waves 0.703 ms (24 VGPRs spilled outside the pivot loop), 6 waves 0.706 ms: the kernel is
// VALU-issue bound from ~4 waves on, and the 5/6-wave builds pay for their spills with scratch
// traffic (WRITE_SIZE 4.7 MB -> 132 MB per launch).  Larger d needs the registers more than the
// occupancy.
#ifndef PLP_REDUCE_WAVES
#define PLP_REDUCE_WAVES(D) ((D) <= 4 ? 4 : ((D) <= 8 ? 3 : 2))
#endif

template <int D>
__global__ __launch_bounds__(BLOCK, PLP_REDUCE_WAVES(D)) void reduce_kernel(long long B, int m_max, int gs,
                                                       const double* __restrict__ Ag,
                                                       const double* __restrict__ bg,
                                                       const int* __restrict__ mrows, double abs_tol,
                                                       int retry_only,
                                                       unsigned long long* __restrict__ keep_out,
                                                       int* __restrict__ flags_out,
                                                       double* __restrict__ r_out,
                                                       double* __restrict__ xc_out,
                                                       int* __restrict__ nlp_out,
                                                       const unsigned long long* __restrict__ retry_word,
                                                       unsigned long long epoch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // second pass: the fast kernels raise the call's word when they hand a polytope back; normally they did not
    // The word lives in a ring of 64 (slot = epoch & 63, raised with atomicMax): a value BELOW this call's epoch means no
    // tile of this call asked for the second pass; this call's own epoch means some did; a LARGER value is a later call
    // (64 or more calls of one context in flight on other streams) that took the slot over -- then nothing is known and
    // the flags of the batch are swept as if the word were not there.
    if (retry_only && retry_word && *retry_word < epoch) return;
    const int NG = BLOCK / gs;
    const int gib = threadIdx.x / gs;

    if (retry_only) {
        // normally nothing was handed back: all flag loads of this workgroup's tiles are issued at once (one
        // memory round trip instead of one per tile of the sweep below) and the workgroup leaves
        bool any = false;
        for (long long tile = (long long)blockIdx.x * NG; tile < B; tile += (long long)gridDim.x * NG)
            any = any | ((tile + gib < B) && (flags_out[tile + gib] & RF_RETRY) != 0);
        if (!__syncthreads_or(any)) return;
    }
    for (long long tile = (long long)blockIdx.x * NG; tile < B; tile += (long long)gridDim.x * NG) {
        const int ntile = (B - tile) < NG ? (int)(B - tile) : NG;
        // second pass after reduce_r_kernel: only polytopes it flagged RF_RETRY (tiles without one are skipped)
        bool mine = true;
        if (retry_only) {
            mine = (gib < ntile) && (flags_out[tile + gib] & RF_RETRY) != 0;
            if (!__syncthreads_or(mine)) continue;
        }
        reduce_general_tile<D, BLOCK>(smem_raw, tile, ntile, mine, m_max, gs, Ag, bg, mrows, abs_tol, keep_out, flags_out, r_out,
                                      xc_out, nlp_out);
    }
}

template <int D>
static int launch_reduce_d(long long B, int m_max, int gs, const double* A, const double* b, const int* mrows,
                           double abs_tol, int retry_only, unsigned long long* keep, int* flags, double* r,
                           double* xc, int* nlp, hipStream_t st) {
    const size_t smem = reduce_smem_bytes(gs, D);
    const long long NG = BLOCK / gs;
    long long blocks = (B + NG - 1) / NG;
    if (blocks > (1ll << 20)) blocks = 1ll << 20;  // one tile per block: the dispatcher balances the tail
    if (blocks < 1) blocks = 1;
    if (retry_only && blocks > 256 * 8) blocks = 256 * 8;  // mostly flag reads: a grid-stride sweep
    hipLaunchKernelGGL(reduce_kernel<D>, dim3((unsigned)blocks), dim3(BLOCK), smem, st, B, m_max, gs, A, b, mrows,
                       abs_tol, retry_only, keep, flags, r, xc, nlp, retry_only ? t_reduce_retry : nullptr, t_reduce_epoch);
    return 0;
}

#define PLP_CASE_R(K) \
    case K: return launch_reduce_d<K>(B, m_max, gs, A, b, mrows, abs_tol, retry, keep, flags, r, xc, nlp, st);

// phase 0: everything (the fast kernel, then the pass that redoes what it flagged RF_RETRY);  phase 1: the first launch
// only -- the caller looks at the flags itself and asks for phase 2 (that second pass) when it finds RF_RETRY.  The
// synchronous host entry point does so for small batches: the normally idle second launch is half of their device time.
int launch_reduce_phase(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                        unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st, int phase) {
    const int gs = group_size_for(m_max);
    if (gs < 0 || d < 1 || d > MAX_D) return 2;
    const char* one = getenv("PLP_REDUCE_1ROW");
    int retry = 0;
    if (phase == 2) retry = 1;
    else if (!(one && one[0] == '1')) {
        const int rc = launch_reduce_r(B, m_max, d, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        if (rc == 3) return 0;   // complete: no polytope can ask for the second pass
        if (rc == 0) {
            if (phase == 1) return 0;
            retry = 1;
        }
    }
    switch (d) {
        PLP_CASE_R(1) PLP_CASE_R(2) PLP_CASE_R(3) PLP_CASE_R(4) PLP_CASE_R(5) PLP_CASE_R(6)
        PLP_CASE_R(7) PLP_CASE_R(8) PLP_CASE_R(9) PLP_CASE_R(10) PLP_CASE_R(11) PLP_CASE_R(12)
        PLP_CASE_R(13) PLP_CASE_R(14) PLP_CASE_R(15) PLP_CASE_R(16)
        default: return 2;
    }
}

int launch_reduce(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                  unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    const int gs = group_size_for(m_max);
    if (gs < 0 || d < 1 || d > MAX_D) return 2;
    // default for d <= 8: four rows per lane (plp_reduce_r.hip); PLP_REDUCE_1ROW=1 keeps this kernel
    // Its F2/F3 LPs run on the fast pivot path, which hands a polytope back (RF_RETRY) when an LP
    // needs Bland's rule; the second launch below redoes exactly those with this file's kernel.
    const char* one = getenv("PLP_REDUCE_1ROW");
    int retry = 0;
    if (!(one && one[0] == '1')) {
        const int rc = launch_reduce_r(B, m_max, d, A, b, mrows, abs_tol, keep, flags, r, xc, nlp, st);
        if (rc == 3) return 0;   // complete (one polytope per workgroup, Bland's rule inside the LPs): no second pass
        if (rc == 0) retry = 1;
    }
    switch (d) {
        PLP_CASE_R(1) PLP_CASE_R(2) PLP_CASE_R(3) PLP_CASE_R(4) PLP_CASE_R(5) PLP_CASE_R(6)
        PLP_CASE_R(7) PLP_CASE_R(8) PLP_CASE_R(9) PLP_CASE_R(10) PLP_CASE_R(11) PLP_CASE_R(12)
        PLP_CASE_R(13) PLP_CASE_R(14) PLP_CASE_R(15) PLP_CASE_R(16)
        default: return 2;
    }
}

}  // namespace plp
