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
    const Grp g(gs);
    const int NG = BLOCK / gs;
    const int gib = threadIdx.x / gs;
    const int i = g.gl;
    double* sA = reinterpret_cast<double*>(smem_raw);   // [NG][gs][D]
    double* sb = sA + (size_t)NG * gs * D;               // [NG][gs]
    const double* myA = sA + (size_t)gib * gs * D;       // rows of my polytope
    const double* myb = sb + (size_t)gib * gs;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);

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
        // ---------------------------------------------------------------- stage rows in LDS
        __syncthreads();  // the previous tile's readers are done
        {
            const int rowsz = m_max * D;
            const int totA = ntile * rowsz;
            const double* src = Ag + tile * rowsz;
            for (int idx = threadIdx.x; idx < totA; idx += BLOCK) {
                const int p = idx / rowsz, rem = idx - p * rowsz;
                const int row = rem / D, k = rem - row * D;
                sA[((size_t)p * gs + row) * D + k] = src[idx];
            }
            const int totb = ntile * m_max;
            const double* srcb = bg + tile * m_max;
            for (int idx = threadIdx.x; idx < totb; idx += BLOCK) {
                const int p = idx / m_max, row = idx - p * m_max;
                sb[p * gs + row] = srcb[idx];
            }
        }
        __syncthreads();
        const long long pg = tile + gib;
        const bool valid = gib < ntile && mine;
        const int m = valid ? (mrows ? mrows[pg] : m_max) : 0;
        const bool has_row = valid && i < m && m <= gs;
        // ---------------------------------------------------------------- my row
        double a[D];
        bool finite = true;
        double nrm2 = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            a[k] = has_row ? myA[i * D + k] : 0.0;
            nrm2 = nrm2 + a[k] * a[k];
            finite = finite && isfinite(a[k]);
        }
        const double bi = has_row ? myb[i] : 0.0;
        finite = finite && isfinite(bi);
        const double nrm = sqrt(nrm2);
        const double an_i = 1.0 / nrm;
        // ---------------------------------------------------------------- F1: Chebyshev ball
        double xc[D];
        double rr = 0.0;
        bool ball, fulldim;
        {
            Simplex<D + 1, false, true> S;
            S.reset(D + 1, m, i);
#pragma unroll
            for (int k = 0; k < D; ++k) S.T[k] = a[k];
            const bool zero = !(nrm > 0.0);
            S.T[D] = nrm;
            S.beta = bi;
            S.rowact = has_row && !zero;
            if (!S.rowact) { S.beta = 0.0; S.T[D] = 0.0; }
            const bool infeasible0 = grp_ballot(has_row && zero && bi < -TOL_FEAS, g) != 0;
            const bool bad = grp_ballot(!finite, g) != 0 || m > gs;
            S.cost[D] = -1.0;
            S.mode = M_INIT;
            S.init_col = D;
            S.init_q = bi / nrm;
            S.init_elig = S.rowact;
            S.mode_after_init = M_P2;
            if (!valid || bad) { S.mode = M_DONE; S.status = ST_NUM; }
            else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
            S.run(g);
            const bool ok = S.status == ST_OPT;
            const double mine = S.x_value();
            const bool holds = S.holds_x();
#pragma unroll
            for (int j = 0; j <= D; ++j) {
                const uint64_t ob = grp_ballot(holds && S.rowvar == j, g);
                const double v = bcast(mine, g.gbase + (ob ? __ffsll((long long)ob) - 1 : 0));
                const double xj = ob ? v : 0.0;
                if (j < D) xc[j < D ? j : 0] = xj; else rr = xj;
            }
            ball = ok && rr >= 0.0;        // cheby_ball: status 0 and r >= 0 (:1289-1293)
            fulldim = ball && rr > abs_tol;
        }
        // ---------------------------------------------------------------- dedupe (:1094-1110)
        // unit rows with dot > 1 - abs_tol are the same hyperplane; of a pair (p<q) the one with the
        // larger normalised offset goes, ties drop p.
        uint64_t live;
        {
            bool removed = false;
            double ni[D];
#pragma unroll
            for (int k = 0; k < D; ++k) ni[k] = a[k] * an_i;
            const double bin_ = bi * an_i;
            for (int j = 0; j < m_max; ++j) {
                const bool jrow = valid && j < m;
                const double an_j = bcast(an_i, g.gbase + (j & (gs - 1)));
                double dot = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) dot = dot + ni[k] * (myA[j * D + k] * an_j);
                const double bjn = myb[j] * an_j;
                const bool par = has_row && jrow && j != i && (dot > 1.0 - abs_tol);
                removed = removed || (par && ((i < j) ? !(bin_ < bjn) : (bjn < bin_)));
            }
            live = grp_ballot(has_row && !removed, g);
        }
        int flags = fulldim ? 0 : RF_EMPTY;
        int nlp = 1;
        uint64_t keep = 0ull;
        int stage = 0;  // 0 done, 1 needs the box, 2 needs the redundancy LPs
        if (fulldim) {
            const int neq = __popcll(live);
            if (neq <= D + 1) { flags = RF_EARLY; keep = live; }
            else stage = (neq > 3 * D) ? 1 : 2;
        }
        // dictionary translated to the Chebyshev centre: beta_i = b_i - a_i.xc
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(a[k], ball ? xc[k] : 0.0, s);
        // ---------------------------------------------------------------- F3: bounding box (:1367-1409)
        if (__any(stage == 1)) {
            double s1 = 0.0, s2 = 0.0;
            bool lpfail = false;
            const bool lrow = (live >> i) & 1ull;
            const double bsh = bi - s;
            const bool go = stage == 1;
            double lbk = 0.0;
            for (int it = 0; it < 2 * D; ++it) {  // lower_0, upper_0, lower_1, upper_1, ...
                const int k = it >> 1;
                const bool up = it & 1;
                double aik = 0.0, xck = 0.0;
                Simplex<D, false, false> S;
                S.reset(D, __popcll(live), i);
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    aik = (kk == k) ? a[kk] : aik;
                    xck = (kk == k) ? xc[kk] : xck;
                    S.T[kk] = lrow ? a[kk] : 0.0;
                    S.cost[kk] = (kk == k) ? (up ? -1.0 : 1.0) : 0.0;
                }
                S.beta = (lrow && bsh > 0.0) ? bsh : 0.0;
                S.rowact = lrow;
                S.mode = go ? M_P2 : M_DONE;
                S.run(g);
                // zeta = c.x' = -negz ; x_k = xc_k + x'_k ; lower: c = +e_k, upper: c = -e_k
                double val;
                if (S.status == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
                else if (S.status == ST_UNBND) val = up ? pinf : -pinf;
                else { val = qnan; lpfail = lpfail || go; }
                if (!up) {
                    lbk = val;
                } else {  // prefilter sums, accumulated in k order (:1131-1134)
                    const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                    s1 = s1 + pa * (val - lbk);
                    s2 = s2 + aik * lbk;
                }
            }
            const bool out = (s1 - (bi - s2)) < -1e-4;
            const uint64_t outb = grp_ballot(go && lrow && out, g);
            if (go) {
                live = live & ~outb;
                nlp += 2 * D;
                if (lpfail) flags |= RF_LPFAIL;
                if (__popcll(live) <= D + 1) { flags |= RF_EARLY; keep = live; stage = 0; }
                else stage = 2;
            }
        }
        // ---------------------------------------------------------------- F2: redundancy LPs (:1142-1160)
        if (__any(stage == 2)) {
            const bool lrow = (live >> i) & 1ull;
            // h[k] += 0.1 for LP k; rows k' < k carry the (+0.1, -0.1) round trip (:1149-1151)
            const double bup = bi + 0.1;
            const double brt = bup - 0.1;
            const double sh_plain = bi - s, sh_up = bup - s, sh_rt = brt - s;
            uint64_t todo = (stage == 2) ? live : 0ull;
            if (stage == 2) nlp += __popcll(live);
            while (__any(todo != 0ull)) {
                const bool go = todo != 0ull;
                const int k = go ? __ffsll((long long)todo) - 1 : 0;
                todo &= todo - 1ull;
                Simplex<D, false, false> S;
                S.reset(D, __popcll(live), i);
                double cxc = 0.0;
#pragma unroll
                for (int kk = 0; kk < D; ++kk) {
                    const double ck = -myA[k * D + kk];  // f = -A[k,:]  (:1145)
                    S.T[kk] = lrow ? a[kk] : 0.0;
                    S.cost[kk] = ck;
                    cxc = fma(ck, xc[kk], cxc);
                }
                const double bsh = (i < k) ? sh_rt : ((i == k) ? sh_up : sh_plain);
                S.beta = (lrow && bsh > 0.0) ? bsh : 0.0;
                S.rowact = lrow;
                S.mode = go ? M_P2 : M_DONE;
                S.run(g);
                const double fun = cxc - S.negz;        // c.xc + zeta, zeta = -negz
                const double bk = myb[k];
                const double hk = (bk + 0.1) - 0.1;
                const double obj = -fun - hk;           // (:1156)
                const bool keepk = go && ((S.status == ST_OPT && obj > abs_tol) || S.status == ST_UNBND);
                keep |= keepk ? (1ull << k) : 0ull;
            }
            if (stage == 2) flags |= RF_MINREP;
        }
        // ---------------------------------------------------------------- results
        if (valid && i == 0) {
            keep_out[pg] = keep;
            flags_out[pg] = flags;
            nlp_out[pg] = nlp;
            r_out[pg] = ball ? rr : 0.0;
        }
        if (valid) {
#pragma unroll
            for (int k = 0; k < D; ++k)
                if (i == (k & (gs - 1)) ) xc_out[pg * D + k] = ball ? xc[k] : qnan;
        }
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
