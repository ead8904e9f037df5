// plp_reduce_general.hpp -- one tile of the general fused reduce (one dictionary row per lane, Bland's rule inside the
// simplex: plp_simplex.hpp): the body of reduce_kernel<D> (plp_reduce.hip), as a device function so that the fast kernels
// can redo the polytopes they hand back (RF_RETRY) in place instead of leaving them to a second launch.
// NT = threads per workgroup; a tile = NT / gs polytopes (gs lanes per polytope, gs >= rows); smem_raw: at least
// (NT / gs) * gs * (D + 1) doubles.  `mine`: my lane group's polytope takes part (others are left untouched).
#pragma once
#include "plp_kernels.hpp"
#include "plp_simplex.hpp"

namespace plp {

template <int D, int NT>
__device__ __forceinline__ void reduce_general_tile(unsigned char* smem_raw, const long long tile, const int ntile, const bool mine,
                                                    int m_max, int gs, const double* __restrict__ Ag,
                                                    const double* __restrict__ bg, const int* __restrict__ mrows,
                                                    double abs_tol, unsigned long long* __restrict__ keep_out,
                                                    int* __restrict__ flags_out, double* __restrict__ r_out,
                                                    double* __restrict__ xc_out, int* __restrict__ nlp_out) {
    const Grp g(gs);
    const int NG = NT / gs;
    const int gib = threadIdx.x / gs;
    const int i = g.gl;
    double* sA = reinterpret_cast<double*>(smem_raw);   // [NG][gs][D]
    double* sb = sA + (size_t)NG * gs * D;               // [NG][gs]
    const double* myA = sA + (size_t)gib * gs * D;       // rows of my polytope
    const double* myb = sb + (size_t)gib * gs;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    // ---------------------------------------------------------------- stage rows in LDS
    __syncthreads();  // the previous tile's readers are done
    {
        const int rowsz = m_max * D;
        const int totA = ntile * rowsz;
        const double* src = Ag + tile * rowsz;
        for (int idx = threadIdx.x; idx < totA; idx += NT) {
            const int p = idx / rowsz, rem = idx - p * rowsz;
            const int row = rem / D, k = rem - row * D;
            sA[((size_t)p * gs + row) * D + k] = src[idx];
        }
        const int totb = ntile * m_max;
        const double* srcb = bg + tile * m_max;
        for (int idx = threadIdx.x; idx < totb; idx += NT) {
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
    bool ball, fulldim, f1open = false;
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
        f1open = valid && !ok && S.status != ST_INFEAS;   // (RF_F1OPEN, plp_common.hpp)
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
        {   // a centre that violates a row (centre_off, plp_common.hpp) is no centre: RF_F1OPEN
            double sk = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) sk = fma(a[k], xc[k], sk);
            if (ball && grp_ballot(has_row && centre_off(bi - sk, an_i, bi, centre_scale<D>(xc)), g) != 0) {
                ball = false; f1open = true;
            }
        }
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
    int flags = fulldim ? 0 : (RF_EMPTY | (f1open ? RF_F1OPEN : 0));
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

}  // namespace plp
