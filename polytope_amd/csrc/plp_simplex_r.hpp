// plp_simplex_r.hpp -- dense simplex with R dictionary rows per lane (gfx950).
//
// plp_simplex.hpp gives every row of the dictionary its own lane: a 16-row LP occupies a 16-lane
// group and one wavefront instruction advances only 4 LPs; measured on MI355X the fused reduce
// kernel built on it is VALU-issue bound (SQ_ACTIVE_INST_VALU ~96 %).  Here lane l of a group owns
// the R consecutive rows l*R .. l*R+R-1 (T[R][NC], beta[R] in VGPRs), the group is GS = rows/R
// lanes, and a wavefront carries 64/GS LPs: 16 for 16-row polytopes at R = 4.  Per pivot
//   * the entering-column scan runs on the replicated reduced costs (no cross-lane traffic),
//   * every lane finds the best of its own R rows by cross-multiplication (no division), takes one
//     reciprocal (v_rcp_f64 + 2 Newton steps) and the group minimum is an exact f64 min done as two
//     u32 min all-reduces on an order-preserving key -- for GS = 4 these are two DPP quad_perm steps,
//   * the pivot row travels by NC+2 ds_bpermute broadcasts, and each lane updates its R rows.
// Pivot rules are those of plp_simplex.hpp / oracle/plp_oracle.c: free variables enter in either
// direction and never leave, Dantzig pricing, ties of the ratio test go to the lowest row; after
// BLAND_AFTER consecutive degenerate pivots Bland's rule (lowest variable id) takes over behind a
// wave-uniform branch.  No phase 1 here: the callers start from primal-feasible dictionaries
// (forced first pivot for the Chebyshev LP, translation to the Chebyshev centre for F2/F3).
#pragma once
#include "plp_simplex.hpp"

namespace plp {

// TRACKX = false drops the sign bookkeeping of the free variables (rneg / cneg): callers that only
// read the optimal VALUE off the dictionary (F2, F3) never ask for x.
// CARRY = true (fast path only): a second cost row (cost2 / negz2) is carried through the pivots and columns
// can be marked dead -- phase 1 of the generic LP, which minimises the artificial variable while the real
// objective rides along (plp_simplex.hpp does the same with one row per lane).
constexpr int ID_TR = -1;  // id of the phase-1 artificial variable

// max(a, b) as ONE v_max_f64.  fmax() makes the compiler canonicalise operands it cannot prove to be the result
// of an arithmetic instruction (values that went through selects or bit operations) with an extra v_max_f64 x, x;
// the operands here are never signalling NaNs, and v_max_f64 itself returns the non-NaN operand like fmax().
#ifndef PLP_FOLD_FIXUP
#define PLP_FOLD_FIXUP 1  // entering column zeroed before the update instead of rewritten after it (0: A/B)
#endif
#ifndef PLP_RAW_MAX
#define PLP_RAW_MAX 1
#endif
__device__ __forceinline__ double max_raw(double a, double b) {
#if PLP_RAW_MAX
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(a, b);
#endif
}
__device__ __forceinline__ double max0_raw(double a) {  // max(a, 0) with the inline constant
#if PLP_RAW_MAX
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a));
    return r;
#else
    return fmax(a, 0.0);
#endif
}

template <int NC, int R, bool INITM, bool TRACKX = true, bool CARRY = false>
struct SimplexR {
    // ---- my R rows
    double T[R][NC];
    double beta[R];
    int rv[R];        // id of the basic variable of my row k (0..n-1 structural, n+i slack)
    unsigned rneg;    // bit k: the basic free variable of row k is stored negated
    unsigned ract;    // bit k: row k takes part in ratio tests
    // ---- replicated per group
    double cost[NC], negz;
    double cost2[CARRY ? NC : 1], negz2;  // carried objective (CARRY)
    unsigned dead;    // bit j: column j never enters (CARRY)
    int cv[NC];       // id of the nonbasic variable of column j
    unsigned cneg;    // bit j: column j holds -x
    unsigned cfree;   // bit j: column j holds a free (structural) variable
    int n, ndeg, iters, maxit;
    int mode, status;
    // ---- INIT pivot request: forced entering column, caller-supplied signed ratios of my rows
    int init_col;
    double init_q[INITM ? R : 1];
    unsigned init_elig;
    int mode_after_init;

    __device__ __forceinline__ void reset(int n_, int m_rows, int first_row) {
        n = n_;
        rneg = 0u;
        ract = 0u;
        negz = 0.0;
        negz2 = 0.0;
        dead = 0u;
#pragma unroll
        for (int j = 0; j < (CARRY ? NC : 1); ++j) cost2[j] = 0.0;
        cneg = 0u;
        cfree = n_ >= 32 ? 0xffffffffu : ((1u << n_) - 1u);
        ndeg = 0;
        iters = 0;
        maxit = 50 * (m_rows + n_) + 100;
        status = -1;
        mode = M_P2;
        init_col = -1;
        init_elig = 0u;
        mode_after_init = M_P2;
#pragma unroll
        for (int j = 0; j < NC; ++j) { cv[j] = j; cost[j] = 0.0; }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            rv[k] = n_ + first_row + k;
            beta[k] = 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) T[k][j] = 0.0;
        }
    }

    __device__ __forceinline__ void step(const Grp& g) {
        const bool running = mode != M_DONE;
        const bool bland = ndeg >= BLAND_AFTER;
        // ------------------------------------------------ entering column (Dantzig)
        int e = -1;
        double best = 0.0;
        bool epos = false;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const double c = cost[j];
            const double ac = fabs(c);
            const bool elig = (ac > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0));
            const bool take = elig & (ac > best);
            e = take ? j : e;
            best = take ? ac : best;
            epos = take ? (c > 0.0) : epos;
        }
        if (__any(bland & running)) {  // Bland: lowest variable id among the eligible columns
            int eb = -1, bid = 0x7fffffff;
            double bb = 0.0;
            bool bp = false;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double c = cost[j];
                const double ac = fabs(c);
                const bool elig = (ac > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0));
                const bool take = elig & (cv[j] < bid);
                eb = take ? j : eb;
                bid = take ? cv[j] : bid;
                bb = take ? ac : bb;
                bp = take ? (c > 0.0) : bp;
            }
            if (bland) { e = eb; best = bb; epos = bp; }
        }
        int fin = -1;
        bool normal = running & (mode == M_P2);
        fin = (normal & (e < 0)) ? ST_OPT : fin;
        normal = normal & (e >= 0);
        fin = (normal & (iters >= maxit)) ? ST_ITER : fin;
        normal = normal & (iters < maxit);
        bool init = false;
        if constexpr (INITM) init = running & (mode == M_INIT);
        if (init) e = init_col;
        bool act = normal | init;
        e = act ? e : -1;
        if (__any(act)) {  // wave-uniform: a step that only detects optimality skips the pivot
            // ------------------------------------------------ entering column of my rows
            const bool flip = normal & epos;  // free variable entering downwards: x := -x
            double a[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < NC; ++j) v = (j == e) ? T[k][j] : v;
                a[k] = flip ? -v : v;
            }
            double ce = -best;  // normal mode: the (sign-flipped) reduced cost of the entering column
            if constexpr (INITM) {
                if (__any(init)) {
                    double cr = 0.0;
#pragma unroll
                    for (int j = 0; j < NC; ++j) cr = (j == e) ? cost[j] : cr;
                    if (init) ce = cr;
                }
            }
            const unsigned efree = (cfree >> (e & 31)) & 1u;
            const unsigned eneg = ((cneg >> (e & 31)) & 1u) ^ (flip ? 1u : 0u);
            int vin = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) vin = (j == e) ? cv[j] : vin;
            // ------------------------------------------------ ratio test, my rows first
            // b_k / a_k < b_n / a_n  <=>  b_k a_n < b_n a_k  (a > 0): the first (lowest) row wins ties
            int kb = -1;
            double bn = 0.0, an = 1.0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const bool elig = normal & (((ract >> k) & 1u) != 0u) & (a[k] > TOL_PIV);
                const double bi = beta[k] > 0.0 ? beta[k] : 0.0;
                const bool better = elig & ((kb < 0) | (bi * an < bn * a[k]));
                kb = better ? k : kb;
                bn = better ? bi : bn;
                an = better ? a[k] : an;
            }
            double qinit = 0.0;
            if constexpr (INITM) {
                if (__any(init)) {
                    int ki = -1;
                    double qi = 0.0, ai = 1.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const bool elig = init & (((init_elig >> k) & 1u) != 0u);
                        const bool better = elig & ((ki < 0) | (init_q[k] < qi));
                        ki = better ? k : ki;
                        qi = better ? init_q[k] : qi;
                        ai = better ? a[k] : ai;
                    }
                    if (init) { kb = ki; an = ai; qinit = qi; }
                }
            }
            const double x0 = __builtin_amdgcn_rcp(an);
            const double x1 = fma(x0, fma(-an, x0, 1.0), x0);
            double pinv = fma(x1, fma(-an, x1, 1.0), x1);  // 1/a of my best row = the pivot's 1/a_re
            double q = bn * pinv;
            if constexpr (INITM) q = init ? qinit : q;
            const bool erow = kb >= 0;
            q = erow ? q : __longlong_as_double(0x7ff0000000000000ll);
            // ------------------------------------------------ group minimum (exact, on a u64 key)
            const int qh = __double2hiint(q), ql = __double2loint(q);
            const int sm = qh >> 31;
            const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
            const unsigned kl = (unsigned)(ql ^ sm);
            const unsigned mh = grp_min(kh, g.gs);
            const unsigned klm = (kh == mh) ? kl : 0xffffffffu;
            const unsigned ml = grp_min(klm, g.gs);
            if (act & (mh >= 0xfff00000u)) {  // +inf: no eligible row (or NaN)
                fin = ((mh == 0xfff00000u) & (ml == 0u)) ? ST_UNBND : ST_NUM;
                act = false;
                e = -1;
            }
            const bool tie = erow & (kh == mh) & (kl == ml);
            const uint64_t tbal = grp_ballot(tie, g);
            int rl = tbal ? __ffsll((long long)tbal) - 1 : 0;  // lowest lane = lowest row among ties
            if (__any(bland & act & !init)) {
                // Bland mode: among ALL rows attaining the minimum, the lowest basic-variable id.
                // Needs every row's own ratio: one reciprocal per row, rare path.
                const double qmin = __hiloint2double((int)(mh ^ 0x80000000u), (int)ml);
                unsigned idb = 0xffffffffu;
                int kbl = -1;
                double pinvb = 1.0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const bool elig = normal & (((ract >> k) & 1u) != 0u) & (a[k] > TOL_PIV);
                    const double y0 = __builtin_amdgcn_rcp(a[k]);
                    const double y1 = fma(y0, fma(-a[k], y0, 1.0), y0);
                    const double y2 = fma(y1, fma(-a[k], y1, 1.0), y1);
                    const double qk = (beta[k] > 0.0 ? beta[k] : 0.0) * y2;
                    const bool t2 = elig & (qk <= qmin);
                    const bool take = t2 & ((unsigned)rv[k] < idb);
                    idb = take ? (unsigned)rv[k] : idb;
                    kbl = take ? k : kbl;
                    pinvb = take ? y2 : pinvb;
                }
                const unsigned idmin = grp_min(idb, g.gs);
                const uint64_t kbal = grp_ballot((idb == idmin) & (idb != 0xffffffffu), g);
                if (bland & (kbal != 0)) {
                    rl = __ffsll((long long)kbal) - 1;
                    if (g.gl == rl) { kb = kbl; pinv = pinvb; }
                }
            }
            const bool is_r = act & (g.gl == rl);
            const int raddr = (g.gbase + rl) << 2;
            if (normal & act) {
                const double qmin = __hiloint2double((int)(mh ^ 0x80000000u), (int)ml);
                ndeg = (qmin <= DEGEN_EPS) ? ndeg + 1 : 0;
            }
            // ------------------------------------------------ pivot row: latch in its lane, broadcast
            double prow[NC], pb = 0.0;
            int prv = 0;
            unsigned prneg = 0u;
#pragma unroll
            for (int j = 0; j < NC; ++j) prow[j] = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (k == kb) {
#pragma unroll
                    for (int j = 0; j < NC; ++j) prow[j] = T[k][j];
                    pb = beta[k];
                    prv = rv[k];
                    prneg = (rneg >> k) & 1u;
                }
            }
            const double p = act ? bcast_addr(pinv, raddr) : 0.0;
            const double rhob = bcast_addr(pb, raddr) * p;
            double rho[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) rho[j] = bcast_addr(prow[j], raddr) * p;
            const int rpack = __builtin_amdgcn_ds_bpermute(raddr, (prv << 1) | (int)prneg);
            const int krow = __builtin_amdgcn_ds_bpermute(raddr, kb);
            // ------------------------------------------------ update my rows
            const double fc = act ? ce : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double c = fma(-fc, rho[j], cost[j]);
                cost[j] = (j == e) ? -(fc * p) : c;
            }
            negz = fma(-fc, rhob, negz);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const bool piv = is_r & (k == krow);
                const double f = (act & !piv) ? a[k] : 0.0;
#pragma unroll
                for (int j = 0; j < NC; ++j) T[k][j] = fma(-f, rho[j], T[k][j]);
                beta[k] = fma(-f, rhob, beta[k]);
                a[k] = f;
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (j == e) {  // entering column: T[k][e] = -a_k p
#pragma unroll
                    for (int k = 0; k < R; ++k) T[k][j] = -(a[k] * p);
                }
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) rho[j] = (j == e) ? p : rho[j];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (is_r & (k == krow)) {  // the pivot row itself
#pragma unroll
                    for (int j = 0; j < NC; ++j) T[k][j] = rho[j];
                    beta[k] = rhob;
                    rv[k] = vin;
                    if constexpr (TRACKX) rneg = (rneg & ~(1u << k)) | (eneg << k);
                    if (efree) ract &= ~(1u << k);  // a free variable never leaves again
                }
            }
            // ------------------------------------------------ bookkeeping of the column
#pragma unroll
            for (int j = 0; j < NC; ++j) cv[j] = (j == e) ? (rpack >> 1) : cv[j];
            if (act) {
                if constexpr (TRACKX) cneg = (cneg & ~(1u << e)) | ((unsigned)(rpack & 1) << e);
                cfree &= ~(1u << e);
                iters += 1;
            }
            // Optimal right after this pivot?  Checking the fresh cost row here saves the whole extra
            // lock-step iteration that would otherwise only discover "no entering column".
            bool more = false;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double c = cost[j];
                more = more | ((fabs(c) > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0)));
            }
            if (normal & act & !more) fin = ST_OPT;
        }
        // ------------------------------------------------ mode transitions
        if (INITM && init) {
            mode = (fin >= 0) ? M_DONE : mode_after_init;
            if (fin >= 0) status = fin;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (((ract >> k) & 1u) && beta[k] < 0.0) beta[k] = 0.0;  // rounding of the forced pivot
        } else if (running & (fin >= 0)) {
            mode = M_DONE;
            status = fin;
        }
    }

    __device__ __forceinline__ void run(const Grp& g) {
        while (__any(mode != M_DONE)) step(g);
    }

    // ------------------------------------------------------------------------------------------
    // Fast path (no forced pivot, Dantzig mode): the same pivot rules as step(), written so that a
    // pivot costs ~half the VALU instructions:
    //   * the entering column for the NEXT pivot is chosen right after the update (that scan doubles
    //     as the optimality test), on an order-preserving key: for a free column |c|, otherwise -c, so
    //     that "eligible and larger than the best so far" is one compare and one v_max_f64;
    //   * the pivot row is latched inside the ratio-test scan (one exec-masked block of moves per row)
    //     instead of in a second pass; the lowest tied lane comes from a DPP min, not a ballot;
    //   * column-e / row-r fix-ups are exec-masked 64-bit moves, not pairs of selects.
    // An LP that reaches Bland mode (>= BLAND_AFTER consecutive degenerate pivots) ends with status
    // ST_RETRY and is redone by step(); both walk the same vertex path (checked against the oracle).
    template <int GS>
    static __device__ __forceinline__ unsigned gmin(unsigned v) {
        PLP_MIN_U32_DPP(v, "quad_perm:[1,0,3,2]");
        if constexpr (GS > 2) PLP_MIN_U32_DPP(v, "quad_perm:[2,3,0,1]");
        if constexpr (GS > 4) PLP_MIN_U32_DPP(v, "row_half_mirror");
        if constexpr (GS > 8) PLP_MIN_U32_DPP(v, "row_mirror");
        if constexpr (GS > 16) v = min_u(v, (unsigned)__shfl_xor((int)v, 16, 64));
        if constexpr (GS > 32) v = min_u(v, (unsigned)__shfl_xor((int)v, 32, 64));
        return v;
    }

    // Dantzig choice on the current cost row: e = -1 when the dictionary is optimal; chi = high word
    // of cost[e] (its sign tells the direction a free variable enters)
    __device__ __forceinline__ void scan_enter(int& e, double& best, int& chi) const {
        e = -1;
        best = TOL_D;
        chi = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int hi = __double2hiint(cost[j]);
            // free column: clear the sign (|c|); bounded column: flip it (-c)
            const int khi = (hi ^ (int)0x80000000) & ~((int)(cfree << (31 - j)) & (int)0x80000000);
            const double key = __hiloint2double(khi, __double2loint(cost[j]));
            bool take = key > best;
            if constexpr (CARRY) take = take & (((dead >> j) & 1u) == 0u);
            e = take ? j : e;
            chi = take ? hi : chi;
            if constexpr (CARRY) best = take ? key : best;  // a dead column must not raise the bar
            else best = max_raw(best, key);
        }
    }

    // One pivot.  KIND 0: Dantzig column (e, best, chi) from the last scan, ratio test on my rows.
    // KIND 1: a forced pivot (plp_simplex.hpp, M_INIT): the LAST column enters, the leaving row is the eligible
    // one (bits of ielig) with the smallest caller-supplied signed ratio qinit[k]; basic values rounded below
    // zero by this pivot are clamped.  KIND 2: the same with the entering column e given by the caller per
    // group (M_DRIVE: the artificial variable is pivoted out of the basis after phase 1); neither forced
    // kind re-chooses the entering column afterwards when KIND is 2.
    template <int GS, int KIND>
    __device__ __forceinline__ void pivot_core(const Grp& g, int& e, double& best, int& chi, const double* qinit,
                                               unsigned ielig) {
        constexpr bool FORCED = KIND != 0;
        const bool running = mode != M_DONE;
        bool act = running;
        if constexpr (KIND == 2) act = running & (e >= 0);
        if constexpr (!FORCED) {
            const bool over = running & (iters >= maxit);
            if (over) { status = ST_ITER; mode = M_DONE; }
            act = running & !over;  // e >= 0 here: an optimal dictionary was retired by the last scan
        }
        // ------------------------------------------------ entering column of my rows
        double a[R];
        int vin;
        unsigned efree, eneg = 0u;
        double c2e = 0.0, cfe = 0.0;  // cost2[e] (CARRY), cost[e] (forced kinds)
        if constexpr (KIND == 1) {
            e = NC - 1;
            cfe = cost[NC - 1];
            if constexpr (CARRY) c2e = cost2[NC - 1];
            vin = cv[NC - 1];
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = T[k][NC - 1];
            efree = (cfree >> (NC - 1)) & 1u;
            if constexpr (TRACKX) eneg = (cneg >> (NC - 1)) & 1u;
        } else {
            vin = cv[0];
            if constexpr (CARRY) c2e = cost2[0];
            if constexpr (KIND == 2) cfe = cost[0];
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = T[k][0];
#pragma unroll
            for (int j = 1; j < NC; ++j) {
                if (e == j) {
#pragma unroll
                    for (int k = 0; k < R; ++k) a[k] = T[k][j];
                    vin = cv[j];
                    if constexpr (CARRY) c2e = cost2[j];
                    if constexpr (KIND == 2) cfe = cost[j];
                }
            }
            // c > 0: the free variable enters downwards, x := -x  (never for a forced pivot)
            const int sgn = (KIND == 0 && chi >= 0) ? (int)0x80000000 : 0;
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = __hiloint2double(__double2hiint(a[k]) ^ sgn, __double2loint(a[k]));
            if constexpr (CARRY) c2e = __hiloint2double(__double2hiint(c2e) ^ sgn, __double2loint(c2e));
            efree = (cfree >> (e & 31)) & 1u;
            if constexpr (TRACKX) eneg = ((cneg >> (e & 31)) & 1u) ^ ((unsigned)sgn >> 31);
        }
        // ------------------------------------------------ ratio test over my rows, pivot row latched on the way
        int kb = 0, prv = 0;
        double bn = __longlong_as_double(0x7ff0000000000000ll), an = 1.0, pb = 0.0;
        double prow[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) prow[j] = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            bool better;
            double bi;
            if constexpr (FORCED) {
                bi = qinit[k];
                better = act & (((ielig >> k) & 1u) != 0u) & (bi < bn);
            } else {
                const bool elig = act & (((ract >> k) & 1u) != 0u) & (a[k] > TOL_PIV);
                bi = max0_raw(beta[k]);
                better = elig & (bi * an < bn * a[k]);
            }
            if (better) {  // strict: the first (lowest) row keeps a tie
                kb = k;
                bn = bi;
                an = a[k];
                pb = beta[k];
                prv = TRACKX ? ((rv[k] << 1) | (int)((rneg >> k) & 1u)) : rv[k];
#pragma unroll
                for (int j = 0; j < NC; ++j) prow[j] = T[k][j];
            }
        }
        const double x0 = __builtin_amdgcn_rcp(an);
        const double x1 = fma(x0, fma(-an, x0, 1.0), x0);
        const double pinv = fma(x1, fma(-an, x1, 1.0), x1);
        const double q = FORCED ? bn : bn * pinv;  // +inf when no row of mine is eligible
        // ------------------------------------------------ group minimum (exact, on a u64 key)
        const int qh = __double2hiint(q), ql = __double2loint(q);
        const int sm = qh >> 31;
        const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
        const unsigned kl = (unsigned)(ql ^ sm);
        const unsigned mh = gmin<GS>(kh);
        const unsigned ml = gmin<GS>((kh == mh) ? kl : 0xffffffffu);
        if (act & (mh >= 0xfff00000u)) {  // +inf: unbounded (or NaN: numerical trouble)
            status = ((mh == 0xfff00000u) & (ml == 0u)) ? ST_UNBND : ST_NUM;
            mode = M_DONE;
            act = false;
        }
        const unsigned rl = gmin<GS>(((kh == mh) & (kl == ml)) ? (unsigned)g.gl : 0xffu);  // lowest tied lane
        const bool is_r = act & ((unsigned)g.gl == rl);
        const int raddr = (g.gbase + (int)rl) << 2;
        if constexpr (!FORCED) {
            const double qmin = __hiloint2double((int)(mh ^ 0x80000000u), (int)ml);  // q >= 0 here
            const int nd = (qmin <= DEGEN_EPS) ? ndeg + 1 : 0;
            ndeg = act ? nd : ndeg;
        }
        // ------------------------------------------------ broadcast the pivot row
        const double p = bcast_addr(pinv, raddr);
        const double rhob = bcast_addr(pb, raddr) * p;
        double rho[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) rho[j] = bcast_addr(prow[j], raddr) * p;
        const int vout = __builtin_amdgcn_ds_bpermute(raddr, prv);
        const int krow = __builtin_amdgcn_ds_bpermute(raddr, kb);
        // ------------------------------------------------ update
        // the (sign-normalised) reduced cost of the entering column
#if PLP_FOLD_FIXUP
        const double fc = act ? (FORCED ? cfe : -best) : 0.0;
        const double fc2 = (CARRY && act) ? c2e : 0.0;
        // The entering column will hold the leaving variable: -a_k p in my rows, -c_e p in the cost row.  Zeroed here, with
        // p in its place in the pivot row, the updates below produce exactly that (fma(-f, p, 0) = -(f p), one rounding
        // either way) -- the oracle's pivot() does the same -- and the column needs no pass of its own afterwards.
#pragma unroll
        for (int j = (KIND == 1) ? NC - 1 : 0; j < NC; ++j) {
            if (act & (e == j)) {
#pragma unroll
                for (int k = 0; k < R; ++k) T[k][j] = 0.0;
                cost[j] = 0.0;
                if constexpr (CARRY) cost2[j] = 0.0;
                rho[j] = p;
                cv[j] = TRACKX ? (vout >> 1) : vout;
                if constexpr (TRACKX) cneg = (cneg & ~(1u << j)) | ((unsigned)(vout & 1) << j);
            }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) cost[j] = fma(-fc, rho[j], cost[j]);
        negz = fma(-fc, rhob, negz);
        if constexpr (CARRY) {
#pragma unroll
            for (int j = 0; j < NC; ++j) cost2[j] = fma(-fc2, rho[j], cost2[j]);
            negz2 = fma(-fc2, rhob, negz2);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool other = act & !(is_r & (k == krow));
            const double f = other ? a[k] : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) T[k][j] = fma(-f, rho[j], T[k][j]);
            beta[k] = fma(-f, rhob, beta[k]);
        }
#else
        const double fc = act ? (FORCED ? cfe : -best) : 0.0;
#pragma unroll
        for (int j = 0; j < NC; ++j) cost[j] = fma(-fc, rho[j], cost[j]);
        negz = fma(-fc, rhob, negz);
        const double fc2 = (CARRY && act) ? c2e : 0.0;
        if constexpr (CARRY) {
#pragma unroll
            for (int j = 0; j < NC; ++j) cost2[j] = fma(-fc2, rho[j], cost2[j]);
            negz2 = fma(-fc2, rhob, negz2);
        }
        double ea[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool other = act & !(is_r & (k == krow));
            const double f = other ? a[k] : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) T[k][j] = fma(-f, rho[j], T[k][j]);
            beta[k] = fma(-f, rhob, beta[k]);
            ea[k] = -(f * p);
        }
        const double ec = -(fc * p);
        const double ec2 = -(fc2 * p);
#pragma unroll
        for (int j = (KIND == 1) ? NC - 1 : 0; j < NC; ++j) {
            if (act & (e == j)) {  // the entering column now holds the leaving variable
#pragma unroll
                for (int k = 0; k < R; ++k) T[k][j] = ea[k];
                cost[j] = ec;
                if constexpr (CARRY) cost2[j] = ec2;
                rho[j] = p;
                cv[j] = TRACKX ? (vout >> 1) : vout;
                if constexpr (TRACKX) cneg = (cneg & ~(1u << j)) | ((unsigned)(vout & 1) << j);
            }
        }
#endif
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (is_r & (k == krow)) {  // the pivot row itself
#pragma unroll
                for (int j = 0; j < NC; ++j) T[k][j] = rho[j];
                beta[k] = rhob;
                rv[k] = vin;
                if constexpr (TRACKX) rneg = (rneg & ~(1u << k)) | (eneg << k);
                ract &= ~(efree << k);  // a free variable never leaves again
            }
        }
        if (act) {
            cfree &= ~(1u << e);
            iters += 1;
        }
        if constexpr (FORCED) {
#pragma unroll
            for (int k = 0; k < R; ++k) {  // rounding of the forced pivot
                const bool neg = running & (((ract >> k) & 1u) != 0u) & (beta[k] < 0.0);
                beta[k] = neg ? 0.0 : beta[k];
            }
        }
        // ------------------------------------------------ next entering column = optimality test
        if constexpr (KIND != 2) {  // (after KIND 2 the caller swaps the cost rows first)
            scan_enter(e, best, chi);
            if (act & (e < 0)) { status = ST_OPT; mode = M_DONE; }
        }
    }

    // Run to completion from a primal-feasible dictionary (mode M_P2, or M_DONE for idle lanes).
    // FORCED: start with the forced pivot (qinit[R] signed ratios of my rows, ielig their eligibility).
    template <int GS, bool FORCED = false>
    __device__ __forceinline__ void run_fast(const Grp& g, const double* qinit = nullptr, unsigned ielig = 0u) {
        static_assert(!INITM, "fast path: the forced pivot is an argument here, not a mode");
        int e, chi;
        double best;
        if constexpr (FORCED) {
            pivot_core<GS, 1>(g, e, best, chi, qinit, ielig);
        } else {
            scan_enter(e, best, chi);
            if ((mode != M_DONE) & (e < 0)) { status = ST_OPT; mode = M_DONE; }
        }
        while (__any(mode != M_DONE)) {
            // rare: Bland's rule lives in step(), kept out of this loop (and of its register budget);
            // the caller redoes the LP with the general engine
            if ((mode != M_DONE) & (ndeg >= BLAND_AFTER)) { status = ST_RETRY; mode = M_DONE; }
            pivot_core<GS, 0>(g, e, best, chi, nullptr, 0u);
        }
    }

    // Generic LP whose origin is infeasible (CARRY): the caller has put the artificial variable t (id ID_TR,
    // -1 on every active row) in the LAST column, cost = e_t (phase 1: minimise t), cost2 = the real
    // objective.  `on`: my group takes part.  Phase 1 = forced pivot "t enters on the smallest right-hand
    // side" + Dantzig; then t is dropped (driven out of the basis first if it stayed basic at ~0) and the
    // carried row becomes the cost row (plp_simplex.hpp: M_INIT -> M_P1 -> M_DRIVE -> M_P2, same pivots).
    template <int GS>
    __device__ __forceinline__ void run_two_phase(const Grp& g, const double* qinit, unsigned ielig, bool on) {
        static_assert(CARRY && !INITM, "two-phase run: fast path with a carried cost row");
        run_fast<GS, true>(g, qinit, ielig);
        // ---- where phase 1 ended
        const bool p1opt = on & (status == ST_OPT);
        if (on & !p1opt & (status != ST_RETRY)) status = (status == ST_ITER) ? ST_ITER : ST_NUM;  // never unbounded
        bool mine = false;   // one of my rows holds t
        int tk = 0;
        double tb = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool h = rv[k] == ID_TR;
            mine = mine | h;
            tk = h ? k : tk;
            tb = h ? beta[k] : tb;
        }
        const uint64_t tbal = grp_ballot(mine, g);
        const int rt = g.gbase + (tbal ? __ffsll((long long)tbal) - 1 : 0);
        const double tval = bcast(tb, rt);
        const bool basic = tbal != 0;
        const bool infeas = p1opt & basic & (tval > TOL_FEAS);
        bool drive = p1opt & basic & !infeas;
        // t basic at ~0: pivot it out on the largest element of its row (columns still alive)
        int eo = -1;
        double big = TOL_PIV;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (mine & (tk == k)) {
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const double aj = fabs(T[k][j]);
                    const bool tkj = (aj > big) & (((dead >> j) & 1u) == 0u);
                    big = tkj ? aj : big;
                    eo = tkj ? j : eo;
                }
            }
        }
        const int ed = bcast(eo, rt);
        if (drive & (ed < 0)) {  // row "0 = t": redundant, t stays basic in a row that takes no further part
            if (mine) ract &= ~(1u << tk);
            drive = false;
        }
        if (infeas) status = ST_INFEAS;
        const bool cont = p1opt & !infeas;
        if (cont) { mode = M_P2; status = -1; }
        int e = drive ? ed : -1, chi = 0;
        double best = 0.0;
        if (__any(e >= 0)) {
            double qz[R];
            unsigned tbits = 0u;
#pragma unroll
            for (int k = 0; k < R; ++k) { qz[k] = 0.0; tbits |= (rv[k] == ID_TR) ? (1u << k) : 0u; }
            pivot_core<GS, 2>(g, e, best, chi, qz, tbits);
        }
        if (cont) {  // the column that now holds t is dropped; the carried cost row becomes active
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (cv[j] == ID_TR) dead |= (1u << j);
                cost[j] = cost2[j];
            }
            negz = negz2;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if ((((ract >> k) & 1u) != 0u) & (beta[k] < 0.0)) beta[k] = 0.0;
            ndeg = 0;
        }
        // ---- phase 2
        run_fast<GS, false>(g);
    }

    // value of structural variable j if one of my rows holds it (found = true)
    __device__ __forceinline__ double x_of(int j, bool& found) const {
        static_assert(TRACKX, "x_of needs the sign bookkeeping");
        double v = 0.0;
        found = false;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool h = rv[k] == j;
            const double xv = ((rneg >> k) & 1u) ? -beta[k] : beta[k];
            v = h ? xv : v;
            found = found | h;
        }
        return v;
    }
};

}  // namespace plp
