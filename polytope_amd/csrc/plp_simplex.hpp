// plp_simplex.hpp -- register-resident dense simplex, one LP per lane group (gfx950).
//
// Problem (reference: polytope/solvers.py:76-106, 149-158):  min c'x  s.t.  Gx <= h,  x free.
//
// Layout: lane i of the group owns constraint row i of the dictionary
//       basic_i = beta - sum_j T[j] * nb_j                       (T[NC], beta : per lane)
//       (-zeta) = negz - sum_j cost[j] * nb_j                    (cost[NC]    : replicated in
//                                                                  every lane of the group)
// NC is a compile-time constant so T[] / cost[] / cv[] live in VGPRs.  One pivot is
//   * a scan over the NC replicated reduced costs (no cross-lane traffic) -> entering column e,
//   * the ratio test: beta_i / T_ie as beta_i * (1/T_ie) with 1/T_ie from v_rcp_f64 + 2 Newton
//     steps, then an EXACT f64 min over the group done as two u32 min all-reduces on an
//     order-preserving key (hi dword, then lo dword among the hi-minima): v_min_u32 takes DPP
//     operands directly, v_min_f64 does not,
//   * NC+1 ds_bpermute broadcasts of the pivot row and NC+1 v_fma_f64 per lane.
// Free variables (the x_j) enter in either direction and never leave.  Dantzig pricing; after
// BLAND_AFTER consecutive degenerate pivots Bland's rule (lowest variable id for the entering
// column and for ties of the ratio test) takes over -- that path and the two special pivots
// (INIT: forced entering column with a caller-supplied signed ratio; DRIVE: artificial out of
// the basis after phase 1) sit behind wave-uniform branches that are almost never taken.
// The same rules are restated in scalar C in oracle/plp_oracle.c (test infrastructure).
//
// All lanes of a wavefront execute every step (finished groups are masked by predication,
// never by divergent control flow, so cross-lane operations stay well defined).
#pragma once
#include "plp_wave.hpp"

namespace plp {

enum : int { M_INIT = 0, M_P1 = 1, M_DRIVE = 2, M_P2 = 3, M_DONE = 4 };

// CARRY : a second cost row is carried through the pivots (phase 1 of the generic LP)
// INITM : the engine may be started in M_INIT (forced first pivot)
template <int NC, bool CARRY, bool INITM = true>
struct Simplex {
    // ---- per-lane row
    double T[NC];
    double beta;
    int rowvar;   // id of my basic variable (0..n-1 structural, n+i slack, -1 artificial)
    int rowneg;   // 1: my basic free variable is stored negated
    bool rowact;  // my row takes part in ratio tests
    // ---- replicated per group
    double cost[NC], negz;
    double cost2[CARRY ? NC : 1], negz2;
    int cv[NC];       // nonbasic variable of column j, packed (id+1)<<1 | negated
    unsigned cfree;   // bit j: column j holds a free (structural) variable
    unsigned dead;    // bit j: column j never enters
    int n;            // ids < n are free structural variables
    int ndeg, iters, maxit;
    int mode, status;
    // ---- INIT pivot request (forced entering column, signed ratio supplied by the caller)
    int init_col;
    double init_q;
    bool init_elig;
    int mode_after_init;

    __device__ __forceinline__ void reset(int n_, int m_rows, int my_row) {
        n = n_;
        rowvar = n_ + my_row;
        rowneg = 0;
        rowact = true;
        negz = 0.0;
        negz2 = 0.0;
        cfree = n_ >= 32 ? 0xffffffffu : ((1u << n_) - 1u);
        dead = 0u;
        ndeg = 0;
        iters = 0;
        maxit = 50 * (m_rows + n_) + 100;
        status = -1;
        mode = M_P2;
        init_col = -1;
        init_q = 0.0;
        init_elig = false;
        mode_after_init = M_P2;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            cv[j] = (j + 1) << 1;
            cost[j] = 0.0;
            T[j] = 0.0;
            if constexpr (CARRY) cost2[j] = 0.0;
        }
        if constexpr (!CARRY) cost2[0] = 0.0;
        beta = 0.0;
    }

    // column j holds the (non-free) variable `id` from the start (e.g. the phase-1 artificial)
    __device__ __forceinline__ void set_col(int j, int id) {
        cv[j] = (id + 1) << 1;
        cfree &= ~(1u << j);
    }

    __device__ __forceinline__ bool holds_x() const { return (unsigned)rowvar < (unsigned)n; }
    __device__ __forceinline__ double x_value() const { return rowneg ? -beta : beta; }

    // One lockstep iteration for every group of the wavefront.
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
            // bitwise, not short-circuit: '&&' chains compile to s_and_saveexec control flow
            const bool elig = (ac > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0)) & (((dead >> j) & 1u) == 0u);
            const bool take = elig & (ac > best);
            e = take ? j : e;
            best = take ? ac : best;
            epos = take ? (c > 0.0) : epos;
        }
        if (__any(bland && running)) {  // Bland: lowest variable id among the eligible columns
            int eb = -1, bid = 0x7fffffff;
            double bb = 0.0;
            bool bp = false;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double c = cost[j];
                const double ac = fabs(c);
                const bool elig = (ac > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0)) & (((dead >> j) & 1u) == 0u);
                const bool take = elig & (cv[j] < bid);
                eb = take ? j : eb;
                bid = take ? cv[j] : bid;
                bb = take ? ac : bb;
                bp = take ? (c > 0.0) : bp;
            }
            if (bland) { e = eb; best = bb; epos = bp; }
        }
        int fin = -1;  // status this group finishes with in this iteration
        bool normal = running & ((mode == M_P1) | (mode == M_P2));
        fin = (normal & (e < 0)) ? ST_OPT : fin;
        normal = normal & (e >= 0);
        fin = (normal & (iters >= maxit)) ? ST_ITER : fin;
        normal = normal & (iters < maxit);
        // ------------------------------------------------ special pivots (rare)
        bool init = false, drive = false;
        int rt = 0;
        if constexpr (INITM) init = running && mode == M_INIT;
        if constexpr (CARRY) {
            drive = running && mode == M_DRIVE;
            if (__any(drive)) {  // t is basic at ~0 after phase 1: pivot it out on its largest element
                const uint64_t tb = grp_ballot(rowvar == ID_T, g);
                rt = g.gbase + (tb ? __ffsll((long long)tb) - 1 : 0);
                int eo = -1;
                double big = TOL_PIV;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const double aj = fabs(T[j]);
                    const bool tk = aj > big && !((dead >> j) & 1u);
                    big = tk ? aj : big;
                    eo = tk ? j : eo;
                }
                const int ed = bcast(eo, rt);
                if (drive) {
                    e = ed;
                    if (ed < 0) {  // row "0 = t": redundant
                        if (g.lane == rt) rowact = false;
                        drive = false;
                        fin = -2;  // -> phase 2 without a pivot
                    }
                }
            }
        }
        if (init) e = init_col;
        bool act = normal | init | drive;
        e = act ? e : -1;
        if (__any(act)) {  // wave-uniform: the step that only detects optimality skips the pivot
            // ------------------------------------------------ selected column
            double a = 0.0, ce2 = 0.0;
            int vin = 0;
    #pragma unroll
            for (int j = 0; j < NC; ++j) {
                const bool mj = (j == e);
                a = mj ? T[j] : a;
                vin = mj ? cv[j] : vin;
                if constexpr (CARRY) ce2 = mj ? cost2[j] : ce2;
            }
            double ce = -best;  // normal mode: the (sign-flipped if necessary) reduced cost is -|c_e|
            const bool flip = normal & epos;  // free variable entering downwards: x := -x
            if (flip) { a = -a; ce2 = -ce2; }
            if constexpr (INITM || CARRY) {
                if (__any(init || drive)) {
                    double cr = 0.0;
    #pragma unroll
                    for (int j = 0; j < NC; ++j) cr = (j == e) ? cost[j] : cr;
                    if (init || drive) ce = cr;
                }
            }
            const bool efree = (cfree >> (e & 31)) & 1u;
            // ------------------------------------------------ ratio test
            // 1/a: v_rcp_f64 + two Newton steps (what the IEEE division expands to, minus scaling/fixup)
            const double x0 = __builtin_amdgcn_rcp(a);
            const double x1 = fma(x0, fma(-a, x0, 1.0), x0);
            const double pinv = fma(x1, fma(-a, x1, 1.0), x1);
            bool erow = normal & rowact & (a > TOL_PIV);
            double q = (beta > 0.0 ? beta : 0.0) * pinv;
            if constexpr (INITM) { if (init) { erow = init_elig; q = init_q; } }
            if constexpr (CARRY) { if (drive) { erow = (g.lane == rt); q = 0.0; } }
            q = erow ? q : __longlong_as_double(0x7ff0000000000000ll);
            // order-preserving u64 key of a double, split in two dwords; exact min = lexicographic min
            const int qh = __double2hiint(q), ql = __double2loint(q);
            const int sm = qh >> 31;
            const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
            const unsigned kl = (unsigned)(ql ^ sm);
            const unsigned mh = grp_min(kh, g.gs);
            const unsigned klm = (kh == mh) ? kl : 0xffffffffu;
            const unsigned ml = grp_min(klm, g.gs);
            if (act && mh >= 0xfff00000u) {  // +inf: no eligible row (or NaN): unbounded / numerical
                fin = (mh == 0xfff00000u && ml == 0u) ? ST_UNBND : ST_NUM;
                act = false;
                e = -1;
            }
            const bool tie = erow & (kh == mh) & (kl == ml);
            const uint64_t tbal = grp_ballot(tie, g);
            int rl = tbal ? __ffsll((long long)tbal) - 1 : 0;  // Dantzig mode: lowest row among ties
            if (__any(bland && act)) {                          // Bland mode: lowest basic-variable id
                const unsigned key = tie ? (unsigned)(rowvar + 1) : 0xffffffffu;
                const unsigned kmin = grp_min(key, g.gs);
                const uint64_t kb = grp_ballot(tie && key == kmin, g);
                if (bland) rl = kb ? __ffsll((long long)kb) - 1 : 0;
            }
            const bool is_r = act & (g.gl == rl);
            const int r = g.gbase + rl;
            if (normal && act) {
                const double qmin = __hiloint2double((int)(mh ^ 0x80000000u), (int)ml);
                ndeg = (qmin <= DEGEN_EPS) ? ndeg + 1 : 0;
            }
            // ------------------------------------------------ pivot (identity when !act)
            {
                const int raddr = r << 2;
                const double p = act ? bcast_addr(pinv, raddr) : 0.0;
                const double rhob = bcast_addr(beta, raddr) * p;
                const double f = (act & !is_r) ? a : 0.0;
                const double fc = act ? ce : 0.0;
                const double fc2 = act ? ce2 : 0.0;
                const double ecol = is_r ? p : -(f * p);
                const double ccol = -(fc * p);
    #pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const bool mj = (j == e);
                    const double rho = bcast_addr(T[j], raddr) * p;
                    double t = fma(-f, rho, T[j]);
                    t = is_r ? rho : t;
                    T[j] = mj ? ecol : t;
                    const double c = fma(-fc, rho, cost[j]);
                    cost[j] = mj ? ccol : c;
                    if constexpr (CARRY) {
                        const double c2 = fma(-fc2, rho, cost2[j]);
                        cost2[j] = mj ? -(fc2 * p) : c2;
                    }
                }
                beta = is_r ? rhob : fma(-f, rhob, beta);
                negz = fma(-fc, rhob, negz);
                if constexpr (CARRY) negz2 = fma(-fc2, rhob, negz2);
                // bookkeeping: entering <-> leaving variable
                const int rpack = __builtin_amdgcn_ds_bpermute(raddr, ((rowvar + 1) << 1) | rowneg);
    #pragma unroll
                for (int j = 0; j < NC; ++j) cv[j] = (j == e) ? rpack : cv[j];
                if (act) {
                    cfree &= ~(1u << e);
                    iters += 1;
                }
                if (is_r) {
                    rowvar = (vin >> 1) - 1;
                    rowneg = (vin & 1) ^ (flip ? 1 : 0);
                    rowact = !efree;  // a free variable never leaves again
                }
            }
            // Optimal right after this pivot?  Checking the fresh cost row here saves the extra lock-step
            // iteration that would otherwise only discover "no entering column".
            bool more = false;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double c = cost[j];
                more = more | ((fabs(c) > TOL_D) & ((((cfree >> j) & 1u) != 0u) | (c < 0.0)) &
                               (((dead >> j) & 1u) == 0u));
            }
            if (normal & act & !more) fin = ST_OPT;
        }
        // ------------------------------------------------ mode transitions
        if (INITM && init) {
            mode = (fin >= 0) ? M_DONE : mode_after_init;
            if (fin >= 0) status = fin;
            if (beta < 0.0 && rowact) beta = 0.0;  // rounding of the forced pivot
        } else if (CARRY && (mode == M_P1 || mode == M_DRIVE) && running) {
            if constexpr (CARRY) {
                bool to_p2 = false;
                if (mode == M_DRIVE) {
                    to_p2 = true;  // pivot done (or row found redundant)
                } else if (fin == ST_OPT) {
                    const uint64_t tb = grp_ballot(rowvar == ID_T, g);
                    const int rtt = g.gbase + (tb ? __ffsll((long long)tb) - 1 : 0);
                    const double tval = bcast(beta, rtt);
                    if (tb != 0 && tval > TOL_FEAS) { mode = M_DONE; status = ST_INFEAS; }
                    else if (tb != 0) mode = M_DRIVE;
                    else to_p2 = true;
                } else if (fin >= 0) {
                    mode = M_DONE;
                    status = (fin == ST_ITER) ? ST_ITER : ST_NUM;
                }
                if (to_p2) {
                    // the column that now holds t is dropped; the carried cost row becomes active
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        if ((cv[j] >> 1) == 0) dead |= (1u << j);
                        cost[j] = cost2[j];
                    }
                    negz = negz2;
                    if (rowact && beta < 0.0) beta = 0.0;
                    ndeg = 0;
                    mode = M_P2;
                }
            }
        } else if (running && fin >= 0) {
            mode = M_DONE;
            status = fin;
        }
    }

    __device__ __forceinline__ void run(const Grp& g) {
        while (__any(mode != M_DONE)) step(g);
    }
};

}  // namespace plp
