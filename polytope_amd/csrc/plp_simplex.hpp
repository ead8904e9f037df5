// plp_simplex.hpp -- register-resident dense simplex, one LP per lane group (gfx950).
//
// Problem (reference: polytope/solvers.py:76-106, 149-158):  min c'x  s.t.  Gx <= h,  x free.
//
// Layout: lane i of the group owns constraint row i of the dictionary
//       basic_i = beta - sum_j T[j] * nb_j                       (T[NC], beta : per lane)
//       (-zeta) = negz - sum_j cost[j] * nb_j                    (cost[NC]    : replicated in
//                                                                  every lane of the group)
// NC is a compile-time constant so T[] / cost[] / colvar[] live in VGPRs; a pivot is
//   * a scan over the NC replicated reduced costs (no cross-lane traffic) for the entering column,
//   * one f64 min all-reduce + one u32 min all-reduce over the group for the ratio test,
//   * NC+1 cross-lane broadcasts of the pivot row, and NC+1 FMAs per lane.
// Free variables (the x_j) enter in either direction and never leave; Dantzig pricing with a
// switch to Bland's rule after BLAND_AFTER consecutive degenerate pivots.  The same rules, in
// the same order, are restated in scalar C in oracle/plp_oracle.c (test infrastructure).
//
// All lanes of a wavefront execute every step (groups that are finished are masked by
// predication, never by divergent control flow, so cross-lane ops stay well defined).
#pragma once
#include "plp_wave.hpp"

namespace plp {

enum : int { M_INIT = 0, M_P1 = 1, M_DRIVE = 2, M_P2 = 3, M_DONE = 4 };

template <int NC, bool CARRY>
struct Simplex {
    // ---- per-lane row
    double T[NC];
    double beta;
    int rowvar;   // id of my basic variable
    int rowneg;   // 1: my basic free variable is stored negated
    bool rowact;  // my row takes part in ratio tests
    // ---- replicated per group
    double cost[NC], negz;
    double cost2[CARRY ? NC : 1], negz2;
    int colvar[NC];
    unsigned colneg;  // bit j: column j holds -x
    unsigned dead;    // bit j: column j never enters
    int n;            // ids < n are free structural variables
    int ndeg, iters, maxit;
    int mode, status;
    // ---- INIT pivot request (forced entering column, signed ratio supplied by the caller)
    int init_col;
    double init_q;
    bool init_elig;
    int mode_after_init;

    __device__ __forceinline__ bool isfree(int id) const { return (unsigned)id < (unsigned)n; }

    __device__ __forceinline__ void reset(int n_, int m_rows, int my_row) {
        n = n_;
        rowvar = n_ + my_row;
        rowneg = 0;
        rowact = true;
        negz = 0.0;
        negz2 = 0.0;
        colneg = 0u;
        dead = 0u;
        ndeg = 0;
        iters = 0;
        maxit = 50 * (m_rows + n_) + 100;
        status = -1;
        init_col = -1;
        init_q = 0.0;
        init_elig = false;
        mode_after_init = M_P2;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            colvar[j] = j;
            cost[j] = 0.0;
            T[j] = 0.0;
            if constexpr (CARRY) cost2[j] = 0.0;
        }
        if constexpr (!CARRY) cost2[0] = 0.0;
        beta = 0.0;
    }

    // One lockstep iteration for every group of the wavefront.
    __device__ __forceinline__ void step(const Grp& g) {
        const bool running = mode != M_DONE;
        // ------------------------------------------------ entering column
        int e = -1;
        {
            const bool bland = ndeg >= BLAND_AFTER;
            double best = 0.0;
            int bestid = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double dj = cost[j];
                const double adj = fabs(dj);
                const int id = colvar[j];
                bool elig = isfree(id) ? (adj > TOL_D) : (dj < -TOL_D);
                elig = elig && !((dead >> j) & 1u);
                const bool take = elig && (bland ? (id < bestid) : (adj > best));
                if (take) { e = j; best = adj; bestid = id; }
            }
        }
        int fin = -1;  // status this group finishes with in this iteration
        bool normal = running && (mode == M_P1 || mode == M_P2);
        if (normal && e < 0) { fin = ST_OPT; normal = false; }
        if (normal && iters >= maxit) { fin = ST_ITER; normal = false; }
        // ------------------------------------------------ special pivots
        const bool init = running && mode == M_INIT;
        bool drive = running && mode == M_DRIVE;
        int rt = -1;
        if constexpr (CARRY) {
            if (__any(drive)) {  // t is basic at ~0 after phase 1: pivot it out on its largest element
                const uint64_t tb = grp_ballot(rowvar == ID_T, g);
                rt = g.gbase + (tb ? __ffsll((long long)tb) - 1 : 0);
                int eo = -1;
                double big = TOL_PIV;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const double aj = fabs(T[j]);
                    if (aj > big && !((dead >> j) & 1u)) { big = aj; eo = j; }
                }
                const int ed = bcast(eo, rt);
                if (drive) {
                    e = ed;
                    if (ed < 0) {  // row "0 = t": redundant
                        if (g.lane == rt) rowact = false;
                        drive = false;
                        fin = -2;  // -> transition to phase 2 without a pivot
                    }
                }
            }
        }
        if (init) e = init_col;
        bool act = normal || init || drive;
        if (!act) e = -1;
        // ------------------------------------------------ selected column
        double a = 0.0, ce = 0.0, ce2 = 0.0;
        int vin = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (j == e) {
                a = T[j];
                ce = cost[j];
                vin = colvar[j];
                if constexpr (CARRY) ce2 = cost2[j];
            }
        }
        const bool flip = normal && ce > 0.0;  // free variable entering downwards: x := -x
        if (flip) { a = -a; ce = -ce; ce2 = -ce2; }
        // ------------------------------------------------ ratio test
        bool erow;
        double q;
        if (init) {
            erow = init_elig;
            q = init_q;
        } else if (drive) {
            erow = (g.lane == rt);
            q = 0.0;
        } else {
            erow = normal && rowact && (a > TOL_PIV);
            q = (beta > 0.0 ? beta : 0.0) / a;
        }
        q = erow ? q : INFINITY;
        const double qmin = grp_min(q, g.gs);
        if (act && qmin == INFINITY) {
            fin = ST_UNBND;
            act = false;
            e = -1;
        }
        const unsigned key = (erow && q == qmin) ? (unsigned)(rowvar + 1) : 0xffffffffu;
        const unsigned kmin = grp_min(key, g.gs);
        if (act && kmin == 0xffffffffu) {  // NaN in the ratio column: no row matched its own minimum
            fin = ST_NUM;
            act = false;
            e = -1;
        }
        const bool is_r = act && (key == kmin);
        const uint64_t rb = grp_ballot(is_r, g);
        const int r = g.gbase + (rb ? __ffsll((long long)rb) - 1 : 0);
        if (normal && act) ndeg = (qmin <= DEGEN_EPS) ? ndeg + 1 : 0;
        // ------------------------------------------------ pivot (identity when !act)
        {
            const double ar = bcast(a, r);
            const double p = act ? 1.0 / ar : 0.0;
            const double rhob = bcast(beta, r) * p;
            const double f = (act && !is_r) ? a : 0.0;
            const double fc = act ? ce : 0.0;
            const double fc2 = act ? ce2 : 0.0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double tr = bcast(T[j], r);
                const double rho = (j == e) ? p : tr * p;
                const double told = (j == e) ? 0.0 : T[j];
                const double tnew = fma(-f, rho, told);
                T[j] = is_r ? rho : tnew;
                const double cold = (j == e) ? 0.0 : cost[j];
                cost[j] = fma(-fc, rho, cold);
                if constexpr (CARRY) {
                    const double cold2 = (j == e) ? 0.0 : cost2[j];
                    cost2[j] = fma(-fc2, rho, cold2);
                }
            }
            beta = is_r ? rhob : fma(-f, rhob, beta);
            negz = fma(-fc, rhob, negz);
            if constexpr (CARRY) negz2 = fma(-fc2, rhob, negz2);
            // bookkeeping: entering <-> leaving variable
            const int rpack = bcast(((rowvar + 1) << 1) | rowneg, r);
            const int vout = (rpack >> 1) - 1;
            const unsigned inneg = ((colneg >> (e & 31)) & 1u) ^ (flip ? 1u : 0u);
#pragma unroll
            for (int j = 0; j < NC; ++j) colvar[j] = (j == e) ? vout : colvar[j];
            if (act) {
                colneg = (colneg & ~(1u << e)) | ((unsigned)(rpack & 1) << e);
                iters += 1;
            }
            if (is_r) {
                rowvar = vin;
                rowneg = (int)inneg;
                rowact = !isfree(vin);
            }
        }
        // ------------------------------------------------ mode transitions
        if (init) {
            mode = (fin >= 0) ? M_DONE : mode_after_init;
            if (fin >= 0) status = fin;
            if (beta < 0.0 && rowact) beta = 0.0;  // rounding of the forced pivot
        } else if (CARRY && (mode == M_P1 || mode == M_DRIVE) && running) {
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
                    if (colvar[j] == ID_T) dead |= (1u << j);
                    cost[j] = cost2[j];
                }
                negz = negz2;
                if (rowact && beta < 0.0) beta = 0.0;
                ndeg = 0;
                mode = M_P2;
            }
        } else if (running && fin >= 0) {
            mode = M_DONE;
            status = fin;
        }
    }

    __device__ __forceinline__ void run(const Grp& g) {
        while (__any(mode != M_DONE)) step(g);
    }

    // value of structural variable `id` held by my row (0 if my basic variable is another one)
    __device__ __forceinline__ bool holds_x() const { return isfree(rowvar); }
    __device__ __forceinline__ double x_value() const { return rowneg ? -beta : beta; }
};

}  // namespace plp
