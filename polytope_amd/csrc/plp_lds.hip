// plp_lds.hip -- LP engine with the dictionary in LDS (gfx950): one LP per wavefront, rows striped over the lanes.
//
//   lp_lds_kernel    : lpsolve() batches (solvers.py:76-106, 149-158) of ANY row count that fits the CU's LDS
//   cheby_lds_kernel : Chebyshev-ball LPs (form F1, polytope.py:1283-1288) likewise
//
// The register-resident engines (plp_simplex.hpp, plp_simplex_r.hpp) give every dictionary row a lane (or a
// quarter of one) and stop at 64 rows.  region_diff (polytope.py:2117-2282) stacks m_poly + sum(active rows) and
// passes that limit on a 1000-cell Region (BASELINE config 4: up to 73 rows; the reference has no limit), and its
// leaf pieces go through reduce() with as many rows.  Here the dictionary T[m][nc], beta[m] and the cost rows
// live in LDS, where a dynamic column index is an address instead of a chain of selects; lane l owns rows
// l, l+64, l+128, ...  The pivot rules, tolerances and the arithmetic of every dictionary entry are those of
// plp_simplex.hpp's step() (two-phase, Dantzig pricing, Bland's rule after BLAND_AFTER degenerate pivots, free
// variables enter in either direction and never leave, 1/a = v_rcp_f64 + 2 Newton steps), so an LP that fits
// both engines walks the same vertex path and ends with bitwise the same numbers (tests: PLP_LDS=1).
#include <stdlib.h>

#include "plp_kernels.hpp"
#include "plp_wave.hpp"
#include "plp_simplex.hpp"  // mode names

namespace plp {

namespace {

constexpr int LDS_MAXC = 32;  // column slots (n <= 17 structural + artificial)

struct LdsDict {
    double* T;      // [m][ld]
    double* beta;   // [m]
    double* qk;     // [m]   ratio of row i in the current pivot / forced-pivot ratios (INIT)
    int* rowvar;    // [m]   id of the basic variable
    int* rowfl;     // [m]   bit 0: stored negated, bit 1: takes part in ratio tests
    double* cost;   // [LDS_MAXC]
    double* cost2;  // [LDS_MAXC]
    double* rho;    // [LDS_MAXC]
    int* cv;        // [LDS_MAXC] packed (id+1)<<1 | negated
    int ld;
};

__device__ __forceinline__ LdsDict lds_carve(unsigned char* base, int m, int ld) {
    LdsDict D;
    D.ld = ld;
    D.T = reinterpret_cast<double*>(base);
    D.beta = D.T + (size_t)m * ld;
    D.qk = D.beta + m;
    D.cost = D.qk + m;
    D.cost2 = D.cost + LDS_MAXC;
    D.rho = D.cost2 + LDS_MAXC;
    D.rowvar = reinterpret_cast<int*>(D.rho + LDS_MAXC);
    D.rowfl = D.rowvar + m;
    D.cv = D.rowfl + m;
    return D;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

// order-preserving u64 key of a double (plp_simplex.hpp: the exact f64 minimum is the minimum of the keys)
__device__ __forceinline__ unsigned long long key_of(double q) {
    const int qh = __double2hiint(q), ql = __double2loint(q);
    const int sm = qh >> 31;
    const unsigned kh = (unsigned)(qh ^ (sm | (int)0x80000000));
    const unsigned kl = (unsigned)(ql ^ sm);
    return ((unsigned long long)kh << 32) | kl;
}
__device__ __forceinline__ double unkey(unsigned long long k) {  // for keys of non-negative doubles
    return __hiloint2double((int)((unsigned)(k >> 32) ^ 0x80000000u), (int)(unsigned)k);
}

__device__ __forceinline__ double rcp_newton(double a) {
    const double x0 = __builtin_amdgcn_rcp(a);
    const double x1 = fma(x0, fma(-a, x0, 1.0), x0);
    return fma(x1, fma(-a, x1, 1.0), x1);
}

// Wave-uniform solver state (one LP per wavefront).
struct LdsState {
    int m, n, nc;
    unsigned cfree, dead;
    int ndeg, iters, maxit, mode, status;
    int init_col, mode_after_init;
    double negz, negz2;
    bool carry;
};

__device__ __forceinline__ bool col_eligible(const LdsState& S, int j, double c) {
    return (fabs(c) > TOL_D) & ((((S.cfree >> j) & 1u) != 0u) | (c < 0.0)) & (((S.dead >> j) & 1u) == 0u);
}

// One iteration (plp_simplex.hpp Simplex::step, same order of decisions).  All lanes take every branch together.
__device__ void lds_step(LdsState& S, const LdsDict& D, int lane) {
    const bool bland = S.ndeg >= BLAND_AFTER;
    const int m = S.m, nc = S.nc, ld = D.ld;
    // ---- entering column: lane j looks at column j
    int e = -1;
    double best = 0.0;
    bool epos = false;
    {
        const double c = lane < nc ? D.cost[lane] : 0.0;
        const bool elig = (lane < nc) && col_eligible(S, lane, c);
        const uint64_t eb = __ballot(elig);
        if (eb) {
            if (!bland) {  // largest |c|, first column among equals
                const unsigned long long k = elig ? key_of(-fabs(c)) : ~0ull;
                const unsigned long long kmin = wave_min_u64(k);
                const uint64_t tb = __ballot(elig && k == kmin);
                e = __ffsll((long long)tb) - 1;
            } else {  // lowest variable id
                const unsigned long long k = elig ? (unsigned long long)(unsigned)D.cv[lane] : ~0ull;
                const unsigned long long kmin = wave_min_u64(k);
                const uint64_t tb = __ballot(elig && k == kmin);
                e = __ffsll((long long)tb) - 1;
            }
            const double ce_ = D.cost[e];
            best = fabs(ce_);
            epos = ce_ > 0.0;
        }
    }
    int fin = -1;
    bool normal = (S.mode == M_P1) | (S.mode == M_P2);
    if (normal && e < 0) { fin = ST_OPT; normal = false; }
    if (normal && S.iters >= S.maxit) { fin = ST_ITER; normal = false; }
    const bool init = S.mode == M_INIT;
    bool drive = S.carry && S.mode == M_DRIVE;
    int rt = -1;
    if (drive) {  // t is basic at ~0 after phase 1: pivot it out on the largest element of its row
        int cand = 0x7fffffff;
        for (int i = lane; i < m; i += 64) cand = (D.rowvar[i] == ID_T && i < cand) ? i : cand;
        rt = (int)wave_min_u64((unsigned long long)(unsigned)cand);
        if (rt == 0x7fffffff) rt = 0;
        const double aj = lane < nc ? fabs(D.T[(size_t)rt * ld + lane]) : 0.0;
        const bool ok = (lane < nc) && (aj > TOL_PIV) && !((S.dead >> lane) & 1u);
        int eo = -1;
        if (__ballot(ok)) {
            const unsigned long long k = ok ? key_of(-aj) : ~0ull;
            const unsigned long long kmin = wave_min_u64(k);
            eo = __ffsll((long long)__ballot(ok && k == kmin)) - 1;
        }
        e = eo;
        if (eo < 0) {  // row "0 = t": redundant
            if (lane == 0) D.rowfl[rt] &= ~2;
            drive = false;
            fin = -2;  // -> phase 2 without a pivot
        }
    }
    if (init) e = S.init_col;
    bool act = normal | init | drive;
    if (act) {
        const bool flip = normal & epos;  // free variable entering downwards: x := -x
        double ce = -best;
        if (init | drive) ce = D.cost[e];
        double ce2 = S.carry ? D.cost2[e] : 0.0;
        if (flip) ce2 = -ce2;
        const int vin = D.cv[e];
        const bool efree = (S.cfree >> e) & 1u;
        // ---- ratio test over my rows
        unsigned long long kbest = ~0ull;
        for (int i = lane; i < m; i += 64) {
            double a = D.T[(size_t)i * ld + e];
            a = flip ? -a : a;
            const double b = D.beta[i];
            bool erow = normal & ((D.rowfl[i] & 2) != 0) & (a > TOL_PIV);
            double q = (b > 0.0 ? b : 0.0) * rcp_newton(a);
            if (init) { erow = (D.rowfl[i] & 2) != 0; q = D.qk[i]; }  // forced pivot: caller-supplied ratios
            if (drive) { erow = i == rt; q = 0.0; }
            q = erow ? q : __longlong_as_double(0x7ff0000000000000ll);
            const unsigned long long k = key_of(q);
            if (!init) D.qk[i] = __longlong_as_double((long long)k);
            kbest = k < kbest ? k : kbest;
        }
        const unsigned long long kmin = wave_min_u64(kbest);
        const unsigned mh = (unsigned)(kmin >> 32), ml = (unsigned)kmin;
        if (mh >= 0xfff00000u) {  // +inf: no eligible row (or NaN): unbounded / numerical
            fin = (mh == 0xfff00000u && ml == 0u) ? ST_UNBND : ST_NUM;
            act = false;
        }
        if (act) {
            // lowest row among the ties (Dantzig), lowest basic-variable id among them (Bland)
            unsigned long long pick = ~0ull;
            for (int i = lane; i < m; i += 64) {
                const unsigned long long k = init ? key_of(((D.rowfl[i] & 2) != 0) ? D.qk[i] : __longlong_as_double(0x7ff0000000000000ll))
                                                  : (unsigned long long)__double_as_longlong(D.qk[i]);
                if (k == kmin) {
                    const unsigned long long c =
                        (bland & normal) ? (((unsigned long long)(unsigned)(D.rowvar[i] + 1) << 32) | (unsigned)i)
                                         : (unsigned long long)(unsigned)i;
                    pick = c < pick ? c : pick;
                }
            }
            const int r = (int)(unsigned)wave_min_u64(pick);
            if (normal) S.ndeg = (unkey(kmin) <= DEGEN_EPS) ? S.ndeg + 1 : 0;
            // ---- pivot row scaled: rho_j = T[r][j] * p, rho_e = p
            double ar = D.T[(size_t)r * ld + e];
            ar = flip ? -ar : ar;
            const double p = rcp_newton(ar);
            const double rhob = D.beta[r] * p;
            if (lane < nc) D.rho[lane] = D.T[(size_t)r * ld + lane] * p;
            const int rpack = ((D.rowvar[r] + 1) << 1) | (D.rowfl[r] & 1);
            __syncthreads();
            // ---- update my rows
            for (int i = lane; i < m; i += 64) {
                double* Ti = D.T + (size_t)i * ld;
                if (i == r) continue;
                double f = Ti[e];
                f = flip ? -f : f;
                for (int j = 0; j < nc; ++j) Ti[j] = fma(-f, D.rho[j], Ti[j]);
                Ti[e] = -(f * p);
                D.beta[i] = fma(-f, rhob, D.beta[i]);
            }
            __syncthreads();
            if (lane < nc) {
                const double rj = D.rho[lane];
                D.T[(size_t)r * ld + lane] = lane == e ? p : rj;
                D.cost[lane] = lane == e ? -(ce * p) : fma(-ce, rj, D.cost[lane]);
                if (S.carry) D.cost2[lane] = lane == e ? -(ce2 * p) : fma(-ce2, rj, D.cost2[lane]);
            }
            S.negz = fma(-ce, rhob, S.negz);
            if (S.carry) S.negz2 = fma(-ce2, rhob, S.negz2);
            if (lane == 0) {
                D.beta[r] = rhob;
                D.cv[e] = rpack;
                D.rowvar[r] = (vin >> 1) - 1;
                D.rowfl[r] = ((vin & 1) ^ (flip ? 1 : 0)) | (efree ? 0 : 2);  // a free variable never leaves again
            }
            S.cfree &= ~(1u << e);
            S.iters += 1;
            __syncthreads();
            // optimal right after this pivot?
            if (normal) {
                const double c = lane < nc ? D.cost[lane] : 0.0;
                if (!__ballot((lane < nc) && col_eligible(S, lane, c))) fin = ST_OPT;
            }
        }
    }
    // ---- mode transitions
    if (init) {
        S.mode = (fin >= 0) ? M_DONE : S.mode_after_init;
        if (fin >= 0) S.status = fin;
        for (int i = lane; i < m; i += 64)
            if ((D.rowfl[i] & 2) && D.beta[i] < 0.0) D.beta[i] = 0.0;  // rounding of the forced pivot
        __syncthreads();
    } else if (S.carry && (S.mode == M_P1 || S.mode == M_DRIVE)) {
        bool to_p2 = false;
        if (S.mode == M_DRIVE) {
            to_p2 = true;
        } else if (fin == ST_OPT) {
            int cand = 0x7fffffff;
            for (int i = lane; i < m; i += 64) cand = (D.rowvar[i] == ID_T && i < cand) ? i : cand;
            const int rtt = (int)wave_min_u64((unsigned long long)(unsigned)cand);
            if (rtt != 0x7fffffff) {
                const double tval = D.beta[rtt];
                if (tval > TOL_FEAS) { S.mode = M_DONE; S.status = ST_INFEAS; }
                else S.mode = M_DRIVE;
            } else {
                to_p2 = true;
            }
        } else if (fin >= 0) {
            S.mode = M_DONE;
            S.status = (fin == ST_ITER) ? ST_ITER : ST_NUM;
        }
        if (to_p2) {  // the column that now holds t is dropped; the carried cost row becomes active
            unsigned dd = 0u;
            if (lane < nc) {
                dd = ((D.cv[lane] >> 1) == 0) ? (1u << lane) : 0u;
                D.cost[lane] = D.cost2[lane];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) dd |= (unsigned)__shfl_xor((int)dd, o, 64);
            S.dead |= dd;
            S.negz = S.negz2;
            for (int i = lane; i < m; i += 64)
                if ((D.rowfl[i] & 2) && D.beta[i] < 0.0) D.beta[i] = 0.0;
            S.ndeg = 0;
            S.mode = M_P2;
            __syncthreads();
        }
    } else if (fin >= 0) {
        S.mode = M_DONE;
        S.status = fin;
    }
}

// x_j of the final dictionary (0 when nonbasic), every lane gets the value
__device__ __forceinline__ double lds_x_of(const LdsDict& D, int m, int j, int lane) {
    double v = 0.0;
    bool found = false;
    for (int i = lane; i < m; i += 64) {
        if (D.rowvar[i] == j) { v = (D.rowfl[i] & 1) ? -D.beta[i] : D.beta[i]; found = true; }
    }
    const uint64_t ob = __ballot(found);
    const double w = __shfl(v, ob ? __ffsll((long long)ob) - 1 : 0, 64);
    return ob ? w : 0.0;
}

}  // namespace

__global__ __launch_bounds__(64) void lp_lds_kernel(long long B, int m_max, int n, const double* __restrict__ c,
                                                    const double* __restrict__ G, const double* __restrict__ h,
                                                    const int* __restrict__ mrows, double* __restrict__ x,
                                                    double* __restrict__ fun, int* __restrict__ status,
                                                    int* __restrict__ iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int nc = n + 1;  // + phase-1 artificial
    const int ld = nc | 1;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    for (long long lp = blockIdx.x; lp < B; lp += gridDim.x) {
        const int m = mrows ? mrows[lp] : m_max;
        const LdsDict D = lds_carve(smem_raw, m_max, ld);
        __syncthreads();
        LdsState S;
        S.m = m; S.n = n; S.nc = nc;
        S.cfree = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
        S.dead = 0u; S.ndeg = 0; S.iters = 0; S.maxit = 50 * (m + n) + 100;
        S.mode = M_P2; S.status = -1; S.init_col = -1; S.mode_after_init = M_P2;
        S.negz = 0.0; S.negz2 = 0.0; S.carry = true;
        bool finite = true, inf0 = false, neg = false;
        for (int i = lane; i < m; i += 64) {
            const double* Gr = G + (lp * m_max + i) * n;
            double* Ti = D.T + (size_t)i * ld;
            bool zero = true;
            for (int j = 0; j < n; ++j) {
                const double v = Gr[j];
                Ti[j] = v;
                zero = zero & (v == 0.0);
                finite = finite & isfinite(v);
            }
            Ti[n] = 0.0;
            const double hi = h[lp * m_max + i];
            finite = finite & isfinite(hi);
            D.beta[i] = zero ? 0.0 : hi;
            D.rowvar[i] = n + i;
            D.rowfl[i] = zero ? 0 : 2;
            inf0 = inf0 | (zero & (hi < -TOL_FEAS));  // 0 <= h_i < 0
            neg = neg | (!zero & (hi < 0.0));
        }
        double cj = 0.0;
        if (lane < nc) {
            cj = lane < n ? c[lp * n + lane] : 0.0;
            finite = finite & isfinite(cj);
            D.cv[lane] = lane < n ? ((lane + 1) << 1) : 0;  // last column: the artificial variable t (id -1)
        }
        const bool bad = __ballot(!finite) != 0;
        const bool infeasible0 = __ballot(inf0) != 0;
        const bool need_p1 = __ballot(neg) != 0;
        if (need_p1) {
            for (int i = lane; i < m; i += 64) {
                if (D.rowfl[i] & 2) D.T[(size_t)i * ld + n] = -1.0;
                D.qk[i] = D.beta[i];
            }
            if (lane < nc) { D.cost[lane] = lane == n ? 1.0 : 0.0; D.cost2[lane] = cj; }
            S.mode = M_INIT; S.init_col = n; S.mode_after_init = M_P1;
        } else {
            if (lane < nc) { D.cost[lane] = cj; D.cost2[lane] = 0.0; }
            S.dead = 1u << n;
        }
        if (bad) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
        __syncthreads();
        while (S.mode != M_DONE) lds_step(S, D, lane);
        const bool ok = S.status == ST_OPT;
        double f = 0.0;
        for (int j = 0; j < n; ++j) {
            const double xj = lds_x_of(D, m, j, lane);
            f = fma(c[lp * n + j], xj, f);
            if (lane == 0) x[lp * n + j] = ok ? xj : qnan;
        }
        if (lane == 0) {
            fun[lp] = ok ? f : qnan;
            status[lp] = S.status;
            if (iters) iters[lp] = S.iters;
        }
    }
}

__global__ __launch_bounds__(64) void cheby_lds_kernel(long long B, int m_max, int d, const double* __restrict__ A,
                                                       const double* __restrict__ b, const int* __restrict__ mrows,
                                                       double* __restrict__ r, double* __restrict__ xc,
                                                       int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int nc = d + 1;
    const int ld = nc | 1;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    for (long long p = blockIdx.x; p < B; p += gridDim.x) {
        const int m = mrows ? mrows[p] : m_max;
        const LdsDict D = lds_carve(smem_raw, m_max, ld);
        __syncthreads();
        LdsState S;
        S.m = m; S.n = nc; S.nc = nc;
        S.cfree = (1u << nc) - 1u;
        S.dead = 0u; S.ndeg = 0; S.iters = 0; S.maxit = 50 * (m + nc) + 100;
        S.mode = M_INIT; S.status = -1; S.init_col = d; S.mode_after_init = M_P2;
        S.negz = 0.0; S.negz2 = 0.0; S.carry = false;
        bool finite = true, inf0 = false;
        for (int i = lane; i < m; i += 64) {
            const double* Ar = A + (p * m_max + i) * d;
            double* Ti = D.T + (size_t)i * ld;
            double nrm2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double v = Ar[k];
                Ti[k] = v;
                nrm2 = nrm2 + v * v;
                finite = finite & isfinite(v);
            }
            const double bi = b[p * m_max + i];
            finite = finite & isfinite(bi);
            const double nrm = sqrt(nrm2);
            const bool zero = !(nrm > 0.0);
            Ti[d] = zero ? 0.0 : nrm;
            D.beta[i] = zero ? 0.0 : bi;
            D.qk[i] = bi / nrm;
            D.rowvar[i] = nc + i;
            D.rowfl[i] = zero ? 0 : 2;
            inf0 = inf0 | (zero & (bi < -TOL_FEAS));
        }
        if (lane < nc) {
            D.cost[lane] = lane == d ? -1.0 : 0.0;
            D.cost2[lane] = 0.0;
            D.cv[lane] = (lane + 1) << 1;
        }
        const bool bad = __ballot(!finite) != 0;
        const bool infeasible0 = __ballot(inf0) != 0;
        if (bad) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (infeasible0) { S.mode = M_DONE; S.status = ST_INFEAS; }
        __syncthreads();
        while (S.mode != M_DONE) lds_step(S, D, lane);
        const bool ok = S.status == ST_OPT;
        for (int j = 0; j < nc; ++j) {
            const double xj = lds_x_of(D, m, j, lane);
            if (lane == 0) {
                if (j < d) xc[p * d + j] = ok ? xj : qnan; else r[p] = ok ? xj : qnan;
            }
        }
        if (lane == 0) status[p] = S.status;
    }
}

// Chebyshev LPs on row subsets of one resident table (region_diff's search, see plp_rdiff.hip): index lists of any
// length that fits LDS.  out[p] = x[-1] if optimal with r >= 0, else 0 (cheby_ball's reading, polytope.py:1289-1297).
__global__ __launch_bounds__(64) void cheby_gather_lds_kernel(long long nlp, int m_cap, int d, const int* __restrict__ off,
                                                              const int* __restrict__ rows, const int* __restrict__ sel,
                                                              const double* __restrict__ A, const double* __restrict__ b,
                                                              double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int nc = d + 1;
    const int ld = nc | 1;
    for (long long q = blockIdx.x; q < nlp; q += gridDim.x) {
        const int p = sel[q];
        const int o = off[p];
        const int m = off[p + 1] - o;
        const LdsDict D = lds_carve(smem_raw, m_cap, ld);
        __syncthreads();
        LdsState S;
        S.m = m; S.n = nc; S.nc = nc;
        S.cfree = (1u << nc) - 1u;
        S.dead = 0u; S.ndeg = 0; S.iters = 0; S.maxit = 50 * (m + nc) + 100;
        S.mode = M_INIT; S.status = -1; S.init_col = d; S.mode_after_init = M_P2;
        S.negz = 0.0; S.negz2 = 0.0; S.carry = false;
        bool finite = true, inf0 = false;
        for (int i = lane; i < m; i += 64) {
            const long long src = rows[o + i];
            const double* Ar = A + src * d;
            double* Ti = D.T + (size_t)i * ld;
            double nrm2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double v = Ar[k];
                Ti[k] = v;
                nrm2 = nrm2 + v * v;
                finite = finite & isfinite(v);
            }
            const double bi = b[src];
            finite = finite & isfinite(bi);
            const double nrm = sqrt(nrm2);
            const bool zero = !(nrm > 0.0);
            Ti[d] = zero ? 0.0 : nrm;
            D.beta[i] = zero ? 0.0 : bi;
            D.qk[i] = bi / nrm;
            D.rowvar[i] = nc + i;
            D.rowfl[i] = zero ? 0 : 2;
            inf0 = inf0 | (zero & (bi < -TOL_FEAS));
        }
        if (lane < nc) {
            D.cost[lane] = lane == d ? -1.0 : 0.0;
            D.cost2[lane] = 0.0;
            D.cv[lane] = (lane + 1) << 1;
        }
        if (__ballot(!finite) != 0) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (__ballot(inf0) != 0) { S.mode = M_DONE; S.status = ST_INFEAS; }
        __syncthreads();
        while (S.mode != M_DONE) lds_step(S, D, lane);
        const double rr = lds_x_of(D, m, d, lane);
        if (lane == 0) out[p] = S.status != ST_OPT ? __builtin_nan("") : (rr >= 0.0 ? rr : 0.0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// reduce_lds_kernel -- fused reduce() (polytope/polytope.py:1053-1163) of polytopes with MORE THAN 64 ROWS: the stacks
// Polytope.intersect builds (m1 + m2 rows, :268-275) and the leaves of region_diff (:2276) have no row limit in the
// reference.  One polytope per wavefront; its rows (A, b, s = A xc, a state word per row) stay in LDS next to the
// dictionary of the LP being solved; lane l owns rows l, l + 64, ...  The pipeline and every reference step are those
// of reduce_r_tile (plp_reduce_r_impl.hpp): F1 -> parallel-row dedupe -> 2d F3 LPs + prefilter (rows > 3d) -> the F2
// presolve (same two witness points) -> one F2 LP per row it leaves, each on lds_step (the general engine: Bland's rule
// in the loop, no hand-over pass).  keep is W = ceil(m_max / 64) words per polytope.
namespace {

// ROW_SETTLED: live and settled as "keep" by the presolve.  ROW_DROPPED: found redundant by its F2 LP -- still a row of
// every later LP (the reference solves all of them over G = A_arr, every row, :1142-1160; only the keep mask loses it).
constexpr int ROW_DEAD = 0, ROW_LIVE = 1, ROW_SETTLED = 2, ROW_DROPPED = 3;

struct RedRows {
    double* A;    // [m][d]
    double* b;    // [m]
    double* s;    // [m]  1 / ||a_i|| during the dedupe, then a_i . xc
    double* w1;   // [m]  prefilter sums (:1131-1134)
    double* w2;   // [m]
    double* xc;   // [16] the Chebyshev centre (every lane reads it; no cross-lane reads inside loops of uneven trip count)
    int* state;   // [m]
};

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dictionary of an F2 / F3 LP: rows as they are (dead rows are zero), beta = max(b - s, 0), given cost row
__device__ __forceinline__ void red_setup_lp(LdsState& S, const LdsDict& D, const RedRows& R, int m, int d, int nlive,
                                             int lane, double cost_lane) {
    const int ld = D.ld;
    S.m = m; S.n = d; S.nc = d;
    S.cfree = d >= 32 ? 0xffffffffu : ((1u << d) - 1u);
    S.dead = 0u; S.ndeg = 0; S.iters = 0; S.maxit = 50 * (nlive + d) + 100;
    S.mode = M_P2; S.status = -1; S.init_col = -1; S.mode_after_init = M_P2;
    S.negz = 0.0; S.negz2 = 0.0; S.carry = false;
    for (int i = lane; i < m; i += 64) {
        double* Ti = D.T + (size_t)i * ld;
        for (int k = 0; k < d; ++k) Ti[k] = R.A[i * d + k];
        D.beta[i] = fmax(R.b[i] - R.s[i], 0.0);
        D.rowvar[i] = d + i;
        D.rowfl[i] = R.state[i] != ROW_DEAD ? 2 : 0;
    }
    if (lane < d) {
        D.cost[lane] = cost_lane;
        D.cost2[lane] = 0.0;
        D.cv[lane] = (lane + 1) << 1;
    }
    __syncthreads();
}

}  // namespace

__global__ __launch_bounds__(64) void reduce_lds_kernel(long long B, int m_max, int d, int W, const double* __restrict__ Ag,
                                                        const double* __restrict__ bg, const int* __restrict__ mrows,
                                                        double abs_tol, unsigned long long* __restrict__ keep_out,
                                                        int* __restrict__ flags_out, double* __restrict__ r_out,
                                                        double* __restrict__ xc_out, int* __restrict__ nlp_out,
                                                        size_t dict_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int nc1 = d + 1;
    const int ld = nc1 | 1;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double pinf = __longlong_as_double(0x7ff0000000000000ll);
    RedRows R;
    R.A = reinterpret_cast<double*>(smem_raw + dict_bytes);
    R.b = R.A + (size_t)m_max * d;
    R.s = R.b + m_max;
    R.w1 = R.s + m_max;
    R.w2 = R.w1 + m_max;
    R.xc = R.w2 + m_max;
    R.state = reinterpret_cast<int*>(R.xc + 16);
    for (long long p = blockIdx.x; p < B; p += gridDim.x) {
        const int m = mrows ? mrows[p] : m_max;
        if (m < 0 || m > m_max) {   // a row count the carved LDS does not hold (caller's device array): no verdict
            for (int w = lane; w < W; w += 64) keep_out[p * W + w] = 0ull;
            if (lane == 0) { flags_out[p] = RF_EMPTY | RF_LPFAIL; nlp_out[p] = 0; r_out[p] = 0.0; }
            if (lane < d) xc_out[p * d + lane] = qnan;
            continue;
        }
        const LdsDict D = lds_carve(smem_raw, m_max, ld);
        __syncthreads();
        // ---------------------------------------------------------------- rows -> LDS; F1 (as cheby_lds_kernel)
        LdsState S;
        S.m = m; S.n = nc1; S.nc = nc1;
        S.cfree = (1u << nc1) - 1u;
        S.dead = 0u; S.ndeg = 0; S.iters = 0; S.maxit = 50 * (m + nc1) + 100;
        S.mode = M_INIT; S.status = -1; S.init_col = d; S.mode_after_init = M_P2;
        S.negz = 0.0; S.negz2 = 0.0; S.carry = false;
        bool finite = true, inf0 = false;
        for (int i = lane; i < m_max; i += 64) {
            const bool h = i < m;
            double* Ti = D.T + (size_t)i * ld;
            double nrm2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double v = h ? Ag[(p * m_max + i) * d + k] : 0.0;
                R.A[i * d + k] = v;
                if (h) Ti[k] = v;
                nrm2 = nrm2 + v * v;
                finite = finite & isfinite(v);
            }
            const double bi = h ? bg[p * m_max + i] : 0.0;
            finite = finite & isfinite(bi);
            R.b[i] = bi;
            const double nrm = sqrt(nrm2);
            R.s[i] = 1.0 / nrm;
            R.state[i] = h ? ROW_LIVE : ROW_DEAD;
            if (h) {
                const bool zero = !(nrm > 0.0);
                Ti[d] = zero ? 0.0 : nrm;
                D.beta[i] = zero ? 0.0 : bi;
                D.qk[i] = bi / nrm;
                D.rowvar[i] = nc1 + i;
                D.rowfl[i] = zero ? 0 : 2;
                inf0 = inf0 | (zero & (bi < -TOL_FEAS));
            }
        }
        if (lane < nc1) {
            D.cost[lane] = lane == d ? -1.0 : 0.0;
            D.cost2[lane] = 0.0;
            D.cv[lane] = (lane + 1) << 1;
        }
        if (__ballot(!finite) != 0) { S.mode = M_DONE; S.status = ST_NUM; }
        else if (__ballot(inf0) != 0) { S.mode = M_DONE; S.status = ST_INFEAS; }
        __syncthreads();
        while (S.mode != M_DONE) lds_step(S, D, lane);
        double xcl = 0.0;  // lane j < d: xc[j]
        for (int j = 0; j < d; ++j) {
            const double xj = lds_x_of(D, m, j, lane);
            xcl = (lane == j) ? xj : xcl;
        }
        const double rr = lds_x_of(D, m, d, lane);
        if (lane < d) R.xc[lane] = xcl;
        bool ball = (S.status == ST_OPT) & (rr >= 0.0);  // cheby_ball: status 0 and r >= 0 (:1289-1293)
        __syncthreads();
        {   // a centre that violates a row (centre_off, plp_common.hpp) is no centre: RF_F1OPEN
            double xs = 1.0;
            for (int k = 0; k < d; ++k) xs = fmax(xs, fabs(R.xc[k]));
            bool off = false;
            for (int i = lane; i < m; i += 64) {
                double sk = 0.0;
                for (int k = 0; k < d; ++k) sk = fma(R.A[i * d + k], R.xc[k], sk);
                off = off | centre_off(R.b[i] - sk, R.s[i], R.b[i], xs);
            }
            if (ball & (__ballot(off) != 0)) { ball = false; S.status = ST_NUM; }
        }
        const bool fulldim = ball & (rr > abs_tol);
        int flags = fulldim ? 0 : (RF_EMPTY | (((S.status != ST_OPT) & (S.status != ST_INFEAS)) ? RF_F1OPEN : 0));
        int nlp = 1;
        int stage = 0;  // 0 done, 1 needs the box, 2 needs the redundancy LPs
        int neq = 0;
        __syncthreads();
        if (fulldim) {
            // ---------------------------------------------------------------- dedupe (:1094-1110), as reduce_r_tile
            for (int i = lane; i < m; i += 64) {
                const double an_i = R.s[i];
                const double bin_ = R.b[i] * an_i;
                bool removed = false;
                for (int j = 0; j < m; ++j) {
                    const double an_j = R.s[j];
                    double dot = 0.0;
                    for (int k = 0; k < d; ++k) dot = dot + (R.A[i * d + k] * an_i) * (R.A[j * d + k] * an_j);
                    const double bjn = R.b[j] * an_j;
                    const bool par = (j != i) & (dot > 1.0 - abs_tol);
                    removed = removed | (par & ((i < j) ? !(bin_ < bjn) : (bjn < bin_)));
                }
                if (removed) R.state[i] = ROW_DEAD;
            }
            __syncthreads();
            // s_i = a_i . xc replaces 1 / ||a_i||; dead rows are zeroed (A, b, s) so that the LP set-ups load them as they are
            int cnt = 0;
            for (int i = lane; i < m_max; i += 64) {
                const bool alive = R.state[i] != ROW_DEAD;
                double sk = 0.0;
                for (int k = 0; k < d; ++k) {
                    const double v = alive ? R.A[i * d + k] : 0.0;
                    R.A[i * d + k] = v;
                    sk = fma(v, R.xc[k], sk);
                }
                R.s[i] = sk;
                if (!alive) R.b[i] = 0.0;
                cnt += alive ? 1 : 0;
            }
            neq = wave_sum_i(cnt);
            __syncthreads();
            if (neq <= d + 1) flags = RF_EARLY;
            else stage = (neq > 3 * d) ? 1 : 2;
        }
        // ---------------------------------------------------------------- F3: bounding box (:1367-1409) + prefilter (:1118-1134)
        if (stage == 1) {
            for (int i = lane; i < m; i += 64) { R.w1[i] = 0.0; R.w2[i] = 0.0; }
            bool lpfail = false;
            double lbk = 0.0;
            for (int it = 0; it < 2 * d; ++it) {  // lower_0, upper_0, lower_1, upper_1, ...
                const int kx = it >> 1;
                const bool up = it & 1;
                red_setup_lp(S, D, R, m, d, neq, lane, lane == kx ? (up ? -1.0 : 1.0) : 0.0);
                while (S.mode != M_DONE) lds_step(S, D, lane);
                const double xck = R.xc[kx];
                double val;
                if (S.status == ST_OPT) val = up ? (xck + S.negz) : (xck - S.negz);
                else if (S.status == ST_UNBND) val = up ? pinf : -pinf;
                else { val = qnan; lpfail = true; }
                if (!up) {
                    lbk = val;
                } else {
                    for (int i = lane; i < m; i += 64) {
                        const double aik = R.A[i * d + kx];
                        const double pa = (aik > 0.0 ? 1.0 : 0.0) * aik;
                        R.w1[i] = R.w1[i] + pa * (val - lbk);
                        R.w2[i] = R.w2[i] + aik * lbk;
                    }
                }
                __syncthreads();
            }
            int cnt = 0;
            for (int i = lane; i < m; i += 64) {
                const bool alive = R.state[i] != ROW_DEAD;
                const bool out = alive & ((R.w1[i] - (R.b[i] - R.w2[i])) < -1e-4);
                if (out) {
                    R.state[i] = ROW_DEAD;
                    for (int k = 0; k < d; ++k) R.A[i * d + k] = 0.0;
                    R.b[i] = 0.0;
                    R.s[i] = 0.0;
                }
                cnt += (alive & !out) ? 1 : 0;
            }
            neq = wave_sum_i(cnt);
            nlp += 2 * d;
            if (lpfail) flags |= RF_LPFAIL;
            if (neq <= d + 1) { flags |= RF_EARLY; stage = 0; }
            else stage = 2;
            __syncthreads();
        }
        // ---------------------------------------------------------------- F2: presolve, then one LP per row it leaves (:1142-1160)
        if (stage == 2) {
            nlp += neq;
            const bool presolve = abs_tol < 0.04;
            for (int k = lane; presolve && k < m; k += 64) {   // (see f2_presolve in plp_reduce_r_impl.hpp: same witnesses)
                if (R.state[k] == ROW_DEAD) continue;
                double gkk = 0.0;
                for (int c = 0; c < d; ++c) gkk = fma(R.A[k * d + c], R.A[k * d + c], gkk);
                const double bk = R.b[k];
                const double sk = fmax(bk - R.s[k], 0.0);
                const double tau = abs_tol + 1e-9 * (1.0 + fabs(bk) + sk);
                const double skt = sk + tau;
                if (!((gkk > 0.0) & (tau < 0.05))) continue;
                bool ok = true;
                int jb = 0;
                for (int i = 0; i < m; ++i) {
                    double gik = 0.0;
                    for (int c = 0; c < d; ++c) gik = fma(R.A[i * d + c], R.A[k * d + c], gik);
                    const double si = fmax(R.b[i] - R.s[i], 0.0);
                    const bool fine = (i == k) | (skt * gik <= si * gkk);
                    ok = ok & fine;
                    jb = fine ? jb : i;
                }
                if (!ok) {  // second witness: up to the blocking row jb, then along its plane
                    double gjk = 0.0, gjj = 0.0;
                    for (int c = 0; c < d; ++c) {
                        gjk = fma(R.A[jb * d + c], R.A[k * d + c], gjk);
                        gjj = fma(R.A[jb * d + c], R.A[jb * d + c], gjj);
                    }
                    const double sj = fmax(R.b[jb] - R.s[jb], 0.0);
                    const double rho = gjk / gjj;
                    const double t1 = (sj / gjk) * (1.0 - 0x1p-40);
                    const double akd = fma(-rho, gjk, gkk);
                    const double t2 = fma(-t1, gkk, skt) / akd;
                    const double c1 = t1 + t2, c2 = t2 * rho;
                    ok = (gjk > 0.0) & (akd > 1e-12 * gkk) & (t2 >= 0.0) & (t2 < 1e300);
                    for (int i = 0; ok && i < m; ++i) {
                        double gik = 0.0, gij = 0.0;
                        for (int c = 0; c < d; ++c) {
                            gik = fma(R.A[i * d + c], R.A[k * d + c], gik);
                            gij = fma(R.A[i * d + c], R.A[jb * d + c], gij);
                        }
                        const double si = fmax(R.b[i] - R.s[i], 0.0);
                        const double lhs = fma(-c2, gij, c1 * gik);
                        ok = ok & ((i == k) | (lhs <= si));
                    }
                }
                if (ok) R.state[k] = ROW_SETTLED;
            }
            __syncthreads();
            // settled rows: the reference's in-place round trip of h[k] right away (see f2_presolve)
            for (int k = lane; k < m; k += 64)
                if (R.state[k] == ROW_SETTLED) R.b[k] = (R.b[k] + 0.1) - 0.1;
            __syncthreads();
            for (int kr = 0; kr < m; ++kr) {
                if (R.state[kr] != ROW_LIVE) continue;   // (wave-uniform: an LDS word)
                double cxc = 0.0;
                for (int c = 0; c < d; ++c) cxc = fma(-R.A[kr * d + c], R.xc[c], cxc);
                __syncthreads();
                if (lane == 0) R.b[kr] = R.b[kr] + 0.1;  // h[k] += 0.1 in place (:1149)
                __syncthreads();
                red_setup_lp(S, D, R, m, d, neq, lane, lane < d ? -R.A[kr * d + (lane < d ? lane : 0)] : 0.0);
                while (S.mode != M_DONE) lds_step(S, D, lane);
                const double fun = cxc - S.negz;  // c.xc + zeta, zeta = -negz
                const double hk = R.b[kr] - 0.1;  // (:1151)
                __syncthreads();
                if (lane == 0) R.b[kr] = hk;
                const double obj = -fun - hk;     // (:1156)
                const bool keepk = ((S.status == ST_OPT) & (obj > abs_tol)) | (S.status == ST_UNBND);
                if (!keepk && lane == 0) R.state[kr] = ROW_DROPPED;
                __syncthreads();
            }
            flags |= RF_MINREP;
        }
        // ---------------------------------------------------------------- results
        __syncthreads();
        for (int w = 0; w < W; ++w) {
            const int i = w * 64 + lane;
            const bool kept = (flags & (RF_EARLY | RF_MINREP)) && i < m && R.state[i] != ROW_DEAD && R.state[i] != ROW_DROPPED;
            const unsigned long long word = __ballot(kept);
            if (lane == 0) keep_out[p * W + w] = word;
        }
        if (lane == 0) {
            flags_out[p] = flags;
            nlp_out[p] = nlp;
            r_out[p] = ball ? rr : 0.0;
        }
        if (lane < d) xc_out[p * d + lane] = ball ? xcl : qnan;
        __syncthreads();
    }
}

// LDS bytes one LP of (m_max, columns nc) needs; 0 when it does not fit a workgroup (160 KB per CU on gfx950)
size_t lds_lp_bytes(int m_max, int nc) {
    const int ld = nc | 1;
    const size_t need = (size_t)m_max * ld * 8 + (size_t)m_max * 24 + 3 * LDS_MAXC * 8 + LDS_MAXC * 4 + 64;
    return need <= 160 * 1024 ? ((need + 15) & ~(size_t)15) : 0;
}

static int lds_grid(long long B, size_t smem) {
    // resident wavefronts: limited by LDS per CU and 8 single-wave workgroups per SIMD-set; the loop strides the rest
    long long per_cu = (long long)(160 * 1024 / smem);
    if (per_cu > 32) per_cu = 32;
    if (per_cu < 1) per_cu = 1;
    long long blocks = 256 * per_cu * 4;
    if (blocks > B) blocks = B;
    return (int)(blocks < 1 ? 1 : blocks);
}

int launch_lp_lds(long long B, int m_max, int n, const double* c, const double* G, const double* h, const int* mrows,
                  double* x, double* fun, int* status, int* iters, hipStream_t st) {
    if (n < 1 || n > MAX_D + 1 || m_max < 0) return 2;
    const size_t smem = lds_lp_bytes(m_max < 1 ? 1 : m_max, n + 1);
    if (!smem) return 2;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lp_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
    hipLaunchKernelGGL(lp_lds_kernel, dim3(lds_grid(B, smem)), dim3(64), smem, st, B, m_max, n, c, G, h, mrows, x, fun,
                       status, iters);
    return 0;
}

int launch_cheby_lds(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double* r,
                     double* xc, int* status, hipStream_t st) {
    if (d < 1 || d > MAX_D || m_max < 0) return 2;
    const size_t smem = lds_lp_bytes(m_max < 1 ? 1 : m_max, d + 1);
    if (!smem) return 2;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cheby_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cheby_lds_kernel, dim3(lds_grid(B, smem)), dim3(64), smem, st, B, m_max, d, A, b, mrows, r, xc,
                       status);
    return 0;
}

int launch_cheby_gather_lds(int d, int m_cap, long long nlp, const int* off, const int* rows, const int* sel,
                            const double* A, const double* b, double* out, hipStream_t st) {
    if (d < 1 || d > MAX_D || nlp < 1) return nlp < 1 ? 0 : 2;
    const size_t smem = lds_lp_bytes(m_cap < 1 ? 1 : m_cap, d + 1);
    if (!smem) return 2;
    if (smem > 48 * 1024)   // per device and cheap: set on every launch that needs it, like the other LDS launchers
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cheby_gather_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(cheby_gather_lds_kernel, dim3(lds_grid(nlp, smem)), dim3(64), smem, st, nlp, m_cap, d, off, rows, sel,
                       A, b, out);
    return 0;
}

// fused reduce() beyond 64 rows (reduce_lds_kernel); returns 2 when the polytope does not fit the CU's LDS
int launch_reduce_lds(long long B, int m_max, int d, const double* A, const double* b, const int* mrows, double abs_tol,
                      unsigned long long* keep, int* flags, double* r, double* xc, int* nlp, hipStream_t st) {
    if (d < 1 || d > MAX_D || m_max < 1) return 2;
    const size_t dict = lds_lp_bytes(m_max, d + 1);
    if (!dict) return 2;
    const size_t rows = ((size_t)m_max * (d + 4) * 8 + 16 * 8 + (size_t)m_max * 4 + 15) & ~(size_t)15;
    const size_t smem = dict + rows;
    if (smem > 160 * 1024) return 2;
    const int W = (m_max + 63) / 64;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(reduce_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(reduce_lds_kernel, dim3(lds_grid(B, smem)), dim3(64), smem, st, B, m_max, d, W, A, b, mrows, abs_tol,
                       keep, flags, r, xc, nlp, dict);
    return 0;
}

}  // namespace plp
